"""Randomised differential check of whole UNet evaluations (spider_amd.unet.UNetEngine / unet3d.UNet3DEngine on the HIP kernels) against
the fp32 CPU oracle on the tiny configurations, over what the fixed tests do not enumerate: odd and non-square latent sizes (the
diffusers `upsample_size` rule, GroupNorm chunking, fused-path gating all depend on them), CFG batch 1..6, frame counts, both 16-bit
formats, the fp32 residual stream on / off, eager and hipGraph replay (must be bit-identical). Not part of the suite; on the GPU box:

    PYTHONPATH=. python scripts/fuzz_unet.py [cases] [seed]          # default 60 cases

One line per failing case with everything needed to reproduce it; exit code 1 if anything failed."""
import random
import sys

import torch

from oracle.unet import UNetCfg, UNetOracle, random_unet_weights
from oracle.unet3d import UNet3DCfg, UNet3DOracle, random_unet3d_weights
from spider_amd.unet import UNetConfig, UNetEngine
from spider_amd.unet3d import UNet3DConfig, UNet3DEngine

dev = torch.device("cuda:0")
TOL = {torch.float16: 3.5e-3, torch.bfloat16: 2.6e-2}     # typical 1.8e-3 / 1.7e-2; the tiniest maps (40 pixels) scatter up to 3.1e-3


def rel(a, b):
    return float((a.float().cpu() - b).norm() / b.norm())


def case_unet2d(r, g):
    kind = r.choice(["tiny", "tiny8", "sdxl", "audio"])
    ocfg = {"tiny": UNetCfg.tiny, "tiny8": UNetCfg.tiny8, "sdxl": lambda: UNetCfg.tiny(True), "audio": UNetCfg.tiny_audio}[kind]()
    dt = r.choice([torch.float16, torch.bfloat16])
    s32 = r.random() < 0.5
    B2 = r.choice([1, 2, 2, 4, 6])
    levels = len(ocfg.block_out)
    h, w = r.randint(2 ** (levels - 1), 40), r.randint(2 ** (levels - 1), 40)       # odd sizes included
    t = r.choice([1, 21, 500, 981])
    seed = r.randint(0, 1 << 30)
    desc = f"unet2d kind={kind} dt={dt} stream32={s32} B2={B2} h={h} w={w} t={t} wseed={seed}"
    wts = random_unet_weights(ocfg, seed=seed)
    x = torch.randn(B2, ocfg.in_ch, h, w, generator=g).bfloat16().float()
    enc = added = cl = None
    if kind == "audio":
        cl = torch.nn.functional.normalize(torch.randn(B2, ocfg.class_in, generator=g), dim=-1).bfloat16().float()
    else:
        enc = torch.randn(B2, r.choice([77, 77, 16, 80]), ocfg.cross_dim, generator=g).bfloat16().float()
    if kind == "sdxl":
        added = dict(text_embeds=torch.randn(B2, 64, generator=g).bfloat16().float(),
                     time_ids=torch.tensor([[h * 8, w * 8, 0, 0, h * 8, w * 8]] * B2, dtype=torch.float32))
    ref = UNetOracle(ocfg, wts).forward(x, torch.tensor(t), enc, added, cl)
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), wts, dev, dtype=dt, stream32=s32)
    eng.prepare(torch.tensor([t]), None if enc is None else enc.to(dev), added, None if cl is None else cl.to(dev))
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev).to(dt)
    eager = eng.step(xn, 0, use_graph=False).permute(0, 3, 1, 2).clone()
    graph = eng.step(xn, 0, use_graph=True).permute(0, 3, 1, 2)
    why = "" if torch.equal(eager, graph) else "graph replay differs from eager"
    if not torch.isfinite(eager).all():
        why = "non-finite output"
    return desc, rel(eager, ref), why, dt


def case_unet3d(r, g):
    ocfg = UNet3DCfg.tiny()
    dt = r.choice([torch.float16, torch.bfloat16])
    s32 = r.random() < 0.5
    B2, frames = r.choice([1, 2, 2, 4]), r.choice([1, 2, 3, 4, 5, 8, 16])
    h, w = r.randint(4, 24), r.randint(4, 24)
    t = r.choice([1, 301, 976])
    seed = r.randint(0, 1 << 30)
    desc = f"unet3d dt={dt} stream32={s32} B2={B2} frames={frames} h={h} w={w} t={t} wseed={seed}"
    wts = random_unet3d_weights(ocfg, seed)
    x = torch.randn(B2, 4, frames, h, w, generator=g).bfloat16().float()
    enc = torch.randn(B2, 77, ocfg.cross_dim, generator=g).bfloat16().float()
    ref = UNet3DOracle(ocfg, wts).forward(x, torch.tensor(t), enc)
    eng = UNet3DEngine(UNet3DConfig(**ocfg.__dict__), wts, dev, dtype=dt, stream32=s32)
    eng.prepare(torch.tensor([t]), enc.to(dev), frames=frames)
    xn = x.permute(0, 2, 3, 4, 1).reshape(B2 * frames, h, w, 4).contiguous().to(dev).to(dt)
    eager = eng.step(xn, 0, use_graph=False).clone()
    graph = eng.step(xn, 0, use_graph=True)
    why = "" if torch.equal(eager, graph) else "graph replay differs from eager"
    got = eager.view(B2, frames, h, w, -1).permute(0, 4, 1, 2, 3)
    if not torch.isfinite(got).all():
        why = "non-finite output"
    return desc, rel(got, ref), why, dt


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    r, g = random.Random(seed), torch.Generator().manual_seed(seed)
    bad, worst = 0, {}
    for i in range(n):
        fn = r.choice([case_unet2d, case_unet2d, case_unet2d, case_unet3d])
        try:
            desc, e, why, dt = fn(r, g)
        except Exception as ex:
            print(f"RAISED {fn.__name__} case {i}: {type(ex).__name__}: {str(ex).splitlines()[0][:300]}", flush=True)
            bad += 1
            continue
        key = (fn.__name__[5:], str(dt)[6:])
        worst[key] = max(worst.get(key, 0.0), e)
        if why or not (e < TOL[dt]):
            bad += 1
            print(f"FAIL case {i}: {desc}: rel {e:.3e} {why}", flush=True)
    print(f"fuzz_unet: {n} cases, {bad} failed (seed {seed}); worst relative L2: " + ", ".join(f"{k[0]}/{k[1]} {v:.2e}" for k, v in sorted(worst.items())))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
