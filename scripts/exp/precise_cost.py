"""Round 5: what the precise modes cost -- ms per graph-replayed evaluation of each diffusion UNet at its BASELINE size, f16, for
stream32 / precise=1 / precise=2 (UNetEngine / UNet3DEngine). Alternating rounds in one process, median of 5."""
import os, sys, time
import torch
from spider_amd.unet import UNetConfig, UNetEngine
from spider_amd.unet3d import UNet3DConfig, UNet3DEngine

dev = torch.device("cuda:0")
which = sys.argv[1:] or ["sd15", "sdxl", "zeroscope"]


MODES = tuple(int(m) for m in os.environ.get("PRECISE_MODES", "0,1,2").split(","))      # (a subset: quick A/B of one mode)


def bench(make, prep, xs, modes=MODES):
    engs = {}
    for m in modes:
        e = make(m)
        prep(e)
        engs[m] = e
    t = {m: [] for m in modes}
    for m in modes:
        engs[m].step(xs[bool(m)], 0)
    torch.cuda.synchronize()
    for _ in range(5):
        for m in modes:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                engs[m].step(xs[bool(m)], 0)
            e1.record(); e1.synchronize()
            t[m].append(e0.elapsed_time(e1) / 5)
    return {m: sorted(v)[2] for m, v in t.items()}


g = torch.Generator(device=dev).manual_seed(0)
for name in which:
    if name == "sd15":
        cfg = UNetConfig.sd15()
        enc = torch.randn(2, 77, 768, generator=g, device=dev).half()
        x32 = torch.randn(2, 64, 64, 4, generator=g, device=dev)
        r = bench(lambda m: UNetEngine.random_init(cfg, dev, seed=1, dtype=torch.float16, stream32=True, precise=m),
                  lambda e: e.prepare(torch.tensor([500]), enc), {False: x32.half(), True: x32})
    elif name == "sdxl":
        cfg = UNetConfig.sdxl()
        enc = torch.randn(2, 77, 2048, generator=g, device=dev).half()
        added = dict(text_embeds=torch.randn(2, 1280, generator=g, device=dev), time_ids=torch.tensor([[512, 512, 0, 0, 512, 512]] * 2, dtype=torch.float32))
        x32 = torch.randn(2, 64, 64, 4, generator=g, device=dev)
        r = bench(lambda m: UNetEngine.random_init(cfg, dev, seed=1, dtype=torch.float16, stream32=True, precise=m),
                  lambda e: e.prepare(torch.tensor([500]), enc, added), {False: x32.half(), True: x32})
    else:
        cfg = UNet3DConfig.zeroscope()
        caps = int(os.environ.get("VIDEO_CAPS", "1"))          # captions per call (SpiderDecoder.generate_batch batches them)
        enc = torch.randn(2 * caps, 77, cfg.cross_dim, generator=g, device=dev).half()
        x32 = torch.randn(2 * caps * 16, 40, 72, 4, generator=g, device=dev)
        r = bench(lambda m: UNet3DEngine.random_init(cfg, dev, seed=1, dtype=torch.float16, stream32=True, precise=m),
                  lambda e: e.prepare(torch.tensor([500]), enc, frames=16), {False: x32.half(), True: x32})
    print(f"{name:10s} ms per evaluation: " + "   ".join((f"stream32 {r[m]:.3f}" if m == 0 else f"precise={m} {r[m]:.3f}" + (f" ({r[m] / r[0]:.2f}x)" if 0 in r else "")) for m in MODES), flush=True)
    torch.cuda.empty_cache()
