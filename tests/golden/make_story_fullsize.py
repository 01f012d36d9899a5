"""Fixture generator (build container, CPU): ONE full-size evaluation of BASELINE configs[2]'s actual arithmetic through the fp32
oracle -- the SDXL UNet (2.57 B parameters, random weights of seed 4) at the 64^2 latent (north_star's 512^2) with FreeU(0.6, 0.4,
1.1, 1.2) and StoryDiffusion's consistent self-attention on the 36 up-block attn1 processors (Comic_Generation.py:94-118, 353-371,
440): the WRITE phase at CFG batch 8 (4 identity panels x cond / uncond, cur_step = 5, the masked-attention branch with fixed keep
vectors) followed by the READ phase at CFG batch 2 against the bank the write evaluation left (cur_step = 5, :104-109).

    python tests/golden/make_story_fullsize.py        ->  tests/golden/oracle_story_sdxl_fullsize.npz   (~ 2 min of host time)

The test (tests/test_fullsize_parity.py::test_story_sdxl_fullsize_write_then_read_matches_oracle_fixture) regenerates the seeded
inputs with `story_inputs()` and runs the HIP engine in the mode init_story_generation loads (f16 + fp32 residual stream).
PARITY UNPINNED upstream like every UNet-side oracle: diffusers is absent from the image and from /root/reference."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

HH = WW = 512            # image size: latent 64 x 64; consistent attention at N = 1024 (32^2) and N = 256 (16^2) tokens
WEIGHTS_SEED, STEP = 4, 5
FREEU = (0.6, 0.4, 1.1, 1.2)


def story_inputs():
    """seeded inputs of the two evaluations + the injected random numbers (coins above both thresholds: the masked branch)"""
    g = torch.Generator().manual_seed(21)
    unis = [torch.rand(5 * 1024, generator=g) for _ in range(8)]       # regen_masks draws (5 N1) then (5 N4) uniforms per step
    out = {"uniforms": unis}
    for phase, B2 in (("write", 8), ("read", 2)):
        out[phase] = dict(x=torch.randn(B2, 4, HH // 8, WW // 8, generator=g).bfloat16().float(),
                          enc=torch.randn(B2, 77, 2048, generator=g).bfloat16().float(),
                          text_embeds=torch.randn(B2, 1280, generator=g).bfloat16().float(),
                          time_ids=torch.tensor([[HH, WW, 0, 0, HH, WW]] * B2, dtype=torch.float32),
                          t=torch.tensor(801))
    return out


class OracleHook:
    """oracle.story.ProcessorOracle behind UNetOracle.attn_hook (one processor per up-block attn1, Comic_Generation.py:353-371)"""

    def __init__(self, st, w):
        from oracle import story as ostory
        self.st, self.w, self.os, self.procs = st, w, ostory, {}

    def wants(self, name):
        return name.startswith("up_blocks") and name.endswith("attn1")

    def __call__(self, unet, name, y, heads):
        w = self.w
        aw = self.os.AttnWeights(w[name + ".to_q.weight"], w[name + ".to_k.weight"], w[name + ".to_v.weight"],
                                 w[name + ".to_out.0.weight"], w[name + ".to_out.0.bias"], heads)
        return self.procs.setdefault(name, self.os.ProcessorOracle())(self.st, aw, y)


def main():
    from oracle import story as ostory
    from oracle.unet import UNetCfg, UNetOracle, random_unet_weights
    torch.set_num_threads(os.cpu_count() or 8)
    ocfg = UNetCfg.sdxl()
    w = random_unet_weights(ocfg, seed=WEIGHTS_SEED)
    inp = story_inputs()
    ui = iter(inp["uniforms"])
    unet = UNetOracle(ocfg, w)
    unet.freeu = FREEU
    n_proc = sum(1 for k in w if k.startswith("up_blocks") and k.endswith(".attn1.to_q.weight"))
    assert n_proc == 36, n_proc
    st = ostory.StoryState(total_count=n_proc, height=HH, width=WW, coin=lambda: 0.95, uniforms=lambda n: next(ui)[:n])
    st.regen_masks()
    unet.attn_hook = OracleHook(st, unet.w)
    res = {}
    for phase in ("write", "read"):
        c = inp[phase]
        st.write, st.cur_step, st.attn_count = phase == "write", STEP, 0
        out = unet.forward(c["x"], c["t"], c["enc"], dict(text_embeds=c["text_embeds"], time_ids=c["time_ids"]))
        res[phase] = out.numpy()
        print(phase, "done: output norm", float(out.norm()), "cur_step ->", st.cur_step, flush=True)
    # what the consistent attention changed at all (an engine that ran plain attention would sit this far from the fixture)
    st2 = ostory.StoryState(total_count=n_proc, height=HH, width=WW, coin=lambda: 0.0, uniforms=lambda n: torch.rand(n))
    st2.regen_masks()
    st2.write, st2.cur_step = True, STEP
    unet.attn_hook = OracleHook(st2, unet.w)
    c = inp["write"]
    plain = unet.forward(c["x"], c["t"], c["enc"], dict(text_embeds=c["text_embeds"], time_ids=c["time_ids"]))
    eff = float((torch.from_numpy(res["write"]) - plain).norm() / plain.norm())
    print("effect of the masked consistent attention on the write evaluation: rel", eff)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_story_sdxl_fullsize.npz"), write=res["write"], read=res["read"],
                        weights_seed=WEIGHTS_SEED, step=STEP, freeu=np.array(FREEU), consistent_effect=eff,
                        x_write_sum=float(inp["write"]["x"].double().sum()), x_read_sum=float(inp["read"]["x"].double().sum()))


if __name__ == "__main__":
    main()
