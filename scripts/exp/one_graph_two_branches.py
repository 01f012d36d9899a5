"""Experiment (round 4): does ONE hipGraph with two parallel branches share the chip better than TWO graphs replayed on two streams?

The two-stream schedule of SpiderFreeInfer pays ~5 us on every launch of the LLM decode chain while the decoder stream is active
("the command processor's handling of two active queues"). Here the same work -- 2 decode tokens of the
Qwen2.5-7B-shaped engine (342 launches) beside 1 SD-v1.5 UNet evaluation at CFG batch 2 (332 launches), the ratio the headline step
has -- is timed as
  (a) the decode tokens alone, (b) the UNet evaluation alone,
  (c) two graphs replayed on two streams from two host threads (the product's form),
  (d) ONE graph captured with the UNet evaluation forked onto a side stream beside the two decode steps (parallel branches).
Times are per replay pair, HIP events on the capture / LLM stream, 30 replays each."""
import os
import threading

import torch

from spider_amd import ops
from spider_amd.llm import LlamaEngine, LLMConfig
from spider_amd.schedulers import PNDMScheduler
from spider_amd.unet import UNetConfig, UNetEngine

dev = torch.device("cuda:0")
DT = torch.float16
N = 30


def timed(fn, n=N):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    cfg = LLMConfig.qwen25_7b()
    eng = LlamaEngine.random_init(cfg, dev, max_batch=1, max_len=1536 + 256, seed=0)
    ids = torch.randint(3, cfg.vocab, (1, 1536), generator=torch.Generator().manual_seed(1))
    hd = eng.prefill_begin(input_ids=ids, max_new_tokens=200, eos_token_id=[])
    st = hd.st
    unet = UNetEngine.random_init(UNetConfig.sd15(), dev, seed=1, dtype=DT, stream32=True)
    g = torch.Generator(device=dev).manual_seed(0)
    enc = torch.randn(2, 77, 768, generator=g, device=dev).to(DT)
    unet.prepare(PNDMScheduler().set_timesteps(40), enc)
    x2 = ops.latent_to_nhwc(torch.randn(1, 4, 64, 64, generator=g, device=dev), reps=2, dtype=DT)
    unet.tproj_cur.copy_(unet.tproj_steps[0])

    def two_tokens():
        eng._decode_step(st)
        eng._decode_step(st)

    def reset():            # keep the context length fixed from measurement to measurement
        st["pos"].fill_(1536); st["slot"].fill_(1536); st["kv_end"].fill_(1537); st["n_hist"].fill_(1)

    s_main = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s_main):
        two_tokens()                                        # warm-up outside capture (the UNet in the workspace scope it is captured in)
        with ops.workspace_scope("u"):
            unet._forward(x2)
        torch.cuda.synchronize()
        g_dec, g_unet, g_both = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        reset()
        with torch.cuda.graph(g_dec, stream=s_main):
            two_tokens()
        with ops.workspace_scope("u"):
            with torch.cuda.graph(g_unet, stream=s_main):
                unet._forward(x2)
        reset()
        with torch.cuda.graph(g_both, stream=s_main):
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(s_main)
            with torch.cuda.stream(side), ops.workspace_scope("u"):
                unet._forward(x2)
            two_tokens()
            s_main.wait_stream(side)
        torch.cuda.synchronize()

        def run_dec():
            reset(); g_dec.replay()
        def run_both():
            reset(); g_both.replay()
        a = timed(run_dec)
        b = timed(lambda: g_unet.replay())
        d = timed(run_both)

    # (c) two graphs, two streams, two host threads: the UNet replays on sU while the decode tokens replay on sL
    sL, sU = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)
    res = {}

    def u_side(n):
        with torch.cuda.stream(sU):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                g_unet.replay()
            e1.record(); e1.synchronize()
            res["u"] = e0.elapsed_time(e1) / n

    torch.cuda.synchronize()
    th = threading.Thread(target=u_side, args=(N,))
    with torch.cuda.stream(sL):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        th.start()
        e0.record()
        for _ in range(N):
            reset(); g_dec.replay()
        e1.record(); e1.synchronize()
        res["l"] = e0.elapsed_time(e1) / N
    th.join()
    torch.cuda.synchronize()
    print(f"(a) 2 decode tokens alone            {a:7.3f} ms   ({a / 2:.3f} per token)")
    print(f"(b) 1 UNet evaluation alone          {b:7.3f} ms")
    print(f"(c) two graphs on two streams        decode pair {res['l']:7.3f} ms, UNet evaluation {res['u']:7.3f} ms  (both loops run {N} replays concurrently)")
    print(f"(d) ONE graph, two parallel branches {d:7.3f} ms per replay (2 tokens + 1 UNet evaluation)   serial sum (a)+(b) = {a + b:.3f}")


if __name__ == "__main__":
    main()
