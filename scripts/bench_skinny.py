"""Tuning aid: weight-streaming GEMV / skinny GEMM of the decode step at Qwen2.5-7B shapes, per shape and batch, as
microseconds and TB/s of weight stream (each call walks through enough weight copies to defeat the 256 MiB Infinity Cache).
    python scripts/bench_skinny.py [B ...]"""
import sys, torch
from spider_amd import ops
dev = torch.device("cuda:0")
Bs = [int(a) for a in sys.argv[1:]] or [1, 8]
SHAPES = [("qkv", 4608, 3584, False), ("o", 3584, 3584, False), ("gate_up", 18944, 3584, True), ("down", 3584, 18944, False)]
for B in Bs:
    line = [f"B={B}"]
    for tag, N, K, gu in SHAPES:
        rows = 2 * N if gu else N
        nbytes = rows * K * 2
        ncopy = max(2, (320 << 20) // nbytes + 1)
        Ws = [(torch.randn(rows, K, device=dev) * 0.02).bfloat16() for _ in range(ncopy)]
        x = torch.randn(B, K, device=dev).bfloat16()
        out = torch.empty(B, N, device=dev, dtype=torch.bfloat16)
        f = (lambda W: ops.gemv_swiglu(W, x, out=out)) if gu else (lambda W: ops.gemv(W, x, out=out))
        for W in Ws[:2]:
            f(W)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        reps = max(1, 24 // ncopy)
        with torch.cuda.graph(g):
            for _ in range(reps):
                for W in Ws:
                    f(W)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (2 * reps * ncopy)
        line.append(f"{tag} {us:6.1f} us {nbytes / us / 1e6:5.2f} TB/s")
        del Ws
    print(" | ".join(line), flush=True)
