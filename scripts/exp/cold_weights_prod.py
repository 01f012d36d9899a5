"""Experiment (round 4): what cold weights cost the UNet's weight-streaming problems ON THE PRODUCTION DISPATCH (split-K, tile-major
copies, LDS-DMA kernels as ops picks them): each shape is launched 96 times in a hipGraph, walking N distinct weight tensors round-robin --
N chosen so that the set is (cold) 480 MB: every launch streams its weights from HBM, as in a real UNet step (1.47 GB of weights per
step); (mall) 150 MB: past the L2s, inside the 256 MB Infinity Cache; (warm) one tensor."""
import json, sys
import torch
from spider_amd import ops

CASES = [  # (tag, M rows, N, K, conv Cin or 0, hw)
    ("c8 1280>1280", 128, 1280, 11520, 1280, 8), ("c8 2560>1280", 128, 1280, 23040, 2560, 8),
    ("c16 1280>1280", 512, 1280, 11520, 1280, 16), ("c16 2560>1280", 512, 1280, 23040, 2560, 16),
    ("u16 out", 512, 1280, 1280, 0, 0), ("u16 ff2", 512, 1280, 5120, 0, 0), ("u8 out", 128, 1280, 1280, 0, 0), ("u8 ff2", 128, 1280, 5120, 0, 0),
    ("c32 640>640", 2048, 640, 5760, 640, 32), ("c64 320>320", 8192, 320, 2880, 320, 64),
]
dev = torch.device("cuda:0")
DT = torch.float16
print(f"{'shape':16s} {'warm':>8s} {'mall':>8s} {'cold':>8s}   us per launch (production dispatch, graph of 96)   weights MB")
for tag, M, N, K, cin, hw in CASES:
    row = []
    for mode, total in (("warm", 0), ("mall", 150e6), ("cold", 480e6)):
        wbytes = N * K * 2
        nw = 1 if mode == "warm" else max(2, min(96, int(total / wbytes) + 1))
        if cin:
            x = torch.randn(M // (hw * hw), hw, hw, cin, device=dev).to(DT)
            Ws = [ops.mark_weight((torch.randn(N, 3, 3, cin, device=dev) * 0.02).to(DT)) for _ in range(nw)]
            f = lambda w: ops.conv2d(x, w)
        else:
            A = torch.randn(M, K, device=dev).to(DT)
            Ws = [ops.mark_weight((torch.randn(N, K, device=dev) * 0.02).to(DT)) for _ in range(nw)]
            f = lambda w: ops.gemm(A, w)
        for w in Ws:
            f(w)                       # builds the tile-major copies outside capture
        torch.cuda.synchronize()
        n = 96
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(n):
                f(Ws[i % nw])
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record(); e1.synchronize()
        row.append(e0.elapsed_time(e1) * 1e3 / (3 * n))
        del Ws, g
        torch.cuda.empty_cache()
    print(f"{tag:16s} {row[0]:8.1f} {row[1]:8.1f} {row[2]:8.1f}   {N * K * 2 / 1e6:6.1f}", flush=True)
