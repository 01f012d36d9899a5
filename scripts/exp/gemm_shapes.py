"""Time ops.gemm on given M,N,K triples under the current SPIDER_GEMM_TILE: python gemm_shapes.py M,N,K [M,N,K ...]"""
import sys, torch
from spider_amd import ops
dev = torch.device("cuda:0")
def t(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for a in sys.argv[1:]:
    M, N, K = (int(v) for v in a.split(","))
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    us = t(lambda: ops.gemm(A, W))
    print(f"{M}x{N}x{K}: {us:8.1f} us  {2 * M * N * K / us / 1e6:.0f} TF/s", flush=True)
