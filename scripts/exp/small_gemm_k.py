"""How much of a small GEMM's time is its K loop? 512 x 1280 x K and 8192 x 320 x K for K = 64 .. 2560 (graph-timed)."""
import torch
from spider_amd import ops
dev = torch.device("cuda:0")
def t(f, n=40):
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for M, N in [(512, 1280), (8192, 320), (2048, 640)]:
    row = []
    for K in (64, 320, 640, 1280, 2560):
        # rotate over 8 weight tensors so that W comes from HBM / MALL as in the UNet, not from L2
        Ws = [(torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(8)]
        A = torch.randn(M, K, device=dev).bfloat16()
        i = [0]
        def f():
            i[0] = (i[0] + 1) % 8
            return ops.gemm(A, Ws[i[0]])
        row.append(f"K={K}: {t(f):5.1f}")
    print(f"{M}x{N}: " + "  ".join(row), flush=True)
