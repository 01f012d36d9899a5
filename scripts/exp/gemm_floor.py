"""Experiment: where does the ~7.7 us floor of the small GEMMs come from? Times (graph of n dependent calls, and graph of n
INDEPENDENT calls on different outputs) for shapes that isolate the launch, the prologue / epilogue and the K loop."""
import torch
from spider_amd import ops
dev = torch.device("cuda:0")
def t(f, n=40):
    for _ in range(3): f(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n): f(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (2 * n)
for M, N, K in [(64, 64, 64), (64, 64, 320), (64, 64, 1280), (8192, 320, 64), (8192, 320, 320), (8192, 320, 1280), (128, 3840, 64), (128, 3840, 1280),
                (512, 1280, 1280), (2048, 640, 640)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    res = torch.randn(M, N, device=dev).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(40)]
    plain = t(lambda i: ops.gemm(A, W, out=outs[i]))
    epi = t(lambda i: ops.gemm(A, W, bias=b, res=res, out=outs[i]))
    chain = t(lambda i: ops.gemm(outs[i - 1][:, :K].contiguous() if False else A, W, out=outs[0]))   # same output: WAW-dependent chain
    print(f"M={M:5d} N={N:5d} K={K:5d}: plain {plain:6.2f} us  +bias+res {epi:6.2f} us  same-out {chain:6.2f} us", flush=True)
# empty-ish kernel for the launch floor
x = torch.zeros(64, device=dev)
print("torch tiny add_ per launch:", round(t(lambda i: x.add_(1.0)), 2), "us")
