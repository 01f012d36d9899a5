"""VAE-decoder conv shapes at the 512^2 / 256^2 levels under the default dispatch or a forced tile (SPIDER_GEMM_TILE)."""
import torch
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
for hw, cin, cout in [(512, 128, 128), (512, 256, 128), (256, 256, 256), (256, 512, 256), (128, 512, 512)]:
    x = torch.randn(1, hw, hw, cin, device=dev).bfloat16(); w = (torch.randn(cout, 3, 3, cin, device=dev) * 0.02).bfloat16()
    us = t(lambda: ops.conv2d(x, w))
    print(f"{hw}^2 {cin}->{cout}: {us:8.1f} us  {2 * hw * hw * cout * 9 * cin / us / 1e6:.0f} TF/s", flush=True)
