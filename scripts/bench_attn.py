"""UNet attention shapes (SD-v1.5, CFG batch 2): self- and cross-attention at the three resolutions, graph-timed.
With an argument ("cross64" | "self64" | ...) it just loops that case (for rocprofv3 --pmc)."""
import sys, torch
import os as _os
DT = torch.bfloat16 if _os.environ.get("UNET_DTYPE", "f16") == "bf16" else torch.float16   # the diffusion engines\' format (f16 by default)
from spider_amd import ops
dev = torch.device("cuda:0")
CASES = {"self64": (4096, 4096, 8, 40), "cross64": (4096, 77, 8, 40), "self32": (1024, 1024, 8, 80), "cross32": (1024, 77, 8, 80),
         "self16": (256, 256, 8, 160), "cross16": (256, 77, 8, 160), "sdxl_cross48": (2304, 77, 10, 64), "sdxl_cross24": (576, 77, 20, 64)}


def mk(case):
    Lq, Lk, H, d = CASES[case]
    C = H * d
    q = torch.randn(2, Lq, C, device=dev).to(DT)
    kv = torch.randn(2, Lk, 2 * C, device=dev).to(DT)
    return (lambda: ops.attention(q, kv[..., :C], kv[..., C:], H)), 4.0 * 2 * Lq * Lk * C


if len(sys.argv) > 1:
    f, _ = mk(sys.argv[1])
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    sys.exit(0)
for case in CASES:
    f, fl = mk(case)
    for _ in range(3):
        f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 60
    print(f"{case:14s} {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s useful")
