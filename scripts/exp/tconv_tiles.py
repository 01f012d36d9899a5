"""Round 5: the (3,1,1) temporal convs of the zeroscope UNet3D (conv_ex on the view [B, F, HW, C] with a 3 x 1 kernel) per forced tile
(SPIDER_GEMM_TILE is read once per process: run once per setting). Graph of 20 launches, median of 5."""
import os, sys, torch
from spider_amd import ops
dev = torch.device("cuda:0")
tile = os.environ.get("SPIDER_GEMM_TILE", "auto")
for C, HW in ((320, 2880), (640, 720), (1280, 180), (1280, 45)):
    x = torch.randn(2, 16, HW, C, device=dev).half()
    w = (torch.randn(C, 3, 1, C, device=dev) * 0.02).half()
    b = torch.randn(C, device=dev).half()
    if os.environ.get("STREAMS", "0") != "0":      # the 4th conv of a temporal block: fp32 identity added, fp32 master written
        r32 = torch.randn(2, 16, HW, C, device=dev)
        f = lambda: ops.conv_ex(x, w, bias=b, pad=(1, 0), res32=r32, want32=True)
    else:
        f = lambda: ops.conv_ex(x, w, bias=b, pad=(1, 0))
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    print(f"tile {tile:5s} streams {os.environ.get('STREAMS', '0')} temporal conv C={C:4d} rows={32 * HW:6d} K={3 * C:5d}: {sorted(ts)[2]:8.1f} us", flush=True)
