"""Round 6: the fixed cost of a 64^2 conv launch -- 3 x 3 convs [2, 64, 64, Cin] -> 320 for Cin = 32 ... 960 (K = 288 ... 8640), both epilogue
roles, graph of 10 launches, median of 5; SPIDER_GEMM_TILE forces one kernel for every K (161 = 64-row LDS-DMA tiles, 160 = 128-row)."""
import os, torch
from spider_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
tile = os.environ.get("SPIDER_GEMM_TILE", "auto")
for Cin in (32, 64, 128, 192, 320, 640, 960):
    row = []
    for role in ("gn", "res32", "plain"):
        x = torch.randn(2, 64, 64, Cin, device=dev, generator=g).half()
        w = (torch.randn(320, 3, 3, Cin, device=dev, generator=g) * (9 * Cin) ** -0.5).half()
        b = torch.randn(320, device=dev, generator=g).half()
        r32 = torch.randn(2, 64, 64, 320, device=dev, generator=g)
        f = {"gn": lambda: ops.conv_ex(x, w, bias=b, pad=(1, 1), gn_groups=32),
             "res32": lambda: ops.conv_ex(x, w, bias=b, pad=(1, 1), res32=r32, want32=True),
             "plain": lambda: ops.conv_ex(x, w, bias=b, pad=(1, 1))}[role]
        f(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10):
                f()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 10)
        row.append(sorted(ts)[2])
    print(f"tile {tile:5s} Cin {Cin:4d} (K tiles {9 * Cin // 64:3d}): stats epilogue {row[0]:6.1f} us   fp32 stream {row[1]:6.1f} us   plain {row[2]:6.1f} us", flush=True)
