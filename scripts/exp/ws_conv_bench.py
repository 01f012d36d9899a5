"""Round 5: the weight-stationary streaming conv (wstream_kernel) against the tile kernels on the UNet's weight-bound shapes, each
launched 96 times in a hipGraph walking ~480 MB of distinct weight tensors (cold HBM, as in a real step). One process, modes
alternating (cdna guide rule 24); fp32 residual stream operands as the resnet's conv2 has them."""
import sys
import torch
from spider_amd import ops

CASES = [("c8 1280>1280", 2, 8, 1280, 1280, None), ("c8 2560>1280", 2, 8, 2560, 1280, None),
         ("c16 1280>1280", 2, 16, 1280, 1280, None), ("c16 2560>1280", 2, 16, 2560, 1280, None), ("c16 1920>1280", 2, 16, 1920, 1280, None),
         ("c16 640>1280", 2, 16, 640, 1280, None), ("up 8>16 1280", 2, 8, 1280, 1280, (16, 16))]
dev = torch.device("cuda:0")
DT = torch.float16
ops.WS_MAX_M = 512
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if len(sys.argv) > 2:          # a single case by index (profiling runs)
    CASES = [CASES[int(sys.argv[2])]]
print(f"{'shape':16s} {'tile us':>9s} {'ws us':>9s} {'ratio':>6s}   weights MB   TB/s(ws)   [conv2 role: bias + temb + fp32 residual in / out]")
for tag, B, hw, cin, cout, up in CASES:
    wbytes = cout * 9 * cin * 2
    nw = max(2, min(48, int(480e6 / wbytes) + 1))
    x = torch.randn(B, hw, hw, cin, device=dev).to(DT)
    Ws = [ops.mark_weight((torch.randn(cout, 3, 3, cin, device=dev) * 0.01).to(DT)) for _ in range(nw)]
    bias = torch.randn(cout, device=dev).to(DT)
    rb = torch.randn(B, cout, device=dev).to(DT)
    ho = up[0] if up else hw
    r32 = torch.randn(B, ho, ho, cout, device=dev)
    f = lambda w: ops.conv_ex(x, w, bias=bias, rowbias=rb, pad=(1, 1), up_size=up, res32=r32, want32=True)
    graphs = {}
    for mode in (0, 1):
        ops.WS_ENABLE = bool(mode)
        for w in Ws:
            f(w)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(96):
                f(Ws[i % nw])
        graphs[mode] = g
    t = {0: [], 1: []}
    for _ in range(rounds):
        for mode in (0, 1):
            graphs[mode].replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); graphs[mode].replay(); graphs[mode].replay(); e1.record(); e1.synchronize()
            t[mode].append(e0.elapsed_time(e1) * 1e3 / 192)
    a, b = sorted(t[0])[len(t[0]) // 2], sorted(t[1])[len(t[1]) // 2]
    print(f"{tag:16s} {a:9.1f} {b:9.1f} {a / b:6.2f}   {wbytes / 1e6:8.1f}   {wbytes / b / 1e6:6.2f}", flush=True)
    del Ws, graphs
    torch.cuda.empty_cache()
ops.WS_ENABLE = True
