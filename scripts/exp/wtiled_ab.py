"""A/B aid: the UNet's weight-streaming GEMMs / convs (M <= 512) on row-major vs tile-major weights, with COLD weights: every
call of the timed graph reads a different weight tensor and the pool per shape exceeds the 256 MiB Infinity Cache, as in a real
UNet step (1.7 GB of weights per step). Same process, interleaved rounds (guide rule 24)."""
import json, os, sys
import torch
from spider_amd import ops

dev = torch.device("cuda:0")
DT = torch.float16
SHAPES = [  # (tag, M, N, K, conv_cin or 0, hw)
    ("u16 out", 512, 1280, 1280, 0, 0), ("u16 qkv", 512, 3840, 1280, 0, 0), ("u16 ff1", 512, 10240, 1280, 0, 0), ("u16 ff2", 512, 1280, 5120, 0, 0),
    ("u8 out", 128, 1280, 1280, 0, 0), ("u8 ff1", 128, 10240, 1280, 0, 0), ("u8 ff2", 128, 1280, 5120, 0, 0),
    ("c16 1280>1280", 512, 1280, 11520, 1280, 16), ("c16 2560>1280", 512, 1280, 23040, 2560, 16), ("c16 640>1280", 512, 1280, 5760, 640, 16),
    ("c8 1280>1280", 128, 1280, 11520, 1280, 8), ("c8 2560>1280", 128, 1280, 23040, 2560, 8),
    ("u32 out", 2048, 640, 640, 0, 0), ("u32 ff1", 2048, 5120, 640, 0, 0), ("u32 ff2", 2048, 640, 2560, 0, 0),
    ("c32 640>640", 2048, 640, 5760, 640, 32), ("c32 1280>640", 2048, 640, 11520, 1280, 32),
]
if len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if any(a in s[0] for a in sys.argv[1:])]
ops.WTILED_MAX_M = 1 << 30      # the A/B decides by marking, not by the threshold


def bench(tag, M, N, K, cin, hw):
    wbytes = N * K * 2
    nW = max(4, min(24, (600 << 20) // wbytes))
    if cin:
        x = torch.randn(M // (hw * hw), hw, hw, cin, device=dev).to(DT)
        ws = [(torch.randn(N, 3, 3, cin, device=dev) * 0.02).to(DT) for _ in range(nW)]
        call = lambda w: ops.conv2d(x, w)
    else:
        A = torch.randn(M, K, device=dev).to(DT)
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(DT) for _ in range(nW)]
        call = lambda w: ops.gemm(A, w)
    wt = [ops.mark_weight(w.clone()) for w in ws]
    ref, got = call(ws[0]), call(wt[0])
    torch.cuda.synchronize()
    assert getattr(wt[0], "_spider_tiled", None) is not None
    assert torch.equal(ref, got), f"{tag}: tile-major result differs"
    graphs = []
    for pool in (ws, wt):
        for w in pool:
            call(w)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for w in pool:
                call(w)
        graphs.append(g)
    t = [[], []]
    for r in range(5):
        for i, g in enumerate(graphs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); e1.synchronize()
            t[i].append(e0.elapsed_time(e1) * 1e3 / nW)
    med = [sorted(v)[len(v) // 2] for v in t]
    floor = wbytes / 6.3e12 * 1e6
    print(f"{tag:16s} M={M:5d} W={wbytes / 1e6:6.1f} MB  row-major {med[0]:7.1f} us  tile-major {med[1]:7.1f} us  ({med[0] / med[1]:.2f}x)  "
          f"weight-stream floor {floor:5.1f} us", flush=True)
    return tag, med


res = dict(bench(*s) for s in SHAPES)
json.dump(res, open("gpurun_out/wtiled_ab.json", "w"))
