"""Round 5: the 3 x 3 spatial convs of the tall maps (zeroscope: 32 frame images of 40 x 72; SD-v1.5 at CFG batch 16: 64^2) per forced tile
(SPIDER_GEMM_TILE is read once per process). STREAMS=1: fp32 residual read + fp32 master written (a resnet's conv2 without producer
statistics). Graph of 10 launches, median of 5; checked against torch on the first case."""
import os, torch
import torch.nn.functional as F
from spider_amd import ops
dev = torch.device("cuda:0")
tile = os.environ.get("SPIDER_GEMM_TILE", "auto")
streams = os.environ.get("STREAMS", "0") != "0"
g = torch.Generator(device=dev).manual_seed(0)
SHAPES = {"v3d": ((32, 40, 72, 320, 320), (32, 40, 72, 640, 320), (32, 20, 36, 640, 640), (32, 20, 36, 1280, 640), (16, 64, 64, 320, 320), (16, 32, 32, 640, 640)),
          "sdxl": ((8, 96, 96, 320, 320), (8, 96, 96, 640, 320), (8, 48, 48, 640, 640), (8, 48, 48, 1280, 640), (8, 24, 24, 1280, 1280), (4, 64, 64, 320, 320))}
for B, H, W, Cin, Cout in SHAPES[os.environ.get("CONV_SHAPES", "v3d")] + ((32, 40, 72, 320, 320), )[:0]:
    x = torch.randn(B, H, W, Cin, device=dev, generator=g).half()
    w = (torch.randn(Cout, 3, 3, Cin, device=dev, generator=g) * (9 * Cin) ** -0.5).half()
    b = torch.randn(Cout, device=dev, generator=g).half()
    r32 = torch.randn(B, H, W, Cout, device=dev, generator=g) if streams else None
    f = (lambda: ops.conv_ex(x, w, bias=b, pad=(1, 1), res32=r32, want32=True)) if streams else (lambda: ops.conv_ex(x, w, bias=b, pad=(1, 1)))
    out = f(); torch.cuda.synchronize()
    if (B, H, Cin) == (32, 40, 320):
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=1).permute(0, 2, 3, 1) + (r32 if streams else 0)
        got = out[1] if streams else out.float()
        rel = float((got - ref).norm() / ref.norm())
        assert rel < (2e-5 if streams else 1e-3), rel
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            f()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 10)
    us = sorted(ts)[2]
    print(f"tile {tile:5s} streams {int(streams)} conv3x3 [{B},{H},{W}] {Cin:4d}->{Cout:4d} rows {B * H * W:6d} K {9 * Cin:5d}: {us:8.1f} us  {2.0 * B * H * W * Cout * 9 * Cin / us / 1e6:7.1f} TFLOP/s", flush=True)
