"""PMC aid: repeated launches of one GEMM / conv shape (run under rocprofv3 --pmc ...)."""
import sys, torch
import os as _os
DT = torch.bfloat16 if _os.environ.get("UNET_DTYPE", "f16") == "bf16" else torch.float16   # the diffusion engines\' format (f16 by default)
from spider_amd import ops
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "gate_up"
if which == "gate_up":
    A = torch.randn(1536, 3584, device=dev).to(DT); W = (torch.randn(37888, 3584, device=dev) * 0.02).to(DT)
    f = lambda: ops.gemm(A, W)
elif which == "conv64":
    x = torch.randn(2, 64, 64, 640, device=dev).to(DT); w = (torch.randn(320, 3, 3, 640, device=dev) * 0.02).to(DT)
    f = lambda: ops.conv2d(x, w)
elif which == "conv64_320":   # the roofline_unet_conv shape of bench.py
    x = torch.randn(2, 64, 64, 320, device=dev).to(DT); w = (torch.randn(320, 3, 3, 320, device=dev) * 0.02).to(DT)
    f = lambda: ops.conv2d(x, w)
elif which == "conv64_320_res32":   # the same conv as a ResnetBlock2D.conv2 of the fp32 residual stream: fp32 master in (shortcut) and out
    x = torch.randn(2, 64, 64, 320, device=dev).to(DT); w = (torch.randn(320, 3, 3, 320, device=dev) * 0.02).to(DT)
    r32 = torch.randn(2, 64, 64, 320, device=dev)
    f = lambda: ops.conv2d(x, w, res32=r32, want32=True, gn_groups=32)
elif which == "conv64_320_gn":      # ResnetBlock2D.conv1: 16-bit in / out, GroupNorm partials of the output from the epilogue
    x = torch.randn(2, 64, 64, 320, device=dev).to(DT); w = (torch.randn(320, 3, 3, 320, device=dev) * 0.02).to(DT)
    f = lambda: ops.conv2d(x, w, gn_groups=32)
elif which == "conv48_640":   # SDXL 48^2 level (CFG batch 8): 18432 rows, input 23.6 MB
    x = torch.randn(8, 48, 48, 640, device=dev).to(DT); w = (torch.randn(640, 3, 3, 640, device=dev) * 0.02).to(DT)
    f = lambda: ops.conv2d(x, w)
elif which == "conv3d_640":   # zeroscope 20 x 36 level, 2 x 16 frames: 23040 rows, input 29.5 MB
    x = torch.randn(32, 20, 36, 640, device=dev).to(DT); w = (torch.randn(640, 3, 3, 640, device=dev) * 0.02).to(DT)
    f = lambda: ops.conv2d(x, w)
elif which == "ff1":
    A = torch.randn(8192, 320, device=dev).to(DT); W = (torch.randn(2560, 320, device=dev) * 0.02).to(DT)
    f = lambda: ops.gemm(A, W)
else:
    A = torch.randn(8192, 320, device=dev).to(DT); W = (torch.randn(960, 320, device=dev) * 0.02).to(DT)
    f = lambda: ops.gemm(A, W)
for _ in range(6):
    f()
torch.cuda.synchronize()
