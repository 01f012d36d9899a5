"""Round 6: where does a 3 x 3 conv of the CFG-batch-2 SD-v1.5 UNet spend its time? Shapes of the 64^2 / 32^2 / 16^2 levels, plain and with
the fp32 residual stream (conv2 role), graph of 10 launches, median of 5. Run once per SPIDER_GEMM_DBG value (0 = real, 1 = DMA only,
2 = MFMA + fragment reads only; results are garbage for 1 / 2)."""
import os, torch
from spider_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
dbg = os.environ.get("SPIDER_GEMM_DBG", "0")
for B, H, W, Cin, Cout in ((2, 64, 64, 320, 320), (2, 64, 64, 640, 320), (2, 64, 64, 960, 320), (2, 32, 32, 640, 640), (2, 32, 32, 1280, 640), (2, 16, 16, 1280, 1280), (2, 16, 16, 2560, 1280)):
    for streams in (0, 1):
        x = torch.randn(B, H, W, Cin, device=dev, generator=g).half()
        w = (torch.randn(Cout, 3, 3, Cin, device=dev, generator=g) * (9 * Cin) ** -0.5).half()
        b = torch.randn(Cout, device=dev, generator=g).half()
        r32 = torch.randn(B, H, W, Cout, device=dev, generator=g) if streams else None
        f = (lambda: ops.conv_ex(x, w, bias=b, pad=(1, 1), res32=r32, want32=True)) if streams else (lambda: ops.conv_ex(x, w, bias=b, pad=(1, 1), gn_groups=32))
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
        us = sorted(ts)[2]
        role = "conv2 (+res32, fp32 master)" if streams else "conv1 (+GroupNorm statistics)"
        print(f"dbg {dbg} conv3x3 [{B},{H},{W}] {Cin:4d}->{Cout:4d} {role:30s}: {us:7.1f} us  {2.0 * B * H * W * Cout * 9 * Cin / us / 1e6:7.1f} TFLOP/s", flush=True)
