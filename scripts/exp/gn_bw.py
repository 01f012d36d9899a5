"""Round 5: bandwidth of the GroupNorm passes at the UNet3D / batched-UNet shapes: stand-alone groupnorm (statistics pass + apply pass),
the apply pass alone with the producer's partials, 16-bit and fp32 input. Graph of 10 launches, median of 5."""
import torch
from spider_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for B, HW, C in ((32, 2880, 320), (2, 16 * 2880, 320), (32, 720, 640), (2, 16 * 720, 640), (32, 180, 1280), (16, 4096, 320), (16, 1024, 640), (2, 4096, 320)):
    x = torch.randn(B, HW, C, device=dev, generator=g).half()
    x32 = x.float()
    ga, be = torch.ones(C, device=dev).half(), torch.zeros(C, device=dev).half()
    part = ops.groupnorm_stats(x, 32, HW // 16) if HW % 16 == 0 and HW // 16 <= 1024 else None
    cases = [("stats + apply, 16-bit in", lambda: ops.groupnorm(x, ga, be, 32, 1e-5, True), B * HW * C * 6),
             ("stats + apply, fp32 in", lambda: ops.groupnorm_f32in(x32, ga, be, 32, 1e-5, True), B * HW * C * 10)]
    if part is not None:
        cases += [("apply with partials, 16-bit in", lambda: ops.groupnorm(x, ga, be, 32, 1e-5, True, partial=part), B * HW * C * 4),
                  ("apply with partials, fp32 in", lambda: ops.groupnorm_f32in(x32, ga, be, 32, 1e-5, True, partial=part), B * HW * C * 6)]
    for name, f, byt in cases:
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
        print(f"[{B:3d}, {HW:6d}, {C:4d}] {name:32s}: {us:7.1f} us   {byt / 1e6:7.1f} MB moved at least = {byt / us / 1e6:5.2f} TB/s", flush=True)
