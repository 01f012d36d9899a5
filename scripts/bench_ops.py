"""Tuning aid: graph-timed single-op latencies at the SD-v1.5 UNet shapes (CFG batch 2)."""
import torch
from spider_amd import ops
dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * n)


r = lambda *s: torch.randn(*s, device=dev).to(BF)
print("--- groupnorm+silu [2,HW,C]")
for HW, C in [(4096, 320), (4096, 640), (4096, 960), (1024, 640), (1024, 1280), (1024, 1920), (256, 1280), (256, 2560), (64, 1280), (64, 2560)]:
    x, g, b = r(2, HW, C), r(C), r(C)
    t = timeit(lambda: ops.groupnorm(x, g, b, 32, 1e-5, True))
    print(f"  HW {HW:5d} C {C:5d}: {t:6.1f} us   {3 * x.numel() * 2 / t / 1e6:6.2f} TB/s")
print("--- layernorm [rows,C]")
for rows, C in [(8192, 320), (2048, 640), (512, 1280), (128, 1280)]:
    x, g, b = r(rows, C), r(C), r(C)
    t = timeit(lambda: ops.layernorm(x, g, b))
    print(f"  rows {rows:5d} C {C:5d}: {t:6.1f} us")
print("--- geglu [M, 2*inner]")
for M, inner in [(8192, 1280), (2048, 2560), (512, 5120)]:
    x = r(M, 2 * inner)
    t = timeit(lambda: ops.geglu(x))
    print(f"  M {M:5d} inner {inner:5d}: {t:6.1f} us")
print("--- self attention [2,N,8*d] and cross attention (Lk=77)")
for N, d in [(4096, 40), (1024, 80), (256, 160), (64, 160)]:
    C = 8 * d
    qkv, kv = r(2, N, 3 * C), r(2, 77, 2 * C)
    q = r(2, N, C)
    t = timeit(lambda: ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], 8))
    t2 = timeit(lambda: ops.attention(q, kv[..., :C], kv[..., C:], 8))
    print(f"  N {N:5d} d {d:4d}: self {t:6.1f} us ({4 * N * N * C * 2 / t / 1e6:6.1f} TF/s)   cross {t2:6.1f} us")
print("--- concat / small convs")
a, b = r(2, 4096, 320), r(2, 4096, 320)
print(f"  concat 320+320 @64^2: {timeit(lambda: ops.concat_channels(a, b)):6.1f} us")
x4, w4, b4 = r(2, 64, 64, 4), r(320, 3, 3, 4), r(320)
print(f"  conv_in: {timeit(lambda: ops.conv2d_small_cin(x4, w4, b4)):6.1f} us")
xo, wo, bo = r(2, 64, 64, 320), r(4, 3, 3, 320), r(4)
print(f"  conv_out: {timeit(lambda: ops.conv2d_small_cout(xo, wo, bo)):6.1f} us")
