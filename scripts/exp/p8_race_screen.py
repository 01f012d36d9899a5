"""Race screen for the 256x256 LDS-DMA GEMM (counted vmcnt + raw barriers + staggered wave groups): the same launch repeated
many times must give bit-identical outputs, for plain / split-K / conv / GEGLU / LayerNorm-folded forms at several sizes, with a
bandwidth-hungry kernel interleaved on a second stream to perturb DMA arrival order."""
import torch
from spider_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
side = torch.cuda.Stream()
junk = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
bad = 0
def screen(tag, f, n=150):
    global bad
    ref = f().clone()
    torch.cuda.synchronize()
    diff = 0
    for i in range(n):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                junk.add_(1)                      # HBM traffic beside the GEMM
        out = f()
        if not torch.equal(out, ref):
            diff += 1
    torch.cuda.synchronize()
    print(f"{tag}: {diff} of {n} repeats differ", flush=True)
    bad += diff
for M, N, K in [(4608, 3840, 1280), (1536, 3584, 3584), (4000, 2504, 1096), (18432, 640, 2560), (1536, 37888, 3584)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    screen(f"gemm {M}x{N}x{K}", lambda: ops.gemm(A, W))
A = torch.randn(4608, 1280, device=dev).bfloat16(); W = (torch.randn(10240, 1280, device=dev) * 0.05).bfloat16()
b = torch.randn(10240, device=dev).bfloat16()
screen("geglu 4608x5120x1280", lambda: ops.gemm(A, W, bias=b, act="geglu"))
ga = torch.ones(1280, device=dev).bfloat16(); be = torch.zeros(1280, device=dev).bfloat16()
Wf, cs, cb = ops.fold_layernorm(W, ga, be, b)
screen("gemm_ln geglu 4608x5120x1280", lambda: ops.gemm_ln(A, Wf, cs, cb, act="geglu", eps=1e-5))
screen("gemm_ln 4608x10240x1280", lambda: ops.gemm_ln(A, Wf, cs, cb, eps=1e-5))
x = torch.randn(8, 48, 48, 640, device=dev).bfloat16(); w = (torch.randn(640, 3, 3, 640, device=dev) * 0.02).bfloat16()
screen("conv 8x48x48 640->640", lambda: ops.conv2d(x, w), n=80)
x = torch.randn(8, 24, 24, 1280, device=dev).bfloat16(); w = (torch.randn(1280, 3, 3, 1280, device=dev) * 0.02).bfloat16()
screen("conv 8x24x24 1280->1280 (2 K splits)", lambda: ops.conv2d(x, w), n=80)
print("TOTAL differing repeats:", bad)
