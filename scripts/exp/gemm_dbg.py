"""Tuning aid: time one GEMM shape under the current SPIDER_GEMM_TILE / SPIDER_GEMM_DBG settings."""
import sys, torch
from spider_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
for _ in range(3): ops.gemm(A, W)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.gemm(A, W)
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print(f"{M}x{N}x{K}: {us:.1f} us  {2*M*N*K/us/1e6:.0f} TF/s")
