"""One shape on the 256x256 kernel, for counter passes: python p8_one.py M N K [reps]."""
import sys, torch
from spider_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(reps): ops.gemm(A, W, out=out)
torch.cuda.synchronize()
