"""PMC aid: repeated flash-attention launches at an SD-v1.5 shape (run under rocprofv3 --pmc ...)."""
import sys, torch
from spider_amd import ops
dev = torch.device("cuda:0")
N, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 40)
C = 8 * d
qkv = torch.randn(2, N, 3 * C, device=dev).bfloat16()
for _ in range(6):
    ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], 8)
torch.cuda.synchronize()
