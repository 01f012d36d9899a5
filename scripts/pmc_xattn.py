"""PMC aid: repeated launches of the fused cross-attention sub-block at the 64^2 SD-v1.5 site (run under rocprofv3 --pmc ...)."""
import torch
import os as _os
DT = torch.bfloat16 if _os.environ.get("UNET_DTYPE", "f16") == "bf16" else torch.float16   # the diffusion engines\' format (f16 by default)
from spider_amd import ops
dev = torch.device("cuda:0")
B2, n_tok, C, H, LP = 2, 4096, 320, 8, 80
x = torch.randn(B2, n_tok, C, device=dev).to(DT)
mq = (torch.randn(B2 * H * LP, C, device=dev) * 0.05).to(DT); mo = (torch.randn(B2 * C, H * LP, device=dev) * 0.05).to(DT)
mqf, mof = ops.repack_fm16(mq), ops.repack_fm16(mo)
z = torch.zeros(B2 * H * LP, device=dev); bo = torch.zeros(C, device=dev).to(DT)
for _ in range(6):
    ops.xattn_fused(x, mqf, mof, z, z, bo, B2, H, 77)
torch.cuda.synchronize()
