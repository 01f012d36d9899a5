"""SD-v1.5 UNet evaluation time against the CFG batch (f16 + fp32 residual stream, 64x64 latent, hipGraph replay): 2 = one prompt,
4 / 8 / 16 = 2 / 4 / 8 prompts answered together."""
import torch
from spider_amd import ops
from spider_amd.schedulers import PNDMScheduler
from spider_amd.unet import UNetConfig, UNetEngine

dev = torch.device("cuda:0")
DT = torch.float16
unet = UNetEngine.random_init(UNetConfig.sd15(), dev, seed=1, dtype=DT, stream32=True)
g = torch.Generator(device=dev).manual_seed(0)
ts = PNDMScheduler().set_timesteps(40)
print("CFG batch  ms/evaluation  ms per sample")
for B2 in (2, 4, 6, 8, 12, 16):
    enc = torch.randn(B2, 77, 768, generator=g, device=dev).to(DT)
    unet.prepare(ts, enc)
    x = ops.latent_to_nhwc(torch.randn(B2 // 2, 4, 64, 64, generator=g, device=dev), reps=2, dtype=DT)
    unet.step(x, 0); unet.step(x, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        unet.step(x, i)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{B2:9d} {ms:14.3f} {ms / B2:14.3f}", flush=True)
