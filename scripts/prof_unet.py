"""Profiling aid: N graph-replayed SD-v1.5 UNet steps at CFG batch 2 (run under rocprofv3 --kernel-trace --stats)."""
import sys, torch
from spider_amd import ops
from spider_amd.schedulers import PNDMScheduler
from spider_amd.unet import UNetConfig, UNetEngine
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
import os
DT = torch.float16 if os.environ.get("UNET_DTYPE", "f16") == "f16" else torch.bfloat16
PRECISE = int(os.environ.get("UNET_PRECISE", "0"))
unet = UNetEngine.random_init(UNetConfig.sd15(), dev, seed=1, dtype=DT, stream32=os.environ.get("UNET_STREAM32", "0") == "1", precise=PRECISE)
g = torch.Generator(device=dev).manual_seed(0)
lat = torch.randn(1, 4, 64, 64, generator=g, device=dev)
CB = int(os.environ.get("UNET_BATCH", "2"))          # CFG batch (2 = one prompt)
enc = torch.randn(CB, 77, 768, generator=g, device=dev).to(DT)
ts = PNDMScheduler().set_timesteps(40)
unet.prepare(ts, enc)
x2 = ops.latent_to_nhwc_f32(lat, reps=CB) if PRECISE else ops.latent_to_nhwc(lat, reps=CB, dtype=DT)
unet.step(x2, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    unet.step(x2, i)
e1.record(); e1.synchronize()
print("unet step ms", e0.elapsed_time(e1) / n)
out = unet.step(x2, 0)
print("out checksum", float(out.double().abs().sum()))
