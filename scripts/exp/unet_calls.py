"""Profiling aid: log every C-ABI call (name + integer arguments) of ONE eager SD-v1.5 UNet step, to be joined with a
rocprofv3 kernel trace of the same process by scripts/exp/join_calls.py (per-call device time, in launch order)."""
import json, sys, torch
from spider_amd import lib, ops
from spider_amd.schedulers import PNDMScheduler
from spider_amd.unet import UNetConfig, UNetEngine
dev = torch.device("cuda:0")
import os
DT = torch.float16 if os.environ.get("UNET_DTYPE", "f16") == "f16" else torch.bfloat16
unet = UNetEngine.random_init(UNetConfig.sd15(), dev, seed=1, dtype=DT)
g = torch.Generator(device=dev).manual_seed(0)
lat = torch.randn(1, 4, 64, 64, generator=g, device=dev)
enc = torch.randn(2, 77, 768, generator=g, device=dev).to(DT)
ts = PNDMScheduler().set_timesteps(40)
unet.prepare(ts, enc)
x2 = ops.latent_to_nhwc(lat, reps=2, dtype=DT)
log = []
orig = lib.call
def logged(name, *args):
    log.append([name] + [int(a) if isinstance(a, (int, float)) and not isinstance(a, bool) and abs(a) < 1 << 31 else None for a in args])
    return orig(name, *args)
for i in range(3):
    if i == 2:
        lib.call = logged
    unet.step(x2, i, use_graph=False)
    torch.cuda.synchronize()
lib.call = orig
json.dump(log, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/unet_calls.json", "w"))
print("calls logged", len(log))
