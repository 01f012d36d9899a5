"""Phase timing of one response (CLIP, UNet loop, VAE) with events; LLM phases come from bench.py."""
import torch, time
from spider_amd import ops
from spider_amd.clip import CLIPTextConfig, CLIPTextEngine
from spider_amd.schedulers import PNDMScheduler
from spider_amd.unet import UNetConfig, UNetEngine, denoise
from spider_amd.vae import VAEConfig, VAEDecoderEngine
dev = torch.device("cuda:0")
unet = UNetEngine.random_init(UNetConfig.sd15(), dev, seed=1)
te = CLIPTextEngine.random_init(CLIPTextConfig.sd15(), dev, seed=2)
vae = VAEDecoderEngine.random_init(VAEConfig.sd15(), dev, seed=3)
g = torch.Generator(device=dev).manual_seed(0)
ids = torch.randint(1000, 40000, (2, 77), generator=g, device=dev, dtype=torch.int32)
lat0 = torch.randn(1, 4, 64, 64, generator=g, device=dev)
def T(f, n=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r
t_clip, enc = T(lambda: te.encode(ids))
sched = PNDMScheduler()
t_loop, lat = T(lambda: denoise(unet, sched, lat0, enc, 7.5, 40))
t_vae, img = T(lambda: vae.decode(lat))
ts = sched.set_timesteps(40)
t_prep, _ = T(lambda: unet.prepare(ts, enc))
print(f"clip {t_clip:.2f} ms | denoise loop (41 UNet calls, incl. prepare) {t_loop:.1f} ms | unet.prepare {t_prep:.2f} ms | vae decode {t_vae:.2f} ms")
