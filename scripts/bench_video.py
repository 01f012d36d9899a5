"""Text-to-video at the true shapes (zeroscope_v2_576w configs, random weights): UNet3D step at [2*16, 40, 72, 4]
(16 frames, 320x576, CFG) and the VAE decode of the 16 frames. Prints milliseconds per stage."""
import sys, time, torch
from spider_amd import ops
from spider_amd.schedulers import DDIMScheduler
from spider_amd.unet3d import UNet3DConfig, UNet3DEngine
from spider_amd.vae import VAEConfig, VAEDecoderEngine

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16
caps = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # captions per call (SpiderDecoder.generate_batch: video_batch)
h, w = 40, 72
unet = UNet3DEngine.random_init(UNet3DConfig.zeroscope(), dev, seed=1)
g = torch.Generator(device=dev).manual_seed(0)
enc = torch.randn(2 * caps, 77, 1024, generator=g, device=dev).bfloat16()
ts = DDIMScheduler().set_timesteps(40)
unet.prepare(ts, enc, frames=frames)
lat = torch.randn(caps * frames, 4, h, w, generator=g, device=dev)
x2 = ops.latent_to_nhwc(lat, reps=2)
unet.step(x2, 0); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    unet.step(x2, i)
torch.cuda.synchronize()
t_unet = (time.perf_counter() - t0) / n * 1e3
print(f"unet3d step ms {t_unet:.2f}  (batch 2 x {caps} x {frames} frames at {h}x{w}); HBM in use {torch.cuda.memory_allocated() / 2**30:.1f} GiB")
vae = VAEDecoderEngine.random_init(VAEConfig.sd15(), dev, seed=2)
vae.decode(lat[:4]); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(0, frames, 4):
    vae.decode(lat[i:i + 4], to_image=False)
torch.cuda.synchronize()
print(f"vae decode of {frames} frames {((time.perf_counter() - t0) * 1e3):.1f} ms; 40-step video ~ {(40 * t_unet) / 1e3:.2f} s + decode")
