"""SD-v1.5 VAE decode of one 64x64 latent (512^2 image) and of 4 video-frame latents (40x72)."""
import time, torch
from spider_amd.vae import VAEConfig, VAEDecoderEngine
dev = torch.device("cuda:0")
vae = VAEDecoderEngine.random_init(VAEConfig.sd15(), dev, seed=2)
for shape in ((1, 4, 64, 64), (4, 4, 40, 72)):
    lat = torch.randn(*shape, device=dev)
    vae.decode(lat); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        vae.decode(lat)
    torch.cuda.synchronize()
    print(shape, f"{(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
