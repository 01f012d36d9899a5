"""AudioLDM path at the true shapes (cvssp/audioldm-s-full-v2 configs, random weights): CLAP text, 40-step DDIM loop on the
[2,8,125,16] latent (5.0 s of audio, CFG), mel VAE decode, HiFi-GAN vocoder. Prints per-stage milliseconds."""
import sys, time, torch
from spider_amd.clap import ClapTextConfig, ClapTextEngine
from spider_amd.pipelines import AudioLDMPipeline
from spider_amd.schedulers import DDIMScheduler
from spider_amd.unet import UNetConfig, UNetEngine, unet_flops
from spider_amd.vae import VAEConfig, VAEDecoderEngine
from spider_amd.vocoder import HifiGanConfig, HifiGanEngine

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
pipe = AudioLDMPipeline(VAEDecoderEngine.random_init(VAEConfig.audioldm(), dev, 1), ClapTextEngine.random_init(ClapTextConfig(), dev, 2),
                        None, UNetEngine.random_init(UNetConfig.audioldm(), dev, 3), DDIMScheduler(beta_start=0.0015, beta_end=0.0195),
                        HifiGanEngine.random_init(HifiGanConfig.audioldm(), dev, 4))
ids = torch.randint(3, 50000, (1, 12)); ids[0, 0] = 0; ids[0, -1] = 2


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


t_clap, emb = timed(lambda: pipe.text_encoder.text_embeds(torch.cat([ids, ids]), normalize=True))
g = torch.Generator(device=dev).manual_seed(0)
t_all, out = timed(lambda: pipe(prompt_embeds=emb[1:], negative_prompt_embeds=emb[:1], audio_length_in_s=secs,
                                num_inference_steps=steps, guidance_scale=2.5, generator=g), n=2)
lat = torch.randn(1, 8, int(secs * 100) // 4, 16, device=dev)
t_vae, mel = timed(lambda: pipe.decode_latents(lat))
t_voc, wav = timed(lambda: pipe.vocoder(mel.squeeze(1)))
fl = unet_flops(pipe.unet.cfg, lat.shape[2], 16, self_cross=True)
t_unet = (t_all - t_vae - t_voc) / steps
print(f"audio {secs}s: total {t_all:.1f} ms = CLAP(2 prompts) {t_clap:.2f} (outside) + {steps} x UNet {t_unet:.3f} ms "
      f"({2 * fl['total'] / t_unet / 1e9:.1f} TF/s of {2 * fl['total'] / 1e9:.1f} GF) + VAE {t_vae:.2f} + vocoder {t_voc:.2f}; "
      f"samples {out.audios.shape}")
