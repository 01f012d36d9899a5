"""Config 3 (SpiderStory-free image half): 4-panel story on random-init SDXL shapes at 768^2 (and 512^2), 50 DDIM steps,
guidance 5, FreeU, consistent self-attention (Comic_Generation.py:320-467). Prints per-UNet-step time."""
import sys, time, torch
sys.path.insert(0, "tests")
from helpers import FakeTokenizer
from spider_amd.clip import CLIPTextConfig, CLIPTextEngine
from spider_amd.schedulers import DDIMScheduler
from spider_amd.story import StableDiffusionXLPipeline, story_generation
from spider_amd.unet import UNetConfig, UNetEngine, unet_flops
from spider_amd.vae import VAEConfig, VAEDecoderEngine
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
sizes = [int(a) for a in sys.argv[2:]] or [768]
unet = UNetEngine.random_init(UNetConfig.sdxl(), dev, seed=1)
c2 = CLIPTextConfig.sdxl_2()
te1 = CLIPTextEngine.random_init(CLIPTextConfig.sd15(), dev, seed=2)
te2 = CLIPTextEngine.random_init(c2, dev, seed=3)
te2.text_projection = (torch.randn(1280, 1280, device=dev) * 0.02).bfloat16()
vcfg = VAEConfig.sd15(); vcfg.scaling = 0.13025
vae = VAEDecoderEngine.random_init(vcfg, dev, seed=4)
pipe = StableDiffusionXLPipeline(unet, vae, te1, te2, FakeTokenizer(40000), FakeTokenizer(40000), DDIMScheduler())
pipe.enable_freeu(0.6, 0.4, 1.1, 1.2)
for size in sizes:
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        imgs = story_generation(pipe, "a man with a black suit", ["wake up in the bed", "have breakfast", "work in the company", "reading book in the home"],
                                "Comic book", height=size, width=size, num_steps=steps, output_type="np")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fl = unet_flops(UNetConfig.sdxl(), size // 8, size // 8)
    print(f"story {size}x{size}: {len(imgs)} panels, {steps} steps: {dt:.2f} s total, {dt / steps * 1e3:.1f} ms per UNet step (CFG batch 8); "
          f"plain UNet flops/sample {fl['total'] / 1e12:.2f} TF")
