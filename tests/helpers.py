"""Test helpers: tiny seeded pipelines; the tokenizer stand-ins live in benchkit/synthetic.py."""
from benchkit.synthetic import FakeRobertaTokenizer, FakeTokenizer  # noqa: F401


def tiny_audio_pipe(dev, dtype=None):
    """AudioLDMPipeline over tiny seeded engines (same configs as tests/test_audio_engine.py), in the mode from_pretrained
    loads by default: f16 operands, fp32 residual stream in the UNet."""
    import torch
    d = dtype or torch.float16
    from oracle.audio import ClapTextCfg, HifiGanCfg, clap_param_shapes, hifigan_param_shapes, random_weights
    from oracle.clip_vae import VAECfg, vae_param_shapes
    from oracle.unet import UNetCfg, random_unet_weights
    from spider_amd.clap import ClapTextConfig, ClapTextEngine
    from spider_amd.pipelines import AudioLDMPipeline
    from spider_amd.schedulers import DDIMScheduler
    from spider_amd.unet import UNetConfig, UNetEngine
    from spider_amd.vae import VAEConfig, VAEDecoderEngine
    from spider_amd.vocoder import HifiGanConfig, HifiGanEngine
    ccfg = ClapTextCfg(400, 64, 2, 4, 128, 40, 48, 1e-12, 1)
    ucfg = UNetCfg.tiny_audio()
    vcfg = VAECfg(latent=8, out_ch=1, block_out=(64, 128, 128), layers_per_block=1, scaling=0.9227)
    hcfg = HifiGanCfg(16, 16000, 64, (5, 4, 2), (16, 16, 8), (3, 7), ((1, 3, 5), (1, 3, 5)), 0.1, False)
    return AudioLDMPipeline(VAEDecoderEngine(VAEConfig(**vcfg.__dict__), random_weights(vae_param_shapes(vcfg), 23), dev, dtype=d),
                            ClapTextEngine(ClapTextConfig(**ccfg.__dict__), random_weights(clap_param_shapes(ccfg), 21), dev, dtype=d),
                            FakeRobertaTokenizer(400),
                            UNetEngine(UNetConfig(**ucfg.__dict__), random_unet_weights(ucfg, 22), dev, dtype=d, stream32=d == torch.float16),
                            DDIMScheduler(), HifiGanEngine(HifiGanConfig(**hcfg.__dict__), random_weights(hifigan_param_shapes(hcfg), 24), dev, dtype=d),
                            sample_size=16)


def tiny_video_pipe(dev, dtype=None):
    import torch
    d = dtype or torch.float16
    from oracle.clip_vae import CLIPCfg, VAECfg, clip_param_shapes, random_weights, vae_param_shapes
    from oracle.unet3d import UNet3DCfg, random_unet3d_weights
    from spider_amd.clip import CLIPTextConfig, CLIPTextEngine
    from spider_amd.pipelines import TextToVideoSDPipeline
    from spider_amd.schedulers import DDIMScheduler
    from spider_amd.unet3d import UNet3DConfig, UNet3DEngine
    from spider_amd.vae import VAEConfig, VAEDecoderEngine
    ocfg, ccfg, vcfg = UNet3DCfg.tiny(), CLIPCfg.tiny(), VAECfg.tiny()
    return TextToVideoSDPipeline(UNet3DEngine(UNet3DConfig(**ocfg.__dict__), random_unet3d_weights(ocfg, 6), dev, dtype=d),
                                 VAEDecoderEngine(VAEConfig(**vcfg.__dict__), random_weights(vae_param_shapes(vcfg), 8), dev, dtype=d),
                                 CLIPTextEngine(CLIPTextConfig(**ccfg.__dict__), random_weights(clip_param_shapes(ccfg), 7), dev, dtype=d),
                                 FakeTokenizer(ccfg.vocab), DDIMScheduler(), sample_size=8)
