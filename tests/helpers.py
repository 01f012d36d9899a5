"""Test helpers: a deterministic stand-in for CLIPTokenizer (no vocabulary files travel with the repo)."""
import torch


class FakeTokenizer:
    """Hashes whitespace-separated words to ids; BOS=0, EOS/pad=2; same call surface the pipeline uses."""
    model_max_length = 77

    def __init__(self, vocab=400):
        self.vocab = vocab

    def _ids(self, text):
        return [0] + [3 + (sum(ord(c) * (i + 1) for i, c in enumerate(w)) % (self.vocab - 3)) for w in text.split()] + [2]

    def __call__(self, prompts, padding="longest", max_length=None, truncation=False, return_tensors="pt"):
        if isinstance(prompts, str):
            prompts = [prompts]
        rows = [self._ids(p) for p in prompts]
        if truncation and max_length:
            rows = [r[:max_length - 1] + [2] if len(r) > max_length else r for r in rows]
        L = max_length if padding == "max_length" else max(len(r) for r in rows)
        rows = [r + [2] * (L - len(r)) for r in rows]
        class Out:
            pass
        o = Out()
        o.input_ids = torch.tensor(rows, dtype=torch.long)
        return o

    def batch_decode(self, ids):
        return [" ".join(f"w{int(t)}" for t in row if int(t) > 2) for row in ids]


class FakeRobertaTokenizer(FakeTokenizer):
    """RoBERTa-style stand-in: BOS=0, EOS=2, PAD=1, right padding, returns input_ids + attention_mask."""
    model_max_length = 32

    def __call__(self, prompts, padding="longest", max_length=None, truncation=False, return_tensors="pt"):
        if isinstance(prompts, str):
            prompts = [prompts]
        rows = [self._ids(p) for p in prompts]
        if truncation and max_length:
            rows = [r[:max_length - 1] + [2] if len(r) > max_length else r for r in rows]
        L = max_length if padding == "max_length" else max(len(r) for r in rows)
        class Out:
            pass
        o = Out()
        o.input_ids = torch.tensor([r + [1] * (L - len(r)) for r in rows], dtype=torch.long)
        o.attention_mask = torch.tensor([[1] * len(r) + [0] * (L - len(r)) for r in rows], dtype=torch.long)
        return o


def tiny_audio_pipe(dev):
    """AudioLDMPipeline over tiny seeded engines (same configs as tests/test_audio_engine.py)."""
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
    return AudioLDMPipeline(VAEDecoderEngine(VAEConfig(**vcfg.__dict__), random_weights(vae_param_shapes(vcfg), 23), dev),
                            ClapTextEngine(ClapTextConfig(**ccfg.__dict__), random_weights(clap_param_shapes(ccfg), 21), dev),
                            FakeRobertaTokenizer(400), UNetEngine(UNetConfig(**ucfg.__dict__), random_unet_weights(ucfg, 22), dev),
                            DDIMScheduler(), HifiGanEngine(HifiGanConfig(**hcfg.__dict__), random_weights(hifigan_param_shapes(hcfg), 24), dev),
                            sample_size=16)


def tiny_video_pipe(dev):
    from oracle.clip_vae import CLIPCfg, VAECfg, clip_param_shapes, random_weights, vae_param_shapes
    from oracle.unet3d import UNet3DCfg, random_unet3d_weights
    from spider_amd.clip import CLIPTextConfig, CLIPTextEngine
    from spider_amd.pipelines import TextToVideoSDPipeline
    from spider_amd.schedulers import DDIMScheduler
    from spider_amd.unet3d import UNet3DConfig, UNet3DEngine
    from spider_amd.vae import VAEConfig, VAEDecoderEngine
    ocfg, ccfg, vcfg = UNet3DCfg.tiny(), CLIPCfg.tiny(), VAECfg.tiny()
    return TextToVideoSDPipeline(UNet3DEngine(UNet3DConfig(**ocfg.__dict__), random_unet3d_weights(ocfg, 6), dev),
                                 VAEDecoderEngine(VAEConfig(**vcfg.__dict__), random_weights(vae_param_shapes(vcfg), 8), dev),
                                 CLIPTextEngine(CLIPTextConfig(**ccfg.__dict__), random_weights(clip_param_shapes(ccfg), 7), dev),
                                 FakeTokenizer(ccfg.vocab), DDIMScheduler(), sample_size=8)
