"""GPU: a checkpoint DIRECTORY through the loaders the reference calls -- `cls.from_pretrained(ckpt, torch_dtype=torch.float16)`
(base_model.py:207-219, spider_decoder.py:109) -- on a tiny synthetic Stable-Diffusion directory written in the published diffusers
layout: unet/ as a torch pickle (.bin) next to an fp16 variant that must not be read, vae/ with encoder tensors beside the decoder's,
text_encoder/ as an indexed two-shard safetensors set, tokenizer/ as CLIP BPE files, scheduler/scheduler_config.json. The loaded
pipeline must equal, bit for bit, one assembled directly from the same tensors; SpiderDecoder must reach it from its config dict."""
import json
import os

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

pytestmark = pytest.mark.gpu


def _clip_bpe_files(d):
    """A CLIP tokenizer with an empty merge table: byte alphabet, the same with the end-of-word mark, the two specials."""
    from tokenizers.pre_tokenizers import ByteLevel
    os.makedirs(d, exist_ok=True)
    chars = sorted(ByteLevel.alphabet())
    vocab = chars + [c + "</w>" for c in chars] + ["<|startoftext|>", "<|endoftext|>"]
    json.dump({t: i for i, t in enumerate(vocab)}, open(os.path.join(d, "vocab.json"), "w"))
    open(os.path.join(d, "merges.txt"), "w").write("#version: 0.2\n")
    json.dump({"model_max_length": 77, "bos_token": "<|startoftext|>", "eos_token": "<|endoftext|>", "unk_token": "<|endoftext|>",
               "pad_token": "<|endoftext|>", "tokenizer_class": "CLIPTokenizer"}, open(os.path.join(d, "tokenizer_config.json"), "w"))
    return len(vocab)


def _write_sd_directory(root):
    from oracle.clip_vae import CLIPCfg, VAECfg, clip_param_shapes, random_weights, vae_param_shapes
    from oracle.unet import UNetCfg, random_unet_weights
    n_vocab = _clip_bpe_files(os.path.join(root, "tokenizer"))
    uc, vc = UNetCfg.tiny(), VAECfg.tiny()
    cc = CLIPCfg(n_vocab, 64, 2, 2, 128, 77)
    wu, wv, wc = random_unet_weights(uc, seed=1), random_weights(vae_param_shapes(vc), seed=2), random_weights(clip_param_shapes(cc), seed=3)
    half = lambda w: {k: v.to(torch.float16).contiguous() for k, v in w.items()}
    # unet/: pickle + a misleading fp16 variant (zeros) that a glob over *.safetensors would pick up
    os.makedirs(os.path.join(root, "unet"))
    json.dump({"_class_name": "UNet2DConditionModel", "in_channels": 4, "out_channels": 4, "block_out_channels": [64, 128, 128],
               "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
               "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"], "attention_head_dim": 2,
               "layers_per_block": 2, "cross_attention_dim": 64, "norm_num_groups": 32, "sample_size": 16,
               "use_linear_projection": False}, open(os.path.join(root, "unet", "config.json"), "w"))
    torch.save(half(wu), os.path.join(root, "unet", "diffusion_pytorch_model.bin"))
    save_file({k: torch.zeros_like(v) for k, v in half(wu).items()}, os.path.join(root, "unet", "diffusion_pytorch_model.fp16.safetensors"))
    # vae/: the published file holds encoder + quant_conv tensors too
    os.makedirs(os.path.join(root, "vae"))
    json.dump({"_class_name": "AutoencoderKL", "latent_channels": vc.latent, "out_channels": vc.out_ch, "block_out_channels": list(vc.block_out),
               "layers_per_block": vc.layers_per_block, "norm_num_groups": vc.groups, "scaling_factor": 0.18215},
              open(os.path.join(root, "vae", "config.json"), "w"))
    extra = {"encoder.conv_in.weight": torch.randn(8, 3, 3, 3).half(), "quant_conv.weight": torch.randn(8, 8, 1, 1).half()}
    save_file({**half(wv), **extra}, os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"))
    # text_encoder/: two shards behind an index, plus the position_ids buffer old exports carry
    os.makedirs(os.path.join(root, "text_encoder"))
    json.dump({"architectures": ["CLIPTextModel"], "vocab_size": cc.vocab, "hidden_size": cc.hidden, "num_hidden_layers": cc.layers,
               "num_attention_heads": cc.heads, "intermediate_size": cc.inter, "max_position_embeddings": cc.max_pos,
               "layer_norm_eps": 1e-5, "hidden_act": "quick_gelu"}, open(os.path.join(root, "text_encoder", "config.json"), "w"))
    hc = half(wc)
    names = sorted(hc)
    shard = {n: ("model-00001-of-00002.safetensors" if i % 2 == 0 else "model-00002-of-00002.safetensors") for i, n in enumerate(names)}
    for f in set(shard.values()):
        t = {n: hc[n] for n in names if shard[n] == f}
        if f.startswith("model-00001"):
            t["text_model.embeddings.position_ids"] = torch.arange(77)[None]
        save_file(t, os.path.join(root, "text_encoder", f))
    json.dump({"metadata": {}, "weight_map": shard}, open(os.path.join(root, "text_encoder", "model.safetensors.index.json"), "w"))
    os.makedirs(os.path.join(root, "scheduler"))
    json.dump({"_class_name": "PNDMScheduler", "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear",
               "num_train_timesteps": 1000, "skip_prk_steps": True, "steps_offset": 1, "set_alpha_to_one": False},
              open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    json.dump({"_class_name": "StableDiffusionPipeline"}, open(os.path.join(root, "model_index.json"), "w"))
    return (uc, vc, cc), (half(wu), half(wv), hc)


def test_sd_directory_loads_and_equals_direct_assembly(dev, tmp_path):
    from transformers import CLIPTokenizer
    from spider_amd.clip import CLIPTextConfig, CLIPTextEngine
    from spider_amd.pipelines import StableDiffusionPipeline
    from spider_amd.schedulers import PNDMScheduler
    from spider_amd.unet import UNetConfig, UNetEngine
    from spider_amd.vae import VAEConfig, VAEDecoderEngine
    root = str(tmp_path / "sd")
    (uc, vc, cc), (wu, wv, wc) = _write_sd_directory(root)
    pipe = StableDiffusionPipeline.from_pretrained(root, torch_dtype=torch.float16, device=dev)
    assert pipe.unet.dtype == torch.float16 and pipe.unet.stream32
    assert isinstance(pipe.scheduler, PNDMScheduler) and pipe.sample_size == 16
    assert not any(k.startswith(("encoder.", "quant_conv.")) for k in pipe.vae.w)        # the decoder half only
    vcfg = VAEConfig(**vc.__dict__)
    vcfg.scaling = 0.18215
    ref = StableDiffusionPipeline(UNetEngine(UNetConfig(**uc.__dict__), wu, dev, dtype=torch.float16, stream32=True),
                                  VAEDecoderEngine(vcfg, wv, dev, dtype=torch.float16),
                                  CLIPTextEngine(CLIPTextConfig(cc.vocab, cc.hidden, cc.layers, cc.heads, cc.inter, cc.max_pos), wc, dev,
                                                 dtype=torch.float16),
                                  CLIPTokenizer.from_pretrained(os.path.join(root, "tokenizer")), sample_size=16)
    lat = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(5))
    a = pipe(prompt=["a red door"], num_inference_steps=6, latents=lat, output_type="np").images
    b = ref(prompt=["a red door"], num_inference_steps=6, latents=lat, output_type="np").images
    assert np.isfinite(a).all() and a.std() > 1e-3          # the zero-filled fp16 variant was not what got loaded
    assert np.array_equal(a, b)


def test_spider_decoder_reaches_the_directory_from_its_config(dev, tmp_path):
    """Decoders-Controller config -> pipeline class by registry name -> from_pretrained(ckpt) on THIS rank's device
    (spider_decoder.py:104-119)."""
    from spider_amd.spider_decoder import SpiderDecoder
    root = str(tmp_path / "sd")
    _write_sd_directory(root)
    dec = SpiderDecoder(diffusion_modules={"IMAGE": {"type": "sd", "ckpt": root}}, device=dev,
                        decode_kwargs={"IMAGE": {"num_inference_steps": 4}})
    out = dec.decode_image({"llm_text_res": ["a red door"]}, num_inference_steps=4)
    assert len(out) == 1 and out[0].size == (64, 64)
    assert str(dec._pipes["IMAGE"].unet.device) == str(dev)
    # a directory that is not a checkpoint fails with the path in the message, not with a KeyError deep in an engine
    bad = SpiderDecoder(diffusion_modules={"IMAGE": {"type": "sd", "ckpt": str(tmp_path / "nothing")}}, device=dev)
    with pytest.raises(FileNotFoundError, match="nothing"):
        bad.decode_image({"llm_text_res": ["x"]})


# ------------------------------------------------------------------------------------------------------------------------------
# the video and audio decoders' directories (spider_decoder.py:122-166 -> base_model.py:207-219)

def _roberta_bpe_files(d):
    """A RoBERTa tokenizer with an empty merge table: specials, the byte alphabet, <mask>."""
    from tokenizers.pre_tokenizers import ByteLevel
    os.makedirs(d, exist_ok=True)
    vocab = ["<s>", "<pad>", "</s>", "<unk>"] + sorted(ByteLevel.alphabet()) + ["<mask>"]
    json.dump({t: i for i, t in enumerate(vocab)}, open(os.path.join(d, "vocab.json"), "w"))
    open(os.path.join(d, "merges.txt"), "w").write("#version: 0.2\n")
    json.dump({"model_max_length": 32, "bos_token": "<s>", "eos_token": "</s>", "unk_token": "<unk>", "pad_token": "<pad>",
               "cls_token": "<s>", "sep_token": "</s>", "mask_token": "<mask>", "tokenizer_class": "RobertaTokenizer"},
              open(os.path.join(d, "tokenizer_config.json"), "w"))
    return len(vocab)


def _half(w):
    return {k: v.to(torch.float16).contiguous() for k, v in w.items()}


def _write_component(root, name, config, weights, as_bin=False):
    os.makedirs(os.path.join(root, name))
    json.dump(config, open(os.path.join(root, name, "config.json"), "w"))
    stem = "diffusion_pytorch_model" if name in ("unet", "vae") else ("pytorch_model" if as_bin else "model")
    if as_bin:
        torch.save(weights, os.path.join(root, name, stem + ".bin"))
    else:
        save_file(weights, os.path.join(root, name, stem + ".safetensors"))


_DDIM = {"_class_name": "DDIMScheduler", "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear",
         "num_train_timesteps": 1000, "clip_sample": False, "set_alpha_to_one": False, "steps_offset": 1}


def test_video_directory_bin_only_loads_and_equals_direct_assembly(dev, tmp_path):
    """cerspense/zeroscope_v2_576w ships torch pickles only (unet/diffusion_pytorch_model.bin, vae/…bin, text_encoder/pytorch_model.bin)."""
    from transformers import CLIPTokenizer
    from oracle.clip_vae import CLIPCfg, VAECfg, clip_param_shapes, random_weights, vae_param_shapes
    from oracle.unet3d import UNet3DCfg, random_unet3d_weights
    from spider_amd.clip import CLIPTextConfig, CLIPTextEngine
    from spider_amd.pipelines import TextToVideoSDPipeline
    from spider_amd.schedulers import DDIMScheduler
    from spider_amd.unet3d import UNet3DConfig, UNet3DEngine
    from spider_amd.vae import VAEConfig, VAEDecoderEngine
    root = str(tmp_path / "vd")
    n_vocab = _clip_bpe_files(os.path.join(root, "tokenizer"))
    uc = UNet3DCfg(4, 4, (64, 128, 128), (True, True, False), (False, True, True), 32, 1, 64, 32, 8)   # transformer_in: always 8 heads in diffusers
    vc, cc = VAECfg.tiny(), CLIPCfg(n_vocab, 64, 2, 2, 128, 77)
    wu, wv, wc = _half(random_unet3d_weights(uc, 6)), _half(random_weights(vae_param_shapes(vc), 8)), _half(random_weights(clip_param_shapes(cc), 7))
    _write_component(root, "unet", {"_class_name": "UNet3DConditionModel", "in_channels": 4, "out_channels": 4, "block_out_channels": [64, 128, 128],
                                    "down_block_types": ["CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"],
                                    "up_block_types": ["UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"], "attention_head_dim": 32,
                                    "layers_per_block": 1, "cross_attention_dim": 64, "norm_num_groups": 32, "sample_size": 8}, wu, as_bin=True)
    _write_component(root, "vae", {"_class_name": "AutoencoderKL", "latent_channels": vc.latent, "out_channels": vc.out_ch,
                                   "block_out_channels": list(vc.block_out), "layers_per_block": vc.layers_per_block,
                                   "norm_num_groups": vc.groups, "scaling_factor": 0.18215}, wv, as_bin=True)
    _write_component(root, "text_encoder", {"vocab_size": cc.vocab, "hidden_size": cc.hidden, "num_hidden_layers": cc.layers,
                                            "num_attention_heads": cc.heads, "intermediate_size": cc.inter, "max_position_embeddings": cc.max_pos,
                                            "layer_norm_eps": 1e-5, "hidden_act": "quick_gelu"}, wc, as_bin=True)
    os.makedirs(os.path.join(root, "scheduler"))
    json.dump(_DDIM, open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    pipe = TextToVideoSDPipeline.from_pretrained(root, torch_dtype=torch.float16, device=dev)
    assert pipe.unet.stream32 and isinstance(pipe.scheduler, DDIMScheduler) and pipe.sample_size == 8
    ref = TextToVideoSDPipeline(UNet3DEngine(UNet3DConfig(**uc.__dict__), wu, dev, dtype=torch.float16, stream32=True),
                                VAEDecoderEngine(VAEConfig(**vc.__dict__), wv, dev, dtype=torch.float16),
                                CLIPTextEngine(CLIPTextConfig(cc.vocab, cc.hidden, cc.layers, cc.heads, cc.inter, cc.max_pos), wc, dev,
                                               dtype=torch.float16),
                                CLIPTokenizer.from_pretrained(os.path.join(root, "tokenizer")), DDIMScheduler(**{k: v for k, v in _DDIM.items() if not k.startswith("_")}),
                                sample_size=8)
    lat = torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(3))
    kw = dict(prompt=["a boat at sea"], num_frames=4, num_inference_steps=4, height=32, width=32, latents=lat, output_type="pt")
    a, b = pipe(**kw).frames, ref(**kw).frames
    assert torch.isfinite(a).all() and float(a.std()) > 1e-3 and torch.equal(a, b)


def test_audio_directory_loads_and_equals_direct_assembly(dev, tmp_path):
    """cvssp/audioldm-l-full layout: unet/ (class-conditioned, per-block cross_attention_dim list), vae/ (mel), text_encoder/
    (ClapTextModelWithProjection), tokenizer/ (RoBERTa BPE), vocoder/ (SpeechT5HifiGan), scheduler/."""
    from transformers import RobertaTokenizer
    from oracle.audio import ClapTextCfg, HifiGanCfg, clap_param_shapes, hifigan_param_shapes, random_weights
    from oracle.clip_vae import VAECfg, vae_param_shapes
    from oracle.unet import UNetCfg, random_unet_weights
    from spider_amd.clap import ClapTextConfig, ClapTextEngine
    from spider_amd.pipelines import AudioLDMPipeline
    from spider_amd.schedulers import DDIMScheduler
    from spider_amd.unet import UNetConfig, UNetEngine
    from spider_amd.vae import VAEConfig, VAEDecoderEngine
    from spider_amd.vocoder import HifiGanConfig, HifiGanEngine
    root = str(tmp_path / "ad")
    n_vocab = _roberta_bpe_files(os.path.join(root, "tokenizer"))
    ccfg = ClapTextCfg(n_vocab + 3, 64, 2, 4, 128, 40, 48, 1e-12, 1)
    ucfg = UNetCfg.tiny_audio()
    vcfg = VAECfg(latent=8, out_ch=1, block_out=(64, 128, 128), layers_per_block=1, scaling=0.9227)
    hcfg = HifiGanCfg(16, 16000, 64, (5, 4, 2), (16, 16, 8), (3, 7), ((1, 3, 5), (1, 3, 5)), 0.1, False)
    wu, wv = _half(random_unet_weights(ucfg, 22)), _half(random_weights(vae_param_shapes(vcfg), 23))
    wc, wh = _half(random_weights(clap_param_shapes(ccfg), 21)), _half(random_weights(hifigan_param_shapes(hcfg), 24))
    _write_component(root, "unet", {"_class_name": "UNet2DConditionModel", "in_channels": 8, "out_channels": 8, "block_out_channels": [64, 128, 128],
                                    "down_block_types": ["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
                                    "up_block_types": ["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"], "attention_head_dim": [2, 4, 4],
                                    "layers_per_block": 2, "cross_attention_dim": [64, 128, 128], "norm_num_groups": 32, "sample_size": 16,
                                    "class_embed_type": "simple_projection", "projection_class_embeddings_input_dim": 48,
                                    "class_embeddings_concat": True, "use_linear_projection": False}, wu)
    _write_component(root, "vae", {"_class_name": "AutoencoderKL", "latent_channels": 8, "out_channels": 1, "block_out_channels": [64, 128, 128],
                                   "layers_per_block": 1, "norm_num_groups": 32, "scaling_factor": 0.9227}, wv)
    _write_component(root, "text_encoder", {"architectures": ["ClapTextModelWithProjection"], "vocab_size": ccfg.vocab, "hidden_size": 64,
                                            "num_hidden_layers": 2, "num_attention_heads": 4, "intermediate_size": 128,
                                            "max_position_embeddings": 40, "projection_dim": 48, "layer_norm_eps": 1e-12, "pad_token_id": 1}, wc)
    _write_component(root, "vocoder", {"architectures": ["SpeechT5HifiGan"], "model_in_dim": 16, "sampling_rate": 16000,
                                       "upsample_initial_channel": 64, "upsample_rates": [5, 4, 2], "upsample_kernel_sizes": [16, 16, 8],
                                       "resblock_kernel_sizes": [3, 7], "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5]],
                                       "leaky_relu_slope": 0.1, "normalize_before": False}, wh)
    os.makedirs(os.path.join(root, "scheduler"))
    json.dump(_DDIM, open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    pipe = AudioLDMPipeline.from_pretrained(root, torch_dtype=torch.float16, device=dev)
    assert pipe.unet.stream32 and pipe.vae.cfg.scaling == 0.9227 and pipe.vocoder.cfg.sampling_rate == 16000
    ref = AudioLDMPipeline(VAEDecoderEngine(VAEConfig(**vcfg.__dict__), wv, dev, dtype=torch.float16),
                           ClapTextEngine(ClapTextConfig(**ccfg.__dict__), wc, dev, dtype=torch.float16),
                           RobertaTokenizer.from_pretrained(os.path.join(root, "tokenizer")),
                           UNetEngine(UNetConfig(**ucfg.__dict__), wu, dev, dtype=torch.float16, stream32=True),
                           DDIMScheduler(**{k: v for k, v in _DDIM.items() if not k.startswith("_")}),
                           HifiGanEngine(HifiGanConfig(**hcfg.__dict__), wh, dev, dtype=torch.float16), sample_size=16)
    g = lambda: torch.Generator(device=dev).manual_seed(5)
    a = pipe(prompt=["rain on a tin roof"], num_inference_steps=4, audio_length_in_s=0.5, generator=g()).audios
    b = ref(prompt=["rain on a tin roof"], num_inference_steps=4, audio_length_in_s=0.5, generator=g()).audios
    assert np.isfinite(a).all() and a.std() > 1e-4 and np.array_equal(a, b)


def test_sdxl_directory_loads_through_init_story_generation(dev, tmp_path):
    """stabilityai/stable-diffusion-xl-base-1.0 layout (Comic_Generation.py:297-318): unet/ (text_time conditioning, per-block
    transformer depth, linear projections), vae/ with force_upcast (diffusers runs that VAE in fp32: the engine takes bf16),
    text_encoder/ (CLIPTextModel), text_encoder_2/ (CLIPTextModelWithProjection, gelu), tokenizer/ + tokenizer_2/ (pad token '!')."""
    from transformers import CLIPTokenizer
    from oracle.clip_vae import CLIPCfg, VAECfg, clip_param_shapes, random_weights, vae_param_shapes
    from oracle.unet import UNetCfg, random_unet_weights
    from spider_amd.clip import CLIPTextConfig, CLIPTextEngine
    from spider_amd.schedulers import DDIMScheduler
    from spider_amd.story import StableDiffusionXLPipeline, init_story_generation
    from spider_amd.unet import UNetConfig, UNetEngine
    from spider_amd.vae import VAEConfig, VAEDecoderEngine
    root = str(tmp_path / "sdxl")
    n_vocab = _clip_bpe_files(os.path.join(root, "tokenizer"))
    _clip_bpe_files(os.path.join(root, "tokenizer_2"))
    tc = json.load(open(os.path.join(root, "tokenizer_2", "tokenizer_config.json")))
    tc["pad_token"] = "!"
    json.dump(tc, open(os.path.join(root, "tokenizer_2", "tokenizer_config.json"), "w"))
    uc = UNetCfg.tiny(sdxl_like=True)          # cross_dim 64 = 32 + 32 (two text encoders), pooled 64, 6 time ids x 32
    c1 = CLIPCfg(n_vocab, 32, 2, 2, 64, 77)
    vc = VAECfg(4, 3, (64, 64, 64, 128), 1, 32)
    wu, wv = _half(random_unet_weights(uc, seed=9)), _half(random_weights(vae_param_shapes(vc), seed=3))
    w1, w2 = _half(random_weights(clip_param_shapes(c1), seed=1)), _half(random_weights(clip_param_shapes(c1), seed=2))
    w2["text_projection.weight"] = (torch.randn(64, 32, generator=torch.Generator().manual_seed(4)) * 0.1).half()
    _write_component(root, "unet", {"_class_name": "UNet2DConditionModel", "in_channels": 4, "out_channels": 4, "block_out_channels": [64, 128, 128],
                                    "down_block_types": ["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
                                    "up_block_types": ["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"], "attention_head_dim": [1, 2, 2],
                                    "transformer_layers_per_block": [1, 2, 2], "layers_per_block": 2, "cross_attention_dim": 64,
                                    "norm_num_groups": 32, "sample_size": 16, "use_linear_projection": True, "addition_embed_type": "text_time",
                                    "addition_time_embed_dim": 32, "projection_class_embeddings_input_dim": 64 + 6 * 32}, wu)
    _write_component(root, "vae", {"_class_name": "AutoencoderKL", "latent_channels": 4, "out_channels": 3, "block_out_channels": [64, 64, 64, 128],
                                   "layers_per_block": 1, "norm_num_groups": 32, "scaling_factor": 0.13025, "force_upcast": True}, wv)
    te = {"vocab_size": c1.vocab, "hidden_size": 32, "num_hidden_layers": 2, "num_attention_heads": 2, "intermediate_size": 64,
          "max_position_embeddings": 77, "layer_norm_eps": 1e-5}
    _write_component(root, "text_encoder", {**te, "architectures": ["CLIPTextModel"], "hidden_act": "quick_gelu"}, w1)
    _write_component(root, "text_encoder_2", {**te, "architectures": ["CLIPTextModelWithProjection"], "hidden_act": "gelu", "projection_dim": 64}, w2)
    pipe = init_story_generation(root, device=dev)
    assert pipe.unet.freeu == (0.6, 0.4, 1.1, 1.2) and pipe.unet.stream32 and pipe.unet.dtype == torch.float16
    assert pipe.vae.dtype == torch.bfloat16 and pipe.vae.cfg.scaling == 0.13025          # force_upcast -> the wide-range 16-bit format
    assert pipe.text_encoder_2.text_projection is not None and pipe.text_encoder.text_projection is None
    assert pipe.tokenizer_2.pad_token == "!" and pipe.tokenizer.pad_token == "<|endoftext|>"
    vcfg = VAEConfig(**vc.__dict__)
    vcfg.scaling = 0.13025
    ref = StableDiffusionXLPipeline(UNetEngine(UNetConfig(**uc.__dict__), wu, dev, dtype=torch.float16, stream32=True),
                                    VAEDecoderEngine(vcfg, wv, dev, dtype=torch.bfloat16),
                                    CLIPTextEngine(CLIPTextConfig(c1.vocab, 32, 2, 2, 64, 77), w1, dev, dtype=torch.float16),
                                    CLIPTextEngine(CLIPTextConfig(c1.vocab, 32, 2, 2, 64, 77, 1e-5, "gelu"), w2, dev, dtype=torch.float16),
                                    CLIPTokenizer.from_pretrained(os.path.join(root, "tokenizer")),
                                    CLIPTokenizer.from_pretrained(os.path.join(root, "tokenizer_2")), DDIMScheduler())
    ref.enable_freeu(0.6, 0.4, 1.1, 1.2)
    lat = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(2))
    kw = dict(prompt=["a lighthouse"], num_inference_steps=5, height=64, width=64, latents=lat, output_type="np")
    a, b = pipe(**kw).images, ref(**kw).images
    assert np.isfinite(a).all() and a.std() > 1e-3 and np.array_equal(a, b)


def test_llama_directory_sharded_with_generation_config(dev, tmp_path):
    """deepseek-ai/DeepSeek-R1-Distill-Llama-8B layout (r1_llama3_8B_infer.py:4, demo/inference_api.py:92-95): config.json with
    GQA + llama3 rope scaling, model-0000x-of-0000y.safetensors behind model.safetensors.index.json, generation_config.json whose eos
    list ends generation when the caller passes none. The loaded engine generates what an engine built on the tensors generates."""
    from oracle.llama import LlamaCfg, LlamaOracle
    from spider_amd.llm import LlamaEngine, LLMConfig
    d = str(tmp_path / "llm")
    os.makedirs(d)
    rs = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=64)
    ocfg = LlamaCfg(256, 2, 4, 2, 128, 512, 331, 500000.0, rs, 1e-5, False, 512)
    w = {k: v.bfloat16() for k, v in LlamaOracle.random_weights(ocfg, seed=2, std=0.08).items()}
    json.dump({"architectures": ["LlamaForCausalLM"], "model_type": "llama", "hidden_size": 256, "num_hidden_layers": 2,
               "num_attention_heads": 4, "num_key_value_heads": 2, "head_dim": 128, "intermediate_size": 512, "vocab_size": 331,
               "rope_theta": 500000.0, "rope_scaling": rs, "rms_norm_eps": 1e-5, "max_position_embeddings": 512,
               "tie_word_embeddings": False, "attention_bias": False, "bos_token_id": 1, "eos_token_id": 2, "torch_dtype": "bfloat16"},
              open(os.path.join(d, "config.json"), "w"))
    names = sorted(w)
    shard = {n: f"model-0000{1 + (i % 2)}-of-00002.safetensors" for i, n in enumerate(names)}
    for f in set(shard.values()):
        save_file({n: w[n].contiguous() for n in names if shard[n] == f}, os.path.join(d, f))
    json.dump({"metadata": {"total_size": 0}, "weight_map": shard}, open(os.path.join(d, "model.safetensors.index.json"), "w"))
    ref = LlamaEngine(LLMConfig(**ocfg.__dict__), w, dev, max_batch=1, max_len=96)
    ids = torch.randint(3, 331, (1, 11), generator=torch.Generator().manual_seed(6))
    full = ref.generate(input_ids=ids, max_new_tokens=12)
    stop_at = int(full[0, 11 + 5])                       # the 6th generated token becomes an EOS of the checkpoint's generation config
    json.dump({"bos_token_id": 1, "eos_token_id": [2, stop_at], "pad_token_id": 0, "max_new_tokens": 12},
              open(os.path.join(d, "generation_config.json"), "w"))
    eng = LlamaEngine.from_pretrained(d, dev, max_batch=1, max_len=96)
    assert eng.cfg == LLMConfig(**ocfg.__dict__) and eng.generation_config["eos_token_id"] == [2, stop_at]
    got = eng.generate(input_ids=ids)                    # no max_new_tokens / eos passed: the checkpoint's defaults apply
    first = next(i for i in range(12) if int(full[0, 11 + i]) == stop_at)
    assert torch.equal(got[0, :11 + first + 1], full[0, :11 + first + 1]) and got.shape[1] == 11 + first + 1
    assert torch.equal(eng.generate(input_ids=ids, max_new_tokens=12, eos_token_id=[]), full)
