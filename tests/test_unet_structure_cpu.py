"""CPU structural pins for the parity-unpinned diffusers restatements (oracle/unet.py, oracle/clip_vae.py VAE): the
restated topologies have exactly the parameter counts of the published checkpoints and use the diffusers state-dict key
layout (what `UNet2DConditionModel.from_pretrained` / `AutoencoderKL.from_pretrained` read for custom_sd.py:634-639,
custom_ad.py:575-581, Comic_Generation.py:313-317). This pins topology and naming, not arithmetic: diffusers itself is
absent from the image, which DESIGN.md section 4 records as "parity unpinned"."""
import math

import pytest

from oracle.clip_vae import VAECfg, vae_param_shapes
from oracle.unet import UNetCfg, unet_param_shapes


def _count(S):
    return sum(math.prod(s) for s in S.values())


@pytest.mark.parametrize("name,n_params,n_tensors", [
    ("sd15", 859_520_964, 686),            # runwayml/stable-diffusion-v1-5 unet (train_configs/spider_decoder_cfg.py:35)
    ("sdxl", 2_567_463_684, 1680),         # stabilityai/stable-diffusion-xl-base-1.0 unet (Comic_Generation.py:313)
    ("audioldm", 185_036_552, 690),        # cvssp/audioldm-s-full-v2 unet
    ("audioldm_l", 739_139_080, 690),      # cvssp/audioldm-l-full unet (train_configs/spider_decoder_cfg.py:37): 4 x the s widths
])
def test_unet_parameter_counts(name, n_params, n_tensors):
    S = unet_param_shapes(getattr(UNetCfg, name)())
    assert len(S) == n_tensors
    assert _count(S) == n_params


def test_vae_decoder_parameter_count():
    S = vae_param_shapes(VAECfg())
    assert _count(S) == 49_490_179 + 20    # AutoencoderKL decoder + post_quant_conv (4x4 + 4) of the SD-v1.5 / SDXL VAE
    assert len(S) == 140
    assert S["decoder.conv_in.weight"] == (512, 4, 3, 3) and S["decoder.conv_out.weight"] == (3, 128, 3, 3)
    assert S["decoder.mid_block.attentions.0.to_q.weight"] == (512, 512)
    assert S["decoder.up_blocks.0.upsamplers.0.conv.weight"] == (512, 512, 3, 3)
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in S          # the last up block has no upsampler
    assert S["decoder.up_blocks.2.resnets.0.conv_shortcut.weight"] == (256, 512, 1, 1)


def test_sd15_state_dict_layout():
    S = unet_param_shapes(UNetCfg.sd15())
    expect = {
        "conv_in.weight": (320, 4, 3, 3), "time_embedding.linear_1.weight": (1280, 320), "time_embedding.linear_2.bias": (1280,),
        "down_blocks.0.resnets.0.norm1.weight": (320,), "down_blocks.0.resnets.0.conv1.weight": (320, 320, 3, 3),
        "down_blocks.0.resnets.0.time_emb_proj.weight": (320, 1280), "down_blocks.0.resnets.1.conv2.bias": (320,),
        "down_blocks.1.resnets.0.conv_shortcut.weight": (640, 320, 1, 1),
        "down_blocks.0.attentions.0.norm.weight": (320,), "down_blocks.0.attentions.0.proj_in.weight": (320, 320, 1, 1),
        "down_blocks.0.attentions.0.transformer_blocks.0.norm1.weight": (320,),
        "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight": (320, 320),
        "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0.weight": (320, 320),
        "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0.bias": (320,),
        "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight": (320, 768),
        "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_v.weight": (320, 768),
        "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight": (2560, 320),
        "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.2.weight": (320, 1280),
        "down_blocks.0.attentions.0.transformer_blocks.0.norm3.bias": (320,),
        "down_blocks.0.attentions.0.proj_out.weight": (320, 320, 1, 1),
        "down_blocks.0.downsamplers.0.conv.weight": (320, 320, 3, 3),
        "down_blocks.3.resnets.1.conv1.weight": (1280, 1280, 3, 3),
        "mid_block.resnets.0.conv1.weight": (1280, 1280, 3, 3), "mid_block.resnets.1.conv2.weight": (1280, 1280, 3, 3),
        "mid_block.attentions.0.transformer_blocks.0.attn2.to_k.weight": (1280, 768),
        "up_blocks.0.resnets.0.conv1.weight": (1280, 2560, 3, 3), "up_blocks.0.upsamplers.0.conv.weight": (1280, 1280, 3, 3),
        "up_blocks.1.resnets.2.conv1.weight": (1280, 1920, 3, 3), "up_blocks.2.resnets.2.conv1.weight": (640, 960, 3, 3),
        "up_blocks.3.resnets.0.conv1.weight": (320, 960, 3, 3), "up_blocks.3.resnets.2.conv_shortcut.weight": (320, 640, 1, 1),
        "up_blocks.3.attentions.2.transformer_blocks.0.attn1.to_v.weight": (320, 320),
        "conv_norm_out.weight": (320,), "conv_out.weight": (4, 320, 3, 3), "conv_out.bias": (4,),
    }
    for k, shp in expect.items():
        assert S.get(k) == shp, (k, S.get(k))
    absent = ["down_blocks.3.attentions.0.norm.weight", "down_blocks.3.downsamplers.0.conv.weight", "up_blocks.0.attentions.0.norm.weight",
              "up_blocks.3.upsamplers.0.conv.weight", "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.bias",
              "add_embedding.linear_1.weight", "class_embedding.weight"]
    for k in absent:
        assert k not in S, k
    # every key belongs to one of the module families of the diffusers layout
    heads = {k.split(".")[0] for k in S}
    assert heads == {"conv_in", "time_embedding", "down_blocks", "mid_block", "up_blocks", "conv_norm_out", "conv_out"}
    assert sum(1 for k in S if k.endswith("attn1.to_q.weight")) == 16 and sum(1 for k in S if ".resnets." in k and k.endswith("conv1.weight")) == 22


def test_sdxl_and_audioldm_layout():
    S = unet_param_shapes(UNetCfg.sdxl())
    assert S["add_embedding.linear_1.weight"] == (1280, 2816) and S["add_embedding.linear_2.weight"] == (1280, 1280)
    assert S["down_blocks.1.attentions.0.proj_in.weight"] == (640, 640)           # use_linear_projection
    assert S["down_blocks.2.attentions.1.transformer_blocks.9.attn2.to_k.weight"] == (1280, 2048)
    assert "down_blocks.0.attentions.0.norm.weight" not in S and "down_blocks.2.downsamplers.0.conv.weight" not in S
    assert S["mid_block.attentions.0.transformer_blocks.9.ff.net.0.proj.weight"] == (10240, 1280)
    assert S["up_blocks.0.attentions.2.transformer_blocks.9.attn1.to_q.weight"] == (1280, 1280)
    assert S["up_blocks.1.attentions.2.transformer_blocks.1.attn1.to_q.weight"] == (640, 640)
    assert sum(1 for k in S if k.endswith("attn1.to_q.weight")) == 70             # 10 + 60 transformer layers
    assert sum(1 for k in S if k.startswith("up_blocks") and k.endswith("attn1.to_q.weight")) == 36   # the 36 story processors
    for name, w in (("audioldm", 1), ("audioldm_l", 2)):
        A = unet_param_shapes(getattr(UNetCfg, name)())
        assert A["class_embedding.weight"] == (512 * w, 512)                      # simple_projection of the CLAP embedding
        assert A["down_blocks.0.resnets.0.time_emb_proj.weight"] == (128 * w, 2 * 512 * w)   # class_embeddings_concat doubles the width
        assert A["down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_k.weight"] == (256 * w, 256 * w)
        assert A["conv_in.weight"] == (128 * w, 8, 3, 3) and "down_blocks.0.attentions.0.norm.weight" not in A


def test_audioldm_l_loader_config():
    """The unet/config.json values of cvssp/audioldm-l-full (the AD checkpoint of train_configs/spider_decoder_cfg.py:37)
    parse to the audioldm_l preset, and the product / oracle presets agree."""
    from spider_amd.unet import UNetConfig
    cfg_json = {"act_fn": "silu", "attention_head_dim": 8, "block_out_channels": [256, 512, 768, 1280], "center_input_sample": False,
                "class_embed_type": "simple_projection", "class_embeddings_concat": True, "cross_attention_dim": [256, 512, 768, 1280],
                "down_block_types": ["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
                "downsample_padding": 1, "flip_sin_to_cos": True, "freq_shift": 0, "in_channels": 8, "layers_per_block": 2,
                "mid_block_type": "UNetMidBlock2DCrossAttn", "norm_eps": 1e-05, "norm_num_groups": 32, "out_channels": 8,
                "projection_class_embeddings_input_dim": 512, "sample_size": 128, "use_linear_projection": False,
                "up_block_types": ["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"]}
    got = UNetConfig.from_diffusers_dict(cfg_json)
    assert got == UNetConfig.audioldm_l()
    assert UNetConfig.audioldm_l().__dict__ == UNetCfg.audioldm_l().__dict__
    assert UNetConfig.audioldm().__dict__ == UNetCfg.audioldm().__dict__ and UNetConfig.sdxl().__dict__ == UNetCfg.sdxl().__dict__
