"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the two models that bracket every
diffusion call of the reference's image decoder:
  * CLIP text encoder  -- `self.text_encoder(text_input_ids)[0]`, spider/models/custom_sd.py:306-310,352-356
                          (transformers CLIPTextModel; pinned against the installed transformers implementation
                          in tests/test_oracle_golden.py)
  * VAE decoder        -- `self.vae.decode(latents / 0.18215).sample`, `(image/2+0.5).clamp(0,1)`,
                          spider/models/custom_sd.py:386-393 (diffusers==0.25.0 AutoencoderKL; PARITY UNPINNED:
                          external, absent from the image, no reference test -- restated from the published code)
Weight names follow the HF / diffusers state dicts.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn.functional as F


@dataclass
class CLIPCfg:
    vocab: int = 49408
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    inter: int = 3072
    max_pos: int = 77
    eps: float = 1e-5
    act: str = "quick_gelu"

    @staticmethod
    def tiny():
        return CLIPCfg(400, 64, 2, 2, 128, 77)


def clip_param_shapes(c: CLIPCfg) -> dict:
    S = {"text_model.embeddings.token_embedding.weight": (c.vocab, c.hidden),
         "text_model.embeddings.position_embedding.weight": (c.max_pos, c.hidden)}
    for l in range(c.layers):
        p = f"text_model.encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            S[p + f"self_attn.{n}.weight"] = (c.hidden, c.hidden); S[p + f"self_attn.{n}.bias"] = (c.hidden,)
        for n in ("layer_norm1", "layer_norm2"):
            S[p + n + ".weight"] = (c.hidden,); S[p + n + ".bias"] = (c.hidden,)
        S[p + "mlp.fc1.weight"] = (c.inter, c.hidden); S[p + "mlp.fc1.bias"] = (c.inter,)
        S[p + "mlp.fc2.weight"] = (c.hidden, c.inter); S[p + "mlp.fc2.bias"] = (c.hidden,)
    S["text_model.final_layer_norm.weight"] = (c.hidden,); S["text_model.final_layer_norm.bias"] = (c.hidden,)
    return S


def random_weights(shapes: dict, seed=0, bf16_round=True) -> dict:
    g = torch.Generator().manual_seed(seed)
    w = {}
    for n, shp in shapes.items():
        if n.endswith(".bias"):
            t = torch.randn(shp, generator=g) * 0.05
        elif "norm" in n and n.endswith(".weight"):
            t = 1.0 + torch.randn(shp, generator=g) * 0.1
        elif "embedding" in n:
            t = torch.randn(shp, generator=g) * 0.3
        else:
            t = torch.randn(shp, generator=g) / math.sqrt(math.prod(shp[1:]))
        w[n] = t.bfloat16().float() if bf16_round else t
    return w


def clip_text_forward(c: CLIPCfg, w: dict, ids: torch.Tensor) -> torch.Tensor:
    """ids [B,S] -> last_hidden_state [B,S,H] (causal mask, pre-LN, final LayerNorm)."""
    B, S = ids.shape
    h = w["text_model.embeddings.token_embedding.weight"][ids] + w["text_model.embeddings.position_embedding.weight"][:S][None]
    d = c.hidden // c.heads
    mask = torch.full((S, S), float("-inf")).triu(1)
    act = (lambda x: x * torch.sigmoid(1.702 * x)) if c.act == "quick_gelu" else F.gelu
    for l in range(c.layers):
        p = f"text_model.encoder.layers.{l}."
        x = F.layer_norm(h, (c.hidden,), w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], c.eps)
        q = F.linear(x, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.q_proj.bias"]) * d ** -0.5
        k = F.linear(x, w[p + "self_attn.k_proj.weight"], w[p + "self_attn.k_proj.bias"])
        v = F.linear(x, w[p + "self_attn.v_proj.weight"], w[p + "self_attn.v_proj.bias"])
        sh = lambda t: t.view(B, S, c.heads, d).transpose(1, 2)
        a = torch.softmax(sh(q) @ sh(k).transpose(-1, -2) + mask, -1) @ sh(v)
        a = a.transpose(1, 2).reshape(B, S, c.hidden)
        h = h + F.linear(a, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"])
        x = F.layer_norm(h, (c.hidden,), w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], c.eps)
        h = h + F.linear(act(F.linear(x, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"])), w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
    return F.layer_norm(h, (c.hidden,), w["text_model.final_layer_norm.weight"], w["text_model.final_layer_norm.bias"], c.eps)


# ------------------------------------------------------------------------------------------------ VAE decoder
@dataclass
class VAECfg:
    latent: int = 4
    out_ch: int = 3
    block_out: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    groups: int = 32
    scaling: float = 0.18215

    @staticmethod
    def tiny():
        return VAECfg(4, 3, (64, 64, 128), 1, 32)


def vae_param_shapes(c: VAECfg) -> dict:
    S = {}
    def conv(n, co, ci, k): S[n + ".weight"] = (co, ci, k, k); S[n + ".bias"] = (co,)
    def lin(n, co, ci): S[n + ".weight"] = (co, ci); S[n + ".bias"] = (co,)
    def norm(n, ch): S[n + ".weight"] = (ch,); S[n + ".bias"] = (ch,)
    def resnet(n, ci, co):
        norm(n + ".norm1", ci); conv(n + ".conv1", co, ci, 3); norm(n + ".norm2", co); conv(n + ".conv2", co, co, 3)
        if ci != co: conv(n + ".conv_shortcut", co, ci, 1)
    conv("post_quant_conv", c.latent, c.latent, 1)
    top = c.block_out[-1]
    conv("decoder.conv_in", top, c.latent, 3)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        lin(f"{a}.{n}", top, top)
    resnet("decoder.mid_block.resnets.1", top, top)
    rev = list(reversed(c.block_out))
    prev = rev[0]
    for i, co in enumerate(rev):
        for j in range(c.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
        prev = co
        if i != len(rev) - 1: conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    norm("decoder.conv_norm_out", rev[-1]); conv("decoder.conv_out", c.out_ch, rev[-1], 3)
    return S


def vae_decode(c: VAECfg, w: dict, latents: torch.Tensor, to_image: bool = True) -> torch.Tensor:
    """latents [B,4,h,w] (scheduler space) -> image [B,3,8h,8w] in [0,1] (custom_sd.py:386-393).
    to_image=False: the raw decoder output, i.e. AudioLDM's mel spectrogram (custom_ad.py:287-291)."""
    gn = lambda n, x: F.group_norm(x, c.groups, w[n + ".weight"], w[n + ".bias"], 1e-6)
    conv = lambda n, x, pad=1: F.conv2d(x, w[n + ".weight"], w[n + ".bias"], padding=pad)
    lin = lambda n, x: F.linear(x, w[n + ".weight"], w[n + ".bias"])
    def resnet(n, x):
        h = conv(n + ".conv1", F.silu(gn(n + ".norm1", x)))
        h = conv(n + ".conv2", F.silu(gn(n + ".norm2", h)))
        if n + ".conv_shortcut.weight" in w: x = conv(n + ".conv_shortcut", x, 0)
        return x + h
    z = conv("post_quant_conv", latents / c.scaling, 0)
    h = conv("decoder.conv_in", z)
    h = resnet("decoder.mid_block.resnets.0", h)
    a = "decoder.mid_block.attentions.0"
    B, C, H, W = h.shape
    x = gn(a + ".group_norm", h).view(B, C, H * W).transpose(1, 2)
    q, k, v = lin(a + ".to_q", x), lin(a + ".to_k", x), lin(a + ".to_v", x)
    o = torch.softmax(q @ k.transpose(1, 2) * C ** -0.5, -1) @ v
    h = h + lin(a + ".to_out.0", o).transpose(1, 2).reshape(B, C, H, W)
    h = resnet("decoder.mid_block.resnets.1", h)
    n = len(c.block_out)
    for i in range(n):
        for j in range(c.layers_per_block + 1):
            h = resnet(f"decoder.up_blocks.{i}.resnets.{j}", h)
        if i != n - 1:
            h = conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    img = conv("decoder.conv_out", F.silu(gn("decoder.conv_norm_out", h)))
    return (img / 2 + 0.5).clamp(0, 1) if to_image else img
