"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the two stock networks that bracket
the AudioLDM denoising loop of the reference (spider/models/custom_ad.py):

  * the CLAP text branch -- `self.text_encoder(ids, attention_mask).text_embeds` followed by `F.normalize`
    (custom_ad.py:214-219 cond, :266-273 uncond). The class is transformers' `ClapTextModelWithProjection`
    (imported at custom_ad.py:22); its arithmetic is RoBERTa (post-LN encoder, padding-offset position ids),
    a tanh pooler on token 0 and a Linear-ReLU-Linear projection.
  * the HiFi-GAN vocoder -- `self.vocoder(mel_spectrogram)` in mel_spectrogram_to_waveform
    (custom_ad.py:293-300); the class is transformers' `SpeechT5HifiGan`.

Both live in the third-party dependency `transformers` (reference pins 4.43.1 / 4.50.0, requirements*.txt:5);
this image has 5.15.0. PINNED: tests/golden/make_golden_audio.py runs those two transformers classes here on tiny
seeded configs and commits weights + inputs + outputs (tests/golden/clap_text_ref.npz, hifigan_ref.npz);
tests/test_oracle_golden.py checks this file against them.
Weight names are the transformers state-dict names, so a real checkpoint loads into either side.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------- CLAP text
@dataclass
class ClapTextCfg:
    vocab: int = 50265
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    inter: int = 3072
    max_pos: int = 514
    proj_dim: int = 512
    eps: float = 1e-12
    pad_id: int = 1

    @staticmethod
    def tiny():
        return ClapTextCfg(100, 64, 2, 4, 128, 40, 32, 1e-12, 1)


def clap_position_ids(ids: torch.Tensor, pad_id: int) -> torch.Tensor:
    """RoBERTa create_position_ids_from_input_ids: non-pad tokens count up from pad_id+1, pads stay at pad_id."""
    m = (ids != pad_id).int()
    return (torch.cumsum(m, 1) * m).long() + pad_id


@torch.no_grad()
def clap_text_embeds(w: Dict[str, torch.Tensor], cfg: ClapTextCfg, ids: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """ids, mask [B,S] -> text_embeds [B, proj_dim] (NOT yet L2-normalised; the pipeline does that)."""
    p = "text_model."
    lin = lambda n, x: F.linear(x, w[n + ".weight"], w[n + ".bias"])
    ln = lambda n, x: F.layer_norm(x, (cfg.hidden,), w[n + ".weight"], w[n + ".bias"], cfg.eps)
    e = p + "embeddings."
    h = w[e + "word_embeddings.weight"][ids] + w[e + "position_embeddings.weight"][clap_position_ids(ids, cfg.pad_id)] \
        + w[e + "token_type_embeddings.weight"][0]
    h = ln(e + "LayerNorm", h)
    B, S, H = h.shape
    d = H // cfg.heads
    bias = (1.0 - mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    for l in range(cfg.layers):
        a = f"{p}encoder.layer.{l}."
        sh = lambda t: t.view(B, S, cfg.heads, d).transpose(1, 2)
        q, k, v = sh(lin(a + "attention.self.query", h)), sh(lin(a + "attention.self.key", h)), sh(lin(a + "attention.self.value", h))
        pr = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d) + bias, -1)
        ctx = (pr @ v).transpose(1, 2).reshape(B, S, H)
        h = ln(a + "attention.output.LayerNorm", lin(a + "attention.output.dense", ctx) + h)
        m = lin(a + "output.dense", F.gelu(lin(a + "intermediate.dense", h)))
        h = ln(a + "output.LayerNorm", m + h)
    pooled = torch.tanh(lin(p + "pooler.dense", h[:, 0]))
    return lin("text_projection.linear2", F.relu(lin("text_projection.linear1", pooled)))


def clap_param_shapes(c: ClapTextCfg) -> Dict[str, Tuple[int, ...]]:
    S = {"text_model.embeddings.word_embeddings.weight": (c.vocab, c.hidden),
         "text_model.embeddings.position_embeddings.weight": (c.max_pos, c.hidden),
         "text_model.embeddings.token_type_embeddings.weight": (1, c.hidden),
         "text_model.embeddings.LayerNorm.weight": (c.hidden,), "text_model.embeddings.LayerNorm.bias": (c.hidden,)}
    for l in range(c.layers):
        a = f"text_model.encoder.layer.{l}."
        for n, (o, i) in {"attention.self.query": (c.hidden, c.hidden), "attention.self.key": (c.hidden, c.hidden),
                          "attention.self.value": (c.hidden, c.hidden), "attention.output.dense": (c.hidden, c.hidden),
                          "intermediate.dense": (c.inter, c.hidden), "output.dense": (c.hidden, c.inter)}.items():
            S[a + n + ".weight"] = (o, i); S[a + n + ".bias"] = (o,)
        for n in ("attention.output.LayerNorm", "output.LayerNorm"):
            S[a + n + ".weight"] = (c.hidden,); S[a + n + ".bias"] = (c.hidden,)
    S["text_model.pooler.dense.weight"] = (c.hidden, c.hidden); S["text_model.pooler.dense.bias"] = (c.hidden,)
    S["text_projection.linear1.weight"] = (c.proj_dim, c.hidden); S["text_projection.linear1.bias"] = (c.proj_dim,)
    S["text_projection.linear2.weight"] = (c.proj_dim, c.proj_dim); S["text_projection.linear2.bias"] = (c.proj_dim,)
    return S


# ---------------------------------------------------------------------------------------------- HiFi-GAN
@dataclass
class HifiGanCfg:
    model_in_dim: int = 64
    sampling_rate: int = 16000
    upsample_initial_channel: int = 1024
    upsample_rates: Tuple[int, ...] = (5, 4, 2, 2, 2)
    upsample_kernel_sizes: Tuple[int, ...] = (16, 16, 8, 4, 4)
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    leaky_relu_slope: float = 0.1
    normalize_before: bool = False

    @staticmethod
    def tiny():
        return HifiGanCfg(16, 16000, 64, (5, 4, 2), (16, 16, 8), (3, 7), ((1, 3, 5), (1, 3, 5)), 0.1, True)


def hifigan_param_shapes(c: HifiGanCfg) -> Dict[str, Tuple[int, ...]]:
    S = {"mean": (c.model_in_dim,), "scale": (c.model_in_dim,),
         "conv_pre.weight": (c.upsample_initial_channel, c.model_in_dim, 7), "conv_pre.bias": (c.upsample_initial_channel,)}
    ch = c.upsample_initial_channel
    for i, k in enumerate(c.upsample_kernel_sizes):
        S[f"upsampler.{i}.weight"] = (ch, ch // 2, k); S[f"upsampler.{i}.bias"] = (ch // 2,)
        ch //= 2
        for j, (rk, dils) in enumerate(zip(c.resblock_kernel_sizes, c.resblock_dilation_sizes)):
            r = f"resblocks.{i * len(c.resblock_kernel_sizes) + j}."
            for u in range(len(dils)):
                for cv in ("convs1", "convs2"):
                    S[f"{r}{cv}.{u}.weight"] = (ch, ch, rk); S[f"{r}{cv}.{u}.bias"] = (ch,)
    S["conv_post.weight"] = (1, ch, 7); S["conv_post.bias"] = (1,)
    return S


@torch.no_grad()
def hifigan_forward(w: Dict[str, torch.Tensor], c: HifiGanCfg, mel: torch.Tensor) -> torch.Tensor:
    """mel [B, frames, model_in_dim] -> waveform [B, samples] (SpeechT5HifiGan.forward, batched form)."""
    if c.normalize_before:
        mel = (mel - w["mean"]) / w["scale"]
    h = F.conv1d(mel.transpose(2, 1), w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
    nk = len(c.resblock_kernel_sizes)
    for i, (r, k) in enumerate(zip(c.upsample_rates, c.upsample_kernel_sizes)):
        h = F.leaky_relu(h, c.leaky_relu_slope)
        h = F.conv_transpose1d(h, w[f"upsampler.{i}.weight"], w[f"upsampler.{i}.bias"], stride=r, padding=(k - r) // 2)
        acc = None
        for j, (rk, dils) in enumerate(zip(c.resblock_kernel_sizes, c.resblock_dilation_sizes)):
            p = f"resblocks.{i * nk + j}."
            x = h
            for u, dl in enumerate(dils):
                t = F.leaky_relu(x, c.leaky_relu_slope)
                t = F.conv1d(t, w[f"{p}convs1.{u}.weight"], w[f"{p}convs1.{u}.bias"], dilation=dl, padding=(rk * dl - dl) // 2)
                t = F.leaky_relu(t, c.leaky_relu_slope)
                t = F.conv1d(t, w[f"{p}convs2.{u}.weight"], w[f"{p}convs2.{u}.bias"], padding=(rk - 1) // 2)
                x = t + x
            acc = x if acc is None else acc + x
        h = acc / nk
    h = F.leaky_relu(h)                      # default slope 0.01 here, as the reference class has it
    h = torch.tanh(F.conv1d(h, w["conv_post.weight"], w["conv_post.bias"], padding=3))
    return h.squeeze(1)


def random_weights(shapes: Dict[str, Tuple[int, ...]], seed=0, bf16_round=True) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    w = {}
    for n, shp in shapes.items():
        if n == "scale":
            t = 1.0 + torch.rand(shp, generator=g)
        elif n.endswith(".bias") or n == "mean":
            t = torch.randn(shp, generator=g) * 0.05
        elif "LayerNorm" in n:
            t = 1.0 + torch.randn(shp, generator=g) * 0.1
        elif "embeddings" in n:
            t = torch.randn(shp, generator=g) * 0.5
        else:
            t = torch.randn(shp, generator=g) / math.sqrt(math.prod(shp[1:]))
        w[n] = t.bfloat16().float() if bf16_round else t
    return w
