"""HiFi-GAN vocoder on the HIP kernels: `self.vocoder(mel_spectrogram)` of the reference's AudioLDM pipeline
(spider/models/custom_ad.py:293-300, class SpeechT5HifiGan imported at custom_ad.py:22).

Channels-last [B, L, C] bf16 throughout (a mel spectrogram [B, frames, n_mel] already is). Every Conv1d is the
implicit-GEMM conv with a 1 x k dilated kernel (spider_conv_nhwc_ex_bf16); the second leaky-ReLU of each residual unit
and the residual add live in conv epilogues. ConvTranspose1d = one GEMM against the per-tap weight matrix
[k*Cout, Cin] (fp32 output) + an overlap-add kernel, i.e. no zero-stuffed input and no wasted MACs."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch

from . import ops

BF16 = torch.bfloat16


@dataclass
class HifiGanConfig:
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
    def audioldm():   # cvssp/audioldm-s-full-v2 vocoder/config.json
        return HifiGanConfig()

    @staticmethod
    def from_hf_dict(c: dict):
        return HifiGanConfig(c.get("model_in_dim", 80), c.get("sampling_rate", 16000), c.get("upsample_initial_channel", 512),
                             tuple(c.get("upsample_rates", (4, 4, 4, 4))), tuple(c.get("upsample_kernel_sizes", (8, 8, 8, 8))),
                             tuple(c.get("resblock_kernel_sizes", (3, 7, 11))),
                             tuple(tuple(d) for d in c.get("resblock_dilation_sizes", ((1, 3, 5),) * 3)),
                             c.get("leaky_relu_slope", 0.1), c.get("normalize_before", True))


def _shapes(c: HifiGanConfig) -> dict:
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


class HifiGanEngine:
    def __init__(self, cfg: HifiGanConfig, weights: Dict[str, torch.Tensor], device="cuda:0", dtype=BF16):
        assert dtype in (torch.bfloat16, torch.float16), "HifiGanEngine: dtype must be bfloat16 or float16"
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        BF16 = dtype
        dv = self.device
        self.w: Dict[str, torch.Tensor] = {}
        for n, t in weights.items():
            t = t.to(dv)
            if n.startswith("upsampler.") and n.endswith(".weight"):      # [Cin, Cout, k] -> per-tap rows [k*Cout, Cin]
                cin, cout, k = t.shape
                t = t.permute(2, 1, 0).reshape(k * cout, cin)
            elif t.ndim == 3:                                             # Conv1d [Cout, Cin, k] -> [Cout, k, Cin]
                t = t.permute(0, 2, 1)
            self.w[n] = t.to(BF16).contiguous() if n not in ("mean", "scale") else t.float().contiguous()
        # conv_post has one output channel: pad to the 4-column granularity of the GEMM epilogue
        wp = torch.zeros(4, *self.w["conv_post.weight"].shape[1:], dtype=BF16, device=dv)
        wp[:1] = self.w["conv_post.weight"]
        bp = torch.zeros(4, dtype=BF16, device=dv)
        bp[:1] = self.w["conv_post.bias"]
        self.w["conv_post.weight4"], self.w["conv_post.bias4"] = wp.contiguous(), bp

    @classmethod
    def random_init(cls, cfg: HifiGanConfig, device="cuda:0", seed=0, dtype=BF16):
        gen = torch.Generator(device=device).manual_seed(seed)
        w = {}
        for n, shp in _shapes(cfg).items():
            if n == "scale":
                t = torch.ones(shp, device=device)
            elif n.endswith(".bias") or n == "mean":
                t = torch.zeros(shp, device=device)
            else:
                t = torch.randn(shp, generator=gen, device=device) / math.sqrt(math.prod(shp[1:]))
            w[n] = t.to(BF16)
        return cls(cfg, w, device, dtype=dtype)

    @classmethod
    def from_pretrained(cls, path: str, device="cuda:0", dtype=BF16):
        from .checkpoint import load_state_dict, read_config
        cfg = HifiGanConfig.from_hf_dict(read_config(path))
        return cls(cfg, load_state_dict(path), device, dtype=dtype)

    @property
    def config(self):   # the pipeline reads vocoder.config.{upsample_rates, sampling_rate, model_in_dim} (custom_ad.py:490-500)
        return self.cfg

    @torch.no_grad()
    def __call__(self, mel: torch.Tensor) -> torch.Tensor:
        """mel [B, frames, model_in_dim] (any float dtype) -> waveform [B, samples] fp32."""
        c, w = self.cfg, self.w
        mel = mel.to(self.device).float()
        if c.normalize_before:
            mel = (mel - w["mean"]) / w["scale"]
        slope = c.leaky_relu_slope
        x = mel.to(self.dtype).contiguous()
        # conv_pre; its output is only ever read through the first leaky-ReLU, so that is fused here
        h = ops.conv1d(x, w["conv_pre.weight"], bias=w["conv_pre.bias"], pad=3, act="leaky_relu", act_param=slope)
        nk = len(c.resblock_kernel_sizes)
        for i, (r, k) in enumerate(zip(c.upsample_rates, c.upsample_kernel_sizes)):
            if i > 0:
                h = ops.act(h, "leaky_relu", slope)
            h = ops.conv_transpose1d(h, w[f"upsampler.{i}.weight"], w[f"upsampler.{i}.bias"], k, r, (k - r) // 2)
            h_act = ops.act(h, "leaky_relu", slope)              # shared first activation of the nk parallel branches
            acc = None
            for j, (rk, dils) in enumerate(zip(c.resblock_kernel_sizes, c.resblock_dilation_sizes)):
                p = f"resblocks.{i * nk + j}."
                xb = h
                for u, dl in enumerate(dils):
                    t = h_act if u == 0 else ops.act(xb, "leaky_relu", slope)
                    t = ops.conv1d(t, w[f"{p}convs1.{u}.weight"], bias=w[f"{p}convs1.{u}.bias"], pad=(rk * dl - dl) // 2, dil=dl,
                                   act="leaky_relu", act_param=slope)
                    xb = ops.conv1d(t, w[f"{p}convs2.{u}.weight"], bias=w[f"{p}convs2.{u}.bias"], pad=(rk - 1) // 2, res=xb)
                if acc is None:
                    acc = xb
                elif j < nk - 1:
                    acc = ops.add(acc, xb)
                else:
                    acc = ops.add_scaled(acc, xb, 1.0 / nk)
            h = acc if nk > 1 else acc
        h = ops.act(h, "leaky_relu", 0.01)                       # F.leaky_relu default slope before conv_post
        y = ops.conv1d(h, w["conv_post.weight4"], bias=w["conv_post.bias4"], pad=3, act="tanh")
        return y[..., 0].float()
