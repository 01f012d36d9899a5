"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the text-to-video UNet
(`UNet3DConditionModel`) the reference's TextToVideoSDPipeline drives (spider/models/custom_vd.py:671-676, checkpoint
zeroscope / modelscope text-to-video) and of that pipeline's latent plumbing (custom_vd.py:664-697 loop with the
per-step [B,C,F,H,W] <-> [B*F,C,H,W] reshapes :684-692, decode_latents :381-408, tensor2vid :59-74).

PARITY UNPINNED, like oracle/unet.py: the network lives in the third-party dependency `diffusers==0.25.0`
(requirements.txt:11; imported at custom_vd.py:25), absent from /root/reference and from this image; the reference holds
no test or vector for it. This file restates diffusers 0.25.0's published modules: UNet3DConditionModel,
CrossAttnDownBlock3D / DownBlock3D / UNetMidBlock3DCrossAttn / (CrossAttn)UpBlock3D, TemporalConvLayer,
TransformerTemporalModel (BasicTransformerBlock with double_self_attention) on top of the 2-D pieces of oracle/unet.py.
Weight names are diffusers' state-dict names.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn.functional as F

from .unet import UNetCfg, UNetOracle, random_unet_weights, unet_param_shapes


@dataclass
class UNet3DCfg:
    in_ch: int = 4
    out_ch: int = 4
    block_out: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_attn: Tuple[bool, ...] = (True, True, True, False)    # CrossAttnDownBlock3D x3, DownBlock3D
    up_attn: Tuple[bool, ...] = (False, True, True, True)
    head_dim: int = 64            # config key `attention_head_dim`; heads per block = channels // head_dim
    layers_per_block: int = 2
    cross_dim: int = 1024
    groups: int = 32
    tin_heads: int = 8            # transformer_in: 8 heads of head_dim

    @staticmethod
    def zeroscope():
        return UNet3DCfg()

    @staticmethod
    def tiny():
        return UNet3DCfg(4, 4, (64, 128, 128), (True, True, False), (False, True, True), 32, 1, 64, 32, 2)

    def as2d(self) -> UNetCfg:
        nb = len(self.block_out)
        return UNetCfg(self.in_ch, self.out_ch, self.block_out, self.down_attn, self.up_attn, (1,) * nb,
                       tuple(c // self.head_dim for c in self.block_out), self.layers_per_block, self.cross_dim, self.groups,
                       True, 0, 0, None)


def unet3d_param_shapes(c: UNet3DCfg) -> dict:
    c2 = c.as2d()
    S = dict(unet_param_shapes(c2))
    nb = len(c.block_out)

    def norm(n, ch): S[n + ".weight"] = (ch,); S[n + ".bias"] = (ch,)
    def lin(n, co, ci, bias=True):
        S[n + ".weight"] = (co, ci)
        if bias: S[n + ".bias"] = (co,)

    def temp_conv(n, ch):   # TemporalConvLayer: conv1 = [GN, SiLU, Conv3d], conv2..4 = [GN, SiLU, Dropout, Conv3d]
        for i, ci in ((1, 2), (2, 3), (3, 3), (4, 3)):
            norm(f"{n}.conv{i}.0", ch)
            S[f"{n}.conv{i}.{ci}.weight"] = (ch, ch, 3, 1, 1); S[f"{n}.conv{i}.{ci}.bias"] = (ch,)

    def temp_tr(n, ch, heads):
        inner = heads * c.head_dim
        norm(n + ".norm", ch); lin(n + ".proj_in", inner, ch); lin(n + ".proj_out", ch, inner)
        b = n + ".transformer_blocks.0"
        for k in ("norm1", "norm2", "norm3"): norm(f"{b}.{k}", inner)
        for a in ("attn1", "attn2"):
            for p in ("to_q", "to_k", "to_v"): lin(f"{b}.{a}.{p}", inner, inner, False)
            lin(f"{b}.{a}.to_out.0", inner, inner)
        lin(b + ".ff.net.0.proj", 8 * inner, inner); lin(b + ".ff.net.2", inner, 4 * inner)

    temp_tr("transformer_in", c.block_out[0], c.tin_heads)
    for i, co in enumerate(c.block_out):
        for j in range(c.layers_per_block):
            temp_conv(f"down_blocks.{i}.temp_convs.{j}", co)
            if c.down_attn[i]: temp_tr(f"down_blocks.{i}.temp_attentions.{j}", co, co // c.head_dim)
    cm = c.block_out[-1]
    temp_conv("mid_block.temp_convs.0", cm); temp_conv("mid_block.temp_convs.1", cm)
    temp_tr("mid_block.temp_attentions.0", cm, cm // c.head_dim)
    for i, co in enumerate(reversed(c.block_out)):
        for j in range(c.layers_per_block + 1):
            temp_conv(f"up_blocks.{i}.temp_convs.{j}", co)
            if c.up_attn[i]: temp_tr(f"up_blocks.{i}.temp_attentions.{j}", co, co // c.head_dim)
    return S


def random_unet3d_weights(c: UNet3DCfg, seed=0) -> dict:
    import math
    g = torch.Generator().manual_seed(seed)
    w = {}
    for n, shp in unet3d_param_shapes(c).items():
        last2 = n.split(".")[-2]
        if n.endswith(".bias"):
            t = torch.randn(shp, generator=g) * 0.05
        elif len(shp) == 1:            # every 1-D weight is a norm scale (GroupNorm / LayerNorm)
            t = 1.0 + torch.randn(shp, generator=g) * 0.1
        else:
            t = torch.randn(shp, generator=g) * (1.0 / math.sqrt(math.prod(shp[1:])))
        w[n] = t.bfloat16().float()
    return w


class UNet3DOracle(UNetOracle):
    def __init__(self, cfg: UNet3DCfg, weights: dict, dtype=torch.float32):
        super().__init__(cfg.as2d(), weights, dtype)
        self.c3 = cfg

    def temp_conv(self, n, x, frames):
        """x [(B F), C, H, W] -> same; identity + 4 x [GroupNorm (over C/G x F x H x W), SiLU, Conv3d (3,1,1)]."""
        BF, C, H, W = x.shape
        h = x[None, :].reshape(-1, frames, C, H, W).permute(0, 2, 1, 3, 4)
        ident = h
        for i, ci in ((1, 2), (2, 3), (3, 3), (4, 3)):
            h = F.silu(F.group_norm(h, self.cfg.groups, self.w[f"{n}.conv{i}.0.weight"], self.w[f"{n}.conv{i}.0.bias"], 1e-5))
            h = F.conv3d(h, self.w[f"{n}.conv{i}.{ci}.weight"], self.w[f"{n}.conv{i}.{ci}.bias"], padding=(1, 0, 0))
        h = ident + h
        return h.permute(0, 2, 1, 3, 4).reshape(BF, C, H, W)

    def temp_transformer(self, n, x, frames, heads):
        """TransformerTemporalModel: attention along the frame axis for every (batch, pixel); attn2 is a second
        self-attention (double_self_attention=True)."""
        BF, C, H, W = x.shape
        B = BF // frames
        h = x[None, :].reshape(B, frames, C, H, W).permute(0, 2, 1, 3, 4)
        h = F.group_norm(h, self.cfg.groups, self.w[n + ".norm.weight"], self.w[n + ".norm.bias"], 1e-6)
        h = h.permute(0, 3, 4, 2, 1).reshape(B * H * W, frames, C)
        h = self._lin(n + ".proj_in", h)
        inner = h.shape[-1]
        b = n + ".transformer_blocks.0"
        ln = lambda k, t: F.layer_norm(t, (inner,), self.w[f"{b}.{k}.weight"], self.w[f"{b}.{k}.bias"], 1e-5)
        y = ln("norm1", h); h = self.attention(b + ".attn1", y, y, heads) + h
        y = ln("norm2", h); h = self.attention(b + ".attn2", y, y, heads) + h
        p = self._lin(b + ".ff.net.0.proj", ln("norm3", h))
        a, gate = p.chunk(2, -1)
        h = self._lin(b + ".ff.net.2", a * F.gelu(gate)) + h
        h = self._lin(n + ".proj_out", h)
        h = h[None, None, :].reshape(B, H, W, frames, C).permute(0, 3, 4, 1, 2).reshape(BF, C, H, W)
        return h + x

    @torch.no_grad()
    def forward(self, sample, t, enc):
        """sample [B,C,F,H,W], t scalar, enc [B,77,cross] -> [B,C,F,H,W] (UNet3DConditionModel.forward)."""
        cfg, c3 = self.cfg, self.c3
        B, _, Fr, H, W = sample.shape
        sample, enc = sample.to(self.dtype), enc.to(self.dtype)
        temb = self.time_embed(torch.as_tensor(t), B).repeat_interleave(Fr, 0)
        enc = enc.repeat_interleave(Fr, 0)
        h = self._conv("conv_in", sample.permute(0, 2, 1, 3, 4).reshape(B * Fr, -1, H, W))
        h = self.temp_transformer("transformer_in", h, Fr, c3.tin_heads)
        skips = [h]
        nb = len(cfg.block_out)
        for i in range(nb):
            for j in range(cfg.layers_per_block):
                h = self.resnet(f"down_blocks.{i}.resnets.{j}", h, temb)
                h = self.temp_conv(f"down_blocks.{i}.temp_convs.{j}", h, Fr)
                if cfg.down_attn[i]:
                    h = self.transformer(f"down_blocks.{i}.attentions.{j}", h, enc, cfg.heads[i], 1)
                    h = self.temp_transformer(f"down_blocks.{i}.temp_attentions.{j}", h, Fr, cfg.heads[i])
                skips.append(h)
            if i != nb - 1:
                h = self._conv(f"down_blocks.{i}.downsamplers.0.conv", h, stride=2, pad=1)
                skips.append(h)
        h = self.resnet("mid_block.resnets.0", h, temb)
        h = self.temp_conv("mid_block.temp_convs.0", h, Fr)
        h = self.transformer("mid_block.attentions.0", h, enc, cfg.heads[-1], 1)
        h = self.temp_transformer("mid_block.temp_attentions.0", h, Fr, cfg.heads[-1])
        h = self.resnet("mid_block.resnets.1", h, temb)
        h = self.temp_conv("mid_block.temp_convs.1", h, Fr)
        rheads = list(reversed(cfg.heads))
        for i in range(nb):
            for j in range(cfg.layers_per_block + 1):
                h = self.resnet(f"up_blocks.{i}.resnets.{j}", torch.cat([h, skips.pop()], 1), temb)
                h = self.temp_conv(f"up_blocks.{i}.temp_convs.{j}", h, Fr)
                if cfg.up_attn[i]:
                    h = self.transformer(f"up_blocks.{i}.attentions.{j}", h, enc, rheads[i], 1)
                    h = self.temp_transformer(f"up_blocks.{i}.temp_attentions.{j}", h, Fr, rheads[i])
            if i != nb - 1:
                h = F.interpolate(h, size=skips[-1].shape[2:], mode="nearest")
                h = self._conv(f"up_blocks.{i}.upsamplers.0.conv", h)
        h = self._conv("conv_out", F.silu(self._gn("conv_norm_out", h)))
        return h[None, :].reshape(B, Fr, -1, H, W).permute(0, 2, 1, 3, 4).float()


@torch.no_grad()
def video_denoise_loop(unet: UNet3DOracle, sched, latents, enc_uncond_cond, guidance, steps):
    """custom_vd.py:664-697: latents [B,C,F,h,w]; CFG batch 2B; the scheduler sees frames as batch (:684-692)."""
    ts = sched.set_timesteps(steps)
    latents = latents * sched.init_noise_sigma
    for t in ts:
        e = unet.forward(torch.cat([latents] * 2), t, enc_uncond_cond)
        eu, ec = e.chunk(2)
        eps = eu + guidance * (ec - eu)
        B, C, Fr, H, W = latents.shape
        flat = lambda x: x.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W)
        latents = sched.step(flat(eps), t, flat(latents))[None, :].reshape(B, Fr, C, H, W).permute(0, 2, 1, 3, 4)
    return latents


def tensor2vid(video: torch.Tensor):
    """custom_vd.py:59-74: [B,3,F,H,W] in [-1,1] -> list of F uint8 frames [H, B*W, 3] (batch tiled horizontally)."""
    import numpy as np
    video = (video * 0.5 + 0.5).clamp(0, 1)
    i, c, f, h, w = video.shape
    images = video.permute(2, 3, 0, 4, 1).reshape(f, h, i * w, c)
    return [(im.cpu().numpy() * 255).astype("uint8") for im in images.unbind(0)]
