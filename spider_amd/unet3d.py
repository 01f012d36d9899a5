"""Native UNet3DConditionModel step (text-to-video, zeroscope / modelscope) on the HIP kernels: the network the
reference's TextToVideoSDPipeline calls at spider/models/custom_vd.py:671-676.
The layer graph restates diffusers==0.25.0 (see oracle/unet3d.py, parity unpinned upstream).

MI355X-first layout: a video batch is ONE NHWC bf16 tensor [B*F, H, W, C] (frames are images), so every spatial layer
is exactly the 2-D engine's kernel at batch B*F; the temporal layers need NO permutes or copies:
  * TemporalConvLayer: GroupNorm over (C/G, F, H, W) = the same GroupNorm kernel on the view [B, F*H*W, C]; the (3,1,1)
    Conv3d = the general implicit-GEMM conv on the view [B, F, H*W, C] with a 3x1 kernel, residual fused
  * TransformerTemporalModel: LayerNorm / projections / GEGLU act on rows in any order; the attention along the frame
    axis reads q,k,v through strided views (batch = pixel, row stride = H*W*C) and writes its output the same way
  * cross-attention K/V of the 77 text tokens are projected once per prompt for the B CFG samples and shared by all
    F frames (queries of one sample form one [F*H*W] sequence) instead of repeat_interleave'ing the text F times
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import ops
from .unet import UNetConfig, UNetEngine, _param_shapes

BF16 = torch.bfloat16


@dataclass
class UNet3DConfig:
    in_ch: int = 4
    out_ch: int = 4
    block_out: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_attn: Tuple[bool, ...] = (True, True, True, False)
    up_attn: Tuple[bool, ...] = (False, True, True, True)
    head_dim: int = 64            # config key `attention_head_dim`; heads per block = channels // head_dim
    layers_per_block: int = 2
    cross_dim: int = 1024
    groups: int = 32
    tin_heads: int = 8

    @staticmethod
    def zeroscope():
        return UNet3DConfig()

    @staticmethod
    def from_diffusers_dict(c: dict) -> "UNet3DConfig":
        ahd = c.get("attention_head_dim", 64)
        if isinstance(ahd, (list, tuple)):
            if len(set(ahd)) != 1:
                raise NotImplementedError("per-block attention_head_dim")
            ahd = ahd[0]
        return UNet3DConfig(c["in_channels"], c["out_channels"], tuple(c["block_out_channels"]),
                            tuple(t.startswith("CrossAttn") for t in c["down_block_types"]),
                            tuple(t.startswith("CrossAttn") for t in c["up_block_types"]), ahd,
                            c.get("layers_per_block", 2), c["cross_attention_dim"], c.get("norm_num_groups", 32), 8)

    def as2d(self) -> UNetConfig:
        nb = len(self.block_out)
        return UNetConfig(self.in_ch, self.out_ch, self.block_out, self.down_attn, self.up_attn, (1,) * nb,
                          tuple(c // self.head_dim for c in self.block_out), self.layers_per_block, self.cross_dim, self.groups,
                          True, 0, 0, None)


def _param_shapes3d(c: UNet3DConfig) -> dict:
    S = dict(_param_shapes(c.as2d()))
    def norm(n, ch): S[n + ".weight"] = (ch,); S[n + ".bias"] = (ch,)
    def lin(n, co, ci, bias=True):
        S[n + ".weight"] = (co, ci)
        if bias: S[n + ".bias"] = (co,)
    def temp_conv(n, ch):
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


class UNet3DEngine(UNetEngine):
    def __init__(self, cfg: UNet3DConfig, weights: Dict[str, torch.Tensor], device="cuda:0", dtype=BF16, stream32: bool = False,
                 precise: bool = False):
        """stream32: fp32 master of the residual stream beside its 16-bit shadow, as UNetEngine (the spatial resnets / transformers
        are the base class's; the temporal convs and temporal transformers below carry the master through their own residual adds).
        precise: as UNetEngine(precise=True) -- every read of the stream on its master, here also in the temporal layers."""
        self.cfg3 = cfg
        # Conv3d (3,1,1) weights [O,I,3,1,1] -> [O,I,3,1]; the base class turns 4-D conv weights into OHWI = [O,3,1,I]
        w = {n: (t[..., 0] if t.ndim == 5 else t) for n, t in weights.items()}
        super().__init__(cfg.as2d(), w, device, dtype=dtype, stream32=stream32, precise=precise)
        # Producer-side GroupNorm statistics (UNetEngine.gn_producer) are OFF for the video UNet: its large convs run on the 256^2 kernel
        # (no statistics epilogue -> the fallback pass costs what the norm's own pass costs) and its temporal norms span 16 frames
        # (720 chunks per image); measured on MI355X 50.4 ms per evaluation with them against 50.2 without. SPIDER_GN_PRODUCER_3D=1 turns
        # them on (the path stays covered by tests/test_video_engine.py).
        import os
        self.gn_producer = self.gn_producer and os.environ.get("SPIDER_GN_PRODUCER_3D", "0") == "1"
        self.gn_fuse_in = self.gn_fuse_in and self.gn_producer
        is_temporal = lambda l: ".temp_attentions." in l or l.startswith("transformer_in")
        self.cross_layers = [l for l in self.cross_layers if not is_temporal(l)]
        self.frames = 1

    @classmethod
    def random_init(cls, cfg: UNet3DConfig, device="cuda:0", seed=0, dtype=BF16, stream32: bool = False, precise: bool = False):
        gen = torch.Generator(device=device).manual_seed(seed)
        w = {}
        for n, shp in _param_shapes3d(cfg).items():
            if n.endswith(".bias"):
                t = torch.randn(shp, generator=gen, device=device) * 0.02
            elif len(shp) == 1:
                t = torch.ones(shp, device=device)
            else:
                t = torch.randn(shp, generator=gen, device=device) * (1.0 / math.sqrt(math.prod(shp[1:])))
            w[n] = t.to(BF16)
        return cls(cfg, w, device, dtype=dtype, stream32=stream32, precise=precise)

    @classmethod
    def from_pretrained(cls, path: str, device="cuda:0", dtype=BF16, stream32: bool = False, precise: bool = False):
        """diffusers layout; cerspense/zeroscope_v2_576w publishes diffusion_pytorch_model.bin only (spider_amd/checkpoint.py)."""
        from .checkpoint import load_state_dict, read_config
        cfg = UNet3DConfig.from_diffusers_dict(read_config(path))
        return cls(cfg, load_state_dict(path), device, dtype=dtype, stream32=stream32, precise=precise)

    def prepare(self, timesteps, enc, added=None, class_labels=None, frames: int = 1):
        """enc [B2, 77, cross]; the UNet input of step() is [B2*frames, h, w, C] (sample-major, frame-minor)."""
        if frames != self.frames:
            self.frames = frames
            self._graph = None
        super().prepare(timesteps, enc, added, class_labels)

    # ------------------------------------------------------------------ temporal layers
    def _temp_conv(self, n, x):
        w, F_ = self.w, self.frames
        BF, H, W_, C = x.shape
        B = BF // F_
        h = x
        hp = getattr(x, "_gnp", None)
        if hp is not None:       # statistics left by the producer of x, chunked over the same rows: re-viewed per SAMPLE (F * HW rows)
            cr = (x.shape[0] * H * W_) // (hp.t.shape[0] * hp.nchunk)
            hp = ops.GnPartial(hp.t.view(B, F_ * H * W_ // cr, hp.groups, 2), F_ * H * W_ // cr, hp.groups) if (F_ * H * W_) % cr == 0 else None
        G = self.cfg.groups if self.gn_producer else None
        for i, ci in ((1, 2), (2, 3), (3, 3), (4, 3)):
            if self.precise and i == 1:       # the stream's master, one rounding on the way out
                a = ops.groupnorm_f32in(x._s32.view(B, F_ * H * W_, C), w[f"{n}.conv{i}.0.weight"], w[f"{n}.conv{i}.0.bias"], self.cfg.groups,
                                        1e-5, True, partial=hp)
            else:
                a = ops.groupnorm(h.view(B, F_ * H * W_, C), w[f"{n}.conv{i}.0.weight"], w[f"{n}.conv{i}.0.bias"], self.cfg.groups, 1e-5, True,
                                  partial=hp)
            hp = None
            if i == 4 and self.stream32:     # x + conv4(...): the identity is added in fp32 and the new master handed on
                x32 = getattr(x, "_s32", None)
                h, h32 = ops.conv_ex(a.view(B, F_, H * W_, C), w[f"{n}.conv{i}.{ci}.weight"], bias=w[f"{n}.conv{i}.{ci}.bias"], pad=(1, 0),
                                     res=None if x32 is not None else x.view(B, F_, H * W_, C),
                                     res32=None if x32 is None else x32.view(B, F_, H * W_, C), want32=True)
                out = h.view(BF, H, W_, C)
                out._s32 = h32.view(BF, H, W_, C)
                return out
            if G and i < 4 and F_ * H * W_ <= 16384:      # conv1..3 feed the next GroupNorm of this layer: its statistics come from the conv
                h, hp = ops.conv_ex(a.view(B, F_, H * W_, C), w[f"{n}.conv{i}.{ci}.weight"], bias=w[f"{n}.conv{i}.{ci}.bias"], pad=(1, 0),
                                    gn_groups=G)
                continue
            h = ops.conv_ex(a.view(B, F_, H * W_, C), w[f"{n}.conv{i}.{ci}.weight"], bias=w[f"{n}.conv{i}.{ci}.bias"], pad=(1, 0),
                            res=x.view(B, F_, H * W_, C) if i == 4 else None)
        return h.view(BF, H, W_, C)

    def _frame_attention(self, qkv, inner, heads, B, HW):
        """qkv [B*F*HW, 3*inner] (rows ordered sample, frame, pixel) -> attention along the frame axis, same row order."""
        F_ = self.frames
        o = torch.empty(qkv.shape[0], inner, dtype=qkv.dtype, device=qkv.device)
        for b in range(B):
            qv = qkv[b * F_ * HW:(b + 1) * F_ * HW].view(F_, HW, 3 * inner).permute(1, 0, 2)     # [HW, F, 3*inner], no copy
            ov = o[b * F_ * HW:(b + 1) * F_ * HW].view(F_, HW, inner).permute(1, 0, 2)
            ops.attention(qv[..., :inner], qv[..., inner:2 * inner], qv[..., 2 * inner:], heads, out=ov)
        return o

    def _temp_transformer(self, n, x, heads):
        w, F_ = self.w, self.frames
        BF, H, W_, C = x.shape
        B, HW = BF // F_, H * W_
        s32, h32 = self.stream32, None
        P = self.precise
        geglu = "geglu_exact" if P else "geglu"
        if P:        # norm on the master, its fp32 output split hi / lo inside proj_in
            a32 = ops.groupnorm_f32in(x._s32.view(B, F_ * HW, C), w[n + ".norm.weight"], w[n + ".norm.bias"], self.cfg.groups, 1e-6, False,
                                      want16=False, want32=True)
            h, h32 = ops.gemm_a32(a32.view(BF * HW, C), w[n + ".proj_in.weight"], bias=w[n + ".proj_in.bias"], want32=True)
        else:
            a = ops.groupnorm(x.view(B, F_ * HW, C), w[n + ".norm.weight"], w[n + ".norm.bias"], self.cfg.groups, 1e-6, False)
        if P:
            pass
        elif s32:
            h, h32 = ops.gemm(a.view(BF * HW, C), w[n + ".proj_in.weight"], bias=w[n + ".proj_in.bias"], want32=True)
        else:
            h = ops.gemm(a.view(BF * HW, C), w[n + ".proj_in.weight"], bias=w[n + ".proj_in.bias"])
        rg = (lambda A_, W_w, bias, h_, h32_: ops.gemm(A_, W_w, bias=bias, res32=h32_, want32=True)) if s32 else \
             (lambda A_, W_w, bias, h_, h32_: (ops.gemm(A_, W_w, bias=bias, res=h_), None))
        inner = h.shape[-1]
        b = n + ".transformer_blocks.0"
        for at, nm in (("attn1", "norm1"), ("attn2", "norm2")):     # double_self_attention: both attend over the frames
            if self.fuse_ln and self.precise_ln:
                qkv = ops.gemm_ln_a32(h32, *self.lnx[f"{b}.{at}"])
            elif self.fuse_ln:   # norm + q/k/v projection in one launch (LayerNorm folded, see UNetEngine.__init__)
                qkv = ops.gemm_ln(h, *self.ln[f"{b}.{at}"])
            else:
                qkv = ops.gemm(ops.layernorm(h, w[f"{b}.{nm}.weight"], w[f"{b}.{nm}.bias"]), w[f"{b}.{at}.qkv"])
            o = self._frame_attention(qkv, inner, heads, B, HW)
            h, h32 = rg(o, w[f"{b}.{at}.to_out.0.weight"], w[f"{b}.{at}.to_out.0.bias"], h, h32)
        if self.fuse_ln and self.precise_ln:
            g = ops.gemm_ln_a32(h32, *self.lnx[b + ".ff"], act=geglu)
        elif self.fuse_ln:
            g = ops.gemm_ln(h, *self.ln[b + ".ff"], act=geglu)
        else:
            y = ops.layernorm(h, w[b + ".norm3.weight"], w[b + ".norm3.bias"])
            g = ops.gemm(y, w[b + ".ff.net.0.proj.weight"], bias=w[b + ".ff.net.0.proj.bias"], act=geglu)
        h, h32 = rg(g, w[b + ".ff.net.2.weight"], w[b + ".ff.net.2.bias"], h, h32)
        if P:
            out, out32 = ops.gemm_a32(h32, w[n + ".proj_out.weight"], bias=w[n + ".proj_out.bias"], res32=x._s32.view(BF * HW, C), want32=True)
            outv = out.view(BF, H, W_, C)
            outv._s32 = out32.view(BF, H, W_, C)
            return outv
        if s32:
            x32 = getattr(x, "_s32", None)
            out, out32 = ops.gemm(h, w[n + ".proj_out.weight"], bias=w[n + ".proj_out.bias"], want32=True,
                                  res=None if x32 is not None else x.view(BF * HW, C),
                                  res32=None if x32 is None else x32.view(BF * HW, C))
            outv = out.view(BF, H, W_, C)
            outv._s32 = out32.view(BF, H, W_, C)
            return outv
        out = ops.gemm(h, w[n + ".proj_out.weight"], bias=w[n + ".proj_out.bias"], res=x.view(BF * HW, C))
        return out.view(BF, H, W_, C)

    def _cross_attn(self, b, y, heads, ln_input: bool = False, y32=None):
        """y [B2*F, HW, C]: the F frames of a sample share its text K/V, so they form one query sequence of F*HW rows."""
        C = y.shape[-1]
        B2 = self.B2
        q = self._proj2(b, y, ln_input, y32).view(B2, -1, C)
        kv = self.kv[b]
        return ops.attention(q, kv[..., :C], kv[..., C:], heads).view(y.shape)

    # ------------------------------------------------------------------ forward
    def _og(self, h) -> bool:
        """ask a resnet's conv2 for the GroupNorm statistics of its output? They feed the temporal conv's first norm, whose image is
        one SAMPLE (frames x H x W rows): every block of that norm reduces all chunks of its image, which pays only up to ~256 chunks
        (16 frames of 40 x 72 would be 720: measured slower than the norm's own statistics pass)."""
        return self.gn_producer and (self.frames * h.shape[1] * h.shape[2]) // 64 <= 256

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B2*F, h, w, in_ch] bf16 NHWC -> eps [B2*F, h, w, out_ch] fp32."""
        cfg, w = self.cfg, self.w
        P = self.precise
        if P:
            h, h32 = ops.conv2d_small_cin_f32in(x if x.dtype == torch.float32 else x.float(), w["conv_in.weight"], w["conv_in.bias"], want32=True)
            h._s32 = h32
        else:
            h = ops.conv2d_small_cin(x, w["conv_in.weight"], w["conv_in.bias"])
        h = self._temp_transformer("transformer_in", h, self.cfg3.tin_heads)
        skips = [h]
        nb = len(cfg.block_out)
        for i in range(nb):
            for j in range(cfg.layers_per_block):
                h = self._resnet(f"down_blocks.{i}.resnets.{j}", h, out_gn=self._og(h))
                h = self._temp_conv(f"down_blocks.{i}.temp_convs.{j}", h)
                if cfg.down_attn[i]:
                    h = self._transformer(f"down_blocks.{i}.attentions.{j}", h, cfg.heads[i], 1)
                    h = self._temp_transformer(f"down_blocks.{i}.temp_attentions.{j}", h, cfg.heads[i])
                skips.append(h)
            if i != nb - 1:
                if P:
                    h32 = h._s32
                    h, o32 = ops.conv_a32(h32, w[f"down_blocks.{i}.downsamplers.0.conv.weight"], bias=w[f"down_blocks.{i}.downsamplers.0.conv.bias"],
                                          stride=2, pad=(1, 1), want32=True)
                    h._s32 = o32
                else:
                    h = ops.conv2d(h, w[f"down_blocks.{i}.downsamplers.0.conv.weight"], bias=w[f"down_blocks.{i}.downsamplers.0.conv.bias"],
                                   stride=2, pad=1)
                skips.append(h)
        h = self._resnet("mid_block.resnets.0", h, out_gn=self._og(h))
        h = self._temp_conv("mid_block.temp_convs.0", h)
        h = self._transformer("mid_block.attentions.0", h, cfg.heads[-1], 1)
        h = self._temp_transformer("mid_block.temp_attentions.0", h, cfg.heads[-1])
        h = self._resnet("mid_block.resnets.1", h, out_gn=self._og(h))
        h = self._temp_conv("mid_block.temp_convs.1", h)
        rheads = list(reversed(cfg.heads))
        for i in range(nb):
            for j in range(cfg.layers_per_block + 1):
                h = self._resnet(f"up_blocks.{i}.resnets.{j}", (h, skips.pop()), out_gn=self._og(h))
                h = self._temp_conv(f"up_blocks.{i}.temp_convs.{j}", h)
                if cfg.up_attn[i]:
                    h = self._transformer(f"up_blocks.{i}.attentions.{j}", h, rheads[i], 1)
                    h = self._temp_transformer(f"up_blocks.{i}.temp_attentions.{j}", h, rheads[i])
            if i != nb - 1:
                th, tw = skips[-1].shape[1], skips[-1].shape[2]
                if P:
                    h32 = h._s32
                    h, o32 = ops.conv_a32(h32, w[f"up_blocks.{i}.upsamplers.0.conv.weight"], bias=w[f"up_blocks.{i}.upsamplers.0.conv.bias"],
                                          pad=(1, 1), up_size=(th, tw), want32=True)
                    h._s32 = o32
                else:
                    h = ops.conv_ex(h, w[f"up_blocks.{i}.upsamplers.0.conv.weight"], bias=w[f"up_blocks.{i}.upsamplers.0.conv.bias"],
                                    pad=(1, 1), up_size=(th, tw))
        if P:
            a32 = ops.groupnorm_f32in(h._s32, w["conv_norm_out.weight"], w["conv_norm_out.bias"], cfg.groups, 1e-5, True, want16=False, want32=True)
            return ops.conv2d_small_cout_f32in(a32, w["conv_out.weight"], w["conv_out.bias"])
        a = self._gn("conv_norm_out", h, True)
        return ops.conv2d_small_cout(a, w["conv_out.weight"], w["conv_out.bias"], out_f32=True)


def video_denoise(unet: UNet3DEngine, scheduler, latents: torch.Tensor, enc: torch.Tensor, guidance: float, steps: int,
                  use_graph: bool = True) -> torch.Tensor:
    """The reference's video denoising loop (custom_vd.py:664-697). latents fp32 [B,C,F,h,w]; enc [2B,77,X] (uncond
    first). The reference reshapes [B,C,F,h,w] <-> [B*F,C,h,w] around every scheduler step (:684-692); here the latents
    simply LIVE as [B*F,C,h,w] (frames as batch) for the whole loop -- the scheduler update and the CFG combine are
    elementwise -- and are reshaped once at the end."""
    B, C, F_, h, w = latents.shape
    ts = scheduler.set_timesteps(steps)
    unet.prepare(ts, enc, frames=F_)
    lat = (latents.permute(0, 2, 1, 3, 4).reshape(B * F_, C, h, w) * scheduler.init_noise_sigma).contiguous()
    do_cfg = guidance > 1.0
    for i, t in enumerate(ts):
        if getattr(unet, "precise", False):
            x2 = ops.latent_to_nhwc_f32(lat, reps=2 if do_cfg else 1)
        else:
            x2 = ops.latent_to_nhwc(lat, reps=2 if do_cfg else 1, dtype=unet.dtype)
        e = unet.step(x2, i, use_graph=use_graph)
        eps = ops.cfg_combine(e, guidance) if do_cfg else ops.nhwc_to_nchw(e)
        lat = scheduler.step(eps, t, lat)
    return lat.view(B, F_, C, h, w).permute(0, 2, 1, 3, 4).contiguous()
