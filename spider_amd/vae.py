"""VAE decoder (AutoencoderKL.decode) on the HIP kernels: `decode_latents` of the reference
(spider/models/custom_sd.py:386-393: latents / 0.18215 -> vae.decode -> (x/2+0.5).clamp(0,1)).
NHWC bf16 activations; convs are implicit GEMMs with the nearest-2x upsample fused into the input addressing;
the mid-block attention has ONE 512-wide head, so it runs as GEMM (fp32 scores) -> row softmax -> GEMM against a
V^T that is produced directly by a GEMM (no transpose pass; the value bias is folded in after P.V because the
rows of P sum to one)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch

from . import ops

BF16 = torch.bfloat16


@dataclass
class VAEConfig:
    latent: int = 4
    out_ch: int = 3
    block_out: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    groups: int = 32
    scaling: float = 0.18215   # hard-coded in the reference (custom_sd.py:388), not read from the checkpoint

    @staticmethod
    def sd15():
        return VAEConfig()

    @staticmethod
    def audioldm():   # cvssp/audioldm-s-full-v2 vae/config.json: mel-spectrogram VAE, 8 latent channels, 1 output channel
        return VAEConfig(8, 1, (128, 256, 512), 2, 32, 0.9227)

    @staticmethod
    def from_diffusers_dict(c: dict) -> "VAEConfig":
        return VAEConfig(c.get("latent_channels", 4), c.get("out_channels", 3), tuple(c.get("block_out_channels", (128, 256, 512, 512))),
                         c.get("layers_per_block", 2), c.get("norm_num_groups", 32), c.get("scaling_factor", 0.18215))


def _shapes(c: VAEConfig) -> dict:
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
    rev, prev = list(reversed(c.block_out)), c.block_out[-1]
    for i, co in enumerate(rev):
        for j in range(c.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
        prev = co
        if i != len(rev) - 1: conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    norm("decoder.conv_norm_out", rev[-1]); conv("decoder.conv_out", c.out_ch, rev[-1], 3)
    return S


class VAEDecoderEngine:
    def __init__(self, cfg: VAEConfig, weights: Dict[str, torch.Tensor], device="cuda:0", dtype=BF16):
        """dtype: torch.bfloat16 or torch.float16 (the reference decodes in torch.float16, spider_decoder.py:109). bf16 is the
        choice for checkpoints whose config sets force_upcast (the SDXL VAE overflows IEEE half: diffusers upcasts it to fp32)."""
        assert dtype in (torch.bfloat16, torch.float16), "VAEDecoderEngine: dtype must be bfloat16 or float16"
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        BF16 = dtype
        self.w = {}
        for n, t in weights.items():
            if not (n.startswith("decoder.") or n.startswith("post_quant_conv")):
                continue
            t = t.to(self.device)
            if t.ndim == 4:
                t = t.permute(0, 2, 3, 1)
            self.w[n] = t.to(BF16).contiguous()
        # post_quant_conv (1x1, 4->4) and the 1/scaling factor commute with nothing else: fold 1/scaling into its weight
        self.w["post_quant_conv.weight_scaled"] = (self.w["post_quant_conv.weight"].float() / cfg.scaling).to(BF16).contiguous()

    @classmethod
    def random_init(cls, cfg: VAEConfig, device="cuda:0", seed=0, dtype=BF16):
        gen = torch.Generator(device=device).manual_seed(seed)
        w = {}
        for n, shp in _shapes(cfg).items():
            if n.endswith(".bias"):
                t = torch.zeros(shp, device=device)
            elif "norm" in n:
                t = torch.ones(shp, device=device)
            else:
                t = torch.randn(shp, generator=gen, device=device) / math.sqrt(math.prod(shp[1:]))
            w[n] = t.to(BF16)
        return cls(cfg, w, device, dtype=dtype)

    @classmethod
    def from_pretrained(cls, path: str, device="cuda:0", scaling=None, dtype=BF16):
        """diffusers layout: <path>/config.json + *.safetensors. `scaling` overrides the checkpoint's scaling_factor (the
        reference's SD pipeline hard-codes 0.18215, custom_sd.py:388; its AudioLDM pipeline reads the config,
        custom_ad.py:289)."""
        import json, os
        from .checkpoint import load_state_dict
        cj = os.path.join(path, "config.json")
        cj_d = json.load(open(cj)) if os.path.exists(cj) else {}
        cfg = VAEConfig.from_diffusers_dict(cj_d) if cj_d else VAEConfig.sd15()
        if cj_d.get("force_upcast", False) and dtype == torch.float16:
            dtype = torch.bfloat16      # diffusers runs such a VAE in fp32 (it overflows IEEE half): take the wide-range 16-bit format
        if scaling is not None:
            cfg.scaling = scaling
        # the decoder half only: the encoder's tensors are never read
        w = load_state_dict(path, keep=lambda k: k if k.startswith(("decoder.", "post_quant_conv.")) else None)
        return cls(cfg, w, device, dtype=dtype)

    def _gn(self, n, x, silu):
        return ops.groupnorm(x, self.w[n + ".weight"], self.w[n + ".bias"], self.cfg.groups, 1e-6, silu)

    def _resnet(self, n, x):
        w = self.w
        h = ops.conv2d(self._gn(n + ".norm1", x, True), w[n + ".conv1.weight"], bias=w[n + ".conv1.bias"])
        sc = x
        if n + ".conv_shortcut.weight" in w:
            sc = ops.conv2d(x, w[n + ".conv_shortcut.weight"], bias=w[n + ".conv_shortcut.bias"], pad=0)
        return ops.conv2d(self._gn(n + ".norm2", h, True), w[n + ".conv2.weight"], bias=w[n + ".conv2.bias"], res=sc)

    def _mid_attention(self, h):
        w, a = self.w, "decoder.mid_block.attentions.0"
        B, H, W_, C = h.shape
        N = H * W_
        x = self._gn(a + ".group_norm", h, False).view(B, N, C)
        out = torch.empty_like(x)
        N8 = (N + 7) // 8 * 8      # the P.V GEMM contracts over tokens: pad them to the 16-byte row granularity
        for b in range(B):
            xb = x[b]
            xk = xb
            if N8 != N:
                xk = torch.zeros(N8, C, dtype=x.dtype, device=x.device)
                xk[:N] = xb
            q = ops.gemm(xb, w[a + ".to_q.weight"], bias=w[a + ".to_q.bias"])
            k = ops.gemm(xk, w[a + ".to_k.weight"], bias=w[a + ".to_k.bias"])
            vT = ops.gemm(w[a + ".to_v.weight"], xk)                        # [C, N8] = Wv . x^T (bias folded below)
            s = ops.gemm(q, k, out_f32=True)                                # [N, N8] fp32 scores
            p = ops.softmax_rows(s, scale=C ** -0.5, n_valid=N, dtype=x.dtype)   # padded columns -> 0
            o = ops.gemm(p, vT, bias=w[a + ".to_v.bias"])                   # P.V + b_v (rows of P sum to 1)
            ops.gemm(o, w[a + ".to_out.0.weight"], bias=w[a + ".to_out.0.bias"], res=h.view(B, N, C)[b], out=out[b])
        return out.view(B, H, W_, C)

    @torch.no_grad()
    def decode(self, latents: torch.Tensor, to_image: bool = True, use_graph: bool = True) -> torch.Tensor:
        """latents fp32 NCHW [B,4,h,w] (scheduler space) -> image fp32 NCHW [B,3,8h,8w] in [0,1].
        to_image=False returns the raw decoder output (AudioLDM's mel spectrogram, custom_ad.py:288-291).
        One hipGraph per latent shape (the decoder is ~150 launches of 5-40 us kernels)."""
        if use_graph:
            if not hasattr(self, "_graphs"):
                from .graphs import GraphRunner
                self._graphs = {True: GraphRunner(lambda z: self._decode(z, True)), False: GraphRunner(lambda z: self._decode(z, False))}
            return self._graphs[bool(to_image)](latents.contiguous())
        return self._decode(latents, to_image)

    def _decode(self, latents: torch.Tensor, to_image: bool) -> torch.Tensor:
        c, w = self.cfg, self.w
        z = ops.latent_to_nhwc(latents.contiguous(), dtype=self.dtype)
        z = ops.conv2d_small_cin(z, w["post_quant_conv.weight_scaled"], w["post_quant_conv.bias"]) if c.latent % 8 == 0 else \
            self._post_quant(z)
        h = ops.conv2d_small_cin(z, w["decoder.conv_in.weight"], w["decoder.conv_in.bias"])
        h = self._resnet("decoder.mid_block.resnets.0", h)
        h = self._mid_attention(h)
        h = self._resnet("decoder.mid_block.resnets.1", h)
        n = len(c.block_out)
        for i in range(n):
            for j in range(c.layers_per_block + 1):
                h = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", h)
            if i != n - 1:
                h = ops.conv2d(h, w[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"], bias=w[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], ups=True)
        img = ops.conv2d_small_cout(self._gn("decoder.conv_norm_out", h, True), w["decoder.conv_out.weight"], w["decoder.conv_out.bias"])
        return ops.nhwc_to_nchw(img, 0.5, 0.5, True) if to_image else ops.nhwc_to_nchw(img, 1.0, 0.0, False)

    def _post_quant(self, z):
        """4 -> 4 channel 1x1 conv: Cout = 4 fits the small-Cout kernel only when Cin % 8 == 0, so pad Cin to 8."""
        w = self.w
        B, H, W_, C = z.shape
        zp = torch.zeros(B, H, W_, 8, dtype=z.dtype, device=z.device); zp[..., :C] = z
        wp = torch.zeros(C, 1, 1, 8, dtype=z.dtype, device=z.device); wp[..., :C] = w["post_quant_conv.weight_scaled"]
        return ops.conv2d_small_cout(zp, wp, w["post_quant_conv.bias"], out_f32=False)
