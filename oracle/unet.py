"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the diffusion UNet step and the
denoising loop the reference drives.

PARITY UNPINNED at the UNet boundary: the UNet/scheduler arithmetic lives in the third-party dependency
`diffusers==0.25.0` (requirements.txt:11, StoryDiffusion/requirements.txt:5), whose source is not under
/root/reference, is not installed in this image, and for which the reference holds no test or golden vector.
This file restates diffusers 0.25.0's published algorithm (UNet2DConditionModel, ResnetBlock2D,
Transformer2DModel, BasicTransformerBlock, Attention, GEGLU, Timesteps/TimestepEmbedding, Down/Upsample2D,
PNDMScheduler(skip_prk_steps), DDIMScheduler) and anchors on the reference's own call sites:
    denoising loop      spider/models/custom_sd.py:627-652  (CFG concat :631, scale_model_input :632,
                        unet(...) :634-639, guidance :642-644, scheduler.step :647)
    prepare_latents     spider/models/custom_sd.py:459-474
    decode_latents      spider/models/custom_sd.py:386-393  (1/0.18215, (x/2+0.5).clamp(0,1))
    SDXL + FreeU + DDIM StoryDiffusion/Comic_Generation.py:313-317,440
Weight names are diffusers' state-dict names, so a real checkpoint can be loaded into either side.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class UNetCfg:
    in_ch: int = 4
    out_ch: int = 4
    block_out: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_attn: Tuple[bool, ...] = (True, True, True, False)   # CrossAttnDownBlock2D vs DownBlock2D
    up_attn: Tuple[bool, ...] = (False, True, True, True)     # UpBlock2D vs CrossAttnUpBlock2D
    depth: Tuple[int, ...] = (1, 1, 1, 1)                      # transformer_layers_per_block (per down block)
    heads: Tuple[int, ...] = (8, 8, 8, 8)                      # attention heads per down block
    layers_per_block: int = 2
    cross_dim: int = 768
    groups: int = 32
    linear_proj: bool = False                                  # use_linear_projection (SDXL: True)
    addition_time_dim: int = 0                                 # SDXL: 256 (text_time addition embedding)
    addition_in: int = 0                                       # SDXL: 2816
    mid_depth: Optional[int] = None
    # AudioLDM (custom_ad.py:575-581 calls unet(x, t, encoder_hidden_states=None, class_labels=prompt_embeds)):
    class_in: int = 0                                          # class_embed_type="simple_projection": Linear(class_in, T)
    class_concat: bool = False                                 # class_embeddings_concat: emb = cat([temb, class_emb])
    cross_dims: Optional[Tuple[int, ...]] = None               # per-down-block cross_attention_dim (None: cross_dim everywhere)

    @staticmethod
    def sd15():
        return UNetCfg()

    @staticmethod
    def audioldm():   # cvssp/audioldm-s-full-v2 unet/config.json (checkpoint-side values, SURVEY 8 marks them with a dagger)
        return UNetCfg(8, 8, (128, 256, 384, 640), (False, True, True, True), (True, True, True, False), (1, 1, 1, 1),
                       (8, 8, 8, 8), 2, 0, 32, False, 0, 0, None, 512, True, (128, 256, 384, 640))

    @staticmethod
    def audioldm_l():   # cvssp/audioldm-l-full unet/config.json: the decoder the reference configures (train_configs/spider_decoder_cfg.py:37)
        return UNetCfg(8, 8, (256, 512, 768, 1280), (False, True, True, True), (True, True, True, False), (1, 1, 1, 1),
                     (8, 8, 8, 8), 2, 0, 32, False, 0, 0, None, 512, True, (256, 512, 768, 1280))

    @staticmethod
    def tiny_audio():
        return UNetCfg(8, 8, (64, 128, 128), (False, True, True), (True, True, False), (1, 1, 1), (2, 4, 4), 2, 0, 32,
                       False, 0, 0, None, 48, True, (64, 128, 128))

    def cross_dim_of(self, down_idx: int) -> int:
        return self.cross_dims[down_idx] if self.cross_dims is not None else self.cross_dim

    @property
    def temb_in(self):   # width of the embedding each ResnetBlock2D.time_emb_proj consumes
        return self.temb_dim * (2 if (self.class_in and self.class_concat) else 1)

    @staticmethod
    def sdxl():
        return UNetCfg(4, 4, (320, 640, 1280), (False, True, True), (True, True, False), (1, 2, 10), (5, 10, 20), 2,
                       2048, 32, True, 256, 2816, 10)

    @staticmethod
    def tiny8():     # SD-v1.5-like with 8 attention heads per level (the head count of every SD-v1.5 transformer block)
        return UNetCfg(4, 4, (64, 128, 128), (True, True, False), (False, True, True), (1, 1, 1), (8, 8, 8), 2, 64, 32, False, 0, 0, None)

    @staticmethod
    def tiny(sdxl_like=False):
        if sdxl_like:
            return UNetCfg(4, 4, (64, 128, 128), (False, True, True), (True, True, False), (1, 2, 2), (1, 2, 2), 2, 64,
                           32, True, 32, 64 + 6 * 32, 2)
        return UNetCfg(4, 4, (64, 128, 128), (True, True, False), (False, True, True), (1, 1, 1), (2, 2, 2), 2, 64, 32,
                       False, 0, 0, None)

    @property
    def temb_dim(self):
        return self.block_out[0] * 4


def timestep_embedding(t: torch.Tensor, dim: int, flip_sin_to_cos=True, shift=0.0) -> torch.Tensor:
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / (half - shift)
    emb = t.float()[:, None] * torch.exp(exponent)[None]
    emb = torch.cat([emb.sin(), emb.cos()], -1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], -1)
    return emb


def unet_param_shapes(cfg: UNetCfg) -> dict:
    """name -> shape for every parameter, in diffusers naming (conv weights OIHW)."""
    S = {}
    T = cfg.temb_dim
    c0 = cfg.block_out[0]

    def conv(n, co, ci, k): S[n + ".weight"] = (co, ci, k, k); S[n + ".bias"] = (co,)
    def lin(n, co, ci, bias=True):
        S[n + ".weight"] = (co, ci)
        if bias: S[n + ".bias"] = (co,)
    def norm(n, c): S[n + ".weight"] = (c,); S[n + ".bias"] = (c,)

    def resnet(n, ci, co):
        norm(n + ".norm1", ci); conv(n + ".conv1", co, ci, 3); lin(n + ".time_emb_proj", co, cfg.temb_in)
        norm(n + ".norm2", co); conv(n + ".conv2", co, co, 3)
        if ci != co: conv(n + ".conv_shortcut", co, ci, 1)

    def transformer(n, c, depth, xd):
        norm(n + ".norm", c)
        if cfg.linear_proj: lin(n + ".proj_in", c, c); lin(n + ".proj_out", c, c)
        else: conv(n + ".proj_in", c, c, 1); conv(n + ".proj_out", c, c, 1)
        for d in range(depth):
            b = f"{n}.transformer_blocks.{d}"
            norm(b + ".norm1", c); norm(b + ".norm2", c); norm(b + ".norm3", c)
            for a, kd in (("attn1", c), ("attn2", xd)):
                lin(f"{b}.{a}.to_q", c, c, False); lin(f"{b}.{a}.to_k", c, kd, False); lin(f"{b}.{a}.to_v", c, kd, False)
                lin(f"{b}.{a}.to_out.0", c, c)
            lin(b + ".ff.net.0.proj", 8 * c, c); lin(b + ".ff.net.2", c, 4 * c)

    conv("conv_in", c0, cfg.in_ch, 3)
    lin("time_embedding.linear_1", T, c0); lin("time_embedding.linear_2", T, T)
    if cfg.addition_in:
        lin("add_embedding.linear_1", T, cfg.addition_in); lin("add_embedding.linear_2", T, T)
    if cfg.class_in:
        lin("class_embedding", T, cfg.class_in)
    nb = len(cfg.block_out)
    ch = c0
    for i, co in enumerate(cfg.block_out):
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{i}.resnets.{j}", ch if j == 0 else co, co)
            if cfg.down_attn[i]: transformer(f"down_blocks.{i}.attentions.{j}", co, cfg.depth[i], cfg.cross_dim_of(i))
        ch = co
        if i != nb - 1: conv(f"down_blocks.{i}.downsamplers.0.conv", co, co, 3)
    cm = cfg.block_out[-1]
    resnet("mid_block.resnets.0", cm, cm)
    transformer("mid_block.attentions.0", cm, cfg.mid_depth if cfg.mid_depth is not None else cfg.depth[-1], cfg.cross_dim_of(nb - 1))
    resnet("mid_block.resnets.1", cm, cm)
    rev = list(reversed(cfg.block_out))
    rdepth = list(reversed(cfg.depth))
    prev = rev[0]
    for i, co in enumerate(rev):
        cin_skip = rev[min(i + 1, nb - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = cin_skip if j == cfg.layers_per_block else co
            rin = prev if j == 0 else co
            resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, co)
            if cfg.up_attn[i]: transformer(f"up_blocks.{i}.attentions.{j}", co, rdepth[i], cfg.cross_dim_of(nb - 1 - i))
        prev = co
        if i != nb - 1: conv(f"up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    norm("conv_norm_out", c0); conv("conv_out", cfg.out_ch, c0, 3)
    return S


def random_unet_weights(cfg: UNetCfg, seed=0, bf16_round=True) -> dict:
    g = torch.Generator().manual_seed(seed)
    w = {}
    for n, shp in unet_param_shapes(cfg).items():
        if n.endswith(".bias"):
            t = torch.randn(shp, generator=g) * 0.05
        elif "norm" in n.split(".")[-2] and n.endswith(".weight"):
            t = 1.0 + torch.randn(shp, generator=g) * 0.1
        else:
            fan_in = math.prod(shp[1:])
            t = torch.randn(shp, generator=g) * (1.0 / math.sqrt(fan_in))
        w[n] = t.bfloat16().float() if bf16_round else t
    return w


class UNetOracle:
    def __init__(self, cfg: UNetCfg, weights: dict, dtype=torch.float32):
        """dtype=torch.float32: the fp32 restatement (the parity target). dtype=torch.bfloat16 runs the SAME graph
        with torch's CPU bf16 kernels, i.e. what the reference's module would compute in bf16 (every op output
        rounded to bf16): used to separate 'bf16 storage' error from implementation error."""
        self.cfg = cfg
        self.dtype = dtype
        self.w = {k: v.to(dtype) for k, v in weights.items()}
        self.attn_hook = None   # optional: fn(name, self_attn_ctx) -> tensor, used by the StoryDiffusion oracle
        self.freeu = None       # optional (s1, s2, b1, b2)

    def _w(self, n):
        return self.w[n]

    def _gn(self, n, x, eps=1e-5):
        return F.group_norm(x, self.cfg.groups, self.w[n + ".weight"], self.w[n + ".bias"], eps)

    def _conv(self, n, x, stride=1, pad=1):
        return F.conv2d(x, self.w[n + ".weight"], self.w[n + ".bias"], stride=stride, padding=pad)

    def _lin(self, n, x):
        return F.linear(x, self.w[n + ".weight"], self.w.get(n + ".bias"))

    def resnet(self, n, x, temb):
        h = self._conv(n + ".conv1", F.silu(self._gn(n + ".norm1", x)))
        h = h + self._lin(n + ".time_emb_proj", F.silu(temb))[:, :, None, None]
        h = self._conv(n + ".conv2", F.silu(self._gn(n + ".norm2", h)))
        if n + ".conv_shortcut.weight" in self.w:
            x = self._conv(n + ".conv_shortcut", x, pad=0)
        return x + h

    def attention(self, n, x, ctx, heads):
        q = self._lin(n + ".to_q", x)
        k = self._lin(n + ".to_k", ctx)
        v = self._lin(n + ".to_v", ctx)
        B, L, C = q.shape
        d = C // heads
        sh = lambda t: t.view(B, -1, heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(sh(q), sh(k), sh(v)).transpose(1, 2).reshape(B, L, C)
        return self._lin(n + ".to_out.0", o)

    def transformer(self, n, x, enc, heads, depth):
        B, C, H, W = x.shape
        res = x
        h = self._gn(n + ".norm", x, eps=1e-6)
        if self.cfg.linear_proj:
            h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
            h = self._lin(n + ".proj_in", h)
        else:
            h = self._conv(n + ".proj_in", h, pad=0).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for d in range(depth):
            b = f"{n}.transformer_blocks.{d}"
            y = F.layer_norm(h, (C,), self.w[b + ".norm1.weight"], self.w[b + ".norm1.bias"], 1e-5)
            if self.attn_hook is not None and self.attn_hook.wants(b + ".attn1"):
                h = self.attn_hook(self, b + ".attn1", y, heads) + h
            else:
                h = self.attention(b + ".attn1", y, y, heads) + h
            y = F.layer_norm(h, (C,), self.w[b + ".norm2.weight"], self.w[b + ".norm2.bias"], 1e-5)
            # encoder_hidden_states=None (AudioLDM): diffusers' Attention falls back to its own input as the K/V source
            h = self.attention(b + ".attn2", y, enc if enc is not None else y, heads) + h
            y = F.layer_norm(h, (C,), self.w[b + ".norm3.weight"], self.w[b + ".norm3.bias"], 1e-5)
            p = self._lin(b + ".ff.net.0.proj", y)
            a, gate = p.chunk(2, -1)
            h = self._lin(b + ".ff.net.2", a * F.gelu(gate)) + h
        if self.cfg.linear_proj:
            h = self._lin(n + ".proj_out", h).reshape(B, H, W, C).permute(0, 3, 1, 2)
        else:
            h = self._conv(n + ".proj_out", h.reshape(B, H, W, C).permute(0, 3, 1, 2), pad=0)
        return h + res

    def time_embed(self, t: torch.Tensor, B: int, added: Optional[dict] = None, class_labels=None):
        cfg = self.cfg
        te = timestep_embedding(t.expand(B) if t.ndim == 0 else t, cfg.block_out[0]).to(self.dtype)
        emb = self._lin("time_embedding.linear_2", F.silu(self._lin("time_embedding.linear_1", te)))
        if cfg.addition_in:
            tid = timestep_embedding(added["time_ids"].flatten(), cfg.addition_time_dim).reshape(B, -1)
            add = torch.cat([added["text_embeds"].float(), tid], -1).to(self.dtype)
            emb = emb + self._lin("add_embedding.linear_2", F.silu(self._lin("add_embedding.linear_1", add)))
        if cfg.class_in:   # class_embed_type="simple_projection" (+ class_embeddings_concat)
            ce = self._lin("class_embedding", class_labels.to(self.dtype))
            emb = torch.cat([emb, ce], -1) if cfg.class_concat else emb + ce
        return emb

    @torch.no_grad()
    def forward(self, sample, t, enc, added: Optional[dict] = None, class_labels=None):
        """sample [B,4,h,w] fp32, t scalar tensor, enc [B,77,cross] (None: AudioLDM, conditioning comes in through
        class_labels [B, class_in]) -> [B,4,h,w]. Latent sizes that are not a multiple of 2^(levels-1) follow diffusers'
        forward_upsample_size rule: every upsampler interpolates to the size of the next skip connection."""
        cfg = self.cfg
        B = sample.shape[0]
        sample = sample.to(self.dtype)
        enc = enc.to(self.dtype) if enc is not None else None
        temb = self.time_embed(torch.as_tensor(t), B, added, class_labels)
        h = self._conv("conv_in", sample)
        skips = [h]
        nb = len(cfg.block_out)
        for i in range(nb):
            for j in range(cfg.layers_per_block):
                h = self.resnet(f"down_blocks.{i}.resnets.{j}", h, temb)
                if cfg.down_attn[i]:
                    h = self.transformer(f"down_blocks.{i}.attentions.{j}", h, enc, cfg.heads[i], cfg.depth[i])
                skips.append(h)
            if i != nb - 1:
                h = self._conv(f"down_blocks.{i}.downsamplers.0.conv", h, stride=2, pad=1)
                skips.append(h)
        h = self.resnet("mid_block.resnets.0", h, temb)
        h = self.transformer("mid_block.attentions.0", h, enc, cfg.heads[-1],
                             cfg.mid_depth if cfg.mid_depth is not None else cfg.depth[-1])
        h = self.resnet("mid_block.resnets.1", h, temb)
        rheads, rdepth = list(reversed(cfg.heads)), list(reversed(cfg.depth))
        for i in range(nb):
            for j in range(cfg.layers_per_block + 1):
                skip = skips.pop()
                hh = h
                if self.freeu is not None:
                    hh, skip = apply_freeu(i, hh, skip, *self.freeu)
                h = self.resnet(f"up_blocks.{i}.resnets.{j}", torch.cat([hh, skip], 1), temb)
                if cfg.up_attn[i]:
                    h = self.transformer(f"up_blocks.{i}.attentions.{j}", h, enc, rheads[i], rdepth[i])
            if i != nb - 1:
                h = F.interpolate(h, size=skips[-1].shape[2:], mode="nearest")   # == scale_factor 2 on even sizes
                h = self._conv(f"up_blocks.{i}.upsamplers.0.conv", h)
        h = F.silu(self._gn("conv_norm_out", h))
        return self._conv("conv_out", h).float()


def fourier_filter(x, threshold, scale):
    """diffusers.utils.torch_utils.fourier_filter (FreeU): scale the low-frequency box of the 2-D spectrum."""
    B, C, H, W = x.shape
    xf = torch.fft.fftshift(torch.fft.fftn(x.float(), dim=(-2, -1)), dim=(-2, -1))
    mask = torch.ones((B, C, H, W))
    cr, cc = H // 2, W // 2
    mask[..., cr - threshold:cr + threshold, cc - threshold:cc + threshold] = scale
    xf = torch.fft.ifftshift(xf * mask, dim=(-2, -1))
    return torch.fft.ifftn(xf, dim=(-2, -1)).real


def apply_freeu(res_idx, hidden, skip, s1, s2, b1, b2):
    """diffusers.utils.torch_utils.apply_freeu as enabled at Comic_Generation.py:315 (s1=.6, s2=.4, b1=1.1, b2=1.2)."""
    if res_idx == 0:
        n = hidden.shape[1] // 2
        hidden = torch.cat([hidden[:, :n] * b1, hidden[:, n:]], 1)
        skip = fourier_filter(skip, 1, s1)
    if res_idx == 1:
        n = hidden.shape[1] // 2
        hidden = torch.cat([hidden[:, :n] * b2, hidden[:, n:]], 1)
        skip = fourier_filter(skip, 1, s2)
    return hidden, skip


# ------------------------------------------------------------------------------------------ schedulers
def alphas_cumprod_scaled_linear(beta_start=0.00085, beta_end=0.012, n=1000) -> torch.Tensor:
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, 0)


class PNDMOracle:
    """PNDMScheduler(skip_prk_steps=True, steps_offset=1, set_alpha_to_one=False): SD-v1.5's scheduler config.
    40 inference steps -> 41 UNet calls."""

    def __init__(self, n_train=1000, steps_offset=1):
        self.ac = alphas_cumprod_scaled_linear(n=n_train)
        self.final_alpha = self.ac[0]
        self.n_train, self.offset = n_train, steps_offset
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        self.n = n
        ratio = self.n_train // n
        ts = (torch.arange(0, n) * ratio).round().long() + self.offset
        plms = torch.cat([ts[:-1], ts[-2:-1], ts[-1:]]).flip(0)
        self.timesteps = plms
        self.ets, self.counter, self.cur_sample = [], 0, None
        return plms

    def coeffs(self, t, prev_t):
        a_t = self.ac[t]
        a_p = self.ac[prev_t] if prev_t >= 0 else self.final_alpha
        b_t, b_p = 1 - a_t, 1 - a_p
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        return float(sample_coeff), float((a_p - a_t) / denom)

    def step(self, eps, t, sample):
        t = int(t)
        ratio = self.n_train // self.n
        prev_t = t - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(eps)
        else:
            prev_t = t
            t = t + ratio
        if len(self.ets) == 1 and self.counter == 0:
            m = eps
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            m = (eps + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            m = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            m = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            m = (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4]) / 24
        cs, cm = self.coeffs(t, prev_t)
        self.counter += 1
        return cs * sample - cm * m


class DDIMOracle:
    """DDIMScheduler (eta = 0, leading spacing, steps_offset=1, clip_sample=False) as configured from the SDXL
    checkpoint at Comic_Generation.py:316-317."""

    def __init__(self, n_train=1000, steps_offset=1):
        self.ac = alphas_cumprod_scaled_linear(n=n_train)
        self.final_alpha = self.ac[0]
        self.n_train, self.offset = n_train, steps_offset
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        self.n = n
        ratio = self.n_train // n
        self.timesteps = ((torch.arange(0, n) * ratio).round().flip(0).long() + self.offset)
        return self.timesteps

    def coeffs(self, t):
        prev_t = int(t) - self.n_train // self.n
        a_t = self.ac[int(t)]
        a_p = self.ac[prev_t] if prev_t >= 0 else self.final_alpha
        # x_prev = sqrt(a_p) * (x - sqrt(1-a_t) eps) / sqrt(a_t) + sqrt(1-a_p) eps
        cx = float((a_p / a_t) ** 0.5)
        ce = float((1 - a_p) ** 0.5 - (a_p / a_t) ** 0.5 * (1 - a_t) ** 0.5)
        return cx, ce

    def step(self, eps, t, sample):
        cx, ce = self.coeffs(t)
        return cx * sample + ce * eps


@torch.no_grad()
def denoise_loop(unet: UNetOracle, sched, latents, enc_uncond_cond, guidance, steps, added=None, class_labels=None):
    """custom_sd.py:627-652 with CFG batch = 2x latents. enc_uncond_cond [2B,77,C] (uncond first). AudioLDM
    (custom_ad.py:568-594): enc_uncond_cond=None and class_labels [2B, class_in] (uncond first)."""
    ts = sched.set_timesteps(steps)
    latents = latents * sched.init_noise_sigma
    for t in ts:
        x2 = torch.cat([latents] * 2)
        e = unet.forward(x2, t, enc_uncond_cond, added, class_labels)
        eu, ec = e.chunk(2)
        eps = eu + guidance * (ec - eu)
        latents = sched.step(eps, t, latents)
    return latents
