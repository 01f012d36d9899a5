"""PyTorch-ROCm custom-op surface of the HIP kernels: `torch.ops.spider_hip.*` (north_star: "hand-written CDNA4 HIP kernels
surfaced as PyTorch-ROCm custom ops"; SURVEY.md section 8b export list).

Each op is a `torch.library.custom_op` whose implementation is the C-ABI call of spider_amd.ops (ctypes -> libspider_hip.so,
launched on torch's current stream) and whose fake (meta) implementation gives output shapes / dtypes, so the kernels are
visible to the dispatcher, to FakeTensor tracing and to `torch.compile` graphs as opaque nodes. They are what a maintainer
binds in place of the reference's module forwards:
    LlamaRMSNorm.forward / LlamaMLP.forward / LlamaAttention.forward      spider/models/modeling_llama3.py:77-82,197-199,214-313
    UNet2DConditionModel blocks called from the denoising loop             spider/models/custom_sd.py:629-647
There is no CPU implementation: the ops raise on non-HIP tensors like the rest of the product path.

    import spider_amd.torch_ops            # registers the library
    y = torch.ops.spider_hip.rmsnorm(x, w, 1e-6, None)
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

BF16 = torch.bfloat16
_lib = "spider_hip"


def _op(name, mutates=()):
    return torch.library.custom_op(f"{_lib}::{name}", mutates_args=mutates, device_types="cuda")


# ------------------------------------------------------------------------------------------ LLM path (a1 - a6)
@_op("rmsnorm")
def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = w * bf16((x + residual) * rsqrt(mean((x + residual)^2) + eps))     modeling_llama3.py:77-82"""
    return ops.rmsnorm(x.contiguous(), w, eps, res=residual)


@rmsnorm.register_fake
def _(x, w, eps, residual=None):
    return torch.empty_like(x)


@_op("rope_qk_", mutates=("q_out", "k_cache", "v_cache"))
def rope_qk_(qkv: torch.Tensor, pos: torch.Tensor, slot: torch.Tensor, cos_sin: torch.Tensor, q_out: torch.Tensor,
             k_cache: torch.Tensor, v_cache: torch.Tensor, n_q: int, n_kv: int, head_dim: int) -> None:
    """RoPE on q / k of the fused projection rows + append of k, v at `slot` (apply_rotary_pos_emb + cache update,
    modeling_llama3.py:128-183,240-313). qkv [B, S, (n_q + 2 n_kv) d]; q_out [B, S, n_q, d]; caches [B, n_kv, T, d]."""
    B, S = qkv.shape[0], qkv.shape[1]
    ops.rope_kv_append(qkv.contiguous(), pos, slot, cos_sin, q_out, k_cache, v_cache, B, S, n_q, n_kv, head_dim)


@_op("attn_decode")
def attn_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, kv_end: torch.Tensor,
                kv_beg: Optional[torch.Tensor] = None) -> torch.Tensor:
    """one query token per sequence against the KV cache (GQA, fp32 softmax); q [B, n_q, d] -> [B, n_q * d]"""
    return ops.attn_decode(q.contiguous(), k_cache, v_cache, kv_end, kv_beg=kv_beg)


@attn_decode.register_fake
def _(q, k_cache, v_cache, kv_end, kv_beg=None):
    return q.new_empty(q.shape[0], q.shape[1] * q.shape[2])


@_op("attn_prefill_causal")
def attn_prefill_causal(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, n_heads: int, n_kv_heads: int) -> torch.Tensor:
    """causal GQA attention over a prompt; q [B, S, n_heads d], k / v [B, S, n_kv_heads d] -> [B, S, n_heads d]"""
    return ops.attention(q, k, v, n_heads, n_kv_heads=n_kv_heads, causal=True)


@attn_prefill_causal.register_fake
def _(q, k, v, n_heads, n_kv_heads):
    return q.new_empty(q.shape)


@_op("swiglu")
def swiglu(gate_up: torch.Tensor) -> torch.Tensor:
    """silu(gate) * up on a fused [gate | up] projection (LlamaMLP, modeling_llama3.py:197-199)"""
    return ops.swiglu(gate_up.contiguous())


@swiglu.register_fake
def _(gate_up):
    return gate_up.new_empty(*gate_up.shape[:-1], gate_up.shape[-1] // 2)


@_op("linear_bf16")
def linear_bf16(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x @ w^T + b on the MFMA GEMM (nn.Linear layout w [N, K])"""
    return ops.gemm(x.contiguous(), w, bias=b)


@linear_bf16.register_fake
def _(x, w, b=None):
    return x.new_empty(*x.shape[:-1], w.shape[0])


@_op("lm_head_argmax")
def lm_head_argmax(h: torch.Tensor, w: torch.Tensor, norm_w: Optional[torch.Tensor] = None, eps: float = 1e-6) -> torch.Tensor:
    """final RMSNorm (optional) + lm_head + greedy argmax, ties -> lowest id; h [B <= 8, H] -> int32 [B]"""
    return ops.lm_head_argmax(w, h.contiguous(), norm_w=norm_w, eps=eps)


@lm_head_argmax.register_fake
def _(h, w, norm_w=None, eps=1e-6):
    return h.new_empty(h.shape[0], dtype=torch.int32)


# ------------------------------------------------------------------------------------------ UNet path (a9 - a12)
@_op("groupnorm_silu")
def groupnorm_silu(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, groups: int, eps: float, act: bool) -> torch.Tensor:
    """GroupNorm (+ SiLU when act) on NHWC [B, H, W, C] (ResnetBlock2D.norm1/2, Transformer2DModel.norm)"""
    return ops.groupnorm(x.contiguous(), w, b, groups, eps, act)


@groupnorm_silu.register_fake
def _(x, w, b, groups, eps, act):
    return torch.empty_like(x)


@_op("conv2d_nhwc")
def conv2d_nhwc(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], stride: int, pad: int) -> torch.Tensor:
    """implicit-GEMM conv: x [B, H, W, Cin], w [Cout, k, k, Cin] (OHWI) -> [B, Ho, Wo, Cout]"""
    return ops.conv2d(x.contiguous(), w, bias=b, stride=stride, pad=pad)


@conv2d_nhwc.register_fake
def _(x, w, b, stride, pad):
    B, H, W, _ = x.shape
    k = w.shape[1]
    return x.new_empty(B, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1, w.shape[0])


@_op("attn_self")
def attn_self(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, n_heads: int) -> torch.Tensor:
    """UNet self-attention (q, k, v may be column slices of one fused projection); [B, N, C] -> [B, N, C]"""
    return ops.attention(q, k, v, n_heads)


@attn_self.register_fake
def _(q, k, v, n_heads):
    return q.new_empty(q.shape)


@_op("attn_cross_kv77")
def attn_cross_kv77(q: torch.Tensor, kc: torch.Tensor, vc: torch.Tensor, n_heads: int) -> torch.Tensor:
    """cross-attention against the 77 projected text tokens (K / V computed once per prompt)"""
    return ops.attention(q, kc, vc, n_heads)


@attn_cross_kv77.register_fake
def _(q, kc, vc, n_heads):
    return q.new_empty(q.shape)


@_op("attn_consistent")
def attn_consistent(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, n_heads: int, keep_bits: torch.Tensor, blk: int,
                    q_off: int) -> torch.Tensor:
    """StoryDiffusion consistent self-attention: a key is visible if its keep bit is set or it lies in the query's own image
    block of `blk` tokens (SpatialAttnProcessor2_0 + cal_attn_mask_xl, Comic_Generation.py:129-196, gradio_utils.py:241-287)"""
    return ops.attention(q, k, v, n_heads, keep_bits=keep_bits, blk=blk, q_off=q_off)


@attn_consistent.register_fake
def _(q, k, v, n_heads, keep_bits, blk, q_off):
    return q.new_empty(q.shape)


@_op("geglu")
def geglu(x: torch.Tensor) -> torch.Tensor:
    """value * gelu(gate) on a fused [value | gate] projection (diffusers FeedForward GEGLU)"""
    return ops.geglu(x.contiguous())


@geglu.register_fake
def _(x):
    return x.new_empty(*x.shape[:-1], x.shape[-1] // 2)


@_op("cfg_step")
def cfg_step(eps2: torch.Tensor, latents: torch.Tensor, guidance: float, c_sample: float, c_eps: float) -> torch.Tensor:
    """classifier-free-guidance combine + a linear scheduler update (custom_sd.py:642-647):
    eps = e_u + g (e_c - e_u) from the UNet output eps2 [2B, h, w, C] fp32 NHWC, result = c_sample * latents + c_eps * eps
    (fp32 NCHW; DDIM's step is exactly this form, PNDM adds stored history terms through spider_lincomb)."""
    eps = ops.cfg_combine(eps2.contiguous(), guidance)
    return ops.lincomb([latents.contiguous(), eps], [c_sample, c_eps])


@cfg_step.register_fake
def _(eps2, latents, guidance, c_sample, c_eps):
    return torch.empty_like(latents)


OP_NAMES = ("rmsnorm", "rope_qk_", "attn_decode", "attn_prefill_causal", "swiglu", "linear_bf16", "lm_head_argmax", "groupnorm_silu",
            "conv2d_nhwc", "attn_self", "attn_cross_kv77", "attn_consistent", "geglu", "cfg_step")
