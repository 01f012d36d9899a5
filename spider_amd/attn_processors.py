"""diffusers-style attention-processor registry on top of UNetEngine's `self_attn_hook` seam.

The reference plugs its own processors into the SDXL UNet the diffusers way (StoryDiffusion/Comic_Generation.py:353-371):

    attn_procs = {}
    for name in unet.attn_processors.keys():              # "...attn1.processor" / "...attn2.processor"
        attn_procs[name] = SpatialAttnProcessor2_0(...) if name.startswith("up_blocks") and "attn1" in name else AttnProcessor()
    unet.set_attn_processor(copy.deepcopy(attn_procs))

and a processor is `proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None) -> hidden_states`
(gradio_utils.py:400-472, Comic_Generation.py:129-268). This module offers the same two entry points on the HIP engine:

* `attn_processors(unet)` -> the dict of names the reference iterates over (values: the installed processor or None = native
  kernels), `set_attn_processor(unet, procs)` installs a dict (or one processor for every self-attention).
* a user processor receives an `AttnShim` in place of diffusers' `Attention` module: `heads`, `scale`, `to_q / to_k / to_v`
  (the HIP GEMM on the engine's weights), `to_out` (identities: the engine applies the real output projection, bias and residual
  itself right after the hook, so a processor's result is the attention output), the fields processors test
  (`residual_connection`, `rescale_output_factor`, `group_norm`, `spatial_norm`, `norm_cross`) and `prepare_attention_mask`.

Scope: self-attention (attn1) processors. The cross-attention of the engine is fused with its projections and the text K / V are
folded once per prompt, so an "attn2" entry must be a default processor (class name AttnProcessor / AttnProcessor2_0) or None.
StoryDiffusion's own processor does not go through this adapter: `spider_amd.story.ConsistentSelfAttention` implements it on the
visible-key-list kernels. hipGraph capture is off while a hook is installed (UNetEngine.step)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops

_DEFAULT_NAMES = ("AttnProcessor", "AttnProcessor2_0")


class _Linear:
    """`attn.to_q(x)`: x [..., K] -> [..., N] on the HIP GEMM (weight rows [lo, hi) of the engine's fused projection)"""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None):
        self.weight, self.bias = weight, bias

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        lead = x.shape[:-1]
        y = ops.gemm(x.reshape(-1, x.shape[-1]).contiguous(), self.weight, bias=self.bias)
        return y.view(*lead, self.weight.shape[0])


class _Identity:
    def __call__(self, x):
        return x


class AttnShim:
    """what a diffusers attention processor reads from its `attn` argument"""

    def __init__(self, eng, block: str, heads: int):
        w = eng.w[block + ".attn1.qkv"]
        C = w.shape[1]
        self.heads = heads
        self.scale = (C // heads) ** -0.5
        self.to_q, self.to_k, self.to_v = _Linear(w[:C]), _Linear(w[C:2 * C]), _Linear(w[2 * C:])
        self.to_out = [_Identity(), _Identity()]      # the engine applies attn1.to_out.0 (+ bias, + residual) after the hook
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.group_norm = self.spatial_norm = self.norm_cross = None
        self.upcast_attention = self.upcast_softmax = False

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        return attention_mask

    def head_to_batch_dim(self, t: torch.Tensor) -> torch.Tensor:
        b, n, c = t.shape
        return t.view(b, n, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, n, c // self.heads)

    def batch_to_head_dim(self, t: torch.Tensor) -> torch.Tensor:
        bh, n, d = t.shape
        return t.view(bh // self.heads, self.heads, n, d).permute(0, 2, 1, 3).reshape(bh // self.heads, n, d * self.heads)

    def get_attention_scores(self, q, k, attention_mask=None):
        s = torch.baddbmm(torch.zeros(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype, device=q.device), q, k.transpose(1, 2), beta=0, alpha=self.scale)
        if attention_mask is not None:
            s = s + attention_mask
        return s.softmax(dim=-1)


def processor_names(unet) -> list:
    """the keys of diffusers' `unet.attn_processors` for this topology in ITS order: `named_children()` of UNet2DConditionModel
    registers down_blocks, up_blocks, mid_block (Comic_Generation.py:355 walks the dict in that order and counts the up-block
    processors it installs); inside a block family by block / attention / transformer-block index."""
    import re
    fam = {"down_blocks": 0, "up_blocks": 1, "mid_block": 2}

    def key(b):
        return (fam.get(b.split(".")[0], 3), [int(t) for t in re.findall(r"\d+", b)])
    blocks = sorted((k[: -len(".attn1.qkv")] for k in unet.w if k.endswith(".attn1.qkv")), key=key)
    return [f"{b}.{a}.processor" for b in blocks for a in ("attn1", "attn2")]


def attn_processors(unet) -> Dict[str, object]:
    installed = getattr(unet, "_attn_procs", {})
    return {n: installed.get(n) for n in processor_names(unet)}


class _ProcessorHook:
    """UNetEngine.self_attn_hook adapter: routes the attn1 sites that have a user processor through it"""

    def __init__(self, procs: Dict[str, object]):
        self.procs = procs            # "<block>.attn1" -> processor
        self._shims: Dict[str, AttnShim] = {}

    def wants(self, name: str) -> bool:
        return name in self.procs

    def __call__(self, eng, name: str, y: torch.Tensor, heads: int) -> torch.Tensor:
        shim = self._shims.get(name)
        if shim is None:
            shim = self._shims[name] = AttnShim(eng, name[: -len(".attn1")], heads)
        out = self.procs[name](shim, y)
        if out.shape != y.shape or out.dtype != y.dtype:
            raise ValueError(f"attention processor at {name}: returned {tuple(out.shape)} {out.dtype}, expected {tuple(y.shape)} {y.dtype}")
        return out.contiguous()


def _is_default(proc) -> bool:
    return proc is None or type(proc).__name__ in _DEFAULT_NAMES


def set_attn_processor(unet, processor) -> None:
    """`unet.set_attn_processor(dict | processor)` of diffusers. Default processors (AttnProcessor / AttnProcessor2_0 / None) keep
    the native kernels; anything else at an attn1 site is called through an AttnShim."""
    names = processor_names(unet)
    if not isinstance(processor, dict):
        processor = {n: (processor if ".attn1." in n else None) for n in names}
    unknown = [n for n in processor if n not in names]
    if unknown:
        raise ValueError(f"set_attn_processor: unknown processor names {unknown[:3]} (of {len(unknown)})")
    if len(processor) != len(names):
        raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does not match the number of "
                         f"attention layers: {len(names)}. Please make sure to pass {len(names)} processor classes.")
    custom = {}
    for n, p in processor.items():
        if _is_default(p):
            continue
        if ".attn2." in n:
            raise NotImplementedError(f"{n}: the engine's cross-attention is fused with its projections; only default processors there")
        custom[n[: -len(".processor")]] = p
    unet._attn_procs = dict(processor)
    unet.self_attn_hook = _ProcessorHook(custom) if custom else None
