"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of StoryDiffusion's consistent
self-attention as Spider drives it.

Follows
  cal_attn_mask_xl           StoryDiffusion/utils/gradio_utils.py:241-287 (the torch.rand draws are explicit
                             inputs u1024/u4096 so the product and the oracle can share one random stream)
  SpatialAttnProcessor2_0    StoryDiffusion/Comic_Generation.py:74-127 (schedule), :129-196 (__call1__),
                             :198-268 (__call2__)
Module globals of the reference (total_count, attn_count, cur_step, mask1024, mask4096, sa32, sa64, write,
height, width; Comic_Generation.py:82-84) become the explicit StoryState below; coin flips
(random.random(), :98) are drawn from an injected callable.

Pinned against tests/golden/story_ref.npz (generated from the reference's own functions).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F


def keep_vectors(total_length, id_length, sa32, sa64, height, width, u1024, u4096):
    """The 1-D column keep vectors (before the per-row 'own block' override)."""
    n1, n4 = (height // 32) * (width // 32), (height // 16) * (width // 16)
    k1 = (u1024.reshape(-1) < sa32).clone()
    k4 = (u4096.reshape(-1) < sa64).clone()
    k1[id_length * n1:] = False
    k4[id_length * n4:] = False
    return k1, k4


def cal_attn_mask_xl(total_length, id_length, sa32, sa64, height, width, u1024, u4096):
    n1, n4 = (height // 32) * (width // 32), (height // 16) * (width // 16)
    k1, k4 = keep_vectors(total_length, id_length, sa32, sa64, height, width, u1024, u4096)
    b1 = k1[None].repeat(total_length, 1)
    b4 = k4[None].repeat(total_length, 1)
    for i in range(total_length):
        b1[i, i * n1:(i + 1) * n1] = True
        b4[i, i * n4:(i + 1) * n4] = True
    m1 = b1[:, None].repeat(1, n1, 1).reshape(-1, total_length * n1)
    m4 = b4[:, None].repeat(1, n4, 1).reshape(-1, total_length * n4)
    return m1, m4


@dataclass
class AttnWeights:
    to_q: torch.Tensor
    to_k: torch.Tensor
    to_v: torch.Tensor
    to_out_w: torch.Tensor
    to_out_b: Optional[torch.Tensor]
    heads: int
    rescale_output_factor: float = 1.0
    residual_connection: bool = False


def _sdpa(q, k, v, heads, mask):
    B, Lq, C = q.shape
    d = C // heads
    q = q.view(B, Lq, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, is_causal=False)
    return o.transpose(1, 2).reshape(B, Lq, C)


def call1(aw: AttnWeights, hs, enc, mask, id_length=4):
    """Consistent self-attention (Comic_Generation.py:129-196). hs [8,N,C] -> [8,N,C]."""
    tb, N, C = hs.shape
    img = tb // 2
    x = hs.view(-1, img, N, C).reshape(-1, img * N, C)
    q = F.linear(x, aw.to_q)
    e = x if enc is None else enc.view(-1, id_length + 1, N, C).reshape(-1, (id_length + 1) * N, C)
    k, v = F.linear(e, aw.to_k), F.linear(e, aw.to_v)
    o = _sdpa(q, k, v, aw.heads, mask).reshape(tb, -1, C)
    o = F.linear(o, aw.to_out_w, aw.to_out_b)
    if aw.residual_connection:
        o = o + hs
    return o / aw.rescale_output_factor


def call2(aw: AttnWeights, hs, enc, mask, id_length=4):
    """Plain attention (Comic_Generation.py:198-268)."""
    B, N, C = hs.shape
    q = F.linear(hs, aw.to_q)
    e = hs if enc is None else enc.view(-1, id_length + 1, N, C).reshape(-1, (id_length + 1) * N, C)
    k, v = F.linear(e, aw.to_k), F.linear(e, aw.to_v)
    o = F.linear(_sdpa(q, k, v, aw.heads, mask), aw.to_out_w, aw.to_out_b)
    if aw.residual_connection:
        o = o + hs
    return o / aw.rescale_output_factor


@dataclass
class StoryState:
    total_count: int
    height: int
    width: int
    id_length: int = 4
    sa32: float = 0.5
    sa64: float = 0.5
    write: bool = True
    cur_step: int = 0
    attn_count: int = 0
    mask1024: Optional[torch.Tensor] = None
    mask4096: Optional[torch.Tensor] = None
    coin: Callable[[], float] = None            # random.random stand-in
    uniforms: Callable[[int], torch.Tensor] = None  # torch.rand stand-in: n -> [n] uniforms

    @property
    def total_length(self):
        return self.id_length + 1

    def regen_masks(self):
        n1, n4 = (self.height // 32) * (self.width // 32), (self.height // 16) * (self.width // 16)
        u1 = self.uniforms(self.total_length * n1)
        u4 = self.uniforms(self.total_length * n4)
        self.mask1024, self.mask4096 = cal_attn_mask_xl(self.total_length, self.id_length, self.sa32, self.sa64,
                                                        self.height, self.width, u1, u4)


class ProcessorOracle:
    """One SpatialAttnProcessor2_0 instance (its own id_bank)."""

    def __init__(self):
        self.id_bank: Dict[int, List[torch.Tensor]] = {}

    def __call__(self, st: StoryState, aw: AttnWeights, hs: torch.Tensor):
        L = st.id_length
        enc = None
        if st.write:
            self.id_bank[st.cur_step] = [hs[:L], hs[L:]]
        else:
            enc = torch.cat((self.id_bank[st.cur_step][0], hs[:1], self.id_bank[st.cur_step][1], hs[1:]))
        if st.cur_step < 5:
            out = call2(aw, hs, enc, None, L)
        else:
            r = st.coin()
            thr = 0.3 if st.cur_step < 20 else 0.1
            if r > thr:
                use1024 = hs.shape[1] == (st.height // 32) * (st.width // 32)
                mk = st.mask1024 if use1024 else st.mask4096
                n = mk.shape[0] // st.total_length * L
                mask = mk[n:] if not st.write else mk[:n, :n]
                out = call1(aw, hs, enc, mask, L)
            else:
                out = call2(aw, hs, None, None, L)
        st.attn_count += 1
        if st.attn_count == st.total_count:
            st.attn_count = 0
            st.cur_step += 1
            st.regen_masks()
        return out
