"""Torch-tensor front end of the C ABI (spider_amd/lib.py -> libspider_hip.so).

PyTorch is used here only for device memory and streams: every function checks shapes/dtypes on the
host, passes raw device pointers + sizes to the HIP library and enqueues on torch's current stream
(so the calls can be captured by ``torch.cuda.graph``). Nothing falls back to torch math.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import lib as _lib

BF16 = torch.bfloat16
ACT = {None: 0, "none": 0, "silu": 1, "gelu": 2, "quick_gelu": 3, "geglu": 4, "leaky_relu": 5, "relu": 6, "tanh": 7, "swiglu": 8,
       "geglu_exact": 9}       # 9: GEGLU with ONE final rounding (precise mode; 4 rounds value, gate and gelu(gate) like an f16 module)
GLU_ACTS = ("geglu", "swiglu", "geglu_exact")


def _stream() -> int:
    """torch's current stream on the CURRENT device: the kernels are launched on that device, so every tensor handed to
    the library must live there (`_chk` enforces it; engines on another GPU run under `torch.cuda.device(...)`)."""
    return torch.cuda.current_stream().cuda_stream


_WS = {}
WS_BYTES = 64 << 20
WS_TAIL = 4096        # zero-initialised bytes behind the workspace: arrival counters of the streaming conv's in-launch combine


import threading as _threading


class _ScopeStack(_threading.local):       # per host thread: two threads may enqueue on two streams at once (bench.py)
    def __init__(self):
        self.names = ["default"]


_WS_SCOPE = _ScopeStack()


class workspace_scope:
    """`with ops.workspace_scope("llm"): ...` -- calls (and graph captures) inside the block use a split-K workspace of their own.
    Reuse of a workspace is stream-ordered, so work that runs on two streams AT ONCE must not share one: bench.py's two-stream
    schedule puts the LLM pass and the diffusion decoder in different scopes. A captured graph keeps the
    workspace of the scope it was captured in. The same holds for the WS_TAIL bytes behind a workspace (arrival counters of the
    streaming conv's in-launch split-K combine: zero whenever no launch of that scope is in flight; two launches that overlapped on
    one workspace would mix their tickets and their partial slabs)."""

    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        _WS_SCOPE.names.append(self.name)
        return self

    def __exit__(self, *exc):
        _WS_SCOPE.names.pop()
        return False


def _workspace(device) -> torch.Tensor:
    """Per-(device, scope) fp32 split-K workspace (stream-ordered reuse; allocated on first use, outside any graph capture:
    engines run an eager warm-up pass before they capture)."""
    key = (device.type, device.index, _WS_SCOPE.names[-1])
    if key not in _WS:
        _WS[key] = torch.zeros((WS_BYTES + WS_TAIL) // 4, dtype=torch.float32, device=device)
    return _WS[key]


F16 = torch.float16


def _h16(t: torch.Tensor):
    """(dtype, entry-point suffix) of a diffusion-side operand: the UNet / VAE / text-encoder operators exist as bf16 and as
    IEEE-half (f16, the reference's torch_dtype) instantiations; every 16-bit operand of one call must share the dtype."""
    if t.dtype == BF16:
        return BF16, "bf16"
    if t.dtype == F16:
        return F16, "f16"
    raise ValueError(f"expected a bfloat16 or float16 tensor, got {t.dtype}")


# ---- tile-major weight copies for the weight-streaming GEMMs / convs (include/spider_hip.h: w_tiled) ----
# An engine marks its long-lived weight tensors with mark_weight(); a GEMM / conv call with at most WTILED_MAX_M output rows then
# runs on the tile-major copy [ceil(N/64)][ceil(K/64)][64][64] of that weight (one contiguous 8 KiB read per K tile of 64 rows
# instead of 64 strided 128-byte pieces), built on first use -- outside graph capture: engines run one eager warm-up pass before
# they capture -- and kept for the life of the tensor object. Unmarked tensors (activations passed as the W operand, slices)
# always take the row-major path, so a recycled allocation can never meet a stale copy.
import os as _os

WTILED_MAX_M = int(_os.environ.get("SPIDER_WTILED_MAX_M", "512"))


def mark_weight(t: torch.Tensor) -> torch.Tensor:
    t._spider_weight = True
    return t


def tile_weight64(W2: torch.Tensor) -> torch.Tensor:
    """[N, K] 16-bit -> tile-major [ceil(N/64), ceil(K/64), 64, 64], zero-padded (pure data movement)."""
    N, K = W2.shape
    Np, Kp = (N + 63) // 64 * 64, (K + 63) // 64 * 64
    if (Np, Kp) != (N, K):
        Wp = torch.zeros(Np, Kp, dtype=W2.dtype, device=W2.device)
        Wp[:N, :K] = W2
        W2 = Wp
    return W2.view(Np // 64, 64, Kp // 64, 64).permute(0, 2, 1, 3).contiguous()


def _tiled(W: torch.Tensor, M: int):
    """the tile-major copy of a marked weight when the problem is weight-streaming (M <= WTILED_MAX_M), else None"""
    if M > WTILED_MAX_M or not getattr(W, "_spider_weight", False):
        return None
    t = getattr(W, "_spider_tiled", None)
    tag = (W._version, W.data_ptr())           # an in-place update of the weight (copy_, merge) invalidates the copy
    if t is None or getattr(W, "_spider_tiled_tag", None) != tag:
        if torch.cuda.is_current_stream_capturing():
            if t is not None:
                raise RuntimeError("a marked weight was modified in place and is first used again under stream capture: call "
                                   "ops.prebuild_tiled() (or run one eager pass) before capturing")
            return None
        t = tile_weight64(W.reshape(W.shape[0], -1))
        W._spider_tiled, W._spider_tiled_tag = t, tag
    return t


def prebuild_tiled(weights, max_rows: int = 64 << 20) -> int:
    """Build the tile-major copies of every marked weight in `weights` now (engine constructors: no lazy first-use build, nothing
    left for a stream capture to miss). Returns the bytes the copies occupy (INTEGRATION.md states them per model)."""
    n = 0
    for W in weights:
        if getattr(W, "_spider_weight", False) and W.ndim >= 2 and W.numel() <= max_rows:
            t = _tiled(W, 1)
            n += 0 if t is None else t.numel() * t.element_size()
    return n


# ---- fragment-major conv weights for the weight-stationary streaming kernel (csrc/gemm.hip: wstream_kernel, w_tiled == 2) ----
WS_ENABLE = _os.environ.get("SPIDER_WS", "1") != "0"      # tuning aid: 0 = the tile kernels serve every conv
WS_INLAUNCH = _os.environ.get("SPIDER_WS_INLAUNCH", "1") != "0"
WS_MAX_M = int(_os.environ.get("SPIDER_WS_MAX_M", "128"))   # output pixels up to which the streaming kernel is used (it exists up to 512)


def set_ws_inlaunch(on: bool) -> bool:
    """Switch the split-K combine form of the streaming conv at run time (library + this module's mirror of it, which decides whether
    the producer-statistics entry point is used): True = inside the launch, False = partial slabs + reduce kernel. -> previous form.
    Both forms sum the slabs in split order (bit-identical outputs); tests compare them in one process."""
    global WS_INLAUNCH
    prev = WS_INLAUNCH
    _lib.load().spider_set_ws_inlaunch(int(bool(on)))
    WS_INLAUNCH = bool(on)
    return prev


def repack_fm_conv(W: torch.Tensor) -> torch.Tensor:
    """W [Cout, 3, 3, Cin] 16-bit (OHWI, Cin % 32 == 0) -> [ceil(Cout / 32) * 2, (Cin / 32) * 9, 64, 8]: per group of 16 output
    channels and per (32-channel block cb, tap) one 1 KiB piece whose lane 16 g + r holds W[16 rg + r, tap, 32 cb + 8 g : + 8] --
    the A fragment of mfma_f32_16x16x32; the 9 taps of a channel block are contiguous. Output channels are zero-padded to a
    multiple of 32 (one 2-tile strip per block). Pure data movement, once per weight."""
    N, kh, kw, Cin = W.shape
    assert kh == 3 and kw == 3 and Cin % 32 == 0
    Np = (N + 31) // 32 * 32
    if Np != N:
        W = torch.cat([W, torch.zeros(Np - N, kh, kw, Cin, dtype=W.dtype, device=W.device)], 0)
    # [rg, r, tap, cb, g, e] -> [rg, cb, tap, g, r, e]
    return W.reshape(Np // 16, 16, 9, Cin // 32, 4, 8).permute(0, 3, 2, 4, 1, 5).contiguous().view(Np // 16, (Cin // 32) * 9, 64, 8)


def _ws_eligible(B, Hin, Win, Cin, Cout, kh, kw, stride, pad, dil, up_size, Ho, Wo, act) -> bool:
    """mirror of csrc/gemm.hip:ws_eligible -- 3 x 3 / stride 1 / pad 1 convs with <= 512 output pixels (and an input that fits the
    LDS slab of the kernel's row-group form)"""
    if not WS_ENABLE or (kh, kw) != (3, 3) or stride != 1 or dil != 1 or tuple(pad) != (1, 1) or Cin % 32 or Cout % 4 or act:
        return False
    M, rows = B * Ho * Wo, B * Hin * Win
    if M > WS_MAX_M:
        return False
    if M <= 128:
        return rows <= 128 and up_size is None
    return M <= 512 and rows <= 512


def _wsfm(W: torch.Tensor):
    """the fragment-major copy of a marked conv weight (built on first use, outside stream capture), or None"""
    if not getattr(W, "_spider_weight", False):
        return None
    t = getattr(W, "_spider_fm", None)
    tag = (W._version, W.data_ptr())
    if t is None or getattr(W, "_spider_fm_tag", None) != tag:
        if torch.cuda.is_current_stream_capturing():
            if t is not None:
                raise RuntimeError("a marked weight was modified in place and is first used again under stream capture")
            return None
        t = repack_fm_conv(W)
        W._spider_fm, W._spider_fm_tag = t, tag
    return t


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, name: str, contiguous: bool = True):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA/HIP tensor (the HIP path has no CPU fallback)")
    if t.device.index != torch.cuda.current_device():
        raise ValueError(f"{name}: tensor lives on {t.device} but the current device is cuda:{torch.cuda.current_device()} "
                         "(kernels launch on the current device: wrap the call in torch.cuda.device(...))")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")


# --------------------------------------------------------------------------- LLM decode
def embed(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """Row gather of 16-bit rows (a copy of bit patterns: one kernel serves bf16 and f16 tables)."""
    dt, _ = _h16(table)
    _chk(table, dt, "table"); _chk(ids, torch.int32, "ids")
    V, H = table.shape
    out = torch.empty(*ids.shape, H, dtype=dt, device=table.device)
    _lib.call("spider_embed_bf16", _p(table), _p(ids), _p(out), ids.numel(), H, V, _stream())
    return out


def rmsnorm(x, w, eps, res=None, res_out=None, out=None):
    _chk(x, BF16, "x"); _chk(w, BF16, "w")
    H = x.shape[-1]
    rows = x.numel() // H
    if out is None:
        out = torch.empty_like(x)
    if res is not None:
        _chk(res, BF16, "res")
    _lib.call("spider_rmsnorm_bf16", _p(x), _p(res), _p(w), _p(out), _p(res_out), rows, H, float(eps), _stream())
    return out


def decode_advance(next_ids, cur_ids, pos, slot, kv_end, hist=None, n_hist=None):
    """End of a decode step in one launch: cur_ids <- next_ids; pos, slot, kv_end += 1; hist[b, n_hist[b]++] = next_ids[b]."""
    for t, n in ((next_ids, "next_ids"), (cur_ids, "cur_ids"), (pos, "pos"), (slot, "slot"), (kv_end, "kv_end")):
        _chk(t, torch.int32, n)
    B = next_ids.numel()
    assert cur_ids.numel() == B and pos.numel() == B and slot.numel() == B and kv_end.numel() == B
    cap = 0
    if hist is not None:
        _chk(hist, torch.int32, "hist"); _chk(n_hist, torch.int32, "n_hist")
        assert hist.dim() == 2 and hist.shape[0] == B and n_hist.numel() == B
        cap = hist.shape[1]
    _lib.call("spider_decode_advance_i32", _p(next_ids), _p(cur_ids), _p(pos), _p(slot), _p(kv_end), _p(hist), _p(n_hist), cap, B, _stream())


def gemv(W, x, bias=None, res=None, norm_w=None, eps=0.0, out=None):
    """out[b,n] = sum_k xin[b,k] W[n,k] (+bias) (+res); xin = rmsnorm(x)*norm_w if norm_w is given. 1 <= B <= 8."""
    _chk(W, BF16, "W"); _chk(x, BF16, "x")
    N, K = W.shape
    B = x.numel() // K
    if out is None:
        out = torch.empty(B, N, dtype=BF16, device=x.device)
    _lib.call("spider_gemv_bf16", _p(W), _p(x), _p(out), _p(bias), _p(res), _p(norm_w), float(eps), B, N, K, _stream())
    return out


def gemv_swiglu(W_gate_up, x, norm_w=None, eps=0.0, out=None):
    _chk(W_gate_up, BF16, "W_gate_up"); _chk(x, BF16, "x")
    I2, K = W_gate_up.shape
    I = I2 // 2
    B = x.numel() // K
    if out is None:
        out = torch.empty(B, I, dtype=BF16, device=x.device)
    _lib.call("spider_gemv_swiglu_bf16", _p(W_gate_up), _p(x), _p(out), _p(norm_w), float(eps), B, I, K, _stream())
    return out


def repack_fm16(W: torch.Tensor, norm_w: Optional[torch.Tensor] = None) -> torch.Tensor:
    """W [N, K] bf16 (K % 64 == 0); with norm_w [K] the copy holds bf16(W * norm_w) for the kernels' fold_rmsnorm form.
    W -> fragment-major copy [ceil(N/16), K/64, 2, 64, 8] for the *_fm kernels: piece (rg, kb, sx)
    holds, at lane 16 g + r, W[16 rg + r, 64 kb + 32 sx + 8 g : + 8]. Pure data movement, done once when an engine loads its
    weights (like the OIHW -> OHWI conv repack); rows are zero-padded to a multiple of 16."""
    dt, _ = _h16(W)
    _chk(W, dt, "W")
    N, K = W.shape
    assert K % 64 == 0, "repack_fm16: K must be a multiple of 64"
    if norm_w is not None:
        W = (W.float() * norm_w.float()[None, :]).to(dt)
    NG = (N + 15) // 16
    if NG * 16 != N:
        W = torch.cat([W, torch.zeros(NG * 16 - N, K, dtype=dt, device=W.device)], 0)
    # [rg, r, kb, sx, g, e] -> [rg, kb, sx, g, r, e]
    return W.view(NG, 16, K // 64, 2, 4, 8).permute(0, 2, 3, 4, 1, 5).contiguous().view(NG, K // 64, 2, 64, 8)


def gemv_fm(Wfm, x, N, bias=None, res=None, out=None, norm_eps=None):
    """out[b, n] = sum_k x[b, k] W[n, k] (+bias) (+res) on the fragment-major copy Wfm = repack_fm16(W); 1 <= B <= 16.
    norm_eps: RMSNorm folded in -- Wfm = repack_fm16(W, norm_w), x un-normalised."""
    _chk(Wfm, BF16, "Wfm"); _chk(x, BF16, "x")
    K = Wfm.shape[1] * 64
    B = x.numel() // K
    assert x.shape[-1] == K and Wfm.shape[0] == (N + 15) // 16
    if out is None:
        out = torch.empty(B, N, dtype=BF16, device=x.device)
    _lib.call("spider_gemv_fm_bf16", _p(Wfm), _p(x), _p(out), _p(bias), _p(res), B, N, K, int(norm_eps is not None),
              float(norm_eps or 0.0), _stream())
    return out


def gemv_swiglu_fm(Wfm_gate_up, x, out=None, norm_eps=None):
    """out = silu(x @ Wg^T) * (x @ Wu^T) on repack_fm16(cat([Wg, Wu])); I % 16 == 0."""
    _chk(Wfm_gate_up, BF16, "Wfm_gate_up"); _chk(x, BF16, "x")
    K = Wfm_gate_up.shape[1] * 64
    I = Wfm_gate_up.shape[0] * 16 // 2
    B = x.numel() // K
    assert x.shape[-1] == K and I % 16 == 0
    if out is None:
        out = torch.empty(B, I, dtype=BF16, device=x.device)
    _lib.call("spider_gemv_swiglu_fm_bf16", _p(Wfm_gate_up), _p(x), _p(out), B, I, K, int(norm_eps is not None), float(norm_eps or 0.0),
              _stream())
    return out


def lm_head_argmax_fm(Wfm, x, V, out_ids=None, logits=None, ws=None, norm_eps=None):
    """Greedy argmax of x @ W^T on the fragment-major lm_head copy; 1 <= B <= 16. x normalised, or (norm_eps given, Wfm =
    repack_fm16(W, norm_w)) the un-normalised last hidden state."""
    _chk(Wfm, BF16, "Wfm"); _chk(x, BF16, "x")
    K = Wfm.shape[1] * 64
    B = x.numel() // K
    npart = lm_head_nparts(V)
    if ws is None:
        ws = (torch.empty(B * npart, dtype=torch.float32, device=x.device), torch.empty(B * npart, dtype=torch.int32, device=x.device))
    assert ws[0].numel() >= B * npart and ws[1].numel() >= B * npart
    if out_ids is None:
        out_ids = torch.empty(B, dtype=torch.int32, device=x.device)
    _lib.call("spider_lm_head_argmax_fm_bf16", _p(Wfm), _p(x), _p(out_ids), _p(logits), _p(ws[0]), _p(ws[1]), B, V, K,
              int(norm_eps is not None), float(norm_eps or 0.0), _stream())
    return out_ids


def lm_head_nparts(V: int) -> int:
    return _lib.load().spider_lm_head_nparts(V)


def lm_head_argmax(W, x, norm_w=None, eps=0.0, out_ids=None, logits=None, ws=None):
    _chk(W, BF16, "W"); _chk(x, BF16, "x")
    V, K = W.shape
    B = x.numel() // K
    npart = lm_head_nparts(V)
    if ws is None:
        ws = (torch.empty(B * npart, dtype=torch.float32, device=x.device),
              torch.empty(B * npart, dtype=torch.int32, device=x.device))
    if out_ids is None:
        out_ids = torch.empty(B, dtype=torch.int32, device=x.device)
    _lib.call("spider_lm_head_argmax_bf16", _p(W), _p(x), _p(norm_w), float(eps), _p(out_ids), _p(logits),
              _p(ws[0]), _p(ws[1]), B, V, K, _stream())
    return out_ids


def rope_kv_append(qkv, pos, slot, cos_sin, q_out, k_cache, v_cache, B, S, n_q, n_kv, d, mrope_section=None):
    """pos [B*S] int32, or [3, B*S] with mrope_section=(t, h, w) rotary pairs per component (Qwen2.5-Omni: 16, 24, 24)."""
    _chk(qkv, BF16, "qkv"); _chk(pos, torch.int32, "pos"); _chk(slot, torch.int32, "slot")
    _chk(cos_sin, torch.float32, "cos_sin"); _chk(q_out, BF16, "q_out")
    _chk(k_cache, BF16, "k_cache"); _chk(v_cache, BF16, "v_cache")
    T_max = k_cache.shape[2]
    sec_t, sec_h = (mrope_section[0], mrope_section[1]) if mrope_section is not None else (0, 0)
    assert qkv.numel() == B * S * (n_q + 2 * n_kv) * d and slot.numel() == B * S
    assert pos.numel() == (3 if mrope_section is not None else 1) * B * S
    assert mrope_section is None or sum(mrope_section) == d // 2
    assert k_cache.shape[0] >= B and k_cache.shape[1] == n_kv and k_cache.shape[3] == d
    assert cos_sin.shape[1] == d
    _lib.call("spider_rope_kv_append_mrope_bf16", _p(qkv), _p(pos), _p(slot), _p(cos_sin), _p(q_out), _p(k_cache),
              _p(v_cache), B, S, n_q, n_kv, d, T_max, sec_t, sec_h, _stream())
    return q_out


def attn_decode(q, k_cache, v_cache, kv_end, kv_beg=None, nsplit=1, ws=None, out=None, scale=None):
    _chk(q, BF16, "q"); _chk(k_cache, BF16, "k_cache"); _chk(v_cache, BF16, "v_cache"); _chk(kv_end, torch.int32, "kv_end")
    B, n_q, d = q.shape
    n_kv, T_max = k_cache.shape[1], k_cache.shape[2]
    if scale is None:
        scale = 1.0 / math.sqrt(d)
    if out is None:
        out = torch.empty(B, n_q * d, dtype=BF16, device=q.device)
    if nsplit > 1 and ws is None:
        ws = (torch.empty(B * n_q * nsplit * d, dtype=torch.float32, device=q.device),
              torch.empty(B * n_q * nsplit * 2, dtype=torch.float32, device=q.device))
    _lib.call("spider_attn_decode_bf16", _p(q), _p(k_cache), _p(v_cache), _p(kv_beg), _p(kv_end), _p(out),
              _p(ws[0]) if ws else None, _p(ws[1]) if ws else None, B, n_q, n_kv, d, T_max, float(scale), nsplit, _stream())
    return out


def attn_decode_fused(qkv, pos, cos_sin, k_cache, v_cache, kv_end, kv_beg, counters, n_q, nsplit, ws, out, scale=None):
    """RoPE + KV append + split-KV decode attention + combine in one launch. kv_end includes the current token."""
    _chk(qkv, BF16, "qkv"); _chk(pos, torch.int32, "pos"); _chk(cos_sin, torch.float32, "cos_sin")
    _chk(k_cache, BF16, "k_cache"); _chk(v_cache, BF16, "v_cache"); _chk(kv_end, torch.int32, "kv_end")
    _chk(counters, torch.int32, "counters"); _chk(out, BF16, "out")
    B = kv_end.numel()
    n_kv, T_max, d = k_cache.shape[1], k_cache.shape[2], k_cache.shape[3]
    assert qkv.numel() == B * (n_q + 2 * n_kv) * d and counters.numel() >= B * n_kv and cos_sin.shape[1] == d
    if scale is None:
        scale = 1.0 / math.sqrt(d)
    _lib.call("spider_attn_decode_fused_bf16", _p(qkv), _p(pos), _p(cos_sin), _p(k_cache), _p(v_cache), _p(kv_beg), _p(kv_end),
              _p(out), _p(ws[0]) if ws else None, _p(ws[1]) if ws else None, _p(counters), B, n_q, n_kv, d, T_max,
              float(scale), nsplit, _stream())
    return out


# --------------------------------------------------------------------------- GEMM / conv / attention
def gemm(A, W, bias=None, res=None, rowbias=None, rows_per_group=0, act=None, out_scale=1.0, out=None, out_f32=False,
         res32=None, want32=False):
    """C = act(A @ W^T + bias + rowbias[row // rows_per_group]) (+ res) * out_scale.  A [..., K], W [N, K].
    fp32 residual stream: res32 (fp32, shape of C) replaces `res` and is added to the unrounded result; want32=True returns
    (C, C32) with C32 the fp32 value that C rounds."""
    dt, sfx = _h16(A)
    _chk(A, dt, "A"); _chk(W, dt, "W")
    N, K = W.shape
    assert A.shape[-1] == K, f"gemm: A[..., {A.shape[-1]}] vs W[{N},{K}]"
    M = A.numel() // K
    # GEGLU epilogue: W = [value rows | gate rows]; SwiGLU epilogue (LlamaMLP): W = [gate rows | up rows]; the output has N/2 columns
    n_out = N // 2 if act in GLU_ACTS else N
    if out is None:
        out = torch.empty(*A.shape[:-1], n_out, dtype=torch.float32 if out_f32 else dt, device=A.device)
    c16, c32 = (None, out) if out.dtype == torch.float32 else (out, None)
    wt = _tiled(W, M)
    o32 = torch.empty(out.shape, dtype=torch.float32, device=A.device) if want32 else None
    if res32 is not None:
        _chk(res32, torch.float32, "res32")
    _lib.call(f"spider_gemm_{sfx}", _p(A), _p(W if wt is None else wt), _p(c16), _p(c32), _p(bias), _p(res), _p(rowbias), rows_per_group,
              M, N, K, K, n_out, ACT[act], float(out_scale), int(wt is not None), _p(res32), _p(o32), _p(_workspace(A.device)), WS_BYTES,
              _stream())
    return (out, o32) if want32 else out


def fold_layernorm(W, gamma, beta, bias=None):
    """One-time parameter folding for gemm_ln (done when an engine loads its weights, like the OIHW -> OHWI repack):
    LayerNorm(x) @ W^T + bias == rstd * (x @ Wf^T - mean * colsum) + colbias with
    Wf = W * gamma (bf16), colsum = Wf.sum(1) (fp32, of the ROUNDED Wf so that the mean term cancels exactly),
    colbias = W @ beta + bias (fp32). Returns (Wf, colsum, colbias)."""
    W32 = W.float()
    Wf = (W32 * gamma.float()[None, :]).to(W.dtype).contiguous()
    colsum = Wf.float().sum(1).contiguous()
    colbias = (W32 @ beta.float())
    if bias is not None:
        colbias = colbias + bias.float()
    return Wf, colsum, colbias.contiguous()


def gemm_ln(A, Wf, colsum, colbias, res=None, act=None, eps=1e-5, out=None):
    """LayerNorm(A) @ W^T + bias (+ res) in one launch; (Wf, colsum, colbias) = fold_layernorm(W, gamma, beta, bias).
    act: None or "geglu". A [..., K] bf16 contiguous rows."""
    dt, sfx = _h16(A)
    _chk(A, dt, "A"); _chk(Wf, dt, "Wf"); _chk(colsum, torch.float32, "colsum"); _chk(colbias, torch.float32, "colbias")
    N, K = Wf.shape
    assert A.shape[-1] == K and colsum.numel() == N and colbias.numel() == N, "gemm_ln: shape mismatch"
    assert act in (None, "geglu", "geglu_exact"), "gemm_ln: only the plain and GEGLU epilogues exist"
    M = A.numel() // K
    n_out = N // 2 if act in GLU_ACTS else N
    if out is None:
        out = torch.empty(*A.shape[:-1], n_out, dtype=dt, device=A.device)
    if res is not None:
        _chk(res, dt, "res")
    wt = _tiled(Wf, M)
    _lib.call(f"spider_gemm_ln_{sfx}", _p(A), _p(Wf if wt is None else wt), _p(out), _p(colsum), _p(colbias), _p(res), M, N, K, n_out,
              ACT[act], float(eps), int(wt is not None), _p(_workspace(A.device)), WS_BYTES, _stream())
    return out


XATTN_LP = 80     # keys per head in the folded cross-attention operands (77 text tokens padded to a multiple of 16)


def xattn_fused(x, mq_fm, mo_fm, colsum, colbias, bias_o, B2, heads, n_keys, eps=1e-5, out=None, x32=None, want32=False):
    """x + to_out(softmax(to_q(LayerNorm(x)) K^T / sqrt(d)) V) with the prompt's K / V folded into mq_fm / mo_fm (see
    include/spider_hip.h, spider_xattn_fused_bf16, and UNetEngine.prepare). x [B2, n_tok, C] or [B2 * n_tok, C] bf16."""
    dt, sfx = _h16(x)
    _chk(x, dt, "x"); _chk(mq_fm, dt, "mq_fm"); _chk(mo_fm, dt, "mo_fm")
    _chk(colsum, torch.float32, "colsum"); _chk(colbias, torch.float32, "colbias"); _chk(bias_o, dt, "bias_o")
    C = x.shape[-1]
    n_tok = x.numel() // (C * B2)
    if out is None:
        out = torch.empty_like(x)
    o32 = torch.empty(out.shape, dtype=torch.float32, device=x.device) if want32 else None
    if x32 is not None:
        _chk(x32, torch.float32, "x32")
    _lib.call(f"spider_xattn_fused_{sfx}", _p(x), _p(mq_fm), _p(mo_fm), _p(colsum), _p(colbias), _p(bias_o), _p(out), B2, n_tok, C,
              heads, n_keys, float(eps), _p(x32), _p(o32), _stream())
    return (out, o32) if want32 else out


def conv2d(x, w, bias=None, res=None, rowbias=None, stride=1, pad=None, ups=False, out_scale=1.0, out=None, res32=None, want32=False,
           gn_groups=None):
    """NHWC conv. x [B,H,W,Cin] bf16, w [Cout,ks,ks,Cin] bf16 -> [B,Ho,Wo,Cout]. res32 / want32: fp32 residual stream as in gemm().
    gn_groups: see conv_ex (GroupNorm partial statistics of the output as the last element of the result)."""
    if gn_groups and not ups:
        ks_ = w.shape[1]
        pd = ks_ // 2 if pad is None else pad
        return conv_ex(x, w, bias=bias, res=res, rowbias=rowbias, stride=stride, pad=(pd, pd), out_scale=out_scale, out=out,
                       res32=res32, want32=want32, gn_groups=gn_groups)
    dt, sfx = _h16(x)
    _chk(x, dt, "x"); _chk(w, dt, "w")
    B, H, Wd, Cin = x.shape
    Cout, ks = w.shape[0], w.shape[1]
    if pad is None:
        pad = ks // 2
    if ks == 3 and stride == 1 and pad == 1 and WS_ENABLE and B * H * Wd * (4 if ups else 1) <= WS_MAX_M and getattr(w, "_spider_weight", False):
        # weight-bound maps (<= 512 output pixels): the general entry point picks the weight-stationary streaming kernel
        return conv_ex(x, w, bias=bias, res=res, rowbias=rowbias, stride=1, pad=(1, 1), up_size=(2 * H, 2 * Wd) if ups else None,
                       out_scale=out_scale, out=out, res32=res32, want32=want32)
    Hs, Ws = (H * 2, Wd * 2) if ups else (H, Wd)
    Ho, Wo = (Hs + 2 * pad - ks) // stride + 1, (Ws + 2 * pad - ks) // stride + 1
    if out is None:
        out = torch.empty(B, Ho, Wo, Cout, dtype=dt, device=x.device)
    wt = _tiled(w, B * Ho * Wo)
    o32 = torch.empty(out.shape, dtype=torch.float32, device=x.device) if want32 else None
    if res32 is not None:
        _chk(res32, torch.float32, "res32")
    _lib.call(f"spider_conv2d_nhwc_{sfx}", _p(x), _p(w if wt is None else wt), _p(out), _p(bias), _p(res), _p(rowbias), B, H, Wd, Cin, Cout,
              ks, stride, pad, int(ups), float(out_scale), int(wt is not None), _p(res32), _p(o32), _p(_workspace(x.device)), WS_BYTES,
              _stream())
    return (out, o32) if want32 else out


class GnPartial:
    """GroupNorm partial statistics of a [B, HW, C] tensor: t [B, nchunk, G, 2] fp32 = (sum, sum of squares) per pixel chunk and group
    (include/spider_hip.h: spider_conv_nhwc_gn). Produced by `conv_ex(..., gn_groups=G)` for its own output, consumed by
    `groupnorm(..., partial=...)` and `gemm_gn_in`."""
    __slots__ = ("t", "nchunk", "groups")

    def __init__(self, t, nchunk, groups):
        self.t, self.nchunk, self.groups = t, nchunk, groups


def groupnorm_stats(x, groups: int, nchunk: int) -> GnPartial:
    """pass 1 of GroupNorm alone (x [B, ..., C] NHWC) with `nchunk` pixel chunks per image"""
    dt, sfx = _h16(x)
    _chk(x, dt, "x")
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    part = torch.empty(B, nchunk, groups, 2, dtype=torch.float32, device=x.device)
    _lib.call(f"spider_groupnorm_stats_nhwc_{sfx}", _p(x), _p(part), B, HW, C, groups, nchunk, _stream())
    return GnPartial(part, nchunk, groups)


def conv_ex(x, w, bias=None, res=None, rowbias=None, stride=1, pad=(0, 0), dil=1, up_size=None, act=None, act_param=0.0,
            out_scale=1.0, out=None, res32=None, want32=False, gn_groups=None):
    """General NHWC conv. x [B,H,W,Cin] bf16, w [Cout,kh,kw,Cin] bf16 -> [B,Ho,Wo,Cout]. up_size=(uh,uw): x is read through
    a nearest upsample to that size (each in (in, 2*in]) before the conv. res32 / want32: fp32 residual stream as in gemm().
    gn_groups=G: also return the GroupNorm partial statistics of the OUTPUT as the last element of the result (a GnPartial):
    written by the conv's own epilogue / split-K reduce where that kernel can, else by one statistics pass -- either way the
    consumer GroupNorm needs none of its own. None is returned in their place when the map is not a multiple of the chunk rows."""
    dt, sfx = _h16(x)
    _chk(x, dt, "x"); _chk(w, dt, "w")
    B, H, Wd, Cin = x.shape
    Cout, kh, kw = w.shape[0], w.shape[1], w.shape[2]
    assert w.shape[3] == Cin, f"conv: x has {Cin} channels, w expects {w.shape[3]}"
    Hs, Ws = up_size if up_size is not None else (H, Wd)
    Ho = (Hs + 2 * pad[0] - dil * (kh - 1) - 1) // stride + 1
    Wo = (Ws + 2 * pad[1] - dil * (kw - 1) - 1) // stride + 1
    if out is None:
        out = torch.empty(B, Ho, Wo, Cout, dtype=dt, device=x.device)
    uh, uw = up_size if up_size is not None else (0, 0)
    wt, wmode = None, 0
    if _ws_eligible(B, H, Wd, Cin, Cout, kh, kw, stride, pad, dil, up_size, Ho, Wo, act):
        wt = _wsfm(w)
        wmode = 2 if wt is not None else 0
    if wt is None:
        wt = _tiled(w, B * Ho * Wo)
        wmode = int(wt is not None)
    o32 = torch.empty(out.shape, dtype=torch.float32, device=x.device) if want32 else None
    if res32 is not None:
        _chk(res32, torch.float32, "res32")
    args = (_p(x), _p(w if wt is None else wt), _p(out), _p(bias), _p(res), _p(rowbias), B, H, Wd, Cin,
            Cout, kh, kw, stride, pad[0], pad[1], dil, uh, uw, ACT[act], float(act_param), float(out_scale), wmode,
            _p(res32), _p(o32), _p(_workspace(x.device)), WS_BYTES, _stream())
    part = None
    HWo = Ho * Wo
    # (every block of the consuming GroupNorm reduces the partials of its image: beyond ~256 chunks -- the UNet3D's temporal norms
    # over 16 frames x 2880 pixels would be 720 -- that prologue costs more than the statistics pass it replaces: measured +0.8 %)
    # (the streaming conv combines its K splits inside the launch and leaves no statistics: on its small maps the consumer's
    # one-launch GroupNorm is cheaper than a reduce kernel that would make them)
    if gn_groups and GN_PRODUCER and HWo % 16 == 0 and HWo <= 16384 and not (wmode == 2 and WS_INLAUNCH):
        buf = torch.empty(B * (HWo // 16) * gn_groups * 2, dtype=torch.float32, device=x.device)
        produced = C.c_int(0)
        _lib.call(f"spider_conv_nhwc_gn_{sfx}", *args, _p(buf), int(gn_groups), C.byref(produced))
        cr = produced.value          # rows per chunk the producing kernel used (64: conv epilogue, 16: split-K reduce), 0: none
        if not cr:                   # this shape's kernel cannot write them: one statistics pass (what the GroupNorm would have run)
            cr = 64 if HWo % 64 == 0 and HWo >= 1024 else 16
            _lib.call(f"spider_groupnorm_stats_nhwc_{sfx}", _p(out), _p(buf), B, HWo, Cout, int(gn_groups), HWo // cr, _stream())
        nchunk = HWo // cr
        part = GnPartial(buf[:B * nchunk * gn_groups * 2].view(B, nchunk, gn_groups, 2), nchunk, gn_groups)
    else:
        _lib.call(f"spider_conv_nhwc_ex_{sfx}", *args)
    if gn_groups:         # part is None where no statistics were made: the consumer runs its own pass
        return (out, o32, part) if want32 else (out, part)
    return (out, o32) if want32 else out


GN_PRODUCER = _os.environ.get("SPIDER_GN_PRODUCER", "1") != "0"     # tuning aid: 0 = every GroupNorm runs its own statistics pass


def gemm_gn_in(A, W, part: GnPartial, gamma, beta, HW: int, eps: float, bias=None, want32=False):
    """C = GroupNorm(A) @ W^T + bias with the normalisation applied inside the GEMM (Transformer2DModel.norm + proj_in in one launch).
    A [B, HW, K] un-normalised, part = its partial statistics."""
    dt, sfx = _h16(A)
    _chk(A, dt, "A"); _chk(W, dt, "W")
    N, K = W.shape
    M = A.numel() // K
    out = torch.empty(*A.shape[:-1], N, dtype=dt, device=A.device)
    o32 = torch.empty(out.shape, dtype=torch.float32, device=A.device) if want32 else None
    wt = _tiled(W, M)
    _lib.call(f"spider_gemm_gn_in_{sfx}", _p(A), _p(W if wt is None else wt), _p(out), _p(bias), M, N, K, N, int(wt is not None),
              _p(part.t), part.nchunk, _p(gamma), _p(beta), part.groups, float(eps), HW, _p(o32), _stream())
    return (out, o32) if want32 else out


def conv1d(x, w, bias=None, res=None, pad=0, dil=1, act=None, act_param=0.0, out=None):
    """x [B,L,Cin] bf16, w [Cout,k,Cin] bf16 (nn.Conv1d weight permuted to O,K,I) -> [B,Lo,Cout]."""
    B, L, Cin = x.shape
    y = conv_ex(x.view(B, 1, L, Cin), w.view(w.shape[0], 1, w.shape[1], Cin), bias=bias,
                res=None if res is None else res.view(B, 1, res.shape[1], res.shape[2]), pad=(0, pad), dil=dil, act=act,
                act_param=act_param, out=None if out is None else out.view(B, 1, out.shape[1], out.shape[2]))
    return y.view(B, y.shape[2], y.shape[3])


def conv_transpose1d(x, w_taps, bias, k: int, stride: int, pad: int):
    """nn.ConvTranspose1d. x [B,L,Cin] bf16; w_taps [k*Cout, Cin] bf16 with row j*Cout+o = weight[:, o, j]
    -> [B, (L-1)*stride - 2*pad + k, Cout]. Per-tap GEMM (fp32) + overlap-add."""
    B, L, Cin = x.shape
    Cout = w_taps.shape[0] // k
    cols = gemm(x, w_taps, out_f32=True)                      # [B, L, k*Cout] fp32
    dt, sfx = _h16(x)
    out = torch.empty(B, (L - 1) * stride - 2 * pad + k, Cout, dtype=dt, device=x.device)
    _lib.call(f"spider_col2im1d_f32_{sfx}", _p(cols), _p(bias), _p(out), B, L, k, stride, pad, Cout, _stream())
    return out


def attention(q, k, v, n_heads, n_kv_heads=None, scale=None, causal=False, kv_off=None, kv_beg=None,
              keep_bits=None, blk=0, q_off=0, out=None):
    """q [B,Lq,Hq*d], k/v [B,Lk,Hkv*d] (last dim contiguous; batch/row strides free) -> [B,Lq,Hq*d]."""
    dt, sfx = _h16(q)
    _chk(q, dt, "q", False); _chk(k, dt, "k", False); _chk(v, dt, "v", False)
    B, Lq, Cq = q.shape
    Lk = k.shape[1]
    Hq = n_heads
    Hkv = n_kv_heads or n_heads
    d = Cq // Hq
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    if scale is None:
        scale = 1.0 / math.sqrt(d)
    if kv_off is None:
        kv_off = Lk - Lq
    if out is None:
        out = torch.empty(B, Lq, Cq, dtype=dt, device=q.device)
    _lib.call(f"spider_attn_{sfx}", _p(q), _p(k), _p(v), _p(out),
              q.stride(0), d, q.stride(1), k.stride(0), d, k.stride(1), v.stride(0), d, v.stride(1),
              out.stride(0), d, out.stride(1),
              B, Hq, Hkv, Lq, Lk, d, float(scale), int(causal), int(kv_off), _p(kv_beg), _p(keep_bits), blk, q_off,
              _stream())
    return out


def story_key_lists(keep_bits, n_keys: int, N: int, img0: int, n_lists: int, q_img0: int):
    """Visible-key lists of the consistent self-attention: list l (query image img0 + l) = keys whose keep bit is set or that lie in
    the image's own block of N tokens. Returns (key_idx int32 [n_lists * stride], tiles int32 [n_lists * ceil(N/128), 4])."""
    _chk(keep_bits, torch.int64, "keep_bits")
    assert keep_bits.numel() * 64 >= n_keys
    stride = (n_keys + 63) // 64 * 64
    key_idx = torch.empty(n_lists * stride, dtype=torch.int32, device=keep_bits.device)
    tiles = torch.empty(n_lists * ((N + 127) // 128), 4, dtype=torch.int32, device=keep_bits.device)
    _lib.call("spider_story_key_lists_i32", _p(keep_bits), n_keys, N, img0, n_lists, q_img0, stride, _p(key_idx), _p(tiles), _stream())
    return key_idx, tiles


def attention_keylist(q, k, v, n_heads, key_idx, tiles, scale=None, out=None):
    """Consistent self-attention over visible-key lists (story_key_lists): q [B,Lq,H*64], k / v [B,Lk,H*64]."""
    dt, sfx = _h16(q)
    _chk(q, dt, "q", False); _chk(k, dt, "k", False); _chk(v, dt, "v", False)
    _chk(key_idx, torch.int32, "key_idx"); _chk(tiles, torch.int32, "tiles")
    B, Lq, Cq = q.shape
    Lk, d = k.shape[1], Cq // n_heads
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1 and tiles.dim() == 2 and tiles.shape[1] == 4
    if scale is None:
        scale = 1.0 / math.sqrt(d)
    if out is None:
        out = torch.empty(B, Lq, Cq, dtype=dt, device=q.device)
    _lib.call(f"spider_attn_keylist_{sfx}", _p(q), _p(k), _p(v), _p(out),
              q.stride(0), d, q.stride(1), k.stride(0), d, k.stride(1), v.stride(0), d, v.stride(1),
              out.stride(0), d, out.stride(1), B, n_heads, n_heads, Lq, Lk, d, float(scale),
              _p(key_idx), key_idx.numel(), _p(tiles), tiles.shape[0], _stream())
    return out


def varlen_tiles(cu_seqlens, device) -> torch.Tensor:
    """cu_seqlens (python ints / tensor, [n_seg + 1]) -> int32 [n_tiles, 4] records {q_start, q_len <= 128, k_start, k_len}
    for attention_varlen: every segment is cut into query tiles of at most 128 rows that all see the whole segment."""
    cu = [int(v) for v in (cu_seqlens.tolist() if hasattr(cu_seqlens, "tolist") else cu_seqlens)]
    rec = []
    for a, b in zip(cu[:-1], cu[1:]):
        if b < a:
            raise ValueError("cu_seqlens must be non-decreasing")
        for q0 in range(a, b, 128):
            rec.append((q0, min(128, b - q0), a, b - a))
    if not rec:
        raise ValueError("cu_seqlens describes no rows")
    return torch.tensor(rec, dtype=torch.int32, device=device)


def attention_varlen(q, k, v, n_heads, tiles, n_kv_heads=None, scale=None, out=None):
    """Packed segments: q [T,Hq*d], k/v [T,Hkv*d] (last dim contiguous, row stride free); tiles from varlen_tiles()."""
    dt, sfx = _h16(q)
    _chk(q, dt, "q", False); _chk(k, dt, "k", False); _chk(v, dt, "v", False)
    T, Cq = q.shape
    Hq, Hkv = n_heads, (n_kv_heads or n_heads)
    d = Cq // Hq
    assert q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1 and k.shape[0] == T and v.shape[0] == T
    assert tiles.dtype == torch.int32 and tiles.dim() == 2 and tiles.shape[1] == 4 and tiles.is_contiguous()
    if scale is None:
        scale = 1.0 / math.sqrt(d)
    if out is None:
        out = torch.empty(T, Cq, dtype=dt, device=q.device)
    _lib.call(f"spider_attn_varlen_{sfx}", _p(q), _p(k), _p(v), _p(out), q.stride(0), k.stride(0), v.stride(0), out.stride(0),
              T, Hq, Hkv, d, float(scale), _p(tiles), tiles.shape[0], _stream())
    return out


def rope_rows_(x, cos_sin, n_heads):
    """In-place half-rotation RoPE. x [T, n_heads*d] bf16 view (row stride free), cos_sin [T, d] fp32 = [cos | sin]."""
    _chk(x, BF16, "x", False); _chk(cos_sin, torch.float32, "cos_sin")
    T, Cx = x.shape
    d = Cx // n_heads
    assert x.stride(1) == 1 and tuple(cos_sin.shape) == (T, d)
    _lib.call("spider_rope_rows_bf16", _p(x), _p(cos_sin), x.stride(0), T, n_heads, d, _stream())
    return x


def attention_cache(q, k_cache, v_cache, Lk, scale=None, causal=True, kv_off=None, kv_beg=None, out=None):
    """Prefill attention against the KV cache. q [B,S,n_q,d]; caches [B,n_kv,T_max,d]; keys [0, Lk)."""
    dt, sfx = _h16(q)
    _chk(q, dt, "q"); _chk(k_cache, dt, "k_cache"); _chk(v_cache, dt, "v_cache")
    B, S, n_q, d = q.shape
    n_kv, T_max = k_cache.shape[1], k_cache.shape[2]
    if scale is None:
        scale = 1.0 / math.sqrt(d)
    if kv_off is None:
        kv_off = Lk - S
    if out is None:
        out = torch.empty(B, S, n_q * d, dtype=dt, device=q.device)
    _lib.call(f"spider_attn_{sfx}", _p(q), _p(k_cache), _p(v_cache), _p(out),
              S * n_q * d, d, n_q * d, n_kv * T_max * d, T_max * d, d, n_kv * T_max * d, T_max * d, d,
              S * n_q * d, d, n_q * d,
              B, n_q, n_kv, S, Lk, d, float(scale), int(causal), int(kv_off), _p(kv_beg), None, 0, 0, _stream())
    return out


# --------------------------------------------------------------------------- "precise" fp32-operand forms (ABI v4)
# The A operand (or the NHWC image) is an fp32 tensor -- the master of the residual stream, or a GroupNorm output kept in fp32 -- and
# is split into hi / lo 16-bit halves inside the kernel (two MFMAs per K step): include/spider_hip.h, DESIGN.md section 4. The
# 16-bit format of the call is that of W.
# "Split once, doubled K": above A32_DUP_MIN_FLOP the operand is written ONCE in the operand-split form [round16(v) | round16(v -
# round16(v))] ([M, 2K]; v = the fp32 tensor, its LayerNorm or its GroupNorm) and the 16-bit tile kernels run over it against the
# weight repeated along K ([W | W], built once per marked weight): the same hi . W + lo . W sum in the fp32 accumulator as the a32
# kernels form with two MFMAs per K step on register-staged tiles, at the speed of the LDS-DMA / 256^2 kernels and with all of their
# epilogues (GEGLU, rowbias, res32 / c32d, GroupNorm statistics).
A32_DUP_MIN_FLOP = float(_os.environ.get("SPIDER_A32_DUP_GF", "4")) * 1e9
A32_DUP_MIN_M = int(_os.environ.get("SPIDER_A32_DUP_MIN_M", "0"))      # rows below which the a32 kernels keep the call ([W | W] doubles the weight bytes)


def _dup_ok(M: int, N: int, K: int) -> bool:
    return 2.0 * M * N * K >= A32_DUP_MIN_FLOP and M >= A32_DUP_MIN_M and K % 8 == 0


def _wdup(W: torch.Tensor):
    """[W | W] along the last axis of a marked weight ([N, K] -> [N, 2K]; OHWI [Cout, kh, kw, Cin] -> [Cout, kh, kw, 2 Cin]), built on
    first use outside stream capture and kept with the tensor; None for unmarked tensors (slices, activations as W)."""
    if not getattr(W, "_spider_weight", False):
        return None
    t = getattr(W, "_spider_dup", None)
    tag = (W._version, W.data_ptr())
    if t is None or getattr(W, "_spider_dup_tag", None) != tag:
        if torch.cuda.is_current_stream_capturing():
            if t is not None:
                raise RuntimeError("a marked weight was modified in place and is first used again under stream capture")
            return None
        t = mark_weight(torch.cat([W, W], -1).contiguous())
        W._spider_dup, W._spider_dup_tag = t, tag
    return t


dup_weight = _wdup


def row_split(x32, dtype, gamma=None, beta=None, eps=1e-5):
    """x32 [..., K] fp32 -> [..., 2K] 16-bit = [hi | lo] of x32 itself (gamma None) or of LayerNorm(x32) * gamma + beta (fp32 statistics)"""
    _chk(x32, torch.float32, "x32")
    K = x32.shape[-1]
    y = torch.empty(*x32.shape[:-1], 2 * K, dtype=dtype, device=x32.device)
    _, sfx = _h16(y)
    if gamma is not None:
        _chk(gamma, dtype, "gamma")
    if beta is not None:
        _chk(beta, dtype, "beta")
    _lib.call(f"spider_row_split_f32_{sfx}", _p(x32), _p(gamma), _p(beta), _p(y), x32.numel() // K, K, float(eps), _stream())
    return y


def groupnorm_f32in_split(x32, gamma, beta, groups=32, eps=1e-5, silu=False, partial: Optional["GnPartial"] = None):
    """GroupNorm (+ SiLU) of the fp32 tensor x32 [B, ..., C] in the operand-split form [B, ..., 2C]"""
    dt, sfx = _h16(gamma)
    _chk(x32, torch.float32, "x32"); _chk(gamma, dt, "gamma"); _chk(beta, dt, "beta")
    B, Cn = x32.shape[0], x32.shape[-1]
    HW = x32.numel() // (B * Cn)
    y = torch.empty(*x32.shape[:-1], 2 * Cn, dtype=dt, device=x32.device)
    ws, pt, nch = None, None, 0
    if partial is not None:
        assert partial.groups == groups and tuple(partial.t.shape) == (B, partial.nchunk, groups, 2)
        pt, nch = partial.t, partial.nchunk
    else:
        ws = torch.empty(B * groupnorm_nchunk(HW) * groups * 2, dtype=torch.float32, device=x32.device)
    _lib.call(f"spider_groupnorm_f32in_split_nhwc_{sfx}", _p(x32), _p(pt), nch, _p(gamma), _p(beta), _p(y), _p(ws), B, HW, Cn, groups,
              float(eps), int(silu), _stream())
    return y


def gemm_a32(A32, W, bias=None, res=None, out_scale=1.0, res32=None, want32=False):
    dt, sfx = _h16(W)
    if _dup_ok(A32.numel() // W.shape[1], W.shape[0], W.shape[1]):
        W2 = _wdup(W)
        if W2 is not None:
            return gemm(row_split(A32, dt), W2, bias=bias, res=res, out_scale=out_scale, res32=res32, want32=want32)
    _chk(A32, torch.float32, "A32"); _chk(W, dt, "W")
    N, K = W.shape
    assert A32.shape[-1] == K, f"gemm_a32: A[..., {A32.shape[-1]}] vs W[{N},{K}]"
    M = A32.numel() // K
    out = torch.empty(*A32.shape[:-1], N, dtype=dt, device=A32.device)
    o32 = torch.empty(out.shape, dtype=torch.float32, device=A32.device) if want32 else None
    if res32 is not None:
        _chk(res32, torch.float32, "res32")
    wt = _tiled(W, M)
    _lib.call(f"spider_gemm_a32_{sfx}", _p(A32), _p(W if wt is None else wt), _p(out), _p(bias), _p(res), M, N, K, K, N, float(out_scale),
              int(wt is not None), _p(res32), _p(o32), _p(_workspace(A32.device)), WS_BYTES, _stream())
    return (out, o32) if want32 else out


def fold_layernorm_exact(W, gamma, beta, bias=None):
    """parameters of gemm_ln_a32 (once per layer): (W itself, gamma, colsum = sum_k gamma_k W[n,k] in fp32, colbias = W @ beta + bias)"""
    W32 = W.float()
    colsum = (W32 * gamma.float()[None, :]).sum(1).contiguous()
    colbias = W32 @ beta.float()
    if bias is not None:
        colbias = colbias + bias.float()
    return (W, gamma.to(W.dtype).contiguous(), colsum, colbias.contiguous(), beta.to(W.dtype).contiguous(),
            None if bias is None else bias.to(W.dtype).contiguous())


def gemm_ln_a32(A32, W, gamma, colsum, colbias, beta=None, bias=None, act=None, eps=1e-5):
    """LayerNorm(A32) @ W^T + bias (+ GEGLU) on the fp32 rows A32 [..., K]: statistics in fp32, gamma applied to A before the hi / lo
    split, W exact (no re-rounded W * gamma). (W, gamma, colsum, colbias, beta, bias) = fold_layernorm_exact(W, gamma, beta, bias);
    beta / bias (the LayerNorm's shift and the projection's bias as 16-bit vectors) serve the split-once route of large calls, which
    applies the LayerNorm exactly as written -- (x - mean) rstd gamma + beta in fp32 -- before the split."""
    dt, sfx = _h16(W)
    if beta is not None and _dup_ok(A32.numel() // W.shape[1], W.shape[0], W.shape[1]):
        W2 = _wdup(W)
        if W2 is not None:
            return gemm(row_split(A32, dt, gamma, beta, eps), W2, bias=bias, act=act)
    _chk(A32, torch.float32, "A32"); _chk(W, dt, "W"); _chk(gamma, dt, "gamma")
    _chk(colsum, torch.float32, "colsum"); _chk(colbias, torch.float32, "colbias")
    N, K = W.shape
    assert A32.shape[-1] == K and colsum.numel() == N and colbias.numel() == N and gamma.numel() == K
    assert act in (None, "geglu", "geglu_exact")
    M = A32.numel() // K
    n_out = N // 2 if act in GLU_ACTS else N
    out = torch.empty(*A32.shape[:-1], n_out, dtype=dt, device=A32.device)
    wt = _tiled(W, M)
    _lib.call(f"spider_gemm_ln_a32_{sfx}", _p(A32), _p(W if wt is None else wt), _p(out), _p(gamma), _p(colsum), _p(colbias), M, N, K, n_out,
              ACT[act], float(eps), int(wt is not None), _stream())
    return out


def gemm_gn_in_a32(A32, W, part: "GnPartial", gamma, beta, HW: int, eps: float, bias=None, want32=False):
    """GroupNorm(A32) @ W^T + bias with A32 [B, HW, K] fp32, normalised in fp32 inside the GEMM and split hi / lo"""
    dt, sfx = _h16(W)
    _chk(A32, torch.float32, "A32"); _chk(W, dt, "W")
    N, K = W.shape
    M = A32.numel() // K
    if _dup_ok(M, N, K):
        W2 = _wdup(W)
        if W2 is not None:
            return gemm(groupnorm_f32in_split(A32, gamma, beta, part.groups, eps, False, partial=part), W2, bias=bias, want32=want32)
    out = torch.empty(*A32.shape[:-1], N, dtype=dt, device=A32.device)
    o32 = torch.empty(out.shape, dtype=torch.float32, device=A32.device) if want32 else None
    wt = _tiled(W, M)
    _lib.call(f"spider_gemm_gn_in_a32_{sfx}", _p(A32), _p(W if wt is None else wt), _p(out), _p(bias), M, N, K, N, int(wt is not None),
              _p(part.t), part.nchunk, _p(gamma), _p(beta), part.groups, float(eps), HW, _p(o32), _stream())
    return (out, o32) if want32 else out


def conv_a32(x32, w, bias=None, res=None, rowbias=None, stride=1, pad=(0, 0), dil=1, up_size=None, out_scale=1.0, res32=None, want32=False):
    """NHWC conv with the fp32 image x32 [B,H,W,Cin] as the A operand; w [Cout,kh,kw,Cin] 16-bit -> [B,Ho,Wo,Cout] 16-bit (+ fp32)."""
    dt, sfx = _h16(w)
    _chk(x32, torch.float32, "x32"); _chk(w, dt, "w")
    B, H, Wd, Cin = x32.shape
    Cout, kh, kw = w.shape[0], w.shape[1], w.shape[2]
    assert w.shape[3] == Cin, f"conv_a32: x has {Cin} channels, w expects {w.shape[3]}"
    Hs, Ws = up_size if up_size is not None else (H, Wd)
    Ho = (Hs + 2 * pad[0] - dil * (kh - 1) - 1) // stride + 1
    Wo = (Ws + 2 * pad[1] - dil * (kw - 1) - 1) // stride + 1
    if _dup_ok(B * Ho * Wo, Cout, kh * kw * Cin) and Cin % 8 == 0:
        # large conv (the UNet's samplers, shortcuts and every conv of the video UNet): the in-kernel hi / lo split lives on the
        # register-staged tiles, 4-5x slower than the LDS-DMA / 256^2 kernels at this size -- split once into [hi | lo] channels and run
        # the fast conv over 2 Cin channels against [W | W]
        w2 = _wdup(w)
        if w2 is not None:
            return conv_ex(row_split(x32, dt), w2, bias=bias, res=res, rowbias=rowbias, stride=stride, pad=pad, dil=dil, up_size=up_size,
                           out_scale=out_scale, res32=res32, want32=want32)
    if 2.0 * B * Ho * Wo * Cout * kh * kw * Cin >= A32_SPLIT_MIN_FLOP and x32.numel() % 8 == 0 and rowbias is None and res is None and out_scale == 1.0:
        # (unmarked weight: no [W | W] copy is kept) split once, run the fast conv on each half, sum in fp32
        hi, lo = split_hilo(x32, dt)
        kw_ = dict(stride=stride, pad=pad, dil=dil, up_size=up_size)
        _, y32 = conv_ex(hi, w, bias=bias, res32=res32, want32=True, **kw_)
        r = conv_ex(lo, w, res32=y32, want32=True, **kw_)
        return r if want32 else r[0]
    out = torch.empty(B, Ho, Wo, Cout, dtype=dt, device=x32.device)
    o32 = torch.empty(out.shape, dtype=torch.float32, device=x32.device) if want32 else None
    if res32 is not None:
        _chk(res32, torch.float32, "res32")
    uh, uw = up_size if up_size is not None else (0, 0)
    wt = _tiled(w, B * Ho * Wo)
    _lib.call(f"spider_conv_nhwc_a32_{sfx}", _p(x32), _p(w if wt is None else wt), _p(out), _p(bias), _p(res), _p(rowbias), B, H, Wd, Cin,
              Cout, kh, kw, stride, pad[0], pad[1], dil, uh, uw, float(out_scale), int(wt is not None), _p(res32), _p(o32),
              _p(_workspace(x32.device)), WS_BYTES, _stream())
    return (out, o32) if want32 else out


A32_SPLIT_MIN_FLOP = float(_os.environ.get("SPIDER_A32_SPLIT_GF", "4")) * 1e9      # conv_a32 above this: split pass + two fast convs


def split_hilo(x32, dtype):
    """fp32 tensor -> (hi, lo) 16-bit tensors with hi + lo == x to ~22 bits"""
    _chk(x32, torch.float32, "x32")
    hi = torch.empty(x32.shape, dtype=dtype, device=x32.device)
    lo = torch.empty_like(hi)
    _, sfx = _h16(hi)
    _lib.call(f"spider_split_hilo_f32_{sfx}", _p(x32), _p(hi), _p(lo), x32.numel(), _stream())
    return hi, lo


def groupnorm_f32in(x32, gamma, beta, groups=32, eps=1e-5, silu=False, partial: Optional["GnPartial"] = None, want16=True, want32=False):
    """GroupNorm (+ SiLU) of the fp32 tensor x32 [B, ..., C]; returns the 16-bit result (rounded once), the fp32 result, or both."""
    dt, sfx = _h16(gamma)
    _chk(x32, torch.float32, "x32"); _chk(gamma, dt, "gamma"); _chk(beta, dt, "beta")
    B, Cn = x32.shape[0], x32.shape[-1]
    HW = x32.numel() // (B * Cn)
    y = torch.empty(x32.shape, dtype=dt, device=x32.device) if want16 else None
    y32 = torch.empty_like(x32) if want32 else None
    ws, pt, nch = None, None, 0
    if partial is not None:
        assert partial.groups == groups and tuple(partial.t.shape) == (B, partial.nchunk, groups, 2)
        pt, nch = partial.t, partial.nchunk
    else:
        ws = torch.empty(B * groupnorm_nchunk(HW) * groups * 2, dtype=torch.float32, device=x32.device)
    _lib.call(f"spider_groupnorm_f32in_nhwc_{sfx}", _p(x32), _p(pt), nch, _p(gamma), _p(beta), _p(y), _p(y32), _p(ws), B, HW, Cn, groups,
              float(eps), int(silu), _stream())
    return (y, y32) if (want16 and want32) else (y if want16 else y32)


def conv2d_small_cin_f32in(x32, w, bias=None, want32=False):
    dt, sfx = _h16(w)
    _chk(x32, torch.float32, "x32"); _chk(w, dt, "w")
    B, H, Wd, Cin = x32.shape
    Cout, ks = w.shape[0], w.shape[1]
    out = torch.empty(B, H, Wd, Cout, dtype=dt, device=x32.device)
    o32 = torch.empty(out.shape, dtype=torch.float32, device=x32.device) if want32 else None
    _lib.call(f"spider_conv2d_small_cin_f32in_{sfx}", _p(x32), _p(w), _p(bias), _p(out), _p(o32), B, H, Wd, Cin, Cout, ks, _stream())
    return (out, o32) if want32 else out


def conv2d_small_cout_f32in(x32, w, bias=None):
    dt, sfx = _h16(w)
    _chk(x32, torch.float32, "x32"); _chk(w, dt, "w")
    B, H, Wd, Cin = x32.shape
    Cout, ks = w.shape[0], w.shape[1]
    out = torch.empty(B, H, Wd, Cout, dtype=torch.float32, device=x32.device)
    _lib.call(f"spider_conv2d_small_cout_f32in_{sfx}", _p(x32), _p(w), _p(bias), _p(out), B, H, Wd, Cin, Cout, ks, _stream())
    return out


def latent_to_nhwc_f32(lat, reps=1, scale=1.0):
    """fp32 NCHW [B,C,H,W] -> fp32 NHWC [reps*B,H,W,C]"""
    _chk(lat, torch.float32, "lat")
    B, Cn, H, Wd = lat.shape
    out = torch.empty(reps * B, H, Wd, Cn, dtype=torch.float32, device=lat.device)
    _lib.call("spider_latent_to_nhwc_f32", _p(lat), _p(out), B, Cn, H * Wd, reps, float(scale), _stream())
    return out


def concat_channels_f32(a32, b32):
    """channel concat of two fp32 NHWC tensors: a copy of bit patterns, done by the 16-bit concat kernel on the 2-halves-per-float view"""
    _chk(a32, torch.float32, "a32"); _chk(b32, torch.float32, "b32")
    C1, C2 = a32.shape[-1], b32.shape[-1]
    out = torch.empty(*a32.shape[:-1], C1 + C2, dtype=torch.float32, device=a32.device)
    rows = a32.numel() // C1
    assert b32.numel() // C2 == rows
    _lib.call("spider_concat_channels_bf16", _p(a32), _p(b32), _p(out), rows, 2 * C1, 2 * C2, _stream())
    return out


# --------------------------------------------------------------------------- UNet elementwise
def groupnorm_nchunk(HW: int) -> int:
    return _lib.load().spider_groupnorm_nchunk(HW)


def groupnorm(x, gamma, beta, groups=32, eps=1e-5, silu=False, out=None, ws=None, partial: Optional["GnPartial"] = None):
    """x [B, ..., C] NHWC bf16."""
    dt, sfx = _h16(x)
    _chk(x, dt, "x"); _chk(gamma, dt, "gamma"); _chk(beta, dt, "beta")
    B, Cn = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cn)
    if out is None:
        out = torch.empty_like(x)
    if partial is not None:      # statistics from the producer of x (conv_ex(..., gn_groups=...)): the apply pass alone
        assert partial.groups == groups and tuple(partial.t.shape) == (B, partial.nchunk, groups, 2)
        _lib.call(f"spider_groupnorm_apply_nhwc_{sfx}", _p(x), _p(partial.t), partial.nchunk, _p(gamma), _p(beta), _p(out), B, HW, Cn,
                  groups, float(eps), int(silu), _stream())
        return out
    if ws is None:
        ws = torch.empty(B * groupnorm_nchunk(HW) * groups * 2, dtype=torch.float32, device=x.device)
    _lib.call(f"spider_groupnorm_nhwc_{sfx}", _p(x), _p(gamma), _p(beta), _p(out), _p(ws), B, HW, Cn, groups, float(eps),
              int(silu), _stream())
    return out


def groupnorm_cat(x1, x2, gamma, beta, groups=32, eps=1e-5, silu=False, ws=None):
    """GroupNorm(+SiLU) of cat([x1, x2], channel) without a concat launch; returns (normalised, concatenated input)."""
    dt, sfx = _h16(x1)
    _chk(x1, dt, "x1"); _chk(x2, dt, "x2"); _chk(gamma, dt, "gamma"); _chk(beta, dt, "beta")
    B, C1, C2 = x1.shape[0], x1.shape[-1], x2.shape[-1]
    HW = x1.numel() // (B * C1)
    assert x2.shape[:-1] == x1.shape[:-1] and gamma.numel() == C1 + C2, "groupnorm_cat: sources must share [B, ..., *]"
    out = torch.empty(*x1.shape[:-1], C1 + C2, dtype=dt, device=x1.device)
    cat = torch.empty_like(out)
    if ws is None:
        ws = torch.empty(B * groupnorm_nchunk(HW) * groups * 2, dtype=torch.float32, device=x1.device)
    _lib.call(f"spider_groupnorm_cat_nhwc_{sfx}", _p(x1), _p(x2), _p(gamma), _p(beta), _p(out), _p(cat), _p(ws), B, HW, C1, C2,
              groups, float(eps), int(silu), _stream())
    return out, cat


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    dt, sfx = _h16(x)
    _chk(x, dt, "x"); _chk(gamma, dt, "gamma"); _chk(beta, dt, "beta")
    Cn = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    _lib.call(f"spider_layernorm_{sfx}", _p(x), _p(gamma), _p(beta), _p(out), x.numel() // Cn, Cn, float(eps), _stream())
    return out


def geglu(x, out=None):
    dt, sfx = _h16(x)
    _chk(x, dt, "x")
    inner = x.shape[-1] // 2
    M = x.numel() // (2 * inner)
    if out is None:
        out = torch.empty(*x.shape[:-1], inner, dtype=dt, device=x.device)
    _lib.call(f"spider_geglu_{sfx}", _p(x), _p(out), M, inner, _stream())
    return out


def swiglu(x, out=None):
    dt, sfx = _h16(x)
    _chk(x, dt, "x")
    inner = x.shape[-1] // 2
    M = x.numel() // (2 * inner)
    if out is None:
        out = torch.empty(*x.shape[:-1], inner, dtype=dt, device=x.device)
    _lib.call(f"spider_swiglu_{sfx}", _p(x), _p(out), M, inner, _stream())
    return out


def concat_channels(a, b, out=None):
    dt, sfx = _h16(a)
    _chk(a, dt, "a"); _chk(b, dt, "b")
    C1, C2 = a.shape[-1], b.shape[-1]
    rows = a.numel() // C1
    assert b.numel() // C2 == rows
    if out is None:
        out = torch.empty(*a.shape[:-1], C1 + C2, dtype=dt, device=a.device)
    _lib.call(f"spider_concat_channels_{sfx}", _p(a), _p(b), _p(out), rows, C1, C2, _stream())
    return out


def act(x, kind: str, param: float = 0.0, out=None):
    dt, sfx = _h16(x)
    _chk(x, dt, "x")
    if out is None:
        out = torch.empty_like(x)
    _lib.call(f"spider_act_ex_{sfx}", _p(x), _p(out), x.numel(), ACT[kind], float(param), _stream())
    return out


def add_scaled(a, b, scale: float, out=None):
    """bf16((a + b) * scale)"""
    dt, sfx = _h16(a)
    _chk(a, dt, "a"); _chk(b, dt, "b")
    if out is None:
        out = torch.empty_like(a)
    _lib.call(f"spider_add_scaled_{sfx}", _p(a), _p(b), _p(out), a.numel(), float(scale), _stream())
    return out


def axpby(a, b, alpha: float, beta: float, out=None):
    """bf16(alpha * a + beta * b)"""
    dt, sfx = _h16(a)
    _chk(a, dt, "a"); _chk(b, dt, "b")
    assert a.shape == b.shape, (a.shape, b.shape)
    if out is None:
        out = torch.empty_like(a)
    _lib.call(f"spider_axpby_{sfx}", _p(a), _p(b), _p(out), a.numel(), float(alpha), float(beta), _stream())
    return out


def mean_tokens(x, out=None):
    """x [B,T,C] bf16 -> [B,C] mean over T"""
    dt, sfx = _h16(x)
    _chk(x, dt, "x")
    B, T, Cn = x.shape
    if out is None:
        out = torch.empty(B, Cn, dtype=dt, device=x.device)
    _lib.call(f"spider_mean_tokens_{sfx}", _p(x), _p(out), B, T, Cn, _stream())
    return out


def moe_combine(xs, logits, out=None):
    """xs: list of E tensors [B, ...] bf16; logits [B, ld >= E] bf16 (first E columns used) -> sum_e r_e * xs[e] with
    r = sigmoid(logits) / sum(sigmoid(logits))."""
    for t in xs:
        dt, sfx = _h16(t)
        _chk(t, dt, "expert output")
    _chk(logits, dt, "logits")
    B, ld = logits.shape
    E = len(xs)
    assert ld >= E and all(t.shape == xs[0].shape for t in xs) and xs[0].shape[0] == B
    if out is None:
        out = torch.empty_like(xs[0])
    ptrs = (C.c_void_p * E)(*[t.data_ptr() for t in xs])
    _lib.call(f"spider_moe_combine_{sfx}", ptrs, E, _p(logits), ld, _p(out), B, xs[0].numel() // B, _stream())
    return out


def l2_normalize(x, eps: float = 1e-12, out=None):
    """rows of x [..., n] divided by max(L2 norm, eps) (torch.nn.functional.normalize)."""
    dt, sfx = _h16(x)
    _chk(x, dt, "x")
    if out is None:
        out = torch.empty_like(x)
    n = x.shape[-1]
    _lib.call(f"spider_l2_normalize_rows_{sfx}", _p(x), _p(out), x.numel() // n, n, float(eps), _stream())
    return out


def add(a, b, out=None):
    dt, sfx = _h16(a)
    _chk(a, dt, "a"); _chk(b, dt, "b")
    if out is None:
        out = torch.empty_like(a)
    _lib.call(f"spider_add_{sfx}", _p(a), _p(b), _p(out), a.numel(), _stream())
    return out


def conv2d_small_cin(x, w, bias=None, out=None):
    dt, sfx = _h16(x)
    _chk(x, dt, "x"); _chk(w, dt, "w")
    B, H, Wd, Cin = x.shape
    Cout, ks = w.shape[0], w.shape[1]
    if out is None:
        out = torch.empty(B, H, Wd, Cout, dtype=dt, device=x.device)
    _lib.call(f"spider_conv2d_small_cin_{sfx}", _p(x), _p(w), _p(bias), _p(out), B, H, Wd, Cin, Cout, ks, _stream())
    return out


def conv2d_small_cout(x, w, bias=None, out_f32=True, out=None):
    dt, sfx = _h16(x)
    _chk(x, dt, "x"); _chk(w, dt, "w")
    B, H, Wd, Cin = x.shape
    Cout, ks = w.shape[0], w.shape[1]
    if out is None:
        out = torch.empty(B, H, Wd, Cout, dtype=torch.float32 if out_f32 else dt, device=x.device)
    y32, y16 = (out, None) if out.dtype == torch.float32 else (None, out)
    _lib.call(f"spider_conv2d_small_cout_{sfx}", _p(x), _p(w), _p(bias), _p(y32), _p(y16), B, H, Wd, Cin, Cout, ks, _stream())
    return out


def latent_to_nhwc(lat, reps=1, scale=1.0, out=None, dtype=BF16):
    """fp32 NCHW [B,C,H,W] -> 16-bit (dtype, or out's) NHWC [reps*B,H,W,C]."""
    _chk(lat, torch.float32, "lat")
    B, Cn, H, Wd = lat.shape
    if out is None:
        out = torch.empty(reps * B, H, Wd, Cn, dtype=dtype, device=lat.device)
    _, sfx = _h16(out)
    _lib.call(f"spider_latent_to_nhwc_{sfx}", _p(lat), _p(out), B, Cn, H * Wd, reps, float(scale), _stream())
    return out


def cfg_combine(eps2, guidance, out=None):
    """eps2 fp32 NHWC [2*B,H,W,C] (uncond first) -> fp32 NCHW [B,C,H,W]."""
    _chk(eps2, torch.float32, "eps2")
    B2, H, Wd, Cn = eps2.shape
    B = B2 // 2
    if out is None:
        out = torch.empty(B, Cn, H, Wd, dtype=torch.float32, device=eps2.device)
    _lib.call("spider_cfg_combine_f32", _p(eps2), _p(out), B, Cn, H * Wd, float(guidance), _stream())
    return out


def lincomb(tensors, coefs, out=None):
    n = len(tensors)
    for t in tensors:
        _chk(t, torch.float32, "lincomb input")
    if out is None:
        out = torch.empty_like(tensors[0])
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    cf = (C.c_float * n)(*[float(c) for c in coefs])
    _lib.call("spider_lincomb_f32", ptrs, cf, n, _p(out), out.numel(), _stream())
    return out


def nhwc_to_nchw(x, mul=1.0, add_=0.0, clamp01=False, out=None):
    _chk(x, torch.float32, "x")
    B, H, Wd, Cn = x.shape
    if out is None:
        out = torch.empty(B, Cn, H, Wd, dtype=torch.float32, device=x.device)
    _lib.call("spider_nhwc_to_nchw_f32", _p(x), _p(out), B, Cn, H * Wd, float(mul), float(add_), int(clamp01), _stream())
    return out


def softmax_rows(x, scale=1.0, n_valid=None, out=None, dtype=BF16):
    """fp32 [rows, n] -> 16-bit softmax(scale * x) per row over the first n_valid (default n) columns; the rest -> 0."""
    _chk(x, torch.float32, "x")
    n = x.shape[-1]
    if out is None:
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _, sfx = _h16(out)
    _lib.call(f"spider_softmax_rows_f32_{sfx}", _p(x), _p(out), x.numel() // n, n, n if n_valid is None else n_valid, float(scale),
              _stream())
    return out


def pack_keep_bits(u: torch.Tensor, thr: float, n_valid: int) -> torch.Tensor:
    """u [n] fp32 uniforms on the device -> int64 [ceil(n / 64)] bit words: bit j = (u[j] < thr) for j < n_valid (keep vector of
    StoryDiffusion's consistent self-attention, layout of spider_attn_bf16's keep_bits). No host synchronisation."""
    _chk(u, torch.float32, "u")
    n = u.numel()
    words = torch.empty((n + 63) // 64, dtype=torch.int64, device=u.device)
    _lib.call("spider_pack_keep_bits_f32", _p(u), _p(words), n, int(n_valid), float(thr), _stream())
    return words
