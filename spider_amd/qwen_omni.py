"""Multimodal INPUT side of the Qwen2.5-Omni thinker on the HIP kernels.

The reference reaches it through
    inputs = processor(text=text, audios=audios, images=images, videos=videos, return_tensors="pt", padding=True)
    text_ids, audio = model.generate(**inputs, spk=voice, use_audio_in_video=True)     (qwen2.5omni_spider_web.py:461-468)
with `Qwen2_5OmniModel.from_pretrained` (:376-381, qwen2.5omni_infer.py:3). The arithmetic lives in `transformers`
(models/qwen2_5_omni/modeling_qwen2_5_omni.py: Qwen2_5OmniVisionEncoder, Qwen2_5OmniAudioEncoder, get_rope_index and the
masked_scatter splice of Qwen2_5OmniThinkerForConditionalGeneration.forward); oracle/qwen_towers.py restates it and is
pinned to vectors from those classes.

    VisionTowerEngine(pixel_values [patches, C*Tp*P*P], grid_thw) -> image / video embeddings [patches / 4, out_hidden]
    AudioTowerEngine(input_features [mel, frames], feature_lens)  -> audio embeddings [sum(out_len), out_dim]
    QwenOmniThinker.generate(**processor_outputs)                 -> tower(s) -> splice into token embeddings -> (t, h, w)
                                                                    rotary positions -> LlamaEngine.generate

Layout: activations are packed [tokens, C] bf16 (all images / all audio chunks of a call concatenated); q/k/v are column
slices of ONE fused projection output consumed in place by the packed variable-length attention kernel, whose per-block
records {q_start, q_len, k_start, k_len} replace both the window permutation masks and the per-segment Python loop of the
stock implementation. The window re-ordering and its inverse are row gathers (embedding kernel). All index logic
(window index, cu_seqlens, rope index) is host integer work and exact.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import ops

BF16 = torch.bfloat16


# ---------------------------------------------------------------------------------------------- configs
@dataclass
class VisionTowerConfig:
    depth: int = 32
    hidden: int = 1280
    heads: int = 16
    inter: int = 3420
    in_channels: int = 3
    patch: int = 14
    temporal_patch: int = 2
    merge: int = 2
    window: int = 112
    out_hidden: int = 3584
    fullatt: Tuple[int, ...] = (7, 15, 23, 31)
    eps: float = 1e-6

    @staticmethod
    def qwen25_omni_7b():
        return VisionTowerConfig()

    @staticmethod
    def from_hf_dict(c: dict):
        v = c.get("thinker_config", c).get("vision_config", c)
        return VisionTowerConfig(v.get("depth", 32), v.get("hidden_size", 1280), v.get("num_heads", 16),
                                 v.get("intermediate_size", 3420), v.get("in_channels", v.get("in_chans", 3)),
                                 v.get("patch_size", 14), v.get("temporal_patch_size", 2), v.get("spatial_merge_size", 2),
                                 v.get("window_size", 112), v.get("out_hidden_size", 3584),
                                 tuple(v.get("fullatt_block_indexes", (7, 15, 23, 31))), 1e-6)

    @property
    def patch_dim(self):
        return self.in_channels * self.temporal_patch * self.patch * self.patch


@dataclass
class AudioTowerConfig:
    mel: int = 128
    layers: int = 32
    heads: int = 20
    ffn: int = 5120
    d_model: int = 1280
    max_pos: int = 1500
    n_window: int = 100
    out_dim: int = 3584
    eps: float = 1e-5

    @staticmethod
    def qwen25_omni_7b():
        return AudioTowerConfig()

    @staticmethod
    def from_hf_dict(c: dict):
        a = c.get("thinker_config", c).get("audio_config", c)
        return AudioTowerConfig(a.get("num_mel_bins", 128), a.get("encoder_layers", 32), a.get("encoder_attention_heads", 20),
                                a.get("encoder_ffn_dim", 5120), a.get("d_model", 1280), a.get("max_source_positions", 1500),
                                a.get("n_window", 100), a.get("output_dim", 3584), 1e-5)


@dataclass
class OmniTokenIds:
    image: int = 151655
    video: int = 151656
    audio: int = 151646
    vision_start: int = 151652
    audio_start: int = 151647
    position_id_per_seconds: int = 25
    seconds_per_chunk: int = 2

    @staticmethod
    def from_hf_dict(c: dict):
        t = c.get("thinker_config", c)
        return OmniTokenIds(t.get("image_token_index", t.get("image_token_id", 151655)),
                            t.get("video_token_index", t.get("video_token_id", 151656)),
                            t.get("audio_token_index", t.get("audio_token_id", 151646)),
                            t.get("vision_start_token_id", 151652), t.get("audio_start_token_id", 151647),
                            t.get("position_id_per_seconds", 25), t.get("seconds_per_chunk", 2))


def _load_safetensors(path: str, prefixes: Sequence[str]) -> Dict[str, torch.Tensor]:
    from .checkpoint import load_state_dict

    def keep(k: str):
        for p in prefixes:
            if k.startswith(p):
                return k[len(p):]
        return None
    return load_state_dict(path, keep=keep)


def _cache_get(cache: dict, key, make, limit: int = 8):
    """Small per-engine cache (insertion-ordered, oldest entry evicted) of index plans + captured graphs per input geometry."""
    if key not in cache:
        if len(cache) >= limit:
            cache.pop(next(iter(cache)))
        cache[key] = make()
    return cache[key]


def _capture(device, ent: dict, fn) -> None:
    """Warm `fn` on a side stream, then capture it; ent gets "graph" and "out" (the static outputs)."""
    s = torch.cuda.Stream(device=device)
    s.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream(device).wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ent["out"] = fn()
    ent["graph"] = g


# ---------------------------------------------------------------------------------------------- host index logic (exact)
def vision_position_ids(grid: Sequence[Tuple[int, int, int]], merge: int) -> torch.Tensor:
    """(h, w) coordinate of every patch in the processor's block-major order (vision_utils.get_vision_position_ids)."""
    out = []
    for t, h, w in grid:
        hp = torch.arange(h)[:, None].expand(h, w).reshape(h // merge, merge, w // merge, merge).transpose(1, 2).flatten()
        wp = torch.arange(w)[None, :].expand(h, w).reshape(h // merge, merge, w // merge, merge).transpose(1, 2).flatten()
        out.append(torch.stack([hp, wp], -1).repeat(t, 1))
    return torch.cat(out, 0)


def vision_window_index(grid, merge: int, window: int, patch: int) -> Tuple[torch.Tensor, List[int]]:
    """Window-major permutation of the merge groups + cumulative window boundaries in patches
    (vision_utils.get_vision_window_index, incl. its unique_consecutive of the empty padded windows)."""
    win = window // merge // patch
    unit = merge * merge
    idx_all, cu, base = [], [0], 0
    for t, h, w in grid:
        gh, gw = h // merge, w // merge
        index = torch.arange(t * gh * gw).reshape(t, gh, gw)
        pad_h, pad_w = win - gh % win, win - gw % win
        nh, nw = (gh + pad_h) // win, (gw + pad_w) // win
        ip = F.pad(index, (0, pad_w, 0, pad_h), "constant", -100)
        ip = ip.reshape(t, nh, win, nw, win).permute(0, 1, 3, 2, 4).reshape(t, nh * nw, win, win)
        seqlens = (ip != -100).sum([2, 3]).reshape(-1)
        ip = ip.reshape(-1)
        idx_all.append(ip[ip != -100] + base)
        cu.extend((seqlens.cumsum(0) * unit + cu[-1]).tolist())
        base += t * gh * gw
    dedup = [cu[0]]
    for v in cu[1:]:
        if v != dedup[-1]:
            dedup.append(v)
    return torch.cat(idx_all, 0), dedup


def audio_chunk_lengths(feature_lens: Sequence[int], n_window: int) -> List[int]:
    out = []
    for L in feature_lens:
        n = -(-L // (2 * n_window))
        tail = L % (2 * n_window)
        out += [2 * n_window] * (n - 1) + [tail if tail else 2 * n_window]
    return out


def audio_output_lengths(feature_lens: Sequence[int]) -> List[int]:
    """Qwen2_5OmniAudioEncoder._get_feat_extract_output_lengths: one stride-2 conv, then stride-2 average pooling."""
    return [(((int(L) - 1) // 2 + 1) - 2) // 2 + 1 for L in feature_lens]


# ---------------------------------------------------------------------------------------------- vision tower
class VisionTowerEngine:
    """Qwen2_5OmniVisionEncoder.forward. Weight names are the state-dict names under `thinker.visual.`."""

    def __init__(self, cfg: VisionTowerConfig, weights: Dict[str, torch.Tensor], device="cuda:0"):
        self.cfg, self.device = cfg, torch.device(device)
        c = cfg
        g = lambda k: weights[k].to(device=self.device, dtype=BF16).contiguous()
        H, I = c.hidden, c.inter
        if (H // c.heads) % 16 or H % 8 or c.patch_dim % 8:
            raise ValueError("vision tower: head_dim must be a multiple of 16, hidden and patch size multiples of 8")
        self.Ip = Ip = (I + 7) // 8 * 8    # 3420 -> 3424: zero rows / columns keep the GEMM's 16-byte row granularity
        self.w_patch = g("patch_embed.proj.weight").reshape(H, -1).contiguous()
        zrow = lambda n, k: torch.zeros(n, k, dtype=BF16, device=self.device)
        self.blocks = []
        for l in range(c.depth):
            b = f"blocks.{l}."
            gu = torch.cat([g(b + "mlp.gate_proj.weight"), zrow(Ip - I, H), g(b + "mlp.up_proj.weight"), zrow(Ip - I, H)], 0)
            gub = torch.cat([g(b + "mlp.gate_proj.bias"), zrow(1, Ip - I)[0], g(b + "mlp.up_proj.bias"), zrow(1, Ip - I)[0]], 0)
            self.blocks.append(dict(
                n1=g(b + "norm1.weight"), n2=g(b + "norm2.weight"),
                w_qkv=torch.cat([g(b + "attn.q.weight"), g(b + "attn.k.weight"), g(b + "attn.v.weight")], 0).contiguous(),
                b_qkv=torch.cat([g(b + "attn.q.bias"), g(b + "attn.k.bias"), g(b + "attn.v.bias")], 0).contiguous(),
                w_o=g(b + "attn.proj.weight"), b_o=g(b + "attn.proj.bias"),
                w_gu=gu.contiguous(), b_gu=gub.contiguous(),
                w_dn=torch.cat([g(b + "mlp.down_proj.weight"), zrow(H, Ip - I)], 1).contiguous(), b_dn=g(b + "mlp.down_proj.bias")))
        self.ln_q = g("merger.ln_q.weight")
        self.m0 = (g("merger.mlp.0.weight"), g("merger.mlp.0.bias"))
        self.m2 = (g("merger.mlp.2.weight"), g("merger.mlp.2.bias"))
        d = H // c.heads
        self.inv_freq = 1.0 / (10000.0 ** (torch.arange(0, d // 2, 2, dtype=torch.float) / (d // 2)))
        self._cache: dict = {}

    @classmethod
    def random_init(cls, cfg: VisionTowerConfig, device="cuda:0", seed=0):
        gen = torch.Generator(device=device).manual_seed(seed)
        c, w = cfg, {}
        r = lambda *s: (torch.randn(*s, generator=gen, device=device) / math.sqrt(s[-1] if len(s) == 2 else math.prod(s[1:]))).to(BF16)
        zb = lambda n: torch.zeros(n, device=device, dtype=BF16)
        one = lambda n: torch.ones(n, device=device, dtype=BF16)
        w["patch_embed.proj.weight"] = r(c.hidden, c.in_channels, c.temporal_patch, c.patch, c.patch)
        for l in range(c.depth):
            b = f"blocks.{l}."
            w[b + "norm1.weight"] = one(c.hidden); w[b + "norm2.weight"] = one(c.hidden)
            for n in ("q", "k", "v", "proj"):
                w[b + f"attn.{n}.weight"] = r(c.hidden, c.hidden); w[b + f"attn.{n}.bias"] = zb(c.hidden)
            for n, (o, i) in {"gate_proj": (c.inter, c.hidden), "up_proj": (c.inter, c.hidden), "down_proj": (c.hidden, c.inter)}.items():
                w[b + f"mlp.{n}.weight"] = r(o, i); w[b + f"mlp.{n}.bias"] = zb(o)
        m = c.hidden * c.merge * c.merge
        w["merger.ln_q.weight"] = one(c.hidden)
        w["merger.mlp.0.weight"] = r(m, m); w["merger.mlp.0.bias"] = zb(m)
        w["merger.mlp.2.weight"] = r(c.out_hidden, m); w["merger.mlp.2.bias"] = zb(c.out_hidden)
        return cls(cfg, w, device)

    @classmethod
    def from_pretrained(cls, path: str, device="cuda:0"):
        import json, os
        cfg = VisionTowerConfig.from_hf_dict(json.load(open(os.path.join(path, "config.json"))))
        return cls(cfg, _load_safetensors(path, ("thinker.visual.", "visual.")), device)

    def plan(self, grid_thw) -> dict:
        """Host-side (exact) index work for one packed call: window permutation, rotary table, attention tile records."""
        c, dv = self.cfg, self.device
        grid = [tuple(int(v) for v in g) for g in (grid_thw.tolist() if hasattr(grid_thw, "tolist") else grid_thw)]
        for t, h, w in grid:
            if h % c.merge or w % c.merge or t < 1:
                raise ValueError(f"grid_thw {t, h, w}: h and w must be multiples of the spatial merge size {c.merge}")
        unit = c.merge * c.merge
        T = sum(t * h * w for t, h, w in grid)
        win_idx, cu_win = vision_window_index(grid, c.merge, c.window, c.patch)
        cu_full = [0]
        for t, h, w in grid:
            cu_full += [cu_full[-1] + (i + 1) * h * w for i in range(t)]
        patch_idx = (win_idx[:, None] * unit + torch.arange(unit)[None]).flatten()
        ang = (vision_position_ids(grid, c.merge).unsqueeze(-1).float() * self.inv_freq).flatten(1)[patch_idx]
        return dict(T=T, gather=patch_idx.to(torch.int32).to(dv), reverse=torch.argsort(win_idx).to(torch.int32).to(dv),
                    cos_sin=torch.cat([ang.cos(), ang.sin()], 1).contiguous().to(dv),
                    tiles_win=ops.varlen_tiles(cu_win, dv), tiles_full=ops.varlen_tiles(cu_full, dv))

    def _run(self, px: torch.Tensor, p: dict):
        c = self.cfg
        T, H, nh = p["T"], c.hidden, c.heads
        x = ops.embed(ops.gemm(px, self.w_patch), p["gather"])                     # patch embedding, window order
        for l, bw in enumerate(self.blocks):
            h = ops.rmsnorm(x, bw["n1"], c.eps)
            qkv = ops.gemm(h, bw["w_qkv"], bias=bw["b_qkv"])
            ops.rope_rows_(qkv[:, :H], p["cos_sin"], nh)
            ops.rope_rows_(qkv[:, H:2 * H], p["cos_sin"], nh)
            a = ops.attention_varlen(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], nh,
                                     p["tiles_full"] if l in c.fullatt else p["tiles_win"])
            x = ops.gemm(a, bw["w_o"], bias=bw["b_o"], res=x)
            h = ops.rmsnorm(x, bw["n2"], c.eps)
            m = ops.swiglu(ops.gemm(h, bw["w_gu"], bias=bw["b_gu"]))
            x = ops.gemm(m, bw["w_dn"], bias=bw["b_dn"], res=x)
        unit = c.merge * c.merge
        m = ops.rmsnorm(x, self.ln_q, c.eps).view(T // unit, unit * H)
        m = ops.gemm(ops.gemm(m, self.m0[0], bias=self.m0[1], act="gelu"), self.m2[0], bias=self.m2[1])
        return x, ops.embed(m, p["reverse"])

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor, grid_thw, return_last_hidden: bool = False, use_graph: bool = True):
        """pixel_values [patches, C*Tp*P*P]; grid_thw [[t, h, w], ...] in patches -> pooler_output [patches / merge^2,
        out_hidden] bf16 in the processor's original order (and last_hidden_state in window order on request).
        The index plan and the captured hipGraph (~420 launches) are cached per grid_thw: a repeated image size costs one
        copy + one graph replay."""
        c = self.cfg
        key = tuple(tuple(int(v) for v in g) for g in (grid_thw.tolist() if hasattr(grid_thw, "tolist") else grid_thw))
        ent = _cache_get(self._cache, key, lambda: {"plan": self.plan(key)})
        p = ent["plan"]
        if tuple(pixel_values.shape) != (p["T"], c.patch_dim):
            raise ValueError(f"pixel_values must be [{p['T']}, {c.patch_dim}] for grid_thw, got {tuple(pixel_values.shape)}")
        px = pixel_values.to(device=self.device, dtype=BF16).contiguous()
        if not use_graph:
            last, pooled = self._run(px, p)
        else:
            if "graph" not in ent:
                ent["px"] = px.clone()
                _capture(self.device, ent, lambda: self._run(ent["px"], p))
            ent["px"].copy_(px)
            ent["graph"].replay()
            last, pooled = (t.clone() for t in ent["out"])
        return (last, pooled) if return_last_hidden else pooled

    __call__ = forward

    def has_graph(self, grid_thw) -> bool:
        """is the hipGraph of this exact grid_thw captured (and still in the 8-entry cache)? A miss means forward() will capture."""
        key = tuple(tuple(int(x) for x in g) for g in (grid_thw.tolist() if hasattr(grid_thw, "tolist") else grid_thw))
        ent = self._cache.get(key)
        return ent is not None and "graph" in ent


# ---------------------------------------------------------------------------------------------- audio tower
class AudioTowerEngine:
    """Qwen2_5OmniAudioEncoder.forward. Weight names are the state-dict names under `thinker.audio_tower.`."""

    def __init__(self, cfg: AudioTowerConfig, weights: Dict[str, torch.Tensor], device="cuda:0"):
        self.cfg, self.device = cfg, torch.device(device)
        c = cfg
        g = lambda k: weights[k].to(device=self.device, dtype=BF16).contiguous()
        D = c.d_model
        if (D // c.heads) % 8 or c.mel % 8 or D % 8:
            raise ValueError("audio tower: head_dim, mel bins and d_model must be multiples of 8")
        # nn.Conv1d weight [Cout, Cin, k] -> [Cout, 1, k, Cin] (OHWI with a 1 x k kernel)
        self.c1 = (g("conv1.weight").permute(0, 2, 1).contiguous()[:, None].contiguous(), g("conv1.bias"))
        self.c2 = (g("conv2.weight").permute(0, 2, 1).contiguous()[:, None].contiguous(), g("conv2.bias"))
        inc = math.log(10000.0) / (D // 2 - 1)
        st = torch.arange(c.max_pos)[:, None] * torch.exp(-inc * torch.arange(D // 2).float())[None, :]
        self.pos = torch.cat([torch.sin(st), torch.cos(st)], 1).to(device=self.device, dtype=BF16).contiguous()
        self.layers = []
        for l in range(c.layers):
            b = f"layers.{l}."
            self.layers.append(dict(
                ln1=(g(b + "self_attn_layer_norm.weight"), g(b + "self_attn_layer_norm.bias")),
                w_qkv=torch.cat([g(b + "self_attn.q_proj.weight"), g(b + "self_attn.k_proj.weight"), g(b + "self_attn.v_proj.weight")], 0).contiguous(),
                b_qkv=torch.cat([g(b + "self_attn.q_proj.bias"), torch.zeros(D, dtype=BF16, device=self.device),
                                 g(b + "self_attn.v_proj.bias")], 0).contiguous(),
                w_o=g(b + "self_attn.out_proj.weight"), b_o=g(b + "self_attn.out_proj.bias"),
                ln2=(g(b + "final_layer_norm.weight"), g(b + "final_layer_norm.bias")),
                w1=g(b + "fc1.weight"), b1=g(b + "fc1.bias"), w2=g(b + "fc2.weight"), b2=g(b + "fc2.bias")))
        self.ln_post = (g("ln_post.weight"), g("ln_post.bias"))
        self.proj = (g("proj.weight"), g("proj.bias"))
        self._cache: dict = {}

    @classmethod
    def random_init(cls, cfg: AudioTowerConfig, device="cuda:0", seed=0):
        gen = torch.Generator(device=device).manual_seed(seed)
        c, w = cfg, {}
        r = lambda *s: (torch.randn(*s, generator=gen, device=device) / math.sqrt(math.prod(s[1:]))).to(BF16)
        zb = lambda n: torch.zeros(n, device=device, dtype=BF16)
        one = lambda n: torch.ones(n, device=device, dtype=BF16)
        D = c.d_model
        w["conv1.weight"] = r(D, c.mel, 3); w["conv1.bias"] = zb(D)
        w["conv2.weight"] = r(D, D, 3); w["conv2.bias"] = zb(D)
        for l in range(c.layers):
            b = f"layers.{l}."
            w[b + "self_attn.k_proj.weight"] = r(D, D)
            for n in ("q_proj", "v_proj", "out_proj"):
                w[b + f"self_attn.{n}.weight"] = r(D, D); w[b + f"self_attn.{n}.bias"] = zb(D)
            for n in ("self_attn_layer_norm", "final_layer_norm"):
                w[b + n + ".weight"] = one(D); w[b + n + ".bias"] = zb(D)
            w[b + "fc1.weight"] = r(c.ffn, D); w[b + "fc1.bias"] = zb(c.ffn)
            w[b + "fc2.weight"] = r(D, c.ffn); w[b + "fc2.bias"] = zb(D)
        w["ln_post.weight"] = one(D); w["ln_post.bias"] = zb(D)
        w["proj.weight"] = r(c.out_dim, D); w["proj.bias"] = zb(c.out_dim)
        return cls(cfg, w, device)

    @classmethod
    def from_pretrained(cls, path: str, device="cuda:0"):
        import json, os
        cfg = AudioTowerConfig.from_hf_dict(json.load(open(os.path.join(path, "config.json"))))
        return cls(cfg, _load_safetensors(path, ("thinker.audio_tower.", "audio_tower.")), device)

    def _cnn(self, feats_lc: torch.Tensor) -> torch.Tensor:
        """[B, L, mel] bf16 -> GELU(conv2(GELU(conv1))) + positions, [B, (L-1)//2+1, D]. A right-padded, masked chunk of the
        stock implementation equals the chunk convolved on its own with zero padding, so no mask is needed."""
        e = ops.conv_ex(feats_lc[:, None], self.c1[0], bias=self.c1[1], pad=(0, 1), act="gelu")
        e = ops.conv_ex(e, self.c2[0], bias=self.c2[1], stride=2, pad=(0, 1), act="gelu")[:, 0]
        B, L2, D = e.shape
        return ops.add(e, self.pos[:L2][None].expand(B, L2, D).contiguous())

    def plan(self, lens: Tuple[int, ...]) -> dict:
        c, dv = self.cfg, self.device
        chunks = audio_chunk_lengths(lens, c.n_window)
        if (max(chunks) - 1) // 2 + 1 > c.max_pos:
            raise ValueError("chunk longer than the position table")
        starts, cu = [0], [0]
        for L in chunks:
            starts.append(starts[-1] + L)
            cu.append(cu[-1] + (L - 1) // 2 + 1)
        idx, off = [], 0
        for L in lens:   # stride-2 average pooling of consecutive post-CNN frames, per audio (an odd last frame is dropped)
            after = (L - 1) // 2 + 1
            idx += [off + 2 * i for i in range((after - 2) // 2 + 1)]
            off += after
        i0 = torch.tensor(idx, dtype=torch.int32, device=dv)
        return dict(chunks=chunks, starts=starts, tiles=ops.varlen_tiles(cu, dv), pool0=i0, pool1=i0 + 1)

    def _run(self, f: torch.Tensor, p: dict) -> torch.Tensor:
        """f [frames, mel] bf16."""
        c = self.cfg
        chunks, starts, full = p["chunks"], p["starts"], 2 * c.n_window
        full_ids = [i for i, L in enumerate(chunks) if L == full]
        emb: List[Optional[torch.Tensor]] = [None] * len(chunks)
        if full_ids:   # all full chunks as one batch
            e = self._cnn(torch.stack([f[starts[i]:starts[i] + full] for i in full_ids], 0))
            for j, i in enumerate(full_ids):
                emb[i] = e[j]
        for i, L in enumerate(chunks):
            if emb[i] is None:
                emb[i] = self._cnn(f[starts[i]:starts[i] + L][None].contiguous())[0]
        x = torch.cat(emb, 0).contiguous()
        D, nh = c.d_model, c.heads
        for lw in self.layers:
            h = ops.layernorm(x, *lw["ln1"], c.eps)
            qkv = ops.gemm(h, lw["w_qkv"], bias=lw["b_qkv"])
            a = ops.attention_varlen(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], nh, p["tiles"])
            x = ops.gemm(a, lw["w_o"], bias=lw["b_o"], res=x)
            h = ops.layernorm(x, *lw["ln2"], c.eps)
            x = ops.gemm(ops.gemm(h, lw["w1"], bias=lw["b1"], act="gelu"), lw["w2"], bias=lw["b2"], res=x)
        pooled = ops.add_scaled(ops.embed(x, p["pool0"]), ops.embed(x, p["pool1"]), 0.5)
        return ops.gemm(ops.layernorm(pooled, *self.ln_post, c.eps), self.proj[0], bias=self.proj[1])

    @torch.no_grad()
    def forward(self, input_features: torch.Tensor, feature_lens: Sequence[int], use_graph: bool = True) -> torch.Tensor:
        """input_features [mel, total_frames] (audios concatenated along time, as get_audio_features hands them over);
        feature_lens: mel frames per audio. -> [sum(out_len), out_dim] bf16. Plan + hipGraph cached per feature_lens."""
        c, dv = self.cfg, self.device
        lens = tuple(int(v) for v in (feature_lens.tolist() if hasattr(feature_lens, "tolist") else feature_lens))
        if input_features.dim() != 2 or input_features.shape[0] != c.mel or input_features.shape[1] != sum(lens):
            raise ValueError(f"input_features must be [{c.mel}, {sum(lens)}], got {tuple(input_features.shape)}")
        if min(lens) < 3:
            raise ValueError("every audio needs at least 3 mel frames (stride-2 conv + stride-2 pooling)")
        ent = _cache_get(self._cache, lens, lambda: {"plan": self.plan(lens)})
        f = input_features.to(device=dv, dtype=BF16).t().contiguous()              # [frames, mel]
        if not use_graph:
            return self._run(f, ent["plan"])
        if "graph" not in ent:
            ent["f"] = f.clone()
            _capture(dv, ent, lambda: self._run(ent["f"], ent["plan"]))
        ent["f"].copy_(f)
        ent["graph"].replay()
        return ent["out"].clone()

    __call__ = forward

    def has_graph(self, feature_lens) -> bool:
        lens = tuple(int(x) for x in (feature_lens.tolist() if hasattr(feature_lens, "tolist") else feature_lens))
        ent = self._cache.get(lens)
        return ent is not None and "graph" in ent


# ---------------------------------------------------------------------------------------------- prompt assembly
def _vision_pos(start: int, t_index: List[int], gh: int, gw: int) -> torch.Tensor:
    n = len(t_index)
    hi = torch.arange(gh).view(1, -1, 1).expand(n, -1, gw).flatten()
    wi = torch.arange(gw).view(1, 1, -1).expand(n, gh, -1).flatten()
    ti = torch.tensor(t_index, dtype=torch.long).view(-1, 1).expand(-1, gh * gw).flatten()
    return torch.stack([ti, hi, wi]) + start


def _chunk_bounds(tok: torch.Tensor, per_chunk: int, remove: int) -> List[Tuple[int, int]]:
    out, start, cur = [], 0, 1
    for i in range(len(tok)):
        if int(tok[i]) - remove >= cur * per_chunk:
            out.append((start, i)); start = i; cur += 1
    out.append((start, len(tok)))
    return out


def get_rope_index(ids: OmniTokenIds, merge: int, input_ids: torch.Tensor, image_grid_thw=None, video_grid_thw=None,
                   attention_mask: Optional[torch.Tensor] = None, use_audio_in_video: bool = False,
                   audio_seqlens: Optional[Sequence[int]] = None, second_per_grids: Optional[Sequence[float]] = None):
    """(t, h, w) rotary positions of a prompt with image / audio / video placeholders: text advances all three components
    together, a vision block spreads (t * position_id_per_seconds, row, column) from the running offset, audio advances
    one position per output frame, and a video with its audio track interleaves both in chunks of seconds_per_chunk
    (Qwen2_5OmniPreTrainedModelForConditionalGeneration.get_rope_index). -> (position_ids [3, B, S] long, deltas [B, 1])."""
    input_ids = input_ids.cpu()
    attention_mask = attention_mask.cpu() if attention_mask is not None else None
    B, S = input_ids.shape
    if image_grid_thw is None and video_grid_thw is None:
        am = attention_mask if attention_mask is not None else torch.ones_like(input_ids)
        p = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 1)
        pos = p.unsqueeze(0).expand(3, -1, -1).clone()
        return pos, pos.max(0)[0].max(-1, keepdim=True)[0] + 1 - am.sum(-1, keepdim=True)
    as_list = lambda g: [tuple(int(v) for v in r) for r in (g.tolist() if hasattr(g, "tolist") else (g or []))]
    img, vid = as_list(image_grid_thw), as_list(video_grid_thw)
    pos = torch.ones(3, B, S, dtype=torch.long)
    deltas = []
    ii = vi = ai = 0
    for b in range(B):
        keep = attention_mask[b] == 1 if attention_mask is not None else torch.ones(S, dtype=torch.bool)
        toks = input_ids[b][keep].tolist()
        vtok = [toks[i + 1] for i, t in enumerate(toks) if t == ids.vision_start]
        r_aud = sum(1 for t in toks if t == ids.audio_start)
        r_img = sum(1 for t in vtok if t == ids.image)
        r_vid = sum(1 for t in vtok if t == (ids.audio_start if use_audio_in_video else ids.video))
        parts: List[torch.Tensor] = []
        nxt = lambda: int(parts[-1].max()) + 1 if parts else 0
        span = lambda n: torch.arange(n).view(1, -1).expand(3, -1) + nxt()
        a_len = lambda k: ((int(audio_seqlens[k]) - 1) // 2 + 1 - 2) // 2 + 1
        v_t = lambda k: (torch.arange(vid[k][0]) * float(second_per_grids[k]) * ids.position_id_per_seconds).long().tolist()
        st = 0
        for _ in range(r_img + r_aud if use_audio_in_video else r_img + r_vid + r_aud):
            big = len(toks) + 1
            e_img = toks.index(ids.image, st) if (ids.image in toks and r_img > 0) else big
            e_vid = toks.index(ids.video, st) if (ids.video in toks and r_vid > 0) else big
            e_aud = toks.index(ids.audio, st) if (ids.audio in toks and r_aud > 0) else big
            m = min(e_img, e_vid, e_aud)
            two = (m == e_vid and m != e_aud and m != e_img and use_audio_in_video)
            tl = m - st - (2 if two else 1)
            if tl:
                parts.append(span(tl))
            if m == e_aud:
                parts.append(span(1)); parts.append(span(a_len(ai))); parts.append(span(1))
                st += tl + 2 + a_len(ai)
                ai += 1; r_aud -= 1
            elif m == e_img:
                t, h, w = img[ii]
                parts.append(span(1))
                parts.append(_vision_pos(nxt(), [i * ids.position_id_per_seconds for i in range(t)], h // merge, w // merge))
                parts.append(span(1))
                st += tl + 2 + t * h * w // (merge * merge)
                ii += 1; r_img -= 1
            elif not use_audio_in_video:
                t, h, w = vid[vi]
                parts.append(span(1))
                parts.append(_vision_pos(nxt(), v_t(vi), h // merge, w // merge))
                parts.append(span(1))
                st += tl + 2 + t * h * w // (merge * merge)
                vi += 1; r_vid -= 1
            else:
                t, h, w = vid[vi]
                bos = span(1)
                parts.append(bos); parts.append(bos.clone())
                s1 = nxt()
                apos = torch.arange(a_len(ai)).view(1, -1).expand(3, -1) + s1
                vpos = _vision_pos(s1, v_t(vi), h // merge, w // merge)
                per = int(ids.position_id_per_seconds * ids.seconds_per_chunk)
                vc, ac = _chunk_bounds(vpos[0], per, s1), _chunk_bounds(apos[0], per, s1)
                for j in range(max(len(vc), len(ac))):
                    if j < len(vc):
                        parts.append(vpos[:, vc[j][0]:vc[j][1]])
                    if j < len(ac):
                        parts.append(apos[:, ac[j][0]:ac[j][1]])
                eos = span(1)
                parts.append(eos); parts.append(eos.clone())
                st += tl + 4 + a_len(ai) + t * h * w // (merge * merge)
                ai += 1; vi += 1; r_vid -= 1; r_aud -= 1
        if st < len(toks):
            parts.append(span(len(toks) - st))
        lp = torch.cat(parts, 1).reshape(3, -1)
        pos[:, b, keep] = lp
        deltas.append(int(lp.max()) + 1 - len(toks))
    return pos, torch.tensor(deltas).unsqueeze(1)


class QwenOmniThinker:
    """Text + image / audio / video in, text out: the `model.generate(**inputs)` call of qwen2.5omni_spider_web.py:468 for
    the thinker (the talker / speech output is not part of the Spider path: the demo keeps text_ids only, :470-472).
    `llm` is a LlamaEngine with cfg.mrope_section set; towers are optional (text-only prompts need none)."""

    def __init__(self, llm, vision: Optional[VisionTowerEngine] = None, audio: Optional[AudioTowerEngine] = None,
                 token_ids: Optional[OmniTokenIds] = None):
        self.llm, self.vision, self.audio = llm, vision, audio
        self.ids = token_ids or OmniTokenIds()
        self.merge = vision.cfg.merge if vision is not None else 2

    @classmethod
    def from_pretrained(cls, path: str, device="cuda:0", max_len: int = 4096, load_vision: bool = True, load_audio: bool = True,
                        max_batch: int = 4):
        """A Qwen2.5-Omni checkpoint directory (config.json + *.safetensors with `thinker.model.*`, `thinker.visual.*`,
        `thinker.audio_tower.*`): what `Qwen2_5OmniModel.from_pretrained(args.checkpoint_path)` reads
        (qwen2.5omni_spider_web.py:376-381), minus the talker / token2wav weights this path never uses.
        `max_batch` rows of KV cache are preallocated (left-padded processor batches, padding=True); at most 8 rows share a
        decode graph, more are processed in groups."""
        import json, os
        from .llm import LlamaEngine
        cfg = json.load(open(os.path.join(path, "config.json")))
        if max_batch < 1:
            raise ValueError("max_batch must be >= 1")
        return cls(LlamaEngine.from_pretrained(path, device, max_batch=max_batch, max_len=max_len),
                   VisionTowerEngine.from_pretrained(path, device) if load_vision else None,
                   AudioTowerEngine.from_pretrained(path, device) if load_audio else None, OmniTokenIds.from_hf_dict(cfg))

    def _splice(self, emb: torch.Tensor, input_ids: torch.Tensor, token: int, feats: torch.Tensor, what: str) -> None:
        mask = (input_ids == token).to(emb.device)
        n = int(mask.sum())
        if n != feats.shape[0]:
            raise ValueError(f"{what} features and {what} tokens do not match, tokens: {n}, features: {feats.shape[0]}")
        emb[mask] = feats.to(emb.dtype)                                            # masked_scatter: rows in order

    @torch.no_grad()
    def prepare_inputs(self, input_ids, attention_mask=None, pixel_values=None, image_grid_thw=None, pixel_values_videos=None,
                       video_grid_thw=None, input_features=None, feature_attention_mask=None, audio_feature_lengths=None,
                       use_audio_in_video=False, video_second_per_grid=None):
        """-> (inputs_embeds [B, S, H] bf16, position_ids [3, B, S] long). Same argument names as the processor emits."""
        dv = self.llm.device
        input_ids = input_ids.to(dv)
        emb = self.llm.embed_tokens(input_ids)
        aud_lens = None
        if input_features is not None:
            if self.audio is None:
                raise ValueError("input_features given but no audio tower loaded")
            if feature_attention_mask is not None:     # [B, mel, frames] padded -> [mel, total valid frames]
                fam = feature_attention_mask.bool()
                aud_lens = fam.sum(1).tolist()
                feats = input_features.permute(0, 2, 1)[fam].permute(1, 0)
            else:
                aud_lens = [int(v) for v in audio_feature_lengths]
                feats = input_features
            self._splice(emb, input_ids, self.ids.audio, self.audio(feats, aud_lens), "audio")
        if pixel_values is not None:
            if self.vision is None:
                raise ValueError("pixel_values given but no vision tower loaded")
            self._splice(emb, input_ids, self.ids.image, self.vision(pixel_values, image_grid_thw), "image")
        if pixel_values_videos is not None:
            if self.vision is None:
                raise ValueError("pixel_values_videos given but no vision tower loaded")
            self._splice(emb, input_ids, self.ids.video, self.vision(pixel_values_videos, video_grid_thw), "video")
        pos, _ = get_rope_index(self.ids, self.merge, input_ids, image_grid_thw, video_grid_thw, attention_mask,
                                use_audio_in_video, aud_lens, video_second_per_grid)
        return emb, pos

    # Qwen2_5OmniModel.generate's own default for the thinker (thinker_max_new_tokens=1024) and the chat terminators
    # (<|im_end|>, <|endoftext|>) of the Qwen2.5 vocabulary, used when the checkpoint's config files name none
    THINKER_MAX_NEW_TOKENS = 1024
    DEFAULT_EOS = (151645, 151643)

    @torch.no_grad()
    def generate(self, input_ids, attention_mask=None, max_new_tokens: Optional[int] = None, thinker_max_new_tokens: Optional[int] = None,
                 **kw):
        """Returns [B, S + new] token ids like GenerationMixin.generate with input_ids. Length and EOS default to the
        checkpoint's generation config as in the reference's bare `model.generate(**inputs, spk=..., use_audio_in_video=True)`
        (qwen2.5omni_spider_web.py:468): finished rows are pad-filled, the call ends when every row has emitted EOS.
        = `prefill_begin` + `decode_finish` back to back."""
        return self.decode_finish(self.prefill_begin(input_ids, attention_mask, max_new_tokens, thinker_max_new_tokens, **kw))

    @torch.no_grad()
    def prefill_begin(self, input_ids, attention_mask=None, max_new_tokens: Optional[int] = None,
                      thinker_max_new_tokens: Optional[int] = None, **kw):
        """First half of `generate`: towers, embedding splice, rotary positions and the prompt pass of the text decoder, enqueued on
        the current stream (LlamaEngine.prefill_begin; `cache_set` selects the KV cache the request lives in). -> handle for
        `decode_finish`."""
        gc = getattr(self.llm, "generation_config", None) or {}
        if max_new_tokens is None:
            max_new_tokens = (thinker_max_new_tokens or gc.get("thinker_max_new_tokens") or gc.get("max_new_tokens")
                              or self.THINKER_MAX_NEW_TOKENS)
            max_new_tokens = max(1, min(max_new_tokens, self.llm.max_len - input_ids.shape[1]))
        if kw.get("eos_token_id") is None:
            kw["eos_token_id"] = gc.get("thinker_eos_token_id") or gc.get("eos_token_id") or (
                list(self.DEFAULT_EOS) if self.llm.cfg.vocab > max(self.DEFAULT_EOS) else None)
        kw.setdefault("sync_every", 8)
        tower_keys = ("pixel_values", "image_grid_thw", "pixel_values_videos", "video_grid_thw", "input_features",
                      "feature_attention_mask", "audio_feature_lengths", "use_audio_in_video", "video_second_per_grid")
        emb, pos = self.prepare_inputs(input_ids, attention_mask, **{k: kw.pop(k) for k in tower_keys if k in kw})
        kw.pop("spk", None); kw.pop("return_audio", None)                          # talker options: no speech on this path
        h = self.llm.prefill_begin(inputs_embeds=emb, position_ids=pos, attention_mask=attention_mask, max_new_tokens=max_new_tokens, **kw)
        return (h, input_ids)

    def would_capture(self, input_ids, attention_mask=None, cache_set: int = 0, pixel_values=None, image_grid_thw=None,
                      pixel_values_videos=None, video_grid_thw=None, input_features=None, feature_attention_mask=None,
                      audio_feature_lengths=None, output_hidden_states: bool = False, return_logits: bool = False, decode: bool = True, **_):
        """Would `generate(**inputs)` capture a hipGraph -- a tower graph for these grid_thw VALUES / audio lengths, or the decode step
        for this row count and KV cache set? Answered from the engines' own caches, so evictions and resets count. A pass that
        captures must not run beside another host thread that enqueues or allocates (SpiderFreeInfer runs it alone)."""
        if input_features is not None and self.audio is not None:
            lens = feature_attention_mask.bool().sum(1).tolist() if feature_attention_mask is not None else audio_feature_lengths
            if not self.audio.has_graph(lens):
                return True
        for pv, grid in ((pixel_values, image_grid_thw), (pixel_values_videos, video_grid_thw)):
            if pv is not None and self.vision is not None and not self.vision.has_graph(grid):
                return True
        B = int(input_ids.shape[0])
        if B > 8:          # grouped generate inside prefill_begin: graphs per group size
            return True
        if not decode:     # the prompt pass alone (SpiderFreeInfer depth 3): only the towers capture
            return False
        return self.llm.would_capture(B, output_hidden_states, return_logits, cache_set)

    @torch.no_grad()
    def adopt(self, handle, cache_set: int = 0):
        """move a prefilled request into KV cache set `cache_set` (LlamaEngine.adopt: device copies on the current stream)"""
        h, input_ids = handle
        return (self.llm.adopt(h, cache_set) if hasattr(h, "st") else h, input_ids)

    @torch.no_grad()
    def decode_finish(self, handle):
        """Second half of `generate`: the decode loop; prepends the prompt ids like GenerationMixin does."""
        h, input_ids = handle
        out = self.llm.decode_finish(h) if hasattr(h, "st") else h          # (> 8 rows: prefill_begin already ran the grouped generate)
        if isinstance(out, torch.Tensor):
            return torch.cat([input_ids.to(out.device).long(), out], 1)
        out.sequences = torch.cat([input_ids.to(out.sequences.device).long(), out.sequences], 1)
        return out
