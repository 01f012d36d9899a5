"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the multimodal INPUT side of the
Qwen2.5-Omni thinker that SpiderFree drives.

Reference call path: qwen2.5omni_spider_web.py:461-468 --
    inputs = processor(text=text, audios=audios, images=images, videos=videos, ...)
    text_ids, audio = model.generate(**inputs, spk=voice, use_audio_in_video=True)
with `model = Qwen2_5OmniModel.from_pretrained(...)` (:376-381; also qwen2.5omni_infer.py:3). The arithmetic is not in
/root/reference: it lives in the third-party dependency `transformers` (reference pins 4.50.0,
requirements_qwen2.5omni.txt:5; this image has 5.15.0, file models/qwen2_5_omni/modeling_qwen2_5_omni.py). Restated here:

  * vision tower  -- Qwen2_5OmniVisionEncoder.forward: Conv3d patch embedding (kernel = stride, i.e. a linear map of the
    flattened 3 x 2 x 14 x 14 patch), window re-ordering in units of 2x2 merge groups, 2-D rotary embedding on (h, w)
    patch coordinates (apply_rotary_pos_emb_vision: fp32, half-rotation), RMSNorm -> attention over cu_seqlens
    segments (windows, or whole frames at fullatt_block_indexes) -> RMSNorm -> SwiGLU MLP with biases, then the
    patch merger (RMSNorm, 4 tokens concatenated, Linear-GELU-Linear) and the inverse window permutation;
  * audio tower   -- Qwen2_5OmniAudioEncoder.forward: chunks of 2*n_window mel frames, Conv1d(k3,p1)+GELU, mask,
    Conv1d(k3,s2,p1)+GELU, sinusoid positions, pre-LN Whisper layers attending inside each chunk, stride-2 average
    pooling of consecutive frames per audio, LayerNorm, projection to the LLM width;
  * get_rope_index -- the 3-component (t, h, w) rotary positions of a prompt with image / audio / video placeholders;
  * the masked_scatter splice of tower outputs into the token embeddings
    (Qwen2_5OmniThinkerForConditionalGeneration.forward).

PINNED: tests/golden/make_golden_towers.py runs the transformers classes themselves on tiny seeded configs in this
container and commits weights + inputs + outputs (tests/golden/qwen_towers_ref.npz); tests/test_oracle_golden.py checks
this file against them. Weight names are the transformers state-dict names relative to `thinker.visual.` /
`thinker.audio_tower.`, so a real checkpoint loads into either side.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------- configs
@dataclass
class VisionCfg:
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
    def tiny():
        # head_dim 32 (16 rotary pairs: 8 for h, 8 for w); window = 2 merge groups = 4 patches per side
        return VisionCfg(depth=3, hidden=64, heads=2, inter=88, in_channels=3, patch=4, temporal_patch=2, merge=2, window=16,
                         out_hidden=96, fullatt=(1,), eps=1e-6)

    @property
    def patch_dim(self):
        return self.in_channels * self.temporal_patch * self.patch * self.patch


@dataclass
class AudioCfg:
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
    def tiny():
        return AudioCfg(mel=16, layers=2, heads=2, ffn=96, d_model=32, max_pos=40, n_window=10, out_dim=48)


def vision_param_shapes(c: VisionCfg) -> Dict[str, Tuple[int, ...]]:
    S = {"patch_embed.proj.weight": (c.hidden, c.in_channels, c.temporal_patch, c.patch, c.patch)}
    for l in range(c.depth):
        b = f"blocks.{l}."
        S[b + "norm1.weight"] = (c.hidden,); S[b + "norm2.weight"] = (c.hidden,)
        for n in ("q", "k", "v", "proj"):
            S[b + f"attn.{n}.weight"] = (c.hidden, c.hidden); S[b + f"attn.{n}.bias"] = (c.hidden,)
        for n, (o, i) in {"gate_proj": (c.inter, c.hidden), "up_proj": (c.inter, c.hidden), "down_proj": (c.hidden, c.inter)}.items():
            S[b + f"mlp.{n}.weight"] = (o, i); S[b + f"mlp.{n}.bias"] = (o,)
    m = c.hidden * c.merge * c.merge
    S["merger.ln_q.weight"] = (c.hidden,)
    S["merger.mlp.0.weight"] = (m, m); S["merger.mlp.0.bias"] = (m,)
    S["merger.mlp.2.weight"] = (c.out_hidden, m); S["merger.mlp.2.bias"] = (c.out_hidden,)
    return S


def audio_param_shapes(c: AudioCfg) -> Dict[str, Tuple[int, ...]]:
    S = {"conv1.weight": (c.d_model, c.mel, 3), "conv1.bias": (c.d_model,),
         "conv2.weight": (c.d_model, c.d_model, 3), "conv2.bias": (c.d_model,),
         "audio_bos_eos_token.weight": (2, c.out_dim)}
    for l in range(c.layers):
        b = f"layers.{l}."
        S[b + "self_attn.k_proj.weight"] = (c.d_model, c.d_model)          # no bias (Whisper)
        for n in ("v_proj", "q_proj", "out_proj"):
            S[b + f"self_attn.{n}.weight"] = (c.d_model, c.d_model); S[b + f"self_attn.{n}.bias"] = (c.d_model,)
        S[b + "self_attn_layer_norm.weight"] = (c.d_model,); S[b + "self_attn_layer_norm.bias"] = (c.d_model,)
        S[b + "fc1.weight"] = (c.ffn, c.d_model); S[b + "fc1.bias"] = (c.ffn,)
        S[b + "fc2.weight"] = (c.d_model, c.ffn); S[b + "fc2.bias"] = (c.d_model,)
        S[b + "final_layer_norm.weight"] = (c.d_model,); S[b + "final_layer_norm.bias"] = (c.d_model,)
    S["ln_post.weight"] = (c.d_model,); S["ln_post.bias"] = (c.d_model,)
    S["proj.weight"] = (c.out_dim, c.d_model); S["proj.bias"] = (c.out_dim,)
    return S


def random_weights(shapes: Dict[str, Tuple[int, ...]], seed=0, bf16_round=True) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    w = {}
    for n, shp in shapes.items():
        if n.endswith(".bias"):
            t = torch.randn(shp, generator=g) * 0.05
        elif "norm" in n or n.startswith("ln_") or ".ln_q." in n:
            t = 1.0 + torch.randn(shp, generator=g) * 0.1
        elif "bos_eos" in n:
            t = torch.randn(shp, generator=g) * 0.5
        else:
            t = torch.randn(shp, generator=g) / math.sqrt(math.prod(shp[1:]))
        w[n] = t.bfloat16().float() if bf16_round else t
    return w


# ---------------------------------------------------------------------------------------------- host index logic
def vision_position_ids(grid_thw: Sequence[Sequence[int]], merge: int) -> torch.Tensor:
    """transformers.vision_utils.get_vision_position_ids: (h, w) coordinates of every patch, laid out block-major over
    merge x merge groups (the order the processor emits patches in), repeated t times. -> [tokens, 2] long."""
    out = []
    for t, h, w in grid_thw:
        hp = torch.arange(h)[:, None].expand(h, w)
        wp = torch.arange(w)[None, :].expand(h, w)
        blk = (h // merge, merge, w // merge, merge)
        hp = hp.reshape(blk).transpose(1, 2).flatten()
        wp = wp.reshape(blk).transpose(1, 2).flatten()
        out.append(torch.stack([hp, wp], -1).repeat(t, 1))
    return torch.cat(out, 0)


def vision_window_index(grid_thw, merge: int, window: int, patch: int) -> Tuple[torch.Tensor, List[int]]:
    """transformers.vision_utils.get_vision_window_index: permutation of the merge groups into window-major order and
    the cumulative window boundaries (in patches). Windows are (window // merge // patch)^2 groups; edge windows are
    smaller; a grid that divides exactly still gets a padded (empty) extra window row / column, removed by
    unique_consecutive."""
    win = window // merge // patch
    unit = merge * merge
    idx_all, cu = [], [0]
    base = 0
    for t, h, w in grid_thw:
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


def vision_cu_seqlens(grid_thw) -> List[int]:
    """get_vision_cu_seqlens(merge_temporal=False): every frame is one full-attention segment of h*w patches."""
    cu = [0]
    for t, h, w in grid_thw:
        for _ in range(t):
            cu.append(cu[-1] + h * w)
    return cu


def vision_rope_table(c: VisionCfg, pos_hw: torch.Tensor) -> torch.Tensor:
    """Qwen2_5_VisionRotaryEmbedding(head_dim // 2) on [tokens, 2] positions -> angles [tokens, head_dim // 2]:
    the first quarter of the head dim follows h, the second quarter w."""
    d = c.hidden // c.heads
    dim = d // 2
    inv = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
    return (pos_hw.unsqueeze(-1).float() * inv).flatten(1)


def audio_chunk_lengths(feature_lens: Sequence[int], n_window: int) -> List[int]:
    """chunk_and_pad_features: every audio is cut into chunks of 2*n_window mel frames, the last one shorter."""
    out = []
    for L in feature_lens:
        n = -(-L // (2 * n_window))
        tail = L % (2 * n_window)
        out += [2 * n_window] * (n - 1) + [tail if tail else 2 * n_window]
    return out


def sinusoids(length: int, channels: int, max_timescale=10000.0) -> torch.Tensor:
    """SinusoidsPositionEmbedding: [sin | cos] of arange(length) x exp(-log(ts)/(channels/2-1) * arange(channels/2))."""
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2).float())
    st = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(st), torch.cos(st)], 1)


# ---------------------------------------------------------------------------------------------- towers
def _rmsnorm(x, w, eps):
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(v + eps))


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), -1)


def _seg_attention(q, k, v, cu: Sequence[int], scale: float):
    """q/k/v [T, heads, d]; softmax inside each [cu[i], cu[i+1]) segment (eager_attention_forward per split)."""
    out = torch.empty_like(q)
    for a, b in zip(cu[:-1], cu[1:]):
        s = torch.einsum("qhd,khd->hqk", q[a:b], k[a:b]) * scale
        p = torch.softmax(s.float(), -1)
        out[a:b] = torch.einsum("hqk,khd->qhd", p, v[a:b])
    return out


def vision_forward(c: VisionCfg, w: Dict[str, torch.Tensor], pixel_values: torch.Tensor, grid_thw) -> Tuple[torch.Tensor, torch.Tensor]:
    """pixel_values [patches, C*Tp*P*P] (processor layout), grid_thw [[t, h, w], ...] (in patches).
    Returns (last_hidden_state [patches, hidden] in window order, pooler_output [patches / merge^2, out_hidden])."""
    grid = [tuple(int(v) for v in g) for g in (grid_thw.tolist() if hasattr(grid_thw, "tolist") else grid_thw)]
    unit = c.merge * c.merge
    H, nh = c.hidden, c.heads
    d = H // nh
    x = pixel_values.float() @ w["patch_embed.proj.weight"].reshape(H, -1).t()
    T = x.shape[0]
    win_idx, cu_win = vision_window_index(grid, c.merge, c.window, c.patch)
    cu_full = vision_cu_seqlens(grid)
    x = x.reshape(T // unit, unit, H)[win_idx].reshape(T, H)
    ang = vision_rope_table(c, vision_position_ids(grid, c.merge))
    ang = ang.reshape(T // unit, unit, -1)[win_idx].reshape(T, -1)
    cos = ang.cos().repeat(1, 2)[:, None, :]
    sin = ang.sin().repeat(1, 2)[:, None, :]
    for l in range(c.depth):
        b = f"blocks.{l}."
        h = _rmsnorm(x, w[b + "norm1.weight"], c.eps)
        q = (h @ w[b + "attn.q.weight"].t() + w[b + "attn.q.bias"]).reshape(T, nh, d)
        k = (h @ w[b + "attn.k.weight"].t() + w[b + "attn.k.bias"]).reshape(T, nh, d)
        v = (h @ w[b + "attn.v.weight"].t() + w[b + "attn.v.bias"]).reshape(T, nh, d)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        a = _seg_attention(q, k, v, cu_full if l in c.fullatt else cu_win, d ** -0.5).reshape(T, H)
        x = x + a @ w[b + "attn.proj.weight"].t() + w[b + "attn.proj.bias"]
        h = _rmsnorm(x, w[b + "norm2.weight"], c.eps)
        g = F.silu(h @ w[b + "mlp.gate_proj.weight"].t() + w[b + "mlp.gate_proj.bias"])
        u = h @ w[b + "mlp.up_proj.weight"].t() + w[b + "mlp.up_proj.bias"]
        x = x + (g * u) @ w[b + "mlp.down_proj.weight"].t() + w[b + "mlp.down_proj.bias"]
    m = _rmsnorm(x, w["merger.ln_q.weight"], c.eps).reshape(-1, H * unit)
    m = F.gelu(m @ w["merger.mlp.0.weight"].t() + w["merger.mlp.0.bias"])
    m = m @ w["merger.mlp.2.weight"].t() + w["merger.mlp.2.bias"]
    return x, m[torch.argsort(win_idx)]


def audio_forward(c: AudioCfg, w: Dict[str, torch.Tensor], input_features: torch.Tensor, feature_lens: Sequence[int]) -> torch.Tensor:
    """input_features [mel, total_frames] (all audios concatenated, as get_audio_features passes them),
    feature_lens per audio. Returns [sum(out_len), out_dim], out_len = ((L - 1) // 2 + 1 - 2) // 2 + 1 per audio."""
    feature_lens = [int(v) for v in feature_lens]
    D, nh = c.d_model, c.heads
    d = D // nh
    chunks = audio_chunk_lengths(feature_lens, c.n_window)
    pos = sinusoids(c.max_pos, D)
    hs, cu = [], [0]
    off = 0
    for L in chunks:   # a right-padded + masked chunk equals the chunk convolved on its own with zero padding
        f = input_features[:, off:off + L].float()[None]
        off += L
        e = F.gelu(F.conv1d(f, w["conv1.weight"], w["conv1.bias"], padding=1))
        e = F.gelu(F.conv1d(e, w["conv2.weight"], w["conv2.bias"], stride=2, padding=1))[0].t()
        hs.append(e + pos[: e.shape[0]])
        cu.append(cu[-1] + e.shape[0])
    x = torch.cat(hs, 0)
    T = x.shape[0]
    for l in range(c.layers):
        b = f"layers.{l}."
        h = F.layer_norm(x, (D,), w[b + "self_attn_layer_norm.weight"], w[b + "self_attn_layer_norm.bias"], c.eps)
        q = (h @ w[b + "self_attn.q_proj.weight"].t() + w[b + "self_attn.q_proj.bias"]).reshape(T, nh, d)
        k = (h @ w[b + "self_attn.k_proj.weight"].t()).reshape(T, nh, d)
        v = (h @ w[b + "self_attn.v_proj.weight"].t() + w[b + "self_attn.v_proj.bias"]).reshape(T, nh, d)
        a = _seg_attention(q, k, v, cu, d ** -0.5).reshape(T, D)
        x = x + a @ w[b + "self_attn.out_proj.weight"].t() + w[b + "self_attn.out_proj.bias"]
        h = F.layer_norm(x, (D,), w[b + "final_layer_norm.weight"], w[b + "final_layer_norm.bias"], c.eps)
        h = F.gelu(h @ w[b + "fc1.weight"].t() + w[b + "fc1.bias"])
        x = x + h @ w[b + "fc2.weight"].t() + w[b + "fc2.bias"]
    idx = audio_pool_indices(feature_lens)
    x = (x[idx] + x[idx + 1]) / 2
    x = F.layer_norm(x, (D,), w["ln_post.weight"], w["ln_post.bias"], c.eps)
    return x @ w["proj.weight"].t() + w["proj.bias"]


def audio_pool_indices(feature_lens: Sequence[int]) -> torch.Tensor:
    """get_pool_indices: first element of every stride-2 pair, per audio, in the concatenated post-CNN sequence."""
    out, off = [], 0
    for L in feature_lens:
        after = (L - 1) // 2 + 1
        n = (after - 2) // 2 + 1
        out += [off + 2 * i for i in range(n)]
        off += after
    return torch.tensor(out, dtype=torch.long)


def audio_output_lengths(feature_lens: Sequence[int]) -> List[int]:
    return [(((L - 1) // 2 + 1) - 2) // 2 + 1 for L in feature_lens]


# ---------------------------------------------------------------------------------------------- prompt assembly
@dataclass
class OmniTokenIds:
    image: int = 151655
    video: int = 151656
    audio: int = 151646
    vision_start: int = 151652
    audio_start: int = 151647
    position_id_per_seconds: int = 25
    seconds_per_chunk: int = 2


def _vision_pos(start: int, t_index: List[int], gh: int, gw: int) -> torch.Tensor:
    n = len(t_index)
    hi = torch.arange(gh).view(1, -1, 1).expand(n, -1, gw).flatten()
    wi = torch.arange(gw).view(1, 1, -1).expand(n, gh, -1).flatten()
    ti = torch.tensor(t_index, dtype=torch.long).view(-1, 1).expand(-1, gh * gw).flatten()
    return torch.stack([ti, hi, wi]) + start


def _chunked(tok: torch.Tensor, per_chunk: int, remove: int) -> List[Tuple[int, int]]:
    out, start, cur = [], 0, 1
    for i in range(len(tok)):
        if int(tok[i]) - remove >= cur * per_chunk:
            out.append((start, i)); start = i; cur += 1
    out.append((start, len(tok)))
    return out


def get_rope_index(ids: OmniTokenIds, merge: int, input_ids: torch.Tensor, image_grid_thw=None, video_grid_thw=None,
                   attention_mask: Optional[torch.Tensor] = None, use_audio_in_video: bool = False,
                   audio_seqlens: Optional[Sequence[int]] = None, second_per_grids: Optional[Sequence[float]] = None):
    """Qwen2_5OmniPreTrainedModelForConditionalGeneration.get_rope_index. -> (position_ids [3, B, S] long, deltas [B, 1])."""
    B, S = input_ids.shape
    if image_grid_thw is None and video_grid_thw is None:
        am = attention_mask if attention_mask is not None else torch.ones_like(input_ids)
        p = am.long().cumsum(-1) - 1
        p = p.masked_fill(am == 0, 1)
        pos = p.unsqueeze(0).expand(3, -1, -1).clone()
        mx = pos.max(0)[0].max(-1, keepdim=True)[0]
        return pos, mx + 1 - am.sum(-1, keepdim=True)
    img = [tuple(int(v) for v in g) for g in (image_grid_thw.tolist() if hasattr(image_grid_thw, "tolist") else (image_grid_thw or []))]
    vid = [tuple(int(v) for v in g) for g in (video_grid_thw.tolist() if hasattr(video_grid_thw, "tolist") else (video_grid_thw or []))]
    pos = torch.ones(3, B, S, dtype=torch.long)
    deltas = []
    ii = vi = ai = 0
    for b in range(B):
        row = input_ids[b]
        keep = attention_mask[b] == 1 if attention_mask is not None else torch.ones(S, dtype=torch.bool)
        toks = row[keep].tolist()
        vs = [i for i, t in enumerate(toks) if t == ids.vision_start]
        vtok = [toks[i + 1] for i in vs]
        n_aud = sum(1 for t in toks if t == ids.audio_start)
        n_img = sum(1 for t in vtok if t == ids.image)
        n_vid = sum(1 for t in vtok if t == (ids.audio_start if use_audio_in_video else ids.video))
        parts: List[torch.Tensor] = []
        nxt = lambda: int(parts[-1].max()) + 1 if parts else 0
        span = lambda n: torch.arange(n).view(1, -1).expand(3, -1) + nxt()
        st = 0
        r_img, r_vid, r_aud = n_img, n_vid, n_aud
        for _ in range(n_img + n_aud if use_audio_in_video else n_img + n_vid + n_aud):
            big = len(toks) + 1
            e_img = toks.index(ids.image, st) if (ids.image in toks and r_img > 0) else big
            e_vid = toks.index(ids.video, st) if (ids.video in toks and r_vid > 0) else big
            e_aud = toks.index(ids.audio, st) if (ids.audio in toks and r_aud > 0) else big
            m = min(e_img, e_vid, e_aud)
            if m == e_aud:
                tl = m - st - 1
                if tl:
                    parts.append(span(tl))
                parts.append(span(1))
                al = ((int(audio_seqlens[ai]) - 1) // 2 + 1 - 2) // 2 + 1
                parts.append(span(al))
                parts.append(span(1))
                st += tl + 1 + al + 1
                ai += 1; r_aud -= 1
            elif m == e_img:
                tl = m - st - 1
                if tl:
                    parts.append(span(tl))
                parts.append(span(1))
                t, h, w = img[ii]
                parts.append(_vision_pos(nxt(), [i * ids.position_id_per_seconds for i in range(t)], h // merge, w // merge))
                parts.append(span(1))
                st += tl + 1 + t * h * w // (merge * merge) + 1
                ii += 1; r_img -= 1
            elif m == e_vid and not use_audio_in_video:
                tl = m - st - 1
                if tl:
                    parts.append(span(tl))
                parts.append(span(1))
                t, h, w = vid[vi]
                ti = (torch.arange(t) * float(second_per_grids[vi]) * ids.position_id_per_seconds).long().tolist()
                parts.append(_vision_pos(nxt(), ti, h // merge, w // merge))
                parts.append(span(1))
                st += tl + 1 + t * h * w // (merge * merge) + 1
                vi += 1; r_vid -= 1
            else:   # video with its audio track interleaved in chunks of seconds_per_chunk
                tl = m - st - 2
                if tl:
                    parts.append(span(tl))
                s0 = nxt()
                bos = torch.arange(1).view(1, -1).expand(3, -1) + s0
                parts.append(bos); parts.append(bos.clone())
                s1 = nxt()
                al = ((int(audio_seqlens[ai]) - 1) // 2 + 1 - 2) // 2 + 1
                apos = torch.arange(al).view(1, -1).expand(3, -1) + s1
                t, h, w = vid[vi]
                ti = (torch.arange(t) * float(second_per_grids[vi]) * ids.position_id_per_seconds).long().tolist()
                vpos = _vision_pos(s1, ti, h // merge, w // merge)
                per = int(ids.position_id_per_seconds * ids.seconds_per_chunk)
                vc, ac = _chunked(vpos[0], per, s1), _chunked(apos[0], per, s1)
                for j in range(max(len(vc), len(ac))):
                    if j < len(vc):
                        parts.append(vpos[:, vc[j][0]:vc[j][1]])
                    if j < len(ac):
                        parts.append(apos[:, ac[j][0]:ac[j][1]])
                s2 = nxt()
                eos = torch.arange(1).view(1, -1).expand(3, -1) + s2
                parts.append(eos); parts.append(eos.clone())
                st += tl + 2 + al + t * h * w // (merge * merge) + 2
                ai += 1; vi += 1; r_vid -= 1; r_aud -= 1
        if st < len(toks):
            parts.append(span(len(toks) - st))
        lp = torch.cat(parts, 1).reshape(3, -1)
        pos[:, b, keep] = lp
        deltas.append(int(lp.max()) + 1 - len(toks))
    return pos, torch.tensor(deltas).unsqueeze(1)


def splice_features(ids: OmniTokenIds, input_ids: torch.Tensor, inputs_embeds: torch.Tensor, audio_features=None,
                    image_embeds=None, video_embeds=None) -> torch.Tensor:
    """The masked_scatter merges of Qwen2_5OmniThinkerForConditionalGeneration.forward: placeholder positions take the
    tower rows in order (audio, then image, then video)."""
    out = inputs_embeds.clone()
    for tok, feat in ((ids.audio, audio_features), (ids.image, image_embeds), (ids.video, video_embeds)):
        if feat is None:
            continue
        mask = input_ids == tok
        if int(mask.sum()) != feat.shape[0]:
            raise ValueError(f"features and placeholder tokens do not match, tokens: {int(mask.sum())}, features: {feat.shape[0]}")
        out[mask] = feat.to(out.dtype)
    return out
