"""Native LLM engine: greedy autoregressive decode on the HIP kernels (no transformers / torch math).

Drop-in for the reference's LLM seam B3:
    llama_model.generate(inputs_embeds | input_ids, attention_mask, max_new_tokens, num_beams=1,
                         do_sample=False, use_cache=True, stopping_criteria, output_hidden_states,
                         return_dict_in_generate, output_attentions)      spider/models/spider.py:1492-1508
    -> .sequences  [B, T_new] (generated tokens only for an inputs_embeds call, prompt + generated for
                   input_ids -- HF semantics the reference relies on, spider.py:1428-1449)
    -> .hidden_states[step][layer]  [B, S|1, H]
    embed_tokens(ids)                                                     spider/models/base_model.py:253-258
Arithmetic follows spider/models/modeling_llama3.py:68-313 (Llama-3 GQA + rope scaling) and the Qwen2.5
text decoder (qkv bias) that qwen2.5omni_spider_web.py:468 drives.

Data layout in HBM (per engine):
    embed  [V, H] bf16 | per layer: w_qkv [(n_q+2n_kv)d, H], b_qkv?, w_o [H, n_q d], w_gate_up [2I, H],
    w_down [H, I], ln1 [H], ln2 [H] | norm [H] | lm_head [V, H]
    KV cache: 2 x [L, B_max, n_kv, T_max, d] bf16, preallocated once; rope table [max_pos, d] fp32.
Decode = 5 weight-streaming launches per layer (+1 tiny split-KV combine), captured in a hipGraph.
"""
from __future__ import annotations

import os
import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import torch

from . import ops

BF16 = torch.bfloat16


@dataclass
class LLMConfig:
    hidden: int
    layers: int
    n_q: int
    n_kv: int
    head_dim: int
    inter: int
    vocab: int
    rope_theta: float = 10000.0
    rope_scaling: Optional[dict] = None
    eps: float = 1e-6
    qkv_bias: bool = False
    max_pos: int = 8192
    tie_embeddings: bool = False
    mrope_section: Optional[tuple] = None   # Qwen2.5-Omni thinker: (16, 24, 24) rotary pairs follow (t, h, w) positions

    @staticmethod
    def llama3_8b():   # DeepSeek-R1-Distill-Llama-8B (r1_llama3_8B_infer.py:4, demo/inference_api.py:92-95)
        return LLMConfig(4096, 32, 32, 8, 128, 14336, 128256, 500000.0,
                         dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                              original_max_position_embeddings=8192), 1e-5, False, 8192)

    @staticmethod
    def qwen25_7b():   # Qwen2.5-Omni-7B thinker text decoder (qwen2.5omni_spider_web.py:368-384)
        return LLMConfig(3584, 28, 28, 4, 128, 18944, 152064, 1000000.0, None, 1e-6, True, 8192, False, (16, 24, 24))

    @staticmethod
    def from_hf_dict(c: dict) -> "LLMConfig":
        if "thinker_config" in c:  # Qwen2.5-Omni nests the text decoder config
            c = c["thinker_config"].get("text_config", c["thinker_config"])
        hd = c.get("head_dim") or c["hidden_size"] // c["num_attention_heads"]
        return LLMConfig(c["hidden_size"], c["num_hidden_layers"], c["num_attention_heads"],
                         c.get("num_key_value_heads", c["num_attention_heads"]), hd, c["intermediate_size"],
                         c["vocab_size"], float(c.get("rope_theta", 10000.0)), c.get("rope_scaling"),
                         float(c.get("rms_norm_eps", 1e-6)),
                         bool(c.get("attention_bias", c.get("model_type", "").startswith("qwen"))),
                         int(c.get("max_position_embeddings", 8192)), bool(c.get("tie_word_embeddings", False)),
                         tuple((c.get("rope_scaling") or {}).get("mrope_section") or ()) or None)


def rope_inv_freq(cfg: LLMConfig) -> torch.Tensor:
    """fp32 inverse frequencies incl. llama3 scaling (modeling_llama3.py:91-113 -> ROPE_INIT_FUNCTIONS)."""
    d = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
    rs = cfg.rope_scaling
    if rs and rs.get("rope_type", rs.get("type")) == "llama3":
        factor, lo, hi = rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"]
        old = rs["original_max_position_embeddings"]
        wl = 2 * math.pi / inv
        inv_l = torch.where(wl > old / lo, inv / factor, inv)
        smooth = (old / wl - lo) / (hi - lo)
        smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
        mid = ~(wl < old / hi) * ~(wl > old / lo)
        inv = torch.where(mid, smoothed, inv_l)
    return inv


def rope_table(cfg: LLMConfig, n_pos: int) -> torch.Tensor:
    fr = torch.outer(torch.arange(n_pos, dtype=torch.float32), rope_inv_freq(cfg))
    return torch.cat([fr.cos(), fr.sin()], dim=-1).contiguous()


def read_generation_config(path: str) -> dict:
    """The defaults HF `generate` takes from the checkpoint when the caller passes none (the reference's
    `model.generate(**inputs, spk=..., use_audio_in_video=True)`, qwen2.5omni_spider_web.py:468, and
    `r1_llama3_8B_chat.py:14` rely on them): eos_token_id (int or list), pad_token_id, max_new_tokens / max_length,
    and Qwen2.5-Omni's thinker_max_new_tokens. generation_config.json wins over config.json, as in transformers."""
    import json
    gc: dict = {}
    for name in ("config.json", "generation_config.json"):
        f = os.path.join(path, name)
        if not os.path.exists(f):
            continue
        d = json.load(open(f))
        srcs = [d]
        if name == "config.json" and "thinker_config" in d:
            srcs += [d["thinker_config"], d["thinker_config"].get("text_config", {})]
        for src in srcs:
            for k in ("eos_token_id", "pad_token_id", "bos_token_id", "max_new_tokens", "max_length",
                      "thinker_max_new_tokens", "thinker_eos_token_id"):
                if src.get(k) is not None:
                    gc[k] = src[k]
    return gc


def _id_list(v) -> Optional[List[int]]:
    if v is None:
        return None
    return [int(v)] if isinstance(v, int) else [int(x) for x in v]


def finalize_greedy(tokens: torch.Tensor, eos: Optional[List[int]], pad: Optional[int], stopping_criteria,
                    prompt: Optional[torch.Tensor], checked: int = 0):
    """HF greedy-search bookkeeping (`_sample` with do_sample=False, as driven by spider.py:1492-1508), applied to a
    block of already generated tokens [B, n] (CPU int64):
      * a row is finished after its first EOS token; every later token of that row is pad_token_id
        (`next_tokens * unfinished + pad * (1 - unfinished)`);
      * the loop ends at the first length k where every row is finished, or where a stopping criterion returns True
        (StoppingCriteriaSub, spider.py:55-73, looks at sequence 0 and ends the whole batch);
    Returns (tokens with pads applied, k, stopped). `checked` = lengths < checked were already examined (the criteria
    are re-run only on the new prefixes, so `sync_every` > 1 finds the exact stop step after the fact)."""
    B, n = tokens.shape
    tk = tokens.clone()
    if eos:
        is_eos = torch.isin(tk, torch.tensor(eos))
        seen = is_eos.long().cumsum(1)
        after = (seen - is_eos.long()) > 0          # strictly after the row's first EOS
        if pad is None:
            pad = eos[0]                            # HF: "Setting pad_token_id to eos_token_id"
        tk[after] = pad
        done_at = torch.where(is_eos.any(1), is_eos.long().argmax(1) + 1, torch.full((B,), n + 1))
        k_eos = int(done_at.max())                  # all rows finished once the slowest has emitted its EOS
    else:
        k_eos = n + 1
    k_sc = n + 1
    if stopping_criteria:
        for k in range(max(1, checked), min(n, k_eos) + 1):
            seq = tk[:, :k] if prompt is None else torch.cat([prompt, tk[:, :k]], 1)
            if any(bool(torch.as_tensor(sc(seq, None)).all()) for sc in stopping_criteria):
                k_sc = k
                break
    k = min(k_eos, k_sc)
    if k <= n:
        return tk[:, :k], k, True
    return tk, n, False


_FUSE_SWIGLU = os.environ.get("SPIDER_PREFILL_SWIGLU_FUSE", "1") != "0"     # tuning aid: 0 = separate SwiGLU launch after the gate/up GEMM


class _PrefillHandle:
    """what LlamaEngine.prefill_begin hands to decode_finish (the locals of `generate` at its half-way point)"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class GenerateOutput:
    def __init__(self, sequences, hidden_states=None):
        self.sequences = sequences
        self.hidden_states = hidden_states

    def __getitem__(self, k):
        return getattr(self, k)


class StoppingCriteriaSub:
    """spider/models/spider.py:55-73: stop when the tail of sequence 0 equals any stop-id list."""

    def __init__(self, stops: Sequence[Sequence[int]] = ()):
        self.stops = [list(s) for s in stops]

    def __call__(self, input_ids: torch.Tensor, scores=None) -> bool:
        row = input_ids[0].tolist()
        for s in self.stops:
            if len(row) >= len(s) and row[len(row) - len(s):] == s:
                return True
        return False


class LlamaEngine:
    DECODE_ROWS = 8     # sequences per decode graph (lm_head / split-KV workspaces are sized for 8)
    # From this many sequences on the decode GEMVs run as skinny MFMA GEMMs on fragment-major weight copies (SPIDER_DECODE_FM_MIN).
    # Measured on MI355X, Qwen2.5-7B shapes at context 1536 (scripts/exp/decode_vs_batch.py), ms per decode step by rows 1 / 2 / 3 / 4 / 5 / 8:
    # row-major GEMVs 2.84 / 3.28 / 3.75 / 4.36 (then fragment-major 3.21 / 3.43); fragment-major from 2 rows: 2.99 / 3.09 / 3.16 / 3.23 / 3.43.
    FM_MIN_BATCH = int(os.environ.get("SPIDER_DECODE_FM_MIN", "2"))

    def __init__(self, cfg: LLMConfig, weights: dict, device="cuda:0", max_batch: int = 1, max_len: int = 4096):
        self.cfg, self.device = cfg, torch.device(device)
        self.max_batch, self.max_len = max_batch, max_len
        dv = self.device
        g = lambda k: weights[k].to(device=dv, dtype=BF16).contiguous()
        self.embed_w = g("model.embed_tokens.weight")
        self.lm_head = self.embed_w if (cfg.tie_embeddings or "lm_head.weight" not in weights) else g("lm_head.weight")
        self.norm = g("model.norm.weight")
        self.layers = []
        for l in range(cfg.layers):
            p = f"model.layers.{l}."
            lw = dict(
                w_qkv=torch.cat([g(p + "self_attn.q_proj.weight"), g(p + "self_attn.k_proj.weight"),
                                 g(p + "self_attn.v_proj.weight")], 0).contiguous(),
                b_qkv=(torch.cat([g(p + "self_attn.q_proj.bias"), g(p + "self_attn.k_proj.bias"),
                                  g(p + "self_attn.v_proj.bias")], 0).contiguous() if cfg.qkv_bias else None),
                w_o=g(p + "self_attn.o_proj.weight"),
                w_gu=torch.cat([g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight")], 0).contiguous(),
                w_down=g(p + "mlp.down_proj.weight"),
                ln1=g(p + "input_layernorm.weight"), ln2=g(p + "post_attention_layernorm.weight"))
            self.layers.append(lw)
        # Batched decode (5..8 sequences per graph) streams a second, fragment-major copy of every decode weight (ops.repack_fm16:
        # 1 KiB contiguous per wave instruction of the skinny MFMA kernels). Memory is laid out for 288 GB of HBM: the copy
        # costs one more model size (15 GB for the 7B / 8B decoders) and buys 1.4 - 1.6x on the batched weight streams.
        self.fm_batch = max_batch >= self.FM_MIN_BATCH and cfg.hidden % 64 == 0 and cfg.inter % 64 == 0 \
            and (cfg.n_q * cfg.head_dim) % 64 == 0 and os.environ.get("SPIDER_DECODE_FM", "1") != "0"
        if self.fm_batch:
            for lw in self.layers:     # the projections behind a LlamaRMSNorm carry its weight (fold_rmsnorm form of the kernels)
                lw["w_qkv_fm"] = ops.repack_fm16(lw["w_qkv"], lw["ln1"])
                lw["w_gu_fm"] = ops.repack_fm16(lw["w_gu"], lw["ln2"])
                lw["w_o_fm"] = ops.repack_fm16(lw["w_o"])
                lw["w_down_fm"] = ops.repack_fm16(lw["w_down"])
            self.lm_head_fm = ops.repack_fm16(self.lm_head, self.norm)
        self._alloc()

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def random_init(cls, cfg: LLMConfig, device="cuda:0", max_batch=1, max_len=4096, seed=0, std=0.02):
        """Random N(0, std^2) weights of the architecture's true shapes, created directly in HBM
        (init scheme of modeling_llama3.py:405-414). Used by bench.py: timing is weight-value independent."""
        gen = torch.Generator(device=device).manual_seed(seed)
        r = lambda *s: (torch.randn(*s, generator=gen, device=device, dtype=torch.float32) * std).to(BF16)
        one = lambda n: torch.ones(n, device=device, dtype=BF16)
        w = {"model.embed_tokens.weight": r(cfg.vocab, cfg.hidden), "model.norm.weight": one(cfg.hidden)}
        if not cfg.tie_embeddings:
            w["lm_head.weight"] = r(cfg.vocab, cfg.hidden)
        qd, kd = cfg.n_q * cfg.head_dim, cfg.n_kv * cfg.head_dim
        for l in range(cfg.layers):
            p = f"model.layers.{l}."
            w[p + "self_attn.q_proj.weight"] = r(qd, cfg.hidden)
            w[p + "self_attn.k_proj.weight"] = r(kd, cfg.hidden)
            w[p + "self_attn.v_proj.weight"] = r(kd, cfg.hidden)
            w[p + "self_attn.o_proj.weight"] = r(cfg.hidden, qd)
            if cfg.qkv_bias:
                w[p + "self_attn.q_proj.bias"] = r(qd)
                w[p + "self_attn.k_proj.bias"] = r(kd)
                w[p + "self_attn.v_proj.bias"] = r(kd)
            w[p + "mlp.gate_proj.weight"] = r(cfg.inter, cfg.hidden)
            w[p + "mlp.up_proj.weight"] = r(cfg.inter, cfg.hidden)
            w[p + "mlp.down_proj.weight"] = r(cfg.hidden, cfg.inter)
            w[p + "input_layernorm.weight"] = one(cfg.hidden)
            w[p + "post_attention_layernorm.weight"] = one(cfg.hidden)
        return cls(cfg, w, device, max_batch, max_len)

    @classmethod
    def from_pretrained(cls, path: str, device="cuda:0", max_batch=1, max_len=4096):
        """Load an HF checkpoint directory: config.json + model.safetensors(.index.json + shards) or pytorch_model.bin
        (spider_amd/checkpoint.py). From a Qwen2.5-Omni directory only `thinker.model.*` / `thinker.lm_head.*` are read."""
        from .checkpoint import load_state_dict, read_config

        def keep(k: str):
            kk = k.replace("thinker.model.", "model.").replace("thinker.lm_head.", "lm_head.")
            return kk if kk.startswith(("model.", "lm_head.")) else None
        cfg = LLMConfig.from_hf_dict(read_config(path))
        eng = cls(cfg, load_state_dict(path, keep=keep), device, max_batch, max_len)
        eng.generation_config = read_generation_config(path)
        return eng

    def _alloc(self):
        c, dv, B, T = self.cfg, self.device, self.max_batch, self.max_len
        self.k_cache = torch.zeros(c.layers, B, c.n_kv, T, c.head_dim, dtype=BF16, device=dv)
        self.v_cache = torch.zeros_like(self.k_cache)
        # KV cache sets: set 0 is the engine's cache; further sets (allocated on first use) let TWO requests be in flight at once --
        # the prefill of one on one stream while another decodes on a second stream (SpiderFreeInfer's three-stage pipelining)
        self._kv_sets = [(self.k_cache, self.v_cache)]
        self.cos_sin = rope_table(c, max(c.max_pos, T)).to(dv)
        self._graphs = {}

    def _kv(self, cache_set: int):
        while len(self._kv_sets) <= cache_set:
            self._kv_sets.append((torch.zeros_like(self.k_cache), torch.zeros_like(self.v_cache)))
        return self._kv_sets[cache_set]

    # ------------------------------------------------------------------ embeddings
    def embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        return ops.embed(self.embed_w, ids.to(device=self.device, dtype=torch.int32).contiguous())

    def resize_token_embeddings(self, new_num_tokens: Optional[int] = None, std: float = 0.02, seed: Optional[int] = None):
        """`PreTrainedModel.resize_token_embeddings` as the reference calls it after adding its signal tokens to the tokenizer
        (spider.py:177: `self.llama_model.resize_token_embeddings(len(self.llama_tokenizer))`): the input embedding and -- unless
        tied -- the lm_head grow (or shrink) to `new_num_tokens` rows. Old rows keep their values; new rows are N(0, std^2), the
        `_init_weights` rule of the pinned transformers 4.43 (modeling_llama3.py:405-414) -- a trained Spider checkpoint then
        overwrites them (`load_token_rows`). Captured decode graphs and the vocabulary-sized buffers are rebuilt on the next call.
        Returns the embedding matrix, like the HF method returns the embedding module."""
        c = self.cfg
        old = self.embed_w.shape[0]
        if new_num_tokens is None or new_num_tokens == old:
            return self.embed_w
        if new_num_tokens <= 0:
            raise ValueError(f"resize_token_embeddings: new_num_tokens={new_num_tokens}")
        gen = None
        if seed is not None:
            gen = torch.Generator(device=self.device).manual_seed(seed)

        def grow(w):
            out = torch.empty(new_num_tokens, w.shape[1], dtype=w.dtype, device=w.device)
            n = min(old, new_num_tokens)
            out[:n] = w[:n]
            if new_num_tokens > old:
                out[old:] = (torch.randn(new_num_tokens - old, w.shape[1], generator=gen, device=w.device, dtype=torch.float32) * std).to(w.dtype)
            return out.contiguous()
        tied = self.lm_head is self.embed_w
        self.embed_w = grow(self.embed_w)
        self.lm_head = self.embed_w if tied else grow(self.lm_head)
        import dataclasses
        self.cfg = dataclasses.replace(c, vocab=new_num_tokens)     # (the config object may be shared with other engines)
        self._vocab_changed()
        return self.embed_w

    def load_token_rows(self, first_row: int, embed_rows: Optional[torch.Tensor] = None, lm_head_rows: Optional[torch.Tensor] = None):
        """Overwrite rows [first_row, first_row + n) of the input embedding and / or the lm_head: the trained rows a Spider
        checkpoint stores for its signal tokens (the reference keeps `old_embed_tokens` / `old_lm_head` copies for exactly
        this split, spider.py:165-173)."""
        for w, rows, name in ((self.embed_w, embed_rows, "embed_rows"), (self.lm_head, lm_head_rows, "lm_head_rows")):
            if rows is None:
                continue
            if rows.ndim != 2 or rows.shape[1] != w.shape[1] or first_row < 0 or first_row + rows.shape[0] > w.shape[0]:
                raise ValueError(f"load_token_rows: {name} {tuple(rows.shape)} does not fit rows [{first_row}, ...) of {tuple(w.shape)}")
            w[first_row:first_row + rows.shape[0]] = rows.to(device=w.device, dtype=w.dtype)
        self._vocab_changed()

    def would_capture(self, B: int, output_hidden_states: bool = False, return_logits: bool = False, cache_set: int = 0) -> bool:
        """True when the decode loop of such a request would capture its hipGraph (state missing or not captured yet)"""
        ent = self._graphs.get((int(B), bool(output_hidden_states), bool(return_logits), int(cache_set)))
        return ent is None or ent[1] is None

    def _vocab_changed(self):
        """the lm_head moved or changed: drop what was derived from it (fragment-major copy, captured decode graphs, lm_head
        workspaces and logits buffers, all keyed in self._graphs)"""
        if self.fm_batch:
            self.lm_head_fm = ops.repack_fm16(self.lm_head, self.norm)
        self._graphs = {}

    # ------------------------------------------------------------------ prefill
    def _prefill(self, h: torch.Tensor, pos: torch.Tensor, slot: torch.Tensor, kv_beg: Optional[torch.Tensor],
                 B: int, S: int, hidden_out: Optional[list], mrope: bool = False, cache_set: int = 0):
        """h [B*S, H] bf16; pos [B*S] (or [3, B*S] with mrope); writes KV slots, returns the final residual stream [B*S, H]."""
        c = self.cfg
        k_cache, v_cache = self._kv(cache_set)
        sec = c.mrope_section if mrope else None
        if hidden_out is not None:
            hidden_out.append(h.view(B, S, -1).clone())
        for l, lw in enumerate(self.layers):
            x = ops.rmsnorm(h, lw["ln1"], c.eps)
            qkv = ops.gemm(x, lw["w_qkv"], bias=lw["b_qkv"])
            q = torch.empty(B, S, c.n_q, c.head_dim, dtype=BF16, device=self.device)
            ops.rope_kv_append(qkv, pos, slot, self.cos_sin, q, k_cache[l], v_cache[l], B, S, c.n_q, c.n_kv, c.head_dim,
                               mrope_section=sec)
            a = ops.attention_cache(q, k_cache[l], v_cache[l], Lk=S, causal=True, kv_off=0, kv_beg=kv_beg)
            h = ops.gemm(a.view(B * S, -1), lw["w_o"], res=h)
            x = ops.rmsnorm(h, lw["ln2"], c.eps)
            if _FUSE_SWIGLU:
                act = ops.gemm(x, lw["w_gu"], act="swiglu")  # gate / up projection with SwiGLU in the epilogue (no [S, 2I] round trip)
            else:
                act = ops.swiglu(ops.gemm(x, lw["w_gu"]))
            h = ops.gemm(act, lw["w_down"], res=h)
            if hidden_out is not None:
                hidden_out.append(h.view(B, S, -1).clone())
        return h

    # ------------------------------------------------------------------ one decode step (graph-capturable)
    def _decode_step(self, st: dict):
        c, B = self.cfg, st["B"]
        k_cache, v_cache = st["kv"]
        h = ops.embed(self.embed_w, st["cur_ids"]) if st["embeds_in"] is None else st["embeds_in"]
        hs = st.get("hidden_buf")
        if hs is not None:
            hs[0].copy_(h)
        fm = self.fm_batch and B >= self.FM_MIN_BATCH
        fuse_norm = not fm and B < 5      # the row-major GEMV folds the RMSNorm for up to 4 rows; the skinny MFMA GEMM (fm) carries it in its weights
        for l, lw in enumerate(self.layers):
            if fuse_norm:
                ops.gemv(lw["w_qkv"], h, bias=lw["b_qkv"], norm_w=lw["ln1"], eps=c.eps, out=st["qkv"])
            elif fm:
                ops.gemv_fm(lw["w_qkv_fm"], h, lw["w_qkv"].shape[0], bias=lw["b_qkv"], out=st["qkv"], norm_eps=c.eps)
            else:
                ops.gemv(lw["w_qkv"], ops.rmsnorm(h, lw["ln1"], c.eps, out=st["xn"]), bias=lw["b_qkv"], out=st["qkv"])
            if c.head_dim == 128:   # RoPE + KV append + split-KV attention + combine: one launch
                ops.attn_decode_fused(st["qkv"], st["pos"], self.cos_sin, k_cache[l], v_cache[l], st["kv_end"],
                                      st["kv_beg"], st["attn_cnt"], c.n_q, st["nsplit"], st["attn_ws"], st["attn"])
            else:
                ops.rope_kv_append(st["qkv"], st["pos"], st["slot"], self.cos_sin, st["q"], k_cache[l], v_cache[l],
                                   B, 1, c.n_q, c.n_kv, c.head_dim)
                ops.attn_decode(st["q"], k_cache[l], v_cache[l], st["kv_end"], kv_beg=st["kv_beg"],
                                nsplit=st["nsplit"], ws=st["attn_ws"], out=st["attn"])
            if fm:
                h1 = ops.gemv_fm(lw["w_o_fm"], st["attn"], c.hidden, res=h, out=st["h1"])
                ops.gemv_swiglu_fm(lw["w_gu_fm"], h1, out=st["act"], norm_eps=c.eps)
                h = ops.gemv_fm(lw["w_down_fm"], st["act"], c.hidden, res=h1, out=st["h2"][l & 1])
            else:
                h1 = ops.gemv(lw["w_o"], st["attn"], res=h, out=st["h1"])
                if fuse_norm:
                    ops.gemv_swiglu(lw["w_gu"], h1, norm_w=lw["ln2"], eps=c.eps, out=st["act"])
                else:
                    ops.gemv_swiglu(lw["w_gu"], ops.rmsnorm(h1, lw["ln2"], c.eps, out=st["xn"]), out=st["act"])
                h = ops.gemv(lw["w_down"], st["act"], res=h1, out=st["h2"][l & 1])
            if hs is not None:
                hs[l + 1].copy_(h)
        if fm:
            ops.lm_head_argmax_fm(self.lm_head_fm, h, c.vocab, out_ids=st["next_ids"], ws=st["lm_ws"], logits=st.get("logits"),
                                  norm_eps=c.eps)
        else:
            ops.lm_head_argmax(self.lm_head, h, norm_w=self.norm, eps=c.eps, out_ids=st["next_ids"], ws=st["lm_ws"],
                               logits=st.get("logits"))
        if hs is not None:  # HF reports the normed state as the last hidden state (modeling_llama3.py:619-623)
            ops.rmsnorm(h, self.norm, c.eps, out=hs[c.layers])
        # advance the device-side cursors and append the token to the on-device history (index math only, one launch)
        ops.decode_advance(st["next_ids"], st["cur_ids"], st["pos"], st["slot"], st["kv_end"], st["hist"], st["n_hist"])

    def _make_state(self, B: int, want_hidden: bool, want_logits: bool, cache_set: int = 0) -> dict:
        c, dv = self.cfg, self.device
        nq_d = c.n_q * c.head_dim
        # one split-KV block per CU (256): measured on Qwen-7B shapes at T~1.6k: 2.93 / 2.90 / 3.14 ms per token at 32 / 64 / 96 splits
        nsplit = max(1, min(64, 256 // max(1, B * c.n_kv)))
        if os.environ.get("SPIDER_ATTN_NSPLIT"):       # tuning aid
            nsplit = int(os.environ["SPIDER_ATTN_NSPLIT"])
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dv)
        bf = lambda *s: torch.empty(*s, dtype=BF16, device=dv)
        npart = ops.lm_head_nparts(c.vocab)
        st = dict(B=B, nsplit=nsplit, cur_ids=i32(B), next_ids=i32(B), pos=i32(B), slot=i32(B), kv_end=i32(B), kv_beg=i32(B),
                  qkv=bf(B, (c.n_q + 2 * c.n_kv) * c.head_dim), q=bf(B, c.n_q, c.head_dim), attn=bf(B, nq_d),
                  h1=bf(B, c.hidden), h2=[bf(B, c.hidden), bf(B, c.hidden)], act=bf(B, c.inter), xn=bf(B, c.hidden),
                  attn_ws=(torch.empty(B * c.n_q * nsplit * c.head_dim, dtype=torch.float32, device=dv),
                           torch.empty(B * c.n_q * nsplit * 2, dtype=torch.float32, device=dv)),
                  lm_ws=(torch.empty(B * npart, dtype=torch.float32, device=dv), i32(B * npart)),
                  attn_cnt=i32(B * c.n_kv), embeds_in=None, hist=i32(B, self.max_len), n_hist=i32(B), kv=self._kv(cache_set))
        if want_hidden:
            st["hidden_buf"] = bf(c.layers + 1, B, c.hidden)
        if want_logits:
            st["logits"] = bf(B, c.vocab)
        return st

    # ------------------------------------------------------------------ public generate
    @torch.no_grad()
    def generate(self, input_ids: Optional[torch.Tensor] = None, inputs_embeds: Optional[torch.Tensor] = None, **kw):
        """Greedy decode = `prefill_begin` + `decode_finish` back to back (arguments: see prefill_begin)."""
        h = self.prefill_begin(input_ids, inputs_embeds, **kw)
        return self.decode_finish(h) if isinstance(h, _PrefillHandle) else h

    @torch.no_grad()
    def prefill_begin(self, input_ids: Optional[torch.Tensor] = None, inputs_embeds: Optional[torch.Tensor] = None,
                      attention_mask: Optional[torch.Tensor] = None, max_new_tokens: Optional[int] = None,
                      stopping_criteria: Optional[Sequence[Callable]] = None, eos_token_id=None, pad_token_id=None,
                      output_hidden_states: bool = False, return_dict_in_generate: bool = False,
                      num_beams: int = 1, do_sample: bool = False, use_cache: bool = True, output_attentions: bool = False,
                      use_graph: bool = True, sync_every: int = 1, return_logits: bool = False,
                      position_ids: Optional[torch.Tensor] = None, cache_set: int = 0, **unused):
        """First half of `generate`: the prompt pass (KV cache of `cache_set` filled, first token chosen, decode cursors set), all
        ENQUEUED on the current stream without a host sync; returns a handle for `decode_finish`. Two requests can be in flight on
        two streams when they use different cache sets (prefill of one beside the decode loop of the other: SpiderFreeInfer's
        three-stage pipelining); the caller orders `decode_finish(h)` after this call's stream work. More than DECODE_ROWS rows: the
        whole grouped generate runs here and its result is returned instead of a handle.

        Greedy decode. Left-padded batches are described by attention_mask (0 = pad), as
        prepare_generation_embedding does (spider.py:1658-1661). `sync_every` > 1 checks the stop
        conditions only every N tokens (one device->host copy per check instead of per token).
        position_ids [3, B, S]: multimodal (t, h, w) rotary positions of the prompt (Qwen2.5-Omni thinker with image / audio
        embeddings spliced into inputs_embeds; cfg.mrope_section required). Generated tokens continue at
        max(position_ids) + 1 on all three components, as transformers' rope_deltas bookkeeping does.
        eos_token_id / pad_token_id / max_new_tokens default to the checkpoint's generation config (HF behaviour):
        a row is finished at its first EOS and padded with pad_token_id afterwards; the call returns when every row is
        finished. More than DECODE_ROWS (8) rows are processed in groups of 8 (rows are independent)."""
        if num_beams != 1 or do_sample:
            raise NotImplementedError("the reference path is greedy: num_beams=1, do_sample=False (spider.py:1471-1477)")
        c, dv = self.cfg, self.device
        gc = getattr(self, "generation_config", None) or {}
        if eos_token_id is None:
            eos_token_id = gc.get("eos_token_id")
        if pad_token_id is None:
            pad_token_id = gc.get("pad_token_id")
        embeds_only = input_ids is None
        S_in = inputs_embeds.shape[1] if embeds_only else input_ids.shape[1]
        if max_new_tokens is None:   # HF: generation_config.max_new_tokens, else max_length (default 20) counts the prompt
            max_new_tokens = gc.get("max_new_tokens") or max(1, int(gc.get("max_length", 20)) - (0 if embeds_only else S_in))
        B_all = inputs_embeds.shape[0] if embeds_only else input_ids.shape[0]
        if B_all > self.DECODE_ROWS:
            return self._generate_grouped(input_ids, inputs_embeds, attention_mask, position_ids, B_all, dict(
                max_new_tokens=max_new_tokens, stopping_criteria=stopping_criteria, eos_token_id=eos_token_id,
                pad_token_id=pad_token_id, output_hidden_states=output_hidden_states, use_graph=use_graph,
                sync_every=sync_every, return_logits=return_logits, cache_set=cache_set), return_dict_in_generate)
        if embeds_only:
            h0 = inputs_embeds.to(device=dv, dtype=BF16).contiguous()
            B, S = h0.shape[0], h0.shape[1]
        else:
            input_ids = input_ids.to(dv)
            B, S = input_ids.shape
            h0 = self.embed_tokens(input_ids)
        if B > self.max_batch or S + max_new_tokens > self.max_len:
            raise ValueError(f"batch {B} / length {S}+{max_new_tokens} exceed the preallocated KV cache "
                             f"({self.max_batch} x {self.max_len})")
        am = (attention_mask.to(dv).to(torch.int32) if attention_mask is not None
              else torch.ones(B, S, dtype=torch.int32, device=dv))
        pos2d = (am.cumsum(-1) - 1).clamp(min=0).to(torch.int32).contiguous()
        slot2d = torch.arange(S, dtype=torch.int32, device=dv)[None].expand(B, S).contiguous()
        kv_beg = (S - am.sum(-1)).to(torch.int32).contiguous()     # first valid slot (left padding)
        has_pad = attention_mask is not None and bool((am == 0).any())

        hidden_steps: Optional[List] = [] if output_hidden_states else None
        step0: Optional[list] = [] if output_hidden_states else None
        if position_ids is not None:
            if c.mrope_section is None:
                raise ValueError("position_ids with 3 components need a model with mrope_section (Qwen2.5-Omni thinker)")
            if tuple(position_ids.shape) != (3, B, S):
                raise ValueError(f"position_ids must be [3, {B}, {S}], got {tuple(position_ids.shape)}")
            pos3 = position_ids.to(device=dv, dtype=torch.int32).contiguous()
            h = self._prefill(h0.view(B * S, -1), pos3.view(3, -1), slot2d.view(-1), kv_beg if has_pad else None, B, S, step0, mrope=True,
                              cache_set=cache_set)
            # left-padded rows: the pad slots carry a dummy position (get_rope_index writes 1 there) and are masked by kv_beg
            next_pos = torch.where(am.bool()[None], pos3, torch.zeros_like(pos3)).amax(dim=(0, 2)) + 1
        else:
            h = self._prefill(h0.view(B * S, -1), pos2d.view(-1), slot2d.view(-1), kv_beg if has_pad else None, B, S, step0,
                              cache_set=cache_set)
            next_pos = pos2d[:, -1] + 1

        # decode state (static buffers + captured hipGraph) is cached per (batch, outputs): repeated generate() calls
        # replay the same graph instead of re-capturing ~200 launches
        skey = (B, bool(output_hidden_states), bool(return_logits), int(cache_set))
        if skey not in self._graphs:
            self._graphs[skey] = [self._make_state(B, output_hidden_states, return_logits, cache_set), None]
        st = self._graphs[skey][0]
        st["kv_beg"].copy_(kv_beg)
        last = h.view(B, S, -1)[:, -1].contiguous()
        ops.lm_head_argmax(self.lm_head, last, norm_w=self.norm, eps=c.eps, out_ids=st["next_ids"], ws=st["lm_ws"],
                           logits=st.get("logits"))
        if output_hidden_states:
            step0[-1] = ops.rmsnorm(h, self.norm, c.eps).view(B, S, -1)
            hidden_steps.append(tuple(step0))
        st["cur_ids"].copy_(st["next_ids"])
        st["pos"].copy_(next_pos)
        st["slot"].fill_(S)
        st["kv_end"].fill_(S + 1)

        # generated ids live in the decode state's history buffer: every decode step appends its token there itself
        # (decode_advance), so the loop below issues nothing but the graph replay
        if max_new_tokens > st["hist"].shape[1]:
            raise ValueError(f"max_new_tokens={max_new_tokens} exceeds the engine's max_len={st['hist'].shape[1]}")
        tokens = st["hist"][:, :max_new_tokens]
        tokens[:, 0].copy_(st["next_ids"])
        st["n_hist"].fill_(1)
        logits_steps = [st["logits"].clone()] if return_logits else None
        return _PrefillHandle(B=B, S=S, st=st, skey=skey, embeds_only=embeds_only, input_ids=input_ids, max_new_tokens=max_new_tokens,
                              stopping_criteria=stopping_criteria, eos_token_id=eos_token_id, pad_token_id=pad_token_id,
                              output_hidden_states=output_hidden_states, return_dict_in_generate=return_dict_in_generate,
                              use_graph=use_graph, sync_every=sync_every, return_logits=return_logits, hidden_steps=hidden_steps,
                              logits_steps=logits_steps, tokens=tokens)

    @torch.no_grad()
    def adopt(self, hd: "_PrefillHandle", cache_set: int = 0) -> "_PrefillHandle":
        """Move a prefilled request into KV cache set `cache_set` (device copies of the prompt's K / V rows and of the decode cursors,
        enqueued on the current stream) and return the handle bound to that set. Used by SpiderFreeInfer's three-stage pipelining: the
        prompt pass of request k+2 fills a STAGING set while request k+1 decodes from set 0; before its own decode loop the request is
        adopted into set 0, so every decode loop replays the ONE decode graph of set 0 (a second graph executable alive in the
        process was measured to put the two-stream schedule into a time-slicing regime: every kernel + 10-25 us, step 513 -> 920 ms)."""
        src = hd.skey[3]
        if src == cache_set:
            return hd
        B, S = hd.B, hd.S
        skey = hd.skey[:3] + (int(cache_set),)
        if skey not in self._graphs:
            self._graphs[skey] = [self._make_state(B, hd.output_hidden_states, hd.return_logits, cache_set), None]
        dst, st = self._graphs[skey][0], hd.st
        (ks, vs), (kd, vd) = self._kv(src), self._kv(cache_set)
        kd[:, :B, :, :S + 1].copy_(ks[:, :B, :, :S + 1])
        vd[:, :B, :, :S + 1].copy_(vs[:, :B, :, :S + 1])
        for k in ("cur_ids", "next_ids", "pos", "slot", "kv_end", "kv_beg", "n_hist"):
            dst[k].copy_(st[k])
        dst["hist"][:, :1].copy_(st["hist"][:, :1])
        for k in ("logits", "hidden_buf"):
            if k in st:
                dst[k].copy_(st[k])
        nd = _PrefillHandle(**hd.__dict__)
        nd.st, nd.skey, nd.tokens = dst, skey, dst["hist"][:, :hd.max_new_tokens]
        return nd

    @torch.no_grad()
    def decode_finish(self, hd: "_PrefillHandle"):
        """Second half of `generate`: the decode loop (one hipGraph replay per token), the stop checks and the HF bookkeeping."""
        c, dv = self.cfg, self.device
        B, S, st, skey, embeds_only, input_ids = hd.B, hd.S, hd.st, hd.skey, hd.embeds_only, hd.input_ids
        max_new_tokens, stopping_criteria, eos_token_id, pad_token_id = hd.max_new_tokens, hd.stopping_criteria, hd.eos_token_id, hd.pad_token_id
        output_hidden_states, return_dict_in_generate, use_graph = hd.output_hidden_states, hd.return_dict_in_generate, hd.use_graph
        sync_every, return_logits, hidden_steps, logits_steps, tokens = hd.sync_every, hd.return_logits, hd.hidden_steps, hd.logits_steps, hd.tokens
        eos = _id_list(eos_token_id)
        prompt_cpu = None if embeds_only else input_ids.cpu().long()
        need_check = bool(eos) or bool(stopping_criteria)
        final = None      # (padded tokens [B, k] on the host, k) once a stop condition has been met

        def check(n_done: int, already: int):
            if not need_check:
                return None
            tk, k, hit = finalize_greedy(tokens[:, :n_done].cpu().long(), eos, pad_token_id, stopping_criteria,
                                         prompt_cpu, already + 1)
            return (tk, k) if hit else None

        graph = self._graphs[skey][1] if use_graph else None
        n = 1
        checked = 1
        final = check(1, 0)
        if use_graph and graph is None and final is None and max_new_tokens > 2:
            # warm the kernels outside capture, then capture one decode step; cursors live on device
            snap = {k: st[k].clone() for k in ("cur_ids", "next_ids", "pos", "slot", "kv_end", "n_hist")}
            s = torch.cuda.Stream(device=dv)
            s.wait_stream(torch.cuda.current_stream(dv))
            with torch.cuda.stream(s):
                self._decode_step(st)
            torch.cuda.current_stream(dv).wait_stream(s)
            for k, v in snap.items():
                st[k].copy_(v)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._decode_step(st)
            for k, v in snap.items():   # capture does not execute; restore is a no-op safety net
                st[k].copy_(v)
            self._graphs[skey][1] = graph
        while n < max_new_tokens and final is None:
            if graph is not None:
                graph.replay()
            else:
                self._decode_step(st)
            if output_hidden_states:
                hidden_steps.append(tuple(t.clone().unsqueeze(1) for t in st["hidden_buf"]))
            if return_logits:
                logits_steps.append(st["logits"].clone())
            n += 1
            if n % sync_every == 0 or n == max_new_tokens:
                final = check(n, checked)
                checked = n
        if final is not None:   # rows padded after their first EOS; steps that ran past the stop (sync_every > 1) dropped
            gen, n = final[0].to(dv), final[1]
            if output_hidden_states:
                del hidden_steps[n:]
            if return_logits:
                del logits_steps[n:]
        elif need_check:        # no stop met within max_new_tokens: finished rows are still padded
            gen = finalize_greedy(tokens[:, :n].cpu().long(), eos, pad_token_id, None, None)[0].to(dv)
        else:
            gen = tokens[:, :n].long()
        seqs = gen if embeds_only else torch.cat([input_ids.long(), gen], 1)
        out = GenerateOutput(seqs, tuple(hidden_steps) if output_hidden_states else None)
        if return_logits:
            out.logits = torch.stack(logits_steps, 1)
        return out if return_dict_in_generate else seqs

    def _generate_grouped(self, input_ids, inputs_embeds, attention_mask, position_ids, B_all: int, kw: dict, as_dict: bool):
        """More rows than one decode graph holds: groups of DECODE_ROWS, results joined the way one HF call would return
        them (every row padded to the longest group with pad_token_id; a group that ended early repeats the LAST POSITION of its
        last state, [rows, 1, H], for the steps it did not run -- its first entry is the prompt state [rows, S, H])."""
        if kw.get("stopping_criteria"):
            raise NotImplementedError("stopping_criteria look at sequence 0 and end the whole batch (spider.py:55-73); "
                                      f"use them with at most {self.DECODE_ROWS} rows per call")
        outs = []
        R = self.DECODE_ROWS
        for b0 in range(0, B_all, R):
            sl = slice(b0, min(B_all, b0 + R))
            outs.append(self.generate(
                input_ids=None if input_ids is None else input_ids[sl],
                inputs_embeds=None if inputs_embeds is None else inputs_embeds[sl],
                attention_mask=None if attention_mask is None else attention_mask[sl],
                position_ids=None if position_ids is None else position_ids[:, sl], return_dict_in_generate=True, **kw))
        eos = _id_list(kw.get("eos_token_id"))
        pad = kw.get("pad_token_id")
        pad = pad if pad is not None else (eos[0] if eos else 0)
        T = max(o.sequences.shape[1] for o in outs)
        seqs = torch.cat([torch.nn.functional.pad(o.sequences, (0, T - o.sequences.shape[1]), value=pad) for o in outs], 0)
        out = GenerateOutput(seqs, None)
        if kw.get("output_hidden_states"):
            n_steps = max(len(o.hidden_states) for o in outs)
            hs = []
            for stp in range(n_steps):
                per = [o.hidden_states[stp] if stp < len(o.hidden_states) else tuple(h[:, -1:, :] for h in o.hidden_states[-1])
                       for o in outs]
                hs.append(tuple(torch.cat([p[l] for p in per], 0) for l in range(len(per[0]))))
            out.hidden_states = tuple(hs)
        if kw.get("return_logits"):
            n_steps = max(o.logits.shape[1] for o in outs)
            out.logits = torch.cat([torch.cat([o.logits, o.logits[:, -1:].expand(-1, n_steps - o.logits.shape[1], -1)], 1)
                                    for o in outs], 0)
        return out if as_dict else seqs
