"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the trained-Spider OUTPUT side
(SURVEY.md section 8f, N3):

  * `TextFcLayerMoE.forward`, mode 'moe_transformer', inference (reconstruct_loss off) -- spider/models/layers.py:147-279:
    sigmoid router over the token mean (:254-257), 3 experts = Linear(in,512) + nn.Transformer(d=512, 4 encoder + 4
    decoder layers, 4 heads, ff 2048, norm_first, batch_first, ReLU, dropout 0) fed with the learned modality tokens as
    the decoder input (:262-263), routing-weighted sum (:265-267), out_fc (:268). nn.Transformer's arithmetic is
    written out here (pre-LN encoder/decoder layers, fused in_proj, final LayerNorms) rather than called.
  * `Spider.preparing_output_embeds_infer` -- spider/models/spider.py:1413-1463: positions of the `<M>` / `</M>` signal
    tokens in the generated ids, hidden states of the last `modality_tokens[M]` steps before `</M>` at the alignment
    layers, embeddings of those ids, and the caption span in between.
  * the projection + blend of `Spider.decode_image` -- spider/models/spider.py:346-385,420-447: sum over alignment
    layers of fc_layer(hidden + input embeds), then 0.1 * projected + 0.9 * text-encoder embeds.

PINNED: tests/golden/make_golden.py::gen_moe executes the reference's own `Mlp` / `TextFcLayerMoE` class bodies and
`preparing_output_embeds_infer` on seeded inputs (moe_proj_ref.npz). The expert stack has 88 M parameters (hidden 512
and 4+4 layers are hard-coded in the reference), so the fixture stores inputs and outputs only; the weights are
regenerated from the seed by `random_moe_weights` on both sides.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

HIDDEN, EXPERTS, LAYERS, HEADS = 512, 3, 4, 4   # hard-coded in layers.py:156-157,163,172-174


def moe_param_shapes(in_dim: int, modalities: Dict[str, Dict[str, int]]) -> Dict[str, tuple]:
    """state-dict names of TextFcLayerMoE(in_dim, output_alignment_modules=modalities, mode='moe_transformer')."""
    S = {}
    E, FFD = HIDDEN, 4 * HIDDEN
    for e in range(EXPERTS):
        S[f"expert_fc_layers.{e}.weight"] = (E, in_dim); S[f"expert_fc_layers.{e}.bias"] = (E,)
        t = f"expert_tfm_layers.{e}."
        for side, n_attn in (("encoder", ("self_attn",)), ("decoder", ("self_attn", "multihead_attn"))):
            for l in range(LAYERS):
                p = f"{t}{side}.layers.{l}."
                for a in n_attn:
                    S[p + a + ".in_proj_weight"] = (3 * E, E); S[p + a + ".in_proj_bias"] = (3 * E,)
                    S[p + a + ".out_proj.weight"] = (E, E); S[p + a + ".out_proj.bias"] = (E,)
                S[p + "linear1.weight"] = (FFD, E); S[p + "linear1.bias"] = (FFD,)
                S[p + "linear2.weight"] = (E, FFD); S[p + "linear2.bias"] = (E,)
                for k in range(len(n_attn) + 1):
                    S[p + f"norm{k + 1}.weight"] = (E,); S[p + f"norm{k + 1}.bias"] = (E,)
            S[f"{t}{side}.norm.weight"] = (E,); S[f"{t}{side}.norm.bias"] = (E,)
    for m, cfg in modalities.items():
        S[f"routers.{m}.fc1.weight"] = (in_dim, in_dim); S[f"routers.{m}.fc1.bias"] = (in_dim,)
        S[f"routers.{m}.fc2.weight"] = (EXPERTS, in_dim); S[f"routers.{m}.fc2.bias"] = (EXPERTS,)
        S[f"out_fc.{m}.weight"] = (cfg["alignment_output_dim"], E); S[f"out_fc.{m}.bias"] = (cfg["alignment_output_dim"],)
        S[f"modality_tokens.{m}"] = (1, cfg["alignment_output_tokens"], E)
    return S


def random_moe_weights(in_dim: int, modalities: Dict[str, Dict[str, int]], seed=0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    w = {}
    for n, shp in moe_param_shapes(in_dim, modalities).items():
        if n.endswith("bias"):
            t = torch.randn(shp, generator=g) * 0.05
        elif ".norm" in n and n.endswith(".weight") and len(shp) == 1:
            t = 1.0 + torch.randn(shp, generator=g) * 0.1
        elif n.startswith("modality_tokens"):
            t = torch.randn(shp, generator=g)
        else:
            t = torch.randn(shp, generator=g) / math.sqrt(shp[-1])
        w[n] = t.bfloat16().float()
    return w


def _mha(w, p, q_in, kv_in):
    E = HIDDEN
    d = E // HEADS
    W, b = w[p + ".in_proj_weight"], w[p + ".in_proj_bias"]
    q = F.linear(q_in, W[:E], b[:E]); k = F.linear(kv_in, W[E:2 * E], b[E:2 * E]); v = F.linear(kv_in, W[2 * E:], b[2 * E:])
    B, Lq, _ = q.shape
    sh = lambda t: t.view(B, -1, HEADS, d).transpose(1, 2)
    pr = torch.softmax(sh(q) @ sh(k).transpose(-1, -2) / math.sqrt(d), -1)
    o = (pr @ sh(v)).transpose(1, 2).reshape(B, Lq, E)
    return F.linear(o, w[p + ".out_proj.weight"], w[p + ".out_proj.bias"])


def _ln(w, p, x):
    return F.layer_norm(x, (HIDDEN,), w[p + ".weight"], w[p + ".bias"], 1e-5)


def transformer_forward(w: Dict[str, torch.Tensor], t: str, src: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """nn.Transformer(batch_first, norm_first, relu, dropout 0)(src, tgt) with no masks."""
    ff = lambda p, x: F.linear(F.relu(F.linear(x, w[p + "linear1.weight"], w[p + "linear1.bias"])), w[p + "linear2.weight"], w[p + "linear2.bias"])
    x = src
    for l in range(LAYERS):
        p = f"{t}encoder.layers.{l}."
        y = _ln(w, p + "norm1", x); x = x + _mha(w, p + "self_attn", y, y)
        x = x + ff(p, _ln(w, p + "norm2", x))
    mem = _ln(w, t + "encoder.norm", x)
    x = tgt
    for l in range(LAYERS):
        p = f"{t}decoder.layers.{l}."
        y = _ln(w, p + "norm1", x); x = x + _mha(w, p + "self_attn", y, y)
        x = x + _mha(w, p + "multihead_attn", _ln(w, p + "norm2", x), mem)
        x = x + ff(p, _ln(w, p + "norm3", x))
    return _ln(w, t + "decoder.norm", x)


@torch.no_grad()
def moe_forward(w: Dict[str, torch.Tensor], x: torch.Tensor, modality: str) -> torch.Tensor:
    """x [1, tokens, in_dim] -> [1, alignment_output_tokens, alignment_output_dim] (layers.py:249-268,331). Batch 1 only:
    the reference's `x_expert * routing_weights[:, :, expert]` ([B,T,512] * [B,1]) is well-formed only for B == 1."""
    assert x.shape[0] == 1
    xr = x.mean(dim=1, keepdim=True)
    r = F.linear(F.gelu(F.linear(xr, w[f"routers.{modality}.fc1.weight"], w[f"routers.{modality}.fc1.bias"])),
                 w[f"routers.{modality}.fc2.weight"], w[f"routers.{modality}.fc2.bias"]).sigmoid()
    r = r / r.sum(dim=-1, keepdim=True)                                    # [B,1,3]
    tgt = w[f"modality_tokens.{modality}"].repeat(x.shape[0], 1, 1)
    acc = 0
    for e in range(EXPERTS):
        h = F.linear(x, w[f"expert_fc_layers.{e}.weight"], w[f"expert_fc_layers.{e}.bias"])
        acc = acc + transformer_forward(w, f"expert_tfm_layers.{e}.", h, tgt) * r[:, :, e]
    return F.linear(acc, w[f"out_fc.{modality}.weight"], w[f"out_fc.{modality}.bias"])


def signal_positions(targets: Sequence[int], begin_id: int, end_id: int):
    """spider.py:1431-1432: step indices of every `<M>` and `</M>` token in the generated ids (BOS already dropped)."""
    t = list(int(v) for v in targets)
    return [i for i, v in enumerate(t) if v == begin_id], [i for i, v in enumerate(t) if v == end_id]


def capture_spans(targets: Sequence[int], begin_id: int, end_id: int, n_modality_tokens: int, modality_i: int):
    """Index ranges preparing_output_embeds_infer reads for caption number `modality_i` (spider.py:1441-1452):
    signal span [end - n, end) and caption span [start + 1, end - n), as (lo, hi) pairs of generation-step indices."""
    start_pos, end_pos = signal_positions(targets, begin_id, end_id)
    e, s = end_pos[modality_i], start_pos[modality_i]
    return (e - n_modality_tokens, e), (s + 1, e - n_modality_tokens)


def blend(projected: torch.Tensor, condition_embeds: torch.Tensor, hidden_embeds_scale: float = 0.1) -> torch.Tensor:
    """spider.py:420,432,444: hidden_embeds_scale * proj + (1 - hidden_embeds_scale) * text-encoder embeds."""
    return hidden_embeds_scale * projected + (1 - hidden_embeds_scale) * condition_embeds


# ----------------------------------------------------------------------------------------------------------------------
# TextFcLayer, mode 'qformer' (spider/models/layers.py:30-44 init_Qformer, :76-98 constructor, :125-139 forward): fc to 768,
# then the BLIP-2 style Q-Former of spider/models/Qformer.py driven by learned query tokens ONLY (no text branch: the constructor
# deletes the word / position embeddings and every layer's text feed-forward), then Linear(768, out_dim).
#   BertEmbeddings.forward with query_embeds only (Qformer.py:78-108):  h = LayerNorm(query_tokens)
#   BertLayer.forward with query_length = Q (Qformer.py:402-476), for each layer (cross_attention_freq = 1: every layer):
#       h = LN(dense(SelfAttn(h)) + h)                         BertSelfAttention :169-275 (scores / sqrt(d), softmax), BertSelfOutput :285-289
#       h = LN(dense(CrossAttn(h, x)) + h)                     keys / values from the fc output (encoder_width = 768); all-ones mask
#       h = LN(dense(gelu(dense(h))) + h)                      intermediate_query / output_query :349-375,479-484 (hidden_act gelu)
# PINNED: tests/golden/make_golden.py::gen_qformer runs the reference's own TextFcLayer (layers.py) and Q-Former classes
# (Qformer.py) on seeded inputs (textfc_qformer_ref.npz); only `init_Qformer`'s from_pretrained of a local bert-base-uncased
# directory is replaced by a BertConfig of the same fields. Weights are regenerated from the seed on both sides (10 M parameters).
# ----------------------------------------------------------------------------------------------------------------------
QF_HIDDEN, QF_HEADS, QF_LAYERS, QF_EPS = 768, 12, 2, 1e-12      # bert-base-uncased fields + init_Qformer(num_hidden_layers=2)


def qformer_param_shapes(in_dim: int, out_dim: int, n_query: int, inter: int = 3072, max_pos: int = 512) -> Dict[str, tuple]:
    """state-dict keys of the reference module after its constructor has pruned the text branch (layers.py:84-90)"""
    H = QF_HIDDEN
    s = {"fc.weight": (H, in_dim), "fc.bias": (H,), "query_tokens": (1, n_query, H), "model.weight": (out_dim, H), "model.bias": (out_dim,),
         "Qformer.bert.embeddings.LayerNorm.weight": (H,), "Qformer.bert.embeddings.LayerNorm.bias": (H,)}
    for l in range(QF_LAYERS):
        p = f"Qformer.bert.encoder.layer.{l}."
        for att in ("attention", "crossattention"):
            for n in ("query", "key", "value"):
                s[p + f"{att}.self.{n}.weight"], s[p + f"{att}.self.{n}.bias"] = (H, H), (H,)
            s[p + f"{att}.output.dense.weight"], s[p + f"{att}.output.dense.bias"] = (H, H), (H,)
            s[p + f"{att}.output.LayerNorm.weight"], s[p + f"{att}.output.LayerNorm.bias"] = (H,), (H,)
        s[p + "intermediate_query.dense.weight"], s[p + "intermediate_query.dense.bias"] = (inter, H), (inter,)
        s[p + "output_query.dense.weight"], s[p + "output_query.dense.bias"] = (H, inter), (H,)
        s[p + "output_query.LayerNorm.weight"], s[p + "output_query.LayerNorm.bias"] = (H,), (H,)
    return s


def random_qformer_weights(in_dim: int, out_dim: int, n_query: int, inter: int = 3072, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    w = {}
    for k, shp in qformer_param_shapes(in_dim, out_dim, n_query, inter).items():
        if k.endswith("LayerNorm.weight"):
            w[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            w[k] = 0.05 * torch.randn(shp, generator=g)
        elif k == "query_tokens":
            w[k] = 0.5 * torch.randn(shp, generator=g)
        else:
            w[k] = torch.randn(shp, generator=g) / math.sqrt(shp[-1])
    return {k: v.bfloat16().float() for k, v in w.items()}       # bf16-representable: the HIP engine holds them in bf16


def _qf_attn(w, p, h, kv):
    B, Q, H = h.shape
    d = H // QF_HEADS
    lin = lambda n, t: F.linear(t, w[p + f"self.{n}.weight"], w[p + f"self.{n}.bias"])
    q = lin("query", h).view(B, Q, QF_HEADS, d).transpose(1, 2)
    k = lin("key", kv).view(B, -1, QF_HEADS, d).transpose(1, 2)
    v = lin("value", kv).view(B, -1, QF_HEADS, d).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, Q, H)
    o = F.linear(a, w[p + "output.dense.weight"], w[p + "output.dense.bias"])
    return F.layer_norm(o + h, (H,), w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], QF_EPS)


def qformer_textfc_forward(w: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """x [B, T, in_dim] fp32 -> [B, n_query, out_dim] (layers.py:125-139)"""
    H = QF_HIDDEN
    x = F.linear(x, w["fc.weight"], w["fc.bias"])
    h = w["query_tokens"].expand(x.shape[0], -1, -1)
    h = F.layer_norm(h, (H,), w["Qformer.bert.embeddings.LayerNorm.weight"], w["Qformer.bert.embeddings.LayerNorm.bias"], QF_EPS)
    for l in range(QF_LAYERS):
        p = f"Qformer.bert.encoder.layer.{l}."
        h = _qf_attn(w, p + "attention.", h, h)
        h = _qf_attn(w, p + "crossattention.", h, x)
        f = F.gelu(F.linear(h, w[p + "intermediate_query.dense.weight"], w[p + "intermediate_query.dense.bias"]))
        o = F.linear(f, w[p + "output_query.dense.weight"], w[p + "output_query.dense.bias"])
        h = F.layer_norm(o + h, (H,), w[p + "output_query.LayerNorm.weight"], w[p + "output_query.LayerNorm.bias"], QF_EPS)
    return F.linear(h, w["model.weight"], w["model.bias"])
