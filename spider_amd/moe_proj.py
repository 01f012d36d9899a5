"""Trained-Spider output projector on the HIP kernels: `TextFcLayerMoE`, mode
'moe_transformer', inference form (spider/models/layers.py:147-279,331): sigmoid router over the token mean, three
experts of Linear(in,512) + nn.Transformer(d=512, 4+4 pre-LN layers, 4 heads, ReLU FFN 2048) decoding the learned
modality tokens against the LLM states, routing-weighted sum, out_fc.

Same class name, constructor arguments and `forward(x, modality)` contract as the reference; weights are the
reference module's state dict. Every matmul is the MFMA GEMM (fused in_proj, bias / ReLU / residual in the epilogue), the
attentions the flash kernel at head_dim 128; the router's sigmoid + normalisation and the expert mixing are one kernel
reading the router logits on the device (no host sync)."""
from __future__ import annotations

from typing import Dict

import torch

from . import ops

BF16 = torch.bfloat16
HIDDEN, EXPERTS, LAYERS, HEADS = 512, 3, 4, 4      # hard-coded in the reference (layers.py:156-157,163,172-174)


def split_cross_attention(w: Dict[str, torch.Tensor], prefix: str) -> None:
    """decoder cross-attention of an nn.Transformer state dict under `prefix`: q from the target stream, fused k|v from the memory"""
    E = HIDDEN
    for l in range(LAYERS):
        p = f"{prefix}decoder.layers.{l}.multihead_attn."
        W, b = w[p + "in_proj_weight"], w[p + "in_proj_bias"]
        w[p + "q_w"], w[p + "q_b"] = W[:E].contiguous(), b[:E].contiguous()
        w[p + "kv_w"], w[p + "kv_b"] = W[E:].contiguous(), b[E:].contiguous()


def nn_transformer(w: Dict[str, torch.Tensor], t: str, src: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """torch.nn.Transformer(batch_first=True, norm_first=True, d_model=512, 4 + 4 layers, nhead=4, ReLU FFN 2048, dropout 0)
    as the reference instantiates it (layers.py:67-69,172-174), on the HIP kernels; `t` = state-dict prefix, src [B, S, 512]
    (encoder input), tgt [B, T, 512] (decoder input); no masks (the reference passes none)."""
    E = HIDDEN
    ln = lambda n, x: ops.layernorm(x, w[n + ".weight"], w[n + ".bias"], 1e-5)

    def self_attn(p, y):
        qkv = ops.gemm(y, w[p + "in_proj_weight"], bias=w[p + "in_proj_bias"])
        return ops.attention(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], HEADS)

    x = src
    for l in range(LAYERS):
        p = f"{t}encoder.layers.{l}."
        a = self_attn(p + "self_attn.", ln(p + "norm1", x))
        x = ops.gemm(a, w[p + "self_attn.out_proj.weight"], bias=w[p + "self_attn.out_proj.bias"], res=x)
        h = ops.gemm(ln(p + "norm2", x), w[p + "linear1.weight"], bias=w[p + "linear1.bias"], act="relu")
        x = ops.gemm(h, w[p + "linear2.weight"], bias=w[p + "linear2.bias"], res=x)
    mem = ln(t + "encoder.norm", x)
    x = tgt
    for l in range(LAYERS):
        p = f"{t}decoder.layers.{l}."
        a = self_attn(p + "self_attn.", ln(p + "norm1", x))
        x = ops.gemm(a, w[p + "self_attn.out_proj.weight"], bias=w[p + "self_attn.out_proj.bias"], res=x)
        q = ops.gemm(ln(p + "norm2", x), w[p + "multihead_attn.q_w"], bias=w[p + "multihead_attn.q_b"])
        kv = ops.gemm(mem, w[p + "multihead_attn.kv_w"], bias=w[p + "multihead_attn.kv_b"])
        a = ops.attention(q, kv[..., :E], kv[..., E:], HEADS)
        x = ops.gemm(a, w[p + "multihead_attn.out_proj.weight"], bias=w[p + "multihead_attn.out_proj.bias"], res=x)
        h = ops.gemm(ln(p + "norm3", x), w[p + "linear1.weight"], bias=w[p + "linear1.bias"], act="relu")
        x = ops.gemm(h, w[p + "linear2.weight"], bias=w[p + "linear2.bias"], res=x)
    return ln(t + "decoder.norm", x)


class TextFcLayer:
    """Per-modality alignment projector of trained Spider when no MoE mode is configured (`output_alignment_MoE_mode is None`,
    spider.py:200-209): spider/models/layers.py:26-144. Same constructor arguments and `forward(x, modality=None)` contract;
    `weights` = the reference module's state dict.
      mode 'linear':       outputs = model(x)                                                  (layers.py:64-65)
      mode 'transformer':  outputs = model(tfm(fc(x), query_embs.repeat(B, 1, 1)))             (layers.py:66-75,113-124)
      mode 'qformer':      outputs = model(Qformer.bert(query_tokens, encoder_hidden_states=fc(x)))  (layers.py:76-98,125-139):
                           the 2-layer BLIP-2 style Q-Former of spider/models/Qformer.py on its query branch only (embedding
                           LayerNorm; per layer post-LN self-attention over the queries, cross-attention to fc(x), GELU feed-forward),
                           hidden 768; `qformer_heads` = the BERT config's num_attention_heads (12 for bert-base-uncased, which
                           init_Qformer loads; a state dict does not carry it)."""

    def __init__(self, in_dim: int, out_dim: int, num_input_tokens: int = 1, num_output_tokens: int = 1, mode: str = "linear",
                 device="cuda:0", freeze_qformer=False, weights: Dict[str, torch.Tensor] = None, qformer_heads: int = 12):
        if mode not in ("linear", "transformer", "qformer"):
            raise NotImplementedError(mode)
        if weights is None:
            raise ValueError("TextFcLayer needs the reference module's state dict (weights=...)")
        self.num_input_tokens, self.num_output_tokens, self.mode, self.out_dim = num_input_tokens, num_output_tokens, mode, out_dim
        self.device = torch.device(device)
        self.w = {k: v.to(device=self.device, dtype=BF16).contiguous() for k, v in weights.items()}
        if mode == "transformer":
            split_cross_attention(self.w, "tfm.")
        if mode == "qformer":
            self.qf_heads = qformer_heads
            w = self.w
            self.qf_layers = 0
            while f"Qformer.bert.encoder.layer.{self.qf_layers}.attention.self.query.weight" in w:
                p = f"Qformer.bert.encoder.layer.{self.qf_layers}."
                cat = lambda names, suf: torch.cat([w[p + n + suf] for n in names]).contiguous()
                # fused projections: self-attention q|k|v of the queries; cross-attention k|v of the encoder states
                w[p + "self.qkv_w"] = cat([f"attention.self.{n}" for n in ("query", "key", "value")], ".weight")
                w[p + "self.qkv_b"] = cat([f"attention.self.{n}" for n in ("query", "key", "value")], ".bias")
                w[p + "cross.kv_w"] = cat([f"crossattention.self.{n}" for n in ("key", "value")], ".weight")
                w[p + "cross.kv_b"] = cat([f"crossattention.self.{n}" for n in ("key", "value")], ".bias")
                self.qf_layers += 1
            if self.qf_layers == 0:
                raise ValueError("TextFcLayer(mode='qformer'): no Qformer.bert.encoder.layer.* weights in the state dict")
            H = w["fc.weight"].shape[0]
            if H % qformer_heads or H // qformer_heads % 8:
                raise ValueError(f"TextFcLayer(mode='qformer'): hidden {H} / {qformer_heads} heads is not a multiple of 8")

    def eval(self):
        return self

    @torch.no_grad()
    def forward(self, x: torch.Tensor, modality=None) -> torch.Tensor:
        w = self.w
        x = x.to(device=self.device, dtype=BF16).contiguous()
        if self.mode == "linear":
            outputs = ops.gemm(x, w["model.weight"], bias=w["model.bias"])
        elif self.mode == "qformer":
            outputs = self._qformer(x)
        else:
            h = ops.gemm(x, w["fc.weight"], bias=w["fc.bias"])
            tgt = w["query_embs"].expand(x.shape[0], -1, -1).contiguous()
            outputs = ops.gemm(nn_transformer(w, "tfm.", h, tgt), w["model.weight"], bias=w["model.bias"])
        assert outputs.shape[1] == 1 or (outputs.shape[1] * outputs.shape[2] == self.num_output_tokens * self.out_dim), \
            (tuple(outputs.shape), self.num_output_tokens)          # layers.py:141-143
        return outputs

    def _qformer(self, x: torch.Tensor) -> torch.Tensor:
        """layers.py:125-139 with Qformer.py's BertEmbeddings (query_embeds only, :78-108) and BertLayer (query_length = Q, :402-484)"""
        w, nh, eps = self.w, self.qf_heads, 1e-12
        B = x.shape[0]
        enc = ops.gemm(x, w["fc.weight"], bias=w["fc.bias"])                                   # [B, T, 768]
        H = enc.shape[-1]
        ln = lambda n, t: ops.layernorm(t, w[n + ".weight"], w[n + ".bias"], eps)
        h = ln("Qformer.bert.embeddings.LayerNorm", w["query_tokens"].expand(B, -1, -1).contiguous())
        for l in range(self.qf_layers):
            p = f"Qformer.bert.encoder.layer.{l}."
            qkv = ops.gemm(h, w[p + "self.qkv_w"], bias=w[p + "self.qkv_b"])
            a = ops.attention(qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:], nh)
            h = ln(p + "attention.output.LayerNorm", ops.gemm(a, w[p + "attention.output.dense.weight"], bias=w[p + "attention.output.dense.bias"], res=h))
            q = ops.gemm(h, w[p + "crossattention.self.query.weight"], bias=w[p + "crossattention.self.query.bias"])
            kv = ops.gemm(enc, w[p + "cross.kv_w"], bias=w[p + "cross.kv_b"])
            a = ops.attention(q, kv[..., :H], kv[..., H:], nh)
            h = ln(p + "crossattention.output.LayerNorm",
                   ops.gemm(a, w[p + "crossattention.output.dense.weight"], bias=w[p + "crossattention.output.dense.bias"], res=h))
            f = ops.gemm(h, w[p + "intermediate_query.dense.weight"], bias=w[p + "intermediate_query.dense.bias"], act="gelu")
            h = ln(p + "output_query.LayerNorm", ops.gemm(f, w[p + "output_query.dense.weight"], bias=w[p + "output_query.dense.bias"], res=h))
        return ops.gemm(h, w["model.weight"], bias=w["model.bias"])

    __call__ = forward


class TextFcLayerMoE:
    def __init__(self, in_dim: int, output_alignment_modules: Dict[str, dict], mode: str = "moe_transformer",
                 reconstruct_loss: bool = False, device="cuda:0", weights: Dict[str, torch.Tensor] = None):
        if mode != "moe_transformer":
            raise NotImplementedError(mode)            # as the reference raises for unknown modes (layers.py:246-247)
        if reconstruct_loss:
            raise NotImplementedError("reconstruct_loss is a training-time branch (layers.py:270-297)")
        if weights is None:
            raise ValueError("TextFcLayerMoE needs the reference module's state dict (weights=...)")
        self.in_dim, self.output_alignment_modules, self.mode = in_dim, output_alignment_modules, mode
        self.device = torch.device(device)
        self.w = {k: v.to(device=self.device, dtype=BF16).contiguous() for k, v in weights.items()}
        for e in range(EXPERTS):
            split_cross_attention(self.w, f"expert_tfm_layers.{e}.")
        # the router's 3 output rows padded to the 4-column granularity of the GEMM epilogue
        for m in output_alignment_modules:
            fw, fb = self.w[f"routers.{m}.fc2.weight"], self.w[f"routers.{m}.fc2.bias"]
            pw = torch.zeros(4, fw.shape[1], dtype=BF16, device=self.device); pw[:EXPERTS] = fw
            pb = torch.zeros(4, dtype=BF16, device=self.device); pb[:EXPERTS] = fb
            self.w[f"routers.{m}.fc2.weight4"], self.w[f"routers.{m}.fc2.bias4"] = pw, pb

    def eval(self):
        return self

    def _transformer(self, t: str, src: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
        return nn_transformer(self.w, t, src, tgt)

    # ------------------------------------------------------------------ forward (layers.py:249-268,331)
    @torch.no_grad()
    def forward(self, x: torch.Tensor, modality: str = "IMAGE") -> torch.Tensor:
        """x [1, tokens, in_dim] -> [1, alignment_output_tokens, alignment_output_dim] bf16."""
        w = self.w
        if f"out_fc.{modality}.weight" not in w:
            raise KeyError(modality)                    # ModuleDict lookup of an unknown modality raises KeyError too
        x = x.to(device=self.device, dtype=BF16).contiguous()
        B = x.shape[0]
        if B != 1:   # the reference's `x_expert * routing_weights[:, :, expert]` (layers.py:265) only broadcasts for B == 1
            raise ValueError(f"TextFcLayerMoE.forward takes one caption at a time (batch {B})")
        r = ops.gemm(ops.mean_tokens(x), w[f"routers.{modality}.fc1.weight"], bias=w[f"routers.{modality}.fc1.bias"], act="gelu")
        logits = ops.gemm(r, w[f"routers.{modality}.fc2.weight4"], bias=w[f"routers.{modality}.fc2.bias4"])      # [B, 4], 3 used
        tgt = w[f"modality_tokens.{modality}"].expand(B, -1, -1).contiguous()
        outs = []
        for e in range(EXPERTS):
            h = ops.gemm(x, w[f"expert_fc_layers.{e}.weight"], bias=w[f"expert_fc_layers.{e}.bias"])
            outs.append(self._transformer(f"expert_tfm_layers.{e}.", h, tgt))
        mixed = ops.moe_combine(outs, logits)
        return ops.gemm(mixed, w[f"out_fc.{modality}.weight"], bias=w[f"out_fc.{modality}.bias"])

    __call__ = forward
