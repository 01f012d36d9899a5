"""CLAP text branch on the HIP kernels: `self.text_encoder(ids, attention_mask=mask).text_embeds` of the reference's
AudioLDM prompt encoding (spider/models/custom_ad.py:214-219 cond, :266-273 uncond; class
ClapTextModelWithProjection, custom_ad.py:22). RoBERTa arithmetic: post-LN encoder, padding-offset position ids,
tanh pooler on token 0, Linear-ReLU-Linear projection.

The tokenizer pads on the right and padded keys are masked out of every softmax, so token 0 -- the only row the pooler
reads -- depends on the valid prefix alone: each prompt is encoded over its own unpadded length (no mask tensor, no
wasted rows), with one fused [3H,H] QKV GEMM per layer consumed in place by the flash-attention kernel."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch

from . import ops

BF16 = torch.bfloat16


@dataclass
class ClapTextConfig:
    vocab: int = 50265
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    inter: int = 3072
    max_pos: int = 514
    proj_dim: int = 512
    eps: float = 1e-12
    pad_id: int = 1

    @staticmethod
    def from_hf_dict(c: dict):
        t = c.get("text_config", c)
        return ClapTextConfig(t.get("vocab_size", 50265), t.get("hidden_size", 768), t.get("num_hidden_layers", 12),
                              t.get("num_attention_heads", 12), t.get("intermediate_size", 3072),
                              t.get("max_position_embeddings", 514), t.get("projection_dim", c.get("projection_dim", 512)),
                              t.get("layer_norm_eps", 1e-12), t.get("pad_token_id", 1))


def _shapes(c: ClapTextConfig) -> dict:
    S = {"text_model.embeddings.word_embeddings.weight": (c.vocab, c.hidden),
         "text_model.embeddings.position_embeddings.weight": (c.max_pos, c.hidden),
         "text_model.embeddings.token_type_embeddings.weight": (1, c.hidden),
         "text_model.embeddings.LayerNorm.weight": (c.hidden,), "text_model.embeddings.LayerNorm.bias": (c.hidden,)}
    for l in range(c.layers):
        a = f"text_model.encoder.layer.{l}."
        for n, (o, i) in {"attention.self.query": (c.hidden, c.hidden), "attention.self.key": (c.hidden, c.hidden),
                          "attention.self.value": (c.hidden, c.hidden), "attention.output.dense": (c.hidden, c.hidden),
                          "intermediate.dense": (c.inter, c.hidden), "output.dense": (c.hidden, c.inter)}.items():
            S[a + n + ".weight"] = (o, i); S[a + n + ".bias"] = (o,)
        for n in ("attention.output.LayerNorm", "output.LayerNorm"):
            S[a + n + ".weight"] = (c.hidden,); S[a + n + ".bias"] = (c.hidden,)
    S["text_model.pooler.dense.weight"] = (c.hidden, c.hidden); S["text_model.pooler.dense.bias"] = (c.hidden,)
    S["text_projection.linear1.weight"] = (c.proj_dim, c.hidden); S["text_projection.linear1.bias"] = (c.proj_dim,)
    S["text_projection.linear2.weight"] = (c.proj_dim, c.proj_dim); S["text_projection.linear2.bias"] = (c.proj_dim,)
    return S


class ClapTextEngine:
    def __init__(self, cfg: ClapTextConfig, weights: Dict[str, torch.Tensor], device="cuda:0", dtype=BF16):
        assert dtype in (torch.bfloat16, torch.float16), "ClapTextEngine: dtype must be bfloat16 or float16"
        self.cfg, self.device = cfg, torch.device(device)
        self.dtype = dtype
        g = lambda k: weights[k].to(device=self.device, dtype=dtype).contiguous()
        e = "text_model.embeddings."
        self.tok = g(e + "word_embeddings.weight")
        # position + the single token-type row, added once
        self.pos = (weights[e + "position_embeddings.weight"].float() + weights[e + "token_type_embeddings.weight"].float()[0]) \
            .to(device=self.device, dtype=torch.float32).contiguous()
        self.ln_e = (g(e + "LayerNorm.weight"), g(e + "LayerNorm.bias"))
        self.layers = []
        for l in range(cfg.layers):
            a = f"text_model.encoder.layer.{l}."
            self.layers.append(dict(
                w_qkv=torch.cat([g(a + "attention.self.query.weight"), g(a + "attention.self.key.weight"), g(a + "attention.self.value.weight")], 0).contiguous(),
                b_qkv=torch.cat([g(a + "attention.self.query.bias"), g(a + "attention.self.key.bias"), g(a + "attention.self.value.bias")], 0).contiguous(),
                w_o=g(a + "attention.output.dense.weight"), b_o=g(a + "attention.output.dense.bias"),
                ln1=(g(a + "attention.output.LayerNorm.weight"), g(a + "attention.output.LayerNorm.bias")),
                w1=g(a + "intermediate.dense.weight"), b1=g(a + "intermediate.dense.bias"),
                w2=g(a + "output.dense.weight"), b2=g(a + "output.dense.bias"),
                ln2=(g(a + "output.LayerNorm.weight"), g(a + "output.LayerNorm.bias"))))
        self.pool = (g("text_model.pooler.dense.weight"), g("text_model.pooler.dense.bias"))
        self.p1 = (g("text_projection.linear1.weight"), g("text_projection.linear1.bias"))
        self.p2 = (g("text_projection.linear2.weight"), g("text_projection.linear2.bias"))

    @classmethod
    def random_init(cls, cfg: ClapTextConfig, device="cuda:0", seed=0, dtype=BF16):
        gen = torch.Generator(device=device).manual_seed(seed)
        w = {}
        for n, shp in _shapes(cfg).items():
            if n.endswith(".bias"):
                t = torch.zeros(shp, device=device)
            elif "LayerNorm" in n:
                t = torch.ones(shp, device=device)
            else:
                t = torch.randn(shp, generator=gen, device=device) * 0.02
            w[n] = t.to(BF16)
        return cls(cfg, w, device, dtype=dtype)

    @classmethod
    def from_pretrained(cls, path: str, device="cuda:0", dtype=BF16):
        from .checkpoint import load_state_dict, read_config
        cfg = ClapTextConfig.from_hf_dict(read_config(path))
        return cls(cfg, load_state_dict(path), device, dtype=dtype)

    def _encode_one(self, ids: torch.Tensor) -> torch.Tensor:
        """ids [n] int32 (no padding) -> pooled+projected embedding [1, proj_dim] bf16."""
        c = self.cfg
        n = ids.shape[0]
        H, d = c.hidden, c.hidden // c.heads
        # RoBERTa position ids of an unpadded sequence: pad_id+1 .. pad_id+n
        emb = (ops.embed(self.tok, ids[None]).float() + self.pos[c.pad_id + 1: c.pad_id + 1 + n][None]).to(self.dtype)
        h = ops.layernorm(emb, *self.ln_e, c.eps)
        for lw in self.layers:
            qkv = ops.gemm(h, lw["w_qkv"], bias=lw["b_qkv"])
            a = ops.attention(qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:], c.heads, scale=d ** -0.5)
            h = ops.layernorm(ops.gemm(a, lw["w_o"], bias=lw["b_o"], res=h), *lw["ln1"], c.eps)
            m = ops.gemm(h, lw["w1"], bias=lw["b1"], act="gelu")
            h = ops.layernorm(ops.gemm(m, lw["w2"], bias=lw["b2"], res=h), *lw["ln2"], c.eps)
        pooled = ops.gemm(h[:, 0].contiguous(), self.pool[0], bias=self.pool[1], act="tanh")
        return ops.gemm(ops.gemm(pooled, self.p1[0], bias=self.p1[1], act="relu"), self.p2[0], bias=self.p2[1])

    @torch.no_grad()
    def text_embeds(self, ids: torch.Tensor, attention_mask: torch.Tensor = None, normalize: bool = False) -> torch.Tensor:
        """ids [B,S] (right-padded), attention_mask [B,S] (1 = token) -> [B, proj_dim] bf16; normalize=True applies the
        pipeline's F.normalize (custom_ad.py:219)."""
        ids = ids.to(self.device)
        if attention_mask is None:
            attention_mask = (ids != self.cfg.pad_id)
        am = attention_mask.to(self.device).bool()
        lens = am.sum(-1).tolist()
        for b, n in enumerate(lens):
            if n < 1 or not bool(am[b, :n].all()):
                raise ValueError("ClapTextEngine expects right-padded sequences with at least one token")
            if n + self.cfg.pad_id + 1 > self.cfg.max_pos:
                raise ValueError(f"sequence of {n} tokens exceeds the position table ({self.cfg.max_pos})")
        out = torch.cat([self._encode_one(ids[b, :n].to(torch.int32).contiguous()) for b, n in enumerate(lens)], 0)
        return ops.l2_normalize(out) if normalize else out
