"""CLIP text encoder on the HIP kernels: `self.text_encoder(text_input_ids)[0]` of the reference's prompt encoding
(spider/models/custom_sd.py:306-310 cond, :352-356 uncond). Pre-LN transformer, causal attention, quick-GELU MLP,
final LayerNorm. One fused [3H,H] QKV GEMM per layer; q,k,v consumed in place by the flash-attention kernel."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch

from . import ops

BF16 = torch.bfloat16


@dataclass
class CLIPTextConfig:
    vocab: int = 49408
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    inter: int = 3072
    max_pos: int = 77
    eps: float = 1e-5
    act: str = "quick_gelu"

    @staticmethod
    def sd15():
        return CLIPTextConfig()

    @staticmethod
    def sdxl_2():   # OpenCLIP ViT-bigG text tower (SDXL text_encoder_2)
        return CLIPTextConfig(49408, 1280, 32, 20, 5120, 77, 1e-5, "gelu")

    @staticmethod
    def from_hf_dict(c: dict):
        return CLIPTextConfig(c["vocab_size"], c["hidden_size"], c["num_hidden_layers"], c["num_attention_heads"],
                              c["intermediate_size"], c["max_position_embeddings"], c.get("layer_norm_eps", 1e-5),
                              c.get("hidden_act", "quick_gelu"))


def _shapes(c: CLIPTextConfig) -> dict:
    S = {"text_model.embeddings.token_embedding.weight": (c.vocab, c.hidden),
         "text_model.embeddings.position_embedding.weight": (c.max_pos, c.hidden)}
    for l in range(c.layers):
        p = f"text_model.encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            S[p + f"self_attn.{n}.weight"] = (c.hidden, c.hidden); S[p + f"self_attn.{n}.bias"] = (c.hidden,)
        for n in ("layer_norm1", "layer_norm2"):
            S[p + n + ".weight"] = (c.hidden,); S[p + n + ".bias"] = (c.hidden,)
        S[p + "mlp.fc1.weight"] = (c.inter, c.hidden); S[p + "mlp.fc1.bias"] = (c.inter,)
        S[p + "mlp.fc2.weight"] = (c.hidden, c.inter); S[p + "mlp.fc2.bias"] = (c.hidden,)
    S["text_model.final_layer_norm.weight"] = (c.hidden,); S["text_model.final_layer_norm.bias"] = (c.hidden,)
    return S


class CLIPTextEngine:
    def __init__(self, cfg: CLIPTextConfig, weights: Dict[str, torch.Tensor], device="cuda:0", dtype=BF16):
        assert dtype in (torch.bfloat16, torch.float16), "CLIPTextEngine: dtype must be bfloat16 or float16"
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        g = lambda k: weights[k].to(device=self.device, dtype=dtype).contiguous()
        self.tok, self.pos = g("text_model.embeddings.token_embedding.weight"), g("text_model.embeddings.position_embedding.weight")
        self.layers = []
        for l in range(cfg.layers):
            p = f"text_model.encoder.layers.{l}."
            self.layers.append(dict(
                w_qkv=torch.cat([g(p + "self_attn.q_proj.weight"), g(p + "self_attn.k_proj.weight"), g(p + "self_attn.v_proj.weight")], 0).contiguous(),
                b_qkv=torch.cat([g(p + "self_attn.q_proj.bias"), g(p + "self_attn.k_proj.bias"), g(p + "self_attn.v_proj.bias")], 0).contiguous(),
                w_o=g(p + "self_attn.out_proj.weight"), b_o=g(p + "self_attn.out_proj.bias"),
                ln1=(g(p + "layer_norm1.weight"), g(p + "layer_norm1.bias")), ln2=(g(p + "layer_norm2.weight"), g(p + "layer_norm2.bias")),
                w1=g(p + "mlp.fc1.weight"), b1=g(p + "mlp.fc1.bias"), w2=g(p + "mlp.fc2.weight"), b2=g(p + "mlp.fc2.bias")))
        self.lnf = (g("text_model.final_layer_norm.weight"), g("text_model.final_layer_norm.bias"))
        self.text_projection = g("text_projection.weight") if "text_projection.weight" in weights else None

    @classmethod
    def random_init(cls, cfg: CLIPTextConfig, device="cuda:0", seed=0, dtype=BF16):
        gen = torch.Generator(device=device).manual_seed(seed)
        w = {}
        for n, shp in _shapes(cfg).items():
            if n.endswith(".bias"):
                t = torch.zeros(shp, device=device)
            elif "norm" in n:
                t = torch.ones(shp, device=device)
            else:
                t = torch.randn(shp, generator=gen, device=device) * 0.02
            w[n] = t.to(BF16)
        return cls(cfg, w, device, dtype=dtype)

    @classmethod
    def from_pretrained(cls, path: str, device="cuda:0", dtype=BF16):
        from .checkpoint import load_state_dict, read_config
        cfg = CLIPTextConfig.from_hf_dict(read_config(path))
        return cls(cfg, load_state_dict(path), device, dtype=dtype)

    @torch.no_grad()
    def encode(self, ids: torch.Tensor, return_all: bool = False, use_graph: bool = True):
        """ids [B, S<=77] int -> last_hidden_state [B, S, H] bf16. With return_all: dict(last, penultimate, pooled) --
        SDXL conditions on hidden_states[-2] of both encoders and on the projected pooled state of the second
        (StableDiffusionXLPipeline.encode_prompt, reached from Comic_Generation.py:440).
        The plain form replays one hipGraph per id shape (12-32 layers x 7 launches of a few microseconds each)."""
        ids = ids.to(device=self.device, dtype=torch.int32).contiguous()
        if use_graph and not return_all:
            if not hasattr(self, "_graph"):
                from .graphs import GraphRunner
                self._graph = GraphRunner(lambda i: self._encode(i, False))
            return self._graph(ids)
        return self._encode(ids, return_all)

    def _encode(self, ids: torch.Tensor, return_all: bool):
        c = self.cfg
        B, S = ids.shape
        h = ops.add(ops.embed(self.tok, ids), self.pos[:S][None].expand(B, S, c.hidden).contiguous())
        d = c.hidden // c.heads
        H = c.hidden
        penult = None
        for li, lw in enumerate(self.layers):
            if li == len(self.layers) - 1:
                penult = h
            x = ops.layernorm(h, *lw["ln1"], c.eps)
            qkv = ops.gemm(x, lw["w_qkv"], bias=lw["b_qkv"])
            a = ops.attention(qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:], c.heads, scale=d ** -0.5, causal=True)
            h = ops.gemm(a, lw["w_o"], bias=lw["b_o"], res=h)
            x = ops.layernorm(h, *lw["ln2"], c.eps)
            m = ops.gemm(x, lw["w1"], bias=lw["b1"], act=c.act if c.act in ("quick_gelu", "gelu") else None)
            h = ops.gemm(m, lw["w2"], bias=lw["b2"], res=h)
        last = ops.layernorm(h, *self.lnf, c.eps)
        if not return_all:
            return last
        eos = ids.long().argmax(-1)                                   # EOS has the largest id in CLIP vocabularies
        pooled = last[torch.arange(B, device=self.device), eos].contiguous()
        if self.text_projection is not None:
            pooled = ops.gemm(pooled, self.text_projection)
        return dict(last=last, penultimate=penult, pooled=pooled)
