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
        # decoder cross-attention: q from the target stream, fused k|v from the encoder memory
        E = HIDDEN
        for e in range(EXPERTS):
            for l in range(LAYERS):
                p = f"expert_tfm_layers.{e}.decoder.layers.{l}.multihead_attn."
                W, b = self.w[p + "in_proj_weight"], self.w[p + "in_proj_bias"]
                self.w[p + "q_w"], self.w[p + "q_b"] = W[:E].contiguous(), b[:E].contiguous()
                self.w[p + "kv_w"], self.w[p + "kv_b"] = W[E:].contiguous(), b[E:].contiguous()
        # the router's 3 output rows padded to the 4-column granularity of the GEMM epilogue
        for m in output_alignment_modules:
            fw, fb = self.w[f"routers.{m}.fc2.weight"], self.w[f"routers.{m}.fc2.bias"]
            pw = torch.zeros(4, fw.shape[1], dtype=BF16, device=self.device); pw[:EXPERTS] = fw
            pb = torch.zeros(4, dtype=BF16, device=self.device); pb[:EXPERTS] = fb
            self.w[f"routers.{m}.fc2.weight4"], self.w[f"routers.{m}.fc2.bias4"] = pw, pb

    def eval(self):
        return self

    # ------------------------------------------------------------------ nn.Transformer pieces
    def _self_attn(self, p, y):
        E = HIDDEN
        qkv = ops.gemm(y, self.w[p + "in_proj_weight"], bias=self.w[p + "in_proj_bias"])
        return ops.attention(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], HEADS)

    def _transformer(self, t: str, src: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
        w = self.w
        E = HIDDEN
        ln = lambda n, x: ops.layernorm(x, w[n + ".weight"], w[n + ".bias"], 1e-5)
        x = src
        for l in range(LAYERS):
            p = f"{t}encoder.layers.{l}."
            a = self._self_attn(p + "self_attn.", ln(p + "norm1", x))
            x = ops.gemm(a, w[p + "self_attn.out_proj.weight"], bias=w[p + "self_attn.out_proj.bias"], res=x)
            h = ops.gemm(ln(p + "norm2", x), w[p + "linear1.weight"], bias=w[p + "linear1.bias"], act="relu")
            x = ops.gemm(h, w[p + "linear2.weight"], bias=w[p + "linear2.bias"], res=x)
        mem = ln(t + "encoder.norm", x)
        x = tgt
        for l in range(LAYERS):
            p = f"{t}decoder.layers.{l}."
            a = self._self_attn(p + "self_attn.", ln(p + "norm1", x))
            x = ops.gemm(a, w[p + "self_attn.out_proj.weight"], bias=w[p + "self_attn.out_proj.bias"], res=x)
            q = ops.gemm(ln(p + "norm2", x), w[p + "multihead_attn.q_w"], bias=w[p + "multihead_attn.q_b"])
            kv = ops.gemm(mem, w[p + "multihead_attn.kv_w"], bias=w[p + "multihead_attn.kv_b"])
            a = ops.attention(q, kv[..., :E], kv[..., E:], HEADS)
            x = ops.gemm(a, w[p + "multihead_attn.out_proj.weight"], bias=w[p + "multihead_attn.out_proj.bias"], res=x)
            h = ops.gemm(ln(p + "norm3", x), w[p + "linear1.weight"], bias=w[p + "linear1.bias"], act="relu")
            x = ops.gemm(h, w[p + "linear2.weight"], bias=w[p + "linear2.bias"], res=x)
        return ln(t + "decoder.norm", x)

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
