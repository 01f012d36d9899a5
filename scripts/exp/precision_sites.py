"""Precision attribution (CPU, test-side aid; round 5): WHERE does the f16 + fp32-stream engine's 1.2e-3 come from?

The oracle graph (oracle/unet.py) is run with every tensor the HIP engine stores in HBM rounded to f16 (the emulation of
scripts/exp/precision_emul.py: fp32 residual stream, 16-bit MFMA operands), and then again with ONE kind of store kept exact
("site X exact"), for every kind. Rounding errors of different sites are independent to first order, so
    var_removed(X) = err_all^2 - err_without_X^2
is that kind's share of the error variance; the table says which sites a hi/lo operand split (or an fp32 store) must cover to
bring the evaluation inside north_star's 1e-3, and a final run keeps the chosen subset exact together.

usage: python scripts/exp/precision_sites.py [sd15|tiny] [latent side] [--subset a,b,c]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from oracle.unet import UNetCfg, UNetOracle, random_unet_weights

SITES = ["gn_in", "gn1", "conv1_out", "gn2", "shortcut", "tnorm", "ln1", "qkv1", "p1", "o1", "ln2", "q2", "kv2", "p2", "o2", "ln3", "geglu",
         "proj_out_in", "sampler_in", "lnfold_w", "xfold_w", "temb", "mimic"]
DESC = {
    "gn_in": "the stream's 16-bit shadow read by a GroupNorm (ResnetBlock2D.norm1, Transformer2DModel.norm, conv_norm_out)",
    "gn1": "ResnetBlock2D norm1 + SiLU output (conv1's A operand)",
    "conv1_out": "conv1 + temb output (norm2's input; 16-bit, not on the fp32 stream)",
    "gn2": "norm2 + SiLU output (conv2's A operand)",
    "shortcut": "1x1 conv_shortcut: its 16-bit input read and its output",
    "tnorm": "Transformer2DModel.norm output (proj_in's A operand)",
    "ln1": "norm1 output / the stream's 16-bit shadow read by the folded q/k/v projection",
    "qkv1": "self-attention q, k, v stores",
    "p1": "self-attention probabilities as the PV operand",
    "o1": "self-attention output (to_out's A operand)",
    "ln2": "norm2 shadow read by the cross-attention to_q",
    "q2": "cross-attention q store",
    "kv2": "cross-attention K / V of the prompt",
    "p2": "cross-attention probabilities as the PV operand",
    "o2": "cross-attention output (to_out's A operand)",
    "ln3": "norm3 shadow read by the GEGLU projection",
    "geglu": "GEGLU product (ff.net.2's A operand)",
    "proj_out_in": "the stream's shadow read by proj_out",
    "sampler_in": "stream shadows read as conv operands by conv_in / down- / upsamplers / conv_out",
    "lnfold_w": "LayerNorm folded into q/k/v, to_q and the GEGLU projection: W * gamma re-rounded to 16 bits (ops.fold_layernorm)",
    "xfold_w": "fused cross-attention (>= 1024 token rows): the prompt's K / V folded into Mq = scale K Wq gamma and Mo = Wo V^T, both re-rounded",
    "mimic": "op-boundary roundings the kernels add to mirror an f16 module: GroupNorm output before SiLU; GEGLU's value, gate and gelu(gate)",
    "temb": "time-embedding MLP and the per-resnet projections computed and stored in 16 bits (rowbias operand)",
}


class SiteOracle(UNetOracle):
    """f16 + fp32-stream emulation with the sites in `exact` left unrounded"""

    def __init__(self, cfg, w, exact=(), fmt=torch.float16):
        super().__init__(cfg, w)
        self.exact, self.fmt = set(exact), fmt

    def q(self, t, site):
        return t if site in self.exact else t.to(self.fmt).float()

    def resnet(self, n, x, temb):
        a = self.q(F.silu(self.q(self._gn(n + ".norm1", self.q(x, "gn_in")), "mimic")), "gn1")
        tp = self.q(self._lin(n + ".time_emb_proj", self.q(F.silu(temb), "temb")), "temb")
        h = UNetOracle._conv(self, n + ".conv1", a) + tp[:, :, None, None]
        h = self.q(h, "conv1_out")
        a = self.q(F.silu(self.q(self._gn(n + ".norm2", h), "mimic")), "gn2")
        h = UNetOracle._conv(self, n + ".conv2", a)
        if n + ".conv_shortcut.weight" in self.w:
            x = UNetOracle._conv(self, n + ".conv_shortcut", self.q(x, "shortcut"), pad=0)       # fp32 master of the shortcut kept (want32)
        return x + h

    def ln_lin(self, b, k, lin, h, site):
        """LayerNorm(h) @ W^T + bias as the engine computes it: rows of the 16-bit shadow, normalised without the affine, times
        the re-rounded W * gamma, plus (W beta + bias) in fp32"""
        C = h.shape[-1]
        xq = self.q(h, site)
        xh = F.layer_norm(xq, (C,))
        W, g, be = self.w[lin + ".weight"], self.w[b + k + ".weight"], self.w[b + k + ".bias"]
        Wf = self.q(W * g[None, :], "lnfold_w")
        out = xh @ Wf.t() + W @ be
        bias = self.w.get(lin + ".bias")
        return out if bias is None else out + bias

    def xattn_fused(self, b, h, enc, heads):
        """the engine's fused cross-attention sub-block (csrc/xattn_fused.hip): returns to_out(attention) WITHOUT the residual"""
        n = b + ".attn2"
        C = h.shape[-1]
        d = C // heads
        B, N, _ = h.shape
        xh = F.layer_norm(self.q(h, "ln2"), (C,))
        g2, b2 = self.w[b + ".norm2.weight"], self.w[b + ".norm2.bias"]
        K = self.q(self._lin(n + ".to_k", enc), "kv2").view(B, -1, heads, d)      # [B, L, H, d]
        V = self.q(self._lin(n + ".to_v", enc), "kv2").view(B, -1, heads, d)
        Wq = self.w[n + ".to_q.weight"].view(heads, d, C)                            # rows h*d .. of [C, C]
        Wo = self.w[n + ".to_out.0.weight"].view(C, heads, d)
        scale = d ** -0.5
        # the engine builds Mq from the ROUNDED Wq * gamma (wqt_g) and the rounded K
        Wqg = self.q(Wq * g2[None, None, :], "xfold_w")
        Mq = self.q(scale * torch.einsum("blhd,hdc->bhlc", K, Wqg), "xfold_w")
        cb = scale * torch.einsum("blhd,hd->bhl", K, self.q(torch.einsum("hdc,c->hd", Wq, b2), "xfold_w"))
        Mo = self.q(torch.einsum("chd,blhd->bhlc", Wo, V), "xfold_w")
        s_ = torch.einsum("bnc,bhlc->bhnl", xh, Mq) + cb[:, :, None, :]
        m = s_.amax(-1, keepdim=True)
        pr = torch.exp(s_ - m)
        l = pr.sum(-1, keepdim=True)
        o = torch.einsum("bhnl,bhlc->bnc", self.q(pr, "p2") / l, Mo)
        return o + self.w[n + ".to_out.0.bias"]

    def attention(self, n, x, ctx, heads, tag):
        q = self.q(self._lin(n + ".to_q", x), "qkv1" if tag == "1" else "q2")
        ks = "qkv1" if tag == "1" else "kv2"
        k = self.q(self._lin(n + ".to_k", ctx), ks); v = self.q(self._lin(n + ".to_v", ctx), ks)
        B, L, C = q.shape
        d = C // heads
        sh = lambda t: t.view(B, -1, heads, d).transpose(1, 2)
        s = (sh(q) @ sh(k).transpose(-1, -2)) / d ** 0.5
        m = s.amax(-1, keepdim=True)
        p = torch.exp(s - m)
        l = p.sum(-1, keepdim=True)
        o = (self.q(p, "p" + tag) @ sh(v)) / l
        o = self.q(o.transpose(1, 2).reshape(B, L, C), "o" + tag)
        return self._lin(n + ".to_out.0", o)

    def _attn_core(self, q, k, v, heads, tag):
        B, L, C = q.shape
        d = C // heads
        sh = lambda t: t.view(B, -1, heads, d).transpose(1, 2)
        s_ = (sh(q) @ sh(k).transpose(-1, -2)) / d ** 0.5
        m = s_.amax(-1, keepdim=True)
        pr = torch.exp(s_ - m)
        l = pr.sum(-1, keepdim=True)
        o = (self.q(pr, "p" + tag) @ sh(v)) / l
        return self.q(o.transpose(1, 2).reshape(B, L, C), "o" + tag)

    def attention_folded(self, b, h, heads):       # attn1: norm1 folded into the fused q/k/v projection
        n = b + ".attn1"
        q = self.q(self.ln_lin(b, ".norm1", n + ".to_q", h, "ln1"), "qkv1")
        k = self.q(self.ln_lin(b, ".norm1", n + ".to_k", h, "ln1"), "qkv1")
        v = self.q(self.ln_lin(b, ".norm1", n + ".to_v", h, "ln1"), "qkv1")
        return self._lin(n + ".to_out.0", self._attn_core(q, k, v, heads, "1"))

    def attention_q_folded(self, b, h, enc, heads):     # attn2 below the fused kernel's size: norm2 folded into to_q
        n = b + ".attn2"
        q = self.q(self.ln_lin(b, ".norm2", n + ".to_q", h, "ln2"), "q2")
        k = self.q(self._lin(n + ".to_k", enc), "kv2"); v = self.q(self._lin(n + ".to_v", enc), "kv2")
        return self._lin(n + ".to_out.0", self._attn_core(q, k, v, heads, "2"))

    def time_embed(self, t, B, added=None, class_labels=None):
        if self.cfg.addition_in or self.cfg.class_in:
            return self.q(super().time_embed(t, B, added, class_labels), "temb")
        from oracle.unet import timestep_embedding
        te = self.q(timestep_embedding(t.expand(B) if t.ndim == 0 else t, self.cfg.block_out[0]), "temb")
        h1 = self.q(F.silu(self._lin("time_embedding.linear_1", te)), "temb")
        return self.q(self._lin("time_embedding.linear_2", h1), "temb")

    def transformer(self, n, x, enc, heads, depth):
        B, C, H, W = x.shape
        res = x
        h = self.q(self._gn(n + ".norm", self.q(x, "gn_in"), eps=1e-6), "tnorm")
        if self.cfg.linear_proj:
            h = self._lin(n + ".proj_in", h.permute(0, 2, 3, 1).reshape(B, H * W, C))
        else:
            h = UNetOracle._conv(self, n + ".proj_in", h, pad=0).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for d in range(depth):
            b = f"{n}.transformer_blocks.{d}"
            # the engine folds LayerNorm into the consuming GEMM: the MFMA operand is the 16-bit SHADOW of the stream (rounded before
            # the normalisation); mean / rstd come from those rounded values too
            h = self.attention_folded(b, h, heads) + h
            if enc is not None and B * H * W >= 1024 and heads == 8:
                h = self.xattn_fused(b, h, enc, heads) + h
            else:
                h = self.attention_q_folded(b, h, enc, heads) + h
            p = self.ln_lin(b, ".norm3", b + ".ff.net.0.proj", h, "ln3")
            a, gate = p.chunk(2, -1)
            g = self.q(self.q(a, "mimic") * self.q(F.gelu(self.q(gate, "mimic")), "mimic"), "geglu")
            h = self._lin(b + ".ff.net.2", g) + h
        hq = self.q(h, "proj_out_in")
        if self.cfg.linear_proj:
            h = self._lin(n + ".proj_out", hq).reshape(B, H, W, C).permute(0, 3, 1, 2)
        else:
            h = UNetOracle._conv(self, n + ".proj_out", hq.reshape(B, H, W, C).permute(0, 3, 1, 2), pad=0)
        return h + res

    def _conv(self, n, x, stride=1, pad=1):
        # reached only from forward(): conv_in, the samplers, conv_out -- stream tensors read as 16-bit operands
        return UNetOracle._conv(self, n, self.q(x, "sampler_in"), stride, pad)

    @torch.no_grad()
    def forward(self, sample, t, enc, added=None, class_labels=None):
        # GroupNorm inputs are the 16-bit shadows as well (gn reads the shadow): fold that into the gn sites by rounding the
        # resnet / transformer inputs where the engine's GroupNorm reads them
        return super().forward(sample, t, enc, added, class_labels)


def run(cfg, hw, subset=None, seed=0):
    w = random_unet_weights(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, cfg.in_ch, hw, hw, generator=g).bfloat16().float()
    enc = torch.randn(2, 77, cfg.cross_dim, generator=g).bfloat16().float()
    added = None
    if cfg.addition_in:
        added = dict(text_embeds=torch.randn(2, 1280, generator=g).bfloat16().float(),
                     time_ids=torch.tensor([[512, 512, 0, 0, 512, 512]] * 2, dtype=torch.float32))
    t = torch.tensor(500)
    t0 = time.time()
    ref = UNetOracle(cfg, w).forward(x, t, enc, added)
    print(f"# fp32 oracle {time.time() - t0:.1f}s   latent {hw}x{hw}, CFG batch 2", flush=True)
    rel = lambda exact: float((SiteOracle(cfg, w, exact).forward(x, t, enc, added) - ref).norm() / ref.norm())
    e_all = rel(())
    print(f"all sites rounded (the engine's design): rel-L2 {e_all:.3e}", flush=True)
    rows = []
    for s in SITES:
        e = rel((s,))
        share = (e_all ** 2 - e ** 2) / e_all ** 2
        rows.append((share, s, e))
        print(f"  exact {s:12s} rel-L2 {e:.3e}   variance share {100 * share:5.1f} %   {DESC[s]}", flush=True)
    rows.sort(reverse=True)
    print("sorted by share:", ", ".join(f"{s} {100 * sh:.1f}%" for sh, s, _ in rows))
    acc = []
    for sh, s, _ in rows[:8]:
        acc.append(s)
        print(f"  exact top-{len(acc)} {{{','.join(acc)}}}: rel-L2 {rel(tuple(acc)):.3e}", flush=True)
    if subset:
        print(f"  exact subset {{{','.join(subset)}}}: rel-L2 {rel(tuple(subset)):.3e}", flush=True)
    print(f"  exact ALL sites (sanity, must be ~0): rel-L2 {rel(tuple(SITES)):.3e}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    subset = None
    for a in sys.argv:
        if a.startswith("--subset"):
            subset = a.split("=", 1)[1].split(",")
    if which == "tiny":
        run(UNetCfg.tiny8(), 16, subset, seed=1)
    elif which == "sdxl":
        run(UNetCfg.sdxl(), int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 64, subset, seed=4)
    else:
        run(UNetCfg.sd15(), int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 64, subset)
