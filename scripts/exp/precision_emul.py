"""Precision study (CPU, test-side aid): the UNet oracle graph with every tensor the HIP engine stores in HBM rounded to a
16-bit format (bf16 or f16) and, optionally, the residual stream kept in fp32. Prints rel-L2 against the fp32 oracle for the
2 x 2 table {bf16, f16} x {16-bit stream, fp32 stream} that DESIGN.md section 4 quotes next to north_star's 1e-3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from oracle.unet import UNetCfg, UNetOracle, random_unet_weights


class EmulOracle(UNetOracle):
    def __init__(self, cfg, w, fmt, stream32):
        super().__init__(cfg, w)
        self.fmt, self.stream32 = fmt, stream32

    def q(self, t):          # what a store to HBM in the 16-bit format does
        return t.to(self.fmt).float()

    def qs(self, t):         # residual-stream store
        return t if self.stream32 else self.q(t)

    def _gn(self, n, x, eps=1e-5):
        return super()._gn(n, x, eps)

    def resnet(self, n, x, temb):
        a = self.q(F.silu(self._gn(n + ".norm1", x)))
        h = self._conv(n + ".conv1", a) + self._lin(n + ".time_emb_proj", F.silu(temb))[:, :, None, None]
        h = self.q(h)
        a = self.q(F.silu(self._gn(n + ".norm2", h)))
        h = self._conv(n + ".conv2", a)
        if n + ".conv_shortcut.weight" in self.w:
            x = self.q(self._conv(n + ".conv_shortcut", self.q(x), pad=0))
        return self.qs(x + h)

    def attention(self, n, x, ctx, heads):
        q = self.q(self._lin(n + ".to_q", x)); k = self.q(self._lin(n + ".to_k", ctx)); v = self.q(self._lin(n + ".to_v", ctx))
        B, L, C = q.shape
        d = C // heads
        sh = lambda t: t.view(B, -1, heads, d).transpose(1, 2)
        s = (sh(q) @ sh(k).transpose(-1, -2)) / d ** 0.5
        m = s.amax(-1, keepdim=True)
        p = torch.exp(s - m)
        l = p.sum(-1, keepdim=True)
        o = (self.q(p) @ sh(v)) / l            # P rounded to the operand format inside the flash kernel
        o = self.q(o.transpose(1, 2).reshape(B, L, C))
        return self._lin(n + ".to_out.0", o)

    def transformer(self, n, x, enc, heads, depth):
        B, C, H, W = x.shape
        res = x
        h = self.q(self._gn(n + ".norm", x, eps=1e-6))
        if self.cfg.linear_proj:
            h = self._lin(n + ".proj_in", h.permute(0, 2, 3, 1).reshape(B, H * W, C))
        else:
            h = self._conv(n + ".proj_in", h, pad=0).permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = self.qs(h)
        for d in range(depth):
            b = f"{n}.transformer_blocks.{d}"
            ln = lambda k, t: self.q(F.layer_norm(t, (C,), self.w[b + k + ".weight"], self.w[b + k + ".bias"], 1e-5))
            y = ln(".norm1", h)
            h = self.qs(self.attention(b + ".attn1", y, y, heads) + h)
            y = ln(".norm2", h)
            h = self.qs(self.attention(b + ".attn2", y, enc if enc is not None else y, heads) + h)
            y = ln(".norm3", h)
            p = self._lin(b + ".ff.net.0.proj", y)
            a, gate = p.chunk(2, -1)
            g = self.q(a * F.gelu(gate))
            h = self.qs(self._lin(b + ".ff.net.2", g) + h)
        hq = self.q(h)
        if self.cfg.linear_proj:
            h = self._lin(n + ".proj_out", hq).reshape(B, H, W, C).permute(0, 3, 1, 2)
        else:
            h = self._conv(n + ".proj_out", hq.reshape(B, H, W, C).permute(0, 3, 1, 2), pad=0)
        return self.qs(h + res)

    def _conv(self, n, x, stride=1, pad=1):
        # conv operands are 16-bit: a stream tensor read as an operand is rounded on the way in
        out = super()._conv(n, self.q(x), stride, pad)
        if n in ("conv_in",) or "samplers" in n:
            out = self.qs(out)
        return out


def run(cfg, hw, seed=0):
    w = random_unet_weights(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, cfg.in_ch, hw, hw, generator=g).bfloat16().float()
    enc = torch.randn(2, 77, cfg.cross_dim, generator=g).bfloat16().float()
    t = torch.tensor(500)
    t0 = time.time()
    ref = UNetOracle(cfg, w).forward(x, t, enc)
    print(f"fp32 oracle {time.time() - t0:.1f}s", flush=True)
    for fmt in (torch.bfloat16, torch.float16):
        for s32 in (False, True):
            got = EmulOracle(cfg, w, fmt, s32).forward(x, t, enc)
            rel = float((got - ref).norm() / ref.norm())
            print(f"{str(fmt):16s} stream32={s32!s:5s} rel-L2 {rel:.3e}", flush=True)
        plain = UNetOracle(cfg, w, dtype=fmt).forward(x, t, enc)
        print(f"{str(fmt):16s} torch-native   rel-L2 {float((plain - ref).norm() / ref.norm()):.3e}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    if which == "tiny":
        run(UNetCfg.tiny(), 16)
        run(UNetCfg.tiny8(), 16, seed=1)
    else:
        run(UNetCfg.sd15(), int(sys.argv[2]) if len(sys.argv) > 2 else 32)
