"""Randomised differential check of the GEMM / conv / attention / GroupNorm front ends (spider_amd.ops -> C ABI -> HIP kernels) against
fp32 torch arithmetic on the CPU, over shapes the fixed test matrix does not enumerate: ragged M / N / K, odd image sizes, strides,
dilations, asymmetric pads, nearest-upsampled inputs, every epilogue combination, both 16-bit formats, the fp32 residual stream and the
producer-side GroupNorm statistics. Not part of the suite (its shapes are random); run on the GPU box:

    PYTHONPATH=. python scripts/fuzz_ops.py [cases] [seed]          # default 300 cases, seed 0

Prints one line per failing case (with the arguments to reproduce it) and a summary; exit code 1 if anything failed."""
import sys
import random

import torch
import torch.nn.functional as F

from spider_amd import ops

dev = torch.device("cuda:0")
DTS = (torch.float16, torch.bfloat16)


def rnd(g, *shape, scale=1.0, dt=torch.float16):
    return (torch.randn(*shape, generator=g) * scale).to(dt)


def err(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    if got.shape != ref.shape:
        return float("inf"), f"shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    if not torch.isfinite(got).all():
        return float("inf"), "non-finite output"
    return float((got - ref).norm() / (ref.norm() + 1e-12)), ""


def case_gemm(r, g):
    dt = r.choice(DTS)
    M = r.choice([1, 2, 7, 16, 33, 64, 100, 128, 257, 512, 777, 1024, 2048, 3000, 4096, 8192])
    N = r.choice([8, 24, 64, 72, 160, 320, 328, 640, 1000, 1280, 2560])
    K = r.choice([8, 32, 64, 72, 320, 640, 1096, 1280, 2560, 5120])
    act = r.choice([None, None, "silu", "gelu", "quick_gelu", "relu", "tanh"])
    use_bias, use_res, s32 = r.random() < 0.7, r.random() < 0.4, r.random() < 0.3
    rows_per_group = r.choice([0, 0, max(1, M // 2)])
    A, W = rnd(g, M, K, dt=dt), rnd(g, N, K, scale=K ** -0.5, dt=dt)
    b = rnd(g, N, dt=dt) if use_bias else None
    res = rnd(g, M, N, dt=dt) if use_res and not s32 else None
    r32 = torch.randn(M, N, generator=g) if s32 else None
    rb = rnd(g, (M + rows_per_group - 1) // rows_per_group, N, dt=dt) if rows_per_group else None
    desc = f"gemm dt={dt} M={M} N={N} K={K} act={act} bias={use_bias} res={use_res} s32={s32} rpg={rows_per_group}"
    kw = dict(bias=None if b is None else b.to(dev), act=act, res=None if res is None else res.to(dev),
              rowbias=None if rb is None else rb.to(dev), rows_per_group=rows_per_group)
    if s32:
        y, y32 = ops.gemm(A.to(dev), W.to(dev), res32=r32.to(dev), want32=True, **kw)
    else:
        y, y32 = ops.gemm(A.to(dev), W.to(dev), **kw), None
    ref = A.float() @ W.float().T
    if b is not None:
        ref = ref + b.float()
    if rb is not None:
        ref = ref + rb.float()[torch.arange(M) // rows_per_group]
    ref = {None: lambda t: t, "silu": F.silu, "gelu": F.gelu, "quick_gelu": lambda t: t * torch.sigmoid(1.702 * t), "relu": F.relu,
           "tanh": torch.tanh}[act](ref)
    if res is not None:
        ref = ref.to(dt).float() + res.float()
    if s32:
        ref = ref + r32
    e, why = err(y, ref)
    if y32 is not None and not why:
        e2, why = err(y32, ref)
        e = max(e, e2 * 4)            # the fp32 master must be much closer than the 16-bit shadow
    return desc, e, why, dt


def case_conv(r, g):
    dt = r.choice(DTS)
    B = r.choice([1, 2, 3])
    H, W = r.choice([5, 8, 9, 16, 17, 24, 32, 33, 64]), r.choice([4, 8, 11, 16, 20, 32, 36, 64])
    Cin = r.choice([8, 16, 24, 32, 64, 96, 128, 192, 320, 640])
    Cout = r.choice([8, 16, 32, 40, 64, 160, 320, 640])
    kh, kw = r.choice([(3, 3), (3, 3), (1, 1), (3, 1), (1, 3), (1, 7), (5, 5)])
    stride = r.choice([1, 1, 2])
    dil = r.choice([1, 1, 1, 2, 3]) if stride == 1 else 1
    pad = (r.choice([0, dil * (kh // 2)]), r.choice([0, dil * (kw // 2)]))
    up = None
    if r.random() < 0.2 and stride == 1:
        up = (2 * H - r.choice([0, 1]), 2 * W - r.choice([0, 1]))
    Hs, Ws = up if up else (H, W)
    if Hs + 2 * pad[0] < dil * (kh - 1) + 1 or Ws + 2 * pad[1] < dil * (kw - 1) + 1:
        return None
    act = r.choice([None, None, None, "silu", "leaky_relu", "tanh"])
    gn = r.random() < 0.35 and Cout % 32 == 0 and act is None
    s32 = r.random() < 0.3 and act is None
    x, w, b = rnd(g, B, H, W, Cin, dt=dt), rnd(g, Cout, kh, kw, Cin, scale=(kh * kw * Cin) ** -0.5, dt=dt), rnd(g, Cout, dt=dt)
    desc = f"conv dt={dt} B={B} H={H} W={W} Cin={Cin} Cout={Cout} k=({kh},{kw}) stride={stride} pad={pad} dil={dil} up={up} act={act} gn={gn} s32={s32}"
    xi = x.float().permute(0, 3, 1, 2)
    if up is not None:
        xi = F.interpolate(xi, size=up, mode="nearest")
    ref = F.conv2d(xi, w.float().permute(0, 3, 1, 2), b.float(), stride=stride, padding=pad, dilation=dil).permute(0, 2, 3, 1)
    ref = {None: lambda t: t, "silu": F.silu, "leaky_relu": lambda t: F.leaky_relu(t, 0.1), "tanh": torch.tanh}[act](ref)
    r32 = torch.randn(ref.shape, generator=g) if s32 else None
    if s32:
        ref = ref + r32
    out = ops.conv_ex(x.to(dev), w.to(dev), bias=b.to(dev), stride=stride, pad=pad, dil=dil, up_size=up, act=act, act_param=0.1,
                      res32=None if r32 is None else r32.to(dev), want32=s32, gn_groups=32 if gn else None)
    out = out if isinstance(out, tuple) else (out,)
    y = out[0]
    e, why = err(y, ref)
    if s32 and not why:
        e2, why = err(out[1], ref)
        e = max(e, e2 * 4)
    if gn and not why and out[-1] is not None:       # producer statistics == statistics of the stored 16-bit output
        part = out[-1]
        got = part.t.float().sum(1).cpu()                                  # [B, G, 2]
        yy = y.float().cpu().reshape(B, -1, 32, Cout // 32)
        want = torch.stack([yy.sum((1, 3)), (yy * yy).sum((1, 3))], -1)
        e3 = float((got - want).abs().max() / (want.abs().max() + 1e-12))
        if e3 > 1e-4:
            why = f"GroupNorm partials off by {e3:.2e}"
    return desc, e, why, dt


def case_attn(r, g):
    dt = r.choice(DTS)
    B, heads = r.choice([1, 2, 3]), r.choice([1, 2, 5, 8])
    d = r.choice([32, 40, 64, 80, 128, 160])
    Lq = r.choice([1, 7, 16, 64, 77, 100, 256, 1000, 1024])
    Lk = r.choice([1, 16, 77, 80, 100, 256, 1024, 1500])
    q, k, v = rnd(g, B, Lq, heads * d, dt=dt), rnd(g, B, Lk, heads * d, dt=dt), rnd(g, B, Lk, heads * d, dt=dt)
    desc = f"attn dt={dt} B={B} heads={heads} d={d} Lq={Lq} Lk={Lk}"
    y = ops.attention(q.to(dev), k.to(dev), v.to(dev), heads)
    sp = lambda t, L: t.float().view(B, L, heads, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q, Lq), sp(k, Lk), sp(v, Lk)).transpose(1, 2).reshape(B, Lq, heads * d)
    e, why = err(y, ref)
    return desc, e, why, dt


def case_groupnorm(r, g):
    dt = r.choice(DTS)
    B, HW = r.choice([1, 2, 3]), r.choice([1, 9, 64, 100, 256, 1000, 4096])
    G = 32
    C = G * r.choice([1, 2, 4, 10, 20, 40])
    silu = r.random() < 0.5
    if HW * (C // G) == 1:
        return None                  # one value per group: torch's own group_norm refuses the case
    x = rnd(g, B, HW, C, scale=2.0, dt=dt) + 0.5
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(dt), (0.1 * torch.randn(C, generator=g)).to(dt)
    desc = f"groupnorm dt={dt} B={B} HW={HW} C={C} silu={silu}"
    y = ops.groupnorm(x.to(dev).view(B, HW, 1, C) if r.random() < 0.5 else x.to(dev), gamma.to(dev), beta.to(dev), G, 1e-5, silu)
    ref = F.group_norm(x.float().transpose(1, 2), G, gamma.float(), beta.float(), 1e-5).transpose(1, 2)
    ref = F.silu(ref) if silu else ref
    e, why = err(y.reshape(B, HW, C), ref)
    return desc, e, why, dt


def _rms(x, w, eps):
    xf = x.float()
    return ((xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(torch.bfloat16).float() * w.float()).to(torch.bfloat16)


def case_llm(r, g):
    """decode-side operators (bf16): LlamaRMSNorm (+ residual), weight-streaming GEMV with the norm / bias / residual fused,
    gate-up GEMV with SwiGLU, lm_head + argmax, and the prefill GEMM's SwiGLU epilogue"""
    bf = torch.bfloat16
    kind = r.choice(["rmsnorm", "gemv", "gemv_swiglu", "lm_head", "gemm_swiglu"])
    K = r.choice([64, 128, 256, 512, 1000, 1024, 3584, 4096])
    B = r.choice([1, 1, 2, 3, 4, 8])
    eps = r.choice([1e-5, 1e-6])
    x = rnd(g, B, K, dt=bf)
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(bf)
    if kind == "rmsnorm":
        rows = r.choice([1, 3, 64, 300])
        x = rnd(g, rows, K, dt=bf)
        res = rnd(g, rows, K, dt=bf) if r.random() < 0.5 else None
        desc = f"rmsnorm rows={rows} H={K} eps={eps} res={res is not None}"
        if res is None:
            y = ops.rmsnorm(x.to(dev), nw.to(dev), eps)
            ref = _rms(x, nw, eps)
        else:
            ro = torch.empty(rows, K, dtype=bf, device=dev)
            y = ops.rmsnorm(x.to(dev), nw.to(dev), eps, res=res.to(dev), res_out=ro)
            h = (x.float() + res.float()).to(bf)
            ref = _rms(h, nw, eps)
            if not torch.equal(ro.cpu(), h):
                return desc, float("inf"), "residual sum differs from bf16(x + res)", bf
        e, why = err(y, ref)
        return desc, e, why, bf
    if kind == "gemv":
        N = r.choice([8, 64, 100, 512, 1000, 4608])
        W = rnd(g, N, K, scale=K ** -0.5, dt=bf)
        bias = rnd(g, N, dt=bf) if r.random() < 0.5 else None
        res = rnd(g, B, N, dt=bf) if r.random() < 0.5 else None
        fuse = r.random() < 0.5
        desc = f"gemv B={B} N={N} K={K} bias={bias is not None} res={res is not None} norm={fuse}"
        y = ops.gemv(W.to(dev), x.to(dev), bias=None if bias is None else bias.to(dev), res=None if res is None else res.to(dev),
                     norm_w=nw.to(dev) if fuse else None, eps=eps)
        xin = _rms(x, nw, eps) if fuse else x
        ref = xin.float() @ W.float().T
        if bias is not None:
            ref = ref + bias.float()
        if res is not None:
            ref = ref + res.float()
        e, why = err(y, ref)
        return desc, e, why, bf
    if kind in ("gemv_swiglu", "gemm_swiglu"):
        I = r.choice([64, 256, 1000, 2048])
        W = rnd(g, 2 * I, K, scale=K ** -0.5, dt=bf)
        if kind == "gemm_swiglu":
            M = r.choice([16, 100, 300, 1536])
            x = rnd(g, M, K, dt=bf)
            desc = f"gemm act=swiglu M={M} I={I} K={K}"
            y = ops.gemm(x.to(dev), W.to(dev), act="swiglu")
        else:
            fuse = r.random() < 0.5
            desc = f"gemv_swiglu B={B} I={I} K={K} norm={fuse}"
            y = ops.gemv_swiglu(W.to(dev), x.to(dev), norm_w=nw.to(dev) if fuse else None, eps=eps)
            x = _rms(x, nw, eps) if fuse else x
        p_ = x.float() @ W.float().T
        gate, up = p_[:, :I].to(bf).float(), p_[:, I:].to(bf).float()
        ref = F.silu(gate).to(bf).float() * up
        e, why = err(y, ref)
        return desc, e, why, bf
    V = r.choice([300, 1000, 32000])
    W = rnd(g, V, K, scale=K ** -0.5, dt=bf)
    desc = f"lm_head_argmax B={B} V={V} K={K}"
    logits = torch.empty(B, V, dtype=bf, device=dev)
    ids = ops.lm_head_argmax(W.to(dev), x.to(dev), norm_w=nw.to(dev), eps=eps, logits=logits)
    ref = _rms(x, nw, eps).float() @ W.float().T
    e, why = err(logits, ref)
    top2 = ref.topk(2, -1).values
    for b in range(B):        # the id must be the argmax wherever the decision is above bf16 resolution
        if float(top2[b, 0] - top2[b, 1]) > 0.05 * float(ref[b].abs().max()) and int(ids[b]) != int(ref[b].argmax()):
            why = f"row {b}: id {int(ids[b])} is not the argmax {int(ref[b].argmax())}"
    if not why and not torch.equal(ids.cpu().long(), logits.float().cpu().argmax(-1)):
        why = "ids differ from the argmax of the kernel's own logits"
    return desc, e, why, bf


def case_wstream(r, g):
    """round 5: the weight-stationary streaming conv (marked 3 x 3 weight, <= 512 output pixels) with the resnet epilogue operands"""
    dt = r.choice(DTS)
    B = r.choice([1, 2, 2, 3, 4])
    H, W = r.choice([2, 3, 4, 5, 7, 8, 8, 11, 16]), r.choice([2, 3, 4, 6, 8, 8, 9, 16])
    up = r.random() < 0.25
    Ho, Wo = (2 * H - r.choice([0, 1]), 2 * W - r.choice([0, 1])) if up else (H, W)
    if B * Ho * Wo > 512 or B * H * W > (128 if B * Ho * Wo <= 128 else 512) or (up and B * Ho * Wo <= 128):
        return None
    Cin, Cout = r.choice([32, 64, 96, 160, 320, 640, 1280]), r.choice([4, 8, 32, 36, 64, 80, 320, 1280])
    x, w = rnd(g, B, H, W, Cin, dt=dt), rnd(g, Cout, 3, 3, Cin, scale=(9 * Cin) ** -0.5, dt=dt)
    b, rb = rnd(g, Cout, dt=dt), rnd(g, B, Cout, dt=dt)
    s32 = r.random() < 0.5
    r32 = torch.randn(B, Ho, Wo, Cout, generator=g) if s32 else None
    res = rnd(g, B, Ho, Wo, Cout, dt=dt) if (not s32 and r.random() < 0.5) else None
    old = ops.WS_MAX_M
    ops.WS_MAX_M = 512
    try:
        wm = ops.mark_weight(w.to(dev))
        out = ops.conv_ex(x.to(dev), wm, bias=b.to(dev), rowbias=rb.to(dev), pad=(1, 1), up_size=(Ho, Wo) if up else None,
                          res=None if res is None else res.to(dev), res32=None if r32 is None else r32.to(dev), want32=s32)
    finally:
        ops.WS_MAX_M = old
    desc = f"wstream dt={dt} B={B} {H}x{W}->{Ho}x{Wo} Cin={Cin} Cout={Cout} s32={s32} res={res is not None}"
    if getattr(wm, "_spider_fm", None) is None:
        return desc, float("inf"), "the streaming kernel's weight copy was not built (wrong path)", dt
    xi = x.float().permute(0, 3, 1, 2)
    if up:
        xi = F.interpolate(xi, size=(Ho, Wo), mode="nearest")
    ref = F.conv2d(xi, w.float().permute(0, 3, 1, 2), b.float(), padding=1).permute(0, 2, 3, 1) + rb.float()[:, None, None, :]
    if res is not None:
        ref = ref.to(dt).float() + res.float()
    if s32:
        ref = ref + r32
        e, why = err(out[0], ref)
        e2, why2 = err(out[1], ref)
        return desc, max(e, 100 * e2), why or why2, dt            # the fp32 master: accumulation error only
    e, why = err(out, ref)
    return desc, e, why, dt


def case_a32(r, g):
    """round 5: the fp32-operand (hi / lo split) forms -- gemm_a32, gemm_ln_a32 (+ GEGLU), conv_a32, groupnorm_f32in. The fp32 output
    must sit at fp32-accumulation distance from the reference although W is 16-bit: the A operand carries ~22 bits."""
    dt = r.choice(DTS)
    kind = r.choice(["gemm", "ln", "ln_geglu", "conv", "gn"])
    # route: the a32 kernels, or "split once, doubled K" (marked weight + the flop threshold at 0 for this call)
    dup = r.random() < 0.5
    ops.A32_DUP_MIN_FLOP, ops.A32_DUP_MIN_M = (0.0 if dup else float("inf")), 0
    mk = ops.mark_weight if dup else (lambda t: t)
    if kind == "gn":
        B, HW, C = r.choice([1, 2, 3]), r.choice([16, 60, 64, 256, 1000, 4096]), r.choice([32, 64, 320, 640, 1280])
        x = torch.randn(B, HW, C, generator=g) * 2 + 0.5
        ga, be = rnd(g, C, dt=dt) * 0.1 + 1, rnd(g, C, dt=dt) * 0.1
        silu = r.random() < 0.5
        y, y32 = ops.groupnorm_f32in(x.to(dev), ga.to(dev), be.to(dev), 32, 1e-5, silu, want16=True, want32=True)
        ref = F.group_norm(x.transpose(1, 2), 32, ga.float(), be.float(), 1e-5).transpose(1, 2)
        ref = F.silu(ref) if silu else ref
        e, why = err(y, ref)
        e2, why2 = err(y32, ref)
        return f"a32 groupnorm dt={dt} B={B} HW={HW} C={C} silu={silu}", max(e, 100 * e2), why or why2, dt
    if kind == "conv":
        B, H, W = r.choice([1, 2]), r.choice([4, 8, 9, 16, 32]), r.choice([4, 8, 11, 16, 32])
        Cin, Cout, ks, stride = r.choice([8, 64, 320, 640]), r.choice([8, 64, 320]), r.choice([1, 3]), r.choice([1, 1, 2])
        up = (2 * H, 2 * W) if (ks == 3 and stride == 1 and r.random() < 0.3) else None
        x = torch.randn(B, H, W, Cin, generator=g)
        w, b = rnd(g, Cout, ks, ks, Cin, scale=(ks * ks * Cin) ** -0.5, dt=dt), rnd(g, Cout, dt=dt)
        y, y32 = ops.conv_a32(x.to(dev), mk(w.to(dev)), bias=b.to(dev), stride=stride, pad=(ks // 2, ks // 2), up_size=up, want32=True)
        xi = x.permute(0, 3, 1, 2)
        if up:
            xi = F.interpolate(xi, size=up, mode="nearest")
        ref = F.conv2d(xi, w.float().permute(0, 3, 1, 2), b.float(), stride=stride, padding=ks // 2).permute(0, 2, 3, 1)
        e, why = err(y, ref)
        e2, why2 = err(y32, ref)
        return f"a32 conv dt={dt} dup={dup} B={B} {H}x{W} Cin={Cin} Cout={Cout} ks={ks} s={stride} up={up}", max(e, 30 * e2), why or why2, dt
    M, N, K = r.choice([16, 64, 100, 512, 2048, 8192]), r.choice([8, 64, 320, 640, 1280, 2560]), r.choice([64, 320, 640, 1280])
    A = torch.randn(M, K, generator=g) * 1.5 + r.choice([0.0, 2.0])
    W = rnd(g, N, K, scale=K ** -0.5, dt=dt)
    b = rnd(g, N, dt=dt)
    if kind == "gemm":
        r32 = torch.randn(M, N, generator=g)
        y, y32 = ops.gemm_a32(A.to(dev), mk(W.to(dev)), bias=b.to(dev), res32=r32.to(dev), want32=True)
        ref = A @ W.float().T + b.float() + r32
        e, why = err(y, ref)
        e2, why2 = err(y32, ref)
        return f"a32 gemm dt={dt} dup={dup} M={M} N={N} K={K}", max(e, 30 * e2), why or why2, dt
    ga, be = rnd(g, K, dt=dt) * 0.1 + 1, rnd(g, K, dt=dt) * 0.1
    geglu = kind == "ln_geglu"
    if geglu and N % 8:
        return None
    fold = ops.fold_layernorm_exact(mk(W.to(dev)), ga.to(dev), be.to(dev), b.to(dev))
    y = ops.gemm_ln_a32(A.to(dev), *fold, act="geglu_exact" if geglu else None)
    ref = F.layer_norm(A, (K,), ga.float(), be.float(), 1e-5) @ W.float().T + b.float()
    if geglu:
        v, gt = ref.chunk(2, -1)
        ref = v * F.gelu(gt)
    e, why = err(y, ref)
    return f"a32 {kind} dt={dt} dup={dup} M={M} N={N} K={K}", e, why, dt


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    r, g = random.Random(seed), torch.Generator().manual_seed(seed)
    kinds = [case_gemm] * 4 + [case_conv] * 4 + [case_attn] * 2 + [case_groupnorm] + [case_llm] * 3 + [case_wstream] * 3 + [case_a32] * 3
    if len(sys.argv) > 3:            # only the kinds named on the command line (comma separated: wstream,a32,...)
        kinds = [globals()["case_" + k] for k in sys.argv[3].split(",")]
    tol = {torch.float16: 4e-3, torch.bfloat16: 2.5e-2}
    bad, done, worst, refused = 0, 0, {}, {}
    for i in range(n):
        fn = r.choice(kinds)
        try:
            out = fn(r, g)
        except Exception as ex:                       # a raise is a finding too, unless the library refuses the shape by contract
            msg = str(ex).splitlines()[0][:200]
            if type(ex).__name__ == "SpiderHipError" and any(t in msg for t in ("unsupported", "must be", "multiple of", "bad shape", "head dim")):
                refused[msg] = refused.get(msg, 0) + 1
                continue
            print(f"RAISED {fn.__name__} case {i}: {type(ex).__name__}: {msg}", flush=True)
            bad += 1
            continue
        if out is None:
            continue
        desc, e, why, dt = out
        done += 1
        worst[fn.__name__] = max(worst.get(fn.__name__, 0.0), e if e != float("inf") else 0.0)
        if why or e > tol[dt]:
            bad += 1
            print(f"FAIL case {i}: {desc}: rel {e:.3e} {why}", flush=True)
    torch.cuda.synchronize()
    print(f"fuzz: {done} cases checked, {bad} failed (seed {seed}); worst relative L2 per kind: " +
          ", ".join(f"{k[5:]} {v:.2e}" for k, v in sorted(worst.items())))
    for m, c in refused.items():
        print(f"refused by contract x{c}: {m}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
