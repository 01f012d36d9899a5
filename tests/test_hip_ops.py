"""GPU parity tests: every C-ABI kernel (called through spider_amd.ops -> ctypes -> libspider_hip.so) against
the fp32 CPU oracle / plain torch fp32 restatement on the same seeded inputs.

Tolerances: outputs are bf16 (8 mantissa bits, eps = 2^-8 ~ 3.9e-3); inputs are pre-rounded to bf16 so the
only differences are accumulation order and the output rounding. atol/rtol are written per test.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def close(got, ref, atol, rtol, what="", rel_to_std=False):
    """|got - ref| <= atol (* std(ref) if rel_to_std) + rtol * |ref| elementwise."""
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = (got - ref).abs()
    if rel_to_std:
        atol = atol * float(ref.std()) if ref.numel() > 1 else atol
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bool(bad.any()), f"{what}: max err {float(err.max()):.4g} (allowed {float(tol[err.argmax()] if err.ndim == 0 else tol.flatten()[err.flatten().argmax()]):.4g}), {int(bad.sum())} / {bad.numel()} elements out of tolerance"


@pytest.fixture(scope="module")
def ops(dev):
    from spider_amd import ops as o
    return o


# ------------------------------------------------------------------------------------------ LLM decode kernels
@pytest.mark.parametrize("rows,H", [(1, 3584), (5, 4096), (3, 64), (2, 8192)])
def test_rmsnorm(ops, dev, rows, H):
    from oracle.llama import rmsnorm
    x, w, r = rnd(rows, H, seed=1), (1 + 0.1 * rnd(H, seed=2).float()).to(BF), rnd(rows, H, seed=3)
    y = ops.rmsnorm(x.to(dev), w.to(dev), 1e-6)
    close(y, rmsnorm(x.float(), w.float(), 1e-6), 1e-2, 1e-2, "rmsnorm")
    ro = torch.empty_like(x, device=dev)
    y2 = ops.rmsnorm(x.to(dev), w.to(dev), 1e-6, res=r.to(dev), res_out=ro)
    hsum = (x.float() + r.float()).to(BF)
    assert torch.equal(ro.cpu(), hsum)  # bf16 + bf16 -> bf16 is exact-by-definition
    close(y2, rmsnorm(hsum.float(), w.float(), 1e-6), 1e-2, 1e-2, "rmsnorm+res")


@pytest.mark.parametrize("B,N,K", [(1, 4608, 3584), (1, 3584, 18944), (2, 512, 4096), (8, 130, 1024), (3, 7, 64), (8, 64, 18944)])
def test_gemv(ops, dev, B, N, K):
    from oracle.llama import rmsnorm
    W, x = rnd(N, K, seed=1, scale=0.05), rnd(B, K, seed=2)
    bias, res, nw = rnd(N, seed=3), rnd(B, N, seed=4), (1 + 0.1 * rnd(K, seed=5).float()).to(BF)
    ref = x.float() @ W.float().T
    close(ops.gemv(W.to(dev), x.to(dev)), ref, 1e-2, 1e-2, "gemv", rel_to_std=True)
    close(ops.gemv(W.to(dev), x.to(dev), bias=bias.to(dev), res=res.to(dev)), ref + bias.float() + res.float(), 1.5e-2, 1e-2,
          "gemv+bias+res", rel_to_std=True)
    if B * K * 2 <= 65536:
        # the fused prologue rounds the normalised activations to bf16 exactly like the separate rmsnorm kernel;
        # each of the K products then carries a 2^-9 relative error -> ~2^-9 * std(out) * few sigma overall
        xn = rmsnorm(x.float(), nw.float(), 1e-5).to(BF).float()
        close(ops.gemv(W.to(dev), x.to(dev), norm_w=nw.to(dev), eps=1e-5), xn @ W.float().T, 2e-2, 1e-2, "gemv+norm",
              rel_to_std=True)


@pytest.mark.parametrize("B,I,K", [(1, 18944, 3584), (4, 300, 512), (8, 64, 4096)])
def test_gemv_swiglu(ops, dev, B, I, K):
    W, x = rnd(2 * I, K, seed=1, scale=0.05), rnd(B, K, seed=2)
    g, u = x.float() @ W[:I].float().T, x.float() @ W[I:].float().T
    close(ops.gemv_swiglu(W.to(dev), x.to(dev)), F.silu(g) * u, 1.5e-2, 2e-2, "gemv_swiglu", rel_to_std=True)


@pytest.mark.parametrize("B,V,K", [(1, 152064, 3584), (3, 1000, 256), (8, 97, 64)])
def test_lm_head_argmax(ops, dev, B, V, K):
    W, x = rnd(V, K, seed=1, scale=0.05), rnd(B, K, seed=2)
    logits = torch.empty(B, V, dtype=BF, device=dev)
    ids = ops.lm_head_argmax(W.to(dev), x.to(dev), logits=logits)
    ref = x.float() @ W.float().T
    close(logits, ref, 1e-2, 1e-2, "logits", rel_to_std=True)
    # ids must be the argmax of the kernel's own bf16 logits with lowest-index tie-break ...
    lg = logits.float().cpu()
    assert torch.equal(ids.cpu().long(), lg.argmax(-1))
    # ... and must agree with the fp32 reference wherever the reference margin is resolvable in bf16
    top2 = ref.topk(2, -1).values
    for b in range(B):
        if float(top2[b, 0] - top2[b, 1]) > 0.05:
            assert int(ids[b]) == int(ref[b].argmax())


def test_lm_head_argmax_ties(ops, dev):
    W = torch.zeros(300, 64, dtype=BF)
    W[[17, 123, 250], 0] = 1.0   # three identical rows -> identical logits -> lowest id wins
    x = torch.zeros(2, 64, dtype=BF); x[:, 0] = 1.0
    assert ops.lm_head_argmax(W.to(dev), x.to(dev)).cpu().tolist() == [17, 17]


@pytest.mark.parametrize("d,n_q,n_kv,B,S", [(128, 28, 4, 1, 1), (128, 8, 2, 2, 5), (64, 4, 4, 1, 3)])
def test_rope_kv_append(ops, dev, d, n_q, n_kv, B, S):
    from oracle.llama import LlamaCfg, apply_rope, rope_table
    cs = rope_table(LlamaCfg(head_dim=d, rope_theta=1e6), 64)
    qkv = rnd(B, S, (n_q + 2 * n_kv) * d, seed=1)
    pos = torch.randint(0, 50, (B, S), generator=torch.Generator().manual_seed(2), dtype=torch.int32)
    slot = torch.stack([torch.randperm(16, generator=torch.Generator().manual_seed(3 + b))[:S] for b in range(B)]).to(torch.int32)
    T = 16
    kc = torch.zeros(B, n_kv, T, d, dtype=BF, device=dev); vc = torch.zeros_like(kc)
    q = torch.empty(B, S, n_q, d, dtype=BF, device=dev)
    ops.rope_kv_append(qkv.to(dev), pos.to(dev).view(-1), slot.to(dev).view(-1), cs.to(dev), q, kc, vc, B, S, n_q, n_kv, d)
    x = qkv.float().view(B, S, n_q + 2 * n_kv, d)
    qr = apply_rope(x[:, :, :n_q].transpose(1, 2), cs, pos.long()).transpose(1, 2)
    kr = apply_rope(x[:, :, n_q:n_q + n_kv].transpose(1, 2), cs, pos.long())   # [B, n_kv, S, d]
    close(q, qr, 2e-2, 1e-2, "rope q")
    for b in range(B):
        for s in range(S):
            close(kc[b, :, int(slot[b, s])], kr[b, :, s], 2e-2, 1e-2, "rope k")
            assert torch.equal(vc[b, :, int(slot[b, s])].cpu(), qkv.view(B, S, -1, d)[b, s, n_q + n_kv:])


@pytest.mark.parametrize("B,n_q,n_kv,T,nsplit,beg", [(1, 28, 4, 1537, 64, 0), (2, 32, 8, 300, 8, 0), (3, 8, 8, 77, 1, 0),
                                                    (2, 4, 2, 130, 5, 17), (1, 7, 1, 5, 4, 0), (1, 28, 4, 4096, 128, 0)])
def test_attn_decode(ops, dev, B, n_q, n_kv, T, nsplit, beg):
    from oracle.llama import attention
    d, Tmax = 128, T + 3
    q, k, v = rnd(B, n_q, d, seed=1), rnd(B, n_kv, Tmax, d, seed=2), rnd(B, n_kv, Tmax, d, seed=3)
    kv_end = torch.tensor([T - (b % 2) * 3 for b in range(B)], dtype=torch.int32)
    kv_beg = torch.full((B,), beg, dtype=torch.int32)
    out = ops.attn_decode(q.to(dev), k.to(dev), v.to(dev), kv_end.to(dev), kv_beg=kv_beg.to(dev), nsplit=nsplit)
    for b in range(B):
        e = int(kv_end[b])
        ref = attention(q[b:b + 1, :, None].float(), k[b:b + 1, :, beg:e].float(), v[b:b + 1, :, beg:e].float(), None, 1 / math.sqrt(d))
        close(out[b].view(n_q, d), ref[0, :, 0], 1e-2, 1e-2, f"attn_decode b={b}")


# ------------------------------------------------------------------------------------------ GEMM / conv / attention
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1536, 4608, 3584), (77, 320, 768), (8192, 320, 320), (200, 36, 72),
                                   (1, 1280, 320), (130, 132, 1032),
                                   (4096, 320, 2560),      # long-K linear with >= 2048 rows: the LDS-DMA kernel, 4 K splits
                                   (2100, 312, 2568),      # same path with ragged M / N / K tails (N, K multiples of 8 / 4 only)
                                   (4608, 3840, 1280),     # 256^2 LDS-DMA kernel (>= 160 tiles): SDXL 24^2 qkv
                                   (1536, 3584, 3584),     # 256^2 kernel with 3 K splits (84 tiles): LLM prefill o-proj
                                   (4000, 2500, 1096),     # 256^2 kernel, ragged M / N / K tails
                                   # >= 16384 rows (UNet3D / SDXL maps): tile picked by the rounds x (t0 + nk tk) cost model
                                   (16384, 320, 320),      # K = 320 (5 K tiles): 128 x 160 LDS-DMA tiles
                                   (16500, 960, 320),      # 5 K tiles on the 256^2 kernel, ragged M
                                   (23040, 640, 640),      # 256^2, one round
                                   (16384, 320, 1280)])    # N = 320 on 256-wide tiles (62 % of the tile used)
def test_gemm(ops, dev, M, N, K):
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    ref = A.float() @ W.float().T
    tol = 2e-2 * max(1.0, math.sqrt(K) * 0.05)
    close(ops.gemm(A.to(dev), W.to(dev)), ref, tol, 1e-2, "gemm")
    close(ops.gemm(A.to(dev), W.to(dev), out_f32=True), ref, tol * 0.2, 1e-3, "gemm f32 out")
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    close(ops.gemm(A.to(dev), W.to(dev), bias=bias.to(dev), res=res.to(dev), out_scale=0.5),
          0.5 * (ref + bias.float() + res.float()), tol, 1.5e-2, "gemm+bias+res")
    close(ops.gemm(A.to(dev), W.to(dev), bias=bias.to(dev), act="silu"), F.silu(ref + bias.float()), tol, 1.5e-2, "gemm+silu")
    close(ops.gemm(A.to(dev), W.to(dev), act="gelu"), F.gelu(ref), tol, 1.5e-2, "gemm+gelu")
    if M % 4 == 0:
        rb = rnd(4, N, seed=5)
        close(ops.gemm(A.to(dev), W.to(dev), rowbias=rb.to(dev), rows_per_group=M // 4),
              ref + rb.float().repeat_interleave(M // 4, 0), tol, 1.5e-2, "gemm+rowbias")


@pytest.mark.parametrize("B,H,W,Cin,Cout,ks,stride,ups", [(2, 16, 16, 64, 128, 3, 1, False), (2, 64, 64, 320, 320, 3, 1, False),
                                                          (1, 16, 12, 128, 64, 3, 2, False), (2, 8, 8, 128, 128, 3, 1, True),
                                                          (2, 16, 16, 192, 64, 1, 1, False), (1, 9, 7, 64, 68, 3, 1, False),
                                                          (2, 16, 16, 640, 1280, 3, 1, False),    # 16^2 map: LDS-DMA kernel, 8 K splits
                                                          (2, 64, 64, 320, 320, 3, 2, False),     # stride 2 on the DMA path (M = 2048)
                                                          (8, 48, 48, 64, 640, 3, 1, False),      # 256^2 LDS-DMA kernel (216 tiles), halo taps
                                                          (8, 47, 49, 128, 632, 3, 1, False),     # the same, ragged rows / columns
                                                          (8, 24, 24, 64, 1280, 3, 1, True),      # the same through the fused 2x upsample
                                                          (8, 40, 72, 64, 640, 3, 1, False),      # 23040 rows: cost-model dispatch (UNet3D 20x36 x 32 frames)
                                                          (32, 40, 72, 64, 320, 3, 1, False)])    # 92160 rows (UNet3D 40x72 x 32 frames)
def test_conv2d(ops, dev, B, H, W, Cin, Cout, ks, stride, ups):
    x, w, bias = rnd(B, H, W, Cin, seed=1), rnd(Cout, ks, ks, Cin, seed=2, scale=0.05), rnd(Cout, seed=3)
    xin = x.float().permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w.float().permute(0, 3, 1, 2), bias.float(), stride=stride, padding=ks // 2).permute(0, 2, 3, 1)
    tol = 2e-2 * max(1.0, math.sqrt(ks * ks * Cin) * 0.05)
    y = ops.conv2d(x.to(dev), w.to(dev), bias=bias.to(dev), stride=stride, ups=ups)
    close(y, ref, tol, 1e-2, "conv2d")
    rb, res = rnd(B, Cout, seed=4), rnd(*ref.shape, seed=5)
    y = ops.conv2d(x.to(dev), w.to(dev), bias=bias.to(dev), rowbias=rb.to(dev), res=res.to(dev), stride=stride, ups=ups)
    close(y, ref + rb.float()[:, None, None, :] + res.float(), tol, 1.5e-2, "conv2d+temb+res")


def _attn_ref(q, k, v, heads, kv_heads, causal=False, kv_off=0, mask=None):
    B, Lq, C = q.shape
    d = C // heads
    qh = q.float().view(B, Lq, heads, d).transpose(1, 2)
    kh = k.float().view(B, -1, kv_heads, d).transpose(1, 2)
    vh = v.float().view(B, -1, kv_heads, d).transpose(1, 2)
    Lk = kh.shape[2]
    m = torch.zeros(Lq, Lk)
    if causal:
        i, j = torch.arange(Lq)[:, None], torch.arange(Lk)[None]
        m = m.masked_fill(j > i + kv_off, float("-inf"))
    if mask is not None:
        m = m.masked_fill(~mask, float("-inf"))
    from oracle.llama import attention
    return attention(qh, kh, vh, m[None, None], 1 / math.sqrt(d)).transpose(1, 2).reshape(B, Lq, C)


@pytest.mark.parametrize("B,heads,kvh,Lq,Lk,d,causal", [(2, 8, 8, 4096, 4096, 40, False), (2, 8, 8, 1024, 1024, 80, False),
                                                      (2, 8, 8, 256, 256, 160, False), (2, 8, 8, 64, 64, 160, False),
                                                      (2, 8, 8, 1024, 77, 80, False), (1, 28, 4, 300, 300, 128, True),
                                                      (2, 12, 12, 77, 77, 64, True), (1, 10, 10, 200, 333, 64, False),
                                                      (1, 4, 2, 130, 130, 128, True)])
def test_attention(ops, dev, B, heads, kvh, Lq, Lk, d, causal):
    q, k, v = rnd(B, Lq, heads * d, seed=1), rnd(B, Lk, kvh * d, seed=2), rnd(B, Lk, kvh * d, seed=3)
    out = ops.attention(q.to(dev), k.to(dev), v.to(dev), heads, kvh, causal=causal)
    close(out, _attn_ref(q, k, v, heads, kvh, causal, Lk - Lq), 1.5e-2, 1.5e-2, "attention")


@pytest.mark.parametrize("HW,F_,heads", [(37, 16, 5), (50, 5, 2), (9, 3, 10), (2880, 16, 5), (3, 1, 1)])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_attention_short_sequences(ops, dev, HW, F_, heads, dtype):
    """The one-wave-per-sequence kernel (Lq = Lk <= 16, head_dim 64) on the UNet3D's temporal layout: q / k / v are column slices of
    the fused projection of [frame, pixel] rows viewed as [pixel, frame] (batch stride = one row, row stride = HW rows, no copies),
    the output goes into the same kind of strided view. Against the fp32 reference and against the 128-row flash kernel."""
    import os
    dt = torch.bfloat16 if dtype == "bf16" else torch.float16
    inner = heads * 64
    qkv = rnd(F_ * HW, 3 * inner, seed=1).float().to(dt)
    qv = qkv.to(dev).view(F_, HW, 3 * inner).permute(1, 0, 2)          # [HW, F, 3*inner]
    o = torch.zeros(F_ * HW, inner, dtype=dt, device=dev)
    ov = o.view(F_, HW, inner).permute(1, 0, 2)
    ops.attention(qv[..., :inner], qv[..., inner:2 * inner], qv[..., 2 * inner:], heads, out=ov)
    c = qkv.float().view(F_, HW, 3 * inner).permute(1, 0, 2)
    ref = _attn_ref(c[..., :inner], c[..., inner:2 * inner], c[..., 2 * inner:], heads, heads)       # [HW, F, inner]
    tol = 1.5e-2 if dtype == "bf16" else 3e-3
    close(ov, ref, tol, tol, "short-sequence attention")
    assert bool((o.view(F_, HW, inner).permute(1, 0, 2).float().cpu() - ref).abs().max() < 10 * tol)


def test_attention_fused_qkv_strides(ops, dev):
    """q/k/v as column slices of one fused projection output (row stride 3C): no copies needed."""
    B, N, heads, d = 2, 256, 8, 40
    C = heads * d
    qkv = rnd(B, N, 3 * C, seed=1).to(dev)
    out = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads)
    c = qkv.cpu()
    close(out, _attn_ref(c[..., :C], c[..., C:2 * C], c[..., 2 * C:], heads, heads), 1.5e-2, 1.5e-2, "attention strided")


@pytest.mark.parametrize("N,heads,d,read_mode", [(64, 4, 64, False), (256, 10, 64, False), (64, 4, 64, True)])
def test_attention_consistent_mask(ops, dev, N, heads, d, read_mode):
    """Column keep vector + own-block rule == the dense cal_attn_mask_xl mask (gradio_utils.py:241-287)."""
    from oracle import story as ostory
    C = heads * d
    g = torch.Generator().manual_seed(9)
    u = torch.rand(5 * N, generator=g)
    keep = (u < 0.5).clone(); keep[4 * N:] = False
    dense = keep[None].repeat(5, 1)
    for i in range(5):
        dense[i, i * N:(i + 1) * N] = True
    dense = dense[:, None].repeat(1, N, 1).reshape(5 * N, 5 * N)
    if read_mode:
        Lq, Lk, mask, q_off = N, 5 * N, dense[4 * N:], 4 * N
    else:
        Lq, Lk, mask, q_off = 4 * N, 4 * N, dense[:4 * N, :4 * N], 0
    bits = torch.zeros((Lk + 63) // 64, dtype=torch.int64)
    for j in range(Lk):
        if keep[j]:
            bits[j // 64] |= (1 << (j % 64)) if (j % 64) < 63 else -(1 << 63)
    q, k, v = rnd(2, Lq, C, seed=1), rnd(2, Lk, C, seed=2), rnd(2, Lk, C, seed=3)
    out = ops.attention(q.to(dev), k.to(dev), v.to(dev), heads, keep_bits=bits.to(dev), blk=N, q_off=q_off)
    close(out, _attn_ref(q, k, v, heads, heads, mask=mask), 1.5e-2, 1.5e-2, "consistent attention")


@pytest.mark.parametrize("N,heads,d,read_mode", [(64, 4, 64, False), (256, 10, 64, False), (64, 4, 64, True), (200, 3, 64, False),
                                                  (576, 5, 64, True), (128, 4, 16, False), (64, 2, 40, True)])
def test_attention_consistent_keylists(ops, dev, N, heads, d, read_mode):
    """The visible-key-list form of the consistent self-attention (masked keys skipped, not scored) == the dense
    cal_attn_mask_xl mask (gradio_utils.py:241-287) == the keep-bits kernel; the lists themselves are checked exactly.
    N = 200 / 576: image blocks that are not multiples of the 128-row query tile (ragged last tile per image)."""
    C = heads * d
    g = torch.Generator().manual_seed(9)
    keep = (torch.rand(5 * N, generator=g) < 0.5).clone(); keep[4 * N:] = False
    dense = keep[None].repeat(5, 1)
    for i in range(5):
        dense[i, i * N:(i + 1) * N] = True
    if read_mode:
        Lq, Lk, q_off, img0, n_lists = N, 5 * N, 4 * N, 4, 1
    else:
        Lq, Lk, q_off, img0, n_lists = 4 * N, 4 * N, 0, 0, 4
    mask = dense[img0:img0 + n_lists, None, :Lk].repeat(1, N, 1).reshape(Lq, Lk)
    bits = ops.pack_keep_bits((~keep).float().to(dev), 0.5, 4 * N)       # u < thr <=> keep
    ki, tl = ops.story_key_lists(bits, Lk, N, img0, n_lists, img0)
    stride = (Lk + 63) // 64 * 64
    tpl = (N + 127) // 128
    for l in range(n_lists):
        want = torch.nonzero(dense[img0 + l, :Lk]).flatten().to(torch.int32)
        rec = tl[l * tpl:(l + 1) * tpl].cpu()
        assert int(rec[0, 3]) == want.numel() and int(rec[0, 2]) == l * stride
        assert torch.equal(ki[l * stride:l * stride + want.numel()].cpu(), want)
        assert rec[:, 0].tolist() == [l * N + 128 * t for t in range(tpl)] and int(rec[:, 1].sum()) == N
    q, k, v = rnd(2, Lq, C, seed=1), rnd(2, Lk, C, seed=2), rnd(2, Lk, C, seed=3)
    out = ops.attention_keylist(q.to(dev), k.to(dev), v.to(dev), heads, ki, tl)
    close(out, _attn_ref(q, k, v, heads, heads, mask=mask), 1.5e-2, 1.5e-2, "key-list attention")
    ref2 = ops.attention(q.to(dev), k.to(dev), v.to(dev), heads, keep_bits=bits, blk=N, q_off=q_off)
    close(out, ref2.float().cpu(), 1.5e-2, 1.5e-2, "key-list vs keep-bits kernel")


def test_attention_prefill_cache_leftpad(ops, dev):
    from oracle.llama import attention
    B, S, n_q, n_kv, d, T = 2, 70, 8, 2, 128, 96
    q = rnd(B, S, n_q, d, seed=1)
    kc, vc = rnd(B, n_kv, T, d, seed=2), rnd(B, n_kv, T, d, seed=3)
    beg = torch.tensor([0, 9], dtype=torch.int32)
    out = ops.attention_cache(q.to(dev), kc.to(dev), vc.to(dev), Lk=S, causal=True, kv_off=0, kv_beg=beg.to(dev))
    for b in range(B):
        i, j = torch.arange(S)[:, None], torch.arange(S)[None]
        m = torch.zeros(S, S).masked_fill((j > i) | (j < int(beg[b])), float("-inf"))
        ref = attention(q[b:b + 1].float().transpose(1, 2), kc[b:b + 1, :, :S].float(), vc[b:b + 1, :, :S].float(),
                        m[None, None], 1 / math.sqrt(d)).transpose(1, 2).reshape(S, n_q * d)
        close(out[b, int(beg[b]):], ref[int(beg[b]):], 1.5e-2, 1.5e-2, "prefill attention")


# ------------------------------------------------------------------------------------------ UNet elementwise
@pytest.mark.parametrize("B,HW,C", [(2, 4096, 320), (2, 1024, 640), (2, 64, 1280), (1, 256, 1920), (2, 100, 2560), (1, 16384, 128)])
@pytest.mark.parametrize("silu", [False, True])
def test_groupnorm(ops, dev, B, HW, C, silu):
    x = (rnd(B, HW, C, seed=1).float() * 2 + 0.5).to(BF)
    ga, be = (1 + 0.2 * rnd(C, seed=2).float()).to(BF), rnd(C, seed=3, scale=0.2)
    ref = F.group_norm(x.float().transpose(1, 2), 32, ga.float(), be.float(), 1e-5).transpose(1, 2)
    if silu:
        ref = F.silu(ref)
    close(ops.groupnorm(x.to(dev), ga.to(dev), be.to(dev), 32, 1e-5, silu), ref, 2e-2, 1.5e-2, "groupnorm")


@pytest.mark.parametrize("B,HW,C1,C2,silu", [(2, 4096, 320, 320, True), (2, 4096, 640, 320, True), (2, 1024, 1280, 640, False),
                                             (2, 256, 1280, 1280, True), (2, 64, 1280, 1280, True), (1, 1000, 256, 128, True)])
def test_groupnorm_of_skip_concat(ops, dev, B, HW, C1, C2, silu):
    """norm1 of an up block: GroupNorm over cat([hidden, skip]) read in place == the kernel on the materialised concat (bit for
    bit: same arithmetic, same order), the concatenated copy it hands to the shortcut is exact, and both match torch fp32.
    (640 + 320 / 1280 + 640: a group straddles the seam between the two tensors.)"""
    x1 = (rnd(B, HW, C1, seed=1).float() * 2 + 0.5).to(BF).to(dev)
    x2 = (rnd(B, HW, C2, seed=4).float() * 0.7 - 0.3).to(BF).to(dev)
    C = C1 + C2
    ga, be = (1 + 0.2 * rnd(C, seed=2).float()).to(BF).to(dev), rnd(C, seed=3, scale=0.2).to(dev)
    y, cat = ops.groupnorm_cat(x1, x2, ga, be, 32, 1e-5, silu)
    xc = torch.cat([x1, x2], -1)
    assert torch.equal(cat, xc)
    assert torch.equal(y, ops.groupnorm(xc, ga, be, 32, 1e-5, silu))
    ref = F.group_norm(xc.float().cpu().transpose(1, 2), 32, ga.float().cpu(), be.float().cpu(), 1e-5).transpose(1, 2)
    close(y, F.silu(ref) if silu else ref, 2e-2, 1.5e-2, "groupnorm_cat")


@pytest.mark.parametrize("rows,C", [(8192, 320), (77, 768), (3, 1280), (10, 2048)])
def test_layernorm(ops, dev, rows, C):
    x, ga, be = rnd(rows, C, seed=1), (1 + 0.2 * rnd(C, seed=2).float()).to(BF), rnd(C, seed=3, scale=0.2)
    close(ops.layernorm(x.to(dev), ga.to(dev), be.to(dev), 1e-5), F.layer_norm(x.float(), (C,), ga.float(), be.float(), 1e-5),
          2e-2, 1e-2, "layernorm")


def test_geglu_swiglu_concat_act_add(ops, dev):
    x = rnd(100, 2 * 1280, seed=1)
    a, g = x.float().chunk(2, -1)
    close(ops.geglu(x.to(dev)), a * F.gelu(g), 1e-2, 1.5e-2, "geglu")
    close(ops.swiglu(x.to(dev)), F.silu(a) * g, 1e-2, 1.5e-2, "swiglu")
    p, q = rnd(2, 30, 64, seed=2), rnd(2, 30, 128, seed=3)
    assert torch.equal(ops.concat_channels(p.to(dev), q.to(dev)).cpu(), torch.cat([p, q], -1))
    close(ops.act(p.to(dev), "silu"), F.silu(p.float()), 1e-2, 1e-2, "silu")
    close(ops.act(p.to(dev), "quick_gelu"), p.float() * torch.sigmoid(1.702 * p.float()), 1e-2, 1e-2, "quick_gelu")
    close(ops.add(p.to(dev), p.to(dev)), 2 * p.float(), 1e-6, 1e-2, "add")


def test_small_convs(ops, dev):
    x, w, b = rnd(2, 64, 64, 4, seed=1), rnd(320, 3, 3, 4, seed=2, scale=0.2), rnd(320, seed=3)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=1).permute(0, 2, 3, 1)
    close(ops.conv2d_small_cin(x.to(dev), w.to(dev), b.to(dev)), ref, 2e-2, 1e-2, "conv_in")
    x, w, b = rnd(2, 64, 64, 320, seed=4), rnd(4, 3, 3, 320, seed=5, scale=0.05), rnd(4, seed=6)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=1).permute(0, 2, 3, 1)
    close(ops.conv2d_small_cout(x.to(dev), w.to(dev), b.to(dev)), ref, 5e-3, 1e-3, "conv_out f32")
    x, w = rnd(1, 32, 32, 128, seed=7), rnd(3, 3, 3, 128, seed=8, scale=0.05)  # VAE conv_out: Cout = 3
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, padding=1).permute(0, 2, 3, 1)
    close(ops.conv2d_small_cout(x.to(dev), w.to(dev), None), ref, 5e-3, 1e-3, "conv_out cout=3")


def test_latent_plumbing(ops, dev):
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(1, 4, 64, 64, generator=g)
    nh = ops.latent_to_nhwc(lat.to(dev), reps=2, scale=0.5)
    ref = (lat * 0.5).permute(0, 2, 3, 1).to(BF)
    assert torch.equal(nh.cpu(), torch.cat([ref, ref], 0))
    e = torch.randn(2, 64, 64, 4, generator=g)
    cf = ops.cfg_combine(e.to(dev), 7.5)
    close(cf, (e[0] + 7.5 * (e[1] - e[0])).permute(2, 0, 1)[None], 1e-5, 1e-5, "cfg")
    ts = [torch.randn(1, 4, 64, 64, generator=g) for _ in range(5)]
    cs = [0.9, -0.3, 0.2, 0.05, -1.1]
    close(ops.lincomb([t.to(dev) for t in ts], cs), sum(c * t for c, t in zip(cs, ts)), 1e-5, 1e-5, "lincomb")
    img = torch.randn(1, 16, 16, 3, generator=g)
    close(ops.nhwc_to_nchw(img.to(dev), 0.5, 0.5, True), (img * 0.5 + 0.5).clamp(0, 1).permute(0, 3, 1, 2), 1e-6, 1e-6, "post")


@pytest.mark.parametrize("B,n_q,n_kv,T,nsplit,beg", [(1, 28, 4, 1537, 32, 0), (2, 32, 8, 300, 8, 0), (3, 8, 8, 1, 4, 0),
                                                    (2, 4, 2, 130, 5, 17), (1, 7, 1, 6, 1, 0), (1, 28, 4, 2000, 32, 0)])
@pytest.mark.parametrize("inline", ["0", "1", "2"])
def test_attn_decode_fused(ops, dev, B, n_q, n_kv, T, nsplit, beg, inline, monkeypatch):
    """One-launch RoPE + KV append + split-KV attention + cross-block combine == the three separate kernels
    (bit-identical cache rows, outputs equal up to fp32 summation order), repeated to exercise the self-resetting
    ticket counters the way hipGraph replays do. inline = 1 / 2: the combine by the last-arriving block of the same launch
    (write-through partial stores + ticket; 1: acq_rel ticket, 2: relaxed ticket + sc1 loads, no fence); 0: by attn_combine_kernel."""
    from oracle.llama import LlamaCfg, rope_table
    from spider_amd import lib as slib
    prev = slib.load().spider_set_attn_inline(int(inline))
    d, Tmax = 128, T + 5
    cs = rope_table(LlamaCfg(head_dim=d, rope_theta=1e6), Tmax + 8).to(dev)
    kc0, vc0 = rnd(B, n_kv, Tmax, d, seed=2).to(dev), rnd(B, n_kv, Tmax, d, seed=3).to(dev)
    kv_end = torch.tensor([T - (b % 2) * (1 if T > 3 else 0) for b in range(B)], dtype=torch.int32, device=dev)
    kv_beg = torch.full((B,), beg, dtype=torch.int32, device=dev)
    pos = (kv_end - 1 - kv_beg).to(torch.int32)
    slot = (kv_end - 1).to(torch.int32)
    cnt = torch.zeros(B * n_kv, dtype=torch.int32, device=dev)
    ws = (torch.empty(B * n_q * nsplit * d, dtype=torch.float32, device=dev), torch.empty(B * n_q * nsplit * 2, dtype=torch.float32, device=dev))
    for rep in range(3):
        qkv = rnd(B, (n_q + 2 * n_kv) * d, seed=10 + rep).to(dev)
        # reference path: separate kernels on a copy of the caches
        k1, v1 = kc0.clone(), vc0.clone()
        q = torch.empty(B, 1, n_q, d, dtype=BF, device=dev)
        ops.rope_kv_append(qkv, pos, slot, cs, q, k1, v1, B, 1, n_q, n_kv, d)
        ref = ops.attn_decode(q.view(B, n_q, d), k1, v1, kv_end, kv_beg=kv_beg, nsplit=nsplit)
        k2, v2 = kc0.clone(), vc0.clone()
        out = torch.empty(B, n_q * d, dtype=BF, device=dev)
        ops.attn_decode_fused(qkv, pos, cs, k2, v2, kv_end, kv_beg, cnt, n_q, nsplit, ws, out)
        assert torch.equal(k1, k2) and torch.equal(v1, v2), "KV append must be bit-identical"
        close(out, ref.float(), 4e-3, 1e-2, f"fused decode attention rep {rep}")
        assert int(cnt.abs().sum()) == 0, "ticket counters must be back to zero after every launch"
        out2 = torch.empty_like(out)
        ops.attn_decode_fused(qkv, pos, cs, kc0.clone(), vc0.clone(), kv_end, kv_beg, cnt, n_q, nsplit, ws, out2)
        assert torch.equal(out, out2), "the merge order is fixed: repeats are bit-identical whichever block arrives last"
    slib.load().spider_set_attn_inline(0 if prev < 0 else prev)


def _busy_stream(dev, ms: float = 60.0):
    """a side stream that keeps HBM and every CU's memory queue busy for roughly `ms`: hand-offs must hold under load, not on an idle chip"""
    side = torch.cuda.Stream()
    big = torch.zeros(256 << 20, dtype=torch.float32, device=dev)          # 1 GiB: every pass goes to HBM
    with torch.cuda.stream(side):
        for _ in range(int(ms / 0.45) + 1):
            big.add_(1.0)
    return side, big


@pytest.mark.parametrize("nsplit,n_q,n_kv,T", [(64, 28, 4, 1600), (32, 28, 4, 1600), (16, 32, 8, 900)])
def test_attn_decode_fused_fence_free_combine_under_load(ops, dev, nsplit, n_q, n_kv, T):
    """Form 2 of the in-launch split-KV combine has no release / acquire fence (write-through partials, drained, a relaxed ticket, sc1
    loads in the reducer). Its failure mode is silent: stale partials of the PREVIOUS launch summed into this one. So: 8 different
    query rows, each answered once on a quiet chip, then the 8 launched back to back 40 times (the workspace lines of launch i are L2 /
    L1 resident from launch i - 1) while a second stream streams 1 GiB passes; every output word must equal the quiet answer and the
    ticket counters must be zero at the end."""
    from oracle.llama import LlamaCfg, rope_table
    from spider_amd import lib as slib
    prev = slib.load().spider_set_attn_inline(2)
    try:
        B, d, Tmax = 1, 128, T + 8
        cs = rope_table(LlamaCfg(head_dim=d, rope_theta=1e6), Tmax + 8).to(dev)
        kc, vc = rnd(B, n_kv, Tmax, d, seed=21).to(dev), rnd(B, n_kv, Tmax, d, seed=22).to(dev)
        kv_end = torch.tensor([T], dtype=torch.int32, device=dev)
        kv_beg = torch.zeros(B, dtype=torch.int32, device=dev)
        pos = (kv_end - 1).to(torch.int32)
        cnt = torch.zeros(B * n_kv, dtype=torch.int32, device=dev)
        ws = (torch.empty(B * n_q * nsplit * d, dtype=torch.float32, device=dev), torch.empty(B * n_q * nsplit * 2, dtype=torch.float32, device=dev))
        qkvs = [rnd(B, (n_q + 2 * n_kv) * d, seed=40 + i).to(dev) for i in range(8)]
        quiet = []
        for qkv in qkvs:
            o = torch.empty(B, n_q * d, dtype=BF, device=dev)
            ops.attn_decode_fused(qkv, pos, cs, kc, vc, kv_end, kv_beg, cnt, n_q, nsplit, ws, o)
            torch.cuda.synchronize()
            quiet.append(o)
        assert not torch.equal(quiet[0], quiet[1])
        side, big = _busy_stream(dev)
        outs = [[torch.empty(B, n_q * d, dtype=BF, device=dev) for _ in qkvs] for _ in range(40)]
        for rep in range(40):
            for i, qkv in enumerate(qkvs):
                ops.attn_decode_fused(qkv, pos, cs, kc, vc, kv_end, kv_beg, cnt, n_q, nsplit, ws, outs[rep][i])
        torch.cuda.synchronize()
        bad = sum(int(not torch.equal(outs[rep][i], quiet[i])) for rep in range(40) for i in range(8))
        assert bad == 0, f"{bad} of 320 launches under load differ from the quiet-chip answer"
        assert int(cnt.abs().sum()) == 0
    finally:
        slib.load().spider_set_attn_inline(0 if prev < 0 else prev)


def test_wstream_inlaunch_combine_stress_under_load(ops, dev, monkeypatch):
    """advisor (round 5): the in-launch split-K combine of the streaming conv (sc1 slab stores, drained, ONE relaxed counter add, sc1
    slab loads; no fence) had one eager run + three replays of one shape as its evidence. Here: four shapes (128- and 512-pixel forms,
    2560 -> 1280 included), each captured in a hipGraph of 8 launches over 4 different inputs, replayed 250 times (2000 launches per
    shape) while a second stream streams 1 GiB passes; every replay's outputs must be bit-identical to the reduce-kernel form
    (ops.set_ws_inlaunch(False): same slabs, same order) and the arrival counters must be zero afterwards."""
    monkeypatch.setattr(ops, "WS_MAX_M", 512)
    shapes = [(2, 8, 8, 1280, 1280), (2, 8, 8, 2560, 1280), (2, 16, 16, 640, 640), (1, 8, 8, 1920, 1280)]
    for (B, H, W, Cin, Cout) in shapes:
        g = torch.Generator(device=dev).manual_seed(Cin + H)
        xs = [torch.randn(B, H, W, Cin, generator=g, device=dev).half() for _ in range(4)]
        w = ops.mark_weight((torch.randn(Cout, 3, 3, Cin, generator=g, device=dev) * (1.0 / (3 * Cin ** 0.5))).half())
        bias = (torch.randn(Cout, generator=g, device=dev) * 0.1).half()
        prev = ops.set_ws_inlaunch(False)
        try:
            want = [ops.conv_ex(x, w, bias=bias, pad=(1, 1)) for x in xs]
            ops.set_ws_inlaunch(True)
            got0 = [ops.conv_ex(x, w, bias=bias, pad=(1, 1)) for x in xs]           # eager, also the graph's warm-up
            torch.cuda.synchronize()
            assert getattr(w, "_spider_fm", None) is not None, "the streaming kernel's weight copy was not built: wrong path"
            for a, b_ in zip(got0, want):
                assert torch.equal(a, b_), "in-launch combine and reduce kernel sum the same slabs in the same order"
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                outs = [ops.conv_ex(xs[i % 4], w, bias=bias, pad=(1, 1)) for i in range(8)]
            side, big = _busy_stream(dev, 120.0)
            bad = 0
            for rep in range(250):
                gr.replay()
                if rep % 25 == 24:           # (comparisons are enqueued behind the replay on the same stream)
                    bad += sum(int(not torch.equal(outs[i], want[i % 4])) for i in range(8))
            torch.cuda.synchronize()
            bad += sum(int(not torch.equal(outs[i], want[i % 4])) for i in range(8))
            assert bad == 0, f"shape {(B, H, W, Cin, Cout)}: {bad} outputs differ from the reduce-kernel form"
            cntv = ops._workspace(xs[0].device)[ops.WS_BYTES // 4:]
            assert int(cntv.view(torch.int32).abs().sum()) == 0, "arrival counters must be zero between calls"
        finally:
            ops.set_ws_inlaunch(prev)


def test_gemm_256_tile_repeatable_under_load(ops, dev):
    """Race screen of the 256 x 256 LDS-DMA kernel (counted vmcnt, raw barriers, wave groups half a phase apart): repeated launches
    must be bit-identical while another stream keeps HBM busy (a DMA tile read before it landed would show up as a differing
    repeat). Longer form: scripts/exp/p8_race_screen.py."""
    side = torch.cuda.Stream()
    junk = torch.empty(128 << 20, dtype=torch.uint8, device=dev)
    A, W = rnd(4608, 1280, seed=1).to(dev), rnd(3840, 1280, seed=2, scale=0.05).to(dev)
    A2, W2 = rnd(1536, 3584, seed=3).to(dev), rnd(3584, 3584, seed=4, scale=0.05).to(dev)     # 256 x 128 tiles (168 of them)
    W3 = rnd(10240, 1280, seed=7, scale=0.05).to(dev)                                         # 256 x 256 tiles (720)
    A4, W4 = rnd(1536, 18944, seed=8).to(dev), rnd(3584, 18944, seed=9, scale=0.02).to(dev)   # 256 x 256 tiles, 3 K splits
    x, w = rnd(8, 48, 48, 128, seed=5).to(dev), rnd(640, 3, 3, 128, seed=6, scale=0.03).to(dev)
    for f in (lambda: ops.gemm(A, W), lambda: ops.gemm(A2, W2), lambda: ops.gemm(A, W3), lambda: ops.gemm(A4, W4),
              lambda: ops.gemm(A, W, act="geglu"), lambda: ops.conv2d(x, w)):
        ref = f().clone()
        for i in range(40):
            if i % 2 == 0:
                with torch.cuda.stream(side):
                    junk.add_(1)
            assert torch.equal(f(), ref), f"repeat {i} differs"
    torch.cuda.synchronize()


def test_lds_dma_kernels_repeatable_under_load(ops, dev):
    """Race screen of the LDS-DMA GEMM / conv family (gemm_dma_kernel: `buffer_load ... lds` rings of 2 / 3 / 4 stages with counted
    vmcnt waits and ONE raw barrier per K tile; 64- and 128-row tiles; split-K slabs + the statistics-producing reduce; the GroupNorm
    statistics epilogue; the two-blocks-per-CU form with the row-order epilogue through LDS) and of the fused cross-attention
    sub-block: the shapes the SD-v1.5 / zeroscope engines send there. A DMA tile consumed before it landed, or an epilogue staging
    buffer reused too early, shows up as a repeat that differs while a second stream shifts the timing of the waves."""
    import math
    side = torch.cuda.Stream()
    junk = torch.empty(128 << 20, dtype=torch.uint8, device=dev)
    H = torch.float16
    def t(*shape, seed, scale=1.0):
        return rnd(*shape, seed=seed, scale=scale).to(H).to(dev)
    x64, w64, b64 = t(2, 64, 64, 320, seed=1), t(320, 3, 3, 320, seed=2, scale=0.02), t(320, seed=3)
    x64b, w64b = t(2, 64, 64, 640, seed=4), t(320, 3, 3, 640, seed=5, scale=0.015)
    x32, w32 = t(2, 32, 32, 1280, seed=6), t(640, 3, 3, 1280, seed=7, scale=0.01)
    x16, w16 = t(2, 16, 16, 1280, seed=8), t(1280, 3, 3, 1280, seed=9, scale=0.01)
    r64 = torch.randn(2, 64, 64, 320, generator=torch.Generator().manual_seed(10)).to(dev)
    At, Wt, rt = t(92160, 320, seed=11), t(320, 320, seed=12, scale=0.05), torch.randn(92160, 320, generator=torch.Generator().manual_seed(13)).to(dev)
    xt, wt3 = t(32, 40, 72, 320, seed=14), t(320, 3, 3, 320, seed=15, scale=0.02)          # zeroscope: 92160 rows, 3 x 3
    A5, W5 = t(512, 5120, seed=16), t(1280, 5120, seed=17, scale=0.02)                      # 16^2 ff2: long-K linear on the DMA kernel, K splits
    # fused cross-attention at the 64^2 site
    B2, n_tok, C, LP = 2, 4096, 320, 80
    xx = t(B2, n_tok, C, seed=18)
    mq, mo = t(B2 * 8 * LP, C, seed=19, scale=0.05), t(B2 * C, 8 * LP, seed=20, scale=0.05)
    mqf, mof = ops.repack_fm16(mq), ops.repack_fm16(mo)
    c1, c2, bo = torch.zeros(B2 * 8 * LP, device=dev), torch.zeros(B2 * 8 * LP, device=dev), t(C, seed=21)
    x32s = torch.randn(B2, n_tok, C, generator=torch.Generator().manual_seed(22)).to(dev)

    def flat(o):
        o = o if isinstance(o, (tuple, list)) else (o,)
        return [v.t if hasattr(v, "t") and not torch.is_tensor(v) else v for v in o if v is not None]
    fs = [lambda: ops.conv_ex(x64, w64, bias=b64, pad=(1, 1), gn_groups=32),                         # 64-row tiles, statistics epilogue
          lambda: ops.conv_ex(x64, w64, bias=b64, pad=(1, 1), res32=r64, want32=True),               # conv2 role: fp32 residual stream
          lambda: ops.conv_ex(x64b, w64b, pad=(1, 1), gn_groups=32),                                 # 128-row tiles, K splits + reduce
          lambda: ops.conv_ex(x32, w32, pad=(1, 1), gn_groups=32),
          lambda: ops.conv_ex(x16, w16, pad=(1, 1), gn_groups=32),                                   # 8 K splits, statistics from the reduce
          lambda: ops.gemm(At, Wt, res32=rt, want32=True),                                           # two blocks per CU + row-order epilogue
          lambda: ops.conv_ex(xt, wt3, pad=(1, 1)),                                                  # tall conv on the 2-stage ring
          lambda: ops.gemm(A5, W5),
          lambda: ops.xattn_fused(xx, mqf, mof, c1, c2, bo, B2, 8, 77, x32=x32s, want32=True)]
    for k, f in enumerate(fs):
        ref = [v.clone() for v in flat(f())]
        for i in range(25):
            if i % 2 == 0:
                with torch.cuda.stream(side):
                    junk.add_(1)
            got = flat(f())
            assert len(got) == len(ref) and all(torch.equal(a, b_) for a, b_ in zip(got, ref)), f"case {k}: repeat {i} differs"
    torch.cuda.synchronize()


def test_attention_kernels_repeatable_under_load(ops, dev):
    """Race screen of the attention kernels that overlap their own loads with compute: the software-pipelined dense flash kernel
    (K / V images re-staged while the previous tile is still being scored: the round-2 advisor found its prologue one barrier
    short), the one-wave-per-sequence kernel (wave-private LDS tile, no barrier) and the split-KV decode attention. Repeated
    launches must be bit-identical while another stream keeps HBM busy and shifts the timing of the waves."""
    side = torch.cuda.Stream()
    junk = torch.empty(128 << 20, dtype=torch.uint8, device=dev)
    q40 = rnd(2, 4096, 3 * 320, seed=1).to(dev)                       # SD-v1.5 64^2 self-attention: d = 40, pipelined kernel, KQ = 3
    q80 = rnd(2, 1024, 3 * 640, seed=2).to(dev)                       # 32^2: d = 80 in 96-wide tiles
    q64 = rnd(8, 2304, 3 * 640, seed=3).to(torch.float16).to(dev)     # SDXL 48^2: d = 64, f16 instantiation
    t16 = rnd(16 * 700, 3 * 320, seed=4).to(dev).view(16, 700, 960).permute(1, 0, 2)      # temporal layout: 700 pixels x 16 frames
    fs = [lambda: ops.attention(q40[..., :320], q40[..., 320:640], q40[..., 640:], 8),
          lambda: ops.attention(q80[..., :640], q80[..., 640:1280], q80[..., 1280:], 8),
          lambda: ops.attention(q64[..., :640], q64[..., 640:1280], q64[..., 1280:], 10),
          lambda: ops.attention(t16[..., :320], t16[..., 320:640], t16[..., 640:], 5)]
    for f in fs:
        ref = f().clone()
        for i in range(30):
            if i % 2 == 0:
                with torch.cuda.stream(side):
                    junk.add_(1)
            assert torch.equal(f(), ref), f"repeat {i} differs"
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,inner,K", [(8192, 1280, 320), (2048, 2560, 640), (512, 5120, 1280), (100, 72, 64), (128, 5120, 1280),
                                       (4608, 5120, 1280),    # 256^2 LDS-DMA kernel with the GEGLU epilogue (SDXL 24^2 ff1)
                                       (4000, 2504, 1096),    # the same, ragged M / N / K
                                       (16400, 1280, 320)])   # >= 16384 rows: cost-model dispatch, 5 K tiles on the 256^2 kernel (UNet3D ff1)
def test_gemm_fused_geglu(ops, dev, M, inner, K):
    A, W, b = rnd(M, K, seed=1), rnd(2 * inner, K, seed=2, scale=0.05), rnd(2 * inner, seed=3, scale=0.2)
    p = A.float() @ W.float().T + b.float()
    ref = p[:, :inner] * F.gelu(p[:, inner:])
    got = ops.gemm(A.to(dev), W.to(dev), bias=b.to(dev), act="geglu")
    assert got.shape == (M, inner)
    close(got, ref, 1.5e-2, 2e-2, "fused geglu", rel_to_std=True)
    close(got, ops.geglu(ops.gemm(A.to(dev), W.to(dev), bias=b.to(dev))).float(), 1e-6, 1e-2, "fused == unfused geglu")


@pytest.mark.parametrize("B,N,K", [(5, 4608, 3584), (8, 3584, 18944), (16, 130, 1024), (6, 7, 64), (8, 18944, 3584)])
def test_skinny_mfma_gemm(ops, dev, B, N, K):
    """Batched-decode GEMV on MFMA (2..16 rows): plain + bias + residual, and the fused gate/up SwiGLU form."""
    W, x = rnd(N, K, seed=1, scale=0.05), rnd(B, K, seed=2)
    bias, res = rnd(N, seed=3), rnd(B, N, seed=4)
    ref = x.float() @ W.float().T
    close(ops.gemv(W.to(dev), x.to(dev)), ref, 1e-2, 1e-2, "skinny gemm", rel_to_std=True)
    close(ops.gemv(W.to(dev), x.to(dev), bias=bias.to(dev), res=res.to(dev)), ref + bias.float() + res.float(), 1.5e-2, 1e-2,
          "skinny gemm + bias + res", rel_to_std=True)
    if N % 2 == 0:
        I = N // 2
        gg, uu = ref[:, :I], ref[:, I:]
        close(ops.gemv_swiglu(W.to(dev), x.to(dev)), F.silu(gg) * uu, 1.5e-2, 2e-2, "skinny swiglu", rel_to_std=True)


# ------------------------------------------------------------------------------------------ general conv / vocoder kernels
@pytest.mark.parametrize("B,H,W,Cin,Cout,kh,kw,stride,pad,dil,up", [
    (2, 1, 300, 64, 128, 1, 7, 1, (0, 3), 1, None),      # Conv1d k=7 (HiFi-GAN conv_pre)
    (1, 1, 257, 32, 32, 1, 11, 1, (0, 25), 5, None),     # Cin=32: per-lane taps; dilation 5
    (1, 1, 100, 8, 16, 1, 3, 1, (0, 3), 3, None),        # Cin=8
    (2, 6, 35, 64, 64, 3, 1, 1, (1, 0), 1, None),        # (3,1) temporal kernel over "frames"
    (2, 8, 5, 128, 64, 3, 3, 1, (1, 1), 1, (15, 9)),     # upsample to 2n-1 (odd skip size), then 3x3
    (1, 7, 9, 64, 64, 3, 3, 2, (1, 1), 1, None),         # stride 2 on odd sizes
    (2, 16, 720, 128, 640, 3, 1, 1, (1, 0), 1, None),    # UNet3D temporal conv at the 20 x 36 level: 23040 rows (cost-model dispatch)
])
def test_conv_ex(ops, dev, B, H, W, Cin, Cout, kh, kw, stride, pad, dil, up):
    x, w, b = rnd(B, H, W, Cin, seed=1), rnd(Cout, kh, kw, Cin, seed=2, scale=(kh * kw * Cin) ** -0.5), rnd(Cout, seed=3)
    y = ops.conv_ex(x.to(dev), w.to(dev), bias=b.to(dev), stride=stride, pad=pad, dil=dil, up_size=up)
    xi = x.float().permute(0, 3, 1, 2)
    if up is not None:
        xi = F.interpolate(xi, size=up, mode="nearest")
    ref = F.conv2d(xi, w.float().permute(0, 3, 1, 2), b.float(), stride=stride, padding=pad, dilation=dil).permute(0, 2, 3, 1)
    close(y, ref, 2e-2, 1e-2, "conv_ex", rel_to_std=True)


def test_conv_ex_fused_act_and_res(ops, dev):
    x, w, b, r = rnd(1, 1, 90, 32, seed=1), rnd(32, 1, 3, 32, seed=2, scale=0.1), rnd(32, seed=3), rnd(1, 1, 90, 32, seed=4)
    y = ops.conv_ex(x.to(dev), w.to(dev), bias=b.to(dev), pad=(0, 1), act="leaky_relu", act_param=0.1)
    ref = F.leaky_relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=(0, 1)), 0.1).permute(0, 2, 3, 1)
    close(y, ref, 2e-2, 1e-2, "conv+lrelu", rel_to_std=True)
    y = ops.conv_ex(x.to(dev), w.to(dev), bias=b.to(dev), pad=(0, 1), res=r.to(dev), act="tanh")
    ref = torch.tanh(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=(0, 1))).permute(0, 2, 3, 1) + r.float()
    close(y, ref, 2e-2, 1e-2, "conv+tanh+res", rel_to_std=True)


@pytest.mark.parametrize("L,Cin,Cout,k,stride", [(50, 64, 32, 16, 5), (33, 32, 16, 16, 4), (40, 16, 8, 8, 2), (20, 8, 4, 4, 2)])
def test_conv_transpose1d(ops, dev, L, Cin, Cout, k, stride):
    B, pad = 2, (k - stride) // 2
    x, w, b = rnd(B, L, Cin, seed=1), rnd(Cin, Cout, k, seed=2, scale=Cin ** -0.5), rnd(Cout, seed=3)
    taps = w.permute(2, 1, 0).reshape(k * Cout, Cin).contiguous()
    y = ops.conv_transpose1d(x.to(dev), taps.to(dev), b.to(dev), k, stride, pad)
    ref = F.conv_transpose1d(x.float().transpose(1, 2), w.float(), b.float(), stride=stride, padding=pad).transpose(1, 2)
    close(y, ref, 2e-2, 1e-2, "conv_transpose1d", rel_to_std=True)


def test_act_kinds_add_scaled_l2norm(ops, dev):
    x, y = rnd(3, 264, seed=1, scale=2.0), rnd(3, 264, seed=2)
    xf = x.float()
    for kind, fn in (("leaky_relu", lambda t: F.leaky_relu(t, 0.1)), ("relu", F.relu), ("tanh", torch.tanh), ("silu", F.silu)):
        close(ops.act(x.to(dev), kind, 0.1), fn(xf), 1e-2, 1e-2, kind)
    close(ops.add_scaled(x.to(dev), y.to(dev), 1 / 3), (xf + y.float()) / 3, 1e-2, 1e-2, "add_scaled")
    close(ops.l2_normalize(x.to(dev)), F.normalize(xf, dim=-1), 2e-3, 1e-2, "l2_normalize")
    z = torch.zeros(2, 16, dtype=BF)
    assert torch.equal(ops.l2_normalize(z.to(dev)).cpu(), z), "zero rows stay zero (eps clamp)"


def test_conv_small_cout_8(ops, dev):
    x, w, b = rnd(2, 9, 5, 64, seed=1), rnd(8, 3, 3, 64, seed=2, scale=0.05), rnd(8, seed=3)
    y = ops.conv2d_small_cout(x.to(dev), w.to(dev), b.to(dev))
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=1).permute(0, 2, 3, 1)
    close(y, ref, 2e-2, 1e-2, "conv_small_cout(8)", rel_to_std=True)


@pytest.mark.parametrize("spike_tile", [0, 3, 17])
def test_attention_late_max_jump(ops, dev, spike_tile):
    """Online-softmax rescale exercised on purpose: the running max of one row jumps far above everything seen before at a chosen
    KV tile (one key row aligned with one query row), and another row gets a modest bump
    elsewhere. Full-tensor fp64 reference (random data alone rarely moves a max after the first tiles)."""
    B, N, heads, d = 1, 64 * 20, 2, 40
    g = torch.Generator().manual_seed(9)
    q = torch.randn(B, N, heads * d, generator=g) * 0.5
    k = torch.randn(B, N, heads * d, generator=g) * 0.5
    v = torch.randn(B, N, heads * d, generator=g)
    row = 5
    k[0, spike_tile * 64 + 11, :d] = q[0, row, :d] * 40.0          # head 0, query 5: score ~ 40 * |q|^2 / sqrt(d) >> threshold
    k[0, 64 * 9 + 3, d:] = q[0, row + 1, d:] * 1.5                 # head 1, query 6: a modest bump, below the threshold
    q, k, v = q.to(BF), k.to(BF), v.to(BF)
    out = ops.attention(q.to(dev), k.to(dev), v.to(dev), heads).float().cpu()
    qd, kd, vd = [t.double().view(B, N, heads, d).transpose(1, 2) for t in (q, k, v)]
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) / math.sqrt(d), -1) @ vd).transpose(1, 2).reshape(B, N, heads * d).float()
    close(out, ref, 2e-2, 1e-2, "attention late max jump")
    assert float((out[0, row, :d] - v[0, spike_tile * 64 + 11, :d].float()).abs().max()) < 0.05   # the spiked key takes all the weight


@pytest.mark.parametrize("M,N,K,mean", [(8192, 960, 320, 0.0), (2048, 640, 640, 3.0), (520, 1280, 1280, -8.0), (77, 72, 64, 0.5),
                                        (300, 3840, 1280, 0.0), (8192, 320, 320, 1.0),
                                        (4608, 10240, 1280, 2.0),     # 256^2 LDS-DMA kernel, row statistics from their own pass
                                        (4100, 3848, 1096, -1.0),     # the same, ragged M / N / K
                                        (16400, 960, 320, 1.5),       # >= 16384 rows: cost-model dispatch (UNet3D qkv, 5 K tiles)
                                        (16384, 2560, 320, 0.0)])     # GEGLU ff1 with the LayerNorm folded in
def test_gemm_layernorm_folded(ops, dev, M, N, K, mean):
    """LayerNorm folded into the consuming GEMM (spider_gemm_ln_bf16) against torch fp32 LayerNorm -> linear, plain / + residual /
    GEGLU; rows with a large common offset (|mean| >> std) exercise the mean-cancellation term."""
    x = (rnd(M, K, seed=1).float() + mean).to(BF)
    W, b = rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3, scale=0.2)
    ga, be = (1 + 0.2 * rnd(K, seed=4).float()).to(BF), rnd(K, seed=5, scale=0.2)
    res = rnd(M, N, seed=6)
    y = F.layer_norm(x.float(), (K,), ga.float(), be.float(), 1e-5)
    ref = y @ W.float().T + b.float()
    Wf, cs, cb = ops.fold_layernorm(W.to(dev), ga.to(dev), be.to(dev), b.to(dev))
    close(ops.gemm_ln(x.to(dev), Wf, cs, cb, eps=1e-5), ref, 1.5e-2, 1e-2, "gemm_ln", rel_to_std=True)
    close(ops.gemm_ln(x.to(dev), Wf, cs, cb, res=res.to(dev), eps=1e-5), ref + res.float(), 1.5e-2, 1e-2, "gemm_ln + res", rel_to_std=True)
    # the two-launch form it replaces (LayerNorm rounds its output to bf16 first): agreement to bf16 rounding
    two = ops.gemm(ops.layernorm(x.to(dev), ga.to(dev), be.to(dev), 1e-5), W.to(dev), bias=b.to(dev))
    close(ops.gemm_ln(x.to(dev), Wf, cs, cb, eps=1e-5), two.float(), 2.5e-2, 1.5e-2, "gemm_ln vs layernorm + gemm", rel_to_std=True)
    if N % 8 == 0:
        inner = N // 2
        # value * gelu(gate): where the value is ~0 and the gate is several sigma, the product inherits the value's absolute
        # error (the fold rounds W * gamma to bf16 once more, ~1e-3 of the pre-activation scale) times the gate -- a handful of
        # elements per million sit there, so the elementwise bound is wider and the aggregate error is bounded separately
        got, rg = ops.gemm_ln(x.to(dev), Wf, cs, cb, act="geglu", eps=1e-5), ref[:, :inner] * F.gelu(ref[:, inner:])
        close(got, rg, 5e-2, 2e-2, "gemm_ln geglu", rel_to_std=True)
        assert float((got.float().cpu() - rg).norm() / rg.norm()) < 6e-3    # three bf16 roundings (value, gelu(gate), product)



@pytest.mark.parametrize("B,N,K", [(8, 4608, 3584), (5, 3584, 18944), (16, 130, 1024), (1, 7, 64), (8, 18944, 3584), (3, 1000, 256)])
def test_skinny_gemm_fragment_major(ops, dev, B, N, K):
    """Batched-decode GEMV on the fragment-major weight copy (repack_fm16): plain / + bias + residual / gate-up SwiGLU, against
    fp32 torch and against the row-major kernels (same products, different summation order)."""
    W, x = rnd(N, K, seed=1, scale=0.05), rnd(B, K, seed=2)
    bias, res = rnd(N, seed=3), rnd(B, N, seed=4)
    ref = x.float() @ W.float().T
    Wd = W.to(dev)
    Wfm = ops.repack_fm16(Wd)
    assert Wfm.shape == ((N + 15) // 16, K // 64, 2, 64, 8)
    # the layout contract of include/spider_hip.h: piece (rg, kb, sx), lane 16 g + r -> W[16 rg + r, 64 kb + 32 sx + 8 g : + 8]
    rg, kb, sx, g, r = (N - 1) // 16, K // 64 - 1, 1, 3, (N - 1) % 16
    assert torch.equal(Wfm[rg, kb, sx, 16 * g + r].cpu(), W[16 * rg + r, 64 * kb + 32 * sx + 8 * g: 64 * kb + 32 * sx + 8 * g + 8])
    close(ops.gemv_fm(Wfm, x.to(dev), N), ref, 1e-2, 1e-2, "gemv_fm", rel_to_std=True)
    close(ops.gemv_fm(Wfm, x.to(dev), N, bias=bias.to(dev), res=res.to(dev)), ref + bias.float() + res.float(), 1.5e-2, 1e-2,
          "gemv_fm + bias + res", rel_to_std=True)
    if B <= 8:
        close(ops.gemv_fm(Wfm, x.to(dev), N), ops.gemv(Wd, x.to(dev)).float(), 1e-2, 1e-2, "gemv_fm vs row-major", rel_to_std=True)
    if N % 32 == 0:
        I = N // 2
        gg, uu = ref[:, :I], ref[:, I:]
        close(ops.gemv_swiglu_fm(Wfm, x.to(dev)), F.silu(gg) * uu, 1.5e-2, 2e-2, "gemv_swiglu_fm", rel_to_std=True)


@pytest.mark.parametrize("B,V,K", [(8, 152064, 3584), (5, 1000, 256), (16, 97, 64), (1, 33000, 128)])
def test_lm_head_argmax_fragment_major(ops, dev, B, V, K):
    W, x = rnd(V, K, seed=1, scale=0.05), rnd(B, K, seed=2)
    Wd, xd = W.to(dev), x.to(dev)
    Wfm = ops.repack_fm16(Wd)
    logits = torch.empty(B, V, dtype=BF, device=dev)
    ids = ops.lm_head_argmax_fm(Wfm, xd, V, logits=logits)
    ref = (x.float() @ W.float().T)
    close(logits, ref, 1e-2, 1e-2, "lm_head logits", rel_to_std=True)
    # the arg-max is taken over the bf16-rounded logits with ties -> lowest id: recompute it from the kernel's own logits
    lg = logits.float().cpu()
    assert ids.cpu().tolist() == lg.argmax(-1).tolist() or all(
        float(lg[b, int(ids[b])]) == float(lg[b].max()) and int(ids[b]) == int((lg[b] == lg[b].max()).nonzero()[0]) for b in range(B))
    # forced ties: identical rows of W give identical logits, the lowest token id must win
    W2 = W.clone(); W2[V - 1] = W2[3]; W2[V // 2] = W2[3]
    big = x.float() @ W2[3].float()
    xs = torch.where(big[:, None] > 0, x.float(), -x.float()).to(BF)       # make row 3's logit positive for every sequence
    W2[3] = W2[3] * 8; W2[V - 1] = W2[3]; W2[V // 2] = W2[3]
    ids2 = ops.lm_head_argmax_fm(ops.repack_fm16(W2.to(dev)), xs.to(dev), V)
    lg2 = (xs.float() @ W2.float().T)
    for b in range(B):
        if int(lg2[b].argmax()) in (3, V // 2, V - 1):
            assert int(ids2[b]) == min(3, V // 2), (b, int(ids2[b]))


@pytest.mark.parametrize("B,N,K", [(8, 4608, 3584), (6, 130, 1024)])
def test_skinny_fragment_major_folded_rmsnorm(ops, dev, B, N, K):
    """RMSNorm folded into the fragment-major projection (weights carry norm_w, the kernel applies rsqrt(mean x^2 + eps)):
    against fp32 RMSNorm -> linear, for the plain, gate/up and lm_head forms."""
    W, x, nw = rnd(N, K, seed=1, scale=0.05), rnd(B, K, seed=2, scale=3.0), (1 + 0.2 * rnd(K, seed=5).float()).to(BF)
    bias = rnd(N, seed=3)
    xn = x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6) * nw.float()
    ref = xn @ W.float().T
    Wfm = ops.repack_fm16(W.to(dev), nw.to(dev))
    close(ops.gemv_fm(Wfm, x.to(dev), N, bias=bias.to(dev), norm_eps=1e-6), ref + bias.float(), 1.5e-2, 1e-2, "gemv_fm folded norm", rel_to_std=True)
    if N % 32 == 0:
        I = N // 2
        close(ops.gemv_swiglu_fm(Wfm, x.to(dev), norm_eps=1e-6), F.silu(ref[:, :I]) * ref[:, I:], 1.5e-2, 2e-2, "gemv_swiglu_fm folded norm",
              rel_to_std=True)
    logits = torch.empty(B, N, dtype=BF, device=dev)
    ids = ops.lm_head_argmax_fm(Wfm, x.to(dev), N, logits=logits, norm_eps=1e-6)
    close(logits, ref, 1.5e-2, 1e-2, "lm_head_fm folded norm", rel_to_std=True)
    assert ids.cpu().tolist() == logits.float().cpu().argmax(-1).tolist()


@pytest.mark.parametrize("B2,n_tok,C,n_keys", [(2, 4096, 320, 77), (2, 1024, 640, 77), (2, 256, 1280, 77), (1, 64, 1280, 77), (2, 48, 64, 33)])
def test_xattn_fused(ops, dev, B2, n_tok, C, n_keys):
    """Fused cross-attention sub-block (LayerNorm + to_q + attention over <= 80 text keys + to_out + residual in one launch, the
    text K / V folded into the projections) against the fp32 composition of the separate ops."""
    H, d, LP = 8, C // 8, 80
    x = rnd(B2, n_tok, C, seed=1)
    Wq, Wo, bo = rnd(C, C, seed=2, scale=1.5 / math.sqrt(C)), rnd(C, C, seed=3, scale=1.0 / math.sqrt(C)), rnd(C, seed=4, scale=0.1)
    K, V = rnd(B2, n_keys, C, seed=5), rnd(B2, n_keys, C, seed=6)
    ga, be = (1 + 0.2 * rnd(C, seed=7).float()).to(BF), rnd(C, seed=8, scale=0.2)
    # fp32 reference
    y = F.layer_norm(x.float(), (C,), ga.float(), be.float(), 1e-5)
    q = (y @ Wq.float().T).view(B2, n_tok, H, d).transpose(1, 2)
    kh, vh = K.float().view(B2, n_keys, H, d).transpose(1, 2), V.float().view(B2, n_keys, H, d).transpose(1, 2)
    att = torch.softmax(q @ kh.transpose(-1, -2) / math.sqrt(d), -1) @ vh
    ref = x.float() + att.transpose(1, 2).reshape(B2, n_tok, C) @ Wo.float().T + bo.float()
    # fold (what UNetEngine._fold_cross does), with torch fp32 for the small products
    HL = H * LP
    kexp, vexp = torch.zeros(B2, H, LP, H, d), torch.zeros(B2, H, LP, H, d)
    ar = torch.arange(H)
    kexp[:, ar, :n_keys, ar, :] = K.float().view(B2, n_keys, H, d).permute(2, 0, 1, 3)
    vexp[:, ar, :n_keys, ar, :] = V.float().view(B2, n_keys, H, d).permute(2, 0, 1, 3)
    kexp, vexp = kexp.view(B2 * HL, C), vexp.view(B2, HL, C)
    scale = 1 / math.sqrt(d)
    mq = ((kexp @ (Wq.float() * ga.float()[None, :])) * scale).to(BF)                       # [B2*HL, C]
    cs = mq.float().sum(1)
    cb = (kexp @ (Wq.float() @ be.float())) * scale
    mo = torch.cat([(Wo.float() @ vexp[b].T) for b in range(B2)], 0).to(BF)                  # [B2*C, HL]
    got = ops.xattn_fused(x.to(dev), ops.repack_fm16(mq.to(dev)), ops.repack_fm16(mo.to(dev)), cs.to(dev).contiguous(), cb.to(dev).contiguous(),
                          bo.to(dev), B2, H, n_keys, eps=1e-5)
    close(got, ref, 2e-2, 1e-2, "xattn_fused", rel_to_std=True)
    assert float((got.float().cpu() - ref).norm() / ref.norm()) < 6e-3


# ---------------------------------------------------------------------------------------------------------------------------
# round 4: GroupNorm statistics from the producer of the normalised tensor (include/spider_hip.h: spider_conv_nhwc_gn,
# spider_groupnorm_apply_nhwc, spider_gemm_gn_in). Replaces diffusers' stand-alone torch.nn.GroupNorm pass between
# ResnetBlock2D.conv1 and norm2 / conv2 and the next norm / Transformer2DModel.norm and proj_in (custom_sd.py:634-639).
# ---------------------------------------------------------------------------------------------------------------------------
def _group_sums(y, groups, cr):
    """fp64 reference of the partials: y [B, H, W, C] (16-bit values as stored) -> [B, HW / cr, G, 2]"""
    B, H, W, C = y.shape
    v = y.double().reshape(B, H * W // cr, cr, groups, C // groups)
    return torch.stack([v.sum((2, 4)), (v * v).sum((2, 4))], -1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [
    (2, 64, 64, 320, 320, 1),      # 8192 rows, K = 2880: LDS-DMA kernel, 64-row tiles, no split-K -> epilogue statistics
    (2, 32, 32, 640, 640, 1),      # 2048 rows, K = 5760: LDS-DMA kernel + split-K -> statistics from the reduce
    (2, 32, 32, 320, 640, 1),      # channel change (conv1 of a down block)
    (2, 16, 16, 1280, 1280, 1),    # 512 rows: split-K, 16-row chunks
    (2, 8, 8, 1280, 1280, 1),      # 128 rows
    (2, 64, 64, 320, 320, 2),      # stride-2 downsampler -> 32 x 32 output
    (1, 24, 24, 640, 640, 1),      # 576 rows: 9 chunks of 64... below 2048 rows: 16-row chunks
])
def test_conv_gn_partials_match_output(dev, shape, dtype):
    from spider_amd import ops
    B, H, W, Cin, Cout, stride = shape
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, H, W, Cin, generator=g, device=dev).to(dtype)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g, device=dev) * (1.0 / (3 * Cin ** 0.5))).to(dtype)
    bias = (torch.randn(Cout, generator=g, device=dev) * 0.1).to(dtype)
    rb = (torch.randn(B, Cout, generator=g, device=dev) * 0.1).to(dtype)
    res = torch.randn(B, (H - 1) // stride + 1, (W - 1) // stride + 1, Cout, generator=g, device=dev).to(dtype)
    ref = ops.conv2d(x, w, bias=bias, rowbias=rb, res=res, stride=stride, pad=1)
    out, part = ops.conv2d(x, w, bias=bias, rowbias=rb, res=res, stride=stride, pad=1, gn_groups=32)
    assert torch.equal(out, ref), "the statistics-producing conv must write the same output"
    HW = out.shape[1] * out.shape[2]
    cr = HW // part.nchunk
    assert cr in (16, 64) and part.nchunk * cr == HW and tuple(part.t.shape) == (B, HW // cr, 32, 2)
    want = _group_sums(out.cpu(), 32, cr)
    got = part.t.double().cpu()
    err = (got - want).abs().max() / want.abs().max()
    assert float(err) < 1e-5, float(err)            # fp32 sums of up to 64 x 40 values against fp64
    # consumer 1: apply-only GroupNorm on the producer's partials == the two-pass GroupNorm (same values up to the fp32 sum order)
    gam = (1 + 0.1 * torch.randn(Cout, generator=g, device=dev)).to(dtype)
    bet = (0.1 * torch.randn(Cout, generator=g, device=dev)).to(dtype)
    y_ref = ops.groupnorm(out, gam, bet, 32, 1e-5, True)
    y = ops.groupnorm(out, gam, bet, 32, 1e-5, True, partial=part)
    d = (y.float() - y_ref.float()).abs().max() / y_ref.float().abs().max()
    assert float(d) < 4e-3, float(d)                      # at most a 16-bit rounding flip
    assert float((y.float() - y_ref.float()).norm() / y_ref.float().norm()) < 2e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,HW,C,N", [(2, 4096, 320, 320), (2, 1024, 640, 640), (2, 256, 1280, 1280), (2, 64, 1280, 1280)])
def test_gemm_with_groupnorm_on_A_matches_groupnorm_then_gemm(dev, B, HW, C, N, dtype):
    from spider_amd import ops
    g = torch.Generator(device=dev).manual_seed(1)
    x = (torch.randn(B, HW, C, generator=g, device=dev) * 1.5 + 0.3).to(dtype)
    W = (torch.randn(N, C, generator=g, device=dev) / C ** 0.5).to(dtype)
    bias = (torch.randn(N, generator=g, device=dev) * 0.1).to(dtype)
    gam = (1 + 0.1 * torch.randn(C, generator=g, device=dev)).to(dtype)
    bet = (0.1 * torch.randn(C, generator=g, device=dev)).to(dtype)
    part = ops.groupnorm_stats(x, 32, HW // 16)
    ref = ops.gemm(ops.groupnorm(x, gam, bet, 32, 1e-6, False), W, bias=bias)
    got, got32 = ops.gemm_gn_in(x, W, part, gam, bet, HW, 1e-6, bias=bias, want32=True)
    r = float((got.float() - ref.float()).norm() / ref.float().norm())
    assert r < 3e-4, r                                    # same math; the group mean / rstd differ in their last fp32 bits
    assert torch.equal(got, got32.to(dtype)), "the fp32 copy rounds to the 16-bit output"


# ------------------------------------------------------------------------------------------ weight-stationary streaming conv (round 5)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W,Cin,Cout,up", [
    (2, 8, 8, 64, 64, None),          # 128 output pixels: 8 wave-private K-groups, 2 channel blocks
    (2, 8, 8, 1280, 1280, None),      # SD-v1.5 8^2 conv2: 40 channel blocks, 5 K splits
    (2, 8, 8, 2560, 1280, None),      # up-block conv1 on the skip concat
    (1, 8, 8, 96, 80, None),          # 64 pixels (rows past M are zero rows), Cout not a multiple of 32
    (2, 7, 5, 160, 48, None),         # odd, non-square map
    (2, 16, 16, 320, 96, None),       # 512 pixels: 4 row groups x 2 K-groups, shared slabs
    (2, 16, 16, 1280, 1280, None),    # SD-v1.5 16^2 conv2
    (2, 16, 16, 1920, 1280, None),    # 60 channel blocks: uneven K splits
    (3, 12, 12, 64, 32, None),        # 432 pixels over three images
    (2, 8, 8, 1280, 1280, (16, 16)),  # the 8^2 -> 16^2 upsampler (fused nearest 2x)
    (2, 7, 8, 64, 64, (13, 16)),      # upsample to 2 * in - 1
])
def test_wstream_conv_matches_reference_and_tile_path(dev, B, H, W, Cin, Cout, up, dtype, monkeypatch):
    """conv_ex on a MARKED weight with <= 512 output pixels runs wstream_kernel (w_tiled = 2) -- against torch fp32 and against the
    tile kernels (same products, another summation order), with every epilogue operand of the resnet convs."""
    from spider_amd import ops
    monkeypatch.setattr(ops, "WS_MAX_M", 512)        # the 512-pixel form exists and is tested; the engines use it up to ops.WS_MAX_M
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(B, H, W, Cin, generator=g, device=dev).to(dtype)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g, device=dev) * (1.0 / (3 * Cin ** 0.5))).to(dtype)
    bias = (torch.randn(Cout, generator=g, device=dev) * 0.1).to(dtype)
    rb = (torch.randn(B, Cout, generator=g, device=dev) * 0.1).to(dtype)
    Ho, Wo = up if up is not None else (H, W)
    res32 = torch.randn(B, Ho, Wo, Cout, generator=g, device=dev)
    wm = ops.mark_weight(w.clone())
    assert ops._ws_eligible(B, H, W, Cin, Cout, 3, 3, 1, (1, 1), 1, up, Ho, Wo, None)
    got, got32 = ops.conv_ex(x, wm, bias=bias, rowbias=rb, pad=(1, 1), up_size=up, res32=res32, want32=True)
    assert getattr(wm, "_spider_fm", None) is not None, "the streaming kernel's weight copy was not built: wrong path"
    xi = x.float().permute(0, 3, 1, 2)
    if up is not None:
        xi = F.interpolate(xi, size=up, mode="nearest")
    ref = F.conv2d(xi, w.float().permute(0, 3, 1, 2), bias.float(), padding=1).permute(0, 2, 3, 1) + rb.float()[:, None, None, :] + res32
    rel = float((got32 - ref).norm() / ref.norm())
    assert rel < 1e-5 * max(1.0, (9 * Cin) ** 0.5 / 30), f"fp32 output vs torch fp32: {rel}"       # fp32 accumulation only
    assert torch.equal(got, got32.to(dtype)), "the 16-bit output is the rounded fp32 output"
    old, old32 = ops.conv_ex(x, w, bias=bias, rowbias=rb, pad=(1, 1), up_size=up, res32=res32, want32=True)   # unmarked weight: tile kernels
    assert float((got32 - old32).norm() / old32.norm()) < 2e-6
    # 16-bit residual + plain output, no fp32 stream
    res = torch.randn(B, Ho, Wo, Cout, generator=g, device=dev).to(dtype)
    a = ops.conv_ex(x, wm, bias=bias, pad=(1, 1), up_size=up, res=res)
    b_ = ops.conv_ex(x, w, bias=bias, pad=(1, 1), up_size=up, res=res)
    assert float((a.float() - b_.float()).norm() / b_.float().norm()) < (2e-4 if dtype == torch.float16 else 2e-3)
    # statistics for the consumer GroupNorm still arrive (reduce kernel or the caller's pass)
    if Cout % 32 == 0 and (Ho * Wo) % 16 == 0:
        o2, part = ops.conv_ex(x, wm, bias=bias, pad=(1, 1), up_size=up, gn_groups=32)
        assert torch.equal(o2, ops.conv_ex(x, wm, bias=bias, pad=(1, 1), up_size=up))
        if part is not None:        # (none with the in-launch combine: the consumer GroupNorm runs its own one-launch kernel)
            cr = (Ho * Wo) // part.nchunk
            want = _group_sums(o2.cpu(), 32, cr)
            assert float((part.t.double().cpu() - want).abs().max() / want.abs().max()) < 1e-5
    # (the in-launch combine against the reduce kernel, bit for bit and under load: test_wstream_inlaunch_combine_stress_under_load)
    cnt = ops._workspace(x.device)[ops.WS_BYTES // 4:]
    assert int(cnt.view(torch.int32).abs().sum()) == 0, "arrival counters must be zero between calls"


def test_wstream_graph_replay_is_bit_identical(dev, monkeypatch):
    from spider_amd import ops
    monkeypatch.setattr(ops, "WS_MAX_M", 512)
    g = torch.Generator(device=dev).manual_seed(6)
    x = torch.randn(2, 16, 16, 640, generator=g, device=dev).half()
    w = ops.mark_weight((torch.randn(1280, 3, 3, 640, generator=g, device=dev) * 0.01).half())
    eager = ops.conv_ex(x, w, pad=(1, 1))
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = ops.conv_ex(x, w, pad=(1, 1))
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


# ------------------------------------------------------------------------------------------ tall store-heavy linears (round 5)
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,streams,extras", [
    # (shapes the cost model sends to the two-blocks-per-CU kernel: checked with SPIDER_GEMM_TRACE=1, gpurun_out/r05_tall_dispatch.txt)
    (92160, 320, 320, "both", True),         # the zeroscope shape: + bias + rowbias + activation + out_scale, row-order epilogue through LDS
    (92160 + 40, 320, 640, "both", False),   # ragged last row tile (rows past M are masked in the row-order pass)
    (65536, 320, 320, "res32", False),       # SD-v1.5 at CFG batch 16
    (32768, 320, 1280, "c32d", False),
    (65536, 640, 640, "both", False),        # 4 column tiles
    (92160, 320, 320, "none", True),         # no fp32 stream: fragment-order stores on the same kernel
    # (and neighbours that stay on the older kernels)
    (16384, 320, 320, "both", False),
    (20000, 480, 192, "both", False),        # K not a multiple of 64 (zero-filled tail chunks), 3 column tiles
])
def test_tall_linears_on_the_two_blocks_per_cu_kernel(ops, dev, dt, M, N, K, streams, extras):
    """gemm_dma_kernel<160, 2> + epilogue_lds (csrc/gemm.hip): chosen by the tall-problem cost model for >= 16384 rows and N in 160-wide
    tiles. fp32 output (c32d) at accumulation distance from the fp64 reference; the 16-bit output is that value rounded once."""
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(dt)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dt)
    b = torch.randn(N, generator=g).to(dt)
    kw, ref = {}, A.double() @ W.double().T + b.double()
    if extras:
        G = 16
        rb = torch.randn(G, N, generator=g).to(dt)
        kw = dict(rowbias=rb.to(dev), rows_per_group=M // G, act="silu", out_scale=0.5)
        ref = F.silu(ref + rb.double().repeat_interleave(M // G, 0))
    r32 = torch.randn(M, N, generator=g)
    if streams in ("both", "res32"):
        kw["res32"] = r32.to(dev)
        ref = ref + r32.double()
    if extras:
        ref = ref * 0.5
    want32 = streams in ("both", "c32d")
    out = ops.gemm(A.to(dev), W.to(dev), bias=b.to(dev), want32=want32, **kw)
    y, y32 = (out if want32 else (out, None))
    ulp = 2.0 ** -11 if dt == torch.float16 else 2.0 ** -8
    rel = lambda t: float((t.double().cpu() - ref).norm() / ref.norm())
    assert rel(y) < ulp, (rel(y), ulp)
    if y32 is not None:
        assert rel(y32) < 3e-6, rel(y32)                                   # fp32 accumulation over K <= 1280
        assert torch.equal(y.cpu(), y32.cpu().to(dt))                      # C is c32d rounded once
