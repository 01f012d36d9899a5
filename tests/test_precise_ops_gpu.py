"""GPU parity tests of the "precise" fp32-operand entry points (include/spider_hip.h, ABI v4) on BOTH of their
routes: the a32 kernels (hi / lo split inside the GEMM, two MFMAs per K step) and "split once, doubled K" (spider_row_split_f32 /
spider_groupnorm_f32in_split_nhwc, then the 16-bit tile kernels over [hi | lo] against [W | W]). Reference: plain torch in fp64 on the
same seeded inputs (W, gamma, beta, bias pre-rounded to the 16-bit format, as the engines hold them).

What the mode promises is the A operand at ~22 bits: the FP32 output of a call (c32d) must sit at fp32-accumulation distance from the
reference (f16: bound 2e-5 relative L2, 30x under one 16-bit rounding of A; bf16's two halves carry 16 bits: 6e-5), and the two routes must agree with each other to the same
distance; the 16-bit output is that value rounded once (bound: one ulp of the format in relative L2).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTS = [torch.float16, torch.bfloat16]
ULP = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}
B32 = {torch.float16: 2e-5, torch.bfloat16: 6e-5}      # f16: hi + lo carry 22 bits of A; bf16: 16


def rnd(g, *shape, scale=1.0, dt=torch.float16):
    return (torch.randn(*shape, generator=g) * scale).to(dt)


def rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert bool(torch.isfinite(got).all())
    return float((got - ref).norm() / ref.norm())


@pytest.fixture(scope="module")
def ops(dev):
    from spider_amd import ops as o
    return o


@pytest.fixture(params=["a32_kernel", "split_once"])
def route(request, ops, monkeypatch):
    """a32_kernel: the threshold out of reach; split_once: every call large enough (weights are marked so [W | W] is kept)"""
    monkeypatch.setattr(ops, "A32_DUP_MIN_FLOP", float("inf") if request.param == "a32_kernel" else 0.0)
    monkeypatch.setattr(ops, "A32_SPLIT_MIN_FLOP", float("inf"))
    monkeypatch.setattr(ops, "A32_DUP_MIN_M", 0)
    return request.param


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,K", [(1, 8), (7, 320), (64, 512), (33, 520), (16, 1280), (5, 2560), (3, 4104), (2, 8192)])
def test_row_split_plain_is_the_two_term_expansion(ops, dev, dt, M, K):
    g = torch.Generator().manual_seed(M * 131 + K)
    x = torch.randn(M, K, generator=g) * 3 + 0.25
    y = ops.row_split(x.to(dev), dt).cpu()
    hi, lo = y[:, :K], y[:, K:]
    assert torch.equal(hi, x.to(dt))                                      # round-to-nearest-even of x
    assert torch.equal(lo, (x - hi.float()).to(dt))                       # and of the exact remainder
    assert float(((hi.double() + lo.double()) - x.double()).abs().max() / x.abs().max()) < ULP[dt] ** 2 * 4


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,K,with_beta", [(9, 320, True), (64, 640, True), (5, 1280, False), (3, 2048, True), (2, 5120, True)])
def test_row_split_layernorm(ops, dev, dt, M, K, with_beta):
    g = torch.Generator().manual_seed(K + M)
    x = torch.randn(M, K, generator=g) * 2 + 1.5
    ga, be = (rnd(g, K, dt=dt).float() * 0.1 + 1).to(dt), rnd(g, K, scale=0.1, dt=dt)
    y = ops.row_split(x.to(dev), dt, ga.to(dev), be.to(dev) if with_beta else None, 1e-5).cpu()
    ref = F.layer_norm(x.double(), (K,), ga.double(), be.double() if with_beta else None, 1e-5)
    assert rel(y[:, :K].double() + y[:, K:].double(), ref) < (2e-6 if dt == torch.float16 else 3e-5)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,K", [(64, 320, 320), (100, 8, 64), (2048, 1280, 640), (8192, 320, 1280), (512, 2560, 1280)])
def test_gemm_a32(ops, dev, route, dt, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * 1.5 + 2.0
    W, b = ops.mark_weight(rnd(g, N, K, scale=K ** -0.5, dt=dt).to(dev)), rnd(g, N, dt=dt)
    r32 = torch.randn(M, N, generator=g)
    y, y32 = ops.gemm_a32(A.to(dev), W, bias=b.to(dev), res32=r32.to(dev), want32=True)
    ref = A.double() @ W.cpu().double().T + b.double() + r32.double()
    assert rel(y32, ref) < B32[dt], route
    assert rel(y, ref) < ULP[dt], route


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("geglu", [False, True])
@pytest.mark.parametrize("M,N,K", [(64, 960, 320), (2048, 1920, 640), (8192, 2560, 320), (100, 640, 1280)])
def test_gemm_ln_a32(ops, dev, route, dt, geglu, M, N, K):
    g = torch.Generator().manual_seed(M + N + K + geglu)
    A = torch.randn(M, K, generator=g) * 1.5 + 1.0
    W, b = ops.mark_weight(rnd(g, N, K, scale=K ** -0.5, dt=dt).to(dev)), rnd(g, N, dt=dt)
    ga, be = (rnd(g, K, dt=dt).float() * 0.1 + 1).to(dt), rnd(g, K, scale=0.1, dt=dt)
    fold = ops.fold_layernorm_exact(W, ga.to(dev), be.to(dev), b.to(dev))
    y = ops.gemm_ln_a32(A.to(dev), *fold, act="geglu_exact" if geglu else None)
    ref = F.layer_norm(A.double(), (K,), ga.double(), be.double(), 1e-5) @ W.cpu().double().T + b.double()
    if geglu:
        v, gt = ref.chunk(2, -1)
        ref = v * F.gelu(gt)
    assert rel(y, ref) < ULP[dt], route                                    # 16-bit output only: one rounding of an ~exact value


def gn_partial(x, G, rows=16):
    """[B, HW / rows, G, 2] (sum, sum of squares) per chunk of `rows` pixels and group -- the form a producing conv leaves behind"""
    B, HW, C = x.shape
    xg = x.double().view(B, HW // rows, rows, G, C // G)
    return torch.stack([xg.sum((2, 4)), (xg * xg).sum((2, 4))], -1).float().contiguous()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,HW,N,K", [(2, 64, 320, 320), (2, 1024, 640, 640), (1, 4096, 320, 320), (3, 256, 1280, 1280)])
def test_gemm_gn_in_a32(ops, dev, route, dt, B, HW, N, K):
    g = torch.Generator().manual_seed(B + HW + N)
    x = torch.randn(B, HW, K, generator=g) * 2 + 0.5
    W, b = ops.mark_weight(rnd(g, N, K, scale=K ** -0.5, dt=dt).to(dev)), rnd(g, N, dt=dt)
    ga, be = (rnd(g, K, dt=dt).float() * 0.1 + 1).to(dt), rnd(g, K, scale=0.1, dt=dt)
    part = ops.GnPartial(gn_partial(x, 32).to(dev), HW // 16, 32)
    y, y32 = ops.gemm_gn_in_a32(x.to(dev), W, part, ga.to(dev), be.to(dev), HW, 1e-6, bias=b.to(dev), want32=True)
    ref = F.group_norm(x.double().transpose(1, 2), 32, ga.double(), be.double(), 1e-6).transpose(1, 2) @ W.cpu().double().T + b.double()
    assert rel(y32, ref) < B32[dt], route
    assert rel(y, ref) < ULP[dt], route


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("silu", [False, True])
@pytest.mark.parametrize("B,HW,C", [(2, 64, 320), (1, 1000, 640), (2, 4096, 320), (3, 60, 1280), (1, 256, 2560)])
def test_groupnorm_f32in_split(ops, dev, dt, silu, B, HW, C):
    g = torch.Generator().manual_seed(B + HW + C)
    x = torch.randn(B, HW, C, generator=g) * 2 + 0.5
    ga, be = (rnd(g, C, dt=dt).float() * 0.1 + 1).to(dt), rnd(g, C, scale=0.1, dt=dt)
    y = ops.groupnorm_f32in_split(x.to(dev), ga.to(dev), be.to(dev), 32, 1e-5, silu).cpu()
    ref = F.group_norm(x.double().transpose(1, 2), 32, ga.double(), be.double(), 1e-5).transpose(1, 2)
    ref = F.silu(ref) if silu else ref
    assert y.shape == (B, HW, 2 * C)
    assert rel(y[..., :C].double() + y[..., C:].double(), ref) < (3e-6 if dt == torch.float16 else 3e-5)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,H,W,Cin,Cout,ks,stride,up,extras", [
    (2, 16, 16, 320, 640, 1, 1, False, False),        # ResnetBlock2D.conv_shortcut
    (2, 32, 32, 320, 320, 3, 2, False, False),        # Downsample2D
    (2, 8, 8, 1280, 1280, 3, 1, True, False),         # Upsample2D (nearest 2x read in place)
    (1, 9, 11, 64, 8, 3, 1, False, True),             # ragged map, rowbias + out_scale
    (2, 16, 16, 640, 320, 3, 1, False, True),
])
def test_conv_a32(ops, dev, route, dt, B, H, W, Cin, Cout, ks, stride, up, extras):
    g = torch.Generator().manual_seed(H * W + Cin + Cout)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = ops.mark_weight(rnd(g, Cout, ks, ks, Cin, scale=(ks * ks * Cin) ** -0.5, dt=dt).to(dev))
    b = rnd(g, Cout, dt=dt)
    up_size = (2 * H, 2 * W) if up else None
    xi = x.double().permute(0, 3, 1, 2)
    if up:
        xi = F.interpolate(xi, size=up_size, mode="nearest")
    ref = F.conv2d(xi, w.cpu().double().permute(0, 3, 1, 2), b.double(), stride=stride, padding=ks // 2).permute(0, 2, 3, 1)
    r32 = torch.randn(*ref.shape, generator=g)
    kw = {}
    if extras:         # + time-embedding row bias and an output scale (a 16-bit `res` would round the sum first, by its contract)
        rb = rnd(g, B, Cout, dt=dt)
        kw = dict(rowbias=rb.to(dev), out_scale=0.5)
        ref = (ref + rb.double()[:, None, None, :] + r32.double()) * 0.5
    else:
        ref = ref + r32.double()
    y, y32 = ops.conv_a32(x.to(dev), w, bias=b.to(dev), stride=stride, pad=(ks // 2, ks // 2), up_size=up_size, res32=r32.to(dev), want32=True, **kw)
    assert rel(y32, ref) < B32[dt], route
    assert rel(y, ref) < ULP[dt], route


def test_split_once_needs_a_marked_weight(ops, dev, monkeypatch):
    """an unmarked W (a slice, an activation used as W) never gets a [W | W] copy attached: the call stays on the a32 kernel"""
    monkeypatch.setattr(ops, "A32_DUP_MIN_FLOP", 0.0)
    monkeypatch.setattr(ops, "A32_DUP_MIN_M", 0)
    g = torch.Generator().manual_seed(5)
    W = rnd(g, 320, 320, scale=320 ** -0.5).to(dev)
    A = torch.randn(64, 320, generator=g)
    y32 = ops.gemm_a32(A.to(dev), W, want32=True)[1]
    assert not hasattr(W, "_spider_dup")
    assert rel(y32, A.double() @ W.cpu().double().T) < 2e-5
    Wm = ops.mark_weight(W.clone())
    y32m = ops.gemm_a32(A.to(dev), Wm, want32=True)[1]
    assert tuple(Wm._spider_dup.shape) == (320, 640) and torch.equal(Wm._spider_dup[:, :320], Wm) and torch.equal(Wm._spider_dup[:, 320:], Wm)
    assert rel(y32m, y32.cpu().double()) < 2e-5
    Wm.mul_(2)                                                             # an in-place update invalidates the copy
    y32u = ops.gemm_a32(A.to(dev), Wm, want32=True)[1]
    assert rel(y32u, 2 * (A.double() @ W.cpu().double().T)) < 2e-5
