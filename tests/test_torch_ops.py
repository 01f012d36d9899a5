"""`torch.ops.spider_hip.*`: the HIP kernels as PyTorch custom ops (spider_amd/torch_ops.py). CPU: every op of SURVEY.md
section 8b's export list is registered with a schema and a fake (meta) implementation that FakeTensor tracing can run without a
GPU. GPU: the ops return what spider_amd.ops returns, pass torch.library.opcheck, and survive torch.compile as opaque nodes."""
import pytest
import torch

import spider_amd.torch_ops as T

BF = torch.bfloat16


def test_all_ops_registered_with_schema_and_fake_impl():
    assert len(T.OP_NAMES) == 14
    for n in T.OP_NAMES:
        op = getattr(torch.ops.spider_hip, n)
        assert op.default._schema.name == f"spider_hip::{n}"
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        e = lambda *s, dt=BF: torch.empty(*s, dtype=dt, device="cuda")
        o = torch.ops.spider_hip
        assert o.rmsnorm(e(4, 64), e(64), 1e-6, None).shape == (4, 64)
        assert o.swiglu(e(4, 128)).shape == (4, 64) and o.geglu(e(2, 9, 128)).shape == (2, 9, 64)
        assert o.linear_bf16(e(3, 5, 64), e(96, 64), None).shape == (3, 5, 96)
        assert o.lm_head_argmax(e(2, 64), e(1000, 64), None, 1e-6).dtype == torch.int32
        assert o.attn_decode(e(2, 8, 128), e(2, 2, 64, 128), e(2, 2, 64, 128), e(2, dt=torch.int32), None).shape == (2, 1024)
        assert o.attn_prefill_causal(e(1, 7, 512), e(1, 7, 128), e(1, 7, 128), 4, 1).shape == (1, 7, 512)
        assert o.groupnorm_silu(e(2, 8, 8, 64), e(64), e(64), 32, 1e-5, True).shape == (2, 8, 8, 64)
        assert o.conv2d_nhwc(e(2, 16, 16, 64), e(128, 3, 3, 64), None, 2, 1).shape == (2, 8, 8, 128)
        assert o.attn_self(e(2, 64, 320), e(2, 64, 320), e(2, 64, 320), 8).shape == (2, 64, 320)
        assert o.attn_cross_kv77(e(2, 64, 320), e(2, 77, 320), e(2, 77, 320), 8).shape == (2, 64, 320)
        assert o.attn_consistent(e(2, 256, 128), e(2, 256, 128), e(2, 256, 128), 2, e(4, dt=torch.int64), 64, 0).shape == (2, 256, 128)
        assert o.cfg_step(e(2, 8, 8, 4, dt=torch.float32), e(1, 4, 8, 8, dt=torch.float32), 7.5, 1.0, -0.1).shape == (1, 4, 8, 8)
        assert o.rope_qk_(e(1, 3, 4 * 128), e(3, dt=torch.int32), e(3, dt=torch.int32), e(64, 128, dt=torch.float32), e(1, 3, 2, 128),
                          e(1, 1, 16, 128), e(1, 1, 16, 128), 2, 1, 128) is None


@pytest.mark.gpu
def test_custom_ops_match_ops_and_pass_opcheck(dev):
    from spider_amd import ops
    g = torch.Generator().manual_seed(0)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(BF).to(dev)
    o = torch.ops.spider_hip
    x, w, res = r(5, 256), r(256), r(5, 256)
    assert torch.equal(o.rmsnorm(x, w, 1e-6, res), ops.rmsnorm(x, w, 1e-6, res=res))
    A, W, b = r(130, 64), r(96, 64, sc=0.1), r(96)
    assert torch.equal(o.linear_bf16(A, W, b), ops.gemm(A, W, bias=b))
    gu = r(7, 256)
    assert torch.equal(o.swiglu(gu), ops.swiglu(gu)) and torch.equal(o.geglu(gu), ops.geglu(gu))
    xi, wc, bc = r(2, 16, 16, 64), r(128, 3, 3, 64, sc=0.05), r(128)
    assert torch.equal(o.conv2d_nhwc(xi, wc, bc, 1, 1), ops.conv2d(xi, wc, bias=bc, stride=1, pad=1))
    ga, be = r(64), r(64)
    assert torch.equal(o.groupnorm_silu(xi, ga, be, 32, 1e-5, True), ops.groupnorm(xi, ga, be, 32, 1e-5, True))
    q, k, v = r(2, 256, 320), r(2, 256, 320), r(2, 256, 320)
    assert torch.equal(o.attn_self(q, k, v, 8), ops.attention(q, k, v, 8))
    kc, vc = r(2, 77, 320), r(2, 77, 320)
    assert torch.equal(o.attn_cross_kv77(q, kc, vc, 8), ops.attention(q, kc, vc, 8))
    assert torch.equal(o.attn_prefill_causal(q, k[..., :80], v[..., :80], 4, 1), ops.attention(q, k[..., :80], v[..., :80], 4, n_kv_heads=1, causal=True))
    Wl = r(1000, 256, sc=0.1)
    assert torch.equal(o.lm_head_argmax(x, Wl, w, 1e-6), ops.lm_head_argmax(Wl, x, norm_w=w, eps=1e-6))
    eps2 = torch.randn(2, 8, 8, 4, generator=g).to(dev)
    lat = torch.randn(1, 4, 8, 8, generator=g).to(dev)
    ref = 0.9 * lat - 0.2 * ops.cfg_combine(eps2, 7.5)
    assert torch.allclose(o.cfg_step(eps2, lat, 7.5, 0.9, -0.2), ref, atol=1e-6)
    # schema / fake-tensor / functionalisation checks of the registration itself
    for op, args in ((o.rmsnorm, (x, w, 1e-6, res)), (o.linear_bf16, (A, W, b)), (o.geglu, (gu,)), (o.attn_self, (q, k, v, 8)),
                     (o.conv2d_nhwc, (xi, wc, bc, 1, 1))):
        torch.library.opcheck(op, args, test_utils=("test_schema", "test_faketensor"))
    # an in-place op declares what it mutates: rope + KV append
    qkv = r(1, 3, 4 * 128)
    pos = torch.arange(3, dtype=torch.int32, device=dev)
    cs = torch.cat([torch.ones(16, 64), torch.zeros(16, 64)], 1).to(dev)           # angle 0: rotation is the identity
    q_out, kcache, vcache = torch.zeros(1, 3, 2, 128, dtype=BF, device=dev), torch.zeros(1, 1, 16, 128, dtype=BF, device=dev), torch.zeros(1, 1, 16, 128, dtype=BF, device=dev)
    o.rope_qk_(qkv, pos, pos, cs, q_out, kcache, vcache, 2, 1, 128)
    assert torch.equal(q_out.view(1, 3, 256), qkv[..., :256]) and torch.equal(kcache[0, 0, :3], qkv[0, :, 256:384]) and torch.equal(vcache[0, 0, :3], qkv[0, :, 384:])


@pytest.mark.gpu
def test_custom_ops_inside_torch_compile(dev):
    """The ops are opaque nodes for the tracer: a function mixing them with torch ops compiles (backend "eager": graph capture
    and fake-tensor propagation only, no code generation) and gives the eager result."""
    g = torch.Generator().manual_seed(1)
    x, w = (torch.randn(6, 128, generator=g)).to(BF).to(dev), torch.ones(128, dtype=BF, device=dev)
    W = (torch.randn(256, 128, generator=g) * 0.1).to(BF).to(dev)

    def f(x):
        h = torch.ops.spider_hip.rmsnorm(x, w, 1e-6, None)
        return torch.ops.spider_hip.swiglu(torch.ops.spider_hip.linear_bf16(h, W, None)) + 1

    ref = f(x)
    got = torch.compile(f, backend="eager", fullgraph=True)(x)
    assert torch.equal(got, ref)
