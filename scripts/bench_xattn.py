"""Tuning aid: the fused cross-attention sub-block (spider_xattn_fused_bf16) against the three launches it replaces
(gemm_ln to_q, 77-key flash attention, to_out GEMM + residual) at the SD-v1.5 shapes. python scripts/bench_xattn.py"""
import math, sys, torch
from spider_amd import ops
dev = torch.device("cuda:0")
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (2 * n)
for n_tok, C in [(4096, 320), (1024, 640), (256, 1280), (64, 1280)]:
    B2, H, LP = 2, 8, 80
    x = torch.randn(B2, n_tok, C, device=dev).bfloat16()
    Wq = (torch.randn(C, C, device=dev) / math.sqrt(C)).bfloat16(); Wo = (torch.randn(C, C, device=dev) / math.sqrt(C)).bfloat16()
    bo = torch.zeros(C, device=dev).bfloat16(); ga = torch.ones(C, device=dev).bfloat16(); be = torch.zeros(C, device=dev).bfloat16()
    kv = torch.randn(B2, 77, 2 * C, device=dev).bfloat16()
    Wf, cs, cb = ops.fold_layernorm(Wq, ga, be)
    def unfused():
        q = ops.gemm_ln(x, Wf, cs, cb)
        o = ops.attention(q, kv[..., :C], kv[..., C:], H)
        return ops.gemm(o, Wo, bias=bo, res=x)
    mq = torch.randn(B2 * H * LP, C, device=dev).bfloat16() * 0.05; mo = torch.randn(B2 * C, H * LP, device=dev).bfloat16() * 0.05
    mqf, mof = ops.repack_fm16(mq), ops.repack_fm16(mo)
    c1 = torch.zeros(B2 * H * LP, device=dev); c2 = torch.zeros(B2 * H * LP, device=dev)
    fused = lambda: ops.xattn_fused(x, mqf, mof, c1, c2, bo, B2, H, 77)
    fl = 4.0 * B2 * n_tok * C * H * LP
    tu, tf = t(unfused), t(fused)
    print(f"N={n_tok:5d} C={C:5d}: unfused (3 launches) {tu:6.1f} us   fused {tf:6.1f} us = {fl / tf / 1e6:6.1f} TF/s of the folded GEMMs", flush=True)
