"""Consistent self-attention (keep-bits mask) vs dense attention at the SDXL story shapes (768^2, 4 panels, 2 CFG groups)."""
import torch
from spider_amd import ops
dev = torch.device("cuda:0")
def t(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best
for N, H, d in [(2304, 10, 64), (576, 20, 64)]:
    L = 4 * N; C = H * d
    q = torch.randn(2, L, C, device=dev).bfloat16(); kv = torch.randn(2, L, 2 * C, device=dev).bfloat16()
    bits = ops.pack_keep_bits(torch.rand(L, device=dev), 0.5, L)
    dense = t(lambda: ops.attention(q, kv[..., :C], kv[..., C:], H))
    keep = t(lambda: ops.attention(q, kv[..., :C], kv[..., C:], H, keep_bits=bits, blk=N, q_off=0))
    ki, tl = ops.story_key_lists(bits, L, N, 0, 4, 0)
    kl = t(lambda: ops.attention_keylist(q, kv[..., :C], kv[..., C:], H, ki, tl))
    fl = 4 * 2 * H * L * L * d
    print(f"4x{N} tokens, {H} heads: dense {dense:7.1f} us ({fl / dense / 1e6:.0f} TF/s)   consistent-SA mask {keep:7.1f} us ({fl / keep / 1e6:.0f} TF/s dense-equivalent)   key lists {kl:7.1f} us", flush=True)
