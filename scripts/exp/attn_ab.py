"""In-process A/B: PLAIN instantiation of the flash kernel vs the general one (forced by an all-zero kv_beg) at the UNet shapes."""
import torch
from spider_amd import ops
dev = torch.device("cuda:0")
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best
for Lq, Lk, H, d in [(4096, 4096, 8, 40), (1024, 1024, 8, 80), (256, 256, 8, 160), (9216, 9216, 10, 64), (2304, 2304, 20, 64)]:
    C = H * d
    q = torch.randn(2, Lq, C, device=dev).bfloat16(); kv = torch.randn(2, Lk, 2 * C, device=dev).bfloat16()
    zb = torch.zeros(2, dtype=torch.int32, device=dev)
    a = t(lambda: ops.attention(q, kv[..., :C], kv[..., C:], H))
    b = t(lambda: ops.attention(q, kv[..., :C], kv[..., C:], H, kv_beg=zb))
    print(f"Lq={Lq} Lk={Lk} H={H} d={d}: plain {a:7.1f} us   general {b:7.1f} us", flush=True)
