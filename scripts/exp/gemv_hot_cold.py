"""Does the 256 MB Infinity Cache help a decode GEMV? Time the o-proj / qkv / down shaped GEMVs (1) on ONE weight tensor
repeatedly (L2 / MALL-hot) and (2) rotating over 40 tensors (cold, as in the decode loop)."""
import torch
from spider_amd import ops
dev = torch.device("cuda:0")
def t(fs, reps=3):
    for f in fs: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fs: f()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): g.replay()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (reps * len(fs)))
    return best
for name, N, K in [("o", 3584, 3584), ("qkv", 4608, 3584), ("down", 3584, 18944), ("gate_up_half", 18944, 3584)]:
    Ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(40)]
    x = torch.randn(1, K, device=dev).bfloat16()
    out = torch.empty(1, N, device=dev, dtype=torch.bfloat16)
    hot = t([lambda: ops.gemv(Ws[0], x, out=out)] * 40)
    cold = t([(lambda W=W: ops.gemv(W, x, out=out)) for W in Ws])
    mb = N * K * 2 / 1e6
    print(f"{name}: {mb:.1f} MB  hot {hot:6.2f} us ({mb / hot / 1e3:.2f} TB/s)   cold {cold:6.2f} us ({mb / cold / 1e3:.2f} TB/s)", flush=True)
