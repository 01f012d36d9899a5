"""Round 5: the zeroscope UNet3D's top-level linears (92160 token rows) per forced tile (SPIDER_GEMM_TILE is read once per process:
run once per setting). Graph of 10 launches, median of 5; TFLOP/s and the HBM bytes the call must move at least."""
import os, torch
from spider_amd import ops
dev = torch.device("cuda:0")
tile = os.environ.get("SPIDER_GEMM_TILE", "auto")
M = int(os.environ.get("LIN_M", "92160"))
g = torch.Generator(device=dev).manual_seed(0)
EXTRA = os.environ.get("LIN_EXTRA", "0") != "0"
cases = [("attn out / proj 320->320 + res32 + c32d", 320, 320, "res32"), ("plain 320->320", 320, 320, None), ("plain 320->960 (qkv without the LN fold)", 960, 320, None),
         ("ln qkv 320->960", 960, 320, "ln"), ("ff out 1280->320 + res32 + c32d", 320, 1280, "res32"), ("ln geglu 320->2x1280", 2560, 320, "geglu"),
         ("640: attn out 640->640 + res32", 640, 640, "res32q"), ("640: ln geglu 640->2x2560", 5120, 640, "geglu4"), ("640: ln qkv 640->1920", 1920, 640, "lnq")]
if EXTRA:      # (rows, N, K, fp32 streams): the other residual / plain linears of the video and batched image UNets
    cases = [(f"{K}->{N} rows {r}" + (" + res32 + c32d" if f else ""), N, K, ("res32x" if f else "plainx"), r)
             for r, N, K, f in ((92160, 320, 1280, False), (92160, 320, 640, True), (23040, 640, 640, False), (23040, 640, 2560, True),
                                (23040, 640, 1280, True), (5760, 1280, 1280, True), (5760, 1280, 5120, True), (65536, 320, 320, True),
                                (65536, 320, 1280, True), (16384, 640, 640, True), (16384, 640, 2560, True), (8192, 320, 320, True), (8192, 320, 1280, True))]
ONLY = os.environ.get("LIN_ONLY")          # substring of a case name: run only the cases that contain it (counter passes)
for case in cases:
    if ONLY and ONLY not in case[0]:
        continue
    name, N, K, kind = case[:4]
    m = M // 4 if kind in ("res32q", "geglu4", "lnq") else (case[4] if len(case) > 4 else M)
    if kind == "lnq": kind = "ln"
    if kind == "res32x": kind = "res32"
    if kind == "plainx": kind = None
    A = torch.randn(m, K, device=dev, generator=g).half()
    W = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).half()
    b = torch.randn(N, device=dev, generator=g).half()
    if kind in ("res32", "res32q"):
        r32 = torch.randn(m, N, device=dev, generator=g)
        f = lambda: ops.gemm(A, W, bias=b, res32=r32, want32=True)
        byt = m * K * 2 + m * N * (2 + 4 + 4)
    elif kind == "ln":
        fold = ops.fold_layernorm(W, torch.ones(K, device=dev).half(), torch.zeros(K, device=dev).half())
        f = lambda: ops.gemm_ln(A, *fold)
        byt = m * K * 2 + m * N * 2
    elif kind in ("geglu", "geglu4"):
        fold = ops.fold_layernorm(W, torch.ones(K, device=dev).half(), torch.zeros(K, device=dev).half(), b)
        f = lambda: ops.gemm_ln(A, *fold, act=os.environ.get("GEGLU_ACT", "geglu"))
        byt = m * K * 2 + m * (N // 2) * 2
    else:
        f = lambda: ops.gemm(A, W, bias=b)
        byt = m * K * 2 + m * N * 2
    out = f(); torch.cuda.synchronize()
    if kind in ("res32", "res32q", None):       # (check of the forced kernel against torch on the same inputs; test infrastructure only)
        ref = A.float() @ W.float().T + b.float() + (r32 if kind else 0)
        got = out[1] if kind else out.float()
        rel = float((got - ref).norm() / ref.norm())
        assert rel < (2e-5 if kind else 1e-3), (name, rel)
    if kind == "ln":
        ref = torch.nn.functional.layer_norm(A.float(), (K,)) @ W.float().T
        rel = float((out.float() - ref).norm() / ref.norm())
        assert rel < 2e-3, (name, rel)
    if kind in ("geglu", "geglu4"):
        y = torch.nn.functional.layer_norm(A.float(), (K,)) @ W.float().T + b.float()
        v, gt = y.chunk(2, -1)
        ref = v * torch.nn.functional.gelu(gt)
        rel = float((out.float() - ref).norm() / ref.norm())
        assert rel < 4e-3, (name, rel)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            f()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 10)
    us = sorted(ts)[2]
    print(f"tile {tile:5s} {name:44s} rows {m:6d}: {us:8.1f} us  {2.0 * m * N * K / us / 1e6:7.1f} TFLOP/s  min HBM {byt / 1e6:7.1f} MB = {byt / us / 1e6:5.2f} TB/s", flush=True)
