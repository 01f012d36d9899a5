"""Tuning aid: 128^2 vs 64^2 GEMM tiles as a function of K for large M x N problems (tile forced through SPIDER_GEMM_TILE,
read once per process -> one child process per tile)."""
import json, os, subprocess, sys
import torch

SHAPES = [(M, N, K) for (M, N) in ((8192, 2560), (8192, 960), (2048, 5120), (4608, 10240), (18432, 5120)) for K in (320, 640, 960, 1280, 1920, 2560)]


def child():
    from spider_amd import ops
    dev = torch.device("cuda:0")
    out = {}
    for M, N, K in SHAPES:
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        f = lambda: ops.gemm(A, W)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                f()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record(); e1.synchronize()
        out[f"{M}x{N}x{K}"] = e0.elapsed_time(e1) * 1e3 / 30
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    res = {}
    for tile in (128, 64):
        env = dict(os.environ, PYTHONPATH=".", SPIDER_GEMM_TILE=str(tile), SPIDER_GEMM_SPLITS="1")
        o = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        res[tile] = json.loads([l for l in o.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    for k in res[128]:
        M, N, K = map(int, k.split("x"))
        a, b = res[128][k], res[64][k]
        print(f"{k:20s} 128^2 {a:7.1f} us ({2*M*N*K/a/1e6:5.0f} TF/s)   64^2 {b:7.1f} us ({2*M*N*K/b/1e6:5.0f} TF/s)   {'64' if b < a else '128'}")
