"""Round 6: the dma-eligible linears of the UNet's 16^2 / 8^2 levels (to_out / proj_in / proj_out: 1280 <- 1280; ff2: 1280 <- 5120) with COLD
weights (64 distinct weight tensors walked round-robin, > the Infinity Cache for the larger shape), graph of 64 launches, per forced tile
(SPIDER_GEMM_TILE is read once per process: one child per setting). + bias + fp32 residual stream as the engine calls them."""
import json, os, subprocess, sys
import torch

CASES = [(M, 1280, K) for M in (128, 512) for K in (1280, 5120)]
TILES = ["auto", "64", "65", "160", "161"]


def child():
    from spider_amd import ops
    dev = torch.device("cuda:0")
    out = {}
    for M, N, K in CASES:
        A = torch.randn(M, K, device=dev).half()
        nw = 64
        Ws = [(torch.randn(N, K, device=dev) * 0.02).half() for _ in range(nw)]
        b = torch.randn(N, device=dev).half()
        r32 = torch.randn(M, N, device=dev)
        f = lambda i: ops.gemm(A, Ws[i % nw], bias=b, res32=r32, want32=True)
        for i in range(4):
            f(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(64):
                f(i)
        g.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 64)
        out[f"{M}x{N}x{K}"] = sorted(ts)[2]
        del Ws, g
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    res = {}
    for t in TILES:
        env = dict(os.environ, PYTHONPATH=".")
        if t != "auto":
            env["SPIDER_GEMM_TILE"] = t
        o = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
        res[t] = json.loads(line[0][7:]) if line else {"error": (o.stderr or o.stdout)[-300:]}
    print(f"{'shape (cold weights, + bias + res32 + c32d)':46s}" + "".join(f"{('tile ' + t):>12s}" for t in TILES))
    for M, N, K in CASES:
        k = f"{M}x{N}x{K}"
        print(f"{k:46s}" + "".join(f"{res[t].get(k, float('nan')):12.1f}" if "error" not in res[t] else f"{'err':>12s}" for t in TILES))
    for t in TILES:
        if "error" in res[t]:
            print(t, res[t]["error"])
