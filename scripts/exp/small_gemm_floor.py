"""Experiment (round 4): what a small-M linear of the UNet's 16^2 / 8^2 levels costs inside a hipGraph on the 64^2 register-staged
kernel without split-K, as a function of K and of whether its weights are cold (128 distinct weight tensors walked round-robin:
420 MB > the 256 MB Infinity Cache, as in a real UNet step) or warm (one tensor replayed). Result (MI355X, DESIGN.md section 5d
item 11): 2.9 - 3.3 us launch floor + 0.25 us (warm) / 0.38 us (cold) per 64-deep K tile; the variant with 2 / 4 groups of waves per
block splitting the K tiles of ONE output tile (measured in the same run, not kept) pays +1.0 / +2.7 us at launch and runs a K tile
in 0.21 / 0.17 us: with 40 - 160 blocks on 256 CUs the K loop is bound by the ~60 - 80 GB/s one CU pulls through its L2 port, not
by the length of the dependent chain."""
import json, os, subprocess, sys
import torch

CASES = [(M, 1280, K) for M in (128, 512) for K in (64, 320, 1280, 2560, 5120)] + [(64, 64, 64)]


def child():
    from spider_amd import ops
    dev = torch.device("cuda:0")
    out = {}
    for M, N, K in CASES:
        A = torch.randn(M, K, device=dev).bfloat16()
        for mode in ("warm", "mall", "cold"):      # mall: 150 MB of weights -- past the 8 x 4 MiB L2s, inside the 256 MB Infinity Cache
            total = {"warm": 0, "mall": 150e6, "cold": 420e6}[mode]
            nw = 1 if mode == "warm" else max(2, min(128, int(total / (N * K * 2)) + 1))
            Ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(nw)]
            n = 128
            for i in range(min(nw, 4)):
                ops.gemm(A, Ws[i])
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(n):
                    ops.gemm(A, Ws[i % nw])
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                g.replay()
            e1.record(); e1.synchronize()
            out[f"{M}x{N}x{K} {mode}"] = e0.elapsed_time(e1) * 1e3 / (3 * n)
            del Ws, g
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    res = {}
    env = dict(os.environ, PYTHONPATH=".", SPIDER_GEMM_TILE="64", SPIDER_GEMM_SPLITS="1")
    o = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print("failed:", o.stderr[-800:]); sys.exit(1)
    res = json.loads(line[0][7:])
    print(f"{'case (64^2 tiles, no split)':32s}  us per launch in a graph of 128")
    for c, v in res.items():
        print(f"{c:32s} {v:8.1f}")
