"""Tuning aid: time the GEMM / implicit-GEMM conv at the shapes of the hot path under a forced tile / split-K
(env SPIDER_GEMM_TILE / SPIDER_GEMM_SPLITS are read once per process, so each config runs in a child process)."""
import json, os, subprocess, sys
import torch

SHAPES = [  # (tag, M, N, K, conv_cin or 0, hw)
    ("prefill qkv", 1536, 4608, 3584, 0, 0), ("prefill o", 1536, 3584, 3584, 0, 0),
    ("prefill gate_up", 1536, 37888, 3584, 0, 0), ("prefill down", 1536, 3584, 18944, 0, 0),
    ("u64 qkv", 8192, 960, 320, 0, 0), ("u64 out", 8192, 320, 320, 0, 0), ("u64 ff1", 8192, 2560, 320, 0, 0), ("u64 ff2", 8192, 320, 1280, 0, 0),
    ("u32 qkv", 2048, 1920, 640, 0, 0), ("u32 out", 2048, 640, 640, 0, 0), ("u32 ff1", 2048, 5120, 640, 0, 0), ("u32 ff2", 2048, 640, 2560, 0, 0),
    ("u16 qkv", 512, 3840, 1280, 0, 0), ("u16 out", 512, 1280, 1280, 0, 0), ("u16 ff1", 512, 10240, 1280, 0, 0), ("u16 ff2", 512, 1280, 5120, 0, 0),
    ("u8 qkv", 128, 3840, 1280, 0, 0), ("u8 ff1", 128, 10240, 1280, 0, 0),
    ("c64 320>320", 8192, 320, 2880, 320, 64), ("c64 640>320", 8192, 320, 5760, 640, 64), ("c64 960>320", 8192, 320, 8640, 960, 64),
    ("c32 640>640", 2048, 640, 5760, 640, 32), ("c32 320>640", 2048, 640, 2880, 320, 32), ("c32 1280>640", 2048, 640, 11520, 1280, 32),
    ("c32 1920>640", 2048, 640, 17280, 1920, 32),
    ("c16 1280>1280", 512, 1280, 11520, 1280, 16), ("c16 640>1280", 512, 1280, 5760, 640, 16), ("c16 2560>1280", 512, 1280, 23040, 2560, 16),
    ("c8 1280>1280", 128, 1280, 11520, 1280, 8), ("c8 2560>1280", 128, 1280, 23040, 2560, 8),
]

if os.environ.get("SHAPES") == "sdxl":   # SDXL story step at 768^2, CFG batch 8: 48x48 (C=640) and 24x24 (C=1280) token maps
    SHAPES = [
        ("x48 qkv", 18432, 1920, 640, 0, 0), ("x48 out", 18432, 640, 640, 0, 0), ("x48 ff2", 18432, 640, 2560, 0, 0),
        ("x48 q", 18432, 640, 640, 0, 0),
        ("x24 qkv", 4608, 3840, 1280, 0, 0), ("x24 out", 4608, 1280, 1280, 0, 0), ("x24 ff2", 4608, 1280, 5120, 0, 0),
        ("c96 320>320", 73728, 320, 2880, 320, 96), ("c48 640>640", 18432, 640, 5760, 640, 48), ("c48 320>640", 18432, 640, 2880, 320, 48),
        ("c24 1280>1280", 4608, 1280, 11520, 1280, 24), ("c24 640>1280", 4608, 1280, 5760, 640, 24), ("c48 1280>640", 18432, 640, 11520, 1280, 48),
        ("c24 2560>1280", 4608, 1280, 23040, 2560, 24),
    ]

if os.environ.get("SHAPES") == "deep":   # SD-v1.5 UNet, 16^2 / 8^2 maps (CFG batch 2): the weight-bound small-M problems
    SHAPES = [
        ("c8 1280>1280", 128, 1280, 11520, 1280, 8), ("c8 2560>1280", 128, 1280, 23040, 2560, 8),
        ("c16 1280>1280", 512, 1280, 11520, 1280, 16), ("c16 2560>1280", 512, 1280, 23040, 2560, 16), ("c16 640>1280", 512, 1280, 5760, 640, 16),
        ("u16 out", 512, 1280, 1280, 0, 0), ("u16 ff2", 512, 1280, 5120, 0, 0), ("u16 qkv", 512, 3840, 1280, 0, 0),
        ("u8 out", 128, 1280, 1280, 0, 0), ("u8 ff2", 128, 1280, 5120, 0, 0),
    ]

if os.environ.get("SHAPES") == "b16":   # SD-v1.5 UNet at CFG batch 16 (8 prompts per GPU): 16^2 / 8^2 maps = 4096 / 1024 rows, between the tuned size classes
    SHAPES = [
        ("c16 1280>1280", 4096, 1280, 11520, 1280, 16), ("c16 2560>1280", 4096, 1280, 23040, 2560, 16), ("c16 640>1280", 4096, 1280, 5760, 640, 16),
        ("c8 1280>1280", 1024, 1280, 11520, 1280, 8), ("c8 2560>1280", 1024, 1280, 23040, 2560, 8),
        ("u16 qkv", 4096, 3840, 1280, 0, 0), ("u16 out", 4096, 1280, 1280, 0, 0), ("u16 ff2", 4096, 1280, 5120, 0, 0),
        ("u8 out", 1024, 1280, 1280, 0, 0), ("u8 ff2", 1024, 1280, 5120, 0, 0),
        ("c32 640>640", 16384, 640, 5760, 640, 32), ("c32 1280>640", 16384, 640, 11520, 1280, 32),
        ("u32 out", 16384, 640, 640, 0, 0), ("u32 ff2", 16384, 640, 2560, 0, 0),
    ]

if os.environ.get("SHAPES") == "vae":    # SD-v1.5 VAE decoder, one 512^2 image (up blocks: 64^2 x 512 ... 512^2 x 128)
    SHAPES = [
        ("vae 64 512>512", 4096, 512, 4608, 512, 64), ("vae 128 512>512", 16384, 512, 4608, 512, 128),
        ("vae 256 512>256", 65536, 256, 4608, 512, 256), ("vae 256 256>256", 65536, 256, 2304, 256, 256),
        ("vae 512 256>128", 262144, 128, 2304, 256, 512), ("vae 512 128>128", 262144, 128, 1152, 128, 512),
    ]

if os.environ.get("SHAPES") == "v3d":   # zeroscope UNet3D step, 2 x 16 frames at 40 x 72 (rows = sample, frame, pixel)
    SHAPES = [
        ("v0 out", 92160, 320, 320, 0, 0), ("v0 qkv", 92160, 960, 320, 0, 0), ("v0 ff1", 92160, 2560, 320, 0, 0), ("v0 ff2", 92160, 320, 1280, 0, 0),
        ("v1 out", 23040, 640, 640, 0, 0), ("v1 qkv", 23040, 1920, 640, 0, 0), ("v1 ff2", 23040, 640, 2560, 0, 0),
        ("v2 out", 5760, 1280, 1280, 0, 0), ("v2 qkv", 5760, 3840, 1280, 0, 0), ("v2 ff2", 5760, 1280, 5120, 0, 0),
        ("vc0 320>320", 92160, 320, 2880, 320, (40, 72)), ("vc1 640>640", 23040, 640, 5760, 640, (20, 36)), ("vc2 1280>1280", 5760, 1280, 11520, 1280, (10, 18)),
        ("vt0 320 (3,1,1)", 92160, 320, 960, 320, (16, 2880), "t"), ("vt1 640 (3,1,1)", 23040, 640, 1920, 640, (16, 720), "t"),
        ("vt2 1280 (3,1,1)", 5760, 1280, 3840, 1280, (16, 180), "t"),
        # LayerNorm-folded projections (plain / GEGLU) of the transformer blocks
        ("v0 qkv ln", 92160, 960, 320, 0, 0, "ln"), ("v0 ff1 geglu ln", 92160, 2560, 320, 0, 0, "geglu"),
        ("v1 qkv ln", 23040, 1920, 640, 0, 0, "ln"), ("v1 ff1 geglu ln", 23040, 5120, 640, 0, 0, "geglu"),
        ("x48 qkv ln", 18432, 1920, 640, 0, 0, "ln"), ("x48 ff1 geglu ln", 18432, 5120, 640, 0, 0, "geglu"),
        ("v0x4 qkv ln", 368640, 960, 320, 0, 0, "ln"), ("v0x4 ff1 geglu ln", 368640, 2560, 320, 0, 0, "geglu"),
    ]



def child():
    from spider_amd import ops
    dev = torch.device("cuda:0")
    out = {}
    for tag, M, N, K, cin, hw, *kind in SHAPES:
        if kind and kind[0] in ("ln", "geglu"):
            A = torch.randn(M, K, device=dev).bfloat16()
            W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
            Wf, cs, cb = ops.fold_layernorm(W, torch.ones(K, device=dev).bfloat16(), torch.zeros(K, device=dev).bfloat16(), torch.zeros(N, device=dev).bfloat16())
            act = "geglu" if kind[0] == "geglu" else None
            f = lambda: ops.gemm_ln(A, Wf, cs, cb, act=act, eps=1e-5)
        elif cin and kind:       # temporal conv (3,1,1) over [sample, frame, pixel, C]
            x = torch.randn(M // (hw[0] * hw[1]), hw[0], hw[1], cin, device=dev).bfloat16()
            w = (torch.randn(N, 3, 1, cin, device=dev) * 0.02).bfloat16()
            f = lambda: ops.conv_ex(x, w, pad=(1, 0))
        elif cin:
            h_, w_ = hw if isinstance(hw, tuple) else (hw, hw)
            x = torch.randn(M // (h_ * w_), h_, w_, cin, device=dev).bfloat16()
            w = (torch.randn(N, 3, 3, cin, device=dev) * 0.02).bfloat16()
            f = lambda: ops.conv2d(x, w)
        else:
            A = torch.randn(M, K, device=dev).bfloat16()
            W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
            f = lambda: ops.gemm(A, W)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        # time inside a hipGraph: Python/ctypes launch overhead (~10 us per call) would otherwise hide small kernels
        n = 20
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                f()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record(); e1.synchronize()
        out[tag] = e0.elapsed_time(e1) * 1e3 / (3 * n)
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    res = {}
    cfgs = [(0, 0), (128, 1), (64, 1)] if os.environ.get("QUICK") else [(0, 0), (128, 1), (128, 2), (128, 4), (128, 8), (64, 1), (64, 2), (64, 4), (64, 8), (64, 16)]
    if os.environ.get("GEMM_CFGS"):   # e.g. GEMM_CFGS="0:0,160:1,161:2" (tile 160/161/129 = the LDS-DMA kernel, see gemm.hip)
        cfgs = [tuple(int(v) for v in c.split(":")) for c in os.environ["GEMM_CFGS"].split(",")]
    for cfg in cfgs:
        env = dict(os.environ, PYTHONPATH=".")
        if cfg != (0, 0):
            env["SPIDER_GEMM_TILE"], env["SPIDER_GEMM_SPLITS"] = str(cfg[0]), str(cfg[1])
        o = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print("config", cfg, "failed:", o.stderr[-500:]); continue
        res[cfg] = json.loads(line[0][7:])
    tags = [s[0] for s in SHAPES]
    print(f"{'shape':16s} " + " ".join(f"{str(c):>9s}" for c in res))
    for i, t in enumerate(tags):
        M, N, K = SHAPES[i][1:4]
        row = [res[c][t] for c in res]
        best = min(row)
        print(f"{t:16s} " + " ".join(f"{v:9.1f}" for v in row) + f"   best {list(res)[row.index(best)]} {2*M*N*K/best/1e6:.0f} TF/s")
