"""A/B aid: dense self-attention at the long-sequence shapes of the path with the 64-rows-per-wave kernel (SPIDER_ATTN_PIPE2=1,
default) and without it (=0, the 32-rows-per-wave pipelined kernel). The switch is read once per process: the parent runs both
arms as child processes on the same device, twice, interleaved; each child also checks its output against torch fp32 SDPA."""
import json, os, subprocess, sys
import torch

SHAPES = [  # (tag, B, L, H, d)
    ("sd15 32^2", 2, 1024, 8, 80),
    ("sd15 64^2", 2, 4096, 8, 40), ("sdxl 48^2 b8", 8, 2304, 10, 64), ("sdxl 4x48^2", 2, 9216, 10, 64),
    ("sdxl 24^2 b8", 8, 576, 20, 64), ("zeroscope 40x72 b32", 32, 2880, 5, 64), ("sd15 96^2", 2, 9216, 8, 40),
]


def child():
    from spider_amd import ops
    dev = torch.device("cuda:0")
    out = {}
    for dt_name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        for tag, B, L, H, d in SHAPES:
            C = H * d
            qkv = torch.randn(B, L, 3 * C, device=dev).to(dt)
            f = lambda: ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], H)
            o = f()
            q, k, v = [t.float().view(B, L, H, d).transpose(1, 2) for t in (qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:])]
            ref = torch.nn.functional.scaled_dot_product_attention(q[:1], k[:1], v[:1]).transpose(1, 2).reshape(1, L, C)
            err = float((o[:1].float() - ref).abs().max())
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            n = 10
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(n):
                    f()
            g.replay(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); e1.record(); e1.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / n)
            out[f"{tag} {dt_name}"] = (best, err, 4.0 * B * H * L * L * d / best / 1e6)
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    res = {}
    for rnd in range(2):
        for arm in ("1", "0"):
            env = dict(os.environ, PYTHONPATH=".", SPIDER_ATTN_PIPE2=arm)
            o = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print("arm", arm, "failed:", o.stderr[-800:]); continue
            for k, v in json.loads(line[0][7:]).items():
                res.setdefault(k, {}).setdefault(arm, []).append(v)
    for k, v in res.items():
        a = min(x[0] for x in v.get("1", [[0, 0, 0]])); b = min(x[0] for x in v.get("0", [[0, 0, 0]]))
        ea = max(x[1] for x in v.get("1", [[0, 0, 0]])); eb = max(x[1] for x in v.get("0", [[0, 0, 0]]))
        tf = max(x[2] for x in v.get("1", [[0, 0, 0]]))
        print(f"{k:28s} 64-row {a:8.1f} us ({tf:6.1f} TF/s useful, max err {ea:.4f})   32-row {b:8.1f} us (max err {eb:.4f})   {b / max(a, 1e-9):.2f}x", flush=True)
    json.dump(res, open("gpurun_out/attn_pipe2_ab.json", "w"))
