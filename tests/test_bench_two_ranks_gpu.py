"""GPU: the whole `bench.py --gpus 2` path with real responders -- two rank processes, each with its own engines, two-stream
schedule, barrier + max-over-ranks timing, ONE gather of the padded outputs to rank 0 -- rehearsed on ONE GPU: both ranks compute on
cuda:0 (SPIDER_SHARE_GPU=1) and the collective runs over gloo (SPIDER_DIST_BACKEND=gloo) because RCCL refuses two ranks on one
device. The RCCL transport itself is covered by tests/test_dp_nccl_gpu.py where two GPUs are visible."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_bench_two_ranks_share_one_gpu():
    env = dict(os.environ, SPIDER_SHARE_GPU="1", SPIDER_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--headline-only",
           "--prompt-len", "192", "--new-tokens", "12", "--denoise-steps", "5", "--prompt-len-jitter", "48"]      # ragged prompts: left-padded
    o = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=560, cwd=ROOT)
    assert o.returncode == 0, o.stderr[-2000:]
    lines = [l for l in o.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["dist"] == {"world_size": 2, "backend": "gloo"}
    assert "order_by_length" in d["config"]["prompt_sharding"] and d["config"]["product_class"] == "spider_amd.SpiderFreeInfer"
    assert d["gathered"]["tokens"] == [2, 1, 12] and d["gathered"]["out"] == [2, 1, 3, 512, 512] and d["gathered"]["count"] == [2]
    assert d["value"] > 0 and abs(d["value"] - 2 * 1 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-2 * d["value"]   # all ranks' responses / max time


@pytest.mark.timeout(600)
def test_bench_two_ranks_under_torchrun():
    """the driver's launch form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher), same one-GPU rehearsal as above"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SPIDER_SHARE_GPU="1", SPIDER_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--headline-only",
           "--prompt-len", "192", "--new-tokens", "12", "--denoise-steps", "5"]
    o = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=560, cwd=ROOT)
    assert o.returncode == 0, o.stderr[-2000:]
    lines = [l for l in o.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dist"] == {"world_size": 2, "backend": "gloo"} and d["gathered"]["tokens"] == [2, 1, 12]
    assert d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "dp2"


@pytest.mark.timeout(900)
def test_bench_two_ranks_any2many_share_one_gpu():
    """BASELINE configs[4] with 2 real ranks (one GPU, gloo): every rank answers its 2 prompts with text + image + audio + video through
    SpiderFreeInfer -> SpiderDecoder.generate_batch, then the ONE gather of tokens / images / audio / video to rank 0."""
    env = dict(os.environ, SPIDER_SHARE_GPU="1", SPIDER_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "any2many", "--batch", "2",
           "--prompt-len", "96", "--new-tokens", "8", "--denoise-steps", "2", "--prompt-len-jitter", "16"]
    o = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=860, cwd=ROOT)
    assert o.returncode == 0, o.stderr[-2000:]
    lines = [l for l in o.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dist"] == {"world_size": 2, "backend": "gloo"} and d["config"]["prompts_per_gpu"] == 2
    g = d["gathered"]
    assert g["tokens"] == [2, 2, 8] and g["image"] == [2, 2, 3, 512, 512] and g["audio"] == [2, 2, 80000] and g["video"] == [2, 2, 16, 320, 576, 3]
    assert g["count"] == [2] and d["value"] > 0
    assert set(d["rank0_stage_ms_last_step"]) >= {"image_decoder_ms", "audio_decoder_ms", "video_decoder_ms"}
