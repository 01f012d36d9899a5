"""CPU, 2 processes over gloo: the data-parallel layer (spider_amd/dp.py) -- strided prompt sharding, ONE gather of the
padded outputs to rank 0, and the un-sharding on the root. The per-rank 'engine' is a stand-in (token ids derived
from the prompt index) so that the test exercises only the N > 1 host logic that bench.py --gpus N runs."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from spider_amd import dp
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    mine = dp.shard_indices(n_items, rank, world)
    per_rank_max = (n_items + world - 1) // world
    toks = torch.stack([torch.arange(6, dtype=torch.int32) + 100 * i for i in mine]) if mine else torch.zeros(0, 6, dtype=torch.int32)
    imgs = torch.stack([torch.full((3, 4, 4), i, dtype=torch.uint8) for i in mine]) if mine else torch.zeros(0, 3, 4, 4, dtype=torch.uint8)
    # the any-to-many payload (scripts/bench_any2many.py): + float32 audio and 5-D uint8 video; the 7-byte "flags" field makes
    # the following int32 field start unaligned unless the flat buffer pads every field
    aud = torch.stack([torch.full((9,), i + 0.5, dtype=torch.float32) for i in mine]) if mine else torch.zeros(0, 9)
    vid = torch.stack([torch.full((2, 3, 5, 3), i, dtype=torch.uint8) for i in mine]) if mine else torch.zeros(0, 2, 3, 5, 3, dtype=torch.uint8)
    flg = torch.stack([torch.full((7,), i, dtype=torch.uint8) for i in mine]) if mine else torch.zeros(0, 7, dtype=torch.uint8)
    g = dp.gather_padded({"tokens": toks, "images": imgs, "audio": aud, "video": vid, "flags": flg}, per_rank_max, rank, world, dst=0)
    if rank == 0:
        assert g["count"].tolist() == [len(dp.shard_indices(n_items, r_, world)) for r_ in range(world)]
        full = dp.unshard(g, n_items, world)
        assert full["audio"][:, 0].tolist() == [i + 0.5 for i in range(n_items)] and full["video"].shape == (n_items, 2, 3, 5, 3)
        assert full["video"][:, 1, 2, 4, 2].tolist() == list(range(n_items)) and full["flags"][:, 6].tolist() == list(range(n_items))
        q.put((full["tokens"][:, 0].tolist(), full["images"][:, 0, 0, 0].tolist()))
    else:
        assert g is None
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_shard_gather_unshard_world2():
    n_items = 5   # ragged: rank 0 gets 3 prompts, rank 1 gets 2 (padding exercised)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    toks, imgs = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert toks == [100 * i for i in range(n_items)] and imgs == list(range(n_items))


def test_single_process_gather_and_length_ordering():
    from spider_amd import dp
    g = dp.gather_padded({"x": torch.ones(2, 3)}, 4, rank=0, world=1)
    assert g["x"].shape == (1, 4, 3) and g["count"].tolist() == [2]
    assert dp.order_by_length([5, 50, 20]) == [1, 2, 0]
    assert dp.shard_indices(10, 3, 8) == [3] and dp.shard_indices(3, 5, 8) == []


def test_bench_self_launch_spawns_one_rank_per_gpu(tmp_path, monkeypatch):
    """`python bench.py --gpus N` outside torchrun starts N rank processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set
    (here: a stand-in rank script that rendezvouses over gloo and all-reduces its rank), and refuses when fewer devices are
    visible than ranks requested."""
    import argparse
    import importlib.util
    import pytest
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    script = tmp_path / "rank.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "dist.init_process_group('gloo', rank=r, world_size=w)\n"
        "t = torch.tensor([r + 1]); dist.all_reduce(t)\n"
        "open(os.path.join(sys.argv[1], f'rank{r}.txt'), 'w').write(str(int(t)))\n"
        "dist.destroy_process_group()\n"
        "sys.exit(3 if (len(sys.argv) > 2 and r == 1) else 0)\n")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 2)
    rc = bench.launch_ranks(argparse.Namespace(gpus=2), script=str(script), argv=[str(tmp_path)])
    assert rc == 0
    assert (tmp_path / "rank0.txt").read_text() == "3" and (tmp_path / "rank1.txt").read_text() == "3"
    assert bench.launch_ranks(argparse.Namespace(gpus=2), script=str(script), argv=[str(tmp_path), "fail"]) == 3   # a failing rank is reported
    with pytest.raises(SystemExit, match="only 2 GPU"):
        bench.launch_ranks(argparse.Namespace(gpus=4), script=str(script), argv=[str(tmp_path)])


def _bench_worker(rank, world, port, workload, q):
    """Two ranks drive bench.run_timed (the contract's timed region) with a stand-in responder over gloo."""
    import argparse
    import importlib.util
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from spider_amd import dp
    dp.init_from_env(backend="gloo")
    B = 3

    class Stub:
        calls = 0

        def respond(self):
            Stub.calls += 1
            base = 10 * rank
            toks = torch.stack([torch.full((5,), base + b, dtype=torch.int32) for b in range(B)])
            if workload == "any2many":      # the payload keys and shapes of AnyToManyResponder.respond (smaller tensors)
                return {"tokens": toks, "image": torch.full((B, 4, 4, 3), rank, dtype=torch.uint8),
                        "audio": torch.full((B, 7), rank + 0.25, dtype=torch.float32), "video": torch.full((B, 2, 3, 3, 3), rank, dtype=torch.uint8)}
            return {"tokens": toks, "out": torch.full((B, 3, 4, 4), rank, dtype=torch.uint8)}      # Responder.respond

    args = argparse.Namespace(workload=workload, batch=B, warmup=1, steps=2)
    dt, g, info = bench.run_timed(Stub(), args, rank, world, torch.device("cpu"))
    assert Stub.calls == 3 and dt > 0 and info == {"world_size": 2, "backend": "gloo"}
    import torch.distributed as dist
    assert not dist.is_initialized(), "run_timed leaves the process group before the rank-0 extras"
    if rank == 0:
        keys = sorted(k for k in g if k != "count")
        full = dp.unshard(g, world * B, world)           # item i came from rank i % world, slot i // world
        q.put((keys, g["count"].tolist(), full["tokens"][:, 0].tolist()))
    else:
        assert g is None


def test_bench_timed_region_two_ranks_both_workloads():
    import pytest  # noqa: F401
    for workload, want in (("text_image", ["out", "tokens"]), ("any2many", ["audio", "image", "tokens", "video"])):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, workload, q)) for r in range(2)]
        for p in procs:
            p.start()
        keys, counts, first = q.get(timeout=180)
        for p in procs:
            p.join(timeout=180)
            assert p.exitcode == 0
        assert keys == want and counts == [3, 3]
        assert first == [0, 10, 1, 11, 2, 12]            # strided un-sharding: rank 0 slot 0, rank 1 slot 0, rank 0 slot 1, ...
