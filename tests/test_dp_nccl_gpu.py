"""GPU, 2 ranks over RCCL (backend "nccl"): the one gather of spider_amd/dp.py on device buffers, as bench.py --gpus N runs it.
Skipped unless at least 2 GPUs are visible (the gpurun box has one; the driver's 8-GPU node runs it)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from spider_amd import dp
    r, w, local = dp.init_from_env(backend="nccl")
    dev = torch.device(f"cuda:{local}")
    n_local = 2 - rank                         # ragged: rank 0 sends 2 rows, rank 1 sends 1 (padding exercised)
    toks = (torch.arange(n_local * 6, dtype=torch.int32, device=dev).view(n_local, 6) + 1000 * rank)
    img = torch.full((n_local, 8, 8, 3), 7 + rank, dtype=torch.uint8, device=dev)
    aud = torch.full((n_local, 5), 0.5 + rank, dtype=torch.float32, device=dev)
    g = dp.gather_padded({"tokens": toks, "image": img, "audio": aud}, 2, rank, world, dst=0)
    torch.cuda.synchronize(dev)
    if rank == 0:
        ok = (g["count"].tolist() == [2, 1] and g["tokens"].shape == (2, 2, 6) and int(g["tokens"][1, 0, 0]) == 1000
              and int(g["image"][1, 0, 0, 0, 0]) == 8 and float(g["audio"][1, 0, 0]) == 1.5 and int(g["tokens"][1, 1].abs().sum()) == 0)
        q.put((ok, dist.get_backend(), dist.get_world_size()))
    else:
        assert g is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_over_rccl_two_ranks():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (one process per GPU over RCCL)")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, backend, world = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert ok and backend == "nccl" and world == 2


def _solo_worker(port, q):
    """one rank, backend nccl (= RCCL): the collective calls of bench.run_timed / dp.gather_padded with their dtypes, in a process that
    also launches this library's kernels -- what can be checked of the RCCL path on a one-GPU box"""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from spider_amd import ops
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    x = torch.randn(4, 256, device=dev).bfloat16()
    y0 = ops.rmsnorm(x, torch.ones(256, device=dev).bfloat16(), 1e-6)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    dist.barrier()
    flat = torch.arange(4096, dtype=torch.int32, device=dev).view(torch.uint8)           # the flat byte buffer of gather_padded
    bufs = [torch.empty_like(flat)]
    dist.gather(flat, bufs, dst=0)
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)                            # run_timed's MAX over ranks
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    y1 = ops.rmsnorm(x, torch.ones(256, device=dev).bfloat16(), 1e-6)
    torch.cuda.synchronize(dev)
    ok = bool(torch.equal(bufs[0], flat)) and float(t.item()) == 1.25 and bool(torch.equal(y0, y1))
    q.put((ok, dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_collectives_of_the_bench_path_single_rank():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_solo_worker, args=(port, q))
    p.start()
    ok, backend = q.get(timeout=300)
    p.join(timeout=300)
    assert p.exitcode == 0 and ok and backend == "nccl"
