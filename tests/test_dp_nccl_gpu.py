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
