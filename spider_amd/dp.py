"""Data-parallel sharding of independent prompts over the GPUs of one node + the final gather.

The reference has no data-parallel inference (its only multi-GPU mode is DeepSpeed ZeRO-2 training,
train_configs/ds_config.json:7-15; spider/common/dist_utils.py:76-91). The path shards naturally because every
prompt / diffusion sample is independent (only samples[...][0] is ever read: spider_decoder.py:311), so:
  * one process per GPU (torch.distributed; backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests)
  * rank r owns prompts {i : i mod world == r}; full model replica per rank (weights << 288 GB HBM)
  * NO collective on the data path; exactly ONE gather of fixed-shape padded outputs to rank 0 per batch --
    each peer sends over its own xGMI link to the root (point-to-point fabric: a ring all-gather would be
    per-link bound and is avoided on purpose).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Returns (rank, world, local_rank). No-op single process when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:     # SPIDER_DIST_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks
            backend = os.environ.get("SPIDER_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Strided ownership i mod world == rank (keeps per-rank counts within 1 of each other)."""
    return list(range(rank, n_items, world))


def order_by_length(lengths: Sequence[int]) -> List[int]:
    """Sort prompts by expected decode length before striding so every rank gets a similar mix
    (variable token counts are the main load-imbalance source)."""
    return sorted(range(len(lengths)), key=lambda i: -lengths[i])


def gather_padded(local: Dict[str, torch.Tensor], counts_max: int, rank: int, world: int, dst: int = 0):
    """One gather of fixed-shape outputs. `local[name]` is [n_local, ...] (n_local <= counts_max); every rank pads to
    counts_max rows and contributes an int32 count. Returns on dst {name: [world, counts_max, ...]}, 'count': [world]}
    and None elsewhere. All fields travel in ONE flat byte buffer -> one collective call."""
    names = sorted(local)
    n_local = next(iter(local.values())).shape[0] if names else 0
    dev = next(iter(local.values())).device if names else torch.device("cpu")
    parts, meta = [], []
    for nme in names:
        t = local[nme]
        pad = torch.zeros((counts_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        pad[:n_local] = t
        b = pad.contiguous().view(torch.uint8).reshape(-1)
        nb = b.numel()
        if nb % 16:   # every field starts 16-byte aligned in the flat buffer (a wider dtype may follow an odd-sized uint8 field)
            b = torch.cat([b, torch.zeros(16 - nb % 16, dtype=torch.uint8, device=dev)])
        meta.append((nme, t.dtype, (counts_max,) + tuple(t.shape[1:]), nb, b.numel()))
        parts.append(b)
    parts.append(torch.tensor([n_local], dtype=torch.int32, device=dev).view(torch.uint8))
    flat = torch.cat(parts)
    if world == 1:
        bufs = [flat]
    else:
        if flat.is_cuda and dist.get_backend() == "gloo":     # gloo gathers host buffers only (rehearsal transport; RCCL takes the
            flat = flat.cpu()                                 # device buffer as it is)
        bufs = [torch.empty_like(flat) for _ in range(world)] if rank == dst else None
        dist.gather(flat, bufs, dst=dst)
        if rank != dst:
            return None
        if bufs[0].device != dev:
            bufs = [b_.to(dev) for b_ in bufs]
    out, off = {}, 0
    for nme, dt, shp, nb, padded in meta:
        out[nme] = torch.stack([b[off:off + nb].view(dt).reshape(shp) for b in bufs])
        off += padded
    out["count"] = torch.stack([b[off:off + 4].view(torch.int32) for b in bufs]).reshape(-1)
    return out


def unshard(gathered: Dict[str, torch.Tensor], n_items: int, world: int) -> Dict[str, torch.Tensor]:
    """Undo the strided sharding: item i lives at [i % world, i // world]."""
    out = {}
    for nme, t in gathered.items():
        if nme == "count":
            continue
        out[nme] = torch.stack([t[i % world, i // world] for i in range(n_items)])
    return out
