"""Data-parallel plumbing for the FMC path: one process per GPU (`torchrun`), RCCL through `torch.distributed`
(backend "nccl" on ROCm), gloo for CPU tests.

Inference shards by clip and needs no data-path collective: each rank denoises the clips
`clip_shard(...)` hands it.  The partitioning reproduces `DistributedSampler(num_replicas, rank, shuffle, seed)` as the
trainers use it (`train_cam_obj_ctrl.py:445-452`) so a run is reproducible against the reference's sharding.
`max_over_ranks` / `barrier` implement the timing contract of `bench.py`."""
from __future__ import annotations

import math
from typing import List

import torch
import torch.distributed as dist


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def clip_shard(num_clips: int, rank_: int, world_: int, shuffle: bool = False, seed: int = 0, epoch: int = 0,
               drop_last: bool = False) -> List[int]:
    """Indices of the clips rank `rank_` processes: DistributedSampler semantics (pad by wrapping, stride by world)."""
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        idx = torch.randperm(num_clips, generator=g).tolist()
    else:
        idx = list(range(num_clips))
    if drop_last and num_clips % world_:
        per = math.ceil((num_clips - world_) / world_)
        idx = idx[: per * world_]
    else:
        per = math.ceil(num_clips / world_)
        pad = per * world_ - len(idx)
        if pad:
            idx += (idx * math.ceil(pad / len(idx)))[:pad]
    return idx[rank_: per * world_: world_]


def barrier() -> None:
    if world() > 1:
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    if world() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if world() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
