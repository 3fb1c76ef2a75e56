"""Multi-GPU layer of the batched-episode harness: one process per GPU, episodes sharded by environment id,
NO tensor data ever crosses GPUs; the only collectives are all-reduces of a tiny metrics vector (RCCL over xGMI when the
backend is "nccl"; "gloo" on CPU for the tests).  The reference has no distributed path at all (it raises when
distributed, vlfm/utils/vlfm_trainer.py:65-66)."""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str, device: torch.device | None = None) -> None:
    # VLFM_FORCE_DIST=1 initialises the process group even for a single rank (smoke test of the RCCL path on a 1-GPU box)
    force = os.environ.get("VLFM_FORCE_DIST") == "1" and "MASTER_PORT" in os.environ
    if (world()[2] > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend, **kw)


def shard_env_ids(rank: int, world_size: int, envs_per_rank: int) -> List[int]:
    """Weak scaling: every rank owns ``envs_per_rank`` environments; global env e lives on rank e // envs_per_rank
    (contiguous blocks, so that env ids -- which seed the synthetic episodes -- are disjoint and cover
    range(world_size * envs_per_rank))."""
    return list(range(rank * envs_per_rank, (rank + 1) * envs_per_rank))


def owner_of(env_id: int, envs_per_rank: int) -> int:
    return env_id // envs_per_rank


def reduce_metrics(elapsed_s: float, sums: Sequence[float], device) -> Tuple[float, List[float]]:
    """Job time = MAX over ranks of the local elapsed time; counters (env-steps, frontier counts, checksums ...) = SUM
    over ranks.  Two all-reduces of <= 64 bytes each: latency-bound, algorithm choice immaterial."""
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    v = torch.tensor(list(sums), dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in v.tolist()]


def barrier(device=None) -> None:
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def shutdown() -> None:
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
