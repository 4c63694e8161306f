"""Multi-GPU plumbing: volumes shard as an embarrassingly-parallel batch split.

Every volume (batch element) is augmented independently (per-instance parameters,
reference transform.py:300-328), so the hot path has NO data-path collective:
one process per GPU takes a contiguous slice of the global batch, and the only
exchange is an ``all_gather`` of three float64 counters per rank
(``[n_volumes, elapsed_s, algorithmic_bytes]`` — RCCL over xGMI on MI355X, gloo
in the CPU tests).  The reference has no distributed code at all; its feeding
side would use ``Queue(subject_sampler=DistributedSampler)`` (queue.py:48-50).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class RankInfo:
    rank: int
    local_rank: int
    world_size: int


def rank_info() -> RankInfo:
    """Read RANK / LOCAL_RANK / WORLD_SIZE as set by ``torch.distributed.run``."""
    return RankInfo(
        rank=int(os.environ.get("RANK", "0")),
        local_rank=int(os.environ.get("LOCAL_RANK", "0")),
        world_size=int(os.environ.get("WORLD_SIZE", "1")),
    )


def init_process_group(backend: str | None = None) -> RankInfo:
    """Initialise the default group when WORLD_SIZE > 1 (``nccl`` == RCCL on ROCm, ``gloo`` on CPU)."""
    info = rank_info()
    # under a launcher (torch.distributed.run sets RANK / MASTER_ADDR) the group is created even for one rank, so that
    # `--nproc-per-node 1` walks the same RCCL path (communicator, barrier, all-gather) as N ranks do
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if (info.world_size > 1 or launched) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
        if backend == "nccl":
            # bind the rank to its GPU before RCCL creates the communicator, so that barriers and
            # collectives never have to guess the device
            torch.cuda.set_device(info.local_rank)
            dist.init_process_group(
                backend=backend, rank=info.rank, world_size=info.world_size,
                device_id=torch.device("cuda", info.local_rank),
            )
        else:
            dist.init_process_group(backend=backend, rank=info.rank, world_size=info.world_size)
    return info


def shard_range(n_items: int, rank: int, world_size: int) -> range:
    """Contiguous slice of ``range(n_items)`` owned by *rank* (sizes differ by at most one)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} not in [0, {world_size})")
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def gather_counters(n_volumes: float, elapsed_s: float, algorithmic_bytes: float, device=None) -> torch.Tensor:
    """All-gather ``[n_volumes, elapsed_s, bytes]`` from every rank → ``(world, 3)`` float64 on the CPU."""
    local = torch.tensor([n_volumes, elapsed_s, algorithmic_bytes], dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local.cpu()[None]
    gathered = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, local)
    return torch.stack(gathered).cpu()


def aggregate_throughput(counters: torch.Tensor) -> dict:
    """Whole-job numbers from gathered counters: total volumes over the SLOWEST rank's time."""
    volumes = float(counters[:, 0].sum())
    slowest = float(counters[:, 1].max())
    return {
        "volumes": volumes,
        "elapsed_s": slowest,
        "volumes_per_s": volumes / slowest if slowest > 0 else 0.0,
        "algorithmic_bytes": float(counters[:, 2].sum()),
    }


def barrier() -> None:
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
