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


def local_world_size() -> int:
    """Ranks that share this host (``LOCAL_WORLD_SIZE`` as set by ``torch.distributed.run``; 1 outside a launcher)."""
    try:
        return max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        return 1


def _usable_cpus() -> list[int]:
    try:
        return sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return list(range(os.cpu_count() or 1))


def _core_groups(cpus: list[int]) -> list[list[int]]:
    """*cpus* grouped by PHYSICAL core (sysfs ``thread_siblings_list``), cores in the order of their first logical CPU.  Without
    sysfs every CPU is its own core."""
    allowed, seen, groups = set(cpus), set(), []
    for cpu in cpus:
        if cpu in seen:
            continue
        siblings = [cpu]
        try:
            with open(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list") as handle:
                text = handle.read().strip()
            found: list[int] = []
            for part in text.split(","):
                lo, _, hi = part.partition("-")
                found.extend(range(int(lo), int(hi or lo) + 1))
            siblings = [c for c in found if c in allowed] or [cpu]
        except (OSError, ValueError):
            pass
        seen.update(siblings)
        groups.append(siblings)
    return groups


def host_thread_budget(cap: int = 32) -> int:
    """Host threads ONE rank may keep busy (the mt19937 jump-ahead workers of the reference-identical noise stream,
    ``csrc/host_rng*.cpp``): the CPUs this process may run on, minus one per rank for its Python enqueue thread, divided by
    the ranks of the host, capped at the 32 beyond which the plan does not get faster (profiles/r03_plan_timing_final.log).
    One rank on a 128-core host: 32.  Eight ranks: (128 - 8) / 8 = 15 each — 120 workers + 8 enqueue threads on 128 cores
    instead of the 256 + 8 the per-process default asked for."""
    ranks = local_world_size()
    usable = _usable_cpus()
    cpus, host = len(usable), os.cpu_count() or 1
    if ranks > 1:
        # Round 5 (scripts/host_stress_ranks.py on the MI355X box's 128-core / 256-thread host): with one worker per LOGICAL CPU
        # of a rank's share the plan of the noise stream took 6.4 - 7.3 ms per step under eight ranks against 1.1 ms alone — two
        # compute-bound workers per physical core, and cores shared between ranks.  Workers are counted in PHYSICAL cores.
        cores = len(_core_groups(usable))
        if cpus >= host:  # not pinned: an even share of the host (pinned: the mask already is this rank's share)
            cores = cores // ranks
        return max(1, min(cap, cores - 1))
    return max(1, min(cap, cpus - 1))


_BUDGET_CACHE: dict = {}


def cached_host_thread_budget() -> int:
    """``host_thread_budget()`` remembered per LOCAL_WORLD_SIZE: the hot path asks once per Noise call, and reading a 256-CPU
    affinity mask costs tens of microseconds.  ``pin_host_threads`` invalidates it; an affinity change made by anybody else
    (``taskset`` on a running process, ``os.sched_setaffinity`` elsewhere) is NOT seen until then — call
    ``invalidate_host_thread_budget()`` after one."""
    key = os.environ.get("LOCAL_WORLD_SIZE", "1")
    value = _BUDGET_CACHE.get(key)
    if value is None:
        value = _BUDGET_CACHE[key] = host_thread_budget()
    return value


def invalidate_host_thread_budget() -> None:
    _BUDGET_CACHE.clear()


def _gpu_local_cpus(index: int) -> list[int] | None:
    """CPUs of the NUMA node GPU *index* hangs off (sysfs ``local_cpulist`` of its PCI function), or None."""
    try:
        props = torch.cuda.get_device_properties(index)
        bdf = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as handle:
            text = handle.read().strip()
    except (AttributeError, OSError, RuntimeError, AssertionError):
        return None
    cpus: list[int] = []
    for part in text.split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus or None


def plan_host_cpus(local_rank: int, n_local: int, usable: list[int], gpu_cpus: list[list[int] | None]) -> list[int]:
    """The CPUs rank *local_rank* of *n_local* should run on: the ranks whose GPUs share a NUMA node split that node's
    CPUs evenly (in rank order); a rank whose GPU reports no node — or an empty share — gets an even slice of *usable*."""
    # (the even slice is made of whole PHYSICAL cores: logical CPUs n and n + cores are siblings on the usual numbering, and a plain
    # slice of the sorted ids gave every core to two ranks)
    cores = _core_groups(usable)
    even = len(cores) // max(n_local, 1)
    fallback = [c for group in cores[local_rank * even : (local_rank + 1) * even] for c in group] if even > 0 else usable
    mine = gpu_cpus[local_rank] if local_rank < len(gpu_cpus) else None
    if not mine:
        return fallback or usable
    node = _core_groups([c for c in mine if c in set(usable)])  # whole physical cores, as above
    sharers = [r for r in range(n_local) if r < len(gpu_cpus) and gpu_cpus[r] == mine]
    share = len(node) // max(len(sharers), 1)
    position = sharers.index(local_rank)
    chosen = [c for group in node[position * share : (position + 1) * share] for c in group]
    return chosen or fallback or usable


def pin_host_threads(info: RankInfo | None = None) -> list[int] | None:
    """Pin this process (and every thread it starts afterwards: the RNG workers, the allocator) to its share of the CPUs of
    its GPU's NUMA node.  No-op for a single rank per host, or when ``TIO_NO_PINNING`` is set.  Returns the CPU list."""
    info = info or rank_info()
    n_local = local_world_size()
    if n_local <= 1 or os.environ.get("TIO_NO_PINNING"):
        return None
    usable = _usable_cpus()
    gpu_cpus = [_gpu_local_cpus(r) for r in range(n_local)] if torch.cuda.is_available() else [None] * n_local
    if any(c is None for c in gpu_cpus):  # a topology that is known for some ranks only: even slices for everybody (no overlap)
        gpu_cpus = [None] * n_local
    cpus = plan_host_cpus(info.local_rank, n_local, usable, gpu_cpus)
    try:
        os.sched_setaffinity(0, cpus)
    except (AttributeError, OSError):
        return None
    _BUDGET_CACHE.clear()  # the share changed
    return cpus


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
