"""The N>1 path on CPU: world_size-2 gloo processes, batch split, counter all-gather.

Each rank augments its contiguous slice of the global batch (per-subject seeds, so
the result does not depend on the world size) and the ranks exchange ONLY the three
throughput counters.  The union of the shards must equal the single-process run.
"""
from __future__ import annotations

import os
import socket
import sys
import warnings

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SUBJECTS = 5
SIZE = 8


def _augment(index: int) -> torch.Tensor:
    import torchio_amd as tio
    from parity_harness import make_subjects

    subject = make_subjects(SIZE, 1, seed=100 + index, with_label=False)[0]
    transform = tio.Compose([tio.Affine(degrees=(-10, 10)), tio.BiasField(), tio.Noise(std=0.1)])
    torch.manual_seed(1000 + index)
    return transform(subject).t1.data


def _worker(rank: int, world_size: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    warnings.simplefilter("ignore")
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle.oracle import oracle_engine
    from parity_harness import use_engine
    from torchio_amd import distributed as tdist

    info = tdist.init_process_group("gloo")
    assert (info.rank, info.world_size) == (rank, world_size)
    mine = tdist.shard_range(N_SUBJECTS, info.rank, info.world_size)
    with use_engine(oracle_engine()):
        outputs = {index: _augment(index) for index in mine}
    tdist.barrier()
    counters = tdist.gather_counters(len(mine), 0.5 + rank, 123.0 * len(mine))
    total = tdist.aggregate_throughput(counters)
    assert counters.shape == (world_size, 3)
    assert total["volumes"] == N_SUBJECTS and total["elapsed_s"] == 0.5 + (world_size - 1)
    torch.save({"outputs": outputs, "total": total}, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_range_partitions_everything():
    from torchio_amd.distributed import shard_range

    for n, world in [(64, 8), (5, 2), (3, 4), (0, 2), (7, 7)]:
        seen = [i for rank in range(world) for i in shard_range(n, rank, world)]
        assert seen == list(range(n))
        sizes = [len(shard_range(n, rank, world)) for rank in range(world)]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


def test_single_process_gather_is_a_no_op():
    from torchio_amd import distributed as tdist

    counters = tdist.gather_counters(8, 2.0, 1e9)
    assert counters.shape == (1, 3)
    assert tdist.aggregate_throughput(counters)["volumes_per_s"] == 4.0


def test_two_gloo_ranks_reproduce_the_single_process_result(tmp_path):
    world_size = 2
    mp.spawn(_worker, args=(world_size, _free_port(), str(tmp_path)), nprocs=world_size, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.oracle import oracle_engine
    from parity_harness import use_engine

    merged = {}
    for rank in range(world_size):
        payload = torch.load(tmp_path / f"rank{rank}.pt", weights_only=True)
        assert not set(payload["outputs"]) & set(merged)
        merged.update(payload["outputs"])
        assert payload["total"]["volumes_per_s"] == N_SUBJECTS / 1.5
    assert sorted(merged) == list(range(N_SUBJECTS))
    with use_engine(oracle_engine()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for index in range(N_SUBJECTS):
            assert torch.equal(merged[index], _augment(index))


# ---- config 4 (BASELINE.json): a batch of 64 volumes as 8 ranks x 8, per-instance parameters ---------------------------
# the only thing that reduces across a batch is the default fill value, which the reference takes from the
# FIRST element of the (rank-local) batch (spatial.py:2054-2055) — so a rank's batch of 8 must use ITS element 0, and the
# union of the 8 rank-local results must equal what one process computes shard by shard.
CONFIG4_GLOBAL, CONFIG4_WORLD, CONFIG4_SIZE = 64, 8, 8


def _config4_volume(index: int) -> torch.Tensor:
    """Volume `index` of the global batch: U[0, 1) noise with ONE voxel at -(index + 1): its minimum names it."""
    data = torch.rand(1, CONFIG4_SIZE, CONFIG4_SIZE, CONFIG4_SIZE, generator=torch.Generator().manual_seed(500 + index))
    data[0, CONFIG4_SIZE // 2, CONFIG4_SIZE // 2, CONFIG4_SIZE // 2] = -float(index + 1)
    return data


def _config4_shard(rank: int, world_size: int) -> dict:
    import torchio_amd as tio
    from torchio_amd.distributed import shard_range

    mine = shard_range(CONFIG4_GLOBAL, rank, world_size)
    data = torch.stack([_config4_volume(index) for index in mine])
    batch = tio.SubjectsBatch({"t1": tio.ImagesBatch(data, [tio.AffineMatrix() for _ in mine], image_class=tio.ScalarImage)})
    # a shift of two voxels along every axis (part of the field of view leaves the volume: the fill value shows), a per-element
    # rotation and scale, and per-element noise: per-instance parameters drawn from the rank's own seed
    transform = tio.Compose([
        tio.Affine(degrees=(-8, 8), scales=(0.95, 1.05), translation=(2, 2)),
        tio.Noise(std=(0.01, 0.02)),
    ])
    torch.manual_seed(4321 + rank)
    out = transform(batch)
    return {"indices": list(mine), "data": out.t1.data.clone()}


def _config4_worker(rank: int, world_size: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    warnings.simplefilter("ignore")
    torch.set_num_threads(1)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), LOCAL_WORLD_SIZE=str(world_size),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    from oracle.oracle import oracle_engine
    from parity_harness import use_engine
    from torchio_amd import distributed as tdist

    info = tdist.init_process_group("gloo")
    budget = tdist.host_thread_budget()
    with use_engine(oracle_engine()):
        shard = _config4_shard(info.rank, info.world_size)
    tdist.barrier()
    counters = tdist.gather_counters(len(shard["indices"]), 1.0 + 0.1 * rank, 10.0 * len(shard["indices"]))
    shard["counters"] = counters
    shard["thread_budget"] = budget
    torch.save(shard, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_config4_partition_over_eight_gloo_ranks(tmp_path):
    """64 volumes -> 8 ranks x 8: every volume processed exactly once, each rank-local result equal to the single-process
    computation of that shard, every rank's fill value taken from ITS OWN element 0, one (8, 3) counter table on every rank."""
    world = CONFIG4_WORLD
    mp.spawn(_config4_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.oracle import oracle_engine
    from parity_harness import use_engine

    seen: list[int] = []
    host_cpus = os.cpu_count() or 1
    for rank in range(world):
        payload = torch.load(tmp_path / f"rank{rank}.pt", weights_only=True)
        indices, data = payload["indices"], payload["data"]
        assert len(indices) == CONFIG4_GLOBAL // world and data.shape[0] == len(indices)
        seen += indices
        # the same shard in this process: bit-identical (nothing of the result depends on the other ranks)
        with use_engine(oracle_engine()), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            alone = _config4_shard(rank, world)
        assert torch.equal(alone["data"], data), f"rank {rank}: the shard computed alone differs"
        # the fill value: the minimum of the rank's LOCAL element 0, i.e. -(first global index + 1), in every element of the shard
        fill = -float(indices[0] + 1)
        noise_free = tio_affine_only(rank, world)
        for element in range(len(indices)):
            filled = int((noise_free[element] == fill).sum())
            assert filled >= CONFIG4_SIZE * CONFIG4_SIZE, f"rank {rank} element {element}: {filled} voxels carry the local fill {fill}"
            foreign = [-(g + 1.0) for g in range(CONFIG4_GLOBAL) if g not in indices]
            assert not torch.isin(noise_free[element], torch.tensor(foreign)).any(), "a fill value from another rank's batch"
        counters = payload["counters"]
        assert counters.shape == (world, 3)
        assert counters[:, 0].tolist() == [CONFIG4_GLOBAL // world] * world
        assert counters[:, 1].tolist() == pytest.approx([1.0 + 0.1 * r for r in range(world)])
        # the per-rank host budget: the ranks together never ask for more threads than the host has CPUs
        assert 1 <= payload["thread_budget"] <= max(1, host_cpus // world)
    assert sorted(seen) == list(range(CONFIG4_GLOBAL))


def tio_affine_only(rank: int, world_size: int) -> torch.Tensor:
    """The shard of `rank` through the Affine alone (same seed, hence the same per-element draws): no noise on the fill."""
    import torchio_amd as tio
    from oracle.oracle import oracle_engine
    from parity_harness import use_engine
    from torchio_amd.distributed import shard_range

    mine = shard_range(CONFIG4_GLOBAL, rank, world_size)
    data = torch.stack([_config4_volume(index) for index in mine])
    batch = tio.SubjectsBatch({"t1": tio.ImagesBatch(data, [tio.AffineMatrix() for _ in mine], image_class=tio.ScalarImage)})
    torch.manual_seed(4321 + rank)
    with use_engine(oracle_engine()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return tio.Affine(degrees=(-8, 8), scales=(0.95, 1.05), translation=(2, 2))(batch).t1.data


def test_host_cpu_plan_splits_numa_nodes_between_their_ranks():
    from torchio_amd.distributed import plan_host_cpus

    usable = list(range(128))
    node0, node1 = list(range(0, 64)), list(range(64, 128))
    gpu_cpus = [node0] * 4 + [node1] * 4
    plans = [plan_host_cpus(r, 8, usable, gpu_cpus) for r in range(8)]
    assert all(len(p) == 16 for p in plans)
    assert sorted(c for p in plans for c in p) == usable  # disjoint, complete
    assert set(plans[0]) <= set(node0) and set(plans[7]) <= set(node1)
    # no topology information: even slices
    plans = [plan_host_cpus(r, 8, usable, [None] * 8) for r in range(8)]
    assert [p[0] for p in plans] == [16 * r for r in range(8)] and all(len(p) == 16 for p in plans)
    # a mask smaller than the rank count still gives every rank something to run on
    assert plan_host_cpus(3, 8, [0, 1], [None] * 8) == [0, 1]


def test_host_thread_budget_divides_by_the_ranks_of_the_host(monkeypatch):
    from torchio_amd import distributed as tdist

    monkeypatch.setattr(tdist, "_usable_cpus", lambda: list(range(128)))
    monkeypatch.setattr(tdist.os, "cpu_count", lambda: 128)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    assert tdist.host_thread_budget() == 32
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert tdist.host_thread_budget() == 15  # 8 x (15 workers + 1 enqueue thread) = 128
    monkeypatch.setattr(tdist, "_usable_cpus", lambda: list(range(16)))  # pinned to its share already
    assert tdist.host_thread_budget() == 15
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    monkeypatch.setattr(tdist, "_usable_cpus", lambda: list(range(128)))
    assert tdist.host_thread_budget() == 32  # capped
