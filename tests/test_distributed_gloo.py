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
