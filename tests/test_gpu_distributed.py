"""GPU: the multi-process path on real hardware.

* ``bench.py`` under ``torch.distributed.run --nproc-per-node 1``: the RCCL (``nccl``) process group is created,
  the barrier and the counter all-gather run on the MI355X — the N-rank code path with N = 1, on every box;
* two ranks with ``backend="nccl"`` on two GPUs when the box has them (skipped on the 1-GPU test boxes): each rank
  augments its own shard, the gathered counters describe both.

An 8-GPU scaling curve is the driver's to measure (SCALE_rNN.json); nothing here claims one.
"""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(nproc: int, extra: list[str]) -> dict:
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1",
        "--size", "128", "--batch", "2", "--prewarm", "2", "--no-cpu-baseline", "--no-aten-baseline", "--no-mode-matrix", *extra,
    ]
    done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [line for line in done.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, done.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_under_torchrun_one_rank_uses_rccl(hip):
    line = _run_bench(1, [])
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["value"] > 0
    assert line["config"]["global_batch"] == 2 and line["distributed"]["backend"] == "nccl"
    assert line["distributed"]["world_size"] == 1 and line["distributed"]["counters_shape"] == [1, 3]
    assert len(json.dumps(line)) < 6000  # (the line must survive the driver's 8 KB tail)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on one node")
def test_bench_under_torchrun_two_ranks(hip):
    line = _run_bench(2, [])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4
    assert line["distributed"]["world_size"] == 2 and line["distributed"]["counters_shape"] == [2, 3]
    assert line["scaling"] == "weak" and line["value"] > 0
