"""Eight ranks' HOST side on one box, concurrently, without GPUs.

    python scripts/host_stress_ranks.py [--ranks 8] [--steps 300] [--noise-rng reference|philox] [--out profiles/r05_host_stress.json]

What the first real `torchrun --nproc-per-node 8 bench.py --gpus 8` will ask of the host, measured before there is such a
node: N processes (a `gloo` group: init, barrier, the bench's all-gather of counters), each running the Python / marshalling
side of the bench step with the HIP entry points replaced by no-ops (the null engine of scripts/host_null_profile.py: every
`tio_*` call returns at once, tensors live on the host) — and, in the reference noise mode, the REAL plan of the generator's
stream (`tio_host_mt19937_plan`: the mt19937 state chain by jump-ahead on this rank's worker threads, the one piece of host
COMPUTE in a step) for the bench batch's 134 M draws.  Every rank pins itself the way bench.py does (`pin_host_threads`:
its share of the CPUs) and takes its thread budget from `host_thread_budget()`.  Reported per rank: enqueue ms per step
(to be held against the GPU's ~1.4 ms per step) and plan ms per step, alone (1 rank) and under contention (N ranks).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank: int, world: int, steps: int, noise_rng: str, port: int, queue, plan_where: str = "host") -> None:
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    warnings.simplefilter("ignore")
    import ctypes as C

    import bench  # noqa: PLC0415
    import torchio_amd as tio  # noqa: PLC0415
    from torchio_amd import _abi, _lib, distributed as tdist, ops  # noqa: PLC0415
    from torchio_amd.data import _pending  # noqa: PLC0415
    from torchio_amd.transforms import spatial as _sp  # noqa: PLC0415

    dist.init_process_group("gloo", rank=rank, world_size=world)
    info = tdist.rank_info()
    try:
        pinned = tdist.pin_host_threads(info)
    except Exception:  # noqa: BLE001 — no GPU topology here: the even slice is what pin_host_threads falls back to anyway
        pinned = None
    budget = tdist.host_thread_budget()

    class _Null(dict):
        def __missing__(self, key):
            return lambda *a: 0

        def __contains__(self, key):
            return key != "last_error"

    engine = ops.Engine(_Null(), "cpu", "null")
    ops._ENGINE = engine
    _pending.eligible = lambda data: _pending.enabled() and data.dtype == torch.float32 and data.ndim == 5 and not data.requires_grad
    ops.h2d = lambda tensor, device: tensor
    _sp._folding_grid_spacing = lambda extent, mesh: float("inf")
    tio.set_noise_rng("philox")  # (the null engine cannot stand in for the device's draw kernel: the plan is timed below, by itself)
    tio.set_resample_precision("tight")
    transform = bench.build_transform()
    batch = bench.make_batch(16, 8, rank, "cpu")
    torch.manual_seed(100 + rank)
    for _ in range(100):
        transform(batch)

    # the plan of the reference's noise stream for the bench batch (8 x 256^3 draws), through the real host library
    plan_ms = None
    if noise_rng == "reference":
        fn = _lib.load()[1]
        count = 8 * 256**3
        words = int(fn["host_mt19937_plan_words"](count))
        buffer = torch.empty(words, dtype=torch.int32)

    dist.barrier()
    start = time.perf_counter()
    plan_s = 0.0
    for step in range(steps):
        transform(batch)
        if noise_rng == "reference":
            stream = ops.HostNormalStream(1000 * rank + step)
            t0 = time.perf_counter()
            used = C.c_int64(0)
            if plan_where == "device":  # round 6: the device makes the snapshots; the host's share is the 5 KB prefix
                prefix, blocks = C.c_int64(0), C.c_int64(0)
                status = stream._fn["host_mt19937_plan_prefix"](C.addressof(stream._state), count, C.c_void_p(buffer.data_ptr()), words, C.byref(prefix), C.byref(used), C.byref(blocks))
            else:
                status = stream._fn["host_mt19937_plan"](C.addressof(stream._state), count, C.c_void_p(buffer.data_ptr()), words, C.byref(used), stream.threads)
            plan_s += time.perf_counter() - t0
            assert status == _abi.OK, status
    elapsed = time.perf_counter() - start
    dist.barrier()
    counters = torch.tensor([steps * 8.0, elapsed, 0.0], dtype=torch.float64)
    gathered = [torch.zeros_like(counters) for _ in range(world)]
    dist.all_gather(gathered, counters)
    queue.put({
        "rank": rank, "enqueue_ms_per_step": 1e3 * (elapsed - plan_s) / steps, "plan_ms_per_step": 1e3 * plan_s / steps if noise_rng == "reference" else None,
        "threads": budget, "pinned_cpus": len(pinned) if pinned else None,
    })
    dist.destroy_process_group()


def run(world: int, steps: int, noise_rng: str, port: int, plan_where: str = "host") -> list[dict]:
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(rank, world, steps, noise_rng, port, queue, plan_where)) for rank in range(world)]
    for p in procs:
        p.start()
    results = [queue.get(timeout=1800) for _ in procs]
    for p in procs:
        p.join()
    return sorted(results, key=lambda r: r["rank"])


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--ranks", type=int, default=8)
    parser.add_argument("--steps", type=int, default=300)
    parser.add_argument("--noise-rng", choices=["reference", "philox"], default="reference")
    parser.add_argument("--plan", choices=["host", "device"], default="host",
                        help="who runs the mt19937 state chain (ops.set_noise_plan): with `device` only the host's share of the plan (the prefix) is timed")
    parser.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_host_stress.json"))
    args = parser.parse_args()
    report = {"host_cpus": os.cpu_count(), "steps": args.steps, "noise_rng": args.noise_rng, "plan": args.plan,
              "what": "host side of the bench step (null engine) + the real mt19937 plan of 8 x 256^3 draws per step, per rank"}
    report["alone"] = run(1, args.steps, args.noise_rng, 29731, args.plan)
    report[f"{args.ranks}_ranks"] = run(args.ranks, args.steps, args.noise_rng, 29732, args.plan)
    worst = max(r["enqueue_ms_per_step"] for r in report[f"{args.ranks}_ranks"])
    report["summary"] = {
        "enqueue_ms_alone": report["alone"][0]["enqueue_ms_per_step"], f"enqueue_ms_worst_of_{args.ranks}": worst,
        "plan_ms_alone": report["alone"][0]["plan_ms_per_step"],
        f"plan_ms_worst_of_{args.ranks}": max((r["plan_ms_per_step"] or 0.0) for r in report[f"{args.ranks}_ranks"]) or None,
    }
    print(json.dumps(report["summary"], indent=1))
    with open(args.out, "w") as handle:
        json.dump(report, handle, indent=1)


if __name__ == "__main__":
    main()
