"""The REAL reference timed on this container's host cores (SURVEY.md 8(d) CPU-baseline protocol).

    PYTHONDONTWRITEBYTECODE=1 python scripts/time_reference_cpu.py [--size 256] [--out profiles/r05_reference_cpu_timing.json]

Imports TorchIO 2.0.0a2 from /root/reference unmodified (tests/golden/ref_import.py stubs its uninstalled non-hot-path
imports), builds the metric's Compose with the explicit ranges of SURVEY 8(d) on ONE synthetic 1 x S^3 float32 Subject and
times, with time.perf_counter, (a) `transform(subject)` — deepcopy + wrap included, what a user sees — and (b)
`apply_transform` only on recorded parameters; min and median of >= 3 runs after one warm-up, at every core of the box and at
ONE thread.  /root/reference does not exist on the GPU box: this recording is made in the build container (8 cores here, not
the MI355X host's 128) and committed; bench.py carries it as a RECORDED leg next to the legs it measures live.
"""
from __future__ import annotations

import argparse, json, os, platform, statistics, sys, time, warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from ref_import import import_reference  # noqa: E402

tio = import_reference()
warnings.simplefilter("ignore")


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--size", type=int, default=256)
    parser.add_argument("--runs", type=int, default=3)
    parser.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_reference_cpu_timing.json"))
    args = parser.parse_args()
    size = args.size
    torch.manual_seed(0)
    subject = tio.Subject(t1=tio.ScalarImage(torch.rand(1, size, size, size, dtype=torch.float32)))
    transform = tio.Compose([
        tio.Affine(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5)),
        tio.ElasticDeformation(),
        tio.BiasField(),
        tio.Blur(std=(0.5, 2)),
        tio.Noise(),
    ])
    cores = os.cpu_count() or 1
    report = {
        "what": "the unmodified reference (TorchIO 2.0.0a2, /root/reference/src, stub-imported) on the BUILD CONTAINER's host cores",
        "pipeline": "Compose[Affine(+-10 deg, 0.9-1.1, +-5 mm), ElasticDeformation(), BiasField(), Blur(0.5-2), Noise()]",
        "volume": f"1x{size}^3 float32 (torch.rand, seed 0)", "torch": torch.__version__, "host_cpus": cores, "machine": platform.processor() or platform.machine(),
        "legs": {},
    }
    for threads in sorted({cores, 1}, reverse=True):
        torch.set_num_threads(threads)
        torch.manual_seed(1)
        out = transform(subject)  # warm-up
        history = out.applied_transforms
        whole, apply_only = [], []
        for run in range(args.runs):
            torch.manual_seed(2 + run)
            start = time.perf_counter()
            out = transform(subject)
            whole.append(time.perf_counter() - start)
        # apply_transform only: the children on the recorded parameters, no envelope (no deepcopy, no parameter sampling)
        for run in range(args.runs):
            batch = None
            data = tio.SubjectsBatch.from_subjects([subject]) if hasattr(tio, "SubjectsBatch") else None
            start = time.perf_counter()
            current = data
            for child, record in zip(transform.transforms, history):
                current = child.apply_transform(current, record.params)
            apply_only.append(time.perf_counter() - start)
        report["legs"][str(threads)] = {
            "threads": threads,
            "transform(subject)_s": {"min": min(whole), "median": statistics.median(whole), "runs": whole},
            "apply_transform_only_s": {"min": min(apply_only), "median": statistics.median(apply_only), "runs": apply_only},
            "volumes_per_s": 1.0 / min(whole),
        }
        print(threads, report["legs"][str(threads)], flush=True)
    with open(args.out, "w") as handle:
        json.dump(report, handle, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
