#!/bin/bash
# round 5 measurements on the GPU box: (1) rocprofv3 --kernel-trace --stats of the default bench command (headline mode), (2) HBM
# traffic of the exact-coordinate kernels from the TCC counters (separate --pmc passes, kernel-trace only), (3) eight ranks' host
# side on this box's host cores.  Every step under `timeout`.
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r5_measure; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tests/native/_build/resample_bench
timeout 400 rocprofv3 --kernel-trace --stats -d $O -o headline --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix > $O/headline_bench.json 2> $O/headline_bench.err
ls $O | head; python - <<PY
import csv, glob
for path in glob.glob("$O/*kernel_stats.csv"):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print(r["Name"][:100], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O -o calib_$c --output-format csv -- $B --cases calib > $O/calib_$c.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O -o tight_$c --output-format csv -- $B --cases perf --reps 2 --case "f32 fill" --path "tight" > $O/tight_$c.log 2>&1
done
python $R/scripts/pmc_summary.py $O calib 2>/dev/null | head -20
python $R/scripts/pmc_summary.py $O lean_exact 2>/dev/null | head -40
cd $R && timeout 600 python scripts/host_stress_ranks.py --ranks 8 --steps 200 --out gpurun_out/r5_measure/host_stress_gpu_box.json 2>&1 | tail -8
