#!/bin/bash
# round 5: rocprofv3 --kernel-trace --stats of the bench's headline mode (the kernel durations roofline.achieved is checked against)
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r5_rocprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O -o headline --output-format csv -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix > $O/headline_bench.json 2> $O/headline_bench.err
python - <<PY
import csv, glob
for path in glob.glob("$O/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print(r["Name"][:100], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
tail -c 600 $O/headline_bench.json
