#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r5_rocprof_config5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o config5 --output-format csv -- python $R/scripts/r5_config5_kernels.py > $O/run.log 2>&1
python - <<PY
import csv, glob
for path in glob.glob("$O/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:10]:
        print(r["Name"][:110], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
