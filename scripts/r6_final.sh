#!/bin/bash
# Round 6, final tree: smoke, the whole GPU suite, the native harness's parity cases, the default bench line, rocprofv3 kernel
# statistics of the default bench command (headline) and of config 5's call.  Every step under `timeout`.
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6_final; mkdir -p $O
cd $R
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/gpu_tests.txt
(cd tests/native/_build && timeout 200 ./resample_bench --cases parity 2>&1 | tail -1) | tee $O/native_parity.txt
timeout 200 tests/native/_build/resample_bench --cases perf --reps 20 --path tight 2>&1 | grep -E " tight " | tee $O/native_perf_tight.txt
timeout 200 tests/native/_build/resample_bench --cases perf --reps 20 --path lean-exact 2>&1 | grep -E " lean-exact " | tee $O/native_perf_exact.txt
timeout 480 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 400 $O/bench_n1.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O -o headline --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix > $O/headline_bench.json 2> $O/headline_bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o config5 --output-format csv -- python $R/scripts/r5_config5_kernels.py > $O/config5_run.log 2>&1
python - <<PY
import csv, glob
for name in ("headline", "config5"):
    print("==", name)
    for path in glob.glob("$O/%s_kernel_stats.csv" % name):
        rows = sorted(csv.DictReader(open(path)), key=lambda r: -float(r["TotalDurationNs"]))
        for r in rows[:12]:
            print(r["Name"][:110], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
