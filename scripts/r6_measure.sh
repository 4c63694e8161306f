#!/bin/bash
# round 6, final tree: what the bench line's `roofline` is checked against —
#   (1) rocprofv3 --kernel-trace --stats of the headline bench command and of a config-5 call,
#   (2) SQ instruction counters of the exact-coordinate kernel (tight), affine and elastic launch: vector instructions per voxel,
#   (3) HBM traffic of the same launches from the TCC counters (separate --pmc passes, kernel trace only; calibrated).
# Everything lands in gpurun_out/r6_measure/; copy the summaries into profiles/.
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6_measure; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tests/native/_build/resample_bench
timeout 400 rocprofv3 --kernel-trace --stats -d $O -o headline --output-format csv -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix > $O/headline_bench.json 2> $O/headline_bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o config5 --output-format csv -- python $R/scripts/r5_config5_kernels.py > $O/config5.log 2>&1
i=0
for set in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  for c in affine elastic; do
    timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O -o sq${i}_$c --output-format csv -- $B --cases perf --reps 2 --case "$c f32 fill" --path "tight" > $O/sq${i}_$c.log 2>&1 || echo "set $i $c failed"
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O -o calib_$c --output-format csv -- $B --cases calib > $O/calib_$c.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O -o tight_$c --output-format csv -- $B --cases perf --reps 2 --case "f32 fill" --path "tight" > $O/tight_$c.log 2>&1
done
{
python - <<PY
import csv, glob
for tag in ("headline", "config5"):
    for path in glob.glob("$O/**/%s_kernel_stats.csv" % tag, recursive=True):
        print("==", tag)
        rows = sorted(csv.DictReader(open(path)), key=lambda r: -float(r["TotalDurationNs"]))
        for r in rows[:12]:
            print(r["Name"][:110], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
python $R/scripts/pmc_summary.py $O calib 2>/dev/null
python $R/scripts/pmc_summary.py $O lean_exact 2>/dev/null
python $R/scripts/pmc_summary.py $O plan_bricks 2>/dev/null
} | tee $O/summary.txt
tail -c 400 $O/headline_bench.json
