#!/bin/bash
# round 5, final tree: HBM traffic of the exact-coordinate kernel (tight) from the TCC counters — separate --pmc passes, kernel trace only,
# calibrated on known byte counts (MI355X_MICROARCH.md: FETCH_SIZE tallies 128-byte requests at 64 B on gfx950).
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r5_pmc_final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tests/native/_build/resample_bench
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O -o calib_$c --output-format csv -- $B --cases calib > $O/calib_$c.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O -o tight_$c --output-format csv -- $B --cases perf --reps 2 --case "f32 fill" --path "tight" > $O/tight_$c.log 2>&1
done
python $R/scripts/pmc_summary.py $O calib 2>/dev/null | tee $O/summary.txt
python $R/scripts/pmc_summary.py $O lean_exact 2>/dev/null | tee -a $O/summary.txt
