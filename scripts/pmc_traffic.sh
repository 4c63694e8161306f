# HBM traffic of the resampling kernels from the TCC counters, with a calibration run of
# known byte counts (MI355X_MICROARCH.md §HBM: FETCH_SIZE needs the gfx950 correction,
# WRITE_SIZE is uncalibrated).  Separate --pmc passes; kernel-trace only.
#   bash scripts/pmc_traffic.sh            exact brick kernel (tile16x16x16) and the FAST planned bricks (fast)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; B=$R/tests/native/_build/resample_bench; O=$R/gpurun_out/pmc_traffic; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O -o calib_$c --output-format csv -- $B --cases calib > $O/calib_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c -d $O -o resample_$c --output-format csv -- $B --cases perf --reps 2 --case "f32 fill" --path tile16x16x16 > $O/resample_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c -d $O -o planned_$c --output-format csv -- $B --cases perf --reps 2 --case "f32 fill" --path "fast" > $O/planned_$c.log 2>&1
done
ls $O | head -30
