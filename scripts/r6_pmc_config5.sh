#!/bin/bash
# Round 6: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, kernel trace only) and vector / scalar instruction counts of config 5's ONE launch
# (resample_lean_exact_label_kernel: two float32 images + the int16 label map of a 512^3 subject) from the native harness.
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6_pmc_config5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tests/native/_build/resample_bench
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O -o c5_$tag --output-format csv -- $B --cases perf --reps 2 --case "subject 512^3" --path "tight" > $O/c5_$tag.log 2>&1 || echo "$c failed"
done
python $R/scripts/pmc_summary.py $O label_kernel 2>/dev/null | tee $O/summary.txt
python $R/scripts/pmc_summary.py $O plan_bricks 2>/dev/null | tee -a $O/summary.txt
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
