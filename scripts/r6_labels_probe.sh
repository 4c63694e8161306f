#!/bin/bash
# Round 6: what bounds the label kernel.  Timings of the labels cases (element size x chain), then kernel statistics and the
# HBM counters of the same launches (separate --pmc passes, kernel trace only).
R=$GRAFT_REPO_ROOT; B=$R/tests/native/_build/resample_bench; O=$R/gpurun_out/r6_labels; mkdir -p $O
$B --cases labels --reps 20 --path tight 2>&1 | grep -E " tight " > $O/times.txt
TIO_NEAREST_EXACT=0 $B --cases labels --reps 20 --path tight 2>&1 | grep -E " tight " | sed 's/^/line-kernel /' >> $O/times.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o stats --output-format csv -- $B --cases labels --reps 5 --path tight > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c -d $O -o pmc_$tag --output-format csv -- $B --cases labels --reps 2 --path tight > $O/pmc_$tag.log 2>&1
done
cd $R
python scripts/pmc_summary.py $O nearest > $O/pmc_summary.txt 2>&1
cat $O/times.txt
grep -h "nearest" $O/stats_kernel_stats.csv | head
cat $O/pmc_summary.txt | head -120
