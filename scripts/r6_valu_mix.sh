#!/bin/bash
# Round 6: the opcode-class mix of the exact-coordinate kernel's vector instructions (tight; affine, elastic and fused launches of the
# native harness) — float add / mul / fma issue at ~2.6 - 2.9 cycles per wave64 instruction per SIMD with three to four resident waves,
# everything else at ~4.3 - 4.8 (profiles/r01_valu_rates_w1-8.log) — for the issue-time estimate beside the HBM roofline.
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6_valu_mix; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tests/native/_build/resample_bench
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  for c in "affine f32 fill" "elastic f32 fill" "affine+elastic f32 nofill"; do
    tag=$(echo "$c" | cut -d' ' -f1 | tr '+' '_')
    timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O -o mix${i}_$tag --output-format csv -- $B --cases perf --reps 2 --case "$c" --path "tight" > $O/mix${i}_$tag.log 2>&1 || echo "set $i $c failed"
  done
done
python $R/scripts/pmc_summary.py $O lean_exact 2>/dev/null | tee $O/summary.txt
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
