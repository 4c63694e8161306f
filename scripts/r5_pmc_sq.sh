#!/bin/bash
# SQ counters of the exact-coordinate kernel, one block per brick ("tight") vs persistent ("tight-pdb"): scalar vs vector issue
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r5_pmc_sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tests/native/_build/resample_bench
i=0
for set in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O -o set$i --output-format csv -- $B --cases perf --reps 2 --case "affine f32" --path "tight" > $O/set$i.log 2>&1 || echo "set $i failed: $(tail -2 $O/set$i.log)"
done
python $R/scripts/pmc_summary.py $O lean_exact
