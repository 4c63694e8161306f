#!/bin/bash
# Round 6: the label kernel's launch time against resident blocks per CU (unused dynamic LDS as the knob)
R=$GRAFT_REPO_ROOT; B=$R/tests/native/_build/resample_bench; O=$R/gpurun_out/r6_labels; mkdir -p $O
for lds in 0 20000 26000 32000 40000 52000; do
  TIO_NEAREST_LDS=$lds $B --cases labels --reps 20 --path tight 2>&1 | grep -E " tight " | sed "s/mismatch.*//; s/^/lds $lds  /" | tee -a $O/occ_${1:-x}.txt
done
