#!/bin/bash
R=$GRAFT_REPO_ROOT; B=$R/tests/native/_build/resample_bench; O=$R/gpurun_out/r6_label_ablate; mkdir -p $O
for rep in 1 2; do
for ab in 0 256 512 768; do
  TIO_TILE_ABLATE=$ab $B --cases perf --reps 20 --path tight --case "subject" 2>&1 | grep -E " tight " | sed "s/^/ablate=$ab  /; s/(max.*//" | tee -a $O/ab.txt
done
TIO_LEAN_LABEL=0 $B --cases perf --reps 20 --path tight --case "subject" 2>&1 | grep -E " tight " | sed "s/^/own-kernel  /; s/(max.*//" | tee -a $O/ab.txt
done
