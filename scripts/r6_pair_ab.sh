#!/bin/bash
# Round 6: two channels per exact-coordinate launch (TIO_LEAN_PAIR) — parity cases of the native harness, then the subject cases both ways
R=$GRAFT_REPO_ROOT; B=$R/tests/native/_build/resample_bench; O=$R/gpurun_out/r6_pair; mkdir -p $O
$B --cases parity 2>&1 | tail -4
for rep in 1 2; do
for p in 1 0; do
  for path in tight lean-exact; do
    TIO_LEAN_PAIR=$p $B --cases perf --reps 20 --path $path --case subject 2>&1 | grep -E " $path " | sed "s/^/pair=$p  /; s/mismatch vs gather: 0 *//" | tee -a $O/ab.txt
  done
done
done
python -m pytest tests/test_gpu_golden.py tests/test_gpu_full_size.py tests/test_gpu_tight.py tests/test_cabi.py tests/test_gpu_resample.py -x -q 2>&1 | tail -3
