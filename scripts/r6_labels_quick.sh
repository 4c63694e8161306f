#!/bin/bash
# Round 6: the nearest tests, then timings of the labels cases (element size x chain) and the subject cases
R=$GRAFT_REPO_ROOT; B=$R/tests/native/_build/resample_bench; O=$R/gpurun_out/r6_labels; mkdir -p $O
python -m pytest tests/test_gpu_nearest_kernel.py tests/test_gpu_golden.py -x -q 2>&1 | tail -3
$B --cases labels --reps 20 --path tight 2>&1 | grep -E " tight " | tee $O/times_${1:-new}.txt
$B --cases perf --reps 20 --path tight 2>&1 | grep -E "(subject|labels).* tight " | tee -a $O/times_${1:-new}.txt
