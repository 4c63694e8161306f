#!/bin/bash
# round 5, experiment 1: the lean exact-coordinate kernel — native parity (every path vs gather / oracle), native perf, GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
B=tests/native/_build/resample_bench
timeout 900 $B --cases parity > gpurun_out/r5_native_parity.txt 2>&1; echo "parity rc $?" >> gpurun_out/r5_native_parity.txt
timeout 900 $B --cases perf --reps 20 > gpurun_out/r5_native_perf.txt 2>&1; echo "perf rc $?" >> gpurun_out/r5_native_perf.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gpu_tests1.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r5_gpu_tests1.txt
tail -3 gpurun_out/r5_native_parity.txt; grep -v "tile16x8\|tile8x" gpurun_out/r5_native_perf.txt | tail -60; tail -3 gpurun_out/r5_gpu_tests1.txt
