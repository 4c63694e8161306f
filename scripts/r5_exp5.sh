#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
B=tests/native/_build/resample_bench
timeout 300 $B --cases parity --path pdb > gpurun_out/r5_native_parity5.txt 2>&1; echo "parity rc $?" >> gpurun_out/r5_native_parity5.txt
grep -v "mismatch vs gather: 0  .*vs oracle: 0$\|mismatch vs gather: 0  vs oracle: 0$" gpurun_out/r5_native_parity5.txt | tail -15
timeout 300 $B --cases perf --reps 20 --case "affine f32" --path "t" > gpurun_out/r5_native_perf5.txt 2>&1
grep -v "tile16x8\|tile8x\|brick\|general\|seq\|dma1st" gpurun_out/r5_native_perf5.txt | cut -c1-170
timeout 300 $B --cases perf --reps 10 --case "subject" --path "pdb" 2>&1 | cut -c1-170
