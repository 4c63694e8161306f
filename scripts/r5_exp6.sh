#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
B=tests/native/_build/resample_bench
timeout 300 $B --cases perf --reps 20 --case "f32" --path "tight" > gpurun_out/r5_native_perf6.txt 2>&1
grep -v "pdb" gpurun_out/r5_native_perf6.txt | cut -c1-175
timeout 300 $B --cases perf --reps 20 --case "f32" --path "lean-exact" 2>&1 | grep -v "pdb\|gather " | cut -c1-130
