#!/bin/bash
# A/B: resample.hip compiled WITH the SLP vectoriser (v_pk_*_f32 in the coordinate chains and lerps) against the tree's build
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
B=tests/native/_build/resample_bench
for round in 1 2; do
  echo "== tree"; timeout 200 $B --cases perf --reps 20 --case "f32" --path "tight" 2>&1 | grep " tight \| lean-exact " | cut -c1-120
  echo "== slp"; LD_LIBRARY_PATH=$PWD/tests/native/_build/slp timeout 200 $B --cases perf --reps 20 --case "f32" --path "tight" 2>&1 | grep " tight \| lean-exact " | cut -c1-120
done
echo "== slp, other paths"; LD_LIBRARY_PATH=$PWD/tests/native/_build/slp timeout 300 $B --cases perf --reps 20 --case "f32" --path "e" 2>&1 | grep "tile16x16x16 \| fast \| lean-exact " | cut -c1-120
LD_LIBRARY_PATH=$PWD/tests/native/_build/slp timeout 300 $B --cases parity 2>&1 | tail -1
