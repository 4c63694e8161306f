#!/bin/bash
# Round 6: the bench launches with the large-box hint forced to 0 / 1 (does listing the few bricks beyond the tile pay at the bench's ranges?)
R=$GRAFT_REPO_ROOT; B=$R/tests/native/_build/resample_bench; O=$R/gpurun_out/r6_hint; mkdir -p $O
for rep in 1 2; do for h in 0 1; do
  TIO_BENCH_HINT=$h $B --cases perf --reps 20 --path tight 2>&1 | grep -E "f32.* tight " | sed "s/^/hint=$h  /; s/mismatch vs gather: 0 *//" | tee -a $O/ab.txt
done; done
