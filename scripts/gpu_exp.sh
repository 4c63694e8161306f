#!/bin/bash
# Round-4 GPU session driver (through gpurun, from the repository root).  Every step under `timeout`; logs under gpurun_out/.
#   bash scripts/gpu_exp.sh <step> [<step> ...]
# steps: tests | parity | ab_fast (new vs tests/native/_build/base/libtio_hip.so on the FAST resampling cases) |
#        perf_fast | perf_exact | bench | bench_quick | stencil | prof
set -u
mkdir -p gpurun_out
NATIVE=tests/native/_build
for step in "$@"; do
  case "$step" in
    tests)
      timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -200 > gpurun_out/r4_gpu_tests.txt
      grep -v "Warning\|^$\|warnings.html\|^  " gpurun_out/r4_gpu_tests.txt | tail -40 ;;
    parity)
      (cd $NATIVE && timeout 300 ./resample_bench --cases parity 2>&1) > gpurun_out/r4_native_parity_full.txt
      grep -v "mismatch vs gather: 0 .*vs oracle: 0$\|mismatch vs gather: 0  vs oracle: 0$" gpurun_out/r4_native_parity_full.txt | tail -30 | tee gpurun_out/r4_native_parity.txt ;;
    ab_fast)
      L=gpurun_out/r4_ab_fast.log; : > $L
      for rep in 1 2; do
        for lib in new base; do
          echo "== $lib (rep $rep)" >> $L
          if [ $lib = base ]; then export LD_LIBRARY_PATH=$PWD/$NATIVE/base; else unset LD_LIBRARY_PATH; fi
          (cd $NATIVE && timeout 200 ./resample_bench --cases perf --case "f32" --path "fast" --reps 20 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general\|gather  " | cut -c1-130) >> $L
        done
      done
      unset LD_LIBRARY_PATH; cat $L ;;
    perf_fast)
      (cd $NATIVE && timeout 200 ./resample_bench --cases perf --case "f32" --path "fast" --reps 20 2>&1 | grep " ms " | cut -c1-130) | tee gpurun_out/r4_perf_fast.log ;;
    perf_exact)
      (cd $NATIVE && timeout 300 ./resample_bench --cases perf --case "f32" --reps 20 2>&1 | grep " ms " | cut -c1-130) | tee gpurun_out/r4_perf_all.log ;;
    bench)
      timeout 600 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; tail -c 1500 gpurun_out/r4_bench.json; tail -3 gpurun_out/r4_bench.err ;;
    bench_quick)
      timeout 300 python bench.py --no-cpu-baseline --no-aten-baseline --no-other-configs > gpurun_out/r4_bench_quick.json 2> gpurun_out/r4_bench_quick.err; tail -c 1200 gpurun_out/r4_bench_quick.json ;;
    stencil)
      timeout 200 python scripts/bench_blur_stages.py 2>&1 | tail -20 | tee gpurun_out/r4_stencil.log ;;
    prof)
      cd /tmp && export TMPDIR=/tmp
      timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix > $GRAFT_REPO_ROOT/gpurun_out/r4_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4_prof.err
      cd $GRAFT_REPO_ROOT; find gpurun_out/r4_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -25 {}' ;;
    *) echo "unknown step $step" ;;
  esac
done
