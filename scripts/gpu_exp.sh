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
      # rocprofv3 kernel statistics of (1) the headline bench command, (2) the reference-identical (library default) mode
      R=$PWD; O=$R/gpurun_out/r4_prof; mkdir -p $O
      (cd /tmp && export TMPDIR=/tmp
       timeout 300 rocprofv3 --kernel-trace --stats -d $O -o headline --output-format csv -- python $R/bench.py --steps 20 --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix > $O/headline.log 2>&1
       timeout 300 rocprofv3 --kernel-trace --stats -d $O -o reference --output-format csv -- python $R/bench.py --steps 20 --noise-rng reference --resample-precision exact --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix > $O/reference.log 2>&1)
      for f in headline reference; do echo "== $f"; find $O -name "${f}_kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-170'; tail -c 400 $O/$f.log | head -c 400; echo; done ;;
    prof_refnoise)
      # reference-noise modes: where the host's time goes (cProfile) and what the GPU runs (rocprofv3 kernel statistics), with the
      # draws ahead on the draw stream and on the previous road (TIO_NO_DRAW_STREAM=1)
      R=$PWD; O=$R/gpurun_out/r4b_prof_refnoise; mkdir -p $O
      for road in 0 1; do
        echo "== host profile, reference,fast, TIO_NO_DRAW_STREAM=$road"
        TIO_NO_DRAW_STREAM=$road TIO_PROFILE_MODE=reference,fast timeout 200 python scripts/host_profile.py tottime 2>&1 | grep -v "^$" | head -34 | cut -c1-150 | tee $O/host_profile_road$road.txt
      done
      (cd /tmp && export TMPDIR=/tmp
       for road in 0 1; do
         TIO_NO_DRAW_STREAM=$road timeout 300 rocprofv3 --kernel-trace --stats -d $O -o road$road --output-format csv -- python $R/bench.py --steps 30 --noise-rng reference --resample-precision fast --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix > $O/road$road.log 2>&1
       done)
      for road in 0 1; do echo "== kernels, TIO_NO_DRAW_STREAM=$road"; find $O -name "road${road}_kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -14 {} | cut -c1-150'; python -c "import json,sys; d=json.loads(open('$O/road$road.log').read().strip().splitlines()[-1]); print(round(d['value'],1), 'vol/s', round(d['ms_per_step'],4), 'ms/step host', round(d['host_enqueue_ms_per_step'],3))"; done ;;
    pmc)
      # HBM traffic of the lean planned kernels (FETCH_SIZE / WRITE_SIZE, separate passes, with the calibration kernels of known byte counts)
      R=$PWD; B=$R/tests/native/_build/resample_bench; O=$R/gpurun_out/r4_pmc; mkdir -p $O
      (cd /tmp && export TMPDIR=/tmp
       for c in FETCH_SIZE WRITE_SIZE; do
         timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O -o calib_$c --output-format csv -- $B --cases calib > $O/calib_$c.log 2>&1
         timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O -o planned_$c --output-format csv -- $B --cases perf --reps 2 --case "f32 fill" --path "fast" > $O/planned_$c.log 2>&1
       done)
      python scripts/pmc_summary.py $(dirname $(find $O -name "calib_FETCH_SIZE_counter_collection.csv" | head -1)) "" 2>&1 | grep -v "fast-\|gather" | tee gpurun_out/r4_pmc_traffic_lean.txt | tail -60 ;;
    tests_ahead)
      timeout 900 python -m pytest tests/test_gpu_ahead.py tests/test_gpu_resample_planned.py tests/test_gpu_full_size.py -m gpu -q --tb=short -x 2>&1 | tail -60 > gpurun_out/r4_gpu_tests_ahead.txt
      grep -v "Warning\|^$\|warnings.html\|^  " gpurun_out/r4_gpu_tests_ahead.txt | tail -40 ;;
    ab_ahead)
      # the side stream (uploads + brick plans ahead of the data) and the announced minimum, on and off, alternating
      L=gpurun_out/r4_ab_ahead.log; : > $L
      for rep in 1 2 3; do
        for off in 0 3 1 2; do
          TIO_NO_AHEAD_STREAM=$((off & 1)) TIO_NO_ANNOUNCED_MIN=$((off >> 1)) timeout 200 python bench.py --steps 60 --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix 2>/dev/null |
            python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('off(1=stream,2=announced min)=$off', round(d['value'],1), 'vol/s', round(d['ms_per_step'],4), 'ms/step host', round(d['host_enqueue_ms_per_step'],3), 'launch_ms', round(d['roofline']['launch_ms'],4))" >> $L
        done
      done
      cat $L ;;
    *) echo "unknown step $step" ;;
  esac
done
