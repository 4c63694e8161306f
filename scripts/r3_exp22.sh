# round 3, experiment 22: 32 planes per column in the nearest kernel; the drawing kernel with 320 threads and no per-value division
cd tests/native/_build
L=../../../gpurun_out/r3_exp22_native.log
: > $L
for pl in 16 32; do
  echo "== parity, planes $pl" >> $L
  TIO_NEAREST_PLANES=$pl timeout 600 ./resample_bench --cases parity --path fast 2>&1 | grep -v "fast-brick\|fast-general\| gather  " | grep "failures" >> $L
  echo "== timing, planes $pl" >> $L
  TIO_NEAREST_PLANES=$pl timeout 300 ./resample_bench --cases perf --case "labels" --path "fast" --reps 20 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general\| gather  " | cut -c1-150 >> $L
  TIO_NEAREST_PLANES=$pl timeout 300 ./resample_bench --cases perf --case "subject 512" --path "fast" --reps 10 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general\| gather  " | cut -c1-150 >> $L
done
cat $L
cd ../../..
timeout 900 python -m pytest tests/test_gpu_device_rng.py tests/test_gpu_nearest_kernel.py -q 2>&1 | tail -3
TIO_HOST_RNG_THREADS=16 timeout 300 python bench.py --noise-rng reference --resample-precision exact --steps 30 --no-other-configs --no-aten-baseline --no-cpu-baseline --no-mode-matrix 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step')})"
