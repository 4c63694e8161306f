mkdir -p gpurun_out
python scripts/r3_plan_timing.py 2>&1 | tail -9 | tee gpurun_out/r3_plan_timing.log
timeout 900 python -m pytest tests/test_gpu_device_rng.py tests/test_host_rng.py -q 2>&1 | tail -4
for thr in 8 16; do
TIO_HOST_RNG_THREADS=$thr timeout 300 python bench.py --noise-rng reference --resample-precision exact --steps 30 --no-other-configs --no-aten-baseline --no-cpu-baseline --no-mode-matrix 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('threads $thr', {k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step')})"
done
