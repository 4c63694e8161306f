mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_device_rng.py -q 2>&1 | tail -12
timeout 900 python -m pytest tests -m gpu -q -k "noise or Noise or golden or lazy or full_size" 2>&1 | tail -4
TIO_HOST_RNG_THREADS=16 timeout 300 python bench.py --noise-rng reference --resample-precision exact --steps 30 --no-other-configs --no-aten-baseline --no-cpu-baseline --no-mode-matrix 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step')})"
