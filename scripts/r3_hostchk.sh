mkdir -p gpurun_out
timeout 300 python bench.py --steps 50 --no-cpu-baseline --no-aten-baseline --no-mode-matrix 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step','host_settling_ms_per_step')}, d['roofline']['frac'])"
timeout 300 python scripts/host_variants.py 2>&1 | tail -6
