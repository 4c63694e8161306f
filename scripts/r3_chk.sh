mkdir -p gpurun_out
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_lazy_fusion.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python bench.py --steps 30 > gpurun_out/r3_full_bench.json 2> gpurun_out/r3_full_bench.err; tail -3 gpurun_out/r3_full_bench.err | cut -c1-300; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_full_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step')}, d['roofline']['frac'], d['roofline']['launch_ms'])
print({k:(v['volumes_per_s'], v.get('resample_launch_ms')) for k,v in d['mode_matrix'].items()})
PY
