mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r3_full_pytest.log; cat gpurun_out/r3_full_pytest.log
timeout 300 python bench.py --steps 30 > gpurun_out/r3_full_bench.json 2> gpurun_out/r3_full_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_full_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step')}, d['roofline']['frac'], d['roofline']['launch_ms'])
print({k:(v['volumes_per_s'], v.get('resample_launch_ms')) for k,v in d['mode_matrix'].items()})
PY
