mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r3_full_pytest.log; cat gpurun_out/r3_full_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py > gpurun_out/r3_full_bench.json 2> gpurun_out/r3_full_bench.err; tail -2 gpurun_out/r3_full_bench.err | cut -c1-200; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_full_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step')}, d['roofline']['frac'], d['roofline']['launch_ms'], d['roofline']['traffic'])
print({k:(round(v['volumes_per_s'],1), v['ms_per_step'], v.get('resample_frac_of_hbm_peak')) for k,v in d['mode_matrix'].items()})
for k,v in d['other_configs'].items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})
print(d['aten_baseline']['value'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
