#!/bin/bash
# round 5: the whole GPU suite, the native harness, the default bench line — every step under timeout
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 90 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r5_gpu_tests_full.txt 2>&1; tail -3 gpurun_out/r5_gpu_tests_full.txt; grep -E "^(FAILED|ERROR)" gpurun_out/r5_gpu_tests_full.txt | head
(timeout 300 tests/native/_build/resample_bench --cases parity > gpurun_out/r5_native_parity_full.txt 2>&1; tail -1 gpurun_out/r5_native_parity_full.txt)
(timeout 300 tests/native/_build/resample_bench --cases perf --reps 20 > gpurun_out/r5_native_perf_full.txt 2>&1; grep -c "ms " gpurun_out/r5_native_perf_full.txt)
timeout 600 python bench.py > gpurun_out/r5_bench_full.json 2> gpurun_out/r5_bench_full.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench_full.json').read().strip().splitlines()[-1])
print('value',round(d['value']),'ms',round(d['ms_per_step'],3),'host',round(d['host_enqueue_ms_per_step'],3),'roof',round(d['roofline']['frac'],3),round(d['roofline']['launch_ms'],3), 'traffic', d['roofline']['traffic'])
for k,v in d['mode_matrix'].items(): print(' ',k, round(v['volumes_per_s']), round(v['ms_per_step'],3), round(v.get('resample_launch_ms',0),3))
print(d.get('draw_policy')); print('ref identical', d.get('value_reference_identical'))
print({k:round(v['volumes_per_s']) for k,v in d['multi_stream'].items() if isinstance(v,dict)})
for k,v in d['other_configs'].items(): print(' ',k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sample']); print('recorded', d['cpu_baseline_reference_recorded']['value'])
print('hbm', d['hbm_measured_ceiling_GBps']['d2d_copy_GBps'], d['hbm_measured_ceiling_GBps']['triad_GBps'], 'aten', d['aten_baseline']['value'])
PY
