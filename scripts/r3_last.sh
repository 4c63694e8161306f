# the round's last GPU call: the tests around the fused reference noise after the review fixes, the smoke run, and the
# host-side profiles of one bench step ON THE GPU BOX (cProfile with the real driver, the same step against no-op entry points)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_device_rng.py tests/test_autograd.py tests/test_gpu_lazy_fusion.py tests/test_gpu_golden.py tests/test_gpu_config5.py tests/test_gpu_aggregator.py tests/test_gpu_full_size.py -q 2>&1 | tail -5 > gpurun_out/r3_last_pytest.log; cat gpurun_out/r3_last_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(nproc; lscpu | grep -i "model name\|socket\|thread(s)") > gpurun_out/r3_last_host.txt 2>&1
timeout 200 python scripts/host_profile.py tottime > gpurun_out/r3_host_profile.txt 2>&1; head -3 gpurun_out/r3_host_profile.txt | cut -c1-160
timeout 200 python scripts/host_null_profile.py > gpurun_out/r3_host_null.txt 2>&1; cat gpurun_out/r3_host_null.txt | tail -3
timeout 200 python scripts/host_segments.py > gpurun_out/r3_host_segments.txt 2>&1; cat gpurun_out/r3_host_segments.txt | tail -13
timeout 100 python scripts/r3_plan_timing.py 2>&1 | grep threads > gpurun_out/r3_plan_timing2.log; cat gpurun_out/r3_plan_timing2.log
timeout 400 python bench.py > gpurun_out/r3_last_bench.json 2> gpurun_out/r3_last_bench.err; tail -2 gpurun_out/r3_last_bench.err | cut -c1-200; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_last_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step')}, d['roofline']['frac'], d['roofline']['launch_ms'])
print({k:(round(v['volumes_per_s'],1), v['ms_per_step']) for k,v in d['mode_matrix'].items()})
PY
