#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops_parity.py tests/test_gpu_lazy_fusion.py tests/test_gpu_full_size.py -m gpu -q -k "philox or headline or noise or fused" 2>&1 | tail -5
python scripts/r5_headline_error_budget.py 2>&1 | grep "noise=True"
for f in gpurun_out/headline_parity_tight_*.json; do echo "$f: $(cat $f)"; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix > gpurun_out/r5_bench4.json 2> gpurun_out/r5_bench4.err
python -c "
import json; d=json.loads(open('gpurun_out/r5_bench4.json').read().strip().splitlines()[-1]); print('value',d['value'],'ms',d['ms_per_step'],'host',d['host_enqueue_ms_per_step'],'launch',d['roofline']['launch_ms'])"
timeout 300 python scripts/bench_blur_stages.py 2>&1 | tail -12
