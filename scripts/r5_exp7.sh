#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
B=tests/native/_build/resample_bench
timeout 300 $B --cases parity --path "t" 2>&1 | tail -2
timeout 300 $B --cases perf --reps 20 --case "f32" --path "t" > gpurun_out/r5_native_perf7.txt 2>&1
grep -v "pdb\|seq\|dma1st\|tile8\|tile16x8\|brick\|general" gpurun_out/r5_native_perf7.txt | cut -c1-150
timeout 600 python scripts/host_stress_ranks.py --ranks 8 --steps 150 --out gpurun_out/r5_measure_host_stress_gpu_box_v2.json 2>&1 | tail -7
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_measure_host_stress_gpu_box_v2.json'))
for r in d['8_ranks']: print(r)
PY
