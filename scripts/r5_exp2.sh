#!/bin/bash
# round 5, experiment 2: order of control-point loads / DMA in the exact-coordinate kernel; first bench line in the tight mode
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
B=tests/native/_build/resample_bench
timeout 600 $B --cases perf --reps 20 --path t > gpurun_out/r5_native_perf2.txt 2>&1; echo "perf rc $?" >> gpurun_out/r5_native_perf2.txt
grep -v "tile16x8\|tile8x\|labels" gpurun_out/r5_native_perf2.txt | cut -c1-150
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-aten-baseline > gpurun_out/r5_bench2.json 2> gpurun_out/r5_bench2.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench2.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'host',d['host_enqueue_ms_per_step'],'roof',d['roofline']['frac'],d['roofline']['launch_ms'])
for k,v in d['mode_matrix'].items(): print(k, round(v['volumes_per_s']), round(v['ms_per_step'],3), round(v.get('resample_launch_ms',0),3))
print(d.get('draw_policy'))
print({k:round(v['volumes_per_s']) for k,v in d['multi_stream'].items() if isinstance(v,dict)})
for k,v in d['other_configs'].items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})
PY
tail -5 gpurun_out/r5_bench2.err
