mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_lazy_fusion.py tests/test_gpu_full_size.py -m gpu -q -x 2>&1 | tail -4
timeout 400 python bench.py --steps 50 --no-cpu-baseline --no-aten-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step')}, d['roofline']['frac'], d['roofline']['launch_ms'])
print({k:(round(v['volumes_per_s'],1), v['ms_per_step']) for k,v in d['mode_matrix'].items()})"
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3 -o r3 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_r3 | head; head -12 $GRAFT_REPO_ROOT/gpurun_out/prof_r3/*kernel_stats.csv | cut -c1-220
