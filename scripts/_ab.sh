timeout 600 python -m pytest tests/test_gpu_resample_planned.py tests/test_gpu_ops_parity.py -x -q 2>&1 | tail -25
Q="--no-cpu-baseline --no-aten-baseline --no-mode-matrix"
j() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(sys.argv[1], 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'launch_ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'host', round(d['host_enqueue_ms_per_step'],3))" "$1"; }
for rep in 1 2; do
python bench.py $Q 2>/dev/null | j "fast folded    "
TIO_NO_FOLDED_MIN=1 python bench.py $Q 2>/dev/null | j "fast unfolded  "
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_bench_fast3 -o bf --output-format csv -- python /root/repo/bench.py $Q --steps 20 > /dev/null 2>&1
head -12 /root/repo/gpurun_out/prof_bench_fast3/bf_kernel_stats.csv | cut -c1-150
