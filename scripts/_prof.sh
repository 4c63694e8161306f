cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_bench_fast -o bf --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --no-aten-baseline --no-mode-matrix --steps 20 > /root/repo/gpurun_out/prof_bench_fast.json 2>/root/repo/gpurun_out/prof_bench_fast.err
ls /root/repo/gpurun_out/prof_bench_fast | head
head -30 /root/repo/gpurun_out/prof_bench_fast/*kernel_stats.csv | cut -c1-200
