python scripts/bench_configs.py > /dev/null 2>&1
for i in 1 2; do
TIO_CONFIGS_PRECISION=exact python scripts/bench_configs.py 2>/dev/null | grep '"config": "2' | cut -c1-150 | sed 's/^/exact /'
python scripts/bench_configs.py 2>/dev/null | grep '"config": "2' | cut -c1-150 | sed 's/^/fast  /'
done
python scripts/bench_configs.py > gpurun_out/r02_other_configs_fast.json 2>/dev/null
TIO_CONFIGS_PRECISION=exact python scripts/bench_configs.py > gpurun_out/r02_other_configs_exact.json 2>/dev/null
