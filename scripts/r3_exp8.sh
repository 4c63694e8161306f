mkdir -p gpurun_out
(cd tests/native/_build && timeout 300 ./resample_bench --cases all --path fast --reps 10 > ../../../gpurun_out/r3_exp8_native.log 2>&1)
grep "failures\|subject" gpurun_out/r3_exp8_native.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_resample_planned.py tests/test_gpu_full_size.py tests/test_gpu_config5.py tests/test_gpu_ops_parity.py tests/test_gpu_resample_tile.py tests/test_gpu_golden.py -m gpu -q -x 2>&1 | tail -5
TIO_CONFIGS_PRECISION=fast timeout 300 python scripts/bench_configs.py > gpurun_out/r3_configs_fast.json 2>&1
TIO_CONFIGS_PRECISION=exact timeout 300 python scripts/bench_configs.py > gpurun_out/r3_configs_exact.json 2>&1
cat gpurun_out/r3_configs_fast.json gpurun_out/r3_configs_exact.json | cut -c1-260
