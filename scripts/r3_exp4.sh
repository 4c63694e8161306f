mkdir -p gpurun_out
cd tests/native/_build
timeout 300 ./resample_bench --cases parity --path fast > ../../../gpurun_out/r3_exp4_native.log 2>&1
timeout 400 ./resample_bench --cases perf --path fast --reps 20 >> ../../../gpurun_out/r3_exp4_native.log 2>&1
cd ../../..
grep " ms \|failures" gpurun_out/r3_exp4_native.log | cut -c1-200 | grep -v "brick\|packed\|x4 "
timeout 600 python -m pytest tests/test_gpu_resample_planned.py tests/test_gpu_full_size.py -m gpu -q -x 2>&1 | tail -5
