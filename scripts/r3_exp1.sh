# round 3, experiment 1: the new GPU tests + every planned2 variant (parity with the split self-check, then timing)
mkdir -p gpurun_out
cd tests/native/_build
echo "== parity, split self-check on (TIO_TILE_ABLATE=8)" > ../../../gpurun_out/r3_exp1_native.log
timeout 300 ./resample_bench --cases parity --path fast --ablate 8 >> ../../../gpurun_out/r3_exp1_native.log 2>&1
echo "== perf" >> ../../../gpurun_out/r3_exp1_native.log
unset TIO_TILE_ABLATE
timeout 400 ./resample_bench --cases perf --path fast --reps 20 >> ../../../gpurun_out/r3_exp1_native.log 2>&1
echo "== perf with check (NaN = hint violated)" >> ../../../gpurun_out/r3_exp1_native.log
timeout 300 ./resample_bench --cases perf --path fast-v2 --reps 3 --ablate 8 >> ../../../gpurun_out/r3_exp1_native.log 2>&1
cd ../../..
timeout 900 python -m pytest tests/test_gpu_lazy_fusion.py tests/test_gpu_full_size.py tests/test_gpu_resample_planned.py tests/test_autograd.py tests/test_torch_ops.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r3_exp1_pytest.log
tail -5 gpurun_out/r3_exp1_pytest.log
grep -c "" gpurun_out/r3_exp1_native.log
