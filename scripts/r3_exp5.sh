mkdir -p gpurun_out
cd tests/native/_build
timeout 300 ./resample_bench --cases parity --path fast-lean > ../../../gpurun_out/r3_exp5_native.log 2>&1
timeout 400 ./resample_bench --cases perf --path fast-lean --reps 20 >> ../../../gpurun_out/r3_exp5_native.log 2>&1
timeout 100 ./resample_bench --cases perf --path fast-lean-w8 --reps 20 --case "affine f32" --ablate 2 >> ../../../gpurun_out/r3_exp5_native.log 2>&1
cd ../../..
grep " ms \|failures" gpurun_out/r3_exp5_native.log | cut -c1-200
