mkdir -p gpurun_out
cd tests/native/_build
export TIO_STAMPS=1
timeout 300 ./resample_bench --cases perf --case "affine f32 fill" --path fast-v2 --reps 10 --ablate 16 > ../../../gpurun_out/r3_exp2_stamps.log 2>&1
timeout 300 ./resample_bench --cases perf --case "elastic f32 fill" --path fast-v2-1:1 --reps 10 --ablate 16 >> ../../../gpurun_out/r3_exp2_stamps.log 2>&1
cd ../../..
grep "stamps\| ms " gpurun_out/r3_exp2_stamps.log | cut -c1-330
