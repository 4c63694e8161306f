# round 3, experiment 15: is the LDS port what the DMA and the sampling share?  (128: every lane samples one address);
# residency of blocks per CU from the stamps (64)
cd tests/native/_build
L=../../../gpurun_out/r3_exp15_native.log
: > $L
for ab in 0 1 2 128 129 4 8 16 32; do
  echo "== ablate $ab" >> $L
  timeout 200 ./resample_bench --cases perf --case "affine f32 fill" --path "fast" --reps 20 --ablate $ab 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general\|gather  " | cut -c1-130 >> $L
done
for ab in 64 65 66; do
  echo "== stamps, ablate $ab" >> $L
  timeout 200 ./resample_bench --cases perf --case "affine f32 fill" --path "fast" --reps 3 --ablate $ab 2>&1 | grep -v "fast-brick\|fast-general\|gather  " | cut -c1-160 >> $L
done
cat $L
