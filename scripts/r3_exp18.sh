# round 3, experiment 18: brick shapes of the lean kernel (TIO_LEAN_SHAPE: 0 = 16x16x16 / 256 threads / 3 per CU,
# 1 = 16x16x32 / 512 threads / 2 per CU, 2 = 8x16x32 / 512 threads / 3 per CU)
cd tests/native/_build
L=../../../gpurun_out/r3_exp18_native.log
: > $L
for sh in 1 2; do
  echo "== parity, shape $sh" >> $L
  TIO_LEAN_SHAPE=$sh timeout 300 ./resample_bench --cases parity --path fast 2>&1 | grep -v "fast-brick\|gather  " | grep "fast \|failures" | cut -c1-150 >> $L
done
for sh in 0 1 2 0 1 2; do
  echo "== shape $sh" >> $L
  TIO_LEAN_SHAPE=$sh timeout 200 ./resample_bench --cases perf --case "f32" --path "fast" --reps 20 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general\|gather  " | cut -c1-130 >> $L
done
for sh in 1 2; do
  echo "== stamps: shape $sh" >> $L
  TIO_LEAN_SHAPE=$sh timeout 200 ./resample_bench --cases perf --case "affine f32 fill" --path "fast" --reps 3 --ablate 64 2>&1 | grep -v "fast-brick\|fast-general\|gather  \|none found" | cut -c1-160 >> $L
  for ab in 1 2; do
    echo "== shape $sh ablate $ab" >> $L
    TIO_LEAN_SHAPE=$sh timeout 200 ./resample_bench --cases perf --case "affine f32 fill" --path "fast" --reps 20 --ablate $ab 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general\|gather  " | cut -c1-130 >> $L
  done
done
cat $L
