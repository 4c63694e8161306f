# round 3, experiment 16: the box through registers (global_load_dwordx4 + ds_write_b128; 256) instead of the LDS-DMA
cd tests/native/_build
L=../../../gpurun_out/r3_exp16_native.log
: > $L
echo "== parity, 256" >> $L
timeout 300 ./resample_bench --cases parity --path fast --ablate 256 2>&1 | tail -2 >> $L
for ab in 0 256 2 258 0 256; do
  echo "== ablate $ab" >> $L
  timeout 200 ./resample_bench --cases perf --case "f32 fill" --path "fast" --reps 20 --ablate $ab 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general\|gather  \|b1" | cut -c1-130 >> $L
done
for ab in 320 322; do
  echo "== stamps, ablate $ab" >> $L
  timeout 200 ./resample_bench --cases perf --case "affine f32 fill" --path "fast" --reps 3 --ablate $ab 2>&1 | grep -v "fast-brick\|fast-general\|gather  \|none found" | cut -c1-160 >> $L
done
cat $L
