# round 3, experiment 14: lean kernel — wave priority (4: high until the DMA is issued, 8: high while sampling), 8 x 8 wave
# footprint (16), nontemporal stores (32), shader-clock stamps (64; scripts/r3_exp14.sh's first run is in profiles/)
cd tests/native/_build
L=../../../gpurun_out/r3_exp14b_native.log
: > $L
for ab in 0 4 8 12 16 32 20 36 0; do
  echo "== ablate $ab" >> $L
  timeout 200 ./resample_bench --cases perf --case "c f32 fill" --path "fast" --reps 20 --ablate $ab 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general\|gather  " | cut -c1-130 >> $L
done
for gs in 0 0.5 1; do
  for ab in 2 0; do
    echo "== geometry scale $gs, ablate $ab" >> $L
    TIO_BENCH_GEOM_SCALE=$gs timeout 200 ./resample_bench --cases perf --case "affine f32 fill" --path "fast" --reps 20 --ablate $ab 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general\|gather  " | cut -c1-130 >> $L
  done
done
cat $L
