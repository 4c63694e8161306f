# round 3, experiment 17: 12 / 14 planes per brick with a fourth (fifth) block per CU (TIO_LEAN_TI, TIO_TILE_LDS_FLOATS)
cd tests/native/_build
L=../../../gpurun_out/r3_exp17_native.log
: > $L
echo "== parity, TI 12 cap 10112" >> $L
TIO_LEAN_TI=12 TIO_TILE_LDS_FLOATS=10112 timeout 300 ./resample_bench --cases parity --path fast 2>&1 | tail -2 >> $L
for cfg in "16 0" "14 0" "14 10112" "12 0" "12 10112" "12 8064" "16 0"; do
  set -- $cfg
  echo "== planes $1, LDS floats $2" >> $L
  TIO_LEAN_TI=$1 TIO_TILE_LDS_FLOATS=$2 timeout 200 ./resample_bench --cases perf --case "f32 fill" --path "fast" --reps 20 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general\|gather  \|b1" | cut -c1-130 >> $L
done
for cfg in "14 10112" "12 10112"; do
  set -- $cfg
  echo "== stamps: planes $1, LDS floats $2" >> $L
  TIO_LEAN_TI=$1 TIO_TILE_LDS_FLOATS=$2 timeout 200 ./resample_bench --cases perf --case "affine f32 fill" --path "fast" --reps 3 --ablate 64 2>&1 | grep -v "fast-brick\|fast-general\|gather  \|none found" | cut -c1-160 >> $L
done
cat $L
