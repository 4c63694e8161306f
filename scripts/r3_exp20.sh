# round 3, experiment 20: the nearest kernel after the rewrite (compact argument block, 24-bit offsets, one compare per axis)
cd tests/native/_build
L=../../../gpurun_out/r3_exp20_native.log
: > $L
echo "== parity (fast paths + exact tile path)" >> $L
timeout 600 ./resample_bench --cases parity --path fast 2>&1 | grep -v "fast-brick\|fast-general\| gather  " | grep "nearest\|subject\|failures" | cut -c1-150 >> $L
timeout 600 ./resample_bench --cases parity --path tile16x16x16 2>&1 | grep -v " gather  " | grep "nearest\|subject\|failures" | cut -c1-150 >> $L
echo "== margin scan" >> $L
for eps in 0 1e-8 3e-8 1e-7 1e-6; do
  echo "-- eps $eps" >> $L
  TIO_NEAREST_EPS=$eps timeout 300 ./resample_bench --cases perf --case "labels" --path "fast" --reps 5 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general\| gather  " | cut -c1-170 >> $L
done
echo "== timing" >> $L
timeout 300 ./resample_bench --cases perf --case "labels" --path "fast" --reps 20 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general" | cut -c1-170 >> $L
timeout 300 ./resample_bench --cases perf --case "subject" --path "fast" --reps 10 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general" | cut -c1-170 >> $L
timeout 300 ./resample_bench --cases perf --case "subject" --path "tile16x16x16" --reps 10 2>&1 | grep " ms " | cut -c1-170 >> $L
cat $L
