# round 3, experiment 19: the nearest-neighbour kernel (resample_nearest.hpp): parity, the decision margin, timing
cd tests/native/_build
L=../../../gpurun_out/r3_exp19_native.log
: > $L
echo "== parity (fast paths + exact tile path)" >> $L
timeout 600 ./resample_bench --cases parity --path fast 2>&1 | grep -v "fast-brick\|fast-general" | grep "nearest\|subject\|failures" | cut -c1-170 >> $L
timeout 600 ./resample_bench --cases parity --path tile16x16x16 2>&1 | grep "nearest\|subject\|failures" | cut -c1-170 >> $L
echo "== margin scan: mismatches against the exact chain with the FAST line deciding at eps (S + |x|)" >> $L
for eps in 0 1e-8 3e-8 1e-7 3e-7 1e-6 2e-6; do
  echo "-- eps $eps" >> $L
  TIO_NEAREST_EPS=$eps timeout 300 ./resample_bench --cases perf --case "labels" --path "fast" --reps 5 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general" | cut -c1-170 >> $L
done
echo "== timing" >> $L
timeout 300 ./resample_bench --cases perf --case "labels" --path "fast" --reps 20 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general" | cut -c1-170 >> $L
timeout 300 ./resample_bench --cases perf --case "subject" --path "fast" --reps 10 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general" | cut -c1-170 >> $L
timeout 300 ./resample_bench --cases perf --case "subject" --path "tile16x16x16" --reps 10 2>&1 | grep " ms " | cut -c1-170 >> $L
cat $L
