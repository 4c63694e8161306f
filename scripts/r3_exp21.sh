# round 3, experiment 21: the nearest kernel with one explicit wait between its loads and its stores
cd tests/native/_build
L=../../../gpurun_out/r3_exp21_native.log
: > $L
echo "== parity (fast paths + exact tile path)" >> $L
timeout 600 ./resample_bench --cases parity --path fast 2>&1 | grep -v "fast-brick\|fast-general\| gather  " | grep "nearest\|subject\|failures" | cut -c1-150 >> $L
timeout 600 ./resample_bench --cases parity --path tile16x16x16 2>&1 | grep -v " gather  " | grep "nearest\|subject\|failures" | cut -c1-150 >> $L
echo "== timing (eps 0 = no voxel goes the exact way; then the default)" >> $L
TIO_NEAREST_EPS=0 timeout 300 ./resample_bench --cases perf --case "labels" --path "fast" --reps 20 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general\| gather  " | cut -c1-170 >> $L
timeout 300 ./resample_bench --cases perf --case "labels" --path "fast" --reps 20 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general" | cut -c1-170 >> $L
timeout 300 ./resample_bench --cases perf --case "subject" --path "fast" --reps 10 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general" | cut -c1-170 >> $L
timeout 300 ./resample_bench --cases perf --case "subject" --path "tile16x16x16" --reps 10 2>&1 | grep " ms " | cut -c1-170 >> $L
cat $L
