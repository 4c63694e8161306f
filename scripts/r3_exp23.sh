# round 3, experiment 23: planned boxes for EXACT elastic launches (TIO_EXACT_PLAN=3) with this round's planner
cd tests/native/_build
L=../../../gpurun_out/r3_exp23_native.log
: > $L
for v in 2 3 2 3; do
  echo "== TIO_EXACT_PLAN=$v" >> $L
  TIO_EXACT_PLAN=$v timeout 300 ./resample_bench --cases perf --case "f32" --path "tile16x16x16" --reps 20 2>&1 | grep " ms " | grep -v " gather  \|subject" | cut -c1-140 >> $L
done
echo "== parity with TIO_EXACT_PLAN=3" >> $L
TIO_EXACT_PLAN=3 timeout 600 ./resample_bench --cases parity --path tile16x16x16 2>&1 | tail -1 >> $L
cat $L
