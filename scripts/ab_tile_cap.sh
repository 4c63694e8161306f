cd tests/native/_build
for cap in 0 13440 13568 13650; do
  echo "== TIO_TILE_LDS_FLOATS=$cap"
  if [ $cap = 0 ]; then unset TIO_TILE_LDS_FLOATS; else export TIO_TILE_LDS_FLOATS=$cap; fi
  timeout 200 ./resample_bench --cases perf --case "f32" --path "fast" --reps 20 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general\|gather  " | cut -c1-130
done
