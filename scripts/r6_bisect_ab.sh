#!/bin/bash
# Round 6: the native harness's float32 perf cases with the libraries of several commits (tests/native/_build/bisect/libtio_hip_<sha>.so,
# built by scripts/r6_build_base.sh) and the working tree's, alternating, on ONE box — where did a launch's time move between two rounds?
P=${1:-tight}; ROUNDS=${2:-3}
B=tests/native/_build; O=gpurun_out/r6_bisect.txt; mkdir -p gpurun_out; : > $O
for r in $(seq $ROUNDS); do
  for lib in $(ls $B/bisect/libtio_hip_*.so) HEAD; do
    if [ $lib = HEAD ]; then unset LD_LIBRARY_PATH; tag=HEAD; else D=$(mktemp -d); cp $lib $D/libtio_hip.so; export LD_LIBRARY_PATH=$D; tag=$(basename $lib .so | sed 's/libtio_hip_//'); fi
    for c in "affine f32 fill" "elastic f32 fill" "affine+elastic f32 nofill"; do
      timeout 100 $B/resample_bench --cases perf --reps 20 --path $P --case "$c" 2>&1 | grep -E " $P " | sed "s/^/$tag r$r  /" >> $O
    done
  done
done
unset LD_LIBRARY_PATH
python - <<PY
import re, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
order = []
for line in open("$O"):
    m = re.match(r"(\S+) r\d+\s+(.*?)\s{2,}$P\s+([0-9.]+) ms", line)
    if m:
        rows[m.group(2)][m.group(1)].append(float(m.group(3)))
        if m.group(1) not in order: order.append(m.group(1))
for case, d in rows.items():
    print(case)
    for tag in order:
        v = d.get(tag, [])
        if v: print(f"   {tag:10s} min {min(v):.3f}  ({' '.join(f'{x:.3f}' for x in v)})")
PY
