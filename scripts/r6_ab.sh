#!/bin/bash
# Same-box A/B on the GPU box: tests/native/resample_bench perf cases with the working tree's library and with the baseline
# (tests/native/_build/base/libtio_hip.so, scripts/r6_build_base.sh), alternating, ROUNDS times.  $1 = path filter, $2 = tag
P=${1:-tight}; TAG=${2:-ab}; ROUNDS=${3:-2}
B=tests/native/_build
for r in $(seq $ROUNDS); do
  for lib in new base; do
    if [ $lib = base ]; then export LD_LIBRARY_PATH=$PWD/$B/base; else unset LD_LIBRARY_PATH; fi
    timeout 200 $B/resample_bench --cases perf --reps 20 --path $P 2>&1 | grep -E " $P " | sed "s/^/$lib r$r  /" >> gpurun_out/r6_$TAG.txt
  done
done
unset LD_LIBRARY_PATH
python - <<PY
import re, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for line in open("gpurun_out/r6_$TAG.txt"):
    m = re.match(r"(new|base) r\d+\s+(.*?)\s{2,}$P\s+([0-9.]+) ms", line)
    if m: rows[m.group(2)][m.group(1)].append(float(m.group(3)))
for case, d in rows.items():
    n, b = d.get("new", []), d.get("base", [])
    if n and b: print(f"{case:36s} new {min(n):.3f} ({' '.join(f'{x:.3f}' for x in n)})  base {min(b):.3f} ({' '.join(f'{x:.3f}' for x in b)})  {100 * (min(n) / min(b) - 1):+.1f} %")
PY
