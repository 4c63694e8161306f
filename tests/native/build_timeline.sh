#!/bin/bash
# Builds tests/native/_build/timeline/libtio_hip.so: the library with the exact-coordinate kernel's block timeline compiled in
# (-DTIO_LE_TIMELINE, csrc/resample_lean_exact.hpp).  Measurement infrastructure: put it in the place of the production library
# on the GPU box (cp over torchio_amd/csrc/libtio_hip.so of the box's scratch copy) and run resample_bench --cases perf.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
make -s -C "$ROOT/torchio_amd/csrc"
mkdir -p "$HERE/_build/timeline"
cd "$ROOT/torchio_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -DTIO_LE_TIMELINE \
  -c resample.hip -o "$HERE/_build/timeline/resample.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o "$HERE/_build/timeline/libtio_hip.so" api.o "$HERE/_build/timeline/resample.o" \
  intensity.o aggregate.o interpolate.o kspace.o labels.o mt19937.o host_rng.o host_rng_jump.o
echo "built $HERE/_build/timeline/libtio_hip.so"
