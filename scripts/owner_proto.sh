#!/bin/bash
# GPU (gpurun, repository root): the owner-brick prototype (tests/native/owner_proto.hip) next to the shipped lean kernel, same box.
# Builds its variants first (hipcc on the GPU box: seconds): shapes BZ_GROUP and the ablations of the 28_32 shape.
B=tests/native/_build; mkdir -p $B
for v in "28 32" "14 16" "60 64"; do set -- $v; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DOWNER_BZ=$1 -DOWNER_GROUP=$2 tests/native/owner_proto.hip -o $B/owner_proto_$1_$2 2>/dev/null; done
for ab in 1 2 4; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DOWNER_ABLATE=$ab tests/native/owner_proto.hip -o $B/owner_proto_ab$ab 2>/dev/null; done
cd $B
for v in owner_proto_28_32 owner_proto_14_16 owner_proto_60_64 owner_proto_ab1 owner_proto_ab2 owner_proto_ab4; do echo "== $v"; timeout 120 ./$v 256 8 20 2>&1 | tail -2; done
echo "== shipped"
timeout 200 ./resample_bench --cases perf --case "affine f32" --path "fast" --reps 20 2>&1 | grep " ms " | grep -v "fast-brick\|fast-general\|gather  " | cut -c1-130
