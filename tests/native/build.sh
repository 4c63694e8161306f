#!/bin/bash
# Builds tests/native/_build/resample_bench (C++ host driver of the C ABI; links the HIP
# library and the CPU oracle).  Test infrastructure — see resample_bench.cpp.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
make -s -C "$ROOT/torchio_amd/csrc"
make -s -C "$ROOT/oracle"
mkdir -p "$HERE/_build"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 "$HERE/resample_bench.cpp" -o "$HERE/_build/resample_bench" \
  -L"$ROOT/torchio_amd/csrc" -ltio_hip -L"$ROOT/oracle" -ltio_oracle -ldl \
  -Wl,-rpath,'$ORIGIN/../../../torchio_amd/csrc' -Wl,-rpath,'$ORIGIN/../../../oracle'
for tool in valu_rates dpp_check lds_rates; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 "$HERE/$tool.cpp" -o "$HERE/_build/$tool"
done
echo "built $HERE/_build/resample_bench"
