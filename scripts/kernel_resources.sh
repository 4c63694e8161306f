#!/bin/bash
# Registers, spills and LDS of the gfx950 kernels in an object file (default: the resampler): the code object's own metadata.
#   scripts/kernel_resources.sh [object] [name filter] [mangled kernel name: instruction histogram] [lines]
OBJ=${1:-torchio_amd/csrc/resample.o}; FILTER=${2:-lean_exact}
TMP=$(mktemp -d); trap 'rm -rf $TMP' EXIT
cp "$OBJ" $TMP/in.o && (cd $TMP && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading in.o > /dev/null) && mv $TMP/in.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 $TMP/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/dev.co | awk -v f="$FILTER" '
  /\.name:/ {name=$2} /\.sgpr_count:/ {s=$2} /\.sgpr_spill_count:/ {ss=$2} /\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {vs=$2}
  /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2}
  /\.wavefront_size:/ { if (name ~ f) printf "%-90s vgpr %3d (spill %d) sgpr %3d (spill %d) scratch %d lds %d\n", name, v, vs, s, ss, p, g }'
if [ -n "$3" ]; then
  /opt/rocm/lib/llvm/bin/llvm-objdump -d --disassemble-symbols=$3 $TMP/dev.co | awk '/^\s+[vsdgb][_a-z0-9]+ /{print $1}' | sort | uniq -c | sort -rn | head -${4:-40}
fi
