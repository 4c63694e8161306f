#!/bin/bash
# Builds tests/native/_build/base/libtio_hip.so from the csrc/ + include/ of a git ref (default HEAD): the baseline of a
# same-box A/B (scripts/r6_ab.sh runs both libraries alternately in ONE gpurun call).  Objects other than resample.o are
# reused from the working tree when their sources are unchanged against the ref.
set -e
REF=${1:-HEAD}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
W=$(mktemp -d); trap 'rm -rf $W' EXIT
mkdir -p $W/torchio_amd $W/include "$ROOT/tests/native/_build/base"
git -C "$ROOT" archive $REF torchio_amd/csrc include | tar -x -C $W
cd $W/torchio_amd/csrc
for o in "$ROOT"/torchio_amd/csrc/*.o; do
  src=$(basename ${o%.o}); f=$src.hip; [ -f $f ] || f=$src.cpp
  if [ $src != resample ] && git -C "$ROOT" diff --quiet $REF -- torchio_amd/csrc/$f torchio_amd/csrc/common.hpp include/tio_hip.h; then cp $o . && touch $src.o; fi
done
make -s libtio_hip.so 2>&1 | grep -E "error" || true
cp libtio_hip.so "$ROOT/tests/native/_build/base/libtio_hip.so"
echo "built tests/native/_build/base/libtio_hip.so from $REF"
