#!/bin/bash
# A/B: the Philox normal transform with the IEEE polynomial sin / cos (the tree) vs the hardware v_sin / v_cos (a second library,
# built on the side from the same sources with that one line changed): the fused stencil stages, 8 x 256^3, fast taps
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for round in 1 2; do
for lib in tree nativetrig; do
python - "$lib" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
from torchio_amd import _lib
if sys.argv[1] != "tree":
    _lib.LIBRARY_PATH = os.path.join(os.getcwd(), "tests", "native", "_build", "libtio_hip_nativetrig.so")
print("library:", _lib.LIBRARY_PATH.split("/")[-1])
sys.argv = ["bench_blur_stages.py", "6", "fast"]
import bench_blur_stages
bench_blur_stages.main()
PY
done; done 2>&1 | grep -v Warning | grep "library\|noise\|blur (I"
