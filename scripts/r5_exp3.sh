#!/bin/bash
# round 5, run 3: the GPU suite on the new tree (stop at the 5th failure), smoke
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/r5_gpu_tests3.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r5_gpu_tests3.txt
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r5_gpu_tests3.txt | tail -20
ls gpurun_out/*parity*.json 2>/dev/null; for f in gpurun_out/tight_parity_*.json gpurun_out/headline_parity_*.json gpurun_out/fast_precision_256.json; do echo "$f: $(cat $f)"; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
