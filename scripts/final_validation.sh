# Round-end validation on the GPU box (through gpurun from the repository root).  EVERY step runs under `timeout`: a box
# whose GPU faults (seen once: every process died with "Memory access fault" and a profiler run then hung until gpurun's
# own limit, 30 GPU-minutes) must cost seconds, not the budget.
#   bash scripts/final_validation.sh [quick]      quick = smoke + default bench only
timeout 90 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
if [ "$1" != "quick" ]; then
  timeout 300 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
  (cd tests/native/_build && timeout 200 ./resample_bench --cases parity 2>&1 | tail -1)
fi
timeout 420 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -c 300 gpurun_out/final_bench.json; echo
