#!/bin/bash
# GPU (gpurun, repository root): the reference-noise modes with the draw stream at the device's lowest priority (the default
# of ops.draw_stream) and at torch's default priority — one PROCESS per variant (a new stream has its own allocator pool).
L=gpurun_out/r4b_ab_draw_stream.log; : > $L
for prio in low default; do
  echo "== TIO_DRAW_STREAM_PRIORITY=$prio" >> $L
  TIO_DRAW_STREAM_PRIORITY=$prio timeout 200 python scripts/bench_reference_noise.py --modes "reference,fast;reference,exact" --rounds 2 2>&1 | grep -v amdgpu >> $L
done
cat $L
