#!/bin/bash
# GPU (gpurun): the marching stencil kernels with 2 (shipped) / 3 / 4 rows of loads in flight per lane (TIO_MARCH_AHEAD, compile time):
# libraries built ahead into tests/native/_build/ahead{2,3,4}/, swapped in one after the other.
L=gpurun_out/r4i_march_ahead.log; : > $L
for rep in 1 2; do for n in 2 3 4 2; do
  cp tests/native/_build/ahead$n/libtio_hip.so torchio_amd/csrc/libtio_hip.so
  echo "== TIO_MARCH_AHEAD=$n (rep $rep)" >> $L
  for p in fast exact; do timeout 100 python scripts/bench_blur_stages.py 6 $p 2>&1 | grep "precision\|bias + blur + noise\|explicit" >> $L; done
done; done
cp tests/native/_build/ahead2/libtio_hip.so torchio_amd/csrc/libtio_hip.so
cat $L
