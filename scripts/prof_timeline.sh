#!/bin/bash
# kernel + memory-copy timeline of the headline bench step (gpurun): gpurun_out/r4_timeline/
R=$PWD; O=$R/gpurun_out/r4_timeline; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O -o tl --output-format csv -- python $R/bench.py --steps 20 --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix > $O/tl.log 2>&1
tail -c 300 $O/tl.log; ls $O
