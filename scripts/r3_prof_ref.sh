cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r3_final; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o reference --output-format csv -- python $R/bench.py --steps 20 --noise-rng reference --resample-precision exact --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix > $O/reference.log 2>&1
head -8 $O/reference_kernel_stats.csv | cut -c1-160
