# round 3, final: rocprofv3 kernel statistics of (1) the headline bench command, (2) the reference-identical mode, (3) config 5
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r3_final; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o headline --output-format csv -- python $R/bench.py --steps 20 --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix > $O/headline.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o reference --output-format csv -- python $R/bench.py --steps 20 --noise-rng reference --resample-precision exact --no-cpu-baseline --no-aten-baseline --no-other-configs --no-mode-matrix > $O/reference.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o others --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --prewarm 5 --no-cpu-baseline --no-aten-baseline --no-mode-matrix > $O/others.log 2>&1
ls $O | head -20
for f in headline reference others; do echo "== $f"; head -14 $O/${f}_kernel_stats.csv | cut -c1-200; done
