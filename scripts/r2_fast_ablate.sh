cd tests/native/_build
B=./resample_bench
timeout 100 $B --cases parity --path fast-s 2>&1 | grep -v "^dtype" | grep -v "mismatch vs gather: 0 .*vs oracle: 0" | tail -20
run() { echo "== $*"; timeout 60 env "$@" $B --cases perf --reps 20 --path ${P:-fast-s} 2>&1 | grep -E "fast" | grep -E "(affine f32 fill|elastic f32 fill)" | cut -c1-130; }
run TIO_STREAM_BPC=2
run TIO_STREAM_BPC=1
