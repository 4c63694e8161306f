cd tests/native/_build
B=./resample_bench
timeout 100 $B --cases parity --path fast-pbrick 2>&1 | grep -v "^dtype" | grep -v "mismatch vs gather: 0 .*vs oracle: 0" | tail -12
run() { echo "== $*"; timeout 60 env "$@" $B --cases perf --reps 20 --path ${P:-fast} 2>&1 | grep -E "fast" | grep -E "(affine f32 fill|elastic f32 fill|nofill)" | cut -c1-130; }
P="fast-pbrick" run TIO_X=0
P="fast-pbrick" run TIO_FAST_BPC=2
P="fast-pbrick" run TIO_FAST_BPC=4
P="fast-pbrick" run TIO_TILE_ABLATE=1
P="fast-pbrick" run TIO_TILE_ABLATE=2
P="fast-pbrick" run TIO_TILE_ABLATE=3
