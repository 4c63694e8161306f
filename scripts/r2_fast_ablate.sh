# A/B of the experimental FAST kernels on the GPU box (run through gpurun from the repository root):
#   bash scripts/r2_fast_ablate.sh [path]      path = fast-pipe (default) | fast-lean | fast-stream ...
cd tests/native/_build
B=./resample_bench
K=${1:-fast-pipe}
timeout 150 $B --cases parity --path $K 2>&1 | grep -v "^dtype" | grep -v "mismatch vs gather: 0 .*vs oracle: 0" | tail -12
run() { echo "== $*"; timeout 60 env "$@" $B --cases perf --reps 20 --path ${P:-fast} 2>&1 | grep -E "fast" | grep -E "(affine f32 fill|elastic f32 fill|nofill)" | cut -c1-130; }
P="fast" run TIO_X=0
P="$K" run TIO_X=0
P="$K" run TIO_FAST_BPC=2
P="$K" run TIO_TILE_ABLATE=8
P="$K" run TIO_TILE_ABLATE=1
P="$K" run TIO_TILE_ABLATE=2
P="$K" run TIO_TILE_ABLATE=3
P="$K" run TIO_TILE_ABLATE=9
