# A/B of the FAST paths on the GPU box (run through gpurun from the repository root):
#   bash scripts/r2_fast_ablate.sh [path]      path = fast (planned bricks, default) | fast-brick
# (round 2's other experimental kernels — lean, stream, pipe, ring — live in commit 45474dd)
cd tests/native/_build
B=./resample_bench
K=${1:-fast}
timeout 150 $B --cases parity --path $K 2>&1 | grep -v "^dtype" | grep -v "mismatch vs gather: 0 .*vs oracle: 0" | tail -12
run() { echo "== $*"; timeout 60 env "$@" $B --cases perf --reps 20 --path ${P:-fast} 2>&1 | grep -E "fast" | grep -E "(affine f32 fill|elastic f32 fill|nofill)" | cut -c1-130; }
P="fast" run TIO_X=0
P="$K" run TIO_X=0
P="$K" run TIO_TILE_ABLATE=1
P="$K" run TIO_TILE_ABLATE=2
P="$K" run TIO_TILE_ABLATE=3
