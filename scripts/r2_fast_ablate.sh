cd tests/native/_build
B=./resample_bench
run() { echo "== $*"; timeout 60 env "$@" $B --cases perf --reps 20 --path ${P:-fast} 2>&1 | grep -E "fast" | grep -E "(affine f32 fill|elastic f32 fill)" | cut -c1-130; }
P="fast " run TIO_X=0
P="fast " run TIO_TILE_ABLATE=1
P="fast " run TIO_TILE_ABLATE=2
P="fast " run TIO_TILE_ABLATE=3
