cd tests/native/_build
for ab in 0 64; do echo "ablate $ab"; timeout 200 ./resample_bench --cases perf --path "fast" --reps 20 --ablate $ab 2>&1 | grep " ms " | grep -v "brick\|general\|subject\|b1" | cut -c1-150; done
