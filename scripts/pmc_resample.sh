# PMC passes over tests/native/resample_bench (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; B=$R/tests/native/_build/resample_bench; O=$R/gpurun_out/pmc3; mkdir -p $O
for lds in 12160 6144 9000; do
  echo "== timing lds $lds"; $B --cases perf --reps 10 --case "f32 fill" --lds $lds | grep -E "tile|gather"
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $O -o w${lds} --output-format csv -- $B --cases perf --reps 1 --case "f32 fill" --lds $lds > $O/w${lds}.log 2>&1
done
