# PMC passes over tests/native/resample_bench (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; B=$R/tests/native/_build/resample_bench; O=$R/gpurun_out/pmc4; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $O -o i0 --output-format csv -- $B --cases perf --reps 2 --case "f32 fill" --path tile16x16x16 > $O/i0.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O -o w0 --output-format csv -- $B --cases perf --reps 2 --case "f32 fill" --path tile16x16x16 > $O/w0.log 2>&1
