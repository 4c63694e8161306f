# PMC passes over tests/native/resample_bench (run on the GPU box through gpurun); $1 = output tag, $2 = path filter, $3 = case filter
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; B=$R/tests/native/_build/resample_bench; O=$R/gpurun_out/pmc_$1; mkdir -p $O
P=${2:-fast-s16x16}; C=${3:-affine f32 fill}
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O -o w --output-format csv -- $B --cases perf --reps 2 --case "$C" --path $P > $O/w.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O -o i --output-format csv -- $B --cases perf --reps 2 --case "$C" --path $P > $O/i.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_BRANCH -d $O -o m --output-format csv -- $B --cases perf --reps 2 --case "$C" --path $P > $O/m.log 2>&1
ls $O
