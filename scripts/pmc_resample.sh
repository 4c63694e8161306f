# PMC passes over tests/native/resample_bench (run on the GPU box through gpurun); $1 = output tag, $2 = path filter, $3 = case filter
# (counters in their own runs with --kernel-trace only, as the guide prescribes; FETCH_SIZE / WRITE_SIZE cannot share a pass)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; B=$R/tests/native/_build/resample_bench; O=$R/gpurun_out/pmc_$1; mkdir -p $O
P=${2:-fast}; C=${3:-affine f32 fill}
run() { timeout 300 rocprofv3 --kernel-trace --pmc $2 -d $O -o $1 --output-format csv -- $B --cases perf --reps 2 --case "$C" --path $P > $O/$1.log 2>&1; }
run w "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run i "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
run v "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"
run f "FETCH_SIZE"
run x "WRITE_SIZE"
ls $O | head -3
