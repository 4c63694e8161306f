# round 3: which shared resource is busy?  memory-path and dispatch counters of the lean planned kernel (affine bench launch)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; B=$R/tests/native/_build/resample_bench; O=$R/gpurun_out/pmc_r3; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
P=${1:-fast-lean-nox4}; C=${2:-affine f32 fill}
run() { timeout 200 rocprofv3 --kernel-trace --pmc $2 -d $O -o $1 --output-format csv -- $B --cases perf --reps 2 --case "$C" --path $P > $O/$1.log 2>&1; }
run w "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run m "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
run t "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
run t2 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"
run c "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
run c2 "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_REQ_sum"
run s "SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN"
python $R/scripts/pmc_summary.py $O planned > $O/summary.txt 2>&1
cat $O/summary.txt | head -120
