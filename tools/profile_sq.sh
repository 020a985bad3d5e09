# extra SQ counter passes for the attention kernel (two passes of 8 counters)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof_sq
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq/pmc_a -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_sq/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq/pmc_b -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_sq/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq/pmc_c -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_sq/c.log 2>&1
find $R/gpurun_out/prof_sq -name "*counter_collection.csv"
