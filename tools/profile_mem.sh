# VMEM / LDS in-flight level counters of the bench kernels (one --pmc pass, kernel-trace only).
# NOTE: passes with TCP_* / TA_* / TCC_* counters abort inside rocprofv3 on this image (signal 6 after the first
# kernels) and then sit until the timeout -- do not add them back without a short timeout.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_mem
mkdir -p $O
CMD="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline"
timeout 120 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_a -- $CMD > $O/a.log 2>&1
find $O -name "*counter_collection.csv" | head
