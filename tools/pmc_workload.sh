# PMC passes (FETCH_SIZE, WRITE_SIZE, two SQ passes; never combined with other trace domains) of bench.py at one workload:
#   WL=cl-dec bash tools/pmc_workload.sh   -> gpurun_out/pmc_$WL/{fetch,write,sq,sq2}, summary in gpurun_out/pmc_$WL/summary_kernels.json
R=$GRAFT_REPO_ROOT
WL=${WL:-cl-dec}
OUT=$R/gpurun_out/pmc_$WL
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
B="--workload $WL --workloads none --no-cpu-baseline --no-parity --block-steps 0 --train-steps 0 --steps 4 --warmup 2 --precondition-s 0"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --workload $WL --workloads none --no-cpu-baseline --no-parity --block-steps 0 --train-steps 0 --steps 20 --warmup 5 --precondition-s 0.6 > $OUT/stats.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py $B > $OUT/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $R/bench.py $B > $OUT/write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/pmc_sq -- python $R/bench.py $B > $OUT/sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python $R/bench.py $B > $OUT/sq2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_TRANS --kernel-trace --output-format csv -d $OUT/pmc_sq3 -- python $R/bench.py $B > $OUT/sq3.log 2>&1
cd $R
python tools/summarize_prof.py gpurun_out/pmc_$WL gpurun_out/pmc_$WL/summary
python - <<PY
import json
k = json.load(open("gpurun_out/pmc_$WL/summary_kernels.json"))
for n, d in k.items():
    if "SQ_BUSY_CYCLES" in d:
        simd = d["SQ_BUSY_CYCLES"] / 32.0
        print(n, {kk: round(v / (1024.0 * simd), 3) for kk, v in d.items() if kk.startswith("SQ_") and ("CYCLES" in kk or "ACTIVE" in kk or "WAIT" in kk)}, "kernel cycles", round(simd))
PY
