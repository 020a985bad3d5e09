# Profile (ROUND=rNN, default r05) of the bench command on the MI355X box: rocprofv3 kernel stats + trace, then SEPARATE --pmc passes
# (FETCH_SIZE, WRITE_SIZE do not fit one pass; never combined with trace domains other than --kernel-trace).
set -x
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r05}
OUT=$R/gpurun_out/prof_$ROUND
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
B="--no-cpu-baseline --no-parity --block-steps 0 --workloads none"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --steps 20 --warmup 5 --precondition-s 0.6 $B --train-steps 10 > $OUT/bench_stats.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --steps 4 --warmup 2 --precondition-s 0 $B --train-steps 2 > $OUT/bench_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $R/bench.py --steps 4 --warmup 2 --precondition-s 0 $B --train-steps 2 > $OUT/bench_write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/pmc_sq -- python $R/bench.py --steps 4 --warmup 2 --precondition-s 0 $B --train-steps 0 > $OUT/bench_sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python $R/bench.py --steps 4 --warmup 2 --precondition-s 0 $B --train-steps 0 > $OUT/bench_sq2.log 2>&1
cd $R
python tools/step_gaps.py gpurun_out/prof_$ROUND/stats > $OUT/step_gaps.txt 2>&1; cat $OUT/step_gaps.txt
python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; tail -c 2500 $OUT/bench_line.json
python tools/summarize_prof.py gpurun_out/prof_$ROUND gpurun_out/prof_$ROUND/summary
