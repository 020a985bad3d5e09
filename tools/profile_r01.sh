set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof_v6
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v6/stats -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_v6/bench_stats.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_v6/pmc_fetch -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_v6/bench_fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_v6/pmc_write -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_v6/bench_write.log 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/prof_v6/pmc_sq -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_v6/bench_sq.log 2>&1
find $R/gpurun_out/prof_v6 -name "*.csv" | head -20
tail -2 $R/gpurun_out/prof_v6/bench_stats.log | cut -c1-300
cd $R && timeout 200 python bench.py > $R/gpurun_out/prof_v6/bench_line.json 2> $R/gpurun_out/prof_v6/bench_line.err; tail -c 600 $R/gpurun_out/prof_v6/bench_line.json
