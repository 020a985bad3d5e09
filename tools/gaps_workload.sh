# per-step kernel timeline (rocprofv3 kernel trace) of one bench.py workload: tools/gaps_workload.sh <workload>
R=$GRAFT_REPO_ROOT; W=${1:-cl-enc}; OUT=$R/gpurun_out/gaps_$W; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 0 --workload $W > $OUT/bench.log 2>&1
cd $R
python tools/step_gaps.py $OUT/stats | tee $OUT/step_gaps.txt
