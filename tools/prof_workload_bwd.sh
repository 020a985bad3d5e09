# kernel times of the operator's forward + backward at one bench workload (rocprofv3 --kernel-trace --stats):  WL=cl-enc bash tools/prof_workload_bwd.sh
R=$GRAFT_REPO_ROOT; WL=${WL:-cl-enc}; TAG=${TAG:-$WL}; OUT=$R/gpurun_out/prof_bwd_$TAG   # (XARGS="--dtype f32 --precise" TAG=cl-enc-f32-faithful: the fp32-faithful mode)
cd /tmp && export TMPDIR=/tmp; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --workload $WL --workloads none --no-cpu-baseline --no-parity --block-steps 0 --steps 20 --warmup 5 --train-steps 20 --precondition-s 0.6 $XARGS > $OUT/bench.log 2>&1
cd $R; python tools/summarize_prof.py gpurun_out/prof_bwd_$TAG gpurun_out/prof_bwd_$TAG/summary | head -20
