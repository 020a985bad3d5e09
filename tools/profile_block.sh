# rocprofv3 kernel stats of one fused MSN encoder layer (forward-only and forward+backward passes) + the layer benchmark with
# kernel names and launch counts per layer (torch profiler); outputs under gpurun_out/prof_block.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_block
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT && mkdir -p $OUT
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o blk -- python $R/tools/bench_block.py --mode fused --no-launch-count --iters 10 > $OUT/rocprof.log 2>&1
cd $R
timeout 200 python tools/bench_block.py > $OUT/bench_block.txt 2>&1
grep "us/layer\|kernels per layer" $OUT/bench_block.txt
