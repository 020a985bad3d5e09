# kernel times of the operator's forward + backward under rocprofv3
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_bwd; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bwd -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 10 > $OUT/bwd.log 2>&1
f=$(ls $OUT/bwd/*/*kernel_stats.csv | head -1)
head -8 $f | cut -d, -f1-5 | cut -c1-150
# the same with the generated dQ and dK/dV kernels as two launches (GTA_FLAG_BWD_SPLIT): their separate times
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bwd_split -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 10 --kv-mode prepass_bwd_keys64_split > $OUT/bwd_split.log 2>&1
f=$(ls $OUT/bwd_split/*/*kernel_stats.csv | head -1)
head -8 $f | cut -d, -f1-5 | cut -c1-150
grep -o '"fwd_bwd": {"ms_per_step": [0-9.]*' $OUT/bwd.log $OUT/bwd_split.log
