# round 4, GPU call 6: where the fp32-faithful fwd+bwd leg spends its time (rocprofv3 kernel stats)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c6; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --workload cl-enc --dtype f32 --precise --steps 5 --warmup 2 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 10 > $OUT/bench.log 2>&1
f=$(ls $OUT/stats/*/*kernel_stats.csv | head -1); head -24 $f | cut -d, -f1-5 | cut -c1-200
