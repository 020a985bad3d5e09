# round 4, GPU call 12: backward -- accumulators start at -lse2 / -D (no subtractions), 64-keys-per-wave dK/dV (GTA_BWD_DKV2)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c12; rm -rf $OUT; mkdir -p $OUT
cd $R
for v in 0 1; do
GTA_BWD_DKV2=$v timeout 600 python -m pytest tests/test_gpu_backward.py -x -q > $OUT/bwd_tests.$v.log 2>&1; echo "backward tests (dkv2=$v) rc=$?" | tee -a $OUT/summary.txt
tail -2 $OUT/bwd_tests.$v.log
done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
GTA_BWD_DKV2=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$v -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 10 > $OUT/prof$v.log 2>&1
f=$(ls $OUT/prof$v/*/*kernel_stats.csv | head -1)
echo "dkv2=$v"; head -6 $f | cut -d, -f1-4 | cut -c1-120
grep -o '"fwd_bwd": {"ms_per_step": [0-9.]*' $OUT/prof$v.log
done
