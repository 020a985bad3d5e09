# round 4, GPU call 4: precise-mode tests (euclid fix, whole-model CLEVR fixture), kernel stats of the bench step
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c4; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_precise.py tests/test_gpu_modules.py -q > $OUT/tests.log 2>&1; echo "precise + module tests rc=$?" | tee -a $OUT/summary.txt
tail -30 $OUT/tests.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 10 > $OUT/bench_stats.log 2>&1
cd $R
f=$(ls $OUT/stats/*/*kernel_stats.csv | head -1); head -12 $f | cut -d, -f1-6 | cut -c1-170
python tools/step_gaps.py gpurun_out/r04c4/stats 2>&1 | tail -12
