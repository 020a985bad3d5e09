# round 4, GPU call 13: backward with accumulators started at -lse2 / -D, tile loops unrolled by ring stage: whole GPU suite, kernel times
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c13; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; echo "gpu suite rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/gpu_tests.log
bash tools/prof_bwd.sh > $OUT/prof_bwd.log 2>&1; tail -8 $OUT/prof_bwd.log
for w in ms-enc ms-dec cl-enc cl-dec dit; do
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 10 --workload $w 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w', 'fwd %.4f ms' % d['ms_per_step'], 'fwd_bwd %.4f ms' % d['fwd_bwd']['ms_per_step'])"
done
