# round 4, GPU call 15: dK/dV statistics through registers + DPP row broadcast instead of LDS
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c15; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_backward.py -x -q > $OUT/bwd_tests.log 2>&1; echo "backward tests rc=$?" | tee -a $OUT/summary.txt
tail -2 $OUT/bwd_tests.log
bash tools/prof_bwd.sh > $OUT/prof_bwd.log 2>&1; tail -8 $OUT/prof_bwd.log | cut -c1-110
grep -o '"fwd_bwd": {"ms_per_step": [0-9.]*' gpurun_out/prof_bwd/bwd.log
for w in ms-enc ms-dec cl-dec dit; do
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 20 --workload $w 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w', 'fwd %.4f ms' % d['ms_per_step'], 'fwd_bwd %.4f ms' % d['fwd_bwd']['ms_per_step'])"
done
