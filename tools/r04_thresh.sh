# where the generated backward kernels start to pay: fwd+bwd at smaller per-GPU batches, 32-per-wave kernels against the generated ones
R=$GRAFT_REPO_ROOT; cd $R
for B in 4 8 12 16; do
for m in prepass_bwd_keys32 prepass_bwd_keys64; do
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 30 --batch $B --kv-mode $m 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=$B $m', 'fwd_bwd %.4f ms' % d['fwd_bwd']['ms_per_step'])"
done
done
