cd $GRAFT_REPO_ROOT
for rep in 1 2; do for m in ${MODES:-prepass prepass_rows32}; do
python bench.py --workload dit --kv-mode $m --no-cpu-baseline --block-steps 0 --train-steps 0 --workloads none --no-parity --steps 50 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d.get('roofline') or {}
print('dit $m', 'ms/step %.4f' % d['ms_per_step'], r.get('kernel'), 'kernel_ms %.4f' % r.get('kernel_ms'), 'frac %.3f' % r['frac'], 'MHz %.0f' % r['sclk_mhz'], 'cycles %.1fk' % (r['kernel_cycles']/1e3), 'cold %.4f' % d['cold_start']['ms_per_step'])"
done; done
