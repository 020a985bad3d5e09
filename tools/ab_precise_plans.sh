cd $GRAFT_REPO_ROOT
for wl in cl-enc cl-dec dit; do for m in prepass fused; do
python bench.py --workload $wl --dtype f32 --precise --kv-mode $m --no-cpu-baseline --block-steps 0 --train-steps 0 --workloads none 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d.get('roofline') or {}
print('$wl $m', 'ms/step %.4f' % d['ms_per_step'], 'kernel', r.get('kernel'), 'kernel_ms', r.get('kernel_ms'), 'parity', d['parity']['parity_max_abs'], d['parity']['rel_rms'])"
done; done
