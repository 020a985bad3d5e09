# A/B of the backward at the smaller workloads: previous library vs this one
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2; do
for v in prev new; do
LIB=$R/gta_amd/csrc/libgta_hip.so; [ $v = prev ] && LIB=$R/gta_amd/csrc/libgta_hip_prev.so
for w in cl-enc cl-dec dit; do
GTA_HIP_LIB=$LIB timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 20 --workload $w 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v $w', 'fwd %.4f ms' % d['ms_per_step'], 'fwd_bwd %.4f ms' % d['fwd_bwd']['ms_per_step'])"
done
done
done
