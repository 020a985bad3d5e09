# A/B of the fp32-faithful mode (rho kernels of gta_apply.hip): libgta_hip_prev.so against libgta_hip.so, cl-enc fp32, forward and forward + backward
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2; do
for v in prev new; do
LIB=$R/gta_amd/csrc/libgta_hip.so; [ $v = prev ] && LIB=$R/gta_amd/csrc/libgta_hip_prev.so
GTA_HIP_LIB=$LIB timeout 300 python bench.py --workload cl-enc --dtype f32 --precise --block-steps 0 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v$i fwd %.1f us  fwd+bwd %.3f ms  parity %.2e' % (d['ms_per_step']*1e3, d['fwd_bwd']['ms_per_step'], (d.get('parity') or {}).get('parity_max_abs', -1)))"
done
done
