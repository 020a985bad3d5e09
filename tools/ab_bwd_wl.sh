# A/B of forward + backward at the other BASELINE workloads (compiled backward kernels): libgta_hip_prev.so against libgta_hip.so, alternating
R=$GRAFT_REPO_ROOT; cd $R
for w in ${AB_WORKLOADS:-cl-enc cl-dec dit}; do
for i in 1 2; do
for v in prev new; do
LIB=$R/gta_amd/csrc/libgta_hip.so; [ $v = prev ] && LIB=$R/gta_amd/csrc/libgta_hip_prev.so
GTA_HIP_LIB=$LIB timeout 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w $v$i fwd+bwd %.4f ms' % d['fwd_bwd']['ms_per_step'])"
done
done
done
