# A/B of the forward step: libgta_hip_prev.so against libgta_hip.so, alternating on one box
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do
for v in prev new; do
LIB=$R/gta_amd/csrc/libgta_hip.so; [ $v = prev ] && LIB=$R/gta_amd/csrc/libgta_hip_prev.so
GTA_HIP_LIB=$LIB timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 0 ${AB_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$v$i value %.1f ms %.4f kernel %.1f us sclk %.0f parity %.2e' % (d['value'], d['ms_per_step'], r['kernel_ms']*1e3, r.get('sclk_mhz') or 0, (d.get('parity') or {}).get('parity_max_abs', -1)))"
done
done
