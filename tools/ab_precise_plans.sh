# fp32-faithful forward: the two-stage plan (dh <= 64) of two library builds, and the single-kernel plan, alternating on one box
cd $GRAFT_REPO_ROOT
one() { GTA_HIP_LIB=$PWD/gta_amd/csrc/$1 python bench.py --workload $2 --dtype f32 --precise --kv-mode $3 --no-cpu-baseline --block-steps 0 --train-steps 0 --workloads none 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d.get('roofline') or {}
print('$1 $2 $3', 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms', r.get('kernel_ms'), 'cycles', r.get('kernel_cycles'), 'MHz', r.get('sclk_mhz'), 'parity', d['parity']['parity_max_abs'])"; }
for rep in 1 2; do for wl in ${WLS:-cl-enc cl-dec}; do for lib in ${LIBS:-libgta_hip_prev.so libgta_hip.so}; do one $lib $wl prepass; done; done; done
