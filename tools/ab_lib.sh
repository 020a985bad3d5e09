# A/B of two library builds on the headline bench line (sustained protocol), alternating: LIBS="libgta_hip.so libgta_hip_x.so" bash tools/ab_lib.sh
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for lib in ${LIBS:-libgta_hip.so libgta_hip_prev.so}; do
GTA_HIP_LIB=$PWD/gta_amd/csrc/$lib python bench.py --gpus 1 --steps 50 --warmup 5 ${AB_ARGS} --block-steps 0 --no-cpu-baseline --workloads none --train-steps 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', 'value %.1f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernel %.1f us' % (r['kernel_ms']*1e3), 'frac %.3f' % r['frac'], 'MHz %.0f' % r['sclk_mhz'], 'cycles %.1fk' % (r['kernel_cycles']/1e3), 'parity %.2e' % d['parity']['parity_max_abs'])"
done; done
