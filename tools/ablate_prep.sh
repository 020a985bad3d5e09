# timing-only ablation of the K/V pre-pass inside the step (results meaningless): what would a pre-pass without its VALU work buy?
#   full = the product library; abl1 = no rho transform (unpack / pack / |k'|^2 kept); abl2 = abl1 + the V rows go out as they came in
# built by: hipcc ... -DGTA_PREP_ABL={1,2} -c gta_prep.hip, linked with the other objects into libgta_hip_abl{1,2}.so (dev_log.md, call 39)
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do
for v in full abl1 abl2; do
LIB=$R/gta_amd/csrc/libgta_hip.so; [ $v != full ] && LIB=$R/gta_amd/csrc/libgta_hip_$v.so
GTA_HIP_LIB=$LIB timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$v$i step %.1f us kernel %.1f us rest %.1f us sclk %.0f' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, (d['ms_per_step']-r['kernel_ms'])*1e3, r.get('sclk_mhz') or 0))"
done
done
