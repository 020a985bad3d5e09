# round 4, GPU call 10: A/B of the relaxed first-item entry (first tiles requested behind the first item's inputs, 12 pieces in flight at
# the first tile loop) against the previous build, alternating on one box
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c10; rm -rf $OUT; mkdir -p $OUT
cd $R
for i in 1 2 3; do
for v in prev new; do
LIB=$R/gta_amd/csrc/libgta_hip.so; [ $v = prev ] && LIB=$R/gta_amd/csrc/libgta_hip_prev.so
GTA_HIP_LIB=$LIB timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 0 > $OUT/bench_$v.$i.json 2>>$OUT/bench.err
done
done
for v in prev new; do
LIB=$R/gta_amd/csrc/libgta_hip.so; [ $v = prev ] && LIB=$R/gta_amd/csrc/libgta_hip_prev.so
GTA_HIP_LIB=$LIB timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 0 --workload ms-dec > $OUT/bench_msdec_$v.json 2>>$OUT/bench.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04c10/bench_*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1]); r = d.get("roofline", {})
    print(os.path.basename(f), "value %.1f ms %.4f kernel %.1f us cyc %.0f sclk %.0f frac %.3f fgc %.3f busy %.3f" % (d["value"], d["ms_per_step"], (r.get("kernel_ms") or 0) * 1e3, r.get("kernel_cycles") or 0, r.get("sclk_mhz") or 0, r.get("frac") or 0, r.get("frac_at_granted_clock") or 0, r.get("mfma_busy") or 0))
PY
