# round 4, GPU call 8: epilogue memory instructions woven into its arithmetic: parity, phases, bench
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c8; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_attn64.py -x -q > $OUT/attn64_tests.log 2>&1; echo "attn64 tests rc=$?" | tee -a $OUT/summary.txt
tail -2 $OUT/attn64_tests.log
GTA_HIP_LIB=$R/gta_amd/csrc/libgta_hip_diag.so GTA_ATTN64_VARIANT=5 timeout 200 python tools/check_attn64.py phases ms-enc 2>&1 | grep "attn64" | tee $OUT/phases_items.txt
for i in 1 2 3; do
timeout 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 0 > $OUT/bench_items.$i.json 2>>$OUT/bench.err
done
timeout 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 0 --workload ms-dec > $OUT/bench_msdec.json 2>>$OUT/bench.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04c8/bench_*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1]); r = d.get("roofline", {})
    print(os.path.basename(f), "value %.1f ms %.4f kernel %.1f us cyc %.0f sclk %.0f frac %.3f fgc %.3f busy %.3f" % (d["value"], d["ms_per_step"], (r.get("kernel_ms") or 0) * 1e3, r.get("kernel_cycles") or 0, r.get("sclk_mhz") or 0, r.get("frac") or 0, r.get("frac_at_granted_clock") or 0, r.get("mfma_busy") or 0))
PY
