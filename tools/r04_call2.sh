# round 4, GPU call 2: item stream without the loop-head wait (parity again), its phases, power attribution by the loop's ablations
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c2; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_attn64.py -x -q > $OUT/attn64_tests.log 2>&1; echo "attn64 tests rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/attn64_tests.log
for i in 1 2; do
timeout 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 0 > $OUT/bench_items.$i.json 2>>$OUT/bench.err
done
GTA_HIP_LIB=$R/gta_amd/csrc/libgta_hip_diag.so GTA_ATTN64_VARIANT=5 timeout 200 python tools/check_attn64.py phases ms-enc > $OUT/phases_items.txt 2>&1
GTA_HIP_LIB=$R/gta_amd/csrc/libgta_hip_diag.so GTA_ATTN64_ITEMS=0 timeout 200 python tools/check_attn64.py phases ms-enc > $OUT/phases_cxx.txt 2>&1
GTA_HIP_LIB=$R/gta_amd/csrc/libgta_hip_dev.so GTA_ATTN64_ITEMS=0 timeout 400 python tools/check_attn64.py variants 0,3,4,5,6,7 > $OUT/variants_dev.txt 2>&1
cat $OUT/phases_items.txt $OUT/phases_cxx.txt $OUT/variants_dev.txt | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04c2/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d.get("roofline", {})
        print(os.path.basename(f), "value %.1f ms %.4f kernel %s %.1f us cyc %s sclk %s frac %.3f busy %s parity %s" % (d["value"], d["ms_per_step"], r.get("kernel"), (r.get("kernel_ms") or 0) * 1e3, r.get("kernel_cycles"), r.get("sclk_mhz"), r.get("frac") or 0, r.get("mfma_busy"), (d.get("parity") or {}).get("parity_max_abs")))
    except Exception as e:
        print(f, "unreadable", e)
PY
