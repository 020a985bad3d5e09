# round 4, GPU call 1: the item stream on hardware -- parity first, then the whole GPU suite, then A/B timings
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c1; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_attn64.py -x -q -k "item_stream" > $OUT/items_tests.log 2>&1; echo "items tests rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/items_tests.log
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "gpu suite rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/gpu_tests.log
for mode in prepass prepass_item_cxx prepass prepass_item_cxx; do
  timeout 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 0 --kv-mode $mode > $OUT/bench_$mode.$RANDOM.json 2>>$OUT/bench.err
done
GTA_HIP_LIB=$R/gta_amd/csrc/libgta_hip_diag.so GTA_ATTN64_VARIANT=4 timeout 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 0 > $OUT/bench_pad4.json 2>>$OUT/bench.err
GTA_HIP_LIB=$R/gta_amd/csrc/libgta_hip_diag.so timeout 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 0 > $OUT/bench_diag_pad0.json 2>>$OUT/bench.err
timeout 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 0 --workload ms-dec > $OUT/bench_msdec.json 2>>$OUT/bench.err
timeout 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 0 --workload ms-dec --kv-mode prepass_item_cxx > $OUT/bench_msdec_cxx.json 2>>$OUT/bench.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04c1/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d.get("roofline", {})
        print(os.path.basename(f), "value %.1f ms %.4f kernel %s %.1f us cyc %s sclk %s frac %.3f busy %s parity %s" % (d["value"], d["ms_per_step"], r.get("kernel"), (r.get("kernel_ms") or 0) * 1e3, r.get("kernel_cycles"), r.get("sclk_mhz"), r.get("frac") or 0, r.get("mfma_busy"), (d.get("parity") or {}).get("parity_max_abs")))
    except Exception as e:
        print(f, "unreadable", e)
PY
