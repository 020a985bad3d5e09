# round 4, GPU call 5: rep build one batch ahead on a side stream; non-temporal O stores (diagnostic variant)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c5; rm -rf $OUT; mkdir -p $OUT
cd $R
B="--steps 30 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 0"
for i in 1 2; do
timeout 200 python bench.py $B --rep-stream side > $OUT/bench_side.$i.json 2>>$OUT/bench.err
timeout 200 python bench.py $B --rep-stream main > $OUT/bench_main.$i.json 2>>$OUT/bench.err
GTA_HIP_LIB=$R/gta_amd/csrc/libgta_hip_diag.so GTA_ATTN64_VARIANT=4 timeout 200 python bench.py $B --rep-stream main > $OUT/bench_nt_main.$i.json 2>>$OUT/bench.err
done
timeout 200 python bench.py $B --workload ms-dec > $OUT/bench_msdec_side.json 2>>$OUT/bench.err
timeout 200 python bench.py $B --workload cl-enc > $OUT/bench_clenc_side.json 2>>$OUT/bench.err
timeout 200 python bench.py $B --workload cl-enc --rep-stream main > $OUT/bench_clenc_main.json 2>>$OUT/bench.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04c5/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d.get("roofline", {})
        print(os.path.basename(f), "value %.1f ms %.4f serial %s kernel %s %.1f us cyc %.0f sclk %.0f frac %.3f parity %s" % (d["value"], d["ms_per_step"], d.get("serial_ms_per_step"), r.get("kernel"), (r.get("kernel_ms") or 0) * 1e3, r.get("kernel_cycles") or 0, r.get("sclk_mhz") or 0, r.get("frac") or 0, (d.get("parity") or {}).get("parity_max_abs")))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 $OUT/bench.err
