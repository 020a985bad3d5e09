# round 4, GPU call 3: fp32-faithful gradients (gta_plain32.hip), lane-parallel view-rep builder, bench lines
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c3; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_precise.py -q > $OUT/precise_tests.log 2>&1; echo "precise tests rc=$?" | tee -a $OUT/summary.txt
tail -25 $OUT/precise_tests.log | cut -c1-220
timeout 300 python -m pytest tests/test_gpu_reps.py tests/test_gpu_variants.py -q > $OUT/reps_tests.log 2>&1; echo "reps/variants tests rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/reps_tests.log
timeout 100 python tools/time_reps.py 2>&1 | tail -1 | tee -a $OUT/summary.txt
timeout 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 10 > $OUT/bench_msenc.json 2>>$OUT/bench.err
timeout 200 python bench.py --workload cl-enc --dtype f32 --precise --steps 30 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 10 > $OUT/bench_clenc_precise.json 2>>$OUT/bench.err
timeout 200 python bench.py --workload cl-enc --dtype f32 --steps 30 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 10 > $OUT/bench_clenc_f32.json 2>>$OUT/bench.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04c3/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d.get("roofline", {})
        print(os.path.basename(f), "value %.1f ms %.4f kernel %s %.1f us | fwd_bwd %s | parity %s" % (d["value"], d["ms_per_step"], r.get("kernel"), (r.get("kernel_ms") or 0) * 1e3, d.get("fwd_bwd"), d.get("parity")))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -5 $OUT/bench.err
