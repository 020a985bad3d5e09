# round 4, GPU call 7: rep-apply kernels through LDS (generic path + precise mode): parity and timing
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c7; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_variants.py tests/test_gpu_precise.py tests/test_gpu_modules.py -q > $OUT/tests.log 2>&1; echo "variants + precise + modules rc=$?" | tee -a $OUT/summary.txt
tail -8 $OUT/tests.log | cut -c1-250
timeout 200 python bench.py --workload cl-enc --dtype f32 --precise --steps 30 --warmup 10 --no-cpu-baseline --block-steps 0 --train-steps 10 > $OUT/bench_clenc_precise.json 2>>$OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --workload cl-enc --dtype f32 --precise --steps 5 --warmup 2 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 10 > $OUT/bench.log 2>&1
f=$(ls $OUT/stats/*/*kernel_stats.csv | head -1); head -8 $f | cut -d, -f1-5 | cut -c1-160
python - <<'PY'
import json, os
f = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04c7/bench_clenc_precise.json"
d = json.loads([l for l in open(f) if l.startswith("{")][-1])
print("cl-enc precise: value %.1f ms %.4f fwd_bwd %s parity %s" % (d["value"], d["ms_per_step"], d.get("fwd_bwd"), d.get("parity")))
PY
