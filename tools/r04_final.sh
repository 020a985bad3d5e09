# round 4: the whole GPU suite, then the profile set of the bench command and the workloads table (-> gpurun_out/prof_r04, collected into profiles/r04 by tools/collect_r04.py)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04final; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; echo "gpu suite rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
bash tools/profile_r04.sh > $OUT/profile.log 2>&1
bash tools/workloads_r04.sh > $OUT/workloads.log 2>&1; tail -8 $OUT/workloads.log
bash tools/prof_bwd.sh > $OUT/prof_bwd.log 2>&1; tail -8 $OUT/prof_bwd.log
cat gpurun_out/prof_r04/step_gaps.txt
tail -c 1500 gpurun_out/prof_r04/bench_line.json
python tools/check_attn64.py ramp ms-enc > $OUT/ramp.txt 2>&1; grep -v amdgpu.ids $OUT/ramp.txt | tail -20
GTA_HIP_LIB=$R/gta_amd/csrc/libgta_hip_diag.so GTA_ATTN64_VARIANT=5 timeout 200 python tools/check_attn64.py phases ms-enc 2>&1 | grep "attn64" | tee $OUT/phases_items.txt
