# round 6's profile call: rocprofv3 stats + PMC passes of the bench command (profile_round.sh), PMC passes at the four other workloads,
# forward + backward kernel times at cl-enc in both arithmetic modes, the fused block's launch list, item phases of every attention kernel
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
ROUND=r06 bash tools/profile_round.sh > gpurun_out/r06/profile.log 2>&1; tail -3 gpurun_out/r06/profile.log
for w in cl-dec cl-enc dit ms-dec; do WL=$w bash tools/pmc_workload.sh > gpurun_out/r06/pmc_$w.log 2>&1; tail -2 gpurun_out/r06/pmc_$w.log; done
for w in cl-enc cl-dec dit; do WL=$w bash tools/prof_workload_bwd.sh > gpurun_out/r06/prof_bwd_$w.txt 2>&1; done
WL=cl-enc TAG=cl-enc-f32-faithful XARGS="--dtype f32 --precise" bash tools/prof_workload_bwd.sh > gpurun_out/r06/prof_bwd_cl-enc-f32-faithful.txt 2>&1
bash tools/trace_block.sh > gpurun_out/r06/trace_block.txt 2>&1
python tools/item_phases.py ms-enc,ms-dec,cl-enc,cl-dec,dit > gpurun_out/r06/item_phases.txt 2>&1
python tools/wg_timeline.py cl-dec,cl-enc > gpurun_out/r06/wg_timeline.txt 2>&1
