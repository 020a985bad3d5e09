# the round's profile call: rocprofv3 stats + PMC passes of the bench command (profile_round.sh), PMC passes at cl-dec, the backward's
# kernel times at three workloads, the MFMA power probe
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
ROUND=r05 bash tools/profile_round.sh > gpurun_out/r05/profile.log 2>&1; tail -3 gpurun_out/r05/profile.log
WL=cl-dec bash tools/pmc_workload.sh > gpurun_out/r05/pmc_cl-dec.log 2>&1; tail -2 gpurun_out/r05/pmc_cl-dec.log
for w in cl-enc cl-dec dit; do WL=$w bash tools/prof_workload_bwd.sh > gpurun_out/r05/prof_bwd_$w.txt 2>&1; done
python tools/probe_mfma_power.py > gpurun_out/r05/c4_probe.json 2> gpurun_out/r05/c4_probe.err; grep -c gap gpurun_out/r05/c4_probe.err
