# the round's profile call: rocprofv3 stats + PMC passes of the bench command (profile_round.sh), PMC at cl-dec, the MFMA power probe
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
ROUND=r05 bash tools/profile_round.sh > gpurun_out/r05/profile.log 2>&1; tail -5 gpurun_out/r05/profile.log
WL=cl-dec bash tools/pmc_workload.sh > gpurun_out/r05/pmc_cl-dec.log 2>&1; tail -4 gpurun_out/r05/pmc_cl-dec.log
python tools/probe_mfma_power.py > gpurun_out/r05/c3_probe.json 2> gpurun_out/r05/c3_probe.err; grep -c gap gpurun_out/r05/c3_probe.err
