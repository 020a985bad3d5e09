set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
N=${N:-2}
timeout 500 python -m pytest tests -m gpu -q > gpurun_out/r05/c${N}_pytest.log 2>&1; tail -5 gpurun_out/r05/c${N}_pytest.log
timeout 200 python tools/probe_mfma_power.py > gpurun_out/r05/c${N}_probe.json 2> gpurun_out/r05/c${N}_probe.err; grep -c gap gpurun_out/r05/c${N}_probe.err
timeout 300 python bench.py > gpurun_out/r05/c${N}_bench.json 2> gpurun_out/r05/c${N}_bench.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r05/c${N}_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("ms-enc", round(d["value"],1), round(d["ms_per_step"]*1e3,1), "kernel", round(r["kernel_ms"]*1e3,1), round(r["frac"],3), round(r["sclk_mhz"]), round(r["kernel_cycles"]/1e3,1), "fwd_bwd", d["fwd_bwd"]["ms_per_step"], "block", d["block_layer"]["fused"])
for w,x in d["workloads"].items(): print(w, {k:(round(v,3) if isinstance(v,float) else v) for k,v in x.items() if k in ("value","ms_per_step","kernel","kernel_ms","frac","sclk_mhz","kernel_cycles","fwd_bwd_ms","error")})
PY
