cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
t0=$(date +%s.%N)
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05_bench.json
t1=$(date +%s.%N)
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("wall", round($t1-$t0,1), "ms-enc", round(d["value"],1), round(d["ms_per_step"]*1e3,1), "kernel", round(r["kernel_ms"]*1e3,1), round(r["frac"],3), round(r["sclk_mhz"]), "cold", round(d["cold_start"]["value"],1), "fwd_bwd", round(d["fwd_bwd"]["ms_per_step"],3), "block", [round(v) for v in d["block_layer"]["fused"].values()], "cpu", round(d["cpu_baseline"]["value"],4), d["cpu_baseline"]["cores"], "errs", d.get("extra_leg_errors"))
print("   ", {w:[round(v,4) for v in x.get("ms_per_step_regions",[])] for w,x in d["workloads"].items()}, {w:round(x.get("fwd_bwd_ms",0),3) for w,x in d["workloads"].items()})
PY
done
grep thrott /sys/fs/cgroup/cpu.stat
