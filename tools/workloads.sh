# One bench.py line per BASELINE workload (SURVEY 8d shapes) -> gpurun_out/prof_$ROUND/workloads.jsonl (copied to profiles/$ROUND by collect_round.py)
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r05}
OUT=$R/gpurun_out/prof_$ROUND
mkdir -p $OUT
: > $OUT/workloads.jsonl
for w in ms-enc ms-dec cl-enc cl-dec dit; do
  timeout 300 python $R/bench.py --workload $w --block-steps 0 2>/dev/null | tail -1 >> $OUT/workloads.jsonl
done
# the fp32-faithful mode at the CLEVR-TR encoder shape (runs/clevrtr/GTA/gta/config.yaml:55 mixed_prec: False)
timeout 300 python $R/bench.py --workload cl-enc --dtype f32 --precise --block-steps 0 2>/dev/null | tail -1 >> $OUT/workloads.jsonl
python - <<PY
import json
for l in open("$OUT/workloads.jsonl"):
    d = json.loads(l); r = d.get("roofline") or {"kernel": "(single-kernel plan)", "kernel_ms": float("nan"), "frac": float("nan"), "step_frac": float("nan")}
    print(d["config"]["workload"].split(":")[0], round(d["value"], 1), "Mtok/s", round(d["ms_per_step"] * 1e3, 1), "us/step", r["kernel"], round(r["kernel_ms"] * 1e3, 1), "us frac", round(r["frac"], 3),
          "step_frac", round(r["step_frac"], 3), "fwd_bwd", round(d["fwd_bwd"]["ms_per_step"], 3) if "fwd_bwd" in d else None, "parity", d["parity"]["parity_max_abs"], d["dtype"])
PY
