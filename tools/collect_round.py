#!/usr/bin/env python3
"""Copy the judged summaries of tools/profile_round.sh (gpurun_out/prof_<round>, scratch) into profiles/<round> (tracked; ROUND env, default r05) and derive
pmc_traffic.json -- what bench.py quotes as roofline.traffic / mfma_busy_sq for the kernel it was measured on."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r05")
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + ROUND)
DST = os.path.join(ROOT, "profiles", ROUND)
os.makedirs(DST, exist_ok=True)
for a, b in (("summary_kernel_stats.csv", "kernel_stats.csv"), ("summary_kernels.json", "kernels.json"), ("step_gaps.txt", "step_gaps.txt"),
             ("bench_line.json", "bench_line.json")):      # (workloads.jsonl is copied by hand: its lines may come from different calls)
    if os.path.exists(os.path.join(SRC, a)):
        shutil.copy(os.path.join(SRC, a), os.path.join(DST, b))
k = json.load(open(os.path.join(SRC, "summary_kernels.json")))
name = next(n for n in k if "gta_attn64_items_kernel" in n)
d = k[name]
simd_cycles = d["SQ_BUSY_CYCLES"] / 32.0                 # (per-SE counter: /32 = shader cycles of the launch, profiles/r02/README.md)
out = {"workload": "ms-enc", "batch": 32, "dtype": "bf16", "kernel": "gta_attn64_items_kernel", "kernel_instance": name,
       "fetch_bytes_corrected": d["fetch_bytes_corrected"], "write_bytes": d["write_bytes"],
       "bytes_per_launch": d["fetch_bytes_corrected"] + d["write_bytes"], "algorithmic_bytes": 251658240,
       "mfma_busy_sq": d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * simd_cycles), "kernel_cycles_sq": simd_cycles,
       "valu_per_mfma": d["SQ_INSTS_VALU"] / d["SQ_INSTS_MFMA"], "lds_bank_conflict_cycles": d.get("SQ_LDS_BANK_CONFLICT"),
       "avg_us_under_rocprof": d["avg_us"],
       "source": "tools/profile_round.sh: separate rocprofv3 --pmc passes over bench.py (FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B, "
                 "MI355X_MICROARCH.md HBM section; WRITE_SIZE as reported; SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32))"}
json.dump(out, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
