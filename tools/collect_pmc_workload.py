#!/usr/bin/env python3
"""gpurun_out/pmc_<workload> (tools/pmc_workload.sh, scratch) -> profiles/<round>/pmc_<workload>.json + raw/pmc_<workload>_kernels.json (tracked).

    ROUND=r05 python tools/collect_pmc_workload.py cl-dec

The derived figures: shader cycles of a launch = SQ_BUSY_CYCLES / 32 (per-SE counter); busy shares = counter / (1024 SIMDs x that);
SQ_ACTIVE_INST_* count quad-cycles (x 4); per-wave instruction counts = SQ_INSTS_* / SQ_WAVES; FETCH_SIZE doubled (gfx950 tallies 128-B
requests at 64 B, MI355X_MICROARCH.md)."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "cl-dec"
rnd = os.environ.get("ROUND", "r05")
src = os.path.join(ROOT, "gpurun_out", "pmc_" + wl, "summary_kernels.json")
dst = os.path.join(ROOT, "profiles", rnd)
k = json.load(open(src))
name = max((n for n in k if "SQ_INSTS_MFMA" in k[n] and ("gta_fwd2_kernel" in n or "gta_fwdc_kernel" in n or "gta_attn64" in n)), key=lambda n: k[n]["pct_of_gpu_time"])
d = k[name]
simd = d["SQ_BUSY_CYCLES"] / 32.0
H, Nq, Pq, Nk, Pk, f_dims, so2, so3, B = bench.WORKLOADS[wl]
dh = sum(f_dims.values())
alg = 2 * B * H * (2 * Nq * Pq + 2 * Nk * Pk) * dh          # bf16 q, out, k, v
w = d["SQ_WAVES"]
out = {"workload": wl, "batch": B, "dtype": "bf16", "kernel": name.split("<")[0], "kernel_instance": name,
       "fetch_bytes_corrected": d["fetch_bytes_corrected"], "write_bytes": d["write_bytes"],
       "bytes_per_launch": d["fetch_bytes_corrected"] + d["write_bytes"], "algorithmic_bytes": alg, "kernel_cycles_sq": simd,
       "mfma_busy_sq": d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * simd), "mfma_valu_coexec": d["SQ_VALU_MFMA_COEXEC_CYCLES"] / (1024.0 * simd),
       "valu_port_busy": 4.0 * d["SQ_ACTIVE_INST_VALU"] / (1024.0 * simd),
       "per_wave": {"valu_class_instructions": d["SQ_INSTS_VALU"] / w, "mfma": d["SQ_INSTS_MFMA"] / w, "salu": d["SQ_INSTS_SALU"] / w,
                    "lds": d["SQ_INSTS_LDS"] / w, "vmem": d["SQ_INSTS_VMEM"] / w, "smem": d["SQ_INSTS_SMEM"] / w},
       "lds_bank_conflict_over_active": d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], "avg_us_under_rocprof": d["avg_us"],
       "source": f"tools/pmc_workload.sh (WL={wl}): separate rocprofv3 --pmc passes over bench.py --workload {wl}; FETCH_SIZE doubled (gfx950 tallies "
                 "128-B requests at 64 B, MI355X_MICROARCH.md); SQ_ACTIVE_INST_* count quad-cycles"}
os.makedirs(os.path.join(dst, "raw"), exist_ok=True)
json.dump(out, open(os.path.join(dst, f"pmc_{wl}.json"), "w"), indent=1)
shutil.copy(src, os.path.join(dst, "raw", f"pmc_{wl}_kernels.json"))
print(json.dumps(out, indent=1))
