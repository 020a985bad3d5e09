"""Accuracy and cost of the fp32-faithful mode (precise=True) vs the default bf16-product path, fp32 inputs, against the
fp64 oracle.  Usage: python tools/precise_report.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gta_amd  # noqa: E402
from tests import _hip_cases as C  # noqa: E402
from tests.test_gpu_forward import SHAPES  # noqa: E402

for shape in ("C1", "CL-enc", "CL-dec", "MS-enc", "DT"):
    B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = SHAPES[shape]
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, torch.float32, seed=21)
    ref = C.oracle_forward(q, k, v, ex, ak, cross, 0.01, dtype=torch.float64).float()
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    tc = torch.tensor([0.01], device="cuda") if f_dims.get("se3", 0) > 0 else None
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    row = []
    for precise in (False, True):
        fn = lambda: gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=exd.get("gta_so3_degree", 0), trans_coeff=tc,
                                           precise=precise, kv_mode="fused" if precise else "auto")
        out = fn()
        torch.cuda.synchronize()
        st = C.err_stats(out.float().cpu(), ref)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        row.append((st["max_abs"] / st["ref_max"], st["rel_rms"], (time.perf_counter() - t0) / 10 * 1e6))
    print(f"{shape:8s} B={B} default: max/refmax {row[0][0]:.2e} rel-rms {row[0][1]:.2e} {row[0][2]:8.1f} us | "
          f"precise: max/refmax {row[1][0]:.2e} rel-rms {row[1][1]:.2e} {row[1][2]:8.1f} us", flush=True)
