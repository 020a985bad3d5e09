"""Localise differences between the pipelined flash kernel (kv_mode='prepass8') and the plain one."""
import sys
import torch
sys.path.insert(0, ".")
from tests import _hip_cases as C

MS = {"triv": 0, "se3": 48, "so3": 24, "so2": 24}
CL = {"se3": 32, "so2": 32}
for name, (B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3) in {
        "ms 1 tile": (1, 1, 1, 256, 1, 64, MS, 6, 2), "ms 2 tiles": (1, 1, 1, 256, 1, 128, MS, 6, 2),
        "ms 3 tiles": (1, 1, 1, 256, 1, 192, MS, 6, 2), "ms 4 tiles": (1, 1, 1, 256, 1, 256, MS, 6, 2),
        "ms 5 tiles": (1, 1, 1, 256, 1, 320, MS, 6, 2), "ms 8 tiles": (1, 1, 1, 256, 2, 256, MS, 6, 2),
        "cl 2 tiles": (1, 1, 1, 256, 1, 128, CL, 8, 0), "ms tail": (1, 1, 1, 256, 1, 100, MS, 6, 2)}.items():
    for dtype in (torch.float32, torch.bfloat16):
        q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, dtype, seed=3)
        a = C.hip_forward(q, k, v, ex, ak, cross, 0.01, dtype, kv_mode="prepass").float().cpu()
        b = C.hip_forward(q, k, v, ex, ak, cross, 0.01, dtype, kv_mode="prepass8").float().cpu()
        d = (a - b).abs()[0, 0]                       # [Tq, dh]
        rows = d.reshape(-1, 32, d.shape[-1]).amax(dim=(1, 2))
        ch = d.reshape(d.shape[0], -1, 8).amax(dim=(0, 2))
        print(f"{name:12s} {str(dtype)[6:]:9s} max {d.max():.4f} (ref max {a.abs().max():.3f}) | per 32-row block:",
              " ".join(f"{x:.3f}" for x in rows.tolist()), "| per chunk:", " ".join(f"{x:.2f}" for x in ch.tolist()), flush=True)
