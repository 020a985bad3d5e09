import sys, torch
sys.path.insert(0,'.')
from tests import _hip_cases as C
B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = 1, 4, 2, 160, 2, 160, {"triv": 0, "se3": 48, "so3": 24, "so2": 24}, 6, 2
for pattern in ["hot_logits","late_spike","early_spike","plain"]:
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, torch.float32, seed=21)
    if pattern == "hot_logits": q, k = q * 3.5, k * 3.5
    elif pattern == "late_spike": k[:, :, -64:] *= 30.0; q = q * 1.5
    elif pattern == "early_spike": k[:, :, :64] *= 30.0; q = q * 1.5
    q, k, v = q.bfloat16().float(), k.bfloat16().float(), v.bfloat16().float()
    ref = C.oracle_forward(q, k, v, ex, ak, cross, 0.01)
    outs = {m: C.hip_forward(q, k, v, ex, ak, cross, 0.01, torch.bfloat16, kv_mode=m).float().cpu() for m in ("fused","prepass","prepass8")}
    o32 = C.hip_forward(q, k, v, ex, ak, cross, 0.01, torch.float32, kv_mode="prepass").float().cpu()
    e = lambda a,b: ((a-b).abs().max().item(), ((a-b).pow(2).mean().sqrt()/b.pow(2).mean().sqrt()).item())
    print(pattern, "ref_max %.2f"%ref.abs().max(), "| fused vs ref", e(outs["fused"],ref), "| prepass vs ref", e(outs["prepass"],ref), "| prepass8 vs ref", e(outs["prepass8"],ref), "| prepass vs fused", e(outs["prepass"],outs["fused"]), "| fp32-io prepass vs ref", e(o32, ref))
