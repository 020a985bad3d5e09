import os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gta_amd
from tests import _hip_cases as C
MS = ({"triv": 0, "se3": 48, "so3": 24, "so2": 24}, 6, 2)
random.seed(5)
bad = 0
for it, (B, H, Nq, Pq, Nk, Pk) in enumerate([(4, 8, 5, 512, 5, 256), (2, 8, 10, 256, 5, 512), (8, 4, 5, 256, 5, 256), (3, 8, 7, 300, 9, 200), (1, 16, 20, 128, 20, 128), (2, 8, 5, 1000, 3, 700)]):
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, MS[0], MS[1], MS[2], torch.bfloat16, seed=700 + it)
    w = torch.randn(q.shape, generator=torch.Generator().manual_seed(900 + it)).cuda()
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, MS[0])
    res = {}
    for mode in ("prepass_bwd_keys32", "prepass_bwd_keys64"):
        qd, kd, vd = (t.bfloat16().cuda().requires_grad_() for t in (q, k, v))
        tcd = torch.tensor([0.37], device="cuda", requires_grad=True)
        out = gta_amd.gta_attention(qd, kd, vd, MS[0], packed, so3_degree=2, trans_coeff=tcd, kv_mode=mode)
        (out.float() * w).sum().backward()
        torch.cuda.synchronize()
        res[mode] = (qd.grad.float().cpu(), kd.grad.float().cpu(), vd.grad.float().cpu(), float(tcd.grad.item()))
    a, b = res["prepass_bwd_keys32"], res["prepass_bwd_keys64"]
    eq = [bool(torch.equal(a[i], b[i])) for i in range(3)]
    ok = all(eq) and abs(a[3] - b[3]) <= 1e-4 * max(1.0, abs(a[3]))
    bad += not ok
    print(("ok  " if ok else "BAD ") + f"B{B} H{H} q {Nq}x{Pq} k {Nk}x{Pk}: dq {eq[0]} dk {eq[1]} dv {eq[2]} dtc {a[3]:.4f} / {b[3]:.4f}", flush=True)
print("BAD:", bad)
