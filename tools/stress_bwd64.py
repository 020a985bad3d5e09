"""Randomised shapes through the generated backward kernels (gta_bwd_dkv64_kernel, gta_bwd_dq64_kernel: gen_bwd64.py) against the compiled
32-per-wave ones: dq, dk, dv must agree BIT FOR BIT, d trans_coeff to 1e-4 (developer tool):
python tools/stress_bwd64.py <seed> <count>"""
import os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gta_amd
from tests import _hip_cases as C
MS = ({"triv": 0, "se3": 48, "so3": 24, "so2": 24}, 6, 2)
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    Nk = random.choice([1, 2, 3, 4, 5, 8])
    Pk = random.choice([16, 40, 64, 96, 100, 128, 150, 192, 250, 256, 300, 320])
    Nq = random.choice([1, 2, 3, 5, 6])
    Pq = random.choice([16, 40, 48, 64, 100, 128, 200, 256, 300])
    B, H = random.choice([1, 2, 3]), random.choice([1, 2, 3, 6, 8])
    vt = random.random() < 0.8
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, MS[0], MS[1], MS[2], torch.bfloat16, seed=500 + it)
    qm = random.choice([1.0, 1.0, 3.0])
    q = q * qm
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
        out = gta_amd.gta_attention(qd, kd, vd, MS[0], packed, so3_degree=2, trans_coeff=tcd, kv_mode=mode, v_transform=vt)
        (out.float() * w).sum().backward()
        torch.cuda.synchronize()
        res[mode] = (qd.grad.float().cpu(), kd.grad.float().cpu(), vd.grad.float().cpu(), float(tcd.grad.item()))
    a, b = res["prepass_bwd_keys32"], res["prepass_bwd_keys64"]
    eq = [bool(torch.equal(a[i], b[i])) for i in range(3)]
    fin = all(bool(torch.isfinite(b[i]).all()) for i in range(3))
    tc_ok = abs(a[3] - b[3]) <= 1e-4 * max(1.0, abs(a[3]))
    ok = all(eq) and fin and tc_ok
    bad += not ok
    print(("ok  " if ok else "BAD ") + f"B{B} H{H} q {Nq}x{Pq} k {Nk}x{Pk} vt {int(vt)} qx{qm:g}: dq {eq[0]} dk {eq[1]} dv {eq[2]} finite {fin} dtc {a[3]:.5f} / {b[3]:.5f}"
          + ("" if all(eq) else f"  max diff dq {(a[0] - b[0]).abs().max():.3e} dk {(a[1] - b[1]).abs().max():.3e} dv {(a[2] - b[2]).abs().max():.3e}"), flush=True)
print("BAD:", bad)
