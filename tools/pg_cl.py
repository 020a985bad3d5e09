"""cl-dec / cl-enc / dit forward (rep tables ready): one workgroup per item against the persistent grid (kv_mode prepass_pg), step time by events"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, gta_amd
from gta_amd import synth
for wl in (sys.argv[1] if len(sys.argv) > 1 else "cl-dec,cl-enc").split(","):
    H, Nq, Pq, Nk, Pk, f_dims, so2, so3, B = bench.WORKLOADS[wl]
    q, k, v, ex, ak, cross = synth.attention_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, seed=1)
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    qd, kd, vd = (t.bfloat16().cuda() for t in (q, k, v))
    tc = torch.tensor([0.01], device="cuda")
    res = {}
    for rep in range(3):
        for mode in ("prepass", "prepass_pg"):
            fn = lambda: gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=exd.get("gta_so3_degree", 0), trans_coeff=tc, kv_mode=mode)
            with torch.no_grad():
                for _ in range(5):
                    o = fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    o = fn()
                e1.record()
                torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / 30 * 1e3)
    print(wl, {m: ["%.1f" % x for x in v_] for m, v_ in res.items()}, flush=True)
