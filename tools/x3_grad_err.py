"""developer tool: d tau / d trans_coeff / dq / dk / dv errors of the fp32-faithful fused backward (X3 walks) against fp64 autograd through the
oracle, per BASELINE geometry -- the numbers behind the bars of tests/test_gpu_precise.py::test_baseline_shape_gradients_fp32_faithful"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gta_amd
from oracle import gta_oracle as O
from tests import _hip_cases as C
from tests.test_gpu_backward import SHAPES

for shape in sys.argv[1:] or ["C1", "CL-enc", "CL-dec", "ragged"]:
    B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = SHAPES[shape]
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, torch.float32, seed=7)
    w = torch.randn(q.shape, generator=torch.Generator().manual_seed(11))
    qo, ko, vo = (t.double().requires_grad_() for t in (q, k, v))
    tco = torch.tensor([0.37], dtype=torch.float64, requires_grad=True)
    taus = torch.tensor([0.8], dtype=torch.float64, requires_grad=True)
    ex64 = {kk: (vv.double() if vv.is_floating_point() else vv) for kk, vv in ex.items()}
    reps = O.encoder_reps(ak, ex64)
    if cross:
        reps = O.decoder_reps(ak, ex64, reps)
    out_o, _ = O.gta_attention(qo, ko, vo, f_dims, reps, tco, tau=taus)
    (out_o * w.double()).sum().backward()
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    for mode in ("default", "fused"):
        qd, kd, vd = (t.cuda().requires_grad_() for t in (q, k, v))
        tcd = torch.tensor([0.37], device="cuda", requires_grad=True)
        taud = torch.tensor([0.8], device="cuda", requires_grad=True)
        kw = {} if mode == "default" else {"kv_mode": "fused"}
        out = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=exd.get("gta_so3_degree", 0),
                                    trans_coeff=tcd if f_dims.get("se3", 0) > 0 else None, tau=taud, precise=True, **kw)
        (out * w.cuda()).sum().backward()
        torch.cuda.synchronize()
        line = [f"{shape:12s} {mode:8s}"]
        for name, a, b in (("out", out.detach(), out_o.detach()), ("dq", qd.grad, qo.grad), ("dk", kd.grad, ko.grad), ("dv", vd.grad, vo.grad)):
            s = C.err_stats(a.float().cpu(), b.float())
            line.append(f"{name} rms {s['rel_rms']:.2e}")
        if tcd.grad is not None:
            line.append(f"dtc rel {abs(tcd.grad.item() - tco.grad.item()) / max(1, abs(tco.grad.item())):.2e}")
        line.append(f"dtau rel {abs(taud.grad.item() - taus.grad.item()) / max(1, abs(taus.grad.item())):.2e} (ref {taus.grad.item():+.4f})")
        print("  ".join(line), flush=True)
