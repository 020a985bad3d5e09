"""GPU: the planned serving path (gta_amd.plan) -- same results as the autograd path, and the FULL-SIZE benchmark batch
(B = 32: 2 560 work items, the XCD work map) checked against the oracle on sampled scenes."""
import pytest
import torch

import gta_amd
from gta_amd import native, plan, synth
from oracle import gta_oracle as O
from tests import _hip_cases as C

pytestmark = pytest.mark.gpu

MS = {"triv": 0, "se3": 48, "so3": 24, "so2": 24}


def _oracle(q, k, v, ex, ak, cross, idx):
    sub = {kk: vv[idx] for kk, vv in ex.items()}
    return C.oracle_forward(q[idx], k[idx], v[idx], sub, ak, cross, 0.01)


@pytest.mark.parametrize("persist", [False, True])
def test_full_size_batch_matches_oracle_on_sampled_scenes(persist):
    B, H, N, P = 32, 8, 5, 256
    q, k, v, ex, ak, cross = synth.attention_inputs(B, H, N, P, N, P, MS, 6, 2, seed=77)
    qd, kd, vd = (synth.as_projection(t, torch.bfloat16, "cuda") for t in (q, k, v))
    rp = plan.RepPlan(B, N, P, 2, 6)
    vrep, cs = rp(ex["input_transforms"].cuda().contiguous(), ex["input_coord"].cuda().contiguous())
    fp = plan.ForwardPlan(qd, kd, vd, MS, so3_degree=2, Nq=N, Nk=N, flags=native.FLAG_PERSIST if persist else 0)
    tc = torch.tensor([0.01], device="cuda")
    out = fp(qd, kd, vd, vrep, vrep, cs, cs, tc)
    torch.cuda.synchronize()
    idx = torch.tensor([0, 9, 22, 31])
    ref = _oracle(q, k, v, ex, ak, cross, idx)
    st = C.err_stats(out[idx.cuda()].float().cpu(), ref)
    assert st["finite"] and st["max_abs"] <= 2.5e-2 * st["ref_max"] and st["rel_rms"] <= 1.2e-2, st
    # every scene is finite and no scene was left untouched (first / last work items of every XCD included)
    assert torch.isfinite(out.float()).all()
    assert (out.float().abs().amax(dim=(1, 2, 3)) > 0).all()


def test_plan_equals_autograd_path_and_is_reusable():
    B, H, Nq, Pq, Nk, Pk = 2, 2, 3, 50, 2, 70
    f_dims = {"se3": 32, "so2": 32}
    q, k, v, ex, ak, cross = synth.attention_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, 8, 0, seed=5)
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    qd, kd, vd = (synth.as_projection(t, torch.bfloat16, "cuda") for t in (q, k, v))
    tc = torch.tensor([0.01], device="cuda")
    ref = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, trans_coeff=tc, kv_mode="prepass")
    fp = plan.ForwardPlan(qd, kd, vd, f_dims, Nq=Nq, Nk=Nk)
    for _ in range(2):                                       # second call: buffers are reused
        out = fp(qd, kd, vd, packed["vrep_q"], packed["vrep_k"], packed["cs_q"], packed["cs_k"], tc)
        assert torch.equal(out, ref)
    with pytest.raises(native.GtaError):
        fp(qd[:, :, :100], kd, vd, packed["vrep_q"], packed["vrep_k"], packed["cs_q"], packed["cs_k"], tc)
    with pytest.raises(native.GtaError):
        fp(qd, kd, vd, packed["vrep_q"], packed["vrep_k"], packed["cs_q"][:, :7], packed["cs_k"], tc)


def test_rep_plan_equals_the_builders():
    g = torch.Generator().manual_seed(3)
    E = O.random_extrinsics(4, 5, g).cuda()
    coord = torch.rand(4, 5, 256, 2, generator=g).cuda()
    rp = plan.RepPlan(4, 5, 256, 2, 6)
    vrep, cs = rp(E, coord)
    v2, c2 = native.build_reps(E, 2, coord.reshape(4, -1, 2), 6, 1.0, 1.0, False)
    assert torch.equal(vrep, v2) and torch.equal(cs, c2)


def test_bench_line_contract_on_the_gpu():
    """`bench.py` end to end with small counts (the driver's command line shape): ONE JSON line with the contract's keys, the roofline and
    cpu-baseline objects, the r05 additions (`cold_start`, `preconditioning`, `workloads` with per-workload kernel time and parity) and no failed
    extra leg."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--precondition-s", "0.05",
                        "--workloads", "cl-enc", "--block-steps", "0", "--train-steps", "1", "--batch", "4"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "parity", "cold_start", "preconditioning", "workloads"):
        assert key in d, key
    assert d["steps"] == 4 and d["warmup"] == 2 and d["n_gpus"] == 1 and d["extra_leg_errors"] is None
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["kernel"] == "gta_attn64_items_kernel" and 0 < rf["frac"] < 1 and rf["kernel_ms"] < d["ms_per_step"]
    assert d["parity"]["parity_max_abs"] <= 2.5e-2 * d["parity"]["ref_max_abs"]
    w = d["workloads"]["cl-enc"]
    assert "error" not in w and w["kernel"] == "gta_fwdc_kernel" and 0 < w["frac"] < 1 and len(w["ms_per_step_regions"]) == 3
    assert w["parity"]["parity_max_abs"] <= 2.5e-2 * w["parity"]["ref_max_abs"]
    assert d["cold_start"]["value"] > 0 and d["preconditioning"]["steps"] > 0
