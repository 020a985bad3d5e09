"""GPU: the fp32-faithful arithmetic mode (GTA_FLAG_FP32_PRODUCTS, ``gta_attention(precise=True)``).

The reference's ``mixed_prec: False`` configs (runs/clevrtr/GTA/gta/config.yaml:55) compute the operator in true fp32.
The default kernels round q', k', v', P to bf16 once (the reference's bf16-autocast accuracy: tolerance 2.5e-2 * max,
1.2e-2 rel-RMS, tests/test_gpu_forward.py).  The precise mode keeps every operand as a bf16 hi+lo pair (16 significant
bits) and spends three MFMAs per product; its tolerance, stated here and in DESIGN.md section 7, is

    max |hip - ref| <= 1e-4 * max |ref|        rms(hip - ref) <= 3e-5 * rms(ref)

against the REFERENCE's fp64 fixtures and against the fp64 oracle at the BASELINE shapes -- 250x / 400x tighter than the
bf16 bars."""
import pytest
import torch

import gta_amd
from gta_amd import native
from tests import _golden as G
from tests import _hip_cases as C

pytestmark = pytest.mark.gpu

REL_MAX, REL_RMS = 1e-4, 3e-5
FUSED_CASES = [c for c in G.list_cases("op_") if C.FUSED_OK(G.load("op_" + c)[1])]


def _forward(case, precise):
    from types import SimpleNamespace
    d, meta = G.load("op_" + case)
    ex = G.extras_of(d, torch.float32, "cuda")
    q, k, v = (torch.from_numpy(d[n]).float().cuda() for n in "qkv")
    tc = torch.tensor([float(d["trans_coeff"])], device="cuda")
    out, _ = gta_amd.multihead_geometric_transform_attention(
        q, k, v, attn_fn=SimpleNamespace(scale=float(d["scale"]), tau=G.tau_of(d, torch.float32, "cuda", grad=False)),
        f_dims=meta["f_dims"], reps=ex, trans_coeff=tc, v_transform=meta["v_transform"], euclid=meta["euclid"], precise=precise)
    torch.cuda.synchronize()
    return out.float().cpu(), torch.from_numpy(d["out"]).float()


@pytest.mark.parametrize("case", G.list_cases("op_"))
def test_reference_fixture_fp32_faithful(case):
    """Every reference operator fixture (fused layouts and the generic-path ablations t2 / euclid) at fp32 accuracy."""
    got, ref = _forward(case, True)
    st = C.err_stats(got, ref)
    assert st["finite"] and st["max_abs"] <= REL_MAX * st["ref_max"] and st["rel_rms"] <= REL_RMS, st
    # and it IS a different arithmetic: the default path sits where bf16 products put it
    got_bf, _ = _forward(case, False)
    assert C.err_stats(got_bf, ref)["rel_rms"] > 10 * st["rel_rms"]


@pytest.mark.parametrize("shape", ["C1", "CL-enc", "MS-enc", "ragged", "wide-ragged"])
def test_baseline_shapes_fp32_faithful(shape):
    from tests.test_gpu_forward import SHAPES
    B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = SHAPES[shape]
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, torch.float32, seed=21)
    ref = C.oracle_forward(q, k, v, ex, ak, cross, 0.01, dtype=torch.float64).float()
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    tc = torch.tensor([0.01], device="cuda") if f_dims.get("se3", 0) > 0 else None
    out = gta_amd.gta_attention(q.cuda(), k.cuda(), v.cuda(), f_dims, packed, so3_degree=exd.get("gta_so3_degree", 0),
                                trans_coeff=tc, precise=True)
    torch.cuda.synchronize()
    st = C.err_stats(out.float().cpu(), ref)
    assert st["finite"] and st["max_abs"] <= REL_MAX * st["ref_max"] and st["rel_rms"] <= REL_RMS, st


def test_precise_mode_contract():
    """bf16 inputs are refused (they ask for bf16 arithmetic); the mode is a per-call argument (per-module: Attention.precise); gradients
    flow (the backward stays on bf16 products)."""
    f_dims = {"se3": 32, "so2": 32}
    q, k, v, ex, ak, _ = C.synth_inputs(1, 2, 2, 40, 2, 40, f_dims, 8, 0, torch.float32, seed=2)
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    tc = torch.tensor([0.01], device="cuda")
    with pytest.raises(native.GtaError):
        gta_amd.gta_attention(q.cuda().bfloat16(), k.cuda().bfloat16(), v.cuda().bfloat16(), f_dims, packed, trans_coeff=tc, precise=True)
    a = gta_amd.gta_attention(q.cuda(), k.cuda(), v.cuda(), f_dims, packed, trans_coeff=tc, precise=True)
    b = gta_amd.gta_attention(q.cuda(), k.cuda(), v.cuda(), f_dims, packed, trans_coeff=tc, precise=True)
    assert torch.equal(a, b)
    qg = q.cuda().requires_grad_()
    out = gta_amd.gta_attention(qg, k.cuda(), v.cuda(), f_dims, packed, trans_coeff=tc, precise=True)
    out.sum().backward()
    assert torch.isfinite(qg.grad).all() and qg.grad.abs().max() > 0


# ---------------------------------------------------------------------------------------------------------------------------
# gradients of the fp32-faithful mode: rho in fp32 (gta_rep_apply / its adjoint), split-bf16 forward, EXACT-fp32 backward
# (gta_plain32.hip: v_mfma_f32_32x32x2_f32).  Same bars as the forward: 1e-4 * max, 3e-5 rel-RMS.
# ---------------------------------------------------------------------------------------------------------------------------
def _grad_check(got, ref, name, rel_max=REL_MAX, rel_rms=1e-4):
    # (the gradient's rel-RMS bar is 1e-4: the forward it differentiates is the split-bf16 one, ~6e-6 off the fp64 value, and the
    #  softmax Jacobian amplifies that by a few; the default mode's bars are 4e-2 / 2e-2)
    st = C.err_stats(got, ref)
    assert st["finite"] and st["max_abs"] <= rel_max * max(st["ref_max"], 1e-30) and st["rel_rms"] <= rel_rms, (name, st)


@pytest.mark.parametrize("case", G.list_cases("op_"))
def test_reference_fixture_gradients_fp32_faithful(case):
    """dq, dk, dv, d trans_coeff, d tau of the REFERENCE's own autograd (fixtures generated by oracle/make_golden.py from the imported
    reference, fp64) on every operator case: the CLEVR-TR and MSN layouts, v_transform off, so2-only, t2, euclid, adjustable softmax."""
    from types import SimpleNamespace
    d, meta = G.load("op_" + case)
    ex = G.extras_of(d, torch.float32, "cuda")
    q, k, v = (torch.from_numpy(d[n]).float().cuda().requires_grad_() for n in "qkv")
    tc = torch.tensor([float(d["trans_coeff"])], device="cuda", requires_grad=True)
    tau = G.tau_of(d, torch.float32, "cuda")
    out, _ = gta_amd.multihead_geometric_transform_attention(
        q, k, v, attn_fn=SimpleNamespace(scale=float(d["scale"]), tau=tau), f_dims=meta["f_dims"], reps=ex,
        trans_coeff=tc, v_transform=meta["v_transform"], euclid=meta["euclid"], precise=True)
    (out.float() * torch.from_numpy(d["w"]).float().cuda()).sum().backward()
    torch.cuda.synchronize()
    _grad_check(out.detach().float().cpu(), torch.from_numpy(d["out"]).float(), "out", REL_MAX, REL_RMS)
    for name, t in (("dq", q), ("dk", k), ("dv", v)):
        _grad_check(t.grad.float().cpu(), torch.from_numpy(d[name]).float(), name)
    if meta["f_dims"].get("se3", 0) > 0 and "dtrans_coeff" in d:
        ref, got = float(d["dtrans_coeff"][0]), float(tc.grad.item())
        assert abs(got - ref) <= 2e-4 * max(1.0, abs(ref)), (got, ref)
    if tau is not None:
        ref, got = float(d["dtau"][0]), float(tau.grad.item())
        assert abs(got - ref) <= 2e-4 * max(1.0, abs(ref)), (got, ref)
    # and it IS a different arithmetic from the default backward (bf16 products): that one sits two orders of magnitude away
    q2, k2, v2 = (torch.from_numpy(d[n]).float().cuda().requires_grad_() for n in "qkv")
    out2, _ = gta_amd.multihead_geometric_transform_attention(
        q2, k2, v2, attn_fn=SimpleNamespace(scale=float(d["scale"]), tau=G.tau_of(d, torch.float32, "cuda", grad=False)), f_dims=meta["f_dims"],
        reps=ex, trans_coeff=float(d["trans_coeff"]), v_transform=meta["v_transform"], euclid=meta["euclid"])
    (out2.float() * torch.from_numpy(d["w"]).float().cuda()).sum().backward()
    torch.cuda.synchronize()
    assert C.err_stats(q2.grad.float().cpu(), torch.from_numpy(d["dq"]).float())["rel_rms"] > 10 * C.err_stats(q.grad.float().cpu(), torch.from_numpy(d["dq"]).float())["rel_rms"]


@pytest.mark.parametrize("shape", ["C1", "CL-enc", "CL-dec", "MS-enc", "ragged", "wide", "wide-ragged"])
def test_baseline_shape_gradients_fp32_faithful(shape):
    """the same against fp64 autograd through the oracle at the BASELINE geometries (CLEVR-TR encoder / decoder: the configs the
    reference trains in fp32; dh = 96 and the padded dh = 104 -> 128 layouts; ragged tiles)"""
    from oracle import gta_oracle as O
    from tests.test_gpu_backward import SHAPES
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
    qd, kd, vd = (t.cuda().requires_grad_() for t in (q, k, v))
    tcd = torch.tensor([0.37], device="cuda", requires_grad=True)
    taud = torch.tensor([0.8], device="cuda", requires_grad=True)
    out = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=exd.get("gta_so3_degree", 0),
                                trans_coeff=tcd if f_dims.get("se3", 0) > 0 else None, tau=taud, precise=True)
    (out * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    _grad_check(out.detach().cpu(), out_o.detach().float(), "out", REL_MAX, REL_RMS)
    for name, a, b in (("dq", qd, qo), ("dk", kd, ko), ("dv", vd, vo)):
        _grad_check(a.grad.cpu(), b.grad.float(), name)
    if f_dims.get("se3", 0) > 0:
        ref, got = float(tco.grad.item()), float(tcd.grad.item())
        assert abs(got - ref) <= 3e-4 * max(1.0, abs(ref)), (got, ref)
    ref, got = float(taus.grad.item()), float(taud.grad.item())
    assert abs(got - ref) <= 3e-4 * max(1.0, abs(ref)), (got, ref)


# ---------------------------------------------------------------------------------------------------------------------------
# r05: the fp32-faithful forward on the TWO-STAGE plan at dh <= 64 (gta_prep.hip writes hi and lo images, gta_fwd2_kernel<..., X3> runs
# three MFMAs per product): same bars, against the fp64 oracle and against the single-kernel plan of the same mode.
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", ["CL-enc", "CL-dec", "DT", "ragged"])
def test_two_stage_plan_fp32_faithful(shape):
    from tests.test_gpu_forward import SHAPES
    B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = SHAPES[shape]
    if shape == "ragged":
        Pq, Pk = 137, 151                      # (dh = 32; ragged rows, a masked key tail, more than 256 query rows)
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, torch.float32, seed=33)
    ref = C.oracle_forward(q, k, v, ex, ak, cross, 0.3, dtype=torch.float64).float()
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    tc = torch.tensor([0.3], device="cuda") if f_dims.get("se3", 0) > 0 else None
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    need_view = f_dims.get("se3", 0) > 0 or f_dims.get("so3", 0) > 0
    desc = native.make_desc(qd, kd, vd, qd, f_dims, 0, Nq if need_view else 1, Nk if need_view else 1, q.shape[-1] ** -0.5,
                            native.FLAG_V_TRANSFORM | native.FLAG_FP32_PRODUCTS)
    assert native.attention_kernel(desc)[0] == "gta_fwd2_kernel" and native.attn_fwd_workspace_bytes(desc) > 0
    outs = {}
    for mode in ("prepass", "fused"):
        outs[mode] = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=0, trans_coeff=tc, precise=True, kv_mode=mode).float().cpu()
        st = C.err_stats(outs[mode], ref)
        assert st["finite"] and st["max_abs"] <= REL_MAX * st["ref_max"] and st["rel_rms"] <= REL_RMS, (mode, st)
    st = C.err_stats(outs["prepass"], outs["fused"])
    assert st["max_abs"] <= 3e-5 * st["ref_max"] and st["rel_rms"] <= 2e-5, st
    # the key side's images serve a second query set (chunked decode): same result as the uncached call
    if Nq * Pq > 300:
        cache = {}
        with torch.no_grad():
            a = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=0, trans_coeff=tc, precise=True, kv_cache=cache)
            b = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=0, trans_coeff=tc, precise=True, kv_cache=cache)
        assert torch.equal(a, b) and torch.equal(a.float().cpu(), outs["prepass"])
        #  a cache written under one plan is refused by the other: the fp32-faithful plan stores four images per tile, the default
        # plan two, and bf16 inputs pick other instances again -- before, only the byte size was looked at and the mixed call read lo parts as tiles
        with torch.no_grad():
            with pytest.raises(native.GtaError, match="another plan"):
                gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=0, trans_coeff=tc, precise=False, kv_cache=cache)
            with pytest.raises(native.GtaError, match="another plan"):
                gta_amd.gta_attention(qd.bfloat16(), kd.bfloat16(), vd.bfloat16(), f_dims, packed, so3_degree=0, trans_coeff=tc, kv_cache=cache)
            fresh = {}
            c = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=0, trans_coeff=tc, precise=False, kv_cache=fresh)
            d = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=0, trans_coeff=tc, precise=False, kv_cache=fresh)
            with pytest.raises(native.GtaError, match="another plan"):
                gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=0, trans_coeff=tc, precise=True, kv_cache=fresh)
        assert torch.equal(c, d)
        st = C.err_stats(c.float().cpu(), ref)
        assert st["finite"] and st["max_abs"] <= 2.5e-2 * st["ref_max"], st


def test_two_stage_plan_fp32_faithful_not_at_dh96():
    """dh = 96 keeps the single-kernel plan in this mode: the library says so through the workspace size (0), the host mirror follows"""
    from tests.test_gpu_forward import SHAPES
    B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = SHAPES["MS-enc"]
    q = torch.zeros(B, H, Nq * Pq, 96, device="cuda")
    desc = native.make_desc(q, q, q, q, f_dims, so3, Nq, Nk, 96 ** -0.5, native.FLAG_V_TRANSFORM | native.FLAG_FP32_PRODUCTS)
    assert native.attn_fwd_workspace_bytes(desc) == 0 and native.attention_kernel(desc)[0] == "gta_fwd_kernel"
