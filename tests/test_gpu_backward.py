"""GPU parity tests of the backward kernels: dq, dk, dv and d trans_coeff vs the reference fixtures
(autograd of the reference's own code) and vs autograd through the CPU oracle at the BASELINE shapes.
Tolerances are relative to each gradient's own magnitude (bf16 MFMA products, fp32 accumulation)."""
from types import SimpleNamespace

import pytest
import torch

import gta_amd
from tests import _golden as G
from tests import _hip_cases as C

pytestmark = pytest.mark.gpu

REL_MAX = 4e-2
REL_RMS = 2e-2
FUSED_CASES = [c for c in G.list_cases("op_") if C.FUSED_OK(G.load("op_" + c)[1])]


def _check(got, ref, name, rel_max=REL_MAX, rel_rms=REL_RMS):
    st = C.err_stats(got, ref)
    assert st["finite"], (name, st)
    assert st["max_abs"] <= rel_max * st["ref_max"] + 1e-6, (name, st)
    assert st["rel_rms"] <= rel_rms, (name, st)


def _dtc_sensitivity(d, meta):
    """|change| of the ORACLE's d trans_coeff under bf16-size relative perturbations of q, k, v (2^-9, three draws).  d trans_coeff is one
    number summed over every token with both signs; at fixture size it can cancel to a few units, and what bf16 operands leave of it is set
    by this sensitivity, not by the value (tests/test_gpu_run_configs.py, tools/dtc_matrix.py)."""
    from oracle import gta_oracle as O
    ex = G.extras_of(d)
    ak = G.attn_kwargs_of(meta)
    reps = O.encoder_reps(ak, ex)
    if meta["cross"]:
        reps = O.decoder_reps(ak, ex, reps)
    tau = G.tau_of(d, torch.float64, grad=False)
    g = torch.Generator().manual_seed(1)
    base, worst = None, 0.0
    for rep in range(4):
        q, k, v = (torch.from_numpy(d[n]).double() for n in "qkv")
        if rep:
            q, k, v = (t * (1 + (torch.rand(t.shape, generator=g, dtype=torch.float64) - 0.5) * 2.0 ** -8) for t in (q, k, v))
        tc = torch.tensor([float(d["trans_coeff"])], dtype=torch.float64, requires_grad=True)
        out, _ = O.gta_attention(q, k, v, meta["f_dims"], reps, tc, meta["v_transform"], meta["euclid"], float(d["scale"]), 1.0 if tau is None else tau)
        (out * torch.from_numpy(d["w"]).double()).sum().backward()
        if rep == 0:
            base = float(tc.grad.item())
        else:
            worst = max(worst, abs(float(tc.grad.item()) - base))
    return worst


@pytest.mark.parametrize("kv_mode", ["prepass", "fused"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", FUSED_CASES)
def test_golden_gradients(case, dtype, kv_mode):
    d, meta = G.load("op_" + case)
    ex = G.extras_of(d, torch.float32, "cuda")
    q, k, v = (torch.from_numpy(d[n]).to(dtype).cuda().requires_grad_() for n in "qkv")
    tc = torch.tensor([float(d["trans_coeff"])], device="cuda", requires_grad=True)
    tau = G.tau_of(d, torch.float32, "cuda")
    out, _ = gta_amd.multihead_geometric_transform_attention(
        q, k, v, attn_fn=SimpleNamespace(scale=float(d["scale"]), tau=tau), f_dims=meta["f_dims"], reps=ex,
        trans_coeff=tc, v_transform=meta["v_transform"], kv_mode=kv_mode)
    (out.float() * torch.from_numpy(d["w"]).float().cuda()).sum().backward()
    torch.cuda.synchronize()
    for name, t in (("dq", q), ("dk", k), ("dv", v)):
        _check(t.grad.float().cpu(), torch.from_numpy(d[name]).float(), name)
    if meta["f_dims"].get("se3", 0) > 0:
        ref = float(d["dtrans_coeff"][0])
        got = float(tc.grad.item())
        if abs(got - ref) > 2e-2 * max(1.0, abs(ref)):   # (the plain bar holds for all but the most cancelling fixtures: see _dtc_sensitivity)
            sens = _dtc_sensitivity(d, meta)
            assert abs(got - ref) <= 2e-2 * max(1.0, abs(ref)) + 3.0 * sens, (got, ref, sens)
    if tau is not None:                                   # softmax: adjustable (layers.py:195-200)
        ref, got = float(d["dtau"][0]), float(tau.grad.item())
        assert abs(got - ref) <= 2e-2 * max(1.0, abs(ref)), (got, ref)


SHAPES = {
    "C1": (2, 4, 2, 64, 2, 64, {"se3": 32, "so2": 32}, 8, 0),
    "CL-enc": (1, 6, 2, 300, 2, 300, {"se3": 32, "so2": 32}, 8, 0),
    "CL-dec": (1, 6, 3, 853, 2, 300, {"se3": 32, "so2": 32}, 8, 0),
    "MS-enc": (1, 8, 5, 256, 5, 256, {"triv": 0, "se3": 48, "so3": 24, "so2": 24}, 6, 2),
    "MS-dec": (1, 8, 5, 512, 5, 256, {"triv": 0, "se3": 48, "so3": 24, "so2": 24}, 6, 2),
    "ragged": (1, 3, 3, 37, 2, 45, {"triv": 8, "se3": 16, "so2": 8}, 2, 0),
    "many-views": (2, 2, 12, 20, 9, 28, {"triv": 0, "se3": 48, "so3": 24, "so2": 24}, 6, 2),
    "wide": (1, 2, 2, 160, 2, 96, {"se3": 64, "so2": 64}, 16, 0),                # dh = 128
    "wide-ragged": (1, 2, 3, 50, 2, 70, {"triv": 8, "se3": 48, "so3": 24, "so2": 24}, 6, 2),   # dh = 104 -> padded 128
    "DT": (1, 16, 1, 1024, 1, 1024, {"so2": 64}, 16, 0),                           # BASELINE config 5: pure so2, one 32 x 32 "view"
}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_gradients_vs_oracle_autograd(shape, dtype):
    from oracle import gta_oracle as O
    B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = SHAPES[shape]
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, dtype, seed=7)
    if dtype == torch.bfloat16:
        q, k, v = q.bfloat16().float(), k.bfloat16().float(), v.bfloat16().float()
    g = torch.Generator().manual_seed(11)
    w = torch.randn(q.shape, generator=g)
    # oracle
    qo, ko, vo = (t.clone().requires_grad_() for t in (q, k, v))
    tco = torch.tensor([0.37], requires_grad=True)
    reps = O.encoder_reps(ak, ex)
    if cross:
        reps = O.decoder_reps(ak, ex, reps)
    out_o, _ = O.gta_attention(qo, ko, vo, f_dims, reps, tco)
    (out_o * w).sum().backward()
    # HIP
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    qd, kd, vd = (t.to(dtype).cuda().requires_grad_() for t in (q, k, v))
    tcd = torch.tensor([0.37], device="cuda", requires_grad=True)
    out = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=exd.get("gta_so3_degree", 0),
                                trans_coeff=tcd if f_dims.get("se3", 0) > 0 else None)
    (out.float() * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    _check(out.float().cpu(), out_o.detach(), "out", 2.5e-2, 1.2e-2)
    for name, a, b in (("dq", qd, qo), ("dk", kd, ko), ("dv", vd, vo)):
        _check(a.grad.float().cpu(), b.grad, name)
    if f_dims.get("se3", 0) > 0:
        ref, got = float(tco.grad.item()), float(tcd.grad.item())
        assert abs(got - ref) <= 3e-2 * max(1.0, abs(ref)), (got, ref)


@pytest.mark.parametrize("shape,dtype", [("MS-dec", torch.bfloat16), ("CL-enc", torch.float32), ("ragged", torch.float32)])
def test_tau_gradient_vs_oracle_autograd(shape, dtype):
    """`softmax: adjustable` at the BASELINE shapes: forward with tau != 1 and d loss / d tau from the dQ kernel's
    epilogue against autograd through the oracle."""
    from oracle import gta_oracle as O
    B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = SHAPES[shape]
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, dtype, seed=17)
    if dtype == torch.bfloat16:
        q, k, v = q.bfloat16().float(), k.bfloat16().float(), v.bfloat16().float()
    w = torch.randn(q.shape, generator=torch.Generator().manual_seed(19))
    qo, ko, vo = (t.clone().requires_grad_() for t in (q, k, v))
    tauo = torch.tensor([0.8], requires_grad=True)
    reps = O.encoder_reps(ak, ex)
    if cross:
        reps = O.decoder_reps(ak, ex, reps)
    out_o, _ = O.gta_attention(qo, ko, vo, f_dims, reps, 0.37, tau=tauo)
    (out_o * w).sum().backward()
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    qd, kd, vd = (t.to(dtype).cuda().requires_grad_() for t in (q, k, v))
    taud = torch.tensor([0.8], device="cuda", requires_grad=True)
    out = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=exd.get("gta_so3_degree", 0),
                                trans_coeff=0.37 if f_dims.get("se3", 0) > 0 else None, tau=taud)
    (out.float() * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    _check(out.float().cpu(), out_o.detach(), "out", 2.5e-2, 1.2e-2)
    for name, a, b in (("dq", qd, qo), ("dk", kd, ko), ("dv", vd, vo)):
        _check(a.grad.float().cpu(), b.grad, name)
    ref, got = float(tauo.grad.item()), float(taud.grad.item())
    assert abs(got - ref) <= 3e-2 * max(1.0, abs(ref)), (got, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", ["MS-enc", "MS-dec", "CL-dec", "ragged", "many-views", "wide-ragged"])
def test_backward_is_bit_reproducible(shape, dtype):
    """Two runs of the backward on the same inputs agree bit for bit: no atomics anywhere (the per-row D, d trans_coeff and d tau are
    fixed-order sums)."""
    B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = SHAPES[shape]
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, dtype, seed=23)
    w = torch.randn(q.shape, generator=torch.Generator().manual_seed(29)).cuda()
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    res = {}
    for run in range(2):
        qd, kd, vd = (t.to(dtype).cuda().requires_grad_() for t in (q, k, v))
        tcd = torch.tensor([0.37], device="cuda", requires_grad=True)
        out = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=exd.get("gta_so3_degree", 0),
                                    trans_coeff=tcd if f_dims.get("se3", 0) > 0 else None, kv_mode="prepass")
        (out.float() * w).sum().backward()
        torch.cuda.synchronize()
        res[run] = (qd.grad.float().cpu(), kd.grad.float().cpu(), vd.grad.float().cpu(), float(tcd.grad.item()) if tcd.grad is not None else 0.0)
    assert all(torch.equal(res[0][i], res[1][i]) for i in range(3)) and res[0][3] == res[1][3]


@pytest.mark.parametrize("shape", ["MS-enc", "MS-dec", "ragged-ms"])
def test_dkv64_stream_matches_keys32_kernel_bit_for_bit(shape):
    """gta_bwd_dkv64_kernel / gta_bwd_dq64_kernel (64 keys / 64 query rows per wave, the walks ONE generated instruction stream each:
    gen_bwd64.py) against gta_bwd_dkv_kernel / gta_bwd_dq_kernel (32 per wave, compiled): per (key, query) the same arithmetic and per
    accumulator the same order of tiles, so dq, dk and dv agree bit for bit (d trans_coeff sums the same per-token terms in another
    grouping).  The ragged shape's key side is not whole tiles: its dq comes from the compiled kernel in every run."""
    shapes = dict(SHAPES)
    shapes["ragged-ms"] = (2, 3, 3, 100, 3, 150, {"se3": 48, "so3": 24, "so2": 24}, 6, 2)      # Tq = 300 (5 query tiles, the last ragged), Tk = 450 (2 key blocks, ragged)
    B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = shapes[shape]
    assert sum(f_dims.values()) == 96
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, torch.bfloat16, seed=41)
    w = torch.randn(q.shape, generator=torch.Generator().manual_seed(43)).cuda()
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    res = {}
    for mode in ("prepass_bwd_keys32", "prepass_bwd_keys64", "prepass_bwd_keys64_split"):
        qd, kd, vd = (t.bfloat16().cuda().requires_grad_() for t in (q, k, v))
        tcd = torch.tensor([0.37], device="cuda", requires_grad=True)
        out = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=exd.get("gta_so3_degree", 0), trans_coeff=tcd, kv_mode=mode)
        (out.float() * w).sum().backward()
        torch.cuda.synchronize()
        res[mode] = (qd.grad.float().cpu(), kd.grad.float().cpu(), vd.grad.float().cpu(), float(tcd.grad.item()))
    a, b = res["prepass_bwd_keys32"], res["prepass_bwd_keys64"]
    assert torch.isfinite(b[1]).all() and torch.isfinite(b[2]).all() and b[1].abs().max() > 0
    assert torch.equal(a[0], b[0])
    assert torch.equal(a[2], b[2]), (a[2] - b[2]).abs().max()
    assert torch.equal(a[1], b[1]), (a[1] - b[1]).abs().max()
    assert abs(a[3] - b[3]) <= 1e-4 * max(1.0, abs(a[3])), (a[3], b[3])
    # the two generated kernels as ONE launch (gta_bwd_dqkv64_kernel: the default where both run and the dQ blocks are a multiple of 8 -- MS-enc,
    # MS-dec here) and as two (GTA_FLAG_BWD_SPLIT): the same workgroups doing the same work
    c = res["prepass_bwd_keys64_split"]
    assert all(torch.equal(b[i], c[i]) for i in range(3)) and b[3] == c[3]


@pytest.mark.parametrize("shape", ["MS-enc", "MS-dec"])
def test_generated_backward_kernels_vs_oracle_autograd(shape):
    """the generated dQ and dK/dV kernels (forced: the shapes are below their launch-size threshold) against autograd through the oracle, with
    the bounds of test_gradients_vs_oracle_autograd"""
    from oracle import gta_oracle as O
    B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = SHAPES[shape]
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, torch.bfloat16, seed=7)
    q, k, v = q.bfloat16().float(), k.bfloat16().float(), v.bfloat16().float()
    w = torch.randn(q.shape, generator=torch.Generator().manual_seed(11))
    qo, ko, vo = (t.clone().requires_grad_() for t in (q, k, v))
    tco = torch.tensor([0.37], requires_grad=True)
    reps = O.encoder_reps(ak, ex)
    if cross:
        reps = O.decoder_reps(ak, ex, reps)
    out_o, _ = O.gta_attention(qo, ko, vo, f_dims, reps, tco)
    (out_o * w).sum().backward()
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    qd, kd, vd = (t.bfloat16().cuda().requires_grad_() for t in (q, k, v))
    tcd = torch.tensor([0.37], device="cuda", requires_grad=True)
    out = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=exd.get("gta_so3_degree", 0), trans_coeff=tcd, kv_mode="prepass_bwd_keys64")
    (out.float() * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    _check(out.float().cpu(), out_o.detach(), "out", 2.5e-2, 1.2e-2)
    for name, a, b in (("dq", qd, qo), ("dk", kd, ko), ("dv", vd, vo)):
        _check(a.grad.float().cpu(), b.grad, name)
    ref, got = float(tco.grad.item()), float(tcd.grad.item())
    assert abs(got - ref) <= 3e-2 * max(1.0, abs(ref)), (got, ref)


@pytest.mark.parametrize("shape", ["MS-enc", "MS-dec"])
def test_generated_backward_is_bit_reproducible(shape):
    """two runs of the generated dQ / dK/dV kernels on the same inputs agree bit for bit (no atomics; d trans_coeff is a fixed-order sum)"""
    B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = SHAPES[shape]
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, torch.bfloat16, seed=23)
    w = torch.randn(q.shape, generator=torch.Generator().manual_seed(29)).cuda()
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    res = {}
    for run in range(2):
        qd, kd, vd = (t.bfloat16().cuda().requires_grad_() for t in (q, k, v))
        tcd = torch.tensor([0.37], device="cuda", requires_grad=True)
        out = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=exd.get("gta_so3_degree", 0), trans_coeff=tcd, kv_mode="prepass_bwd_keys64")
        (out.float() * w).sum().backward()
        torch.cuda.synchronize()
        res[run] = (qd.grad.float().cpu(), kd.grad.float().cpu(), vd.grad.float().cpu(), float(tcd.grad.item()))
    assert all(torch.equal(res[0][i], res[1][i]) for i in range(3)) and res[0][3] == res[1][3]


def test_backward_properties_at_the_bench_size():
    """Size-independent properties of the operator's backward at the BASELINE size the bench times (MSN encoder, B = 32 per GPU, bf16: the default
    kernel selection there is the joint launch of the two generated streams), where autograd through the oracle would take minutes:
    (1) every step of the backward is linear in dout and a factor 2 is exact in bf16 and fp32, so the gradients of 2 w are EXACTLY twice those of w;
    (2) replacing every extrinsic E_n by E_n g (a change of the global frame) leaves dq, dk, dv unchanged to bf16 accuracy."""
    from oracle import gta_oracle as O
    _, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = SHAPES["MS-enc"]
    B = 32
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, torch.bfloat16, seed=3)
    w = torch.randn(q.shape, generator=torch.Generator().manual_seed(4)).bfloat16().cuda()
    g = O.random_extrinsics(1, 2, torch.Generator().manual_seed(9))[:, 1:2]

    def grads(ex_in, wscale):
        exd = {kk: vv.cuda() for kk, vv in ex_in.items()}
        gta_amd.pre_compute_reps_encoder(ak, exd)
        packed = gta_amd.pack_reps(exd, f_dims)
        qd, kd, vd = (t.bfloat16().cuda().requires_grad_() for t in (q, k, v))
        tcd = torch.tensor([0.37], device="cuda", requires_grad=True)
        out = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=exd.get("gta_so3_degree", 0), trans_coeff=tcd)
        out.backward(w * wscale)
        torch.cuda.synchronize()
        return qd.grad, kd.grad, vd.grad, float(tcd.grad.item())

    a = grads(ex, 1.0)
    b = grads(ex, 2.0)
    for x, y in zip(a[:3], b[:3]):
        assert torch.isfinite(x.float()).all() and x.float().abs().max() > 0
        assert torch.equal(y.float(), 2.0 * x.float())
    assert abs(b[3] - 2.0 * a[3]) <= 1e-5 * max(1.0, abs(a[3]))
    c = grads(dict(ex, input_transforms=ex["input_transforms"] @ g), 1.0)
    for x, y, name in zip(a[:3], c[:3], ("dq", "dk", "dv")):
        _check(y.float().cpu(), x.float().cpu(), name)


@pytest.mark.parametrize("shape", ["MS-enc", "MS-dec", "CL-enc", "CL-dec", "DT"])
def test_bench_size_sampled_scenes_gradients_vs_oracle(shape):
    """The backward at the batch the bench times (B = 32 per GPU, bf16: the kernels that launch selects -- at the MSN shapes the joint launch of
    the generated streams): the whole batch on the device, two of its scenes through autograd over the oracle (a scene's dq, dk, dv depend on
    that scene alone; d trans_coeff sums over the batch and is covered by the small-batch tests)."""
    from oracle import gta_oracle as O
    _, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = SHAPES[shape]
    B = 32
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, torch.bfloat16, seed=13)
    q, k, v = q.bfloat16().float(), k.bfloat16().float(), v.bfloat16().float()
    w = torch.randn(q.shape, generator=torch.Generator().manual_seed(14)).bfloat16().float()
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    qd, kd, vd = (t.bfloat16().cuda().requires_grad_() for t in (q, k, v))
    tcd = torch.tensor([0.37], device="cuda", requires_grad=True) if f_dims.get("se3", 0) > 0 else None      # (DT: no se3 slab, no trans_coeff)
    out = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=exd.get("gta_so3_degree", 0), trans_coeff=tcd)
    out.backward(w.bfloat16().cuda())
    torch.cuda.synchronize()
    idx = torch.tensor([5, 27])
    qo, ko, vo = (t[idx].clone().requires_grad_() for t in (q, k, v))
    exs = {kk: vv[idx] for kk, vv in ex.items()}
    reps = O.encoder_reps(ak, exs)
    if cross:
        reps = O.decoder_reps(ak, exs, reps)
    out_o, _ = O.gta_attention(qo, ko, vo, f_dims, reps, torch.tensor([0.37]))
    (out_o * w[idx]).sum().backward()
    for name, a, b in (("dq", qd, qo), ("dk", kd, ko), ("dv", vd, vo)):
        _check(a.grad[idx.cuda()].float().cpu(), b.grad, name)


@pytest.mark.parametrize("shape", ["C1", "MS-enc", "MS-dec"])
def test_compiled_backward_joint_launch_matches_two_launches(shape):
    """the compiled dQ and dK/dV kernels as ONE launch (gta_bwd_dqkv_kernel: the default where neither generated stream runs and the dK/dV blocks
    are a multiple of 8 -- these shapes) against two launches (GTA_FLAG_BWD_SPLIT): the same workgroups doing the same work, bit for bit"""
    B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = SHAPES[shape]
    assert (B * H * ((Nk * Pk + 127) // 128)) % 8 == 0
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, torch.bfloat16, seed=31)
    w = torch.randn(q.shape, generator=torch.Generator().manual_seed(32)).cuda()
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    res = {}
    for mode in ("prepass", "prepass_bwd_split"):
        qd, kd, vd = (t.bfloat16().cuda().requires_grad_() for t in (q, k, v))
        tcd = torch.tensor([0.37], device="cuda", requires_grad=True)
        out = gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=exd.get("gta_so3_degree", 0), trans_coeff=tcd, kv_mode=mode)
        (out.float() * w).sum().backward()
        torch.cuda.synchronize()
        res[mode] = (qd.grad.float().cpu(), kd.grad.float().cpu(), vd.grad.float().cpu(), float(tcd.grad.item()))
    a, b = res["prepass"], res["prepass_bwd_split"]
    assert a[1].abs().max() > 0 and all(torch.equal(a[i], b[i]) for i in range(3)) and a[3] == b[3]
