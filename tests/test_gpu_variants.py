"""GPU: the reference's ablation variants through the generic path
(gta_rep_apply + gta_attn_fwd_plain): euclid similarity, t2 slab, and layouts the fused kernels refuse."""
from types import SimpleNamespace

import pytest
import torch

import gta_amd
from gta_amd import native
from oracle import gta_oracle as O
from tests import _golden as G
from tests import _hip_cases as C

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", ["euclid", "t2"])
def test_ablation_fixture_forward(case, dtype):
    d, meta = G.load("op_" + case)
    ex = G.extras_of(d, torch.float32, "cuda")
    q, k, v = (torch.from_numpy(d[n]).to(dtype).cuda() for n in "qkv")
    tc = torch.tensor([float(d["trans_coeff"])], device="cuda")
    out, _ = gta_amd.multihead_geometric_transform_attention(
        q, k, v, attn_fn=SimpleNamespace(scale=float(d["scale"])), f_dims=meta["f_dims"], reps=ex,
        trans_coeff=tc if meta["f_dims"].get("se3", 0) > 0 else None, v_transform=meta["v_transform"], euclid=meta["euclid"])
    torch.cuda.synchronize()
    st = C.err_stats(out.float().cpu(), torch.from_numpy(d["out"]).float())
    assert st["finite"] and st["max_abs"] <= 3e-2 * st["ref_max"] and st["rel_rms"] <= 1.5e-2, st


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_euclid_adjustable_softmax_fixture(dtype):
    """euclid similarity + `softmax: adjustable` (layers.py:195-224): out, dq, dk, dv, d trans_coeff and d tau of the
    generic path against the reference's autograd (fixture op_euclid_tau)."""
    d, meta = G.load("op_euclid_tau")
    ex = G.extras_of(d, torch.float32, "cuda")
    q, k, v = (torch.from_numpy(d[n]).to(dtype).cuda().requires_grad_() for n in "qkv")
    tc = torch.tensor([float(d["trans_coeff"])], device="cuda", requires_grad=True)
    tau = G.tau_of(d, torch.float32, "cuda")
    out, _ = gta_amd.multihead_geometric_transform_attention(
        q, k, v, attn_fn=SimpleNamespace(scale=float(d["scale"]), tau=tau), f_dims=meta["f_dims"], reps=ex,
        trans_coeff=tc, v_transform=meta["v_transform"], euclid=True)
    (out.float() * torch.from_numpy(d["w"]).float().cuda()).sum().backward()
    torch.cuda.synchronize()
    st = C.err_stats(out.float().cpu(), torch.from_numpy(d["out"]).float())
    assert st["finite"] and st["max_abs"] <= 3e-2 * st["ref_max"] and st["rel_rms"] <= 1.5e-2, st
    for name, t in (("dq", q), ("dk", k), ("dv", v)):
        st = C.err_stats(t.grad.float().cpu(), torch.from_numpy(d[name]).float())
        assert st["finite"] and st["max_abs"] <= 5e-2 * st["ref_max"] + 1e-6 and st["rel_rms"] <= 2.5e-2, (name, st)
    for name, t in (("dtrans_coeff", tc), ("dtau", tau)):
        ref, got = float(d[name].reshape(-1)[0]), float(t.grad.item())
        assert abs(got - ref) <= 5e-2 * max(1.0, abs(ref)), (name, got, ref)


def test_generic_path_matches_fused_on_a_fused_layout():
    """Same inputs through gta_rep_apply + plain attention and through the fused kernel."""
    from tests.test_gpu_forward import SHAPES
    from gta_amd import gta as G2
    B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3 = SHAPES["MS-dec"]
    q, k, v, ex, ak, cross = C.synth_inputs(1, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, torch.float32, seed=21)
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    tc = torch.tensor([0.3], device="cuda")
    fused = gta_amd.gta_attention(q.cuda(), k.cuda(), v.cuda(), f_dims, packed, so3_degree=2, trans_coeff=tc)
    gen = G2._generic_forward(q.cuda(), k.cuda(), v.cuda(), f_dims, packed, 2, tc, None, 96 ** -0.5, True, False)
    torch.cuda.synchronize()
    st = C.err_stats(gen.cpu(), fused.cpu())
    assert st["finite"] and st["rel_rms"] < 8e-3, st


def test_generic_path_backward_matches_reference_t2():
    """t2 ablation (runs/*/GTA/gta_t2): dq, dk, dv through gta_rep_apply_bwd + the plain backward against the
    reference's autograd gradients (fixture op_t2)."""
    d, meta = G.load("op_t2")
    ex = G.extras_of(d, torch.float32, "cuda")
    q, k, v = (torch.from_numpy(d[n]).float().cuda().requires_grad_() for n in "qkv")
    out, _ = gta_amd.multihead_geometric_transform_attention(
        q, k, v, attn_fn=SimpleNamespace(scale=float(d["scale"])), f_dims=meta["f_dims"], reps=ex, trans_coeff=None)
    (out * torch.from_numpy(d["w"]).float().cuda()).sum().backward()
    torch.cuda.synchronize()
    for name, t in (("dq", q), ("dk", k), ("dv", v)):
        st = C.err_stats(t.grad.cpu(), torch.from_numpy(d[name]).float())
        assert st["finite"] and st["max_abs"] <= 4e-2 * st["ref_max"] and st["rel_rms"] <= 2e-2, (name, st)


def test_generic_path_backward_matches_fused_backward():
    """On a layout both paths support, the generic path's gradients (incl. d trans_coeff) equal the fused backward's."""
    from gta_amd import gta as G2
    f_dims, so2, so3 = {"triv": 0, "se3": 48, "so3": 24, "so2": 24}, 6, 2
    q, k, v, ex, ak, cross = C.synth_inputs(1, 2, 2, 96, 2, 80, f_dims, so2, so3, torch.float32, seed=5)
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    w = torch.randn(1, 2, 192, 96, device="cuda")
    grads = {}
    for name in ("fused", "generic"):
        qq, kk, vv = (t.clone().cuda().requires_grad_() for t in (q, k, v))
        tc = torch.tensor([0.3], device="cuda", requires_grad=True)
        if name == "fused":
            o = gta_amd.gta_attention(qq, kk, vv, f_dims, packed, so3_degree=2, trans_coeff=tc)
        else:
            o = G2._generic_forward(qq, kk, vv, f_dims, packed, 2, tc, None, 96 ** -0.5, True, False)
        (o.float() * w).sum().backward()
        grads[name] = (qq.grad, kk.grad, vv.grad, tc.grad)
    torch.cuda.synchronize()
    for a, b_, nm in zip(grads["generic"][:3], grads["fused"][:3], ("dq", "dk", "dv")):
        st = C.err_stats(a.cpu(), b_.cpu())
        assert st["finite"] and st["rel_rms"] < 2e-2, (nm, st)
    ga, gb = grads["generic"][3].item(), grads["fused"][3].item()
    assert abs(ga - gb) <= 2e-2 * max(1.0, abs(gb)), (ga, gb)


def test_generic_path_backward_matches_reference_euclid():
    """euclid_sim ablation (runs/*/GTA/gta_euclid, gta_so3_euclid): affine SE(3) action on 3-vectors + the
    -|q'-k'|^2/2 similarity; dq, dk, dv, d trans_coeff against the reference's autograd (fixture op_euclid)."""
    d, meta = G.load("op_euclid")
    ex = G.extras_of(d, torch.float32, "cuda")
    q, k, v = (torch.from_numpy(d[n]).float().cuda().requires_grad_() for n in "qkv")
    tc = torch.tensor([float(d["trans_coeff"].reshape(-1)[0])], device="cuda", requires_grad=True)
    out, _ = gta_amd.multihead_geometric_transform_attention(
        q, k, v, attn_fn=SimpleNamespace(scale=float(d["scale"])), f_dims=meta["f_dims"], reps=ex, trans_coeff=tc,
        euclid=True)
    (out * torch.from_numpy(d["w"]).float().cuda()).sum().backward()
    torch.cuda.synchronize()
    for name, t in (("dq", q), ("dk", k), ("dv", v)):
        st = C.err_stats(t.grad.cpu(), torch.from_numpy(d[name]).float())
        assert st["finite"] and st["max_abs"] <= 5e-2 * st["ref_max"] and st["rel_rms"] <= 2.5e-2, (name, st)
    ref_tc = float(d["dtrans_coeff"].reshape(-1)[0])
    assert abs(tc.grad.item() - ref_tc) <= 5e-2 * max(1.0, abs(ref_tc)), (tc.grad.item(), ref_tc)


@pytest.mark.parametrize("case", ["cl_cross", "ms_self", "euclid", "t2"])
def test_attention_map_matches_reference(case):
    """return_attmap: the dense matrix against the fixture's `attn` (the reference's own softmax matrix)."""
    d, meta = G.load("op_" + case)
    ex = G.extras_of(d, torch.float32, "cuda")
    q, k = (torch.from_numpy(d[n]).float().cuda() for n in "qk")
    packed = gta_amd.pack_reps(ex, meta["f_dims"])
    L = len(ex["so3rep_q"]) if "so3rep_q" in ex else 0
    attn = gta_amd.attention_map(q, k, meta["f_dims"], packed, so3_degree=L,
                                 trans_coeff=float(d["trans_coeff"]) if meta["f_dims"].get("se3", 0) > 0 else None,
                                 scale=float(d["scale"]), euclid=meta["euclid"])
    ref = torch.from_numpy(d["attn"]).float()
    assert (attn.cpu() - ref).abs().max() < 2e-5
    assert (attn.sum(-1).cpu() - 1).abs().max() < 1e-5


def test_vecrep_attention_forward_backward():
    """elementwise_mul ablation (gta.py:282-298) vs the oracle, gradients included."""
    from oracle import gta_oracle as O
    g = torch.Generator().manual_seed(3)
    B, H, Tq, Tk, dh = 2, 3, 70, 90, 32
    q, k, v = (torch.randn(B, H, T, dh, generator=g) for T in (Tq, Tk, Tk))
    vq, vk, vi = torch.randn(B, Tq, dh, generator=g), torch.randn(B, Tk, dh, generator=g), torch.randn(B, Tq, dh, generator=g)
    w = torch.randn(B, H, Tq, dh, generator=g)
    qo, ko, vo = (t.clone().requires_grad_() for t in (q, k, v))
    ref, _ = O.vecrep_attention(qo, ko, vo, vq, vk, vi, dh ** -0.5)
    (ref * w).sum().backward()
    qd, kd, vd = (t.cuda().requires_grad_() for t in (q, k, v))
    out, _ = gta_amd.multihead_vecrep_attention(qd, kd, vd, attn_fn=SimpleNamespace(scale=dh ** -0.5),
                                                extras=dict(vecrep_q=vq.cuda(), vecrep_k=vk.cuda(), vecinvrep_q=vi.cuda()))
    (out * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    for name, a, b in (("out", out.detach(), ref.detach()), ("dq", qd.grad, qo.grad), ("dk", kd.grad, ko.grad), ("dv", vd.grad, vo.grad)):
        st = C.err_stats(a.cpu(), b)
        assert st["finite"] and st["rel_rms"] < 2e-2, (name, st)


def test_vecrep_fixture_forward_backward():
    """``multihead_vecrep_attention`` against the REFERENCE's own output and autograd gradients (fixture vecrep_attn.npz,
    gta.py:282-298 run on vectors from the reference's pre_compute_reps + rep_to_vec)."""
    from tests import _golden as G
    d, meta = G.load("vecrep_attn")
    q, k, v = (torch.from_numpy(d[n]).float().cuda().requires_grad_() for n in "qkv")
    ex = {n: torch.from_numpy(d[n]).float().cuda() for n in ("vecrep_q", "vecrep_k", "vecinvrep_q")}
    out, _ = gta_amd.multihead_vecrep_attention(q, k, v, attn_fn=SimpleNamespace(scale=float(d["scale"])), extras=ex)
    (out * torch.from_numpy(d["w"]).float().cuda()).sum().backward()
    torch.cuda.synchronize()
    for name, a, b in (("out", out.detach(), d["out"]), ("dq", q.grad, d["dq"]), ("dk", k.grad, d["dk"]), ("dv", v.grad, d["dv"])):
        st = C.err_stats(a.cpu(), torch.from_numpy(b).float())
        assert st["finite"] and st["rel_rms"] < 2e-2 and st["max_abs"] < 4e-2 * st["ref_max"], (name, st)


def test_t2_dict_shared_by_encoder_and_decoder():
    """Level-2 drop-in: ONE reference-style reps dict goes through an encoder call (q side = k side,
    600 tokens) and then a decoder call that replaced the q side with the target's tensors (more query tokens than keys).
    The packed coordinate / (cos,sin) tables cached in the dict must follow the replacement (a stale q-side
    t2 table of the encoder would be indexed out of bounds)."""
    from oracle import gta_oracle as O
    f_dims = {"so2": 8, "t2": 6}
    ak = {"f_dims": f_dims, "so2": 2, "so3": 0, "max_freq_h": 1, "max_freq_w": 1}
    g = torch.Generator().manual_seed(11)
    B, H, Nk, Pk, Nq, Pq = 1, 2, 2, 20, 3, 30
    ex = {"input_transforms": O.random_extrinsics(B, Nk, g).double(), "input_coord": torch.rand(B, Nk, Pk, 2, generator=g).double(),
          "target_transforms": O.random_extrinsics(B, Nq, g).double(), "target_coord": torch.rand(B, Nq, Pq, 2, generator=g).double()}
    dh = sum(f_dims.values())
    qe, ke, ve = (torch.randn(B, H, Nk * Pk, dh, generator=g) for _ in range(3))
    qd = torch.randn(B, H, Nq * Pq, dh, generator=g)
    reps = O.encoder_reps(ak, ex)                                 # reference-style tensors (t2rep_q/k, so2rep_q/k)
    ref_e, _ = O.gta_attention(qe.double(), ke.double(), ve.double(), f_dims, reps, None, True)
    shared = {kk: ([u.float().cuda() for u in vv] if isinstance(vv, list) else vv.float().cuda()) for kk, vv in reps.items()
              if torch.is_tensor(vv) or isinstance(vv, list)}
    fn = SimpleNamespace(scale=dh ** -0.5)
    out_e, _ = gta_amd.multihead_geometric_transform_attention(qe.cuda(), ke.cuda(), ve.cuda(), attn_fn=fn, f_dims=f_dims, reps=shared)
    reps_d = O.decoder_reps(ak, ex, reps)
    ref_d, _ = O.gta_attention(qd.double(), ke.double(), ve.double(), f_dims, reps_d, None, True)
    for kk, vv in reps_d.items():                                  # the decoder overwrites entries of the SAME dict
        if torch.is_tensor(vv):
            shared[kk] = vv.float().cuda()
    out_d, _ = gta_amd.multihead_geometric_transform_attention(qd.cuda(), ke.cuda(), ve.cuda(), attn_fn=fn, f_dims=f_dims, reps=shared)
    torch.cuda.synchronize()
    for name, a, b in (("encoder", out_e, ref_e), ("decoder", out_d, ref_d)):
        st = C.err_stats(a.float().cpu(), b.float())
        assert st["finite"] and st["rel_rms"] < 1.5e-2, (name, st)


def test_rep_tables_are_validated():
    """Tables of another batch / token count, on the host, or of the wrong dtype raise before any kernel sees them."""
    f_dims = {"se3": 16, "so2": 16}
    q, k, v, ex, ak, _ = C.synth_inputs(2, 2, 2, 20, 2, 20, f_dims, 4, 0, torch.float32, seed=1)
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    gta_amd.gta_attention(qd, kd, vd, f_dims, packed, trans_coeff=0.01)                       # fine
    for key, bad in (("vrep_q", packed["vrep_q"][:1]), ("cs_q", packed["cs_q"][:, :10]), ("cs_k", packed["cs_k"].cpu()),
                     ("vrep_k", packed["vrep_k"].double()), ("cs_q", packed["cs_q"][..., :4, :])):
        broken = dict(packed)
        broken[key] = bad
        with pytest.raises(gta_amd.native.GtaError):
            gta_amd.gta_attention(qd, kd, vd, f_dims, broken, trans_coeff=0.01)
    with pytest.raises(gta_amd.native.GtaError):
        gta_amd.gta_attention(qd, kd, vd, f_dims, packed, trans_coeff=torch.tensor([0.01]))    # host scalar


def test_elementwise_mul_module_runs_and_trains():
    ak = {"f_dims": {"se3": 16, "so2": 16}, "so2": 4, "so3": 0, "max_freq_h": 1, "max_freq_w": 1, "elementwise_mul": True}
    att = gta_amd.Attention(48, heads=2, dim_head=32, attn_args={"method": {"name": "gta", "args": ak}}).cuda()
    assert att.trans_coeff is None and att.rep_to_vec.in_features == 16 + 2 * 4 * 2 * 2
    q, k, v, ex, _, _ = C.synth_inputs(2, 2, 2, 20, 2, 20, {"se3": 16, "so2": 16}, 4, 0, torch.float32, seed=1)
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    x = torch.randn(2, 40, 48, device="cuda", requires_grad=True)
    y, attn = att(x, extras=exd, return_attmap=True)
    y.sum().backward()
    assert torch.isfinite(y).all() and torch.isfinite(x.grad).all() and attn.shape == (2, 2, 40, 40)


@pytest.mark.parametrize("variant", ["t2", "euclid"])
def test_ablation_transformer_trains_and_matches_oracle(variant):
    """A Transformer block with the gta_t2 / gta_euclid attention settings (runs/*/GTA/gta_t2, gta_euclid): device
    rep builders + generic path forward and backward against the CPU oracle module under the same weights."""
    if variant == "t2":
        f_dims, extra = {"so2": 8, "t2": 6}, {}
        dim, H, dh = 28, 2, 14
    else:
        f_dims, extra = {"se3": 6, "so2": 8}, {"euclid_sim": True}
        dim, H, dh = 28, 2, 14
    ak = {"f_dims": f_dims, "so2": 2, "so3": 0, "max_freq_h": 1, "max_freq_w": 1, **extra}
    aa = {"method": {"name": "gta", "args": ak}}
    torch.manual_seed(3)
    tr = gta_amd.Transformer(dim, 2, H, dh, 2 * dim, 0.0, True, None, False, aa)
    ot = O.OracleTransformer(dim, 2, H, dh, 2 * dim, 0.0, True, None, False, aa)
    ot.load_state_dict(tr.state_dict(), strict=True)
    g = torch.Generator().manual_seed(9)
    N, P = 2, 24
    ex = {"input_transforms": O.random_extrinsics(2, N, g), "input_coord": torch.rand(2, N, P, 2, generator=g)}
    x = torch.randn(2, N * P, dim, generator=g)
    w = torch.randn(2, N * P, dim, generator=g)
    xo = x.clone().requires_grad_()
    yo = ot(xo, None, O.encoder_reps(ak, ex))
    (yo * w).sum().backward()
    tr = tr.cuda()
    exd = {k: v.cuda() for k, v in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    xd = x.clone().cuda().requires_grad_()
    yd = tr(xd, None, exd)
    (yd * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    st = C.err_stats(yd.detach().cpu(), yo.detach())
    assert st["finite"] and st["rel_rms"] < 1.5e-2, st
    st = C.err_stats(xd.grad.cpu(), xo.grad)
    assert st["finite"] and st["rel_rms"] < 4e-2, st
    for (n, p), (_, po) in zip(tr.named_parameters(), ot.named_parameters()):
        if n.endswith("trans_coeff"):
            continue
        st = C.err_stats(p.grad.cpu(), po.grad)
        assert st["finite"] and st["max_abs"] <= 6e-2 * max(st["ref_max"], 1e-3) + 1e-5, (n, st)
