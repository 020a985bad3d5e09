"""CPU: the oracle restatement reproduces every committed reference fixture (fp64, ~1e-12)."""
import numpy as np
import pytest
import torch

from oracle import gta_oracle as O
from tests import _golden as G

TOL = 1e-10


@pytest.mark.parametrize("case", G.list_cases("op_"))
def test_operator_matches_reference(case):
    d, meta = G.load("op_" + case)
    ex = G.extras_of(d)
    ak = G.attn_kwargs_of(meta)
    reps = O.encoder_reps(ak, ex)
    if meta["cross"]:
        reps = O.decoder_reps(ak, ex, reps)
    # (one fixture is float32 -- op_ms_id_so3: the reference's id_so3 identities are float32 and its einsum refuses float64 operands,
    #  oracle/make_golden.py -- and is held to float32 round-off; every other one to 1e-10)
    TOL = 2e-5 if d["q"].dtype == np.float32 else 1e-10
    # rep builders against the reference's pre_compute_reps output
    for key in ("se3rep_q", "se3rep_k", "inv_se3rep_q", "so2rep_q", "so2rep_k", "t2rep_q", "inv_t2rep_q"):
        if key in ex:
            assert (reps[key] - ex[key]).abs().max() < TOL, key
    if "so3rep_q" in ex:
        for a, b in zip(reps["so3rep_q"], ex["so3rep_q"]):
            assert (a - b).abs().max() < TOL
        for a, b in zip(reps["so3rep_k"], ex["so3rep_k"]):
            assert (a - b).abs().max() < TOL
    q, k, v = (torch.from_numpy(d[n]).double().requires_grad_() for n in "qkv")
    tc = torch.tensor([float(d["trans_coeff"])], dtype=torch.float64, requires_grad=True)
    tau = G.tau_of(d, torch.float64)            # softmax: adjustable fixtures carry tau and d tau
    out, attn = O.gta_attention(q, k, v, meta["f_dims"], reps, tc, meta["v_transform"], meta["euclid"],
                                float(d["scale"]), 1.0 if tau is None else tau)
    (out * torch.from_numpy(d["w"]).double()).sum().backward()
    assert np.abs(out.detach().numpy() - d["out"]).max() < TOL
    assert np.abs(attn.detach().numpy() - d["attn"]).max() < TOL
    for n, t in (("dq", q), ("dk", k), ("dv", v)):
        assert np.abs(t.grad.numpy() - d[n]).max() < TOL, n
    if meta["f_dims"].get("se3", 0) > 0:
        assert np.abs(tc.grad.numpy() - d["dtrans_coeff"]).max() < TOL
    if tau is not None:
        assert np.abs(tau.grad.numpy() - d["dtau"]).max() < TOL


@pytest.mark.parametrize("case", G.list_cases("mod_"))
def test_transformer_module_matches_reference(case):
    d, meta = G.load("mod_" + case)
    ex = G.extras_of(d)
    ak = {"f_dims": meta["f_dims"], "so2": meta["so2"], "so3": meta["so3"], "max_freq_h": 1, "max_freq_w": 1}
    aa = {"method": {"name": "gta", "args": ak}}
    tr = O.OracleTransformer(meta["dim"], meta["depth"], meta["H"], meta["dh"], 2 * meta["dim"], 0.0,
                             not meta["cross"], meta["kv_dim"], False, aa).double()
    sd = {k[len("param."):]: torch.from_numpy(v) for k, v in d.items() if k.startswith("param.")}
    tr.load_state_dict(sd, strict=True)      # identical state-dict keys to the reference Transformer
    reps = O.encoder_reps(ak, ex)
    if meta["cross"]:
        reps = O.decoder_reps(ak, ex, reps)
    x = torch.from_numpy(d["x"]).requires_grad_()
    z = torch.from_numpy(d["z"]) if "z" in d else None
    y = tr(x, z, reps)
    (y * torch.from_numpy(d["w"])).sum().backward()
    assert np.abs(y.detach().numpy() - d["y"]).max() < TOL
    assert np.abs(x.grad.numpy() - d["dx"]).max() < TOL
    for n, p in tr.named_parameters():
        assert np.abs(p.grad.numpy() - d["grad." + n]).max() < TOL, n


def test_vecrep_operator_matches_reference():
    """``multihead_vecrep_attention`` (gta.py:282-298, the elementwise_mul ablation) with autograd gradients: fixture
    vecrep_attn.npz was produced by the reference function on vectors from the reference's ``pre_compute_reps`` +
    ``rep_to_vec`` (oracle/make_golden.py::vecrep_case)."""
    d, meta = G.load("vecrep_attn")
    q, k, v = (torch.from_numpy(d[n]).requires_grad_() for n in "qkv")
    vq, vk, vi = (torch.from_numpy(d[n]) for n in ("vecrep_q", "vecrep_k", "vecinvrep_q"))
    out, _ = O.vecrep_attention(q, k, v, vq, vk, vi, float(d["scale"]))
    (out * torch.from_numpy(d["w"]).double()).sum().backward()
    assert np.abs(out.detach().numpy() - d["out"]).max() < TOL
    for n, t in (("dq", q), ("dk", k), ("dv", v)):
        assert np.abs(t.grad.numpy() - d[n]).max() < TOL, n
    # and the vectors themselves are rep_to_vec of the reference's flattened reps
    W, b = torch.from_numpy(d["rep_to_vec.weight"]), torch.from_numpy(d["rep_to_vec.bias"])
    assert (torch.from_numpy(d["extras.flattened_rep_q"]) @ W.T + b - vq).abs().max() < TOL


def test_wigner_euler_matches_reference():
    d, _ = G.load("wigner")
    R = torch.from_numpy(d["R"])
    Ds = O.wigner_d_euler(2, R)
    assert np.abs(Ds[1].numpy() - d["D1"]).max() < 1e-12
    assert np.abs(Ds[2].numpy() - d["D2"]).max() < 1e-12
    # closed form == Euler form away from the reference's R22=-1 branch (last row of the fixture)
    D1, D2 = O.wigner_d_closed_form(R[:-1])
    assert np.abs(D1.numpy() - d["D1"][:-1]).max() < 1e-9
    assert np.abs(D2.numpy() - d["D2"][:-1]).max() < 1e-9


def test_wigner_properties():
    g = torch.Generator().manual_seed(3)
    R1 = O.random_extrinsics(1, 9, g, torch.float64)[0, 1:, :3, :3]
    R2 = O.random_extrinsics(1, 9, g, torch.float64)[0, 1:, :3, :3]
    for l in (1, 2):
        J = O.J_MATRICES[l]
        assert torch.allclose(J, J.T) and torch.allclose(J @ J, torch.eye(2 * l + 1, dtype=J.dtype))
        Da, Db, Dab = (O.wigner_d_euler(2, R)[l] for R in (R1, R2, R1 @ R2))
        assert (Da @ Db - Dab).abs().max() < 1e-12                   # homomorphism
        assert (Da @ Da.transpose(-1, -2) - torch.eye(2 * l + 1, dtype=Da.dtype)).abs().max() < 1e-12
    assert (O.wigner_d_euler(2, torch.eye(3, dtype=torch.float64)[None])[2][0]
            - torch.eye(5, dtype=torch.float64)).abs().max() < 1e-12


def test_so2_tables_bit_identical():
    d, _ = G.load("so2_tables")
    coord = torch.from_numpy(d["coord"])
    for key, ref in d.items():
        if not key.startswith("so2_F"):
            continue
        F = int(key.split("_")[1][1:])
        mf = (int(key.split("_")[2][2]), int(key.split("_")[2][3]))
        sh = key.endswith("sh1")
        assert np.array_equal(O.make_so2_reps(coord, F, mf, sh).numpy(), ref), key
    assert np.array_equal(O.make_t2_reps(coord).numpy(), d["t2"])


def test_known_answer_properties():
    """Reference-derived invariants: global-frame invariance; triv-only == plain softmax."""
    g = torch.Generator().manual_seed(5)
    B, H, N, P, dt = 2, 2, 3, 5, torch.float64
    f_dims = {"se3": 8, "so3": 8, "so2": 8}
    ak = {"f_dims": f_dims, "so2": 2, "so3": 2, "max_freq_h": 1, "max_freq_w": 1}
    E = O.random_extrinsics(B, N, g, dt)
    coord = torch.rand(B, N, P, 2, generator=g, dtype=dt)
    q, k, v = (torch.randn(B, H, N * P, 24, generator=g, dtype=dt) for _ in range(3))
    out0, _ = O.gta_attention(q, k, v, f_dims, O.encoder_reps(ak, {"input_transforms": E, "input_coord": coord}), 1.0)
    gl = O.random_extrinsics(1, 2, g, dt)[:, 1:2]                                  # random global SE(3)
    out1, _ = O.gta_attention(q, k, v, f_dims,
                              O.encoder_reps(ak, {"input_transforms": E @ gl, "input_coord": coord}), 1.0)
    assert (out0 - out1).abs().max() < 1e-10
    out2, attn2 = O.gta_attention(q, k, v, {"triv": 24}, {}, 1.0)
    ref = torch.softmax(q @ k.transpose(-1, -2) * 24 ** -0.5, -1) @ v
    assert (out2 - ref).abs().max() < 1e-12


@pytest.mark.parametrize("fixture", ["srt_ms_tiny", "srt_ms_rays"])
def test_srt_wrapper_matches_reference(fixture):
    """OracleSRT (encoder / decoder wrappers, gta path) against the reference TransformingSRT fixtures (2 x 5 and 2 x 128 rays
    per scene): its own weights, rendered pixels, per-sample loss, PSNR and every parameter gradient."""
    import ast
    import numpy as np
    d, _ = G.load(fixture)
    cfg = ast.literal_eval(str(np.load(G.GOLDEN + f"/{fixture}.npz")["meta"]))
    om = O.OracleSRT(cfg).double()
    sd = {k[len("param."):]: torch.from_numpy(v) for k, v in d.items() if k.startswith("param.")}
    om.load_state_dict(sd, strict=True)
    ex = {k[len("extras."):]: torch.from_numpy(v) for k, v in d.items() if k.startswith("extras.")}
    t = lambda n: torch.from_numpy(d[n]).double()
    pred = om(t("images"), t("cam_in"), t("rays_in"), t("cam_t"), t("rays_t"), ex)
    loss = ((pred - t("target").flatten(1, 2)) ** 2).mean((1, 2))
    loss.sum().backward()
    assert (pred - t("pred")).abs().max() < 1e-9
    assert (loss - t("loss")).abs().max() < 1e-10
    assert (O.mse2psnr(loss.detach()) - t("psnr")).abs().max() < 1e-8
    for n, p in om.named_parameters():
        assert (p.grad - t("grad." + n)).abs().max() < 1e-9, n
