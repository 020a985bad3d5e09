"""GPU parity of the rep builders (encoder.py:183-265, decoder.py:247-353) against the oracle."""
import pytest
import torch

import gta_amd
from gta_amd import native
from oracle import gta_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("so3", [0, 1, 2])
@pytest.mark.parametrize("n_views", [1, 5, 37, 160])
def test_view_reps_match_oracle(n_views, so3):
    """E, inverse(E), D^1, D^2 per view.  Tolerances: copies exact; the inverse 1e-5 relative to its largest entry
    (the reference's own fp32 LU is not better); Wigner entries 5e-6 absolute (|D| <= 1)."""
    g = torch.Generator().manual_seed(11 + n_views)
    E = O.random_extrinsics(1, n_views, g)                      # [1, N, 4, 4]
    got = native.build_view_reps(E.cuda(), so3).cpu()[0]         # [N, 72]
    se3rep, inv, Ds = O.build_view_reps(E.double(), so3)
    assert torch.equal(got[:, native.VREP_INV:native.VREP_INV + 16].reshape(-1, 4, 4), E[0])
    ref_inv = se3rep[0].float()
    err = (got[:, native.VREP_REP:native.VREP_REP + 16].reshape(-1, 4, 4) - ref_inv).abs().amax()
    assert err <= 1e-5 * ref_inv.abs().amax(), err
    if so3 >= 1:
        d1 = got[:, native.VREP_D1:native.VREP_D1 + 9].reshape(-1, 3, 3)
        assert (d1 - Ds[0].reshape(-1, 3, 3).float()).abs().amax() <= 5e-6
    else:
        assert got[:, native.VREP_D1:native.VREP_D1 + 9].abs().amax() == 0
    if so3 >= 2:
        d2 = got[:, native.VREP_D2:native.VREP_D2 + 25].reshape(-1, 5, 5)
        assert (d2 - Ds[1].reshape(-1, 5, 5).float()).abs().amax() <= 5e-6
    else:
        assert got[:, native.VREP_D2:native.VREP_D2 + 25].abs().amax() == 0
    assert got[:, native.VREP_D2 + 25:].abs().amax() == 0         # padding


def test_view_reps_gimbal_rows_follow_the_reference_formula():
    """R22 = +1 (identity rotation, translation only) takes the reference's masked branch (wigner_d.py:43-45)."""
    E = torch.eye(4).repeat(1, 3, 1, 1)
    E[0, 1, :3, 3] = torch.tensor([0.3, -1.2, 2.0])
    c, s_ = torch.cos(torch.tensor(0.7)), torch.sin(torch.tensor(0.7))
    E[0, 2, :2, :2] = torch.tensor([[c, -s_], [s_, c]])       # rotation about z: R22 stays 1
    got = native.build_view_reps(E.cuda(), 2).cpu()[0]
    _, _, Ds = O.build_view_reps(E.double(), 2)
    assert (got[:, native.VREP_D1:native.VREP_D1 + 9].reshape(-1, 3, 3) - Ds[0].reshape(-1, 3, 3).float()).abs().amax() <= 5e-6
    assert (got[:, native.VREP_D2:native.VREP_D2 + 25].reshape(-1, 5, 5) - Ds[1].reshape(-1, 5, 5).float()).abs().amax() <= 5e-6


def test_view_reps_r22_minus_one_rows_follow_the_reference_formula():
    """R22 = -1 takes the reference's other masked branch, gamma1 = atan2(-R10, -R00) (wigner_d.py:46-47) -- whose
    output is NOT the closed-form representation (a quirk kept on purpose).  The rotations of
    fixture wigner.npz (the last one is that gimbal case) go through the HIP builder as inverse(E) blocks and must
    reproduce the REFERENCE's matrices, quirk included."""
    from tests import _golden as G
    d, _ = G.load("wigner")
    R = torch.from_numpy(d["R"])                                   # [n, 3, 3] fp64; R = inverse(E)[:3,:3]
    n = R.shape[0]
    assert abs(float(R[-1, 2, 2]) + 1.0) < 1e-12                    # the R22 = -1 case is in the fixture
    E = torch.eye(4, dtype=torch.float64).repeat(n, 1, 1)
    E[:, :3, :3] = R.transpose(-1, -2)                              # E = inverse of [R | 0]
    got = native.build_view_reps(E.float().reshape(1, n, 4, 4).cuda(), 2).cpu()[0].double()
    D1 = got[:, native.VREP_D1:native.VREP_D1 + 9].reshape(n, 3, 3)
    D2 = got[:, native.VREP_D2:native.VREP_D2 + 25].reshape(n, 5, 5)
    assert (D1 - torch.from_numpy(d["D1"])).abs().amax() <= 2e-5
    assert (D2 - torch.from_numpy(d["D2"])).abs().amax() <= 4e-5
    # the quirk is visible: on that row the reference (and the builder) is far from the closed form
    D1c, _ = O.wigner_d_closed_form(R)
    assert (D1[-1] - D1c[-1]).abs().max() > 0.5


def test_flattened_reps_match_reference():
    """``flattened_rep_q`` / ``flattened_invrep_q`` of the elementwise_mul ablation (encoder.py:200-206,238-243,263-265)
    rebuilt from the HIP tables against the reference's tensors (fixture vecrep_attn.npz)."""
    import gta_amd
    from tests import _golden as G
    d, meta = G.load("vecrep_attn")
    ak = {"f_dims": meta["f_dims"], "so2": meta["so2"], "so3": 0, "max_freq_h": 1, "max_freq_w": 1, "elementwise_mul": True}
    ex = {"input_transforms": torch.from_numpy(d["extras.input_transforms"]).float().cuda(),
          "input_coord": torch.from_numpy(d["extras.input_coord"]).float().cuda()}
    gta_amd.pre_compute_reps_encoder(ak, ex)
    for key in ("flattened_rep_q", "flattened_rep_k", "flattened_invrep_q"):
        assert (ex[key].cpu().double() - torch.from_numpy(d["extras." + key])).abs().max() < 5e-6, key


@pytest.mark.parametrize("shared", [False, True])
def test_so2_table_matches_oracle(shared):
    g = torch.Generator().manual_seed(5)
    coord = torch.rand(2, 300, 2, generator=g)
    F = 6
    got = native.build_so2_table(coord.cuda(), F, 1.0, 1.0, shared).cpu()          # [B, T, 2F, 2]
    th = O.so2_angles(coord, F, (1.0, 1.0), shared_freqs=shared)                    # [B, T, 2F]
    assert (got[..., 0] - torch.cos(th)).abs().amax() <= 2e-6
    assert (got[..., 1] - torch.sin(th)).abs().amax() <= 2e-6


def test_fused_builder_equals_the_two_calls():
    g = torch.Generator().manual_seed(3)
    E = O.random_extrinsics(4, 5, g).cuda()
    coord = torch.rand(4, 1280, 2, generator=g).cuda()
    vrep, cs = native.build_reps(E, 2, coord, 6, 1.0, 1.0, False)
    assert torch.equal(vrep, native.build_view_reps(E, 2))
    assert torch.equal(cs, native.build_so2_table(coord, 6, 1.0, 1.0, False))
