"""GPU: the reference's ablation variants through the generic path
(gta_rep_apply + gta_attn_fwd_plain): euclid similarity, t2 slab, and layouts the fused kernels refuse."""
from types import SimpleNamespace

import pytest
import torch

import gta_amd
from gta_amd import native
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


def test_generic_path_is_forward_only():
    d, meta = G.load("op_t2")
    ex = G.extras_of(d, torch.float32, "cuda")
    q, k, v = (torch.from_numpy(d[n]).float().cuda().requires_grad_() for n in "qkv")
    with pytest.raises(native.GtaError):
        gta_amd.multihead_geometric_transform_attention(q, k, v, attn_fn=SimpleNamespace(scale=float(d["scale"])),
                                                        f_dims=meta["f_dims"], reps=ex, trans_coeff=None)
