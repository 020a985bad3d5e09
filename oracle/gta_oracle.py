"""CPU oracle for the GTA attention hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch CPU ops, the algorithm of the reference's
geometric-transform attention (autonomousvision/gta).  It exists so that the HIP
kernels in ``gta_amd/csrc`` can be parity-checked; it is never imported by the
product package ``gta_amd``.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.

Pinning: every function here is checked against outputs of the reference itself
(``source/utils/gta.py``, ``source/layers.py`` imported in the build container) through
the fixtures under ``tests/golden/`` -- see ``oracle/make_golden.py``.  The one
exception is the Pinchon-Hoggan ``J`` data behind the Wigner-D matrices: the
reference loads it from ``J_dense.pt`` which is absent from the checkout, so the J
*values* are **parity unpinned** (the Euler-angle/Z-matrix code around them is pinned).

Reference citations are ``file:line`` relative to the reference checkout.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

# Fixed channel order of the per-head slabs (gta.py:115-122).
GROUP_ORDER = ("triv", "se3", "so3", "so2", "t2")


# --------------------------------------------------------------------------------------
# coordinate grids and per-token reps
# --------------------------------------------------------------------------------------
def make_2dcoord(H: int, W: int) -> torch.Tensor:
    """[H, W, 2] grid with entry (i, j) = (i/H, j/W)  (gta.py:9-16)."""
    r = torch.arange(H, dtype=torch.float32) / H
    c = torch.arange(W, dtype=torch.float32) / W
    return torch.stack(torch.meshgrid(r, c, indexing="ij"), dim=-1)


def so2_frequencies(nfreqs: int, shared_freqs: bool = False, device=None) -> torch.Tensor:
    """freq_f = 2^(f+1) / 2^F for f = 0..F-1, or all ones (gta.py:57-61)."""
    if shared_freqs:
        return torch.ones(nfreqs, device=device)
    return (2.0 ** torch.arange(1.0, nfreqs + 1.0, device=device)) / (2.0 ** float(nfreqs))


def so2_angles(coord: torch.Tensor, nfreqs: int, max_freqs: Sequence[float] = (1, 1),
               shared_freqs: bool = False) -> torch.Tensor:
    """theta[..., c] with c = 2*f + d (d = coordinate axis), in the reference's op order.

    gta.py:62-63 forms ``max_freq_d * 2 * pi`` as a Python double, multiplies it into the
    fp32 tensor ``coord_d * freq_f``; gta.py:68 + encoder.py:195 interleave the axes so that
    block index = f * dim + d.
    """
    freqs = so2_frequencies(nfreqs, shared_freqs, coord.device).to(coord.dtype)
    per_axis = []
    for d in range(coord.shape[-1]):
        prod = coord[..., d:d + 1] * freqs            # [..., F]
        per_axis.append((max_freqs[d] * 2 * math.pi) * prod)
    return torch.stack(per_axis, dim=-1).flatten(-2, -1)  # [..., F*dim]


def make_so2_reps(coord, nfreqs, max_freqs=(1, 1), shared_freqs=False) -> torch.Tensor:
    """[..., 2F, 2, 2] rotation blocks [[cos, -sin], [sin, cos]]  (gta.py:64-68)."""
    th = so2_angles(coord, nfreqs, max_freqs, shared_freqs)
    c, s = torch.cos(th), torch.sin(th)
    return torch.stack([torch.stack([c, -s], -1), torch.stack([s, c], -1)], -2)


def make_t2_reps(coord: torch.Tensor) -> torch.Tensor:
    """[..., 3, 3] matrices [[1,0,0],[0,1,0],[cx,cy,1]]  (gta.py:72-89)."""
    shape = coord.shape[:-1]
    T = torch.eye(3, dtype=coord.dtype, device=coord.device).expand(*shape, 3, 3).clone()
    T[..., 2, 0] = coord[..., 0]
    T[..., 2, 1] = coord[..., 1]
    return T


def scale_mask(trans_coeff, device=None, dtype=torch.float32) -> torch.Tensor:
    """4x4 mask: ones, translation column (rows 0..2) = c, last row = [0,0,0,1] (gta.py:40-44)."""
    m = torch.ones(4, 4, device=device, dtype=dtype)
    m[3, :3] = 0.0
    col = torch.ones(3, device=device, dtype=dtype) * trans_coeff
    rows = [torch.cat([m[i, :3], col[i:i + 1]]) for i in range(3)]
    return torch.stack(rows + [m[3]], 0)


# --------------------------------------------------------------------------------------
# SO(3) irreps (Wigner-D), degrees 1 and 2
# --------------------------------------------------------------------------------------
_SQ3 = math.sqrt(3.0)

# J matrices in the reference's real-SH basis (wigner_d.py:16-25 fixes the basis:
# Y_1 = (y, z, x), Y_2 = (sqrt3 xy, sqrt3 yz, (3z^2-r^2)/2, sqrt3 xz, sqrt3/2 (x^2-y^2))).
# J_l = D^l(rotation by pi about (0,1,1)/sqrt2).  Values are a restatement of the
# Pinchon-Hoggan construction, NOT read from the (absent) J_dense.pt -> parity unpinned.
J_MATRICES = {
    0: torch.tensor([[1.0]], dtype=torch.float64),
    1: torch.tensor([[0.0, 1.0, 0.0],
                     [1.0, 0.0, 0.0],
                     [0.0, 0.0, -1.0]], dtype=torch.float64),
    2: torch.tensor([[0.0, 0.0, 0.0, -1.0, 0.0],
                     [0.0, 1.0, 0.0, 0.0, 0.0],
                     [0.0, 0.0, -0.5, 0.0, -_SQ3 / 2],
                     [-1.0, 0.0, 0.0, 0.0, 0.0],
                     [0.0, 0.0, -_SQ3 / 2, 0.0, 0.5]], dtype=torch.float64),
}


def _z_rot(angle: torch.Tensor, l: int) -> torch.Tensor:
    """cos(m a) on the diagonal, sin(m a) on the anti-diagonal, m = l..-l (wigner_d.py:16-25)."""
    n = 2 * l + 1
    out = angle.new_zeros(angle.shape[0], n, n)
    m = torch.arange(l, -l - 1, -1, dtype=angle.dtype, device=angle.device)[None]
    idx = torch.arange(n)
    out[:, idx, n - 1 - idx] = torch.sin(m * angle[:, None])
    out[:, idx, idx] = torch.cos(m * angle[:, None])
    return out


def rotmat_to_zyz(R: torch.Tensor, eps: float = 1e-5):
    """ZYZ Euler angles with the reference's gimbal masks (wigner_d.py:37-49)."""
    g1 = torch.atan2(R[..., 2, 1], -R[..., 2, 0])
    g2 = torch.atan2(torch.sqrt(R[..., 0, 2] ** 2 + R[..., 1, 2] ** 2), R[..., 2, 2])
    g3 = torch.atan2(R[..., 1, 2], R[..., 0, 2])
    up = (torch.abs(R[..., 2, 2] - 1) < eps).to(torch.float32)
    dn = (torch.abs(R[..., 2, 2] + 1) < eps).to(torch.float32)
    reg = (1 - up) * (1 - dn)
    g1 = reg * g1 + up * torch.atan2(R[..., 1, 0], R[..., 0, 0]) \
        + dn * torch.atan2(-R[..., 1, 0], -R[..., 0, 0])
    g3 = reg * g3
    return g1, g2, g3


def wigner_d_euler(max_degree: int, R: torch.Tensor) -> List[torch.Tensor]:
    """D^l = Z(g3) J Z(g2) J Z(g1) for l = 0..max_degree (wigner_d.py:28-35,52-58)."""
    g1, g2, g3 = rotmat_to_zyz(R)
    mats = []
    for l in range(max_degree + 1):
        J = J_MATRICES[l].to(g1.dtype).to(g1.device)
        mats.append(_z_rot(g3, l) @ J @ _z_rot(g2, l) @ J @ _z_rot(g1, l))
    return mats


def _y2_basis(dtype=torch.float64) -> torch.Tensor:
    """Symmetric 3x3 matrices A_i with Y_2,i(x) = x^T A_i x in the reference basis."""
    A = torch.zeros(5, 3, 3, dtype=dtype)
    h = _SQ3 / 2
    A[0, 0, 1] = A[0, 1, 0] = h            # sqrt3 x y
    A[1, 1, 2] = A[1, 2, 1] = h            # sqrt3 y z
    A[2, 0, 0] = A[2, 1, 1] = -0.5         # (3 z^2 - r^2) / 2
    A[2, 2, 2] = 1.0
    A[3, 0, 2] = A[3, 2, 0] = h            # sqrt3 x z
    A[4, 0, 0] = h                         # sqrt3/2 (x^2 - y^2)
    A[4, 1, 1] = -h
    return A


def wigner_d_closed_form(R: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(D^1, D^2) without Euler angles: Y_l(R x) = D^l(R) Y_l(x).

    D^1 = P R P^T with P the (x,y,z)->(y,z,x) permutation; D^2_ij = <R^T A_i R, A_j>/<A_j,A_j>.
    This is what the HIP rep builder computes (no atan2, no gimbal branch).
    """
    perm = [1, 2, 0]
    D1 = R[..., perm, :][..., :, perm]
    A = _y2_basis(R.dtype).to(R.device)
    M = torch.einsum("...ki,akl,...lj->...aij", R, A, R)       # R^T A_a R
    D2 = torch.einsum("...aij,bij->...ab", M, A) / 1.5          # <A_b, A_b> = 3/2
    return D1, D2


# --------------------------------------------------------------------------------------
# rep builders (encoder.py:183-265, decoder.py:247-353)
# --------------------------------------------------------------------------------------
def build_view_reps(transforms: torch.Tensor, so3_degree: int = 0, wigner: str = "euler"):
    """From extrinsics [B,N,4,4]: (se3rep = inv(E), inv_se3rep = E, [D^1..D^L] of inv(E)[:3,:3])."""
    se3rep = torch.linalg.inv(transforms)
    Ds: List[torch.Tensor] = []
    if so3_degree > 0:
        B, N = transforms.shape[:2]
        R = se3rep[..., :3, :3].flatten(0, 1)
        if wigner == "euler":
            Ds = [D.reshape(B, N, D.shape[-2], D.shape[-1])
                  for D in wigner_d_euler(so3_degree, R)[1:]]
        else:
            assert so3_degree <= 2
            Ds = [D.reshape(B, N, D.shape[-2], D.shape[-1])
                  for D in wigner_d_closed_form(R)[:so3_degree]]
    return se3rep, transforms, Ds


def _so3_override(attn_kwargs: dict, Ds: List[torch.Tensor]) -> List[torch.Tensor]:
    """The two so3 ablation knobs of the rep builders (encoder.py:250-258, decoder.py:337-345): ``zeroout_so3`` replaces every D^l by
    zeros, ``id_so3`` (checked second) by identities; otherwise D^l as computed."""
    if attn_kwargs.get("zeroout_so3", False):
        return [torch.zeros_like(D) for D in Ds]
    if attn_kwargs.get("id_so3", False):
        return [torch.eye(D.shape[-1], dtype=D.dtype, device=D.device).expand_as(D).clone() for D in Ds]
    return Ds


def encoder_reps(attn_kwargs: dict, extras: dict, wigner: str = "euler") -> dict:
    """Self-attention reps: q-side == k-side (encoder.py:183-265).  Returns a new dict."""
    f = attn_kwargs["f_dims"]
    reps = {}
    if f.get("so2", 0) > 0:
        coord = extras["input_coord"]
        coord = coord.reshape(coord.shape[0], -1, 2)
        rep = make_so2_reps(coord, attn_kwargs["so2"],
                            (attn_kwargs["max_freq_h"], attn_kwargs["max_freq_w"]),
                            attn_kwargs.get("shared_freqs", False))
        reps["so2rep_q"] = reps["so2rep_k"] = rep
    if f.get("t2", 0) > 0:
        coord = extras["input_coord"]
        coord = coord.reshape(coord.shape[0], -1, 2)
        T = make_t2_reps(coord)
        reps["t2rep_q"] = reps["t2rep_k"] = T
        reps["inv_t2rep_q"] = torch.linalg.inv(T)
    need_se3 = f.get("se3", 0) > 0
    need_so3 = f.get("so3", 0) > 0
    if need_se3 or need_so3:
        se3rep, inv, Ds = build_view_reps(extras["input_transforms"],
                                          attn_kwargs.get("so3", 0) if need_so3 else 0, wigner)
        if need_se3:
            reps["se3rep_q"] = reps["se3rep_k"] = se3rep
            reps["inv_se3rep_q"] = inv
        if need_so3:
            reps["so3rep_q"] = reps["so3rep_k"] = _so3_override(attn_kwargs, Ds)
    return reps


def decoder_reps(attn_kwargs: dict, extras: dict, enc_reps: dict, wigner: str = "euler") -> dict:
    """Cross-attention reps: q-side from target_*; k-side kept from the encoder call
    (decoder.py:247-353: only ``*_q`` keys are overwritten, ``se3rep_k`` is rebuilt from
    ``input_transforms`` only when missing, ``so2rep_k`` only under ``recompute_so2``)."""
    f = attn_kwargs["f_dims"]
    reps = dict(enc_reps)
    if f.get("so2", 0) > 0:
        coord = extras["target_coord"]
        coord = coord.reshape(coord.shape[0], -1, 2)
        mf = (attn_kwargs["max_freq_h"], attn_kwargs["max_freq_w"])
        sh = attn_kwargs.get("shared_freqs", False)
        reps["so2rep_q"] = make_so2_reps(coord, attn_kwargs["so2"], mf, sh)
        if attn_kwargs.get("recompute_so2", False):
            ci = extras["input_coord"]
            ci = ci.reshape(ci.shape[0], -1, 2)
            reps["so2rep_k"] = make_so2_reps(ci, attn_kwargs["so2"], mf, sh)
    if f.get("t2", 0) > 0:
        coord = extras["target_coord"]
        coord = coord.reshape(coord.shape[0], -1, 2)
        T = make_t2_reps(coord)
        reps["t2rep_q"] = T
        reps["inv_t2rep_q"] = torch.linalg.inv(T)
    need_se3 = f.get("se3", 0) > 0
    need_so3 = f.get("so3", 0) > 0
    if need_se3 or need_so3:
        se3rep, inv, Ds = build_view_reps(extras["target_transforms"],
                                          attn_kwargs.get("so3", 0) if need_so3 else 0, wigner)
        if need_se3:
            reps["se3rep_q"] = se3rep
            reps["inv_se3rep_q"] = inv
            if "se3rep_k" not in reps:
                reps["se3rep_k"] = torch.linalg.inv(extras["input_transforms"])
        if need_so3:
            reps["so3rep_q"] = _so3_override(attn_kwargs, Ds)
    return reps


# --------------------------------------------------------------------------------------
# the operator (gta.py:92-279 + layers.py:202-224)
# --------------------------------------------------------------------------------------
def slab_bounds(f_dims: dict) -> Dict[str, Tuple[int, int]]:
    """Channel ranges per group, in GROUP_ORDER; groups absent from f_dims are skipped,
    groups present with size 0 occupy an empty range (gta.py:115-122)."""
    out, cur = {}, 0
    for key in GROUP_ORDER:
        if key in f_dims:
            out[key] = (cur, cur + f_dims[key])
            cur += f_dims[key]
    return out


def _per_view(mat: torch.Tensor, x: torch.Tensor, n_views: int, width: int) -> torch.Tensor:
    """Apply per-view [B,N,w,w] matrices to x [B,H,T,C] viewed as [B,H,N,T/N,C/w,w]."""
    B, H, T, C = x.shape
    xv = x.reshape(B, H, n_views, T // n_views, C // width, width)
    y = torch.einsum("bnij,bhntcj->bhntci", mat.to(x.dtype), xv)
    return y.reshape(B, H, T, C)


def _per_token(mat: torch.Tensor, x: torch.Tensor, width: int) -> torch.Tensor:
    """Apply per-token matrices [B,T,(C/w),w,w] or [B,T,w,w] to x [B,H,T,C]."""
    B, H, T, C = x.shape
    xv = x.reshape(B, H, T, C // width, width)
    if mat.dim() == 5:
        y = torch.einsum("btcij,bhtcj->bhtci", mat.to(x.dtype), xv)
    else:
        y = torch.einsum("btij,bhtcj->bhtci", mat.to(x.dtype), xv)
    return y.reshape(B, H, T, C)


def _so3_apply(Ds: List[torch.Tensor], x: torch.Tensor, n_views: int, transpose=False):
    """x channels are groups of [3 | 5 | ...] (one sub-block per degree) (gta.py:174-201)."""
    B, H, T, C = x.shape
    dims = [D.shape[-1] for D in Ds]
    tot = sum(dims)
    xv = x.reshape(B, H, n_views, -1, tot)
    outs, st = [], 0
    for D, d in zip(Ds, dims):
        M = D.detach().to(x.dtype)
        if transpose:
            M = M.transpose(-1, -2)
        outs.append(torch.einsum("bnij,bhnkj->bhnki", M, xv[..., st:st + d]))
        st += d
    return torch.cat(outs, -1).reshape(B, H, T, C)


def _affine3(mat: torch.Tensor, x: torch.Tensor, n_views: int) -> torch.Tensor:
    """euclid mode: 3-vectors acted on affinely, y = M[:3,:3] x + M[:3,3] (gta.py:146-156)."""
    B, H, T, C = x.shape
    xv = x.reshape(B, H, n_views, T // n_views, C // 3, 3)
    M = mat.to(x.dtype)
    y = torch.einsum("bnij,bhntcj->bhntci", M[..., :3, :3], xv) + M[:, None, :, None, None, :3, 3]
    return y.reshape(B, H, T, C)


def transform_qkv(q, k, v, f_dims, reps, trans_coeff=1.0, v_transform=True, euclid=False):
    """rho-transformed (q', k', v') of gta.py:127-242."""
    sl = slab_bounds(f_dims)
    qs, ks, vs = [], [], []
    for key in GROUP_ORDER:
        if key not in sl or f_dims[key] <= 0:
            continue
        a, b = sl[key]
        qg, kg, vg = q[..., a:b], k[..., a:b], v[..., a:b]
        if key == "triv":
            pass
        elif key == "se3":
            msk = scale_mask(trans_coeff, q.device, reps["se3rep_q"].dtype)
            cq, ck = reps["se3rep_q"] * msk, reps["se3rep_k"] * msk
            icq = reps["inv_se3rep_q"] * msk
            Nq, Nk = cq.shape[1], ck.shape[1]
            if euclid:
                qg, kg = _affine3(cq, qg, Nq), _affine3(ck, kg, Nk)
                vg = _affine3(ck, vg, Nk) if v_transform else vg
            else:
                qg = _per_view(icq.transpose(-1, -2), qg, Nq, 4)
                kg = _per_view(ck, kg, Nk, 4)
                vg = _per_view(ck, vg, Nk, 4) if v_transform else vg
        elif key == "so3":
            Dq, Dk = reps["so3rep_q"], reps["so3rep_k"]
            Nq, Nk = Dq[0].shape[1], Dk[0].shape[1]
            qg = _so3_apply(Dq, qg, Nq)
            kg = _so3_apply(Dk, kg, Nk)
            vg = _so3_apply(Dk, vg, Nk) if v_transform else vg
        elif key == "so2":
            qg = _per_token(reps["so2rep_q"], qg, 2)
            kg = _per_token(reps["so2rep_k"], kg, 2)
            vg = _per_token(reps["so2rep_k"], vg, 2) if v_transform else vg
        elif key == "t2":
            qg = _per_token(reps["inv_t2rep_q"].transpose(-1, -2), qg, 3)
            kg = _per_token(reps["t2rep_k"], kg, 3)
            vg = _per_token(reps["t2rep_k"], vg, 3) if v_transform else vg
        qs.append(qg), ks.append(kg), vs.append(vg)
    return torch.cat(qs, -1), torch.cat(ks, -1), torch.cat(vs, -1)


def inverse_transform_out(o, f_dims, reps, trans_coeff=1.0, euclid=False):
    """rho(g_q)^-1 on the attention output (gta.py:246-276)."""
    sl = slab_bounds(f_dims)
    outs = []
    for key in GROUP_ORDER:
        if key not in sl or f_dims[key] <= 0:
            continue
        a, b = sl[key]
        og = o[..., a:b]
        if key == "se3":
            msk = scale_mask(trans_coeff, o.device, reps["inv_se3rep_q"].dtype)
            icq = reps["inv_se3rep_q"] * msk
            og = _affine3(icq, og, icq.shape[1]) if euclid else _per_view(icq, og, icq.shape[1], 4)
        elif key == "so3":
            Dq = reps["so3rep_q"]
            og = _so3_apply(Dq, og, Dq[0].shape[1], transpose=True)
        elif key == "so2":
            og = _per_token(reps["so2rep_q"].transpose(-1, -2), og, 2)
        elif key == "t2":
            og = _per_token(reps["inv_t2rep_q"], og, 3)
        outs.append(og)
    return torch.cat(outs, -1)


def softmax_attention(qt, kt, vt, scale, tau=1.0, euclid=False):
    """softmax(sim * scale / tau) @ v  (layers.py:202-224)."""
    sim = qt @ kt.transpose(-1, -2)
    if euclid:
        sim = sim - 0.5 * qt.pow(2).sum(-1)[..., None] - 0.5 * kt.pow(2).sum(-1)[..., None, :]
    attn = torch.softmax(sim * scale / tau, dim=-1)
    return attn @ vt, attn


def gta_attention(q, k, v, f_dims, reps, trans_coeff=1.0, v_transform=True, euclid=False,
                  scale: Optional[float] = None, tau=1.0):
    """Full operator: (out, attn) as ``multihead_geometric_transform_attention`` +
    ``AttnFn`` return them (gta.py:92-279, layers.py:202-211)."""
    if scale is None:
        scale = q.shape[-1] ** -0.5
    qt, kt, vt = transform_qkv(q, k, v, f_dims, reps, trans_coeff, v_transform, euclid)
    o, attn = softmax_attention(qt, kt, vt, scale, tau, euclid)
    if v_transform:
        o = inverse_transform_out(o, f_dims, reps, trans_coeff, euclid)
    return o, attn


def vecrep_attention(q, k, v, vecrep_q, vecrep_k, vecinvrep_q, scale, tau=1.0):
    """``elementwise_mul`` ablation (gta.py:282-298)."""
    o, attn = softmax_attention(vecrep_q[:, None] * q, vecrep_k[:, None] * k,
                                vecrep_k[:, None] * v, scale, tau)
    return vecinvrep_q[:, None] * o, attn


# --------------------------------------------------------------------------------------
# modules (layers.py:146-169, 172-444, 447-488) -- same parameter names / state-dict keys
# --------------------------------------------------------------------------------------
class OracleAttention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0, kv_dim=None, attn_args=None):
        super().__init__()
        args = attn_args["method"]["args"]
        inner = heads * dim_head
        self.heads, self.scale, self.args = heads, dim_head ** -0.5, args
        self.euclid = args.get("euclid_sim", False)
        if args["f_dims"].get("se3", 0) > 0:
            self.trans_coeff = nn.Parameter(torch.tensor([0.01]))
        else:
            self.trans_coeff = None
        if attn_args.get("softmax") == "adjustable":
            self.attend = nn.Module()
            self.attend.tau = nn.Parameter(torch.tensor([1.0]))
        if kv_dim is None:
            self.to_qkv = nn.Linear(dim, 3 * inner, bias=False)
        else:
            self.to_q = nn.Linear(dim, inner, bias=False)
            self.to_kv = nn.Linear(kv_dim, 2 * inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))

    def forward(self, x, z=None, return_attmap=False, extras=None):
        B, Tq, _ = x.shape
        if z is None:
            q, k, v = self.to_qkv(x).chunk(3, dim=-1)
        else:
            q = self.to_q(x)
            k, v = self.to_kv(z).chunk(2, dim=-1)
        split = lambda t: t.reshape(t.shape[0], t.shape[1], self.heads, -1).transpose(1, 2)
        q, k, v = split(q), split(k), split(v)
        tau = self.attend.tau if hasattr(self, "attend") else 1.0
        tc = self.trans_coeff if self.trans_coeff is not None else 1.0
        out, attn = gta_attention(q, k, v, self.args["f_dims"], extras, tc,
                                  self.args.get("v_transform", True), self.euclid, self.scale, tau)
        out = self.to_out(out.transpose(1, 2).reshape(B, Tq, -1))
        return (out, attn) if return_attmap else out


class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm, self.fn = nn.LayerNorm(dim), fn

    def forward(self, x, **kw):
        return self.fn(self.norm(x), **kw)


class _FeedForward(nn.Module):
    def __init__(self, dim, hidden, dropout=0.0):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(),
                                 nn.Dropout(dropout) if dropout > 0 else nn.Identity(),
                                 nn.Linear(hidden, dim),
                                 nn.Dropout(dropout) if dropout > 0 else nn.Identity())

    def forward(self, x):
        return self.net(x)


class OracleTransformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.0, selfatt=True,
                 kv_dim=None, return_last_attmap=False, attn_args=None):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([
                _PreNorm(dim, OracleAttention(dim, heads, dim_head, dropout, kv_dim, attn_args)),
                _PreNorm(dim, _FeedForward(dim, mlp_dim, dropout))])
            for _ in range(depth)])
        self.return_last_attmap = return_last_attmap

    def forward(self, x, z=None, extras=None):
        attmap = None
        for i, (attn, ff) in enumerate(self.layers):
            if i == len(self.layers) - 1 and self.return_last_attmap:
                out, attmap = attn(x, z=z, return_attmap=True, extras=extras)
                x = out + x
            else:
                x = attn(x, z=z, extras=extras) + x
            x = ff(x) + x
        return (x, attmap) if self.return_last_attmap else x


# --------------------------------------------------------------------------------------
# encoder / decoder wrappers, gta path (encoder.py:37-345, decoder.py:27-384, models_nvs.py:14-91)
# --------------------------------------------------------------------------------------
class _ConvBlock(nn.Module):
    """encoder.py:13-34."""

    def __init__(self, idim, hdim=None, odim=None):
        super().__init__()
        hdim = idim if hdim is None else hdim
        odim = 2 * hdim if odim is None else odim
        self.layers = nn.Sequential(nn.Conv2d(idim, hdim, 3, 1, 1, bias=False), nn.ReLU(),
                                    nn.Conv2d(hdim, odim, 3, 2, 1, bias=False), nn.ReLU())

    def forward(self, x):
        return self.layers(x)


class OracleSRT(nn.Module):
    """CPU restatement of TransformingSRT for the GTA configs (encoder ``emb: False``, decoder ``emb: const``):
    conv stem -> 1x1 projection -> self-attention Transformer; one learned query vector per target ray ->
    cross-attention Transformer -> render MLP.  Parameter names follow the reference's state dict."""

    def __init__(self, cfg):
        super().__init__()
        ek, dk = cfg["encoder_kwargs"], cfg["decoder_kwargs"]
        enc = nn.Module()
        dim, attdim = ek.get("dim", 768), ek.get("attdim", 768)
        blocks = [_ConvBlock(3, dim // 8)]
        cur = dim // 4
        for _ in range(1, ek.get("num_conv_blocks", 3)):
            blocks.append(_ConvBlock(cur))
            cur *= 2
        enc.conv_blocks = nn.Sequential(*blocks)
        enc.per_patch_linear = nn.Conv2d(cur, attdim, 1)
        heads = ek.get("heads", 12)
        enc.transformer = OracleTransformer(attdim, ek.get("num_att_blocks", 5), heads, attdim // heads, attdim * 2,
                                            ek.get("dropout") or 0.0, True, None, False, ek["attn_args"])
        self.encoder = enc
        self.enc_args = ek["attn_args"]["method"]["args"]
        dec = nn.Module()
        ddim, z_dim, dheads = dk.get("dim", 180), dk.get("z_dim", 768), dk.get("heads", 12)
        alloc = nn.Module()
        alloc.initial_emb = nn.Parameter(torch.randn(ddim))
        alloc.transformer = OracleTransformer(ddim, dk.get("num_att_blocks", 2), dheads, dk.get("dim_head") or z_dim // dheads,
                                              dk.get("mlp_dim") or z_dim * 2, dk.get("dropout") or 0.0, False, z_dim,
                                              False, dk["attn_args"])
        dec.allocation_transformer = alloc
        r = dk.get("rmlp_dim", 1536)
        act = {"relu": nn.ReLU, "lrelu": nn.LeakyReLU, "gelu": nn.GELU}[dk.get("act", "lrelu")]
        mlp = [nn.Linear(ddim, r), act()]
        for _ in range(3):
            mlp += [nn.Linear(r, r), act()]
        mlp += [nn.Linear(r, 3), nn.Sigmoid() if dk.get("sigmoid", True) else nn.Identity()]
        dec.render_mlp = nn.Sequential(*mlp)
        self.decoder = dec
        self.dec_args = dk["attn_args"]["method"]["args"]

    def forward(self, input_images, input_camera_pos, input_rays, target_camera_pos, target_rays, extras):
        B, N = input_images.shape[:2]
        reps = encoder_reps(self.enc_args, extras)
        x = self.encoder.per_patch_linear(self.encoder.conv_blocks(input_images.flatten(0, 1)))
        x = x.flatten(2, 3).permute(0, 2, 1)
        x = x.reshape(B, N * x.shape[1], x.shape[2])
        z = self.encoder.transformer(x, None, reps)
        reps = decoder_reps(self.dec_args, extras, reps)
        rays = target_rays.flatten(1, 2) if target_rays.dim() == 4 else target_rays
        q = self.decoder.allocation_transformer.initial_emb[None, None].expand(B, rays.shape[1], -1)
        out = self.decoder.allocation_transformer.transformer(q, z, reps)
        return self.decoder.render_mlp(out)


def mse2psnr(mse: torch.Tensor) -> torch.Tensor:
    """common.py:14-15."""
    return -10.0 * torch.log(mse) / math.log(10.0)


# --------------------------------------------------------------------------------------
# synthetic inputs following the data loaders' output contract
# --------------------------------------------------------------------------------------
def random_extrinsics(B: int, N: int, gen: torch.Generator, dtype=torch.float32) -> torch.Tensor:
    """E_0 = I (canonical view, clevr_tr.py:248-249); E_n = [R | t], R = QR-random, det +1."""
    A = torch.randn(B, N, 3, 3, generator=gen, dtype=torch.float64)
    Q, Rr = torch.linalg.qr(A)
    Q = Q * torch.sign(torch.diagonal(Rr, dim1=-2, dim2=-1))[..., None, :]
    Q[..., :, 0] = Q[..., :, 0] * torch.linalg.det(Q)[..., None]
    E = torch.zeros(B, N, 4, 4, dtype=torch.float64)
    E[..., :3, :3] = Q
    E[..., :3, 3] = torch.randn(B, N, 3, generator=gen, dtype=torch.float64)
    E[..., 3, 3] = 1.0
    E[:, 0] = torch.eye(4, dtype=torch.float64)
    return E.to(dtype)


def patch_coords(H: int, W: int, stride: int) -> torch.Tensor:
    """Patch-centre coordinates [h*w, 2] (common.downsample, common.py:105-110)."""
    g = make_2dcoord(H, W)
    return g[stride // 2::stride, stride // 2::stride].reshape(-1, 2)
