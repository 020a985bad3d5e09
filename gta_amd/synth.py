"""Seeded synthetic inputs of the attention operator: camera poses, token coordinates, q/k/v.

Used by ``bench.py``, ``tools/`` and the GPU tests; there is no dataset in the image.  Poses follow the data loaders'
contract (multishapenet.py / clevr_tr.py:248-249): view 0 is canonical (identity), the others are rigid [R | t] with a
QR-random rotation of determinant +1."""
from __future__ import annotations

import torch


def random_extrinsics(B: int, N: int, gen: torch.Generator, dtype=torch.float32) -> torch.Tensor:
    """[B, N, 4, 4] world-to-camera matrices."""
    A = torch.randn(B, N, 3, 3, generator=gen, dtype=torch.float64)
    Q, R = torch.linalg.qr(A)
    Q = Q * torch.sign(torch.diagonal(R, dim1=-2, dim2=-1))[..., None, :]
    Q[..., :, 0] = Q[..., :, 0] * torch.linalg.det(Q)[..., None]
    E = torch.zeros(B, N, 4, 4, dtype=torch.float64)
    E[..., :3, :3] = Q
    E[..., :3, 3] = torch.randn(B, N, 3, generator=gen, dtype=torch.float64)
    E[..., 3, 3] = 1.0
    E[:, 0] = torch.eye(4, dtype=torch.float64)
    return E.to(dtype)


def attention_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, seed=0, cross=None):
    """CPU fp32 masters: (q [B,H,Nq*Pq,dh], k, v [B,H,Nk*Pk,dh], extras, attn_kwargs, cross).

    ``extras`` holds what the reference's ``pre_compute_reps`` reads (encoder.py:183-265, decoder.py:247-353):
    ``input_transforms`` / ``input_coord`` and, for cross-attention, ``target_transforms`` / ``target_coord``."""
    g = torch.Generator().manual_seed(seed)
    dh = sum(f_dims.values())
    cross = (Nq, Pq) != (Nk, Pk) if cross is None else cross
    ex = {"input_transforms": random_extrinsics(B, Nk, g), "input_coord": torch.rand(B, Nk, Pk, 2, generator=g)}
    if cross:
        ex["target_transforms"] = random_extrinsics(B, Nq, g)
        ex["target_coord"] = torch.rand(B, Nq, Pq, 2, generator=g)
    q = torch.randn(B, H, Nq * Pq, dh, generator=g)
    k = torch.randn(B, H, Nk * Pk, dh, generator=g)
    v = torch.randn(B, H, Nk * Pk, dh, generator=g)
    ak = {"f_dims": dict(f_dims), "so2": so2, "so3": so3, "max_freq_h": 1, "max_freq_w": 1}
    return q, k, v, ex, ak, cross


def as_projection(t: torch.Tensor, dtype, device) -> torch.Tensor:
    """[B,H,T,dh] master -> the layout the module's packed projection leaves: [B,T,H,dh] in memory, viewed [B,H,T,dh]."""
    return t.to(dtype).to(device).permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
