"""Backward of the fused GTA attention (binds ``gta_attn_bwd`` of the C ABI).

Replaces PyTorch autograd over gta.py:92-279 + layers.py:202-211: three HIP kernels (q-side
pre-pass, dQ, dK/dV) plus a deterministic reduction for ``d trans_coeff``.
"""
import torch

from . import native


def _rows_ok(t: torch.Tensor) -> bool:
    esz = t.element_size()
    return t.stride(3) == 1 and t.data_ptr() % 16 == 0 and all((s * esz) % 16 == 0 for s in t.stride()[:3])


def attn_bwd(cfg, q, k, v, out, dout, lse, tc, ta, vrep_q, vrep_k, cs_q, cs_k, kv_images=None, want_dtau=False):
    """Returns (dq, dk, dv, dtrans_coeff or None, dtau or None); dtau ([1] fp32) only with ``want_dtau``."""
    f_dims, so3_degree, Nq, Nk, scale, flags = cfg
    flags = flags & ~(native.FLAG_FUSED_KV | native.FLAG_KV_READY | native.FLAG_PREP_ONLY | native.FLAG_PERSIST | native.FLAG_FP32_PRODUCTS)
    dt = q.dtype
    dout = dout.to(dt)
    if not _rows_ok(dout):
        dout = dout.contiguous()
    B, H, Tq, dh = q.shape
    Tk = k.shape[2]
    # gradients in the projection's memory order [B,T,H,dh] (viewed [B,H,T,dh]) like the forward output
    dq = torch.empty(B, Tq, H, dh, device=q.device, dtype=dt).permute(0, 2, 1, 3)
    dk = torch.empty(B, Tk, H, dh, device=q.device, dtype=dt).permute(0, 2, 1, 3)
    dv = torch.empty(B, Tk, H, dh, device=q.device, dtype=dt).permute(0, 2, 1, 3)
    dtc = torch.zeros(1, device=q.device, dtype=torch.float32) if f_dims.get("se3", 0) > 0 else None
    desc = native.make_desc(q, k, v, out, f_dims, so3_degree, Nq, Nk, scale, flags)
    ws = torch.empty(native.attn_bwd_workspace_bytes(desc), device=q.device, dtype=torch.uint8)
    dta = torch.zeros(1, device=q.device, dtype=torch.float32) if (want_dtau and ta is not None) else None
    native.attn_bwd(desc, q, k, v, out, dout, lse, vrep_q, vrep_k, cs_q, cs_k, tc, ta, kv_images, dq, dk, dv, dtc, ws, dta)
    return dq, dk, dv, dtc, dta
