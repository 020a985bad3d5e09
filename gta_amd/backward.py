"""Backward of the fused GTA attention (binds ``gta_attn_bwd`` of the C ABI).

Replaces PyTorch autograd over gta.py:92-279 + layers.py:202-211: three HIP kernels (q-side
pre-pass, dQ, dK/dV) plus a deterministic reduction for ``d trans_coeff``.
"""
import torch

from . import native


def _rows_ok(t: torch.Tensor) -> bool:
    esz = t.element_size()
    return t.stride(3) == 1 and t.data_ptr() % 16 == 0 and all((s * esz) % 16 == 0 for s in t.stride()[:3])


def packed_slices(ts) -> bool:
    """True when the [B,H,T,dh] tensors ``ts`` are slices 0..n-1 of ONE [B,T,n,H,dh] buffer (the packed projections
    of layers.py:389,394: ``to_qkv(x).view(B,T,3,H,dh)[:, :, i].permute(0,2,1,3)``)."""
    t0 = ts[0]
    n = len(ts)
    B, H, T, dh = t0.shape
    want = (T * n * H * dh, dh, n * H * dh, 1)
    esz = t0.element_size()
    return all(t.shape == t0.shape and t.dtype == t0.dtype and tuple(t.stride()) == want
               and t.data_ptr() == t0.data_ptr() + i * H * dh * esz for i, t in enumerate(ts))


def _grad_buffers(q, k, v):
    """dq, dk, dv in the memory layout of the forward's q, k, v: slices of one packed buffer when those were (so the
    gradient of the packed projection is that buffer, no gather), else separate [B,T,H,dh] buffers viewed [B,H,T,dh]."""
    B, H, Tq, dh = q.shape
    Tk = k.shape[2]
    new = lambda T, n: torch.empty(B, T, n, H, dh, device=q.device, dtype=q.dtype)
    if Tq == Tk and packed_slices((q, k, v)):
        g = new(Tq, 3)
        return tuple(g[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    dq = new(Tq, 1)[:, :, 0].permute(0, 2, 1, 3)
    if packed_slices((k, v)):
        g = new(Tk, 2)
        return (dq,) + tuple(g[:, :, i].permute(0, 2, 1, 3) for i in range(2))
    return dq, new(Tk, 1)[:, :, 0].permute(0, 2, 1, 3), new(Tk, 1)[:, :, 0].permute(0, 2, 1, 3)


def attn_bwd(cfg, q, k, v, out, dout, lse, tc, ta, vrep_q, vrep_k, cs_q, cs_k, kv_images=None, want_dtau=False):
    """Returns (dq, dk, dv, dtrans_coeff or None, dtau or None); dtau ([1] fp32) only with ``want_dtau``."""
    f_dims, so3_degree, Nq, Nk, scale, flags = cfg
    flags = flags & ~(native.FLAG_FUSED_KV | native.FLAG_KV_READY | native.FLAG_PREP_ONLY | native.FLAG_PERSIST)      # (GTA_FLAG_FP32_PRODUCTS stays: the X3 walks)
    dt = q.dtype
    dout = dout.to(dt)
    if not _rows_ok(dout):
        dout = dout.contiguous()
    B, H, Tq, dh = q.shape
    Tk = k.shape[2]
    # gradients in the projections' memory order (packed like the forward's q, k, v where those were packed)
    dq, dk, dv = _grad_buffers(q, k, v)
    # (both scalars are WRITTEN by the library -- the fixed-order reduction stores its sum -- so no zero-fill launch)
    dtc = torch.empty(1, device=q.device, dtype=torch.float32) if f_dims.get("se3", 0) > 0 else None
    desc = native.make_desc(q, k, v, out, f_dims, so3_degree, Nq, Nk, scale, flags)
    ws = torch.empty(native.attn_bwd_workspace_bytes(desc), device=q.device, dtype=torch.uint8)
    dta = torch.empty(1, device=q.device, dtype=torch.float32) if (want_dtau and ta is not None) else None
    native.attn_bwd(desc, q, k, v, out, dout, lse, vrep_q, vrep_k, cs_q, cs_k, tc, ta, kv_images, dq, dk, dv, dtc, ws, dta)
    return dq, dk, dv, dtc, dta
