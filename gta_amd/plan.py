"""Pre-planned forward calls of the GTA attention (inference / serving path).

``gta_attention`` allocates its output, LSE and the K'/V' workspace, builds a descriptor and goes through
``torch.autograd.Function`` on every call.  For a fixed shape called many times (a serving loop, ``bench.py``) a
``ForwardPlan`` does that once: caller-visible buffers are allocated at construction and every call is one
``gta_attn_fwd`` through the C ABI (pre-pass + attention kernel), nothing else on the host.  No autograd.

``RepPlan`` is the same for the rep builders (``gta_build_reps``): poses + patch coordinates -> per-view records and
the per-token (cos,sin) table (what ``pre_compute_reps`` does per forward, source/encoder.py:183-265)."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import native


class RepPlan:
    """Encoder-side reps (q side == k side) for B scenes of N views x P tokens."""

    def __init__(self, B: int, N: int, P: int, so3_degree: int, nfreqs: int, max_freq_h: float = 1.0,
                 max_freq_w: float = 1.0, shared_freqs: bool = False, device="cuda"):
        self.args = (B * N, int(so3_degree), B * N * P, int(nfreqs), float(max_freq_h), float(max_freq_w),
                     int(bool(shared_freqs)))
        self.vrep = torch.empty(B, N, native.VREP_STRIDE, device=device, dtype=torch.float32)
        self.cs = torch.empty(B, N * P, 2 * nfreqs, 2, device=device, dtype=torch.float32)
        self._lib = native.lib()

    def __call__(self, transforms: torch.Tensor, coord: torch.Tensor):
        """transforms [B,N,4,4] fp32 contiguous, coord [B,N,P,2] (or [B,N*P,2]) fp32 contiguous, on the device."""
        if not (transforms.is_cuda and coord.is_cuda and transforms.is_contiguous() and coord.is_contiguous()
                and transforms.dtype == torch.float32 and coord.dtype == torch.float32):
            raise native.GtaError("RepPlan wants contiguous fp32 device tensors")
        nv, L, nt, F, mh, mw, sh = self.args
        if transforms.numel() != nv * 16 or coord.numel() != nt * 2:
            raise native.GtaError("RepPlan: shapes differ from the plan's")
        native.check(self._lib.gta_build_reps(ctypes.c_void_p(transforms.data_ptr()), nv, L,
                                              ctypes.c_void_p(self.vrep.data_ptr()), ctypes.c_void_p(coord.data_ptr()),
                                              nt, F, mh, mw, sh, ctypes.c_void_p(self.cs.data_ptr()),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gta_build_reps")
        return self.vrep, self.cs


class ForwardPlan:
    """One fused-eligible attention shape, two-stage plan (K/V pre-pass + attention kernel).

    q, k, v: example tensors [B,H,T,dh] (any strides with unit channel stride) -- later calls must use tensors of the
    same shape, dtype and strides.  ``out`` ([B,H,Tq,dh] view of [B,Tq,H,dh] memory), ``lse`` and the workspace belong
    to the plan and are overwritten by every call."""

    def __init__(self, q, k, v, f_dims: dict, *, so3_degree: int = 0, Nq: int = 1, Nk: int = 1,
                 scale: Optional[float] = None, v_transform: bool = True, flags: int = 0):
        native._require_cuda(q, k, v)
        B, H, Tq, dh = q.shape
        self.out = torch.empty(B, Tq, H, dh, device=q.device, dtype=q.dtype).permute(0, 2, 1, 3)
        self.lse = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
        fl = flags | (native.FLAG_V_TRANSFORM if v_transform else 0)
        self.desc = native.make_desc(q, k, v, self.out, f_dims, so3_degree, Nq, Nk,
                                     float(scale if scale is not None else dh ** -0.5), fl)
        rc = native.attn_fwd_supported(self.desc)
        if rc:
            native.check(rc, "gta_attn_fwd_supported")
        self.ws = torch.empty(native.attn_fwd_workspace_bytes(self.desc), device=q.device, dtype=torch.uint8)
        self._sig = (tuple(q.shape), tuple(q.stride()), tuple(k.shape), tuple(k.stride()), tuple(v.stride()), q.dtype)
        self._lib = native.lib()
        self._need_view = f_dims.get("se3", 0) > 0 or f_dims.get("so3", 0) > 0
        self._need_cs = f_dims.get("so2", 0) > 0
        self._shapes = ((B, Nq, native.VREP_STRIDE), (B, Nk, native.VREP_STRIDE), (B, Tq, f_dims.get("so2", 0) // 2, 2),
                        (B, k.shape[2], f_dims.get("so2", 0) // 2, 2))

    def __call__(self, q, k, v, vrep_q=None, vrep_k=None, cs_q=None, cs_k=None, trans_coeff=None, tau=None,
                 flags_extra: int = 0):
        if (tuple(q.shape), tuple(q.stride()), tuple(k.shape), tuple(k.stride()), tuple(v.stride()), q.dtype) != self._sig:
            raise native.GtaError("ForwardPlan: q/k/v differ from the planned shape / strides / dtype")
        if self._need_view:
            native.check_table("vrep_q", vrep_q, self._shapes[0], q.device)
            native.check_table("vrep_k", vrep_k, self._shapes[1], q.device)
        if self._need_cs:
            native.check_table("cs_q", cs_q, self._shapes[2], q.device)
            native.check_table("cs_k", cs_k, self._shapes[3], q.device)
        # everything else that reaches the kernels as a raw pointer: the scalars, the tensors' device (the launch goes to the
        # CURRENT device's stream)
        native._require_cuda(q, k, v)
        native.check_scalar("trans_coeff", trans_coeff, q.device)
        native.check_scalar("tau", tau, q.device)
        if q.device != self.out.device or torch.cuda.current_device() != q.device.index:
            raise native.GtaError("ForwardPlan: call under the device the plan was built on (torch.cuda.set_device)")
        p = native._ptr
        d = self.desc
        base = d.flags
        if flags_extra:
            d.flags = base | flags_extra
        try:
            native.check(self._lib.gta_attn_fwd(ctypes.byref(d), p(q), p(k), p(v), p(vrep_q), p(vrep_k), p(cs_q), p(cs_k),
                                                p(trans_coeff), p(tau), p(self.out), p(self.lse), p(self.ws),
                                                self.ws.numel(), native._stream()), "gta_attn_fwd")
        finally:
            d.flags = base
        return self.out
