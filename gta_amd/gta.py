"""Functional GTA operator on MI355X -- host-side mirror of ``source/utils/gta.py``.

``multihead_geometric_transform_attention`` keeps the reference's name, argument meaning and
``reps`` dict contract (gta.py:92-279) but runs the whole operator as one fused HIP kernel.
What differs, deliberately:
  * the dense attention matrix is never formed, so the second return value is ``None``
    (the reference only consumes it under ``return_attmap``, layers.py:441-442);
  * ``attn_fn`` is accepted for signature compatibility; its ``scale`` attribute is honoured,
    the softmax itself is the kernel's.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import weakref

import os

import torch

from . import native

GROUP_ORDER = ("triv", "se3", "so3", "so2", "t2")   # gta.py:115


def make_2dcoord(H: int, W: int) -> np.ndarray:
    """[H, W, 2] with entry (i, j) = (i/H, j/W)   (gta.py:9-16)."""
    r = np.arange(H, dtype=np.float32) / H
    c = np.arange(W, dtype=np.float32) / W
    rr, cc = np.meshgrid(r, c, indexing="ij")
    return np.stack([rr, cc], -1)


def make_SO2mats(coord: torch.Tensor, nfreqs: int, max_freqs=(1, 1), shared_freqs: bool = False) -> torch.Tensor:
    """[..., F, 2, 2, 2] rotation blocks like gta.py:47-69, built from the device (cos,sin) table."""
    lead = coord.shape[:-1]
    cs = native.build_so2_table(coord.reshape(1, -1, 2), nfreqs, max_freqs[0], max_freqs[1], shared_freqs)
    c, s = cs[0, ..., 0], cs[0, ..., 1]                       # [T, 2F]
    m = torch.stack([torch.stack([c, -s], -1), torch.stack([s, c], -1)], -2)   # [T, 2F, 2, 2]
    return m.reshape(*lead, nfreqs, 2, 2, 2)


# --------------------------------------------------------------------------------------------
# reps: reference-style dict  ->  packed device tables the kernels read
# --------------------------------------------------------------------------------------------
def _pack_view(inv: Optional[torch.Tensor], rep: Optional[torch.Tensor], Ds) -> torch.Tensor:
    ref = inv if inv is not None else (rep if rep is not None else Ds[0])
    B, N = ref.shape[:2]
    out = torch.zeros(B, N, native.VREP_STRIDE, device=ref.device, dtype=torch.float32)
    if inv is not None:
        out[..., native.VREP_INV:native.VREP_INV + 16] = inv.detach().reshape(B, N, 16)
    if rep is not None:
        out[..., native.VREP_REP:native.VREP_REP + 16] = rep.detach().reshape(B, N, 16)
    if Ds:
        out[..., native.VREP_D1:native.VREP_D1 + 9] = Ds[0].detach().reshape(B, N, 9)
        if len(Ds) > 1:
            out[..., native.VREP_D2:native.VREP_D2 + 25] = Ds[1].detach().reshape(B, N, 25)
    return out


def _pack_so2(rep: torch.Tensor) -> torch.Tensor:
    """[B,T,C,2,2] rotation blocks -> [B,T,C,2] (cos, sin) = (M[0,0], M[1,0])."""
    return torch.stack([rep[..., 0, 0], rep[..., 1, 0]], -1).detach().to(torch.float32).contiguous()


def _src_key(*tensors):
    """Identity of the tensors a packed table was built from: storage address, shape and in-place version counter
    (``id()`` can alias after garbage collection; a tensor replaced by the decoder changes at least the address)."""
    out = []
    for t in tensors:
        if torch.is_tensor(t):
            out.append((t.data_ptr(), tuple(t.shape), t._version))
        elif isinstance(t, (list, tuple)):
            out.append(tuple((u.data_ptr(), tuple(u.shape), u._version) for u in t))
        else:
            out.append(None)
    return tuple(out)


def _flat(sources):
    """the tensors among a table's sources (lists of tensors flattened, None dropped)"""
    out = []
    for t in sources:
        if torch.is_tensor(t):
            out.append(t)
        elif isinstance(t, (list, tuple)):
            out.extend(u for u in t if torch.is_tensor(u))
    return out


def pack_reps(reps: dict, f_dims: dict) -> dict:
    """Return (and cache in ``reps``) the packed tables for a reference-style ``reps`` dict.

    Accepts either the packed keys written by ``gta_amd.reps.pre_compute_reps_*``
    (``gta_vrep_q/k``, ``gta_cs_q/k``, ``gta_coord_q/k``) or the reference's own tensors (``se3rep_q/k``,
    ``inv_se3rep_q``, ``so3rep_q/k`` lists, ``so2rep_q/k``, ``t2rep_q/k``; encoder.py:197,208-215,235-236,259).
    A table packed from reference tensors is tied to them (``*_src``): when the decoder replaces the q side of the
    shared dict (decoder.py:283-311), the table is rebuilt -- for all three kinds.
    """
    need_view = f_dims.get("se3", 0) > 0 or f_dims.get("so3", 0) > 0
    need_so2 = f_dims.get("so2", 0) > 0
    out = {}

    def cached(key, sources, build):
        # tables written by gta_amd.reps carry no *_src: they are authoritative
        if key in reps and (key + "_src") not in reps:
            return reps[key]
        src = _src_key(*sources)
        # (address, shape, version) alone can alias: the caching allocator hands a freed block to the next tensor of the same
        # shape, at version 0.  The table therefore also holds WEAK references to its source tensors -- a dead one, or one that
        # is not the tensor now in the dict, forces a rebuild.
        alive = reps.get(key + "_ref")
        same = alive is not None and len(alive) == len(_flat(sources)) and all(r() is t for r, t in zip(alive, _flat(sources)))
        if key not in reps or reps.get(key + "_src") != src or not same:
            reps[key] = build()
            reps[key + "_src"] = src
            reps[key + "_ref"] = tuple(weakref.ref(t) for t in _flat(sources))
        return reps[key]

    if need_view:
        for side in ("q", "k"):
            inv = reps.get("inv_se3rep_q") if side == "q" else None
            rep, so3 = reps.get(f"se3rep_{side}"), reps.get(f"so3rep_{side}")
            Ds = list(so3 or []) if f_dims.get("so3", 0) > 0 else []
            out[f"vrep_{side}"] = cached(f"gta_vrep_{side}", (rep, inv, so3), lambda: _pack_view(inv, rep, Ds))
    if f_dims.get("t2", 0) > 0:
        # make_T2mats (gta.py:72-89): T = [[1,0,0],[0,1,0],[cx,cy,1]] -> the kernel wants (cx, cy) per token
        for side in ("q", "k"):
            T = reps.get(f"t2rep_{side}")
            out[f"coord_{side}"] = cached(
                f"gta_coord_{side}", (T,),
                lambda: torch.stack([T[..., 2, 0], T[..., 2, 1]], -1).detach().float().contiguous())
    if need_so2:
        for side in ("q", "k"):
            R = reps.get(f"so2rep_{side}")
            out[f"cs_{side}"] = cached(f"gta_cs_{side}", (R,), lambda: _pack_so2(R))
    return out


def _so3_degree(f_dims: dict, packed: dict, reps: dict) -> int:
    if f_dims.get("so3", 0) <= 0:
        return 0
    if "gta_so3_degree" in reps:
        return int(reps["gta_so3_degree"])
    return len(reps["so3rep_q"])


def _views(f_dims, packed, q, k) -> Tuple[int, int]:
    if "vrep_q" in packed:
        return packed["vrep_q"].shape[1], packed["vrep_k"].shape[1]
    return 1, 1


# --------------------------------------------------------------------------------------------
# autograd wrapper around the C ABI
# --------------------------------------------------------------------------------------------
def _as_kernel_layout(t: torch.Tensor) -> torch.Tensor:
    """[B,H,T,dh] view usable by the kernel as is (unit channel stride, 16-B aligned rows)."""
    return t if _kernel_layout_ok(t) else t.contiguous()


def _kernel_layout_ok(t: torch.Tensor) -> bool:
    esz = t.element_size()
    return t.stride(3) == 1 and t.data_ptr() % 16 == 0 and all((s * esz) % 16 == 0 for s in t.stride()[:3])


def _check_tables(q, k, f_dims, Nq, Nk, vrep_q, vrep_k, cs_q, cs_k, coord_q, coord_k, tc, ta, k_side=True):
    """Shape / dtype / device validation of everything that reaches the kernels as a raw pointer."""
    B, _, Tq, _ = q.shape
    Tk = k.shape[2]
    dev = q.device
    if f_dims.get("se3", 0) > 0 or f_dims.get("so3", 0) > 0:
        native.check_table("vrep_q", vrep_q, (B, Nq, native.VREP_STRIDE), dev)
        native.check_table("vrep_k", vrep_k, (B, Nk, native.VREP_STRIDE), dev, allow_none=not k_side)
    if f_dims.get("so2", 0) > 0:
        native.check_table("cs_q", cs_q, (B, Tq, f_dims["so2"] // 2, 2), dev)
        native.check_table("cs_k", cs_k, (B, Tk, f_dims["so2"] // 2, 2), dev, allow_none=not k_side)
    if f_dims.get("t2", 0) > 0:
        native.check_table("coord_q", coord_q, (B, Tq, 2), dev)
        native.check_table("coord_k", coord_k, (B, Tk, 2), dev, allow_none=not k_side)
    native.check_scalar("trans_coeff", tc, dev)
    native.check_scalar("tau", ta, dev)


class _SplitPacked(torch.autograd.Function):
    """[B,T,n,H,dh] packed projection -> n strided [B,H,T,dh] views (layers.py:389,394-395), whose backward hands the
    attention backward's packed gradient buffer straight through (``backward._grad_buffers``) instead of letting autograd
    assemble it from three zero-filled scatters and two adds."""

    @staticmethod
    def forward(ctx, packed):
        ctx.pshape = packed.shape
        return tuple(packed[:, :, i].permute(0, 2, 1, 3) for i in range(packed.shape[2]))

    @staticmethod
    def backward(ctx, *grads):
        from .backward import packed_slices
        base = getattr(grads[0], "_base", None) if grads[0] is not None else None
        if (base is not None and all(g is not None and getattr(g, "_base", None) is base for g in grads)
                and tuple(base.shape) == tuple(ctx.pshape) and base.is_contiguous() and packed_slices(grads)):
            return base
        B, T, n, H, dh = ctx.pshape
        ref = next(g for g in grads if g is not None)
        out = torch.empty(ctx.pshape, device=ref.device, dtype=ref.dtype)
        for i, g in enumerate(grads):
            if g is None:
                out[:, :, i].zero_()
            else:
                out[:, :, i].copy_(g.permute(0, 2, 1, 3))
        return out


def split_packed(packed: torch.Tensor):
    """``packed`` [B,T,n,H,dh] -> n views [B,H,T,dh] (no copy in either direction when the consumer is ``gta_attention``)."""
    return _SplitPacked.apply(packed)


class _GtaAttn(torch.autograd.Function):
    flash_events = None     # (start, end) torch.cuda.Event pair set by bench.py, else None

    @staticmethod
    def forward(ctx, q, k, v, trans_coeff, tau, kv_cache, cfg, vrep_q, vrep_k, cs_q, cs_k):
        f_dims, so3_degree, Nq, Nk, scale, flags = cfg
        dt = q.dtype
        if dt not in (torch.float32, torch.bfloat16):
            raise native.GtaError(f"unsupported dtype {dt}: use float32 or bfloat16")
        k = k.to(dt) if k.dtype != dt else k
        v = v.to(dt) if v.dtype != dt else v
        q, k, v = _as_kernel_layout(q), _as_kernel_layout(k), _as_kernel_layout(v)
        B, H, Tq, dh = q.shape
        # out is allocated [B,Tq,H,dh] so the caller's 'b h n d -> b n (h d)' is a free view
        out = torch.empty(B, Tq, H, dh, device=q.device, dtype=dt).permute(0, 2, 1, 3)
        lse = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
        tc = trans_coeff.detach().to(torch.float32).reshape(-1) if trans_coeff is not None else None
        ta = tau.detach().to(torch.float32).reshape(-1) if tau is not None else None
        desc = native.make_desc(q, k, v, out, f_dims, so3_degree, Nq, Nk, scale, flags)
        _check_tables(q, k, f_dims, Nq, Nk, vrep_q, vrep_k, cs_q, cs_k, None, None, tc, ta,
                      k_side=not (flags & native.FLAG_PRETRANSFORMED))
        ws = None
        if not (flags & (native.FLAG_FUSED_KV | native.FLAG_PRETRANSFORMED)):
            # what the images of a cache depend on: the key side's shape and layout, the dtype (bf16 / fp32 inputs pick different instances) and the
            # arithmetic mode -- the fp32-faithful plan stores FOUR images per tile [K'hi | V'hi | K'lo | V'lo], the default plan two: a cache
            # written under one plan must never be streamed under the other (the byte-size check alone let that through)
            plan_key = (bool(flags & native.FLAG_FP32_PRODUCTS), bool(flags & native.FLAG_V_TRANSFORM), str(dt), B, H, int(k.shape[2]), dh,
                        tuple(sorted((g, int(n)) for g, n in f_dims.items() if n)), int(so3_degree), int(Nk))
            if kv_cache is not None and kv_cache.get("images") is not None:
                # K'/V' tile images of an earlier call against the same keys (chunked decode): skip the pre-pass
                ws = kv_cache["images"]
                if kv_cache.get("plan") != plan_key:
                    raise native.GtaError(f"kv_cache holds images written under another plan {kv_cache.get('plan')} (this call: {plan_key}): "
                                          "use one cache dict per (key set, dtype, precise mode)")
                if ws.numel() < native.attn_fwd_workspace_bytes(desc) or ws.device != q.device:
                    raise native.GtaError("kv_cache holds images of a different key set")
                desc.flags = flags | native.FLAG_KV_READY
            else:
                need = native.attn_fwd_workspace_bytes(desc)
                if kv_cache is not None:
                    # the workspace's tail (query-side operand tiles) grows with the number of QUERY views: a cache serves later query
                    # sets against the same keys, so it is sized by asking the library for the most views a call can have (the size
                    # rule -- which instances carry a tail, how large a tile is -- stays the library's)
                    widest = type(desc).from_buffer_copy(desc)
                    widest.Nq = widest.Tq = native.MAX_VIEWS
                    need = max(need, native.attn_fwd_workspace_bytes(widest))
                ws = torch.empty(need, device=q.device, dtype=torch.uint8)
                if kv_cache is not None:
                    kv_cache["images"], kv_cache["plan"] = ws, plan_key
        if ws is not None and _GtaAttn.flash_events is not None and not (desc.flags & native.FLAG_KV_READY):
            # instrumentation (bench.py): bracket the attention kernel alone with stream events
            desc.flags = flags | native.FLAG_PREP_ONLY
            native.attn_fwd(desc, q, k, v, vrep_q, vrep_k, cs_q, cs_k, tc, ta, out, lse, ws)
            desc.flags = flags | native.FLAG_KV_READY
            _GtaAttn.flash_events[0].record()
            native.attn_fwd(desc, q, k, v, vrep_q, vrep_k, cs_q, cs_k, tc, ta, out, lse, ws)
            _GtaAttn.flash_events[1].record()
        else:
            if _GtaAttn.flash_events is not None:
                _GtaAttn.flash_events[0].record()
            native.attn_fwd(desc, q, k, v, vrep_q, vrep_k, cs_q, cs_k, tc, ta, out, lse, ws)
            if _GtaAttn.flash_events is not None:
                _GtaAttn.flash_events[1].record()
        ctx.cfg = cfg
        # K'/V' tile images of the two-stage plan: reused by the backward (fp32-faithful plan: its hi / lo images by the X3 walks, r06)
        ctx.kv_images = ws
        ctx.save_for_backward(q, k, v, out, lse, tc, ta, vrep_q, vrep_k, cs_q, cs_k)
        ctx.tc_shape = None if trans_coeff is None else trans_coeff.shape
        ctx.tc_dtype = None if trans_coeff is None else trans_coeff.dtype
        ctx.tau_meta = None if tau is None else (tau.shape, tau.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import backward as _bw
        q, k, v, out, lse, tc, ta, vrep_q, vrep_k, cs_q, cs_k = ctx.saved_tensors
        want_dtau = ta is not None and ctx.needs_input_grad[4]
        dq, dk, dv, dtc, dta = _bw.attn_bwd(ctx.cfg, q, k, v, out, dout, lse, tc, ta, vrep_q, vrep_k, cs_q, cs_k,
                                            kv_images=ctx.kv_images, want_dtau=want_dtau)
        if ctx.tc_shape is not None and dtc is not None:
            dtc = dtc.reshape(ctx.tc_shape).to(ctx.tc_dtype)
        else:
            dtc = None
        if dta is not None:
            dta = dta.reshape(ctx.tau_meta[0]).to(ctx.tau_meta[1])
        return dq, dk, dv, dtc, dta, None, None, None, None, None, None


class _GenericAttn(torch.autograd.Function):
    """Ablation layouts the fused kernels refuse (t2 slab, so3 degree 1, unaligned slabs, euclid similarity):
    generic rho-apply kernels (gta_rep_apply) around the plain attention kernel, and for the backward their
    adjoints (gta_rep_apply_bwd) around the fused backward on an identity layout."""

    @staticmethod
    def forward(ctx, q, k, v, trans_coeff, tau, cfg, packed):
        f_dims, so3_degree, scale, v_transform, euclid, precise = cfg
        dt = q.dtype
        if dt not in (torch.float32, torch.bfloat16):
            raise native.GtaError(f"unsupported dtype {dt}")
        k, v = k.to(dt), v.to(dt)
        B, H, Tq, dh = q.shape
        Tk = k.shape[2]
        Nq, Nk = _views(f_dims, packed, q, k)
        flags = (native.FLAG_V_TRANSFORM if v_transform else 0) | (native.FLAG_EUCLID if euclid else 0)
        dhp = (dh + 7) // 8 * 8                                    # the attention kernel works on 8-channel chunks
        # (every kernel below writes all dh channels of every row: only PADDING channels need the zero fill -- 6 us + a launch boundary per buffer)
        alloc = torch.zeros if dhp != dh else torch.empty
        mk = lambda T: alloc(B, T, H, dhp, device=q.device, dtype=dt).permute(0, 2, 1, 3)
        qp, kp, vp, op = mk(Tq), mk(Tk), mk(Tk), mk(Tq)
        out = torch.empty(B, Tq, H, dh, device=q.device, dtype=dt).permute(0, 2, 1, 3)
        desc = native.make_desc(q, k, v, out, f_dims, so3_degree, Nq, Nk, scale, flags)
        tc = trans_coeff.detach().float().reshape(-1) if torch.is_tensor(trans_coeff) else None
        ta = tau.detach().float().reshape(-1) if torch.is_tensor(tau) else None
        pitch = (Tk + 63) // 64 * 64
        kbias = torch.zeros(B, H, pitch, device=q.device, dtype=torch.float32) if euclid else None
        vq, vk = packed.get("vrep_q"), packed.get("vrep_k")
        _check_tables(q, k, f_dims, Nq, Nk, vq, vk, packed.get("cs_q"), packed.get("cs_k"), packed.get("coord_q"),
                      packed.get("coord_k"), tc, ta)
        native.rep_apply(desc, 0, q, vq, packed.get("cs_q"), packed.get("coord_q"), tc, qp[..., :dh])
        native.rep_apply(desc, 1, k, vk, packed.get("cs_k"), packed.get("coord_k"), tc, kp[..., :dh], kbias, scale)
        if v_transform:
            native.rep_apply(desc, 1, v, vk, packed.get("cs_k"), packed.get("coord_k"), tc, vp[..., :dh])
        else:
            vp[..., :dh] = v
        pdesc = native.make_desc(qp, kp, vp, op, {"triv": dhp}, 0, 1, 1, scale, native.FLAG_FP32_PRODUCTS if precise else 0)
        lse = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
        native.attn_fwd_plain(pdesc, qp, kp, vp, kbias, ta, op, lse)
        if v_transform:
            native.rep_apply(desc, 2, op[..., :dh], vq, packed.get("cs_q"), packed.get("coord_q"), tc, out)
        else:
            out.copy_(op[..., :dh])
        ctx.cfg, ctx.packed, ctx.flags = cfg, packed, flags
        ctx.tc_meta = None if not torch.is_tensor(trans_coeff) else (trans_coeff.shape, trans_coeff.dtype)
        ctx.tau_meta = None if not torch.is_tensor(tau) else (tau.shape, tau.dtype)
        ctx.save_for_backward(q, k, v, qp, kp, vp, op, lse, tc, ta)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import backward as _bw
        q, k, v, qp, kp, vp, op, lse, tc, ta = ctx.saved_tensors
        f_dims, so3_degree, scale, v_transform, euclid, _precise = ctx.cfg
        packed = ctx.packed
        want_dtau = ta is not None and ctx.needs_input_grad[4]
        dt = q.dtype
        B, H, Tq, dh = q.shape
        Tk = k.shape[2]
        dhp = qp.shape[-1]
        Nq, Nk = _views(f_dims, packed, q, k)
        desc = native.make_desc(q, k, v, q, f_dims, so3_degree, Nq, Nk, scale, ctx.flags)
        vq, vk = packed.get("vrep_q"), packed.get("vrep_k")
        # buffers the kernels write in full are not zero-filled (padding channels -- d > dh -- are); the per-row d trans_coeff terms of the four adjoints
        # share ONE buffer, summed once in fp64 (four separate reductions were 110 us of the fp32-faithful backward at the CLEVR shape)
        mk = lambda T, d=dhp: (torch.zeros if d != dh else torch.empty)(B, T, H, d, device=q.device, dtype=dt).permute(0, 2, 1, 3)
        need_tc = f_dims.get("se3", 0) > 0 and ctx.tc_meta is not None
        n_rq, n_rk = B * H * Tq, B * H * Tk
        n_rows = (n_rq + n_rk + (n_rk + n_rq if v_transform else 0)) if need_tc else 0
        row_buf = torch.empty(n_rows, device=q.device, dtype=torch.float32) if need_tc else None
        row_at = [0]

        def rows(T):
            a = row_at[0]
            row_at[0] = a + B * H * T
            return row_buf[a:row_at[0]].view(B, H, T)
        dout = dout.to(dt)
        # o = rho2(o~)  ->  do~ = rho2^T do
        dop = mk(Tq)
        r_o = rows(Tq) if (need_tc and v_transform) else None
        if v_transform:
            native.rep_apply_bwd(desc, 2, op[..., :dh], dout, vq, packed.get("cs_q"), packed.get("coord_q"), tc,
                                 dop[..., :dh], r_o)
        else:
            dop[..., :dh] = dout
        # plain attention backward on the padded identity layout.  euclid_sim: the key bias -s|k'|^2/2 of the forward
        # is carried by two spare channels, q' = (.., 1, 1), k' = (.., hi, lo) with hi + lo = -|k'|^2/2 split so that
        # bf16 keeps it to 2^-16 -- the same logits, no bias operand in the backward kernels; the gradient of the
        # bias comes back as dk'[hi channel] and is folded in by the adjoint below.
        dbias = None
        if euclid:
            dhp2 = (dh + 2 + 31) // 32 * 32
            grow = lambda x_, T: torch.cat([x_[..., :dh], torch.zeros(B, H, T, dhp2 - dh, device=q.device, dtype=dt)], -1)
            qp, kp, vp, op, dop = grow(qp, Tq), grow(kp, Tk), grow(vp, Tk), grow(op, Tq), grow(dop, Tq)
            bias = -0.5 * kp[..., :dh].float().pow(2).sum(-1)                         # [B,H,Tk]
            hi = bias.to(dt)
            qp[..., dh:dh + 2] = 1.0
            kp[..., dh] = hi
            kp[..., dh + 1] = (bias - hi.float()).to(dt)
            as_rows = lambda x_: x_.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)   # [B,T,H,d] memory like the kernels' outputs
            qp, kp, vp, op, dop = as_rows(qp), as_rows(kp), as_rows(vp), as_rows(op), as_rows(dop)
            dhp = dhp2
        pcfg = ({"triv": dhp}, 0, 1, 1, scale, 0)
        # (with the bias in the augmented channels, <q', dq'> also carries the bias part of d tau)
        if _precise and dt == torch.float32:
            # fp32-faithful mode: exact-fp32 products and sums (gta_plain32.hip) -- the gradients of the reference's fp32 autograd
            dqp, dkp, dvp = mk(Tq, dhp), mk(Tk, dhp), mk(Tk, dhp)
            pdesc = native.make_desc(qp, kp, vp, op, {"triv": dhp}, 0, 1, 1, scale, 0)
            native.attn_bwd_plain_f32(pdesc, qp, kp, vp, op, dop, lse, ta, dqp, dkp, dvp)
            # dL/dtau = -(1/tau) sum_i <q'_i, dq'_i>  (the kernels of the default mode form it in their epilogue)
            dta = (-(qp.double() * dqp.double()).sum() / ta.double()).float().reshape(1) if (want_dtau and ta is not None) else None
        else:
            dqp, dkp, dvp, _, dta = _bw.attn_bwd(pcfg, qp, kp, vp, op, dop, lse, None, ta, None, None, None, None,
                                                 want_dtau=want_dtau)
        if dta is not None:
            dta = dta.reshape(ctx.tau_meta[0]).to(ctx.tau_meta[1])
        if euclid:
            pitch = (Tk + 63) // 64 * 64
            dbias = torch.zeros(B, H, pitch, device=q.device, dtype=torch.float32)
            dbias[..., :Tk] = dkp[..., dh].float()
        dq, dk, dv = mk(Tq, dh), mk(Tk, dh), mk(Tk, dh)
        r_q, r_k = (rows(Tq), rows(Tk)) if need_tc else (None, None)
        r_v = rows(Tk) if (need_tc and v_transform) else None
        native.rep_apply_bwd(desc, 0, q, dqp[..., :dh], vq, packed.get("cs_q"), packed.get("coord_q"), tc, dq, r_q)
        native.rep_apply_bwd(desc, 1, k, dkp[..., :dh], vk, packed.get("cs_k"), packed.get("coord_k"), tc, dk, r_k,
                             dkey_bias=dbias, bias_scale=1.0)
        if v_transform:
            native.rep_apply_bwd(desc, 1, v, dvp[..., :dh], vk, packed.get("cs_k"), packed.get("coord_k"), tc, dv, r_v)
        else:
            dv.copy_(dvp[..., :dh])
            r_v = None
        dtc = None
        if need_tc:
            assert row_at[0] == n_rows
            dtc = row_buf.sum(dtype=torch.float64).to(ctx.tc_meta[1]).reshape(ctx.tc_meta[0])
        return dq, dk, dv, dtc, dta, None, None


def _generic_forward(q, k, v, f_dims, packed, so3_degree, trans_coeff, tau, scale, v_transform, euclid, precise=False):
    """Generic path entry (forward + backward through _GenericAttn)."""
    tc = trans_coeff if torch.is_tensor(trans_coeff) else None
    ta = tau if torch.is_tensor(tau) else None
    return _GenericAttn.apply(q, k, v, tc, ta, (f_dims, so3_degree, scale, v_transform, euclid, precise), packed)


def gta_attention(q, k, v, f_dims: Dict[str, int], packed: dict, *, so3_degree: int = 0,
                  trans_coeff=None, tau=None, scale: Optional[float] = None, v_transform: bool = True,
                  euclid: bool = False, pretransformed: bool = False, use_dma: bool = True,
                  kv_mode: str = "auto", kv_cache: Optional[dict] = None, precise: Optional[bool] = None) -> torch.Tensor:
    """Fused GTA attention on packed reps.  q [B,H,Tq,dh], k/v [B,H,Tk,dh] -> out [B,H,Tq,dh].

    kv_mode: 'prepass' = K/V rep pre-pass + lean attention kernel (two launches; at dh = 96 in the MSN layout the attention kernel is
                         the 64-rows-per-wave one of gta_fwd64.hip, 'prepass_rows32' keeps gta_fwd2.hip's);
             'fused'   = one kernel, rho_k applied inside the attention loop;
             'auto'    = 'prepass' when several query tiles share each key tile, else 'fused'.
    precise: float32 inputs only.  False / None (default): operands are rounded to bf16 once, after
             rho, and the two contractions run on the bf16 MFMA (fp32 accumulation) -- the reference's bf16-autocast
             accuracy.  True: operands are kept as bf16 hi+lo pairs and every product is three MFMAs -- fp32-class results
             (max |error| ~1e-5 of max |out|) at 3x the matrix work; two-stage plan at dh <= 64 (r05), single-kernel plan otherwise.  When a gradient is wanted the call runs
             rho in fp32 (gta_rep_apply), the split-bf16 plain forward and an EXACT-fp32 backward (gta_plain32.hip: f32 matrix
             instructions, 1/16 of the bf16 rate) -- the arithmetic of the reference's ``mixed_prec: False`` training.
    kv_cache: a dict owned by the caller (inference only).  The first call stores the K'/V' tile images of the
             pre-pass in it; later calls with the same keys, reps and trans_coeff (e.g. the next query chunk of a
             full-image decode, trainer.py:137-181) stream them again without re-running the pre-pass."""
    if scale is None:
        scale = q.shape[-1] ** -0.5
    explicit_fused = kv_mode == "fused"        # (asked for by name: with precise=True and gradients, the exact-fp32 route of r04 -- see below)
    flags = 0
    if v_transform:
        flags |= native.FLAG_V_TRANSFORM
    if euclid:
        flags |= native.FLAG_EUCLID
    if pretransformed:
        flags |= native.FLAG_PRETRANSFORMED
    if not use_dma:
        flags |= native.FLAG_NO_DMA
    if kv_mode == "prepass_pg":        # tuning knob: persistent grid instead of one workgroup per query tile
        flags |= native.FLAG_PERSIST
        kv_mode = "prepass"
    if kv_mode == "prepass_rows32":    # tuning knob: keep the 32-rows-per-wave attention kernel where the 64-rows one would run
        flags |= native.FLAG_ROWS32
        kv_mode = "prepass"
    if kv_mode == "prepass_fwd2":      # tuning knob (r06): the generic 32-row kernel gta_fwd2_kernel where the dh = 64 bf16 instance gta_fwdc_kernel would run
        flags |= native.FLAG_ROWS32 | native.FLAG_FWD2_GENERIC
        kv_mode = "prepass"
    if kv_mode == "prepass_item_cxx":  # tuning knob: the 64-rows kernel with its compiler-scheduled item prologue / epilogue (not the item stream)
        flags |= native.FLAG_ITEM_CXX
        kv_mode = "prepass"
    if kv_mode in ("prepass_bwd_keys32", "prepass_bwd_keys64", "prepass_bwd_split", "prepass_bwd_keys64_split"):
        # tuning knobs of the backward: the 32-keys-per-wave dK/dV kernel / the generated 64-keys streams whatever the size / the dQ and dK/dV kernels
        # (whichever pair the size selects; keys64_split: the generated pair) as two launches instead of the joint one
        if "keys32" in kv_mode:
            flags |= native.FLAG_BWD_KEYS32
        if "keys64" in kv_mode:
            flags |= native.FLAG_BWD_KEYS64
        if kv_mode.endswith("split"):
            flags |= native.FLAG_BWD_SPLIT
        kv_mode = "prepass"
    if kv_mode not in ("auto", "prepass", "fused"):
        raise ValueError(f"kv_mode {kv_mode!r}")
    precise = bool(precise)
    if precise:
        if q.dtype != torch.float32:
            raise native.GtaError("precise=True is for float32 inputs (bf16 inputs ask for bf16 arithmetic)")
        flags |= native.FLAG_FP32_PRODUCTS
        # r05: at dh <= 64 (CLEVR-TR, the reference's fp32 config) the mode has a two-stage plan of its own -- the pre-pass writes hi and lo
        # images, the 32-row kernel runs three MFMAs per product; other head sizes keep the single-kernel plan (asked of the library below)
    if kv_cache is not None:
        if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in (q, k, v, trans_coeff)):
            raise native.GtaError("kv_cache is an inference feature: call under torch.no_grad()")
        kv_mode = "prepass"
    if kv_mode == "auto":
        kv_mode = "prepass" if q.shape[2] > 256 else "fused"
    Nq, Nk = _views(f_dims, packed, q, k)
    if precise and kv_mode == "prepass" and q.is_cuda and not pretransformed:
        # (probed with the tensors the kernel will see: _as_kernel_layout fixes unaligned strides that the raw views would be refused for)
        def _shape_of(t):          # same shape in the layout the call will use (no data: only sizes, dtype and strides matter to the question)
            return t if (t.dtype == q.dtype and _kernel_layout_ok(t)) else q.new_empty(t.shape)
        qp_, kp_, vp_ = _shape_of(q), _shape_of(k), _shape_of(v)
        probe = native.make_desc(qp_, kp_, vp_, qp_, f_dims, so3_degree, Nq, Nk, scale, flags)
        if native.attn_fwd_workspace_bytes(probe) == 0:       # no split-bf16 two-stage instance at this head size
            if kv_cache is not None:
                raise native.GtaError("precise=True at this head size runs the single-kernel plan: no kv_cache")
            kv_mode = "fused"
    if kv_mode == "fused" or not use_dma:
        flags |= native.FLAG_FUSED_KV
    if isinstance(trans_coeff, (int, float)):
        trans_coeff = torch.tensor([float(trans_coeff)], device=q.device, dtype=torch.float32)
    if isinstance(tau, (int, float)):
        tau = None if float(tau) == 1.0 else torch.tensor([float(tau)], device=q.device, dtype=torch.float32)
    needs_grad = torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in (q, k, v, trans_coeff, tau))
    if precise and needs_grad and not pretransformed:
        # fp32-faithful TRAINING.  r06: where the fused kernels have split-bf16 instances on BOTH sides (fp32 inputs at dh <= 64: the two-stage forward
        # of r05 and the X3 walks of gta_bwd.hip) the call stays on the fused path -- hi / lo images, three MFMAs per product, rho and its adjoint in
        # fp32 inside the pre-passes and epilogues; the forward's images serve the backward.  Elsewhere: rho in fp32 (gta_rep_apply), split-bf16
        # plain forward, EXACT-fp32 backward (gta_plain32.hip) and the adjoint rho kernels -- the generic path serves every layout that way.
        fused_x3 = False
        if q.is_cuda and q.shape[-1] <= 64 and not euclid and not explicit_fused:
            qp_, kp_, vp_ = (t if (t.dtype == q.dtype and _kernel_layout_ok(t)) else q.new_empty(t.shape) for t in (q, k, v))
            probe = native.make_desc(qp_, kp_, vp_, qp_, f_dims, so3_degree, Nq, Nk, scale, flags & ~native.FLAG_FUSED_KV)
            fused_x3 = native.attn_fwd_supported(probe) == 0 and native.attn_fwd_workspace_bytes(probe) > 0
        if not fused_x3:
            return _generic_forward(q, k, v, f_dims, packed, so3_degree, trans_coeff, tau, scale, v_transform, euclid, precise=True)
        flags &= ~native.FLAG_FUSED_KV                       # (the two-stage plan whatever the number of query rows: its images are the backward's)
    if q.is_cuda and not pretransformed:
        probe = native.make_desc(q, k, v, q, f_dims, so3_degree, Nq, Nk, scale, flags)
        if native.attn_fwd_supported(probe) == -3:       # GTA_E_UNSUPPORTED: valid request, no fused kernel
            return _generic_forward(q, k, v, f_dims, packed, so3_degree, trans_coeff, tau, scale, v_transform, euclid,
                                    precise=bool(precise))
    cfg = ({k_: int(v_) for k_, v_ in f_dims.items()}, int(so3_degree), Nq, Nk, float(scale), flags)
    return _GtaAttn.apply(q, k, v, trans_coeff, tau, kv_cache, cfg, packed.get("vrep_q"), packed.get("vrep_k"),
                          packed.get("cs_q"), packed.get("cs_k"))


def _closure_has_tau(attn_fn) -> bool:
    """True when ``attn_fn`` is one of the reference's ``AttnFn`` / ``EuclidAttnFn`` instances (layers.py:202-224) --
    modules whose ``forward`` closes over the constructor's local ``tau`` -- and that ``tau`` is a tensor, i.e. the
    parameter of ``TemperatureAdjsutableSoftmax`` (layers.py:135-143,195-197) rather than the constant 1.0."""
    for fn in (getattr(attn_fn, "forward", None), attn_fn):
        fn = getattr(fn, "__func__", fn)
        code, cells = getattr(fn, "__code__", None), getattr(fn, "__closure__", None)
        if code is None or not cells:
            continue
        for name, cell in zip(code.co_freevars, cells):
            try:
                val = cell.cell_contents
            except ValueError:
                continue
            if name == "tau" and torch.is_tensor(val):
                return True
    return False


def multihead_geometric_transform_attention(q, k, v, attn_fn=None, f_dims=None, reps=None,
                                            trans_coeff=1.0, v_transform=True, euclid=False, **kwargs):
    """Drop-in for gta.py:92-279.  Returns ``(out_t, None)``.

    Args mirror the reference: q [B,H,Nq*Tq,C]; k, v [B,H,Nk*Tk,C]; ``f_dims`` the slab sizes;
    ``reps`` the dict filled by ``pre_compute_reps`` (reference-style tensors or this package's
    packed tables); ``trans_coeff`` a scalar or the layer's 1-element parameter.

    Softmax temperature (``softmax: adjustable``, layers.py:135-143,195-200): the reference's ``AttnFn`` /
    ``EuclidAttnFn`` capture tau in a closure and expose only ``.scale``, so a reference-style ``attn_fn`` cannot
    hand it over -- pass ``tau=self.attend.tau`` (level 2).  An ``attn_fn`` that is one of the
    reference's closures over a module with a ``tau`` parameter and no ``tau=`` keyword raises instead of silently
    running with tau = 1.

    Arithmetic: products run on the bf16 MFMA with fp32 accumulation for fp32 inputs too (rho and the softmax are
    fp32); see DESIGN.md section 7 for the measured gap to the fp32 reference.
    """
    if f_dims is None or reps is None:
        raise TypeError("f_dims and reps are required")
    f_dims = {k_: v_ for k_, v_ in f_dims.items() if k_ in GROUP_ORDER}
    packed = pack_reps(reps, f_dims)
    scale = getattr(attn_fn, "scale", None)
    tau = kwargs.get("tau", getattr(attn_fn, "tau", None))
    if "tau" not in kwargs and not hasattr(attn_fn, "tau") and _closure_has_tau(attn_fn):
        raise native.GtaError("attn_fn closes over a softmax temperature (softmax: adjustable, layers.py:195-200) that "
                              "this drop-in cannot see: pass tau=self.attend.tau")
    if f_dims.get("se3", 0) <= 0:
        trans_coeff = None
    out = gta_attention(q, k, v, f_dims, packed, so3_degree=_so3_degree(f_dims, packed, reps),
                        trans_coeff=trans_coeff, tau=tau, scale=scale, v_transform=v_transform, euclid=euclid,
                        use_dma=kwargs.get("use_dma", True), kv_mode=kwargs.get("kv_mode", "auto"),
                        precise=kwargs.get("precise"))
    return out, None


def multihead_vecrep_attention(q, k, v, attn_fn=None, extras=None, **kwargs):
    """Drop-in for gta.py:282-298 (the ``elementwise_mul`` ablation): elementwise per-token vectors
    around plain softmax attention.  The products are torch elementwise ops, the attention is the
    fused kernel on an identity layout; everything is differentiable.  Returns ``(out, None)``."""
    scale = getattr(attn_fn, "scale", None)
    dh = q.shape[-1]
    qq = extras["vecrep_q"][:, None].to(q.dtype) * q
    kk = extras["vecrep_k"][:, None].to(q.dtype) * k
    vv = extras["vecrep_k"][:, None].to(q.dtype) * v
    out = gta_attention(qq, kk, vv, {"triv": dh}, {}, scale=scale, tau=kwargs.get("tau"), v_transform=False)
    return extras["vecinvrep_q"][:, None].to(q.dtype) * out, None


def attention_map(q, k, f_dims, packed, *, so3_degree=0, trans_coeff=None, tau=None, scale=None, euclid=False):
    """Dense softmax matrix [B,H,Tq,Tk] (what the reference returns as ``attn``, layers.py:207-211,441-442).
    Materialising it is the point of the request, so after the HIP rho-apply kernels the matrix itself is
    one batched GEMM + softmax in PyTorch-ROCm.  No gradient."""
    with torch.no_grad():
        if scale is None:
            scale = q.shape[-1] ** -0.5
        dt = q.dtype
        B, H, Tq, dh = q.shape
        Tk = k.shape[2]
        Nq, Nk = _views(f_dims, packed, q, k)
        flags = native.FLAG_V_TRANSFORM | (native.FLAG_EUCLID if euclid else 0)
        qp = torch.empty(B, H, Tq, dh, device=q.device, dtype=dt)
        kp = torch.empty(B, H, Tk, dh, device=q.device, dtype=dt)
        desc = native.make_desc(q, k.to(dt), k.to(dt), qp, f_dims, so3_degree, Nq, Nk, scale, flags)
        tc = trans_coeff.detach().float().reshape(-1) if torch.is_tensor(trans_coeff) else (
            torch.tensor([float(trans_coeff)], device=q.device) if trans_coeff is not None else None)
        kb = torch.zeros(B, H, (Tk + 63) // 64 * 64, device=q.device) if euclid else None
        native.rep_apply(desc, 0, q, packed.get("vrep_q"), packed.get("cs_q"), packed.get("coord_q"), tc, qp)
        native.rep_apply(desc, 1, k.to(dt), packed.get("vrep_k"), packed.get("cs_k"), packed.get("coord_k"), tc, kp, kb, scale)
        sim = torch.matmul(qp.float(), kp.float().transpose(-1, -2)) * scale
        if euclid:
            sim = sim + kb[..., None, :Tk]
        t = float(tau.item()) if torch.is_tensor(tau) else (tau or 1.0)
        return torch.softmax(sim / t, dim=-1)
