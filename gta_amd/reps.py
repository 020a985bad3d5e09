"""Rep builders on device -- mirrors ``pre_compute_reps`` of the reference's encoder and decoder
(source/encoder.py:183-265, source/decoder.py:247-353).

Same contract as the reference: ``extras`` is the caller-owned dict shared by encoder, decoder
and every layer; it is mutated in place.  Instead of the reference's dense tensors
(``se3rep_*`` [B,N,4,4], ``so3rep_*`` lists, ``so2rep_*`` [B,T,2F,2,2]) the packed tables the
kernels read are stored:

    extras['gta_vrep_q'], extras['gta_vrep_k']   [B, N, 72]      per-view E, inv(E), D^1, D^2
    extras['gta_cs_q'],   extras['gta_cs_k']     [B, T, 2F, 2]   per-token (cos, sin)
    extras['gta_coord_q'], extras['gta_coord_k'] [B, T, 2]       token coordinates of the t2 slab (make_T2mats, gta.py:72-89)
    extras['gta_so3_degree']

The encoder call sets q-side == k-side (self-attention); the decoder call overwrites only the
q-side from ``target_transforms`` / ``target_coord`` and keeps the encoder's k-side
(decoder.py:309-311,346), rebuilding ``*_k`` only when missing or under ``recompute_so2``
(decoder.py:263-272).  The dead ``ray_to_se3`` branch (undefined ``ray2rotation``) and the
``flattened_*`` tensors of the ``elementwise_mul`` ablation are not reproduced.
"""
from __future__ import annotations

from . import native


def _so2(attn_kwargs, coord):
    c = coord.reshape(coord.shape[0], -1, 2)
    return native.build_so2_table(c, attn_kwargs["so2"], attn_kwargs["max_freq_h"], attn_kwargs["max_freq_w"],
                                  attn_kwargs.get("shared_freqs", False))


def _check(attn_kwargs):
    f = attn_kwargs["f_dims"]
    if attn_kwargs.get("ray_to_se3", False):
        raise NotImplementedError("ray_to_se3 is dead code in the reference (ray2rotation is undefined)")
    return f


def _so3_override(attn_kwargs, vrep, L):
    """encoder.py:250-258 / decoder.py:337-345: ``zeroout_so3`` replaces every D^l of the freshly built view records by zeros, ``id_so3``
    (checked second, as in the reference) by identities.  ``vrep`` [B, N, VREP_STRIDE] is the builder's own output (nothing else holds it yet)."""
    if L <= 0 or not (attn_kwargs.get("zeroout_so3", False) or attn_kwargs.get("id_so3", False)):
        return vrep
    import torch
    blocks = ((native.VREP_D1, 3), (native.VREP_D2, 5))[:L]
    for off, n in blocks:
        if attn_kwargs.get("zeroout_so3", False):
            vrep[..., off:off + n * n] = 0.0
        else:
            vrep[..., off:off + n * n] = torch.eye(n, device=vrep.device, dtype=vrep.dtype).reshape(n * n)
    return vrep


def _flattened(vrep, cs, T):
    """``flattened_rep`` / ``flattened_invrep`` of the elementwise_mul ablation (encoder.py:200-243):
    per token [ so2 blocks (c,-s,s,c) | E^T (16) ] and [ (c,s,-s,c) | E (16) ]."""
    import torch
    parts, iparts = [], []
    if cs is not None:
        c, s = cs[..., 0], cs[..., 1]
        parts.append(torch.stack([c, -s, s, c], -1).flatten(-2, -1))
        iparts.append(torch.stack([c, s, -s, c], -1).flatten(-2, -1))
    if vrep is not None:
        B, N = vrep.shape[:2]
        E = vrep[..., native.VREP_INV:native.VREP_INV + 16].reshape(B, N, 4, 4).repeat_interleave(T // N, 1)
        parts.append(E.transpose(-1, -2).reshape(B, T, 16))
        iparts.append(E.reshape(B, T, 16))
    return torch.cat(parts, -1), torch.cat(iparts, -1)


def pre_compute_reps_encoder(attn_kwargs: dict, extras: dict) -> dict:
    """encoder.py:183-265: reps of the input views, q-side == k-side."""
    f = _check(attn_kwargs)
    need_view = f.get("se3", 0) > 0 or f.get("so3", 0) > 0
    L = attn_kwargs.get("so3", 0) if f.get("so3", 0) > 0 else 0
    if f.get("so2", 0) > 0 and need_view:            # both tables in one launch
        c = extras["input_coord"].reshape(extras["input_coord"].shape[0], -1, 2)
        vrep, cs = native.build_reps(extras["input_transforms"], L, c, attn_kwargs["so2"], attn_kwargs["max_freq_h"],
                                     attn_kwargs["max_freq_w"], attn_kwargs.get("shared_freqs", False))
        extras["gta_cs_q"] = extras["gta_cs_k"] = cs
        extras["gta_vrep_q"] = extras["gta_vrep_k"] = _so3_override(attn_kwargs, vrep, L)
        extras["gta_so3_degree"] = L
    elif f.get("so2", 0) > 0:
        extras["gta_cs_q"] = extras["gta_cs_k"] = _so2(attn_kwargs, extras["input_coord"])
    elif need_view:
        extras["gta_vrep_q"] = extras["gta_vrep_k"] = _so3_override(attn_kwargs, native.build_view_reps(extras["input_transforms"], L), L)
        extras["gta_so3_degree"] = L
    if f.get("t2", 0) > 0:                            # encoder.py:208-215: T2 reps are the raw token coordinates
        c = extras["input_coord"]
        extras["gta_coord_q"] = extras["gta_coord_k"] = c.reshape(c.shape[0], -1, 2).detach().float().contiguous()
    if attn_kwargs.get("elementwise_mul", False):
        T = extras["input_coord"].reshape(extras["input_coord"].shape[0], -1, 2).shape[1]
        fr, fi = _flattened(extras.get("gta_vrep_q") if f.get("se3", 0) > 0 else None, extras.get("gta_cs_q"), T)
        extras["flattened_rep_q"] = extras["flattened_rep_k"] = fr
        extras["flattened_invrep_q"] = fi
    return extras


def pre_compute_reps_decoder(attn_kwargs: dict, extras: dict) -> dict:
    """decoder.py:247-353: q-side from the target views; k-side kept from the encoder call."""
    f = _check(attn_kwargs)
    if f.get("so2", 0) > 0:
        extras["gta_cs_q"] = _so2(attn_kwargs, extras["target_coord"])
        if attn_kwargs.get("recompute_so2", False) or "gta_cs_k" not in extras:
            extras["gta_cs_k"] = _so2(attn_kwargs, extras["input_coord"])
    if f.get("se3", 0) > 0 or f.get("so3", 0) > 0:
        L = attn_kwargs.get("so3", 0) if f.get("so3", 0) > 0 else 0
        extras["gta_vrep_q"] = _so3_override(attn_kwargs, native.build_view_reps(extras["target_transforms"], L), L)
        if "gta_vrep_k" not in extras:                # (decoder.py:300-303 rebuilds only se3rep_k here; the so3 key side is the encoder's)
            extras["gta_vrep_k"] = native.build_view_reps(extras["input_transforms"], L)
        extras["gta_so3_degree"] = L
    if f.get("t2", 0) > 0:                            # decoder.py:283-290: q side only
        c = extras["target_coord"]
        extras["gta_coord_q"] = c.reshape(c.shape[0], -1, 2).detach().float().contiguous()
        if "gta_coord_k" not in extras:
            ci = extras["input_coord"]
            extras["gta_coord_k"] = ci.reshape(ci.shape[0], -1, 2).detach().float().contiguous()
    if attn_kwargs.get("elementwise_mul", False):
        T = extras["target_coord"].reshape(extras["target_coord"].shape[0], -1, 2).shape[1]
        fr, fi = _flattened(extras.get("gta_vrep_q") if f.get("se3", 0) > 0 else None, extras.get("gta_cs_q"), T)
        extras["flattened_rep_q"], extras["flattened_invrep_q"] = fr, fi     # *_k stays the encoder's
    return extras
