"""Shared helpers for the GPU parity tests and tools/gpu_check.py (test infrastructure)."""
from types import SimpleNamespace

import numpy as np
import torch

import gta_amd
from gta_amd import native
from oracle import gta_oracle as O
from tests import _golden as G

FUSED_OK = lambda meta: (sum(meta["f_dims"].values()) % 8 == 0 and not meta["euclid"]
                         and meta["f_dims"].get("t2", 0) == 0)


def err_stats(got: torch.Tensor, ref: torch.Tensor):
    got, ref = got.double().cpu(), ref.double().cpu()
    diff = (got - ref)
    return {"max_abs": diff.abs().max().item(), "ref_max": ref.abs().max().item(),
            "rel_rms": (diff.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-30)).item(),
            "finite": bool(torch.isfinite(got).all())}


def golden_forward(case, dtype, use_dma=True, builder="packed", device="cuda", kv_mode="auto"):
    """Run one op_* fixture through the C ABI.  builder='packed': reference-style reps from the
    fixture are packed by gta_amd.pack_reps; builder='hip': reps are rebuilt on device from the
    fixture's poses/coords by gta_amd.reps (HIP rep builders)."""
    d, meta = G.load("op_" + case)
    f_dims = meta["f_dims"]
    ak = G.attn_kwargs_of(meta)
    if builder == "packed":
        ex = G.extras_of(d, torch.float32, device)
    else:
        ex = {k[len("extras."):]: torch.from_numpy(v).float().to(device) for k, v in d.items()
              if k in ("extras.input_transforms", "extras.target_transforms", "extras.input_coord",
                       "extras.target_coord")}
        gta_amd.pre_compute_reps_encoder(ak, ex)
        if meta["cross"]:
            gta_amd.pre_compute_reps_decoder(ak, ex)
    q, k, v = (torch.from_numpy(d[n]).to(dtype).to(device) for n in "qkv")
    tc = torch.tensor([float(d["trans_coeff"])], device=device)
    out, _ = gta_amd.multihead_geometric_transform_attention(
        q, k, v, attn_fn=SimpleNamespace(scale=float(d["scale"]), tau=G.tau_of(d, torch.float32, device, grad=False)),
        f_dims=f_dims, reps=ex, trans_coeff=tc,
        v_transform=meta["v_transform"], euclid=meta["euclid"], use_dma=use_dma, kv_mode=kv_mode)
    torch.cuda.synchronize()
    return out.float().cpu(), torch.from_numpy(d["out"]).float(), meta


def synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, dtype, seed=0, device="cuda", cross=None,
                 coord_mode="grid"):
    """Seeded synthetic (q,k,v,extras) following SURVEY 8d (gta_amd.synth).  Returns CPU fp32 masters."""
    from gta_amd import synth
    return synth.attention_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, seed=seed, cross=cross)


def oracle_forward(q, k, v, ex, ak, cross, trans_coeff, v_transform=True, dtype=torch.float32):
    reps = O.encoder_reps(ak, {kk: vv.to(dtype) for kk, vv in ex.items()})
    if cross:
        reps = O.decoder_reps(ak, {kk: vv.to(dtype) for kk, vv in ex.items()}, reps)
    out, _ = O.gta_attention(q.to(dtype), k.to(dtype), v.to(dtype), ak["f_dims"], reps, trans_coeff, v_transform)
    return out


def hip_forward(q, k, v, ex, ak, cross, trans_coeff, dtype, v_transform=True, use_dma=True, device="cuda",
                return_lse=False, kv_mode="auto"):
    exd = {kk: vv.to(device) for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, ak["f_dims"])
    tc = torch.tensor([float(trans_coeff)], device=device) if ak["f_dims"].get("se3", 0) > 0 else None
    out = gta_amd.gta_attention(q.to(dtype).to(device), k.to(dtype).to(device), v.to(dtype).to(device),
                                ak["f_dims"], packed, so3_degree=exd.get("gta_so3_degree", 0), trans_coeff=tc,
                                v_transform=v_transform, use_dma=use_dma, kv_mode=kv_mode)
    torch.cuda.synchronize()
    return out
