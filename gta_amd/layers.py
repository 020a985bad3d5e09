"""Transformer blocks with GTA attention -- drop-ins for ``source/layers.py``.

Same constructor arguments, ``forward`` signatures, ``extras`` contract and state-dict keys as
the reference's ``Attention`` (layers.py:172-444), ``Transformer`` (:447-488), ``PreNorm``
(:146-154), ``FeedForward`` (:157-169), ``JaxLinear`` (:14-25) and ``ViTLinear`` (:28-37), so a
reference checkpoint loads with ``load_state_dict(strict=True)``.  The attention core is the
fused HIP kernel.  On an MI355X ``Transformer`` runs each layer as the fused block of ``gta_amd.fused`` (LayerNorm+cast
kernel, hipBLASLt GEMMs with bias / skip / GELU epilogues, LayerNorm backward fused with the skip gradient; SURVEY.md
section 8 row f1); ``Transformer.fused_blocks = False`` on an instance (env ``GTA_FUSED_BLOCKS=0`` sets the default at construction)
keeps the module-by-module path, which is also what ``Attention`` / ``PreNorm`` / ``FeedForward`` do when called on their own.
Only ``method: gta`` is built -- the other positional-encoding baselines of the reference
(repast/ape/mln/gbt/rpe/frustum) are out of scope.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
from torch.nn import init

from . import fused as _fused
from . import gta as _gta
from . import native

def _fused_blocks_default() -> bool:
    """construction-time default of ``Transformer.fused_blocks``: run layers as fused blocks (gta_amd.fused) where they qualify"""
    return os.environ.get("GTA_FUSED_BLOCKS", "1") != "0"


class JaxLinear(nn.Linear):
    """Truncated-normal(std = fan_in^-1/2, +-2 std) weights, zero bias (layers.py:14-25)."""

    def reset_parameters(self):
        std = math.sqrt(1.0 / self.weight.shape[-1])
        init.trunc_normal_(self.weight, std=std, a=-2.0 * std, b=2.0 * std)
        if self.bias is not None:
            init.zeros_(self.bias)


class ViTLinear(nn.Linear):
    """Xavier-uniform weights, N(0, 1e-6) bias (layers.py:28-37)."""

    def reset_parameters(self):
        init.xavier_uniform_(self.weight)
        if self.bias is not None:
            init.normal_(self.bias, std=1e-6)


class TemperatureAdjustableSoftmax(nn.Module):
    """Holds the learnable temperature ``tau`` (layers.py:135-143); the softmax runs in the kernel."""

    def __init__(self, init_tau=1.0):
        super().__init__()
        self.tau = nn.Parameter(torch.tensor([float(init_tau)]))


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(self.norm(x), **kwargs)   # LayerNorm on x only, never on z (layers.py:153-154)


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.0, linear_module=ViTLinear):
        super().__init__()
        self.net = nn.Sequential(
            linear_module(dim, hidden_dim),
            nn.GELU(),
            nn.Dropout(dropout) if dropout > 0.0 else nn.Identity(),
            linear_module(hidden_dim, dim),
            nn.Dropout(dropout) if dropout > 0.0 else nn.Identity(),
        )

    def forward(self, x):
        return self.net(x)


class Attention(nn.Module):
    """GTA multi-head attention (layers.py:172-444, ``method == 'gta'`` branch)."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0, kv_dim=None, attn_args=None,
                 linear_module=JaxLinear, **kwargs):
        super().__init__()
        attn_args = attn_args or {}
        inner_dim = dim_head * heads
        project_out = not (heads == 1 and dim_head == dim)
        self.selfatt = kv_dim is None
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.method = attn_args["method"]["name"]
        self.method_args = attn_args["method"]["args"]
        if self.method != "gta":
            raise NotImplementedError(f"attention method {self.method!r}: only 'gta' is built (see DESIGN.md)")
        self.elementwise_mul = self.method_args.get("elementwise_mul", False)
        if self.method_args.get("rpe", False):
            raise NotImplementedError("rpe baseline is not built")
        use_bias = self.method_args.get("use_bias", False)
        self.f_dims = dict(self.method_args["f_dims"])
        if sum(self.f_dims.values()) != dim_head:
            raise ValueError(f"f_dims {self.f_dims} must sum to dim_head={dim_head}")
        if self.f_dims.get("se3", 0) > 0 and not self.elementwise_mul:
            self.trans_coeff = nn.Parameter(torch.tensor([0.01]))        # layers.py:188-193
        else:
            self.trans_coeff = None
        if attn_args.get("softmax") == "adjustable":
            self.attend = TemperatureAdjustableSoftmax(1.0)               # key 'attend.tau'
        else:
            self.attend = None
        self.euclid = self.method_args.get("euclid_sim", False)
        # fp32-faithful products for float32 inputs (split-bf16 operands): a per-module setting -- the reference's
        # ``mixed_prec: False`` configs (runs/clevrtr/GTA/gta/config.yaml:55) set it through ``attn_args['method']['args']['precise']``
        # or on the instance; bf16 / autocast inputs are not affected
        self.precise = bool(self.method_args.get("precise", False))
        if kv_dim is not None:
            self.to_q = linear_module(dim, inner_dim, bias=use_bias)
            self.to_kv = linear_module(kv_dim, 2 * inner_dim, bias=use_bias)
        else:
            self.to_qkv = linear_module(dim, 3 * inner_dim, bias=use_bias)
        if self.elementwise_mul:                                          # layers.py:265-270
            freqs = self.f_dims.get("so2", 0) // 4
            self.rep_to_vec = nn.Linear(16 + 2 * freqs * 2 * 2, inner_dim // heads)
        self.to_out = nn.Sequential(linear_module(inner_dim, dim), nn.Dropout(dropout)) if project_out \
            else nn.Identity()

    def project(self, x, z, extras, q_packed=None):
        """(q, k, v) as [B,H,T,dh] strided views of the packed projections (layers.py:388-395) + the decode cache.
        ``q_packed``: the query-side projection when the caller has already run it (fused LayerNorm + GEMM)."""
        B, Tq, _ = x.shape
        H, dh = self.heads, self.dim_head
        kv_cache = None
        if z is None:
            qkv = (self.to_qkv(x) if q_packed is None else q_packed).view(B, Tq, 3, H, dh)      # layers.py:389
            q, k, v = _gta.split_packed(qkv)                                  # strided views, no copy (nor in the backward)
        else:
            q = (self.to_q(x) if q_packed is None else q_packed).view(B, Tq, H, dh).permute(0, 2, 1, 3)   # layers.py:391-392
            # chunked decode (srt.render_image): the key side of a cross-attention layer does not change between
            # query chunks, so its projection and its K'/V' images are kept in the caller's cache
            if "gta_kv_cache" in extras and not torch.is_grad_enabled():
                kv_cache = extras["gta_kv_cache"].setdefault(id(self), {})
            if kv_cache is not None and "kv" in kv_cache:
                k, v = kv_cache["kv"]
            else:
                kv = self.to_kv(z).view(B, z.shape[1], 2, H, dh)
                k, v = _gta.split_packed(kv)
                if kv_cache is not None:
                    kv_cache["kv"] = (k, v)
        return q, k, v, kv_cache

    def core(self, q, k, v, extras, kv_cache=None, return_attmap=False):
        """rho, softmax(QK^T)V, rho^-1 (layers.py:409-428) -> [B,Tq,H*dh] (the input of ``to_out``), attmap or None."""
        B, H, Tq, dh = q.shape
        tau = self.attend.tau if self.attend is not None else None
        if self.elementwise_mul:                                          # layers.py:410-419
            ex = dict(vecrep_q=self.rep_to_vec(extras["flattened_rep_q"]), vecrep_k=self.rep_to_vec(extras["flattened_rep_k"]),
                      vecinvrep_q=self.rep_to_vec(extras["flattened_invrep_q"]))
            out, _ = _gta.multihead_vecrep_attention(q, k, v, attn_fn=self, extras=ex, tau=tau)
            out = out.permute(0, 2, 1, 3).reshape(B, Tq, H * dh)
            attn = None
            if return_attmap:
                attn = _gta.attention_map(ex["vecrep_q"][:, None] * q, ex["vecrep_k"][:, None] * k, {"triv": dh}, {},
                                          tau=tau, scale=self.scale)
            return out, attn
        packed = _gta.pack_reps(extras, self.f_dims)
        out = _gta.gta_attention(
            q, k, v, self.f_dims, packed,
            so3_degree=_gta._so3_degree(self.f_dims, packed, extras),
            trans_coeff=self.trans_coeff, tau=tau,
            scale=self.scale, v_transform=self.method_args.get("v_transform", True), euclid=self.euclid,
            kv_cache=kv_cache, precise=self.precise and q.dtype == torch.float32 and kv_cache is None)
        out = out.permute(0, 2, 1, 3).reshape(B, Tq, H * dh)              # free: out is [B,Tq,H,dh] in memory
        attn = None
        if return_attmap:                                                  # layers.py:441-442
            attn = _gta.attention_map(q, k, self.f_dims, packed, so3_degree=_gta._so3_degree(self.f_dims, packed, extras),
                                      trans_coeff=self.trans_coeff, tau=tau, scale=self.scale, euclid=self.euclid)
        return out, attn

    def forward(self, x, z=None, return_attmap=False, extras=None):
        if extras is None:
            raise ValueError("GTA attention needs `extras` (the reps dict)")
        q, k, v, kv_cache = self.project(x, z, extras)
        out, attn = self.core(q, k, v, extras, kv_cache, return_attmap)
        out = self.to_out(out)
        return (out, attn) if return_attmap else out


_PLAIN_LINEARS = (nn.Linear, JaxLinear, ViTLinear)       # Linear classes whose forward is F.linear(x, weight, bias) and nothing else


def _global_forward_hooks() -> bool:
    """module-wide hooks registered through torch.nn.modules.module.register_module_forward(_pre)_hook"""
    import torch.nn.modules.module as M
    return bool(getattr(M, "_global_forward_hooks", None)) or bool(getattr(M, "_global_forward_pre_hooks", None))


class Transformer(nn.Module):
    """Pre-LN Transformer (layers.py:447-488)."""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.0, selfatt=True, kv_dim=None,
                 return_last_attmap=False, attn_args=None):
        super().__init__()
        self.heads = heads
        self.layers = nn.ModuleList([])
        dropout = 0.0 if dropout is None else dropout
        for _ in range(depth):
            attn = PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout, selfatt=selfatt,
                                          kv_dim=kv_dim, attn_args=attn_args, linear_module=JaxLinear))
            ff = PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout, linear_module=ViTLinear))
            self.layers.append(nn.ModuleList([attn, ff]))
        self.return_last_attmap = return_last_attmap
        self.fused_blocks = _fused_blocks_default()       # per instance: False keeps the module-by-module path

    def _fused_dtype(self, x, attn, ff):
        """Compute dtype of the fused block for this layer and input, or None -> module-by-module path."""
        if not self.fused_blocks or not isinstance(attn.fn, Attention):
            return None
        cdt = _fused.compute_dtype(x)
        if cdt is None or x.dim() != 3:
            return None
        a, net = attn.fn, ff.fn.net
        if not (_fused.norm_ok(attn.norm) and _fused.norm_ok(ff.norm) and isinstance(a.to_out, nn.Sequential)):
            return None
        lins = [a.to_qkv if a.selfatt else a.to_q, a.to_out[0], net[0], net[3]]
        if any(l.weight.dtype != torch.float32 or l.in_features % 4 or l.out_features % 4 for l in lins):
            return None
        # The fused block reads the sub-modules' parameters and never calls their forward: anything hung on those calls -- forward /
        # pre-forward hooks (weight_norm's, activation capture, observers), wrappers that override forward (per-module FSDP,
        # checkpointing) -- would be skipped silently.  Such layers take the module-by-module path.
        mods = [attn, attn.norm, a, a.to_out, a.to_out[1], ff, ff.norm, ff.fn, net, net[1], net[2], net[4]] + lins
        if any(m._forward_hooks or m._forward_pre_hooks for m in mods) or _global_forward_hooks():
            return None
        if any(type(l) not in _PLAIN_LINEARS for l in lins) or type(attn.norm).forward is not nn.LayerNorm.forward \
                or type(ff.norm).forward is not nn.LayerNorm.forward:
            return None
        if (x.shape[0] * x.shape[1]) % 2:                  # the elementwise kernels work on multiples of 8 elements
            return None
        return cdt

    def forward(self, x, z=None, extras=None):
        attmap = None
        for l, (attn, ff) in enumerate(self.layers):
            want_map = l == len(self.layers) - 1 and self.return_last_attmap
            cdt = self._fused_dtype(x, attn, ff) if extras is not None else None
            if cdt is not None:
                a = attn.fn
                q_packed, skip = _fused.ln_linear(x, attn.norm, a.to_qkv if z is None else a.to_q, cdt)
                q, k, v, kv_cache = a.project(x, z, extras, q_packed=q_packed)
                out, amap = a.core(q, k, v, extras, kv_cache, return_attmap=want_map)
                if want_map:
                    attmap = amap
                net = ff.fn.net
                x = _fused.linear_skip(out, a.to_out[0], skip, cdt, p=_fused.active_p(a.to_out[1]))   # to_out + `+ x` (:483-486)
                x = _fused.feed_forward_skip(x, ff.norm, net[0], net[3], cdt, p_mid=_fused.active_p(net[2]),
                                             p_out=_fused.active_p(net[4]))                            # ff(norm(x)) + x (:487)
                continue
            if want_map:
                out, attmap = attn(x, z=z, return_attmap=True, extras=extras)
                x = out + x
            else:
                x = attn(x, z=z, extras=extras) + x
            x = ff(x) + x
        return (x, attmap) if self.return_last_attmap else x
