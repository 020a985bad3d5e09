"""Fused pre-LN Transformer block around the GTA attention operator.

What ``Transformer.forward`` (source/layers.py:475-488) does per layer, ``x = attn(norm(x)) + x; x = ff(norm(x)) + x``,
as these launches (``gta_block.h``):

    LayerNorm -> compute dtype        gta_ln_fwd           (PreNorm, layers.py:146-154, + autocast's cast kernel)
    QKV / Q projection                gta_gemm             (to_qkv / to_q, layers.py:388-392)
    rho, softmax(QK^T)V, rho^-1       gta_attn_fwd         (gta.py:92-279; unchanged)
    out-proj + bias + skip            gta_gemm, epilogue   (to_out, layers.py:430, and `+ x`, :483-486)
    LayerNorm -> compute dtype        gta_ln_fwd
    Linear + bias (+ GELU)            gta_gemm, epilogue   (net[0], net[1], layers.py:161-162)
    Linear + bias + skip              gta_gemm, epilogue   (net[3] and `+ x`, layers.py:164,487)

and, in the backward, the LayerNorm gradient fused with the skip connection's (``gta_ln_bwd``), weight and bias
gradients from the hand-written TN kernel (``gta_wgrad``) in fp32 straight from the bf16 operands.

Dropout (both reference configs train with ``dropout: 0.01``; layers.py:163,165,289): the three nn.Dropout of a layer run
inside the block's own kernels -- after the GELU in the GELU kernel, in front of the two skip additions in
``gta_dropout_add`` (the skip epilogue of the GEMM is then not used) -- with masks that are a pure function of a per-call
seed (drawn from torch's CPU generator, so ``torch.manual_seed`` fixes them) and the element index; the backward
regenerates them.  The masks are NOT torch's Philox stream: runs are reproducible, not bit-identical to nn.Dropout's.

Arithmetic.  The compute dtype is autocast's (bf16) when autocast is on, else the dtype of ``x``.  The residual stream
keeps the dtype of ``x`` (fp32 under autocast, as in the reference's mixed_prec runs), LayerNorm statistics and every
accumulation are fp32.  GELU: nn.GELU() is the erf form.  hipBLASLt's GELU epilogue is the tanh form (max |difference|
4.7e-4, below one bf16 ulp of the values it occurs at); it is used only in bf16 inference (``GELU_EPILOGUE``); training
and fp32 use the exact kernels (``gta_gelu_fwd`` / ``gta_gelu_bwd``), so gradients match the forward they belong to.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import native_block as nb

# bf16 inference only: fold the GELU into the GEMM epilogue (tanh form, see the module docstring)
GELU_EPILOGUE = True


def compute_dtype(x: torch.Tensor) -> Optional[torch.dtype]:
    """The dtype the block's GEMMs run in for this input, or None when the fused path does not apply."""
    if not x.is_cuda:
        return None
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda")
        return dt if dt == torch.bfloat16 else None
    return x.dtype if x.dtype in (torch.float32, torch.bfloat16) else None


# Keep the bf16 copy of an fp32 weight on the parameter between calls (opt-in).  Off by default: the copy can only be tied to the
# parameter's version counter and storage address, and an update THROUGH ``.data`` (``p.data.mul_()``, an EMA's
# ``ema_p.data.mul_(d).add_(p.data)``, ``p.data.copy_(w)``) changes neither -- the fused path would keep multiplying by stale
# weights.  A cast is a few microseconds per weight (four per layer); the backward gets the forward's copy through the autograd context.
CACHE_WEIGHT_CASTS = False


def clear_weight_casts(module: torch.nn.Module) -> None:
    """Drop every cached weight copy under ``module`` (only meaningful with ``CACHE_WEIGHT_CASTS``)."""
    for p in module.parameters():
        if hasattr(p, "_gta_cast"):
            del p._gta_cast


def _cast_param(p: torch.Tensor, dt: torch.dtype) -> torch.Tensor:
    """p in the compute dtype: a fresh cast per call (safe under any way of updating the parameter), or -- with
    ``CACHE_WEIGHT_CASTS`` -- a copy kept on the parameter until its version counter or storage address changes."""
    if p.dtype == dt:
        return p.detach()
    if not CACHE_WEIGHT_CASTS:
        return p.detach().to(dt)
    key = (p._version, p.data_ptr())
    tag = getattr(p, "_gta_cast", None)
    if tag is not None and tag[0] == key and tag[1].dtype == dt and tag[1].device == p.device:
        return tag[1]
    c = p.detach().to(dt)
    p._gta_cast = (key, c)
    return c


def _bias_for(b: Optional[torch.Tensor], out_dtype: torch.dtype) -> Optional[torch.Tensor]:
    """hipBLASLt takes the bias in fp32 or in the output's dtype."""
    if b is None:
        return None
    b = b.detach()
    return b if b.dtype in (torch.float32, out_dtype) else b.float()


def _shaped(dx: torch.Tensor, shape) -> torch.Tensor:
    """dx viewed in the input's shape, keeping the bf16 side copy attached (a view is a new Python object)."""
    out = dx.view(shape)
    side = getattr(dx, "_gta_bf16", None)
    if side is not None:
        out._gta_bf16 = (side[0].view(shape), side[1])      # a view shares its base's version counter
    return out


def _rows(x: torch.Tensor) -> torch.Tensor:
    return x.reshape(-1, x.shape[-1]) if x.is_contiguous() else x.contiguous().view(-1, x.shape[-1])


def next_seed() -> int:
    """A fresh dropout seed from torch's default CPU generator (reproducible under torch.manual_seed)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def _in_compute_dtype(d2: torch.Tensor, dout: torch.Tensor, cdt: torch.dtype) -> torch.Tensor:
    """The incoming gradient in the GEMMs' dtype: the bf16 copy the LayerNorm backward of the block downstream wrote
    beside its fp32 result when there is one (``native_block.ln_bwd(bf16_copy=True)``), else a cast."""
    if d2.dtype == cdt:
        return d2
    side = getattr(dout, "_gta_bf16", None)
    if (side is not None and cdt == torch.bfloat16 and side[1] == dout._version and side[0].shape == dout.shape
            and side[0].device == dout.device):
        return side[0].view(d2.shape)
    return d2.to(cdt)


def _weight_grads(g2: torch.Tensor, a2: torch.Tensor, want_w: bool, want_b: bool, wdtype: torch.dtype):
    """(dW = g2^T a2, db = column sums of g2) for y = a2 W^T + b: the hand-written TN kernel where it applies (bf16
    operands, fp32 master weights, tile-aligned shapes), else hipBLASLt's transposed GEMM + the column-sum kernel."""
    dW = db = None
    if want_w and wdtype == torch.float32 and nb.wgrad_supported(g2, a2):
        dW, db = nb.wgrad(g2, a2, want_b)
        return dW, db
    if want_w:
        dW = nb.gemm(g2, a2, trans_a=True, out_dtype=wdtype)
    if want_b:
        db = nb.colsum(g2).to(wdtype)
    return dW, db


class _LNLinear(torch.autograd.Function):
    """(x W^T + b of LayerNorm(x), x): the second output is x itself, handed to the block's skip connection so that the
    gradient arriving through the skip is added inside the LayerNorm backward kernel instead of by autograd."""

    @staticmethod
    def forward(ctx, x, gamma, beta, W, bias, eps, cdt, need):
        x2 = _rows(x)
        y, mean, rstd = nb.ln_fwd(x2, gamma.detach(), beta.detach(), eps, cdt, want_stats=need)
        Wc = _cast_param(W, cdt)
        b = _bias_for(bias, cdt)
        out = nb.gemm(y, Wc, trans_b=True, epilogue=nb.EPI_BIAS if b is not None else nb.EPI_NONE, bias=b)
        if need:
            ctx.save_for_backward(x2, gamma, beta, mean, rstd, Wc, y)       # y kept: 2 B per element against a 35 us recompute
            ctx.eps, ctx.cdt, ctx.has_bias, ctx.wdtype = eps, cdt, bias is not None, W.dtype
            ctx.xshape = x.shape
        return out.view(*x.shape[:-1], W.shape[0]), x

    @staticmethod
    def backward(ctx, dout, dskip):
        x2, gamma, beta, mean, rstd, Wc, y = ctx.saved_tensors
        d2 = _rows(dout)
        if d2.dtype != ctx.cdt:
            d2 = d2.to(ctx.cdt)
        dW, db = _weight_grads(d2, y, ctx.needs_input_grad[3], ctx.has_bias and ctx.needs_input_grad[4], ctx.wdtype)
        dy = nb.gemm(d2, Wc)
        dres = None
        if dskip is not None:
            dres = _rows(dskip)
            if dres.dtype != x2.dtype:
                dres = dres.to(x2.dtype)
        dx, dgamma, dbeta = nb.ln_bwd(dy, x2, gamma.detach(), mean, rstd, dres, bf16_copy=ctx.cdt == torch.bfloat16)
        return _shaped(dx, ctx.xshape), dgamma.to(gamma.dtype), dbeta.to(beta.dtype), dW, db, None, None, None


class _LinearSkip(torch.autograd.Function):
    """skip + dropout_p(a W^T + b).  p = 0: one GEMM (bias epilogue, beta = 1 on the skip); p > 0: GEMM with bias, then
    ``gta_dropout_add``.  Output in the skip's dtype."""

    @staticmethod
    def forward(ctx, a, W, bias, skip, cdt, need, p, seed):
        a2 = _rows(a)
        if a2.dtype != cdt:
            a2 = a2.to(cdt)
        s2 = _rows(skip)
        Wc = _cast_param(W, cdt)
        if p > 0.0:
            b = _bias_for(bias, cdt)
            z = nb.gemm(a2, Wc, trans_b=True, epilogue=nb.EPI_BIAS if b is not None else nb.EPI_NONE, bias=b)
            out = nb.dropout_add(z, s2, p, seed)
        else:
            b = _bias_for(bias, s2.dtype)
            out = nb.gemm(a2, Wc, trans_b=True, epilogue=nb.EPI_BIAS if b is not None else nb.EPI_NONE, bias=b, c=s2, beta=1.0,
                          out_dtype=s2.dtype)
        if need:
            ctx.save_for_backward(a2, Wc)
            ctx.cdt, ctx.has_bias, ctx.wdtype, ctx.ashape, ctx.adtype = cdt, bias is not None, W.dtype, a.shape, a.dtype
            ctx.p, ctx.seed = p, seed
        return out.view(skip.shape)

    @staticmethod
    def backward(ctx, dout):
        a2, Wc = ctx.saved_tensors
        d2 = _rows(dout)
        dc = _in_compute_dtype(d2, dout, ctx.cdt)
        if ctx.p > 0.0:
            dc = nb.dropout_bwd(dc, ctx.cdt, ctx.p, ctx.seed)
        da = nb.gemm(dc, Wc).view(ctx.ashape) if ctx.needs_input_grad[0] else None
        if da is not None and da.dtype != ctx.adtype:
            da = da.to(ctx.adtype)
        dW, db = _weight_grads(dc, a2, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2], ctx.wdtype)
        return da, dW, db, dout, None, None, None, None


class _FeedForwardSkip(torch.autograd.Function):
    """x + drop(W2 drop(gelu(W1 LayerNorm(x) + b1)) + b2)  (PreNorm(FeedForward) + skip, layers.py:146-169,487)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, W1, b1, W2, b2, eps, cdt, need, p_mid, p_out, seed):
        x2 = _rows(x)
        y, mean, rstd = nb.ln_fwd(x2, gamma.detach(), beta.detach(), eps, cdt, want_stats=need)
        W1c, W2c = _cast_param(W1, cdt), _cast_param(W2, cdt)
        bb1 = _bias_for(b1, cdt)
        epi = nb.EPI_BIAS if bb1 is not None else nb.EPI_NONE
        if not need and p_mid == 0.0 and GELU_EPILOGUE and cdt == torch.bfloat16 and bb1 is not None:
            pre = None
            h = nb.gemm(y, W1c, trans_b=True, epilogue=nb.EPI_BIAS_GELU, bias=bb1)
        else:
            pre = nb.gemm(y, W1c, trans_b=True, epilogue=epi, bias=bb1)
            h = nb.gelu_fwd(pre, p_mid, seed)
        if p_out > 0.0:
            bb2 = _bias_for(b2, cdt)
            z = nb.gemm(h, W2c, trans_b=True, epilogue=nb.EPI_BIAS if bb2 is not None else nb.EPI_NONE, bias=bb2)
            out = nb.dropout_add(z, x2, p_out, seed + 1)
        else:
            bb2 = _bias_for(b2, x2.dtype)
            out = nb.gemm(h, W2c, trans_b=True, epilogue=nb.EPI_BIAS if bb2 is not None else nb.EPI_NONE, bias=bb2, c=x2, beta=1.0,
                          out_dtype=x2.dtype)
        if need:
            ctx.save_for_backward(x2, gamma, beta, mean, rstd, W1c, W2c, pre, h, y)
            ctx.eps, ctx.cdt, ctx.wdtype, ctx.xshape = eps, cdt, W1.dtype, x.shape
            ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None
            ctx.p_mid, ctx.p_out, ctx.seed = p_mid, p_out, seed
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, dout):
        x2, gamma, beta, mean, rstd, W1c, W2c, pre, h, y = ctx.saved_tensors
        d2 = _rows(dout)
        dc = _in_compute_dtype(d2, dout, ctx.cdt)
        if ctx.p_out > 0.0:
            dc = nb.dropout_bwd(dc, ctx.cdt, ctx.p_out, ctx.seed + 1)
        dW2, db2 = _weight_grads(dc, h, ctx.needs_input_grad[5], ctx.has_b2 and ctx.needs_input_grad[6], ctx.wdtype)
        dh = nb.gemm(dc, W2c)
        dpre = nb.gelu_bwd(dh, pre, ctx.p_mid, ctx.seed)
        del dh
        dW1, db1 = _weight_grads(dpre, y, ctx.needs_input_grad[3], ctx.has_b1 and ctx.needs_input_grad[4], ctx.wdtype)
        dy = nb.gemm(dpre, W1c)
        dres = d2 if d2.dtype == x2.dtype else d2.to(x2.dtype)
        dx, dgamma, dbeta = nb.ln_bwd(dy, x2, gamma.detach(), mean, rstd, dres, bf16_copy=ctx.cdt == torch.bfloat16)
        return (_shaped(dx, ctx.xshape), dgamma.to(gamma.dtype), dbeta.to(beta.dtype), dW1, db1, dW2, db2, None, None, None,
                None, None, None)


def _need(*ts) -> bool:
    """Will a backward run through this call?  (Inside Function.forward grad mode is always off, so it is decided here.)"""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def ln_linear(x, norm: torch.nn.LayerNorm, lin: torch.nn.Linear, cdt):
    """-> (lin(norm(x)) in the compute dtype, x for the skip connection)."""
    args = (x, norm.weight, norm.bias, lin.weight, lin.bias)
    return _LNLinear.apply(*args, norm.eps, cdt, _need(*args))


def linear_skip(a, lin: torch.nn.Linear, skip, cdt, p: float = 0.0, seed=None):
    """skip + dropout_p(lin(a)); ``seed`` None draws one (``next_seed``)."""
    args = (a, lin.weight, lin.bias, skip)
    p = float(p)
    return _LinearSkip.apply(*args, cdt, _need(*args), p, (next_seed() if seed is None else seed) if p > 0.0 else 0)


def feed_forward_skip(x, norm: torch.nn.LayerNorm, lin1: torch.nn.Linear, lin2: torch.nn.Linear, cdt, p_mid: float = 0.0,
                      p_out: float = 0.0, seed=None):
    """x + dropout_{p_out}(lin2(dropout_{p_mid}(gelu(lin1(norm(x))))))."""
    args = (x, norm.weight, norm.bias, lin1.weight, lin1.bias, lin2.weight, lin2.bias)
    p_mid, p_out = float(p_mid), float(p_out)
    seed = (next_seed() if seed is None else seed) if (p_mid > 0.0 or p_out > 0.0) else 0
    return _FeedForwardSkip.apply(*args, norm.eps, cdt, _need(*args), p_mid, p_out, seed)


def active_p(m) -> float:
    """Dropout probability a module applies right now (0 for nn.Identity and in eval mode)."""
    return float(m.p) if isinstance(m, torch.nn.Dropout) and m.training else 0.0


def norm_ok(norm) -> bool:
    if not (isinstance(norm, torch.nn.LayerNorm) and norm.elementwise_affine and norm.bias is not None
            and len(norm.normalized_shape) == 1 and norm.weight.dtype == torch.float32):
        return False
    d = norm.normalized_shape[0]
    return (d % 8 == 0 and d <= 4096) or (d % 4 == 0 and d <= 2048)       # gta_ln_fwd's row shapes (gta_block.h)
