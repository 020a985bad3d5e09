"""ctypes binding of libgta_block.so (the C ABI declared in include/gta_block.h): LayerNorm / GELU / column-sum
kernels and hipBLASLt GEMMs with epilogues -- the pre-LN Transformer block around the attention operator.

No CPU or eager fallback: a missing library or a refused request raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int32, c_int64, c_void_p
from typing import Optional

import torch

from .native import GtaError, DTYPE_BF16, DTYPE_F32, _ptr

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GTA_BLOCK_LIB") or os.path.join(_HERE, "csrc", "libgta_block.so")

BLOCK_ABI_VERSION = 1
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_AUX, EPI_DGELU, EPI_DGELU_BGRAD, EPI_BGRAD_A = range(7)

# every symbol include/gta_block.h declares
ABI_SYMBOLS = (
    "gta_ln_fwd", "gta_ln_bwd", "gta_ln_bwd_workspace_bytes", "gta_gelu_fwd", "gta_gelu_bwd", "gta_colsum",
    "gta_colsum_workspace_bytes", "gta_gemm", "gta_gemm_workspace_bytes", "gta_block_release", "gta_block_strerror",
    "gta_block_abi_version", "gta_sizeof_gemm_desc", "gta_wgrad", "gta_wgrad_supported", "gta_wgrad_workspace_bytes",
    "gta_dropout_add", "gta_dropout_bwd",
)


class GtaGemmDesc(ctypes.Structure):
    _fields_ = [
        ("abi_version", c_int32), ("epilogue", c_int32), ("m", c_int64), ("n", c_int64), ("k", c_int64),
        ("trans_a", c_int32), ("trans_b", c_int32), ("a_dtype", c_int32), ("b_dtype", c_int32), ("d_dtype", c_int32),
        ("bias_dtype", c_int32), ("aux_dtype", c_int32), ("_pad", c_int32),
        ("lda", c_int64), ("ldb", c_int64), ("ldc", c_int64), ("ldd", c_int64), ("ldaux", c_int64),
        ("alpha", c_float), ("beta", c_float),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GtaError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"or `make -C gta_amd/csrc`. gta_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        L.gta_block_strerror.restype = ctypes.c_char_p
        L.gta_block_strerror.argtypes = [ctypes.c_int]
        L.gta_block_abi_version.restype = ctypes.c_int
        L.gta_sizeof_gemm_desc.restype = ctypes.c_int
        if L.gta_block_abi_version() != BLOCK_ABI_VERSION:
            raise GtaError("libgta_block.so ABI version mismatch")
        if L.gta_sizeof_gemm_desc() != ctypes.sizeof(GtaGemmDesc):
            raise GtaError("GtaGemmDesc layout mismatch between gta_block.h and gta_amd/native_block.py")
        L.gta_ln_fwd.argtypes = [c_void_p, c_int32, c_void_p, c_void_p, c_float, c_int64, c_int32, c_void_p, c_int32,
                                 c_void_p, c_void_p, c_void_p]
        L.gta_ln_bwd_workspace_bytes.argtypes = [c_int64, c_int32]
        L.gta_ln_bwd_workspace_bytes.restype = c_int64
        L.gta_ln_bwd.argtypes = [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                 c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]
        L.gta_gelu_fwd.argtypes = [c_void_p, c_void_p, c_int32, c_int64, c_float, ctypes.c_uint64, c_void_p]
        L.gta_gelu_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_float, ctypes.c_uint64, c_void_p]
        L.gta_dropout_add.argtypes = [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int64, c_float, ctypes.c_uint64, c_void_p]
        L.gta_dropout_bwd.argtypes = [c_void_p, c_int32, c_void_p, c_int32, c_int64, c_float, ctypes.c_uint64, c_void_p]
        L.gta_colsum_workspace_bytes.argtypes = [c_int64, c_int32]
        L.gta_colsum_workspace_bytes.restype = c_int64
        L.gta_colsum.argtypes = [c_void_p, c_int32, c_int64, c_int32, c_int64, c_void_p, c_void_p, c_int64, c_void_p]
        L.gta_gemm_workspace_bytes.restype = c_int64
        L.gta_gemm.argtypes = [ctypes.POINTER(GtaGemmDesc)] + [c_void_p] * 7 + [c_int64, c_void_p]
        L.gta_block_release.restype = None
        L.gta_wgrad_supported.argtypes = [c_int64, c_int64, c_int64]
        L.gta_wgrad_workspace_bytes.argtypes = [c_int64, c_int64, c_int64]
        L.gta_wgrad_workspace_bytes.restype = c_int64
        L.gta_wgrad.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                c_int64, c_void_p]
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise GtaError(f"{what} failed ({rc}): {lib().gta_block_strerror(rc).decode()}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return DTYPE_F32
    if dt == torch.bfloat16:
        return DTYPE_BF16
    raise GtaError(f"dtype {dt}: the block kernels take float32 or bfloat16")


def _stream(t: torch.Tensor):
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need_cuda(*ts):
    """Every tensor on ONE GPU, and that GPU the current device (the kernels are launched on its current stream)."""
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise GtaError("gta_amd block kernels need tensors on an MI355X (there is no CPU path)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise GtaError(f"operands on different devices ({dev} and {t.device})")
    if dev is not None and dev.index != torch.cuda.current_device():
        raise GtaError(f"operands on {dev} while the current device is cuda:{torch.cuda.current_device()}: "
                       f"call under torch.cuda.device({dev.index})")


def _need_vec(name: str, t: Optional[torch.Tensor], n: int, dtype=torch.float32):
    """t is a contiguous [n] vector of ``dtype`` (the kernels index it by the row length alone)."""
    if t is not None and (t.dtype != dtype or t.numel() != n or not t.is_contiguous()):
        raise GtaError(f"{name}: expected a contiguous {dtype} vector of {n} elements, got {tuple(t.shape)} {t.dtype}")


def _need_like(name: str, t: Optional[torch.Tensor], ref: torch.Tensor, dtype=None):
    """t is contiguous, shaped like ref (and of ``dtype`` when given)."""
    if t is not None and (t.shape != ref.shape or not t.is_contiguous() or (dtype is not None and t.dtype != dtype)):
        raise GtaError(f"{name}: expected a contiguous tensor of shape {tuple(ref.shape)}"
                       f"{'' if dtype is None else ' ' + str(dtype)}, got {tuple(t.shape)} {t.dtype}")


_ws_cache: dict = {}


def gemm_workspace(device) -> torch.Tensor:
    """One hipBLASLt workspace per (device, stream): calls on one stream are ordered, calls on different streams may
    overlap and must not share scratch memory."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None:
        ws = torch.empty(lib().gta_gemm_workspace_bytes(), device=device, dtype=torch.uint8)
        _ws_cache[key] = ws
    return ws


def ln_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, out_dtype: torch.dtype, want_stats=True):
    """x [..., d] contiguous -> (y [..., d] out_dtype, mean [rows] or None, rstd [rows] or None)."""
    _need_cuda(x, gamma, beta)
    d = x.shape[-1]
    rows = x.numel() // d
    _need_like("ln_fwd: x", x, x)
    _need_vec("ln_fwd: gamma", gamma, d)
    _need_vec("ln_fwd: beta", beta, d)
    y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32) if want_stats else None
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32) if want_stats else None
    check(lib().gta_ln_fwd(_ptr(x), dtype_code(x.dtype), _ptr(gamma), _ptr(beta), float(eps), rows, d, _ptr(y),
                           dtype_code(out_dtype), _ptr(mean), _ptr(rstd), _stream(x)), "gta_ln_fwd")
    return y, mean, rstd


def ln_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor,
           dres: Optional[torch.Tensor], bf16_copy: bool = False):
    """-> (dx like x [= dres + LayerNorm backward], dgamma [d] fp32, dbeta [d] fp32); with ``bf16_copy`` (fp32 x) the
    kernel also writes dx in bf16, returned as ``dx._gta_bf16`` for the block upstream (gta_amd.fused)."""
    _need_cuda(dy, x, gamma, mean, rstd, dres)
    d = x.shape[-1]
    rows = x.numel() // d
    _need_like("ln_bwd: x", x, x)
    _need_like("ln_bwd: dy", dy, x)
    _need_like("ln_bwd: dres", dres, x, x.dtype)
    _need_vec("ln_bwd: gamma", gamma, d)
    _need_vec("ln_bwd: mean", mean, rows)
    _need_vec("ln_bwd: rstd", rstd, rows)
    dx = torch.empty_like(x)
    dxb = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16) if (bf16_copy and x.dtype == torch.float32) else None
    dgamma = torch.empty(d, device=x.device, dtype=torch.float32)
    dbeta = torch.empty(d, device=x.device, dtype=torch.float32)
    nbytes = lib().gta_ln_bwd_workspace_bytes(rows, d)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    check(lib().gta_ln_bwd(_ptr(dy), dtype_code(dy.dtype), _ptr(x), dtype_code(x.dtype), _ptr(gamma), _ptr(mean), _ptr(rstd),
                           rows, d, _ptr(dres), _ptr(dx), dtype_code(dx.dtype), _ptr(dxb), _ptr(dgamma), _ptr(dbeta), _ptr(ws),
                           nbytes, _stream(x)), "gta_ln_bwd")
    if dxb is not None:
        dx._gta_bf16 = (dxb, dx._version)     # the version pins the copy to this content (an in-place accumulate bumps it)
    return dx, dgamma, dbeta


def gelu_fwd(x: torch.Tensor, p: float = 0.0, seed: int = 0) -> torch.Tensor:
    """gelu(x), followed by dropout(p) with the mask of ``seed`` when p > 0 (x contiguous)."""
    _need_cuda(x)
    _need_like("gelu_fwd: x", x, x)
    y = torch.empty_like(x)
    check(lib().gta_gelu_fwd(_ptr(x), _ptr(y), dtype_code(x.dtype), x.numel(), float(p), int(seed), _stream(x)), "gta_gelu_fwd")
    return y


def gelu_bwd(dy: torch.Tensor, x: torch.Tensor, p: float = 0.0, seed: int = 0) -> torch.Tensor:
    _need_cuda(x, dy)
    _need_like("gelu_bwd: x", x, x)
    _need_like("gelu_bwd: dy", dy, x, x.dtype)
    dx = torch.empty_like(x)
    check(lib().gta_gelu_bwd(_ptr(dy), _ptr(x), _ptr(dx), dtype_code(x.dtype), x.numel(), float(p), int(seed), _stream(x)),
          "gta_gelu_bwd")
    return dx


def dropout_add(z: torch.Tensor, skip: torch.Tensor, p: float, seed: int) -> torch.Tensor:
    """skip + dropout_p(z) (mask of ``seed``), in skip's dtype; z, skip contiguous and equally shaped."""
    _need_cuda(z, skip)
    if z.shape != skip.shape or not z.is_contiguous() or not skip.is_contiguous():
        raise GtaError("dropout_add: contiguous operands of one shape")
    out = torch.empty_like(skip)
    check(lib().gta_dropout_add(_ptr(z), dtype_code(z.dtype), _ptr(skip), _ptr(out), dtype_code(skip.dtype), z.numel(), float(p),
                                int(seed), _stream(z)), "gta_dropout_add")
    return out


def dropout_bwd(dout: torch.Tensor, out_dtype: torch.dtype, p: float, seed: int) -> torch.Tensor:
    """keep * dout / (1 - p) in ``out_dtype`` (the same mask as the forward call with this seed)."""
    _need_cuda(dout)
    if not dout.is_contiguous():
        raise GtaError("dropout_bwd: contiguous gradient")
    dz = torch.empty(dout.shape, device=dout.device, dtype=out_dtype)
    check(lib().gta_dropout_bwd(_ptr(dout), dtype_code(dout.dtype), _ptr(dz), dtype_code(out_dtype), dout.numel(), float(p), int(seed),
                                _stream(dout)), "gta_dropout_bwd")
    return dz


def colsum(a: torch.Tensor) -> torch.Tensor:
    """a [m, n] (row stride a.stride(0)) -> fp32 [n]."""
    _need_cuda(a)
    if a.dim() != 2 or a.stride(1) != 1:
        raise GtaError("colsum: a 2-D operand with unit column stride")
    m, n = a.shape
    out = torch.empty(n, device=a.device, dtype=torch.float32)
    nbytes = lib().gta_colsum_workspace_bytes(m, n)
    ws = torch.empty(nbytes, device=a.device, dtype=torch.uint8)
    check(lib().gta_colsum(_ptr(a), dtype_code(a.dtype), m, n, a.stride(0), _ptr(out), _ptr(ws), nbytes, _stream(a)), "gta_colsum")
    return out


def gemm(a: torch.Tensor, b: torch.Tensor, *, trans_a=False, trans_b=False, out_dtype=None, epilogue=EPI_NONE,
         bias: Optional[torch.Tensor] = None, aux: Optional[torch.Tensor] = None, c: Optional[torch.Tensor] = None,
         beta: float = 0.0, alpha: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """D[m,n] = epilogue(alpha * op_a(a) @ op_b(b) + beta * c); a, b 2-D with unit column stride (see gta_block.h)."""
    _need_cuda(a, b, c, bias, aux)
    if a.dim() != 2 or b.dim() != 2 or a.stride(1) != 1 or b.stride(1) != 1:
        raise GtaError("gemm: 2-D operands with unit column stride")
    m, k = (a.shape[1], a.shape[0]) if trans_a else a.shape
    kb, n = (b.shape[1], b.shape[0]) if trans_b else b.shape
    if kb != k:
        raise GtaError(f"gemm: inner dimensions {k} vs {kb}")
    out_dtype = out_dtype or (c.dtype if c is not None else a.dtype)
    if out is None:
        out = torch.empty(m, n, device=a.device, dtype=out_dtype)
    _need_cuda(out)
    for name, t in (("out", out), ("c", c), ("aux", aux)):
        if t is not None and (t.shape != (m, n) or t.stride(1) != 1):
            raise GtaError(f"gemm: {name} must be [{m}, {n}] with unit column stride, got {tuple(t.shape)}")
    if bias is not None and (bias.numel() != (m if epilogue == EPI_BGRAD_A else n) or not bias.is_contiguous()):
        raise GtaError(f"gemm: bias of {bias.numel()} elements for an [{m}, {n}] result")
    desc = GtaGemmDesc()
    desc.abi_version = BLOCK_ABI_VERSION
    desc.epilogue = epilogue
    desc.m, desc.n, desc.k = m, n, k
    desc.trans_a, desc.trans_b = int(trans_a), int(trans_b)
    desc.a_dtype, desc.b_dtype, desc.d_dtype = dtype_code(a.dtype), dtype_code(b.dtype), dtype_code(out.dtype)
    desc.bias_dtype = dtype_code(bias.dtype) if bias is not None else 0
    desc.aux_dtype = dtype_code(aux.dtype) if aux is not None else 0
    desc.lda, desc.ldb, desc.ldd = a.stride(0), b.stride(0), out.stride(0)
    desc.ldc = c.stride(0) if c is not None else 0
    desc.ldaux = aux.stride(0) if aux is not None else 0
    desc.alpha, desc.beta = alpha, beta
    if c is not None and c.dtype != out.dtype:
        raise GtaError("gemm: C and D share one dtype")
    ws = gemm_workspace(a.device)
    check(lib().gta_gemm(ctypes.byref(desc), _ptr(a), _ptr(b), _ptr(c), _ptr(out), _ptr(bias), _ptr(aux), _ptr(ws), ws.numel(),
                         _stream(a)), f"gta_gemm(m={m}, n={n}, k={k}, epilogue={epilogue})")
    return out


def wgrad_supported(g: torch.Tensor, x: torch.Tensor) -> bool:
    return (g.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and g.dim() == 2 and x.dim() == 2
            and g.stride(1) == 1 and x.stride(1) == 1 and g.stride(0) % 8 == 0 and x.stride(0) % 8 == 0
            and bool(lib().gta_wgrad_supported(g.shape[0], g.shape[1], x.shape[1])))


def wgrad(g: torch.Tensor, x: torch.Tensor, want_bias: bool):
    """g [m,n] bf16 (d out), x [m,k] bf16 (layer input) -> (dW [n,k] fp32 = g^T x, db [n] fp32 = column sums of g or None)."""
    _need_cuda(g, x)
    if not wgrad_supported(g, x) or x.shape[0] != g.shape[0]:
        raise GtaError(f"wgrad: unsupported operands {tuple(g.shape)} {g.dtype} / {tuple(x.shape)} {x.dtype} (gta_block.h)")
    m, n = g.shape
    k = x.shape[1]
    dw = torch.empty(n, k, device=g.device, dtype=torch.float32)
    db = torch.empty(n, device=g.device, dtype=torch.float32) if want_bias else None
    nbytes = lib().gta_wgrad_workspace_bytes(m, n, k)
    ws = torch.empty(max(nbytes, 16), device=g.device, dtype=torch.uint8)
    check(lib().gta_wgrad(_ptr(g), g.stride(0), _ptr(x), x.stride(0), m, n, k, _ptr(dw), _ptr(db), _ptr(ws), nbytes, _stream(g)),
          f"gta_wgrad(m={m}, n={n}, k={k})")
    return dw, db
