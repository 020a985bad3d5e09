"""ctypes binding of libgta_hip.so (the C ABI declared in include/gta_hip.h).

There is no CPU or eager fallback anywhere in this package: if the shared library is missing
or a kernel refuses a request, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int32, c_int64, c_uint32, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GTA_HIP_LIB: developer override (instrumented -DGTA_ABLATE builds of the same library)
LIB_PATH = os.environ.get("GTA_HIP_LIB") or os.path.join(_HERE, "csrc", "libgta_hip.so")

GTA_ABI_VERSION = 2
DTYPE_F32, DTYPE_BF16 = 0, 1
FLAG_V_TRANSFORM = 1 << 0
FLAG_EUCLID = 1 << 1
FLAG_PRETRANSFORMED = 1 << 2
FLAG_FUSED_KV = 1 << 3
FLAG_KV_READY = 1 << 4
FLAG_PREP_ONLY = 1 << 5
FLAG_NO_DMA = 1 << 8
FLAG_PERSIST = 1 << 9
FLAG_FP32_PRODUCTS = 1 << 10
FLAG_ROWS32 = 1 << 11
FLAG_FWD2_GENERIC = 1 << 6      # (r06) keep gta_fwd2_kernel where the dh = 64 bf16 instance gta_fwdc_kernel would run
FLAG_ITEM_CXX = 1 << 12
FLAG_BWD_KEYS32 = 1 << 13
FLAG_BWD_KEYS64 = 1 << 14
FLAG_BWD_SPLIT = 1 << 15
VREP_STRIDE = 72
VREP_INV, VREP_REP, VREP_D1, VREP_D2 = 0, 16, 32, 41
MAX_VIEWS = 16

# every symbol include/gta_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = (
    "gta_build_view_reps", "gta_build_so2_table", "gta_build_reps", "gta_rep_apply_bwd", "gta_attn_fwd", "gta_attn_fwd_supported",
    "gta_attn_fwd_launch_info", "gta_attn_fwd_workspace_bytes", "gta_attn_bwd", "gta_attn_bwd_workspace_bytes",
    "gta_rep_apply", "gta_attn_fwd_plain", "gta_attn_bwd_plain_f32", "gta_attn_bwd_plain_f32_workspace_bytes",
    "gta_strerror", "gta_abi_version", "gta_sizeof_attn_desc",
    "gta_debug_time_next_attention_kernel", "gta_debug_event_create", "gta_debug_event_destroy", "gta_debug_event_elapsed_ms",
    "gta_debug_profile_next_attention_kernel", "gta_debug_attention_kernel",
)


class GtaError(RuntimeError):
    pass


class GtaAttnDesc(ctypes.Structure):
    _fields_ = [
        ("abi_version", c_int32), ("dtype", c_int32), ("B", c_int32), ("H", c_int32),
        ("Tq", c_int32), ("Tk", c_int32), ("Nq", c_int32), ("Nk", c_int32), ("dh", c_int32),
        ("d_triv", c_int32), ("d_se3", c_int32), ("d_so3", c_int32), ("d_so2", c_int32),
        ("d_t2", c_int32), ("so3_degree", c_int32), ("flags", c_uint32), ("scale", c_float),
        ("_pad", c_int32),
        ("q_stride", c_int64 * 3), ("k_stride", c_int64 * 3), ("v_stride", c_int64 * 3),
        ("o_stride", c_int64 * 3),
    ]


_lib = None


def lib():
    """The loaded library; raises GtaError with the build command when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GtaError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C gta_amd/csrc` (hipcc --offload-arch=gfx950). gta_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        L.gta_strerror.restype = ctypes.c_char_p
        L.gta_strerror.argtypes = [ctypes.c_int]
        L.gta_abi_version.restype = ctypes.c_int
        L.gta_sizeof_attn_desc.restype = ctypes.c_int
        if L.gta_abi_version() != GTA_ABI_VERSION:
            raise GtaError("libgta_hip.so ABI version mismatch")
        if L.gta_sizeof_attn_desc() != ctypes.sizeof(GtaAttnDesc):
            raise GtaError("GtaAttnDesc layout mismatch between gta_hip.h and gta_amd/native.py")
        L.gta_build_view_reps.argtypes = [c_void_p, c_int32, c_int32, c_void_p, c_void_p]
        L.gta_build_so2_table.argtypes = [c_void_p, c_int32, c_int32, c_float, c_float, c_int32, c_void_p, c_void_p]
        L.gta_rep_apply_bwd.argtypes = [ctypes.POINTER(GtaAttnDesc), c_int32, c_void_p, ctypes.POINTER(c_int64), c_void_p,
                                        ctypes.POINTER(c_int64), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                        c_int64, c_void_p, ctypes.POINTER(c_int64), c_void_p, c_void_p]
        L.gta_build_reps.argtypes = [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_float, c_float,
                                     c_int32, c_void_p, c_void_p]
        L.gta_attn_fwd.argtypes = [ctypes.POINTER(GtaAttnDesc)] + [c_void_p] * 12 + [c_int64, c_void_p]
        L.gta_attn_fwd_workspace_bytes.argtypes = [ctypes.POINTER(GtaAttnDesc)]
        L.gta_attn_fwd_workspace_bytes.restype = c_int64
        L.gta_attn_fwd_supported.argtypes = [ctypes.POINTER(GtaAttnDesc)]
        L.gta_attn_fwd_launch_info.argtypes = [ctypes.POINTER(GtaAttnDesc)] + [ctypes.POINTER(c_int32)] * 3
        L.gta_attn_bwd.argtypes = ([ctypes.POINTER(GtaAttnDesc)] + [c_void_p] * 16
                                   + [ctypes.POINTER(c_int64), ctypes.POINTER(c_int64), c_void_p, c_void_p, c_void_p, c_int64, c_void_p])
        L.gta_attn_bwd_workspace_bytes.argtypes = [ctypes.POINTER(GtaAttnDesc)]
        L.gta_attn_bwd_workspace_bytes.restype = c_int64
        L.gta_rep_apply.argtypes = [ctypes.POINTER(GtaAttnDesc), c_int32, c_void_p, ctypes.POINTER(c_int64), c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int64), c_void_p, c_float, c_int64, c_void_p]
        L.gta_attn_fwd_plain.argtypes = [ctypes.POINTER(GtaAttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                         c_void_p, c_void_p, c_void_p, c_void_p]
        L.gta_attn_bwd_plain_f32_workspace_bytes.argtypes = [ctypes.POINTER(GtaAttnDesc)]
        L.gta_attn_bwd_plain_f32_workspace_bytes.restype = c_int64
        L.gta_attn_bwd_plain_f32.argtypes = ([ctypes.POINTER(GtaAttnDesc)] + [c_void_p] * 5 + [ctypes.POINTER(c_int64), c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int64), c_void_p, c_int64, c_void_p])
        L.gta_debug_time_next_attention_kernel.argtypes = [c_void_p, c_void_p]
        L.gta_debug_time_next_attention_kernel.restype = None
        L.gta_debug_event_create.restype = c_void_p
        L.gta_debug_event_destroy.argtypes = [c_void_p]
        L.gta_debug_event_destroy.restype = None
        L.gta_debug_event_elapsed_ms.argtypes = [c_void_p, c_void_p]
        L.gta_debug_event_elapsed_ms.restype = c_float
        L.gta_debug_profile_next_attention_kernel.argtypes = [c_void_p, c_int64]
        L.gta_debug_profile_next_attention_kernel.restype = None
        L.gta_debug_attention_kernel.argtypes = [ctypes.POINTER(GtaAttnDesc), ctypes.POINTER(c_int32), ctypes.POINTER(c_int32)]
        L.gta_debug_attention_kernel.restype = ctypes.c_char_p
        _lib = L
    return _lib


def attention_kernel(desc):
    """(kernel name, work items, query rows per item) of the attention kernel gta_attn_fwd launches for desc when it is given a
    workspace (include/gta_hip.h: gta_debug_attention_kernel)"""
    n, rows = c_int32(0), c_int32(0)
    name = lib().gta_debug_attention_kernel(ctypes.byref(desc), ctypes.byref(n), ctypes.byref(rows)) or b""
    return name.decode(), n.value, rows.value


def check(rc: int, what: str):
    if rc != 0:
        raise GtaError(f"{what} failed ({rc}): {lib().gta_strerror(rc).decode()}")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(*ts):
    """Every tensor on a GPU, and that GPU the current device: the kernels are launched on ITS current stream."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise GtaError("gta_amd kernels need CUDA/HIP tensors (MI355X); there is no CPU path")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise GtaError(f"operand on {t.device} while the current device is cuda:{cur}: call under torch.cuda.device({t.device.index})")


def check_table(name: str, t: Optional[torch.Tensor], shape, device, allow_none: bool = False):
    """The kernels index the rep tables from the descriptor alone (vrep[b*N+n], cs[(b*T+t)*d_so2], coord[(b*T+t)*2]):
    a table of another batch / token count / frequency count, or one that lives on the host, must raise here like the
    reference's reshape / einsum would -- not read out of bounds on the device."""
    if t is None:
        if allow_none:
            return
        raise GtaError(f"{name} is required for this f_dims layout")
    if not t.is_cuda or t.device != device:
        raise GtaError(f"{name} must live on {device} (got {t.device})")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise GtaError(f"{name} must be a contiguous float32 tensor (got {t.dtype}, contiguous={t.is_contiguous()})")
    if tuple(t.shape) != tuple(shape):
        raise GtaError(f"{name} has shape {tuple(t.shape)}, the kernels expect {tuple(shape)} for this q/k/f_dims")


def check_scalar(name: str, t: Optional[torch.Tensor], device):
    if t is None:
        return
    if not t.is_cuda or t.device != device or t.dtype != torch.float32 or t.numel() < 1:
        raise GtaError(f"{name} must be a float32 tensor on {device} (it is read through a device pointer)")


def build_view_reps(transforms: torch.Tensor, so3_degree: int) -> torch.Tensor:
    """extrinsics [B,N,4,4] -> packed per-view reps [B,N,VREP_STRIDE] (see gta_hip.h)."""
    _require_cuda(transforms)
    B, N = transforms.shape[:2]
    E = transforms.detach().to(torch.float32).contiguous()
    out = torch.empty(B, N, VREP_STRIDE, device=E.device, dtype=torch.float32)
    check(lib().gta_build_view_reps(_ptr(E), B * N, int(so3_degree), _ptr(out), _stream()),
          "gta_build_view_reps")
    return out


def build_so2_table(coord: torch.Tensor, nfreqs: int, max_freq_h: float, max_freq_w: float,
                    shared_freqs: bool = False) -> torch.Tensor:
    """coord [B,T,2] -> (cos, sin) table [B,T,2*nfreqs,2]."""
    _require_cuda(coord)
    B, T = coord.shape[:2]
    c = coord.detach().to(torch.float32).contiguous()
    out = torch.empty(B, T, 2 * nfreqs, 2, device=c.device, dtype=torch.float32)
    check(lib().gta_build_so2_table(_ptr(c), B * T, int(nfreqs), float(max_freq_h), float(max_freq_w),
                                    int(bool(shared_freqs)), _ptr(out), _stream()), "gta_build_so2_table")
    return out


def build_reps(transforms: torch.Tensor, so3_degree: int, coord: torch.Tensor, nfreqs: int, max_freq_h: float,
               max_freq_w: float, shared_freqs: bool = False):
    """build_view_reps + build_so2_table in one launch -> (vrep [B,N,72], cs [B,T,2F,2])."""
    _require_cuda(transforms, coord)
    B, N = transforms.shape[:2]
    T = coord.shape[1]
    E = transforms.detach().to(torch.float32).contiguous()
    c = coord.detach().to(torch.float32).contiguous()
    vrep = torch.empty(B, N, VREP_STRIDE, device=E.device, dtype=torch.float32)
    cs = torch.empty(B, T, 2 * nfreqs, 2, device=c.device, dtype=torch.float32)
    check(lib().gta_build_reps(_ptr(E), B * N, int(so3_degree), _ptr(vrep), _ptr(c), B * T, int(nfreqs),
                               float(max_freq_h), float(max_freq_w), int(bool(shared_freqs)), _ptr(cs), _stream()),
          "gta_build_reps")
    return vrep, cs


def make_desc(q, k, v, out, f_dims: dict, so3_degree: int, Nq: int, Nk: int, scale: float,
              flags: int) -> GtaAttnDesc:
    """q/k/v/out are [B,H,T,dh] tensors (any strides with unit channel stride)."""
    d = GtaAttnDesc()
    d.abi_version = GTA_ABI_VERSION
    d.dtype = DTYPE_BF16 if q.dtype == torch.bfloat16 else DTYPE_F32
    d.B, d.H, d.Tq, d.dh = q.shape
    d.Tk = k.shape[2]
    d.Nq, d.Nk = Nq, Nk
    d.d_triv = int(f_dims.get("triv", 0)); d.d_se3 = int(f_dims.get("se3", 0))
    d.d_so3 = int(f_dims.get("so3", 0)); d.d_so2 = int(f_dims.get("so2", 0)); d.d_t2 = int(f_dims.get("t2", 0))
    d.so3_degree = int(so3_degree)
    d.flags = flags
    d.scale = float(scale)
    for name, t in (("q_stride", q), ("k_stride", k), ("v_stride", v), ("o_stride", out)):
        if t.stride(3) != 1:
            raise GtaError(f"{name}: channel stride must be 1")
        getattr(d, name)[:] = [t.stride(0), t.stride(1), t.stride(2)]
    return d


def attn_fwd(desc: GtaAttnDesc, q, k, v, vrep_q, vrep_k, cs_q, cs_k, trans_coeff, tau, out, lse,
             workspace: Optional[torch.Tensor] = None):
    """workspace: uint8 CUDA tensor of >= attn_fwd_workspace_bytes(desc) bytes selects the two-stage
    plan (K/V pre-pass + lean attention kernel); None selects the single fused kernel."""
    _require_cuda(q, k, v, out, workspace)
    check(lib().gta_attn_fwd(ctypes.byref(desc), _ptr(q), _ptr(k), _ptr(v), _ptr(vrep_q), _ptr(vrep_k),
                             _ptr(cs_q), _ptr(cs_k), _ptr(trans_coeff), _ptr(tau), _ptr(out), _ptr(lse),
                             _ptr(workspace), 0 if workspace is None else workspace.numel(), _stream()),
          "gta_attn_fwd")


def attn_fwd_workspace_bytes(desc: GtaAttnDesc) -> int:
    return int(lib().gta_attn_fwd_workspace_bytes(ctypes.byref(desc)))


def attn_fwd_supported(desc: GtaAttnDesc) -> int:
    return lib().gta_attn_fwd_supported(ctypes.byref(desc))


def launch_info(desc: GtaAttnDesc):
    a, b, c = c_int32(), c_int32(), c_int32()
    check(lib().gta_attn_fwd_launch_info(ctypes.byref(desc), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)),
          "gta_attn_fwd_launch_info")
    return {"lds_bytes": a.value, "workgroups": b.value, "threads": c.value}


def attn_bwd_workspace_bytes(desc: GtaAttnDesc) -> int:
    return int(lib().gta_attn_bwd_workspace_bytes(ctypes.byref(desc)))


def attn_bwd(desc: GtaAttnDesc, q, k, v, out, dout, lse, vrep_q, vrep_k, cs_q, cs_k, trans_coeff, tau, kv_images,
             dq, dk, dv, dtrans_coeff, workspace, dtau=None):
    """All tensors [B,H,T,dh] views (unit channel stride); dq/dk/dv/dout strides are passed explicitly."""
    _require_cuda(q, k, v, out, dout, dq, dk, dv, workspace)
    gs = (c_int64 * 9)(*(list(dq.stride()[:3]) + list(dk.stride()[:3]) + list(dv.stride()[:3])))
    ds = (c_int64 * 3)(*dout.stride()[:3])
    check(lib().gta_attn_bwd(ctypes.byref(desc), _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(dout), _ptr(lse),
                             _ptr(vrep_q), _ptr(vrep_k), _ptr(cs_q), _ptr(cs_k), _ptr(trans_coeff), _ptr(tau),
                             _ptr(kv_images), _ptr(dq), _ptr(dk), _ptr(dv), gs, ds, _ptr(dtrans_coeff), _ptr(dtau),
                             _ptr(workspace), workspace.numel(), _stream()), "gta_attn_bwd")


def attn_bwd_plain_f32(desc: GtaAttnDesc, q, k, v, out, dout, lse, tau, dq, dk, dv):
    """exact-fp32 backward of plain attention on pre-transformed float32 tensors (include/gta_hip.h: gta_attn_bwd_plain_f32)"""
    _require_cuda(q, k, v, out, dout, dq, dk, dv)
    gs = (c_int64 * 9)(*(list(dq.stride()[:3]) + list(dk.stride()[:3]) + list(dv.stride()[:3])))
    ds = (c_int64 * 3)(*dout.stride()[:3])
    ws = torch.empty(int(lib().gta_attn_bwd_plain_f32_workspace_bytes(ctypes.byref(desc))), device=q.device, dtype=torch.uint8)
    check(lib().gta_attn_bwd_plain_f32(ctypes.byref(desc), _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(dout), ds, _ptr(lse), _ptr(tau),
                                       _ptr(dq), _ptr(dk), _ptr(dv), gs, _ptr(ws), ws.numel(), _stream()), "gta_attn_bwd_plain_f32")


def rep_apply(desc: GtaAttnDesc, mode: int, x, vrep, cs, coord, trans_coeff, y, key_bias=None, bias_scale=0.0):
    """Generic rho application (any layout / t2 / euclid): x, y are [B,H,T,dh] views."""
    _require_cuda(x, y)
    xs = (c_int64 * 3)(*x.stride()[:3])
    ys = (c_int64 * 3)(*y.stride()[:3])
    check(lib().gta_rep_apply(ctypes.byref(desc), int(mode), _ptr(x), xs, _ptr(vrep), _ptr(cs), _ptr(coord),
                              _ptr(trans_coeff), _ptr(y), ys, _ptr(key_bias), float(bias_scale),
                              0 if key_bias is None else key_bias.shape[-1], _stream()), "gta_rep_apply")


def rep_apply_bwd(desc: GtaAttnDesc, mode: int, x, dy, vrep, cs, coord, trans_coeff, dx, dtc_rows=None, dkey_bias=None,
                  bias_scale=0.0):
    """Adjoint of rep_apply(mode): dx = M^T dy; dtc_rows [B,H,T] receives per-row d trans_coeff terms."""
    _require_cuda(x, dy, dx)
    xs = (c_int64 * 3)(*x.stride()[:3])
    ys = (c_int64 * 3)(*dy.stride()[:3])
    ds = (c_int64 * 3)(*dx.stride()[:3])
    check(lib().gta_rep_apply_bwd(ctypes.byref(desc), int(mode), _ptr(x), xs, _ptr(dy), ys, _ptr(vrep), _ptr(cs),
                                  _ptr(coord), _ptr(trans_coeff), _ptr(dkey_bias), float(bias_scale),
                                  0 if dkey_bias is None else dkey_bias.shape[-1], _ptr(dx), ds, _ptr(dtc_rows),
                                  _stream()), "gta_rep_apply_bwd")


def attn_fwd_plain(desc: GtaAttnDesc, q, k, v, key_bias, tau, out, lse):
    _require_cuda(q, k, v, out)
    check(lib().gta_attn_fwd_plain(ctypes.byref(desc), _ptr(q), _ptr(k), _ptr(v), _ptr(key_bias),
                                   0 if key_bias is None else key_bias.shape[-1], _ptr(tau), _ptr(out), _ptr(lse),
                                   _stream()), "gta_attn_fwd_plain")
