"""Backward of the fused GTA attention (binds gta_attn_bwd of the C ABI)."""
from . import native


def attn_bwd(cfg, q, k, v, out, dout, lse, tc, ta, vrep_q, vrep_k, cs_q, cs_k):
    raise native.GtaError("gta_attn_bwd: the backward kernels are not in this build yet")
