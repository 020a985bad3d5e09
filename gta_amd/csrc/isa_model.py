#!/usr/bin/env python3
"""isa_model.py -- a small gfx950 assembler DSL and a FUNCTIONAL one-wave simulator for the generated instruction streams of
gta_attn64 (gen_item64.py: an item's prologue and epilogue around the tile loop of gen_attn64.py).

Test infrastructure of the build, not product code: the kernel includes only the TEXT the generators emit.  What this file gives
the generators on the CPU (run by tests/test_host_logic.py on every build):

  * `Asm`: one method per instruction form the streams use.  A call records the assembly text AND an executable description of
    the same instruction (mnemonic + operands), so the text that hipcc assembles and the semantics that are simulated cannot
    drift apart; `rd` / `wr` register sets feed the wait-state checker of gen_attn64.py.
  * `Wave`: 64 lanes x (256 VGPRs + 256 AGPRs), SGPRs, VCC / EXEC / SCC / M0, a byte-addressed LDS and a flat global memory made of
    registered numpy buffers.  Arithmetic is bit-level (fp32 via numpy, fused multiply-add via float64, bf16 round-to-nearest-even,
    v_mfma_f32_32x32x16_bf16 with the lane layouts of cdna_hip_programming.md section 3: A / B lane (i, kh) holds row / column i,
    k = 8 kh .. 8 kh + 7; D register r of lane (j, h) is row (r & 3) + 8 (r >> 2) + 4 h, column j).
  * memory counters as the hardware keeps them: LDS operations complete in order under lgkmcnt, vector memory loads and stores in
    order under vmcnt (scalar loads: out of order, any use needs lgkmcnt(0)).  A load's destination is POISONED from issue until a
    wait covers it -- reading or overwriting it earlier is an error -- so a missing or mis-counted s_waitcnt fails the simulation
    instead of passing by luck, across the item loop's back edge too.
"""
import re
import struct

import numpy as np

from gen_attn64 import CheckError, Ins

LANES = 64


class XI(Ins):
    """an instruction with an executable description: fx = (mnemonic, operand, ...)"""
    __slots__ = ("fx",)

    def __init__(self, text, kind, rd=(), wr=(), fx=None, sem=None, label=None, target=None):
        super().__init__(text, kind, rd, wr, sem, label, target)
        self.fx = fx


# ------------------------------------------------------------------------------------------------------------------
# operands
# ------------------------------------------------------------------------------------------------------------------
def regs(prefix, first, n):
    return [f"{prefix}{first + i}" for i in range(n)]


def rtext(rl):
    """assembly text of a register list: consecutive registers of one file"""
    if isinstance(rl, str):
        return rl
    p, n0 = rl[0][0], int(rl[0][1:])
    assert all(r == f"{p}{n0 + i}" for i, r in enumerate(rl)), rl
    return f"{p}{n0}" if len(rl) == 1 else f"{p}[{n0}:{n0 + len(rl) - 1}]"


def is_reg(x):
    return isinstance(x, str) and re.fullmatch(r"[vas]\d+", x) is not None


def f2u(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def otext(x):
    """source operand text: register, named statement operand, inline constant or 32-bit literal"""
    if isinstance(x, str):
        return x
    if isinstance(x, float):
        if x in (0.0, 0.5, 1.0, 2.0, 4.0, -0.5, -1.0, -2.0, -4.0):
            return repr(x) if x != 0.0 else "0"
        return f"0x{f2u(x):08x}"
    if isinstance(x, int):
        return str(x) if -16 <= x <= 64 else f"0x{x & 0xffffffff:08x}"
    raise TypeError(x)


def rdset(*ops):
    out = []
    for o in ops:
        if isinstance(o, (list, tuple)):
            out += [r for r in o if isinstance(r, str)]
        elif isinstance(o, str) and (is_reg(o) or o in ("vcc", "exec", "scc", "m0")):
            out.append(o)
    return out


class Asm:
    """collects XI instructions; every emitter returns nothing and appends"""

    def __init__(self):
        self.out = []

    def add(self, text, kind, rd, wr, fx, **kw):
        self.out.append(XI(text, kind, rd, wr, fx, **kw))

    def raw(self, ins):
        self.out.append(ins)

    # ---- VALU ------------------------------------------------------------------------------------------------------
    def _v2(self, mn, d, a, b, kind="valu"):
        self.add(f"{mn} {d}, {otext(a)}, {otext(b)}", kind, rdset(a, b), [d], (mn, d, a, b))

    def v_mov_b32(self, d, a):
        self.add(f"v_mov_b32 {d}, {otext(a)}", "valu", rdset(a), [d], ("v_mov_b32", d, a))

    def v_mul_f32(self, d, a, b):
        self._v2("v_mul_f32", d, a, b)

    def v_add_f32(self, d, a, b):
        self._v2("v_add_f32", d, a, b)

    def v_sub_f32(self, d, a, b):
        self._v2("v_sub_f32", d, a, b)

    def v_max_f32(self, d, a, b):
        self._v2("v_max_f32", d, a, b)

    def v_fma_f32(self, d, a, b, c, neg=(False, False, False)):
        t = [("-" if n else "") + otext(x) for x, n in zip((a, b, c), neg)]
        self.add(f"v_fma_f32 {d}, {t[0]}, {t[1]}, {t[2]}", "valu", rdset(a, b, c), [d], ("v_fma_f32", d, a, b, c, tuple(neg)))

    def v_fmac_f32(self, d, a, b):
        """d += a * b (one rounding)"""
        self.add(f"v_fmac_f32 {d}, {otext(a)}, {otext(b)}", "valu", rdset(a, b, d), [d], ("v_fma_f32", d, a, b, d, (False, False, False)))

    def v_cvt_pk_bf16_f32(self, d, lo, hi):
        self.add(f"v_cvt_pk_bf16_f32 {d}, {otext(lo)}, {otext(hi)}", "valu", rdset(lo, hi), [d], ("v_cvt_pk_bf16_f32", d, lo, hi))

    def v_accvgpr_write_b32(self, a, v):
        self.add(f"v_accvgpr_write_b32 {a}, {otext(v)}", "valu", rdset(v), [a], ("v_mov_b32", a, v))

    def v_accvgpr_read_b32(self, v, a):
        self.add(f"v_accvgpr_read_b32 {v}, {a}", "valu", [a], [v], ("v_mov_b32", v, a))

    def _trans(self, mn, d, a):
        self.add(f"{mn} {d}, {otext(a)}", "trans", rdset(a), [d], (mn, d, a))

    def v_rcp_f32(self, d, a):
        self._trans("v_rcp_f32", d, a)

    def v_sqrt_f32(self, d, a):
        self._trans("v_sqrt_f32", d, a)

    def v_log_f32(self, d, a):
        self._trans("v_log_f32", d, a)

    def v_exp_f32(self, d, a):
        self._trans("v_exp_f32", d, a)

    def v_add_u32(self, d, a, b):
        self._v2("v_add_u32", d, a, b)

    def v_sub_u32(self, d, a, b):
        self._v2("v_sub_u32", d, a, b)

    def v_lshlrev_b32(self, d, sh, a):
        self._v2("v_lshlrev_b32", d, sh, a)

    def v_lshrrev_b32(self, d, sh, a):
        self._v2("v_lshrrev_b32", d, sh, a)

    def v_and_b32(self, d, a, b):
        self._v2("v_and_b32", d, a, b)

    def v_mul_u32_u24(self, d, a, b):
        self._v2("v_mul_u32_u24", d, a, b)

    def v_mul_lo_u32(self, d, a, b):
        self._v2("v_mul_lo_u32", d, a, b)

    def v_mad_u32_u24(self, d, a, b, c):
        self.add(f"v_mad_u32_u24 {d}, {otext(a)}, {otext(b)}, {otext(c)}", "valu", rdset(a, b, c), [d], ("v_mad_u32_u24", d, a, b, c))

    def v_mul_hi_u32(self, d, a, b):
        self._v2("v_mul_hi_u32", d, a, b)

    def v_mbcnt_lane(self, d):
        """d = lane id (v_mbcnt_lo + v_mbcnt_hi over a full mask)"""
        self.add(f"v_mbcnt_lo_u32_b32 {d}, -1, 0", "valu", [], [d], ("v_mbcnt_lo", d))
        self.add(f"v_mbcnt_hi_u32_b32 {d}, -1, {d}", "valu", [d], [d], ("v_mbcnt_hi", d))

    def v_readfirstlane_b32(self, s, v):
        self.add(f"v_readfirstlane_b32 {s}, {v}", "vread", [v], [s], ("v_readfirstlane_b32", s, v))

    def v_permlane32_swap_b32(self, a, b):
        self.add(f"v_permlane32_swap_b32 {a}, {b}", "perm", [a, b], [a, b], ("v_permlane32_swap_b32", a, b))

    def v_cmp(self, cond, ty, a, b):
        """vcc = a <cond> b   (cond: lt, le, eq, ge, gt, lg; ty: f32, u32, i32)"""
        self.add(f"v_cmp_{cond}_{ty} vcc, {otext(a)}, {otext(b)}", "valu", rdset(a, b), ["vcc"], ("v_cmp", cond, ty, a, b))

    def v_cndmask_b32(self, d, a, b):
        """d = vcc ? b : a"""
        self.add(f"v_cndmask_b32 {d}, {otext(a)}, {otext(b)}, vcc", "valu", rdset(a, b, "vcc"), [d], ("v_cndmask_b32", d, a, b))

    def v_div_scale_f32(self, d, sdst, a, b, c):
        """(hipcc's IEEE division sequence; modelled for operands in the normal range: no scaling, vcc = 0)"""
        sd = sdst if sdst == "vcc" else rtext(sdst)
        wr = [d] + (["vcc"] if sdst == "vcc" else list(sdst))
        self.add(f"v_div_scale_f32 {d}, {sd}, {otext(a)}, {otext(b)}, {otext(c)}", "valu", rdset(a, b, c), wr, ("v_div_scale_f32", d, sdst, a, b, c))

    def v_div_fmas_f32(self, d, a, b, c):
        self.add(f"v_div_fmas_f32 {d}, {otext(a)}, {otext(b)}, {otext(c)}", "valu", rdset(a, b, c, "vcc"), [d], ("v_div_fmas_f32", d, a, b, c))

    def v_div_fixup_f32(self, d, a, b, c):
        self.add(f"v_div_fixup_f32 {d}, {otext(a)}, {otext(b)}, {otext(c)}", "valu", rdset(a, b, c), [d], ("v_div_fixup_f32", d, a, b, c))

    def mfma(self, d16, a4, b4, c16):
        """v_mfma_f32_32x32x16_bf16 D, A, B, C   (C: a 16-register list or 0)"""
        ct = "0" if c16 == 0 else rtext(c16)
        rd = list(a4) + list(b4) + ([] if c16 == 0 else list(c16))
        self.add(f"v_mfma_f32_32x32x16_bf16 {rtext(d16)}, {rtext(a4)}, {rtext(b4)}, {ct}", "mfma", rd, list(d16),
                 ("mfma", list(d16), list(a4), list(b4), 0 if c16 == 0 else list(c16)))

    # ---- LDS ------------------------------------------------------------------------------------------------------------
    def ds_read(self, width, d, addr, off=0):
        n = {32: 1, 64: 2, 128: 4}[width]
        assert len(d) == n and 0 <= off < 65536
        self.add(f"ds_read_b{width} {rtext(d)}, {addr}" + (f" offset:{off}" if off else ""), "ds", [addr], list(d), ("ds_read", n, list(d), addr, off))

    def ds_read_tr(self, d2, addr, off=0):
        """ds_read_b64_tr_b16: within each group of 16 lanes, lane l receives element (l & 3) of the four 16-bit elements addressed by
        lanes 4 j + (l >> 2), j = 0..3 (cdna_hip_programming.md, LDS: column l of the 4 x 16 block the group's addresses describe)"""
        assert len(d2) == 2 and 0 <= off < 65536
        self.add(f"ds_read_b64_tr_b16 {rtext(d2)}, {addr}" + (f" offset:{off}" if off else ""), "ds", [addr], list(d2), ("ds_read_tr", list(d2), addr, off))

    def ds_write(self, width, addr, s, off=0):
        n = {32: 1, 64: 2, 128: 4}[width]
        assert len(s) == n and 0 <= off < 65536
        self.add(f"ds_write_b{width} {addr}, {rtext(s)}" + (f" offset:{off}" if off else ""), "dsw", [addr] + list(s), [], ("ds_write", n, addr, list(s), off))

    # ---- vector memory (saddr form: 64-bit SGPR base + 32-bit VGPR offset + 13-bit signed immediate) -----------------------
    def global_load(self, n, d, voff, sbase, off=0, aux=""):
        assert len(d) == n and -4096 <= off <= 4095 and len(sbase) == 2
        mn = {1: "global_load_dword", 2: "global_load_dwordx2", 4: "global_load_dwordx4"}[n]
        self.add(f"{mn} {rtext(d)}, {voff}, {rtext(sbase)}" + (f" offset:{off}" if off else "") + aux, "vmem", [voff] + list(sbase), list(d),
                 ("global_load", n, list(d), voff, list(sbase), off))

    def global_store(self, n, voff, s, sbase, off=0, aux=""):
        assert len(s) == n and -4096 <= off <= 4095 and len(sbase) == 2
        mn = {1: "global_store_dword", 2: "global_store_dwordx2", 4: "global_store_dwordx4"}[n]
        self.add(f"{mn} {voff}, {rtext(s)}, {rtext(sbase)}" + (f" offset:{off}" if off else "") + aux, "vmemst", [voff] + list(s) + list(sbase), [],
                 ("global_store", n, voff, list(s), list(sbase), off))

    # ---- SALU / control -------------------------------------------------------------------------------------------------------
    def _s(self, text, rd, wr, fx):
        self.add(text, "salu", rd, wr, fx)

    def s_mov_b32(self, d, a):
        self._s(f"s_mov_b32 {d}, {otext(a)}", rdset(a), [d], ("s_mov_b32", d, a))

    def s_mov_b64(self, d2, a2):
        """a2: a register pair, a named 64-bit statement operand, or a small constant"""
        at = rtext(a2) if isinstance(a2, (list, tuple)) else otext(a2)
        self._s(f"s_mov_b64 {rtext(d2)}, {at}", rdset(a2), list(d2), ("s_mov_b64", list(d2), a2))

    def s_add_u32(self, d, a, b):
        self._s(f"s_add_u32 {d}, {otext(a)}, {otext(b)}", rdset(a, b), [d, "scc"], ("s_add_u32", d, a, b))

    def s_addc_u32(self, d, a, b):
        self._s(f"s_addc_u32 {d}, {otext(a)}, {otext(b)}", rdset(a, b, "scc"), [d, "scc"], ("s_addc_u32", d, a, b))

    def s_sub_u32(self, d, a, b):
        self._s(f"s_sub_u32 {d}, {otext(a)}, {otext(b)}", rdset(a, b), [d, "scc"], ("s_sub_u32", d, a, b))

    def s_mul_i32(self, d, a, b):
        self._s(f"s_mul_i32 {d}, {otext(a)}, {otext(b)}", rdset(a, b), [d], ("s_mul_i32", d, a, b))

    def s_and_b32(self, d, a, b):
        self._s(f"s_and_b32 {d}, {otext(a)}, {otext(b)}", rdset(a, b), [d, "scc"], ("s_and_b32", d, a, b))

    def s_or_b32(self, d, a, b):
        self._s(f"s_or_b32 {d}, {otext(a)}, {otext(b)}", rdset(a, b), [d, "scc"], ("s_or_b32", d, a, b))

    def s_lshl_b32(self, d, a, b):
        self._s(f"s_lshl_b32 {d}, {otext(a)}, {otext(b)}", rdset(a, b), [d, "scc"], ("s_lshl_b32", d, a, b))

    def s_cmp(self, cond, ty, a, b):
        self._s(f"s_cmp_{cond}_{ty} {otext(a)}, {otext(b)}", rdset(a, b), ["scc"], ("s_cmp", cond, ty, a, b))

    def s_cmp_u64_ne0(self, a2):
        self._s(f"s_cmp_lg_u64 {rtext(a2)}, 0", list(a2), ["scc"], ("s_cmp_u64_ne0", list(a2)))

    def s_exec_set(self, a, hi_pair=None):
        """exec = a named mask: 'all', 'lo' (lanes 0..31), 'lane0', 'hi' (lanes 32..63: through an SGPR pair holding the mask -- a
        64-bit literal is not encodable)"""
        if a == "hi":
            self._s(f"s_mov_b64 exec, {rtext(hi_pair)}", list(hi_pair), ["exec"], ("s_exec_set", a))
        else:
            val = {"all": "-1", "lo": "0xffffffff", "lane0": "1"}[a]
            self._s(f"s_mov_b64 exec, {val}", [], ["exec"], ("s_exec_set", a))

    def s_memtime(self, d2):
        self.add(f"s_memtime {rtext(d2)}", "smem", [], list(d2), ("s_memtime", list(d2)))

    def s_memrealtime(self, d2):
        self.add(f"s_memrealtime {rtext(d2)}", "smem", [], list(d2), ("s_memtime", list(d2)))

    def label(self, name):
        self.out.append(XI(f"{name}:", "label", fx=("label", name), label=name))

    def branch(self, mn, target):
        self.out.append(XI(f"{mn} {target}", "branch", rdset("scc") if "scc" in mn else [], [], ("branch", mn, target), target=target))

    def waitcnt(self, vm=None, lgkm=None):
        parts = ([f"vmcnt({vm})"] if vm is not None else []) + ([f"lgkmcnt({lgkm})"] if lgkm is not None else [])
        self.out.append(XI("s_waitcnt " + " ".join(parts), "wait", fx=("waitcnt", vm, lgkm)))

    def nop(self, n):
        while n > 0:
            k = min(n, 16)
            self.out.append(XI(f"s_nop {k - 1}", "nop", fx=("nop", k), sem=("nop", k)))
            n -= k

    def barrier(self):
        self.out.append(XI("s_barrier", "barrier", fx=("barrier",)))

    def pseudo(self, *fx):
        """no text: a hook for the simulation (checkpoints, stubs)"""
        self.out.append(XI("", "pseudo", fx=("pseudo",) + tuple(fx)))


# ------------------------------------------------------------------------------------------------------------------
# bit-level helpers
# ------------------------------------------------------------------------------------------------------------------
def u2f(a):
    return a.view(np.float32)


def as_u32(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def bf16_rne(x):
    """fp32 array -> bf16 bits (uint32, low 16 used), round to nearest even (v_cvt_pk_bf16_f32; NaNs are not produced by the tests)"""
    u = as_u32(x).astype(np.uint64)
    r = (u + 0x7fff + ((u >> 16) & 1)) >> 16
    return (r & 0xffff).astype(np.uint32)


def bf16_to_f32(b):
    return (np.asarray(b, dtype=np.uint32) << 16).view(np.float32)


def fma32(a, b, c):
    with np.errstate(all="ignore"):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def mfma_32x32x16_bf16(A, B, C):
    """A, B: [4][64] uint32 (bf16 pairs), C: [16][64] float32 -> D [16][64] float32.
    A lane (i, kh): row i, k = 8 kh + 2 w + e (register w, half e); B the same with column j; D register r of lane (j, h): row
    (r & 3) + 8 (r >> 2) + 4 h, column j."""
    a = np.zeros((32, 16), np.float64)
    b = np.zeros((16, 32), np.float64)
    for lane in range(LANES):
        i, kh = lane & 31, lane >> 5
        for w in range(4):
            for e in range(2):
                k = 8 * kh + 2 * w + e
                a[i, k] = bf16_to_f32(np.uint32((int(A[w][lane]) >> (16 * e)) & 0xffff))
                b[k, i] = bf16_to_f32(np.uint32((int(B[w][lane]) >> (16 * e)) & 0xffff))
    d = a @ b
    D = np.zeros((16, LANES), np.float32)
    for lane in range(LANES):
        j, h = lane & 31, lane >> 5
        for r in range(16):
            row = (r & 3) + 8 * (r >> 2) + 4 * h
            D[r][lane] = np.float32(d[row, j] + np.float64(C[r][lane]))
    return D


# ------------------------------------------------------------------------------------------------------------------
# the wave
# ------------------------------------------------------------------------------------------------------------------
class Wave:
    def __init__(self, lds_bytes=160 * 1024, inputs=None, seed=0):
        rng = np.random.default_rng(seed)
        # registers start as junk that is NOT a plausible value: a stream that forgets to initialise something shows
        self.v = rng.integers(0x7f800001, 0x7fffffff, size=(512, LANES), dtype=np.uint32)      # v0..v255, a0..a255 (NaN patterns)
        self.s = {}
        self.vcc = np.zeros(LANES, bool)
        self.exec = np.ones(LANES, bool)
        self.scc = 0
        self.m0 = 0
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.bufs = []                    # (base, numpy uint8 array)
        self.inputs = dict(inputs or {})  # named statement operands: "%[x]" -> int (64-bit for pointers)
        self.lgkm = []                    # outstanding LDS ops / scalar loads, oldest first: ("ds" | "smem", [dest registers])
        self.vm = []                      # outstanding vector memory ops, oldest first: [dest registers] ([] for a store)
        self.poison = {}                  # register -> why it may not be touched
        self.time = 0
        self.hooks = {}

    # -- memory --
    def map_buffer(self, base, arr):
        assert arr.dtype == np.uint8 and arr.ndim == 1
        self.bufs.append((base, arr))

    def _g(self, addr, n):
        for base, arr in self.bufs:
            if base <= addr and addr + n <= base + len(arr):
                return arr, addr - base
        raise CheckError(f"global access of {n} B at 0x{addr:x} outside every buffer")

    # -- operands --
    def ridx(self, r):
        return int(r[1:]) + (256 if r[0] == "a" else 0)

    def touch(self, r, ins, write):
        if r in self.poison:
            raise CheckError(f"{r} {'overwritten' if write else 'read'} while {self.poison[r]} is outstanding   at: {ins.text}")

    def sval(self, x, ins=None):
        """scalar value (python int, 32 bits)"""
        if isinstance(x, str):
            if x.startswith("%["):
                return self.inputs[x] & 0xffffffff
            if x[0] == "s":
                if ins is not None:
                    self.touch(x, ins, False)
                if x not in self.s:
                    raise CheckError(f"{x} read before it was written   at: {ins.text if ins else ''}")
                return self.s[x]
            if x == "scc":
                return self.scc
            raise CheckError(f"not a scalar operand: {x}")
        if isinstance(x, float):
            return f2u(x)
        return int(x) & 0xffffffff

    def s64(self, pair, ins=None):
        if isinstance(pair, str):
            return self.inputs[pair] & 0xffffffffffffffff
        if isinstance(pair, int):
            return pair & 0xffffffffffffffff
        return self.sval(pair[0], ins) | (self.sval(pair[1], ins) << 32)

    def val(self, x, ins):
        """per-lane uint32 array"""
        if isinstance(x, str) and x[0] in "va" and is_reg(x):
            self.touch(x, ins, False)
            return self.v[self.ridx(x)].copy()
        vin = getattr(self, "vector_inputs", None)
        if vin is not None and isinstance(x, str) and x in vin:          # a named statement operand held in a VGPR
            return np.asarray(vin[x], dtype=np.uint32).copy()
        return np.full(LANES, self.sval(x, ins), np.uint32)

    def setv(self, r, arr, ins, masked=True):
        self.touch(r, ins, True)
        i = self.ridx(r)
        arr = np.asarray(arr, dtype=np.uint32)
        if masked:
            self.v[i] = np.where(self.exec, arr, self.v[i])
        else:
            self.v[i] = arr

    def sets(self, r, val, ins):
        self.touch(r, ins, True)
        self.s[r] = int(val) & 0xffffffff

    def _dma_vs_reads(self, dst, size, ins):
        """an LDS-DMA write is not ordered behind outstanding LDS reads: it may only be issued beside reads of OTHER bytes"""
        for ent in self.lgkm:
            if ent[0] == "ds" and ent[1]:
                rng = ent[2] if len(ent) > 2 else None
                if rng is None or (rng[0] < dst + size and dst < rng[1]):
                    raise CheckError(f"LDS-DMA issued with LDS reads of its destination outstanding (nothing orders its write behind them)   at: {ins.text}")

    # -- counters --
    def wait(self, vm, lgkm):
        if lgkm is not None:
            if any(e[0] == "smem" for e in self.lgkm) and lgkm != 0:
                raise CheckError("counted lgkmcnt wait with a scalar memory operation outstanding (they return out of order)")
            while len(self.lgkm) > lgkm:
                dst = self.lgkm.pop(0)[1]
                for r in dst:
                    self.poison.pop(r, None)
        if vm is not None:
            while len(self.vm) > vm:
                ent = self.vm.pop(0)
                if isinstance(ent, dict):          # LDS-DMA: the bytes land in LDS when the wait covers the request
                    ent["apply"]()
                else:
                    for r in ent:
                        self.poison.pop(r, None)

    # -- execution --
    def run(self, prog, max_steps=2_000_000, checker=None):
        labels = {ins.label: i for i, ins in enumerate(prog) if ins.kind == "label"}
        pc = 0
        steps = 0
        hist = []
        while pc < len(prog):
            ins = prog[pc]
            pc += 1
            steps += 1
            if steps > max_steps:
                raise CheckError("simulation does not terminate")
            if ins.kind == "label":
                continue
            if checker is not None and ins.kind != "pseudo":
                checker(hist, ins)
                hist.append(ins)
                if len(hist) > 64:
                    del hist[:32]
            fx = getattr(ins, "fx", None)
            if fx is None:
                raise CheckError(f"instruction without semantics: {ins.text}")
            tgt = self.step(ins, fx)
            if tgt == "__end__":
                return
            if tgt is not None:
                pc = labels[tgt]

    def step(self, ins, fx):
        op = fx[0]
        f32 = np.float32
        if op == "pseudo":
            h = self.hooks.get(fx[1])
            if h is None:
                raise CheckError(f"no hook for pseudo {fx[1]}")
            return h(self, *fx[2:])
        if op == "barrier":
            h = getattr(self, "barrier_hook", None)
            if h is not None:
                h(self)
            return None
        if op == "nop":
            return None
        if op == "waitcnt":
            self.wait(fx[1], fx[2])
            return None
        if op == "branch":
            mn, tgt = fx[1], fx[2]
            take = {"s_branch": True, "s_cbranch_scc0": self.scc == 0, "s_cbranch_scc1": self.scc == 1,
                    "s_cbranch_vccz": not self.vcc.any(), "s_cbranch_vccnz": bool(self.vcc.any()),
                    "s_cbranch_execz": not self.exec.any()}[mn]
            return tgt if take else None
        if op == "v_mov_b32":
            self.setv(fx[1], self.val(fx[2], ins), ins)
        elif op in ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_max_f32"):
            a, b = u2f(self.val(fx[2], ins)), u2f(self.val(fx[3], ins))
            with np.errstate(all="ignore"):
                r = {"v_mul_f32": a * b, "v_add_f32": a + b, "v_sub_f32": a - b, "v_max_f32": np.maximum(a, b)}[op]
            self.setv(fx[1], as_u32(r.astype(f32)), ins)
        elif op == "v_fma_f32":
            a, b, c = (u2f(self.val(x, ins)) for x in fx[2:5])
            n = fx[5]
            r = fma32(-a if n[0] else a, -b if n[1] else b, -c if n[2] else c)
            self.setv(fx[1], as_u32(r), ins)
        elif op == "v_cvt_pk_bf16_f32":
            lo, hi = bf16_rne(u2f(self.val(fx[2], ins))), bf16_rne(u2f(self.val(fx[3], ins)))
            self.setv(fx[1], lo | (hi << 16), ins)
        elif op in ("v_rcp_f32", "v_sqrt_f32", "v_log_f32", "v_exp_f32"):
            a = u2f(self.val(fx[2], ins)).astype(np.float64)
            with np.errstate(all="ignore"):
                r = {"v_rcp_f32": 1.0 / a, "v_sqrt_f32": np.sqrt(a), "v_log_f32": np.log2(a), "v_exp_f32": np.exp2(a)}[op]
            self.setv(fx[1], as_u32(r.astype(f32)), ins)
        elif op in ("v_add_u32", "v_sub_u32", "v_and_b32", "v_mul_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32"):
            a, b = self.val(fx[2], ins).astype(np.uint64), self.val(fx[3], ins).astype(np.uint64)
            r = {"v_add_u32": a + b, "v_sub_u32": a - b, "v_and_b32": a & b, "v_mul_u32_u24": (a & 0xffffff) * (b & 0xffffff),
                 "v_mul_lo_u32": a * b, "v_mul_hi_u32": (a * b) >> 32}[op]
            self.setv(fx[1], (r & 0xffffffff).astype(np.uint32), ins)
        elif op in ("v_lshlrev_b32", "v_lshrrev_b32"):
            sh, a = self.val(fx[2], ins).astype(np.uint64) & 31, self.val(fx[3], ins).astype(np.uint64)
            r = (a << sh) if op == "v_lshlrev_b32" else (a >> sh)
            self.setv(fx[1], (r & 0xffffffff).astype(np.uint32), ins)
        elif op == "v_mad_u32_u24":
            a, b, c = (self.val(x, ins).astype(np.uint64) for x in fx[2:5])
            self.setv(fx[1], (((a & 0xffffff) * (b & 0xffffff) + c) & 0xffffffff).astype(np.uint32), ins)
        elif op == "v_mbcnt_lo":
            self.setv(fx[1], np.minimum(np.arange(LANES), 32).astype(np.uint32), ins)
        elif op == "v_mbcnt_hi":
            self.setv(fx[1], (self.val(fx[1], ins) + np.maximum(np.arange(LANES) - 32, 0)).astype(np.uint32), ins)
        elif op == "v_readfirstlane_b32":
            lane = int(np.argmax(self.exec)) if self.exec.any() else 0
            self.sets(fx[1], self.val(fx[2], ins)[lane], ins)
        elif op == "v_permlane32_swap_b32":
            a, b = self.val(fx[1], ins), self.val(fx[2], ins)
            na, nb = a.copy(), b.copy()
            na[32:], nb[:32] = b[:32], a[32:]
            self.setv(fx[1], na, ins)
            self.setv(fx[2], nb, ins)
        elif op == "v_cmp":
            cond, ty = fx[1], fx[2]
            a, b = self.val(fx[3], ins), self.val(fx[4], ins)
            if ty == "f32":
                a, b = u2f(a), u2f(b)
            elif ty == "i32":
                a, b = a.view(np.int32), b.view(np.int32)
            r = {"lt": a < b, "le": a <= b, "eq": a == b, "ge": a >= b, "gt": a > b, "lg": a != b}[cond]
            self.vcc = np.where(self.exec, r, False)
        elif op == "v_cndmask_b32":
            self.setv(fx[1], np.where(self.vcc, self.val(fx[3], ins), self.val(fx[2], ins)), ins)
        elif op == "v_div_scale_f32":
            d, sdst, a, b, c = fx[1:]
            av = u2f(self.val(a, ins))
            for x in (b, c):
                xv = np.abs(u2f(self.val(x, ins)).astype(np.float64))
                if not ((xv[self.exec] > 1e-30) & (xv[self.exec] < 1e30)).all():
                    raise CheckError(f"v_div_scale_f32 outside the modelled range   at: {ins.text}")
            self.setv(d, as_u32(av), ins)
            if sdst == "vcc":
                self.vcc = np.zeros(LANES, bool)
            else:
                for r in sdst:
                    self.sets(r, 0, ins)
        elif op == "v_div_fmas_f32":
            if self.vcc.any():
                raise CheckError("v_div_fmas_f32 with vcc set (scaled operands are not modelled)")
            a, b, c = (u2f(self.val(x, ins)) for x in fx[2:5])
            self.setv(fx[1], as_u32(fma32(a, b, c)), ins)
        elif op == "v_div_fixup_f32":
            self.setv(fx[1], self.val(fx[2], ins), ins)
        elif op == "mfma":
            d16, a4, b4, c16 = fx[1:]
            A = [self.val(r, ins) for r in a4]
            B = [self.val(r, ins) for r in b4]
            C = [u2f(self.val(r, ins)) for r in c16] if c16 != 0 else [np.zeros(LANES, f32)] * 16
            D = mfma_32x32x16_bf16(A, B, C)
            for r, row in zip(d16, D):
                self.setv(r, as_u32(row), ins, masked=False)
        elif op == "ds_read":
            n, d, addr, off = fx[1:]
            a = self.val(addr, ins).astype(np.int64) + off
            for w, r in enumerate(d):
                vals = np.zeros(LANES, np.uint32)
                for lane in range(LANES):
                    if self.exec[lane]:
                        p = int(a[lane]) + 4 * w
                        if p % 4 or p < 0 or p + 4 > len(self.lds):
                            raise CheckError(f"LDS read at {p}   at: {ins.text}")
                        vals[lane] = int.from_bytes(self.lds[p:p + 4].tobytes(), "little")
                self.setv(r, vals, ins)
            if (a[self.exec] % (4 * min(n, 4))).any() and n == 4:
                raise CheckError(f"ds_read_b128 with an address that is not 16-byte aligned   at: {ins.text}")
            self.lgkm.append(("ds", list(d), (int(a[self.exec].min()), int(a[self.exec].max()) + 4 * n)))
            for r in d:
                self.poison[r] = f"LDS read ({ins.text})"
        elif op == "ds_read_tr":
            d2, addr, off = fx[1:]
            a = self.val(addr, ins).astype(np.int64) + off
            if (a % 8).any() or (a < 0).any() or (a + 8 > len(self.lds)).any():
                raise CheckError(f"ds_read_b64_tr_b16 address   at: {ins.text}")
            elems = np.zeros((LANES, 4), np.uint32)                    # what each lane's address points at: four 16-bit elements
            for lane in range(LANES):
                p = int(a[lane])
                elems[lane] = np.frombuffer(self.lds[p:p + 8].tobytes(), np.uint16)
            res = np.zeros((LANES, 4), np.uint32)
            for lane in range(LANES):
                g, l = lane & ~15, lane & 15
                for j in range(4):
                    res[lane, j] = elems[g + 4 * j + (l >> 2), l & 3]
            self.setv(d2[0], res[:, 0] | (res[:, 1] << 16), ins)
            self.setv(d2[1], res[:, 2] | (res[:, 3] << 16), ins)
            self.lgkm.append(("ds", list(d2), (int(a.min()), int(a.max()) + 8)))
            for r in d2:
                self.poison[r] = f"LDS read ({ins.text})"
        elif op == "ds_write":
            n, addr, src, off = fx[1:]
            a = self.val(addr, ins).astype(np.int64) + off
            if n == 4 and (a[self.exec] % 16).any():
                raise CheckError(f"ds_write_b128 with an address that is not 16-byte aligned   at: {ins.text}")
            for w, r in enumerate(src):
                vals = self.val(r, ins)
                for lane in range(LANES):
                    if self.exec[lane]:
                        p = int(a[lane]) + 4 * w
                        if p < 0 or p + 4 > len(self.lds):
                            raise CheckError(f"LDS write at {p}   at: {ins.text}")
                        self.lds[p:p + 4] = np.frombuffer(int(vals[lane]).to_bytes(4, "little"), np.uint8)
            self.lgkm.append(("ds", []))
        elif op == "global_load":
            n, d, voff, sbase, off = fx[1:]
            base = self.s64(sbase, ins)
            vo = self.val(voff, ins).astype(np.int64)
            for w, r in enumerate(d):
                vals = np.zeros(LANES, np.uint32)
                for lane in range(LANES):
                    if self.exec[lane]:
                        arr, o = self._g(base + int(vo[lane]) + off + 4 * w, 4)
                        vals[lane] = int.from_bytes(arr[o:o + 4].tobytes(), "little")
                self.setv(r, vals, ins)
            self.vm.append(list(d))
            for r in d:
                self.poison[r] = f"global load ({ins.text})"
        elif op == "global_store":
            n, voff, src, sbase, off = fx[1:]
            base = self.s64(sbase, ins)
            vo = self.val(voff, ins).astype(np.int64)
            for w, r in enumerate(src):
                vals = self.val(r, ins)
                for lane in range(LANES):
                    if self.exec[lane]:
                        arr, o = self._g(base + int(vo[lane]) + off + 4 * w, 4)
                        arr[o:o + 4] = np.frombuffer(int(vals[lane]).to_bytes(4, "little"), np.uint8)
            self.vm.append([])
        elif op == "global_load_lds":
            voff, sbase, off = fx[1:]
            base = self.s64(sbase, ins)
            vo = self.val(voff, ins).astype(np.int64)
            data = []
            for lane in range(LANES):
                arr, o = self._g(base + int(vo[lane]) + off, 16)
                data.append(arr[o:o + 16].copy())
            dst = self.m0 + off
            if dst % 16 or dst + 1024 > len(self.lds):
                raise CheckError(f"LDS-DMA destination {dst}   at: {ins.text}")
            self._dma_vs_reads(dst, 1024, ins)

            def apply(dst=dst, data=data):
                for lane in range(LANES):
                    self.lds[dst + 16 * lane:dst + 16 * lane + 16] = data[lane]
            self.vm.append({"apply": apply})
        elif op == "global_load_lds_dword":
            voff, sbase, off = fx[1:]
            base = self.s64(sbase, ins)
            vo = self.val(voff, ins).astype(np.int64)
            data = []
            for lane in range(LANES):
                arr, o = self._g(base + int(vo[lane]) + off, 4)
                data.append(arr[o:o + 4].copy())
            dst = self.m0 + off
            if dst % 4 or dst + 256 > len(self.lds):
                raise CheckError(f"LDS-DMA destination {dst}   at: {ins.text}")
            self._dma_vs_reads(dst, 256, ins)

            def apply4(dst=dst, data=data):
                for lane in range(LANES):
                    self.lds[dst + 4 * lane:dst + 4 * lane + 4] = data[lane]
            self.vm.append({"apply": apply4})
        elif op == "s_cselect_b32":
            self.sets(fx[1], self.sval(fx[2], ins) if self.scc else self.sval(fx[3], ins), ins)
        elif op == "s_min_u32":
            a_, b_ = self.sval(fx[2], ins), self.sval(fx[3], ins)
            self.sets(fx[1], min(a_, b_), ins)
            self.scc = int(a_ <= b_)
        elif op == "s_add_m0":
            self.m0 = (self.sval(fx[1], ins) + fx[2]) & 0xffffffff
        elif op in ("s_or_b32", "s_and_b32"):
            a_, b_ = self.sval(fx[2], ins), self.sval(fx[3], ins)
            r = (a_ | b_) if op == "s_or_b32" else (a_ & b_)
            self.sets(fx[1], r, ins)
            self.scc = int(r != 0)
        elif op == "s_mov_b32":
            self.sets(fx[1], self.sval(fx[2], ins), ins)
        elif op == "s_mov_b64":
            v = self.s64(fx[2], ins)
            self.sets(fx[1][0], v & 0xffffffff, ins)
            self.sets(fx[1][1], v >> 32, ins)
        elif op in ("s_add_u32", "s_addc_u32", "s_sub_u32"):
            a, b = self.sval(fx[2], ins), self.sval(fx[3], ins)
            r = a + b + (self.scc if op == "s_addc_u32" else 0) if op != "s_sub_u32" else a - b
            self.sets(fx[1], r, ins)
            self.scc = int(r > 0xffffffff) if op != "s_sub_u32" else int(r < 0)
        elif op == "s_mul_i32":
            self.sets(fx[1], self.sval(fx[2], ins) * self.sval(fx[3], ins), ins)
        elif op == "s_lshl_b32":
            r = (self.sval(fx[2], ins) << (self.sval(fx[3], ins) & 31)) & 0xffffffff
            self.sets(fx[1], r, ins)
            self.scc = int(r != 0)
        elif op == "s_cmp":
            cond, ty, a, b = fx[1], fx[2], self.sval(fx[3], ins), self.sval(fx[4], ins)
            if ty == "i32":
                a, b = a - (1 << 32) * (a >> 31), b - (1 << 32) * (b >> 31)
            self.scc = int({"lt": a < b, "le": a <= b, "eq": a == b, "ge": a >= b, "gt": a > b, "lg": a != b}[cond])
        elif op == "s_cmp_u64_ne0":
            self.scc = int(self.s64(fx[1], ins) != 0)
        elif op == "s_exec_set":
            m = np.zeros(LANES, bool)
            if fx[1] == "all":
                m[:] = True
            elif fx[1] == "lo":
                m[:32] = True
            elif fx[1] == "hi":
                m[32:] = True
            else:
                m[0] = True
            self.exec = m
        elif op == "s_memtime":
            self.time += 1000
            self.sets(fx[1][0], self.time, ins)
            self.sets(fx[1][1], 0, ins)
            self.lgkm.append(("smem", list(fx[1])))
            for r in fx[1]:
                self.poison[r] = "scalar memory read"
        else:
            raise CheckError(f"no semantics for {op}   at: {ins.text}")
        return None


# ------------------------------------------------------------------------------------------------------------------
# wait states the streams must keep by hand (cdna4 ISA 4.5; LLVM GCNHazardRecognizer, gfx940 rows).  Conservative: every
# instruction counts as one state, an s_nop N as N + 1.
# ------------------------------------------------------------------------------------------------------------------
def check_wait_states(hist, cur):
    if cur.kind in ("label", "pseudo"):
        return
    crd, cwr = set(cur.rd), set(cur.wr)
    dist = 0
    for prev in reversed(hist):
        if prev.kind == "pseudo":
            continue
        pwr = set(prev.wr)
        need = 0
        vgpr_w = {r for r in pwr if r[0] in "va"}
        sgpr_w = {r for r in pwr if r[0] == "s" and r != "scc"}
        if prev.kind == "mfma":
            if cur.kind != "mfma":
                if pwr & (crd | cwr):
                    need = 12                            # XDL (8 pass) write -> VALU / memory read or write of the result
                elif set(prev.rd) & cwr:
                    need = 8                             # XDL reads its operands over its passes -> overwrite by a later instruction
                    if cur.kind in ("ds", "vmem") and prev.fx is not None and prev.fx[0] == "mfma":
                        # a LOAD's data arrives tens of cycles after it issues, and it issues after the MFMA has (in-order issue): only the C
                        # operand, read late, keeps the margin
                        c16 = prev.fx[4]
                        if c16 == 0 or not (set(c16) & cwr):
                            need = 0
            else:
                ov = pwr & crd
                if ov and not (pwr == cwr and ov == pwr):
                    need = 12                            # XDL write -> XDL read as A / B, or as another instruction's C
        elif prev.kind in ("valu", "trans", "perm", "vread"):
            if cur.kind == "mfma" and (vgpr_w & crd):
                need = 2                                 # VALU write -> XDL read
            elif prev.kind == "trans" and cur.kind in ("valu", "perm", "vread", "dsw", "vmemst", "ds", "vmem") and (vgpr_w & crd):
                need = 1                                 # transcendental result forwarded
            elif cur.kind == "perm" and (vgpr_w & crd):
                need = 2
            elif cur.kind == "vread" and (vgpr_w & crd):
                need = 1                                 # VALU write VGPR -> v_readfirstlane / v_readlane
            elif cur.kind in ("vmem", "vmemst", "dma") and (sgpr_w & crd):
                need = 5                                 # VALU write SGPR -> vector memory reads it as base
            elif "vcc" in pwr and cur.fx is not None and cur.fx[0] == "v_div_fmas_f32":
                need = 4                                 # VALU write VCC -> v_div_fmas
        elif prev.kind == "salu":
            if cur.kind == "dma" and "m0" in pwr:
                need = 1
        elif prev.kind in ("vmemst", "dsw"):
            data = set(prev.rd[1:]) if prev.kind == "dsw" else set(prev.rd[1:-2])
            wide = prev.fx is not None and prev.fx[1] >= 3
            if wide and cur.kind in ("valu", "trans", "perm", "mfma", "ds", "vmem") and (data & cwr):
                need = 2                                 # store of more than 64 bits -> overwrite of its data registers
        if need and dist < need:
            raise CheckError(f"wait states: '{cur.text}' needs {need} states after '{prev.text}', has {dist}")
        dist += prev.sem[1] if (prev.kind == "nop") else 1
        if dist > 20:
            break
