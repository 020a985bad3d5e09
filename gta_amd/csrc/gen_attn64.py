#!/usr/bin/env python3
"""gen_attn64.py -- generator of the tile loop of gta_attn64_kernel (gta_fwd64.hip) as one gfx950 assembly stream.

The attention kernel of the two-stage forward plan (source/layers.py:202-211 over the K'/V' images of the pre-pass,
source/utils/gta.py:160-219) with 64 query rows per wave and ONE wave per SIMD: every K' fragment read, every V'
transpose-read pair and every LDS-DMA piece feeds TWO matrix instructions, the O accumulators, the Q' / K' / V' fragments
live in the accumulator half of the register file and never pass through the VALU in the steady state (lazy softmax:
the O rescale is a rare, separate path).  hipcc cannot be made to place ~400 live registers and 5 issue slots per MFMA gap
(r01's gta_fwd3), so the loop is emitted here, instruction by instruction, and included by the kernel as one asm statement.

What this file guarantees on the CPU (run by `make` and by tests/test_host_logic.py):
  * a typed-dataflow simulation of the emitted stream (one wave, the dynamic instruction order for several tile counts,
    with and without forced rebase steps): every MFMA sees exactly the fragments / P words / accumulators it must,
    every score is exponentiated, summed and packed exactly once, no register is overwritten while a consumer is
    outstanding, every LDS read is behind its DMA's wait + barrier and every use of a read behind its lgkmcnt wait;
  * the manual wait-state rules of the ISA on that same order (XDL write -> VALU read, VALU write -> XDL read,
    transcendental forwarding, permlane, M0 -> LDS-DMA).
Numerical parity is the GPU tests' job (tests/test_gpu_forward.py runs every fixture through this kernel).

Register map (dh = 96: KS = 6 k-steps, DB = 3 channel blocks, RB = 2 row blocks of 32 query rows per wave)
  a[ 28:123]  O^T[rb][d]        16 each            v[ 64: 95]  S'[rb], key half 0           16 each
  a[124:171]  Q' fragments      [rb][ks] 4 each    v[ 96:159]  S'[p][rb], key half 1        16 each (p = step parity)
  a[172:219]  K' fragments      [ks][hh] 4 each    v[160:191]  P[p][rb][t], key half 0      4 each (bf16 pairs)
  a[220:255]  V'^T fragments    [slab % 3][d] 4 each   v[192:207]  P[rb][t], key half 1     4 each
  a[  0: 27]  hipcc's (it parks values in the lowest accumulator registers under pressure)
                                                   v[208:239]  -m splat[rb]  16 each (C operand of a tile's first MFMAs)
                                                   v[ 32: 63]  addresses, row sums, m, temporaries
  v[0:31] and the SGPRs not named here belong to the compiler (operands of the statement).
  What crosses the statement's boundary in the accumulator file is in its operand list (emit(): GTA_ATTN64_QFRAGS_*, "+{a[124:127]}"(qfr[0][0]) ...;
  GTA_ATTN64_RESULTS_*, "={a[28:43]}"(oacc[0][0]) ...): hipcc writes the fragments and reads O itself.  The stream ends >= 18 issue states behind
  its last MFMA (the XDL-write -> VALU-read distance: hipcc places no wait states behind inline asm).
"""
import argparse
import sys

RB = 2
KS, DB, CHP = 6, 3, 12       # dh = 96 (configure() below: dh = 64 -> 4, 2, 8)
IMG = 64 * CHP * 16          # one K' or V' tile image (12288 B at dh = 96, 8192 B at dh = 64)
TILE = 2 * IMG               # [K' image | V' image] of one key tile in HBM
NP = IMG // 4096             # 1-KiB LDS-DMA pieces of an image per wave (four waves)
THR_BITS = 0x42c00000        # BOUND_THR = 96.0f (gta_flash_common.h)


# ------------------------------------------------------------------------------------------------------------------
# registers
# ------------------------------------------------------------------------------------------------------------------
def _sbase(p, rb, hh):
    # S' of the hh = 0 half is consumed (phase B of step j - 1) before the next tile's is written (phase A of step j): one
    # buffer; the hh = 1 half of tile j is consumed in phase A of step j while tile j + 1's is written: two (p = step parity)
    return 64 + 16 * rb if hh == 0 else 96 + 16 * (p * 2 + rb)


def S(p, rb, hh, r=None):
    b = _sbase(p, rb, hh)
    return f"v[{b}:{b + 15}]" if r is None else f"v{b + r}"


def Sregs(p, rb, hh):
    b = _sbase(p, rb, hh)
    return [f"v{b + i}" for i in range(16)]


def _pbase(p, rb, hh, t):
    # P of the hh = 0 half of tile j + 1 is packed (phase B of step j) while tile j's is being multiplied: two buffers
    return 160 + 4 * ((p * 2 + rb) * 2 + t) if hh == 0 else 192 + 4 * (rb * 2 + t)


def P(p, rb, hh, t, w=None):
    b = _pbase(p, rb, hh, t)
    return f"v[{b}:{b + 3}]" if w is None else f"v{b + w}"


def Pregs(p, rb, hh, t):
    b = _pbase(p, rb, hh, t)
    return [f"v{b + i}" for i in range(4)]


def MS(rb, i=None):
    b = 208 + 16 * rb
    return f"v[{b}:{b + 15}]" if i is None else f"v{b + i}"


def MSregs(rb):
    return [f"v{208 + 16 * rb + i}" for i in range(16)]


AB = 28        # a[0:27] are left to hipcc: under VGPR pressure it parks values in the lowest accumulator registers (audit_spills.py)
NVS = 3        # V'^T fragment buffers: slab 3 of a tile takes slab 0's registers (its reads sit behind slab 0's MFMAs)


def O(rb, d, i=None):
    b = AB + 16 * (rb * DB + d)
    return f"a[{b}:{b + 15}]" if i is None else f"a{b + i}"


def Oregs(rb, d):
    return [f"a{AB + 16 * (rb * DB + d) + i}" for i in range(16)]


def _qb():
    return AB + 16 * RB * DB             # Q' fragments behind the O accumulators


def _kb():
    return _qb() + 4 * RB * KS


def _vb():
    return _kb() + 8 * KS


def Q(rb, ks):
    b = _qb() + 4 * (rb * KS + ks)
    return f"a[{b}:{b + 3}]"


def Qregs(rb, ks):
    return [f"a{_qb() + 4 * (rb * KS + ks) + i}" for i in range(4)]


def K(ks, hh):
    b = _kb() + 4 * (ks * 2 + hh)
    return f"a[{b}:{b + 3}]"


def Kregs(ks, hh):
    return [f"a{_kb() + 4 * (ks * 2 + hh) + i}" for i in range(4)]


def V(sl, d, half=None):
    b = _vb() + 4 * ((sl % NVS) * DB + d)
    return f"a[{b}:{b + 3}]" if half is None else f"a[{b + 2 * half}:{b + 2 * half + 1}]"


def Vregs(sl, d, half=None):
    b = _vb() + 4 * ((sl % NVS) * DB + d)
    return [f"a{b + i}" for i in range(4)] if half is None else [f"a{b + 2 * half}", f"a{b + 2 * half + 1}"]


def _lb():
    return _vb() + 4 * NVS * DB          # msum: the row sums as two 32x32 accumulators behind the V'^T fragments, then the all-ones A fragment


def Lacc(rb, i=None):
    b = _lb() + 16 * rb
    return f"a[{b}:{b + 15}]" if i is None else f"a{b + i}"


def Lregs(rb):
    return [f"a{_lb() + 16 * rb + i}" for i in range(16)]


ONES = None                              # set by configure(): "a[x:x+3]"


def ONESregs():
    b = _lb() + 16 * RB
    return [f"a{b + i}" for i in range(4)]


# low literal VGPRs
KOFF = [f"v{32 + i}" for i in range(KS)]             # K' fragment byte offsets (ring base 0)
VOFF = [[f"v{38 + 2 * d + h}" for h in range(2)] for d in range(DB)]   # V' transpose-read offsets (V ring base folded in)


def configure(dh):
    """head dimension of the stream: 96 (MSN) or 64 (CLEVR-TR, the 2-D DiT branch).  The accumulator-file map is laid out from the
    counts (O | Q' | K' | V'^T); the vector half does not depend on dh."""
    global KS, DB, CHP, IMG, TILE, NP, KOFF, VOFF
    assert dh in (64, 96)
    KS, DB, CHP = dh // 16, dh // 32, dh // 8
    IMG = 64 * CHP * 16
    TILE = 2 * IMG
    NP = IMG // 4096
    KOFF = [f"v{32 + i}" for i in range(KS)]
    VOFF = [[f"v{38 + 2 * d + h}" for h in range(2)] for d in range(DB)]
    assert _vb() + 4 * NVS * DB <= 256
    global ONES
    b = _lb() + 16 * RB
    ONES = f"a[{b}:{b + 3}]"
LA = [[f"v{44 + 2 * rb + e}" for e in range(2)] for rb in range(RB)]   # row sums of the hh = 1 halves (phase A), even / odd
LB = [[f"v{48 + 2 * rb + e}" for e in range(2)] for rb in range(RB)]   # row sums of the hh = 0 halves (phase B)
MRUN = [f"v{52 + rb}" for rb in range(RB)]
T = [f"v{54 + i}" for i in range(10)]                # temporaries v54..v63
# literal SGPRs
S_J, S_J1, S_KN, S_WOFF, S_T0, S_T1 = "s84", "s85", "s86", "s87", "s88", "s89"
S_KPTR, S_VPTR = "s[90:91]", "s[92:93]"
S_KPTR_LO, S_KPTR_HI, S_VPTR_LO, S_VPTR_HI = "s90", "s91", "s92", "s93"
S_NM1, S_NMR, S_NMR1 = "s94", "s95", "s96"           # n-1, n-R, n-R+1


class Ins:
    __slots__ = ("text", "kind", "rd", "wr", "sem", "label", "target")

    def __init__(self, text, kind, rd=(), wr=(), sem=None, label=None, target=None):
        self.text, self.kind, self.rd, self.wr, self.sem = text, kind, tuple(rd), tuple(wr), sem
        self.label, self.target = label, target

    def __repr__(self):
        return self.text


def nop(n):
    return Ins(f"s_nop {n - 1}", "nop", sem=("nop", n))


def lgkm(n):
    return Ins(f"s_waitcnt lgkmcnt({n})", "wait", sem=("lgkm", n))


def vmw(n):
    return Ins(f"s_waitcnt vmcnt({n})", "wait", sem=("vm", n))


def barrier():
    return Ins("s_barrier", "barrier", sem=("barrier",))


def ready(regs):
    """scheduling hint for insert_waits: all of `regs` must have landed here (one counted wait for a group of fragments
    instead of one in front of each MFMA that first uses one of them)"""
    return Ins("", "ready", regs, (), None)


def label(name):
    return Ins(f"{name}:", "label", label=name)


def salu(text, rd=(), wr=(), sem=None):
    return Ins(text, "salu", rd, wr, sem)


def branch(text, target, sem):
    return Ins(f"{text} {target}", "branch", sem=sem, target=target)


class Gen:
    def __init__(self, R=4, kread_early=True, sched=True, boundary_in_a=False, ablate=(), carry=False, dma_spread=False, fast_ends=False, pk_sum=False,
                 dot_sum=False, first_fast=False, msum=False):
        assert R in (2, 4)
        self.R, self.early, self.sched, self.bina = R, kread_early and R == 4, sched, boundary_in_a
        self.ablate = set(ablate)          # timing-only builds (wrong results): novalu, nods, nodma, nobar
        self.carry = carry                 # K' reads stay in flight across the step labels
        self.dma_spread = dma_spread       # boundary in phase A: one LDS-DMA piece per gap instead of three back to back
        self.pk_sum = pk_sum               # row sums as v_pk_add_f32 on (even, odd) value pairs
        self.msum = msum                   # row sums on the matrix pipe: l^T += 1 P per key slab (dh = 64: the pipe has the slack, the VALU does not)
        assert not msum or _lb() + 16 * RB + 4 <= 256, "msum needs 36 free accumulator registers (dh = 64)"
        self.first_fast = first_fast       # tile 0 without the rebase when its scores cannot leave the lazy softmax's window around m = 0
        self.dot_sum = dot_sum             # row sums from the PACKED words: one v_dot2c_f32_bf16 (x 1.0, 1.0) per pair of scores
        self.fast_ends = fast_ends         # O zeroed inside the head's MFMA gaps; the last step's softmax inside its P V MFMAs
        self.KRING, self.VRING = 0, R * IMG            # LDS byte offsets of the two rings

    # ---- primitive emitters --------------------------------------------------------------------------------------
    def ds_k(self, slot, ks, hh, tile_rel):
        off = self.KRING + slot * IMG + hh * 32 * CHP * 16
        return Ins(f"ds_read_b128 {K(ks, hh)}, {KOFF[ks]} offset:{off}", "ds", [KOFF[ks]], Kregs(ks, hh),
                   ("ds_k", slot, ks, hh, tile_rel))

    def ds_v(self, slot, sl, d, half):
        off = slot * IMG + sl * 16 * CHP * 16
        return Ins(f"ds_read_b64_tr_b16 {V(sl, d, half)}, {VOFF[d][half]} offset:{off}", "ds", [VOFF[d][half]],
                   Vregs(sl, d, half), ("ds_v", slot, sl, d, half))

    def qk(self, p, rb, hh, ks, tile_rel, c_zero=False):
        c = "0" if c_zero else (MS(rb) if ks == 0 else S(p, rb, hh))
        rd = Kregs(ks, hh) + Qregs(rb, ks) + ([] if c_zero else (MSregs(rb) if ks == 0 else Sregs(p, rb, hh)))
        return Ins(f"v_mfma_f32_32x32x16_bf16 {S(p, rb, hh)}, {K(ks, hh)}, {Q(rb, ks)}, {c}", "mfma", rd, Sregs(p, rb, hh),
                   ("qk", p, rb, hh, ks, tile_rel, c_zero))

    def pv(self, p, sl, d, rb):
        hh, t = sl >> 1, sl & 1
        return Ins(f"v_mfma_f32_32x32x16_bf16 {O(rb, d)}, {V(sl, d)}, {P(p, rb, hh, t)}, {O(rb, d)}", "mfma",
                   Vregs(sl, d) + Pregs(p, rb, hh, t) + Oregs(rb, d), Oregs(rb, d), ("pv", p, sl, d, rb))

    def pl(self, p, sl, rb):
        """l^T[rb] += 1 (32 x 16) P (16 keys of slab sl x 32 rows): every output row is the rows' sum over the slab's keys"""
        hh, t = sl >> 1, sl & 1
        return Ins(f"v_mfma_f32_32x32x16_bf16 {Lacc(rb)}, {ONES}, {P(p, rb, hh, t)}, {Lacc(rb)}", "mfma",
                   ONESregs() + Pregs(p, rb, hh, t) + Lregs(rb), Lregs(rb), ("pl", p, sl, rb))

    def pv_list(self, c):
        """the P V MFMAs of a step in issue order: per slab (d, rb) pairs, then (msum) the slab's two row-sum MFMAs"""
        out = []
        for sl in range(4):
            out += [self.pv(c & 1, sl, g >> 1, g & 1) for g in range(2 * DB)]
            if self.msum:
                out += [self.pl(c & 1, sl, rb) for rb in range(RB)]
        return out

    @property
    def SLN(self):
        return 2 * DB + (RB if self.msum else 0)          # MFMAs per key slab in phase B

    def exp(self, p, rb, hh, r, tile_rel):
        x = S(p, rb, hh, r)
        return Ins(f"v_exp_f32 {x}, {x}", "trans", [x], [x], ("exp", p, rb, hh, r, tile_rel))

    def add(self, p, rb, hh, r, acc):
        x = S(p, rb, hh, r)
        return Ins(f"v_add_f32 {acc}, {acc}, {x}", "valu", [acc, x], [acc], ("add", p, rb, hh, r))

    def add2(self, p, rb, hh, e2, acc):
        """both row-sum adds of a pair of values as one packed add (the accumulators of a row block are a register pair)"""
        b0 = int(S(p, rb, hh, 2 * e2)[1:])
        a0 = int(acc[0][1:])
        assert b0 % 2 == 0 and a0 % 2 == 0 and acc[1] == f"v{a0 + 1}"
        x0, x1 = S(p, rb, hh, 2 * e2), S(p, rb, hh, 2 * e2 + 1)
        return Ins(f"v_pk_add_f32 v[{a0}:{a0 + 1}], v[{a0}:{a0 + 1}], v[{b0}:{b0 + 1}]", "valu", [acc[0], acc[1], x0, x1], [acc[0], acc[1]],
                   ("add2", p, rb, hh, 2 * e2))

    def pack(self, p, rb, hh, e2):
        # packed word k = e2 of half hh: values r = 2 e2, 2 e2 + 1 -> P[rb][hh][r >> 3][(r & 7) >> 1]
        r = 2 * e2
        dst = P(p, rb, hh, r >> 3, (r & 7) >> 1)
        a, b = S(p, rb, hh, r), S(p, rb, hh, r + 1)
        return Ins(f"v_cvt_pk_bf16_f32 {dst}, {a}, {b}", "valu", [a, b], [dst], ("pack", p, rb, hh, r >> 3, (r & 7) >> 1))

    # ---- softmax VALU of one 32-key half of one tile, as an ordered list (exp first, sums / packs trail) -----------
    def softmax_items(self, p, hh, tile_rel, phase):
        """-> list of Ins in dependency-safe order: exps of pair k, then (one pair later) its adds and its pack."""
        acc = LA if phase == "A" else LB
        out = []
        pairs = [(rb, e2) for e2 in range(8) for rb in range(RB)]      # interleave the row blocks
        prev = prev2 = None
        for rb, e2 in pairs:
            out.append(self.exp(p, rb, hh, 2 * e2, tile_rel))
            out.append(self.exp(p, rb, hh, 2 * e2 + 1, tile_rel))
            if prev is not None:
                prb, pe2 = prev
                out += self.sums(p, prb, hh, pe2, acc[prb])
                out.append(self.pack(p, prb, hh, pe2))
            if self.dot_sum and prev2 is not None:
                out.append(self.dotsum(p, prev2[0], hh, prev2[1], acc[prev2[0]]))
            prev2, prev = prev, (rb, e2)
        prb, pe2 = prev
        out += self.sums(p, prb, hh, pe2, acc[prb])
        out.append(self.pack(p, prb, hh, pe2))
        if self.dot_sum:
            out.append(self.dotsum(p, prev2[0], hh, prev2[1], acc[prev2[0]]))
            out.append(self.dotsum(p, prb, hh, pe2, acc[prb]))
        return out

    def dotsum(self, p, rb, hh, e2, acc):
        """l += lo + hi of packed word e2 (bf16 pair x (1.0, 1.0) as a literal): the row sum counts exactly the P the matrix core
        multiplies with V', in one instruction per two scores"""
        r = 2 * e2
        w = P(p, rb, hh, r >> 3, (r & 7) >> 1)
        a = acc[e2 & 1]
        return Ins(f"v_dot2c_f32_bf16 {a}, 0x3f803f80, {w}", "valu", [a, w], [a], ("dotsum", p, rb, hh, r >> 3, (r & 7) >> 1))

    def sums(self, p, rb, hh, e2, acc):
        if self.dot_sum or self.msum:
            return []
        if self.pk_sum:
            return [self.add2(p, rb, hh, e2, acc)]
        return [self.add(p, rb, hh, 2 * e2, acc[0]), self.add(p, rb, hh, 2 * e2 + 1, acc[1])]

    # ---- the boundary of step copy c: tiles landed, everyone past the previous step, next DMA requests ------------
    def boundary(self, c, first_of_item=False):
        R = self.R
        out = []
        # landed before this step: with R = 4 everything but the last two boundaries' requests (K'(j+2), V'(j+1) and older);
        # with R = 2 everything (K'(j+1), V'(j) were requested one boundary ago)
        out.append(vmw(2 * NP if R == 4 else 0))
        out.append(barrier())
        out += self.dma_requests(c)
        return out

    def dma_requests(self, c):
        """K'(j + R) -> K slot c (tile j's slot, consumed), V'(j + R - 1) -> V slot c - 1 (tile j - 1's)."""
        R = self.R
        out = []
        # the K stream passes to the next item's images at j == n - R (a step of copy 0), the V stream at j == n - R + 1
        if c == 0:
            out.append(salu(f"s_cmp_eq_u32 {S_J}, {S_NMR}", [S_J, S_NMR], ["scc"], ("cmp_nmr",)))
            out.append(salu(f"s_cselect_b64 {S_KPTR}, %[nxt_k], {S_KPTR}", ["scc", S_KPTR], [S_KPTR], ("ksel",)))
        if c == 1 % R:
            out.append(salu(f"s_cmp_eq_u32 {S_J}, {S_NMR1}", [S_J, S_NMR1], ["scc"], ("cmp_nmr1",)))
            out.append(salu(f"s_cselect_b64 {S_VPTR}, %[nxt_v], {S_VPTR}", ["scc", S_VPTR], [S_VPTR], ("vsel",)))
        kslot, vslot = c, (c - 1) % R
        out.append(salu(f"s_add_u32 m0, {S_WOFF}, {self.KRING + kslot * IMG}", [S_WOFF], ["m0"]))
        out.append(nop(1))
        for i in range(NP):
            out.append(Ins(f"global_load_lds_dwordx4 %[lane16], {S_KPTR} offset:{1024 * i}", "dma", ["m0", S_KPTR], [],
                           ("dma", "K", kslot, i)))
        out.append(salu(f"s_add_u32 m0, {S_WOFF}, {self.VRING + vslot * IMG}", [S_WOFF], ["m0"]))
        out.append(nop(1))
        for i in range(NP):
            out.append(Ins(f"global_load_lds_dwordx4 %[lane16], {S_VPTR} offset:{1024 * i}", "dma", ["m0", S_VPTR], [],
                           ("dma", "V", vslot, i)))
        out.append(salu(f"s_add_u32 {S_KPTR_LO}, {S_KPTR_LO}, {TILE}", [S_KPTR_LO], [S_KPTR_LO, "scc"], ("kadv", 0)))
        out.append(salu(f"s_addc_u32 {S_KPTR_HI}, {S_KPTR_HI}, 0", [S_KPTR_HI, "scc"], [S_KPTR_HI, "scc"], ("kadv", 1)))
        out.append(salu(f"s_add_u32 {S_VPTR_LO}, {S_VPTR_LO}, {TILE}", [S_VPTR_LO], [S_VPTR_LO, "scc"], ("vadv", 0)))
        out.append(salu(f"s_addc_u32 {S_VPTR_HI}, {S_VPTR_HI}, 0", [S_VPTR_HI, "scc"], [S_VPTR_HI, "scc"], ("vadv", 1)))
        return out

    # ---- decision for tile j + 1 (lazy softmax: does |q'| max|k'| - m stay below the threshold?) -----------------
    def decision(self, c, slow_label):
        out = [salu(f"s_add_u32 {S_J1}, {S_J}, 1", [S_J], [S_J1, "scc"], ("j1",)),
               Ins(f"v_readlane_b32 {S_KN}, %[kn], {S_J1}", "valu", [S_J1], [S_KN], ("readkn",)),
               Ins(f"v_fma_f32 {T[0]}, %[qn0], {S_KN}, -{MRUN[0]}", "valu", [S_KN, MRUN[0]], [T[0]]),
               Ins(f"v_fma_f32 {T[1]}, %[qn1], {S_KN}, -{MRUN[1]}", "valu", [S_KN, MRUN[1]], [T[1]]),
               Ins(f"v_max_f32 {T[0]}, {T[0]}, {T[1]}", "valu", [T[0], T[1]], [T[0]]),
               Ins(f"v_cmp_lt_f32 vcc, 0x{THR_BITS:08x}, {T[0]}", "valu", [T[0]], ["vcc"])]
        tail = []
        if c == self.R - 2:      # only this copy can be the step in front of the item's last tile (n % R == 0)
            tail = [salu(f"s_cmp_eq_u32 {S_J1}, %[tailj]", [S_J1], ["scc"]),
                    branch("s_cbranch_scc1", slow_label, ("br_tail",))]
        return out, [branch("s_cbranch_vccnz", slow_label, ("br_need",))] + tail

    # ---- phase A of step copy c: S'(j+1) = K'(j+1) Q'^T - m   ||  softmax of the hh = 1 half of tile j -------------
    def phase_a(self, c, slow_label, with_boundary=False):
        R, p, pn = self.R, c & 1, (c + 1) & 1
        # g -> (ks, hh, rb): the two row blocks of a K' fragment back to back; the hh = 0 accumulators finish first (phase B
        # exponentiates them right away: an XDL result needs 12 states before the VALU may read it)
        mf = [self.qk(pn, g & 1, (g >> 1) & 1, g >> 2, 1) for g in range(4 * KS)]
        valu = self.softmax_items(p, 1, 0, "A")
        dec, dec_br = self.decision(c, slow_label)
        vreads = [self.ds_v(c, sl, d, h) for sl in (0, 1) for d in range(DB) for h in range(2)]
        kreads = [] if self.early else [self.ds_k((c + 1) % R, ks, hh, 1) for ks in range(KS) for hh in range(2)]
        pre, gaps = [], [[] for _ in mf]
        if not self.sched:
            pre = kreads + vreads + valu + dec
            return self.weave(pre, mf, gaps) + dec_br
        # K' reads (not early): ks 0, 1 in front of the MFMAs, the rest two k-steps ahead of their first use
        if kreads:
            pre += kreads[:4]
            for i, kr in enumerate(kreads[4:]):
                gaps[2 * i].append(kr)           # fragment (ks, hh), ks >= 2, is requested in gap 2 i and used from MFMA 4 ks + 2 hh >= 8
        # V' reads of slabs 0, 1: one per gap from gap 10 on (they only have to be there for phase B)
        for i, vr in enumerate(vreads):
            gaps[len(mf) - len(vreads) - 2 + i].append(vr)
        d0 = 2
        if with_boundary:
            # the step's boundary inside the first gaps: nothing this phase reads depends on it (K'(j+1) fragments are in
            # registers, V'(j) was complete one boundary ago); the requests overwrite slots whose readers are behind the barrier
            assert self.early
            bnd = self.boundary(c)
            cut = [i for i, x in enumerate(bnd) if x.kind == "salu" and "m0" in x.wr]
            if self.dma_spread:         # one piece per gap (a piece costs more issue time among other VMEM / LDS requests)
                a0, b0 = cut[0], cut[1]
                groups = [bnd[:a0], bnd[a0:a0 + 3]] + [bnd[a0 + 2 + i:a0 + 3 + i] for i in range(1, NP - 1)] + [bnd[a0 + 1 + NP:b0], bnd[b0:b0 + 3]] + \
                         [bnd[b0 + 2 + i:b0 + 3 + i] for i in range(1, NP)] + [bnd[b0 + 2 + NP:]]
            else:
                groups = [bnd[:cut[0]], bnd[cut[0]:cut[1]], bnd[cut[1]:cut[1] + 5], bnd[cut[1] + 5:]]
            for gi, grp in enumerate(groups):
                gaps[gi] += grp
            d0 = len(groups)
        # decision: needs nothing but scalars; early enough that the branch is resolved at the phase's end
        for i, di in enumerate(dec):
            gaps[d0 + i].append(di)
        self.deal(valu, gaps, range(0, len(mf)))
        if self.early:
            pre.append(ready([r for hh in range(2) for r in Kregs(0, hh) + Kregs(1, hh)]))
            for ks in range(2, KS, 2):
                gaps[4 * ks - 1].append(ready([r for hh in range(2) for r in Kregs(ks, hh) + Kregs(ks + 1, hh)]))
        return self.weave(pre, mf, gaps) + dec_br

    # ---- phase B of step copy c: O += V'(j) P(j)   ||  softmax of the hh = 0 half of tile j + 1 -------------------
    def phase_b(self, c, plain=False, last=False):
        R, pn = self.R, (c + 1) & 1
        mf = self.pv_list(c)                                  # g -> (slab, d, rb) (+ the slab's row-sum MFMAs)
        vreads = [self.ds_v(c, sl, d, h) for sl in (2, 3) for d in range(DB) for h in range(2)]
        kreads = [self.ds_k((c + 2) % R, ks, hh, 2) for ks in range(KS) for hh in range(2)] if (self.early and not last) else []
        valu = [] if plain else self.softmax_items(pn, 0, 1, "B")
        pre, gaps = [], [[] for _ in mf]
        if not self.sched or plain:
            pre = vreads[:2 * DB] + kreads + [x for x in valu if x.sem[0] != "pack"]
            gaps[self.SLN - 1] += vreads[2 * DB:]         # slab 3 takes slab 0's registers: behind slab 0's MFMAs
            packs = [x for x in valu if x.sem[0] == "pack"]
            return self.weave(pre, mf, gaps) + packs
        # V' reads of slabs 2 (needed by MFMA 12) and 3 (MFMA 18): one per gap from gap 0
        for i, vr in enumerate(vreads):
            gaps[i].append(vr)
        # K' reads of tile j + 2: one per gap in the second half
        for i, kr in enumerate(kreads):
            gaps[len(mf) - len(kreads) + i].append(kr)
        # (gaps 0, 1 stay free of softmax work: the S' accumulators of phase A's last MFMAs need 12 states before the VALU reads them)
        self.deal(valu, gaps, range(2, len(mf)))
        pre.append(ready([r for d in range(DB) for r in Vregs(0, d)]))
        for sl in (1, 2, 3):
            gaps[self.SLN * sl - 1].append(ready([r for d in range(DB) for r in Vregs(sl, d)]))
        return self.weave(pre, mf, gaps)

    def deal(self, valu, gaps, grange):
        """deal the ordered VALU list into the gaps so that the issue cost runs evenly along the phase: item k goes into the first
        gap at which the cost dealt so far (with what the gaps already carry) stays within that gap's share."""
        # issue cost of a filler in cycles, one wave per SIMD (r01 probes: v_mul 4.9, v_exp 8.9, v_cvt_pk 5.1; an LDS-DMA piece 25-60)
        W = {"trans": 9, "valu": 5, "ds": 6, "dma": 20, "salu": 4, "wait": 4, "nop": 4, "barrier": 8, "perm": 5, "ready": 2}
        cost = lambda x: W.get(x.kind, 4)
        grange = list(grange)
        pre = [sum(cost(x) for x in gaps[g]) for g in grange]
        total = sum(cost(x) for x in valu) + sum(pre)
        share = total / len(grange)
        cum = 0.0                  # cost dealt into gaps grange[0..gi] so far
        gi = 0
        cum += pre[0]
        for x in valu:
            c = cost(x)
            while gi < len(grange) - 1 and cum + c / 2 > share * (gi + 1):
                gi += 1
                cum += pre[gi]
            gaps[grange[gi]].append(x)
            cum += c

    def steady(self, lst):
        """timing-only ablations of the steady-state steps (development: what each part of the stream costs)"""
        drop = lambda x: (("novalu" in self.ablate and x.sem and x.sem[0] in ("exp", "add", "add2", "pack", "dotsum")) or
                          ("nods" in self.ablate and x.kind == "ds") or ("nodma" in self.ablate and x.kind == "dma") or
                          ("nobar" in self.ablate and x.kind == "barrier") or ("nomfma" in self.ablate and x.kind == "mfma"))
        return [x for x in lst if not drop(x)]

    def weave(self, pre, mf, gaps):
        """pre, then MFMA g followed by the fillers of gap g; lgkmcnt waits are inserted later (insert_waits)."""
        out = list(pre)
        for g, m in enumerate(mf):
            out.append(m)
            out += gaps[g]
        return out

    # ---- last step of an item (j = n - 1, copy R - 1): softmax of the hh = 1 half of the last tile, then O += V' P ---
    def tail_step(self):
        c, p = self.R - 1, (self.R - 1) & 1
        out = self.boundary(c)
        vreads = [self.ds_v(c, sl, d, h) for sl in (0, 1) for d in range(DB) for h in range(2)]
        if not self.fast_ends:
            out += vreads + self.softmax_items(p, 1, 0, "A")
            out += self.phase_b(c, plain=True, last=True)
            return out
        # the softmax of the last tile's hh = 1 half inside the gaps of the P V MFMAs of its hh = 0 half (slabs 0, 1)
        mf = self.pv_list(c)
        gaps = [[] for _ in mf]
        v23 = [self.ds_v(c, sl, d, h) for sl in (2, 3) for d in range(DB) for h in range(2)]
        for i, vr in enumerate(v23):
            gaps[i].append(vr)
        self.deal(self.softmax_items(p, 1, 0, "A"), gaps, range(0, 2 * self.SLN - 1))
        pre = vreads + [ready([r for d in range(DB) for r in Vregs(0, d)])]
        for sl in (1, 2, 3):
            gaps[self.SLN * sl - 1].append(ready([r for d in range(DB) for r in Vregs(sl, d)]))
        return out + self.weave(pre, mf, gaps)

    # ---- rebase (the lazy softmax's full path) on the S' buffer of parity p ----------------------------------------
    def rebase(self, p, first, tile_rel, uid="", zero_o=True):
        """true row max of S' (relative to m), m += delta, l and O rescaled, S' and the -m splat re-based.
        first: tile 0 of an item (m = 0, O and l are set to zero instead of rescaled).  The masked tail tile
        (keys >= Tk) is handled in front of the max when s_J1-or-0 == %[tailj]."""
        out = []
        tid = "first" if first else "nf"
        lab_nomask = f"L_reb_nomask_{p}_{tid}{uid}_%="
        # XDL writes of S' (and, from the plain phase B in front, of O) must have retired before the VALU reads them
        out.append(nop(16))
        out.append(Ins("", "pseudo", sem=("rebase_begin", p, first, tile_rel)))
        # ---- tail mask: key of register r of half hh = 64 t + 4 lh + (r & 3) + 8 (r >> 2) + 32 hh ----
        if first:
            out.append(salu(f"s_mov_b32 {S_T0}, 0", [], [S_T0]))
            out.append(salu(f"s_cmp_eq_u32 0, %[tailj]", [], ["scc"]))
        else:
            out.append(salu(f"s_lshl_b32 {S_T0}, {S_J1}, 6", [S_J1], [S_T0, "scc"]))
            out.append(salu(f"s_cmp_eq_u32 {S_J1}, %[tailj]", [S_J1], ["scc"]))
        out.append(branch("s_cbranch_scc0", lab_nomask, ("br_nomask",)))
        # T[2] = Tk - (64 t + 4 lh) : register (hh, r) is masked when (r & 3) + 8 (r >> 2) + 32 hh >= T[2]
        out.append(Ins(f"v_mbcnt_lo_u32_b32 {T[2]}, -1, 0", "valu", [], [T[2]]))
        out.append(Ins(f"v_mbcnt_hi_u32_b32 {T[2]}, -1, {T[2]}", "valu", [T[2]], [T[2]]))
        out.append(Ins(f"v_lshrrev_b32 {T[2]}, 5, {T[2]}", "valu", [T[2]], [T[2]]))
        out.append(Ins(f"v_lshlrev_b32 {T[2]}, 2, {T[2]}", "valu", [T[2]], [T[2]]))
        out.append(Ins(f"v_add_u32 {T[2]}, {S_T0}, {T[2]}", "valu", [T[2], S_T0], [T[2]]))
        out.append(Ins(f"v_sub_u32 {T[2]}, %[Tk], {T[2]}", "valu", [T[2]], [T[2]]))
        out.append(Ins(f"v_mov_b32 {T[3]}, 0xf149f2ca", "valu", [], [T[3]]))         # -1e30f
        for rb in range(RB):
            for hh in range(2):
                for r in range(16):
                    koff = (r & 3) + 8 * (r >> 2) + 32 * hh
                    x = S(p, rb, hh, r)
                    out.append(Ins(f"v_cmp_ge_i32 vcc, {koff}, {T[2]}", "valu", [T[2]], ["vcc"]))
                    out.append(Ins(f"v_cndmask_b32 {x}, {x}, {T[3]}, vcc", "valu", [x, T[3], "vcc"], [x]))
        out.append(label(lab_nomask))
        for rb in range(RB):
            mx, dl, al = T[4], T[5], T[6]
            vals = Sregs(p, rb, 0) + Sregs(p, rb, 1)
            out.append(Ins(f"v_max3_f32 {mx}, {vals[0]}, {vals[1]}, {vals[2]}", "valu", vals[:3], [mx]))
            for i in range(3, 31, 2):
                out.append(Ins(f"v_max3_f32 {mx}, {mx}, {vals[i]}, {vals[i + 1]}", "valu", [mx, vals[i], vals[i + 1]], [mx]))
            out.append(Ins(f"v_max_f32 {mx}, {mx}, {vals[31]}", "valu", [mx, vals[31]], [mx]))
            # the row's other 32 keys sit in lane ^ 32
            out.append(Ins(f"v_mov_b32 {T[7]}, {mx}", "valu", [mx], [T[7]]))
            out.append(nop(2))
            out.append(Ins(f"v_permlane32_swap_b32 {mx}, {T[7]}", "perm", [mx, T[7]], [mx, T[7]]))
            out.append(Ins(f"v_max_f32 {mx}, {mx}, {T[7]}", "valu", [mx, T[7]], [mx]))
            if first:
                out.append(Ins(f"v_mov_b32 {dl}, {mx}", "valu", [mx], [dl]))
            else:
                out.append(Ins(f"v_max_f32 {dl}, {mx}, 0", "valu", [mx], [dl]))
            out.append(Ins(f"v_add_f32 {MRUN[rb]}, {MRUN[rb]}, {dl}", "valu", [MRUN[rb], dl], [MRUN[rb]]))
            if first:
                for e in range(2):
                    out.append(Ins(f"v_mov_b32 {LA[rb][e]}, 0", "valu", [], [LA[rb][e]]))
                    out.append(Ins(f"v_mov_b32 {LB[rb][e]}, 0", "valu", [], [LB[rb][e]]))
                for d in range(DB if zero_o else 0):
                    for i in range(16):
                        out.append(Ins(f"v_accvgpr_write_b32 {O(rb, d, i)}, 0", "valu", [], [O(rb, d, i)]))
                if self.msum and zero_o:
                    for i in range(16):
                        out.append(Ins(f"v_accvgpr_write_b32 {Lacc(rb, i)}, 0", "valu", [], [Lacc(rb, i)]))
            else:
                out.append(Ins(f"v_sub_f32 {al}, 0, {dl}", "valu", [dl], [al]))
                out.append(Ins(f"v_exp_f32 {al}, {al}", "trans", [al], [al]))
                out.append(nop(1))
                for e in range(2):
                    out.append(Ins(f"v_mul_f32 {LA[rb][e]}, {LA[rb][e]}, {al}", "valu", [LA[rb][e], al], [LA[rb][e]]))
                    out.append(Ins(f"v_mul_f32 {LB[rb][e]}, {LB[rb][e]}, {al}", "valu", [LB[rb][e], al], [LB[rb][e]]))
                for d in range(DB):
                    for i in range(0, 16, 2):
                        a0, a1 = O(rb, d, i), O(rb, d, i + 1)
                        out.append(Ins(f"v_accvgpr_read_b32 {T[8]}, {a0}", "valu", [a0], [T[8]]))
                        out.append(Ins(f"v_accvgpr_read_b32 {T[9]}, {a1}", "valu", [a1], [T[9]]))
                        out.append(Ins(f"v_mul_f32 {T[8]}, {T[8]}, {al}", "valu", [T[8], al], [T[8]]))
                        out.append(Ins(f"v_mul_f32 {T[9]}, {T[9]}, {al}", "valu", [T[9], al], [T[9]]))
                        out.append(Ins(f"v_accvgpr_write_b32 {a0}, {T[8]}", "valu", [T[8]], [a0]))
                        out.append(Ins(f"v_accvgpr_write_b32 {a1}, {T[9]}", "valu", [T[9]], [a1]))
                if self.msum:
                    for i in range(0, 16, 2):
                        a0, a1 = Lacc(rb, i), Lacc(rb, i + 1)
                        out.append(Ins(f"v_accvgpr_read_b32 {T[8]}, {a0}", "valu", [a0], [T[8]]))
                        out.append(Ins(f"v_accvgpr_read_b32 {T[9]}, {a1}", "valu", [a1], [T[9]]))
                        out.append(Ins(f"v_mul_f32 {T[8]}, {T[8]}, {al}", "valu", [T[8], al], [T[8]]))
                        out.append(Ins(f"v_mul_f32 {T[9]}, {T[9]}, {al}", "valu", [T[9], al], [T[9]]))
                        out.append(Ins(f"v_accvgpr_write_b32 {a0}, {T[8]}", "valu", [T[8]], [a0]))
                        out.append(Ins(f"v_accvgpr_write_b32 {a1}, {T[9]}", "valu", [T[9]], [a1]))
            for x in vals:
                out.append(Ins(f"v_sub_f32 {x}, {x}, {dl}", "valu", [x, dl], [x]))
            for i in range(16):
                out.append(Ins(f"v_sub_f32 {MS(rb, i)}, 0, {MRUN[rb]}", "valu", [MRUN[rb]], [MS(rb, i)]))
        out.append(Ins("", "pseudo", sem=("rebase_end", p, first, tile_rel)))
        out.append(nop(4))
        return out

    def exph0_plain(self, p, tile_rel):
        return self.softmax_items(p, 0, tile_rel, "B")

    # ---- the whole statement --------------------------------------------------------------------------------------
    def program(self):
        R = self.R
        L = lambda s: f"L_{s}_%="
        out = []
        # per-lane address tables (written to LDS once per kernel by the C++ side: koff[6], voff[3][2] as 12 dwords)
        for i, r in enumerate(KOFF + [VOFF[d][h] for d in range(DB) for h in range(2)]):
            out.append(Ins(f"ds_read_b32 {r}, %[tab] offset:{256 * i}", "ds", [], [r], ("ds_tab", i)))
        out.append(salu(f"s_mov_b32 {S_J}, 0", [], [S_J], ("j0",)))
        out.append(salu(f"s_mov_b32 {S_WOFF}, %[woff]", [], [S_WOFF]))
        out.append(salu(f"s_mov_b64 {S_KPTR}, %[kptr]", [], [S_KPTR], ("kinit",)))
        out.append(salu(f"s_mov_b64 {S_VPTR}, %[vptr]", [], [S_VPTR], ("vinit",)))
        out.append(salu(f"s_sub_u32 {S_NM1}, %[n], 1", [], [S_NM1, "scc"], ("nm1",)))
        out.append(salu(f"s_sub_u32 {S_NMR}, %[n], {R}", [], [S_NMR, "scc"], ("nmr",)))
        out.append(salu(f"s_sub_u32 {S_NMR1}, %[n], {R - 1}", [], [S_NMR1, "scc"], ("nmr1",)))
        for rb in range(RB):
            out.append(Ins(f"v_mov_b32 {MRUN[rb]}, 0", "valu", [], [MRUN[rb]]))
        if self.msum:                         # the all-ones A fragment (bf16 1.0 pairs)
            out.append(Ins(f"v_mov_b32 {T[0]}, 0x3f803f80", "valu", [], [T[0]]))
            for r in ONESregs():
                out.append(Ins(f"v_accvgpr_write_b32 {r}, {T[0]}", "valu", [T[0]], [r]))
        out.append(lgkm(0))
        # tile 0 (and whatever else of this item's first tiles was requested by the previous item / the kernel prologue)
        out.append(vmw(0))
        out.append(barrier())
        out.append(Ins("", "pseudo", sem=("item_begin",)))
        for ks in range(KS):
            for hh in range(2):
                out.append(self.ds_k(0, ks, hh, 0))
        mf = [self.qk(0, g & 1, (g >> 1) & 1, g >> 2, 0, c_zero=(g >> 2) == 0) for g in range(4 * KS)]
        gaps = [[] for _ in mf]
        k1 = []
        if self.early:                       # K'(1) fragments: re-use the registers of fragments already consumed
            for ks in range(KS):
                for hh in range(2):
                    g_free = 4 * ks + 2 * hh + 1             # last MFMA reading fragment (ks, hh)
                    tgt = min(g_free + 1, len(mf) - 1)
                    (gaps[tgt] if g_free + 1 <= len(mf) - 1 else k1).append(self.ds_k(1 % R, ks, hh, 1))
        if self.fast_ends:                   # O = 0 inside the gaps of the first tile's QK^T (nothing else to do there)
            zero = [Ins(f"v_accvgpr_write_b32 {O(rb, d, i)}, 0", "valu", [], [O(rb, d, i)]) for rb in range(RB) for d in range(DB) for i in range(16)]
            if self.msum:
                zero += [Ins(f"v_accvgpr_write_b32 {Lacc(rb, i)}, 0", "valu", [], [Lacc(rb, i)]) for rb in range(RB) for i in range(16)]
            for i, z in enumerate(zero):
                gaps[i * len(mf) // len(zero)].append(z)
        if self.first_fast:
            # The first tile's scores are taken relative to m = 0 -- no row max, no rebase -- when |q'| max|k'(0)| stays inside the
            # window the lazy softmax allows around its reference anyway (the same test as every later tile's, against m = 0): the
            # -m splats and the row sums are zeroed in the head's gaps, the rebase is jumped over.
            dec0 = [Ins(f"v_readlane_b32 {S_KN}, %[kn], 0", "valu", [], [S_KN], ("readkn",)),
                    Ins(f"v_mul_f32 {T[0]}, %[qn0], {S_KN}", "valu", [S_KN], [T[0]]),
                    Ins(f"v_mul_f32 {T[1]}, %[qn1], {S_KN}", "valu", [S_KN], [T[1]]),
                    Ins(f"v_max_f32 {T[0]}, {T[0]}, {T[1]}", "valu", [T[0], T[1]], [T[0]]),
                    Ins(f"v_cmp_lt_f32 vcc, 0x{THR_BITS:08x}, {T[0]}", "valu", [T[0]], ["vcc"])]
            for i, x in enumerate(dec0):
                gaps[i].append(x)
            z = [Ins(f"v_mov_b32 {MS(rb, i)}, 0", "valu", [], [MS(rb, i)]) for rb in range(RB) for i in range(16)]
            z += [Ins(f"v_mov_b32 {a}, 0", "valu", [], [a]) for rb in range(RB) for a in LA[rb] + LB[rb]]
            for i, x in enumerate(z):
                gaps[i * len(mf) // len(z)].append(x)
        out += self.weave([], mf, gaps) + k1
        if self.first_fast:
            out.append(branch("s_cbranch_vccz", L("first_fast"), ("br_first",)))
        out += self.rebase(0, True, 0, zero_o=not self.fast_ends)
        if self.first_fast:
            out.append(label(L("first_fast")))
            out.append(nop(16))               # (the jump's way here: the head's last XDL writes of S' retire before the first exp)
        out += self.exph0_plain(0, 0)
        # ---- the unrolled steps ----
        for c in range(R):
            out.append(label(L(f"step{c}")))
            if c == R - 1:
                out.append(salu(f"s_cmp_eq_u32 {S_J}, {S_NM1}", [S_J, S_NM1], ["scc"]))
                out.append(branch("s_cbranch_scc1", L("tail"), ("br_tailstep",)))
            if self.bina:
                out += self.steady(self.phase_a(c, L(f"slow{c}"), with_boundary=True))
            else:
                out += self.steady(self.boundary(c))
                out += self.steady(self.phase_a(c, L(f"slow{c}")))
            out += self.steady(self.phase_b(c))
            out.append(salu(f"s_add_u32 {S_J}, {S_J}, 1", [S_J], [S_J, "scc"], ("jinc",)))
            if c == R - 1:
                out.append(branch("s_branch", L("step0"), ("br_always",)))
        # ---- slow continuations: O += V'(j) P(j) alone, rebase for tile j + 1, its hh = 0 half, on to the next step ----
        for c in range(R):
            out.append(label(L(f"slow{c}")))
            out.append(lgkm(0))
            out += self.phase_b(c, plain=True)
            out += self.rebase((c + 1) & 1, False, 1, uid=f"_c{c}")
            out += self.exph0_plain((c + 1) & 1, 1)
            out.append(salu(f"s_add_u32 {S_J}, {S_J}, 1", [S_J], [S_J, "scc"], ("jinc",)))
            out.append(branch("s_branch", L(f"step{(c + 1) % R}"), ("br_always",)))
        # ---- last step ----
        out.append(label(L("tail")))
        out.append(lgkm(0))
        out += self.tail_step()
        out.append(nop(16))                   # the last XDL writes of O retire before the epilogue's v_accvgpr_read
        for rb in range(RB):
            if self.msum:
                # every lane of a row's pair holds the row's whole sum (both key halves went through the matrix instruction): half of
                # it, so that the kernel's l(lane) + l(lane ^ 32) stays what it is for the VALU sums
                out.append(Ins(f"v_accvgpr_read_b32 {T[0]}, {Lacc(rb, 0)}", "valu", [Lacc(rb, 0)], [T[0]]))
                out.append(Ins(f"v_mul_f32 %[lr{rb}], 0.5, {T[0]}", "valu", [T[0]], []))
                # (r05: the running max goes out with the row sum here as well -- this branch used to `continue` past the move below, the
                #  statement's m output stayed unwritten and the dh = 64 instances' LSE lacked m: outputs right, every gradient wrong)
                out.append(Ins(f"v_mov_b32 %[mr{rb}], {MRUN[rb]}", "valu", [MRUN[rb]], []))
                continue
            out.append(Ins(f"v_add_f32 {LA[rb][0]}, {LA[rb][0]}, {LA[rb][1]}", "valu", [LA[rb][0], LA[rb][1]], [LA[rb][0]]))
            out.append(Ins(f"v_add_f32 {LB[rb][0]}, {LB[rb][0]}, {LB[rb][1]}", "valu", [LB[rb][0], LB[rb][1]], [LB[rb][0]]))
            out.append(Ins(f"v_add_f32 %[lr{rb}], {LA[rb][0]}, {LB[rb][0]}", "valu", [LA[rb][0], LB[rb][0]], []))
            out.append(Ins(f"v_mov_b32 %[mr{rb}], {MRUN[rb]}", "valu", [MRUN[rb]], []))
        out.append(Ins("", "pseudo", sem=("item_end",)))
        if not self.carry:
            return insert_waits(out, carry=False)
        # the state every phase B ends in (the K' reads of its last gaps): from a dry run of one phase B
        return insert_waits(out, carry=True)


# ------------------------------------------------------------------------------------------------------------------
# lgkmcnt waits: LDS reads return in order; a use of a read's destination needs lgkmcnt(<= reads issued after it)
# ------------------------------------------------------------------------------------------------------------------
def insert_waits(prog, carry=True, seed_state=None):
    """lgkmcnt waits by a straight-line pass.  Conditional branches keep the outstanding reads (their taken targets -- the rare
    paths -- start drained: the program puts an explicit lgkmcnt(0) there).  The step labels are merge points: the K' fragment
    reads that phase B requests in its last gaps stay in flight across them (carry), which is consistent on every path because
    each phase B ends in the same state and every other way into a step label (the item's head, the slow continuations) arrives
    with nothing outstanding -- a counted wait then simply does not wait.  The simulation checks the dynamic order."""
    out = []
    pend = []                    # outstanding reads, oldest first: sets of destination registers
    step_state = seed_state      # outstanding reads at a step label (the state every phase B ends in)

    def need(regs):
        idx = -1
        for i, d in enumerate(pend):
            if any(r in d for r in regs):
                idx = i
        return idx

    prev = None
    for ins in prog:
        if ins.kind == "label":
            is_step = ins.label.startswith("L_step")
            fallthrough = not (prev is not None and prev.kind == "branch" and prev.text.startswith("s_branch"))
            if is_step and carry:
                if fallthrough and step_state is not None and pend and [sorted(x) for x in pend] != [sorted(x) for x in step_state]:
                    raise CheckError(f"outstanding LDS reads at {ins.label} differ between paths")
                if fallthrough and pend:
                    step_state = [set(x) for x in pend]
                elif step_state is not None:
                    pend = [set(x) for x in step_state]
            else:
                if pend and fallthrough:
                    out.append(lgkm(0))
                pend = []
            out.append(ins)
            prev = ins
            continue
        if ins.kind == "branch":
            uncond = ins.text.startswith("s_branch")
            to_step = ins.target.startswith("L_step")
            rare = ins.target.startswith("L_slow") or ins.target.startswith("L_tail")     # (those blocks start with lgkmcnt(0))
            if pend and ((uncond and not (to_step and carry)) or (not uncond and not rare)):
                out.append(lgkm(0))
                pend = []
            if uncond and to_step and carry and pend:
                if step_state is None:
                    step_state = [set(x) for x in pend]
                elif [sorted(x) for x in pend] != [sorted(x) for x in step_state]:
                    raise CheckError(f"outstanding LDS reads at the jump to {ins.target} differ from the step label's state")
            out.append(ins)
            prev = ins
            continue
        if ins.kind == "wait" and ins.sem[0] == "lgkm":
            n = ins.sem[1]
            del pend[:max(0, len(pend) - n)]
            out.append(ins)
            prev = ins
            continue
        touched = list(ins.rd) + list(ins.wr)
        i = need(touched)
        if i >= 0:
            left = min(len(pend) - 1 - i, 15)            # (the counter has four bits)
            out.append(lgkm(left))
            del pend[:len(pend) - left]
        if ins.kind == "ready":
            continue
        out.append(ins)
        if ins.kind == "ds":
            pend.append(set(ins.wr))
        prev = ins
    if pend:
        out.append(lgkm(0))
    if carry and seed_state is None and step_state is not None:
        return insert_waits(prog, carry=True, seed_state=step_state)      # second pass: the head's way into step 0 knows the state too
    return out


# ------------------------------------------------------------------------------------------------------------------
# checker: dynamic order of one wave, typed registers, counters, wait states
# ------------------------------------------------------------------------------------------------------------------
class CheckError(Exception):
    pass


class Sim:
    def __init__(self, gen, prog, n_tiles, rebase_at=(), has_tail=False, items=2, first_rebase=True):
        self.g, self.prog, self.n, self.rebase_at, self.has_tail, self.items = gen, prog, n_tiles, set(rebase_at), has_tail, items
        self.first_rebase = first_rebase or not gen.first_fast      # does tile 0 take the rebase (its bound leaves the window)?
        self.labels = {ins.label: i for i, ins in enumerate(prog) if ins.kind == "label"}
        self.count = 0

    def fail(self, msg, ins=None):
        raise CheckError(f"{msg}" + (f"   at: {ins.text}" if ins is not None else ""))

    def run(self):
        R = self.g.R
        # persistent state across items: LDS slots, DMA queue
        self.kslot = [None] * R
        self.vslot = [None] * R
        self.vmq = []            # outstanding DMA pieces of this wave: (ring, slot, piece, tile)  tile = (item, t)
        self.got = {}            # (ring, slot) -> set of pieces landed for the pending tile
        self.hist = []           # last instructions (for wait-state rules): list of Ins
        stats = {"instr": 0, "mfma": 0}
        # kernel prologue's requests of the first item: K'(0..R-1), V'(0..R-2)
        for t in range(R):
            for i in range(NP):
                self.vmq.append(("K", t % R, i, (0, t)))
        for t in range(R - 1):
            for i in range(NP):
                self.vmq.append(("V", t % R, i, (0, t)))
        kptr, vptr = (0, R), (0, R - 1)
        for item in range(self.items):
            kptr, vptr = self.run_item(item, kptr, vptr, stats)
        return stats

    # -- helpers --
    def land(self, ent):
        ring, slot, piece, tile = ent
        key = (ring, slot)
        cur = self.pending_tile.get(key)
        if cur is None or cur[0] != tile:
            self.pending_tile[key] = (tile, set())
        self.pending_tile[key][1].add(piece)

    def run_item(self, item, kptr, vptr, stats):
        g, R, n = self.g, self.g.R, self.n
        regs = {}                # physical register -> (tag tuple, need_reads)
        pend_ds = []             # outstanding LDS reads: list of register lists
        pending_regs = set()
        sc = {}                  # scalar state: j, j1
        self.pending_tile = getattr(self, "pending_tile", {})
        msver = [0, 0]
        o_done = {(rb, d): set() for rb in range(RB) for d in range(DB)}
        l_done = {rb: set() for rb in range(RB)}
        sums = {}
        packed_from = {}
        exps = {}
        flags = {"scc": 0, "vcc": 0}
        ksel_done = vsel_done = False
        pc = 0
        prog = self.prog
        rebased_tiles = set()
        steps = 0

        def setreg(r, tag, need):
            old = regs.get(r)
            if old is not None and old[1] > 0:
                self.fail(f"register {r} overwritten with {old[1]} reads of {old[0]} outstanding", cur)
            regs[r] = [tag, need]

        def usereg(r, expect_prefix, cur_ins):
            v = regs.get(r)
            if v is None:
                self.fail(f"register {r} read but never written (expected {expect_prefix})", cur_ins)
            if r in pending_regs:
                self.fail(f"register {r} read before its LDS read was waited for", cur_ins)
            if tuple(v[0][:len(expect_prefix)]) != tuple(expect_prefix):
                self.fail(f"register {r} holds {v[0]}, expected {expect_prefix}", cur_ins)
            v[1] -= 1
            return v[0]

        while True:
            if pc >= len(prog):
                self.fail("fell off the end of the program")
            cur = prog[pc]
            pc += 1
            if cur.kind == "label":
                continue
            stats["instr"] += 1
            self.check_wait_states(cur)
            sem = cur.sem
            k = cur.kind
            # any access to a register with an outstanding LDS read
            for r in list(cur.rd) + list(cur.wr):
                if r in pending_regs and k != "ds":
                    self.fail(f"{r} touched while its LDS read is outstanding", cur)
            if k == "mfma":
                stats["mfma"] += 1
            if sem is None:
                self.hist.append(cur)
                continue
            op = sem[0]
            if op == "nop":
                pass
            elif op == "lgkm":
                keep = sem[1]
                while len(pend_ds) > keep:
                    for r in pend_ds.pop(0):
                        pending_regs.discard(r)
            elif op == "vm":
                keep = sem[1]
                while len(self.vmq) > keep:
                    self.land(self.vmq.pop(0))
            elif op == "barrier":
                # every wave has waited for its share of the same tiles: a tile whose three pieces of THIS wave have landed
                # is complete; the slots named by the requests that follow are free (checked at the request)
                for key, (tile, pieces) in list(self.pending_tile.items()):
                    if len(pieces) == NP:
                        (self.kslot if key[0] == "K" else self.vslot)[key[1]] = tile
                        del self.pending_tile[key]
            elif op == "item_begin":
                for t in range(R):
                    if self.kslot[t] != (item, t):
                        self.fail(f"item {item}: K slot {t} holds {self.kslot[t]} at item begin")
                for t in range(R - 1):
                    if self.vslot[t] != (item, t):
                        self.fail(f"item {item}: V slot {t} holds {self.vslot[t]} at item begin")
            elif op == "j0":
                sc["j"] = 0
            elif op == "j1":
                sc["j1"] = sc["j"] + 1
            elif op == "jinc":
                sc["j"] += 1
                steps += 1
            elif op in ("nm1", "nmr", "nmr1", "readkn", "ds_tab"):
                if op == "ds_tab":
                    pend_ds.append(list(cur.wr))
                    pending_regs.update(cur.wr)
            elif op == "cmp_nmr":
                flags["scc_eq"] = sc["j"] == n - R
            elif op == "cmp_nmr1":
                flags["scc_eq"] = sc["j"] == n - R + 1
            elif op == "kinit":
                sc["kptr"] = kptr
            elif op == "vinit":
                sc["vptr"] = vptr
            elif op == "ksel":
                if flags["scc_eq"]:
                    if sc["kptr"] != (item, n):
                        self.fail(f"K stream switches items at position {sc['kptr']}")
                    sc["kptr"] = (item + 1, 0)
                    ksel_done = True
            elif op == "vsel":
                if flags["scc_eq"]:
                    if sc["vptr"] != (item, n):
                        self.fail(f"V stream switches items at position {sc['vptr']}")
                    sc["vptr"] = (item + 1, 0)
                    vsel_done = True
            elif op == "kadv":
                if sem[1] == 0:
                    sc["kptr"] = (sc["kptr"][0], sc["kptr"][1] + 1)
            elif op == "vadv":
                if sem[1] == 0:
                    sc["vptr"] = (sc["vptr"][0], sc["vptr"][1] + 1)
            elif op == "dma":
                ring, slot, piece = sem[1], sem[2], sem[3]
                tile = sc["kptr"] if ring == "K" else sc["vptr"]
                # the slot's current tile must be consumed: K'(t) by phase A of step t - 1, V'(t) by phase B of step t
                held = (self.kslot if ring == "K" else self.vslot)[slot]
                if held is not None and held[0] == item:
                    last_needed = held[1] - 1 if ring == "K" else held[1]
                    if last_needed >= sc["j"]:
                        self.fail(f"DMA into {ring} slot {slot} at step {sc['j']} while tile {held} is still needed", cur)
                if piece == 0:
                    (self.kslot if ring == "K" else self.vslot)[slot] = None
                self.vmq.append((ring, slot, piece, tile))
            elif op == "ds_k":
                slot, ks, hh, rel = sem[1], sem[2], sem[3], sem[4]
                t = sc.get("j", 0) + rel
                held = self.kslot[slot]
                tag = ("K", t, ks, hh) if held == (item, t) else ("Kjunk", held)
                if held != (item, t) and t < n:
                    self.fail(f"K slot {slot} holds {held}, the read wants tile {t} of item {item}", cur)
                for r in cur.wr:
                    setreg(r, tag, RB if t < n else 0)
                pend_ds.append(list(cur.wr))
                pending_regs.update(cur.wr)
            elif op == "ds_v":
                slot, sl, d, half = sem[1], sem[2], sem[3], sem[4]
                t = sc["j"]
                if self.vslot[slot] != (item, t):
                    self.fail(f"V slot {slot} holds {self.vslot[slot]}, the read wants tile {t} of item {item}", cur)
                for r in cur.wr:
                    setreg(r, ("V", t, sl, d), RB)
                pend_ds.append(list(cur.wr))
                pending_regs.update(cur.wr)
            elif op == "qk":
                p, rb, hh, ks, rel, cz = sem[1:]
                t = sc.get("j", 0) + rel
                for r in Kregs(ks, hh):
                    usereg(r, ("K", t, ks, hh), cur)
                if ks == 0:
                    for r in Sregs(p, rb, hh):
                        old = regs.get(r)
                        if old is not None and old[1] > 0:
                            self.fail(f"S' register {r} overwritten with consumers outstanding: {old}", cur)
                        regs[r] = [("SA", t, rb, hh, 1, msver[rb]), 1]
                else:
                    for r in Sregs(p, rb, hh):
                        v = usereg(r, ("SA", t, rb, hh, ks), cur)
                        regs[r] = [("SA", t, rb, hh, ks + 1, v[5]), 1]
                if ks == KS - 1:
                    for i, r in enumerate(Sregs(p, rb, hh)):
                        regs[r] = [("S", t, rb, hh, i, "raw", regs[r][0][5]), 1]
            elif op == "exp":
                p, rb, hh, r_, rel = sem[1:]
                t = sc.get("j", 0) + rel
                x = S(p, rb, hh, r_)
                v = usereg(x, ("S", t, rb, hh, r_, "raw"), cur)
                if v[6] != msver[rb]:
                    self.fail(f"{x}: score relative to an old running max (version {v[6]} vs {msver[rb]})", cur)
                regs[x] = [("S", t, rb, hh, r_, "exp"), 1 if (g.dot_sum or g.msum) else 2]         # one add (or none: summed as a packed word), one pack
                exps[(t, rb)] = exps.get((t, rb), 0) + 1
            elif op == "add":
                p, rb, hh, r_ = sem[1:]
                x = S(p, rb, hh, r_)
                v = regs.get(x)
                if v is None or v[0][5] != "exp":
                    self.fail(f"{x} summed before its exp", cur)
                t = v[0][1]
                if ("sum", x, t) in sums:
                    self.fail(f"{x} of tile {t} summed twice", cur)
                sums[("sum", x, t)] = 1
                sums[(t, rb)] = sums.get((t, rb), 0) + 1
                v[1] -= 1
            elif op == "add2":
                p, rb, hh, r_ = sem[1:]
                for rr in (r_, r_ + 1):
                    x = S(p, rb, hh, rr)
                    v = regs.get(x)
                    if v is None or v[0][5] != "exp":
                        self.fail(f"{x} summed before its exp", cur)
                    t = v[0][1]
                    if ("sum", x, t) in sums:
                        self.fail(f"{x} of tile {t} summed twice", cur)
                    sums[("sum", x, t)] = 1
                    sums[(t, rb)] = sums.get((t, rb), 0) + 1
                    v[1] -= 1
            elif op == "pack":
                p, rb, hh, tt, w = sem[1:]
                r0 = 8 * tt + 2 * w
                a, b = S(p, rb, hh, r0), S(p, rb, hh, r0 + 1)
                ta = regs.get(a)
                tb = regs.get(b)
                if ta is None or tb is None or ta[0][5] != "exp" or tb[0][5] != "exp" or ta[0][1] != tb[0][1]:
                    self.fail(f"pack of {a}, {b} before both exps", cur)
                ta[1] -= 1
                tb[1] -= 1
                setreg(P(p, rb, hh, tt, w), ("P", ta[0][1], rb, hh, tt, w), DB + (1 if (g.dot_sum or g.msum) else 0))
                if g.dot_sum:
                    packed_from[P(p, rb, hh, tt, w)] = (a, b, ta[0][1])
            elif op == "dotsum":
                p, rb, hh, tt, w = sem[1:]
                wreg = P(p, rb, hh, tt, w)
                a, b, t = packed_from.get(wreg, (None, None, None))
                usereg(wreg, ("P", t, rb, hh, tt, w), cur)
                for x in (a, b):
                    if ("sum", x, t) in sums:
                        self.fail(f"{x} of tile {t} summed twice", cur)
                    sums[("sum", x, t)] = 1
                    sums[(t, rb)] = sums.get((t, rb), 0) + 1
            elif op == "pv":
                p, sl, d, rb = sem[1:]
                t = sc["j"]
                for r in Vregs(sl, d):
                    usereg(r, ("V", t, sl, d), cur)
                for w, r in enumerate(Pregs(p, rb, sl >> 1, sl & 1)):
                    usereg(r, ("P", t, rb, sl >> 1, sl & 1, w), cur)
                if (t, sl) in o_done[(rb, d)]:
                    self.fail(f"O[{rb}][{d}] accumulates tile {t} slab {sl} twice", cur)
                if o_done[(rb, d)] and max(o_done[(rb, d)]) > (t, sl):
                    self.fail(f"O[{rb}][{d}] accumulates out of order", cur)
                o_done[(rb, d)].add((t, sl))
            elif op == "pl":
                p, sl, rb = sem[1:]
                t = sc["j"]
                for w, r in enumerate(Pregs(p, rb, sl >> 1, sl & 1)):
                    usereg(r, ("P", t, rb, sl >> 1, sl & 1, w), cur)
                if (t, sl) in l_done[rb] or (l_done[rb] and max(l_done[rb]) > (t, sl)):
                    self.fail(f"l[{rb}] accumulates tile {t} slab {sl} twice or out of order", cur)
                l_done[rb].add((t, sl))
            elif op == "rebase_begin":
                p, first, rel = sem[1:]
                t = sc.get("j", 0) + rel
                for rb in range(RB):
                    for hh in range(2):
                        for i, r in enumerate(Sregs(p, rb, hh)):
                            v = regs.get(r)
                            if v is None or tuple(v[0][:6]) != ("S", t, rb, hh, i, "raw"):
                                self.fail(f"rebase of tile {t}: {r} holds {v}", cur)
                # no P V of an earlier tile may be outstanding: O must hold exactly tiles < t
                for (rb, d), done in o_done.items():
                    if len(done) != 4 * t:
                        self.fail(f"rebase for tile {t} with O[{rb}][{d}] at {len(done)} slabs", cur)
                if g.msum:
                    for rb, done in l_done.items():
                        if len(done) != 4 * t:
                            self.fail(f"rebase for tile {t} with l[{rb}] at {len(done)} slabs", cur)
                rebased_tiles.add(t)
            elif op == "rebase_end":
                p, first, rel = sem[1:]
                for rb in range(RB):
                    msver[rb] += 1
                    for hh in range(2):
                        for r in Sregs(p, rb, hh):
                            regs[r][0] = regs[r][0][:6] + (msver[rb],)
            elif op in ("br_need", "br_tail", "br_nomask", "br_tailstep", "br_always", "br_first"):
                taken = False
                if op == "br_always":
                    taken = True
                elif op == "br_first":
                    taken = not self.first_rebase
                elif op == "br_need":
                    taken = (sc["j"] + 1) in self.rebase_at
                elif op == "br_tail":
                    taken = self.has_tail and sc["j"] + 1 == n - 1
                elif op == "br_nomask":
                    taken = True       # (the mask block is plain VALU; walk around it)
                elif op == "br_tailstep":
                    taken = sc["j"] == n - 1
                if taken:
                    pc = self.labels[cur.target]
            elif op == "item_end":
                break
            else:
                self.fail(f"unknown semantic {sem}", cur)
            self.hist.append(cur)
            if len(self.hist) > 64:
                del self.hist[:32]
        # ---- end-of-item checks ----
        for rb in range(RB):
            for t in range(n):
                if g.msum:
                    if exps.get((t, rb), 0) != 32 or sum(1 for x in l_done[rb] if x[0] == t) != 4:
                        self.fail(f"item {item}: tile {t} row block {rb}: {exps.get((t, rb), 0)} exps, row-sum slabs {sorted(x for x in l_done[rb] if x[0] == t)}")
                elif exps.get((t, rb), 0) != 32 or sums.get((t, rb), 0) != 32:
                    self.fail(f"item {item}: tile {t} row block {rb}: {exps.get((t, rb), 0)} exps, {sums.get((t, rb), 0)} sums (32 each expected)")
            for d in range(DB):
                if len(o_done[(rb, d)]) != 4 * n:
                    self.fail(f"item {item}: O[{rb}][{d}] got {len(o_done[(rb, d)])} of {4 * n} slabs")
        if pend_ds:
            self.fail("LDS reads outstanding at the end of the item")
        if not ksel_done or not vsel_done:
            self.fail("the DMA streams did not pass to the next item")
        want = ({0} if self.first_rebase else set()) | {t for t in self.rebase_at if 0 < t < n} | ({n - 1} if self.has_tail else set())
        if rebased_tiles != want:
            self.fail(f"rebased tiles {sorted(rebased_tiles)} != {sorted(want)}")
        return sc["kptr"], sc["vptr"]

    # -- manual wait states (cdna4 ISA 4.5; LLVM GCNHazardRecognizer gfx940 rows), conservative: every instruction = 1 state --
    def check_wait_states(self, cur):
        if cur.kind in ("label", "pseudo"):
            return
        dist = 0
        for prev in reversed(self.hist):
            if prev.kind == "pseudo":
                continue
            need = 0
            if prev.kind == "mfma" and cur.kind != "mfma":
                if set(prev.wr) & (set(cur.rd) | set(cur.wr)):
                    need = 12                                    # XDL (8 pass) write -> VALU / LDS read or write (ISA: 11)
            elif prev.kind == "mfma" and cur.kind == "mfma":
                ov = set(prev.wr) & set(cur.rd)
                if ov and not (set(prev.wr) == set(cur.wr) and ov == set(prev.wr)):
                    need = 12                                    # XDL write -> XDL read as A / B (or a different C)
            elif prev.kind in ("valu", "trans", "perm") and cur.kind == "mfma":
                if set(prev.wr) & set(cur.rd):
                    need = 2                                     # VALU write -> XDL read
            elif prev.kind == "trans" and cur.kind in ("valu", "perm"):
                if set(prev.wr) & set(cur.rd):
                    need = 1                                     # transcendental forwarding
            elif prev.kind in ("valu", "trans") and cur.kind == "perm":
                if set(prev.wr) & set(cur.rd):
                    need = 2
            elif prev.kind == "salu" and cur.kind == "dma":
                if "m0" in prev.wr:
                    need = 1
            if need and dist < need:
                raise CheckError(f"wait states: '{cur.text}' needs {need} states after '{prev.text}', has {dist}")
            dist += prev.sem[1] if (prev.kind == "nop") else 1
            if dist > 20:
                break


def stats_of(gen, prog):
    """static listing: instructions per MFMA gap of the steady-state step copies"""
    lines = []
    in_step = None
    cnt = {}
    for ins in prog:
        if ins.kind == "label":
            nm = ins.label
            in_step = nm if nm.startswith("L_step") else None
            if in_step:
                cnt[in_step] = {"mfma": 0, "other": 0, "valu": 0, "trans": 0, "ds": 0, "dma": 0, "salu": 0, "wait": 0}
            continue
        if in_step and ins.kind != "pseudo":
            c = cnt[in_step]
            if ins.kind == "mfma":
                c["mfma"] += 1
            else:
                c["other"] += 1
                kk = ins.kind if ins.kind in c else "salu"
                c[kk] += 1
    for k, c in cnt.items():
        if c["mfma"]:
            lines.append(f"{k}: {c['mfma']} MFMA, {c['other']} other ({c['other'] / c['mfma']:.2f} per gap): " +
                         ", ".join(f"{n} {v}" for v, n in c.items() if v not in ("mfma", "other")))
    return lines


def emit(progs, path):
    """progs: {macro name: program}.  Also the clobber list of the statement (every register the stream names literally)."""
    with open(path, "w") as f:
        f.write("// generated by gen_attn64.py (python3 gen_attn64.py --out gta_attn64_loop.inc) -- do not edit\n")
        for name, prog in progs.items():
            f.write(f"#define {name} \\\n")
            for ins in prog:
                if ins.kind == "pseudo":
                    continue
                f.write(f'    "{ins.text}\\n\\t" \\\n')
            f.write('    ""\n')
        # Per head dimension: what the statement TAKES in the accumulator file (the Q' fragments, written by the compiled prologue) and what it
        # LEAVES there (O) are operands of the statement by register -- "+{a[124:127]}"(qfr[0][0]), "={a[28:43]}"(oacc[0][0]) -- so hipcc knows
        # what lives where on both sides (r05; before, literal v_accvgpr_write / _read statements beside it did the hand-off behind its back);
        # every other register the stream names is clobbered.  a[0:AB-1] stay hipcc's across the statement.
        for dh in (96, 64):
            configure(dh)
            taken = [(f"qfr[{rb}][{ks}]", _qb() + 4 * (rb * KS + ks), 4) for rb in range(RB) for ks in range(KS)]
            left = [(f"oacc[{rb}][{d}]", AB + 16 * (rb * DB + d), 16) for rb in range(RB) for d in range(DB)]
            named = {a0 + i for _, a0, n in taken + left for i in range(n)}
            regs = [f"v{i}" for i in range(32, 256)] + [f"a{i}" for i in range(AB, 256) if i not in named] + [f"s{i}" for i in range(84, 97)]
            f.write(f"#define GTA_ATTN64_CLOBBERS_{dh} \\\n    " + ", ".join(f'"{r}"' for r in regs) + ', "m0", "vcc", "scc", "memory"\n')
            f.write(f"#define GTA_ATTN64_RESULTS_{dh} \\\n    " + ", ".join(f'"={{a[{a0}:{a0 + n - 1}]}}"({lv})' for lv, a0, n in left) + "\n")
            f.write(f"#define GTA_ATTN64_QFRAGS_{dh} \\\n    " + ", ".join(f'"+{{a[{a0}:{a0 + n - 1}]}}"({lv})' for lv, a0, n in taken) + "\n")


def check_all(gen, prog, verbose=False):
    R = gen.R
    cases = [(R, (), False), (2 * R, (), False), (5 * R, (), False), (2 * R, (1,), False), (3 * R, (2, 5, 2 * R + 1), True),
             (2 * R, (R - 1, R, 2 * R - 1), False), (R, (), True), (3 * R, tuple(range(1, 3 * R)), True)]
    for n, reb, tail in cases:
        for first in ((True, False) if gen.first_fast else (True,)):
            st = Sim(gen, prog, n, reb, tail, first_rebase=first).run()
            if verbose:
                print(f"  ok: n_tiles {n:2d} rebase at {reb} tail {tail} first-tile rebase {first}: {st['instr']} instructions, {st['mfma']} MFMAs (two items)")


BEST = dict(boundary_in_a=True, carry=True, fast_ends=True, dma_spread=True)      # measured r03: profiles/r03/attn64_dev_log.md


def production_variants():
    """what gta_attn64_loop.inc holds, as (macro, dh, options): V0 = the shipped schedule, V1 = the same instructions un-interleaved
    (GTA_ATTN64_VARIANT=1), for dh = 96 and for dh = 64 (there with the row sums on the matrix pipe: msum)"""
    return [("GTA_ATTN64_LOOP_V0", 96, dict(BEST)), ("GTA_ATTN64_LOOP_V1", 96, dict(sched=False)),
            # (r05: at dh = 64 tile 0 takes the lazy path when its bound allows -- first_fast -- as the 32-row kernel does for key sides of more
            #  than one tile: these kernels run near the part's clock, where a cycle is a cycle, and the two kernels keep rounding P alike)
            ("GTA_ATTN64_LOOP64_V0", 64, dict(BEST, msum=True, first_fast=True)),
            ("GTA_ATTN64_LOOP64_V1", 64, dict(sched=False, msum=True, first_fast=True))]


def build_program(dh, kw, R=4, kread_early=True, check=True, verbose=False):
    """the stream for head dimension dh (the module's layout constants are set for it while it is built and simulated)"""
    configure(dh)
    gen = Gen(R=R, kread_early=kread_early, **kw)
    prog = gen.program()
    if check and not kw.get("ablate"):
        check_all(gen, prog, verbose)
    return gen, prog


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--ring", type=int, default=4)
    ap.add_argument("--no-early-k", action="store_true")
    ap.add_argument("--plain", action="store_true", help="no interleaving: every phase's fillers in front of its MFMAs")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--dev", action="store_true", help="also emit the development variants")
    a = ap.parse_args()
    progs = {}
    best = BEST
    variants = production_variants() if not a.plain else [(n, dh, dict(sched=False)) for n, dh, _ in production_variants()]
    if a.dev:     # development variants (gta_fwd64.hip -DGTA_ATTN64_DEV, GTA_ATTN64_VARIANT=n): schedules and timing-only ablations
        for dh, pre in ((96, "GTA_ATTN64_LOOP"), (64, "GTA_ATTN64_LOOP64")):
            variants += [(f"{pre}_V2", dh, dict(best) if dh == 64 else dict(best, first_fast=True)),      # (dh = 64: V2 = the VALU row sums)
                         (f"{pre}_V3", dh, dict(best, ablate=("novalu",))),
                         (f"{pre}_V4", dh, dict(best, ablate=("nods",))),
                         (f"{pre}_V5", dh, dict(best, ablate=("nodma", "nobar"))),
                         (f"{pre}_V6", dh, dict(best, ablate=("novalu", "nods", "nodma", "nobar"))),
                         (f"{pre}_V7", dh, dict(best, ablate=("nomfma",))),
                         (f"{pre}_V8", dh, dict(best, boundary_in_a=False))]
    for name, dh, kw in variants:
        gen, prog = build_program(dh, kw, R=a.ring, kread_early=not a.no_early_k, verbose=a.verbose)
        if a.verbose:
            print(name, kw)
            print("\n".join(stats_of(gen, prog)))
        progs[name] = prog
    if a.out:
        emit(progs, a.out)
