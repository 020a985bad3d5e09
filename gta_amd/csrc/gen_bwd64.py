#!/usr/bin/env python3
"""gen_bwd64.py -- generator of the tile loop of gta_bwd_dkv64_kernel (gta_bwd.hip) as one gfx950 assembly stream.

dK' and dV' of GTA attention's backward (autograd over source/utils/gta.py:92-279 + source/layers.py:202-211; the kernel split is
described in gta_bwd.hip) with 64 KEYS per wave and ONE wave per SIMD: a workgroup of four waves owns 256 keys of one (b, h) and
streams the (b, h)'s Q'' / dO~ tile images (64 query rows each) through a ring of four LDS stages; the K' / V' fragments of the wave's
64 keys and its dK'^T / dV'^T accumulators stay in registers for the whole walk.  Every streamed fragment -- a row fragment of Q'' or
dO~ for S and dP, a transpose-read of them for dK' and dV' -- feeds the wave's TWO 32-key blocks, i.e. two MFMAs: half the LDS bytes per
MFMA of the 32-keys-per-wave kernel.  hipcc cannot place ~480 live registers nor interleave one wave's softmax with its own MFMAs
(r04: the compiled form of this tiling ran 539 us against 377), so the loop is emitted here and included as one asm statement.

Per query tile j (stage j % 4) and 32-row block qb the wave computes, for its key blocks kb = 0, 1:
    S  = Q'' K'^T - lse2      (6 k-steps; the accumulator STARTS at -lse2: the pre-pass writes the statistics negated)
    dP = dO~ V'^T - D
    P  = exp2(S),  dS = P dP  (fp32, then bf16 pairs packed IN PLACE into the low halves of the S / dP registers)
    dV'^T[d] += dO~^T P,  dK'^T[d] += Q''^T dS      (3 channel blocks x 2 k-steps; A operands by ds_read_b64_tr_b16)
The MFMA stream is a sequence of GROUPS of four MFMAs that share one 8-register fragment slot (two ds_read_b128, or four transpose
reads); slots form a ring of four and are loaded three groups ahead.  An iteration (tile j) runs four segments of six groups,
    D: dV/dK of (j-1, qb 0)    A: S/dP of (j, qb 0)    B: dV/dK of (j-1, qb 1)    C: S/dP of (j, qb 1)
so that each block's softmax has two segments of MFMAs to hide under: that of (j-1, qb 1) -- produced by C of the previous iteration --
runs beside D and A, that of (j, qb 0) beside B and C.  The wait for tile j's DMA and the workgroup barrier sit at the iteration's top;
the request for tile j + 2 follows them (the stage it overwrites, (j - 2) % 4, was last read by B of iteration j - 1).

What this file guarantees on the CPU (tests/test_host_logic.py): the emitted stream, EXECUTED on the one-wave simulator of isa_model.py
(registers, LDS, LDS-DMA under vmcnt, every read poisoned until its wait) for several tile counts, leaves dK'^T / dV'^T equal to a numpy
model of the same arithmetic; the ISA's manual wait states hold on the executed order.  GPU tests compare the kernel with the
32-keys-per-wave one.

Register map
  a[  0: 95]  dK'^T[kb][d] 16 each      a[ 96:191]  dV'^T[kb][d]      a[192:239]  K' fragments [kb][ks] 4 each
  a[240:255]  V' fragments [0][0..3]    v[ 24: 55]  V' fragments, the other eight
  v[ 56:119]  SE0 = S[0] dP[0] S[1] dP[1] of row block 0, 16 each (bf16 P / dS pairs end up in registers 0..7 of each)
  v[120:183]  SE1, row block 1          v[184:215]  fragment slots, 4 x 8
  v[216:231]  -lse2 of the block's rows (C operand of its first S MFMAs)   v[232:247]  -D
  v[248:255]  temporaries               v[0:23], s[0:19] and what the statement names as operands: the compiler's
"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

from isa_model import (LANES, Asm, CheckError, Wave, XI, as_u32, bf16_rne, bf16_to_f32, check_wait_states, regs, rtext, u2f)

KS, DB, KB, CHP = 6, 3, 2, 12
ROW = CHP * 16                     # 192 bytes per image row
IMG = 64 * ROW                     # 12288
STAGE = 2 * IMG                    # Q'' image | dO~ image of one 64-row tile
R = 4                              # ring stages
RING = R * STAGE
OFF_STATS = RING                   # R x 512 bytes: [-lse2 (64 floats) | -D (64 floats)] of the tile in the stage
SL = 16 * ROW                      # 16 rows further on: the second k-step of a transpose-read
HALF = 32 * ROW                    # row block 1
HI_BASE = 2 * STAGE                # the second set of lane-offset registers points here (immediates stay below 64 KiB)
PF = 3                             # groups of prefetch distance (fragment slots: 4)


def A(first, n=1):
    return regs("a", first, n)


def V(first, n=1):
    return regs("v", first, n)


DK = [[A((kb * DB + d) * 16, 16) for d in range(DB)] for kb in range(KB)]
DV = [[A(96 + (kb * DB + d) * 16, 16) for d in range(DB)] for kb in range(KB)]
KFR = [[A(192 + (kb * KS + ks) * 4, 4) for ks in range(KS)] for kb in range(KB)]
VFR = [[(A(240 + 4 * (kb * KS + ks), 4) if kb * KS + ks < 4 else V(24 + 4 * (kb * KS + ks - 4), 4)) for ks in range(KS)] for kb in range(KB)]
SE = [{"s": [V(56 + 64 * q + 32 * kb, 16) for kb in range(KB)], "e": [V(56 + 64 * q + 32 * kb + 16, 16) for kb in range(KB)]} for q in range(2)]
SLOT = [V(184 + 8 * i, 8) for i in range(4)]
INIT_L, INIT_D = V(216, 16), V(232, 16)
TMP = V(248, 8)
CLOBBER_V = list(range(24, 256))
# SGPRs of the stream
S_J, S_N, S_JF, S_T0, S_T1, S_T2 = "s20", "s21", "s22", "s23", "s24", "s25"
S_QP, S_SP, S_M = ["s26", "s27"], ["s28", "s29"], "s30"
S_WIMG, S_WST = "s31", "s33"                    # this wave's share of a stage's images (wave * 6144) / of its statistics (256 (wave & 1))
CLOBBER_S = [20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 33]
# statement operands: VGPRs (lane offsets, LDS addresses incl. the ring's base; *_h = the same + HI_BASE), SGPRs
VOPS = ("koffl", "koff4", "koff5", "voff00", "voff01", "voff20", "voff21",
        "koffl_h", "koff4_h", "koff5_h", "voff00_h", "voff01_h", "voff20_h", "voff21_h", "lane16", "lane4", "stoff")
SOPS = ("q_lo", "q_hi", "st_lo", "st_hi", "kv_lo", "kv_hi", "n", "wave", "ring", "stats", "side")


def op(name):
    return f"%[{name}]"


def swz(r, u):
    return (u + ((r >> 2) & 3)) % CHP


# ------------------------------------------------------------------------------------------------------------------
# the stream
# ------------------------------------------------------------------------------------------------------------------
class Group:
    """four MFMAs behind one fragment slot"""

    def __init__(self, loads, mfmas):
        self.loads, self.mfmas = loads, mfmas          # callables (asm, slot registers) -> emit


class Gen:
    def __init__(self, sched=True, by_cost=True):
        self.sched = sched
        self.by_cost = by_cost

    # ---- addresses ----
    @staticmethod
    def stage_imm(st):
        return (st & 1) * STAGE, ("_h" if st >= 2 else "")

    def frag_loads(self, st, qb, ks):
        """row fragments of Q'' (slot[0:4]) and dO~ (slot[4:8]) of k-step ks: lane (row l31, unit 2 ks + lh)"""
        base, h = self.stage_imm(st)
        if ks < 4:
            reg, imm = op("koffl" + h), 32 * ks
        else:
            reg, imm = op(f"koff{ks}" + h), 0

        def emit(a, slot):
            a.ds_read(128, slot[0:4], reg, base + qb * HALF + imm)
            a.ds_read(128, slot[4:8], reg, base + IMG + qb * HALF + imm)
        return emit

    def tr_loads(self, st, qb, t, d, rt=None):
        """transpose-reads of Q'' (slot[0:4]) and dO~ (slot[4:8]): channel block d, k-step t.  rt: registers holding the four lane offsets
        of a stage known only at run time (the statement's tail)"""
        base, h = self.stage_imm(st) if rt is None else (0, "")
        names = ("voff00", "voff01") if d < 2 else ("voff20", "voff21")
        imm = base + qb * HALF + t * SL + (64 if d == 1 else 0)

        def emit(a, slot):
            for img in range(2):
                for hf in range(2):
                    reg = op(names[hf] + h) if rt is None else rt[(0 if d < 2 else 2) + hf]
                    dst = slot[4 * img + 2 * hf:4 * img + 2 * hf + 2]
                    a.ds_read_tr(dst, reg, imm + img * IMG)
        return emit

    # ---- MFMA groups ----
    def s_group(self, st, q, ks):
        se = SE[q]

        def mf(a, slot):
            for kb in range(KB):
                a.mfma(se["s"][kb], slot[0:4], KFR[kb][ks], INIT_L if ks == 0 else se["s"][kb])
                a.mfma(se["e"][kb], slot[4:8], VFR[kb][ks], INIT_D if ks == 0 else se["e"][kb])
        return Group(self.frag_loads(st, q, ks), mf)

    def d_group(self, st, q, t, d, rt=None):
        se = SE[q]

        def mf(a, slot):
            for kb in range(KB):
                a.mfma(DV[kb][d], slot[4:8], se["s"][kb][4 * t:4 * t + 4], DV[kb][d])       # dV'^T += dO~^T P
                a.mfma(DK[kb][d], slot[0:4], se["e"][kb][4 * t:4 * t + 4], DK[kb][d])       # dK'^T += Q''^T dS
        return Group(self.tr_loads(st, q, t, d, rt), mf)

    def s_segment(self, st, q):
        return [self.s_group(st, q, ks) for ks in range(KS)]

    def d_segment(self, st, q, rt=None):
        return [self.d_group(st, q, t, d, rt) for t in range(2) for d in range(DB)]

    # ---- softmax of one row block: closures, one instruction each, in deadline order (k-step 0's registers first) ----
    def softmax(self, q):
        se = SE[q]
        ops = []
        for t in range(2):
            for kb in range(KB):
                s, e = se["s"][kb], se["e"][kb]
                rr = range(8 * t, 8 * t + 8)
                for r in rr:
                    ops.append(lambda a, r=r, s=s: a.v_exp_f32(s[r], s[r]))
                for r in rr:
                    ops.append(lambda a, r=r, s=s, e=e: a.v_mul_f32(e[r], s[r], e[r]))
                for i in range(4 * t, 4 * t + 4):            # in place: pair i of (2 i, 2 i + 1) -> register i (already consumed)
                    ops.append(lambda a, i=i, s=s: a.v_cvt_pk_bf16_f32(s[i], s[2 * i], s[2 * i + 1]))
                for i in range(4 * t, 4 * t + 4):
                    ops.append(lambda a, i=i, e=e: a.v_cvt_pk_bf16_f32(e[i], e[2 * i], e[2 * i + 1]))
        return ops

    def init_loads(self, st, qb):
        """the rows' -lse2 / -D: register r of a lane half <-> row 32 qb + 8 (r >> 2) + 4 lh + (r & 3)"""
        ops = []
        for g in range(4):
            ops.append(lambda a, g=g: a.ds_read(128, INIT_L[4 * g:4 * g + 4], op("stoff"), st * 512 + (32 * qb + 8 * g) * 4))
            ops.append(lambda a, g=g: a.ds_read(128, INIT_D[4 * g:4 * g + 4], op("stoff"), st * 512 + 256 + (32 * qb + 8 * g) * 4))
        return ops

    # ---- LDS-DMA of a tile (this wave's 6 KiB of the 24-KiB stage + its share of the statistics): 7 operations ----
    def dma_addr(self, a):
        """source pointers of tile S_JF (clamped to the last one: the count of operations per iteration is what the counted waits rely on)"""
        a.s_mul_i32(S_T0, S_JF, STAGE)
        a.s_add_u32(S_QP[0], op("q_lo"), S_T0)
        a.s_addc_u32(S_QP[1], op("q_hi"), 0)
        a.s_add_u32(S_QP[0], S_QP[0], S_WIMG)
        a.s_addc_u32(S_QP[1], S_QP[1], 0)
        a.s_lshl_b32(S_T0, S_JF, 9)
        a.s_add_u32(S_SP[0], op("st_lo"), S_T0)
        a.s_addc_u32(S_SP[1], op("st_hi"), 0)
        a.s_add_u32(S_SP[0], S_SP[0], S_WST)
        a.s_addc_u32(S_SP[1], S_SP[1], 0)
        a.s_add_u32(S_M, op("ring"), S_WIMG)
        a.s_add_u32(S_T2, op("stats"), S_WST)

    def dma_ops(self, st):
        """the requests themselves, one closure per instruction (they ride in the gaps of an iteration's first MFMA groups)"""
        ops = []

        def m0(reg, off):
            ops.append(lambda a: a.add(f"s_add_u32 m0, {reg}, {off}", "salu", [reg], ["m0", "scc"], ("s_add_m0", reg, off)))
            ops.append(lambda a: a.nop(1))
        for g in range(2):
            if g == 1:
                ops.append(lambda a: a.s_add_u32(S_QP[0], S_QP[0], 4096))
                ops.append(lambda a: a.s_addc_u32(S_QP[1], S_QP[1], 0))
            m0(S_M, st * STAGE + 4096 * g)
            for p in range(4 if g == 0 else 2):
                ops.append(lambda a, p=p: a.add(f"global_load_lds_dwordx4 {op('lane16')}, {rtext(S_QP)}" + (f" offset:{1024 * p}" if p else ""), "dma",
                                               ["m0", op("lane16")] + S_QP, [], ("global_load_lds", op("lane16"), S_QP, 1024 * p)))
        m0(S_T2, st * 512)
        ops.append(lambda a: a.add(f"global_load_lds_dword {op('lane4')}, {rtext(S_SP)}", "dma", ["m0", op("lane4")] + S_SP, [],
                                   ("global_load_lds_dword", op("lane4"), S_SP, 0)))
        return ops

    def dma_tile(self, a, st):
        self.dma_addr(a)
        for c in self.dma_ops(st):
            c(a)

    # ---- weaving: MFMA groups with the loads PF groups ahead and a share of the VALU list in their gaps ----
    COST = {"trans": 16, "valu": 5, "ds": 10, "dma": 12, "salu": 2, "nop": 4, "wait": 0, "dsw": 10}      # (8 .. 24 / 6 .. 16 for the first and third: no difference on the GPU)

    def weave(self, a, groups, valu, first_slot, extra=None, preloaded=0):
        """groups: the MFMA groups in order; valu: [(first group, last group, [closures])]: each list is dealt over the gaps of its group range
        BY ISSUE COST (a matrix instruction leaves ~28 cycles of issue behind it: a v_exp_f32 takes 16 of them, a plain VALU instruction 5, an LDS
        read ~10), counting what the gaps already hold; extra: {group index: [closures]} issued with that group's loads.  Loads of groups
        [0, preloaded) are already out.  Returns the slot index after the last group."""
        extra = extra or {}
        n = len(groups)
        cost = lambda x: self.COST.get(x.kind, 4)
        for g in range(min(PF, n)):
            if g >= preloaded:
                groups[g].loads(a, SLOT[(first_slot + g) % 4])
        # what every gap holds anyway: the loads of the group PF ahead (and the extras), over the gaps behind the group's MFMAs 1..3
        mf, gap = [], {}
        for g, grp in enumerate(groups):
            body = Asm()
            grp.mfmas(body, SLOT[(first_slot + g) % 4])
            mf.append(list(body.out))
            ld = Asm()
            if g + PF < n:
                groups[g + PF].loads(ld, SLOT[(first_slot + g + PF) % 4])
            for c in extra.get(g, []):
                c(ld)
            lds = list(ld.out)
            for m in range(4):
                gap[(g, m)] = []
                if lds and m >= 1:                                     # (a slot is re-loaded once the group PF + 1 back has issued its MFMAs)
                    k = (len(lds) + (3 - m)) // (4 - m)
                    gap[(g, m)] += lds[:k]
                    del lds[:k]
        for g0, g1, lst in valu:
            tmp = Asm()
            for c in lst:
                c(tmp)
            ops = list(tmp.out)
            gaps = [(g, m) for g in range(g0, g1 + 1) for m in range(4)]
            if not self.sched or not self.by_cost:
                for i, x in enumerate(ops):
                    gap[gaps[i * len(gaps) // len(ops)]].append(x)
                continue
            fixed = [sum(cost(x) for x in gap[k]) for k in gaps]
            target = (sum(fixed) + sum(cost(x) for x in ops)) / len(gaps)
            gi, cum = 0, fixed[0]
            for x in ops:
                c = cost(x)
                while gi < len(gaps) - 1 and cum + c / 2 > target * (gi + 1):
                    gi += 1
                    cum += fixed[gi]
                gap[gaps[gi]].append(x)
                cum += c
        for g in range(n):
            if self.sched:
                for m, ins in enumerate(mf[g]):
                    a.raw(ins)
                    for x in gap[(g, m)]:
                        a.raw(x)
            else:
                for ins in mf[g]:
                    a.raw(ins)
                for m in range(4):
                    for x in gap[(g, m)]:
                        a.raw(x)
        return (first_slot + n) % 4

    # ---- the statement ----
    def head(self, a):
        """once: accumulators, this wave's K' / V' fragments (global loads straight into their registers), the first two tiles' requests"""
        a.s_mov_b32(S_N, op("n"))
        a.s_mul_i32(S_WIMG, op("wave"), IMG // 2)                      # 6144 bytes of a stage per wave
        a.s_and_b32(S_T0, op("wave"), 1)
        a.s_lshl_b32(S_WST, S_T0, 8)
        # lane offsets of the image layout without the ring's base: K' / V' rows 32 kb + l31 of the tile in global memory
        for i, nm in enumerate(("koffl", "koff4", "koff5")):
            a.v_sub_u32(TMP[i], op(nm), op("ring"))
        a.s_mov_b32(S_QP[0], op("kv_lo"))                              # K' image; V' image 12 KiB on: offsets beyond the 13-bit field ride in a second base
        a.s_mov_b32(S_QP[1], op("kv_hi"))
        a.s_add_u32(S_SP[0], op("kv_lo"), IMG)
        a.s_addc_u32(S_SP[1], op("kv_hi"), 0)
        for kb in range(KB):
            if kb == 1:
                for pr in (S_QP, S_SP):
                    a.s_add_u32(pr[0], pr[0], HALF)
                    a.s_addc_u32(pr[1], pr[1], 0)
            for ks in range(KS):
                reg, imm = (TMP[0], 32 * ks) if ks < 4 else (TMP[ks - 3], 0)
                a.global_load(4, KFR[kb][ks], reg, S_QP, imm)
                a.global_load(4, VFR[kb][ks], reg, S_SP, imm)
        for acc in [x for kb in range(KB) for d in range(DB) for x in (DK[kb][d], DV[kb][d])]:
            for r in acc:
                a.v_accvgpr_write_b32(r, 0)
        a.s_mov_b32(S_JF, 0)
        self.dma_tile(a, 0)
        a.s_cmp("gt", "u32", S_N, 1)
        a.add(f"s_cselect_b32 {S_JF}, 1, 0", "salu", ["scc"], [S_JF], ("s_cselect_b32", S_JF, 1, 0))
        self.dma_tile(a, 1)

    def top(self, a, st):
        """iteration top of tile j (stage st): its DMA has landed everywhere; the pointers of tile min(j + 2, n - 1), whose request follows"""
        a.waitcnt(vm=7)
        a.barrier()
        a.s_add_u32(S_JF, S_J, 2)
        a.s_sub_u32(S_T1, S_N, 1)
        a.add(f"s_min_u32 {S_JF}, {S_JF}, {S_T1}", "salu", [S_JF, S_T1], [S_JF, "scc"], ("s_min_u32", S_JF, S_JF, S_T1))
        self.dma_addr(a)

    def program(self, n_static=None):
        a = Asm()
        self.head(a)
        # ---- tile 0: A, C (nothing to add to dK / dV yet) ----
        a.s_mov_b32(S_J, 0)
        self.top(a, 0)
        for c in self.dma_ops(2):
            c(a)
        for c in self.init_loads(0, 0):
            c(a)
        slot = 0
        sm0 = self.softmax(0)
        g = self.s_segment(0, 0) + self.s_segment(0, 1)
        slot = self.weave(a, g, [(7, 11, sm0)], slot, extra={2: self.init_loads(0, 1)})
        a.s_mov_b32(S_J, 1)
        # ---- tiles 1 .. n - 1: four copies, one per ring stage; the walk enters at stage 1.  An iteration starts with the transpose-reads of
        # its first three groups (tile j - 1: nothing to wait for), THEN waits for tile j and meets the other waves; the requests for tile j + 2
        # ride in the gaps of its first groups ----
        a.label("L_top_%=")
        for c in (1, 2, 3, 0):
            a.s_cmp("ge", "u32", S_J, S_N)
            a.branch("s_cbranch_scc1", "L_tail_%=")
            prev = (c - 1) % R
            g = self.d_segment(prev, 0) + self.s_segment(c, 0) + self.d_segment(prev, 1) + self.s_segment(c, 1)
            for i in range(PF):
                g[i].loads(a, SLOT[(slot + i) % 4])
            self.top(a, c)
            dma = self.dma_ops((c + 2) % R)
            sm1, sm0 = self.softmax(1), self.softmax(0)
            slot = self.weave(a, g, [(0, 11, sm1), (13, 23, sm0)], slot, preloaded=PF,
                              extra={0: self.init_loads(c, 0), 1: dma[:8], 2: dma[8:], 10: self.init_loads(c, 1)})
            a.s_add_u32(S_J, S_J, 1)
        a.branch("s_branch", "L_top_%=")
        # ---- after the last tile: its D and B; the tile's stage is (S_J - 1) % 4, known at run time ----
        a.label("L_tail_%=")
        a.waitcnt(vm=0, lgkm=0)
        a.s_sub_u32(S_T0, S_J, 1)
        a.s_and_b32(S_T0, S_T0, 3)
        a.s_mul_i32(S_T0, S_T0, STAGE)
        rt = TMP[0:4]
        for i, nm in enumerate(("voff00", "voff01", "voff20", "voff21")):
            a.v_add_u32(rt[i], S_T0, op(nm))
        g = self.d_segment(0, 0, rt) + self.d_segment(0, 1, rt)
        self.weave(a, g, [(0, 5, self.softmax(1))], slot)
        # what the epilogue's d trans_coeff terms need of the raw k / v rows -- element 3 and 7 of the se3 chunks -- is what the K' / V'
        # fragments hold in those places (B_k's last row is (0, 0, 0, 1)): dwords 1 and 3 of fragment (kb, ks) of lane (key, lh) = chunk
        # 2 ks + lh, handed over lane by lane: side[((which * 2 + kb) * 3 + ks) * 2 + i][lane]
        a.v_add_u32(TMP[4], op("side"), op("lane4"))
        for which, fr in enumerate((KFR, VFR)):
            for kb in range(KB):
                for ks in range(3):
                    for i in range(2):
                        a.ds_write(32, TMP[4], [fr[kb][ks][1 + 2 * i]], (((which * 2 + kb) * 3 + ks) * 2 + i) * 256)
        a.waitcnt(lgkm=0)
        a.pseudo("end")
        return auto_waits(a.out)


# ------------------------------------------------------------------------------------------------------------------
# dQ: 64 query rows per wave (gta_bwd_dq64_kernel).  The same machinery with the roles turned: the wave's Q'' / dO~ fragments (row blocks
# rb = 0, 1 of 32 rows) are stationary, the (b, h)'s K' / V' tile images are streamed; per 32-key half hh of a tile
#     S^T = K' Q''^T - lse2,  dP^T = V' dO~^T - D   (the seeds are per-lane constants here: a lane is a query row),
#     dS^T = exp2(S^T) dP^T  (bf16 pairs in place of dP^T),   dQ'^T[d] += K'^T dS^T   (K'^T by transpose-reads, 2 k-steps of 16 keys)
# Groups: S/dP k-step ks = K' and V' row fragments of the half (one slot) -> 4 MFMAs (2 row blocks x S, dP); dQ: two transposed operands
# (t, d) per slot -> 4 MFMAs (2 row blocks each).  An iteration: dQ of (j-1, hh 0) [3 groups] | S/dP of (j, 0) [6] | dQ of (j-1, 1) [3] |
# S/dP of (j, 1) [6].  Key tiles in whole (Tk % 64 == 0: the kernel keeps the compiled form otherwise).
# Registers: a[0:95] dQ'^T[rb][d], a[96:143] Q'' fragments [rb][ks], a[144:191] dO~ fragments; v[56:183] the two S / dP sets (per half),
# v[184:215] slots, v[216:247] -lse2 splats [rb], v[24:55] -D splats [rb].
# ------------------------------------------------------------------------------------------------------------------
DQ = [[A((rb * DB + d) * 16, 16) for d in range(DB)] for rb in range(2)]
QF = [[A(96 + (rb * KS + ks) * 4, 4) for ks in range(KS)] for rb in range(2)]
DOF = [[A(144 + (rb * KS + ks) * 4, 4) for ks in range(KS)] for rb in range(2)]
QINIT_L = [V(216 + 16 * rb, 16) for rb in range(2)]
QINIT_D = [V(24 + 16 * rb, 16) for rb in range(2)]
Q_VOPS = ("koffl", "koff4", "koff5", "voff00", "voff01", "voff20", "voff21",
          "koffl_h", "koff4_h", "koff5_h", "voff00_h", "voff01_h", "voff20_h", "voff21_h", "lane16", "lrow4")
Q_SOPS = ("kv_lo", "kv_hi", "qi_lo", "qi_hi", "st_lo", "st_hi", "n", "wave", "ring")
DQ_PAIRS = (((0, 0), (0, 1)), ((0, 2), (1, 0)), ((1, 1), (1, 2)))       # the (t, d) operands of a half's three dQ groups


class GenDQ(Gen):
    def kv_loads(self, st, hh, ks):
        """row fragments of K' (slot[0:4]) and V' (slot[4:8]) of key half hh, k-step ks"""
        base, h = self.stage_imm(st)
        reg, imm = (op("koffl" + h), 32 * ks) if ks < 4 else (op(f"koff{ks}" + h), 0)

        def emit(a, slot):
            a.ds_read(128, slot[0:4], reg, base + hh * HALF + imm)
            a.ds_read(128, slot[4:8], reg, base + IMG + hh * HALF + imm)
        return emit

    def ktr_loads(self, st, hh, pair, rt=None):
        """K'^T operands (t, d) of the pair: slot[0:4], slot[4:8]"""
        base, h = self.stage_imm(st) if rt is None else (0, "")

        def emit(a, slot):
            for i, (t, d) in enumerate(pair):
                names = ("voff00", "voff01") if d < 2 else ("voff20", "voff21")
                imm = base + hh * HALF + t * SL + (64 if d == 1 else 0)
                for hf in range(2):
                    reg = op(names[hf] + h) if rt is None else rt[(0 if d < 2 else 2) + hf]
                    a.ds_read_tr(slot[4 * i + 2 * hf:4 * i + 2 * hf + 2], reg, imm)
        return emit

    def s_group(self, st, hh, ks):
        se = SE[hh]

        def mf(a, slot):
            for rb in range(2):
                a.mfma(se["s"][rb], slot[0:4], QF[rb][ks], QINIT_L[rb] if ks == 0 else se["s"][rb])      # S^T  = K' Q''^T - lse2
                a.mfma(se["e"][rb], slot[4:8], DOF[rb][ks], QINIT_D[rb] if ks == 0 else se["e"][rb])     # dP^T = V' dO~^T - D
        return Group(self.kv_loads(st, hh, ks), mf)

    def s_segment(self, st, hh):
        return [self.s_group(st, hh, ks) for ks in range(KS)]

    def d_segment(self, st, hh, rt=None):
        se = SE[hh]
        out = []
        for pair in DQ_PAIRS:
            def mf(a, slot, pair=pair):
                for i, (t, d) in enumerate(pair):
                    for rb in range(2):
                        a.mfma(DQ[rb][d], slot[4 * i:4 * i + 4], se["e"][rb][4 * t:4 * t + 4], DQ[rb][d])   # dQ'^T += K'^T dS^T
            out.append(Group(self.ktr_loads(st, hh, pair, rt), mf))
        return out

    def softmax(self, hh):
        se = SE[hh]
        ops = []
        for t in range(2):
            for rb in range(2):
                s, e = se["s"][rb], se["e"][rb]
                rr = range(8 * t, 8 * t + 8)
                for r in rr:
                    ops.append(lambda a, r=r, s=s: a.v_exp_f32(s[r], s[r]))
                for r in rr:
                    ops.append(lambda a, r=r, s=s, e=e: a.v_mul_f32(e[r], s[r], e[r]))
                for i in range(4 * t, 4 * t + 4):
                    ops.append(lambda a, i=i, e=e: a.v_cvt_pk_bf16_f32(e[i], e[2 * i], e[2 * i + 1]))
        return ops

    def dma_addr(self, a):
        a.s_mul_i32(S_T0, S_JF, STAGE)
        a.s_add_u32(S_QP[0], op("kv_lo"), S_T0)
        a.s_addc_u32(S_QP[1], op("kv_hi"), 0)
        a.s_add_u32(S_QP[0], S_QP[0], S_WIMG)
        a.s_addc_u32(S_QP[1], S_QP[1], 0)
        a.s_add_u32(S_M, op("ring"), S_WIMG)

    def dma_ops(self, st):
        return Gen.dma_ops(self, st)[:-3]                              # (no statistics with a key tile)

    def head(self, a):
        a.s_mov_b32(S_N, op("n"))
        a.s_mul_i32(S_WIMG, op("wave"), IMG // 2)
        for i, nm in enumerate(("koffl", "koff4", "koff5")):
            a.v_sub_u32(TMP[i], op(nm), op("ring"))
        a.s_mov_b32(S_QP[0], op("qi_lo"))                              # this wave's 64 rows: Q'' image; dO~ image 12 KiB on
        a.s_mov_b32(S_QP[1], op("qi_hi"))
        a.s_add_u32(S_SP[0], op("qi_lo"), IMG)
        a.s_addc_u32(S_SP[1], op("qi_hi"), 0)
        for rb in range(2):
            if rb == 1:
                for pr in (S_QP, S_SP):
                    a.s_add_u32(pr[0], pr[0], HALF)
                    a.s_addc_u32(pr[1], pr[1], 0)
            for ks in range(KS):
                reg, imm = (TMP[0], 32 * ks) if ks < 4 else (TMP[ks - 3], 0)
                a.global_load(4, QF[rb][ks], reg, S_QP, imm)
                a.global_load(4, DOF[rb][ks], reg, S_SP, imm)
        # the rows' -lse2 / -D (the wave's 128 statistics: -lse2 of rows 0..63, then -D), splat over the 16 registers of a C operand
        st = [op("st_lo"), op("st_hi")]
        a.s_mov_b32(S_SP[0], st[0])
        a.s_mov_b32(S_SP[1], st[1])
        for rb in range(2):
            a.global_load(1, [TMP[4 + rb]], op("lrow4"), S_SP, 128 * rb)
            a.global_load(1, [TMP[6 + rb]], op("lrow4"), S_SP, 256 + 128 * rb)
        for acc in [x for rb in range(2) for x in DQ[rb]]:
            for r in acc:
                a.v_accvgpr_write_b32(r, 0)
        for rb in range(2):
            for r in QINIT_L[rb]:
                a.v_mov_b32(r, TMP[4 + rb])
            for r in QINIT_D[rb]:
                a.v_mov_b32(r, TMP[6 + rb])
        a.s_mov_b32(S_JF, 0)
        self.dma_tile(a, 0)
        a.s_cmp("gt", "u32", S_N, 1)
        a.add(f"s_cselect_b32 {S_JF}, 1, 0", "salu", ["scc"], [S_JF], ("s_cselect_b32", S_JF, 1, 0))
        self.dma_tile(a, 1)

    def top(self, a, st):
        a.waitcnt(vm=6)
        a.barrier()
        a.s_add_u32(S_JF, S_J, 2)
        a.s_sub_u32(S_T1, S_N, 1)
        a.add(f"s_min_u32 {S_JF}, {S_JF}, {S_T1}", "salu", [S_JF, S_T1], [S_JF, "scc"], ("s_min_u32", S_JF, S_JF, S_T1))
        self.dma_addr(a)

    def program(self):
        a = Asm()
        self.head(a)
        a.s_mov_b32(S_J, 0)
        self.top(a, 0)
        for c in self.dma_ops(2):
            c(a)
        slot = 0
        g = self.s_segment(0, 0) + self.s_segment(0, 1)
        slot = self.weave(a, g, [(7, 11, self.softmax(0))], slot)
        a.s_mov_b32(S_J, 1)
        a.label("L_top_%=")
        for c in (1, 2, 3, 0):
            a.s_cmp("ge", "u32", S_J, S_N)
            a.branch("s_cbranch_scc1", "L_tail_%=")
            prev = (c - 1) % R
            g = self.d_segment(prev, 0) + self.s_segment(c, 0) + self.d_segment(prev, 1) + self.s_segment(c, 1)
            for i in range(PF):
                g[i].loads(a, SLOT[(slot + i) % 4])
            self.top(a, c)
            dma = self.dma_ops((c + 2) % R)
            slot = self.weave(a, g, [(0, 8, self.softmax(1)), (10, 17, self.softmax(0))], slot, preloaded=PF, extra={1: dma[:7], 2: dma[7:]})
            a.s_add_u32(S_J, S_J, 1)
        a.branch("s_branch", "L_top_%=")
        a.label("L_tail_%=")
        a.waitcnt(vm=0, lgkm=0)
        a.s_sub_u32(S_T0, S_J, 1)
        a.s_and_b32(S_T0, S_T0, 3)
        a.s_mul_i32(S_T0, S_T0, STAGE)
        rt = TMP[0:4]
        for i, nm in enumerate(("voff00", "voff01", "voff20", "voff21")):
            a.v_add_u32(rt[i], S_T0, op(nm))
        g = self.d_segment(0, 0, rt) + self.d_segment(0, 1, rt)
        self.weave(a, g, [(0, 2, self.softmax(1))], slot)
        # the stream ends on its last MFMAs and its results are read by COMPILED code (outputs of the statement): hipcc places no wait states
        # behind inline asm, so the XDL-write -> VALU-read distance of a 16-pass MFMA (18 states) is kept here
        a.nop(20)
        a.pseudo("end")
        return auto_waits(a.out)


def model_dq(kt, vt, qrows, dorows, stats):
    """kt, vt: [n][64][96] (K', V' tiles), qrows, dorows: [64][96] (the wave's rows), stats [128] (-lse2 | -D).  dQ'^T as [rb][96][32]"""
    n = kt.shape[0]
    out = np.zeros((2, 96, 32), np.float64)
    for j in range(n):
        for rb in range(2):
            q, do = qrows[32 * rb:32 * rb + 32].astype(np.float64), dorows[32 * rb:32 * rb + 32].astype(np.float64)
            s = (kt[j].astype(np.float64) @ q.T + stats[None, 32 * rb:32 * rb + 32]).astype(np.float32)             # [key][row]
            e = (vt[j].astype(np.float64) @ do.T + stats[None, 64 + 32 * rb:64 + 32 * rb + 32]).astype(np.float32)
            with np.errstate(all="ignore"):
                p = np.exp2(s.astype(np.float64)).astype(np.float32)
            ds = bf16_to_f32(bf16_rne((p * e).astype(np.float32))).astype(np.float64)
            out[rb] += kt[j].astype(np.float64).T @ ds
    return out


def run_case_dq(prog, n, wave, seed=0):
    rng = np.random.default_rng(seed + 100)

    def rb_(shape, scale):
        return bf16_to_f32(bf16_rne((rng.standard_normal(shape) * scale).astype(np.float32)))
    kt, vt = rb_((n, 64, 96), 0.6), rb_((n, 64, 96), 0.6)
    qrows, dorows = rb_((64, 96), 0.6), rb_((64, 96), 0.6)
    stats = np.concatenate([-(rng.uniform(3.0, 6.0, size=64)), rng.standard_normal(64) * 0.3]).astype(np.float32)
    stats[61:64] = -1e30
    A_KV, A_QI, A_ST = 0x1000000, 0x2000000, 0x3000000
    kvbuf = np.concatenate([np.concatenate([image_of(bf16_rne(kt[j]).astype(np.uint16)), image_of(bf16_rne(vt[j]).astype(np.uint16))]) for j in range(n)])
    qibuf = np.concatenate([image_of(bf16_rne(qrows).astype(np.uint16)), image_of(bf16_rne(dorows).astype(np.uint16))])
    stbuf = stats.view(np.uint8).copy()
    lo = lane_offsets(0)
    lo["lrow4"] = ((np.arange(LANES) & 31) * 4).astype(np.uint32)
    inputs = {op("kv_lo"): A_KV & 0xffffffff, op("kv_hi"): A_KV >> 32, op("qi_lo"): A_QI & 0xffffffff, op("qi_hi"): A_QI >> 32,
              op("st_lo"): A_ST & 0xffffffff, op("st_hi"): A_ST >> 32, op("n"): n, op("wave"): wave, op("ring"): 0}
    w = Wave(inputs=inputs, seed=wave + 5)
    w.vector_inputs = {op(k): lo[k] for k in Q_VOPS}
    for base, arr in ((A_KV, kvbuf), (A_QI, qibuf), (A_ST, stbuf)):
        w.map_buffer(base, arr)
    state = {"barriers": 0}

    def barrier_hook(wv):
        j = state["barriers"]
        state["barriers"] += 1
        if j >= n:
            raise CheckError(f"barrier {j} with {n} tiles")
        st = j % R
        own = slice(wave * 6144, wave * 6144 + 6144)
        src = kvbuf[j * STAGE:(j + 1) * STAGE]
        dst = wv.lds[st * STAGE:(st + 1) * STAGE]
        if not (dst[own] == src[own]).all():
            raise CheckError(f"wave {wave}: its share of key tile {j} has not landed at the tile's barrier")
        dst[:] = src
    w.hooks = {"end": lambda wv: "__end__"}
    w.barrier_hook = barrier_hook
    w.run(prog, checker=check_wait_states)
    if w.vm or w.lgkm:
        raise CheckError("memory operations outstanding at the end of the statement")
    if state["barriers"] != n:
        raise CheckError(f"{state['barriers']} barriers for {n} tiles")
    want = model_dq(kt, vt, qrows, dorows, stats.astype(np.float64))
    for rb in range(2):
        for d in range(DB):
            got = np.stack([u2f(w.v[w.ridx(r)]) for r in DQ[rb][d]])
            for lane in range(LANES):
                jn, h = lane & 31, lane >> 5
                for r in range(16):
                    ch = 32 * d + (r & 3) + 8 * (r >> 2) + 4 * h
                    ref = want[rb][ch, jn]
                    if not abs(float(got[r, lane]) - ref) <= 2e-3 * max(1.0, abs(ref)):
                        raise CheckError(f"wave {wave} n {n}: dQ'[rb {rb}][channel {ch}][row {jn}] = {got[r, lane]}, expected {ref}")
    return w


def check_dq(verbose=False, cases=((1, 0), (2, 1), (5, 3), (6, 2))):
    prog = GenDQ().program()
    stats = {"instructions": sum(1 for x in prog if x.kind not in ("label", "pseudo")), "mfma": sum(1 for x in prog if x.kind == "mfma")}
    for n, wave in cases:
        run_case_dq(prog, n, wave, seed=n)
        if verbose:
            print(f"  ok (dQ): {n} key tiles, wave {wave}")
    return stats


def auto_waits(prog):
    """lgkmcnt waits in front of the first use of every LDS read's destination (LDS operations complete in order); vmcnt waits for the
    global loads of the head.  Labels and branches sit where nothing is outstanding (the generator's iteration tops)."""
    out = []
    lg, vm = [], []
    for ins in prog:
        if ins.kind in ("label", "branch"):
            if lg:
                raise CheckError(f"LDS operations outstanding at {ins.text}")
            out.append(ins)
            continue
        if ins.kind == "wait":
            _, nv, nl = ins.fx
            if nl is not None:
                del lg[:max(0, len(lg) - nl)]
            if nv is not None:
                del vm[:max(0, len(vm) - nv)]
            out.append(ins)
            continue
        touched = set(ins.rd) | set(ins.wr)
        for q, name in ((lg, "lgkm"), (vm, "vm")):
            idx = max((i for i, d in enumerate(q) if d & touched), default=-1)
            if idx >= 0:
                left = min(len(q) - 1 - idx, 15 if name == "lgkm" else 63)
                out.append(XI(f"s_waitcnt {'lgkmcnt' if name == 'lgkm' else 'vmcnt'}({left})", "wait",
                              fx=("waitcnt", left if name == "vm" else None, left if name == "lgkm" else None)))
                del q[:len(q) - left]
        out.append(ins)
        if ins.kind == "ds":
            lg.append(set(ins.wr))
        elif ins.kind == "vmem":
            vm.append(set(ins.wr))
        elif ins.kind == "dma":
            vm.append(set())
    return out


# ------------------------------------------------------------------------------------------------------------------
# numpy model and the simulation
# ------------------------------------------------------------------------------------------------------------------
def image_of(rows_bf16):
    """[64][96] bf16 bits -> the 12288-byte tile image (row-major 16-byte units, per-row rotation swz)"""
    img = np.zeros((64, CHP, 8), np.uint16)
    for r in range(64):
        for u in range(CHP):
            img[r, swz(r, u)] = rows_bf16[r, 8 * u:8 * u + 8]
    return img.reshape(-1).view(np.uint8)


def lane_offsets(ring_base):
    lane = np.arange(LANES)
    l31, lh, p16, g16 = lane & 31, lane >> 5, lane & 15, lane >> 4
    out = {}
    for nm, ks in (("koffl", 0), ("koff4", 4), ("koff5", 5)):
        out[nm] = (l31 * CHP + np.array([swz(int(r), 2 * ks + int(h)) for r, h in zip(l31, lh)])) * 16 + ring_base
    for d, nm in ((0, "voff0"), (2, "voff2")):
        for hf in range(2):
            r = 4 * lh + (p16 >> 2) + 8 * hf
            u = 4 * d + 2 * (g16 & 1) + ((p16 & 3) >> 1)
            out[f"{nm}{hf}"] = (r * CHP + np.array([swz(int(a), int(b)) for a, b in zip(r, u)])) * 16 + (p16 & 1) * 8 + ring_base
    for k in list(out):
        out[k + "_h"] = out[k] + HI_BASE
    out["lane16"], out["lane4"] = lane * 16, lane * 4
    out["stoff"] = OFF_STATS + ring_base + 16 * lh
    return {k: v.astype(np.uint32) for k, v in out.items()}


def model(qt, dot, stats, kt, vt):
    """qt, dot: [n][64][96] float (bf16-exact), stats [n][128] (-lse2 | -D), kt, vt: [64][96].  Returns dK'^T, dV'^T as [kb][96][32] in
    the kernel's arithmetic (fp32 accumulation, P and dS rounded to bf16 before the second products)"""
    n = qt.shape[0]
    dk = np.zeros((KB, 96, 32)), np.zeros((KB, 96, 32))
    dK, dV = np.zeros((KB, 96, 32), np.float64), np.zeros((KB, 96, 32), np.float64)
    for j in range(n):
        for kb in range(KB):
            k, v = kt[32 * kb:32 * kb + 32].astype(np.float64), vt[32 * kb:32 * kb + 32].astype(np.float64)
            s = (qt[j].astype(np.float64) @ k.T + stats[j, :64, None]).astype(np.float32)
            e = (dot[j].astype(np.float64) @ v.T + stats[j, 64:, None]).astype(np.float32)
            with np.errstate(all="ignore"):
                p = np.exp2(s.astype(np.float64)).astype(np.float32)
            ds = (p * e).astype(np.float32)
            pb = bf16_to_f32(bf16_rne(p)).astype(np.float64)
            dsb = bf16_to_f32(bf16_rne(ds)).astype(np.float64)
            dV[kb] += dot[j].astype(np.float64).T @ pb
            dK[kb] += qt[j].astype(np.float64).T @ dsb
    return dK, dV


def run_case(prog, n, wave, seed=0):
    rng = np.random.default_rng(seed)

    def rb(shape, scale):
        return bf16_to_f32(bf16_rne((rng.standard_normal(shape) * scale).astype(np.float32)))
    qt, dot = rb((n, 64, 96), 0.6), rb((n, 64, 96), 0.6)
    kt, vt = rb((64, 96), 0.6), rb((64, 96), 0.6)
    stats = np.concatenate([-(rng.uniform(3.0, 6.0, size=(n, 64))), rng.standard_normal((n, 64)) * 0.3], axis=1).astype(np.float32)
    stats[n - 1, 60:64] = -1e30                                          # rows past Tq
    A_Q, A_ST, A_KV = 0x1000000, 0x2000000, 0x3000000
    RING_BASE = 0
    SIDE_BASE = OFF_STATS + R * 512
    qbuf = np.concatenate([np.concatenate([image_of(bf16_rne(qt[j]).astype(np.uint16)), image_of(bf16_rne(dot[j]).astype(np.uint16))]) for j in range(n)])
    kvbuf = np.concatenate([image_of(bf16_rne(kt).astype(np.uint16)), image_of(bf16_rne(vt).astype(np.uint16))])
    stbuf = stats.reshape(-1).view(np.uint8).copy()
    lo = lane_offsets(RING_BASE)
    inputs = {op(k): 0 for k in VOPS}
    inputs.update({op("q_lo"): A_Q & 0xffffffff, op("q_hi"): A_Q >> 32, op("st_lo"): A_ST & 0xffffffff, op("st_hi"): A_ST >> 32,
                   op("kv_lo"): A_KV & 0xffffffff, op("kv_hi"): A_KV >> 32, op("n"): n, op("wave"): wave, op("ring"): RING_BASE,
                   op("stats"): RING_BASE + OFF_STATS, op("side"): SIDE_BASE})
    w = Wave(inputs=inputs, seed=wave + 3)
    w.vector_inputs = {op(k): lo[k] for k in VOPS}
    for base, arr in ((A_Q, qbuf), (A_ST, stbuf), (A_KV, kvbuf)):
        w.map_buffer(base, arr)
    state = {"barriers": 0}

    def barrier_hook(wv):
        # barrier k: tile k's DMA has landed in every wave.  This wave's own pieces must already be there; the other waves' are put in place
        j = state["barriers"]
        state["barriers"] += 1
        if j >= n:
            raise CheckError(f"barrier {j} with {n} tiles")
        st = j % R
        own = slice(wave * 6144, wave * 6144 + 6144)
        src = qbuf[j * STAGE:(j + 1) * STAGE]
        dst = wv.lds[RING_BASE + st * STAGE:RING_BASE + (st + 1) * STAGE]
        if not (dst[own] == src[own]).all():
            raise CheckError(f"wave {wave}: its share of tile {j} has not landed at the tile's barrier")
        dst[:] = src
        sd = wv.lds[RING_BASE + OFF_STATS + st * 512:RING_BASE + OFF_STATS + st * 512 + 512]
        ss = stbuf[j * 512:(j + 1) * 512]
        o2 = slice(256 * (wave & 1), 256 * (wave & 1) + 256)
        if not (sd[o2] == ss[o2]).all():
            raise CheckError(f"wave {wave}: its share of tile {j}'s statistics has not landed at the tile's barrier")
        sd[:] = ss
    w.hooks = {"end": lambda wv: "__end__"}
    w.barrier_hook = barrier_hook
    w.run(prog, checker=check_wait_states)
    if w.vm or w.lgkm:
        raise CheckError("memory operations outstanding at the end of the statement")
    if state["barriers"] != n:
        raise CheckError(f"{state['barriers']} barriers for {n} tiles")
    lane = np.arange(LANES)
    for which, tile in enumerate((kt, vt)):
        for kb in range(KB):
            for ks in range(3):
                for i in range(2):
                    o = ((which * 2 + kb) * 3 + ks) * 2 + i
                    got = w.lds[SIDE_BASE + o * 256:SIDE_BASE + o * 256 + 256].view(np.uint32)
                    want = tile[32 * kb + (lane & 31), 8 * (2 * ks + (lane >> 5)) + 3 + 4 * i]
                    if not (bf16_to_f32(got >> 16) == want).all():
                        raise CheckError(f"wave {wave}: side value ({which}, {kb}, {ks}, {i}) is not element {3 + 4 * i} of chunk 2 ks + lh of the lane's key")
    dK, dV = model(qt, dot, stats.astype(np.float64), kt, vt)
    for name, accs, want in (("dK'", DK, dK), ("dV'", DV, dV)):
        for kb in range(KB):
            for d in range(DB):
                got = np.stack([u2f(w.v[w.ridx(r)]) for r in accs[kb][d]])            # [16][64]: register r of lane (j, h): row (r&3)+8(r>>2)+4h of the 32-channel block, key j
                for lane in range(LANES):
                    jn, h = lane & 31, lane >> 5
                    for r in range(16):
                        ch = 32 * d + (r & 3) + 8 * (r >> 2) + 4 * h
                        ref = want[kb][ch, jn]
                        if not abs(float(got[r, lane]) - ref) <= 2e-3 * max(1.0, abs(ref)):
                            raise CheckError(f"wave {wave} n {n}: {name}[kb {kb}][channel {ch}][key {jn}] = {got[r, lane]}, expected {ref}")
    return w


def check(verbose=False, cases=((1, 0), (2, 1), (5, 3), (6, 2))):
    gen = Gen()
    prog = gen.program()
    stats = {"instructions": sum(1 for x in prog if x.kind not in ("label", "pseudo")), "mfma": sum(1 for x in prog if x.kind == "mfma")}
    for n, wave in cases:
        run_case(prog, n, wave, seed=n)
        if verbose:
            print(f"  ok: {n} query tiles, wave {wave}")
    return stats


# ------------------------------------------------------------------------------------------------------------------
# emission
# ------------------------------------------------------------------------------------------------------------------
# what a stream leaves in the accumulator file, as (C lvalue of the kernel, first register) per 16-register tile: named as OUTPUTS of the
# statement ("={a[0:15]}"(dk[0][0]) ...), so that hipcc knows the values live there behind it and reads them out itself (r05; before, literal
# v_accvgpr_read statements fetched them from registers the compiler believed dead)
DKV_RESULTS = [(f"dk[{kb}][{d}]", 16 * (3 * kb + d)) for kb in range(2) for d in range(3)] + \
              [(f"dv[{kb}][{d}]", 96 + 16 * (3 * kb + d)) for kb in range(2) for d in range(3)]
DQ_RESULTS = [(f"dq[{rb}][{d}]", 16 * (3 * rb + d)) for rb in range(2) for d in range(3)]


def emit(path, prog, macro="GTA_BWD64_DKV", vops=VOPS, sops=SOPS, results=DKV_RESULTS):
    with open(path, "w") as f:
        f.write("// generated by gen_bwd64.py (make regen) -- do not edit\n")
        for name, val in (("STAGES", R), ("OFF_STATS", OFF_STATS), ("HI_BASE", HI_BASE)):
            f.write(f"#ifndef GTA_BWD64_{name}\n#define GTA_BWD64_{name} {val}\n#endif\n")
        f.write(f"#define {macro} \\\n")
        for ins in prog:
            if ins.kind != "pseudo":
                f.write(f'    "{ins.text}\\n\\t" \\\n')
        f.write('    ""\n')
        out_regs = {a0 + i for _, a0 in results for i in range(16)}
        regs_ = [f"v{i}" for i in CLOBBER_V] + [f"a{i}" for i in range(256) if i not in out_regs] + [f"s{i}" for i in CLOBBER_S]
        f.write(f"#define {macro}_CLOBBERS \\\n    " + ", ".join(f'"{r}"' for r in regs_) + ', "m0", "vcc", "scc", "memory"\n')
        f.write(f"#define {macro}_RESULTS \\\n    " + ", ".join(f'"={{a[{a0}:{a0 + 15}]}}"({lv})' for lv, a0 in results) + "\n")
        f.write(f"#define {macro}_OPERANDS \\\n    " + ", ".join(f'[{n}] "v"({n})' for n in vops) + ", \\\n    " + ", ".join(f'[{n}] "s"({n})' for n in sops) + "\n")


def assemble_check(prog, vops=VOPS, sops=SOPS):
    """the text through the assembler alone (operands replaced by registers hipcc could pick): syntax, encodable operands"""
    rep = {f"%[{n}]": f"v{i}" for i, n in enumerate(vops)}
    rep.update({f"%[{n}]": f"s{i}" for i, n in enumerate(sops)})
    lines = [".amdgcn_target \"amdgcn-amd-amdhsa--gfx950\"", ".text", "k:"]
    for ins in prog:
        if ins.kind == "pseudo":
            continue
        t = ins.text.replace("%=", "0")
        for k in sorted(rep, key=len, reverse=True):
            t = t.replace(k, rep[k])
        lines.append("  " + t)
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        return None
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "k.s")
        open(src, "w").write("\n".join(lines) + "\n")
        r = subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", src, "-o", os.path.join(td, "k.o")],
                           capture_output=True, text=True)
        if r.returncode:
            raise CheckError("the assembler rejects the stream:\n" + r.stderr[:3000])
    return True


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--out-dq")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    st = check(args.verbose)
    print(f"gen_bwd64: dK/dV {st['instructions']} instructions, {st['mfma']} MFMAs; simulation ok", file=sys.stderr)
    st = check_dq(args.verbose)
    print(f"gen_bwd64: dQ {st['instructions']} instructions, {st['mfma']} MFMAs; simulation ok", file=sys.stderr)
    if args.out:
        emit(args.out, Gen().program())
    if args.out_dq:
        emit(args.out_dq, GenDQ().program(), "GTA_BWD64_DQ", Q_VOPS, Q_SOPS, DQ_RESULTS)
