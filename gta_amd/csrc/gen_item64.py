#!/usr/bin/env python3
"""gen_item64.py -- the WHOLE work item of gta_attn64_kernel as one generated gfx950 instruction stream (dh = 96, MSN gta_so3
layout, bf16, whole 256-row items inside one view): a workgroup's persistent item loop = [item prologue | tile loop | item
epilogue], emitted as one asm statement for gta_fwd64.hip (macro GTA_ATTN64_ITEMS).

What the stream computes per item is what the C++ prologue / epilogue of gta_attn64_kernel compute (source/utils/gta.py:165,193,216
for rho_q, :246-276 for rho_q^-1, source/layers.py:202-211 between them) -- same arithmetic in the same order, so the two forms of the
kernel agree bit for bit (tests/test_gpu_attn64.py) -- but what r03 measured as 29 % of an item (15k of 52k cycles: ~2 500
hipcc-scheduled instructions of one wave per SIMD with nothing to switch to) is ~1 100 hand-placed ones here:

  * a per-item DESCRIPTOR TABLE in LDS (base addresses of Q, O, (cos, sin) rows, LSE, K'/V' images, key norms, q-side tiles), built by
    the C++ side in parallel lanes: the request for the next item's inputs is ~50 instructions off scalar bases (it was ~450 of item
    decode and 64-bit address arithmetic);
  * everything lane-dependent (gather / scatter offsets of the coalesced Q / O accesses, LDS scratch positions, (cos, sin) row
    addresses, the tile loop's LDS offsets) is computed ONCE per statement and kept in registers the tile loop does not touch;
  * the next item's Q rows, Aq tiles and key norms are requested at the START of the epilogue into registers nothing else uses (the
    accumulator-file areas the loop has just left), its (cos, sin) rows by LDS-DMA straight to where rho reads them;
  * rho_q / rho_q^-1: the 12 matrix instructions of a row block run with the OTHER row block's VALU work (so2 rotations, bf16 packs,
    accumulator-file moves, |q'|^2) between them; the XDL -> VALU wait states are filled with work instead of s_nop 15.

Checked on the CPU on every build (tests/test_host_logic.py -> check()): the stream is EXECUTED by the functional wave simulator of
isa_model.py -- 64 lanes, LDS, global memory, in-order memory counters that poison a load's destination until a wait covers it --
on random and on exactly representable inputs, several items deep, for all four waves; its Q' fragments, |q'| bounds, stream
pointers, O rows and LSE are compared with a numpy model of the C++ code it replaces; the ISA's manual wait states are checked on the
dynamic instruction order.  The tile loop inside is gen_attn64.py's own (its typed-dataflow simulation is unchanged).
"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

import gen_attn64 as G
from isa_model import (LANES, Asm, CheckError, Wave, XI, as_u32, bf16_rne, bf16_to_f32, check_wait_states, fma32, regs, rtext, u2f)

# ------------------------------------------------------------------------------------------------------------------
# shapes and the LDS map (A64<96, GTA_LAYOUT_MS> of gta_fwd64.hip with coalesced item I/O and no view records: the kernel
# static_asserts the GTA_ITEM64_* constants emitted below)
# ------------------------------------------------------------------------------------------------------------------
KS, DB, RB, R = 6, 3, 2, 4
IMG = 64 * 12 * 16
TILE = 2 * IMG
OFF_K, OFF_V, OFF_TAB = 0, R * IMG, 2 * R * IMG
OFF_CS = OFF_TAB + (KS + 2 * DB) * 256
CS_ROW, CS_WAVE = 96, 64 * 96
OFF_X = OFF_CS + 256 * CS_ROW
XROW, XB = 208, 32 * 208
OFF_ITEMS = OFF_X + 4 * XB
ITEM_BYTES = 64
MAX_ITEMS = 128                      # items per statement; the table holds two more entries (what the last item requests / reads ahead)
LDS_BYTES = OFF_ITEMS + (MAX_ITEMS + 2) * ITEM_BYTES
QT_BYTES = 1024
LN2_BITS, QN_BITS = 0x3f317218, 0x3f808659     # ln 2; 1.0041f (the slack of the |q'| bound, gta_fwd64.hip)
assert LDS_BYTES <= 160 * 1024 and OFF_ITEMS % 16 == 0

# item descriptor (dwords), written by the C++ side (gta_fwd64.hip: a64_write_item_desc)
D_Q, D_O, D_CS, D_LSE, D_IMG, D_KN, D_AQ, D_V = 0, 2, 4, 6, 8, 10, 12, 13
D_WORDS = 14


def row0(i):
    """first row of coalesced access i of a block of 192-byte rows (a64_lin_row0)"""
    return 5 * i + i // 3


# ------------------------------------------------------------------------------------------------------------------
# registers.  v0..v7 and s0..s19 are left to hipcc (the statement's operands); everything else is named here.
# ------------------------------------------------------------------------------------------------------------------
def V(n, k=1):
    return regs("v", n, k)


def A(n, k=1):
    return regs("a", n, k)


def Sr(n, k=1):
    return regs("s", n, k)


# persistent lane constants -- the tile loop names v32..v239 only
V_LANE16, V_LANE4 = "v8", "v9"
V_XFR = "v10"                           # scratch: this lane's fragment row  XW + l31 * 208 + lh * 16
V_XL = ["v11", "v12", "v13"]            # scratch: linear position of coalesced access i % 3
V_QG = ["v14", "v15", "v16"]            # global Q gather offsets of access i % 3
V_OG = ["v17", "v18", "v19"]            # global O scatter offsets
V_CSG, V_CSG2 = "v20", "v21"            # (cos, sin) rows: this lane's 16 bytes of the wave's 6 KiB (and + 4 KiB)
V_CSR, V_CSRA, V_CSRB = "v23", "v24", "v25"   # LDS (cos, sin) row of this lane: + 0, + 32 lh, + 64 lh
V_OXW = "v26"                           # scratch: where this lane's 32 output bytes of a channel block go
V_LSE = "v27"
V_QN = ["v28", "v29"]
V_KN = "v30"
V_LR, V_MR = ["v240", "v241"], ["v242", "v243"]      # the loop's outputs: row sums, running max
TMP = V(244, 12)                        # v244..v255
TMPX = TMP + ["v31", "v22"]             # (descriptor reads: 14 dwords)
# loads in flight from an item's epilogue to the next item's prologue
QL = [A(220 + 4 * i, 4) for i in range(9)] + [A(4 * i, 4) for i in range(3)]      # the wave's 64 raw Q rows as twelve coalesced quads
AT = [A(172 + 4 * i, 4) for i in range(12)]                                       # Aq tiles (hi 0..5, lo 6..11): the K' fragment area
CT = [A(124 + 4 * i, 4) for i in range(12)]                                       # Cq tiles: the Q' fragment area (free behind the loop)
# scalars
S_STAMP = Sr(20, 4)                     # s20:21 shader cycles, s22:23 the 100-MHz clock
S_B = [Sr(24 + 2 * j, 2) for j in range(4)]       # four 64-bit bases of a group of coalesced accesses
S_Q16, S_O16, S_XW = "s54", "s55", "s34"       # (s32: the ABI's stack pointer, s100 / s101: reserved -- left alone)
S_PREV, S_ROW = Sr(36, 2), Sr(38, 2)    # stamp rows of the previous / this item (profiling)
S_CUR, S_NXT = 40, 56                   # s[40:53] this item's descriptor, s[56:69] the next item's
S_KPTR, S_VPTR, S_NXK, S_NXV = Sr(72, 2), Sr(74, 2), Sr(76, 2), Sr(78, 2)
S_WOFF = "s80"
S_T = Sr(81, 3)                         # s81..s83 temporaries (s84..s96: the tile loop's)
S_K, S_TAB = "s97", "s35"               # items done, LDS address of the current item's table entry
S_PH3 = Sr(98, 2)                       # (diagnostic stream: the epilogue-done stamp, stored with the next boundary's stamps)
S_HI = Sr(70, 2)                        # exec mask of lanes 32..63
CLOBBER_S = [i for i in range(20, 100) if i != 32]

LOOP_OPERANDS = {"%[kptr]": rtext(S_KPTR), "%[vptr]": rtext(S_VPTR), "%[nxt_k]": rtext(S_NXK), "%[nxt_v]": rtext(S_NXV),
                 "%[woff]": S_WOFF, "%[lane16]": V_LANE16, "%[qn0]": V_QN[0], "%[qn1]": V_QN[1], "%[kn]": V_KN,
                 "%[lr0]": V_LR[0], "%[lr1]": V_LR[1], "%[mr0]": V_MR[0], "%[mr1]": V_MR[1]}
# the statement's operands (all "s"): LDS address of the item table, items in this chunk, wave, Q / O row bytes, q-side tile base,
# profile buffer (or 0), and the loop's n / tailj / Tk
OPERANDS = ("tabl", "nit", "wave", "qrb", "orb", "qtb_lo", "qtb_hi", "prof_lo", "prof_hi", "n", "tailj", "Tk")


def cur(d, k=2):
    return Sr(S_CUR + d, k) if k > 1 else f"s{S_CUR + d}"


def nxt(d, k=2):
    return Sr(S_NXT + d, k) if k > 1 else f"s{S_NXT + d}"


# ------------------------------------------------------------------------------------------------------------------
# s_waitcnt for the straight-line parts: LDS operations and vector memory operations complete in order
# ------------------------------------------------------------------------------------------------------------------
def auto_waits(prog):
    out = []
    lg, vm = [], []            # outstanding operations, oldest first: sets of destination registers (empty for stores / DMA)
    smem = False
    for ins in prog:
        if ins.kind in ("label", "branch"):
            if lg or smem:
                raise CheckError(f"LDS / scalar memory operations outstanding at {ins.text}: put an explicit lgkmcnt(0) in front")
            vm = []            # (what crosses the item loop's back edge is waited for explicitly; the simulation checks it)
            out.append(ins)
            continue
        if ins.kind == "wait":
            _, nv, nl = ins.fx
            if nl is not None:
                del lg[:max(0, len(lg) - nl)]
                if nl == 0:
                    smem = False
            if nv is not None:
                del vm[:max(0, len(vm) - nv)]
            out.append(ins)
            continue
        touched = set(ins.rd) | set(ins.wr)
        for q, name in ((lg, "lgkm"), (vm, "vm")):
            idx = max((i for i, d in enumerate(q) if d & touched), default=-1)
            if idx >= 0:
                left = min(len(q) - 1 - idx, 15 if name == "lgkm" else 63)      # (the counters have four / six bits)
                if name == "lgkm" and smem:
                    left = 0
                w = XI(f"s_waitcnt {'lgkmcnt' if name == 'lgkm' else 'vmcnt'}({left})", "wait",
                       fx=("waitcnt", left if name == "vm" else None, left if name == "lgkm" else None))
                out.append(w)
                del q[:len(q) - left]
                if name == "lgkm" and left == 0:
                    smem = False
        out.append(ins)
        if ins.kind == "ds":
            lg.append(set(ins.wr))
        elif ins.kind == "dsw":
            lg.append(set())
        elif ins.kind == "smem":
            smem = True
            lg.append(set(ins.wr))
        elif ins.kind == "vmem":
            vm.append(set(ins.wr))
        elif ins.kind in ("vmemst", "dma"):
            vm.append(set())
    return out


class ItemGen:
    def __init__(self, loop_kw=None, stamps=True, head_wait=False, phases=False):
        self.loop_kw = dict(G.BEST) if loop_kw is None else loop_kw
        self.stamps = stamps
        self.head_wait = head_wait     # keep the tile loop's own vmcnt(0) at its head (the item's prologue already waited for the tiles)
        self.phases = phases           # diagnostic: shader-cycle stamps [1] inputs landed, [2] prologue done, [3] loop done, [7] epilogue done

    def phase(self, a, k):
        """diagnostic stream only: shader-cycle stamp k -> slot (1, 2, 3, 7)[k] of the item's profile row, stored at once by lane 0 (the
        extra store of stamp 3 sits behind the request: the item's first wait counts it)"""
        if not self.phases:
            return
        t = TMP
        self._nph = getattr(self, "_nph", 0) + 1
        lab = f"L_noph{k}_{self._nph}_%="
        a.waitcnt(lgkm=0)
        a.s_or_b32(S_T[0], "%[prof_lo]", "%[prof_hi]")
        a.branch("s_cbranch_scc0", lab)
        if k == 3:                        # (no store here: it would sit behind the request and change what the item's first wait counts)
            a.s_memtime(S_PH3)
            a.waitcnt(lgkm=0)
            a.label(lab)
            return
        a.s_memtime(S_T[1:3])
        a.waitcnt(lgkm=0)
        a.s_exec_set("lane0")
        a.v_mov_b32(t[4], 0)
        a.v_mov_b32(t[6], S_T[1])
        a.v_mov_b32(t[7], S_T[2])
        a.global_store(2, t[4], t[6:8], S_ROW, 8 * (1, 2, 3)[k])
        a.s_exec_set("all")
        a.label(lab)

    def hi(self, a):
        a.s_exec_set("hi", S_HI)

    # ---- once per statement: lane constants, the loop's address table, the first item's inputs ------------------------
    def chunk_prologue(self, a):
        """once per statement.  First what the first item's request needs (lane ids, the gather offsets of the coalesced accesses, the
        (cos, sin) source offsets), then the request itself; every other lane constant and the tile loop's offset table are set up while
        those loads fly (r04: done in this order the launch reaches its first tile loop ~2 us sooner)."""
        t = TMP
        lane, l31, lh = t[0], t[1], t[2]
        a.v_mbcnt_lane(lane)
        a.v_and_b32(l31, 31, lane)
        a.v_lshrrev_b32(lh, 5, lane)
        a.v_lshlrev_b32(V_LANE16, 4, lane)
        a.v_lshlrev_b32(V_LANE4, 2, lane)
        a.s_mul_i32(S_XW, "%[wave]", XB)
        a.s_add_u32(S_XW, S_XW, OFF_X)
        a.s_mul_i32(S_WOFF, "%[wave]", IMG // 4)
        a.s_lshl_b32(S_Q16, "%[qrb]", 4)
        a.s_lshl_b32(S_O16, "%[orb]", 4)
        a.s_mov_b32(S_HI[0], 0)
        a.s_mov_b32(S_HI[1], -1)
        # coalesced access i of a block of 192-byte rows: unit 64 i + lane = row row0(i) + g, column Lm - 12 g, Lm = lane + 4 (i % 3),
        # g = Lm / 12 (= (43 Lm) >> 9 for Lm < 76); the wave's block starts at row 64 wave of the item.  row0(i) = row0(i % 3) +
        # 16 (i / 3): the row0(i % 3) part rides in the lane offset, the 16-row steps in four scalar bases
        a.s_lshl_b32(S_T[0], "%[wave]", 6)
        for m in range(3):
            lm, g, col = t[3], t[4], t[5]
            a.v_add_u32(lm, 4 * m, lane)
            a.v_mul_u32_u24(g, 43, lm)
            a.v_lshrrev_b32(g, 9, g)
            a.v_mul_u32_u24(col, 12, g)
            a.v_sub_u32(col, lm, col)
            a.v_lshlrev_b32(col, 4, col)                      # column * 16 bytes
            a.v_mul_u32_u24(t[6], XROW, g)
            a.v_add_u32(t[6], t[6], col)
            a.v_add_u32(V_XL[m], S_XW, t[6])
            a.v_add_u32(t[7], row0(m), g)
            a.v_add_u32(t[7], S_T[0], t[7])                   # row of the item: 64 wave + row0(m) + g
            a.v_mul_lo_u32(t[6], t[7], "%[qrb]")
            a.v_add_u32(V_QG[m], t[6], col)
            a.v_add_u32(t[7], row0(m), g)                     # O: the row blocks' and halves' bases carry 64 wave + 32 rb + 16 j
            a.v_mul_lo_u32(t[6], t[7], "%[orb]")
            a.v_add_u32(V_OG[m], t[6], col)
        a.s_mul_i32(S_T[1], "%[wave]", CS_WAVE)
        a.v_add_u32(V_CSG, S_T[1], V_LANE16)
        a.v_add_u32(V_CSG2, 4096, V_CSG)
        a.s_mov_b32(S_K, 0)
        a.s_mov_b32(S_TAB, "%[tabl]")
        if self.phases:
            a.s_mov_b64(S_PH3, 0)
        self.read_desc(a, 0)                                  # item 0's descriptor, as "next" ...
        self.request(a)                                       # ... its inputs requested ...
        self.cs_dma(a)
        self.advance_desc(a)                                  # ... and it becomes the current item
        # ---- under the loads' flight: the rest of the lane constants, the tile loop's LDS offsets, the second descriptor ----
        a.v_mbcnt_lane(lane)                                  # (the descriptor read used the temporaries)
        a.v_and_b32(l31, 31, lane)
        a.v_lshrrev_b32(lh, 5, lane)
        # scratch rows: XW + l31 * 208 (+ 16 lh: the lane's fragment column; + 32 lh: its 16 output channels of a block)
        a.v_mul_u32_u24(t[3], XROW, l31)
        a.v_add_u32(t[3], S_XW, t[3])
        a.v_lshlrev_b32(t[4], 4, lh)
        a.v_add_u32(V_XFR, t[3], t[4])
        a.v_lshlrev_b32(t[4], 5, lh)
        a.v_add_u32(V_OXW, t[3], t[4])
        a.s_mul_i32(S_T[1], "%[wave]", CS_WAVE)
        a.s_add_u32(S_T[1], S_T[1], OFF_CS)
        a.v_mul_u32_u24(t[3], CS_ROW, l31)
        a.v_add_u32(V_CSR, S_T[1], t[3])
        a.v_lshlrev_b32(t[4], 5, lh)
        a.v_add_u32(V_CSRA, V_CSR, t[4])
        a.v_lshlrev_b32(t[4], 6, lh)
        a.v_add_u32(V_CSRB, V_CSR, t[4])
        a.s_lshl_b32(S_T[0], "%[wave]", 6)
        a.v_add_u32(t[3], S_T[0], l31)
        a.v_lshlrev_b32(V_LSE, 2, t[3])
        # the tile loop's per-lane LDS offsets (written by the C++ side once per kernel): read back ONCE per statement
        a.v_add_u32(t[3], OFF_TAB, V_LANE4)
        for i, r in enumerate(G.KOFF + [G.VOFF[d][h] for d in range(DB) for h in range(2)]):
            a.ds_read(32, [r], t[3], 256 * i)
        a.waitcnt(lgkm=0)
        self.read_desc(a, 1)
        self.stamp_row(a, S_PREV)                             # (the first boundary's "previous item" is the item itself)
        a.waitcnt(vm=0)

    def read_desc(self, a, entry):
        """table entry S_TAB + 64 entry -> s[S_NXT ..] (broadcast LDS reads, v_readfirstlane)"""
        t = TMPX
        a.v_mov_b32(V_QN[0], S_TAB)                           # (a |q'| register: free outside prologue .. loop)
        for i in range(3):
            a.ds_read(128, t[4 * i:4 * i + 4], V_QN[0], ITEM_BYTES * entry + 16 * i)
        a.ds_read(32, [t[12]], V_QN[0], ITEM_BYTES * entry + 48)
        a.ds_read(32, [t[13]], V_QN[0], ITEM_BYTES * entry + 52)
        a.waitcnt(lgkm=0)
        for d in range(D_WORDS):
            a.v_readfirstlane_b32(f"s{S_NXT + d}", t[d])

    def advance_desc(self, a):
        for d in range(0, D_WORDS, 2):
            a.s_mov_b64(Sr(S_CUR + d, 2), Sr(S_NXT + d, 2))

    def bases(self, a, first, step, n=4):
        """S_B[j] = first + j * step (64-bit), j < n"""
        a.s_mov_b64(S_B[0], first)
        for j in range(1, n):
            a.s_add_u32(S_B[j][0], S_B[j - 1][0], step)
            a.s_addc_u32(S_B[j][1], S_B[j - 1][1], 0)

    def tile_bases(self, a, desc, typ):
        """S_B[0..2] = qtiles + the item's view offset + 12 KiB * type (0: Aq, 1: Cq) + 4 KiB * j"""
        a.s_add_u32(S_B[0][0], "%[qtb_lo]", desc(D_AQ, 1))
        a.s_addc_u32(S_B[0][1], "%[qtb_hi]", 0)
        if typ:
            a.s_add_u32(S_B[0][0], S_B[0][0], 12 * QT_BYTES)
            a.s_addc_u32(S_B[0][1], S_B[0][1], 0)
        for j in range(1, 3):
            a.s_add_u32(S_B[j][0], S_B[j - 1][0], 4096)
            a.s_addc_u32(S_B[j][1], S_B[j - 1][1], 0)

    def request(self, a):
        """the NEXT item's inputs (descriptor in S_NXT): Q rows (twelve coalesced 1-KiB accesses), Aq tiles, key norms -> registers
        nothing touches until that item's prologue"""
        self.bases(a, nxt(D_Q), S_Q16)
        for i in range(12):
            a.global_load(4, QL[i], V_QG[i % 3], S_B[i // 3])
        self.tile_bases(a, nxt, 0)
        for i in range(12):
            a.global_load(4, AT[i], V_LANE16, S_B[i // 4], 1024 * (i % 4))
        a.global_load(1, [V_KN], V_LANE4, nxt(D_KN))

    def cs_dma(self, a):
        """the next item's (cos, sin) rows of this wave (6 KiB, contiguous) by LDS-DMA to where rho_q / rho_q^-1 read them"""
        a.s_mul_i32(S_T[0], "%[wave]", CS_WAVE)
        for part, (voff, cnt) in enumerate(((V_CSG, 4), (V_CSG2, 2))):
            a.add(f"s_add_u32 m0, {S_T[0]}, {OFF_CS + 4096 * part}", "salu", [S_T[0]], ["m0", "scc"], ("s_add_m0", S_T[0], OFF_CS + 4096 * part))
            a.nop(1)
            for i in range(cnt):
                a.add(f"global_load_lds_dwordx4 {voff}, {rtext(nxt(D_CS))}" + (f" offset:{1024 * i}" if i else ""), "dma",
                      ["m0", voff] + nxt(D_CS), [], ("global_load_lds", voff, nxt(D_CS), 1024 * i))

    # ---- profiling: ONE (shader cycles, 100-MHz clock) pair per item boundary = end of the item before, start of this one --------
    def stamp_row(self, a, dst):
        """dst = prof + 64 * (virtual item id of the current descriptor)"""
        a.s_lshl_b32(S_T[0], cur(D_V, 1), 6)
        a.s_add_u32(dst[0], "%[prof_lo]", S_T[0])
        a.s_addc_u32(dst[1], "%[prof_hi]", 0)

    def stamp(self, a, tag, last=False):
        if not self.stamps:
            return
        lab = f"L_nostamp_{tag}_%="
        t = TMP
        a.s_or_b32(S_T[0], "%[prof_lo]", "%[prof_hi]")
        a.branch("s_cbranch_scc0", lab)
        a.s_memtime(S_STAMP[0:2])
        a.s_memrealtime(S_STAMP[2:4])
        a.waitcnt(lgkm=0)
        if not last:
            self.stamp_row(a, S_ROW)
        a.s_exec_set("lane0")
        a.v_mov_b32(t[4], 0)
        for i in range(4):
            a.v_mov_b32(t[i], S_STAMP[i])
        a.global_store(2, t[4], t[0:2], S_PREV, 8 * 4)                # [4] end of the previous item (cycles)
        a.global_store(2, t[4], t[2:4], S_PREV, 8 * 6)                # [6] ... by the 100-MHz clock
        if self.phases:
            a.v_mov_b32(t[6], S_PH3[0])
            a.v_mov_b32(t[7], S_PH3[1])
            a.global_store(2, t[4], t[6:8], S_PREV, 8 * 7)            # [7] the previous item's epilogue done
        if not last:
            a.global_store(2, t[4], t[0:2], S_ROW, 8 * 0)             # [0] start of this item
            a.global_store(2, t[4], t[2:4], S_ROW, 8 * 5)             # [5]
            a.s_mov_b64(S_PREV, S_ROW)
        a.s_exec_set("all")
        a.label(lab)

    # ---- item prologue ---------------------------------------------------------------------------------------------
    def prologue(self, a):
        t = TMP
        QR = [[V(44 + 4 * (rb * KS + ks), 4) for ks in range(KS)] for rb in range(RB)]          # raw-Q B fragments  v44..v91
        ACC = [[V(92 + 48 * rb + 16 * d, 16) for d in range(DB)] for rb in range(RB)]           # v92..v187
        CSP = [[V(188 + 16 * rb + 8 * sl, 8) for sl in range(2)] for rb in range(RB)]           # (cos, sin) pairs: slot 0 = k-step 4, slot 1 = k-step 5
        QSQ = ["v220", "v221"]
        PK = V(222, 8)                                                                          # bf16 packs on their way to the accumulator file
        # everything requested one item ago has landed -- Q rows, Aq tiles, key norms, (cos, sin) rows (LDS-DMA) -- when only the
        # previous epilogue's twelve O stores and two LSE stores are left in flight
        a.waitcnt(vm=14)
        self.phase(a, 0)
        # Q rows -> scratch (row-major, 208-byte rows) -> this lane's fragments (row l31, units 2 ks + lh), one row block at a time
        for rb in range(RB):
            for i in range(KS):
                a.ds_write(128, V_XL[i % 3], QL[rb * KS + i], row0(i) * XROW)
            for ks in range(KS):
                a.ds_read(128, QR[rb][ks], V_XFR, 32 * ks)
        # (cos, sin) pairs of the so2 chunks this lane transforms: k-step 4 = chunk 9 (lanes 32..63 only), k-step 5 = chunk 10 + lh
        for rb in range(RB):
            for h in range(2):
                a.ds_read(128, CSP[rb][0][4 * h:4 * h + 4], V_CSR, rb * 32 * CS_ROW + 16 * h)
            for h in range(2):
                a.ds_read(128, CSP[rb][1][4 * h:4 * h + 4], V_CSRA, rb * 32 * CS_ROW + 32 + 16 * h)

        def mfmas(rb):
            out = []
            for i in range(12):           # (the three accumulators take turns; per accumulator: hi kk 0, hi kk 1, lo kk 0, lo kk 1)
                d, kk, lo = i % 3, (i // 3) & 1, i // 6
                tl = 2 * d + kk
                out.append((ACC[rb][d], AT[6 * lo + tl], QR[rb][tl], 0 if i < 3 else ACC[rb][d]))
            return out

        def post(rb):
            """k-step by k-step: per-token so2 rotations, bf16, accumulator file, |q'|^2 -- closures, one instruction each"""
            ops = []
            for ks in range(KS):
                x = ACC[rb][ks >> 1][8 * (ks & 1):8 * (ks & 1) + 8]
                if ks >= 4:
                    cs = CSP[rb][ks - 4]
                    if ks == 4:
                        ops.append(lambda: self.hi(a))
                    for p in range(4):            # x0' = fma(x0, c, -(x1 s)),  x1' = fma(x0, s, x1 c)   (hipcc's contraction of rot2_apply<false>)
                        x0, x1, c, s = x[2 * p], x[2 * p + 1], cs[2 * p], cs[2 * p + 1]
                        tt = t[p]
                        ops.append(lambda x1=x1, s=s, tt=tt: a.v_mul_f32(tt, x1, s))
                        ops.append(lambda x1=x1, c=c: a.v_mul_f32(x1, x1, c))
                        ops.append(lambda x0=x0, x1=x1, s=s: a.v_fmac_f32(x1, x0, s))
                        ops.append(lambda x0=x0, c=c, tt=tt: a.v_fma_f32(x0, x0, c, tt, neg=(False, False, True)))
                    if ks == 4:
                        ops.append(lambda: a.s_exec_set("all"))
                pk = PK[4 * (ks & 1):4 * (ks & 1) + 4]
                for w in range(4):
                    ops.append(lambda w=w, pk=pk, x=x: a.v_cvt_pk_bf16_f32(pk[w], x[2 * w], x[2 * w + 1]))
                for w in range(4):
                    ops.append(lambda w=w, pk=pk, ks=ks: a.v_accvgpr_write_b32(G.Qregs(rb, ks)[w], pk[w]))
                for i in range(8):
                    if ks == 0 and i == 0:
                        ops.append(lambda x=x: a.v_mul_f32(QSQ[rb], x[0], x[0]))
                    else:
                        ops.append(lambda i=i, x=x: a.v_fmac_f32(QSQ[rb], x[i], x[i]))
            return ops

        m0, m1 = mfmas(0), mfmas(1)
        for d16, a4, b4, c in m0:
            a.mfma(d16, a4, b4, c)
        # row block 1's matrix instructions with row block 0's VALU work between them (the first accumulator of block 0 is final 12
        # states behind its last MFMA: three of block 1's MFMAs and a short pad stand there)
        p0 = post(0)
        k = 0
        per = (len(p0) + 8) // 9
        for i, (d16, a4, b4, c) in enumerate(m1):
            a.mfma(d16, a4, b4, c)
            if i == 2:
                a.nop(7)
            if i >= 3:
                for _ in range(per):
                    if k < len(p0):
                        p0[k]()
                        k += 1
        assert k == len(p0)
        for op in post(1):
            op()
        # |q'| bound per row: both halves of the row, sqrt, slack
        for rb in range(RB):
            a.v_mov_b32(t[8 + rb], QSQ[rb])
        a.nop(1)
        for rb in range(RB):
            a.v_permlane32_swap_b32(QSQ[rb], t[8 + rb])
        for rb in range(RB):
            a.v_add_f32(QSQ[rb], QSQ[rb], t[8 + rb])
        for rb in range(RB):
            a.v_sqrt_f32(QSQ[rb], QSQ[rb])
        a.nop(1)
        for rb in range(RB):
            a.v_mul_f32(V_QN[rb], QN_BITS, QSQ[rb])
        # the loop's stream pointers: K'(R ..), V'(R - 1 ..) of this item; the next item's first tiles
        for dst, src, off in ((S_KPTR, cur(D_IMG), R * TILE), (S_VPTR, cur(D_IMG), (R - 1) * TILE + IMG), (S_NXK, nxt(D_IMG), 0), (S_NXV, nxt(D_IMG), IMG)):
            a.s_add_u32(S_T[0], S_WOFF, off)
            a.s_add_u32(dst[0], src[0], S_T[0])
            a.s_addc_u32(dst[1], src[1], 0)

    # ---- item epilogue ---------------------------------------------------------------------------------------------------
    def epilogue(self, a, last=False):
        """last: the statement's final item -- nothing is requested for an item after it (the launch ends with its stores, not with loads
        nobody reads)"""
        t = TMP
        OF = [[[V(44 + 24 * rb + 8 * d + 4 * kk, 4) for kk in range(2)] for d in range(DB)] for rb in range(RB)]     # v44..v91: packed O~ B fragments
        ACC = [[V(92 + 48 * rb + 16 * d, 16) for d in range(DB)] for rb in range(RB)]                                 # v92..v187
        CSP = [[V(188 + 16 * rb + 8 * sl, 8) for sl in range(2)] for rb in range(RB)]     # slot 0: channels 8..15 of the lane's last block, slot 1: 0..7 (lanes 32..63)
        INV, LT = ["v220", "v221"], ["v222", "v223"]
        ET = [V(224, 8), V(232, 8)]
        # Two instruction lists, woven: (i) the memory side -- Cq tiles of this item's view, the next item's inputs (Q rows, Aq tiles, key
        # norms: request), its (cos, sin) rows by LDS-DMA -- 43 vector-memory instructions whose address processing would stall one
        # another back to back; (ii) the arithmetic that needs none of it yet: l, 1 / l, LSE, and (below) row block 0's B fragments.
        main = a
        # this item's (cos, sin) pairs for rho_q^-1 first: channels 8..15 of the lane's block 2 = chunk 9 + 2 lh; 0..7 (lanes 32..63) = chunk 10
        for rb in range(RB):
            for h in range(2):
                a.ds_read(128, CSP[rb][0][4 * h:4 * h + 4], V_CSRB, rb * 32 * CS_ROW + 16 * h)
            for h in range(2):
                a.ds_read(128, CSP[rb][1][4 * h:4 * h + 4], V_CSR, rb * 32 * CS_ROW + 32 + 16 * h)
        mem = Asm()
        self.tile_bases(mem, cur, 1)
        for i in range(12):
            mem.global_load(4, CT[i], V_LANE16, S_B[i // 4], 1024 * (i % 4))
        if not last:
            self.request(mem)
            mem.waitcnt(lgkm=0)            # the pairs are in registers: the (cos, sin) region may take the next item's rows
            self.cs_dma(mem)
        a = Asm()
        # l = l(lane) + l(lane ^ 32); 1 / l by hipcc's IEEE division sequence (bit-compatible with the C++ epilogue);
        # LSE = (m + log2 l) ln 2
        for rb in range(RB):
            a.v_mov_b32(t[rb], V_LR[rb])
        a.nop(1)
        for rb in range(RB):
            a.v_permlane32_swap_b32(V_LR[rb], t[rb])
        for rb in range(RB):
            a.v_add_f32(LT[rb], V_LR[rb], t[rb])
        for rb in range(RB):
            den, rc, num, e, q, r2 = t[2], t[3], t[4], t[5], t[6], t[7]
            a.v_div_scale_f32(den, S_T[1:3], LT[rb], LT[rb], 1.0)
            a.v_rcp_f32(rc, den)
            a.v_div_scale_f32(num, "vcc", 1.0, LT[rb], 1.0)
            a.v_fma_f32(e, den, rc, 1.0, neg=(True, False, False))
            a.v_fmac_f32(rc, e, rc)
            a.v_mul_f32(q, num, rc)
            a.v_fma_f32(r2, den, q, num, neg=(True, False, False))
            a.v_fmac_f32(q, r2, rc)
            a.v_fma_f32(den, den, q, num, neg=(True, False, False))
            a.v_div_fmas_f32(den, den, rc, q)
            a.v_div_fixup_f32(INV[rb], den, LT[rb], 1.0)
            a.v_log_f32(t[8 + rb], LT[rb])
        a.nop(1)
        for rb in range(RB):
            a.v_add_f32(t[8 + rb], V_MR[rb], t[8 + rb])
            a.v_mul_f32(t[8 + rb], LN2_BITS, t[8 + rb])

        def build_of(rb):
            """the normalised accumulators of a lane, packed pairwise, ARE the B fragments of rho_q^-1"""
            ops = []
            for d in range(DB):
                for kk in range(2):
                    src = G.Oregs(rb, d)[8 * kk:8 * kk + 8]
                    tt = ET[(2 * d + kk) & 1]
                    for i in range(8):
                        ops.append(lambda s=src[i], r=tt[i]: a.v_accvgpr_read_b32(r, s))
                    for i in range(8):
                        ops.append(lambda r=tt[i], rb=rb: a.v_mul_f32(r, r, INV[rb]))
                    for w in range(4):
                        ops.append(lambda w=w, tt=tt, rb=rb, d=d, kk=kk: a.v_cvt_pk_bf16_f32(OF[rb][d][kk][w], tt[2 * w], tt[2 * w + 1]))
            return ops

        def mfmas(rb):
            out = []
            for i in range(12):
                d, kk, lo = i % 3, (i // 3) & 1, i // 6
                out.append((ACC[rb][d], CT[6 * lo + 2 * d + kk], OF[rb][d][kk], 0 if i < 3 else ACC[rb][d]))
            return out

        def post(rb):
            """bf16 of the lane's 48 output channels -> scratch (row-major); the last block's so2 channels rotated back first"""
            ops = []
            x = ACC[rb][2]

            def inv_rot(base, cs, tmp):
                # x0' = fma(x0, c, x1 s),  x1' = fma(x1, c, -(x0 s))   (the values of hipcc's contraction of rot2_apply<true>), in place
                for p in range(4):
                    x0, x1, c, s = x[base + 2 * p], x[base + 2 * p + 1], cs[2 * p], cs[2 * p + 1]
                    tt, uu = tmp[2 * p], tmp[2 * p + 1]
                    ops.append(lambda x1=x1, s=s, tt=tt: a.v_mul_f32(tt, x1, s))
                    ops.append(lambda x0=x0, s=s, uu=uu: a.v_mul_f32(uu, x0, s))
                    ops.append(lambda x0=x0, c=c, tt=tt: a.v_fma_f32(x0, x0, c, tt))
                    ops.append(lambda x1=x1, c=c, uu=uu: a.v_fma_f32(x1, x1, c, uu, neg=(False, False, True)))

            def packs(d):
                for half in range(2):
                    pk = OF[rb][d][half]                      # (the B fragments are consumed: their registers take the packed output)
                    xs = ACC[rb][d][8 * half:8 * half + 8]
                    for w in range(4):
                        ops.append(lambda w=w, pk=pk, xs=xs: a.v_cvt_pk_bf16_f32(pk[w], xs[2 * w], xs[2 * w + 1]))
                    ops.append(lambda pk=pk, d=d, half=half: a.ds_write(128, V_OXW, pk, 64 * d + 16 * half))
            packs(0)
            packs(1)
            inv_rot(8, CSP[rb][0], ET[0])          # channels 8..15 of block 2: every lane (chunk 9 + 2 lh)
            ops.append(lambda: self.hi(a))
            inv_rot(0, CSP[rb][1], ET[1])          # channels 0..7: lanes 32..63 (chunk 10)
            ops.append(lambda: a.s_exec_set("all"))
            packs(2)
            return ops

        def stores(rb):
            ops = []
            xs = [ACC[rb][i // 4][4 * (i % 4):4 * (i % 4) + 4] for i in range(KS)]     # (the accumulators are consumed: 24 of their registers stage the rows)
            for i in range(KS):
                ops.append(lambda i=i: a.ds_read(128, xs[i], V_XL[i % 3], row0(i) * XROW))
            for i in range(KS):
                ops.append(lambda i=i, rb=rb: a.global_store(4, V_OG[i % 3], xs[i], S_B[2 * rb + i // 3]))
            return ops

        def weave(mf, fill, first=0, pad_at=None, pad=0):
            n = max(len(mf) - first, 1)
            per = (len(fill) + n - 1) // n
            k = 0
            for i, (d16, a4, b4, c) in enumerate(mf):
                a.mfma(d16, a4, b4, c)
                if pad_at is not None and i == pad_at:
                    a.nop(pad)
                if i >= first:
                    for _ in range(per):
                        if k < len(fill):
                            fill[k]()
                            k += 1
            while k < len(fill):
                fill[k]()
                k += 1

        for op in build_of(0):
            op()
        # weave: one memory-side instruction per three of the arithmetic
        arith, a = a.out, main
        mi, per = 0, max(1, len(arith) // max(len(mem.out), 1))
        for i, ins in enumerate(arith):
            a.raw(ins)
            if (i + 1) % per == 0 and mi < len(mem.out):
                a.raw(mem.out[mi])
                mi += 1
        for ins in mem.out[mi:]:
            a.raw(ins)
        # O stores: four bases (row block, half): this item's O rows + (64 wave + 32 rb + 16 j) rows
        a.s_lshl_b32(S_T[0], "%[wave]", 2)
        a.s_mul_i32(S_T[0], S_T[0], S_O16)                    # 64 wave rows
        a.s_add_u32(S_B[0][0], cur(D_O)[0], S_T[0])
        a.s_addc_u32(S_B[0][1], cur(D_O)[1], 0)
        for j in range(1, 4):
            a.s_add_u32(S_B[j][0], S_B[j - 1][0], S_O16)
            a.s_addc_u32(S_B[j][1], S_B[j - 1][1], 0)
        a.waitcnt(vm=0 if last else 31)    # the Cq tiles (behind them: 12 Q quads, 12 Aq tiles, the key norms, six DMA pieces)
        weave(mfmas(0), build_of(1))
        weave(mfmas(1), post(0) + stores(0), first=3, pad_at=2, pad=8)
        a.nop(8)
        for op in post(1) + stores(1):
            op()
        # LSE of the wave's 64 rows (lanes 0..31 hold a row each)
        a.s_exec_set("lo")
        for rb in range(RB):
            a.global_store(1, V_LSE, [t[8 + rb]], cur(D_LSE), 128 * rb)
        a.s_exec_set("all")

    # ---- the statement ---------------------------------------------------------------------------------------------------------
    def loop_program(self):
        G.configure(96)
        gen = G.Gen(R=R, **self.loop_kw)
        out = []
        head_vm = not self.head_wait
        for ins in gen.program():
            if ins.sem and ins.sem[0] == "ds_tab":
                continue
            if head_vm and ins.kind == "wait" and ins.sem == ("vm", 0):
                # the loop's own wait for its first tiles: in the item stream the prologue's vmcnt(14) has already seen them land
                # (they were requested before the item's inputs), and what is still in flight here -- the previous item's O / LSE
                # stores -- must not hold the loop's first matrix instructions
                head_vm = False
                continue
            text = ins.text
            for k, v in LOOP_OPERANDS.items():
                text = text.replace(k, v)
            out.append(XI(text, ins.kind, ins.rd, ins.wr, None, ins.sem, ins.label, ins.target))
        return gen, out

    def program(self, stub_loop=False, pad4=False):
        a = Asm()
        self.chunk_prologue(a)
        a.label("L_item_%=")
        self.stamp(a, "item")
        self.prologue(a)
        a.waitcnt(lgkm=0)
        self.phase(a, 1)
        if pad4:                           # (diagnostic twin: the tile loop four bytes further on -- MI355X_MICROARCH.md, code placement)
            a.nop(1)
        head = auto_waits(a.out)
        a = Asm()
        self.phase(a, 2)
        a.s_add_u32(S_T[0], S_K, 1)
        a.s_cmp("ge", "u32", S_T[0], "%[nit]")
        a.branch("s_cbranch_scc1", "L_last_%=")
        self.epilogue(a)
        # next item: its descriptor becomes the current one, the one after it is read from the table
        a.waitcnt(lgkm=0)
        self.phase(a, 3)
        self.advance_desc(a)
        a.s_add_u32(S_K, S_K, 1)
        a.s_add_u32(S_TAB, S_TAB, ITEM_BYTES)
        self.read_desc(a, 1)
        a.branch("s_branch", "L_item_%=")
        a.label("L_last_%=")
        self.epilogue(a, last=True)
        a.waitcnt(lgkm=0)
        self.phase(a, 3)
        self.stamp(a, "end", last=True)
        a.waitcnt(vm=0)
        a.pseudo("end")
        tail = auto_waits(a.out)
        if stub_loop:
            mid = [XI("", "pseudo", fx=("pseudo", "loop"))]
        else:
            mid = self.loop_program()[1]
        return head + mid + tail


# ------------------------------------------------------------------------------------------------------------------
# numpy model of what the stream replaces (the C++ prologue / epilogue of gta_attn64_kernel, gta_flash_common.h's tile layout)
# ------------------------------------------------------------------------------------------------------------------
def qt_chan_q(d, rho):
    lh, j, i = (rho >> 2) & 1, rho >> 3, rho & 3
    return 32 * d + 8 * lh + 4 * j + i if j < 2 else 32 * d + 16 + 8 * lh + 4 * (j - 2) + i


def qt_chan_o_in(d, kk, k):
    lh, i = k >> 3, k & 7
    return 32 * d + 16 * kk + (i & 3) + 8 * (i >> 2) + 4 * lh


def qt_chan_o_out(d, rho):
    lh, r = (rho >> 2) & 1, (rho & 3) + 4 * (rho >> 3)
    return 32 * d + 16 * lh + r


def build_view_tiles(Mf, Mi):
    """gta_qt_build_view: the 24 operand tiles (1 KiB each) of one view from its 96 x 96 forward / inverse matrices"""
    out = np.zeros(24 * 1024, np.uint8)
    o16 = out.view(np.uint16)
    for typ, M in ((0, Mf), (1, Mi)):
        for d in range(3):
            for kk in range(2):
                for rho in range(32):
                    r = qt_chan_o_out(d, rho) if typ else qt_chan_q(d, rho)
                    for k in range(16):
                        c = qt_chan_o_in(d, kk, k) if typ else 32 * d + 16 * kk + k
                        x = np.float32(M[r, c])
                        hi = bf16_rne(np.array([x], np.float32))[0]
                        lo = bf16_rne(np.array([x - bf16_to_f32(np.uint32(hi))], np.float32))[0]
                        for part, val in ((0, hi), (1, lo)):
                            tile = 12 * typ + 6 * part + 2 * d + kk
                            o16[(tile * 1024 + (((k >> 3) * 32 + rho) * 16 + (k & 7) * 2)) // 2] = val
    return out


def decode_matrix(tiles, typ):
    """the matrix (hi + lo) the tiles of one view stand for"""
    M = np.zeros((96, 96), np.float64)
    t16 = tiles.view(np.uint16)
    for d in range(3):
        for kk in range(2):
            for rho in range(32):
                r = qt_chan_o_out(d, rho) if typ else qt_chan_q(d, rho)
                for k in range(16):
                    c = qt_chan_o_in(d, kk, k) if typ else 32 * d + 16 * kk + k
                    for part in range(2):
                        tile = 12 * typ + 6 * part + 2 * d + kk
                        M[r, c] += float(bf16_to_f32(np.uint32(t16[(tile * 1024 + (((k >> 3) * 32 + rho) * 16 + (k & 7) * 2)) // 2])))
    return M


class Case:
    """synthetic launch: buffers, descriptors, and the expected results of the prologue / epilogue for every item and wave"""

    def __init__(self, seed, n_items, exact, n_tiles=8):
        rng = np.random.default_rng(seed)
        self.rng, self.exact, self.n_items, self.n_tiles = rng, exact, n_items, n_tiles
        B, H, Nq, Tq = 2, 3, 2, 512                                  # two 256-row items per (b, h), one view each
        self.B, self.H, self.Nq, self.Tq = B, H, Nq, Tq
        self.q_st, self.o_st = 3 * H * 96, H * 96                    # the packed projection's row; the output's [B, Tq, H, dh] rows
        if exact:
            q = rng.integers(-3, 4, size=(B, Tq, 3 * H * 96)).astype(np.float32)
        else:
            q = rng.standard_normal((B, Tq, 3 * H * 96)).astype(np.float32)
        self.qbits = bf16_rne(q).astype(np.uint16)                   # [B, Tq, q_st] bf16
        self.obits = np.zeros((B, Tq, H * 96), np.uint16)
        if exact:
            ang = rng.integers(0, 4, size=(B, Tq, 12))
            cs = np.stack([np.round(np.cos(ang * np.pi / 2)), np.round(np.sin(ang * np.pi / 2))], -1).astype(np.float32)
        else:
            ang = rng.uniform(0, 2 * np.pi, size=(B, Tq, 12))
            cs = np.stack([np.cos(ang), np.sin(ang)], -1).astype(np.float32)
        self.cs = cs                                                  # [B, Tq, 12, 2]
        self.lse = np.zeros((B, H, Tq), np.float32)
        self.kn = rng.uniform(0.5, 2.0, size=(B, H, 64)).astype(np.float32)      # (pitch 64 here; the kernel's is n_tiles)
        self.Mf, self.Mi, tiles = [], [], []
        for v in range(B * Nq):
            Mf, Mi = np.zeros((96, 96)), np.zeros((96, 96))
            scale = 2.0 if exact else 0.1472
            for M, inv in ((Mf, 0), (Mi, 1)):
                blocks = [(4 * i, 4) for i in range(12)] + [x for g in range(3) for x in ((48 + 8 * g, 3), (51 + 8 * g, 5))]
                for o, n in blocks:
                    if exact:
                        blk = rng.integers(-2, 3, size=(n, n)).astype(np.float64)
                    else:
                        blk = rng.standard_normal((n, n)) * (1.0 if inv else scale)
                    M[o:o + n, o:o + n] = blk
                for c in range(72, 96):
                    M[c, c] = 1.0 if inv else scale
            self.Mf.append(Mf); self.Mi.append(Mi)
            tiles.append(build_view_tiles(Mf, Mi))
        self.tiles = np.concatenate(tiles)
        self.MfD = [decode_matrix(self.tiles[24 * 1024 * v:24 * 1024 * (v + 1)], 0) for v in range(B * Nq)]
        self.MiD = [decode_matrix(self.tiles[24 * 1024 * v:24 * 1024 * (v + 1)], 1) for v in range(B * Nq)]
        # global address space
        self.A_Q, self.A_O, self.A_CS, self.A_LSE, self.A_KN, self.A_QT, self.A_IMG, self.A_PROF = (0x1000000 * (i + 1) for i in range(8))
        self.items = []
        order = rng.permutation(B * H * 2)[:n_items]
        for w in order:
            bh, qt = divmod(int(w), 2)
            b, h = divmod(bh, H)
            self.items.append(dict(b=b, h=h, q0=256 * qt, view=(256 * qt) // (Tq // Nq), V=int(w)))
        # the loop's results per item and wave: O accumulators (fp32), row sums per lane (two halves), running max
        self.oacc, self.lrun, self.mrun = {}, {}, {}

    def desc(self, it):
        b, h, q0 = it["b"], it["h"], it["q0"]
        bh = b * self.H + h
        d = np.zeros(16, np.uint32)

        def put(i, addr):
            d[i], d[i + 1] = addr & 0xffffffff, addr >> 32
        put(D_Q, self.A_Q + ((b * self.Tq + q0) * self.q_st + h * 96) * 2)
        put(D_O, self.A_O + ((b * self.Tq + q0) * self.o_st + h * 96) * 2)
        put(D_CS, self.A_CS + ((b * self.Tq + q0) * 24) * 4)
        put(D_LSE, self.A_LSE + (bh * self.Tq + q0) * 4)
        put(D_IMG, self.A_IMG + bh * self.n_tiles * TILE)
        put(D_KN, self.A_KN + bh * 64 * 4)
        d[D_AQ] = (b * self.Nq + it["view"]) * 24 * 1024
        d[D_V] = it["V"]
        return d

    def loop_results(self, item, wave):
        key = (item, wave)
        if key not in self.oacc:
            rng = np.random.default_rng(1000 * item + wave + 7)
            if self.exact:
                l_half = np.full((RB, LANES), 2.0, np.float32)                         # l = 4 per row
                o = (rng.integers(-8, 9, size=(RB, DB, 16, LANES)) * 4).astype(np.float32)
            else:
                l_half = rng.uniform(1.0, 40.0, size=(RB, LANES)).astype(np.float32)
                o = (rng.standard_normal((RB, DB, 16, LANES)) * 8).astype(np.float32)
            self.oacc[key], self.lrun[key] = o, l_half
            self.mrun[key] = rng.uniform(-3, 9, size=(RB, 32)).astype(np.float32)      # per ROW (both lane halves hold the same running max)
        return self.oacc[key], self.lrun[key], self.mrun[key]

    # -- expected --
    def expect_q(self, item, wave):
        """per (rb, row): q' fp32 [96] before rounding (fp64 arithmetic: compared with a tolerance unless the case is exact), |q'| bound"""
        it = self.items[item]
        out = np.zeros((RB, 32, 96), np.float64)
        for rb in range(RB):
            for r in range(32):
                t = it["q0"] + wave * 64 + rb * 32 + r
                q = bf16_to_f32(self.qbits[it["b"], t, it["h"] * 96:(it["h"] + 1) * 96].astype(np.uint32)).astype(np.float64)
                x = self.MfD[it["b"] * self.Nq + it["view"]] @ q
                for p in range(12):
                    c, s = (float(v) for v in self.cs[it["b"], t, p])
                    x0, x1 = x[72 + 2 * p], x[73 + 2 * p]
                    x[72 + 2 * p], x[73 + 2 * p] = c * x0 - s * x1, s * x0 + c * x1
                out[rb, r] = x
        return out

    def expect_o(self, item, wave):
        it = self.items[item]
        o, lh_, m = self.loop_results(item, wave)
        rows = np.zeros((RB, 32, 96), np.float64)
        lse = np.zeros((RB, 32), np.float64)
        for rb in range(RB):
            for r in range(32):
                l = np.float32(lh_[rb, r] + lh_[rb, r + 32])
                inv = np.float32(1.0) / l
                ot = np.zeros(96, np.float64)
                for d in range(DB):
                    for reg in range(16):
                        for h in range(2):
                            chan = 32 * d + (reg & 3) + 8 * (reg >> 2) + 4 * h
                            ot[chan] = bf16_to_f32(bf16_rne(np.array([o[rb, d, reg, r + 32 * h] * inv], np.float32)))[0]
                t = it["q0"] + wave * 64 + rb * 32 + r
                x = self.MiD[it["b"] * self.Nq + it["view"]] @ ot
                for p in range(12):
                    c, s = (float(v) for v in self.cs[it["b"], t, p])
                    x0, x1 = x[72 + 2 * p], x[73 + 2 * p]
                    x[72 + 2 * p], x[73 + 2 * p] = c * x0 + s * x1, c * x1 - s * x0
                rows[rb, r] = x
                lse[rb, r] = (float(m[rb, r]) + np.log2(float(l))) * np.log(2.0)
        return rows, lse


def run_case(prog, case, wave, prof=False):
    """one wave through the whole statement; checks at the loop stub and at the end"""
    LDS0 = np.zeros(160 * 1024, np.uint8)
    n = case.n_items
    for k in range(n + 2):
        it = case.items[min(k, n - 1)]
        LDS0[OFF_ITEMS + 64 * k:OFF_ITEMS + 64 * k + 64] = case.desc(it).view(np.uint8)
    # the loop's per-lane offset table: any recognisable values
    tab = (np.arange(12 * 64, dtype=np.uint32) * 4 + 0x100000).view(np.uint8)
    LDS0[OFF_TAB:OFF_TAB + len(tab)] = tab
    inputs = {"%[tabl]": OFF_ITEMS, "%[nit]": n, "%[wave]": wave, "%[qrb]": case.q_st * 2, "%[orb]": case.o_st * 2,
              "%[qtb_lo]": case.A_QT & 0xffffffff, "%[qtb_hi]": case.A_QT >> 32,
              "%[prof_lo]": (case.A_PROF & 0xffffffff) if prof else 0, "%[prof_hi]": (case.A_PROF >> 32) if prof else 0,
              "%[n]": case.n_tiles, "%[tailj]": 0xffffffff, "%[Tk]": 64 * case.n_tiles}
    w = Wave(inputs=inputs, seed=wave + 11)
    w.lds[:] = LDS0
    qb = case.qbits.reshape(-1).view(np.uint8)
    ob = case.obits.reshape(-1).view(np.uint8)
    profb = np.zeros(64 * 64, np.uint8)
    for base, arr in ((case.A_Q, qb), (case.A_O, ob), (case.A_CS, case.cs.reshape(-1).view(np.uint8)), (case.A_LSE, case.lse.reshape(-1).view(np.uint8)),
                      (case.A_KN, case.kn.reshape(-1).view(np.uint8)), (case.A_QT, case.tiles), (case.A_PROF, profb)):
        w.map_buffer(base, arr)
    state = {"item": 0}
    tol = 0.0 if case.exact else 2.0 ** -7

    def chk(name, got, want, rel):
        got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
        err = np.abs(got - want)
        lim = rel * np.maximum(np.abs(want), 1e-3) + (0 if rel == 0 else 1e-6)
        if not (err <= lim).all():
            i = np.unravel_index(np.argmax(err - lim), err.shape)
            raise CheckError(f"wave {wave} item {state['item']}: {name} differs at {i}: got {got[i]!r}, expected {want[i]!r}")

    def loop_hook(wv):
        k = state["item"]
        it = case.items[k]
        wv.wait(0, None)
        # ---- what the prologue must have produced ----
        exp = case.expect_q(k, wave)
        for rb in range(RB):
            for ks in range(KS):
                rr = G.Qregs(rb, ks)
                for lane in range(LANES):
                    l31, lh = lane & 31, lane >> 5
                    chunk = 2 * ks + lh
                    got = []
                    for wd in range(4):
                        u = int(wv.v[wv.ridx(rr[wd])][lane])
                        got += [float(bf16_to_f32(np.uint32(u & 0xffff))), float(bf16_to_f32(np.uint32(u >> 16)))]
                    want = exp[rb, l31, 8 * chunk:8 * chunk + 8]
                    if case.exact:
                        chk(f"Q' rb {rb} k-step {ks} lane {lane}", got, want, 0.0)
                    else:
                        wb = bf16_to_f32(bf16_rne(want.astype(np.float32))).astype(np.float64)
                        err = np.abs(np.array(got) - wb)
                        if not (err <= 2.0 ** -7 * np.maximum(np.abs(wb), 1e-2)).all():
                            raise CheckError(f"wave {wave} item {k}: Q' rb {rb} k-step {ks} lane {lane}: {got} vs {wb}")
            qn = u2f(wv.v[wv.ridx(V_QN[rb])])
            want = np.sqrt((exp[rb] ** 2).sum(-1)) * 1.0041
            chk(f"|q'| rb {rb}", qn[:32], want, 1e-5)
            chk(f"|q'| rb {rb} (upper lanes)", qn[32:], want, 1e-5)
        bh = it["b"] * case.H + it["h"]
        chk("key norms", u2f(wv.v[wv.ridx(V_KN)])[:case.n_tiles], case.kn[it["b"], it["h"], :case.n_tiles], 0.0)
        img = case.A_IMG + bh * case.n_tiles * TILE + wave * (IMG // 4)
        nit_ = case.items[min(k + 1, n - 1)]
        nimg = case.A_IMG + (nit_["b"] * case.H + nit_["h"]) * case.n_tiles * TILE + wave * (IMG // 4)
        for nm, pair, want in (("kptr", S_KPTR, img + R * TILE), ("vptr", S_VPTR, img + (R - 1) * TILE + IMG), ("nxt_k", S_NXK, nimg), ("nxt_v", S_NXV, nimg + IMG)):
            if wv.s64(pair) != want:
                raise CheckError(f"wave {wave} item {k}: {nm} = 0x{wv.s64(pair):x}, expected 0x{want:x}")
        for i, r in enumerate(G.KOFF + [G.VOFF[d][h] for d in range(DB) for h in range(2)]):
            if not (wv.v[wv.ridx(r)] == (np.arange(64) + 64 * i) * 4 + 0x100000).all():
                raise CheckError(f"the loop's LDS offset register {r} does not hold table row {i}")
        # ---- what the loop leaves: junk in its registers, O / l / m of this item, 21 DMA pieces of the next item's first tiles in flight ----
        junk = np.random.default_rng(5).integers(0x7f800001, 0x7fffffff, size=(208, LANES), dtype=np.uint32)
        wv.v[32 + 12:240] = junk[:196]
        wv.v[256 + 124:512] = np.random.default_rng(6).integers(0x7f800001, 0x7fffffff, size=(132, LANES), dtype=np.uint32)
        o, lh_, m = case.loop_results(k, wave)
        for rb in range(RB):
            for d in range(DB):
                for i, r in enumerate(G.Oregs(rb, d)):
                    wv.v[wv.ridx(r)] = as_u32(o[rb, d, i])
            wv.v[wv.ridx(V_LR[rb])] = as_u32(lh_[rb])
            wv.v[wv.ridx(V_MR[rb])] = as_u32(np.concatenate([m[rb], m[rb]]))
        for _ in range(21):
            wv.vm.append([])
        return None

    def end_hook(wv):
        return "__end__"

    # the O rows and LSE of an item are checked when the NEXT item reaches the loop stub (its prologue waited for them) / at the end
    def check_outputs(k):
        rows, lse = case.expect_o(k, wave)
        it = case.items[k]
        for rb in range(RB):
            for r in range(32):
                t = it["q0"] + wave * 64 + rb * 32 + r
                got = bf16_to_f32(case.obits[it["b"], t, it["h"] * 96:(it["h"] + 1) * 96].astype(np.uint32)).astype(np.float64)
                want = rows[rb, r]
                if case.exact:
                    chk(f"O row {t}", got, want, 0.0)
                else:
                    wb = bf16_to_f32(bf16_rne(want.astype(np.float32))).astype(np.float64)
                    err = np.abs(got - wb)
                    if not (err <= 2.0 ** -6 * np.maximum(np.abs(wb), 2e-2)).all():
                        i = int(np.argmax(err))
                        raise CheckError(f"wave {wave} item {k}: O row {t} channel {i}: {got[i]} vs {wb[i]}")
                chk(f"LSE row {t}", [case.lse[it["b"], it["h"], t]], [lse[rb, r]], 1e-5)

    def loop_hook2(wv):
        r = loop_hook(wv)
        state["item"] += 1
        return r
    w.hooks = {"loop": loop_hook2, "end": end_hook}
    w.run(prog, checker=check_wait_states)
    if w.vm or w.lgkm:
        raise CheckError("memory operations outstanding at the end of the statement")
    for k in range(n):
        state["item"] = k
        check_outputs(k)
    if prof:
        P = profb.view(np.uint64).reshape(-1, 8)
        for k, it in enumerate(case.items):
            row = P[it["V"]]
            if not (row[0] > 0 and row[4] > row[0] and row[5] > 0 and row[6] > row[5]):
                raise CheckError(f"profile row of item {k}: {row}")
    return w


def check(verbose=False, waves=(0, 1, 2, 3), **kw):
    gen = ItemGen(**kw)
    prog = gen.program(stub_loop=True)
    stats = {"instructions": sum(1 for x in prog if x.kind not in ("label", "pseudo")), "mfma": sum(1 for x in prog if x.kind == "mfma")}
    # distinct items of one launch write distinct rows: the waves of a case share the O / LSE buffers
    for seed, n_items, exact in ((1, 3, True), (2, 3, False), (3, 1, False)):
        case = Case(seed, n_items, exact)
        for wave in waves:
            run_case(prog, case, wave, prof=(seed == 2 and wave == 1))
        if verbose:
            print(f"  ok: case seed {seed}: {n_items} items, {'exact' if exact else 'random'} inputs, waves {list(waves)}")
    return stats


# ------------------------------------------------------------------------------------------------------------------
# emission
# ------------------------------------------------------------------------------------------------------------------
def clobbers():
    regs_ = [f"v{i}" for i in range(8, 256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in CLOBBER_S]
    return ", ".join(f'"{r}"' for r in regs_) + ', "m0", "vcc", "scc", "memory"'


def emit(path, prog, prog_p4=None, prog_ph=None):
    with open(path, "w") as f:
        f.write("// generated by gen_item64.py (make regen) -- do not edit\n")
        for name, val in (("OFF_CS", OFF_CS), ("OFF_X", OFF_X), ("OFF_ITEMS", OFF_ITEMS), ("ITEM_BYTES", ITEM_BYTES), ("MAX_ITEMS", MAX_ITEMS),
                          ("LDS_BYTES", LDS_BYTES), ("D_Q", D_Q), ("D_O", D_O), ("D_CS", D_CS), ("D_LSE", D_LSE), ("D_IMG", D_IMG), ("D_KN", D_KN),
                          ("D_AQ", D_AQ), ("D_V", D_V)):
            f.write(f"#define GTA_ITEM64_{name} {val}\n")
        f.write("#define GTA_ATTN64_ITEMS \\\n")
        for ins in prog:
            if ins.kind == "pseudo":
                continue
            f.write(f'    "{ins.text}\\n\\t" \\\n')
        f.write('    ""\n')
        if prog_p4 is not None:
            f.write("#ifdef GTA_ATTN64_DIAG\n#define GTA_ATTN64_ITEMS_P4 \\\n")
            for ins in prog_p4:
                if ins.kind != "pseudo":
                    f.write(f'    "{ins.text}\\n\\t" \\\n')
            f.write('    ""\n')
            if prog_ph is not None:
                f.write("#define GTA_ATTN64_ITEMS_PH \\\n")
                for ins in prog_ph:
                    if ins.kind != "pseudo":
                        f.write(f'    "{ins.text}\\n\\t" \\\n')
                f.write('    ""\n')
            f.write("#endif\n")
        f.write("#define GTA_ATTN64_ITEMS_CLOBBERS \\\n    " + clobbers() + "\n")


def assemble_check(prog):
    """the text through the assembler alone (operands replaced by registers hipcc could pick): syntax, encodable operands"""
    rep = {f"%[{n}]": f"s{i}" for i, n in enumerate(OPERANDS)}
    lines = [".amdgcn_target \"amdgcn-amd-amdhsa--gfx950\"", ".text", "k:"]
    for ins in prog:
        if ins.kind == "pseudo":
            continue
        t = ins.text.replace("%=", "0")
        for k, v in rep.items():
            t = t.replace(k, v)
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
    ap.add_argument("--out", default=None)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    a_ = ap.parse_args()
    if not a_.no_check:
        st = check(verbose=a_.verbose)
        if a_.verbose:
            print("item stream without the tile loop:", st)
    full = ItemGen().program()
    assemble_check(full)
    if a_.out:
        emit(a_.out, full, ItemGen().program(pad4=True), ItemGen(phases=True).program())
