// gta_fwd_cl.hip -- the attention kernel of the two-stage forward plan for bf16 inputs at dh = 64 in the CLEVR-TR layout
// (se3 32 | so2 32, runs/clevrtr/GTA/gta/config.yaml:19-52) and the pure-so2 layout (the *_no3demb configs): gta.py:160-219 on Q,
// layers.py:202-211 over the K'/V' tile images of gta_prep.hip, gta.py:246-276 on the output.
//
// Same work decomposition as gta_fwd2_kernel (one work item = 128 query rows of one (b,h), 4 waves x 32 rows, three workgroups per
// CU, 3-stage LDS-DMA ring of [K' | V'] tile images, lazy online softmax on the per-tile key-norm bound) and the same arithmetic per
// score; what differs is everything AROUND the matrix instructions (the shape's 2 032 VALU-class instructions per
// item and wave, 43 % of them per-ITEM work):
//   * the ring stage is a COMPILE-TIME constant of each tile step (the tile loop is unrolled by the three stages): every K' fragment
//     read and V' transpose-read is `lane constant + immediate`, the DMA destination is `M0 = base + immediate`, the image pointer
//     one 64-bit scalar add per tile -- no per-tile vector address arithmetic;
//   * both score accumulators take the -m splat as the C operand of their first MFMA without being copied into (the first MFMA of
//     each chain is an asm statement with an early-clobber destination: hipcc's two-address form cost 8 v_mov_b64 per tile);
//   * P = exp2(S') slab by slab with the P V MFMAs of the previous slab in between (a wave whose own stream alternates matrix and
//     vector work overlaps the pipes; phase-separated streams of co-resident waves do not: tests/probes/probe_mix64.hip), row sums
//     from the PACKED words (16 v_dot2c_f32_bf16 instead of 32 adds; through the builtin -- see dot2_bf16);
//   * rho_q / rho_q^-1 without LDS view records and without their barrier: the 4x4 of the wave's view sits in SGPRs (one
//     s_load_dwordx16 of the view record, the trans_coeff mask of gta.py:40-44 folded into the arithmetic: its zero row and its
//     unit entries cost nothing); a wave whose 32 rows straddle a view boundary walks its views under a select;
//   * the epilogue applies rho_q^-1 to the HALF-chunks the accumulators already hold (an se3 4-vector and an so2 2x2 block never
//     straddle a 4-channel half): no v_permlane32_swap, eight 8-byte stores per lane;
//   * an item's first two tiles are requested before its Q rows; |q'| from the packed bf16 words; item decode through float
//     reciprocals; nothing lane-derived lives through the tile loop (re-derived from v_mbcnt where needed): 168 VGPRs, no scratch.
// Measured (profiles/r06/README.md section 1): 1 470 instead of 2 032 VALU-class instructions per item and wave, -5 % shader cycles at
// cl-enc and cl-dec -- and the SAME microseconds (the part grants the denser kernel 1.95 instead of 2.08 GHz).  The ablations of this
// kernel say why instruction count was not the lever: without ANY matrix or softmax instruction the tile loop still takes 68 % of its
// time -- the per-tile barrier among four waves that each share a SIMD with two other workgroups (~15 %), the L2 -> LDS stream
// (16 KiB per 128 rows and tile: twice the bytes per flop of the 256-row items of gta_fwd64.hip) and the chain of their latencies.
// Key sides of ONE tile stay with gta_fwd2_kernel (there it and the single-kernel plan agree to the last bit, which the chunked
// decode's cached / uncached layers rely on), as does every other layout, dtype and head size.
#include <atomic>
#include <cstdlib>
#include <hip/hip_ext.h>
#include "gta_flash_common.h"

extern thread_local void* gta_dbg_fwd_ev_start;      // gta_fwd2.hip: events for the next attention-kernel launch of this thread
extern thread_local void* gta_dbg_fwd_ev_stop;

// timing-only ablations of the development builds (tools/_ab_fwdc.sh: -DGTA_FWDC_ABL=<bits>; wrong results): 1 no tile barrier, 2 no softmax
// vector work, 4 no matrix instructions, 8 no LDS reads, 16 no tile DMA, 32 no rho_q arithmetic, 64 no epilogue arithmetic; 512 (correct results)
// fragment-direct item I/O instead of the coalesced form
#ifndef GTA_FWDC_ABL
#define GTA_FWDC_ABL 0
#endif

namespace {

constexpr int C_BM = 128, C_DHP = 64, C_CHP = 8, C_KS = 4, C_DB = 2;
constexpr int C_IMG = BN * C_DHP * 2;          // one K' or V' tile image: 8 KiB
constexpr int C_STAGE = 2 * C_IMG;             // [K' | V']
constexpr int C_NST = 3;
constexpr int C_LDS = C_NST * C_STAGE;         // 48 KiB: three workgroups per CU
constexpr int C_MAX_TILES = 4096;

GTA_DEV int c_item_of(int V, int n_items) {    // all query tiles of one (b,h) on one XCD (gta_fwd2.hip: item_of)
    const int xcd = V & 7, idx = V >> 3, q8 = n_items >> 3, r8 = n_items & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
}

typedef __attribute__((ext_vector_type(16))) float s16f_t;

// the 16 floats at GTA_VREP_INV of view record (b, n): M with Aq = (M (.) m)^T, Oq = M (.) m (gta_flash_common.h: qrec_src)
GTA_DEV s16f_t sload_view(const float* vrep_q, int b, int Nq, int n) {
    const float* ptr = vrep_q + ((long)b * Nq + n) * GTA_VREP_STRIDE + GTA_VREP_INV;
    s16f_t m;
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(m) : "s"(ptr) : "memory");
    return m;
}

// q' = qs (M (.) m)^T q on one 4-vector; m = [[1,1,1,c],[1,1,1,c],[1,1,1,c],[0,0,0,1]] (gta.py:40-44): row 3 of M (.) m is
// (0,0,0,M33), column 3 carries c.  qs_c = qs * c, qs_m33 = qs * M33 (wave-uniform VGPRs)
GTA_DEV void se3_q(const s16f_t& M, float qs, float qs_c, float qs_m33, const float* x, float* y) {
    const float a = x[0], b = x[1], c = x[2], d = x[3];
    y[0] = (M[0] * a + M[4] * b + M[8] * c) * qs;
    y[1] = (M[1] * a + M[5] * b + M[9] * c) * qs;
    y[2] = (M[2] * a + M[6] * b + M[10] * c) * qs;
    y[3] = (M[3] * a + M[7] * b + M[11] * c) * qs_c + qs_m33 * d;
}
// o = (M (.) m) o~ on one 4-vector
GTA_DEV void se3_o(const s16f_t& M, float tc, const float* x, float* y) {
    const float a = x[0], b = x[1], c = x[2], dt = x[3] * tc;
    y[0] = M[0] * a + M[1] * b + M[2] * c + M[3] * dt;
    y[1] = M[4] * a + M[5] * b + M[6] * c + M[7] * dt;
    y[2] = M[8] * a + M[9] * b + M[10] * c + M[11] * dt;
    y[3] = M[15] * x[3];
}

// acc + a.lo * b.lo + a.hi * b.hi (v_dot2c_f32_bf16).  Through the BUILTIN, never inline asm: a dot instruction's result needs 3 wait states
// before a different VALU instruction may read it (gfx940 hazard table), which hipcc inserts only for instructions it knows -- the asm form
// returned row sums that lacked their last terms whenever an add followed too closely (r06: outputs up to 1.6x too large, schedule-dependent)
GTA_DEV float dot2_bf16(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), acc, false);
}

// Full path of the lazy softmax (gta_fwd2.hip: softmax_rebase): true row max of S' (= S - m_run), move m_run there, rescale l and O,
// re-base S' and the -m splat.  key of register r = kbase + (r & 3) + 8 (r >> 2) (+ 32 for s1).
GTA_DEV void c_rebase(f32x16_t& s0, f32x16_t& s1, float& m_run, float& l_run, f32x16_t (&oacc)[C_DB], f32x16_t& msplat,
                      bool first, bool mask, int kbase, int Tk) {
    if (mask) {
        asm volatile("" : "+v"(kbase));        // (keeps the 32 lane-dependent compares inside this rare path: hoisted, their masks are SGPR spills)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kbase + (r & 3) + 8 * (r >> 2);
            if (key >= Tk) s0[r] = -1e30f;
            if (key + 32 >= Tk) s1[r] = -1e30f;
        }
    }
    float mx = s0[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s1[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float delta = first ? mx : fmaxf(mx, 0.f);
    const float alpha = __builtin_amdgcn_exp2f(-delta);
    m_run += delta;
    l_run *= alpha;
#pragma unroll
    for (int d = 0; d < C_DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] -= delta; s1[r] -= delta; msplat[r] = -m_run; }
}
// the masked last key tile whose valid keys are whole 8-key groups (CLEVR-TR: 600 = 9 x 64 + 24): the dead registers are the same in
// every lane (gta_fwd2.hip: mask_tail8)
GTA_DEV void c_mask_tail8(f32x16_t& s0, f32x16_t& s1, int g) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q >= g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s0[4 * q + i] = -1e30f;
        }
        if (q + 4 >= g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s1[4 * q + i] = -1e30f;
        }
    }
}

// transpose-reads of one 16-key slab of V' (stage ST) for both channel blocks
template <int ST, int SL>
GTA_DEV void c_vreads(const uint32_t (&voff)[C_DB][2], u32x2_t (&vl)[C_DB], u32x2_t (&vh)[C_DB]) {
    if (GTA_FWDC_ABL & 8) {
#pragma unroll
        for (int d = 0; d < C_DB; ++d) { vl[d] = u32x2_t{voff[d][0], 0u}; vh[d] = u32x2_t{voff[d][1], 0u}; }
        return;
    }
#pragma unroll
    for (int d = 0; d < C_DB; ++d) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vl[d]) : "v"(voff[d][0]), "n"(ST * C_STAGE + SL * 16 * C_CHP * 16));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vh[d]) : "v"(voff[d][1]), "n"(ST * C_STAGE + SL * 16 * C_CHP * 16));
    }
}
// O^T += V'^T P^T for one slab: two independent accumulators
GTA_DEV void c_pv(const u32x2_t (&vl)[C_DB], const u32x2_t (&vh)[C_DB], const bf16x8_t& pf, f32x16_t (&oacc)[C_DB]) {
    if (GTA_FWDC_ABL & 4) { asm volatile("" :: "v"(vl[0]), "v"(vh[0]), "v"(vl[1]), "v"(vh[1]), "v"(pf)); return; }
#pragma unroll
    for (int d = 0; d < C_DB; ++d) {
        const u32x4_t av = {vl[d].x, vl[d].y, vh[d].x, vh[d].y};
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pf, oacc[d], 0, 0, 0);
    }
}
GTA_DEV void c_ready(u32x2_t (&vl)[C_DB], u32x2_t (&vh)[C_DB]) { asm volatile("" : "+v"(vl[0]), "+v"(vh[0]), "+v"(vl[1]), "+v"(vh[1])); }

// COAL: the wave's 32 rows of a [rows x 128 B] array (Q rows, (cos, sin) rows) -> 4 KiB of LDS at `lds` by LDS-DMA, four 1-KiB accesses (lane ->
// row 8 i + (lane >> 3), position lane & 7); the per-lane SOURCE unit carries the inverse of the images' 8-unit rotation, so the rows lie in
// LDS as a tile image's do (conflict-free fragment reads: swz<8>).  `row0` = first row, rows clamped to n_rows - 1, `stride` in bytes.
GTA_DEV void c_dma_rows32(uint32_t lds, const char* base, int row0, int n_rows, unsigned stride, int lane) {
    const int rr = lane >> 3, pos = lane & 7;
    unsigned vo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int t = row0 + 8 * i + rr;
        t = t < n_rows ? t : n_rows - 1;
        const int gu = (pos - swz_rot<C_CHP>(8 * i + rr)) & 7;
        vo[i] = (unsigned)t * stride + (unsigned)gu * 16u;
    }
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5"
                 ::"s"(lds), "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3]), "s"(base) : "memory", "scc");
}

// kernel arguments through laundered pointers to the kernarg segment, one per code region (gta_fwd2.hip: with by-value access hipcc keeps every
// field the item loop touches in an SGPR across the tile loop and spills them into VGPR lanes)
typedef const __attribute__((address_space(4))) GtaFwdParams* CArgs;
GTA_DEV CArgs c_kargs() {
    CArgs a = (CArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a));
    return a;
}

template <int LAYOUT>
__global__ __launch_bounds__(256, 3) void gta_fwdc_kernel(const GtaFwdParams p_kernarg) {
    static_assert(LAYOUT == GTA_LAYOUT_CL || LAYOUT == GTA_LAYOUT_SO2, "dh = 64 layouts");
    constexpr int NSE3 = LAYOUT == GTA_LAYOUT_CL ? 4 : 0;          // chunks 0 .. NSE3-1 are se3 chunks, the rest so2 chunks
    constexpr int NSO = C_CHP - NSE3;
    constexpr bool COAL = LAYOUT == GTA_LAYOUT_CL && !(GTA_FWDC_ABL & 512);                 // coalesced item I/O through LDS (below); the pure-so2 layout's 256-B (cos, sin) rows do not fit the scratch
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CArgs pp = c_kargs();
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_items = pp->n_items, G = gridDim.x;
    const int n_tiles = (pp->Tk + BN - 1) / BN;
    const int rem = pp->Tk & (BN - 1);
    const bool tail8 = rem != 0 && (rem & 7) == 0;                 // masked last tile on the lazy path
    const bool tail_any = rem != 0 && !tail8;                      // ... on the full path (per-register mask)
    // (wave-uniform: pinned to SGPRs -- a uniform value left in a VGPR is a register the tile loop cannot have)
    const float qs = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pp->scale * LOG2E / (pp->tau ? *pp->tau : 1.0f))));
    const float tc = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pp->trans_coeff ? *pp->trans_coeff : 1.0f)));
    const uint32_t ring = lds_addr(smem);
    const uint32_t dma_lds = ring + wave * 4096;                   // this wave's four 1-KiB pieces of a stage

    for (int V = blockIdx.x; V < n_items; V += G) {
        CArgs pp = c_kargs();
        // (every lane-derived value is re-derived per item from a laundered thread id: left to itself hipcc hoists the lane-dependent
        //  address arithmetic of the whole item out of the item loop and then spills it -- gta_fwd2.hip)
        int tid_i = tid;
        asm volatile("" : "+v"(tid_i));
        const int lane = tid_i & 63, l31 = lane & 31, lh = lane >> 5;
        const unsigned dma_voff = (unsigned)lane * 16u;
        if (pp->prof && tid == 0) { pp->prof[(long)V * 8 + 0] = __builtin_amdgcn_s_memtime(); pp->prof[(long)V * 8 + 5] = __builtin_amdgcn_s_memrealtime(); }
        // ---- item decode (exact below 2^22 items: the dispatch checks) ----
        const int w = c_item_of(V, n_items);
        const int bh = view_of(w, pp->n_qtiles, pp->inv_nqt), qt = w - bh * pp->n_qtiles;
        const int b = view_of(bh, pp->H, pp->invH), h = bh - b * pp->H;
        const int q0 = qt * C_BM;
        const int tE = q0 + wave * 32 + l31;
        const bool rowok = tE < pp->Tq;
        const int my_t = rowok ? tE : pp->Tq - 1;
        const char* kimg = (const char*)pp->kp + (long)bh * n_tiles * (long)C_STAGE + wave * 4096;
        // The item's first two tiles are requested before anything else (they need the scalar item decode only): the Q rows, their (cos, sin) and
        // the key norms then travel beside them, not in front of them.  The ring belongs to this item from here: a workgroup's later items wait
        // for its previous item's last V' reads (persistent grid).
        if (V != (int)blockIdx.x) __builtin_amdgcn_s_barrier();
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072"
                     ::"s"(dma_lds), "v"(dma_voff), "s"(kimg) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072"
                     ::"s"(dma_lds + C_STAGE), "v"(dma_voff), "s"(kimg + C_STAGE) : "memory");
        const char* knext = kimg + 2 * C_STAGE;                    // image of the next tile to request (tile 2)

        // ---- every load of the prologue is requested up front ----
        // COAL (the CLEVR-TR layout: 128-B Q rows, 128-B (cos, sin) rows): the wave's 32 rows travel as four 1-KiB accesses each (lane -> row
        // 8 i + (lane >> 3), 16-B unit lane & 7) and change hands in LDS.  A fragment-direct access (lane = row) touches 32 cache lines with
        // 32 B each; the twelve waves of a CU issue 20 of them per item each, and the tile DMAs queue behind them in the same address path:
        // timing-only builds with coalesced (wrong) addresses ran cl-dec in 87-89 us against 98-104 (profiles/r06/README.md section 1).
        // Scratch: ring stage 2 -- free from the item's start until step 0 requests tile 2 behind its barrier -- 4 KiB per wave, rows in the
        // images' own 8-unit rotation (conflict-free fragment reads); Q first, then the (cos, sin) rows through the same 4 KiB.
        u32x4_t qraw[C_KS];
        f32x4_t qcs[C_KS - NSE3 / 2][2];
        if constexpr (COAL) {
            // Q rows: LDS-DMA into ring stage 2 (4 KiB per wave), requested right behind the tiles; fragments read back below
            c_dma_rows32(ring + 2 * C_STAGE + wave * 4096, (const char*)pp->q + ((long)b * pp->q_sb + (long)h * pp->q_sh) * 2, q0 + wave * 32, pp->Tq,
                         (unsigned)pp->q_st * 2u, lane);
            // (cos, sin) rows: fragment-direct as before (the prologue has one free ring stage: rho_q^-1's copies travel by DMA in the last tile step)
            const float* csrow = pp->cs_q + ((long)b * pp->Tq + my_t) * 2 * pp->nso2;
#pragma unroll
            for (int ks = NSE3 / 2; ks < C_KS; ++ks) {
                const float* cp = csrow + 8 * (2 * ks + lh - NSE3);
                qcs[ks - NSE3 / 2][0] = *reinterpret_cast<const f32x4_t*>(cp);
                qcs[ks - NSE3 / 2][1] = *reinterpret_cast<const f32x4_t*>(cp + 4);
            }
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // the tiles and the Q rows have landed; the four (cos, sin) loads may still fly
            const char* sc = smem + 2 * C_STAGE + wave * 4096;
#pragma unroll
            for (int ks = 0; ks < C_KS; ++ks) qraw[ks] = *reinterpret_cast<const u32x4_t*>(sc + (l31 * C_CHP + swz<C_CHP>(l31, 2 * ks + lh)) * 16);
        } else {
            const char* qrow = (const char*)pp->q + ((long)b * pp->q_sb + (long)h * pp->q_sh + (long)my_t * pp->q_st) * 2 + lh * 16;
#pragma unroll
            for (int ks = 0; ks < C_KS; ++ks) qraw[ks] = *reinterpret_cast<const u32x4_t*>(qrow + ks * 32);
            // (cos, sin) of the lane's so2 chunks 2 ks + lh, ks >= NSE3 / 2: blocks 4 (c - NSE3) .. + 3
            const float* csrow = pp->cs_q + ((long)b * pp->Tq + my_t) * 2 * pp->nso2;
#pragma unroll
            for (int ks = NSE3 / 2; ks < C_KS; ++ks) {
                const float* cp = csrow + 8 * (2 * ks + lh - NSE3);
                qcs[ks - NSE3 / 2][0] = *reinterpret_cast<const f32x4_t*>(cp);
                qcs[ks - NSE3 / 2][1] = *reinterpret_cast<const f32x4_t*>(cp + 4);
            }
        }
        // key norms of the item's tiles: one scalar load per tile step (a per-lane copy read with v_readlane would be one more register through the loop)
        const float* kn_base = pp->kn + (long)bh * n_tiles;

        // ---- rho_q on the lane's chunks -> bf16 MFMA B fragments qf[ks] (chunk 2 ks + lh), |q'| ----
        const int myview = NSE3 ? view_of(my_t, pp->Pq, pp->invPq) : 0;
        const int vfirst = NSE3 ? __builtin_amdgcn_readfirstlane(myview) : 0;
        const int vlast = NSE3 ? __builtin_amdgcn_readlane(myview, 31) : 0;     // (rows ascend with l31; lanes 32..63 repeat them)
        // the 4x4 of the wave's first view stays in SGPRs for the epilogue (the other views of a wave that straddles a boundary are re-read)
        s16f_t Mfirst;
        if constexpr (NSE3 > 0) {
            Mfirst = sload_view(pp->vrep_q, b, pp->Nq, vfirst);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(Mfirst));
        }
        bf16x8_t qf[C_KS];
        float qn;
        {
            u32x4_t qw[C_KS];
            if constexpr (NSE3 > 0) {
                float x[NSE3 / 2][8], y[NSE3 / 2][8];
#pragma unroll
                for (int ks = 0; ks < NSE3 / 2; ++ks) {
                    unpack8(qraw[ks], x[ks]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) y[ks][i] = 0.f;
                }
                const float qs_c = qs * tc;
                for (int n = vfirst; n <= vlast; ++n) {
                    s16f_t M = Mfirst;
                    if (n != vfirst) {
                        M = sload_view(pp->vrep_q, b, pp->Nq, n);
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(M));
                    }
                    const float qs_m33 = qs * M[15];
                    const bool mine = n == vfirst || myview == n;
#pragma unroll
                    for (int ks = 0; ks < NSE3 / 2; ++ks) {
                        float t[8];
                        se3_q(M, qs, qs_c, qs_m33, x[ks], t);
                        se3_q(M, qs, qs_c, qs_m33, x[ks] + 4, t + 4);
#pragma unroll
                        for (int i = 0; i < 8; ++i) y[ks][i] = mine ? t[i] : y[ks][i];
                    }
                }
#pragma unroll
                for (int ks = 0; ks < NSE3 / 2; ++ks) qw[ks] = pack8(y[ks]);
            }
#pragma unroll
            for (int ks = NSE3 / 2; ks < C_KS; ++ks) {
                float x[8];
                unpack8(qraw[ks], x);
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
                    const f32x4_t cs = qcs[ks - NSE3 / 2][hlf];
                    const float c0 = cs.x * qs, s0_ = cs.y * qs, c1 = cs.z * qs, s1_ = cs.w * qs;
                    float* xx = x + 4 * hlf;
                    const float a0 = xx[0], b0 = xx[1], a1 = xx[2], b1 = xx[3];
                    xx[0] = c0 * a0 - s0_ * b0; xx[1] = s0_ * a0 + c0 * b0;
                    xx[2] = c1 * a1 - s1_ * b1; xx[3] = s1_ * a1 + c1 * b1;
                }
                qw[ks] = pack8(x);
            }
            float qsq = 0.f;                                   // |q'_row|^2 over this lane's chunks, from the values the MFMA will see
#pragma unroll
            for (int ks = 0; ks < C_KS; ++ks) {
                qsq = dot2_bf16(qw[ks].x, qw[ks].x, qsq); qsq = dot2_bf16(qw[ks].y, qw[ks].y, qsq);
                qsq = dot2_bf16(qw[ks].z, qw[ks].z, qsq); qsq = dot2_bf16(qw[ks].w, qw[ks].w, qsq);
                qf[ks] = __builtin_bit_cast(bf16x8_t, qw[ks]);
            }
            qsq += __shfl_xor(qsq, 32);                        // the row's other chunk parity
            qn = sqrtf(qsq) * 1.0005f;
        }

        // lane-constant LDS offsets (stage 0): K' fragment (row l31, unit 2 ks + lh; rows 32..63: + 32 * CHP * 16),
        // V' transpose-read (key row 4 lh + (p16 >> 2) (+ 8), channel unit of block d; slab s: + s * 16 * CHP * 16)
        uint32_t koff[C_KS], voff[C_DB][2];
#pragma unroll
        for (int ks = 0; ks < C_KS; ++ks) koff[ks] = ring + (l31 * C_CHP + swz<C_CHP>(l31, 2 * ks + lh)) * 16;
        {
            const int g16 = lane >> 4, p16 = lane & 15;
#pragma unroll
            for (int d = 0; d < C_DB; ++d) {
                const int u = 4 * d + 2 * (g16 & 1) + ((p16 & 3) >> 1);
                const int hb = (p16 & 1) * 8;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int r = 4 * lh + (p16 >> 2) + 8 * hf;
                    voff[d][hf] = ring + C_IMG + (r * C_CHP + swz<C_CHP>(r, u)) * 16 + hb;
                }
            }
        }
        if (pp->prof && tid == 0) pp->prof[(long)V * 8 + 2] = __builtin_amdgcn_s_memtime();      // rho_q done
        f32x16_t oacc[C_DB];
        float m_run = 0.f, l_run = 0.f;
        f32x16_t msplat;                                       // -m_run in every element: C operand of each tile's first MFMAs
#pragma unroll
        for (int i = 0; i < 16; ++i) msplat[i] = 0.f;
#pragma unroll
        for (int d = 0; d < C_DB; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;

        // ---- one tile step; ST = ring stage of tile j (compile time).  ONE barrier per tile, in front of the step (r06 measured the alternatives on
        // this kernel: the boundary behind the step's QK^T with the next tile's first K' fragments read ahead costs a tile of DMA distance, +11 % loop
        // cycles; two barriers per tile -- request at the top, landed-check behind P V -- +43 %: a barrier among four waves that each share their SIMD
        // with two other workgroups' waves costs ~900 cycles of skew) ----
        auto step = [&](auto STC, int j) {
            constexpr int ST = decltype(STC)::value, STN = (ST + 2) % C_NST;
            const bool last = j == n_tiles - 1;
            uint32_t kn_bits;                 // max_k |k'_k| of tile j: requested here, awaited with the last K' fragments (lgkmcnt(0) there anyway)
            {
                const float* kn_ptr = kn_base + __builtin_amdgcn_readfirstlane(j);
                asm volatile("s_load_dword %0, %1, 0x0" : "=s"(kn_bits) : "s"(kn_ptr) : "memory");
            }
            // tile j has landed (only tile j + 1 may still be in flight), everyone is past tile j - 1
            if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (!(GTA_FWDC_ABL & 1)) __builtin_amdgcn_s_barrier();
            if constexpr (COAL) {
                if (last) {
                    // the item's (cos, sin) rows for rho_q^-1 -> this step's request slot (stage STN), by DMA: no register lives through the step
                    CArgs pp = c_kargs();
                    unsigned zc;
                    asm volatile("s_mov_b32 %0, 0" : "=s"(zc));
                    const int lane_c = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zc));
                    c_dma_rows32(ring + STN * C_STAGE + wave * 4096, (const char*)(pp->cs_q + (long)b * pp->Tq * 32), q0 + wave * 32, pp->Tq, 128u, lane_c);
                }
            }
            if (j + 2 < n_tiles && !(GTA_FWDC_ABL & 16)) {
                // (the lane's DMA offset is re-derived per step from nothing -- v_mbcnt on a zero that an asm statement pins inside the step: three
                //  instructions, and neither it nor the thread id lives through the loop)
                unsigned zero;
                asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
                const unsigned dma_voff = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero)) * 16u;
                asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                             "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072"
                             ::"s"(dma_lds), "v"(dma_voff), "s"(knext), "n"(STN * C_STAGE) : "memory", "scc");
                knext += C_STAGE;
            }
            // ---- S^T = K' Q'^T - m ----
            f32x16_t s0, s1;
            {
                bf16x8_t ka[C_KS], kb[C_KS];
#pragma unroll
                for (int ks = 0; ks < C_KS; ++ks) {
                    if (GTA_FWDC_ABL & 8) { ka[ks] = qf[ks]; kb[ks] = qf[ks]; asm volatile("" : "+v"(ka[ks]), "+v"(kb[ks])); continue; }
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ka[ks]) : "v"(koff[ks]), "n"(ST * C_STAGE));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kb[ks]) : "v"(koff[ks]), "n"(ST * C_STAGE + 32 * C_CHP * 16));
                }
                if (GTA_FWDC_ABL & 4) {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ka[0]), "+v"(kb[0]), "+v"(ka[1]), "+v"(kb[1]), "+v"(ka[2]), "+v"(kb[2]), "+v"(ka[3]), "+v"(kb[3]), "+s"(kn_bits));
                    s0 = msplat; s1 = msplat;
                    asm volatile("" : "+v"(s0), "+v"(s1));
                } else {
                // (the scalar load of the key norm is in the same counter and returns out of order: a wait that leaves N outstanding has completed at
                //  least issued - N operations, LDS reads in order -- the fragments a wait names are among them whatever the scalar load does)
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(ka[0]), "+v"(kb[0]));
                // (asm with early-clobber destinations: the splat stays where it is -- no copy into the accumulators)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s0) : "v"(ka[0]), "v"(qf[0]), "v"(msplat));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s1) : "v"(kb[0]), "v"(qf[0]), "v"(msplat));
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ka[1]), "+v"(kb[1]));
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[1], qf[1], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb[1], qf[1], s1, 0, 0, 0);
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ka[2]), "+v"(kb[2]));
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[2], qf[2], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb[2], qf[2], s1, 0, 0, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ka[3]), "+v"(kb[3]), "+s"(kn_bits));
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[3], qf[3], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb[3], qf[3], s1, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // V' slabs 0 and 1 fly under the softmax
            u32x2_t vl[4][C_DB], vh[4][C_DB];
            c_vreads<ST, 0>(voff, vl[0], vh[0]);
            c_vreads<ST, 1>(voff, vl[1], vh[1]);
            // ---- lazy online softmax (gta_fwd2.hip): the bound |q'| max|k'| - m decides; tile 0 of a longer key side and a masked tail
            // of whole 8-key groups stay on the lazy path ----
            {
                const float kn_j = __uint_as_float(kn_bits);
                const bool need = (last && tail_any) || (qn * kn_j - m_run > BOUND_THR);
                if (last && tail8) c_mask_tail8(s0, s1, __builtin_amdgcn_readfirstlane(rem >> 3));
                if (__builtin_amdgcn_ballot_w64(need) != 0) {
                    CArgs pp = c_kargs();
                    unsigned zr;                               // (the rare path derives its lane half afresh: nothing of it lives through the loop)
                    asm volatile("s_mov_b32 %0, 0" : "=s"(zr));
                    const int lane_r = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zr));
                    c_rebase(s0, s1, m_run, l_run, oacc, msplat, j == 0, last && tail_any, j * BN + 4 * (lane_r >> 5), pp->Tk);
                }
            }
            // ---- P = exp2(S') slab by slab (16 keys: 8 scores of this lane), packed to bf16 as the matrix cores will see it, the row sum taken from
            // the PACKED words (one v_dot2c_f32_bf16 per pair against (1.0, 1.0): l counts exactly the probabilities that multiply V', in half the
            // instructions of 32 adds); O^T += V'^T P^T of slab s - 1 sits between the exponentials of slab s and those of slab s + 1: a wave whose
            // stream alternates matrix and vector work overlaps the two pipes, phase-separated streams of co-resident waves do not
            // (tests/probes/probe_mix64.hip).  V' reads stay two slabs ahead. ----
            float rs0 = 0.f, rs1 = 0.f;
            u32x4_t pw[4];
            auto p_slab = [&](const f32x16_t& sv, int t, u32x4_t& w) {
                if (GTA_FWDC_ABL & 2) {
                    w = u32x4_t{__float_as_uint(sv[8 * t]), __float_as_uint(sv[8 * t + 1]), __float_as_uint(sv[8 * t + 2]), __float_as_uint(sv[8 * t + 3])};
                    return;
                }
                float e[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(sv[8 * t + i]);
                w.x = pack_bf16x2(e[0], e[1]); w.y = pack_bf16x2(e[2], e[3]); w.z = pack_bf16x2(e[4], e[5]); w.w = pack_bf16x2(e[6], e[7]);
                rs0 = dot2_bf16(w.x, 0x3f803f80u, rs0); rs1 = dot2_bf16(w.y, 0x3f803f80u, rs1);
                rs0 = dot2_bf16(w.z, 0x3f803f80u, rs0); rs1 = dot2_bf16(w.w, 0x3f803f80u, rs1);
            };
            p_slab(s0, 0, pw[0]);
            c_vreads<ST, 2>(voff, vl[2], vh[2]);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); c_ready(vl[0], vh[0]);
            c_pv(vl[0], vh[0], __builtin_bit_cast(bf16x8_t, pw[0]), oacc);
            p_slab(s0, 1, pw[1]);
            c_vreads<ST, 3>(voff, vl[3], vh[3]);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); c_ready(vl[1], vh[1]);
            c_pv(vl[1], vh[1], __builtin_bit_cast(bf16x8_t, pw[1]), oacc);
            p_slab(s1, 0, pw[2]);
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); c_ready(vl[2], vh[2]);
            c_pv(vl[2], vh[2], __builtin_bit_cast(bf16x8_t, pw[2]), oacc);
            p_slab(s1, 1, pw[3]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); c_ready(vl[3], vh[3]);
            c_pv(vl[3], vh[3], __builtin_bit_cast(bf16x8_t, pw[3]), oacc);
            l_run += rs0 + rs1;
            asm volatile("" : "+v"(l_run));       // (the sum is taken HERE: left alone, hipcc sinks the dot chain towards l's next reader)
        };

        {
            int j = 0;
            for (; j + 3 <= n_tiles; j += 3) {
                step(std::integral_constant<int, 0>{}, j);
                step(std::integral_constant<int, 1>{}, j + 1);
                step(std::integral_constant<int, 2>{}, j + 2);
            }
            if (j < n_tiles) step(std::integral_constant<int, 0>{}, j);
            if (j + 1 < n_tiles) step(std::integral_constant<int, 1>{}, j + 1);
        }

        // ---- epilogue: the accumulators hold, per lane (row l31), half lh (4 channels) of every chunk: channel 32 d + 8 g + 4 lh + i
        // = oacc[d][4 g + i].  rho_q^-1 never straddles a half: applied where the values are, stored as eight 8-byte pieces. ----
        {
            CArgs pp = c_kargs();
            if (pp->prof && wave == 0) pp->prof[(long)V * 8 + 3] = __builtin_amdgcn_s_memtime();  // tile loop done (every lane of wave 0 writes the same word)
            const bool xo = (pp->flags & GTA_FLAG_V_TRANSFORM) != 0;
            // (row, view and lane half are derived afresh from the thread id: kept from the prologue they would live through the tile loop)
            unsigned ze;
            asm volatile("s_mov_b32 %0, 0" : "=s"(ze));
            const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, ze));
            const int lh = lane_e >> 5;
            const int tE = q0 + wave * 32 + (lane_e & 31);
            const bool rowok = tE < pp->Tq;
            const int my_t = rowok ? tE : pp->Tq - 1;
            const int myview = NSE3 ? view_of(my_t, pp->Pq, pp->invPq) : 0;
            // (cos, sin) of the half-chunks the epilogue rotates: half lh of so2 chunk c -> blocks 4 (c - NSE3) + 2 lh, + 1  (requested here: inside the
            // last tile step the 16 registers spill the step's own state -- tried)
            f32x4_t ocs[NSO];
            // COAL: the epilogue's scratch is the two ring stages the last tile does not use -- every wave is past the last step's barrier, i.e.
            // done with the two tiles before the last one; the next item's first requests wait for the barrier at its top.  O goes through
            // stage n_tiles % 3
            char* const sce = smem + (n_tiles % C_NST) * C_STAGE + wave * 4096;
            const int rr = lane_e >> 3, ru = lane_e & 7;
            if constexpr (COAL) {
                // the (cos, sin) rows came by DMA in the last tile step (stage (n_tiles + 1) % 3: that step's request slot)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const char* scs = smem + ((n_tiles + 1) % C_NST) * C_STAGE + wave * 4096;
                const int l31e = lane_e & 31;
#pragma unroll
                for (int i = 0; i < NSO; ++i) ocs[i] = *reinterpret_cast<const f32x4_t*>(scs + (l31e * C_CHP + swz<C_CHP>(l31e, 2 * i + lh)) * 16);
            } else {
                const float* csrow = pp->cs_q + ((long)b * pp->Tq + my_t) * 2 * pp->nso2;
#pragma unroll
                for (int i = 0; i < NSO; ++i) ocs[i] = *reinterpret_cast<const f32x4_t*>(csrow + 8 * i + 4 * lh);
            }
            const float l_tot = l_run + __shfl_xor(l_run, 32);
            const float inv_l = 1.0f / l_tot;
            if (pp->lse && lh == 0 && rowok) pp->lse[((long)b * pp->H + h) * pp->Tq + tE] = (m_run + __log2f(l_tot)) * LN2;
            float o[C_CHP][4];
#pragma unroll
            for (int c = 0; c < C_CHP; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) o[c][i] = oacc[c >> 2][4 * (c & 3) + i] * inv_l;
            if (xo) {
                if constexpr (NSE3 > 0) {
                    float y[NSE3][4];
#pragma unroll
                    for (int c = 0; c < NSE3; ++c)
#pragma unroll
                        for (int i = 0; i < 4; ++i) y[c][i] = 0.f;
                    for (int n = vfirst; n <= vlast; ++n) {
                        s16f_t M = Mfirst;
                        if (n != vfirst) {
                            M = sload_view(pp->vrep_q, b, pp->Nq, n);
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(M));
                        }
                        const bool mine = n == vfirst || myview == n;
#pragma unroll
                        for (int c = 0; c < NSE3; ++c) {
                            float t[4];
                            se3_o(M, tc, o[c], t);
#pragma unroll
                            for (int i = 0; i < 4; ++i) y[c][i] = mine ? t[i] : y[c][i];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NSE3; ++c)
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[c][i] = y[c][i];
                }
#pragma unroll
                for (int c = NSE3; c < C_CHP; ++c) {
                    const f32x4_t cs = ocs[c - NSE3];
                    const float a0 = o[c][0], b0 = o[c][1], a1 = o[c][2], b1 = o[c][3];
                    o[c][0] = cs.x * a0 + cs.y * b0; o[c][1] = cs.x * b0 - cs.y * a0;          // transpose of [[c,-s],[s,c]] (gta.py:270-271)
                    o[c][2] = cs.z * a1 + cs.w * b1; o[c][3] = cs.z * b1 - cs.w * a1;
                }
            }
            if constexpr (COAL) {
                // the lane's eight 8-byte half-chunks -> its row of the scratch (behind the (cos, sin) reads: LDS serves a wave in order), then the
                // wave's 32 rows out in four 1-KiB stores
                const int l31e = lane_e & 31;
#pragma unroll
                for (int c = 0; c < C_CHP; ++c) {
                    u32x2_t wv;
                    wv.x = pack_bf16x2(o[c][0], o[c][1]); wv.y = pack_bf16x2(o[c][2], o[c][3]);
                    *reinterpret_cast<u32x2_t*>(sce + (l31e * C_CHP + swz<C_CHP>(l31e, c)) * 16 + lh * 8) = wv;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = q0 + wave * 32 + 8 * i + rr;
                    const u32x4_t wv = *reinterpret_cast<const u32x4_t*>(sce + ((8 * i + rr) * C_CHP + swz<C_CHP>(8 * i + rr, ru)) * 16);
                    if (t < pp->Tq) *reinterpret_cast<u32x4_t*>((char*)pp->o + ((long)b * pp->o_sb + (long)h * pp->o_sh + (long)t * pp->o_st) * 2 + ru * 16) = wv;
                }
            } else if (rowok) {
                char* orow = (char*)pp->o + ((long)b * pp->o_sb + (long)h * pp->o_sh + (long)tE * pp->o_st) * 2 + lh * 8;
#pragma unroll
                for (int c = 0; c < C_CHP; ++c) {
                    u32x2_t wv;
                    wv.x = pack_bf16x2(o[c][0], o[c][1]); wv.y = pack_bf16x2(o[c][2], o[c][3]);
                    *reinterpret_cast<u32x2_t*>(orow + c * 16) = wv;
                }
            }
            if (pp->prof && wave == 0) {
                pp->prof[(long)V * 8 + 4] = __builtin_amdgcn_s_memtime(); pp->prof[(long)V * 8 + 6] = __builtin_amdgcn_s_memrealtime();
                pp->prof[(long)V * 8 + 1] = (long)(unsigned)__builtin_amdgcn_s_getreg(63492) | ((long)(unsigned)__builtin_amdgcn_s_getreg(63508) << 32);
            }
        }
    }
}

template <int LAYOUT>
int launch_fwdc(const GtaFwdParams& p, hipStream_t stream) {
    const void* kfn = reinterpret_cast<const void*>(&gta_fwdc_kernel<LAYOUT>);
    GtaFwdParams pl = p;
    pl.inv_nqt = 1.0f / (float)p.n_qtiles;
    pl.invH = 1.0f / (float)p.H;
    pl.per_cu = 0;
    long grid = p.n_items;
    {
        // resident workgroups of this instance on this device (gta_fwd2.hip: launch_fwd2): one packed word per device
        static std::atomic<uint64_t> cache[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return GTA_E_NODEVICE;
        long g = 0;
        const uint64_t c = (dev >= 0 && dev < 64) ? cache[dev].load(std::memory_order_acquire) : 0;
        if (c) g = (long)c;
        else {
            int cus = 0, per_cu = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, C_LDS) != hipSuccess || per_cu < 1) per_cu = 1;
            g = (long)cus * per_cu;
            g -= g % 8;
            if (dev >= 0 && dev < 64 && g > 0) cache[dev].store((uint64_t)g, std::memory_order_release);
        }
        // launches of more than one but at most two rounds of resident workgroups (the 600-token CLEVR-TR encoder: 960 items on 768
        // slots) walk the items with a resident grid; every other size: one workgroup per item (profiles/r02, r05)
        const bool few_rounds = p.n_items > g && p.n_items <= 2 * g;
        if (((p.flags & GTA_FLAG_PERSIST) || few_rounds) && g >= 8 && g < grid) grid = g;
    }
#ifdef GTA_ABLATE
    if (const char* e = getenv("GTA_GRID")) { const long g = atol(e); if (g > 0) grid = g < p.n_items ? g : p.n_items; }
#endif
    if (gta_dbg_fwd_ev_start && gta_dbg_fwd_ev_stop) {
        hipExtLaunchKernelGGL((gta_fwdc_kernel<LAYOUT>), dim3((unsigned)grid), dim3(256), C_LDS, stream,
                              (hipEvent_t)gta_dbg_fwd_ev_start, (hipEvent_t)gta_dbg_fwd_ev_stop, 0, pl);
        gta_dbg_fwd_ev_start = gta_dbg_fwd_ev_stop = nullptr;
    } else {
        hipLaunchKernelGGL((gta_fwdc_kernel<LAYOUT>), dim3((unsigned)grid), dim3(256), C_LDS, stream, pl);
    }
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

}  // namespace

// does the dh = 64 instance take this call?  bf16 inputs, dh = 64 exactly, CLEVR-TR or pure-so2 layout, key sides of 2 .. 64 tiles
// (one-tile key sides stay with gta_fwd2_kernel: bit-for-bit agreement with the single-kernel plan), default arithmetic
bool gta_fwdc_takes(const GtaFwdParams& p, int dhp, int layout, int esz) {
    const int n_tiles = (p.Tk + BN - 1) / BN;
    const long n_items = (long)p.B * p.H * ((p.Tq + C_BM - 1) / C_BM);
    return dhp == 64 && p.dh == 64 && esz == 2 && (layout == GTA_LAYOUT_CL || layout == GTA_LAYOUT_SO2) && p.cs_q != nullptr && p.kn != nullptr &&
           (layout != GTA_LAYOUT_CL || p.vrep_q != nullptr) && !(p.flags & (GTA_FLAG_FP32_PRODUCTS | GTA_FLAG_FWD2_GENERIC)) && n_tiles >= 2 &&
           n_tiles <= C_MAX_TILES && n_items < (1L << 22) && p.Tq < (1 << 22) && p.nso2 == (layout == GTA_LAYOUT_CL ? 16 : 32) &&
           // (the coalesced item I/O of the CLEVR-TR layout addresses a (b,h)'s Q rows and a scene's (cos, sin) rows with 32-bit byte offsets)
           (layout != GTA_LAYOUT_CL || ((long)p.Tq * p.q_st * 2 < (1L << 31) && (long)p.Tq * 128 < (1L << 31)));
}
int gta_fwdc_dispatch(const GtaFwdParams& p, int layout, hipStream_t stream) {
    return layout == GTA_LAYOUT_CL ? launch_fwdc<GTA_LAYOUT_CL>(p, stream) : launch_fwdc<GTA_LAYOUT_SO2>(p, stream);
}
