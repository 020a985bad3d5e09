// gta_fwd2.hip -- two-stage GTA attention forward for gfx950: the attention kernel and the dispatch of the plan
// (K/V rep pre-pass: gta_prep.hip).
//
// Why two kernels (measured against the single fused kernel of gta_fwd.hip):
// rho_k acts on K and V per key TOKEN, but a flash kernel re-reads every key tile once per query
// tile, so fusing rho_k re-does the same fp32 VALU work Tq/BM times (10x at the MSN shape) in the
// loop that should be feeding the matrix cores.  Here it is done exactly once:
//
//   gta_kv_prep_kernel   64-key tile per workgroup: raw K,V rows -> LDS by LDS-DMA (coalesced),
//                        lane == key row applies rho_k per 8-channel chunk in fp32 registers,
//                        writes K' and V' as bf16 TILE IMAGES: the exact rotation-swizzled byte
//                        image the attention kernel wants in LDS (gta.py:160-219 for K and V).
//   gta_fwd2_kernel      one work item = 128 query rows of one (b,h): 4 waves x 32 rows; two workgroups per CU (three at
//                        dh = 64).  Prologue: rho on Q (gta.py:165,193,216), prescale, bf16, MFMA B fragments in VGPRs.
//                        Main loop: K'/V' images stream HBM/L2 -> LDS through a 3-stage LDS-DMA ring (linear 1-KiB pieces,
//                        two tiles in flight, counted vmcnt, ONE raw s_barrier per tile, no VGPR staging, no VALU on the
//                        K/V path at all); S^T = K' Q'^T and O^T = V'^T P^T on v_mfma_f32_32x32x16_bf16; V'^T operands
//                        come from the row-major V' image with ds_read_b64_tr_b16; online softmax in registers.
//                        Epilogue in registers: rho_q^-1 per chunk (gta.py:246-276) -> out, LSE.
//                        The kernel body is an ITEM LOOP: by default the grid has one workgroup per item (one iteration).
//                        With GTA_FLAG_PERSIST the resident workgroups walk the items, the DMA ring runs on as ONE stream
//                        across them (the last two tile steps of an item request tiles 0 and 1 of the next), and the
//                        workgroups that share a CU take turns at the raised issue priority, one item each, which keeps
//                        their chains of items the same length.  Measured r02 (profiles/r02/README.md): in SHADER CYCLES
//                        that grid is the fastest form (324k against 338k per launch; 352k without the turns), in
//                        MICROSECONDS it is 3-12 % slower at every BASELINE shape but the 600-token CLEVR-TR encoder --
//                        the better-filled matrix pipes draw more power and the part answers with a lower clock
//                        (1.40-1.66 GHz against 1.66-1.83).  Wall time is what counts: it stays opt-in.
//                        What the rewrite did buy is a spill-free kernel (see the three notes in the body).
// Measured dead ends (r01, profiles/r01/README.md): an explicit ping-pong of an 8-wave kernel, 64 query rows per wave
// with an asm-owned accumulator file (one wave per SIMD), 8-wave workgroups sharing one ring -- none beat two 4-wave
// workgroups per CU; the sources of those variants are in the history (gta_fwd3.hip, removed in r02).
#include <atomic>
#include <cstdlib>
#include <hip/hip_ext.h>
#include "gta_flash_common.h"

int gta_prep_dispatch(const GtaFwdParams& p, int dhp, int esz, hipStream_t stream);                  // gta_prep.hip
bool gta_attn64_takes(const GtaFwdParams& p, int dhp, int layout, int esz);                                  // gta_fwd64.hip
int gta_qtiles_dispatch(const GtaFwdParams& p, hipStream_t stream);
int gta_attn64_dispatch(const GtaFwdParams& p, int esz, int layout, hipStream_t stream);
const char* gta_attn64_kernel_name(const GtaFwdParams& p, int esz, int layout);
bool gta_fwdc_takes(const GtaFwdParams& p, int dhp, int layout, int esz);                                    // gta_fwd_cl.hip
int gta_fwdc_dispatch(const GtaFwdParams& p, int layout, hipStream_t stream);

// profiling hook (not part of the product ABI, see gta_hip.h): events for the NEXT attention-kernel launch of this thread
thread_local void* gta_dbg_fwd_ev_start = nullptr;      // (also read by gta_fwd64.hip)
thread_local void* gta_dbg_fwd_ev_stop = nullptr;
#define g_fwd2_ev_start gta_dbg_fwd_ev_start
#define g_fwd2_ev_stop gta_dbg_fwd_ev_stop
extern "C" void gta_debug_time_next_attention_kernel(void* start_event, void* stop_event) {
    g_fwd2_ev_start = start_event;
    g_fwd2_ev_stop = stop_event;
}
extern "C" void* gta_debug_event_create(void) {
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? (void*)e : nullptr;
}
extern "C" void gta_debug_event_destroy(void* e) { if (e) (void)hipEventDestroy((hipEvent_t)e); }
extern "C" float gta_debug_event_elapsed_ms(void* a, void* b) {
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) return -1.f;
    return ms;
}

namespace {

// X3 = the fp32-faithful instances (GTA_FLAG_FP32_PRODUCTS on the two-stage plan, fp32 inputs, dh <= 64): a tile is FOUR images
// [K'hi | V'hi | K'lo | V'lo] (gta_prep.hip), the ring has two stages (2 x 32 KiB at dh = 64: two workgroups per CU).
template <int DHP, bool X3 = false>
struct Smem2 {
    static constexpr int NW = 4;
    static constexpr int BM = 32 * NW;                  // 128 query rows per work item
    static constexpr int NT = 64 * NW;
    static constexpr int CHP = DHP / 8;
    static constexpr int IMG = BN * DHP * 2;            // one K' or V' tile image
    static constexpr int STAGE = (X3 ? 4 : 2) * IMG;    // K' image then V' image (X3: then their lo parts)
    static constexpr int NST = X3 ? 2 : NSTAGE;         // ring stages
    static constexpr int RING_BYTES = NST * STAGE;
    // layout: [ring | q-side view records, two buffers (item parity) of nrec records each]
    static constexpr int OFF_RING = 0;
    static constexpr int OFF_QREC = RING_BYTES;
    __host__ __device__ static int total(int nrec) { return RING_BYTES + 2 * nrec * GTA_QREC * 4; }
};

// issue the LDS-DMA of one K'/V' tile image pair (STAGE bytes, linear) into ring stage `st` (dma_group: gta_common.h)
template <int DHP, bool X3 = false>
GTA_DEV void dma_stage(char* ring, int st, const char* img, int wave, int lane) {
    using S = Smem2<DHP, X3>;
    constexpr int PIECES = S::STAGE / 1024;             // 1 KiB per wave-instruction
    constexpr int PER_WAVE = PIECES / 4;
    static_assert(PIECES % 4 == 0, "stage must split evenly over the waves");
    const unsigned voff = (unsigned)lane * 16u;
    const char* base = img + wave * (PER_WAVE * 1024);
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(ring + st * S::STAGE + wave * (PER_WAVE * 1024));
    static_for<(PER_WAVE + 3) / 4>([&](auto GC) {
        constexpr int g = decltype(GC)::value, np = PER_WAVE - 4 * g < 4 ? PER_WAVE - 4 * g : 4;
        dma_group<np>(lds + g * 4096, base + g * 4096, voff);
    });
}

// Full path of the lazy softmax (tile 0, masked tail, violated bound): true row max of S' (= S - m_run), move
// m_run there, rescale l and O, re-base S' and the -m splat.  key of register r = kbase + (r&3) + 8(r>>2) (+32).
// The masked last key tile: the rows past Tk are zero rows of the images (score 0, not -inf), so their probabilities must be struck
// from the row sums.  Key of register r = 4 lh + (r & 3) + 8 (r >> 2) (+ 32 for s1): when the tile's valid keys are a whole number g of
// 8-key groups (CLEVR-TR: 600 = 9 x 64 + 24) the dead registers are the same in every lane -- wave-uniform branches and moves, no
// per-register compare / select pairs, and no reason to leave the lazy softmax (other remainders keep the full path's per-register mask).
GTA_DEV void mask_tail8(f32x16_t& s0, f32x16_t& s1, int g) {
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
template <int DHP, bool MASK = true>
GTA_DEV void softmax_rebase(f32x16_t& s0, f32x16_t& s1, float& m_run, float& l_run, f32x16_t (&oacc)[DHP / 32],
                            f32x16_t& msplat, bool first, bool tail, int kbase, int Tk) {
    if (MASK && tail) {                         // (the skewed dh = 96 loop masks here, per register: a call of mask_tail costs that instance 10 spills)
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
    for (int d = 0; d < DHP / 32; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] -= delta; s1[r] -= delta; msplat[r] = -m_run; }
}
// P = exp2(S'), row sum, bf16 MFMA B fragments
GTA_DEV void softmax_exp_pack(f32x16_t& s0, f32x16_t& s1, float& l_run, bf16x8_t (&pf)[2][2]) {
    float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = __builtin_amdgcn_exp2f(s0[r]); rs0 += s0[r]; }
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1[r] = __builtin_amdgcn_exp2f(s1[r]); rs1 += s1[r]; }
    l_run += rs0 + rs1;
    u32x4_t ww;
    ww.x = pack_bf16x2(s0[0], s0[1]); ww.y = pack_bf16x2(s0[2], s0[3]);
    ww.z = pack_bf16x2(s0[4], s0[5]); ww.w = pack_bf16x2(s0[6], s0[7]);
    pf[0][0] = __builtin_bit_cast(bf16x8_t, ww);
    ww.x = pack_bf16x2(s0[8], s0[9]); ww.y = pack_bf16x2(s0[10], s0[11]);
    ww.z = pack_bf16x2(s0[12], s0[13]); ww.w = pack_bf16x2(s0[14], s0[15]);
    pf[0][1] = __builtin_bit_cast(bf16x8_t, ww);
    ww.x = pack_bf16x2(s1[0], s1[1]); ww.y = pack_bf16x2(s1[2], s1[3]);
    ww.z = pack_bf16x2(s1[4], s1[5]); ww.w = pack_bf16x2(s1[6], s1[7]);
    pf[1][0] = __builtin_bit_cast(bf16x8_t, ww);
    ww.x = pack_bf16x2(s1[8], s1[9]); ww.y = pack_bf16x2(s1[10], s1[11]);
    ww.z = pack_bf16x2(s1[12], s1[13]); ww.w = pack_bf16x2(s1[14], s1[15]);
    pf[1][1] = __builtin_bit_cast(bf16x8_t, ww);
}

// transpose-reads of one 16-key slab of V' for all DB channel blocks (2*DB reads)
template <int DHP, int SLAB>
GTA_DEV void pv_reads_slab(uint32_t vbase, const int (&voff)[DHP / 32][2], u32x2_t (&vlo)[DHP / 32], u32x2_t (&vhi)[DHP / 32]) {
    constexpr int OFF = SLAB * 16 * (DHP / 8) * 16;
#pragma unroll
    for (int d = 0; d < DHP / 32; ++d) {
        vlo[d] = lds_tr16_b64<OFF>(vbase + voff[d][0]);
        vhi[d] = lds_tr16_b64<OFF>(vbase + voff[d][1]);
    }
}
// SLAB-major PV: one slab's fragments multiply into DB independent accumulators
// (a chain on one accumulator would run at the dependent latency instead of the issue rate)
template <int DHP>
GTA_DEV void pv_mfma_slab(const u32x2_t (&vlo)[DHP / 32], const u32x2_t (&vhi)[DHP / 32], const bf16x8_t (&pf)[2][2],
                          int kb, int t, f32x16_t (&oacc)[DHP / 32]) {
#pragma unroll
    for (int d = 0; d < DHP / 32; ++d) {
        const u32x4_t av = {vlo[d].x, vlo[d].y, vhi[d].x, vhi[d].y};
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pf[kb][t], oacc[d], 0, 0, 0);
    }
}

// work item V (virtual workgroup id) -> w: all query tiles of one (b,h) land on one XCD (K'/V' stay in that XCD's L2).
// Virtual ids V = blockIdx.x + k * gridDim.x keep the XCD of blockIdx.x when gridDim.x is a multiple of 8.
GTA_DEV int item_of(int V, int n_items) {
    const int xcd = V & 7, idx = V >> 3, q8 = n_items >> 3, r8 = n_items & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
}

typedef const __attribute__((address_space(4))) GtaFwdParams* KArgs;
GTA_DEV KArgs kargs() {
    KArgs a = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a));
    return a;
}

template <int DHP, int ESZ, int LAYOUT, bool X3 = false>
__global__ __launch_bounds__(256, (X3 ? 2 : DHP <= 64 ? 3 : 2)) void gta_fwd2_kernel(const GtaFwdParams p_kernarg) {
    static_assert(!X3 || (ESZ == 4 && DHP <= 64), "the fp32-faithful two-stage instances: fp32 inputs, dh <= 64");
    using S = Smem2<DHP, X3>;
    // chunk descriptor: a compile-time constant for the shipped layouts (c is constant per unrolled item)
#define GTA_DESC(c) (LAYOUT == GTA_LAYOUT_GENERIC ? pp->ctab[c] : gta_layout_desc(LAYOUT, c))
    constexpr int CHP = S::CHP, KS = DHP / 16, DB = DHP / 32, BM = S::BM;
    constexpr int DMA_PER_WAVE = S::STAGE / 1024 / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // The 83 dwords of arguments are read through LAUNDERED pointers to the kernarg segment, one per code region (kernel
    // setup, item prologue, epilogue, and the rare paths inside lambdas): a field used on both sides of the tile loop is
    // loaded again behind it instead of living in an SGPR across it.  With plain by-value access hipcc keeps every field
    // the item loop touches live across the whole loop: 170 SGPR spills into VGPR lanes, 500 v_readlane.
    KArgs pp = kargs();

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_items = pp->n_items, G = gridDim.x;
    const int n_tiles = (pp->Tk + BN - 1) / BN;
    const int ch_real = pp->dh >> 3;
    char* ring = smem + S::OFF_RING;

    // per work item, when a profile buffer is set (gta_debug_profile_next_attention_kernel): [0] start, [4] end (s_memtime: shader cycles),
    // [5] / [6] start / end by s_memrealtime (100 MHz) -- what bench.py turns into kernel cycles and the granted clock.
    // Instrumented builds add [1] Q loads + records landed, [2] rho_q done, [3] tile loop done, [7] next item's loads issued.
#define GTA_STAMP_ON(V_, k) do { if (pp->prof && tid == 0) pp->prof[(long)(V_) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define GTA_STAMPR(V_, k) do { if (pp->prof && tid == 0) pp->prof[(long)(V_) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#ifdef GTA_ABLATE
#define GTA_STAMP(V_, k) GTA_STAMP_ON(V_, k)
#define GTA_STAMP_HW(V_) do { } while (0)
#else
#define GTA_STAMP(V_, k) do { if ((k) == 0 || (k) == 4) GTA_STAMP_ON(V_, k); } while (0)
    // [1] where the item ran: HW_ID (wave / SIMD / CU / SH / SE fields) | XCC_ID << 32 -- tools/wg_timeline.py rebuilds every CU's occupancy from it
#define GTA_STAMP_HW(V_) do { if (pp->prof && tid == 0) pp->prof[(long)(V_) * 8 + 1] = (long)(unsigned)__builtin_amdgcn_s_getreg(63492) | \
                                  ((long)(unsigned)__builtin_amdgcn_s_getreg(63508) << 32); } while (0)
#endif

    const bool has_tail = (pp->Tk & (BN - 1)) != 0;
    const bool full = ch_real == CHP;    // dh fills the padded head: no per-lane chunk guards (exec save/restore per load)
    constexpr int RAWN = ESZ == 2 ? 1 : 2;
    // chunk descriptor of this lane's chunk 2ks + lh
#define GTA_DL(ks) (lh ? GTA_DESC(2 * (ks) + 1) : GTA_DESC(2 * (ks)))

    // ---- the K'/V' DMA stream: tiles of this workgroup's items in consumption order, ring stage = running index % 3 ----
    int dma_V = blockIdx.x, dma_t = 0, dma_st = 0;
    const char* dma_img = nullptr;
    auto dma_img_of = [&](int V) -> const char* {
        KArgs pp = kargs();
        const int bh = item_of(V, n_items) / pp->n_qtiles;
        return (const char*)pp->kp + (long)bh * n_tiles * (long)S::STAGE;
    };
    if (dma_V < n_items) dma_img = dma_img_of(dma_V);
    auto dma_next = [&](int lane) {
        if (dma_V < n_items) {
            dma_stage<DHP, X3>(ring, dma_st, dma_img + (long)dma_t * S::STAGE, wave, lane);
            dma_st = dma_st == S::NST - 1 ? 0 : dma_st + 1;
            if (++dma_t == n_tiles) {
                dma_t = 0;
                dma_V += G;
                if (dma_V < n_items) dma_img = dma_img_of(dma_V);
            }
        }
    };

    int cons_st = 0;                      // ring stage of the current item's next unconsumed tile
    int par = 0;                          // view-record buffer of the current item
    const float qscale = pp->scale * LOG2E / (pp->tau ? *pp->tau : 1.0f);
    const float tc = pp->trans_coeff ? *pp->trans_coeff : 1.0f;

    for (int V = blockIdx.x; V < n_items; V += G) {
    KArgs pp = kargs();
    // Every lane-derived value is re-derived per item from a laundered thread id.  Left to itself hipcc hoists the
    // lane-dependent address arithmetic of the whole item (DMA source offsets, row pointers, LDS offsets) out of the item
    // loop, ~25 more VGPRs live through the tile loop, and then spills -- among others -- each freshly loaded Q chunk
    // behind a vmcnt(0) (93 scratch stores; none with this).
    int tid_i = tid;
    asm volatile("" : "+v"(tid_i));
    const int lane = tid_i & 63, l31 = lane & 31, lh = lane >> 5;
    GTA_STAMP(V, 0); GTA_STAMPR(V, 5);
    // Persistent grid (GTA_FLAG_PERSIST): the workgroups that share a CU take turns at the raised issue priority, one item each.  Without it
    // the one that started first wins the age-based arbitration (MI355X_MICROARCH.md, "two waves per SIMD") on EVERY item,
    // its chain of items ends at 80 % of the kernel's span and the CU idles half-empty behind it (measured r02: 352k ->
    // 324k cycles).  Which workgroups share a CU is not architecturally defined; observed: an XCD deals its consecutive
    // workgroups over its 32 CUs, so workgroup L sits in "slot" (L >> 8) of its CU.  A wrong guess costs nothing.
    if (G < n_items) {
        const int per_cu = pp->per_cu > 1 ? pp->per_cu : 2;
        const int turn = ((V - (int)blockIdx.x) / G + ((int)blockIdx.x >> 8)) % per_cu;
        if (__builtin_amdgcn_readfirstlane(turn) == 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
    // ---- prologue, first half: every load of the item is requested up front ----
    // (defined on every path: a variable of the item loop's body that is only conditionally assigned becomes a
    //  loop-carried value -- "whatever the last iteration left" -- and then lives through the tile loop)
    u32x4_t qraw[KS][RAWN];
    f32x2_t qcs[KS][4];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int k2 = 0; k2 < RAWN; ++k2) qraw[ks][k2] = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) qcs[ks][i] = f32x2_t{1.f, 0.f};
    }
    QrecItem rb0;
    rb0.val = 0.f; rb0.kind = 1; rb0.rr = 0; rb0.cc = 0; rb0.dst = -1;
    int b, h, q0, n_first, n_cnt, my_t;
    {
        const int w = item_of(V, n_items);
        const int bh = w / pp->n_qtiles, qt = w - bh * pp->n_qtiles;
        b = bh / pp->H; h = bh - b * pp->H; q0 = qt * BM;
        // views touched by this query tile: records are staged relative to n_first
        const int t_last = (q0 + BM - 1 < pp->Tq ? q0 + BM - 1 : pp->Tq - 1);
        n_first = q0 / pp->Pq;
        n_cnt = t_last / pp->Pq - n_first + 1;
        // fragment-direct: lane (l31, lh) of a wave IS the owner of MFMA B fragment (row 32*wave + l31, chunks 2ks + lh),
        // and rho_q acts inside a chunk -- so the lane loads exactly those raw chunks.  The chunk kind of lanes 0-31 (even
        // chunk) and 32-63 (odd chunk) is the same constant for every se3/se3, so3/so3, so2/so2 pair of the shipped layouts
        // (the select folds); a mixed pair runs both kinds under exec masks.
        my_t = q0 + wave * 32 + l31;
        my_t = my_t < pp->Tq ? my_t : pp->Tq - 1;
        const char* qrow = (const char*)pp->q + ((long)b * pp->q_sb + (long)h * pp->q_sh + (long)my_t * pp->q_st) * ESZ + lh * 8 * ESZ;
        const float* csrow = pp->cs_q ? pp->cs_q + ((long)b * pp->Tq + my_t) * 2 * pp->nso2 : nullptr;
        auto q_loads = [&](auto FULLC) {
            constexpr bool FULL = decltype(FULLC)::value;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (FULL || 2 * ks + lh < ch_real) {
#pragma unroll
                    for (int k2 = 0; k2 < RAWN; ++k2) qraw[ks][k2] = *reinterpret_cast<const u32x4_t*>(qrow + ks * 16 * ESZ + 16 * k2);
                    if (csrow) load_cs(GTA_DL(ks), csrow, qcs[ks]);
                }
            }
        };
        if (full) q_loads(std::true_type{}); else q_loads(std::false_type{});
        if (pp->vrep_q) qrec_seg_load(rb0, pp->vrep_q, b, pp->Nq, n_first, n_cnt, wave, lane);
        if (V == (int)blockIdx.x) {          // the stream's first NSTAGE - 1 tiles (later ones: requested by the tile steps)
#pragma unroll
            for (int i0 = 0; i0 < S::NST - 1; ++i0) dma_next(lane);
        }
    }
    float* qrec = reinterpret_cast<float*>(smem + S::OFF_QREC) + par * pp->nrec * GTA_QREC;
    // ---- prologue, second half: view records -> LDS (this item's buffer), rho_q on the lane's chunks -> qf ----
    if (pp->vrep_q) {
        const float rs = qscale;          // (the matrices that act on Q carry the score scale: see q_xform)
        qrec_seg_store(rb0, qrec, tc, rs);
        for (int idx = lane + 64; idx < qrec_seg_count(wave, n_cnt); idx += 64) {       // (more than one view per tile only)
            QrecItem it;
            qrec_seg_load(it, pp->vrep_q, b, pp->Nq, n_first, n_cnt, wave, idx);
            qrec_seg_store(it, qrec, tc, rs);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    GTA_STAMP(V, 1);
    bf16x8_t qf[KS];
    bf16x8_t qfl[X3 ? KS : 1];     // (X3) lo parts: q' = hi + lo to 2^-17
    float qn;                      // |q'| of this lane's MFMA row: with the pre-pass's per-tile max |k'| it bounds every score of a tile
    {
        float qsq = 0.f;                                   // this lane's share of |q'_row|^2 (bf16-rounded values)
        const float* rec_q = qrec + (view_of(my_t, pp->Pq, pp->invPq) - n_first) * GTA_QREC;
        auto q_xform = [&](auto FULLC) {
            constexpr bool FULL = decltype(FULLC)::value;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float x[1][8];
                if (FULL || 2 * ks + lh < ch_real) {
                    if (ESZ == 2) {
                        unpack8(qraw[ks][0], x[0]);
                    } else {
#pragma unroll
                        for (int k2 = 0; k2 < RAWN; ++k2) {
                            x[0][4 * k2 + 0] = __uint_as_float(qraw[ks][k2].x); x[0][4 * k2 + 1] = __uint_as_float(qraw[ks][k2].y);
                            x[0][4 * k2 + 2] = __uint_as_float(qraw[ks][k2].z); x[0][4 * k2 + 3] = __uint_as_float(qraw[ks][k2].w);
                        }
                    }
                    if (GTA_DL(ks)) chunk_apply<false, 1>(GTA_DL(ks), rec_q + GTA_QREC_A, rec_q + GTA_QREC_D1, rec_q + GTA_QREC_D2, qcs[ks], x);
                    // (the q-side matrices were staged pre-multiplied by qscale: only identity / so2 halves still need it)
                    {
                        const uint32_t dl = GTA_DL(ks);
                        const bool m_lo = (dl & GTA_CHUNK_SO3) || cd_lo(dl) == GTA_HALF_SE3, m_hi = (dl & GTA_CHUNK_SO3) || cd_hi(dl) == GTA_HALF_SE3;
                        if (!m_lo) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) x[0][i] *= qscale;
                        }
                        if (!m_hi) {
#pragma unroll
                            for (int i = 4; i < 8; ++i) x[0][i] *= qscale;
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[0][i] = 0.f;
                }
                const u32x4_t qw = pack8(x[0]);
                qf[ks] = __builtin_bit_cast(bf16x8_t, qw);
                if constexpr (X3) {
                    float xr[8];
                    unpack8(qw, xr);
#pragma unroll
                    for (int i = 0; i < 8; ++i) xr[i] = x[0][i] - xr[i];
                    qfl[ks] = __builtin_bit_cast(bf16x8_t, pack8(xr));
                }
                // |q'|^2 from the fp32 values; rounding to bf16 moves a value by at most 2^-8 of itself (covered by the factor below)
#pragma unroll
                for (int i = 0; i < 8; ++i) qsq += x[0][i] * x[0][i];
            }
        };
        if (full) q_xform(std::true_type{}); else q_xform(std::false_type{});
        qsq += __shfl_xor(qsq, 32);                                          // the row's other chunk parity
        qn = sqrtf(qsq) * 1.0041f;
    }
    GTA_STAMP(V, 2);

    // lane-constant LDS offsets (per item, see above).  The rotation swizzle has period 16 rows, so a fragment of rows
    // r + 16m sits at the same in-row position: per-slab offsets are compile-time immediates.
    const int lane_i = lane;
    int koff[KS];            // K' fragment: row l31, unit 2ks+lh  (rows 32.. : + 32*CHP*16)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = ((lane_i & 31) * CHP + swz<CHP>(lane_i & 31, 2 * ks + (lane_i >> 5))) * 16;
    int voff[DB][2];         // V' transpose-read: key row 4lh + (p16>>2) (+8), channel unit of block d
    {
        const int g16 = lane_i >> 4, p16 = lane_i & 15;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            const int u = 4 * d + 2 * (g16 & 1) + ((p16 & 3) >> 1);
            const int hb = (p16 & 1) * 8;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int r = 4 * (lane_i >> 5) + (p16 >> 2) + 8 * hf;
                voff[d][hf] = (r * CHP + swz<CHP>(r, u)) * 16 + hb;
            }
        }
    }
    f32x16_t oacc[DB];
    float m_run = 0.f, l_run = 0.f;
    f32x16_t msplat;                      // -m_run in every element: C operand of each tile's first MFMA (S' = S - m)
#pragma unroll
    for (int i = 0; i < 16; ++i) msplat[i] = 0.f;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
    const float* kn_base = pp->kn + (long)(b * pp->H + h) * n_tiles;

    static_assert(!(PIPE1 && DHP == 96) || NSTAGE == 3, "the skewed loop keeps K'(j+1), V'(j) and one tile in flight: three stages");
    if constexpr (X3) {
    // ---- fp32-faithful tile loop: every product of the two contractions as three MFMAs on bf16 (hi, lo) operand pairs, small terms
    // first (lo*hi + hi*lo + hi*hi, as gta_fwd_kernel<..., x3>); P = exp2(S') is split the same way; softmax, row sums, O in fp32.
    // Plain order (QK^T, softmax, P V per tile); two ring stages: tile j + 1 streams in under tile j's 96 matrix instructions. ----
    for (int j = 0; j < n_tiles; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // tile j has landed (nothing else is in flight), everyone is past tile j - 1
        __builtin_amdgcn_s_barrier();
        dma_next(lane);
        const char* kf = ring + cons_st * S::STAGE;
        cons_st = cons_st == S::NST - 1 ? 0 : cons_st + 1;
        uint32_t kn_bits;
        {
            const float* kn_ptr = kn_base + __builtin_amdgcn_readfirstlane(j);
            asm volatile("s_load_dword %0, %1, 0x0" : "=s"(kn_bits) : "s"(kn_ptr) : "memory");
        }
        f32x16_t s[2];
        {
            bf16x8_t kh[KS][2], kl[KS][2];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    kh[ks][hh] = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks] + hh * 32 * CHP * 16);
                    kl[ks][hh] = *reinterpret_cast<const bf16x8_t*>(kf + 2 * S::IMG + koff[ks] + hh * 32 * CHP * 16);
                }
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) s[hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl[0][hh], qf[0], msplat, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    if (ks > 0) s[hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl[ks][hh], qf[ks], s[hh], 0, 0, 0);
                    s[hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[ks][hh], qfl[ks], s[hh], 0, 0, 0);
                }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) s[hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[ks][hh], qf[ks], s[hh], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(kn_bits));
        bf16x8_t pf[2][2], pl[2][2];
        {
            const float kn_j = __uint_as_float(kn_bits);
            const bool tail = has_tail && j == n_tiles - 1;
            const bool need = (j == 0) || tail || (qn * kn_j - m_run > BOUND_THR);
            if (__builtin_amdgcn_ballot_w64(need) != 0)
                softmax_rebase<DHP>(s[0], s[1], m_run, l_run, oacc, msplat, j == 0, tail, j * BN + 4 * lh, pp->Tk);
            softmax_exp_pack(s[0], s[1], l_run, pf);               // s now holds P in fp32
            // lo = bf16(P - float(hi))
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float xh[8], xl[8];
                    unpack8(__builtin_bit_cast(u32x4_t, pf[hh][t]), xh);
#pragma unroll
                    for (int i = 0; i < 8; ++i) xl[i] = s[hh][8 * t + i] - xh[i];
                    pl[hh][t] = __builtin_bit_cast(bf16x8_t, pack8(xl));
                }
        }
        // ---- O^T += V'^T P^T, slab by slab: (V'lo P_hi + V'hi P_lo) + V'hi P_hi ----
        const uint32_t vb_h = lds_addr(kf + S::IMG), vb_l = lds_addr(kf + 3 * S::IMG);
        static_for<4>([&](auto SC) {
            constexpr int sl = decltype(SC)::value, kb = sl >> 1, t = sl & 1;
            u32x2_t hl[DB], hh_[DB], ll[DB], lh_[DB];
            pv_reads_slab<DHP, sl>(vb_h, voff, hl, hh_);
            pv_reads_slab<DHP, sl>(vb_l, voff, ll, lh_);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const u32x4_t ah = {hl[d].x, hl[d].y, hh_[d].x, hh_[d].y}, al = {ll[d].x, ll[d].y, lh_[d].x, lh_[d].y};
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, al), pf[kb][t], oacc[d], 0, 0, 0);
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ah), pl[kb][t], oacc[d], 0, 0, 0);
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ah), pf[kb][t], oacc[d], 0, 0, 0);
            }
        });
    }
    } else
    if constexpr (PIPE1 && DHP == 96) {      // (dh = 64: 230 VGPRs would cost the third workgroup per CU)
    // ---- skewed tile loop: the QK^T MFMAs of tile j+1 issue beside the softmax VALU of tile j ----
    // Two of these waves share a SIMD (two workgroups per CU).  Measured (tests/probes/probe_coissue.hip): two
    // waves whose streams each mix MFMA and VALU reach the matrix-pipe rate together, while the unskewed loop
    // below (all MFMAs, then all VALU, per wave) leaves the pipes idle half the time.
    //   A(j): S'(j+1) = K'(j+1) Q'^T - m   ||  decision(j); exp / sum / pack of P(j); K' fragments 2 steps ahead
    //   B(j): O += V'(j) P(j), slab-major  ||  V'(j) transpose-reads one slab ahead
    f32x16_t sA[2], sB[2];
    u32x4_t pfr[2][2];                                // P as packed bf16 words: [key half][slab in half]
    float rs0 = 0.f, rs1 = 0.f;                       // row-sum halves (even / odd values)
    constexpr int NEV = 32;                           // P values per lane and tile: e -> (half = e >> 4, r = e & 15)
    constexpr int GA = 2 * KS;                        // MFMAs of A: g -> (ks = g >> 1, half = g & 1)
    constexpr int KLA = 2;                      // K' fragment reads run this many k steps ahead of their MFMAs
    auto e_first = [](int g) constexpr { return g * NEV / GA; };
    auto s_fence1 = [&](f32x16_t (&s)[2]) { asm volatile("" : "+v"(s[0])); asm volatile("" : "+v"(s[1])); };
    auto step = [&](f32x16_t (&sc)[2], f32x16_t (&sn)[2], int j, auto LASTC) {
        constexpr bool LAST = decltype(LASTC)::value;
        // the tile's key-norm bound is asked for BEFORE the waits below, so it is there when they are over (measured:
        // awaiting it together with the first K' fragment reads cost 4 % of the loop's cycles)
        uint32_t kn_bits;
        {
            const float* kn_ptr = kn_base + __builtin_amdgcn_readfirstlane(j);
            asm volatile("s_load_dword %0, %1, 0x0" : "=s"(kn_bits) : "s"(kn_ptr) : "memory");
        }
        // tile j+1 has landed, everyone is past B(j-1): its stage takes the stream's next tile (j+2, or a tile of the next item)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        dma_next(lane);
        const int st_next = cons_st == NSTAGE - 1 ? 0 : cons_st + 1;
        const char* kf = ring + st_next * S::STAGE;                     // K'(j+1)
        const uint32_t vbase = lds_addr(ring + cons_st * S::STAGE + S::IMG);   // V'(j)
        cons_st = st_next;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(kn_bits));           // (nothing else is outstanding on that counter here)
        bf16x8_t kfr[KS][2];
        auto k_load = [&](auto KC) {
            constexpr int ks = decltype(KC)::value;
            if constexpr (!LAST && ks < KS) {
                kfr[ks][0] = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks]);
                kfr[ks][1] = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks] + 32 * CHP * 16);
            }
        };
        static_for<KLA>([&](auto KC) { k_load(KC); });
        // decision for tile j (sc = S'(j) relative to m_run): lazy-softmax full path only when needed
        {
            const float kn_j = __uint_as_float(kn_bits);
            const bool tail = has_tail && j == n_tiles - 1;
            const bool need = (j == 0) || tail || (qn * kn_j - m_run > BOUND_THR);
            if (__builtin_amdgcn_ballot_w64(need) != 0) {
                asm volatile("" ::: "memory");
                l_run += rs0 + rs1; rs0 = 0.f; rs1 = 0.f;
                softmax_rebase<DHP>(sc[0], sc[1], m_run, l_run, oacc, msplat, j == 0, tail, j * BN + 4 * lh, pp->Tk);
            }
            s_fence1(sc);
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<GA>([&](auto GC) {
            constexpr int g = decltype(GC)::value, ks = g >> 1, hh = g & 1;
            if constexpr (!LAST) {
                if constexpr (ks == 0) sn[hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0][hh], qf[0], msplat, 0, 0, 0);
                else sn[hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks][hh], qf[ks], sn[hh], 0, 0, 0);
                if constexpr (hh == 0) k_load(std::integral_constant<int, ks + KLA>{});
            }
            // this gap's exps, then the sums and packs of the values finished in earlier gaps
            constexpr int e0 = e_first(g), e1 = e_first(g + 1), ep = g > 0 ? e_first(g - 1) : 0;
            static_for<e1 - e0>([&](auto DC) {
                constexpr int e = e0 + decltype(DC)::value;
                sc[e >> 4][e & 15] = __builtin_amdgcn_exp2f(sc[e >> 4][e & 15]);
            });
            static_for<e0 - ep>([&](auto DC) {
                constexpr int e = ep + decltype(DC)::value;
                if (e & 1) rs1 += sc[e >> 4][e & 15]; else rs0 += sc[e >> 4][e & 15];
            });
            static_for<e0 / 2 - ep / 2>([&](auto DC) {
                constexpr int k = ep / 2 + decltype(DC)::value, e = 2 * k;
                pfr[e >> 4][(e & 15) >> 3][(e & 7) >> 1] = pack_bf16x2(sc[e >> 4][e & 15], sc[e >> 4][(e & 15) + 1]);
            });
            if constexpr (g == GA - 1) {                      // tail of the list
                static_for<NEV - e0>([&](auto DC) {
                    constexpr int e = e0 + decltype(DC)::value;
                    if (e & 1) rs1 += sc[e >> 4][e & 15]; else rs0 += sc[e >> 4][e & 15];
                });
                static_for<NEV / 2 - e0 / 2>([&](auto DC) {
                    constexpr int k = e0 / 2 + decltype(DC)::value, e = 2 * k;
                    pfr[e >> 4][(e & 15) >> 3][(e & 7) >> 1] = pack_bf16x2(sc[e >> 4][e & 15], sc[e >> 4][(e & 15) + 1]);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- B(j): O^T += V'^T P^T, slab-major; reads stay one slab ahead (LDS returns in order) ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (no compiler-tracked LDS read is in flight past here)
        u32x2_t v0l[DB], v0h[DB], v1l[DB], v1h[DB], v2l[DB], v2h[DB], v3l[DB], v3h[DB];
        bf16x8_t pfb[2][2];
        pfb[0][0] = __builtin_bit_cast(bf16x8_t, pfr[0][0]); pfb[0][1] = __builtin_bit_cast(bf16x8_t, pfr[0][1]);
        pfb[1][0] = __builtin_bit_cast(bf16x8_t, pfr[1][0]); pfb[1][1] = __builtin_bit_cast(bf16x8_t, pfr[1][1]);
        pv_reads_slab<DHP, 0>(vbase, voff, v0l, v0h);
        pv_reads_slab<DHP, 1>(vbase, voff, v1l, v1h);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DB) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP>(v0l, v0h, pfb, 0, 0, oacc);
        pv_reads_slab<DHP, 2>(vbase, voff, v2l, v2h);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DB) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP>(v1l, v1h, pfb, 0, 1, oacc);
        pv_reads_slab<DHP, 3>(vbase, voff, v3l, v3h);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DB) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP>(v2l, v2h, pfb, 1, 0, oacc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP>(v3l, v3h, pfb, 1, 1, oacc);
        if constexpr (!LAST) s_fence1(sn);
    };
    // tile 0: S'(0) by itself (its image was requested by the previous item's last steps, or at kernel start)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        const char* kf = ring + cons_st * S::STAGE;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t k0 = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks]);
            const bf16x8_t k1 = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks] + 32 * CHP * 16);
            sA[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[ks], ks == 0 ? msplat : sA[0], 0, 0, 0);
            sA[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[ks], ks == 0 ? msplat : sA[1], 0, 0, 0);
        }
    }
    {
        int j = 0;
        for (; j + 2 < n_tiles; j += 2) {
            step(sA, sB, j, std::false_type{});
            step(sB, sA, j + 1, std::false_type{});
        }
        if (j + 2 == n_tiles) {
            step(sA, sB, j, std::false_type{});
            step(sB, sA, j + 1, std::true_type{});
        } else {
            step(sA, sB, j, std::true_type{});
        }
    }
    l_run += rs0 + rs1;
    } else
    for (int j = 0; j < n_tiles; ++j) {
        // tile j has landed (only the stream's next tile may still be in flight), everyone is past tile j-1.
        // (younger requests -- the next item's Q loads, this item's predecessor's stores -- only make the wait longer)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S::NST - 2) * DMA_PER_WAVE) : "memory");
        if (dma_V >= n_items) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the stream has ended: nothing younger to count on)
        __builtin_amdgcn_s_barrier();
        dma_next(lane);
        const char* kf = ring + cons_st * S::STAGE;
        const char* vf = kf + S::IMG;
        cons_st = cons_st == NSTAGE - 1 ? 0 : cons_st + 1;
        uint32_t kn_bits;                 // max_k |k'_k| of tile j (scalar load by hand)
        {
            const float* kn_ptr = kn_base + __builtin_amdgcn_readfirstlane(j);
            asm volatile("s_load_dword %0, %1, 0x0" : "=s"(kn_bits) : "s"(kn_ptr) : "memory");
        }

        // ---- S^T = K' Q'^T ----
        f32x16_t s[2];
        {
            bf16x8_t ka[KS], kb2[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                ka[ks] = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks]);
                kb2[ks] = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks] + 32 * CHP * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[0], qf[0], msplat, 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb2[0], qf[0], msplat, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < KS; ++ks) {
                s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks], qf[ks], s[0], 0, 0, 0);
                s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb2[ks], qf[ks], s[1], 0, 0, 0);
            }
        }
        // the tile's key-norm bound (scalar load from the top of the iteration; the K' reads are consumed, so this
        // wait is free -- it must sit BEFORE the V' reads below or it would drain them too)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(kn_bits));
        // V' slab 0 transpose-reads fly under the softmax
        const uint32_t vbase = lds_addr(vf);
        u32x2_t v0l[DB], v0h[DB], v1l[DB], v1h[DB], v2l[DB], v2h[DB], v3l[DB], v3h[DB];
        pv_reads_slab<DHP, 0>(vbase, voff, v0l, v0h);

        // Lazy online softmax: S' already has -m_run folded in.  |q'| max_k|k'| bounds the tile's scores, so
        // while bound - m_run stays below BOUND_THR no exponent can overflow and neither the row max nor the
        // O rescale is needed; tile 0, the masked tail tile and a violated bound take the full path.
        bf16x8_t pf[2][2];
        {
            // (r05) A masked tail tile whose valid keys are whole 8-key groups is no reason for the full path: its dead keys are struck
            // (mask_tail8) and the lazy softmax's bound decides as for any other tile.  At the CLEVR-TR shapes (10 key tiles, 24 keys in
            // the last) the forced full path was ~250 VALU instructions per item and wave, a ninth of its vector work: -3 % kernel cycles
            // (profiles/r05).  Tile 0 is no reason either when more tiles follow: the state before it (m = 0, l = 0, O = 0, splat = 0) is a
            // valid lazy-softmax state and |S| <= |q'| max|k'| bounds the scores from BOTH sides, so while the bound holds exp2(S) neither
            // overflows nor leaves a row all zero (-2.6 % cycles at cl-dec).  One-tile key sides keep the true row max: there this kernel
            // and the single-kernel plan agree to the last bits (the chunked decode's cached against uncached layers).
            const float kn_j = __uint_as_float(kn_bits);
            const bool tail = has_tail && j == n_tiles - 1;
            const int rem = pp->Tk & (BN - 1);
            const bool tail8 = tail && (rem & 7) == 0;
            const bool need = (j == 0 && n_tiles == 1) || (tail && !tail8) || (qn * kn_j - m_run > BOUND_THR);
            if (tail8) mask_tail8(s[0], s[1], __builtin_amdgcn_readfirstlane(rem >> 3));
            if (__builtin_amdgcn_ballot_w64(need) != 0)
                softmax_rebase<DHP>(s[0], s[1], m_run, l_run, oacc, msplat, j == 0, tail && !tail8, j * BN + 4 * lh, pp->Tk);
            softmax_exp_pack(s[0], s[1], l_run, pf);
        }

        // ---- O^T += V'^T P^T, slab-major; reads stay one slab ahead (LDS returns in order) ----
        pv_reads_slab<DHP, 1>(vbase, voff, v1l, v1h);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DB) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP>(v0l, v0h, pf, 0, 0, oacc);
        pv_reads_slab<DHP, 2>(vbase, voff, v2l, v2h);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DB) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP>(v1l, v1h, pf, 0, 1, oacc);
        pv_reads_slab<DHP, 3>(vbase, voff, v3l, v3h);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DB) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP>(v2l, v2h, pf, 1, 0, oacc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP>(v3l, v3h, pf, 1, 1, oacc);
    }
    GTA_STAMP(V, 3);

    {
    KArgs pp = kargs();
    // ---- epilogue in registers: the accumulators hold, per lane (row l31), the 4-channel HALF lh of every chunk;
    // one v_permlane32_swap per value hands lanes 0-31 the whole even chunk of a pair and lanes 32-63 the whole odd
    // one (the same ownership as the prologue), rho_q^-1 acts inside the chunk, and the chunk is stored.  No O
    // staging tile, no barrier.
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv_l = 1.0f / l_tot;
    const bool xo = (pp->flags & GTA_FLAG_V_TRANSFORM) != 0;
    const int tE = q0 + wave * 32 + l31;
    const bool rowok = tE < pp->Tq;
    const int tC = rowok ? tE : pp->Tq - 1;
    constexpr int NP = CHP / 2;
    f32x2_t ocs[NP][4];
#pragma unroll
    for (int kp = 0; kp < NP; ++kp)
#pragma unroll
        for (int i = 0; i < 4; ++i) ocs[kp][i] = f32x2_t{1.f, 0.f};
    const bool fast = full && q0 + BM <= pp->Tq;          // no ragged channel, no ragged row: no per-lane guards
    const float* csrow_o = (xo && pp->cs_q) ? pp->cs_q + ((long)b * pp->Tq + tC) * 2 * pp->nso2 : nullptr;
#define GTA_CE(kp) (2 * (kp) + lh)
#define GTA_DLE(kp) GTA_DL(kp)
    if (csrow_o) {
#pragma unroll
        for (int kp = 0; kp < NP; ++kp)
            if (fast || GTA_CE(kp) < ch_real) load_cs(GTA_DLE(kp), csrow_o, ocs[kp]);
    }
    GTA_STAMP(V, 7);
    if (pp->lse && lh == 0 && rowok) pp->lse[((long)b * pp->H + h) * pp->Tq + tE] = (m_run + __log2f(l_tot)) * LN2;
    {
        const float* rec_o = qrec + (view_of(tC, pp->Pq, pp->invPq) - n_first) * GTA_QREC;
        char* orow = (char*)pp->o + ((long)b * pp->o_sb + (long)h * pp->o_sh + (long)tC * pp->o_st) * ESZ;
        auto o_items = [&](auto FASTC) {
            constexpr bool FAST = decltype(FASTC)::value;
#pragma unroll
            for (int kp = 0; kp < NP; ++kp) {
                float x[1][8];
                const int d = (2 * kp) >> 2, ge = (2 * kp) & 3;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t ua = __float_as_uint(oacc[d][4 * ge + i] * inv_l);         // half lh of the even chunk
                    const uint32_t ub = __float_as_uint(oacc[d][4 * (ge + 1) + i] * inv_l);   // half lh of the odd chunk
                    const auto sw = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);   // ua[32..63] <-> ub[0..31]
                    x[0][i] = __uint_as_float(sw[0]);
                    x[0][4 + i] = __uint_as_float(sw[1]);
                }
                if (FAST || (rowok && GTA_CE(kp) < ch_real)) {
                    if (xo && GTA_DLE(kp)) chunk_apply<true, 1>(GTA_DLE(kp), rec_o + GTA_QREC_O, rec_o + GTA_QREC_D1T, rec_o + GTA_QREC_D2T, ocs[kp], x);
                    gstore_chunk2<ESZ>(orow + GTA_CE(kp) * 8 * ESZ, 0, x[0]);
                }
            }
        };
        if (fast) o_items(std::true_type{}); else o_items(std::false_type{});
    }
    }
#undef GTA_CE
#undef GTA_DLE
    GTA_STAMP(V, 4); GTA_STAMPR(V, 6); GTA_STAMP_HW(V);
    par ^= 1;
    }
#undef GTA_STAMP
#undef GTA_STAMP_ON
#undef GTA_STAMP_HW
#undef GTA_STAMPR
#undef GTA_DL
#undef GTA_DESC
}

template <int DHP, int ESZ, int LAYOUT, bool X3 = false>
int launch_fwd2(const GtaFwdParams& p, hipStream_t stream) {
    using S = Smem2<DHP, X3>;
    const void* kfn = reinterpret_cast<const void*>(&gta_fwd2_kernel<DHP, ESZ, LAYOUT, X3>);
    if (int rc = gta_lds_optin<&gta_fwd2_kernel<DHP, ESZ, LAYOUT, X3>>(S::total(GTA_MAX_VIEWS))) return rc;
    int lds = S::total(p.vrep_q ? p.nrec : 0);
    // One workgroup per item, or (GTA_FLAG_PERSIST / the few-rounds rule below) a persistent grid of as many workgroups as are resident at once
    // (registers and LDS: two per CU at dh = 96, three at dh = 64), a multiple of 8 so that the virtual ids of a workgroup
    // stay on its XCD.
    GtaFwdParams pl = p;
    long grid = p.n_items;
    pl.per_cu = 0;
    {
        // resident workgroups of this instance on this device: queried once per (device, LDS size) and published as ONE word
        // (lds << 32 | slots << 8 | per_cu), so a racing thread sees a consistent triple or none
        static std::atomic<uint64_t> cache[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return GTA_E_NODEVICE;
        int cus = 0, per_cu = 0;
        long g = 0;
        const uint64_t c = (dev >= 0 && dev < 64) ? cache[dev].load(std::memory_order_acquire) : 0;
        if (c && (int)(c >> 32) == lds) { g = (long)((c >> 8) & 0xffffff); per_cu = (int)(c & 0xff); }      // (the cache is per template instance: a static of this function)
        else {
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
            g = (long)cus * per_cu;
            g -= g % 8;
            if (dev >= 0 && dev < 64 && g > 0 && g < (1 << 24))
                cache[dev].store(((uint64_t)(uint32_t)lds << 32) | ((uint64_t)g << 8) | (uint64_t)(per_cu & 0xff), std::memory_order_release);
        }
        // Persistent grid on request, and by default for launches of more than one but at most two rounds of resident
        // workgroups (the 600-token CLEVR-TR encoder: 960 items on 768 slots): there the second round runs a quarter
        // full, and walking the items with the ring running on is the faster form (35.9 against 39.3 us); at every
        // shape of five and more rounds it is the slower one (profiles/r02/README.md).
        const bool few_rounds = p.n_items > g && p.n_items <= 2 * g;
        if (((p.flags & GTA_FLAG_PERSIST) || few_rounds) && g >= 8 && g < grid) { grid = g; pl.per_cu = per_cu; }
    }
#ifdef GTA_ABLATE
    if (const char* e = getenv("GTA_LDS_PAD")) {        // occupancy experiment: inflate LDS so fewer workgroups share a CU
        lds += atoi(e);
        (void)hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    if (const char* e = getenv("GTA_GRID")) { const long g = atol(e); if (g > 0) grid = g < p.n_items ? g : p.n_items; }
#endif
    if (g_fwd2_ev_start && g_fwd2_ev_stop) {
        // profiling hook (bench.py): start / stop events taken from the dispatch itself -- no marker packets, so the
        // kernel's neighbours in the stream are not pushed apart the way two hipEventRecord calls push them (~3 us each)
        hipExtLaunchKernelGGL((gta_fwd2_kernel<DHP, ESZ, LAYOUT, X3>), dim3((unsigned)grid), dim3(256), lds, stream,
                              (hipEvent_t)g_fwd2_ev_start, (hipEvent_t)g_fwd2_ev_stop, 0, pl);
        g_fwd2_ev_start = g_fwd2_ev_stop = nullptr;
    } else {
        hipLaunchKernelGGL((gta_fwd2_kernel<DHP, ESZ, LAYOUT, X3>), dim3((unsigned)grid), dim3(256), lds, stream, pl);
    }
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

}  // namespace

// (x3: the fp32-faithful two-stage instances store every image twice, hi and lo)
long gta_fwd2_image_bytes(int B, int H, int Tk, int dhp, bool x3) {
    const long n_tiles = (Tk + BN - 1) / BN;
    return (long)B * H * n_tiles * (x3 ? 4L : 2L) * BN * dhp * 2;
}
// workspace = [K'/V' tile images | per-tile key norms | q-side rep tiles (dh = 96: gta_flash_common.h, 24 KiB per (scene, view))]
long gta_fwd2_qtiles_offset(int B, int H, int Tk, int dhp, bool x3) {
    const long n_tiles = (Tk + BN - 1) / BN;
    const long img = (gta_fwd2_image_bytes(B, H, Tk, dhp, x3) + 255) & ~255L;
    return img + (((long)B * H * n_tiles * 4 + 255) & ~255L);
}
// (the q-side tiles exist for the one instance that uses them: bf16 inputs at dh = 96 -- the only part of the workspace whose size
//  depends on the QUERY side, through Nq)
long gta_fwd2_workspace_bytes(int B, int H, int Tk, int dhp, int Nq, int esz, bool x3) {
    return gta_fwd2_qtiles_offset(B, H, Tk, dhp, x3) + (dhp == 96 && esz == 2 ? (long)B * Nq * GTA_QT_TILES * GTA_QT_BYTES : 0L);
}
// the fp32-faithful mode on the two-stage plan: fp32 inputs at dh <= 64 (CLEVR-TR, runs/clevrtr/GTA/gta/config.yaml:55); other head sizes
// keep the single-kernel plan (gta_fwd_kernel<..., x3>)
bool gta_fwd2_x3_takes(int dhp, int esz) { return esz == 4 && dhp <= 64; }
int gta_fwd2_lds_bytes(int dhp, int nrec) {
    switch (dhp) {
        case 32: return Smem2<32>::total(nrec);
        case 64: return Smem2<64>::total(nrec);
        case 96: return Smem2<96>::total(nrec);
        case 128: return Smem2<128>::total(nrec);
    }
    return -1;
}

// which compile-time layout (if any) the run-time chunk table is
static int layout_of(const GtaFwdParams& p, int dhp) {
    const int ch = p.dh / 8;
    if (p.dh != dhp) return GTA_LAYOUT_GENERIC;
    for (int L : {GTA_LAYOUT_MS, GTA_LAYOUT_MSG, GTA_LAYOUT_SE3, GTA_LAYOUT_CL, GTA_LAYOUT_SO2}) {
        if (((L == GTA_LAYOUT_MS || L == GTA_LAYOUT_MSG || L == GTA_LAYOUT_SE3) && dhp != 96) || (L == GTA_LAYOUT_CL && dhp != 64)) continue;
        bool same = true;
        for (int c = 0; c < ch; ++c) same = same && p.ctab[c] == gta_layout_desc(L, c);
        if (same) return L;
    }
    return GTA_LAYOUT_GENERIC;
}

int gta_fwd2_rows_per_item(const GtaFwdParams& p, int dhp, int esz) {
    return !(p.flags & GTA_FLAG_FP32_PRODUCTS) && gta_attn64_takes(p, dhp, layout_of(p, dhp), esz) ? 256 : 128;
}
const char* gta_fwd2_attention_kernel_name(const GtaFwdParams& p, int dhp, int esz) {
    if (p.flags & GTA_FLAG_FP32_PRODUCTS) return "gta_fwd2_kernel";
    if (gta_attn64_takes(p, dhp, layout_of(p, dhp), esz)) return gta_attn64_kernel_name(p, esz, layout_of(p, dhp));
    return gta_fwdc_takes(p, dhp, layout_of(p, dhp), esz) ? "gta_fwdc_kernel" : "gta_fwd2_kernel";
}

// Compile-time layouts exist for the shipped configs; others read the chunk table.
template <int DHP, int ESZ>
static int launch_flash(const GtaFwdParams& p, hipStream_t stream) {
    if (p.flags & GTA_FLAG_FP32_PRODUCTS) {              // split-bf16 operands, three MFMAs per product: the X3 instances of the 32-row kernel
        if constexpr (ESZ == 4 && DHP <= 64) {
            switch (layout_of(p, DHP)) {
                case GTA_LAYOUT_CL:  if (DHP == 64) return launch_fwd2<DHP, ESZ, (DHP == 64 ? GTA_LAYOUT_CL : GTA_LAYOUT_GENERIC), true>(p, stream); break;
                case GTA_LAYOUT_SO2: return launch_fwd2<DHP, ESZ, GTA_LAYOUT_SO2, true>(p, stream);
            }
            return launch_fwd2<DHP, ESZ, GTA_LAYOUT_GENERIC, true>(p, stream);
        } else {
            return GTA_E_UNSUPPORTED;
        }
    }
    if (gta_attn64_takes(p, DHP, layout_of(p, DHP), ESZ)) return gta_attn64_dispatch(p, ESZ, layout_of(p, DHP), stream);   // 64 rows per wave (gta_fwd64.hip)
    if (gta_fwdc_takes(p, DHP, layout_of(p, DHP), ESZ)) return gta_fwdc_dispatch(p, layout_of(p, DHP), stream);             // the dh = 64 bf16 instance (gta_fwd_cl.hip)
    switch (layout_of(p, DHP)) {
        case GTA_LAYOUT_MS:  if (DHP == 96) return launch_fwd2<DHP, ESZ, (DHP == 96 ? GTA_LAYOUT_MS : GTA_LAYOUT_GENERIC)>(p, stream); break;
        case GTA_LAYOUT_CL:  if (DHP == 64) return launch_fwd2<DHP, ESZ, (DHP == 64 ? GTA_LAYOUT_CL : GTA_LAYOUT_GENERIC)>(p, stream); break;
        case GTA_LAYOUT_SO2: return launch_fwd2<DHP, ESZ, GTA_LAYOUT_SO2>(p, stream);
    }
    return launch_fwd2<DHP, ESZ, GTA_LAYOUT_GENERIC>(p, stream);
}

// prep (unless the caller says K'/V' images are already in the workspace) + attention kernel.
int gta_fwd2_dispatch(GtaFwdParams& p, int dhp, int esz, bool run_prep, bool run_flash, hipStream_t stream) {
    p.n_qtiles = (p.Tq + 127) / 128;
    p.n_items = p.B * p.H * p.n_qtiles;
    // view records a 128-row query tile can touch (staged per item, two buffers)
    p.nrec = 128 / p.Pq + 2 < p.Nq ? 128 / p.Pq + 2 : p.Nq;
    int rc = GTA_OK;
    // the q-side rep tiles (rho_q / rho_q^-1 on the matrix cores) only where the 64-rows-per-wave kernel will use them
    // (the operand tiles are the MSN gta_so3 layout's: GTA_LAYOUT_MS)
    if (!(p.qtiles && esz == 2 && run_flash && layout_of(p, dhp) == GTA_LAYOUT_MS && gta_attn64_takes(p, dhp, GTA_LAYOUT_MS, esz))) p.qtiles = nullptr;
    if (run_prep) rc = gta_prep_dispatch(p, dhp, esz, stream);
    else if (p.qtiles) rc = gta_qtiles_dispatch(p, stream);
    if (rc != GTA_OK || !run_flash) return rc;
    switch (dhp) {
        case 32: return esz == 2 ? launch_flash<32, 2>(p, stream) : launch_flash<32, 4>(p, stream);
        case 64: return esz == 2 ? launch_flash<64, 2>(p, stream) : launch_flash<64, 4>(p, stream);
        case 96: return esz == 2 ? launch_flash<96, 2>(p, stream) : launch_flash<96, 4>(p, stream);
        case 128: return esz == 2 ? launch_flash<128, 2>(p, stream) : launch_flash<128, 4>(p, stream);
    }
    return GTA_E_UNSUPPORTED;
}
