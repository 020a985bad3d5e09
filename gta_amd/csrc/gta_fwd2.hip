// gta_fwd2.hip -- two-stage GTA attention forward for gfx950: the flash kernel and the dispatch of the plan
// (K/V rep pre-pass: gta_prep.hip; software-pipelined variant of the flash kernel: gta_fwd3.hip).
//
// Why two kernels (measured against the single fused kernel of gta_fwd.hip):
// rho_k acts on K and V per key TOKEN, but a flash kernel re-reads every key tile once per query
// tile, so fusing rho_k re-does the same fp32 VALU work Tq/BM times (10x at the MSN shape) in the
// loop that should be feeding the matrix cores.  Here it is done exactly once:
//
//   gta_kv_prep_kernel   64-key tile per workgroup: raw K,V rows -> LDS by LDS-DMA (coalesced),
//                        lane == key row applies rho_k per 8-channel chunk in fp32 registers,
//                        writes K' and V' as bf16 TILE IMAGES: the exact rotation-swizzled byte
//                        image the flash kernel wants in LDS (gta.py:160-219 for K and V).
//   gta_fwd2_kernel      128 query rows per workgroup (4 waves x 32), two workgroups per CU.  Prologue: rho on Q (gta.py
//                        :165,193,216), prescale, bf16, MFMA B fragments in VGPRs.  Main loop: K'/V'
//                        images stream HBM/L2 -> LDS through a 3-stage LDS-DMA ring (linear 1-KiB
//                        pieces, two tiles in flight, counted vmcnt, ONE raw s_barrier per tile, no
//                        VGPR staging, no VALU on the K/V path at all); S^T = K' Q'^T and
//                        O^T = V'^T P^T on v_mfma_f32_32x32x16_bf16; V'^T operands come from the
//                        row-major V' image with ds_read_b64_tr_b16; online softmax in registers.
//                        Epilogue: O/l -> LDS -> rho_q^-1 per chunk (gta.py:246-276) -> out, LSE.
// Measured dead ends (r01, profiles/r01/README.md): an explicit ping-pong of the 8-wave kernel (M segment
// = PV(j-1)+QK(j), V segment = softmax, wave groups one segment apart, 4..6-stage ring, all operands
// prefetched) ran the MFMA-only stretches at full rate (355 cycles / 12 MFMAs) but gained nothing
// end to end (280 us vs 243 us for two 4-wave workgroups per CU): LDS-read issue, LDS-DMA issue
// (~100+ cycles per 1-KiB piece for the issuing wave) and two barriers per tile ate the overlap.
#include <cstdlib>
#include "gta_flash_common.h"

int gta_prep_dispatch(const GtaFwdParams& p, int dhp, int esz, hipStream_t stream);                  // gta_prep.hip
int gta_fwd3_dispatch(const GtaFwdParams& p, int dhp, int esz, int layout, hipStream_t stream);      // gta_fwd3.hip

namespace {

// ================================================================================================
// 2. lean flash kernel
// ================================================================================================
// 4 waves per workgroup; every wave owns RB blocks of 32 query rows and re-uses each K'/V' fragment it
// reads from LDS for all of them.  Measured (profiles/r01): with RB = 1 and eight waves per CU the kernel
// is bound by LDS fragment reads (~35 cycles per ds_read_b128 under load, 8 x 24 KB per tile per CU --
// as long as all the MFMAs of the tile).  RB = 2 halves the LDS bytes per MFMA and runs one wave per SIMD
// with the whole 512-entry register file, so one block's softmax VALU can issue under the other's MFMAs.
template <int DHP, int RB>
struct Smem2 {
    static constexpr int NW = 4;
    static constexpr int BM = 32 * NW * RB;             // 128 or 256 query rows per workgroup
    static constexpr int NT = 64 * NW;
    static constexpr int CHP = DHP / 8;
    static constexpr int IMG = BN * DHP * 2;            // one K' or V' tile image
    static constexpr int STAGE = 2 * IMG;               // K' image then V' image
    static constexpr int RING_BYTES = NSTAGE * STAGE;
    static constexpr int QS_BYTES = BM * DHP * 2;       // Q' staging (aliases ring stages 1..)
    static constexpr int OROW = DHP + 4;
    static constexpr int OST_ROWS = (BM * OROW * 4 <= RING_BYTES) ? BM : BM / 2;   // rows per epilogue pass
    static constexpr int OST_BYTES = OST_ROWS * OROW * 4;
    static_assert(QS_BYTES <= RING_BYTES - STAGE, "Q staging must fit ring stages 1..");
    static_assert(OST_BYTES <= RING_BYTES, "O staging must fit the ring");
    static_assert(OST_ROWS == 128, "the epilogue item map assumes 128-row passes");
    // layout: [ring | q-side rep records (runtime size: Nq records)]
    static constexpr int OFF_RING = 0;
    static constexpr int OFF_QS = STAGE;
    static constexpr int OFF_QREC = RING_BYTES;
    // [ring | q-side rep records | |q'|^2 partial sums: NPAR x BM floats]
    __host__ __device__ static int off_qsq(int Nq) { return RING_BYTES + Nq * GTA_QREC * 4; }
    static int total(int Nq) { return off_qsq(Nq) + 2 * BM * 4; }
};

// issue the LDS-DMA of one K'/V' tile image pair (STAGE bytes, linear) into ring stage `st`
template <int DHP, int RB>
GTA_DEV void dma_stage(char* ring, int st, const char* img, int wave, int lane) {
    using S = Smem2<DHP, RB>;
    constexpr int PIECES = S::STAGE / 1024;             // 1 KiB per wave-instruction
    constexpr int PER_WAVE = PIECES / 4;
    static_assert(PIECES % 4 == 0, "stage must split evenly over the waves");
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int piece = wave * PER_WAVE + i;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(img + piece * 1024 + lane * 16),
            (__attribute__((address_space(3))) void*)(ring + st * S::STAGE + piece * 1024), 16, 0, 0);
    }
}

// Full path of the lazy softmax (tile 0, masked tail, violated bound): true row max of S' (= S - m_run), move
// m_run there, rescale l and O, re-base S' and the -m splat.  key of register r = kbase + (r&3) + 8(r>>2) (+32).
template <int DHP>
GTA_DEV void softmax_rebase(f32x16_t& s0, f32x16_t& s1, float& m_run, float& l_run, f32x16_t (&oacc)[DHP / 32],
                            f32x16_t& msplat, bool first, bool tail, int kbase, int Tk) {
    if (tail) {
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
        if constexpr (ABL & 8) { vlo[d] = u32x2_t{0, 0}; vhi[d] = u32x2_t{0, 0}; continue; }
        vlo[d] = lds_tr16_b64<OFF>(vbase + voff[d][0]);
        vhi[d] = lds_tr16_b64<OFF>(vbase + voff[d][1]);
    }
}
// SLAB-major PV for RB row blocks: one slab's fragments multiply into RB*DB independent accumulators
// (a chain on one accumulator would run at the dependent latency instead of the issue rate)
template <int DHP, int RB>
GTA_DEV void pv_mfma_slab(const u32x2_t (&vlo)[DHP / 32], const u32x2_t (&vhi)[DHP / 32], const bf16x8_t (&pf)[RB][2][2],
                          int kb, int t, f32x16_t (&oacc)[RB][DHP / 32]) {
#pragma unroll
    for (int d = 0; d < DHP / 32; ++d) {
        const u32x4_t av = {vlo[d].x, vlo[d].y, vhi[d].x, vhi[d].y};
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if constexpr (ABL & 128) { oacc[rb][d][0] += __uint_as_float(av.x); continue; }
            oacc[rb][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pf[rb][kb][t], oacc[rb][d], 0, 0, 0);
        }
    }
}

template <int DHP, int ESZ, int RB, int LAYOUT>
__global__ __launch_bounds__(256, (RB == 1 ? 2 : 1)) void gta_fwd2_kernel(const GtaFwdParams p) {
    using S = Smem2<DHP, RB>;
    // chunk descriptor: a compile-time constant for the shipped layouts (c is constant per unrolled item)
#define GTA_DESC(c) (LAYOUT == GTA_LAYOUT_GENERIC ? p.ctab[c] : gta_layout_desc(LAYOUT, c))
    constexpr int NW = 4, CHP = S::CHP, KS = DHP / 16, DB = DHP / 32, BM = S::BM, NT = S::NT;
    constexpr int DMA_PER_WAVE = S::STAGE / 1024 / NW;
    constexpr int RG = BM / 64;                      // 64-row groups of the Q tile (lane == row staging)
    constexpr int NPAR = NW / RG;                    // chunk parities: 2 (RB = 1) or 1 (RB = 2)
    constexpr int QITEMS = CHP / NPAR;               // (row group, chunk) items per wave in the prologue
    static_assert(NPAR * RG == NW && QITEMS * NPAR == CHP, "prologue item map");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    int w;
    {
        const int nwg = gridDim.x, L = blockIdx.x;
        const int xcd = L & 7, idx = L >> 3, q8 = nwg >> 3, r8 = nwg & 7;
        w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int bh = w / p.n_qtiles, qt = w - bh * p.n_qtiles;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * BM;
    const int n_tiles = (p.Tk + BN - 1) / BN;
    const int ch_real = p.dh >> 3;

    const char* qg = (const char*)p.q + ((long)b * p.q_sb + (long)h * p.q_sh) * ESZ;
    char* og = (char*)p.o + ((long)b * p.o_sb + (long)h * p.o_sh) * ESZ;
    const long q_rs = p.q_st * ESZ, o_rs = p.o_st * ESZ;
    const char* kvimg = (const char*)p.kp + ((long)b * p.H + h) * n_tiles * (long)S::STAGE;
    char* ring = smem + S::OFF_RING;
    float* qrec = reinterpret_cast<float*>(smem + S::OFF_QREC);

    if (GTA_DBG(512u)) return;                                          // ablation: bare launch
#ifdef GTA_ABLATE
#define GTA_STAMP(k) do { if (p.prof && tid == 0) p.prof[(long)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define GTA_STAMP2(k) do { if (p.prof && tid == 0) p.prof[((long)gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GTA_STAMP(k) do { } while (0)
#define GTA_STAMP2(k) do { } while (0)
#endif
    GTA_STAMP(0);
#ifdef GTA_ABLATE
    if (p.prof && tid == 0) p.prof[(long)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();   // 100 MHz reference
#endif
    if constexpr (RB != 1) dma_stage<DHP, RB>(ring, 0, kvimg, wave, lane);      // tile 0 on its way

    // views touched by this query tile: records are staged relative to n_first
    const int t_last = (q0 + BM - 1 < p.Tq ? q0 + BM - 1 : p.Tq - 1);
    const int n_first = q0 / p.Pq;
    const int n_cnt = t_last / p.Pq - n_first + 1;
    const float qscale = p.scale * LOG2E / (p.tau ? *p.tau : 1.0f);
    constexpr int RAWN = ESZ == 2 ? 1 : 2;
    bf16x8_t qf[RB][KS];
    float qn[RB];                  // |q'| of this lane's MFMA rows: with the pre-pass's per-tile max |k'| it bounds every score of a tile
    if constexpr (RB == 1) {
    // ---- prologue, fragment-direct: lane (l31, lh) of a wave IS the owner of MFMA B fragment (row 32*wave + l31,
    // chunks 2ks + lh), and rho_q acts inside a chunk -- so the lane loads exactly those 6 raw chunks, applies rho_q,
    // scales, rounds, and the result is qf[ks].  No Q staging tile, no LDS round trip, no barrier besides the one
    // behind the view records; ring stage 1 is free from the start, so tile 1 is requested together with tile 0.
    // The chunk kind of lanes 0-31 (even chunk) and 32-63 (odd chunk) is the same constant for every se3/se3,
    // so3/so3, so2/so2 pair of the shipped layouts (the select folds); a mixed pair runs both kinds under exec masks.
    // Request order = need order (loads return in order): the Q chunks gate the transform, the view records gate the
    // barrier in front of it, the tiles are not read before the loop -- so the 12 tile-DMA instructions go out last.
    const int my_r = wave * 32 + l31;
    int my_t = q0 + my_r;
    my_t = my_t < p.Tq ? my_t : p.Tq - 1;
    u32x4_t qraw[KS][RAWN];
    f32x2_t qcs[KS][4];
    const float tc = p.trans_coeff ? *p.trans_coeff : 1.0f;            // (asked for first: the record stores below need it)
    // chunk descriptor of this lane's chunk 2ks + lh
#define GTA_DL(ks) (lh ? GTA_DESC(2 * (ks) + 1) : GTA_DESC(2 * (ks)))
    // `full`: dh fills the padded head, so every chunk of every lane exists and no per-lane guard (exec save/restore per
    // load) is needed: the specialised path is straight-line code with immediate offsets off one row pointer.
    const bool full = ch_real == CHP;
    const char* qrow = qg + (long)my_t * q_rs + lh * 8 * ESZ;           // chunk 2ks + lh sits at + ks * 16 * ESZ
    const float* csrow = p.cs_q ? p.cs_q + ((long)b * p.Tq + my_t) * 2 * p.nso2 : nullptr;
    GTA_STAMP2(0);
    auto q_loads = [&](auto FULLC) {
        constexpr bool FULL = decltype(FULLC)::value;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if ((FULL || 2 * ks + lh < ch_real) && !GTA_DBG(32u)) {
#pragma unroll
                for (int k2 = 0; k2 < RAWN; ++k2) qraw[ks][k2] = *reinterpret_cast<const u32x4_t*>(qrow + ks * 16 * ESZ + 16 * k2);
                if (csrow) load_cs(GTA_DL(ks), csrow, qcs[ks]);
            }
        }
    };
    if (full) q_loads(std::true_type{}); else q_loads(std::false_type{});
    GTA_STAMP2(1);
    const bool want_rec = p.vrep_q && !GTA_DBG(128u);
    QrecItem rb0;
    if (want_rec) qrec_seg_load(rb0, p.vrep_q, b, p.Nq, n_first, n_cnt, wave, lane);
    GTA_STAMP2(2);
    dma_stage<DHP, RB>(ring, 0, kvimg, wave, lane);
    if (n_tiles > 1) dma_stage<DHP, RB>(ring, 1, kvimg + (long)S::STAGE, wave, lane);
    GTA_STAMP(7);                                                        // (all prologue loads issued)
    if (want_rec) {
        qrec_seg_store(rb0, qrec, tc);
        for (int idx = lane + 64; idx < qrec_seg_count(wave, n_cnt); idx += 64) {       // (more than one view per tile only)
            QrecItem it;
            qrec_seg_load(it, p.vrep_q, b, p.Nq, n_first, n_cnt, wave, idx);
            qrec_seg_store(it, qrec, tc);
        }
    }
    GTA_STAMP2(3);
    __syncthreads();
    GTA_STAMP(1);
    float qsq = 0.f;                                   // this lane's share of |q'_row|^2 (bf16-rounded values)
    const float* rec_q = qrec + (view_of(my_t, p.Pq, p.invPq) - n_first) * GTA_QREC;
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
#pragma unroll
                for (int i = 0; i < 8; ++i) x[0][i] *= qscale;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[0][i] = 0.f;
            }
            const u32x4_t qw = pack8(x[0]);
            qf[0][ks] = __builtin_bit_cast(bf16x8_t, qw);
            float qr[8];
            unpack8(qw, qr);
#pragma unroll
            for (int i = 0; i < 8; ++i) qsq += qr[i] * qr[i];
        }
    };
    if (full) q_xform(std::true_type{}); else q_xform(std::false_type{});
    qsq += __shfl_xor(qsq, 32);                                          // the row's other chunk parity
    qn[0] = sqrtf(qsq) * 1.0001f;
    GTA_STAMP(2);
    } else {
    // ---- prologue: every global load is issued up front (one latency exposure, not one per item).
    // item map: wave -> (row group rg = wave % RG, chunk parity par = wave / RG); item it -> chunk
    // NPAR*it + par: a constant in each of the NPAR straight-line code paths.
    u32x4_t qraw[QITEMS][RAWN];
    f32x2_t qcs[QITEMS][4];
    const int rg = wave % RG, par = wave / RG;
    const int my_r = lane + 64 * rg;
    int my_t = q0 + my_r;
    my_t = my_t < p.Tq ? my_t : p.Tq - 1;
    auto load_items = [&](auto PARC) {
        constexpr int PAR = decltype(PARC)::value;
#pragma unroll
        for (int it = 0; it < QITEMS; ++it) {
            const int c = NPAR * it + PAR;
            if (c < ch_real && !GTA_DBG(32u)) {
                const char* rp = qg + (long)my_t * q_rs + c * 8 * ESZ;
#pragma unroll
                for (int k2 = 0; k2 < RAWN; ++k2) qraw[it][k2] = *reinterpret_cast<const u32x4_t*>(rp + 16 * k2);
                if (p.cs_q) load_cs(GTA_DESC(c), p.cs_q + ((long)b * p.Tq + my_t) * 2 * p.nso2, qcs[it]);
            }
        }
    };
    if (NPAR == 2 && par) load_items(std::integral_constant<int, 1>{}); else load_items(std::integral_constant<int, 0>{});
    GTA_STAMP(7);                                                        // (all prologue loads issued)
    if (p.vrep_q && !GTA_DBG(128u)) stage_qrec(qrec, p.vrep_q, b, p.Nq, n_first, n_cnt, p.trans_coeff ? *p.trans_coeff : 1.0f, tid, NT);
    __syncthreads();

    GTA_STAMP(1);
    // ---- Q: rho, prescale, bf16 -> LDS ----
    float qsq = 0.f;                                   // this thread's share of |q'_row|^2 (bf16-rounded values)
    float* qsq_l = reinterpret_cast<float*>(smem + S::off_qsq(p.vrep_q ? p.Nq : 0));
    auto xform_items = [&](auto PARC) {
        constexpr int PAR = decltype(PARC)::value;
        char* qs = smem + S::OFF_QS;
        const int r = my_r;
#pragma unroll
        for (int it = 0; it < QITEMS; ++it) {
            const int c = NPAR * it + PAR;
            float x[1][8];
            if (c < ch_real) {
                const uint32_t desc = GTA_DESC(c);
                if (ESZ == 2) {
                    unpack8(qraw[it][0], x[0]);
                } else {
#pragma unroll
                    for (int k2 = 0; k2 < RAWN; ++k2) {
                        x[0][4 * k2 + 0] = __uint_as_float(qraw[it][k2].x); x[0][4 * k2 + 1] = __uint_as_float(qraw[it][k2].y);
                        x[0][4 * k2 + 2] = __uint_as_float(qraw[it][k2].z); x[0][4 * k2 + 3] = __uint_as_float(qraw[it][k2].w);
                    }
                }
                if (desc) {
                    const int n = view_of(my_t, p.Pq, p.invPq) - n_first;
                    const float* rec = qrec + n * GTA_QREC;
                    chunk_apply<false, 1>(desc, rec + GTA_QREC_A, rec + GTA_QREC_D1, rec + GTA_QREC_D2, qcs[it], x);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) x[0][i] *= qscale;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[0][i] = 0.f;
            }
            const u32x4_t qw = pack8(x[0]);
            *reinterpret_cast<u32x4_t*>(qs + (r * CHP + swz<CHP>(r, c)) * 16) = qw;
            float qr[8];
            unpack8(qw, qr);
#pragma unroll
            for (int i = 0; i < 8; ++i) qsq += qr[i] * qr[i];
        }
    };
    if (NPAR == 2 && par) xform_items(std::integral_constant<int, 1>{}); else xform_items(std::integral_constant<int, 0>{});
    qsq_l[par * BM + my_r] = qsq;
    __syncthreads();      // (also drains tile 0's DMA: harmless)
    // |q'| of this lane's MFMA rows: with the pre-pass's per-tile max |k'| it bounds every score of a tile
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int r = wave * (32 * RB) + 32 * rb + l31;
        qn[rb] = sqrtf(qsq_l[r] + (NPAR == 2 ? qsq_l[BM + r] : 0.f)) * 1.0001f;
    }
    {
        const char* qs = smem + S::OFF_QS;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = wave * (32 * RB) + 32 * rb + l31;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qf[rb][ks] = *reinterpret_cast<const bf16x8_t*>(qs + (r * CHP + swz<CHP>(r, 2 * ks + lh)) * 16);
        }
    }
    __syncthreads();      // Q staging (ring stages 1..2) is free again
    if (n_tiles > 1) dma_stage<DHP, RB>(ring, 1, kvimg + (long)S::STAGE, wave, lane);
    GTA_STAMP(2);

    }

    f32x16_t oacc[RB][DB];
    float m_run[RB], l_run[RB];
    f32x16_t msplat[RB];                  // -m_run in every element: C operand of each tile's first MFMA (S' = S - m)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        m_run[rb] = 0.f; l_run[rb] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) msplat[rb][i] = 0.f;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[rb][d][i] = 0.f;
    }

    // lane-constant LDS offsets.  The rotation swizzle has period 16 rows, so a fragment of rows
    // r + 16m sits at the same in-row position: per-slab offsets are compile-time immediates.
    int koff[KS];            // K' fragment: row l31, unit 2ks+lh  (rows 32.. : + 32*CHP*16)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = (l31 * CHP + swz<CHP>(l31, 2 * ks + lh)) * 16;
    const int g16 = lane >> 4, p16 = lane & 15;
    int voff[DB][2];         // V' transpose-read: key row 4lh + (p16>>2) (+8), channel unit of block d
#pragma unroll
    for (int d = 0; d < DB; ++d) {
        const int u = 4 * d + 2 * (g16 & 1) + ((p16 & 3) >> 1);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int r = 4 * lh + (p16 >> 2) + 8 * hf;
            voff[d][hf] = (r * CHP + swz<CHP>(r, u)) * 16 + (p16 & 1) * 8;
        }
    }
    const bool has_tail = (p.Tk & (BN - 1)) != 0;

    if constexpr (RB == 1 && PIPE1 && DHP == 96) {      // (dh = 64: 230 VGPRs would cost the third workgroup per CU)
    // ---- skewed tile loop (RB == 1): the QK^T MFMAs of tile j+1 issue beside the softmax VALU of tile j ----
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
#ifndef GTA_KLA
#define GTA_KLA 2
#endif
    constexpr int KLA = GTA_KLA;                      // K' fragment reads run this many k steps ahead of their MFMAs
    auto e_first = [](int g) constexpr { return g * NEV / GA; };
    auto s_fence1 = [&](f32x16_t (&s)[2]) { asm volatile("" : "+v"(s[0])); asm volatile("" : "+v"(s[1])); };
    auto step = [&](f32x16_t (&sc)[2], f32x16_t (&sn)[2], int j, auto LASTC) {
        constexpr bool LAST = decltype(LASTC)::value;
        // the tile's key-norm bound is asked for BEFORE the waits below, so it is there when they are over (measured:
        // awaiting it together with the first K' fragment reads cost 4 % of the loop's cycles)
        uint32_t kn_bits;
        {
            const float* kn_ptr = p.kn + __builtin_amdgcn_readfirstlane((b * p.H + h) * n_tiles + j);
            asm volatile("s_load_dword %0, %1, 0x0" : "=s"(kn_bits) : "s"(kn_ptr) : "memory");
        }
        // tile j+1 has landed, everyone is past B(j-1): its stage takes tile j+2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!(ABL & 256)) __builtin_amdgcn_s_barrier();
        if (!(ABL & 32) && j + 2 < n_tiles) dma_stage<DHP, RB>(ring, (j + 2) % NSTAGE, kvimg + (long)(j + 2) * S::STAGE, wave, lane);
        const char* kf = ring + ((j + 1) % NSTAGE) * S::STAGE;          // K'(j+1)
        const uint32_t vbase = lds_addr(ring + (j % NSTAGE) * S::STAGE + S::IMG);   // V'(j)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(kn_bits));           // (nothing else is outstanding on that counter here)
        bf16x8_t kfr[KS][2];
        auto k_load = [&](auto KC) {
            constexpr int ks = decltype(KC)::value;
            if constexpr (!LAST && ks < KS && !(ABL & 16)) {
                kfr[ks][0] = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks]);
                kfr[ks][1] = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks] + 32 * CHP * 16);
            }
        };
        static_for<KLA>([&](auto KC) { k_load(KC); });
        // decision for tile j (sc = S'(j) relative to m_run): lazy-softmax full path only when needed
        {
            const float kn_j = __uint_as_float(kn_bits);
            const bool tail = has_tail && j == n_tiles - 1;
            const bool need = (j == 0) || tail || (qn[0] * kn_j - m_run[0] > BOUND_THR);
            if (__builtin_amdgcn_ballot_w64(need) != 0) {
                asm volatile("" ::: "memory");
                l_run[0] += rs0 + rs1; rs0 = 0.f; rs1 = 0.f;
                softmax_rebase<DHP>(sc[0], sc[1], m_run[0], l_run[0], oacc[0], msplat[0], j == 0, tail, j * BN + 4 * lh, p.Tk);
            }
            s_fence1(sc);
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<GA>([&](auto GC) {
            constexpr int g = decltype(GC)::value, ks = g >> 1, hh = g & 1;
            if constexpr (!LAST) {
                if constexpr (ABL & 64) { if (ks == 0) sn[hh] = msplat[0]; }
                else if constexpr (ks == 0) sn[hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0][hh], qf[0][0], msplat[0], 0, 0, 0);
                else sn[hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks][hh], qf[0][ks], sn[hh], 0, 0, 0);
                if constexpr (hh == 0) k_load(std::integral_constant<int, ks + KLA>{});
            }
            // this gap's exps, then the sums and packs of the values finished in earlier gaps
            constexpr int e0 = e_first(g), e1 = e_first(g + 1), ep = g > 0 ? e_first(g - 1) : 0;
            if constexpr (!(ABL & 1))
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
        bf16x8_t pfb[1][2][2];
        pfb[0][0][0] = __builtin_bit_cast(bf16x8_t, pfr[0][0]); pfb[0][0][1] = __builtin_bit_cast(bf16x8_t, pfr[0][1]);
        pfb[0][1][0] = __builtin_bit_cast(bf16x8_t, pfr[1][0]); pfb[0][1][1] = __builtin_bit_cast(bf16x8_t, pfr[1][1]);
        pv_reads_slab<DHP, 0>(vbase, voff, v0l, v0h);
        pv_reads_slab<DHP, 1>(vbase, voff, v1l, v1h);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DB) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP, RB>(v0l, v0h, pfb, 0, 0, oacc);
        pv_reads_slab<DHP, 2>(vbase, voff, v2l, v2h);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DB) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP, RB>(v1l, v1h, pfb, 0, 1, oacc);
        pv_reads_slab<DHP, 3>(vbase, voff, v3l, v3h);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DB) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP, RB>(v2l, v2h, pfb, 1, 0, oacc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP, RB>(v3l, v3h, pfb, 1, 1, oacc);
        if constexpr (!LAST) s_fence1(sn);
    };
    // tile 0: S'(0) by itself
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        const char* kf = ring;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t k0 = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks]);
            const bf16x8_t k1 = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks] + 32 * CHP * 16);
            sA[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[0][ks], ks == 0 ? msplat[0] : sA[0], 0, 0, 0);
            sA[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[0][ks], ks == 0 ? msplat[0] : sA[1], 0, 0, 0);
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
    l_run[0] += rs0 + rs1;
    } else
    for (int j = 0; j < n_tiles; ++j) {
        // tile j has landed (only tile j+1's pieces may still be in flight), everyone is past tile j-1
        if (j + 1 < n_tiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
        else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (j + 2 < n_tiles) dma_stage<DHP, RB>(ring, (j + 2) % NSTAGE, kvimg + (long)(j + 2) * S::STAGE, wave, lane);

        if constexpr (ABL & 512) {        // sensitivity experiment: 24 extra scalar issue slots per tile
            asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
                         "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
        }
        if constexpr (ABL & 1024) {       // sensitivity experiment: 24 extra VALU issue slots per tile
            float dummy = __uint_as_float(lane);
            asm volatile("v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n"
                         "v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n"
                         "v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n"
                         "v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0" : "+v"(dummy));
        }
        const char* kf = ring + (j % NSTAGE) * S::STAGE;
        const char* vf = kf + S::IMG;
        uint32_t kn_bits;                 // max_k |k'_k| of tile j (scalar load by hand: see section 3)
        {
            const float* kn_ptr = p.kn + __builtin_amdgcn_readfirstlane((b * p.H + h) * n_tiles + j);
            asm volatile("s_load_dword %0, %1, 0x0" : "=s"(kn_bits) : "s"(kn_ptr) : "memory");
        }

        // ---- S^T = K' Q'^T for the RB row blocks: each K' fragment is read once and used RB times ----
        f32x16_t s[RB][2];
        {
            bf16x8_t ka[KS], kb2[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                ka[ks] = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks]);
                kb2[ks] = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks] + 32 * CHP * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                s[rb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[0], qf[rb][0], msplat[rb], 0, 0, 0);
                s[rb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb2[0], qf[rb][0], msplat[rb], 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < KS; ++ks) {
                    s[rb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks], qf[rb][ks], s[rb][0], 0, 0, 0);
                    s[rb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb2[ks], qf[rb][ks], s[rb][1], 0, 0, 0);
                }
            }
        }
        // the tile's key-norm bound (scalar load from the top of the iteration; the K' reads are consumed, so this
        // wait is free -- it must sit BEFORE the V' reads below or it would drain them too)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(kn_bits));
        // V' slab 0 transpose-reads fly under the softmax
        const uint32_t vbase = lds_addr(vf);
        u32x2_t v0l[DB], v0h[DB], v1l[DB], v1h[DB], v2l[DB], v2h[DB], v3l[DB], v3h[DB];
        pv_reads_slab<DHP, 0>(vbase, voff, v0l, v0h);

        // ---- online softmax per row block (the scheduler may run block 0's VALU under block 1's MFMAs) ----
        // Lazy online softmax: S' already has -m_run folded in.  |q'| max_k|k'| bounds the tile's scores, so
        // while bound - m_run stays below BOUND_THR no exponent can overflow and neither the row max nor the
        // O rescale is needed; tile 0, the masked tail tile and a violated bound take the full path.
        bf16x8_t pf[RB][2][2];
        {
            const float kn_j = __uint_as_float(kn_bits);
            const bool tail = has_tail && j == n_tiles - 1;
            bool need = (j == 0) || tail;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) need = need || (qn[rb] * kn_j - m_run[rb] > BOUND_THR);
            if (__builtin_amdgcn_ballot_w64(need) != 0) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    softmax_rebase<DHP>(s[rb][0], s[rb][1], m_run[rb], l_run[rb], oacc[rb], msplat[rb], j == 0, tail,
                                        j * BN + 4 * lh, p.Tk);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) softmax_exp_pack(s[rb][0], s[rb][1], l_run[rb], pf[rb]);
        }

        // ---- O^T += V'^T P^T, slab-major; reads stay one slab ahead (LDS returns in order) ----
        pv_reads_slab<DHP, 1>(vbase, voff, v1l, v1h);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DB) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP, RB>(v0l, v0h, pf, 0, 0, oacc);
        pv_reads_slab<DHP, 2>(vbase, voff, v2l, v2h);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DB) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP, RB>(v1l, v1h, pf, 0, 1, oacc);
        pv_reads_slab<DHP, 3>(vbase, voff, v3l, v3h);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DB) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP, RB>(v2l, v2h, pf, 1, 0, oacc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma_slab<DHP, RB>(v3l, v3h, pf, 1, 1, oacc);
    }

    GTA_STAMP(3);
    if (GTA_DBG(256u)) {                                                // ablation: no epilogue at all
        if (oacc[0][0][0] == 123.f) p.lse[0] = l_run[0];
        return;
    }
    // ---- epilogue through the O staging tile, 128 rows per pass ----
    float inv_l[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const float l_tot = l_run[rb] + __shfl_xor(l_run[rb], 32);
        inv_l[rb] = 1.0f / l_tot;
        if (p.lse && lh == 0) {
            const int t = q0 + wave * (32 * RB) + 32 * rb + l31;
            if (t < p.Tq) p.lse[((long)b * p.H + h) * p.Tq + t] = (m_run[rb] + __log2f(l_tot)) * LN2;
        }
    }
    float* ost = reinterpret_cast<float*>(smem + S::OFF_RING);
    const bool xo = (p.flags & GTA_FLAG_V_TRANSFORM) != 0;
    constexpr int NPASS = BM / S::OST_ROWS;                          // RB
    constexpr int WPP = NW / NPASS;                                  // waves whose rows go in one pass
    constexpr int EITEMS = CHP / 2;                                  // epilogue map: 2 row groups x 2 parities
    constexpr bool SAMEMAP = false;   // (the prologue's (cos,sin) are NOT kept across the loop: 48 VGPRs the skewed loop needs)
    if constexpr (RB == 1) {
        // ---- epilogue in registers: the accumulators hold, per lane (row l31), the 4-channel HALF lh of every chunk;
        // one v_permlane32_swap per value hands lanes 0-31 the whole even chunk of a pair and lanes 32-63 the whole odd
        // one (the same ownership as the prologue), rho_q^-1 acts inside the chunk, and the chunk is stored.  No O
        // staging tile, no barrier.
        const int tE = q0 + wave * 32 + l31;
        const bool rowok = tE < p.Tq;
        const int tC = rowok ? tE : p.Tq - 1;
        constexpr int NP = CHP / 2;
        f32x2_t ocs[NP][4];
        const bool fast = ch_real == CHP && q0 + BM <= p.Tq;          // no ragged channel, no ragged row: no per-lane guards
        const float* csrow_o = (xo && p.cs_q) ? p.cs_q + ((long)b * p.Tq + tC) * 2 * p.nso2 : nullptr;
        const float* rec_o = qrec + (view_of(tC, p.Pq, p.invPq) - n_first) * GTA_QREC;
        char* orow = og + (long)tC * o_rs + lh * 8 * ESZ;                // chunk 2kp + lh goes to + kp * 16 * ESZ
        auto o_items = [&](auto FASTC) {
            constexpr bool FAST = decltype(FASTC)::value;
            if (csrow_o) {
#pragma unroll
                for (int kp = 0; kp < NP; ++kp)
                    if (FAST || 2 * kp + lh < ch_real) load_cs(GTA_DL(kp), csrow_o, ocs[kp]);
            }
#pragma unroll
            for (int kp = 0; kp < NP; ++kp) {
                const int d = (2 * kp) >> 2, ge = (2 * kp) & 3;
                float x[1][8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t ua = __float_as_uint(oacc[0][d][4 * ge + i] * inv_l[0]);         // half lh of the even chunk
                    const uint32_t ub = __float_as_uint(oacc[0][d][4 * (ge + 1) + i] * inv_l[0]);   // half lh of the odd chunk
                    const auto sw = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);         // ua[32..63] <-> ub[0..31]
                    x[0][i] = __uint_as_float(sw[0]);
                    x[0][4 + i] = __uint_as_float(sw[1]);
                }
                if (FAST || (rowok && 2 * kp + lh < ch_real)) {
                    if (xo && GTA_DL(kp)) chunk_apply<true, 1>(GTA_DL(kp), rec_o + GTA_QREC_O, rec_o + GTA_QREC_D1T, rec_o + GTA_QREC_D2T, ocs[kp], x);
                    if (!GTA_DBG(64u) || x[0][0] == 123.f) gstore_chunk2<ESZ>(orow + kp * 16 * ESZ, 0, x[0]);
                }
            }
        };
        if (fast) o_items(std::true_type{}); else o_items(std::false_type{});
    } else {
    const int rgE = wave & 1, parE = wave >> 1;
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
        const int rE = lane + 64 * rgE;
        const int tE = q0 + pass * S::OST_ROWS + rE;
        f32x2_t ocs[EITEMS][4];
        if (!SAMEMAP && xo && p.cs_q && tE < p.Tq) {                  // prefetched before the barriers
            auto load_ocs = [&](auto PARC) {
                constexpr int PAR = decltype(PARC)::value;
#pragma unroll
                for (int it = 0; it < EITEMS; ++it) {
                    const int c = 2 * it + PAR;
                    if (c < ch_real) load_cs(GTA_DESC(c), p.cs_q + ((long)b * p.Tq + tE) * 2 * p.nso2, ocs[it]);
                }
            };
            if (parE) load_ocs(std::integral_constant<int, 1>{}); else load_ocs(std::integral_constant<int, 0>{});
        }
        __syncthreads();
        if (wave / WPP == pass) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = (wave % WPP) * (32 * RB) + 32 * rb + l31;
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4_t v = {oacc[rb][d][4 * g] * inv_l[rb], oacc[rb][d][4 * g + 1] * inv_l[rb],
                                           oacc[rb][d][4 * g + 2] * inv_l[rb], oacc[rb][d][4 * g + 3] * inv_l[rb]};
                        *reinterpret_cast<f32x4_t*>(ost + r * S::OROW + 32 * d + 8 * g + 4 * lh) = v;
                    }
            }
        }
        __syncthreads();
        // rho_q^-1 on one (row, chunk) item and the store
        auto out_item = [&](const uint32_t desc, const int c, const f32x2_t* cs) {
            float x[1][8];
            const f32x4_t a = *reinterpret_cast<const f32x4_t*>(ost + rE * S::OROW + 8 * c);
            const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(ost + rE * S::OROW + 8 * c + 4);
            x[0][0] = a.x; x[0][1] = a.y; x[0][2] = a.z; x[0][3] = a.w;
            x[0][4] = bb.x; x[0][5] = bb.y; x[0][6] = bb.z; x[0][7] = bb.w;
            if (xo && desc) {
                const int n = view_of(tE, p.Pq, p.invPq) - n_first;
                const float* rec = qrec + n * GTA_QREC;
                chunk_apply<true, 1>(desc, rec + GTA_QREC_O, rec + GTA_QREC_D1T, rec + GTA_QREC_D2T, cs, x);
            }
            if (!GTA_DBG(64u) || x[0][0] == 123.f) gstore_chunk2<ESZ>(og + (long)tE * o_rs, c, x[0]);
        };
        auto out_items = [&](auto PARC) {
            constexpr int PAR = decltype(PARC)::value;
#pragma unroll
            for (int it = 0; it < EITEMS; ++it) {
                const int c = 2 * it + PAR;
                if (c < ch_real && tE < p.Tq) {
                    out_item(GTA_DESC(c), c, ocs[it]);
                }
            }
        };
        if (parE) out_items(std::integral_constant<int, 1>{}); else out_items(std::integral_constant<int, 0>{});
    }
    }
    GTA_STAMP(4);
#ifdef GTA_ABLATE
    if (p.prof && tid == 0) p.prof[(long)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
#endif
#undef GTA_STAMP
#undef GTA_STAMP2
#undef GTA_DL
}

template <int DHP, int ESZ, int RB, int LAYOUT>
int launch_fwd2(const GtaFwdParams& p, hipStream_t stream) {
    using S = Smem2<DHP, RB>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gta_fwd2_kernel<DHP, ESZ, RB, LAYOUT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, S::total(GTA_MAX_VIEWS)) != hipSuccess)
            return GTA_E_LAUNCH;
        attr_set = true;
    }
    const long n_wg = (long)p.B * p.H * p.n_qtiles;
    int lds = S::total(p.vrep_q ? p.Nq : 0);
#ifdef GTA_ABLATE
    if (const char* e = getenv("GTA_LDS_PAD")) {        // occupancy experiment: inflate LDS so fewer workgroups share a CU
        lds += atoi(e);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gta_fwd2_kernel<DHP, ESZ, RB, LAYOUT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
#endif
    hipLaunchKernelGGL((gta_fwd2_kernel<DHP, ESZ, RB, LAYOUT>), dim3((unsigned)n_wg), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

}  // namespace

// workspace = [K'/V' tile images | per-tile key norms (float, 256-byte aligned start)]
long gta_fwd2_image_bytes(int B, int H, int Tk, int dhp) {
    const long n_tiles = (Tk + BN - 1) / BN;
    return (long)B * H * n_tiles * 2L * BN * dhp * 2;
}
long gta_fwd2_workspace_bytes(int B, int H, int Tk, int dhp) {
    const long n_tiles = (Tk + BN - 1) / BN;
    const long img = (gta_fwd2_image_bytes(B, H, Tk, dhp) + 255) & ~255L;
    return img + (((long)B * H * n_tiles * 4 + 255) & ~255L);
}
int gta_fwd2_lds_bytes(int dhp, int nq) {
    switch (dhp) {
        case 32: return Smem2<32, 1>::total(nq);
        case 64: return Smem2<64, 1>::total(nq);
        case 96: return Smem2<96, 1>::total(nq);
        case 128: return Smem2<128, 1>::total(nq);
    }
    return -1;
}

// which compile-time layout (if any) the run-time chunk table is
static int layout_of(const GtaFwdParams& p, int dhp) {
    const int ch = p.dh / 8;
    if (p.dh != dhp) return GTA_LAYOUT_GENERIC;
    for (int L : {GTA_LAYOUT_MS, GTA_LAYOUT_CL, GTA_LAYOUT_SO2}) {
        if ((L == GTA_LAYOUT_MS && dhp != 96) || (L == GTA_LAYOUT_CL && dhp != 64)) continue;
        bool same = true;
        for (int c = 0; c < ch; ++c) same = same && p.ctab[c] == gta_layout_desc(L, c);
        if (same) return L;
    }
    return GTA_LAYOUT_GENERIC;
}

// rb = 32-row query blocks per wave: 1 (128-row workgroups, two per CU) or 2 (256-row workgroups, one
// wave per SIMD).  Compile-time layouts exist for the shipped configs; others read the chunk table.
template <int DHP, int ESZ, int RB>
static int launch_flash_rb(const GtaFwdParams& p, hipStream_t stream) {
    switch (layout_of(p, DHP)) {
        case GTA_LAYOUT_MS:  if (DHP == 96) return launch_fwd2<DHP, ESZ, RB, (DHP == 96 ? GTA_LAYOUT_MS : GTA_LAYOUT_GENERIC)>(p, stream); break;
        case GTA_LAYOUT_CL:  if (DHP == 64) return launch_fwd2<DHP, ESZ, RB, (DHP == 64 ? GTA_LAYOUT_CL : GTA_LAYOUT_GENERIC)>(p, stream); break;
        case GTA_LAYOUT_SO2: return launch_fwd2<DHP, ESZ, RB, GTA_LAYOUT_SO2>(p, stream);
    }
    return launch_fwd2<DHP, ESZ, RB, GTA_LAYOUT_GENERIC>(p, stream);
}
template <int DHP, int ESZ>
static int launch_flash(const GtaFwdParams& p, int rb, hipStream_t stream) {
    if (rb == 2) {
        // the software-pipelined 256-row kernel exists for dh <= 96; dh = 128 / 32 keep the plain loop
        if constexpr (DHP == 64 || DHP == 96) return gta_fwd3_dispatch(p, DHP, ESZ, layout_of(p, DHP), stream);
        else return launch_flash_rb<DHP, ESZ, 2>(p, stream);
    }
    return launch_flash_rb<DHP, ESZ, 1>(p, stream);
}

// prep (unless the caller says K'/V' images are already in the workspace) + flash.
// nw: 4 = 128-row workgroups (default, two per CU); 8 = GTA_FLAG_WG8, 256-row workgroups (64 rows per wave)
int gta_fwd2_dispatch(GtaFwdParams& p, int dhp, int esz, bool run_prep, bool run_flash, int nw, hipStream_t stream) {
    const int rb = nw == 8 ? 2 : 1;
    p.n_qtiles = (p.Tq + 128 * rb - 1) / (128 * rb);
    int rc = GTA_OK;
    if (run_prep) rc = gta_prep_dispatch(p, dhp, esz, stream);
    if (rc != GTA_OK || !run_flash) return rc;
    switch (dhp) {
        case 32: return esz == 2 ? launch_flash<32, 2>(p, rb, stream) : launch_flash<32, 4>(p, rb, stream);
        case 64: return esz == 2 ? launch_flash<64, 2>(p, rb, stream) : launch_flash<64, 4>(p, rb, stream);
        case 96: return esz == 2 ? launch_flash<96, 2>(p, rb, stream) : launch_flash<96, 4>(p, rb, stream);
        case 128: return esz == 2 ? launch_flash<128, 2>(p, rb, stream) : launch_flash<128, 4>(p, rb, stream);
    }
    return GTA_E_UNSUPPORTED;
}
