// gta_fwd3.hip -- software-pipelined flash kernel of the two-stage forward (gfx950), kv_mode="prepass8" /
// GTA_FLAG_WG8.  Experimental: parity-green, measured 240 us vs 212 us for gta_fwd2_kernel at the MSN shape
// (profiles/r01/README.md has the anatomy); kept because it documents what one wave per SIMD with an asm-owned
// accumulator file can and cannot do under hipcc.
#include "gta_flash_common.h"

namespace {

// ================================================================================================
// 3. software-pipelined flash kernel
// ================================================================================================
// Same tiles and images as section 2, but the tile loop is skewed so that every MFMA burst has VALU /
// LDS work of a DIFFERENT tile to issue in its shadow (measured on gfx950, tests/probes/probe_issue.hip:
// one 32x32x16 MFMA = 32 cycles of matrix pipe during which the wave can issue ~28 cycles of other
// instructions for free: v_mul 5, v_max3 / v_cvt_pk 5, v_exp 9 cycles each; the un-skewed loop above runs
// MFMA and softmax back to back and a second wave on the SIMD does not hide it):
//
//   step i:   R3  S'(i+1) = K'(i+1) Q'^T - m      ||  exp / sum / pack the late half of P(i), V'(i) tr-reads
//             R1  O += V'(i) P(i)   (slabs 0,1)   ||  row max of S'(i+1), V'(i) tr-reads of slabs 2,3
//             --  deferred-max decision for tile i+1 (wave-uniform, rare slow path)
//             R2  O += V'(i) P(i)   (slabs 2,3)   ||  exp / sum of the early half of P(i+1), K'(i+2) fragment reads
//
// Deferred max (THR = 8 in log2 units): the running max m only moves when a row's tile max exceeds it by
// more than THR, so P <= 2^8 and the O rescale is off the common path.  -m rides in as the C operand of
// each row block's first MFMA (a 16-register splat), so S' needs no subtract.  When the slow path fires
// for tile i+1, O still has P(i) V'(i) MFMAs in flight at the OLD scale: its rescale is applied after R2.
//
// LDS: K' ring of 3 images (K'(i+2) is read while K'(i+3) lands), V' ring of 2: [K0 | K1 | K2 | V1 | V0].
// One barrier per tile; the DMA of V'(i+1) and K'(i+3) is issued right after it.
#ifndef GTA_PK_SUM
#define GTA_PK_SUM 0      // (1 = packed row sums: fewer issue slots, but wrong rows on some instantiations -- not understood yet)
#endif
constexpr bool PK_SUM = GTA_PK_SUM != 0;
constexpr bool SKIP_MAX = (ABL & 4) != 0;

// single VALU instructions kept single: a plain -O3 build SLP-packs adjacent f32 adds into v_pk_add_f32 (slower
// beside MFMAs) and puts a canonicalising v_max in front of every fmaxf on an MFMA result
GTA_DEV void add_f32(float& acc, float x) { asm("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x)); }
// m = max(m, x0..x7): four dependent v_max3 in one statement (hipcc pads every asm statement whose result
// the next one reads with an s_nop; one pad per eight values instead of one per two)
GTA_DEV void max8_f32(float& m, float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7) {
    asm("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4\n\tv_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %0, %0, %7, %8"
        : "+v"(m) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
}
// ---- accumulator-file primitives (literal AGPR numbers; see the register map in the kernel) ----
template <int A0>
GTA_DEV void acc_zero() { asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"i"(A0)); }
template <int A0>
GTA_DEV void acc_scale(float f) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a[%c2], %0" : "=&v"(t) : "v"(f), "i"(A0));
}
template <int A0>
GTA_DEV f32x4_t acc_read4() {
    f32x4_t v;
    asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                 : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w) : "i"(A0), "i"(A0 + 1), "i"(A0 + 2), "i"(A0 + 3));
    return v;
}
template <int A0, int OFF>
GTA_DEV void lds_b128_to_acc(uint32_t addr) {
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "i"(A0), "i"(A0 + 3), "i"(OFF) : "memory");
}
template <int A0, int OFF>
GTA_DEV void lds_tr_to_acc(uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "i"(A0), "i"(A0 + 1), "i"(OFF) : "memory");
}
// S'(first k step) = K' Q'^T + C ;  S' += K' Q'^T ;  O^T += V'^T P^T
template <int K0, int Q0>
GTA_DEV void mfma_qk_first(f32x16_t& d, const f32x16_t& c) {
    // (s_nop: hipcc may materialise or copy the C operand right in front of the statement; a VALU write
    //  needs two wait states before an MFMA reads it, and nothing inside an asm string is padded)
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, a[%c2:%c3], a[%c4:%c5], %1" : "=&v"(d) : "v"(c), "i"(K0), "i"(K0 + 3), "i"(Q0), "i"(Q0 + 3));
}
template <int K0, int Q0>
GTA_DEV void mfma_qk(f32x16_t& d) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(d) : "i"(K0), "i"(K0 + 3), "i"(Q0), "i"(Q0 + 3));
}
template <int O0, int V0>
GTA_DEV void mfma_pv(const u32x4_t& pb) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c1:%c2], a[%c3:%c4], %0, a[%c1:%c2]" ::"v"(pb), "i"(O0), "i"(O0 + 15), "i"(V0), "i"(V0 + 3));
}



template <int DHP, int RB>
struct Smem3 {
    static constexpr int NW = 4;
    static constexpr int BM = 32 * NW * RB;
    static constexpr int NT = 64 * NW;
    static constexpr int CHP = DHP / 8;
    static constexpr int IMG = BN * DHP * 2;
    static constexpr int RING_BYTES = 6 * IMG;          // [K0 K1 K2 | V0 V1 V2]
    static constexpr int QS_BYTES = BM * DHP * 2;
    static constexpr int OFF_QS = RING_BYTES;
    static constexpr int TOP = RING_BYTES + QS_BYTES;
    static constexpr int OROW = DHP + 4;
    static constexpr int OST_ROWS = (BM * OROW * 4 <= TOP) ? BM : 128;   // rows per epilogue pass
    static constexpr int OST_BYTES = OST_ROWS * OROW * 4;
    static_assert(OST_BYTES <= TOP, "O staging must fit");
    static_assert(TOP + GTA_MAX_VIEWS * GTA_QREC * 4 <= 160 * 1024, "LDS budget");
    static constexpr int OFF_QREC = TOP;
    GTA_DEV static constexpr int off_k(int s) { return s * IMG; }
    GTA_DEV static constexpr int off_v(int s) { return (3 + s) * IMG; }
    static int total(int Nq) { return TOP + Nq * GTA_QREC * 4; }
};

// LDS-DMA of one tile image (IMG bytes, linear)
template <int DHP>
GTA_DEV void dma_image(char* dst, const char* img, int wave, int lane) {
    constexpr int PER_WAVE = BN * DHP * 2 / 1024 / 4;
    static_assert(PER_WAVE >= 1, "image must split over the waves");
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int piece = wave * PER_WAVE + i;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(img + piece * 1024 + lane * 16),
            (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
    }
}

// one 1-KiB piece of a tile image: piece PER_WAVE*wave + i
template <int DHP>
GTA_DEV void dma_piece(char* dst, const char* img, int wave, int lane, int i) {
    constexpr int PER_WAVE = BN * DHP * 2 / 1024 / 4;
    const int piece = wave * PER_WAVE + i;
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(img + piece * 1024 + lane * 16),
        (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
}

template <int DHP, int ESZ, int RB, int LAYOUT>
__global__ __launch_bounds__(256, (RB == 1 ? 2 : 1)) void gta_fwd3_kernel(const GtaFwdParams p) {
    using S = Smem3<DHP, RB>;
#define GTA_DESC(c) (LAYOUT == GTA_LAYOUT_GENERIC ? p.ctab[c] : gta_layout_desc(LAYOUT, c))
    constexpr int NW = 4, CHP = S::CHP, KS = DHP / 16, DB = DHP / 32, BM = S::BM, NT = S::NT, IMG = S::IMG;
    constexpr int RG = BM / 64, NPAR = NW / RG, QITEMS = CHP / NPAR;
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
    const char* kvimg = (const char*)p.kp + ((long)b * p.H + h) * n_tiles * (long)(2 * IMG);
    float* qrec = reinterpret_cast<float*>(smem + S::OFF_QREC);
#define K_IMG(j) (kvimg + (long)(j) * (2 * IMG))
#define V_IMG(j) (kvimg + (long)(j) * (2 * IMG) + IMG)

#ifdef GTA_ABLATE
    // per-region cycle sums (s_memtime) of wave 0: prof[blockIdx*16 + k], k = 0 start, 1 loop start, 2 loop end,
    // 3 end, 8.. region sums {wait+barrier, dma issue, R3, R1, decide, R2, rescale}
    unsigned long long t_reg[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long n_slow = 0;
    unsigned long long t_prev = 0;
#define GTA_T0() do { if (p.prof) t_prev = __builtin_amdgcn_s_memtime(); } while (0)
#define GTA_TR(k) do { if (p.prof) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_reg[k] += t_now - t_prev; t_prev = t_now; } } while (0)
#define GTA_STAMP3(k) do { if (p.prof && tid == 0) p.prof[(long)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GTA_T0() do { } while (0)
#define GTA_TR(k) do { } while (0)
#define GTA_STAMP3(k) do { } while (0)
#endif
    GTA_STAMP3(0);
    dma_image<DHP>(smem + S::off_k(0), K_IMG(0), wave, lane);
    dma_image<DHP>(smem + S::off_v(0), V_IMG(0), wave, lane);
    if (n_tiles > 1) {
        dma_image<DHP>(smem + S::off_k(1), K_IMG(1), wave, lane);
        dma_image<DHP>(smem + S::off_v(1), V_IMG(1), wave, lane);
    }
    if (n_tiles > 2) dma_image<DHP>(smem + S::off_k(2), K_IMG(2), wave, lane);

    // ---- prologue: Q tile -> rho -> prescale -> bf16 LDS tile -> MFMA B fragments (as in section 2) ----
    constexpr int RAWN = ESZ == 2 ? 1 : 2;
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
            if (c < ch_real) {
                const char* rp = qg + (long)my_t * q_rs + c * 8 * ESZ;
#pragma unroll
                for (int k2 = 0; k2 < RAWN; ++k2) qraw[it][k2] = *reinterpret_cast<const u32x4_t*>(rp + 16 * k2);
                if (p.cs_q) load_cs(GTA_DESC(c), p.cs_q + ((long)b * p.Tq + my_t) * 2 * p.nso2, qcs[it]);
            }
        }
    };
    if (NPAR == 2 && par) load_items(std::integral_constant<int, 1>{}); else load_items(std::integral_constant<int, 0>{});
    const int t_last = (q0 + BM - 1 < p.Tq ? q0 + BM - 1 : p.Tq - 1);
    const int n_first = q0 / p.Pq;
    const int n_cnt = t_last / p.Pq - n_first + 1;
    if (p.vrep_q) stage_qrec(qrec, p.vrep_q, b, p.Nq, n_first, n_cnt, p.trans_coeff ? *p.trans_coeff : 1.0f, tid, NT);
    __syncthreads();

    const float qscale = p.scale * LOG2E / (p.tau ? *p.tau : 1.0f);
    float qsq = 0.f;
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
            unpack8(qw, qr);                          // |q'|^2 of the bf16 row the MFMA will see
#pragma unroll
            for (int i2 = 0; i2 < 8; ++i2) qsq += qr[i2] * qr[i2];
        }
    };
    static_assert(NPAR == 1, "row norms assume one wave owns whole rows");
    if (NPAR == 2 && par) xform_items(std::integral_constant<int, 1>{}); else xform_items(std::integral_constant<int, 0>{});
    // |q'| of this lane's two MFMA rows (transform lane L owns row 64*wave + L; MFMA lane (l31, lh) rows 32*rb + l31)
    float qn[RB];
    {
        const float nrm = sqrtf(qsq) * 1.0001f;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) qn[rb] = __shfl(nrm, 32 * rb + l31);
    }
    __syncthreads();
    // lane-constant LDS offsets (see section 2)
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = (l31 * CHP + swz<CHP>(l31, 2 * ks + lh)) * 16;
    const int g16 = lane >> 4, p16 = lane & 15;
    int voff[DB][2];
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

    // ---- pipeline state ----
    // The accumulator file from a16 up is owned by the asm below, by literal register number.  hipcc does not
    // know: a[0:15] are left to it (it parks a few values there under VGPR pressure, lowest numbers first), and
    // tools/audit_agpr.py (run by tests/test_host_logic.py) checks in the .s that it touches nothing above:
    //   O^T  a[A_O + 16*(rb*DB + d) ..+15]        Q' fragments a[A_Q + 4*(rb*KS + ks) ..+3]
    //   K'   a[A_K + 4*(half*KS + ks) ..+3]       V' fragments a[A_V + 4*(slab*DB + d) ..+3]
    // S'/P, the packed P and the -m splats stay in arch VGPRs where the VALU reaches them.  (With builtin MFMAs
    // hipcc picks one accumulator form per kernel and pays a v_accvgpr copy per S' element or per fragment.)
    constexpr int A_O = 16, A_Q = A_O + 16 * RB * DB, A_K = A_Q + 4 * RB * KS, A_V = A_K + 8 * KS, A_END = A_V + 16 * DB;
    static_assert(A_END <= 256, "accumulator file budget");
    f32x16_t sA[RB][2], sB[RB][2];        // S' / P of two consecutive tiles (roles swap every step)
    f32x16_t msplat[RB];                  // -m_run in every element: the C operand of a row block's first MFMA
    float m_run[RB], l0[RB], l1[RB], alpha_pend[RB];
    f32x2_t l01[RB];                      // (PK_SUM) row sums of the even / odd values as one packed accumulator
    bool pend = false;
    u32x4_t pf[RB][2][2];                 // P as bf16 MFMA B fragments: [row block][key half][slab in half]
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        m_run[rb] = 0.f; l0[rb] = 0.f; l1[rb] = 0.f; alpha_pend[rb] = 1.f; l01[rb] = f32x2_t{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) msplat[rb][i] = 0.f;
    }
    asm volatile("" ::: "a0", "a255");    // (makes the kernel descriptor allocate the whole accumulator file)
    static_for<16 * RB * DB>([&](auto NC) { acc_zero<A_O + decltype(NC)::value>(); });
    {   // Q' fragments: LDS -> accumulator file
        const uint32_t qs = lds_addr(smem + S::OFF_QS);
        static_for<RB * KS>([&](auto NC) {
            constexpr int n = decltype(NC)::value, rb = n / KS, ks = n % KS;
            const int r = wave * (32 * RB) + 32 * rb + l31;
            lds_b128_to_acc<A_Q + 4 * n, 0>(qs + (r * CHP + swz<CHP>(r, 2 * ks + lh)) * 16);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // MFMAs and fillers per region (see the header of this section)
    constexpr int G3 = 2 * RB * KS;       // QK^T MFMAs per tile
    constexpr int GS = DB * RB;           // PV MFMAs per slab
    constexpr int G1 = 2 * GS, G2 = 2 * GS;
    constexpr int NE = 32 * RB;           // P values per lane per tile; value e -> (rb = e>>5, half = (e>>4)&1, r = e&15)
    constexpr int NE2 = 8 * RB;           // exponentiated in R2 (beside the DMA issue), the rest in R3
    constexpr int NDMA = 2 * DB;          // LDS-DMA pieces per wave and tile (V' image then K' image)

    constexpr int NV = 4 * DB;            // V' transpose-reads per pair of slabs
    constexpr int NKR = 2 * KS;           // K' fragment reads per tile
    // R3 item list: NV reads (cost 5), NE2/2 packs of early values (5), then (NE-NE2)/2 groups of
    // {exp, exp, add, add, pack} (33); items are dealt to the G3 gaps by cumulative issue cost
    constexpr int R3_NC0 = NE2 / 2, R3_NQ = (NE - NE2) / 2, R3_N = NV + R3_NC0 + R3_NQ;
    constexpr int R3_COST = 5 * (NV + R3_NC0) + 33 * R3_NQ;
    auto r3_first = [](int g) constexpr {      // first item of gap g (reads excluded: they go one per gap)
        int c = 0;
        for (int n = NV; n < R3_N; ++n) {
            int gg = (int)((long)c * G3 / (R3_COST - 5 * NV));
            if (gg > G3 - 1) gg = G3 - 1;
            if (gg >= g) return n;
            c += n < NV + R3_NC0 ? 5 : 33;
        }
        return R3_N;
    };
    static_assert(NV <= G3 && NV + NKR <= 2 * G1, "LDS read placement");
    constexpr int GH = G1 / 3 > 3 ? G1 / 3 : 3; // R1: gaps [0, GH) carry the LDS reads, [GH, G1) the row max
    static_assert(GH < G1 && DB <= G1 && DB <= G2, "filler placement");
    constexpr int NM = 4 * RB;                 // row-max chunks of 8 values

    auto expadd = [&](f32x16_t (&s)[RB][2], auto EC) {
        constexpr int e = decltype(EC)::value, rb = e >> 5, hh = (e >> 4) & 1, r = e & 15;
        if constexpr (ABL & 1) return;
        const float pv = __builtin_amdgcn_exp2f(s[rb][hh][r]);
        s[rb][hh][r] = pv;
        if constexpr (PK_SUM) {
            if (e & 1) l01[rb] += f32x2_t{s[rb][hh][r - 1], pv};      // one v_pk_add_f32 per pair of values
        } else {
            if (e & 1) add_f32(l1[rb], pv); else add_f32(l0[rb], pv);
        }
    };
    // exp and its row-sum add as separate fillers: placed a few instructions apart, the add does not wait for
    // the transcendental (hipcc pads a dependent instruction right behind a v_exp with an s_nop)
    auto exp_only = [&](f32x16_t (&s)[RB][2], auto EC) {
        constexpr int e = decltype(EC)::value, rb = e >> 5, hh = (e >> 4) & 1, r = e & 15;
        if constexpr (ABL & 1) return;
        s[rb][hh][r] = __builtin_amdgcn_exp2f(s[rb][hh][r]);
    };
    auto add_only = [&](f32x16_t (&s)[RB][2], auto EC) {
        constexpr int e = decltype(EC)::value, rb = e >> 5, hh = (e >> 4) & 1, r = e & 15;
        if constexpr (ABL & 1) return;
        if constexpr (PK_SUM) {
            if (e & 1) l01[rb] += f32x2_t{s[rb][hh][r - 1], s[rb][hh][r]};
        } else {
            if (e & 1) add_f32(l1[rb], s[rb][hh][r]); else add_f32(l0[rb], s[rb][hh][r]);
        }
    };
    auto cvt_pair = [&](f32x16_t (&s)[RB][2], auto KC) {       // values 2k, 2k+1 -> one packed word
        constexpr int e = 2 * decltype(KC)::value, rb = e >> 5, hh = (e >> 4) & 1, r = e & 15;
        if constexpr (ABL & 2) return;
        pf[rb][hh][r >> 3][(r & 7) >> 1] = pack_bf16x2(s[rb][hh][r], s[rb][hh][r + 1]);
    };
    auto k_read = [&](auto NC, uint32_t kbase) {               // item n: key half n&1, k step n>>1
        constexpr int n = decltype(NC)::value, hh = n & 1, ks = n >> 1;
        if constexpr (ABL & 16) return;
        lds_b128_to_acc<A_K + 4 * (hh * KS + ks), hh * 32 * CHP * 16>(kbase + koff[ks]);
    };
    auto v_read = [&](auto NC, auto SLB, uint32_t vbase) {     // item n of slabs SLB, SLB+1
        constexpr int n = decltype(NC)::value, sl = decltype(SLB)::value + n / (2 * DB), d = (n >> 1) % DB, hf = n & 1;
        if constexpr (ABL & 8) return;
        lds_tr_to_acc<A_V + 4 * (sl * DB + d) + 2 * hf, sl * 16 * CHP * 16>(vbase + voff[d][hf]);
    };
    auto qk_mfma = [&](f32x16_t (&s)[RB][2], auto GC) {        // MFMA g of S' = K' Q'^T - m_run (4 chains)
        constexpr int g = decltype(GC)::value, ks = g / (2 * RB), rb = (g % (2 * RB)) >> 1, hh = g & 1;
        if constexpr (ABL & 64) return;
        if constexpr (ks == 0) mfma_qk_first<A_K + 4 * (hh * KS + ks), A_Q + 4 * (rb * KS + ks)>(s[rb][hh], msplat[rb]);
        else mfma_qk<A_K + 4 * (hh * KS + ks), A_Q + 4 * (rb * KS + ks)>(s[rb][hh]);
    };
    auto pv_mfma = [&](auto SL, auto JC) {                      // MFMA j of slab SL: O^T += V'^T P^T
        constexpr int sl = decltype(SL)::value, j = decltype(JC)::value, d = j / RB, rb = j % RB;
        if constexpr (ABL & 128) return;
        mfma_pv<A_O + 16 * (rb * DB + d), A_V + 4 * (sl * DB + d)>(pf[rb][sl >> 1][sl & 1]);
    };
    // every reader of an asm MFMA's S' result sits behind this (and behind enough issue time; see callers)
    auto s_fence = [&](f32x16_t (&s)[RB][2]) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            asm volatile("" : "+v"(s[rb][0]));
            asm volatile("" : "+v"(s[rb][1]));
        }
    };
    auto mask_tail = [&](f32x16_t (&s)[RB][2], int j) {        // keys >= Tk of the last tile
        int kbase = j * BN + 4 * lh;
        asm volatile("" : "+v"(kbase));                         // (keeps the 32 compares inside the rare block: hipcc hoists them otherwise)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key >= p.Tk) s[rb][0][r] = -1e30f;
                if (key + 32 >= p.Tk) s[rb][1][r] = -1e30f;
            }
    };
    float mxp[RB];                                              // row max under construction
    auto max_chunk = [&](f32x16_t (&s)[RB][2], auto CC) {       // chunk c: 8 values of row block c / 4
        constexpr int c = decltype(CC)::value, rb = c >> 2, hh = (c >> 1) & 1, r = 8 * (c & 1);
        if ((c & 3) == 0) mxp[rb] = s[rb][hh][r];
        const f32x16_t& u = s[rb][hh];
        max8_f32(mxp[rb], u[r], u[r + 1], u[r + 2], u[r + 3], u[r + 4], u[r + 5], u[r + 6], u[r + 7]);
    };
    // deferred-max decision for the tile whose S' (relative to the current m_run) is in s
    auto decide = [&](f32x16_t (&s)[RB][2], bool first) {
        float mx[RB];
        bool grow = first;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            // lanes l and l+32 hold the two key halves of one row: half exchange, no LDS round trip
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mxp[rb]), __float_as_uint(mxp[rb]), false, false);
            mx[rb] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            grow = grow || (mx[rb] > DEFER_THR);
        }
        if (__builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float delta = first ? mx[rb] : fmaxf(mx[rb], 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_run[rb] += delta;
                l0[rb] *= alpha; l1[rb] *= alpha; l01[rb] *= alpha;
                alpha_pend[rb] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { s[rb][0][i] -= delta; s[rb][1][i] -= delta; msplat[rb][i] = -m_run[rb]; }
            }
            pend = true;
        }
    };

    // one pipeline step: tile i is finished, tile i+1 is started (unless LAST)
    auto step = [&](f32x16_t (&sc)[RB][2], f32x16_t (&sn)[RB][2], int i, auto LASTC) {
        constexpr bool LAST = decltype(LASTC)::value;
        // V'(i), K'(i+2) have landed (this wave's share), everyone is past R3(i-1) / R1(i-1)
        GTA_T0();
        // V'(i), K'(i+2) have landed (issued two steps ago; only the last step's pieces may still fly)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        if constexpr (!(ABL & 256)) __builtin_amdgcn_s_barrier();
        GTA_TR(0);
        if (i == 0 && n_tiles > 3) dma_image<DHP>(smem + S::off_k(0), K_IMG(3), wave, lane);   // (K'(0) is consumed)
        // this step's DMA, issued beside the MFMAs of R1 / R2: V'(i+2) and K'(i+4) into the slots of V'(i-1)
        // and K'(i+1).  Past the end the last tile is fetched again into a slot nobody reads (uniform code,
        // uniform vmcnt accounting).
        const int jv = i + 2 < n_tiles ? i + 2 : n_tiles - 1, jk = i + 4 < n_tiles ? i + 4 : n_tiles - 1;
        char* dma_dst_v = smem + S::off_v((i + 2) % 3);
        char* dma_dst_k = smem + S::off_k((i + 1) % 3);
        const char* dma_src_v = V_IMG(jv);
        const char* dma_src_k = K_IMG(jk);
        auto dma_step = [&](auto NC) {
            constexpr int n = decltype(NC)::value;
            if constexpr (ABL & 32) return;
            if constexpr (n < DB) dma_piece<DHP>(dma_dst_v, dma_src_v, wave, lane, n);
            else dma_piece<DHP>(dma_dst_k, dma_src_k, wave, lane, n - DB);
        };
        const uint32_t vbase = lds_addr(smem + S::off_v(i % 3));
        const uint32_t kbase = lds_addr(smem + S::off_k((i + 2) % 3));
        // key-norm bound of tile i+1: scalar load by hand (hipcc picks a VMEM load here, and then drains
        // vmcnt(0) -- this step's DMA pieces included -- where the value is used); landed by R3's lgkmcnt(0)
        uint32_t kn_bits;
        {
            const int kn_idx = __builtin_amdgcn_readfirstlane((b * p.H + h) * n_tiles + (i + 1 < n_tiles ? i + 1 : i));
            const float* kn_ptr = p.kn + kn_idx;
            asm volatile("s_load_dword %0, %1, 0x0" : "=s"(kn_bits) : "s"(kn_ptr) : "memory");
        }
        s_fence(sc);                                  // (keeps R3's VALU work in R3: hipcc hoists pure code otherwise)
        GTA_TR(1);
        __builtin_amdgcn_sched_barrier(0);

        // ---- R3: S'(i+1) MFMAs || V'(i) reads of slabs 0,1; packs of the early P(i); exp/sum/pack of the rest ----
        static_for<G3>([&](auto GC) {
            constexpr int g = decltype(GC)::value;
            if constexpr (!LAST) qk_mfma(sn, GC);
            if constexpr (g < NV) v_read(std::integral_constant<int, g>{}, std::integral_constant<int, 0>{}, vbase);
            constexpr int n0 = r3_first(g), n1 = r3_first(g + 1);
            static_for<n1 - n0>([&](auto DC) {
                constexpr int n = n0 + decltype(DC)::value;
                if constexpr (n < NV + R3_NC0) {
                    cvt_pair(sc, std::integral_constant<int, n - NV>{});
                } else {
                    // pair q: its two exps, then the sums and the pack of pair q-1
                    constexpr int q = n - NV - R3_NC0, e = NE2 + 2 * q;
                    exp_only(sc, std::integral_constant<int, e>{});
                    exp_only(sc, std::integral_constant<int, e + 1>{});
                    if constexpr (q > 0) {
                        add_only(sc, std::integral_constant<int, e - 2>{});
                        add_only(sc, std::integral_constant<int, e - 1>{});
                        cvt_pair(sc, std::integral_constant<int, e / 2 - 1>{});
                    }
                    if constexpr (q == R3_NQ - 1) {
                        add_only(sc, std::integral_constant<int, e>{});
                        add_only(sc, std::integral_constant<int, e + 1>{});
                        cvt_pair(sc, std::integral_constant<int, e / 2>{});
                    }
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {             // (all of P(i) is packed here, not sunk to its use in R2)
            asm volatile("" : "+v"(pf[rb][0][0]), "+v"(pf[rb][0][1]), "+v"(pf[rb][1][0]), "+v"(pf[rb][1][1]));
        }
        GTA_TR(2);
        __builtin_amdgcn_sched_barrier(0);

        // ---- R1: PV slabs 0,1 || K'(i+2) fragment reads, V'(i) reads of slabs 2,3, then the row max of S'(i+1) ----
        asm volatile("s_nop 1" ::: "memory");       // the last packs of R3 -> first MFMA reading them
        static_for<G1>([&](auto GC) {
            constexpr int g = decltype(GC)::value;
            pv_mfma(std::integral_constant<int, g / GS>{}, std::integral_constant<int, g % GS>{});
            {   // LDS reads spread over the region: V'(i) slabs 2,3 first (R2 needs them), then K'(i+2)
                constexpr int NR = NV + NKR, a0 = g * NR / G1, a1 = (g + 1) * NR / G1;
                static_for<a1 - a0>([&](auto DC) {
                    constexpr int n = a0 + decltype(DC)::value;
                    if constexpr (n < NV) v_read(std::integral_constant<int, n>{}, std::integral_constant<int, 2>{}, vbase);
                    else if constexpr (!LAST) k_read(std::integral_constant<int, n - NV>{}, kbase);
                });
            }
            if constexpr (g >= G1 - DB) dma_step(std::integral_constant<int, g - (G1 - DB)>{});     // V'(i+2) pieces
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GTA_TR(3);
        if constexpr (!LAST) {
            // Deferred max without a per-tile row max: S'(i+1) <= |q'| max_k |k'_k| - m_run (Cauchy-Schwarz, the
            // pre-pass stored the tile's key norm).  While that bound stays below BOUND_THR no exponent can
            // overflow and m_run need not move; only otherwise (or on the masked tail tile) is the true row max
            // computed and the rescale decision taken.
            const bool tail_tile = has_tail && i + 2 == n_tiles;
            asm volatile("" : "+s"(kn_bits));          // (first consumer sits behind R1's lgkmcnt(0))
            const float kn_next = __uint_as_float(kn_bits);
            bool need = tail_tile;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) need = need || (qn[rb] * kn_next - m_run[rb] > BOUND_THR);
            if (!SKIP_MAX && __builtin_amdgcn_ballot_w64(need) != 0) {
#ifdef GTA_ABLATE
                n_slow++;
#endif
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");        // (rare path; S'(i+1) MFMAs have landed)
                s_fence(sn);
                if (tail_tile) mask_tail(sn, i + 1);
                static_for<NM>([&](auto CC) { max_chunk(sn, CC); });
                decide(sn, false);
            }
            s_fence(sn);                              // (keeps R2's exps behind the decision, in R2)
        }
        GTA_TR(4);
        __builtin_amdgcn_sched_barrier(0);

        // ---- R2: PV slabs 2,3 || exp / sum of the first NE2 values of P(i+1) ----
        static_for<G2>([&](auto GC) {
            constexpr int g = decltype(GC)::value;
            pv_mfma(std::integral_constant<int, 2 + g / GS>{}, std::integral_constant<int, g % GS>{});
            if constexpr (!LAST) {
                // this gap's exps, then the sums of the previous gap's
                constexpr int e0 = g * NE2 / G2, e1 = (g + 1) * NE2 / G2, ep = g > 0 ? (g - 1) * NE2 / G2 : 0;
                static_for<e1 - e0>([&](auto DC) { exp_only(sn, std::integral_constant<int, e0 + decltype(DC)::value>{}); });
                static_for<e0 - ep>([&](auto DC) { add_only(sn, std::integral_constant<int, ep + decltype(DC)::value>{}); });
                if constexpr (g == G2 - 1)
                    static_for<e1 - e0>([&](auto DC) { add_only(sn, std::integral_constant<int, e0 + decltype(DC)::value>{}); });
            }
            {   // K'(i+4) pieces spread over the region
                constexpr int d0 = g * DB / G2, d1 = (g + 1) * DB / G2;
                static_for<d1 - d0>([&](auto DC) { dma_step(std::integral_constant<int, DB + d0 + decltype(DC)::value>{}); });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (!LAST) s_fence(sn);             // (R2's exps stay in R2, not sunk to their first use)
        GTA_TR(5);
        // a slow-path decision rescales O once the MFMAs at the old scale are in
        if (pend) {
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float a = alpha_pend[rb];
                if (rb == 0) static_for<16 * DB>([&](auto NC) { acc_scale<A_O + decltype(NC)::value>(a); });
                else         static_for<16 * DB>([&](auto NC) { acc_scale<A_O + 16 * DB * (RB - 1) + decltype(NC)::value>(a); });
                alpha_pend[rb] = 1.f;
            }
            asm volatile("s_nop 3" ::: "memory");
            pend = false;
        }
        GTA_TR(6);
    };

    // ---- tile 0 by hand: S'(0), K'(1) fragments, decision, first exps ----
    GTA_STAMP3(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    static_for<NKR>([&](auto NC) { k_read(NC, lds_addr(smem + S::off_k(0))); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    static_for<G3>([&](auto GC) { qk_mfma(sA, GC); });
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // last MFMA -> first reader / K' overwrite
    if (n_tiles > 1) static_for<NKR>([&](auto NC) { k_read(NC, lds_addr(smem + S::off_k(1))); });
    s_fence(sA);
    if (has_tail && n_tiles == 1) {
        asm volatile("" ::: "memory");
        mask_tail(sA, 0);
    }
    static_for<NM>([&](auto CC) { max_chunk(sA, CC); });
    decide(sA, true);
    pend = false;                                // O is still zero
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) alpha_pend[rb] = 1.f;
    static_for<NE2>([&](auto EC) { expadd(sA, EC); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        int i = 0;
        for (; i + 2 < n_tiles; i += 2) {
            step(sA, sB, i, std::false_type{});
            step(sB, sA, i + 1, std::false_type{});
        }
        if (i + 2 == n_tiles) {
            step(sA, sB, i, std::false_type{});
            step(sB, sA, i + 1, std::true_type{});
        } else {
            step(sA, sB, i, std::true_type{});
        }
    }

    // ---- epilogue through the O staging tile, 128 rows per pass (as in section 2) ----
    GTA_STAMP3(2);
#ifdef GTA_ABLATE
    if (p.prof && lane == 0 && wave == 0) {
#pragma unroll
        for (int k2 = 0; k2 < 7; ++k2) p.prof[(long)blockIdx.x * 16 + 8 + k2] = t_reg[k2];
        p.prof[(long)blockIdx.x * 16 + 15] = n_slow;
    }
#endif
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");             // last asm MFMAs -> O readers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // (trailing DMA pieces before the ring is reused)
    float inv_l[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const float lsum = l0[rb] + l1[rb] + l01[rb].x + l01[rb].y;
        const float l_tot = lsum + __shfl_xor(lsum, 32);
        inv_l[rb] = 1.0f / l_tot;
        if (p.lse && lh == 0) {
            const int t = q0 + wave * (32 * RB) + 32 * rb + l31;
            if (t < p.Tq) p.lse[((long)b * p.H + h) * p.Tq + t] = (m_run[rb] + __log2f(l_tot)) * LN2;
        }
    }
    float* ost = reinterpret_cast<float*>(smem);
    const bool xo = (p.flags & GTA_FLAG_V_TRANSFORM) != 0;
    constexpr int NPASS = BM / S::OST_ROWS;
    constexpr int WPP = NW / NPASS;                 // waves whose rows go in one pass
    constexpr int ERG = S::OST_ROWS / 64;           // 64-row groups per pass: item map (row group, chunk parity)
    constexpr int EPAR = NW / ERG;
    constexpr int EITEMS = CHP / EPAR;
    static_assert(ERG * EPAR == NW && EITEMS * EPAR == CHP, "epilogue item map");
    const int rgE = wave % ERG, parE = wave / ERG;
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
        const int rE = lane + 64 * rgE;
        const int tE = q0 + pass * S::OST_ROWS + rE;
        f32x2_t ocs[EITEMS][4];
        if (xo && p.cs_q && tE < p.Tq) {            // (cos, sin) of the row, prefetched before the barriers
            auto load_ocs = [&](auto PARC) {
                constexpr int PAR = decltype(PARC)::value;
#pragma unroll
                for (int it = 0; it < EITEMS; ++it) {
                    const int c = EPAR * it + PAR;
                    if (c < ch_real) load_cs(GTA_DESC(c), p.cs_q + ((long)b * p.Tq + tE) * 2 * p.nso2, ocs[it]);
                }
            };
            if (EPAR == 2 && parE) load_ocs(std::integral_constant<int, 1>{}); else load_ocs(std::integral_constant<int, 0>{});
        }
        __syncthreads();
        if (wave / WPP == pass) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = (wave % WPP) * (32 * RB) + 32 * rb + l31;
                const float il = inv_l[rb];
                auto put = [&](auto NC) {
                    constexpr int n = decltype(NC)::value, d = n >> 2, g = n & 3;
                    f32x4_t v;
                    if (rb == 0) v = acc_read4<A_O + 16 * d + 4 * g>(); else v = acc_read4<A_O + 16 * DB * (RB - 1) + 16 * d + 4 * g>();
                    v.x *= il; v.y *= il; v.z *= il; v.w *= il;
                    *reinterpret_cast<f32x4_t*>(ost + r * S::OROW + 32 * d + 8 * g + 4 * lh) = v;
                };
                static_for<4 * DB>(put);
            }
        }
        __syncthreads();
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
            gstore_chunk2<ESZ>(og + (long)tE * o_rs, c, x[0]);
        };
        auto out_items = [&](auto PARC) {
            constexpr int PAR = decltype(PARC)::value;
#pragma unroll
            for (int it = 0; it < EITEMS; ++it) {
                const int c = EPAR * it + PAR;
                if (c < ch_real && tE < p.Tq) out_item(GTA_DESC(c), c, ocs[it]);
            }
        };
        if (EPAR == 2 && parE) out_items(std::integral_constant<int, 1>{}); else out_items(std::integral_constant<int, 0>{});
    }
    GTA_STAMP3(3);
#undef GTA_T0
#undef GTA_TR
#undef GTA_STAMP3
#undef K_IMG
#undef V_IMG
}

template <int DHP, int ESZ, int RB, int LAYOUT>
int launch_fwd3(const GtaFwdParams& p, hipStream_t stream) {
    using S = Smem3<DHP, RB>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gta_fwd3_kernel<DHP, ESZ, RB, LAYOUT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, S::total(GTA_MAX_VIEWS)) != hipSuccess)
            return GTA_E_LAUNCH;
        attr_set = true;
    }
    const long n_wg = (long)p.B * p.H * p.n_qtiles;
    hipLaunchKernelGGL((gta_fwd3_kernel<DHP, ESZ, RB, LAYOUT>), dim3((unsigned)n_wg), dim3(256), S::total(p.vrep_q ? p.Nq : 0),
                       stream, p);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

}  // namespace

// layout: GTA_LAYOUT_* id chosen by the caller (gta_fwd2.hip: layout_of)
int gta_fwd3_dispatch(const GtaFwdParams& p, int dhp, int esz, int layout, hipStream_t stream) {
#define GTA_F3(D, E)                                                                                     \
    switch (layout) {                                                                                    \
        case GTA_LAYOUT_MS:  if (D == 96) return launch_fwd3<D, E, 2, (D == 96 ? GTA_LAYOUT_MS : GTA_LAYOUT_GENERIC)>(p, stream); break; \
        case GTA_LAYOUT_CL:  if (D == 64) return launch_fwd3<D, E, 2, (D == 64 ? GTA_LAYOUT_CL : GTA_LAYOUT_GENERIC)>(p, stream); break; \
        case GTA_LAYOUT_SO2: return launch_fwd3<D, E, 2, GTA_LAYOUT_SO2>(p, stream);                     \
    }                                                                                                    \
    return launch_fwd3<D, E, 2, GTA_LAYOUT_GENERIC>(p, stream);
    if (dhp == 64) { if (esz == 2) { GTA_F3(64, 2) } else { GTA_F3(64, 4) } }
    if (dhp == 96) { if (esz == 2) { GTA_F3(96, 2) } else { GTA_F3(96, 4) } }
#undef GTA_F3
    return GTA_E_UNSUPPORTED;
}
