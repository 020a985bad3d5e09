// gta_fwd2.hip -- two-stage GTA attention forward for gfx950: K/V rep pre-pass + lean flash kernel.
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
//   gta_fwd2_kernel      256 query rows per workgroup (8 waves x 32).  Prologue: rho on Q (gta.py
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
#include <type_traits>
#include "gta_common.h"
#include "gta_fwd_params.h"
#include "../../include/gta_hip.h"

// Ablation hooks (tools/bench_kernels.py ablate/timeline) exist only in -DGTA_ABLATE builds.
#ifdef GTA_ABLATE
#define GTA_DBG(bit) ((p.dbg & (bit)) != 0)
#else
#define GTA_DBG(bit) false
#endif

namespace {

constexpr int BN = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// ds_read_b64_tr_b16 through inline asm: the builtin makes hipcc drain vmcnt(0) (it cannot prove
// the read does not alias the LDS-DMA in flight), which would serialise the DMA ring.  The caller
// waits with lgkmcnt(0) + sched_barrier(0) before the first use (cdna_hip_programming.md 5.7).
template <int IMM>
GTA_DEV u32x2_t lds_tr16_b64(uint32_t addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(IMM));
    return v;
}
GTA_DEV uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// ------------------------------------------------------------------------------------------------
// shared pieces
// ------------------------------------------------------------------------------------------------
template <int ESZ>
GTA_DEV void gload_chunk2(const char* rowptr, int c, float* x) {
    if (ESZ == 2) {
        unpack8(*reinterpret_cast<const u32x4_t*>(rowptr + c * 16), x);
    } else {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(rowptr + c * 32);
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(rowptr + c * 32 + 16);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    }
}
template <int ESZ>
GTA_DEV void gstore_chunk2(char* rowptr, int c, const float* x) {
    if (ESZ == 2) {
        *reinterpret_cast<u32x4_t*>(rowptr + c * 16) = pack8(x);
    } else {
        *reinterpret_cast<f32x4_t*>(rowptr + c * 32) = f32x4_t{x[0], x[1], x[2], x[3]};
        *reinterpret_cast<f32x4_t*>(rowptr + c * 32 + 16) = f32x4_t{x[4], x[5], x[6], x[7]};
    }
}

// q-side / k-side per-view records -> LDS, trans_coeff mask folded in (gta.py:40-44,135-141).
// Only views n0 .. n0+cnt-1 (the ones a query tile touches) are staged, as records 0..cnt-1; the
// global loads are issued three at a time so the prologue pays one round trip, not one per element.
GTA_DEV int qrec_src(int e, int* kind, int* r_, int* c_) {
    // returns the vrep offset feeding record element e, and how to mask it
    if (e < 32) {
        const int ee = e & 15, r = ee >> 2, c = ee & 3;
        const int sr = (e < 16) ? c : r, sc = (e < 16) ? r : c;        // Aq = (E.m)^T, Oq = E.m
        *kind = 0; *r_ = sr; *c_ = sc;
        return GTA_VREP_INV + sr * 4 + sc;
    } else if (e < GTA_QREC_D2) {
        const int ee = e - GTA_QREC_D1, r = ee >> 2, c = ee & 3;
        *kind = c < 3 ? 1 : 2;
        return GTA_VREP_D1 + r * 3 + (c < 3 ? c : 0);
    } else if (e < GTA_QREC_D1T) {
        const int ee = e - GTA_QREC_D2, r = ee >> 3, c = ee & 7;
        *kind = c < 5 ? 1 : 2;
        return GTA_VREP_D2 + r * 5 + (c < 5 ? c : 0);
    } else if (e < GTA_QREC_D2T) {
        const int ee = e - GTA_QREC_D1T, r = ee >> 2, c = ee & 3;
        *kind = c < 3 ? 1 : 2;
        return GTA_VREP_D1 + (c < 3 ? c : 0) * 3 + r;
    } else {
        const int ee = e - GTA_QREC_D2T, r = ee >> 3, c = ee & 7;
        *kind = c < 5 ? 1 : 2;
        return GTA_VREP_D2 + (c < 5 ? c : 0) * 5 + r;
    }
}
GTA_DEV void stage_qrec(float* qrec, const float* vrep_q, int b, int Nq, int n0, int cnt, float tc, int tid,
                        int nthreads) {
    const int total = cnt * GTA_QREC;
    const float* base = vrep_q + ((long)b * Nq + n0) * GTA_VREP_STRIDE;
    for (int i0 = tid; i0 < total; i0 += 3 * nthreads) {
        float val[3];
        int kind[3], rr[3], cc[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = i0 + u * nthreads;
            kind[u] = 3;
            if (i < total) {
                const int n = i / GTA_QREC, e = i - n * GTA_QREC;
                val[u] = base[(long)n * GTA_VREP_STRIDE + qrec_src(e, &kind[u], &rr[u], &cc[u])];
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = i0 + u * nthreads;
            if (kind[u] == 3) continue;
            float v = val[u];
            if (kind[u] == 0) v *= (rr[u] == 3) ? (cc[u] == 3 ? 1.f : 0.f) : (cc[u] == 3 ? tc : 1.f);
            else if (kind[u] == 2) v = 0.f;
            qrec[i] = v;
        }
    }
}
GTA_DEV void stage_krec(float* krec, const float* vrep_k, int b, int Nk, float tc, int tid, int nthreads) {
    for (int i = tid; i < Nk * GTA_KREC; i += nthreads) {
        const int n = i / GTA_KREC, e = i - n * GTA_KREC;
        const float* src = vrep_k + ((long)b * Nk + n) * GTA_VREP_STRIDE;
        float val = 0.f;
        if (e < 16) {
            const int r = e >> 2, c = e & 3;
            const float m = (r == 3) ? (c == 3 ? 1.f : 0.f) : (c == 3 ? tc : 1.f);
            val = src[GTA_VREP_REP + e] * m;
        } else if (e < GTA_KREC_D2) {
            const int ee = e - GTA_KREC_D1, r = ee >> 2, c = ee & 3;
            val = c < 3 ? src[GTA_VREP_D1 + r * 3 + c] : 0.f;
        } else {
            const int ee = e - GTA_KREC_D2, r = ee >> 3, c = ee & 7;
            val = c < 5 ? src[GTA_VREP_D2 + r * 5 + c] : 0.f;
        }
        krec[i] = val;
    }
}

// ================================================================================================
// 1. K/V pre-pass
// ================================================================================================
template <int DHP, int ESZ>
struct PrepSmem {
    static constexpr int CHP = DHP / 8;
    static constexpr int RAW_UNITS = DHP * ESZ / 16;
    static constexpr int RAW_BYTES = BN * DHP * ESZ;
    static constexpr int OFF_KREC = 0;
    static constexpr int KREC_BYTES = GTA_MAX_VIEWS * GTA_KREC * 4;
    static constexpr int OFF_RAWK = KREC_BYTES;
    static constexpr int OFF_RAWV = OFF_RAWK + RAW_BYTES;
    // bf16 input: a raw unit and its image unit have the same (row, position) -> transform in place;
    // fp32 input: the image (half the bytes) gets its own region
    static constexpr int IMG = BN * DHP * 2;
    static constexpr int OFF_IMGK = (ESZ == 2) ? OFF_RAWK : OFF_RAWV + RAW_BYTES;
    static constexpr int OFF_IMGV = (ESZ == 2) ? OFF_RAWV : OFF_IMGK + IMG;
    static constexpr int TOTAL = (ESZ == 2) ? OFF_RAWV + RAW_BYTES : OFF_IMGV + IMG;
};

template <int DHP, int ESZ>
__global__ __launch_bounds__(256) void gta_kv_prep_kernel(const GtaFwdParams p) {
    using S = PrepSmem<DHP, ESZ>;
    constexpr int CHP = S::CHP, U = S::RAW_UNITS;
    constexpr int IMG = BN * DHP * 2;                       // bytes of one bf16 tile image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int n_tiles = gridDim.x;

    const char* kg = (const char*)p.k + ((long)b * p.k_sb + (long)h * p.k_sh) * ESZ;
    const char* vg = (const char*)p.v + ((long)b * p.v_sb + (long)h * p.v_sh) * ESZ;
    const long k_rs = p.k_st * ESZ, v_rs = p.v_st * ESZ;
    const int ch_real = p.dh >> 3, real_units = p.dh * ESZ / 16;

    // raw rows -> LDS (coalesced LDS-DMA; the per-lane source address carries the swizzle)
    constexpr int NI = BN * U / 256;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int u0 = (wave * NI + i) * 64, u = u0 + lane;
        const int r = u / U, pos = u - r * U;
        constexpr int tz = (U % 16 == 0) ? 4 : (U % 8 == 0) ? 3 : (U % 4 == 0) ? 2 : (U % 2 == 0) ? 1 : 0;
        const int rot = (r >> (4 - tz)) & ((1 << tz) - 1);
        int gu = pos - rot;
        gu = gu < 0 ? gu + U : gu;
        gu = gu < real_units ? gu : real_units - 1;
        int gr = j * BN + r;
        gr = gr < p.Tk ? gr : p.Tk - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kg + (long)gr * k_rs + gu * 16),
                                         (__attribute__((address_space(3))) void*)(smem + S::OFF_RAWK + u0 * 16), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vg + (long)gr * v_rs + gu * 16),
                                         (__attribute__((address_space(3))) void*)(smem + S::OFF_RAWV + u0 * 16), 16, 0, 0);
    }
    float* krec = reinterpret_cast<float*>(smem + S::OFF_KREC);
    if (p.vrep_k) stage_krec(krec, p.vrep_k, b, p.Nk, p.trans_coeff ? *p.trans_coeff : 1.0f, tid, 256);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const bool xv = (p.flags & GTA_FLAG_V_TRANSFORM) != 0;
    const int r = lane;
    const int t_raw = j * BN + r;
    const bool valid = t_raw < p.Tk;
    const int t = valid ? t_raw : p.Tk - 1;
    const int n = view_of(t, p.Pk, p.invPk);
    const float* rec = krec + n * GTA_KREC;
    char* kimg_l = smem + S::OFF_IMGK;
    char* vimg_l = smem + S::OFF_IMGV;
#pragma unroll
    for (int it = 0; it < CHP / 4; ++it) {
        const int c = wave + 4 * it;
        float x[2][8];
        if (c < ch_real && valid) {
            const uint32_t desc = p.ctab[c];
            if (ESZ == 2) {
                unpack8(*reinterpret_cast<const u32x4_t*>(smem + S::OFF_RAWK + (r * U + swz<U>(r, c)) * 16), x[0]);
                unpack8(*reinterpret_cast<const u32x4_t*>(smem + S::OFF_RAWV + (r * U + swz<U>(r, c)) * 16), x[1]);
            } else {
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    const char* raw = smem + (w2 ? S::OFF_RAWV : S::OFF_RAWK);
                    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(raw + (r * U + swz<U>(r, 2 * c)) * 16);
                    const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(raw + (r * U + swz<U>(r, 2 * c + 1)) * 16);
                    x[w2][0] = a.x; x[w2][1] = a.y; x[w2][2] = a.z; x[w2][3] = a.w;
                    x[w2][4] = bb.x; x[w2][5] = bb.y; x[w2][6] = bb.z; x[w2][7] = bb.w;
                }
            }
            if (desc) {
                f32x2_t cs[4];
                if (p.cs_k) load_cs(desc, p.cs_k + ((long)b * p.Tk + t) * 2 * p.nso2, cs);
                if (xv) chunk_apply<false, 2>(desc, rec + GTA_KREC_B, rec + GTA_KREC_D1, rec + GTA_KREC_D2, cs, x);
                else    chunk_apply<false, 1>(desc, rec + GTA_KREC_B, rec + GTA_KREC_D1, rec + GTA_KREC_D2, cs, x);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { x[0][i] = 0.f; x[1][i] = 0.f; }
        }
        const int off = (r * CHP + swz<CHP>(r, c)) * 16;
        *reinterpret_cast<u32x4_t*>(kimg_l + off) = pack8(x[0]);
        *reinterpret_cast<u32x4_t*>(vimg_l + off) = pack8(x[1]);
    }
    __syncthreads();
    // LDS image -> workspace, 1 KiB contiguous per wave-instruction
    char* gimg = (char*)p.kp + (((long)b * p.H + h) * n_tiles + j) * (2L * IMG);
    constexpr int PIECES = IMG / 1024;            // per image
#pragma unroll
    for (int i = 0; i < (2 * PIECES + 3) / 4; ++i) {
        const int piece = wave + 4 * i;            // 0 .. 2*PIECES-1 : K' pieces then V' pieces
        if (piece < 2 * PIECES) {
            const char* src = (piece < PIECES ? kimg_l + piece * 1024 : vimg_l + (piece - PIECES) * 1024) + lane * 16;
            *reinterpret_cast<u32x4_t*>(gimg + piece * 1024 + lane * 16) = *reinterpret_cast<const u32x4_t*>(src);
        }
    }
}

// ================================================================================================
// 2. lean flash kernel
// ================================================================================================
constexpr int NSTAGE = 3;

template <int DHP, int NW>
struct Smem2 {
    static constexpr int BM = 32 * NW;
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
    // layout: [ring | q-side rep records (runtime size: Nq records)]
    static constexpr int OFF_RING = 0;
    static constexpr int OFF_QS = STAGE;
    static constexpr int OFF_QREC = RING_BYTES;
    static int total(int Nq) { return RING_BYTES + Nq * GTA_QREC * 4; }
};

// issue the LDS-DMA of one K'/V' tile image pair (STAGE bytes, linear) into ring stage `st`
template <int DHP, int NW>
GTA_DEV void dma_stage(char* ring, int st, const char* img, int wave, int lane) {
    using S = Smem2<DHP, NW>;
    constexpr int PIECES = S::STAGE / 1024;             // 1 KiB per wave-instruction
    constexpr int PER_WAVE = PIECES / NW;
    static_assert(PIECES % NW == 0, "stage must split evenly over the waves");
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int piece = wave * PER_WAVE + i;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(img + piece * 1024 + lane * 16),
            (__attribute__((address_space(3))) void*)(ring + st * S::STAGE + piece * 1024), 16, 0, 0);
    }
}

template <int DHP, int ESZ, int NW, int LAYOUT>
__global__ __launch_bounds__(64 * NW, 2) void gta_fwd2_kernel(const GtaFwdParams p) {
    using S = Smem2<DHP, NW>;
    // chunk descriptor: a compile-time constant for the shipped layouts (c is constant per unrolled item)
#define GTA_DESC(c) (LAYOUT == GTA_LAYOUT_GENERIC ? p.ctab[c] : gta_layout_desc(LAYOUT, c))
    constexpr int CHP = S::CHP, KS = DHP / 16, DB = DHP / 32, BM = S::BM, NT = S::NT;
    constexpr int DMA_PER_WAVE = S::STAGE / 1024 / NW;
    constexpr int QITEMS = (BM / 64) * CHP / NW;         // (row group, chunk) items per wave = CHP/2
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
#else
#define GTA_STAMP(k) do { } while (0)
#endif
    GTA_STAMP(0);
    dma_stage<DHP, NW>(ring, 0, kvimg, wave, lane);                   // tile 0 on its way

    // ---- issue every prologue global load up front (one latency exposure, not one per item) ----
    constexpr int RAWN = ESZ == 2 ? 1 : 2;
    u32x4_t qraw[QITEMS][RAWN];
    f32x2_t qcs[QITEMS][4];
    int qt_row[QITEMS];
    // item map: wave -> (row group rg = wave % RG, chunk parity par = wave / RG); item it -> chunk 2*it+par.
    // par takes two values: each gets its own straight-line code path in which c is a constant.
    constexpr int RG = BM / 64;
    static_assert(NW == 2 * RG && QITEMS * 2 == CHP, "item map assumes two chunk parities");
    const int rg = wave % RG, par = wave / RG;
    const int my_r = lane + 64 * rg;
    {
        int t = q0 + my_r;
        t = t < p.Tq ? t : p.Tq - 1;
#pragma unroll
        for (int it = 0; it < QITEMS; ++it) qt_row[it] = t;
    }
    auto load_items = [&](auto PARC) {
        constexpr int PAR = decltype(PARC)::value;
#pragma unroll
        for (int it = 0; it < QITEMS; ++it) {
            const int c = 2 * it + PAR;
            if (c < ch_real && !GTA_DBG(32u)) {
                const char* rp = qg + (long)qt_row[it] * q_rs + c * 8 * ESZ;
#pragma unroll
                for (int k2 = 0; k2 < RAWN; ++k2) qraw[it][k2] = *reinterpret_cast<const u32x4_t*>(rp + 16 * k2);
                if (p.cs_q) load_cs(GTA_DESC(c), p.cs_q + ((long)b * p.Tq + qt_row[it]) * 2 * p.nso2, qcs[it]);
            }
        }
    };
    if (par) load_items(std::integral_constant<int, 1>{}); else load_items(std::integral_constant<int, 0>{});
    // views touched by this query tile: records are staged relative to n_first
    const int t_last = (q0 + BM - 1 < p.Tq ? q0 + BM - 1 : p.Tq - 1);
    const int n_first = q0 / p.Pq;
    const int n_cnt = t_last / p.Pq - n_first + 1;
    if (p.vrep_q && !GTA_DBG(128u)) stage_qrec(qrec, p.vrep_q, b, p.Nq, n_first, n_cnt, p.trans_coeff ? *p.trans_coeff : 1.0f, tid, NT);
    __syncthreads();

    GTA_STAMP(1);
    // ---- Q: rho, prescale, bf16 -> LDS ----
    const float qscale = p.scale * LOG2E / (p.tau ? *p.tau : 1.0f);
    auto xform_items = [&](auto PARC) {
        constexpr int PAR = decltype(PARC)::value;
        char* qs = smem + S::OFF_QS;
        const int r = my_r;
#pragma unroll
        for (int it = 0; it < QITEMS; ++it) {
            const int c = 2 * it + PAR;
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
                    const int n = view_of(qt_row[it], p.Pq, p.invPq) - n_first;
                    const float* rec = qrec + n * GTA_QREC;
                    chunk_apply<false, 1>(desc, rec + GTA_QREC_A, rec + GTA_QREC_D1, rec + GTA_QREC_D2, qcs[it], x);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) x[0][i] *= qscale;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[0][i] = 0.f;
            }
            *reinterpret_cast<u32x4_t*>(qs + (r * CHP + swz<CHP>(r, c)) * 16) = pack8(x[0]);
        }
    };
    if (par) xform_items(std::integral_constant<int, 1>{}); else xform_items(std::integral_constant<int, 0>{});
    __syncthreads();      // (also drains tile 0's DMA: harmless)
    bf16x8_t qf[KS];
    {
        const char* qs = smem + S::OFF_QS;
        const int r = wave * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[ks] = *reinterpret_cast<const bf16x8_t*>(qs + (r * CHP + swz<CHP>(r, 2 * ks + lh)) * 16);
    }
    __syncthreads();      // Q staging (ring stages 1..2) is free again
    if (n_tiles > 1) dma_stage<DHP, NW>(ring, 1, kvimg + (long)S::STAGE, wave, lane);
    GTA_STAMP(2);

    f32x16_t oacc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

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

    for (int j = 0; j < n_tiles; ++j) {
        // tile j has landed (only tile j+1's pieces may still be in flight), everyone is past tile j-1
        const bool dbg_nodma = GTA_DBG(1u);
        if (!dbg_nodma) {
            if (j + 1 < n_tiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
            else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!GTA_DBG(2u)) __builtin_amdgcn_s_barrier();
        if (!dbg_nodma && j + 2 < n_tiles)
            dma_stage<DHP, NW>(ring, (j + 2) % NSTAGE, kvimg + (long)(j + 2) * S::STAGE, wave, lane);

        const char* kf = ring + (dbg_nodma ? 0 : (j % NSTAGE)) * S::STAGE;
        const char* vf = kf + S::IMG;

        // ---- S^T = K' Q'^T : every fragment read is issued before the first MFMA ----
        f32x16_t s0, s1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
        if (!GTA_DBG(16u)) {
            bf16x8_t ka[KS], kb2[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                ka[ks] = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks]);
                kb2[ks] = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks] + 32 * CHP * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks], qf[ks], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb2[ks], qf[ks], s1, 0, 0, 0);
            }
        }
        // V' block 0 transpose-reads fly under the softmax
        const uint32_t vbase = lds_addr(vf);
        constexpr int SL = 16 * CHP * 16;                     // bytes between 16-key slabs
        u32x2_t vlo[DB][4], vhi[DB][4];
        const bool dbg_nopv = GTA_DBG(8u);
        if (!dbg_nopv) {
            const uint32_t a0 = vbase + voff[0][0], a1 = vbase + voff[0][1];
            vlo[0][0] = lds_tr16_b64<0>(a0);      vhi[0][0] = lds_tr16_b64<0>(a1);
            vlo[0][1] = lds_tr16_b64<SL>(a0);     vhi[0][1] = lds_tr16_b64<SL>(a1);
            vlo[0][2] = lds_tr16_b64<2 * SL>(a0); vhi[0][2] = lds_tr16_b64<2 * SL>(a1);
            vlo[0][3] = lds_tr16_b64<3 * SL>(a0); vhi[0][3] = lds_tr16_b64<3 * SL>(a1);
        }
        if (j == n_tiles - 1 && (p.Tk & (BN - 1))) {
            const int kbase = j * BN + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key >= p.Tk) s0[r] = -1e30f;
                if (key + 32 >= p.Tk) s1[r] = -1e30f;
            }
        }

        // ---- online softmax ----
        if (GTA_DBG(4u)) {     // ablation: no max / exp / rescale
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] *= 1e-3f; s1[r] *= 1e-3f; }
            l_run += s0[0];
        } else {
        float mx = s0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s1[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_new); rs += s0[r]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1[r] = __builtin_amdgcn_exp2f(s1[r] - m_new); rs += s1[r]; }
        l_run = l_run * alpha + rs;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;
        }

        bf16x8_t pf[2][2];
        {
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

        // ---- O^T += V'^T P^T ; A = V'^T via transpose-read of the row-major V' image ----
        // 16-lane group g reads [4 keys][16 channels]: lane p supplies key row (p>>2), channels
        // 4*(p&3)..+3; it receives channel (p) x 4 keys.  k-slot e of slab (kb,t) in half h is key
        // 32kb + 16t + 8(e>>2) + 4h + (e&3): two reads (e>>2 = 0, 1).
        if (!dbg_nopv)
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            if (d + 1 < DB) {                                  // next block's reads before this block's MFMAs
                const uint32_t a0 = vbase + voff[d + 1][0], a1 = vbase + voff[d + 1][1];
                vlo[d + 1][0] = lds_tr16_b64<0>(a0);      vhi[d + 1][0] = lds_tr16_b64<0>(a1);
                vlo[d + 1][1] = lds_tr16_b64<SL>(a0);     vhi[d + 1][1] = lds_tr16_b64<SL>(a1);
                vlo[d + 1][2] = lds_tr16_b64<2 * SL>(a0); vhi[d + 1][2] = lds_tr16_b64<2 * SL>(a1);
                vlo[d + 1][3] = lds_tr16_b64<3 * SL>(a0); vhi[d + 1][3] = lds_tr16_b64<3 * SL>(a1);
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // LDS returns in order: block d landed
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {                  // slab sl = 2*kb + t
                const u32x4_t av = {vlo[d][sl].x, vlo[d][sl].y, vhi[d][sl].x, vhi[d][sl].y};
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pf[sl >> 1][sl & 1],
                                                                  oacc[d], 0, 0, 0);
            }
        }
    }

    GTA_STAMP(3);
    if (GTA_DBG(256u)) {                                                // ablation: no epilogue at all
        if (oacc[0][0] == 123.f) p.lse[0] = l_run;
        return;
    }
    // ---- epilogue through the O staging tile (all rows at once when it fits the ring) ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv_l = 1.0f / l_tot;
    if (p.lse && lh == 0) {
        const int t = q0 + wave * 32 + l31;
        if (t < p.Tq) p.lse[((long)b * p.H + h) * p.Tq + t] = (m_run + __log2f(l_tot)) * LN2;
    }
    float* ost = reinterpret_cast<float*>(smem + S::OFF_RING);
    const bool xo = (p.flags & GTA_FLAG_V_TRANSFORM) != 0;
    constexpr int NPASS = BM / S::OST_ROWS;
    constexpr int WPP = NW / NPASS;                                  // waves whose rows go in one pass
    constexpr int EITEMS = (S::OST_ROWS / 64) * CHP / NW;
    static_assert((S::OST_ROWS / 64) * CHP % NW == 0, "epilogue items must split evenly");
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
        // per-token (cos,sin): the prologue's registers when the item map is the same (one pass),
        // otherwise prefetched here before the barriers
        constexpr bool SAMEMAP = (NPASS == 1);
        static_assert(!SAMEMAP || EITEMS == QITEMS, "epilogue must reuse the prologue item map");
        f32x2_t ocs[EITEMS][4];
        if (!SAMEMAP && xo && p.cs_q) {
#pragma unroll
            for (int it = 0; it < EITEMS; ++it) {
                const int item = wave + NW * it;
                const int c = item / (S::OST_ROWS / 64);
                const int t = q0 + pass * S::OST_ROWS + lane + 64 * (item % (S::OST_ROWS / 64));
                if (c < ch_real && t < p.Tq) load_cs(p.ctab[c], p.cs_q + ((long)b * p.Tq + t) * 2 * p.nso2, ocs[it]);
            }
        }
        __syncthreads();
        if (wave / WPP == pass) {
            const int r = (wave % WPP) * 32 + l31;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4_t v = {oacc[d][4 * g] * inv_l, oacc[d][4 * g + 1] * inv_l,
                                       oacc[d][4 * g + 2] * inv_l, oacc[d][4 * g + 3] * inv_l};
                    *reinterpret_cast<f32x4_t*>(ost + r * S::OROW + 32 * d + 8 * g + 4 * lh) = v;
                }
        }
        __syncthreads();
        // rho_q^-1 on one (row, chunk) item and the store
        auto out_item = [&](const uint32_t desc, const int c, const int r, const int t, const f32x2_t* cs) {
            float x[1][8];
            const f32x4_t a = *reinterpret_cast<const f32x4_t*>(ost + r * S::OROW + 8 * c);
            const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(ost + r * S::OROW + 8 * c + 4);
            x[0][0] = a.x; x[0][1] = a.y; x[0][2] = a.z; x[0][3] = a.w;
            x[0][4] = bb.x; x[0][5] = bb.y; x[0][6] = bb.z; x[0][7] = bb.w;
            if (xo && desc) {
                const int n = view_of(t, p.Pq, p.invPq) - n_first;
                const float* rec = qrec + n * GTA_QREC;
                chunk_apply<true, 1>(desc, rec + GTA_QREC_O, rec + GTA_QREC_D1T, rec + GTA_QREC_D2T, cs, x);
            }
            if (!GTA_DBG(64u) || x[0][0] == 123.f) gstore_chunk2<ESZ>(og + (long)t * o_rs, c, x[0]);
        };
        if constexpr (SAMEMAP) {
            // same item map as the prologue: c = 2*it + par is a constant in each code path, and the
            // per-token (cos,sin) registers loaded there are reused
            auto out_items = [&](auto PARC) {
                constexpr int PAR = decltype(PARC)::value;
                const int t = q0 + my_r;
#pragma unroll
                for (int it = 0; it < QITEMS; ++it) {
                    const int c = 2 * it + PAR;
                    if (c < ch_real && t < p.Tq) out_item(GTA_DESC(c), c, my_r, t, qcs[it]);
                }
            };
            if (par) out_items(std::integral_constant<int, 1>{}); else out_items(std::integral_constant<int, 0>{});
        } else {
#pragma unroll
            for (int it = 0; it < EITEMS; ++it) {
                const int item = wave + NW * it;
                const int c = item / (S::OST_ROWS / 64);
                const int r = lane + 64 * (item % (S::OST_ROWS / 64));
                const int t = q0 + pass * S::OST_ROWS + r;
                if (c < ch_real && t < p.Tq) out_item(p.ctab[c], c, r, t, ocs[it]);
            }
        }
    }
    GTA_STAMP(4);
#undef GTA_STAMP
}

template <int DHP, int ESZ>
int launch_prep(const GtaFwdParams& p, hipStream_t stream) {
    using S = PrepSmem<DHP, ESZ>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gta_kv_prep_kernel<DHP, ESZ>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, S::TOTAL) != hipSuccess) return GTA_E_LAUNCH;
        attr_set = true;
    }
    const int n_tiles = (p.Tk + BN - 1) / BN;
    hipLaunchKernelGGL((gta_kv_prep_kernel<DHP, ESZ>), dim3(n_tiles, p.H, p.B), dim3(256), S::TOTAL, stream, p);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}
template <int DHP, int ESZ, int NW, int LAYOUT>
int launch_fwd2(const GtaFwdParams& p, hipStream_t stream) {
    using S = Smem2<DHP, NW>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gta_fwd2_kernel<DHP, ESZ, NW, LAYOUT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, S::total(GTA_MAX_VIEWS)) != hipSuccess)
            return GTA_E_LAUNCH;
        attr_set = true;
    }
    const long n_wg = (long)p.B * p.H * p.n_qtiles;
    hipLaunchKernelGGL((gta_fwd2_kernel<DHP, ESZ, NW, LAYOUT>), dim3((unsigned)n_wg), dim3(64 * NW), S::total(p.vrep_q ? p.Nq : 0),
                       stream, p);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

}  // namespace

long gta_fwd2_workspace_bytes(int B, int H, int Tk, int dhp) {
    const long n_tiles = (Tk + BN - 1) / BN;
    return (long)B * H * n_tiles * 2L * BN * dhp * 2;
}
int gta_fwd2_lds_bytes(int dhp, int nq) {
    switch (dhp) {
        case 32: return Smem2<32, 4>::total(nq);
        case 64: return Smem2<64, 4>::total(nq);
        case 96: return Smem2<96, 4>::total(nq);
        case 128: return Smem2<128, 4>::total(nq);
    }
    return -1;
}

// prep (unless the caller says K'/V' images are already in the workspace) + flash.
// nw = waves per workgroup of the flash kernel: 4 (128 query rows, two workgroups share a CU) or 8.
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

template <int DHP, int ESZ>
static int launch_flash(const GtaFwdParams& p, int nw, hipStream_t stream) {
    if (nw == 8) return launch_fwd2<DHP, ESZ, 8, GTA_LAYOUT_GENERIC>(p, stream);
    switch (layout_of(p, DHP)) {
        case GTA_LAYOUT_MS:  if (DHP == 96) return launch_fwd2<DHP, ESZ, 4, (DHP == 96 ? GTA_LAYOUT_MS : GTA_LAYOUT_GENERIC)>(p, stream); break;
        case GTA_LAYOUT_CL:  if (DHP == 64) return launch_fwd2<DHP, ESZ, 4, (DHP == 64 ? GTA_LAYOUT_CL : GTA_LAYOUT_GENERIC)>(p, stream); break;
        case GTA_LAYOUT_SO2: return launch_fwd2<DHP, ESZ, 4, GTA_LAYOUT_SO2>(p, stream);
    }
    return launch_fwd2<DHP, ESZ, 4, GTA_LAYOUT_GENERIC>(p, stream);
}

int gta_fwd2_dispatch(GtaFwdParams& p, int dhp, int esz, bool run_prep, bool run_flash, int nw, hipStream_t stream) {
    p.n_qtiles = (p.Tq + 32 * nw - 1) / (32 * nw);
    int rc = GTA_OK;
#define GTA_CASE2(D)                                                                    \
    case D:                                                                             \
        if (run_prep) rc = (esz == 2) ? launch_prep<D, 2>(p, stream) : launch_prep<D, 4>(p, stream); \
        if (rc == GTA_OK && run_flash) rc = (esz == 2) ? launch_flash<D, 2>(p, nw, stream) : launch_flash<D, 4>(p, nw, stream); \
        return rc;
    switch (dhp) {
        GTA_CASE2(32)
        GTA_CASE2(64)
        GTA_CASE2(96)
        GTA_CASE2(128)
    }
#undef GTA_CASE2
    return GTA_E_UNSUPPORTED;
}
