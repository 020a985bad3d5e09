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
// same read into the accumulator file (MFMA A operands may live there; keeps the arch VGPRs for the softmax)
template <int IMM>
GTA_DEV u32x2_t lds_tr16_b64_acc(uint32_t addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=a"(v) : "v"(addr), "i"(IMM));
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
    float ksq = 0.f;                                     // this thread's share of |k'_r|^2 (bf16-rounded values)
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
        const u32x4_t kw = pack8(x[0]);
        *reinterpret_cast<u32x4_t*>(kimg_l + off) = kw;
        *reinterpret_cast<u32x4_t*>(vimg_l + off) = pack8(x[1]);
        float kr[8];
        unpack8(kw, kr);
#pragma unroll
        for (int i = 0; i < 8; ++i) ksq += kr[i] * kr[i];
    }
    // per-tile bound for the flash kernel's deferred max: max over the tile's keys of |k'| (exactly the rows
    // the MFMA will see).  krec is dead by now (every thread is past its last chunk_apply after the barrier).
    __syncthreads();
    float* rowsq = reinterpret_cast<float*>(smem + S::OFF_KREC);
    if (p.kn) rowsq[wave * 64 + lane] = ksq;
    __syncthreads();
    if (p.kn && wave == 0) {
        float tot = rowsq[lane] + rowsq[64 + lane] + rowsq[128 + lane] + rowsq[192 + lane];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot = fmaxf(tot, __shfl_xor(tot, o));
        if (lane == 0) p.kn[((long)b * p.H + h) * n_tiles + j] = sqrtf(tot) * 1.0001f;
    }
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
// 4 waves per workgroup; every wave owns RB blocks of 32 query rows and re-uses each K'/V' fragment it
// reads from LDS for all of them.  Measured (profiles/r01): with RB = 1 and eight waves per CU the kernel
// is bound by LDS fragment reads (~35 cycles per ds_read_b128 under load, 8 x 24 KB per tile per CU --
// as long as all the MFMAs of the tile).  RB = 2 halves the LDS bytes per MFMA and runs one wave per SIMD
// with the whole 512-entry register file, so one block's softmax VALU can issue under the other's MFMAs.
// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <class F, int... Is>
GTA_DEV void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
GTA_DEV void static_for(F&& f) { static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// GTA_ABL: ablation mask for timing experiments only (results are WRONG with any bit set):
//   1 no exp/sum   2 no pack   4 no row max / decision   8 no V' reads   16 no K' reads   32 no DMA in the loop
//   64 no QK^T MFMAs   128 no PV MFMAs   256 no barrier
#ifndef GTA_ABL
#define GTA_ABL 0
#endif
constexpr int ABL = GTA_ABL;
constexpr int NSTAGE = 3;
// Skewed tile loop in the 128-row kernel (QK^T of tile j+1 beside the softmax of tile j).  Measured on MI355X
// (MSN encoder, B=32): 214 us vs 212 us un-skewed at 20 key tiles, 363 vs 372 us at 40, 105 vs 96 us at 5 --
// the softmax VALU is already hidden by the co-resident wave (ablation: removing every exp saves 2 %), what is
// left is MFMA + LDS-DMA issue + the per-workgroup prologue/epilogue.  Off by default (it also spills ~14 VGPRs
// outside the loop at dh = 96); build with -DGTA_PIPE1=1 to select it.
#ifndef GTA_PIPE1
#define GTA_PIPE1 0
#endif
constexpr bool PIPE1 = GTA_PIPE1 != 0;
constexpr float DEFER_THR = 8.0f;      // (pipelined kernel) running max moves when a row's tile max exceeds it by this
constexpr float BOUND_THR = 96.0f;     // exp2 arguments stay below this without looking at the scores

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

// online softmax of one 32-row block's tile (S in log2 units), O rescale, P -> bf16 MFMA B fragments.
// key of register r = kbase + (r&3) + 8(r>>2) (+32 for s1); keys >= Tk are masked when `tail`.
template <int DHP>
GTA_DEV void softmax_tile(f32x16_t& s0, f32x16_t& s1, float& m_run, float& l_run, f32x16_t (&oacc)[DHP / 32],
                          bf16x8_t (&pf)[2][2], bool tail, int kbase, int Tk) {
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
    for (int d = 0; d < DHP / 32; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;
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
#else
#define GTA_STAMP(k) do { } while (0)
#endif
    GTA_STAMP(0);
    dma_stage<DHP, RB>(ring, 0, kvimg, wave, lane);                   // tile 0 on its way

    // ---- prologue: every global load is issued up front (one latency exposure, not one per item).
    // item map: wave -> (row group rg = wave % RG, chunk parity par = wave / RG); item it -> chunk
    // NPAR*it + par: a constant in each of the NPAR straight-line code paths.
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
            if (c < ch_real && !GTA_DBG(32u)) {
                const char* rp = qg + (long)my_t * q_rs + c * 8 * ESZ;
#pragma unroll
                for (int k2 = 0; k2 < RAWN; ++k2) qraw[it][k2] = *reinterpret_cast<const u32x4_t*>(rp + 16 * k2);
                if (p.cs_q) load_cs(GTA_DESC(c), p.cs_q + ((long)b * p.Tq + my_t) * 2 * p.nso2, qcs[it]);
            }
        }
    };
    if (NPAR == 2 && par) load_items(std::integral_constant<int, 1>{}); else load_items(std::integral_constant<int, 0>{});
    // views touched by this query tile: records are staged relative to n_first
    const int t_last = (q0 + BM - 1 < p.Tq ? q0 + BM - 1 : p.Tq - 1);
    const int n_first = q0 / p.Pq;
    const int n_cnt = t_last / p.Pq - n_first + 1;
    if (p.vrep_q && !GTA_DBG(128u)) stage_qrec(qrec, p.vrep_q, b, p.Nq, n_first, n_cnt, p.trans_coeff ? *p.trans_coeff : 1.0f, tid, NT);
    __syncthreads();

    GTA_STAMP(1);
    // ---- Q: rho, prescale, bf16 -> LDS ----
    const float qscale = p.scale * LOG2E / (p.tau ? *p.tau : 1.0f);
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
    float qn[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int r = wave * (32 * RB) + 32 * rb + l31;
        qn[rb] = sqrtf(qsq_l[r] + (NPAR == 2 ? qsq_l[BM + r] : 0.f)) * 1.0001f;
    }
    bf16x8_t qf[RB][KS];
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

    if constexpr (RB == 1 && PIPE1 && DHP <= 96) {
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
        // tile j+1 has landed, everyone is past B(j-1): its stage takes tile j+2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!(ABL & 256)) __builtin_amdgcn_s_barrier();
        if (!(ABL & 32) && j + 2 < n_tiles) dma_stage<DHP, RB>(ring, (j + 2) % NSTAGE, kvimg + (long)(j + 2) * S::STAGE, wave, lane);
        const char* kf = ring + ((j + 1) % NSTAGE) * S::STAGE;          // K'(j+1)
        const uint32_t vbase = lds_addr(ring + (j % NSTAGE) * S::STAGE + S::IMG);   // V'(j)
        uint32_t kn_bits;
        {
            const float* kn_ptr = p.kn + __builtin_amdgcn_readfirstlane((b * p.H + h) * n_tiles + j);
            asm volatile("s_load_dword %0, %1, 0x0" : "=s"(kn_bits) : "s"(kn_ptr) : "memory");
        }
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
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(kn_bits));
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
                    if constexpr (SAMEMAP) out_item(GTA_DESC(c), c, qcs[it]);
                    else out_item(GTA_DESC(c), c, ocs[it]);
                }
            }
        };
        if (parE) out_items(std::integral_constant<int, 1>{}); else out_items(std::integral_constant<int, 0>{});
    }
    GTA_STAMP(4);
#undef GTA_STAMP
}

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

// exp2 + row-sum of one 16-value unit (in place: S' -> P)
GTA_DEV void exp_unit(f32x16_t& s, float& l0, float& l1) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        s[r] = __builtin_amdgcn_exp2f(s[r]);
        s[r + 1] = __builtin_amdgcn_exp2f(s[r + 1]);
        l0 += s[r];
        l1 += s[r + 1];
    }
}
GTA_DEV void pack_unit(const f32x16_t& s, bf16x8_t (&pf)[2]) {
    u32x4_t ww;
    ww.x = pack_bf16x2(s[0], s[1]); ww.y = pack_bf16x2(s[2], s[3]);
    ww.z = pack_bf16x2(s[4], s[5]); ww.w = pack_bf16x2(s[6], s[7]);
    pf[0] = __builtin_bit_cast(bf16x8_t, ww);
    ww.x = pack_bf16x2(s[8], s[9]); ww.y = pack_bf16x2(s[10], s[11]);
    ww.z = pack_bf16x2(s[12], s[13]); ww.w = pack_bf16x2(s[14], s[15]);
    pf[1] = __builtin_bit_cast(bf16x8_t, ww);
}
GTA_DEV float max_unit(const f32x16_t& s) {
    float mx = fmaxf(s[0], s[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
    return mx;
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
    hipLaunchKernelGGL((gta_fwd2_kernel<DHP, ESZ, RB, LAYOUT>), dim3((unsigned)n_wg), dim3(256), S::total(p.vrep_q ? p.Nq : 0),
                       stream, p);
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
template <int DHP, int ESZ, int RB>
static int launch_pipe_rb(const GtaFwdParams& p, hipStream_t stream) {
    switch (layout_of(p, DHP)) {
        case GTA_LAYOUT_MS:  if (DHP == 96) return launch_fwd3<DHP, ESZ, RB, (DHP == 96 ? GTA_LAYOUT_MS : GTA_LAYOUT_GENERIC)>(p, stream); break;
        case GTA_LAYOUT_CL:  if (DHP == 64) return launch_fwd3<DHP, ESZ, RB, (DHP == 64 ? GTA_LAYOUT_CL : GTA_LAYOUT_GENERIC)>(p, stream); break;
        case GTA_LAYOUT_SO2: return launch_fwd3<DHP, ESZ, RB, GTA_LAYOUT_SO2>(p, stream);
    }
    return launch_fwd3<DHP, ESZ, RB, GTA_LAYOUT_GENERIC>(p, stream);
}
template <int DHP, int ESZ>
static int launch_flash(const GtaFwdParams& p, int rb, hipStream_t stream) {
    if (rb == 2) {
        if constexpr (DHP == 64 || DHP == 96) return launch_pipe_rb<DHP, ESZ, 2>(p, stream);
        else return launch_flash_rb<DHP, ESZ, 2>(p, stream);
    }
    return launch_flash_rb<DHP, ESZ, 1>(p, stream);
}

int gta_fwd2_dispatch(GtaFwdParams& p, int dhp, int esz, bool run_prep, bool run_flash, int nw, hipStream_t stream) {
    const int rb = nw == 8 ? 2 : 1;                    // GTA_FLAG_WG8 = 256-row workgroups
    p.n_qtiles = (p.Tq + 128 * rb - 1) / (128 * rb);
    int rc = GTA_OK;
#define GTA_CASE2(D)                                                                    \
    case D:                                                                             \
        if (run_prep) rc = (esz == 2) ? launch_prep<D, 2>(p, stream) : launch_prep<D, 4>(p, stream); \
        if (rc == GTA_OK && run_flash) rc = (esz == 2) ? launch_flash<D, 2>(p, rb, stream) : launch_flash<D, 4>(p, rb, stream); \
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
