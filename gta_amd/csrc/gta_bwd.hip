// gta_bwd.hip -- backward of GTA attention for gfx950.
//
// The reference has no hand-written backward: autograd differentiates gta.py:92-279 and
// layers.py:202-211, saving q', k', v' and the dense [B,H,Tq,Tk] attention matrix per layer.
// Here, with A_q = (E_q.m)^T, B_k = inv(E_k).m, C_q = E_q.m (and D, R for so3/so2):
//     q' = A_q q      k' = B_k k      v' = B_k v      o~ = softmax(c1 q' k'^T) v'      o = C_q o~
//   do~ = C_q^T do  = A_q do          (C_q^T == A_q: same matrices as the Q transform)
//   D   = rowsum(do~ . o~) = rowsum(do . o)
//   dS  = P . (dP - D),  dP = do~ v'^T
//   dq' = c1 dS k'      dk' = c1 dS^T q'      dv' = P^T do~
//   dq  = A_q^T dq'     dk  = B_k^T dk'       dv  = B_k^T dv'
//   d trans_coeff = sum over tokens / se3 blocks of the entries of A_q, B_k, C_q that carry c
//                   (gta.py:40-44): dq'_3 (t_E . q_{0:3}) + (dk'_{0:3} . t_Einv) k_3 + (dv'.t_Einv) v_3
//                   + (do_{0:3} . t_E) o_3,  t_M = M[0:3,3]
// No gradient flows into the so3/so2 reps or the poses (gta.py:194-198 detach; data).
//
// Kernels (all tiles are the same rotation-swizzled bf16 "tile images" the forward streams):
//   gta_bwd_prep_kernel  per 64-query tile: raw q, do, o -> LDS (LDS-DMA); rho on q (prescaled by
//                        c1*log2e) and on do; writes Q''/dO~ images + [-lse*log2e | -D] per row
//   gta_bwd_dq_kernel    128 query rows / workgroup, loops over K'/V' images (forward workspace):
//                        S^T, dP^T = V' dO~^T, dS^T, dQ'^T += K'^T dS^T (K'^T by transpose-read);
//                        epilogue applies A_q^T per chunk and stores dq
//   gta_bwd_dkv_kernel   128 keys / workgroup (K'/V' fragments in VGPRs), loops over Q''/dO~ images:
//                        S, P, dP, dS; dV'^T += dO~^T P, dK'^T += Q''^T dS (transpose-reads);
//                        epilogue applies B_k^T per chunk and stores dk, dv
//   gta_reduce_kernel    deterministic sum of the per-workgroup d trans_coeff partials
#include "gta_common.h"
#include "gta_fwd_params.h"
#include "gta_bwd_params.h"
// The generated statements write M0 (LDS-DMA bases) and name it in their clobber lists; M0 is a reserved register for LLVM, which
// warns about that by default (the clobber is what keeps a hoisted M0 initialisation of compiler-emitted code from living across them):
// GTA_ASM_M0_BEGIN / _END silence -Winline-asm around those statements only -- every hand-written asm of this file keeps its diagnostics.
#define GTA_ASM_M0_BEGIN _Pragma("clang diagnostic push") _Pragma("clang diagnostic ignored \"-Winline-asm\"")
#define GTA_ASM_M0_END _Pragma("clang diagnostic pop")
#include "../../include/gta_hip.h"

namespace {

template <int N, class F>
GTA_DEV void static_for_bwd(F&& f) { gta_static_for<N>(static_cast<F&&>(f)); }

constexpr int BN = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

template <int IMM>
GTA_DEV u32x2_t lds_tr16_b64(uint32_t addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(IMM));
    return v;
}
// transpose-read at register address + constant: the constant rides in the instruction's 16-bit offset field; beyond it, in a second
// address register 32 KiB further on (loop-invariant: the compiler keeps it)
template <int OFF>
GTA_DEV u32x2_t lds_tr16_at(uint32_t addr) {
    if constexpr (OFF < 65536) return lds_tr16_b64<OFF>(addr);
    else return lds_tr16_b64<OFF - 32768>(addr + 32768u);
}
GTA_DEV uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// backward per-view records (floats).  q side:
//   [0:16) A = (E.m)^T   [16:32) C = E.m   [32:44) D1  [44:84) D2  [84:96) D1^T  [96:136) D2^T
//   [136:140) t_E = E[0:3,3] (unmasked), pad
// k side:
//   [0:16) B = inv(E).m  [16:32) B^T       [32:44) D1  [44:84) D2  [84:96) D1^T  [96:136) D2^T
//   [136:140) t_Einv = inv(E)[0:3,3], pad
#define BREC 140
#define BREC_M 0
#define BREC_MT 16
#define BREC_D1 32
#define BREC_D2 44
#define BREC_D1T 84
#define BREC_D2T 96
#define BREC_T 136

// side 0 = q (A, C from the "inv" slot), side 1 = k (B, B^T from the "rep" slot)
GTA_DEV void stage_brec(float* rec, const float* vrep, long view0, int cnt, int side, float tc, int tid, int nthreads) {
    for (int i = tid; i < cnt * BREC; i += nthreads) {
        const int n = i / BREC, e = i - n * BREC;
        const float* src = vrep + (view0 + n) * GTA_VREP_STRIDE;
        const int slot = side == 0 ? GTA_VREP_INV : GTA_VREP_REP;
        float val = 0.f;
        if (e < 32) {
            const int ee = e & 15, r = ee >> 2, c = ee & 3;
            // q side: first = (E.m)^T, second = E.m ; k side: first = M.m, second = (M.m)^T
            const bool transpose = (side == 0) ? (e < 16) : (e >= 16);
            const int sr = transpose ? c : r, sc = transpose ? r : c;
            const float m = (sr == 3) ? (sc == 3 ? 1.f : 0.f) : (sc == 3 ? tc : 1.f);
            val = src[slot + sr * 4 + sc] * m;
        } else if (e < BREC_D2) {
            const int ee = e - BREC_D1, r = ee >> 2, c = ee & 3;
            val = c < 3 ? src[GTA_VREP_D1 + r * 3 + c] : 0.f;
        } else if (e < BREC_D1T) {
            const int ee = e - BREC_D2, r = ee >> 3, c = ee & 7;
            val = c < 5 ? src[GTA_VREP_D2 + r * 5 + c] : 0.f;
        } else if (e < BREC_D2T) {
            const int ee = e - BREC_D1T, r = ee >> 2, c = ee & 3;
            val = c < 3 ? src[GTA_VREP_D1 + c * 3 + r] : 0.f;
        } else if (e < BREC_T) {
            const int ee = e - BREC_D2T, r = ee >> 3, c = ee & 7;
            val = c < 5 ? src[GTA_VREP_D2 + c * 5 + r] : 0.f;
        } else {
            const int r = e - BREC_T;
            val = r < 3 ? src[slot + r * 4 + 3] : 0.f;
        }
        rec[i] = val;
    }
}

// raw rows of one 64-row tile -> LDS by LDS-DMA, rotation-swizzled (as the forward pre-pass)
template <int U>
GTA_DEV void dma_raw_tile(char* dst, const char* gbase, long row_stride_bytes, int row0, int n_rows_total,
                          int real_units, int wave, int lane) {
    constexpr int NI = BN * U / 256;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int u0 = (wave * NI + i) * 64, u = u0 + lane;
        const int r = u / U, pos = u - r * U;
        const int rot = swz_rot<U>(r);
        int gu = pos - rot;
        gu = gu < 0 ? gu + U : gu;
        gu = gu < real_units ? gu : real_units - 1;
        int gr = row0 + r;
        gr = gr < n_rows_total ? gr : n_rows_total - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + (long)gr * row_stride_bytes + gu * 16),
                                         (__attribute__((address_space(3))) void*)(dst + u0 * 16), 16, 0, 0);
    }
}
template <int U, int ESZ>
GTA_DEV void raw_chunk(const char* raw, int r, int c, float* x) {
    if (ESZ == 2) {
        unpack8(*reinterpret_cast<const u32x4_t*>(raw + (r * U + swz<U>(r, c)) * 16), x);
    } else {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(raw + (r * U + swz<U>(r, 2 * c)) * 16);
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(raw + (r * U + swz<U>(r, 2 * c + 1)) * 16);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    }
}
template <int ESZ>
GTA_DEV void g_load_chunk(const char* rowptr, int c, float* x) {
    if (ESZ == 2) {
        unpack8(*reinterpret_cast<const u32x4_t*>(rowptr + c * 16), x);
    } else {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(rowptr + c * 32);
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(rowptr + c * 32 + 16);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    }
}
template <int ESZ>
GTA_DEV void g_store_chunk(char* rowptr, int c, const float* x) {
    if (ESZ == 2) {
        *reinterpret_cast<u32x4_t*>(rowptr + c * 16) = pack8(x);
    } else {
        *reinterpret_cast<f32x4_t*>(rowptr + c * 32) = f32x4_t{x[0], x[1], x[2], x[3]};
        *reinterpret_cast<f32x4_t*>(rowptr + c * 32 + 16) = f32x4_t{x[4], x[5], x[6], x[7]};
    }
}
GTA_DEV u32x4_t pack_acc8(const f32x16_t& a, int t) {   // accumulator registers 8t..8t+7 -> bf16x8
    u32x4_t w;
    w.x = pack_bf16x2(a[8 * t + 0], a[8 * t + 1]); w.y = pack_bf16x2(a[8 * t + 2], a[8 * t + 3]);
    w.z = pack_bf16x2(a[8 * t + 4], a[8 * t + 5]); w.w = pack_bf16x2(a[8 * t + 6], a[8 * t + 7]);
    return w;
}
// workgroup sum of one float per thread (256 threads) -> returned to thread 0
GTA_DEV float wg_sum256(float v, float* scratch /*>= 4 floats LDS*/, int tid) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// ================================================================================================
// 1. q-side pre-pass
// ================================================================================================
// X3 = the fp32-faithful instances (GTA_FLAG_FP32_PRODUCTS, fp32 inputs, dh <= 64; r06): a tile is FOUR images [Q''hi | dO~hi | Q''lo | dO~lo]
// (hi = bf16(x), lo = bf16(x - hi): x = hi + lo to 2^-17), as the forward's [K'hi | V'hi | K'lo | V'lo] of gta_prep.hip
template <int DHP, int ESZ, bool X3 = false>
struct BPrepSmem {
    static_assert(!X3 || ESZ == 4, "the split images are built from fp32 inputs");
    static constexpr int U = DHP * ESZ / 16;
    static constexpr int RAW = BN * DHP * ESZ;
    static constexpr int IMG = BN * DHP * 2;
    static constexpr int NIMG = X3 ? 4 : 2;
    static constexpr int OFF_RQ = 0;
    static constexpr int OFF_RDO = OFF_RQ + RAW;
    static constexpr int OFF_RO = OFF_RDO + RAW;
    static constexpr int OFF_IQ = (ESZ == 2) ? OFF_RQ : OFF_RO + RAW;      // images in place for bf16 input
    static constexpr int OFF_IDO = (ESZ == 2) ? OFF_RDO : OFF_IQ + IMG;
    static constexpr int OFF_IQL = OFF_IDO + IMG;                            // (X3) the lo images, in the tile's order
    static constexpr int OFF_IDOL = OFF_IQL + IMG;
    static constexpr int OFF_D = (ESZ == 2) ? OFF_RO + RAW : OFF_IDO + IMG + (X3 ? 2 * IMG : 0);  // D partials [4 waves][64 rows] + 4 scratch floats
    static constexpr int OFF_REC = OFF_D + 4 * 64 * 4 + 16;                  // records last, sized by the actual number of views
    static int total(int nviews) { return OFF_REC + nviews * BREC * 4; }
};

template <int DHP, int ESZ, bool X3 = false>
__global__ __launch_bounds__(256) void gta_bwd_prep_kernel(const GtaBwdParams p) {
    using S = BPrepSmem<DHP, ESZ, X3>;
    constexpr int CHP = DHP / 8, U = S::U, IMG = S::IMG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // work map as in gta_kv_prep_kernel: the H workgroups of one 64-row tile are consecutive on ONE XCD (L % 8), whose L2
    // then serves the 128-B lines that neighbouring heads of the packed q / dout / out rows share
    const int n_qt = (p.Tq + BN - 1) / BN;
    int j, h, b;
    {
        const int L = blockIdx.x, x = L & 7, i = L >> 3;
        const int r = x + 8 * (i / p.H);
        h = i - (i / p.H) * p.H;
        if (r >= p.B * n_qt) return;
        b = r / n_qt;
        j = r - b * n_qt;
    }
    const int ch_real = p.dh >> 3, real_units = p.dh * ESZ / 16;

    const char* qg = (const char*)p.q + ((long)b * p.q_sb + (long)h * p.q_sh) * ESZ;
    const char* dog = (const char*)p.dout + ((long)b * p.do_sb + (long)h * p.do_sh) * ESZ;
    const char* og = (const char*)p.out + ((long)b * p.o_sb + (long)h * p.o_sh) * ESZ;
    dma_raw_tile<U>(smem + S::OFF_RQ, qg, p.q_st * ESZ, j * BN, p.Tq, real_units, wave, lane);
    dma_raw_tile<U>(smem + S::OFF_RDO, dog, p.do_st * ESZ, j * BN, p.Tq, real_units, wave, lane);
    dma_raw_tile<U>(smem + S::OFF_RO, og, p.o_st * ESZ, j * BN, p.Tq, real_units, wave, lane);

    float* rec = reinterpret_cast<float*>(smem + S::OFF_REC);
    float* dsum = reinterpret_cast<float*>(smem + S::OFF_D);
    const float tc = p.trans_coeff ? *p.trans_coeff : 1.0f;
    if (p.vrep_q) stage_brec(rec, p.vrep_q, (long)b * p.Nq, p.Nq, 0, tc, tid, 256);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const float qscale = p.scale * LOG2E / (p.tau ? *p.tau : 1.0f);
    const int r = lane;
    const int t_raw = j * BN + r;
    const bool valid = t_raw < p.Tq;
    const int t = valid ? t_raw : p.Tq - 1;
    const int n = view_of(t, p.Pq, p.invPq);
    const float* rc = rec + n * BREC;
    float dpart = 0.f, dcpart = 0.f;
    const float* cs_base = p.cs_q ? p.cs_q + ((long)b * p.Tq + t) * 2 * p.nso2 : nullptr;
    // One chunk of q, dout and out per wave and iteration, lane == query row.  As in gta_kv_prep_kernel: the chunk's kind
    // is wave-uniform and every kind has its own load -> rho -> pack -> store body (no register arrays merged behind branches).
    auto chunks = [&](auto XO) {
        constexpr bool xo = decltype(XO)::value;
#pragma unroll
        for (int it = 0; it < CHP / 4; ++it) {
            const int c = wave + 4 * it;
            const int off = (r * CHP + swz<CHP>(r, c)) * 16;
            auto run = [&](auto SE3LO, auto SE3HI, auto&& apply) {
                float x[2][8], o8[8];
                raw_chunk<U, ESZ>(smem + S::OFF_RQ, r, c, x[0]);
                raw_chunk<U, ESZ>(smem + S::OFF_RDO, r, c, x[1]);
                raw_chunk<U, ESZ>(smem + S::OFF_RO, r, c, o8);
#pragma unroll
                for (int i = 0; i < 8; ++i) dpart += x[1][i] * o8[i];
                if constexpr (xo && (decltype(SE3LO)::value || decltype(SE3HI)::value)) {   // d trans_coeff through C_q = E.m (output rep)
                    const float t0 = rc[BREC_T], t1 = rc[BREC_T + 1], t2 = rc[BREC_T + 2];
                    if constexpr (decltype(SE3LO)::value) dcpart += (x[1][0] * t0 + x[1][1] * t1 + x[1][2] * t2) * o8[3];
                    if constexpr (decltype(SE3HI)::value) dcpart += (x[1][4] * t0 + x[1][5] * t1 + x[1][6] * t2) * o8[7];
                }
                apply(x);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[0][i] *= qscale;
                const u32x4_t hq = pack8(x[0]), hd = pack8(x[1]);
                *reinterpret_cast<u32x4_t*>(smem + S::OFF_IQ + off) = hq;
                *reinterpret_cast<u32x4_t*>(smem + S::OFF_IDO + off) = hd;
                if constexpr (X3) {
                    float tq[8], td[8];
                    unpack8(hq, tq); unpack8(hd, td);
#pragma unroll
                    for (int i = 0; i < 8; ++i) { x[0][i] -= tq[i]; x[1][i] -= td[i]; }
                    *reinterpret_cast<u32x4_t*>(smem + S::OFF_IQL + off) = pack8(x[0]);
                    *reinterpret_cast<u32x4_t*>(smem + S::OFF_IDOL + off) = pack8(x[1]);
                }
            };
            constexpr std::true_type T{};
            constexpr std::false_type F{};
            if (c < ch_real && valid) {
                const uint32_t desc = p.ctab[c];
                const uint32_t lo = cd_lo(desc), hi = cd_hi(desc);
                if (desc == 0) {
                    run(F, F, [](float (*)[8]) {});
                } else if (desc & GTA_CHUNK_SO3) {
                    run(F, F, [&](float (*x)[8]) {
                        float M1[12], M2[40];
                        lds_loadN4<3>(rc + BREC_D1, M1);
                        lds_loadN4<10>(rc + BREC_D2, M2);
                        mat3_apply_p4(M1, x[0]); mat5_apply_p8(M2, x[0] + 3);
                        if constexpr (xo) { mat3_apply_p4(M1, x[1]); mat5_apply_p8(M2, x[1] + 3); }
                    });
                } else if (lo == GTA_HALF_SE3 && hi == GTA_HALF_SE3) {
                    run(T, T, [&](float (*x)[8]) {
                        float M[16];
                        lds_load16(rc + BREC_M, M);
                        mat4_apply(M, x[0]); mat4_apply(M, x[0] + 4);
                        if constexpr (xo) { mat4_apply(M, x[1]); mat4_apply(M, x[1] + 4); }
                    });
                } else if (lo == GTA_HALF_SO2 && hi == GTA_HALF_SO2 && cs_base) {
                    run(F, F, [&](float (*x)[8]) {
                        const f32x4_t t0 = *reinterpret_cast<const f32x4_t*>(cs_base + 2 * cd_so2_lo(desc));
                        const f32x4_t t1 = *reinterpret_cast<const f32x4_t*>(cs_base + 2 * cd_so2_hi(desc));
                        rot2_apply<false>(t0.x, t0.y, x[0]); rot2_apply<false>(t0.z, t0.w, x[0] + 2);
                        rot2_apply<false>(t1.x, t1.y, x[0] + 4); rot2_apply<false>(t1.z, t1.w, x[0] + 6);
                        if constexpr (xo) {
                            rot2_apply<false>(t0.x, t0.y, x[1]); rot2_apply<false>(t0.z, t0.w, x[1] + 2);
                            rot2_apply<false>(t1.x, t1.y, x[1] + 4); rot2_apply<false>(t1.z, t1.w, x[1] + 6);
                        }
                    });
                } else {                                 // mixed halves (no shipped config): the generic body
                    auto generic = [&](float (*x)[8]) {
                        f32x2_t cs[4];
                        if (cs_base) load_cs(desc, cs_base, cs);
                        chunk_apply<false, xo ? 2 : 1>(desc, rc + BREC_M, rc + BREC_D1, rc + BREC_D2, cs, x);   // q and do: same A_q
                    };
                    if (lo == GTA_HALF_SE3) run(T, F, generic);
                    else if (hi == GTA_HALF_SE3) run(F, T, generic);
                    else run(F, F, generic);
                }
            } else {
                const u32x4_t z = {0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4_t*>(smem + S::OFF_IQ + off) = z;
                *reinterpret_cast<u32x4_t*>(smem + S::OFF_IDO + off) = z;
                if constexpr (X3) {
                    *reinterpret_cast<u32x4_t*>(smem + S::OFF_IQL + off) = z;
                    *reinterpret_cast<u32x4_t*>(smem + S::OFF_IDOL + off) = z;
                }
            }
        }
    };
    if (p.flags & GTA_FLAG_V_TRANSFORM) chunks(std::true_type{}); else chunks(std::false_type{});
    dsum[wave * 64 + r] = dpart;       // (every wave's share of the row's <dout, out>: summed below in a fixed order, no atomics)
    const float dc_wg = wg_sum256(dcpart, dsum + 256, tid);     // (includes a __syncthreads)
    __syncthreads();
    const long tile = ((long)b * p.H + h) * n_qt + j;
    char* gimg = (char*)p.qimg + tile * ((long)S::NIMG * IMG);
    constexpr int PIECES = IMG / 1024;
    static_assert(S::OFF_IDO == S::OFF_IQ + IMG || ESZ == 2, "fp32 input: the images are contiguous in LDS, in the tile's order");
#pragma unroll
    for (int i = 0; i < (S::NIMG * PIECES + 3) / 4; ++i) {
        const int piece = wave + 4 * i;
        if (piece < S::NIMG * PIECES) {
            const int im = piece / PIECES, pc = piece - im * PIECES;
            const char* src = smem + (im == 0 ? S::OFF_IQ : im == 1 ? S::OFF_IDO : im == 2 ? S::OFF_IQL : S::OFF_IDOL) + pc * 1024 + lane * 16;
            *reinterpret_cast<u32x4_t*>(gimg + piece * 1024 + lane * 16) = *reinterpret_cast<const u32x4_t*>(src);
        }
    }
    // per-row statistics, NEGATED: [-lse * log2e | -D] -- the dQ and dK/dV kernels start their S and dP accumulators from them (the
    // MFMA's C operand), so S - lse and dP - D cost no instruction; rows past Tq get -lse = -big so that P == 0 there
    float* st = p.stats + tile * 128;
    if (tid < 64) {
        const int tt = j * BN + tid;
        st[tid] = tt < p.Tq ? -(p.lse[((long)b * p.H + h) * p.Tq + tt] * LOG2E) : -1e30f;
        st[64 + tid] = tt < p.Tq ? -((dsum[tid] + dsum[64 + tid]) + (dsum[128 + tid] + dsum[192 + tid])) : 0.f;
    }
    if (tid == 0) p.dc_partial[p.dc_off_prep + tile] = dc_wg;
}

// ================================================================================================
// 2. dQ
// ================================================================================================
constexpr int NSTAGE = 3;
template <int DHP>
struct DqSmem {
    static constexpr int CHP = DHP / 8;
    static constexpr int IMG = BN * DHP * 2;
    static constexpr int STAGE = 2 * IMG;
    static constexpr int RING = NSTAGE * STAGE;
    static constexpr int OROW = DHP + 4;
    static constexpr int OST = 128 * OROW * 4;
    static_assert(OST <= RING, "dQ staging must fit the ring");
    static constexpr int OFF_RING = 0;
    static constexpr int OFF_SCR = RING;                   // 8 floats of reduction scratch
    static constexpr int OFF_REC = RING + 32;
    static constexpr int total(int nviews) { return OFF_REC + nviews * BREC * 4; }
};

template <int BYTES>
GTA_DEV void dma_linear4(char* dst, const char* src, int wave, int lane) {     // 4 waves, 1 KiB pieces
    dma_linear_4waves<BYTES>(dst, src, wave, lane);      // scalar-base asm form (gta_common.h): no per-piece vector arithmetic
}

// one LDS-DMA operation with per-lane source offsets: LDS bytes [dst, dst + 64 N) <- N bytes at base + voff(lane) + IMM per lane (the
// instruction's immediate moves the LDS address too: M0 = dst - IMM)
template <int IMM>
GTA_DEV void dma_lanes_dword(uint32_t dst, uint32_t voff, const void* base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 offset:%3" ::"s"(dst - (uint32_t)IMM), "v"(voff), "s"(base), "n"(IMM) : "memory");
}
template <int IMM>
GTA_DEV void dma_lanes_x4(uint32_t dst, uint32_t voff, const void* base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" ::"s"(dst - (uint32_t)IMM), "v"(voff), "s"(base), "n"(IMM) : "memory");
}
// the 512 bytes of a query tile's statistics by LDS-DMA: one dword per lane, waves 0 / 1 carry the halves, waves 2 / 3 repeat them (same
// bytes to the same place: every wave issues the same number of vector-memory operations, which the counted waits rely on)
GTA_DEV void dma_stats(float* dst, const float* src, int wave, int lane) {
    const uint32_t lds = lds_addr(dst + 64 * (wave & 1));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds), "v"((unsigned)lane * 4u), "s"(src + 64 * (wave & 1)) : "memory");
}

template <int DHP, int ESZ>
// (body: workgroup L of nwg of this kernel's grid -- its own launch, or its share of the joint launch gta_bwd_dqkv_kernel)
GTA_DEV void bwd_dq_body(const GtaBwdParams& p, char* smem, const int L, const int nwg) {
    using S = DqSmem<DHP>;
    constexpr int CHP = S::CHP, KS = DHP / 16, DB = DHP / 32, BM = 128;
    constexpr int DMA_PER_WAVE = S::STAGE / 1024 / 4;
    constexpr int ITEMS = 2 * CHP / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    int w;
    {
        const int xcd = L & 7, idx = L >> 3, q8 = nwg >> 3, r8 = nwg & 7;
        w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int n_q128 = (p.Tq + BM - 1) / BM;
    const int bh = w / n_q128, qt = w - bh * n_q128;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * BM;
    const int n_tiles = (p.Tk + BN - 1) / BN;
    const int n_qt64 = (p.Tq + BN - 1) / BN;
    const int ch_real = p.dh >> 3;
    char* ring = smem + S::OFF_RING;
    float* rec = reinterpret_cast<float*>(smem + S::OFF_REC);
    const char* kvimg = (const char*)p.kvimg + ((long)b * p.H + h) * n_tiles * (long)S::STAGE;

    // Q''/dO~ images of this workgroup's two 64-row tiles -> ring stages 1,2 ; K'/V' tile 0 -> stage 0
    const long qtile0 = ((long)b * p.H + h) * n_qt64 + 2 * qt;
    const int n_my_qt = (2 * qt + 1 < n_qt64) ? 2 : 1;
    dma_linear4<S::STAGE>(ring, kvimg, wave, lane);
    dma_linear4<S::STAGE>(ring + S::STAGE, (const char*)p.qimg + qtile0 * S::STAGE, wave, lane);
    if (n_my_qt == 2) dma_linear4<S::STAGE>(ring + 2 * S::STAGE, (const char*)p.qimg + (qtile0 + 1) * S::STAGE, wave, lane);

    const int t_last = (q0 + BM - 1 < p.Tq ? q0 + BM - 1 : p.Tq - 1);
    const int n_first = q0 / p.Pq;
    const int n_cnt = t_last / p.Pq - n_first + 1;
    const float tc = p.trans_coeff ? *p.trans_coeff : 1.0f;
    if (p.vrep_q) stage_brec(rec, p.vrep_q, (long)b * p.Nq + n_first, n_cnt, 0, tc, tid, 256);
    // per-row statistics of my query row
    const int my_row = wave * 32 + l31;                 // 0..127 in the workgroup
    const int my_tile = my_row >> 6;
    float lse2n = -1e30f, Dn = 0.f;                      // (-lse * log2e, -D: gta_bwd_prep_kernel)
    if (my_tile < n_my_qt) {
        const float* st = p.stats + (qtile0 + my_tile) * 128;
        lse2n = st[my_row & 63];
        Dn = st[64 + (my_row & 63)];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    bf16x8_t qf[KS], dof[KS];
    {
        const char* qi = ring + (1 + my_tile) * S::STAGE;
        const int r = my_row & 63;
        if (my_tile < n_my_qt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (r * CHP + swz<CHP>(r, 2 * ks + lh)) * 16;
                qf[ks] = *reinterpret_cast<const bf16x8_t*>(qi + off);
                dof[ks] = *reinterpret_cast<const bf16x8_t*>(qi + S::IMG + off);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4_t z = {0, 0, 0, 0};
                qf[ks] = __builtin_bit_cast(bf16x8_t, z);
                dof[ks] = __builtin_bit_cast(bf16x8_t, z);
            }
        }
    }
    __syncthreads();
    if (n_tiles > 1) dma_linear4<S::STAGE>(ring + S::STAGE, kvimg + (long)S::STAGE, wave, lane);

    f32x16_t dq[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[d][i] = 0.f;

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

    // one key tile.  ST >= 0: the ring stage as a constant (the tile loop is unrolled by NSTAGE, so the fragment reads' stage offsets are
    // immediates instead of per-tile address arithmetic); ST < 0: taken from j.  TAIL: the variant that may hold the masked last tile
    auto tile_step = [&](int j, auto STC, auto TAILC) __attribute__((always_inline)) {
        constexpr int ST = decltype(STC)::value;
        constexpr bool TAIL = decltype(TAILC)::value;
        if (j + 1 < n_tiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
        else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (j + 2 < n_tiles) dma_linear4<S::STAGE>(ring + (ST < 0 ? (j + 2) % NSTAGE : (ST + 2) % NSTAGE) * S::STAGE, kvimg + (long)(j + 2) * S::STAGE, wave, lane);
        const char* kf = ring + (ST < 0 ? j % NSTAGE : ST) * S::STAGE;
        const char* vf = kf + S::IMG;

        f32x16_t s0, s1, e0, e1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = lse2n; s1[i] = lse2n; e0[i] = Dn; e1[i] = Dn; }      // accumulators start at -lse2, -D
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks]);
            const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(kf + koff[ks] + 32 * CHP * 16);
            const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(vf + koff[ks]);
            const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(vf + koff[ks] + 32 * CHP * 16);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qf[ks], s0, 0, 0, 0);     // S^T  = K' Q''^T
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, qf[ks], s1, 0, 0, 0);
            e0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, dof[ks], e0, 0, 0, 0);    // dP^T = V' dO~^T
            e1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, dof[ks], e1, 0, 0, 0);
        }
        // dQ'^T += K'^T dS^T below takes A = K'^T by transpose-reads of the K' image: channel block 0's are requested here, under
        // the softmax arithmetic; block d + 1's before the MFMAs of block d (two register sets, counted waits)
        const uint32_t kbase_l = lds_addr(kf);
        constexpr int SL = 16 * CHP * 16;
        u32x2_t klo[2][4], khi[2][4];
        auto tr_reads = [&](int d, int set) {
            const uint32_t a0 = kbase_l + voff[d][0], a1 = kbase_l + voff[d][1];
            klo[set][0] = lds_tr16_b64<0>(a0);      khi[set][0] = lds_tr16_b64<0>(a1);
            klo[set][1] = lds_tr16_b64<SL>(a0);     khi[set][1] = lds_tr16_b64<SL>(a1);
            klo[set][2] = lds_tr16_b64<2 * SL>(a0); khi[set][2] = lds_tr16_b64<2 * SL>(a1);
            klo[set][3] = lds_tr16_b64<3 * SL>(a0); khi[set][3] = lds_tr16_b64<3 * SL>(a1);
        };
        tr_reads(0, 0);
        // P = exp2(S - lse2);  dS = P (dP - D);  keys past Tk contribute nothing
        const bool tail = TAIL && (j == n_tiles - 1) && (p.Tk & (BN - 1));
        const int kbase = j * BN + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p0 = __builtin_amdgcn_exp2f(s0[r]), p1 = __builtin_amdgcn_exp2f(s1[r]);
            if (tail) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key >= p.Tk) p0 = 0.f;
                if (key + 32 >= p.Tk) p1 = 0.f;
            }
            s0[r] = p0 * e0[r];
            s1[r] = p1 * e1[r];
        }
        bf16x8_t dsf[2][2];
        dsf[0][0] = __builtin_bit_cast(bf16x8_t, pack_acc8(s0, 0)); dsf[0][1] = __builtin_bit_cast(bf16x8_t, pack_acc8(s0, 1));
        dsf[1][0] = __builtin_bit_cast(bf16x8_t, pack_acc8(s1, 0)); dsf[1][1] = __builtin_bit_cast(bf16x8_t, pack_acc8(s1, 1));

        static_for_bwd<DB>([&](auto DC) {
            constexpr int d = decltype(DC)::value, set = d & 1;
            if constexpr (d + 1 < DB) {
                tr_reads(d + 1, set ^ 1);
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const u32x4_t av = {klo[set][sl].x, klo[set][sl].y, khi[set][sl].x, khi[set][sl].y};
                dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), dsf[sl >> 1][sl & 1], dq[d], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        };
    {
        int j = 0;
        for (; j + NSTAGE < n_tiles; j += NSTAGE) {          // (never the last tile)
            tile_step(j, std::integral_constant<int, 0>{}, std::false_type{});
            tile_step(j + 1, std::integral_constant<int, 1>{}, std::false_type{});
            tile_step(j + 2, std::integral_constant<int, 2>{}, std::false_type{});
        }
#pragma unroll 1
        for (; j < n_tiles; ++j) tile_step(j, std::integral_constant<int, -1>{}, std::true_type{});
    }

    // ---- epilogue: dq = A_q^T (c1 dQ') ; d trans_coeff through A_q ----
    const float c1 = p.scale / (p.tau ? *p.tau : 1.0f);
    __syncthreads();
    float* ost = reinterpret_cast<float*>(smem + S::OFF_RING);
    {
        const int r = wave * 32 + l31;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t v = {dq[d][4 * g] * c1, dq[d][4 * g + 1] * c1, dq[d][4 * g + 2] * c1, dq[d][4 * g + 3] * c1};
                *reinterpret_cast<f32x4_t*>(ost + r * S::OROW + 32 * d + 8 * g + 4 * lh) = v;
            }
    }
    __syncthreads();
    const char* qg = (const char*)p.q + ((long)b * p.q_sb + (long)h * p.q_sh) * ESZ;
    char* dqg = (char*)p.dq + ((long)b * p.dq_sb + (long)h * p.dq_sh) * ESZ;
    float dcpart = 0.f, dtpart = 0.f;
    const bool want_dtau = p.dt_partial != nullptr;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int item = wave + 4 * it;
        const int c = item >> 1;
        const int r = lane + 64 * (item & 1);
        const int t = q0 + r;
        if (c < ch_real && t < p.Tq) {
            const uint32_t desc = p.ctab[c];
            float x[1][8];
            const f32x4_t a = *reinterpret_cast<const f32x4_t*>(ost + r * S::OROW + 8 * c);
            const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(ost + r * S::OROW + 8 * c + 4);
            x[0][0] = a.x; x[0][1] = a.y; x[0][2] = a.z; x[0][3] = a.w;
            x[0][4] = bb.x; x[0][5] = bb.y; x[0][6] = bb.z; x[0][7] = bb.w;
            if (desc) {
                const int n = view_of(t, p.Pq, p.invPq) - n_first;
                const float* rc = rec + n * BREC;
                if (!(desc & GTA_CHUNK_SO3) && (cd_lo(desc) == GTA_HALF_SE3 || cd_hi(desc) == GTA_HALF_SE3)) {
                    float q8[8];
                    g_load_chunk<ESZ>(qg + (long)t * p.q_st * ESZ, c, q8);
                    const float t0 = rc[BREC_T], t1 = rc[BREC_T + 1], t2 = rc[BREC_T + 2];
                    if (cd_lo(desc) == GTA_HALF_SE3) dcpart += x[0][3] * (t0 * q8[0] + t1 * q8[1] + t2 * q8[2]);
                    if (cd_hi(desc) == GTA_HALF_SE3) dcpart += x[0][7] * (t0 * q8[4] + t1 * q8[5] + t2 * q8[6]);
                }
                f32x2_t cs[4];
                if (p.cs_q) load_cs(desc, p.cs_q + ((long)b * p.Tq + t) * 2 * p.nso2, cs);
                // A_q^T = E.m (record slot MT), D^T, R^T
                chunk_apply<true, 1>(desc, rc + BREC_MT, rc + BREC_D1T, rc + BREC_D2T, cs, x);
            }
            g_store_chunk<ESZ>(dqg + (long)t * p.dq_st * ESZ, c, x[0]);
            if (want_dtau) {                  // d tau = -(1/tau) sum <q, dq>  (gta_hip.h)
                float q8[8];
                g_load_chunk<ESZ>(qg + (long)t * p.q_st * ESZ, c, q8);
#pragma unroll
                for (int i = 0; i < 8; ++i) dtpart += x[0][i] * q8[i];
            }
        }
    }
    const float dc_wg = wg_sum256(dcpart, reinterpret_cast<float*>(smem + S::OFF_SCR), tid);
    if (tid == 0) p.dc_partial[p.dc_off_dq + w] = dc_wg;
    if (want_dtau) {
        __syncthreads();
        const float dt_wg = wg_sum256(dtpart, reinterpret_cast<float*>(smem + S::OFF_SCR), tid);
        if (tid == 0) p.dt_partial[w] = dt_wg;
    }
}

// ================================================================================================
// 3. dK, dV
// ================================================================================================
template <int DHP>
struct DkvSmem {
    static constexpr int CHP = DHP / 8;
    static constexpr int IMG = BN * DHP * 2;
    static constexpr int STAGE = 2 * IMG;                  // Q'' image + dO~ image of one 64-row tile
    static constexpr int RING = NSTAGE * STAGE;
    // this workgroup's two K'/V' tiles are only needed until their fragments sit in VGPRs: they alias
    // ring stages 1..2, so the kernel needs just the ring and two workgroups share a CU
    static constexpr int OFF_RING = 0;
    static constexpr int OFF_KV = STAGE;
    static constexpr int OFF_STATS = OFF_RING + RING;      // [NSTAGE][128] floats: tile j's statistics in slot j % NSTAGE, like its images
    static constexpr int OFF_SCR = OFF_STATS + NSTAGE * 128 * 4;
    static constexpr int OFF_REC = OFF_SCR + 32;
    static constexpr int OROW = DHP + 4;
    static constexpr int OST = 128 * OROW * 4;             // staging of dK' (then dV'): 128 keys
    static_assert(OST <= RING, "staging must fit the ring");
    static constexpr int total(int nviews) { return OFF_REC + nviews * BREC * 4; }
};

template <int DHP, int ESZ>
GTA_DEV void bwd_dkv_body(const GtaBwdParams& p, char* smem, const int L, const int nwg) {
    using S = DkvSmem<DHP>;
    constexpr int CHP = S::CHP, KS = DHP / 16, DB = DHP / 32, BK = 128;
    constexpr int DMA_PER_WAVE = S::STAGE / 1024 / 4;
    constexpr int ITEMS = 2 * CHP / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    int w;
    {
        const int xcd = L & 7, idx = L >> 3, q8 = nwg >> 3, r8 = nwg & 7;
        w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int n_k128 = (p.Tk + BK - 1) / BK;
    const int bh = w / n_k128, kt = w - bh * n_k128;
    const int b = bh / p.H, h = bh - b * p.H;
    const int k0 = kt * BK;
    const int n_kt64 = (p.Tk + BN - 1) / BN;
    const int n_qt = (p.Tq + BN - 1) / BN;
    const int ch_real = p.dh >> 3;
    char* ring = smem + S::OFF_RING;
    float* stats = reinterpret_cast<float*>(smem + S::OFF_STATS);
    float* rec = reinterpret_cast<float*>(smem + S::OFF_REC);

    const long ktile0 = ((long)b * p.H + h) * n_kt64 + 2 * kt;
    const int n_my_kt = (2 * kt + 1 < n_kt64) ? 2 : 1;
    const char* qimg = (const char*)p.qimg + ((long)b * p.H + h) * n_qt * (long)S::STAGE;
    const float* gstats = p.stats + ((long)b * p.H + h) * n_qt * 128;
    dma_linear4<S::STAGE>(smem + S::OFF_KV, (const char*)p.kvimg + ktile0 * S::STAGE, wave, lane);
    if (n_my_kt == 2) dma_linear4<S::STAGE>(smem + S::OFF_KV + S::STAGE, (const char*)p.kvimg + (ktile0 + 1) * S::STAGE, wave, lane);
    dma_linear4<S::STAGE>(ring, qimg, wave, lane);
    dma_stats(stats, gstats, wave, lane);

    const int t_last = (k0 + BK - 1 < p.Tk ? k0 + BK - 1 : p.Tk - 1);
    const int n_first = k0 / p.Pk;
    const int n_cnt = t_last / p.Pk - n_first + 1;
    const float tc = p.trans_coeff ? *p.trans_coeff : 1.0f;
    if (p.vrep_k) stage_brec(rec, p.vrep_k, (long)b * p.Nk + n_first, n_cnt, 1, tc, tid, 256);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // this wave's 32 keys: K' and V' fragments (MFMA B operands) stay in VGPRs
    const int my_key = wave * 32 + l31;                 // 0..127
    const int my_kt = my_key >> 6;
    bf16x8_t kfr[KS], vfr[KS];
    {
        const char* ki = smem + S::OFF_KV + my_kt * S::STAGE;
        const int r = my_key & 63;
        if (my_kt < n_my_kt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (r * CHP + swz<CHP>(r, 2 * ks + lh)) * 16;
                kfr[ks] = *reinterpret_cast<const bf16x8_t*>(ki + off);
                vfr[ks] = *reinterpret_cast<const bf16x8_t*>(ki + S::IMG + off);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4_t z = {0, 0, 0, 0};
                kfr[ks] = __builtin_bit_cast(bf16x8_t, z);
                vfr[ks] = __builtin_bit_cast(bf16x8_t, z);
            }
        }
    }
    __syncthreads();      // every wave holds its K'/V' fragments: ring stages 1..2 are free
    if (n_qt > 1) {
        dma_linear4<S::STAGE>(ring + S::STAGE, qimg + (long)S::STAGE, wave, lane);
        dma_stats(stats + 128, gstats + 128, wave, lane);
    }

    f32x16_t dk[DB], dv[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[d][i] = 0.f; dv[d][i] = 0.f; }

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
    uint32_t vaddr[DB][2];                               // ... as LDS addresses of the ring's stage 0, Q'' image, row block 0
#pragma unroll
    for (int d = 0; d < DB; ++d) { vaddr[d][0] = lds_addr(ring) + voff[d][0]; vaddr[d][1] = lds_addr(ring) + voff[d][1]; }

    // one query tile.  ST >= 0: the ring stage as a constant (the tile loop is unrolled by NSTAGE: the fragment reads' stage offsets
    // are immediates instead of per-tile address arithmetic); ST < 0: taken from j
    auto tile_step = [&](int j, auto STC) __attribute__((always_inline)) {
        constexpr int ST = decltype(STC)::value;
        // tile j's images and statistics have landed when only tile j + 1's (requested one tile ago) are in flight.  No compiler-visible
        // load in the loop: one would be waited for with vmcnt(0) -- the compiler does not count the DMA -- and drain the request
        // made two tiles ahead after one
        if (j + 1 < n_qt) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE + 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (j + 2 < n_qt) {
            dma_linear4<S::STAGE>(ring + (ST < 0 ? (j + 2) % NSTAGE : (ST + 2) % NSTAGE) * S::STAGE, qimg + (long)(j + 2) * S::STAGE, wave, lane);
            dma_stats(stats + (ST < 0 ? (j + 2) % NSTAGE : (ST + 2) % NSTAGE) * 128, gstats + (long)(j + 2) * 128, wave, lane);
        }
        const char* qi = ring + (ST < 0 ? j % NSTAGE : ST) * S::STAGE;       // Q'' image
        const char* di = qi + S::IMG;                           // dO~ image
        const float* stj = stats + (ST < 0 ? j % NSTAGE : ST) * 128;
        constexpr int SL = 16 * CHP * 16;

        static_for_bwd<2>([&](auto QBC) {                       // 32 query rows at a time
            constexpr int qb = decltype(QBC)::value;
            // register r <-> query row 32qb + 8(r>>2) + 4lh + (r&3): the accumulators start at the rows' -lse2 and -D (float4 groups
            // of the statistics, the MFMA's C operand), so S - lse2 and dP - D cost no instruction
            f32x16_t s, e;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(stj + 32 * qb + 8 * g + 4 * lh);
                const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(stj + 64 + 32 * qb + 8 * g + 4 * lh);
                s[4 * g] = l4.x; s[4 * g + 1] = l4.y; s[4 * g + 2] = l4.z; s[4 * g + 3] = l4.w;
                e[4 * g] = d4.x; e[4 * g + 1] = d4.y; e[4 * g + 2] = d4.z; e[4 * g + 3] = d4.w;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(qi + koff[ks] + qb * 32 * CHP * 16);
                const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(di + koff[ks] + qb * 32 * CHP * 16);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, kfr[ks], s, 0, 0, 0);    // S - lse2  = Q'' K'^T - lse2
                e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, vfr[ks], e, 0, 0, 0);    // dP - D    = dO~ V'^T - D
            }
            f32x16_t ds;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = __builtin_amdgcn_exp2f(s[i]);
                s[i] = pv;
                ds[i] = pv * e[i];
            }
            bf16x8_t pf[2], dsf[2];
            pf[0] = __builtin_bit_cast(bf16x8_t, pack_acc8(s, 0));   pf[1] = __builtin_bit_cast(bf16x8_t, pack_acc8(s, 1));
            dsf[0] = __builtin_bit_cast(bf16x8_t, pack_acc8(ds, 0)); dsf[1] = __builtin_bit_cast(bf16x8_t, pack_acc8(ds, 1));
            // dV'^T += dO~^T P ; dK'^T += Q''^T dS   (A operands by transpose-read of the row-major images)
            // the transpose-reads of channel block d + 1 are requested before the MFMAs of block d (two register sets).  Addresses: the
            // lane's offset (+ the ring's LDS base) in ONE register per (d, half) for the whole loop; stage, image and row-block offsets
            // are immediates where the stage is a constant (one address register per (stage, image, block, d, half) spilled otherwise)
            constexpr int QOFF = (ST < 0 ? 0 : ST * S::STAGE) + qb * 32 * CHP * 16, DOFF = QOFF + S::IMG;
            const uint32_t rt = ST < 0 ? (uint32_t)((j % NSTAGE) * S::STAGE) : 0u;
            u32x2_t qlo[2][2], qhi[2][2], dlo[2][2], dhi[2][2];
            auto tr_reads = [&](auto DC2, int set) {
                constexpr int d2 = decltype(DC2)::value;
                const uint32_t a0 = vaddr[d2][0] + rt, a1 = vaddr[d2][1] + rt;
                qlo[set][0] = lds_tr16_at<QOFF>(a0);      qhi[set][0] = lds_tr16_at<QOFF>(a1);
                qlo[set][1] = lds_tr16_at<QOFF + SL>(a0); qhi[set][1] = lds_tr16_at<QOFF + SL>(a1);
                dlo[set][0] = lds_tr16_at<DOFF>(a0);      dhi[set][0] = lds_tr16_at<DOFF>(a1);
                dlo[set][1] = lds_tr16_at<DOFF + SL>(a0); dhi[set][1] = lds_tr16_at<DOFF + SL>(a1);
            };
            tr_reads(std::integral_constant<int, 0>{}, 0);
            static_for_bwd<DB>([&](auto DC) {
                constexpr int d = decltype(DC)::value, set = d & 1;
                if constexpr (d + 1 < DB) {
                    tr_reads(std::integral_constant<int, d + 1>{}, set ^ 1);
                    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const u32x4_t aq = {qlo[set][t].x, qlo[set][t].y, qhi[set][t].x, qhi[set][t].y};
                    const u32x4_t ad = {dlo[set][t].x, dlo[set][t].y, dhi[set][t].x, dhi[set][t].y};
                    dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ad), pf[t], dv[d], 0, 0, 0);
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, aq), dsf[t], dk[d], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        };
    {
        int j = 0;
        for (; j + NSTAGE <= n_qt; j += NSTAGE) {
            tile_step(j, std::integral_constant<int, 0>{});
            tile_step(j + 1, std::integral_constant<int, 1>{});
            tile_step(j + 2, std::integral_constant<int, 2>{});
        }
#pragma unroll 1
        for (; j < n_qt; ++j) tile_step(j, std::integral_constant<int, -1>{});
    }

    // ---- epilogue: dk = B_k^T (ln2 dK'), dv = B_k^T dV' ; d trans_coeff through B_k ----
    const bool xv = (p.flags & GTA_FLAG_V_TRANSFORM) != 0;
    const char* kg = (const char*)p.k + ((long)b * p.k_sb + (long)h * p.k_sh) * ESZ;
    const char* vg = (const char*)p.v + ((long)b * p.v_sb + (long)h * p.v_sh) * ESZ;
    char* dkg = (char*)p.dk + ((long)b * p.dk_sb + (long)h * p.dk_sh) * ESZ;
    char* dvg = (char*)p.dv + ((long)b * p.dv_sb + (long)h * p.dv_sh) * ESZ;
    float* ost = reinterpret_cast<float*>(smem + S::OFF_RING);
    float dcpart = 0.f;
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {            // 0: dK, 1: dV
        __syncthreads();
        {
            const int r = wave * 32 + l31;
            const float sc = which == 0 ? LN2 : 1.0f;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16_t& acc = which == 0 ? dk[d] : dv[d];
                    const f32x4_t v = {acc[4 * g] * sc, acc[4 * g + 1] * sc, acc[4 * g + 2] * sc, acc[4 * g + 3] * sc};
                    *reinterpret_cast<f32x4_t*>(ost + r * S::OROW + 32 * d + 8 * g + 4 * lh) = v;
                }
        }
        __syncthreads();
        const bool xf = which == 0 || xv;
        const char* rawg = which == 0 ? kg : vg;
        const long raw_st = (which == 0 ? p.k_st : p.v_st) * ESZ;
        char* outg = which == 0 ? dkg : dvg;
        const long out_st = (which == 0 ? p.dk_st : p.dv_st) * ESZ;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int item = wave + 4 * it;
            const int c = item >> 1;
            const int r = lane + 64 * (item & 1);
            const int t = k0 + r;
            if (c < ch_real && t < p.Tk) {
                const uint32_t desc = p.ctab[c];
                float x[1][8];
                const f32x4_t a = *reinterpret_cast<const f32x4_t*>(ost + r * S::OROW + 8 * c);
                const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(ost + r * S::OROW + 8 * c + 4);
                x[0][0] = a.x; x[0][1] = a.y; x[0][2] = a.z; x[0][3] = a.w;
                x[0][4] = bb.x; x[0][5] = bb.y; x[0][6] = bb.z; x[0][7] = bb.w;
                if (desc && xf) {
                    const int n = view_of(t, p.Pk, p.invPk) - n_first;
                    const float* rc = rec + n * BREC;
                    if (!(desc & GTA_CHUNK_SO3) && (cd_lo(desc) == GTA_HALF_SE3 || cd_hi(desc) == GTA_HALF_SE3)) {
                        float r8[8];
                        g_load_chunk<ESZ>(rawg + (long)t * raw_st, c, r8);
                        const float t0 = rc[BREC_T], t1 = rc[BREC_T + 1], t2 = rc[BREC_T + 2];
                        if (cd_lo(desc) == GTA_HALF_SE3) dcpart += (x[0][0] * t0 + x[0][1] * t1 + x[0][2] * t2) * r8[3];
                        if (cd_hi(desc) == GTA_HALF_SE3) dcpart += (x[0][4] * t0 + x[0][5] * t1 + x[0][6] * t2) * r8[7];
                    }
                    f32x2_t cs[4];
                    if (p.cs_k) load_cs(desc, p.cs_k + ((long)b * p.Tk + t) * 2 * p.nso2, cs);
                    chunk_apply<true, 1>(desc, rc + BREC_MT, rc + BREC_D1T, rc + BREC_D2T, cs, x);
                }
                g_store_chunk<ESZ>(outg + (long)t * out_st, c, x[0]);
            }
        }
    }
    const float dc_wg = wg_sum256(dcpart, reinterpret_cast<float*>(smem + S::OFF_SCR), tid);
    if (tid == 0) p.dc_partial[p.dc_off_dkv + w] = dc_wg;
}

// the two compiled kernels, and both in ONE launch (as gta_bwd_dqkv64_kernel for the generated pair, below: they depend on the q-side pre-pass
// only; the dK/dV blocks -- the longer ones -- first).  dh = 128: one workgroup per CU (the 96-KiB ring admits one anyway; with two, the
// 256-register budget left 117 dwords of scratch in the dK/dV tile loop)
template <int DHP, int ESZ>
__global__ __launch_bounds__(256, (DHP > 96 ? 1 : 2)) void gta_bwd_dq_kernel(const GtaBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bwd_dq_body<DHP, ESZ>(p, smem, blockIdx.x, gridDim.x);
}
template <int DHP, int ESZ>
__global__ __launch_bounds__(256, (DHP > 96 ? 1 : 2)) void gta_bwd_dkv_kernel(const GtaBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bwd_dkv_body<DHP, ESZ>(p, smem, blockIdx.x, gridDim.x);
}
template <int DHP, int ESZ>
__global__ __launch_bounds__(256, (DHP > 96 ? 1 : 2)) void gta_bwd_dqkv_kernel(const GtaBwdParams p, const int n_dq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int L = blockIdx.x, n_dkv = (int)gridDim.x - n_dq;
    if (L < n_dkv) bwd_dkv_body<DHP, ESZ>(p, smem, L, n_dkv);
    else bwd_dq_body<DHP, ESZ>(p, smem, L - n_dkv, n_dq);
}

// ------------------------------------------------------------------------------------------------
// 3b. dK, dV with 64 keys per wave, one wave per SIMD, the tile loop as ONE generated instruction stream (gen_bwd64.py ->
// gta_bwd64_dkv.inc; r04).  A workgroup owns 256 keys of one (b, h); its four waves keep the K' / V' fragments of 64 keys and the
// dK'^T / dV'^T accumulators in registers (all 256 AGPRs + 232 VGPRs are the stream's) and walk the (b, h)'s Q'' / dO~ tile images
// through a ring of four LDS stages.  Every streamed fragment feeds the wave's two 32-key blocks -- half the LDS bytes per MFMA of
// gta_bwd_dkv_kernel -- and each block's softmax runs in the gaps of the MFMAs of the two neighbouring segments (the generator's
// header has the schedule).  The compiler's part: operands (lane offsets of the image layout), the view records, and the epilogue
// (B_k^T per chunk, stores, d trans_coeff) on the accumulators the stream leaves in a[0:191].  bf16 images, dh = 96.
// ------------------------------------------------------------------------------------------------
#include "gta_bwd64_dkv.inc"
struct Dkv64Smem {
    static constexpr int STAGE = 2 * BN * 96 * 2;
    static constexpr int RING = GTA_BWD64_STAGES * STAGE;
    static constexpr int OFF_STATS = RING;
    // what the epilogue reads per key besides the accumulators, per wave: 24 x 64 dwords the stream leaves at its end (elements 3 and 7 of the
    // se3 chunks of its keys' K' / V' fragments = of the raw rows: d trans_coeff) and the six 16-byte pieces of the keys' 96-byte (cos, sin)
    // rows, fetched by LDS-DMA before the walk (their latency lies under it)
    static constexpr int OFF_SIDE = OFF_STATS + GTA_BWD64_STAGES * 512;
    static constexpr int SIDE_RAW = 24 * 256, SIDE_W = SIDE_RAW + 6 * 1024;
    static constexpr int OFF_SCR = OFF_SIDE + 4 * SIDE_W;
    static constexpr int OFF_REC = OFF_SCR + 32;
    static_assert(OFF_STATS == GTA_BWD64_OFF_STATS && GTA_BWD64_HI_BASE == 2 * STAGE, "gen_bwd64.py's LDS map");
    static constexpr int total(int nviews) { return OFF_REC + nviews * BREC * 4; }
};

// (body: workgroup L of nwg of this kernel's grid -- its own launch, or its share of the joint launch gta_bwd_dqkv64_kernel below)
template <int ESZ>
GTA_DEV void bwd_dkv64_body(const GtaBwdParams& p, char* smem, const int L, const int nwg) {
    using S = Dkv64Smem;
    constexpr int CHP = 12, DB = 3, KB = 2, BK = 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    int w;
    {
        const int xcd = L & 7, idx = L >> 3, q8 = nwg >> 3, r8 = nwg & 7;
        w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int n_kblk = (p.Tk + BK - 1) / BK;
    const int bh = w / n_kblk, kt = w - bh * n_kblk;
    const int b = bh / p.H, h = bh - b * p.H;
    const int k0 = kt * BK;
    const int n_kt64 = (p.Tk + BN - 1) / BN;
    const int n_qt = (p.Tq + BN - 1) / BN;
    char* ring = smem;
    float* rec = reinterpret_cast<float*>(smem + S::OFF_REC);

    const int t_last = (k0 + BK - 1 < p.Tk ? k0 + BK - 1 : p.Tk - 1);
    const int n_first = k0 / p.Pk;
    const int n_cnt = t_last / p.Pk - n_first + 1;
    const float tc = p.trans_coeff ? *p.trans_coeff : 1.0f;
    if (p.vrep_k) stage_brec(rec, p.vrep_k, (long)b * p.Nk + n_first, n_cnt, 1, tc, tid, 256);

    f32x16_t dk[KB][DB], dv[KB][DB];
    {
        // operands of the stream: the lane's offsets in a tile image (rows of 12 rotation-swizzled 16-byte units) as LDS addresses of ring
        // stage 0 (and of stage 2: immediates stay inside the 16-bit field); fragment rows: lane (row l31, unit 2 ks + lh) -- linear in ks
        // up to ks = 3, the rotation wraps for ks = 4, 5; transpose-reads: 4-row groups, linear over channel blocks 0, 1
        const uint32_t rb = lds_addr(ring);
        auto koff_of = [&](int ks) { return (uint32_t)((l31 * CHP + swz<CHP>(l31, 2 * ks + lh)) * 16) + rb; };
        const int g16 = lane >> 4, p16 = lane & 15;
        auto voff_of = [&](int d, int hf) {
            const int r = 4 * lh + (p16 >> 2) + 8 * hf, u = 4 * d + 2 * (g16 & 1) + ((p16 & 3) >> 1);
            return (uint32_t)((r * CHP + swz<CHP>(r, u)) * 16 + (p16 & 1) * 8) + rb;
        };
        const uint32_t koffl = koff_of(0), koff4 = koff_of(4), koff5 = koff_of(5);
        const uint32_t voff00 = voff_of(0, 0), voff01 = voff_of(0, 1), voff20 = voff_of(2, 0), voff21 = voff_of(2, 1);
        constexpr uint32_t HB = GTA_BWD64_HI_BASE;
        const uint32_t koffl_h = koffl + HB, koff4_h = koff4 + HB, koff5_h = koff5 + HB;
        const uint32_t voff00_h = voff00 + HB, voff01_h = voff01 + HB, voff20_h = voff20 + HB, voff21_h = voff21 + HB;
        const uint32_t lane16 = (uint32_t)lane * 16u, lane4 = (uint32_t)lane * 4u;
        const uint32_t stoff = rb + (uint32_t)S::OFF_STATS + 16u * (uint32_t)lh;
        const char* qimg = (const char*)p.qimg + ((long)b * p.H + h) * n_qt * (long)S::STAGE;
        const float* gstats = p.stats + ((long)b * p.H + h) * n_qt * 128;
        // this wave's 64 keys = 64-key tile 4 kt + wave (a tile past the end: the last one's images again -- nothing of it is stored)
        int my_tile = 4 * kt + wave;
        my_tile = my_tile < n_kt64 ? my_tile : n_kt64 - 1;
        const char* kv = (const char*)p.kvimg + (((long)b * p.H + h) * n_kt64 + my_tile) * (long)S::STAGE;
        const uint32_t q_lo = (uint32_t)(uintptr_t)qimg, q_hi = (uint32_t)((uintptr_t)qimg >> 32);
        const uint32_t st_lo = (uint32_t)(uintptr_t)gstats, st_hi = (uint32_t)((uintptr_t)gstats >> 32);
        const uint32_t kv_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)kv), kv_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)kv >> 32));
        const uint32_t n = (uint32_t)n_qt, ring_s = __builtin_amdgcn_readfirstlane(rb), stats_s = ring_s + (uint32_t)S::OFF_STATS;
        const uint32_t ring = ring_s, stats = stats_s;
        const uint32_t side = ring_s + (uint32_t)(S::OFF_SIDE + wave * S::SIDE_W);
        {
            // the (cos, sin) rows of this wave's 64 keys -> its side buffer, row-major [key][96 B] (the rows of consecutive tokens are contiguous:
            // six linear KiB; keys past Tk: clamped into the table): the epilogue reads them from LDS, their latency lies under the walk
            if (p.cs_k) {
                const float* csb = p.cs_k + (long)b * p.Tk * 2 * p.nso2;
                const long last = (long)p.Tk * 96 - 16;                       // (96 bytes per token: nso2 = 12)
                static_for_bwd<6>([&](auto PC) {
                    constexpr int pc = decltype(PC)::value;
                    long off = (long)(k0 + 64 * wave) * 96 + pc * 1024 + lane * 16;
                    off = off < last ? off : last;
                    dma_lanes_x4<0>(side + (uint32_t)(S::SIDE_RAW + pc * 1024), (uint32_t)off, csb);
                });
            }
        }
        // the accumulators are OUTPUTS of the statement, by register: dK'^T[kb][d] = a[16 (3 kb + d) ..], dV'^T[kb][d] = a[96 + 16 (3 kb + d) ..]
        // (GTA_BWD64_DKV_RESULTS, gen_bwd64.py) -- hipcc knows they live there and reads them out itself where the epilogue wants them
        GTA_ASM_M0_BEGIN
        asm volatile(GTA_BWD64_DKV : GTA_BWD64_DKV_RESULTS : GTA_BWD64_DKV_OPERANDS : GTA_BWD64_DKV_CLOBBERS);
        GTA_ASM_M0_END
    }

    // ---- epilogue: dk = B_k^T (ln2 dK'), dv = B_k^T dV' ; d trans_coeff through B_k.  No staging: lane (key l31, half lh) of a 32-key
    // block holds channels 32 d + 8 g + 4 lh + i of ITS key, one v_permlane32_swap per value hands lanes 0-31 the whole even chunk of a
    // pair (g, g + 1) and lanes 32-63 the odd one (as the forward's epilogue); every lane then transforms and stores whole 8-channel
    // chunks of its key -- no LDS round trip, no barrier, every load of the 24 chunks in flight at once (one workgroup per CU: there is
    // no second workgroup to hide a serial epilogue behind; with the staged form it was 28 % of the kernel) ----
    const bool xv = (p.flags & GTA_FLAG_V_TRANSFORM) != 0;
    char* dkg = (char*)p.dk + ((long)b * p.dk_sb + (long)h * p.dk_sh) * ESZ;
    char* dvg = (char*)p.dv + ((long)b * p.dv_sb + (long)h * p.dv_sh) * ESZ;
    float dcpart = 0.f;
    __syncthreads();                                     // (the view records: written before the stream, whose barriers ordered them too)
    const char* side = smem + S::OFF_SIDE + wave * S::SIDE_W;
#pragma unroll
    for (int pass = 0; pass < 2 * KB; ++pass) {          // (0: dK, 1: dV) x key block
        const int which = pass >> 1, kb = pass & 1;
        const float sc = which == 0 ? LN2 : 1.0f;
        const bool xf = which == 0 || xv;
        char* outg = which == 0 ? dkg : dvg;
        const long out_st = (which == 0 ? p.dk_st : p.dv_st) * ESZ;
        const int kl = 32 * kb + l31;                    // the lane's key of this pass, within the wave
        const int t = k0 + 64 * wave + kl;
        const bool valid = t < p.Tk;
        const float* rc = rec + (view_of(valid ? t : p.Tk - 1, p.Pk, p.invPk) - n_first) * BREC;
        // slot sl = the chunk pair (2 sl, 2 sl + 1) of the MSN layout: 0..2 se3 | se3, 3 so3 | so3, 4 so3 | so2, 5 so2 | so2; the low lane half
        // takes the even chunk, the high half the odd one
        static_for_bwd<2 * DB>([&](auto SC) {
            constexpr int sl = decltype(SC)::value, d = sl >> 1, gp = sl & 1;
            constexpr uint32_t DE = gta_layout_desc(GTA_LAYOUT_MS, 2 * sl), DO = gta_layout_desc(GTA_LAYOUT_MS, 2 * sl + 1);
            const f32x16_t& acc = which == 0 ? dk[kb][d] : dv[kb][d];
            float x[1][8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float fa = acc[8 * gp + i] * sc, fb = acc[8 * gp + 4 + i] * sc;       // half lh of the even / of the odd chunk
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(fa), __float_as_uint(fb), false, false);   // fa[32..63] <-> fb[0..31]
                x[0][i] = __uint_as_float(sw[0]);
                x[0][4 + i] = __uint_as_float(sw[1]);
            }
            const int c = 2 * sl + lh;
            if (xf) {
                if constexpr (sl < 3) {                  // se3 | se3: d trans_coeff through B_k (raw elements 3 and 7 of the chunk), then B_k^T
                    // (raw elements 3 and 7 of the lane's chunk = those of its K' / V' fragment, left by the stream: gen_bwd64.py)
                    const uint32_t w1 = *reinterpret_cast<const uint32_t*>(side + (((which * 2 + kb) * 3 + sl) * 2) * 256 + lane * 4);
                    const uint32_t w3 = *reinterpret_cast<const uint32_t*>(side + (((which * 2 + kb) * 3 + sl) * 2 + 1) * 256 + lane * 4);
                    const float r3 = ESZ == 2 ? bf16_hi(w1) : 0.f, r7 = ESZ == 2 ? bf16_hi(w3) : 0.f;
                    const float t0 = rc[BREC_T], t1 = rc[BREC_T + 1], t2 = rc[BREC_T + 2];
                    if (valid) {
                        dcpart += (x[0][0] * t0 + x[0][1] * t1 + x[0][2] * t2) * r3;
                        dcpart += (x[0][4] * t0 + x[0][5] * t1 + x[0][6] * t2) * r7;
                    }
                    chunk_apply<true, 1>(DE, rc + BREC_MT, rc + BREC_D1T, rc + BREC_D2T, nullptr, x);
                } else if constexpr (sl == 3) {          // so3 | so3
                    chunk_apply<true, 1>(DE, rc + BREC_MT, rc + BREC_D1T, rc + BREC_D2T, nullptr, x);
                } else {                                 // so2 chunks 9 (sl 4, high half), 10, 11: pieces 2 (c - 9), 2 (c - 9) + 1 of the key's (cos, sin) row
                    f32x2_t cs[4];
                    const int pc = 2 * (c - 9);
                    const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(side + S::SIDE_RAW + kl * 96 + (pc < 0 ? 0 : pc) * 16);
                    const f32x4_t c1 = *reinterpret_cast<const f32x4_t*>(side + S::SIDE_RAW + kl * 96 + (pc < 0 ? 1 : pc + 1) * 16);
                    cs[0] = f32x2_t{c0.x, c0.y}; cs[1] = f32x2_t{c0.z, c0.w}; cs[2] = f32x2_t{c1.x, c1.y}; cs[3] = f32x2_t{c1.z, c1.w};
                    if constexpr (sl == 4) {
                        if (lh == 0) chunk_apply<true, 1>(DE, rc + BREC_MT, rc + BREC_D1T, rc + BREC_D2T, cs, x);        // chunk 8: so3
                        else chunk_apply<true, 1>(DO, rc + BREC_MT, rc + BREC_D1T, rc + BREC_D2T, cs, x);                 // chunk 9: so2
                    } else {
                        chunk_apply<true, 1>(DE, rc + BREC_MT, rc + BREC_D1T, rc + BREC_D2T, cs, x);                      // (the halves differ in cs only)
                    }
                }
            }
            if (valid) g_store_chunk<ESZ>(outg + (long)t * out_st, c, x[0]);
        });
    }
    __syncthreads();
    const float dc_wg = wg_sum256(dcpart, reinterpret_cast<float*>(smem + S::OFF_SCR), tid);
    // the partials' slots are per 128 keys (gta_abi.cpp): this workgroup owns two of them
    if (tid == 0) {
        const int n_k128 = (p.Tk + 127) / 128;
        p.dc_partial[p.dc_off_dkv + bh * n_k128 + 2 * kt] = dc_wg;
        if (2 * kt + 1 < n_k128) p.dc_partial[p.dc_off_dkv + bh * n_k128 + 2 * kt + 1] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// 2b. dQ with 64 query rows per wave, one wave per SIMD, the walk over the key tiles ONE generated instruction stream (gen_bwd64.py:
// GenDQ -> gta_bwd64_dq.inc; r04).  A workgroup owns 256 query rows of one (b, h); its waves keep the Q'' / dO~ fragments of 64 rows and
// the dQ'^T accumulators in registers and stream the (b, h)'s K' / V' tile images through a ring of four LDS stages; every streamed
// fragment feeds the wave's two 32-row blocks.  Whole key tiles (Tk % 64 == 0), bf16, dh = 96, MSN layout, no d tau (the compiled kernel
// serves the rest).  Epilogue as gta_bwd_dkv64_kernel's: no staging, compile-time chunk descriptors, the rows' raw q chunks (d
// trans_coeff) and (cos, sin) rows fetched by LDS-DMA before the walk.
// ------------------------------------------------------------------------------------------------
#include "gta_bwd64_dq.inc"
struct Dq64Smem {
    static constexpr int STAGE = 2 * BN * 96 * 2;
    static constexpr int RING = GTA_BWD64_STAGES * STAGE;
    static constexpr int OFF_SIDE = RING;
    static constexpr int SIDE_RAW = 7 * 1024, SIDE_W = SIDE_RAW + 6 * 1024;      // per wave, row-major [row][96 B]: the se3 chunks of its 64 raw q rows, the rows' (cos, sin) rows
    static constexpr int OFF_SCR = OFF_SIDE + 4 * SIDE_W;
    static constexpr int OFF_REC = OFF_SCR + 32;
    static_assert(GTA_BWD64_HI_BASE == 2 * STAGE, "gen_bwd64.py's LDS map");
    static constexpr int total(int nviews) { return OFF_REC + nviews * BREC * 4; }
};

template <int ESZ>
GTA_DEV void bwd_dq64_body(const GtaBwdParams& p, char* smem, const int L, const int nwg) {
    using S = Dq64Smem;
    constexpr int CHP = 12, DB = 3, BM = 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    int w;
    {
        const int xcd = L & 7, idx = L >> 3, q8 = nwg >> 3, r8 = nwg & 7;
        w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int n_qblk = (p.Tq + BM - 1) / BM;
    const int bh = w / n_qblk, qt = w - bh * n_qblk;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * BM;
    const int n_kt64 = (p.Tk + BN - 1) / BN;
    const int n_qt64 = (p.Tq + BN - 1) / BN;
    char* ring = smem;
    float* rec = reinterpret_cast<float*>(smem + S::OFF_REC);

    const int t_last = (q0 + BM - 1 < p.Tq ? q0 + BM - 1 : p.Tq - 1);
    const int n_first = q0 / p.Pq;
    const int n_cnt = t_last / p.Pq - n_first + 1;
    const float tc = p.trans_coeff ? *p.trans_coeff : 1.0f;
    if (p.vrep_q) stage_brec(rec, p.vrep_q, (long)b * p.Nq + n_first, n_cnt, 0, tc, tid, 256);

    f32x16_t dq[2][DB];
    {
        const uint32_t rb = lds_addr(ring);
        auto koff_of = [&](int ks) { return (uint32_t)((l31 * CHP + swz<CHP>(l31, 2 * ks + lh)) * 16) + rb; };
        const int g16 = lane >> 4, p16 = lane & 15;
        auto voff_of = [&](int d, int hf) {
            const int r = 4 * lh + (p16 >> 2) + 8 * hf, u = 4 * d + 2 * (g16 & 1) + ((p16 & 3) >> 1);
            return (uint32_t)((r * CHP + swz<CHP>(r, u)) * 16 + (p16 & 1) * 8) + rb;
        };
        const uint32_t koffl = koff_of(0), koff4 = koff_of(4), koff5 = koff_of(5);
        const uint32_t voff00 = voff_of(0, 0), voff01 = voff_of(0, 1), voff20 = voff_of(2, 0), voff21 = voff_of(2, 1);
        constexpr uint32_t HB = GTA_BWD64_HI_BASE;
        const uint32_t koffl_h = koffl + HB, koff4_h = koff4 + HB, koff5_h = koff5 + HB;
        const uint32_t voff00_h = voff00 + HB, voff01_h = voff01 + HB, voff20_h = voff20 + HB, voff21_h = voff21 + HB;
        const uint32_t lane16 = (uint32_t)lane * 16u, lrow4 = (uint32_t)l31 * 4u;
        // this wave's 64 rows = 64-row tile 4 qt + wave of the (b, h) (a tile past the end: the last one's images again -- nothing of it is stored)
        int my_tile = 4 * qt + wave;
        my_tile = my_tile < n_qt64 ? my_tile : n_qt64 - 1;
        const char* qi = (const char*)p.qimg + (((long)b * p.H + h) * n_qt64 + my_tile) * (long)S::STAGE;
        const float* st = p.stats + (((long)b * p.H + h) * n_qt64 + my_tile) * 128;
        const char* kv = (const char*)p.kvimg + ((long)b * p.H + h) * n_kt64 * (long)S::STAGE;
        const uint32_t qi_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)qi), qi_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)qi >> 32));
        const uint32_t st_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)st), st_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)st >> 32));
        const uint32_t kv_lo = (uint32_t)(uintptr_t)kv, kv_hi = (uint32_t)((uintptr_t)kv >> 32);
        const uint32_t n = (uint32_t)n_kt64, ring_s = __builtin_amdgcn_readfirstlane(rb);
        const uint32_t ring = ring_s;
        {
            // the first 96 bytes (chunks 0..5, se3: d trans_coeff) of this wave's 64 raw q rows and the rows' (cos, sin) rows -> its side buffer,
            // both ROW-MAJOR [row][96 B]: a DMA operation's lanes walk (row, 16-byte piece) pairs, so a row's cache line is touched once (lane
            // l of raw operation i: row 10 i + l / 6, piece l % 6 -- lanes 60..63 already write the next operation's first pieces, the same
            // bytes; the (cos, sin) rows of consecutive tokens are contiguous: six linear KiB).  Rows past Tq: the last row's.  The epilogue
            // reads them from LDS, their latency lies under the walk
            const uint32_t side = ring_s + (uint32_t)(S::OFF_SIDE + wave * S::SIDE_W);
            const char* qgs = (const char*)p.q + ((long)b * p.q_sb + (long)h * p.q_sh) * ESZ;
            const int rowl = lane / 6, piece = lane - 6 * rowl;
            static_for_bwd<7>([&](auto OC) {
                constexpr int i = decltype(OC)::value;
                int tq = q0 + 64 * wave + 10 * i + rowl;
                tq = tq < p.Tq ? tq : p.Tq - 1;
                dma_lanes_x4<0>(side + (uint32_t)(i * 960), (uint32_t)((long)tq * p.q_st * ESZ) + (uint32_t)piece * 16u, qgs);
            });
            if (p.cs_q) {
                const float* csb = p.cs_q + (long)b * p.Tq * 2 * p.nso2;
                const long last = (long)p.Tq * 96 - 16;                       // (96 bytes per token: nso2 = 12)
                static_for_bwd<6>([&](auto PC) {
                    constexpr int pc = decltype(PC)::value;
                    long off = (long)(q0 + 64 * wave) * 96 + pc * 1024 + lane * 16;
                    off = off < last ? off : last;
                    dma_lanes_x4<0>(side + (uint32_t)(S::SIDE_RAW + pc * 1024), (uint32_t)off, csb);
                });
            }
        }
        // (the accumulators are outputs of the statement: dQ'^T[rb][d] = a[16 (3 rb + d) ..], GTA_BWD64_DQ_RESULTS)
        GTA_ASM_M0_BEGIN
        asm volatile(GTA_BWD64_DQ : GTA_BWD64_DQ_RESULTS : GTA_BWD64_DQ_OPERANDS : GTA_BWD64_DQ_CLOBBERS);
        GTA_ASM_M0_END
    }

    // ---- epilogue: dq = A_q^T (c1 dQ') ; d trans_coeff through A_q.  As gta_bwd_dkv64_kernel's: lane (row l31, half lh) of a 32-row block holds
    // channels 32 d + 8 g + 4 lh + i of ITS row, one v_permlane32_swap per value hands the low lane half the even chunk of a pair, the high half the odd one ----
    const float c1 = p.scale / (p.tau ? *p.tau : 1.0f);
    char* dqg = (char*)p.dq + ((long)b * p.dq_sb + (long)h * p.dq_sh) * ESZ;
    float dcpart = 0.f;
    __syncthreads();
    const char* side = smem + S::OFF_SIDE + wave * S::SIDE_W;
#pragma unroll
    for (int rbk = 0; rbk < 2; ++rbk) {
        const int rl = 32 * rbk + l31;                   // the lane's row of this pass, within the wave
        const int t = q0 + 64 * wave + rl;
        const bool valid = t < p.Tq;
        const float* rc = rec + (view_of(valid ? t : p.Tq - 1, p.Pq, p.invPq) - n_first) * BREC;
        static_for_bwd<2 * DB>([&](auto SC) {
            constexpr int sl = decltype(SC)::value, d = sl >> 1, gp = sl & 1;
            constexpr uint32_t DE = gta_layout_desc(GTA_LAYOUT_MS, 2 * sl), DO = gta_layout_desc(GTA_LAYOUT_MS, 2 * sl + 1);
            const f32x16_t& acc = dq[rbk][d];
            float x[1][8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float fa = acc[8 * gp + i] * c1, fb = acc[8 * gp + 4 + i] * c1;
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(fa), __float_as_uint(fb), false, false);   // fa[32..63] <-> fb[0..31]
                x[0][i] = __uint_as_float(sw[0]);
                x[0][4 + i] = __uint_as_float(sw[1]);
            }
            const int c = 2 * sl + lh;
            if constexpr (sl < 3) {                      // se3 | se3: d trans_coeff = dq'_3 (t_E . q_{0:3}) per 4-vector, then A_q^T
                float q8[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(side + rl * 96 + c * 16), q8);
                const float t0 = rc[BREC_T], t1 = rc[BREC_T + 1], t2 = rc[BREC_T + 2];
                if (valid) {
                    dcpart += x[0][3] * (t0 * q8[0] + t1 * q8[1] + t2 * q8[2]);
                    dcpart += x[0][7] * (t0 * q8[4] + t1 * q8[5] + t2 * q8[6]);
                }
                chunk_apply<true, 1>(DE, rc + BREC_MT, rc + BREC_D1T, rc + BREC_D2T, nullptr, x);
            } else if constexpr (sl == 3) {
                chunk_apply<true, 1>(DE, rc + BREC_MT, rc + BREC_D1T, rc + BREC_D2T, nullptr, x);
            } else {
                f32x2_t cs[4];
                const int pc = 2 * (c - 9);
                const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(side + S::SIDE_RAW + rl * 96 + (pc < 0 ? 0 : pc) * 16);
                const f32x4_t c1v = *reinterpret_cast<const f32x4_t*>(side + S::SIDE_RAW + rl * 96 + (pc < 0 ? 1 : pc + 1) * 16);
                cs[0] = f32x2_t{c0.x, c0.y}; cs[1] = f32x2_t{c0.z, c0.w}; cs[2] = f32x2_t{c1v.x, c1v.y}; cs[3] = f32x2_t{c1v.z, c1v.w};
                if constexpr (sl == 4) {
                    if (lh == 0) chunk_apply<true, 1>(DE, rc + BREC_MT, rc + BREC_D1T, rc + BREC_D2T, cs, x);
                    else chunk_apply<true, 1>(DO, rc + BREC_MT, rc + BREC_D1T, rc + BREC_D2T, cs, x);
                } else {
                    chunk_apply<true, 1>(DE, rc + BREC_MT, rc + BREC_D1T, rc + BREC_D2T, cs, x);
                }
            }
            if (valid) g_store_chunk<ESZ>(dqg + (long)t * p.dq_st * ESZ, c, x[0]);
        });
    }
    __syncthreads();
    const float dc_wg = wg_sum256(dcpart, reinterpret_cast<float*>(smem + S::OFF_SCR), tid);
    if (tid == 0) {                                      // the partials' slots are per 128 rows (gta_abi.cpp): this workgroup owns two of them
        const int n_q128 = (p.Tq + 127) / 128;
        p.dc_partial[p.dc_off_dq + bh * n_q128 + 2 * qt] = dc_wg;
        if (2 * qt + 1 < n_q128) p.dc_partial[p.dc_off_dq + bh * n_q128 + 2 * qt + 1] = 0.f;
    }
}

template <int ESZ>
__global__ __launch_bounds__(256, 1) void gta_bwd_dkv64_kernel(const GtaBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bwd_dkv64_body<ESZ>(p, smem, blockIdx.x, gridDim.x);
}
template <int ESZ>
__global__ __launch_bounds__(256, 1) void gta_bwd_dq64_kernel(const GtaBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bwd_dq64_body<ESZ>(p, smem, blockIdx.x, gridDim.x);
}
// Both generated kernels in ONE launch: workgroups [0, n_dkv) walk dK/dV blocks, the rest dQ blocks (the longer blocks first: the launch ends on the
// shorter ones).  The two depend on the q-side pre-pass only, not on each other; as two launches the second waited for the first's last workgroups
// (their ends spread over ~10 us) and paid its own ramp; here the second kind takes the CUs as the first leaves them.  (n_dkv a multiple of 8:
// workgroup L of either part runs on XCD L % 8, which both bodies' work maps rely on.)
template <int ESZ>
__global__ __launch_bounds__(256, 1) void gta_bwd_dqkv64_kernel(const GtaBwdParams p, const int n_dq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int L = blockIdx.x, n_dkv = (int)gridDim.x - n_dq;
    if (L < n_dkv) bwd_dkv64_body<ESZ>(p, smem, L, n_dkv);
    else bwd_dq64_body<ESZ>(p, smem, L - n_dkv, n_dq);
}

// deterministic two-level sum: 1024 threads each take a fixed strided subset, then a fixed tree
__global__ __launch_bounds__(1024) void gta_reduce_kernel(const float* __restrict__ part, int n, float* __restrict__ out,
                                                          const float* __restrict__ neg_div) {
    __shared__ float sm[1024];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;                 // 4 independent loads in flight per thread
    int i = threadIdx.x;
    for (; i + 3 * 1024 < n; i += 4 * 1024) { a0 += part[i]; a1 += part[i + 1024]; a2 += part[i + 2048]; a3 += part[i + 3072]; }
    for (; i < n; i += 1024) a0 += part[i];
    sm[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    for (int o = 512; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = neg_div ? -sm[0] / *neg_div : sm[0];
}

// ================================================================================================
// 5. The fp32-faithful instances (r06; GTA_FLAG_FP32_PRODUCTS, fp32 inputs, dh <= 64): the compiled dQ and dK/dV walks over hi / lo image
// pairs, every product of the five contractions as THREE matrix instructions (lo*hi + hi*lo + hi*hi, as the forward's X3 instances:
// gta_fwd2.hip), P and dS split the same way before they become operands; rho_q / adjoint rho stay where they are (fp32, in the pre-pass and
// the epilogues).  This is the backward of the reference's `mixed_prec: False` training (runs/clevrtr/GTA/gta/config.yaml:55,
// source/trainer.py:106) on the matrix cores at 3/16 of the bf16 rate instead of 1/16 (gta_plain32.hip) and without the eight per-row rho
// launches of the generic route.  A tile is four images [A hi | B hi | A lo | B lo] (K', V' from the forward's workspace; Q'', dO~ from
// gta_bwd_prep_kernel<.., X3>); two ring stages of 4 images: tile j + 1 streams in under tile j's 72 matrix instructions.
// ================================================================================================
GTA_DEV void split_acc8(const f32x16_t& a, int t, bf16x8_t& hi, bf16x8_t& lo) {      // accumulator registers 8t..8t+7 -> bf16 hi, lo
    const u32x4_t h = pack_acc8(a, t);
    float x[8], r[8];
    unpack8(h, x);
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = a[8 * t + i] - x[i];
    hi = __builtin_bit_cast(bf16x8_t, h);
    lo = __builtin_bit_cast(bf16x8_t, pack8(r));
}
GTA_DEV f32x16_t mfma3(const bf16x8_t& ah, const bf16x8_t& al, const bf16x8_t& bh, const bf16x8_t& bl, f32x16_t c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}

template <int DHP>
struct DqX3Smem {
    static constexpr int CHP = DHP / 8;
    static constexpr int IMG = BN * DHP * 2;
    static constexpr int STAGE = 4 * IMG;
    static constexpr int RING = 2 * STAGE;
    static constexpr int OROW = DHP + 4;
    static constexpr int OST = 128 * OROW * 4;
    static_assert(OST <= RING, "dQ staging must fit the ring");
    static constexpr int OFF_RING = 0;
    static constexpr int OFF_SCR = RING;
    static constexpr int OFF_REC = RING + 32;
    static constexpr int total(int nviews) { return OFF_REC + nviews * BREC * 4; }
};

template <int DHP>
GTA_DEV void bwd_dq_x3_body(const GtaBwdParams& p, char* smem, const int L, const int nwg) {
    using S = DqX3Smem<DHP>;
    constexpr int CHP = S::CHP, KS = DHP / 16, DB = DHP / 32, BM = 128, ESZ = 4, IMG = S::IMG;
    constexpr int ITEMS = 2 * CHP / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    int w;
    {
        const int xcd = L & 7, idx = L >> 3, q8 = nwg >> 3, r8 = nwg & 7;
        w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int n_q128 = (p.Tq + BM - 1) / BM;
    const int bh = w / n_q128, qt = w - bh * n_q128;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * BM;
    const int n_tiles = (p.Tk + BN - 1) / BN;
    const int n_qt64 = (p.Tq + BN - 1) / BN;
    const int ch_real = p.dh >> 3;
    char* ring = smem + S::OFF_RING;
    float* rec = reinterpret_cast<float*>(smem + S::OFF_REC);
    const char* kvimg = (const char*)p.kvimg + ((long)b * p.H + h) * n_tiles * (long)S::STAGE;

    // Q''/dO~ images (hi and lo) of this workgroup's two 64-row tiles -> ring stages 0, 1
    const long qtile0 = ((long)b * p.H + h) * n_qt64 + 2 * qt;
    const int n_my_qt = (2 * qt + 1 < n_qt64) ? 2 : 1;
    dma_linear4<S::STAGE>(ring, (const char*)p.qimg + qtile0 * S::STAGE, wave, lane);
    if (n_my_qt == 2) dma_linear4<S::STAGE>(ring + S::STAGE, (const char*)p.qimg + (qtile0 + 1) * S::STAGE, wave, lane);
    const int t_last = (q0 + BM - 1 < p.Tq ? q0 + BM - 1 : p.Tq - 1);
    const int n_first = q0 / p.Pq;
    const int n_cnt = t_last / p.Pq - n_first + 1;
    const float tc = p.trans_coeff ? *p.trans_coeff : 1.0f;
    if (p.vrep_q) stage_brec(rec, p.vrep_q, (long)b * p.Nq + n_first, n_cnt, 0, tc, tid, 256);
    const int my_row = wave * 32 + l31;
    const int my_tile = my_row >> 6;
    float lse2n = -1e30f, Dn = 0.f;
    if (my_tile < n_my_qt) {
        const float* st = p.stats + (qtile0 + my_tile) * 128;
        lse2n = st[my_row & 63];
        Dn = st[64 + (my_row & 63)];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16x8_t qfh[KS], qfl[KS], dfh[KS], dfl[KS];
    {
        const char* qi = ring + my_tile * S::STAGE;
        const int r = my_row & 63;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4_t z = {0, 0, 0, 0};
            qfh[ks] = qfl[ks] = dfh[ks] = dfl[ks] = __builtin_bit_cast(bf16x8_t, z);
            if (my_tile < n_my_qt) {
                const int off = (r * CHP + swz<CHP>(r, 2 * ks + lh)) * 16;
                qfh[ks] = *reinterpret_cast<const bf16x8_t*>(qi + off);
                dfh[ks] = *reinterpret_cast<const bf16x8_t*>(qi + IMG + off);
                qfl[ks] = *reinterpret_cast<const bf16x8_t*>(qi + 2 * IMG + off);
                dfl[ks] = *reinterpret_cast<const bf16x8_t*>(qi + 3 * IMG + off);
            }
        }
    }
    __syncthreads();
    dma_linear4<S::STAGE>(ring, kvimg, wave, lane);

    f32x16_t dq[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[d][i] = 0.f;
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
    const bool want_dtau = p.dt_partial != nullptr;
    float tA = 0.f, tB = 0.f, tC = 0.f;
#pragma unroll 1
    for (int j = 0; j < n_tiles; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // tile j has landed (nothing else is in flight), everyone is past tile j - 1
        __builtin_amdgcn_s_barrier();
        if (j + 1 < n_tiles) dma_linear4<S::STAGE>(ring + ((j + 1) & 1) * S::STAGE, kvimg + (long)(j + 1) * S::STAGE, wave, lane);
        const char* kf = ring + (j & 1) * S::STAGE;
        f32x16_t s0, s1, e0, e1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = lse2n; s1[i] = lse2n; e0[i] = Dn; e1[i] = Dn; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const char* a = kf + koff[ks];
            const bf16x8_t k0h = *reinterpret_cast<const bf16x8_t*>(a), k1h = *reinterpret_cast<const bf16x8_t*>(a + 32 * CHP * 16);
            const bf16x8_t v0h = *reinterpret_cast<const bf16x8_t*>(a + IMG), v1h = *reinterpret_cast<const bf16x8_t*>(a + IMG + 32 * CHP * 16);
            const bf16x8_t k0l = *reinterpret_cast<const bf16x8_t*>(a + 2 * IMG), k1l = *reinterpret_cast<const bf16x8_t*>(a + 2 * IMG + 32 * CHP * 16);
            const bf16x8_t v0l = *reinterpret_cast<const bf16x8_t*>(a + 3 * IMG), v1l = *reinterpret_cast<const bf16x8_t*>(a + 3 * IMG + 32 * CHP * 16);
            s0 = mfma3(k0h, k0l, qfh[ks], qfl[ks], s0);      // S^T  = K' Q''^T
            s1 = mfma3(k1h, k1l, qfh[ks], qfl[ks], s1);
            e0 = mfma3(v0h, v0l, dfh[ks], dfl[ks], e0);      // dP^T = V' dO~^T
            e1 = mfma3(v1h, v1l, dfh[ks], dfl[ks], e1);
        }
        const bool tail = (j == n_tiles - 1) && (p.Tk & (BN - 1));
        const int kbase = j * BN + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sv0 = s0[r], sv1 = s1[r];            // S - lse2 (log2 units)
            float p0 = __builtin_amdgcn_exp2f(sv0), p1 = __builtin_amdgcn_exp2f(sv1);
            if (tail) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key >= p.Tk) p0 = 0.f;
                if (key + 32 >= p.Tk) p1 = 0.f;
            }
            const float d0 = p0 * e0[r], d1 = p1 * e1[r];
            if (want_dtau) {                                 // row sums of dS (S - lse2), dS and P (S - lse2): see the epilogue
                tA = __builtin_fmaf(d0, sv0, tA); tA = __builtin_fmaf(d1, sv1, tA);
                tB += d0 + d1;
                tC = __builtin_fmaf(p0, sv0, tC); tC = __builtin_fmaf(p1, sv1, tC);
            }
            s0[r] = d0;
            s1[r] = d1;
        }
        bf16x8_t dsh[4], dsl[4];                             // slab sl = 2 half + t
        split_acc8(s0, 0, dsh[0], dsl[0]); split_acc8(s0, 1, dsh[1], dsl[1]);
        split_acc8(s1, 0, dsh[2], dsl[2]); split_acc8(s1, 1, dsh[3], dsl[3]);
        // dQ'^T += K'^T dS^T: A = K'^T by transpose-reads of the hi and the lo image
        const uint32_t kb_h = lds_addr(kf), kb_l = lds_addr(kf + 2 * IMG);
        constexpr int SL = 16 * CHP * 16;
        static_for_bwd<DB>([&](auto DC) {
            constexpr int d = decltype(DC)::value;
            u32x2_t hl[4], hh[4], ll[4], lh_[4];
            {
                const uint32_t a0 = kb_h + voff[d][0], a1 = kb_h + voff[d][1], c0 = kb_l + voff[d][0], c1 = kb_l + voff[d][1];
                hl[0] = lds_tr16_b64<0>(a0);      hh[0] = lds_tr16_b64<0>(a1);      ll[0] = lds_tr16_b64<0>(c0);      lh_[0] = lds_tr16_b64<0>(c1);
                hl[1] = lds_tr16_b64<SL>(a0);     hh[1] = lds_tr16_b64<SL>(a1);     ll[1] = lds_tr16_b64<SL>(c0);     lh_[1] = lds_tr16_b64<SL>(c1);
                hl[2] = lds_tr16_b64<2 * SL>(a0); hh[2] = lds_tr16_b64<2 * SL>(a1); ll[2] = lds_tr16_b64<2 * SL>(c0); lh_[2] = lds_tr16_b64<2 * SL>(c1);
                hl[3] = lds_tr16_b64<3 * SL>(a0); hh[3] = lds_tr16_b64<3 * SL>(a1); ll[3] = lds_tr16_b64<3 * SL>(c0); lh_[3] = lds_tr16_b64<3 * SL>(c1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const u32x4_t ah = {hl[sl].x, hl[sl].y, hh[sl].x, hh[sl].y}, al = {ll[sl].x, ll[sl].y, lh_[sl].x, lh_[sl].y};
                dq[d] = mfma3(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, al), dsh[sl], dsl[sl], dq[d]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    // ---- epilogue: dq = A_q^T (c1 dQ') ; d trans_coeff through A_q (as bwd_dq_body, fp32) ----
    const float c1 = p.scale / (p.tau ? *p.tau : 1.0f);
    __syncthreads();
    float* ost = reinterpret_cast<float*>(smem + S::OFF_RING);
    {
        const int r = wave * 32 + l31;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t v = {dq[d][4 * g] * c1, dq[d][4 * g + 1] * c1, dq[d][4 * g + 2] * c1, dq[d][4 * g + 3] * c1};
                *reinterpret_cast<f32x4_t*>(ost + r * S::OROW + 32 * d + 8 * g + 4 * lh) = v;
            }
    }
    __syncthreads();
    const char* qg = (const char*)p.q + ((long)b * p.q_sb + (long)h * p.q_sh) * ESZ;
    char* dqg = (char*)p.dq + ((long)b * p.dq_sb + (long)h * p.dq_sh) * ESZ;
    // d tau = -(1/tau) sum_i <q_i, dq_i> = -(ln2/tau) sum_ij dS_ij S2_ij  (gta_hip.h; S2 = the log2-unit logits).  Here the sum is formed in the walk,
    // in fp32, and CENTRED per row: sum_j dS_ij (S2_ij - m_i) with m_i = sum_j P_ij (S2_ij - lse2_i).  Exactly, sum_j dS_ij = 0 and the centring
    // changes nothing; in split-bf16 arithmetic a row's dS carries a common-mode error (D_i comes from the forward's o, lse2_i from the
    // forward's row sums: 2^-17 each) that the uncentred forms -- this one and <q, dq> alike -- multiply by the row's mean logit.
    float dcpart = 0.f, dtpart = 0.f;
    if (want_dtau) {
        const float tBf = tB + __shfl_xor(tB, 32), tCf = tC + __shfl_xor(tC, 32), tAf = tA + __shfl_xor(tA, 32);
        if (lh == 0 && q0 + my_row < p.Tq) dtpart = LN2 * (tAf - tBf * tCf);
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int item = wave + 4 * it;
        const int c = item >> 1;
        const int r = lane + 64 * (item & 1);
        const int t = q0 + r;
        if (c < ch_real && t < p.Tq) {
            const uint32_t desc = p.ctab[c];
            float x[1][8];
            const f32x4_t a = *reinterpret_cast<const f32x4_t*>(ost + r * S::OROW + 8 * c);
            const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(ost + r * S::OROW + 8 * c + 4);
            x[0][0] = a.x; x[0][1] = a.y; x[0][2] = a.z; x[0][3] = a.w;
            x[0][4] = bb.x; x[0][5] = bb.y; x[0][6] = bb.z; x[0][7] = bb.w;
            if (desc) {
                const int n = view_of(t, p.Pq, p.invPq) - n_first;
                const float* rc = rec + n * BREC;
                if (!(desc & GTA_CHUNK_SO3) && (cd_lo(desc) == GTA_HALF_SE3 || cd_hi(desc) == GTA_HALF_SE3)) {
                    float q8[8];
                    g_load_chunk<ESZ>(qg + (long)t * p.q_st * ESZ, c, q8);
                    const float t0 = rc[BREC_T], t1 = rc[BREC_T + 1], t2 = rc[BREC_T + 2];
                    if (cd_lo(desc) == GTA_HALF_SE3) dcpart += x[0][3] * (t0 * q8[0] + t1 * q8[1] + t2 * q8[2]);
                    if (cd_hi(desc) == GTA_HALF_SE3) dcpart += x[0][7] * (t0 * q8[4] + t1 * q8[5] + t2 * q8[6]);
                }
                f32x2_t cs[4];
                if (p.cs_q) load_cs(desc, p.cs_q + ((long)b * p.Tq + t) * 2 * p.nso2, cs);
                chunk_apply<true, 1>(desc, rc + BREC_MT, rc + BREC_D1T, rc + BREC_D2T, cs, x);
            }
            g_store_chunk<ESZ>(dqg + (long)t * p.dq_st * ESZ, c, x[0]);
        }
    }
    const float dc_wg = wg_sum256(dcpart, reinterpret_cast<float*>(smem + S::OFF_SCR), tid);
    if (tid == 0) p.dc_partial[p.dc_off_dq + w] = dc_wg;
    if (want_dtau) {
        __syncthreads();
        const float dt_wg = wg_sum256(dtpart, reinterpret_cast<float*>(smem + S::OFF_SCR), tid);
        if (tid == 0) p.dt_partial[w] = dt_wg;
    }
}

template <int DHP>
struct DkvX3Smem {
    static constexpr int CHP = DHP / 8;
    static constexpr int IMG = BN * DHP * 2;
    static constexpr int STAGE = 4 * IMG;                  // [Q''hi | dO~hi | Q''lo | dO~lo] of one 64-row tile
    static constexpr int RING = 2 * STAGE;
    static constexpr int OFF_RING = 0;                     // (the workgroup's two K'/V' tiles sit in the ring until their fragments are in VGPRs)
    static constexpr int OFF_STATS = RING;                 // [2][128] floats
    static constexpr int OFF_SCR = OFF_STATS + 2 * 128 * 4;
    static constexpr int OFF_REC = OFF_SCR + 32;
    static constexpr int OROW = DHP + 4;
    static constexpr int OST = 128 * OROW * 4;
    static_assert(OST <= RING, "staging must fit the ring");
    static constexpr int total(int nviews) { return OFF_REC + nviews * BREC * 4; }
};

template <int DHP>
GTA_DEV void bwd_dkv_x3_body(const GtaBwdParams& p, char* smem, const int L, const int nwg) {
    using S = DkvX3Smem<DHP>;
    constexpr int CHP = S::CHP, KS = DHP / 16, DB = DHP / 32, BK = 128, ESZ = 4, IMG = S::IMG;
    constexpr int ITEMS = 2 * CHP / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    int w;
    {
        const int xcd = L & 7, idx = L >> 3, q8 = nwg >> 3, r8 = nwg & 7;
        w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int n_k128 = (p.Tk + BK - 1) / BK;
    const int bh = w / n_k128, kt = w - bh * n_k128;
    const int b = bh / p.H, h = bh - b * p.H;
    const int k0 = kt * BK;
    const int n_kt64 = (p.Tk + BN - 1) / BN;
    const int n_qt = (p.Tq + BN - 1) / BN;
    const int ch_real = p.dh >> 3;
    char* ring = smem + S::OFF_RING;
    float* stats = reinterpret_cast<float*>(smem + S::OFF_STATS);
    float* rec = reinterpret_cast<float*>(smem + S::OFF_REC);
    const long ktile0 = ((long)b * p.H + h) * n_kt64 + 2 * kt;
    const int n_my_kt = (2 * kt + 1 < n_kt64) ? 2 : 1;
    const char* qimg = (const char*)p.qimg + ((long)b * p.H + h) * n_qt * (long)S::STAGE;
    const float* gstats = p.stats + ((long)b * p.H + h) * n_qt * 128;
    dma_linear4<S::STAGE>(ring, (const char*)p.kvimg + ktile0 * S::STAGE, wave, lane);
    if (n_my_kt == 2) dma_linear4<S::STAGE>(ring + S::STAGE, (const char*)p.kvimg + (ktile0 + 1) * S::STAGE, wave, lane);
    const int t_last = (k0 + BK - 1 < p.Tk ? k0 + BK - 1 : p.Tk - 1);
    const int n_first = k0 / p.Pk;
    const int n_cnt = t_last / p.Pk - n_first + 1;
    const float tc = p.trans_coeff ? *p.trans_coeff : 1.0f;
    if (p.vrep_k) stage_brec(rec, p.vrep_k, (long)b * p.Nk + n_first, n_cnt, 1, tc, tid, 256);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // this wave's 32 keys: K' and V' fragments, hi and lo (MFMA B operands), stay in VGPRs
    const int my_key = wave * 32 + l31;
    const int my_kt = my_key >> 6;
    bf16x8_t kh[KS], kl[KS], vh[KS], vl[KS];
    {
        const char* ki = ring + my_kt * S::STAGE;
        const int r = my_key & 63;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4_t z = {0, 0, 0, 0};
            kh[ks] = kl[ks] = vh[ks] = vl[ks] = __builtin_bit_cast(bf16x8_t, z);
            if (my_kt < n_my_kt) {
                const int off = (r * CHP + swz<CHP>(r, 2 * ks + lh)) * 16;
                kh[ks] = *reinterpret_cast<const bf16x8_t*>(ki + off);
                vh[ks] = *reinterpret_cast<const bf16x8_t*>(ki + IMG + off);
                kl[ks] = *reinterpret_cast<const bf16x8_t*>(ki + 2 * IMG + off);
                vl[ks] = *reinterpret_cast<const bf16x8_t*>(ki + 3 * IMG + off);
            }
        }
    }
    __syncthreads();      // every wave holds its fragments: the ring is free
    dma_linear4<S::STAGE>(ring, qimg, wave, lane);
    dma_stats(stats, gstats, wave, lane);

    f32x16_t dk[DB], dv[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[d][i] = 0.f; dv[d][i] = 0.f; }
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
#pragma unroll 1
    for (int j = 0; j < n_qt; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // tile j's images and statistics have landed, everyone is past tile j - 1
        __builtin_amdgcn_s_barrier();
        if (j + 1 < n_qt) {
            dma_linear4<S::STAGE>(ring + ((j + 1) & 1) * S::STAGE, qimg + (long)(j + 1) * S::STAGE, wave, lane);
            dma_stats(stats + ((j + 1) & 1) * 128, gstats + (long)(j + 1) * 128, wave, lane);
        }
        const char* qi = ring + (j & 1) * S::STAGE;         // [Q''hi | dO~hi | Q''lo | dO~lo]
        const float* stj = stats + (j & 1) * 128;
        constexpr int SL = 16 * CHP * 16;
        static_for_bwd<2>([&](auto QBC) {                   // 32 query rows at a time
            constexpr int qb = decltype(QBC)::value;
            f32x16_t s, e;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(stj + 32 * qb + 8 * g + 4 * lh);
                const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(stj + 64 + 32 * qb + 8 * g + 4 * lh);
                s[4 * g] = l4.x; s[4 * g + 1] = l4.y; s[4 * g + 2] = l4.z; s[4 * g + 3] = l4.w;
                e[4 * g] = d4.x; e[4 * g + 1] = d4.y; e[4 * g + 2] = d4.z; e[4 * g + 3] = d4.w;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const char* a = qi + koff[ks] + qb * 32 * CHP * 16;
                const bf16x8_t qh = *reinterpret_cast<const bf16x8_t*>(a), dh_ = *reinterpret_cast<const bf16x8_t*>(a + IMG);
                const bf16x8_t ql = *reinterpret_cast<const bf16x8_t*>(a + 2 * IMG), dl = *reinterpret_cast<const bf16x8_t*>(a + 3 * IMG);
                s = mfma3(qh, ql, kh[ks], kl[ks], s);        // S - lse2 = Q'' K'^T - lse2
                e = mfma3(dh_, dl, vh[ks], vl[ks], e);       // dP - D   = dO~ V'^T - D
            }
            f32x16_t ds;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = __builtin_amdgcn_exp2f(s[i]);
                s[i] = pv;
                ds[i] = pv * e[i];
            }
            bf16x8_t pfh[2], pfl[2], dsh[2], dsl[2];
            split_acc8(s, 0, pfh[0], pfl[0]); split_acc8(s, 1, pfh[1], pfl[1]);
            split_acc8(ds, 0, dsh[0], dsl[0]); split_acc8(ds, 1, dsh[1], dsl[1]);
            // dV'^T += dO~^T P ; dK'^T += Q''^T dS   (A operands by transpose-reads of the row-major hi and lo images)
            const uint32_t base = lds_addr(qi) + qb * 32 * CHP * 16;
            static_for_bwd<DB>([&](auto DC) {
                constexpr int d = decltype(DC)::value;
                u32x2_t r_[4][2][2];                          // [image: Qhi, dOhi, Qlo, dOlo][t][half]
#pragma unroll
                for (int im = 0; im < 4; ++im) {
                    const uint32_t a0 = base + im * IMG + voff[d][0], a1 = base + im * IMG + voff[d][1];
                    r_[im][0][0] = lds_tr16_b64<0>(a0);  r_[im][0][1] = lds_tr16_b64<0>(a1);
                    r_[im][1][0] = lds_tr16_b64<SL>(a0); r_[im][1][1] = lds_tr16_b64<SL>(a1);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    auto frag = [&](int im) {
                        const u32x4_t v = {r_[im][t][0].x, r_[im][t][0].y, r_[im][t][1].x, r_[im][t][1].y};
                        return __builtin_bit_cast(bf16x8_t, v);
                    };
                    dv[d] = mfma3(frag(1), frag(3), pfh[t], pfl[t], dv[d]);
                    dk[d] = mfma3(frag(0), frag(2), dsh[t], dsl[t], dk[d]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }

    // ---- epilogue: dk = B_k^T (ln2 dK'), dv = B_k^T dV' ; d trans_coeff through B_k (as bwd_dkv_body, fp32) ----
    const bool xv = (p.flags & GTA_FLAG_V_TRANSFORM) != 0;
    const char* kg = (const char*)p.k + ((long)b * p.k_sb + (long)h * p.k_sh) * ESZ;
    const char* vg = (const char*)p.v + ((long)b * p.v_sb + (long)h * p.v_sh) * ESZ;
    char* dkg = (char*)p.dk + ((long)b * p.dk_sb + (long)h * p.dk_sh) * ESZ;
    char* dvg = (char*)p.dv + ((long)b * p.dv_sb + (long)h * p.dv_sh) * ESZ;
    float* ost = reinterpret_cast<float*>(smem + S::OFF_RING);
    float dcpart = 0.f;
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {            // 0: dK, 1: dV
        __syncthreads();
        {
            const int r = wave * 32 + l31;
            const float sc = which == 0 ? LN2 : 1.0f;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16_t& acc = which == 0 ? dk[d] : dv[d];
                    const f32x4_t v = {acc[4 * g] * sc, acc[4 * g + 1] * sc, acc[4 * g + 2] * sc, acc[4 * g + 3] * sc};
                    *reinterpret_cast<f32x4_t*>(ost + r * S::OROW + 32 * d + 8 * g + 4 * lh) = v;
                }
        }
        __syncthreads();
        const bool xf = which == 0 || xv;
        const char* rawg = which == 0 ? kg : vg;
        const long raw_st = (which == 0 ? p.k_st : p.v_st) * ESZ;
        char* outg = which == 0 ? dkg : dvg;
        const long out_st = (which == 0 ? p.dk_st : p.dv_st) * ESZ;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int item = wave + 4 * it;
            const int c = item >> 1;
            const int r = lane + 64 * (item & 1);
            const int t = k0 + r;
            if (c < ch_real && t < p.Tk) {
                const uint32_t desc = p.ctab[c];
                float x[1][8];
                const f32x4_t a = *reinterpret_cast<const f32x4_t*>(ost + r * S::OROW + 8 * c);
                const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(ost + r * S::OROW + 8 * c + 4);
                x[0][0] = a.x; x[0][1] = a.y; x[0][2] = a.z; x[0][3] = a.w;
                x[0][4] = bb.x; x[0][5] = bb.y; x[0][6] = bb.z; x[0][7] = bb.w;
                if (desc && xf) {
                    const int n = view_of(t, p.Pk, p.invPk) - n_first;
                    const float* rc = rec + n * BREC;
                    if (!(desc & GTA_CHUNK_SO3) && (cd_lo(desc) == GTA_HALF_SE3 || cd_hi(desc) == GTA_HALF_SE3)) {
                        float r8[8];
                        g_load_chunk<ESZ>(rawg + (long)t * raw_st, c, r8);
                        const float t0 = rc[BREC_T], t1 = rc[BREC_T + 1], t2 = rc[BREC_T + 2];
                        if (cd_lo(desc) == GTA_HALF_SE3) dcpart += (x[0][0] * t0 + x[0][1] * t1 + x[0][2] * t2) * r8[3];
                        if (cd_hi(desc) == GTA_HALF_SE3) dcpart += (x[0][4] * t0 + x[0][5] * t1 + x[0][6] * t2) * r8[7];
                    }
                    f32x2_t cs[4];
                    if (p.cs_k) load_cs(desc, p.cs_k + ((long)b * p.Tk + t) * 2 * p.nso2, cs);
                    chunk_apply<true, 1>(desc, rc + BREC_MT, rc + BREC_D1T, rc + BREC_D2T, cs, x);
                }
                g_store_chunk<ESZ>(outg + (long)t * out_st, c, x[0]);
            }
        }
    }
    const float dc_wg = wg_sum256(dcpart, reinterpret_cast<float*>(smem + S::OFF_SCR), tid);
    if (tid == 0) p.dc_partial[p.dc_off_dkv + w] = dc_wg;
}

template <int DHP>
__global__ __launch_bounds__(256, 2) void gta_bwd_dqkv_x3_kernel(const GtaBwdParams p, const int n_dq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int L = blockIdx.x, n_dkv = (int)gridDim.x - n_dq;
    if (L < n_dkv) bwd_dkv_x3_body<DHP>(p, smem, L, n_dkv);
    else bwd_dq_x3_body<DHP>(p, smem, L - n_dkv, n_dq);
}

__global__ __launch_bounds__(1024) void gta_reduce_kernel(const float* __restrict__ part, int n, float* __restrict__ out, const float* __restrict__ tau);

template <int DHP>
int run_bwd_x3(const GtaBwdParams& p, hipStream_t stream) {
    const int n_qt = (p.Tq + BN - 1) / BN;
    using PS = BPrepSmem<DHP, 4, true>;
    if (int rc = gta_lds_optin<&gta_bwd_prep_kernel<DHP, 4, true>>(PS::total(GTA_MAX_VIEWS))) return rc;
    constexpr int LDS_MAX = DqX3Smem<DHP>::total(GTA_MAX_VIEWS) > DkvX3Smem<DHP>::total(GTA_MAX_VIEWS) ? DqX3Smem<DHP>::total(GTA_MAX_VIEWS) : DkvX3Smem<DHP>::total(GTA_MAX_VIEWS);
    if (int rc = gta_lds_optin<&gta_bwd_dqkv_x3_kernel<DHP>>(LDS_MAX)) return rc;
    const long prep_grid = ((long)p.B * n_qt + 7) / 8 * 8 * p.H;
    const long n_dq = (long)p.B * p.H * ((p.Tq + 127) / 128), n_dkv = (long)p.B * p.H * ((p.Tk + 127) / 128);
    if (prep_grid > 0x7fffffffL || n_dq + n_dkv > 0x7fffffffL) return GTA_E_UNSUPPORTED;
    hipLaunchKernelGGL((gta_bwd_prep_kernel<DHP, 4, true>), dim3((unsigned)prep_grid), dim3(256), PS::total(p.vrep_q ? p.Nq : 0), stream, p);
    const int lds_dq = DqX3Smem<DHP>::total(p.vrep_q ? p.Nq : 0), lds_dkv = DkvX3Smem<DHP>::total(p.vrep_k ? p.Nk : 0);
    hipLaunchKernelGGL((gta_bwd_dqkv_x3_kernel<DHP>), dim3((unsigned)(n_dq + n_dkv)), dim3(256), lds_dq > lds_dkv ? lds_dq : lds_dkv, stream, p, (int)n_dq);
    if (p.dtrans_coeff)
        hipLaunchKernelGGL(gta_reduce_kernel, dim3(1), dim3(1024), 0, stream, p.dc_partial, p.dc_total, p.dtrans_coeff, (const float*)nullptr);
    if (p.dtau)
        hipLaunchKernelGGL(gta_reduce_kernel, dim3(1), dim3(1024), 0, stream, p.dt_partial, (int)n_dq, p.dtau, p.tau);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

template <int DHP, int ESZ>
int run_bwd(const GtaBwdParams& p, hipStream_t stream) {
    const int n_qt = (p.Tq + BN - 1) / BN;
    if (int rc = gta_lds_optin<&gta_bwd_prep_kernel<DHP, ESZ>>(BPrepSmem<DHP, ESZ>::total(GTA_MAX_VIEWS))) return rc;
    if (int rc = gta_lds_optin<&gta_bwd_dq_kernel<DHP, ESZ>>(DqSmem<DHP>::total(GTA_MAX_VIEWS))) return rc;
    if (int rc = gta_lds_optin<&gta_bwd_dkv_kernel<DHP, ESZ>>(DkvSmem<DHP>::total(GTA_MAX_VIEWS))) return rc;
    const int lds_prep = BPrepSmem<DHP, ESZ>::total(p.vrep_q ? p.Nq : 0);
    const long prep_grid = ((long)p.B * n_qt + 7) / 8 * 8 * p.H;
    if (prep_grid > 0x7fffffffL) return GTA_E_UNSUPPORTED;
    hipLaunchKernelGGL((gta_bwd_prep_kernel<DHP, ESZ>), dim3((unsigned)prep_grid), dim3(256), lds_prep, stream, p);
    const int n_dq = p.B * p.H * ((p.Tq + 127) / 128);
    const int n_dkv = p.B * p.H * ((p.Tk + 127) / 128);
    bool ms_layout = false;
    if constexpr (DHP == 96 && ESZ == 2) {
        ms_layout = p.dh == 96 && p.nso2 == 12;          // (the generated kernels' epilogues are written for the MSN chunk layout)
        for (int c = 0; c < 12; ++c) ms_layout = ms_layout && p.ctab[c] == gta_layout_desc(GTA_LAYOUT_MS, c);
    }
    bool dq64 = false, dkv64 = false;
    if constexpr (DHP == 96 && ESZ == 2) {
        // 64 rows / keys per wave, 256 per workgroup (the generated streams) from half a chip of workgroups on (measured at B = 4 .. 32 per GPU at the MSN
        // shape: ahead at every size; below that the 128-row kernels' finer grain spreads the work over more CUs)
        const long n_dq64 = (long)p.B * p.H * ((p.Tq + 255) / 256);
        const long n_dkv64 = (long)p.B * p.H * ((p.Tk + 255) / 256);
        const int lds_dq = Dq64Smem::total(p.vrep_q ? p.Nq : 0), lds_dkv = Dkv64Smem::total(p.vrep_k ? p.Nk : 0);
        dq64 = ms_layout && p.Tk % BN == 0 && p.dt_partial == nullptr && lds_dq <= 160 * 1024 &&
               (long)p.Tq * p.q_st * ESZ < (1L << 31) && (n_dq64 >= 128 || (p.flags & GTA_FLAG_BWD_KEYS64)) && !(p.flags & GTA_FLAG_BWD_KEYS32);
        dkv64 = ms_layout && lds_dkv <= 160 * 1024 && (n_dkv64 >= 128 || (p.flags & GTA_FLAG_BWD_KEYS64)) && !(p.flags & GTA_FLAG_BWD_KEYS32);
        constexpr int LDS_MAX = Dq64Smem::total(GTA_MAX_VIEWS) > Dkv64Smem::total(GTA_MAX_VIEWS) ? Dq64Smem::total(GTA_MAX_VIEWS) : Dkv64Smem::total(GTA_MAX_VIEWS);
        if (dq64 && dkv64 && n_dkv64 % 8 == 0 && n_dq64 + n_dkv64 < 0x7fffffffL && !(p.flags & GTA_FLAG_BWD_SPLIT)) {                 // one launch for both (see the kernel)
            if (int rc = gta_lds_optin<&gta_bwd_dqkv64_kernel<ESZ>>(LDS_MAX)) return rc;
            hipLaunchKernelGGL((gta_bwd_dqkv64_kernel<ESZ>), dim3((unsigned)(n_dq64 + n_dkv64)), dim3(256), lds_dq > lds_dkv ? lds_dq : lds_dkv, stream,
                               p, (int)n_dq64);
        } else {
            if (dq64) {
                if (int rc = gta_lds_optin<&gta_bwd_dq64_kernel<ESZ>>(Dq64Smem::total(GTA_MAX_VIEWS))) return rc;
                hipLaunchKernelGGL((gta_bwd_dq64_kernel<ESZ>), dim3((unsigned)n_dq64), dim3(256), lds_dq, stream, p);
            }
            if (dkv64) {
                if (!dq64) hipLaunchKernelGGL((gta_bwd_dq_kernel<DHP, ESZ>), dim3(n_dq), dim3(256), DqSmem<DHP>::total(p.vrep_q ? p.Nq : 0), stream, p);
                if (int rc = gta_lds_optin<&gta_bwd_dkv64_kernel<ESZ>>(Dkv64Smem::total(GTA_MAX_VIEWS))) return rc;
                hipLaunchKernelGGL((gta_bwd_dkv64_kernel<ESZ>), dim3((unsigned)n_dkv64), dim3(256), lds_dkv, stream, p);
            }
        }
    }
    if (!dq64 && !dkv64 && n_dkv % 8 == 0 && (long)n_dq + n_dkv < 0x7fffffffL && !(p.flags & GTA_FLAG_BWD_SPLIT)) {      // the compiled pair in one launch
        const int lds_dq = DqSmem<DHP>::total(p.vrep_q ? p.Nq : 0), lds_dkv = DkvSmem<DHP>::total(p.vrep_k ? p.Nk : 0);
        constexpr int LDS_MAX = DqSmem<DHP>::total(GTA_MAX_VIEWS) > DkvSmem<DHP>::total(GTA_MAX_VIEWS) ? DqSmem<DHP>::total(GTA_MAX_VIEWS) : DkvSmem<DHP>::total(GTA_MAX_VIEWS);
        if (int rc = gta_lds_optin<&gta_bwd_dqkv_kernel<DHP, ESZ>>(LDS_MAX)) return rc;
        hipLaunchKernelGGL((gta_bwd_dqkv_kernel<DHP, ESZ>), dim3((unsigned)(n_dq + n_dkv)), dim3(256), lds_dq > lds_dkv ? lds_dq : lds_dkv, stream, p, n_dq);
    } else {
        if (!dq64 && !dkv64)
            hipLaunchKernelGGL((gta_bwd_dq_kernel<DHP, ESZ>), dim3(n_dq), dim3(256), DqSmem<DHP>::total(p.vrep_q ? p.Nq : 0), stream, p);
        if (!dkv64)
            hipLaunchKernelGGL((gta_bwd_dkv_kernel<DHP, ESZ>), dim3(n_dkv), dim3(256), DkvSmem<DHP>::total(p.vrep_k ? p.Nk : 0), stream, p);
    }
    if (p.dtrans_coeff)
        hipLaunchKernelGGL(gta_reduce_kernel, dim3(1), dim3(1024), 0, stream, p.dc_partial, p.dc_total, p.dtrans_coeff,
                           (const float*)nullptr);
    if (p.dtau)
        hipLaunchKernelGGL(gta_reduce_kernel, dim3(1), dim3(1024), 0, stream, p.dt_partial, n_dq, p.dtau, p.tau);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

}  // namespace

// the fp32-faithful backward on the matrix cores exists where the forward's two-stage X3 plan does: fp32 inputs, dh <= 64
bool gta_bwd_x3_takes(int dhp, int esz) { return esz == 4 && dhp <= 64; }
int gta_bwd_dispatch(const GtaBwdParams& p, int dhp, int esz, hipStream_t stream) {
    if (p.flags & GTA_FLAG_FP32_PRODUCTS) {
        if (!gta_bwd_x3_takes(dhp, esz)) return GTA_E_UNSUPPORTED;
        return dhp == 32 ? run_bwd_x3<32>(p, stream) : run_bwd_x3<64>(p, stream);
    }
#define GTA_CASEB(D) case D: return esz == 2 ? run_bwd<D, 2>(p, stream) : run_bwd<D, 4>(p, stream);
    switch (dhp) {
        GTA_CASEB(32)
        GTA_CASEB(64)
        GTA_CASEB(96)
        GTA_CASEB(128)
    }
#undef GTA_CASEB
    return GTA_E_UNSUPPORTED;
}
