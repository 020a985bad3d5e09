// gta_fwd.hip -- fused GTA attention forward for gfx950 (MI355X).
//
// One kernel replaces the ~70 ATen launches of one reference Attention call
// (gta.py:92-279 + AttnFn layers.py:202-211):
//     q' = rho_q^T-side(q)   k' = rho_k(k)   v' = rho_k(v)          (gta.py:134-242)
//     P  = softmax(scale/tau * q' k'^T)      o~ = P v'              (layers.py:207-210)
//     o  = rho_q^-1(o~)                                             (gta.py:246-276)
// without ever materialising [B,H,Tq,Tk].
//
// Structure (one workgroup = 4 waves = 128 query rows of one (batch, head)):
//   prologue   per-view rep blocks (trans_coeff-masked 4x4, D^1, D^2) -> LDS; Q tile -> rho, pre-
//              scaled by scale*log2(e)/tau, bf16 -> LDS -> each wave's MFMA B fragments in VGPRs
//   K/V tile   64 keys: raw rows HBM -> LDS by LDS-DMA (global_load_lds, 16 B/lane, fully
//              coalesced, no VGPRs); then lane == key row reads one 8-channel chunk, applies rho in
//              fp32 registers (chunk kind is wave-uniform -> scalar branch), writes K' row-major
//              and V'^T (key-permuted) bf16 tiles, both rotation-swizzled (conflict-free b128)
//   S^T=K'Q'^T v_mfma_f32_32x32x16_bf16, A = K' rows, B = Q' -> lane owns ONE query row (col=lane&31)
//              and 16 of each 32 keys: row max/sum are in-register + one cross-half exchange
//   O^T=V'^T P^T  A = V'^T rows (ds_read_b128), B = P straight from the S accumulators (the key
//              permutation of the V'^T tile makes the S C-layout the PV B-layout: no shuffles)
//   epilogue   O/l -> LDS (fp32) -> lane == row applies rho_q^-1 per chunk, stores out + LSE
// Next K/V tile's DMA is in flight while the MFMAs of the current one run; two workgroups per CU
// overlap one's VALU phases with the other's MFMA phases.
//
// X3 (GTA_FLAG_FP32_PRODUCTS, fp32 inputs): every operand of the two contractions is kept as a bf16 PAIR
// (hi = bf16(x), lo = bf16(x - hi): 16 significant bits) and every product is three MFMAs, lo*hi + hi*lo + hi*hi
// (the lo*lo term is below 2^-16 of the product).  The matrix cores have no fp32 input form at a usable rate
// (v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate); the split costs 3x the MFMAs and twice the LDS tiles and gives
// fp32-class results (measured against the fp64 oracle: see tests/test_gpu_precise.py) for the reference's
// ``mixed_prec: False`` configs (runs/clevrtr/GTA/gta/config.yaml:55).
#include "gta_common.h"
#include "gta_fwd_params.h"
#include "../../include/gta_hip.h"

namespace {

constexpr int BM = 128;  // query rows per workgroup (4 waves x 32)
constexpr int BN = 64;   // keys per tile
constexpr int NTHREADS = 256;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

template <int DHP, int ESZ, bool X3 = false>
struct Smem {
    static constexpr int NPL = X3 ? 2 : 1;                 // bf16 planes per operand tile: hi (and lo)
    static constexpr int CHP = DHP / 8;                    // padded chunks per row
    static constexpr int RAW_UNITS = DHP * ESZ / 16;       // 16-B units per raw row
    static constexpr int REP_BYTES = GTA_MAX_VIEWS * (GTA_QREC + GTA_KREC) * 4;
    static constexpr int RAW_BYTES = BN * DHP * ESZ;       // one raw tile (K or V)
    static constexpr int KF_BYTES = BN * DHP * 2;          // one plane of K' bf16 [64][DHP]
    static constexpr int VT_BYTES = DHP * BN * 2;          // one plane of V'^T bf16 [DHP][64]
    static constexpr int OROW = DHP + 4;                   // padded fp32 row of the O staging tile
    static constexpr int OST_BYTES = BM * OROW * 4;
    static constexpr int OFF_REP = 0;
    static constexpr int OFF_RAWK = REP_BYTES;
    static constexpr int OFF_RAWV = OFF_RAWK + RAW_BYTES;
    static constexpr int OFF_KF = OFF_RAWV + RAW_BYTES;    // K' planes, then V'^T planes
    static constexpr int OFF_VT = OFF_KF + NPL * KF_BYTES;
    static constexpr int TILE_END = OFF_VT + NPL * VT_BYTES;
    static constexpr int QS_PLANE = BM * DHP * 2;          // one plane of the Q' staging tile
    static constexpr int OFF_QS = OFF_KF;                  // Q' staging aliases K'/V'^T (BM*DHP*2 == KF+VT per plane)
    static constexpr int OFF_OST = OFF_RAWK;               // O staging aliases raw + final tiles
    static constexpr int TOTAL = (OFF_OST + OST_BYTES > TILE_END) ? OFF_OST + OST_BYTES : TILE_END;
    static_assert(BM * DHP * 2 == KF_BYTES + VT_BYTES, "Q staging must fit the final tiles");
    static_assert(TOTAL <= 160 * 1024, "LDS budget");
};

// x -> (hi, lo) bf16 planes: hi = bf16(x), lo = bf16(x - hi)
GTA_DEV void split8(const float* x, u32x4_t& hi, u32x4_t& lo) {
    hi = pack8(x);
    float xh[8], r[8];
    unpack8(hi, xh);
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = x[i] - xh[i];
    lo = pack8(r);
}

// ------------------------------------------------------------------------------------------
// global <-> register helpers for one 8-channel chunk
// ------------------------------------------------------------------------------------------
template <int ESZ>
GTA_DEV void gload_chunk(const char* rowptr, int c, float* x) {
    if (ESZ == 2) {
        const u32x4_t w = *reinterpret_cast<const u32x4_t*>(rowptr + c * 16);
        unpack8(w, x);
    } else {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(rowptr + c * 32);
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(rowptr + c * 32 + 16);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    }
}
template <int ESZ>
GTA_DEV void gstore_chunk(char* rowptr, int c, const float* x) {
    if (ESZ == 2) {
        *reinterpret_cast<u32x4_t*>(rowptr + c * 16) = pack8(x);
    } else {
        *reinterpret_cast<f32x4_t*>(rowptr + c * 32) = f32x4_t{x[0], x[1], x[2], x[3]};
        *reinterpret_cast<f32x4_t*>(rowptr + c * 32 + 16) = f32x4_t{x[4], x[5], x[6], x[7]};
    }
}
// raw LDS tile (rows x RAW_UNITS, rotation-swizzled) -> 8 floats of chunk c of row r
template <int DHP, int ESZ>
GTA_DEV void raw_read_chunk(const char* raw, int r, int c, float* x) {
    constexpr int U = Smem<DHP, ESZ>::RAW_UNITS;
    if (ESZ == 2) {
        const u32x4_t w = *reinterpret_cast<const u32x4_t*>(raw + (r * U + swz<U>(r, c)) * 16);
        unpack8(w, x);
    } else {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(raw + (r * U + swz<U>(r, 2 * c)) * 16);
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(raw + (r * U + swz<U>(r, 2 * c + 1)) * 16);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    }
}

// ------------------------------------------------------------------------------------------
// HBM -> raw LDS tile.  Each wave-instruction moves 64 consecutive 16-B units of the (swizzled)
// LDS image; the per-lane SOURCE address carries the swizzle (LDS-DMA destinations are
// wave-uniform base + lane*16, cdna_hip_programming.md 5.4 rule 21).
// ------------------------------------------------------------------------------------------
template <int DHP, int ESZ, bool DMA>
GTA_DEV void stage_raw_tile(char* smem_raw, const char* gbase, long row_stride_bytes, int row0,
                            int n_rows_total, int real_units, int wave, int lane) {
    constexpr int U = Smem<DHP, ESZ>::RAW_UNITS;
    constexpr int NI = BN * U / NTHREADS;   // wave-instructions per wave
    static_assert(BN * U % NTHREADS == 0, "tile must split evenly");
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int u0 = (wave * NI + i) * 64;          // wave-uniform first unit
        const int u = u0 + lane;
        const int r = u / U;
        const int pos = u - r * U;
        // logical unit stored at `pos` of row r: inverse rotation
        const int rot = swz_rot<U>(r);
        int gu = pos - rot;
        gu = gu < 0 ? gu + U : gu;
        gu = gu < real_units ? gu : real_units - 1;    // padding units: harmless duplicate
        int gr = row0 + r;
        gr = gr < n_rows_total ? gr : n_rows_total - 1; // rows past the end: duplicate (masked later)
        const char* src = gbase + (long)gr * row_stride_bytes + gu * 16;
        if (DMA) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)src,
                (__attribute__((address_space(3))) void*)(smem_raw + u0 * 16), 16, 0, 0);
        } else {
            const u32x4_t w = *reinterpret_cast<const u32x4_t*>(src);
            *reinterpret_cast<u32x4_t*>(smem_raw + u * 16) = w;
        }
    }
}

// ------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------
template <int DHP, int ESZ, bool DMA, bool X3 = false>
__global__ __launch_bounds__(NTHREADS, ((DHP <= 96 && !X3) ? 2 : 1)) void gta_fwd_kernel(const GtaFwdParams p) {
    using S = Smem<DHP, ESZ, X3>;
    constexpr int CHP = S::CHP;
    constexpr int NPL = S::NPL;
    constexpr int KS = DHP / 16;   // MFMA k-steps of Q K^T
    constexpr int DB = DHP / 32;   // 32-channel blocks of O^T
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- XCD-aware work remap: all query tiles of one (b,h) run on the same XCD so its K/V stay
    // in that XCD's L2 (bijective form, cdna_hip_programming.md 5.5 T1) ----
    int w;
    {
        const int nwg = gridDim.x, L = blockIdx.x;
        const int xcd = L & 7, idx = L >> 3, q8 = nwg >> 3, r8 = nwg & 7;
        w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int bh = w / p.n_qtiles;
    const int qt = w - bh * p.n_qtiles;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * BM;

    const char* qg = (const char*)p.q + ((long)b * p.q_sb + (long)h * p.q_sh) * ESZ;
    const char* kg = (const char*)p.k + ((long)b * p.k_sb + (long)h * p.k_sh) * ESZ;
    const char* vg = (const char*)p.v + ((long)b * p.v_sb + (long)h * p.v_sh) * ESZ;
    char* og = (char*)p.o + ((long)b * p.o_sb + (long)h * p.o_sh) * ESZ;
    const long q_rs = p.q_st * ESZ, k_rs = p.k_st * ESZ, v_rs = p.v_st * ESZ, o_rs = p.o_st * ESZ;
    const int ch_real = p.dh >> 3;
    const int real_units = p.dh * ESZ / 16;
    const int n_tiles = (p.Tk + BN - 1) / BN;

    // ---- first K/V tile on its way before anything else ----
    stage_raw_tile<DHP, ESZ, DMA>(smem + S::OFF_RAWK, kg, k_rs, 0, p.Tk, real_units, wave, lane);
    stage_raw_tile<DHP, ESZ, DMA>(smem + S::OFF_RAWV, vg, v_rs, 0, p.Tk, real_units, wave, lane);

    // ---- per-view rep records -> LDS (trans_coeff mask folded in, gta.py:40-44,135-141) ----
    float* rep = reinterpret_cast<float*>(smem + S::OFF_REP);
    float* qrec = rep;
    float* krec = rep + GTA_MAX_VIEWS * GTA_QREC;
    {
        const float tc = p.trans_coeff ? *p.trans_coeff : 1.0f;
        if (p.vrep_q) {
            for (int i = tid; i < p.Nq * GTA_QREC; i += NTHREADS) {
                const int n = i / GTA_QREC, e = i - n * GTA_QREC;
                const float* src = p.vrep_q + ((long)b * p.Nq + n) * GTA_VREP_STRIDE;
                float val = 0.f;
                if (e < 32) {                       // Aq = (E.m)^T  /  Oq = E.m
                    const int ee = e & 15, r = ee >> 2, c = ee & 3;
                    const int sr = (e < 16) ? c : r, sc = (e < 16) ? r : c;   // transpose for Aq
                    const float m = (sr == 3) ? (sc == 3 ? 1.f : 0.f) : (sc == 3 ? tc : 1.f);
                    val = src[GTA_VREP_INV + sr * 4 + sc] * m;
                } else if (e < GTA_QREC_D2) {        // D1 padded 3x4
                    const int ee = e - GTA_QREC_D1, r = ee >> 2, c = ee & 3;
                    val = c < 3 ? src[GTA_VREP_D1 + r * 3 + c] : 0.f;
                } else if (e < GTA_QREC_D1T) {       // D2 padded 5x8
                    const int ee = e - GTA_QREC_D2, r = ee >> 3, c = ee & 7;
                    val = c < 5 ? src[GTA_VREP_D2 + r * 5 + c] : 0.f;
                } else if (e < GTA_QREC_D2T) {       // D1^T
                    const int ee = e - GTA_QREC_D1T, r = ee >> 2, c = ee & 3;
                    val = c < 3 ? src[GTA_VREP_D1 + c * 3 + r] : 0.f;
                } else {                             // D2^T
                    const int ee = e - GTA_QREC_D2T, r = ee >> 3, c = ee & 7;
                    val = c < 5 ? src[GTA_VREP_D2 + c * 5 + r] : 0.f;
                }
                qrec[i] = val;
            }
        }
        if (p.vrep_k) {
            for (int i = tid; i < p.Nk * GTA_KREC; i += NTHREADS) {
                const int n = i / GTA_KREC, e = i - n * GTA_KREC;
                const float* src = p.vrep_k + ((long)b * p.Nk + n) * GTA_VREP_STRIDE;
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
    }
    __syncthreads();

    // ---- Q tile: load, rho, prescale, bf16 -> LDS (rotation swizzle) ----
    const float qscale = p.scale * LOG2E / (p.tau ? *p.tau : 1.0f);
    const bool xq = !(p.flags & GTA_FLAG_PRETRANSFORMED);
    {
        char* qs = smem + S::OFF_QS;
#pragma unroll
        for (int it = 0; it < 2 * CHP / 4; ++it) {
            const int c = wave + 4 * (it >> 1);                 // wave-uniform chunk
            const int r = lane + 64 * (it & 1);
            int t = q0 + r;
            t = t < p.Tq ? t : p.Tq - 1;
            float x[1][8];
            if (c < ch_real) {
                const uint32_t desc = p.ctab[c];
                gload_chunk<ESZ>(qg + (long)t * q_rs, c, x[0]);
                if (xq && desc) {
                    const int n = view_of(t, p.Pq, p.invPq);
                    f32x2_t cs[4];
                    if (p.cs_q) load_cs(desc, p.cs_q + ((long)b * p.Tq + t) * 2 * p.nso2, cs);
                    const float* rec = qrec + n * GTA_QREC;
                    chunk_apply<false, 1>(desc, rec + GTA_QREC_A, rec + GTA_QREC_D1, rec + GTA_QREC_D2, cs, x);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) x[0][i] *= qscale;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[0][i] = 0.f;
            }
            if constexpr (X3) {
                u32x4_t hi, lo;
                split8(x[0], hi, lo);
                *reinterpret_cast<u32x4_t*>(qs + (r * CHP + swz<CHP>(r, c)) * 16) = hi;
                *reinterpret_cast<u32x4_t*>(qs + S::QS_PLANE + (r * CHP + swz<CHP>(r, c)) * 16) = lo;
            } else {
                *reinterpret_cast<u32x4_t*>(qs + (r * CHP + swz<CHP>(r, c)) * 16) = pack8(x[0]);
            }
        }
    }
    __syncthreads();
    bf16x8_t qf[NPL][KS];
    {
        const char* qs = smem + S::OFF_QS;
        const int r = wave * 32 + l31;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qf[pl][ks] = *reinterpret_cast<const bf16x8_t*>(qs + pl * S::QS_PLANE + (r * CHP + swz<CHP>(r, 2 * ks + lh)) * 16);
    }
    // (the barrier at the top of the first loop iteration orders these reads before the first
    //  transform overwrites the aliased K'/V'^T region)

    f32x16_t oacc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const bool xk = !(p.flags & GTA_FLAG_PRETRANSFORMED);
    const bool xv = xk && (p.flags & GTA_FLAG_V_TRANSFORM);

    for (int j = 0; j < n_tiles; ++j) {
        // raw(j) landed (LDS-DMA is tracked by vmcnt) and visible to every wave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- transform: raw K,V rows -> K' (row-major) and V'^T (key-permuted), bf16 ----
        {
            const char* rawk = smem + S::OFF_RAWK;
            const char* rawv = smem + S::OFF_RAWV;
            char* kf = smem + S::OFF_KF;
            char* vt = smem + S::OFF_VT;
            const int r = lane;
            int t = j * BN + r;
            t = t < p.Tk ? t : p.Tk - 1;
            const int n = view_of(t, p.Pk, p.invPk);
            const float* rec = krec + n * GTA_KREC;
            const int pos = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);   // swap key bits 2,3
#pragma unroll
            for (int it = 0; it < CHP / 4; ++it) {
                const int c = wave + 4 * it;
                float x[2][8];
                if (c < ch_real) {
                    const uint32_t desc = p.ctab[c];
                    raw_read_chunk<DHP, ESZ>(rawk, r, c, x[0]);
                    raw_read_chunk<DHP, ESZ>(rawv, r, c, x[1]);
                    if (xk && desc) {
                        f32x2_t cs[4];
                        if (p.cs_k) load_cs(desc, p.cs_k + ((long)b * p.Tk + t) * 2 * p.nso2, cs);
                        if (xv) {
                            chunk_apply<false, 2>(desc, rec + GTA_KREC_B, rec + GTA_KREC_D1, rec + GTA_KREC_D2, cs, x);
                        } else {
                            chunk_apply<false, 1>(desc, rec + GTA_KREC_B, rec + GTA_KREC_D1, rec + GTA_KREC_D2, cs, x);
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { x[0][i] = 0.f; x[1][i] = 0.f; }
                }
                u32x4_t kpl[2], vpl[2];
                if constexpr (X3) { split8(x[0], kpl[0], kpl[1]); split8(x[1], vpl[0], vpl[1]); }
                else { kpl[0] = pack8(x[0]); vpl[0] = pack8(x[1]); }
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    *reinterpret_cast<u32x4_t*>(kf + pl * S::KF_BYTES + (r * CHP + swz<CHP>(r, c)) * 16) = kpl[pl];
                    const uint32_t wvv[4] = {vpl[pl].x, vpl[pl].y, vpl[pl].z, vpl[pl].w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int d = 8 * c + i;
                        const uint16_t hv = (uint16_t)((i & 1) ? (wvv[i >> 1] >> 16) : (wvv[i >> 1] & 0xffffu));
                        *reinterpret_cast<uint16_t*>(vt + pl * S::VT_BYTES + d * 128 + swz<8>(d, pos >> 3) * 16 + (pos & 7) * 2) = hv;
                    }
                }
            }
        }
        __syncthreads();

        // raw buffers are free again: next tile's DMA flies under this tile's MFMAs
        if (j + 1 < n_tiles) {
            stage_raw_tile<DHP, ESZ, DMA>(smem + S::OFF_RAWK, kg, k_rs, (j + 1) * BN, p.Tk, real_units, wave, lane);
            stage_raw_tile<DHP, ESZ, DMA>(smem + S::OFF_RAWV, vg, v_rs, (j + 1) * BN, p.Tk, real_units, wave, lane);
        }

        // ---- S^T = K' Q'^T ----
        f32x16_t s0, s1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
        {
            const char* kf = smem + S::OFF_KF;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int u = 2 * ks + lh;
                const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(kf + (l31 * CHP + swz<CHP>(l31, u)) * 16);
                const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(kf + ((32 + l31) * CHP + swz<CHP>(32 + l31, u)) * 16);
                if constexpr (X3) {      // small terms first: lo*hi + hi*lo + hi*hi
                    const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(kf + S::KF_BYTES + (l31 * CHP + swz<CHP>(l31, u)) * 16);
                    const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(kf + S::KF_BYTES + ((32 + l31) * CHP + swz<CHP>(32 + l31, u)) * 16);
                    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, qf[0][ks], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, qf[0][ks], s1, 0, 0, 0);
                    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qf[NPL - 1][ks], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, qf[NPL - 1][ks], s1, 0, 0, 0);
                }
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qf[0][ks], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, qf[0][ks], s1, 0, 0, 0);
            }
        }
        if (p.kbias) {     // additive per-key bias (euclid: -scale |k'|^2 / 2), given before the temperature
            const float bsc = LOG2E / (p.tau ? *p.tau : 1.0f);
            const float* kb = p.kbias + ((long)b * p.H + h) * p.kbias_pitch + j * BN + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(kb + 8 * g);
                const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(kb + 32 + 8 * g);
                s0[4 * g] += bsc * b0.x; s0[4 * g + 1] += bsc * b0.y; s0[4 * g + 2] += bsc * b0.z; s0[4 * g + 3] += bsc * b0.w;
                s1[4 * g] += bsc * b1.x; s1[4 * g + 1] += bsc * b1.y; s1[4 * g + 2] += bsc * b1.z; s1[4 * g + 3] += bsc * b1.w;
            }
        }
        // mask keys past Tk (last tile only): key = j*64 + 32*kb + (r&3) + 8*(r>>2) + 4*lh
        if (j == n_tiles - 1 && (p.Tk & (BN - 1))) {
            const int kbase = j * BN + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key >= p.Tk) s0[r] = -1e30f;
                if (key + 32 >= p.Tk) s1[r] = -1e30f;
            }
        }

        // ---- online softmax (S is already in log2 units) ----
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

        // P fragments: k-slot e of slab (kb,t) == accumulator register 8t+e (see header comment)
        bf16x8_t pf[NPL][2][2];
        {
            auto frag = [](const f32x16_t& sv, int t8) {
                u32x4_t w;
                w.x = pack_bf16x2(sv[t8 + 0], sv[t8 + 1]); w.y = pack_bf16x2(sv[t8 + 2], sv[t8 + 3]);
                w.z = pack_bf16x2(sv[t8 + 4], sv[t8 + 5]); w.w = pack_bf16x2(sv[t8 + 6], sv[t8 + 7]);
                return w;
            };
            const u32x4_t h00 = frag(s0, 0), h01 = frag(s0, 8), h10 = frag(s1, 0), h11 = frag(s1, 8);
            pf[0][0][0] = __builtin_bit_cast(bf16x8_t, h00); pf[0][0][1] = __builtin_bit_cast(bf16x8_t, h01);
            pf[0][1][0] = __builtin_bit_cast(bf16x8_t, h10); pf[0][1][1] = __builtin_bit_cast(bf16x8_t, h11);
            if constexpr (X3) {
                auto resid = [](f32x16_t& sv, const u32x4_t& h, int t8) {          // sv[t8..t8+7] -= float(hi)
                    float xh[8];
                    unpack8(h, xh);
#pragma unroll
                    for (int i = 0; i < 8; ++i) sv[t8 + i] -= xh[i];
                };
                resid(s0, h00, 0); resid(s0, h01, 8); resid(s1, h10, 0); resid(s1, h11, 8);
                pf[NPL - 1][0][0] = __builtin_bit_cast(bf16x8_t, frag(s0, 0)); pf[NPL - 1][0][1] = __builtin_bit_cast(bf16x8_t, frag(s0, 8));
                pf[NPL - 1][1][0] = __builtin_bit_cast(bf16x8_t, frag(s1, 0)); pf[NPL - 1][1][1] = __builtin_bit_cast(bf16x8_t, frag(s1, 8));
            }
        }

        // ---- O^T += V'^T P^T ----
        {
            const char* vt = smem + S::OFF_VT;
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const int row = 32 * d + l31;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(
                            vt + row * 128 + swz<8>(row, 4 * kb + 2 * t + lh) * 16);
                        if constexpr (X3) {
                            const bf16x8_t al = *reinterpret_cast<const bf16x8_t*>(
                                vt + S::VT_BYTES + row * 128 + swz<8>(row, 4 * kb + 2 * t + lh) * 16);
                            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, pf[0][kb][t], oacc[d], 0, 0, 0);
                            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[NPL - 1][kb][t], oacc[d], 0, 0, 0);
                        }
                        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[0][kb][t], oacc[d], 0, 0, 0);
                    }
            }
        }
    }

    // ---- epilogue ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv_l = 1.0f / l_tot;
    if (p.lse && lh == 0) {
        const int t = q0 + wave * 32 + l31;
        if (t < p.Tq) p.lse[((long)b * p.H + h) * p.Tq + t] = (m_run + __log2f(l_tot)) * LN2;
    }
    __syncthreads();   // every wave is done with K'/V'^T (and no DMA is in flight) -> reuse as O staging
    {
        float* ost = reinterpret_cast<float*>(smem + S::OFF_OST);
        const int r = wave * 32 + l31;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // accumulator registers 4g..4g+3 = channels 32d + 8g + 4lh + 0..3 of query row r
                const f32x4_t v = {oacc[d][4 * g] * inv_l, oacc[d][4 * g + 1] * inv_l,
                                   oacc[d][4 * g + 2] * inv_l, oacc[d][4 * g + 3] * inv_l};
                *reinterpret_cast<f32x4_t*>(ost + r * S::OROW + 32 * d + 8 * g + 4 * lh) = v;
            }
    }
    __syncthreads();
    {
        const float* ost = reinterpret_cast<const float*>(smem + S::OFF_OST);
        const bool xo = (p.flags & GTA_FLAG_V_TRANSFORM) != 0;
#pragma unroll
        for (int it = 0; it < 2 * CHP / 4; ++it) {
            const int c = wave + 4 * (it >> 1);
            const int r = lane + 64 * (it & 1);
            const int t = q0 + r;
            if (c < ch_real && t < p.Tq) {
                const uint32_t desc = p.ctab[c];
                float x[1][8];
                const f32x4_t a = *reinterpret_cast<const f32x4_t*>(ost + r * S::OROW + 8 * c);
                const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(ost + r * S::OROW + 8 * c + 4);
                x[0][0] = a.x; x[0][1] = a.y; x[0][2] = a.z; x[0][3] = a.w;
                x[0][4] = bb.x; x[0][5] = bb.y; x[0][6] = bb.z; x[0][7] = bb.w;
                if (xo && desc) {
                    const int n = view_of(t, p.Pq, p.invPq);
                    f32x2_t cs[4];
                    if (p.cs_q) load_cs(desc, p.cs_q + ((long)b * p.Tq + t) * 2 * p.nso2, cs);
                    const float* rec = qrec + n * GTA_QREC;
                    chunk_apply<true, 1>(desc, rec + GTA_QREC_O, rec + GTA_QREC_D1T, rec + GTA_QREC_D2T, cs, x);
                }
                gstore_chunk<ESZ>(og + (long)t * o_rs, c, x[0]);
            }
        }
    }
}

template <int DHP, int ESZ, bool DMA, bool X3 = false>
int launch(const GtaFwdParams& p, int n_wg, hipStream_t stream) {
    using S = Smem<DHP, ESZ, X3>;
    if (int rc = gta_lds_optin<&gta_fwd_kernel<DHP, ESZ, DMA, X3>>(S::TOTAL)) return rc;
    hipLaunchKernelGGL((gta_fwd_kernel<DHP, ESZ, DMA, X3>), dim3(n_wg), dim3(NTHREADS), S::TOTAL, stream, p);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

template <int DHP>
int lds_bytes_for(int esz) { return esz == 2 ? Smem<DHP, 2>::TOTAL : Smem<DHP, 4>::TOTAL; }

}  // namespace

int gta_fwd_lds_bytes(int dhp, int esz) {
    switch (dhp) {
        case 32: return lds_bytes_for<32>(esz);
        case 64: return lds_bytes_for<64>(esz);
        case 96: return lds_bytes_for<96>(esz);
        case 128: return lds_bytes_for<128>(esz);
    }
    return -1;
}

int gta_fwd_dispatch(const GtaFwdParams& p, int dhp, int esz, bool dma, int n_wg, hipStream_t stream) {
    const bool x3 = (p.flags & GTA_FLAG_FP32_PRODUCTS) != 0;
    if (x3 && esz != 4) return GTA_E_UNSUPPORTED;      // (the split serves fp32 inputs; bf16 inputs are bf16 arithmetic by request)
#define GTA_CASE(D)                                                                       \
    case D:                                                                               \
        if (x3) return launch<D, 4, true, true>(p, n_wg, stream);                         \
        if (esz == 2) return dma ? launch<D, 2, true>(p, n_wg, stream) : launch<D, 2, false>(p, n_wg, stream); \
        else          return dma ? launch<D, 4, true>(p, n_wg, stream) : launch<D, 4, false>(p, n_wg, stream);
    switch (dhp) {
        GTA_CASE(32)
        GTA_CASE(64)
        GTA_CASE(96)
        GTA_CASE(128)
    }
#undef GTA_CASE
    return GTA_E_UNSUPPORTED;
}
