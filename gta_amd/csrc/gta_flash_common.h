// gta_flash_common.h -- pieces shared by the two-stage forward kernels (gta_prep.hip, gta_fwd2.hip):
// LDS / transpose-read helpers, per-view record staging, the compile-time loop, tile constants and the
// ablation switches.  Everything has internal linkage (anonymous namespace): each .hip is its own module.
#pragma once
#include <type_traits>
#include <utility>
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
// One batch = three record elements per thread; split into its load and its store half so a caller can put other
// memory requests between them (the attention kernel issues its tile DMA there: off the prologue's critical path).
struct QrecBatch { float val[3]; int kind[3], rr[3], cc[3]; };
GTA_DEV void qrec_batch_load(QrecBatch& q, const float* vrep_q, int b, int Nq, int n0, int cnt, int i0, int nthreads) {
    const int total = cnt * GTA_QREC;
    const float* base = vrep_q + ((long)b * Nq + n0) * GTA_VREP_STRIDE;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int i = i0 + u * nthreads;
        q.kind[u] = 3; q.val[u] = 0.f; q.rr[u] = 0; q.cc[u] = 0;
        if (i < total) {
            const int n = i / GTA_QREC, e = i - n * GTA_QREC;
            q.val[u] = base[(long)n * GTA_VREP_STRIDE + qrec_src(e, &q.kind[u], &q.rr[u], &q.cc[u])];
        }
    }
}
GTA_DEV void qrec_batch_store(const QrecBatch& q, float* qrec, float tc, int i0, int nthreads) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int i = i0 + u * nthreads;
        if (q.kind[u] == 3) continue;
        float v = q.val[u];
        if (q.kind[u] == 0) v *= (q.rr[u] == 3) ? (q.cc[u] == 3 ? 1.f : 0.f) : (q.cc[u] == 3 ? tc : 1.f);
        else if (q.kind[u] == 2) v = 0.f;
        qrec[i] = v;
    }
}
// The same staging with the record cut into four segments, one per wave of a 256-thread workgroup, so that the kind
// of element a lane handles is WAVE-uniform (qrec_src above is a five-way per-lane branch: every path runs, masked):
//   wave 0: Aq, Oq (32 / view)   wave 1: D1, D1^T (24 / view)   wave 2: D2 (40 / view)   wave 3: D2^T (40 / view)
// idx = position in the wave's segment list (lane, lane + 64, ...).  dst < 0: nothing to do.
struct QrecItem { float val; int kind, rr, cc, dst; };
GTA_DEV void qrec_seg_load(QrecItem& q, const float* vrep_q, int b, int Nq, int n0, int cnt, int seg, int idx) {
    q.val = 0.f; q.kind = 1; q.rr = 0; q.cc = 0; q.dst = -1;
    const float* base = vrep_q + ((long)b * Nq + n0) * GTA_VREP_STRIDE;
    int n, e, src;
    if (seg == 0) {
        if (idx >= cnt * 32) return;
        n = idx >> 5;
        const int k = idx & 31, ee = k & 15, r = ee >> 2, c = ee & 3;
        const int sr = (k < 16) ? c : r, sc = (k < 16) ? r : c;          // Aq = (E.m)^T, Oq = E.m
        q.kind = 0; q.rr = sr; q.cc = sc;
        src = GTA_VREP_INV + sr * 4 + sc; e = k;
    } else if (seg == 1) {
        if (idx >= cnt * 24) return;
        n = idx / 24;
        const int k = idx - 24 * n, t = k >= 12, kk = t ? k - 12 : k, r = kk >> 2, c = kk & 3, cz = c < 3 ? c : 0;
        q.kind = c < 3 ? 1 : 2;
        src = GTA_VREP_D1 + (t ? cz * 3 + r : r * 3 + cz); e = (t ? GTA_QREC_D1T : GTA_QREC_D1) + kk;
    } else {
        if (idx >= cnt * 40) return;
        n = idx / 40;
        const int k = idx - 40 * n, r = k >> 3, c = k & 7, cz = c < 5 ? c : 0;
        q.kind = c < 5 ? 1 : 2;
        src = GTA_VREP_D2 + (seg == 3 ? cz * 5 + r : r * 5 + cz); e = (seg == 3 ? GTA_QREC_D2T : GTA_QREC_D2) + k;
    }
    q.dst = n * GTA_QREC + e;
    q.val = base[(long)n * GTA_VREP_STRIDE + src];
}
// fwd_scale multiplies the matrices that act on Q (Aq, D1q, D2q: record offsets below GTA_QREC_O, and D1 / D2), not the
// output-side ones (Oq, D1q^T, D2q^T): the attention kernel folds its scale * log2(e) / tau there.
GTA_DEV void qrec_seg_store(const QrecItem& q, float* qrec, float tc, float fwd_scale = 1.0f) {
    if (q.dst < 0) return;
    float v = q.val;
    if (q.kind == 0) v *= (q.rr == 3) ? (q.cc == 3 ? 1.f : 0.f) : (q.cc == 3 ? tc : 1.f);
    else if (q.kind == 2) v = 0.f;
    const int e = q.dst % GTA_QREC;
    if (e < GTA_QREC_O || (e >= GTA_QREC_D1 && e < GTA_QREC_D1T)) v *= fwd_scale;
    qrec[q.dst] = v;
}
// items a wave's segment list holds for cnt views
GTA_DEV int qrec_seg_count(int seg, int cnt) { return cnt * (seg == 0 ? 32 : seg == 1 ? 24 : 40); }
GTA_DEV void stage_qrec(float* qrec, const float* vrep_q, int b, int Nq, int n0, int cnt, float tc, int tid,
                        int nthreads, int first = 0) {
    const int total = cnt * GTA_QREC;
    for (int i0 = tid + first; i0 < total; i0 += 3 * nthreads) {
        QrecBatch q;
        qrec_batch_load(q, vrep_q, b, Nq, n0, cnt, i0, nthreads);
        qrec_batch_store(q, qrec, tc, i0, nthreads);
    }
}
// element i of the k-side records of scene b (record layout: gta_common.h) = vrep_k[krec_source(..., &m)] * m.
// Branch-free, and the load is the caller's, so a thread's several elements can be requested back to back.
GTA_DEV long krec_source(int b, int Nk, int i, float tc, float* m) {
    const int n = i / GTA_KREC, e = i - n * GTA_KREC;
    const bool se3 = e < GTA_KREC_D1, d1 = e < GTA_KREC_D2;
    const int ee = se3 ? e : d1 ? e - GTA_KREC_D1 : e - GTA_KREC_D2;
    const int r = (se3 || d1) ? ee >> 2 : ee >> 3, c = (se3 || d1) ? ee & 3 : ee & 7;
    const int idx = se3 ? GTA_VREP_REP + e : d1 ? GTA_VREP_D1 + r * 3 + (c < 3 ? c : 2) : GTA_VREP_D2 + r * 5 + (c < 5 ? c : 4);
    *m = se3 ? ((r == 3) ? (c == 3 ? 1.f : 0.f) : (c == 3 ? tc : 1.f)) : d1 ? (c < 3 ? 1.f : 0.f) : (c < 5 ? 1.f : 0.f);
    return ((long)b * Nk + n) * GTA_VREP_STRIDE + idx;
}
GTA_DEV void stage_krec(float* krec, const float* vrep_k, int b, int Nk, float tc, int tid, int nthreads) {
    for (int i = tid; i < Nk * GTA_KREC; i += nthreads) {
        float m;
        const long src = krec_source(b, Nk, i, tc, &m);
        krec[i] = vrep_k[src] * m;
    }
}

// ------------------------------------------------------------------------------------------------
// rho_q / rho_q^-1 on the matrix cores (gta_fwd64.hip, MSN layout dh = 96)
// ------------------------------------------------------------------------------------------------
// rho_q (gta.py:165,193) and rho_q^-1 (gta.py:255-271) are block-diagonal per-VIEW matrices on the se3 / so3 channels, so for
// the 32 query rows of one MFMA block -- all of one view -- they are two small GEMMs: Q'^T = Aq Q^T and O^T = Cq O~^T, 32 x 32
// output blocks that only see their own 32 input channels (no rep block straddles a multiple of 32).  The q-side matrices are
// expanded ONCE per (scene, view) into bf16 A-operand tiles [32 rows][16 k] (hi + lo parts: the products are exact to 2^-17),
// written by extra workgroups of the K/V pre-pass launch; the attention kernel's prologue / epilogue are then a few MFMAs and
// the per-token so2 rotations.  Tile t of a view:  t = 12*type + 6*part + 2*d + kk   type 0: Aq (forward, carries
// scale*log2e/tau), 1: Cq (inverse);  part 0: hi, 1: lo;  d: 32-channel output block;  kk: 16-channel k-step of that block.
// Rows and k-slots are PERMUTED so that no data moves between lanes around the MFMAs:
//   Aq tile: k-slot k  <-> input channel 32d + 16kk + k (the raw-Q B fragment as loaded);  row rho <-> output channel
//            gta_qt_chan_q(d, rho): a lane's 16 result registers are its two B fragments (chunks 4d + lh and 4d + 2 + lh) of Q'.
//   Cq tile: k-slot k  <-> input channel gta_qt_chan_o_in(d, kk, k): the O~ accumulator registers of a lane, packed pairwise, ARE
//            the B fragments;  row rho <-> output channel gta_qt_chan_o_out(d, rho): a lane ends with 16 contiguous channels.
// A tile is stored in fragment order: element (rho, k) at byte ((k >> 3) * 32 + rho) * 16 + (k & 7) * 2 (lane l of a wave reads
// 16 bytes at l * 16).
#define GTA_QT_TILES 24
#define GTA_QT_BYTES 1024
__host__ __device__ constexpr int gta_qt_chan_q(int d, int rho) {
    const int lh = (rho >> 2) & 1, j = rho >> 3, i = rho & 3;
    return j < 2 ? 32 * d + 8 * lh + 4 * j + i : 32 * d + 16 + 8 * lh + 4 * (j - 2) + i;
}
__host__ __device__ constexpr int gta_qt_chan_o_in(int d, int kk, int k) {
    const int lh = k >> 3, i = k & 7;
    return 32 * d + 16 * kk + (i & 3) + 8 * (i >> 2) + 4 * lh;
}
__host__ __device__ constexpr int gta_qt_chan_o_out(int d, int rho) {
    const int lh = (rho >> 2) & 1, r = (rho & 3) + 4 * (rho >> 3);
    return 32 * d + 16 * lh + r;
}
// entry (r, c) of the 96 x 96 q-side matrix of the MSN layout (se3 48 | so3 24 (L = 2) | so2 24) from a staged q-side
// record (gta_common.h: Aq / D1q / D2q carry the score scale, Oq / D1q^T / D2q^T do not); so2 channels: the identity
// (times the scale on the forward side) -- their per-token rotation stays on the VALU
GTA_DEV float gta_qt_entry_ms(const float* rec, int inverse, int r, int c, float fwd_scale) {
    if ((r >> 2) != (c >> 2) && (r < 48 || c < 48)) return 0.f;
    if (r < 48) return rec[(inverse ? GTA_QREC_O : GTA_QREC_A) + 4 * (r & 3) + (c & 3)];
    if (r >= 72 || c >= 72) return r == c ? (inverse ? 1.0f : fwd_scale) : 0.f;
    if (((r - 48) >> 3) != ((c - 48) >> 3)) return 0.f;
    const int er = (r - 48) & 7, ec = (c - 48) & 7;
    if (er < 3 && ec < 3) return rec[(inverse ? GTA_QREC_D1T : GTA_QREC_D1) + 4 * er + ec];
    if (er >= 3 && ec >= 3) return rec[(inverse ? GTA_QREC_D2T : GTA_QREC_D2) + 8 * (er - 3) + (ec - 3)];
    return 0.f;
}
// one workgroup (256 threads) expands the record of view (b, n) into its 24 tiles
GTA_DEV void gta_qt_build_view(char* tiles, const float* rec, float fwd_scale, int tid) {
    // 2 types x 3 blocks x 2 k-steps x 32 rows = 384 rows of 16 entries; hi and lo rows go to tiles t and t + 6
    for (int idx = tid; idx < 384; idx += 256) {
        const int rho = idx & 31, kk = (idx >> 5) & 1, d = (idx >> 6) % 3, type = idx / 192;
        const int r = type ? gta_qt_chan_o_out(d, rho) : gta_qt_chan_q(d, rho);
        uint32_t hi[8], lo[8];
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            float x[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 2 * k2 + e;
                const int c = type ? gta_qt_chan_o_in(d, kk, k) : 32 * d + 16 * kk + k;
                x[e] = gta_qt_entry_ms(rec, type, r, c, fwd_scale);
            }
            hi[k2] = pack_bf16x2(x[0], x[1]);
            lo[k2] = pack_bf16x2(x[0] - bf16_lo(hi[k2]), x[1] - bf16_hi(hi[k2]));
        }
        char* t_hi = tiles + (12 * type + 2 * d + kk) * GTA_QT_BYTES;
        char* t_lo = t_hi + 6 * GTA_QT_BYTES;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            *reinterpret_cast<u32x4_t*>(t_hi + (half * 32 + rho) * 16) = u32x4_t{hi[4 * half], hi[4 * half + 1], hi[4 * half + 2], hi[4 * half + 3]};
            *reinterpret_cast<u32x4_t*>(t_lo + (half * 32 + rho) * 16) = u32x4_t{lo[4 * half], lo[4 * half + 1], lo[4 * half + 2], lo[4 * half + 3]};
        }
    }
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <class F, int... Is>
GTA_DEV void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
GTA_DEV void static_for(F&& f) { static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

constexpr int NSTAGE = 3;                // ring stages (K' + V' tile images each); the skewed loop needs 3, the plain loop 2 or 3
// Skewed tile loop in the 128-row kernel (QK^T of tile j+1 beside the softmax of tile j), used at dh = 96.
// Measured on MI355X by CYCLE counts (tools/bench_kernels.py timeline on an instrumented build; MSN encoder, B=32):
// 362-366k shader cycles per launch against 396-406k un-skewed (tile loop 48.5k vs 52.7k cycles per workgroup), -9 %.
// In microseconds it first looked neutral (214 vs 212 us): the more efficient loop draws more power and is granted a
// lower clock (1.88 vs 1.92 GHz), and timings from separate launches differ by more than that anyway (the clock moves
// between 1.69 and 2.13 GHz, profiles/r01/README.md).  7 VGPRs spill outside the loop.  The dh <= 64 instances keep the
// plain loop (three workgroups per CU).
constexpr bool PIPE1 = true;
constexpr float BOUND_THR = 96.0f;     // exp2 arguments stay below this without looking at the scores

}  // namespace
