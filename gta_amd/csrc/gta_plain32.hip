// gta_plain32.hip -- EXACT-fp32 plain attention backward (identity layout), the gradient leg of the fp32-faithful mode.
//
// The reference's `mixed_prec: False` configs (runs/clevrtr/GTA/gta/config.yaml:55, source/trainer.py:106) run the operator and its
// autograd in true fp32.  The default kernels multiply on the bf16 matrix cores; the forward of the faithful mode keeps operands as
// bf16 hi + lo pairs (three MFMAs per product, gta_fwd.hip) -- and until r04 the backward stayed on bf16 products in both modes.
// This file is the backward of softmax(scale q' k'^T / tau) v' (source/layers.py:202-211 under autograd) on PRE-TRANSFORMED fp32
// tensors -- q' = rho_q q, k' = rho_k k, v' = rho_k v are produced by gta_rep_apply in fp32, the gradients go back through its adjoint
// gta_rep_apply_bwd (gta_apply.hip), so every layout of gta.py:92-279 (se3 / so3 / so2, t2, euclid) is served by this one pair of kernels:
//
//      P = exp(z - lse),  z = scale q'.k' / tau        dV' = P^T dO~        dP = dO~ V'^T
//      dS = P (dP - D),   D_i = <dO~_i, O~_i>          dQ' = (scale / tau) dS K'      dK' = (scale / tau) dS^T Q'
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- fp32 operands, fp32 accumulation, bit-identical to an fmaf chain (MI355X_MICROARCH.md: the
// f32-input matrix instruction runs at the fp32 VECTOR rate, 1/16 of the bf16 rate: this is the accuracy mode, not the fast one) and
// exp2f / fp32 VALU for the softmax.  Two kernels, no atomics (deterministic): dQ (128 query rows per workgroup, key tiles of 32
// streamed through LDS; it also writes D) and dK/dV (128 keys per workgroup, query tiles of 32 streamed).  Per 32 x 32 score block a
// wave recomputes S and dP and issues 1.5 dh (dQ) / 2 dh (dK/dV) matrix instructions.
#include "gta_common.h"
#include "../../include/gta_hip.h"

namespace {

constexpr float P32_LOG2E = 1.4426950408889634f;

struct Plain32Params {
    const float *q, *k, *v, *o, *dout, *lse, *tau;
    float *dq, *dk, *dv, *D;
    long q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st, do_sb, do_sh, do_st;
    long dq_sb, dq_sh, dq_st, dk_sb, dk_sh, dk_st, dv_sb, dv_sh, dv_st;
    int B, H, Tq, Tk, dh;
    float scale;
};

GTA_DEV f32x16_t mfma32(float a, float b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// rows of 32 x DHP fp32 tiles in LDS are DHP + 4 floats apart: the 16 lanes of a ds_read_b128 group then cover all 64 banks
template <int DHP> struct P32 { static constexpr int HC = DHP / 2, ROW = DHP + 4, NB = DHP / 32; };

// rows [t0, t0 + 32) of a [T][dh] fp32 matrix (row stride st) as a 32 x DHP tile in LDS (rows past T repeat the last one, channels past dh
// are zero), 256 threads, in two halves: the loads into registers (tile_fetch: issued BEFORE the previous tile's arithmetic, so their latency
// lies under it -- the first form loaded and wrote a tile between two barriers and ran at 52 % of the fp32 matrix rate) and the LDS writes
// (tile_put, between the barriers)
template <int DHP> struct TileRegs { f32x4_t x[(32 * (DHP / 4) + 255) / 256]; };
template <int DHP>
GTA_DEV void tile_fetch(TileRegs<DHP>& R, const float* src, long st, int t0, int T, int dh, int tid) {
    constexpr int Q4 = DHP / 4, N = (32 * Q4 + 255) / 256;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const int i = tid + 256 * n;
        const int r = i / Q4, c4 = i - r * Q4;
        int t = t0 + r;
        t = t < T ? t : T - 1;
        R.x[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (i < 32 * Q4 && 4 * c4 < dh) R.x[n] = *reinterpret_cast<const f32x4_t*>(src + (long)t * st + 4 * c4);
    }
}
template <int DHP>
GTA_DEV void tile_put(float* dst, const TileRegs<DHP>& R, int tid) {
    constexpr int Q4 = DHP / 4, N = (32 * Q4 + 255) / 256;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const int i = tid + 256 * n;
        const int r = i / Q4, c4 = i - r * Q4;
        if (i < 32 * Q4) *reinterpret_cast<f32x4_t*>(dst + r * P32<DHP>::ROW + 4 * c4) = R.x[n];
    }
}

// this lane's half of row `row` of a [T][dh] matrix: channels kk * HC + c, c < HC (zeros past dh)
template <int DHP>
GTA_DEV void load_half_row(float* x, const float* rowptr, int kk, int dh) {
    constexpr int HC = P32<DHP>::HC;
#pragma unroll
    for (int c4 = 0; c4 < HC / 4; ++c4) {
        f32x4_t t = {0.f, 0.f, 0.f, 0.f};
        if (kk * HC + 4 * c4 < dh) t = *reinterpret_cast<const f32x4_t*>(rowptr + kk * HC + 4 * c4);
        x[4 * c4] = t.x; x[4 * c4 + 1] = t.y; x[4 * c4 + 2] = t.z; x[4 * c4 + 3] = t.w;
    }
}

// acc (32 x 32) += Tile (32 rows x DHP, LDS) . X^T with X's row of this lane in registers (xr: its half of the channels):
// A[i][kk] = tile[i][kk HC + c], B[kk][j] = x_j[kk HC + c]
template <int DHP>
GTA_DEV f32x16_t tile_times_rows(const float* tile, const float* xr, int j, int kk, f32x16_t acc) {
    constexpr int HC = P32<DHP>::HC, ROW = P32<DHP>::ROW;
#pragma unroll
    for (int c4 = 0; c4 < HC / 4; ++c4) {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(tile + j * ROW + kk * HC + 4 * c4);
        acc = mfma32(a.x, xr[4 * c4], acc);
        acc = mfma32(a.y, xr[4 * c4 + 1], acc);
        acc = mfma32(a.z, xr[4 * c4 + 2], acc);
        acc = mfma32(a.w, xr[4 * c4 + 3], acc);
    }
    return acc;
}

// out[d] (32 channels x 32 columns) += Tile^T . W, W (32 tile rows x 32 columns) being an accumulator of the MFMA's own D layout:
// register r of lane (j, kk) = W[(r & 3) + 8 (r >> 2) + 4 kk][j];  A[i][kk] = tile[that row][32 d + i]
template <int DHP>
GTA_DEV void tile_t_times_acc(const float* tile, const f32x16_t& w, int j, int kk, f32x16_t* out) {
    constexpr int ROW = P32<DHP>::ROW, NB = P32<DHP>::NB;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float* row = tile + ((r & 3) + 8 * (r >> 2) + 4 * kk) * ROW + j;
#pragma unroll
        for (int d = 0; d < NB; ++d) out[d] = mfma32(row[32 * d], w[r], out[d]);
    }
}

template <int DHP>
__global__ __launch_bounds__(256) void plain32_dq_kernel(const Plain32Params p) {
    constexpr int HC = P32<DHP>::HC, ROW = P32<DHP>::ROW, NB = P32<DHP>::NB;
    __shared__ __attribute__((aligned(16))) float ks[32 * ROW];
    __shared__ __attribute__((aligned(16))) float vs[32 * ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kk = lane >> 5;
    const int nq = (p.Tq + 127) / 128;
    const int bh = blockIdx.x / nq, qt = blockIdx.x - bh * nq, b = bh / p.H, h = bh - b * p.H;
    const int row = qt * 128 + wave * 32 + j, rowc = row < p.Tq ? row : p.Tq - 1;
    float qf[HC], df[HC];
    load_half_row<DHP>(qf, p.q + (long)b * p.q_sb + (long)h * p.q_sh + (long)rowc * p.q_st, kk, p.dh);
    load_half_row<DHP>(df, p.dout + (long)b * p.do_sb + (long)h * p.do_sh + (long)rowc * p.do_st, kk, p.dh);
    float Drow;
    {
        float of[HC];
        load_half_row<DHP>(of, p.o + (long)b * p.o_sb + (long)h * p.o_sh + (long)rowc * p.o_st, kk, p.dh);
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < HC; ++c) part = fmaf(df[c], of[c], part);
        Drow = part + __shfl_xor(part, 32);
        if (kk == 0 && row < p.Tq) p.D[((long)b * p.H + h) * p.Tq + row] = Drow;
    }
    const float inv_tau = p.tau ? 1.0f / *p.tau : 1.0f;
    const float sl2 = p.scale * P32_LOG2E * inv_tau;
    const float lse2 = p.lse[((long)b * p.H + h) * p.Tq + rowc] * P32_LOG2E;
    const float* kb = p.k + (long)b * p.k_sb + (long)h * p.k_sh;
    const float* vb = p.v + (long)b * p.v_sb + (long)h * p.v_sh;
    f32x16_t dq[NB];
#pragma unroll
    for (int d = 0; d < NB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[d][i] = 0.f;
    TileRegs<DHP> rk, rv;
    tile_fetch<DHP>(rk, kb, p.k_st, 0, p.Tk, p.dh, tid);
    tile_fetch<DHP>(rv, vb, p.v_st, 0, p.Tk, p.dh, tid);
    for (int k0 = 0; k0 < p.Tk; k0 += 32) {
        __syncthreads();                                         // (the previous tile's arithmetic is through with the LDS tiles)
        tile_put<DHP>(ks, rk, tid);
        tile_put<DHP>(vs, rv, tid);
        __syncthreads();
        if (k0 + 32 < p.Tk) {                                    // the next tile's rows: in flight under this tile's arithmetic
            tile_fetch<DHP>(rk, kb, p.k_st, k0 + 32, p.Tk, p.dh, tid);
            tile_fetch<DHP>(rv, vb, p.v_st, k0 + 32, p.Tk, p.dh, tid);
        }
        f32x16_t s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
        s = tile_times_rows<DHP>(ks, qf, j, kk, s);              // S^T (keys x queries)
        dp = tile_times_rows<DHP>(vs, df, j, kk, dp);            // dP^T = V' dO~^T
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            const float pv = key < p.Tk ? exp2f(fmaf(s[r], sl2, -lse2)) : 0.f;
            s[r] = pv * (dp[r] - Drow);                          // dS^T
        }
        tile_t_times_acc<DHP>(ks, s, j, kk, dq);                 // dQ'^T += K'^T dS^T
    }
    if (row < p.Tq) {
        float* out = p.dq + (long)b * p.dq_sb + (long)h * p.dq_sh + (long)row * p.dq_st;
        const float g = p.scale * inv_tau;
#pragma unroll
        for (int d = 0; d < NB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * d + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (c < p.dh) out[c] = dq[d][r] * g;
            }
    }
}

template <int DHP>
__global__ __launch_bounds__(256) void plain32_dkv_kernel(const Plain32Params p) {
    constexpr int HC = P32<DHP>::HC, ROW = P32<DHP>::ROW, NB = P32<DHP>::NB;
    __shared__ __attribute__((aligned(16))) float qs[32 * ROW];
    __shared__ __attribute__((aligned(16))) float ds_[32 * ROW];
    __shared__ __attribute__((aligned(16))) float st[64];        // lse log2(e) | D of the tile's 32 query rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kk = lane >> 5;
    const int nk = (p.Tk + 127) / 128;
    const int bh = blockIdx.x / nk, kt = blockIdx.x - bh * nk, b = bh / p.H, h = bh - b * p.H;
    const int key = kt * 128 + wave * 32 + j, keyc = key < p.Tk ? key : p.Tk - 1;
    float kf[HC], vf[HC];
    load_half_row<DHP>(kf, p.k + (long)b * p.k_sb + (long)h * p.k_sh + (long)keyc * p.k_st, kk, p.dh);
    load_half_row<DHP>(vf, p.v + (long)b * p.v_sb + (long)h * p.v_sh + (long)keyc * p.v_st, kk, p.dh);
    const float inv_tau = p.tau ? 1.0f / *p.tau : 1.0f;
    const float sl2 = p.scale * P32_LOG2E * inv_tau;
    const float* qb = p.q + (long)b * p.q_sb + (long)h * p.q_sh;
    const float* db = p.dout + (long)b * p.do_sb + (long)h * p.do_sh;
    const float* lb = p.lse + ((long)b * p.H + h) * p.Tq;
    const float* Db = p.D + ((long)b * p.H + h) * p.Tq;
    f32x16_t dk[NB], dv[NB];
#pragma unroll
    for (int d = 0; d < NB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[d][i] = 0.f; dv[d][i] = 0.f; }
    TileRegs<DHP> rq, rd;
    float rs = 0.f;
    auto fetch = [&](int q0) {
        tile_fetch<DHP>(rq, qb, p.q_st, q0, p.Tq, p.dh, tid);
        tile_fetch<DHP>(rd, db, p.do_st, q0, p.Tq, p.dh, tid);
        if (tid < 64) {
            const int t = q0 + (tid & 31), tc = t < p.Tq ? t : p.Tq - 1;
            rs = tid < 32 ? lb[tc] * P32_LOG2E : Db[tc];
        }
    };
    fetch(0);
    for (int q0 = 0; q0 < p.Tq; q0 += 32) {
        __syncthreads();                                         // (the previous tile's arithmetic is through with the LDS tiles)
        tile_put<DHP>(qs, rq, tid);
        tile_put<DHP>(ds_, rd, tid);
        if (tid < 64) st[tid] = rs;
        __syncthreads();
        if (q0 + 32 < p.Tq) fetch(q0 + 32);                      // the next tile's rows: in flight under this tile's arithmetic
        f32x16_t s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
        s = tile_times_rows<DHP>(qs, kf, j, kk, s);              // S (queries x keys)
        dp = tile_times_rows<DHP>(ds_, vf, j, kk, dp);           // dP = dO~ V'^T
        f32x16_t pr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(st + 8 * g + 4 * kk);
            const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(st + 32 + 8 * g + 4 * kk);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i, t = q0 + 8 * g + 4 * kk + i;
                const float pv = (t < p.Tq && key < p.Tk) ? exp2f(fmaf(s[r], sl2, -l4[i])) : 0.f;
                pr[r] = pv;
                s[r] = pv * (dp[r] - d4[i]);                     // dS
            }
        }
        tile_t_times_acc<DHP>(ds_, pr, j, kk, dv);               // dV'^T += dO~^T P
        tile_t_times_acc<DHP>(qs, s, j, kk, dk);                 // dK'^T += Q'^T dS
    }
    if (key < p.Tk) {
        float* ok = p.dk + (long)b * p.dk_sb + (long)h * p.dk_sh + (long)key * p.dk_st;
        float* ov = p.dv + (long)b * p.dv_sb + (long)h * p.dv_sh + (long)key * p.dv_st;
        const float g = p.scale * inv_tau;
#pragma unroll
        for (int d = 0; d < NB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * d + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (c < p.dh) { ok[c] = dk[d][r] * g; ov[c] = dv[d][r]; }
            }
    }
}

template <int DHP>
int run_plain32(const Plain32Params& p, hipStream_t stream) {
    const long n_dq = (long)p.B * p.H * ((p.Tq + 127) / 128), n_dkv = (long)p.B * p.H * ((p.Tk + 127) / 128);
    if (n_dq > 0x7fffffffL || n_dkv > 0x7fffffffL) return GTA_E_UNSUPPORTED;
    hipLaunchKernelGGL(plain32_dq_kernel<DHP>, dim3((unsigned)n_dq), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(plain32_dkv_kernel<DHP>, dim3((unsigned)n_dkv), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

}  // namespace

extern "C" int64_t gta_attn_bwd_plain_f32_workspace_bytes(const GtaAttnDesc* d) {
    if (!d || d->B <= 0 || d->H <= 0 || d->Tq <= 0) return 0;
    return (int64_t)d->B * d->H * d->Tq * 4;
}

extern "C" int gta_attn_bwd_plain_f32(const GtaAttnDesc* d, const void* q, const void* k, const void* v, const void* out,
                                      const void* dout, const int64_t* dout_stride, const float* lse, const float* tau,
                                      void* dq, void* dk, void* dv, const int64_t* dqkv_stride, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
    if (!d || !q || !k || !v || !out || !dout || !dout_stride || !lse || !dq || !dk || !dv || !dqkv_stride || !workspace) return GTA_E_BADARG;
    if (d->abi_version != GTA_ABI_VERSION || d->dtype != GTA_DTYPE_F32) return GTA_E_BADARG;
    if (d->B <= 0 || d->H <= 0 || d->Tq <= 0 || d->Tk <= 0 || d->dh <= 0) return GTA_E_BADARG;
    if (d->dh % 8 || d->dh > 128) return GTA_E_UNSUPPORTED;                 // (the generic path pads the channels to a multiple of 8)
    if (workspace_bytes < gta_attn_bwd_plain_f32_workspace_bytes(d)) return GTA_E_BADARG;
    const int64_t* st[4] = {d->q_stride, d->k_stride, d->v_stride, d->o_stride};
    for (int i = 0; i < 4; ++i)
        for (int jx = 0; jx < 3; ++jx)
            if (st[i][jx] % 4) return GTA_E_BADARG;                         // 16-byte rows
    for (int i = 0; i < 9; ++i) if (dqkv_stride[i] % 4) return GTA_E_BADARG;
    for (int i = 0; i < 3; ++i) if (dout_stride[i] % 4) return GTA_E_BADARG;
    Plain32Params p;
    p.q = (const float*)q; p.k = (const float*)k; p.v = (const float*)v; p.o = (const float*)out; p.dout = (const float*)dout;
    p.lse = lse; p.tau = tau; p.dq = (float*)dq; p.dk = (float*)dk; p.dv = (float*)dv; p.D = (float*)workspace;
    p.q_sb = d->q_stride[0]; p.q_sh = d->q_stride[1]; p.q_st = d->q_stride[2];
    p.k_sb = d->k_stride[0]; p.k_sh = d->k_stride[1]; p.k_st = d->k_stride[2];
    p.v_sb = d->v_stride[0]; p.v_sh = d->v_stride[1]; p.v_st = d->v_stride[2];
    p.o_sb = d->o_stride[0]; p.o_sh = d->o_stride[1]; p.o_st = d->o_stride[2];
    p.do_sb = dout_stride[0]; p.do_sh = dout_stride[1]; p.do_st = dout_stride[2];
    p.dq_sb = dqkv_stride[0]; p.dq_sh = dqkv_stride[1]; p.dq_st = dqkv_stride[2];
    p.dk_sb = dqkv_stride[3]; p.dk_sh = dqkv_stride[4]; p.dk_st = dqkv_stride[5];
    p.dv_sb = dqkv_stride[6]; p.dv_sh = dqkv_stride[7]; p.dv_st = dqkv_stride[8];
    p.B = d->B; p.H = d->H; p.Tq = d->Tq; p.Tk = d->Tk; p.dh = d->dh; p.scale = d->scale;
    const int dhp = d->dh <= 32 ? 32 : d->dh <= 64 ? 64 : d->dh <= 96 ? 96 : 128;
    switch (dhp) {
        case 32: return run_plain32<32>(p, (hipStream_t)stream);
        case 64: return run_plain32<64>(p, (hipStream_t)stream);
        case 96: return run_plain32<96>(p, (hipStream_t)stream);
        default: return run_plain32<128>(p, (hipStream_t)stream);
    }
}
