// gta_block.hip -- the row kernels of the fused Transformer block (include/gta_block.h): LayerNorm forward / backward,
// exact GELU forward / backward, column sums.  All of them are HBM-bound: one pass over their operands, 16-byte accesses,
// fp32 arithmetic, no re-reads.  gfx950 only.
//
// Reference semantics: nn.LayerNorm (source/layers.py:149), nn.GELU (source/layers.py:162), the `+ x` skip connections
// of Transformer.forward (source/layers.py:483-487) whose gradient the LayerNorm backward folds in.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/gta_hip.h"
#include "../../include/gta_block.h"

#define BLK_DEV __device__ __forceinline__

namespace {

BLK_DEV float bf16_to_f(uint32_t h) { return __uint_as_float(h << 16); }
BLK_DEV uint32_t f_to_bf16(float f) {            // round to nearest even (inputs are finite)
    uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// eight consecutive elements of a row <-> eight floats, for both element types
template <int DT> struct Vec8;
template <> struct Vec8<GTA_DTYPE_F32> {
    static BLK_DEV void load(const void* p, int64_t e, float (&v)[8]) {
        const float4* q = reinterpret_cast<const float4*>(static_cast<const float*>(p) + e);
        float4 a = q[0], b = q[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static BLK_DEV void store(void* p, int64_t e, const float (&v)[8]) {
        float4* q = reinterpret_cast<float4*>(static_cast<float*>(p) + e);
        q[0] = make_float4(v[0], v[1], v[2], v[3]);
        q[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct Vec8<GTA_DTYPE_BF16> {
    static BLK_DEV void load(const void* p, int64_t e, float (&v)[8]) {
        uint4 a = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p) + e);
        uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = bf16_to_f(w[i] & 0xffffu); v[2 * i + 1] = bf16_to_f(w[i] >> 16); }
    }
    static BLK_DEV void store(void* p, int64_t e, const float (&v)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = f_to_bf16(v[2 * i]) | (f_to_bf16(v[2 * i + 1]) << 16);
        *reinterpret_cast<uint4*>(static_cast<uint16_t*>(p) + e) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// VW consecutive elements of a row <-> VW floats: VW = 8 (16-byte accesses of bf16, rows that are multiples of 8 elements) or
// VW = 4 (8-byte accesses of bf16 / 16-byte of fp32, rows that are multiples of 4: the MSN decoder's d = 180)
template <int DT, int VW> struct VecN;
template <int DT> struct VecN<DT, 8> {
    static BLK_DEV void load(const void* p, int64_t e, float (&v)[8]) { Vec8<DT>::load(p, e, v); }
    static BLK_DEV void store(void* p, int64_t e, const float (&v)[8]) { Vec8<DT>::store(p, e, v); }
};
template <> struct VecN<GTA_DTYPE_F32, 4> {
    static BLK_DEV void load(const void* p, int64_t e, float (&v)[4]) {
        const float4 a = *reinterpret_cast<const float4*>(static_cast<const float*>(p) + e);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    }
    static BLK_DEV void store(void* p, int64_t e, const float (&v)[4]) {
        *reinterpret_cast<float4*>(static_cast<float*>(p) + e) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct VecN<GTA_DTYPE_BF16, 4> {
    static BLK_DEV void load(const void* p, int64_t e, float (&v)[4]) {
        const uint2 a = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(p) + e);
        v[0] = bf16_to_f(a.x & 0xffffu); v[1] = bf16_to_f(a.x >> 16); v[2] = bf16_to_f(a.y & 0xffffu); v[3] = bf16_to_f(a.y >> 16);
    }
    static BLK_DEV void store(void* p, int64_t e, const float (&v)[4]) {
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p) + e) =
            make_uint2(f_to_bf16(v[0]) | (f_to_bf16(v[1]) << 16), f_to_bf16(v[2]) | (f_to_bf16(v[3]) << 16));
    }
};

BLK_DEV float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm forward: one 64-lane wave per row, the row held in registers (NJ chunks of 8 elements per lane, d <= 512 NJ).
// ---------------------------------------------------------------------------------------------------------------
template <int XDT, int YDT, int NJ, int VW>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, int64_t rows, int d,
                                                     void* __restrict__ y, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t base = row * d;
    float v[NJ][VW];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e = (lane + 64 * j) * VW;
        if (e < d) {
            VecN<XDT, VW>::load(x, base + e, v[j]);
#pragma unroll
            for (int i = 0; i < VW; ++i) s += v[j][i];
        }
    }
    const float inv_d = 1.0f / (float)d;
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e = (lane + 64 * j) * VW;
        if (e < d) {
#pragma unroll
            for (int i = 0; i < VW; ++i) { const float c = v[j][i] - mean; q += c * c; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e = (lane + 64 * j) * VW;
        if (e < d) {
            float g[VW], b[VW], o[VW];
            VecN<GTA_DTYPE_F32, VW>::load(gamma, e, g);
            VecN<GTA_DTYPE_F32, VW>::load(beta, e, b);
#pragma unroll
            for (int i = 0; i < VW; ++i) o[i] = (v[j][i] - mean) * rstd * g[i] + b[i];
            VecN<YDT, VW>::store(y, base + e, o);
        }
    }
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm backward.  Grid of G workgroups of 4 waves; wave w of workgroup g walks rows (4g + w), + 4G, ...
// dgamma / dbeta partial sums live in registers across the walk, are reduced over the 4 waves through LDS and written
// to part[g][2][d]; colsum_finish_kernel adds the G partials in a fixed order.
// ---------------------------------------------------------------------------------------------------------------
template <int GDT, int XDT, int NJ, int VW>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, int64_t rows, int d,
                                                     const void* dres, void* dx, void* dx_bf16, float* __restrict__ part) {
    extern __shared__ float lds[];            // [4][2][d]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ag[NJ][VW], ab[NJ][VW], gm[NJ][VW];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e = (lane + 64 * j) * VW;
        if (e < d) VecN<GTA_DTYPE_F32, VW>::load(gamma, e, gm[j]);
#pragma unroll
        for (int i = 0; i < VW; ++i) { ag[j][i] = 0.f; ab[j][i] = 0.f; }
    }
    const float inv_d = 1.0f / (float)d;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int64_t base = row * d;
        const float mu = mean[row], rs = rstd[row];
        float g[NJ][VW], xh[NJ][VW];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = (lane + 64 * j) * VW;
            if (e < d) {
                float dyv[VW], xv[VW];
                VecN<GDT, VW>::load(dy, base + e, dyv);
                VecN<XDT, VW>::load(x, base + e, xv);
#pragma unroll
                for (int i = 0; i < VW; ++i) {
                    xh[j][i] = (xv[i] - mu) * rs;
                    g[j][i] = dyv[i] * gm[j][i];
                    s1 += g[j][i];
                    s2 += g[j][i] * xh[j][i];
                    ag[j][i] += dyv[i] * xh[j][i];
                    ab[j][i] += dyv[i];
                }
            }
        }
        const float m1 = wave_sum(s1) * inv_d, m2 = wave_sum(s2) * inv_d;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = (lane + 64 * j) * VW;
            if (e < d) {
                float o[VW];
                if (dres) VecN<XDT, VW>::load(dres, base + e, o);
                else {
#pragma unroll
                    for (int i = 0; i < VW; ++i) o[i] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < VW; ++i) o[i] += rs * (g[j][i] - m1 - xh[j][i] * m2);
                VecN<XDT, VW>::store(dx, base + e, o);
                if (dx_bf16) VecN<GTA_DTYPE_BF16, VW>::store(dx_bf16, base + e, o);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e = (lane + 64 * j) * VW;
        if (e < d) {
#pragma unroll
            for (int i = 0; i < VW; ++i) {
                lds[(wave * 2 + 0) * d + e + i] = ag[j][i];
                lds[(wave * 2 + 1) * d + e + i] = ab[j][i];
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * d; c += 256)
        part[(int64_t)blockIdx.x * 2 * d + c] = (lds[c] + lds[2 * d + c]) + (lds[4 * d + c] + lds[6 * d + c]);
}

// out[c] = sum_p part[p][c], p in a fixed order: 1024 threads = 32 row groups x 32 columns, tree over the groups.
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ part, int np, int n,
                                                            float* __restrict__ out0, float* __restrict__ out1, int split) {
    __shared__ float red[32][33];
    const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < n) {
        int p = rg;
        for (; p + 96 < np; p += 128) {                                   // four independent loads in flight
            s0 += part[(int64_t)p * n + c];
            s1 += part[(int64_t)(p + 32) * n + c];
            s2 += part[(int64_t)(p + 64) * n + c];
            s3 += part[(int64_t)(p + 96) * n + c];
        }
        for (; p < np; p += 32) s0 += part[(int64_t)p * n + c];
    }
    red[rg][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int w = 16; w >= 1; w >>= 1) {
        if (rg < w) red[rg][cx] += red[rg + w][cx];
        __syncthreads();
    }
    if (rg == 0 && c < n) {
        const float t = red[0][cx];
        if (c < split) out0[c] = t; else out1[c - split] = t;
    }
}

// Column sums, stage 1: grid (ceil(n / 512), S strips); wave w of a workgroup walks rows (4 s + w), + 4 S, ...; a lane
// owns 8 columns.  part[s][n].
template <int DT, int VW>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ a, int64_t m, int n, int64_t ld,
                                                     float* __restrict__ part) {
    __shared__ float lds[4][64 * VW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = (blockIdx.x * 64 + lane) * VW;
    float acc[VW];
#pragma unroll
    for (int i = 0; i < VW; ++i) acc[i] = 0.f;
    if (e < n)
        for (int64_t row = (int64_t)blockIdx.y * 4 + wave; row < m; row += (int64_t)gridDim.y * 4) {
            float v[VW];
            VecN<DT, VW>::load(a, row * ld + e, v);
#pragma unroll
            for (int i = 0; i < VW; ++i) acc[i] += v[i];
        }
#pragma unroll
    for (int i = 0; i < VW; ++i) lds[wave][lane * VW + i] = acc[i];
    __syncthreads();
    for (int c = threadIdx.x; c < 64 * VW; c += 256) {
        const int col = blockIdx.x * 64 * VW + c;
        if (col < n) part[(int64_t)blockIdx.y * n + col] = (lds[0][c] + lds[1][c]) + (lds[2][c] + lds[3][c]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// exact GELU (erf form)
// ---------------------------------------------------------------------------------------------------------------
BLK_DEV float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
BLK_DEV float dgelu_erf(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// ---------------------------------------------------------------------------------------------------------------
// Dropout (nn.Dropout of to_out / FeedForward.net, source/layers.py:163,165,289).  The keep mask is never stored: it is a
// pure function of (seed, element index) -- Philox4x32-10 keyed by the call's seed, counter = index / 4 -- regenerated by
// the backward.  keep <=> random word >= p * 2^32; kept values are scaled by 1 / (1 - p).
// ---------------------------------------------------------------------------------------------------------------
struct Drop {
    uint32_t k0, k1, thr;
    float scale;
};

BLK_DEV void philox4(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t (&r)[4]) {
    uint32_t c2 = 0u, c3 = 0u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// keep factors (0 or 1/(1-p)) of the eight elements 8*i8 .. 8*i8+7
BLK_DEV void keep8(const Drop& d, int64_t i8, float (&m)[8]) {
    uint32_t r[4];
    philox4((uint32_t)(2 * i8), (uint32_t)((2 * i8) >> 32), d.k0, d.k1, r);
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = r[j] >= d.thr ? d.scale : 0.f;
    philox4((uint32_t)(2 * i8 + 1), (uint32_t)((2 * i8 + 1) >> 32), d.k0, d.k1, r);
#pragma unroll
    for (int j = 0; j < 4; ++j) m[4 + j] = r[j] >= d.thr ? d.scale : 0.f;
}

template <int DT, bool DROP>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const void* __restrict__ x, void* __restrict__ y, int64_t n8, Drop d) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8];
        Vec8<DT>::load(x, i * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = gelu_erf(v[k]);
        if (DROP) {
            float m[8];
            keep8(d, i, m);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= m[k];
        }
        Vec8<DT>::store(y, i * 8, v);
    }
}

template <int DT, bool DROP>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                       void* __restrict__ dx, int64_t n8, Drop d) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8], g[8];
        Vec8<DT>::load(x, i * 8, v);
        Vec8<DT>::load(dy, i * 8, g);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] *= dgelu_erf(v[k]);
        if (DROP) {
            float m[8];
            keep8(d, i, m);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] *= m[k];
        }
        Vec8<DT>::store(dx, i * 8, g);
    }
}

// out = skip + keep * z / (1 - p)      (z: the GEMM output with its bias; out, skip: the residual stream)
template <int ZDT, int SDT>
__global__ __launch_bounds__(256) void dropout_add_kernel(const void* __restrict__ z, const void* __restrict__ skip,
                                                          void* __restrict__ out, int64_t n8, Drop d) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8], s[8], m[8];
        Vec8<ZDT>::load(z, i * 8, v);
        Vec8<SDT>::load(skip, i * 8, s);
        keep8(d, i, m);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += v[k] * m[k];
        Vec8<SDT>::store(out, i * 8, s);
    }
}

// dz = keep * dout / (1 - p), written in the GEMMs' dtype
template <int GDT, int ZDT>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const void* __restrict__ dout, void* __restrict__ dz, int64_t n8, Drop d) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float g[8], m[8];
        Vec8<GDT>::load(dout, i * 8, g);
        keep8(d, i, m);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] *= m[k];
        Vec8<ZDT>::store(dz, i * 8, g);
    }
}

inline bool make_drop(float p, uint64_t seed, Drop* d) {
    if (!(p >= 0.f) || p >= 1.f) return false;
    d->k0 = (uint32_t)seed;
    d->k1 = (uint32_t)(seed >> 32);
    const double t = (double)p * 4294967296.0;
    d->thr = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    d->scale = 1.0f / (1.0f - p);
    return true;
}
inline unsigned ew_grid(int64_t n8) {
    const int64_t want = (n8 + 255) / 256;
    return (unsigned)(want < 8192 ? want : 8192);
}

inline bool dtype_ok(int dt) { return dt == GTA_DTYPE_F32 || dt == GTA_DTYPE_BF16; }
inline int launch_status() { return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int LN_BWD_MAX_WG = 1024;      // 4 workgroups of 4 waves per CU
inline int ln_bwd_grid(int64_t rows) {
    int64_t g = (rows + 3) / 4;
    return (int)(g < LN_BWD_MAX_WG ? g : LN_BWD_MAX_WG);
}
constexpr int COLSUM_MAX_STRIPS = 512;
inline int colsum_strips(int64_t m, int n) {
    const int64_t per = (n + 511) / 512;                         // workgroups per strip
    int64_t cap = COLSUM_MAX_STRIPS / per;
    if (cap < 8) cap = 8;
    int64_t s = (m + 63) / 64;                                   // at least 16 rows per wave
    if (s > cap) s = cap;
    return (int)(s < 1 ? 1 : s);
}

// dispatch over the vector width (8 elements per lane and chunk when the row length allows, else 4) and the
// chunks-per-lane count: d <= 64 * VW * NJ
template <class F> inline int by_row(int d, F&& f) {
    if (d % 8 == 0) {
        if (d <= 512) return f(std::integral_constant<int, 1>{}, std::integral_constant<int, 8>{});
        if (d <= 1024) return f(std::integral_constant<int, 2>{}, std::integral_constant<int, 8>{});
        if (d <= 2048) return f(std::integral_constant<int, 4>{}, std::integral_constant<int, 8>{});
        return f(std::integral_constant<int, 8>{}, std::integral_constant<int, 8>{});
    }
    if (d <= 256) return f(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
    if (d <= 512) return f(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
    if (d <= 1024) return f(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
    return f(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
}
inline bool row_ok(int d) { return d > 0 && d % 4 == 0 && (d % 8 == 0 ? d <= 4096 : d <= 2048); }
inline bool aligned_for(const void* p, int d, int dtype) {       // base alignment the row accesses need
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    return (d % 8 == 0 || dtype == GTA_DTYPE_F32) ? (a & 15u) == 0 : (a & 7u) == 0;
}
template <class F> inline int by_dtype(int dt, F&& f) {
    return dt == GTA_DTYPE_F32 ? f(std::integral_constant<int, GTA_DTYPE_F32>{}) : f(std::integral_constant<int, GTA_DTYPE_BF16>{});
}

}  // namespace

extern "C" {

int gta_ln_fwd(const void* x, int32_t x_dtype, const float* gamma, const float* beta, float eps, int64_t rows, int32_t d,
               void* y, int32_t y_dtype, float* mean, float* rstd, void* stream) {
    if (!x || !gamma || !beta || !y || rows <= 0 || d <= 0 || !dtype_ok(x_dtype) || !dtype_ok(y_dtype)) return GTA_E_BADARG;
    if (!row_ok(d)) return GTA_E_UNSUPPORTED;
    if (!aligned_for(x, d, x_dtype) || !aligned_for(y, d, y_dtype) || !aligned16(gamma) || !aligned16(beta)) return GTA_E_BADARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((rows + 3) / 4);
    return by_row(d, [&](auto nj, auto vw) {
        return by_dtype(x_dtype, [&](auto xd) {
            return by_dtype(y_dtype, [&](auto yd) {
                hipLaunchKernelGGL((ln_fwd_kernel<decltype(xd)::value, decltype(yd)::value, decltype(nj)::value, decltype(vw)::value>),
                                   dim3(grid), dim3(256), 0, s, x, gamma, beta, eps, rows, d, y, mean, rstd);
                return launch_status();
            });
        });
    });
}

int64_t gta_ln_bwd_workspace_bytes(int64_t rows, int32_t d) {
    if (rows <= 0 || d <= 0) return 0;
    return (int64_t)ln_bwd_grid(rows) * 2 * d * (int64_t)sizeof(float);
}

int gta_ln_bwd(const void* dy, int32_t dy_dtype, const void* x, int32_t x_dtype, const float* gamma, const float* mean,
               const float* rstd, int64_t rows, int32_t d, const void* dres, void* dx, int32_t dx_dtype, void* dx_bf16,
               float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace || rows <= 0 || d <= 0 ||
        !dtype_ok(dy_dtype) || !dtype_ok(x_dtype))
        return GTA_E_BADARG;
    if (dx_dtype != x_dtype) return GTA_E_UNSUPPORTED;            // the skip connection keeps the stream's type
    if (!row_ok(d)) return GTA_E_UNSUPPORTED;
    if (workspace_bytes < gta_ln_bwd_workspace_bytes(rows, d)) return GTA_E_BADARG;
    if (!aligned_for(dy, d, dy_dtype) || !aligned_for(x, d, x_dtype) || !aligned_for(dx, d, x_dtype) || !aligned16(gamma) ||
        (dres && !aligned_for(dres, d, x_dtype)) || (dx_bf16 && !aligned_for(dx_bf16, d, GTA_DTYPE_BF16)))
        return GTA_E_BADARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = ln_bwd_grid(rows);
    float* part = static_cast<float*>(workspace);
    const size_t lds = (size_t)8 * d * sizeof(float);            // 128 KiB at d = 4096
    int rc = by_row(d, [&](auto nj, auto vw) {
        return by_dtype(dy_dtype, [&](auto gd) {
            return by_dtype(x_dtype, [&](auto xd) {
                auto kern = ln_bwd_kernel<decltype(gd)::value, decltype(xd)::value, decltype(nj)::value, decltype(vw)::value>;
                if (lds > 64 * 1024 &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                    return GTA_E_LAUNCH;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, dy, x, gamma, mean, rstd, rows, d, dres, dx, dx_bf16, part);
                return launch_status();
            });
        });
    });
    if (rc != GTA_OK) return rc;
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((2 * d + 31) / 32), dim3(1024), 0, s, part, grid, 2 * d, dgamma, dbeta, d);
    return launch_status();
}

int gta_gelu_fwd(const void* x, void* y, int32_t dtype, int64_t n, float p, uint64_t seed, void* stream) {
    if (!x || !y || n <= 0 || !dtype_ok(dtype) || !aligned16(x) || !aligned16(y)) return GTA_E_BADARG;
    if (n % 8 != 0) return GTA_E_UNSUPPORTED;
    Drop d;
    if (!make_drop(p, seed, &d)) return GTA_E_BADARG;
    const int64_t n8 = n / 8;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return by_dtype(dtype, [&](auto dt) {
        if (p > 0.f) hipLaunchKernelGGL((gelu_fwd_kernel<decltype(dt)::value, true>), dim3(ew_grid(n8)), dim3(256), 0, s, x, y, n8, d);
        else         hipLaunchKernelGGL((gelu_fwd_kernel<decltype(dt)::value, false>), dim3(ew_grid(n8)), dim3(256), 0, s, x, y, n8, d);
        return launch_status();
    });
}

int gta_gelu_bwd(const void* dy, const void* x, void* dx, int32_t dtype, int64_t n, float p, uint64_t seed, void* stream) {
    if (!dy || !x || !dx || n <= 0 || !dtype_ok(dtype) || !aligned16(x) || !aligned16(dy) || !aligned16(dx)) return GTA_E_BADARG;
    if (n % 8 != 0) return GTA_E_UNSUPPORTED;
    Drop d;
    if (!make_drop(p, seed, &d)) return GTA_E_BADARG;
    const int64_t n8 = n / 8;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return by_dtype(dtype, [&](auto dt) {
        if (p > 0.f) hipLaunchKernelGGL((gelu_bwd_kernel<decltype(dt)::value, true>), dim3(ew_grid(n8)), dim3(256), 0, s, dy, x, dx, n8, d);
        else         hipLaunchKernelGGL((gelu_bwd_kernel<decltype(dt)::value, false>), dim3(ew_grid(n8)), dim3(256), 0, s, dy, x, dx, n8, d);
        return launch_status();
    });
}

int gta_dropout_add(const void* z, int32_t z_dtype, const void* skip, void* out, int32_t skip_dtype, int64_t n, float p,
                    uint64_t seed, void* stream) {
    if (!z || !skip || !out || n <= 0 || !dtype_ok(z_dtype) || !dtype_ok(skip_dtype) || !aligned16(z) || !aligned16(skip) || !aligned16(out))
        return GTA_E_BADARG;
    if (n % 8 != 0) return GTA_E_UNSUPPORTED;
    Drop d;
    if (!make_drop(p, seed, &d)) return GTA_E_BADARG;
    const int64_t n8 = n / 8;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return by_dtype(z_dtype, [&](auto zd) {
        return by_dtype(skip_dtype, [&](auto sd) {
            hipLaunchKernelGGL((dropout_add_kernel<decltype(zd)::value, decltype(sd)::value>), dim3(ew_grid(n8)), dim3(256), 0, s, z,
                               skip, out, n8, d);
            return launch_status();
        });
    });
}

int gta_dropout_bwd(const void* dout, int32_t dout_dtype, void* dz, int32_t dz_dtype, int64_t n, float p, uint64_t seed, void* stream) {
    if (!dout || !dz || n <= 0 || !dtype_ok(dout_dtype) || !dtype_ok(dz_dtype) || !aligned16(dout) || !aligned16(dz)) return GTA_E_BADARG;
    if (n % 8 != 0) return GTA_E_UNSUPPORTED;
    Drop d;
    if (!make_drop(p, seed, &d)) return GTA_E_BADARG;
    const int64_t n8 = n / 8;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return by_dtype(dout_dtype, [&](auto gd) {
        return by_dtype(dz_dtype, [&](auto zd) {
            hipLaunchKernelGGL((dropout_bwd_kernel<decltype(gd)::value, decltype(zd)::value>), dim3(ew_grid(n8)), dim3(256), 0, s, dout, dz,
                               n8, d);
            return launch_status();
        });
    });
}

int64_t gta_colsum_workspace_bytes(int64_t m, int32_t n) {
    if (m <= 0 || n <= 0) return 0;
    return (int64_t)colsum_strips(m, n) * n * (int64_t)sizeof(float);
}

int gta_colsum(const void* a, int32_t dtype, int64_t m, int32_t n, int64_t ld, float* out, void* workspace,
               int64_t workspace_bytes, void* stream) {
    if (!a || !out || !workspace || m <= 0 || n <= 0 || ld < n || !dtype_ok(dtype)) return GTA_E_BADARG;
    if (n % 4 != 0 || ld % 4 != 0) return GTA_E_UNSUPPORTED;
    const bool wide = n % 8 == 0 && ld % 8 == 0;
    if (!aligned_for(a, wide ? 8 : 4, dtype)) return GTA_E_BADARG;
    if (workspace_bytes < gta_colsum_workspace_bytes(m, n)) return GTA_E_BADARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int strips = colsum_strips(m, n);
    float* part = static_cast<float*>(workspace);
    int rc = by_dtype(dtype, [&](auto dt) {
        if (wide) hipLaunchKernelGGL((colsum_kernel<decltype(dt)::value, 8>), dim3((n + 511) / 512, strips), dim3(256), 0, s, a, m, n, ld, part);
        else      hipLaunchKernelGGL((colsum_kernel<decltype(dt)::value, 4>), dim3((n + 255) / 256, strips), dim3(256), 0, s, a, m, n, ld, part);
        return launch_status();
    });
    if (rc != GTA_OK) return rc;
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((n + 31) / 32), dim3(1024), 0, s, part, strips, n, out, out, n);
    return launch_status();
}

}  // extern "C"
