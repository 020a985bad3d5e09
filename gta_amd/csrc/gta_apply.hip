// gta_apply.hip -- generic rho application for ANY f_dims layout (ablation paths of the reference):
// t2 slab (gta.py:221-238,272-274), euclid similarity (gta.py:146-156,251-253; layers.py:213-224),
// so3 of degree 1, unaligned slabs -- and the fp32 rho of the fp32-faithful mode (gta.py: precise=True with gradients).  One thread per
// (batch, head, token) row, block by block.  r04: the rows pass through LDS -- a wave copies its 64 rows in (one coalesced access per row:
// the lanes take the row's channels), every lane then works on ITS row in LDS (rows 4 (dh + 1) bytes apart: an odd word stride, no bank
// conflicts) and the wave copies the results out the same way; straight from global memory every lane's element access touched another
// row -- 64 lines per instruction (the direct form is kept for head dimensions whose rows do not fit: dh > 300).
//   mode 0: q side      q' = blockdiag((E_q.m)^T | D(R_q) | R(th_q) | (T_q^-1)^T) q      (euclid: affine inv(E_q).m)
//   mode 1: k side      k' = blockdiag(inv(E_k).m | D(R_k) | R(th_k) | T_k) k  (also v)  (euclid: affine inv(E_k).m)
//   mode 2: output      o  = blockdiag(E_q.m | D(R_q)^T | R(th_q)^T | T_q^-1) o~        (euclid: affine E_q.m)
#include "gta_common.h"
#include "../../include/gta_hip.h"

namespace {

struct ApplyParams {
    const void* x; void* y;
    long x_sb, x_sh, x_st, y_sb, y_sh, y_st;
    const float *vrep, *cs, *coord, *trans_coeff;
    float* key_bias; float bias_scale; long bias_pitch;
    int B, H, T, N, P;
    int d_triv, d_se3, d_so3, d_so2, d_t2, L;
    int mode, euclid, esz;
};

template <int ESZ> GTA_DEV float ld(const char* p, int i) {
    if (ESZ == 4) return reinterpret_cast<const float*>(p)[i];
    return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(p)[i] << 16);
}
template <int ESZ> GTA_DEV void st(char* p, int i, float v) {
    if (ESZ == 4) { reinterpret_cast<float*>(p)[i] = v; return; }
    reinterpret_cast<uint16_t*>(p)[i] = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
}
// a row as the bodies below see it: element ch in, element ch out -- in global memory (direct form) or as fp32 in the wave's LDS stage
template <int ESZ> struct GIn  { const char* p; GTA_DEV float operator()(int ch) const { return ld<ESZ>(p, ch); } };
template <int ESZ> struct GOut { char* p;       GTA_DEV void operator()(int ch, float v) const { st<ESZ>(p, ch, v); } };
struct LIn  { const float* p; GTA_DEV float operator()(int ch) const { return p[ch]; } };
struct LOut { float* p;       GTA_DEV void operator()(int ch, float v) const { p[ch] = v; } };

GTA_DEV void row_decode(long row, int T, int H, int& b, int& h, int& t) {
    t = (int)(row % T);
    h = (int)((row / T) % H);
    b = (int)(row / ((long)T * H));
}
// 64 rows of dh elements between global memory (base + (b sb + h sh + t st) elements, rows [row0, row0 + 64) of the launch) and a wave's stage
// [64][dh + 1] floats: one access per row and 64 channels.  Lane l knows the offset of row row0 + l (my_off, bytes; < 0 past the end); the
// wave reads them back lane by lane (uniform values), SIXTEEN rows' loads in flight before their LDS writes (a load -> write chain per row
// cost one memory latency per row: 64 of them in a row were most of these kernels' time)
template <int ESZ, bool IN>
GTA_DEV void stage_rows(float* stage, const void* base, const long my_off, const int dh, const int lane) {
    constexpr int NB = 16;                               // rows in flight per batch
    // (uniform trip count: the readlanes below run under the full EXEC mask of the wave -- a `ch = lane; ch < dh` loop would run its last
    //  round, at dh % 64 != 0, with the source lanes of some rows masked off and rely on their registers' stale contents)
    for (int c0 = 0; c0 < dh; c0 += 64) {
        const int ch = c0 + lane;
        const bool chok = ch < dh;
#pragma unroll 1
        for (int r0 = 0; r0 < 64; r0 += NB) {
            long off[NB];
            float v[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)(uint64_t)my_off, r0 + u);
                const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)my_off >> 32), r0 + u);
                off[u] = (long)(((uint64_t)hi << 32) | lo);
            }
            if (IN) {
#pragma unroll
                for (int u = 0; u < NB; ++u) v[u] = (chok && off[u] >= 0) ? ld<ESZ>((const char*)base + off[u], ch) : 0.f;
#pragma unroll
                for (int u = 0; u < NB; ++u)
                    if (chok) stage[(r0 + u) * (dh + 1) + ch] = v[u];
            } else {
#pragma unroll
                for (int u = 0; u < NB; ++u) v[u] = chok ? stage[(r0 + u) * (dh + 1) + ch] : 0.f;
#pragma unroll
                for (int u = 0; u < NB; ++u)
                    if (chok && off[u] >= 0) st<ESZ>((char*)base + off[u], ch, v[u]);
            }
        }
    }
}
GTA_DEV long row_offset(long row, long total, int T, int H, long sb, long sh, long st_, int esz) {
    if (row >= total) return -1;
    int b, h, t;
    row_decode(row, T, H, b, h, t);
    return ((long)b * sb + (long)h * sh + (long)t * st_) * esz;
}

template <class In, class Out>
GTA_DEV void apply_row(const ApplyParams& p, const int b, const int h, const int t, const In X, const Out Y) {
    const float tc = p.trans_coeff ? *p.trans_coeff : 1.0f;
    const int n = t / p.P;
    const float* vr = p.vrep ? p.vrep + ((long)b * p.N + n) * GTA_VREP_STRIDE : nullptr;
    float sq = 0.f;
    int ch = 0;
    for (int i = 0; i < p.d_triv; ++i, ++ch) { const float v = X(ch); Y(ch, v); sq += v * v; }
    if (p.d_se3 > 0) {
        // matrix used: mode 0 non-euclid: (E.m)^T ; mode 0 euclid: inv(E).m ; mode 1: inv(E).m ; mode 2: E.m
        float M[16];
        const bool use_inv_slot = (p.mode == 2) || (p.mode == 0 && !p.euclid);     // "inv" slot holds E
        const float* src = vr + (use_inv_slot ? GTA_VREP_INV : GTA_VREP_REP);
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                const float m = (r == 3) ? (c == 3 ? 1.f : 0.f) : (c == 3 ? tc : 1.f);
                const float v = src[r * 4 + c] * m;
                if (p.mode == 0 && !p.euclid) M[c * 4 + r] = v; else M[r * 4 + c] = v;
            }
        if (p.euclid) {
            for (int blk = 0; blk < p.d_se3 / 3; ++blk, ch += 3) {
                const float a = X(ch), bb = X(ch + 1), c = X(ch + 2);
                for (int r = 0; r < 3; ++r) {
                    const float v = M[r * 4] * a + M[r * 4 + 1] * bb + M[r * 4 + 2] * c + M[r * 4 + 3];   // homogenisation
                    Y(ch + r, v); sq += v * v;
                }
            }
        } else {
            for (int blk = 0; blk < p.d_se3 / 4; ++blk, ch += 4) {
                const float a = X(ch), bb = X(ch + 1), c = X(ch + 2), d = X(ch + 3);
                for (int r = 0; r < 4; ++r) {
                    const float v = M[r * 4] * a + M[r * 4 + 1] * bb + M[r * 4 + 2] * c + M[r * 4 + 3] * d;
                    Y(ch + r, v); sq += v * v;
                }
            }
        }
    }
    if (p.d_so3 > 0) {
        const int tot = p.L >= 2 ? 8 : 3;
        for (int g = 0; g < p.d_so3 / tot; ++g) {
            for (int l = 1; l <= p.L; ++l) {
                const int dim = 2 * l + 1;
                const float* D = vr + (l == 1 ? GTA_VREP_D1 : GTA_VREP_D2);
                float in[5];
                for (int i = 0; i < dim; ++i) in[i] = X(ch + i);
                for (int r = 0; r < dim; ++r) {
                    float v = 0.f;
                    for (int c = 0; c < dim; ++c) v += (p.mode == 2 ? D[c * dim + r] : D[r * dim + c]) * in[c];
                    Y(ch + r, v); sq += v * v;
                }
                ch += dim;
            }
        }
    }
    if (p.d_so2 > 0) {
        const int nblk = p.d_so2 / 2;
        const float* cs = p.cs + ((long)b * p.T + t) * 2 * nblk;
        for (int blk = 0; blk < nblk; ++blk, ch += 2) {
            const float c = cs[2 * blk], s = (p.mode == 2 ? -1.f : 1.f) * cs[2 * blk + 1];
            const float a = X(ch), bb = X(ch + 1);
            const float v0 = c * a - s * bb, v1 = s * a + c * bb;
            Y(ch, v0); Y(ch + 1, v1); sq += v0 * v0 + v1 * v1;
        }
    }
    if (p.d_t2 > 0) {
        const float cx = p.coord[((long)b * p.T + t) * 2], cy = p.coord[((long)b * p.T + t) * 2 + 1];
        for (int blk = 0; blk < p.d_t2 / 3; ++blk, ch += 3) {
            const float a = X(ch), bb = X(ch + 1), c = X(ch + 2);
            float v0, v1, v2;
            if (p.mode == 0)      { v0 = a - cx * c; v1 = bb - cy * c; v2 = c; }                   // (T^-1)^T
            else if (p.mode == 1) { v0 = a; v1 = bb; v2 = cx * a + cy * bb + c; }                  // T
            else                  { v0 = a; v1 = bb; v2 = c - cx * a - cy * bb; }                  // T^-1
            Y(ch, v0); Y(ch + 1, v1); Y(ch + 2, v2); sq += v0 * v0 + v1 * v1 + v2 * v2;
        }
    }
    if (p.key_bias) p.key_bias[((long)b * p.H + h) * p.bias_pitch + t] = -0.5f * p.bias_scale * sq;
}

template <int ESZ, bool STAGED>
__global__ void gta_apply_kernel(const ApplyParams p) {
    const long total = (long)p.B * p.H * p.T;
    int b, h, t;
    if constexpr (!STAGED) {
        const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
        if (row >= total) return;
        row_decode(row, p.T, p.H, b, h, t);
        const char* x = (const char*)p.x + ((long)b * p.x_sb + (long)h * p.x_sh + (long)t * p.x_st) * ESZ;
        char* y = (char*)p.y + ((long)b * p.y_sb + (long)h * p.y_sh + (long)t * p.y_st) * ESZ;
        apply_row(p, b, h, t, GIn<ESZ>{x}, GOut<ESZ>{y});
    } else {                                             // one wave per workgroup, 64 rows through its stage (in place)
        extern __shared__ float stage[];
        const int dh = p.d_triv + p.d_se3 + p.d_so3 + p.d_so2 + p.d_t2, lane = threadIdx.x;
        const long row0 = (long)blockIdx.x * 64;
        stage_rows<ESZ, true>(stage, p.x, row_offset(row0 + lane, total, p.T, p.H, p.x_sb, p.x_sh, p.x_st, ESZ), dh, lane);
        __syncthreads();
        if (row0 + lane < total) {
            row_decode(row0 + lane, p.T, p.H, b, h, t);
            float* r = stage + lane * (dh + 1);
            apply_row(p, b, h, t, LIn{r}, LOut{r});
        }
        __syncthreads();
        stage_rows<ESZ, false>(stage, p.y, row_offset(row0 + lane, total, p.T, p.H, p.y_sb, p.y_sh, p.y_st, ESZ), dh, lane);
    }
}

// Adjoint of the above: dx = M^T dy for the block-diagonal M of `mode`, plus (optionally) this row's contribution
// to d loss / d trans_coeff (the entries of the masked se3 matrices that carry c, gta.py:40-44) -- one float per row,
// summed by the caller in a fixed order.  Under euclid the key side may also carry d key_bias: the bias is
// -0.5 s |y|^2, so dy_eff = dy - s * dbias * y with y recomputed from x.
struct ApplyBwdParams {
    ApplyParams f;                 // x = forward input, y unused
    const void* dy; void* dx;
    long dy_sb, dy_sh, dy_st, dx_sb, dx_sh, dx_st;
    const float* dbias;            // [B,H,bias_pitch] or null
    float* dtc_rows;               // [B,H,T] or null
};

template <class In, class InG, class Out>
GTA_DEV float apply_bwd_row(const ApplyBwdParams& q, const int b, const int h, const int t, const In X, const InG DY, const Out DX) {
    const ApplyParams& p = q.f;
    const float tc = p.trans_coeff ? *p.trans_coeff : 1.0f;
    const int n = t / p.P;
    const float* vr = p.vrep ? p.vrep + ((long)b * p.N + n) * GTA_VREP_STRIDE : nullptr;
    const float kb = q.dbias ? -p.bias_scale * q.dbias[((long)b * p.H + h) * p.bias_pitch + t] : 0.f;   // d/dy of the bias = kb * y
    float dc = 0.f;
    int ch = 0;
    for (int i = 0; i < p.d_triv; ++i, ++ch) DX(ch, DY(ch) + kb * X(ch));
    if (p.d_se3 > 0) {
        float M[16];                                     // the forward's matrix, row-major (y = M x)
        const bool use_inv_slot = (p.mode == 2) || (p.mode == 0 && !p.euclid);
        const float* src = vr + (use_inv_slot ? GTA_VREP_INV : GTA_VREP_REP);
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                const float m = (r == 3) ? (c == 3 ? 1.f : 0.f) : (c == 3 ? tc : 1.f);
                const float v = src[r * 4 + c] * m;
                if (p.mode == 0 && !p.euclid) M[c * 4 + r] = v; else M[r * 4 + c] = v;
            }
        if (p.euclid) {
            for (int blk = 0; blk < p.d_se3 / 3; ++blk, ch += 3) {
                const float a = X(ch), bb = X(ch + 1), c = X(ch + 2);
                float g[3];
                for (int r = 0; r < 3; ++r) {
                    const float y = M[r * 4] * a + M[r * 4 + 1] * bb + M[r * 4 + 2] * c + M[r * 4 + 3];
                    g[r] = DY(ch + r) + kb * y;
                    dc += g[r] * src[r * 4 + 3];                            // y_r = ... + c * t_r
                }
                for (int col = 0; col < 3; ++col) DX(ch + col, M[col] * g[0] + M[4 + col] * g[1] + M[8 + col] * g[2]);
            }
        } else {
            for (int blk = 0; blk < p.d_se3 / 4; ++blk, ch += 4) {
                float xi[4], g[4];
                for (int i = 0; i < 4; ++i) { xi[i] = X(ch + i); g[i] = DY(ch + i); }
                if (kb != 0.f)
                    for (int r = 0; r < 4; ++r)
                        g[r] += kb * (M[r * 4] * xi[0] + M[r * 4 + 1] * xi[1] + M[r * 4 + 2] * xi[2] + M[r * 4 + 3] * xi[3]);
                for (int col = 0; col < 4; ++col)
                    DX(ch + col, M[col] * g[0] + M[4 + col] * g[1] + M[8 + col] * g[2] + M[12 + col] * g[3]);
                if (p.mode == 0) dc += g[3] * (src[3] * xi[0] + src[7] * xi[1] + src[11] * xi[2]);       // y_3 = sum_c E[c][3] c x_c + x_3
                else             dc += (g[0] * src[3] + g[1] * src[7] + g[2] * src[11]) * xi[3];           // y_r = ... + M[r][3] c x_3
            }
        }
    }
    if (p.d_so3 > 0) {
        const int tot = p.L >= 2 ? 8 : 3;
        for (int gI = 0; gI < p.d_so3 / tot; ++gI) {
            for (int l = 1; l <= p.L; ++l) {
                const int dim = 2 * l + 1;
                const float* D = vr + (l == 1 ? GTA_VREP_D1 : GTA_VREP_D2);
                float in[5];
                for (int i = 0; i < dim; ++i) in[i] = DY(ch + i);
                for (int r = 0; r < dim; ++r) {
                    float v = kb * X(ch + r);                     // D orthogonal: D^T (kb D x) = kb x
                    for (int c = 0; c < dim; ++c) v += (p.mode == 2 ? D[r * dim + c] : D[c * dim + r]) * in[c];
                    DX(ch + r, v);
                }
                ch += dim;
            }
        }
    }
    if (p.d_so2 > 0) {
        const int nblk = p.d_so2 / 2;
        const float* cs = p.cs + ((long)b * p.T + t) * 2 * nblk;
        for (int blk = 0; blk < nblk; ++blk, ch += 2) {
            const float c = cs[2 * blk], s = (p.mode == 2 ? -1.f : 1.f) * cs[2 * blk + 1];
            const float a = DY(ch), bb = DY(ch + 1);
            DX(ch, c * a + s * bb + kb * X(ch)); DX(ch + 1, -s * a + c * bb + kb * X(ch + 1));
        }
    }
    if (p.d_t2 > 0) {
        const float cx = p.coord[((long)b * p.T + t) * 2], cy = p.coord[((long)b * p.T + t) * 2 + 1];
        for (int blk = 0; blk < p.d_t2 / 3; ++blk, ch += 3) {
            float a = DY(ch), bb = DY(ch + 1), c = DY(ch + 2);
            if (kb != 0.f) {                                               // y = T x recomputed for the bias term
                const float xa = X(ch), xb = X(ch + 1), xc = X(ch + 2);
                float y0, y1, y2;
                if (p.mode == 0)      { y0 = xa - cx * xc; y1 = xb - cy * xc; y2 = xc; }
                else if (p.mode == 1) { y0 = xa; y1 = xb; y2 = cx * xa + cy * xb + xc; }
                else                  { y0 = xa; y1 = xb; y2 = xc - cx * xa - cy * xb; }
                a += kb * y0; bb += kb * y1; c += kb * y2;
            }
            float v0, v1, v2;
            if (p.mode == 0)      { v0 = a; v1 = bb; v2 = c - cx * a - cy * bb; }
            else if (p.mode == 1) { v0 = a + cx * c; v1 = bb + cy * c; v2 = c; }
            else                  { v0 = a - cx * c; v1 = bb - cy * c; v2 = c; }
            DX(ch, v0); DX(ch + 1, v1); DX(ch + 2, v2);
        }
    }
    return dc;
}

template <int ESZ, bool STAGED>
__global__ void gta_apply_bwd_kernel(const ApplyBwdParams q) {
    const ApplyParams& p = q.f;
    const long total = (long)p.B * p.H * p.T;
    int b, h, t;
    if constexpr (!STAGED) {
        const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
        if (row >= total) return;
        row_decode(row, p.T, p.H, b, h, t);
        const char* x = (const char*)p.x + ((long)b * p.x_sb + (long)h * p.x_sh + (long)t * p.x_st) * ESZ;
        const char* dy = (const char*)q.dy + ((long)b * q.dy_sb + (long)h * q.dy_sh + (long)t * q.dy_st) * ESZ;
        char* dx = (char*)q.dx + ((long)b * q.dx_sb + (long)h * q.dx_sh + (long)t * q.dx_st) * ESZ;
        const float dc = apply_bwd_row(q, b, h, t, GIn<ESZ>{x}, GIn<ESZ>{dy}, GOut<ESZ>{dx});
        if (q.dtc_rows) q.dtc_rows[row] = dc;
    } else {                                             // one wave per workgroup: x in one stage, dy -> dx in place in a second
        extern __shared__ float stage[];
        const int dh = p.d_triv + p.d_se3 + p.d_so3 + p.d_so2 + p.d_t2, lane = threadIdx.x;
        float* sx = stage;
        float* sg = stage + 64 * (dh + 1);
        const long row0 = (long)blockIdx.x * 64;
        stage_rows<ESZ, true>(sx, p.x, row_offset(row0 + lane, total, p.T, p.H, p.x_sb, p.x_sh, p.x_st, ESZ), dh, lane);
        stage_rows<ESZ, true>(sg, q.dy, row_offset(row0 + lane, total, p.T, p.H, q.dy_sb, q.dy_sh, q.dy_st, ESZ), dh, lane);
        __syncthreads();
        if (row0 + lane < total) {
            row_decode(row0 + lane, p.T, p.H, b, h, t);
            const float dc = apply_bwd_row(q, b, h, t, LIn{sx + lane * (dh + 1)}, LIn{sg + lane * (dh + 1)}, LOut{sg + lane * (dh + 1)});
            if (q.dtc_rows) q.dtc_rows[row0 + lane] = dc;
        }
        __syncthreads();
        stage_rows<ESZ, false>(sg, q.dx, row_offset(row0 + lane, total, p.T, p.H, q.dx_sb, q.dx_sh, q.dx_st, ESZ), dh, lane);
    }
}

constexpr int APPLY_LDS_MAX = 160 * 1024;
}  // namespace

extern "C" int gta_rep_apply(const GtaAttnDesc* d, int32_t mode, const void* x, const int64_t* x_stride,
                             const float* vrep, const float* cs, const float* coord, const float* trans_coeff,
                             void* y, const int64_t* y_stride, float* key_bias, float bias_scale, int64_t bias_pitch,
                             void* stream) {
    if (!d || !x || !y || !x_stride || !y_stride || mode < 0 || mode > 2) return GTA_E_BADARG;
    if (d->abi_version != GTA_ABI_VERSION) return GTA_E_BADARG;
    if (d->d_triv + d->d_se3 + d->d_so3 + d->d_so2 + d->d_t2 != d->dh) return GTA_E_LAYOUT;
    const bool euclid = (d->flags & GTA_FLAG_EUCLID) != 0;
    if (d->d_se3 % (euclid ? 3 : 4) || d->d_so2 % 2 || d->d_t2 % 3) return GTA_E_LAYOUT;
    if (d->d_so3 > 0 && (d->so3_degree < 1 || d->so3_degree > 2 || d->d_so3 % (d->so3_degree == 2 ? 8 : 3))) return GTA_E_UNSUPPORTED;
    if ((d->d_se3 > 0 || d->d_so3 > 0) && !vrep) return GTA_E_BADARG;
    if (d->d_so2 > 0 && !cs) return GTA_E_BADARG;
    if (d->d_t2 > 0 && !coord) return GTA_E_BADARG;
    ApplyParams p;
    p.x = x; p.y = y;
    p.x_sb = x_stride[0]; p.x_sh = x_stride[1]; p.x_st = x_stride[2];
    p.y_sb = y_stride[0]; p.y_sh = y_stride[1]; p.y_st = y_stride[2];
    p.vrep = vrep; p.cs = cs; p.coord = coord; p.trans_coeff = trans_coeff;
    p.key_bias = key_bias; p.bias_scale = bias_scale; p.bias_pitch = bias_pitch;
    p.B = d->B; p.H = d->H;
    p.T = mode == 1 ? d->Tk : d->Tq;
    p.N = mode == 1 ? d->Nk : d->Nq;
    p.P = p.T / p.N;
    p.d_triv = d->d_triv; p.d_se3 = d->d_se3; p.d_so3 = d->d_so3; p.d_so2 = d->d_so2; p.d_t2 = d->d_t2; p.L = d->so3_degree;
    p.mode = mode; p.euclid = euclid ? 1 : 0; p.esz = d->dtype == GTA_DTYPE_BF16 ? 2 : 4;
    const long total = (long)p.B * p.H * p.T;
    const int lds = 64 * (d->dh + 1) * 4;                // the staged form: one wave per workgroup, its 64 rows in LDS
    if (lds <= APPLY_LDS_MAX && (total + 63) / 64 < 0x7fffffffL) {
        const unsigned nb = (unsigned)((total + 63) / 64);
        if (p.esz == 2) {
            if (int rc = gta_lds_optin<&gta_apply_kernel<2, true>>(APPLY_LDS_MAX)) return rc;
            hipLaunchKernelGGL((gta_apply_kernel<2, true>), dim3(nb), dim3(64), lds, (hipStream_t)stream, p);
        } else {
            if (int rc = gta_lds_optin<&gta_apply_kernel<4, true>>(APPLY_LDS_MAX)) return rc;
            hipLaunchKernelGGL((gta_apply_kernel<4, true>), dim3(nb), dim3(64), lds, (hipStream_t)stream, p);
        }
        return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
    }
    const int th = 256;
    const unsigned nb = (unsigned)((total + th - 1) / th);
    if (p.esz == 2) hipLaunchKernelGGL((gta_apply_kernel<2, false>), dim3(nb), dim3(th), 0, (hipStream_t)stream, p);
    else            hipLaunchKernelGGL((gta_apply_kernel<4, false>), dim3(nb), dim3(th), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

extern "C" int gta_rep_apply_bwd(const GtaAttnDesc* d, int32_t mode, const void* x, const int64_t* x_stride,
                                 const void* dy, const int64_t* dy_stride, const float* vrep, const float* cs,
                                 const float* coord, const float* trans_coeff, const float* dkey_bias, float bias_scale,
                                 int64_t bias_pitch, void* dx, const int64_t* dx_stride, float* dtc_rows, void* stream) {
    if (!d || !x || !dy || !dx || !x_stride || !dy_stride || !dx_stride || mode < 0 || mode > 2) return GTA_E_BADARG;
    if (d->abi_version != GTA_ABI_VERSION) return GTA_E_BADARG;
    if (d->d_triv + d->d_se3 + d->d_so3 + d->d_so2 + d->d_t2 != d->dh) return GTA_E_LAYOUT;
    const bool euclid = (d->flags & GTA_FLAG_EUCLID) != 0;
    if (d->d_se3 % (euclid ? 3 : 4) || d->d_so2 % 2 || d->d_t2 % 3) return GTA_E_LAYOUT;
    if (d->d_so3 > 0 && (d->so3_degree < 1 || d->so3_degree > 2 || d->d_so3 % (d->so3_degree == 2 ? 8 : 3))) return GTA_E_UNSUPPORTED;
    if ((d->d_se3 > 0 || d->d_so3 > 0) && !vrep) return GTA_E_BADARG;
    if (d->d_so2 > 0 && !cs) return GTA_E_BADARG;
    if (d->d_t2 > 0 && !coord) return GTA_E_BADARG;
    ApplyBwdParams q;
    ApplyParams& p = q.f;
    p.x = x; p.y = nullptr;
    p.x_sb = x_stride[0]; p.x_sh = x_stride[1]; p.x_st = x_stride[2];
    p.y_sb = p.y_sh = p.y_st = 0;
    p.vrep = vrep; p.cs = cs; p.coord = coord; p.trans_coeff = trans_coeff;
    p.key_bias = nullptr; p.bias_scale = bias_scale; p.bias_pitch = bias_pitch;
    p.B = d->B; p.H = d->H;
    p.T = mode == 1 ? d->Tk : d->Tq;
    p.N = mode == 1 ? d->Nk : d->Nq;
    p.P = p.T / p.N;
    p.d_triv = d->d_triv; p.d_se3 = d->d_se3; p.d_so3 = d->d_so3; p.d_so2 = d->d_so2; p.d_t2 = d->d_t2; p.L = d->so3_degree;
    p.mode = mode; p.euclid = euclid ? 1 : 0; p.esz = d->dtype == GTA_DTYPE_BF16 ? 2 : 4;
    q.dy = dy; q.dx = dx;
    q.dy_sb = dy_stride[0]; q.dy_sh = dy_stride[1]; q.dy_st = dy_stride[2];
    q.dx_sb = dx_stride[0]; q.dx_sh = dx_stride[1]; q.dx_st = dx_stride[2];
    q.dbias = dkey_bias; q.dtc_rows = dtc_rows;
    const long total = (long)p.B * p.H * p.T;
    const int lds = 2 * 64 * (d->dh + 1) * 4;            // the staged form: x and dy -> dx of the wave's 64 rows in LDS
    if (lds <= APPLY_LDS_MAX && (total + 63) / 64 < 0x7fffffffL) {
        const unsigned nb = (unsigned)((total + 63) / 64);
        if (p.esz == 2) {
            if (int rc = gta_lds_optin<&gta_apply_bwd_kernel<2, true>>(APPLY_LDS_MAX)) return rc;
            hipLaunchKernelGGL((gta_apply_bwd_kernel<2, true>), dim3(nb), dim3(64), lds, (hipStream_t)stream, q);
        } else {
            if (int rc = gta_lds_optin<&gta_apply_bwd_kernel<4, true>>(APPLY_LDS_MAX)) return rc;
            hipLaunchKernelGGL((gta_apply_bwd_kernel<4, true>), dim3(nb), dim3(64), lds, (hipStream_t)stream, q);
        }
        return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
    }
    const int th = 256;
    const unsigned nb = (unsigned)((total + th - 1) / th);
    if (p.esz == 2) hipLaunchKernelGGL((gta_apply_bwd_kernel<2, false>), dim3(nb), dim3(th), 0, (hipStream_t)stream, q);
    else            hipLaunchKernelGGL((gta_apply_bwd_kernel<4, false>), dim3(nb), dim3(th), 0, (hipStream_t)stream, q);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}
