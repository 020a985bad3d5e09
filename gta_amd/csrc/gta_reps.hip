// gta_reps.hip -- rep builders on device (replace encoder.py:183-265 / decoder.py:247-353).
//
// Tiny kernels (B*N views, B*T tokens): HBM traffic is a few hundred KB per forward and they run
// once per encoder/decoder call, shared by all layers exactly like the reference's
// pre_compute_reps output.  One wave per view / one thread per (token, block); nothing to tile.
#include "gta_common.h"
#include "../../include/gta_hip.h"

namespace {

// J matrices of the reference's real-SH basis (wigner_d.py:16-25 fixes the basis).  Restated from
// the Pinchon-Hoggan construction; the reference reads them from J_dense.pt (absent from the
// checkout) -> values are "parity unpinned", see DESIGN.md.
__constant__ float kJ1[9] = {0.f, 1.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, -1.f};
#define S3H 0.8660254037844386f
__constant__ float kJ2[25] = {0.f, 0.f, 0.f,  -1.f, 0.f,
                              0.f, 1.f, 0.f,   0.f, 0.f,
                              0.f, 0.f, -0.5f, 0.f, -S3H,
                              -1.f, 0.f, 0.f,  0.f, 0.f,
                              0.f, 0.f, -S3H,  0.f, 0.5f};

// Z(angle) of wigner_d.py:16-25: cos(m a) on the diagonal, sin(m a) on the anti-diagonal, m = l..-l (the
// diagonal is written last: centre element = cos 0 = 1);  D = Z(g3) J Z(g2) J Z(g1)  (wigner_d.py:28-35).
// General 4x4 inverse: Gauss-Jordan with partial pivoting in fp64 (the reference calls torch.linalg.inv, encoder.py:219 -- a general
// inverse, not the rigid closed form), see inv4_lanes below.
// One wave per view.  The work of a view is a chain of dependent scalar steps (a 4x4 fp64 inverse, five atan2f, six sincosf, the
// Wigner products): run redundantly in every lane it took ~6.5 us of pure latency (r03: the rep build was 8.3 us of a 240-us step).
// r04: the independent pieces run in DIFFERENT LANES at once -- the eight columns of the augmented matrix [E | I] of the Gauss-Jordan
// inverse (row operations act column by column; only the pivot column is broadcast per step), the five atan2f (one call, a different
// argument pair per lane), the six sincosf -- and are handed round by wave shuffles; per element the arithmetic and its order are those
// of the one-lane form.  Then lanes split the Wigner products by output COLUMN: D[:, j] = Z(g3) J Z(g2) J Z(g1)[:, j] is a chain of
// matrix-vector products (Z has two non-zeros per row), ~70 FMAs per lane.
__device__ __forceinline__ void inv4_lanes(const float* E, float* out, int lane) {
    // lane (j = lane & 7) owns column j of [E | I] (every lane of the wave takes part: the shuffles want all of them)
    const int j = lane & 7;
    double col[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) col[i] = j < 4 ? (double)E[i * 4 + j] : ((i == j - 4) ? 1.0 : 0.0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double pc[4];                                   // the pivot column as every lane sees it
#pragma unroll
        for (int r = 0; r < 4; ++r) pc[r] = __shfl(col[r], c);
        // partial pivoting by compare-and-swap with every lower row, decided on the broadcast column
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            const bool sw = fabs(pc[r]) > fabs(pc[c]);
            const double x = col[c], y = col[r], px = pc[c], py = pc[r];
            col[c] = sw ? y : x; col[r] = sw ? x : y;
            pc[c] = sw ? py : px; pc[r] = sw ? px : py;
        }
        const double inv = 1.0 / pc[c];
        col[c] *= inv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            col[r] -= pc[r] * col[c];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) out[i * 4 + jj] = (float)__shfl(col[i], 4 + jj);
}

template <int N>
__device__ __forceinline__ void wigner_column(const float* J, const float (&cs1)[3][2], const float (&cs2)[3][2],
                                              const float (&cs3)[3][2], int j, float* col) {
    // cs*[m] = (cos, sin)(m * angle), m = 0..2;  Z(a)[i][i] = cos((l-i)a), Z(a)[i][N-1-i] = sin((l-i)a), centre = 1
    constexpr int l = (N - 1) / 2;
    auto c_of = [](const float (&cs)[3][2], int m) { return cs[m < 0 ? -m : m][0]; };
    auto s_of = [](const float (&cs)[3][2], int m) { return m < 0 ? -cs[-m][1] : cs[m][1]; };
    float v[N], w[N];
    // v = Z1[:, j]
#pragma unroll
    for (int d = 0; d < N; ++d) {
        const float diag = c_of(cs1, l - d), anti = s_of(cs1, l - d);
        v[d] = (d == j) ? diag : ((d == N - 1 - j) ? anti : 0.f);
    }
    // w = J v
#pragma unroll
    for (int r = 0; r < N; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < N; ++c) acc += J[r * N + c] * v[c];
        w[r] = acc;
    }
    // v = Z2 w
#pragma unroll
    for (int r = 0; r < N; ++r)
        v[r] = (r == N - 1 - r) ? w[r] : (s_of(cs2, l - r) * w[N - 1 - r] + c_of(cs2, l - r) * w[r]);
    // w = J v
#pragma unroll
    for (int r = 0; r < N; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < N; ++c) acc += J[r * N + c] * v[c];
        w[r] = acc;
    }
    // col = Z3 w
#pragma unroll
    for (int r = 0; r < N; ++r)
        col[r] = (r == N - 1 - r) ? w[r] : (s_of(cs3, l - r) * w[N - 1 - r] + c_of(cs3, l - r) * w[r]);
}

__device__ __forceinline__ void view_reps_body(int block, const float* __restrict__ E, int n_views, int L,
                                               float* __restrict__ vrep) {
    const int lane = threadIdx.x & 63;
    const int i = block * 4 + (threadIdx.x >> 6);
    if (i >= n_views) return;
    const float* e = E + (size_t)i * 16;
    float* o = vrep + (size_t)i * GTA_VREP_STRIDE;
    float inv[16];
    inv4_lanes(e, inv, lane);
    // records: lanes 0..15 write E and inverse(E); lanes 16.. zero the padding
    if (lane < 16) {
        float b2 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { if (j == lane) b2 = inv[j]; }
        o[GTA_VREP_INV + lane] = e[lane];
        o[GTA_VREP_REP + lane] = b2;
    }
    for (int j = GTA_VREP_D2 + 25 + lane; j < GTA_VREP_STRIDE; j += 64) o[j] = 0.f;
    if (L < 1 && lane < 9) o[GTA_VREP_D1 + lane] = 0.f;
    if (L < 2 && lane < 25) o[GTA_VREP_D2 + lane] = 0.f;
    if (L < 1) return;
    // R = inverse(E)[:3,:3]; ZYZ angles with the reference's gimbal masks (wigner_d.py:39-49): the five atan2f in lanes 0..4
#define R_(r, c) inv[(r) * 4 + (c)]
    const float EPS = 1e-5f;
    const float ay = lane == 0 ? R_(2, 1) : lane == 1 ? sqrtf(R_(0, 2) * R_(0, 2) + R_(1, 2) * R_(1, 2)) : lane == 2 ? R_(1, 2) : lane == 3 ? R_(1, 0) : -R_(1, 0);
    const float ax = lane == 0 ? -R_(2, 0) : lane == 1 ? R_(2, 2) : lane == 2 ? R_(0, 2) : lane == 3 ? R_(0, 0) : -R_(0, 0);
    const float at = atan2f(ay, ax);
    float g1 = __shfl(at, 0);
    const float g2 = __shfl(at, 1);
    float g3 = __shfl(at, 2);
    const float up = (fabsf(R_(2, 2) - 1.f) < EPS) ? 1.f : 0.f;
    const float dn = (fabsf(R_(2, 2) + 1.f) < EPS) ? 1.f : 0.f;
    const float reg = (1.f - up) * (1.f - dn);
    g1 = reg * g1 + up * __shfl(at, 3) + dn * __shfl(at, 4);
    g3 = reg * g3;
#undef R_
    // (cos, sin)(m angle), m = 1, 2, of the three angles: six sincosf in lanes 0..5
    float cs1[3][2], cs2[3][2], cs3[3][2];
    {
        const float ga = lane < 2 ? g1 : lane < 4 ? g2 : g3;
        float sn, cn;
        sincosf((lane & 1) ? 2.f * ga : ga, &sn, &cn);
        auto fill = [&](int l0, float (&cs)[3][2]) {
            cs[0][0] = 1.f; cs[0][1] = 0.f;
            cs[1][0] = __shfl(cn, l0); cs[1][1] = __shfl(sn, l0);
            cs[2][0] = __shfl(cn, l0 + 1); cs[2][1] = __shfl(sn, l0 + 1);
        };
        fill(0, cs1); fill(2, cs2); fill(4, cs3);
    }
    if (lane < 3) {                       // D^1 column `lane`
        float col[3];
        wigner_column<3>(kJ1, cs1, cs2, cs3, lane, col);
#pragma unroll
        for (int r = 0; r < 3; ++r) o[GTA_VREP_D1 + r * 3 + lane] = col[r];
    } else if (L >= 2 && lane >= 8 && lane < 13) {   // D^2 column `lane - 8`
        float col[5];
        wigner_column<5>(kJ2, cs1, cs2, cs3, lane - 8, col);
#pragma unroll
        for (int r = 0; r < 5; ++r) o[GTA_VREP_D2 + r * 5 + (lane - 8)] = col[r];
    }
}

// theta_{t,c=2f+d} = (float)(max_freq_d * 2 pi) * (coord_d * freq_f), freq_f = 2^(f+1-F) or 1
// (gta.py:57-63: the double scalar is rounded to fp32 when it meets the fp32 tensor).
// One thread = one (token, frequency): both axes' (cos, sin) pairs, c = 2 f and 2 f + 1, in one 16-B store (two threads' worth of
// the former one-pair-per-thread grid: the launch is latency-sized, half the workgroups end it sooner).
__device__ __forceinline__ void so2_table_body(int block, const float* __restrict__ coord, int n_tokens, int F,
                                               float k_h, float k_w, int shared, float* __restrict__ cs) {
    const int i = block * 256 + threadIdx.x;
    if (i >= n_tokens * F) return;
    const int t = i / F, f = i - t * F;
    const float freq = shared ? 1.f : exp2f((float)(f + 1 - F));
    const float2 xy = *reinterpret_cast<const float2*>(coord + 2 * t);
    const float th0 = k_h * (xy.x * freq), th1 = k_w * (xy.y * freq);
    float s0, c0, s1, c1;
    sincosf(th0, &s0, &c0);
    sincosf(th1, &s1, &c1);
    *reinterpret_cast<float4*>(cs + 4 * (long)i) = float4{c0, s0, c1, s1};      // blocks 2 f (row coordinate), 2 f + 1 (column coordinate)
}

__global__ __launch_bounds__(256) void build_view_reps_kernel(const float* __restrict__ E, int n_views, int L,
                                                              float* __restrict__ vrep) {
    view_reps_body(blockIdx.x, E, n_views, L, vrep);
}
__global__ __launch_bounds__(256) void build_so2_table_kernel(const float* __restrict__ coord, int n_tokens, int F,
                                                              float k_h, float k_w, int shared, float* __restrict__ cs) {
    so2_table_body(blockIdx.x, coord, n_tokens, F, k_h, k_w, shared, cs);
}
// both builders in one launch (they are independent and each is launch-latency sized): blocks [0, n_vb) build
// view records, the rest the SO(2) table
__global__ __launch_bounds__(256) void build_reps_kernel(const float* __restrict__ E, int n_views, int L,
                                                         float* __restrict__ vrep, int n_vb,
                                                         const float* __restrict__ coord, int n_tokens, int F,
                                                         float k_h, float k_w, int shared, float* __restrict__ cs) {
    if ((int)blockIdx.x < n_vb) view_reps_body(blockIdx.x, E, n_views, L, vrep);
    else so2_table_body(blockIdx.x - n_vb, coord, n_tokens, F, k_h, k_w, shared, cs);
}

}  // namespace

extern "C" int gta_build_view_reps(const float* extrinsics, int32_t n_views, int32_t so3_degree,
                                   float* vrep, void* stream) {
    if (!extrinsics || !vrep || n_views <= 0) return GTA_E_BADARG;
    if (so3_degree < 0 || so3_degree > 2) return GTA_E_UNSUPPORTED;
    hipLaunchKernelGGL(build_view_reps_kernel, dim3((n_views + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, extrinsics, n_views, so3_degree, vrep);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

extern "C" int gta_build_so2_table(const float* coord, int32_t n_tokens, int32_t nfreqs,
                                   float max_freq_h, float max_freq_w, int32_t shared_freqs,
                                   float* cs, void* stream) {
    if (!coord || !cs || n_tokens <= 0 || nfreqs <= 0) return GTA_E_BADARG;
    if (((uintptr_t)coord & 7) || ((uintptr_t)cs & 15)) return GTA_E_BADARG;      // float2 loads of coord, float4 stores of cs (gta_hip.h)
    const double two_pi = 6.283185307179586;
    const float k_h = (float)((double)max_freq_h * two_pi);
    const float k_w = (float)((double)max_freq_w * two_pi);
    const long total = (long)n_tokens * nfreqs;         // one thread per (token, frequency)
    const int th = 256;                  // (so2_table_body assumes 256-thread blocks)
    hipLaunchKernelGGL(build_so2_table_kernel, dim3((unsigned)((total + th - 1) / th)), dim3(th), 0,
                       (hipStream_t)stream, coord, n_tokens, nfreqs, k_h, k_w, shared_freqs, cs);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

extern "C" int gta_build_reps(const float* extrinsics, int32_t n_views, int32_t so3_degree, float* vrep,
                              const float* coord, int32_t n_tokens, int32_t nfreqs, float max_freq_h,
                              float max_freq_w, int32_t shared_freqs, float* cs, void* stream) {
    if (!extrinsics || !vrep || n_views <= 0 || !coord || !cs || n_tokens <= 0 || nfreqs <= 0) return GTA_E_BADARG;
    if (((uintptr_t)coord & 7) || ((uintptr_t)cs & 15)) return GTA_E_BADARG;      // float2 loads of coord, float4 stores of cs (gta_hip.h)
    if (so3_degree < 0 || so3_degree > 2) return GTA_E_UNSUPPORTED;
    const double two_pi = 6.283185307179586;
    const float k_h = (float)((double)max_freq_h * two_pi);
    const float k_w = (float)((double)max_freq_w * two_pi);
    const int n_vb = (n_views + 3) / 4;
    const long total = (long)n_tokens * nfreqs;         // one thread per (token, frequency)
    const long n_tb = (total + 255) / 256;
    hipLaunchKernelGGL(build_reps_kernel, dim3((unsigned)(n_vb + n_tb)), dim3(256), 0, (hipStream_t)stream, extrinsics,
                       n_views, so3_degree, vrep, n_vb, coord, n_tokens, nfreqs, k_h, k_w, shared_freqs, cs);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}
