// gta_reps.hip -- rep builders on device (replace encoder.py:183-265 / decoder.py:247-353).
//
// Tiny kernels (B*N views, B*T tokens): HBM traffic is a few hundred KB per forward and they run
// once per encoder/decoder call, shared by all layers exactly like the reference's
// pre_compute_reps output.  One thread per view / per (token, block); nothing to tile.
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

// Z(angle) of wigner_d.py:16-25: cos(m a) on the diagonal, sin(m a) on the anti-diagonal,
// m = l..-l; the diagonal is written last (centre element = cos 0 = 1).
template <int N>
__device__ __forceinline__ void z_rot(float a, float* Z) {
    constexpr int l = (N - 1) / 2;
#pragma unroll
    for (int i = 0; i < N * N; ++i) Z[i] = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) Z[i * N + (N - 1 - i)] = sinf((float)(l - i) * a);
#pragma unroll
    for (int i = 0; i < N; ++i) Z[i * N + i] = cosf((float)(l - i) * a);
}
template <int N>
__device__ __forceinline__ void matmul(const float* A, const float* B, float* C) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < N; ++k) s += A[i * N + k] * B[k * N + j];
            C[i * N + j] = s;
        }
}
// D = Z(g3) J Z(g2) J Z(g1)   (wigner_d.py:28-35)
template <int N>
__device__ __forceinline__ void wigner(const float* J, float g1, float g2, float g3, float* D) {
    float Z[N * N], T0[N * N], T1[N * N];
    z_rot<N>(g3, Z);
    matmul<N>(Z, J, T0);
    z_rot<N>(g2, Z);
    matmul<N>(T0, Z, T1);
    matmul<N>(T1, J, T0);
    z_rot<N>(g1, Z);
    matmul<N>(T0, Z, D);
}

// General 4x4 inverse, Gauss-Jordan with partial pivoting in fp64 (the reference calls
// torch.linalg.inv, encoder.py:219 -- a general inverse, not the rigid closed form).
__device__ __forceinline__ void inv4(const float* E, float* out) {
    double a[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[i][j] = (double)E[i * 4 + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // partial pivoting by compare-and-swap with every lower row (all indices static: registers)
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            const bool sw = fabs(a[r][c]) > fabs(a[c][c]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double x = a[c][j], y = a[r][j];
                a[c][j] = sw ? y : x;
                a[r][j] = sw ? x : y;
            }
        }
        const double inv = 1.0 / a[c][c];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[c][j] *= inv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) out[i * 4 + j] = (float)a[i][4 + j];
}

// three threads per view: role 0 writes E and inverse(E), role 1 D^1, role 2 D^2 (each recomputes the
// cheap inverse; the trig-heavy Wigner products run side by side instead of back to back)
__global__ void build_view_reps_kernel(const float* __restrict__ E, int n_views, int L,
                                       float* __restrict__ vrep) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gid / 3, role = gid - 3 * i;
    if (i >= n_views) return;
    const float* e = E + (size_t)i * 16;
    float* o = vrep + (size_t)i * GTA_VREP_STRIDE;
    float inv[16];
    inv4(e, inv);
    if (role == 0) {
        for (int j = 0; j < 16; ++j) { o[GTA_VREP_INV + j] = e[j]; o[GTA_VREP_REP + j] = inv[j]; }
        for (int j = GTA_VREP_D2 + 25; j < GTA_VREP_STRIDE; ++j) o[j] = 0.f;
        if (L < 1) for (int j = GTA_VREP_D1; j < GTA_VREP_D2; ++j) o[j] = 0.f;
        if (L < 2) for (int j = GTA_VREP_D2; j < GTA_VREP_D2 + 25; ++j) o[j] = 0.f;
        return;
    }
    if (role > L) return;
    // R = inverse(E)[:3,:3]; ZYZ angles with the reference's gimbal masks (wigner_d.py:39-49)
#define R_(r, c) inv[(r) * 4 + (c)]
    const float EPS = 1e-5f;
    float g1 = atan2f(R_(2, 1), -R_(2, 0));
    const float g2 = atan2f(sqrtf(R_(0, 2) * R_(0, 2) + R_(1, 2) * R_(1, 2)), R_(2, 2));
    float g3 = atan2f(R_(1, 2), R_(0, 2));
    const float up = (fabsf(R_(2, 2) - 1.f) < EPS) ? 1.f : 0.f;
    const float dn = (fabsf(R_(2, 2) + 1.f) < EPS) ? 1.f : 0.f;
    const float reg = (1.f - up) * (1.f - dn);
    g1 = reg * g1 + up * atan2f(R_(1, 0), R_(0, 0)) + dn * atan2f(-R_(1, 0), -R_(0, 0));
    g3 = reg * g3;
#undef R_
    if (role == 1) {
        float D1[9];
        wigner<3>(kJ1, g1, g2, g3, D1);
#pragma unroll
        for (int j = 0; j < 9; ++j) o[GTA_VREP_D1 + j] = D1[j];
    } else {
        float D2[25];
        wigner<5>(kJ2, g1, g2, g3, D2);
#pragma unroll
        for (int j = 0; j < 25; ++j) o[GTA_VREP_D2 + j] = D2[j];
    }
}

// theta_{t,c=2f+d} = (float)(max_freq_d * 2 pi) * (coord_d * freq_f), freq_f = 2^(f+1-F) or 1
// (gta.py:57-63: the double scalar is rounded to fp32 when it meets the fp32 tensor).
__global__ void build_so2_table_kernel(const float* __restrict__ coord, int n_tokens, int F,
                                       float k_h, float k_w, int shared, float* __restrict__ cs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nblk = 2 * F;
    if (i >= n_tokens * nblk) return;
    const int t = i / nblk, c = i - t * nblk;
    const int f = c >> 1, d = c & 1;
    const float freq = shared ? 1.f : exp2f((float)(f + 1 - F));
    const float prod = coord[2 * t + d] * freq;
    const float th = (d == 0 ? k_h : k_w) * prod;
    float s, co;
    sincosf(th, &s, &co);
    cs[2 * i] = co;
    cs[2 * i + 1] = s;
}

}  // namespace

extern "C" int gta_build_view_reps(const float* extrinsics, int32_t n_views, int32_t so3_degree,
                                   float* vrep, void* stream) {
    if (!extrinsics || !vrep || n_views <= 0) return GTA_E_BADARG;
    if (so3_degree < 0 || so3_degree > 2) return GTA_E_UNSUPPORTED;
    const int th = 64;
    hipLaunchKernelGGL(build_view_reps_kernel, dim3((3 * n_views + th - 1) / th), dim3(th), 0,
                       (hipStream_t)stream, extrinsics, n_views, so3_degree, vrep);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

extern "C" int gta_build_so2_table(const float* coord, int32_t n_tokens, int32_t nfreqs,
                                   float max_freq_h, float max_freq_w, int32_t shared_freqs,
                                   float* cs, void* stream) {
    if (!coord || !cs || n_tokens <= 0 || nfreqs <= 0) return GTA_E_BADARG;
    const double two_pi = 6.283185307179586;
    const float k_h = (float)((double)max_freq_h * two_pi);
    const float k_w = (float)((double)max_freq_w * two_pi);
    const long total = (long)n_tokens * 2 * nfreqs;
    const int th = 256;
    hipLaunchKernelGGL(build_so2_table_kernel, dim3((unsigned)((total + th - 1) / th)), dim3(th), 0,
                       (hipStream_t)stream, coord, n_tokens, nfreqs, k_h, k_w, shared_freqs, cs);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}
