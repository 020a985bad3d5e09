// gta_wgrad.hip -- weight gradient of a Linear as a hand-written gfx950 kernel (gta_wgrad of include/gta_block.h):
//
//     dW[n][k] = sum_m G[m][n] * X[m][k]      G = d(out) [M,N] bf16, X = the layer's input [M,K] bf16, dW fp32
//     db[n]    = sum_m G[m][n]                (optional; the bias gradient falls out of the G fragments)
//
// This is the "TN" GEMM of the block's backward (autograd over nn.Linear, source/layers.py:161-164,388-395,430): the
// reduction runs over the M = batch x tokens rows, the SLOW index of both row-major operands, and the outputs are small
// (768..2304 squared).  hipBLASLt's best kernel for it reaches 370-540 TFLOP/s at the MSN shapes (profiles/r02/README.md).
//
// Structure.  One workgroup (4 waves, 256 threads, one per CU) owns a 256 x 256 tile of dW over a slice of the tokens
// (split-M, partial tiles reduced in a fixed order by wgrad_finish_kernel: deterministic).  Per step of 64 tokens the two
// operand tiles [64][256] are brought in by LDS-DMA (global_load_lds, 16 B per lane) as [16 token][32 column] sub-tiles of
// 1 KiB -- one DMA instruction each: lane l fetches 16 B of row l/4 at column chunk l%4, i.e. 64-B row segments, and the
// lane-linear LDS image of that instruction IS the row-major sub-tile with 64-B rows.  MFMA operands need 8 consecutive
// TOKENS of one column per lane; ds_read_b64_tr_b16 delivers 4 (a 16-lane group reads a [4 token][16 column] block and
// transposes it), so an operand of v_mfma_f32_32x32x16_bf16 is two such reads.  With 64-B rows the 32 lanes serviced
// together read 4 rows x 64 B = 256 contiguous bytes: conflict-free.  Two LDS stages of 64 KiB: the DMA of step t+1 runs
// under the MFMAs of step t (one barrier per step).  Each wave owns 128 x 128 of the tile: 16 accumulators of 32 x 32
// (256 accumulator registers), 64 MFMAs against 64 transpose-reads per step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gta_common.h"
#include "../../include/gta_hip.h"
#include "../../include/gta_block.h"

namespace {

constexpr int WG_TILE = 256;                       // dW tile edge (both ways)
constexpr int WG_BT = 64;                          // tokens per step
constexpr int WG_STAGE = 2 * WG_BT * WG_TILE * 2;  // G tile + X tile, bytes (64 KiB)
constexpr int WG_LDS = 2 * WG_STAGE;

struct WgradParams {
    const char* g;       // [M][ldg] bf16
    const char* x;       // [M][ldx] bf16
    float* part;         // [S][N][K] partial tiles (or the final dW when S == 1, row pitch ldo)
    float* db_part;      // [S][N] or nullptr
    long ldg, ldx, ldo;  // elements
    int M, N, K;
    int tiles_k;         // K / 256
    int n_tiles;         // (N / 256) * tiles_k
    int steps_per_split; // 64-token steps per split (the last split may have fewer)
};

template <int IMM>
GTA_DEV u32x2_t tr16(uint32_t addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(IMM));
    return v;
}

GTA_DEV float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
GTA_DEV float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

template <bool BIAS>
__global__ __launch_bounds__(256, 1) void wgrad_kernel(WgradParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    // XCD-aware work map: workgroup L runs on XCD L % 8.  Workgroups of one token split share their operand tiles (the G
    // tile of a row of dW tiles, the X tile of a column), so each XCD takes a CONTIGUOUS run of the (split, tile) list and
    // its L2 serves the re-reads (measured: the kernel is bound by the fabric otherwise, 4.8 TB/s into the LDS).
    const int total = gridDim.x, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + jx;
    const int tile = v % p.n_tiles, split = v / p.n_tiles;
    const int n0 = (tile / p.tiles_k) * WG_TILE, k0 = (tile % p.tiles_k) * WG_TILE;
    const int step0 = split * p.steps_per_split;
    int n_steps = p.M / WG_BT - step0;
    if (n_steps > p.steps_per_split) n_steps = p.steps_per_split;

    // ---- DMA plan: 64 sub-tiles of 1 KiB per stage; waves 0,1 bring G (32 sub-tiles), waves 2,3 bring X ----
    const bool is_x = wave >= 2;
    const char* src = is_x ? p.x : p.g;
    const long ld = is_x ? p.ldx : p.ldg;
    const int c0 = is_x ? k0 : n0;
    const int piece0 = (wave & 1) * 16;                                   // this wave's 16 sub-tiles of its operand
    // lane's byte offset inside a sub-tile's source: row lane/4, 16-B chunk lane%4
    const long lane_src = ((long)(lane >> 2) * ld + (lane & 3) * 8) * 2;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;

    auto stage_dma = [&](int step, int stage) {
        const long m0 = (long)(step0 + step) * WG_BT;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int pc = piece0 + q, tg = pc >> 3, cg = pc & 7;
            const char* gp = src + ((m0 + tg * 16) * ld + c0 + cg * 32) * 2 + lane_src;
            char* lp = smem + stage * WG_STAGE + (is_x ? WG_BT * WG_TILE * 2 : 0) + pc * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                             (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
        }
    };

    // ---- operand addresses: lane -> its [4 token][16 column] block inside a sub-tile (64-B rows) ----
    const uint32_t la = (uint32_t)(((lane >> 5) * 8 + ((lane & 15) >> 2)) * 64 + (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2);
    const uint32_t a_base = lds0 + wn * 4096 + la;                         // G sub-tiles wn*4 .. wn*4+3 of a token group
    const uint32_t b_base = lds0 + WG_BT * WG_TILE * 2 + wk * 4096 + la;   // X sub-tiles wk*4 ..

    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float dbs[4] = {0.f, 0.f, 0.f, 0.f};

    if (n_steps > 0) stage_dma(0, 0);

    for (int t = 0; t < n_steps; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // my pieces of step t have landed
        __builtin_amdgcn_s_barrier();                                       // everyone's have; stage (t+1)&1 is free again
        if (t + 1 < n_steps) stage_dma(t + 1, (t + 1) & 1);
        const uint32_t ab = a_base + (t & 1) * WG_STAGE, bb = b_base + (t & 1) * WG_STAGE;

        u32x2_t af[2][4][2], bf[2][4][2];                                   // [buffer][32-column group][token half]
        auto load_frags = [&](auto TG, int buf) {
            constexpr int tg = decltype(TG)::value;
            gta_static_for<4>([&](auto I) {
                constexpr int i = decltype(I)::value;
                af[buf][i][0] = tr16<(tg * 8 + i) * 1024>(ab);
                af[buf][i][1] = tr16<(tg * 8 + i) * 1024 + 256>(ab);
            });
            gta_static_for<4>([&](auto J) {
                constexpr int j = decltype(J)::value;
                bf[buf][j][0] = tr16<(tg * 8 + j) * 1024>(bb);
                bf[buf][j][1] = tr16<(tg * 8 + j) * 1024 + 256>(bb);
            });
        };
        load_frags(std::integral_constant<int, 0>{}, 0);
        gta_static_for<4>([&](auto TG) {
            constexpr int tg = decltype(TG)::value, cur = tg & 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (tg < 3) load_frags(std::integral_constant<int, tg + 1>{}, cur ^ 1);
            bf16x8_t a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4_t av = {af[cur][i][0].x, af[cur][i][0].y, af[cur][i][1].x, af[cur][i][1].y};
                const u32x4_t bv = {bf[cur][i][0].x, bf[cur][i][0].y, bf[cur][i][1].x, bf[cur][i][1].y};
                a[i] = __builtin_bit_cast(bf16x8_t, av);
                b[i] = __builtin_bit_cast(bf16x8_t, bv);
                if (BIAS && wk == 0 && k0 == 0)
                    dbs[i] += ((bf16_lo(av.x) + bf16_hi(av.x)) + (bf16_lo(av.y) + bf16_hi(av.y))) +
                              ((bf16_lo(av.z) + bf16_hi(av.z)) + (bf16_lo(av.w) + bf16_hi(av.w)));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        });
    }

    // ---- epilogue: accumulators -> this split's partial tile (rows n, 128-B segments along k) ----
    float* out = p.part + (long)split * p.N * p.ldo;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 128 + i * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                const int k = k0 + wk * 128 + j * 32 + (lane & 31);
                out[(long)n * p.ldo + k] = acc[i][j][r];
            }
    if (BIAS && wk == 0 && k0 == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = dbs[i] + __shfl_xor(dbs[i], 32, 64);           // the two token halves of the fragment
            if (lane < 32) p.db_part[(long)split * p.N + n0 + wn * 128 + i * 32 + lane] = s;
        }
    }
}

// dW[n][k] = sum_s part[s][n][k] (fixed order), db[n] = sum_s db_part[s][n]
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ part, int S, long slab, long n4,
                                                          float* __restrict__ dw, const float* __restrict__ db_part,
                                                          float* __restrict__ db, int N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        f32x4_t s = reinterpret_cast<const f32x4_t*>(part)[i];
        for (int q = 1; q < S; ++q) {
            const f32x4_t v = reinterpret_cast<const f32x4_t*>(part + q * slab)[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<f32x4_t*>(dw)[i] = s;
    }
    if (db && i < N) {
        float s = db_part[i];
        for (int q = 1; q < S; ++q) s += db_part[(long)q * N + i];
        db[i] = s;
    }
}

struct Split { int S, steps; };
inline Split choose_split(long m, long n, long k) {
    const int tiles = (int)((n / WG_TILE) * (k / WG_TILE));
    const int total = (int)(m / WG_BT);
    int want = 256 / tiles;                     // one workgroup per CU
    if (want < 1) want = 1;
    if (want > total / 4) want = total / 4 > 0 ? total / 4 : 1;            // at least four steps per workgroup
    const int steps = (total + want - 1) / want;
    return {(total + steps - 1) / steps, steps};
}

}  // namespace

extern "C" {

int gta_wgrad_supported(int64_t m, int64_t n, int64_t k) {
    return (m > 0 && n > 0 && k > 0 && m % WG_BT == 0 && n % WG_TILE == 0 && k % WG_TILE == 0 && m <= (1ll << 30)) ? 1 : 0;
}

int64_t gta_wgrad_workspace_bytes(int64_t m, int64_t n, int64_t k) {
    if (!gta_wgrad_supported(m, n, k)) return 0;
    const Split sp = choose_split(m, n, k);
    return (int64_t)sp.S * (n * k + n) * (int64_t)sizeof(float);
}

int gta_wgrad(const void* g, int64_t ldg, const void* x, int64_t ldx, int64_t m, int64_t n, int64_t k, float* dw, float* dbias,
              void* workspace, int64_t workspace_bytes, void* stream) {
    if (!g || !x || !dw || !workspace || ldg < n || ldx < k) return GTA_E_BADARG;
    if (!gta_wgrad_supported(m, n, k)) return GTA_E_UNSUPPORTED;
    if (ldg % 8 != 0 || ldx % 8 != 0 || (reinterpret_cast<uintptr_t>(g) & 15) || (reinterpret_cast<uintptr_t>(x) & 15) ||
        (reinterpret_cast<uintptr_t>(dw) & 15))
        return GTA_E_BADARG;
    if (workspace_bytes < gta_wgrad_workspace_bytes(m, n, k)) return GTA_E_BADARG;
    const Split sp = choose_split(m, n, k);
    WgradParams p;
    p.g = static_cast<const char*>(g);
    p.x = static_cast<const char*>(x);
    p.part = static_cast<float*>(workspace);
    p.db_part = dbias ? p.part + (long)sp.S * n * k : nullptr;
    p.ldg = ldg; p.ldx = ldx; p.ldo = k;
    p.M = (int)m; p.N = (int)n; p.K = (int)k;
    p.tiles_k = (int)(k / WG_TILE);
    p.n_tiles = (int)(n / WG_TILE) * p.tiles_k;
    p.steps_per_split = sp.steps;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = dbias ? gta_lds_optin<&wgrad_kernel<true>>(WG_LDS) : gta_lds_optin<&wgrad_kernel<false>>(WG_LDS);
    if (rc != 0) return rc;
    if (dbias) hipLaunchKernelGGL(wgrad_kernel<true>, dim3(p.n_tiles * sp.S), dim3(256), WG_LDS, s, p);
    else       hipLaunchKernelGGL(wgrad_kernel<false>, dim3(p.n_tiles * sp.S), dim3(256), WG_LDS, s, p);
    if (hipGetLastError() != hipSuccess) return GTA_E_LAUNCH;
    const long n4 = n * k / 4;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p.part, sp.S, (long)n * k, n4, dw,
                       p.db_part, dbias, (int)n);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

}  // extern "C"
