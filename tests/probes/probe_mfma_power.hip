// Probe (test infrastructure; not part of any library): the ENERGY floor of the headline attention launch's matrix work.
//
// `gta_attn64_items_kernel` at the MSN-encoder shape (B = 32, H = 8, Tq = Tk = 1280, dh = 96) issues, per launch,
// 256 workgroups x 4 waves x 5 items x 960 v_mfma_f32_32x32x16_bf16 (= 4.9 M matrix instructions, 161 GFLOP).  This kernel issues exactly
// that -- same grid, same one-wave-per-SIMD shape, 512 registers per wave -- and NOTHING else: operands are random bf16 fragments that sit
// in registers, six independent accumulators so no instruction waits for the one before it.  With GAP > 0 an `s_nop` of GAP cycles follows
// every matrix instruction: the same work at a lower duty cycle of the matrix pipe (GAP = 16 -> 32 / 48 = 0.67 busy, the full kernel's
// figure), which tells "the part is held by its power budget" (time does not move with GAP, the granted clock does) from "the pipe is
// simply not kept busy" (time scales with 32 + GAP).  ZERO = all-zero operands (no data toggling in the multipliers).
//
// Every workgroup leaves its s_memtime / s_memrealtime stamps, so the launch's shader cycles and the granted clock come out as
// bench.py derives them for the product kernel.  tools/probe_mfma_power.py drives it: alone, and inside the bench step (between the
// real rep-build + K/V pre-pass launches).
//     hipcc -O3 --offload-arch=gfx950 -shared -fPIC tests/probes/probe_mfma_power.hip -o tests/probes/libprobe_mfma_power.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

namespace {
constexpr int NFRAG = 12;         // operand fragments per side held in registers (the real loop: 12 K' + 12 Q' fragments per tile)
constexpr int NACC = 6;           // independent accumulators (the real loop: 6 O^T blocks per wave)
constexpr int PER_ITEM = 960;     // matrix instructions per 256-row item and wave: 20 key tiles x 48

template <int GAP>
__device__ __forceinline__ void gap() {
    if constexpr (GAP >= 16) { asm volatile("s_nop 15"); gap<GAP - 16>(); }
    else if constexpr (GAP > 0) asm volatile("s_nop %0" ::"n"(GAP - 1));
}

template <int GAP>
__global__ __launch_bounds__(256, 1) void mfma_power_kernel(const uint32_t* __restrict__ rnd, float* __restrict__ sink,
                                                            unsigned long long* __restrict__ stamps, int items, int zero) {
    const int tid = threadIdx.x;
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    u32x4_t a[NFRAG], b[NFRAG];
    // random bf16 values in [-2, 2) (sign + exponent 0x3f / 0x3e.. from the random bits' low part): no NaN / Inf, full mantissa toggling
#pragma unroll
    for (int f = 0; f < NFRAG; ++f) {
        u32x4_t ra = reinterpret_cast<const u32x4_t*>(rnd)[(f * 256 + tid) * 2 + 0];
        u32x4_t rb = reinterpret_cast<const u32x4_t*>(rnd)[(f * 256 + tid) * 2 + 1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = zero ? 0u : ((ra[i] & 0x80ff80ffu) | 0x3f003f00u);
            rb[i] = zero ? 0u : ((rb[i] & 0x80ff80ffu) | 0x3f003f00u);
        }
        a[f] = ra; b[f] = rb;
    }
    f32x16_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < items; ++it) {
        for (int tile = 0; tile < PER_ITEM / 48; ++tile) {
#pragma unroll
            for (int g = 0; g < 48; ++g) {
                acc[g % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[g % NFRAG]),
                                                                        __builtin_bit_cast(bf16x8_t, b[(g * 5 + 1) % NFRAG]), acc[g % NACC], 0, 0, 0);
                gap<GAP>();
            }
            // keep the accumulators bounded (random products sum to ~sqrt(n)): one cheap rescale per tile would add VALU work -- instead
            // the operands' signs make the sums a random walk, and fp32 has the range for 4 800 steps of |x| < 64
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[tid] = s;                  // (keeps the chain alive; never true in practice)
    __syncthreads();
    if (tid == 0) {
        stamps[blockIdx.x * 4 + 0] = t0;
        stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
        stamps[blockIdx.x * 4 + 2] = r0;
        stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}
}  // namespace

// C entry for the Python driver (ctypes): launch on `stream`; gap in {0, 4, 8, 16, 32}; returns 0 or a HIP error code
extern "C" int probe_mfma_power_launch(const void* rnd, void* sink, void* stamps, int grid, int items, int gap_cycles, int zero, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const uint32_t* r = (const uint32_t*)rnd;
    float* sk = (float*)sink;
    unsigned long long* sp = (unsigned long long*)stamps;
    switch (gap_cycles) {
        case 0: hipLaunchKernelGGL(mfma_power_kernel<0>, dim3(grid), dim3(256), 0, st, r, sk, sp, items, zero); break;
        case 4: hipLaunchKernelGGL(mfma_power_kernel<4>, dim3(grid), dim3(256), 0, st, r, sk, sp, items, zero); break;
        case 8: hipLaunchKernelGGL(mfma_power_kernel<8>, dim3(grid), dim3(256), 0, st, r, sk, sp, items, zero); break;
        case 16: hipLaunchKernelGGL(mfma_power_kernel<16>, dim3(grid), dim3(256), 0, st, r, sk, sp, items, zero); break;
        case 32: hipLaunchKernelGGL(mfma_power_kernel<32>, dim3(grid), dim3(256), 0, st, r, sk, sp, items, zero); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
extern "C" int probe_mfma_power_rnd_words(void) { return NFRAG * 256 * 2 * 4; }
