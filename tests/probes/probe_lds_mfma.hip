// Probe (test infrastructure, standalone; not part of any library): does LDS fragment-read traffic overlap with the matrix
// pipe?  gta_wgrad.hip's switch experiments say its transpose-reads ADD to the MFMA time instead of hiding under it.  This
// kernel runs the same per-step stream -- NM MFMAs and NR pairs of LDS reads per block, reads placed in the gaps between the
// MFMAs -- with either part switched off, for ds_read_b64_tr_b16 pairs and for ds_read_b128, with 4 or 8 waves per CU:
//     hipcc -O3 --offload-arch=gfx950 tests/probes/probe_lds_mfma.hip -o /tmp/probe_lds_mfma && /tmp/probe_lds_mfma
// prints microseconds per variant (host events), the LDS bytes read per CU and nanosecond, and the MFMA rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

template <int IMM>
__device__ __forceinline__ u32x2_t tr16(uint32_t addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(IMM));
    return v;
}
template <int IMM>
__device__ __forceinline__ u32x4_t rd128(uint32_t addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(IMM));
    return v;
}

// JB = X fragments per wave (2: the 128 x 64 wave tile, 8 waves;  4: 128 x 128, 4 waves).  MODE bit 0: MFMAs, bit 1: reads.
// FORM 0: pairs of ds_read_b64_tr_b16 (the kernel's), 1: one ds_read_b128 per fragment (same bytes, half the instructions).
template <int JB, int MODE, int FORM>
__global__ __launch_bounds__(JB == 4 ? 256 : 512) void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];          // one 32-KiB stage: [2 operands][32 tokens][256 columns] bf16
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WK = JB == 4 ? 2 : 4, NF = 4 + JB, NM = 4 * JB;
    const int wn = wave / WK, wk = wave % WK;
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    // the kernel's conflict-free lane addresses (gta_wgrad.hip)
    const int grp = lane >> 4, pl = lane & 15, rrow = pl >> 2;
    const int chunk0 = ((grp & 1) * 2 + ((pl & 3) >> 1)) ^ (((rrow >> 1) & 1) << 2);
    const uint32_t la0 = (uint32_t)((lane >> 5) * 4096 + rrow * 128 + chunk0 * 16 + (pl & 1) * 8);
    const uint32_t lb = FORM ? (uint32_t)(lane * 16) : la0;                // b128: lane-linear 1-KiB pieces (conflict-free)
    const uint32_t a_base = lds0 + wn * 2048 + lb, b_base = lds0 + 16384 + wk * (JB / 2) * 1024 + lb;

    f32x16_t acc[4][JB];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < JB; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    u32x4_t fr[2][NF];
    for (int b = 0; b < 2; ++b) for (int f = 0; f < NF; ++f) fr[b][f] = u32x4_t{0x3f803f80u, (uint32_t)lane, 0x3f803f80u, (uint32_t)f};

    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {                                // two 16-token blocks per step, double-buffered fragments
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NM; ++g) {
                if (MODE & 1) {
                    const int i = g / JB, j = g % JB;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fr[blk][i]),
                                                                        __builtin_bit_cast(bf16x8_t, fr[blk][4 + j]), acc[i][j], 0, 0, 0);
                }
                if ((MODE & 2) && g < NF) {
                    const uint32_t base = (g < 4 ? a_base : b_base) ^ ((g & 1) ? 64u : 0u);
                    if (FORM == 0) {
                        const u32x2_t lo = tr16<0>(base + blk * 8192 + ((g & 3) >> 1) * 1024);
                        const u32x2_t hi = tr16<512>(base + blk * 8192 + ((g & 3) >> 1) * 1024);
                        fr[blk ^ 1][g] = u32x4_t{lo.x, lo.y, hi.x, hi.y};
                    } else {
                        fr[blk ^ 1][g] = rd128<0>(base + blk * 8192 + ((g & 3) >> 1) * 1024);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(MODE & 2)) asm volatile("" : "+v"(fr[0][0]), "+v"(fr[1][0]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < JB; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    for (int f = 0; f < NF; ++f) s += (float)(fr[0][f].x ^ fr[1][f].w);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int JB, int MODE, int FORM>
void run(float* out) {
    const int iters = 2000, threads = JB == 4 ? 256 : 512, waves = threads / 64;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<JB, MODE, FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL((probe<JB, MODE, FORM>), dim3(256), dim3(threads), 131072, 0, out, iters);     // 128 KiB: one workgroup per CU, as the kernel
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<JB, MODE, FORM>), dim3(256), dim3(threads), 131072, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3;
    const double mfma = (MODE & 1) ? 256.0 * waves * iters * 2 * (4 * JB) * 32768.0 / us * 1e-6 : 0.0;          // TFLOP/s
    const double ldsb = (MODE & 2) ? (double)waves * iters * 2 * (4 + JB) * 1024.0 / (us * 1e3) : 0.0;          // bytes per CU and ns
    printf("%d waves/CU (%3d x %3d per wave), %-9s %-12s: %8.1f us  %6.0f TFLOP/s  %6.1f LDS B/ns/CU\n", waves, 128, 32 * JB,
           MODE == 1 ? "MFMA" : MODE == 2 ? "reads" : "MFMA+reads", FORM ? "ds_read_b128" : "b64_tr pairs", us, mfma, ldsb);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    run<2, 1, 0>(out); run<2, 2, 0>(out); run<2, 3, 0>(out); run<2, 2, 1>(out); run<2, 3, 1>(out);
    run<4, 1, 0>(out); run<4, 2, 0>(out); run<4, 3, 0>(out); run<4, 2, 1>(out); run<4, 3, 1>(out);
    return 0;
}
