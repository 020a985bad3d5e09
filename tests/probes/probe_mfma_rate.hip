// Probe (test infrastructure, standalone executable): cycles per v_mfma_f32_32x32x16_bf16 and SIMD for back-to-back
// streams of independent MFMAs -- NACC accumulators per wave used round-robin, WAVES waves per workgroup (one workgroup per
// CU), with or without a barrier every 16 MFMAs.  hipcc --offload-arch=gfx950 -O3 probe_mfma_rate.hip -o probe_mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

template <int NACC, bool BAR>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* cyc, int iters, uint32_t seed) {
    f32x16_t acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u32x4_t av = {seed, seed + threadIdx.x, seed * 3u, 0x3f803f80u}, bv = {0x3f803f80u, seed ^ 0x55u, threadIdx.x, seed};
    const bf16x8_t A = __builtin_bit_cast(bf16x8_t, av), B = __builtin_bit_cast(bf16x8_t, bv);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[k % NACC], 0, 0, 0);
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[256 + blockIdx.x] = r1 - r0; }
}

template <int NACC, bool BAR>
void run(int waves, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC, BAR>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, 12345u);     // warm-up
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NACC, BAR>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, 12345u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ev_ms = 0.f;
    hipEventElapsedTime(&ev_ms, e0, e1);
    unsigned long long h[512];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, rt = 0;
    for (int i = 0; i < 256; ++i) { m += (double)h[i]; rt += (double)h[256 + i]; }
    m /= 256; rt /= 256;      // rt: 100-MHz ticks
    const double per_simd = (double)iters * 16 * waves / 4.0;          // MFMAs each SIMD executed
    const double us = rt * 0.01, flops = 256.0 * waves * iters * 16 * 32768.0;
    printf("waves/WG %d (%.0f per SIMD)  accumulators %2d  barrier %d : %7.1f ticks per MFMA and wave, %6.1f per MFMA and SIMD; %7.1f us, s_memtime %.2f GHz, %7.0f TFLOP/s (host events: %7.1f us = %6.0f TFLOP/s)\n",
           waves, waves / 4.0, NACC, (int)BAR, m / (iters * 16.0), m / per_simd, us, m / (rt * 10.0), flops / us * 1e-6, ev_ms * 1e3, flops / (ev_ms * 1e3) * 1e-6);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 512 * 8);
    for (int waves : {4, 8}) {
        run<1, false>(waves, out, cyc); run<2, false>(waves, out, cyc); run<4, false>(waves, out, cyc);
    }
    return 0;
}
