// Probe (test infrastructure, standalone): the MFMA stream of a 128 x 64 wave tile (4 A fragments x 2 B fragments = 8
// accumulators of 32 x 32) as gta_wgrad.hip issues it, in different orders, 1 or 2 waves per SIMD: cycles per MFMA and SIMD
// from host events (the in-kernel stamps of one wave mislead: the older wave of a SIMD wins every arbitration).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

template <int ORDER, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(float* out, int iters, uint32_t seed) {
    f32x16_t acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t A[4], B[2];
    for (int i = 0; i < 4; ++i) { u32x4_t w = {seed + i, seed * (i + 2) + threadIdx.x, 0x3f803f80u, seed ^ (i * 77u)}; A[i] = __builtin_bit_cast(bf16x8_t, w); }
    for (int j = 0; j < 2; ++j) { u32x4_t w = {seed * 5u + j, threadIdx.x + j, 0x3f803f80u, seed + 9u * j}; B[j] = __builtin_bit_cast(bf16x8_t, w); }
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {                                  // i-major (gta_wgrad.hip): (0,0),(0,1),(1,0),...
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], B[j], acc[i][j], 0, 0, 0);
        } else {                                           // j-major
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], B[j], acc[i][j], 0, 0, 0);
        }
        // the fragments change between blocks in the real kernel: keep the compiler from hoisting anything
        asm volatile("" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(B[0]), "+v"(B[1]));
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ORDER, int WAVES>
void run(float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<ORDER, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, iters, 12345u);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<ORDER, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, iters, 12345u);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * WAVES * iters * 8 * 32768.0;
    printf("order %s, %d waves per SIMD: %7.1f us = %6.0f TFLOP/s\n", ORDER == 0 ? "i-major" : "j-major", WAVES / 4, ms * 1e3, flops / (ms * 1e3) * 1e-6);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    run<0, 4>(out); run<1, 4>(out); run<0, 8>(out); run<1, 8>(out);
    return 0;
}
