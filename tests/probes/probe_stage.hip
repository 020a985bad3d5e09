// probe_stage.hip -- what does it cost a wave to stage 6 KiB per loop iteration into LDS beside 24 MFMAs:
// (a) LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction), (b) global_load_dwordx4 into VGPRs + ds_write_b128
// one iteration later, (c) no staging.  gfx950; 256-thread workgroups, 1 or 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const char* __restrict__ src, uint64_t* out, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    bf16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(0.01f * i); B[i] = (__bf16)(0.02f * i); }
    const char* my = src + (size_t)(blockIdx.x % 64) * 65536 + wave * 6144 + lane * 16;
    u32x4 r[6];
    for (int i = 0; i < 6; ++i) r[i] = u32x4{0, 0, 0, 0};
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const char* p = my + (it & 7) * 24576 % 40000;
        char* dst = smem + (it & 1) * 24576 + wave * 6144;
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        } else if (MODE == 3) {
            // the same 6 DMA pieces, one after every 4th MFMA
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c1, 0, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + k * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + k * 1024), 16, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c3, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            continue;
        } else if (MODE == 4) {
            // the DMA alone: how many bytes per clock does a CU's LDS-DMA path move (L2-resident source)?
#pragma unroll
            for (int i = 0; i < 6; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            continue;
        } else if (MODE == 2) {
            // write last iteration's registers, then refill them
#pragma unroll
            for (int i = 0; i < 6; ++i) *reinterpret_cast<u32x4*>(dst + i * 1024 + lane * 16) = r[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) r[i] = *reinterpret_cast<const u32x4*>(p + i * 1024);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c3, 0, 0, 0);
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    float s = c0[0] + c1[0] + c2[0] + c3[0] + __uint_as_float(r[0].x) + smem[lane];
    if (s == 1234.5f) sink[0] = s;
}

template <int MODE>
void run(const char* name, const char* src, uint64_t* d, float* sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
    for (int grid : {256, 512}) {
        const int iters = 2000;
        hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 49152, 0, src, d, sink, 100);
        hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 49152, 0, src, d, sink, iters);
        hipDeviceSynchronize();
        uint64_t h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("%-44s %d workgroup(s)/CU: %7.1f cycles per iteration (24 MFMAs = 768)\n", name, grid / 256, (double)h / iters);
    }
}

int main() {
    char* src; uint64_t* d; float* sink;
    hipMalloc(&src, 64 * 65536 + 65536); hipMemset(src, 1, 64 * 65536 + 65536);
    hipMalloc(&d, 64); hipMalloc(&sink, 4);
    run<0>("24 MFMAs + barrier", src, d, sink);
    run<1>("+ 6 x LDS-DMA (global_load_lds_dwordx4)", src, d, sink);
    run<2>("+ 6 x global_load_dwordx4 + 6 x ds_write_b128", src, d, sink);
    run<3>("+ 6 x LDS-DMA interleaved with the MFMAs", src, d, sink);
    run<4>("6 x LDS-DMA per wave alone (24 KiB / workgroup)", src, d, sink);
    return 0;
}
