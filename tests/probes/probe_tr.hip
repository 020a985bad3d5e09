// Probe (test infrastructure): lane semantics of ds_read_b64_tr_b16 and of the LDS-DMA destination.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;

extern "C" __global__ void probe_tr_kernel(short* out /*[64][4]*/, int row_stride_elems) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // 16-lane group g reads a [4 rows][16 cols] block: lane p=(l&15) supplies row p>>2, cols 4*(p&3)..+3
    const int g = l >> 4, p = l & 15;
    const int addr = (g * 4 + (p >> 2)) * row_stride_elems + (p & 3) * 4;   // element index
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = t[j];
}

extern "C" __global__ void probe_dma_kernel(const uint32_t* src /*[64*4]*/, uint32_t* out /*[64*4+64*4]*/) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    // every lane sources its own 16 B (reversed lane order) ; destination = uniform base + 256 B
    const uint32_t* s = src + (63 - threadIdx.x) * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                     (__attribute__((address_space(3))) void*)(lds + 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}

extern "C" int probe_run(short* tr_out, int row_stride_elems, const uint32_t* dma_src, uint32_t* dma_out) {
    hipLaunchKernelGGL(probe_tr_kernel, dim3(1), dim3(64), 0, 0, tr_out, row_stride_elems);
    hipLaunchKernelGGL(probe_dma_kernel, dim3(1), dim3(64), 0, 0, dma_src, dma_out);
    return (int)hipDeviceSynchronize();
}
