// probe_issue.hip -- issue-rate microbenchmarks on gfx950 (not part of the library).
// One kernel per instruction mix (template), so the loop body is exactly the asm block.
//   ticks = s_memtime (shader cycles) of wave 0 of block 0; ns = hipEvent wall time per loop iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define R8(x) x x x x x x x x
#define M(acc) "v_mfma_f32_32x32x16_bf16 %" #acc ", %12, %13, %" #acc "\n"
#define F(op, r) op " %" #r ", %" #r "\n"
#define F2(op, r) op " %" #r ", %" #r ", %" #r "\n"

enum { EXP, ADD, PKMUL, MAX3, CVT, MFMA, MFMA_ADD3, MFMA_ADD5, MFMA_ADD7, MFMA_EXP3, MFMA_EXP5, MFMA_MIX5, QK_ADD5, PV_ADD5, QK_ONLY, PV_ONLY, PV_EXP2ADD2, PV_TR2_ADD3 };

template <int MODE>
__global__ __launch_bounds__(256) void probe(uint64_t* out, float* sink, int iters) {
    asm volatile("" ::: "a0", "a127");
    __shared__ float lds_pad[4096];
    lds_pad[threadIdx.x] = 0.f;
    float a0 = threadIdx.x * 0.001f + 1.f, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    bf16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(0.01f * i); B[i] = (__bf16)(0.02f * i); }
    __syncthreads();
    if (MODE == PV_TR2_ADD3) a0 = __uint_as_float((threadIdx.x & 63) * 16);   // an LDS address
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#define OPS : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(A), "v"(B)
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE == EXP)   { R8(asm volatile(F("v_exp_f32", 4) F("v_exp_f32", 5) F("v_exp_f32", 6) F("v_exp_f32", 7) F("v_exp_f32", 8) F("v_exp_f32", 9) F("v_exp_f32", 10) F("v_exp_f32", 11) OPS);) }
        if (MODE == ADD)   { R8(asm volatile(F2("v_mul_f32", 4) F2("v_mul_f32", 5) F2("v_mul_f32", 6) F2("v_mul_f32", 7) F2("v_mul_f32", 8) F2("v_mul_f32", 9) F2("v_mul_f32", 10) F2("v_mul_f32", 11) OPS);) }
        if (MODE == PKMUL) { R8(asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n" 
                                            "v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n"
                                            : "+v"(*(double*)&c0), "+v"(*(double*)&c1), "+v"(*(double*)&c2), "+v"(*(double*)&c3));) }
        if (MODE == MAX3)  { R8(asm volatile("v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %8\n v_max3_f32 %7, %7, %8, %9\n"
                                            "v_max3_f32 %8, %8, %9, %10\n v_max3_f32 %9, %9, %10, %11\n v_max3_f32 %10, %10, %11, %4\n v_max3_f32 %11, %11, %4, %5\n" OPS);) }
        if (MODE == CVT)   { R8(asm volatile("v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %8\n"
                                            "v_cvt_pk_bf16_f32 %8, %8, %9\n v_cvt_pk_bf16_f32 %9, %9, %10\n v_cvt_pk_bf16_f32 %10, %10, %11\n v_cvt_pk_bf16_f32 %11, %11, %4\n" OPS);) }
        if (MODE == MFMA)  { R8(asm volatile(M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) OPS);) }
        if (MODE == MFMA_ADD3) { R8(asm volatile(M(0) F2("v_mul_f32", 4) F2("v_mul_f32", 5) F2("v_mul_f32", 6) M(1) F2("v_mul_f32", 7) F2("v_mul_f32", 8) F2("v_mul_f32", 9)
                                                 M(2) F2("v_mul_f32", 10) F2("v_mul_f32", 11) F2("v_mul_f32", 4) M(3) F2("v_mul_f32", 5) F2("v_mul_f32", 6) F2("v_mul_f32", 7) OPS);) }
        if (MODE == MFMA_ADD5) { R8(asm volatile(M(0) F2("v_mul_f32", 4) F2("v_mul_f32", 5) F2("v_mul_f32", 6) F2("v_mul_f32", 7) F2("v_mul_f32", 8)
                                                 M(1) F2("v_mul_f32", 9) F2("v_mul_f32", 10) F2("v_mul_f32", 11) F2("v_mul_f32", 4) F2("v_mul_f32", 5)
                                                 M(2) F2("v_mul_f32", 6) F2("v_mul_f32", 7) F2("v_mul_f32", 8) F2("v_mul_f32", 9) F2("v_mul_f32", 10)
                                                 M(3) F2("v_mul_f32", 11) F2("v_mul_f32", 4) F2("v_mul_f32", 5) F2("v_mul_f32", 6) F2("v_mul_f32", 7) OPS);) }
        if (MODE == MFMA_ADD7) { R8(asm volatile(M(0) F2("v_mul_f32", 4) F2("v_mul_f32", 5) F2("v_mul_f32", 6) F2("v_mul_f32", 7) F2("v_mul_f32", 8) F2("v_mul_f32", 9) F2("v_mul_f32", 10)
                                                 M(1) F2("v_mul_f32", 11) F2("v_mul_f32", 4) F2("v_mul_f32", 5) F2("v_mul_f32", 6) F2("v_mul_f32", 7) F2("v_mul_f32", 8) F2("v_mul_f32", 9)
                                                 M(2) F2("v_mul_f32", 10) F2("v_mul_f32", 11) F2("v_mul_f32", 4) F2("v_mul_f32", 5) F2("v_mul_f32", 6) F2("v_mul_f32", 7) F2("v_mul_f32", 8)
                                                 M(3) F2("v_mul_f32", 9) F2("v_mul_f32", 10) F2("v_mul_f32", 11) F2("v_mul_f32", 4) F2("v_mul_f32", 5) F2("v_mul_f32", 6) F2("v_mul_f32", 7) OPS);) }
        if (MODE == MFMA_EXP3) { R8(asm volatile(M(0) F("v_exp_f32", 4) F("v_exp_f32", 5) F("v_exp_f32", 6) M(1) F("v_exp_f32", 7) F("v_exp_f32", 8) F("v_exp_f32", 9)
                                                 M(2) F("v_exp_f32", 10) F("v_exp_f32", 11) F("v_exp_f32", 4) M(3) F("v_exp_f32", 5) F("v_exp_f32", 6) F("v_exp_f32", 7) OPS);) }
        if (MODE == MFMA_EXP5) { R8(asm volatile(M(0) F("v_exp_f32", 4) F("v_exp_f32", 5) F("v_exp_f32", 6) F("v_exp_f32", 7) F("v_exp_f32", 8)
                                                 M(1) F("v_exp_f32", 9) F("v_exp_f32", 10) F("v_exp_f32", 11) F("v_exp_f32", 4) F("v_exp_f32", 5)
                                                 M(2) F("v_exp_f32", 6) F("v_exp_f32", 7) F("v_exp_f32", 8) F("v_exp_f32", 9) F("v_exp_f32", 10)
                                                 M(3) F("v_exp_f32", 11) F("v_exp_f32", 4) F("v_exp_f32", 5) F("v_exp_f32", 6) F("v_exp_f32", 7) OPS);) }
#define MQ(acc) "v_mfma_f32_32x32x16_bf16 %" #acc ", a[100:103], a[104:107], %" #acc "\n"
#define MP(o) "v_mfma_f32_32x32x16_bf16 a[" #o "], a[100:103], %12, a[" #o "]\n"
#define MUL5(a,b,c,d,e) F2("v_mul_f32", a) F2("v_mul_f32", b) F2("v_mul_f32", c) F2("v_mul_f32", d) F2("v_mul_f32", e)
        if (MODE == QK_ONLY) { R8(asm volatile(MQ(0) MQ(1) MQ(2) MQ(3) MQ(0) MQ(1) MQ(2) MQ(3) OPS);) }
        if (MODE == PV_ONLY) { R8(asm volatile(MP(0:15) MP(16:31) MP(32:47) MP(48:63) MP(0:15) MP(16:31) MP(32:47) MP(48:63) OPS);) }
        if (MODE == QK_ADD5) { R8(asm volatile(MQ(0) MUL5(4,5,6,7,8) MQ(1) MUL5(9,10,11,4,5) MQ(2) MUL5(6,7,8,9,10) MQ(3) MUL5(11,4,5,6,7) OPS);) }
        if (MODE == PV_ADD5) { R8(asm volatile(MP(0:15) MUL5(4,5,6,7,8) MP(16:31) MUL5(9,10,11,4,5) MP(32:47) MUL5(6,7,8,9,10) MP(48:63) MUL5(11,4,5,6,7) OPS);) }
        if (MODE == PV_EXP2ADD2) { R8(asm volatile(MP(0:15) F("v_exp_f32", 4) F("v_exp_f32", 5) F2("v_mul_f32", 6) F2("v_mul_f32", 7) MP(16:31) F("v_exp_f32", 8) F("v_exp_f32", 9) F2("v_mul_f32", 10) F2("v_mul_f32", 11)
                                                MP(32:47) F("v_exp_f32", 4) F("v_exp_f32", 5) F2("v_mul_f32", 6) F2("v_mul_f32", 7) MP(48:63) F("v_exp_f32", 8) F("v_exp_f32", 9) F2("v_mul_f32", 10) F2("v_mul_f32", 11) OPS);) }
        if (MODE == PV_TR2_ADD3) { R8(asm volatile(MP(0:15) "ds_read_b64_tr_b16 a[110:111], %4\n ds_read_b64_tr_b16 a[112:113], %4 offset:512\n" F2("v_mul_f32", 6) F2("v_mul_f32", 7) F2("v_mul_f32", 8)
                                                MP(16:31) "ds_read_b128 a[114:117], %4\n ds_read_b64_tr_b16 a[118:119], %4 offset:512\n" F2("v_mul_f32", 9) F2("v_mul_f32", 10) F2("v_mul_f32", 11)
                                                MP(32:47) "ds_read_b64_tr_b16 a[110:111], %4\n ds_read_b64_tr_b16 a[112:113], %4 offset:512\n" F2("v_mul_f32", 6) F2("v_mul_f32", 7) F2("v_mul_f32", 8)
                                                MP(48:63) "ds_read_b128 a[114:117], %4\n ds_read_b64_tr_b16 a[118:119], %4 offset:512\n s_waitcnt lgkmcnt(4)\n" F2("v_mul_f32", 9) F2("v_mul_f32", 10) F2("v_mul_f32", 11) OPS);) }
        if (MODE == MFMA_MIX5) { R8(asm volatile(M(0) F("v_exp_f32", 4) F2("v_mul_f32", 5) F("v_exp_f32", 6) F2("v_mul_f32", 7) F("v_exp_f32", 8)
                                                 M(1) F2("v_mul_f32", 9) F("v_exp_f32", 10) F2("v_mul_f32", 11) F("v_exp_f32", 4) F2("v_mul_f32", 5)
                                                 M(2) F("v_exp_f32", 6) F2("v_mul_f32", 7) F("v_exp_f32", 8) F2("v_mul_f32", 9) F("v_exp_f32", 10)
                                                 M(3) F2("v_mul_f32", 11) F("v_exp_f32", 4) F2("v_mul_f32", 5) F("v_exp_f32", 6) F2("v_mul_f32", 7) OPS);) }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0[0] + c1[0] + c2[0] + c3[0];
    if (s == 1234.5f) sink[0] = s;
}

template <int MODE>
void run(const char* name, int n_instr, int n_mfma, uint64_t* d, float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1, 256, 512, 1024}) {     // 256 threads: 1, 1, 2, 4 waves per SIMD
        const int iters = 4000;
        hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, d, sink, 200);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, d, sink, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        uint64_t h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        const double ns_it = ms * 1e6 / iters;
        const int wps = grid <= 256 ? 1 : grid / 256;
        printf("%-22s grid=%4d (%d wave/SIMD) %8.1f ns/iter  wave0: %7.1f cyc/iter = %5.1f cyc/instr", name, grid, wps, ns_it,
               (double)h / iters, (double)h / iters / n_instr);
        if (n_mfma) printf("  %5.1f cyc/mfma  chip %.0f TF", (double)h / iters / n_mfma, (double)(grid < 256 ? grid : 256) * 4 * wps * n_mfma * 32768.0 / ns_it / 1e3);
        printf("  | SIMD ns/instr %.2f\n", ns_it / (n_instr * wps));
    }
}

int main() {
    uint64_t* d; float* sink;
    hipMalloc(&d, 64); hipMalloc(&sink, 4);
    run<QK_ONLY>("mfma vD,aA,aB x64", 64, 64, d, sink);
    run<PV_ONLY>("mfma aD,aA,vB x64", 64, 64, d, sink);
    run<QK_ADD5>("32x(mfma vD,aA,aB +5mul)", 192, 32, d, sink);
    run<PV_ADD5>("32x(mfma aD,aA,vB +5mul)", 192, 32, d, sink);
    run<PV_EXP2ADD2>("32x(mfma aD + 2exp 2mul)", 160, 32, d, sink);
    run<PV_TR2_ADD3>("32x(mfma aD + 2ds + 3mul)", 192, 32, d, sink);
    return 0;
    run<ADD>("v_mul_f32 x64", 64, 0, d, sink);
    run<EXP>("v_exp_f32 x64", 64, 0, d, sink);
    run<PKMUL>("v_pk_mul_f32 x64", 64, 0, d, sink);
    run<MAX3>("v_max3_f32 x64", 64, 0, d, sink);
    run<CVT>("v_cvt_pk_bf16_f32 x64", 64, 0, d, sink);
    run<MFMA>("mfma x64", 64, 64, d, sink);
    run<MFMA_ADD3>("32 x (mfma + 3 mul)", 128, 32, d, sink);
    run<MFMA_ADD5>("32 x (mfma + 5 mul)", 192, 32, d, sink);
    run<MFMA_ADD7>("32 x (mfma + 7 mul)", 256, 32, d, sink);
    run<MFMA_EXP3>("32 x (mfma + 3 exp)", 128, 32, d, sink);
    run<MFMA_EXP5>("32 x (mfma + 5 exp)", 192, 32, d, sink);
    run<MFMA_MIX5>("32 x (mfma + 3exp2mul)", 192, 32, d, sink);
    return 0;
}
