// probe_coissue.hip -- two waves of one SIMD with different instruction streams (gfx950): how far do the matrix
// pipe and the VALU / transcendental pipe overlap ACROSS waves?  512-thread workgroup: waves w and w+4 share a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define R8(x) x x x x x x x x
#define M(acc) "v_mfma_f32_32x32x16_bf16 %" #acc ", %12, %13, %" #acc "\n"
#define F(op, r) op " %" #r ", %" #r "\n"
#define F2(op, r) op " %" #r ", %" #r ", %" #r "\n"
#define OPS : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(A), "v"(B)
enum { S_NONE, S_MFMA, S_EXP, S_MUL, S_MIX };   // MIX = mfma + 2 exp + 2 mul per gap (a softmax-like body)

template <int SA, int SB>
__global__ __launch_bounds__(512) void probe(uint64_t* out, float* sink, int iters) {
    const int wave = threadIdx.x >> 6;
    float a0 = threadIdx.x * 0.001f + 1.f, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    bf16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(0.01f * i); B[i] = (__bf16)(0.02f * i); }
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    auto body = [&](auto SC) {
        constexpr int S = decltype(SC)::value;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            if (S == S_MFMA) { R8(asm volatile(M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) OPS);) }
            if (S == S_EXP)  { R8(asm volatile(F("v_exp_f32", 4) F("v_exp_f32", 5) F("v_exp_f32", 6) F("v_exp_f32", 7) F("v_exp_f32", 8) F("v_exp_f32", 9) F("v_exp_f32", 10) F("v_exp_f32", 11) OPS);) }
            if (S == S_MUL)  { R8(asm volatile(F2("v_mul_f32", 4) F2("v_mul_f32", 5) F2("v_mul_f32", 6) F2("v_mul_f32", 7) F2("v_mul_f32", 8) F2("v_mul_f32", 9) F2("v_mul_f32", 10) F2("v_mul_f32", 11) OPS);) }
            if (S == S_MIX)  { R8(asm volatile(M(0) F("v_exp_f32", 4) F("v_exp_f32", 5) F2("v_mul_f32", 6) F2("v_mul_f32", 7) M(1) F("v_exp_f32", 8) F("v_exp_f32", 9) F2("v_mul_f32", 10) F2("v_mul_f32", 11)
                                               M(2) F("v_exp_f32", 4) F("v_exp_f32", 5) F2("v_mul_f32", 6) F2("v_mul_f32", 7) M(3) F("v_exp_f32", 8) F("v_exp_f32", 9) F2("v_mul_f32", 10) F2("v_mul_f32", 11) OPS);) }
        }
    };
    if (wave < 4) body(std::integral_constant<int, SA>{}); else body(std::integral_constant<int, SB>{});
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0[0] + c1[0] + c2[0] + c3[0];
    if (s == 1234.5f) sink[0] = s;
}

template <int SA, int SB>
void run(const char* name, uint64_t* d, float* sink) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<SA, SB>), dim3(256), dim3(512), 0, 0, d, sink, 100);
    hipLaunchKernelGGL((probe<SA, SB>), dim3(256), dim3(512), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
    uint64_t h[8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-34s cycles per 64-instr block: waves 0-3: %7.1f   waves 4-7: %7.1f\n", name, (double)h[0] / iters, (double)h[4] / iters);
}

int main() {
    uint64_t* d; float* sink;
    hipMalloc(&d, 64); hipMalloc(&sink, 4);
    run<S_MFMA, S_NONE>("mfma | idle", d, sink);
    run<S_EXP, S_NONE>("exp | idle", d, sink);
    run<S_MUL, S_NONE>("mul | idle", d, sink);
    run<S_MIX, S_NONE>("mix(32 mfma+64exp+64mul) | idle", d, sink);
    run<S_MFMA, S_EXP>("mfma | exp", d, sink);
    run<S_MFMA, S_MUL>("mfma | mul", d, sink);
    run<S_MFMA, S_MFMA>("mfma | mfma", d, sink);
    run<S_EXP, S_EXP>("exp | exp", d, sink);
    run<S_MUL, S_MUL>("mul | mul", d, sink);
    run<S_EXP, S_MUL>("exp | mul", d, sink);
    run<S_MIX, S_MIX>("mix | mix", d, sink);
    run<S_MIX, S_MFMA>("mix | mfma", d, sink);
    run<S_MIX, S_EXP>("mix | exp", d, sink);
    return 0;
}
