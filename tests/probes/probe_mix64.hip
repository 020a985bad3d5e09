// probe_mix64.hip -- r06: what the dh = 64 tile loop's instruction mix can reach on one SIMD (not part of the library).
// A dh = 64 tile of a 32-row wave is 16 v_mfma_f32_32x32x16_bf16 beside ~32 v_exp_f32, ~16 v_cvt_pk_bf16_f32 and 16-48 plain VALU.
// Question: with 1 / 2 / 3 waves per SIMD, how close to the matrix-pipe rate (32 cycles per MFMA and SIMD) does the SIMD run when
//   SEP  each wave's stream is phase-separated as hipcc emits it (8 MFMA | 82 VALU | 8 MFMA | 10 VALU), against
//   MIXn each MFMA is followed by n fillers of the same multiset (hand-interleaved stream)?
// Output per (mode, waves per SIMD): cycles per tile of wave 0, and the SIMD's matrix-pipe utilisation = waves * 16 * 32 / cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define M(acc) "v_mfma_f32_32x32x16_bf16 %" #acc ", %12, %13, %" #acc "\n"
#define E(r) "v_exp_f32 %" #r ", %" #r "\n"
#define A(r) "v_add_f32 %" #r ", %" #r ", %" #r "\n"
#define C(r) "v_cvt_pk_bf16_f32 %" #r ", %" #r ", %" #r "\n"
#define OPS : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(Af), "v"(Bf)

enum { SEP, MIX6, MIX5, MIX4, MIX3, MFMA_ONLY, SEP_LEAN, MIX6_DOT };

// the filler multisets (per 16 MFMAs): SEP / MIX6 = 32 E + 48 A + 16 C (96); MIX5 = 32 E + 32 A + 16 C (80); MIX4 = 32 E + 16 A + 16 C (64); MIX3 = 32 E + 16 C (48)
#define G6(m, e1, a1, e2, a2, c1, a3) M(m) E(e1) A(a1) E(e2) A(a2) C(c1) A(a3)
#define G5(m, e1, a1, e2, a2, c1) M(m) E(e1) A(a1) E(e2) A(a2) C(c1)
#define G4(m, e1, e2, a1, c1) M(m) E(e1) E(e2) A(a1) C(c1)
#define G3(m, e1, e2, c1) M(m) E(e1) E(e2) C(c1)

template <int MODE>
__global__ __launch_bounds__(256, 3) void probe(uint64_t* out, float* sink, int iters) {
    float a0 = threadIdx.x * 0.001f + 1.f, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    bf16x8 Af, Bf;
    for (int i = 0; i < 8; ++i) { Af[i] = (__bf16)(0.01f * i + 0.003f * (threadIdx.x & 31)); Bf[i] = (__bf16)(0.02f * i - 0.001f * (threadIdx.x & 63)); }
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE == MFMA_ONLY) {
            asm volatile(M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) OPS);
        }
        if (MODE == SEP) {       // hipcc's order: QK^T (two chains), softmax, PV (two chains), addresses
            asm volatile(M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1)
                         E(4) E(5) E(6) E(7) E(8) E(9) E(10) E(11) E(4) E(5) E(6) E(7) E(8) E(9) E(10) E(11)
                         A(4) A(5) A(6) A(7) A(8) A(9) A(10) A(11) A(4) A(5) A(6) A(7) A(8) A(9) A(10) A(11)
                         E(4) E(5) E(6) E(7) E(8) E(9) E(10) E(11) E(4) E(5) E(6) E(7) E(8) E(9) E(10) E(11)
                         A(4) A(5) A(6) A(7) A(8) A(9) A(10) A(11) A(4) A(5) A(6) A(7) A(8) A(9) A(10) A(11)
                         C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11)
                         A(4) A(5) A(6) A(7) A(8) A(9)
                         M(2) M(3) M(2) M(3) M(2) M(3) M(2) M(3)
                         A(4) A(5) A(6) A(7) A(8) A(9) A(10) A(11) A(4) A(5) OPS);
        }
        if (MODE == SEP_LEAN) {  // the same order with the MIX4 multiset (32 E + 16 A + 16 C)
            asm volatile(M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1)
                         E(4) E(5) E(6) E(7) E(8) E(9) E(10) E(11) E(4) E(5) E(6) E(7) E(8) E(9) E(10) E(11)
                         E(4) E(5) E(6) E(7) E(8) E(9) E(10) E(11) E(4) E(5) E(6) E(7) E(8) E(9) E(10) E(11)
                         A(4) A(5) A(6) A(7) A(8) A(9) A(10) A(11)
                         C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11)
                         M(2) M(3) M(2) M(3) M(2) M(3) M(2) M(3)
                         A(4) A(5) A(6) A(7) A(8) A(9) A(10) A(11) OPS);
        }
        if (MODE == MIX6) {
            asm volatile(G6(0, 4, 5, 6, 7, 8, 9) G6(1, 10, 11, 4, 5, 6, 7) G6(2, 8, 9, 10, 11, 4, 5) G6(3, 6, 7, 8, 9, 10, 11)
                         G6(0, 4, 5, 6, 7, 8, 9) G6(1, 10, 11, 4, 5, 6, 7) G6(2, 8, 9, 10, 11, 4, 5) G6(3, 6, 7, 8, 9, 10, 11)
                         G6(0, 4, 5, 6, 7, 8, 9) G6(1, 10, 11, 4, 5, 6, 7) G6(2, 8, 9, 10, 11, 4, 5) G6(3, 6, 7, 8, 9, 10, 11)
                         G6(0, 4, 5, 6, 7, 8, 9) G6(1, 10, 11, 4, 5, 6, 7) G6(2, 8, 9, 10, 11, 4, 5) G6(3, 6, 7, 8, 9, 10, 11) OPS);
        }
        if (MODE == MIX5) {
            asm volatile(G5(0, 4, 5, 6, 7, 8) G5(1, 9, 10, 11, 4, 5) G5(2, 6, 7, 8, 9, 10) G5(3, 11, 4, 5, 6, 7)
                         G5(0, 8, 9, 10, 11, 4) G5(1, 5, 6, 7, 8, 9) G5(2, 10, 11, 4, 5, 6) G5(3, 7, 8, 9, 10, 11)
                         G5(0, 4, 5, 6, 7, 8) G5(1, 9, 10, 11, 4, 5) G5(2, 6, 7, 8, 9, 10) G5(3, 11, 4, 5, 6, 7)
                         G5(0, 8, 9, 10, 11, 4) G5(1, 5, 6, 7, 8, 9) G5(2, 10, 11, 4, 5, 6) G5(3, 7, 8, 9, 10, 11) OPS);
        }
        if (MODE == MIX4) {
            asm volatile(G4(0, 4, 5, 6, 7) G4(1, 8, 9, 10, 11) G4(2, 4, 5, 6, 7) G4(3, 8, 9, 10, 11)
                         G4(0, 4, 5, 6, 7) G4(1, 8, 9, 10, 11) G4(2, 4, 5, 6, 7) G4(3, 8, 9, 10, 11)
                         G4(0, 4, 5, 6, 7) G4(1, 8, 9, 10, 11) G4(2, 4, 5, 6, 7) G4(3, 8, 9, 10, 11)
                         G4(0, 4, 5, 6, 7) G4(1, 8, 9, 10, 11) G4(2, 4, 5, 6, 7) G4(3, 8, 9, 10, 11) OPS);
        }
        if (MODE == MIX3) {
            asm volatile(G3(0, 4, 5, 6) G3(1, 7, 8, 9) G3(2, 10, 11, 4) G3(3, 5, 6, 7)
                         G3(0, 8, 9, 10) G3(1, 11, 4, 5) G3(2, 6, 7, 8) G3(3, 9, 10, 11)
                         G3(0, 4, 5, 6) G3(1, 7, 8, 9) G3(2, 10, 11, 4) G3(3, 5, 6, 7)
                         G3(0, 8, 9, 10) G3(1, 11, 4, 5) G3(2, 6, 7, 8) G3(3, 9, 10, 11) OPS);
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0[0] + c1[0] + c2[0] + c3[0];
    if (s == 1234.5f) sink[0] = s;
}

template <int MODE>
void run(const char* name, int n_fill, uint64_t* d, float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 3}) {
        const int grid = 256 * wps, iters = 3000;
        hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, d, sink, 300);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, d, sink, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        static uint64_t h[768];
        hipMemcpy(h, d, 8 * grid, hipMemcpyDeviceToHost);
        double mean = 0, mx = 0;
        for (int i = 0; i < grid; ++i) { mean += (double)h[i]; if ((double)h[i] > mx) mx = (double)h[i]; }
        mean /= grid;
        printf("%-10s fillers/tile %3d  waves/SIMD %d: %7.1f cyc/tile (mean over wgs; max %7.1f)  pipe util %.3f  | %.1f ns/tile -> clock %.2f GHz\n", name, n_fill, wps,
               mean / iters, mx / iters, wps * 16 * 32.0 / (mx / iters), ms * 1e6 / iters, (mx / iters) / (ms * 1e6 / iters));
    }
}

int main() {
    uint64_t* d; float* sink;
    hipMalloc(&d, 8 * 768); hipMalloc(&sink, 4);
    run<MFMA_ONLY>("mfma", 0, d, sink);
    run<SEP>("sep", 96, d, sink);
    run<MIX6>("mix6", 96, d, sink);
    run<MIX5>("mix5", 80, d, sink);
    run<SEP_LEAN>("sep_lean", 64, d, sink);
    run<MIX4>("mix4", 64, d, sink);
    run<MIX3>("mix3", 48, d, sink);
    return 0;
}
