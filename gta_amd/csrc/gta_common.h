// gta_common.h -- device helpers shared by the gfx950 GTA kernels.
//
// Data model used by every kernel in this directory
//   * a head's dh channels are cut into 8-channel CHUNKS (16 B of bf16 = one ds_read_b128 /
//     one MFMA operand fragment).  A chunk is two 4-channel HALVES.  The reference's
//     block-diagonal reps (gta.py:160-219) never straddle a chunk when the slab offsets are
//     8-aligned (true for every shipped config), so rho acts on a chunk as:
//        half kind ID   : identity                      (triv slab, gta.py:127-132)
//        half kind SE3  : one 4x4 per-VIEW matrix       (gta.py:160-168)
//        half kind SO2  : two 2x2 per-TOKEN rotations   (gta.py:212-219)
//        chunk kind SO3 : [D^1 (3x3) | D^2 (5x5)] per view (gta.py:174-201, degree-2 layout)
//   * a chunk descriptor (uint32) says which; it is wave-uniform wherever it is used, so the
//     type switch is a scalar branch, never lane divergence.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define GTA_DEV __device__ __forceinline__

#include <type_traits>
#include <utility>
// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <class F, int... Is>
GTA_DEV void gta_static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
GTA_DEV void gta_static_for(F&& f) { gta_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// ---- chunk descriptor -----------------------------------------------------------------------
#define GTA_HALF_ID  0u
#define GTA_HALF_SE3 1u
#define GTA_HALF_SO2 2u
#define GTA_CHUNK_SO3 (1u << 4)
#define GTA_CHUNK_ZERO (1u << 5)   // padding chunk beyond dh (fused kernels pad dh up to 32k)
GTA_DEV uint32_t cd_lo(uint32_t d) { return d & 3u; }
GTA_DEV uint32_t cd_hi(uint32_t d) { return (d >> 2) & 3u; }
GTA_DEV uint32_t cd_so2_lo(uint32_t d) { return (d >> 8) & 0xffu; }   // first so2 block of the lo half
GTA_DEV uint32_t cd_so2_hi(uint32_t d) { return (d >> 16) & 0xffu; }  // first so2 block of the hi half

// ---- compile-time layouts of the shipped configs (the generic path reads the table at run time) ----
#define GTA_LAYOUT_GENERIC 0
#define GTA_LAYOUT_MS 1      // dh 96: se3 48 | so3 24 (L=2) | so2 24   runs/msn/GTA/gta_so3/config.yaml
#define GTA_LAYOUT_CL 2      // dh 64: se3 32 | so2 32                  runs/clevrtr/GTA/gta/config.yaml
#define GTA_LAYOUT_SO2 3     // pure so2 (any dh): 2-D GTA / *_no3demb configs
#define GTA_LAYOUT_MSG 4     // dh 96: se3 48 | so2 48                  runs/msn/GTA/{gta,gta_novtrnsfm,gta_sharedfreqs}, the decoders of gta_no2demb / gta_no3demb
#define GTA_LAYOUT_SE3 5     // se3 only (dh 96 instance)               runs/msn/GTA/gta_no2demb (encoder)
__host__ __device__ constexpr uint32_t gta_so2_desc(int first_block) {
    return GTA_HALF_SO2 | (GTA_HALF_SO2 << 2) | ((uint32_t)first_block << 8) | ((uint32_t)(first_block + 2) << 16);
}
__host__ __device__ constexpr uint32_t gta_layout_desc(int layout, int c) {
    return layout == GTA_LAYOUT_MS ? (c < 6 ? (GTA_HALF_SE3 | (GTA_HALF_SE3 << 2)) : c < 9 ? GTA_CHUNK_SO3 : gta_so2_desc(4 * (c - 9)))
         : layout == GTA_LAYOUT_CL ? (c < 4 ? (GTA_HALF_SE3 | (GTA_HALF_SE3 << 2)) : gta_so2_desc(4 * (c - 4)))
         : layout == GTA_LAYOUT_MSG ? (c < 6 ? (GTA_HALF_SE3 | (GTA_HALF_SE3 << 2)) : gta_so2_desc(4 * (c - 6)))
         : layout == GTA_LAYOUT_SE3 ? (GTA_HALF_SE3 | (GTA_HALF_SE3 << 2))
         : gta_so2_desc(4 * c);
}

// ---- bf16 <-> f32 -------------------------------------------------------------------------------
GTA_DEV uint32_t pack_bf16x2(float lo, float hi) {
    f32x2_t v = {lo, hi};
    bf16x2_t b = __builtin_convertvector(v, bf16x2_t);   // v_cvt_pk_bf16_f32 (RNE) on gfx950
    return __builtin_bit_cast(uint32_t, b);
}
GTA_DEV float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
GTA_DEV float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
GTA_DEV void unpack8(const u32x4_t w, float* x) {
    x[0] = bf16_lo(w.x); x[1] = bf16_hi(w.x); x[2] = bf16_lo(w.y); x[3] = bf16_hi(w.y);
    x[4] = bf16_lo(w.z); x[5] = bf16_hi(w.z); x[6] = bf16_lo(w.w); x[7] = bf16_hi(w.w);
}
GTA_DEV u32x4_t pack8(const float* x) {
    u32x4_t w;
    w.x = pack_bf16x2(x[0], x[1]); w.y = pack_bf16x2(x[2], x[3]);
    w.z = pack_bf16x2(x[4], x[5]); w.w = pack_bf16x2(x[6], x[7]);
    return w;
}

// ---- LDS tile swizzle -----------------------------------------------------------------------
// A tile is rows x UNITS 16-byte units.  Unit `u` of row `r` lives at position swz<UNITS>(r,u):
// a per-row ROTATION (period 16 rows) chosen so that 16 rows distinct mod 16 reading the same
// logical unit with ds_read_b128 touch 16 distinct 16-B slots of the 256-B bank row
// (conflict-free both for the lane==row staging reads and for the MFMA fragment reads).
// r06: for 8- and 16-unit rows (dh = 64 / 128 images) the rotation's bits are permuted so that
// the TRANSPOSE reads (ds_read_b64_tr_b16: a 32-lane group covers 4 consecutive rows x 4
// consecutive units x 2 halves) are conflict-free as well -- with rot = (r >> 1) & 7 rows r and
// r + 2 of an 8-unit image met in the same banks (PMC: SQ_LDS_BANK_CONFLICT = 33 % of the LDS
// cycles of the CLEVR-TR attention kernel, two extra cycles per transpose read).
// tests/test_host_logic.py brute forces both properties against the lane groups of
// MI355X_MICROARCH.md.
template <int UNITS>
GTA_DEV int swz_rot(int r) {
    if constexpr (UNITS == 8) return 4 * ((r >> 1) & 1) + ((r >> 2) & 3);
    else if constexpr (UNITS == 16) return 4 * (r & 3) + ((r >> 2) & 3);
    else {
        constexpr int tz = (UNITS % 16 == 0) ? 4 : (UNITS % 8 == 0) ? 3 : (UNITS % 4 == 0) ? 2
                           : (UNITS % 2 == 0) ? 1 : 0;
        return (r >> (4 - tz)) & ((1 << tz) - 1);
    }
}
template <int UNITS>
GTA_DEV int swz(int r, int u) {
    const int rot = swz_rot<UNITS>(r);
    int p = u + rot;
    return p >= UNITS ? p - UNITS : p;
}

// ---- small block transforms (all in fp32 registers) ------------------------------------------
GTA_DEV void mat4_apply(const float* M, float* x) {           // x[0..3] <- M(4x4 row-major) x
    const float a = x[0], b = x[1], c = x[2], d = x[3];
    x[0] = M[0] * a + M[1] * b + M[2] * c + M[3] * d;
    x[1] = M[4] * a + M[5] * b + M[6] * c + M[7] * d;
    x[2] = M[8] * a + M[9] * b + M[10] * c + M[11] * d;
    x[3] = M[12] * a + M[13] * b + M[14] * c + M[15] * d;
}
GTA_DEV void mat3_apply_p4(const float* M, float* x) {        // rows padded to 4 floats
    const float a = x[0], b = x[1], c = x[2];
    x[0] = M[0] * a + M[1] * b + M[2] * c;
    x[1] = M[4] * a + M[5] * b + M[6] * c;
    x[2] = M[8] * a + M[9] * b + M[10] * c;
}
GTA_DEV void mat5_apply_p8(const float* M, float* x) {        // rows padded to 8 floats
    const float a = x[0], b = x[1], c = x[2], d = x[3], e = x[4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
        x[i] = M[8 * i] * a + M[8 * i + 1] * b + M[8 * i + 2] * c + M[8 * i + 3] * d + M[8 * i + 4] * e;
}
// (c, s) rotation [[c,-s],[s,c]] (gta.py:64-67); INV = transpose
template <bool INV>
GTA_DEV void rot2_apply(float c, float s, float* x) {
    const float a = x[0], b = x[1];
    if (INV) { x[0] = c * a + s * b; x[1] = c * b - s * a; }
    else     { x[0] = c * a - s * b; x[1] = s * a + c * b; }
}

// LDS -> registers for the small per-view matrices (all lanes of a view read the same address:
// broadcast, no bank conflict)
GTA_DEV void lds_load16(const float* p, float* M) {
    const f32x4_t* v = reinterpret_cast<const f32x4_t*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f32x4_t t = v[i]; M[4*i] = t.x; M[4*i+1] = t.y; M[4*i+2] = t.z; M[4*i+3] = t.w; }
}
template <int N4>
GTA_DEV void lds_loadN4(const float* p, float* M) {
    const f32x4_t* v = reinterpret_cast<const f32x4_t*>(p);
#pragma unroll
    for (int i = 0; i < N4; ++i) { f32x4_t t = v[i]; M[4*i] = t.x; M[4*i+1] = t.y; M[4*i+2] = t.z; M[4*i+3] = t.w; }
}

// Per-view records staged in LDS by the attention kernels (floats):
//   q side: [  0: 16) Aq  = (E (.) m)^T      applied to Q           (gta.py:165)
//           [ 16: 32) Oq  =  E (.) m         applied to the output  (gta.py:255-257)
//           [ 32: 44) D1q (rows padded to 4) [ 44: 84) D2q (rows padded to 8)   (gta.py:193-194)
//           [ 84: 96) D1q^T                  [ 96:136) D2q^T                    (gta.py:259-267)
//   k side: [  0: 16) Bk  = inv(E) (.) m     applied to K and V     (gta.py:166-168)
//           [ 16: 28) D1k                    [ 28: 68) D2k
// with m the trans_coeff mask of gta.py:40-44.
#define GTA_QREC 136
#define GTA_KREC 68
#define GTA_QREC_A 0
#define GTA_QREC_O 16
#define GTA_QREC_D1 32
#define GTA_QREC_D2 44
#define GTA_QREC_D1T 84
#define GTA_QREC_D2T 96
#define GTA_KREC_B 0
#define GTA_KREC_D1 16
#define GTA_KREC_D2 28

// Apply one chunk's rho to up to two 8-vectors that share it (K and V of one token).
//   se3  : LDS pointer to the 4x4 to use (already masked / transposed as the caller needs)
//   d1,d2: LDS pointers to padded D^1 / D^2 (or their transposes)
//   cs   : this token's (cos,sin) pairs for the chunk: cs[0..1] lo half, cs[2..3] hi half
template <bool SO2_INV, int NV>
GTA_DEV void chunk_apply(uint32_t desc, const float* se3, const float* d1, const float* d2,
                         const f32x2_t* cs, float (*x)[8]) {
    if (desc & GTA_CHUNK_SO3) {
        float M1[12], M2[40];
        lds_loadN4<3>(d1, M1);
        lds_loadN4<10>(d2, M2);
#pragma unroll
        for (int v = 0; v < NV; ++v) { mat3_apply_p4(M1, x[v]); mat5_apply_p8(M2, x[v] + 3); }
        return;
    }
    const uint32_t lo = cd_lo(desc), hi = cd_hi(desc);
    if (lo == GTA_HALF_SE3 || hi == GTA_HALF_SE3) {
        float M[16];
        lds_load16(se3, M);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (lo == GTA_HALF_SE3) mat4_apply(M, x[v]);
            if (hi == GTA_HALF_SE3) mat4_apply(M, x[v] + 4);
        }
    }
    if (lo == GTA_HALF_SO2) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            rot2_apply<SO2_INV>(cs[0].x, cs[0].y, x[v]);
            rot2_apply<SO2_INV>(cs[1].x, cs[1].y, x[v] + 2);
        }
    }
    if (hi == GTA_HALF_SO2) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            rot2_apply<SO2_INV>(cs[2].x, cs[2].y, x[v] + 4);
            rot2_apply<SO2_INV>(cs[3].x, cs[3].y, x[v] + 6);
        }
    }
}

// Load the (cos,sin) pairs a chunk needs for token row `t` of a [T, nso2] float2 table.
GTA_DEV void load_cs(uint32_t desc, const float* __restrict__ cs_row, f32x2_t* cs) {
    if (desc & GTA_CHUNK_SO3) return;
    if (cd_lo(desc) == GTA_HALF_SO2) {
        const f32x4_t t = *reinterpret_cast<const f32x4_t*>(cs_row + 2 * cd_so2_lo(desc));
        cs[0] = f32x2_t{t.x, t.y}; cs[1] = f32x2_t{t.z, t.w};
    }
    if (cd_hi(desc) == GTA_HALF_SO2) {
        const f32x4_t t = *reinterpret_cast<const f32x4_t*>(cs_row + 2 * cd_so2_hi(desc));
        cs[2] = f32x2_t{t.x, t.y}; cs[3] = f32x2_t{t.z, t.w};
    }
}

// view index of token t when every view has P tokens: exact for t < 2^22 (float estimate + fix)
GTA_DEV int view_of(int t, int P, float invP) {
    int n = (int)((float)t * invP);
    if (n * P > t) --n;
    if ((n + 1) * P <= t) ++n;
    return n;
}

// LDS-DMA of consecutive 1-KiB pieces (global -> LDS, 16 B per lane).  Written as asm in the scalar-base form -- global address = SGPR pair + 32-bit lane offset + immediate,
// LDS address = M0 + the same immediate + 16 * lane -- so a group of four pieces needs ONE s_mov to M0 and no vector
// arithmetic at all; through the builtin hipcc forms a 64-bit per-lane address and a new M0 for every piece (17 VALU
// instructions per tile and wave in a loop whose issue slots are the scarce resource).  The compiler does not see these
// as memory operations: every consumer sits behind an explicit s_waitcnt vmcnt + barrier (as with the builtin).
template <int NP>
GTA_DEV void dma_group(uint32_t lds, const char* base, unsigned voff) {
    static_assert(NP >= 1 && NP <= 4, "13-bit immediates: four 1-KiB pieces per base");
    if constexpr (NP == 1)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(base) : "memory");
    else if constexpr (NP == 2)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024"
                     ::"s"(lds), "v"(voff), "s"(base) : "memory");
    else if constexpr (NP == 3)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048" ::"s"(lds), "v"(voff), "s"(base) : "memory");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072"
                     ::"s"(lds), "v"(voff), "s"(base) : "memory");
}

// `bytes` consecutive bytes (a multiple of 4 KiB) split over the 4 waves of a 256-thread workgroup
template <int BYTES>
GTA_DEV void dma_linear_4waves(char* dst, const char* src, int wave, int lane) {
    constexpr int PER_WAVE = BYTES / 1024 / 4;
    static_assert(BYTES % 4096 == 0, "piece split");
    const unsigned voff = (unsigned)lane * 16u;
    const char* base = src + wave * (PER_WAVE * 1024);
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(dst + wave * (PER_WAVE * 1024));
    gta_static_for<(PER_WAVE + 3) / 4>([&](auto GC) {
        constexpr int g = decltype(GC)::value, np = PER_WAVE - 4 * g < 4 ? PER_WAVE - 4 * g : 4;
        dma_group<np>(lds + g * 4096, base + g * 4096, voff);
    });
}

// hipFuncAttributeMaxDynamicSharedMemorySize (the opt-in above 64 KiB of dynamic LDS) is a per-DEVICE attribute of a
// kernel: set it once per (kernel instantiation, device).  The flags are plain bools: two racing threads at worst
// repeat the same idempotent call.
template <auto Kernel>
inline int gta_lds_optin(int bytes) {
    constexpr int MAXDEV = 64;
    static bool done[MAXDEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -5;                 // GTA_E_NODEVICE
    if (dev >= 0 && dev < MAXDEV && done[dev]) return 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return -4;                                                   // GTA_E_LAUNCH
    if (dev >= 0 && dev < MAXDEV) done[dev] = true;
    return 0;
}
