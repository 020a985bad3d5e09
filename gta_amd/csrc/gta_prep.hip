// gta_prep.hip -- K/V rep pre-pass of the two-stage forward (gfx950).
//
// rho_k acts on K and V per key TOKEN, but a flash kernel re-reads every key tile once per query tile, so fusing
// rho_k there re-does the same fp32 VALU work Tq/BM times (10x at the MSN shape).  Here it is done exactly once:
// one 64-key tile of one (b,h) per workgroup; raw K,V rows -> LDS by LDS-DMA (coalesced), lane == key row applies
// rho_k per 8-channel chunk in fp32 registers (gta.py:160-219 for K and V), the bf16 TILE IMAGES -- the exact
// rotation-swizzled byte image the flash kernels want in LDS -- go out with 1-KiB wave stores, together with the
// tile's max |k'| (the flash kernels' lazy-softmax bound).
#include "gta_flash_common.h"

namespace {

template <int DHP, int ESZ>
struct PrepSmem {
    static constexpr int CHP = DHP / 8;
    static constexpr int RAW_UNITS = DHP * ESZ / 16;
    static constexpr int RAW_BYTES = BN * DHP * ESZ;
    static constexpr int OFF_RAWK = 0;
    static constexpr int OFF_RAWV = OFF_RAWK + RAW_BYTES;
    // bf16 input: a raw unit and its image unit have the same (row, position) -> transform in place;
    // fp32 input: the image (half the bytes) gets its own region
    static constexpr int IMG = BN * DHP * 2;
    static constexpr int OFF_IMGK = (ESZ == 2) ? OFF_RAWK : OFF_RAWV + RAW_BYTES;
    static constexpr int OFF_IMGV = (ESZ == 2) ? OFF_RAWV : OFF_IMGK + IMG;
    // the k-side view records (and later the 4 x 64 row-norm partials) sit behind the data, sized by the actual
    // number of views: 5 workgroups per CU with 16 views' worth reserved, 6 with the 5 views of the MSN config
    static constexpr int OFF_KREC = (ESZ == 2) ? OFF_RAWV + RAW_BYTES : OFF_IMGV + IMG;
    static int total(int Nk) { const int rec = Nk * GTA_KREC * 4; return OFF_KREC + (rec > 1024 ? rec : 1024); }
};

template <int DHP, int ESZ>
__global__ __launch_bounds__(256) void gta_kv_prep_kernel(const GtaFwdParams p) {
    using S = PrepSmem<DHP, ESZ>;
    constexpr int CHP = S::CHP, U = S::RAW_UNITS;
    constexpr int IMG = BN * DHP * 2;                       // bytes of one bf16 tile image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Work map (1-D grid): workgroup L runs on XCD L % 8 (MI355X_MICROARCH.md).  The K (and V) rows of the H heads of one
    // token are contiguous (the packed projection, layers.py:389), and a head's 2*dh bytes do not end on 128-B lines, so
    // neighbouring heads share lines: the H workgroups of one 64-token row tile are made consecutive on ONE XCD, whose L2
    // then serves the shared lines (r01 measured 1.40x over-fetch with heads spread over the XCDs).
    const int n_tiles = (p.Tk + BN - 1) / BN;
    int j, h, b;
    {
        const int L = blockIdx.x, x = L & 7, i = L >> 3;
        const int r = x + 8 * (i / p.H);                 // row tile = (b, j)
        h = i - (i / p.H) * p.H;
        if (r >= p.B * n_tiles) return;
        b = r / n_tiles;
        j = r - b * n_tiles;
    }

    const char* kg = (const char*)p.k + ((long)b * p.k_sb + (long)h * p.k_sh) * ESZ;
    const char* vg = (const char*)p.v + ((long)b * p.v_sb + (long)h * p.v_sh) * ESZ;
    const long k_rs = p.k_st * ESZ, v_rs = p.v_st * ESZ;
    const int ch_real = p.dh >> 3, real_units = p.dh * ESZ / 16;

    // raw rows -> LDS (coalesced LDS-DMA; the per-lane source address carries the swizzle)
    constexpr int NI = BN * U / 256;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int u0 = (wave * NI + i) * 64, u = u0 + lane;
        const int r = u / U, pos = u - r * U;
        constexpr int tz = (U % 16 == 0) ? 4 : (U % 8 == 0) ? 3 : (U % 4 == 0) ? 2 : (U % 2 == 0) ? 1 : 0;
        const int rot = (r >> (4 - tz)) & ((1 << tz) - 1);
        int gu = pos - rot;
        gu = gu < 0 ? gu + U : gu;
        gu = gu < real_units ? gu : real_units - 1;
        int gr = j * BN + r;
        gr = gr < p.Tk ? gr : p.Tk - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kg + (long)gr * k_rs + gu * 16),
                                         (__attribute__((address_space(3))) void*)(smem + S::OFF_RAWK + u0 * 16), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vg + (long)gr * v_rs + gu * 16),
                                         (__attribute__((address_space(3))) void*)(smem + S::OFF_RAWV + u0 * 16), 16, 0, 0);
    }
    float* krec = reinterpret_cast<float*>(smem + S::OFF_KREC);
    if (p.vrep_k) stage_krec(krec, p.vrep_k, b, p.Nk, p.trans_coeff ? *p.trans_coeff : 1.0f, tid, 256);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const bool xv = (p.flags & GTA_FLAG_V_TRANSFORM) != 0;
    const int r = lane;
    const int t_raw = j * BN + r;
    const bool valid = t_raw < p.Tk;
    const int t = valid ? t_raw : p.Tk - 1;
    const int n = view_of(t, p.Pk, p.invPk);
    const float* rec = krec + n * GTA_KREC;
    char* kimg_l = smem + S::OFF_IMGK;
    char* vimg_l = smem + S::OFF_IMGV;
    float ksq = 0.f;                                     // this thread's share of |k'_r|^2 (bf16-rounded values)
#pragma unroll
    for (int it = 0; it < CHP / 4; ++it) {
        const int c = wave + 4 * it;
        float x[2][8];
        if (c < ch_real && valid) {
            const uint32_t desc = p.ctab[c];
            if (ESZ == 2) {
                unpack8(*reinterpret_cast<const u32x4_t*>(smem + S::OFF_RAWK + (r * U + swz<U>(r, c)) * 16), x[0]);
                unpack8(*reinterpret_cast<const u32x4_t*>(smem + S::OFF_RAWV + (r * U + swz<U>(r, c)) * 16), x[1]);
            } else {
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    const char* raw = smem + (w2 ? S::OFF_RAWV : S::OFF_RAWK);
                    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(raw + (r * U + swz<U>(r, 2 * c)) * 16);
                    const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(raw + (r * U + swz<U>(r, 2 * c + 1)) * 16);
                    x[w2][0] = a.x; x[w2][1] = a.y; x[w2][2] = a.z; x[w2][3] = a.w;
                    x[w2][4] = bb.x; x[w2][5] = bb.y; x[w2][6] = bb.z; x[w2][7] = bb.w;
                }
            }
            if (desc) {
                f32x2_t cs[4];
                if (p.cs_k) load_cs(desc, p.cs_k + ((long)b * p.Tk + t) * 2 * p.nso2, cs);
                if (xv) chunk_apply<false, 2>(desc, rec + GTA_KREC_B, rec + GTA_KREC_D1, rec + GTA_KREC_D2, cs, x);
                else    chunk_apply<false, 1>(desc, rec + GTA_KREC_B, rec + GTA_KREC_D1, rec + GTA_KREC_D2, cs, x);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { x[0][i] = 0.f; x[1][i] = 0.f; }
        }
        const int off = (r * CHP + swz<CHP>(r, c)) * 16;
        const u32x4_t kw = pack8(x[0]);
        *reinterpret_cast<u32x4_t*>(kimg_l + off) = kw;
        *reinterpret_cast<u32x4_t*>(vimg_l + off) = pack8(x[1]);
        float kr[8];
        unpack8(kw, kr);
#pragma unroll
        for (int i = 0; i < 8; ++i) ksq += kr[i] * kr[i];
    }
    // per-tile bound for the flash kernel's deferred max: max over the tile's keys of |k'| (exactly the rows
    // the MFMA will see).  krec is dead by now (every thread is past its last chunk_apply after the barrier).
    __syncthreads();
    float* rowsq = reinterpret_cast<float*>(smem + S::OFF_KREC);
    if (p.kn) rowsq[wave * 64 + lane] = ksq;
    __syncthreads();
    if (p.kn && wave == 0) {
        float tot = rowsq[lane] + rowsq[64 + lane] + rowsq[128 + lane] + rowsq[192 + lane];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot = fmaxf(tot, __shfl_xor(tot, o));
        if (lane == 0) p.kn[((long)b * p.H + h) * n_tiles + j] = sqrtf(tot) * 1.0001f;
    }
    // LDS image -> workspace, 1 KiB contiguous per wave-instruction
    char* gimg = (char*)p.kp + (((long)b * p.H + h) * n_tiles + j) * (2L * IMG);
    constexpr int PIECES = IMG / 1024;            // per image
#pragma unroll
    for (int i = 0; i < (2 * PIECES + 3) / 4; ++i) {
        const int piece = wave + 4 * i;            // 0 .. 2*PIECES-1 : K' pieces then V' pieces
        if (piece < 2 * PIECES) {
            const char* src = (piece < PIECES ? kimg_l + piece * 1024 : vimg_l + (piece - PIECES) * 1024) + lane * 16;
            *reinterpret_cast<u32x4_t*>(gimg + piece * 1024 + lane * 16) = *reinterpret_cast<const u32x4_t*>(src);
        }
    }
}

template <int DHP, int ESZ>
int launch_prep(const GtaFwdParams& p, hipStream_t stream) {
    using S = PrepSmem<DHP, ESZ>;
    if (int rc = gta_lds_optin<&gta_kv_prep_kernel<DHP, ESZ>>(S::total(GTA_MAX_VIEWS))) return rc;
    const int n_tiles = (p.Tk + BN - 1) / BN;
    const long rows = (long)p.B * n_tiles;
    const long grid = (rows + 7) / 8 * 8 * p.H;
    if (grid > 0x7fffffffL) return GTA_E_UNSUPPORTED;
    hipLaunchKernelGGL((gta_kv_prep_kernel<DHP, ESZ>), dim3((unsigned)grid), dim3(256), S::total(p.vrep_k ? p.Nk : 0), stream, p);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}
}  // namespace

int gta_prep_dispatch(const GtaFwdParams& p, int dhp, int esz, hipStream_t stream) {
    switch (dhp) {
        case 32: return esz == 2 ? launch_prep<32, 2>(p, stream) : launch_prep<32, 4>(p, stream);
        case 64: return esz == 2 ? launch_prep<64, 2>(p, stream) : launch_prep<64, 4>(p, stream);
        case 96: return esz == 2 ? launch_prep<96, 2>(p, stream) : launch_prep<96, 4>(p, stream);
        case 128: return esz == 2 ? launch_prep<128, 2>(p, stream) : launch_prep<128, 4>(p, stream);
    }
    return GTA_E_UNSUPPORTED;
}
