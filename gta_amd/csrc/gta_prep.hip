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

// X3 (fp32 inputs, GTA_FLAG_FP32_PRODUCTS on the two-stage plan): every image is written twice -- hi = bf16(x) and lo = bf16(x - hi), 16
// significant bits together -- and the tile's four images lie [K'hi | V'hi | K'lo | V'lo] in the workspace (the first half is the plain layout).
template <int DHP, int ESZ, bool X3 = false>
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
    // the 4 x 64 row-norm partials and the k-side view records sit behind the data, sized by the actual number of views:
    // 5 workgroups per CU with 16 views' worth reserved, 6 with the 5 views of the MSN config
    static constexpr int OFF_IMGK_LO = OFF_IMGV + IMG;             // (X3 only)
    static constexpr int OFF_IMGV_LO = OFF_IMGK_LO + IMG;
    static constexpr int OFF_ROWSQ = X3 ? OFF_IMGV_LO + IMG : (ESZ == 2) ? OFF_RAWV + RAW_BYTES : OFF_IMGV + IMG;
    static constexpr int OFF_KREC = OFF_ROWSQ + 1024;
    static int total(int Nk) { return OFF_KREC + Nk * GTA_KREC * 4; }
};

// One 1-KiB LDS-DMA piece with per-lane source addresses (the row gather), as asm (hipcc puts a vmcnt(0) in front of every
// LDS access that follows the builtin form: it cannot tell the LDS ranges apart).  Consumers sit behind an explicit
// s_waitcnt vmcnt + s_barrier.
GTA_DEV void dma_piece(const char* lds_dst, const char* src) {
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)lds_dst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(lds)), "v"(src) : "memory");
}

template <int DHP, int ESZ, bool X3 = false>
__global__ __launch_bounds__(256) void gta_kv_prep_kernel(const GtaFwdParams p) {
    static_assert(!X3 || ESZ == 4, "split-bf16 images are for fp32 inputs");
    using S = PrepSmem<DHP, ESZ, X3>;
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
    // The launch's FIRST workgroups (when the attention kernel of this call wants them: p.qtiles; B * Nq of them, rounded up to a
    // multiple of 8 so that the XCD map below is unchanged) expand the q-side view records into MFMA operand tiles
    // (gta_flash_common.h) -- one view each, under the K/V tiles' memory traffic from the start (at the grid's end they were a 3.6 us tail).
    const int n_qt_wgs = (p.qtiles && p.vrep_q) ? (p.B * p.Nq + 7) / 8 * 8 : 0;
    {
        if ((int)blockIdx.x < n_qt_wgs) {
            const int vw = blockIdx.x;
            if (vw >= p.B * p.Nq) return;
            const int bq = vw / p.Nq, nq = vw - bq * p.Nq;
            float* rec = reinterpret_cast<float*>(smem);
            const float tc = p.trans_coeff ? *p.trans_coeff : 1.0f;
            const float fs = p.scale * LOG2E / (p.tau ? *p.tau : 1.0f);
            for (int idx = lane; idx < qrec_seg_count(wave, 1); idx += 64) {
                QrecItem it;
                qrec_seg_load(it, p.vrep_q, bq, p.Nq, nq, 1, wave, idx);
                qrec_seg_store(it, rec, tc, fs);
            }
            __syncthreads();
            gta_qt_build_view((char*)p.qtiles + (long)vw * GTA_QT_TILES * GTA_QT_BYTES, rec, fs, tid);
            return;
        }
    }
    int j, h, b;
    {
        const int L = blockIdx.x - n_qt_wgs, x = L & 7, i = L >> 3;
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

    // The k-side view records: requested first (plain loads into registers), written to LDS once the tile's DMAs are on
    // their way, so the latencies overlap.
    float* krec = reinterpret_cast<float*>(smem + S::OFF_KREC);
    constexpr int KQ = (GTA_MAX_VIEWS * GTA_KREC + 255) / 256;
    float kval[KQ], kmul[KQ];
    const int krec_n = p.vrep_k ? p.Nk * GTA_KREC : 0;
    {
        const float tc = p.trans_coeff ? *p.trans_coeff : 1.0f;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int i = tid + 256 * q;
            kval[q] = 0.f; kmul[q] = 0.f;
            if (256 * q < krec_n) kval[q] = p.vrep_k[krec_source(b, p.Nk, i < krec_n ? i : 0, tc, &kmul[q])];
        }
    }
    // ... and this row's (cos, sin) pairs of the so2 chunks this wave will handle (lane == key row)
    const int r = lane;
    const int t_raw = j * BN + r;
    const bool valid = t_raw < p.Tk;
    const int t = valid ? t_raw : p.Tk - 1;
    f32x4_t cs_pre[CHP / 4][2] = {};
    if (p.cs_k) {
        const float* cs_base = p.cs_k + ((long)b * p.Tk + t) * 2 * p.nso2;
#pragma unroll
        for (int it = 0; it < CHP / 4; ++it) {
            const int c = wave + 4 * it;
            const uint32_t desc = c < ch_real ? p.ctab[c] : 0u;
            if (!(desc & GTA_CHUNK_SO3) && cd_lo(desc) == GTA_HALF_SO2 && cd_hi(desc) == GTA_HALF_SO2) {
                cs_pre[it][0] = *reinterpret_cast<const f32x4_t*>(cs_base + 2 * cd_so2_lo(desc));
                cs_pre[it][1] = *reinterpret_cast<const f32x4_t*>(cs_base + 2 * cd_so2_hi(desc));
            }
        }
    }
    // raw rows -> LDS (coalesced LDS-DMA; the per-lane source address carries the swizzle): all of K, then all of V
    constexpr int NI = BN * U / 256;
#pragma unroll
    for (int w2 = 0; w2 < 2; ++w2) {
        const char* g = w2 ? vg : kg;
        const long rs = w2 ? v_rs : k_rs;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int u0 = (wave * NI + i) * 64, u = u0 + lane;
            const int r = u / U, pos = u - r * U;
            const int rot = swz_rot<U>(r);
            int gu = pos - rot;
            gu = gu < 0 ? gu + U : gu;
            gu = gu < real_units ? gu : real_units - 1;
            int gr = j * BN + r;
            gr = gr < p.Tk ? gr : p.Tk - 1;
#if defined(GTA_PREP_ABL) && GTA_PREP_ABL >= 4       // (levels 4, 5: no loads)
            if (p.Tk < 0)
#endif
            dma_piece(smem + (w2 ? S::OFF_RAWV : S::OFF_RAWK) + u0 * 16, g + (long)gr * rs + gu * 16);
        }
    }
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const int i = tid + 256 * q;
        if (i < krec_n) krec[i] = kval[q] * kmul[q];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this wave's pieces have landed, its records are written
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // a use of the preloaded pairs here: hipcc's own wait for them lands at this point (nothing is outstanding) instead of
    // in the V pass, where it would wait for the K' stores
#pragma unroll
    for (int it = 0; it < CHP / 4; ++it) asm volatile("" ::"v"(cs_pre[it][0]), "v"(cs_pre[it][1]));

    const bool xv = (p.flags & GTA_FLAG_V_TRANSFORM) != 0;
    const int n = view_of(t, p.Pk, p.invPk);
    const float* rec = krec + n * GTA_KREC;
    const float* cs_base = p.cs_k ? p.cs_k + ((long)b * p.Tk + t) * 2 * p.nso2 : nullptr;
    float ksq = 0.f;                                     // this thread's share of |k'_r|^2 (bf16-rounded values)

    // One pass over a tile's rows (K, then V): an 8-channel chunk per wave and iteration, lane == key row.  The chunk's
    // kind is wave-uniform; every kind runs its own load -> rho -> pack -> store body, so no register arrays are merged
    // behind the branches (the phi copies of a shared body were a third of this kernel's VALU instructions, and its
    // compute phase is VALU-issue bound: profiles/r02/README.md).
    // (JOINTC: K and V chunk of the row side by side in one pass -- one load of the view matrices / (cos, sin) pairs for both, one
    //  barrier less.  Faster where a tile is small (dh <= 64: 13.9 against 15.2 us at CLEVR-TR's encoder shape, r02); at dh = 96 the
    //  two-pass form wins because K' goes out while V is transformed.)
    auto pass = [&](auto ISK, auto XF, auto JOINTC) {
        constexpr bool is_k = decltype(ISK)::value, xf = decltype(XF)::value, joint = decltype(JOINTC)::value;
        const char* raw = smem + (is_k ? S::OFF_RAWK : S::OFF_RAWV);
        char* img = smem + (is_k ? S::OFF_IMGK : S::OFF_IMGV);
        char* img_lo = smem + (is_k ? S::OFF_IMGK_LO : S::OFF_IMGV_LO);      // (X3)
#pragma unroll
        for (int it = 0; it < CHP / 4; ++it) {
            const int c = wave + 4 * it;
            const int off = (r * CHP + swz<CHP>(r, c)) * 16;
            auto run = [&](auto&& apply) {
                float x[1][8];
                if (ESZ == 2) {
                    unpack8(*reinterpret_cast<const u32x4_t*>(raw + (r * U + swz<U>(r, c)) * 16), x[0]);
                } else {
                    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(raw + (r * U + swz<U>(r, 2 * c)) * 16);
                    const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(raw + (r * U + swz<U>(r, 2 * c + 1)) * 16);
                    x[0][0] = a.x; x[0][1] = a.y; x[0][2] = a.z; x[0][3] = a.w;
                    x[0][4] = bb.x; x[0][5] = bb.y; x[0][6] = bb.z; x[0][7] = bb.w;
                }
                apply(x);
                if constexpr (joint) {                     // the V chunk of the same row and position, the same matrices
                    float y[1][8];
                    const char* rawv = smem + S::OFF_RAWV;
                    if (ESZ == 2) {
                        unpack8(*reinterpret_cast<const u32x4_t*>(rawv + (r * U + swz<U>(r, c)) * 16), y[0]);
                    } else {
                        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(rawv + (r * U + swz<U>(r, 2 * c)) * 16);
                        const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(rawv + (r * U + swz<U>(r, 2 * c + 1)) * 16);
                        y[0][0] = a.x; y[0][1] = a.y; y[0][2] = a.z; y[0][3] = a.w;
                        y[0][4] = bb.x; y[0][5] = bb.y; y[0][6] = bb.z; y[0][7] = bb.w;
                    }
                    if (xv) apply(y);
                    *reinterpret_cast<u32x4_t*>(smem + S::OFF_IMGV + off) = pack8(y[0]);
                }
                const u32x4_t w = pack8(x[0]);
                *reinterpret_cast<u32x4_t*>(img + off) = w;
                if constexpr (X3) {                        // the residual the rounding left: x = hi + lo to 2^-17
                    float kr[8], lo8[8];
                    unpack8(w, kr);
#pragma unroll
                    for (int i = 0; i < 8; ++i) lo8[i] = x[0][i] - kr[i];
                    *reinterpret_cast<u32x4_t*>(img_lo + off) = pack8(lo8);
                    if (is_k) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) ksq += x[0][i] * x[0][i];
                    }
                } else if (is_k) {
                    float kr[8];
                    unpack8(w, kr);
#pragma unroll
                    for (int i = 0; i < 8; ++i) ksq += kr[i] * kr[i];
                }
            };
            if (c < ch_real && valid) {
#ifdef GTA_PREP_ABL          // timing-only ablation (tools/ablate_prep.sh): no transform
                const uint32_t desc = 0u;
#else
                const uint32_t desc = xf ? p.ctab[c] : 0u;
#endif
                const uint32_t lo = cd_lo(desc), hi = cd_hi(desc);
                if (desc == 0) {
                    run([](float (*)[8]) {});
                } else if (desc & GTA_CHUNK_SO3) {
                    run([&](float (*x)[8]) {
                        float M1[12], M2[40];
                        lds_loadN4<3>(rec + GTA_KREC_D1, M1);
                        lds_loadN4<10>(rec + GTA_KREC_D2, M2);
                        mat3_apply_p4(M1, x[0]); mat5_apply_p8(M2, x[0] + 3);
                    });
                } else if (lo == GTA_HALF_SE3 && hi == GTA_HALF_SE3) {
                    run([&](float (*x)[8]) {
                        float M[16];
                        lds_load16(rec + GTA_KREC_B, M);
                        mat4_apply(M, x[0]); mat4_apply(M, x[0] + 4);
                    });
                } else if (lo == GTA_HALF_SO2 && hi == GTA_HALF_SO2 && cs_base) {
                    run([&](float (*x)[8]) {
                        const f32x4_t t0 = cs_pre[it][0], t1 = cs_pre[it][1];
                        rot2_apply<false>(t0.x, t0.y, x[0]); rot2_apply<false>(t0.z, t0.w, x[0] + 2);
                        rot2_apply<false>(t1.x, t1.y, x[0] + 4); rot2_apply<false>(t1.z, t1.w, x[0] + 6);
                    });
                } else {                                 // mixed halves (no shipped config): the generic body
                    run([&](float (*x)[8]) {
                        f32x2_t cs[4];
                        if (cs_base) load_cs(desc, cs_base, cs);
                        chunk_apply<false, 1>(desc, rec + GTA_KREC_B, rec + GTA_KREC_D1, rec + GTA_KREC_D2, cs, x);
                    });
                }
            } else {
                *reinterpret_cast<u32x4_t*>(img + off) = u32x4_t{0u, 0u, 0u, 0u};
                if constexpr (X3) *reinterpret_cast<u32x4_t*>(img_lo + off) = u32x4_t{0u, 0u, 0u, 0u};
                if constexpr (joint) *reinterpret_cast<u32x4_t*>(smem + S::OFF_IMGV + off) = u32x4_t{0u, 0u, 0u, 0u};
            }
        }
    };
    // LDS image -> workspace, 1 KiB contiguous per wave-instruction
    char* gimg = (char*)p.kp + (((long)b * p.H + h) * n_tiles + j) * ((X3 ? 4L : 2L) * IMG);
    constexpr int PIECES = IMG / 1024;            // per image
    auto store_image = [&](const char* img_l, char* g) {
#pragma unroll
        for (int i = 0; i < (PIECES + 3) / 4; ++i) {
            const int piece = wave + 4 * i;
#if defined(GTA_PREP_ABL) && (GTA_PREP_ABL == 3 || GTA_PREP_ABL == 5)       // (levels 3, 5: no image stores)
            if (piece < PIECES && p.Tk < 0)
#else
            if (piece < PIECES)
#endif
                *reinterpret_cast<u32x4_t*>(g + piece * 1024 + lane * 16) = *reinterpret_cast<const u32x4_t*>(img_l + piece * 1024 + lane * 16);
        }
    };

    constexpr bool JOINT = DHP <= 64 && ESZ == 2;                  // (fp32 inputs: the V image is not in place, keep the two passes)
    if constexpr (JOINT) pass(std::true_type{}, std::true_type{}, std::true_type{});
    else pass(std::true_type{}, std::true_type{}, std::false_type{});      // K rows -> K' image (in place for bf16 input)
    float* rowsq = reinterpret_cast<float*>(smem + S::OFF_ROWSQ);
    if (p.kn) rowsq[wave * 64 + lane] = ksq;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // this wave's K' units and partials are written
    __builtin_amdgcn_s_barrier();                                  // K' image complete
    asm volatile("" ::: "memory");
    store_image(smem + S::OFF_IMGK, gimg);                         // K' goes out while V is transformed
    if constexpr (X3) store_image(smem + S::OFF_IMGK_LO, gimg + 2 * IMG);
    // per-tile bound for the flash kernel's deferred max: max over the tile's keys of |k'| (exactly the rows the MFMA will see)
    if (p.kn && wave == 0) {
        float tot = rowsq[lane] + rowsq[64 + lane] + rowsq[128 + lane] + rowsq[192 + lane];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot = fmaxf(tot, __shfl_xor(tot, o));
        if (lane == 0) p.kn[((long)b * p.H + h) * n_tiles + j] = sqrtf(tot) * (X3 ? 1.0002f : 1.0001f);
    }
    if constexpr (!JOINT) {
#if !defined(GTA_PREP_ABL) || GTA_PREP_ABL < 2      // (level 2: the V rows go out as they came in)
        if (xv) pass(std::false_type{}, std::true_type{}, std::false_type{}); else pass(std::false_type{}, std::false_type{}, std::false_type{});
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (the K' stores may still be in flight)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    store_image(smem + S::OFF_IMGV, gimg + IMG);
    if constexpr (X3) store_image(smem + S::OFF_IMGV_LO, gimg + 3 * IMG);
}

template <int DHP, int ESZ, bool X3 = false>
int launch_prep(const GtaFwdParams& p, hipStream_t stream) {
    using S = PrepSmem<DHP, ESZ, X3>;
    if (int rc = gta_lds_optin<&gta_kv_prep_kernel<DHP, ESZ, X3>>(S::total(GTA_MAX_VIEWS))) return rc;
    const int n_tiles = (p.Tk + BN - 1) / BN;
    const long rows = (long)p.B * n_tiles;
    long grid = (rows + 7) / 8 * 8 * p.H;
    if (p.qtiles && p.vrep_q) grid += ((long)p.B * p.Nq + 7) / 8 * 8;        // the q-side tile builders (see the kernel's head)
    if (grid > 0x7fffffffL) return GTA_E_UNSUPPORTED;
    hipLaunchKernelGGL((gta_kv_prep_kernel<DHP, ESZ, X3>), dim3((unsigned)grid), dim3(256), S::total(p.vrep_k ? p.Nk : 0), stream, p);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}
}  // namespace

int gta_prep_dispatch(const GtaFwdParams& p, int dhp, int esz, hipStream_t stream) {
    if (p.flags & GTA_FLAG_FP32_PRODUCTS) {             // split-bf16 images (gta_fwd2_x3_takes: fp32 inputs, dh <= 64)
        if (esz != 4) return GTA_E_UNSUPPORTED;
        switch (dhp) {
            case 32: return launch_prep<32, 4, true>(p, stream);
            case 64: return launch_prep<64, 4, true>(p, stream);
        }
        return GTA_E_UNSUPPORTED;
    }
    switch (dhp) {
        case 32: return esz == 2 ? launch_prep<32, 2>(p, stream) : launch_prep<32, 4>(p, stream);
        case 64: return esz == 2 ? launch_prep<64, 2>(p, stream) : launch_prep<64, 4>(p, stream);
        case 96: return esz == 2 ? launch_prep<96, 2>(p, stream) : launch_prep<96, 4>(p, stream);
        case 128: return esz == 2 ? launch_prep<128, 2>(p, stream) : launch_prep<128, 4>(p, stream);
    }
    return GTA_E_UNSUPPORTED;
}
