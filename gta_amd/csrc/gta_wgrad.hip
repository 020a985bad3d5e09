// gta_wgrad.hip -- weight gradient of a Linear as a hand-written gfx950 kernel (gta_wgrad of include/gta_block.h):
//
//     dW[n][k] = sum_m G[m][n] * X[m][k]      G = d(out) [M,N] bf16, X = the layer's input [M,K] bf16, dW fp32
//     db[n]    = sum_m G[m][n]                (optional; the bias gradient falls out of the G fragments)
//
// This is the "TN" GEMM of the block's backward (autograd over nn.Linear, source/layers.py:161-164,388-395,430): the
// reduction runs over the M = batch x tokens rows, the SLOW index of both row-major operands, and the outputs are small
// (768..2304 squared).  hipBLASLt's best kernel for it reaches 370-540 TFLOP/s at the MSN shapes (profiles/r02/README.md).
//
// Structure.  One workgroup (4 waves, one per SIMD and CU) owns a 256 x 256 tile of dW over a slice of the tokens
// (split-M, partial tiles reduced in a fixed order by wgrad_finish_kernel: deterministic).  Per step of 32 tokens the two
// operand tiles [32][256] are brought in by LDS-DMA (global_load_lds, 16 B per lane) as [8 token][64 column] sub-tiles of
// 1 KiB -- one DMA instruction each: eight lanes fetch one whole 128-B line of a row, and the lane-linear LDS image of
// the instruction IS the row-major sub-tile with 128-B rows (16-B chunks XOR-swizzled by the row through the SOURCE
// addresses).  MFMA operands need 8 consecutive TOKENS of one column per lane; ds_read_b64_tr_b16 delivers 4 (a 16-lane
// group reads a [4 token][16 column] block and transposes it), so an operand of v_mfma_f32_32x32x16_bf16 is two such
// reads.  With the swizzle the 32 lanes serviced together cover all 64 banks once: conflict-free (SQ_LDS_BANK_CONFLICT 0).
// A ring of four 32-KiB stages keeps three steps of DMA in flight under the MFMAs; the ring turns in the MIDDLE of a step,
// between its two 16-token MFMA blocks, so every fragment read flies under an MFMA block.  Each wave owns 128 x 128 of the
// tile: 16 accumulators of 32 x 32 (256 registers: the AGPR half of the wave's file), 32 MFMAs against 32 transpose-reads and
// 8 DMA instructions per step.  (WG_WAVES = 8 is the earlier tiling: 128 x 64 per wave, two waves per SIMD.)
//
// What bounds it (profiles/r02/README.md).  The steady-state step keeps everything that is not an MFMA BETWEEN the MFMAs of
// its two blocks (one fragment's two transpose-reads or one DMA request per gap, pinned by scheduling barriers), runs on a
// running scalar source pointer and writes M0 with one s_add per request: 75 instructions per step and wave, 16 of them
// MFMAs (the first form of this loop had 127, with ~60 scalar / address instructions and the 12 reads of a block standing
// between the blocks -- where, with the eight waves leaving the barrier in step, all four matrix pipes idle).  Switch
// experiments on the final loop, dW[2304,768] over 40 960 tokens, finish kernel (14 us) included: 191.6 us as is; 167.3 without
// the DMA (stale LDS); 116.4 without the fragment reads (the DMA's own floor is 114: 10-11 TB/s from L2 into the LDS);
// 190.2 without the barrier; **105.8 with all three off = the MFMA stream alone**.  Inside the kernel the parts add (92 + 60 + 24
// us) rather than overlap, but that is not a property of fragment reads beside the matrix pipe: tests/probes/probe_lds_mfma.hip
// (the same MFMA stream with the reads in its gaps, no DMA, no barrier) runs at 2 004-2 066 TFLOP/s against 2 144-2 194 without
// the reads, at ~230 LDS bytes per clock and CU.  The cost sits in the per-step barrier and in the DMA writes next door; the firm
// bound is the DMA's own floor, 114 us at the 10-11 TB/s the L2 -> LDS path delivers (73 % of this kernel's time).  The read form
// does not matter either (one ds_read_b128 per pair of transpose-reads: 193 us).  The 128 x 128 wave tile on four waves (a third
// fewer fragment bytes, half the waves at the barrier) took 191.6 -> 169.6 us on that shape (135 -> 115, 133 -> 114, 241 -> 197 us
// on the layer's others; the experiments above are of the 8-wave tiling).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gta_common.h"
#include "../../include/gta_hip.h"
#include "../../include/gta_block.h"

namespace {

constexpr int WG_TILE = 256;                       // dW tile edge (both ways)
constexpr int WG_BT = 32;                          // tokens per step (two 16-token groups)
constexpr int WG_OPER = WG_BT * WG_TILE * 2;       // one operand tile [32][256] bf16, bytes (16 KiB)
constexpr int WG_STAGE = 2 * WG_OPER;              // G tile + X tile (32 KiB)
constexpr int WG_NSTAGE = 4;        // ring depth: NSTAGE - 1 steps of DMA in flight under the MFMAs
constexpr int WG_LDS = WG_NSTAGE * WG_STAGE;
constexpr int WG_WAVES = 4;          // 4: 2 (n) x 2 (k), 128 x 128 of the tile per wave, one wave per SIMD, accumulators
                                                   //    in AGPRs;  8: 2 x 4, 128 x 64 per wave, two waves per SIMD
static_assert(WG_WAVES == 4 || WG_WAVES == 8, "wave tilings of the 256 x 256 workgroup tile");
constexpr int WG_WK = WG_WAVES / 2;                // waves along k
constexpr int WG_JB = 256 / WG_WK / 32;            // 32-column X fragments per wave (4 or 2); G fragments: always 4
constexpr int WG_NF = 4 + WG_JB;                   // fragments a wave reads per 16-token block
constexpr int WG_NM = 4 * WG_JB;                   // MFMAs per block
constexpr int WG_DMA_PER_WAVE = 32 / WG_WAVES;     // 1-KiB sub-tiles a wave brings per step

struct WgradParams {
    const char* g;       // [M][ldg] bf16
    const char* x;       // [M][ldx] bf16
    float* part;         // [S][N][K] partial tiles (or the final dW when S == 1, row pitch ldo)
    float* db_part;      // [S][N] or nullptr
    long ldg, ldx, ldo;  // elements
    int M, N, K;
    int tiles_k;         // K / 256
    int n_tiles;         // (N / 256) * tiles_k
    int steps_per_split; // 32-token steps per split (the last split may have fewer)
};

template <int IMM>
GTA_DEV u32x2_t tr16(uint32_t addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(IMM));
    return v;
}

GTA_DEV float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
GTA_DEV float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

template <bool BIAS>
__global__ __launch_bounds__(64 * WG_WAVES) void wgrad_kernel(WgradParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);             // wave-uniform: everything derived from it stays scalar
    const int wn = wave / WG_WK, wk = wave % WG_WK;
    // XCD-aware work map: workgroup L runs on XCD L % 8.  Workgroups of one token split share their operand tiles (the G
    // tile of a row of dW tiles, the X tile of a column), so each XCD takes a CONTIGUOUS run of the (split, tile) list and
    // its L2 serves the re-reads (measured: the kernel is bound by the fabric otherwise, 4.8 TB/s into the LDS).
    const int total = gridDim.x, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + jx;
    const int tile = v % p.n_tiles, split = v / p.n_tiles;
    const int n0 = (tile / p.tiles_k) * WG_TILE, k0 = (tile % p.tiles_k) * WG_TILE;
    const int step0 = split * p.steps_per_split;
    int n_steps = p.M / WG_BT - step0;                                     // 32-token steps of this split
    if (n_steps > p.steps_per_split) n_steps = p.steps_per_split;

    // ---- DMA plan: 32 sub-tiles of 1 KiB per step; the first half of the waves bring G (16 sub-tiles), the second X ----
    // A sub-tile is [8 tokens][64 columns]: lane l fetches 16 B of row l/8 -- eight lanes cover one whole 128-B line (the
    // vector-memory path works per line: 64-B row segments cost twice the address work for the same bytes).  Which of
    // the row's eight 16-B chunks a lane fetches is swizzled, chunk = (l % 8) ^ 4*((row >> 1) & 1), so that in the
    // lane-linear LDS image (128-B rows) rows 0..3 of a 64-B column band land in four different bank quarters.
    const bool is_x = wave >= WG_WAVES / 2;
    const char* src = is_x ? p.x : p.g;
    const long ld = is_x ? p.ldx : p.ldg;
    const int c0 = is_x ? k0 : n0;
    const int piece0 = (wave % (WG_WAVES / 2)) * WG_DMA_PER_WAVE;          // this wave's sub-tiles of its operand
    const int drow = lane >> 3, dchunk = (lane & 7) ^ (((drow >> 1) & 1) << 2);
    const unsigned lane_src = (unsigned)(((long)drow * ld + dchunk * 8) * 2);   // (< 2^32: checked by the launcher)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;

    // scalar-base form of the LDS-DMA (gta_common.h dma_group): global address = SGPR pair + the lane's 32-bit offset,
    // LDS address = M0 + 16 * lane -- no vector arithmetic per piece
    auto stage_dma = [&](int step, int stage) {
        const long m0 = (long)(step0 + step) * WG_BT;
        const char* sbase = src + (m0 * ld + c0) * 2;
        const uint32_t lbase = lds0 + stage * WG_STAGE + (is_x ? WG_OPER : 0);
#pragma unroll
        for (int q = 0; q < WG_DMA_PER_WAVE; ++q) {
            const int pc = piece0 + q, t8 = pc >> 2, cg = pc & 3;           // [token group of 8][column group of 64]
            dma_group<1>(lbase + pc * 1024, sbase + ((long)t8 * 8 * ld + cg * 64) * 2, lane_src);
        }
    };
    // wait until at most `groups` of my DMA groups (WG_DMA_PER_WAVE instructions each) are still in flight
    auto dma_wait = [&](int groups) {
        if (groups >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * WG_DMA_PER_WAVE) : "memory");
        else if (groups == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WG_DMA_PER_WAVE) : "memory");
        else if (groups == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WG_DMA_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    static_assert(WG_NSTAGE >= 2 && WG_NSTAGE <= 5, "ring depth (dma_wait counts up to three groups in flight beside the awaited one)");

    // ---- operand addresses.  An MFMA operand is [32 columns][16 tokens]: lanes 0-31 hold tokens 0-7 (sub-tile row group
    // t8 = 2u), lanes 32-63 tokens 8-15 (t8 = 2u + 1, 4 KiB further); a 16-lane group reads the [4 token][16 column] block
    // of rows 0-3 (first read) or 4-7 (second, +512 B) of its 16 columns: lane p of the group points at row p/4, columns
    // 4(p%4)..+3, i.e. 16-B chunk (16-column block)*2 + (p%4)/2 of the 64-column band, un-swizzled as it was stored.
    // The two 32-column halves of a band differ in chunk bit 2, which the swizzle XORs: two lane bases, ^64 B apart. ----
    const int grp = lane >> 4, pl = lane & 15, rrow = pl >> 2;
    const int chunk0 = ((grp & 1) * 2 + ((pl & 3) >> 1)) ^ (((rrow >> 1) & 1) << 2);
    const uint32_t la0 = (uint32_t)((lane >> 5) * 4096 + rrow * 128 + chunk0 * 16 + (pl & 1) * 8);
    const uint32_t la1 = la0 ^ 64u;
    const uint32_t a_base0 = lds0 + wn * 2048 + la0, a_base1 = lds0 + wn * 2048 + la1;                 // G bands wn*2, wn*2+1
    const uint32_t b_base0 = lds0 + WG_OPER + wk * (WG_JB / 2) * 1024 + la0, b_base1 = b_base0 ^ 64u;     // X bands of wk

    f32x16_t acc[4][WG_JB];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WG_JB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float dbs[4] = {0.f, 0.f, 0.f, 0.f};

    u32x2_t af[2][4][2], bf[2][WG_JB][2];                                   // [buffer][32-column group][token half]
    // fragments of the 16-token group u (sub-tile row groups 2u, 2u+1) of the stage at byte offset `so`
    // operands of block `cur` as MFMA registers; one MFMA; the bias share of one G fragment (token sums of its 32 columns)
    auto a_of = [&](int cur, int i) {
        const u32x4_t av = {af[cur][i][0].x, af[cur][i][0].y, af[cur][i][1].x, af[cur][i][1].y};
        return av;
    };
    auto mfma_one = [&](int cur, auto I, auto J) {
        constexpr int i = decltype(I)::value, j = decltype(J)::value;
        const u32x4_t bv = {bf[cur][j][0].x, bf[cur][j][0].y, bf[cur][j][1].x, bf[cur][j][1].y};
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a_of(cur, i)), __builtin_bit_cast(bf16x8_t, bv),
                                                            acc[i][j], 0, 0, 0);
    };
    auto bias_acc = [&](int cur, int i) {
        const u32x4_t av = a_of(cur, i);
        dbs[i] += ((bf16_lo(av.x) + bf16_hi(av.x)) + (bf16_lo(av.y) + bf16_hi(av.y))) +
                  ((bf16_lo(av.z) + bf16_hi(av.z)) + (bf16_lo(av.w) + bf16_hi(av.w)));
    };
    auto mfma_block = [&](int cur, bool bias_wave) {
        if (bias_wave) {
#pragma unroll
            for (int i = 0; i < 4; ++i) bias_acc(cur, i);
        }
        gta_static_for<4>([&](auto I) { gta_static_for<WG_JB>([&](auto J) { mfma_one(cur, I, J); }); });
    };
    // fragment reads of one 32-column group (i < 4: G group i, else X group i - 4) of token group U of the stage at `so`
    auto load_frag = [&](uint32_t so, auto U, int buf, auto IDX) {
        constexpr int u = decltype(U)::value, idx = decltype(IDX)::value;
        constexpr int f = idx < 4 ? idx : idx - 4, off = (2 * u * 4 + (f >> 1)) * 1024;
        if constexpr (idx < 4) {
            const uint32_t ab = ((f & 1) ? a_base1 : a_base0) + so;
            af[buf][f][0] = tr16<off>(ab);
            af[buf][f][1] = tr16<off + 512>(ab);
        } else {
            const uint32_t bb = ((f & 1) ? b_base1 : b_base0) + so;
            bf[buf][f][0] = tr16<off>(bb);
            bf[buf][f][1] = tr16<off + 512>(bb);
        }
    };
    auto load_frags = [&](uint32_t so, auto U, int buf) {
        gta_static_for<WG_NF>([&](auto IDX) { load_frag(so, U, buf, IDX); });
    };
    const bool bias_wave = BIAS && wk == 0 && k0 == 0;

    // prologue: the ring's first NSTAGE steps are requested; step 0 is awaited
    const int pre = n_steps < WG_NSTAGE ? n_steps : WG_NSTAGE;
    for (int s0 = 0; s0 < pre; ++s0) stage_dma(s0, s0);
    dma_wait(pre - 1);
    __builtin_amdgcn_s_barrier();
    load_frags(0u, std::integral_constant<int, 0>{}, 0);

    int st = 0;                                                             // ring stage of step t
    // The steady-state requests (step t + NSTAGE while step t runs) keep a running scalar source pointer -- this wave's
    // row group and column band folded in -- and write M0 with one s_add: LDS address = M0 + offset + 16 * lane and global
    // address = base + offset + lane offset share the instruction's immediate, so piece q (128 B further in memory,
    // 1 KiB further in LDS) takes M0 = stage base + 896 q and offset 128 q.
    const char* dma_sb = src + (((long)(step0 + WG_NSTAGE) * WG_BT + (piece0 >> 2) * 8) * ld + c0) * 2;
    const long dma_stride = (long)WG_BT * ld * 2;
    const uint32_t lb_wave = lds0 + (is_x ? WG_OPER : 0) + piece0 * 1024;
    const long dma_row8 = 8 * ld * 2;                                       // bytes between token groups of 8
    auto dma_piece = [&](uint32_t lbase, auto Q) {                          // piece q of this wave: token group q / 4, column group q % 4
        constexpr int q = decltype(Q)::value, qt = q >> 2, qc = q & 3;
        const unsigned voff = lane_src;                                     // (asm operands do not capture by themselves)
        const char* sb = dma_sb + qt * dma_row8;
        asm volatile("s_add_i32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4"
                     ::"s"(lbase), "v"(voff), "s"(sb), "n"(qt * 4096 + qc * 896), "n"(qc * 128) : "memory");
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
#define WG_SB __builtin_amdgcn_sched_barrier(0)
    // One steady-state step (t + NSTAGE < n_steps: the ring is full behind it, no conditions).  Everything that is not an
    // MFMA sits BETWEEN the MFMAs of a block, pinned by scheduling barriers: the eight waves leave the barrier in step, so
    // whatever runs outside the MFMA blocks runs with the matrix pipes of all four SIMDs idle (the loop took 62 cycles per
    // MFMA with the memory side switched off when fragment reads, DMA requests and address arithmetic stood between the blocks).
    auto steady_step = [&](auto BW) {
        constexpr bool bw = decltype(BW)::value;
        // A: token group 0 of step t; group 1's fragments are requested under its MFMAs
        const uint32_t so = (uint32_t)(st * WG_STAGE);
        WG_SB; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); WG_SB;
        gta_static_for<WG_NM>([&](auto G) {
            constexpr int g = decltype(G)::value;
            mfma_one(0, std::integral_constant<int, g / WG_JB>{}, std::integral_constant<int, g % WG_JB>{});
            if constexpr (g < WG_NF) load_frag(so, I1{}, 1, G);
            if constexpr (bw && g >= WG_NM - 4) bias_acc(0, g - (WG_NM - 4));
            WG_SB;
        });
        // B: token group 1.  The ring turns first: step t + 1 has landed (my share, then everyone's), stage st is read out
        // and takes step t + NSTAGE; the first fragments of step t + 1 are requested under these MFMAs.
        const int st1 = st + 1 == WG_NSTAGE ? 0 : st + 1;
        const uint32_t so1 = (uint32_t)(st1 * WG_STAGE), lbase = lb_wave + so;
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(%0)" ::"n"((WG_NSTAGE - 2) * WG_DMA_PER_WAVE) : "memory");
        __builtin_amdgcn_s_barrier();
        WG_SB;
        gta_static_for<WG_NM>([&](auto G) {
            constexpr int g = decltype(G)::value;
            mfma_one(1, std::integral_constant<int, g / WG_JB>{}, std::integral_constant<int, g % WG_JB>{});
            if constexpr (g < WG_NF) load_frag(so1, I0{}, 0, G);
            if constexpr (g >= WG_NM - WG_DMA_PER_WAVE) dma_piece(lbase, std::integral_constant<int, g - (WG_NM - WG_DMA_PER_WAVE)>{});
            if constexpr (bw && g >= WG_NM - 4) bias_acc(1, g - (WG_NM - 4));
            WG_SB;
        });
        dma_sb += dma_stride;
        st = st1;
    };
#undef WG_SB
    // The ring's last NSTAGE steps: conditions on what is still to request and to await.
    auto tail_step = [&](int t) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        load_frags((uint32_t)(st * WG_STAGE), std::integral_constant<int, 1>{}, 1);
        mfma_block(0, bias_wave);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int st1 = st + 1 == WG_NSTAGE ? 0 : st + 1;
        if (t + 1 < n_steps) {
            const int last_issued = t + WG_NSTAGE - 1 < n_steps - 1 ? t + WG_NSTAGE - 1 : n_steps - 1;
            dma_wait(last_issued - (t + 1));
            __builtin_amdgcn_s_barrier();
            load_frags((uint32_t)(st1 * WG_STAGE), std::integral_constant<int, 0>{}, 0);
        }
        mfma_block(1, bias_wave);
        st = st1;
    };
    int t = 0;
    if (bias_wave) { for (; t + WG_NSTAGE < n_steps; ++t) steady_step(std::true_type{}); }
    else           { for (; t + WG_NSTAGE < n_steps; ++t) steady_step(std::false_type{}); }
    for (; t < n_steps; ++t) tail_step(t);

    // ---- epilogue: accumulators -> this split's partial tile (rows n, 128-B segments along k) ----
    float* out = p.part + (long)split * p.N * p.ldo;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WG_JB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 128 + i * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                const int k = k0 + wk * (32 * WG_JB) + j * 32 + (lane & 31);
                out[(long)n * p.ldo + k] = acc[i][j][r];
            }
    if (BIAS && wk == 0 && k0 == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = dbs[i] + __shfl_xor(dbs[i], 32, 64);           // the two token halves of the fragment
            if (lane < 32) p.db_part[(long)split * p.N + n0 + wn * 128 + i * 32 + lane] = s;
        }
    }
}

// dW[n][k] = sum_s part[s][n][k] (fixed order), db[n] = sum_s db_part[s][n]
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ part, int S, long slab, long n4,
                                                          float* __restrict__ dw, const float* __restrict__ db_part,
                                                          float* __restrict__ db, int N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        f32x4_t s = reinterpret_cast<const f32x4_t*>(part)[i];
        for (int q = 1; q < S; ++q) {
            const f32x4_t v = reinterpret_cast<const f32x4_t*>(part + q * slab)[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<f32x4_t*>(dw)[i] = s;
    }
    if (db && i < N) {
        float s = db_part[i];
        for (int q = 1; q < S; ++q) s += db_part[(long)q * N + i];
        db[i] = s;
    }
}

struct Split { int S, steps; };
inline Split choose_split(long m, long n, long k) {
    const int tiles = (int)((n / WG_TILE) * (k / WG_TILE));
    const int total = (int)(m / WG_BT);
    int want = 256 / tiles;                     // one workgroup per CU
    if (want < 1) want = 1;
    if (want > total / 8) want = total / 8 > 0 ? total / 8 : 1;            // at least eight 32-token steps per workgroup
    const int steps = (total + want - 1) / want;
    return {(total + steps - 1) / steps, steps};
}

}  // namespace

extern "C" {

int gta_wgrad_supported(int64_t m, int64_t n, int64_t k) {
    return (m > 0 && n > 0 && k > 0 && m % WG_BT == 0 && n % WG_TILE == 0 && k % WG_TILE == 0 && m <= (1ll << 30)) ? 1 : 0;
}

int64_t gta_wgrad_workspace_bytes(int64_t m, int64_t n, int64_t k) {
    if (!gta_wgrad_supported(m, n, k)) return 0;
    const Split sp = choose_split(m, n, k);
    return (int64_t)sp.S * (n * k + n) * (int64_t)sizeof(float);
}

int gta_wgrad(const void* g, int64_t ldg, const void* x, int64_t ldx, int64_t m, int64_t n, int64_t k, float* dw, float* dbias,
              void* workspace, int64_t workspace_bytes, void* stream) {
    if (!g || !x || !dw || !workspace || ldg < n || ldx < k) return GTA_E_BADARG;
    if (!gta_wgrad_supported(m, n, k)) return GTA_E_UNSUPPORTED;
    if (ldg % 8 != 0 || ldx % 8 != 0 || (reinterpret_cast<uintptr_t>(g) & 15) || (reinterpret_cast<uintptr_t>(x) & 15) ||
        (reinterpret_cast<uintptr_t>(dw) & 15))
        return GTA_E_BADARG;
    if (ldg > (1ll << 27) || ldx > (1ll << 27)) return GTA_E_UNSUPPORTED;     // the DMA's per-lane offsets are 32-bit
    if (workspace_bytes < gta_wgrad_workspace_bytes(m, n, k)) return GTA_E_BADARG;
    const Split sp = choose_split(m, n, k);
    WgradParams p;
    p.g = static_cast<const char*>(g);
    p.x = static_cast<const char*>(x);
    p.part = static_cast<float*>(workspace);
    p.db_part = dbias ? p.part + (long)sp.S * n * k : nullptr;
    p.ldg = ldg; p.ldx = ldx; p.ldo = k;
    p.M = (int)m; p.N = (int)n; p.K = (int)k;
    p.tiles_k = (int)(k / WG_TILE);
    p.n_tiles = (int)(n / WG_TILE) * p.tiles_k;
    p.steps_per_split = sp.steps;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = dbias ? gta_lds_optin<&wgrad_kernel<true>>(WG_LDS) : gta_lds_optin<&wgrad_kernel<false>>(WG_LDS);
    if (rc != 0) return rc;
    if (dbias) hipLaunchKernelGGL(wgrad_kernel<true>, dim3(p.n_tiles * sp.S), dim3(64 * WG_WAVES), WG_LDS, s, p);
    else       hipLaunchKernelGGL(wgrad_kernel<false>, dim3(p.n_tiles * sp.S), dim3(64 * WG_WAVES), WG_LDS, s, p);
    if (hipGetLastError() != hipSuccess) return GTA_E_LAUNCH;
    const long n4 = n * k / 4;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p.part, sp.S, (long)n * k, n4, dw,
                       p.db_part, dbias, (int)n);
    return hipGetLastError() == hipSuccess ? GTA_OK : GTA_E_LAUNCH;
}

}  // extern "C"
