// gta_fwd_params.h -- kernel argument block of the fused forward (host fills it in gta_abi.cpp).
#pragma once
#include <stdint.h>

struct GtaFwdParams {
    const void* q; const void* k; const void* v; void* o; float* lse;
    void* kp;                                   // K'/V' tile-image workspace (two-stage path)
    float* kn;                                  // per key tile: max_k |k'_k| of the bf16 image rows, [B,H,n_tiles] (or null)
    void* qtiles;                               // q-side rep matrices as bf16 MFMA operand tiles, [B,Nq,GTA_QT_TILES][1 KiB] (gta_flash_common.h; or null)
    const float* vrep_q; const float* vrep_k;   // [B,N,GTA_VREP_STRIDE]
    const float* cs_q; const float* cs_k;       // [B,T,nso2,2] (cos,sin)
    const float* trans_coeff; const float* tau; // device scalars or null
    const float* kbias; long kbias_pitch;       // optional additive per-key bias (log2 units), [B,H,pitch]
    long q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st;  // element strides
    int B, H, Tq, Tk, Nq, Nk, Pq, Pk;           // P* = tokens per view
    float invPq, invPk;
    float inv_nqt, invH;                        // 1 / n_qtiles, 1 / H (gta_attn64_kernel's item decode)
    int dh, nso2, n_qtiles;
    int n_items;                                // work items of the attention kernel: B * H * n_qtiles
    int per_cu;                                 // persistent grid: workgroups resident per CU (0: one workgroup per item)
    int nrec;                                   // q-side view records staged per item (views a 128-row tile can touch)
    uint32_t flags;
    unsigned long long* prof;                   // debug: per-workgroup phase timestamps (or null)
    uint32_t dbg;                               // ablation bits (GTA_DBG env; 0 in production)
    float scale;
    uint32_t ctab[16];                          // chunk descriptors (gta_common.h)
};
