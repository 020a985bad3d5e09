// gta_bwd_params.h -- kernel argument block of the backward kernels (filled in gta_abi.cpp).
#pragma once
#include <stdint.h>

struct GtaBwdParams {
    const void *q, *k, *v, *out, *dout;
    const float* lse;
    void *dq, *dk, *dv;
    const float *vrep_q, *vrep_k, *cs_q, *cs_k, *trans_coeff, *tau;
    const void* kvimg;        // K'/V' tile images (forward workspace layout)
    void* qimg;               // Q''/dO~ tile images   [B,H,n_qt64][2 images]
    float* stats;             // [B,H,n_qt64][128] = lse*log2e | D
    float* dc_partial;        // per-workgroup d trans_coeff partial sums
    float* dtrans_coeff;      // [1] or null
    float* dt_partial;        // per-dQ-workgroup sum of <q, dq> (d tau), or null
    float* dtau;              // [1] or null
    long q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st, do_sb, do_sh, do_st;
    long dq_sb, dq_sh, dq_st, dk_sb, dk_sh, dk_st, dv_sb, dv_sh, dv_st;
    int dc_off_prep, dc_off_dq, dc_off_dkv, dc_total;
    int B, H, Tq, Tk, Nq, Nk, Pq, Pk;
    float invPq, invPk;
    int dh, nso2;
    uint32_t flags;
    float scale;
    uint32_t ctab[16];
};
