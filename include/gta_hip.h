/*
 * gta_hip.h -- C ABI of libgta_hip.so: MI355X (gfx950) kernels for GTA attention.
 *
 * The reference (autonomousvision/gta) has no FFI: its seam for this path is the Python
 * function ``multihead_geometric_transform_attention`` (source/utils/gta.py:92-279) called
 * from ``Attention.forward`` (source/layers.py:409-430) and the rep builders
 * ``pre_compute_reps`` (source/encoder.py:183-265, source/decoder.py:247-353).  The entry
 * points below are what a ctypes binding of those call sites needs.
 *
 * Conventions
 *   - plain C, raw DEVICE pointers, no torch types; the caller allocates every buffer, the
 *     library owns nothing and keeps no state; every call is asynchronous on `stream`
 *     (a hipStream_t passed as void*) and re-entrant.
 *   - return 0 on success, a negative GTA_E_* code otherwise; nothing throws, nothing is
 *     printed.  gta_strerror() names a code.
 *   - tokens are view-major (t = view * tokens_per_view + p), gta.py:160-162.
 *   - per-head channels are laid out [triv | se3 | so3 | so2 | t2], gta.py:115-122.
 */
#ifndef GTA_HIP_H
#define GTA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GTA_ABI_VERSION 2   /* r05: flag bit 12 is GTA_FLAG_ITEM_CXX, 8/16-byte alignment rules of the rep builders, dtype-dependent workspace size */

/* element types of q/k/v/out (all four share one type per call) */
#define GTA_DTYPE_F32  0
#define GTA_DTYPE_BF16 1

/* flags of GtaAttnDesc.flags */
#define GTA_FLAG_V_TRANSFORM   (1u << 0) /* gta.py: v_transform=True (default)                 */
#define GTA_FLAG_EUCLID        (1u << 1) /* gta.py:146-156 + layers.py:213-224 (not fused yet) */
#define GTA_FLAG_PRETRANSFORMED (1u << 2) /* q,k,v already carry rho; only rho_q^-1 on output   */
#define GTA_FLAG_FUSED_KV      (1u << 3) /* force the single-kernel path (rho_k inside the loop)  */
#define GTA_FLAG_PREP_ONLY     (1u << 5) /* two-stage plan: run only the K/V pre-pass (fills workspace) */
#define GTA_FLAG_KV_READY      (1u << 4) /* workspace already holds K'/V' for these k,v,reps:     */
                                         /* skip the pre-pass (several query sets, one key set)   */
#define GTA_FLAG_PERSIST       (1u << 9) /* tuning: persistent grid (resident workgroups walk the query tiles) instead of one workgroup per tile; */
                                         /* chosen automatically for launches of 1..2 rounds of resident workgroups (short sequences) */
#define GTA_FLAG_FP32_PRODUCTS (1u << 10) /* fp32 inputs: split-bf16 (hi+lo) operands, three MFMAs per product: fp32-class */
                                         /* results for the reference's mixed_prec: False configs, 3x the MFMAs; with a workspace   */
                                         /* at padded dh <= 64 the two-stage plan (hi and lo images: twice the image bytes; such a    */
                                         /* workspace is NOT what gta_attn_bwd's kv_images wants), the single-kernel plan otherwise   */
                                         /* (gta_attn_fwd_workspace_bytes then returns 0)                                              */
#define GTA_FLAG_ROWS32        (1u << 11) /* tuning: keep the 32-rows-per-wave attention kernel (gta_fwd2.hip) where the     */
                                          /* 64-rows-per-wave one (gta_fwd64.hip: dh = 96, whole ring turns of key tiles) would run */
#define GTA_FLAG_ITEM_CXX      (1u << 12) /* tuning / diagnostics: keep the 64-rows-per-wave kernel's compiler-scheduled item prologue and epilogue  */
                                          /* where the generated item stream (gen_item64.py) would run; same results bit for bit               */
#define GTA_FLAG_BWD_KEYS32    (1u << 13) /* tuning / diagnostics (gta_attn_bwd): keep the 32-keys-per-wave dK/dV kernel where the generated    */
                                          /* 64-per-wave streams (gen_bwd64.py: bf16, dh = 96, MSN layout, >= 128 blocks of 256 keys / rows) would run        */
#define GTA_FLAG_BWD_KEYS64    (1u << 14) /* tuning / diagnostics (gta_attn_bwd): run that stream at bf16, dh = 96 whatever the number of blocks    */
#define GTA_FLAG_BWD_SPLIT     (1u << 15) /* tuning / diagnostics (gta_attn_bwd): the dQ and dK/dV kernels as two launches (per-kernel times under a profiler)     */
                                          /* where one joint launch would run; same results bit for bit                                                         */
#define GTA_FLAG_FWD2_GENERIC  (1u << 6) /* tuning / diagnostics (r06): keep gta_fwd2_kernel where the dh = 64 instance of gta_fwd_cl.hip (gta_fwdc_kernel) would run */
#define GTA_FLAG_NO_DMA        (1u << 8) /* debug: stage K/V tiles through VGPRs, not LDS-DMA  */

/* error codes */
#define GTA_OK              0
#define GTA_E_BADARG       -1   /* null pointer / non-positive size / bad dtype                 */
#define GTA_E_LAYOUT       -2   /* f_dims do not add up to dh, or a slab is mis-sized           */
#define GTA_E_UNSUPPORTED  -3   /* valid request this build has no kernel for (says which in    */
                                /* gta_strerror); never silently falls back                     */
#define GTA_E_LAUNCH       -4   /* HIP launch error                                             */
#define GTA_E_NODEVICE     -5

/* Per-view rep record produced by gta_build_view_reps and consumed by the attention kernels.
 * One record of GTA_VREP_STRIDE floats per (batch, view):
 *   [ 0:16)  "inv" matrix  = extrinsic E           (reference: extras['inv_se3rep_q'], encoder.py:236)
 *   [16:32)  "rep" matrix  = inverse(E)            (reference: extras['se3rep_q/k'],  encoder.py:219,235)
 *   [32:41)  D^1(R), R = inverse(E)[:3,:3]         (reference: extras['so3rep_*'][0], encoder.py:247-259)
 *   [41:66)  D^2(R)                                (reference: extras['so3rep_*'][1])
 * all row-major, fp32.  trans_coeff is NOT folded in (it is a per-layer parameter). */
#define GTA_VREP_STRIDE 72
#define GTA_VREP_INV    0
#define GTA_VREP_REP    16
#define GTA_VREP_D1     32
#define GTA_VREP_D2     41

#define GTA_MAX_VIEWS   16   /* per side, fused kernels */

typedef struct GtaAttnDesc {
    int32_t abi_version;      /* GTA_ABI_VERSION                                              */
    int32_t dtype;            /* GTA_DTYPE_*                                                  */
    int32_t B, H;             /* batch, heads                                                 */
    int32_t Tq, Tk;           /* tokens (all views) on the query / key side                   */
    int32_t Nq, Nk;           /* views per side; Tq % Nq == 0, Tk % Nk == 0 (gta.py:160-162)  */
    int32_t dh;               /* channels per head = d_triv+d_se3+d_so3+d_so2+d_t2            */
    int32_t d_triv, d_se3, d_so3, d_so2, d_t2;   /* f_dims (gta.py:115-122)                  */
    int32_t so3_degree;       /* L: so3 sub-blocks of 3,5,..,2L+1 channels (gta.py:174-198)   */
    uint32_t flags;           /* GTA_FLAG_*                                                   */
    float   scale;            /* dim_head ** -0.5 (layers.py:181)                             */
    int32_t _pad;
    /* element strides of (batch, head, token); the channel stride is 1.  q/k/v may be views
     * into a packed [B, T, 3*H*dh] projection (layers.py:389,394-395) -- no copy needed.     */
    int64_t q_stride[3], k_stride[3], v_stride[3], o_stride[3];
} GtaAttnDesc;

/* -------------------------------------------------------------------------------------------
 * Rep builders  (replace encoder.py:183-265 / decoder.py:247-353, gta.py:47-69, wigner_d.py:16-58)
 * ------------------------------------------------------------------------------------------- */

/* extrinsics [n_views,4,4] fp32 (extras['input_transforms'] / ['target_transforms'], flattened
 * over batch) -> vrep [n_views, GTA_VREP_STRIDE].  so3_degree in {0,1,2}.  The Wigner-D path is
 * the reference's ZYZ-Euler formula D = Z(g3) J Z(g2) J Z(g1) incl. its 1e-5 gimbal masks
 * (wigner_d.py:28-49); J is a restatement of the absent J_dense.pt (parity unpinned there). */
int gta_build_view_reps(const float* extrinsics, int32_t n_views, int32_t so3_degree,
                        float* vrep, void* stream);

/* coord [n_tokens,2] fp32 in [0,1) (extras['input_coord'] / ['target_coord'] flattened) ->
 * cs [n_tokens, 2*nfreqs, 2] = (cos, sin) of theta_{t, c=2f+d} = max_freq_d * 2pi * coord_d *
 * 2^(f+1-F)  (make_SO2mats, gta.py:47-69; block order c = 2f+d from gta.py:68 + encoder.py:195).
 * Alignment: coord 8 bytes, cs 16 bytes (the kernel moves a token's coordinate pair and a frequency's two (cos, sin) pairs in one
 * access each); anything else returns GTA_E_BADARG. */
int gta_build_so2_table(const float* coord, int32_t n_tokens, int32_t nfreqs,
                        float max_freq_h, float max_freq_w, int32_t shared_freqs,
                        float* cs, void* stream);

/* Both of the above in ONE launch (they are independent and launch-latency sized): what
 * pre_compute_reps (source/encoder.py:183-265) does for a gta_so3-style config in one call.
 * Arguments (and alignment requirements) as for gta_build_view_reps followed by those of gta_build_so2_table. */
int gta_build_reps(const float* extrinsics, int32_t n_views, int32_t so3_degree, float* vrep,
                   const float* coord, int32_t n_tokens, int32_t nfreqs, float max_freq_h,
                   float max_freq_w, int32_t shared_freqs, float* cs, void* stream);

/* -------------------------------------------------------------------------------------------
 * Fused forward  (replaces gta.py:92-279 + AttnFn, layers.py:202-211)
 *   q [B,H,Tq,dh], k,v [B,H,Tk,dh] through the strides in desc; out like q.
 *   vrep_q [B,Nq,GTA_VREP_STRIDE], vrep_k [B,Nk,...] (may be NULL when d_se3 = d_so3 = 0)
 *   cs_q [B,Tq,d_so2/2,2], cs_k [B,Tk,d_so2/2,2]      (may be NULL when d_so2 = 0)
 *   trans_coeff: device pointer to the layer's scalar parameter (layers.py:191) or NULL (=1)
 *   tau:         device pointer to the softmax temperature (layers.py:195-200) or NULL (=1)
 *   lse [B,H,Tq] fp32 out (natural-log sum-exp of the scaled logits; needed by backward), or NULL
 * ------------------------------------------------------------------------------------------- */
int gta_attn_fwd(const GtaAttnDesc* desc,
                 const void* q, const void* k, const void* v,
                 const float* vrep_q, const float* vrep_k,
                 const float* cs_q, const float* cs_k,
                 const float* trans_coeff, const float* tau,
                 void* out, float* lse,
                 void* workspace, int64_t workspace_bytes, void* stream);

/* Two execution plans, same results:
 *   workspace == NULL (or GTA_FLAG_FUSED_KV): ONE kernel; rho_k is applied to each K/V tile inside
 *       the attention loop (re-done once per 128-row query tile).
 *   workspace of >= gta_attn_fwd_workspace_bytes(desc) bytes: a K/V pre-pass writes K' = rho_k K,
 *       V' = rho_k V once as bf16 tile images into the workspace, then a lean attention kernel
 *       streams them.  The workspace content stays valid for other query sets against the same
 *       keys (GTA_FLAG_KV_READY) and is what the backward consumes.
 *   Size: [K'/V' images | per-tile key norms] depend on (B, H, Tk, dh) only.  For bf16 inputs at dh in (64, 96] the workspace ends
 *       with B * Nq * 24 KiB of query-side operand tiles (rewritten by every call), so the size ALSO depends on the query side's
 *       number of views: a workspace kept for GTA_FLAG_KV_READY calls must be sized for the largest Nq it will see (query the size
 *       with that Nq; a too-small buffer returns GTA_E_BADARG).  The size depends on desc->flags: GTA_FLAG_FP32_PRODUCTS doubles the
 *       images where that mode has a two-stage plan (fp32 inputs, padded dh <= 64) and makes the size 0 where it has none. */
int64_t gta_attn_fwd_workspace_bytes(const GtaAttnDesc* desc);

/* -------------------------------------------------------------------------------------------
 * Backward  (replaces PyTorch autograd over gta.py:92-279 + layers.py:202-211)
 *   inputs : q, k, v (as given to the forward; strides from desc), out and lse from the forward,
 *            dout [B,H,Tq,dh] with element strides dout_stride[3] (b,h,t), reps as in the forward.
 *   kv_images: the forward's workspace (K'/V' tile images) or NULL -> recomputed here.
 *   plan     : Q''/dO~ pre-pass -> dQ kernel -> dK/dV kernel, each recomputing S and dP from the images (14 GEMM-units).
 *              Deterministic (no atomics: fixed-order reductions).
 *   outputs: dq, dk, dv with element strides dqkv_stride[9] = dq(b,h,t), dk(b,h,t), dv(b,h,t);
 *            dtrans_coeff [1] fp32 (d loss / d trans_coeff, layers.py:191; written, not accumulated) or NULL;
 *            dtau [1] fp32 (d loss / d tau of TemperatureAdjsutableSoftmax, layers.py:135-143,195-200) or
 *            NULL.  The logits are z = scale q'.k' / tau, so dL/dtau = -(1/tau) sum_ij dz_ij z_ij
 *            = -(1/tau) sum_i <q'_i, dq'_i> = -(1/tau) sum_i <q_i, dq_i>  (rho_q is linear): it falls out of
 *            the dQ kernel's epilogue, no extra pass over the tiles.
 *   No gradient is produced for the reps/poses (gta.py:194-198 detaches them; poses are data).
 *   GTA_FLAG_FP32_PRODUCTS (r06; fp32 inputs, dh <= 64; other sizes: GTA_E_UNSUPPORTED -> gta_rep_apply + gta_attn_bwd_plain_f32): the fp32-faithful
 *            backward on the matrix cores -- every product of the five contractions as three bf16 MFMAs over hi / lo operand images (the
 *            arithmetic of the forward under the same flag); kv_images must then be the workspace of a forward run WITH the flag (four
 *            images per tile) or NULL; the workspace is twice as large (gta_attn_bwd_workspace_bytes depends on desc->flags); dtau is formed
 *            in the dQ walk as sum_ij dS_ij (S_ij - m_i), m_i = sum_j P_ij S_ij (centred per row: DESIGN.md section 4.6).
 * ------------------------------------------------------------------------------------------- */
int64_t gta_attn_bwd_workspace_bytes(const GtaAttnDesc* desc);
int gta_attn_bwd(const GtaAttnDesc* desc,
                 const void* q, const void* k, const void* v, const void* out, const void* dout,
                 const float* lse,
                 const float* vrep_q, const float* vrep_k, const float* cs_q, const float* cs_k,
                 const float* trans_coeff, const float* tau,
                 const void* kv_images,
                 void* dq, void* dk, void* dv, const int64_t* dqkv_stride, const int64_t* dout_stride,
                 float* dtrans_coeff, float* dtau,
                 void* workspace, int64_t workspace_bytes, void* stream);

/* -------------------------------------------------------------------------------------------
 * Generic path for the reference's ablations (any f_dims layout): t2 slab (gta.py:221-238,272-274),
 * euclid similarity (GTA_FLAG_EUCLID; gta.py:146-156,251-253 + EuclidAttnFn layers.py:213-224),
 * so3 of degree 1, unaligned slabs.  Usage: q' = apply(mode 0), k' = apply(mode 1)
 * (+ key_bias under euclid), v' = apply(mode 1), o~ = gta_attn_fwd_plain(q',k',v',key_bias),
 * o = apply(mode 2).
 *   x, y: [B,H,T,dh] through x_stride/y_stride (b,h,t), dtype from desc; T,N = Tq,Nq (modes 0,2) or Tk,Nk.
 *   coord [B,T,2]: the token coordinates of the t2 slab (make_T2mats, gta.py:72-89) or NULL.
 *   key_bias (mode 1, or NULL): [B,H,bias_pitch] <- -0.5 * bias_scale * |y|^2 per token.
 * ------------------------------------------------------------------------------------------- */
int gta_rep_apply(const GtaAttnDesc* desc, int32_t mode, const void* x, const int64_t* x_stride,
                  const float* vrep, const float* cs, const float* coord, const float* trans_coeff,
                  void* y, const int64_t* y_stride, float* key_bias, float bias_scale, int64_t bias_pitch,
                  void* stream);

/* Adjoint of gta_rep_apply(mode): dx = M^T dy (what autograd produces for gta.py:134-238 on the generic path).
 *   x: the tensor gta_rep_apply(mode) was given; dy, dx: [B,H,T,dh] through their strides.
 *   dtc_rows [B,H,T] fp32 or NULL: this row's contribution to d loss / d trans_coeff (sum them in a fixed order).
 *   dkey_bias [B,H,bias_pitch] or NULL (mode 1 under euclid): gradient w.r.t. the key bias the forward call wrote;
 *   folded in as dy - bias_scale * dkey_bias * y.  Rows whose slabs carry no bias term (so3/so2/t2) ignore it. */
int gta_rep_apply_bwd(const GtaAttnDesc* desc, int32_t mode, const void* x, const int64_t* x_stride,
                      const void* dy, const int64_t* dy_stride, const float* vrep, const float* cs,
                      const float* coord, const float* trans_coeff, const float* dkey_bias, float bias_scale,
                      int64_t bias_pitch, void* dx, const int64_t* dx_stride, float* dtc_rows, void* stream);

/* softmax((scale * q k^T + key_bias) / tau) v with no rep at all (the fused kernel on an identity layout).
 * key_bias: [B,H,bias_pitch] fp32, added to scale*q.k BEFORE the division by tau; bias_pitch a multiple of 64 >= Tk; or NULL.
 * Only desc->{dtype,B,H,Tq,Tk,dh,scale,*_stride} are read. */
int gta_attn_fwd_plain(const GtaAttnDesc* desc, const void* q, const void* k, const void* v,
                       const float* key_bias, int64_t bias_pitch, const float* tau,
                       void* out, float* lse, void* stream);

/* -------------------------------------------------------------------------------------------
 * fp32-FAITHFUL backward of plain attention (identity layout, fp32 tensors): the gradient leg of the accuracy mode for the
 * reference's `mixed_prec: False` configs (runs/clevrtr/GTA/gta/config.yaml:55).  q', k', v' (pre-transformed by gta_rep_apply),
 * o~ and lse from gta_attn_fwd_plain (GTA_FLAG_FP32_PRODUCTS), do~ from gta_rep_apply_bwd(mode 2) -> dq', dk', dv' for
 * gta_rep_apply_bwd(modes 0 / 1).  Exact fp32 products and accumulation (v_mfma_f32_32x32x2_f32 = an fmaf chain), 1/16 of the bf16
 * matrix rate: accuracy, not speed.  desc: dtype F32, dh % 8 == 0 (<= 128), strides of q / k / v / out in elements (multiples of
 * 4); dout_stride[3], dqkv_stride[9] as in gta_attn_bwd; workspace >= gta_attn_bwd_plain_f32_workspace_bytes (B H Tq floats).
 * ------------------------------------------------------------------------------------------- */
int64_t gta_attn_bwd_plain_f32_workspace_bytes(const GtaAttnDesc* desc);
int gta_attn_bwd_plain_f32(const GtaAttnDesc* desc, const void* q, const void* k, const void* v, const void* out,
                           const void* dout, const int64_t* dout_stride, const float* lse, const float* tau,
                           void* dq, void* dk, void* dv, const int64_t* dqkv_stride, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* 0 when gta_attn_fwd has a fused kernel for this desc, else the error it would return. */
int gta_attn_fwd_supported(const GtaAttnDesc* desc);

/* bytes of LDS / number of workgroups the fused forward uses for desc (diagnostics). */
int gta_attn_fwd_launch_info(const GtaAttnDesc* desc, int32_t* lds_bytes, int32_t* n_workgroups,
                             int32_t* threads_per_wg);

/* Profiling hooks (diagnostics; bench.py and tools/ use them, no product path does):
 *   gta_debug_time_next_attention_kernel: the next attention-kernel launch made by THIS thread through gta_attn_fwd is
 *     bracketed by the two events through the dispatch packet itself (hipExtLaunchKernelGGL), i.e. without the marker
 *     packets of hipEventRecord that push neighbouring kernels apart.  One-shot.
 *   gta_debug_event_*: thin wrappers so a ctypes caller needs no second HIP binding.
 *   gta_debug_profile_next_attention_kernel: device buffer [capacity_items][8] of uint64 (or NULL): the NEXT attention kernel launched
 *       by THIS thread through gta_attn_fwd writes, per work item, [0] / [4] = s_memtime at its start / end (shader cycles) and
 *       [5] / [6] = s_memrealtime (100 MHz) -- kernel cycles and the granted shader clock of a launch follow from them; gta_fwd2_kernel
 *       also writes [1] = HW_ID | XCC_ID << 32 of the item's first wave (which CU it ran on: tools/wg_timeline.py); -DGTA_ABLATE
 *       builds write per-phase stamps to [1], [2], [3], [7] instead.  One-shot and thread-local; a launch with more work items than capacity_items writes nothing.
 *   gta_debug_attention_kernel: name of the attention kernel a workspace call of gta_attn_fwd launches for desc, its number of
 *       work items and query rows per item ("" if desc is not supported). */
void gta_debug_time_next_attention_kernel(void* start_event, void* stop_event);
void* gta_debug_event_create(void);
void gta_debug_event_destroy(void* event);
float gta_debug_event_elapsed_ms(void* start_event, void* stop_event);
void gta_debug_profile_next_attention_kernel(void* device_buffer, int64_t capacity_items);
const char* gta_debug_attention_kernel(const GtaAttnDesc* desc, int32_t* n_items, int32_t* rows_per_item);

const char* gta_strerror(int code);
int gta_abi_version(void);
int gta_sizeof_attn_desc(void);   /* sizeof(GtaAttnDesc) as compiled: binding self-check */

#ifdef __cplusplus
}
#endif
#endif /* GTA_HIP_H */
