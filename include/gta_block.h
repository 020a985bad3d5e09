/*
 * gta_block.h -- C ABI of libgta_block.so: the pre-LN Transformer block AROUND the GTA attention operator on MI355X
 * (SURVEY.md section 8, row f1).
 *
 * What it replaces in the reference (autonomousvision/gta), per layer:
 *   PreNorm                     source/layers.py:146-154   nn.LayerNorm -> fn
 *   Attention projections       source/layers.py:388-395   to_qkv / to_q, to_kv
 *   Attention output            source/layers.py:429-430   to_out (Linear + bias), then `+ x` at :483-486
 *   FeedForward                 source/layers.py:157-169   Linear, GELU, Linear, then `+ x` at :487
 * i.e. everything of `Transformer.forward` (layers.py:475-488) except the attention operator itself, which stays in
 * libgta_hip.so (gta_hip.h).  Launches per layer, forward: LayerNorm(+cast) | QKV GEMM | [attention] | out-proj GEMM with
 * bias + residual epilogue | LayerNorm(+cast) | GEMM with bias + GELU epilogue | GEMM with bias + residual epilogue.
 *
 * The GEMMs are hipBLASLt calls (plain library GEMMs with library epilogues); the row kernels are hand-written HIP.
 *
 * Conventions: as gta_hip.h (plain C, raw device pointers, caller-allocated buffers, asynchronous on `stream`, 0 or a
 * negative GTA_E_* code).  One exception to "no state": the hipBLASLt handle and the per-shape algorithm choice are
 * cached per host thread inside the library (creating them costs milliseconds); gta_block_release() drops them.
 * All matrices are ROW-major with the given leading dimension (elements).
 */
#ifndef GTA_BLOCK_H
#define GTA_BLOCK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GTA_BLOCK_ABI_VERSION 1

/* element types (same values as gta_hip.h) */
#ifndef GTA_DTYPE_F32
#define GTA_DTYPE_F32  0
#define GTA_DTYPE_BF16 1
#endif

/* ---------------------------------------------------------------------------------------------------------------
 * LayerNorm over the last dimension (nn.LayerNorm(dim), layers.py:149; elementwise affine, biased variance).
 *   x [rows, d] (x_dtype) -> y [rows, d] (y_dtype) = (x - mean) * rstd * gamma + beta, statistics and arithmetic in
 *   fp32; mean, rstd [rows] fp32 out (needed by the backward) or NULL.  d % 8 == 0 and d <= 4096, or d % 4 == 0 and d <= 2048
 *   (8-byte row accesses: the MSN decoder's d = 180).
 *   Writing y as bf16 is the "cast for the GEMM" that autocast does in a separate kernel.
 * --------------------------------------------------------------------------------------------------------------- */
int gta_ln_fwd(const void* x, int32_t x_dtype, const float* gamma, const float* beta, float eps,
               int64_t rows, int32_t d, void* y, int32_t y_dtype, float* mean, float* rstd, void* stream);

/* Backward of the above, fused with the residual branch:
 *   dx = (dres ? dres : 0) + rstd * (g - mean_d(g) - xhat * mean_d(g * xhat)),  g = dy * gamma, xhat = (x - mean) * rstd
 *   dgamma[d] = sum_rows dy * xhat, dbeta[d] = sum_rows dy   (fixed summation order: deterministic)
 *   dres: gradient arriving through the skip connection (`fn(norm(x)) + x`), dtype dx_dtype, or NULL; may alias dx.
 *   dx_bf16: optional second copy of dx in bf16 (NULL = none): the block upstream feeds its GEMMs from it, which saves
 *   the separate fp32 -> bf16 cast kernel autocast would run on the residual-stream gradient.
 *   workspace: gta_ln_bwd_workspace_bytes(rows, d) bytes. */
int64_t gta_ln_bwd_workspace_bytes(int64_t rows, int32_t d);
int gta_ln_bwd(const void* dy, int32_t dy_dtype, const void* x, int32_t x_dtype, const float* gamma,
               const float* mean, const float* rstd, int64_t rows, int32_t d,
               const void* dres, void* dx, int32_t dx_dtype, void* dx_bf16, float* dgamma, float* dbeta,
               void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * GELU, exact erf form (nn.GELU(), layers.py:162) -- hipBLASLt's epilogue is the tanh form, a parity break in fp32 and
 * not available with the pre-activation output training needs -- fused with the Dropout that follows it in
 * FeedForward.net (layers.py:163):  y = keep * gelu(x) / (1 - p);  dx = dy * keep / (1 - p) * gelu'(x).
 * p = 0: plain GELU.  n elements, n % 8 == 0.
 *
 * Dropout masks are never stored: keep(i) is a pure function of (seed, element index) -- Philox4x32-10 keyed by `seed`,
 * counter = i / 4, keep <=> word >= p * 2^32 -- so the backward regenerates the forward's mask from the same seed.
 * --------------------------------------------------------------------------------------------------------------- */
int gta_gelu_fwd(const void* x, void* y, int32_t dtype, int64_t n, float p, uint64_t seed, void* stream);
int gta_gelu_bwd(const void* dy, const void* x, void* dx, int32_t dtype, int64_t n, float p, uint64_t seed, void* stream);

/* Dropout + skip connection, forward (to_out's / net's trailing nn.Dropout and the `+ x`, layers.py:165,289,483-487):
 *   out = skip + keep * z / (1 - p);  z [n] (z_dtype): the GEMM output incl. bias; skip, out [n] (skip_dtype).
 * Backward of the dropout:  dz = keep * dout / (1 - p), written in the GEMMs' dtype. */
int gta_dropout_add(const void* z, int32_t z_dtype, const void* skip, void* out, int32_t skip_dtype, int64_t n,
                    float p, uint64_t seed, void* stream);
int gta_dropout_bwd(const void* dout, int32_t dout_dtype, void* dz, int32_t dz_dtype, int64_t n, float p, uint64_t seed,
                    void* stream);

/* Column sums: out[n] (fp32) = sum over the m rows of a [m, n] (ld elements between rows); the bias gradient of a
 * Linear.  Deterministic.  workspace: gta_colsum_workspace_bytes(m, n).  n % 4 == 0, ld % 4 == 0. */
int64_t gta_colsum_workspace_bytes(int64_t m, int32_t n);
int gta_colsum(const void* a, int32_t dtype, int64_t m, int32_t n, int64_t ld, float* out,
               void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * GEMM with epilogue (hipBLASLt):   D[m,n] = epilogue( alpha * op_a(A) . op_b(B) + beta * C )
 *   op_a(A) is [m,k]: A is stored [m,k] (trans_a = 0) or [k,m] (trans_a = 1), row-major, leading dimension lda;
 *   op_b(B) is [k,n]: B is stored [k,n] (trans_b = 0) or [n,k] (trans_b = 1).  nn.Linear's y = x W^T is trans_b = 1.
 *   C (c_dtype == d_dtype) may be NULL when beta == 0; C == D is allowed (in place).
 * Epilogues (what they replace):
 *   GTA_EPI_NONE            plain GEMM                                  to_qkv / to_q / to_kv (bias=False), dgrad, wgrad
 *   GTA_EPI_BIAS            + bias[n]; with beta = 1, C = residual      to_out + `+ x` (layers.py:430,483-486), net[3] + `+ x` (:487)
 *   GTA_EPI_BIAS_GELU       gelu(. + bias[n])           (tanh form)     net[0] + nn.GELU (layers.py:161-162), inference
 *   GTA_EPI_BIAS_GELU_AUX   same, and aux[m,n] <- (. + bias) before the GELU (for the backward)
 *   GTA_EPI_DGELU           D = (.) * gelu'(aux[m,n])                   d/d pre-activation in the dgrad GEMM of net[3]
 *   GTA_EPI_DGELU_BGRAD     same, and bias[n] <- column sums of D       ... + d bias of net[0]
 *   GTA_EPI_BGRAD_A         bias[m] <- sum over k of op_a(A)            d bias of a Linear inside its wgrad GEMM
 * bias: bias_dtype (F32 or the d_dtype); aux: aux_dtype, leading dimension ldaux.
 * The first call of a problem (per host thread) times hipBLASLt's candidate kernels on the caller's own buffers and keeps
 * the fastest: that call runs the GEMM several times and waits on `stream` (environment GTA_GEMM_TUNE=0 turns it off,
 * =2 searches every kernel of the library).  Not done while `stream` is being captured into a graph.
 * --------------------------------------------------------------------------------------------------------------- */
#define GTA_EPI_NONE          0
#define GTA_EPI_BIAS          1
#define GTA_EPI_BIAS_GELU     2
#define GTA_EPI_BIAS_GELU_AUX 3
#define GTA_EPI_DGELU         4
#define GTA_EPI_DGELU_BGRAD   5
#define GTA_EPI_BGRAD_A       6

typedef struct GtaGemmDesc {
    int32_t abi_version;            /* GTA_BLOCK_ABI_VERSION */
    int32_t epilogue;               /* GTA_EPI_* */
    int64_t m, n, k;
    int32_t trans_a, trans_b;
    int32_t a_dtype, b_dtype, d_dtype;   /* C has d_dtype */
    int32_t bias_dtype, aux_dtype;
    int32_t _pad;
    int64_t lda, ldb, ldc, ldd, ldaux;
    float alpha, beta;
} GtaGemmDesc;

int64_t gta_gemm_workspace_bytes(void);      /* what gta_gemm wants for hipBLASLt (a fixed 32 MiB) */
int gta_gemm(const GtaGemmDesc* desc, const void* a, const void* b, const void* c, void* d,
             void* bias, void* aux, void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Weight gradient of a Linear, hand-written (gta_wgrad.hip):  dW[n,k] = sum_m G[m,n] X[m,k],  db[n] = sum_m G[m,n]
 *   G = d(out) [m,n] bf16 (leading dimension ldg), X = the layer's input [m,k] bf16 (ldx); dW [n,k] fp32 contiguous,
 *   dbias [n] fp32 or NULL.  The reduction index m (batch x tokens) is the slow index of both operands -- the shape
 *   hipBLASLt is weakest at.  Split over m across the chip, partial tiles reduced in a fixed order (deterministic).
 *   Supported when m % 32 == 0, n % 256 == 0, k % 256 == 0 (gta_wgrad_supported); else use gta_gemm(trans_a) + gta_colsum.
 *   workspace: gta_wgrad_workspace_bytes(m, n, k).
 * --------------------------------------------------------------------------------------------------------------- */
int gta_wgrad_supported(int64_t m, int64_t n, int64_t k);
int64_t gta_wgrad_workspace_bytes(int64_t m, int64_t n, int64_t k);
int gta_wgrad(const void* g, int64_t ldg, const void* x, int64_t ldx, int64_t m, int64_t n, int64_t k,
              float* dw, float* dbias, void* workspace, int64_t workspace_bytes, void* stream);

/* drops this thread's hipBLASLt handle and algorithm cache */
void gta_block_release(void);

const char* gta_block_strerror(int code);    /* names a code; for GTA_E_LAUNCH also the last hipBLASLt status */
int gta_block_abi_version(void);
int gta_sizeof_gemm_desc(void);

#ifdef __cplusplus
}
#endif
#endif /* GTA_BLOCK_H */
