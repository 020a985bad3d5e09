// gta_abi.cpp -- extern "C" boundary of libgta_hip.so: argument validation, the chunk table that
// encodes the reference's slab layout (gta.py:115-122), kernel dispatch.  No state, no allocation.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdlib.h>

#include "../../include/gta_hip.h"
#include "gta_fwd_params.h"
#include "gta_bwd_params.h"

// chunk-descriptor constants (mirrors gta_common.h, which is device-only)
#define HALF_ID 0u
#define HALF_SE3 1u
#define HALF_SO2 2u
#define CHUNK_SO3 (1u << 4)

int gta_fwd_lds_bytes(int dhp, int esz);
int gta_fwd_dispatch(const GtaFwdParams& p, int dhp, int esz, bool dma, int n_wg, hipStream_t stream);
long gta_fwd2_workspace_bytes(int B, int H, int Tk, int dhp, int Nq, int esz, bool x3);
long gta_fwd2_qtiles_offset(int B, int H, int Tk, int dhp, bool x3);
bool gta_fwd2_x3_takes(int dhp, int esz);                                              // fp32-faithful products on the two-stage plan
int gta_fwd2_rows_per_item(const GtaFwdParams& p, int dhp, int esz);                   // 256: gta_attn64_kernel, 128: gta_fwd2_kernel
const char* gta_fwd2_attention_kernel_name(const GtaFwdParams& p, int dhp, int esz);
long gta_fwd2_image_bytes(int B, int H, int Tk, int dhp, bool x3);
int gta_fwd2_lds_bytes(int dhp, int nq);
int gta_fwd2_dispatch(GtaFwdParams& p, int dhp, int esz, bool run_prep, bool run_flash, hipStream_t stream);
int gta_bwd_dispatch(const GtaBwdParams& p, int dhp, int esz, hipStream_t stream);

namespace {

thread_local const char* g_detail = "";

int fail(int code, const char* why) { g_detail = why; return code; }

int padded_dh(int dh) { return dh <= 32 ? 32 : dh <= 64 ? 64 : dh <= 96 ? 96 : dh <= 128 ? 128 : -1; }

// Build the per-chunk descriptors for the fused kernels, or say why this layout needs the
// generic (unfused) path.
int build_ctab(const GtaAttnDesc* d, uint32_t* ctab) {
    if (d->d_triv < 0 || d->d_se3 < 0 || d->d_so3 < 0 || d->d_so2 < 0 || d->d_t2 < 0)
        return fail(GTA_E_LAYOUT, "negative slab size");
    if (d->d_triv + d->d_se3 + d->d_so3 + d->d_so2 + d->d_t2 != d->dh)
        return fail(GTA_E_LAYOUT, "f_dims do not sum to dh");
    if (d->flags & GTA_FLAG_EUCLID) {
        if (d->d_se3 % 3) return fail(GTA_E_LAYOUT, "under euclid_sim the se3 slab holds 3-vectors (gta.py:147)");
        return fail(GTA_E_UNSUPPORTED, "euclid similarity has no fused kernel (gta_rep_apply + gta_attn_fwd_plain)");
    }
    if (d->d_se3 % 4) return fail(GTA_E_LAYOUT, "se3 slab must be a multiple of 4 channels (gta.py:161)");
    if (d->d_so2 % 4) return fail(GTA_E_LAYOUT, "so2 slab must be 4*nfreqs channels (gta.py:212-214)");
    if (d->d_t2 % 3) return fail(GTA_E_LAYOUT, "t2 slab must be a multiple of 3 channels (gta.py:231)");
    if (d->d_so3 > 0) {
        int tot = 0;
        for (int l = 1; l <= d->so3_degree; ++l) tot += 2 * l + 1;
        if (d->so3_degree < 1 || d->d_so3 % tot) return fail(GTA_E_LAYOUT, "so3 slab must be r*sum(2l+1) channels (gta.py:182)");
    }
    if (d->dh % 8 || d->dh > 128) return fail(GTA_E_UNSUPPORTED, "fused kernel needs dh % 8 == 0 and dh <= 128");
    if (d->d_t2 > 0) return fail(GTA_E_UNSUPPORTED, "t2 slab has no fused kernel (ablation; use the unfused path)");
    if (d->flags & GTA_FLAG_EUCLID) return fail(GTA_E_UNSUPPORTED, "euclid similarity has no fused kernel");
    if (d->d_so3 > 0 && (d->so3_degree != 2 || d->d_so3 % 8))
        return fail(GTA_E_UNSUPPORTED, "fused so3 needs degree 2 ([3|5] groups of 8 channels)");
    const int s_se3 = d->d_triv, s_so3 = s_se3 + d->d_se3, s_so2 = s_so3 + d->d_so3;
    if (s_se3 % 4 || (d->d_so3 > 0 && s_so3 % 8) || s_so2 % 4)
        return fail(GTA_E_UNSUPPORTED, "fused kernel needs 4-aligned se3/so2 slabs and an 8-aligned so3 slab");
    for (int c = 0; c < 16; ++c) ctab[c] = 0;
    for (int c = 0; c < d->dh / 8; ++c) {
        uint32_t desc = 0;
        const int lo = 8 * c, hi = 8 * c + 4;
        if (d->d_so3 > 0 && lo >= s_so3 && lo < s_so2) { ctab[c] = CHUNK_SO3; continue; }
        for (int half = 0; half < 2; ++half) {
            const int ch = half ? hi : lo;
            uint32_t kind = HALF_ID, blk = 0;
            if (ch >= s_se3 && ch < s_so3) kind = HALF_SE3;
            else if (ch >= s_so2 && ch < s_so2 + d->d_so2) { kind = HALF_SO2; blk = (uint32_t)(ch - s_so2) / 2; }
            desc |= kind << (2 * half);
            desc |= blk << (8 + 8 * half);
        }
        ctab[c] = desc;
    }
    return GTA_OK;
}

int check_common(const GtaAttnDesc* d) {
    if (!d) return fail(GTA_E_BADARG, "null descriptor");
    if (d->abi_version != GTA_ABI_VERSION) return fail(GTA_E_BADARG, "abi_version mismatch");
    if (d->dtype != GTA_DTYPE_F32 && d->dtype != GTA_DTYPE_BF16) return fail(GTA_E_BADARG, "bad dtype");
    if (d->B <= 0 || d->H <= 0 || d->Tq <= 0 || d->Tk <= 0 || d->dh <= 0) return fail(GTA_E_BADARG, "non-positive size");
    if (d->Nq <= 0 || d->Nk <= 0 || d->Tq % d->Nq || d->Tk % d->Nk)
        return fail(GTA_E_BADARG, "tokens must split evenly into views (gta.py:160-162)");
    if (d->Nq > GTA_MAX_VIEWS || d->Nk > GTA_MAX_VIEWS) return fail(GTA_E_UNSUPPORTED, "more than GTA_MAX_VIEWS views per side");
    if (d->Tq >= (1 << 22) || d->Tk >= (1 << 22)) return fail(GTA_E_UNSUPPORTED, "more than 2^22 tokens per side");
    const int esz = d->dtype == GTA_DTYPE_BF16 ? 2 : 4;
    const int64_t* st[4] = {d->q_stride, d->k_stride, d->v_stride, d->o_stride};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j)
            if ((st[i][j] * esz) % 16) return fail(GTA_E_BADARG, "strides must keep every head row 16-byte aligned");
    return GTA_OK;
}

}  // namespace

// debug hook (not part of the product ABI): device buffer [capacity_items][8] for the per-item s_memtime stamps of the NEXT attention
// launch this thread makes through gta_attn_fwd -- thread-local and one-shot, like gta_debug_time_next_attention_kernel, and dropped
// (no stamps) when the launch has more work items than the buffer holds
static thread_local unsigned long long* t_prof = nullptr;
static thread_local int64_t t_prof_items = 0;
extern "C" void gta_debug_profile_next_attention_kernel(void* p, int64_t capacity_items) {
    t_prof = (unsigned long long*)p;
    t_prof_items = p ? capacity_items : 0;
}
extern "C" int gta_abi_version(void) { return GTA_ABI_VERSION; }
extern "C" int gta_sizeof_attn_desc(void) { return (int)sizeof(GtaAttnDesc); }

extern "C" const char* gta_strerror(int code) {
    static thread_local char buf[256];
    const char* base = "unknown";
    switch (code) {
        case GTA_OK: return "ok";
        case GTA_E_BADARG: base = "bad argument"; break;
        case GTA_E_LAYOUT: base = "inconsistent f_dims layout"; break;
        case GTA_E_UNSUPPORTED: base = "unsupported by this build"; break;
        case GTA_E_LAUNCH: base = "HIP launch failed"; break;
        case GTA_E_NODEVICE: base = "no HIP device"; break;
    }
    snprintf(buf, sizeof buf, "%s: %s", base, g_detail);
    return buf;
}

extern "C" int gta_attn_fwd_supported(const GtaAttnDesc* desc) {
    if (!desc) return fail(GTA_E_BADARG, "null descriptor");
    if (desc->abi_version != GTA_ABI_VERSION) return fail(GTA_E_BADARG, "abi_version mismatch");
    uint32_t ctab[16];
    int rc = build_ctab(desc, ctab);      // layout first: "no fused kernel" must not be masked by a stride complaint
    if (rc) return rc;
    return check_common(desc);
}

// does this call run the two-stage plan when it is given a workspace?  (GTA_FLAG_FP32_PRODUCTS: where the split-bf16 instances exist)
static bool two_stage_plan(const GtaAttnDesc* d) {
    if (d->flags & (GTA_FLAG_FUSED_KV | GTA_FLAG_PRETRANSFORMED)) return false;
    if (d->flags & GTA_FLAG_FP32_PRODUCTS) return gta_fwd2_x3_takes(padded_dh(d->dh), d->dtype == GTA_DTYPE_BF16 ? 2 : 4);
    return true;
}
extern "C" int64_t gta_attn_fwd_workspace_bytes(const GtaAttnDesc* desc) {
    if (gta_attn_fwd_supported(desc)) return 0;
    if ((desc->flags & GTA_FLAG_FP32_PRODUCTS) && !two_stage_plan(desc)) return 0;       // (that mode runs the single-kernel plan here)
    return gta_fwd2_workspace_bytes(desc->B, desc->H, desc->Tk, padded_dh(desc->dh), desc->Nq, desc->dtype == GTA_DTYPE_BF16 ? 2 : 4,
                                    (desc->flags & GTA_FLAG_FP32_PRODUCTS) != 0);
}

extern "C" int gta_attn_fwd_launch_info(const GtaAttnDesc* desc, int32_t* lds_bytes, int32_t* n_workgroups,
                                        int32_t* threads_per_wg) {
    int rc = gta_attn_fwd_supported(desc);
    if (rc) return rc;
    const int esz = desc->dtype == GTA_DTYPE_BF16 ? 2 : 4;
    if (lds_bytes) *lds_bytes = gta_fwd_lds_bytes(padded_dh(desc->dh), esz);
    if (n_workgroups) *n_workgroups = desc->B * desc->H * ((desc->Tq + 127) / 128);
    if (threads_per_wg) *threads_per_wg = 256;
    return GTA_OK;
}

// which attention kernel gta_attn_fwd launches for desc when given a workspace (diagnostic; the names are the kernels' own)
extern "C" const char* gta_debug_attention_kernel(const GtaAttnDesc* d, int32_t* n_items, int32_t* rows_per_item) {
    if (gta_attn_fwd_supported(d)) return "";
    GtaFwdParams p;
    memset(&p, 0, sizeof p);
    if (build_ctab(d, p.ctab)) return "";
    const int esz = d->dtype == GTA_DTYPE_BF16 ? 2 : 4;
    const bool need_view = d->d_se3 > 0 || d->d_so3 > 0;
    p.B = d->B; p.H = d->H; p.Tq = d->Tq; p.Tk = d->Tk; p.dh = d->dh; p.flags = d->flags; p.kn = (float*)256; p.cs_q = d->d_so2 ? (const float*)256 : nullptr; p.kp = (void*)256;
    // (what gta_attn_fwd would hand the dispatch for a full call with LSE: only null / non-null matters here)
    p.Nq = d->Nq; p.Nk = d->Nk; p.Pq = d->Tq / d->Nq; p.Pk = d->Tk / d->Nk; p.nso2 = d->d_so2 / 2; p.lse = (float*)256;
    p.vrep_q = need_view ? (const float*)256 : nullptr; p.q_st = d->q_stride[2]; p.o_st = d->o_stride[2];
    p.qtiles = (padded_dh(d->dh) == 96 && esz == 2 && need_view) ? (void*)256 : nullptr;
    const bool two_stage = two_stage_plan(d);
    const int rows = !two_stage ? 128 : gta_fwd2_rows_per_item(p, padded_dh(d->dh), esz);
    if (n_items) *n_items = d->B * d->H * ((d->Tq + rows - 1) / rows);
    if (rows_per_item) *rows_per_item = rows;
    return !two_stage ? "gta_fwd_kernel" : gta_fwd2_attention_kernel_name(p, padded_dh(d->dh), esz);
}

extern "C" int gta_attn_fwd(const GtaAttnDesc* d, const void* q, const void* k, const void* v,
                            const float* vrep_q, const float* vrep_k, const float* cs_q, const float* cs_k,
                            const float* trans_coeff, const float* tau, void* out, float* lse,
                            void* workspace, int64_t workspace_bytes, void* stream) {
    int rc = check_common(d);
    if (rc) return rc;
    if (!k || !v || ((!q || !out) && !(d->flags & GTA_FLAG_PREP_ONLY))) return fail(GTA_E_BADARG, "null q/k/v/out");
    GtaFwdParams p;
    memset(&p, 0, sizeof p);
    rc = build_ctab(d, p.ctab);
    if (rc) return rc;
    const bool pre = (d->flags & GTA_FLAG_PRETRANSFORMED) != 0;
    const bool need_view = d->d_se3 > 0 || d->d_so3 > 0;
    const bool need_cs = d->d_so2 > 0;
    if (need_view && (!vrep_q || (!pre && !vrep_k))) return fail(GTA_E_BADARG, "se3/so3 slabs need vrep_q and vrep_k");
    if (need_cs && (!cs_q || (!pre && !cs_k))) return fail(GTA_E_BADARG, "so2 slab needs cs_q and cs_k");
    const int esz = d->dtype == GTA_DTYPE_BF16 ? 2 : 4;
    p.q = q; p.k = k; p.v = v; p.o = out; p.lse = lse;
    p.vrep_q = need_view ? vrep_q : nullptr; p.vrep_k = need_view ? vrep_k : nullptr;
    p.cs_q = need_cs ? cs_q : nullptr; p.cs_k = need_cs ? cs_k : nullptr;
    p.trans_coeff = trans_coeff; p.tau = tau;
    p.q_sb = d->q_stride[0]; p.q_sh = d->q_stride[1]; p.q_st = d->q_stride[2];
    p.k_sb = d->k_stride[0]; p.k_sh = d->k_stride[1]; p.k_st = d->k_stride[2];
    p.v_sb = d->v_stride[0]; p.v_sh = d->v_stride[1]; p.v_st = d->v_stride[2];
    p.o_sb = d->o_stride[0]; p.o_sh = d->o_stride[1]; p.o_st = d->o_stride[2];
    p.B = d->B; p.H = d->H; p.Tq = d->Tq; p.Tk = d->Tk; p.Nq = d->Nq; p.Nk = d->Nk;
    p.Pq = d->Tq / d->Nq; p.Pk = d->Tk / d->Nk;
    p.invPq = 1.0f / (float)p.Pq; p.invPk = 1.0f / (float)p.Pk;
    p.dh = d->dh; p.nso2 = d->d_so2 / 2;
    p.n_qtiles = (d->Tq + 127) / 128;
    p.flags = d->flags; p.scale = d->scale;
#ifdef GTA_ABLATE
    { const char* e = getenv("GTA_DBG"); p.dbg = e ? (uint32_t)atoi(e) : 0u; }
#endif
    const long n_wg = (long)d->B * d->H * p.n_qtiles;
    if (t_prof && !(d->flags & GTA_FLAG_PREP_ONLY)) {      // (start / end stamps of every work item: gta_debug_profile_next_attention_kernel)
        GtaFwdParams pk = p;
        pk.kn = (float*)1;
        const bool two_stage = workspace && two_stage_plan(d);
        const int rows = two_stage ? gta_fwd2_rows_per_item(pk, padded_dh(d->dh), esz) : 128;
        if ((int64_t)d->B * d->H * ((d->Tq + rows - 1) / rows) <= t_prof_items) p.prof = t_prof;
        t_prof = nullptr;
        t_prof_items = 0;
    }
    if (n_wg > 0x7fffffffL) return fail(GTA_E_UNSUPPORTED, "grid too large");
    if ((d->flags & GTA_FLAG_FP32_PRODUCTS) && d->dtype != GTA_DTYPE_F32)
        return fail(GTA_E_BADARG, "GTA_FLAG_FP32_PRODUCTS is for fp32 inputs (bf16 inputs ask for bf16 arithmetic)");
    if (workspace && two_stage_plan(d)) {
        const bool x3 = (d->flags & GTA_FLAG_FP32_PRODUCTS) != 0;
        if (workspace_bytes < gta_fwd2_workspace_bytes(d->B, d->H, d->Tk, padded_dh(d->dh), d->Nq, esz, x3))
            return fail(GTA_E_BADARG, "workspace smaller than gta_attn_fwd_workspace_bytes()");
        if (d->H > 65535 || d->B > 65535) return fail(GTA_E_UNSUPPORTED, "B or H above 65535");
        p.kp = workspace;
        p.kn = (float*)((char*)workspace + ((gta_fwd2_image_bytes(d->B, d->H, d->Tk, padded_dh(d->dh), x3) + 255) & ~255L));
        if (padded_dh(d->dh) == 96 && esz == 2 && need_view) p.qtiles = (char*)workspace + gta_fwd2_qtiles_offset(d->B, d->H, d->Tk, 96, false);
        rc = gta_fwd2_dispatch(p, padded_dh(d->dh), esz, !(d->flags & GTA_FLAG_KV_READY), !(d->flags & GTA_FLAG_PREP_ONLY),
                               (hipStream_t)stream);
        if (rc) return fail(rc, rc == GTA_E_LAUNCH ? hipGetErrorString(hipGetLastError()) : "no kernel instance");
        return GTA_OK;
    }
    rc = gta_fwd_dispatch(p, padded_dh(d->dh), esz, !(d->flags & GTA_FLAG_NO_DMA), (int)n_wg, (hipStream_t)stream);
    if (rc) return fail(rc, rc == GTA_E_LAUNCH ? hipGetErrorString(hipGetLastError()) : "no kernel instance");
    return GTA_OK;
}


// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
namespace {
struct BwdLayout { int64_t off_qimg, off_stats, off_dc, off_dt, off_kv, total; int n_prep, n_dq, n_dkv; };
BwdLayout bwd_layout(const GtaAttnDesc* d) {
    BwdLayout L;
    const int dhp = padded_dh(d->dh);
    const int64_t n_qt = (d->Tq + 63) / 64, n_kt = (d->Tk + 63) / 64;
    // (the fp32-faithful backward, r06: four images per tile -- hi and lo -- on both sides)
    const int64_t stage = ((d->flags & GTA_FLAG_FP32_PRODUCTS) ? 4LL : 2LL) * 64 * dhp * 2;
    L.n_prep = (int)(d->B * d->H * n_qt);
    L.n_dq = d->B * d->H * ((d->Tq + 127) / 128);
    L.n_dkv = d->B * d->H * ((d->Tk + 127) / 128);
    auto al = [](int64_t x) { return (x + 255) & ~255LL; };
    L.off_qimg = 0;
    L.off_stats = al(L.off_qimg + (int64_t)d->B * d->H * n_qt * stage);
    L.off_dc = al(L.off_stats + (int64_t)d->B * d->H * n_qt * 128 * 4);
    L.off_dt = al(L.off_dc + (int64_t)(L.n_prep + L.n_dq + L.n_dkv) * 4);
    L.off_kv = al(L.off_dt + (int64_t)L.n_dq * 4);
    L.total = al(L.off_kv + (int64_t)d->B * d->H * n_kt * stage);
    return L;
}
}  // namespace

extern "C" int64_t gta_attn_bwd_workspace_bytes(const GtaAttnDesc* desc) {
    if (gta_attn_fwd_supported(desc)) return 0;
    return bwd_layout(desc).total;
}

extern "C" int gta_attn_bwd(const GtaAttnDesc* d, const void* q, const void* k, const void* v, const void* out,
                            const void* dout, const float* lse, const float* vrep_q, const float* vrep_k,
                            const float* cs_q, const float* cs_k, const float* trans_coeff, const float* tau,
                            const void* kv_images, void* dq, void* dk, void* dv, const int64_t* dqkv_stride,
                            const int64_t* dout_stride, float* dtrans_coeff, float* dtau, void* workspace,
                            int64_t workspace_bytes,
                            void* stream) {
    int rc = check_common(d);
    if (rc) return rc;
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !dqkv_stride || !dout_stride || !workspace)
        return fail(GTA_E_BADARG, "null argument");
    if (d->flags & GTA_FLAG_PRETRANSFORMED) return fail(GTA_E_UNSUPPORTED, "backward of the pretransformed mode");
    if ((d->flags & GTA_FLAG_FP32_PRODUCTS) && (d->dtype != GTA_DTYPE_F32 || padded_dh(d->dh) > 64))
        return fail(GTA_E_UNSUPPORTED, "GTA_FLAG_FP32_PRODUCTS backward: fp32 inputs at dh <= 64 (other sizes: gta_rep_apply + gta_attn_bwd_plain_f32)");
    GtaBwdParams p;
    memset(&p, 0, sizeof p);
    rc = build_ctab(d, p.ctab);
    if (rc) return rc;
    const bool need_view = d->d_se3 > 0 || d->d_so3 > 0, need_cs = d->d_so2 > 0;
    if (need_view && (!vrep_q || !vrep_k)) return fail(GTA_E_BADARG, "se3/so3 slabs need vrep_q and vrep_k");
    if (need_cs && (!cs_q || !cs_k)) return fail(GTA_E_BADARG, "so2 slab needs cs_q and cs_k");
    const int esz = d->dtype == GTA_DTYPE_BF16 ? 2 : 4;
    for (int i = 0; i < 9; ++i) if ((dqkv_stride[i] * esz) % 16) return fail(GTA_E_BADARG, "gradient strides must keep rows 16-byte aligned");
    for (int i = 0; i < 3; ++i) if ((dout_stride[i] * esz) % 16) return fail(GTA_E_BADARG, "dout strides must keep rows 16-byte aligned");
    const BwdLayout L = bwd_layout(d);
    if (workspace_bytes < L.total) return fail(GTA_E_BADARG, "workspace smaller than gta_attn_bwd_workspace_bytes()");
    char* ws = (char*)workspace;
    if (!kv_images) {     // recompute K'/V' images with the forward's pre-pass
        GtaFwdParams f;
        memset(&f, 0, sizeof f);
        memcpy(f.ctab, p.ctab, sizeof f.ctab);
        f.k = k; f.v = v; f.kp = ws + L.off_kv;
        f.vrep_k = need_view ? vrep_k : nullptr; f.cs_k = need_cs ? cs_k : nullptr; f.trans_coeff = trans_coeff;
        f.k_sb = d->k_stride[0]; f.k_sh = d->k_stride[1]; f.k_st = d->k_stride[2];
        f.v_sb = d->v_stride[0]; f.v_sh = d->v_stride[1]; f.v_st = d->v_stride[2];
        f.B = d->B; f.H = d->H; f.Tq = d->Tq; f.Tk = d->Tk; f.Nq = d->Nq; f.Nk = d->Nk;
        f.Pq = d->Tq / d->Nq; f.Pk = d->Tk / d->Nk; f.invPq = 1.0f / f.Pq; f.invPk = 1.0f / f.Pk;
        f.dh = d->dh; f.nso2 = d->d_so2 / 2; f.flags = d->flags; f.scale = d->scale;      // (GTA_FLAG_FP32_PRODUCTS: the pre-pass writes hi and lo images, as the X3 walks read them)
        rc = gta_fwd2_dispatch(f, padded_dh(d->dh), esz, true, false, (hipStream_t)stream);
        if (rc) return fail(rc, "K/V pre-pass launch failed");
        kv_images = ws + L.off_kv;
    }
    p.q = q; p.k = k; p.v = v; p.out = out; p.dout = dout; p.lse = lse; p.dq = dq; p.dk = dk; p.dv = dv;
    p.vrep_q = need_view ? vrep_q : nullptr; p.vrep_k = need_view ? vrep_k : nullptr;
    p.cs_q = need_cs ? cs_q : nullptr; p.cs_k = need_cs ? cs_k : nullptr;
    p.trans_coeff = trans_coeff; p.tau = tau;
    p.kvimg = kv_images; p.qimg = ws + L.off_qimg; p.stats = (float*)(ws + L.off_stats);
    p.dc_partial = (float*)(ws + L.off_dc); p.dtrans_coeff = (d->d_se3 > 0) ? dtrans_coeff : nullptr;
    p.dt_partial = (tau && dtau) ? (float*)(ws + L.off_dt) : nullptr; p.dtau = (tau && dtau) ? dtau : nullptr;
    p.q_sb = d->q_stride[0]; p.q_sh = d->q_stride[1]; p.q_st = d->q_stride[2];
    p.k_sb = d->k_stride[0]; p.k_sh = d->k_stride[1]; p.k_st = d->k_stride[2];
    p.v_sb = d->v_stride[0]; p.v_sh = d->v_stride[1]; p.v_st = d->v_stride[2];
    p.o_sb = d->o_stride[0]; p.o_sh = d->o_stride[1]; p.o_st = d->o_stride[2];
    p.do_sb = dout_stride[0]; p.do_sh = dout_stride[1]; p.do_st = dout_stride[2];
    p.dq_sb = dqkv_stride[0]; p.dq_sh = dqkv_stride[1]; p.dq_st = dqkv_stride[2];
    p.dk_sb = dqkv_stride[3]; p.dk_sh = dqkv_stride[4]; p.dk_st = dqkv_stride[5];
    p.dv_sb = dqkv_stride[6]; p.dv_sh = dqkv_stride[7]; p.dv_st = dqkv_stride[8];
    p.dc_off_prep = 0; p.dc_off_dq = L.n_prep; p.dc_off_dkv = L.n_prep + L.n_dq; p.dc_total = L.n_prep + L.n_dq + L.n_dkv;
    p.B = d->B; p.H = d->H; p.Tq = d->Tq; p.Tk = d->Tk; p.Nq = d->Nq; p.Nk = d->Nk;
    p.Pq = d->Tq / d->Nq; p.Pk = d->Tk / d->Nk; p.invPq = 1.0f / (float)p.Pq; p.invPk = 1.0f / (float)p.Pk;
    p.dh = d->dh; p.nso2 = d->d_so2 / 2; p.flags = d->flags; p.scale = d->scale;
    if (d->H > 65535 || d->B > 65535) return fail(GTA_E_UNSUPPORTED, "B or H above 65535");
    rc = gta_bwd_dispatch(p, padded_dh(d->dh), esz, (hipStream_t)stream);
    if (rc) return fail(rc, rc == GTA_E_LAUNCH ? hipGetErrorString(hipGetLastError()) : "no kernel instance");
    return GTA_OK;
}


// ---------------------------------------------------------------------------------------------
// plain attention (identity layout) with an optional key bias: the attention stage of the generic path
// ---------------------------------------------------------------------------------------------
extern "C" int gta_attn_fwd_plain(const GtaAttnDesc* d, const void* q, const void* k, const void* v,
                                  const float* key_bias, int64_t bias_pitch, const float* tau, void* out, float* lse,
                                  void* stream) {
    if (!d || !q || !k || !v || !out) return fail(GTA_E_BADARG, "null argument");
    if (d->abi_version != GTA_ABI_VERSION) return fail(GTA_E_BADARG, "abi_version mismatch");
    if (d->dtype != GTA_DTYPE_F32 && d->dtype != GTA_DTYPE_BF16) return fail(GTA_E_BADARG, "bad dtype");
    if (d->B <= 0 || d->H <= 0 || d->Tq <= 0 || d->Tk <= 0 || d->dh <= 0) return fail(GTA_E_BADARG, "non-positive size");
    const int dhp = (d->dh + 7) / 8 * 8;
    if (dhp != d->dh || d->dh > 128) return fail(GTA_E_UNSUPPORTED, "plain attention needs dh % 8 == 0 and dh <= 128 (pad the channels)");
    if (key_bias && (bias_pitch % 64 || bias_pitch < d->Tk)) return fail(GTA_E_BADARG, "bias_pitch must be a multiple of 64 >= Tk");
    const int esz = d->dtype == GTA_DTYPE_BF16 ? 2 : 4;
    GtaFwdParams p;
    memset(&p, 0, sizeof p);
    p.q = q; p.k = k; p.v = v; p.o = out; p.lse = lse; p.tau = tau;
    p.kbias = key_bias; p.kbias_pitch = bias_pitch;
    p.q_sb = d->q_stride[0]; p.q_sh = d->q_stride[1]; p.q_st = d->q_stride[2];
    p.k_sb = d->k_stride[0]; p.k_sh = d->k_stride[1]; p.k_st = d->k_stride[2];
    p.v_sb = d->v_stride[0]; p.v_sh = d->v_stride[1]; p.v_st = d->v_stride[2];
    p.o_sb = d->o_stride[0]; p.o_sh = d->o_stride[1]; p.o_st = d->o_stride[2];
    p.B = d->B; p.H = d->H; p.Tq = d->Tq; p.Tk = d->Tk; p.Nq = 1; p.Nk = 1; p.Pq = d->Tq; p.Pk = d->Tk;
    p.invPq = 1.0f / p.Pq; p.invPk = 1.0f / p.Pk;
    p.dh = d->dh; p.nso2 = 0; p.n_qtiles = (d->Tq + 127) / 128; p.scale = d->scale;
    p.flags = d->flags & GTA_FLAG_FP32_PRODUCTS;
    if (p.flags && d->dtype != GTA_DTYPE_F32) return fail(GTA_E_BADARG, "GTA_FLAG_FP32_PRODUCTS is for fp32 inputs");
    const long n_wg = (long)d->B * d->H * p.n_qtiles;
    int rc = gta_fwd_dispatch(p, padded_dh(d->dh), esz, true, (int)n_wg, (hipStream_t)stream);
    if (rc) return fail(rc, rc == GTA_E_LAUNCH ? hipGetErrorString(hipGetLastError()) : "no kernel instance");
    return GTA_OK;
}
