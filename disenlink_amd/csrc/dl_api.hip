// C ABI of libdisenlink_hip.so: argument validation, workspace carving and dispatch.
#include <stdarg.h>
#include <string.h>
#include "dl_common.h"
#include "dl_kernels.h"

namespace dl {

static thread_local char g_err[512] = "";
static int g_force_generic = 0;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DL_E_LAUNCH;
    }
    return DL_OK;
}

static int check_shape(int K, int d) {
    DL_REQUIRE(K >= 1 && K <= DL_MAX_FACTORS, "K=%d outside 1..%d", K, DL_MAX_FACTORS);
    DL_REQUIRE(d >= 1 && d <= 4096, "d=%d outside 1..4096", d);
    return DL_OK;
}

static int check_plan(const dl_csr_plan* c, const char* what) {
    DL_REQUIRE(c != nullptr, "%s is NULL", what);
    DL_REQUIRE(c->n_rows >= 0 && c->n_entries >= 0 && c->row_offset >= 0, "%s: negative size", what);
    DL_REQUIRE((long long)c->row_offset + c->n_rows <= c->n_total, "%s: rows [%d, %d) exceed n_total=%d", what,
               c->row_offset, c->row_offset + c->n_rows, c->n_total);
    if (c->n_rows > 0) DL_REQUIRE(c->rowptr != nullptr, "%s.rowptr is NULL", what);
    if (c->n_entries > 0) DL_REQUIRE(c->col != nullptr, "%s.col is NULL", what);
    return DL_OK;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Workspace layout (256-byte aligned blocks):
//   dw[E] | dwr[E] | ds[n_total*K] | vec_part[n_slots*K] | row_part[n_slots*2*K*d]
struct Workspace {
    float *dw, *dwr, *ds, *vec_part, *row_part;
    size_t bytes;
};

static Workspace carve(const dl_csr_plan* c, int K, int d, void* ws) {
    Workspace w;
    const size_t e = align256((size_t)c->n_entries * sizeof(float));
    const size_t nk = align256((size_t)c->n_total * K * sizeof(float));
    const size_t vp = align256((size_t)c->n_slots * K * sizeof(float));
    const size_t rp = align256((size_t)c->n_slots * 2 * K * d * sizeof(float));
    char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    w.dw = (float*)base;
    w.dwr = (float*)(base + e);
    w.ds = (float*)(base + 2 * e);
    w.vec_part = (float*)(base + 2 * e + nk);
    w.row_part = (float*)(base + 2 * e + nk + vp);
    w.bytes = 2 * e + nk + vp + rp + 256;
    return w;
}

static int check_workspace(const dl_csr_plan* c, int K, int d, void* ws, size_t ws_bytes, Workspace* out) {
    *out = carve(c, K, d, ws);
    if (!ws || ws_bytes < out->bytes) {
        set_error("workspace too small: have %zu, need %zu (dl_workspace_bytes)", ws ? ws_bytes : (size_t)0, out->bytes);
        return DL_E_WORKSPACE;
    }
    return DL_OK;
}

// A usable segment plan: the arrays are there and the position counts are whole workgroups (a workgroup reads the
// DL_UNIT_SEGS descriptors of its group unconditionally; plans of another layout take the generic kernels instead).
static bool has_seg_plan(const dl_csr_plan* c) {
    return c->seg_len > 0 && c->seg_len <= DL_WAVE && c->n_seg > 0 && c->seg_row && c->seg_beg && c->seg_end && c->seg_slot &&
           c->n_slices >= 1 && c->slice_seg0 && c->slice_max_seg > 0 &&
           c->n_seg % DL_UNIT_SEGS == 0 && c->slice_max_seg % DL_UNIT_SEGS == 0 &&
           (c->n_multi == 0 || (c->multi_row && c->multi_slot0));
}

static bool use_fast(const dl_csr_plan* c, int K, int d, int dtype) {
    return !g_force_generic && fast_supported(K, d, dtype) && has_seg_plan(c);
}

// bf16 tables exist only on the tuned path
static int check_dtype(const dl_csr_plan* c, int K, int d, int dtype) {
    DL_REQUIRE(dtype == DL_F32 || dtype == DL_BF16, "unknown dtype %d", dtype);
    if (dtype == DL_BF16)
        DL_REQUIRE(c && use_fast(c, K, d, dtype),
                   "bf16 tables need a tuned kernel for K=%d d=%d and a segment plan (no generic bf16 path)", K, d);
    return DL_OK;
}

}  // namespace dl

using namespace dl;

extern "C" {

const char* dl_version(void) { return "disenlink_hip 0.8 (gfx950)"; }
const char* dl_last_error(void) { return g_err; }
int dl_has_fast_path(int K, int d) { return fast_supported(K, d, DL_F32) ? 1 : 0; }
int dl_has_fast_path_dtype(int K, int d, dl_dtype dtype) { return fast_supported(K, d, (int)dtype) ? 1 : 0; }

int dl_set_force_generic(int on) {
    int old = g_force_generic;
    if (on >= 0) g_force_generic = on ? 1 : 0;
    return old;
}

size_t dl_workspace_bytes(const dl_csr_plan* plan, int K, int d) {
    if (!plan || K < 1 || d < 1) return 0;
    return carve(plan, K, d, nullptr).bytes;
}

int dl_project_supported(int d) { return project_supported(d) ? 1 : 0; }

size_t dl_project_fwd_workspace_bytes(int N, int F, int K, int nhid, int d, int two_layer) {
    (void)F;
    if (N <= 0 || K < 1 || nhid < 1 || d < 1) return 0;
    return project_fwd_workspace_bytes(N, F, K, nhid, d, two_layer != 0);
}

size_t dl_project_hidden_floats(int N, int K, int nhid) {
    if (N <= 0 || K < 1 || nhid < 1) return 0;
    return (size_t)K * nhid * (size_t)((N + 3) & ~3);
}

size_t dl_project_xplanes_bytes(int N, int F) { return (N > 0 && F > 0) ? project_xplanes_bytes(N, F) : 0; }

int dl_project_xplanes_build(const float* x, int N, int F, void* xplanes, size_t xplanes_bytes, void* stream) {
    DL_REQUIRE(N >= 0 && F >= 1, "bad size N=%d F=%d", N, F);
    if (N == 0) return DL_OK;
    DL_REQUIRE(x && xplanes && xplanes_bytes >= project_xplanes_bytes(N, F) && ((uintptr_t)xplanes & 15) == 0,
               "x planes buffer: %zu bytes given, %zu needed (dl_project_xplanes_bytes), 16-byte aligned", xplanes_bytes,
               project_xplanes_bytes(N, F));
    return project_xplanes_build(x, N, F, xplanes, (hipStream_t)stream);
}

int dl_project_fwd(const float* x, int N, int F, int K, int nhid, int d, const float* W1, const float* b1,
                   const float* W2, const float* b2, float* Z, float* hid_out, void* ws, size_t ws_bytes,
                   void* stream) {
    return dl_project_fwd_xp(x, N, F, K, nhid, d, W1, b1, W2, b2, Z, hid_out, ws, ws_bytes, nullptr, stream);
}

int dl_project_fwd_xp(const float* x, int N, int F, int K, int nhid, int d, const float* W1, const float* b1,
                      const float* W2, const float* b2, float* Z, float* hid_out, void* ws, size_t ws_bytes,
                      const void* xplanes, void* stream) {
    if (int rc = check_shape(K, d)) return rc;
    DL_REQUIRE(project_supported(d), "projection kernel supports d in {32, 64, 128}, got %d", d);
    DL_REQUIRE(N >= 0 && F >= 1 && nhid >= 1, "bad size N=%d F=%d nhid=%d", N, F, nhid);
    DL_REQUIRE(K <= 65535, "K too large for the launch grid");
    DL_REQUIRE((W2 == nullptr) == (b2 == nullptr), "W2 and b2 must both be given (two-layer) or both NULL");
    if (N == 0) return DL_OK;
    DL_REQUIRE(x && W1 && b1 && Z, "NULL argument");
    DL_REQUIRE((long long)N * K * d < (1LL << 40) && (long long)128 * F < (1LL << 31), "projection sizes out of range");
    DL_REQUIRE(hid_out == nullptr || W2 != nullptr, "hid_out is for the two-layer form only");
    DL_REQUIRE(xplanes == nullptr || ((uintptr_t)xplanes & 15) == 0, "x planes must be 16-byte aligned");
    return project_fwd(x, N, F, K, nhid, d, W1, b1, W2, b2, Z, ws, ws_bytes, hid_out, (hipStream_t)stream,
                       W2 != nullptr ? xplanes : nullptr);
}

size_t dl_project_bwd_workspace_bytes(int N, int F, int K, int nhid, int d, int two_layer) {
    if (N <= 0 || F < 1 || K < 1 || nhid < 1 || d < 1) return 0;
    return project_bwd_workspace_bytes(N, F, K, two_layer ? nhid : 1, d, two_layer != 0);
}

int dl_project_bwd(const float* x, int N, int F, int K, int nhid, int d, const float* W1, const float* b1,
                   const float* W2, const float* dZ, const float* hid, float* dW1, float* db1, float* dW2, float* db2,
                   void* ws, size_t ws_bytes, void* stream) {
    return dl_project_bwd_xp(x, N, F, K, nhid, d, W1, b1, W2, dZ, hid, dW1, db1, dW2, db2, ws, ws_bytes, nullptr, stream);
}

int dl_project_bwd_xp(const float* x, int N, int F, int K, int nhid, int d, const float* W1, const float* b1,
                      const float* W2, const float* dZ, const float* hid, float* dW1, float* db1, float* dW2, float* db2,
                      void* ws, size_t ws_bytes, const void* xplanes, void* stream) {
    if (int rc = check_shape(K, d)) return rc;
    DL_REQUIRE(project_supported(d), "projection kernel supports d in {32, 64, 128}, got %d", d);
    DL_REQUIRE(N >= 0 && F >= 1 && nhid >= 1, "bad size N=%d F=%d nhid=%d", N, F, nhid);
    DL_REQUIRE(K <= 4096, "K too large for the launch grid");
    const bool two = W2 != nullptr;
    DL_REQUIRE(dW1 && db1 && (!two || (dW2 && db2)), "NULL gradient output");
    if (N == 0) {                                              // no nodes: every gradient is zero
        hipStream_t st = (hipStream_t)stream;
        const size_t m = two ? (size_t)nhid : (size_t)d;
        hipError_t e = hipMemsetAsync(dW1, 0, sizeof(float) * K * m * F, st);
        if (e == hipSuccess) e = hipMemsetAsync(db1, 0, sizeof(float) * K * m, st);
        if (two && e == hipSuccess) e = hipMemsetAsync(dW2, 0, sizeof(float) * (size_t)K * d * nhid, st);
        if (two && e == hipSuccess) e = hipMemsetAsync(db2, 0, sizeof(float) * (size_t)K * d, st);
        DL_REQUIRE(e == hipSuccess, "hipMemsetAsync: %s", hipGetErrorString(e));
        return DL_OK;
    }
    DL_REQUIRE(x && W1 && b1 && dZ, "NULL argument");
    const size_t need = project_bwd_workspace_bytes(N, F, K, two ? nhid : 1, d, two);
    DL_REQUIRE(ws != nullptr && ws_bytes >= need, "workspace too small: %zu < %zu bytes (dl_project_bwd_workspace_bytes)",
               ws_bytes, need);
    DL_REQUIRE(hid == nullptr || two, "hid is for the two-layer form only");
    DL_REQUIRE(xplanes == nullptr || ((uintptr_t)xplanes & 15) == 0, "x planes must be 16-byte aligned");
    return project_bwd(x, N, F, K, two ? nhid : 1, d, W1, b1, W2, dZ, hid, dW1, db1, dW2, db2, ws, (hipStream_t)stream,
                       xplanes);
}

int dl_route_fwd(const dl_graph* g, const void* Z, int K, int d, dl_dtype dtype, float t, uint8_t* p, float* a,
                 float* s, void* ws, size_t ws_bytes, void* stream) {
    DL_REQUIRE(g != nullptr, "graph is NULL");
    const dl_csr_plan* c = &g->csr;
    if (int rc = check_plan(c, "graph")) return rc;
    if (int rc = check_shape(K, d)) return rc;
    if (int rc = check_dtype(c, K, d, dtype)) return rc;
    DL_REQUIRE(t != 0.0f, "temperature is 0");
    if (c->n_rows == 0) return DL_OK;
    DL_REQUIRE(Z && s, "Z or s is NULL");
    if (c->n_entries > 0) DL_REQUIRE(p && a, "p or a is NULL");
    if (use_fast(c, K, d, dtype)) {
        Workspace w;
        if (int rc = check_workspace(c, K, d, ws, ws_bytes, &w)) return rc;
        // optional routing plan (XCD-sliced and / or upper-triangle with mirrored writes)
        const dl_csr_plan* rp = &g->route;
        const bool have = has_seg_plan(rp) && rp->n_rows == c->n_rows && rp->n_total == c->n_total &&
                          rp->row_offset == c->row_offset && rp->rowptr == c->rowptr && rp->col == c->col;
        const bool mirror = have && g->route_mirror != 0;
        if (mirror)
            DL_REQUIRE((g->rev != nullptr || c->n_entries == 0) && c->row_offset == 0 && c->n_rows == c->n_total,
                       "route_mirror needs rev and an unsharded plan");
        return fast_route_fwd(c, have ? rp : nullptr, mirror, g->rev, Z, K, d, dtype, t, p, a, s, w.vec_part,
                              (hipStream_t)stream);
    }
    return generic_route_fwd(c, (const float*)Z, K, d, t, p, a, s, (hipStream_t)stream);
}

int dl_aggregate_fwd(const dl_graph* g, const void* Z, int K, int d, dl_dtype dtype, float beta, const uint8_t* p,
                     const float* a, const float* s, void* H, void* ws, size_t ws_bytes, void* stream) {
    DL_REQUIRE(g != nullptr, "graph is NULL");
    const dl_csr_plan* c = &g->csr;
    if (int rc = check_plan(c, "graph")) return rc;
    if (int rc = check_shape(K, d)) return rc;
    if (int rc = check_dtype(c, K, d, dtype)) return rc;
    if (c->n_rows == 0) return DL_OK;
    DL_REQUIRE(Z && s && H, "Z, s or H is NULL");
    if (c->n_entries > 0) DL_REQUIRE(p && a, "p or a is NULL");
    if (use_fast(c, K, d, dtype)) {
        Workspace w;
        if (int rc = check_workspace(c, K, d, ws, ws_bytes, &w)) return rc;
        return fast_aggregate_fwd(c, Z, K, d, dtype, beta, p, a, s, H, w.row_part, (hipStream_t)stream);
    }
    return generic_aggregate_fwd(c, (const float*)Z, K, d, beta, p, a, s, (float*)H, (hipStream_t)stream);
}

int dl_score_pairs_fwd(const void* Z, const void* H, int N, int K, int d, dl_dtype dtype, float t,
                       const int32_t* pu, const int32_t* pv, int n_pairs, const dl_pair_incidence* by_u,
                       float* prob, float* coef, void* stream) {
    if (int rc = check_shape(K, d)) return rc;
    if (int rc = check_dtype(by_u ? &by_u->csr : nullptr, K, d, dtype)) return rc;
    DL_REQUIRE(N >= 0 && n_pairs >= 0, "negative size");
    DL_REQUIRE(t != 0.0f, "temperature is 0");
    if (n_pairs == 0) return DL_OK;
    DL_REQUIRE(Z && H && pu && pv && prob, "NULL argument");
    if (by_u) {
        if (int rc = check_plan(&by_u->csr, "by_u")) return rc;
        DL_REQUIRE(by_u->csr.n_entries == n_pairs && by_u->n_pairs == n_pairs && by_u->inc_pair,
                   "by_u must list each of the %d pairs exactly once", n_pairs);
        DL_REQUIRE(by_u->csr.n_total == N, "by_u.n_total=%d != N=%d", by_u->csr.n_total, N);
        if (use_fast(&by_u->csr, K, d, dtype))
            return fast_score_pairs_fwd(by_u, Z, H, K, d, dtype, t, prob, coef, (hipStream_t)stream);
    }
    DL_REQUIRE(coef == nullptr, "coef output needs the tuned scorer (a (K,d) with a fast path and a by_u plan)");
    return generic_score_pairs_fwd((const float*)Z, (const float*)H, K, d, t, pu, pv, n_pairs, prob,
                                   (hipStream_t)stream);
}

size_t dl_score_allpairs_workspace_bytes(int N, int K, int d, dl_dtype dtype) {
    if (N <= 0 || K < 1 || d < 1 || dtype != DL_F32 || g_force_generic) return 0;
    return dense_score_workspace_bytes(N, K, d);
}

int dl_score_allpairs_fwd(const void* Z, const void* H, int N, int K, int d, dl_dtype dtype, float t, float* prob,
                          void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_shape(K, d)) return rc;
    DL_REQUIRE(dtype == DL_F32 || dtype == DL_BF16, "unknown dtype %d", dtype);
    DL_REQUIRE(N >= 0 && N <= 46340, "dense [N,N] scoring needs 0 <= N <= 46340, got %d", N);
    DL_REQUIRE(t != 0.0f, "temperature is 0");
    if (N == 0) return DL_OK;
    DL_REQUIRE(Z && H && prob, "NULL argument");
    if (!g_force_generic && dtype == DL_F32 && dense_mfma_supported(d))      // Gram products on the matrix cores
        return dense_mfma_score_allpairs_fwd((const float*)Z, (const float*)H, N, K, d, t, prob, ws, ws_bytes,
                                             (hipStream_t)stream);
    if (!g_force_generic && fast_supported(K, d, dtype))
        return fast_score_allpairs_fwd(Z, H, N, K, d, dtype, t, prob, (hipStream_t)stream);
    DL_REQUIRE(dtype == DL_F32, "bf16 tables need a tuned kernel for K=%d d=%d", K, d);
    return generic_score_allpairs_fwd((const float*)Z, (const float*)H, N, K, d, t, prob, (hipStream_t)stream);
}

int dl_auc_pair_counts_supported(int n_pos, int n_neg) {
    return n_pos >= 0 && n_neg >= 0 && auc_counts_supported(n_pos, n_neg) ? 1 : 0;
}

int dl_auc_pair_counts(const float* score, const int64_t* pos_idx, int n_pos, const int64_t* neg_idx, int n_neg,
                       unsigned long long* u2, void* stream) {
    DL_REQUIRE(n_pos >= 0 && n_neg >= 0, "negative size");
    DL_REQUIRE(u2 != nullptr, "u2 is NULL");
    DL_REQUIRE(auc_counts_supported(n_pos, n_neg), "n_pos * n_neg too large for the slice-and-search form (%d x %d)",
               n_pos, n_neg);
    if (n_pos > 0 && n_neg > 0) DL_REQUIRE(score && pos_idx && neg_idx, "NULL argument");
    return auc_pair_counts(score, pos_idx, n_pos, neg_idx, n_neg, u2, (hipStream_t)stream);
}

int dl_auc_pair_counts_add(const float* score, const int64_t* pos_idx, int n_pos, const int64_t* neg_idx, int n_neg,
                           unsigned long long* u2, void* stream) {
    DL_REQUIRE(n_pos >= 0 && n_neg >= 0, "negative size");
    DL_REQUIRE(u2 != nullptr, "u2 is NULL");
    DL_REQUIRE(auc_counts_supported(n_pos, n_neg), "n_pos * n_neg too large for the slice-and-search form (%d x %d)",
               n_pos, n_neg);
    if (n_pos > 0 && n_neg > 0) DL_REQUIRE(score && pos_idx && neg_idx, "NULL argument");
    return auc_pair_counts(score, pos_idx, n_pos, neg_idx, n_neg, u2, (hipStream_t)stream, /*clear=*/false);
}

size_t dl_epoch_state_bytes(void) { return epoch_state_bytes(); }

int dl_epoch_finish(int n_bufs, const float* const* params, float* const* best, const size_t* numel, const float* loss,
                    unsigned long long* u2, double denom2, void* state, double* hist, long long max_epochs,
                    long long patience, double* host_ring, int ring, void* stream) {
    DL_REQUIRE(n_bufs >= 0 && n_bufs <= DL_ADAM_MAX_BUFS, "n_bufs=%d outside 0..%d", n_bufs, DL_ADAM_MAX_BUFS);
    DL_REQUIRE(loss && u2 && state, "loss / u2 / state is NULL");
    DL_REQUIRE(max_epochs >= 0 && (max_epochs == 0 || hist != nullptr), "hist is NULL");
    DL_REQUIRE(patience >= 0, "negative patience");
    if (n_bufs > 0) DL_REQUIRE(params && best && numel, "NULL argument");
    for (int i = 0; i < n_bufs; ++i) DL_REQUIRE(numel[i] == 0 || (params[i] && best[i]), "buffer %d: NULL pointer", i);
    DL_REQUIRE(host_ring == nullptr || ring >= 1, "ring=%d", ring);
    return epoch_finish(n_bufs, params, best, numel, loss, u2, denom2, state, hist, max_epochs, patience, host_ring, ring,
                        (hipStream_t)stream);
}

int dl_pair_bce(const float* prob, const float* y, const float* w, int n_pairs, float* loss, float* g, void* ws,
                size_t ws_bytes, void* stream) {
    DL_REQUIRE(n_pairs >= 0, "negative size");
    DL_REQUIRE(loss != nullptr, "loss is NULL");
    DL_REQUIRE(ws != nullptr && ws_bytes >= 4096 + 256, "dl_pair_bce needs >= 4352 bytes of workspace");
    if (n_pairs > 0) DL_REQUIRE(prob && y && w && g, "NULL argument");
    float* partial = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    return pair_bce(prob, y, w, n_pairs, loss, g, partial, (hipStream_t)stream);
}

int dl_adam_step(int n_bufs, float* const* params, const float* const* grads, float* const* exp_avg,
                 float* const* exp_avg_sq, const size_t* numel, float* state, double lr, double beta1, double beta2,
                 double eps, double weight_decay, void* stream) {
    DL_REQUIRE(n_bufs >= 0 && n_bufs <= DL_ADAM_MAX_BUFS, "n_bufs=%d outside 0..%d", n_bufs, DL_ADAM_MAX_BUFS);
    DL_REQUIRE(state != nullptr, "state is NULL");
    DL_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, "bad Adam hyper-parameters");
    if (n_bufs > 0) DL_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel, "NULL argument");
    for (int i = 0; i < n_bufs; ++i)
        DL_REQUIRE(numel[i] == 0 || (params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i]), "buffer %d: NULL pointer", i);
    return adam_step(n_bufs, params, grads, exp_avg, exp_avg_sq, numel, state, lr, beta1, beta2, eps, weight_decay,
                     (hipStream_t)stream);
}

int dl_adam_step_at(int n_bufs, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const size_t* numel, float* state, long long step, double lr, double beta1,
                    double beta2, double eps, double weight_decay, void* stream) {
    DL_REQUIRE(n_bufs >= 0 && n_bufs <= DL_ADAM_MAX_BUFS, "n_bufs=%d outside 0..%d", n_bufs, DL_ADAM_MAX_BUFS);
    DL_REQUIRE(step >= 1 && step < (1ll << 24), "step=%lld outside 1..2^24-1", step);
    DL_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, "bad Adam hyper-parameters");
    if (n_bufs > 0) DL_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel, "NULL argument");
    for (int i = 0; i < n_bufs; ++i)
        DL_REQUIRE(numel[i] == 0 || (params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i]), "buffer %d: NULL pointer", i);
    return adam_step(n_bufs, params, grads, exp_avg, exp_avg_sq, numel, state, lr, beta1, beta2, eps, weight_decay,
                     (hipStream_t)stream, step);
}

int dl_score_pairs_bwd(const void* Z, const void* H, int K, int d, dl_dtype dtype, float t,
                       const dl_pair_incidence* inc, const float* prob, const float* g_prob, const float* coef,
                       float* dZ, float* dH, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_shape(K, d)) return rc;
    DL_REQUIRE(inc != nullptr, "incidence is NULL");
    const dl_csr_plan* c = &inc->csr;
    if (int rc = check_plan(c, "incidence")) return rc;
    if (int rc = check_dtype(c, K, d, dtype)) return rc;
    DL_REQUIRE(t != 0.0f, "temperature is 0");
    if (c->n_rows == 0) return DL_OK;
    DL_REQUIRE(Z && H && dZ && dH, "NULL argument");
    if (c->n_entries > 0) DL_REQUIRE(inc->inc_pair && prob && g_prob, "NULL pair argument");
    if (use_fast(c, K, d, dtype)) {
        Workspace w;
        if (int rc = check_workspace(c, K, d, ws, ws_bytes, &w)) return rc;
        return fast_score_pairs_bwd(inc, Z, H, K, d, dtype, t, prob, g_prob, coef, dZ, dH, w.row_part,
                                    (hipStream_t)stream);
    }
    return generic_score_pairs_bwd(inc, (const float*)Z, (const float*)H, K, d, t, prob, g_prob, dZ, dH,
                                   (hipStream_t)stream);
}

int dl_score_allpairs_bwd(const void* Z, const void* H, int N, int K, int d, dl_dtype dtype, float t,
                          const dl_pair_incidence* inc, const int32_t* pu, const int32_t* pv, int n_pairs,
                          const float* prob, const float* g_prob, float* dZ, float* dH, void* ws, size_t ws_bytes,
                          void* stream) {
    if (int rc = check_shape(K, d)) return rc;
    DL_REQUIRE(inc != nullptr, "incidence is NULL");
    const dl_csr_plan* c = &inc->csr;
    if (int rc = check_plan(c, "incidence")) return rc;
    DL_REQUIRE(N >= 0 && N <= 46340, "dense [N,N] scoring needs 0 <= N <= 46340, got %d", N);
    DL_REQUIRE(c->n_total == N, "incidence.n_total=%d != N=%d", c->n_total, N);
    DL_REQUIRE(n_pairs >= 0 && inc->n_pairs == n_pairs && c->n_entries == 2 * (long long)n_pairs,
               "the incidence plan must list each of the %d pairs once per endpoint", n_pairs);
    if (n_pairs > 0) DL_REQUIRE(pu && pv && prob && g_prob, "NULL pair argument");
    // the gathered vectors use the per-entry scratch of the plan's workspace (2 x n_entries floats >= 2 x n_pairs)
    Workspace w;
    if (int rc = check_workspace(c, K, d, ws, ws_bytes, &w)) return rc;
    if (int rc = gather_dense_pairs(pu, pv, N, n_pairs, prob, g_prob, w.dw, w.dwr, (hipStream_t)stream)) return rc;
    return dl_score_pairs_bwd(Z, H, K, d, dtype, t, inc, w.dw, w.dwr, nullptr, dZ, dH, ws, ws_bytes, stream);
}

int dl_score_pairs_train_supported(const dl_pair_incidence* inc, int K, int d, dl_dtype dtype) {
    return inc != nullptr && use_fast(&inc->csr, K, d, dtype) ? 1 : 0;
}

int dl_score_pairs_train(const void* Z, const void* H, int K, int d, dl_dtype dtype, float t,
                         const dl_pair_incidence* inc, const float* y, const float* w, float* prob, float* dZ,
                         float* dH, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_shape(K, d)) return rc;
    DL_REQUIRE(inc != nullptr, "incidence is NULL");
    const dl_csr_plan* c = &inc->csr;
    if (int rc = check_plan(c, "incidence")) return rc;
    if (int rc = check_dtype(c, K, d, dtype)) return rc;
    DL_REQUIRE(t != 0.0f, "temperature is 0");
    DL_REQUIRE(use_fast(c, K, d, dtype), "dl_score_pairs_train needs a tuned kernel for K=%d d=%d (dl_score_pairs_train_supported)",
               K, d);
    if (c->n_rows == 0) return DL_OK;
    DL_REQUIRE(Z && H && dZ && dH, "NULL argument");
    if (c->n_entries > 0) DL_REQUIRE(inc->inc_pair && y && w && prob, "NULL pair argument");
    Workspace wsp;
    if (int rc = check_workspace(c, K, d, ws, ws_bytes, &wsp)) return rc;
    return fast_score_pairs_train(inc, Z, H, K, d, dtype, t, y, w, prob, dZ, dH, wsp.row_part, (hipStream_t)stream);
}

int dl_route_aggregate_bwd_phase1(const dl_graph* g, const void* Z, int K, int d, dl_dtype dtype, float beta,
                                  const uint8_t* p, const float* a, const float* s, const float* dH, float* dw,
                                  float* dwr, float* ds, void* ws, size_t ws_bytes, void* stream) {
    DL_REQUIRE(g != nullptr, "graph is NULL");
    const dl_csr_plan* c = &g->csr;
    if (int rc = check_plan(c, "graph")) return rc;
    if (int rc = check_shape(K, d)) return rc;
    if (int rc = check_dtype(c, K, d, dtype)) return rc;
    if (c->n_rows == 0) return DL_OK;
    DL_REQUIRE(Z && s && dH && ds, "NULL argument");
    if (c->n_entries > 0) DL_REQUIRE(p && a && dw && dwr, "NULL per-edge argument");
    if (use_fast(c, K, d, dtype)) {
        Workspace w;
        if (int rc = check_workspace(c, K, d, ws, ws_bytes, &w)) return rc;
        return fast_bwd_phase1(c, Z, K, d, dtype, beta, p, a, s, dH, dw, dwr, ds, w.vec_part, (hipStream_t)stream);
    }
    return generic_bwd_phase1(c, (const float*)Z, K, d, beta, p, a, s, dH, dw, dwr, ds, (hipStream_t)stream);
}

static int bwd_phase2_impl(const dl_graph* g, const void* Z, int K, int d, dl_dtype dtype, float beta, float t,
                           const uint8_t* p, const float* a, const float* s, const float* dH, const float* dw,
                           const float* dwr, const float* ds, const float* dz_in, const float* scale, float* dZ, void* ws,
                           size_t ws_bytes, void* stream) {
    DL_REQUIRE(g != nullptr, "graph is NULL");
    const dl_csr_plan* c = &g->csr;
    if (int rc = check_plan(c, "graph")) return rc;
    if (int rc = check_shape(K, d)) return rc;
    if (int rc = check_dtype(c, K, d, dtype)) return rc;
    DL_REQUIRE(t != 0.0f, "temperature is 0");
    if (c->n_rows == 0) return DL_OK;
    DL_REQUIRE(Z && s && dH && ds && dZ, "NULL argument");
    if (c->n_entries > 0) DL_REQUIRE(p && a && dw && dwr, "NULL per-edge argument");
    if (use_fast(c, K, d, dtype)) {
        Workspace w;
        if (int rc = check_workspace(c, K, d, ws, ws_bytes, &w)) return rc;
        return fast_bwd_phase2(c, Z, K, d, dtype, beta, t, p, a, s, dH, dw, dwr, ds, dz_in, scale, dZ, w.row_part,
                               (hipStream_t)stream);
    }
    return generic_bwd_phase2(c, (const float*)Z, K, d, beta, t, p, a, s, dH, dw, dwr, ds, dz_in, scale, dZ,
                              (hipStream_t)stream);
}

int dl_route_aggregate_bwd_phase2(const dl_graph* g, const void* Z, int K, int d, dl_dtype dtype, float beta,
                                  float t, const uint8_t* p, const float* a, const float* s, const float* dH,
                                  const float* dw, const float* dwr, const float* ds, float* dZ, int accumulate,
                                  void* ws, size_t ws_bytes, void* stream) {
    return bwd_phase2_impl(g, Z, K, d, dtype, beta, t, p, a, s, dH, dw, dwr, ds, accumulate ? dZ : nullptr, nullptr, dZ,
                           ws, ws_bytes, stream);
}

static int bwd_both_impl(const dl_graph* g, const void* Z, int K, int d, dl_dtype dtype, float beta, float t,
                         const uint8_t* p, const float* a, const float* s, const float* dH, const float* dz_in,
                         const float* scale, float* dZ, void* ws, size_t ws_bytes, void* stream) {
    DL_REQUIRE(g != nullptr, "graph is NULL");
    const dl_csr_plan* c = &g->csr;
    if (int rc = check_plan(c, "graph")) return rc;
    if (int rc = check_shape(K, d)) return rc;
    DL_REQUIRE(c->row_offset == 0 && c->n_rows == c->n_total,
               "dl_route_aggregate_bwd needs an unsharded plan; call the two phases with an all-gather of ds between");
    Workspace w;
    if (int rc = check_workspace(c, K, d, ws, ws_bytes, &w)) return rc;
    if (int rc = dl_route_aggregate_bwd_phase1(g, Z, K, d, dtype, beta, p, a, s, dH, w.dw, w.dwr, w.ds, ws, ws_bytes,
                                               stream))
        return rc;
    return bwd_phase2_impl(g, Z, K, d, dtype, beta, t, p, a, s, dH, w.dw, w.dwr, w.ds, dz_in, scale, dZ, ws, ws_bytes,
                           stream);
}

int dl_route_aggregate_bwd(const dl_graph* g, const void* Z, int K, int d, dl_dtype dtype, float beta, float t,
                           const uint8_t* p, const float* a, const float* s, const float* dH, float* dZ,
                           int accumulate, void* ws, size_t ws_bytes, void* stream) {
    return bwd_both_impl(g, Z, K, d, dtype, beta, t, p, a, s, dH, accumulate ? dZ : nullptr, nullptr, dZ, ws, ws_bytes,
                         stream);
}

int dl_route_aggregate_bwd_scaled(const dl_graph* g, const void* Z, int K, int d, dl_dtype dtype, float beta, float t,
                                  const uint8_t* p, const float* a, const float* s, const float* dH, const float* dZ_in,
                                  const float* scale, float* dZ, void* ws, size_t ws_bytes, void* stream) {
    return bwd_both_impl(g, Z, K, d, dtype, beta, t, p, a, s, dH, dZ_in, scale, dZ, ws, ws_bytes, stream);
}

}  // extern "C"
