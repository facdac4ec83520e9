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

static int check_graph(const dl_graph* g) {
    DL_REQUIRE(g != nullptr, "graph is NULL");
    DL_REQUIRE(g->n_nodes >= 0 && g->n_edges >= 0, "negative graph size");
    if (g->n_nodes > 0) DL_REQUIRE(g->rowptr != nullptr, "graph.rowptr is NULL");
    if (g->n_edges > 0) DL_REQUIRE(g->col != nullptr && g->rev != nullptr, "graph.col/rev is NULL");
    return DL_OK;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Workspace layout (all 256-byte aligned): dw[E] | da[E] | s_part[n_slots*K] | h_part[n_slots*K*d]
struct Workspace {
    float *dw, *da, *s_part, *h_part;
    size_t bytes;
};

static Workspace carve(const dl_graph* g, int K, int d, void* ws) {
    Workspace w;
    const size_t e = align256((size_t)g->n_edges * sizeof(float));
    const size_t sp = align256((size_t)g->n_slots * K * sizeof(float));
    const size_t hp = align256((size_t)g->n_slots * K * d * sizeof(float));
    char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    w.dw = (float*)base;
    w.da = (float*)(base + e);
    w.s_part = (float*)(base + 2 * e);
    w.h_part = (float*)(base + 2 * e + sp);
    w.bytes = 2 * e + sp + hp + 256;
    return w;
}

static int check_workspace(const dl_graph* g, int K, int d, void* ws, size_t ws_bytes, Workspace* out) {
    *out = carve(g, K, d, ws);
    if (!ws || ws_bytes < out->bytes) {
        set_error("workspace too small: have %zu, need %zu (dl_workspace_bytes)", ws ? ws_bytes : (size_t)0, out->bytes);
        return DL_E_WORKSPACE;
    }
    return DL_OK;
}

static bool use_fast(const dl_graph* g, int K, int d) {
    return !g_force_generic && fast_supported(K, d) && g->seg_len > 0 && g->n_seg > 0 && g->seg_row && g->seg_beg &&
           g->seg_slot && (g->n_multi == 0 || (g->multi_row && g->multi_slot0));
}

}  // namespace dl

using namespace dl;

extern "C" {

const char* dl_version(void) { return "disenlink_hip 0.1 (gfx950)"; }
const char* dl_last_error(void) { return g_err; }

int dl_has_fast_path(int K, int d) { return fast_supported(K, d) ? 1 : 0; }

int dl_set_force_generic(int on) {
    int old = g_force_generic;
    g_force_generic = on ? 1 : 0;
    return old;
}

size_t dl_workspace_bytes(const dl_graph* g, int K, int d) {
    if (!g || K < 1 || d < 1) return 0;
    return carve(g, K, d, nullptr).bytes;
}

int dl_route_fwd(const dl_graph* g, const float* Z, int K, int d, float t, uint8_t* p, float* a, float* s,
                 void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_graph(g)) return rc;
    if (int rc = check_shape(K, d)) return rc;
    DL_REQUIRE(t != 0.0f, "temperature is 0");
    if (g->n_nodes == 0) return DL_OK;
    DL_REQUIRE(Z && s, "Z or s is NULL");
    if (g->n_edges > 0) DL_REQUIRE(p && a, "p or a is NULL");
    if (use_fast(g, K, d)) {
        Workspace w;
        if (int rc = check_workspace(g, K, d, ws, ws_bytes, &w)) return rc;
        return fast_route_fwd(g, Z, K, d, t, p, a, s, w.s_part, (hipStream_t)stream);
    }
    return generic_route_fwd(g, Z, K, d, t, p, a, s, (hipStream_t)stream);
}

int dl_aggregate_fwd(const dl_graph* g, const float* Z, int K, int d, float beta, const uint8_t* p,
                     const float* a, const float* s, float* H, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_graph(g)) return rc;
    if (int rc = check_shape(K, d)) return rc;
    if (g->n_nodes == 0) return DL_OK;
    DL_REQUIRE(Z && s && H, "Z, s or H is NULL");
    if (g->n_edges > 0) DL_REQUIRE(p && a, "p or a is NULL");
    if (use_fast(g, K, d)) {
        Workspace w;
        if (int rc = check_workspace(g, K, d, ws, ws_bytes, &w)) return rc;
        return fast_aggregate_fwd(g, Z, K, d, beta, p, a, s, H, w.h_part, (hipStream_t)stream);
    }
    return generic_aggregate_fwd(g, Z, K, d, beta, p, a, s, H, (hipStream_t)stream);
}

int dl_score_pairs_fwd(const float* Z, const float* H, int N, int K, int d, float t, const int32_t* pu,
                       const int32_t* pv, int n_pairs, float* prob, void* stream) {
    if (int rc = check_shape(K, d)) return rc;
    DL_REQUIRE(N >= 0 && n_pairs >= 0, "negative size");
    DL_REQUIRE(t != 0.0f, "temperature is 0");
    if (n_pairs == 0) return DL_OK;
    DL_REQUIRE(Z && H && pu && pv && prob, "NULL argument");
    return generic_score_pairs_fwd(Z, H, K, d, t, pu, pv, n_pairs, prob, (hipStream_t)stream);
}

int dl_score_pairs_bwd(const float* Z, const float* H, int N, int K, int d, float t,
                       const dl_pair_incidence* inc, const float* prob, const float* g_prob, float* dZ,
                       float* dH, void* stream) {
    if (int rc = check_shape(K, d)) return rc;
    DL_REQUIRE(inc != nullptr, "incidence is NULL");
    DL_REQUIRE(N >= 0 && inc->n_nodes == N, "incidence.n_nodes=%d != N=%d", inc->n_nodes, N);
    DL_REQUIRE(t != 0.0f, "temperature is 0");
    if (N == 0) return DL_OK;
    DL_REQUIRE(Z && H && dZ && dH && inc->inc_ptr, "NULL argument");
    if (inc->n_pairs > 0) DL_REQUIRE(inc->inc_other && inc->inc_pair && prob && g_prob, "NULL pair argument");
    return generic_score_pairs_bwd(Z, H, N, K, d, t, inc, prob, g_prob, dZ, dH, (hipStream_t)stream);
}

int dl_route_aggregate_bwd(const dl_graph* g, const float* Z, int K, int d, float beta, float t,
                           const uint8_t* p, const float* a, const float* s, const float* dH, float* dZ,
                           int accumulate, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_graph(g)) return rc;
    if (int rc = check_shape(K, d)) return rc;
    DL_REQUIRE(t != 0.0f, "temperature is 0");
    if (g->n_nodes == 0) return DL_OK;
    DL_REQUIRE(Z && s && dH && dZ, "NULL argument");
    if (g->n_edges > 0) DL_REQUIRE(p && a, "p or a is NULL");
    Workspace w;
    if (int rc = check_workspace(g, K, d, ws, ws_bytes, &w)) return rc;
    return generic_route_aggregate_bwd(g, Z, K, d, beta, t, p, a, s, dH, dZ, accumulate, w.dw, w.da,
                                       (hipStream_t)stream);
}

}  // extern "C"
