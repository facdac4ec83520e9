/*
 * disenlink_hip.h — C ABI of libdisenlink_hip.so (MI355X / gfx950).
 *
 * The reference (sjz5202/DisenLink) has no FFI / plugin interface: its hot path is the ATen op
 * sequence inside model.py.  This header is the boundary a maintainer would bind instead of
 * those op sequences; each entry point cites the reference lines it replaces.  INTEGRATION.md
 * shows the ctypes stub.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Nothing here
 *     allocates, frees or synchronises; scratch comes from the caller (`ws`, sized by
 *     dl_workspace_bytes).  Launches are asynchronous on `stream`.
 *   - return 0 on success, a negative DL_E_* code on error; dl_last_error() returns the
 *     message of the calling thread's last failing call.
 *   - layouts: Z, H, dZ, dH are fp32 [N][K][d] row-major (== torch.cat(h_k, dim=1) of
 *     model.py:114); indices int32; factor ids uint8; s is [N][K] RAW row sums (the
 *     zero -> 1 substitution of model.py:72 is applied where s is read).
 *   - results do not depend on launch order / placement; no float atomics are used, so every
 *     entry point is bitwise reproducible run to run.
 */
#ifndef DISENLINK_HIP_H
#define DISENLINK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DL_OK            0
#define DL_E_ARG        -1   /* null pointer / bad size / unsupported K or d */
#define DL_E_LAUNCH     -2   /* hipLaunch / hipGetLastError failure */
#define DL_E_WORKSPACE  -3   /* workspace missing or too small */

#define DL_MAX_FACTORS  64   /* K <= 64 */

/* CSR of the binarised, symmetrised training adjacency (main_disentangled.py:137-142) plus the
 * row-segment plan used to balance skewed degrees.  All arrays live on the device. */
typedef struct dl_graph {
    int32_t n_nodes;
    int32_t n_edges;            /* directed non-zeros of adj_sym (both directions present) */
    const int32_t* rowptr;      /* [n_nodes+1] */
    const int32_t* col;         /* [n_edges], ascending inside a row */
    const int32_t* rev;         /* [n_edges], rev[e] = index of the edge (col[e], row(e)) */
    /* segment plan: every row is cut into >=1 segments of <= seg_len consecutive edges */
    int32_t seg_len;
    int32_t n_seg;
    const int32_t* seg_row;     /* [n_seg] */
    const int32_t* seg_beg;     /* [n_seg] first edge of the segment */
    const int32_t* seg_slot;    /* [n_seg] partial-sum slot of the segment, -1 if its row has one segment */
    const int32_t* row_seg0;    /* [n_nodes+1] first segment of each row */
    int32_t n_multi;
    int32_t n_slots;            /* segments that belong to multi-segment rows */
    const int32_t* multi_row;   /* [n_multi] rows with more than one segment */
    const int32_t* multi_slot0; /* [n_multi+1] first slot of each such row (slots are consecutive) */
} dl_graph;

/* Node-incidence list of a scored pair list: for node u, entries inc_ptr[u]..inc_ptr[u+1]-1 name
 * the other endpoint and the pair id of every pair slot u occupies (a pair (u,u) appears twice). */
typedef struct dl_pair_incidence {
    int32_t n_nodes;
    int32_t n_pairs;
    const int32_t* inc_ptr;     /* [n_nodes+1] */
    const int32_t* inc_other;   /* [2*n_pairs] */
    const int32_t* inc_pair;    /* [2*n_pairs] */
} dl_pair_incidence;

const char* dl_version(void);
const char* dl_last_error(void);

/* 1 if (K,d) runs on the tuned wavefront-tiled kernels, 0 if it falls back to the generic ones. */
int dl_has_fast_path(int K, int d);
/* Force the generic kernels (parity cross-check of the two implementations).  Returns old value. */
int dl_set_force_generic(int on);

/* Scratch needed by the calls below for this graph and shape. */
size_t dl_workspace_bytes(const dl_graph* g, int K, int d);

/* Routing: replaces model.py:56-72 restricted to adj==1 entries.
 *   per edge e=(i,j):  sigma_k = z_k[i].z_k[j] / t ; e_k = exp(sigma_k) ; alpha_k = e_k / sum_k e_k
 *                      p[e] = argmax_k alpha_k (first max; NaN counts as max) ; a[e] = alpha_p
 *   per node:          s[i][k] = sum_{e in row i, p[e]=k} a[e]          (raw; model.py:70-71) */
int dl_route_fwd(const dl_graph* g, const float* Z, int K, int d, float t,
                 uint8_t* p, float* a, float* s, void* ws, size_t ws_bytes, void* stream);

/* Aggregation ("K-factor edge scatter"): replaces model.py:73-75.
 *   H[i][k] = beta*Z[i][k] + (1-beta) * sum_{e=(i,j), p[e]=k} a[e] / s~[j][k] * Z[j][k]
 *   with s~ = (s==0 ? 1 : s) and the normaliser taken at the NEIGHBOUR j (model.py:73 broadcast). */
int dl_aggregate_fwd(const dl_graph* g, const float* Z, int K, int d, float beta,
                     const uint8_t* p, const float* a, const float* s,
                     float* H, void* ws, size_t ws_bytes, void* stream);

/* Pair-list link scorer: replaces model.py:109-113 evaluated at the listed (u,v) only.
 *   prob[q] = sigmoid( sum_k (h_k[u].h_k[v]) * exp(z_k[u].z_k[v] / t) )    (raw exp, not softmax) */
int dl_score_pairs_fwd(const float* Z, const float* H, int N, int K, int d, float t,
                       const int32_t* pu, const int32_t* pv, int n_pairs,
                       float* prob, void* stream);

/* Backward of dl_score_pairs_fwd (autograd of model.py:109-113 + sigmoid, as triggered at
 * main_disentangled.py:198).  g_prob = dLoss/dprob per pair.  Writes dZ and dH for all N rows. */
int dl_score_pairs_bwd(const float* Z, const float* H, int N, int K, int d, float t,
                       const dl_pair_incidence* inc, const float* prob, const float* g_prob,
                       float* dZ, float* dH, void* stream);

/* Backward of aggregate + normaliser + routing softmax (autograd of model.py:56-75; argmax and
 * masks carry no gradient).  dZ_out = (accumulate ? dZ_out : 0) + d/dZ.  SURVEY.md Appendix A.3. */
int dl_route_aggregate_bwd(const dl_graph* g, const float* Z, int K, int d, float beta, float t,
                           const uint8_t* p, const float* a, const float* s,
                           const float* dH, float* dZ, int accumulate,
                           void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DISENLINK_HIP_H */
