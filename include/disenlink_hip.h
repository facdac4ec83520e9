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
 *   - every pointer is a DEVICE pointer.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Nothing here
 *     allocates, frees or synchronises; scratch comes from the caller (`ws`, sized by
 *     dl_workspace_bytes).  Launches are asynchronous on `stream`.
 *   - the library is linked WITHOUT a HIP runtime and binds to the one the host process uses,
 *     so the caller's streams and allocations are valid here.
 *   - return 0 on success, a negative DL_E_* code on error; dl_last_error() returns the
 *     message of the calling thread's last failing call.
 *   - layouts: Z, H are [n_total][K][d] row-major (== torch.cat(h_k, dim=1) of model.py:114) in the
 *     storage type named by the `dtype` argument (fp32 or bf16); dZ, dH are always fp32 of the same
 *     shape; indices int32; factor ids uint8; s is fp32 [n_total][K] RAW row sums (the zero -> 1
 *     substitution of model.py:72 is applied where s is read).
 *   - sharding: a plan may cover only rows [row_offset, row_offset + n_rows) of the n_total
 *     nodes (one shard per GPU).  Node-indexed arrays are always indexed by GLOBAL node id and
 *     only the plan's rows are written; per-entry arrays (p, a, ...) are local to the plan.
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

/* Storage type of the node tables Z and H.  Arithmetic, per-edge values, probabilities and every
 * gradient are fp32 in both cases; bf16 only halves the bytes of the gathered rows (tuned kernels
 * only; the reference has no bf16 path — parity is defined against the fp32 restatement). */
typedef enum dl_dtype { DL_F32 = 0, DL_BF16 = 1 } dl_dtype;

/* A CSR over (a shard of) the nodes plus the segment plan that balances skewed rows: every row is
 * cut into >= 1 segments of <= seg_len consecutive entries; one wavefront owns one segment and one
 * workgroup (DL_UNIT_SEGS = 4 wavefronts) serves DL_UNIT_SEGS consecutive POSITIONS of the seg_* arrays.
 *
 * Units.  The segments of a row (of one column slice of a row, see below) are grouped, counting from the
 * row's first segment, into UNITS of at most DL_UNIT_SEGS consecutive segments (the kernels recognise a unit
 * as a run of positions with the same row AND the same slot inside one group of DL_UNIT_SEGS positions).  A unit never straddles a
 * group of DL_UNIT_SEGS positions (units of 3 are padded to 4, units are stored largest first inside a
 * slice, every slice region is padded to a multiple of DL_UNIT_SEGS; seg_row = -1 marks a padding
 * position), so the workgroup that holds it sums it on chip, in segment order.  Only rows with more than
 * one unit reduce through partial slots in the workspace — one slot per unit, summed in slot (= entry)
 * order by a combine kernel.  How a row is cut depends on that row alone, so a row shard gives the same
 * bits as the whole graph.
 *
 * XCD-aware slicing (optional, n_slices = 8 on MI355X): the column space is cut into n_slices
 * node ranges of equal entry count and no segment spans two of them; positions are stored slice-major and
 * workgroup b serves slice b % n_slices.  Workgroups b and b+8 are observed to share an XCD, so
 * each XCD's 4 MiB L2 only ever gathers rows of "its" 1/8 of the node table.  Placement is a
 * speed matter only: results do not depend on it. */
#define DL_UNIT_SEGS 4
typedef struct dl_csr_plan {
    int32_t n_rows;             /* rows of this plan */
    int32_t row_offset;         /* global node id of row 0 */
    int32_t n_total;            /* global node count (extent of node-indexed arrays) */
    int32_t n_entries;
    const int32_t* rowptr;      /* [n_rows+1] */
    const int32_t* col;         /* [n_entries] global node ids */
    int32_t seg_len;
    int32_t n_seg;              /* segment POSITIONS, padding included (a multiple of DL_UNIT_SEGS per slice) */
    const int32_t* seg_row;     /* [n_seg] local row of the segment, -1 = padding position */
    const int32_t* seg_beg;     /* [n_seg] first entry of the segment */
    const int32_t* seg_end;     /* [n_seg] one past its last entry */
    const int32_t* seg_slot;    /* [n_seg] partial slot of the segment's UNIT, -1 if the row is a single unit */
    int32_t n_slices;           /* >= 1 */
    int32_t slice_max_seg;      /* largest number of positions in one slice stream (sizes the launch grid) */
    const int32_t* slice_seg0;  /* [n_slices+1] first position of each slice stream (multiples of DL_UNIT_SEGS) */
    int32_t n_multi;            /* rows with more than one unit */
    int32_t n_slots;            /* units belonging to such rows */
    const int32_t* multi_row;   /* [n_multi] local row */
    const int32_t* multi_slot0; /* [n_multi+1] first slot of each such row (slots are consecutive) */
    /* Optional pair (both NULL = rows of several units are always summed by a separate combine launch): with them the
     * kernels that know how sum such a row INSIDE the launch — every unit stores its partial slot, adds 1 to the row's
     * counter, and the unit whose add comes last adds the slots in the combine kernel's own order and writes the row
     * (same bits as the separate launch; no float atomics).  unit_count belongs to the plan: n_multi zero-initialised
     * int32 in device memory that only the library writes — the last unit of a row puts its counter back to 0, so the
     * array is all zero again when a launch has completed; one launch at a time per plan. */
    const int32_t* slot_multi;  /* [n_slots] index (into multi_row / multi_slot0) of the row a slot belongs to */
    int32_t* unit_count;        /* [n_multi] */
} dl_csr_plan;

/* The binarised, symmetrised training adjacency (main_disentangled.py:137-142): entries are the
 * directed non-zeros of adj_sym, col ascending inside a row, both directions present. */
typedef struct dl_graph {
    dl_csr_plan csr;
    /* Optional second segment plan over the SAME rowptr/col arrays, used by dl_route_fwd only (route.n_seg == 0
     * disables it).  It may be XCD-sliced (routing gathers whole Z rows, K*d*4 bytes per edge), and with
     * route_mirror != 0 it covers only the entries with col >= row: routing is symmetric — (i,j) and (j,i)
     * evaluate the same fma chain — so each undirected edge is computed once and written to both entries
     * through rev[e] = index of (col[e], row(e)).  Mirroring needs an unsharded plan. */
    dl_csr_plan route;
    const int32_t* rev;         /* [n_entries], needed when route_mirror != 0 */
    int32_t route_mirror;
} dl_graph;

/* A CSR over pair slots: row u lists other endpoints (csr.col) and pair ids (inc_pair).
 *   - as the backward's node-incidence list, every pair occupies two slots (one per endpoint;
 *     a pair (u,u) appears twice in row u);
 *   - as the forward's "pairs by first endpoint" list, every pair occupies one slot in row pu. */
typedef struct dl_pair_incidence {
    dl_csr_plan csr;
    const int32_t* inc_pair;    /* [csr.n_entries] */
    int32_t n_pairs;            /* extent of the prob / g_prob arrays */
    /* Optional (NULL = not given), read by dl_score_pairs_train only: the labels and loss weights of its `y` / `w`
     * arguments laid out PER ENTRY, [csr.n_entries][2] = (y[inc_pair[e]], +-w[inc_pair[e]]) — a coalesced stream instead of
     * two random 4-byte reads per entry.  The sign of the weight marks the entry that writes prob[]: + in the row of the
     * pair's FIRST endpoint (csr row == pu), - (also -0.0) in the other one, so every probability is written once.
     * Whoever sets it keeps it consistent with the y / w passed alongside (disenlink_amd/graph.py caches it per label /
     * weight tensor). */
    const float* entry_yw;
} dl_pair_incidence;

/* ---- host-side graph preparation (no GPU; the only entry points that allocate: malloc'd outputs are
 * released by the matching *_free).  Counterpart of the dense adjacency construction of
 * main_disentangled.py:137-142 for hosts without Python; disenlink_amd/graph.py builds identical arrays
 * with torch index ops on the device.  Upload the arrays and point a dl_csr_plan at them. */
typedef struct dl_host_csr {
    int32_t n_nodes;
    int32_t n_entries;
    int32_t* rowptr;            /* [n_nodes+1] */
    int32_t* col;               /* [n_entries], ascending inside a row */
    int32_t* rev;               /* [n_entries] */
} dl_host_csr;

typedef struct dl_host_plan {   /* the segment-plan fields of dl_csr_plan, in host memory */
    int32_t seg_len, n_seg, n_slices, slice_max_seg, n_multi, n_slots;
    int32_t *seg_row, *seg_beg, *seg_end, *seg_slot, *slice_seg0, *multi_row, *multi_slot0;
    int32_t* slot_multi;        /* [n_slots]; unit_count is not built here: n_multi zeroed int32 on the device */
} dl_host_plan;

/* Directed edge rows (duplicates allowed) -> CSR of the binarised adjacency; symmetrise != 0 reproduces
 * adj_sym = (adj + adj.T) != 0.  Fails if the result is not symmetric. */
int dl_host_csr_from_edges(const int64_t* src, const int64_t* dst, int64_t n_edge_rows, int32_t n_nodes,
                           int symmetrise, dl_host_csr* out);
void dl_host_csr_free(dl_host_csr* csr);

/* Segment plan of a CSR (see dl_csr_plan): segments of <= seg_len entries, never spanning two of the
 * n_col_slices column slices (1 = unsliced; a multiple of 8 = XCD streams x time), optionally only over
 * the entries with keep[e] != 0 (one contiguous run per row, e.g. col >= row for symmetric routing).
 * unit_segs = DL_UNIT_SEGS groups the segments of a row into units (plans whose kernels sum over a row);
 * unit_segs = 1 makes every segment its own unit, positions in entry order (routing plan, forward scorer).
 * by_length != 0 places the units of one size class by their number of entries, most first, instead of in entry order:
 * the wavefronts of a workgroup then finish together.  Use it when the gathered tables (2 * n_total * K * d * element
 * size) fit the 256 MiB Infinity Cache; where the row streams come from HBM, entry order is faster.  Results do not
 * depend on it. */
int dl_host_plan_build(int32_t n_rows, int32_t n_total, const int32_t* rowptr, const int32_t* col, int32_t seg_len,
                       int32_t n_col_slices, const uint8_t* keep, int32_t unit_segs, int32_t by_length,
                       dl_host_plan* out);
void dl_host_plan_free(dl_host_plan* plan);

const char* dl_version(void);
const char* dl_last_error(void);
/* The library's environment switches (measurement / test knobs, csrc/dl_config.h) are read once, at the first call that
 * needs one: no launch path touches the environment.  This reads them again (for a process that changes one). */
void dl_config_reload(void);

/* 1 if (K,d) runs on the tuned wavefront-tiled kernels, 0 if it falls back to the generic ones. */
int dl_has_fast_path(int K, int d);
int dl_has_fast_path_dtype(int K, int d, dl_dtype dtype);
/* Force the generic kernels (parity cross-check of the two implementations): 1 = on, 0 = off,
 * negative = query only.  Returns the previous value. */
int dl_set_force_generic(int on);

/* Scratch needed by the calls below for this plan and shape. */
size_t dl_workspace_bytes(const dl_csr_plan* plan, int K, int d);

/* Factor projection on the matrix cores: replaces model.py:13-15 / 24-27 fanned out at model.py:106.
 *   two-layer (Factor2):   Z[n][k][:] = W2[k] . relu(W1[k] . x[n] + b1[k]) + b2[k]
 *                          W1 [K][nhid][F], b1 [K][nhid], W2 [K][d][nhid], b2 [K][d]
 *   single layer (Factor): pass W2 = b2 = NULL, nhid = 1:  Z[n][k][:] = W1[k] . x[n] + b1[k],  W1 [K][d][F], b1 [K][d]
 * x is fp32 [N][F] row-major, Z fp32 [N][K][d].  d must be 32, 64 or 128 (dl_project_supported).
 * fp32 in, fp32 results.  With the workspace, both layers run on the bf16 matrix path at fp32-grade accuracy: x, W1,
 * W2 (once per call) and the hidden activations (in registers) are split into three bf16 planes each
 * (v = hi + mid + lo) and every term is the sum of six exact bf16 products in an fp32 accumulator; when ws is
 * NULL / too small or DL_PROJECT_FP32_MFMA=1 is set, everything is plain fp32 MFMA (v_mfma_f32_32x32x2_f32: an
 * exact k-ordered fmaf chain).
 * ws (optional, dl_project_fwd_workspace_bytes): the plane arrays, and on small graphs the partial sums of the
 * several workgroups per node tile that share the hidden layer (added in a fixed order); without it one workgroup
 * walks the whole hidden layer with fp32 MFMA — same result up to rounding / summation order, slower. */
int dl_project_supported(int d);
size_t dl_project_fwd_workspace_bytes(int N, int F, int K, int nhid, int d, int two_layer);
/* hid_out (optional, two-layer form, dl_project_hidden_floats(N, K, nhid) floats): keep the hidden layer
 * relu(W1 x + b1), laid out hidT [K][nhid][(N+3)&~3], for dl_project_bwd — worth its 8 bytes of traffic per
 * hidden unit from F of about 150 up; NULL = the backward recomputes it (no [N,K,nhid] memory at all). */
size_t dl_project_hidden_floats(int N, int K, int nhid);
int dl_project_fwd(const float* x, int N, int F, int K, int nhid, int d,
                   const float* W1, const float* b1, const float* W2, const float* b2,
                   float* Z, float* hid_out, void* ws, size_t ws_bytes, void* stream);

/* Persistent operand planes of the feature matrix (round 5).  model.py:106 evaluates the K MLPs on the SAME x every epoch
 * (main_disentangled.py:194): dl_project_fwd / dl_project_bwd split x (and x^T) into the three bf16 planes of the matrix
 * path inside ws on every call.  A caller that keeps x for a run builds them ONCE into a buffer of its own
 * (dl_project_xplanes_bytes; 16-byte aligned; tile-major planes of x, then of x^T) and passes it to the _xp forms, which
 * then skip those splits — same products, same bits.  The buffer is valid for exactly the (x contents, N, F) it was built
 * from; xplanes == NULL behaves like the plain entry points.  Two-layer form on the bf16 matrix path only (the single
 * layer and DL_PROJECT_FP32_MFMA=1 ignore it); graphs processed in node blocks re-split per block and ignore it too. */
size_t dl_project_xplanes_bytes(int N, int F);
int dl_project_xplanes_build(const float* x, int N, int F, void* xplanes, size_t xplanes_bytes, void* stream);
int dl_project_fwd_xp(const float* x, int N, int F, int K, int nhid, int d,
                      const float* W1, const float* b1, const float* W2, const float* b2,
                      float* Z, float* hid_out, void* ws, size_t ws_bytes, const void* xplanes, void* stream);
int dl_project_bwd_xp(const float* x, int N, int F, int K, int nhid, int d,
                      const float* W1, const float* b1, const float* W2, const float* dZ, const float* hid,
                      float* dW1, float* db1, float* dW2, float* db2,
                      void* ws, size_t ws_bytes, const void* xplanes, void* stream);

/* Backward of the projection: replaces autograd of model.py:13-15 / 24-27 under loss.backward()
 * (main_disentangled.py:198).  dZ fp32 [N][K][d] in; weight and bias gradients out, shaped like the weights
 * (dW1 like W1, db1 like b1, dW2 like W2, db2 like b2; single layer: W2 = dW2 = db2 = NULL, nhid = 1).
 * x is data and gets no gradient.  The hidden layer is recomputed on the matrix cores (never read from HBM);
 * ws needs dl_project_bwd_workspace_bytes(...) bytes (the masked hidden gradient of one node block — kept as the
 * three bf16 planes of its transpose, the operand of the dW1 contraction on the bf16 matrix path, fp32-grade like
 * the forward's layer 1 — the planes of x^T for that block, and the per-node-range partial slabs).  Sums over nodes are taken range by range in a fixed order (no float
 * atomics): the gradients are bitwise reproducible. */
size_t dl_project_bwd_workspace_bytes(int N, int F, int K, int nhid, int d, int two_layer);
int dl_project_bwd(const float* x, int N, int F, int K, int nhid, int d,
                   const float* W1, const float* b1, const float* W2, const float* dZ,
                   const float* hid /* hid_out of the forward, or NULL = recompute */,
                   float* dW1, float* db1, float* dW2, float* db2,
                   void* ws, size_t ws_bytes, void* stream);

/* Routing: replaces model.py:56-72 restricted to adj==1 entries.
 *   per edge e=(i,j):  sigma_k = z_k[i].z_k[j] / t ; e_k = exp(sigma_k) ; alpha_k = e_k / sum_k e_k
 *                      p[e] = argmax_k alpha_k (first max; NaN counts as max) ; a[e] = alpha_p
 *   per node:          s[i][k] = sum_{e in row i, p[e]=k} a[e]          (raw; model.py:70-71) */
int dl_route_fwd(const dl_graph* g, const void* Z, int K, int d, dl_dtype dtype, float t,
                 uint8_t* p, float* a, float* s, void* ws, size_t ws_bytes, void* stream);

/* Aggregation ("K-factor edge scatter"): replaces model.py:73-75.
 *   H[i][k] = beta*Z[i][k] + (1-beta) * sum_{e=(i,j), p[e]=k} a[e] / s~[j][k] * Z[j][k]
 *   with s~ = (s==0 ? 1 : s) and the normaliser taken at the NEIGHBOUR j (model.py:73 broadcast);
 *   s must hold the rows of every neighbour (all-gathered when sharded). */
int dl_aggregate_fwd(const dl_graph* g, const void* Z, int K, int d, dl_dtype dtype, float beta,
                     const uint8_t* p, const float* a, const float* s,
                     void* H, void* ws, size_t ws_bytes, void* stream);

/* Pair-list link scorer: replaces model.py:109-113 evaluated at the listed (u,v) only.
 *   prob[q] = sigmoid( sum_k (h_k[u].h_k[v]) * exp(z_k[u].z_k[v] / t) )    (raw exp, not softmax)
 * by_u (optional, may be NULL): the same pairs as a CSR by first endpoint (each pair once, inc_pair =
 * position in pu/pv/prob); lets a wavefront keep the u rows in LDS for a whole segment and, when
 * sliced, keeps the gathered v rows inside one XCD's L2.
 * coef (optional, may be NULL; training only): [2][n_pairs][K] — coef[0][q][k] = e_k = exp(z_k[u].z_k[v]/t)
 * and coef[1][q][k] = (h_k[u].h_k[v]) * e_k, the per-factor terms of the logit.  Handing them to
 * dl_score_pairs_bwd turns the backward into two plain weighted row gathers (no dot products). */
int dl_score_pairs_fwd(const void* Z, const void* H, int N, int K, int d, dl_dtype dtype, float t,
                       const int32_t* pu, const int32_t* pv, int n_pairs,
                       const dl_pair_incidence* by_u,
                       float* prob, float* coef, void* stream);

/* Dense scorer: replaces model.py:109-113 as written — prob[u][v] for ALL N*N ordered pairs (row-major
 * fp32 [N][N]), the link_pred the reference's caller indexes with dense masks (main_disentangled.py:195).
 * No pair list is materialised.  fp32 tables with d % 32 == 0 go to the matrix cores (two Gram products per
 * factor, six exact bf16 products per term from three bf16 planes per operand; only the tile pairs u <= v are computed and mirrored: prob is symmetric bit for bit);
 * other shapes use the vector kernels.  Its backward is dl_score_allpairs_bwd over the support of the caller's
 * loss masks.  N*N must stay below 2^31: N <= 46340.
 * ws (optional, dl_score_allpairs_workspace_bytes): the bf16 planes of Z and H, split once per call; without it
 * (NULL / too small) every tile pair splits the rows it stages — same result bit for bit, slower. */
size_t dl_score_allpairs_workspace_bytes(int N, int K, int d, dl_dtype dtype);
int dl_score_allpairs_fwd(const void* Z, const void* H, int N, int K, int d, dl_dtype dtype, float t,
                          float* prob, void* ws, size_t ws_bytes, void* stream);

/* Backward of dl_score_allpairs_fwd (autograd of model.py:109-113 + sigmoid under main_disentangled.py:195-198) on a
 * FIXED pair plan: the reference's caller takes its loss on link_pred[mask == 1] with masks built once per run
 * (main_disentangled.py:167-190), so d loss / d link_pred can be non-zero only on the masks' support.  The caller
 * hands that support over ONCE as a pair list (pu, pv) with its incidence plan `inc` (the one dl_score_pairs_bwd
 * takes; inc->n_pairs = n_pairs) and every step the dense prob [N][N] of the forward and the dense gradient
 * g_prob [N][N] autograd produced; the call reads both at the listed entries and writes
 *   dZ, dH = sum over the listed (u,v) of the scorer's backward terms (see dl_score_pairs_bwd),
 * for the plan's rows.  Entries of g_prob OUTSIDE the list are not read: the plan must cover every entry that can
 * carry gradient (keying the plan on the gradient's non-zero entries instead is wrong — a saturated positive,
 * p == 1.0 with y == 1, has exactly zero BCE gradient one epoch and a non-zero one the next).  Listed entries whose
 * gradient happens to be zero add exactly zero.  ws: dl_workspace_bytes(&inc->csr, K, d) — the gathered
 * prob / g_prob vectors live in it. */
int dl_score_allpairs_bwd(const void* Z, const void* H, int N, int K, int d, dl_dtype dtype, float t,
                          const dl_pair_incidence* inc, const int32_t* pu, const int32_t* pv, int n_pairs,
                          const float* prob, const float* g_prob, float* dZ, float* dH,
                          void* ws, size_t ws_bytes, void* stream);

/* Tie-averaged AUC of a score vector against FIXED labels: replaces sklearn.metrics.roc_auc_score at
 * main_disentangled.py:202-204 / 217-219 (validation AUC every epoch, test AUC at the end).  pos_idx / neg_idx
 * (int64, device) are the positions of the positive and negative labels in score, found once per run; the call
 * writes u2[0] = sum over positives p of ( 2 * #{negatives n: s_n < s_p} + #{n: s_n == s_p} ) as an exact 64-bit
 * integer, so AUC = u2 / (2 * n_pos * n_neg) — the Mann-Whitney statistic with tie-averaged ranks.  One launch:
 * the smaller class is cut into slices of 1,024 scores, a workgroup sorts its slice (bitonic network, shuffles +
 * LDS) and binary-searches its chunk of the other class in it; the counts of the slices add up.  Work grows as
 * ceil(min/1024) * max: dl_auc_pair_counts_supported says whether the sizes are in range (n_pos * n_neg <= 4e11;
 * beyond that a device sort on the caller's side is the better tool).  Scores are expected to be finite. */
int dl_auc_pair_counts_supported(int n_pos, int n_neg);
int dl_auc_pair_counts(const float* score, const int64_t* pos_idx, int n_pos, const int64_t* neg_idx, int n_neg,
                       unsigned long long* u2, void* stream);
/* The same, ADDING the counts to *u2 instead of overwriting it (no memset in front of the launch): for a caller that keeps
 * *u2 at zero between evaluations — dl_epoch_finish reads and clears it. */
int dl_auc_pair_counts_add(const float* score, const int64_t* pos_idx, int n_pos, const int64_t* neg_idx, int n_neg,
                           unsigned long long* u2, void* stream);

/* End-of-epoch bookkeeping of the training loop ON THE DEVICE (main_disentangled.py:199-214: loss and validation AUC of
 * the epoch, `if auc > best_auc: best_auc = auc; weights = deepcopy(state_dict); stale = 0 else stale += 1`, patience), in
 * one launch, so that the host need not read anything back before it launches the next epoch:
 *   auc = *u2 / denom2 in double (u2 = the counts of dl_auc_pair_counts[_add], denom2 = 2 n_pos n_neg; NaN if denom2 <= 0);
 *   if !stopped && auc > best_auc: best[i][:] = params[i][:] for the n_bufs <= DL_ADAM_MAX_BUFS buffers (numel[i] floats
 *   each: the weights AFTER the step, like :209), best_auc = auc, stale = 0, best_epoch = epoch;  else stale += 1;
 *   hist[2 epoch] = loss[0], hist[2 epoch + 1] = auc;  epoch += 1;  stale > patience: stopped = 1;  *u2 = 0.
 * Once stopped, or once max_epochs epochs are recorded, the call changes nothing but *u2 = 0 (epochs the host queued
 * before it saw the stop; the tail of a replayed graph that holds several epochs).
 * host_ring (or NULL): PINNED, device-accessible host memory of ring x 4 doubles — slot (epoch mod ring) receives
 *   { loss, auc, epoch + 1, unused } as well, for a host that reads the history behind an event without a copy.
 * state: dl_epoch_state_bytes() bytes owned by the caller, zero-initialised once (best_auc = 0 as at :189):
 *   { double best_auc; int64 stale, epoch, stopped, best_epoch; uint32 internal[2]; }
 * params / best / numel: HOST arrays like dl_adam_step's; loss, u2, hist, state: device memory. */
size_t dl_epoch_state_bytes(void);
int dl_epoch_finish(int n_bufs, const float* const* params, float* const* best, const size_t* numel, const float* loss,
                    unsigned long long* u2, double denom2, void* state, double* hist, long long max_epochs,
                    long long patience, double* host_ring, int ring, void* stream);

/* Pair-list loss of main_disentangled.py:195 and its gradient in one pass:
 *   loss[0] = sum_q w[q] * BCE(prob[q], y[q])       (log clamped at -100, like F.binary_cross_entropy)
 *   g[q]    = w[q] * (prob[q] - y[q]) / max(prob[q] (1 - prob[q]), 1e-12)       = dloss / dprob[q]
 * With w = 1/n_pos on the positive pairs and 1/(m n_neg) on the negative ones this is the reference's
 * BCE(pos) + BCE(neg)/m.  ws: at least 4352 bytes of scratch (1,024 partial sums, added in a fixed order). */
int dl_pair_bce(const float* prob, const float* y, const float* w, int n_pairs, float* loss, float* g,
                void* ws, size_t ws_bytes, void* stream);

/* The optimiser step of the training loop: torch.optim.Adam's update (main_disentangled.py:150 — weight decay added to
 * the gradient, bias-corrected first and second moments) over n_bufs <= DL_ADAM_MAX_BUFS contiguous fp32 buffers in one
 * launch:
 *   g' = g + weight_decay p;  m = m + (1 - beta1)(g' - m);  v = beta2 v + (1 - beta2) g'^2;
 *   p -= (lr / (1 - beta1^step)) * m / ( sqrt(v) / sqrt(1 - beta2^step) + eps )
 * params / grads / exp_avg / exp_avg_sq: HOST arrays of n_bufs device pointers, numel: host array of element counts.
 * state: 3 device floats owned by the caller, zero-initialised once: the step counter, the step size lr / (1 - beta1^step)
 * and sqrt(1 - beta2^step) — the call increments the counter ON THE DEVICE first (no host sync; a captured graph replays
 * it).  Hyper-parameters are doubles like torch's (the bias corrections are formed in double).  Same arithmetic as
 * torch's fused Adam up to rounding. */
#define DL_ADAM_MAX_BUFS 8
int dl_adam_step(int n_bufs, float* const* params, const float* const* grads, float* const* exp_avg,
                 float* const* exp_avg_sq, const size_t* numel, float* state,
                 double lr, double beta1, double beta2, double eps, double weight_decay, void* stream);
/* The same update with the step number counted by the CALLER (step = 1 for the first update): no counter launch in front
 * of the update — for loops that are not replayed from a graph.  The bias corrections are formed on the device with the
 * expressions dl_adam_step uses (the same bits for the same step); state (or NULL) receives {step, step size, sqrt(1 - beta2^step)}. */
int dl_adam_step_at(int n_bufs, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const size_t* numel, float* state, long long step,
                    double lr, double beta1, double beta2, double eps, double weight_decay, void* stream);

/* Backward of dl_score_pairs_fwd (autograd of model.py:109-113 + sigmoid, as triggered at
 * main_disentangled.py:198).  g_prob = dLoss/dprob per pair.  Writes dZ and dH for the plan's rows:
 *   gl = g_prob * prob * (1 - prob);  dH[u] += gl e_k H[v][k];  dZ[u] += gl (q_k e_k)/t Z[v][k]
 * coef: the array dl_score_pairs_fwd filled (n_pairs = inc->n_pairs), or NULL to recompute e and q. */
int dl_score_pairs_bwd(const void* Z, const void* H, int K, int d, dl_dtype dtype, float t,
                       const dl_pair_incidence* inc, const float* prob, const float* g_prob,
                       const float* coef, float* dZ, float* dH, void* ws, size_t ws_bytes, void* stream);

/* Training step of the scorer in ONE pass over the incidence plan: scorer forward (model.py:109-113), the weighted
 * BCE gradient of main_disentangled.py:195 (g = w (p - y) / max(p (1 - p), 1e-12), as dl_pair_bce computes it) and
 * the scorer backward (main_disentangled.py:198) together.  The wave that owns a node's pair slots has the
 * per-factor dot products of every entry, so it forms prob itself and accumulates dZ / dH from the partner rows it
 * just gathered: the partner rows are gathered once per direction (2 x n_pairs x 2 rows) instead of once for the
 * forward plus twice for the backward.  Writes prob[q] for every pair of the plan (for the loss VALUE — call
 * dl_pair_bce on it — and for the validation AUC) and dZ, dH for the plan's rows = d(sum_q w BCE(prob, y)) / d(Z, H).
 * Pairs with w = 0 (validation pairs riding along) contribute exactly nothing to the gradients.
 * Tuned (K, d) only: dl_score_pairs_train_supported; otherwise use dl_score_pairs_fwd / dl_pair_bce / _bwd. */
int dl_score_pairs_train_supported(const dl_pair_incidence* inc, int K, int d, dl_dtype dtype);
int dl_score_pairs_train(const void* Z, const void* H, int K, int d, dl_dtype dtype, float t,
                         const dl_pair_incidence* inc, const float* y, const float* w,
                         float* prob, float* dZ, float* dH, void* ws, size_t ws_bytes, void* stream);

/* Backward of aggregate + normaliser + routing softmax (autograd of model.py:56-75; argmax and
 * masks carry no gradient), SURVEY.md Appendix A.3, split at its one global dependency:
 *   phase 1:  dw[e] = (1-beta) dH[i][p].Z[j][p],  dwr[e] = (1-beta) dH[j][p].Z[i][p]  (reverse edge)
 *             ds[i][k] = -(sum_{e in row i, p=k} dwr[e] a[e]) / s~[i][k]^2          (0 where s == 0)
 *   phase 2:  da = dw/s~[j][p] + ds[i][p],  dar = dwr/s~[i][p] + ds[j][p]
 *             dZ[i] (+)= beta dH[i] + sum_e (1-beta) a/s~[i][p] dH[j][p]
 *                                   + sum_e sum_k (da+dar) a ([k==p]-alpha_k)/t Z[j][k]
 * dH and (for phase 2) ds must hold the rows of every neighbour (all-gathered when sharded).
 * dw, dwr are per-entry scratch of the caller ([n_entries] each).
 * dl_route_aggregate_bwd runs both phases back to back (single GPU). */
int dl_route_aggregate_bwd_phase1(const dl_graph* g, const void* Z, int K, int d, dl_dtype dtype, float beta,
                                  const uint8_t* p, const float* a, const float* s, const float* dH,
                                  float* dw, float* dwr, float* ds, void* ws, size_t ws_bytes, void* stream);
int dl_route_aggregate_bwd_phase2(const dl_graph* g, const void* Z, int K, int d, dl_dtype dtype, float beta, float t,
                                  const uint8_t* p, const float* a, const float* s, const float* dH,
                                  const float* dw, const float* dwr, const float* ds,
                                  float* dZ, int accumulate, void* ws, size_t ws_bytes, void* stream);
int dl_route_aggregate_bwd(const dl_graph* g, const void* Z, int K, int d, dl_dtype dtype, float beta, float t,
                           const uint8_t* p, const float* a, const float* s,
                           const float* dH, float* dZ, int accumulate,
                           void* ws, size_t ws_bytes, void* stream);

/* dZ = scale[0] * (dZ_in + the backward above applied to dH): dl_route_aggregate_bwd with the accumulated input read
 * from its own array (dZ_in: NULL = 0, may be dZ itself) and the result multiplied by a DEVICE scalar (scale: NULL = 1).
 * For a caller whose incoming gradients are g * dH and g * dZ_in with g = d(total)/d(loss) known only on the device
 * (autograd of main_disentangled.py:198 through a fused loss): the backward is linear, so the two scaling passes over
 * [N,K,d] arrays and their copies disappear.  scale[0] == 1 gives the bits of dl_route_aggregate_bwd. */
int dl_route_aggregate_bwd_scaled(const dl_graph* g, const void* Z, int K, int d, dl_dtype dtype, float beta, float t,
                                  const uint8_t* p, const float* a, const float* s,
                                  const float* dH, const float* dZ_in, const float* scale, float* dZ,
                                  void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DISENLINK_HIP_H */
