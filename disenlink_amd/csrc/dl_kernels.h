// Internal launch functions behind the C ABI (dl_api.hip dispatches between them).
#pragma once
#include <hip/hip_runtime.h>
#include "disenlink_hip.h"

namespace dl {

// generic: any K <= 64, any d (dl_generic.hip)
int generic_route_fwd(const dl_csr_plan* c, const float* Z, int K, int d, float t, uint8_t* p, float* a, float* s,
                      hipStream_t st);
int generic_aggregate_fwd(const dl_csr_plan* c, const float* Z, int K, int d, float beta, const uint8_t* p,
                          const float* a, const float* s, float* H, hipStream_t st);
int generic_score_pairs_fwd(const float* Z, const float* H, int K, int d, float t, const int32_t* pu,
                            const int32_t* pv, int P, float* prob, hipStream_t st);
int generic_score_pairs_bwd(const dl_pair_incidence* inc, const float* Z, const float* H, int K, int d, float t,
                            const float* prob, const float* g_prob, float* dZ, float* dH, hipStream_t st);
int generic_bwd_phase1(const dl_csr_plan* c, const float* Z, int K, int d, float beta, const uint8_t* p,
                       const float* a, const float* s, const float* dH, float* dw, float* dwr, float* ds,
                       hipStream_t st);
int generic_bwd_phase2(const dl_csr_plan* c, const float* Z, int K, int d, float beta, float t, const uint8_t* p,
                       const float* a, const float* s, const float* dH, const float* dw, const float* dwr,
                       const float* ds, const float* dz_in, const float* scale, float* dZ, hipStream_t st);

// tuned, per-(K, D, table type) instantiations (dl_fast.hip)
bool fast_supported(int K, int d, int dtype);
int fast_route_fwd(const dl_csr_plan* g, const dl_csr_plan* route, bool mirror, const int32_t* rev, const void* Z,
                   int K, int d, int dtype, float t, uint8_t* p, float* a, float* s, float* s_part, hipStream_t st);
int fast_aggregate_fwd(const dl_csr_plan* g, const void* Z, int K, int d, int dtype, float beta, const uint8_t* p,
                       const float* a, const float* s, void* H, float* h_part, hipStream_t st);
int fast_bwd_phase1(const dl_csr_plan* g, const void* Z, int K, int d, int dtype, float beta, const uint8_t* p,
                    const float* a, const float* s, const float* dH, float* dw, float* dwr, float* ds,
                    float* ds_part, hipStream_t st);
int fast_bwd_phase2(const dl_csr_plan* g, const void* Z, int K, int d, int dtype, float beta, float t,
                    const uint8_t* p, const float* a, const float* s, const float* dH, const float* dw,
                    const float* dwr, const float* ds, const float* dz_in, const float* scale, float* dZ, float* dz_part, hipStream_t st);
int fast_score_pairs_fwd(const dl_pair_incidence* by_u, const void* Z, const void* H, int K, int d, int dtype,
                         float t, float* prob, float* coef, hipStream_t st);
int fast_score_pairs_train(const dl_pair_incidence* inc, const void* Z, const void* H, int K, int d, int dtype, float t,
                           const float* y, const float* w, float* prob, float* dZ, float* dH, float* part, hipStream_t st);
int fast_score_pairs_bwd(const dl_pair_incidence* inc, const void* Z, const void* H, int K, int d, int dtype,
                         float t, const float* prob, const float* g_prob, const float* coef, float* dZ, float* dH,
                         float* part, hipStream_t st);

// prob / g_prob of the dense [N][N] scorer at the listed pairs (dl_generic.hip)
int gather_dense_pairs(const int32_t* pu, const int32_t* pv, int N, int P, const float* prob, const float* g_prob,
                       float* prob_q, float* g_q, hipStream_t st);

int fast_score_allpairs_fwd(const void* Z, const void* H, int N, int K, int d, int dtype, float t, float* prob,
                            hipStream_t st);
int generic_score_allpairs_fwd(const float* Z, const float* H, int N, int K, int d, float t, float* prob,
                               hipStream_t st);

// dense scorer on the matrix cores (dl_score_dense.hip): fp32 tables, d % 32 == 0
bool dense_mfma_supported(int d);
size_t dense_score_workspace_bytes(int N, int K, int d);
int dense_mfma_score_allpairs_fwd(const float* Z, const float* H, int N, int K, int d, float t, float* prob, void* ws, size_t ws_bytes,
                                  hipStream_t st);

// tie-averaged AUC counts (dl_metrics.hip)
bool auc_counts_supported(int n_pos, int n_neg);           // the smaller class fits the LDS
int auc_pair_counts(const float* score, const int64_t* pos_idx, int n_pos, const int64_t* neg_idx, int n_neg,
                    unsigned long long* u2, hipStream_t st, bool clear = true);

size_t epoch_state_bytes();
int epoch_finish(int n_bufs, const float* const* params, float* const* best, const size_t* numel, const float* loss,
                 unsigned long long* u2, double denom2, void* state, double* hist, long long max_epochs, long long patience,
                 double* host_ring, int ring, hipStream_t st);
int adam_step(int n_bufs, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
              const size_t* numel, float* state, double lr, double beta1, double beta2, double eps, double weight_decay,
              hipStream_t st, long long host_step = 0);

int pair_bce(const float* prob, const float* y, const float* w, int n, float* loss, float* g, float* partial,
             hipStream_t st);

// factor projection on the matrix cores (dl_project.hip)
bool project_supported(int d);
size_t project_fwd_workspace_bytes(int N, int F, int K, int nhid, int d, bool two_layer);
bool split_products();   // layer-1 / dW1 products from three bf16 planes per operand (off: DL_PROJECT_FP32_MFMA=1)
int project_fwd(const float* x, int N, int F, int K, int nhid, int d, const float* W1, const float* b1,
                const float* W2, const float* b2, float* Z, void* ws, size_t ws_bytes, float* hid_out, hipStream_t st,
                const void* xplanes = nullptr);
// persistent bf16 planes of x and x^T (x is constant for a run): built once, handed to project_fwd / project_bwd
size_t project_xplanes_bytes(int N, int F);
const void* project_xplanes_xT(const void* xplanes, int N, int F);
int project_xplanes_build(const float* x, int N, int F, void* xplanes, hipStream_t st);
// its backward (dl_project_bwd.hip): weight / bias gradients, W2 == nullptr for the single layer
size_t project_bwd_workspace_bytes(int N, int F, int K, int nhid, int d, bool two_layer);
int project_bwd(const float* x, int N, int F, int K, int nhid, int d, const float* W1, const float* b1,
                const float* W2, const float* dZ, const float* hid, float* dW1, float* db1, float* dW2, float* db2,
                void* ws, hipStream_t st, const void* xplanes = nullptr);

}  // namespace dl
