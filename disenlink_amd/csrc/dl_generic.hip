// Generic (any K <= 64, any d) kernels: one 64-lane wave per row / pair / node, runtime K and d.
// They are the fallback for shapes without a tuned instantiation and the second, independent
// implementation the tuned kernels are cross-checked against.  Arithmetic follows
// oracle/sparse_ref.py (the executable spec), which follows model.py:56-75,109-113.
#include "dl_common.h"
#include "dl_kernels.h"

namespace dl {
namespace generic {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = WAVES_PER_BLOCK * DL_WAVE;

// sigma_k = z_k[i].z_k[j] over d, lanes stride the d index; result identical in all lanes.
__device__ __forceinline__ float wave_dot(const float* __restrict__ x, const float* __restrict__ y, int d) {
    float part = 0.0f;
    for (int c = lane_id(); c < d; c += DL_WAVE) part = fmaf(x[c], y[c], part);
    return wave_allreduce_sum(part);
}

// Computes e_k (lane k keeps it), S = sum_k e_k (sequential in k, like the reference's sum over
// dim 0) and returns alpha_k in lane k.  Lanes >= K return 0.
__device__ __forceinline__ float edge_softmax(const float* __restrict__ zi, const float* __restrict__ zj,
                                              int K, int d, float t, float& mine_e) {
    const int lane = lane_id();
    float S = 0.0f;
    mine_e = 0.0f;
    for (int k = 0; k < K; ++k) {
        float ek = expf(wave_dot(zi + k * d, zj + k * d, d) / t);
        S += ek;
        if (lane == k) mine_e = ek;
    }
    return lane < K ? mine_e / S : 0.0f;
}

// first-max arg over lanes 0..K-1 (NaN beats everything); returns the winning lane, uniform.
__device__ __forceinline__ int wave_argmax_first(float v, int K) {
    const int lane = lane_id();
    float best = v;
    int idx = lane < K ? lane : DL_WAVE;        // lanes >= K never win
    if (lane >= K) best = -__builtin_inff();
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float ov = __shfl_xor(best, off, DL_WAVE);
        int oi = __shfl_xor(idx, off, DL_WAVE);
        bool take = (oi < DL_WAVE) && (idx >= DL_WAVE || beats(ov, best) || (!beats(best, ov) && oi < idx));
        if (take) { best = ov; idx = oi; }
    }
    return idx;
}

__global__ __launch_bounds__(BLOCK) void route_fwd_kernel(
    const float* __restrict__ Z, int N, int K, int d, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, float t, uint8_t* __restrict__ p, float* __restrict__ a,
    float* __restrict__ s) {
    const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = lane_id();
    const size_t stride = (size_t)K * d;
    const float* zi = Z + (size_t)row * stride;
    float s_acc = 0.0f;                          // lane k accumulates s_k
    const int beg = rowptr[row], end = rowptr[row + 1];
    for (int e = beg; e < end; ++e) {
        const float* zj = Z + (size_t)col[e] * stride;
        float mine_e;
        float alpha = edge_softmax(zi, zj, K, d, t, mine_e);
        int win = wave_argmax_first(alpha, K);
        float aw = __shfl(alpha, win, DL_WAVE);
        if (lane == win) s_acc += aw;
        if (lane == 0) { p[e] = (uint8_t)win; a[e] = aw; }
    }
    if (lane < K) s[(size_t)row * K + lane] = s_acc;
}

__global__ __launch_bounds__(BLOCK) void aggregate_fwd_kernel(
    const float* __restrict__ Z, int N, int K, int d, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, float beta, const uint8_t* __restrict__ p,
    const float* __restrict__ a, const float* __restrict__ s, float* __restrict__ H) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6;
    const int row = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (row >= N) return;
    const int lane = lane_id();
    const int KD = K * d;
    float* acc = lds + (size_t)wave * KD;
    for (int x = lane; x < KD; x += DL_WAVE) acc[x] = 0.0f;
    const int beg = rowptr[row], end = rowptr[row + 1];
    for (int e = beg; e < end; ++e) {
        const int j = col[e];
        const int k = p[e];
        const float w = a[e] / one_if_zero(s[(size_t)j * K + k]);
        const float* zj = Z + (size_t)j * KD + k * d;
        for (int c = lane; c < d; c += DL_WAVE) acc[k * d + c] = fmaf(w, zj[c], acc[k * d + c]);
    }
    const float* zi = Z + (size_t)row * KD;
    float* hi = H + (size_t)row * KD;
    const float omb = 1.0f - beta;
    for (int x = lane; x < KD; x += DL_WAVE) hi[x] = beta * zi[x] + omb * acc[x];
}

__global__ __launch_bounds__(BLOCK) void score_pairs_fwd_kernel(
    const float* __restrict__ Z, const float* __restrict__ H, int K, int d, float t,
    const int32_t* __restrict__ pu, const int32_t* __restrict__ pv, int P, float* __restrict__ prob) {
    const int q = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (q >= P) return;
    const size_t stride = (size_t)K * d;
    const float* zu = Z + (size_t)pu[q] * stride;
    const float* zv = Z + (size_t)pv[q] * stride;
    const float* hu = H + (size_t)pu[q] * stride;
    const float* hv = H + (size_t)pv[q] * stride;
    float logit = 0.0f;
    for (int k = 0; k < K; ++k) {
        float qk = wave_dot(hu + k * d, hv + k * d, d);
        float ek = expf(wave_dot(zu + k * d, zv + k * d, d) / t);
        logit += qk * ek;                          // separate mul and add, like (q*e).sum(0)
    }
    if (lane_id() == 0) prob[q] = sigmoid_ref(logit);
}

// dH[u] = sum_inc gl*e_k*h_k[v],  dZ[u] = sum_inc gl*q_k*e_k/t*z_k[v]  over the pair slots of node u.
__global__ __launch_bounds__(BLOCK) void score_pairs_bwd_kernel(
    const float* __restrict__ Z, const float* __restrict__ H, int N, int K, int d, float t,
    const int32_t* __restrict__ inc_ptr, const int32_t* __restrict__ inc_other,
    const int32_t* __restrict__ inc_pair, const float* __restrict__ prob,
    const float* __restrict__ g_prob, float* __restrict__ dZ, float* __restrict__ dH) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6;
    const int u = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (u >= N) return;
    const int lane = lane_id();
    const int KD = K * d;
    float* accZ = lds + (size_t)wave * 2 * KD;
    float* accH = accZ + KD;
    for (int x = lane; x < KD; x += DL_WAVE) { accZ[x] = 0.0f; accH[x] = 0.0f; }
    const float* zu = Z + (size_t)u * KD;
    const float* hu = H + (size_t)u * KD;
    for (int it = inc_ptr[u]; it < inc_ptr[u + 1]; ++it) {
        const int v = inc_other[it];
        const int q = inc_pair[it];
        const float pr = prob[q];
        const float gl = g_prob[q] * pr * (1.0f - pr);      // sigmoid backward p(1-p)
        const float* zv = Z + (size_t)v * KD;
        const float* hv = H + (size_t)v * KD;
        for (int k = 0; k < K; ++k) {
            float qk = wave_dot(hu + k * d, hv + k * d, d);
            float ek = expf(wave_dot(zu + k * d, zv + k * d, d) / t);
            float ch = gl * ek;
            float cz = gl * qk * ek / t;
            for (int c = lane; c < d; c += DL_WAVE) {
                accH[k * d + c] = fmaf(ch, hv[k * d + c], accH[k * d + c]);
                accZ[k * d + c] = fmaf(cz, zv[k * d + c], accZ[k * d + c]);
            }
        }
    }
    for (int x = lane; x < KD; x += DL_WAVE) {
        dZ[(size_t)u * KD + x] = accZ[x];
        dH[(size_t)u * KD + x] = accH[x];
    }
}

// B1: dw[e] = (1-beta) * dH[i][p].Z[j][p]
__global__ __launch_bounds__(BLOCK) void bwd_dw_kernel(
    const float* __restrict__ Z, const float* __restrict__ dH, int N, int K, int d,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, float beta,
    const uint8_t* __restrict__ p, float* __restrict__ dw) {
    const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= N) return;
    const size_t stride = (size_t)K * d;
    const float omb = 1.0f - beta;
    for (int e = rowptr[row]; e < rowptr[row + 1]; ++e) {
        const int k = p[e];
        float v = wave_dot(dH + (size_t)row * stride + k * d, Z + (size_t)col[e] * stride + k * d, d);
        if (lane_id() == 0) dw[e] = omb * v;
    }
}

// B2: ds_k[i] = -(1/s~^2) sum_{e in row i} [p[rev e]=k] dw[rev e] a[rev e]   (0 where raw s == 0)
//     da[e]  = dw[e]/s~[j][p] + ds_p[i]
__global__ __launch_bounds__(BLOCK) void bwd_da_kernel(
    int N, int K, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ rev, const uint8_t* __restrict__ p, const float* __restrict__ a,
    const float* __restrict__ s, const float* __restrict__ dw, float* __restrict__ da) {
    const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = lane_id();
    const int beg = rowptr[row], end = rowptr[row + 1];
    float acc = 0.0f;                            // lane k: sum for factor k, edge order
    for (int e = beg; e < end; ++e) {
        const int r = rev[e];
        if (lane == p[r]) acc += dw[r] * a[r];
    }
    float ds = 0.0f;
    if (lane < K) {
        float sr = s[(size_t)row * K + lane];
        ds = sr == 0.0f ? 0.0f : -acc / (sr * sr);
    }
    for (int base = beg; base < end; base += DL_WAVE) {     // uniform trip count: the shuffle needs all lanes
        const int e = base + lane;
        const bool live = e < end;
        const int k = live ? p[e] : 0;
        const float dsk = __shfl(ds, k, DL_WAVE);
        if (live) da[e] = dw[e] / one_if_zero(s[(size_t)col[e] * K + k]) + dsk;
    }
}

// B3: dZ[i] (+)= beta*dH[i] + sum_e (1-beta) a[rev e]/s~[i][p_rev] dH[j][p_rev]
//                           + sum_e sum_k (da[e]+da[rev e]) a[e] ([k==p[e]]-alpha_k)/t * Z[j][k]
__global__ __launch_bounds__(BLOCK) void bwd_dz_kernel(
    const float* __restrict__ Z, const float* __restrict__ dH, int N, int K, int d,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ rev, float beta, float t, const uint8_t* __restrict__ p,
    const float* __restrict__ a, const float* __restrict__ s, const float* __restrict__ da,
    float* __restrict__ dZ, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6;
    const int row = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (row >= N) return;
    const int lane = lane_id();
    const int KD = K * d;
    float* acc = lds + (size_t)wave * KD;
    const float* zi = Z + (size_t)row * KD;
    const float* dhi = dH + (size_t)row * KD;
    for (int x = lane; x < KD; x += DL_WAVE) acc[x] = beta * dhi[x];
    const float omb = 1.0f - beta;
    for (int e = rowptr[row]; e < rowptr[row + 1]; ++e) {
        const int j = col[e];
        const int r = rev[e];
        const float* zj = Z + (size_t)j * KD;
        {   // aggregation term seen from the neighbour's row: edge r = (j -> row)
            const int kr = p[r];
            const float w = omb * a[r] / one_if_zero(s[(size_t)row * K + kr]);
            const float* dhj = dH + (size_t)j * KD + kr * d;
            for (int c = lane; c < d; c += DL_WAVE) acc[kr * d + c] = fmaf(w, dhj[c], acc[kr * d + c]);
        }
        float mine_e;
        const float alpha = edge_softmax(zi, zj, K, d, t, mine_e);
        const int pe = p[e];
        const float cc = (da[e] + da[r]) * a[e];
        const float ck_mine = cc * ((lane == pe ? 1.0f : 0.0f) - alpha) / t;
        for (int k = 0; k < K; ++k) {
            const float ck = __shfl(ck_mine, k, DL_WAVE);
            for (int c = lane; c < d; c += DL_WAVE) acc[k * d + c] = fmaf(ck, zj[k * d + c], acc[k * d + c]);
        }
    }
    float* out = dZ + (size_t)row * KD;
    for (int x = lane; x < KD; x += DL_WAVE) out[x] = (accumulate ? out[x] : 0.0f) + acc[x];
}

static inline unsigned blocks_for(int n) { return (unsigned)((n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK); }

}  // namespace generic

using namespace generic;

// The generic kernels keep one K*d fp32 accumulator row per wave in LDS.
static int check_lds(int K, int d, int rows_per_wave) {
    size_t bytes = (size_t)WAVES_PER_BLOCK * rows_per_wave * K * d * sizeof(float);
    if (bytes > 64 * 1024) {
        set_error("generic kernels need %zu B of LDS for K=%d d=%d (limit 65536); use a shape with a tuned path",
                  bytes, K, d);
        return DL_E_ARG;
    }
    return DL_OK;
}

int generic_route_fwd(const dl_graph* g, const float* Z, int K, int d, float t, uint8_t* p, float* a,
                      float* s, hipStream_t st) {
    if (g->n_nodes == 0) return DL_OK;
    hipLaunchKernelGGL(route_fwd_kernel, dim3(blocks_for(g->n_nodes)), dim3(BLOCK), 0, st, Z, g->n_nodes, K, d,
                       g->rowptr, g->col, t, p, a, s);
    return check_launch("route_fwd(generic)");
}

int generic_aggregate_fwd(const dl_graph* g, const float* Z, int K, int d, float beta, const uint8_t* p,
                          const float* a, const float* s, float* H, hipStream_t st) {
    if (g->n_nodes == 0) return DL_OK;
    if (int rc = check_lds(K, d, 1)) return rc;
    size_t lds = (size_t)WAVES_PER_BLOCK * K * d * sizeof(float);
    hipLaunchKernelGGL(aggregate_fwd_kernel, dim3(blocks_for(g->n_nodes)), dim3(BLOCK), lds, st, Z, g->n_nodes, K,
                       d, g->rowptr, g->col, beta, p, a, s, H);
    return check_launch("aggregate_fwd(generic)");
}

int generic_score_pairs_fwd(const float* Z, const float* H, int K, int d, float t, const int32_t* pu,
                            const int32_t* pv, int P, float* prob, hipStream_t st) {
    if (P == 0) return DL_OK;
    hipLaunchKernelGGL(score_pairs_fwd_kernel, dim3(blocks_for(P)), dim3(BLOCK), 0, st, Z, H, K, d, t, pu, pv, P,
                       prob);
    return check_launch("score_pairs_fwd(generic)");
}

int generic_score_pairs_bwd(const float* Z, const float* H, int N, int K, int d, float t,
                            const dl_pair_incidence* inc, const float* prob, const float* g_prob, float* dZ,
                            float* dH, hipStream_t st) {
    if (N == 0) return DL_OK;
    if (int rc = check_lds(K, d, 2)) return rc;
    size_t lds = (size_t)WAVES_PER_BLOCK * 2 * K * d * sizeof(float);
    hipLaunchKernelGGL(score_pairs_bwd_kernel, dim3(blocks_for(N)), dim3(BLOCK), lds, st, Z, H, N, K, d, t,
                       inc->inc_ptr, inc->inc_other, inc->inc_pair, prob, g_prob, dZ, dH);
    return check_launch("score_pairs_bwd(generic)");
}

int generic_route_aggregate_bwd(const dl_graph* g, const float* Z, int K, int d, float beta, float t,
                                const uint8_t* p, const float* a, const float* s, const float* dH, float* dZ,
                                int accumulate, float* dw, float* da, hipStream_t st) {
    const int N = g->n_nodes;
    if (N == 0) return DL_OK;
    if (int rc = check_lds(K, d, 1)) return rc;
    size_t lds = (size_t)WAVES_PER_BLOCK * K * d * sizeof(float);
    hipLaunchKernelGGL(bwd_dw_kernel, dim3(blocks_for(N)), dim3(BLOCK), 0, st, Z, dH, N, K, d, g->rowptr, g->col,
                       beta, p, dw);
    hipLaunchKernelGGL(bwd_da_kernel, dim3(blocks_for(N)), dim3(BLOCK), 0, st, N, K, g->rowptr, g->col, g->rev, p,
                       a, s, dw, da);
    hipLaunchKernelGGL(bwd_dz_kernel, dim3(blocks_for(N)), dim3(BLOCK), lds, st, Z, dH, N, K, d, g->rowptr, g->col,
                       g->rev, beta, t, p, a, s, da, dZ, accumulate);
    return check_launch("route_aggregate_bwd(generic)");
}

}  // namespace dl
