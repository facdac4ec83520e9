// Generic (any K <= 64, any d) kernels: one 64-lane wave per row / pair / node, runtime K and d.
// They are the fallback for shapes without a tuned instantiation and the second, independent
// implementation the tuned kernels are cross-checked against.  Arithmetic follows
// oracle/sparse_ref.py (the executable spec), which follows model.py:56-75,109-113.
#include "dl_common.h"
#include "dl_kernels.h"

namespace dl {
namespace generic {

// sigma_k = z_k[i].z_k[j] over d, lanes stride the d index; result identical in all lanes.
__device__ __forceinline__ float wave_dot(const float* __restrict__ x, const float* __restrict__ y, int d) {
    float part = 0.0f;
    for (int c = lane_id(); c < d; c += DL_WAVE) part = fmaf(x[c], y[c], part);
    return wave_allreduce_sum(part);
}

// Computes e_k (lane k keeps it), S = sum_k e_k (sequential in k, like the reference's sum over
// dim 0) and returns alpha_k in lane k.  Lanes >= K return 0.
__device__ __forceinline__ float edge_softmax(const float* __restrict__ zi, const float* __restrict__ zj,
                                              int K, int d, float t) {
    const int lane = lane_id();
    float S = 0.0f, mine_e = 0.0f;
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

__global__ __launch_bounds__(BLOCK) void route_fwd_kernel(dl_csr_plan c, const float* __restrict__ Z, int K, int d,
                                                          float t, uint8_t* __restrict__ p, float* __restrict__ a,
                                                          float* __restrict__ s) {
    const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= c.n_rows) return;
    const int lane = lane_id();
    const size_t stride = (size_t)K * d;
    const size_t grow = (size_t)row + c.row_offset;
    const float* zi = Z + grow * stride;
    float s_acc = 0.0f;                          // lane k accumulates s_k
    for (int e = c.rowptr[row]; e < c.rowptr[row + 1]; ++e) {
        const float* zj = Z + (size_t)c.col[e] * stride;
        float alpha = edge_softmax(zi, zj, K, d, t);
        int win = wave_argmax_first(alpha, K);
        float aw = __shfl(alpha, win, DL_WAVE);
        if (lane == win) s_acc += aw;
        if (lane == 0) { p[e] = (uint8_t)win; a[e] = aw; }
    }
    if (lane < K) s[grow * K + lane] = s_acc;
}

__global__ __launch_bounds__(BLOCK) void aggregate_fwd_kernel(dl_csr_plan c, const float* __restrict__ Z, int K,
                                                              int d, float beta, const uint8_t* __restrict__ p,
                                                              const float* __restrict__ a,
                                                              const float* __restrict__ s, float* __restrict__ H) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6;
    const int row = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (row >= c.n_rows) return;
    const int lane = lane_id();
    const int KD = K * d;
    const size_t grow = (size_t)row + c.row_offset;
    float* acc = lds + (size_t)wave * KD;
    for (int x = lane; x < KD; x += DL_WAVE) acc[x] = 0.0f;
    for (int e = c.rowptr[row]; e < c.rowptr[row + 1]; ++e) {
        const int j = c.col[e];
        const int k = p[e];
        const float w = a[e] / one_if_zero(s[(size_t)j * K + k]);
        const float* zj = Z + (size_t)j * KD + k * d;
        for (int x = lane; x < d; x += DL_WAVE) acc[k * d + x] = fmaf(w, zj[x], acc[k * d + x]);
    }
    const float* zi = Z + grow * KD;
    float* hi = H + grow * KD;
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

// dense [N][N] scorer: one wave per (u, v) entry, row-major
__global__ __launch_bounds__(BLOCK) void score_allpairs_fwd_kernel(const float* __restrict__ Z,
                                                                   const float* __restrict__ H, int N, int K, int d,
                                                                   float t, float* __restrict__ prob) {
    const long long q = (long long)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (q >= (long long)N * N) return;
    const size_t stride = (size_t)K * d;
    const size_t u = (size_t)(q / N), v = (size_t)(q % N);
    float logit = 0.0f;
    for (int k = 0; k < K; ++k) {
        float qk = wave_dot(H + u * stride + k * d, H + v * stride + k * d, d);
        float ek = expf(wave_dot(Z + u * stride + k * d, Z + v * stride + k * d, d) / t);
        logit += qk * ek;
    }
    if (lane_id() == 0) prob[q] = sigmoid_ref(logit);
}

// dH[u] = sum_inc gl*e_k*h_k[v],  dZ[u] = sum_inc gl*q_k*e_k/t*z_k[v]  over the pair slots of node u.
__global__ __launch_bounds__(BLOCK) void score_pairs_bwd_kernel(
    dl_csr_plan c, const int32_t* __restrict__ inc_pair, const float* __restrict__ Z, const float* __restrict__ H,
    int K, int d, float t, const float* __restrict__ prob, const float* __restrict__ g_prob,
    float* __restrict__ dZ, float* __restrict__ dH) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6;
    const int row = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (row >= c.n_rows) return;
    const int lane = lane_id();
    const int KD = K * d;
    const size_t u = (size_t)row + c.row_offset;
    float* accZ = lds + (size_t)wave * 2 * KD;
    float* accH = accZ + KD;
    for (int x = lane; x < KD; x += DL_WAVE) { accZ[x] = 0.0f; accH[x] = 0.0f; }
    const float* zu = Z + u * KD;
    const float* hu = H + u * KD;
    for (int it = c.rowptr[row]; it < c.rowptr[row + 1]; ++it) {
        const int v = c.col[it];
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
            for (int x = lane; x < d; x += DL_WAVE) {
                accH[k * d + x] = fmaf(ch, hv[k * d + x], accH[k * d + x]);
                accZ[k * d + x] = fmaf(cz, zv[k * d + x], accZ[k * d + x]);
            }
        }
    }
    for (int x = lane; x < KD; x += DL_WAVE) {
        dZ[u * KD + x] = accZ[x];
        dH[u * KD + x] = accH[x];
    }
}

// phase 1: dw[e], dwr[e] and ds[i][k] (see include/disenlink_hip.h)
__global__ __launch_bounds__(BLOCK) void bwd_phase1_kernel(dl_csr_plan c, const float* __restrict__ Z,
                                                           const float* __restrict__ dH, int K, int d, float beta,
                                                           const uint8_t* __restrict__ p,
                                                           const float* __restrict__ a,
                                                           const float* __restrict__ s, float* __restrict__ dw,
                                                           float* __restrict__ dwr, float* __restrict__ ds) {
    const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= c.n_rows) return;
    const int lane = lane_id();
    const size_t stride = (size_t)K * d;
    const size_t grow = (size_t)row + c.row_offset;
    const float omb = 1.0f - beta;
    float acc = 0.0f;                            // lane k: sum_e [p=k] dwr*a, edge order
    for (int e = c.rowptr[row]; e < c.rowptr[row + 1]; ++e) {
        const int k = p[e];
        const size_t j = (size_t)c.col[e];
        const float v = omb * wave_dot(dH + grow * stride + k * d, Z + j * stride + k * d, d);
        const float vr = omb * wave_dot(dH + j * stride + k * d, Z + grow * stride + k * d, d);
        if (lane == 0) { dw[e] = v; dwr[e] = vr; }
        if (lane == k) acc += vr * a[e];
    }
    if (lane < K) ds[grow * K + lane] = ds_from_acc(acc, s[grow * K + lane]);
}

// phase 2: dZ[i] (+)= beta*dH[i] + sum_e (1-beta) a/s~[i][p] dH[j][p] + sum_e sum_k c_k Z[j][k]
__global__ __launch_bounds__(BLOCK) void bwd_phase2_kernel(dl_csr_plan c, const float* __restrict__ Z,
                                                           const float* __restrict__ dH, int K, int d, float beta,
                                                           float t, const uint8_t* __restrict__ p,
                                                           const float* __restrict__ a,
                                                           const float* __restrict__ s,
                                                           const float* __restrict__ dw,
                                                           const float* __restrict__ dwr,
                                                           const float* __restrict__ ds, const float* dz_in,
                                                           const float* __restrict__ scale, float* dZ) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6;
    const int row = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (row >= c.n_rows) return;
    const int lane = lane_id();
    const int KD = K * d;
    const size_t grow = (size_t)row + c.row_offset;
    float* acc = lds + (size_t)wave * KD;
    const float* zi = Z + grow * KD;
    const float* dhi = dH + grow * KD;
    for (int x = lane; x < KD; x += DL_WAVE) acc[x] = beta * dhi[x];
    const float omb = 1.0f - beta;
    for (int e = c.rowptr[row]; e < c.rowptr[row + 1]; ++e) {
        const size_t j = (size_t)c.col[e];
        const int k = p[e];
        const float ae = a[e];
        const float si = one_if_zero(s[grow * K + k]);
        const float sj = one_if_zero(s[j * K + k]);
        const float* zj = Z + j * KD;
        {   // aggregation term of the reverse edge (j -> i): a and p are symmetric
            const float w = omb * ae / si;
            const float* dhj = dH + j * KD + k * d;
            for (int x = lane; x < d; x += DL_WAVE) acc[k * d + x] = fmaf(w, dhj[x], acc[k * d + x]);
        }
        const float alpha = edge_softmax(zi, zj, K, d, t);
        const float da = dw[e] / sj + ds[grow * K + k];
        const float dar = dwr[e] / si + ds[j * K + k];
        const float cc = (da + dar) * ae;
        const float ck_mine = cc * ((lane == k ? 1.0f : 0.0f) - alpha) / t;
        for (int kk = 0; kk < K; ++kk) {
            const float ck = __shfl(ck_mine, kk, DL_WAVE);
            for (int x = lane; x < d; x += DL_WAVE) acc[kk * d + x] = fmaf(ck, zj[kk * d + x], acc[kk * d + x]);
        }
    }
    float* out = dZ + grow * KD;
    const float gs = scale ? scale[0] : 1.0f;
    for (int x = lane; x < KD; x += DL_WAVE) {
        const float v = (dz_in ? dz_in[grow * KD + x] : 0.0f) + acc[x];
        out[x] = scale ? v * gs : v;
    }
}

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

int generic_route_fwd(const dl_csr_plan* c, const float* Z, int K, int d, float t, uint8_t* p, float* a,
                      float* s, hipStream_t st) {
    hipLaunchKernelGGL(route_fwd_kernel, dim3(wave_blocks(c->n_rows)), dim3(BLOCK), 0, st, *c, Z, K, d, t, p, a, s);
    return check_launch("route_fwd(generic)");
}

int generic_aggregate_fwd(const dl_csr_plan* c, const float* Z, int K, int d, float beta, const uint8_t* p,
                          const float* a, const float* s, float* H, hipStream_t st) {
    if (int rc = check_lds(K, d, 1)) return rc;
    size_t lds = (size_t)WAVES_PER_BLOCK * K * d * sizeof(float);
    hipLaunchKernelGGL(aggregate_fwd_kernel, dim3(wave_blocks(c->n_rows)), dim3(BLOCK), lds, st, *c, Z, K, d, beta, p,
                       a, s, H);
    return check_launch("aggregate_fwd(generic)");
}

int generic_score_pairs_fwd(const float* Z, const float* H, int K, int d, float t, const int32_t* pu,
                            const int32_t* pv, int P, float* prob, hipStream_t st) {
    hipLaunchKernelGGL(score_pairs_fwd_kernel, dim3(wave_blocks(P)), dim3(BLOCK), 0, st, Z, H, K, d, t, pu, pv, P,
                       prob);
    return check_launch("score_pairs_fwd(generic)");
}

int generic_score_allpairs_fwd(const float* Z, const float* H, int N, int K, int d, float t, float* prob,
                               hipStream_t st) {
    const long long n = (long long)N * N;
    hipLaunchKernelGGL(score_allpairs_fwd_kernel, dim3((unsigned)((n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK)),
                       dim3(BLOCK), 0, st, Z, H, N, K, d, t, prob);
    return check_launch("score_allpairs_fwd(generic)");
}

namespace generic {
// prob / g_prob of the dense [N][N] scorer at the listed pairs (the caller's loss entries): the front end of
// dl_score_allpairs_bwd.  One thread per pair; the two 4-byte reads per pair are scattered by nature.
__global__ __launch_bounds__(BLOCK) void gather_dense_pairs_kernel(const int32_t* __restrict__ pu,
                                                                   const int32_t* __restrict__ pv, int N, int P,
                                                                   const float* __restrict__ prob,
                                                                   const float* __restrict__ g_prob,
                                                                   float* __restrict__ prob_q, float* __restrict__ g_q) {
    const int q = blockIdx.x * BLOCK + threadIdx.x;
    if (q >= P) return;
    const size_t o = (size_t)pu[q] * N + pv[q];
    prob_q[q] = prob[o];
    g_q[q] = g_prob[o];
}

}  // namespace generic

int gather_dense_pairs(const int32_t* pu, const int32_t* pv, int N, int P, const float* prob, const float* g_prob,
                       float* prob_q, float* g_q, hipStream_t st) {
    if (P == 0) return DL_OK;
    hipLaunchKernelGGL(generic::gather_dense_pairs_kernel, dim3((unsigned)((P + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st,
                       pu, pv, N, P, prob, g_prob, prob_q, g_q);
    return check_launch("gather_dense_pairs");
}

int generic_score_pairs_bwd(const dl_pair_incidence* inc, const float* Z, const float* H, int K, int d, float t,
                            const float* prob, const float* g_prob, float* dZ, float* dH, hipStream_t st) {
    if (int rc = check_lds(K, d, 2)) return rc;
    size_t lds = (size_t)WAVES_PER_BLOCK * 2 * K * d * sizeof(float);
    hipLaunchKernelGGL(score_pairs_bwd_kernel, dim3(wave_blocks(inc->csr.n_rows)), dim3(BLOCK), lds, st, inc->csr,
                       inc->inc_pair, Z, H, K, d, t, prob, g_prob, dZ, dH);
    return check_launch("score_pairs_bwd(generic)");
}

int generic_bwd_phase1(const dl_csr_plan* c, const float* Z, int K, int d, float beta, const uint8_t* p,
                       const float* a, const float* s, const float* dH, float* dw, float* dwr, float* ds,
                       hipStream_t st) {
    hipLaunchKernelGGL(bwd_phase1_kernel, dim3(wave_blocks(c->n_rows)), dim3(BLOCK), 0, st, *c, Z, dH, K, d, beta, p,
                       a, s, dw, dwr, ds);
    return check_launch("route_aggregate_bwd_phase1(generic)");
}

int generic_bwd_phase2(const dl_csr_plan* c, const float* Z, int K, int d, float beta, float t, const uint8_t* p,
                       const float* a, const float* s, const float* dH, const float* dw, const float* dwr,
                       const float* ds, const float* dz_in, const float* scale, float* dZ, hipStream_t st) {
    if (int rc = check_lds(K, d, 1)) return rc;
    size_t lds = (size_t)WAVES_PER_BLOCK * K * d * sizeof(float);
    hipLaunchKernelGGL(bwd_phase2_kernel, dim3(wave_blocks(c->n_rows)), dim3(BLOCK), lds, st, *c, Z, dH, K, d, beta, t,
                       p, a, s, dw, dwr, ds, dz_in, scale, dZ);
    return check_launch("route_aggregate_bwd_phase2(generic)");
}

}  // namespace dl
