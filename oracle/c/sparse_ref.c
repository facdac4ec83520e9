/* Edge-list CPU restatement of the DisenLink hot path in plain C (OpenMP).  TEST INFRASTRUCTURE ONLY.
 *
 * Same arithmetic as oracle/sparse_ref.py (which is pinned to the reference's golden vectors and is
 * checked against this file in tests/test_oracle_golden.py); exists so that parity can be checked at
 * the FULL benchmark sizes in seconds and so that bench.py has a multi-threaded sparse CPU baseline
 * for graphs where the reference's dense [K,N,N] form cannot run (baseline B).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Reference lines followed:
 *   route      model.py:56-66   e = exp(z.z/t), alpha = e / sum_k e, p = argmax (first max), a = alpha_p
 *   normaliser model.py:70-72   s_k[i] = sum_{j in N(i), p=k} a ; zero -> 1 on read
 *   aggregate  model.py:73-75   h_k[i] = b z_k[i] + (1-b) sum_j a_ij / s_k[j] z_k[j]   (s of the NEIGHBOUR)
 *   score      model.py:110-113 P = sigmoid(sum_k (h_k[u].h_k[v]) exp(z_k[u].z_k[v]/t))  (raw exp)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float dotf(const float* x, const float* y, int d) {
    float r = 0.0f;
    for (int c = 0; c < d; ++c) r += x[c] * y[c];
    return r;
}

/* p[e], a[e] per edge and raw s[N][K] per node. */
void dlo_route(const float* Z, int N, int K, int d, const int32_t* rowptr, const int32_t* col, float t,
               uint8_t* p, float* a, float* s) {
    const size_t row = (size_t)K * d;
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < N; ++i) {
        float ex[64];
        float* si = s + (size_t)i * K;
        for (int k = 0; k < K; ++k) si[k] = 0.0f;
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const float* zi = Z + (size_t)i * row;
            const float* zj = Z + (size_t)col[e] * row;
            float S = 0.0f;
            for (int k = 0; k < K; ++k) {
                ex[k] = expf(dotf(zi + k * d, zj + k * d, d) / t);
                S += ex[k];
            }
            int win = 0;
            float best = ex[0] / S;
            for (int k = 1; k < K; ++k) {
                const float al = ex[k] / S;
                if (al > best || (al != al && best == best)) { best = al; win = k; }   /* NaN counts as max */
            }
            p[e] = (uint8_t)win;
            a[e] = best;
            si[win] += best;
        }
    }
}

void dlo_aggregate(const float* Z, int N, int K, int d, const int32_t* rowptr, const int32_t* col, float beta,
                   const uint8_t* p, const float* a, const float* s, float* H) {
    const size_t row = (size_t)K * d;
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < N; ++i) {
        float* hi = H + (size_t)i * row;
        for (size_t x = 0; x < row; ++x) hi[x] = 0.0f;
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const int j = col[e], k = p[e];
            float sj = s[(size_t)j * K + k];
            if (sj == 0.0f) sj = 1.0f;
            const float w = a[e] / sj;
            const float* zj = Z + (size_t)j * row + (size_t)k * d;
            for (int c = 0; c < d; ++c) hi[k * d + c] += w * zj[c];
        }
        const float* zi = Z + (size_t)i * row;
        for (size_t x = 0; x < row; ++x) hi[x] = beta * zi[x] + (1.0f - beta) * hi[x];
    }
}

void dlo_score_pairs(const float* Z, const float* H, int K, int d, float t, const int32_t* pu, const int32_t* pv,
                     int64_t P, float* prob) {
    const size_t row = (size_t)K * d;
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < P; ++q) {
        const float* zu = Z + (size_t)pu[q] * row;
        const float* zv = Z + (size_t)pv[q] * row;
        const float* hu = H + (size_t)pu[q] * row;
        const float* hv = H + (size_t)pv[q] * row;
        float logit = 0.0f;
        for (int k = 0; k < K; ++k)
            logit += dotf(hu + k * d, hv + k * d, d) * expf(dotf(zu + k * d, zv + k * d, d) / t);
        prob[q] = 1.0f / (1.0f + expf(-logit));
    }
}

/* ---- backward, gather form: every output row is produced by its owner -------------------
 * The same mathematics as oracle/sparse_ref.py's scatter form (which is what the reference's autograd fixtures pin):
 *   dw_e   = (1-b) dH_p[i].z_p[j]                       weight gradient of edge e = (i, j), p = p[e]
 *   ds_k[i]= -(sum_{e in row i, p=k} dw_(j,i) a_e) / s~_k[i]^2     (0 where the raw sum was 0; dw_(j,i) = (1-b) dH_p[j].z_p[i])
 *   da_e   = dw_e / s~_p[j] + ds_p[i]                   (normaliser: every edge of row i, factor k, feeds s_k[i])
 *   dz_k[i]= b dH_k[i] + sum_e [k=p] (1-b) a_e / s~_p[i] dH_p[j]                       (the reverse edge's aggregation term)
 *            + sum_e (da_e + da_(j,i)) a_e ([k=p] - alpha_k) / t  z_k[j]               (softmax at the winning index)
 * a and p are symmetric in (i, j), so the reverse edge's quantities are recomputed from row i's own data.
 * model.py:56-75 differentiated; main_disentangled.py:198 triggers it in the reference. */
void dlo_route_aggregate_bwd(const float* Z, int N, int K, int d, const int32_t* rowptr, const int32_t* col,
                             float beta, float t, const uint8_t* p, const float* a, const float* s, const float* dH,
                             float* dZ) {
    const size_t row = (size_t)K * d;
    const float omb = 1.0f - beta;
    const int64_t E = rowptr[N];
    float* dw = (float*)malloc(sizeof(float) * (size_t)(E > 0 ? E : 1));
    float* dwr = (float*)malloc(sizeof(float) * (size_t)(E > 0 ? E : 1));
    float* ds = (float*)malloc(sizeof(float) * (size_t)N * K);
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < N; ++i) {
        float acc[64];
        for (int k = 0; k < K; ++k) acc[k] = 0.0f;
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const int j = col[e], k = p[e];
            dw[e] = omb * dotf(dH + (size_t)i * row + (size_t)k * d, Z + (size_t)j * row + (size_t)k * d, d);
            dwr[e] = omb * dotf(dH + (size_t)j * row + (size_t)k * d, Z + (size_t)i * row + (size_t)k * d, d);
            acc[k] += dwr[e] * a[e];
        }
        for (int k = 0; k < K; ++k) {
            const float sr = s[(size_t)i * K + k];
            ds[(size_t)i * K + k] = sr == 0.0f ? 0.0f : -acc[k] / (sr * sr);
        }
    }
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < N; ++i) {
        float ex[64];
        float* dzi = dZ + (size_t)i * row;
        const float* zi = Z + (size_t)i * row;
        for (size_t x = 0; x < row; ++x) dzi[x] = beta * dH[(size_t)i * row + x];
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const int j = col[e], k = p[e];
            const float* zj = Z + (size_t)j * row;
            float si = s[(size_t)i * K + k], sj = s[(size_t)j * K + k];
            if (si == 0.0f) si = 1.0f;
            if (sj == 0.0f) sj = 1.0f;
            const float da = dw[e] / sj + ds[(size_t)i * K + k];
            const float dar = dwr[e] / si + ds[(size_t)j * K + k];
            const float cc = (da + dar) * a[e];
            float S = 0.0f;
            for (int kk = 0; kk < K; ++kk) {
                ex[kk] = expf(dotf(zi + kk * d, zj + kk * d, d) / t);
                S += ex[kk];
            }
            for (int kk = 0; kk < K; ++kk) {
                const float ck = cc * ((kk == k ? 1.0f : 0.0f) - ex[kk] / S) / t;
                for (int c = 0; c < d; ++c) dzi[kk * d + c] += ck * zj[kk * d + c];
            }
            const float w2 = omb * a[e] / si;
            const float* dhj = dH + (size_t)j * row + (size_t)k * d;
            for (int c = 0; c < d; ++c) dzi[k * d + c] += w2 * dhj[c];
        }
    }
    free(dw);
    free(dwr);
    free(ds);
}

/* Scorer backward over the node-incidence list (every pair once per endpoint; incptr[N+1], inc_other, inc_pair):
 *   gl = g_prob p (1 - p);  dH_k[u] += gl e_k H_k[v];  dZ_k[u] += gl q_k e_k / t  Z_k[v]      (model.py:110-113 differentiated) */
void dlo_score_pairs_bwd(const float* Z, const float* H, int N, int K, int d, float t, const int32_t* incptr,
                         const int32_t* inc_other, const int32_t* inc_pair, const float* prob, const float* g_prob,
                         float* dZ, float* dH) {
    const size_t row = (size_t)K * d;
#pragma omp parallel for schedule(dynamic, 16)
    for (int u = 0; u < N; ++u) {
        float* dzu = dZ + (size_t)u * row;
        float* dhu = dH + (size_t)u * row;
        for (size_t x = 0; x < row; ++x) { dzu[x] = 0.0f; dhu[x] = 0.0f; }
        const float* zu = Z + (size_t)u * row;
        const float* hu = H + (size_t)u * row;
        for (int e = incptr[u]; e < incptr[u + 1]; ++e) {
            const int v = inc_other[e], q = inc_pair[e];
            const float pr = prob[q];
            const float gl = g_prob[q] * pr * (1.0f - pr);
            const float* zv = Z + (size_t)v * row;
            const float* hv = H + (size_t)v * row;
            for (int k = 0; k < K; ++k) {
                const float ek = expf(dotf(zu + k * d, zv + k * d, d) / t);
                const float qk = dotf(hu + k * d, hv + k * d, d);
                const float ch = gl * ek, cz = gl * qk * ek / t;
                for (int c = 0; c < d; ++c) {
                    dhu[k * d + c] += ch * hv[k * d + c];
                    dzu[k * d + c] += cz * zv[k * d + c];
                }
            }
        }
    }
}
