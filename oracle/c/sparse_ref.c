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
