// Pair scorer (model.py:109-113 on a pair list), the gather-form dense scorer, and the separate scorer backward.
// (one of the tuned-kernel translation units; the shared pieces and the design notes are in dl_fast.h)
#include "dl_fast.h"
#include "dl_score_bwd.h"

namespace dl {
namespace fast {

// One wave per segment of the "pairs by first endpoint" plan: the u rows of Z and H are staged once
// in LDS, every lane group then scores one pair per iteration from the gathered v rows.
template <int K, int D, typename T, bool COEF>
// (4 waves per SIMD: pinned at 5 / 6 / 8 hipcc keeps fewer row gathers in flight per wave — 301 / 543 / 612 us against
// 170 on squirrel; whether the u rows sit in registers or are re-read from LDS every iteration makes no difference:
// the kernel is bound by the vector-L1 / L2 pipeline, 4.3 GB through 256 x 64 B/clk.)
__global__ __launch_bounds__(BLOCK, K <= 8 ? 4 : 1) void score_fwd_seg_kernel(dl_csr_plan g, const int32_t* __restrict__ pair_id,
                                                              const T* __restrict__ Z, const T* __restrict__ H,
                                                              float t, float* __restrict__ prob,
                                                              float* __restrict__ coef_e,
                                                              float* __restrict__ coef_q) {
    using GE = Geo<K, D, T>;
    constexpr int VEC = GE::VEC, G = GE::G, EPW = GE::EPW, ROW = GE::ROW;
    constexpr int KB = K > 8 ? 8 : K;                        // factor block
    using FLB = FactorLanes<G, KB>;
    constexpr int KBP = FLB::KP;
    __shared__ __attribute__((aligned(16))) float urow[WAVES_PER_BLOCK][2 * ROW];
    const WaveSeg ws = load_wave_seg(g);
    const SegInfo si = ws.si;
    const int wave = ws.wave, lane = lane_id();
    const bool active = ws.active;
    if (active) stage_u_rows<K, D, T>(urow[wave], Z, H, (size_t)si.grow);
    __syncthreads();
    if (!active) return;
    const int c = lane % G, grp = lane / G;
    const int kbb = FLB::factor_base(c);
    int my_col = si.grow, my_pair = 0;
    if (si.beg + lane < si.end) {
        my_col = g.col[si.beg + lane];
        my_pair = pair_id[si.beg + lane];
    }
    for (int base = si.beg; base < si.end; base += EPW) {
        const int it = base + grp;
        const bool live = it < si.end;
        const size_t v = (size_t)entry_scalar<EPW>(my_col, base - si.beg, grp);
        const int q = entry_scalar<EPW>(my_pair, base - si.beg, grp);
        // factors are processed in blocks of KB <= 8: at most 2*KB row chunks live at a time, whatever K is
        float term = 0.0f;
#pragma unroll
        for (int b0 = 0; b0 < K; b0 += KB) {
            float pq[KBP], ps[KBP];
#pragma unroll
            for (int k = 0; k < KBP; ++k) {
                const bool in = k < KB && b0 + k < K;
                const int kk = in ? b0 + k : 0;
                pq[k] = in ? dot(load_f32<VEC>(&urow[wave][ROW + kk * D + c * VEC]), Tab<T>::load(H + v * ROW + kk * D + c * VEC)) : 0.0f;
                ps[k] = in ? dot(load_f32<VEC>(&urow[wave][kk * D + c * VEC]), Tab<T>::load(Z + v * ROW + kk * D + c * VEC)) : 0.0f;
            }
            TransposedReduce<KBP, G / 2>::run(pq, c);
            TransposedReduce<KBP, G / 2>::run(ps, c);
#pragma unroll
            for (int i = 0; i < FLB::VPL; ++i) {
                const int k = b0 + kbb + i;
                if (FLB::primary(c) && kbb + i < KB && k < K) {
                    const float ek = expf(div_t(ps[i], t));
                    const float qe = pq[i] * ek;
                    term += qe;
                    if (COEF && live) {                         // per-factor logit terms for the backward
                        coef_e[(size_t)q * K + k] = ek;
                        coef_q[(size_t)q * K + k] = qe;
                    }
                }
            }
        }
        const float logit = group_allreduce_sum<G>(term);
        if (live && c == 0) prob[q] = sigmoid_ref(logit);
    }
}

// Dense [N][N] scorer (the reference's link_pred, model.py:109-113): no pair list at all.  One wave = one
// row u x one chunk of <= VCH consecutive columns; the column space is cut into n_slices XCD slices exactly
// like the pair plans (workgroup b serves slice b % n_slices), so an XCD's L2 holds the v rows it gathers.
template <int K, int D, typename T>
__global__ __launch_bounds__(BLOCK) void score_allpairs_kernel(const T* __restrict__ Z, const T* __restrict__ H, int N,
                                                               float t, int n_slices, int slice_w, int chunks_per_u,
                                                               float* __restrict__ prob) {
    using GE = Geo<K, D, T>;
    constexpr int VEC = GE::VEC, G = GE::G, EPW = GE::EPW, ROW = GE::ROW;
    constexpr int KB = K > 8 ? 8 : K;
    using FLB = FactorLanes<G, KB>;
    constexpr int KBP = FLB::KP;
    constexpr int VCH = 256;
    __shared__ __attribute__((aligned(16))) float urow[WAVES_PER_BLOCK][2 * ROW];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int x = blockIdx.x % n_slices;
    const int item = (blockIdx.x / n_slices) * WAVES_PER_BLOCK + wave;
    const int u = item / chunks_per_u, ch = item - u * chunks_per_u;
    const int v0 = x * slice_w + ch * VCH;
    const int v1 = min(min(v0 + VCH, (x + 1) * slice_w), N);
    const bool active = u < N && v0 < v1;
    if (active) stage_u_rows<K, D, T>(urow[wave], Z, H, (size_t)u);
    __syncthreads();
    if (!active) return;
    const int c = lane % G, grp = lane / G;
    const int kbb = FLB::factor_base(c);
    for (int base = v0; base < v1; base += EPW) {
        const int vi = base + grp;
        const bool live = vi < v1;
        const size_t v = (size_t)(live ? vi : v0);
        float term = 0.0f;
#pragma unroll
        for (int b0 = 0; b0 < K; b0 += KB) {
            float pq[KBP], ps[KBP];
#pragma unroll
            for (int k = 0; k < KBP; ++k) {
                const bool in = k < KB && b0 + k < K;
                const int kk = in ? b0 + k : 0;
                pq[k] = in ? dot(load_f32<VEC>(&urow[wave][ROW + kk * D + c * VEC]), Tab<T>::load(H + v * ROW + kk * D + c * VEC)) : 0.0f;
                ps[k] = in ? dot(load_f32<VEC>(&urow[wave][kk * D + c * VEC]), Tab<T>::load(Z + v * ROW + kk * D + c * VEC)) : 0.0f;
            }
            TransposedReduce<KBP, G / 2>::run(pq, c);
            TransposedReduce<KBP, G / 2>::run(ps, c);
#pragma unroll
            for (int i = 0; i < FLB::VPL; ++i)
                if (FLB::primary(c) && kbb + i < KB && b0 + kbb + i < K) term += pq[i] * expf(div_t(ps[i], t));
        }
        const float logit = group_allreduce_sum<G>(term);
        if (live && c == 0) prob[(size_t)u * N + vi] = sigmoid_ref(logit);
    }
}

// Scorer backward from stored per-factor terms: a weighted row gather, one launch per output.
//   PASS 0: dZ[u] = sum_inc (gl/t) * (q_k e_k) * Z[v][k]      PASS 1: dH[u] = sum_inc gl * e_k * H[v][k]
template <int K, int D, typename T, int PASS>
__global__ __launch_bounds__(BLOCK) void score_bwd_coef_seg_kernel(dl_csr_plan g, const int32_t* __restrict__ inc_pair,
                                                                   const T* __restrict__ X, float t,
                                                                   const float* __restrict__ prob,
                                                                   const float* __restrict__ g_prob,
                                                                   const float* __restrict__ coef,
                                                                   float* __restrict__ out, float* __restrict__ part) {
    using GE = Geo<K, D, T>;
    constexpr int VEC = GE::VEC, G = GE::G, EPW = GE::EPW, ROW = GE::ROW;
    using US = Stage<K, D, T, 1>;
    __shared__ __attribute__((aligned(16))) float red[US::FLOATS];
    const WaveSeg ws = load_wave_seg(g);
    const SegInfo si = ws.si;
    const int lane = lane_id();
    const int c = lane % G, grp = lane / G;
    if (ws.active) {
        Chunk<VEC> acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = zero_chunk<VEC>();
        int my_col = si.grow, my_pair = 0;
        float my_gl = 0.0f;
        if (si.beg + lane < si.end) {
            my_col = g.col[si.beg + lane];
            my_pair = inc_pair[si.beg + lane];
            const float pr = prob[my_pair];
            my_gl = g_prob[my_pair] * pr * (1.0f - pr);      // sigmoid backward p(1-p)
            if (PASS == 0) my_gl = div_t(my_gl, t);
        }
        for (int base = si.beg; base < si.end; base += EPW) {
            const int idx = base + grp - si.beg;
            const size_t v = (size_t)__shfl(my_col, idx, DL_WAVE);
            const int q = __shfl(my_pair, idx, DL_WAVE);
            const float gl = __shfl(my_gl, idx, DL_WAVE);     // 0 past the segment end
            float ck[K];
            if constexpr (K % 4 == 0) {
#pragma unroll
                for (int k = 0; k < K; k += 4) {
                    const float4 t4 = *reinterpret_cast<const float4*>(coef + (size_t)q * K + k);
                    ck[k] = t4.x; ck[k + 1] = t4.y; ck[k + 2] = t4.z; ck[k + 3] = t4.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) ck[k] = coef[(size_t)q * K + k];
            }
            // gathers in blocks of <= 8 factor slices: bounded live registers for any K
#pragma unroll
            for (int b0 = 0; b0 < K; b0 += 8) {
                Chunk<VEC> xv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (b0 + k < K) xv[k] = Tab<T>::load(X + v * ROW + (b0 + k) * D + c * VEC);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (b0 + k < K) fma_chunk(acc[b0 + k], gl * ck[b0 + k], xv[k]);
            }
        }
        US::put(red, ws.wave, grp, c, acc, 0);
    }
    __syncthreads();
    if (!ws.head) return;
    float4 r[US::NQ];
    US::sum(red, ws.wave, ws.n_unit, lane, r);
    float* o = si.slot < 0 ? out + (size_t)si.grow * ROW : part + (size_t)si.slot * ROW;
#pragma unroll
    for (int q = 0; q < US::NQ; ++q) {
        const int x = q * DL_WAVE + lane;
        if (x < US::F4) store4(o + 4 * x, r[q]);
    }
}

template <int K, int D, typename T>
struct ScoreOps {
    static constexpr int ROW = K * D;
    static int score_fwd(const dl_pair_incidence* by_u, const void* Z, const void* H, float t, float* prob,
                         float* coef, hipStream_t st) {
        const dl_csr_plan* g = &by_u->csr;
        float* coef_q = coef ? coef + (size_t)by_u->n_pairs * K : nullptr;
        if (coef)
            hipLaunchKernelGGL((score_fwd_seg_kernel<K, D, T, true>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g,
                               by_u->inc_pair, (const T*)Z, (const T*)H, t, prob, coef, coef_q);
        else
            hipLaunchKernelGGL((score_fwd_seg_kernel<K, D, T, false>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g,
                               by_u->inc_pair, (const T*)Z, (const T*)H, t, prob, coef, coef_q);
        return check_launch("score_pairs_fwd(fast)");
    }

    static int score_allpairs(const void* Z, const void* H, int N, float t, float* prob, hipStream_t st) {
        // slice the columns 8 ways only while a slice of Z+H can live in an XCD's L2 (like graph.auto_slices)
        const double table = 2.0 * N * ROW * sizeof(T);
        const int n_slices = table <= 8.0 * 8.0 * (4 << 20) && N >= 64 ? 8 : 1;
        const int slice_w = (N + n_slices - 1) / n_slices;
        const int chunks_per_u = (slice_w + 255) / 256;
        const long long items = (long long)N * chunks_per_u;
        const unsigned blocks = (unsigned)(n_slices * ((items + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK));
        hipLaunchKernelGGL((score_allpairs_kernel<K, D, T>), dim3(blocks), dim3(BLOCK), 0, st, (const T*)Z, (const T*)H, N,
                           t, n_slices, slice_w, chunks_per_u, prob);
        return check_launch("score_allpairs_fwd(fast)");
    }

    static int score_bwd(const dl_pair_incidence* inc, const void* Z, const void* H, float t, const float* prob,
                         const float* g_prob, const float* coef, float* dZ, float* dH, float* part, hipStream_t st) {
        const dl_csr_plan* g = &inc->csr;
        const float* no_x = nullptr;
        if (coef) {
            const float* coef_q = coef + (size_t)inc->n_pairs * K;
            float* part_h = part + (size_t)g->n_slots * ROW;
            hipLaunchKernelGGL((score_bwd_coef_seg_kernel<K, D, T, 0>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g,
                               inc->inc_pair, (const T*)Z, t, prob, g_prob, coef_q, dZ, part);
            hipLaunchKernelGGL((score_bwd_coef_seg_kernel<K, D, T, 1>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g,
                               inc->inc_pair, (const T*)H, t, prob, g_prob, coef, dH, part_h);
            if (g->n_multi > 0)
                hipLaunchKernelGGL((row_combine_kernel<ROW, float, float>), dim3(g->n_multi, 2), dim3(BLOCK), 0, st, *g,
                                   part, ROW, no_x, 0.0f, 1.0f, dZ, 0, part_h, dH);
            return check_launch("score_pairs_bwd(fast, stored terms)");
        }
        hipLaunchKernelGGL((score_bwd_seg_kernel<K, D, T, false>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, inc->inc_pair,
                           (const T*)Z, (const T*)H, t, prob, g_prob, dZ, dH, part);
        if (g->n_multi > 0)
            hipLaunchKernelGGL((row_combine_kernel<ROW, float, float>), dim3(g->n_multi, 2), dim3(BLOCK), 0, st, *g, part,
                               2 * ROW, no_x, 0.0f, 1.0f, dZ, 0, part + ROW, dH);
        return check_launch("score_pairs_bwd(fast)");
    }
};

}  // namespace fast

int fast_score_pairs_fwd(const dl_pair_incidence* by_u, const void* Z, const void* H, int K, int d, int dtype,
                         float t, float* prob, float* coef, hipStream_t st) {
#define X_F32(KK, DD) if (K == KK && d == DD) return fast::ScoreOps<KK, DD, float>::score_fwd(by_u, Z, H, t, prob, coef, st);
#define X_BF16(KK, DD) if (K == KK && d == DD) return fast::ScoreOps<KK, DD, fast::bf16_t>::score_fwd(by_u, Z, H, t, prob, coef, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

int fast_score_pairs_bwd(const dl_pair_incidence* inc, const void* Z, const void* H, int K, int d, int dtype,
                         float t, const float* prob, const float* g_prob, const float* coef, float* dZ, float* dH,
                         float* part, hipStream_t st) {
#define X_F32(KK, DD) if (K == KK && d == DD) return fast::ScoreOps<KK, DD, float>::score_bwd(inc, Z, H, t, prob, g_prob, coef, dZ, dH, part, st);
#define X_BF16(KK, DD) if (K == KK && d == DD) return fast::ScoreOps<KK, DD, fast::bf16_t>::score_bwd(inc, Z, H, t, prob, g_prob, coef, dZ, dH, part, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

int fast_score_allpairs_fwd(const void* Z, const void* H, int N, int K, int d, int dtype, float t, float* prob,
                            hipStream_t st) {
#define X_F32(KK, DD) if (K == KK && d == DD) return fast::ScoreOps<KK, DD, float>::score_allpairs(Z, H, N, t, prob, st);
#define X_BF16(KK, DD) if (K == KK && d == DD) return fast::ScoreOps<KK, DD, fast::bf16_t>::score_allpairs(Z, H, N, t, prob, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

}  // namespace dl
