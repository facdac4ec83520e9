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

// ---------------------------------------------------------------------------- forward scorer, wave per entry (round 6)
// The forward pass in the geometry of the one-pass training scorer (dl_train.hip: score_train_wave_kernel) for d = 64,
// K in {4, 8}, fp32 tables: the 64 lanes share ONE pair, lane l holds float4 number j * 64 + l of a row (a DPP row of 16
// lanes = one factor slice), the row base is a scalar (readlane of the segment's columns) and the lane offset a constant,
// the u rows are re-read from the wave's LDS region every step, U = 4 pairs per step, the 16 partial dot products reduced
// over the DPP row by one transposed reduction: one expf and one sigmoid per step.  No accumulators, so fewer registers
// than the training kernel; the arithmetic up to the probability is that kernel's, operation for operation — the two give the
// same bits.  The group-per-entry kernel above stays for every other shape (and as the reference form: -DDL_FWD_WAVE_KERNEL=0).
// Why: the training kernel's bound build ran these very gathers, without arithmetic, at 0.86 of the L2 peak; the
// group-per-entry forward reached 0.79.
#ifndef DL_FWD_WAVE_KERNEL
#define DL_FWD_WAVE_KERNEL 1
#endif
#ifndef DL_FWD_WAVE_MAXW
#define DL_FWD_WAVE_MAXW 6            // waves per SIMD the register allocation aims at (4 / 5 / 6 measured: 156.6 / 156.8 / 154.4 us, profiles/r7n_*)
#endif
template <int K, int D>
struct FwdWave {
    static constexpr bool ok = DL_FWD_WAVE_KERNEL && D == 64 && (K == 4 || K == 8);
    static constexpr int NJ = K * D / 256, U = 4;
};

// (amdgpu_waves_per_eu(4, 5): left to itself hipcc aims at 8 waves per SIMD = 64 registers — exactly the 16 gathered float4 of a
// step — by requesting only 12 of them up front and the rest behind three more vmcnt(0) round trips per step.)
template <int K, int D, bool T1, bool COEF>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, DL_FWD_WAVE_MAXW))) void score_fwd_wave_kernel(dl_csr_plan g, const int32_t* __restrict__ pair_id,
                                                                  const float* __restrict__ Z, const float* __restrict__ H, float t,
                                                                  float* __restrict__ prob, float* __restrict__ coef_e,
                                                                  float* __restrict__ coef_q) {
    using FW = FwdWave<K, D>;
    constexpr int NJ = FW::NJ, U = FW::U, ROW = K * D;
    static_assert(D == 64 && NJ >= 1 && U * NJ <= 8, "one DPP row of 16 lanes per factor slice; at most 8 exponents per row and step");
    __shared__ __attribute__((aligned(16))) float4 urow[WAVES_PER_BLOCK][2 * ROW / 4];      // [Z row | H row] of the segment's u
    __shared__ int ent_q[WAVES_PER_BLOCK][DL_WAVE];
    const WaveSeg ws = load_wave_seg(g);
    if (!ws.active) return;                                         // no barrier in this kernel: every region is its wave's own
    const SegInfo si = ws.si;
    const int wave = ws.wave, lane = lane_id();
    const int i = lane & 15, r = lane >> 4;                         // position in the DPP row; the row holds the factors r, r + 4
    float4* const mine = urow[wave];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        mine[j * 64 + lane] = *reinterpret_cast<const float4*>(Z + (size_t)si.grow * ROW + (j * 64 + lane) * 4);
        mine[ROW / 4 + j * 64 + lane] = *reinterpret_cast<const float4*>(H + (size_t)si.grow * ROW + (j * 64 + lane) * 4);
    }
    int my_col = si.grow, my_q = 0;
    if (si.beg + lane < si.end) {
        my_col = g.col[si.beg + lane];
        my_q = pair_id[si.beg + lane];
    }
    ent_q[wave][lane] = my_q;                                       // written and read by this wave only
    const int nsteps = (si.end - si.beg + U - 1) / U;               // entries past the end repeat a valid row, nothing is stored for them
    for (int step = 0; step < nsteps; ++step) {
        float4 zv[U][NJ], hv[U][NJ];
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const size_t v = (size_t)(unsigned)__builtin_amdgcn_readlane(my_col, (step * U + e) & 63);
            const auto* zr = uniform_row<dl_vf4>(Z, v * ROW * sizeof(float));     // row base in SGPRs (dl_fast.h)
            const auto* hr = uniform_row<dl_vf4>(H, v * ROW * sizeof(float));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                zv[e][j] = as_float4(zr[(unsigned)(lane + j * 64)]);
                hv[e][j] = as_float4(hr[(unsigned)(lane + j * 64)]);
            }
        }
        float val[16];                                              // index = table * 8 + chunk * 4 + entry
#pragma unroll
        for (int x = 0; x < 16; ++x) val[x] = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float4 a4 = mine[j * 64 + lane], b4 = mine[ROW / 4 + j * 64 + lane];
#pragma unroll
            for (int e = 0; e < U; ++e) {
                val[j * 4 + e] = dot4_packed(a4, zv[e][j]);
                val[8 + j * 4 + e] = dot4_packed(b4, hv[e][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        TransposedReduce<16, 8>::run(val, i);                       // lane i: sum number i — below 8: z_u . z_v of (chunk i / 4, entry i % 4), above: h_u . h_v
        const float mine_v = val[0];
        const float ex = expf(T1 ? mine_v : mine_v / t);
        const float ttv = mine_v * xor_lane<8>(ex);                 // lanes 8..15: (h.h) e^(z.z / t) of (chunk, entry)
        float term = ttv;
        if constexpr (NJ == 2) term += xor_lane<4>(ttv);
        const float logit = add_xor<32>(add_xor<16>(term));         // over the 4 DPP rows: all factors
        const float p = sigmoid_ref(logit);
        const int idx = step * U + (i & 3);
        const bool live = si.beg + idx < si.end;
        const int qq = ent_q[wave][idx & 63];
        if (lane >= 8 && lane < 12 && live) prob[qq] = p;
        if constexpr (COEF) {                                       // per-factor terms for the separate backward: e_k, q_k e_k
            const int k = 4 * ((i & 7) >> 2) + r;                   // the factor this lane's sum belongs to (chunk j holds the factors 4 j + r)
            if (live && (NJ == 2 || (i & 7) < 4)) {
                if (i < 8) coef_e[(size_t)qq * K + k] = ex;
                else coef_q[(size_t)qq * K + k] = ttv;
            }
        }
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
        if constexpr (std::is_same<T, float>::value && FwdWave<K, D>::ok) {
            if (g->seg_len <= 64 && g->seg_len % FwdWave<K, D>::U == 0 && !config().fwd_group_kernel) {
                auto launch = [&](auto kern) {
                    hipLaunchKernelGGL(kern, dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, by_u->inc_pair, (const float*)Z,
                                       (const float*)H, t, prob, coef, coef_q);
                };
                if (coef) { if (t == 1.0f) launch(score_fwd_wave_kernel<K, D, true, true>); else launch(score_fwd_wave_kernel<K, D, false, true>); }
                else { if (t == 1.0f) launch(score_fwd_wave_kernel<K, D, true, false>); else launch(score_fwd_wave_kernel<K, D, false, false>); }
                return check_launch("score_pairs_fwd(fast, wave per entry)");
            }
        }
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
