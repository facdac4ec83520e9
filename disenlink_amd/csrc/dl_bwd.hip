// Backward of route + aggregate (autograd of model.py:56-76): phase 1 (per-edge terms, normaliser gradient), phase 2 (dZ).
// (one of the tuned-kernel translation units; the shared pieces and the design notes are in dl_fast.h)
#include "dl_fast.h"

namespace dl {
namespace fast {

// ---------------------------------------------------------------------------- backward, phase 1
// dw[e] = (1-b) dH[i][p].Z[j][p] ; dwr[e] = (1-b) dH[j][p].Z[i][p] ; ds[i][k] = -(sum [p=k] dwr a)/s~^2
template <int K, int D, typename T>
__global__ __launch_bounds__(BLOCK) void bwd_phase1_seg_kernel(dl_csr_plan g, const T* __restrict__ Z,
                                                               const float* __restrict__ dH, float beta,
                                                               const uint8_t* __restrict__ p,
                                                               const float* __restrict__ a,
                                                               const float* __restrict__ s,
                                                               float* __restrict__ dw, float* __restrict__ dwr,
                                                               float* __restrict__ ds, float* __restrict__ ds_part) {
    using GE = Geo<K, D, T>;
    constexpr int VEC = GE::VEC, G = GE::G, EPW = GE::EPW;
    __shared__ float redk[WAVES_PER_BLOCK][K];
    const WaveSeg ws = load_wave_seg(g);
    const SegInfo si = ws.si;
    const int lane = lane_id();
    const int c = lane % G, grp = lane / G;
    const float omb = 1.0f - beta;
    if (ws.active) {
        float acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = 0.0f;
        int my_col = si.grow, my_k = 0;
        float my_a = 0.0f;
        if (si.beg + lane < si.end) {
            my_col = g.col[si.beg + lane];
            my_k = p[si.beg + lane];
            my_a = a[si.beg + lane];
        }
        for (int base = si.beg; base < si.end; base += EPW) {
            const int e = base + grp;
            const bool live = e < si.end;
            const int j = entry_scalar<EPW>(my_col, base - si.beg, grp);
            const int k = entry_scalar<EPW>(my_k, base - si.beg, grp);
            const float ae = entry_scalar<EPW>(my_a, base - si.beg, grp);
            const size_t oi = (size_t)si.grow * GE::ROW + k * D + c * VEC, oj = (size_t)j * GE::ROW + k * D + c * VEC;
            const float v = omb * group_allreduce_sum<G>(dot(load_f32<VEC>(dH + oi), Tab<T>::load(Z + oj)));
            const float vr = omb * group_allreduce_sum<G>(dot(load_f32<VEC>(dH + oj), Tab<T>::load(Z + oi)));
            if (live && c == 0) { dw[e] = v; dwr[e] = vr; }
            const float contrib = live ? vr * ae : 0.0f;
#pragma unroll
            for (int kk = 0; kk < K; ++kk) acc[kk] += (kk == k) ? contrib : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = across_groups_sum<G>(acc[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) redk[ws.wave][k] = acc[k];
        }
    }
    __syncthreads();
    if (!ws.head || lane >= K) return;
    float tot = redk[ws.wave][lane];                              // lane k: factor k of the unit, waves added in order
    for (int u = 1; u < ws.n_unit; ++u) tot += redk[ws.wave + u][lane];
    if (si.slot < 0) {
        const size_t o = (size_t)si.grow * K + lane;
        ds[o] = ds_from_acc(tot, s[o]);
    } else {
        ds_part[(size_t)si.slot * K + lane] = tot;
    }
}

// ---------------------------------------------------------------------------- backward, phase 2
template <int K, int D, typename T>
__global__ __launch_bounds__(BLOCK) void bwd_phase2_seg_kernel(
    dl_csr_plan g, const T* __restrict__ Z, const float* __restrict__ dH, float beta, float t,
    const uint8_t* __restrict__ p, const float* __restrict__ a, const float* __restrict__ s,
    const float* __restrict__ dw, const float* __restrict__ dwr, const float* __restrict__ ds,
    const float* dz_in, const float* __restrict__ scale, float* dZ, float* __restrict__ dz_part, int sum_rows_here) {
    // dZ = scale[0] * (dz_in + ...): dz_in may be NULL (0) or dZ itself (accumulate in place), scale may be NULL (1)
    using GE = Geo<K, D, T>;
    using FL = typename GE::FL;
    constexpr int VEC = GE::VEC, G = GE::G, EPW = GE::EPW, KP = FL::KP, VPL = FL::VPL, ROW = GE::ROW;
    using US = Stage<K, D, T, 1>;
    __shared__ __attribute__((aligned(16))) float red[US::FLOATS];
    const WaveSeg ws = load_wave_seg(g);
    const SegInfo si = ws.si;
    const int lane = lane_id();
    const int c = lane % G, grp = lane / G;
    const float omb = 1.0f - beta;
    const int kb = FL::factor_base(c);
    if (ws.active) {
        Chunk<VEC> zi[K], acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            zi[k] = Tab<T>::load(Z + (size_t)si.grow * ROW + k * D + c * VEC);
            acc[k] = zero_chunk<VEC>();
        }
        // per-entry scalars (routing factor, softmax-gradient scale cc, aggregation weight w2) are
        // computed once by the entry's own lane and shuffled to its group inside the loop
        int my_col = si.grow, my_k = 0;
        float my_cc = 0.0f, my_w2 = 0.0f;
        if (si.beg + lane < si.end) {
            const int e = si.beg + lane;
            my_col = g.col[e];
            my_k = p[e];
            const float ae = a[e];
            const float s_i = one_if_zero(s[(size_t)si.grow * K + my_k]);
            const float s_j = one_if_zero(s[(size_t)my_col * K + my_k]);
            const float da = dw[e] / s_j + ds[(size_t)si.grow * K + my_k];
            const float dar = dwr[e] / s_i + ds[(size_t)my_col * K + my_k];
            my_cc = (da + dar) * ae;
            my_w2 = omb * ae / s_i;
        }
        for (int base = si.beg; base < si.end; base += EPW) {
            const int j = entry_scalar<EPW>(my_col, base - si.beg, grp);
            const int k = entry_scalar<EPW>(my_k, base - si.beg, grp);
            const float cc = entry_scalar<EPW>(my_cc, base - si.beg, grp);          // 0 past the segment end
            const float w2 = entry_scalar<EPW>(my_w2, base - si.beg, grp);
            Chunk<VEC> zj[K];
#pragma unroll
            for (int kk = 0; kk < K; ++kk) zj[kk] = Tab<T>::load(Z + (size_t)j * ROW + kk * D + c * VEC);
            const Chunk<VEC> dhj = load_f32<VEC>(dH + (size_t)j * ROW + k * D + c * VEC);
            float part[KP];
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) part[kk] = kk < K ? dot(zi[kk < K ? kk : 0], zj[kk < K ? kk : 0]) : 0.0f;
            float ex[VPL];
            const float S = lane_exps<K, G>(part, c, t, ex);
            float ck_lane[VPL];                                     // coefficient of the factors this lane owns
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                ck_lane[i] = cc == 0.0f ? 0.0f : cc * ((kb + i == k ? 1.0f : 0.0f) - ex[i] / S) / t;
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const float ck = group_bcast<G>(ck_lane[FL::src_slot(kk)], FL::src_lane(kk));
                fma_chunk(acc[kk], ck, zj[kk]);
                fma_chunk(acc[kk], kk == k ? w2 : 0.0f, dhj);
            }
        }
        US::put(red, ws.wave, grp, c, acc, 0);
    }
    __syncthreads();
    if (!ws.head) return;
    float4 r[US::NQ];
    US::sum(red, ws.wave, ws.n_unit, lane, r);
    if (si.slot >= 0 && sum_rows_here) {                            // a row of several units: the last of them to get here writes dZ[row]
        publish_unit_and_sum_row<US::F4>(g, si.slot, dz_part, ROW, r, lane, [&](int x, const float4& tot) {
            // the combine launch's arithmetic: (dz_in + beta dH + sum) scale
            const size_t o = (size_t)si.grow * ROW + 4 * x;
            const float4 acc = dz_in ? load4<float>(dz_in + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            store4(dZ + o, combine_finish(acc, beta != 0.0f, beta, load4<float>(dH + o), 1.0f, tot, scale));
        });
        return;
    }
#pragma unroll
    for (int q = 0; q < US::NQ; ++q) {
        const int x = q * DL_WAVE + lane;
        if (x < US::F4) {
            if (si.slot < 0) {
                const size_t o = (size_t)si.grow * ROW + 4 * x;
                const float4 dh = load4<float>(dH + o);
                float4 o4 = dz_in ? load4<float>(dz_in + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                o4.x += beta * dh.x + r[q].x; o4.y += beta * dh.y + r[q].y;
                o4.z += beta * dh.z + r[q].z; o4.w += beta * dh.w + r[q].w;
                if (scale) {
                    const float gs = scale[0];
                    o4.x *= gs; o4.y *= gs; o4.z *= gs; o4.w *= gs;
                }
                store4(dZ + o, o4);
            } else {
                store4(dz_part + (size_t)si.slot * ROW + 4 * x, r[q]);
            }
        }
    }
}

template <int K, int D, typename T>
struct BwdOps {
    static constexpr int ROW = K * D;
    static int bwd_phase1(const dl_csr_plan* g, const void* Z, float beta, const uint8_t* p, const float* a,
                          const float* s, const float* dH, float* dw, float* dwr, float* ds, float* ds_part,
                          hipStream_t st) {
        hipLaunchKernelGGL((bwd_phase1_seg_kernel<K, D, T>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, (const T*)Z,
                           dH, beta, p, a, s, dw, dwr, ds, ds_part);
        launch_vec_combine(g, K, ds_part, 1, s, ds, st);
        return check_launch("route_aggregate_bwd_phase1(fast)");
    }

    static int bwd_phase2(const dl_csr_plan* g, const void* Z, float beta, float t, const uint8_t* p, const float* a,
                          const float* s, const float* dH, const float* dw, const float* dwr, const float* ds,
                          const float* dz_in, const float* scale, float* dZ, float* dz_part, hipStream_t st) {
        const bool here = sums_rows_in_launch(g);
        hipLaunchKernelGGL((bwd_phase2_seg_kernel<K, D, T>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, (const T*)Z,
                           dH, beta, t, p, a, s, dw, dwr, ds, dz_in, scale, dZ, dz_part, here ? 1 : 0);
        if (g->n_multi > 0 && !here)
            hipLaunchKernelGGL((row_combine_kernel<ROW, float, float>), dim3(g->n_multi), dim3(BLOCK), 0, st, *g,
                               dz_part, ROW, dH, beta, 1.0f, dZ, 0, (const float*)nullptr, (float*)nullptr, dz_in, scale);
        return check_launch("route_aggregate_bwd_phase2(fast)");
    }
};

}  // namespace fast

int fast_bwd_phase1(const dl_csr_plan* g, const void* Z, int K, int d, int dtype, float beta, const uint8_t* p,
                    const float* a, const float* s, const float* dH, float* dw, float* dwr, float* ds,
                    float* ds_part, hipStream_t st) {
#define X_F32(KK, DD) if (K == KK && d == DD) return fast::BwdOps<KK, DD, float>::bwd_phase1(g, Z, beta, p, a, s, dH, dw, dwr, ds, ds_part, st);
#define X_BF16(KK, DD) if (K == KK && d == DD) return fast::BwdOps<KK, DD, fast::bf16_t>::bwd_phase1(g, Z, beta, p, a, s, dH, dw, dwr, ds, ds_part, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

int fast_bwd_phase2(const dl_csr_plan* g, const void* Z, int K, int d, int dtype, float beta, float t,
                    const uint8_t* p, const float* a, const float* s, const float* dH, const float* dw,
                    const float* dwr, const float* ds, const float* dz_in, const float* scale, float* dZ, float* dz_part,
                    hipStream_t st) {
#define X_F32(KK, DD) if (K == KK && d == DD) return fast::BwdOps<KK, DD, float>::bwd_phase2(g, Z, beta, t, p, a, s, dH, dw, dwr, ds, dz_in, scale, dZ, dz_part, st);
#define X_BF16(KK, DD) if (K == KK && d == DD) return fast::BwdOps<KK, DD, fast::bf16_t>::bwd_phase2(g, Z, beta, t, p, a, s, dH, dw, dwr, ds, dz_in, scale, dZ, dz_part, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

}  // namespace dl
