// Tuned kernels for gfx950, instantiated per (K, D).
//
// Work decomposition: the plan cuts every CSR row into segments of <= seg_len consecutive
// entries; ONE 64-lane wave owns one segment, so a hub row of thousands of edges is spread over
// many waves and CUs while a median row (tens of edges) is a single wave.  Inside a wave, a group
// of G = D/4 lanes owns one entry: lane c of the group holds the float4 chunk c of every factor
// slice, i.e. one neighbour row Z[j] (K*D*4 bytes, contiguous) is fetched by K coalesced
// 16-byte-per-lane loads and 64/G entries are in flight per wave iteration.  The K dot products
// are reduced with log2(G) cross-lane butterflies; the K-way softmax / arg-max is then computed
// redundantly by every lane of the group, so no further exchange is needed.
//
// Rows with one segment write their outputs directly; segments of multi-segment rows write
// per-segment partials (in the caller's workspace) that a combine kernel sums in segment order.
// No float atomics anywhere: results are bitwise reproducible.
#include "dl_common.h"
#include "dl_kernels.h"

namespace dl {
namespace fast {

__device__ __forceinline__ float dot4(const float4& x, const float4& y) {
    return fmaf(x.w, y.w, fmaf(x.z, y.z, fmaf(x.y, y.y, x.x * y.x)));
}
__device__ __forceinline__ void fma4(float4& acc, float w, const float4& v) {
    acc.x = fmaf(w, v.x, acc.x);
    acc.y = fmaf(w, v.y, acc.y);
    acc.z = fmaf(w, v.z, acc.z);
    acc.w = fmaf(w, v.w, acc.w);
}
template <int G>
__device__ __forceinline__ void across_groups_sum4(float4& v) {
    v.x = across_groups_sum<G>(v.x);
    v.y = across_groups_sum<G>(v.y);
    v.z = across_groups_sum<G>(v.z);
    v.w = across_groups_sum<G>(v.w);
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// e_k = exp(z_k[i].z_k[j] / t) for all K factors and their sum S (sequential in k).
template <int K, int G>
__device__ __forceinline__ float edge_exps(const float4 (&zi)[K], const float4 (&zj)[K], float t, float (&ex)[K]) {
    float S = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        ex[k] = expf(group_allreduce_sum<G>(dot4(zi[k], zj[k])) / t);
        S += ex[k];
    }
    return S;
}

// ---------------------------------------------------------------------------- route
template <int K, int D>
__global__ __launch_bounds__(BLOCK) void route_seg_kernel(dl_csr_plan g, const float* __restrict__ Z, float t,
                                                          uint8_t* __restrict__ p, float* __restrict__ a,
                                                          float* __restrict__ s, float* __restrict__ s_part) {
    constexpr int G = D / 4;            // lanes per edge
    constexpr int EPW = DL_WAVE / G;    // edges per wave iteration
    const int seg = wave_segment(g);
    if (seg < 0) return;
    const int lane = lane_id();
    const int c = lane % G, grp = lane / G;
    const SegInfo si = load_seg(g, seg);
    const float4* __restrict__ Z4 = reinterpret_cast<const float4*>(Z);
    const size_t rs = (size_t)K * G;    // float4 per node row

    using FL = FactorLanes<G, K>;
    constexpr int KP = FL::KP, VPL = FL::VPL;
    const int kb = FL::factor_base(c);
    const bool prim = FL::primary(c);

    float4 zi[K];
#pragma unroll
    for (int k = 0; k < K; ++k) zi[k] = Z4[(size_t)si.grow * rs + k * G + c];
    float sacc[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) sacc[i] = 0.0f;

    // lane l pre-loads entry l of the segment (seg_len <= 64): one coalesced load instead of a
    // dependent load per iteration; groups pick their entry up with a shuffle.
    const int my_col = (si.beg + lane < si.end) ? g.col[si.beg + lane] : si.grow;
    for (int base = si.beg; base < si.end; base += EPW) {
        const int e = base + grp;
        const bool live = e < si.end;
        const int j = __shfl(my_col, e - si.beg, DL_WAVE);
        float part[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) part[k] = k < K ? dot4(zi[k < K ? k : 0], Z4[(size_t)j * rs + (k < K ? k : 0) * G + c]) : 0.0f;
        TransposedReduce<KP, G / 2>::run(part, c);         // this lane now owns factors kb .. kb+VPL-1
        float ex[VPL];
        float mine = 0.0f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            ex[i] = expf(div_t(part[i], t));
            if (prim && kb + i < K) mine += ex[i];
        }
        const float S = group_allreduce_sum<G>(mine);
        float best = 0.0f;
        int win = 255;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const float al = ex[i] / S;
            if (kb + i < K && (win == 255 || beats(al, best))) { best = al; win = kb + i; }
        }
        group_argmax_first<G>(best, win);
        if (live && c == 0) { p[e] = (uint8_t)win; a[e] = best; }
#pragma unroll
        for (int i = 0; i < VPL; ++i) sacc[i] += (live && prim && win == kb + i) ? best : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) sacc[i] = across_groups_sum<G>(sacc[i]);
    if (grp == 0 && prim) {
        float* dst = si.slot < 0 ? s + (size_t)si.grow * K : s_part + (size_t)si.slot * K;
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            if (kb + i < K) dst[kb + i] = sacc[i];
    }
}

// Per multi-segment row: out[grow][k] = f(sum of the K-vectors of its slots, in slot order).
// One wave per row: lane handles factor k = lane % KP of slot (lane / KP), stride 64/KP.
// mode 0: plain sum (s);  mode 1: ds_from_acc(sum, s_raw[grow][k]) (normaliser gradient).
__global__ __launch_bounds__(BLOCK) void vec_combine_kernel(dl_csr_plan g, int K, int KP,
                                                            const float* __restrict__ part, int mode,
                                                            const float* __restrict__ s_raw,
                                                            float* __restrict__ out) {
    const int m = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (m >= g.n_multi) return;
    const int lane = lane_id();
    const int k = lane % KP, sl = lane / KP, step = DL_WAVE / KP;
    float acc = 0.0f;
    if (k < K)
        for (int slot = g.multi_slot0[m] + sl; slot < g.multi_slot0[m + 1]; slot += step)
            acc += part[(size_t)slot * K + k];
    for (int off = KP; off < DL_WAVE; off <<= 1) acc += __shfl_xor(acc, off, DL_WAVE);
    if (lane < K) {
        const size_t o = ((size_t)g.multi_row[m] + g.row_offset) * K + lane;
        out[o] = mode == 0 ? acc : ds_from_acc(acc, s_raw[o]);
    }
}

// ---------------------------------------------------------------------------- aggregate
template <int K, int D>
__global__ __launch_bounds__(BLOCK) void aggregate_seg_kernel(dl_csr_plan g, const float* __restrict__ Z, float beta,
                                                              const uint8_t* __restrict__ p,
                                                              const float* __restrict__ a,
                                                              const float* __restrict__ s, float* __restrict__ H,
                                                              float* __restrict__ h_part) {
    constexpr int G = D / 4;
    constexpr int EPW = DL_WAVE / G;
    const int seg = wave_segment(g);
    if (seg < 0) return;
    const int lane = lane_id();
    const int c = lane % G, grp = lane / G;
    const SegInfo si = load_seg(g, seg);
    const float4* __restrict__ Z4 = reinterpret_cast<const float4*>(Z);
    const size_t rs = (size_t)K * G;

    float4 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = zero4();

    // per-entry scalars are computed once by the entry's own lane, then shuffled to its group
    int my_col = si.grow, my_k = 0;
    float my_w = 0.0f;
    if (si.beg + lane < si.end) {
        my_col = g.col[si.beg + lane];
        my_k = p[si.beg + lane];
        my_w = a[si.beg + lane] / one_if_zero(s[(size_t)my_col * K + my_k]);
    }
    for (int base = si.beg; base < si.end; base += EPW) {
        const int idx = base + grp - si.beg;
        const int j = __shfl(my_col, idx, DL_WAVE);
        const int k = __shfl(my_k, idx, DL_WAVE);
        const float w = __shfl(my_w, idx, DL_WAVE);
        const float4 v = Z4[(size_t)j * rs + k * G + c];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) fma4(acc[kk], (kk == k) ? w : 0.0f, v);
    }
#pragma unroll
    for (int kk = 0; kk < K; ++kk) across_groups_sum4<G>(acc[kk]);
    if (grp == 0) {
        if (si.slot < 0) {
            const float omb = 1.0f - beta;
            float4* __restrict__ H4 = reinterpret_cast<float4*>(H);
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const float4 z = Z4[(size_t)si.grow * rs + kk * G + c];
                float4 h;
                h.x = beta * z.x + omb * acc[kk].x;
                h.y = beta * z.y + omb * acc[kk].y;
                h.z = beta * z.z + omb * acc[kk].z;
                h.w = beta * z.w + omb * acc[kk].w;
                H4[(size_t)si.grow * rs + kk * G + c] = h;
            }
        } else {
            float4* __restrict__ P4 = reinterpret_cast<float4*>(h_part);
#pragma unroll
            for (int kk = 0; kk < K; ++kk) P4[(size_t)si.slot * rs + kk * G + c] = acc[kk];
        }
    }
}

// Per multi-segment row:  out[grow] = (accumulate ? out[grow] : 0) + cx * X[grow] + cp * sum_slots part[slot].
// One 256-thread block per row: each of the 4 waves sums every 4th slot, LDS combines them in
// wave order.  `part` rows are `pstride4` float4 apart, starting at `poff4` (so the dZ and dH
// halves of the scorer backward's partials can be combined separately).
template <int TOT4>
__global__ __launch_bounds__(BLOCK) void row_combine_kernel(dl_csr_plan g, const float* __restrict__ part,
                                                            int pstride4, int poff4, const float* __restrict__ X,
                                                            float cx, float cp, float* __restrict__ out,
                                                            int accumulate) {
    constexpr int NQ = (TOT4 + DL_WAVE - 1) / DL_WAVE;
    __shared__ float4 red[WAVES_PER_BLOCK][NQ * DL_WAVE];
    const int m = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const float4* __restrict__ P4 = reinterpret_cast<const float4*>(part);
    float4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = zero4();
    const int s0 = g.multi_slot0[m], s1 = g.multi_slot0[m + 1];
#pragma unroll 2
    for (int slot = s0 + wave; slot < s1; slot += WAVES_PER_BLOCK) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int x = q * DL_WAVE + lane;
            if (x < TOT4) {
                const float4 v = P4[(size_t)slot * pstride4 + poff4 + x];
                acc[q].x += v.x; acc[q].y += v.y; acc[q].z += v.z; acc[q].w += v.w;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) red[wave][q * DL_WAVE + lane] = acc[q];
    __syncthreads();
    if (wave != 0) return;
    const size_t grow = (size_t)g.multi_row[m] + g.row_offset;
    const float4* __restrict__ X4 = reinterpret_cast<const float4*>(X);
    float4* __restrict__ O4 = reinterpret_cast<float4*>(out);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int x = q * DL_WAVE + lane;
        if (x < TOT4) {
            float4 t = red[0][x];
#pragma unroll
            for (int w = 1; w < WAVES_PER_BLOCK; ++w) {
                const float4 v = red[w][x];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            const float4 xv = X4[grow * TOT4 + x];
            float4 o = accumulate ? O4[grow * TOT4 + x] : zero4();
            o.x += cx * xv.x + cp * t.x;
            o.y += cx * xv.y + cp * t.y;
            o.z += cx * xv.z + cp * t.z;
            o.w += cx * xv.w + cp * t.w;
            O4[grow * TOT4 + x] = o;
        }
    }
}

// ---------------------------------------------------------------------------- backward, phase 1
// dw[e] = (1-b) dH[i][p].Z[j][p] ; dwr[e] = (1-b) dH[j][p].Z[i][p] ; ds[i][k] = -(sum [p=k] dwr a)/s~^2
template <int K, int D>
__global__ __launch_bounds__(BLOCK) void bwd_phase1_seg_kernel(dl_csr_plan g, const float* __restrict__ Z,
                                                               const float* __restrict__ dH, float beta,
                                                               const uint8_t* __restrict__ p,
                                                               const float* __restrict__ a,
                                                               const float* __restrict__ s,
                                                               float* __restrict__ dw, float* __restrict__ dwr,
                                                               float* __restrict__ ds, float* __restrict__ ds_part) {
    constexpr int G = D / 4;
    constexpr int EPW = DL_WAVE / G;
    const int seg = wave_segment(g);
    if (seg < 0) return;
    const int lane = lane_id();
    const int c = lane % G, grp = lane / G;
    const SegInfo si = load_seg(g, seg);
    const float4* __restrict__ Z4 = reinterpret_cast<const float4*>(Z);
    const float4* __restrict__ D4 = reinterpret_cast<const float4*>(dH);
    const size_t rs = (size_t)K * G;
    const float omb = 1.0f - beta;
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
        const int j = __shfl(my_col, e - si.beg, DL_WAVE);
        const int k = __shfl(my_k, e - si.beg, DL_WAVE);
        const float ae = __shfl(my_a, e - si.beg, DL_WAVE);
        const size_t oi = (size_t)si.grow * rs + k * G + c, oj = (size_t)j * rs + k * G + c;
        const float v = omb * group_allreduce_sum<G>(dot4(D4[oi], Z4[oj]));
        const float vr = omb * group_allreduce_sum<G>(dot4(D4[oj], Z4[oi]));
        if (live && c == 0) { dw[e] = v; dwr[e] = vr; }
        const float contrib = live ? vr * ae : 0.0f;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) acc[kk] += (kk == k) ? contrib : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = across_groups_sum<G>(acc[k]);
    if (lane == 0) {
        if (si.slot < 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const size_t o = (size_t)si.grow * K + k;
                ds[o] = ds_from_acc(acc[k], s[o]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) ds_part[(size_t)si.slot * K + k] = acc[k];
        }
    }
}

// ---------------------------------------------------------------------------- backward, phase 2
template <int K, int D>
__global__ __launch_bounds__(BLOCK) void bwd_phase2_seg_kernel(
    dl_csr_plan g, const float* __restrict__ Z, const float* __restrict__ dH, float beta, float t,
    const uint8_t* __restrict__ p, const float* __restrict__ a, const float* __restrict__ s,
    const float* __restrict__ dw, const float* __restrict__ dwr, const float* __restrict__ ds,
    float* __restrict__ dZ, int accumulate, float* __restrict__ dz_part) {
    constexpr int G = D / 4;
    constexpr int EPW = DL_WAVE / G;
    const int seg = wave_segment(g);
    if (seg < 0) return;
    const int lane = lane_id();
    const int c = lane % G, grp = lane / G;
    const SegInfo si = load_seg(g, seg);
    const float4* __restrict__ Z4 = reinterpret_cast<const float4*>(Z);
    const float4* __restrict__ D4 = reinterpret_cast<const float4*>(dH);
    const size_t rs = (size_t)K * G;
    const float omb = 1.0f - beta;

    float4 zi[K], acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        zi[k] = Z4[(size_t)si.grow * rs + k * G + c];
        acc[k] = zero4();
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
        const int e = base + grp;
        const bool live = e < si.end;
        const int j = __shfl(my_col, e - si.beg, DL_WAVE);
        const int k = __shfl(my_k, e - si.beg, DL_WAVE);
        const float cc = __shfl(my_cc, e - si.beg, DL_WAVE);
        const float w2 = __shfl(my_w2, e - si.beg, DL_WAVE);
        float4 zj[K];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) zj[kk] = Z4[(size_t)j * rs + kk * G + c];
        const float4 dhj = D4[(size_t)j * rs + k * G + c];
        float ex[K];
        const float S = edge_exps<K, G>(zi, zj, t, ex);
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const bool hit = kk == k;
            const float ck = live ? cc * ((hit ? 1.0f : 0.0f) - ex[kk] / S) / t : 0.0f;
            fma4(acc[kk], ck, zj[kk]);
            fma4(acc[kk], hit ? w2 : 0.0f, dhj);
        }
    }
#pragma unroll
    for (int kk = 0; kk < K; ++kk) across_groups_sum4<G>(acc[kk]);
    if (grp == 0) {
        if (si.slot < 0) {
            float4* __restrict__ O4 = reinterpret_cast<float4*>(dZ);
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const size_t o = (size_t)si.grow * rs + kk * G + c;
                const float4 dh = D4[o];
                float4 r = accumulate ? O4[o] : zero4();
                r.x += beta * dh.x + acc[kk].x;
                r.y += beta * dh.y + acc[kk].y;
                r.z += beta * dh.z + acc[kk].z;
                r.w += beta * dh.w + acc[kk].w;
                O4[o] = r;
            }
        } else {
            float4* __restrict__ P4 = reinterpret_cast<float4*>(dz_part);
#pragma unroll
            for (int kk = 0; kk < K; ++kk) P4[(size_t)si.slot * rs + kk * G + c] = acc[kk];
        }
    }
}

// ---------------------------------------------------------------------------- pair scorer
// One wave per segment of the "pairs by first endpoint" plan: the u rows of Z and H are staged once
// in LDS, every lane group then scores one pair per iteration from the gathered v rows.
template <int K, int D, bool COEF>
__global__ __launch_bounds__(BLOCK) void score_fwd_seg_kernel(dl_csr_plan g, const int32_t* __restrict__ pair_id,
                                                              const float* __restrict__ Z,
                                                              const float* __restrict__ H, float t,
                                                              float* __restrict__ prob, float* __restrict__ coef_e,
                                                              float* __restrict__ coef_q) {
    constexpr int G = D / 4;
    constexpr int EPW = DL_WAVE / G;
    constexpr int RS = K * G;                       // float4 per node row
    __shared__ float4 urow[WAVES_PER_BLOCK][2 * RS];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int seg = wave_segment(g);
    const bool active = seg >= 0;
    const float4* __restrict__ Z4 = reinterpret_cast<const float4*>(Z);
    const float4* __restrict__ H4 = reinterpret_cast<const float4*>(H);
    SegInfo si{0, 0, 0, 0, -1};
    if (active) {
        si = load_seg(g, seg);
        for (int x = lane; x < RS; x += DL_WAVE) {
            urow[wave][x] = Z4[(size_t)si.grow * RS + x];
            urow[wave][RS + x] = H4[(size_t)si.grow * RS + x];
        }
    }
    __syncthreads();
    if (!active) return;
    const int c = lane % G, grp = lane / G;
    int my_col = si.grow, my_pair = 0;
    if (si.beg + lane < si.end) {
        my_col = g.col[si.beg + lane];
        my_pair = pair_id[si.beg + lane];
    }
    for (int base = si.beg; base < si.end; base += EPW) {
        const int it = base + grp;
        const bool live = it < si.end;
        const size_t v = (size_t)__shfl(my_col, it - si.beg, DL_WAVE);
        const int q = __shfl(my_pair, it - si.beg, DL_WAVE);
        float4 zv[K], hv[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            zv[k] = Z4[v * RS + k * G + c];
            hv[k] = H4[v * RS + k * G + c];
        }
        using FL = FactorLanes<G, K>;
        constexpr int KP = FL::KP, VPL = FL::VPL;
        float pq[KP], ps[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            pq[k] = k < K ? dot4(urow[wave][RS + (k < K ? k : 0) * G + c], hv[k < K ? k : 0]) : 0.0f;
            ps[k] = k < K ? dot4(urow[wave][(k < K ? k : 0) * G + c], zv[k < K ? k : 0]) : 0.0f;
        }
        TransposedReduce<KP, G / 2>::run(pq, c);
        TransposedReduce<KP, G / 2>::run(ps, c);
        const int kb = FL::factor_base(c);
        float term = 0.0f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            if (FL::primary(c) && kb + i < K) {
                const float ek = expf(div_t(ps[i], t));
                const float qe = pq[i] * ek;
                term += qe;
                if (COEF && live) {                             // per-factor logit terms for the backward
                    coef_e[(size_t)q * K + kb + i] = ek;
                    coef_q[(size_t)q * K + kb + i] = qe;
                }
            }
        }
        const float logit = group_allreduce_sum<G>(term);
        if (live && c == 0) prob[q] = sigmoid_ref(logit);
    }
}

// Scorer backward over the node-incidence plan: one wave per segment of node u's pair slots.
// Partials (multi-segment rows) hold [dZ row | dH row] per slot.
template <int K, int D>
__global__ __launch_bounds__(BLOCK) void score_bwd_seg_kernel(dl_csr_plan g, const int32_t* __restrict__ inc_pair,
                                                              const float* __restrict__ Z,
                                                              const float* __restrict__ H, float t,
                                                              const float* __restrict__ prob,
                                                              const float* __restrict__ g_prob,
                                                              float* __restrict__ dZ, float* __restrict__ dH,
                                                              float* __restrict__ part) {
    constexpr int G = D / 4;
    constexpr int EPW = DL_WAVE / G;
    constexpr int RS = K * G;
    __shared__ float4 urow[WAVES_PER_BLOCK][2 * RS];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int seg = wave_segment(g);
    const bool active = seg >= 0;
    const float4* __restrict__ Z4 = reinterpret_cast<const float4*>(Z);
    const float4* __restrict__ H4 = reinterpret_cast<const float4*>(H);
    SegInfo si{0, 0, 0, 0, -1};
    if (active) {
        si = load_seg(g, seg);
        for (int x = lane; x < RS; x += DL_WAVE) {
            urow[wave][x] = Z4[(size_t)si.grow * RS + x];
            urow[wave][RS + x] = H4[(size_t)si.grow * RS + x];
        }
    }
    __syncthreads();
    if (!active) return;
    const int c = lane % G, grp = lane / G;
    float4 accZ[K], accH[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { accZ[k] = zero4(); accH[k] = zero4(); }
    int my_col = si.grow;
    float my_gl = 0.0f;
    if (si.beg + lane < si.end) {
        my_col = g.col[si.beg + lane];
        const int q = inc_pair[si.beg + lane];
        const float pr = prob[q];
        my_gl = g_prob[q] * pr * (1.0f - pr);           // sigmoid backward p(1-p)
    }
    for (int base = si.beg; base < si.end; base += EPW) {
        const int it = base + grp;
        const bool live = it < si.end;
        const size_t v = (size_t)__shfl(my_col, it - si.beg, DL_WAVE);
        const float gl = __shfl(my_gl, it - si.beg, DL_WAVE);
        float4 zv[K], hv[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            zv[k] = Z4[v * RS + k * G + c];
            hv[k] = H4[v * RS + k * G + c];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float qk = group_allreduce_sum<G>(dot4(urow[wave][RS + k * G + c], hv[k]));
            const float ek = expf(group_allreduce_sum<G>(dot4(urow[wave][k * G + c], zv[k])) / t);
            const float ch = live ? gl * ek : 0.0f;
            const float cz = live ? gl * qk * ek / t : 0.0f;
            fma4(accH[k], ch, hv[k]);
            fma4(accZ[k], cz, zv[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) { across_groups_sum4<G>(accZ[k]); across_groups_sum4<G>(accH[k]); }
    if (grp == 0) {
        float4* __restrict__ OZ = reinterpret_cast<float4*>(si.slot < 0 ? dZ : part);
        float4* __restrict__ OH = reinterpret_cast<float4*>(si.slot < 0 ? dH : part);
        const size_t oz = si.slot < 0 ? (size_t)si.grow * RS : (size_t)si.slot * 2 * RS;
        const size_t oh = si.slot < 0 ? (size_t)si.grow * RS : (size_t)si.slot * 2 * RS + RS;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            OZ[oz + k * G + c] = accZ[k];
            OH[oh + k * G + c] = accH[k];
        }
    }
}

// ---------------------------------------------------------------------------- host launchers
static inline int pow2_at_least(int k) {
    int p = 1;
    while (p < k) p <<= 1;
    return p;
}

static void launch_vec_combine(const dl_csr_plan* g, int K, const float* part, int mode, const float* s_raw,
                               float* out, hipStream_t st) {
    if (g->n_multi <= 0) return;
    hipLaunchKernelGGL(vec_combine_kernel, dim3(wave_blocks(g->n_multi)), dim3(BLOCK), 0, st, *g, K, pow2_at_least(K),
                       part, mode, s_raw, out);
}

template <int K, int D>
int route_fwd_t(const dl_csr_plan* g, const float* Z, float t, uint8_t* p, float* a, float* s, float* s_part,
                hipStream_t st) {
    hipLaunchKernelGGL((route_seg_kernel<K, D>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, Z, t, p, a, s,
                       s_part);
    launch_vec_combine(g, K, s_part, 0, nullptr, s, st);
    return check_launch("route_fwd(fast)");
}

template <int K, int D>
int aggregate_fwd_t(const dl_csr_plan* g, const float* Z, float beta, const uint8_t* p, const float* a,
                    const float* s, float* H, float* h_part, hipStream_t st) {
    constexpr int TOT4 = K * D / 4;
    hipLaunchKernelGGL((aggregate_seg_kernel<K, D>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, Z, beta, p,
                       a, s, H, h_part);
    if (g->n_multi > 0)
        hipLaunchKernelGGL((row_combine_kernel<TOT4>), dim3(g->n_multi), dim3(BLOCK), 0, st, *g, h_part, TOT4, 0, Z,
                           beta, 1.0f - beta, H, 0);
    return check_launch("aggregate_fwd(fast)");
}

template <int K, int D>
int bwd_phase1_t(const dl_csr_plan* g, const float* Z, float beta, const uint8_t* p, const float* a, const float* s,
                 const float* dH, float* dw, float* dwr, float* ds, float* ds_part, hipStream_t st) {
    hipLaunchKernelGGL((bwd_phase1_seg_kernel<K, D>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, Z, dH, beta,
                       p, a, s, dw, dwr, ds, ds_part);
    launch_vec_combine(g, K, ds_part, 1, s, ds, st);
    return check_launch("route_aggregate_bwd_phase1(fast)");
}

template <int K, int D>
int bwd_phase2_t(const dl_csr_plan* g, const float* Z, float beta, float t, const uint8_t* p, const float* a,
                 const float* s, const float* dH, const float* dw, const float* dwr, const float* ds, float* dZ,
                 int accumulate, float* dz_part, hipStream_t st) {
    constexpr int TOT4 = K * D / 4;
    hipLaunchKernelGGL((bwd_phase2_seg_kernel<K, D>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, Z, dH, beta,
                       t, p, a, s, dw, dwr, ds, dZ, accumulate, dz_part);
    if (g->n_multi > 0)
        hipLaunchKernelGGL((row_combine_kernel<TOT4>), dim3(g->n_multi), dim3(BLOCK), 0, st, *g, dz_part, TOT4, 0, dH,
                           beta, 1.0f, dZ, accumulate);
    return check_launch("route_aggregate_bwd_phase2(fast)");
}

template <int K, int D>
int score_fwd_t(const dl_pair_incidence* by_u, const float* Z, const float* H, float t, float* prob, float* coef,
                hipStream_t st) {
    const dl_csr_plan* g = &by_u->csr;
    float* coef_q = coef ? coef + (size_t)by_u->n_pairs * K : nullptr;
    if (coef)
        hipLaunchKernelGGL((score_fwd_seg_kernel<K, D, true>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g,
                           by_u->inc_pair, Z, H, t, prob, coef, coef_q);
    else
        hipLaunchKernelGGL((score_fwd_seg_kernel<K, D, false>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g,
                           by_u->inc_pair, Z, H, t, prob, coef, coef_q);
    return check_launch("score_pairs_fwd(fast)");
}

// Scorer backward from stored per-factor terms: a weighted row gather, one launch per output.
//   PASS 0: dZ[u] = sum_inc gl * (q_k e_k) / t * Z[v][k]      PASS 1: dH[u] = sum_inc gl * e_k * H[v][k]
template <int K, int D, int PASS>
__global__ __launch_bounds__(BLOCK) void score_bwd_coef_seg_kernel(dl_csr_plan g, const int32_t* __restrict__ inc_pair,
                                                                   const float* __restrict__ X, float t,
                                                                   const float* __restrict__ prob,
                                                                   const float* __restrict__ g_prob,
                                                                   const float* __restrict__ coef,
                                                                   float* __restrict__ out, float* __restrict__ part) {
    constexpr int G = D / 4;
    constexpr int EPW = DL_WAVE / G;
    constexpr int RS = K * G;
    constexpr int K4 = (K + 3) / 4;
    const int seg = wave_segment(g);
    if (seg < 0) return;
    const int lane = lane_id();
    const int c = lane % G, grp = lane / G;
    const SegInfo si = load_seg(g, seg);
    const float4* __restrict__ X4 = reinterpret_cast<const float4*>(X);
    float4 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = zero4();
    int my_col = si.grow, my_pair = 0;
    float my_gl = 0.0f;
    if (si.beg + lane < si.end) {
        my_col = g.col[si.beg + lane];
        my_pair = inc_pair[si.beg + lane];
        const float pr = prob[my_pair];
        my_gl = g_prob[my_pair] * pr * (1.0f - pr);      // sigmoid backward p(1-p)
        if (PASS == 0) my_gl = div_t(my_gl, t);          // NOTE: (gl/t)*qe instead of gl*qe/t (rounding-level)
    }
    for (int base = si.beg; base < si.end; base += EPW) {
        const int idx = base + grp - si.beg;
        const size_t v = (size_t)__shfl(my_col, idx, DL_WAVE);
        const int q = __shfl(my_pair, idx, DL_WAVE);
        const float gl = __shfl(my_gl, idx, DL_WAVE);     // 0 for lanes past the segment end
        float4 xv[K];
#pragma unroll
        for (int k = 0; k < K; ++k) xv[k] = X4[v * RS + k * G + c];
        float ck[K4 * 4];
        if constexpr (K % 4 == 0) {
            const float4* __restrict__ C4 = reinterpret_cast<const float4*>(coef + (size_t)q * K);
#pragma unroll
            for (int i = 0; i < K4; ++i) {
                const float4 t4 = C4[i];
                ck[4 * i] = t4.x; ck[4 * i + 1] = t4.y; ck[4 * i + 2] = t4.z; ck[4 * i + 3] = t4.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) ck[k] = coef[(size_t)q * K + k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) fma4(acc[k], gl * ck[k], xv[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) across_groups_sum4<G>(acc[k]);
    if (grp == 0) {
        float4* __restrict__ O4 = reinterpret_cast<float4*>(si.slot < 0 ? out : part);
        const size_t o = si.slot < 0 ? (size_t)si.grow * RS : (size_t)si.slot * RS;
#pragma unroll
        for (int k = 0; k < K; ++k) O4[o + k * G + c] = acc[k];
    }
}

template <int K, int D>
int score_bwd_coef_t(const dl_pair_incidence* inc, const float* Z, const float* H, float t, const float* prob,
                     const float* g_prob, const float* coef, float* dZ, float* dH, float* part, hipStream_t st) {
    constexpr int TOT4 = K * D / 4;
    const dl_csr_plan* g = &inc->csr;
    const float* coef_e = coef;
    const float* coef_q = coef + (size_t)inc->n_pairs * K;
    float* part_h = part + (size_t)g->n_slots * K * D;
    hipLaunchKernelGGL((score_bwd_coef_seg_kernel<K, D, 0>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, inc->inc_pair,
                       Z, t, prob, g_prob, coef_q, dZ, part);
    hipLaunchKernelGGL((score_bwd_coef_seg_kernel<K, D, 1>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, inc->inc_pair,
                       H, t, prob, g_prob, coef_e, dH, part_h);
    if (g->n_multi > 0) {
        hipLaunchKernelGGL((row_combine_kernel<TOT4>), dim3(g->n_multi), dim3(BLOCK), 0, st, *g, part, TOT4, 0, Z, 0.0f,
                           1.0f, dZ, 0);
        hipLaunchKernelGGL((row_combine_kernel<TOT4>), dim3(g->n_multi), dim3(BLOCK), 0, st, *g, part_h, TOT4, 0, Z,
                           0.0f, 1.0f, dH, 0);
    }
    return check_launch("score_pairs_bwd(fast, stored terms)");
}

template <int K, int D>
int score_bwd_t(const dl_pair_incidence* inc, const float* Z, const float* H, float t, const float* prob,
                const float* g_prob, float* dZ, float* dH, float* part, hipStream_t st) {
    constexpr int TOT4 = K * D / 4;
    const dl_csr_plan* g = &inc->csr;
    hipLaunchKernelGGL((score_bwd_seg_kernel<K, D>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g,
                       inc->inc_pair, Z, H, t, prob, g_prob, dZ, dH, part);
    if (g->n_multi > 0) {
        // X is unused (cx = 0) but must be a valid row pointer: pass Z
        hipLaunchKernelGGL((row_combine_kernel<TOT4>), dim3(g->n_multi), dim3(BLOCK), 0, st, *g, part, 2 * TOT4, 0, Z,
                           0.0f, 1.0f, dZ, 0);
        hipLaunchKernelGGL((row_combine_kernel<TOT4>), dim3(g->n_multi), dim3(BLOCK), 0, st, *g, part, 2 * TOT4, TOT4,
                           Z, 0.0f, 1.0f, dH, 0);
    }
    return check_launch("score_pairs_bwd(fast)");
}

}  // namespace fast

// (K, D) pairs with a tuned instantiation.  D must be 4 * a power of two <= 256.
#define DL_FAST_SHAPES(X) \
    X(4, 32) X(8, 64) X(16, 128) X(5, 32) X(5, 64) X(10, 32) X(10, 64) X(20, 32) X(8, 32) X(4, 64) X(4, 8) X(8, 8) X(3, 8)

#define DL_DISPATCH(CALL)                                     \
    DL_FAST_SHAPES(CALL)                                      \
    set_error("no tuned kernel for K=%d d=%d", K, d);         \
    return DL_E_ARG;

bool fast_supported(int K, int d) {
#define X(KK, DD) if (K == KK && d == DD) return true;
    DL_FAST_SHAPES(X)
#undef X
    return false;
}

int fast_route_fwd(const dl_csr_plan* g, const float* Z, int K, int d, float t, uint8_t* p, float* a, float* s,
                   float* s_part, hipStream_t st) {
#define X(KK, DD) if (K == KK && d == DD) return fast::route_fwd_t<KK, DD>(g, Z, t, p, a, s, s_part, st);
    DL_DISPATCH(X)
#undef X
}

int fast_aggregate_fwd(const dl_csr_plan* g, const float* Z, int K, int d, float beta, const uint8_t* p,
                       const float* a, const float* s, float* H, float* h_part, hipStream_t st) {
#define X(KK, DD) if (K == KK && d == DD) return fast::aggregate_fwd_t<KK, DD>(g, Z, beta, p, a, s, H, h_part, st);
    DL_DISPATCH(X)
#undef X
}

int fast_bwd_phase1(const dl_csr_plan* g, const float* Z, int K, int d, float beta, const uint8_t* p,
                    const float* a, const float* s, const float* dH, float* dw, float* dwr, float* ds,
                    float* ds_part, hipStream_t st) {
#define X(KK, DD) \
    if (K == KK && d == DD) return fast::bwd_phase1_t<KK, DD>(g, Z, beta, p, a, s, dH, dw, dwr, ds, ds_part, st);
    DL_DISPATCH(X)
#undef X
}

int fast_bwd_phase2(const dl_csr_plan* g, const float* Z, int K, int d, float beta, float t, const uint8_t* p,
                    const float* a, const float* s, const float* dH, const float* dw, const float* dwr,
                    const float* ds, float* dZ, int accumulate, float* dz_part, hipStream_t st) {
#define X(KK, DD)           \
    if (K == KK && d == DD) \
        return fast::bwd_phase2_t<KK, DD>(g, Z, beta, t, p, a, s, dH, dw, dwr, ds, dZ, accumulate, dz_part, st);
    DL_DISPATCH(X)
#undef X
}

int fast_score_pairs_fwd(const dl_pair_incidence* by_u, const float* Z, const float* H, int K, int d, float t,
                         float* prob, float* coef, hipStream_t st) {
#define X(KK, DD) if (K == KK && d == DD) return fast::score_fwd_t<KK, DD>(by_u, Z, H, t, prob, coef, st);
    DL_DISPATCH(X)
#undef X
}

int fast_score_pairs_bwd(const dl_pair_incidence* inc, const float* Z, const float* H, int K, int d, float t,
                         const float* prob, const float* g_prob, float* dZ, float* dH, float* part, hipStream_t st) {
#define X(KK, DD) \
    if (K == KK && d == DD) return fast::score_bwd_t<KK, DD>(inc, Z, H, t, prob, g_prob, dZ, dH, part, st);
    DL_DISPATCH(X)
#undef X
}

int fast_score_pairs_bwd_coef(const dl_pair_incidence* inc, const float* Z, const float* H, int K, int d, float t,
                              const float* prob, const float* g_prob, const float* coef, float* dZ, float* dH,
                              float* part, hipStream_t st) {
#define X(KK, DD) \
    if (K == KK && d == DD) return fast::score_bwd_coef_t<KK, DD>(inc, Z, H, t, prob, g_prob, coef, dZ, dH, part, st);
    DL_DISPATCH(X)
#undef X
}

}  // namespace dl
