// Tuned forward kernels for gfx950, instantiated per (K, D).
//
// Work decomposition: the graph plan cuts every CSR row into segments of <= seg_len consecutive
// edges; ONE 64-lane wave owns one segment, so a hub row of thousands of edges is spread over
// many waves and CUs while a median row (tens of edges) is a single wave.  Inside a wave, a group
// of G = D/4 lanes owns one edge: lane c of the group holds the float4 chunk c of every factor
// slice, i.e. one neighbour row Z[j] (K*D*4 bytes, contiguous) is fetched by K coalesced
// 16-byte-per-lane loads and 64/G edges are in flight per wave iteration.  The K dot products
// are reduced with log2(G) cross-lane butterflies; the K-way softmax / arg-max is then computed
// redundantly by every lane of the group, so no further exchange is needed.
//
// Rows with one segment write their outputs directly; segments of multi-segment rows write
// per-segment partials (s_part / h_part, in the caller's workspace) that a small combine kernel
// sums in segment order.  No float atomics: results are bitwise reproducible.
#include "dl_common.h"
#include "dl_kernels.h"

namespace dl {
namespace fast {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = WAVES_PER_BLOCK * DL_WAVE;

__device__ __forceinline__ float dot4(const float4& x, const float4& y) {
    return fmaf(x.w, y.w, fmaf(x.z, y.z, fmaf(x.y, y.y, x.x * y.x)));
}

struct SegInfo {
    int row, beg, end, slot;
};

__device__ __forceinline__ SegInfo load_seg(const dl_graph& g, int seg) {
    SegInfo s;
    s.row = g.seg_row[seg];
    s.beg = g.seg_beg[seg];
    const int row_end = g.rowptr[s.row + 1];
    s.end = min(s.beg + g.seg_len, row_end);
    s.slot = g.seg_slot[seg];
    return s;
}

// ---------------------------------------------------------------------------- route
template <int K, int D>
__global__ __launch_bounds__(BLOCK) void route_seg_kernel(dl_graph g, const float* __restrict__ Z, float t,
                                                          uint8_t* __restrict__ p, float* __restrict__ a,
                                                          float* __restrict__ s, float* __restrict__ s_part) {
    constexpr int G = D / 4;            // lanes per edge
    constexpr int EPW = DL_WAVE / G;    // edges per wave iteration
    const int seg = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (seg >= g.n_seg) return;
    const int lane = lane_id();
    const int c = lane % G, grp = lane / G;
    const SegInfo si = load_seg(g, seg);
    const float4* __restrict__ Z4 = reinterpret_cast<const float4*>(Z);
    const size_t rs = (size_t)K * G;    // float4 per node row

    float4 zi[K];
#pragma unroll
    for (int k = 0; k < K; ++k) zi[k] = Z4[(size_t)si.row * rs + k * G + c];
    float sacc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) sacc[k] = 0.0f;

    for (int base = si.beg; base < si.end; base += EPW) {
        const int e = base + grp;
        const bool live = e < si.end;
        const int j = live ? g.col[e] : si.row;
        float4 zj[K];
#pragma unroll
        for (int k = 0; k < K; ++k) zj[k] = Z4[(size_t)j * rs + k * G + c];
        float ex[K];
        float S = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float sig = group_allreduce_sum<G>(dot4(zi[k], zj[k])) / t;
            ex[k] = expf(sig);
            S += ex[k];                                   // sequential in k, like sum(dim=0)
        }
        float best = ex[0] / S;
        int win = 0;
#pragma unroll
        for (int k = 1; k < K; ++k) {
            const float al = ex[k] / S;
            if (beats(al, best)) { best = al; win = k; }
        }
        if (live && c == 0) { p[e] = (uint8_t)win; a[e] = best; }
#pragma unroll
        for (int k = 0; k < K; ++k) sacc[k] += (live && win == k) ? best : 0.0f;
    }
    // lanes with equal c in different groups hold different edges' sums
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int off = G; off < DL_WAVE; off <<= 1) sacc[k] += __shfl_xor(sacc[k], off, DL_WAVE);
    }
    if (lane == 0) {
        float* dst = si.slot < 0 ? s + (size_t)si.row * K : s_part + (size_t)si.slot * K;
#pragma unroll
        for (int k = 0; k < K; ++k) dst[k] = sacc[k];
    }
}

// s[row][k] = sum over the row's segments, in segment order
__global__ void s_combine_kernel(dl_graph g, int K, const float* __restrict__ s_part, float* __restrict__ s) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.n_multi * K) return;
    const int m = idx / K, k = idx - m * K;
    float acc = 0.0f;
    for (int slot = g.multi_slot0[m]; slot < g.multi_slot0[m + 1]; ++slot) acc += s_part[(size_t)slot * K + k];
    s[(size_t)g.multi_row[m] * K + k] = acc;
}

// ---------------------------------------------------------------------------- aggregate
template <int K, int D>
__global__ __launch_bounds__(BLOCK) void aggregate_seg_kernel(dl_graph g, const float* __restrict__ Z, float beta,
                                                              const uint8_t* __restrict__ p,
                                                              const float* __restrict__ a,
                                                              const float* __restrict__ s, float* __restrict__ H,
                                                              float* __restrict__ h_part) {
    constexpr int G = D / 4;
    constexpr int EPW = DL_WAVE / G;
    const int seg = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (seg >= g.n_seg) return;
    const int lane = lane_id();
    const int c = lane % G, grp = lane / G;
    const SegInfo si = load_seg(g, seg);
    const float4* __restrict__ Z4 = reinterpret_cast<const float4*>(Z);
    const size_t rs = (size_t)K * G;

    float4 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int base = si.beg; base < si.end; base += EPW) {
        const int e = base + grp;
        const bool live = e < si.end;
        const int j = live ? g.col[e] : si.row;
        const int k = live ? (int)p[e] : 0;
        const float w = live ? a[e] / one_if_zero(s[(size_t)j * K + k]) : 0.0f;
        const float4 v = Z4[(size_t)j * rs + k * G + c];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const float wk = (kk == k) ? w : 0.0f;
            acc[kk].x = fmaf(wk, v.x, acc[kk].x);
            acc[kk].y = fmaf(wk, v.y, acc[kk].y);
            acc[kk].z = fmaf(wk, v.z, acc[kk].z);
            acc[kk].w = fmaf(wk, v.w, acc[kk].w);
        }
    }
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
#pragma unroll
        for (int off = G; off < DL_WAVE; off <<= 1) {
            acc[kk].x += __shfl_xor(acc[kk].x, off, DL_WAVE);
            acc[kk].y += __shfl_xor(acc[kk].y, off, DL_WAVE);
            acc[kk].z += __shfl_xor(acc[kk].z, off, DL_WAVE);
            acc[kk].w += __shfl_xor(acc[kk].w, off, DL_WAVE);
        }
    }
    if (grp == 0) {
        if (si.slot < 0) {
            const float omb = 1.0f - beta;
            float4* __restrict__ H4 = reinterpret_cast<float4*>(H);
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const float4 z = Z4[(size_t)si.row * rs + kk * G + c];
                float4 h;
                h.x = beta * z.x + omb * acc[kk].x;
                h.y = beta * z.y + omb * acc[kk].y;
                h.z = beta * z.z + omb * acc[kk].z;
                h.w = beta * z.w + omb * acc[kk].w;
                H4[(size_t)si.row * rs + kk * G + c] = h;
            }
        } else {
            float4* __restrict__ P4 = reinterpret_cast<float4*>(h_part);
#pragma unroll
            for (int kk = 0; kk < K; ++kk) P4[(size_t)si.slot * rs + kk * G + c] = acc[kk];
        }
    }
}

// H[row] = beta*Z[row] + (1-beta) * sum of the row's segment partials, in segment order.
// One wave per multi-segment row; lane q handles float4 q, q+64, ... of the K*D row.
template <int TOT4>
__global__ __launch_bounds__(BLOCK) void h_combine_kernel(dl_graph g, const float* __restrict__ Z, float beta,
                                                          const float* __restrict__ h_part, float* __restrict__ H) {
    constexpr int NQ = (TOT4 + DL_WAVE - 1) / DL_WAVE;
    const int m = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (m >= g.n_multi) return;
    const int lane = lane_id();
    const int row = g.multi_row[m];
    const float4* __restrict__ P4 = reinterpret_cast<const float4*>(h_part);
    const float4* __restrict__ Z4 = reinterpret_cast<const float4*>(Z);
    float4* __restrict__ H4 = reinterpret_cast<float4*>(H);
    float4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int s0 = g.multi_slot0[m], s1 = g.multi_slot0[m + 1];
#pragma unroll 4
    for (int slot = s0; slot < s1; ++slot) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int x = q * DL_WAVE + lane;
            if (x < TOT4) {
                const float4 v = P4[(size_t)slot * TOT4 + x];
                acc[q].x += v.x; acc[q].y += v.y; acc[q].z += v.z; acc[q].w += v.w;
            }
        }
    }
    const float omb = 1.0f - beta;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int x = q * DL_WAVE + lane;
        if (x < TOT4) {
            const float4 z = Z4[(size_t)row * TOT4 + x];
            float4 h;
            h.x = beta * z.x + omb * acc[q].x;
            h.y = beta * z.y + omb * acc[q].y;
            h.z = beta * z.z + omb * acc[q].z;
            h.w = beta * z.w + omb * acc[q].w;
            H4[(size_t)row * TOT4 + x] = h;
        }
    }
}

static inline unsigned wave_blocks(int n) { return (unsigned)((n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK); }

template <int K, int D>
int route_fwd_t(const dl_graph* g, const float* Z, float t, uint8_t* p, float* a, float* s, float* s_part,
                hipStream_t st) {
    hipLaunchKernelGGL((route_seg_kernel<K, D>), dim3(wave_blocks(g->n_seg)), dim3(BLOCK), 0, st, *g, Z, t, p, a, s,
                       s_part);
    if (g->n_multi > 0) {
        const int n = g->n_multi * K;
        hipLaunchKernelGGL(s_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, st, *g, K, s_part, s);
    }
    return check_launch("route_fwd(fast)");
}

template <int K, int D>
int aggregate_fwd_t(const dl_graph* g, const float* Z, float beta, const uint8_t* p, const float* a,
                    const float* s, float* H, float* h_part, hipStream_t st) {
    hipLaunchKernelGGL((aggregate_seg_kernel<K, D>), dim3(wave_blocks(g->n_seg)), dim3(BLOCK), 0, st, *g, Z, beta, p,
                       a, s, H, h_part);
    if (g->n_multi > 0)
        hipLaunchKernelGGL((h_combine_kernel<K * D / 4>), dim3(wave_blocks(g->n_multi)), dim3(BLOCK), 0, st, *g, Z,
                           beta, h_part, H);
    return check_launch("aggregate_fwd(fast)");
}

}  // namespace fast

// (K, D) pairs with a tuned instantiation.  D must be 4 * a power of two <= 256.
#define DL_FAST_SHAPES(X) \
    X(4, 32) X(8, 64) X(16, 128) X(5, 32) X(5, 64) X(10, 32) X(10, 64) X(20, 32) X(8, 32) X(4, 64) X(4, 8) X(8, 8) X(3, 8)

bool fast_supported(int K, int d) {
#define X(KK, DD) if (K == KK && d == DD) return true;
    DL_FAST_SHAPES(X)
#undef X
    return false;
}

int fast_route_fwd(const dl_graph* g, const float* Z, int K, int d, float t, uint8_t* p, float* a, float* s,
                   float* s_part, hipStream_t st) {
#define X(KK, DD) if (K == KK && d == DD) return fast::route_fwd_t<KK, DD>(g, Z, t, p, a, s, s_part, st);
    DL_FAST_SHAPES(X)
#undef X
    set_error("no tuned route kernel for K=%d d=%d", K, d);
    return DL_E_ARG;
}

int fast_aggregate_fwd(const dl_graph* g, const float* Z, int K, int d, float beta, const uint8_t* p,
                       const float* a, const float* s, float* H, float* h_part, hipStream_t st) {
#define X(KK, DD) if (K == KK && d == DD) return fast::aggregate_fwd_t<KK, DD>(g, Z, beta, p, a, s, H, h_part, st);
    DL_FAST_SHAPES(X)
#undef X
    set_error("no tuned aggregate kernel for K=%d d=%d", K, d);
    return DL_E_ARG;
}

}  // namespace dl
