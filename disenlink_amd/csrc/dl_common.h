// Shared device helpers and host-side launch plumbing for libdisenlink_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "disenlink_hip.h"

#define DL_WAVE 64

namespace dl {

// ---- error plumbing (host) -------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define DL_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            dl::set_error(__VA_ARGS__);       \
            return DL_E_ARG;                  \
        }                                     \
    } while (0)

constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = WAVES_PER_BLOCK * DL_WAVE;
static inline unsigned wave_blocks(int n) { return (unsigned)((n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK); }
// grid of a segment kernel: n_slices interleaved streams of workgroups, one per column slice
static inline unsigned seg_blocks(const dl_csr_plan* c) {
    return (unsigned)c->n_slices * wave_blocks(c->slice_max_seg);
}

// ---- device helpers ---------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (DL_WAVE - 1); }

// Butterfly all-reduce over the 64 lanes of a wave; every lane ends with the same bits.
__device__ __forceinline__ float wave_allreduce_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, DL_WAVE);
    return v;
}

// All-reduce over aligned groups of G lanes (G a power of two <= 64).
template <int G>
__device__ __forceinline__ float group_allreduce_sum(float v) {
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, DL_WAVE);
    return v;
}

// Sum over the 64/G groups of a wave: lanes with equal (lane % G) are added together.
template <int G>
__device__ __forceinline__ float across_groups_sum(float v) {
#pragma unroll
    for (int off = G; off < DL_WAVE; off <<= 1) v += __shfl_xor(v, off, DL_WAVE);
    return v;
}

// ---- transposed group reduction ------------------------------------------------------------
// Every lane of a G-lane group holds KP partial sums (one per factor, KP = K rounded up to a power
// of two).  A butterfly that HALVES the value count at every exchange leaves each lane with the
// complete sum of VPL = max(1, KP/G) factors after log2(G) steps and KP-ish shuffles in total,
// instead of KP * log2(G) for KP independent all-reduces.  Lane c (0..G-1) ends up with factors
// factor_base(c) .. +VPL-1; when G > KP, DUP = G/KP neighbouring lanes hold the same factor and
// only the "primary" one may contribute to group-wide sums.
constexpr int pow2_ceil(int k) { int p = 1; while (p < k) p <<= 1; return p; }
constexpr int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

template <int G, int K>
struct FactorLanes {
    static constexpr int KP = pow2_ceil(K);
    static constexpr int VPL = KP > G ? KP / G : 1;
    static constexpr int DUP = G > KP ? G / KP : 1;
    static constexpr int DSH = ilog2(DUP);
    __device__ static __forceinline__ int factor_base(int c) { return (c >> DSH) * VPL; }
    __device__ static __forceinline__ bool primary(int c) { return (c & (DUP - 1)) == 0; }
    // lane (inside the group) and slot that hold factor kk
    __host__ __device__ static constexpr int src_lane(int kk) { return (kk / VPL) << DSH; }
    __host__ __device__ static constexpr int src_slot(int kk) { return kk % VPL; }
};

template <int N, int OFF>
struct TransposedReduce {
    static __device__ __forceinline__ void run(float* v, int c) {
        if constexpr (OFF >= 1) {
            if constexpr (N > 1) {
                const bool up = (c & OFF) != 0;
#pragma unroll
                for (int i = 0; i < N / 2; ++i) {
                    const float send = up ? v[i] : v[i + N / 2];
                    const float keep = up ? v[i + N / 2] : v[i];
                    v[i] = keep + __shfl_xor(send, OFF, DL_WAVE);
                }
                TransposedReduce<N / 2, OFF / 2>::run(v, c);
            } else {
                v[0] += __shfl_xor(v[0], OFF, DL_WAVE);
                TransposedReduce<1, OFF / 2>::run(v, c);
            }
        }
    }
};

// group-wide first-max arg-max of (value, index) pairs; lanes without a candidate pass idx = 255.
template <int G>
__device__ __forceinline__ void group_argmax_first(float& best, int& win);

// torch.argmax order on floats: NaN beats everything, otherwise strictly greater wins, so the
// first maximal element is kept when scanning k upward.
__device__ __forceinline__ bool beats(float v, float best) {
    return (v > best) || (v != v && best == best);
}

template <int G>
__device__ __forceinline__ void group_argmax_first(float& best, int& win) {
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off, DL_WAVE);
        const int ow = __shfl_xor(win, off, DL_WAVE);
        const bool take = ow != 255 && (win == 255 || beats(ob, best) || (!beats(best, ob) && ow < win));
        if (take) { best = ob; win = ow; }
    }
}

// x / t exactly as the reference divides; t == 1 (the usual temperature) skips the IEEE division
__device__ __forceinline__ float div_t(float x, float t) { return t == 1.0f ? x : x / t; }

__device__ __forceinline__ float one_if_zero(float s) { return s == 0.0f ? 1.0f : s; }

// sigmoid as ATen's CPU kernel writes it: 1 / (1 + exp(-x)).
__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

// d s_raw -> ds of the normaliser: -(acc) / s~^2, zero where the raw sum was zero (model.py:72).
__device__ __forceinline__ float ds_from_acc(float acc, float s_raw) {
    return s_raw == 0.0f ? 0.0f : -acc / (s_raw * s_raw);
}

struct SegInfo {
    int row, grow, beg, end, slot;
};

__device__ __forceinline__ SegInfo load_seg(const dl_csr_plan& c, int seg) {
    SegInfo s;
    s.row = c.seg_row[seg];
    s.grow = s.row + c.row_offset;
    s.beg = c.seg_beg[seg];
    s.end = c.seg_end[seg];
    s.slot = c.seg_slot[seg];
    return s;
}

// Segment served by this wave, or -1.  Segments are stored slice-major and workgroup b serves column
// slice b % n_slices: workgroups b and b+8 are observed to land on the same XCD, so with 8 slices an
// XCD's L2 only gathers rows of one eighth of the node table.  Placement changes speed only.
__device__ __forceinline__ int wave_segment(const dl_csr_plan& c) {
    const int x = blockIdx.x % c.n_slices;
    const int i = (blockIdx.x / c.n_slices) * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int seg = c.slice_seg0[x] + i;
    return seg < c.slice_seg0[x + 1] ? seg : -1;
}

}  // namespace dl
