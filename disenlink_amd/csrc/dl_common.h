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

// torch.argmax order on floats: NaN beats everything, otherwise strictly greater wins, so the
// first maximal element is kept when scanning k upward.
__device__ __forceinline__ bool beats(float v, float best) {
    return (v > best) || (v != v && best == best);
}

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
