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

// torch.argmax order on floats: NaN beats everything, otherwise strictly greater wins, so the
// first maximal element is kept when scanning k upward.
__device__ __forceinline__ bool beats(float v, float best) {
    return (v > best) || (v != v && best == best);
}

__device__ __forceinline__ float one_if_zero(float s) { return s == 0.0f ? 1.0f : s; }

// sigmoid as ATen's CPU kernel writes it: 1 / (1 + exp(-x)).
__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace dl
