// Shared device helpers and host-side launch plumbing for libdisenlink_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "disenlink_hip.h"
#include "dl_config.h"

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

// ---- cross-lane exchange without the LDS pipe ----------------------------------------------------------------
// HIP's __shfl_xor compiles to ds_bpermute_b32 whatever the mask: every butterfly step of every reduction then goes
// through the LDS crossbar, and the segment kernels (88 of them per wave in the first aggregate kernel, 22 per loop
// iteration in the router and the scorer) became LDS-issue-bound (SQ_WAIT_INST_LDS 22 % of the wave cycles,
// profiles/).  A constant xor mask needs none of that on gfx950: masks 1 / 2 / 4 / 8 are DPP modifiers inside a row of
// 16 lanes (quad_perm, row_half_mirror + quad reverse, row_ror:8), masks 16 / 32 are v_permlane16_swap /
// v_permlane32_swap — all VALU, and a DPP move folds into the add that consumes it.  Every lane must be active.
template <int OFF>
__device__ __forceinline__ int xor_lane_i(int x) {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "xor mask must be a power of two < 64");
    if constexpr (OFF == 1) {
        return __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, true);            // quad_perm:[1,0,3,2]
    } else if constexpr (OFF == 2) {
        return __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, true);            // quad_perm:[2,3,0,1]
    } else if constexpr (OFF == 4) {
        const int y = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, true);    // row_half_mirror: i -> 7 - i
        return __builtin_amdgcn_update_dpp(y, y, 0x1B, 0xF, 0xF, true);            // quad_perm:[3,2,1,0]: -> i ^ 4
    } else if constexpr (OFF == 8) {
        return __builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, true);           // row_ror:8
    } else if constexpr (OFF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);       // r[0] = rows [0,0,2,2], r[1] = rows [1,1,3,3]
        return (threadIdx.x & 16) ? (int)r[0] : (int)r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);       // r[0] = lower half twice, r[1] = upper half twice
        return (threadIdx.x & 32) ? (int)r[0] : (int)r[1];
    }
}
template <int OFF>
__device__ __forceinline__ float xor_lane(float v) { return __int_as_float(xor_lane_i<OFF>(__float_as_int(v))); }

// v + (v of lane ^ OFF): for the two wide masks the swap already delivers both addends to every lane
template <int OFF>
__device__ __forceinline__ float add_xor(float v) {
    if constexpr (OFF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
        return __int_as_float((int)r[0]) + __int_as_float((int)r[1]);
    } else if constexpr (OFF == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
        return __int_as_float((int)r[0]) + __int_as_float((int)r[1]);
    } else {
        return v + xor_lane<OFF>(v);
    }
}

// Value of lane `src` (0 .. G-1, a constant once the caller's loop is unrolled) of this lane's aligned group of G lanes.
// A group of 16 lanes is a DPP row: row_newbcast:src broadcasts inside it as a modifier of a plain VALU move — no trip
// through the LDS crossbar (HIP's __shfl is ds_bpermute_b32: the scorer backward did 16 of them per loop iteration).
template <int CTRL>
__device__ __forceinline__ float dpp_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int G>
__device__ __forceinline__ float group_bcast(float v, int src) {
    if constexpr (G == 16) {
        switch (src) {
            case 0: return dpp_bcast<0x150>(v);   case 1: return dpp_bcast<0x151>(v);   case 2: return dpp_bcast<0x152>(v);
            case 3: return dpp_bcast<0x153>(v);   case 4: return dpp_bcast<0x154>(v);   case 5: return dpp_bcast<0x155>(v);
            case 6: return dpp_bcast<0x156>(v);   case 7: return dpp_bcast<0x157>(v);   case 8: return dpp_bcast<0x158>(v);
            case 9: return dpp_bcast<0x159>(v);   case 10: return dpp_bcast<0x15A>(v);  case 11: return dpp_bcast<0x15B>(v);
            case 12: return dpp_bcast<0x15C>(v);  case 13: return dpp_bcast<0x15D>(v);  case 14: return dpp_bcast<0x15E>(v);
            default: return dpp_bcast<0x15F>(v);
        }
    } else {
        return __shfl(v, (int)((threadIdx.x & (DL_WAVE - 1)) & ~(G - 1)) + src, DL_WAVE);
    }
}

// butterfly v += xor(v, OFF) for OFF = HI, HI/2, ..., LO (compile-time recursion: the masks must be constants)
template <int HI, int LO>
__device__ __forceinline__ float butterfly_down(float v) {
    if constexpr (HI >= LO && HI >= 1) {
        return butterfly_down<HI / 2, LO>(add_xor<HI>(v));
    } else {
        return v;
    }
}
template <int LO, int HI>
__device__ __forceinline__ float butterfly_up(float v) {
    if constexpr (LO <= HI) {
        return butterfly_up<LO * 2, HI>(add_xor<LO>(v));
    } else {
        return v;
    }
}

// ---- hand-off between units of a row that run on DIFFERENT XCDs, inside one launch -------------------------------------
// The XCD L2s are not coherent with each other.  The form used here is the one MI355X_MICROARCH.md lists as measured
// for "each storing wave for itself": the producer wave stores every byte with `sc1` (written through to memory, not
// kept in its L2), waits for its stores (s_waitcnt vmcnt(0)), then adds 1 to an agent-scope counter; the wave whose add
// RETURNED the last count reads every byte with `sc1` loads (never served by its L1).  One wave on either side: no
// workgroup barrier is involved.  hipcc does not count inline-asm memory operations in its own s_waitcnt bookkeeping,
// so the waits are written out and the loaded registers are tied to the wait.
typedef float dl_vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4_sc1(float* p, const float4& v) {
    const dl_vf4 t = {v.x, v.y, v.z, v.w};
    // (s_nop: the hazard recogniser does not see inside inline asm — a VALU write of the data registers right behind a store
    // of more than 8 bytes needs a wait state; without it the next address computation landed in the stored row)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ dl_vf4 load4_sc1_issue(const float* p) {          // valid only behind wait_loads_sc1()
    dl_vf4 t;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t) : "v"(p) : "memory");
    return t;
}
__device__ __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
__device__ __forceinline__ void wait_loads_sc1(dl_vf4& a, dl_vf4& b, dl_vf4& c, dl_vf4& d) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}

// Butterfly all-reduce over the 64 lanes of a wave; every lane ends with the same bits.
__device__ __forceinline__ float wave_allreduce_sum(float v) { return butterfly_down<32, 1>(v); }

// All-reduce over aligned groups of G lanes (G a power of two <= 64).
template <int G>
__device__ __forceinline__ float group_allreduce_sum(float v) {
    return butterfly_down<G / 2, 1>(v);
}

// Sum over the 64/G groups of a wave: lanes with equal (lane % G) are added together.
template <int G>
__device__ __forceinline__ float across_groups_sum(float v) {
    return butterfly_up<G, 32>(v);
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
                    v[i] = keep + xor_lane<OFF>(send);
                }
                TransposedReduce<N / 2, OFF / 2>::run(v, c);
            } else {
                v[0] = add_xor<OFF>(v[0]);
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

template <int OFF>
__device__ __forceinline__ void argmax_step(float& best, int& win) {
    if constexpr (OFF >= 1) {
        const float ob = xor_lane<OFF>(best);
        const int ow = xor_lane_i<OFF>(win);
        const bool take = ow != 255 && (win == 255 || beats(ob, best) || (!beats(best, ob) && ow < win));
        if (take) { best = ob; win = ow; }
        argmax_step<OFF / 2>(best, win);
    }
}
template <int G>
__device__ __forceinline__ void group_argmax_first(float& best, int& win) {
    argmax_step<G / 2>(best, win);
}

// x / t exactly as the reference divides; t == 1 (the usual temperature) skips the IEEE division
// (t is wave-uniform — a kernel argument.  Written as `t == 1 ? x : x / t` hipcc computed the ten-instruction IEEE division
// on every call and selected afterwards; the empty asm cannot be speculated, so the division sits behind a scalar branch.)
__device__ __forceinline__ float div_t(float x, float t) {
    if (t != 1.0f) {
        x = x / t;
        asm volatile("" : "+v"(x));
    }
    return x;
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

static_assert(WAVES_PER_BLOCK == DL_UNIT_SEGS, "a workgroup serves DL_UNIT_SEGS segment positions, one per wavefront");

// Wave index inside the workgroup as a scalar (every lane of a wave has the same value; the compiler cannot know).
__device__ __forceinline__ int wave_index() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// What a wave needs to know about its segment and its UNIT (the aligned run of segments of one row inside the
// workgroup, summed on chip).  Everything comes from scalar loads, and from as few DEPENDENT rounds of them as possible
// — these kernels run tens of microseconds on small graphs and every round trip to memory in a wave's prologue shows:
// round 1 = the stream bounds (skipped for unsliced plans: they are [0, n_seg)), round 2 = the DL_UNIT_SEGS rows of the
// workgroup and the wave's own beg / end / slot, all unconditional (a clamped, valid position; selects afterwards).
//
// Positions are stored slice-major and workgroup b serves column slice b % n_slices: workgroups b and b+8 are observed
// to land on the same XCD, so with 8 slices an XCD's L2 only gathers rows of one eighth of the node table.  Placement
// changes speed only.  Every slice stream holds a multiple of DL_UNIT_SEGS positions.
struct WaveSeg {
    SegInfo si;
    bool active;      // a real segment (not padding, not past the end)
    bool head;        // first wave of its unit: sums the unit and writes its result
    int wave;         // wave index in the workgroup
    int n_unit;       // number of waves (segments) of the unit from this wave on (head: the whole unit)
    int upos;         // position of this wave inside its unit (head: 0)
};

__device__ __forceinline__ WaveSeg load_wave_seg(const dl_csr_plan& c) {
    WaveSeg w;
    w.wave = wave_index();
    int s0 = 0, s1 = c.n_seg, wg = (int)blockIdx.x;
    if (c.n_slices > 1) {
        const int x = blockIdx.x % c.n_slices;
        s0 = c.slice_seg0[x];
        s1 = c.slice_seg0[x + 1];
        wg = (int)(blockIdx.x / c.n_slices);
    }
    const int pos0 = s0 + wg * WAVES_PER_BLOCK;
    const bool in = pos0 < s1;
    const int q = in ? pos0 : 0;                                   // launches have n_seg >= DL_UNIT_SEGS
    // a unit = a run of positions with the same row AND the same partial slot (plans built with one segment per unit
    // give the segments of a multi-segment row different slots: each is then its own unit)
    int rows[WAVES_PER_BLOCK], slots[WAVES_PER_BLOCK];
#pragma unroll
    for (int u = 0; u < WAVES_PER_BLOCK; ++u) {
        rows[u] = c.seg_row[q + u];
        slots[u] = c.seg_slot[q + u];
    }
    const int beg = c.seg_beg[q + w.wave], end = c.seg_end[q + w.wave];
    int mine = -1, prev = -1, slot = -1, prev_slot = -1;
#pragma unroll
    for (int u = 0; u < WAVES_PER_BLOCK; ++u) {
        if (!in) rows[u] = -1;
        if (u == w.wave) { mine = rows[u]; slot = slots[u]; }
        if (u + 1 == w.wave) { prev = rows[u]; prev_slot = slots[u]; }
    }
    w.active = mine >= 0;
    w.head = w.active && (w.wave == 0 || prev != mine || prev_slot != slot);
    w.n_unit = 0;
    bool run = true;
#pragma unroll
    for (int u = 0; u < WAVES_PER_BLOCK; ++u) {
        if (u >= w.wave) {
            run = run && rows[u] == mine && slots[u] == slot;
            w.n_unit += run ? 1 : 0;
        }
    }
    w.upos = 0;
    bool back = true;
#pragma unroll
    for (int u = WAVES_PER_BLOCK - 1; u >= 0; --u) {
        if (u < w.wave) {
            back = back && rows[u] == mine && slots[u] == slot;
            w.upos += back ? 1 : 0;
        }
    }
    w.si.row = mine;
    w.si.grow = mine + c.row_offset;
    w.si.beg = w.active ? beg : 0;
    w.si.end = w.active ? end : 0;
    w.si.slot = w.active ? slot : -1;
    return w;
}

}  // namespace dl
