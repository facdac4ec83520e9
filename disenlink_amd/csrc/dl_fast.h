// Shared pieces of the tuned gfx950 kernels (dl_route.hip, dl_aggregate.hip, dl_bwd.hip, dl_score.hip, dl_train.hip):
// typed 16-byte chunks, the lane geometry, the unit reduction through LDS, the row combine, shape lists and dispatch.
//
//
// Work decomposition: the plan cuts every CSR row into segments of <= seg_len (<= 64) consecutive
// entries; ONE 64-lane wave owns one segment, so a hub row of thousands of edges is spread over
// many waves and CUs while a median row (tens of edges) is a single wave.  Inside a wave, a group
// of G = D/4 lanes owns one entry: lane c of the group holds elements 4c..4c+3 (16 bytes fp32, 8 bytes
// bf16) of every factor slice, i.e. one neighbour row Z[j] (contiguous in HBM) is fetched by K
// coalesced loads per lane and 64/G entries are in flight per wave iteration.
//
// Per-entry scalars (column, routing factor, weights) are loaded once per segment, one entry per
// lane, and handed to the groups by shuffles: no dependent index load inside the loop.
//
// The K per-factor dot products are reduced with a TRANSPOSED butterfly (dl_common.h): after
// log2(G) exchange steps every lane owns the complete dot product of one factor, so exp, softmax
// weight and the per-factor terms are computed once per factor, not once per lane.
//
// A workgroup (4 waves) serves 4 consecutive segment positions; the segments of a row sit in aligned
// runs (UNITS, dl_csr_plan) that the workgroup sums on chip through LDS, in segment order.  Rows of one
// unit (<= 4 segments) write their outputs directly; only rows with several units write per-unit
// partials (fp32, in the caller's workspace) that a combine kernel sums in unit order.  No float
// atomics anywhere: results are bitwise reproducible, and independent of how the rows are sharded.
//
// Tables Z and H may be stored as fp32 or bf16 (dl_dtype); all arithmetic and all gradients are fp32.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "dl_common.h"
#include "dl_kernels.h"

namespace dl {
namespace fast {

// ---------------------------------------------------------------------------- typed 16-byte chunks
typedef unsigned short bf16_t;      // raw bf16 bits

template <int VEC>
struct Chunk {
    float v[VEC];
};

__device__ __forceinline__ float bf16_to_f32(unsigned int hi16) { return __uint_as_float(hi16 << 16); }
__device__ __forceinline__ unsigned int f32_to_bf16(float f) {   // round to nearest even; NaN stays NaN
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <typename T>
struct Tab;
template <>
struct Tab<float> {
    static constexpr int VEC = 4;
    static __device__ __forceinline__ Chunk<4> load(const float* p) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        return Chunk<4>{{q.x, q.y, q.z, q.w}};
    }
    static __device__ __forceinline__ void store(float* p, const Chunk<4>& c) {
        *reinterpret_cast<float4*>(p) = make_float4(c.v[0], c.v[1], c.v[2], c.v[3]);
    }
};
// bf16 tables keep the fp32 lane geometry (4 elements per lane, 8-byte loads): the same registers
// per lane as the fp32 kernels, half the bytes per gathered row.  (8 elements per lane was tried:
// it doubles the fp32 working set per lane and halves the occupancy.)
template <>
struct Tab<bf16_t> {
    static constexpr int VEC = 4;
    static __device__ __forceinline__ Chunk<4> load(const bf16_t* p) {
        const uint2 q = *reinterpret_cast<const uint2*>(p);
        return Chunk<4>{{bf16_to_f32(q.x & 0xffffu), bf16_to_f32(q.x >> 16), bf16_to_f32(q.y & 0xffffu),
                         bf16_to_f32(q.y >> 16)}};
    }
    static __device__ __forceinline__ void store(bf16_t* p, const Chunk<4>& c) {
        uint2 q;
        q.x = f32_to_bf16(c.v[0]) | (f32_to_bf16(c.v[1]) << 16);
        q.y = f32_to_bf16(c.v[2]) | (f32_to_bf16(c.v[3]) << 16);
        *reinterpret_cast<uint2*>(p) = q;
    }
};

// fp32 arrays (gradients, partials, LDS) accessed VEC elements at a time
template <int VEC>
__device__ __forceinline__ Chunk<VEC> load_f32(const float* p) {
    Chunk<VEC> c;
#pragma unroll
    for (int i = 0; i < VEC; i += 4) {
        const float4 q = *reinterpret_cast<const float4*>(p + i);
        c.v[i] = q.x; c.v[i + 1] = q.y; c.v[i + 2] = q.z; c.v[i + 3] = q.w;
    }
    return c;
}
template <int VEC>
__device__ __forceinline__ void store_f32(float* p, const Chunk<VEC>& c) {
#pragma unroll
    for (int i = 0; i < VEC; i += 4)
        *reinterpret_cast<float4*>(p + i) = make_float4(c.v[i], c.v[i + 1], c.v[i + 2], c.v[i + 3]);
}
template <int VEC>
__device__ __forceinline__ Chunk<VEC> zero_chunk() {
    Chunk<VEC> c;
#pragma unroll
    for (int i = 0; i < VEC; ++i) c.v[i] = 0.0f;
    return c;
}
template <int VEC>
__device__ __forceinline__ float dot(const Chunk<VEC>& x, const Chunk<VEC>& y) {
    float r = x.v[0] * y.v[0];
#pragma unroll
    for (int i = 1; i < VEC; ++i) r = fmaf(x.v[i], y.v[i], r);
    return r;
}
template <int VEC>
__device__ __forceinline__ void fma_chunk(Chunk<VEC>& acc, float w, const Chunk<VEC>& x) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.v[i] = fmaf(w, x.v[i], acc.v[i]);
}
template <int G, int VEC>
__device__ __forceinline__ void across_groups_sum_chunk(Chunk<VEC>& c) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) c.v[i] = across_groups_sum<G>(c.v[i]);
}

template <int K, int D, typename T>
struct Geo {
    static constexpr int VEC = Tab<T>::VEC;
    static constexpr int G = D / VEC;             // lanes per entry
    static constexpr int EPW = DL_WAVE / G;       // entries per wave iteration
    static constexpr int ROW = K * D;             // elements per node row
    using FL = FactorLanes<G, K>;
    static_assert(D % VEC == 0 && (G & (G - 1)) == 0 && G <= DL_WAVE, "D must be VEC * a power of two <= 64");
};

// The per-entry scalar (column, pair id, ...) of the entry this lane's GROUP works on in the current iteration: entry
// i0 + grp of the segment, held by lane i0 + grp (one entry per lane, loaded once per segment).  i0 is wave-uniform, so
// with up to four groups per wave (d >= 64) it is four v_readlane with a scalar lane index and three selects — all VALU /
// SALU; HIP's __shfl is ds_bpermute_b32, a trip through the LDS crossbar whose latency sits in front of every
// iteration's row gathers (the address needs the column).  More groups (d <= 32): the permute.
template <int EPW>
__device__ __forceinline__ int entry_scalar(int v, int i0, int grp) {
    if constexpr (EPW <= 4) {
        int r = __builtin_amdgcn_readlane(v, i0 & 63);
#pragma unroll
        for (int g = 1; g < EPW; ++g) {
            const int o = __builtin_amdgcn_readlane(v, (i0 + g) & 63);
            r = grp == g ? o : r;
        }
        return r;
    } else {
        return __shfl(v, i0 + grp, DL_WAVE);
    }
}
template <int EPW>
__device__ __forceinline__ float entry_scalar(float v, int i0, int grp) {
    return __int_as_float(entry_scalar<EPW>(__float_as_int(v), i0, grp));
}

// Per-lane softmax pieces of one entry after the transposed reduce: this lane owns factors
// kb .. kb+VPL-1; ex[i] = exp(sigma/t); S = sum over all K factors (group-wide).
template <int K, int G>
__device__ __forceinline__ float lane_exps(float* part, int c, float t, float (&ex)[FactorLanes<G, K>::VPL]) {
    using FL = FactorLanes<G, K>;
    TransposedReduce<FL::KP, G / 2>::run(part, c);
    const int kb = FL::factor_base(c);
    float mine = 0.0f;
#pragma unroll
    for (int i = 0; i < FL::VPL; ++i) {
        ex[i] = expf(div_t(part[i], t));
        if (FL::primary(c) && kb + i < K) mine += ex[i];
    }
    return group_allreduce_sum<G>(mine);
}

// 4 consecutive elements of a table (fp32 or bf16 storage) as a float4
template <typename T>
__device__ __forceinline__ float4 load4(const T* p);
template <>
__device__ __forceinline__ float4 load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ float4 load4<bf16_t>(const bf16_t* p) {
    const uint2 q = *reinterpret_cast<const uint2*>(p);
    return make_float4(bf16_to_f32(q.x & 0xffffu), bf16_to_f32(q.x >> 16), bf16_to_f32(q.y & 0xffffu),
                       bf16_to_f32(q.y >> 16));
}
__device__ __forceinline__ void store4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, const float4& v) {
    uint2 q;
    q.x = f32_to_bf16(v.x) | (f32_to_bf16(v.y) << 16);
    q.y = f32_to_bf16(v.z) | (f32_to_bf16(v.w) << 16);
    *reinterpret_cast<uint2*>(p) = q;
}
// Streaming (non-temporal) store: the line is not kept in the caches for re-use.  For an output table far larger than the
// caches (the H rows of the aggregation where HBM binds) that leaves the L2 / Infinity Cache to the gathered slices and
// the normalisers: snap-patents x0.25 aggregation 789 -> 750 us.  On cache-resident graphs the next kernel WANTS the rows
// in cache (the scorer reads H right away): the caller decides per launch.
__device__ __forceinline__ void store4_stream(float* p, const float4& v) {
    typedef float vf4 __attribute__((ext_vector_type(4)));
    vf4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<vf4*>(p));
}
__device__ __forceinline__ void store4_stream(bf16_t* p, const float4& v) {
    typedef unsigned int vu2 __attribute__((ext_vector_type(2)));
    vu2 q = {f32_to_bf16(v.x) | (f32_to_bf16(v.y) << 16), f32_to_bf16(v.z) | (f32_to_bf16(v.w) << 16)};
    __builtin_nontemporal_store(q, reinterpret_cast<vu2*>(p));
}

// 4-element dot product as two packed operations and one add (v_pk_mul_f32, v_pk_fma_f32: two lanes of fp32 per
// instruction on gfx950) instead of a chain of four; symmetric in its arguments, so both endpoints of a pair still
// compute the same bits.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot4_packed(const float4& a, const float4& b) {
    const v2f a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
    const v2f p = __builtin_elementwise_fma(a1, b1, a0 * b0);
    return p.x + p.y;
}

// A table row whose index is WAVE-UNIFORM (the wave-per-entry kernels: the 64 lanes share one gathered row), as a pointer
// the compiler keeps in SGPRs and knows to be global: `row[(unsigned)(lane + j * 64)]` then compiles to
//     global_load_dwordx4 v[..], v_lane_off, s[base:base+1] offset:1024 j
// — one 32-bit lane offset register for every load of the kernel and the row base in two SGPRs.  (The empty asm keeps the
// byte offset a scalar that is not folded into a hoisted per-table vector base.  Round 6: the earlier spelling passed the
// POINTER through the empty asm, which hid its address space — the gathers came out as flat_load_dwordx4 with a 64-bit
// vector address each, counted on lgkmcnt as well as vmcnt.)
template <typename V>
__device__ __forceinline__ const __attribute__((address_space(1))) V* uniform_row(const void* table, size_t byte_off) {
    const __attribute__((address_space(1))) char* row =
        reinterpret_cast<const __attribute__((address_space(1))) char*>((const __attribute__((address_space(1))) void*)table) + byte_off;
    asm volatile("" : "+s"(row));        // a scalar the optimiser cannot take apart again — and, typed, still a GLOBAL pointer behind it
    return reinterpret_cast<const __attribute__((address_space(1))) V*>(row);
}
__device__ __forceinline__ float4 as_float4(dl_vf4 v) { return make_float4(v.x, v.y, v.z, v.w); }

// ---------------------------------------------------------------------------- unit reduction through LDS
// A segment kernel ends with per-GROUP partial results: lane c of group g holds elements kk*D + c*VEC .. of every factor
// kk, summed over the entries its group walked.  Adding the 64/G groups of a wave with cross-lane butterflies costs
// 2 log2(64/G) moves + adds per VALUE (K*VEC of them, twice that in the scorer backward): a third of all vector
// instructions of these kernels.  Instead every group stages its partial row in the wave's LDS region and, after the
// workgroup barrier, the head wave of each unit adds groups and segments straight out of LDS — in a fixed order
// (segment by segment, group 0 .. NG-1 inside) — with all 64 lanes at work: lane l ends up with the float4s
// x = q*64 + l of the row.  Where the staged rows would not leave room for two workgroups per CU (K = 16, d = 128 in the
// scorer backward) the groups are added in registers first and only group 0 is staged.
template <int K, int D, int VEC>
__device__ __forceinline__ void stage_row(float* dst, const Chunk<VEC> (&acc)[K], int c) {
#pragma unroll
    for (int kk = 0; kk < K; ++kk) store_f32<VEC>(dst + kk * D + c * VEC, acc[kk]);
}

template <int K, int D, typename T, int NROWS, bool NO_GROUP_ROWS = false>
struct Stage {
    using GE = Geo<K, D, T>;
    static constexpr int VEC = GE::VEC, G = GE::G;
    static constexpr int NG = DL_WAVE / G;                         // lane groups per wave
    static constexpr int ROWF = NROWS * GE::ROW;                   // floats of one wave's result
    // Measured (profiles/r2p vs r2m): the two-row scorer backward gains 5 % from staging the groups (64 cross-lane sums
    // fewer per wave); the one-row kernels do not — their waves are short, and on low-degree graphs (snap-patents-shaped:
    // ~8 entries per row) writing four group rows per wave instead of one made the aggregate kernel 29 % slower.
    static constexpr bool GROUPS_IN_LDS = !NO_GROUP_ROWS && NROWS == 2 && (size_t)WAVES_PER_BLOCK * NG * ROWF * sizeof(float) <= 64 * 1024;
    static constexpr int SG = GROUPS_IN_LDS ? NG : 1;              // group rows staged per wave
    static constexpr int FLOATS = WAVES_PER_BLOCK * SG * ROWF;     // LDS floats of the workgroup
    static constexpr int F4 = ROWF / 4;
    static constexpr int NQ = (F4 + DL_WAVE - 1) / DL_WAVE;
    static_assert(ROWF % 4 == 0, "row length must be a multiple of 4 floats");

    // this wave's region: [SG][ROWF] floats
    static __device__ __forceinline__ float* region(float* red, int wave) { return red + (size_t)wave * SG * ROWF; }

    // stage result row `r` (0 .. NROWS-1) of this lane's group
    static __device__ __forceinline__ void put(float* red, int wave, int grp, int c, Chunk<VEC> (&acc)[K], int r) {
        if constexpr (GROUPS_IN_LDS) {
            stage_row<K, D, VEC>(region(red, wave) + grp * ROWF + r * GE::ROW, acc, c);
        } else {
#pragma unroll
            for (int kk = 0; kk < K; ++kk) across_groups_sum_chunk<G>(acc[kk]);
            if (grp == 0) stage_row<K, D, VEC>(region(red, wave) + r * GE::ROW, acc, c);
        }
    }

    // head wave, after the barrier: sum of the unit's n waves (segments), groups 0 .. SG-1 inside each
    static __device__ __forceinline__ void sum(const float* red, int wave, int n, int lane, float4 (&out)[NQ]) {
        const float4* red4 = reinterpret_cast<const float4*>(red);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int x = q * DL_WAVE + lane;
            out[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (x < F4) {
                const float4* base = red4 + (size_t)wave * SG * F4 + x;
                out[q] = base[0];
#pragma unroll
                for (int g = 1; g < SG; ++g) {
                    const float4 v = base[g * F4];
                    out[q].x += v.x; out[q].y += v.y; out[q].z += v.z; out[q].w += v.w;
                }
                for (int u = 1; u < n; ++u) {
#pragma unroll
                    for (int g = 0; g < SG; ++g) {
                        const float4 v = base[(u * SG + g) * F4];
                        out[q].x += v.x; out[q].y += v.y; out[q].z += v.z; out[q].w += v.w;
                    }
                }
            }
        }
    }
};

// ---------------------------------------------------------------------------- pair scorer
// Stage the u rows of Z and H (as fp32) in this wave's LDS region.
template <int K, int D, typename T>
__device__ __forceinline__ void stage_u_rows(float* urow, const T* __restrict__ Z, const T* __restrict__ H, size_t u) {
    using GE = Geo<K, D, T>;
    constexpr int VEC = GE::VEC;
    for (int x = lane_id(); x < GE::ROW / VEC; x += DL_WAVE) {
        store_f32<VEC>(urow + x * VEC, Tab<T>::load(Z + u * GE::ROW + x * VEC));
        store_f32<VEC>(urow + GE::ROW + x * VEC, Tab<T>::load(H + u * GE::ROW + x * VEC));
    }
}

// The arithmetic behind a row of several units once its slots are added (t): r + cx * x + cp * t, times a device scalar —
// written out with fmaf so that row_combine_kernel and the kernels that sum such rows inside their own launch
// (publish_unit_and_sum_row) produce the same bits.
__device__ __forceinline__ float4 combine_finish(float4 r, bool has_x, float cx, const float4& xv, float cp, const float4& t,
                                                 const float* __restrict__ scale) {
    if (has_x) { r.x = fmaf(cx, xv.x, r.x); r.y = fmaf(cx, xv.y, r.y); r.z = fmaf(cx, xv.z, r.z); r.w = fmaf(cx, xv.w, r.w); }
    r.x = fmaf(cp, t.x, r.x); r.y = fmaf(cp, t.y, r.y); r.z = fmaf(cp, t.z, r.z); r.w = fmaf(cp, t.w, r.w);
    if (scale) {
        const float g = scale[0];
        r.x *= g; r.y *= g; r.z *= g; r.w *= g;
    }
    return r;
}

// Per multi-segment row:  out[grow] = (accumulate ? out[grow] : 0) + cx * X[grow] + cp * sum_slots part[slot].
// One 256-thread block per row: each of the 4 waves sums every 4th slot, LDS combines them in wave
// order.  Partials are fp32 rows `pstride` floats apart; X and out are tables of type TX / TO.
template <int TOT, typename TX, typename TO>
__global__ __launch_bounds__(BLOCK) void row_combine_kernel(dl_csr_plan g, const float* __restrict__ part0, int pstride,
                                                            const TX* __restrict__ X, float cx, float cp,
                                                            TO* __restrict__ out0, int accumulate,
                                                            const float* __restrict__ part1 = nullptr,
                                                            TO* __restrict__ out1 = nullptr,
                                                            const TO* acc_in = nullptr,
                                                            const float* __restrict__ scale = nullptr) {
    // acc_in: the accumulated input read from its own array (may be `out0` itself); scale: a device scalar on the result
    // gridDim.y == 2: two independent (partials, output) pairs over the same plan in one launch
    const float* __restrict__ part = blockIdx.y ? part1 : part0;
    TO* __restrict__ out = blockIdx.y ? out1 : out0;
    constexpr int TOT4 = TOT / 4;
    constexpr int NQ = (TOT4 + DL_WAVE - 1) / DL_WAVE;
    __shared__ float4 red[WAVES_PER_BLOCK][NQ * DL_WAVE];
    const int m = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    float4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int s0 = g.multi_slot0[m], s1 = g.multi_slot0[m + 1];
#pragma unroll 2
    for (int slot = s0 + wave; slot < s1; slot += WAVES_PER_BLOCK) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int x = q * DL_WAVE + lane;
            if (x < TOT4) {
                const float4 v = load4<float>(part + (size_t)slot * pstride + 4 * x);
                acc[q].x += v.x; acc[q].y += v.y; acc[q].z += v.z; acc[q].w += v.w;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) red[wave][q * DL_WAVE + lane] = acc[q];
    __syncthreads();
    if (wave != 0) return;
    const size_t grow = (size_t)g.multi_row[m] + g.row_offset;
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
            const size_t o = grow * TOT + 4 * x;
            float4 r = acc_in ? load4<TO>(acc_in + o) : accumulate ? load4<TO>(out + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cx != 0.0f) xv = load4<TX>(X + o);
            store4(out + o, combine_finish(r, cx != 0.0f, cx, xv, cp, t, scale));
        }
    }
}

// ---------------------------------------------------------------------------- rows of several units, summed inside the launch
// The head wave of a unit whose row has several units (slot >= 0) calls this with its unit's result in o[]: lane l holds
// the float4s x = q * 64 + l (x < TOT4) of the [TOT]-float partial.  The partial goes to its slot (sc1 stores), the row's
// counter (plan.unit_count, all zero between launches) is incremented, and the wave whose increment completes the row —
// it may run on any XCD, at any time — adds the row's slots in EXACTLY the order row_combine_kernel uses (wave w of that
// kernel takes the slots s0 + w, s0 + w + 4, ...; then ((w0 + w1) + w2) + w3), so both forms give the same bits; it puts
// the counter back to zero, hands every float4 column of the total to `finish(x, total)` (x = float4 index in the row; the
// caller's epilogue: it writes the row) and returns true.  Every other caller returns false.  (Protocol and its
// limits: dl_common.h, "hand-off between units".)  pstride = floats between consecutive slots.
template <int TOT4, typename Finish>
__device__ __forceinline__ bool publish_unit_and_sum_row(const dl_csr_plan& g, int slot, float* __restrict__ part, int pstride,
                                                         const float4 (&o)[(TOT4 + DL_WAVE - 1) / DL_WAVE], int lane, Finish&& finish) {
    constexpr int NQ = (TOT4 + DL_WAVE - 1) / DL_WAVE;
    float* mine = part + (size_t)slot * pstride;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if (q * DL_WAVE + lane < TOT4) store4_sc1(mine + 4 * (q * DL_WAVE + lane), o[q]);
    wait_vmem();                                                   // every byte of this wave's partial has left
    const int m = g.slot_multi[slot];
    const int s0 = g.multi_slot0[m], s1 = g.multi_slot0[m + 1];
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(g.unit_count + m, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != s1 - s0 - 1) return false;
    if (lane == 0) __hip_atomic_store(g.unit_count + m, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // nobody else touches it any more
    // One float4 column of the row at a time, four slots in flight per round (one per accumulator), in a loop that is NOT
    // unrolled and indexes no register array: 16 + 16 registers live, so this tail does not set the kernel's register count
    // (summing all columns at once took the aggregation kernel from 60 to 88 registers — 8 to 5 waves per SIMD — and cost it
    // 18 % where HBM binds: profiles/r7_hbm_bound_kernel_stats.csv against r6).
#pragma unroll 1
    for (int x = lane; x < TOT4; x += DL_WAVE) {
        float4 acc[WAVES_PER_BLOCK];
#pragma unroll
        for (int w = 0; w < WAVES_PER_BLOCK; ++w) acc[w] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sl = s0; sl < s1; sl += WAVES_PER_BLOCK) {
            dl_vf4 v[WAVES_PER_BLOCK];
#pragma unroll
            for (int w = 0; w < WAVES_PER_BLOCK; ++w) {
                const int sw = sl + w < s1 ? sl + w : s1 - 1;           // clamped: a valid slot, left out below
                v[w] = load4_sc1_issue(part + (size_t)sw * pstride + 4 * x);
            }
            wait_loads_sc1(v[0], v[1], v[2], v[3]);
#pragma unroll
            for (int w = 0; w < WAVES_PER_BLOCK; ++w) {
                if (sl + w < s1) {
                    acc[w].x += v[w].x; acc[w].y += v[w].y; acc[w].z += v[w].z; acc[w].w += v[w].w;
                }
            }
        }
        float4 t = acc[0];
#pragma unroll
        for (int w = 1; w < WAVES_PER_BLOCK; ++w) { t.x += acc[w].x; t.y += acc[w].y; t.z += acc[w].z; t.w += acc[w].w; }
        finish(x, t);
    }
    return true;
}
static inline bool sums_rows_in_launch(const dl_csr_plan* g) {
    const int mode = config().inkernel_combine;
    return g->n_multi > 0 && g->slot_multi != nullptr && g->unit_count != nullptr && mode > 0 &&
           (mode > 1 || (long long)g->n_slots <= 5LL * g->n_multi);
}

static inline int pow2_at_least(int k) {
    int p = 1;
    while (p < k) p <<= 1;
    return p;
}

// vec_combine_kernel / the row-sum pass live in dl_route.hip
void launch_vec_combine(const dl_csr_plan* g, int K, const float* part, int mode, const float* s_raw, float* out, hipStream_t st);

// The H rows of the aggregation go out as streaming stores when the table is far beyond the caches (256 MiB Infinity
// Cache); on cache-resident graphs the scorer wants them in cache.  (The dZ / dH rows of the training kernels were
// tried too: a snap-patents-sized epoch 311 -> 319 ms with all of them streaming — only the forward H store pays.)
static inline int stream_rows(const dl_csr_plan* g, int row_elems, size_t elem) {
    const int force = config().stream_rows;                     // DL_STREAM_ROWS=0 / 1: measurements only
    if (force >= 0) return force;
    return (size_t)g->n_total * row_elems * elem > ((size_t)256 << 20) ? 1 : 0;
}

}  // namespace fast

// (K, D) pairs with a tuned instantiation.  D must be 4 * a power of two.
#define DL_FAST_SHAPES_F32(X) \
    X(4, 32) X(8, 64) X(16, 128) X(5, 32) X(5, 64) X(10, 32) X(10, 64) X(20, 32) X(8, 32) X(4, 64) X(4, 8) X(8, 8) X(3, 8)
#define DL_FAST_SHAPES_BF16(X) X(4, 32) X(8, 64) X(16, 128) X(5, 64) X(8, 32)

// CALL(OPS) is expanded with OPS = fast::Ops<K, D, T> of the matching instantiation
#define DL_DISPATCH(CALL)                                                                         \
    if (dtype == DL_F32) {                                                                        \
        DL_FAST_SHAPES_F32(CALL##_F32)                                                            \
    } else if (dtype == DL_BF16) {                                                                \
        DL_FAST_SHAPES_BF16(CALL##_BF16)                                                          \
    }                                                                                             \
    set_error("no tuned kernel for K=%d d=%d dtype=%d", K, d, dtype);                             \
    return DL_E_ARG;

}  // namespace dl
