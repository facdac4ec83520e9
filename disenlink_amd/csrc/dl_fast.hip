// Tuned kernels for gfx950, instantiated per (K, D, table type).
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

// ---------------------------------------------------------------------------- route
// p[e], a[e] for the entries of the plan's segments.  MIRROR: the plan covers col >= row only and
// every result is also written to the reverse entry (routing is symmetric, bitwise).
template <int K, int D, typename T, bool MIRROR, bool BALLOT = false>
__global__ __launch_bounds__(BLOCK, (K <= 8 && sizeof(T) == 4) ? 8 : (K <= 10 ? 6 : (K <= 16 ? 4 : 1))) void route_seg_kernel(dl_csr_plan g, const int32_t* __restrict__ rev,
                                                          const T* __restrict__ Z, float t,
                                                          uint8_t* __restrict__ p, float* __restrict__ a) {
    using GE = Geo<K, D, T>;
    using FL = typename GE::FL;
    constexpr int VEC = GE::VEC, G = GE::G, EPW = GE::EPW, KP = FL::KP, VPL = FL::VPL;
    const WaveSeg ws = load_wave_seg(g);
    if (!ws.active) return;
    const SegInfo si = ws.si;
    const int lane = lane_id();
    const int c = lane % G, grp = lane / G;
    const int kb = FL::factor_base(c);

    Chunk<VEC> zi[K];
#pragma unroll
    for (int k = 0; k < K; ++k) zi[k] = Tab<T>::load(Z + (size_t)si.grow * GE::ROW + k * D + c * VEC);

    int my_col = si.grow, my_rev = 0;
    if (si.beg + lane < si.end) {
        my_col = g.col[si.beg + lane];
        if (MIRROR) my_rev = rev[si.beg + lane];
    }
    for (int base = si.beg; base < si.end; base += EPW) {
        const int e = base + grp;
        const bool live = e < si.end;
        const int j = __shfl(my_col, e - si.beg, DL_WAVE);
        float part[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k)
            part[k] = k < K ? dot(zi[k < K ? k : 0], Tab<T>::load(Z + (size_t)j * GE::ROW + (k < K ? k : 0) * D + c * VEC))
                            : 0.0f;
        float ex[VPL];
        const float S = lane_exps<K, G>(part, c, t, ex);
        float best = 0.0f;
        int win = 255;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const float al = ex[i] / S;
            if (kb + i < K && (win == 255 || beats(al, best))) { best = al; win = kb + i; }
        }
        if constexpr (BALLOT && VPL == 1 && G <= 32) {
            // BALLOT arg-max (round 4): every lane holds at most one candidate, and lanes follow the factor order — so the
            // first maximal factor is the lowest set bit of the group's "equals the group maximum" mask: four DPP max
            // steps, one compare into a wave mask, a shift and a find-first-bit (~14 instructions) instead of four
            // (value, index) exchange-and-compare steps (32).  torch.argmax order kept: a NaN beats everything, the first
            // one wins (v_max ignores NaN, so NaN candidates get a mask of their own).
            const bool valid = win != 255;
            const bool isn = valid && best != best;
            float m = valid && !isn ? best : -__builtin_inff();
            m = fmaxf(m, xor_lane<1>(m));
            if constexpr (G >= 4) m = fmaxf(m, xor_lane<2>(m));
            if constexpr (G >= 8) m = fmaxf(m, xor_lane<4>(m));
            if constexpr (G >= 16) m = fmaxf(m, xor_lane<8>(m));
            if constexpr (G >= 32) m = fmaxf(m, xor_lane<16>(m));
            const unsigned long long eqm = __builtin_amdgcn_ballot_w64(valid && best == m);
            const unsigned long long nam = __builtin_amdgcn_ballot_w64(isn);
            const int sh = lane & ~(G - 1);
            constexpr unsigned GM = G == 32 ? 0xffffffffu : ((1u << G) - 1u);
            const unsigned ge = (unsigned)(eqm >> sh) & GM, gn = (unsigned)(nam >> sh) & GM;
            const int src = __builtin_ctz(gn ? gn : (ge | (1u << (G - 1))));           // (ge is never empty; the guard bit keeps ctz defined)
            win = FL::factor_base(src);
            best = gn ? __builtin_nanf("") : m;
        } else {
            group_argmax_first<G>(best, win);
        }
        if (MIRROR) {
            const int r = __shfl(my_rev, e - si.beg, DL_WAVE);
            if (live && c == 0) { p[e] = (uint8_t)win; a[e] = best; }
            if (live && c == 1 % G && r != e) { p[r] = (uint8_t)win; a[r] = best; }
        } else {
            if (live && c == 0) { p[e] = (uint8_t)win; a[e] = best; }
        }
    }
}

// s[i][k] = sum_{e in row i, p[e]=k} a[e] (raw; model.py:70-71).  Four LANES per segment position (a wavefront per
// segment would leave most lanes idle and spend its time in K all-reduces; one thread per segment walks 32 entries in
// eight dependent round trips): lane `sub` of a position adds the entries sub, sub + 4, ... of the segment in order and
// keeps the K sums in registers; the four lanes are then added as (0 + 1) + (2 + 3), and the (<= 4) positions of a unit —
// one aligned group of 16 lanes, a DPP row — in segment order by the unit's first position.  All exchanges are DPP
// (quad_perm, row_shl): no LDS.  KP = K rounded up to 4 / 8 / 16 / 32.
constexpr int ROWSUM_SUB = 4;                                   // lanes per segment position
constexpr int ROWSUM_POS_PER_BLOCK = BLOCK / ROWSUM_SUB;

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_move_i(int v) {
    return __builtin_amdgcn_update_dpp(-1, v, CTRL, 0xF, 0xF, false);     // lanes shifted in from outside the row keep -1
}

//
// Rows of SEVERAL units (slot >= 0; more than DL_UNIT_SEGS segments) are not summed from their units at all (round 2
// wrote one partial per unit and launched a combine kernel for a handful of rows: 5 us of launch for microseconds of
// work): the workgroups behind the first n_reg_blocks take one such row per WAVE — lane l adds the entries l, l + 64,
// ... of the row in order (loads in batches of four), then the 64 lanes are added by the wave butterfly.  One launch;
// the order depends on the row alone.
template <int KP>
__global__ __launch_bounds__(BLOCK) void s_rowsum_thread_kernel(dl_csr_plan g, int K, const uint8_t* __restrict__ p,
                                                                const float* __restrict__ a, float* __restrict__ s,
                                                                int n_reg_blocks, const int32_t* __restrict__ exp_rev = nullptr,
                                                                float* __restrict__ exp_wn = nullptr) {
    if ((int)blockIdx.x >= n_reg_blocks) {
        const int m = ((int)blockIdx.x - n_reg_blocks) * WAVES_PER_BLOCK + (int)(threadIdx.x >> 6);
        if (m >= g.n_multi) return;
        const int lane = lane_id();
        const int row = g.multi_row[m];
        const int beg = g.rowptr[row], end = g.rowptr[row + 1];
        float acc[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[k] = 0.0f;
        for (int e = beg + lane; e < end; e += 4 * DL_WAVE) {
            int k[4];
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = min(e + j * DL_WAVE, end - 1);
                k[j] = e + j * DL_WAVE < end ? (int)p[i] : 255;
                v[j] = a[i];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kk = 0; kk < KP; ++kk) acc[kk] += k[j] == kk ? v[j] : 0.0f;
        }
        float mine = 0.0f;
#pragma unroll
        for (int kk = 0; kk < KP; ++kk) {
            const float tot = wave_allreduce_sum(acc[kk]);
            if (lane == kk) mine = tot;
        }
        if (lane < K) s[((size_t)row + g.row_offset) * K + lane] = mine;
        return;
    }
    const int seg = blockIdx.x * ROWSUM_POS_PER_BLOCK + (int)(threadIdx.x / ROWSUM_SUB);
    const int sub = threadIdx.x % ROWSUM_SUB;
    const int row = seg < g.n_seg ? g.seg_row[seg] : -1;
    int beg = 0, end = 0, slot = -1;
    if (row >= 0) { beg = g.seg_beg[seg]; end = g.seg_end[seg]; slot = g.seg_slot[seg]; }
    if (slot >= 0) beg = end = 0;                                 // a row of several units: summed by its own wave (above)
    float acc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) acc[k] = 0.0f;
    for (int e = beg + sub; e < end; e += 4 * ROWSUM_SUB) {     // four entries in flight per lane (clamped loads), added in order
        int k[4];
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = min(e + j * ROWSUM_SUB, end - 1);
            k[j] = e + j * ROWSUM_SUB < end ? (int)p[i] : 255;
            v[j] = a[i];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) acc[kk] += k[j] == kk ? v[j] : 0.0f;
    }
    // segment sum over the four lanes of the position: (0 + 1) + (2 + 3), every lane of the quad gets it
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) acc[kk] = add_xor<2>(add_xor<1>(acc[kk]));
    // unit sum: the 4 positions of a workgroup-sized group are the 4 quads of one DPP row; a unit is a run of equal rows
    // among them, added in segment order by its first position (row_shl:4q brings quad +q)
    const int pos_in_grp = (threadIdx.x / ROWSUM_SUB) % WAVES_PER_BLOCK;
    // (a unit = same row AND same slot: plans with one segment per unit give every segment of a row its own slot)
    const int prev_row = dpp_move_i<0x114>(row);                 // row_shr:4: the previous position's row (-1 at the group's start)
    const int prev_slot = dpp_move_i<0x114>(slot);
    const bool head = row >= 0 && (pos_in_grp == 0 || prev_row != row || prev_slot != slot);
    const int r1 = dpp_move_i<0x104>(row), r2 = dpp_move_i<0x108>(row), r3 = dpp_move_i<0x10C>(row);   // row_shl:4 / 8 / 12
    const int t1 = dpp_move_i<0x104>(slot), t2 = dpp_move_i<0x108>(slot), t3 = dpp_move_i<0x10C>(slot);
    const bool ok1 = r1 == row && t1 == slot, ok2 = ok1 && r2 == row && t2 == slot, ok3 = ok2 && r3 == row && t3 == slot;
    float tot[KP];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
        const float v1 = dpp_move<0x104>(acc[kk]), v2 = dpp_move<0x108>(acc[kk]), v3 = dpp_move<0x10C>(acc[kk]);
        tot[kk] = acc[kk];
        tot[kk] += ok1 ? v1 : 0.0f;
        tot[kk] += ok2 ? v2 : 0.0f;
        tot[kk] += ok3 ? v3 : 0.0f;
    }
#ifdef DL_EXP_ROWSUM_SCATTER      // TIMING EXPERIMENT ONLY: what it would cost the row-sum pass to write the normalised weight of every
    // entry through the reverse-edge map (one more coalesced read + one scattered 4-byte store per entry; the totals used here
    // are the head position's only, so the values are not the real ones)
    if (exp_rev != nullptr) {
        for (int e = beg + sub; e < end; e += ROWSUM_SUB) {
            float tt = 1.0f;
            const int kq = (int)p[e];
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) tt = kq == kk ? tot[kk] : tt;
            exp_wn[exp_rev[e]] = a[e] / one_if_zero(tt);
        }
    }
#endif
    if (!head || sub != 0 || slot >= 0) return;
    float* dst = s + ((size_t)row + g.row_offset) * K;
#pragma unroll
    for (int kk = 0; kk < KP; ++kk)
        if (kk < K) dst[kk] = tot[kk];
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
// Bits of m below this lane.
__device__ __forceinline__ int bits_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Aggregation with CLASS-OWNED accumulators (round 3).  The round-2 kernel let every lane group take every G-th entry,
// so each group needed an accumulator for every factor: K * VEC registers per lane, K * VEC FMAs + K selects per gathered
// entry to add it to the one accumulator of its factor (40 of the ~55 vector instructions per entry at K = 8), a
// cross-group butterfly at the end — and at 8 waves per SIMD the K = 8 instantiation spilled 4 registers per lane
// (1 KB of scratch written and read back per WAVE: the 0.8 GB of unexplained WRITE_SIZE per launch at snap-patents
// size, profiles/r2q).  Here the entries of a segment are sorted by CLASS = factor % NC first (NC = min(K, groups per
// wave); ballots + a bit count give every entry its slot, the sorted (slice index, accumulator, weight) triples live in
// LDS) and group g walks the entries of class g: a lane accumulates only the factors g, g + NC, ... — ACC = K / NC
// accumulators (2 at K = 8, d = 64) — and every (factor, chunk) of the result row is owned by exactly one lane of the
// wave: no cross-group sum, the lane stores its chunks straight into the wave's staged row.  Inside a (segment, factor)
// the entries are added in ascending entry order by ONE lane group, whatever the lane geometry: the summation order
// depends on the row alone.  Cost: the walk takes max_g |class g| steps instead of |segment| / groups (a segment routed
// entirely to one factor is walked by one group).  U = gathers in flight per group (4: 2 / 8 measured no better).
template <int K, int D, typename T, int U>
__global__ __launch_bounds__(BLOCK) void aggregate_cls_kernel(dl_csr_plan g, const T* __restrict__ Z, float beta,
                                                              const uint8_t* __restrict__ p,
                                                              const float* __restrict__ a,
                                                              const float* __restrict__ s, T* __restrict__ H,
                                                              float* __restrict__ h_part, int stream_out) {
    using GE = Geo<K, D, T>;
    constexpr int VEC = GE::VEC, G = GE::G, NG = GE::EPW, ROW = GE::ROW;
    constexpr int NC = K < NG ? K : NG;                           // classes = lane groups at work
    constexpr int ACC = (K + NC - 1) / NC;                        // factors per class
    using US = Stage<K, D, T, 1>;
    __shared__ __attribute__((aligned(16))) float red[US::FLOATS];
    __shared__ int ent_col[WAVES_PER_BLOCK][DL_WAVE];
    __shared__ int ent_k[WAVES_PER_BLOCK][DL_WAVE];
    __shared__ float ent_w[WAVES_PER_BLOCK][DL_WAVE];
    const WaveSeg ws = load_wave_seg(g);
    const SegInfo si = ws.si;
    const int lane = lane_id();
    const int c = lane % G, grp = lane / G;
    // the row's own z is needed last, by the head wave of a single-unit row only: fetched first
    const bool direct = ws.head && si.slot < 0;
    float4 zrow[US::NQ];
    if (direct) {
#pragma unroll
        for (int q = 0; q < US::NQ; ++q) {
            const int x = q * DL_WAVE + lane;
            if (x < US::F4) zrow[q] = load4<T>(Z + (size_t)si.grow * ROW + 4 * x);
        }
    }
    if (ws.active) {
        const int cnt = si.end - si.beg;
        const bool mine = lane < cnt;
        int my_col = si.grow, my_k = 0;
        float my_a = 0.0f, my_s = 1.0f;
        if (mine) {
            my_col = g.col[si.beg + lane];
            my_k = p[si.beg + lane];
            my_a = a[si.beg + lane];
        }
        // the neighbour's normaliser stays in flight while the entries are sorted and the first row gathers go out
#ifndef DL_EXP_NO_SGATHER         // -DDL_EXP_NO_SGATHER: TIMING EXPERIMENT ONLY (wrong results): what the aggregation would gain
        if (mine) my_s = s[(size_t)my_col * K + my_k];      // if the normalised weight a / s~ arrived with p / a in the per-entry stream
#endif
        const int cls = my_k % NC;
        int pos = 0, my_off = 0, my_cnt = 0, run = 0, trip = 0;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const unsigned long long m = __ballot(mine && cls == cc);
            const int n = __popcll(m);
            if (cls == cc) pos = run + bits_below(m);
            if (grp == cc) { my_off = run; my_cnt = n; }
            run += n;
            trip = n > trip ? n : trip;
        }
        // sorted per-entry scalars in LDS: the index of the gathered slice (col * K + factor), the accumulator it goes to
        // and its weight.  The loop below is branch-free on purpose: with `if (live)` around the accumulation hipcc built a
        // chain of exec-mask branches with s_waitcnt vmcnt(0) inside — the four gathers of a batch ran one at a time and a
        // slot cost ~37 vector instructions (tools/kernel_isa.py); dead slots now gather a valid slice with weight 0.
        int* wsl = ent_col[ws.wave];
        int* wli = ent_k[ws.wave];
        float* ww = ent_w[ws.wave];
        if (mine) { wsl[pos] = my_col * K + my_k; wli[pos] = my_k / NC; }
        __builtin_amdgcn_wave_barrier();
        Chunk<VEC> acc[ACC];
#pragma unroll
        for (int i = 0; i < ACC; ++i) acc[i] = zero_chunk<VEC>();
        const T* zc = Z + c * VEC;
        for (int it = 0; it < trip; it += U) {
            Chunk<VEC> v[U];
            int sx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                sx[u] = it + u < my_cnt ? my_off + it + u : -1;
                v[u] = Tab<T>::load(zc + (size_t)(unsigned)wsl[sx[u] < 0 ? 0 : sx[u]] * D);   // trip > 0: slot 0 holds a real entry
            }
            if (it == 0) {                                        // weights: behind the first batch of gathers
                if (mine) ww[pos] = my_a / one_if_zero(my_s);
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sl = sx[u] < 0 ? 0 : sx[u];
                const float w = sx[u] < 0 ? 0.0f : ww[sl];
                if constexpr (ACC == 1) {
                    fma_chunk(acc[0], w, v[u]);
                } else {
                    const int li = wli[sl];
#pragma unroll
                    for (int i = 0; i < ACC; ++i) fma_chunk(acc[i], li == i ? w : 0.0f, v[u]);
                }
            }
        }
        float* row = US::region(red, ws.wave);
#pragma unroll
        for (int i = 0; i < ACC; ++i) {
            const int k = i * NC + grp;
            if (grp < NC && k < K) store_f32<VEC>(row + k * D + c * VEC, acc[i]);
        }
    }
    __syncthreads();
    if (!ws.head) return;
    float4 r[US::NQ];
    US::sum(red, ws.wave, ws.n_unit, lane, r);
    const float omb = 1.0f - beta;
#pragma unroll
    for (int q = 0; q < US::NQ; ++q) {
        const int x = q * DL_WAVE + lane;
        if (x < US::F4) {
            if (direct) {
                const float4 z = zrow[q];
                const float4 h = make_float4(beta * z.x + omb * r[q].x, beta * z.y + omb * r[q].y, beta * z.z + omb * r[q].z,
                                             beta * z.w + omb * r[q].w);
                if (stream_out) store4_stream(H + (size_t)si.grow * ROW + 4 * x, h);
                else store4(H + (size_t)si.grow * ROW + 4 * x, h);
            } else {
                store4(h_part + (size_t)si.slot * ROW + 4 * x, r[q]);
            }
        }
    }
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
            if (cx != 0.0f) {
                const float4 xv = load4<TX>(X + o);
                r.x += cx * xv.x; r.y += cx * xv.y; r.z += cx * xv.z; r.w += cx * xv.w;
            }
            r.x += cp * t.x; r.y += cp * t.y; r.z += cp * t.z; r.w += cp * t.w;
            if (scale) {
                const float g = scale[0];
                r.x *= g; r.y *= g; r.z *= g; r.w *= g;
            }
            store4(out + o, r);
        }
    }
}

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
            const int j = __shfl(my_col, e - si.beg, DL_WAVE);
            const int k = __shfl(my_k, e - si.beg, DL_WAVE);
            const float ae = __shfl(my_a, e - si.beg, DL_WAVE);
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
    const float* dz_in, const float* __restrict__ scale, float* dZ, float* __restrict__ dz_part) {
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
            const int idx = base + grp - si.beg;
            const int j = __shfl(my_col, idx, DL_WAVE);
            const int k = __shfl(my_k, idx, DL_WAVE);
            const float cc = __shfl(my_cc, idx, DL_WAVE);          // 0 past the segment end
            const float w2 = __shfl(my_w2, idx, DL_WAVE);
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
#ifdef DL_EXP_SKIP_GATHER
    Chunk<VEC> exp_h[KBP], exp_z[KBP];
#endif
    for (int base = si.beg; base < si.end; base += EPW) {
        const int it = base + grp;
        const bool live = it < si.end;
        const size_t v = (size_t)__shfl(my_col, it - si.beg, DL_WAVE);
        const int q = __shfl(my_pair, it - si.beg, DL_WAVE);
        // factors are processed in blocks of KB <= 8: at most 2*KB row chunks live at a time, whatever K is
        float term = 0.0f;
#pragma unroll
        for (int b0 = 0; b0 < K; b0 += KB) {
            float pq[KBP], ps[KBP];
#ifdef DL_EXP_SKIP_GATHER     // TIMING EXPERIMENT ONLY (wrong results): 2 of every 5 iterations re-use the rows gathered before — would
            // the kernel speed up in proportion if fewer rows had to be gathered (rows shared between the pairs of a u block)?
            Chunk<VEC> hx[KBP], zx[KBP];
            if (K > 8 || ((base - si.beg) / EPW) % 5 < 3 || base == si.beg) {
#pragma unroll
                for (int k = 0; k < KBP; ++k) {
                    const int kk = (k < KB && b0 + k < K) ? b0 + k : 0;
                    exp_h[k] = Tab<T>::load(H + v * ROW + kk * D + c * VEC);
                    exp_z[k] = Tab<T>::load(Z + v * ROW + kk * D + c * VEC);
                }
            }
#pragma unroll
            for (int k = 0; k < KBP; ++k) { hx[k] = exp_h[k]; zx[k] = exp_z[k]; }
#pragma unroll
            for (int k = 0; k < KBP; ++k) {
                const bool in = k < KB && b0 + k < K;
                const int kk = in ? b0 + k : 0;
                pq[k] = in ? dot(load_f32<VEC>(&urow[wave][ROW + kk * D + c * VEC]), hx[k]) : 0.0f;
                ps[k] = in ? dot(load_f32<VEC>(&urow[wave][kk * D + c * VEC]), zx[k]) : 0.0f;
            }
#else
#pragma unroll
            for (int k = 0; k < KBP; ++k) {
                const bool in = k < KB && b0 + k < K;
                const int kk = in ? b0 + k : 0;
                pq[k] = in ? dot(load_f32<VEC>(&urow[wave][ROW + kk * D + c * VEC]), Tab<T>::load(H + v * ROW + kk * D + c * VEC)) : 0.0f;
                ps[k] = in ? dot(load_f32<VEC>(&urow[wave][kk * D + c * VEC]), Tab<T>::load(Z + v * ROW + kk * D + c * VEC)) : 0.0f;
            }
#endif
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

// Scorer backward, recomputing e_k and q_k (used when the forward did not store them): one wave per
// segment of node u's pair slots.  Partials (multi-segment rows) hold [dZ row | dH row] per slot.
// FUSED (training step, dl_score_pairs_train): the same walk IS the forward — the wave has S_k and Q_k of every
// entry, so it forms prob itself, applies the weighted-BCE gradient of main_disentangled.py:195 inline
// (g = w (p - y) / max(p (1 - p), 1e-12) as in dl_pair_bce, folded with the sigmoid backward) and writes prob[q] (both directions of a pair
// compute the same bits and both write them).  prob_in / g_prob are then unused, y / w / prob_out are used instead:
// one pass that gathers the partner rows once per direction, instead of a forward pass plus two backward passes.
// The one-pass training kernel at K = 8, d = 64 runs THREE waves per SIMD: 165 registers without a spill once the groups' partial
// rows are added in registers (one staged row per wave: 16 KB of LDS per workgroup instead of 64, so LDS does not cap the
// occupancy at two either) — squirrel 478 -> 457 us, chameleon 99 -> 96; four waves (128 registers) spill 46 and take 3x;
// d = 32 / 8 would spill a few registers at three waves and stay at two.
#ifndef DL_TRAIN_WAVES
#define DL_TRAIN_WAVES 3              // -DDL_TRAIN_WAVES=1 (DL_CXXFLAGS): the two-wave form with the groups' rows in LDS, for A/B runs
#endif
template <int K_, int D_, bool FUSED_>
struct TrainWaves { static constexpr int value = (FUSED_ && K_ == 8 && D_ == 64) ? DL_TRAIN_WAVES : 1; };
template <int K, int D, typename T, bool FUSED>
__global__ __launch_bounds__(BLOCK, (TrainWaves<K, D, FUSED>::value)) void score_bwd_seg_kernel(dl_csr_plan g, const int32_t* __restrict__ inc_pair,
                                                              const T* __restrict__ Z, const T* __restrict__ H,
                                                              float t, const float* __restrict__ prob,
                                                              const float* __restrict__ g_prob,
                                                              float* __restrict__ dZ, float* __restrict__ dH,
                                                              float* __restrict__ part,
                                                              const float* __restrict__ y = nullptr,
                                                              const float* __restrict__ w = nullptr,
                                                              float* __restrict__ prob_out = nullptr) {
    using GE = Geo<K, D, T>;
    using FL = typename GE::FL;
    constexpr int VEC = GE::VEC, G = GE::G, EPW = GE::EPW, KP = FL::KP, VPL = FL::VPL, ROW = GE::ROW;
    using US = Stage<K, D, T, 2, (TrainWaves<K, D, FUSED>::value > 1)>;
    // per wave: the u rows of Z and H during the walk (the first 2 ROW floats of the wave's region), then — the same
    // memory — the groups' [dZ row | dH row] partials for the unit sum
    __shared__ __attribute__((aligned(16))) float red[US::FLOATS];
    const WaveSeg ws = load_wave_seg(g);
    const SegInfo si = ws.si;
    const int wave = ws.wave, lane = lane_id();
    const bool active = ws.active;
    float* const urow_w = US::region(red, wave);
    if (active) stage_u_rows<K, D, T>(urow_w, Z, H, (size_t)si.grow);
    __syncthreads();
    const int c = lane % G, grp = lane / G;
    if (active) {
        Chunk<VEC> accZ[K], accH[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { accZ[k] = zero_chunk<VEC>(); accH[k] = zero_chunk<VEC>(); }
        int my_col = si.grow, my_q = 0;
        float my_gl = 0.0f, my_y = 0.0f, my_w = 0.0f;
        if (si.beg + lane < si.end) {
            my_col = g.col[si.beg + lane];
            const int q = inc_pair[si.beg + lane];
            if constexpr (FUSED) {
                my_q = q;
                my_y = y[q];
                my_w = w[q];
            } else {
                const float pr = prob[q];
                my_gl = g_prob[q] * pr * (1.0f - pr);       // sigmoid backward p(1-p)
            }
        }
        for (int base = si.beg; base < si.end; base += EPW) {
            const int idx = base + grp - si.beg;
            const size_t v = (size_t)__shfl(my_col, idx, DL_WAVE);
            float gl = __shfl(my_gl, idx, DL_WAVE);          // 0 past the segment end
            Chunk<VEC> zv[K], hv[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                zv[k] = Tab<T>::load(Z + v * ROW + k * D + c * VEC);
                hv[k] = Tab<T>::load(H + v * ROW + k * D + c * VEC);
            }
            float pq[KP], ps[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int kk = k < K ? k : 0;
                pq[k] = k < K ? dot(load_f32<VEC>(&urow_w[ROW + kk * D + c * VEC]), hv[kk]) : 0.0f;
                ps[k] = k < K ? dot(load_f32<VEC>(&urow_w[kk * D + c * VEC]), zv[kk]) : 0.0f;
            }
            TransposedReduce<KP, G / 2>::run(pq, c);
            TransposedReduce<KP, G / 2>::run(ps, c);
            float ch_lane[VPL], cz_lane[VPL], ek_lane[VPL];
#pragma unroll
            for (int i = 0; i < VPL; ++i) ek_lane[i] = expf(div_t(ps[i], t));
            if constexpr (FUSED) {
                float term = 0.0f;
#pragma unroll
                for (int i = 0; i < VPL; ++i)
                    if (FL::primary(c) && FL::factor_base(c) + i < K) term += pq[i] * ek_lane[i];
                const float p = sigmoid_ref(group_allreduce_sum<G>(term));
                const float yy = __shfl(my_y, idx, DL_WAVE), ww = __shfl(my_w, idx, DL_WAVE);   // w = 0 past the segment end
                const int qq = __shfl(my_q, idx, DL_WAVE);
                // dl_pair_bce's gradient times the sigmoid backward: w (p - y) / max(r, 1e-12) * r with r = p (1 - p) — i.e.
                // w (p - y) itself unless r underflows the clamp (saturated scores: r = 0 gives exactly 0), without the division
                const float r = p * (1.0f - p);
                gl = ww == 0.0f ? 0.0f : ww * (p - yy) * (r >= 1e-12f ? 1.0f : r * 1e12f);
                if (base + grp < si.end && c == 0) prob_out[qq] = p;
            }
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const float ek = ek_lane[i];
                ch_lane[i] = gl == 0.0f ? 0.0f : gl * ek;
                cz_lane[i] = gl == 0.0f ? 0.0f : div_t(gl * pq[i] * ek, t);     // x / 1 == x: t == 1 skips the IEEE division
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float ch = group_bcast<G>(ch_lane[FL::src_slot(k)], FL::src_lane(k));
                const float cz = group_bcast<G>(cz_lane[FL::src_slot(k)], FL::src_lane(k));
                fma_chunk(accH[k], ch, hv[k]);
                fma_chunk(accZ[k], cz, zv[k]);
            }
        }
        // the wave is done with its u rows (its own LDS region, program order): the region now takes its results
        US::put(red, wave, grp, c, accZ, 0);
        US::put(red, wave, grp, c, accH, 1);
    }
    __syncthreads();
    if (!ws.head) return;
    float4 r[US::NQ];
    US::sum(red, wave, ws.n_unit, lane, r);
#pragma unroll
    for (int q = 0; q < US::NQ; ++q) {
        const int x = q * DL_WAVE + lane;
        if (x < US::F4) {
            if (si.slot < 0) {
                float* o = x < ROW / 4 ? dZ + (size_t)si.grow * ROW + 4 * x : dH + (size_t)si.grow * ROW + 4 * (x - ROW / 4);
                store4(o, r[q]);
            } else {
                store4(part + (size_t)si.slot * 2 * ROW + 4 * x, r[q]);       // [dZ row | dH row]
            }
        }
    }
}

// ---------------------------------------------------------------------------- one-pass training scorer, wave per entry
// Round 4.  The kernel above gives every entry to a group of 16 lanes (4 entries per wave step): each lane then carries
// the WHOLE K x 2 accumulator set (64 registers, replicated in all four groups) next to the 64 registers of gathered
// rows — 165 registers, three waves per SIMD.  Measured on squirrel: the vector pipe 51 % busy, the L1 at 48 % of its
// 64 B/clk, neither hidden behind the other.
//
// Here the 64 lanes of the wave share ONE entry: lane l holds float4 number j * 64 + l of a row (j < NJ = K*D/256), i.e.
// with D = 64 a DPP row of 16 lanes holds one factor slice.  Per lane: 2 NJ accumulators (16 registers at K = 8
// instead of 64, and no sum over lane groups at the end), the node's own rows re-read from the wave's LDS region, the
// per-entry scalars (label, weight, pair id) in LDS too, row addresses as scalar base + lane offset: 128 registers, FOUR
// waves per SIMD.  A step still gathers U = 4 entries (16 KB per wave in flight).  The 16 partial dot products of a
// step (4 entries x 2 chunks x {z.z, h.h}) are reduced over the 16 lanes of the DPP row by ONE transposed reduction
// (lane i ends with complete sum number i), so the step needs one expf (lanes 0..7) and one sigmoid (lanes 8..15),
// and the two coefficients of every (entry, chunk) are formed in the lane that holds them and handed to the row by
// row_newbcast moves: 226 vector instructions per step against 271 above.
// Entries are accumulated in ascending order by every lane: the sums depend on the row alone (shard-independent).
// Same-box A/B (profiles/r4g_train_ab.txt): real squirrel 442 -> 387 us, chameleon 96 -> 78.
// Tried and dropped: a second register set with the next step's rows requested before the current step is computed
// (two waves per SIMD; hipcc renamed the sets in the two-step unrolled loop, kept three of them live and spilled 28-50
// registers: 630-980 us); the per-value xor / rotation all-reduces instead of the transposed reduction (+35 instructions
// per step, the same time: at four waves the kernel is not bound by vector issue).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {             // old value undefined: no zeroing move in front of the DPP move
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int I>
__device__ __forceinline__ float row_bcast(float v) { return dpp_mov<0x150 + I>(v); }      // row_newbcast:I
// Sum over the 16 lanes of a DPP row, in every lane, by ROTATIONS (row_ror 8, 4, 2, 1): each step is ONE v_add_f32_dpp
// (the xor butterfly needs a move + an add for its xor-4 step: 5 instructions per value, 80 per step of this kernel).
// All 16 lanes end with the same bits: after the rotation by r every lane holds the sum of its coset of <r>, formed as
// (coset of the previous step) + (the other one) — the same two addends in every lane of the coset, addition commutes.
template <int CTRL>
__device__ __forceinline__ float add_dpp(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_allreduce_sum(float v) {
    v = add_dpp<0x128>(v);      // row_ror:8
    v = add_dpp<0x124>(v);      // row_ror:4
    v = add_dpp<0x122>(v);      // row_ror:2
    return add_dpp<0x121>(v);   // row_ror:1
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

#ifndef DL_TRAIN_WAVE_KERNEL
#define DL_TRAIN_WAVE_KERNEL 1        // -DDL_TRAIN_WAVE_KERNEL=0: the group-per-entry kernel above, for A/B runs
#endif
template <int K, int D>
struct TrainWave {
    static constexpr bool ok = DL_TRAIN_WAVE_KERNEL && D == 64 && (K == 4 || K == 8);
    static constexpr int NJ = K * D / 256;                          // float4 per lane per row
    static constexpr int U = 4;                                     // entries per step
};

template <int K, int D, bool T1, bool UREG = false>
__global__ __launch_bounds__(BLOCK, (UREG ? 3 : 4)) void score_train_wave_kernel(
        dl_csr_plan g, const int32_t* __restrict__ inc_pair, const float* __restrict__ Z, const float* __restrict__ H, float t,
        float* __restrict__ dZ, float* __restrict__ dH, float* __restrict__ part, const float* __restrict__ y,
        const float* __restrict__ w, float* __restrict__ prob_out) {
    using TW = TrainWave<K, D>;
    constexpr int NJ = TW::NJ, U = TW::U, ROW = K * D, NV = U * NJ;
    static_assert(D == 64 && NJ >= 1 && NV <= 8, "one DPP row of 16 lanes per factor slice; at most 8 exponents per row and step");
    using US = Stage<K, D, float, 2, true>;                         // one [dZ row | dH row] per wave for the unit sum
    __shared__ __attribute__((aligned(16))) float red[US::FLOATS];
    __shared__ float ent_y[WAVES_PER_BLOCK][DL_WAVE], ent_w[WAVES_PER_BLOCK][DL_WAVE];     // per-entry scalars of the segment:
    __shared__ int ent_q[WAVES_PER_BLOCK][DL_WAVE];                                        // 3 registers fewer than lane copies
    const WaveSeg ws = load_wave_seg(g);
    const SegInfo si = ws.si;
    const int wave = ws.wave, lane = lane_id();
    const int i = lane & 15;                                       // position in the DPP row (the row holds the factors r, r + 4, ... of row r)
    float4 az[NJ], ah[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) az[j] = ah[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    // the node's own rows: in registers (UREG) or re-read from the wave's LDS region at every step (4 ds_read_b128 per
    // lane and step — 16 registers fewer, which is what lets a fourth wave onto the SIMD)
    float4* const mine = reinterpret_cast<float4*>(US::region(red, wave));
    if (ws.active) {
        float4 uz[UREG ? NJ : 1], uh[UREG ? NJ : 1];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float4 a4 = *reinterpret_cast<const float4*>(Z + (size_t)si.grow * ROW + (j * 64 + lane) * 4);
            const float4 b4 = *reinterpret_cast<const float4*>(H + (size_t)si.grow * ROW + (j * 64 + lane) * 4);
            if constexpr (UREG) {
                uz[j] = a4;
                uh[j] = b4;
            } else {
                mine[j * 64 + lane] = a4;                           // read back by this lane only: no barrier needed
                mine[ROW / 4 + j * 64 + lane] = b4;
            }
        }
        int my_col = si.grow;
        {
            int my_q = 0;
            float my_y = 0.0f, my_w = 0.0f;                         // w = 0 past the segment end: no gradient, no output
            if (si.beg + lane < si.end) {
                my_col = g.col[si.beg + lane];
                my_q = inc_pair[si.beg + lane];
                my_y = y[my_q];
                my_w = w[my_q];
            }
            ent_y[wave][lane] = my_y;                               // written and read by this wave only: no barrier
            ent_w[wave][lane] = my_w;
            ent_q[wave][lane] = my_q;
        }
        auto load_rows = [&](float4 (&zv)[U][NJ], float4 (&hv)[U][NJ], int step) {
#pragma unroll
            for (int e = 0; e < U; ++e) {
                // the entry is wave-uniform: its row address is a scalar, the lane offset a constant
                const size_t v = (size_t)(unsigned)__builtin_amdgcn_readlane(my_col, (step * U + e) & 63);
                // the row base stays a SCALAR (the empty asm keeps the compiler from folding the loop-invariant lane offset
                // into a hoisted 64-bit vector base per table): global_load ... v_lane_offset, s[base] — 3 registers fewer
                const float* zs = Z + v * ROW;
                const float* hs = H + v * ROW;
                asm volatile("" : "+s"(zs), "+s"(hs));
                const float* zr = zs + lane * 4;
                const float* hr = hs + lane * 4;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    zv[e][j] = *reinterpret_cast<const float4*>(zr + j * 256);
                    hv[e][j] = *reinterpret_cast<const float4*>(hr + j * 256);
                }
            }
        };
        auto consume = [&](const float4 (&zv)[U][NJ], const float4 (&hv)[U][NJ], int step) {
            // 16 partial dot products per lane: index = table * 8 + chunk * 4 + entry (z_u . z_v below 8, h_u . h_v above)
            float val[16];
#pragma unroll
            for (int x = 0; x < 16; ++x) val[x] = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float4 a4, b4;
                if constexpr (UREG) {
                    a4 = uz[j];
                    b4 = uh[j];
                } else {                                            // (one chunk of the node's rows live at a time)
                    a4 = mine[j * 64 + lane];
                    b4 = mine[ROW / 4 + j * 64 + lane];
                }
#pragma unroll
                for (int e = 0; e < U; ++e) {
                    val[j * 4 + e] = dot4_packed(a4, zv[e][j]);
                    val[8 + j * 4 + e] = dot4_packed(b4, hv[e][j]);
                }
                if constexpr (!UREG) __builtin_amdgcn_sched_barrier(0);   // keep the chunks apart: fewer temporaries live at once
            }
            // 16 values over the 16 lanes of the DPP row, halving the value count at every exchange: lane i ends with the
            // complete sum number i — lanes 0..7: z_u . z_v of (chunk i / 4, entry i % 4), lanes 8..15: h_u . h_v of the same
            TransposedReduce<16, 8>::run(val, i);
            const float mine_v = val[0];
            // ONE expf for the step (lanes 8..15 exponentiate a value nobody reads) ...
            const float ex = expf(T1 ? mine_v : mine_v / t);
            // ... its partner lane (i ^ 8) forms (h_u . h_v) exp(z_u . z_v / t); two chunks of an entry sit 4 lanes apart
            const float ttv = mine_v * xor_lane<8>(ex);                                      // valid in lanes 8..15
            float term = ttv;
            if constexpr (NJ == 2) term += xor_lane<4>(ttv);                                  // lanes 8..15: entry i % 4, both chunks
            const float logit = add_xor<32>(add_xor<16>(term));                              // ... over the 4 rows (all factors)
            const float p = sigmoid_ref(logit);
            const int idx = step * U + (i & 3);
            const float yy = ent_y[wave][idx & 63], ww = ent_w[wave][idx & 63];
            const int qq = ent_q[wave][idx & 63];
            // dl_pair_bce's gradient times the sigmoid backward: w (p - y) / max(q, 1e-12) * q with q = p (1 - p) — i.e.
            // w (p - y) itself unless q underflows the clamp (saturated scores: q = 0 gives exactly 0), without the division
            const float pr = p * (1.0f - p);
            const float gl = ww == 0.0f ? 0.0f : ww * (p - yy) * (pr >= 1e-12f ? 1.0f : pr * 1e12f);     // valid in lanes 8..15
            if (lane >= 8 && lane < 12 && si.beg + idx < si.end) prob_out[qq] = p;
            // the two coefficients of (entry, chunk), formed ONCE in the lane that holds its exponent / its product and
            // handed to the row afterwards: lanes 0..7: gl e^., lanes 8..15: gl (h.h) e^. / t.  0 * inf must stay 0 (an
            // overflowed exponent saturates p, so its gl is exactly 0): the factors are clamped to the largest finite value
            // first — finite values pass unchanged, a NaN gl still gives NaN
            const float exc = fminf(ex, 3.402823466e38f);
            const float ttc = __builtin_amdgcn_fmed3f(T1 ? ttv : ttv / t, -3.402823466e38f, 3.402823466e38f);
            const float gl_partner = xor_lane<8>(gl);              // OUTSIDE the select: a DPP move under a divergent branch reads 0 from the masked-off lanes
            const float coef = (i & 8) ? gl * ttc : gl_partner * exc;
            float Eb[NJ][U], Tb[NJ][U];
            Eb[0][0] = row_bcast<0>(coef); Eb[0][1] = row_bcast<1>(coef); Eb[0][2] = row_bcast<2>(coef); Eb[0][3] = row_bcast<3>(coef);
            Tb[0][0] = row_bcast<8>(coef); Tb[0][1] = row_bcast<9>(coef); Tb[0][2] = row_bcast<10>(coef); Tb[0][3] = row_bcast<11>(coef);
            if constexpr (NJ == 2) {
                Eb[1][0] = row_bcast<4>(coef); Eb[1][1] = row_bcast<5>(coef); Eb[1][2] = row_bcast<6>(coef); Eb[1][3] = row_bcast<7>(coef);
                Tb[1][0] = row_bcast<12>(coef); Tb[1][1] = row_bcast<13>(coef); Tb[1][2] = row_bcast<14>(coef); Tb[1][3] = row_bcast<15>(coef);
            }
#pragma unroll
            for (int e = 0; e < U; ++e) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float ch = Eb[j][e];
                    const float cz = Tb[j][e];
                    ah[j].x = fmaf(ch, hv[e][j].x, ah[j].x); ah[j].y = fmaf(ch, hv[e][j].y, ah[j].y);
                    ah[j].z = fmaf(ch, hv[e][j].z, ah[j].z); ah[j].w = fmaf(ch, hv[e][j].w, ah[j].w);
                    az[j].x = fmaf(cz, zv[e][j].x, az[j].x); az[j].y = fmaf(cz, zv[e][j].y, az[j].y);
                    az[j].z = fmaf(cz, zv[e][j].z, az[j].z); az[j].w = fmaf(cz, zv[e][j].w, az[j].w);
                }
            }
        };
        const int nsteps = (si.end - si.beg + U - 1) / U;           // entries past the end repeat a valid row with w = 0
        float4 zA[U][NJ], hA[U][NJ];
        for (int s = 0; s < nsteps; ++s) {
            load_rows(zA, hA, s);
            consume(zA, hA, s);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            mine[j * 64 + lane] = az[j];
            mine[ROW / 4 + j * 64 + lane] = ah[j];
        }
    }
    __syncthreads();
    if (!ws.head) return;
    float4 o[US::NQ];
    US::sum(red, wave, ws.n_unit, lane, o);
#pragma unroll
    for (int q = 0; q < US::NQ; ++q) {
        const int x = q * DL_WAVE + lane;
        if (x < US::F4) {
            if (si.slot < 0) {
                float* dst = x < ROW / 4 ? dZ + (size_t)si.grow * ROW + 4 * x : dH + (size_t)si.grow * ROW + 4 * (x - ROW / 4);
                store4(dst, o[q]);
            } else {
                store4(part + (size_t)si.slot * 2 * ROW + 4 * x, o[q]);       // [dZ row | dH row]
            }
        }
    }
}

// ---------------------------------------------------------------------------- one-pass training scorer, wave per entry, wide rows
// Round 5: the wave-per-entry form for rows of K*D = 2,048 elements (K = 16, d = 128: BASELINE configs[4]), where the
// group-per-entry kernel above needs 256 registers (one wave per SIMD, 104 ms on the Penn94-shaped graph with bf16 tables)
// and the module fell back to three separate kernels (22.5 ms, 3x the forward's gathers).
//
// A lane holds CHUNKS of 16 bytes of a table row as they lie in memory — 8 bf16 or 4 fp32 elements — chunk number
// j * 64 + lane of the row, NJ chunks per lane and table: a factor slice (d = 128) is G = 16 consecutive lanes with bf16
// tables (one DPP row), 32 with fp32 tables; lane group r holds the factors r, r + 64/G, ...  bf16 chunks stay PACKED in
// the registers (a gathered entry = 32 registers instead of 64): the dot products with the node's own rows (packed too,
// re-read from the wave's LDS region every step) are v_dot2c_f32_bf16 — two exact products and the running fp32 sum per
// instruction, symmetric in its operands, so both endpoints of a pair still form the same bits — and the elements are
// widened only where they are accumulated (a shift / a mask each).  Per lane: 64 accumulator registers (the node's
// [dZ row | dH row], 2 x 32 elements) + U gathered entries.  bf16: U = 1, four waves per SIMD, 35 KB of LDS per
// workgroup; fp32: U = 1 at two waves per SIMD (its own rows alone are 64 KB of LDS per workgroup).
// The 2 U NJ partial dot products of a step are reduced over the G lanes by one transposed reduction (value index = the
// top bits of the lane's position in its group: z.z below G/2, h.h above), one expf and one sigmoid per step, the
// coefficients formed in the lanes that hold them and broadcast by DPP moves, as in the D = 64 kernel above.
// Unit sum: the own-row regions (4 x 8 KB with bf16 tables) are too small to stage four [dZ | dH] rows of 16 KB at
// once, so the unit is summed as a TREE in two rounds through two 16 KB slots — (s0 + s1) + (s2 + s3); which waves pair
// up depends on the row's segments alone (shard-independent, reproducible).
#ifndef DL_TRAIN_WIDE_KERNEL
#define DL_TRAIN_WIDE_KERNEL 1        // -DDL_TRAIN_WIDE_KERNEL=0: the group-per-entry kernel, for A/B runs
#endif
#ifndef DL_TRAIN_WIDE_U_BF16
#define DL_TRAIN_WIDE_U_BF16 1        // entries per step with bf16 tables ...
#endif
#ifndef DL_TRAIN_WIDE_WAVES_BF16
#define DL_TRAIN_WIDE_WAVES_BF16 4    // ... and waves per SIMD (U = 2 needs 3)
#endif
#ifndef DL_TRAIN_WIDE_U_F32
#define DL_TRAIN_WIDE_U_F32 1         // fp32 tables: U = 2 spills 17 registers at the 256 the two waves per SIMD allow
#endif
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 v2bf16 __attribute__((ext_vector_type(2)));

template <typename T>
struct WideChunk;
template <>
struct WideChunk<bf16_t> {
    static constexpr int CH = 8;                                    // elements per 16-byte chunk
    // (the dwords are copied into scalars first: __builtin_bit_cast applied to a vector ELEMENT lvalue — bit_cast(a.y) — read
    // element 0 every time with hipcc 7.2: four dot2c on the same registers)
    static __device__ __forceinline__ float dot2(unsigned a, unsigned b, float s) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf16, a), __builtin_bit_cast(v2bf16, b), s, false);
    }
    static __device__ __forceinline__ float dot(const u32x4& a, const u32x4& b) {
        const unsigned a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
        return dot2(a3, b3, dot2(a2, b2, dot2(a1, b1, dot2(a0, b0, 0.0f))));
    }
    static __device__ __forceinline__ void fma(float (&acc)[8], float c, const u32x4& x) {
        acc[0] = fmaf(c, __uint_as_float(x.x << 16), acc[0]); acc[1] = fmaf(c, __uint_as_float(x.x & 0xffff0000u), acc[1]);
        acc[2] = fmaf(c, __uint_as_float(x.y << 16), acc[2]); acc[3] = fmaf(c, __uint_as_float(x.y & 0xffff0000u), acc[3]);
        acc[4] = fmaf(c, __uint_as_float(x.z << 16), acc[4]); acc[5] = fmaf(c, __uint_as_float(x.z & 0xffff0000u), acc[5]);
        acc[6] = fmaf(c, __uint_as_float(x.w << 16), acc[6]); acc[7] = fmaf(c, __uint_as_float(x.w & 0xffff0000u), acc[7]);
    }
};
template <>
struct WideChunk<float> {
    static constexpr int CH = 4;
    static __device__ __forceinline__ float dot(const u32x4& a, const u32x4& b) {
        return dot4_packed(make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)),
                           make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)));
    }
    static __device__ __forceinline__ void fma(float (&acc)[4], float c, const u32x4& x) {
        acc[0] = fmaf(c, __uint_as_float(x.x), acc[0]); acc[1] = fmaf(c, __uint_as_float(x.y), acc[1]);
        acc[2] = fmaf(c, __uint_as_float(x.z), acc[2]); acc[3] = fmaf(c, __uint_as_float(x.w), acc[3]);
    }
};

template <int K, int D, typename T>
struct TrainWide {
    static constexpr bool ok = DL_TRAIN_WIDE_KERNEL && K == 16 && D == 128;
    static constexpr int U = sizeof(T) == 2 ? DL_TRAIN_WIDE_U_BF16 : DL_TRAIN_WIDE_U_F32;
    static constexpr int WAVES = sizeof(T) == 2 ? DL_TRAIN_WIDE_WAVES_BF16 : 2;      // fp32: 67 KB of LDS per workgroup
};

// value of lane `idx` (0 .. 15, a constant once the caller's loops are unrolled) of this lane's DPP row: row_newbcast
__device__ __forceinline__ float row_bcast_idx(float v, int idx) {
    switch (idx) {
        case 0: return dpp_mov<0x150>(v);   case 1: return dpp_mov<0x151>(v);   case 2: return dpp_mov<0x152>(v);
        case 3: return dpp_mov<0x153>(v);   case 4: return dpp_mov<0x154>(v);   case 5: return dpp_mov<0x155>(v);
        case 6: return dpp_mov<0x156>(v);   case 7: return dpp_mov<0x157>(v);   case 8: return dpp_mov<0x158>(v);
        case 9: return dpp_mov<0x159>(v);   case 10: return dpp_mov<0x15A>(v);  case 11: return dpp_mov<0x15B>(v);
        case 12: return dpp_mov<0x15C>(v);  case 13: return dpp_mov<0x15D>(v);  case 14: return dpp_mov<0x15E>(v);
        default: return dpp_mov<0x15F>(v);
    }
}

// sum over the factor chunks j of a step: xor offsets OFF, OFF/2, ..., LO inside the lane group
template <int OFF, int LO>
__device__ __forceinline__ float sum_over_chunks(float v) {
    if constexpr (OFF >= LO) {
        return sum_over_chunks<OFF / 2, LO>(v + xor_lane<OFF>(v));
    } else {
        return v;
    }
}

template <int K, int D, typename T, int U, int WAVES, bool T1>
__global__ __launch_bounds__(BLOCK, WAVES) void score_train_wide_kernel(
        dl_csr_plan g, const int32_t* __restrict__ inc_pair, const T* __restrict__ Z, const T* __restrict__ H, float t,
        float* __restrict__ dZ, float* __restrict__ dH, float* __restrict__ part, const float* __restrict__ y,
        const float* __restrict__ w, float* __restrict__ prob_out) {
    using WC = WideChunk<T>;
    constexpr int CH = WC::CH, ROW = K * D;
    constexpr int NJ = ROW / (DL_WAVE * CH);                        // chunks per lane and table row
    constexpr int G = D / CH;                                       // lanes per factor slice
    constexpr int NVAL = 2 * U * NJ;                                // partial dot products per lane and step
    constexpr int DUPL = G / NVAL;                                  // lanes that end up with the same complete sum
    constexpr int F4 = CH / 4;                                      // float4 per accumulator chunk
    constexpr int OWN16 = 2 * NJ * DL_WAVE;                         // 16-byte chunks of one wave's own rows [Z | H]
    constexpr int SLOT16 = 2 * NJ * F4 * DL_WAVE;                   // float4s of one [dZ row | dH row]
    constexpr int LDS16 = WAVES_PER_BLOCK * OWN16 > 2 * SLOT16 ? WAVES_PER_BLOCK * OWN16 : 2 * SLOT16;
    static_assert(ROW % (DL_WAVE * CH) == 0 && (G == 16 || G == 32) && NVAL <= G && DUPL * NVAL == G, "lane geometry");
    static_assert((U & (U - 1)) == 0 && U * DUPL * NJ * 2 == G, "value index = (chunk, entry) in the top bits of the group position");
    __shared__ __attribute__((aligned(16))) u32x4 lds[LDS16];
    __shared__ float ent_y[WAVES_PER_BLOCK][DL_WAVE], ent_w[WAVES_PER_BLOCK][DL_WAVE];
    __shared__ int ent_q[WAVES_PER_BLOCK][DL_WAVE];
    const WaveSeg ws = load_wave_seg(g);
    const SegInfo si = ws.si;
    const int wave = ws.wave, lane = lane_id();
    const int c = lane & (G - 1);                                   // position in the lane group
    float az[NJ][CH], ah[NJ][CH];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < CH; ++e) az[j][e] = ah[j][e] = 0.0f;
    u32x4* const mine = lds + wave * OWN16;
    if (ws.active) {
        const u32x4* zu = reinterpret_cast<const u32x4*>(Z + (size_t)si.grow * ROW);
        const u32x4* hu = reinterpret_cast<const u32x4*>(H + (size_t)si.grow * ROW);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            mine[j * DL_WAVE + lane] = zu[j * DL_WAVE + lane];       // read back by this lane only: no barrier needed
            mine[(NJ + j) * DL_WAVE + lane] = hu[j * DL_WAVE + lane];
        }
        int my_col = si.grow;
        {
            int my_q = 0;
            float my_y = 0.0f, my_w = 0.0f;                         // w = 0 past the segment end: no gradient, no output
            if (si.beg + lane < si.end) {
                my_col = g.col[si.beg + lane];
                my_q = inc_pair[si.beg + lane];
                my_y = y[my_q];
                my_w = w[my_q];
            }
            ent_y[wave][lane] = my_y;                               // written and read by this wave only: no barrier
            ent_w[wave][lane] = my_w;
            ent_q[wave][lane] = my_q;
        }
        const int nsteps = (si.end - si.beg + U - 1) / U;           // entries past the end repeat a valid row with w = 0
        for (int step = 0; step < nsteps; ++step) {
            u32x4 zv[U][NJ], hv[U][NJ];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // the entry is wave-uniform: its row address is a scalar base, the lane offset a constant
                const size_t v = (size_t)(unsigned)__builtin_amdgcn_readlane(my_col, (step * U + u) & 63);
                const T* zs = Z + v * ROW;
                const T* hs = H + v * ROW;
                asm volatile("" : "+s"(zs), "+s"(hs));
                const u32x4* zr = reinterpret_cast<const u32x4*>(zs) + lane;
                const u32x4* hr = reinterpret_cast<const u32x4*>(hs) + lane;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    zv[u][j] = zr[j * DL_WAVE];
                    hv[u][j] = hr[j * DL_WAVE];
                }
            }
            // partial dot products: index = table * (U NJ) + j * U + u
            float val[NVAL];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const u32x4 a16 = mine[j * DL_WAVE + lane], b16 = mine[(NJ + j) * DL_WAVE + lane];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    val[j * U + u] = WC::dot(a16, zv[u][j]);
                    val[U * NJ + j * U + u] = WC::dot(b16, hv[u][j]);
                }
            }
            // NVAL values over the G lanes of the group, halving the value count at every exchange: the lane at position c
            // ends with complete sum number c / DUPL — below G/2: z_u . z_v of (chunk, entry), above: h_u . h_v of the same
            TransposedReduce<NVAL, G / 2>::run(val, c);
            const float mine_v = val[0];
            const float ex = expf(T1 ? mine_v : mine_v / t);        // (the upper half exponentiates a value nobody reads)
            const float ttv = mine_v * xor_lane<G / 2>(ex);          // valid above G/2: (h.h) e^(z.z/t) of (chunk, entry)
            // sum over the chunks of this lane group (the chunk index sits in the bits above the entry and the duplicates),
            // then over the lane groups: all factors
            float logit = sum_over_chunks<G / 4, G / (2 * NJ)>(ttv);
            if constexpr (G == 16) logit = add_xor<16>(logit);
            logit = add_xor<32>(logit);
            const float p = sigmoid_ref(logit);
            const int idx = step * U + (((c & (G / 2 - 1)) / DUPL) & (U - 1));
            const float yy = ent_y[wave][idx & 63], ww = ent_w[wave][idx & 63];
            const int qq = ent_q[wave][idx & 63];
            const float pr = p * (1.0f - p);
            const float gl = ww == 0.0f ? 0.0f : ww * (p - yy) * (pr >= 1e-12f ? 1.0f : pr * 1e12f);     // valid above G/2
            if (lane >= G / 2 && lane < G / 2 + U * DUPL && (lane & (DUPL - 1)) == 0 && si.beg + idx < si.end) prob_out[qq] = p;
            const float exc = fminf(ex, 3.402823466e38f);          // 0 * inf must stay 0 (see the D = 64 kernel)
            const float ttc = __builtin_amdgcn_fmed3f(T1 ? ttv : ttv / t, -3.402823466e38f, 3.402823466e38f);
            const float gl_partner = xor_lane<G / 2>(gl);           // OUTSIDE the select (a DPP move under a divergent branch reads 0)
            const float coef = (c & (G / 2)) ? gl * ttc : gl_partner * exc;
            // the DPP row that holds the coefficient of (side, chunk, entry): G = 16: this row, the h.h side 8 lanes up;
            // G = 32: the lower row of the group holds the z.z side (-> dH), the upper row the h.h side (-> dZ)
            float ce = coef, ct = coef;
            if constexpr (G == 32) {
                const float other = xor_lane<16>(coef);
                ce = (c & 16) ? other : coef;
                ct = (c & 16) ? coef : other;
            }
            constexpr int TB = G == 16 ? 8 : 0;                     // position of the h.h side inside its DPP row
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float ch = row_bcast_idx(ce, (j * U + u) * DUPL);
                    const float cz = row_bcast_idx(ct, TB + (j * U + u) * DUPL);
                    WC::fma(ah[j], ch, hv[u][j]);
                    WC::fma(az[j], cz, zv[u][j]);
                }
            }
        }
    }
    // ---- unit sum, as a tree in two rounds through two [dZ row | dH row] slots (the own rows are dead behind the barrier)
    float4* const slot4 = reinterpret_cast<float4*>(lds);
    auto put = [&](int slot) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int f = 0; f < F4; ++f) {
                slot4[slot * SLOT16 + (j * F4 + f) * DL_WAVE + lane] = make_float4(az[j][4 * f], az[j][4 * f + 1], az[j][4 * f + 2], az[j][4 * f + 3]);
                slot4[slot * SLOT16 + ((NJ + j) * F4 + f) * DL_WAVE + lane] = make_float4(ah[j][4 * f], ah[j][4 * f + 1], ah[j][4 * f + 2], ah[j][4 * f + 3]);
            }
    };
    auto add = [&](int slot) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int f = 0; f < F4; ++f) {
                const float4 a4 = slot4[slot * SLOT16 + (j * F4 + f) * DL_WAVE + lane];
                const float4 b4 = slot4[slot * SLOT16 + ((NJ + j) * F4 + f) * DL_WAVE + lane];
                az[j][4 * f] += a4.x; az[j][4 * f + 1] += a4.y; az[j][4 * f + 2] += a4.z; az[j][4 * f + 3] += a4.w;
                ah[j][4 * f] += b4.x; ah[j][4 * f + 1] += b4.y; ah[j][4 * f + 2] += b4.z; ah[j][4 * f + 3] += b4.w;
            }
    };
    const int upos = ws.upos, nfwd = ws.active ? ws.n_unit : 0;
    __syncthreads();
    if (ws.active && (upos & 1)) put(wave >> 1);                                      // round A: odd positions hand over ...
    __syncthreads();
    if (ws.active && !(upos & 1) && nfwd >= 2) add((wave + 1) >> 1);                  // ... to the even position below them
    __syncthreads();
    if (ws.active && upos == 2) put(0);                                               // round B: (s2 + s3) ...
    __syncthreads();
    if (!ws.head) return;
    if (nfwd >= 3) add(0);                                                            // ... joins (s0 + s1)
    float* oz = si.slot < 0 ? dZ + (size_t)si.grow * ROW : part + (size_t)si.slot * 2 * ROW;        // [dZ row | dH row]
    float* oh = si.slot < 0 ? dH + (size_t)si.grow * ROW : part + (size_t)si.slot * 2 * ROW + ROW;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int f = 0; f < F4; ++f) {
            store4(oz + (size_t)(j * DL_WAVE + lane) * CH + 4 * f, make_float4(az[j][4 * f], az[j][4 * f + 1], az[j][4 * f + 2], az[j][4 * f + 3]));
            store4(oh + (size_t)(j * DL_WAVE + lane) * CH + 4 * f, make_float4(ah[j][4 * f], ah[j][4 * f + 1], ah[j][4 * f + 2], ah[j][4 * f + 3]));
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

// ---------------------------------------------------------------------------- pair-list BCE
// loss = sum_q w[q] * bce(prob[q], y[q]),  g[q] = dloss/dprob[q], in PROBABILITY space exactly as
// F.binary_cross_entropy does it (main_disentangled.py:195): log clamped at -100, gradient
// (p - y) / max(p (1 - p), 1e-12).  Saturated fp32 sigmoids keep their zero gradient downstream because the
// scorer backward multiplies by p (1 - p).  Deterministic two-stage reduction (no float atomics).
constexpr int BCE_BLOCKS = 1024;     // 4 workgroups per CU (256 left one: 12.8 us for 1.1M pairs, latency-bound)

__global__ __launch_bounds__(BLOCK) void pair_bce_kernel(const float* __restrict__ prob, const float* __restrict__ y,
                                                         const float* __restrict__ w, int n, float* __restrict__ g,
                                                         float* __restrict__ partial) {
    __shared__ float red[WAVES_PER_BLOCK];
    float acc = 0.0f;
    for (int q = blockIdx.x * BLOCK + threadIdx.x; q < n; q += BCE_BLOCKS * BLOCK) {
        const float p = prob[q], yy = y[q], ww = w[q];
        // a NaN probability must stay visible as a NaN loss (fmaxf would turn its log into -100 and hide it; torch's
        // BCE refuses such input outright)
        const float lg = logf(p), lg1 = logf(1.0f - p);
        const float lp = lg < -100.0f ? -100.0f : lg, l1p = lg1 < -100.0f ? -100.0f : lg1;
        // weight 0 = "not part of the loss" (e.g. validation pairs riding along): exactly nothing, even for a NaN p
        acc += ww == 0.0f ? 0.0f : ww * -(yy * lp + (1.0f - yy) * l1p);
        g[q] = ww == 0.0f ? 0.0f : ww * (p - yy) / fmaxf(p * (1.0f - p), 1e-12f);
    }
    acc = wave_allreduce_sum(acc);
    if (lane_id() == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = red[0];
        for (int i = 1; i < WAVES_PER_BLOCK; ++i) t += red[i];
        partial[blockIdx.x] = t;
    }
}

__global__ void pair_bce_finish_kernel(const float* __restrict__ partial, float* __restrict__ loss) {
    const int lane = threadIdx.x;                           // one wave
    float acc = 0.0f;
    for (int i = lane; i < BCE_BLOCKS; i += DL_WAVE) acc += partial[i];
    acc = wave_allreduce_sum(acc);
    if (lane == 0) loss[0] = acc;
}

// ---------------------------------------------------------------------------- host launchers
static inline int pow2_at_least(int k) {
    int p = 1;
    while (p < k) p <<= 1;
    return p;
}

#ifdef DL_EXP_ROWSUM_SCATTER
static const int32_t* exp_rev_ptr = nullptr;      // set by route_fwd below (experiment builds only)
static float* exp_wn_ptr = nullptr;
#endif
static void launch_s_rowsum(const dl_csr_plan* g, int K, const uint8_t* p, const float* a, float* s, hipStream_t st) {
    if (g->n_seg <= 0) return;
#ifdef DL_EXP_ROWSUM_SCATTER
    {
        const int reg = (g->n_seg + ROWSUM_POS_PER_BLOCK - 1) / ROWSUM_POS_PER_BLOCK;
        const dim3 grid((unsigned)reg + wave_blocks(g->n_multi)), block(BLOCK);
        if (K <= 8 && K > 4) {
            hipLaunchKernelGGL(s_rowsum_thread_kernel<8>, grid, block, 0, st, *g, K, p, a, s, reg, exp_rev_ptr, exp_wn_ptr);
            return;
        }
    }
#endif
    const int reg = (g->n_seg + ROWSUM_POS_PER_BLOCK - 1) / ROWSUM_POS_PER_BLOCK;
    const dim3 grid((unsigned)reg + wave_blocks(g->n_multi)), block(BLOCK);
    if (K <= 4) hipLaunchKernelGGL(s_rowsum_thread_kernel<4>, grid, block, 0, st, *g, K, p, a, s, reg);
    else if (K <= 8) hipLaunchKernelGGL(s_rowsum_thread_kernel<8>, grid, block, 0, st, *g, K, p, a, s, reg);
    else if (K <= 16) hipLaunchKernelGGL(s_rowsum_thread_kernel<16>, grid, block, 0, st, *g, K, p, a, s, reg);
    else hipLaunchKernelGGL(s_rowsum_thread_kernel<32>, grid, block, 0, st, *g, K, p, a, s, reg);   // tuned shapes: K <= 32
}

static void launch_vec_combine(const dl_csr_plan* g, int K, const float* part, int mode, const float* s_raw,
                               float* out, hipStream_t st) {
    if (g->n_multi <= 0) return;
    hipLaunchKernelGGL(vec_combine_kernel, dim3(wave_blocks(g->n_multi)), dim3(BLOCK), 0, st, *g, K, pow2_at_least(K),
                       part, mode, s_raw, out);
}

template <int K, int D, typename T>
struct Ops {
    static constexpr int ROW = K * D;

    // The H rows of the aggregation go out as streaming stores when the table is far beyond the caches (256 MiB Infinity
    // Cache); on cache-resident graphs the scorer wants them in cache.  (The dZ / dH rows of the training kernels were
    // tried too: a snap-patents-sized epoch 311 -> 319 ms with all of them streaming — only the forward H store pays.)
    static int stream_rows(const dl_csr_plan* g, size_t elem = sizeof(float)) {
        static const char* force = getenv("DL_STREAM_ROWS");      // 0 / 1: measurements only
        if (force) return force[0] == '1';
        return (size_t)g->n_total * ROW * elem > ((size_t)256 << 20) ? 1 : 0;
    }

    // `route`: the (possibly sliced / upper-triangle) plan the routing kernel walks, NULL = g itself;
    // mirror: it covers col >= row only and every result is also written through rev
    static int route_fwd(const dl_csr_plan* g, const dl_csr_plan* route, bool mirror, const int32_t* rev,
                         const void* Z, float t, uint8_t* p, float* a, float* s, float* s_part, hipStream_t st) {
        const dl_csr_plan* rp = route ? route : g;
        // DL_ROUTE_BALLOT=1: the ballot arg-max (A/B switch; see the kernel)
        static const bool ballot = getenv("DL_ROUTE_BALLOT") && atoi(getenv("DL_ROUTE_BALLOT")) != 0;
        if (mirror) {
            if (ballot)
                hipLaunchKernelGGL((route_seg_kernel<K, D, T, true, true>), dim3(seg_blocks(rp)), dim3(BLOCK), 0, st, *rp, rev,
                                   (const T*)Z, t, p, a);
            else
                hipLaunchKernelGGL((route_seg_kernel<K, D, T, true>), dim3(seg_blocks(rp)), dim3(BLOCK), 0, st, *rp, rev,
                                   (const T*)Z, t, p, a);
        } else {
            if (ballot)
                hipLaunchKernelGGL((route_seg_kernel<K, D, T, false, true>), dim3(seg_blocks(rp)), dim3(BLOCK), 0, st, *rp, rev,
                                   (const T*)Z, t, p, a);
            else
                hipLaunchKernelGGL((route_seg_kernel<K, D, T, false>), dim3(seg_blocks(rp)), dim3(BLOCK), 0, st, *rp, rev,
                                   (const T*)Z, t, p, a);
        }
#ifdef DL_EXP_ROWSUM_SCATTER
        exp_rev_ptr = mirror ? rev : nullptr;          // unsharded graphs only (they have the reverse-edge map)
        if (exp_wn_ptr == nullptr && hipMalloc(&exp_wn_ptr, (size_t)64 << 20) != hipSuccess) exp_rev_ptr = nullptr;   // experiment scratch
        if ((size_t)g->n_entries * sizeof(float) > ((size_t)64 << 20)) exp_rev_ptr = nullptr;
        (void)s_part;
#else
        (void)s_part;
#endif
        launch_s_rowsum(g, K, p, a, s, st);
        return check_launch("route_fwd(fast)");
    }

    static int aggregate_fwd(const dl_csr_plan* g, const void* Z, float beta, const uint8_t* p, const float* a,
                             const float* s, void* H, float* h_part, hipStream_t st) {
        hipLaunchKernelGGL((aggregate_cls_kernel<K, D, T, 4>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, (const T*)Z,
                           beta, p, a, s, (T*)H, h_part, stream_rows(g, sizeof(T)));
        if (g->n_multi > 0)
            hipLaunchKernelGGL((row_combine_kernel<ROW, T, T>), dim3(g->n_multi), dim3(BLOCK), 0, st, *g, h_part, ROW,
                               (const T*)Z, beta, 1.0f - beta, (T*)H, 0);
        return check_launch("aggregate_fwd(fast)");
    }

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
        hipLaunchKernelGGL((bwd_phase2_seg_kernel<K, D, T>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, (const T*)Z,
                           dH, beta, t, p, a, s, dw, dwr, ds, dz_in, scale, dZ, dz_part);
        if (g->n_multi > 0)
            hipLaunchKernelGGL((row_combine_kernel<ROW, float, float>), dim3(g->n_multi), dim3(BLOCK), 0, st, *g,
                               dz_part, ROW, dH, beta, 1.0f, dZ, 0, (const float*)nullptr, (float*)nullptr, dz_in, scale);
        return check_launch("route_aggregate_bwd_phase2(fast)");
    }

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

    // training step of the scorer in one pass: prob, and dZ / dH for the weighted BCE of (y, w)
    static int score_train(const dl_pair_incidence* inc, const void* Z, const void* H, float t, const float* y,
                           const float* w, float* prob, float* dZ, float* dH, float* part, hipStream_t st) {
        const dl_csr_plan* g = &inc->csr;
        const float* no_x = nullptr;
        if constexpr (std::is_same<T, float>::value && TrainWave<K, D>::ok) {
            if (g->seg_len <= 64 && g->seg_len % TrainWave<K, D>::U == 0 && !getenv("DL_TRAIN_GROUP_KERNEL")) {
                auto launch = [&](auto kern) {
                    hipLaunchKernelGGL(kern, dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, inc->inc_pair, (const float*)Z,
                                       (const float*)H, t, dZ, dH, part, y, w, prob);
                };
                if (t == 1.0f) launch(score_train_wave_kernel<K, D, true>);
                else launch(score_train_wave_kernel<K, D, false>);
                if (g->n_multi > 0)
                    hipLaunchKernelGGL((row_combine_kernel<ROW, float, float>), dim3(g->n_multi, 2), dim3(BLOCK), 0, st, *g,
                                       part, 2 * ROW, no_x, 0.0f, 1.0f, dZ, 0, part + ROW, dH);
                return check_launch("score_pairs_train(fast, wave per entry)");
            }
        }
        if constexpr (TrainWide<K, D, T>::ok) {
            constexpr int U = TrainWide<K, D, T>::U, WV = TrainWide<K, D, T>::WAVES;
            if (g->seg_len <= 64 && g->seg_len % U == 0 && !getenv("DL_TRAIN_GROUP_KERNEL")) {
                auto launch = [&](auto kern) {
                    hipLaunchKernelGGL(kern, dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, inc->inc_pair, (const T*)Z, (const T*)H,
                                       t, dZ, dH, part, y, w, prob);
                };
                if (t == 1.0f) launch(score_train_wide_kernel<K, D, T, U, WV, true>);
                else launch(score_train_wide_kernel<K, D, T, U, WV, false>);
                if (g->n_multi > 0)
                    hipLaunchKernelGGL((row_combine_kernel<ROW, float, float>), dim3(g->n_multi, 2), dim3(BLOCK), 0, st, *g,
                                       part, 2 * ROW, no_x, 0.0f, 1.0f, dZ, 0, part + ROW, dH);
                return check_launch("score_pairs_train(fast, wave per entry, wide rows)");
            }
        }
        hipLaunchKernelGGL((score_bwd_seg_kernel<K, D, T, true>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, inc->inc_pair,
                           (const T*)Z, (const T*)H, t, no_x, no_x, dZ, dH, part, y, w, prob);
        if (g->n_multi > 0)
            hipLaunchKernelGGL((row_combine_kernel<ROW, float, float>), dim3(g->n_multi, 2), dim3(BLOCK), 0, st, *g, part,
                               2 * ROW, no_x, 0.0f, 1.0f, dZ, 0, part + ROW, dH);
        return check_launch("score_pairs_train(fast)");
    }
};

}  // namespace fast

int pair_bce(const float* prob, const float* y, const float* w, int n, float* loss, float* g, float* partial,
             hipStream_t st) {
    hipLaunchKernelGGL(fast::pair_bce_kernel, dim3(fast::BCE_BLOCKS), dim3(BLOCK), 0, st, prob, y, w, n, g, partial);
    hipLaunchKernelGGL(fast::pair_bce_finish_kernel, dim3(1), dim3(DL_WAVE), 0, st, partial, loss);
    return check_launch("pair_bce");
}

// (K, D) pairs with a tuned instantiation.  D must be 4 * a power of two.
#define DL_FAST_SHAPES_F32(X) \
    X(4, 32) X(8, 64) X(16, 128) X(5, 32) X(5, 64) X(10, 32) X(10, 64) X(20, 32) X(8, 32) X(4, 64) X(4, 8) X(8, 8) X(3, 8)
#define DL_FAST_SHAPES_BF16(X) X(4, 32) X(8, 64) X(16, 128) X(5, 64) X(8, 32)

bool fast_supported(int K, int d, int dtype) {
#define X(KK, DD) if (K == KK && d == DD) return true;
    if (dtype == DL_F32) { DL_FAST_SHAPES_F32(X) }
    if (dtype == DL_BF16) { DL_FAST_SHAPES_BF16(X) }
#undef X
    return false;
}

// CALL(OPS) is expanded with OPS = fast::Ops<K, D, T> of the matching instantiation
#define DL_DISPATCH(CALL)                                                                         \
    if (dtype == DL_F32) {                                                                        \
        DL_FAST_SHAPES_F32(CALL##_F32)                                                            \
    } else if (dtype == DL_BF16) {                                                                \
        DL_FAST_SHAPES_BF16(CALL##_BF16)                                                          \
    }                                                                                             \
    set_error("no tuned kernel for K=%d d=%d dtype=%d", K, d, dtype);                             \
    return DL_E_ARG;

int fast_route_fwd(const dl_csr_plan* g, const dl_csr_plan* route, bool mirror, const int32_t* rev, const void* Z,
                   int K, int d, int dtype, float t, uint8_t* p, float* a, float* s, float* s_part, hipStream_t st) {
#define X_F32(KK, DD) \
    if (K == KK && d == DD) return fast::Ops<KK, DD, float>::route_fwd(g, route, mirror, rev, Z, t, p, a, s, s_part, st);
#define X_BF16(KK, DD)      \
    if (K == KK && d == DD) \
        return fast::Ops<KK, DD, fast::bf16_t>::route_fwd(g, route, mirror, rev, Z, t, p, a, s, s_part, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

int fast_aggregate_fwd(const dl_csr_plan* g, const void* Z, int K, int d, int dtype, float beta, const uint8_t* p,
                       const float* a, const float* s, void* H, float* h_part, hipStream_t st) {
#define X_F32(KK, DD) if (K == KK && d == DD) return fast::Ops<KK, DD, float>::aggregate_fwd(g, Z, beta, p, a, s, H, h_part, st);
#define X_BF16(KK, DD) if (K == KK && d == DD) return fast::Ops<KK, DD, fast::bf16_t>::aggregate_fwd(g, Z, beta, p, a, s, H, h_part, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

int fast_bwd_phase1(const dl_csr_plan* g, const void* Z, int K, int d, int dtype, float beta, const uint8_t* p,
                    const float* a, const float* s, const float* dH, float* dw, float* dwr, float* ds,
                    float* ds_part, hipStream_t st) {
#define X_F32(KK, DD) if (K == KK && d == DD) return fast::Ops<KK, DD, float>::bwd_phase1(g, Z, beta, p, a, s, dH, dw, dwr, ds, ds_part, st);
#define X_BF16(KK, DD) if (K == KK && d == DD) return fast::Ops<KK, DD, fast::bf16_t>::bwd_phase1(g, Z, beta, p, a, s, dH, dw, dwr, ds, ds_part, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

int fast_bwd_phase2(const dl_csr_plan* g, const void* Z, int K, int d, int dtype, float beta, float t,
                    const uint8_t* p, const float* a, const float* s, const float* dH, const float* dw,
                    const float* dwr, const float* ds, const float* dz_in, const float* scale, float* dZ, float* dz_part,
                    hipStream_t st) {
#define X_F32(KK, DD) if (K == KK && d == DD) return fast::Ops<KK, DD, float>::bwd_phase2(g, Z, beta, t, p, a, s, dH, dw, dwr, ds, dz_in, scale, dZ, dz_part, st);
#define X_BF16(KK, DD) if (K == KK && d == DD) return fast::Ops<KK, DD, fast::bf16_t>::bwd_phase2(g, Z, beta, t, p, a, s, dH, dw, dwr, ds, dz_in, scale, dZ, dz_part, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

int fast_score_pairs_fwd(const dl_pair_incidence* by_u, const void* Z, const void* H, int K, int d, int dtype,
                         float t, float* prob, float* coef, hipStream_t st) {
#define X_F32(KK, DD) if (K == KK && d == DD) return fast::Ops<KK, DD, float>::score_fwd(by_u, Z, H, t, prob, coef, st);
#define X_BF16(KK, DD) if (K == KK && d == DD) return fast::Ops<KK, DD, fast::bf16_t>::score_fwd(by_u, Z, H, t, prob, coef, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

int fast_score_pairs_bwd(const dl_pair_incidence* inc, const void* Z, const void* H, int K, int d, int dtype,
                         float t, const float* prob, const float* g_prob, const float* coef, float* dZ, float* dH,
                         float* part, hipStream_t st) {
#define X_F32(KK, DD) if (K == KK && d == DD) return fast::Ops<KK, DD, float>::score_bwd(inc, Z, H, t, prob, g_prob, coef, dZ, dH, part, st);
#define X_BF16(KK, DD) if (K == KK && d == DD) return fast::Ops<KK, DD, fast::bf16_t>::score_bwd(inc, Z, H, t, prob, g_prob, coef, dZ, dH, part, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

int fast_score_pairs_train(const dl_pair_incidence* inc, const void* Z, const void* H, int K, int d, int dtype, float t,
                           const float* y, const float* w, float* prob, float* dZ, float* dH, float* part, hipStream_t st) {
#define X_F32(KK, DD) if (K == KK && d == DD) return fast::Ops<KK, DD, float>::score_train(inc, Z, H, t, y, w, prob, dZ, dH, part, st);
#define X_BF16(KK, DD) if (K == KK && d == DD) return fast::Ops<KK, DD, fast::bf16_t>::score_train(inc, Z, H, t, y, w, prob, dZ, dH, part, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

int fast_score_allpairs_fwd(const void* Z, const void* H, int N, int K, int d, int dtype, float t, float* prob,
                            hipStream_t st) {
#define X_F32(KK, DD) if (K == KK && d == DD) return fast::Ops<KK, DD, float>::score_allpairs(Z, H, N, t, prob, st);
#define X_BF16(KK, DD) if (K == KK && d == DD) return fast::Ops<KK, DD, fast::bf16_t>::score_allpairs(Z, H, N, t, prob, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

}  // namespace dl
