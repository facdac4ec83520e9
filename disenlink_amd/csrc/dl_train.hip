// One-pass training scorer (scorer forward + weighted BCE gradient of main_disentangled.py:195 + scorer backward) and the pair-list BCE.
// (one of the tuned-kernel translation units; the shared pieces and the design notes are in dl_fast.h)
#include "dl_fast.h"
#include "dl_score_bwd.h"

namespace dl {
namespace fast {

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

// (dot4_packed: dl_fast.h — shared with the wave-per-entry forward scorer of dl_score.hip)

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
                if (w == nullptr) {                                 // y = per-entry (label, signed weight) pairs: dl_pair_incidence.entry_yw
                    const float2 yw = reinterpret_cast<const float2*>(y)[si.beg + lane];
                    my_y = yw.x;
                    my_w = yw.y;
                } else {
                    my_y = y[my_q];
                    my_w = w[my_q];
                }
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
                const auto* zr = uniform_row<dl_vf4>(Z, v * ROW * sizeof(float));
                const auto* hr = uniform_row<dl_vf4>(H, v * ROW * sizeof(float));
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    zv[e][j] = as_float4(zr[(unsigned)(lane + j * 64)]);
                    hv[e][j] = as_float4(hr[(unsigned)(lane + j * 64)]);
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
            const float yy = ent_y[wave][idx & 63], wsg = ent_w[wave][idx & 63];
            const float ww = fabsf(wsg);                           // a negative sign (per-entry weights only) = the pair's other entry writes prob
            const int qq = ent_q[wave][idx & 63];
            // dl_pair_bce's gradient times the sigmoid backward: w (p - y) / max(q, 1e-12) * q with q = p (1 - p) — i.e.
            // w (p - y) itself unless q underflows the clamp (saturated scores: q = 0 gives exactly 0), without the division
            const float pr = p * (1.0f - p);
            const float gl = ww == 0.0f ? 0.0f : ww * (p - yy) * (pr >= 1e-12f ? 1.0f : pr * 1e12f);     // valid in lanes 8..15
            if (lane >= 8 && lane < 12 && si.beg + idx < si.end && !(__float_as_uint(wsg) >> 31)) prob_out[qq] = p;
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
    // (rows of several units summed by their last unit inside this launch — publish_unit_and_sum_row, as the aggregation does —
    // were measured here and lose: with XCD slicing EVERY row has 4-16 units, and the last unit's chain of slot reads sits
    // in the kernel's tail: squirrel +27 us at 8 slices, Penn94-sized K = 8 +540 us at 16; profiles/r7b_train_scorer_ab.txt)
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
                if (w == nullptr) {                                 // y = per-entry (label, signed weight) pairs: dl_pair_incidence.entry_yw
                    const float2 yw = reinterpret_cast<const float2*>(y)[si.beg + lane];
                    my_y = yw.x;
                    my_w = yw.y;
                } else {
                    my_y = y[my_q];
                    my_w = w[my_q];
                }
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
                // (the scalar-base global_load form of the d = 64 kernels — uniform_row, dl_fast.h — measured 0.8 % SLOWER here:
                // Penn94-shaped bf16 14.50 vs 14.38 ms, profiles/r7n_*; this spelling gives flat loads with 64-bit vector addresses)
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
            const float yy = ent_y[wave][idx & 63], wsg = ent_w[wave][idx & 63];
            const float ww = fabsf(wsg);                           // (sign: see the D = 64 kernel)
            const int qq = ent_q[wave][idx & 63];
            const float pr = p * (1.0f - p);
            const float gl = ww == 0.0f ? 0.0f : ww * (p - yy) * (pr >= 1e-12f ? 1.0f : pr * 1e12f);     // valid above G/2
            if (lane >= G / 2 && lane < G / 2 + U * DUPL && (lane & (DUPL - 1)) == 0 && si.beg + idx < si.end && !(__float_as_uint(wsg) >> 31)) prob_out[qq] = p;
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

template <int K, int D, typename T>
struct TrainOps {
    static constexpr int ROW = K * D;
    // training step of the scorer in one pass: prob, and dZ / dH for the weighted BCE of (y, w)
    static int score_train(const dl_pair_incidence* inc, const void* Z, const void* H, float t, const float* y,
                           const float* w, float* prob, float* dZ, float* dH, float* part, hipStream_t st) {
        const dl_csr_plan* g = &inc->csr;
        const float* no_x = nullptr;
        if constexpr (std::is_same<T, float>::value && TrainWave<K, D>::ok) {
            if (g->seg_len <= 64 && g->seg_len % TrainWave<K, D>::U == 0 && !config().train_group_kernel) {
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
            if (g->seg_len <= 64 && g->seg_len % U == 0 && !config().train_group_kernel) {
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

int fast_score_pairs_train(const dl_pair_incidence* inc, const void* Z, const void* H, int K, int d, int dtype, float t,
                           const float* y, const float* w, float* prob, float* dZ, float* dH, float* part, hipStream_t st) {
    if (inc->entry_yw != nullptr) {      // labels / weights per entry (dl_pair_incidence.entry_yw): the kernels take them through `y`, w = NULL
        y = inc->entry_yw;
        w = nullptr;
    }
#define X_F32(KK, DD) if (K == KK && d == DD) return fast::TrainOps<KK, DD, float>::score_train(inc, Z, H, t, y, w, prob, dZ, dH, part, st);
#define X_BF16(KK, DD) if (K == KK && d == DD) return fast::TrainOps<KK, DD, fast::bf16_t>::score_train(inc, Z, H, t, y, w, prob, dZ, dH, part, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

}  // namespace dl
