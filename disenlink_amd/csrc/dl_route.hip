// Routing: exp-Gram softmax over K, first-max arg-max per edge, row sums of the winning weights (model.py:56-72).
// (one of the tuned-kernel translation units; the shared pieces and the design notes are in dl_fast.h)
#include "dl_fast.h"

namespace dl {
namespace fast {

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
        const int j = entry_scalar<EPW>(my_col, base - si.beg, grp);
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
            const int r = entry_scalar<EPW>(my_rev, base - si.beg, grp);
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
                                                                int n_reg_blocks) {
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

static void launch_s_rowsum(const dl_csr_plan* g, int K, const uint8_t* p, const float* a, float* s, hipStream_t st) {
    if (g->n_seg <= 0) return;
    const int reg = (g->n_seg + ROWSUM_POS_PER_BLOCK - 1) / ROWSUM_POS_PER_BLOCK;
    const dim3 grid((unsigned)reg + wave_blocks(g->n_multi)), block(BLOCK);
    if (K <= 4) hipLaunchKernelGGL(s_rowsum_thread_kernel<4>, grid, block, 0, st, *g, K, p, a, s, reg);
    else if (K <= 8) hipLaunchKernelGGL(s_rowsum_thread_kernel<8>, grid, block, 0, st, *g, K, p, a, s, reg);
    else if (K <= 16) hipLaunchKernelGGL(s_rowsum_thread_kernel<16>, grid, block, 0, st, *g, K, p, a, s, reg);
    else hipLaunchKernelGGL(s_rowsum_thread_kernel<32>, grid, block, 0, st, *g, K, p, a, s, reg);   // tuned shapes: K <= 32
}

void launch_vec_combine(const dl_csr_plan* g, int K, const float* part, int mode, const float* s_raw,
                        float* out, hipStream_t st) {
    if (g->n_multi <= 0) return;
    hipLaunchKernelGGL(vec_combine_kernel, dim3(wave_blocks(g->n_multi)), dim3(BLOCK), 0, st, *g, K, pow2_at_least(K),
                       part, mode, s_raw, out);
}

template <int K, int D, typename T>
struct RouteOps {
    // `route`: the (possibly sliced / upper-triangle) plan the routing kernel walks, NULL = g itself;
    // mirror: it covers col >= row only and every result is also written through rev
    static int route_fwd(const dl_csr_plan* g, const dl_csr_plan* route, bool mirror, const int32_t* rev,
                         const void* Z, float t, uint8_t* p, float* a, float* s, float* s_part, hipStream_t st) {
        const dl_csr_plan* rp = route ? route : g;
        // DL_ROUTE_BALLOT=1: the ballot arg-max (A/B switch; see the kernel)
        const bool ballot = config().route_ballot;
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
        (void)s_part;
        launch_s_rowsum(g, K, p, a, s, st);
        return check_launch("route_fwd(fast)");
    }
};

}  // namespace fast

bool fast_supported(int K, int d, int dtype) {
#define X(KK, DD) if (K == KK && d == DD) return true;
    if (dtype == DL_F32) { DL_FAST_SHAPES_F32(X) }
    if (dtype == DL_BF16) { DL_FAST_SHAPES_BF16(X) }
#undef X
    return false;
}

int fast_route_fwd(const dl_csr_plan* g, const dl_csr_plan* route, bool mirror, const int32_t* rev, const void* Z,
                   int K, int d, int dtype, float t, uint8_t* p, float* a, float* s, float* s_part, hipStream_t st) {
#define X_F32(KK, DD) \
    if (K == KK && d == DD) return fast::RouteOps<KK, DD, float>::route_fwd(g, route, mirror, rev, Z, t, p, a, s, s_part, st);
#define X_BF16(KK, DD)      \
    if (K == KK && d == DD) \
        return fast::RouteOps<KK, DD, fast::bf16_t>::route_fwd(g, route, mirror, rev, Z, t, p, a, s, s_part, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

}  // namespace dl
