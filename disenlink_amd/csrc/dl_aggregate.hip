// Aggregation with the neighbour's normaliser (model.py:73-76).
// (one of the tuned-kernel translation units; the shared pieces and the design notes are in dl_fast.h)
#include "dl_fast.h"

namespace dl {
namespace fast {

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
                                                              float* __restrict__ h_part, int stream_out, int sum_rows_here) {
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
        if (mine) my_s = s[(size_t)my_col * K + my_k];      // if the normalised weight a / s~ arrived with p / a in the per-entry stream
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
    if (!direct && sum_rows_here) {                                 // a row of several units: the last of them to get here writes H[row]
        publish_unit_and_sum_row<US::F4>(g, si.slot, h_part, ROW, r, lane, [&](int x, const float4& tot) {
            // the combine launch's arithmetic: 0 + beta z + (1 - beta) sum
            const float4 z = load4<T>(Z + (size_t)si.grow * ROW + 4 * x);
            store4(H + (size_t)si.grow * ROW + 4 * x, combine_finish(make_float4(0.f, 0.f, 0.f, 0.f), beta != 0.0f, beta, z, omb, tot, nullptr));
        });
        return;
    }
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

template <int K, int D, typename T>
struct AggOps {
    static constexpr int ROW = K * D;
    static int aggregate_fwd(const dl_csr_plan* g, const void* Z, float beta, const uint8_t* p, const float* a,
                             const float* s, void* H, float* h_part, hipStream_t st) {
        const bool here = sums_rows_in_launch(g);
        hipLaunchKernelGGL((aggregate_cls_kernel<K, D, T, 4>), dim3(seg_blocks(g)), dim3(BLOCK), 0, st, *g, (const T*)Z,
                           beta, p, a, s, (T*)H, h_part, stream_rows(g, ROW, sizeof(T)), here ? 1 : 0);
        if (g->n_multi > 0 && !here)
            hipLaunchKernelGGL((row_combine_kernel<ROW, T, T>), dim3(g->n_multi), dim3(BLOCK), 0, st, *g, h_part, ROW,
                               (const T*)Z, beta, 1.0f - beta, (T*)H, 0);
        return check_launch("aggregate_fwd(fast)");
    }
};

}  // namespace fast

int fast_aggregate_fwd(const dl_csr_plan* g, const void* Z, int K, int d, int dtype, float beta, const uint8_t* p,
                       const float* a, const float* s, void* H, float* h_part, hipStream_t st) {
#define X_F32(KK, DD) if (K == KK && d == DD) return fast::AggOps<KK, DD, float>::aggregate_fwd(g, Z, beta, p, a, s, H, h_part, st);
#define X_BF16(KK, DD) if (K == KK && d == DD) return fast::AggOps<KK, DD, fast::bf16_t>::aggregate_fwd(g, Z, beta, p, a, s, H, h_part, st);
    DL_DISPATCH(X)
#undef X_F32
#undef X_BF16
}

}  // namespace dl
