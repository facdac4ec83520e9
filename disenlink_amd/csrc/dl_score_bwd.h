// Scorer backward by recomputation, and (FUSED) the group-per-entry one-pass training scorer: shared by dl_score.hip
// (separate backward) and dl_train.hip (fallback of the one-pass scorer for shapes without a wave-per-entry kernel).
#pragma once
#include "dl_fast.h"

namespace dl {
namespace fast {

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
                if (w == nullptr) {                                 // y = per-entry (label, signed weight) pairs: dl_pair_incidence.entry_yw
                    const float2 yw = reinterpret_cast<const float2*>(y)[si.beg + lane];
                    my_y = yw.x;
                    my_w = yw.y;
                } else {
                    my_y = y[q];
                    my_w = w[q];
                }
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
                const float yy = __shfl(my_y, idx, DL_WAVE), wsg = __shfl(my_w, idx, DL_WAVE);  // w = 0 past the segment end
                const float ww = fabsf(wsg);                       // a negative sign (per-entry weights only) = the pair's other entry writes prob
                const int qq = __shfl(my_q, idx, DL_WAVE);
                // dl_pair_bce's gradient times the sigmoid backward: w (p - y) / max(r, 1e-12) * r with r = p (1 - p) — i.e.
                // w (p - y) itself unless r underflows the clamp (saturated scores: r = 0 gives exactly 0), without the division
                const float r = p * (1.0f - p);
                gl = ww == 0.0f ? 0.0f : ww * (p - yy) * (r >= 1e-12f ? 1.0f : r * 1e12f);
                if (base + grp < si.end && c == 0 && !(__float_as_uint(wsg) >> 31)) prob_out[qq] = p;
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

}  // namespace fast
}  // namespace dl
