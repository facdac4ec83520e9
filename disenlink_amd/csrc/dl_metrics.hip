// Tie-averaged AUC (sklearn.roc_auc_score as used at main_disentangled.py:202-204, 217-219) from integer counts in
// ONE launch: for a FIXED label vector the positive and negative index sets are known, and
//     2 * U = sum over positives p of ( #{n: s_n < s_p} + #{n: s_n <= s_p} )
//           = sum over negatives n of ( #{p: s_p > s_n} + #{p: s_p >= s_n} ).
// Counts are additive over disjoint slices of either class: the smaller class is cut into slices of 1,024 scores; a
// workgroup (one score per thread) sorts ITS slice with a bitonic network — wave shuffles while the partner is inside
// the wavefront, LDS for the four outer strides — and then locates its chunk of the other class in the sorted slice
// with two 10-step binary searches per element.  grid = slices x chunks; no sort of a whole class, no inter-workgroup
// dependency, integer atomics (exact, order-independent).  ~10 us at 10^4 x 5*10^4 scores against ~130 us for a
// device sort of the negatives plus the searches.
// (Tried first: all n_pos * n_neg comparisons with one v_cmp per 64 pairs and s_bcnt1: 115 us — the scalar unit, one
// instruction per cycle per CU, is the bound; and a whole-class bitonic sort in LDS by every workgroup: 212 us.)
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include "dl_common.h"
#include "dl_kernels.h"

namespace dl {

constexpr int AUC_THREADS = 1024;                               // = slice length

// small_is_pos: 1 = the positives are the sliced (sorted) class and the negatives are searched; 0 = the other way round
__global__ __launch_bounds__(AUC_THREADS) void auc_slice_search_kernel(const float* __restrict__ score,
                                                                       const int64_t* __restrict__ small_idx, int n_small,
                                                                       const int64_t* __restrict__ large_idx, int n_large,
                                                                       int large_per_block, int small_is_pos,
                                                                       unsigned long long* __restrict__ u2) {
    __shared__ float a[AUC_THREADS];
    __shared__ unsigned long long red[AUC_THREADS / 64];
    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * AUC_THREADS;
    const int cnt = min(AUC_THREADS, n_small - s0);             // real scores in this slice (>= 1), the rest is +inf
    float v = tid < cnt ? score[small_idx[s0 + tid]] : INFINITY;
    for (int k = 2; k <= AUC_THREADS; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            float other;
            if (j >= 64) {                                      // partner in another wavefront: through LDS
                __syncthreads();
                a[tid] = v;
                __syncthreads();
                other = a[tid ^ j];
            } else {
                other = __shfl_xor(v, j, 64);
            }
            const bool up = (tid & k) == 0, lower = (tid & j) == 0;
            // the lower index of a pair keeps the minimum in an ascending run (the maximum in a descending one)
            v = (lower == up) ? fminf(v, other) : fmaxf(v, other);
        }
    __syncthreads();
    a[tid] = v;
    __syncthreads();
    const int g0 = blockIdx.y * large_per_block, g1 = min(n_large, g0 + large_per_block);
    unsigned long long mine = 0;
    for (int g = g0 + tid; g < g1; g += AUC_THREADS) {
        const float x = score[large_idx[g]];
        int lb = 0, ub = 0;                                     // first index with a[i] >= x / a[i] > x, within [0, cnt]
        for (int len = cnt; len > 0;) {
            const int half = len >> 1;
            if (a[lb + half] < x) { lb += half + 1; len -= half + 1; } else len = half;
        }
        for (int len = cnt; len > 0;) {
            const int half = len >> 1;
            if (a[ub + half] <= x) { ub += half + 1; len -= half + 1; } else len = half;
        }
        mine += small_is_pos ? (unsigned long long)(cnt - lb) + (unsigned long long)(cnt - ub)
                             : (unsigned long long)lb + (unsigned long long)ub;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = mine;
    __syncthreads();
    if (tid == 0) {
        unsigned long long tot = 0;
        for (int w = 0; w < AUC_THREADS / 64; ++w) tot += red[w];
        if (tot) atomicAdd(u2, tot);
    }
}

// work ~ slices x n_large x 20 LDS reads: fine up to a few 10^10 pairs (Penn94-sized validation sets ~60 us)
bool auc_counts_supported(int n_pos, int n_neg) {
    return (double)std::min(n_pos, n_neg) * (double)std::max(n_pos, n_neg) <= 4.0e11;
}

int auc_pair_counts(const float* score, const int64_t* pos_idx, int n_pos, const int64_t* neg_idx, int n_neg,
                    unsigned long long* u2, hipStream_t st, bool clear) {
    // clear = false: the counts are ADDED to *u2 (the caller keeps it at zero between evaluations: dl_epoch_finish reads
    // and clears it in its own launch — no memset node in the epoch)
    if (clear && hipMemsetAsync(u2, 0, sizeof(unsigned long long), st) != hipSuccess) return check_launch("auc_pair_counts(memset)");
    if (n_pos == 0 || n_neg == 0) return DL_OK;
    const bool small_is_pos = n_pos <= n_neg;
    const int n_small = small_is_pos ? n_pos : n_neg, n_large = small_is_pos ? n_neg : n_pos;
    const int slices = (n_small + AUC_THREADS - 1) / AUC_THREADS;
    // ~1,024 workgroups where the sizes allow; a chunk of the searched class is at least one pass of the workgroup
    int target = 1024;                                          // tools/auc_time.py: flat from 128 to 2,048 at 11 slices x 54k, 1,024+ best at 67 x 340k
    if (config().auc_target > 0) target = config().auc_target;   // DL_AUC_TARGET
    int chunks = std::max(1, std::min((n_large + AUC_THREADS - 1) / AUC_THREADS, (target + slices - 1) / slices));
    chunks = std::min(chunks, 65535);
    const int per = (n_large + chunks - 1) / chunks;
    chunks = (n_large + per - 1) / per;
    hipLaunchKernelGGL(auc_slice_search_kernel, dim3((unsigned)slices, (unsigned)chunks), dim3(AUC_THREADS), 0, st, score,
                       small_is_pos ? pos_idx : neg_idx, n_small, small_is_pos ? neg_idx : pos_idx, n_large, per,
                       small_is_pos ? 1 : 0, u2);
    return check_launch("auc_pair_counts");
}

}  // namespace dl

// ---------------------------------------------------------------------------- Adam (main_disentangled.py:150)
// torch.optim.Adam's update (weight decay added to the gradient, bias-corrected moments) over up to DL_ADAM_MAX_BUFS
// contiguous buffers in ONE launch, one float4 per thread: torch's fused implementation walks 65,536-element chunks with
// one 512-thread block each — 13 blocks for this model's 0.8M parameters, 44 us of pure latency per step.  The step
// counter lives on the device (no host sync, graph-capturable): a one-thread kernel increments it and leaves the bias
// corrections next to it, then every thread of the update reads those three floats.
//   state[0] = step (as float), state[1] = lr / (1 - beta1^step), state[2] = sqrt(1 - beta2^step)
namespace dl {

struct AdamBufs {
    float* p[DL_ADAM_MAX_BUFS];
    const float* g[DL_ADAM_MAX_BUFS];
    float* m[DL_ADAM_MAX_BUFS];
    float* v[DL_ADAM_MAX_BUFS];
    unsigned long long n4_end[DL_ADAM_MAX_BUFS];               // running end of each buffer in float4 units (padded up)
    unsigned long long n[DL_ADAM_MAX_BUFS];                    // elements of each buffer
    int count;
};

__global__ void adam_advance_kernel(float* __restrict__ state, double lr, double beta1, double beta2) {
    const float step = state[0] + 1.0f;
    state[0] = step;
    state[1] = (float)(lr / (1.0 - pow(beta1, (double)step)));               // step size lr / (1 - beta1^step)
    state[2] = (float)sqrt(1.0 - pow(beta2, (double)step));
}

// w1 = 1 - beta1, w2 = 1 - beta2 (formed in double on the host, like torch's kernel arguments)
// HOST_STEP (dl_adam_step_at): the caller counts the steps — an eager loop knows the number — so there is no one-thread
// launch in front: thread 0 of every workgroup forms the two bias corrections from `step` with the SAME device
// expressions as adam_advance_kernel (the same bits) and hands them over through LDS; workgroup 0 leaves them in `state`.
template <bool HOST_STEP>
__global__ __launch_bounds__(256) void adam_update_kernel(AdamBufs b, float* __restrict__ state, float w1, float beta2,
                                                          float w2, float eps, float weight_decay, float step, double lr,
                                                          double beta1d, double beta2d) {
    __shared__ float corr[2];
    if constexpr (HOST_STEP) {
        if (threadIdx.x == 0) {
            corr[0] = (float)(lr / (1.0 - pow(beta1d, (double)step)));
            corr[1] = (float)sqrt(1.0 - pow(beta2d, (double)step));
            if (blockIdx.x == 0 && state != nullptr) { state[0] = step; state[1] = corr[0]; state[2] = corr[1]; }
        }
        __syncthreads();
    }
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    int i = 0;
    while (i < b.count && q >= b.n4_end[i]) ++i;
    if (i >= b.count) return;
    const unsigned long long e0 = (q - (i ? b.n4_end[i - 1] : 0ull)) * 4;
    const float step_size = HOST_STEP ? corr[0] : state[1], bc2s = HOST_STEP ? corr[1] : state[2];
    float* __restrict__ p = b.p[i];
    const float* __restrict__ g = b.g[i];
    float* __restrict__ m = b.m[i];
    float* __restrict__ v = b.v[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned long long e = e0 + j;
        if (e < b.n[i]) {
            const float pe = p[e];
            float ge = g[e];
            if (weight_decay != 0.0f) ge = fmaf(pe, weight_decay, ge);
            const float me = fmaf(ge - m[e], w1, m[e]);                              // lerp(m, g, 1 - beta1)
            const float ve = beta2 * v[e] + w2 * ge * ge;
            m[e] = me;
            v[e] = ve;
            const float denom = sqrtf(ve) / bc2s + eps;
            p[e] = pe - step_size * me / denom;
        }
    }
}

int adam_step(int n_bufs, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
              const size_t* numel, float* state, double lr, double beta1, double beta2, double eps, double weight_decay,
              hipStream_t st, long long host_step) {
    AdamBufs b;
    unsigned long long run = 0;
    for (int i = 0; i < n_bufs; ++i) {
        b.p[i] = params[i]; b.g[i] = grads[i]; b.m[i] = exp_avg[i]; b.v[i] = exp_avg_sq[i];
        b.n[i] = numel[i];
        run += (numel[i] + 3) / 4;
        b.n4_end[i] = run;
    }
    b.count = n_bufs;
    const dim3 grid((unsigned)std::max<unsigned long long>(1, (run + 255) / 256));
    if (host_step > 0) {                                        // the caller counts: one launch
        hipLaunchKernelGGL(adam_update_kernel<true>, grid, dim3(256), 0, st, b, state, (float)(1.0 - beta1), (float)beta2,
                           (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)host_step, lr, beta1, beta2);
        return check_launch("adam_step_at");
    }
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, st, state, lr, beta1, beta2);
    if (run > 0)
        hipLaunchKernelGGL(adam_update_kernel<false>, grid, dim3(256), 0, st, b, state, (float)(1.0 - beta1), (float)beta2,
                           (float)(1.0 - beta2), (float)eps, (float)weight_decay, 0.0f, lr, beta1, beta2);
    return check_launch("adam_step");
}

}  // namespace dl

// ---------------------------------------------------------------------------- end of an epoch (main_disentangled.py:199-214)
// The reference reads the loss and the validation AUC on the host every epoch, compares the AUC with the best one so far,
// deep-copies the state_dict when it improved and counts the epochs since (patience).  Done that way the GPU idles from
// the read-back until the host has launched the next epoch's first kernel (74 us of a 940 us epoch on the squirrel-shaped
// graph, profiles/r5z_epoch_sequence.txt) and torch's scalar glue adds five launches.  Here the bookkeeping lives on the
// device, in ONE launch at the end of the epoch:
//   auc = u2 / denom2 (double, correctly rounded: the same value the host formed);  improved = !stopped && auc > best_auc
//   improved: the parameter buffers are copied to the best-weights buffers (state AFTER the step, like :209), stale = 0;
//   else stale += 1;  stale > patience: stopped = 1 — from then on the launch changes nothing (the host, which reads the
//   history one epoch behind, may have queued an epoch or two more: they must not touch the best weights);
//   hist[epoch] = (loss, auc);  *u2 = 0 for the next evaluation.
// Every workgroup takes its decision from the state as the previous launch left it; the LAST workgroup to finish (a
// counter in the state) writes the new state, so no workgroup can read a half-updated one.
namespace dl {

struct SnapBufs {
    const float* src[DL_ADAM_MAX_BUFS];
    float* dst[DL_ADAM_MAX_BUFS];
    unsigned long long n4_end[DL_ADAM_MAX_BUFS];
    unsigned long long n[DL_ADAM_MAX_BUFS];
    int count;
};

struct EpochState {                 // dl_epoch_state_bytes() = sizeof; zero-initialised by the caller
    double best_auc;
    long long stale, epoch, stopped, best_epoch;
    unsigned int blocks_done, pad;
};
static_assert(sizeof(EpochState) == 48, "layout documented in include/disenlink_hip.h");

__global__ __launch_bounds__(256) void epoch_finish_kernel(SnapBufs b, const float* __restrict__ loss, unsigned long long* u2,
                                                           double denom2, EpochState* st, double* __restrict__ hist,
                                                           long long max_epochs, long long patience, double* host_ring,
                                                           int ring) {
    const unsigned long long cnt = *reinterpret_cast<volatile unsigned long long*>(u2);
    const double auc = denom2 > 0.0 ? (double)cnt / denom2 : (double)NAN;
    // (epochs past max_epochs — the tail of a replayed graph that holds several epochs — count as stopped too)
    const bool stopped = *reinterpret_cast<volatile long long*>(&st->stopped) != 0 ||
                         *reinterpret_cast<volatile long long*>(&st->epoch) >= max_epochs;
    const bool improved = !stopped && auc > *reinterpret_cast<volatile double*>(&st->best_auc);
    if (improved) {
        // grid-stride over the float4 slots of all buffers: FEW workgroups (the end-of-kernel counter below is one atomic
        // per workgroup on one address — 768 of them cost 15 us, more than the 3 MB copy itself)
        const unsigned long long total = b.count ? b.n4_end[b.count - 1] : 0ull;
        for (unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (unsigned long long)gridDim.x * 256) {
            int i = 0;
            while (q >= b.n4_end[i]) ++i;
            const unsigned long long e0 = (q - (i ? b.n4_end[i - 1] : 0ull)) * 4;
            const float* __restrict__ s = b.src[i];
            float* __restrict__ d = b.dst[i];
            if (e0 + 4 <= b.n[i] && ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
                *reinterpret_cast<float4*>(d + e0) = *reinterpret_cast<const float4*>(s + e0);
            } else {
                for (unsigned long long e = e0; e < e0 + 4 && e < b.n[i]; ++e) d[e] = s[e];
            }
        }
    }
    __syncthreads();                                            // every read of the old state by this workgroup is done
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&st->blocks_done, 1u) == gridDim.x - 1) { // the last one: nobody reads the state any more
            __threadfence();
            st->blocks_done = 0;
            if (!stopped) {
                const long long e = st->epoch;
                hist[2 * e] = (double)loss[0];
                hist[2 * e + 1] = auc;
                if (host_ring != nullptr) {                     // pinned host memory: the host reads it after an event, no copy
                    volatile double* slot = host_ring + 4 * (e % ring);
                    slot[0] = (double)loss[0];
                    slot[1] = auc;
                    slot[2] = (double)(e + 1);                  // which epoch the slot holds (+1: 0 = never written)
                    __threadfence_system();
                }
                if (improved) {
                    st->best_auc = auc;
                    st->stale = 0;
                    st->best_epoch = e;
                } else {
                    st->stale += 1;
                }
                if (st->stale > patience) st->stopped = 1;
                st->epoch = e + 1;
            }
            *u2 = 0;
        }
    }
}

size_t epoch_state_bytes() { return sizeof(EpochState); }

int epoch_finish(int n_bufs, const float* const* params, float* const* best, const size_t* numel, const float* loss,
                 unsigned long long* u2, double denom2, void* state, double* hist, long long max_epochs, long long patience,
                 double* host_ring, int ring, hipStream_t st) {
    SnapBufs b;
    unsigned long long run = 0;
    for (int i = 0; i < n_bufs; ++i) {
        b.src[i] = params[i]; b.dst[i] = best[i];
        b.n[i] = numel[i];
        run += (numel[i] + 3) / 4;
        b.n4_end[i] = run;
    }
    b.count = n_bufs;
    const unsigned blocks = (unsigned)std::min<unsigned long long>(64, std::max<unsigned long long>(1, (run + 255) / 256));
    hipLaunchKernelGGL(epoch_finish_kernel, dim3(blocks), dim3(256), 0, st, b, loss, u2, denom2,
                       reinterpret_cast<EpochState*>(state), hist, max_epochs, patience, host_ring, ring);
    return check_launch("epoch_finish");
}

}  // namespace dl
