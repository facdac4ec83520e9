// Factor projection Z[n][k][:] = MLP_k(x[n]) on the matrix cores (model.py:13-15, 24-27, 106) — the
// only dense contraction of the path.  fp32 in / fp32 results: both layers as six exact bf16 products per term
// from three bf16 planes per operand (dl_tiles.h; fp32-grade accuracy), or — without a workspace or with
// DL_PROJECT_FP32_MFMA=1 — v_mfma_f32_32x32x2_f32, which is bit-for-bit a k-ordered fmaf chain; either way
// the result keeps the reference's fp32 semantics up to rounding / summation order.
//
// Two-layer form (Factor2): one workgroup = 8 waves = 128 nodes x ONE factor k, looping over (a group
// of) 128-unit chunks of the hidden layer.  Waves are 4 x 2: node quarter wn, hidden half wh; each wave
// owns a (64 hidden) x (32 nodes) tile = two independent accumulators; two waves share a SIMD, so one
// wave's LDS reads and barrier waits hide behind the other's MFMA chain.
//   layer 1 (transposed):  hidT[hidden][node] = W1_k[hidden][F] . x^T            A = W1 rows, B = x rows
//   bias + ReLU in the accumulator registers
//   layer 2:               Z^T[d][node] += W2_k[d][hidden] . hidT                 B = the accumulator itself
// A 32x32 accumulator has its column on the lane and its rows in the 16 registers, which is exactly
// the B-operand shape of the next MFMA when that product sums over the accumulator's ROW index
// (register r supplies the k-pair {acc_row(r,0), acc_row(r,1)}); so the hidden activations never leave
// the register file — no [N, K*nhid] tensor is written to HBM and re-read, unlike two library GEMMs.
// The layer-1 operand tiles ([128 rows][32 features]: three bf16 planes at an 80-byte row pitch, or fp32 at
// 36 / 68 floats: aligned, conflict-free b128 reads) are double-buffered in LDS and the next step's tiles are fetched into registers behind the
// current step's MFMAs; the W2 operand (A[i = d][k = hidden], 4 consecutive hidden units per register
// quad) is read straight from global memory, one d-tile ahead of its use.  The two hidden halves of a
// node's Z are added through LDS at the end (fixed order).
// Small graphs: the hidden chunks are split over G workgroups per (node tile, k) whose partial Z go to
// G slabs, added in slab order by z_slab_sum_kernel (no atomics; needs the workspace).
#include <algorithm>
#include <cstdlib>
#include "dl_common.h"
#include "dl_kernels.h"
#include "dl_tiles.h"

namespace dl {
namespace project {

constexpr int TN = 128;       // nodes per workgroup (4 wave quarters x 32)
constexpr int TH = 128;       // hidden units per chunk (2 wave halves x 64)
// feature chunk staged per step: 64 (one barrier per 64 MFMAs of a wave) where the registers allow, else 32;
// LDS row pitch of the operand tiles = FC + 4 floats
constexpr int fwd_fc(int D) { return D <= 64 ? 64 : 32; }
constexpr int NTHR = 512;

// stage rows [row0, row0+rows) x cols [col0, col0+cols) of a row-major [n_rows][n_cols] matrix into
// LDS with leading dimension ld (zero fill outside the matrix)
__device__ __forceinline__ void stage_tile(float* lds, int ld, const float* __restrict__ src, int n_rows, int n_cols,
                                           int row0, int col0, int rows, int cols) {
    for (int i = threadIdx.x; i < rows * cols; i += blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        const int gr = row0 + r, gc = col0 + c;
        lds[r * ld + c] = (gr < n_rows && gc < n_cols) ? src[(size_t)gr * n_cols + gc] : 0.0f;
    }
}

// 4 consecutive floats src[base .. base+3] of a vector of length n, loaded unconditionally from a clamped
// address; zero4() afterwards clears what lies outside the vector.  Two steps so that the s_waitcnt of the
// load lands at the use, not at the issue.  VEC: n % 4 == 0 and base % 4 == 0 (one aligned dwordx4).
template <bool VEC>
__device__ __forceinline__ float4 load4_raw(const float* __restrict__ src, int base, int n) {
    if constexpr (VEC) {
        return *reinterpret_cast<const float4*>(src + (base < n ? base : 0));
    } else {
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = src[base + i < n ? base + i : 0];
        return make_float4(e[0], e[1], e[2], e[3]);
    }
}
template <bool VEC>
__device__ __forceinline__ float4 zero4(const float4& q, int base, int n) {
    if constexpr (VEC) {
        const unsigned m = base < n ? 0xFFFFFFFFu : 0u;
        return make_float4(mask_bits(q.x, m), mask_bits(q.y, m), mask_bits(q.z, m), mask_bits(q.w, m));
    } else {
        return make_float4(mask_bits(q.x, base + 0 < n ? 0xFFFFFFFFu : 0u), mask_bits(q.y, base + 1 < n ? 0xFFFFFFFFu : 0u),
                           mask_bits(q.z, base + 2 < n ? 0xFFFFFFFFu : 0u), mask_bits(q.w, base + 3 < n ? 0xFFFFFFFFu : 0u));
    }
}

// The W2 operand of layer 2 for one hidden chunk, staged through LDS (round 5).  The planes of W2 lie in global memory as
// [3][K*D][nhid_p] bf16 with the hidden units in layer 2's k-slot order; a chunk's slice — D rows x 128 hidden x 3 planes —
// was read straight from there by every wave, one K = 16 group ahead of its use: 16 bytes per lane from 32 different rows
// per instruction (a quarter of every 128-byte line used), and a global round trip per group that no prefetch depth hid
// (stamps, tools/proj_stamps.py: 9,000 cycles per chunk for 48 MFMAs = 1,536 cycles of issue).  Now the whole workgroup
// copies the slice with fully coalesced 16-byte loads (issued at the top of the chunk's last layer-1 step) into the LDS tile
// buffer that step has just finished reading, and layer 2 takes its A operands from there with ds_read_b128.
template <int D, int THREADS>
struct W2Stage {
    static constexpr int PIECES = 3 * D * 16;                  // 16-byte pieces: 3 planes x D rows x (128 hidden / 8)
    static constexpr int PER = PIECES / THREADS;
    static constexpr int PITCH = 128 + 8;                      // LDS row pitch (bf16): 68 dwords -> conflict-free b128 reads
    static_assert(PIECES % THREADS == 0, "the slice divides over the workgroup");
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v[PER];
    __device__ __forceinline__ void fetch(const __bf16* __restrict__ w2, size_t plane_stride, size_t row0, int nhid_p, int col0, int tid) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int q = tid + THREADS * j, p = q / (D * 16), r = (q % (D * 16)) / 16, c = (q % 16) * 8;
            v[j] = *reinterpret_cast<const u32x4*>(w2 + p * plane_stride + (row0 + r) * nhid_p + col0 + c);
        }
    }
    __device__ __forceinline__ void stash(__bf16* lds, int tid) const {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int q = tid + THREADS * j, p = q / (D * 16), r = (q % (D * 16)) / 16, c = (q % 16) * 8;
            *reinterpret_cast<u32x4*>(lds + (p * D + r) * PITCH + c) = v[j];
        }
    }
};
#ifndef DL_PROJ_STAGGER
#define DL_PROJ_STAGGER 0         // 1: waves 4..7 of the forward stage before their first MFMA block — measured SLOWER (same box: 90.7 ->
                                  // 95.4 us at the bench shape, 675 -> 748 at F = 2,088, 845 -> 868 at d = 128): the staging under a wave-uniform
                                  // branch costs more than the overlap buys; kept for the record
#endif
#ifndef DL_W2_LDS
#define DL_W2_LDS 1               // -DDL_W2_LDS=0: the W2 operand straight from global memory (the round-2 form), for A/B runs
#endif

// Layer-1 tiles by LDS-DMA (round 6; -DDL_PROJ_DMA=0: the register-staged form, for A/B runs).  A [3][128][32] bf16 plane tile
// (24 KB, contiguous in the plane array) goes to its padded LDS image — 384 rows of 80 bytes: 4 data pieces + 1 pad piece of
// 16 bytes — by 30 global_load_lds_dwordx4 wave-instructions: the LDS side of such an instruction is linear (wave-uniform
// base + 16 bytes per lane), the GLOBAL address is per lane, so lane l of instruction i fetches the piece that belongs at
// padded position i * 64 + l (a pad position fetches its row's last piece again: a fifth of the requests buy nothing, all of
// them L2 hits).  No staging registers (the register form held a whole tile pair: 30 registers), no ds_write pass, and no
// vmcnt(0) in the middle of a step waiting for the tile loads: stamps of the register form showed 1,000-2,200 of a step's
// 3,800 cycles in "stash" + "fetch issue".  Wave w issues instructions w, w + 8, w + 16, w + 24 of each tile.
#ifndef DL_PROJ_DMA
#define DL_PROJ_DMA 1
#endif
struct PlaneDma {
    static constexpr int PIECES = 3 * PLANE_ROWS * 5;           // padded 16-byte positions of a tile
    static constexpr int INSTR = PIECES / DL_WAVE;               // 30
    static constexpr int PER_WAVE = (INSTR + NTHR / DL_WAVE - 1) / (NTHR / DL_WAVE);
    static_assert(PIECES % DL_WAVE == 0 && SPLIT_COLS == 32 && SPLIT_PITCH == 40, "80-byte LDS rows of 64-byte tile rows");
    unsigned off[PER_WAVE];                                     // byte offset of this lane's piece inside the global tile
    __device__ __forceinline__ void init(int wave, int lane) {
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const unsigned pos = (unsigned)((wave + (NTHR / DL_WAVE) * j) * DL_WAVE + lane), row = pos / 5u, slot = pos - row * 5u;
            off[j] = (row * 4u + (slot < 4u ? slot : 3u)) * 16u;
        }
    }
    __device__ __forceinline__ void issue(const __bf16* __restrict__ tile, __bf16* lds_tile, int wave) const {
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const int i = wave + (NTHR / DL_WAVE) * j;           // wave-uniform
            if (i < INSTR)
                __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(tile) + off[j]),
                                                 (lptr_t)(reinterpret_cast<char*>(lds_tile) + i * (DL_WAVE * 16)), 16, 0, 0);
        }
    }
};

// Two-layer projection.  W1 [K][nhid][F], b1 [K][nhid], W2 [K][D][nhid], b2 [K][D].
// VEC: F % 4 == 0 and nhid % 4 == 0.  1-D grid of xcd_grid(node tiles of 128, K * G hidden-chunk groups).
// out: Z [N][K][D] with b2 != nullptr (G == 1), or slab [G][N][K][D] of partial sums with b2 == nullptr.
// SPLIT: layer 1 on the bf16 matrix path from three bf16 planes per operand (dl_tiles.h: fp32-grade accuracy at a
// multiple of the fp32 MFMA rate); x and W1 arrive as padded plane arrays (dl_planes.hip) and the [128][32] tiles are
// copied without masks.  VEC then only says nhid % 4 == 0 (W2 / b1 quads).
struct FwdPlanes {
    const __bf16* x; const __bf16* w; size_t w_batch; int ncb;     // tile-major planes of x and of the K matrices W1_k
    const __bf16* w2; size_t w2_ps; int nhid_p;                    // planes [3][K*D][nhid_p] of W2 in layer 2's k-slot order
};

// -DDL_PROJ_STAMPS=<workgroup index>: DIAGNOSTIC build — waves 0 and 4 of that workgroup (two waves of one SIMD) record
// s_memtime at the phase boundaries of every pipeline step into dl_proj_stamps (read back by dl_debug_read_stamps; the
// stamps go nowhere else).  tools/proj_stamps.py prints the timeline.
#ifdef DL_PROJ_STAMPS
__device__ unsigned long long dl_proj_stamps[2][512];
#define DL_STAMP(code)                                                                                  \
    do {                                                                                                \
        if (stamp_on && stamp_n < 510) {                                                                \
            dl_proj_stamps[stamp_w][stamp_n++] = ((unsigned long long)(code) << 56) | (__builtin_amdgcn_s_memtime() & 0x00FFFFFFFFFFFFFFull); \
            dl_proj_stamps[stamp_w][511] = stamp_n;                                                     \
        }                                                                                               \
    } while (0)
#else
#define DL_STAMP(code) do {} while (0)
#endif

template <int D, bool VEC, bool SPLIT>
__global__ __launch_bounds__(NTHR) void project2_fwd_kernel(const float* __restrict__ x, int N, int F, int nhid,
                                                            const float* __restrict__ W1, const float* __restrict__ b1,
                                                            const float* __restrict__ W2, const float* __restrict__ b2,
                                                            float* __restrict__ out, int K, int G, int chunks_per_group,
                                                            float* __restrict__ hid_out, int ldh, FwdPlanes P) {
    constexpr int DT = D / 32;
    constexpr int FC = SPLIT ? SPLIT_COLS : fwd_fc(D), LDT = FC + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                               // [2][TN][LDT]
    float* w1s = xs + 2 * TN * LDT;                // [2][TH][LDT]
    // SPLIT: two parity buffers, each [3][TN][SPLIT_PITCH] (x tile) followed by [3][TH][SPLIT_PITCH] (W1 tile): one buffer is
    // one contiguous block (it also takes the W2 slice of a chunk's layer 2, W2Stage)
    constexpr int XBUF = 3 * TN * SPLIT_PITCH, PBUF = 3 * (TN + TH) * SPLIT_PITCH;
    __bf16* pbuf = reinterpret_cast<__bf16*>(lds);
    constexpr bool W2LDS = SPLIT && DL_W2_LDS && D <= 64 && (size_t)W2Stage<D, NTHR>::PITCH * 3 * D <= (size_t)PBUF;
    float* bias_s = reinterpret_cast<float*>(pbuf + 2 * PBUF);      // SPLIT: the 128 biases of the current hidden chunk
    float bias_q = 0.0f;
    const XcdItem item = xcd_item(blockIdx.x, (N + TN - 1) / TN, K * G);
    if (!item.valid) return;
    const int k = item.b % K, grp = item.b / K;
    const int n0 = item.a * TN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, half = lane >> 5;
    const int wn = wave >> 1, wh = wave & 1;
#ifdef DL_PROJ_STAMPS
    const bool stamp_on = (int)blockIdx.x == DL_PROJ_STAMPS && (wave == 0 || wave == 4) && lane == 0;
    const int stamp_w = wave >> 2;
    int stamp_n = 0;
#endif
    DL_STAMP(1);
    const float* W1k = W1 + (size_t)k * nhid * F;
    const float* W2k = W2 + (size_t)k * D * nhid;
    const float* b1k = b1 + (size_t)k * nhid;
    const int nfc = SPLIT ? P.ncb : (F + FC - 1) / FC, nhc = (nhid + TH - 1) / TH;
    const int hc0 = grp * chunks_per_group;
    const int steps = nfc * max(0, min(chunks_per_group, nhc - hc0));

    TileStage<TN, FC, VEC, NTHR> xt;
    TileStage<TH, FC, VEC, NTHR> wt;
    static_assert(TN == PLANE_ROWS && TH == PLANE_ROWS, "tiles of the plane arrays");
    constexpr bool DMA = SPLIT && DL_PROJ_DMA;
    PlaneStage<NTHR, DMA ? 8 * NTHR / PLANE_ROWS : SPLIT_COLS> xq, wq;   // (DMA: unused — the smallest instantiation)
    PlaneDma dma;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    if constexpr (DMA) dma.init(wave_s, lane);
    // DMA: both tiles of step s straight into its parity buffer (which every wave left at the barrier before)
    auto dma_tile = [&](int s) {
        const int hc = hc0 + s / nfc, fc = s % nfc;
        dma.issue(P.x + plane_tile<SPLIT_COLS>(item.a, fc, P.ncb), pbuf + (s & 1) * PBUF, wave_s);
        dma.issue(P.w + (size_t)k * P.w_batch + plane_tile<SPLIT_COLS>(hc, fc, P.ncb), pbuf + (s & 1) * PBUF + XBUF, wave_s);
    };
    auto fetch = [&](int s) {
        const int hc = hc0 + s / nfc, fc = s % nfc;
        if constexpr (DMA) {
            (void)hc; (void)fc;
        } else if constexpr (SPLIT) {
            xq.fetch(P.x + plane_tile<SPLIT_COLS>(item.a, fc, P.ncb), tid);
            wq.fetch(P.w + (size_t)k * P.w_batch + plane_tile<SPLIT_COLS>(hc, fc, P.ncb), tid);
        } else {
            xt.fetch(x + (size_t)n0 * F + fc * FC, F, N - n0, F - fc * FC, tid);
            wt.fetch(W1k + (size_t)hc * TH * F + fc * FC, F, nhid - hc * TH, F - fc * FC, tid);
        }
    };
    auto stash = [&](int s) {
        if constexpr (DMA) {
            (void)s;
        } else if constexpr (SPLIT) {
            xq.stash(pbuf + (s & 1) * PBUF, tid);
            wq.stash(pbuf + (s & 1) * PBUF + XBUF, tid);
        } else {
            xt.template stash<LDT>(xs + (s & 1) * TN * LDT, tid);
            wt.template stash<LDT>(w1s + (s & 1) * TH * LDT, tid);
        }
    };
    // W2_k[dt*32 + li][hidden quad g of tile ht]: the A operand of layer 2 for d-tile dt
    auto load_w2 = [&](float4 (&wv)[2][4], int dt, int hbase) {
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                wv[ht][g] = load4_raw<VEC>(W2k + (size_t)(dt * 32 + li) * nhid, hbase + ht * 32 + 8 * g + 4 * half, nhid);
    };

    f32x16 hacc[2], zacc[DT];
#pragma unroll
    for (int ht = 0; ht < 2; ++ht) zero_acc(hacc[ht]);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) zero_acc(zacc[dt]);

    // Software pipeline (one register set per tile, write-after-barrier): during step s the registers hold
    // tile s+1, whose global loads were issued one step earlier; it goes to the other LDS buffer in the shadow
    // of the first MFMA block of step s (every wave left that buffer at the barrier before), and the loads of
    // tile s+2 are issued right behind it.
    if (steps > 0) {
        if constexpr (DMA) {
            dma_tile(0);                                          // tiles 0 and 1 on their way into the two buffers together
            if (steps > 1) dma_tile(1);
        } else if constexpr (SPLIT) {
            // tiles 0 and 1 in flight together (a second register set while no accumulator is live yet): one global
            // round trip in the prologue instead of two — a workgroup runs only 8 steps at F = 128
            PlaneStage<NTHR, SPLIT_COLS> x0, w0;
            x0.fetch(P.x + plane_tile<SPLIT_COLS>(item.a, 0, P.ncb), tid);
            w0.fetch(P.w + (size_t)k * P.w_batch + plane_tile<SPLIT_COLS>(hc0, 0, P.ncb), tid);
            fetch(min(1, steps - 1));
            x0.stash(pbuf, tid);
            w0.stash(pbuf + XBUF, tid);
        } else {
            fetch(0);
            stash(0);
            if (steps > 1) fetch(1);
        }
    }
    DL_STAMP(2);
    if constexpr (DMA) wait_vmem();
    __syncthreads();
    DL_STAMP(3);
    for (int s = 0; s < steps; ++s) {
        const int hc = hc0 + s / nfc, fc = s % nfc;
        const bool last = fc == nfc - 1;
        const int hbase = hc * TH + wh * 64;                    // first hidden unit of this wave's tile
        DL_STAMP(10);
        if constexpr (DMA) {
            // tile s + 1 into the other buffer — free since the barrier that ended step s - 1 — with the whole step to land
            // (tile 1 left in the prologue); the barrier at the end of this step waits for it (vmcnt(0) + s_barrier)
            if (s >= 1 && s + 1 < steps) dma_tile(s + 1);
        }
        float4 bias[2][4], wnext[2][4];                         // quad g of tile ht = hidden rows 8g+4*half .. +3
        if constexpr (!SPLIT) {
            if (last) {
#pragma unroll
                for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                    for (int g = 0; g < 4; ++g) bias[ht][g] = load4_raw<VEC>(b1k, hbase + ht * 32 + 8 * g + 4 * half, nhid);
                load_w2(wnext, 0, hbase);
            }
        }
        // SPLIT: nothing is requested from global memory at the top of a step.  s_waitcnt vmcnt counts in issue order, and the
        // tile of step s+1 — requested a step ago — is written to LDS in the middle of this step: a load issued HERE would be
        // younger than the tile's, the compiler cannot tell the two apart at the join of the `last` branch (it waits for
        // vmcnt(0)), and the stash would sit out a whole global round trip (stamps: +2,800 cycles in every chunk's last
        // step).  The chunk's biases and the W2 slice are requested BEHIND the stash instead (below); the biases — asked
        // for a step ahead — go to LDS here, at the top of the chunk's last step, where everything younger than them was
        // requested a step ago too.  (F <= 32, one step per chunk: requested here, staged before their use.)
        if constexpr (SPLIT) {
            if (tid < TH) {
                if (nfc == 1) {
                    const int h = hc * TH + tid;
                    bias_q = b1k[h < nhid ? h : 0];
                    bias_q = h < nhid ? bias_q : 0.0f;
                } else if (last) {
                    bias_s[tid] = bias_q;
                }
            }
        }
        // SPLIT: the A operand of layer 2 for group grp = (d-tile, hidden tile, K = 16 block): one 16-byte load per plane
        auto load_w2p = [&](bf16x8 (&a)[3], int grp) {
            const int dt = grp >> 2, ht = (grp >> 1) & 1, b = grp & 1;
            if constexpr (W2LDS) {                               // the slice staged in this step's (already read) tile buffer
                const __bf16* src = pbuf + (s & 1) * PBUF + (dt * 32 + li) * W2Stage<D, NTHR>::PITCH + wh * 64 + ht * 32 + 16 * b + 8 * half;
#pragma unroll
                for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(src + p * D * W2Stage<D, NTHR>::PITCH);
            } else {
                const __bf16* src = P.w2 + ((size_t)k * D + dt * 32 + li) * P.nhid_p + hbase + ht * 32 + 16 * b + 8 * half;
#pragma unroll
                for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(src + p * P.w2_ps);
            }
        };
        // one group in flight (issued here for the first, a layer-1 step ahead; three in flight measured no faster)
#ifndef DL_W2PF
#define DL_W2PF 1
#endif
        constexpr int W2PF = DL_W2PF;
        bf16x8 w2a[W2PF + 1][3];
        W2Stage<D, NTHR> w2st;
        if constexpr (SPLIT) {
            // lane half h supplies features 8h .. 8h+7 of each 16-wide block: A = W1 rows (two hidden tiles), B = x rows
            const __bf16* xb = pbuf + (s & 1) * PBUF + (wn * 32 + li) * SPLIT_PITCH + half * 8;
            const __bf16* wb = pbuf + (s & 1) * PBUF + XBUF + (wh * 64 + li) * SPLIT_PITCH + half * 8;
            // both K = 16 blocks of the chunk are read up front: block 1's LDS latency hides behind block 0's MFMAs
            bf16x8 a0[2][3], a1[2][3], b[2][3];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    b[kb][p] = *reinterpret_cast<const bf16x8*>(xb + p * TN * SPLIT_PITCH + kb * 16);
                    a0[kb][p] = *reinterpret_cast<const bf16x8*>(wb + p * TH * SPLIT_PITCH + kb * 16);
                    a1[kb][p] = *reinterpret_cast<const bf16x8*>(wb + (p * TH + 32) * SPLIT_PITCH + kb * 16);
                }
            // the step's staging work: tile s+1 to LDS, then the requests that must come behind that stash
            auto stage_work = [&]() {
                if (s + 1 < steps) stash(s + 1);
                // behind the stash (see the top of the step): the chunk's 128 biases — one float per thread of the first two
                // waves, a step ahead of their use — and, in the chunk's last step, the W2 operand of layer 2
                if (nfc >= 2 && fc == nfc - 2 && tid < TH) {
                    const int h = hc * TH + tid;
                    bias_q = b1k[h < nhid ? h : 0];
                    bias_q = h < nhid ? bias_q : 0.0f;
                }
                if (last) {
                    if constexpr (W2LDS) {
                        w2st.fetch(P.w2, P.w2_ps, (size_t)k * D, P.nhid_p, hc * TH, tid);  // coalesced; lands under the rest of the step
                    } else {
#pragma unroll
                        for (int q = 0; q < W2PF; ++q) load_w2p(w2a[q], q);
                    }
                }
                // unconditional (the last steps fetch the last tile again): a fetch under a condition makes the
                // registers a merge of old and new values, and hipcc then waits for the loads right here to copy them
                DL_STAMP(13);
                fetch(min(s + 2, steps - 1));
                DL_STAMP(14);
            };
#if DL_PROJ_STAGGER
            // The two waves of a SIMD (w and w + 4) run the same program between the same barriers: in lockstep both stage
            // at the same time and the matrix pipe idles meanwhile.  Waves 4..7 stage BEFORE their first MFMA block, waves
            // 0..3 behind it: one wave's MFMAs run beside its SIMD partner's LDS stores and global requests.
            const bool early = __builtin_amdgcn_readfirstlane(wave) >= 4;
            if (early) stage_work();
#endif
            DL_STAMP(11);
            mfma_split6(hacc[0], a0[0], b[0]);
            mfma_split6(hacc[1], a1[0], b[0]);
            DL_STAMP(12);
#if DL_PROJ_STAGGER
            if (!early) stage_work();
#else
            stage_work();
#endif
            mfma_split6(hacc[0], a0[1], b[1]);
            mfma_split6(hacc[1], a1[1], b[1]);
            DL_STAMP(15);
        } else {
        // lane half h owns features h*FC/2 .. h*FC/2 + FC/2-1 of the chunk; MFMA block j takes 2 quads of them
        // per operand row (3 ds_read_b128, 16 MFMAs), block j+1's reads are issued ahead of block j's MFMAs
        const float* xb = xs + (s & 1) * TN * LDT + (wn * 32 + li) * LDT + half * (FC / 2);
        const float* wb = w1s + (s & 1) * TH * LDT + (wh * 64 + li) * LDT + half * (FC / 2);
        constexpr int NB = FC / 16;
        float4 a[2][2][2], b[2][2];                             // [parity of block][...]
        auto read_block = [&](int j) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                b[j & 1][q] = *reinterpret_cast<const float4*>(xb + 8 * j + 4 * q);
                a[j & 1][0][q] = *reinterpret_cast<const float4*>(wb + 8 * j + 4 * q);
                a[j & 1][1][q] = *reinterpret_cast<const float4*>(wb + 32 * LDT + 8 * j + 4 * q);
            }
        };
        read_block(0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j + 1 < NB) read_block(j + 1);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                DL_MFMA(hacc[0], a[j & 1][0][q].x, b[j & 1][q].x);
                DL_MFMA(hacc[1], a[j & 1][1][q].x, b[j & 1][q].x);
                DL_MFMA(hacc[0], a[j & 1][0][q].y, b[j & 1][q].y);
                DL_MFMA(hacc[1], a[j & 1][1][q].y, b[j & 1][q].y);
                DL_MFMA(hacc[0], a[j & 1][0][q].z, b[j & 1][q].z);
                DL_MFMA(hacc[1], a[j & 1][1][q].z, b[j & 1][q].z);
                DL_MFMA(hacc[0], a[j & 1][0][q].w, b[j & 1][q].w);
                DL_MFMA(hacc[1], a[j & 1][1][q].w, b[j & 1][q].w);
            }
            if (j == 0) {
                if (s + 1 < steps) stash(s + 1);
                if (s + 2 < steps) fetch(s + 2);
            }
        }
        }   // !SPLIT
        if (last) {
            if constexpr (SPLIT) {
                if (nfc == 1 && tid < TH) bias_s[tid] = bias_q;
                __syncthreads();        // the biases are staged; every wave has this step's operands in registers: the tile buffer is free
#pragma unroll
                for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        bias[ht][g] = *reinterpret_cast<const float4*>(bias_s + wh * 64 + ht * 32 + 8 * g + 4 * half);
            }
            // bias + ReLU on hidT (row = hidden unit, column = node), then layer 2 straight from the registers
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (!SPLIT) bias[ht][g] = zero4<VEC>(bias[ht][g], hbase + ht * 32 + 8 * g + 4 * half, nhid);   // (SPLIT: staged as zeros)
                    hacc[ht][4 * g + 0] = fmaxf(hacc[ht][4 * g + 0] + bias[ht][g].x, 0.0f);
                    hacc[ht][4 * g + 1] = fmaxf(hacc[ht][4 * g + 1] + bias[ht][g].y, 0.0f);
                    hacc[ht][4 * g + 2] = fmaxf(hacc[ht][4 * g + 2] + bias[ht][g].z, 0.0f);
                    hacc[ht][4 * g + 3] = fmaxf(hacc[ht][4 * g + 3] + bias[ht][g].w, 0.0f);
                }
            if (hid_out != nullptr) {                           // keep the hidden layer for the backward: hidT[k][h][n]
                const int n = n0 + wn * 32 + li;
#pragma unroll
                for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int h = hbase + ht * 32 + acc_row(r, half);
                        if (h < nhid && n < N) hid_out[((size_t)k * nhid + h) * ldh + n] = hacc[ht][r];
                    }
            }
            DL_STAMP(20);
            if constexpr (SPLIT) {
                // layer 2 on the bf16 matrix path too: the post-ReLU accumulator is split into its three planes in
                // registers (slot s of block b = register 8b + s), W2 comes pre-split in the matching order
                bf16x8 hp[2][2][3];
#pragma unroll
                for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int s8 = 0; s8 < 8; ++s8) {
                            __bf16 hi, mid, lo;
                            split3(hacc[ht][8 * b + s8], hi, mid, lo);
                            hp[ht][b][0][s8] = hi; hp[ht][b][1][s8] = mid; hp[ht][b][2][s8] = lo;
                        }
                if constexpr (W2LDS) {
                    // the slice was requested at the top of this step (a layer-1 step, the bias / ReLU and the plane split ago)
                    w2st.stash(pbuf + (s & 1) * PBUF, tid);
                    __syncthreads();                                 // the staged slice is complete
#pragma unroll
                    for (int q = 0; q < W2PF; ++q) load_w2p(w2a[q], q);
                }
                DL_STAMP(21);
#pragma unroll
                for (int grp = 0; grp < 4 * DT; ++grp) {
                    if (grp + W2PF < 4 * DT) load_w2p(w2a[(grp + W2PF) % (W2PF + 1)], grp + W2PF);
                    mfma_split6(zacc[grp >> 2], w2a[grp % (W2PF + 1)], hp[(grp >> 1) & 1][grp & 1]);
                }
                DL_STAMP(22);
            } else {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                float4 wv[2][4];
#pragma unroll
                for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                    for (int g = 0; g < 4; ++g) wv[ht][g] = zero4<VEC>(wnext[ht][g], hbase + ht * 32 + 8 * g + 4 * half, nhid);
                if (dt + 1 < DT) load_w2(wnext, dt + 1, hbase);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int ht = 0; ht < 2; ++ht) {
                        DL_MFMA(zacc[dt], wv[ht][g].x, hacc[ht][4 * g + 0]);
                        DL_MFMA(zacc[dt], wv[ht][g].y, hacc[ht][4 * g + 1]);
                        DL_MFMA(zacc[dt], wv[ht][g].z, hacc[ht][4 * g + 2]);
                        DL_MFMA(zacc[dt], wv[ht][g].w, hacc[ht][4 * g + 3]);
                    }
            }
            }   // !SPLIT
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) zero_acc(hacc[ht]);
        }
        DL_STAMP(16);
        // LDS-DMA data is ordered for a ds_read only by the issuing wave's vmcnt followed by a barrier the reader has passed;
        // the workgroup fence of __syncthreads() waits for LDS operations only.  (hipcc already emitted this wait here, for
        // the pending bias load: written out so that the tile of step s + 1 does not depend on that.)
        if constexpr (DMA) wait_vmem();
        __syncthreads();
        DL_STAMP(17);
    }
    // The two hidden halves (wh = 0, 1) of a node quarter hold partial Z sums: wave wh = 1 hands its half
    // over through LDS and wave wh = 0 adds (fixed order), adds b2 and stores (registers 4g..4g+3 are 4
    // consecutive dd of node column li).
    float* red = lds;                                           // [wn][dt][16][64]
    if (wh == 1) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wn * DT + dt) * 16 + r) * 64 + lane] = zacc[dt][r];
    }
    __syncthreads();
    const int n = n0 + wn * 32 + li;
    DL_STAMP(30);
    if (wh == 0 && n < N) {
        float* orow = out + (((size_t)grp * N + n) * K + k) * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dd = dt * 32 + 8 * g + 4 * half;
                float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
                if (b2 != nullptr) bb = *reinterpret_cast<const float4*>(b2 + (size_t)k * D + dd);
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = zacc[dt][4 * g + j] + red[((wn * DT + dt) * 16 + 4 * g + j) * 64 + lane];
                *reinterpret_cast<float4*>(orow + dd) = make_float4(o[0] + bb.x, o[1] + bb.y, o[2] + bb.z, o[3] + bb.w);
            }
    }
    DL_STAMP(31);
}

#ifdef DL_PROJ_STAMPS
}  // namespace project
}  // namespace dl
extern "C" int dl_debug_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dl::project::dl_proj_stamps), sizeof(unsigned long long) * 2 * 512);
}
namespace dl {
namespace project {
#endif

constexpr size_t project2_lds(int D, bool split) {
    const size_t red = sizeof(float) * 4 * (D / 32) * 16 * 64;                 // the Z hand-over at the end
    const size_t stage = split ? sizeof(__bf16) * 2 * 3 * (TN + TH) * SPLIT_PITCH + sizeof(float) * TH : sizeof(float) * (2 * TN + 2 * TH) * (fwd_fc(D) + 4);
    return stage > red ? stage : red;
}

// Z[n][c] = b2[c] + sum_g slab[g][n][c] (g ascending), c over K*D; 4 columns per thread.
__global__ __launch_bounds__(256) void z_slab_sum_kernel(const float* __restrict__ slab, int G, size_t NC, int C,
                                                         const float* __restrict__ b2, float* __restrict__ Z) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= NC) return;
    float4 acc = *reinterpret_cast<const float4*>(b2 + i % C);
    for (int g = 0; g < G; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)g * NC + i);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(Z + i) = acc;
}

// Single-layer projection (Factor): W [K][D][F], b [K][D]:  Z[n][k][:] = W_k x[n] + b_k.
template <int D>
__global__ __launch_bounds__(256) void project1_fwd_kernel(const float* __restrict__ x, int N, int F,
                                                           const float* __restrict__ W, const float* __restrict__ b,
                                                           float* __restrict__ Z, int K) {
    constexpr int DT = D / 32;
    constexpr int TILE_N = 128;                               // nodes per workgroup (4 waves x 32)
    constexpr int FC = D == 128 ? 32 : 64;                    // keep xs + ws inside 64 KiB of static LDS
    __shared__ float xs[TILE_N * (FC + 1)];
    __shared__ float ws[D * (FC + 1)];
    const int k = blockIdx.y;
    const int n0 = blockIdx.x * TILE_N;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, half = lane >> 5;
    const float* Wk = W + (size_t)k * D * F;
    f32x16 zacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) zacc[dt][r] = 0.0f;
    for (int f0 = 0; f0 < F; f0 += FC) {
        __syncthreads();
        stage_tile(xs, FC + 1, x, N, F, n0, f0, TILE_N, FC);
        stage_tile(ws, FC + 1, Wk, D, F, 0, f0, D, FC);
        __syncthreads();
        const float* xa = xs + (wave * 32 + li) * (FC + 1) + half;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const float* wa = ws + (dt * 32 + li) * (FC + 1) + half;
#pragma unroll 8
            for (int s = 0; s < FC / 2; ++s)
                zacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[2 * s], xa[2 * s], zacc[dt], 0, 0, 0);
        }
    }
    const int n = n0 + wave * 32 + li;
    if (n < N) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = dt * 32 + 8 * g4 + 4 * half;
                const float4 bb = *reinterpret_cast<const float4*>(b + (size_t)k * D + dd);
                float4 o;
                o.x = zacc[dt][4 * g4 + 0] + bb.x;
                o.y = zacc[dt][4 * g4 + 1] + bb.y;
                o.z = zacc[dt][4 * g4 + 2] + bb.z;
                o.w = zacc[dt][4 * g4 + 3] + bb.w;
                *reinterpret_cast<float4*>(Z + ((size_t)n * K + k) * D + dd) = o;
            }
        }
    }
}

}  // namespace project

// Layer-1 products from three bf16 planes per operand (default) or plain fp32 MFMA (DL_PROJECT_FP32_MFMA=1).
bool split_products() {
    return !config().project_fp32_mfma;
}

bool project_supported(int d) { return d == 32 || d == 64 || d == 128; }

// Hidden-chunk groups per (node tile, factor): 1 when the grid already covers the 256 CUs twice over, else
// enough groups to get there (each group = whole 128-unit chunks; measured at N=5,201: 2 groups beat 1 and 4).
static int project2_groups(int N, int K, int nhid) {
    const long long wg = (long long)((N + project::TN - 1) / project::TN) * K;
    const int nhc = (nhid + project::TH - 1) / project::TH;
    if (config().fwd_groups > 0) return std::max(1, std::min(nhc, config().fwd_groups));           // DL_FWD_GROUPS: tuning knob
    if (wg >= 512) return 1;
    const int cpg = std::max(1, (int)(nhc / std::min<long long>(nhc, (512 + wg - 1) / wg)));
    return (nhc + cpg - 1) / cpg;
}

// Node block of the two-layer forward: the x planes are made per block of rows, so the workspace does not grow with
// the graph (2.9M nodes x 288 features would be 5 GB of planes).  Large graphs only — they never use the group split.
static int fwd_block_rows(int N) {
    long long rows = 1 << 17;
    if (config().fwd_block_rows > 0) rows = config().fwd_block_rows * project::TN;                  // DL_FWD_BLOCK_ROWS (tests: force blocking)
    return N <= 2 * rows ? N : (int)rows;
}

// Workspace of the two-layer forward: [Z slabs of the G groups][x planes of one node block][W1 planes]
struct FwdLayout { int G, R, nhid_p; size_t off_xp, off_wp, off_w2p, bytes; };
static FwdLayout fwd_layout(int N, int F, int K, int nhid, int d) {
    FwdLayout L{};
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    L.R = fwd_block_rows(N);
    L.G = L.R < N ? 1 : project2_groups(N, K, nhid);
    size_t off = L.G > 1 ? al(sizeof(float) * (size_t)L.G * N * K * d) : 0;
    L.off_xp = off;
    if (split_products()) {
        off += al(sizeof(__bf16) * project::plane_array_elems(L.R, F, project::SPLIT_COLS));
        L.off_wp = off;
        off += al(sizeof(__bf16) * K * project::plane_array_elems(nhid, F, project::SPLIT_COLS));
        L.off_w2p = off;
        L.nhid_p = (int)project::round_up(nhid, project::TH);
        off += al(sizeof(__bf16) * 3 * (size_t)K * d * L.nhid_p);
    }
    L.bytes = off;
    return L;
}

size_t project_fwd_workspace_bytes(int N, int F, int K, int nhid, int d, bool two_layer) {
    if (!two_layer || N <= 0) return 0;
    return fwd_layout(N, F, K, nhid, d).bytes;
}

template <int D, bool VEC, bool SPLIT>
static void launch2_t(int N, int K, int G, int cpg, hipStream_t st, const float* x, int F, int nhid, const float* W1,
                      const float* b1, const float* W2, const float* b2, float* out, float* hid_out, int ldh,
                      project::FwdPlanes P) {
    using namespace project;
    static unsigned long long lds_done = 0;
    constexpr size_t lds = project2_lds(D, SPLIT);
    ensure_dynamic_lds(reinterpret_cast<const void*>(&project2_fwd_kernel<D, VEC, SPLIT>), lds, lds_done);
    const dim3 grid((unsigned)xcd_grid((N + TN - 1) / TN, K * G));
    hipLaunchKernelGGL((project2_fwd_kernel<D, VEC, SPLIT>), grid, dim3(NTHR), lds, st, x, N, F, nhid, W1, b1, W2, b2,
                       out, K, G, cpg, hid_out, ldh, P);
}

// Persistent planes of the (constant) feature matrix: [x planes, 32-column tiles | x^T planes, 16-column tiles], both
// tile-major and zero-padded like the per-call arrays (dl_tiles.h).  Only single-block problems use them (fwd_block_rows,
// bwd_block_rows): a blocked run re-splits the block it works on.
size_t project_xplanes_bytes(int N, int F) {
    using namespace project;
    if (N <= 0 || F <= 0) return 0;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return al(sizeof(__bf16) * plane_array_elems(N, F, SPLIT_COLS)) + al(sizeof(__bf16) * plane_array_elems(F, N, PLANE_ROWS));
}
static size_t xplanes_xT_offset(int N, int F) {
    using namespace project;
    return (sizeof(__bf16) * plane_array_elems(N, F, SPLIT_COLS) + 255) & ~(size_t)255;
}
const void* project_xplanes_xT(const void* xplanes, int N, int F) {
    return xplanes ? static_cast<const char*>(xplanes) + xplanes_xT_offset(N, F) : nullptr;
}
int project_xplanes_build(const float* x, int N, int F, void* xplanes, hipStream_t st) {
    using namespace project;
    char* base = static_cast<char*>(xplanes);
    split_rows(x, 1, N, F, F, 0, reinterpret_cast<__bf16*>(base), st);
    split_transposed(x, N, F, F, reinterpret_cast<__bf16*>(base + xplanes_xT_offset(N, F)), st);
    return check_launch("project_xplanes_build");
}

int project_fwd(const float* x, int N, int F, int K, int nhid, int d, const float* W1, const float* b1,
                const float* W2, const float* b2, float* Z, void* ws, size_t ws_bytes, float* hid_out, hipStream_t st,
                const void* xplanes) {
    using namespace project;
    const int ldh = (N + 3) & ~3;
    if (W2 == nullptr) {          // single Linear(F -> d): W1 is [K][d][F], b1 is [K][d]
        const dim3 grid((unsigned)((N + 127) / 128), (unsigned)K), block(256);
        if (d == 32) hipLaunchKernelGGL(project1_fwd_kernel<32>, grid, block, 0, st, x, N, F, W1, b1, Z, K);
        else if (d == 64) hipLaunchKernelGGL(project1_fwd_kernel<64>, grid, block, 0, st, x, N, F, W1, b1, Z, K);
        else hipLaunchKernelGGL(project1_fwd_kernel<128>, grid, block, 0, st, x, N, F, W1, b1, Z, K);
        return check_launch("project_fwd");
    }
    const FwdLayout L = fwd_layout(N, F, K, nhid, d);
    const bool fits = ws != nullptr && ws_bytes >= L.bytes;
    const bool split = split_products() && fits;            // no workspace: fp32 MFMA straight from x and W1
    const bool vec = (split || F % 4 == 0) && (nhid % 4 == 0);      // quads never straddle a row end
    const int G = fits ? L.G : 1;                           // no workspace: one group
    FwdPlanes P{};
    if (split) {
        char* base = static_cast<char*>(ws);
        __bf16* xP = reinterpret_cast<__bf16*>(base + L.off_xp);
        __bf16* wP = reinterpret_cast<__bf16*>(base + L.off_wp);
        __bf16* w2P = reinterpret_cast<__bf16*>(base + L.off_w2p);
        if (L.R >= N) {                                     // one node block: all three operand splits in one launch
            if (xplanes) xP = const_cast<__bf16*>(static_cast<const __bf16*>(xplanes));     // x was split once for the run
            split_fwd_operands(xplanes ? nullptr : x, N, F, xP, W1, K, nhid, wP, W2, d, w2P, L.nhid_p, st);
        } else {
            split_rows(W1, K, nhid, F, F, (size_t)nhid * F, wP, st);
            split_w2(W2, K * d, nhid, w2P, L.nhid_p, st);
        }
        P = FwdPlanes{xP, wP, plane_array_elems(nhid, F, SPLIT_COLS), plane_chunks<SPLIT_COLS>(F, SPLIT_COLS),
                      w2P, (size_t)K * d * L.nhid_p, L.nhid_p};
    }
    const int nhc = (nhid + TH - 1) / TH;
    const int cpg = (nhc + G - 1) / G;
    float* out = G > 1 ? static_cast<float*>(ws) : Z;
    const float* bias2 = G > 1 ? nullptr : b2;
    const int R = fits ? L.R : N;                           // rows per launch (G == 1 whenever R < N)
    for (int row0 = 0; row0 < N; row0 += R) {
        const int rows = std::min(R, N - row0);
        const float* xb = x + (size_t)row0 * F;
        float* ob = out + (size_t)row0 * K * d;
        float* hb = hid_out ? hid_out + row0 : nullptr;
        if (split && L.R < N) split_rows(xb, 1, rows, F, F, 0, const_cast<__bf16*>(P.x), st);
#define DL_P2(DD)                                                                                       \
    if (d == DD) {                                                                                      \
        if (split && vec) launch2_t<DD, true, true>(rows, K, G, cpg, st, xb, F, nhid, W1, b1, W2, bias2, ob, hb, ldh, P);   \
        else if (split) launch2_t<DD, false, true>(rows, K, G, cpg, st, xb, F, nhid, W1, b1, W2, bias2, ob, hb, ldh, P);    \
        else if (vec) launch2_t<DD, true, false>(rows, K, G, cpg, st, xb, F, nhid, W1, b1, W2, bias2, ob, hb, ldh, P);      \
        else launch2_t<DD, false, false>(rows, K, G, cpg, st, xb, F, nhid, W1, b1, W2, bias2, ob, hb, ldh, P);              \
    }
        DL_P2(32) DL_P2(64) DL_P2(128)
#undef DL_P2
    }
    if (G > 1) {
        const size_t NC = (size_t)N * K * d;
        hipLaunchKernelGGL(z_slab_sum_kernel, dim3((unsigned)((NC / 4 + 255) / 256)), dim3(256), 0, st, out, G, NC, K * d, b2, Z);
    }
    return check_launch("project_fwd");
}

}  // namespace dl
