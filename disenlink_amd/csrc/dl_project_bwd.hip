// Backward of the factor projection on the matrix cores: the weight gradients of the K factor MLPs
// (autograd of model.py:13-15, 24-27 as driven by main_disentangled.py:198).  x is data: no dx.
//
//   two-layer:   hid = relu(x W1_k^T + b1_k)               recomputed, never read from HBM
//                dhid = (dZ_k W2_k) . [hid > 0]            -> workspace [N][K][nhid]
//                dW2_k = dZ_k^T hid,  db1_k = colsum(dhid)  (kernel A: project2_bwd_hidden_kernel)
//                dW1_k = dhid^T x                           (kernel B: nodes_contract_kernel)
//                db2   = colsum(dZ)                         (colsum_kernel)
//   one layer:   dW_k = dZ_k^T x (kernel B),  db = colsum(dZ)
//
// Every contraction over the NODE index is split into S node ranges whose partial results go to
// separate slabs and are added in slab order by slab_sum_kernel: no float atomics, so the gradients
// are bitwise reproducible run to run (like the rest of the path).
//
// Register layout used throughout: a 32x32 MFMA accumulator keeps its COLUMN on the lane and its 16
// rows (acc_row(r, half)) in the registers.  Kernel A computes hid and dhid as [node][hidden] tiles
// (node = row), so both can be fed back unchanged as the B operand of products that contract over the
// node index — dW2 += dZ^T . hid takes hid straight from the accumulator registers.
#include <algorithm>
#include <cstdlib>
#include "dl_common.h"
#include "dl_kernels.h"
#include "dl_tiles.h"

namespace dl {
namespace project {

constexpr int TILE_N = 128;   // nodes per workgroup step (4 wave quarters x 32)
constexpr int BFC = 32;       // feature chunk staged per step
constexpr int LDB = BFC + 4;  // LDS row pitch of the layer-1 operand tiles
constexpr int BTHR = 512;

// hidden units per workgroup of kernel A: 2 wave halves x HT tiles of 32 (HT = 1 for D = 128: LDS budget)
constexpr int bwd_ht(int D) { return D <= 64 ? 2 : 1; }

// ------------------------------------------------------------------------------------------------
// Kernel A.  1-D grid of xcd_grid(S node ranges, hidden chunks of 64*HT x K); one workgroup = 8 waves = 128 nodes per
// node tile: node quarter wn (32 nodes) x hidden half wh (32*HT units).  Same write-after-barrier staging
// pipeline as the forward kernel (dl_project.hip).  Per node tile, once the layer-1 sum over F is complete:
//   hid  = relu(acc + b1)                       [node][hidden]: node rows in registers, hidden on lanes
//   dW2 += dZ^T . hid                           A = dZ^T from LDS (b32), B = hid registers
//   dhid = (dZ . W2^T) masked by hid > 0        A = dZ rows, B = W2^T rows, both b128 from LDS
// VEC: F % 4 == 0 (x and W1 rows are sequences of aligned quads).
// RECOMPUTE = false: the forward kept the hidden layer (hidT [K][nhid][ldh], post-ReLU); it is read straight into the
// B-operand registers (lane = hidden unit, 4 consecutive nodes per register quad) and layer 1 is not run again.
// PLANES: dhid leaves as the tile-major bf16 planes of dhid_k^T ([hidden][node], 16-node tiles: the A operand of
// nodes_contract_planes_kernel) instead of fp32 [node][K][hidden]; every element of the padded array that kernel
// reads along the node axis is written (zeros past N and past nhid).
struct DhidPlanes { __bf16* base; size_t batch; int ncb; };       // factor k at base + k * batch; ncb 16-node chunks per row block

#ifndef DL_BWD_A_BF16
#define DL_BWD_A_BF16 1           // -DDL_BWD_A_BF16=0: kernel A's two products on fp32 MFMA (the round-2 form), for A/B runs
#endif
#ifndef DL_BWD_A_BF16_MAXD
#define DL_BWD_A_BF16_MAXD 128    // largest factor width that takes the bf16 path (its planes must fit the kernel's LDS)
#endif
// LDS bytes of kernel A (also used by its launcher below)
constexpr size_t project2_bwd_lds_bytes(int D) {
    const size_t ht = D <= 64 ? 2 : 1;
    const size_t stage = 2 * 128 * (32 + 4) + 2 * 64 * ht * (32 + 4) + 128 * (size_t)(D + 4) + 64 * ht * (size_t)(D + 4);
    const size_t red = 8 * (size_t)((D / 32) * ht * 16 + ht) * 64;
    return sizeof(float) * (stage > red ? stage : red);
}

// -DDL_PROJA_STAMPS=<workgroup index>: DIAGNOSTIC build (like DL_PROJ_STAMPS in dl_project.hip): s_memtime at the phase
// boundaries of every node tile, waves 0 and 4 of one workgroup; read back by dl_debug_read_stamps_a.
#ifdef DL_PROJA_STAMPS
__device__ unsigned long long dl_proja_stamps[2][512];
#define DLA_STAMP(code)                                                                                 \
    do {                                                                                                \
        if (stamp_on && stamp_n < 510) {                                                                \
            dl_proja_stamps[stamp_w][stamp_n++] = ((unsigned long long)(code) << 56) | (__builtin_amdgcn_s_memtime() & 0x00FFFFFFFFFFFFFFull); \
            dl_proja_stamps[stamp_w][511] = stamp_n;                                                    \
        }                                                                                               \
    } while (0)
#else
#define DLA_STAMP(code) do {} while (0)
#endif

template <int D, bool VEC, bool RECOMPUTE, bool PLANES>
__global__ __launch_bounds__(BTHR) void project2_bwd_hidden_kernel(
    const float* __restrict__ x, int N, int F, int nhid, const float* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ W2, const float* __restrict__ dZ, int K, int tiles_per_range,
    float* __restrict__ dhid, float* __restrict__ dW2p, float* __restrict__ db1p,
    const float* __restrict__ hidT, int ldh, int hid_cols, DhidPlanes dhp) {
    constexpr int DT = D / 32, HT = bwd_ht(D), HB = 64 * HT;
    constexpr int LDZ = D + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                               // [2][TILE_N][LDB]
    float* w1s = xs + 2 * TILE_N * LDB;            // [2][HB][LDB]
    float* dzs = w1s + 2 * HB * LDB;               // [TILE_N][LDZ]   dZ_k rows of the current node tile
    float* w2t = dzs + TILE_N * LDZ;               // [HB][LDZ]       W2_k[:, chunk]^T, staged once
    // BF (round 5; kept hidden layer, d <= 64): both products of a node tile — dW2 += dZ^T . hid and dhid = dZ . W2^T — on
    // the bf16 matrix path from three bf16 planes per operand (dl_tiles.h: six exact products per term, fp32-grade) instead
    // of fp32 MFMA: 96 MFMAs of 32 cycles per wave and tile instead of 128 of 64 (stamps, tools/proja_stamps.py: fp32 MFMA
    // issue was 58 % of a tile).  dZ tile and W2^T chunk live in LDS as planes [3][rows][ZP]; the A operand of the node
    // contraction (dZ^T: lane = dd, k-slots = 8 nodes) comes out of the [node][dd] image by the transposed LDS read
    // ds_read_b64_tr_b16; its B operand is the hidden layer in registers, split in place (slot s of block b = register 8b + s).
    constexpr bool BF = DL_BWD_A_BF16 && PLANES && !RECOMPUTE && D <= DL_BWD_A_BF16_MAXD;
    static_assert(!BF || sizeof(__bf16) * 3 * (TILE_N + HB) * (D + 8) <= project2_bwd_lds_bytes(D), "planes of the dZ tile and the W2^T chunk fit the kernel's LDS");
    constexpr int ZP = D + 8;                      // plane row pitch (bf16): 16-byte rows reads conflict-free, 8-byte aligned
    __bf16* dzp = reinterpret_cast<__bf16*>(lds);  // BF: [3][TILE_N][ZP]
    __bf16* w2p = dzp + 3 * TILE_N * ZP;           //     [3][HB][ZP]      W2_k[:, chunk]^T as planes
    const int n_tiles = (N + TILE_N - 1) / TILE_N;
    const int nhc = (nhid + HB - 1) / HB;
    // a = node range (shares the x tiles), b = (hidden chunk, factor) (shares the W1 chunk): see xcd_item
    const XcdItem item = xcd_item(blockIdx.x, (n_tiles + tiles_per_range - 1) / tiles_per_range, nhc * K);
    if (!item.valid) return;
    const int hc = item.b % nhc, k = item.b / nhc, rng = item.a;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, half = lane >> 5;
    const int wn = wave >> 1, wh = wave & 1;
    const int tile0 = rng * tiles_per_range;
    const int my_tiles = max(0, min(tiles_per_range, n_tiles - tile0));
    const float* W1k = W1 + (size_t)k * nhid * F;
    const float* W2k = W2 + (size_t)k * D * nhid;
    const float* dZk = dZ + (size_t)k * D;         // row n at dZk + n*K*D
    const int nfc = RECOMPUTE ? (F + BFC - 1) / BFC : 1;
    const int steps = my_tiles * nfc;
    const int hw0 = hc * HB + wh * 32 * HT;        // first hidden unit of this wave
#ifdef DL_PROJA_STAMPS
    const bool stamp_on = (int)blockIdx.x == DL_PROJA_STAMPS && (wave == 0 || wave == 4) && lane == 0;
    const int stamp_w = wave >> 2;
    int stamp_n = 0;
#endif
    DLA_STAMP(1);

    {   // W2^T chunk: w2t[h][dd] = W2_k[dd][hc*HB + h].  All loads first, then the LDS stores: written as one loop the
        // compiler waited for every load before its store — 16 global round trips in a row, 16,000 cycles of a workgroup's
        // 183,000 (stamps, tools/proja_stamps.py).
        constexpr int NW = HB * D / BTHR;
        static_assert(HB * D % BTHR == 0, "the W2^T chunk divides over the workgroup");
        float wv[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = tid + j * BTHR, dd = i / HB, h = i - dd * HB;
            const int hh = hc * HB + h;
            wv[j] = W2k[(size_t)dd * nhid + (hh < nhid ? hh : 0)];
            wv[j] = hh < nhid ? wv[j] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = tid + j * BTHR, dd = i / HB, h = i - dd * HB;
            if constexpr (BF) {
                __bf16 hi, mid, lo;
                split3(wv[j], hi, mid, lo);
                w2p[(0 * HB + h) * ZP + dd] = hi;
                w2p[(1 * HB + h) * ZP + dd] = mid;
                w2p[(2 * HB + h) * ZP + dd] = lo;
            } else {
                w2t[h * LDZ + dd] = wv[j];
            }
        }
    }
    float b1v[HT];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) {
        const int h = hw0 + ht * 32 + li;
        b1v[ht] = h < nhid ? b1[(size_t)k * nhid + h] : 0.0f;
    }

    TileStage<TILE_N, BFC, VEC, BTHR> xt;
    TileStage<HB, BFC, VEC, BTHR> wt;
    TileStage<TILE_N, D, true, BTHR> zt;           // dZ rows: D % 32 == 0, row stride K*D: always aligned quads
    auto fetch = [&](int s) {
        const int tl = s / nfc, fc = s % nfc;
        const int n0 = (tile0 + tl) * TILE_N;
        xt.fetch(x + (size_t)n0 * F + fc * BFC, F, N - n0, F - fc * BFC, tid);
        wt.fetch(W1k + (size_t)hc * HB * F + fc * BFC, F, nhid - hc * HB, F - fc * BFC, tid);
    };
    auto stash = [&](int s) {
        xt.template stash<LDB>(xs + (s & 1) * TILE_N * LDB, tid);
        wt.template stash<LDB>(w1s + (s & 1) * HB * LDB, tid);
    };

    f32x16 hacc[HT], w2acc[DT][HT];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) {
        zero_acc(hacc[ht]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) zero_acc(w2acc[dt][ht]);
    }
    float b1acc[HT];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) b1acc[ht] = 0.0f;

    // Kept hidden layer: hidT rows of this wave's hidden units, quad g = nodes 8g + 4*half .. +3 of its node quarter.
    // Raw loads from clamped addresses; the masking happens at the use (hq_masked), so a tile's loads can be issued
    // one tile ahead, behind the MFMAs of the current one.
    float4 hraw[HT][4];
    auto load_hq = [&](int n0t) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
            const int h = hw0 + ht * 32 + li;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0t + wn * 32 + 8 * g + 4 * half;
                const bool ok = h < nhid && n + 3 < hid_cols;
                hraw[ht][g] = *reinterpret_cast<const float4*>(hidT + (ok ? ((size_t)k * nhid + h) * ldh + n : 0));
            }
        }
    };
    if (RECOMPUTE && steps > 0) {
        fetch(0);
        stash(0);
        if (steps > 1) fetch(1);
    }
    if (!RECOMPUTE && steps > 0) {
        zt.fetch(dZk + (size_t)tile0 * TILE_N * K * D, K * D, N - tile0 * TILE_N, D, tid);
        load_hq(tile0 * TILE_N);
    }
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int tl = s / nfc, fc = s % nfc;
        const bool last = fc == nfc - 1;
        const int n0 = (tile0 + tl) * TILE_N;
        if (RECOMPUTE && last) zt.fetch(dZk + (size_t)n0 * K * D, K * D, N - n0, D, tid);   // consumed after the MFMAs below
        if constexpr (RECOMPUTE) {
        // hid[node][hidden] += x[node][f] . W1[hidden][f]: A = x rows of this node quarter, B = W1 rows
        const float* xb = xs + (s & 1) * TILE_N * LDB + (wn * 32 + li) * LDB + half * (BFC / 2);
        const float* wb = w1s + (s & 1) * HB * LDB + (wh * 32 * HT + li) * LDB + half * (BFC / 2);
        constexpr int NB = BFC / 16;
        float4 a[2][2], b[2][HT][2];
        auto read_block = [&](int j) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                a[j & 1][q] = *reinterpret_cast<const float4*>(xb + 8 * j + 4 * q);
#pragma unroll
                for (int ht = 0; ht < HT; ++ht)
                    b[j & 1][ht][q] = *reinterpret_cast<const float4*>(wb + ht * 32 * LDB + 8 * j + 4 * q);
            }
        };
        read_block(0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j + 1 < NB) read_block(j + 1);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int ht = 0; ht < HT; ++ht) DL_MFMA(hacc[ht], a[j & 1][q].x, b[j & 1][ht][q].x);
#pragma unroll
                for (int ht = 0; ht < HT; ++ht) DL_MFMA(hacc[ht], a[j & 1][q].y, b[j & 1][ht][q].y);
#pragma unroll
                for (int ht = 0; ht < HT; ++ht) DL_MFMA(hacc[ht], a[j & 1][q].z, b[j & 1][ht][q].z);
#pragma unroll
                for (int ht = 0; ht < HT; ++ht) DL_MFMA(hacc[ht], a[j & 1][q].w, b[j & 1][ht][q].w);
            }
            if (j == 0) {
                if (s + 1 < steps) stash(s + 1);
                if (s + 2 < steps) fetch(s + 2);
            }
        }
        }   // RECOMPUTE
        if (last) {
            DLA_STAMP(10);
            // every wave is past the previous tile's use of dzs (barrier at the end of that step)
            if constexpr (BF) {                                 // the dZ tile as three bf16 planes [node][dd]
                const bool full = zt.rows_valid >= TILE_N && zt.cols_valid >= D;
#pragma unroll
                for (int j = 0; j < zt.NV / 4; ++j) {
                    const int i = tid + BTHR * j, r = i / (D / 4), c = 4 * (i % (D / 4));
                    const unsigned m = (full || (r < zt.rows_valid && c < zt.cols_valid)) ? 0xFFFFFFFFu : 0u;
                    bf16x4 p0, p1, p2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        __bf16 hi, mid, lo;
                        split3(mask_bits(zt.v[4 * j + e], m), hi, mid, lo);
                        p0[e] = hi; p1[e] = mid; p2[e] = lo;
                    }
                    *reinterpret_cast<bf16x4*>(dzp + (0 * TILE_N + r) * ZP + c) = p0;
                    *reinterpret_cast<bf16x4*>(dzp + (1 * TILE_N + r) * ZP + c) = p1;
                    *reinterpret_cast<bf16x4*>(dzp + (2 * TILE_N + r) * ZP + c) = p2;
                }
            } else {
                zt.template stash<LDZ>(dzs, tid);
            }
            DLA_STAMP(11);
            __syncthreads();
            DLA_STAMP(12);
            float hid[HT][16];
            unsigned relu_bits[HT];                             // bit r: hid[ht][r] > 0 (the mask of the dhid epilogue)
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) {
                if constexpr (RECOMPUTE) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) hid[ht][r] = fmaxf(hacc[ht][r] + b1v[ht], 0.0f);
                } else {
                    const int h = hw0 + ht * 32 + li;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // per element: the columns N .. ldh-1 of hidT are padding nobody wrote — whatever they hold
                        // (possibly NaN bit patterns) must not meet the zero dZ rows of those nodes in an MFMA (0 * NaN)
                        const int n = n0 + wn * 32 + 8 * g + 4 * half;
                        const bool ok = h < nhid && n + 3 < hid_cols;
                        hid[ht][4 * g + 0] = mask_bits(hraw[ht][g].x, ok && n + 0 < N ? 0xFFFFFFFFu : 0u);
                        hid[ht][4 * g + 1] = mask_bits(hraw[ht][g].y, ok && n + 1 < N ? 0xFFFFFFFFu : 0u);
                        hid[ht][4 * g + 2] = mask_bits(hraw[ht][g].z, ok && n + 2 < N ? 0xFFFFFFFFu : 0u);
                        hid[ht][4 * g + 3] = mask_bits(hraw[ht][g].w, ok && n + 3 < N ? 0xFFFFFFFFu : 0u);
                    }
                }
                relu_bits[ht] = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) relu_bits[ht] |= (hid[ht][r] > 0.0f ? 1u : 0u) << r;
            }
            // kept form: the next tile's dZ rows and hidden quads are fetched behind this tile's MFMAs (unconditional —
            // the last tile is fetched twice — see project2_fwd_kernel)
            DLA_STAMP(13);
            const int n0n = (tile0 + min(tl + 1, my_tiles - 1)) * TILE_N;
            if constexpr (!RECOMPUTE) zt.fetch(dZk + (size_t)n0n * K * D, K * D, N - n0n, D, tid);
            // dW2[dd][hidden] += dZ[node][dd] . hid[node][hidden]: the k-pair of register r is the node pair
            // {acc_row(r,0), acc_row(r,1)} of this quarter
            if constexpr (BF) {
                // B operand: the hidden layer split in place — registers 8b .. 8b+7 of lane half t are the nodes
                // 16b + 8(s>>2) + 4t + (s&3), s = 0..7, of this node quarter: the 8 k-slots of K = 16 block b
                bf16x8 hidp[HT][2][3];
#pragma unroll
                for (int ht = 0; ht < HT; ++ht)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int s8 = 0; s8 < 8; ++s8) {
                            __bf16 hi, mid, lo;
                            split3(hid[ht][8 * b + s8], hi, mid, lo);
                            hidp[ht][b][0][s8] = hi; hidp[ht][b][1][s8] = mid; hidp[ht][b][2][s8] = lo;
                        }
                // A operand: dZ^T[dd = dt*32 + li][those 8 nodes] by two transposed reads per plane (4 node rows x 16 dd
                // columns per group of 16 lanes: lane 4q+p gives the address of row q, columns 4p..4p+3, and receives
                // column (lane & 15) of the 4 rows).  Every lane is active here (the read needs EXEC all ones).
                const int i16 = lane & 15, trq = i16 >> 2, trp = i16 & 3, cb = (li >> 4) * 16;
                typedef short v4s __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        bf16x8 za[3];
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            const __bf16* a0 = dzp + (p * TILE_N + wn * 32 + 16 * b + 4 * half + trq) * ZP + dt * 32 + cb + 4 * trp;
                            const v4s lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)a0);
                            const v4s hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(a0 + 8 * ZP));
                            typedef short v8s __attribute__((ext_vector_type(8)));
                            const v8s both = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
                            za[p] = __builtin_bit_cast(bf16x8, both);
                        }
#pragma unroll
                        for (int ht = 0; ht < HT; ++ht) mfma_split6(w2acc[dt][ht], za, hidp[ht][b]);
                    }
            } else {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                float zv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) zv[r] = dzs[(wn * 32 + acc_row(r, half)) * LDZ + dt * 32 + li];
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int ht = 0; ht < HT; ++ht) DL_MFMA(w2acc[dt][ht], zv[r], hid[ht][r]);
            }
            }
            DLA_STAMP(14);
            if constexpr (!RECOMPUTE) load_hq(n0n);             // hid has been consumed by the MFMAs above
            // dhid[node][hidden] = dZ[node][dd] . W2^T[hidden][dd], masked by the ReLU
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) zero_acc(hacc[ht]);
            if constexpr (BF) {
#pragma unroll
                for (int b = 0; b < D / 16; ++b) {              // A = dZ rows of this node quarter, B = W2^T rows, planes from LDS
                    bf16x8 za[3], vb[HT][3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        za[p] = *reinterpret_cast<const bf16x8*>(dzp + (p * TILE_N + wn * 32 + li) * ZP + 16 * b + 8 * half);
#pragma unroll
                        for (int ht = 0; ht < HT; ++ht)
                            vb[ht][p] = *reinterpret_cast<const bf16x8*>(w2p + (p * HB + wh * 32 * HT + ht * 32 + li) * ZP + 16 * b + 8 * half);
                    }
#pragma unroll
                    for (int ht = 0; ht < HT; ++ht) mfma_split6(hacc[ht], za, vb[ht]);
                }
            }
            const float* za = dzs + (wn * 32 + li) * LDZ + half * (D / 2);
            const float* va = w2t + (wh * 32 * HT + li) * LDZ + half * (D / 2);
#pragma unroll
            for (int q = 0; q < (BF ? 0 : D / 8); ++q) {
                const float4 zq = *reinterpret_cast<const float4*>(za + 4 * q);
                float4 vq[HT];
#pragma unroll
                for (int ht = 0; ht < HT; ++ht) vq[ht] = *reinterpret_cast<const float4*>(va + ht * 32 * LDZ + 4 * q);
#pragma unroll
                for (int ht = 0; ht < HT; ++ht) DL_MFMA(hacc[ht], zq.x, vq[ht].x);
#pragma unroll
                for (int ht = 0; ht < HT; ++ht) DL_MFMA(hacc[ht], zq.y, vq[ht].y);
#pragma unroll
                for (int ht = 0; ht < HT; ++ht) DL_MFMA(hacc[ht], zq.z, vq[ht].z);
#pragma unroll
                for (int ht = 0; ht < HT; ++ht) DL_MFMA(hacc[ht], zq.w, vq[ht].w);
            }
            DLA_STAMP(15);
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) {
                const int h = hw0 + ht * 32 + li;
                float colsum = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float g = (relu_bits[ht] >> r) & 1u ? hacc[ht][r] : 0.0f;
                    const int n = n0 + wn * 32 + acc_row(r, half);
                    if constexpr (!PLANES) {
                        if (n < N && h < nhid) dhid[((size_t)n * K + k) * nhid + h] = g;
                    }
                    colsum += g;
                    hacc[ht][r] = g;
                }
                if constexpr (PLANES) {
                    // Registers 4q..4q+3 of lane half t are the nodes 8q + 4t .. +3 of row h.  A 16-node tile row
                    // (32 bytes per plane) is [t=0,q=0][t=1,q=0][t=0,q=1][t=1,q=1]: the halves swap one quad (half 0
                    // gives its q=1, half 1 its q=0) and each writes 16 contiguous bytes — whole 32-byte sectors.
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        bf16x4 pl[2][3];                                    // [q - 2qq][plane]
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                __bf16 hi, mid, lo;
                                split3(hacc[ht][4 * (2 * qq + j) + e], hi, mid, lo);
                                pl[j][0][e] = hi; pl[j][1][e] = mid; pl[j][2][e] = lo;
                            }
                        const int n = n0 + wn * 32 + 16 * qq;
                        __bf16* o = dhp.base + (size_t)k * dhp.batch + plane_tile<16>(h / PLANE_ROWS, n / 16, dhp.ncb) +
                                    (h % PLANE_ROWS) * 16 + half * 8;
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                            const u32x2 mine0 = __builtin_bit_cast(u32x2, pl[0][p]), mine1 = __builtin_bit_cast(u32x2, pl[1][p]);
                            const u32x2 give = half ? mine0 : mine1;
                            u32x2 got;
                            got.x = __shfl_xor(give.x, 32);
                            got.y = __shfl_xor(give.y, 32);
                            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                            u32x4 outv;
                            if (half) { outv.x = got.x; outv.y = got.y; outv.z = mine1.x; outv.w = mine1.y; }
                            else { outv.x = mine0.x; outv.y = mine0.y; outv.z = got.x; outv.w = got.y; }
                            *reinterpret_cast<u32x4*>(o + p * PLANE_ROWS * 16) = outv;
                        }
                    }
                }
                b1acc[ht] += colsum;
                zero_acc(hacc[ht]);
            }
            DLA_STAMP(16);
        }
        __syncthreads();
        DLA_STAMP(17);
    }
    // cross-wave reduction (fixed order over the 4 node quarters) of the dW2 / db1 partials of this range
    constexpr int RW = DT * HT * 16 + HT;                       // floats per lane and wave
    float* red = lds;                                           // [8 waves][RW][64 lanes]
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * RW + (dt * HT + ht) * 16 + r) * 64 + lane] = w2acc[dt][ht][r];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) red[(wave * RW + DT * HT * 16 + ht) * 64 + lane] = b1acc[ht];
    __syncthreads();
    if (wn == 0) {                                              // waves 0 (wh = 0) and 1 (wh = 1) write out
        float* out = dW2p + ((size_t)rng * K + k) * D * nhid;
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
            const int h = hw0 + ht * 32 + li;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = 0.0f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v += red[((2 * q + wh) * RW + (dt * HT + ht) * 16 + r) * 64 + lane];
                    if (h < nhid) out[(size_t)(dt * 32 + acc_row(r, half)) * nhid + h] = v;
                }
            float v = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* p = red + ((2 * q + wh) * RW + DT * HT * 16 + ht) * 64;
                v += p[li] + p[32 + li];
            }
            if (half == 0 && h < nhid) db1p[((size_t)rng * K + k) * nhid + h] = v;
        }
    }
}

constexpr size_t project2_bwd_lds(int D) {
    const size_t stage = 2 * TILE_N * LDB + 2 * 64 * bwd_ht(D) * LDB + TILE_N * (D + 4) + 64 * bwd_ht(D) * (D + 4);
    const size_t red = 8 * (size_t)((D / 32) * bwd_ht(D) * 16 + bwd_ht(D)) * 64;
    return sizeof(float) * (stage > red ? stage : red);
}

// ------------------------------------------------------------------------------------------------
// Kernel B.  C[k][m][f] (+slab) = sum over the nodes of one range of  Y[n][k][m] * X[n][f]
//   Y row n at Y + n*ldY + k*M (M columns), X [N][F] row-major.  Output slab [K][M][F].
// 1-D grid of xcd_grid(S node ranges x f tiles of 128, m tiles of 128 x K); 4 waves as 2 x 2, each 64 m x 64 f (4 accumulators),
// two workgroups per CU.  Node chunk of 32 per step; MFMA step q contracts the node pair {base(q), base(q)+8}:
// with a row pitch of 132 floats the two halves of a wavefront read banks 32 apart -> conflict-free ds_read_b32.
constexpr int CT = 128, NC = 32, LDC = CT + 4;

template <bool VEC>
__global__ __launch_bounds__(256, 2) void nodes_contract_kernel(const float* __restrict__ Y, int ldY, int M,
                                                                const float* __restrict__ X, int F, int N, int K,
                                                                int chunks_per_range, float* __restrict__ C) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ys = lds;                    // [2][NC][LDC]
    float* xs = ys + 2 * NC * LDC;      // [2][NC][LDC]
    const int n_chunks = (N + NC - 1) / NC;
    const int nf = (F + CT - 1) / CT, nm = (M + CT - 1) / CT;
    // a = (node range, f tile): shares the X tile (and, over adjacent f tiles, the Y tile); b = (m tile, factor)
    const XcdItem item = xcd_item(blockIdx.x, ((n_chunks + chunks_per_range - 1) / chunks_per_range) * nf, nm * K);
    if (!item.valid) return;
    const int f0 = (item.a % nf) * CT, rng = item.a / nf;
    const int m0 = (item.b % nm) * CT, k = item.b / nm;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wf = wave & 1;
    const int chunk0 = rng * chunks_per_range;
    const int my_chunks = max(0, min(chunks_per_range, n_chunks - chunk0));
    const float* Yk = Y + (size_t)k * M + m0;

    TileStage<NC, CT, VEC, 256> yt, xt;
    auto fetch = [&](int c) {
        const int n0 = (chunk0 + c) * NC;
        yt.fetch(Yk + (size_t)n0 * ldY, ldY, N - n0, M - m0, tid);
        xt.fetch(X + (size_t)n0 * F + f0, F, N - n0, F - f0, tid);
    };
    auto stash = [&](int c) {
        yt.template stash<LDC>(ys + (c & 1) * NC * LDC, tid);
        xt.template stash<LDC>(xs + (c & 1) * NC * LDC, tid);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) zero_acc(acc[a][b]);
    if (my_chunks > 0) {
        fetch(0);
        stash(0);
        if (my_chunks > 1) fetch(1);
    }
    __syncthreads();
    for (int c = 0; c < my_chunks; ++c) {
        const float* yb = ys + (c & 1) * NC * LDC + wm * 64 + li;
        const float* xb = xs + (c & 1) * NC * LDC + wf * 64 + li;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int node = (q >> 3) * 16 + (q & 7) + 8 * half;
            const float a0 = yb[node * LDC], a1 = yb[node * LDC + 32];
            const float b0 = xb[node * LDC], b1 = xb[node * LDC + 32];
            DL_MFMA(acc[0][0], a0, b0);
            DL_MFMA(acc[0][1], a0, b1);
            DL_MFMA(acc[1][0], a1, b0);
            DL_MFMA(acc[1][1], a1, b1);
            if (q == 3) {                                       // staging in the shadow of the MFMAs
                if (c + 1 < my_chunks) stash(c + 1);
                if (c + 2 < my_chunks) fetch(c + 2);
            }
        }
        __syncthreads();
    }
    float* out = C + ((size_t)rng * K + k) * M * F;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int f = f0 + wf * 64 + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + a * 32 + acc_row(r, half);
                if (m < M && f < F) out[(size_t)m * F + f] = acc[a][b][r];
            }
        }
}

// Kernel B on the bf16 matrix path: the same contraction from tile-major plane arrays (dl_tiles.h) — Yp = planes of
// dhid_k^T ([hidden][node], written by kernel A), Xp = planes of x^T ([feature][node], dl_planes.hip), both in
// 16-node tiles.  Six exact bf16 products per term (fp32-grade accuracy) at a multiple of the fp32 MFMA rate; the
// tiles are straight copies (no masks: the arrays are zero-filled along the node axis, and rows past M / F only
// reach output rows / columns that are not stored).  Node chunk of 16 per step, double-buffered, 2 workgroups per CU.
constexpr int PC = 16, PPITCH = PC + 8;

// -DDL_PROJB_STAMPS=<workgroup index>: DIAGNOSTIC build (like DL_PROJA_STAMPS): s_memtime at the phase boundaries of every
// node chunk, waves 0 and 2 of one workgroup of kernel B; read back by dl_debug_read_stamps_b (tools/projb_stamps.py).
// What was tried on this kernel from its stamps and counters (profiles/r6_kernel_b_experiments.txt): a dictated issue order
// (sched_group_barrier: staging instructions between the MFMAs), global loads two chunks ahead, two chunks per barrier,
// 512 instead of 256 workgroups, every tile a cache hit (timing bound): all within 35 +- 3 us.
#ifdef DL_PROJB_STAMPS
__device__ unsigned long long dl_projb_stamps[2][512];
#define DLB_STAMP(code)                                                                                 \
    do {                                                                                                \
        if (stamp_on && stamp_n < 510) {                                                                \
            dl_projb_stamps[stamp_w][stamp_n++] = ((unsigned long long)(code) << 56) | (__builtin_amdgcn_s_memtime() & 0x00FFFFFFFFFFFFFFull); \
            dl_projb_stamps[stamp_w][511] = stamp_n;                                                    \
        }                                                                                               \
    } while (0)
#else
#define DLB_STAMP(code) do {} while (0)
#endif

__global__ __launch_bounds__(256, 2) void nodes_contract_planes_kernel(const __bf16* __restrict__ Yp, size_t y_batch,
                                                                       const __bf16* __restrict__ Xp, int ncb, int n_chunks,
                                                                       int M, int F, int K, int chunks_per_range,
                                                                       float* __restrict__ C) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __bf16* ys = reinterpret_cast<__bf16*>(lds);        // [2][3][128][PPITCH]
    __bf16* xs = ys + 2 * 3 * PLANE_ROWS * PPITCH;
    const int nf = (F + CT - 1) / CT, nm = (M + CT - 1) / CT;
    const XcdItem item = xcd_item(blockIdx.x, ((n_chunks + chunks_per_range - 1) / chunks_per_range) * nf, nm * K);
    if (!item.valid) return;
    const int fb = item.a % nf, rng = item.a / nf;
    const int mb = item.b % nm, k = item.b / nm;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wf = wave & 1;
    const int chunk0 = rng * chunks_per_range;
    const int my_chunks = max(0, min(chunks_per_range, n_chunks - chunk0));
    const __bf16* Yk = Yp + (size_t)k * y_batch;
#ifdef DL_PROJB_STAMPS
    const bool stamp_on = (int)blockIdx.x == DL_PROJB_STAMPS && (wave == 0 || wave == 2) && lane == 0;
    const int stamp_w = wave >> 1;
    int stamp_n = 0;
#endif
    DLB_STAMP(1);

    PlaneStage<256, PC> yq, xq;
    auto fetch = [&](int c) {
        yq.fetch(Yk + plane_tile<PC>(mb, chunk0 + c, ncb), tid);
        xq.fetch(Xp + plane_tile<PC>(fb, chunk0 + c, ncb), tid);
    };
    auto stash = [&](int c) {
        yq.stash(ys + (c & 1) * 3 * PLANE_ROWS * PPITCH, tid);
        xq.stash(xs + (c & 1) * 3 * PLANE_ROWS * PPITCH, tid);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) zero_acc(acc[a][b]);
    if (my_chunks > 0) {
        fetch(0);
        stash(0);
        fetch(min(1, my_chunks - 1));
    }
    __syncthreads();
    DLB_STAMP(2);
    for (int c = 0; c < my_chunks; ++c) {
        DLB_STAMP(10);
        // lane half h supplies nodes 8h .. 8h+7 of the chunk: A = hidden rows of this wave (2 tiles), B = feature rows
        const __bf16* yb = ys + (c & 1) * 3 * PLANE_ROWS * PPITCH + (wm * 64 + li) * PPITCH + half * 8;
        const __bf16* xb = xs + (c & 1) * 3 * PLANE_ROWS * PPITCH + (wf * 64 + li) * PPITCH + half * 8;
        bf16x8 a0[3], a1[3], b0[3], b1[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            a0[p] = *reinterpret_cast<const bf16x8*>(yb + p * PLANE_ROWS * PPITCH);
            a1[p] = *reinterpret_cast<const bf16x8*>(yb + (p * PLANE_ROWS + 32) * PPITCH);
            b0[p] = *reinterpret_cast<const bf16x8*>(xb + p * PLANE_ROWS * PPITCH);
            b1[p] = *reinterpret_cast<const bf16x8*>(xb + (p * PLANE_ROWS + 32) * PPITCH);
        }
        mfma_split6(acc[0][0], a0, b0);
        mfma_split6(acc[0][1], a0, b1);
        DLB_STAMP(11);
        if (c + 1 < my_chunks) stash(c + 1);                    // staging in the shadow of the MFMAs
        DLB_STAMP(12);
        fetch(min(c + 2, my_chunks - 1));                       // unconditional: see project2_fwd_kernel
        mfma_split6(acc[1][0], a1, b0);
        mfma_split6(acc[1][1], a1, b1);
        DLB_STAMP(13);
        __syncthreads();
        DLB_STAMP(14);
    }
    float* out = C + ((size_t)rng * K + k) * M * F;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int f = fb * CT + wf * 64 + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb * CT + wm * 64 + a * 32 + acc_row(r, half);
                if (m < M && f < F) out[(size_t)m * F + f] = acc[a][b][r];
            }
        }
    DLB_STAMP(20);
}

// out[i] (+)= sum_s slabs[s][i], s ascending (fixed order), for up to 4 independent (slabs, out) jobs in one
// launch (the four gradients of a backward); accumulate: on top of what out holds (node blocks).
struct SlabJob { const float* slabs; float* out; size_t n; int S; unsigned block0; };
struct SlabJobs { SlabJob j[4]; int count; };
__global__ __launch_bounds__(256) void slab_sum_kernel(SlabJobs jobs, int accumulate) {
    int q = 0;
#pragma unroll
    for (int t = 1; t < 4; ++t)
        if (t < jobs.count && blockIdx.x >= jobs.j[t].block0) q = t;
    const SlabJob job = jobs.j[q];
    const size_t i = (size_t)(blockIdx.x - job.block0) * 256 + threadIdx.x;
    if (i >= job.n) return;
    float v = accumulate ? job.out[i] : 0.0f;
    const float* p = job.slabs + i;
    int s = 0;
    for (; s + 4 <= job.S; s += 4) {                            // four loads in flight, the same order of additions
        const float a0 = p[(size_t)s * job.n], a1 = p[(size_t)(s + 1) * job.n], a2 = p[(size_t)(s + 2) * job.n],
                    a3 = p[(size_t)(s + 3) * job.n];
        v = (((v + a0) + a1) + a2) + a3;
    }
    for (; s < job.S; ++s) v += p[(size_t)s * job.n];
    job.out[i] = v;
}

// Column sums of a row-major [N][C] matrix over S row ranges -> part[S][C].  Block = 64 columns x 4 row lanes.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, int N, int C, int rows_per_range,
                                                     float* __restrict__ part) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_range, r1 = min(N, r0 + rows_per_range);
    float v = 0.0f;
    if (c < C) {
        int r = r0 + rl;
        for (; r + 12 < r1; r += 16) {                          // four loads in flight, the same order of additions
            const float a0 = A[(size_t)r * C + c], a1 = A[(size_t)(r + 4) * C + c], a2 = A[(size_t)(r + 8) * C + c],
                        a3 = A[(size_t)(r + 12) * C + c];
            v = (((v + a0) + a1) + a2) + a3;
        }
        for (; r < r1; r += 4) v += A[(size_t)r * C + c];
    }
    red[rl][cl] = v;
    __syncthreads();
    if (rl == 0 && c < C) part[(size_t)blockIdx.y * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// ---------------------------------------------------------------- host side: ranges, workspace, launches
struct BwdLayout {
    int sA, tiles_per_range;       // kernel A node ranges
    int sB, chunks_per_range;      // kernel B node ranges
    int sC, rows_per_range;        // colsum ranges
    int Mb;                        // rows of the kernel-B output per factor (nhid, or d for one layer)
    size_t off_dhid, off_xT, off_w1p, off_w2p, off_b1p, off_b2p, bytes;
    bool planes;                   // dhid and x^T as bf16 plane arrays (kernel B on the bf16 matrix path)
    int ncb, n_chunks16;           // 16-node chunks per row block of the plane arrays / chunks that hold nodes
};

static int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// Node block of the two-layer backward: the masked hidden gradient of one block ([rows][K][nhid] fp32) is capped
// at 1 GiB, so the workspace does not grow with the graph (2.9M nodes x K=8 x 512 would be 48 GB); the blocks
// are processed in order and accumulate into the gradients.
static int bwd_block_rows(int N, int K, int nhid, bool two_layer) {
    if (!two_layer) return N;
    long long cap_bytes = 1LL << 30;
    if (config().bwd_block_bytes > 0) cap_bytes = config().bwd_block_bytes;                 // DL_BWD_BLOCK_BYTES (tests: force blocking)
    const long long cap = cap_bytes / ((long long)K * nhid * (split_products() ? 6 : 4));
    const long long rows = std::max<long long>(4096, cap / TILE_N * TILE_N);
    return (int)std::min<long long>(N, rows);
}

static BwdLayout bwd_layout(int N, int F, int K, int nhid, int d, bool two_layer, bool blocked = false,
                            bool recompute = true) {
    BwdLayout L{};
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // Node ranges per launch: enough workgroups to cover the 256 CUs (>= 256, at most ~1536 so the partial slabs
    // stay small), but never so many that a workgroup runs fewer than ~24 pipeline steps — its prologue (W2^T
    // staging, first tile) and the slab it writes are per-workgroup costs.  Measured (DL_BWD_TARGET sweeps):
    // narrow features want few long ranges, wide features many short ones.
    auto pick = [](int n_units, int steps_per_unit, long long wg_per_range, int min_steps) {
        if (config().bwd_target > 0)                                       // DL_BWD_TARGET: tuning knob, workgroups per launch
            return (int)std::max(1LL, std::min<long long>(n_units, ceil_div(config().bwd_target, wg_per_range)));
        const long long lo = ceil_div(256, wg_per_range), hi = ceil_div(1536, wg_per_range);
        const long long by_steps = n_units / std::max(1, ceil_div(min_steps, steps_per_unit));
        return (int)std::max(1LL, std::min<long long>(n_units, std::min(hi, std::max(lo, by_steps))));
    };
    L.planes = two_layer && split_products();
    L.ncb = plane_chunks<PC>(N, PLANE_ROWS);
    L.n_chunks16 = ceil_div(N, PC);
    const int n_tiles = ceil_div(N, TILE_N), n_chunks = L.planes ? L.n_chunks16 : ceil_div(N, NC);
    L.Mb = two_layer ? nhid : d;
    const long long wgA = (long long)ceil_div(nhid, 64 * bwd_ht(d)) * K;
    L.sA = pick(n_tiles, recompute ? ceil_div(F, BFC) : 3, wgA, 24);
    L.tiles_per_range = ceil_div(n_tiles, L.sA);
    L.sA = ceil_div(n_tiles, L.tiles_per_range);
    const long long wgB = (long long)ceil_div(L.Mb, CT) * ceil_div(F, CT) * K;
    L.sB = pick(n_chunks, 1, wgB, L.planes ? 40 : 20);
    L.chunks_per_range = ceil_div(n_chunks, L.sB);
    L.sB = ceil_div(n_chunks, L.chunks_per_range);
    const int colblocks = ceil_div((long long)K * d, 64);
    L.sC = (int)std::max(1LL, std::min<long long>(ceil_div(N, 64), ceil_div(768, colblocks)));
    L.rows_per_range = ceil_div(N, L.sC);
    L.sC = ceil_div(N, L.rows_per_range);
    size_t off = 0;
    L.off_dhid = off;
    if (L.planes) {
        off += al(sizeof(__bf16) * K * plane_array_elems(nhid, N, PLANE_ROWS));
        L.off_xT = off;
        off += al(sizeof(__bf16) * plane_array_elems(F, N, PLANE_ROWS));
    } else {
        off += two_layer ? al(sizeof(float) * (size_t)N * K * nhid) : 0;
    }
    L.off_w1p = off;  off += (L.sB > 1 || blocked) ? al(sizeof(float) * (size_t)L.sB * K * L.Mb * F) : 0;
    L.off_w2p = off;  off += two_layer ? al(sizeof(float) * (size_t)L.sA * K * d * nhid) : 0;
    L.off_b1p = off;  off += two_layer ? al(sizeof(float) * (size_t)L.sA * K * nhid) : 0;
    L.off_b2p = off;  off += al(sizeof(float) * (size_t)L.sC * K * d);
    L.bytes = off;
    return L;
}

}  // namespace project

#ifdef DL_PROJA_STAMPS
}  // namespace dl
extern "C" int dl_debug_read_stamps_a(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dl::project::dl_proja_stamps), sizeof(unsigned long long) * 2 * 512);
}
namespace dl {
#endif
#ifdef DL_PROJB_STAMPS
}  // namespace dl
extern "C" int dl_debug_read_stamps_b(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dl::project::dl_projb_stamps), sizeof(unsigned long long) * 2 * 512);
}
namespace dl {
#endif

size_t project_bwd_workspace_bytes(int N, int F, int K, int nhid, int d, bool two_layer) {
    if (N <= 0) return 0;
    using namespace project;
    const int R = bwd_block_rows(N, K, nhid, two_layer);
    const bool blocked = R < N;
    size_t bytes = 0;
    for (int recompute = 0; recompute < 2; ++recompute) {       // either form of the backward fits
        bytes = std::max(bytes, bwd_layout(R, F, K, nhid, d, two_layer, blocked, recompute != 0).bytes);
        if (blocked && N % R != 0)
            bytes = std::max(bytes, bwd_layout(N % R, F, K, nhid, d, two_layer, true, recompute != 0).bytes);
    }
    return bytes;
}

template <int D, bool VEC, bool RECOMPUTE, bool PLANES>
static void launchA_t(dim3 grid, hipStream_t st, const float* x, int N, int F, int nhid, const float* W1,
                      const float* b1, const float* W2, const float* dZ, int K, int tpr, float* dhid, float* dW2p,
                      float* db1p, const float* hidT, int ldh, int hid_cols, project::DhidPlanes dhp) {
    using namespace project;
    static unsigned long long lds_done = 0;
    constexpr size_t lds = project2_bwd_lds(D);
    ensure_dynamic_lds(reinterpret_cast<const void*>(&project2_bwd_hidden_kernel<D, VEC, RECOMPUTE, PLANES>), lds, lds_done);
    hipLaunchKernelGGL((project2_bwd_hidden_kernel<D, VEC, RECOMPUTE, PLANES>), grid, dim3(BTHR), lds, st, x, N, F, nhid, W1,
                       b1, W2, dZ, K, tpr, dhid, dW2p, db1p, hidT, ldh, hid_cols, dhp);
}

template <int D, bool VEC, bool RECOMPUTE>
static void launchA_p(bool planes, dim3 grid, hipStream_t st, const float* x, int N, int F, int nhid, const float* W1,
                      const float* b1, const float* W2, const float* dZ, int K, int tpr, float* dhid, float* dW2p,
                      float* db1p, const float* hidT, int ldh, int hid_cols, project::DhidPlanes dhp) {
    if (planes) launchA_t<D, VEC, RECOMPUTE, true>(grid, st, x, N, F, nhid, W1, b1, W2, dZ, K, tpr, dhid, dW2p, db1p, hidT, ldh,
                                                   hid_cols, dhp);
    else launchA_t<D, VEC, RECOMPUTE, false>(grid, st, x, N, F, nhid, W1, b1, W2, dZ, K, tpr, dhid, dW2p, db1p, hidT, ldh,
                                             hid_cols, dhp);
}

struct SlabBatch {
    project::SlabJobs jobs{};
    unsigned blocks = 0;
    void add(const float* slabs, int S, size_t n, float* out) {
        jobs.j[jobs.count++] = project::SlabJob{slabs, out, n, S, blocks};
        blocks += (unsigned)((n + 255) / 256);
    }
    void run(bool accumulate, hipStream_t st) {
        if (jobs.count) hipLaunchKernelGGL(project::slab_sum_kernel, dim3(blocks), dim3(256), 0, st, jobs, accumulate ? 1 : 0);
    }
};

// One node block [row0, row0 + N) of the backward; acc: add to the gradients instead of overwriting them.
static void project_bwd_block(const float* x, int N, int F, int K, int nhid, int d, const float* W1, const float* b1,
                              const float* W2, const float* dZ, const float* hidT, int ldh, int hid_cols, float* dW1,
                              float* db1, float* dW2, float* db2, void* ws, bool blocked, bool acc, hipStream_t st,
                              const void* xT_planes = nullptr) {
    using namespace project;
    const bool two = W2 != nullptr;
    const BwdLayout L = bwd_layout(N, F, K, nhid, d, two, blocked, hidT == nullptr);
    char* base = static_cast<char*>(ws);
    float* dhid = reinterpret_cast<float*>(base + L.off_dhid);
    float* w1p = reinterpret_cast<float*>(base + L.off_w1p);
    float* w2p = reinterpret_cast<float*>(base + L.off_w2p);
    float* b1p = reinterpret_cast<float*>(base + L.off_b1p);
    float* b2p = reinterpret_cast<float*>(base + L.off_b2p);
    const bool vecA = F % 4 == 0;                                   // kernel A: x and W1 rows
    const bool vecB = F % 4 == 0 && L.Mb % 4 == 0;                  // kernel B: Y rows (stride K*Mb) and x rows

    // bias of the output layer: column sums of dZ [N][K*d]
    float* dbo = two ? db2 : db1;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)ceil_div((long long)K * d, 64), (unsigned)L.sC), dim3(256), 0, st,
                       dZ, N, K * d, L.rows_per_range, b2p);
    SlabBatch sums;                                                 // all slab sums of this block in one launch, at the end
    sums.add(b2p, L.sC, (size_t)K * d, dbo);

    const float* Y = dZ;
    int ldY = K * d;
    __bf16* dhP = reinterpret_cast<__bf16*>(base + L.off_dhid);
    __bf16* xTP = reinterpret_cast<__bf16*>(base + L.off_xT);
    const DhidPlanes dhp{dhP, plane_array_elems(nhid, N, PLANE_ROWS), L.ncb};
    if (two) {
        const dim3 grid((unsigned)xcd_grid(L.sA, ceil_div(nhid, 64 * bwd_ht(d)) * K));
#define DL_PA(DD)                                                                                               \
    if (d == DD) {                                                                                              \
        if (hidT) launchA_p<DD, true, false>(L.planes, grid, st, x, N, F, nhid, W1, b1, W2, dZ, K, L.tiles_per_range, dhid,   \
                                             w2p, b1p, hidT, ldh, hid_cols, dhp);                                       \
        else if (vecA) launchA_p<DD, true, true>(L.planes, grid, st, x, N, F, nhid, W1, b1, W2, dZ, K, L.tiles_per_range,    \
                                                 dhid, w2p, b1p, nullptr, 0, 0, dhp);                                   \
        else launchA_p<DD, false, true>(L.planes, grid, st, x, N, F, nhid, W1, b1, W2, dZ, K, L.tiles_per_range, dhid, w2p,  \
                                        b1p, nullptr, 0, 0, dhp);                                                       \
    }
        DL_PA(32) DL_PA(64) DL_PA(128)
#undef DL_PA
        sums.add(w2p, L.sA, (size_t)K * d * nhid, dW2);
        sums.add(b1p, L.sA, (size_t)K * nhid, db1);
        Y = dhid;
        ldY = K * nhid;
    }
    if (L.planes) {
        static unsigned long long lds_done_p = 0;
        const size_t lds = sizeof(__bf16) * 2 * 2 * 3 * PLANE_ROWS * PPITCH;
        ensure_dynamic_lds(reinterpret_cast<const void*>(&nodes_contract_planes_kernel), lds, lds_done_p);
        if (xT_planes) xTP = const_cast<__bf16*>(static_cast<const __bf16*>(xT_planes));     // split once for the run
        else split_transposed(x, N, F, F, xTP, st);
        const dim3 grid((unsigned)xcd_grid(L.sB * ceil_div(F, CT), ceil_div(L.Mb, CT) * K));
        const bool direct = L.sB == 1 && !blocked;
        float* out = direct ? dW1 : w1p;
        hipLaunchKernelGGL(nodes_contract_planes_kernel, grid, dim3(256), lds, st, dhP, dhp.batch, xTP, L.ncb, L.n_chunks16,
                           L.Mb, F, K, L.chunks_per_range, out);
        if (!direct) sums.add(w1p, L.sB, (size_t)K * L.Mb * F, dW1);
    } else {
        static unsigned long long lds_done_v = 0, lds_done_s = 0;
        const size_t lds = sizeof(float) * 4 * NC * LDC;
        ensure_dynamic_lds(reinterpret_cast<const void*>(&nodes_contract_kernel<true>), lds, lds_done_v);
        ensure_dynamic_lds(reinterpret_cast<const void*>(&nodes_contract_kernel<false>), lds, lds_done_s);
        const dim3 grid((unsigned)xcd_grid(L.sB * ceil_div(F, CT), ceil_div(L.Mb, CT) * K));
        const bool direct = L.sB == 1 && !blocked;                  // one range, one block: straight into dW1
        float* out = direct ? dW1 : w1p;
        if (vecB) hipLaunchKernelGGL(nodes_contract_kernel<true>, grid, dim3(256), lds, st, Y, ldY, L.Mb, x, F, N, K,
                                    L.chunks_per_range, out);
        else hipLaunchKernelGGL(nodes_contract_kernel<false>, grid, dim3(256), lds, st, Y, ldY, L.Mb, x, F, N, K,
                                L.chunks_per_range, out);
        if (!direct) sums.add(w1p, L.sB, (size_t)K * L.Mb * F, dW1);
    }
    sums.run(acc, st);
}

int project_bwd(const float* x, int N, int F, int K, int nhid, int d, const float* W1, const float* b1,
                const float* W2, const float* dZ, const float* hid, float* dW1, float* db1, float* dW2, float* db2,
                void* ws, hipStream_t st, const void* xplanes) {
    const bool two = W2 != nullptr;
    const int R = project::bwd_block_rows(N, K, nhid, two);
    const bool blocked = R < N;
    if (!blocked && xplanes && two && split_products()) {     // one block: the persistent x^T planes serve it
        project_bwd_block(x, N, F, K, nhid, d, W1, b1, W2, dZ, hid, (N + 3) & ~3, (N + 3) & ~3, dW1, db1, dW2, db2, ws, false,
                          false, st, project_xplanes_xT(xplanes, N, F));
        return check_launch("project_bwd");
    }
    const int ldh = (N + 3) & ~3;                               // row stride of the kept hidden layer hidT [K][nhid][ldh]
    for (int row0 = 0; row0 < N; row0 += R)
        project_bwd_block(x + (size_t)row0 * F, std::min(R, N - row0), F, K, nhid, d, W1, b1, W2,
                          dZ + (size_t)row0 * K * d, (two && hid) ? hid + row0 : nullptr, ldh, ldh - row0, dW1, db1, dW2,
                          db2, ws, blocked, row0 > 0, st);
    return check_launch("project_bwd");
}

}  // namespace dl
