// fp32 matrix -> three bf16 planes (x = hi + mid + lo, dl_tiles.h) in global memory, tile-major and zero-filled out
// to the padded extents, for the projection kernels that run fp32-grade products on the bf16 matrix path.
#include "dl_common.h"
#include "dl_kernels.h"
#include "dl_tiles.h"

namespace dl {
namespace project {

// One thread per 16-byte piece (8 consecutive columns of a row); consecutive threads walk a tile in storage order,
// so the writes are contiguous and the reads are whole 128-byte row segments.
struct RowsJob { const float* src; int R, C, ld; size_t sb; __bf16* dst; int nrb, ncb; size_t db; };
__device__ __forceinline__ void split_rows_body(const RowsJob& j, size_t g, int batch) {
    const float* __restrict__ src = j.src;
    __bf16* __restrict__ dst = j.dst;
    const int R = j.R, C = j.C, ld = j.ld, nrb = j.nrb, ncb = j.ncb;
    const size_t sb = j.sb, db = j.db;
    constexpr int PLANE_TILE = PLANE_ROWS * SPLIT_COLS, PIECES = PLANE_TILE / 8;
    if (g >= (size_t)nrb * ncb * PIECES) return;
    const int t = (int)(g / PIECES), q = (int)(g % PIECES);
    const int rb = t / ncb, cb = t % ncb;
    const int r = rb * PLANE_ROWS + (q >> 2), c0 = cb * SPLIT_COLS + (q & 3) * 8;
    const float* s = src + (size_t)batch * sb + (size_t)r * ld;
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bool ok = r < R && c0 + e < C;
        const float x = ok ? s[c0 + e] : 0.0f;
        __bf16 h, m, l;
        split3(x, h, m, l);
        p0[e] = h; p1[e] = m; p2[e] = l;
    }
    __bf16* o = dst + (size_t)batch * db + plane_tile<SPLIT_COLS>(rb, cb, ncb) + q * 8;
    *reinterpret_cast<bf16x8*>(o) = p0;
    *reinterpret_cast<bf16x8*>(o + PLANE_TILE) = p1;
    *reinterpret_cast<bf16x8*>(o + 2 * PLANE_TILE) = p2;
}
__global__ __launch_bounds__(256) void split_rows_kernel(RowsJob j) {
    split_rows_body(j, (size_t)blockIdx.x * 256 + threadIdx.x, blockIdx.y);
}

// Planes of the transpose: rows = c (columns of src), columns = r.  Block = a 64 x 64 tile of src through LDS:
// coalesced reads along c, 16-byte pieces along r on the way out.  grid (ceil(Rp/64), ceil(Cp/64)) over the PADDED extents.
__global__ __launch_bounds__(256) void split_transpose_kernel(const float* __restrict__ src, int R, int C, int ld,
                                                              __bf16* __restrict__ dst, int ncb) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + 4 * i + tr, c = c0 + tc;
        tile[4 * i + tr][tc] = (r < R && c < C) ? src[(size_t)r * ld + c] : 0.0f;
    }
    __syncthreads();
    const int r8 = (threadIdx.x & 7) * 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (threadIdx.x >> 3) + 32 * j;
        bf16x8 p0, p1, p2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 h, m, l;
            split3(tile[r8 + e][c], h, m, l);
            p0[e] = h; p1[e] = m; p2[e] = l;
        }
        const int row = c0 + c, col = r0 + r8;                          // of the transposed matrix
        constexpr int TC = 16;                                          // tile width of the transposed arrays
        __bf16* o = dst + plane_tile<TC>(row / PLANE_ROWS, col / TC, ncb) + (row % PLANE_ROWS) * TC + col % TC;
        *reinterpret_cast<bf16x8*>(o) = p0;
        *reinterpret_cast<bf16x8*>(o + PLANE_ROWS * TC) = p1;
        *reinterpret_cast<bf16x8*>(o + 2 * PLANE_ROWS * TC) = p2;
    }
}

// W2 [rows = K*D][nhid] -> planes [3][rows][nhid_p] for layer 2 of the forward, whose B operand is the layer-1
// accumulator itself: register r of lane half t holds hidden row 8*(r>>2) + 4t + (r&3) of its 32-row tile, so the 8
// k-slots a lane half supplies to one K = 16 block b are the hidden units 16b + 8*(s>>2) + 4t + (s&3), s = 0..7.
// The planes are stored in that order (position 16b + 8t + s within every group of 32 hidden units), so the A operand
// of the block is one 16-byte load per plane.  Hidden units past nhid are zero.
struct W2Job { const float* W2; int rows, nhid; __bf16* dst; int nhid_p; size_t ps; };
__device__ __forceinline__ void split_w2_body(const W2Job& j, size_t g) {
    const float* __restrict__ W2 = j.W2;
    __bf16* __restrict__ dst = j.dst;
    const int rows = j.rows, nhid = j.nhid, nhid_p = j.nhid_p;
    const size_t ps = j.ps;
    const int groups = nhid_p / 8;
    if (g >= (size_t)rows * groups) return;
    const int row = (int)(g / groups), pos = (int)(g % groups) * 8;
    const int g32 = pos / 32, b = (pos % 32) / 16, t = (pos % 16) / 8;
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int h = 32 * g32 + 16 * b + 8 * (s >> 2) + 4 * t + (s & 3);
        const float x = h < nhid ? W2[(size_t)row * nhid + h] : 0.0f;
        __bf16 hi, mid, lo;
        split3(x, hi, mid, lo);
        p0[s] = hi; p1[s] = mid; p2[s] = lo;
    }
    __bf16* o = dst + (size_t)row * nhid_p + pos;
    *reinterpret_cast<bf16x8*>(o) = p0;
    *reinterpret_cast<bf16x8*>(o + ps) = p1;
    *reinterpret_cast<bf16x8*>(o + 2 * ps) = p2;
}

__global__ __launch_bounds__(256) void split_w2_kernel(W2Job j) { split_w2_body(j, (size_t)blockIdx.x * 256 + threadIdx.x); }

// The forward's three operand splits in ONE launch (x, the K matrices W1_k, W2): the jobs are a few microseconds each,
// i.e. mostly launch and drain.  Block ranges: [0, bx) x, [bx, bx + K*bw) W1 (batch-major), the rest W2.
__global__ __launch_bounds__(256) void split_fwd_operands_kernel(RowsJob x, unsigned bx, RowsJob w, unsigned bw, unsigned Kw,
                                                                 W2Job w2) {
    const unsigned b = blockIdx.x;
    if (b < bx) {
        split_rows_body(x, (size_t)b * 256 + threadIdx.x, 0);
    } else if (b < bx + Kw * bw) {
        const unsigned r = b - bx;
        split_rows_body(w, (size_t)(r % bw) * 256 + threadIdx.x, (int)(r / bw));
    } else {
        split_w2_body(w2, (size_t)(b - bx - Kw * bw) * 256 + threadIdx.x);
    }
}

static W2Job w2_job(const float* W2, int rows, int nhid, __bf16* dst, int nhid_p) {
    return W2Job{W2, rows, nhid, dst, nhid_p, (size_t)rows * nhid_p};
}
static RowsJob rows_job(const float* src, int R, int C, int ld, size_t sb, __bf16* dst) {
    const int nrb = (int)(round_up(R, PLANE_ROWS) / PLANE_ROWS), ncb = plane_chunks<SPLIT_COLS>(C, SPLIT_COLS);
    return RowsJob{src, R, C, ld, sb, dst, nrb, ncb, plane_array_elems(R, C, SPLIT_COLS)};
}
static unsigned rows_blocks(const RowsJob& j) {
    return (unsigned)(((size_t)j.nrb * j.ncb * (PLANE_ROWS * SPLIT_COLS / 8) + 255) / 256);
}
static unsigned w2_blocks(const W2Job& j) { return (unsigned)(((size_t)j.rows * (j.nhid_p / 8) + 255) / 256); }

void split_w2(const float* W2, int rows, int nhid, __bf16* dst, int nhid_p, hipStream_t st) {
    const W2Job j = w2_job(W2, rows, nhid, dst, nhid_p);
    hipLaunchKernelGGL(split_w2_kernel, dim3(w2_blocks(j)), dim3(256), 0, st, j);
}

void split_rows(const float* src, int B, int R, int C, int ld, size_t sb, __bf16* dst, hipStream_t st) {
    const RowsJob j = rows_job(src, R, C, ld, sb, dst);
    hipLaunchKernelGGL(split_rows_kernel, dim3(rows_blocks(j), (unsigned)B), dim3(256), 0, st, j);
}

void split_fwd_operands(const float* x, int N, int F, __bf16* xP, const float* W1, int K, int nhid, __bf16* wP,
                        const float* W2, int d, __bf16* w2P, int nhid_p, hipStream_t st) {
    const RowsJob jx = rows_job(x, N, F, F, 0, xP), jw = rows_job(W1, nhid, F, F, (size_t)nhid * F, wP);
    const W2Job j2 = w2_job(W2, K * d, nhid, w2P, nhid_p);
    const unsigned bx = x ? rows_blocks(jx) : 0u, bw = rows_blocks(jw), b2 = w2_blocks(j2);   // x == NULL: its planes exist already
    hipLaunchKernelGGL(split_fwd_operands_kernel, dim3(bx + (unsigned)K * bw + b2), dim3(256), 0, st, jx, bx, jw, bw,
                       (unsigned)K, j2);
}

void split_transposed(const float* src, int R, int C, int ld, __bf16* dst, hipStream_t st) {
    // transposed matrix: C rows, R columns, both padded to 128
    const int rows_p = (int)round_up(C, PLANE_ROWS), cols_p = (int)round_up(R, PLANE_ROWS);
    hipLaunchKernelGGL(split_transpose_kernel, dim3((unsigned)(cols_p / 64), (unsigned)(rows_p / 64)), dim3(256), 0, st, src, R,
                       C, ld, dst, cols_p / 16);
}

}  // namespace project
}  // namespace dl
