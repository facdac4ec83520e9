// Tile staging and MFMA helpers shared by the projection kernels (dl_project.hip, dl_project_bwd.hip).
//
// fp32 matrix-core products use v_mfma_f32_32x32x2_f32: a 32x32 accumulator keeps its COLUMN on the
// lane (lane & 31) and 16 of its rows in the registers — register r of lane half h (lane >> 5) is row
// acc_row(r, h).  Operand A supplies A[row = lane & 31][k = half], operand B supplies B[k = half][col = lane & 31].
#pragma once
#include <hip/hip_runtime.h>

namespace dl {
namespace project {

// Kernels with more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once per
// device (the attribute lives with the device's code object): `done` is the caller's per-kernel bitmask of devices.
inline void ensure_dynamic_lds(const void* kernel, size_t bytes, unsigned long long& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 63;
    if (done & (1ull << dev)) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (dev != 63) done |= 1ull << dev;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ void zero_acc(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.0f;
}

// x where m is all ones, +0.0f where m is 0 — exact for any x (a multiply by 0 would turn inf / NaN garbage
// into NaN), and not something the compiler can turn back into a branch around the load that produced x.
__device__ __forceinline__ float mask_bits(float x, unsigned m) { return __uint_as_float(__float_as_uint(x) & m); }

#define DL_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (acc), 0, 0, 0)

// XCD-aware work assignment for a 1-D grid whose work items form an (n_a x n_b) rectangle — a = node tile,
// b = weight chunk.  Workgroup h runs on XCD h % 8 (round-robin dispatch), and each XCD has its own L2: so
// the 32 workgroups an XCD starts together should share operands.  The items are put in a blocked order
// (blocks of 4 a-values x 8 b-values, edge blocks smaller: 12 distinct operand tiles per 32 items instead of
// up to 64) and dealt to the XCDs in runs of 32 consecutive items, so every XCD gets the same amount of work.
// Workgroups past the last item return at once.  grid = xcd_grid(n_a, n_b).
struct XcdItem { int a, b; bool valid; };
__host__ __device__ inline int xcd_grid(int n_a, int n_b) { return (n_a * n_b + 255) / 256 * 256; }
__device__ __forceinline__ XcdItem xcd_item(int h, int n_a, int n_b) {
    const int xcd = h & 7, j = h >> 3;
    const int i = (j >> 5) * 256 + xcd * 32 + (j & 31);         // position in the blocked order
    XcdItem it;
    it.valid = i < n_a * n_b;
    const int bg = i / (n_a * 8), r = i - bg * n_a * 8;
    const int bw = min(8, n_b - bg * 8);                        // width of this b block (>= 1 while valid)
    const int ag = r / (4 * max(bw, 1)), r2 = r - ag * 4 * bw;
    const int aw = max(1, min(4, n_a - ag * 4));
    it.a = ag * 4 + r2 % aw;
    it.b = bg * 8 + r2 / aw;
    return it;
}

// A [ROWS][COLS] tile of a row-major matrix (row stride ld floats) on its way global -> registers -> LDS
// (row pitch PITCH floats), split so that the global loads can be issued before a long MFMA chain and the
// LDS stores after it.  `origin` points at the tile's first element (always a valid element);
// rows_valid / cols_valid = how much of the tile lies inside the matrix (the rest reads as zero).
//   VEC:  every row start and the matrix width are multiples of 4 floats (16-byte aligned quads that are
//         fully inside or fully outside): one dwordx4 load per quad.
//   else: scalar loads, consecutive lanes on consecutive floats (fully coalesced; rows of odd length have
//         no common alignment).
// Offsets are 32-bit from a wave-uniform origin (saddr + voffset addressing, no 64-bit VALU math).  Full
// tiles take an unchecked path; edge tiles load from clamped, always valid addresses and select afterwards
// (a branch around a load would make hipcc wait for each load separately).
template <int ROWS, int COLS, bool VEC, int THREADS>
struct TileStage {
    static constexpr int NV = ROWS * COLS / THREADS;   // floats per thread
    static_assert(ROWS * COLS % (4 * THREADS) == 0, "tile does not divide over the workgroup");
    float v[NV];
    int rows_valid, cols_valid;                        // of the tile in flight (wave-uniform)

    // Issue the loads only.  Nothing here consumes a loaded value: the zero fill of the part outside the
    // matrix happens in stash(), so the s_waitcnt for these loads lands after the MFMA chain in between.
    __device__ __forceinline__ void fetch(const float* __restrict__ origin, int ld, int rv, int cv, int tid) {
        rows_valid = rv;
        cols_valid = cv;
        const bool full = rv >= ROWS && cv >= COLS;
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < NV / 4; ++j) {
                const int i = tid + THREADS * j, r = i / (COLS / 4), c = 4 * (i % (COLS / 4));
                const bool ok = full || (r < rv && c < cv);
                const float4 q = *reinterpret_cast<const float4*>(origin + (ok ? (unsigned)(r * ld + c) : 0u));
                v[4 * j + 0] = q.x;
                v[4 * j + 1] = q.y;
                v[4 * j + 2] = q.z;
                v[4 * j + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = tid + THREADS * j, r = i / COLS, c = i % COLS;
                const bool ok = full || (r < rv && c < cv);
                v[j] = origin[ok ? (unsigned)(r * ld + c) : 0u];
            }
        }
    }
    template <int PITCH>
    __device__ __forceinline__ void stash(float* lds, int tid) const {
        const bool full = rows_valid >= ROWS && cols_valid >= COLS;
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < NV / 4; ++j) {
                const int i = tid + THREADS * j, r = i / (COLS / 4), c = 4 * (i % (COLS / 4));
                const unsigned m = (full || (r < rows_valid && c < cols_valid)) ? 0xFFFFFFFFu : 0u;
                *reinterpret_cast<float4*>(lds + r * PITCH + c) =
                    make_float4(mask_bits(v[4 * j], m), mask_bits(v[4 * j + 1], m), mask_bits(v[4 * j + 2], m),
                                mask_bits(v[4 * j + 3], m));
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = tid + THREADS * j, r = i / COLS, c = i % COLS;
                const unsigned m = (full || (r < rows_valid && c < cols_valid)) ? 0xFFFFFFFFu : 0u;
                lds[r * PITCH + c] = mask_bits(v[j], m);
            }
        }
    }
};

// ---- fp32 products on the bf16 matrix path ------------------------------------------------------------------
// An fp32 operand is split, while it is staged, into three bf16 planes x = hi + mid + lo (to ~2^-25 relative); each
// bf16 x bf16 product is exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16, and the six products
// mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi (smallest first) carry the full fp32 product — six 32-cycle MFMAs per
// K = 16 block instead of eight 64-cycle fp32 ones (tools/experiments/split_bf16_gram.hip: 1.87x at the same error
// against fp64).  LDS image of a tile: [3 planes][ROWS][PITCH] bf16, PITCH = 32 + 8 (80-byte rows: conflict-free
// b128 reads); lane half h of a wave supplies k = 8h .. 8h+7 of each 16-wide block.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int SPLIT_COLS = 32, SPLIT_PITCH = SPLIT_COLS + 8;

__device__ __forceinline__ void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)x;
    const float r1 = x - (float)hi;
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);
}

// acc += A . B^T for one K = 16 block from the three planes of each operand (8 bf16 per lane and plane)
__device__ __forceinline__ void mfma_split6(f32x16& acc, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
    f32x16 c = acc;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    acc = c;
}

// Write a fetched [ROWS][32] tile (TileStage<ROWS, 32, true, THREADS>) to LDS as three bf16 planes; what lies outside
// the matrix is written as zero.
template <int ROWS, int THREADS>
__device__ __forceinline__ void stash_planes(const TileStage<ROWS, SPLIT_COLS, true, THREADS>& tile, __bf16* base, int tid) {
    const bool full = tile.rows_valid >= ROWS && tile.cols_valid >= SPLIT_COLS;
#pragma unroll
    for (int j = 0; j < tile.NV / 4; ++j) {
        const int q = tid + THREADS * j, r = q / (SPLIT_COLS / 4), c = 4 * (q % (SPLIT_COLS / 4));
        const unsigned m = (full || (r < tile.rows_valid && c < tile.cols_valid)) ? 0xFFFFFFFFu : 0u;
        bf16x4 p0, p1, p2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 h, md, l;
            split3(mask_bits(tile.v[4 * j + e], m), h, md, l);
            p0[e] = h; p1[e] = md; p2[e] = l;
        }
        *reinterpret_cast<bf16x4*>(base + (0 * ROWS + r) * SPLIT_PITCH + c) = p0;
        *reinterpret_cast<bf16x4*>(base + (1 * ROWS + r) * SPLIT_PITCH + c) = p1;
        *reinterpret_cast<bf16x4*>(base + (2 * ROWS + r) * SPLIT_PITCH + c) = p2;
    }
}

// The same three planes kept in global memory (dl_planes.hip: split once per call instead of once per workgroup
// that stages the tile), TILE-MAJOR: the matrix is cut into [128 rows][COLS cols] tiles (COLS = 32 or 16 = the K
// extent one pipeline step of the consuming kernel contracts), each stored as one contiguous block
// [3 planes][128][COLS] bf16 — the order the kernels consume it in, so a tile is a straight, fully coalesced copy
// global -> registers -> LDS image [3][128][COLS + 8] without masks (rows and columns are zero-filled out to
// multiples of 128 rows and of the array's column padding).  Tile (rb, cb) of a matrix with ncb column chunks starts at plane_tile<COLS>(rb, cb, ncb).
inline size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }
constexpr int PLANE_ROWS = 128;
template <int COLS>
__host__ __device__ inline size_t plane_tile(int rb, int cb, int ncb) { return ((size_t)rb * ncb + cb) * 3 * PLANE_ROWS * COLS; }
// elements of the plane array of a [rows][cols] matrix whose columns are padded to a multiple of col_pad
inline size_t plane_array_elems(size_t rows, size_t cols, int col_pad) { return 3 * round_up(rows, PLANE_ROWS) * round_up(cols, col_pad); }
template <int COLS>
inline int plane_chunks(size_t cols, int col_pad) { return (int)(round_up(cols, col_pad) / COLS); }

template <int THREADS, int COLS>
struct PlaneStage {
    static constexpr int PIECES = PLANE_ROWS * COLS / 8;       // 16-byte pieces per plane of the tile
    static constexpr int PITCH = COLS + 8;                     // LDS row pitch (bf16): conflict-free b128 reads
    static_assert(PIECES % THREADS == 0, "a plane of the tile divides over the workgroup");
    static constexpr int PER = PIECES / THREADS;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));    // (HIP's uint4 class kept the array in scratch)
    u32x4 v[3 * PER];
    __device__ __forceinline__ void fetch(const __bf16* __restrict__ tile, int tid) {
#pragma unroll
        for (int j = 0; j < 3 * PER; ++j) v[j] = *reinterpret_cast<const u32x4*>(tile + (unsigned)((tid + THREADS * j) * 8));
    }
    __device__ __forceinline__ void stash(__bf16* lds, int tid) const {
#pragma unroll
        for (int j = 0; j < 3 * PER; ++j) {
            const int q = tid + THREADS * j, p = q / PIECES, r = (q % PIECES) / (COLS / 8), c = (q % (COLS / 8)) * 8;
            *reinterpret_cast<u32x4*>(lds + (p * PLANE_ROWS + r) * PITCH + c) = v[j];
        }
    }
};

// dl_planes.hip: tile-major planes of B matrices src [B][R][C] (row stride ld, batch stride sb; matrix b starts at
// dst + b * plane_array_elems(R, C)), and of the TRANSPOSE of src [R][C] (rows = c, columns = r); everything out to
// the padded extents is written.
void split_rows(const float* src, int B, int R, int C, int ld, size_t sb, __bf16* dst, hipStream_t st);      // 32-column tiles, columns padded to 32
void split_transposed(const float* src, int R, int C, int ld, __bf16* dst, hipStream_t st);
// W2 [rows][nhid] -> planes [3][rows][nhid_p] in the k-slot order of the forward's layer 2 (dl_planes.hip)
void split_w2(const float* W2, int rows, int nhid, __bf16* dst, int nhid_p, hipStream_t st);                 // 16-column tiles, columns (= R) padded to 128
// x [N][F], W1 [K][nhid][F] and W2 [K*d][nhid] in one launch (the two-layer forward's operands)
void split_fwd_operands(const float* x, int N, int F, __bf16* xP, const float* W1, int K, int nhid, __bf16* wP,
                        const float* W2, int d, __bf16* w2P, int nhid_p, hipStream_t st);

}  // namespace project
}  // namespace dl
