// Factor projection Z[n][k][:] = MLP_k(x[n]) on the matrix cores (model.py:13-15, 24-27, 106) — the
// only dense contraction of the path.  fp32 in / fp32 accumulate with v_mfma_f32_32x32x2_f32, which
// is bit-for-bit a k-ordered fmaf chain (no reduced precision), so the result keeps the
// reference's fp32 semantics up to summation order.
//
// Two-layer form (Factor2): one workgroup = 4 waves = 128 nodes x ONE factor k.  Per 32 hidden units:
//   layer 1 (transposed):  hidT[32 hidden][32 nodes] = W1_k[32][F] . x^T        A = W1 rows, B = x^T
//   bias + ReLU in the accumulator registers
//   layer 2:               Z^T[d][32 nodes] += W2_k[d][32 hidden] . hidT         B = the accumulator itself
// A 32x32 accumulator has its column on the lane and its rows in the 16 registers, which is exactly
// the B-operand shape of the next MFMA when that product sums over the accumulator's ROW index
// (register r supplies the k-pair {(r&3)+8(r>>2), +4}); so the hidden activations never leave the
// register file — no [N, K*nhid] tensor is written to HBM and re-read, unlike two library GEMMs.
// Operand tiles are double-buffered in LDS (row pitch + 4 floats: aligned, conflict-free b128 accesses)
// and the next step's tiles are fetched into registers behind the current step's MFMA chain.
#include "dl_common.h"
#include "dl_kernels.h"

namespace dl {
namespace project {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TILE_N = 128;   // nodes per workgroup (4 waves x 32)
constexpr int FC = 64;        // feature chunk staged per step
constexpr int HC = 32;        // hidden units per step

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

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

// 4 consecutive elements (row, col..col+3) of a row-major [n_rows][n_cols] matrix, zero outside.
// Branch-free: every load is issued unconditionally from a clamped (always valid) address and the
// result is selected afterwards, so hipcc keeps all loads of a tile in flight together (a branch
// around a load makes it wait for each one separately).  VEC: n_cols % 4 == 0, so a quad is either
// fully inside a row or fully outside and 16-byte aligned.
template <bool VEC>
__device__ __forceinline__ float4 load_quad(const float* __restrict__ src, int n_rows, int n_cols, int row, int col) {
    const bool row_ok = row < n_rows;
    const size_t rbase = (size_t)(row_ok ? row : 0) * n_cols;
    if constexpr (VEC) {
        const bool ok = row_ok && col < n_cols;
        const float4 q = *reinterpret_cast<const float4*>(src + rbase + (ok ? col : 0));
        return ok ? q : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = row_ok && col + i < n_cols;
            const float v = src[rbase + (ok ? col + i : 0)];
            e[i] = ok ? v : 0.0f;
        }
        return make_float4(e[0], e[1], e[2], e[3]);
    }
}
__device__ __forceinline__ void store_quad(float* lds, const float4& q) {
    *reinterpret_cast<float4*>(lds) = q;                        // rows are padded by 4 floats: 16-byte aligned
}

// Two-layer projection.  W1 [K][nhid][F], b1 [K][nhid], W2 [K][D][nhid], b2 [K][D], Z [N][K][D].
// Software pipeline: the tiles of step s+1 are fetched into registers while the MFMAs of step s run,
// written to the other LDS buffer afterwards; one barrier per step.
template <int D, bool VEC>
__global__ __launch_bounds__(256) void project2_fwd_kernel(const float* __restrict__ x, int N, int F, int nhid,
                                                           const float* __restrict__ W1, const float* __restrict__ b1,
                                                           const float* __restrict__ W2, const float* __restrict__ b2,
                                                           float* __restrict__ Z, int K) {
    constexpr int DT = D / 32;
    // Row pitch = tile width + 4 floats: rows stay 16-byte aligned (ds_write_b128 / ds_read_b128) and the
    // 16 lanes of a b128 read group land on 16 different 4-bank slots (pitch*4 B mod 256 B = 16 B).
    constexpr int LDX = FC + 4, LDW2 = HC + 4;
    constexpr int XQ = TILE_N * FC / 4 / 256;      // float4 per thread of the x tile        (8)
    constexpr int WQ = HC * FC / 4 / 256;          // ... of the W1 tile                     (2)
    constexpr int VQ = D * HC / 4 / 256;           // ... of the W2 tile                     (1, 2 or 4)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                               // [2][TILE_N][LDX]
    float* w1s = xs + 2 * TILE_N * LDX;            // [2][HC][LDX]
    float* w2s = w1s + 2 * HC * LDX;               // [2][D][LDW2]
    const int k = blockIdx.y;
    const int n0 = blockIdx.x * TILE_N;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, half = lane >> 5;
    const float* W1k = W1 + (size_t)k * nhid * F;
    const float* W2k = W2 + (size_t)k * D * nhid;
    const int nfc = (F + FC - 1) / FC, nhc = (nhid + HC - 1) / HC, steps = nfc * nhc;

    // With F <= 2*FC the two x chunks of the node tile fit the two LDS buffers for good: they are staged
    // once (during the first hidden chunk) instead of once per hidden chunk.
    const bool x_resident = nfc <= 2;
    float4 xq[XQ], wq[WQ], vq[VQ];
    float bq[16];
    auto fetch = [&](int s) {
        const int hc = s / nfc, fc = s - hc * nfc;
        if (!x_resident || hc == 0) {
#pragma unroll
            for (int j = 0; j < XQ; ++j) {
                const int i = tid + 256 * j, r = i / (FC / 4), c4 = i % (FC / 4);
                xq[j] = load_quad<VEC>(x, N, F, n0 + r, fc * FC + 4 * c4);
            }
        }
#pragma unroll
        for (int j = 0; j < WQ; ++j) {
            const int i = tid + 256 * j, r = i / (FC / 4), c4 = i % (FC / 4);
            wq[j] = load_quad<VEC>(W1k, nhid, F, hc * HC + r, fc * FC + 4 * c4);
        }
        if (fc == 0) {
#pragma unroll
            for (int j = 0; j < VQ; ++j) {
                const int i = tid + 256 * j, r = i / (HC / 4), c4 = i % (HC / 4);
                vq[j] = load_quad<VEC>(W2k, D, nhid, r, hc * HC + 4 * c4);
            }
        }
        if (fc == nfc - 1) {                                    // bias of this hidden chunk, needed after its last step
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int h = hc * HC + acc_row(r, half);
                const float bv = b1[(size_t)k * nhid + (h < nhid ? h : 0)];
                bq[r] = h < nhid ? bv : 0.0f;
            }
        }
    };
    auto stash = [&](int s) {
        const int hc = s / nfc, fc = s - hc * nfc;
        float* xb = xs + (x_resident ? fc : (s & 1)) * TILE_N * LDX;
        float* wb = w1s + (s & 1) * HC * LDX;
        if (!x_resident || hc == 0) {
#pragma unroll
            for (int j = 0; j < XQ; ++j) {
                const int i = tid + 256 * j, r = i / (FC / 4), c4 = i % (FC / 4);
                store_quad(xb + r * LDX + 4 * c4, xq[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < WQ; ++j) {
            const int i = tid + 256 * j, r = i / (FC / 4), c4 = i % (FC / 4);
            store_quad(wb + r * LDX + 4 * c4, wq[j]);
        }
        if (fc == 0) {
            float* vb = w2s + (hc & 1) * D * LDW2;
#pragma unroll
            for (int j = 0; j < VQ; ++j) {
                const int i = tid + 256 * j, r = i / (HC / 4), c4 = i % (HC / 4);
                store_quad(vb + r * LDW2 + 4 * c4, vq[j]);
            }
        }
    };

    f32x16 zacc[DT], hacc;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) zacc[dt][r] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[r] = 0.0f;

    fetch(0);
    stash(0);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int hc = s / nfc, fc = s - hc * nfc;
        float bias[16];                                         // this chunk's bias, fetched one step ahead
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[r] = bq[r];
        if (s + 1 < steps) fetch(s + 1);                        // global loads in flight behind the MFMAs
        // MFMA step q contracts the feature pair {q, 32+q} of the chunk: lane half h owns features
        // h*32 .. h*32+31, i.e. 32 CONTIGUOUS floats per operand -> 8 ds_read_b128 each, all issued
        // before the 32-MFMA chain (one wave per SIMD: nothing else would hide the LDS latency).
        const float4* xa = reinterpret_cast<const float4*>(xs + (x_resident ? fc : (s & 1)) * TILE_N * LDX +
                                                           (wave * 32 + li) * LDX + half * 32);
        const float4* wa = reinterpret_cast<const float4*>(w1s + (s & 1) * HC * LDX + li * LDX + half * 32);
        float4 av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { av[q] = wa[q]; bv[q] = xa[q]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].x, bv[q].x, hacc, 0, 0, 0);
            hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].y, bv[q].y, hacc, 0, 0, 0);
            hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].z, bv[q].z, hacc, 0, 0, 0);
            hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].w, bv[q].w, hacc, 0, 0, 0);
        }
        if (fc == nfc - 1) {
            // bias + ReLU on hidT (row = hidden unit, column = node), then layer 2 straight from registers
#pragma unroll
            for (int r = 0; r < 16; ++r) hacc[r] = fmaxf(hacc[r] + bias[r], 0.0f);
            // register r of hidT holds hidden rows acc_row(r, half): 4 runs of 4 consecutive rows -> 4 b128 reads
            const float* vb = w2s + (hc & 1) * D * LDW2;
            float4 wv[DT][4];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    wv[dt][g4] = *reinterpret_cast<const float4*>(vb + (dt * 32 + li) * LDW2 + 8 * g4 + 4 * half);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    zacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[dt][g4].x, hacc[4 * g4 + 0], zacc[dt], 0, 0, 0);
                    zacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[dt][g4].y, hacc[4 * g4 + 1], zacc[dt], 0, 0, 0);
                    zacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[dt][g4].z, hacc[4 * g4 + 2], zacc[dt], 0, 0, 0);
                    zacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[dt][g4].w, hacc[4 * g4 + 3], zacc[dt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) hacc[r] = 0.0f;
        }
        if (s + 1 < steps) stash(s + 1);
        __syncthreads();
    }
    // epilogue: Z[n][k][dd] = Z^T[dd][n] + b2[k][dd]; registers 4g..4g+3 are 4 consecutive dd
    const int n = n0 + wave * 32 + li;
    if (n < N) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = dt * 32 + 8 * g4 + 4 * half;
                const float4 bb = *reinterpret_cast<const float4*>(b2 + (size_t)k * D + dd);
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

template <int D>
static size_t project2_lds_bytes() {
    return sizeof(float) * (2 * TILE_N * (FC + 4) + 2 * HC * (FC + 4) + 2 * D * (HC + 4));
}

// Single-layer projection (Factor): W [K][D][F], b [K][D]:  Z[n][k][:] = W_k x[n] + b_k.
template <int D>
__global__ __launch_bounds__(256) void project1_fwd_kernel(const float* __restrict__ x, int N, int F,
                                                           const float* __restrict__ W, const float* __restrict__ b,
                                                           float* __restrict__ Z, int K) {
    constexpr int DT = D / 32;
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

bool project_supported(int d) { return d == 32 || d == 64 || d == 128; }

template <int D, bool VEC>
static void launch2_t(dim3 grid, dim3 block, hipStream_t st, const float* x, int N, int F, int nhid, const float* W1,
                      const float* b1, const float* W2, const float* b2, float* Z, int K) {
    using namespace project;
    static bool attr_done = false;                         // > 64 KiB of dynamic LDS needs the attribute once
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&project2_fwd_kernel<D, VEC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)project2_lds_bytes<D>());
        attr_done = true;
    }
    hipLaunchKernelGGL((project2_fwd_kernel<D, VEC>), grid, block, project2_lds_bytes<D>(), st, x, N, F, nhid, W1, b1, W2,
                       b2, Z, K);
}

static void launch2(int d, bool vec, dim3 grid, dim3 block, hipStream_t st, const float* x, int N, int F, int nhid,
                    const float* W1, const float* b1, const float* W2, const float* b2, float* Z, int K) {
#define DL_P2(DD)                                                                             \
    if (d == DD) {                                                                            \
        if (vec) launch2_t<DD, true>(grid, block, st, x, N, F, nhid, W1, b1, W2, b2, Z, K);   \
        else launch2_t<DD, false>(grid, block, st, x, N, F, nhid, W1, b1, W2, b2, Z, K);      \
        return;                                                                               \
    }
    DL_P2(32) DL_P2(64) DL_P2(128)
#undef DL_P2
}

int project_fwd(const float* x, int N, int F, int K, int nhid, int d, const float* W1, const float* b1,
                const float* W2, const float* b2, float* Z, hipStream_t st) {
    using namespace project;
    const dim3 grid((unsigned)((N + TILE_N - 1) / TILE_N), (unsigned)K), block(256);
    if (W2 == nullptr) {          // single Linear(F -> d): W1 is [K][d][F], b1 is [K][d]
        if (d == 32) hipLaunchKernelGGL(project1_fwd_kernel<32>, grid, block, 0, st, x, N, F, W1, b1, Z, K);
        else if (d == 64) hipLaunchKernelGGL(project1_fwd_kernel<64>, grid, block, 0, st, x, N, F, W1, b1, Z, K);
        else hipLaunchKernelGGL(project1_fwd_kernel<128>, grid, block, 0, st, x, N, F, W1, b1, Z, K);
    } else {
        const bool vec = (F % 4 == 0) && (nhid % 4 == 0);      // quads never straddle a row end
        launch2(d, vec, grid, block, st, x, N, F, nhid, W1, b1, W2, b2, Z, K);
    }
    return check_launch("project_fwd");
}

}  // namespace dl
