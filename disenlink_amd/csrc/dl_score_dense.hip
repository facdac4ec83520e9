// Dense [N][N] link scorer on the matrix cores: the reference's link_pred (model.py:109-113),
//   P[u][v] = sigmoid( sum_k (h_k[u].h_k[v]) * exp(z_k[u].z_k[v] / t) )        for ALL pairs,
// which the drop-in forward(x, adj) must return.  Per factor it is two N x N x d Gram products —
// GEMM-shaped work (4*N^2*K*d FLOP, 55 GFLOP at squirrel size), so unlike the pair-list scorer (a gather
// bound by L2 bandwidth) it belongs on MFMA: fp32 in / fp32 accumulate, v_mfma_f32_32x32x2_f32.
//
// One workgroup = 8 waves = a 128 (u) x 128 (v) tile of P; waves are 4 u-quarters x 2 v-halves, each owning
// 32 u x 64 v = two accumulators, two waves per SIMD.  Per factor k: S = Z_k[u] Z_k[v]^T accumulated over d in
// chunks of 32 features (operand tiles [128 rows][32], pitch 36 floats, double-buffered in LDS with the
// write-after-barrier staging of dl_project.hip), e = exp(S / t) in the accumulator registers, then
// Q = H_k[u] H_k[v]^T the same way and term += Q * e.  Nothing but P is written: no [K][N][N] tensor.
// P is symmetric — bit for bit, because entry (u,v) and entry (v,u) are the same products summed in the same
// order — so only the tile pairs with u tile <= v tile are computed and an off-diagonal tile is also stored
// transposed (4 consecutive u per register quad: 16-byte stores).  Work items are dealt to the XCDs in runs
// of 32 consecutive (same u tile, consecutive v tiles) items.
#include <cstdlib>
#include "dl_common.h"
#include "dl_kernels.h"
#include "dl_tiles.h"

namespace dl {
namespace dense {

using namespace project;       // TileStage, f32x16, acc_row, DL_MFMA, xcd_item

constexpr int TT = 128;        // tile edge (u and v)
constexpr int DTHR = 512;

// DC = features per step (64 when d allows: one barrier per 64 MFMAs of a wave), LDS row pitch DC + 4
template <int DC>
__global__ __launch_bounds__(DTHR) void score_allpairs_mfma_kernel(const float* __restrict__ Z, const float* __restrict__ H,
                                                                   int N, int K, int D, float t, float* __restrict__ prob) {
    constexpr int LDD = DC + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* us = lds;                       // [2][TT][LDD]
    float* vs = us + 2 * TT * LDD;         // [2][TT][LDD]
    const int nt = (N + TT - 1) / TT;
    // item i of the upper triangle in row-major order: rows a = 0.. hold nt - a items
    const int h = blockIdx.x;
    const int i = ((h >> 3) >> 5) * 256 + (h & 7) * 32 + ((h >> 3) & 31);        // runs of 32 items per XCD
    if (i >= nt * (nt + 1) / 2) return;
    int ta = (int)((2.0f * nt + 1.0f - sqrtf((2.0f * nt + 1.0f) * (2.0f * nt + 1.0f) - 8.0f * (float)i)) * 0.5f);
    ta = max(0, min(nt - 1, ta));
    while (ta > 0 && i < ta * nt - ta * (ta - 1) / 2) --ta;                      // first item of row a = a*nt - a(a-1)/2
    while (i >= (ta + 1) * nt - (ta + 1) * ta / 2) ++ta;
    const int tb = ta + (i - (ta * nt - ta * (ta - 1) / 2));
    const int u0 = ta * TT, v0 = tb * TT;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, half = lane >> 5;
    const int wu = wave >> 1, wv = wave & 1;
    const int nd = D / DC;
    const int steps = K * 2 * nd;
    const int ld = K * D;

    TileStage<TT, DC, true, DTHR> ut, vt;
    auto fetch = [&](int s) {
        const int k = s / (2 * nd), r = s - k * 2 * nd;
        const float* src = r < nd ? Z : H;
        const int dc = r < nd ? r : r - nd;
        ut.fetch(src + ((size_t)u0 * K + k) * D + dc * DC, ld, N - u0, DC, tid);
        vt.fetch(src + ((size_t)v0 * K + k) * D + dc * DC, ld, N - v0, DC, tid);
    };
    auto stash = [&](int s) {
        ut.template stash<LDD>(us + (s & 1) * TT * LDD, tid);
        vt.template stash<LDD>(vs + (s & 1) * TT * LDD, tid);
    };

    f32x16 acc[2], term[2];
    float e[2][16];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        zero_acc(acc[b]);
        zero_acc(term[b]);
    }
    fetch(0);
    stash(0);
    if (steps > 1) fetch(1);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int r = s % (2 * nd);
        // A = u rows of this quarter (lane = u), B = v rows (lane = v): acc[u][v], u rows in the registers
        const float* ub = us + (s & 1) * TT * LDD + (wu * 32 + li) * LDD + half * (DC / 2);
        const float* vb = vs + (s & 1) * TT * LDD + (wv * 64 + li) * LDD + half * (DC / 2);
        float4 a[2][2], b[2][2][2];
        auto read_block = [&](int j) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                a[j & 1][q] = *reinterpret_cast<const float4*>(ub + 8 * j + 4 * q);
                b[j & 1][0][q] = *reinterpret_cast<const float4*>(vb + 8 * j + 4 * q);
                b[j & 1][1][q] = *reinterpret_cast<const float4*>(vb + 32 * LDD + 8 * j + 4 * q);
            }
        };
        read_block(0);
#pragma unroll
        for (int j = 0; j < DC / 16; ++j) {
            if (j + 1 < DC / 16) read_block(j + 1);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                DL_MFMA(acc[0], a[j & 1][q].x, b[j & 1][0][q].x);
                DL_MFMA(acc[1], a[j & 1][q].x, b[j & 1][1][q].x);
                DL_MFMA(acc[0], a[j & 1][q].y, b[j & 1][0][q].y);
                DL_MFMA(acc[1], a[j & 1][q].y, b[j & 1][1][q].y);
                DL_MFMA(acc[0], a[j & 1][q].z, b[j & 1][0][q].z);
                DL_MFMA(acc[1], a[j & 1][q].z, b[j & 1][1][q].z);
                DL_MFMA(acc[0], a[j & 1][q].w, b[j & 1][0][q].w);
                DL_MFMA(acc[1], a[j & 1][q].w, b[j & 1][1][q].w);
            }
            if (j == 0) {
                if (s + 1 < steps) stash(s + 1);
                if (s + 2 < steps) fetch(s + 2);
            }
        }
        if (r == nd - 1) {                                      // S complete: e = exp(S / t)  (model.py:56)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) e[bb][i] = expf(div_t(acc[bb][i], t));
                zero_acc(acc[bb]);
            }
        } else if (r == 2 * nd - 1) {                           // Q complete: term += Q * e  (model.py:110-112)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) term[bb][i] += acc[bb][i] * e[bb][i];
                zero_acc(acc[bb]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
        const int v = v0 + wv * 64 + bb * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            term[bb][r] = sigmoid_ref(term[bb][r]);
            const int u = u0 + wu * 32 + acc_row(r, half);
            if (u < N && v < N) prob[(size_t)u * N + v] = term[bb][r];
        }
        if (ta != tb && v < N) {                                // mirror: P[v][u], registers 4g..4g+3 = 4 consecutive u
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int u = u0 + wu * 32 + 8 * g + 4 * half;
                float* dst = prob + (size_t)v * N + u;
                if (u + 3 < N && (N & 3) == 0) {
                    *reinterpret_cast<float4*>(dst) = make_float4(term[bb][4 * g], term[bb][4 * g + 1], term[bb][4 * g + 2],
                                                                  term[bb][4 * g + 3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (u + j < N) dst[j] = term[bb][4 * g + j];
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// The same scorer with the Gram products on the bf16 matrix path at fp32-grade accuracy: every fp32 operand is
// split, while it is staged, into three bf16 planes x = hi + mid + lo (to ~2^-25 relative); each bf16 x bf16
// product is exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16, and the six products hi*hi, hi*mid,
// mid*hi, hi*lo, lo*hi, mid*mid (smallest first) carry the full fp32 product — six 32-cycle MFMAs per K = 16 block
// instead of eight 64-cycle fp32 ones.  Measured on a standalone Gram product (tools/experiments/split_bf16_gram.hip):
// 1.87x at the same error against fp64 (2.2e-7 vs 1.9e-7 of the sum of |terms|).
// LDS image per operand and buffer: [3 planes][128 rows][32 + 8] bf16 (row pitch 80 bytes: conflict-free b128 reads).
// (split3 / stash_planes / mfma_split6 and the LDS image are the shared ones of dl_tiles.h)
constexpr int SDC = SPLIT_COLS, SLD = SPLIT_PITCH;

// PLANES: Z and H arrive pre-split as tile-major plane arrays (dl_planes.hip: per factor k the planes of the [N][D]
// matrix Z[:, k, :], 32-column tiles; batch = elements per factor), split once per call instead of once per tile pair
// that stages them (each node tile is staged by ~nt workgroups); the staging is then a straight copy.
struct DensePlanes { const __bf16* z; const __bf16* h; size_t batch; };

template <bool PLANES>
__global__ __launch_bounds__(DTHR) void score_allpairs_split_kernel(const float* __restrict__ Z, const float* __restrict__ H,
                                                                    int N, int K, int D, float t, float* __restrict__ prob,
                                                                    DensePlanes P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __bf16* us = reinterpret_cast<__bf16*>(lds);               // [2][3][TT][SLD]
    __bf16* vs = us + 2 * 3 * TT * SLD;
    const int nt = (N + TT - 1) / TT;
    const int hb = blockIdx.x;
    const int i = ((hb >> 3) >> 5) * 256 + (hb & 7) * 32 + ((hb >> 3) & 31);     // runs of 32 items per XCD
    if (i >= nt * (nt + 1) / 2) return;
    int ta = (int)((2.0f * nt + 1.0f - sqrtf((2.0f * nt + 1.0f) * (2.0f * nt + 1.0f) - 8.0f * (float)i)) * 0.5f);
    ta = max(0, min(nt - 1, ta));
    while (ta > 0 && i < ta * nt - ta * (ta - 1) / 2) --ta;
    while (i >= (ta + 1) * nt - (ta + 1) * ta / 2) ++ta;
    const int tb = ta + (i - (ta * nt - ta * (ta - 1) / 2));
    const int u0 = ta * TT, v0 = tb * TT;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, half = lane >> 5;
    const int wu = wave >> 1, wv = wave & 1;
    const int nd = D / SDC;
    const int steps = K * 2 * nd;
    const int ld = K * D;

    TileStage<TT, SDC, true, DTHR> ut, vt;                     // raw fp32 quads in flight; split when written to LDS
    PlaneStage<DTHR, SDC> uq, vq;                              // PLANES: 16-byte pieces of the pre-split tiles
    static_assert(TT == PLANE_ROWS, "tiles of the plane arrays");
    auto fetch = [&](int s) {
        const int k = s / (2 * nd), r = s - k * 2 * nd;
        const int dc = r < nd ? r : r - nd;
        if constexpr (PLANES) {
            const __bf16* src = (r < nd ? P.z : P.h) + (size_t)k * P.batch;
            uq.fetch(src + plane_tile<SDC>(ta, dc, nd), tid);
            vq.fetch(src + plane_tile<SDC>(tb, dc, nd), tid);
        } else {
            const float* src = r < nd ? Z : H;
            ut.fetch(src + ((size_t)u0 * K + k) * D + dc * SDC, ld, N - u0, SDC, tid);
            vt.fetch(src + ((size_t)v0 * K + k) * D + dc * SDC, ld, N - v0, SDC, tid);
        }
    };
    auto stash = [&](int s) {
        if constexpr (PLANES) {
            uq.stash(us + (s & 1) * 3 * TT * SLD, tid);
            vq.stash(vs + (s & 1) * 3 * TT * SLD, tid);
        } else {
            stash_planes(ut, us + (s & 1) * 3 * TT * SLD, tid);   // rows past N are written as zero
            stash_planes(vt, vs + (s & 1) * 3 * TT * SLD, tid);
        }
    };

    f32x16 acc[2], term[2];
    float e[2][16];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        zero_acc(acc[b]);
        zero_acc(term[b]);
    }
    fetch(0);
    stash(0);
    fetch(min(1, steps - 1));
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int r = s % (2 * nd);
        // lane half h supplies k = 8h .. 8h+7 of each 16-wide block; A = u rows of this quarter, B = v rows
        const __bf16* ub = us + (s & 1) * 3 * TT * SLD + (wu * 32 + li) * SLD + half * 8;
        const __bf16* vb = vs + (s & 1) * 3 * TT * SLD + (wv * 64 + li) * SLD + half * 8;
#pragma unroll
        for (int kb = 0; kb < SDC / 16; ++kb) {
            bf16x8 a[3], b0[3], b1[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[p] = *reinterpret_cast<const bf16x8*>(ub + p * TT * SLD + kb * 16);
                b0[p] = *reinterpret_cast<const bf16x8*>(vb + p * TT * SLD + kb * 16);
                b1[p] = *reinterpret_cast<const bf16x8*>(vb + (p * TT + 32) * SLD + kb * 16);
            }
            mfma_split6(acc[0], a, b0);                         // six products, smallest terms first
            mfma_split6(acc[1], a, b1);
            if (kb == 0) {
                if (s + 1 < steps) stash(s + 1);
                fetch(min(s + 2, steps - 1));                   // unconditional: see project2_fwd_kernel
            }
        }
        if (r == nd - 1) {                                      // S complete: e = exp(S / t)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
                for (int q = 0; q < 16; ++q) e[bb][q] = expf(div_t(acc[bb][q], t));
                zero_acc(acc[bb]);
            }
        } else if (r == 2 * nd - 1) {                           // Q complete: term += Q * e
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
                for (int q = 0; q < 16; ++q) term[bb][q] += acc[bb][q] * e[bb][q];
                zero_acc(acc[bb]);
            }
        }
        __syncthreads();
    }
    // Stores.  The two products of a swapped pair are accumulated in a different order here (hi*lo before lo*hi), so
    // (u,v) and (v,u) computed independently may differ in the last bit: a diagonal tile therefore writes only its
    // upper triangle directly and mirrors the strict upper triangle — every entry of P is written exactly once, from
    // the (min, max) ordering of its pair, and P stays symmetric bit for bit.
    const bool diag = ta == tb;
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
        const int v = v0 + wv * 64 + bb * 32 + li;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            term[bb][q] = sigmoid_ref(term[bb][q]);
            const int u = u0 + wu * 32 + acc_row(q, half);
            if (u < N && v < N && (!diag || u <= v)) prob[(size_t)u * N + v] = term[bb][q];
        }
        if (v < N && diag) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int u = u0 + wu * 32 + acc_row(q, half);
                if (u < v) prob[(size_t)v * N + u] = term[bb][q];           // u < v < N
            }
        } else if (v < N) {                                     // off-diagonal tile: registers 4g..4g+3 = 4 consecutive u
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int u = u0 + wu * 32 + 8 * g + 4 * half;
                float* dst = prob + (size_t)v * N + u;
                if (u + 3 < N && (N & 3) == 0) {
                    *reinterpret_cast<float4*>(dst) = make_float4(term[bb][4 * g], term[bb][4 * g + 1], term[bb][4 * g + 2],
                                                                  term[bb][4 * g + 3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (u + j < N) dst[j] = term[bb][4 * g + j];
                }
            }
        }
    }
}

}  // namespace dense

bool dense_mfma_supported(int d) { return d % 32 == 0; }

template <int DC>
static void launch_dense(const float* Z, const float* H, int N, int K, int d, float t, float* prob, hipStream_t st) {
    using namespace dense;
    static unsigned long long lds_done = 0;
    constexpr size_t lds = sizeof(float) * 4 * TT * (DC + 4);
    ensure_dynamic_lds(reinterpret_cast<const void*>(&score_allpairs_mfma_kernel<DC>), lds, lds_done);
    const int nt = (N + TT - 1) / TT;
    const int items = nt * (nt + 1) / 2;
    hipLaunchKernelGGL(score_allpairs_mfma_kernel<DC>, dim3((unsigned)((items + 255) / 256 * 256)), dim3(DTHR), lds, st, Z, H,
                       N, K, d, t, prob);
}

size_t dense_score_workspace_bytes(int N, int K, int d) {
    if (!dense_mfma_supported(d) || config().dense_fp32_mfma) return 0;
    return 2 * sizeof(__bf16) * (size_t)K * project::plane_array_elems(N, d, project::SPLIT_COLS);
}

int dense_mfma_score_allpairs_fwd(const float* Z, const float* H, int N, int K, int d, float t, float* prob, void* ws,
                                  size_t ws_bytes, hipStream_t st) {
    if (!config().dense_fp32_mfma) {                        // default: three-plane bf16 products (fp32-grade accuracy)
        using namespace dense;
        static unsigned long long lds_done_p = 0, lds_done_s = 0;
        constexpr size_t lds = (size_t)2 * 2 * 3 * TT * SLD * 2;
        const int nt = (N + TT - 1) / TT;
        const int items = nt * (nt + 1) / 2;
        const dim3 grid((unsigned)((items + 255) / 256 * 256));
        const size_t need = dense_score_workspace_bytes(N, K, d);
        if (ws != nullptr && ws_bytes >= need) {                // planes made once per call
            const size_t batch = project::plane_array_elems(N, d, project::SPLIT_COLS);
            __bf16* zp = static_cast<__bf16*>(ws);
            __bf16* hp = zp + (size_t)K * batch;
            project::split_rows(Z, K, N, d, K * d, (size_t)d, zp, st);
            project::split_rows(H, K, N, d, K * d, (size_t)d, hp, st);
            project::ensure_dynamic_lds(reinterpret_cast<const void*>(&score_allpairs_split_kernel<true>), lds, lds_done_p);
            hipLaunchKernelGGL(score_allpairs_split_kernel<true>, grid, dim3(DTHR), lds, st, Z, H, N, K, d, t, prob,
                               DensePlanes{zp, hp, batch});
        } else {                                                // no workspace: every tile pair splits what it stages
            project::ensure_dynamic_lds(reinterpret_cast<const void*>(&score_allpairs_split_kernel<false>), lds, lds_done_s);
            hipLaunchKernelGGL(score_allpairs_split_kernel<false>, grid, dim3(DTHR), lds, st, Z, H, N, K, d, t, prob,
                               DensePlanes{nullptr, nullptr, 0});
        }
        return check_launch("score_allpairs_fwd(split bf16)");
    }
    if (d % 64 == 0 && !config().dense_dc32) launch_dense<64>(Z, H, N, K, d, t, prob, st);
    else launch_dense<32>(Z, H, N, K, d, t, prob, st);
    return check_launch("score_allpairs_fwd(mfma)");
}

}  // namespace dl
