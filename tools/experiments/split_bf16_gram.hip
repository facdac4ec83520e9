// Experiment (not product code): fp32-grade Gram products on the bf16 matrix path.
//   G = X X^T, X [N][D] fp32.  MODE 0: v_mfma_f32_32x32x2_f32 (what dl_score_dense.hip / dl_project.hip use).
//   MODE 1: every fp32 operand is split at staging time into three bf16 planes hi + mid + lo (x = hi + mid + lo to
//   ~2^-25); the six products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid are each EXACT in the fp32 accumulator of
//   v_mfma_f32_32x32x16_bf16, so the result has fp32-grade accuracy at 6 bf16 MFMAs (32 cycles each) per K=16 block
//   against 8 fp32 MFMAs (64 cycles each): 2.7x fewer matrix-pipe cycles.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/experiments/split_bf16_gram.hip -o /tmp/sbg && /tmp/sbg
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int TT = 128, DC = 32, THR = 256;
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int MODE>
__global__ __launch_bounds__(THR) void gram_kernel(const float* __restrict__ X, int N, int D, float* __restrict__ G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int u0 = blockIdx.y * TT, v0 = blockIdx.x * TT;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, half = lane >> 5;
    const int wu = wave >> 1, wv = wave & 1;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const int steps = D / DC;
    if constexpr (MODE == 0) {
        constexpr int LD = DC + 4;
        float* us = reinterpret_cast<float*>(lds_raw);            // [2][TT][LD]
        float* vs = us + 2 * TT * LD;
        float4 ur[4], vr[4];
        auto fetch = [&](int s) {
            for (int j = 0; j < 4; ++j) {
                const int i = tid + THR * j, r = i / 8, c = 4 * (i % 8);
                ur[j] = *reinterpret_cast<const float4*>(X + (size_t)(u0 + r) * D + s * DC + c);
                vr[j] = *reinterpret_cast<const float4*>(X + (size_t)(v0 + r) * D + s * DC + c);
            }
        };
        auto stash = [&](int s) {
            for (int j = 0; j < 4; ++j) {
                const int i = tid + THR * j, r = i / 8, c = 4 * (i % 8);
                *reinterpret_cast<float4*>(us + ((s & 1) * TT + r) * LD + c) = ur[j];
                *reinterpret_cast<float4*>(vs + ((s & 1) * TT + r) * LD + c) = vr[j];
            }
        };
        fetch(0); stash(0); if (steps > 1) fetch(1);
        __syncthreads();
        for (int s = 0; s < steps; ++s) {
            const float* ub = us + ((s & 1) * TT + wu * 64 + li) * LD + half * 16;
            const float* vb = vs + ((s & 1) * TT + wv * 64 + li) * LD + half * 16;
            for (int q = 0; q < 4; ++q) {
                const float4 a0 = *reinterpret_cast<const float4*>(ub + 4 * q), a1 = *reinterpret_cast<const float4*>(ub + 32 * LD + 4 * q);
                const float4 b0 = *reinterpret_cast<const float4*>(vb + 4 * q), b1 = *reinterpret_cast<const float4*>(vb + 32 * LD + 4 * q);
#define M4(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, B.x, C, 0, 0, 0); C = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, B.y, C, 0, 0, 0); \
                    C = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, B.z, C, 0, 0, 0); C = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, B.w, C, 0, 0, 0);
                M4(a0, b0, acc[0][0]) M4(a0, b1, acc[0][1]) M4(a1, b0, acc[1][0]) M4(a1, b1, acc[1][1])
                if (q == 0) { if (s + 1 < steps) stash(s + 1); if (s + 2 < steps) fetch(s + 2); }
            }
            __syncthreads();
        }
    } else {
        constexpr int LDH = DC + 8;                                // bf16 row pitch: 80 bytes, conflict-free b128 reads
        __bf16* us = reinterpret_cast<__bf16*>(lds_raw);          // [2 buffers][3 planes][TT][LDH]
        __bf16* vs = us + 2 * 3 * TT * LDH;
        float4 ur[4], vr[4];
        auto fetch = [&](int s) {
            for (int j = 0; j < 4; ++j) {
                const int i = tid + THR * j, r = i / 8, c = 4 * (i % 8);
                ur[j] = *reinterpret_cast<const float4*>(X + (size_t)(u0 + r) * D + s * DC + c);
                vr[j] = *reinterpret_cast<const float4*>(X + (size_t)(v0 + r) * D + s * DC + c);
            }
        };
        auto split_store = [&](__bf16* base, int r, int c, float4 x) {   // three planes of 4 consecutive k
            const float xs[4] = {x.x, x.y, x.z, x.w};
            bf16x4 p0, p1, p2;
            for (int e = 0; e < 4; ++e) {
                const __bf16 h = (__bf16)xs[e];
                const float r1 = xs[e] - (float)h;
                const __bf16 m = (__bf16)r1;
                const float r2 = r1 - (float)m;
                p0[e] = h; p1[e] = m; p2[e] = (__bf16)r2;
            }
            *reinterpret_cast<bf16x4*>(base + (0 * TT + r) * LDH + c) = p0;
            *reinterpret_cast<bf16x4*>(base + (1 * TT + r) * LDH + c) = p1;
            *reinterpret_cast<bf16x4*>(base + (2 * TT + r) * LDH + c) = p2;
        };
        auto stash = [&](int s) {
            for (int j = 0; j < 4; ++j) {
                const int i = tid + THR * j, r = i / 8, c = 4 * (i % 8);
                split_store(us + (s & 1) * 3 * TT * LDH, r, c, ur[j]);
                split_store(vs + (s & 1) * 3 * TT * LDH, r, c, vr[j]);
            }
        };
        fetch(0); stash(0); if (steps > 1) fetch(1);
        __syncthreads();
        for (int s = 0; s < steps; ++s) {
            const __bf16* ub = us + (s & 1) * 3 * TT * LDH + (wu * 64 + li) * LDH + half * 8;
            const __bf16* vb = vs + (s & 1) * 3 * TT * LDH + (wv * 64 + li) * LDH + half * 8;
            for (int kb = 0; kb < DC / 16; ++kb) {                // one 32x32x16 block: lane half = which 8 of the 16 k
                bf16x8 a[3][2], b[3][2];
                for (int p = 0; p < 3; ++p)
                    for (int t = 0; t < 2; ++t) {
                        a[p][t] = *reinterpret_cast<const bf16x8*>(ub + (p * TT + t * 32) * LDH + kb * 16);
                        b[p][t] = *reinterpret_cast<const bf16x8*>(vb + (p * TT + t * 32) * LDH + kb * 16);
                    }
                for (int ta = 0; ta < 2; ++ta)
                    for (int tb = 0; tb < 2; ++tb) {
                        f32x16 c = acc[ta][tb];                   // smallest terms first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][ta], b[1][tb], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][ta], b[2][tb], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][ta], b[0][tb], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][ta], b[1][tb], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][ta], b[0][tb], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][ta], b[0][tb], c, 0, 0, 0);
                        acc[ta][tb] = c;
                    }
                if (kb == 0) { if (s + 1 < steps) stash(s + 1); if (s + 2 < steps) fetch(s + 2); }
            }
            __syncthreads();
        }
    }
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            const int v = v0 + wv * 64 + b * 32 + li;
            for (int r = 0; r < 16; ++r) {
                const int u = u0 + wu * 64 + a * 32 + acc_row(r, half);
                G[(size_t)u * N + v] = acc[a][b][r];
            }
        }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const int N = 4096, D = 512;                                   // multiples of 128 / 32: no edge handling in this experiment
    std::vector<float> hx((size_t)N * D);
    srand(1);
    for (auto& v : hx) v = (float)((rand() / (double)RAND_MAX - 0.5) * 2.0) * (1.0f + (rand() % 8));
    float *dx, *dg;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dg, (size_t)N * N * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    const size_t lds0 = sizeof(float) * 4 * TT * (DC + 4), lds1 = 2 * 2 * 3 * TT * (DC + 8) * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    std::vector<float> hg((size_t)N * N);
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const dim3 grid(N / TT, N / TT), block(THR);
        for (int it = 0; it < 3; ++it) {
            if (mode == 0) hipLaunchKernelGGL(gram_kernel<0>, grid, block, lds0, 0, dx, N, D, dg);
            else hipLaunchKernelGGL(gram_kernel<1>, grid, block, lds1, 0, dx, N, D, dg);
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int it = 0; it < reps; ++it) {
            if (mode == 0) hipLaunchKernelGGL(gram_kernel<0>, grid, block, lds0, 0, dx, N, D, dg);
            else hipLaunchKernelGGL(gram_kernel<1>, grid, block, lds1, 0, dx, N, D, dg);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hg.data(), dg, hg.size() * 4, hipMemcpyDeviceToHost));
        double max_rel = 0.0, scale = 0.0;
        for (int t = 0; t < 4000; ++t) {                           // sampled entries against fp64
            const int u = rand() % N, v = rand() % N;
            double ref = 0.0, mag = 0.0;
            for (int k = 0; k < D; ++k) { ref += (double)hx[(size_t)u * D + k] * hx[(size_t)v * D + k]; mag += fabs((double)hx[(size_t)u * D + k] * hx[(size_t)v * D + k]); }
            max_rel = fmax(max_rel, fabs(hg[(size_t)u * N + v] - ref) / mag);
            scale = fmax(scale, mag);
        }
        const double tf = 2.0 * N * (double)N * D / (ms / reps * 1e-3) / 1e12;
        printf("%s: %.1f us per Gram (N=%d, D=%d), %.1f TFLOP/s effective, max |err| / sum|terms| = %.2e\n",
               mode == 0 ? "fp32 MFMA 32x32x2      " : "split bf16 x6 32x32x16 ", ms / reps * 1e3, N, D, tf, max_rel);
    }
    return 0;
}
