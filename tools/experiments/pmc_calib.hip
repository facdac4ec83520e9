// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the ACCESS SHAPES of the aggregation kernel
// (MI355X_MICROARCH.md §HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ...
// Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   stream16   : 16 B per lane, contiguous (the z rows, H stores' counterpart)           known bytes = n * 16
//   slice256   : groups of 16 lanes read one random 256-B slice (16 B per lane)         known bytes = n_slices * 256
//   dword_rand : every lane reads ONE dword from a random 128-B line of a big table     needed = 4 B, line = 128 B
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/experiments/pmc_calib.hip -o /tmp/pmc_calib
//                              rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/cal -o out --output-format csv -- /tmp/pmc_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void stream16(const float4* __restrict__ src, float4* __restrict__ sink, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 acc = make_float4(0, 0, 0, 0);
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = src[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x == 12345.678f) sink[0] = acc;                       // never true: keeps the loads alive
}

// n_slices random slices of 256 B (64 floats) out of a table of n_rows * 2048 B; 16 lanes per slice
__global__ void slice256(const float* __restrict__ tab, const uint32_t* __restrict__ idx, float4* __restrict__ sink, size_t n_slices) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t s = t / 16;
    float4 acc = make_float4(0, 0, 0, 0);
    for (; s < n_slices; s += (size_t)gridDim.x * blockDim.x / 16) {
        const uint32_t r = idx[s];                                 // slice id: row * 8 + factor
        float4 v = *reinterpret_cast<const float4*>(tab + (size_t)r * 64 + (t % 16) * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 12345.678f) sink[0] = acc;
}

__global__ void dword_rand(const float* __restrict__ tab, const uint32_t* __restrict__ idx, float* __restrict__ sink, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += tab[(size_t)idx[i] * 8 + (idx[i] & 7)];   // node * K + factor: 32-B records
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ void store16(float4* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5; return rng_state; }

int main() {
    const size_t table_bytes = (size_t)3 << 30;                    // 3 GiB: far beyond the 256 MiB Infinity Cache
    const size_t n16 = table_bytes / 16;
    float4 *tab, *sink;
    CHECK(hipMalloc(&tab, table_bytes));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(tab, 0, table_bytes));
    const size_t n_slices = 6u << 20, n_dw = 6u << 20;             // ~ the edge count of the hbm_bound workload
    const uint32_t slices_in_table = (uint32_t)(table_bytes / 256);
    const uint32_t nodes_in_s = 730980;                            // s [N][8] fp32 of the hbm_bound workload: 23 MB
    uint32_t* h = (uint32_t*)malloc(n_slices * 4);
    uint32_t *d_idx_s, *d_idx_d;
    CHECK(hipMalloc(&d_idx_s, n_slices * 4));
    CHECK(hipMalloc(&d_idx_d, n_dw * 4));
    for (size_t i = 0; i < n_slices; ++i) h[i] = rnd() % slices_in_table;
    CHECK(hipMemcpy(d_idx_s, h, n_slices * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < n_dw; ++i) h[i] = rnd() % nodes_in_s;
    CHECK(hipMemcpy(d_idx_d, h, n_dw * 4, hipMemcpyHostToDevice));
    float* s_tab;
    CHECK(hipMalloc(&s_tab, (size_t)nodes_in_s * 32));
    CHECK(hipMemset(s_tab, 0, (size_t)nodes_in_s * 32));
    for (int rep = 0; rep < 3; ++rep) {
        stream16<<<4096, 256>>>(tab, sink, n16);
        slice256<<<4096, 256>>>((const float*)tab, d_idx_s, sink, n_slices);
        // between two dword_rand launches 3 GiB stream through: the 23 MB s table is NOT cache-resident at launch
        dword_rand<<<4096, 256>>>(s_tab, d_idx_d, (float*)sink, n_dw);
        store16<<<4096, 256>>>(tab, n16 / 2);
        CHECK(hipDeviceSynchronize());
    }
    printf("known bytes per launch: stream16 %zu  slice256 %zu (+ %zu of indices)  dword_rand needed %zu, lines of 128 B %zu, 64 B %zu (+ %zu of indices)  store16 %zu\n",
           table_bytes, n_slices * 256, n_slices * 4, n_dw * 4, n_dw * 128, n_dw * 64, n_dw * 4, table_bytes / 2);
    return 0;
}
