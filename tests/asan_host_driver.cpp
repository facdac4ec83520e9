// ASan/UBSan driver for the host-side graph preparation of the C ABI (no GPU involved).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "disenlink_hip.h"
int main() {
    std::mt19937_64 rng(1);
    for (int it = 0; it < 300; ++it) {
        const int n = 1 + (int)(rng() % 60);
        const long e = (long)(rng() % (4 * n + 1));
        std::vector<int64_t> src(e), dst(e);
        for (long i = 0; i < e; ++i) { src[i] = rng() % n; dst[i] = rng() % n; }
        dl_host_csr hc;
        if (dl_host_csr_from_edges(src.data(), dst.data(), e, n, 1, &hc) != 0) { printf("csr failed: %s\n", dl_last_error()); return 1; }
        const int seg = 1 + (int)(rng() % 9);
        const int sl[] = {1, 8, 16, 3};
        dl_host_plan hp;
        if (dl_host_plan_build(n, n, hc.rowptr, hc.n_entries ? hc.col : nullptr, seg, sl[rng() % 4], nullptr, (rng() & 1) ? 1 : DL_UNIT_SEGS, (int32_t)(rng() & 1), &hp) != 0) { printf("plan failed: %s\n", dl_last_error()); return 1; }
        // keep mask: entries with col >= row (one contiguous run per row)
        std::vector<uint8_t> keep(hc.n_entries > 0 ? hc.n_entries : 1, 0);
        for (int r = 0; r < n; ++r) for (int x = hc.rowptr[r]; x < hc.rowptr[r + 1]; ++x) keep[x] = hc.col[x] >= r;
        dl_host_plan hk;
        if (dl_host_plan_build(n, n, hc.rowptr, hc.n_entries ? hc.col : nullptr, seg, 8, keep.data(), 1, (int32_t)(rng() & 1), &hk) != 0) { printf("kept plan failed: %s\n", dl_last_error()); return 1; }
        dl_host_plan_free(&hk);
        dl_host_plan_free(&hp);
        dl_host_csr_free(&hc);
    }
    printf("host builders: 300 random graphs clean\n");
    return 0;
}
