// Environment switches of the library, read ONCE (at the first call that needs one) into a static structure: no launch
// path calls getenv.  dl_config_reload() (C ABI) reads the environment again — tests and A/B scripts that flip a switch
// inside one process call it after changing the variable.  Every switch is a measurement / test knob: defaults are what
// ships (README.md lists the few a maintainer would set, tools/README.md the rest).
#pragma once

namespace dl {

struct Config {
    int stream_rows;            // DL_STREAM_ROWS: -1 = by table size (default), 0 / 1 = never / always stream the H rows
    bool route_ballot;          // DL_ROUTE_BALLOT=1: ballot arg-max in the router (measured slower)
    bool train_group_kernel;    // DL_TRAIN_GROUP_KERNEL: group-per-entry one-pass scorer instead of the wave-per-entry ones
    bool fwd_group_kernel;      // DL_FWD_GROUP_KERNEL: group-per-entry forward scorer instead of the wave-per-entry one
    int auc_target;             // DL_AUC_TARGET: workgroups of the AUC count kernel (0 = default)
    bool project_fp32_mfma;     // DL_PROJECT_FP32_MFMA: plain fp32 MFMA projection instead of the three-plane products
    int fwd_groups;             // DL_FWD_GROUPS: hidden-chunk groups of the projection forward (0 = default)
    long long fwd_block_rows;   // DL_FWD_BLOCK_ROWS: x-plane node block, in tiles of 128 rows (0 = default; tests force blocking)
    long long bwd_block_bytes;  // DL_BWD_BLOCK_BYTES: cap of the per-block hidden gradient (0 = default; tests force blocking)
    int bwd_target;             // DL_BWD_TARGET: workgroups per launch of the projection backward (0 = default)
    bool dense_fp32_mfma;       // DL_DENSE_FP32_MFMA: dense scorer on fp32 MFMA
    bool dense_dc32;            // DL_DENSE_DC32: dense scorer in 32-feature steps
    int inkernel_combine;       // DL_INKERNEL_COMBINE: 0 = rows of several units always through the separate combine launch,
                                // 1 (default) = inside the launch where the plan's rows are few units long, 2 = wherever a kernel can
};

const Config& config();
void config_reload();

}  // namespace dl
