// The one place that reads the environment (dl_config.h).
#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include "dl_config.h"
#include "disenlink_hip.h"

namespace dl {
namespace {

Config g_config;
std::once_flag g_once;

bool flag(const char* name) {                 // set and not "0" / "" ; the historical switches were "set = on"
    const char* e = getenv(name);
    return e != nullptr && e[0] != '\0' && !(e[0] == '0' && e[1] == '\0');
}
long long number(const char* name) {
    const char* e = getenv(name);
    return e != nullptr && e[0] != '\0' ? atoll(e) : 0;
}

void read_environment() {
    Config c;
    const char* sr = getenv("DL_STREAM_ROWS");
    c.stream_rows = (sr == nullptr || sr[0] == '\0') ? -1 : (sr[0] == '1' ? 1 : 0);
    c.route_ballot = flag("DL_ROUTE_BALLOT");
    c.train_group_kernel = flag("DL_TRAIN_GROUP_KERNEL");
    c.fwd_group_kernel = flag("DL_FWD_GROUP_KERNEL");
    c.auc_target = (int)std::max(0LL, number("DL_AUC_TARGET"));
    c.project_fp32_mfma = getenv("DL_PROJECT_FP32_MFMA") != nullptr;
    c.fwd_groups = (int)std::max(0LL, number("DL_FWD_GROUPS"));
    c.fwd_block_rows = std::max(0LL, number("DL_FWD_BLOCK_ROWS"));
    c.bwd_block_bytes = std::max(0LL, number("DL_BWD_BLOCK_BYTES"));
    c.bwd_target = (int)std::max(0LL, number("DL_BWD_TARGET"));
    c.dense_fp32_mfma = getenv("DL_DENSE_FP32_MFMA") != nullptr;
    c.dense_dc32 = getenv("DL_DENSE_DC32") != nullptr;
    const char* ic = getenv("DL_INKERNEL_COMBINE");
    c.inkernel_combine = (ic == nullptr || ic[0] == '\0') ? 1 : std::max(0, std::min(2, atoi(ic)));
    g_config = c;
}

}  // namespace

const Config& config() {
    std::call_once(g_once, read_environment);
    return g_config;
}

void config_reload() {
    std::call_once(g_once, read_environment);
    read_environment();
}

}  // namespace dl

extern "C" void dl_config_reload(void) { dl::config_reload(); }
