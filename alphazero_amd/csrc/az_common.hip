// az_common.hip -- error reporting shared by every entry point of libaz_amd.so
#include "az_host.h"

static thread_local char g_err[512] = "";

void az_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *az_last_error(void) { return g_err; }
extern "C" int az_version(void) { return 104; }  // 104: az_trainer_steps takes n_samples, az_trainer_check, az_net_profile_read has eight slots; 103: az_engine_search_begin / _end / _pair; 102: az_trainer_*, az_engine_nodes_used / grow_pools, az_net_stage_kernel
