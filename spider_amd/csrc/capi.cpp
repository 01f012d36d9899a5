// Error plumbing and library identity for libspider_hip.so (C ABI; see include/spider_hip.h).
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" {

void spider_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

const char* spider_last_error(void) { return g_err; }

int spider_abi_version(void) { return 4; }

const char* spider_target_arch(void) { return "gfx950"; }

// split-K combine form of the weight-stationary streaming conv (gemm.hip, w_tiled = 2), shared by the bf16 and the f16 instantiation:
// -1 = not chosen yet (SPIDER_WS_INLAUNCH is read at the first such launch), 1 = inside the launch (last-arriving block), 0 = partial
// slabs + the reduce kernel
int g_spider_ws_inlaunch = -1;

int spider_set_ws_inlaunch(int on) {
    const int prev = g_spider_ws_inlaunch;
    g_spider_ws_inlaunch = on ? 1 : 0;
    return prev;
}

}  // extern "C"
