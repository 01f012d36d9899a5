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

}  // extern "C"
