// ABI version and thread-local error reporting for libeve_hip.so.
#include <stdio.h>
#include <string.h>

#include "common.h"

namespace eve {
static thread_local char g_err[512] = "";
thread_local const char* g_last_kernel = "";
int set_error(hipError_t e, const char* where) {
    snprintf(g_err, sizeof(g_err), "%s: HIP error %d (%s)", where, (int)e, hipGetErrorString(e));
    return 100 + (int)e;
}
int set_error_msg(const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return 1;
}
}  // namespace eve

extern "C" int eve_abi_version(void) { return EVE_ABI_VERSION; }
extern "C" const char* eve_last_error(void) { return eve::g_err; }
extern "C" const char* eve_last_kernel(void) { return eve::g_last_kernel; }
