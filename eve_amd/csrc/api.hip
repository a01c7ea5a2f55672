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
// caller-owned device scratch (eve_set_workspace): the library never allocates; kernels that can use scratch (split-K
// partial sums) fall back to their scratch-free form when it is absent or too small
void* g_workspace = nullptr;
unsigned long long g_workspace_bytes = 0;
int set_error_msg(const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return 1;
}
}  // namespace eve

extern "C" int eve_abi_version(void) { return EVE_ABI_VERSION; }
extern "C" int eve_set_workspace(void* device_ptr, unsigned long long bytes) {
    if ((device_ptr == nullptr) != (bytes == 0) || ((uintptr_t)device_ptr & 15)) return eve::set_error_msg("set_workspace: bad arguments");
    eve::g_workspace = device_ptr;
    eve::g_workspace_bytes = bytes;
    return 0;
}
extern "C" const char* eve_last_error(void) { return eve::g_err; }
extern "C" const char* eve_last_kernel(void) { return eve::g_last_kernel; }
