// Which property of a resident gate-wait wave slows kernels of a concurrently running stream?  (profiles/r06_notes.md section 9)
// Variants of the spinner: kind 0 = poll a word with agent-scope loads + s_sleep(64) (what eve_gate_wait does);
// 1 = s_sleep only, one final check per 4096 sleeps (no memory traffic); 2 = poll with s_sleep(127) x 8 between loads;
// 3 = poll with plain (non-atomic, volatile) loads; 4 = kind 0 at low wave priority (s_setprio 0 is the default; here: explicit 0
// and a long s_sleep first).  All stop when *flag != 0 or after `max_iters`.
#include <hip/hip_runtime.h>
__global__ void spin_kernel(const unsigned* flag, int kind, unsigned max_iters, unsigned* out) {
    if (threadIdx.x != 0) return;
    unsigned it = 0;
    for (; it < max_iters; ++it) {
        if (kind == 1) {
            __builtin_amdgcn_s_sleep(64);
            if ((it & 4095) == 4095 && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            continue;
        }
        unsigned v;
        if (kind == 3) v = *reinterpret_cast<const volatile unsigned*>(flag);
        else v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v) break;
        if (kind == 2) { for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(127); }
        else __builtin_amdgcn_s_sleep(64);
    }
    *out = it;
}
__global__ void set_kernel(unsigned* flag, unsigned v) { __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
extern "C" int spin_launch(const unsigned* flag, int kind, unsigned max_iters, unsigned* out, int threads, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, flag, kind, max_iters, out);
    return (int)hipGetLastError();
}
extern "C" int spin_set(unsigned* flag, unsigned v, void* stream) {
    hipLaunchKernelGGL(set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, v);
    return (int)hipGetLastError();
}
