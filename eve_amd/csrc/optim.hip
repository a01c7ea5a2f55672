// Optimiser step over flat float buffers: global grad-norm + Adam with coupled L2 weight decay.
// Reference semantics: nn.utils.clip_grad_norm_(params, 5.0) (src/core/training.py:492-498) followed by
// torch.optim.Adam(lr, weight_decay) .step() (src/train.py:49-55): g <- g + wd*p; m,v moments with bias
// correction; p <- p - lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
#include "common.h"

namespace eve {

// Two fixed-order stages (per-workgroup partials, then one workgroup summing them in index order): the result is
// bit-reproducible for a given n, which data-parallel ranks rely on -- every rank clips identical reduced gradients by the
// identical factor, so the replicas stay bit-identical (an atomic accumulation would differ in the last bits from run to run).
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, float* __restrict__ part, long long n) {
    float s = 0.f;
    const long long nvec = n / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (long long i = nvec * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        s += g[i] * g[i];
    s = wave_sum(s);
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = wave_sum(s);
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] += (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// One thread decides the step: is the gradient norm finite (float16 training with a loss scale), which clip factor, which
// bias-correction exponent.  The optimiser-step counter lives in the guard and advances HERE, only when the step is taken --
// a skipped step leaves weights, moments AND the counter as they were (round 3 advanced the counter on the host before the
// kernel could refuse).  Loss-scale policy (check_finite only): halve after two consecutive skips (floor 1), double after
// 2 000 consecutive taken steps (ceiling 65 536) -- torch.cuda.amp.GradScaler's shape; the reference is float32 and has none.
__global__ void adam_prepare_kernel(eve_adam_guard* __restrict__ gd, const float* __restrict__ sumsq, float max_norm, float gscale,
                                    float b1, float b2, int check_finite, const float* __restrict__ poison) {
    if (poison && !(*poison == 0.f)) {
        // a stream gate of this step's gradient exchange timed out (gate_wait_kernel): some bucket was all-reduced before its
        // last gradient was written.  The slot is part of the all-reduced buffer, so EVERY rank reads non-zero here and skips;
        // nothing else changes (the gradient was not an overflow: the loss scale stays, and so does the run of good steps).
        gd->skipped_total += 1;
        gd->skipped_gate += 1;
        gd->applied = 0.f;
        return;
    }
    const float ls = gd->loss_scale > 0.f ? gd->loss_scale : 1.f;
    const float gs = gscale / ls;                      // the gradients in the buffer are loss_scale x the true ones
    float clip = gs;
    bool skip = false;
    if (sumsq) {
        const float ss = *sumsq;
        if (check_finite && !(ss < 3.0e38f)) skip = true;           // inf / NaN norm: an overflowed float16 gradient
        else if (max_norm > 0.f) {
            const float total = sqrtf(ss) * gs;
            const float c = max_norm / (total + 1e-6f);
            clip = gs * (c < 1.f ? c : 1.f);
        }
    }
    if (skip) {
        gd->skipped_total += 1;
        gd->skipped_run += 1;
        gd->good_run = 0;
        gd->applied = 0.f;
        if (gd->skipped_run >= 2) { gd->loss_scale = ls * 0.5f > 1.f ? ls * 0.5f : 1.f; gd->skipped_run = 0; }
        return;
    }
    const int t = gd->step + 1;
    gd->step = t;
    gd->skipped_run = 0;
    gd->good_run += 1;
    if (check_finite && gd->good_run >= 2000) { gd->loss_scale = ls * 2.f < 65536.f ? ls * 2.f : 65536.f; gd->good_run = 0; }
    gd->applied = 1.f;
    gd->clip = clip;
    gd->bc1 = 1.f - powf(b1, (float)t);
    gd->bc2_sqrt = sqrtf(1.f - powf(b2, (float)t));
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ sumsq, float max_norm, float gscale,
                                                   float lr, float b1, float b2, float eps, float wd, float bc1,
                                                   float bc2_sqrt, const eve_adam_guard* __restrict__ gd,
                                                   const float* __restrict__ lr_dev, long long n) {
    if (lr_dev) lr = *lr_dev;      // the schedule's value for this step, written by the host before the (replayed) launch
    float clip = gscale;
    if (gd) {                      // everything was decided by adam_prepare_kernel (same stream, just before)
        if (gd->applied == 0.f) return;                 // skipped step (wave-uniform exit)
        clip = gd->clip; bc1 = gd->bc1; bc2_sqrt = gd->bc2_sqrt;
    } else if (sumsq) {
        const float total = sqrtf(*sumsq) * gscale;
        const float c = max_norm / (total + 1e-6f);
        clip = gscale * (c < 1.f ? c : 1.f);
    }
    const float step = lr / bc1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float pi = p[i];
        const float gi = clip * g[i] + wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

// ---- stream gates: a point of one stream's (captured) work releases work queued on another stream --------------------------
// signal: one release + agent-scope increment of *flag, as a kernel node behind the work it publishes;  wait: one lane polls
// *flag with relaxed agent-scope loads and s_sleep until it has reached `value` (wrap-safe compare), then acquires.  The poll
// is BOUNDED (max_polls sleeps of 64 x 64 clocks; default 2^21: seconds): a gate whose signal never comes counts a time-out,
// POISONS the step (*poison = +inf: a float inside the last all-reduced gradient bucket, read by adam_prepare_kernel on every
// rank after the exchange) and lets the stream go on instead of hanging the device; the host reads *timeouts when it likes.
__global__ void gate_signal_kernel(unsigned* flag) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void gate_wait_kernel(const unsigned* flag, unsigned value, const unsigned* value_ref, unsigned* timeouts, float* poison,
                                 unsigned max_polls) {
    if (threadIdx.x != 0) return;
    if (value_ref) value = __hip_atomic_load(value_ref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (written earlier on this stream)
    for (unsigned it = 0; it < max_polls; ++it) {
        const unsigned v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - value) >= 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            return;
        }
        __builtin_amdgcn_s_sleep(64);
    }
    __hip_atomic_fetch_add(timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (poison) *poison = __builtin_inff();        // (same stream as the collective that follows: ordinary stream order)
}

}  // namespace eve

using namespace eve;

extern "C" int eve_gate_signal(unsigned* flag, eve_stream_t stream) {
    if (!flag) return set_error_msg("gate_signal: null flag");
    hipLaunchKernelGGL(gate_signal_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_gate_wait(const unsigned* flag, unsigned value, const unsigned* value_ref, unsigned* timeouts, float* poison,
                             unsigned max_polls, eve_stream_t stream) {
    if (!flag || !timeouts) return set_error_msg("gate_wait: null pointer");
    if (max_polls == 0) max_polls = g_cfg.gate_wait_polls > 0 ? (unsigned)g_cfg.gate_wait_polls : (1u << 21);
    hipLaunchKernelGGL(gate_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, value, value_ref, timeouts, poison, max_polls);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_sumsq(long long n, const float* g, float* out, float* workspace, eve_stream_t stream) {
    if (n <= 0 || !g || !out || !workspace) return set_error_msg("sumsq: bad arguments");
    long long b = (n / 4 + 255) / 256;
    if (b > EVE_SUMSQ_WORKSPACE) b = EVE_SUMSQ_WORKSPACE;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, g, workspace, n);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (int)b, out);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_adam_step(long long n, float* p, const float* g, float* m, float* v, const float* sumsq,
                             float max_norm, float gscale, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int step, eve_adam_guard* guard, int check_finite, const float* lr_dev,
                             const float* poison, eve_stream_t stream) {
    if (n <= 0 || !p || !g || !m || !v || (step < 1 && !guard)) return set_error_msg("adam_step: bad arguments");
    if (poison && !guard) return set_error_msg("adam_step: a poison word needs the device-resident guard");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = sqrtf(1.f - powf(beta2, (float)step));
    long long b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    if (guard)
        hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, guard, sumsq, max_norm, gscale, beta1, beta2, check_finite, poison);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, p, g, m, v, sumsq, max_norm,
                       gscale, lr, beta1, beta2, eps, weight_decay, bc1, bc2, (const eve_adam_guard*)guard, lr_dev, n);
    EVE_CHECK_LAUNCH();
    return 0;
}
