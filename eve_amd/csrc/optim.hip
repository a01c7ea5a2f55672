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

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ sumsq, float max_norm, float gscale,
                                                   float lr, float b1, float b2, float eps, float wd, float bc1,
                                                   float bc2_sqrt, const int* __restrict__ step_dev,
                                                   const float* __restrict__ lr_dev, long long n) {
    if (step_dev) {      // step counter lives on the device (hipGraph replay cannot change kernel arguments)
        const float t = (float)(*step_dev);
        bc1 = 1.f - powf(b1, t);
        bc2_sqrt = sqrtf(1.f - powf(b2, t));
    }
    if (lr_dev) lr = *lr_dev;      // the schedule's value for this step, written by the host before the (replayed) launch
    float clip = gscale;
    if (sumsq) {
        // float16 training (static loss scale, train.Trainer): a gradient that overflowed the format shows up as an
        // infinite / NaN norm -- the step is skipped, weights and moments stay as they are (wave-uniform exit)
        if (!(*sumsq < 3.0e38f)) return;
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

}  // namespace eve

using namespace eve;

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
                             float weight_decay, int step, const int* step_dev, const float* lr_dev,
                             eve_stream_t stream) {
    if (n <= 0 || !p || !g || !m || !v || (step < 1 && !step_dev)) return set_error_msg("adam_step: bad arguments");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = sqrtf(1.f - powf(beta2, (float)step));
    long long b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, p, g, m, v, sumsq, max_norm,
                       gscale, lr, beta1, beta2, eps, weight_decay, bc1, bc2, step_dev, lr_dev, n);
    EVE_CHECK_LAUNCH();
    return 0;
}
