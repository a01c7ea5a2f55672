// Register-resident InstanceNorm forward / backward for planes that fit one workgroup's registers
// (H*W*C <= 64 Ki elements: every EyeNet stage after the stem, every RefineNet level below 36x64).
// One workgroup owns one image plane: it reads it ONCE from HBM, keeps the packed vectors in VGPRs,
// does the per-channel reductions with wave shuffles + one LDS exchange, and writes the result.
//   forward : 1 read (+1 for the residual) + 1 write     (was: stats pass + apply pass = 2 reads + 1 write)
//   backward: 3 reads (dy, x, y) + 1-2 writes             (was: two passes over all three = 6 reads)
// Statistics are the exact two-pass form (mean, then centred second moment) over registers.
#include "common.h"

namespace eve {

// Sum s[0..VEC) over all threads of the block that share this thread's channel vector (cv = tid % cvecs).
// On return every thread holds the totals for its channels.
template <int VEC>
__device__ __forceinline__ void plane_allreduce(float* s, float* sh, int tid, int nthreads, int cvecs) {
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    for (int o = 32; o >= cvecs; o >>= 1)
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] += __shfl_xor(s[e], o, 64);
    __syncthreads();                                   // sh may still be read from a previous call
#pragma unroll
    for (int e = 0; e < VEC; ++e) sh[(wave * 64 + lane) * VEC + e] = s[e];
    __syncthreads();
    const int wstep = cvecs > 64 ? cvecs / 64 : 1;     // waves that cover the same channel vectors
    const int src_lane = cvecs >= 64 ? lane : (lane % cvecs);
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    for (int w = wave % wstep; w < nwaves; w += wstep)
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] += sh[(w * 64 + src_lane) * VEC + e];
}

template <typename T, int VPT>
__global__ __launch_bounds__(1024) void in_fwd_fused_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const T* __restrict__ res,
                                                            int act, T* __restrict__ y, float* __restrict__ mr,
                                                            int HW, int C, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh[1024 * VEC];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int cvecs = C / VEC, nvec = HW * cvecs;
    const int cv = tid % cvecs;
    const size_t base = (size_t)blockIdx.x * nvec;
    uint4 q[VPT];
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = tid + j * nthreads;
        q[j] = i < nvec ? reinterpret_cast<const uint4*>(x)[base + i] : make_uint4(0, 0, 0, 0);
        float f[VEC];
        Elem<T>::unpack(q[j], f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] += f[e];
    }
    plane_allreduce<VEC>(s, sh, tid, nthreads, cvecs);
    float mean[VEC], a[VEC], b[VEC];
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mean[e] = s[e] * inv; s[e] = 0.f; }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        if (tid + j * nthreads < nvec) {
            float f[VEC];
            Elem<T>::unpack(q[j], f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = f[e] - mean[e]; s[e] += d * d; }
        }
    }
    plane_allreduce<VEC>(s, sh, tid, nthreads, cvecs);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const float rstd = rsqrtf(s[e] * inv + eps);
        const int c = cv * VEC + e;
        a[e] = rstd; b[e] = -mean[e] * rstd;
        if (gamma) { a[e] *= gamma[c]; b[e] = b[e] * gamma[c] + beta[c]; }
        if (tid < cvecs) {
            mr[((size_t)blockIdx.x * C + c) * 2] = mean[e];
            mr[((size_t)blockIdx.x * C + c) * 2 + 1] = rstd;
        }
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = tid + j * nthreads;
        if (i < nvec) {
            float f[VEC], r[VEC];
            Elem<T>::unpack(q[j], f);
            if (res) Elem<T>::unpack(reinterpret_cast<const uint4*>(res)[base + i], r);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float z = f[e] * a[e] + b[e];
                if (res) z += r[e];
                f[e] = act_fwd(z, act);
            }
            reinterpret_cast<uint4*>(y)[base + i] = Elem<T>::pack(f);
        }
    }
}

template <typename T, int VPT>
__global__ __launch_bounds__(1024) void in_bwd_fused_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                            const T* __restrict__ x, const float* __restrict__ mr,
                                                            const float* __restrict__ gamma, int act,
                                                            T* __restrict__ dx, T* __restrict__ dres,
                                                            float* __restrict__ sums, int HW, int C) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh[1024 * VEC];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int cvecs = C / VEC, nvec = HW * cvecs;
    const int cv = tid % cvecs;
    const size_t base = (size_t)blockIdx.x * nvec;
    float mean[VEC], rstd[VEC];
    {
        const float* m = mr + ((size_t)blockIdx.x * C + cv * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { mean[e] = m[2 * e]; rstd[e] = m[2 * e + 1]; }
    }
    uint4 qg[VPT], qx[VPT];          // g = dy * act'(y) re-packed, and x
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = tid + j * nthreads;
        qg[j] = make_uint4(0, 0, 0, 0);
        qx[j] = make_uint4(0, 0, 0, 0);
        if (i < nvec) {
            float g[VEC], xx[VEC];
            Elem<T>::unpack(reinterpret_cast<const uint4*>(dy)[base + i], g);
            qx[j] = reinterpret_cast<const uint4*>(x)[base + i];
            Elem<T>::unpack(qx[j], xx);
            if (act != EVE_ACT_NONE) {
                float yy[VEC];
                Elem<T>::unpack(reinterpret_cast<const uint4*>(y)[base + i], yy);
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[e] *= act_grad_from_out(yy[e], act);
                qg[j] = Elem<T>::pack(g);
                if (dres) reinterpret_cast<uint4*>(dres)[base + i] = qg[j];
                // keep the arithmetic on the SAME rounded g the two-pass kernel and dres see
                Elem<T>::unpack(qg[j], g);
            } else {
                qg[j] = Elem<T>::pack(g);
                if (dres) reinterpret_cast<uint4*>(dres)[base + i] = qg[j];
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                s1[e] += g[e];
                s2[e] += g[e] * (xx[e] - mean[e]) * rstd[e];
            }
        }
    }
    plane_allreduce<VEC>(s1, sh, tid, nthreads, cvecs);
    plane_allreduce<VEC>(s2, sh, tid, nthreads, cvecs);
    if (sums && tid < cvecs) {
        float* o = sums + ((size_t)blockIdx.x * C + cv * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { o[2 * e] = s1[e]; o[2 * e + 1] = s2[e]; }
    }
    float k[VEC];
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        s1[e] *= inv; s2[e] *= inv;
        k[e] = rstd[e] * (gamma ? gamma[cv * VEC + e] : 1.f);
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = tid + j * nthreads;
        if (i < nvec) {
            float g[VEC], xx[VEC];
            Elem<T>::unpack(qg[j], g);
            Elem<T>::unpack(qx[j], xx);
#pragma unroll
            for (int e = 0; e < VEC; ++e) g[e] = k[e] * (g[e] - s1[e] - (xx[e] - mean[e]) * rstd[e] * s2[e]);
            reinterpret_cast<uint4*>(dx)[base + i] = Elem<T>::pack(g);
        }
    }
}

// threads per block and vectors per thread for a plane of nvec 16-byte vectors; false if it does not fit
static bool fused_plan(int nvec, int cvecs, int& threads, int& vpt) {
    if (nvec <= 0 || cvecs <= 0 || cvecs > 128 || (cvecs & (cvecs - 1))) return false;
    vpt = 1;
    while (vpt < 8 && (nvec + vpt - 1) / vpt > 1024) vpt *= 2;
    if ((nvec + vpt - 1) / vpt > 1024) return false;
    const int q = cvecs > 64 ? cvecs : 64;
    threads = ((nvec + vpt - 1) / vpt + q - 1) / q * q;
    return threads <= 1024;
}

}  // namespace eve

using namespace eve;

#define LAUNCH_VPT(KERNEL, T, ...)                                                                        \
    switch (vpt) {                                                                                        \
        case 1: hipLaunchKernelGGL((KERNEL<T, 1>), dim3(N), dim3(threads), 0, s, __VA_ARGS__); break;      \
        case 2: hipLaunchKernelGGL((KERNEL<T, 2>), dim3(N), dim3(threads), 0, s, __VA_ARGS__); break;      \
        case 4: hipLaunchKernelGGL((KERNEL<T, 4>), dim3(N), dim3(threads), 0, s, __VA_ARGS__); break;      \
        default: hipLaunchKernelGGL((KERNEL<T, 8>), dim3(N), dim3(threads), 0, s, __VA_ARGS__); break;     \
    }

/* returns 0 on launch, -1 if the plane does not fit the fused kernel (caller falls back), >0 on error */
extern "C" int eve_instnorm_fwd_fused(int dtype, int N, int HW, int C, const void* x, const float* gamma,
                                      const float* beta, const void* res, int act, float eps, void* y,
                                      float* mean_rstd, eve_stream_t stream) {
    const int vec = dtype == EVE_DT_BF16 ? 8 : 4;
    if ((dtype != EVE_DT_F32 && dtype != EVE_DT_BF16) || N <= 0 || HW <= 0 || C <= 0 || C % vec || !x || !y ||
        !mean_rstd || ((gamma == nullptr) != (beta == nullptr)))
        return set_error_msg("instnorm_fwd_fused: bad arguments");
    int threads, vpt;
    if (!fused_plan(HW * (C / vec), C / vec, threads, vpt)) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) {
        LAUNCH_VPT(in_fwd_fused_kernel, bf16_t, (const bf16_t*)x, gamma, beta, (const bf16_t*)res, act, (bf16_t*)y,
                   mean_rstd, HW, C, eps)
    } else {
        LAUNCH_VPT(in_fwd_fused_kernel, float, (const float*)x, gamma, beta, (const float*)res, act, (float*)y,
                   mean_rstd, HW, C, eps)
    }
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_instnorm_bwd_fused(int dtype, int N, int HW, int C, const void* dy, const void* y, const void* x,
                                      const float* mean_rstd, const float* gamma, int act, void* dx, void* dres,
                                      float* sums, eve_stream_t stream) {
    const int vec = dtype == EVE_DT_BF16 ? 8 : 4;
    if ((dtype != EVE_DT_F32 && dtype != EVE_DT_BF16) || N <= 0 || HW <= 0 || C <= 0 || C % vec || !dy || !x ||
        !mean_rstd || !dx || (act != EVE_ACT_NONE && !y))
        return set_error_msg("instnorm_bwd_fused: bad arguments");
    int threads, vpt;
    if (!fused_plan(HW * (C / vec), C / vec, threads, vpt)) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) {
        LAUNCH_VPT(in_bwd_fused_kernel, bf16_t, (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, mean_rstd, gamma,
                   act, (bf16_t*)dx, (bf16_t*)dres, sums, HW, C)
    } else {
        LAUNCH_VPT(in_bwd_fused_kernel, float, (const float*)dy, (const float*)y, (const float*)x, mean_rstd, gamma,
                   act, (float*)dx, (float*)dres, sums, HW, C)
    }
    EVE_CHECK_LAUNCH();
    return 0;
}
