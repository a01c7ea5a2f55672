// RefineNet's output head and the heat-map losses of the train step.
//
//   head    heatmap_final = sigmoid(final conv)   (/root/reference/src/models/refine_net.py:221-223,255): the last 1x1
//           convolution leaves its logits NHWC in the compute dtype; the sigmoid is evaluated in FLOAT straight into the
//           float [N][1][H][W] map the caller's dict holds (a bf16 sigmoid saturates to exactly 1.0 above 0.998, where
//           F.binary_cross_entropy hits its -100 clamp and y(1-y) is 0; the float32 reference does neither).
//   losses  loss_ce_heatmap_* / loss_mse_heatmap_final (src/models/eve.py:350-360): per frame the mean over the map of
//           F.binary_cross_entropy (src/losses/cross_entropy.py:27-35; log terms clamped at -100, gradient
//           (p - g) / max(p (1 - p), 1e-12) as ATen computes it) or of (p - g)^2 (src/losses/mse.py), then the
//           per-clip validity reduction of src/losses/base_loss_with_validity.py:64-73 (sum over valid frames / number
//           of valid frames when that exceeds one; mean over clips).  One workgroup per map; the B x T reduction is one
//           small workgroup that also emits the per-frame weights the backward multiplies with.
#include "common.h"

namespace eve {

template <typename T>
__global__ __launch_bounds__(256) void heatmap_head_fwd_kernel(const T* __restrict__ logits, float* __restrict__ out,
                                                               int Cpad, long long items) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const float z = Elem<T>::ld(logits + i * Cpad);
        out[i] = 1.f / (1.f + expf(-z));
    }
}

// d logits[pixel][c] = c == 0 ? dy * y * (1 - y) : 0   (one 16-byte vector per thread)
template <typename T>
__global__ __launch_bounds__(256) void heatmap_head_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                               T* __restrict__ dlogits, int cvecs, long long nvec) {
    constexpr int VEC = Elem<T>::VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] = 0.f;
        if (i % cvecs == 0) {
            const long long pix = i / cvecs;
            const float p = y[pix];
            f[0] = dy[pix] * p * (1.f - p);
        }
        reinterpret_cast<uint4*>(dlogits)[i] = Elem<T>::pack(f);
    }
}

__device__ __forceinline__ float block_sum256(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

// per_map[m] = mean over the map of BCE (kind 0) or squared error (kind 1)
__global__ __launch_bounds__(256) void heatmap_loss_map_kernel(int kind, int HW, const float* __restrict__ pred,
                                                               const float* __restrict__ gt, float* __restrict__ per_map) {
    __shared__ float sh[4];
    const size_t base = (size_t)blockIdx.x * HW;
    float acc = 0.f;
    for (int i = threadIdx.x * 4; i < HW; i += 1024) {
        float p[4], g[4];
        if (i + 3 < HW && ((base + i) & 3) == 0) {
            const float4 a = *reinterpret_cast<const float4*>(pred + base + i), b = *reinterpret_cast<const float4*>(gt + base + i);
            p[0] = a.x; p[1] = a.y; p[2] = a.z; p[3] = a.w; g[0] = b.x; g[1] = b.y; g[2] = b.z; g[3] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const bool in = i + e < HW; p[e] = in ? pred[base + i + e] : 0.5f; g[e] = in ? gt[base + i + e] : 0.5f; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (i + e >= HW) continue;
            if (kind == 0) acc += (g[e] - 1.f) * fmaxf(log1pf(-p[e]), -100.f) - g[e] * fmaxf(logf(p[e]), -100.f);
            else { const float d = p[e] - g[e]; acc += d * d; }
        }
    }
    acc = block_sum256(acc, sh);
    if (threadIdx.x == 0) per_map[blockIdx.x] = acc / (float)HW;
}

// loss[0] = mean_b( sum_t v x / (n_b > 1 ? n_b : 1) );  w[b][t] = v / (den_b * B)
__global__ __launch_bounds__(256) void masked_clip_mean_kernel(int B, int T, const float* __restrict__ per_step,
                                                               const uint8_t* __restrict__ valid, float* __restrict__ loss,
                                                               float* __restrict__ w) {
    __shared__ float sh[4];
    float total = 0.f;
    for (int b = 0; b < B; ++b) {
        float s = 0.f, n = 0.f;
        for (int t = threadIdx.x; t < T; t += 256)
            if (valid[(size_t)b * T + t]) { s += per_step[(size_t)b * T + t]; n += 1.f; }
        s = block_sum256(s, sh);
        n = block_sum256(n, sh);
        const float den = n > 1.f ? n : 1.f;
        total += s / den;
        for (int t = threadIdx.x; t < T; t += 256)
            w[(size_t)b * T + t] = valid[(size_t)b * T + t] ? 1.f / (den * (float)B) : 0.f;
    }
    if (threadIdx.x == 0) loss[0] = total / (float)B;
}

// dpred = upstream * w[m] / HW * d(per-element loss)/d p
__global__ __launch_bounds__(256) void heatmap_loss_bwd_kernel(int kind, int HW, const float* __restrict__ pred,
                                                               const float* __restrict__ gt, const float* __restrict__ w,
                                                               const float* __restrict__ upstream, float* __restrict__ dpred) {
    const size_t base = (size_t)blockIdx.x * HW;
    const float k = upstream[0] * w[blockIdx.x] / (float)HW;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float p = pred[base + i], g = gt[base + i];
        float d;
        if (kind == 0) d = (p - g) / fmaxf((1.f - p) * p, 1e-12f);
        else d = 2.f * (p - g);
        dpred[base + i] = k * d;
    }
}

static unsigned hm_grid(long long items) {
    long long b = (items + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace eve

using namespace eve;

extern "C" int eve_heatmap_head_fwd(int dtype, long long pixels, int Cpad, const void* logits, float* out,
                                    eve_stream_t stream) {
    if (((unsigned)dtype > (unsigned)EVE_DT_F16) || pixels <= 0 || Cpad <= 0 || !logits || !out)
        return set_error_msg("heatmap_head_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(heatmap_head_fwd_kernel<bf16_t>, dim3(hm_grid(pixels)), dim3(256), 0, s, (const bf16_t*)logits, out, Cpad, pixels);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(heatmap_head_fwd_kernel<f16_t>, dim3(hm_grid(pixels)), dim3(256), 0, s, (const f16_t*)logits, out, Cpad, pixels);
    else                      hipLaunchKernelGGL(heatmap_head_fwd_kernel<float>, dim3(hm_grid(pixels)), dim3(256), 0, s, (const float*)logits, out, Cpad, pixels);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_heatmap_head_bwd(int dtype, long long pixels, int Cpad, const float* dy, const float* y, void* dlogits,
                                    eve_stream_t stream) {
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    if (((unsigned)dtype > (unsigned)EVE_DT_F16) || pixels <= 0 || Cpad <= 0 || Cpad % vec || !dy || !y || !dlogits)
        return set_error_msg("heatmap_head_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int cvecs = Cpad / vec;
    const long long nvec = pixels * cvecs;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(heatmap_head_bwd_kernel<bf16_t>, dim3(hm_grid(nvec)), dim3(256), 0, s, dy, y, (bf16_t*)dlogits, cvecs, nvec);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(heatmap_head_bwd_kernel<f16_t>, dim3(hm_grid(nvec)), dim3(256), 0, s, dy, y, (f16_t*)dlogits, cvecs, nvec);
    else                      hipLaunchKernelGGL(heatmap_head_bwd_kernel<float>, dim3(hm_grid(nvec)), dim3(256), 0, s, dy, y, (float*)dlogits, cvecs, nvec);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_heatmap_loss_fwd(int kind, int B, int T, int HW, const float* pred, const float* gt,
                                    const uint8_t* validity, float* per_map, float* loss, float* w, eve_stream_t stream) {
    if ((kind != 0 && kind != 1) || B <= 0 || T <= 0 || HW <= 0 || !pred || !gt || !validity || !per_map || !loss || !w)
        return set_error_msg("heatmap_loss_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(heatmap_loss_map_kernel, dim3(B * T), dim3(256), 0, s, kind, HW, pred, gt, per_map);
    hipLaunchKernelGGL(masked_clip_mean_kernel, dim3(1), dim3(256), 0, s, B, T, (const float*)per_map, validity, loss, w);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_heatmap_loss_bwd(int kind, int BT, int HW, const float* pred, const float* gt, const float* w,
                                    const float* upstream, float* dpred, eve_stream_t stream) {
    if ((kind != 0 && kind != 1) || BT <= 0 || HW <= 0 || !pred || !gt || !w || !upstream || !dpred)
        return set_error_msg("heatmap_loss_bwd: bad arguments");
    hipLaunchKernelGGL(heatmap_loss_bwd_kernel, dim3(BT), dim3(256), 0, (hipStream_t)stream, kind, HW, pred, gt, w, upstream, dpred);
    EVE_CHECK_LAUNCH();
    return 0;
}
