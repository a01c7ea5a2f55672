// InstanceNorm statistics / apply / backward and small element-wise kernels, NHWC, gfx950.
// All of these are HBM-bound streaming kernels: 16-byte vector accesses, one workgroup per image
// plane-set (n) for the reductions so the second pass over the plane hits L2.
#include "common.h"

namespace eve {

// ---- statistics: two passes (mean, then centred second moment), one workgroup per image ----
template <typename T>
__global__ __launch_bounds__(256) void in_stats_kernel(const T* __restrict__ x, float* __restrict__ mr,
                                                       int HW, int C, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh[256 * VEC];
    __shared__ float sh_mean[1024];
    const int cvecs = C / VEC, phases = 256 / cvecs;
    const int tid = threadIdx.x, cv = tid % cvecs, ph = tid / cvecs;
    const bool act = ph < phases;
    const T* xp = x + (size_t)blockIdx.x * HW * C + cv * VEC;
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    if (act)
        for (int px = ph; px < HW; px += phases) {
            float f[VEC];
            Elem<T>::unpack(*reinterpret_cast<const uint4*>(xp + (size_t)px * C), f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e] += f[e];
        }
    if (!act) { for (int e = 0; e < VEC; ++e) s[e] = 0.f; }
    // inactive threads (ph >= phases) still take part in the barriers with a dummy slot
    const int phc = act ? ph : 0, cvc = act ? cv : 0;
    if (act) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) sh[(phc * cvecs + cvc) * VEC + e] = s[e];
    }
    __syncthreads();
    if (act && ph == 0) {
        for (int q = 1; q < phases; ++q)
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e] += sh[(q * cvecs + cv) * VEC + e];
#pragma unroll
        for (int e = 0; e < VEC; ++e) sh_mean[cv * VEC + e] = s[e] / (float)HW;
    }
    __syncthreads();
    float mean[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mean[e] = sh_mean[cvc * VEC + e]; s[e] = 0.f; }
    if (act)
        for (int px = ph; px < HW; px += phases) {
            float f[VEC];
            Elem<T>::unpack(*reinterpret_cast<const uint4*>(xp + (size_t)px * C), f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = f[e] - mean[e]; s[e] += d * d; }
        }
    if (act) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) sh[(phc * cvecs + cvc) * VEC + e] = s[e];
    }
    __syncthreads();
    if (act && ph == 0) {
        for (int q = 1; q < phases; ++q)
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e] += sh[(q * cvecs + cv) * VEC + e];
        float* o = mr + ((size_t)blockIdx.x * C + cv * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            o[2 * e] = mean[e];
            o[2 * e + 1] = rsqrtf(s[e] / (float)HW + eps);
        }
    }
}

// ---- statistics in ONE pass (bf16 planes too large for L2 to serve a second one: RefineNet's 72x128 level is 295 KB
// per image, 283 MB per tensor): sums of d = x - k and d^2 with the shift k = the plane's first pixel (per channel),
// so that var = E[d^2] - E[d]^2 loses nothing to cancellation unless that pixel is many sigmas off the mean ----
template <typename T>
__global__ __launch_bounds__(256) void in_stats1_kernel(const T* __restrict__ x, float* __restrict__ mr, int HW, int C, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh[2 * 256 * VEC];
    const int cvecs = C / VEC, phases = 256 / cvecs;
    const int tid = threadIdx.x, cv = tid % cvecs, ph = tid / cvecs;
    const bool on = ph < phases;
    const T* xp = x + (size_t)blockIdx.x * HW * C + (on ? cv : 0) * VEC;
    float k[VEC], s1[VEC], s2[VEC];
    // shift = mean of `phases` pixels spread evenly over the plane (a single pixel -- the corner, say, with two thirds of
    // its receptive field in the zero padding -- can sit many sigmas off the mean of a nearly constant plane)
    {
        const int px = (int)(((long long)(on ? ph : 0) * HW) / phases);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(xp + (size_t)px * C), k);
#pragma unroll
        for (int e = 0; e < VEC; ++e) sh[tid * VEC + e] = on ? k[e] : 0.f;
        __syncthreads();
        for (int n = phases; n > 1;) {
            const int half = (n + 1) >> 1;
            if (on && ph + half < n) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) sh[tid * VEC + e] += sh[(tid + half * cvecs) * VEC + e];
            }
            __syncthreads();
            n = half;
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) k[e] = sh[(on ? cv : 0) * VEC + e] * (1.f / (float)phases);
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (on) {
#pragma unroll 4
        for (int px = ph; px < HW; px += phases) {
            float f[VEC];
            Elem<T>::unpack(*reinterpret_cast<const uint4*>(xp + (size_t)px * C), f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = f[e] - k[e]; s1[e] += d; s2[e] = fmaf(d, d, s2[e]); }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sh[tid * VEC + e] = s1[e]; sh[(256 + tid) * VEC + e] = s2[e]; }
    __syncthreads();
    for (int n = phases; n > 1;) {                         // pairwise tree over the phases
        const int half = (n + 1) >> 1;
        if (on && ph + half < n) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                sh[tid * VEC + e] += sh[(tid + half * cvecs) * VEC + e];
                sh[(256 + tid) * VEC + e] += sh[(256 + tid + half * cvecs) * VEC + e];
            }
        }
        __syncthreads();
        n = half;
    }
    if (tid < cvecs) {
        float* o = mr + ((size_t)blockIdx.x * C + tid * VEC) * 2;
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float m = sh[tid * VEC + e] * inv;
            const float var = fmaxf(sh[(256 + tid) * VEC + e] * inv - m * m, 0.f);
            o[2 * e] = k[e] + m;
            o[2 * e + 1] = rsqrtf(var + eps);
        }
    }
}

// ---- y = act(gamma*(x-mean)*rstd + beta + res) ----
// grid = (chunks, planes): a thread walks ONE plane with a stride that is a multiple of the channel-vector count, so
// its channels -- and therefore scale / shift -- are loop invariants (the first version decoded plane and channel
// with 64-bit divisions per vector and switched on the activation per element: 1.1 TB/s on 16-channel planes).
template <typename T, int ACT>
__global__ __launch_bounds__(256) void in_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ mr,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const T* __restrict__ res, const int act_rt,
                                                         T* __restrict__ y, const int HW, const int C) {
    constexpr int VEC = Elem<T>::VEC;
    const int act = ACT >= 0 ? ACT : act_rt;
    const int cvecs = C / VEC, per_img = HW * cvecs;
    const int n = blockIdx.y;
    const int stride = gridDim.x * 256;                       // a multiple of cvecs (launcher)
    const int i0 = blockIdx.x * 256 + threadIdx.x;
    const int cv = i0 % cvecs;
    float a[VEC], b[VEC];
    {
        const float* m = mr + ((size_t)n * C + cv * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int c = cv * VEC + e;
            a[e] = m[2 * e + 1]; b[e] = -m[2 * e] * a[e];
            if (gamma) { a[e] *= gamma[c]; b[e] = b[e] * gamma[c] + beta[c]; }
        }
    }
    const size_t base = (size_t)n * per_img;
#pragma unroll 4
    for (int i = i0; i < per_img; i += stride) {
        float f[VEC], r[VEC];
        Elem<T>::unpack(reinterpret_cast<const uint4*>(x)[base + i], f);
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

// ---- backward: one workgroup per image; pass 1 reduces (sum g, sum g*xhat), pass 2 writes dx ----
template <typename T, int ACT>
__global__ __launch_bounds__(256) void in_act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                         const T* __restrict__ x, const float* __restrict__ mr,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const int act_rt,
                                                         T* __restrict__ dx, T* __restrict__ dres,
                                                         float* __restrict__ sums, const T* __restrict__ dx_add, int HW, int C) {
    constexpr int VEC = Elem<T>::VEC;
    const int act = ACT >= 0 ? ACT : act_rt;          // compile-time activation: no per-element switch
    __shared__ float sh[2 * 256 * VEC];
    __shared__ float sh_tot[2 * 1024];
    const int cvecs = C / VEC, phases = 256 / cvecs;
    const int tid = threadIdx.x, cv = tid % cvecs, ph = tid / cvecs;
    const bool on = ph < phases;
    const int cvc = on ? cv : 0;
    const size_t base = (size_t)blockIdx.x * HW * C + cvc * VEC;
    float mean[VEC], rstd[VEC];
    {
        const float* m = mr + ((size_t)blockIdx.x * C + cvc * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { mean[e] = m[2 * e]; rstd[e] = m[2 * e + 1]; }
    }
    // without a residual the forward output is act(x * za + zb) with exactly these za / zb (in_act_fwd_kernel), so
    // act'(.) is recomputed from x instead of reading y: one tensor less in both passes
    float za[VEC], zb[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        za[e] = rstd[e]; zb[e] = -mean[e] * za[e];
        if (gamma && beta) { const float gm = gamma[cvc * VEC + e]; za[e] *= gm; zb[e] = zb[e] * gm + beta[cvc * VEC + e]; }   // (used only when y is omitted)
    }
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (on)
#pragma unroll 2
        for (int px = ph; px < HW; px += phases) {
            const size_t o = base + (size_t)px * C;
            float g[VEC], yy[VEC], xx[VEC];
            Elem<T>::unpack(*reinterpret_cast<const uint4*>(dy + o), g);
            Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + o), xx);
            if (act != EVE_ACT_NONE) {
                if (y) {
                    Elem<T>::unpack(*reinterpret_cast<const uint4*>(y + o), yy);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) yy[e] = act_fwd(xx[e] * za[e] + zb[e], act);
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[e] *= act_grad_from_out(yy[e], act);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                s1[e] += g[e];
                s2[e] += g[e] * (xx[e] - mean[e]) * rstd[e];
            }
        }
    if (on) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            sh[(ph * cvecs + cv) * VEC + e] = s1[e];
            sh[256 * VEC + (ph * cvecs + cv) * VEC + e] = s2[e];
        }
    }
    __syncthreads();
    if (on && ph == 0) {
        for (int q = 1; q < phases; ++q)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                s1[e] += sh[(q * cvecs + cv) * VEC + e];
                s2[e] += sh[256 * VEC + (q * cvecs + cv) * VEC + e];
            }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            sh_tot[cv * VEC + e] = s1[e];
            sh_tot[1024 + cv * VEC + e] = s2[e];
        }
        if (sums) {
            float* o = sums + ((size_t)blockIdx.x * C + cv * VEC) * 2;
#pragma unroll
            for (int e = 0; e < VEC; ++e) { o[2 * e] = s1[e]; o[2 * e + 1] = s2[e]; }
        }
    }
    __syncthreads();
    float k[VEC];
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        s1[e] = sh_tot[cvc * VEC + e] * inv;
        s2[e] = sh_tot[1024 + cvc * VEC + e] * inv;
        k[e] = rstd[e] * (gamma ? gamma[cvc * VEC + e] : 1.f);
    }
    if (on)
#pragma unroll 2
        for (int px = ph; px < HW; px += phases) {
            const size_t o = base + (size_t)px * C;
            float g[VEC], yy[VEC], xx[VEC];
            Elem<T>::unpack(*reinterpret_cast<const uint4*>(dy + o), g);
            Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + o), xx);
            if (act != EVE_ACT_NONE) {
                if (y) {
                    Elem<T>::unpack(*reinterpret_cast<const uint4*>(y + o), yy);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) yy[e] = act_fwd(xx[e] * za[e] + zb[e], act);
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[e] *= act_grad_from_out(yy[e], act);
            }
            if (dres) *reinterpret_cast<uint4*>(dres + o) = Elem<T>::pack(g);
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                g[e] = k[e] * (g[e] - s1[e] - (xx[e] - mean[e]) * rstd[e] * s2[e]);
            if (dx_add) {           // the other consumer's gradient of x; dx is rounded to the storage format first, as a separate add saw it
                float r[VEC], o2[VEC];
                Elem<T>::unpack(Elem<T>::pack(g), r);
                Elem<T>::unpack(*reinterpret_cast<const uint4*>(dx_add + o), o2);
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[e] = r[e] + o2[e];
            }
            *reinterpret_cast<uint4*>(dx + o) = Elem<T>::pack(g);
        }
}

// ---- two heads over one normalised input (RefineNet's pre-activation blocks, /root/reference/src/models/refine_net.py:46-47,
// 59-60: `layers` and `skip_layer` both start with InstanceNorm(affine) -> activation of the SAME block input) ----
// Forward: x is read once, both heads are written; each head may land in a channel range of a wider tensor (pixel stride
// ldy elements, the caller offsets the pointers): the decoder's torch.cat([upsampled, encoder]) (refine_net.py:125-126) is
// normalised source by source -- statistics are per channel -- straight into the concatenated layout and never exists
// un-normalised.  grid = (chunks, planes) as in_act_fwd_kernel.
// A launch covers ONE source (x2 == nullptr) or BOTH sources of the concatenation: source 1 owns channels [0, C) of the wide
// tensors, source 2 [C, C + C2).  The two sources of an image write (forward) / read (backward) the two parts of the same
// 128-byte lines: their workgroups are placed 8 planes apart in dispatch order -- same XCD, same L2 -- so a line is
// filled / fetched once (as separate launches every line of the wide gradient crossed HBM twice: 1.12 ms per 72 x 128
// decoder launch against 0.72 for its algorithmic bytes).
struct In2Src { uint32_t n, src; };
__device__ __forceinline__ In2Src in2_plane(uint32_t b, bool two, int N) {
    if (!two) return In2Src{b, 0u};
    const uint32_t grp = b >> 4, r = b & 15u;
    return In2Src{grp * 8u + (r & 7u), r >> 3};
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void in_act2_fwd_kernel(const T* __restrict__ x, const float* __restrict__ mr,
                                                          const float* __restrict__ gamma_a, const float* __restrict__ beta_a,
                                                          const float* __restrict__ gamma_b, const float* __restrict__ beta_b,
                                                          const int act_rt, T* __restrict__ y_a, T* __restrict__ y_b,
                                                          const int N, const int HW, const int C, const int ldy,
                                                          const T* __restrict__ x2, const float* __restrict__ mr2, const int C2) {
    constexpr int VEC = Elem<T>::VEC;
    const int act = ACT >= 0 ? ACT : act_rt;
    const In2Src ps = in2_plane(blockIdx.y, x2 != nullptr, N);
    if (ps.n >= (uint32_t)N) return;
    const int n = (int)ps.n;
    const int Cs = ps.src ? C2 : C, coff = ps.src ? C : 0;
    const T* xs = ps.src ? x2 : x;
    const float* mrs = ps.src ? mr2 : mr;
    const int cvecs = Cs / VEC, per_img = HW * cvecs;
    const int stride = gridDim.x * 256;                       // a multiple of both sources' cvecs (launcher)
    const int i0 = blockIdx.x * 256 + threadIdx.x;
    const int cv = i0 % cvecs;
    float aa[VEC], ba[VEC], ab[VEC], bb[VEC];
    {
        const float* m = mrs + ((size_t)n * Cs + cv * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int c = coff + cv * VEC + e;
            const float rstd = m[2 * e + 1], sh = -m[2 * e] * rstd;
            aa[e] = rstd * gamma_a[c]; ba[e] = sh * gamma_a[c] + beta_a[c];
            ab[e] = y_b ? rstd * gamma_b[c] : 0.f; bb[e] = y_b ? sh * gamma_b[c] + beta_b[c] : 0.f;
        }
    }
    const size_t xbase = (size_t)n * per_img;
    const size_t ybase = (size_t)n * HW * ldy + (size_t)(coff + cv * VEC);
    const int pstep = stride / cvecs;
    int px = i0 / cvecs;
#pragma unroll 4
    for (int i = i0; i < per_img; i += stride, px += pstep) {
        float f[VEC], o[VEC];
        Elem<T>::unpack(reinterpret_cast<const uint4*>(xs)[xbase + i], f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = act_fwd(f[e] * aa[e] + ba[e], act);
        *reinterpret_cast<uint4*>(y_a + ybase + (size_t)px * ldy) = Elem<T>::pack(o);
        if (y_b) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = act_fwd(f[e] * ab[e] + bb[e], act);
            *reinterpret_cast<uint4*>(y_b + ybase + (size_t)px * ldy) = Elem<T>::pack(o);
        }
    }
}

// Backward of the two heads: g_h = dy_h * act'(head h's output, recomputed from x), dx = sum over heads of
// k_h * (g_h - mean g_h - xhat * mean(g_h xhat)): the fork's gradient sum is formed in registers (the single-head kernel
// twice + an add kernel moved 13 tensor passes, this one 7).  dy_a / dy_b have pixel stride lddy (channel ranges of the
// wider gradient).  One workgroup per image, two passes like in_act_bwd_kernel.
template <typename T, int ACT>
__global__ __launch_bounds__(256) void in_act2_bwd_kernel(const T* __restrict__ dy_a, const T* __restrict__ dy_b, const int lddy,
                                                          const T* __restrict__ x1, const float* __restrict__ mr1,
                                                          const float* __restrict__ gamma_a, const float* __restrict__ beta_a,
                                                          const float* __restrict__ gamma_b, const float* __restrict__ beta_b,
                                                          const int act_rt, T* __restrict__ dx1,
                                                          float* __restrict__ sums_a, float* __restrict__ sums_b, int N, int HW, int C1,
                                                          const T* __restrict__ x2, const float* __restrict__ mr2, T* __restrict__ dx2,
                                                          int C2) {
    constexpr int VEC = Elem<T>::VEC;
    const int act = ACT >= 0 ? ACT : act_rt;
    __shared__ float sh[4 * 256 * VEC];
    __shared__ float sh_tot[4 * 1024];
    const In2Src ps = in2_plane(blockIdx.x, x2 != nullptr, N);
    if (ps.n >= (uint32_t)N) return;
    const int C = ps.src ? C2 : C1, coff = ps.src ? C1 : 0, ctot = C1 + C2;
    const T* x = ps.src ? x2 : x1;
    const float* mr = ps.src ? mr2 : mr1;
    T* dx = ps.src ? dx2 : dx1;
    const int cvecs = C / VEC, phases = 256 / cvecs;
    const int tid = threadIdx.x, cv = tid % cvecs, ph = tid / cvecs;
    const bool on = ph < phases;
    const int cvc = on ? cv : 0;
    const bool two = dy_b != nullptr;
    const size_t xbase = (size_t)ps.n * HW * C + cvc * VEC;
    const size_t gbase = (size_t)ps.n * HW * lddy + coff + cvc * VEC;
    float mean[VEC], rstd[VEC], za[VEC], zb[VEC], wa[VEC], wb[VEC];     // head a: x * za + zb, head b: x * wa + wb
    {
        const float* m = mr + ((size_t)ps.n * C + cvc * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int c = coff + cvc * VEC + e;
            mean[e] = m[2 * e]; rstd[e] = m[2 * e + 1];
            za[e] = rstd[e] * gamma_a[c]; zb[e] = -mean[e] * za[e] + beta_a[c];
            wa[e] = two ? rstd[e] * gamma_b[c] : 0.f; wb[e] = two ? -mean[e] * wa[e] + beta_b[c] : 0.f;
        }
    }
    auto grads = [&](int px, float (&ga)[VEC], float (&gb)[VEC], float (&xh)[VEC]) {
        float xx[VEC];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + xbase + (size_t)px * C), xx);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(dy_a + gbase + (size_t)px * lddy), ga);
        if (two) Elem<T>::unpack(*reinterpret_cast<const uint4*>(dy_b + gbase + (size_t)px * lddy), gb);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            xh[e] = (xx[e] - mean[e]) * rstd[e];
            if (act != EVE_ACT_NONE) {
                ga[e] *= act_grad_from_out(act_fwd(xx[e] * za[e] + zb[e], act), act);
                gb[e] = two ? gb[e] * act_grad_from_out(act_fwd(xx[e] * wa[e] + wb[e], act), act) : 0.f;
            } else if (!two) gb[e] = 0.f;
        }
    };
    float acc[4][VEC];                  // sum g_a | sum g_a xhat | sum g_b | sum g_b xhat
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[q][e] = 0.f;
    if (on)
#pragma unroll 2
        for (int px = ph; px < HW; px += phases) {
            float ga[VEC], gb[VEC], xh[VEC];
            grads(px, ga, gb, xh);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                acc[0][e] += ga[e]; acc[1][e] += ga[e] * xh[e];
                acc[2][e] += gb[e]; acc[3][e] += gb[e] * xh[e];
            }
        }
    if (on) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < VEC; ++e) sh[q * 256 * VEC + (ph * cvecs + cv) * VEC + e] = acc[q][e];
    }
    __syncthreads();
    if (on && ph == 0) {
        for (int r = 1; r < phases; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[q][e] += sh[q * 256 * VEC + (r * cvecs + cv) * VEC + e];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < VEC; ++e) sh_tot[q * 1024 + cv * VEC + e] = acc[q][e];
        float* oa = sums_a + ((size_t)ps.n * ctot + coff + cv * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { oa[2 * e] = acc[0][e]; oa[2 * e + 1] = acc[1][e]; }
        if (two) {
            float* ob = sums_b + ((size_t)ps.n * ctot + coff + cv * VEC) * 2;
#pragma unroll
            for (int e = 0; e < VEC; ++e) { ob[2 * e] = acc[2][e]; ob[2 * e + 1] = acc[3][e]; }
        }
    }
    __syncthreads();
    const float inv = 1.f / (float)HW;
    float ka[VEC], kb[VEC];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[q][e] = sh_tot[q * 1024 + cvc * VEC + e] * inv;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int c = coff + cvc * VEC + e;
        ka[e] = rstd[e] * gamma_a[c];
        kb[e] = two ? rstd[e] * gamma_b[c] : 0.f;
    }
    if (on)
#pragma unroll 2
        for (int px = ph; px < HW; px += phases) {
            float ga[VEC], gb[VEC], xh[VEC];
            grads(px, ga, gb, xh);
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                ga[e] = ka[e] * (ga[e] - acc[0][e] - xh[e] * acc[1][e]) + kb[e] * (gb[e] - acc[2][e] - xh[e] * acc[3][e]);
            *reinterpret_cast<uint4*>(dx + xbase + (size_t)px * C) = Elem<T>::pack(ga);
        }
}

template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, int act,
                                                      T* __restrict__ dx, long long n) {
    constexpr int VEC = Elem<T>::VEC;
    const long long nvec = n / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        float g[VEC], yy[VEC];
        Elem<T>::unpack(reinterpret_cast<const uint4*>(dy)[i], g);
        Elem<T>::unpack(reinterpret_cast<const uint4*>(y)[i], yy);
#pragma unroll
        for (int e = 0; e < VEC; ++e) g[e] *= act_grad_from_out(yy[e], act);
        reinterpret_cast<uint4*>(dx)[i] = Elem<T>::pack(g);
    }
    // scalar tail
    for (long long i = nvec * VEC + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        Elem<T>::st(dx + i, Elem<T>::ld(dy + i) * act_grad_from_out(Elem<T>::ld(y + i), act));
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                  T* __restrict__ o, long long n) {
    constexpr int VEC = Elem<T>::VEC;
    const long long nvec = n / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        float fa[VEC], fb[VEC];
        Elem<T>::unpack(reinterpret_cast<const uint4*>(a)[i], fa);
        Elem<T>::unpack(reinterpret_cast<const uint4*>(b)[i], fb);
#pragma unroll
        for (int e = 0; e < VEC; ++e) fa[e] += fb[e];
        reinterpret_cast<uint4*>(o)[i] = Elem<T>::pack(fa);
    }
    for (long long i = nvec * VEC + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        Elem<T>::st(o + i, Elem<T>::ld(a + i) + Elem<T>::ld(b + i));
}

// out[j] = sum over rows of in[row][j] (float32): the N-reduction of the per-plane affine-gradient partials
// (sums[n][c][2] -> d(beta), d(gamma): refine_net.py's InstanceNorm2d(affine=True) layers).  Fixed order: 64 columns x 4 row
// phases per workgroup, each phase sums its rows in sequence, the phases are combined in order.
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float sh[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    float s = 0.f;
    if (c < cols) {
#pragma unroll 8
        for (int r = ph; r < rows; r += 4) s += in[(size_t)r * cols + c];
    }
    sh[ph][threadIdx.x & 63] = s;
    __syncthreads();
    if (ph == 0 && c < cols) out[c] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// the same reduction for the (d beta, d gamma) PAIRS of an affine InstanceNorm, ADDED straight into the two gradient vectors
// (the trainer's flat gradient buffer: the autograd route returned two strided views and paid one accumulate launch each)
// (round 5: 16 row phases per 64 columns instead of 4 -- with N = 960 planes a thread walked 240 rows one dependent load
//  after the other, 16.5 us per launch and 39 launches per configs[2] step)
__global__ __launch_bounds__(1024) void sum_rows_pairs_kernel(const float* __restrict__ in, float* __restrict__ out0,
                                                              float* __restrict__ out1, int rows, int C) {
    __shared__ float sh[16][64];
    const int cols = 2 * C;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    float s = 0.f;
    if (c < cols) {
#pragma unroll 8
        for (int r = ph; r < rows; r += 16) s += in[(size_t)r * cols + c];
    }
    sh[ph][threadIdx.x & 63] = s;
    __syncthreads();
    if (ph == 0 && c < cols) {
        float v = sh[0][threadIdx.x];
#pragma unroll
        for (int q = 1; q < 16; ++q) v += sh[q][threadIdx.x];          // fixed order
        float* const dst = (c & 1) ? out1 + (c >> 1) : out0 + (c >> 1);
        *dst += v;
    }
}

// the activations the two networks use get their own instantiation (no per-element switch), the rest the run-time one
#define EVE_IN_ACT_DISPATCH(KERNEL, T, TS, GRID, ...)                                                                                   \
    do { switch (act) {                                                                                                                 \
        case EVE_ACT_NONE:  EVE_LAUNCH(#KERNEL "<" TS ", 0>", (KERNEL<T, EVE_ACT_NONE>), GRID, dim3(256), 0, s, __VA_ARGS__); break;      \
        case EVE_ACT_RELU:  EVE_LAUNCH(#KERNEL "<" TS ", 1>", (KERNEL<T, EVE_ACT_RELU>), GRID, dim3(256), 0, s, __VA_ARGS__); break;      \
        case EVE_ACT_LEAKY: EVE_LAUNCH(#KERNEL "<" TS ", 2>", (KERNEL<T, EVE_ACT_LEAKY>), GRID, dim3(256), 0, s, __VA_ARGS__); break;     \
        default:            EVE_LAUNCH(#KERNEL "<" TS ", -1>", (KERNEL<T, -1>), GRID, dim3(256), 0, s, __VA_ARGS__); break;               \
    } } while (0)

static inline unsigned stream_grid(long long nvec) {
    long long b = (nvec + 255) / 256;
    if (b > 256 * 8) b = 256 * 8;     // 8 workgroups per CU, grid-stride the rest
    if (b < 1) b = 1;
    return (unsigned)b;
}

static int check_plane(int dtype, int N, int HW, int C, const char* who) {
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    if ((unsigned)dtype > (unsigned)EVE_DT_F16) return set_error_msg("instnorm: bad dtype");
    if (N <= 0 || HW <= 0 || C <= 0 || C % vec || C / vec > 256 || C > 1024) return set_error_msg(who);
    return 0;
}

}  // namespace eve

using namespace eve;

extern "C" int eve_instnorm_stats(int dtype, int N, int HW, int C, const void* x, float eps, float* mean_rstd,
                                  eve_stream_t stream) {
    if (int e = check_plane(dtype, N, HW, C, "instnorm_stats: bad shape")) return e;
    if (!x || !mean_rstd) return set_error_msg("instnorm_stats: null pointer");
    hipStream_t s = (hipStream_t)stream;
    // bf16 planes beyond ~64 KB per image: the second pass of the two-pass kernel would come from HBM again
    const int one_pass = g_cfg.in_stats_one_pass;
    if (dtype == EVE_DT_BF16 && one_pass && (long long)HW * C * 2 >= 65536)
        EVE_LAUNCH("in_stats1_kernel<eve::bf16_t>", in_stats1_kernel<bf16_t>, dim3(N), dim3(256), 0, s, (const bf16_t*)x, mean_rstd, HW, C, eps);
    else if (dtype == EVE_DT_F16 && one_pass && (long long)HW * C * 2 >= 65536)
        EVE_LAUNCH("in_stats1_kernel<eve::f16_t>", in_stats1_kernel<f16_t>, dim3(N), dim3(256), 0, s, (const f16_t*)x, mean_rstd, HW, C, eps);
    else if (dtype == EVE_DT_BF16) EVE_LAUNCH("in_stats_kernel<eve::bf16_t>", in_stats_kernel<bf16_t>, dim3(N), dim3(256), 0, s, (const bf16_t*)x, mean_rstd, HW, C, eps);
    else if (dtype == EVE_DT_F16) EVE_LAUNCH("in_stats_kernel<eve::f16_t>", in_stats_kernel<f16_t>, dim3(N), dim3(256), 0, s, (const f16_t*)x, mean_rstd, HW, C, eps);
    else                      EVE_LAUNCH("in_stats_kernel<float>", in_stats_kernel<float>, dim3(N), dim3(256), 0, s, (const float*)x, mean_rstd, HW, C, eps);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_instnorm_act_fwd(int dtype, int N, int HW, int C, const void* x, const float* mean_rstd,
                                    const float* gamma, const float* beta, const void* res, int act, void* y,
                                    eve_stream_t stream) {
    if (int e = check_plane(dtype, N, HW, C, "instnorm_act_fwd: bad shape")) return e;
    if (!x || !mean_rstd || !y || ((gamma == nullptr) != (beta == nullptr)))
        return set_error_msg("instnorm_act_fwd: null pointer / gamma-beta mismatch");
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    const int cvecs = C / vec;
    const long long per_img = (long long)HW * cvecs;
    if (per_img >= (1ll << 31) || N > 65535) return set_error_msg("instnorm_act_fwd: plane / batch too large");
    // chunks of a plane per workgroup column: ~8 vectors per thread, and chunks * 256 a multiple of cvecs
    int g = cvecs, h256 = 256;
    while (h256) { const int t = g % h256; g = h256; h256 = t; }          // gcd(cvecs, 256)
    const int mult = cvecs / g;
    long long chunks = (per_img + 2047) / 2048;
    if (chunks > 32) chunks = 32;
    chunks = (chunks + mult - 1) / mult * mult;
    const dim3 fgrid((unsigned)chunks, (unsigned)N);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16)
        EVE_IN_ACT_DISPATCH(in_act_fwd_kernel, bf16_t, "eve::bf16_t", fgrid, (const bf16_t*)x, mean_rstd, gamma, beta, (const bf16_t*)res, act, (bf16_t*)y, HW, C);
    else if (dtype == EVE_DT_F16)
        EVE_IN_ACT_DISPATCH(in_act_fwd_kernel, f16_t, "eve::f16_t", fgrid, (const f16_t*)x, mean_rstd, gamma, beta, (const f16_t*)res, act, (f16_t*)y, HW, C);
    else
        EVE_IN_ACT_DISPATCH(in_act_fwd_kernel, float, "float", fgrid, (const float*)x, mean_rstd, gamma, beta, (const float*)res, act, (float*)y, HW, C);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_instnorm_act_bwd(int dtype, int N, int HW, int C, const void* dy, const void* y, const void* x,
                                    const float* mean_rstd, const float* gamma, const float* beta, int act, void* dx, void* dres,
                                    float* sums, const void* dx_add, eve_stream_t stream) {
    if (int e = check_plane(dtype, N, HW, C, "instnorm_act_bwd: bad shape")) return e;
    if (!dy || !x || !mean_rstd || !dx || (act != EVE_ACT_NONE && !y && gamma && !beta) || (dres && act != EVE_ACT_NONE && !y))
        return set_error_msg("instnorm_act_bwd: null pointer (y is required with a residual, beta with gamma when y is omitted)");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16)
        EVE_IN_ACT_DISPATCH(in_act_bwd_kernel, bf16_t, "eve::bf16_t", dim3(N), (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, mean_rstd, gamma, beta, act, (bf16_t*)dx, (bf16_t*)dres, sums, (const bf16_t*)dx_add, HW, C);
    else if (dtype == EVE_DT_F16)
        EVE_IN_ACT_DISPATCH(in_act_bwd_kernel, f16_t, "eve::f16_t", dim3(N), (const f16_t*)dy, (const f16_t*)y, (const f16_t*)x, mean_rstd, gamma, beta, act, (f16_t*)dx, (f16_t*)dres, sums, (const f16_t*)dx_add, HW, C);
    else
        EVE_IN_ACT_DISPATCH(in_act_bwd_kernel, float, "float", dim3(N), (const float*)dy, (const float*)y, (const float*)x, mean_rstd, gamma, beta, act, (float*)dx, (float*)dres, sums, (const float*)dx_add, HW, C);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_instnorm_act2_fwd(int dtype, int N, int HW, int C, const void* x, const float* mean_rstd,
                                     const float* gamma_a, const float* beta_a, const float* gamma_b, const float* beta_b,
                                     int act, void* y_a, void* y_b, int ldy, int C2, const void* x2,
                                     const float* mean_rstd2, eve_stream_t stream) {
    if (int e = check_plane(dtype, N, HW, C, "instnorm_act2_fwd: bad shape")) return e;
    if (x2) { if (int e = check_plane(dtype, N, HW, C2, "instnorm_act2_fwd: bad second source")) return e; }
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    if (!x2) C2 = 0;
    if (!x || !mean_rstd || !y_a || !gamma_a || !beta_a || (y_b && (!gamma_b || !beta_b)) || ldy < C + C2 || ldy % vec ||
        ((uintptr_t)y_a & 15) || ((uintptr_t)y_b & 15) || (x2 && !mean_rstd2))
        return set_error_msg("instnorm_act2_fwd: null pointer / misaligned head / ldy < C + C2");
    const int cv1 = C / vec, cv2 = x2 ? C2 / vec : cv1;
    const long long per_img = (long long)HW * (cv1 > cv2 ? cv1 : cv2);
    if ((long long)HW * ldy >= (1ll << 31) || N > 30000) return set_error_msg("instnorm_act2_fwd: plane / batch too large");
    auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
    const int l = cv1 / gcd(cv1, cv2) * cv2;                  // chunks * 256 must be a multiple of both vector counts
    const int mult = l / gcd(l, 256);
    long long chunks = (per_img + 2047) / 2048;
    if (chunks > 32) chunks = 32;
    chunks = (chunks + mult - 1) / mult * mult;
    const unsigned planes = x2 ? (unsigned)((N + 7) / 8) * 16u : (unsigned)N;
    const dim3 fgrid((unsigned)chunks, planes);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16)
        EVE_IN_ACT_DISPATCH(in_act2_fwd_kernel, bf16_t, "eve::bf16_t", fgrid, (const bf16_t*)x, mean_rstd, gamma_a, beta_a, gamma_b, beta_b, act, (bf16_t*)y_a, (bf16_t*)y_b, N, HW, C, ldy, (const bf16_t*)x2, mean_rstd2, C2);
    else if (dtype == EVE_DT_F16)
        EVE_IN_ACT_DISPATCH(in_act2_fwd_kernel, f16_t, "eve::f16_t", fgrid, (const f16_t*)x, mean_rstd, gamma_a, beta_a, gamma_b, beta_b, act, (f16_t*)y_a, (f16_t*)y_b, N, HW, C, ldy, (const f16_t*)x2, mean_rstd2, C2);
    else
        EVE_IN_ACT_DISPATCH(in_act2_fwd_kernel, float, "float", fgrid, (const float*)x, mean_rstd, gamma_a, beta_a, gamma_b, beta_b, act, (float*)y_a, (float*)y_b, N, HW, C, ldy, (const float*)x2, mean_rstd2, C2);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_instnorm_act2_bwd(int dtype, int N, int HW, int C, const void* dy_a, const void* dy_b, int lddy,
                                     const void* x, const float* mean_rstd, const float* gamma_a, const float* beta_a,
                                     const float* gamma_b, const float* beta_b, int act, void* dx, float* sums_a,
                                     float* sums_b, int C2, const void* x2, const float* mean_rstd2, void* dx2,
                                     eve_stream_t stream) {
    if (int e = check_plane(dtype, N, HW, C, "instnorm_act2_bwd: bad shape")) return e;
    if (x2) { if (int e = check_plane(dtype, N, HW, C2, "instnorm_act2_bwd: bad second source")) return e; }
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    if (!x2) C2 = 0;
    if (!dy_a || !x || !mean_rstd || !dx || !gamma_a || !beta_a || !sums_a || (dy_b && (!gamma_b || !beta_b || !sums_b)) ||
        lddy < C + C2 || lddy % vec || ((uintptr_t)dy_a & 15) || ((uintptr_t)dy_b & 15) || (x2 && (!mean_rstd2 || !dx2)))
        return set_error_msg("instnorm_act2_bwd: null pointer / misaligned head / lddy < C + C2");
    const dim3 grid(x2 ? (unsigned)((N + 7) / 8) * 16u : (unsigned)N);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16)
        EVE_IN_ACT_DISPATCH(in_act2_bwd_kernel, bf16_t, "eve::bf16_t", grid, (const bf16_t*)dy_a, (const bf16_t*)dy_b, lddy, (const bf16_t*)x, mean_rstd, gamma_a, beta_a, gamma_b, beta_b, act, (bf16_t*)dx, sums_a, sums_b, N, HW, C, (const bf16_t*)x2, mean_rstd2, (bf16_t*)dx2, C2);
    else if (dtype == EVE_DT_F16)
        EVE_IN_ACT_DISPATCH(in_act2_bwd_kernel, f16_t, "eve::f16_t", grid, (const f16_t*)dy_a, (const f16_t*)dy_b, lddy, (const f16_t*)x, mean_rstd, gamma_a, beta_a, gamma_b, beta_b, act, (f16_t*)dx, sums_a, sums_b, N, HW, C, (const f16_t*)x2, mean_rstd2, (f16_t*)dx2, C2);
    else
        EVE_IN_ACT_DISPATCH(in_act2_bwd_kernel, float, "float", grid, (const float*)dy_a, (const float*)dy_b, lddy, (const float*)x, mean_rstd, gamma_a, beta_a, gamma_b, beta_b, act, (float*)dx, sums_a, sums_b, N, HW, C, (const float*)x2, mean_rstd2, (float*)dx2, C2);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_sum_rows(int rows, int cols, const float* in, float* out, eve_stream_t stream) {
    if (rows <= 0 || cols <= 0 || !in || !out) return set_error_msg("sum_rows: bad arguments");
    hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(256), 0, (hipStream_t)stream, in, out, rows, cols);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_sum_rows_pairs(int rows, int C, const float* in, float* out0, float* out1, eve_stream_t stream) {
    if (rows <= 0 || C <= 0 || !in || !out0 || !out1) return set_error_msg("sum_rows_pairs: bad arguments");
    hipLaunchKernelGGL(sum_rows_pairs_kernel, dim3((unsigned)((2 * C + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, in, out0, out1, rows, C);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_act_bwd(int dtype, long long n, const void* dy, const void* y, int act, void* dx,
                           eve_stream_t stream) {
    if (n <= 0 || !dy || !y || !dx) return set_error_msg("act_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(stream_grid(n / 8 + 1)), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)y, act, (bf16_t*)dx, n);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(act_bwd_kernel<f16_t>, dim3(stream_grid(n / 8 + 1)), dim3(256), 0, s, (const f16_t*)dy, (const f16_t*)y, act, (f16_t*)dx, n);
    else                      hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, s, (const float*)dy, (const float*)y, act, (float*)dx, n);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_add(int dtype, long long n, const void* a, const void* b, void* out, eve_stream_t stream) {
    if (n <= 0 || !a || !b || !out) return set_error_msg("add: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(stream_grid(n / 8 + 1)), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(add_kernel<f16_t>, dim3(stream_grid(n / 8 + 1)), dim3(256), 0, s, (const f16_t*)a, (const f16_t*)b, (f16_t*)out, n);
    else                      hipLaunchKernelGGL(add_kernel<float>, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)out, n);
    EVE_CHECK_LAUNCH();
    return 0;
}
