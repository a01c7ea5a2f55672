// Recurrent pieces of the hot path, gfx950.
//   * GRU scan: nn.GRUCell over T steps (reference: src/models/eye_net.py:69,116-134).  The input-side
//     GEMM (W_ih x + b_ih) for ALL steps is done up front by the batched linear kernel; what is left is a
//     strictly sequential H x 3H mat-vec per step and sequence, so one persistent workgroup owns one
//     sequence for all T steps (sequences are independent): h lives in LDS, W_hh streams from L2.
//   * conv-GRU gate math (reference: src/models/common.py:409-414) as fused element-wise kernels around
//     the two 3x3 gate convolutions.
#include "common.h"

namespace eve {

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + __expf(-z)); }

// grid = S sequences, block = 3H threads (H <= 256).  whh_t is [H][3H] (transposed for coalescing).
__global__ void gru_scan_fwd_kernel(int T, int H, const float* __restrict__ gi, const float* __restrict__ whh_t,
                                    const float* __restrict__ bhh, const float* __restrict__ h0,
                                    float* __restrict__ hs, float* __restrict__ gates, float* __restrict__ hn_pre) {
    extern __shared__ float sm[];
    float* h = sm;            // [H]
    float* gh = sm + H;       // [3H]
    const int s = blockIdx.x, j = threadIdx.x, H3 = 3 * H;
    if (j < H) h[j] = h0 ? h0[(size_t)s * H + j] : 0.f;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        if (j < H3) {
            float a = bhh[j];
            for (int k = 0; k < H; ++k) a = fmaf(whh_t[(size_t)k * H3 + j], h[k], a);
            gh[j] = a;
        }
        __syncthreads();
        float hnew = 0.f;
        if (j < H) {
            const float* g = gi + ((size_t)s * T + t) * H3;
            const float r = sigmoidf_(g[j] + gh[j]);
            const float z = sigmoidf_(g[H + j] + gh[H + j]);
            const float n = tanhf(g[2 * H + j] + r * gh[2 * H + j]);
            hnew = (1.f - z) * n + z * h[j];
            float* go = gates + ((size_t)s * T + t) * H3;
            go[j] = r; go[H + j] = z; go[2 * H + j] = n;
            hn_pre[((size_t)s * T + t) * H + j] = gh[2 * H + j];
            hs[((size_t)s * T + t) * H + j] = hnew;
        }
        __syncthreads();
        if (j < H) h[j] = hnew;
        __syncthreads();
    }
}

// grid = S, block = 3H.  whh is the original [3H][H].
__global__ void gru_scan_bwd_kernel(int T, int H, const float* __restrict__ dhs, const float* __restrict__ whh,
                                    const float* __restrict__ h0, const float* __restrict__ hs,
                                    const float* __restrict__ gates, const float* __restrict__ hn_pre,
                                    float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ dh0) {
    extern __shared__ float sm[];
    float* dh = sm;           // [H]   carried gradient on h_t
    float* dg = sm + H;       // [3H]  dgh of the current step
    const int s = blockIdx.x, j = threadIdx.x, H3 = 3 * H;
    if (j < H) dh[j] = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        float direct = 0.f;
        if (j < H) {
            const size_t o = ((size_t)s * T + t);
            const float d = dh[j] + dhs[o * H + j];
            const float* g = gates + o * H3;
            const float r = g[j], z = g[H + j], n = g[2 * H + j];
            const float hp = t > 0 ? hs[(o - 1) * H + j] : (h0 ? h0[(size_t)s * H + j] : 0.f);
            const float dn_pre = d * (1.f - z) * (1.f - n * n);
            const float dz_pre = d * (hp - n) * z * (1.f - z);
            const float dr_pre = dn_pre * hn_pre[o * H + j] * r * (1.f - r);
            float* gi_o = dgi + o * H3;
            float* gh_o = dgh + o * H3;
            gi_o[j] = dr_pre; gi_o[H + j] = dz_pre; gi_o[2 * H + j] = dn_pre;
            gh_o[j] = dr_pre; gh_o[H + j] = dz_pre; gh_o[2 * H + j] = dn_pre * r;
            dg[j] = dr_pre; dg[H + j] = dz_pre; dg[2 * H + j] = dn_pre * r;
            direct = d * z;
        }
        __syncthreads();
        if (j < H) {
            float a = direct;
            for (int q = 0; q < H3; ++q) a = fmaf(whh[(size_t)q * H + j], dg[q], a);
            dh[j] = a;
        }
        __syncthreads();
    }
    if (dh0 && j < H) dh0[(size_t)s * H + j] = dh[j];
}

// ---------------------------------------------------------------------------------------------------
// nn.RNNCell (tanh) and nn.LSTMCell over T (eye_net.py:60-67 variants of the recurrent stage).  G gate blocks of H
// rows (1 / 4); grid = S sequences, block = G*H threads (H <= 256); whh_t is [H][G*H].
//   RNN : h' = tanh(gi + W_hh h + b_hh)
//   LSTM: (i, f, g, o) = gi + W_hh h + b_hh;  c' = s(f) c + s(i) tanh(g);  h' = s(o) tanh(c')
// ---------------------------------------------------------------------------------------------------
template <int G>
__global__ void cell_scan_fwd_kernel(int T, int H, const float* __restrict__ gi, const float* __restrict__ whh_t,
                                     const float* __restrict__ bhh, const float* __restrict__ h0, const float* __restrict__ c0,
                                     float* __restrict__ hs, float* __restrict__ cs, float* __restrict__ gates) {
    extern __shared__ float sm[];
    float* h = sm;                  // [H]
    float* pre = sm + H;            // [G*H]
    const int s = blockIdx.x, j = threadIdx.x, HG = G * H;
    float c = 0.f;
    if (j < H) {
        h[j] = h0 ? h0[(size_t)s * H + j] : 0.f;
        if (G == 4) c = c0 ? c0[(size_t)s * H + j] : 0.f;
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const size_t o = (size_t)s * T + t;
        if (j < HG) {
            float a = bhh[j] + gi[o * HG + j];
            for (int k = 0; k < H; ++k) a = fmaf(whh_t[(size_t)k * HG + j], h[k], a);
            pre[j] = a;
        }
        __syncthreads();
        float hnew = 0.f;
        if (j < H) {
            if (G == 1) {
                hnew = tanhf(pre[j]);
            } else {
                const float ig = sigmoidf_(pre[j]), fg = sigmoidf_(pre[H + j]), gg = tanhf(pre[2 * H + j]);
                const float og = sigmoidf_(pre[3 * H + j]);
                c = fg * c + ig * gg;
                hnew = og * tanhf(c);
                float* go = gates + o * HG;
                go[j] = ig; go[H + j] = fg; go[2 * H + j] = gg; go[3 * H + j] = og;
                cs[o * H + j] = c;
            }
            hs[o * H + j] = hnew;
        }
        __syncthreads();
        if (j < H) h[j] = hnew;
        __syncthreads();
    }
}

// whh is the original [G*H][H].  dhs / dcs: gradients arriving at every step's h (and c, LSTM, nullable).
template <int G>
__global__ void cell_scan_bwd_kernel(int T, int H, const float* __restrict__ dhs, const float* __restrict__ dcs,
                                     const float* __restrict__ whh, const float* __restrict__ c0, const float* __restrict__ hs,
                                     const float* __restrict__ cs, const float* __restrict__ gates, float* __restrict__ dpre_out,
                                     float* __restrict__ dh0, float* __restrict__ dc0) {
    extern __shared__ float sm[];
    float* dh = sm;                 // [H]   carried gradient on h_t
    float* dp = sm + H;             // [G*H] gradient on the pre-activations of the current step
    const int s = blockIdx.x, j = threadIdx.x, HG = G * H;
    float dc = 0.f;
    if (j < H) dh[j] = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        const size_t o = (size_t)s * T + t;
        if (j < H) {
            const float d = dh[j] + dhs[o * H + j];
            if (G == 1) {
                const float hv = hs[o * H + j];
                dp[j] = d * (1.f - hv * hv);
            } else {
                const float* g = gates + o * HG;
                const float ig = g[j], fg = g[H + j], gg = g[2 * H + j], og = g[3 * H + j];
                const float cv = cs[o * H + j], tc = tanhf(cv);
                const float cp = t > 0 ? cs[(o - 1) * H + j] : (c0 ? c0[(size_t)s * H + j] : 0.f);
                dc += (dcs ? dcs[o * H + j] : 0.f) + d * og * (1.f - tc * tc);
                dp[j] = dc * gg * ig * (1.f - ig);
                dp[H + j] = dc * cp * fg * (1.f - fg);
                dp[2 * H + j] = dc * ig * (1.f - gg * gg);
                dp[3 * H + j] = d * tc * og * (1.f - og);
                dc *= fg;
            }
        }
        __syncthreads();
        if (j < HG) dpre_out[o * HG + j] = dp[j];
        if (j < H) {
            float a = 0.f;
            for (int q = 0; q < HG; ++q) a = fmaf(whh[(size_t)q * H + j], dp[q], a);
            dh[j] = a;
        }
        __syncthreads();
    }
    if (j < H) {
        if (dh0) dh0[(size_t)s * H + j] = dh[j];
        if (G == 4 && dc0) dc0[(size_t)s * H + j] = dc;
    }
}

// ---------------------------------------------------------------------------------------------------
// H = 128 (the configured eye_net_rnn_num_features): the recurrent weights live in REGISTERS -- thread j keeps
// column j of W_hh^T (128 floats) for all T steps instead of re-reading 192 KB from L1/L2 every step -- and the
// per-step operands of the NEXT step are fetched while the current dot products run.  One workgroup per sequence.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(384) void gru_scan_fwd128_kernel(int T, const float* __restrict__ gi, const float* __restrict__ whh_t,
                                                              const float* __restrict__ bhh, const float* __restrict__ h0,
                                                              float* __restrict__ hs, float* __restrict__ gates,
                                                              float* __restrict__ hn_pre) {
    constexpr int H = 128, H3 = 384;
    __shared__ __attribute__((aligned(16))) float h[H];
    __shared__ float gh[H3];
    const int s = blockIdx.x, j = threadIdx.x;
    float w[H];
#pragma unroll
    for (int k = 0; k < H; ++k) w[k] = whh_t[(size_t)k * H3 + j];
    const float b = bhh[j];
    if (j < H) h[j] = h0 ? h0[(size_t)s * H + j] : 0.f;
    const float* g = gi + (size_t)s * T * H3;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (j < H) { g0 = g[j]; g1 = g[H + j]; g2 = g[2 * H + j]; }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        float n0 = 0.f, n1 = 0.f, n2 = 0.f;                  // the next step's input projections, in flight during the dot
        if (j < H && t + 1 < T) {
            const float* gn = g + (size_t)(t + 1) * H3;
            n0 = gn[j]; n1 = gn[H + j]; n2 = gn[2 * H + j];
        }
        float a = b;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(&h[k]);
            a = fmaf(w[k], hv.x, a); a = fmaf(w[k + 1], hv.y, a); a = fmaf(w[k + 2], hv.z, a); a = fmaf(w[k + 3], hv.w, a);
        }
        gh[j] = a;
        __syncthreads();
        float hnew = 0.f;
        if (j < H) {
            const float r = sigmoidf_(g0 + gh[j]);
            const float z = sigmoidf_(g1 + gh[H + j]);
            const float n = tanhf(g2 + r * gh[2 * H + j]);
            hnew = (1.f - z) * n + z * h[j];
            const size_t o = (size_t)s * T + t;
            float* go = gates + o * H3;
            go[j] = r; go[H + j] = z; go[2 * H + j] = n;
            hn_pre[o * H + j] = gh[2 * H + j];
            hs[o * H + j] = hnew;
        }
        __syncthreads();
        if (j < H) h[j] = hnew;
        g0 = n0; g1 = n1; g2 = n2;
        __syncthreads();
    }
}

__global__ __launch_bounds__(384) void gru_scan_bwd128_kernel(int T, const float* __restrict__ dhs, const float* __restrict__ whh,
                                                              const float* __restrict__ h0, const float* __restrict__ hs,
                                                              const float* __restrict__ gates, const float* __restrict__ hn_pre,
                                                              float* __restrict__ dgi, float* __restrict__ dgh,
                                                              float* __restrict__ dh0) {
    constexpr int H = 128, H3 = 384;
    __shared__ float dh[H];
    __shared__ __attribute__((aligned(16))) float dg[H3];
    __shared__ float part[3][H];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int j = tid & (H - 1), pq = tid >> 7;               // this thread sums rows 128*pq .. 128*pq+127 of W_hh, column j
    float w[H];
#pragma unroll
    for (int q = 0; q < H; ++q) w[q] = whh[(size_t)(pq * H + q) * H + j];
    if (tid < H) dh[tid] = 0.f;
    // operands of step t (threads < H): dhs, r, z, n, h_{t-1}, hn_pre
    float c_d = 0.f, c_r = 0.f, c_z = 0.f, c_n = 0.f, c_hp = 0.f, c_hn = 0.f;
    auto load_step = [&](int t, float& d, float& r, float& z, float& n, float& hp, float& hn) {
        const size_t o = (size_t)s * T + t;
        d = dhs[o * H + tid];
        const float* gt = gates + o * H3;
        r = gt[tid]; z = gt[H + tid]; n = gt[2 * H + tid];
        hp = t > 0 ? hs[(o - 1) * H + tid] : (h0 ? h0[(size_t)s * H + tid] : 0.f);
        hn = hn_pre[o * H + tid];
    };
    if (tid < H) load_step(T - 1, c_d, c_r, c_z, c_n, c_hp, c_hn);
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        float n_d = 0.f, n_r = 0.f, n_z = 0.f, n_n = 0.f, n_hp = 0.f, n_hn = 0.f, direct = 0.f;
        if (tid < H) {
            if (t > 0) load_step(t - 1, n_d, n_r, n_z, n_n, n_hp, n_hn);        // in flight during this step
            const size_t o = (size_t)s * T + t;
            const float d = dh[tid] + c_d;
            const float dn_pre = d * (1.f - c_z) * (1.f - c_n * c_n);
            const float dz_pre = d * (c_hp - c_n) * c_z * (1.f - c_z);
            const float dr_pre = dn_pre * c_hn * c_r * (1.f - c_r);
            float* gi_o = dgi + o * H3;
            float* gh_o = dgh + o * H3;
            gi_o[tid] = dr_pre; gi_o[H + tid] = dz_pre; gi_o[2 * H + tid] = dn_pre;
            gh_o[tid] = dr_pre; gh_o[H + tid] = dz_pre; gh_o[2 * H + tid] = dn_pre * c_r;
            dg[tid] = dr_pre; dg[H + tid] = dz_pre; dg[2 * H + tid] = dn_pre * c_r;
            direct = d * c_z;
        }
        __syncthreads();
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < H; q += 4) {
            const float4 v = *reinterpret_cast<const float4*>(&dg[pq * H + q]);
            a = fmaf(w[q], v.x, a); a = fmaf(w[q + 1], v.y, a); a = fmaf(w[q + 2], v.z, a); a = fmaf(w[q + 3], v.w, a);
        }
        part[pq][j] = a;
        __syncthreads();
        if (tid < H) dh[tid] = direct + part[0][tid] + part[1][tid] + part[2][tid];
        c_d = n_d; c_r = n_r; c_z = n_z; c_n = n_n; c_hp = n_hp; c_hn = n_hn;
        __syncthreads();
    }
    if (dh0 && tid < H) dh0[(size_t)s * H + tid] = dh[tid];
}

// ---------------------------------------------------------------------------------------------------
// conv-GRU element-wise gate kernels; P pixels, C hidden channels, NHWC
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cgru_gates1_kernel(const T* __restrict__ g1, const T* __restrict__ h,
                                                          T* __restrict__ ru, T* __restrict__ rh, int C,
                                                          long long items) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = C / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvecs);
        const long long p = i / cvecs;
        float r[VEC], u[VEC], hh[VEC];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(g1 + p * 2 * C + cv * VEC), r);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(g1 + p * 2 * C + C + cv * VEC), u);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(h + p * C + cv * VEC), hh);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { r[e] = sigmoidf_(r[e]); u[e] = sigmoidf_(u[e]); hh[e] *= r[e]; }
        *reinterpret_cast<uint4*>(ru + p * 2 * C + cv * VEC) = Elem<T>::pack(r);
        *reinterpret_cast<uint4*>(ru + p * 2 * C + C + cv * VEC) = Elem<T>::pack(u);
        *reinterpret_cast<uint4*>(rh + p * C + cv * VEC) = Elem<T>::pack(hh);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cgru_gates2_kernel(const T* __restrict__ g2, const T* __restrict__ ru,
                                                          const T* __restrict__ h, T* __restrict__ o,
                                                          T* __restrict__ hnew, int C, long long items) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = C / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvecs);
        const long long p = i / cvecs;
        float og[VEC], u[VEC], hh[VEC];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(g2 + p * C + cv * VEC), og);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(ru + p * 2 * C + C + cv * VEC), u);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(h + p * C + cv * VEC), hh);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { og[e] = tanhf(og[e]); hh[e] = (1.f - u[e]) * og[e] + u[e] * hh[e]; }
        *reinterpret_cast<uint4*>(o + p * C + cv * VEC) = Elem<T>::pack(og);
        *reinterpret_cast<uint4*>(hnew + p * C + cv * VEC) = Elem<T>::pack(hh);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cgru_gates2_bwd_kernel(const T* __restrict__ dhnew, const T* __restrict__ ru,
                                                              const T* __restrict__ h, const T* __restrict__ o,
                                                              T* __restrict__ dg2, T* __restrict__ du,
                                                              T* __restrict__ dh, int C, long long items) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = C / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvecs);
        const long long p = i / cvecs;
        float d[VEC], u[VEC], hh[VEC], og[VEC], a[VEC], b[VEC], c[VEC];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(dhnew + p * C + cv * VEC), d);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(ru + p * 2 * C + C + cv * VEC), u);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(h + p * C + cv * VEC), hh);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(o + p * C + cv * VEC), og);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            a[e] = d[e] * (1.f - u[e]) * (1.f - og[e] * og[e]);   // d(pre-tanh)
            b[e] = d[e] * (hh[e] - og[e]);                         // d(u), post-sigmoid
            c[e] = d[e] * u[e];                                    // direct path to h
        }
        *reinterpret_cast<uint4*>(dg2 + p * C + cv * VEC) = Elem<T>::pack(a);
        *reinterpret_cast<uint4*>(du + p * 2 * C + cv * VEC) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(du + p * 2 * C + C + cv * VEC) = Elem<T>::pack(b);
        *reinterpret_cast<uint4*>(dh + p * C + cv * VEC) = Elem<T>::pack(c);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cgru_gates1_bwd_kernel(const T* __restrict__ drh, const T* __restrict__ du,
                                                              const T* __restrict__ ru, const T* __restrict__ h,
                                                              T* __restrict__ dg1, T* __restrict__ dh_accum, int C,
                                                              long long items) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = C / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvecs);
        const long long p = i / cvecs;
        float d[VEC], dU[VEC], r[VEC], u[VEC], hh[VEC], acc[VEC];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(drh + p * C + cv * VEC), d);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(du + p * 2 * C + C + cv * VEC), dU);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(ru + p * 2 * C + cv * VEC), r);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(ru + p * 2 * C + C + cv * VEC), u);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(h + p * C + cv * VEC), hh);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            acc[e] = d[e] * r[e];                                   // d(rh) -> h
            d[e] = d[e] * hh[e] * r[e] * (1.f - r[e]);              // -> pre-sigmoid reset gate
            dU[e] = dU[e] * u[e] * (1.f - u[e]);                    // -> pre-sigmoid update gate
        }
        *reinterpret_cast<uint4*>(dg1 + p * 2 * C + cv * VEC) = Elem<T>::pack(d);
        *reinterpret_cast<uint4*>(dg1 + p * 2 * C + C + cv * VEC) = Elem<T>::pack(dU);
        *reinterpret_cast<uint4*>(dh_accum + p * C + cv * VEC) = Elem<T>::pack(acc);
    }
}

// conv-LSTM gate math of CLSTMCell.forward (reference: src/models/common.py:376-385), forward only:
// the reference never propagates a gradient through it (Bottleneck drops tuple states, refine_net.py:168-174).
template <typename T>
__global__ __launch_bounds__(256) void clstm_gates_kernel(const T* __restrict__ g, const T* __restrict__ c_prev,
                                                          T* __restrict__ h, T* __restrict__ c, int C,
                                                          long long items) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = C / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvecs);
        const long long p = i / cvecs;
        float gi[VEC], gf[VEC], go[VEC], gc[VEC], cp[VEC], hh[VEC];
        const T* gp = g + p * 4 * C + cv * VEC;
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(gp), gi);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(gp + C), gf);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(gp + 2 * C), go);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(gp + 3 * C), gc);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(c_prev + p * C + cv * VEC), cp);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            cp[e] = sigmoidf_(gf[e]) * cp[e] + sigmoidf_(gi[e]) * tanhf(gc[e]);
            hh[e] = sigmoidf_(go[e]) * tanhf(cp[e]);
        }
        *reinterpret_cast<uint4*>(c + p * C + cv * VEC) = Elem<T>::pack(cp);
        *reinterpret_cast<uint4*>(h + p * C + cv * VEC) = Elem<T>::pack(hh);
    }
}

static inline unsigned rgrid(long long items) {
    long long b = (items + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace eve

using namespace eve;

extern "C" int eve_gru_scan_fwd(int S, int T, int H, const float* gi, const float* whh_t, const float* bhh,
                                const float* h0, float* hs, float* gates, float* hn_pre, eve_stream_t stream) {
    if (S <= 0 || T <= 0 || H <= 0 || H > 256 || !gi || !whh_t || !bhh || !hs || !gates || !hn_pre)
        return set_error_msg("gru_scan_fwd: bad arguments (H <= 256)");
    if (H == 128) {
        hipLaunchKernelGGL(gru_scan_fwd128_kernel, dim3(S), dim3(384), 0, (hipStream_t)stream, T, gi, whh_t, bhh, h0, hs, gates,
                           hn_pre);
        EVE_CHECK_LAUNCH();
        return 0;
    }
    const int threads = ((3 * H + 63) / 64) * 64;
    hipLaunchKernelGGL(gru_scan_fwd_kernel, dim3(S), dim3(threads), 4 * H * sizeof(float), (hipStream_t)stream,
                       T, H, gi, whh_t, bhh, h0, hs, gates, hn_pre);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_gru_scan_bwd(int S, int T, int H, const float* dhs, const float* whh, const float* h0,
                                const float* hs, const float* gates, const float* hn_pre, float* dgi, float* dgh,
                                float* dh0, eve_stream_t stream) {
    if (S <= 0 || T <= 0 || H <= 0 || H > 256 || !dhs || !whh || !hs || !gates || !hn_pre || !dgi || !dgh)
        return set_error_msg("gru_scan_bwd: bad arguments (H <= 256)");
    if (H == 128) {
        hipLaunchKernelGGL(gru_scan_bwd128_kernel, dim3(S), dim3(384), 0, (hipStream_t)stream, T, dhs, whh, h0, hs, gates, hn_pre,
                           dgi, dgh, dh0);
        EVE_CHECK_LAUNCH();
        return 0;
    }
    const int threads = ((3 * H + 63) / 64) * 64;
    hipLaunchKernelGGL(gru_scan_bwd_kernel, dim3(S), dim3(threads), 4 * H * sizeof(float), (hipStream_t)stream,
                       T, H, dhs, whh, h0, hs, gates, hn_pre, dgi, dgh, dh0);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_rnn_scan_fwd(int S, int T, int H, const float* gi, const float* whh_t, const float* bhh, const float* h0,
                                float* hs, eve_stream_t stream) {
    if (S <= 0 || T <= 0 || H <= 0 || H > 256 || !gi || !whh_t || !bhh || !hs) return set_error_msg("rnn_scan_fwd: bad arguments (H <= 256)");
    const int threads = ((H + 63) / 64) * 64;
    hipLaunchKernelGGL(cell_scan_fwd_kernel<1>, dim3(S), dim3(threads), 2 * H * sizeof(float), (hipStream_t)stream, T, H, gi,
                       whh_t, bhh, h0, (const float*)nullptr, hs, (float*)nullptr, (float*)nullptr);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_rnn_scan_bwd(int S, int T, int H, const float* dhs, const float* whh, const float* hs, float* dpre,
                                float* dh0, eve_stream_t stream) {
    if (S <= 0 || T <= 0 || H <= 0 || H > 256 || !dhs || !whh || !hs || !dpre) return set_error_msg("rnn_scan_bwd: bad arguments (H <= 256)");
    const int threads = ((H + 63) / 64) * 64;
    hipLaunchKernelGGL(cell_scan_bwd_kernel<1>, dim3(S), dim3(threads), 2 * H * sizeof(float), (hipStream_t)stream, T, H, dhs,
                       (const float*)nullptr, whh, (const float*)nullptr, hs, (const float*)nullptr, (const float*)nullptr, dpre, dh0,
                       (float*)nullptr);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_lstm_scan_fwd(int S, int T, int H, const float* gi, const float* whh_t, const float* bhh, const float* h0,
                                 const float* c0, float* hs, float* cs, float* gates, eve_stream_t stream) {
    if (S <= 0 || T <= 0 || H <= 0 || H > 256 || !gi || !whh_t || !bhh || !hs || !cs || !gates)
        return set_error_msg("lstm_scan_fwd: bad arguments (H <= 256)");
    hipLaunchKernelGGL(cell_scan_fwd_kernel<4>, dim3(S), dim3(((4 * H + 63) / 64) * 64), 5 * H * sizeof(float), (hipStream_t)stream,
                       T, H, gi, whh_t, bhh, h0, c0, hs, cs, gates);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_lstm_scan_bwd(int S, int T, int H, const float* dhs, const float* dcs, const float* whh, const float* c0,
                                 const float* hs, const float* cs, const float* gates, float* dpre, float* dh0, float* dc0,
                                 eve_stream_t stream) {
    if (S <= 0 || T <= 0 || H <= 0 || H > 256 || !dhs || !whh || !hs || !cs || !gates || !dpre)
        return set_error_msg("lstm_scan_bwd: bad arguments (H <= 256)");
    hipLaunchKernelGGL(cell_scan_bwd_kernel<4>, dim3(S), dim3(((4 * H + 63) / 64) * 64), 5 * H * sizeof(float), (hipStream_t)stream,
                       T, H, dhs, dcs, whh, c0, hs, cs, gates, dpre, dh0, dc0);
    EVE_CHECK_LAUNCH();
    return 0;
}

#define CG_CHECK(who)                                                                     \
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;                                         \
    if (((unsigned)dtype > (unsigned)EVE_DT_F16) || P <= 0 || C <= 0 || C % vec)    \
        return set_error_msg(who ": bad arguments");                                      \
    const long long items = P * (C / vec);                                                \
    hipStream_t s = (hipStream_t)stream;

extern "C" int eve_cgru_gates1(int dtype, long long P, int C, const void* g1, const void* h, void* ru, void* rh,
                               eve_stream_t stream) {
    CG_CHECK("cgru_gates1")
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(cgru_gates1_kernel<bf16_t>, dim3(rgrid(items)), dim3(256), 0, s, (const bf16_t*)g1, (const bf16_t*)h, (bf16_t*)ru, (bf16_t*)rh, C, items);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(cgru_gates1_kernel<f16_t>, dim3(rgrid(items)), dim3(256), 0, s, (const f16_t*)g1, (const f16_t*)h, (f16_t*)ru, (f16_t*)rh, C, items);
    else                      hipLaunchKernelGGL(cgru_gates1_kernel<float>, dim3(rgrid(items)), dim3(256), 0, s, (const float*)g1, (const float*)h, (float*)ru, (float*)rh, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_cgru_gates2(int dtype, long long P, int C, const void* g2, const void* ru, const void* h, void* o,
                               void* hnew, eve_stream_t stream) {
    CG_CHECK("cgru_gates2")
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(cgru_gates2_kernel<bf16_t>, dim3(rgrid(items)), dim3(256), 0, s, (const bf16_t*)g2, (const bf16_t*)ru, (const bf16_t*)h, (bf16_t*)o, (bf16_t*)hnew, C, items);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(cgru_gates2_kernel<f16_t>, dim3(rgrid(items)), dim3(256), 0, s, (const f16_t*)g2, (const f16_t*)ru, (const f16_t*)h, (f16_t*)o, (f16_t*)hnew, C, items);
    else                      hipLaunchKernelGGL(cgru_gates2_kernel<float>, dim3(rgrid(items)), dim3(256), 0, s, (const float*)g2, (const float*)ru, (const float*)h, (float*)o, (float*)hnew, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_cgru_gates2_bwd(int dtype, long long P, int C, const void* dhnew, const void* ru, const void* h,
                                   const void* o, void* dg2, void* du, void* dh, eve_stream_t stream) {
    CG_CHECK("cgru_gates2_bwd")
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(cgru_gates2_bwd_kernel<bf16_t>, dim3(rgrid(items)), dim3(256), 0, s, (const bf16_t*)dhnew, (const bf16_t*)ru, (const bf16_t*)h, (const bf16_t*)o, (bf16_t*)dg2, (bf16_t*)du, (bf16_t*)dh, C, items);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(cgru_gates2_bwd_kernel<f16_t>, dim3(rgrid(items)), dim3(256), 0, s, (const f16_t*)dhnew, (const f16_t*)ru, (const f16_t*)h, (const f16_t*)o, (f16_t*)dg2, (f16_t*)du, (f16_t*)dh, C, items);
    else                      hipLaunchKernelGGL(cgru_gates2_bwd_kernel<float>, dim3(rgrid(items)), dim3(256), 0, s, (const float*)dhnew, (const float*)ru, (const float*)h, (const float*)o, (float*)dg2, (float*)du, (float*)dh, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_cgru_gates1_bwd(int dtype, long long P, int C, const void* drh, const void* du, const void* ru,
                                   const void* h, void* dg1, void* dh_accum, eve_stream_t stream) {
    CG_CHECK("cgru_gates1_bwd")
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(cgru_gates1_bwd_kernel<bf16_t>, dim3(rgrid(items)), dim3(256), 0, s, (const bf16_t*)drh, (const bf16_t*)du, (const bf16_t*)ru, (const bf16_t*)h, (bf16_t*)dg1, (bf16_t*)dh_accum, C, items);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(cgru_gates1_bwd_kernel<f16_t>, dim3(rgrid(items)), dim3(256), 0, s, (const f16_t*)drh, (const f16_t*)du, (const f16_t*)ru, (const f16_t*)h, (f16_t*)dg1, (f16_t*)dh_accum, C, items);
    else                      hipLaunchKernelGGL(cgru_gates1_bwd_kernel<float>, dim3(rgrid(items)), dim3(256), 0, s, (const float*)drh, (const float*)du, (const float*)ru, (const float*)h, (float*)dg1, (float*)dh_accum, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_clstm_gates_fwd(int dtype, long long P, int C, const void* gates, const void* c_prev, void* h,
                                   void* c, eve_stream_t stream) {
    CG_CHECK("clstm_gates_fwd")
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(clstm_gates_kernel<bf16_t>, dim3(rgrid(items)), dim3(256), 0, s, (const bf16_t*)gates, (const bf16_t*)c_prev, (bf16_t*)h, (bf16_t*)c, C, items);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(clstm_gates_kernel<f16_t>, dim3(rgrid(items)), dim3(256), 0, s, (const f16_t*)gates, (const f16_t*)c_prev, (f16_t*)h, (f16_t*)c, C, items);
    else                      hipLaunchKernelGGL(clstm_gates_kernel<float>, dim3(rgrid(items)), dim3(256), 0, s, (const float*)gates, (const float*)c_prev, (float*)h, (float*)c, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
