// Clip-long scans of RefineNet's conv-RNN bottleneck cells in FLOAT32: CGRUCell (forward + backward), CRNNCell (forward +
// backward) and CLSTMCell (forward only: the reference never back-propagates through it, refine_net.py:168-174) --
// /root/reference/src/models/common.py:331-352 (CRNN), :355-385 (CLSTM), :388-415 (CGRU), applied per frame by
// refine_net.py:132-176.  One persistent launch walks all T frames of a clip; before round 5 the float32 parity mode and the
// CLSTM / CRNN cells ran T x (1-2 convolution launches + gate kernels + concatenations).
//
// Geometry is fixed by the model: 5 x 8 pixels, 64 hidden + 64 input channels.  One workgroup (512 threads, 8 waves) per
// sequence.  In LDS, as floats: the zero-bordered 7 x 10 halo of the 128-channel convolution input (pixel stride 132 floats so
// that the 16 pixels of a fragment read fall into different banks), the convolution output [40][<= 256], and the state(s).
// A convolution is an implicit GEMM on v_mfma_f32_16x16x4_f32 (exact float32: an fmaf chain) with A = filter rows (16 output
// channels x 4 k) straight from global memory / L2 as one 16-byte load per lane and 16-channel K block, B = 16 pixels x 4 k as
// one ds_read_b128 from the halo; the four words of a lane's vector feed four MFMAs (K permutation: MFMA s takes word s of every
// lane's vector; a sum is order-free, conv_igemm.hip uses the same trick), so a lane ends up with 4 consecutive output channels
// of one pixel.  The 40 pixels are 2.5 tiles of 16: the third tile's upper half re-reads pixel 39 and is dropped.
// Filter blocks are prefetched one K block ahead.  The matrix pipe runs float32 at the vector rate (157 TFLOP/s chip-wide), so
// a frame is MFMA-time bound at ~13 us (gates_1) + ~7 us (gate_2): the T-sequential floor of this formulation with 8 waves.
//
// Weight / bias gradients are NOT formed here: like the 16-bit scan (cgru_scan.hip) the backward emits the gradients of the
// pre-activations for all frames and the caller runs ONE batched weight-gradient launch over the T*B frames.
#include "common.h"

namespace eve {

constexpr int CS_H = 5, CS_W = 8, CS_PIX = 40, CS_C = 64;
constexpr int CS_STR = 132;                       // floats per halo pixel (128 channels + 4: bank spread)
constexpr int CS_HALO = 7 * 10 * CS_STR;          // floats
constexpr int CS_NT = 512;

__device__ __forceinline__ float cs_sigmoid(float z) { return 1.f / (1.f + __expf(-z)); }   // = recurrent.hip's sigmoidf_

// halo offset (floats) of pixel p's centre
__device__ __forceinline__ int cs_halo_at(int p) { return (((p >> 3) + 1) * 10 + (p & 7) + 1) * CS_STR; }

// acc[nt][pt] += sum over K blocks kb in [kb0, kb1) of W[co][tap][ci] * halo[pixel + tap][ci]
//   W: [COUT][9][CIN] floats (OHWI for a forward convolution; IHWO with FLIP for a data gradient: out[p] takes dy[p - (tap - 1)])
//   a K block = 16 consecutive channels of one tap; kb = tap * (CIN / 16) + block
template <int CIN, int NTILE, bool FLIP>
__device__ __forceinline__ void cs_conv(const float* halo, const float* __restrict__ W, const int co0, const int kb0, const int kb1,
                                        f32x4_t (&acc)[NTILE][3], const int lane) {
    constexpr int KB_PER_TAP = CIN / 16;
    const int i = lane & 15, kk = lane >> 4;
    int boff[3];
#pragma unroll
    for (int pt = 0; pt < 3; ++pt) {
        const int p = min(pt * 16 + i, CS_PIX - 1);
        boff[pt] = (((p >> 3)) * 10 + (p & 7)) * CS_STR + 4 * kk;          // + tap offset (dy * 10 + dx) * CS_STR
    }
    const float* wrow[NTILE];
#pragma unroll
    for (int nt = 0; nt < NTILE; ++nt) wrow[nt] = W + (size_t)(co0 + nt * 16 + i) * (9 * CIN) + 4 * kk;
    float4 a_next[NTILE];
#pragma unroll
    for (int nt = 0; nt < NTILE; ++nt) a_next[nt] = *reinterpret_cast<const float4*>(wrow[nt] + kb0 * 16);
    for (int kb = kb0; kb < kb1; ++kb) {
        float4 a[NTILE];
#pragma unroll
        for (int nt = 0; nt < NTILE; ++nt) a[nt] = a_next[nt];
        const int kn = min(kb + 1, kb1 - 1);
#pragma unroll
        for (int nt = 0; nt < NTILE; ++nt) a_next[nt] = *reinterpret_cast<const float4*>(wrow[nt] + kn * 16);
        const int tap = kb / KB_PER_TAP, blk = kb - tap * KB_PER_TAP;
        const int kh = tap / 3, kw = tap - kh * 3;
        const int dy = FLIP ? 2 - kh : kh, dx = FLIP ? 2 - kw : kw;
        const int toff = (dy * 10 + dx) * CS_STR + blk * 16;
        float4 b[3];
#pragma unroll
        for (int pt = 0; pt < 3; ++pt) b[pt] = *reinterpret_cast<const float4*>(halo + boff[pt] + toff);
#pragma unroll
        for (int nt = 0; nt < NTILE; ++nt)
#pragma unroll
            for (int pt = 0; pt < 3; ++pt) {
                acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt].x, b[pt].x, acc[nt][pt], 0, 0, 0);
                acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt].y, b[pt].y, acc[nt][pt], 0, 0, 0);
                acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt].z, b[pt].z, acc[nt][pt], 0, 0, 0);
                acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt].w, b[pt].w, acc[nt][pt], 0, 0, 0);
            }
    }
}

// out[p][co0 + 16 nt + ..] (= or +=) acc (+ bias): lane holds channels co0 + 16 nt + 4 (lane / 16) + r of pixel 16 pt + lane % 16
template <int NTILE, bool ADD>
__device__ __forceinline__ void cs_store(float* out, const int ostr, const int co0, const float* __restrict__ bias,
                                         const f32x4_t (&acc)[NTILE][3], const int lane) {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NTILE; ++nt) {
        const int co = co0 + nt * 16 + 4 * g;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *reinterpret_cast<const float4*>(bias + co);
#pragma unroll
        for (int pt = 0; pt < 3; ++pt) {
            const int p = pt * 16 + j;
            if (p < CS_PIX) {
                float4* dst = reinterpret_cast<float4*>(out + p * ostr + co);
                float4 v = make_float4(acc[nt][pt][0] + bv.x, acc[nt][pt][1] + bv.y, acc[nt][pt][2] + bv.z, acc[nt][pt][3] + bv.w);
                if (ADD) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *dst = v;
            }
        }
    }
}

template <int NTILE>
__device__ __forceinline__ void cs_zero(f32x4_t (&acc)[NTILE][3]) {
#pragma unroll
    for (int nt = 0; nt < NTILE; ++nt)
#pragma unroll
        for (int pt = 0; pt < 3; ++pt) acc[nt][pt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}

// A whole convolution of the workgroup's halo into out[40][ostr]:  COUT / 16 tiles over the 8 waves.
//   COUT = 256: two tiles per wave;  128: one;  64: one tile per wave PAIR, the pair splits K and the second half adds after
//   a barrier (fixed order: the result does not depend on timing).  Ends with a barrier: `out` is complete, the halo is free.
template <int CIN, int COUT, bool FLIP>
__device__ __forceinline__ void cs_conv_all(const float* halo, const float* __restrict__ W, const float* __restrict__ bias,
                                            float* out, const int ostr, const int wave, const int lane) {
    constexpr int KB = 9 * CIN / 16;
    if constexpr (COUT == 256) {
        f32x4_t acc[2][3];
        cs_zero<2>(acc);
        cs_conv<CIN, 2, FLIP>(halo, W, wave * 32, 0, KB, acc, lane);
        cs_store<2, false>(out, ostr, wave * 32, bias, acc, lane);
    } else if constexpr (COUT == 128) {
        f32x4_t acc[1][3];
        cs_zero<1>(acc);
        cs_conv<CIN, 1, FLIP>(halo, W, wave * 16, 0, KB, acc, lane);
        cs_store<1, false>(out, ostr, wave * 16, bias, acc, lane);
    } else {
        static_assert(COUT == 64, "COUT");
        const int tile = wave & 3, half = wave >> 2;
        f32x4_t acc[1][3];
        cs_zero<1>(acc);
        cs_conv<CIN, 1, FLIP>(halo, W, tile * 16, half * (KB / 2), half ? KB : KB / 2, acc, lane);
        if (half == 0) cs_store<1, false>(out, ostr, tile * 16, bias, acc, lane);
        __syncthreads();
        if (half == 1) cs_store<1, true>(out, ostr, tile * 16, nullptr, acc, lane);
    }
    __syncthreads();
}

__device__ __forceinline__ void cs_zero_halo(float* halo, const int tid) {
    for (int q = tid; q < CS_HALO; q += CS_NT) halo[q] = 0.f;
}

// the 5 elements a thread owns in every element-wise phase: e = tid + 512 k -> pixel e / 64, channel e % 64
#define CS_FOR_ELEMS(k, p, c) _Pragma("unroll") for (int k = 0, p = tid >> 6, c = tid & 63; k < 5; ++k, p += 8)

// ------------------------------------------------------------------------------------------------------------------------
// CGRU forward.  xs [B][T][40][64]; h0 [B][40][64] or null; w1 OHWI [128][9][128] (inputs: x then h), w2 OHWI [64][9][128]
// (inputs: r*h then x).  Outputs: hs [B][T][40][64]; time-major hs_tm, rh, og [T][B][40][64], ru [T][B][40][128].
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CS_NT) void cgru_scan_f32_fwd_kernel(const int B, const int T, const float* __restrict__ xs,
                                                                  const float* __restrict__ h0, const float* __restrict__ w1,
                                                                  const float* __restrict__ b1, const float* __restrict__ w2,
                                                                  const float* __restrict__ b2, float* __restrict__ hs,
                                                                  float* __restrict__ hs_tm, float* __restrict__ ru,
                                                                  float* __restrict__ rh, float* __restrict__ og) {
    extern __shared__ __attribute__((aligned(16))) float cs_lds[];
    float* halo = cs_lds;
    float* g = halo + CS_HALO;                       // [40][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    cs_zero_halo(halo, tid);
    float h[5], x[5], u[5];
    CS_FOR_ELEMS(k, p, c) h[k] = h0 ? h0[((size_t)b * CS_PIX + p) * CS_C + c] : 0.f;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const float* xt = xs + ((size_t)b * T + t) * (CS_PIX * CS_C);
        const size_t tm = ((size_t)t * B + b) * CS_PIX;
        CS_FOR_ELEMS(k, p, c) {
            x[k] = xt[p * CS_C + c];
            halo[cs_halo_at(p) + c] = x[k];
            halo[cs_halo_at(p) + CS_C + c] = h[k];
        }
        __syncthreads();
        cs_conv_all<128, 128, false>(halo, w1, b1, g, 128, wave, lane);
        CS_FOR_ELEMS(k, p, c) {
            const float r = cs_sigmoid(g[p * 128 + c]);
            u[k] = cs_sigmoid(g[p * 128 + CS_C + c]);
            const float v = r * h[k];
            ru[(tm + p) * 128 + c] = r;
            ru[(tm + p) * 128 + CS_C + c] = u[k];
            rh[(tm + p) * CS_C + c] = v;
            halo[cs_halo_at(p) + c] = v;
            halo[cs_halo_at(p) + CS_C + c] = x[k];
        }
        __syncthreads();
        cs_conv_all<128, 64, false>(halo, w2, b2, g, 128, wave, lane);
        CS_FOR_ELEMS(k, p, c) {
            const float o = tanhf(g[p * 128 + c]);
            h[k] = (1.f - u[k]) * o + u[k] * h[k];
            og[(tm + p) * CS_C + c] = o;
            hs_tm[(tm + p) * CS_C + c] = h[k];
            hs[(((size_t)b * T + t) * CS_PIX + p) * CS_C + c] = h[k];
        }
        // (the next frame's halo writes follow conv_all's closing barrier; its reads of g precede the next conv's writes by the
        //  barrier after the halo fill)
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// CGRU backward (common.py:400-415 differentiated; the per-frame kernels are recurrent.hip's cgru_gates{2,1}_bwd).
// Time-major inputs dhs_tm, og, hs_tm [T][B][40][64], ru [T][B][40][128]; w1t IHWO [128][9][128], w2t IHWO [128][9][64].
// Outputs dg1_all [T][B][40][128], dg2_all, dxs_tm [T][B][40][64], dh0 [B][40][64] or null.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CS_NT) void cgru_scan_f32_bwd_kernel(const int B, const int T, const float* __restrict__ dhs_tm,
                                                                  const float* __restrict__ ru, const float* __restrict__ og,
                                                                  const float* __restrict__ hs_tm, const float* __restrict__ h0,
                                                                  const float* __restrict__ w1t, const float* __restrict__ w2t,
                                                                  float* __restrict__ dg1_all, float* __restrict__ dg2_all,
                                                                  float* __restrict__ dxs_tm, float* __restrict__ dh0) {
    extern __shared__ __attribute__((aligned(16))) float cs_lds[];
    float* halo = cs_lds;
    float* g = halo + CS_HALO;                       // [40][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    cs_zero_halo(halo, tid);
    float carry[5], du[5], dhd[5], r[5], u[5], hp[5], dx2[5];
    CS_FOR_ELEMS(k, p, c) carry[k] = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        const size_t tm = ((size_t)t * B + b) * CS_PIX;
        CS_FOR_ELEMS(k, p, c) {
            const float d = dhs_tm[(tm + p) * CS_C + c] + carry[k];
            r[k] = ru[(tm + p) * 128 + c];
            u[k] = ru[(tm + p) * 128 + CS_C + c];
            const float o = og[(tm + p) * CS_C + c];
            hp[k] = t > 0 ? hs_tm[(((size_t)(t - 1) * B + b) * CS_PIX + p) * CS_C + c]
                          : (h0 ? h0[((size_t)b * CS_PIX + p) * CS_C + c] : 0.f);
            const float a = d * (1.f - u[k]) * (1.f - o * o);            // d(pre-tanh)
            du[k] = d * (hp[k] - o);                                     // d(u), post-sigmoid
            dhd[k] = d * u[k];                                           // direct path to h
            dg2_all[(tm + p) * CS_C + c] = a;
            halo[cs_halo_at(p) + c] = a;
        }
        __syncthreads();
        cs_conv_all<64, 128, true>(halo, w2t, nullptr, g, 128, wave, lane);       // d[r*h | x]
        CS_FOR_ELEMS(k, p, c) {
            const float drh = g[p * 128 + c];
            dx2[k] = g[p * 128 + CS_C + c];
            const float a = drh * hp[k] * r[k] * (1.f - r[k]);           // -> pre-sigmoid reset gate
            const float bq = du[k] * u[k] * (1.f - u[k]);                // -> pre-sigmoid update gate
            dhd[k] += drh * r[k];                                        // d(rh) -> h
            dg1_all[(tm + p) * 128 + c] = a;
            dg1_all[(tm + p) * 128 + CS_C + c] = bq;
            halo[cs_halo_at(p) + c] = a;
            halo[cs_halo_at(p) + CS_C + c] = bq;
        }
        __syncthreads();
        cs_conv_all<128, 128, true>(halo, w1t, nullptr, g, 128, wave, lane);      // d[x | h]
        CS_FOR_ELEMS(k, p, c) {
            dxs_tm[(tm + p) * CS_C + c] = g[p * 128 + c] + dx2[k];
            carry[k] = dhd[k] + g[p * 128 + CS_C + c];
        }
        __syncthreads();                                                 // g is read above, written by the next frame's first conv
    }
    if (dh0) CS_FOR_ELEMS(k, p, c) dh0[((size_t)b * CS_PIX + p) * CS_C + c] = carry[k];
}

// ------------------------------------------------------------------------------------------------------------------------
// CRNN: h_t = tanh(conv([x_t | h_{t-1}]) + b)   (common.py:331-352).  w OHWI [64][9][128]; wt IHWO [128][9][64].
// forward: hs [B][T][40][64] (+ time-major copy hs_tm);  backward: dpre_all, dxs_tm [T][B][40][64], dh0.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CS_NT) void crnn_scan_f32_fwd_kernel(const int B, const int T, const float* __restrict__ xs,
                                                                  const float* __restrict__ h0, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ hs,
                                                                  float* __restrict__ hs_tm) {
    extern __shared__ __attribute__((aligned(16))) float cs_lds[];
    float* halo = cs_lds;
    float* g = halo + CS_HALO;                       // [40][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    cs_zero_halo(halo, tid);
    float h[5];
    CS_FOR_ELEMS(k, p, c) h[k] = h0 ? h0[((size_t)b * CS_PIX + p) * CS_C + c] : 0.f;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const float* xt = xs + ((size_t)b * T + t) * (CS_PIX * CS_C);
        CS_FOR_ELEMS(k, p, c) {
            halo[cs_halo_at(p) + c] = xt[p * CS_C + c];
            halo[cs_halo_at(p) + CS_C + c] = h[k];
        }
        __syncthreads();
        cs_conv_all<128, 64, false>(halo, w, bias, g, 64, wave, lane);
        CS_FOR_ELEMS(k, p, c) {
            h[k] = tanhf(g[p * 64 + c]);
            hs_tm[(((size_t)t * B + b) * CS_PIX + p) * CS_C + c] = h[k];
            hs[(((size_t)b * T + t) * CS_PIX + p) * CS_C + c] = h[k];
        }
    }
}

__global__ __launch_bounds__(CS_NT) void crnn_scan_f32_bwd_kernel(const int B, const int T, const float* __restrict__ dhs_tm,
                                                                  const float* __restrict__ hs_tm, const float* __restrict__ wt,
                                                                  float* __restrict__ dpre_all, float* __restrict__ dxs_tm,
                                                                  float* __restrict__ dh0) {
    extern __shared__ __attribute__((aligned(16))) float cs_lds[];
    float* halo = cs_lds;
    float* g = halo + CS_HALO;                       // [40][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    cs_zero_halo(halo, tid);
    float carry[5];
    CS_FOR_ELEMS(k, p, c) carry[k] = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        const size_t tm = ((size_t)t * B + b) * CS_PIX;
        CS_FOR_ELEMS(k, p, c) {
            const float hn = hs_tm[(tm + p) * CS_C + c];
            const float a = (dhs_tm[(tm + p) * CS_C + c] + carry[k]) * (1.f - hn * hn);
            dpre_all[(tm + p) * CS_C + c] = a;
            halo[cs_halo_at(p) + c] = a;
        }
        __syncthreads();
        cs_conv_all<64, 128, true>(halo, wt, nullptr, g, 128, wave, lane);        // d[x | h]
        CS_FOR_ELEMS(k, p, c) {
            dxs_tm[(tm + p) * CS_C + c] = g[p * 128 + c];
            carry[k] = g[p * 128 + CS_C + c];
        }
        __syncthreads();
    }
    if (dh0) CS_FOR_ELEMS(k, p, c) dh0[((size_t)b * CS_PIX + p) * CS_C + c] = carry[k];
}

// ------------------------------------------------------------------------------------------------------------------------
// CLSTM forward (common.py:355-385; gate order in / forget / out / cell).  w OHWI [256][9][128].  hs, cs [B][T][40][64].
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CS_NT) void clstm_scan_f32_fwd_kernel(const int B, const int T, const float* __restrict__ xs,
                                                                   const float* __restrict__ h0, const float* __restrict__ c0,
                                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                                   float* __restrict__ hs, float* __restrict__ cs) {
    extern __shared__ __attribute__((aligned(16))) float cs_lds[];
    float* halo = cs_lds;
    float* g = halo + CS_HALO;                       // [40][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    cs_zero_halo(halo, tid);
    float h[5], cc[5];
    CS_FOR_ELEMS(k, p, c) {
        h[k] = h0 ? h0[((size_t)b * CS_PIX + p) * CS_C + c] : 0.f;
        cc[k] = c0 ? c0[((size_t)b * CS_PIX + p) * CS_C + c] : 0.f;
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const float* xt = xs + ((size_t)b * T + t) * (CS_PIX * CS_C);
        CS_FOR_ELEMS(k, p, c) {
            halo[cs_halo_at(p) + c] = xt[p * CS_C + c];
            halo[cs_halo_at(p) + CS_C + c] = h[k];
        }
        __syncthreads();
        cs_conv_all<128, 256, false>(halo, w, bias, g, 256, wave, lane);
        CS_FOR_ELEMS(k, p, c) {
            const float gi = g[p * 256 + c], gf = g[p * 256 + CS_C + c], go = g[p * 256 + 2 * CS_C + c], gc = g[p * 256 + 3 * CS_C + c];
            cc[k] = cs_sigmoid(gf) * cc[k] + cs_sigmoid(gi) * tanhf(gc);
            h[k] = cs_sigmoid(go) * tanhf(cc[k]);
            const size_t o = (((size_t)b * T + t) * CS_PIX + p) * CS_C + c;
            hs[o] = h[k];
            cs[o] = cc[k];
        }
    }
}

constexpr size_t CS_LDS_128 = (size_t)(CS_HALO + CS_PIX * 128) * sizeof(float);
constexpr size_t CS_LDS_256 = (size_t)(CS_HALO + CS_PIX * 256) * sizeof(float);

static void cs_set_attrs() {
    static bool done = false;
    if (done) return;
    (void)hipFuncSetAttribute((const void*)cgru_scan_f32_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CS_LDS_128);
    (void)hipFuncSetAttribute((const void*)cgru_scan_f32_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CS_LDS_128);
    (void)hipFuncSetAttribute((const void*)crnn_scan_f32_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CS_LDS_128);
    (void)hipFuncSetAttribute((const void*)crnn_scan_f32_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CS_LDS_128);
    (void)hipFuncSetAttribute((const void*)clstm_scan_f32_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CS_LDS_256);
    done = true;
}

}  // namespace eve

using namespace eve;

/* float32 instantiation of eve_cgru_scan_fwd / _bwd (cgru_scan.hip dispatches here for EVE_DT_F32): same operands, float. */
int eve_cgru_scan_f32_fwd(int B, int T, const float* xs, const float* h0, const float* w1, const float* b1, const float* w2,
                          const float* b2, float* hs, float* hs_tm, float* ru, float* rh, float* og, hipStream_t s) {
    cs_set_attrs();
    EVE_LAUNCH("cgru_scan_f32_fwd_kernel", cgru_scan_f32_fwd_kernel, dim3(B), dim3(CS_NT), CS_LDS_128, s, B, T, xs, h0, w1, b1, w2, b2,
               hs, hs_tm, ru, rh, og);
    EVE_CHECK_LAUNCH();
    return 0;
}

int eve_cgru_scan_f32_bwd(int B, int T, const float* dhs_tm, const float* ru, const float* og, const float* hs_tm, const float* h0,
                          const float* w1t, const float* w2t, float* dg1_all, float* dg2_all, float* dxs_tm, float* dh0,
                          hipStream_t s) {
    cs_set_attrs();
    EVE_LAUNCH("cgru_scan_f32_bwd_kernel", cgru_scan_f32_bwd_kernel, dim3(B), dim3(CS_NT), CS_LDS_128, s, B, T, dhs_tm, ru, og, hs_tm, h0,
               w1t, w2t, dg1_all, dg2_all, dxs_tm, dh0);
    EVE_CHECK_LAUNCH();
    return 0;
}

/* CRNNCell over a clip in one launch (float32; common.py:331-352).  xs [B][T][5][8][64], h0 [B][5][8][64] or NULL, w OHWI
   [64][3][3][128] (input channels: x then h), bias [64].  Outputs hs [B][T][5][8][64] and the time-major copy hs_tm
   [T][B][5][8][64] the backward reads. */
extern "C" int eve_crnn_scan_fwd(int B, int T, const float* xs, const float* h0, const float* w, const float* bias, float* hs,
                                 float* hs_tm, eve_stream_t stream) {
    if (B <= 0 || T <= 0 || !xs || !w || !bias || !hs || !hs_tm) return set_error_msg("crnn_scan_fwd: bad arguments");
    cs_set_attrs();
    EVE_LAUNCH("crnn_scan_f32_fwd_kernel", crnn_scan_f32_fwd_kernel, dim3(B), dim3(CS_NT), CS_LDS_128, (hipStream_t)stream, B, T, xs, h0, w,
               bias, hs, hs_tm);
    EVE_CHECK_LAUNCH();
    return 0;
}

/* Backward of eve_crnn_scan_fwd.  Time-major dhs_tm, hs_tm [T][B][5][8][64]; wt = the filter bank IHWO [128][3][3][64].
   Outputs (time-major): dpre_all = gradient of the pre-activation (what the batched weight / bias gradients read), dxs_tm;
   dh0 [B][5][8][64] or NULL. */
extern "C" int eve_crnn_scan_bwd(int B, int T, const float* dhs_tm, const float* hs_tm, const float* wt, float* dpre_all,
                                 float* dxs_tm, float* dh0, eve_stream_t stream) {
    if (B <= 0 || T <= 0 || !dhs_tm || !hs_tm || !wt || !dpre_all || !dxs_tm) return set_error_msg("crnn_scan_bwd: bad arguments");
    cs_set_attrs();
    EVE_LAUNCH("crnn_scan_f32_bwd_kernel", crnn_scan_f32_bwd_kernel, dim3(B), dim3(CS_NT), CS_LDS_128, (hipStream_t)stream, B, T, dhs_tm,
               hs_tm, wt, dpre_all, dxs_tm, dh0);
    EVE_CHECK_LAUNCH();
    return 0;
}

/* CLSTMCell over a clip in one launch (float32, forward only: the reference drops tuple states from the feature path,
   refine_net.py:168-174; common.py:355-385).  w OHWI [256][3][3][128] (gate order in / forget / out / cell), bias [256];
   h0 / c0 [B][5][8][64] or NULL.  Outputs hs, cs [B][T][5][8][64]. */
extern "C" int eve_clstm_scan_fwd(int B, int T, const float* xs, const float* h0, const float* c0, const float* w, const float* bias,
                                  float* hs, float* cs, eve_stream_t stream) {
    if (B <= 0 || T <= 0 || !xs || !w || !bias || !hs || !cs) return set_error_msg("clstm_scan_fwd: bad arguments");
    cs_set_attrs();
    EVE_LAUNCH("clstm_scan_f32_fwd_kernel", clstm_scan_f32_fwd_kernel, dim3(B), dim3(CS_NT), CS_LDS_256, (hipStream_t)stream, B, T, xs, h0,
               c0, w, bias, hs, cs);
    EVE_CHECK_LAUNCH();
    return 0;
}
