// 1x1 / stride 1 convolutions between 16..128 channels as a STREAMING kernel (round 4).
// RefineNet's skip_layer convolutions (refine_net.py:59-60 of the reference: nn.Conv2d(ic, oc, kernel_size=1)) and their data
// gradients on the two outer levels (72x128 and 36x64 pixels, 960 frames: 2.2-8.8 M pixels, 16-128 channels) move 0.7-1.7 GB
// per launch for 4-70 GFLOP: pure HBM streaming.  They ran on the LDS-DMA gather kernel through pixel grouping (F pixels as one
// 64-channel pixel and a block-diagonal filter, ops.PAIR_FACTOR_1X1) at 1.6-2.3 TB/s: that kernel stages a 64-channel K step of
// a 128-256-pixel tile through LDS behind a workgroup barrier, for a product that needs no data reuse at all.
// Here there is no LDS and no barrier: the filter (<= 128 x 128) lives in each wave's registers as MFMA A operands, a wave
// reads 16-pixel tiles as B operands straight from global memory (lane = pixel t, 8 consecutive channels g: 16 contiguous
// bytes; a wave instruction covers 1 KB contiguous for 32 channels), and the transposed product
//        D[cout][pixel] = sum_k W[cout][k] * X[pixel][k]
// leaves every lane with consecutive OUTPUT CHANNELS of ONE pixel -- the rows of two 16-row tiles are dealt so that lane (t, g)
// holds channels 32a + 8g .. 8g + 7 of pixel t: one 16-byte store, 64 contiguous bytes per pixel and tile pair.
// All loads of a batch (TB tiles, and the tensor accumulated into) are issued before the first MFMA; bounds are the buffer
// resource's (out-of-range loads return 0, stores are dropped): no branch in the loop.
// 16 input channels: a load instruction covers 32 pixels (lanes g < 2: pixels 0..15, g >= 2: pixels 16..31, 8 channels each);
// the two halves go through two MFMAs whose B operand is zeroed in the other half's lanes, the filter sits in both k halves.
#pragma once
#include "common.h"

namespace eve {

typedef unsigned int c1_v4u32 __attribute__((ext_vector_type(4)));
typedef unsigned int c1_v2u32 __attribute__((ext_vector_type(2)));

template <int CIN, int COUT>
struct C1Geom {
    static constexpr int KS = CIN <= 32 ? 1 : CIN / 32;        // MFMA k steps (32 channels each)
    static constexpr int NT = COUT / 16;                        // 16-row output tiles
    static constexpr int TB = (KS * NT <= 4) ? 4 : 2;           // 16-pixel tiles per batch and wave
    static constexpr int PIX = 16 * TB;                         // pixels per batch
};

template <typename H, int CIN, int COUT, bool ACC>
__global__ __launch_bounds__(256) void conv1x1_stream_kernel(const H* __restrict__ x, const H* __restrict__ w, const float* __restrict__ bias,
                                                             int act, H* __restrict__ out, uint32_t M, uint32_t batches_per_wave) {
    using G = C1Geom<CIN, COUT>;
    constexpr int KS = G::KS, NT = G::NT, TB = G::TB;
    constexpr int NP = NT >= 2 ? NT / 2 : 1;                    // output pieces per pixel and lane (16 bytes each; 8 for COUT = 16)
    const int lane = threadIdx.x & 63, t = lane & 15, g = lane >> 4;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    // ---- the filter as A operands: tile a, row r = t <-> output channel co(a, t) ----
    uint4 wa[NT][KS];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const int co = NT >= 2 ? 32 * (a >> 1) + 8 * (t >> 2) + 4 * (a & 1) + (t & 3) : t;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = CIN == 16 ? 8 * (g & 1) : 32 * ks + 8 * g;
            wa[a][ks] = *reinterpret_cast<const uint4*>(w + (size_t)co * CIN + k0);
        }
    }
    float bv[NP][8];
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = NT >= 2 ? 32 * a + 8 * g + e : 4 * g + (e & 3);
            bv[a][e] = bias ? bias[c] : 0.f;
        }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(M * (uint32_t)(CIN * 2)), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)(M * (uint32_t)(COUT * 2)), 0x00020000);
    for (uint32_t b = 0; b < batches_per_wave; ++b) {
        const uint32_t p0 = (wave * batches_per_wave + b) * (uint32_t)G::PIX;
        if (p0 >= M) break;                                                    // wave-uniform
        // ---- loads: the pixels (B operands) and, when accumulating, what the output holds ----
        uint4 xb[CIN == 16 ? TB / 2 : TB][KS];
        if constexpr (CIN == 16) {
#pragma unroll
            for (int j = 0; j < TB / 2; ++j) {
                const uint32_t p = p0 + 32u * j + 16u * (g >> 1) + t;
                xb[j][0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(p * 32u + 16u * (g & 1)), 0, 0));
            }
        } else {
#pragma unroll
            for (int i = 0; i < TB; ++i)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const uint32_t p = p0 + 16u * i + t;
                    xb[i][ks] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(p * (uint32_t)(CIN * 2) + 64u * ks + 16u * g), 0, 0));
                }
        }
        uint4 prev[ACC ? TB : 1][NP];
        if constexpr (ACC) {
#pragma unroll
            for (int i = 0; i < TB; ++i)
#pragma unroll
                for (int a = 0; a < NP; ++a) {
                    const uint32_t p = p0 + 16u * i + t;
                    if constexpr (NT >= 2) {
                        prev[i][a] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ro, (int)(p * (uint32_t)(COUT * 2) + 64u * a + 16u * g), 0, 0));
                    } else {
                        const c1_v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(ro, (int)(p * 32u + 8u * g), 0, 0);
                        prev[i][a] = make_uint4(v.x, v.y, 0u, 0u);
                    }
                }
        }
        // ---- products ----
        f32x4_t acc[TB][NT];
#pragma unroll
        for (int i = 0; i < TB; ++i)
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                acc[i][a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if constexpr (CIN == 16) {
                    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
                    const bool mine = (g >> 1) == (i & 1);          // this lane's load holds a pixel of tile i (else: of its neighbour)
                    Elem<H>::mfma(acc[i][a], wa[a][0], mine ? xb[i >> 1][0] : z);
                } else {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) Elem<H>::mfma(acc[i][a], wa[a][ks], xb[i][ks]);
                }
            }
        // ---- epilogue: bias, activation, accumulate, one store per pixel and tile pair ----
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            const uint32_t p = p0 + 16u * i + t;
#pragma unroll
            for (int a = 0; a < NP; ++a) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = acc[i][NT >= 2 ? 2 * a : 0][e] + bv[a][e];
                    o[4 + e] = NT >= 2 ? acc[i][NT >= 2 ? 2 * a + 1 : 0][e] + bv[a][4 + e] : 0.f;
                }
                if (act != EVE_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = act_fwd(o[e], act);
                }
                if constexpr (ACC) {
                    float pv[8];
                    Elem<H>::unpack(prev[i][a], pv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += pv[e];
                }
                const uint4 q = Elem<H>::pack(o);
                // (the whole offset in the vector operand, immediate 0 as the scalar one: see norm_fused.hip `stv`)
                if constexpr (NT >= 2) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(c1_v4u32, q), ro, (int)(p * (uint32_t)(COUT * 2) + 64u * a + 16u * g), 0, 0);
                } else {
                    c1_v2u32 v;
                    v.x = q.x; v.y = q.y;
                    __builtin_amdgcn_raw_buffer_store_b64(v, ro, (int)(p * 32u + 8u * g), 0, 0);
                }
            }
        }
    }
}

// true: launched.  x [M][CIN] -> out [M][COUT] (+= with EVE_EPI_ACC), w [COUT][CIN]
template <typename H>
static bool launch_conv1x1_stream(long long M, int Cin, int Cout, const void* x, const void* w, const float* bias, int epi_act,
                                  void* out, hipStream_t s) {
    if (!g_cfg.conv1x1_stream || M < 16384 || (M + 64) * (long long)Cin * 2 >= (1ll << 32) || (M + 64) * (long long)Cout * 2 >= (1ll << 32)) return false;
    const int act = epi_act & 0xff;
    const bool accf = (epi_act & EVE_EPI_ACC) != 0;
#define EVE_C1_CASE(CI, CO)                                                                                                        \
    if (Cin == CI && Cout == CO) {                                                                                                 \
        using G = C1Geom<CI, CO>;                                                                                                  \
        const long long batches = (M + G::PIX - 1) / G::PIX;                                                                       \
        long long bpw = batches / (4 * 4096);                                                                                      \
        bpw = bpw < 1 ? 1 : (bpw > 8 ? 8 : bpw);                                                                                   \
        const unsigned grid = (unsigned)((batches + 4 * bpw - 1) / (4 * bpw));                                                     \
        if (accf) EVE_LAUNCH(EVE_HNAME(H, "conv1x1_stream_kernel<", ", " #CI ", " #CO ", true>"), (conv1x1_stream_kernel<H, CI, CO, true>),  \
                             dim3(grid), dim3(256), 0, s, (const H*)x, (const H*)w, bias, act, (H*)out, (uint32_t)M, (uint32_t)bpw);         \
        else EVE_LAUNCH(EVE_HNAME(H, "conv1x1_stream_kernel<", ", " #CI ", " #CO ", false>"), (conv1x1_stream_kernel<H, CI, CO, false>),     \
                        dim3(grid), dim3(256), 0, s, (const H*)x, (const H*)w, bias, act, (H*)out, (uint32_t)M, (uint32_t)bpw);              \
        return true;                                                                                                               \
    }
    EVE_C1_CASE(16, 32) EVE_C1_CASE(32, 16) EVE_C1_CASE(16, 64) EVE_C1_CASE(64, 16) EVE_C1_CASE(32, 64) EVE_C1_CASE(64, 32)
    EVE_C1_CASE(32, 128) EVE_C1_CASE(128, 32) EVE_C1_CASE(64, 128) EVE_C1_CASE(128, 64) EVE_C1_CASE(16, 16) EVE_C1_CASE(32, 32)
    EVE_C1_CASE(64, 64)
#undef EVE_C1_CASE
    return false;
}

}  // namespace eve
