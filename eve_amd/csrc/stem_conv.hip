// ResNet stem: 7x7 / stride 2 / pad 3 convolution, 3 -> 64 channels, bf16 (torchvision ResNet.conv1, reference
// src/models/eye_net.py:48-50,106).  K = 147 is too small and too ragged for the tiled implicit-GEMM kernels
// (they pad Cin 3 -> 8 and decode a tap per 16-byte vector: 1.2 ms per step, 5 % of the MFMA peak).
//
// Formulation: the input is repacked once to [N][IH+6][IW+8][4] bf16 (3 channels + 1 zero, zero borders of
// 3 rows / 4 columns), i.e. 8 bytes per pixel with the padding materialised.  One filter ROW is then a single
// MFMA K step: 8 consecutive pixels x 4 channels = 32 K values (7 real taps + 1 zero-weight tap), and a lane's
// fragment (2 pixels x 4 channels = 16 bytes) is ONE unpredicated global load -- no LDS staging for the
// activations at all.  The 28 KB filter bank sits in LDS for the lifetime of the (persistent) workgroup.
// Waves are independent after the filter load: no barriers, each wave walks its own 64-pixel row tiles and keeps
// all 28 fragment loads of a tile in flight.  Output 64 px x 64 channels per wave tile, 112 MFMAs.
#include <stdlib.h>
#include "common.h"

namespace eve {

// dst[n][y+3][x+4][c] = c < C ? src[n][c][y][x] : 0, borders zero (dst is fully written)
template <typename H>
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ src, uint2* __restrict__ dst, int C,
                                                        int IH, int IW, long long items) {
    const int IHp = IH + 6, IWp = IW + 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int xp = (int)(i % IWp);
        long long t = i / IWp;
        const int yp = (int)(t % IHp);
        const long long n = t / IHp;
        const int x = xp - 4, y = yp - 3;
        uint2 q = make_uint2(0u, 0u);
        if (x >= 0 && x < IW && y >= 0 && y < IH) {
            float f[4] = {0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < C && c < 4; ++c) f[c] = src[((n * C + c) * IH + y) * IW + x];
            q.x = Elem<H>::pack2(f[0], f[1]);
            q.y = Elem<H>::pack2(f[2], f[3]);
        }
        dst[i] = q;
    }
}

// the same, four pixels of a row per thread (IW % 4 == 0: a group of four is inside the image or in the border as a
// whole): 16-byte loads from each channel plane, 32 bytes stored -- the one-pixel version moved 20 bytes per thread
// and ran at half the rate of its HBM traffic
template <typename H>
__global__ __launch_bounds__(256) void stem_pack4_kernel(const float* __restrict__ src, uint4* __restrict__ dst, int C,
                                                         int IH, int IW, long long groups) {
    const int IHp = IH + 6, GW = (IW + 8) / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < groups; i += (long long)gridDim.x * 256) {
        const int gx = (int)(i % GW);
        long long t = i / GW;
        const int yp = (int)(t % IHp);
        const long long n = t / IHp;
        const int x = gx * 4 - 4, y = yp - 3;
        uint4 lo = make_uint4(0u, 0u, 0u, 0u), hi = lo;
        if (x >= 0 && x < IW && y >= 0 && y < IH) {
            float4 f[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                f[c] = c < C ? *reinterpret_cast<const float4*>(src + ((n * C + c) * IH + y) * IW + x) : make_float4(0.f, 0.f, 0.f, 0.f);
            lo = make_uint4(Elem<H>::pack2(f[0].x, f[1].x), Elem<H>::pack2(f[2].x, f[3].x), Elem<H>::pack2(f[0].y, f[1].y), Elem<H>::pack2(f[2].y, f[3].y));
            hi = make_uint4(Elem<H>::pack2(f[0].z, f[1].z), Elem<H>::pack2(f[2].z, f[3].z), Elem<H>::pack2(f[0].w, f[1].w), Elem<H>::pack2(f[2].w, f[3].w));
        }
        dst[2 * i] = lo;
        dst[2 * i + 1] = hi;
    }
}

// MT: 16-pixel tiles per wave tile (2: a wave tile is two output rows x 32 pixels, see the tile loop).  Round 5 measured the
// kernel latency-bound -- a wave loads a tile's fragments, multiplies, stores, nothing overlaps inside a wave -- so what counts is
// the loads a SIMD keeps in flight and how few it needs: 64-pixel one-row tiles at two waves per SIMD 3.36 ms (configs[4]), 32-pixel
// one-row tiles at three waves 2.57 ms, two-row tiles (nine input rows for two outputs instead of seven for one) at two waves: see
// profiles/r05_notes.md.
template <typename H, int MT>
__global__ __launch_bounds__(256, 2) void stem7x7_kernel(const int N, const int IH, const int IW,
                                                      const uint2* __restrict__ xp, const H* __restrict__ w8,
                                                      H* __restrict__ y, const uint32_t ntiles) {
    __shared__ uint4 sW[7 * 256];                              // 7 filter rows x (64 output channels x 64 B)
    const int tid = threadIdx.x;
    // ---- filter bank: w8 is [64][7][7][8] (Cin padded to 8); LDS row = 2 output channels x 64 B, slot ^= row & 7 ----
    for (int e = tid; e < 64 * 7 * 8; e += 256) {
        const int kw = e & 7, kh = (e >> 3) % 7, co = e / 56;
        uint2 v = make_uint2(0u, 0u);
        if (kw < 7) v = *reinterpret_cast<const uint2*>(w8 + ((co * 7 + kh) * 7 + kw) * 8);   // channels 0..3
        // LDS channel row c_lds = nt * 16 + 4 g + r holds output channel g * 16 + nt * 4 + r (as stem_fused.hip does): an MFMA lane
        // then owns the 16 CONSECUTIVE channels 16 lg .. 16 lg + 15 of its pixel -- two 16-byte stores per pixel instead of four
        // 8-byte ones into four different 32-byte sectors (round 5: the 4 GB output of configs[4] went out in 8-byte pieces)
        const int c_lds = ((co >> 2) & 3) * 16 + (co >> 4) * 4 + (co & 3);
        const int row = c_lds >> 1;
        const int slot = (((c_lds & 1) << 2) + (kw >> 1)) ^ (row & 7);
        reinterpret_cast<uint2*>(sW)[(kh * 256 + row * 8 + slot) * 2 + (kw & 1)] = v;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int OH = IH / 2, OW = IW / 2, IWp = IW + 8, IHp = IH + 6;
    const int xblocks = OW / (16 * MT);
    int brow[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int c = nt * 16 + li;
        const int row = c >> 1;
        brow[nt] = row * 8 + ((((c & 1) << 2) + lg) ^ (row & 7));
    }
    // A wave tile is TWO output rows x 16 MT pixels (round 5): output rows 2q and 2q + 1 read input rows 4q .. 4q + 8, nine rows
    // for two outputs where one row per tile fetched seven for one -- 36 % fewer fragment loads on a kernel whose time is load
    // latency; filter row kh multiplies fx[kh] for the upper output row and fx[kh + 2] for the lower one.
    const uint32_t gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const uint32_t OH2 = (uint32_t)OH / 2;
    for (uint32_t tile = gw; tile < ntiles; tile += nw) {
        const uint32_t xb = tile % xblocks;
        const uint32_t r = tile / xblocks;
        const uint32_t oy = 2 * (r % OH2), n = r / OH2;
        // padded pixel index of the lane's first fragment pixel for filter row 0: (2*oy, 2*ox + 1 + 2*lg)
        const uint2* base = xp + ((size_t)n * IHp + 2 * oy) * IWp + 2 * (xb * (16 * MT) + li) + 1 + 2 * lg;
        uint4 fx[9][MT];
#pragma unroll
        for (int kh = 0; kh < 9; ++kh)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const uint2* p = base + (size_t)kh * IWp + mt * 32;     // 16 output pixels = 32 input pixels further
                const uint2 a = p[0], b = p[1];                          // 8-byte aligned pair (odd pixel index)
                fx[kh][mt] = make_uint4(a.x, a.y, b.x, b.y);
            }
        f32x4_t acc[2][MT][4];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[q][a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            uint4 fw[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) fw[nt] = sW[kh * 256 + brow[nt]];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) Elem<H>::mfma(acc[q][mt][nt], fw[nt], fx[kh + 2 * q][mt]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            H* orow = y + (((size_t)n * OH + oy + q) * OW + xb * (16 * MT)) * 64;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                uint32_t pk[8];                                 // channels 16 lg + 4 nt + r of pixel 16 mt + li
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    pk[2 * nt] = Elem<H>::pack2(acc[q][mt][nt][0], acc[q][mt][nt][1]);
                    pk[2 * nt + 1] = Elem<H>::pack2(acc[q][mt][nt][2], acc[q][mt][nt][3]);
                }
                H* dst = orow + (mt * 16 + li) * 64 + lg * 16;
                *reinterpret_cast<uint4*>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                *reinterpret_cast<uint4*>(dst + 8) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            }
        }
    }
}

}  // namespace eve

using namespace eve;

extern "C" int eve_stem_pack_input(int dtype, int N, int C, int IH, int IW, const float* src_nchw, void* dst, eve_stream_t stream) {
    if ((dtype != EVE_DT_BF16 && dtype != EVE_DT_F16) || N <= 0 || C <= 0 || C > 4 || IH <= 0 || IW <= 0 || !src_nchw || !dst) return set_error_msg("stem_pack_input: bad arguments");
    const long long items = (long long)N * (IH + 6) * (IW + 8);
    long long blocks = (items + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (IW % 4 == 0 && C <= 4 && (((uintptr_t)src_nchw | (uintptr_t)dst) & 15) == 0) {
        const long long groups = items / 4;
        long long gb = (groups + 255) / 256;
        if (gb > 8192) gb = 8192;
        EVE_DISPATCH_H16(dtype, hipLaunchKernelGGL(stem_pack4_kernel<H>, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, src_nchw, (uint4*)dst, C, IH, IW, groups));
    } else {
        EVE_DISPATCH_H16(dtype, hipLaunchKernelGGL(stem_pack_kernel<H>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src_nchw, (uint2*)dst, C, IH, IW, items));
    }
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_stem7x7s2_fwd(int dtype, int N, int IH, int IW, const void* x_padded, const void* w_ohwi8, void* y,
                                 eve_stream_t stream) {
    if ((dtype != EVE_DT_BF16 && dtype != EVE_DT_F16) || N <= 0 || IH <= 0 || IW <= 0 || (IH & 3) || (IW % 128) || !x_padded || !w_ohwi8 || !y)
        return set_error_msg("stem7x7s2_fwd: needs IH a multiple of 4 and IW a multiple of 128");
    const unsigned long long tiles = (unsigned long long)N * (IH / 4) * (IW / 64);                 // two output rows x 32 pixels per wave tile
    if (tiles >= (1ull << 32)) return set_error_msg("stem7x7s2_fwd: too many tiles");
    unsigned blocks = 512;                                                                         // two workgroups per CU
    if ((tiles + 3) / 4 < blocks) blocks = (unsigned)((tiles + 3) / 4);
    EVE_DISPATCH_H16(dtype, EVE_LAUNCH(EVE_HNAME(H, "stem7x7_kernel<", ", 2>"), (stem7x7_kernel<H, 2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, N, IH, IW,
                                       (const uint2*)x_padded, (const H*)w_ohwi8, (H*)y, (uint32_t)tiles));
    EVE_CHECK_LAUNCH();
    return 0;
}

// -------------------------------------------------------------------------------------------------
// Decoded video frames on the device.  The reference normalises uint8 N x H x W x C frames on the
// host -- astype(float32); *= 2/255; -= 1 for eye patches, *= 1/255 for screen frames, transposed to N x C x H x W
// (/root/reference/src/datasources/eve_sequences.py:196-211) -- and ships the float tensors over PCIe
// (src/core/training.py:257-261): 4x the bytes of the frames themselves.  Same arithmetic here (one rounded multiply,
// one rounded add: no fused multiply-add, so the float values are bit-identical to numpy's), from the uint8 frames:
//   frames_u8_to_nchw_kernel   the reference's float N x C x H x W tensor
//   frames_u8_to_stem_kernel   straight to the stem kernels' packed bf16 [N][IH+6][IW+8][4] input (eye patches)
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float normalise_u8(uint8_t v, float scale, float shift, bool has_shift) {
#pragma clang fp contract(off)          // hipcc contracts a * b + c into one fma by default: 255 * fl(2/255) - 1 would be 1 + 2^-23
    const float f = (float)v * scale;
    return has_shift ? f + shift : f;
}

__global__ __launch_bounds__(256) void frames_u8_to_nchw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int C, int H,
                                                                int W, float scale, float shift, bool has_shift, long long items) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W);
        long long t = i / W;
        const int y = (int)(t % H); t /= H;
        const int c = (int)(t % C);
        const long long n = t / C;
        dst[i] = normalise_u8(src[((n * H + y) * W + x) * C + c], scale, shift, has_shift);
    }
}

template <typename H>
__global__ __launch_bounds__(256) void frames_u8_to_stem_kernel(const uint8_t* __restrict__ src, uint2* __restrict__ dst, int C, int IH,
                                                                int IW, float scale, float shift, long long items) {
    const int IHp = IH + 6, IWp = IW + 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int xp = (int)(i % IWp);
        long long t = i / IWp;
        const int yp = (int)(t % IHp);
        const long long n = t / IHp;
        const int x = xp - 4, y = yp - 3;
        uint2 q = make_uint2(0u, 0u);
        if (x >= 0 && x < IW && y >= 0 && y < IH) {
            float f[4] = {0.f, 0.f, 0.f, 0.f};
            const uint8_t* p = src + ((n * IH + y) * IW + x) * C;
            for (int c = 0; c < C && c < 4; ++c) f[c] = normalise_u8(p[c], scale, shift, true);
            q.x = Elem<H>::pack2(f[0], f[1]);
            q.y = Elem<H>::pack2(f[2], f[3]);
        }
        dst[i] = q;
    }
}

extern "C" int eve_frames_u8_to_nchw(long long N, int H, int W, int C, const uint8_t* src_nhwc, float scale, float shift,
                                     int has_shift, float* dst_nchw, eve_stream_t stream) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || !src_nhwc || !dst_nchw) return set_error_msg("frames_u8_to_nchw: bad arguments");
    const long long items = N * C * H * W;
    long long blocks = (items + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(frames_u8_to_nchw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src_nhwc, dst_nchw, C, H, W,
                       scale, shift, has_shift != 0, items);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_frames_u8_to_stem(int dtype, long long N, int C, int IH, int IW, const uint8_t* src_nhwc, float scale, float shift,
                                     void* x_padded, eve_stream_t stream) {
    if ((dtype != EVE_DT_BF16 && dtype != EVE_DT_F16) || N <= 0 || C <= 0 || C > 4 || IH <= 0 || IW <= 0 || !src_nhwc || !x_padded) return set_error_msg("frames_u8_to_stem: bad arguments");
    const long long items = N * (IH + 6) * (IW + 8);
    long long blocks = (items + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    EVE_DISPATCH_H16(dtype, hipLaunchKernelGGL(frames_u8_to_stem_kernel<H>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src_nhwc,
                                               (uint2*)x_padded, C, IH, IW, scale, shift, items));
    EVE_CHECK_LAUNCH();
    return 0;
}
