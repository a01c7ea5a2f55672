// 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels on 32 x 32 images (ResNet layer 1: torchvision BasicBlock convs
// built at /root/reference/src/models/eye_net.py:48-50; forward and data gradient, 8 launches per train step), with the
// WHOLE FILTER BANK resident in LDS and the image tiles streaming under it.
//
// Layer 1 is the one trunk stage whose convolutions sit at the roofline ridge (288 FLOP per byte: 145 GFLOP over 504 MB
// per launch), so its kernel has to saturate HBM and the matrix pipe at the same time.  conv3x3_halo_pkernel<4,1>
// (conv_fast.h) re-fetches a 4 KB weight tile per step through a ring with one workgroup barrier per 16 MFMAs and ends at
// 0.83 PFLOP/s / 2.9 TB/s -- under both roofs.  Here:
//   * the 9 x 64 x 64 filter bank (72 KB) is loaded into LDS once per workgroup (persistent, one per CU);
//   * a tile is half an image (16 rows x 32 columns, 512 pixels x all 64 output channels); per 32-input-channel slice its
//     18 x 34 halo (39 KB) is fetched once by LDS-DMA into one of two stages -- the two slices of a tile, and across tiles:
//     the stream of (tile, slice) positions ping-pongs between them, the fetch of position q+1 is issued at the top of
//     position q, a whole slice (9 taps = 2.3 us of MFMA work per SIMD) ahead of its first read;
//   * NO barrier and NO DMA inside a slice: a wave owns two image rows (64 pixels x 64 channels = 4 accumulator tiles of
//     32 x 32) and runs its nine taps at its own pace, fragment reads of tap t+1 issued before the 8 MFMAs of tap t
//     (register double buffer); the two waves of a SIMD fill each other's gaps.  One workgroup barrier per slice
//     publishes the next stage and frees the previous one;
//   * per tile a CU moves 78 KB in + 64 KB out for 75.5 MFLOP: 7.3 us at its share of 5 TB/s against 4.6 us of matrix
//     time -- HBM-bound by construction, which is where a 288 FLOP/B kernel belongs.
// LDS rows are 64 bytes (one halo pixel / output channel x 32 channels), chunk-swizzled as in conv_wg8.h (W = 32: key =
// (halo column >> 2) & 3; weight rows: (row >> 2) & 3); weight rows permuted so that a lane's 2 x 16 accumulator rows are
// 32 consecutive output channels of its pixel (four 16-byte stores).
#pragma once
#include <type_traits>
#include "common.h"
#include "lds_dma.h"
#include "conv_wg8.h"

#ifndef EVE_WS64_NB
#define EVE_WS64_NB 2
#endif

namespace eve {

struct Ws64Params {
    int N;                        // images [N][W][W][64] -> [N][W][W][64], W = 32 | 64
    int flip;                     // 0: forward taps, 1: data-gradient taps (filter position t reads weight tap 8 - t)
    uint32_t x_bytes, w_bytes;
};

#define EVE_WS64_MMA(OP)                                                                                                   \
    asm volatile("s_nop 1\n\t"                                                                                            \
                 OP " %0, %4, %6, %0\n\t"  OP " %1, %4, %7, %1\n\t" OP " %2, %5, %6, %2\n\t"  OP " %3, %5, %7, %3"               \
                 : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1])                                     \
                 : "v"(wf[0]), "v"(wf[1]), "v"(xf[0]), "v"(xf[1]))
// the tile's first step starts from zero (C = 0 operand): the accumulators are outputs only, their previous contents
// (the tile stored two positions earlier) are dead -- with "+a" the compiler carried them through 64 VGPRs
#define EVE_WS64_MMA0(OP)                                                                                                  \
    asm volatile("s_nop 1\n\t"                                                                                            \
                 OP " %0, %4, %6, 0\n\t"  OP " %1, %4, %7, 0\n\t" OP " %2, %5, %6, 0\n\t"  OP " %3, %5, %7, 0"                   \
                 : "=a"(acc[0][0]), "=a"(acc[0][1]), "=a"(acc[1][0]), "=a"(acc[1][1])                                     \
                 : "v"(wf[0]), "v"(wf[1]), "v"(xf[0]), "v"(xf[1]))
// 4 MFMAs of one tap and K half: acc[ct][pt] (+)= W[ct] x X[pt]
template <typename H, bool FIRST>
__device__ __forceinline__ void ws64_mma4(f32x16_t (&acc)[2][2], const u32x4_t (&wf)[2], const u32x4_t (&xf)[2]) {
    if constexpr (Elem<H>::IS_BF16) {
        if constexpr (FIRST) { EVE_WS64_MMA0("v_mfma_f32_32x32x16_bf16"); }
        else { EVE_WS64_MMA("v_mfma_f32_32x32x16_bf16"); }
    } else {
        if constexpr (FIRST) { EVE_WS64_MMA0("v_mfma_f32_32x32x16_f16"); }
        else { EVE_WS64_MMA("v_mfma_f32_32x32x16_f16"); }
    }
}
#undef EVE_WS64_MMA
#undef EVE_WS64_MMA0

// W = 64 (round 5: layer 1 on 256 x 256 patches, BASELINE configs[4]): the same stream over tiles of 8 rows x 64 columns -- a
// wave owns ONE image row (its two 32-pixel halves), eight row bands per image, a 10 x 66 halo = 42 pieces per slice: the sixth
// piece of waves 2..7 does not exist and is sent to a 1 KB dummy slot (every wave issues the same number of DMAs, which the
// counted waits rely on); 72 KB filters + 2 x 42 KB stages + bias + dummy = 157.25 of the 160 KB.
template <int W> struct Ws64Geom {
    static_assert(W == 32 || W == 64, "layer-1 planes of 128 x 128 and 256 x 256 patches");
    static constexpr int TH = 512 / W, W2 = W + 2, HP = (TH + 2) * W2;   // 612 | 660 halo pixels per tile and slice
    static constexpr int NP = (HP + 15) / 16;                            // 39 | 42 pieces of 16 pixels
    static constexpr int AP = (NP + 7) / 8;                              // 5 | 6 per wave
    static constexpr int ASTAGE = (W == 32 ? AP * 8 : NP) * 1024;        // 40 | 42 KB
    static constexpr int TPI = W * W / 512;                              // row bands (tiles) per image
    static constexpr int WBYTES = 2 * 9 * 4096;                          // [slice][filter position][64 rows x 64 B]
    static constexpr int BIAS_OFF = WBYTES + 2 * ASTAGE, DUMMY_OFF = BIAS_OFF + 256;
    static constexpr int LDS = DUMMY_OFF + (AP * 8 * 1024 > ASTAGE ? 1024 : 0);
};

template <typename H, int W>
__global__ __launch_bounds__(512, 2) void conv3x3_ws64_kernel(const Ws64Params p, const H* __restrict__ x,
                                                              const H* __restrict__ w, const float* __restrict__ bias,
                                                              const int epi_act, H* __restrict__ out) {
    using Geo = Ws64Geom<W>;
    constexpr int TH = Geo::TH, W2 = Geo::W2, HP = Geo::HP, NP = Geo::NP, AP = Geo::AP, ASTAGE = Geo::ASTAGE, TPI = Geo::TPI;
    constexpr int WBYTES = Geo::WBYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, l5 = lane >> 5;
    const uint32_t G = gridDim.x, T = (uint32_t)TPI * (uint32_t)p.N;
    const uint32_t lid = xcd_remap(blockIdx.x, G);
    const eve_int4 rs_x = make_rsrc_words(x, p.x_bytes);
    const eve_int4 rs_w = make_rsrc_words(w, p.w_bytes);
    const uint32_t ldsW = lds_addr_of(smem), ldsA = ldsW + WBYTES;
    if (lid >= T) return;

    // ---- the filter bank: 72 pieces of 16 rows x 64 B, 9 per wave ----
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int q = j * 8 + wave;                               // piece: (slice, position, 16-row quarter)
        const int s = q / 36, t = (q / 4) % 9, row = (q & 3) * 16 + (lane >> 2), pc = lane & 3;
        const int co = wg8_row_channel(row);
        const int tw = p.flip ? 8 - t : t;
        const int off = (co * 576 + tw * 64 + s * 32) * 2 + ((pc ^ ((row >> 2) & 3)) << 4);
        lds_dma16_asm(rs_w, ldsW + q * 1024, off);
    }
    // ---- halo slots (lane constants): byte offset relative to pixel (row y0, column 0) of the tile's image, validity per band ----
    int a_rel[AP];
    uint32_t a_ok = 0;                                            // bit 2j: valid in every band but the first, bit 2j+1: ... but the last
#pragma unroll
    for (int j = 0; j < AP; ++j) {
        const int hp = (j * 8 + wave) * 16 + (lane >> 2), pc = lane & 3;
        const int hy = hp / W2, hx = hp - hy * W2;
        a_rel[j] = ((hy - 1) * W + (hx - 1)) * 128 + ((pc ^ wg8_key<W>(hy, hx)) << 4);
        const bool col_ok = hp < HP && hx >= 1 && hx <= W;
        a_ok |= (uint32_t)(col_ok && hy >= 1) << (2 * j);         // first band: image row y0 - 1 + hy = hy - 1 >= 0
        a_ok |= (uint32_t)(col_ok && hy <= TH) << (2 * j + 1);    // last band: W - TH - 1 + hy <= W - 1
    }
    auto issue_halo = [&](uint32_t tile, int s, int stage) {      // this wave's AP pieces of (tile, slice s)
        const uint32_t n = tile / TPI, band = tile % TPI;
        const int base = (int)((n * (uint32_t)(W * W) + band * 512u) * 128u) + s * 64;
        const bool live = tile < T;
        const uint32_t need = (band == 0 ? 1u : 0u) | (band == TPI - 1 ? 2u : 0u);   // the validity bits this band asks for
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            const bool ok = live && (((a_ok >> (2 * j)) & need) == need) && (((a_ok >> (2 * j)) & 3u) != 0u);
            const uint32_t dst = (j * 8 + wave) * 1024 < ASTAGE ? ldsA + stage * ASTAGE + (j * 8 + wave) * 1024 : ldsW + Geo::DUMMY_OFF;
            lds_dma16_asm(rs_x, dst, ok ? base + a_rel[j] : EVE_OOB);
        }
    };

    // ---- fragment addresses (lane constants; stage, tap row and filter position are immediates) ----
    // (channel tile ct: +32 rows = +2048 B, same key; image row pt: +1 halo row = +2176 B, the key depends on the column only)
    uint32_t wrd[2][2];                                           // [slice][K half]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
            wrd[s][kh] = ldsW + s * 9 * 4096 + l31 * 64 + (((2 * kh + l5) ^ ((l31 >> 2) & 3)) << 4);
    uint32_t xrd[3][2];                                           // [tap column][K half], image row 0 of the wave, tap row 0
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int hy = (W == 32 ? 2 : 1) * wave, hx = l31 + dx;     // W = 64: the wave's two pixel tiles are the halves of one row
            xrd[dx][kh] = ldsA + (hy * W2 + hx) * 64 + (((2 * kh + l5) ^ wg8_key<W>(hy, hx)) << 4);
        }

    // Two accumulator sets: the tile at an even position of this workgroup's stream accumulates into set 0, the next into
    // set 1, and a finished set is converted and stored in four parts BETWEEN the first taps of the following tile (one
    // 32 x 32 accumulator tile -- 16 reads, bias, activation, 8 packs, two 16-byte stores -- after each of taps 1..4 of
    // slice 0): with every wave of the workgroup at the same barrier-delimited position, an epilogue outside the MFMA
    // stream idles the matrix pipe of the whole CU for its duration (measured: 0.165 ms against 0.13 with it hidden).
    f32x16_t acc[2][2][2];
    const uint32_t co = (uint32_t)l5 * 32;
    const float floor_v = (epi_act & 0xff) == EVE_ACT_RELU ? 0.f : __builtin_nanf("");   // max(o, floor): ReLU; max(o, NaN) = o
    // bias: 64 floats behind the halo stages, read back 16 at a time by the part that needs them (32 VGPRs otherwise)
    constexpr int BIAS_OFF = Geo::BIAS_OFF;
    if (tid < 64) *reinterpret_cast<float*>(smem + BIAS_OFF + tid * 4) = bias ? bias[tid] : 0.f;
    // one accumulator tile (channel tile ct, image row pt of the wave) of a finished tile: 16 channels of one pixel per lane
    // stores are raw-buffer stores: a part that has nothing to store (the first tile of the stream has no predecessor)
    // goes out of range and is dropped, no branch in the MFMA stream -- and every tile issues exactly 8 of them per wave,
    // which the counted wait at the top of slice 1 relies on
    const eve_int4 rs_o = make_rsrc_words(out, p.x_bytes);
    // A lane's accumulators are 32 consecutive channels of ONE pixel = four 16-byte chunks k = 0..3 (64 bytes): stored as they
    // are, an instruction writes 16 bytes into 64 different 128-byte lines (15.7 M write requests per launch; lane-contiguous
    // fake addresses measured 0.146 -> 0.136 ms).  So the four chunks of an image row are packed first (pk_row[k]), then
    // transposed 4 x 4 across each lane quad (two DPP butterfly stages): lane q of a quad ends up with chunk q of the quad's
    // pixels m = 0..3, and store m writes 64 contiguous bytes per quad -- 16 requests per instruction instead of 64.
    auto tile_base = [&](uint32_t tile) -> uint32_t {            // byte offset of (tile, this wave's first row, the lane quad's first pixel, lane's chunk)
        // (a tile is 512 consecutive pixels of its image in both geometries; the wave's are 64 w .. 64 w + 63, pixel tile pt the upper 32)
        return (((tile / TPI) * (uint32_t)(W * W) + (tile % TPI) * 512u + (uint32_t)wave * 64u + (uint32_t)(l31 & ~3)) * 64u + co) * 2u +
               (uint32_t)(l31 & 3) * 16u;
    };
    u32x4_t pk_row[4];
    auto pack_part = [&](const f32x16_t& a, int ct, int half) {   // chunk k = ct * 2 + half of the lane's pixel
        float bv[8];
#pragma unroll
        for (int c = 0; c < 8; c += 4) {
            const float4 b4 = *reinterpret_cast<const float4*>(smem + BIAS_OFF + (co + ct * 16 + half * 8 + c) * 4);
            bv[c] = b4.x; bv[c + 1] = b4.y; bv[c + 2] = b4.z; bv[c + 3] = b4.w;
        }
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            float o0 = a[half * 8 + r] + bv[r], o1 = a[half * 8 + r + 1] + bv[r + 1];
            o0 = fmaxf(o0, floor_v); o1 = fmaxf(o1, floor_v);
            pk_row[ct * 2 + half][r / 2] = Elem<H>::pack2(o0, o1);
        }
    };
    auto store_row = [&](uint32_t obase, bool live, int pt) {
        quad_transpose4x4(pk_row, lane);
        // pk_row[m] = chunk (l31 & 3) of the quad's pixel m
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const uint32_t voff = live ? obase + (uint32_t)(pt * 32 * 128 + m * 128) : (uint32_t)EVE_OOB;
            // (s_nop 1: a store of more than 8 bytes reads its data registers after issue -- the ISA wants wait states before
            //  a VALU write of them, and the compiler, which does not know this statement is a store, put `v_or v6, ...`
            //  right behind `buffer_store_dwordx4 v[6:9]`: dword 0 of a few lanes went out with the new value)
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" :: "v"(pk_row[m]), "v"(voff), "s"(rs_o) : "memory");
        }
    };
    auto run_tile = [&](auto parity, uint32_t tile, uint32_t prev_base, bool prev_live) {
        constexpr int PAR = decltype(parity)::value;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // the stage of (tile, s) -- and, the first time, the filter bank -- has landed for every wave; the other stage
            // is free (every wave is past its last read of it).  Slice 1: the previous tile's 8 stores were issued AFTER
            // this stage's fetch and complete after it (vector memory operations of a wave retire in order), so only they
            // may stay in flight -- waiting for their write acknowledgements cost 10 % of the kernel.
            if (s == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s == 0) issue_halo(tile, 1, 1);
            else issue_halo(tile + G, 0, 0);
            constexpr int NB = EVE_WS64_NB;                       // fragment buffers: step h + NB - 1 is read before step h runs
            u32x4_t wf[NB][2], xf[NB][2];
            auto read_step = [&](int buf, int h) {
                const int t = h >> 1, kh = h & 1, dy = t / 3, dx = t % 3;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    wf[buf][ct] = *reinterpret_cast<const EVE_LDS u32x4_t*>((uintptr_t)(wrd[s][kh] + t * 4096 + ct * 2048));
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
                    xf[buf][pt] = *reinterpret_cast<const EVE_LDS u32x4_t*>((uintptr_t)(xrd[dx][kh] + s * ASTAGE +
                                                                                         (W == 32 ? (dy + pt) * W2 * 64 : dy * W2 * 64 + pt * 2048)));
            };
#pragma unroll
            for (int h = 0; h < NB - 1; ++h) read_step(h, h);
#pragma unroll
            for (int h = 0; h < 18; ++h) {
                if (h + NB - 1 < 18) read_step((h + NB - 1) % NB, h + NB - 1);    // a later step's fragments first ...
                __builtin_amdgcn_sched_barrier(0);
                if (s == 0 && h == 0) ws64_mma4<H, true>(acc[PAR], wf[h % NB], xf[h % NB]);    // ... then this step's MFMAs
                else ws64_mma4<H, false>(acc[PAR], wf[h % NB], xf[h % NB]);
                __builtin_amdgcn_sched_barrier(0);
                if (s == 0 && h >= 2 && h < 10) {                 // half an accumulator tile of the previous tile per step
                    // (the empty statement redefines the tile here: without it the compiler copies all 64 accumulators
                    //  into VGPRs at the top of the tile and carries them through the steps)
                    if ((h & 1) == 0) asm volatile("" : "+a"(acc[PAR ^ 1][((h - 2) >> 1) & 1][(h - 2) >> 2]));
                    pack_part(acc[PAR ^ 1][((h - 2) >> 1) & 1][(h - 2) >> 2], ((h - 2) >> 1) & 1, h & 1);
                    if (((h - 2) & 3) == 3) store_row(prev_base, prev_live, (h - 2) >> 2);     // an image row is complete: 4 stores
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // The MFMAs are inline assembly: the compiler does not know their write latency, and whatever it does with this set
        // after the tile (at the stream's end it COPIES sets between registers for the shared tail code) must find the
        // results landed.
        asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[PAR][0][0]), "+a"(acc[PAR][0][1]), "+a"(acc[PAR][1][0]), "+a"(acc[PAR][1][1]) :: "memory");
    };
    auto store_tile = [&](f32x16_t (&a)[2][2], uint32_t tile) {   // the stream's last tile: nothing left to hide it under
        asm volatile("s_nop 15\n\ts_nop 15" : "+a"(a[0][0]), "+a"(a[0][1]), "+a"(a[1][0]), "+a"(a[1][1]) :: "memory");
        const uint32_t ob = tile_base(tile);
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
            for (int k = 0; k < 4; ++k) pack_part(a[k >> 1][pt], k >> 1, k & 1);
            store_row(ob, true, pt);
        }
    };

    // The stream in pairs: (even position -> set 0, storing set 1), (odd position -> set 1, storing set 0); one back edge,
    // each set in the same registers at the loop header (no copies of 64 accumulators at a merge).
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[1][ct][pt][r] = 0.f;
    issue_halo(lid, 0, 0);
    uint32_t tile = lid, prev_base = 0;
    bool prev_live = false;
    for (; tile + G < T; tile += 2 * G) {
        run_tile(std::integral_constant<int, 0>{}, tile, prev_base, prev_live);
        run_tile(std::integral_constant<int, 1>{}, tile + G, tile_base(tile), true);
        prev_base = tile_base(tile + G); prev_live = true;
    }
    if (tile < T) {
        run_tile(std::integral_constant<int, 0>{}, tile, prev_base, prev_live);
        store_tile(acc[0], tile);
    } else {
        store_tile(acc[1], tile - G);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the last (all out-of-range) halo fetch, before LDS is released
}

}  // namespace eve
