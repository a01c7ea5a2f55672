// 3x3 / stride 1 / pad 1 convolution (forward and data gradient), halo-resident, on the 32x32x16 bf16 MFMA.
//
// Same data movement as conv3x3_halo_kernel (conv_fast.h): per 32-channel slice the (TH+2) x (W+2) halo of the tile is
// fetched ONCE by LDS-DMA and all nine taps read it at lane-constant offsets; the weight tile of every (slice, tap)
// step runs ahead in a 4-slot ring; counted s_waitcnt vmcnt(N) + one s_barrier per step.  The fragment reads of step i+1
// are issued before the MFMAs of step i (register double buffer).
//
// What changes is the matrix instruction.  Measured on MI355X (tools/probes/mfma_rate.hip, random data, the register
// arrangement of this kernel: 64 accumulator registers per wave, two waves per SIMD, ~2 VALU per 16x16x32-equivalent):
//      v_mfma_f32_16x16x32_bf16   1 183 TFLOP/s  (1 657 with no VALU beside it, 1 195 with one wave per SIMD)
//      v_mfma_f32_32x32x16_bf16   1 593 TFLOP/s  (1 812 / 1 798)
// i.e. the 16x16 shape loses a third of the matrix pipe as soon as anything else issues next to it, and the first
// kernel sat exactly on that 1.18 PFLOP/s plateau.  A wave's 64 x 64 tile is now 2 x 2 tiles of 32 x 32; a K step of 32
// channels is two K = 16 halves.
//
// Fragment geometry (A = weights: 32 output channels x 16 k; B = pixels: 16 k x 32 pixels):
//   lane l reads, for K half kh, the 16-byte chunk 2*kh + (l >> 5) of row (l & 31) of its tile, for both operands;
//   D[i][j] (i = output channel, j = pixel): lane holds pixel j = l & 31 and channels i = 8*b + 4*(l >> 5) + r in
//   accumulator register 4*b + r.
// LDS rows are 64 bytes (one halo pixel / output channel x 32 channels) with chunk' = chunk ^ g: g = (hx >> 2) & 3 for
// W >= 32, (hx >> 1) & 3 for W = 8 / 16, a 6-entry table of the halo row for W = 4 (TH = 4), and (row >> 2) & 3 for the
// weight rows -- every ds_read_b128 of every tap is bank-conflict free (brute-forced: tools/lds_banks32.py).
// The weight rows are permuted in LDS so that a lane's 2 x 16 accumulator rows are 32 CONSECUTIVE output channels of
// its pixel: four 16-byte stores per pixel.
#pragma once
#include <type_traits>
#include "common.h"
#include "conv_fast.h"
#include "lds_dma.h"

namespace eve {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// chunk swizzle of halo pixel (halo row hr counted over the whole tile, halo column hx); hy = halo row inside its image
__device__ __forceinline__ int halo32_key(int W, int hy, int hx) {
    if (W >= 32) return (hx >> 2) & 3;
    if (W >= 8) return (hx >> 1) & 3;
    return (0x787 >> (2 * hy)) & 3;                  // W = 4, TH = 4: {3, 1, 0, 2, 3, 1}[hy]
}
// output channel (relative to the wave's 64) that LDS weight row (nt, i) of a wave holds
__device__ __forceinline__ int halo32_row_channel(int row64) {
    const int nt = row64 >> 5, i = row64 & 31;
    return 32 * ((i >> 2) & 1) + 16 * nt + 4 * (i >> 3) + (i & 3);
}

template <int WM, int WN>
__global__ __launch_bounds__(256) void conv3x3_halo32_kernel(const HaloParams p, const bf16_t* __restrict__ x,
                                                             const bf16_t* __restrict__ w,
                                                             const float* __restrict__ bias, const int epi_act,
                                                             bf16_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W2 = p.W + 2, HPI = (p.TH + 2) * W2, HP = p.TI * HPI;
    const int a_stage = p.a_pieces * 4096;                    // bytes per halo stage (256 slots x 16 B per piece)
    char* const sA = smem;                                    // 2 halo stages
    constexpr int BSLOT = 4096 * WN;                          // weight tile: 64*WN output channels x 64 B
    char* const sB = smem + 2 * a_stage;                      // 4 weight slots

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t tm = lid / p.tiles_n, tn = lid % p.tiles_n;
    const uint32_t n0 = (p.TI == 1 ? tm / p.bands : tm * p.TI);
    const int y0 = p.TI == 1 ? (int)(tm % p.bands) * p.TH : 0;
    const uint32_t co0 = tn * (64 * WN);

    const eve_int4 rs_x = make_rsrc_words(x, p.x_bytes);
    const eve_int4 rs_w = make_rsrc_words(w, p.w_bytes);
    const uint32_t ldsA = lds_addr_of(sA), ldsB = lds_addr_of(sB);

    // ---- halo DMA slots owned by this thread (loop invariant): global byte offset without the channel slice ----
    int a_goff[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int L = tid + 256 * j;                          // physical 16-byte slot in the stage
        const int hp = L >> 2, pc = L & 3;
        int off = EVE_OOB;
        if (j < p.a_pieces && hp < HP) {
            const int ti = (int)fd_div((uint32_t)hp, p.fd_hpi);
            const int r = hp - ti * HPI;
            const int hy = (int)fd_div((uint32_t)r, p.fd_w2), hx = r - hy * W2;
            const int gy = y0 - 1 + hy, gx = hx - 1;
            const uint32_t n = n0 + ti;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W && n < (uint32_t)p.N)
                off = (int)((((n * p.H + gy) * p.W + gx) * p.Cin) * 2) + ((pc ^ halo32_key(p.W, hy, hx)) << 4);
        }
        a_goff[j] = off;
    }
    // ---- weight DMA slots: 64*WN rows (output channels) x 64 B ----
    int b_goff[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int L = tid + 256 * j;
        const int cl = L >> 2, pc = L & 3;
        const uint32_t co = co0 + (cl & ~63) + halo32_row_channel(cl & 63);
        b_goff[j] = co < (uint32_t)p.Cout ? (int)(co * (uint32_t)p.K) * 2 + ((pc ^ ((cl >> 2) & 3)) << 4) : EVE_OOB;
    }

    const int nslices = p.Cin / 32;
    const int wave_off = wave * 1024;
    // weight tile of (slice sb, tap tb) into ring slot `slot`; zero-fill past the last slice
    auto issue_b = [&](int sb, int tb, int slot) {
        const int koff = (tb * p.Cin + sb * 32) * 2;
        const uint32_t dst = ldsB + slot * BSLOT + wave_off;
        const bool live = sb < nslices;
#pragma unroll
        for (int j = 0; j < WN; ++j)
            lds_dma16_asm(rs_w, dst + j * 4096, (live && b_goff[j] != EVE_OOB) ? b_goff[j] + koff : EVE_OOB);
    };

    // ---- fragment coordinates: every (tap, pixel tile, K half) LDS offset is a lane constant ----
    const int lane = tid & 63, wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    int aaddr[9][2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = wm * 64 + mt * 32 + li;                 // pixel in the tile: (ti, ty, tx)
        const int rowi = (int)fd_div((uint32_t)m, p.fd_w), tx = m - rowi * p.W;
        const int ti = (int)fd_div((uint32_t)rowi, p.fd_th), ty = rowi - ti * p.TH;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t % 3;
            const int dy = p.flip ? 2 - kh : kh, dx = p.flip ? 2 - kw : kw;
            const int hy = ty + dy, hx = tx + dx;
            const int base = ((ti * (p.TH + 2) + hy) * W2 + hx) << 6;
            const int key = halo32_key(p.W, hy, hx);
            aaddr[t][mt][0] = base + ((lh ^ key) << 4);
            aaddr[t][mt][1] = base + (((2 + lh) ^ key) << 4);
        }
    }
    int brow[2][2];                                           // byte address of the weight fragment inside a ring slot
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int c = wn * 64 + nt * 32 + li;
        const int key = (c >> 2) & 3;
        brow[nt][0] = (c << 6) + ((lh ^ key) << 4);
        brow[nt][1] = (c << 6) + (((2 + lh) ^ key) << 4);
    }

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // ---- synchronisation (as in conv3x3_halo_kernel): loads return in order; before step i+1 only the DMAs issued in
    // steps i-1 and i may still be in flight: N = 2*WN + halo pieces issued in those two steps ----
    auto wait_all_but = [&](int extra) {                      // extra (uniform) = halo pieces among them: 0, 1 or 2
        if (WN == 2) {
            if (extra == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (extra == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            if (extra == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (extra == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
    };
    // ---- software pipeline: the fragments of step i+1 are read from LDS while the MFMAs of step i run (register double
    // buffer), so a wave's matrix instructions issue back to back instead of waiting for its own ds_reads every step
    // (SQ counters of the unpipelined loop: MFMA pipe 48 % busy, waves parked 35-40 % of their cycles).
    // Ring discipline: at the top of step i (after the barrier) the weight tile of step i+1 has landed and slot i & 3 --
    // read by every wave during step i-1 -- is free, so tile i+4 goes there: the tiles of steps i+2, i+3 stay in flight
    // (the same two steps of look-ahead as before).  Prologue: halo of slice 0, weight tiles 0..3. ----
#pragma unroll
    for (int j = 0; j < 7; ++j)
        if (j < p.a_pieces) lds_dma16_asm(rs_x, ldsA + j * 4096 + wave_off, a_goff[j]);
    issue_b(0, 0, 0);
    issue_b(0, 1, 1);
    issue_b(0, 2, 2);
    issue_b(0, 3, 3);
    if (WN == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // halo of slice 0 and weight tile 0 have landed
    else         asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    bf16x8_t fx[2][2][2], fw[2][2][2];                        // [buffer][tile][K half]
    auto load_frags = [&](int buf, const char* la, const char* lb, int t) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                fx[buf][mt][kh] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(la + aaddr[t][mt][kh]));
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                fw[buf][nt][kh] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(lb + brow[nt][kh]));
    };
    load_frags(0, sA, sB, 0);

    for (int s = 0; s < nslices; ++s) {
        const char* la = sA + (s & 1) * a_stage;
        const char* la_next = sA + ((s + 1) & 1) * a_stage;
        const uint32_t na = ldsA + ((s + 1) & 1) * a_stage + wave_off;   // next slice's halo stage
        const int ap = s + 1 < nslices ? p.a_pieces : 0;      // halo pieces this slice still has to fetch
        const int nxt_c = (s + 1) * 64;                       // its channel byte offset
        const bool last_slice = s + 1 == nslices;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int bufc = t & 1, bufn = 1 - bufc;          // compile-time after unrolling (step 8 hands its prefetch back to buffer 0)
            // DMAs that may still be in flight: those issued in steps i-2 and i-1 (2*WN weight pieces + their halo pieces)
            const int extra = (t >= 2 && t - 2 < ap ? 1 : 0) + (t >= 1 && t - 1 < ap ? 1 : 0);
            if (t == 8 && ap == 7) {                          // the halo of the next slice is read in this step: all of it
                if (WN == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else         asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            } else {
                wait_all_but(extra);                          // the weight tile of step i+1 has landed
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // this wave's reads of slot i & 3 are done
            __builtin_amdgcn_s_barrier();
            // tile i+4 into the slot step i-1 read; the halo piece of the next slice
            issue_b(s + (t + 4) / 9, (t + 4) % 9, (s + t + 4) & 3);
            if (t < 7 && t < ap)
                lds_dma16_asm(rs_x, na + t * 4096, a_goff[t < 7 ? t : 0] != EVE_OOB ? a_goff[t < 7 ? t : 0] + nxt_c : EVE_OOB);
            // fragments of step i+1 (the zero-filled tile past the last step is read and never used)
            if (t < 8) load_frags(bufn, la, sB + ((s + t + 1) & 3) * BSLOT, t + 1);
            else if (!last_slice) load_frags(bufn, la_next, sB + ((s + 9) & 3) * BSLOT, 0);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[bufc][nt][kh], fx[bufc][mt][kh], acc[mt][nt], 0, 0, 0);
            if (t == 8) {                                     // nine steps per slice: put the prefetch where step 0 expects it
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) { fx[0][a][b] = fx[1][a][b]; fw[0][a][b] = fw[1][a][b]; }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the zero-fill tiles past the end, before LDS is released

    // ---- epilogue (two bodies, see act_fwd4): the lane owns channels co .. co+31 of two pixels ----
    auto epilogue = [&](auto fast) {
        const uint32_t co = co0 + wn * 64 + lh * 32;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = wm * 64 + mt * 32 + li;
            const int rowi = (int)fd_div((uint32_t)m, p.fd_w), tx = m - rowi * p.W;
            const int ti = (int)fd_div((uint32_t)rowi, p.fd_th), ty = rowi - ti * p.TH;
            const uint32_t n = n0 + ti;
            const int y = y0 + ty;
            if (n >= (uint32_t)p.N || y >= p.H) continue;
            bf16_t* dst = out + ((size_t)(n * p.H + y) * p.W + tx) * p.Cout + co;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                     // four 16-byte groups of 8 channels: co + 8q ..
                if (co + 8 * q + 8 > (uint32_t)p.Cout) continue;       // Cout is a multiple of 8
                float o[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    // channel 8q + c = 16*nt + 4*b + r  ->  accumulator register 4*b + r of tile nt
                    o[c] = acc[mt][q >> 1][4 * (2 * (q & 1) + (c >> 2)) + (c & 3)];
                    if (bias) o[c] += bias[co + 8 * q + c];
                }
                act_fwd4<decltype(fast)::value>(o, epi_act);
                act_fwd4<decltype(fast)::value>(o + 4, epi_act);
                if (epi_act & EVE_EPI_ACC) {
                    float old[8];
                    Elem<bf16_t>::unpack(*reinterpret_cast<const uint4*>(dst + 8 * q), old);
#pragma unroll
                    for (int c = 0; c < 8; ++c) o[c] += old[c];
                }
                *reinterpret_cast<uint4*>(dst + 8 * q) = Elem<bf16_t>::pack(o);
            }
        }
    };
    if (act_is_fast(epi_act)) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

}  // namespace eve
