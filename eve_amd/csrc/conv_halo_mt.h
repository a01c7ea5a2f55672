// 3x3 / stride 1 / pad 1 convolution (forward and data gradient), halo-resident, MACRO TILE: 256 pixels x (64 * NTW) output
// channels per workgroup of four waves -- one wave per SIMD, 128 x (32 * NTW) per wave = 4 x NTW accumulator tiles of the
// 32x32x16 bf16 MFMA (256 accumulator registers at NTW = 4).
//
// OPT-IN (EVE_HALO_MT=1), NOT the default: measured slower than the 128 x 128 kernels (layer 2 / 3 / 4 forward 0.177 / 0.194 /
// 0.178 ms against 0.140 / 0.131 / 0.124).  The bare loop of this tile shape sustains 1.65-1.86 PFLOP/s in isolation
// (tools/probes/macro_tile.hip, LDS-DMA stream and counted waits included) -- but with ONE wave per SIMD everything the wave
// issues besides its MFMAs has to fit the ~5 issue slots a 32-cycle MFMA leaves, and this kernel's step carries 3 VALU + 1.5
// SALU + 0.75 LDS per MFMA at NTW = 2 (SQ counters in profiles/r02_conv_experiments.md: 36 % of the wave's cycles issuing,
// 26 % in waits, 38 % MFMA-limited); at NTW = 4 the budget would fit but hipcc spills (824 B scratch, whose loads also break
// the counted vmcnt waits).  Per K step of 32 channels a wave issues 8 + 2 * NTW ds_read_b128 for 8 * NTW MFMAs (half the
// LDS reads per FLOP of the 64 x 64 wave tile) and NTW weight DMAs (half the DMA issue per FLOP).
//
// Schedule of step i (one tap of one 32-channel slice), software-pipelined across the two K halves:
//     top     counted vmcnt wait: everything but the DMAs issued at the top of step i-1 has landed -> the weight tile of
//             step i+1 is in LDS;  lgkmcnt(0);  s_barrier
//             DMA: weight tile i+3 into the ring slot step i-1 read; up to two halo pieces of the next slice
//     A       read the K-half-1 fragments of step i        | MFMAs of K half 0 (fragments read during step i-1)
//     B       read the K-half-0 fragments of step i+1      | MFMAs of K half 1
// so no MFMA waits for an LDS read issued in its own phase.  Fragment geometry, swizzles and the epilogue permutation are
// those of conv_halo32.h (lane = pixel l & 31, K chunk 2*kh + (l >> 5); a lane's 4 * NTW accumulator rows are 16 * NTW
// consecutive output channels).  Tiles are walked persistently: a workgroup takes tiles lid, lid + G, ... and the first
// halo slice and weight tiles of the next tile are fetched during the last steps of the current one.
#pragma once
#include <type_traits>
#include "common.h"
#include "conv_fast.h"
#include "conv_halo32.h"
#include "lds_dma.h"

namespace eve {

constexpr int MT_MAXP = 10;                                   // halo DMA pieces per thread and slice (256 threads x 16 B each)

template <int NTW>
__global__ __launch_bounds__(256, 1) void conv3x3_halo_mt_kernel(const HaloParams p, const bf16_t* __restrict__ x,
                                                                 const bf16_t* __restrict__ w,
                                                                 const float* __restrict__ bias, const int epi_act,
                                                                 bf16_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W2 = p.W + 2, HPI = (p.TH + 2) * W2, HP = p.TI * HPI;
    const int a_stage = p.a_pieces * 4096;
    char* const sA = smem;                                    // 2 halo stages
    constexpr int BN = 64 * NTW;                              // output channels per workgroup
    constexpr int BSLOT = 64 * BN;                            // weight tile: BN rows x 64 B
    char* const sB = smem + 2 * a_stage;                      // 4 ring slots

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t G = gridDim.x, T = p.tiles_m * p.tiles_n;
    const uint32_t lid = xcd_remap(blockIdx.x, G);
    if (lid >= T) return;

    const eve_int4 rs_x = make_rsrc_words(x, p.x_bytes);
    const eve_int4 rs_w = make_rsrc_words(w, p.w_bytes);
    const uint32_t ldsA = lds_addr_of(sA), ldsB = lds_addr_of(sB);

    // ---- halo DMA slots (lane constants): offset relative to pixel (n0, y0, 0); meta = (ti << 8) | hy, -1 = never live ----
    int a_rel[MT_MAXP], a_meta[MT_MAXP];
#pragma unroll
    for (int j = 0; j < MT_MAXP; ++j) {
        const int L = tid + 256 * j;
        const int hp = L >> 2, pc = L & 3;
        a_rel[j] = 0; a_meta[j] = -1;
        if (j < p.a_pieces && hp < HP) {
            const int ti = (int)fd_div((uint32_t)hp, p.fd_hpi);
            const int r = hp - ti * HPI;
            const int hy = (int)fd_div((uint32_t)r, p.fd_w2), hx = r - hy * W2;
            if (hx >= 1 && hx <= p.W) {
                a_rel[j] = (((ti * p.H + hy - 1) * p.W + hx - 1) * p.Cin) * 2 + ((pc ^ halo32_key(p.W, hy, hx)) << 4);
                a_meta[j] = (ti << 8) | hy;
            }
        }
    }
    // ---- weight DMA slots: BN rows x 64 B; LDS row (wave column wn, tile nt, MFMA row i) holds the output channel that
    //      makes a lane's rows consecutive: wn * 32 NTW + 16 NTW * ((i >> 2) & 1) + 16 nt + 4 (i >> 3) + (i & 3) ----
    int b_rel[NTW], b_ch[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int L = tid + 256 * j;
        const int cl = L >> 2, pc = L & 3;
        const int wcol = cl / (32 * NTW), rr = cl % (32 * NTW), nt = rr >> 5, i = rr & 31;
        b_ch[j] = wcol * 32 * NTW + 16 * NTW * ((i >> 2) & 1) + 16 * nt + 4 * (i >> 3) + (i & 3);
        b_rel[j] = (b_ch[j] * p.K) * 2 + ((pc ^ ((cl >> 2) & 3)) << 4);
    }

    const int nslices = p.Cin / 32;
    const int wave_off = wave * 1024;
    const int lane = tid & 63, wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    // ---- fragment addresses: (tap, pixel tile, K half) ----
    // (K half 1 = the same address with bit 5 flipped: chunk (2 + lh) ^ key = (lh ^ key) ^ 2)
    int aaddr[9][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = wm * 128 + mt * 32 + li;
        const int rowi = (int)fd_div((uint32_t)m, p.fd_w), tx = m - rowi * p.W;
        const int ti = (int)fd_div((uint32_t)rowi, p.fd_th), ty = rowi - ti * p.TH;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t % 3;
            const int dy = p.flip ? 2 - kh : kh, dx = p.flip ? 2 - kw : kw;
            const int hy = ty + dy, hx = tx + dx;
            const int base = ((ti * (p.TH + 2) + hy) * W2 + hx) << 6;
            const int key = halo32_key(p.W, hy, hx);
            aaddr[t][mt] = base + ((lh ^ key) << 4);
        }
    }
    int brow[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int c = wn * 32 * NTW + nt * 32 + li;
        const int key = (c >> 2) & 3;
        brow[nt] = (c << 6) + ((lh ^ key) << 4);
    }

    auto tile_coords = [&](uint32_t t, uint32_t& n0, int& y0, uint32_t& co0) {
        const uint32_t tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
        if (p.TI == 1) { n0 = tm / p.bands; y0 = (int)(tm - n0 * p.bands) * p.TH; }
        else { n0 = tm * p.TI; y0 = 0; }
        co0 = tn * BN;
    };
    auto issue_a = [&](int j, uint32_t n0, int y0, int sl, int st) {
        const int meta = a_meta[j];
        const int hy = meta & 0xff, ti = meta >> 8;
        const bool ok = (meta >= 0) & ((uint32_t)(y0 + hy - 1) < (uint32_t)p.H) & (n0 + (uint32_t)ti < (uint32_t)p.N);
        const int base = (int)(((n0 * p.H + y0) * p.W) * p.Cin) * 2 + sl * 64;
        const int src = base + a_rel[j];
        lds_dma16_asm(rs_x, ldsA + st * a_stage + j * 4096 + wave_off, ok ? src : EVE_OOB);
    };
    auto issue_b = [&](uint32_t co0, int sl, int tb, int slot, bool live) {
        const int koff = (int)(co0 * (uint32_t)p.K) * 2 + (tb * p.Cin + sl * 32) * 2;
        const uint32_t dst = ldsB + slot * BSLOT + wave_off;
#pragma unroll
        for (int j = 0; j < NTW; ++j)
            lds_dma16_asm(rs_w, dst + j * 4096, (live && co0 + (uint32_t)b_ch[j] < (uint32_t)p.Cout) ? b_rel[j] + koff : EVE_OOB);
    };
    // leave at most the DMAs of ONE step in flight: NTW weight pieces + `halo` (0..2) halo pieces
    auto wait_keep = [&](int halo) {
        if (NTW == 4) {
            if (halo == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (halo == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            if (halo == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (halo == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
    };
    auto pieces_at = [&](int t, int ap) { const int left = ap - 2 * t; return t >= 5 || left <= 0 ? 0 : (left >= 2 ? 2 : 1); };

    f32x16_t acc[4][NTW];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    uint32_t n0, co0;
    int y0;
    tile_coords(lid, n0, y0, co0);
    // ---- prologue of the first tile: halo slice 0, weight tiles of steps 0, 1, 2 ----
#pragma unroll
    for (int j = 0; j < MT_MAXP; ++j)
        if (j < p.a_pieces) issue_a(j, n0, y0, 0, 0);
    issue_b(co0, 0, 0, 0, true);
    issue_b(co0, 0, 1, 1, true);
    issue_b(co0, 0, 2, 2, true);
    if (NTW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // halo + tile 0 landed (tiles 1, 2 may be in flight)
    else          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    bf16x8_t fx0[4], fw0[NTW], fx1[4], fw1[NTW];              // K half 0 / 1 fragments
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) fx0[mt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sA + aaddr[0][mt]));
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) fw0[nt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sB + brow[nt]));

    uint32_t gs = 0;                                          // slices consumed so far (halo stage parity, ring phase)
    int prev_halo = 0;                                        // halo pieces issued at the top of the previous step
    for (uint32_t tile = lid; tile < T; tile += G) {
        const uint32_t nxt = tile + G;
        const bool has_next = nxt < T;
        uint32_t n1 = 0, co1 = 0;
        int y1 = 0;
        if (has_next) tile_coords(nxt, n1, y1, co1);
        for (int s = 0; s < nslices; ++s, ++gs) {
            const char* la = sA + (gs & 1) * a_stage;
            const char* la_n = sA + ((gs + 1) & 1) * a_stage;
            const int nst = (int)((gs + 1) & 1);
            const bool last = s + 1 == nslices;
            const bool more = !last || has_next;              // a slice follows in the stream (this tile's next, or the next tile's first)
            const uint32_t an = last ? n1 : n0;
            const int ay = last ? y1 : y0, asl = last ? 0 : s + 1;
            const int ap = more ? p.a_pieces : 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                // ---- top of step i = (s, t): the only serial section -- everything else is issued between MFMAs ----
                wait_keep(prev_halo);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const int halo = pieces_at(t, ap);
                prev_halo = halo;
                // DMA k of this step: k < NTW = piece k of weight tile i+3 (this slice, the tile's next slice, or the next
                // tile's first slice) into the ring slot step i-1 read; then up to two halo pieces of the next slice
                const bool wrap = t + 3 >= 9;
                const bool to_next = wrap && last;
                const uint32_t wco = to_next ? co1 : co0;
                const int wkoff = (int)(wco * (uint32_t)p.K) * 2 + (((t + 3) % 9) * p.Cin + (wrap ? (last ? 0 : s + 1) : s) * 32) * 2;
                const uint32_t wdst = ldsB + (uint32_t)((gs + t + 3) & 3) * BSLOT + wave_off;
                const bool wlive = !to_next || has_next;
                auto dma = [&](int k) {
                    if (k < NTW) {
                        lds_dma16_asm(rs_w, wdst + k * 4096, (wlive && wco + (uint32_t)b_ch[k < NTW ? k : 0] < (uint32_t)p.Cout) ? b_rel[k < NTW ? k : 0] + wkoff : EVE_OOB);
                    } else if (t < 5 && k - NTW < halo) {
                        issue_a(2 * (t < 5 ? t : 0) + (k - NTW), an, ay, asl, nst);
                    }
                };
                // ---- phase A: K half 1 fragments of this step | MFMAs of K half 0, the step's DMAs between them ----
                const char* lb = sB + ((gs + t) & 3) * BSLOT;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) fx1[mt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(la + (aaddr[t][mt] ^ 32)));
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) fw1[nt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(lb + (brow[nt] ^ 32)));
                // (hipcc's scheduler otherwise SINKS these reads below the MFMAs of this phase, next to their first use in
                //  phase B, and every phase then waits out a full LDS latency: the order is pinned)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw0[nt], fx0[mt], acc[mt][nt], 0, 0, 0);
                        const int idx = nt * 4 + mt;              // 4 * NTW MFMAs, NTW + 2 DMA slots
                        if (NTW == 4 ? (idx & 1) == 1 && (idx >> 1) < NTW + 2 : idx < NTW + 2) dma(NTW == 4 ? idx >> 1 : idx);
                    }
                // ---- phase B: K half 0 fragments of step i+1 | MFMAs of K half 1 ----
                __builtin_amdgcn_sched_barrier(0);
                if (t < 8 || more) {
                    const char* la2 = t < 8 ? la : la_n;
                    const char* lb2 = sB + ((gs + t + 1) & 3) * BSLOT;
                    const int t2 = t < 8 ? t + 1 : 0;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) fx0[mt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(la2 + aaddr[t2][mt]));
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) fw0[nt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(lb2 + brow[nt]));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw1[nt], fx1[mt], acc[mt][nt], 0, 0, 0);
            }
        }
        // ---- epilogue of this tile: the lane owns 16 * NTW consecutive channels of four pixels.  Its stores count in vmcnt
        //      too: they are drained here, before the next tile's counted waits resume (the DMAs of the next tile's first
        //      steps were issued above and complete in the meantime) ----
        // (identity / ReLU only -- the launcher sends every other activation to the 128 x 128 kernel -- as one branch-free
        //  clamp: a per-element activation switch unrolled over 256 accumulators is 230 KB of code, far beyond the
        //  instruction cache, and cost ~17 us per tile)
        const uint32_t co = co0 + wn * 32 * NTW + lh * 16 * NTW;
        const float lo = (epi_act & 0xff) == EVE_ACT_RELU ? 0.f : -3.0e38f;
        const bool accumulate = (epi_act & EVE_EPI_ACC) != 0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = wm * 128 + mt * 32 + li;
            const int rowi = (int)fd_div((uint32_t)m, p.fd_w), tx = m - rowi * p.W;
            const int ti = (int)fd_div((uint32_t)rowi, p.fd_th), ty = rowi - ti * p.TH;
            const uint32_t n = n0 + ti;
            const int y = y0 + ty;
            const bool live = n < (uint32_t)p.N && y < p.H;
            bf16_t* dst = out + ((size_t)(n * p.H + y) * p.W + tx) * p.Cout + co;
#pragma unroll
            for (int q = 0; q < 2 * NTW; ++q) {
                float o[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    o[c] = acc[mt][q >> 1][4 * (2 * (q & 1) + (c >> 2)) + (c & 3)];
                    acc[mt][q >> 1][4 * (2 * (q & 1) + (c >> 2)) + (c & 3)] = 0.f;
                }
                const bool ok = live && co + 8 * q + 8 <= (uint32_t)p.Cout;     // Cout is a multiple of 8
                if (bias) {
                    const float4 b0 = ok ? *reinterpret_cast<const float4*>(bias + co + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 b1 = ok ? *reinterpret_cast<const float4*>(bias + co + 8 * q + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    o[0] += b0.x; o[1] += b0.y; o[2] += b0.z; o[3] += b0.w; o[4] += b1.x; o[5] += b1.y; o[6] += b1.z; o[7] += b1.w;
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) o[c] = fmaxf(o[c], lo);
                if (accumulate && ok) {
                    float old[8];
                    Elem<bf16_t>::unpack(*reinterpret_cast<const uint4*>(dst + 8 * q), old);
#pragma unroll
                    for (int c = 0; c < 8; ++c) o[c] += old[c];
                }
                if (ok) *reinterpret_cast<uint4*>(dst + 8 * q) = Elem<bf16_t>::pack(o);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        prev_halo = 0;                                        // (everything has landed: the next wait_keep is trivially true)
        n0 = n1; y0 = y1; co0 = co1;
    }
}

}  // namespace eve
