// 3x3 / stride 1 / pad 1 convolution (forward and data gradient) of the ResNet trunk's 128..512-channel layers
// (torchvision BasicBlock convs built at /root/reference/src/models/eye_net.py:48-50), third generation:
// ONE eight-wave workgroup per CU, 32x32x16 MFMA, the two waves of every SIMD in opposite phases.
//
// What the second-generation kernel (conv3x3_halo_kernel, conv_fast.h) left on the table (profiles/r02_conv_experiments.md):
// a four-wave workgroup per 128 x 128 tile reads 8 fragments per 16 MFMAs, synchronises once per 256 MFMA-cycles and leaves
// the pairing of the two co-resident waves of a SIMD to chance; its loop sat at 1.0-1.15 PFLOP/s whatever was tuned.
// Here a wave owns 128 pixels x 64 channels (8 accumulator tiles of 32 x 32: 12 fragment reads per 16 MFMAs = 512
// matrix-pipe cycles), and the workgroup's waves are split into two groups (waves 0-3 / 4-7 = one wave of each group on
// every SIMD) that run the SAME instruction stream ONE barrier apart:
//
//     phase 2g     group 0: read fragments of step g (+ issue its share of the DMAs)    group 1: 16 MFMAs of step g-1
//     phase 2g+1   group 0: 16 MFMAs of step g                                          group 1: read fragments of step g
//
// so on every SIMD one wave feeds the matrix pipe while the other one does LDS / DMA / address work, by construction
// (the role split of the 8-phase GEMM schedule of cdna_hip_programming.md, 5.5 T3-T5, carried over to the halo-resident
// convolution).  A step = one filter tap x one 32-channel slice (K = 32 = two K=16 MFMA halves).  The two groups do NOT
// need a barrier between their phases: between two workgroup barriers (ONE per step) group 0 runs [read g, multiply g]
// and group 1, at a higher static priority, [multiply g-1, read g]; the matrix pipe itself serialises the two multiplies
// (group 1's first), so each wave's LDS / DMA half lies under its partner's 16 MFMAs.  (The first version synchronised
// the whole workgroup after every phase: 740 cycles per 512-cycle phase; profiles/r03_conv_wg8.md.)
//
// Data movement is the halo design's: per 32-channel slice the (W+2) x (W+2) halo of each of the tile's TI whole images
// is fetched ONCE by LDS-DMA (two stages, the next slice streams in during taps 1..AP of the current one) and all nine
// taps read it at lane-constant addresses + immediate offsets; the weight tile of a step (64*WN channels x 64 B) runs TWO
// steps ahead in a 4-slot ring.  Every wave issues the same share (WN/2 weight pieces, <= 1 halo piece per step), the
// tap / slice part of a source address is the instruction's scalar offset, so a DMA costs no vector instruction at all.
//
// Synchronisation (interval g = between the barriers of steps g-1 and g; both groups read step g's operands in it):
//   * weight tile g+2 -> slot (g+2)&3 = slot of tile g-2, last read in interval g-2 (complete at the first lgkmcnt(0) of
//     interval g-1); issued in interval g, every wave waits for its pieces at the end of interval g+1
//     (s_waitcnt vmcnt(pieces issued in that interval): loads return in order), that barrier publishes them, first read
//     in interval g+2.
//   * halo pieces of slice s+1 go to the stage slice s-1 used (last read in interval 9s-1); issued in intervals 9s+1 ..
//     9s+AP (AP <= 7), waited for one interval later, first read in interval 9s+9.
//
// LDS rows are 64 bytes (one halo pixel / output channel x 32 channels) with the 16-byte chunk XOR-ed by a key of the
// halo column (W >= 8) or halo row (W = 4) / of the weight row: every ds_read_b128 is bank-conflict free under gfx950's
// 4 x 16-lane read groups (tools/lds_banks32.py).  The key is applied to the SOURCE chunk a DMA lane fetches (LDS-DMA
// writes lane-linearly) and to the read address.  Weight rows are permuted in LDS so that a lane's 2 x 16 accumulator
// rows are 32 CONSECUTIVE output channels of its pixel: four 16-byte stores per pixel.
// (Producer-side InstanceNorm from this kernel's epilogue was built and measured in round 6: 0.156 -> 0.209 ms on layer 3, 0.129 ->
// 0.244 ms on layer 4 against the separate launch; profiles/r06_in_epilogue.md.)
#pragma once
#include <type_traits>
#include "common.h"
#include "lds_dma.h"

namespace eve {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct Wg8Params {
    int N, Cin, Cout;             // x: [N][W][W][Cin]  out: [N][W][W][Cout]  (W is a template parameter)
    int flip;                     // 0: forward taps (kh-1, kw-1);  1: data-gradient taps (1-kh, 1-kw)
    int K;                        // 9 * Cin (row stride of the weight matrix [Cout][3][3][Cin])
    uint32_t x_bytes, w_bytes;
    uint32_t tiles_n;             // channel tiles; blockIdx -> (image tile, channel tile)
    int s2_py;                    // NT != 9 (data gradient of a stride-2 convolution): output row parity of this launch
};

template <int W>
__device__ __forceinline__ int wg8_key(int hy, int hx) {
    return W >= 32 ? (hx >> 2) & 3 : (W >= 8 ? (hx >> 1) & 3 : hy & 3);
}
// output channel (inside a wave's 64) held by LDS weight row r64 = 32 * ct + i of that wave
__device__ __forceinline__ int wg8_row_channel(int r64) {
    const int ct = r64 >> 5, i = r64 & 31;
    return 32 * ((i >> 2) & 1) + 16 * ct + 4 * (i >> 3) + (i & 3);
}

// LDS-DMA with the uniform part of the source address in the scalar offset (no VALU work per piece)
__device__ __forceinline__ void wg8_dma(const eve_int4& rsrc, uint32_t lds, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}

// 16 MFMAs of one step: acc[ct][pt] += Wf[ct][kh] x Xf[pt][kh], K half outermost (8 MFMAs between two uses of an accumulator)
#define EVE_WG8_MMA_BODY(OP)                                                                                              \
    asm volatile(                                                                                                        \
        "s_nop 1\n\t"                                                                                                    \
        OP " %0, %8, %12, %0\n\t"  OP " %1, %8, %14, %1\n\t"  OP " %2, %8, %16, %2\n\t"  OP " %3, %8, %18, %3\n\t"          \
        OP " %4, %10, %12, %4\n\t" OP " %5, %10, %14, %5\n\t" OP " %6, %10, %16, %6\n\t" OP " %7, %10, %18, %7\n\t"        \
        OP " %0, %9, %13, %0\n\t"  OP " %1, %9, %15, %1\n\t"  OP " %2, %9, %17, %2\n\t"  OP " %3, %9, %19, %3\n\t"          \
        OP " %4, %11, %13, %4\n\t" OP " %5, %11, %15, %5\n\t" OP " %6, %11, %17, %6\n\t" OP " %7, %11, %19, %7"            \
        : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]),          \
          "+a"(acc[1][2]), "+a"(acc[1][3])                                                                               \
        : "v"(wf[0][0]), "v"(wf[0][1]), "v"(wf[1][0]), "v"(wf[1][1]), "v"(xf[0][0]), "v"(xf[0][1]), "v"(xf[1][0]),        \
          "v"(xf[1][1]), "v"(xf[2][0]), "v"(xf[2][1]), "v"(xf[3][0]), "v"(xf[3][1]))
template <typename H>
__device__ __forceinline__ void wg8_mma16(f32x16_t (&acc)[2][4], const u32x4_t (&wf)[2][2], const u32x4_t (&xf)[4][2]) {
    if constexpr (Elem<H>::IS_BF16) { EVE_WG8_MMA_BODY("v_mfma_f32_32x32x16_bf16"); }
    else { EVE_WG8_MMA_BODY("v_mfma_f32_32x32x16_f16"); }
}
#undef EVE_WG8_MMA_BODY

template <int N> __device__ __forceinline__ void wg8_wait_vm() {
    static_assert(N >= 0 && N <= 9, "pieces per phase");
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
}

// BANDS > 1 (round 5: 32 x 32 x 128, layer 2 on 256 x 256 patches): the tile's unit is a band of W / BANDS rows of one image instead
// of a whole image (its halo rows above / below come from the neighbouring bands or are the image border).
template <int WM, int WN, int W, int BANDS = 1> struct Wg8Geom {
    static constexpr int PIX = 128 * WM, COUT_T = 64 * WN;
    static constexpr int TH = W / BANDS, UPIX = TH * W;              // rows / pixels of a unit (whole image or band)
    static constexpr int TI = PIX / UPIX;                            // units per tile
    static constexpr int W2 = W + 2, HPI = (TH + 2) * W2, HP = TI * HPI;   // halo pixels per unit / per tile
    static constexpr int APT = (HP + 15) / 16;                       // 1 KB halo pieces per slice
    static constexpr int AP = (APT + 7) / 8;                         // ... per wave
    static constexpr int ASTAGE = AP * 8 * 1024;
    static constexpr int BSLOT = COUT_T * 64;
    static constexpr int BP = COUT_T / 16 / 8;                       // weight pieces per wave and step
    static constexpr int LDS = 2 * ASTAGE + 4 * BSLOT;
    static_assert(WM * WN == 8 && W % BANDS == 0 && PIX % UPIX == 0 && TI >= 1 && (BANDS == 1 || TI == 1) && AP <= 7 && BP >= 1 &&
                  LDS <= 160 * 1024, "tile geometry");
};

// NT = 9: the 3x3 convolution.  NT = 2 / 4: the DATA GRADIENT OF A STRIDE-2 3x3 CONVOLUTION (torchvision BasicBlock's first
// convolution of layers 2-4, eye_net.py:48-50) as a stride-1 convolution over dy: d(x)[2 yy + py][2 xx + px] only takes the
// filter taps kh = py + 1 - 2 dy, kw = px + 1 - 2 dx with dy, dx in {0, 1}, i.e. dy[yy + dy][xx + dx] -- a 1 x 2 (py = 0) or
// 2 x 2 (py = 1) window over the SAME halo tile, whose 2 x Cin "output channels" are the column parities px = 0 | 1 of d(x)
// (px = 0 uses half of those taps: its other weights are zero, 12 of 16 tap-classes do real work).  The launch pair reads dy
// twice in total (the per-tap kernel: four launches, nine tap passes) and the epilogue scatters depth-to-space.
template <typename H, int WM, int WN, int W, int NT = 9, int BANDS = 1>
__global__ __launch_bounds__(512, 2) void conv3x3_wg8_kernel(const Wg8Params p, const H* __restrict__ x,
                                                             const H* __restrict__ w, const float* __restrict__ bias,
                                                             const int epi_act, H* __restrict__ out) {
    using G = Wg8Geom<WM, WN, W, BANDS>;
    static_assert(BANDS == 1 || (NT == 9 && W >= 8), "row bands: the 3x3 convolution on W >= 8");
    constexpr int W2 = G::W2, HPI = G::HPI, HP = G::HP, AP = G::AP, BP = G::BP, ASTAGE = G::ASTAGE, BSLOT = G::BSLOT;
    constexpr int TH = G::TH, UPIX = G::UPIX;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t tm = lid / p.tiles_n, tn = lid - tm * p.tiles_n;
    const uint32_t n0 = tm * G::TI, co0 = tn * G::COUT_T;

    const eve_int4 rs_x = make_rsrc_words(x, p.x_bytes);
    const eve_int4 rs_w = make_rsrc_words(w, p.w_bytes);
    const uint32_t ldsA = lds_addr_of(smem), ldsB = ldsA + 2 * ASTAGE;

    // ---- DMA source offsets (lane constants; the channel slice / filter tap goes into the scalar offset) ----
    int a_goff[AP];
#pragma unroll
    for (int j = 0; j < AP; ++j) {
        const int hp = (j * 8 + wave) * 16 + (lane >> 2), pc = lane & 3;
        int off = EVE_OOB;
        if (hp < HP) {
            const int ti = hp / HPI, r = hp - ti * HPI, hy = r / W2, hx = r - hy * W2;
            const uint32_t u = n0 + ti, n = u / BANDS;
            const int gy = (int)(u % BANDS) * TH + hy - 1, gx = hx - 1;
            if (gy >= 0 && gy < W && gx >= 0 && gx < W && n < (uint32_t)p.N)
                off = (int)((((n * W + gy) * W + gx) * p.Cin) * 2) + ((pc ^ wg8_key<W>(hy, hx)) << 4);
        }
        a_goff[j] = off;
    }
    int b_goff[BP];
#pragma unroll
    for (int j = 0; j < BP; ++j) {
        const int row = (j * 8 + wave) * 16 + (lane >> 2), pc = lane & 3;
        const uint32_t co = co0 + (row & ~63) + wg8_row_channel(row & 63);
        b_goff[j] = co < (uint32_t)p.Cout ? (int)(co * (uint32_t)p.K) * 2 + ((pc ^ ((row >> 2) & 3)) << 4) : EVE_OOB;
    }

    // ---- fragment read addresses (lane constants; stage / tap row or column / ring slot are added per step) ----
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, l5 = lane >> 5;
    uint32_t wrd[2][2];                                         // [channel tile][K half], without the ring slot
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int row = wn * 64 + ct * 32 + l31;
            wrd[ct][kh] = ldsB + row * 64 + (((2 * kh + l5) ^ ((row >> 2) & 3)) << 4);
        }
    // W >= 8: the key depends on the halo column -> one address per tap column, the tap row is an immediate (dy * W2 * 64);
    // W == 4: the key depends on the halo row    -> one address per tap row, the tap column is an immediate (dx * 64)
    uint32_t xrd[4][3][2];                                      // [pixel tile][dx or dy][K half]
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const int m = wm * 128 + pt * 32 + l31;
        const int ti = m / UPIX, ty = (m / W) % TH, tx = m % W;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int hy = W >= 8 ? ty : ty + q, hx = W >= 8 ? tx + q : tx;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
                xrd[pt][q][kh] = ldsA + (ti * HPI + hy * W2 + hx) * 64 + (((2 * kh + l5) ^ wg8_key<W>(hy, hx)) << 4);
        }
    }

    f32x16_t acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nslices = p.Cin / 32;                             // even (the launcher requires Cin % 64 == 0)
    const int cin2 = p.Cin * 2;
    // weight tile of filter position (dy, dx): forward tap dy*3+dx, data gradient tap 8 - (dy*3+dx)
    const int tap_base = (NT == 9 && p.flip) ? 8 * cin2 : 0, tap_step = (NT == 9 && p.flip) ? -cin2 : cin2;
    auto issue_b = [&](int s, int t, int slot) {                // this wave's BP pieces of the tile of (slice s, position t)
        const int soff = tap_base + t * tap_step + s * 64;
#pragma unroll
        for (int j = 0; j < BP; ++j) wg8_dma(rs_w, ldsB + slot * BSLOT + (j * 8 + wave) * 1024, b_goff[j], soff);
    };

    // ---- prologue: halo of slice 0, weight tiles of steps 0 and 1 ----
#pragma unroll
    for (int j = 0; j < AP; ++j) wg8_dma(rs_x, ldsA + (j * 8 + wave) * 1024, a_goff[j], 0);
    issue_b(0, 0, 0);
    issue_b(0, 1, 1);
    wg8_wait_vm<0>();
    __builtin_amdgcn_s_barrier();

    // One barrier per step.  Between two barriers group 0 runs [read step g, multiply step g] and group 1 runs
    // [multiply step g-1, read step g]: on every SIMD one wave starts with LDS / DMA work while its partner (higher
    // priority) owns the matrix pipe, then they swap -- no barrier is needed for that, the pipe itself orders them.
    const bool lead = wave < 4;
    u32x4_t wf[2][2], xf[4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[i][j] = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) xf[i][j] = u32x4_t{0u, 0u, 0u, 0u};
    if (!lead) __builtin_amdgcn_s_setprio(2);

    for (int s2 = 0; s2 < nslices; s2 += 2) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const int s = s2 + ss;
            const bool last = s + 1 == nslices;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int dy = NT == 9 ? t / 3 : (NT == 4 ? 1 + t / 2 : 1), dx = NT == 9 ? t % 3 : 1 + (t & 1);
                const int t2 = (t + 2) % NT, sd = (t + 2) / NT;
                const bool more_b = !(last && sd);                                       // a step g + 2 exists
                // halo pieces of the next slice issued in this step: one per step from the second tap on (3x3), or all AP of
                // them dealt over the first NT - 1 steps (short windows; a piece is waited for at the END of the step after
                // its own, so none may be issued in a slice's last step)
                constexpr int PPS = NT == 9 ? 1 : (AP + NT - 2) / (NT - 1);
                const int j0 = NT == 9 ? t - 1 : t * PPS;
                const int np = NT == 9 ? ((t >= 1 && t <= AP) ? 1 : 0) : (j0 >= AP ? 0 : (j0 + PPS <= AP ? PPS : AP - j0));
                if (!lead) wg8_mma16<H>(acc, wf, xf);                                    // step g - 1 (zeros before step 0)
                // ---- fragments of step g = NT s + t ----
                const uint32_t slot_off = (uint32_t)((NT * s + t) & 3) * BSLOT;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh)
                        wf[ct][kh] = *reinterpret_cast<const EVE_LDS u32x4_t*>((uintptr_t)(wrd[ct][kh] + slot_off));
                const int imm = ss * ASTAGE + (W >= 8 ? dy * W2 * 64 : dx * 64);
                const int q = W >= 8 ? dx : dy;
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh)
                        xf[pt][kh] = *reinterpret_cast<const EVE_LDS u32x4_t*>((uintptr_t)(xrd[pt][q][kh] + imm));
                // ---- DMAs of this step (issued while the fragment reads are in flight): weight tile of step g + 2, halo
                //      pieces of the next slice ----
                if (more_b) issue_b(s + sd, t2, (NT * s + t + 2) & 3);
                if (!last) {
#pragma unroll
                    for (int jj = 0; jj < PPS; ++jj)
                        if (jj < np) wg8_dma(rs_x, ldsA + (ss ^ 1) * ASTAGE + ((j0 + jj) * 8 + wave) * 1024, a_goff[jj < np ? j0 + jj : 0], (s + 1) * 64);
                }
                if (lead) wg8_mma16<H>(acc, wf, xf);
                // everything this wave issued BEFORE this step has landed once only this step's pieces are outstanding
                if (more_b) {
                    if (!last) {
                        if (np == 0) wg8_wait_vm<BP>();
                        else if (np == 1) wg8_wait_vm<BP + 1>();
                        else if (np == 2) wg8_wait_vm<BP + 2>();
                        else if (np == 3) wg8_wait_vm<BP + 3>();
                        else if (np == 4) wg8_wait_vm<BP + 4>();
                        else if (np == 5) wg8_wait_vm<BP + 5>();
                        else wg8_wait_vm<BP + 6>();
                    } else wg8_wait_vm<BP>();
                } else {
                    wg8_wait_vm<0>();
                }
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    if (!lead) wg8_mma16<H>(acc, wf, xf);                       // group 1's last step
    __builtin_amdgcn_s_setprio(0);
    // MFMA results -> compiler-generated reads of the accumulators: the wait states must sit BETWEEN the last MFMA and the
    // first v_accvgpr_read, so the statement names the accumulators (an asm with only a memory clobber may be scheduled
    // after register-only reads: the last tile of the last wave then came out stale every few launches)
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]),
                   "+a"(acc[1][2]), "+a"(acc[1][3]) :: "memory");

    // ---- epilogue: the lane owns 32 consecutive channels (64 bytes) of each of its four pixels ----
    const uint32_t co = co0 + wn * 64 + l5 * 32;
    const bool relu = (epi_act & 0xff) == EVE_ACT_RELU;
    float bv[32];
#pragma unroll
    for (int c = 0; c < 32; c += 4) {
        const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + co + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        bv[c] = b4.x; bv[c + 1] = b4.y; bv[c + 2] = b4.z; bv[c + 3] = b4.w;
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        __builtin_amdgcn_sched_barrier(0);                      // one pixel tile at a time: 32 accumulators live, not 128
        const int m = wm * 128 + pt * 32 + l31;
        const int ti = m / UPIX;
        const uint32_t u = n0 + ti, n = u / BANDS;
        const int pix = m - ti * UPIX + (int)(u % BANDS) * UPIX;      // pixel index inside image n
        u32x4_t pk[4];                                          // the lane's pixel: chunks k = 0..3 of 8 channels
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float o0 = acc[ct][pt][r] + bv[ct * 16 + r], o1 = acc[ct][pt][r + 1] + bv[ct * 16 + r + 1];
                if (relu) { o0 = fmaxf(o0, 0.f); o1 = fmaxf(o1, 0.f); }
                pk[ct * 2 + (r >> 3)][(r >> 1) & 3] = Elem<H>::pack2(o0, o1);
            }
        // across each lane quad (four consecutive pixels of one image): lane q takes chunk q of pixel m in pk[m], so a store
        // writes 64 contiguous bytes per quad instead of 16 bytes into 64 lines (conv_ws64.h measured the difference)
        quad_transpose4x4(pk, lane);
        if (n < (uint32_t)p.N) {
            if constexpr (NT == 9) {
                H* dst = out + ((size_t)n * (W * W) + (pix & ~3)) * p.Cout + co + (l31 & 3) * 8;
#pragma unroll
                for (int v = 0; v < 4; ++v) *reinterpret_cast<u32x4_t*>(dst + (size_t)v * p.Cout) = pk[v];
            } else {
                // depth-to-space: "channel" co = px * Cdx + ci of dy-grid pixel (yy, xx) is d(x)[2 yy + py][2 xx + px][ci]
                const int cdx = p.Cout >> 1, px = (int)co >= cdx ? 1 : 0, ci = (int)co - px * cdx;
                const int pb = pix & ~3, yy = pb / W, xx = pb - yy * W;
                H* dst = out + (((size_t)n * (2 * W) + 2 * yy + p.s2_py) * (2 * W) + 2 * xx + px) * cdx + ci + (l31 & 3) * 8;
#pragma unroll
                for (int v = 0; v < 4; ++v) *reinterpret_cast<u32x4_t*>(dst + (size_t)(2 * v) * cdx) = pk[v];
            }
        }
    }
}

}  // namespace eve
