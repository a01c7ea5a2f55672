// 3x3 / stride 2 / pad 1 convolution FORWARD of the ResNet trunk (the first convolution of layers 2-4: torchvision BasicBlock
// with a down-sample, built at /root/reference/src/models/eye_net.py:48-50) on the eight-wave halo design of conv_wg8.h.
//
// out[oy][ox] = sum_{kh,kw} x[2 oy + kh - 1][2 ox + kw - 1] w[kh][kw].  With the four PARITY PLANES of the input,
// x_pq[i][j] = x[2 i + p][2 j + q], every tap reads ONE plane at a unit-stride window position:
//     plane (1,1): kh, kw in {0, 2}  ->  (i, j) in {oy - 1, oy} x {ox - 1, ox}    4 taps
//     plane (1,0): kh in {0, 2}, kw = 1  ->  i in {oy - 1, oy}, j = ox             2 taps
//     plane (0,1): kh = 1, kw in {0, 2}  ->  i = oy, j in {ox - 1, ox}             2 taps
//     plane (0,0): kh = kw = 1           ->  (oy, ox)                              1 tap
// so a 32-channel slice is nine steps again -- exactly the 3x3 convolution's FLOPs -- but its operand is four halo planes.
// All four at once would be 74-83 KB per slice and stage; instead ONE plane is a stage ((W + 1) rows x (W + 2) columns per
// image: pad above / left, one spare column so that the row pitch and with it conv_wg8.h's conflict-free chunk keys carry
// over), three stages rotate, and a plane's pieces are fetched by LDS-DMA two to five steps before its first tap:
//
//     step (slice-relative)   0     1     2     3     4     5     6     7     8
//     multiplies plane       (1,1) (1,1) (1,1) (1,1) (1,0) (1,0) (0,1) (0,1) (0,0)
//     fetches plane          (1,0) (1,0) (0,1) (0,1) (0,0) (0,0) (1,1)'(1,1)'  -        ' = next slice
//
// (a piece issued in step g is waited for at the end of step g + 1 and may be read from step g + 2 on; a stage's previous
// tenant -- three planes earlier -- has had its last tap at least one full step before the first piece is issued.)  The
// LDS-DMA lets every lane fetch any pixel, so the planes are gathered straight from the NHWC input: the lane-constant part
// of a source address is plane-independent (pixel (2 i, 2 j)), the plane is the scalar offset (p * 2W + q) pixels.  Weight
// tiles, the ring, the role-split wave pairs, the fragment layout and the epilogue are conv_wg8.h's; the filter needs no
// re-packing (a step's tile is tap (kh, kw) of the ordinary [Cout][3][3][Cin] bank).
#pragma once
#include "conv_wg8.h"

namespace eve {

template <int WM, int WN, int W> struct Wg8S2Geom {
    static constexpr int PIX = 128 * WM, COUT_T = 64 * WN;
    static constexpr int TI = PIX / (W * W);                         // whole OUTPUT images per tile
    static constexpr int W2 = W + 2, HPI = (W + 1) * W2, HP = TI * HPI;
    static constexpr int APT = (HP + 15) / 16, AP = (APT + 7) / 8;
    static constexpr int ASTAGE = AP * 8 * 1024;
    static constexpr int BSLOT = COUT_T * 64, BP = COUT_T / 16 / 8;
    static constexpr int LDS = 3 * ASTAGE + 4 * BSLOT;
    static_assert(WM * WN == 8 && PIX % (W * W) == 0 && AP <= 6 && BP >= 1 && LDS <= 160 * 1024, "tile geometry");
};

template <typename H, int WM, int WN, int W>
__global__ __launch_bounds__(512, 2) void conv3x3s2_wg8_kernel(const Wg8Params p, const H* __restrict__ x,
                                                               const H* __restrict__ w, const float* __restrict__ bias,
                                                               const int epi_act, H* __restrict__ out) {
    using G = Wg8S2Geom<WM, WN, W>;
    constexpr int W2 = G::W2, HPI = G::HPI, HP = G::HP, AP = G::AP, BP = G::BP, ASTAGE = G::ASTAGE, BSLOT = G::BSLOT;
    constexpr int WIN = 2 * W;                                  // input width
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t tm = lid / p.tiles_n, tn = lid - tm * p.tiles_n;
    const uint32_t n0 = tm * G::TI, co0 = tn * G::COUT_T;

    const eve_int4 rs_x = make_rsrc_words(x, p.x_bytes);
    const eve_int4 rs_w = make_rsrc_words(w, p.w_bytes);
    const uint32_t ldsA = lds_addr_of(smem), ldsB = ldsA + 3 * ASTAGE;

    // ---- halo slots (lane constants, the same for the four planes): input pixel (2 i, 2 j) of plane position (i, j) ----
    int a_goff[AP];
#pragma unroll
    for (int j = 0; j < AP; ++j) {
        const int hp = (j * 8 + wave) * 16 + (lane >> 2), pc = lane & 3;
        int off = EVE_OOB;
        if (hp < HP) {
            const int ti = hp / HPI, r = hp - ti * HPI, hy = r / W2, hx = r - hy * W2;
            const int pi = hy - 1, pj = hx - 1;
            const uint32_t n = n0 + ti;
            if (pi >= 0 && pj >= 0 && pj < W && n < (uint32_t)p.N)
                off = (int)((((n * WIN + 2 * pi) * WIN + 2 * pj) * p.Cin) * 2) + ((pc ^ wg8_key<W>(hy, hx)) << 4);
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

    // ---- fragment read addresses (lane constants; stage, window row / column and ring slot are added per step) ----
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, l5 = lane >> 5;
    uint32_t wrd[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int row = wn * 64 + ct * 32 + l31;
            wrd[ct][kh] = ldsB + row * 64 + (((2 * kh + l5) ^ ((row >> 2) & 3)) << 4);
        }
    // window position (a, b) in {0, 1}^2 of halo pixel (ty + a, tx + b).  W >= 8: the key depends on the halo column -> one
    // address per b, a is an immediate (a * W2 * 64);  W == 4: the key depends on the halo row -> one per a, b is an immediate
    uint32_t xrd[4][2][2];                                      // [pixel tile][b or a][K half]
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const int m = wm * 128 + pt * 32 + l31;
        const int ti = m / (W * W), ty = (m / W) % W, tx = m % W;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
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
    auto issue_b = [&](int s, int t, int slot) {                // weight tile of (slice s, step t): tap (kh, kw) of the filter
        // steps 0..8 -> taps (0,0) (0,2) (2,0) (2,2) | (0,1) (2,1) | (1,0) (1,2) | (1,1)
        const int tap = t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 6 : t == 3 ? 8 : t == 4 ? 1 : t == 5 ? 7 : t == 6 ? 3 : t == 7 ? 5 : 4;
        const int soff = tap * cin2 + s * 64;
#pragma unroll
        for (int j = 0; j < BP; ++j) wg8_dma(rs_w, ldsB + slot * BSLOT + (j * 8 + wave) * 1024, b_goff[j], soff);
    };
    // pieces [j0, j0 + cnt) of plane (pp, pq) of slice s into stage `st`
    auto issue_a = [&](int s, int pp, int pq, uint32_t st_bytes, int j0, int cnt) {
        const int soff = s * 64 + (pp * WIN + pq) * cin2;
#pragma unroll
        for (int j = 0; j < AP; ++j)
            if (j >= j0 && j < j0 + cnt) wg8_dma(rs_x, ldsA + st_bytes + (j * 8 + wave) * 1024, a_goff[j], soff);
    };
    constexpr int PA = (AP + 1) / 2, PB = AP - PA;               // pieces in the first / second step of a plane's fetch window

    // ---- prologue: plane (1,1) of slice 0 -> stage 0, weight tiles of steps 0 and 1 ----
    issue_a(0, 1, 1, 0, 0, AP);
    issue_b(0, 0, 0);
    issue_b(0, 1, 1);
    wg8_wait_vm<0>();
    __builtin_amdgcn_s_barrier();

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

    // stage of plane index g4 = 4 s + pl (pl = 0: (1,1), 1: (1,0), 2: (0,1), 3: (0,0)) is g4 % 3 = (s + pl) % 3: three uniform
    // byte offsets st[k] = stage of the CURRENT slice's plane pl = k (and of pl = 3 for k = 0), rotated by one per slice.
    // (Run-time stage offsets cost one v_add per fragment read; a three-deep unrolled slice loop with compile-time stages needs
    //  an exit in its middle and the register allocator then spills accumulators.)
    uint32_t st0 = 0, st1 = ASTAGE, st2 = 2 * ASTAGE;
    for (int s = 0; s < nslices; ++s) {
        const bool last = s + 1 == nslices;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int pl = t < 4 ? 0 : (t < 6 ? 1 : (t < 8 ? 2 : 3));                     // plane index of this step
            const int a = t < 4 ? t >> 1 : (t == 4 ? 0 : 1), b = t < 4 ? t & 1 : (t == 6 ? 0 : 1);
            const uint32_t stg = pl == 1 ? st1 : (pl == 2 ? st2 : st0);
            const int t2 = (t + 2) % 9, sd = (t + 2) / 9;
            const bool more_b = !(last && sd);
            // plane fetched in this step: t 0,1 -> (1,0) of s; 2,3 -> (0,1) of s; 4,5 -> (0,0) of s; 6,7 -> (1,1) of s + 1
            const int fpl = t < 2 ? 1 : (t < 4 ? 2 : (t < 6 ? 3 : 4));                   // 4 = plane 0 of the next slice
            const bool fetch = t < 8 && !(fpl == 4 && last);
            const int np = !fetch ? 0 : ((t & 1) ? PB : PA);
            if (!lead) wg8_mma16<H>(acc, wf, xf);                                        // step g - 1 (zeros before step 0)
            const uint32_t slot_off = (uint32_t)((s + t) & 3) * BSLOT;                 // (9 s + t) & 3
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
                    wf[ct][kh] = *reinterpret_cast<const EVE_LDS u32x4_t*>((uintptr_t)(wrd[ct][kh] + slot_off));
            const int imm = W >= 8 ? a * W2 * 64 : b * 64;
            const int q = W >= 8 ? b : a;
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
                    xf[pt][kh] = *reinterpret_cast<const EVE_LDS u32x4_t*>((uintptr_t)(xrd[pt][q][kh] + stg + imm));
            if (more_b) issue_b(s + sd, t2, (s + t + 2) & 3);
            if (fetch) {
                // (1,0) -> st1, (0,1) -> st2, (0,0) -> st0 (plane 3 = 0 mod 3: the stage (1,1) left after step 3);
                // (1,1) of the next slice -> next slice's st0 = this slice's st1 (free after step 5)
                const uint32_t fst = fpl == 1 ? st1 : (fpl == 2 ? st2 : (fpl == 3 ? st0 : st1));
                const int fp = fpl == 4 ? 0 : fpl;
                issue_a(fpl == 4 ? s + 1 : s, fp == 0 || fp == 1 ? 1 : 0, fp == 0 || fp == 2 ? 1 : 0, fst, (t & 1) ? PA : 0, np);
            }
            if (lead) wg8_mma16<H>(acc, wf, xf);
            if (more_b) {
                if (np == 0) wg8_wait_vm<BP>();
                else if (np == 1) wg8_wait_vm<BP + 1>();
                else if (np == 2) wg8_wait_vm<BP + 2>();
                else wg8_wait_vm<BP + 3>();
            } else {
                wg8_wait_vm<0>();
            }
            __builtin_amdgcn_s_barrier();
        }
        const uint32_t r = st0; st0 = st1; st1 = st2; st2 = r;
    }
    if (!lead) wg8_mma16<H>(acc, wf, xf);                       // group 1's last step
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]),
                   "+a"(acc[1][2]), "+a"(acc[1][3]) :: "memory");

    // ---- epilogue (conv_wg8.h): the lane owns 32 consecutive channels of each of its four pixels, stored quad-transposed ----
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
        __builtin_amdgcn_sched_barrier(0);
        const int m = wm * 128 + pt * 32 + l31;
        const int ti = m / (W * W), pix = m - ti * (W * W);
        const uint32_t n = n0 + ti;
        u32x4_t pk[4];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float o0 = acc[ct][pt][r] + bv[ct * 16 + r], o1 = acc[ct][pt][r + 1] + bv[ct * 16 + r + 1];
                if (relu) { o0 = fmaxf(o0, 0.f); o1 = fmaxf(o1, 0.f); }
                pk[ct * 2 + (r >> 3)][(r >> 1) & 3] = Elem<H>::pack2(o0, o1);
            }
        quad_transpose4x4(pk, lane);
        if (n < (uint32_t)p.N) {
            H* dst = out + ((size_t)n * (W * W) + (pix & ~3)) * p.Cout + co + (l31 & 3) * 8;
#pragma unroll
            for (int v = 0; v < 4; ++v) *reinterpret_cast<u32x4_t*>(dst + (size_t)v * p.Cout) = pk[v];
        }
    }
}

}  // namespace eve
