// bf16 weight gradient of 3x3 / stride 1 / pad 1 convolutions with FEW channels (16..64 in, 16..32 out) on large planes:
// RefineNet's 72x128 level (/root/reference/src/models/refine_net.py:96-131), 8.8 M pixels per step at B=32 x T=30.
//
// wgrad_tr_kernel gathers its filter-column operand tap by tap: every x pixel travels L2 -> LDS nine times, and with 16
// channels (32 bytes per pixel) that gather, not the MFMAs, sets the time (320-920 us per layer, ~6 TB/s of L2 reads for
// 70 us worth of HBM traffic).  Here a band of TH image rows is made resident once -- x with its one-pixel halo, dy
// without -- and all nine taps read the same LDS tile at shifted addresses:
//     dw[co][kh][kw][ci] += sum over the band's pixels of dy[y][x][co] * x[y + kh - 1][x + kw - 1][ci]
// One MFMA 16x16x32 = 16 output channels x 16 input channels x 32 pixels of one image row; both operands come out of
// the natural [pixel][channel] tiles through the transposing LDS read.  A wave owns whole 32-pixel chunks and carries
// the full 9 x MT x CT accumulator set; workgroups stream bands persistently, and reduce through LDS float atomics before
// touching the global gradient.  The bias gradient rides along as one more MFMA against an all-ones operand.
#pragma once

namespace eve {

struct WgradHaloParams {
    int N, H, W, TH, bands;            // bands per image = ceil(H / TH)
    uint32_t total_bands, x_bytes, dy_bytes;
    int log2_cpr;                      // log2(W / 32): 32-pixel chunks per row
    int nreg;                          // wgrad_halo64_kernel: bands resident in LDS (2 or 3)
};

// three in-place MFMAs sharing the A operand (one filter row): c[i] += a x b[i]
__device__ __forceinline__ void mma3_bf16_inplace(f32x4_t& c0, f32x4_t& c1, f32x4_t& c2, const uint4& a4, const uint4 (&b4)[3]) {
    const u32x4_t a = __builtin_bit_cast(u32x4_t, a4);
    const u32x4_t b0 = __builtin_bit_cast(u32x4_t, b4[0]), b1 = __builtin_bit_cast(u32x4_t, b4[1]), b2 = __builtin_bit_cast(u32x4_t, b4[2]);
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n\t"
        "v_mfma_f32_16x16x32_bf16 %1, %3, %5, %1\n\t"
        "v_mfma_f32_16x16x32_bf16 %2, %3, %6, %2"
        : "+v"(c0), "+v"(c1), "+v"(c2)                       // accumulators in architectural VGPRs (unified file on gfx950):
        : "v"(a), "v"(b0), "v"(b1), "v"(b2));                 // the epilogue reads them without a copy out of the AGPRs
}
__device__ __forceinline__ void mma1_bf16_inplace(f32x4_t& c, const uint4& a4, const uint4& b4) {
    const u32x4_t a = __builtin_bit_cast(u32x4_t, a4), b = __builtin_bit_cast(u32x4_t, b4);
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mma3_f16_inplace(f32x4_t& c0, f32x4_t& c1, f32x4_t& c2, const uint4& a4, const uint4 (&b4)[3]) {
    const u32x4_t a = __builtin_bit_cast(u32x4_t, a4);
    const u32x4_t b0 = __builtin_bit_cast(u32x4_t, b4[0]), b1 = __builtin_bit_cast(u32x4_t, b4[1]), b2 = __builtin_bit_cast(u32x4_t, b4[2]);
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_f16 %0, %3, %4, %0\n\t"
        "v_mfma_f32_16x16x32_f16 %1, %3, %5, %1\n\t"
        "v_mfma_f32_16x16x32_f16 %2, %3, %6, %2"
        : "+v"(c0), "+v"(c1), "+v"(c2)                       // accumulators in architectural VGPRs (unified file on gfx950):
        : "v"(a), "v"(b0), "v"(b1), "v"(b2));                 // the epilogue reads them without a copy out of the AGPRs
}
__device__ __forceinline__ void mma1_f16_inplace(f32x4_t& c, const uint4& a4, const uint4& b4) {
    const u32x4_t a = __builtin_bit_cast(u32x4_t, a4), b = __builtin_bit_cast(u32x4_t, b4);
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <typename H>
__device__ __forceinline__ void mma3_inplace(f32x4_t& c0, f32x4_t& c1, f32x4_t& c2, const uint4& a4, const uint4 (&b4)[3]) {
    if constexpr (Elem<H>::IS_BF16) mma3_bf16_inplace(c0, c1, c2, a4, b4); else mma3_f16_inplace(c0, c1, c2, a4, b4);
}
template <typename H>
__device__ __forceinline__ void mma1_inplace(f32x4_t& c, const uint4& a4, const uint4& b4) {
    if constexpr (Elem<H>::IS_BF16) mma1_bf16_inplace(c, a4, b4); else mma1_f16_inplace(c, a4, b4);
}

// KS = 3: the 3x3 / pad 1 case above.  KS = 1: 1x1 convolutions of the same planes (skip layers): no halo, one tap --
// a plain [Cout][Cin] += dy^T x over the band, where the gather kernel's 256-wide K tile is 6-25 % occupied.
template <typename H, int MT, int CT, int KS = 3>
__global__ __launch_bounds__(256) void wgrad_halo_kernel(const WgradHaloParams p, const H* __restrict__ x,
                                                         const H* __restrict__ dy, float* __restrict__ dw,
                                                         float* __restrict__ db) {
    constexpr int CIN = 16 * CT, COUT = 16 * MT;
    constexpr int XROW = CIN * 2, DROW = COUT * 2;           // bytes per pixel
    constexpr int XS = XROW / 16, DSL = DROW / 16;           // 16-byte slots per pixel
    constexpr int TAPS = KS * KS, PAD = KS / 2;
    constexpr int NRED = TAPS * CIN * COUT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int W = p.W, W2 = W + 2 * PAD, TH = p.TH;
    const int xs_bytes = (TH + 2 * PAD) * W2 * XROW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 15, g = lane >> 4;
    const uint32_t lds0 = lds_addr_of(lds), ldsD = lds0 + (uint32_t)xs_bytes;

    // the two halo columns never receive data: zero them once
    for (int i = tid; i < (KS == 3 ? (TH + 2) * 2 * XS : 0); i += 256) {
        const int row = i / (2 * XS), rem = i - row * 2 * XS;
        const int col = rem / XS ? W + 1 : 0, ch = rem % XS;
        *reinterpret_cast<uint4*>(lds + (size_t)((row * W2 + col) * XROW + ch * 16)) = make_uint4(0u, 0u, 0u, 0u);
    }
    const eve_int4 rs_x = make_rsrc_words(x, p.x_bytes);
    const eve_int4 rs_dy = make_rsrc_words(dy, p.dy_bytes);

    f32x4_t acc[MT][CT][TAPS], accb[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        accb[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < CT; ++b)
#pragma unroll
            for (int c = 0; c < TAPS; ++c) acc[a][b][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const uint4 ones = make_uint4(Elem<H>::ONE2, Elem<H>::ONE2, Elem<H>::ONE2, Elem<H>::ONE2);
    // lane-constant part of the transposing reads: pixel 8g + t/4 (+4 for the second read), channels 4 (t%4) .. +3
    const int lrow = 8 * g + (t >> 2), sub = (t & 3) * 8;
    const int nchunks = TH << p.log2_cpr;
    const int xslots = W * XS, dslots = W * DSL;             // 16-byte slots per image row

    for (uint32_t band = blockIdx.x; band < p.total_bands; band += gridDim.x) {
        const int n = (int)(band / (uint32_t)p.bands), y0 = (int)(band % (uint32_t)p.bands) * TH;
        __syncthreads();                                      // the previous band has been consumed
        for (int hy = 0; hy < TH + 2 * PAD; ++hy) {
            const int gy = y0 - PAD + hy;
            const bool ok = gy >= 0 && gy < p.H;
            const int gbase = ((n * p.H + gy) * W) * XROW;
            const uint32_t lrow0 = lds0 + (uint32_t)((hy * W2 + PAD) * XROW);
            for (int s0 = wave * 64; s0 < xslots; s0 += 256)
                lds_dma16_asm(rs_x, lrow0 + s0 * 16, ok ? gbase + (s0 + lane) * 16 : EVE_OOB);
        }
        for (int ty = 0; ty < TH; ++ty) {
            const int gy = y0 + ty;
            const bool ok = gy < p.H;
            const int gbase = ((n * p.H + gy) * W) * DROW;
            const uint32_t lrow0 = ldsD + (uint32_t)(ty * W * DROW);
            for (int s0 = wave * 64; s0 < dslots; s0 += 256)
                lds_dma16_asm(rs_dy, lrow0 + s0 * 16, ok ? gbase + (s0 + lane) * 16 : EVE_OOB);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        for (int c = wave; c < nchunks; c += 4) {
            const int ty = c >> p.log2_cpr, x0 = (c - (ty << p.log2_cpr)) * 32;
            uint4 fp[MT];
            const uint32_t pa = ldsD + (uint32_t)((ty * W + x0 + lrow) * DROW + sub);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const uint2 a0 = lds_tr_read(pa + mt * 32);
                const uint2 a1 = lds_tr_read<4 * DROW>(pa + mt * 32);
                fp[mt] = make_uint4(a0.x, a0.y, a1.x, a1.y);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if constexpr (KS == 3) {
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {         // one filter row at a time keeps the operand registers few
                        uint4 fq[3];
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const uint32_t qa = lds0 + (uint32_t)(((ty + kh) * W2 + x0 + kw + lrow) * XROW + ct * 32 + sub);
                            const uint2 b0 = lds_tr_read(qa);
                            const uint2 b1 = lds_tr_read<4 * XROW>(qa);
                            fq[kw] = make_uint4(b0.x, b0.y, b1.x, b1.y);
                        }
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            mma3_inplace<H>(acc[mt][ct][kh * 3], acc[mt][ct][kh * 3 + 1], acc[mt][ct][kh * 3 + 2], fp[mt], fq);
                    }
                } else {
                    const uint32_t qa = lds0 + (uint32_t)((ty * W + x0 + lrow) * XROW + ct * 32 + sub);
                    const uint2 b0 = lds_tr_read(qa);
                    const uint2 b1 = lds_tr_read<4 * XROW>(qa);
                    const uint4 fq = make_uint4(b0.x, b0.y, b1.x, b1.y);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) mma1_inplace<H>(acc[mt][ct][0], fp[mt], fq);
                }
            }
            if (db) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mma1_inplace<H>(accb[mt], fp[mt], ones);
            }
        }
    }
    mma_drain();

    // ---- workgroup reduction through LDS float atomics, then one global atomic per filter element ----
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);
    for (int i = tid; i < NRED + COUT; i += 256) red[i] = 0.f;
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);                       // accumulators leave the AGPRs tile by tile, below this line
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = mt * 16 + g * 4 + r, ci = ct * 16 + t;
                    atomicAdd(&red[(co * TAPS + tap) * CIN + ci], acc[mt][ct][tap][r]);
                }
                __builtin_amdgcn_sched_barrier(0);           // (else all 36 tiles are copied out of the AGPRs at once: 144 VGPRs)
            }
    if (db && t == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(&red[NRED + mt * 16 + g * 4 + r], accb[mt][r]);
    }
    __syncthreads();
    for (int i = tid; i < NRED; i += 256) atomicAdd(dw + i, red[i]);
    if (db)
        for (int i = tid; i < COUT; i += 256) atomicAdd(db + i, red[NRED + i]);
}

// 64 -> 64 channels (layer 1 of the ResNet trunk, /root/reference/src/models/eye_net.py:48-50): the 9 x 4 x 4 accumulator
// tiles do not fit one wave, and a single-buffered band leaves the two workgroups of a CU loading and computing in lockstep.
// Here ONE workgroup of eight waves per CU (two per SIMD) keeps nreg = 2 or 3 bands in LDS: the next band(s) stream in while
// band k is multiplied.  Wave w owns input-channel tile ct = w & 3 for ALL 64 output channels (36 accumulator tiles) and
// walks the chunks of parity w >> 2: 4 + 9 fragment reads per 36 MFMAs, no reduction through LDS in the loop (the filter
// elements of the four ct are disjoint).  x crosses HBM (TH + 2) / TH times instead of the gather kernel's 2.8x.
// LDS rows are 128 B per pixel; the 32-byte segment of channel tile c of pixel column q sits at segment c ^ swz(q) (the DMA
// permutes its source chunks), which spreads the eight pixels a 32-lane group of ds_read_b64_tr_b16 touches over all 64
// banks (unswizzled: 4-way conflicts).
// Epilogue: the two chunk parities are summed in LDS, then every workgroup adds its 36 864 partial sums to the global
// gradient with coalesced atomics, starting at a different offset per workgroup.
__device__ __forceinline__ int wgrad64_swz(int col) { return ((col >> 1) & 1) | (((col >> 3) & 1) << 1); }

template <typename H, bool FIXED>
__global__ __launch_bounds__(512) void wgrad_halo64_kernel(const WgradHaloParams p, const H* __restrict__ x,
                                                           const H* __restrict__ dy, float* __restrict__ dw,
                                                           float* __restrict__ db) {
    constexpr int ROW = 128;                                 // bytes per pixel, both operands
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // FIXED = 32-pixel rows in bands of 8 (ResNet layer 1, 4 of the step's launches): the band loop below is unrolled with
    // every fragment address a lane constant + an immediate (the run-time version spends 2.3 SALU + 1.6 VALU per MFMA on
    // chunk / row arithmetic: 39 % matrix-pipe occupancy, profiles/r03_wgrad_wg8.md)
    const int W = FIXED ? 32 : p.W, W2 = W + 2, TH = FIXED ? 8 : p.TH;
    const int xs_bytes = (TH + 2) * W2 * ROW, region = xs_bytes + TH * W * ROW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave & 3, half = wave >> 2;
    const int t = lane & 15, g = lane >> 4;
    const uint32_t lds0 = lds_addr_of(lds);
    const int nreg = p.nreg;

    // the two halo columns of every region never receive data: zero them once
    for (int i = tid; i < nreg * (TH + 2) * 2 * 8; i += 512) {
        const int r = i / ((TH + 2) * 16), j = i - r * (TH + 2) * 16;
        const int row = j >> 4, col = (j >> 3) & 1 ? W + 1 : 0, ch = j & 7;
        *reinterpret_cast<uint4*>(lds + (size_t)r * region + (size_t)((row * W2 + col) * ROW + ch * 16)) = make_uint4(0u, 0u, 0u, 0u);
    }
    const eve_int4 rs_x = make_rsrc_words(x, p.x_bytes);
    const eve_int4 rs_dy = make_rsrc_words(dy, p.dy_bytes);

    f32x4_t acc[4][9], accb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        accb[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 9; ++c) acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const uint4 ones = make_uint4(Elem<H>::ONE2, Elem<H>::ONE2, Elem<H>::ONE2, Elem<H>::ONE2);
    const int lrow = 8 * g + (t >> 2), sub = (t & 3) * 8;    // lane-constant part of the transposing reads (see above)
    const int nchunks = TH << p.log2_cpr;
    const bool bias_wave = db != nullptr && ct == 0;
    // fragment byte offsets inside a row, swizzle included: [.][r] = the read of pixels +0 / +4 (x0 is a multiple of 32 and
    // does not reach the swizzle bits)
    int off_d[4][2], off_x[3][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int pd = lrow + 4 * r;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) off_d[mt][r] = pd * ROW + ((mt ^ wgrad64_swz(pd)) << 5) + sub;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int px = kw + lrow + 4 * r;
            off_x[kw][r] = px * ROW + ((ct ^ wgrad64_swz(px)) << 5) + sub;
        }
    }

    // ---- band DMA: (TH + 2) rows of x and TH rows of dy, dealt to the eight waves in 1 KiB pieces (64 slots of one row) ----
    const int lg_gpr = p.log2_cpr + 2;                       // log2(pieces per row) = log2(W * 8 / 64)
    const int total_pieces = (2 * TH + 2) << lg_gpr;
    const int n_dma = (total_pieces - wave + 7) >> 3;        // DMA instructions per band of this wave (out-of-range rows included)
    const uint32_t stride = gridDim.x;
    uint32_t band = blockIdx.x;
    uint32_t k = 0;                                          // bands this workgroup has started: region = k % nreg
    auto region_of = [&](uint32_t kk) { return lds0 + (uint32_t)((int)(kk % (uint32_t)nreg) * region); };
    auto issue_k = [&](uint32_t b, uint32_t kk) {
        const int n = (int)(b / (uint32_t)p.bands), y0 = (int)(b % (uint32_t)p.bands) * TH;
        const uint32_t base = region_of(kk);
        for (int pc = wave; pc < total_pieces; pc += 8) {
            const int row = pc >> lg_gpr, pr = pc & ((1 << lg_gpr) - 1);
            const int s = (pr << 6) + lane, px = s >> 3, ch = s & 7;
            if (row < TH + 2) {
                const int gy = y0 - 1 + row;
                const int src = ((n * p.H + gy) * W + px) * ROW + ((ch ^ (wgrad64_swz(px + 1) << 1)) << 4);
                lds_dma16_asm(rs_x, base + (uint32_t)((row * W2 + 1) * ROW + (pr << 10)), (gy >= 0 && gy < p.H) ? src : EVE_OOB);
            } else {
                const int ty = row - (TH + 2), gy = y0 + ty;
                const int src = ((n * p.H + gy) * W + px) * ROW + ((ch ^ (wgrad64_swz(px) << 1)) << 4);
                lds_dma16_asm(rs_dy, base + (uint32_t)(xs_bytes + ty * W * ROW + (pr << 10)), gy < p.H ? src : EVE_OOB);
            }
        }
    };
    auto wait_all_but = [&](int n) {                          // loads return in order: "at most n outstanding"
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // (never wrong, only early)
        }
    };
    __syncthreads();                                         // (the zeroed columns are in place before any read)
    for (int a = 0; a < nreg - 1; ++a)
        if (band + a * stride < p.total_bands) issue_k(band + a * stride, (uint32_t)a);
    for (; band < p.total_bands; band += stride, ++k) {
        // bands k+1 .. k+nreg-2 may stay in flight; this thread's share of band k has landed ...
        wait_all_but((nreg == 3 && band + stride < p.total_bands) ? n_dma : 0);
        __syncthreads();                                      // ... everyone's has, and band k-1 has been consumed
        if (band + (nreg - 1) * stride < p.total_bands) issue_k(band + (nreg - 1) * stride, k + nreg - 1);
        const uint32_t xs = region_of(k), ds = xs + (uint32_t)xs_bytes;
        if constexpr (FIXED) {
            // chunk i of this wave = image row half + 2 i of the band (one 32-pixel chunk per row)
            const uint32_t dbase = ds + (uint32_t)(half * (32 * ROW)), xbase = xs + (uint32_t)(half * (34 * ROW));
            uint32_t ad[4][2], ax[3][2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) ad[mt][r] = dbase + (uint32_t)off_d[mt][r];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) ax[kw][r] = xbase + (uint32_t)off_x[kw][r];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint4 fp[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const uint2 a0 = lds_tr_read_at(ad[mt][0], i * 2 * 32 * ROW);
                    const uint2 a1 = lds_tr_read_at(ad[mt][1], i * 2 * 32 * ROW);
                    fp[mt] = make_uint4(a0.x, a0.y, a1.x, a1.y);
                }
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    uint4 fq[3];
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const uint2 b0 = lds_tr_read_at(ax[kw][0], (2 * i + kh) * 34 * ROW);
                        const uint2 b1 = lds_tr_read_at(ax[kw][1], (2 * i + kh) * 34 * ROW);
                        fq[kw] = make_uint4(b0.x, b0.y, b1.x, b1.y);
                    }
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        mma3_inplace<H>(acc[mt][kh * 3], acc[mt][kh * 3 + 1], acc[mt][kh * 3 + 2], fp[mt], fq);
                }
                if (bias_wave) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) mma1_inplace<H>(accb[mt], fp[mt], ones);
                }
            }
        } else
        for (int c = half; c < nchunks; c += 2) {
            const int ty = c >> p.log2_cpr, x0 = (c - (ty << p.log2_cpr)) * 32;
            uint4 fp[4];
            const uint32_t pa = ds + (uint32_t)((ty * W + x0) * ROW);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const uint2 a0 = lds_tr_read(pa + off_d[mt][0]);
                const uint2 a1 = lds_tr_read(pa + off_d[mt][1]);
                fp[mt] = make_uint4(a0.x, a0.y, a1.x, a1.y);
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                uint4 fq[3];
                const uint32_t qa = xs + (uint32_t)(((ty + kh) * W2 + x0) * ROW);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const uint2 b0 = lds_tr_read(qa + off_x[kw][0]);
                    const uint2 b1 = lds_tr_read(qa + off_x[kw][1]);
                    fq[kw] = make_uint4(b0.x, b0.y, b1.x, b1.y);
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    mma3_inplace<H>(acc[mt][kh * 3], acc[mt][kh * 3 + 1], acc[mt][kh * 3 + 2], fp[mt], fq);
            }
            if (bias_wave) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) mma1_inplace<H>(accb[mt], fp[mt], ones);
            }
        }
    }
    mma_drain();
    // ---- epilogue: chunk parity 0 stores its tiles to LDS, parity 1 adds its own, then rotated coalesced global atomics ----
    constexpr int NRED = 9 * 64 * 64;
    float* red = reinterpret_cast<float*>(lds);
    __syncthreads();                                         // the last band has been consumed
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        if (half == hh) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = mt * 16 + g * 4 + r, ci = ct * 16 + t;
                        float* q = red + (co * 9 + tap) * 64 + ci;
                        *q = hh ? *q + acc[mt][tap][r] : acc[mt][tap][r];
                    }
                    __builtin_amdgcn_sched_barrier(0);       // (else all 36 tiles leave the AGPRs at once)
                }
            if (bias_wave && t == 0) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* q = red + NRED + mt * 16 + g * 4 + r;
                        *q = hh ? *q + accb[mt][r] : accb[mt][r];
                    }
            }
        }
        __syncthreads();
    }
    const int rot = (int)((blockIdx.x * 37u) % 72u) * 512;   // NRED = 72 x 512: every workgroup starts elsewhere
    for (int i = 0; i < 72; ++i) {
        int idx = i * 512 + rot + tid;
        idx = idx >= NRED ? idx - NRED : idx;
        atomicAdd(dw + idx, red[idx]);
    }
    if (db && tid < 64) atomicAdd(db + tid, red[NRED + tid]);
}

}  // namespace eve
