// Second-generation conv kernels for gfx950: operands go global -> LDS by LDS-DMA
// (buffer_load ... lds, 16 B per lane, zero-fill for out-of-range lanes), never through VGPRs.
//
// Why: with register staging the 128x128x64 tile writes 32 KB per K step through ds_write_b128
// (~79 B/clk/CU -> ~415 LDS clocks) on top of 256 clocks of fragment reads, against 512 MFMA clocks:
// the v1 kernel is LDS-pipe bound before anything else.  LDS-DMA bypasses the VGPR->LDS store path,
// frees the staging registers and costs one VALU select per 16 bytes.
//
//  * igemm_dma_kernel  forward / stride-1 dgrad when Cin % BK == 0: the filter tap is uniform per K step,
//                      so the K loop does scalar tap arithmetic plus, per row, one mask test and one add.
//  * wgrad_tr_kernel   bf16 weight gradient: tiles stay in their natural [pixel][channel] layout and the
//                      K(=pixel)-major MFMA fragments come out of ds_read_b64_tr_b16 (hardware transpose;
//                      semantics verified by tools/probe_isa.hip: result(i,j) = loaded(16g+4j+i/4, i%4)).
// LDS-DMA writes lane-linearly, so bank-conflict swizzles are applied to the SOURCE slot a lane fetches
// and undone on the read (same XOR on both sides).
#pragma once
#include <type_traits>
#include "common.h"
#include "lds_dma.h"

namespace eve {

// The filter taps a launch iterates over (at most 32), as source-pixel displacements and weight tap ids, plus
// the mapping from the launch's pixel grid to output pixels.  A plain convolution uses all KH*KW taps and the
// identity mapping; the data gradient of a stride-s convolution is s*s launches, one per output parity class,
// each a dense stride-1 problem over only the taps that are divisible for that class (4x fewer MFMAs at s=2).
struct TapPlan {
    int ntaps;
    signed char dy[32], dx[32], wt[32];
    int osy, oy0, osx, ox0, OHf, OWf;     // out pixel = (y'*osy + oy0, x'*osx + ox0) in an OHf x OWf image
};

// =================================================================================================
template <typename T, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void igemm_dma_kernel(const GatherParams p, const T* __restrict__ src,
                                                        const T* __restrict__ w, const float* __restrict__ bias,
                                                        const int epi_act, T* __restrict__ out,
                                                        const uint32_t src_bytes, const uint32_t w_bytes,
                                                        const TapPlan tp) {
    constexpr int VEC = Elem<T>::VEC, ES = (int)sizeof(T);
    constexpr int BM = 64 * WM, BN = 64 * WN, BK = 8 * VEC;
    constexpr int NTHR = 64 * WM * WN, RPP = NTHR / 8;       // threads, tile rows covered per DMA pass
    constexpr int A_DMA = BM / RPP, B_DMA = BN / RPP;
    __shared__ uint4 lds[2 * (BM + BN) * 8];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t tiles_n = (p.Cout + BN - 1) / BN;
    const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t m0 = (lid / tiles_n) * BM;
    const uint32_t n0 = (lid % tiles_n) * BN;

    __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, w_bytes, 0x00020000);

    const int v = tid & 7, r0 = tid >> 3;
    const int vs = v ^ (r0 & 7);                  // source slot (the LDS image is lane-linear)
    int a_off[A_DMA], a_y0[A_DMA], a_x0[A_DMA];
    uint32_t a_mask[A_DMA];
    bool a_ok[A_DMA];
#pragma unroll
    for (int j = 0; j < A_DMA; ++j) {
        const uint32_t m = m0 + r0 + RPP * j;
        a_ok[j] = m < p.M;
        const uint32_t mm = a_ok[j] ? m : 0;
        const uint32_t n = fd_div(mm, p.fd_ohw);
        const uint32_t rem = mm - n * (uint32_t)(p.OH * p.OW);
        const uint32_t oy = fd_div(rem, p.fd_ow);
        const uint32_t ox = rem - oy * (uint32_t)p.OW;
        a_y0[j] = (int)oy * p.o_mul + p.off;
        a_x0[j] = (int)ox * p.o_mul + p.off;
        a_off[j] = (((int)n * p.IH + a_y0[j]) * p.IW + a_x0[j]) * p.Cin * ES + vs * 16;
        a_mask[j] = 0;
    }
#pragma unroll 1
    for (int t = 0; t < tp.ntaps; ++t) {            // one kernarg fetch per tap, all rows per fetch
        const int ty = tp.dy[t], tx = tp.dx[t];
#pragma unroll
        for (int j = 0; j < A_DMA; ++j) {
            const int sy = a_y0[j] + ty, sx = a_x0[j] + tx;
            a_mask[j] |= (uint32_t)(a_ok[j] && sy >= 0 && sy < p.IH && sx >= 0 && sx < p.IW) << t;
        }
    }
    int b_off[B_DMA];
#pragma unroll
    for (int j = 0; j < B_DMA; ++j) {
        const uint32_t co = n0 + r0 + RPP * j;
        b_off[j] = co < (uint32_t)p.Cout ? (int)(co * (uint32_t)p.K) * ES + vs * 16 : EVE_OOB;
    }

    // scalar tap state: the tap's byte displacements are fetched one tap ahead of their use, so the kernarg
    // load latency never sits between the barrier and the DMA issue
    const int pix_bytes = p.Cin * ES;
    int tap = 0, ci0 = 0;
    int d_cur = ((int)tp.dy[0] * p.IW + (int)tp.dx[0]) * pix_bytes, w_cur = (int)tp.wt[0] * pix_bytes;
    int tn = tp.ntaps > 1 ? 1 : 0;
    int d_nxt = ((int)tp.dy[tn] * p.IW + (int)tp.dx[tn]) * pix_bytes, w_nxt = (int)tp.wt[tn] * pix_bytes;

    auto issue = [&](int buf) {
        const int delta = d_cur + ci0 * ES, wk = w_cur + ci0 * ES;
        const uint4* base = lds + buf * (BM + BN) * 8 + wave * 64;
#pragma unroll
        for (int j = 0; j < A_DMA; ++j) {
            const int voff = ((a_mask[j] >> tap) & 1u) ? a_off[j] + delta : EVE_OOB;
            lds_dma16(rs_src, base + j * NTHR, voff);
        }
#pragma unroll
        for (int j = 0; j < B_DMA; ++j) {
            const int voff = b_off[j] == EVE_OOB ? EVE_OOB : b_off[j] + wk;
            lds_dma16(rs_w, base + BM * 8 + j * NTHR, voff);
        }
        ci0 += BK;                                   // advance to the next K step (uniform)
        if (ci0 == p.Cin) {
            ci0 = 0;
            ++tap;
            d_cur = d_nxt; w_cur = w_nxt;
            tn = tap + 1 < tp.ntaps ? tap + 1 : tap;
            d_nxt = ((int)tp.dy[tn] * p.IW + (int)tp.dx[tn]) * pix_bytes;
            w_nxt = (int)tp.wt[tn] * pix_bytes;
        }
    };

    const int lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lg = lane >> 4;

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // float32 (the 1e-4 rad parity mode): the accumulation chain is cut every FLUSH K tiles (256 products) into a second
    // accumulator set, as a blocked GEMM does -- see igemm_kernel (conv_igemm.hip) and profiles/r06_notes.md 13
    constexpr bool SPLIT_SUM = std::is_same<T, float>::value;
    constexpr int FLUSH = 8;
    f32x4_t tot[SPLIT_SUM ? 4 : 1][SPLIT_SUM ? 4 : 1];
    if (SPLIT_SUM) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) tot[SPLIT_SUM ? a : 0][SPLIT_SUM ? b : 0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const int nk = tp.ntaps * p.Cin / BK;
    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // tile kt has landed for every wave; tile kt-1 fully consumed
        if (kt + 1 < nk) issue(cur ^ 1);
        if (SPLIT_SUM && kt > 0 && (kt % FLUSH) == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    tot[SPLIT_SUM ? a : 0][SPLIT_SUM ? b : 0] += acc[a][b];
                    acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
        }
        const uint4* la = lds + cur * (BM + BN) * 8;
        const uint4* lb = la + BM * 8;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int vc = c * 4 + lg;
            uint4 fx[4], fw[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = wm * 64 + mt * 16 + li;
                fx[mt] = la[row * 8 + (vc ^ (row & 7))];
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int row = wn * 64 + nt * 16 + li;
                fw[nt] = lb[row * 8 + (vc ^ (row & 7))];
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) Mma<T>::run(acc[mt][nt], fw[nt], fx[mt]);
        }
    }

    if (SPLIT_SUM) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] += tot[SPLIT_SUM ? a : 0][SPLIT_SUM ? b : 0];
    }
    const bool vec_ok = (p.Cout & 3) == 0;
    auto epilogue = [&](auto fast) {
    #pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const uint32_t co = n0 + wn * 64 + nt * 16 + lg * 4;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
    #pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < (uint32_t)p.Cout) bv[r] = bias[co + r];
            }
    #pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const uint32_t m = m0 + wm * 64 + mt * 16 + li;
                if (m >= p.M || co >= (uint32_t)p.Cout) continue;
                float o[4];
    #pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = acc[mt][nt][r] + bv[r];
                act_fwd4<decltype(fast)::value>(o, epi_act);
                size_t opix = m;
                if (tp.osy != 1 || tp.osx != 1) {       // uniform: only the strided-dgrad sub-problems remap pixels
                    const uint32_t n = fd_div(m, p.fd_ohw);
                    const uint32_t rem = m - n * (uint32_t)(p.OH * p.OW);
                    const uint32_t oy = fd_div(rem, p.fd_ow);
                    const uint32_t ox = rem - oy * (uint32_t)p.OW;
                    opix = ((size_t)n * tp.OHf + oy * tp.osy + tp.oy0) * tp.OWf + ox * tp.osx + tp.ox0;
                }
                T* dst = out + opix * p.Cout + co;
                if (epi_act & EVE_EPI_ACC) {
    #pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < (uint32_t)p.Cout) o[r] += Elem<T>::ld(dst + r);
                }
                if (vec_ok) {
                    if (sizeof(T) == 4) {
                        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                    } else {
                        uint2 pk;
                        pk.x = Elem<T>::pack2(o[0], o[1]);
                        pk.y = Elem<T>::pack2(o[2], o[3]);
                        *reinterpret_cast<uint2*>(dst) = pk;
                    }
                } else {
    #pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < (uint32_t)p.Cout) Elem<T>::st(dst + r, o[r]);
                }
            }
        }
    };
    if (act_is_fast(epi_act)) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

// 16 MFMAs (4 x 4 accumulator tiles, one K=32 chunk) as ONE asm statement with every accumulator tied in place
// ("+a": AGPR, D == C).  Left to itself hipcc ping-pongs loop-carried accumulators between two register sets
// when each gets a single MFMA per trip, and copies them back with v_accvgpr_mov/read/write at the loop edge
// (10 VALU per MFMA measured in the weight-gradient loop).  The leading s_nop covers a VALU-assembled operand
// tuple; consecutive MFMAs here never share an accumulator.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ void mma16_bf16_inplace(f32x4_t (&acc)[4][4], const uint4 (&a4)[4], const uint4 (&b4)[4]) {
    u32x4_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = __builtin_bit_cast(u32x4_t, a4[i]); b[i] = __builtin_bit_cast(u32x4_t, b4[i]); }
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_bf16 %0, %16, %20, %0\n\t"
        "v_mfma_f32_16x16x32_bf16 %1, %16, %21, %1\n\t"
        "v_mfma_f32_16x16x32_bf16 %2, %16, %22, %2\n\t"
        "v_mfma_f32_16x16x32_bf16 %3, %16, %23, %3\n\t"
        "v_mfma_f32_16x16x32_bf16 %4, %17, %20, %4\n\t"
        "v_mfma_f32_16x16x32_bf16 %5, %17, %21, %5\n\t"
        "v_mfma_f32_16x16x32_bf16 %6, %17, %22, %6\n\t"
        "v_mfma_f32_16x16x32_bf16 %7, %17, %23, %7\n\t"
        "v_mfma_f32_16x16x32_bf16 %8, %18, %20, %8\n\t"
        "v_mfma_f32_16x16x32_bf16 %9, %18, %21, %9\n\t"
        "v_mfma_f32_16x16x32_bf16 %10, %18, %22, %10\n\t"
        "v_mfma_f32_16x16x32_bf16 %11, %18, %23, %11\n\t"
        "v_mfma_f32_16x16x32_bf16 %12, %19, %20, %12\n\t"
        "v_mfma_f32_16x16x32_bf16 %13, %19, %21, %13\n\t"
        "v_mfma_f32_16x16x32_bf16 %14, %19, %22, %14\n\t"
        "v_mfma_f32_16x16x32_bf16 %15, %19, %23, %15"
        : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]),
          "+a"(acc[1][2]), "+a"(acc[1][3]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]),
          "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]), "+a"(acc[3][3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
__device__ __forceinline__ void mma16_f16_inplace(f32x4_t (&acc)[4][4], const uint4 (&a4)[4], const uint4 (&b4)[4]) {
    u32x4_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = __builtin_bit_cast(u32x4_t, a4[i]); b[i] = __builtin_bit_cast(u32x4_t, b4[i]); }
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_f16 %0, %16, %20, %0\n\t"
        "v_mfma_f32_16x16x32_f16 %1, %16, %21, %1\n\t"
        "v_mfma_f32_16x16x32_f16 %2, %16, %22, %2\n\t"
        "v_mfma_f32_16x16x32_f16 %3, %16, %23, %3\n\t"
        "v_mfma_f32_16x16x32_f16 %4, %17, %20, %4\n\t"
        "v_mfma_f32_16x16x32_f16 %5, %17, %21, %5\n\t"
        "v_mfma_f32_16x16x32_f16 %6, %17, %22, %6\n\t"
        "v_mfma_f32_16x16x32_f16 %7, %17, %23, %7\n\t"
        "v_mfma_f32_16x16x32_f16 %8, %18, %20, %8\n\t"
        "v_mfma_f32_16x16x32_f16 %9, %18, %21, %9\n\t"
        "v_mfma_f32_16x16x32_f16 %10, %18, %22, %10\n\t"
        "v_mfma_f32_16x16x32_f16 %11, %18, %23, %11\n\t"
        "v_mfma_f32_16x16x32_f16 %12, %19, %20, %12\n\t"
        "v_mfma_f32_16x16x32_f16 %13, %19, %21, %13\n\t"
        "v_mfma_f32_16x16x32_f16 %14, %19, %22, %14\n\t"
        "v_mfma_f32_16x16x32_f16 %15, %19, %23, %15"
        : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]),
          "+a"(acc[1][2]), "+a"(acc[1][3]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]),
          "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]), "+a"(acc[3][3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
// wait states between the last asm MFMA and compiler-generated reads of the accumulators
__device__ __forceinline__ void mma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
// four in-place MFMAs sharing the A operand: c[i] += a x b[i]
__device__ __forceinline__ void mma4_bf16_inplace(f32x4_t& c0, f32x4_t& c1, f32x4_t& c2, f32x4_t& c3, const uint4& a4,
                                                  const uint4 (&b4)[4]) {
    u32x4_t b[4];
    const u32x4_t a = __builtin_bit_cast(u32x4_t, a4);
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(u32x4_t, b4[i]);
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\t"
        "v_mfma_f32_16x16x32_bf16 %1, %4, %6, %1\n\t"
        "v_mfma_f32_16x16x32_bf16 %2, %4, %7, %2\n\t"
        "v_mfma_f32_16x16x32_bf16 %3, %4, %8, %3"
        : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3)
        : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}

// four in-place MFMAs sharing the B operand: c[i] += a[i] x b
__device__ __forceinline__ void mma4_bf16_inplace_b(f32x4_t (&c)[4], const uint4 (&a4)[4], const uint4& b4) {
    u32x4_t a[4];
    const u32x4_t b = __builtin_bit_cast(u32x4_t, b4);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(u32x4_t, a4[i]);
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_bf16 %0, %4, %8, %0\n\t"
        "v_mfma_f32_16x16x32_bf16 %1, %5, %8, %1\n\t"
        "v_mfma_f32_16x16x32_bf16 %2, %6, %8, %2\n\t"
        "v_mfma_f32_16x16x32_bf16 %3, %7, %8, %3"
        : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b));
}

// four in-place MFMAs sharing the A operand: c[i] += a x b[i]
__device__ __forceinline__ void mma4_f16_inplace(f32x4_t& c0, f32x4_t& c1, f32x4_t& c2, f32x4_t& c3, const uint4& a4,
                                                  const uint4 (&b4)[4]) {
    u32x4_t b[4];
    const u32x4_t a = __builtin_bit_cast(u32x4_t, a4);
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(u32x4_t, b4[i]);
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n\t"
        "v_mfma_f32_16x16x32_f16 %1, %4, %6, %1\n\t"
        "v_mfma_f32_16x16x32_f16 %2, %4, %7, %2\n\t"
        "v_mfma_f32_16x16x32_f16 %3, %4, %8, %3"
        : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3)
        : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}

// four in-place MFMAs sharing the B operand: c[i] += a[i] x b
__device__ __forceinline__ void mma4_f16_inplace_b(f32x4_t (&c)[4], const uint4 (&a4)[4], const uint4& b4) {
    u32x4_t a[4];
    const u32x4_t b = __builtin_bit_cast(u32x4_t, b4);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(u32x4_t, a4[i]);
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_f16 %0, %4, %8, %0\n\t"
        "v_mfma_f32_16x16x32_f16 %1, %5, %8, %1\n\t"
        "v_mfma_f32_16x16x32_f16 %2, %6, %8, %2\n\t"
        "v_mfma_f32_16x16x32_f16 %3, %7, %8, %3"
        : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b));
}

// format-generic entry points of the in-place groups
template <typename H>
__device__ __forceinline__ void mma16_inplace(f32x4_t (&acc)[4][4], const uint4 (&a4)[4], const uint4 (&b4)[4]) {
    if constexpr (Elem<H>::IS_BF16) mma16_bf16_inplace(acc, a4, b4); else mma16_f16_inplace(acc, a4, b4);
}
template <typename H>
__device__ __forceinline__ void mma4_inplace(f32x4_t& c0, f32x4_t& c1, f32x4_t& c2, f32x4_t& c3, const uint4& a4, const uint4 (&b4)[4]) {
    if constexpr (Elem<H>::IS_BF16) mma4_bf16_inplace(c0, c1, c2, c3, a4, b4); else mma4_f16_inplace(c0, c1, c2, c3, a4, b4);
}
template <typename H>
__device__ __forceinline__ void mma4_inplace_b(f32x4_t (&c)[4], const uint4 (&a4)[4], const uint4& b4) {
    if constexpr (Elem<H>::IS_BF16) mma4_bf16_inplace_b(c, a4, b4); else mma4_f16_inplace_b(c, a4, b4);
}

// =================================================================================================
// bf16 weight gradient.  Block tile = (64*WCO output channels) x (64*WK filter-K values); every wave owns
// a 64 x 64 piece; each step consumes 64 pixels (two MFMA K=32 chunks).
// =================================================================================================
// rows of 128 B (mod 256) alternate between the two halves of the 64 banks, so two row bits separate the four rows
// a 32-lane read group touches in one half; rows that are a multiple of 256 B all start on bank 0 and need three
template <int ROWB>
__device__ __forceinline__ int tr_key(int row) {
    const int b0 = row & 1, b1 = (row >> 1) & 1, b2 = (row >> 3) & 1;
    return ROWB % 256 == 128 ? (b1 | (b2 << 1)) : (b0 | (b1 << 1) | (b2 << 2));
}

// takes a 32-bit LDS byte address: an integer -> LDS pointer cast is free, a generic -> LDS cast is ~6 VALU per read
// (the compile-time displacement goes through pointer arithmetic so that it folds into the instruction's offset field)
template <int IMM = 0>
__device__ __forceinline__ uint2 lds_tr_read(uint32_t lds_byte_addr) {
    EVE_LDS char* base = (EVE_LDS char*)(size_t)lds_byte_addr;
    bf16x4v_t r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((EVE_LDS bf16x4v_t*)(base + IMM));
    return __builtin_bit_cast(uint2, r);
}

// (run-time displacement that is a constant after unrolling: same pointer arithmetic, folds the same way)
__device__ __forceinline__ uint2 lds_tr_read_at(uint32_t lds_byte_addr, int disp) {
    EVE_LDS char* base = (EVE_LDS char*)(size_t)lds_byte_addr;
    bf16x4v_t r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((EVE_LDS bf16x4v_t*)(base + disp));
    return __builtin_bit_cast(uint2, r);
}

// MODE: 0 = any output size (mul-hi divisions per slot), 1 = OH and OW powers of two, 2 = OW a power of two only
// BIAS: the waves of the first K tile also accumulate db[co] += sum over pixels of dy -- one more MFMA per channel
// tile against an all-ones operand, on fragments that are in registers anyway (the separate column-sum pass re-read
// dy from HBM: 44 launches and 8.4 ms per RefineNet step)
// MT: 16-channel tiles of dy a wave multiplies (4 = all 64; 1 / 2 for layers with <= 16 / 32 output channels, whose
// remaining tiles are zero-fill: RefineNet's outer levels spent 3/4 of their MFMAs on them)
template <typename H, int WCO, int WK, int MODE, bool BIAS = false, int MT = 4>
__global__ __launch_bounds__(64 * WCO * WK) void wgrad_tr_kernel(const GatherParams p, const H* __restrict__ x,
                                                       const H* __restrict__ dy, float* __restrict__ dw,
                                                       const uint32_t rows_per_split, const uint32_t x_bytes,
                                                       const uint32_t dy_bytes, float* __restrict__ db) {
    constexpr int BCO = 64 * WCO, BKK = 64 * WK;
    constexpr int PROW = BCO * 2, QROW = BKK * 2;            // bytes per pixel row
    constexpr int PSL = PROW / 16, QSL = QROW / 16;          // 16-byte slots per row
    constexpr int STEP = 32;                                 // pixels per stage = one MFMA K=32 chunk
    constexpr int NT = 64 * WCO * WK;                        // 256 threads, or 448 / 576 for the one-K-tile variants
    constexpr int P_SLOTS = STEP * PSL, Q_SLOTS = STEP * QSL;
    // (a tile with fewer slots than threads is fetched again by the surplus waves: same data to the same place)
    constexpr int P_DMA = (P_SLOTS + NT - 1) / NT, Q_DMA = (Q_SLOTS + NT - 1) / NT;
    constexpr int NDMA = P_DMA + Q_DMA;                      // LDS-DMA instructions per thread and stage
    constexpr int BUF = STEP * (PROW + QROW);                // bytes per stage (16 KB / 20 KB)
    // 128x128 tiles: 3 stages of 16 KB -> three workgroups per CU (after the address-arithmetic diet this beats the
    // 4-deep ring with two workgroups by 7-9 % on layers 3/4); 64x256 tiles: 4 stages of 20 KB, two workgroups
    // 64 x 256 tiles: three stages measured 9 % faster than four on RefineNet's planes (MODE 2: 2.19 -> 2.0 ms per configs[2]
    // step) and 4 % slower on the stem's weight gradient (MODE 1); both depths leave two workgroups per CU
    // (tools/probes/occupancy.hip)
    constexpr int RING = (WCO == 2 || (WK == 4 && MODE == 2)) ? 3 : 4;
    static_assert(P_SLOTS % 64 == 0 && Q_SLOTS % 64 == 0 && NT % 64 == 0, "wave-granular slot wrap-around");
    extern __shared__ __attribute__((aligned(16))) char lds[];   // RING * BUF bytes

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, XCD-aware: all (k tile, channel tile) workgroups of one pixel range land on the SAME XCD, so the
    // dy / x rows they share are fetched into one L2 once (a 3-D grid spreads them round-robin over the 8 XCDs and
    // every L2 re-fetches them: 3-6x the algorithmic bytes measured with FETCH_SIZE)
    const uint32_t tk = (p.K + BKK - 1) / BKK, tc = (p.Cout + BCO - 1) / BCO;
    // (with fewer than 4 tiles per pixel range the plain order measured faster: nothing to share)
    const uint32_t lid = tk * tc >= 4 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t k0 = (lid % tk) * BKK, co0 = ((lid / tk) % tc) * BCO;
    const uint32_t m_begin = (lid / (tk * tc)) * rows_per_split;
    const uint32_t m_end = min(p.M, m_begin + rows_per_split);
    const uint32_t ohw = (uint32_t)(p.OH * p.OW);

    const eve_int4 rs_x = make_rsrc_words(x, x_bytes);
    const eve_int4 rs_dy = make_rsrc_words(dy, dy_bytes);
    const uint32_t lds0 = lds_addr_of(lds);

    // ---- loop-invariant slot coordinates ----
    int p_row[P_DMA], p_col[P_DMA];                 // row in the stage, byte offset of the channel in dy's row
#pragma unroll
    for (int j = 0; j < P_DMA; ++j) {
        const int q = (tid + NT * j) % P_SLOTS;
        const int row = q / PSL, sl = q % PSL;
        const int sg = sl ^ (tr_key<PROW>(row) << 1);
        const uint32_t co = co0 + sg * 8;
        p_row[j] = row;
        p_col[j] = co < (uint32_t)p.Cout ? (int)co * 2 : EVE_OOB;
    }
    int q_row[Q_DMA], q_col[Q_DMA], q_dy[Q_DMA], q_dx[Q_DMA];
#pragma unroll
    for (int j = 0; j < Q_DMA; ++j) {
        const int q = (tid + NT * j) % Q_SLOTS;
        const int row = q / QSL, sl = q % QSL;
        const int sg = sl ^ (tr_key<QROW>(row) << 1);
        const uint32_t k = k0 + sg * 8;
        q_row[j] = row;
        if (k < (uint32_t)p.K) {
            const uint32_t tap = fd_div(k, p.fd_cin);
            const uint32_t kh = fd_div(tap, p.fd_kw);
            q_col[j] = (int)(k - tap * (uint32_t)p.Cin) * 2;
            q_dy[j] = (int)kh * p.k_mul + p.off;
            q_dx[j] = (int)(tap - kh * (uint32_t)p.KW) * p.k_mul + p.off;
        } else {
            q_col[j] = EVE_OOB; q_dy[j] = 0; q_dx[j] = 0;
        }
    }

    // Source offsets of one stage.  Pixels past the end of the split get the out-of-range offset (zero fill).
    // Power-of-two output sizes (every EyeNet stage): a stage starts at a multiple of 32 pixels and a slot's row is
    // < 32, so pixel = mbase | row and every decoded field (n, oy, ox) is the SUM of a wave-uniform part (from
    // mbase: scalar ALU, once per stage) and a lane constant (from row: computed once).  A slot then costs one add
    // for the address and an add + unsigned compare per bounds test, instead of a full decode (the kernel was
    // issuing 4-6 VALU instructions per MFMA).  Other sizes use the mul-hi division per slot.
    const int sh_w = __builtin_ctz((unsigned)p.OW), sh_hw = __builtin_ctz((unsigned)(p.OH * p.OW));
    const int cin2 = p.Cin * 2, cout2 = p.Cout * 2;
    int p_const[P_DMA], q_const[Q_DMA], q_ty[Q_DMA], q_tx[Q_DMA], q_ry[Q_DMA];
#pragma unroll
    for (int j = 0; j < P_DMA; ++j) p_const[j] = p_col[j] == EVE_OOB ? EVE_OOB : p_row[j] * cout2 + p_col[j];
#pragma unroll
    for (int j = 0; j < Q_DMA; ++j) {
        const uint32_t r = (uint32_t)q_row[j];
        // MODE 2 keeps the row offset whole (the wrap into the next image is resolved per stage)
        const int n_t = MODE == 2 ? 0 : (int)(r >> sh_hw);
        const int oy_t = MODE == 2 ? (int)(r >> sh_w) : (int)((r >> sh_w) & (uint32_t)(p.OH - 1)), ox_t = (int)(r & (uint32_t)(p.OW - 1));
        q_ry[j] = oy_t;
        q_ty[j] = oy_t * p.o_mul + q_dy[j];
        q_tx[j] = ox_t * p.o_mul + q_dx[j];
        q_const[j] = q_col[j] == EVE_OOB ? EVE_OOB : ((n_t * p.IH + q_ty[j]) * p.IW + q_tx[j]) * cin2 + q_col[j];
    }
    // ---- MODE 1, round 6: validity as bit masks.  SQ counters of wgrad_tr_kernel<., 2, 2, 1> on ResNet layer 2 (profiles/r06_notes.md):
    // 3.7 VALU + 2.8 SALU per 16-cycle MFMA -- the SIMDs' issue slots, not the matrix pipe (38 % busy), set its time, and most of
    // those instructions decide, per DMA slot and stage, whether the gathered pixel lies inside the image.  With power-of-two
    // sizes and a padding that does not exceed the stride, a stage (32 consecutive output pixels = part of a row, whole rows or
    // whole images) can only leave the image at the image's FIRST / LAST rows or columns, and whether THIS lane's pixel does
    // there is a lane constant: bit 0 / 1 = outside when the stage holds the image's top / bottom rows, bit 2 / 3 = outside when
    // it holds its left / right columns, bit 4 = a filter column beyond K (always outside).  Per stage the scalar side forms the
    // same five flags from (oy_s, ox_s); a slot is valid iff (bits & flags) == 0: v_and + v_cmp + v_add + v_cndmask per slot,
    // no condition arithmetic on the scalar unit.
    const int c_x = p.OW < STEP ? p.OW : STEP;                                   // columns of a row one stage covers
    const int r_y = p.OH * p.OW <= STEP ? p.OH : (p.OW >= STEP ? 1 : STEP / p.OW);   // rows of an image one stage covers
    const bool bits_ok = MODE == 1 && -p.off <= p.o_mul && (p.KH - 1) * p.k_mul + p.off <= p.o_mul &&
                         (p.KW - 1) * p.k_mul + p.off <= p.o_mul;                // (uniform: padding / filter reach <= stride)
    int p_bits[P_DMA], q_bits[Q_DMA];
#pragma unroll
    for (int j = 0; j < P_DMA; ++j) p_bits[j] = p_col[j] == EVE_OOB ? 16 : 0;
#pragma unroll
    for (int j = 0; j < Q_DMA; ++j) {
        const uint32_t r = (uint32_t)q_row[j];
        const int oy_t = (int)((r >> sh_w) & (uint32_t)(p.OH - 1)), ox_t = (int)(r & (uint32_t)(p.OW - 1));
        int b = q_col[j] == EVE_OOB ? 16 : 0;
        b |= (oy_t * p.o_mul + q_dy[j] < 0) ? 1 : 0;
        b |= ((p.OH - r_y + oy_t) * p.o_mul + q_dy[j] >= p.IH) ? 2 : 0;
        b |= (ox_t * p.o_mul + q_dx[j] < 0) ? 4 : 0;
        b |= ((p.OW - c_x + ox_t) * p.o_mul + q_dx[j] >= p.IW) ? 8 : 0;
        q_bits[j] = b;
    }
    auto offsets = [&](uint32_t mbase, int* vp, int* vq) {   // branch-free: selects only
        if (MODE == 1) {
            const uint32_t left = m_end > mbase ? m_end - mbase : 0u;          // rows of this stage still in range
            const int pbase = (int)mbase * cout2;
            const uint32_t n_s = mbase >> sh_hw, oy_s = (mbase >> sh_w) & (uint32_t)(p.OH - 1), ox_s = mbase & (uint32_t)(p.OW - 1);
            const int sy0 = (int)oy_s * p.o_mul, sx0 = (int)ox_s * p.o_mul;
            const int qbase = (((int)n_s * p.IH + sy0) * p.IW + sx0) * cin2;
            if (bits_ok && left >= (uint32_t)STEP) {                            // (uniform) every row of the stage is in range
                const int flags = 16 | (oy_s == 0u ? 1 : 0) | (oy_s == (uint32_t)(p.OH - r_y) ? 2 : 0) | (ox_s == 0u ? 4 : 0) |
                                  (ox_s == (uint32_t)(p.OW - c_x) ? 8 : 0);
#pragma unroll
                for (int j = 0; j < P_DMA; ++j) vp[j] = (p_bits[j] & flags) == 0 ? pbase + p_const[j] : EVE_OOB;
#pragma unroll
                for (int j = 0; j < Q_DMA; ++j) vq[j] = (q_bits[j] & flags) == 0 ? qbase + q_const[j] : EVE_OOB;
                return;
            }
#pragma unroll
            for (int j = 0; j < P_DMA; ++j)
                vp[j] = ((uint32_t)p_row[j] < left) & (p_const[j] != EVE_OOB) ? pbase + p_const[j] : EVE_OOB;
#pragma unroll
            for (int j = 0; j < Q_DMA; ++j) {
                const bool ok = ((uint32_t)q_row[j] < left) & (q_const[j] != EVE_OOB) &
                                ((uint32_t)(sy0 + q_ty[j]) < (uint32_t)p.IH) & ((uint32_t)(sx0 + q_tx[j]) < (uint32_t)p.IW);
                vq[j] = ok ? qbase + q_const[j] : EVE_OOB;
            }
            return;
        }
        if (MODE == 2) {
            // only the width is a power of two (RefineNet: 72x128 ... 5x8): the column still splits without carries, the
            // (image, row) pair of the stage's first pixel costs ONE scalar division, and a lane whose row offset runs past
            // the last image row wraps into the next image (at most once: a stage is 32 pixels <= one image)
            const uint32_t left = m_end > mbase ? m_end - mbase : 0u;
            const int pbase = (int)mbase * cout2;
            const uint32_t rowq = mbase >> sh_w;
            const uint32_t n_s = fd_div(rowq, p.fd_oh), oy_s = rowq - n_s * (uint32_t)p.OH, ox_s = mbase & (uint32_t)(p.OW - 1);
            const int sy0 = (int)oy_s * p.o_mul, sx0 = (int)ox_s * p.o_mul;
            const int qbase = (((int)n_s * p.IH + sy0) * p.IW + sx0) * cin2;
            const int wrap_y = p.OH * p.o_mul, wrap_addr = (p.IH - wrap_y) * p.IW * cin2;
#pragma unroll
            for (int j = 0; j < P_DMA; ++j)
                vp[j] = ((uint32_t)p_row[j] < left) & (p_const[j] != EVE_OOB) ? pbase + p_const[j] : EVE_OOB;
#pragma unroll
            for (int j = 0; j < Q_DMA; ++j) {
                const bool wrap = (int)oy_s + q_ry[j] >= p.OH;
                const int sy = sy0 + q_ty[j] - (wrap ? wrap_y : 0);
                const bool ok = ((uint32_t)q_row[j] < left) & (q_const[j] != EVE_OOB) &
                                ((uint32_t)sy < (uint32_t)p.IH) & ((uint32_t)(sx0 + q_tx[j]) < (uint32_t)p.IW);
                vq[j] = ok ? qbase + q_const[j] + (wrap ? wrap_addr : 0) : EVE_OOB;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < P_DMA; ++j) {
            const uint32_t m = mbase + p_row[j];
            const int v = (int)m * cout2 + p_col[j];
            vp[j] = (m < m_end) & (p_col[j] != EVE_OOB) ? v : EVE_OOB;
        }
#pragma unroll
        for (int j = 0; j < Q_DMA; ++j) {
            const uint32_t m = mbase + q_row[j];
            const uint32_t n = fd_div(m, p.fd_ohw);
            const uint32_t rem = m - n * ohw;
            const uint32_t oy = fd_div(rem, p.fd_ow), ox = rem - oy * (uint32_t)p.OW;
            const int sy = (int)oy * p.o_mul + q_dy[j], sx = (int)ox * p.o_mul + q_dx[j];
            const bool ok = (m < m_end) & (q_col[j] != EVE_OOB) & (sy >= 0) & (sy < p.IH) & (sx >= 0) & (sx < p.IW);
            const int v = (((int)n * p.IH + sy) * p.IW + sx) * cin2 + q_col[j];
            vq[j] = ok ? v : EVE_OOB;
        }
    };
    auto issue = [&](int slot, const int* vp, const int* vq) {      // always NDMA instructions
        const uint32_t pb = lds0 + slot * BUF;
        const uint32_t qb = pb + STEP * PROW;
#pragma unroll
        for (int j = 0; j < P_DMA; ++j) lds_dma16_asm(rs_dy, pb + ((wave * 64 + NT * j) % P_SLOTS) * 16, vp[j]);
#pragma unroll
        for (int j = 0; j < Q_DMA; ++j) lds_dma16_asm(rs_x, qb + ((wave * 64 + NT * j) % Q_SLOTS) * 16, vq[j]);
    };

    const int lane = tid & 63;
    const int wco = wave / WK, wk = wave % WK;
    const int t = lane & 15, g = lane >> 4;
    // lane-constant parts of the transposing reads: row (8g + t/4), +4 for the second read
    const int lrow = 8 * g + (t >> 2);
    const int keyp = tr_key<PROW>(lrow), keyq = tr_key<QROW>(lrow);     // bits 0,1,3 of the row only
    const int half = (t & 1) * 8, hs = (t & 3) >> 1;
    int poff[4], qoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        poff[i] = lrow * PROW + (((wco * 8 + i * 2 + hs) ^ (keyp << 1)) * 16) + half;
        qoff[i] = STEP * PROW + lrow * QROW + (((wk * 8 + i * 2 + hs) ^ (keyq << 1)) * 16) + half;
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t accb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) accb[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = BIAS && k0 == 0 && wk == 0;                   // wave-uniform
    const bool k_live = k0 + (uint32_t)wk * 64 < (uint32_t)p.K;        // wave-uniform: some filter column is real
    const uint4 ones = make_uint4(Elem<H>::ONE2, Elem<H>::ONE2, Elem<H>::ONE2, Elem<H>::ONE2);   // eight 1.0

    if (m_begin < m_end) {
        const int nsteps = (int)((m_end - m_begin + STEP - 1) / STEP);
        int vp[P_DMA], vq[Q_DMA];
#pragma unroll
        for (int pre = 0; pre < RING - 1; ++pre) {
            offsets(m_begin + pre * STEP, vp, vq);
            issue(pre, vp, vq);
        }
        offsets(m_begin + (RING - 1) * STEP, vp, vq);
        // One step on ring slot `slot` with the fragments' base addresses pbs / qbs.  MODE 1 (round 6): the loop is unrolled over
        // the ring, `slot` is a compile-time constant, pbs / qbs are lane constants with the LDS base folded in and the slot's
        // displacement (SB) is an IMMEDIATE of the transposing reads -- a run-time slot cost a modulo, a multiply and one v_add per
        // read.  Other modes (RefineNet's planes) and the BIAS variants keep the run-time slot: unrolled they need > 168 registers
        // (two workgroups per CU instead of three).
        auto step_body = [&](int st, int slot, auto sb_c, const uint32_t (&pbs)[4], const uint32_t (&qbs)[4]) {
            constexpr int SB = decltype(sb_c)::value;
            // stage st has landed once at most the RING-2 newer stages are outstanding (loads return in order)
            static_assert((RING == 3 && (NDMA == 4 || NDMA == 5 || NDMA == 6)) || (RING == 4 && (NDMA == 4 || NDMA == 5 || NDMA == 6)), "immediates below");
            if (RING == 3 && NDMA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (RING == 3 && NDMA == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (RING == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (NDMA == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (NDMA == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else           asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue((slot + RING - 1) % RING, vp, vq);          // stage st+RING-1 recycles the slot read in step st-1
            if (MT == 4 || k_live) {                          // (a wave whose 64 columns all lie beyond K only fetches)
                uint4 fp[4], fq[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < MT) {
                        const uint2 a0 = lds_tr_read<SB>(pbs[i]);
                        const uint2 a1 = lds_tr_read<SB + 4 * PROW>(pbs[i]);
                        fp[i] = make_uint4(a0.x, a0.y, a1.x, a1.y);
                    } else {
                        fp[i] = make_uint4(0u, 0u, 0u, 0u);
                    }
                    const uint2 b0 = lds_tr_read<SB>(qbs[i]);
                    const uint2 b1 = lds_tr_read<SB + 4 * QROW>(qbs[i]);
                    fq[i] = make_uint4(b0.x, b0.y, b1.x, b1.y);
                }
                if (MT == 4) {
                    mma16_inplace<H>(acc, fp, fq);              // acc[mt][kt] += P[mt] x Q[kt]
                } else {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        mma4_inplace<H>(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3], fp[mt], fq);
                }
                if (BIAS && do_bias) mma4_inplace_b<H>(accb, fp, ones);   // every column = sum over the 32 pixels
            }
            // address arithmetic of stage st+RING: independent VALU work the scheduler can slot between the MFMAs
            offsets(m_begin + (uint32_t)(st + RING) * STEP, vp, vq);
        };
        if constexpr (MODE == 1 && !BIAS) {
            static_assert((RING - 1) * BUF + 4 * (PROW > QROW ? PROW : QROW) < 65536, "slot displacement must fit the DS offset field");
            uint32_t pa[4], qa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { pa[i] = lds0 + (uint32_t)poff[i]; qa[i] = lds0 + (uint32_t)qoff[i]; }
            // the accumulators are updated in place by the asm MFMA groups, so any step count returns them where they were; a
            // tail of < RING steps follows with its slots known
            int st = 0;
            for (; st + RING <= nsteps; st += RING)
                static_for<RING>([&](auto i) { step_body(st + decltype(i)::value, decltype(i)::value, std::integral_constant<int, decltype(i)::value * BUF>{}, pa, qa); });
            static_for<RING - 1>([&](auto i) {
                if (st + decltype(i)::value < nsteps)
                    step_body(st + decltype(i)::value, decltype(i)::value, std::integral_constant<int, decltype(i)::value * BUF>{}, pa, qa);
            });
        } else {
            // two stages per trip (an odd tail stage is all out-of-range, i.e. adds zeros)
            auto do_step = [&](int st) {
                const uint32_t sb = lds0 + (st % RING) * BUF;
                uint32_t pa[4], qa[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { pa[i] = sb + (uint32_t)poff[i]; qa[i] = sb + (uint32_t)qoff[i]; }
                step_body(st, st % RING, std::integral_constant<int, 0>{}, pa, qa);
            };
            for (int st = 0; st < nsteps; st += 2) {
                do_step(st);
                do_step(st + 1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the zero-fill DMAs before LDS is released
        mma_drain();
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const uint32_t k = k0 + wk * 64 + kt * 16 + t;
            if (k >= (uint32_t)p.K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t co = co0 + wco * 64 + mt * 16 + g * 4 + r;
                if (co < (uint32_t)p.Cout) atomicAdd(dw + (size_t)co * p.K + k, acc[mt][kt][r]);
            }
        }
    if (BIAS && do_bias && t == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t co = co0 + wco * 64 + mt * 16 + g * 4 + r;
                if (co < (uint32_t)p.Cout) atomicAdd(db + co, accb[mt][r]);
            }
    }
}

}  // namespace eve

namespace eve {

// =================================================================================================
// 3x3 / stride 1 / pad 1 convolution (forward and data gradient) with the INPUT HALO resident in LDS.
//
// The per-tap kernel above re-fetches the activation tile for each of the 9 taps and has to hide a full
// L2/HBM round trip behind ONE K step of MFMA work; its waves spend half their time in s_waitcnt.  Here a
// block of 128 output pixels (TI images x TH rows x full width W) loads, per 32-channel slice, the
// (TH+2) x (W+2) halo patch ONCE (zero padding materialised by the DMA's out-of-range fill) and all 9 taps
// read it at different offsets.  Only the 8 KB weight tile changes every step, and it runs 3 steps ahead
// in a 4-slot ring; the next slice's halo streams in during the current slice.  The DMA count of every step
// is known (weight tile + at most one halo piece), so a counted s_waitcnt vmcnt(N) (loads return in order)
// is the whole synchronisation, plus one s_barrier per step.
//
// LDS rows are 64 B = one pixel (or output channel) x 32 channels, chunk-swizzled (see the kernel body).
// =================================================================================================
struct HaloParams {
    int N, H, W, Cin, Cout;       // x: [N][H][W][Cin]  out: [N][H][W][Cout]
    int TH, TI;                   // tile = TI images x TH rows x W columns = 128 pixels
    int bands;                    // ceil(H / TH) when TI == 1, else 1
    int flip;                     // 0: forward taps (kh-1, kw-1);  1: dgrad taps (1-kh, 1-kw)
    int K;                        // 9 * Cin (row stride of the weight matrix)
    int a_pieces;                 // halo DMA instructions per thread and slice (<= 7)
    uint32_t x_bytes, w_bytes;
    uint32_t tiles_m, tiles_n;
    FastDiv fd_w2, fd_hpi, fd_w, fd_th;  // divisions by (W+2), (TH+2)*(W+2), W, TH
};

template <typename H, int WM, int WN>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const HaloParams p, const H* __restrict__ x,
                                                           const H* __restrict__ w,
                                                           const float* __restrict__ bias, const int epi_act,
                                                           H* __restrict__ out) {
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
    // LDS layout (both operands): one 64-byte row per pixel / output channel = 32 channels as four 16-byte
    // chunks; chunk' = chunk ^ (key << 1).  key = bit 2 of the halo column for W >= 16 (a 16-lane read group
    // walks along one halo row) and the halo-row parity for W < 16 (the group spans 2..4 rows): every
    // ds_read_b128 of every tap is then bank-conflict free (brute-forced over all taps, tools/lds_banks.py).
    const bool wide = p.W >= 16;
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
            const int key = wide ? (hx >> 2) & 1 : (ti * (p.TH + 2) + hy) & 1;
            const int gy = y0 - 1 + hy, gx = hx - 1;
            const uint32_t n = n0 + ti;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W && n < (uint32_t)p.N)
                off = (int)((((n * p.H + gy) * p.W + gx) * p.Cin) * 2) + ((pc ^ (key << 1)) << 4);
        }
        a_goff[j] = off;
    }
    // ---- weight DMA slots: 64*WN rows (output channels) x 64 B ----
    int b_goff[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int L = tid + 256 * j;
        const int cl = L >> 2, pc = L & 3;
        // LDS row cl = (MFMA tile q, row i) holds channel (cl & ~63) + (i >> 2) * 16 + q * 4 + (i & 3), so that a lane
        // (rows 4g..4g+3 of the four tiles) owns 16 consecutive channels: 2 x 16-byte stores per pixel
        const int i16 = cl & 15, q = (cl >> 4) & 3;
        const uint32_t co = co0 + (cl & ~63) + (i16 >> 2) * 16 + q * 4 + (i16 & 3);
        b_goff[j] = co < (uint32_t)p.Cout ? (int)(co * (uint32_t)p.K) * 2 + ((pc ^ (((cl >> 2) & 1) << 1)) << 4) : EVE_OOB;
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

    // ---- fragment coordinates: every (tap, m-tile) LDS offset is a lane constant ----
    const int lane = tid & 63, wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lg = lane >> 4;
    int aaddr[9][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = wm * 64 + mt * 16 + li;                 // pixel in the tile: (ti, ty, tx)
        const int rowi = (int)fd_div((uint32_t)m, p.fd_w), tx = m - rowi * p.W;
        const int ti = (int)fd_div((uint32_t)rowi, p.fd_th), ty = rowi - ti * p.TH;
        const int hr0 = ti * (p.TH + 2) + ty;                 // halo row of tap dy = 0
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t % 3;
            const int dy = p.flip ? 2 - kh : kh, dx = p.flip ? 2 - kw : kw;
            const int hr = hr0 + dy, hx = tx + dx;
            const int key = wide ? (hx >> 2) & 1 : hr & 1;
            aaddr[t][mt] = ((hr * W2 + hx) << 6) + ((lg ^ (key << 1)) << 4);
        }
    }
    int brow[4];                                              // byte address of the weight fragment inside a ring slot
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int c = wn * 64 + nt * 16 + li;
        brow[nt] = (c << 6) + ((lg ^ (((c >> 2) & 1) << 1)) << 4);
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- synchronisation: loads return in order, so "at most N outstanding" = everything older has landed.
    // Step i issues [halo piece of the next slice (taps 0..a_pieces-1, not in the last slice)] + the WN DMAs
    // of weight tile i+3.  Before step i+1 weight tile i+1 (issued in step i-2) must be in LDS, i.e. only the
    // DMAs of steps i-1 and i may still be in flight: N = 2*WN + pieces issued in those two steps.
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
    // ---- prologue: halo of slice 0, then weight tiles 0, 1, 2 (the "steps -3..-1") ----
#pragma unroll
    for (int j = 0; j < 7; ++j)
        if (j < p.a_pieces) lds_dma16_asm(rs_x, ldsA + j * 4096 + wave_off, a_goff[j]);
    issue_b(0, 0, 0);
    issue_b(0, 1, 1);
    issue_b(0, 2, 2);
    wait_all_but(0);                                          // halo of slice 0 and weight tile 0 have landed
    __builtin_amdgcn_s_barrier();

    for (int s = 0; s < nslices; ++s) {
        const char* la = sA + (s & 1) * a_stage;
        const uint32_t na = ldsA + ((s + 1) & 1) * a_stage + wave_off;   // next slice's halo stage
        const int ap = s + 1 < nslices ? p.a_pieces : 0;      // halo pieces this slice still has to fetch
        const int nxt_c = (s + 1) * 64;                       // its channel byte offset
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const char* lb = sB + ((s + t) & 3) * BSLOT;
            uint4 fx[4], fw[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) fx[mt] = *reinterpret_cast<const uint4*>(la + aaddr[t][mt]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) fw[nt] = *reinterpret_cast<const uint4*>(lb + brow[nt]);
            // the DMA issues are spread between the MFMA groups so that they overlap matrix work
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) Mma<H>::run(acc[mt][nt], fw[nt], fx[mt]);
                if (nt == 0) {
                    if (t < 7 && t < ap)
                        lds_dma16_asm(rs_x, na + t * 4096, a_goff[t < 7 ? t : 0] != EVE_OOB ? a_goff[t < 7 ? t : 0] + nxt_c : EVE_OOB);
                } else if (nt == 1) {
                    issue_b(s + (t + 3) / 9, (t + 3) % 9, (s + t + 3) & 3);
                }
            }
            // pieces issued in steps t-1 and t:  [t-1 < ap] + [t < ap]  (the step before tap 0 never has one)
            int extra = ap - t + 1;
            extra = extra < 0 ? 0 : (extra > 2 ? 2 : extra);
            if (t == 0) extra = ap > 0 ? 1 : 0;
            wait_all_but(extra);
            __builtin_amdgcn_s_barrier();
        }
    }

    // ---- epilogue (two bodies, see act_fwd4): the lane owns channels co .. co+15 of four pixels ----
    auto epilogue = [&](auto fast) {
        const uint32_t co = co0 + wn * 64 + lg * 16;
        float bv[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) bv[c] = (bias && co + c < (uint32_t)p.Cout) ? bias[co + c] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = wm * 64 + mt * 16 + li;
            const int rowi = (int)fd_div((uint32_t)m, p.fd_w), tx = m - rowi * p.W;
            const int ti = (int)fd_div((uint32_t)rowi, p.fd_th), ty = rowi - ti * p.TH;
            const uint32_t n = n0 + ti;
            const int y = y0 + ty;
            if (n >= (uint32_t)p.N || y >= p.H) continue;
            H* dst = out + ((size_t)(n * p.H + y) * p.W + tx) * p.Cout + co;
#pragma unroll
            for (int h = 0; h < 2; ++h) {                     // two 16-byte halves of 8 channels
                if (co + 8 * h + 8 > (uint32_t)p.Cout) continue;       // Cout is a multiple of 8
                float o[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) o[c] = acc[mt][2 * h + (c >> 2)][c & 3] + bv[8 * h + c];
                act_fwd4<decltype(fast)::value>(o, epi_act);
                act_fwd4<decltype(fast)::value>(o + 4, epi_act);
                if (epi_act & EVE_EPI_ACC) {
                    float old[8];
                    Elem<H>::unpack(*reinterpret_cast<const uint4*>(dst + 8 * h), old);
#pragma unroll
                    for (int c = 0; c < 8; ++c) o[c] += old[c];
                }
                *reinterpret_cast<uint4*>(dst + 8 * h) = Elem<H>::pack(o);
            }
        }
    };
    if (act_is_fast(epi_act)) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}


// -------------------------------------------------------------------------------------------------
// Persistent variant of the kernel above.  A workgroup walks tiles t = lid, lid + G, ... and treats their
// (tile, slice, tap) steps as ONE stream: the halo stages keep ping-ponging across the tile boundary (the last
// slice of a tile prefetches the first slice of the next one), the weight ring keeps running 3 steps ahead, and
// the only per-tile work is the epilogue.  With one tile per workgroup the two workgroups of a CU start, fetch
// their first halo (a full HBM round trip with no MFMA work to hide it) and finish in lockstep; for the 64-channel
// layers (18 steps per tile) that prologue was as long as the tile itself.  Tap / fragment addresses and the
// halo-slot geometry are computed once per workgroup instead of once per tile.
// -------------------------------------------------------------------------------------------------
template <typename H, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_pkernel(const HaloParams p, const H* __restrict__ x,
                                                            const H* __restrict__ w,
                                                            const float* __restrict__ bias, const int epi_act,
                                                            H* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W2 = p.W + 2, HPI = (p.TH + 2) * W2, HP = p.TI * HPI;
    const int a_stage = p.a_pieces * 4096;
    char* const sA = smem;
    constexpr int BSLOT = 4096 * WN;
    char* const sB = smem + 2 * a_stage;
    float* const sBias = reinterpret_cast<float*>(smem + 2 * a_stage + 4 * BSLOT);   // Cout rounded up to the tile, if bias

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t G = gridDim.x, T = p.tiles_m * p.tiles_n;
    const uint32_t lid = xcd_remap(blockIdx.x, G);

    const eve_int4 rs_x = make_rsrc_words(x, p.x_bytes);
    const eve_int4 rs_w = make_rsrc_words(w, p.w_bytes);
    const uint32_t ldsA = lds_addr_of(sA), ldsB = lds_addr_of(sB);
    const bool wide = p.W >= 16;
    if (bias) {                                               // (visible to every wave after the first tile's barriers)
        for (int i = threadIdx.x; i < (int)p.tiles_n * 64 * WN; i += 256) sBias[i] = i < p.Cout ? bias[i] : 0.f;
    }

    // ---- halo DMA slots (lane constants): offset relative to pixel (n0, y0, 0), halo row / image of the slot ----
    int a_rel[7], a_meta[7];                                  // meta = (ti << 8) | hy, or -1 for a slot that is never live
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int L = tid + 256 * j;
        const int hp = L >> 2, pc = L & 3;
        a_rel[j] = 0; a_meta[j] = -1;
        if (j < p.a_pieces && hp < HP) {
            const int ti = (int)fd_div((uint32_t)hp, p.fd_hpi);
            const int r = hp - ti * HPI;
            const int hy = (int)fd_div((uint32_t)r, p.fd_w2), hx = r - hy * W2;
            const int key = wide ? (hx >> 2) & 1 : (ti * (p.TH + 2) + hy) & 1;
            if (hx >= 1 && hx <= p.W) {
                a_rel[j] = (((ti * p.H + hy - 1) * p.W + hx - 1) * p.Cin) * 2 + ((pc ^ (key << 1)) << 4);
                a_meta[j] = (ti << 8) | hy;
            }
        }
    }
    // weight slots.  LDS row cl = (16-row MFMA tile q, row i) holds output channel (cl & ~63) + (i >> 2) * 16 + q * 4 +
    // (i & 3): a lane (rows 4g..4g+3 of the four tiles) then owns 16 CONSECUTIVE channels of its pixel and the
    // epilogue writes 2 x 16 bytes per pixel instead of 4 x 8.
    int b_rel[WN], b_ch[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int L = tid + 256 * j;
        const int cl = L >> 2, pc = L & 3;
        const int i16 = cl & 15, q = (cl >> 4) & 3;
        b_ch[j] = (cl & ~63) + (i16 >> 2) * 16 + q * 4 + (i16 & 3);
        b_rel[j] = (b_ch[j] * p.K) * 2 + ((pc ^ (((cl >> 2) & 1) << 1)) << 4);
    }

    const int nslices = p.Cin / 32;
    const int wave_off = wave * 1024;
    const int lane = tid & 63, wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lg = lane >> 4;
    int aaddr[9][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = wm * 64 + mt * 16 + li;
        const int rowi = (int)fd_div((uint32_t)m, p.fd_w), tx = m - rowi * p.W;
        const int ti = (int)fd_div((uint32_t)rowi, p.fd_th), ty = rowi - ti * p.TH;
        const int hr0 = ti * (p.TH + 2) + ty;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t % 3;
            const int dy = p.flip ? 2 - kh : kh, dx = p.flip ? 2 - kw : kw;
            const int hr = hr0 + dy, hx = tx + dx;
            const int key = wide ? (hx >> 2) & 1 : hr & 1;
            aaddr[t][mt] = ((hr * W2 + hx) << 6) + ((lg ^ (key << 1)) << 4);
        }
    }
    int brow[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int c = wn * 64 + nt * 16 + li;
        brow[nt] = (c << 6) + ((lg ^ (((c >> 2) & 1) << 1)) << 4);
    }

    // tile id -> (first image, first row, first output channel)
    auto tile_coords = [&](uint32_t t, uint32_t& n0, int& y0, uint32_t& co0) {
        const uint32_t tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
        if (p.TI == 1) { n0 = tm / p.bands; y0 = (int)(tm - n0 * p.bands) * p.TH; }
        else { n0 = tm * p.TI; y0 = 0; }
        co0 = tn * (64 * WN);
    };
    // halo piece j of the tile at (n0, y0), channel slice sl, into stage `st`
    auto issue_a = [&](int j, uint32_t n0, int y0, int sl, int st, bool live) {
        const int meta = a_meta[j];
        const int hy = meta & 0xff, ti = meta >> 8;
        const bool ok = live & (meta >= 0) & ((uint32_t)(y0 + hy - 1) < (uint32_t)p.H) & (n0 + (uint32_t)ti < (uint32_t)p.N);
        const int base = (int)(((n0 * p.H + y0) * p.W) * p.Cin) * 2 + sl * 64;
        lds_dma16_asm(rs_x, ldsA + st * a_stage + j * 4096 + wave_off, ok ? base + a_rel[j] : EVE_OOB);
    };
    // weight tile (slice sl, tap tb) of the channel block at co0 into ring slot `slot`
    auto issue_b = [&](uint32_t co0, int sl, int tb, int slot, bool live) {
        const int koff = (int)(co0 * (uint32_t)p.K) * 2 + (tb * p.Cin + sl * 32) * 2;
        const uint32_t dst = ldsB + slot * BSLOT + wave_off;
#pragma unroll
        for (int j = 0; j < WN; ++j)
            lds_dma16_asm(rs_w, dst + j * 4096,
                          (live && co0 + (uint32_t)b_ch[j] < (uint32_t)p.Cout) ? b_rel[j] + koff : EVE_OOB);
    };
    auto wait_all_but = [&](int extra) {
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

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (lid >= T) return;
    uint32_t n0, co0;
    int y0;
    tile_coords(lid, n0, y0, co0);
    // ---- prologue: halo slice 0 of the first tile, weight tiles of steps 0..2 ----
#pragma unroll
    for (int j = 0; j < 7; ++j)
        if (j < p.a_pieces) issue_a(j, n0, y0, 0, 0, true);
    issue_b(co0, 0, 0, 0, true);
    issue_b(co0, 0, 1, 1, true);
    issue_b(co0, 0, 2, 2, true);
    wait_all_but(0);
    __builtin_amdgcn_s_barrier();

    uint32_t gs = 0;                                          // slices consumed so far (stage parity, ring phase)
    for (uint32_t tile = lid; tile < T; tile += G) {
        const uint32_t nxt = tile + G;
        const bool has_next = nxt < T;
        uint32_t n1 = 0, co1 = 0;
        int y1 = 0;
        if (has_next) tile_coords(nxt, n1, y1, co1);
        for (int s = 0; s < nslices; ++s, ++gs) {
            const char* la = sA + (gs & 1) * a_stage;
            const int nst = (int)((gs + 1) & 1);
            const bool last = s + 1 == nslices;
            // the slice after this one in the stream: same tile, or slice 0 of the next tile
            const bool more = !last || has_next;
            const uint32_t an = last ? n1 : n0;
            const int ay = last ? y1 : y0, asl = last ? 0 : s + 1;
            const int ap = more ? p.a_pieces : 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const char* lb = sB + ((gs + t) & 3) * BSLOT;
                uint4 fx[4], fw[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) fx[mt] = *reinterpret_cast<const uint4*>(la + aaddr[t][mt]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) fw[nt] = *reinterpret_cast<const uint4*>(lb + brow[nt]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    mma4_inplace<H>(acc[0][nt], acc[1][nt], acc[2][nt], acc[3][nt], fw[nt], fx);
                    if (nt == 0) {
                        if (t < 7 && t < ap) issue_a(t < 7 ? t : 0, an, ay, asl, nst, true);
                    } else if (nt == 1) {
                        // weight tile of step +3: this slice, the next slice of the tile, or the next tile's first slice
                        const bool wrap = t + 3 >= 9;
                        const bool to_next = wrap && last;
                        issue_b(to_next ? co1 : co0, wrap ? (last ? 0 : s + 1) : s, (t + 3) % 9, (int)((gs + t + 3) & 3),
                                !to_next || has_next);
                    }
                }
                int extra = ap - t + 1;
                extra = extra < 0 ? 0 : (extra > 2 ? 2 : extra);
                if (t == 0) extra = ap > 0 ? 1 : 0;
                // The epilogue's global stores sit between the DMAs of two tiles and count in vmcnt as well (their
                // number is not a constant: fully masked stores are skipped).  So the last step of a tile waits one
                // step deeper -- only its own weight DMAs stay in flight, which covers everything the next tile's first
                // step needs -- and that first step does not wait at all; from the second step on the usual count is
                // conservative again (the stores are older than both steps' DMAs).
                if (last && t == 8) {
                    if (WN == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else         asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                } else if (!(t == 0 && s == 0 && tile != lid)) {
                    wait_all_but(extra);
                }
                __builtin_amdgcn_s_barrier();
            }
        }
        // ---- epilogue of this tile (bias from LDS, identity / ReLU: the launcher sends every other activation to the
        //      one-tile kernel); the DMAs of the next tile's first steps are already in flight ----
        mma_drain();
        const uint32_t co = co0 + wn * 64 + lg * 16;          // the lane's 16 consecutive channels: co + nt*4 + r
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = wm * 64 + mt * 16 + li;
            const int rowi = (int)fd_div((uint32_t)m, p.fd_w), tx = m - rowi * p.W;
            const int ti = (int)fd_div((uint32_t)rowi, p.fd_th), ty = rowi - ti * p.TH;
            const uint32_t n = n0 + ti;
            const int y = y0 + ty;
            uint32_t pk[8];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = acc[mt][nt][r];
                acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (bias) {
                    const float4 bq = *reinterpret_cast<const float4*>(sBias + co + nt * 4);
                    o[0] += bq.x; o[1] += bq.y; o[2] += bq.z; o[3] += bq.w;
                }
                act_fwd4<true>(o, epi_act);
                pk[2 * nt] = Elem<H>::pack2(o[0], o[1]);
                pk[2 * nt + 1] = Elem<H>::pack2(o[2], o[3]);
            }
            if (n >= (uint32_t)p.N || y >= p.H) continue;
            H* dst = out + ((size_t)(n * p.H + y) * p.W + tx) * p.Cout + co;
            if (co + 8 <= (uint32_t)p.Cout) *reinterpret_cast<uint4*>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            if (co + 16 <= (uint32_t)p.Cout) *reinterpret_cast<uint4*>(dst + 8) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        }
        n0 = n1; y0 = y1; co0 = co1;
    }
}

}  // namespace eve
