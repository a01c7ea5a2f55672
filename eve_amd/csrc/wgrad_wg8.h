// Weight gradient of the 256..512-channel trunk layers, eight-wave workgroups: 256 output channels x 256 filter-K values per
// workgroup, a wave owns 128 x 64 (32 accumulator tiles of 16 x 16), the waves of a SIMD in opposite halves of a step.
//
// Why (profiles/r03_wgrad_wg8.md): wgrad_tr_kernel<2,2> (conv_fast.h) runs four-wave workgroups on 128 x 128 tiles, three per
// CU.  Its SQ counters on the layer-4 shape: 2.6 VALU + 2.75 SALU + 1 LDS + 0.27 VMEM instructions per (16-cycle) MFMA, three
// waves per SIMD, SQ_ACTIVE_INST_ANY 38 % per wave -- the SIMDs' issue slots are the bound (46 % matrix-pipe occupancy in
// cycles, 0.95 PFLOP/s), not LDS (0 bank conflicts) and not the DMA volume (a 256 x 256 tile run by sixteen 64 x 64 waves,
// i.e. the same per-wave instruction mix on half the DMA bytes, measured SLOWER).  So the fix is the one of conv_wg8.h:
// twice the MFMAs per wave and step (32 per 24 transposing reads + 4 DMAs), every LDS address a lane constant plus an
// IMMEDIATE (the four ring slots are unrolled), and a role split: between two workgroup barriers group 0 (waves 0-3) runs
// [read step s, multiply step s], group 1 (waves 4-7, one of them on every SIMD, static priority) [multiply step s-1,
// read step s].
//
// Operands and layouts are wgrad_tr_kernel's: a step is 32 pixels; dy rows [pixel][256 channels] and the gathered x rows
// [pixel][256 k] (k = (tap, input channel), tap-shifted pixel, zero fill outside the image) land in LDS by LDS-DMA in their
// natural layout and the K(=pixel)-major MFMA fragments come out of ds_read_b64_tr_b16; 512-byte rows, 16-byte slots XOR-ed by
// tr_key<512>(row).  Power-of-two output sizes only (every trunk layer): the decoded pixel of a slot is the sum of a
// wave-uniform part and a lane constant (the MODE == 1 arithmetic of wgrad_tr_kernel).
//
// Ring: 4 stages of 32 KB, stage s+2 is issued in interval s (both groups), waited for at the end of interval s+1, read in
// interval s+2; its slot held stage s-2, whose last reads (group 1, end of interval s-2) completed at the first lgkmcnt(0) of
// interval s-1 -- the schedule of conv_wg8.h.
#pragma once
#include "common.h"
#include "lds_dma.h"
#include "conv_fast.h"

namespace eve {

// SLAB: the workgroup of pixel split s STORES its partial filter tile into slab s of a scratch buffer laid out like dw
// ([split][Cout][K] floats) and wgrad_slab_reduce_kernel adds the slabs into dw; else float atomics straight into dw.
// With one workgroup per CU all workgroups reach their epilogue together, so the 16.5 M atomics of a layer-4 launch
// (252 workgroups x 65 536) are fully exposed: 43 of its 148 us (profiles/r03_wgrad_wg8.md).
template <typename H, bool SLAB>
__global__ __launch_bounds__(512, 2) void wgrad_wg8_kernel(const GatherParams p, const H* __restrict__ x,
                                                           const H* __restrict__ dy, float* __restrict__ dw,
                                                           const uint32_t rows_per_split, const uint32_t x_bytes,
                                                           const uint32_t dy_bytes) {
    constexpr int WK = 4;                                    // waves: 2 (channel halves of 128) x 4 (K quarters of 64)
    constexpr int BCO = 256, BKK = 256;
    constexpr int ROW = 512;                                 // bytes per pixel row of either operand
    constexpr int SL = ROW / 16;                             // 16-byte slots per row
    constexpr int STEP = 32, NT = 512;
    constexpr int SLOTS = STEP * SL;                         // 1024 slots per operand and stage = 2 per thread
    constexpr int BUF = 2 * STEP * ROW;                      // 32 KB per stage
    constexpr int RING = 4;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t tk = (uint32_t)p.K / BKK, tc = (uint32_t)p.Cout / BCO;
    const uint32_t lid = tk * tc >= 4 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t k0 = (lid % tk) * BKK, co0 = ((lid / tk) % tc) * BCO;
    const uint32_t m_begin = (lid / (tk * tc)) * rows_per_split;
    const uint32_t m_end = min(p.M, m_begin + rows_per_split);

    const eve_int4 rs_x = make_rsrc_words(x, x_bytes);
    const eve_int4 rs_dy = make_rsrc_words(dy, dy_bytes);
    const uint32_t lds0 = lds_addr_of(lds);

    // ---- DMA slots (lane constants): dy = P operand, gathered x = Q operand ----
    const int cin2 = p.Cin * 2, cout2 = p.Cout * 2;
    const int sh_w = __builtin_ctz((unsigned)p.OW), sh_hw = __builtin_ctz((unsigned)(p.OH * p.OW));
    int p_row[2], p_const[2], q_row[2], q_const[2], q_ty[2], q_tx[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = tid + NT * j;
        const int row = q / SL, sl = q % SL;
        const int sg = sl ^ (tr_key<ROW>(row) << 1);
        p_row[j] = q_row[j] = row;
        p_const[j] = row * cout2 + (int)(co0 + sg * 8) * 2;                      // (Cout is a multiple of 256: never out of range)
        const uint32_t k = k0 + sg * 8;
        const uint32_t tap = fd_div(k, p.fd_cin), kh = fd_div(tap, p.fd_kw);
        const int col = (int)(k - tap * (uint32_t)p.Cin) * 2;
        const int ddy = (int)kh * p.k_mul + p.off, ddx = (int)(tap - kh * (uint32_t)p.KW) * p.k_mul + p.off;
        const uint32_t r = (uint32_t)row;
        const int n_t = (int)(r >> sh_hw), oy_t = (int)((r >> sh_w) & (uint32_t)(p.OH - 1)), ox_t = (int)(r & (uint32_t)(p.OW - 1));
        q_ty[j] = oy_t * p.o_mul + ddy;
        q_tx[j] = ox_t * p.o_mul + ddx;
        q_const[j] = ((n_t * p.IH + q_ty[j]) * p.IW + q_tx[j]) * cin2 + col;
    }
    // source offsets of the stage that starts at pixel mbase (a multiple of 32); rows past the split's end: zero fill
    auto offsets = [&](uint32_t mbase, int* vp, int* vq) {
        const uint32_t left = m_end > mbase ? m_end - mbase : 0u;
        const int pbase = (int)mbase * cout2;
        const uint32_t n_s = mbase >> sh_hw, oy_s = (mbase >> sh_w) & (uint32_t)(p.OH - 1), ox_s = mbase & (uint32_t)(p.OW - 1);
        const int sy0 = (int)oy_s * p.o_mul, sx0 = (int)ox_s * p.o_mul;
        const int qbase = (((int)n_s * p.IH + sy0) * p.IW + sx0) * cin2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            vp[j] = (uint32_t)p_row[j] < left ? pbase + p_const[j] : EVE_OOB;
            const bool ok = ((uint32_t)q_row[j] < left) & ((uint32_t)(sy0 + q_ty[j]) < (uint32_t)p.IH) &
                            ((uint32_t)(sx0 + q_tx[j]) < (uint32_t)p.IW);
            vq[j] = ok ? qbase + q_const[j] : EVE_OOB;
        }
    };
    auto issue = [&](int slot, const int* vp, const int* vq) {
        const uint32_t pb = lds0 + slot * BUF, qb = pb + STEP * ROW;
#pragma unroll
        for (int j = 0; j < 2; ++j) lds_dma16_asm(rs_dy, pb + (wave * 64 + NT * j) * 16, vp[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) lds_dma16_asm(rs_x, qb + (wave * 64 + NT * j) * 16, vq[j]);
    };

    // ---- fragment addresses (lane constants; ring slot and the +4-row half are immediates) ----
    const int lane = tid & 63;
    const int wco = wave / WK, wk = wave % WK;
    const int t = lane & 15, g = lane >> 4;
    const int lrow = 8 * g + (t >> 2);
    const int key = tr_key<ROW>(lrow);                       // (bits 0, 1, 3 of the row: the same for row + 4)
    const int half = (t & 1) * 8, hs = (t & 3) >> 1;
    uint32_t poff[8], qoff[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) poff[i] = lds0 + lrow * ROW + (((wco * 16 + i * 2 + hs) ^ (key << 1)) * 16) + half;
#pragma unroll
    for (int i = 0; i < 4; ++i) qoff[i] = lds0 + STEP * ROW + lrow * ROW + (((wk * 8 + i * 2 + hs) ^ (key << 1)) * 16) + half;

    f32x4_t acc[2][4][4];                                    // [channel half of 64][16-channel tile][16-k tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][b][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (m_begin < m_end) {
        const int nsteps = (int)((m_end - m_begin + STEP - 1) / STEP);
        const bool lead = wave < 4;
        int vp[2], vq[2];
        offsets(m_begin, vp, vq);
        issue(0, vp, vq);
        offsets(m_begin + STEP, vp, vq);
        issue(1, vp, vq);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!lead) __builtin_amdgcn_s_setprio(2);
        uint4 fp[8], fq[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) fp[i] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int i = 0; i < 4; ++i) fq[i] = make_uint4(0u, 0u, 0u, 0u);
        auto mma = [&]() {
            mma16_inplace<H>(acc[0], reinterpret_cast<const uint4(&)[4]>(fp[0]), fq);
            mma16_inplace<H>(acc[1], reinterpret_cast<const uint4(&)[4]>(fp[4]), fq);
        };
        // one step; SLOT = st & 3 is a template constant so that every fragment address is base + immediate
        auto step = [&](int st, auto slot_c) {
            constexpr int SLOT = decltype(slot_c)::value;
            if (!lead) mma();                                                   // step st - 1 (zeros before step 0)
            constexpr int SB = SLOT * BUF;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint2 a0 = SB < 65536 ? lds_tr_read<SB>(poff[i]) : lds_tr_read<SB - 65536>(poff[i] + 65536);
                const uint2 a1 = SB + 4 * ROW < 65536 ? lds_tr_read<SB + 4 * ROW>(poff[i]) : lds_tr_read<SB + 4 * ROW - 65536>(poff[i] + 65536);
                fp[i] = make_uint4(a0.x, a0.y, a1.x, a1.y);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint2 b0 = SB < 49152 ? lds_tr_read<SB>(qoff[i]) : lds_tr_read<SB - 49152>(qoff[i] + 49152);
                const uint2 b1 = SB + 4 * ROW < 49152 ? lds_tr_read<SB + 4 * ROW>(qoff[i]) : lds_tr_read<SB + 4 * ROW - 49152>(qoff[i] + 49152);
                fq[i] = make_uint4(b0.x, b0.y, b1.x, b1.y);
            }
            // stage st + 2 (all out of range past the end: zero fill, so every step issues and waits for the same count)
            offsets(m_begin + (uint32_t)(st + 2) * STEP, vp, vq);
            issue((SLOT + 2) & 3, vp, vq);
            if (lead) mma();
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                  // everything older than this step's four DMAs
            __builtin_amdgcn_s_barrier();
        };
        int st = 0;
        for (; st + 4 <= nsteps; st += 4) {
            step(st, std::integral_constant<int, 0>{});
            step(st + 1, std::integral_constant<int, 1>{});
            step(st + 2, std::integral_constant<int, 2>{});
            step(st + 3, std::integral_constant<int, 3>{});
        }
        // 0..3 remaining steps (a whole number of ring turns was done, the next slot is 0 again)
        if (st < nsteps) step(st, std::integral_constant<int, 0>{});
        if (st + 1 < nsteps) step(st + 1, std::integral_constant<int, 1>{});
        if (st + 2 < nsteps) step(st + 2, std::integral_constant<int, 2>{});
        if (!lead) mma();                                                       // group 1's last step
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the zero-fill DMAs, before LDS is released
    }
    // wait states between the last MFMA and the first read of an accumulator: the statement NAMES the accumulators (a
    // memory clobber does not order register-only instructions, conv_wg8.h)
#pragma unroll
    for (int a = 0; a < 2; ++a)
        asm volatile("s_nop 15\n\ts_nop 15"
                     : "+a"(acc[a][0][0]), "+a"(acc[a][0][1]), "+a"(acc[a][0][2]), "+a"(acc[a][0][3]), "+a"(acc[a][1][0]), "+a"(acc[a][1][1]),
                       "+a"(acc[a][1][2]), "+a"(acc[a][1][3]), "+a"(acc[a][2][0]), "+a"(acc[a][2][1]), "+a"(acc[a][2][2]), "+a"(acc[a][2][3]),
                       "+a"(acc[a][3][0]), "+a"(acc[a][3][1]), "+a"(acc[a][3][2]), "+a"(acc[a][3][3]) :: "memory");
    float* const dst = SLAB ? dw + (size_t)(lid / (tk * tc)) * ((size_t)p.Cout * p.K) : dw;      // slab of this pixel split
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const uint32_t k = k0 + wk * 64 + kt * 16 + t;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t co = co0 + wco * 128 + a * 64 + mt * 16 + g * 4 + r;
                    if (SLAB) dst[(size_t)co * p.K + k] = acc[a][mt][kt][r];
                    else atomicAdd(dst + (size_t)co * p.K + k, acc[a][mt][kt][r]);
                }
            }
}

// dw[i] += sum over the `splits` slabs (fixed order: the result does not depend on the launch's timing, unlike the atomics)
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                const long long n4, const int splits) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 a = reinterpret_cast<const float4*>(dw)[i];
        for (int s = 0; s < splits; ++s) {
            const float4 b = reinterpret_cast<const float4*>(slabs)[(long long)s * n4 + i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<float4*>(dw)[i] = a;
    }
}

}  // namespace eve
