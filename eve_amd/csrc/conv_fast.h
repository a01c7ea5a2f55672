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
#include "common.h"

namespace eve {

#define EVE_LDS __attribute__((address_space(3)))
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4v_t;

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, const void* lds_generic_ptr, int voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (EVE_LDS void*)(lds_generic_ptr), 16, voffset, 0, 0, 0);
}
#define EVE_OOB ((int)0x80000000u)   // >= num_records for every tensor we accept (< 2^31 bytes)

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

    const int nk = tp.ntaps * p.Cin / BK;
    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // tile kt has landed for every wave; tile kt-1 fully consumed
        if (kt + 1 < nk) issue(cur ^ 1);
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

    const bool vec_ok = (p.Cout & 3) == 0;
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
            for (int r = 0; r < 4; ++r) o[r] = act_fwd(acc[mt][nt][r] + bv[r], epi_act);
            size_t opix = m;
            if (tp.osy != 1 || tp.osx != 1) {       // uniform: only the strided-dgrad sub-problems remap pixels
                const uint32_t n = fd_div(m, p.fd_ohw);
                const uint32_t rem = m - n * (uint32_t)(p.OH * p.OW);
                const uint32_t oy = fd_div(rem, p.fd_ow);
                const uint32_t ox = rem - oy * (uint32_t)p.OW;
                opix = ((size_t)n * tp.OHf + oy * tp.osy + tp.oy0) * tp.OWf + ox * tp.osx + tp.ox0;
            }
            T* dst = out + opix * p.Cout + co;
            if (vec_ok) {
                if (sizeof(T) == 4) {
                    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    uint2 pk;
                    pk.x = f32_to_bf16_bits(o[0]) | (f32_to_bf16_bits(o[1]) << 16);
                    pk.y = f32_to_bf16_bits(o[2]) | (f32_to_bf16_bits(o[3]) << 16);
                    *reinterpret_cast<uint2*>(dst) = pk;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < (uint32_t)p.Cout) Elem<T>::st(dst + r, o[r]);
            }
        }
    }
}

// =================================================================================================
// bf16 weight gradient.  Block tile = (64*WCO output channels) x (64*WK filter-K values); every wave owns
// a 64 x 64 piece; each step consumes 64 pixels (two MFMA K=32 chunks).
// =================================================================================================
template <int ROWB>
__device__ __forceinline__ int tr_key(int row) {
    const int b0 = row & 1, b1 = (row >> 1) & 1, b2 = (row >> 3) & 1;
    return ROWB == 128 ? (b1 | (b2 << 1)) : (b0 | (b1 << 1) | (b2 << 2));
}

__device__ __forceinline__ uint2 lds_tr_read(const char* lds_generic_ptr) {
    bf16x4v_t r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((EVE_LDS bf16x4v_t*)(EVE_LDS void*)(lds_generic_ptr));
    return __builtin_bit_cast(uint2, r);
}

template <int WCO, int WK>
__global__ __launch_bounds__(256) void wgrad_tr_kernel(const GatherParams p, const bf16_t* __restrict__ x,
                                                       const bf16_t* __restrict__ dy, float* __restrict__ dw,
                                                       const uint32_t rows_per_split, const uint32_t x_bytes,
                                                       const uint32_t dy_bytes) {
    constexpr int BCO = 64 * WCO, BKK = 64 * WK;
    constexpr int PROW = BCO * 2, QROW = BKK * 2;            // bytes per pixel row
    constexpr int PSL = PROW / 16, QSL = QROW / 16;          // 16-byte slots per row
    constexpr int P_DMA = 64 * PSL / 256, Q_DMA = 64 * QSL / 256;
    constexpr int BUF = 64 * (PROW + QROW);                  // bytes per stage
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t k0 = blockIdx.x * BKK, co0 = blockIdx.y * BCO;
    const uint32_t m_begin = blockIdx.z * rows_per_split;
    const uint32_t m_end = min(p.M, m_begin + rows_per_split);
    const uint32_t ohw = (uint32_t)(p.OH * p.OW);

    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, dy_bytes, 0x00020000);

    // ---- loop-invariant slot coordinates ----
    int p_row[P_DMA], p_col[P_DMA];                 // row in the stage, byte offset of the channel in dy's row
#pragma unroll
    for (int j = 0; j < P_DMA; ++j) {
        const int q = tid + 256 * j;
        const int row = q / PSL, s = q % PSL;
        const int sg = s ^ (tr_key<PROW>(row) << 1);
        const uint32_t co = co0 + sg * 8;
        p_row[j] = row;
        p_col[j] = co < (uint32_t)p.Cout ? (int)co * 2 : EVE_OOB;
    }
    int q_row[Q_DMA], q_col[Q_DMA], q_dy[Q_DMA], q_dx[Q_DMA];
#pragma unroll
    for (int j = 0; j < Q_DMA; ++j) {
        const int q = tid + 256 * j;
        const int row = q / QSL, s = q % QSL;
        const int sg = s ^ (tr_key<QROW>(row) << 1);
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

    auto issue = [&](uint32_t mbase, int buf) {
        const char* pb = lds + buf * BUF + wave * 1024;
        const char* qb = lds + buf * BUF + 64 * PROW + wave * 1024;
#pragma unroll
        for (int j = 0; j < P_DMA; ++j) {
            const uint32_t m = mbase + p_row[j];
            const int voff = (m < m_end && p_col[j] != EVE_OOB) ? (int)(m * (uint32_t)p.Cout) * 2 + p_col[j] : EVE_OOB;
            lds_dma16(rs_dy, pb + j * 4096, voff);
        }
#pragma unroll
        for (int j = 0; j < Q_DMA; ++j) {
            const uint32_t m = mbase + q_row[j];
            int voff = EVE_OOB;
            if (m < m_end && q_col[j] != EVE_OOB) {
                const uint32_t n = fd_div(m, p.fd_ohw);
                const uint32_t rem = m - n * ohw;
                const uint32_t oy = fd_div(rem, p.fd_ow);
                const uint32_t ox = rem - oy * (uint32_t)p.OW;
                const int sy = (int)oy * p.o_mul + q_dy[j], sx = (int)ox * p.o_mul + q_dx[j];
                if (sy >= 0 && sy < p.IH && sx >= 0 && sx < p.IW)
                    voff = (((int)n * p.IH + sy) * p.IW + sx) * p.Cin * 2 + q_col[j];
            }
            lds_dma16(rs_x, qb + j * 4096, voff);
        }
    };

    const int lane = tid & 63;
    const int wco = wave / WK, wk = wave % WK;
    const int t = lane & 15, g = lane >> 4;
    // lane-constant parts of the transposing reads: row (8g + t/4) (+4 for the second read, +32 per chunk)
    const int lrow = 8 * g + (t >> 2);
    const int keyp = tr_key<PROW>(lrow), keyq = tr_key<QROW>(lrow);     // bits 0,1,3 of the row only
    const int half = (t & 1) * 8, hs = (t & 3) >> 1;
    int poff[4], qoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        poff[i] = lrow * PROW + (((wco * 8 + i * 2 + hs) ^ (keyp << 1)) * 16) + half;
        qoff[i] = 64 * PROW + lrow * QROW + (((wk * 8 + i * 2 + hs) ^ (keyq << 1)) * 16) + half;
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (m_begin < m_end) {
        const int nsteps = (int)((m_end - m_begin + 63) / 64);
        issue(m_begin, 0);
        for (int st = 0; st < nsteps; ++st) {
            const int cur = st & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (st + 1 < nsteps) issue(m_begin + (uint32_t)(st + 1) * 64, cur ^ 1);
            const char* sb = lds + cur * BUF;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                uint4 fp[4], fq[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint2 a0 = lds_tr_read(sb + poff[i] + (32 * c) * PROW);
                    const uint2 a1 = lds_tr_read(sb + poff[i] + (32 * c + 4) * PROW);
                    fp[i] = make_uint4(a0.x, a0.y, a1.x, a1.y);
                    const uint2 b0 = lds_tr_read(sb + qoff[i] + (32 * c) * QROW);
                    const uint2 b1 = lds_tr_read(sb + qoff[i] + (32 * c + 4) * QROW);
                    fq[i] = make_uint4(b0.x, b0.y, b1.x, b1.y);
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) Mma<bf16_t>::run(acc[mt][kt], fp[mt], fq[kt]);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
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
}

}  // namespace eve
