// Implicit-GEMM convolution on MFMA for gfx950: forward, data gradient and weight gradient.
//
// One gather formulation serves forward and dgrad:
//     out[n, oy, ox, co] = sum_{kh,kw,c} src[n, sy, sx, c] * W[co][(kh,kw,c)]
//     sy = (oy*o_mul + kh*k_mul + off) / div     (tap skipped unless divisible and in range)
// forward: o_mul = stride, k_mul = 1, off = -pad, div = 1       (src = x,  W = [Cout][KH][KW][Cin])
// dgrad  : o_mul = 1, k_mul = -1, off = +pad, div = stride      (src = dy, W = [Cin][KH][KW][Cout])
//
// Tiling (both dtypes share it because everything is expressed in 16-byte vectors):
//   workgroup = 256 threads = 4 waves (2 x 2), tile 128 pixels x (64|128) output channels,
//   K step = 8 vectors = 128 bytes per row (64 bf16 / 32 f32), LDS double-buffered, one barrier per
//   K step, next tile's global loads in flight (in registers) while the current tile is multiplied.
//   LDS rows are 128 B with the 16-B slot index XOR-swizzled by (row & 7): ds_read_b128 fragment reads
//   and ds_write_b128 staging writes are both bank-conflict free.
//   MFMA: v_mfma_f32_16x16x32_bf16 (one per 16-B fragment pair) or 4 x v_mfma_f32_16x16x4_f32 with the
//   K permutation "MFMA s takes element s of every lane group's 4-float vector" (a sum is order-free).
//   Operands are swapped (A = weights, B = pixels) so a lane ends up with 4 CONSECUTIVE output
//   channels of one pixel -> 8/16-byte NHWC stores.
//
// The optional prologue x' = act(x*scale[n,c] + shift[n,c]) is applied between the global load and the
// LDS write, i.e. the consumer normalises InstanceNorm'ed inputs on the fly (padding stays exactly 0).
#include <stdlib.h>

#include <type_traits>
#include "common.h"

namespace eve {

struct GatherParams {
    int N, IH, IW, Cin;     // source tensor [N][IH][IW][Cin]
    int OH, OW, Cout;       // output tensor [N][OH][OW][Cout]
    int KH, KW;
    int o_mul, k_mul, off, div;
    int K;                  // KH*KW*Cin
    uint32_t M;             // N*OH*OW
    FastDiv fd_ohw, fd_ow, fd_cin, fd_kw, fd_oh;
};

template <typename T>
__device__ __forceinline__ uint4 prologue_vec(uint4 q, const float* __restrict__ ssp, int act) {
    constexpr int VEC = Elem<T>::VEC;
    float f[VEC];
    Elem<T>::unpack(q, f);
    const float4* s4 = reinterpret_cast<const float4*>(ssp);   // (scale, shift) pairs, VEC of them
#pragma unroll
    for (int e = 0; e < VEC / 2; ++e) {
        float4 s = s4[e];
        f[2 * e] = act_fwd(f[2 * e] * s.x + s.y, act);
        f[2 * e + 1] = act_fwd(f[2 * e + 1] * s.z + s.w, act);
    }
    return Elem<T>::pack(f);
}

// 16-bit formats: one v_mfma_f32_16x16x32_{bf16,f16} per 16-byte fragment pair (Elem<T>::mfma)
template <typename T> struct Mma {
    __device__ static __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) { Elem<T>::mfma(acc, a, b); }
};
template <> struct Mma<float> {
    __device__ static __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc, 0, 0, 0);
    }
};

}  // namespace eve
#include "conv_fast.h"
#include "conv_wg8.h"
#include "conv_wg8s2.h"
#include "conv_ws64.h"
#include "conv_1x1.h"
#include "conv_3x3s.h"
#include "wgrad_halo.h"
#include "wgrad_wg8.h"
namespace eve {

// =================================================================================================
// forward / dgrad kernel
// =================================================================================================
template <typename T, int NT, bool PRO>
__global__ __launch_bounds__(256) void igemm_kernel(const GatherParams p, const T* __restrict__ src,
                                                    const T* __restrict__ w,
                                                    const float* __restrict__ bias,
                                                    const float* __restrict__ ss, const int pro_act,
                                                    const int epi_act, T* __restrict__ out) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int BM = 128, BN = 32 * NT, BK = 8 * VEC;
    __shared__ uint4 lds[2 * (BM + BN) * 8];

    const int tid = threadIdx.x;
    const uint32_t tiles_n = (p.Cout + BN - 1) / BN;
    const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t m0 = (lid / tiles_n) * BM;
    const uint32_t n0 = (lid % tiles_n) * BN;

    // ---- per-thread staging coordinates (fixed across the K loop) ----
    const int v = tid & 7;       // 16-byte slot inside the 128-byte K row
    const int r0 = tid >> 3;     // 0..31
    int a_y0[4], a_x0[4], a_pix[4], a_n[4];
    bool a_ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t m = m0 + r0 + 32 * j;
        a_ok[j] = m < p.M;
        const uint32_t mm = a_ok[j] ? m : 0;
        const uint32_t n = fd_div(mm, p.fd_ohw);
        const uint32_t rem = mm - n * (uint32_t)(p.OH * p.OW);
        const uint32_t oy = fd_div(rem, p.fd_ow);
        const uint32_t ox = rem - oy * (uint32_t)p.OW;
        a_y0[j] = (int)oy * p.o_mul + p.off;
        a_x0[j] = (int)ox * p.o_mul + p.off;
        a_pix[j] = (int)n * p.IH * p.IW;
        a_n[j] = (int)n;
    }

    uint4 ra[4], rb[NT];

    auto load_tile = [&](int kt) {
        const uint32_t kvec = (uint32_t)kt * BK + (uint32_t)v * VEC;
        const bool kvalid = kvec < (uint32_t)p.K;
        const uint32_t tap = fd_div(kvec, p.fd_cin);
        const int ci = (int)(kvec - tap * (uint32_t)p.Cin);
        const uint32_t kh = fd_div(tap, p.fd_kw);
        const int kw = (int)(tap - kh * (uint32_t)p.KW);
        const int dy = (int)kh * p.k_mul, dx = kw * p.k_mul;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int sy = a_y0[j] + dy, sx = a_x0[j] + dx;
            bool ok = a_ok[j] && kvalid && sy >= 0 && sx >= 0;
            if (p.div > 1) {
                ok = ok && (sy % p.div == 0) && (sx % p.div == 0);
                sy /= p.div;
                sx /= p.div;
            }
            ok = ok && sy < p.IH && sx < p.IW;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (ok) {
                const size_t e = ((size_t)(a_pix[j] + sy * p.IW + sx)) * (size_t)p.Cin + (size_t)ci;
                q = *reinterpret_cast<const uint4*>(src + e);
                if (PRO) q = prologue_vec<T>(q, ss + ((size_t)a_n[j] * p.Cin + ci) * 2, pro_act);
            }
            ra[j] = q;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const uint32_t co = n0 + r0 + 32 * j;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (kvalid && co < (uint32_t)p.Cout)
                q = *reinterpret_cast<const uint4*>(w + (size_t)co * p.K + kvec);
            rb[j] = q;
        }
    };
    auto store_tile = [&](int buf) {
        uint4* base = lds + buf * (BM + BN) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = r0 + 32 * j;
            base[row * 8 + (v ^ (row & 7))] = ra[j];
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int row = r0 + 32 * j;
            base[BM * 8 + row * 8 + (v ^ (row & 7))] = rb[j];
        }
    };

    const int lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int li = lane & 15, lg = lane >> 4;

    f32x4_t acc[4][NT];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // float32 (the 1e-4 rad parity mode): K = 9 x Cin runs to 4 608 products per output, and ONE sequential fmaf chain of that length
    // carries ~sqrt(K) ulp of rounding noise -- 4.3e-6 rad on EyeNet's gaze against 1.6e-6 for a blocked CPU GEMM, which RefineNet
    // amplifies ~20 x (profiles/r06_notes.md 13).  The chain is cut every FLUSH tiles (256 products): partial sums are added into a
    // second accumulator set, as a blocked GEMM does.
    constexpr bool SPLIT_SUM = std::is_same<T, float>::value;
    constexpr int FLUSH = 8;
    f32x4_t tot[SPLIT_SUM ? 4 : 1][SPLIT_SUM ? NT : 1];
    if (SPLIT_SUM) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) tot[SPLIT_SUM ? a : 0][SPLIT_SUM ? b : 0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) load_tile(kt + 1);
        if (SPLIT_SUM && kt > 0 && (kt % FLUSH) == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    tot[SPLIT_SUM ? a : 0][SPLIT_SUM ? b : 0] += acc[a][b];
                    acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
        }
        const uint4* la = lds + cur * (BM + BN) * 8;
        const uint4* lb = la + BM * 8;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int vc = c * 4 + lg;
            uint4 fx[4], fw[NT];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = wm * 64 + mt * 16 + li;
                fx[mt] = la[row * 8 + (vc ^ (row & 7))];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int row = wn * 16 * NT + nt * 16 + li;
                fw[nt] = lb[row * 8 + (vc ^ (row & 7))];
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) Mma<T>::run(acc[mt][nt], fw[nt], fx[mt]);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    if (SPLIT_SUM) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] += tot[SPLIT_SUM ? a : 0][SPLIT_SUM ? b : 0];
    }
    // ---- epilogue: lane holds channels co..co+3 (rows of D) of pixel m (column of D) ----
    const bool vec_ok = (p.Cout & 3) == 0;
    auto epilogue = [&](auto fast) {
    #pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const uint32_t co = n0 + wn * 16 * NT + nt * 16 + lg * 4;
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
                T* dst = out + (size_t)m * p.Cout + co;
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

// =================================================================================================
// weight-gradient kernel:  dw[co][k] += sum_m dy[m][co] * x'[m][k]   (k = (kh,kw,ci))
// The reduction index m is the strided one for both operands, so tiles are staged in LDS as
// [row][channel word] with one 32-bit word per (row, channel): for bf16 a row is a PAIR of pixels
// (even pixel in the low half-word) so 4 words from rows 4g..4g+3 are exactly a lane's 8-deep MFMA
// fragment; for f32 a row is one pixel and the 4 words feed the 4 K-permuted 16x16x4 MFMAs.
// Fragment reads are 4 x ds_read_b32 (row stride = tile width + 4 words -> conflict free).
// Grid: (k tiles of 128) x (co tiles of 32*MT) x (splits of the pixel range); float atomics at the end.
// =================================================================================================
template <typename T, int MT, bool PRO>
__global__ __launch_bounds__(256) void wgrad_kernel(const GatherParams p, const T* __restrict__ x,
                                                    const T* __restrict__ dy,
                                                    const float* __restrict__ ss, const int pro_act,
                                                    float* __restrict__ dw, const uint32_t rows_per_split) {
    constexpr int VEC = Elem<T>::VEC;          // channels per 16-byte global vector
    constexpr int G = (sizeof(T) == 2) ? 2 : 1;  // pixels per LDS row
    constexpr int BCO = 32 * MT, BKK = 128;
    constexpr int PS = BCO + 4, QS = BKK + 4;  // row strides in words
    constexpr int ROWS = 32;                   // LDS rows per step  (= 32*G pixels)
    constexpr int P_ITEMS = ROWS * (BCO / VEC) / 256;   // bf16: MT/2 ... computed below
    constexpr int Q_ITEMS = ROWS * (BKK / VEC) / 256;
    static_assert(ROWS * (BCO / VEC) % 256 == 0 && ROWS * (BKK / VEC) % 256 == 0, "tile/threads");
    __shared__ uint32_t lds[2 * ROWS * (PS + QS)];

    const int tid = threadIdx.x;
    const uint32_t k0 = blockIdx.x * BKK;
    const uint32_t co0 = blockIdx.y * BCO;
    const uint32_t m_begin = blockIdx.z * rows_per_split;
    const uint32_t m_end = min(p.M, m_begin + rows_per_split);
    const uint32_t ohw = (uint32_t)(p.OH * p.OW);

    // ---- per-item constants ----
    int q_row[Q_ITEMS], q_col[Q_ITEMS], q_ci[Q_ITEMS], q_dy[Q_ITEMS], q_dx[Q_ITEMS];
    bool q_kvalid[Q_ITEMS];
#pragma unroll
    for (int j = 0; j < Q_ITEMS; ++j) {
        const int it = tid + 256 * j;
        q_row[j] = it / (BKK / VEC);
        q_col[j] = (it % (BKK / VEC)) * VEC;
        const uint32_t kvec = k0 + q_col[j];
        q_kvalid[j] = kvec < (uint32_t)p.K;
        const uint32_t tap = fd_div(kvec, p.fd_cin);
        q_ci[j] = (int)(kvec - tap * (uint32_t)p.Cin);
        const uint32_t kh = fd_div(tap, p.fd_kw);
        q_dy[j] = (int)kh * p.k_mul + p.off;
        q_dx[j] = (int)(tap - kh * (uint32_t)p.KW) * p.k_mul + p.off;
    }
    int p_row[P_ITEMS], p_col[P_ITEMS];
#pragma unroll
    for (int j = 0; j < P_ITEMS; ++j) {
        const int it = tid + 256 * j;
        p_row[j] = it / (BCO / VEC);
        p_col[j] = (it % (BCO / VEC)) * VEC;
    }

    uint4 rq[Q_ITEMS][G], rp[P_ITEMS][G];

    auto load_tile = [&](uint32_t mbase) {
#pragma unroll
        for (int j = 0; j < Q_ITEMS; ++j) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t m = mbase + q_row[j] * G + g;
                uint4 q = make_uint4(0, 0, 0, 0);
                if (m < m_end && q_kvalid[j]) {
                    const uint32_t n = fd_div(m, p.fd_ohw);
                    const uint32_t rem = m - n * ohw;
                    const uint32_t oy = fd_div(rem, p.fd_ow);
                    const uint32_t ox = rem - oy * (uint32_t)p.OW;
                    const int sy = (int)oy * p.o_mul + q_dy[j];
                    const int sx = (int)ox * p.o_mul + q_dx[j];
                    if (sy >= 0 && sy < p.IH && sx >= 0 && sx < p.IW) {
                        const size_t e = ((size_t)n * p.IH * p.IW + (size_t)(sy * p.IW + sx)) * p.Cin + q_ci[j];
                        q = *reinterpret_cast<const uint4*>(x + e);
                        if (PRO) q = prologue_vec<T>(q, ss + ((size_t)n * p.Cin + q_ci[j]) * 2, pro_act);
                    }
                }
                rq[j][g] = q;
            }
        }
#pragma unroll
        for (int j = 0; j < P_ITEMS; ++j) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t m = mbase + p_row[j] * G + g;
                const uint32_t co = co0 + p_col[j];
                uint4 q = make_uint4(0, 0, 0, 0);
                if (m < m_end && co < (uint32_t)p.Cout)
                    q = *reinterpret_cast<const uint4*>(dy + (size_t)m * p.Cout + co);
                rp[j][g] = q;
            }
        }
    };
    // write a (row, 8|4 channels) item as words; bf16: interleave the even/odd pixel
    auto put = [&](uint32_t* dst, const uint4* r) {
        if (G == 2) {
            const uint4 a = r[0], b = r[G - 1];
            uint4 lo, hi;
            lo.x = (a.x & 0xffffu) | (b.x << 16);  lo.y = (a.x >> 16) | (b.x & 0xffff0000u);
            lo.z = (a.y & 0xffffu) | (b.y << 16);  lo.w = (a.y >> 16) | (b.y & 0xffff0000u);
            hi.x = (a.z & 0xffffu) | (b.z << 16);  hi.y = (a.z >> 16) | (b.z & 0xffff0000u);
            hi.z = (a.w & 0xffffu) | (b.w << 16);  hi.w = (a.w >> 16) | (b.w & 0xffff0000u);
            reinterpret_cast<uint4*>(dst)[0] = lo;
            reinterpret_cast<uint4*>(dst)[1] = hi;
        } else {
            reinterpret_cast<uint4*>(dst)[0] = r[0];
        }
    };
    auto store_tile = [&](int buf) {
        uint32_t* pb = lds + buf * ROWS * (PS + QS);
        uint32_t* qb = pb + ROWS * PS;
#pragma unroll
        for (int j = 0; j < P_ITEMS; ++j) put(pb + p_row[j] * PS + p_col[j], rp[j]);
#pragma unroll
        for (int j = 0; j < Q_ITEMS; ++j) put(qb + q_row[j] * QS + q_col[j], rq[j]);
    };

    const int lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int li = lane & 15, lg = lane >> 4;

    f32x4_t acc[MT][4];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    constexpr uint32_t STEP = ROWS * G;    // pixels per step
    if (m_begin < m_end) {
        const int nsteps = (int)((m_end - m_begin + STEP - 1) / STEP);
        load_tile(m_begin);
        store_tile(0);
        __syncthreads();
        for (int st = 0; st < nsteps; ++st) {
            const int cur = st & 1;
            const bool more = st + 1 < nsteps;
            if (more) load_tile(m_begin + (uint32_t)(st + 1) * STEP);
            const uint32_t* pb = lds + cur * ROWS * (PS + QS);
            const uint32_t* qb = pb + ROWS * PS;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int rbase = c * 16 + lg * 4;
                uint4 fp[MT], fq[4];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const uint32_t* s = pb + rbase * PS + wm * 16 * MT + mt * 16 + li;
                    fp[mt] = make_uint4(s[0], s[PS], s[2 * PS], s[3 * PS]);
                }
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const uint32_t* s = qb + rbase * QS + wn * 64 + kt * 16 + li;
                    fq[kt] = make_uint4(s[0], s[QS], s[2 * QS], s[3 * QS]);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) Mma<T>::run(acc[mt][kt], fp[mt], fq[kt]);
            }
            if (more) store_tile(cur ^ 1);
            __syncthreads();
        }
    }
    // D rows = co (4*lg + r), cols = k (li)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const uint32_t k = k0 + wn * 64 + kt * 16 + li;
            if (k >= (uint32_t)p.K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t co = co0 + wm * 16 * MT + mt * 16 + lg * 4 + r;
                if (co < (uint32_t)p.Cout) atomicAdd(dw + (size_t)co * p.K + k, acc[mt][kt][r]);
            }
        }
}

// column sums of dy[M][C] accumulated into db[C]
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(const T* __restrict__ dy, float* __restrict__ db,
                                                        long long M, int C, long long rows_per_block) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = C / VEC;                 // <= 256 assumed by the launcher
    const int phases = 256 / cvecs;
    const int tid = threadIdx.x;
    const int cv = tid % cvecs, ph = tid / cvecs;
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    const long long r_begin = (long long)blockIdx.x * rows_per_block;
    const long long r_end = min(M, r_begin + rows_per_block);
    if (ph < phases) {
#pragma unroll 4
        for (long long r = r_begin + ph; r < r_end; r += phases) {
            float f[VEC];
            Elem<T>::unpack(*reinterpret_cast<const uint4*>(dy + r * C + cv * VEC), f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e] += f[e];
        }
    }
    // one atomic per channel and workgroup: with few channels (RefineNet: 16..64) thousands of workgroups hammering
    // the same handful of addresses from every thread serialised into milliseconds per launch
    __shared__ float red[256 * VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[tid * VEC + e] = ph < phases ? s[e] : 0.f;
    __syncthreads();
    // pairwise tree over the phases (two threads walking 128 partial rows one after the other cost more than the
    // loads of a whole workgroup on 16-channel tensors)
    for (int n = phases; n > 1;) {
        const int half = (n + 1) >> 1;
        if (ph + half < n) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) red[tid * VEC + e] += red[(tid + half * cvecs) * VEC + e];
        }
        __syncthreads();
        n = half;
    }
    if (tid < cvecs) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) atomicAdd(db + tid * VEC + e, red[tid * VEC + e]);
    }
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
static int check_desc(const eve_conv_desc* d, int vec) {
    if (!d) return set_error_msg("conv: null descriptor");
    if ((unsigned)d->dtype > (unsigned)EVE_DT_F16) return set_error_msg("conv: bad dtype");
    if (d->N <= 0 || d->IH <= 0 || d->IW <= 0 || d->OH <= 0 || d->OW <= 0 || d->Cin <= 0 || d->Cout <= 0 ||
        d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->pad < 0)
        return set_error_msg("conv: non-positive dimension");
    if (d->OH != (d->IH + 2 * d->pad - d->KH) / d->stride + 1 || d->OW != (d->IW + 2 * d->pad - d->KW) / d->stride + 1)
        return set_error_msg("conv: OH/OW inconsistent with IH/IW, kernel, stride, pad");
    if (d->Cin % vec) return set_error_msg("conv: Cin must be a multiple of the 16-byte vector (4 f32 / 8 bf16)");
    if ((long long)d->N * d->OH * d->OW >= (1ll << 31) || (long long)d->N * d->IH * d->IW >= (1ll << 31))
        return set_error_msg("conv: more than 2^31 pixels");
    return 0;
}

static GatherParams fwd_params(const eve_conv_desc* d) {
    GatherParams p;
    p.N = d->N; p.IH = d->IH; p.IW = d->IW; p.Cin = d->Cin;
    p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW;
    p.o_mul = d->stride; p.k_mul = 1; p.off = -d->pad; p.div = 1;
    p.K = d->KH * d->KW * d->Cin;
    p.M = (uint32_t)((long long)d->N * d->OH * d->OW);
    p.fd_ohw = make_fastdiv(d->OH * d->OW); p.fd_ow = make_fastdiv(d->OW);
    p.fd_cin = make_fastdiv(d->Cin); p.fd_kw = make_fastdiv(d->KW); p.fd_oh = make_fastdiv(d->OH);
    return p;
}
static GatherParams dgrad_params(const eve_conv_desc* d) {
    GatherParams p;   // source = dy [N][OH][OW][Cout], output = dx [N][IH][IW][Cin]
    p.N = d->N; p.IH = d->OH; p.IW = d->OW; p.Cin = d->Cout;
    p.OH = d->IH; p.OW = d->IW; p.Cout = d->Cin; p.KH = d->KH; p.KW = d->KW;
    p.o_mul = 1; p.k_mul = -1; p.off = d->pad; p.div = d->stride;
    p.K = d->KH * d->KW * d->Cout;
    p.M = (uint32_t)((long long)d->N * d->IH * d->IW);
    p.fd_ohw = make_fastdiv(d->IH * d->IW); p.fd_ow = make_fastdiv(d->IW);
    p.fd_cin = make_fastdiv(d->Cout); p.fd_kw = make_fastdiv(d->KW); p.fd_oh = make_fastdiv(d->IH);
    return p;
}

static bool use_v1() {
    return g_cfg.conv_impl_v1 == 1;
}

template <typename T>
static void launch_dma_one(const GatherParams& p, const TapPlan& tp, const void* src, const void* w,
                           const float* bias, int epi_act, void* out, uint32_t src_bytes, uint32_t w_bytes,
                           hipStream_t s) {
    const T* a = (const T*)src; const T* b = (const T*)w; T* o = (T*)out;
    const int big = g_cfg.conv_tile_big;
    if (p.Cout > 64 && big && p.M >= 256 * 256) {
        const uint32_t tiles = ((p.M + 255) / 256) * ((p.Cout + 127) / 128);
        EVE_LAUNCH("igemm_dma_kernel<T, 4, 2>", (igemm_dma_kernel<T, 4, 2>), dim3(tiles), dim3(512), 0, s, p, a, b, bias, epi_act, o,
                           src_bytes, w_bytes, tp);
    } else if (p.Cout > 64) {
        const uint32_t tiles = ((p.M + 127) / 128) * ((p.Cout + 127) / 128);
        EVE_LAUNCH("igemm_dma_kernel<T, 2, 2>", (igemm_dma_kernel<T, 2, 2>), dim3(tiles), dim3(256), 0, s, p, a, b, bias, epi_act, o,
                           src_bytes, w_bytes, tp);
    } else {
        const uint32_t tiles = (p.M + 255) / 256;
        EVE_LAUNCH("igemm_dma_kernel<T, 4, 1>", (igemm_dma_kernel<T, 4, 1>), dim3(tiles), dim3(256), 0, s, p, a, b, bias, epi_act, o,
                           src_bytes, w_bytes, tp);
    }
}

// LDS-DMA path: no prologue, channel count a multiple of the K step, 32-bit byte offsets.  Strided data
// gradients are decomposed into stride*stride dense sub-problems (one per output parity class).
// 3x3 / stride 1 / pad 1, bf16, halo-resident kernel.  Returns false when the shape does not qualify.
template <typename HT>
static bool launch_halo(const GatherParams& p, const void* src, const void* w, const float* bias, int epi_act,
                        void* out, hipStream_t s) {
    if (!g_cfg.conv_halo) return false;
    const bool fwd = p.k_mul == 1 && p.off == -1, bwd = p.k_mul == -1 && p.off == 1;
    if (p.KH != 3 || p.KW != 3 || p.div != 1 || p.o_mul != 1 || !(fwd || bwd) || p.OH != p.IH || p.OW != p.IW) return false;
    const int W = p.IW, H = p.IH;
    if (p.Cin % 32 || W > 128 || (W & (W - 1)) || W < 4) return false;
    HaloParams h;
    h.N = p.N; h.H = H; h.W = W; h.Cin = p.Cin; h.Cout = p.Cout;
    // ---- ResNet layer 1 (32 x 32 | 64 x 64 images, 64 -> 64 channels): filter bank resident in LDS, streaming tiles (conv_ws64.h) ----
    const int ws64 = g_cfg.conv_ws64;
    if (ws64 && H == W && (W == 32 || W == 64) && p.Cin == 64 && p.Cout == 64 && ((epi_act & ~0xff) == 0) &&
        ((epi_act & 0xff) == EVE_ACT_NONE || (epi_act & 0xff) == EVE_ACT_RELU) && (unsigned long long)p.N * W * W * 128ull < (1ull << 31)) {
        Ws64Params q;
        q.N = p.N; q.flip = bwd ? 1 : 0; q.x_bytes = (uint32_t)((unsigned long long)p.N * W * W * 128ull); q.w_bytes = 64u * 576u * 2u;
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)conv3x3_ws64_kernel<HT, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv3x3_ws64_kernel<HT, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_done = true;
        }
        const uint32_t T64 = (uint32_t)(W * W / 512) * (uint32_t)p.N;
        if (W == 32)
            EVE_LAUNCH(EVE_HNAME(HT, "conv3x3_ws64_kernel<", ", 32>"), (conv3x3_ws64_kernel<HT, 32>), dim3(T64 < 256u ? T64 : 256u), dim3(512),
                       (size_t)Ws64Geom<32>::LDS, s, q, (const HT*)src, (const HT*)w, bias, epi_act, (HT*)out);
        else
            EVE_LAUNCH(EVE_HNAME(HT, "conv3x3_ws64_kernel<", ", 64>"), (conv3x3_ws64_kernel<HT, 64>), dim3(T64 < 256u ? T64 : 256u), dim3(512),
                       (size_t)Ws64Geom<64>::LDS, s, q, (const HT*)src, (const HT*)w, bias, epi_act, (HT*)out);
        return true;
    }
    // ---- eight-wave workgroups, 32x32x16 MFMA, staggered wave groups (conv_wg8.h): whole square images per tile ----
    const int wg8 = g_cfg.conv_wg8;
    // one 512-thread workgroup per CU: below ~7/8 of the CUs' worth of tiles (B = 8 clips per GPU: 60-120 tiles) the
    // four-wave kernels with their 128-pixel tiles fill the chip better
    const int wg8_min_tiles = g_cfg.conv_wg8_min_tiles;
    if (wg8 && H == W && p.Cin % 64 == 0 && ((epi_act & ~0xff) == 0) &&
        ((epi_act & 0xff) == EVE_ACT_NONE || (epi_act & 0xff) == EVE_ACT_RELU)) {
        const unsigned long long xb8 = (unsigned long long)p.N * H * W * p.Cin * 2, wb8 = (unsigned long long)p.Cout * p.K * 2;
        Wg8Params g;
        g.N = p.N; g.Cin = p.Cin; g.Cout = p.Cout; g.flip = bwd ? 1 : 0; g.K = p.K; g.x_bytes = (uint32_t)xb8; g.w_bytes = (uint32_t)wb8; g.s2_py = 0;
#define EVE_WG8_LAUNCH(WM_, WN_, W_)                                                                                      \
        do {                                                                                                             \
            using G8 = Wg8Geom<WM_, WN_, W_>;                                                                              \
            static bool attr_done = false;                                                                               \
            if (!attr_done) {                                                                                            \
                (void)hipFuncSetAttribute((const void*)conv3x3_wg8_kernel<HT, WM_, WN_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                attr_done = true;                                                                                        \
            }                                                                                                            \
            g.tiles_n = (uint32_t)(p.Cout / G8::COUT_T);                                                                   \
            const uint32_t tiles8 = (uint32_t)((p.N + G8::TI - 1) / G8::TI) * g.tiles_n;                                     \
            if ((int)tiles8 < wg8_min_tiles) break;      /* too few whole-CU workgroups (small batches): four-wave kernels */ \
            EVE_LAUNCH(EVE_HNAME(HT, "conv3x3_wg8_kernel<", ", " #WM_ ", " #WN_ ", " #W_ ">"), (conv3x3_wg8_kernel<HT, WM_, WN_, W_>), dim3(tiles8), \
                       dim3(512), G8::LDS, s, g, (const HT*)src, (const HT*)w, bias, epi_act, (HT*)out);                   \
            return true;                                                                                                 \
        } while (0)
        if (xb8 < (1ull << 31) && wb8 < (1ull << 31)) {
            // 16 x 16 images x 128 channels (512-pixel tiles, 36 steps each): a one-tile workgroup per CU spends ~15 % of its
            // time in the un-overlapped prologue / epilogue; since the epilogue stores 64 contiguous bytes per lane quad it is
            // ahead of the four-wave kernel there too (0.141 against 0.151-0.166 ms; EVE_CONV_WG8=3 keeps the four-wave kernel)
            if (wg8 != 3 && W == 16 && p.Cout % 128 == 0 && p.Cout % 256 != 0) EVE_WG8_LAUNCH(4, 2, 16);
            // 16 x 16 x 256 (the trunk's layer 3 on 256 x 256 patches, BASELINE configs[4]): one image x 256 channels per tile
            if (W == 16 && p.Cout % 256 == 0) EVE_WG8_LAUNCH(2, 4, 16);
            if (W == 8 && p.Cout % 256 == 0) EVE_WG8_LAUNCH(2, 4, 8);
            if (W == 4 && p.Cout % 256 == 0) EVE_WG8_LAUNCH(2, 4, 4);
            // 32 x 32 x 128 (the trunk's layer 2 on 256 x 256 patches, BASELINE configs[4]): half-image bands of 16 rows x 128 channels
            if (W == 32 && p.Cout % 128 == 0) {
                using G8 = Wg8Geom<4, 2, 32, 2>;
                static bool attr_done = false;
                if (!attr_done) {
                    (void)hipFuncSetAttribute((const void*)conv3x3_wg8_kernel<HT, 4, 2, 32, 9, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    attr_done = true;
                }
                g.tiles_n = (uint32_t)(p.Cout / G8::COUT_T);
                const uint32_t tiles8 = (uint32_t)p.N * 2u * g.tiles_n;
                if ((int)tiles8 >= wg8_min_tiles) {
                    EVE_LAUNCH(EVE_HNAME(HT, "conv3x3_wg8_kernel<", ", 4, 2, 32, 9, 2>"), (conv3x3_wg8_kernel<HT, 4, 2, 32, 9, 2>), dim3(tiles8),
                               dim3(512), G8::LDS, s, g, (const HT*)src, (const HT*)w, bias, epi_act, (HT*)out);
                    return true;
                }
            }
        }
#undef EVE_WG8_LAUNCH
    }
    const bool narrow = p.Cout <= 64;               // 256 pixels x 64 channels (4x1 waves) instead of 128 x 128 (2x2)
    const int BMp = narrow ? 256 : 128;
    const int rows = BMp / W;
    if (rows < 1) return false;
    if (rows <= H) { h.TI = 1; h.TH = rows; h.bands = (H + rows - 1) / rows; }
    else { if (rows % H) return false; h.TI = rows / H; h.TH = H; h.bands = 1; }
    const int HP = h.TI * (h.TH + 2) * (W + 2);
    h.a_pieces = (HP * 4 + 255) / 256;
    if (h.a_pieces > 7) return false;
    const unsigned long long xb = (unsigned long long)p.N * H * W * p.Cin * 2, wb = (unsigned long long)p.Cout * p.K * 2;
    if (xb >= (1ull << 31) || wb >= (1ull << 31)) return false;
    h.flip = bwd ? 1 : 0; h.K = p.K; h.x_bytes = (uint32_t)xb; h.w_bytes = (uint32_t)wb;
    h.tiles_m = h.TI == 1 ? (uint32_t)p.N * h.bands : (uint32_t)((p.N + h.TI - 1) / h.TI);
    h.tiles_n = narrow ? 1 : (p.Cout + 127) / 128;
    h.fd_w2 = make_fastdiv(W + 2); h.fd_hpi = make_fastdiv((h.TH + 2) * (W + 2)); h.fd_w = make_fastdiv(W); h.fd_th = make_fastdiv(h.TH);
    const size_t lds = 2 * (size_t)h.a_pieces * 4096 + 4 * (narrow ? 4096 : 8192);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<HT, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<HT, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int persist = g_cfg.halo_persist;
    const uint32_t tiles = h.tiles_m * h.tiles_n;
    // (pays where a tile is short -- 18 steps at 64 channels; from 128 channels on the one-tile kernel is as fast)
    if (persist && tiles > 512 && p.Cin <= 64 && p.Cout % 8 == 0 && (epi_act == EVE_ACT_NONE || epi_act == EVE_ACT_RELU)) {           // two resident workgroups per CU walk the tiles as one stream
        const size_t plds = lds + (bias ? (size_t)h.tiles_n * (narrow ? 64 : 128) * 4 : 0);
        static bool pattr = false;
        if (!pattr) {
            (void)hipFuncSetAttribute((const void*)conv3x3_halo_pkernel<HT, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv3x3_halo_pkernel<HT, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            pattr = true;
        }
        if (narrow)
            EVE_LAUNCH(EVE_HNAME(HT, "conv3x3_halo_pkernel<", ", 4, 1>"), (conv3x3_halo_pkernel<HT, 4, 1>), dim3(512), dim3(256), plds, s, h, (const HT*)src,
                               (const HT*)w, bias, epi_act, (HT*)out);
        else
            EVE_LAUNCH(EVE_HNAME(HT, "conv3x3_halo_pkernel<", ", 2, 2>"), (conv3x3_halo_pkernel<HT, 2, 2>), dim3(512), dim3(256), plds, s, h, (const HT*)src,
                               (const HT*)w, bias, epi_act, (HT*)out);
        return true;
    }
    if (narrow)
        EVE_LAUNCH(EVE_HNAME(HT, "conv3x3_halo_kernel<", ", 4, 1>"), (conv3x3_halo_kernel<HT, 4, 1>), dim3(tiles), dim3(256), lds, s, h, (const HT*)src,
                           (const HT*)w, bias, epi_act, (HT*)out);
    else
        EVE_LAUNCH(EVE_HNAME(HT, "conv3x3_halo_kernel<", ", 2, 2>"), (conv3x3_halo_kernel<HT, 2, 2>), dim3(tiles), dim3(256), lds, s, h, (const HT*)src,
                           (const HT*)w, bias, epi_act, (HT*)out);
    return true;
}

template <typename T>
static bool launch_igemm_dma(const GatherParams& p, const void* src, const void* w, const float* bias,
                             int epi_act, void* out, hipStream_t s) {
    constexpr int BK = 8 * Elem<T>::VEC;
    if constexpr (sizeof(T) == 2) { if (launch_halo<T>(p, src, w, bias, epi_act, out, s)) return true; }
    const unsigned long long src_bytes = (unsigned long long)p.N * p.IH * p.IW * p.Cin * sizeof(T);
    const unsigned long long w_bytes = (unsigned long long)p.Cout * p.K * sizeof(T);
    if (p.Cin % BK != 0 || p.KH * p.KW > 32 || src_bytes >= (1ull << 31) || w_bytes >= (1ull << 31)) return false;
    if (p.div == 1) {
        TapPlan tp;
        tp.ntaps = p.KH * p.KW;
        for (int kh = 0; kh < p.KH; ++kh)
            for (int kw = 0; kw < p.KW; ++kw) {
                const int t = kh * p.KW + kw;
                tp.dy[t] = (signed char)(kh * p.k_mul); tp.dx[t] = (signed char)(kw * p.k_mul); tp.wt[t] = (signed char)t;
            }
        tp.osy = tp.osx = 1; tp.oy0 = tp.ox0 = 0; tp.OHf = p.OH; tp.OWf = p.OW;
        launch_dma_one<T>(p, tp, src, w, bias, epi_act, out, (uint32_t)src_bytes, (uint32_t)w_bytes, s);
        return true;
    }
    // data gradient of a stride-`div` convolution (o_mul = 1, k_mul = -1, off = pad): output pixel iy = div*y' + py
    // takes tap kh iff (py + pad - kh) % div == 0, from source row y' + (py + pad - kh) / div
    if (p.o_mul != 1 || p.k_mul != -1 || bias || (epi_act & 0xff) != EVE_ACT_NONE) return false;
    const int sd = p.div;
    bool need_zero = false;
    for (int pass = 0; pass < 2; ++pass) {
        for (int py = 0; py < sd; ++py)
            for (int px = 0; px < sd; ++px) {
                TapPlan tp;
                tp.ntaps = 0;
                for (int kh = 0; kh < p.KH; ++kh)
                    for (int kw = 0; kw < p.KW; ++kw) {
                        const int ny = py + p.off - kh, nx = px + p.off - kw;
                        if (ny % sd != 0 || nx % sd != 0) continue;
                        tp.dy[tp.ntaps] = (signed char)(ny / sd); tp.dx[tp.ntaps] = (signed char)(nx / sd);
                        tp.wt[tp.ntaps] = (signed char)(kh * p.KW + kw);
                        ++tp.ntaps;
                    }
                const int sub_h = (p.OH - py + sd - 1) / sd, sub_w = (p.OW - px + sd - 1) / sd;
                if (sub_h <= 0 || sub_w <= 0) continue;
                if (tp.ntaps == 0) { need_zero = true; continue; }
                if (pass == 0) continue;
                GatherParams q = p;
                q.OH = sub_h; q.OW = sub_w;
                q.o_mul = 1; q.off = 0; q.div = 1;
                q.M = (uint32_t)((long long)p.N * sub_h * sub_w);
                q.fd_ohw = make_fastdiv(sub_h * sub_w); q.fd_ow = make_fastdiv(sub_w);
                tp.osy = tp.osx = sd; tp.oy0 = py; tp.ox0 = px; tp.OHf = p.OH; tp.OWf = p.OW;
                launch_dma_one<T>(q, tp, src, w, bias, epi_act, out, (uint32_t)src_bytes, (uint32_t)w_bytes, s);
            }
        if (pass == 0 && need_zero && !(epi_act & EVE_EPI_ACC))     // accumulating: untouched classes keep their value
            (void)hipMemsetAsync(out, 0, (size_t)p.N * p.OH * p.OW * p.Cout * sizeof(T), s);
    }
    return true;
}

template <typename T>
static int launch_igemm(const GatherParams& p, const void* src, const void* w, const float* bias,
                        const float* ss, int pro_act, int epi_act, void* out, hipStream_t s) {
    if (!ss && !use_v1() && launch_igemm_dma<T>(p, src, w, bias, epi_act, out, s)) return 0;
    const bool wide = p.Cout > 64;
    const uint32_t bn = wide ? 128 : 64;
    const uint32_t tiles = ((p.M + 127) / 128) * ((p.Cout + bn - 1) / bn);
    dim3 grid(tiles), block(256);
    const T* a = (const T*)src; const T* b = (const T*)w; T* o = (T*)out;
    if (wide) {
        if (ss) EVE_LAUNCH("igemm_kernel<T, 4, true>", (igemm_kernel<T, 4, true>), grid, block, 0, s, p, a, b, bias, ss, pro_act, epi_act, o);
        else    EVE_LAUNCH("igemm_kernel<T, 4, false>", (igemm_kernel<T, 4, false>), grid, block, 0, s, p, a, b, bias, ss, pro_act, epi_act, o);
    } else {
        if (ss) EVE_LAUNCH("igemm_kernel<T, 2, true>", (igemm_kernel<T, 2, true>), grid, block, 0, s, p, a, b, bias, ss, pro_act, epi_act, o);
        else    EVE_LAUNCH("igemm_kernel<T, 2, false>", (igemm_kernel<T, 2, false>), grid, block, 0, s, p, a, b, bias, ss, pro_act, epi_act, o);
    }
    return 0;
}

static void wgrad_split(const GatherParams& p, uint32_t tk, uint32_t tc, size_t lds_per_wg, uint32_t& splits, uint32_t& rows) {
    // As many splits of the pixel range as fill the chip ONCE: every workgroup ends with one float atomic per element of its
    // filter tile, and those run at one dword per L2 channel and clock (~270 G/s: 17 M atomics of a 1 024-workgroup launch =
    // 60 us), so a second, partial round of workgroups costs its atomics and a tail.  Resident workgroups per CU: LDS-bound,
    // at most 3 by registers.  (Layers 3 / 4: 0.184 / 0.175 -> 0.160 / 0.148 ms against the former fixed target of 1 024.)
    const int target_env = g_cfg.wgrad_target_wgs;
    uint32_t per_cu = lds_per_wg ? (uint32_t)((160 * 1024) / lds_per_wg) : 3;
    per_cu = per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu);
    const uint32_t target = target_env > 0 ? (uint32_t)target_env : 256u * per_cu;
    uint32_t want = target / (tk * tc);
    // ... but no split shorter than ~48 K-steps: prologue, ring fill and the 64 atomics per thread of the epilogue are per
    // workgroup (at B=8 clips the unbounded split cost 4 % of the step; B=32 is not affected)
    const int min_rows = g_cfg.wgrad_min_rows;
    uint32_t max_splits = (p.M + (uint32_t)min_rows - 1) / (uint32_t)min_rows;
    splits = want < 1 ? 1 : (want > max_splits ? max_splits : want);
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    rows = (p.M + splits - 1) / splits;
    rows = (rows + 63) / 64 * 64;
    splits = (p.M + rows - 1) / rows;
}

// few-channel 3x3 / stride 1 / pad 1 layers on large planes: band-resident kernel (wgrad_halo.h).  False = not this shape.
template <typename HT>
static bool launch_wgrad_halo(const GatherParams& p, const void* x, const void* dy, float* dw, float* db, hipStream_t s) {
    if (!g_cfg.wgrad_halo) return false;
    const int ks = p.KH;
    if ((ks != 3 && ks != 1) || p.KW != ks || p.o_mul != 1 || p.k_mul != 1 || p.off != -(ks / 2) || p.div != 1 || p.OH != p.IH ||
        p.OW != p.IW)
        return false;
    const int W = p.IW, H = p.IH;
    const long long min_m = g_cfg.wgrad_halo_min_m;   // (1 M pixels: below, the gather kernel's re-reads stay in L2 anyway)
    if (W < 32 || W > 128 || (W & (W - 1)) || p.Cin % 16 || p.Cout % 16 || (long long)p.M < min_m) return false;
    const int MT = p.Cout / 16, CT = p.Cin / 16;
    const bool split = ks == 3 && MT == 4 && CT == 4;        // 64 -> 64 channels: wgrad_halo64_kernel (two bands resident)
    if (ks == 3 && !split && !((MT == 1 && (CT == 1 || CT == 2 || CT == 4)) || (MT == 2 && (CT == 1 || CT == 2)))) return false;
    if (ks == 1 && !((MT == 2 && CT == 1) || (MT == 1 && CT == 4) || (MT == 1 && CT == 2) || (MT == 4 && CT == 2) || (MT == 2 && CT == 4)))
        return false;
    const unsigned long long xb = (unsigned long long)p.N * H * W * p.Cin * 2, db_ = (unsigned long long)p.M * p.Cout * 2;
    if (xb >= (1ull << 31) || db_ >= (1ull << 31)) return false;
    const size_t red = (size_t)(ks * ks * p.Cin * p.Cout + p.Cout) * 4;
    const unsigned occ = (ks == 3 && MT * CT >= 4) ? 2 : 3;   // workgroups per CU the accumulator registers allow (36 tiles: two)
    int TH = 0; size_t lds = 0;
    int nreg = 0;
    if (split) {
        // wgrad_halo64_kernel: one workgroup per CU with 2 or 3 resident bands (and room for the epilogue's 9 x 64 x 64 + 64 floats)
        const int force_th = g_cfg.wg64_th, force_nreg = g_cfg.wg64_nreg;
        for (int th : {8, 4, 2, 1}) {
            if (force_th && th != force_th) continue;
            const size_t need = (size_t)(th + 2) * (W + 2) * 128 + (size_t)th * W * 128;
            for (int nr : {3, 2}) {
                if (force_nreg && nr != force_nreg) continue;
                if (nr * need <= (size_t)160 * 1024) { TH = th; nreg = nr; lds = nr * need > red ? nr * need : red; break; }
            }
            if (TH) break;
        }
    } else
    for (size_t budget : {(size_t)(occ == 3 ? 52 : 78) * 1024, (size_t)78 * 1024}) {
        for (int th : {4, 2, 1}) {
            const size_t need = (size_t)(th + ks - 1) * (W + ks - 1) * p.Cin * 2 + (size_t)th * W * p.Cout * 2;
            if (need <= budget && need >= red) { TH = th; lds = need; break; }
        }
        if (TH) break;
    }
    if (!TH) return false;
    WgradHaloParams h;
    h.N = p.N; h.H = H; h.W = W; h.TH = TH; h.bands = (H + TH - 1) / TH;
    h.total_bands = (uint32_t)p.N * h.bands; h.x_bytes = (uint32_t)xb; h.dy_bytes = (uint32_t)db_;
    h.log2_cpr = 0; h.nreg = nreg;
    while ((32 << h.log2_cpr) < W) ++h.log2_cpr;
    const unsigned per_cu = (unsigned)((160 * 1024) / lds);
    unsigned grid = 256 * (per_cu > occ ? occ : per_cu);
    if (grid > h.total_bands) grid = h.total_bands;
#define EVE_WGRAD_HALO_LAUNCH(MT_, CT_, KS_)                                                                            \
    do {                                                                                                                \
        static bool attr_done = false;                                                                                  \
        if (!attr_done) {                                                                                               \
            (void)hipFuncSetAttribute((const void*)wgrad_halo_kernel<HT, MT_, CT_, KS_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_done = true;                                                                                           \
        }                                                                                                               \
        EVE_LAUNCH(EVE_HNAME(HT, "wgrad_halo_kernel<", ", " #MT_ ", " #CT_ ", " #KS_ ">"), (wgrad_halo_kernel<HT, MT_, CT_, KS_>), dim3(grid), dim3(256), lds, s, h, \
                   (const HT*)x, (const HT*)dy, dw, db);                                                                    \
    } while (0)
    if (split) {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)wgrad_halo64_kernel<HT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad_halo64_kernel<HT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_done = true;
        }
        const unsigned g64 = h.total_bands < 256u ? h.total_bands : 256u;
        const int fixed64 = g_cfg.wg64_fixed;
        if (fixed64 && W == 32 && TH == 8 && H % 8 == 0)     // ResNet layer 1: unrolled band loop, immediate fragment addresses
            EVE_LAUNCH(EVE_HNAME(HT, "wgrad_halo64_kernel<", ", fixed>"), (wgrad_halo64_kernel<HT, true>), dim3(g64), dim3(512), lds, s, h, (const HT*)x, (const HT*)dy, dw, db);
        else
            EVE_LAUNCH(EVE_HNAME(HT, "wgrad_halo64_kernel<", ">"), (wgrad_halo64_kernel<HT, false>), dim3(g64), dim3(512), lds, s, h, (const HT*)x, (const HT*)dy, dw, db);
    } else if (ks == 3) {
        if (MT == 1 && CT == 1) EVE_WGRAD_HALO_LAUNCH(1, 1, 3);
        else if (MT == 1 && CT == 2) EVE_WGRAD_HALO_LAUNCH(1, 2, 3);
        else if (MT == 1 && CT == 4) EVE_WGRAD_HALO_LAUNCH(1, 4, 3);
        else if (MT == 2 && CT == 1) EVE_WGRAD_HALO_LAUNCH(2, 1, 3);
        else EVE_WGRAD_HALO_LAUNCH(2, 2, 3);
    } else {
        if (MT == 2 && CT == 1) EVE_WGRAD_HALO_LAUNCH(2, 1, 1);
        else if (MT == 1 && CT == 4) EVE_WGRAD_HALO_LAUNCH(1, 4, 1);
        else if (MT == 1 && CT == 2) EVE_WGRAD_HALO_LAUNCH(1, 2, 1);
        else if (MT == 4 && CT == 2) EVE_WGRAD_HALO_LAUNCH(4, 2, 1);
        else EVE_WGRAD_HALO_LAUNCH(2, 4, 1);
    }
#undef EVE_WGRAD_HALO_LAUNCH
    return true;
}

template <typename T>
static int launch_wgrad(const GatherParams& p, const void* x, const void* dy, const float* ss, int pro_act,
                        float* dw, hipStream_t s, float* db = nullptr, void* workspace = nullptr, unsigned long long workspace_bytes = 0) {
    if constexpr (sizeof(T) == 2) {
    if (!ss && !use_v1() && launch_wgrad_halo<T>(p, x, dy, dw, db, s)) return db ? 1 : 0;
    if (!ss && !use_v1()) {   // bf16: LDS-DMA staging + hardware-transposing fragment reads
        const unsigned long long x_bytes = (unsigned long long)p.N * p.IH * p.IW * p.Cin * 2;
        const unsigned long long dy_bytes = (unsigned long long)p.M * p.Cout * 2;
        if (x_bytes < (1ull << 31) && dy_bytes < (1ull << 31)) {
            uint32_t splits, rows;
            const bool pow2 = ((p.OW & (p.OW - 1)) == 0) && ((p.OH & (p.OH - 1)) == 0);
            // dynamic LDS = RING (3 or 4, as in wgrad_tr_kernel) stages of 32 pixels x (BCO + BKK) bf16
            auto lds_bytes = [](int bco, int bkk, int mode) { return (size_t)((bco == 128 || (bkk == 256 && mode == 2)) ? 3 : 4) * 32 * (bco + bkk) * 2; };
#define EVE_WGRAD_LAUNCH2(WCO_, WK_, P2_, B_, MT_, TK, TC)                                                              \
    do {                                                                                                                \
        static bool attr_done = false;                                                                                  \
        if (!attr_done) {                                                                                               \
            (void)hipFuncSetAttribute((const void*)wgrad_tr_kernel<T, WCO_, WK_, P2_, B_, MT_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                                      \
            attr_done = true;                                                                                           \
        }                                                                                                               \
        EVE_LAUNCH(EVE_HNAME(T, "wgrad_tr_kernel<", ", " #WCO_ ", " #WK_ ", " #P2_ ", mt" #MT_ ">"), (wgrad_tr_kernel<T, WCO_, WK_, P2_, B_, MT_>), \
                   dim3((TK) * (TC) * splits), dim3(64 * WCO_ * WK_), lds_bytes(64 * WCO_, 64 * WK_, P2_), s, p, (const T*)x, \
                   (const T*)dy, dw, rows, (uint32_t)x_bytes, (uint32_t)dy_bytes, db);                               \
    } while (0)
#define EVE_WGRAD_LAUNCH1(WCO_, WK_, P2_, MT_, TK, TC)                                                                  \
    do {                                                                                                                \
        if (db) EVE_WGRAD_LAUNCH2(WCO_, WK_, P2_, true, MT_, TK, TC);                                                   \
        else    EVE_WGRAD_LAUNCH2(WCO_, WK_, P2_, false, MT_, TK, TC);                                                  \
    } while (0)
#define EVE_WGRAD_LAUNCH(WCO_, WK_, P2_, TK, TC) EVE_WGRAD_LAUNCH1(WCO_, WK_, P2_, 4, TK, TC)
            // address-decode mode of the gather (see wgrad_tr_kernel): both sizes powers of two / width only / neither
            const bool pow2w = (p.OW & (p.OW - 1)) == 0;
            const int mode = pow2 ? 1 : ((pow2w && p.OH * p.OW >= 32) ? 2 : 0);
            const int wg8w = g_cfg.wgrad_wg8;
            if (wg8w && !db && mode == 1 && p.Cout % 256 == 0 && p.K % 256 == 0) {
                // 256 x 256 tiles, eight waves of 128 x 64, role-split wave pairs (wgrad_wg8.h): one workgroup per CU
                const uint32_t tk = p.K / 256, tc = p.Cout / 256;
                wgrad_split(p, tk, tc, (size_t)128 * 1024, splits, rows);
                static bool attr_done = false;
                if (!attr_done) {
                    (void)hipFuncSetAttribute((const void*)wgrad_wg8_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    (void)hipFuncSetAttribute((const void*)wgrad_wg8_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    attr_done = true;
                }
                const unsigned long long n = (unsigned long long)p.Cout * p.K;
                // (the caller's scratch for THIS call: partial filters, consumed by the reduce launch that follows on the same stream)
                const bool slab = splits > 1 && workspace && (unsigned long long)splits * n * 4 <= workspace_bytes && ((uintptr_t)dw & 15) == 0 &&
                                  ((uintptr_t)workspace & 15) == 0;
                if (slab) {
                    EVE_LAUNCH(EVE_HNAME(T, "wgrad_wg8_kernel<", ", true>"), (wgrad_wg8_kernel<T, true>), dim3(tk * tc * splits), dim3(512), (size_t)128 * 1024, s, p,
                               (const T*)x, (const T*)dy, (float*)workspace, rows, (uint32_t)x_bytes, (uint32_t)dy_bytes);
                    const long long n4 = (long long)(n / 4);
                    long long blocks = (n4 + 255) / 256;
                    if (blocks > 2048) blocks = 2048;
                    hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)workspace, dw, n4, (int)splits);
                } else {
                    EVE_LAUNCH(EVE_HNAME(T, "wgrad_wg8_kernel<", ", false>"), (wgrad_wg8_kernel<T, false>), dim3(tk * tc * splits), dim3(512), (size_t)128 * 1024, s, p,
                               (const T*)x, (const T*)dy, dw, rows, (uint32_t)x_bytes, (uint32_t)dy_bytes);
                }
            } else if (p.Cout > 64) {
                const uint32_t tk = (p.K + 127) / 128, tc = (p.Cout + 127) / 128;
                wgrad_split(p, tk, tc, lds_bytes(128, 128, mode), splits, rows);
                if (mode == 1)      EVE_WGRAD_LAUNCH(2, 2, 1, tk, tc);
                else if (mode == 2) EVE_WGRAD_LAUNCH(2, 2, 2, tk, tc);
                else                EVE_WGRAD_LAUNCH(2, 2, 0, tk, tc);
            // (one 9-wave workgroup covering the whole 576-wide filter of the 64-channel layers -- operands fetched once
            //  instead of once per K tile -- measured SLOWER: 0.292 vs 0.237 ms; the surplus fetches hit the Infinity Cache)
            } else if (mode == 1 && p.K % 192 == 0 && p.K <= 1152) {
                // 64-channel 3x3 layers: K = 576 = 3 x 192 exactly (three waves per workgroup) instead of 3 x 256 padded
                // (0.235 vs 0.246 ms on layer 1; a 3-stage ring for it measured 0.242)
                const uint32_t tk = p.K / 192;
                wgrad_split(p, tk, 1, lds_bytes(64, 192, mode), splits, rows);
                EVE_WGRAD_LAUNCH(1, 3, 1, tk, 1);
            } else {
                const uint32_t tk = (p.K + 255) / 256, tc = 1;
                wgrad_split(p, tk, tc, lds_bytes(64, 256, mode), splits, rows);
                if (mode == 1)      EVE_WGRAD_LAUNCH(1, 4, 1, tk, tc);
                else if (mode == 2) {
                    // RefineNet's planes (72x128 .. 5x8); its outer levels have 16 / 32 output channels
                    if (p.Cout <= 16)      EVE_WGRAD_LAUNCH1(1, 4, 2, 1, tk, tc);
                    else if (p.Cout <= 32) EVE_WGRAD_LAUNCH1(1, 4, 2, 2, tk, tc);
                    else                   EVE_WGRAD_LAUNCH(1, 4, 2, tk, tc);
                }
                else                EVE_WGRAD_LAUNCH(1, 4, 0, tk, tc);
            }
#undef EVE_WGRAD_LAUNCH
#undef EVE_WGRAD_LAUNCH1
#undef EVE_WGRAD_LAUNCH2
            return db ? 1 : 0;                              // 1: the bias gradient has been taken care of
        }
    }
    }
    const bool wide = p.Cout > 64;
    const uint32_t bco = wide ? 128 : 64;
    const uint32_t tk = (p.K + 127) / 128, tc = (p.Cout + bco - 1) / bco;
    // enough splits of the pixel range for ~4 workgroups per CU, each a multiple of 64 pixels
    uint32_t want = (1024 + tk * tc - 1) / (tk * tc);
    uint32_t max_splits = (p.M + 255) / 256;
    uint32_t splits = want < 1 ? 1 : (want > max_splits ? max_splits : want);
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    uint32_t rows = (p.M + splits - 1) / splits;
    rows = (rows + 63) / 64 * 64;
    splits = (p.M + rows - 1) / rows;
    dim3 grid(tk, tc, splits), block(256);
    const T* a = (const T*)x; const T* b = (const T*)dy;
    if (wide) {
        if (ss) EVE_LAUNCH("wgrad_kernel<T, 4, true>", (wgrad_kernel<T, 4, true>), grid, block, 0, s, p, a, b, ss, pro_act, dw, rows);
        else    EVE_LAUNCH("wgrad_kernel<T, 4, false>", (wgrad_kernel<T, 4, false>), grid, block, 0, s, p, a, b, ss, pro_act, dw, rows);
    } else {
        if (ss) EVE_LAUNCH("wgrad_kernel<T, 2, true>", (wgrad_kernel<T, 2, true>), grid, block, 0, s, p, a, b, ss, pro_act, dw, rows);
        else    EVE_LAUNCH("wgrad_kernel<T, 2, false>", (wgrad_kernel<T, 2, false>), grid, block, 0, s, p, a, b, ss, pro_act, dw, rows);
    }
    return 0;
}

}  // namespace eve

using namespace eve;

namespace eve {

// 3x3 / stride 2 / pad 1 forward of the trunk's down-sampling blocks on conv3x3s2_wg8_kernel (conv_wg8s2.h).  true: launched.
template <typename HT>
static bool launch_s2_fwd_wg8(const eve_conv_desc* d, const void* x, const void* w, const float* bias, int epi_act, void* y, hipStream_t s) {
    const int on = g_cfg.conv_wg8, min_tiles = g_cfg.conv_wg8_s2_min_tiles;
    const int W = d->OW;
    if (!on || d->KH != 3 || d->KW != 3 || d->stride != 2 || d->pad != 1 || d->OH != W || d->IH != 2 * W || d->IW != 2 * W ||
        !(W == 16 || W == 8 || W == 4) || d->Cin % 64 || (epi_act & ~0xff) ||
        !((epi_act & 0xff) == EVE_ACT_NONE || (epi_act & 0xff) == EVE_ACT_RELU))
        return false;
    const int cout_t = W == 16 ? 128 : 256;
    if (d->Cout % cout_t) return false;
    const unsigned long long xb = (unsigned long long)d->N * 4 * W * W * d->Cin * 2, wb = (unsigned long long)d->Cout * 9 * d->Cin * 2;
    if (xb >= (1ull << 31) || wb >= (1ull << 31)) return false;
    const int ti = W == 16 ? 2 : (W == 8 ? 4 : 16);
    const uint32_t tiles = (uint32_t)((d->N + ti - 1) / ti) * (uint32_t)(d->Cout / cout_t);
    if ((int)tiles < min_tiles) return false;
    Wg8Params g;
    g.N = d->N; g.Cin = d->Cin; g.Cout = d->Cout; g.flip = 0; g.K = 9 * d->Cin; g.x_bytes = (uint32_t)xb; g.w_bytes = (uint32_t)wb;
    g.tiles_n = (uint32_t)(d->Cout / cout_t); g.s2_py = 0;
#define EVE_S2F_LAUNCH(WM_, WN_, W_)                                                                                       \
    do {                                                                                                                  \
        using G8 = Wg8S2Geom<WM_, WN_, W_>;                                                                                 \
        static bool attr_done = false;                                                                                    \
        if (!attr_done) {                                                                                                 \
            (void)hipFuncSetAttribute((const void*)conv3x3s2_wg8_kernel<HT, WM_, WN_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_done = true;                                                                                             \
        }                                                                                                                 \
        EVE_LAUNCH(EVE_HNAME(HT, "conv3x3s2_wg8_kernel<", ", " #WM_ ", " #WN_ ", " #W_ ">"), (conv3x3s2_wg8_kernel<HT, WM_, WN_, W_>), \
                   dim3(tiles), dim3(512), G8::LDS, s, g, (const HT*)x, (const HT*)w, bias, epi_act, (HT*)y);                   \
    } while (0)
    if (W == 16) EVE_S2F_LAUNCH(4, 2, 16);
    else if (W == 8) EVE_S2F_LAUNCH(2, 4, 8);
    else EVE_S2F_LAUNCH(2, 4, 4);
#undef EVE_S2F_LAUNCH
    return true;
}

}  // namespace eve

extern "C" int eve_conv2d_fwd(const eve_conv_desc* d, const void* x, const void* w_ohwi, const float* bias,
                              int epi_act, const float* in_scale_shift, int pro_act, void* y,
                              eve_stream_t stream) {
    const int vec = (d && d->dtype != EVE_DT_F32) ? 8 : 4;
    if (int e = check_desc(d, vec)) return e;
    if (!x || !w_ohwi || !y) return set_error_msg("conv2d_fwd: null pointer");
    GatherParams p = fwd_params(d);
    hipStream_t s = (hipStream_t)stream;
    if (!in_scale_shift && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->dtype != EVE_DT_F32) {    // streaming 1x1
        const long long M = (long long)d->N * d->OH * d->OW;
        bool done = false;
        EVE_DISPATCH_H16(d->dtype, done = launch_conv1x1_stream<H>(M, d->Cin, d->Cout, x, w_ohwi, bias, epi_act, y, s));
        if (done) { EVE_CHECK_LAUNCH(); return 0; }
    }
    if (!in_scale_shift && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->dtype != EVE_DT_F32) {    // row-streaming 3x3
        bool done = false;
        EVE_DISPATCH_H16(d->dtype, done = launch_conv3x3_stream<H>(d->N, d->IH, d->IW, d->Cin, d->Cout, 0, x, w_ohwi, bias, epi_act, y, s));
        if (done) { EVE_CHECK_LAUNCH(); return 0; }
    }
    if (!in_scale_shift) {          // the trunk's stride-2 3x3 layers: parity planes of the input as rotating halo stages
        if (d->dtype == EVE_DT_BF16 && launch_s2_fwd_wg8<bf16_t>(d, x, w_ohwi, bias, epi_act, y, s)) { EVE_CHECK_LAUNCH(); return 0; }
        if (d->dtype == EVE_DT_F16 && launch_s2_fwd_wg8<f16_t>(d, x, w_ohwi, bias, epi_act, y, s)) { EVE_CHECK_LAUNCH(); return 0; }
    }
    if (d->dtype == EVE_DT_BF16) launch_igemm<bf16_t>(p, x, w_ohwi, bias, in_scale_shift, pro_act, epi_act, y, s);
    else if (d->dtype == EVE_DT_F16) launch_igemm<f16_t>(p, x, w_ohwi, bias, in_scale_shift, pro_act, epi_act, y, s);
    else                         launch_igemm<float>(p, x, w_ohwi, bias, in_scale_shift, pro_act, epi_act, y, s);
    EVE_CHECK_LAUNCH();
    return 0;
}

/* eve_conv2d_fwd that ALSO emits the InstanceNorm2d statistics of its output (mean, rstd per (n, c); biased variance) when the
   kernel it dispatches walks whole images -- the row-streaming 3x3 kernel (conv_3x3s.h): RefineNet's pre-activation blocks
   normalise every convolution's output (refine_net.py:45-53), and on the big planes that statistics pass was a launch of its
   own.  *stats_written = 1 when mean_rstd [N][Cout][2] was filled, 0 when the shape went to another kernel (the caller then
   runs eve_instnorm_stats as before). */
extern "C" int eve_conv2d_fwd_stats(const eve_conv_desc* d, const void* x, const void* w_ohwi, const float* bias, int epi_act,
                                    void* y, float* mean_rstd, float eps, int* stats_written, eve_stream_t stream) {
    if (!stats_written || !mean_rstd) return set_error_msg("conv2d_fwd_stats: null pointer");
    *stats_written = 0;
    const int vec = (d && d->dtype != EVE_DT_F32) ? 8 : 4;
    if (int e = check_desc(d, vec)) return e;
    if (!x || !w_ohwi || !y) return set_error_msg("conv2d_fwd_stats: null pointer");
    if (d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->dtype != EVE_DT_F32 && !(epi_act & EVE_EPI_ACC)) {
        bool done = false;
        EVE_DISPATCH_H16(d->dtype, done = launch_conv3x3_stream<H>(d->N, d->IH, d->IW, d->Cin, d->Cout, 0, x, w_ohwi, bias, epi_act, y,
                                                                   (hipStream_t)stream, mean_rstd, eps));
        if (done) { *stats_written = 1; EVE_CHECK_LAUNCH(); return 0; }
    }
    return eve_conv2d_fwd(d, x, w_ohwi, bias, epi_act, nullptr, 0, y, stream);
}

namespace eve {

// Filters of the stride-2 data gradient as conv3x3_wg8_kernel<.., NT> wants them (see there): for output row parity py,
// W'[px * Cdx + ci][t][co] = w_ihwo[ci][kh][kw][co] with kh = py + 1 - 2 dy, kw = px + 1 - 2 dx for the window position
// t = (dy, dx) (py = 0: dy = 0 only, NT = 2; py = 1: NT = 4), zero where the tap does not exist (px = 0, dx = 1).
template <typename H>
__global__ __launch_bounds__(256) void s2_dgrad_pack_kernel(const H* __restrict__ w_ihwo, H* __restrict__ w0, H* __restrict__ w1,
                                                            int Cdx, int Cout) {
    const int kv = Cout / 8;                                  // 16-byte vectors per (row, tap)
    const long long n0 = (long long)2 * Cdx * 2 * kv, n1 = (long long)2 * Cdx * 4 * kv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n0 + n1; i += (long long)gridDim.x * 256) {
        const int py = i >= n0 ? 1 : 0, nt = py ? 4 : 2;
        const long long r = py ? i - n0 : i;
        const int v = (int)(r % kv), t = (int)((r / kv) % nt), row = (int)(r / ((long long)kv * nt));
        const int px = row >= Cdx ? 1 : 0, ci = row - px * Cdx;
        const int dy = py ? t >> 1 : 0, dx = t & 1;
        const int kh = py + 1 - 2 * dy, kw = px + 1 - 2 * dx;          // py = 0: 1;  py = 1: 2, 0;  same for kw
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (kh >= 0 && kw >= 0)
            q = reinterpret_cast<const uint4*>(w_ihwo + (((size_t)ci * 3 + kh) * 3 + kw) * Cout)[v];
        reinterpret_cast<uint4*>(py ? w1 : w0)[r] = q;
    }
}

// true: launched.  dy [N][OW][OW][Cout] -> dx [N][2 OW][2 OW][Cdx]
template <typename HT>
static bool launch_s2_dgrad_wg8(const eve_conv_desc* d, const void* dy, const void* w_ihwo, void* dx, void* workspace,
                                unsigned long long workspace_bytes, hipStream_t s) {
    // (its own tile floor: against four per-tap launches the eight-wave pair is ahead from ~48 workgroups on -- B = 8 clips per
    //  GPU: layer 3.0 0.070 -> 0.049 ms with 120 tiles, layer 4.0 0.090 -> 0.070 with 60 -- where the 3x3 convolution wants 224)
    const int on = g_cfg.conv_wg8, min_tiles = g_cfg.conv_wg8_s2_min_tiles;
    const int W = d->OW, Cdx = d->Cin, Co = d->Cout;
    if (!on || d->KH != 3 || d->KW != 3 || d->stride != 2 || d->pad != 1 || d->OH != W || d->IH != 2 * W || d->IW != 2 * W ||
        !(W == 16 || W == 8 || W == 4) || Co % 64 || !workspace || ((uintptr_t)workspace & 255))
        return false;
    const int cout_t = W == 16 ? 128 : 256;
    if ((2 * Cdx) % cout_t) return false;
    const unsigned long long wbytes = (unsigned long long)2 * Cdx * 6 * Co * 2;
    const unsigned long long xb = (unsigned long long)d->N * W * W * Co * 2, ob = (unsigned long long)d->N * 4 * W * W * Cdx * 2;
    if (wbytes > workspace_bytes || xb >= (1ull << 31) || ob >= (1ull << 31) || wbytes >= (1ull << 31)) return false;
    const int ti = W == 16 ? 2 : (W == 8 ? 4 : 16);
    const uint32_t tiles = (uint32_t)((d->N + ti - 1) / ti) * (uint32_t)(2 * Cdx / cout_t);
    if ((int)tiles < min_tiles) return false;
    HT* w0 = (HT*)workspace;
    HT* w1 = w0 + (size_t)2 * Cdx * 2 * Co;
    const long long nvec = (long long)2 * Cdx * 6 * (Co / 8);
    hipLaunchKernelGGL(s2_dgrad_pack_kernel<HT>, dim3((unsigned)((nvec + 255) / 256 > 1024 ? 1024 : (nvec + 255) / 256)), dim3(256), 0, s,
                       (const HT*)w_ihwo, w0, w1, Cdx, Co);
    Wg8Params g;
    g.N = d->N; g.Cin = Co; g.Cout = 2 * Cdx; g.flip = 0; g.x_bytes = (uint32_t)xb; g.tiles_n = (uint32_t)(2 * Cdx / cout_t);
#define EVE_S2_LAUNCH(WM_, WN_, W_, NT_, WPTR, PY)                                                                          \
    do {                                                                                                                  \
        using G8 = Wg8Geom<WM_, WN_, W_>;                                                                                   \
        static bool attr_done = false;                                                                                    \
        if (!attr_done) {                                                                                                 \
            (void)hipFuncSetAttribute((const void*)conv3x3_wg8_kernel<HT, WM_, WN_, W_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_done = true;                                                                                             \
        }                                                                                                                 \
        g.K = NT_ * Co; g.w_bytes = (uint32_t)((unsigned long long)2 * Cdx * NT_ * Co * 2); g.s2_py = PY;                   \
        EVE_LAUNCH(EVE_HNAME(HT, "conv3x3_wg8_kernel<", ", " #WM_ ", " #WN_ ", " #W_ ", s2dgrad" #NT_ ">"), (conv3x3_wg8_kernel<HT, WM_, WN_, W_, NT_>), \
                   dim3(tiles), dim3(512), G8::LDS, s, g, (const HT*)dy, (const HT*)(WPTR), (const float*)nullptr, (int)EVE_ACT_NONE, (HT*)dx); \
    } while (0)
    if (W == 16) { EVE_S2_LAUNCH(4, 2, 16, 2, w0, 0); EVE_S2_LAUNCH(4, 2, 16, 4, w1, 1); }
    else if (W == 8) { EVE_S2_LAUNCH(2, 4, 8, 2, w0, 0); EVE_S2_LAUNCH(2, 4, 8, 4, w1, 1); }
    else { EVE_S2_LAUNCH(2, 4, 4, 2, w0, 0); EVE_S2_LAUNCH(2, 4, 4, 4, w1, 1); }
#undef EVE_S2_LAUNCH
    return true;
}

}  // namespace eve

extern "C" int eve_conv2d_dgrad(const eve_conv_desc* d, const void* dy, const void* w_ihwo, void* dx,
                                void* workspace, unsigned long long workspace_bytes, eve_stream_t stream) {
    const int vec = (d && d->dtype != EVE_DT_F32) ? 8 : 4;
    if (int e = check_desc(d, vec)) return e;
    if (d->Cout % vec) return set_error_msg("conv2d_dgrad: Cout must be a multiple of the 16-byte vector");
    if (!dy || !w_ihwo || !dx) return set_error_msg("conv2d_dgrad: null pointer");
    GatherParams p = dgrad_params(d);
    hipStream_t s = (hipStream_t)stream;
    if (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->dtype != EVE_DT_F32) {    // 1x1: a 1x1 convolution by [Cin][Cout]
        const long long M = (long long)d->N * d->OH * d->OW;
        bool done = false;
        EVE_DISPATCH_H16(d->dtype, done = launch_conv1x1_stream<H>(M, d->Cout, d->Cin, dy, w_ihwo, nullptr, 0, dx, s));
        if (done) { EVE_CHECK_LAUNCH(); return 0; }
    }
    if (d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->dtype != EVE_DT_F32) {    // row-streaming 3x3, mirrored taps
        bool done = false;
        EVE_DISPATCH_H16(d->dtype, done = launch_conv3x3_stream<H>(d->N, d->IH, d->IW, d->Cout, d->Cin, 1, dy, w_ihwo, nullptr, 0, dx, s));
        if (done) { EVE_CHECK_LAUNCH(); return 0; }
    }
    // stride-2 3x3 layers: two eight-wave launches over the dy halo tile instead of four per-tap parity-class launches
    if (d->dtype == EVE_DT_BF16 && launch_s2_dgrad_wg8<bf16_t>(d, dy, w_ihwo, dx, workspace, workspace_bytes, s)) { EVE_CHECK_LAUNCH(); return 0; }
    if (d->dtype == EVE_DT_F16 && launch_s2_dgrad_wg8<f16_t>(d, dy, w_ihwo, dx, workspace, workspace_bytes, s)) { EVE_CHECK_LAUNCH(); return 0; }
    if (d->dtype == EVE_DT_BF16) launch_igemm<bf16_t>(p, dy, w_ihwo, nullptr, nullptr, 0, 0, dx, s);
    else if (d->dtype == EVE_DT_F16) launch_igemm<f16_t>(p, dy, w_ihwo, nullptr, nullptr, 0, 0, dx, s);
    else                         launch_igemm<float>(p, dy, w_ihwo, nullptr, nullptr, 0, 0, dx, s);
    EVE_CHECK_LAUNCH();
    return 0;
}

/* dx += data gradient: the residual join of a ResNet block (autograd's add of the two branch gradients) fused
   into the convolution's epilogue.  dx must already hold the other branch's gradient. */
extern "C" int eve_conv2d_dgrad_acc(const eve_conv_desc* d, const void* dy, const void* w_ihwo, void* dx,
                                    eve_stream_t stream) {
    const int vec = (d && d->dtype != EVE_DT_F32) ? 8 : 4;
    if (int e = check_desc(d, vec)) return e;
    if (d->Cout % vec) return set_error_msg("conv2d_dgrad_acc: Cout must be a multiple of the 16-byte vector");
    if (!dy || !w_ihwo || !dx) return set_error_msg("conv2d_dgrad_acc: null pointer");
    GatherParams p = dgrad_params(d);
    hipStream_t s = (hipStream_t)stream;
    if (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->dtype != EVE_DT_F32) {
        const long long M = (long long)d->N * d->OH * d->OW;
        bool done = false;
        EVE_DISPATCH_H16(d->dtype, done = launch_conv1x1_stream<H>(M, d->Cout, d->Cin, dy, w_ihwo, nullptr, EVE_EPI_ACC, dx, s));
        if (done) { EVE_CHECK_LAUNCH(); return 0; }
    }
    if (d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->dtype != EVE_DT_F32) {
        bool done = false;
        EVE_DISPATCH_H16(d->dtype, done = launch_conv3x3_stream<H>(d->N, d->IH, d->IW, d->Cout, d->Cin, 1, dy, w_ihwo, nullptr, EVE_EPI_ACC, dx, s));
        if (done) { EVE_CHECK_LAUNCH(); return 0; }
    }
    if (d->dtype == EVE_DT_BF16) launch_igemm<bf16_t>(p, dy, w_ihwo, nullptr, nullptr, 0, EVE_EPI_ACC, dx, s);
    else if (d->dtype == EVE_DT_F16) launch_igemm<f16_t>(p, dy, w_ihwo, nullptr, nullptr, 0, EVE_EPI_ACC, dx, s);
    else                         launch_igemm<float>(p, dy, w_ihwo, nullptr, nullptr, 0, EVE_EPI_ACC, dx, s);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_conv2d_wgrad(const eve_conv_desc* d, const void* x, const void* dy,
                                const float* in_scale_shift, int pro_act, float* dw_ohwi,
                                void* workspace, unsigned long long workspace_bytes, eve_stream_t stream) {
    const int vec = (d && d->dtype != EVE_DT_F32) ? 8 : 4;
    if (int e = check_desc(d, vec)) return e;
    if (d->Cout % vec) return set_error_msg("conv2d_wgrad: Cout must be a multiple of the 16-byte vector");
    if (!x || !dy || !dw_ohwi) return set_error_msg("conv2d_wgrad: null pointer");
    GatherParams p = fwd_params(d);
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == EVE_DT_BF16) launch_wgrad<bf16_t>(p, x, dy, in_scale_shift, pro_act, dw_ohwi, s, nullptr, workspace, workspace_bytes);
    else if (d->dtype == EVE_DT_F16) launch_wgrad<f16_t>(p, x, dy, in_scale_shift, pro_act, dw_ohwi, s, nullptr, workspace, workspace_bytes);
    else                         launch_wgrad<float>(p, x, dy, in_scale_shift, pro_act, dw_ohwi, s);
    EVE_CHECK_LAUNCH();
    return 0;
}

/* Weight gradient of the ResNet stem straight from the packed patches the stem kernels read:
   x_padded [N][IH+6][IW+8][4] bf16 is viewed as a 4-channel NHWC image and the filter as 7 x 8 taps (the 8th column
   does not exist; its gradient is written and ignored), so a 16-byte operand slot is two horizontally adjacent taps and
   K = 7*8*4 = 224 fits ONE K tile -- against K = 392 (two padded tiles, 62 % of the MACs on zero channels) for the
   8-channel NHWC copy, which is no longer needed at all.  dw [64][7][8][4] float, accumulated. */
extern "C" int eve_stem_wgrad(int dtype, int N, int IH, int IW, const void* x_padded, const void* dconv, float* dw, eve_stream_t stream) {
    if ((dtype != EVE_DT_BF16 && dtype != EVE_DT_F16) || N <= 0 || IH <= 0 || IW <= 0 || (IH & 1) || (IW & 1) || !x_padded || !dconv || !dw)
        return set_error_msg("stem_wgrad: bad arguments");
    GatherParams p;
    p.N = N; p.IH = IH + 6; p.IW = IW + 8; p.Cin = 4;
    p.OH = IH / 2; p.OW = IW / 2; p.Cout = 64; p.KH = 7; p.KW = 8;
    p.o_mul = 2; p.k_mul = 1; p.off = 0; p.div = 1;          // the padding is materialised in x_padded
    p.K = 7 * 8 * 4;
    p.M = (uint32_t)((long long)N * p.OH * p.OW);
    p.fd_ohw = make_fastdiv(p.OH * p.OW); p.fd_ow = make_fastdiv(p.OW);
    p.fd_cin = make_fastdiv(4); p.fd_kw = make_fastdiv(8); p.fd_oh = make_fastdiv(p.OH);
    // conv pad is 3, the packed rows carry 4 pixels of left padding: start one pixel (8 bytes) in
    const char* x1 = (const char*)x_padded + 8;
    if ((unsigned long long)N * p.IH * p.IW * 8 >= (1ull << 31)) return set_error_msg("stem_wgrad: packed input must stay below 2 GiB");
    // the transposing-read kernel addresses its operands with 32-bit buffer offsets: images are taken in chunks whose d(conv out)
    // stays below 2 GiB (the weight gradient accumulates, so the chunks simply add up).  256 x 256 patches x 1 920 frames
    // (BASELINE configs[4]) are 4 GiB of d(conv out): as ONE launch they fell to the first-generation kernel, 4.4 ms.
    const unsigned long long per_img = (unsigned long long)p.OH * p.OW * 64 * 2;
    int chunk = (int)(((1ull << 31) - 1) / per_img);
    if (chunk < 1) return set_error_msg("stem_wgrad: one image's gradient exceeds 2 GiB");
    const int nchunks = (N + chunk - 1) / chunk;
    chunk = (N + nchunks - 1) / nchunks;
    for (int n0 = 0; n0 < N; n0 += chunk) {
        const int n = N - n0 < chunk ? N - n0 : chunk;
        p.N = n;
        p.M = (uint32_t)((long long)n * p.OH * p.OW);
        const char* xc = x1 + (size_t)n0 * p.IH * p.IW * 8;
        const char* dc = (const char*)dconv + (size_t)n0 * per_img;
        EVE_DISPATCH_H16(dtype, launch_wgrad<H>(p, xc, dc, nullptr, 0, dw, (hipStream_t)stream));
    }
    EVE_CHECK_LAUNCH();
    return 0;
}

static void launch_bias_grad(int dtype, long long M, int C, const void* dy, float* db, hipStream_t s) {
    long long blocks = (M + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const long long rows = (M + blocks - 1) / blocks;
    blocks = (M + rows - 1) / rows;
    if (dtype == EVE_DT_BF16)
        hipLaunchKernelGGL(bias_grad_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)dy, db, M, C, rows);
    else if (dtype == EVE_DT_F16)
        hipLaunchKernelGGL(bias_grad_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const f16_t*)dy, db, M, C, rows);
    else
        hipLaunchKernelGGL(bias_grad_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)dy, db, M, C, rows);
}

extern "C" int eve_bias_grad(int dtype, long long M, int C, const void* dy, float* db, eve_stream_t stream) {
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    if (M <= 0 || C <= 0 || C % vec || C / vec > 256) return set_error_msg("bias_grad: bad shape");
    if (!dy || !db) return set_error_msg("bias_grad: null pointer");
    launch_bias_grad(dtype, M, C, dy, db, (hipStream_t)stream);
    EVE_CHECK_LAUNCH();
    return 0;
}

/* Weight and bias gradient of one convolution in a single pass over dy where the kernel allows it (bf16: the column
   sums ride on the weight-gradient MFMAs); otherwise the weight gradient followed by the column-sum kernel.
   dw and db are accumulated into. */
extern "C" int eve_conv2d_wgrad_bias(const eve_conv_desc* d, const void* x, const void* dy, float* dw_ohwi, float* db,
                                     void* workspace, unsigned long long workspace_bytes, eve_stream_t stream) {
    const int vec = (d && d->dtype != EVE_DT_F32) ? 8 : 4;
    if (int e = check_desc(d, vec)) return e;
    if (d->Cout % vec || d->Cout / vec > 256) return set_error_msg("conv2d_wgrad_bias: bad Cout");
    if (!x || !dy || !dw_ohwi || !db) return set_error_msg("conv2d_wgrad_bias: null pointer");
    GatherParams p = fwd_params(d);
    hipStream_t s = (hipStream_t)stream;
    int fused = 0;
    if (d->dtype == EVE_DT_BF16) fused = launch_wgrad<bf16_t>(p, x, dy, nullptr, 0, dw_ohwi, s, db, workspace, workspace_bytes);
    else if (d->dtype == EVE_DT_F16) fused = launch_wgrad<f16_t>(p, x, dy, nullptr, 0, dw_ohwi, s, db, workspace, workspace_bytes);
    else                         launch_wgrad<float>(p, x, dy, nullptr, 0, dw_ohwi, s);
    if (!fused) launch_bias_grad(d->dtype, (long long)p.M, d->Cout, dy, db, s);
    EVE_CHECK_LAUNCH();
    return 0;
}
