// Small float32 linear layers: the EyeNet tail (fc, fc_common, GRU input projection, gaze / pupil heads --
// /root/reference/src/models/eye_net.py:52-90) works on M = 2*B*T feature rows with K, N <= 512.  Through the
// 128x128-tile implicit-GEMM kernels such a problem is 15 workgroups of f32 MFMAs (1/16 of the bf16 rate) on a
// 256-CU part: ~25 us per launch, 32 launches per step.  Here the tile is 8 rows x 128 columns of plain fp32
// FMAs (exact, same summation order as a dot product over k), 240+ workgroups, and the activation derivative of
// the backward is applied while loading dy instead of in a separate pass.
//
//   eve_linear_fwd     y[M][N]  = act(x[M][K] . wt[K][N] + b)                       wt = IHWO pack ([in][out])
//   eve_linear_dgrad   dx[M][K] = (dy . act'(y))[M][N] . w[N][K]                    w  = OHWI pack ([out][in])
//   eve_linear_wgrad   dw[N][K] += (dy . act'(y))^T . x ;  db[N] += column sums     (float atomics over row splits)
#include "common.h"

namespace eve {

constexpr int LS_TM = 8, LS_TN = 128, LS_KC = 32;     // 8 rows per workgroup: 240 workgroups at M = 1920 (one per CU)
                                                       // (64-deep K chunks measured no faster: 13.4 vs 12.5 us per launch)

// C[M][Nc] = epi( A'[M][R] . B[R][Nc] ),  A' = A * act'(Y) if Y (same shape as A), epi = act(. + bias).
// Both operand chunks go through LDS; the next chunk is fetched into registers while the current one is consumed
// (each workgroup is alone on its SIMDs, so nothing else hides the L2 latency).
// lda: row stride of A and Y (the leading R columns of a wider matrix), ldc: row stride of C (a column range of a wider one),
// n_bias: entries of `bias` that exist (columns beyond take 0), accumulate: C += result (round 4: the tail as one autograd node
// writes fc's output into the head-pose concatenation, reads fc's gradient out of the concatenation's, and sums the two heads'
// input gradients in the second head's epilogue -- no cat / pad / slice / add launches)
__global__ __launch_bounds__(256) void linear_mm_kernel(const float* __restrict__ A, const float* __restrict__ Y, const int pro_act,
                                                        const float* __restrict__ B, const float* __restrict__ bias,
                                                        const int epi_act, float* __restrict__ C, const int M, const int R,
                                                        const int Nc, const int lda, const int ldc, const int n_bias,
                                                        const int accumulate) {
    __shared__ float sA[LS_TM][LS_KC + 4];
    __shared__ float sB[LS_KC][LS_TN];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * LS_TN;
    const int col = c0 + (tid & (LS_TN - 1)), rg = tid >> 7;                  // 2 row groups of 4 rows
    const int m0 = blockIdx.x * LS_TM;
    // staging slots: A' 8 x KC values (KC / 32 per thread), B KC x 128 values (KC / 2 per thread)
    constexpr int NA = LS_TM * LS_KC / 256, NB = LS_KC / 2;
    const int bc = tid & 127, bk = tid >> 7;                                    // k rows bk, bk + 2, ...
    float pa[NA], pb[NB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + 256 * i;
            const int m = m0 + e / LS_KC, k = k0 + e % LS_KC;
            float v = 0.f;
            if (m < M && k < R) {
                v = A[(size_t)m * lda + k];
                if (Y) v *= act_grad_from_out(Y[(size_t)m * lda + k], pro_act);
            }
            pa[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int k = k0 + bk + 2 * i;
            pb[i] = (k < R && c0 + bc < Nc) ? B[(size_t)k * Nc + c0 + bc] : 0.f;
        }
    };
    float acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = 0.f;
    fetch(0);
    for (int k0 = 0; k0 < R; k0 += LS_KC) {
        __syncthreads();                                   // previous chunk fully consumed
#pragma unroll
        for (int i = 0; i < NA; ++i) { const int e = tid + 256 * i; sA[e / LS_KC][e % LS_KC] = pa[i]; }
#pragma unroll
        for (int i = 0; i < NB; ++i) sB[bk + 2 * i][bc] = pb[i];
        __syncthreads();
        if (k0 + LS_KC < R) fetch(k0 + LS_KC);             // in flight during the FMAs below
#pragma unroll
        for (int kk = 0; kk < LS_KC; kk += 4) {
            const float b0 = sB[kk][tid & 127], b1 = sB[kk + 1][tid & 127], b2 = sB[kk + 2][tid & 127], b3 = sB[kk + 3][tid & 127];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 a = *reinterpret_cast<const float4*>(&sA[rg * 4 + i][kk]);
                acc[i] = fmaf(a.x, b0, acc[i]);
                acc[i] = fmaf(a.y, b1, acc[i]);
                acc[i] = fmaf(a.z, b2, acc[i]);
                acc[i] = fmaf(a.w, b3, acc[i]);
            }
        }
    }
    if (col >= Nc) return;
    const float bv = (bias && col < n_bias) ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + rg * 4 + i;
        if (m < M) {
            float v = act_fwd(acc[i] + bv, epi_act);
            float* c = C + (size_t)m * ldc + col;
            if (accumulate) v += *c;
            *c = v;
        }
    }
}

// dW[N][K] += G^T . X,  db[N] += colsum(G),  G = dY * act'(Y);  block = 32 n x 64 k x one row split
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ dY, const float* __restrict__ Y, const int act,
                                                           const float* __restrict__ X, float* __restrict__ dW,
                                                           float* __restrict__ db, const int M, const int N, const int K,
                                                           const int rows_per_split) {
    __shared__ float sG[32][32 + 4];                     // [row in chunk][n]
    const int tid = threadIdx.x;
    const int k = blockIdx.x * 64 + (tid & 63), ng = tid >> 6;      // 4 groups of 8 output rows
    const int n0 = blockIdx.y * 32;
    const int m_begin = blockIdx.z * rows_per_split, m_end = min(M, m_begin + rows_per_split);
    float acc[8], bsum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int mc = m_begin; mc < m_end; mc += 32) {
        float g4[4], xr[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int r = e >> 5, nn = e & 31;
            const int m = mc + r, n = n0 + nn;
            float v = 0.f;
            if (m < m_end && n < N) {
                v = dY[(size_t)m * N + n];
                if (Y) v *= act_grad_from_out(Y[(size_t)m * N + n], act);
            }
            g4[i] = v;
        }
        __syncthreads();                                   // previous chunk consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int e = tid + 256 * i; sG[e >> 5][e & 31] = g4[i]; }
        __syncthreads();
        const int rmax = min(32, m_end - mc);
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += 8) {               // 8 row loads in flight at a time
#pragma unroll
            for (int j = 0; j < 8; ++j) xr[j] = (k < K && r0 + j < rmax) ? X[(size_t)(mc + r0 + j) * K + k] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = xr[j];
                const float4 g0 = *reinterpret_cast<const float4*>(&sG[r0 + j][ng * 8]);
                const float4 g1 = *reinterpret_cast<const float4*>(&sG[r0 + j][ng * 8 + 4]);
                acc[0] = fmaf(g0.x, xv, acc[0]); acc[1] = fmaf(g0.y, xv, acc[1]);
                acc[2] = fmaf(g0.z, xv, acc[2]); acc[3] = fmaf(g0.w, xv, acc[3]);
                acc[4] = fmaf(g1.x, xv, acc[4]); acc[5] = fmaf(g1.y, xv, acc[5]);
                acc[6] = fmaf(g1.z, xv, acc[6]); acc[7] = fmaf(g1.w, xv, acc[7]);
            }
        }
        if (db && blockIdx.x == 0 && tid < 32)
            for (int r = 0; r < rmax; ++r) bsum += sG[r][tid];
    }
    if (k < K) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = n0 + ng * 8 + i;
            if (n < N) atomicAdd(dW + (size_t)n * K + k, acc[i]);
        }
    }
    if (db && blockIdx.x == 0 && tid < 32 && n0 + tid < N) atomicAdd(db + n0 + tid, bsum);
}

}  // namespace eve

using namespace eve;

static int ls_check(int M, int K, int N, const char* what) {
    if (M <= 0 || K <= 0 || N <= 0 || K > 4096 || N > 4096) return set_error_msg(what);
    return 0;
}

extern "C" int eve_linear_fwd(int M, int K, int N, const float* x, const float* w_in_out, const float* bias, int act,
                              float* y, eve_stream_t stream) {
    if (int e = ls_check(M, K, N, "linear_fwd: bad shape")) return e;
    if (!x || !w_in_out || !y) return set_error_msg("linear_fwd: null pointer");
    EVE_LAUNCH("linear_mm_kernel", linear_mm_kernel, dim3((M + LS_TM - 1) / LS_TM, (N + LS_TN - 1) / LS_TN), dim3(256), 0, (hipStream_t)stream,
                       x, (const float*)nullptr, 0, w_in_out, bias, act, y, M, K, N, K, N, N, 0);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_linear_dgrad(int M, int K, int N, const float* dy, const float* y, int act, const float* w_out_in,
                                float* dx, eve_stream_t stream) {
    if (int e = ls_check(M, K, N, "linear_dgrad: bad shape")) return e;
    if (!dy || !w_out_in || !dx || (act != EVE_ACT_NONE && !y)) return set_error_msg("linear_dgrad: null pointer");
    EVE_LAUNCH("linear_mm_kernel", linear_mm_kernel, dim3((M + LS_TM - 1) / LS_TM, (K + LS_TN - 1) / LS_TN), dim3(256), 0, (hipStream_t)stream,
                       dy, act != EVE_ACT_NONE ? y : (const float*)nullptr, act, w_out_in, (const float*)nullptr, 0, dx, M, N, K, N, K, 0, 0);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_linear_wgrad(int M, int K, int N, const float* dy, const float* y, int act, const float* x, float* dw_out_in,
                                float* db, eve_stream_t stream) {
    if (int e = ls_check(M, K, N, "linear_wgrad: bad shape")) return e;
    if (!dy || !x || !dw_out_in || (act != EVE_ACT_NONE && !y)) return set_error_msg("linear_wgrad: null pointer");
    const int tiles = ((K + 63) / 64) * ((N + 31) / 32);
    int splits = (512 + tiles - 1) / tiles;                  // ~2 workgroups per CU
    const int max_splits = (M + 63) / 64;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int rows = (M + splits - 1) / splits;
    rows = (rows + 31) / 32 * 32;
    splits = (M + rows - 1) / rows;
    EVE_LAUNCH("linear_wgrad_kernel", linear_wgrad_kernel, dim3((K + 63) / 64, (N + 31) / 32, splits), dim3(256), 0, (hipStream_t)stream, dy,
                       act != EVE_ACT_NONE ? y : (const float*)nullptr, act, x, dw_out_in, db, M, N, K, rows);
    EVE_CHECK_LAUNCH();
    return 0;
}

// =====================================================================================================================
// Round 4: the tail's weight / bias gradients batched into one launch (the chained forward / data-gradient kernel that
// came with it was measured slower than the per-layer launches -- profiles/r04_notes.md 3 -- and was removed in round 5).
// =====================================================================================================================
namespace eve {

// several weight / bias gradients in one launch: workgroup -> (problem, tile, row split) through a prefix table
__global__ __launch_bounds__(256) void linear_wgrad_batch_kernel(const eve_wgrad_batch b) {
    int pi = 0;
    while (pi + 1 < b.n && (int)blockIdx.x >= b.first_block[pi + 1]) ++pi;
    const eve_wgrad_problem& q = b.p[pi];
    const int local = blockIdx.x - b.first_block[pi];
    const int tk = (q.K + 63) / 64, tn = (q.N + 31) / 32;
    const int bx = local % tk, by = (local / tk) % tn, bz = local / (tk * tn);
    __shared__ float sG[32][32 + 4];
    const int tid = threadIdx.x;
    const int k = bx * 64 + (tid & 63), ng = tid >> 6;
    const int n0 = by * 32;
    const int m_begin = bz * q.rows_per_split, m_end = min(q.M, m_begin + q.rows_per_split);
    float acc[8], bsum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int mc = m_begin; mc < m_end; mc += 32) {
        float g4[4], xr[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int r = e >> 5, nn = e & 31;
            const int m = mc + r, n = n0 + nn;
            float v = 0.f;
            if (m < m_end && n < q.N) {
                const int ld = q.ldY ? q.ldY : q.N;
                v = q.dY[(size_t)m * ld + n];
                if (q.Y) v *= act_grad_from_out(q.Y[(size_t)m * ld + n], q.act);
            }
            g4[i] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int e = tid + 256 * i; sG[e >> 5][e & 31] = g4[i]; }
        __syncthreads();
        const int rmax = min(32, m_end - mc);
#pragma unroll 1
        for (int r0 = 0; r0 < 32; r0 += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float xv = 0.f;
                if (k < q.K && r0 + j < rmax) {
                    const int m = mc + r0 + j;
                    // X may be a concatenation [X | X2] (fc_common.0's input: fc's output and the head pose)
                    if (q.x_shift_T) {                      // X row m - 1 within the sequence, zero at its first step (h_prev of a scan)
                        xv = (m % q.x_shift_T) ? q.X[(size_t)(m - 1) * (q.ldX ? q.ldX : q.K1) + k] : 0.f;
                    } else {
                        xv = k < q.K1 ? q.X[(size_t)m * (q.ldX ? q.ldX : q.K1) + k] : (k - q.K1 < q.K2 ? q.X2[(size_t)m * q.K2 + (k - q.K1)] : 0.f);
                    }
                }
                xr[j] = xv;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = xr[j];
                const float4 g0 = *reinterpret_cast<const float4*>(&sG[r0 + j][ng * 8]);
                const float4 g1 = *reinterpret_cast<const float4*>(&sG[r0 + j][ng * 8 + 4]);
                acc[0] = fmaf(g0.x, xv, acc[0]); acc[1] = fmaf(g0.y, xv, acc[1]);
                acc[2] = fmaf(g0.z, xv, acc[2]); acc[3] = fmaf(g0.w, xv, acc[3]);
                acc[4] = fmaf(g1.x, xv, acc[4]); acc[5] = fmaf(g1.y, xv, acc[5]);
                acc[6] = fmaf(g1.z, xv, acc[6]); acc[7] = fmaf(g1.w, xv, acc[7]);
            }
        }
        if (q.db && bx == 0 && tid < 32)
            for (int r = 0; r < rmax; ++r) bsum += sG[r][tid];
    }
    if (k < q.K) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = n0 + ng * 8 + i;
            if (n < q.N) atomicAdd(q.dW + (size_t)n * (q.ldW ? q.ldW : q.K) + k, acc[i]);
        }
    }
    if (q.db && bx == 0 && tid < 32 && n0 + tid < q.N) atomicAdd(q.db + n0 + tid, bsum);
}

}  // namespace eve

extern "C" int eve_linear_wgrad_batch(const eve_wgrad_problem* problems, int n, eve_stream_t stream) {
    if (!problems || n <= 0 || n > EVE_WGRAD_BATCH_MAX) return set_error_msg("linear_wgrad_batch: bad arguments");
    eve_wgrad_batch b;
    b.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        eve_wgrad_problem q = problems[i];
        if (q.M <= 0 || q.N <= 0 || q.K <= 0 || !q.dY || !q.X || !q.dW || q.K1 <= 0 || q.K1 > q.K || (q.K2 > 0 && !q.X2) ||
            (q.act != EVE_ACT_NONE && !q.Y))
            return set_error_msg("linear_wgrad_batch: bad problem");
        const int tiles = ((q.K + 63) / 64) * ((q.N + 31) / 32);
        int splits = (256 + tiles - 1) / tiles;              // ~1 workgroup per CU and problem: the batch fills the chip together
        const int max_splits = (q.M + 63) / 64;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        int rows = (q.M + splits - 1) / splits;
        rows = (rows + 31) / 32 * 32;
        splits = (q.M + rows - 1) / rows;
        q.rows_per_split = rows;
        b.p[i] = q;
        b.first_block[i] = total;
        total += tiles * splits;
    }
    EVE_LAUNCH("linear_wgrad_batch_kernel", linear_wgrad_batch_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, b);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_linear_fwd_ex(int M, int K, int N, const float* x, int ldx, const float* w_in_out, const float* bias, int n_bias,
                                 int act, float* y, int ldy, eve_stream_t stream) {
    if (int e = ls_check(M, K, N, "linear_fwd_ex: bad shape")) return e;
    if (!x || !w_in_out || !y || ldx < K || ldy < N) return set_error_msg("linear_fwd_ex: bad arguments");
    EVE_LAUNCH("linear_mm_kernel", linear_mm_kernel, dim3((M + LS_TM - 1) / LS_TM, (N + LS_TN - 1) / LS_TN), dim3(256), 0, (hipStream_t)stream,
                       x, (const float*)nullptr, 0, w_in_out, bias, act, y, M, K, N, ldx, ldy, bias ? n_bias : 0, 0);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_linear_dgrad_ex(int M, int K, int N, const float* dy, int lddy, const float* y, int act, const float* w_out_in,
                                   float* dx, int lddx, int accumulate, eve_stream_t stream) {
    if (int e = ls_check(M, K, N, "linear_dgrad_ex: bad shape")) return e;
    if (!dy || !w_out_in || !dx || (act != EVE_ACT_NONE && !y) || lddy < N || lddx < K) return set_error_msg("linear_dgrad_ex: bad arguments");
    EVE_LAUNCH("linear_mm_kernel", linear_mm_kernel, dim3((M + LS_TM - 1) / LS_TM, (K + LS_TN - 1) / LS_TN), dim3(256), 0, (hipStream_t)stream,
                       dy, act != EVE_ACT_NONE ? y : (const float*)nullptr, act, w_out_in, (const float*)nullptr, 0, dx, M, N, K, lddy, lddx, 0,
                       accumulate);
    EVE_CHECK_LAUNCH();
    return 0;
}

namespace eve {
// gaze = pi/2 * tanh-output columns 0, 1 and pupil = ReLU-output column 0 of the two heads' 4-wide (padded) last layers,
// contiguous -- and the way back: d(head outputs) from the loss kernel's per-side unit gradients, scaled by d(full loss) read
// from the device.  One launch each instead of slice / mul / select and their backward's zeros + copies + adds.
__global__ __launch_bounds__(256) void tail_outputs_fwd_kernel(const int M, const float* __restrict__ g2, const float* __restrict__ p2,
                                                               float* __restrict__ gaze, float* __restrict__ pupil) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float4 g = reinterpret_cast<const float4*>(g2)[m];
    gaze[2 * m] = 1.5707963267948966f * g.x;
    gaze[2 * m + 1] = 1.5707963267948966f * g.y;
    pupil[m] = p2[4 * m];
}
__global__ __launch_bounds__(256) void tail_outputs_bwd_kernel(const int BT, const float* __restrict__ dg_l, const float* __restrict__ dg_r,
                                                               const float* __restrict__ dp_l, const float* __restrict__ dp_r,
                                                               const float* __restrict__ g_full, const float c_ang, const float c_l1,
                                                               float* __restrict__ d_g2, float* __restrict__ d_p2) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= 2 * BT) return;
    const float up = g_full ? *g_full : 1.f;
    const bool right = m >= BT;
    const int i = right ? m - BT : m;
    const float* dg = right ? dg_r : dg_l;
    const float* dp = right ? dp_r : dp_l;
    const float sa = 1.5707963267948966f * c_ang * up, sl = c_l1 * up;
    reinterpret_cast<float4*>(d_g2)[m] = make_float4(sa * dg[2 * i], sa * dg[2 * i + 1], 0.f, 0.f);
    reinterpret_cast<float4*>(d_p2)[m] = make_float4(sl * dp[i], 0.f, 0.f, 0.f);
}
// columns 128 .. 131 of the head-pose concatenation [M][132]: (h0, h1, 0, 0) from the left clips' rows, then the right ones'
__global__ __launch_bounds__(256) void tail_head_pose_kernel(const int BT, const float* __restrict__ hl, const float* __restrict__ hr,
                                                             float* __restrict__ cat, const int ld, const int col) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= 2 * BT) return;
    const float* h = m >= BT ? hr + 2 * (size_t)(m - BT) : hl + 2 * (size_t)m;
    *reinterpret_cast<float4*>(cat + (size_t)m * ld + col) = make_float4(h[0], h[1], 0.f, 0.f);
}
}  // namespace eve

extern "C" int eve_tail_head_pose(int BT, const float* h_left, const float* h_right, float* cat, int ld, int col, eve_stream_t stream) {
    if (BT <= 0 || !h_left || !h_right || !cat || ld < col + 4 || (ld & 3) || (col & 3)) return set_error_msg("tail_head_pose: bad arguments");
    EVE_LAUNCH("tail_head_pose_kernel", tail_head_pose_kernel, dim3((2 * BT + 255) / 256), dim3(256), 0, (hipStream_t)stream, BT, h_left, h_right, cat, ld, col);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_tail_outputs_fwd(int M, const float* g2, const float* p2, float* gaze, float* pupil, eve_stream_t stream) {
    if (M <= 0 || !g2 || !p2 || !gaze || !pupil) return set_error_msg("tail_outputs_fwd: bad arguments");
    EVE_LAUNCH("tail_outputs_fwd_kernel", tail_outputs_fwd_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, M, g2, p2, gaze, pupil);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_tail_outputs_bwd(int BT, const float* dg_l, const float* dg_r, const float* dp_l, const float* dp_r, const float* g_full,
                                    float coeff_ang, float coeff_l1, float* d_g2, float* d_p2, eve_stream_t stream) {
    if (BT <= 0 || !dg_l || !dg_r || !dp_l || !dp_r || !d_g2 || !d_p2) return set_error_msg("tail_outputs_bwd: bad arguments");
    EVE_LAUNCH("tail_outputs_bwd_kernel", tail_outputs_bwd_kernel, dim3((2 * BT + 255) / 256), dim3(256), 0, (hipStream_t)stream, BT, dg_l, dg_r, dp_l, dp_r,
                       g_full, coeff_ang, coeff_l1, d_g2, d_p2);
    EVE_CHECK_LAUNCH();
    return 0;
}
