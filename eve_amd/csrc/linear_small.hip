// Small float32 linear layers: the EyeNet tail (fc, fc_common, GRU input projection, gaze / pupil heads --
// /root/reference/src/models/eye_net.py:52-90) works on M = 2*B*T feature rows with K, N <= 512: sixteen products per step of
// 0.03-0.25 GFLOP each, every one a link of a dependent chain.  Rounds 2-5 ran them as 8 x 128 tiles of plain FMAs with both
// operands staged through LDS in 32-deep K chunks: 16 chunks x (global latency + two barriers) for fc, and an LDS pipe that the
// broadcast reads of the row operand kept 2.5 x busier than the FMAs (10-30 us per launch, 0.19 ms per step at any batch
// size).  Round 6: float32 MFMAs (v_mfma_f32_16x16x4_f32: an fmaf chain per output, the vector rate, NO operand traffic
// through LDS -- the broadcast is the matrix pipe's), operands straight from global memory / L2 into registers four K blocks
// ahead, the K range dealt over the four waves of a workgroup and summed through LDS once at the end.
//
//   eve_linear_fwd     y[M][N]  = act(x[M][K] . wt[K][N] + b)                       wt = IHWO pack ([in][out])
//   eve_linear_dgrad   dx[M][K] = (dy . act'(y))[M][N] . w[N][K]                    w  = OHWI pack ([out][in])
//   eve_linear_wgrad   dw[N][K] += (dy . act'(y))^T . x ;  db[N] += column sums     (float atomics over row splits)
#include "common.h"

namespace eve {

constexpr int LM_T = 32;       // a workgroup's output tile: 32 rows x 32 columns (2 x 2 MFMA tiles), K dealt over its 4 waves
constexpr int LM_GW = 2;       // the same for linear_wgrad_batch_kernel (its blocks carry 8 more words: Y), three workgroups per CU
constexpr int LM_G = 4;        // K blocks (16 k) per register group: one group computes while the next one's loads are in flight
                               // (linear_mm_kernel: 4 when a wave has more than two blocks (K > 128 + 16), else 2: the group
                               //  buffers are 64-72 registers each way, and at 250 one workgroup fills a CU)

// One K block of the operands of a 32 x 32 tile in MFMA layout (lane = (i = lane & 15, kk = lane >> 4), k = 16 kb + 4 kk + s):
// w[t][s]: the ROW operand (4 consecutive rows per lane in the result) of tile t at k word s, x[t]: the COLUMN operand's 4 k words.
// MFMA s of a block takes word s of every lane (a K permutation inside the block; a sum is order-free).
struct LmFrag { float w[2][4]; float x[2][4]; };

__device__ __forceinline__ void lm_mfma(f32x4_t (&acc)[2][2], const LmFrag& f) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[a][s], f.x[b][s], acc[a][b], 0, 0, 0);
}

// Sum of the four waves' partial tiles: waves 1-3 park theirs in LDS, wave 0 returns the total (the others return false).
__device__ __forceinline__ bool lm_reduce(f32x4_t (&acc)[2][2], float (*red)[16][64], const int wave, const int lane) {
    if (wave) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int v = 0; v < 4; ++v) red[wave - 1][(a * 2 + b) * 4 + v][lane] = acc[a][b][v];
    }
    __syncthreads();
    if (wave) return false;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[a][b][v] += red[w][(a * 2 + b) * 4 + v][lane];
    return true;
}

// act'(y) through the output, branch-free: (y > 0 ? p0 : n0) + (y > 0 ? 0 : n1) * y + q * y * y with wave-uniform constants --
// the per-element `switch` of act_grad_from_out between the operand loads put a branch and a full vmcnt wait behind every load.
struct LmActGrad { float p0, n0, n1, q; };
__device__ __forceinline__ LmActGrad lm_act_grad(const int act) {
    switch (act) {
        case EVE_ACT_RELU:    return {1.f, 0.f, 0.f, 0.f};
        case EVE_ACT_LEAKY:   return {1.f, 0.01f, 0.f, 0.f};
        case EVE_ACT_SELU:    return {EVE_SELU_SCALE, EVE_SELU_SCALE * EVE_SELU_ALPHA, 1.f, 0.f};
        case EVE_ACT_TANH:    return {1.f, 1.f, 0.f, -1.f};
        case EVE_ACT_SIGMOID: return {0.f, 0.f, 0.f, -1.f};        // y - y^2: the linear term is added below for both signs
        default:              return {1.f, 1.f, 0.f, 0.f};
    }
}
__device__ __forceinline__ float lm_apply_grad(const float g, const float y, const LmActGrad& c, const bool sigmoid) {
    const bool pos = y > 0.f;
    float d = fmaf(pos ? 0.f : c.n1, y, pos ? c.p0 : c.n0);
    d = fmaf(c.q * y, y, d);
    if (sigmoid) d += y;
    return g * d;
}

__device__ __forceinline__ void lm_act16(float (&v)[2][2][4], const int act) {
#define LM_ALL(expr)                                   \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)      \
    _Pragma("unroll") for (int b = 0; b < 2; ++b)      \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) { const float z = v[a][b][r]; v[a][b][r] = (expr); }
    switch (act) {
        case EVE_ACT_RELU:    LM_ALL(z > 0.f ? z : 0.f) break;
        case EVE_ACT_LEAKY:   LM_ALL(z > 0.f ? z : 0.01f * z) break;
        case EVE_ACT_SELU:    LM_ALL(EVE_SELU_SCALE * (z > 0.f ? z : EVE_SELU_ALPHA * (__expf(z) - 1.f))) break;
        case EVE_ACT_TANH:    LM_ALL(tanhf(z)) break;
        case EVE_ACT_SIGMOID: LM_ALL(1.f / (1.f + __expf(-z))) break;
        default: break;
    }
#undef LM_ALL
}

// C[M][Nc] = epi( A'[M][R] . B[R][Nc] ),  A' = A * act'(Y) if HASY (Y: same shape as A), epi = act(. + bias).
// Matrix roles: B's columns are the MFMA ROW operand (a lane ends up with 4 consecutive output columns of one row of C: one
// 16-byte store), A' the column operand (one 16-byte load per lane and K block when VEC: lda % 4 == 0, R % 4 == 0, 16-byte base).
// Every load is unconditional from a clamped address and masked by a select afterwards: a group's 40-72 loads leave back to back.
// lda: row stride of A and Y (the leading R columns of a wider matrix), ldc: row stride of C (a column range of a wider one),
// n_bias: entries of `bias` that exist (columns beyond take 0), accumulate: C += result (round 4: the tail as one autograd node
// writes fc's output into the head-pose concatenation, reads fc's gradient out of the concatenation's, and sums the two heads'
// input gradients in the second head's epilogue -- no cat / pad / slice / add launches)
template <bool VEC, bool HASY, int G>
__global__ __launch_bounds__(256) void linear_mm_kernel(const float* __restrict__ A, const float* __restrict__ Y, const int pro_act,
                                                        const float* __restrict__ B, const float* __restrict__ bias,
                                                        const int epi_act, float* __restrict__ C, const int M, const int R,
                                                        const int Nc, const int lda, const int ldc, const int n_bias,
                                                        const int accumulate) {
    __shared__ float red[3][16][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, kk = lane >> 4;
    const int m0 = blockIdx.x * LM_T, c0 = blockIdx.y * LM_T;
    const LmActGrad ag = lm_act_grad(pro_act);
    const bool sig = pro_act == EVE_ACT_SIGMOID;
    // per-lane row / column bases (clamped into the matrices) and their validity
    const float* bcol[2];
    const float* arow[2];
    const float* yrow[2];
    bool nok[2], mok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = c0 + t * 16 + i, m = m0 + t * 16 + i;
        nok[t] = n < Nc; mok[t] = m < M;
        bcol[t] = B + min(n, Nc - 1);
        arow[t] = A + (size_t)min(m, M - 1) * lda;
        yrow[t] = HASY ? Y + (size_t)min(m, M - 1) * lda : nullptr;
    }
    struct Raw { LmFrag f; float y[2][4]; };
    auto fetch = [&](const int kb, Raw& r) {
        const int k = kb * 16 + 4 * kk;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) r.f.w[t][s] = bcol[t][(size_t)min(k + s, R - 1) * Nc];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (VEC) {
                const int kc = min(k, R - 4);
                const float4 v = *reinterpret_cast<const float4*>(arow[t] + kc);
                r.f.x[t][0] = v.x; r.f.x[t][1] = v.y; r.f.x[t][2] = v.z; r.f.x[t][3] = v.w;
                if (HASY) {
                    const float4 y = *reinterpret_cast<const float4*>(yrow[t] + kc);
                    r.y[t][0] = y.x; r.y[t][1] = y.y; r.y[t][2] = y.z; r.y[t][3] = y.w;
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int kc = min(k + s, R - 1);
                    r.f.x[t][s] = arow[t][kc];
                    if (HASY) r.y[t][s] = yrow[t][kc];
                }
            }
        }
    };
    auto finish = [&](const int kb, Raw& r) {              // masks (and act') once the words have arrived
        const int k = kb * 16 + 4 * kk;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool kok = k + s < R;
                r.f.w[t][s] = (nok[t] && kok) ? r.f.w[t][s] : 0.f;
                float v = r.f.x[t][s];
                if (HASY) v = lm_apply_grad(v, r.y[t][s], ag, sig);
                r.f.x[t][s] = (mok[t] && kok) ? v : 0.f;
            }
    };
    f32x4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float bv[2][4];                                         // the lane's 8 bias words (wave 0 uses them in the epilogue)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = c0 + a * 16 + 4 * kk + r;
            const float w = (bias && n_bias > 0) ? bias[min(n, n_bias - 1)] : 0.f;
            bv[a][r] = n < n_bias ? w : 0.f;
        }
    const int nkb = (R + 15) >> 4;                          // wave w takes K blocks w, w + 4, ...
    Raw cur[G], nxt[G];
#pragma unroll
    for (int g = 0; g < G; ++g) fetch(wave + 4 * g, cur[g]);
    for (int kb = wave; kb < nkb; kb += 4 * G) {
        const bool more = kb + 4 * G < nkb;
        if (more) {
#pragma unroll
            for (int g = 0; g < G; ++g) fetch(kb + 4 * (G + g), nxt[g]);
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (kb + 4 * g < nkb) {
                finish(kb + 4 * g, cur[g]);
                lm_mfma(acc, cur[g].f);
            }
        if (more) {
#pragma unroll
            for (int g = 0; g < G; ++g) cur[g] = nxt[g];
        }
    }
    if (!lm_reduce(acc, red, wave, lane)) return;
    // epilogue of wave 0: the bias words were requested before the K loop, the words to accumulate onto leave in one batch, the
    // activation is one wave-uniform switch around all 16 values (a switch per value serialises on its loads' waits)
    const bool vec_c = VEC && !(ldc & 3) && !(Nc & 3) && !((uintptr_t)C & 15);
    float v[2][2][4], prev[2][2][4];
    float* cptr[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = m0 + b * 16 + i, n = c0 + a * 16 + 4 * kk;
            cptr[a][b] = C + (size_t)min(m, M - 1) * ldc + n;
            if (accumulate) {
                if (vec_c) {
                    const float4 p = *reinterpret_cast<const float4*>(C + (size_t)min(m, M - 1) * ldc + min(n, Nc - 4));
                    prev[a][b][0] = p.x; prev[a][b][1] = p.y; prev[a][b][2] = p.z; prev[a][b][3] = p.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) prev[a][b][r] = C[(size_t)min(m, M - 1) * ldc + min(n + r, Nc - 1)];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[a][b][r] = acc[a][b][r] + bv[a][r];
        }
    lm_act16(v, epi_act);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = m0 + b * 16 + i, n = c0 + a * 16 + 4 * kk;
            if (accumulate) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[a][b][r] += prev[a][b][r];
            }
            if (m >= M || n >= Nc) continue;
            if (vec_c) {
                *reinterpret_cast<float4*>(cptr[a][b]) = make_float4(v[a][b][0], v[a][b][1], v[a][b][2], v[a][b][3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < Nc) cptr[a][b][r] = v[a][b][r];
            }
        }
}

static inline bool lm_vec_ok(const void* a, const void* y, int lda, int R) {
    return !(lda & 3) && !(R & 3) && !((uintptr_t)a & 15) && !((uintptr_t)y & 15);
}
#define LM_LAUNCH(M_, Nc_, R_, ...)                                                                                                  \
    do {                                                                                                                         \
        const dim3 grid_(((M_) + LM_T - 1) / LM_T, ((Nc_) + LM_T - 1) / LM_T);                                                   \
        const int big_ = R_ > 160;                                                                                                          \
        if (vec_ && hasy_ && big_) EVE_LAUNCH("linear_mm_kernel", (linear_mm_kernel<true, true, 4>), grid_, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);   \
        else if (vec_ && hasy_) EVE_LAUNCH("linear_mm_kernel", (linear_mm_kernel<true, true, 2>), grid_, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);      \
        else if (vec_ && big_) EVE_LAUNCH("linear_mm_kernel", (linear_mm_kernel<true, false, 4>), grid_, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);      \
        else if (vec_) EVE_LAUNCH("linear_mm_kernel", (linear_mm_kernel<true, false, 2>), grid_, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);              \
        else if (hasy_) EVE_LAUNCH("linear_mm_kernel", (linear_mm_kernel<false, true, 2>), grid_, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);             \
        else EVE_LAUNCH("linear_mm_kernel", (linear_mm_kernel<false, false, 2>), grid_, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);                       \
    } while (0)

// several weight / bias gradients in one launch: workgroup -> (problem, 32 k x 32 n tile, row split) through a prefix table.
// dW[n][k] += sum_m G[m][n] X[m][k], G = dY * act'(Y): G's columns are the MFMA row operand, X's the column operand (a lane ends up
// with one k of four consecutive dW rows: the 16 lanes of a row group add to 64 contiguous bytes); both operands are read as
// 64-byte row segments per 16 lanes, 16 rows of m per K block.  The four
// waves of the workgroup take the split's 16-row blocks in turn; one set of float atomics per workgroup.  db: the column sums
// of G ride along in the k-tile-0 workgroups (the G words are already in registers).
__global__ __launch_bounds__(256) void linear_wgrad_batch_kernel(const eve_wgrad_batch b) {
    __shared__ float red[3][16][64];
    __shared__ float bred[4][32];
    int pi = 0;
    while (pi + 1 < b.n && (int)blockIdx.x >= b.first_block[pi + 1]) ++pi;
    const eve_wgrad_problem& q = b.p[pi];
    const int local = blockIdx.x - b.first_block[pi];
    const int tk = (q.K + LM_T - 1) / LM_T, tn = (q.N + LM_T - 1) / LM_T;
    const int bx = local % tk, by = (local / tk) % tn, bz = local / (tk * tn);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, kk = lane >> 4;
    const int k0 = bx * LM_T, n0 = by * LM_T;
    const int m_begin = bz * q.rows_per_split, m_end = min(q.M, m_begin + q.rows_per_split);
    const int ldy = q.ldY ? q.ldY : q.N, ldx = q.ldX ? q.ldX : q.K1;
    const bool want_db = q.db && bx == 0;
    // (a problem without Y reads dY in its place and multiplies by act'(.) of "none" = 1: a run-time `if (Y)` around the loads keeps
    //  the register groups from being promoted out of scratch)
    const int act = q.Y ? q.act : EVE_ACT_NONE;
    const LmActGrad ag = lm_act_grad(act);
    const bool sig = act == EVE_ACT_SIGMOID;
    constexpr bool hasy = true;
    const int shiftT = q.x_shift_T;
    float bsum[2] = {0.f, 0.f};
    // per-lane column bases and row strides, fixed for the launch: X may be a concatenation [X | X2] (fc_common.0's input: fc's
    // output and the head pose), or row m - 1 within the sequence, zero at its first step (h_prev of a scan: x_shift_T)
    const float* xcol[2];
    const float* gcol[2];
    const float* ycol[2];
    int xstride[2];
    bool kok[2], nok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int k = k0 + t * 16 + i, n = n0 + t * 16 + i;
        nok[t] = n < q.N;
        gcol[t] = q.dY + min(n, q.N - 1);
        ycol[t] = (q.Y ? q.Y : q.dY) + min(n, q.N - 1);
        if (shiftT || k < q.K1) { kok[t] = k < (shiftT ? q.K : q.K1); xcol[t] = q.X + min(k, q.K1 - 1); xstride[t] = ldx; }
        else if (k - q.K1 < q.K2) { kok[t] = true; xcol[t] = q.X2 + (k - q.K1); xstride[t] = q.K2; }
        else { kok[t] = false; xcol[t] = q.X; xstride[t] = ldx; }
    }
    struct Raw { LmFrag f; float y[2][4]; };
    auto fetch = [&](const int mb, Raw& r) {                // rows m_begin + 16 mb + 4 kk + s: clamped addresses, masked in finish
        const int m = m_begin + mb * 16 + 4 * kk;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int mm = min(m + s, m_end - 1);
            const int mx = shiftT ? max(mm - 1, 0) : mm;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                r.f.x[t][s] = xcol[t][(size_t)mx * xstride[t]];
                r.f.w[t][s] = gcol[t][(size_t)mm * ldy];
                if (hasy) r.y[t][s] = ycol[t][(size_t)mm * ldy];
            }
        }
    };
    auto finish = [&](const int mb, Raw& r) {
        const int m = m_begin + mb * 16 + 4 * kk;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool mok = m + s < m_end;
            const bool xok = mok && (!shiftT || ((m + s) % shiftT) != 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                r.f.x[t][s] = (xok && kok[t]) ? r.f.x[t][s] : 0.f;
                float g = r.f.w[t][s];
                if (hasy) g = lm_apply_grad(g, r.y[t][s], ag, sig);
                r.f.w[t][s] = (mok && nok[t]) ? g : 0.f;
            }
        }
    };
    f32x4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int nmb = (m_end - m_begin + 15) >> 4;
    Raw cur[LM_GW], nxt[LM_GW];
#pragma unroll
    for (int g = 0; g < LM_GW; ++g) fetch(wave + 4 * g, cur[g]);
    for (int mb = wave; mb < nmb; mb += 4 * LM_GW) {
        const bool more = mb + 4 * LM_GW < nmb;
        if (more) {
#pragma unroll
            for (int g = 0; g < LM_GW; ++g) fetch(mb + 4 * (LM_GW + g), nxt[g]);
        }
#pragma unroll
        for (int g = 0; g < LM_GW; ++g)
            if (mb + 4 * g < nmb) {
                finish(mb + 4 * g, cur[g]);
                lm_mfma(acc, cur[g].f);
                if (want_db) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) bsum[t] += (cur[g].f.w[t][0] + cur[g].f.w[t][1]) + (cur[g].f.w[t][2] + cur[g].f.w[t][3]);
                }
            }
        if (more) {
#pragma unroll
            for (int g = 0; g < LM_GW; ++g) cur[g] = nxt[g];
        }
    }
    if (want_db) {                                        // lanes i, i + 16, i + 32, i + 48 hold the same column
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bsum[t] += __shfl_xor(bsum[t], 16, 64);
            bsum[t] += __shfl_xor(bsum[t], 32, 64);
            if (kk == 0) bred[wave][t * 16 + i] = bsum[t];
        }
    }
    const bool first = lm_reduce(acc, red, wave, lane);      // (barrier inside: bred is complete behind it)
    if (want_db && tid < 32 && n0 + tid < q.N) atomicAdd(q.db + n0 + tid, (bred[0][tid] + bred[1][tid]) + (bred[2][tid] + bred[3][tid]));
    if (!first) return;
    const int ldw = q.ldW ? q.ldW : q.K;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // acc[a][c]: rows n0 + 16 a + 4 kk + r, column k0 + 16 c + i: an atomic instruction covers 16 consecutive k of four rows
            const int n = n0 + a * 16 + 4 * kk, k = k0 + c * 16 + i;
            if (k >= q.K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < q.N) atomicAdd(q.dW + (size_t)(n + r) * ldw + k, acc[a][c][r]);
        }
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
    const bool vec_ = lm_vec_ok(x, nullptr, K, K), hasy_ = false;
    LM_LAUNCH(M, N, K, x, (const float*)nullptr, 0, w_in_out, bias, act, y, M, K, N, K, N, N, 0);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_linear_dgrad(int M, int K, int N, const float* dy, const float* y, int act, const float* w_out_in,
                                float* dx, eve_stream_t stream) {
    if (int e = ls_check(M, K, N, "linear_dgrad: bad shape")) return e;
    if (!dy || !w_out_in || !dx || (act != EVE_ACT_NONE && !y)) return set_error_msg("linear_dgrad: null pointer");
    const float* yy = act != EVE_ACT_NONE ? y : (const float*)nullptr;
    const bool vec_ = lm_vec_ok(dy, yy, N, N), hasy_ = yy != nullptr;
    LM_LAUNCH(M, K, N, dy, yy, act, w_out_in, (const float*)nullptr, 0, dx, M, N, K, N, K, 0, 0);
    EVE_CHECK_LAUNCH();
    return 0;
}

// Row splits of a weight-gradient launch: ~512 workgroups over all its tiles (two per CU; every split is a set of atomics), whole
// 64-row steps (four waves x one 16-row block) per split; every problem of a batch has the same M in practice.
static int lm_wgrad_tiles(const eve_wgrad_problem& q) { return ((q.K + LM_T - 1) / LM_T) * ((q.N + LM_T - 1) / LM_T); }
static int lm_wgrad_split(eve_wgrad_problem& q, int total_tiles) {
    int splits = (512 + total_tiles / 2) / total_tiles;
    const int max_splits = (q.M + 63) / 64;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int rows = (q.M + splits - 1) / splits;
    rows = (rows + 63) / 64 * 64;
    q.rows_per_split = rows;
    return (q.M + rows - 1) / rows;
}

extern "C" int eve_linear_wgrad(int M, int K, int N, const float* dy, const float* y, int act, const float* x, float* dw_out_in,
                                float* db, eve_stream_t stream) {
    if (int e = ls_check(M, K, N, "linear_wgrad: bad shape")) return e;
    if (!dy || !x || !dw_out_in || (act != EVE_ACT_NONE && !y)) return set_error_msg("linear_wgrad: null pointer");
    eve_wgrad_batch b = {};
    eve_wgrad_problem& q = b.p[0];
    q.dY = dy; q.Y = act != EVE_ACT_NONE ? y : nullptr; q.X = x; q.dW = dw_out_in; q.db = db;
    q.M = M; q.N = N; q.K = K; q.K1 = K; q.act = act;
    const int tiles = lm_wgrad_tiles(q), splits = lm_wgrad_split(q, tiles);      // (ls_check: tiles >= 1)
    b.n = 1;
    EVE_LAUNCH("linear_wgrad_batch_kernel", linear_wgrad_batch_kernel, dim3(tiles * splits), dim3(256), 0, (hipStream_t)stream, b);
    EVE_CHECK_LAUNCH();
    return 0;
}

// Round 4: the tail's weight / bias gradients batched into one launch (round 6: on float32 MFMAs, see the kernel)
extern "C" int eve_linear_wgrad_batch(const eve_wgrad_problem* problems, int n, eve_stream_t stream) {
    if (!problems || n <= 0 || n > EVE_WGRAD_BATCH_MAX) return set_error_msg("linear_wgrad_batch: bad arguments");
    eve_wgrad_batch b;
    b.n = n;
    int total = 0, total_tiles = 0;
    for (int i = 0; i < n; ++i) total_tiles += lm_wgrad_tiles(problems[i]);
    if (total_tiles < 1) total_tiles = 1;
    for (int i = 0; i < n; ++i) {
        eve_wgrad_problem q = problems[i];
        if (q.M <= 0 || q.N <= 0 || q.K <= 0 || !q.dY || !q.X || !q.dW || q.K1 <= 0 || q.K1 > q.K || (q.K2 > 0 && !q.X2) ||
            (q.act != EVE_ACT_NONE && !q.Y))
            return set_error_msg("linear_wgrad_batch: bad problem");
        const int splits = lm_wgrad_split(q, total_tiles);
        b.p[i] = q;
        b.first_block[i] = total;
        total += lm_wgrad_tiles(q) * splits;
    }
    EVE_LAUNCH("linear_wgrad_batch_kernel", linear_wgrad_batch_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, b);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_linear_fwd_ex(int M, int K, int N, const float* x, int ldx, const float* w_in_out, const float* bias, int n_bias,
                                 int act, float* y, int ldy, eve_stream_t stream) {
    if (int e = ls_check(M, K, N, "linear_fwd_ex: bad shape")) return e;
    if (!x || !w_in_out || !y || ldx < K || ldy < N) return set_error_msg("linear_fwd_ex: bad arguments");
    const bool vec_ = lm_vec_ok(x, nullptr, ldx, K), hasy_ = false;
    LM_LAUNCH(M, N, K, x, (const float*)nullptr, 0, w_in_out, bias, act, y, M, K, N, ldx, ldy, bias ? n_bias : 0, 0);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_linear_dgrad_ex(int M, int K, int N, const float* dy, int lddy, const float* y, int act, const float* w_out_in,
                                   float* dx, int lddx, int accumulate, eve_stream_t stream) {
    if (int e = ls_check(M, K, N, "linear_dgrad_ex: bad shape")) return e;
    if (!dy || !w_out_in || !dx || (act != EVE_ACT_NONE && !y) || lddy < N || lddx < K) return set_error_msg("linear_dgrad_ex: bad arguments");
    const float* yy = act != EVE_ACT_NONE ? y : (const float*)nullptr;
    const bool vec_ = lm_vec_ok(dy, yy, lddy, N), hasy_ = yy != nullptr;
    LM_LAUNCH(M, K, N, dy, yy, act, w_out_in, (const float*)nullptr, 0, dx, M, N, K, lddy, lddx, 0, accumulate);
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
