// 3x3 / stride 1 / pad 1 convolutions between 16..64 (-> 128) channels on wide images (W = 64 or 128) as a ROW-STREAMING kernel (round 4).
// RefineNet's outermost level (refine_net.py:96-131 of the reference: 72x128 pixels, 16-64 channels, 960 frames = 8.8 M pixels)
// ran its 3x3 layers over PIXEL GROUPS on the halo kernel (ops.PAIR_FACTOR: F pixels as one 32-channel pixel, a filter with
// F^2 times the taps' weights, mostly zeros): 4x the MFMA work of the layer on a kernel whose loop tops out near 1 PFLOP/s --
// 0.205 ms for the 16 -> 16 layer's 566 MB, 2.75 TB/s.  Here a workgroup (4 waves) walks ONE image top to bottom with a ring of
// four input rows in LDS: every input row is fetched once (global -> registers two rows ahead -> LDS), an output row of W pixels is
// W / 16 tiles of 16 pixels dealt to the waves, a tile is 9 taps x CIN / 32 MFMAs per 16 output channels with the exact channel
// counts (16 input channels: two taps share one K = 32 MFMA -- lanes g < 2 read tap 2s, lanes g >= 2 tap 2s + 1), the filter sits
// in registers as A operands for the whole image, and the transposed product leaves a lane with consecutive output channels of
// one pixel (conv_1x1.h).  One workgroup barrier per output row.  LDS rows are [pixel -1 .. W][CIN] with the 16-byte chunk
// XOR-ed by a key of the pixel so that the 16 lanes of a read group (consecutive pixels, same chunk) hit 16 different bank groups.
// The data gradient is the same kernel on the [Cin][3][3][Cout] filter with the taps mirrored (FLIP).
#pragma once
#include "common.h"
#include "conv_1x1.h"

namespace eve {

template <int CIN, int COUT>
struct C3Geom {
    static constexpr int CV = CIN / 8;                                   // 16-byte chunks per pixel
    static constexpr int LCV = CV == 2 ? 1 : (CV == 4 ? 2 : 3);
    static constexpr int STEPS = CIN == 16 ? 5 : 9 * (CIN / 32);         // MFMAs per 16-pixel x 16-channel tile
    static constexpr int COUT_W = COUT > 32 ? 32 : COUT;                 // output channels per wave: 64 / 128 are dealt to 2 / 4 waves
    static constexpr int CS = COUT / COUT_W;                             // (each holds its slice of the filter: <= 144 registers)
    static constexpr int NT = COUT_W / 16;
    static constexpr int MAXW = 128;
    static constexpr int ROWB = (MAXW + 2) * CIN * 2;                    // bytes per LDS row
    static constexpr int LDS = 4 * ROWB;
};

struct C3Params {
    int N, H, W;
    float* stats;             // round 5: NULL, or [N][COUT][2] <- (mean, rstd) of every output plane (InstanceNorm2d statistics, biased
    float eps;                //   variance): the workgroup walks a whole image anyway, so the consumer's statistics pass is not needed
    int flip;                 // 0: forward (filter [COUT][3][3][CIN]);  1: data gradient (filter [rows = COUT][3][3][k = CIN] of the
                              //    transposed layout [Cin][3][3][Cout], taps mirrored)
    int act;
};

template <typename H_, int CIN, int COUT, bool ACC>
__global__ __launch_bounds__(256) void conv3x3_stream_kernel(const C3Params p, const H_* __restrict__ x, const H_* __restrict__ w,
                                                             const float* __restrict__ bias, H_* __restrict__ out) {
    using G = C3Geom<CIN, COUT>;
    constexpr int CV = G::CV, LCV = G::LCV, STEPS = G::STEPS, NT = G::NT, CS = G::CS, COUT_W = G::COUT_W;
    constexpr int NP = NT >= 2 ? NT / 2 : 1;
    __shared__ __attribute__((aligned(16))) char smem[G::LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, t = lane & 15, g = lane >> 4;
    const int n = blockIdx.x;
    const int cg = wave % CS;                      // this wave's slice of the output channels
    const int W = p.W, Himg = p.H;
    // ---- the filter as A operands: tile a, step s; row r = t <-> output channel co(a, t) (conv_1x1.h) ----
    uint4 wa[NT][STEPS];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const int co = cg * COUT_W + (NT >= 2 ? 32 * (a >> 1) + 8 * (t >> 2) + 4 * (a & 1) + (t & 3) : t);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            int tap, k0;
            if constexpr (CIN == 16) { tap = 2 * s + (g >> 1); k0 = 8 * (g & 1); }
            else { tap = s / (CIN / 32); k0 = 32 * (s % (CIN / 32)) + 8 * g; }
            uint4 q = make_uint4(0u, 0u, 0u, 0u);
            if (tap < 9) {
                const int wt = p.flip ? 8 - tap : tap;
                q = *reinterpret_cast<const uint4*>(w + ((size_t)co * 9 + wt) * CIN + k0);
            }
            wa[a][s] = q;
        }
    }
    float bv[NP][8];
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cg * COUT_W + (NT >= 2 ? 32 * a + 8 * g + e : 4 * g + (e & 3));
            bv[a][e] = bias ? bias[c] : 0.f;
        }
    // ---- lane constants of the B-operand reads: byte offset inside an LDS row of (tile pixel t + dx - 1, chunk) per step ----
    // pixel q = x0 + t + dx (0 .. W + 1 in halo coordinates: q = p + 1), chunk c: offset = (q * CV + (c ^ key(q))) * 16
    auto lds_off = [&](int q, int c) { return ((q << LCV) + (c ^ ((q >> (4 - LCV)) & (CV - 1)))) << 4; };
    const size_t img = (size_t)n * Himg * W;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + img * COUT), 0, (int)((uint32_t)Himg * W * COUT * 2), 0x00020000);
    const H_* xin = x + img * CIN;
    const int nvec = W * CV;                       // 16-byte vectors per input row
    constexpr int LPT = (G::MAXW * CV + 255) / 256; // loads per thread and row
    // zero the ring (halo columns and the row above the image stay zero)
    for (int i = tid; i < G::LDS / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    auto row_ptr = [&](int r) { return smem + ((r + 1) & 3) * G::ROWB; };          // input row r (-1 .. H) lives in slot (r + 1) & 3
    auto load_row = [&](int r, uint4 (&q)[LPT]) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int i = tid + 256 * j;
            q[j] = (r < Himg && i < nvec) ? reinterpret_cast<const uint4*>(xin + (size_t)r * W * CIN)[i] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto store_row = [&](int r, const uint4 (&q)[LPT]) {
        char* base = row_ptr(r);
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int i = tid + 256 * j;
            if (i < nvec) {
                const int px = i >> LCV, c = i & (CV - 1);
                *reinterpret_cast<uint4*>(base + lds_off(px + 1, c)) = q[j];
            }
        }
    };
    {   // rows 0 and 1
        uint4 q0[LPT], q1[LPT];
        load_row(0, q0);
        load_row(1, q1);
        store_row(0, q0);
        store_row(1, q1);
    }
    __syncthreads();
    const int tiles = W >> 4;
    // per-lane sums of the STORED (rounded) outputs and of their squares, both taken about the lane's FIRST value of the channel
    // (sh): a plane whose mean is far from zero against its spread -- a large bias -- would lose var = E[x^2] - mean^2 to
    // cancellation in float; the lanes' (mean, M2) pairs are combined pairwise at the end (Chan et al.)
    float st1[NP][8], st2[NP][8], sh[NP][8];
    int cnt = 0;                                   // pixels this lane has summed (the same for every lane of the wave)
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
        for (int e = 0; e < 8; ++e) { st1[a][e] = 0.f; st2[a][e] = 0.f; sh[a][e] = 0.f; }
    for (int y = 0; y < Himg; ++y) {
        uint4 qn[LPT];
        load_row(y + 2, qn);                       // (zeros below the image)
        for (int tile = wave / CS; tile < tiles; tile += 4 / CS) {
            const int x0 = tile << 4;
            f32x4_t acc[NT];
#pragma unroll
            for (int a = 0; a < NT; ++a) acc[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                int tap, c;
                if constexpr (CIN == 16) { tap = 2 * s + (g >> 1); c = g & 1; }
                else { tap = s / (CIN / 32); c = 4 * (s % (CIN / 32)) + g; }
                uint4 b = make_uint4(0u, 0u, 0u, 0u);
                if (tap < 9) {
                    const int dy = tap / 3, dx = tap - 3 * dy;
                    b = *reinterpret_cast<const uint4*>(smem + ((y + dy) & 3) * G::ROWB + lds_off(x0 + t + dx, c));      // input row y + dy - 1
                }
#pragma unroll
                for (int a = 0; a < NT; ++a) Elem<H_>::mfma(acc[a], wa[a][s], b);
            }
            // ---- epilogue (conv_1x1.h): bias, activation, accumulate, one store per pixel and tile pair ----
            const uint32_t pix = (uint32_t)(y * W + x0 + t);
#pragma unroll
            for (int a = 0; a < NP; ++a) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = acc[NT >= 2 ? 2 * a : 0][e] + bv[a][e];
                    o[4 + e] = NT >= 2 ? acc[NT >= 2 ? 2 * a + 1 : 0][e] + bv[a][4 + e] : 0.f;
                }
                if (p.act != EVE_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = act_fwd(o[e], p.act);
                }
                if constexpr (NT >= 2) {
                    const int off = (int)(pix * (uint32_t)(COUT * 2) + (uint32_t)(cg * COUT_W * 2) + 64u * a + 16u * g);
                    if constexpr (ACC) {
                        float pv[8];
                        Elem<H_>::unpack(__builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ro, off, 0, 0)), pv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += pv[e];
                    }
                    const uint4 qo = Elem<H_>::pack(o);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(c1_v4u32, qo), ro, off, 0, 0);
                    if (p.stats) {
                        float ro8[8];
                        Elem<H_>::unpack(qo, ro8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (cnt == 0) sh[a][e] = ro8[e];
                            const float d = ro8[e] - sh[a][e];
                            st1[a][e] += d; st2[a][e] += d * d;
                        }
                    }
                } else {
                    const int off = (int)(pix * 32u + 8u * g);
                    if constexpr (ACC) {
                        const c1_v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(ro, off, 0, 0);
                        float pv[8];
                        Elem<H_>::unpack(make_uint4(v.x, v.y, 0u, 0u), pv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] += pv[e];
                    }
                    const uint4 q = Elem<H_>::pack(o);
                    c1_v2u32 v;
                    v.x = q.x; v.y = q.y;
                    __builtin_amdgcn_raw_buffer_store_b64(v, ro, off, 0, 0);
                    if (p.stats) {
                        float ro8[8];
                        Elem<H_>::unpack(q, ro8);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (cnt == 0) sh[a][e] = ro8[e];
                            const float d = ro8[e] - sh[a][e];
                            st1[a][e] += d; st2[a][e] += d * d;
                        }
                    }
                }
            }
            ++cnt;
        }
        store_row(y + 2, qn);                      // slot of row y - 2: last read in iteration y - 1 (barrier below / above)
        __syncthreads();
    }
    if (p.stats) {
        // plane statistics in a fixed order: 16 pixel lanes of a row (butterfly), then the waves that share a channel slice
        // (the ring is free: every wave is past the loop's last barrier)
        float* red = reinterpret_cast<float*>(smem);             // [wave][NP][4 g][8 e][{mean, M2, count}]
        constexpr int EV = NT >= 2 ? 8 : 4;
        const float fc = (float)cnt, rc = cnt > 0 ? 1.f / fc : 0.f;
#pragma unroll
        for (int a = 0; a < NP; ++a)
#pragma unroll
            for (int e = 0; e < EV; ++e) {
                // the lane's (mean, M2 = sum of squared deviations from it) over its cnt pixels ...
                const float d1 = st1[a][e] * rc;
                float mu = sh[a][e] + d1, m2 = fmaxf(st2[a][e] - st1[a][e] * d1, 0.f), nn = fc;
                // ... merged with the 15 other pixel lanes of the row group, equal counts at every level
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    const float omu = __shfl_xor(mu, m, 64), om2 = __shfl_xor(m2, m, 64), dl = omu - mu;
                    m2 = (m2 + om2) + dl * dl * (0.5f * nn);
                    mu = 0.5f * (mu + omu);
                    nn *= 2.f;
                }
                if (t == 0) {
                    float* r3 = red + (((wave * NP + a) * 4 + g) * 8 + e) * 3;
                    r3[0] = mu; r3[1] = m2; r3[2] = nn;
                }
            }
        __syncthreads();
        if (tid < COUT) {
            const int c = tid, slice = c / COUT_W, cl = c - slice * COUT_W;
            int a, gg, e;
            if constexpr (NT >= 2) { a = cl >> 5; gg = (cl >> 3) & 3; e = cl & 7; }
            else { a = 0; gg = cl >> 2; e = cl & 3; }
            float mu = 0.f, m2 = 0.f, nn = 0.f;
            for (int wv = slice; wv < 4; wv += CS) {               // the waves of this channel slice, in order
                const float* r3 = red + (((wv * NP + a) * 4 + gg) * 8 + e) * 3;
                const float nb = r3[2];
                if (nb > 0.f) {
                    const float nt = nn + nb, dl = r3[0] - mu;
                    m2 = (m2 + r3[1]) + dl * dl * (nn * nb / nt);
                    mu += dl * (nb / nt);
                    nn = nt;
                }
            }
            const float var = nn > 0.f ? m2 / nn : 0.f;
            p.stats[((size_t)n * COUT + c) * 2] = mu;
            p.stats[((size_t)n * COUT + c) * 2 + 1] = rsqrtf(var + p.eps);
        }
    }
}

// true: launched.  x [N][H][W][Cin] -> out [N][H][W][Cout]; w: [Cout][3][3][Cin] (forward) or, with flip, the transposed layout of
// the data gradient (rows = this launch's output channels)
template <typename H_>
static bool launch_conv3x3_stream(int N, int Himg, int W, int Cin, int Cout, int flip, const void* x, const void* w, const float* bias,
                                  int epi_act, void* out, hipStream_t s, float* stats = nullptr, float eps = 1e-5f) {
    if (!g_cfg.conv3x3_stream || (W != 64 && W != 128) || Himg < 2 || (long long)N * Himg * W < 65536 ||
        (long long)Himg * W * (Cin > Cout ? Cin : Cout) * 2 >= (1ll << 31))
        return false;
    C3Params p;
    p.N = N; p.H = Himg; p.W = W; p.flip = flip; p.act = epi_act & 0xff; p.stats = stats; p.eps = eps;
    const bool accf = (epi_act & EVE_EPI_ACC) != 0;
#define EVE_C3_CASE(CI, CO)                                                                                                        \
    if (Cin == CI && Cout == CO) {                                                                                                 \
        if (accf) EVE_LAUNCH(EVE_HNAME(H_, "conv3x3_stream_kernel<", ", " #CI ", " #CO ", true>"), (conv3x3_stream_kernel<H_, CI, CO, true>), \
                             dim3(N), dim3(256), 0, s, p, (const H_*)x, (const H_*)w, bias, (H_*)out);                              \
        else EVE_LAUNCH(EVE_HNAME(H_, "conv3x3_stream_kernel<", ", " #CI ", " #CO ", false>"), (conv3x3_stream_kernel<H_, CI, CO, false>),    \
                        dim3(N), dim3(256), 0, s, p, (const H_*)x, (const H_*)w, bias, (H_*)out);                                   \
        return true;                                                                                                               \
    }
    EVE_C3_CASE(16, 16) EVE_C3_CASE(16, 32) EVE_C3_CASE(32, 16) EVE_C3_CASE(32, 32) EVE_C3_CASE(16, 64) EVE_C3_CASE(64, 16)
    // (64 -> 32 / 64 at 36x64 were built and measured: 0.223 ms against the halo kernel's 0.213 -- 18 K steps per tile with 237 registers;
    //  they stay on the halo kernel)
    EVE_C3_CASE(32, 64) EVE_C3_CASE(32, 128)
#undef EVE_C3_CASE
    return false;
}

}  // namespace eve
