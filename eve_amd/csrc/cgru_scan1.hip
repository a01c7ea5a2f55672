// RefineNet's conv-GRU bottleneck over a whole clip, 16-bit formats, ONE SEQUENCE PER WORKGROUP (round 5).
// Same arithmetic, operands, outputs and rounding points as cgru_scan.hip (CGRUCell, /root/reference/src/models/common.py:388-415,
// applied per frame by refine_net.py:132-176); what changes is how the work is dealt.
//
// Why: the scan is T-sequential, so its duration is the latency of one frame times T, and cgru_scan.hip puts three sequences
// into a workgroup (120 pixels = one 128-pixel MFMA tile): a configs[2] batch is 11 workgroups and a configs[4] batch 3, on
// 256 CUs, each running 16 MFMAs and two ~1 000-instruction epilogues per wave and step / frame.  Here a workgroup owns one
// sequence (40 pixels = three 16-pixel tiles, the last half empty) and its four waves split the OUTPUT CHANNELS of every
// GEMM four ways -- 32 of gates_1's 128, 16 of gate_2's 64 -- so a wave runs 6 (3) MFMAs where it ran 16, both epilogues
// shrink by the same factor and no wave idles during gate_2; B workgroups instead of B / 3.  A step takes TWO 32-channel
// slices of one filter tap (K = 64: 18 + 18 steps per frame instead of 36 + 36), because with this little arithmetic per
// step the barrier and the DMA bookkeeping are what a step costs.
//
// What used to stay inside a wave now crosses waves through LDS: the update gate u (waves 2, 3 produce it, every wave
// blends with it) lives in two more 16-bit planes; in the backward the float carry into the previous state and dx_2 live in
// two float planes, every element of which is read and written by exactly one lane per phase (no atomics), phases separated
// by the workgroup barriers that were there anyway.
//
// LDS: 16-bit planes of 72 halo pixels x 64 B (7 x 10 halo of the 5 x 8 image; 32 channels per plane, chunk swizzle keyed by
// the halo row's parity as in cgru_scan.hip), a 4-slot ring of 16 KB filter tiles (two 128-row x 64-B sub-tiles), software
// pipelined like cgru_scan.hip: step g issues tile g+4, reads the fragments of step g+1, multiplies step g.
#include "common.h"
#include "lds_dma.h"

namespace eve {

constexpr int S1_PIX = 40, S1_C = 64;
constexpr int S1_PLANE = 72 * 64;               // bytes per 32-channel halo plane
constexpr int S1_SUB = 128 * 64;                // one filter sub-tile: 128 rows x 32 k
constexpr int S1_SLOT = 2 * S1_SUB;             // a step's tile: two slices
constexpr int S1_NT = 256;

typedef uint32_t s1_u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t s1_u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float s1_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
__device__ __forceinline__ float s1_tanh(float z) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * z)); }

// byte offset inside a plane PAIR base of 4 channels c..c+3 (c % 4 == 0, c < 64) of pixel m: plane c >> 5, halo pixel, chunk
__device__ __forceinline__ int s1_c4(int m, int c) {
    const int hr = (m >> 3) + 1, hc = (m & 7) + 1;
    return (c >> 5) * S1_PLANE + ((hr * 10 + hc) << 6) + (((((c & 31) >> 3)) ^ ((hr & 1) << 1)) << 4) + (c & 7) * 2;
}
// A fragment stays one 16-byte VECTOR value from the LDS load to the MFMA operand (round 6: as HIP's uint4 struct the load was
// split in the middle end and part of the fragments came back as ds_read2_b64 -- twice the LDS cycles of ds_read_b128 and
// banked differently from what the chunk swizzle is built for; see stem_fused.hip).
typedef s1_u32x4_t s1_frag_t;
__device__ __forceinline__ s1_frag_t s1_lds16(uint32_t a) { return *reinterpret_cast<const EVE_LDS s1_u32x4_t*>((uintptr_t)a); }
template <typename H>
__device__ __forceinline__ void s1_mfma(f32x4_t& acc, const s1_frag_t& a, const s1_frag_t& b) {
    if constexpr (Elem<H>::IS_BF16)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    else
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}
__device__ __forceinline__ s1_u32x2_t s1_lds8(uint32_t a) { return *reinterpret_cast<const EVE_LDS s1_u32x2_t*>((uintptr_t)a); }
__device__ __forceinline__ void s1_st8(uint32_t a, uint32_t x, uint32_t y) {
    *reinterpret_cast<EVE_LDS s1_u32x2_t*>((uintptr_t)a) = s1_u32x2_t{x, y};
}
__device__ __forceinline__ float s1_ldsf(uint32_t a) { return *reinterpret_cast<const EVE_LDS float*>((uintptr_t)a); }
__device__ __forceinline__ void s1_stf(uint32_t a, float v) { *reinterpret_cast<EVE_LDS float*>((uintptr_t)a) = v; }

// Fragment addresses shared by both kernels.  Lane (li, lg): pixel tile mt -> pixel 16 mt + li (clamped to 39: the surplus
// columns of the third tile are computed and dropped), K chunk lg.  Tap (dy, dx) adds (dy * 10 + dx) * 64; the chunk key is
// the halo row's parity, so there is one base per parity of dy.
struct S1Lane {
    int abase[2][3];
    __device__ __forceinline__ void init(int li, int lg) {
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            const int m = min(mt * 16 + li, S1_PIX - 1), py = m >> 3, px = m & 7;
#pragma unroll
            for (int q = 0; q < 2; ++q) abase[q][mt] = ((py * 10 + px) << 6) + ((lg ^ (((py + q) & 1) << 1)) << 4);
        }
    }
};

// One K = 64 step: acc[mt][nt] += W[rows of this wave][tap, 64 k] x X[pixels][tap, 64 k].  fx / fw: [slice][tile].
template <typename H, int NTL>
__device__ __forceinline__ void s1_mma(f32x4_t (&acc)[3][NTL], const s1_frag_t (&fx)[2][3], const s1_frag_t (&fw)[2][NTL]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) s1_mfma<H>(acc[mt][nt], fw[j][nt], fx[j][mt]);
}

// ------------------------------------------------------------------------------------------------------------------------
// forward.  planes (16-bit): 0,1 = x; 2,3 = h; 4,5 = r * h; 6,7 = u.
// stream of filter tiles per frame: conv1 (gates_1: 128 rows): (pair 0 = x, pair 1 = h) x 9 taps, then conv2 (gate_2: 64
// rows): (pair 0 = r * h, pair 1 = x) x 9 taps -- 36 tiles.
// ------------------------------------------------------------------------------------------------------------------------
template <typename H>
__global__ __launch_bounds__(S1_NT) void cgru_scan1_fwd_kernel(const int B, const int T, const H* __restrict__ xs,
                                                               const H* __restrict__ h0, const H* __restrict__ w1,
                                                               const float* __restrict__ b1, const H* __restrict__ w2,
                                                               const float* __restrict__ b2, H* __restrict__ hs,
                                                               H* __restrict__ hs_tm, H* __restrict__ ru,
                                                               H* __restrict__ rh, H* __restrict__ og) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int K1 = 9 * 128;
    const uint32_t lds0 = lds_addr_of(smem), ldsB = lds0 + 8 * S1_PLANE;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    for (int i = tid; i < 8 * S1_PLANE / 16; i += S1_NT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

    const eve_int4 rs_w1 = make_rsrc_words(w1, 128 * K1 * 2);
    const eve_int4 rs_w2 = make_rsrc_words(w2, 64 * K1 * 2);
    S1Lane L;
    L.init(li, lg);
    // filter fragment rows: conv1 channels 32 wave + 16 nt + li, conv2 channels 16 wave + li
    int brow1[2], brow2;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int c = wave * 32 + nt * 16 + li;
        brow1[nt] = (c << 6) + ((lg ^ (((c >> 2) & 1) << 1)) << 4);
    }
    {
        const int c = wave * 16 + li;
        brow2 = (c << 6) + ((lg ^ (((c >> 2) & 1) << 1)) << 4);
    }
    // filter DMA: thread -> (row, chunk) of each 128-row sub-tile, two pieces of 64 rows
    int b_rel[2], b_row[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = tid + S1_NT * j;
        b_row[j] = q >> 2;
        b_rel[j] = (b_row[j] * K1) * 2 + (((q & 3) ^ (((b_row[j] >> 2) & 1) << 1)) << 4);
    }
    // tile of stream position pos (0..35 within a frame) into ring slot
    auto issue = [&](int pos, int slot, bool live) {
        const int conv = pos >= 18 ? 1 : 0, q = pos - 18 * conv, pair = q / 9, tap = q - 9 * pair;
        const uint32_t dst = ldsB + slot * S1_SLOT + wave * 1024;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int koff = (tap * 128 + (2 * pair + s2) * 32) * 2;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool ok = live && (conv == 0 || b_row[j] < 64);
                lds_dma16_asm(conv == 0 ? rs_w1 : rs_w2, dst + s2 * S1_SUB + j * 4096, ok ? b_rel[j] + koff : EVE_OOB);
            }
        }
    };
    // x tile staging: 40 pixels x 8 chunks of 16 B = 320 slots
    int x_glob[2], x_lds[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = tid + S1_NT * j, q = e >> 3, part = e & 7;
        const bool ok = e < S1_PIX * 8;
        const int hr = (q >> 3) + 1, hc = (q & 7) + 1;
        x_glob[j] = ok ? ((b * T) * S1_PIX + q) * S1_C + part * 8 : -1;
        x_lds[j] = (part >> 2) * S1_PLANE + ((hr * 10 + hc) << 6) + (((part & 3) ^ ((hr & 1) << 1)) << 4);
    }
    __syncthreads();
    if (h0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (x_glob[j] >= 0)
                *reinterpret_cast<uint4*>(smem + 2 * S1_PLANE + x_lds[j]) =
                    *reinterpret_cast<const uint4*>(h0 + ((size_t)b * S1_PIX + ((tid + S1_NT * j) >> 3)) * S1_C + ((tid + S1_NT * j) & 7) * 8);
    }
    uint4 xq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) xq[j] = x_glob[j] >= 0 ? *reinterpret_cast<const uint4*>(xs + x_glob[j]) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (x_glob[j] >= 0) *reinterpret_cast<uint4*>(smem + x_lds[j]) = xq[j];

    issue(0, 0, true); issue(1, 1, true); issue(2, 2, true); issue(3, 3, true);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // tiles 0 and 1 have landed
    __syncthreads();

    // the lane's output pixels (pixel tile mt, column li)
    bool pok[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) pok[mt] = mt * 16 + li < S1_PIX;

    uint32_t gs = 0;                                              // stream position (ring phase)
    for (int t = 0; t < T; ++t) {
        const bool more_t = t + 1 < T;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            xq[j] = (more_t && x_glob[j] >= 0) ? *reinterpret_cast<const uint4*>(xs + x_glob[j] + (size_t)(t + 1) * S1_PIX * S1_C)
                                               : make_uint4(0, 0, 0, 0);
        const size_t ptm0 = ((size_t)t * B + b) * S1_PIX;         // time-major pixel base
        const size_t pg0 = ((size_t)b * T + t) * S1_PIX;
        // ================= conv1: gates_1 over cat[x, h] =================
        {
            f32x4_t acc[3][2];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            s1_frag_t fxA[2][3], fwA[2][2], fxB[2][3], fwB[2][2];
            auto load = [&](int q, uint32_t gpos, s1_frag_t (&fx)[2][3], s1_frag_t (&fw)[2][2]) {
                const int pair = q / 9, tap = q - 9 * pair;       // (q is a compile-time constant after unrolling)
                const int dy = tap / 3, dx = tap - 3 * dy;
                const uint32_t la = lds0 + (2 * pair) * S1_PLANE + (dy * 10 + dx) * 64, lb = ldsB + (gpos & 3) * S1_SLOT;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) fx[j][mt] = s1_lds16(la + j * S1_PLANE + L.abase[dy & 1][mt]);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) fw[j][nt] = s1_lds16(lb + j * S1_SUB + brow1[nt]);
                }
            };
            load(0, gs, fxA, fwA);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 18; ++q) {
                issue(q + 4, (int)((gs + q + 4) & 3), true);
                if (q < 17) {
                    if (q & 1) load(q + 1, gs + q + 1, fxA, fwA);
                    else       load(q + 1, gs + q + 1, fxB, fwB);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (q & 1) s1_mma<H, 2>(acc, fxB, fwB);
                else       s1_mma<H, 2>(acc, fxA, fwA);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(8)" ::: "memory");
                __syncthreads();
            }
            gs += 18;
            // ---- r, u = sigmoid(gates_1 + b): waves 0, 1 hold r (channels 0..63), waves 2, 3 hold u ----
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int gch = wave * 32 + nt * 16 + lg * 4;     // gate channel 0..127
                const float4 bv = *reinterpret_cast<const float4*>(b1 + gch);
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) {
                    if (!pok[mt]) continue;
                    const int m = mt * 16 + li;
                    float v[4] = {s1_sigmoid(acc[mt][nt][0] + bv.x), s1_sigmoid(acc[mt][nt][1] + bv.y),
                                  s1_sigmoid(acc[mt][nt][2] + bv.z), s1_sigmoid(acc[mt][nt][3] + bv.w)};
                    const uint32_t p0 = Elem<H>::pack2(v[0], v[1]), p1 = Elem<H>::pack2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(ru + (ptm0 + m) * 128 + gch) = make_uint2(p0, p1);
                    if (wave < 2) {                               // the stored (16-bit) gate is the one every later stage sees
                        v[0] = Elem<H>::lo(p0); v[1] = Elem<H>::hi(p0); v[2] = Elem<H>::lo(p1); v[3] = Elem<H>::hi(p1);
                        const s1_u32x2_t hq = s1_lds8(lds0 + 2 * S1_PLANE + s1_c4(m, gch));
                        const uint32_t q0 = Elem<H>::pack2(v[0] * Elem<H>::lo(hq.x), v[1] * Elem<H>::hi(hq.x));
                        const uint32_t q1 = Elem<H>::pack2(v[2] * Elem<H>::lo(hq.y), v[3] * Elem<H>::hi(hq.y));
                        s1_st8(lds0 + 4 * S1_PLANE + s1_c4(m, gch), q0, q1);
                        *reinterpret_cast<uint2*>(rh + (ptm0 + m) * S1_C + gch) = make_uint2(q0, q1);
                    } else {
                        s1_st8(lds0 + 6 * S1_PLANE + s1_c4(m, gch - 64), p0, p1);
                    }
                }
            }
            __syncthreads();                                      // r * h and u visible
        }
        // ================= conv2: gate_2 over cat[r * h, x] =================
        {
            f32x4_t acc[3][1];
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[a][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            s1_frag_t fxA[2][3], fwA[2][1], fxB[2][3], fwB[2][1];
            auto load = [&](int q, uint32_t gpos, s1_frag_t (&fx)[2][3], s1_frag_t (&fw)[2][1]) {
                const int pair = q / 9, tap = q - 9 * pair;
                const int dy = tap / 3, dx = tap - 3 * dy;
                const int plane = pair == 0 ? 4 : 0;              // r * h, then x
                const uint32_t la = lds0 + plane * S1_PLANE + (dy * 10 + dx) * 64, lb = ldsB + (gpos & 3) * S1_SLOT;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) fx[j][mt] = s1_lds16(la + j * S1_PLANE + L.abase[dy & 1][mt]);
                    fw[j][0] = s1_lds16(lb + j * S1_SUB + brow2);
                }
            };
            load(0, gs, fxA, fwA);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 18; ++q) {
                const int np = 18 + q + 4;                        // tile of stream position +4 (wraps into the next frame)
                issue(np >= 36 ? np - 36 : np, (int)((gs + q + 4) & 3), np >= 36 ? more_t : true);
                if (q < 17) {
                    if (q & 1) load(q + 1, gs + q + 1, fxA, fwA);
                    else       load(q + 1, gs + q + 1, fxB, fwB);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (q & 1) s1_mma<H, 1>(acc, fxB, fwB);
                else       s1_mma<H, 1>(acc, fxA, fwA);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(8)" ::: "memory");
                __syncthreads();
            }
            gs += 18;
            // ---- o = tanh(gate_2 + b);  h' = (1 - u) o + u h ----
            const int c = wave * 16 + lg * 4;
            const float4 bv = *reinterpret_cast<const float4*>(b2 + c);
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                if (!pok[mt]) continue;
                const int m = mt * 16 + li;
                float o[4] = {s1_tanh(acc[mt][0][0] + bv.x), s1_tanh(acc[mt][0][1] + bv.y), s1_tanh(acc[mt][0][2] + bv.z),
                              s1_tanh(acc[mt][0][3] + bv.w)};
                const uint32_t o0 = Elem<H>::pack2(o[0], o[1]), o1 = Elem<H>::pack2(o[2], o[3]);
                o[0] = Elem<H>::lo(o0); o[1] = Elem<H>::hi(o0); o[2] = Elem<H>::lo(o1); o[3] = Elem<H>::hi(o1);
                const int off = s1_c4(m, c);
                const s1_u32x2_t hq = s1_lds8(lds0 + 2 * S1_PLANE + off), uq = s1_lds8(lds0 + 6 * S1_PLANE + off);
                const float hv[4] = {Elem<H>::lo(hq.x), Elem<H>::hi(hq.x), Elem<H>::lo(hq.y), Elem<H>::hi(hq.y)};
                const float uv[4] = {Elem<H>::lo(uq.x), Elem<H>::hi(uq.x), Elem<H>::lo(uq.y), Elem<H>::hi(uq.y)};
                float hn[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) hn[r] = (1.f - uv[r]) * o[r] + uv[r] * hv[r];
                const uint32_t n0 = Elem<H>::pack2(hn[0], hn[1]), n1 = Elem<H>::pack2(hn[2], hn[3]);
                s1_st8(lds0 + 2 * S1_PLANE + off, n0, n1);
                *reinterpret_cast<uint2*>(og + (ptm0 + m) * S1_C + c) = make_uint2(o0, o1);
                *reinterpret_cast<uint2*>(hs_tm + (ptm0 + m) * S1_C + c) = make_uint2(n0, n1);
                *reinterpret_cast<uint2*>(hs + (pg0 + m) * S1_C + c) = make_uint2(n0, n1);
            }
        }
        // next frame's x tile (every wave is past its last read of the x planes: the last step's barrier)
        if (more_t) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (x_glob[j] >= 0) *reinterpret_cast<uint4*>(smem + x_lds[j]) = xq[j];
        }
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // drain the zero-fill DMAs before LDS is released
}

// ------------------------------------------------------------------------------------------------------------------------
// backward (see cgru_scan_bwd_kernel for the recursion).  16-bit planes: 0,1 = dg2; 2,3 = dg1 reset part; 4,5 = dg1 update
// part.  Float planes [40][64]: carry (gradient into the previous hidden state), dx2.
// stream per frame: conv A (w2t IHWO [128][9][64]: K = 64 per tap, one pair) 9 tiles, conv B (w1t IHWO [128][9][128]: pairs
// = reset, update) 18 tiles.  Output rows of both: waves 0, 1 = rows 0..63, waves 2, 3 = rows 64..127, i.e.
//   conv A: waves 0, 1 -> d(r h), waves 2, 3 -> dx_2;      conv B: waves 0, 1 -> dx_1, waves 2, 3 -> dh_c.
// ------------------------------------------------------------------------------------------------------------------------
template <typename H>
__global__ __launch_bounds__(S1_NT) void cgru_scan1_bwd_kernel(const int B, const int T, const H* __restrict__ dhs_tm,
                                                               const H* __restrict__ ru, const H* __restrict__ og,
                                                               const H* __restrict__ hs_tm, const H* __restrict__ h0,
                                                               const H* __restrict__ w1t, const H* __restrict__ w2t,
                                                               H* __restrict__ dg1_all, H* __restrict__ dg2_all,
                                                               H* __restrict__ dxs_tm, H* __restrict__ dh0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KA = 9 * 64, KB = 9 * 128;
    constexpr int FPL = S1_PIX * S1_C * 4;                        // a float plane
    const uint32_t lds0 = lds_addr_of(smem), ldsF = lds0 + 6 * S1_PLANE, ldsB = ldsF + 2 * FPL;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    for (int i = tid; i < (6 * S1_PLANE + 2 * FPL) / 16; i += S1_NT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

    const eve_int4 rs_a = make_rsrc_words(w2t, 128 * KA * 2);
    const eve_int4 rs_b = make_rsrc_words(w1t, 128 * KB * 2);
    S1Lane L;
    L.init(li, lg);
    int brow[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int c = wave * 32 + nt * 16 + li;
        brow[nt] = (c << 6) + ((lg ^ (((c >> 2) & 1) << 1)) << 4);
    }
    int b_relA[2], b_relB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = tid + S1_NT * j, row = q >> 2, sw = (((q & 3) ^ (((row >> 2) & 1) << 1)) << 4);
        b_relA[j] = (row * KA) * 2 + sw;
        b_relB[j] = (row * KB) * 2 + sw;
    }
    // tile of stream position pos (0..26 within a frame): conv A taps 0..8, then conv B (pair, tap); mirrored taps
    auto issue = [&](int pos, int slot, bool live) {
        const bool isA = pos < 9;
        const int q = isA ? pos : pos - 9, pair = isA ? 0 : q / 9, tap = q - 9 * pair;
        const uint32_t dst = ldsB + slot * S1_SLOT + wave * 1024;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int koff = ((8 - tap) * (isA ? 64 : 128) + (2 * pair + s2) * 32) * 2;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                lds_dma16_asm(isA ? rs_a : rs_b, dst + s2 * S1_SUB + j * 4096, live ? (isA ? b_relA[j] : b_relB[j]) + koff : EVE_OOB);
        }
    };
    auto unpack4 = [](const uint2 q, float* f) {
        f[0] = Elem<H>::lo(q.x); f[1] = Elem<H>::hi(q.x); f[2] = Elem<H>::lo(q.y); f[3] = Elem<H>::hi(q.y);
    };
    bool pok[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) pok[mt] = mt * 16 + li < S1_PIX;

    issue(0, 0, true); issue(1, 1, true); issue(2, 2, true); issue(3, 3, true);
    __syncthreads();

    uint32_t gs = 0;
    for (int t = T - 1; t >= 0; --t) {
        const bool more_t = t > 0;
        const size_t ptm0 = ((size_t)t * B + b) * S1_PIX;
        const size_t prev0 = ((size_t)(t - 1) * B + b) * S1_PIX;
        // ---- phase A (element-wise, every thread): dg2, the update-gate half of dg1, the direct part of the carry ----
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = tid + S1_NT * i;                        // 640 groups of 4 channels
            if (e >= S1_PIX * 16) continue;
            const int m = e >> 4, c = (e & 15) * 4;
            float d[4], u[4], o[4], hp[4] = {0.f, 0.f, 0.f, 0.f};
            unpack4(*reinterpret_cast<const uint2*>(dhs_tm + (ptm0 + m) * S1_C + c), d);
            unpack4(*reinterpret_cast<const uint2*>(ru + (ptm0 + m) * 128 + 64 + c), u);
            unpack4(*reinterpret_cast<const uint2*>(og + (ptm0 + m) * S1_C + c), o);
            if (t > 0) unpack4(*reinterpret_cast<const uint2*>(hs_tm + (prev0 + m) * S1_C + c), hp);
            else if (h0) unpack4(*reinterpret_cast<const uint2*>(h0 + ((size_t)b * S1_PIX + m) * S1_C + c), hp);
            const uint32_t cf = ldsF + (m * S1_C + c) * 4;
            float g2[4], g1u[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dhn = d[r] + s1_ldsf(cf + 4 * r);
                g2[r] = dhn * (1.f - u[r]) * (1.f - o[r] * o[r]);
                g1u[r] = dhn * (hp[r] - o[r]) * u[r] * (1.f - u[r]);
                s1_stf(cf + 4 * r, dhn * u[r]);
            }
            const uint2 p2 = make_uint2(Elem<H>::pack2(g2[0], g2[1]), Elem<H>::pack2(g2[2], g2[3]));
            const uint2 p1 = make_uint2(Elem<H>::pack2(g1u[0], g1u[1]), Elem<H>::pack2(g1u[2], g1u[3]));
            s1_st8(lds0 + s1_c4(m, c), p2.x, p2.y);
            s1_st8(lds0 + 4 * S1_PLANE + s1_c4(m, c), p1.x, p1.y);
            *reinterpret_cast<uint2*>(dg2_all + (ptm0 + m) * S1_C + c) = p2;
            *reinterpret_cast<uint2*>(dg1_all + (ptm0 + m) * 128 + 64 + c) = p1;
        }
        // the filter tiles of this frame's first two steps (issued four / three steps ago, or in the prologue)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
        f32x4_t acc[3][2];
        s1_frag_t fxA[2][3], fwA[2][2], fxB[2][3], fwB[2][2];
        auto load = [&](int pl0, int tap, uint32_t gpos, s1_frag_t (&fx)[2][3], s1_frag_t (&fw)[2][2]) {
            const int dy = tap / 3, dx = tap - 3 * dy;
            const uint32_t la = lds0 + pl0 * S1_PLANE + (dy * 10 + dx) * 64, lb = ldsB + (gpos & 3) * S1_SLOT;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) fx[j][mt] = s1_lds16(la + j * S1_PLANE + L.abase[dy & 1][mt]);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) fw[j][nt] = s1_lds16(lb + j * S1_SUB + brow[nt]);
            }
        };
        // ================= conv A: d[r h | x] = conv(dg2, W2^T) =================
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        load(0, 0, gs, fxA, fwA);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            issue(q + 4, (int)((gs + q + 4) & 3), true);
            if (q < 8) {
                if (q & 1) load(0, q + 1, gs + q + 1, fxA, fwA);
                else       load(0, q + 1, gs + q + 1, fxB, fwB);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (q & 1) s1_mma<H, 2>(acc, fxB, fwB);
            else       s1_mma<H, 2>(acc, fxA, fwA);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(8)" ::: "memory");
            __syncthreads();
        }
        gs += 9;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int row = wave * 32 + nt * 16 + lg * 4, c = row & 63;
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                if (!pok[mt]) continue;
                const int m = mt * 16 + li;
                const uint32_t cf = ldsF + (wave < 2 ? 0 : FPL) + (m * S1_C + c) * 4;
                if (wave < 2) {
                    // ---- d(r h) -> reset-gate half of dg1, its part of the carry ----
                    float rr[4], hp[4] = {0.f, 0.f, 0.f, 0.f}, g1r[4];
                    unpack4(*reinterpret_cast<const uint2*>(ru + (ptm0 + m) * 128 + c), rr);
                    if (t > 0) unpack4(*reinterpret_cast<const uint2*>(hs_tm + (prev0 + m) * S1_C + c), hp);
                    else if (h0) unpack4(*reinterpret_cast<const uint2*>(h0 + ((size_t)b * S1_PIX + m) * S1_C + c), hp);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float drh = acc[mt][nt][r];
                        g1r[r] = drh * hp[r] * rr[r] * (1.f - rr[r]);
                        s1_stf(cf + 4 * r, s1_ldsf(cf + 4 * r) + drh * rr[r]);
                    }
                    const uint2 p1 = make_uint2(Elem<H>::pack2(g1r[0], g1r[1]), Elem<H>::pack2(g1r[2], g1r[3]));
                    s1_st8(lds0 + 2 * S1_PLANE + s1_c4(m, c), p1.x, p1.y);
                    *reinterpret_cast<uint2*>(dg1_all + (ptm0 + m) * 128 + c) = p1;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s1_stf(cf + 4 * r, acc[mt][nt][r]);     // dx_2
                }
            }
        }
        __syncthreads();                                          // dg1 complete in LDS for conv B
        // ================= conv B: d[x | h] = conv(dg1, W1^T) =================
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        load(2, 0, gs, fxA, fwA);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            const int np = 9 + q + 4;
            issue(np >= 27 ? np - 27 : np, (int)((gs + q + 4) & 3), np >= 27 ? more_t : true);
            if (q < 17) {
                const int qn = q + 1, pl = 2 + 2 * (qn / 9), tp = qn % 9;
                if (q & 1) load(pl, tp, gs + q + 1, fxA, fwA);
                else       load(pl, tp, gs + q + 1, fxB, fwB);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (q & 1) s1_mma<H, 2>(acc, fxB, fwB);
            else       s1_mma<H, 2>(acc, fxA, fwA);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(8)" ::: "memory");
            __syncthreads();
        }
        gs += 18;
        // ---- d x = dx_1 + dx_2 (waves 0, 1);  dh_c joins the carry (waves 2, 3) ----
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int row = wave * 32 + nt * 16 + lg * 4, c = row & 63;
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                if (!pok[mt]) continue;
                const int m = mt * 16 + li;
                if (wave < 2) {
                    const uint32_t xf = ldsF + FPL + (m * S1_C + c) * 4;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = s1_ldsf(xf + 4 * r) + acc[mt][nt][r];
                    *reinterpret_cast<uint2*>(dxs_tm + (ptm0 + m) * S1_C + c) =
                        make_uint2(Elem<H>::pack2(v[0], v[1]), Elem<H>::pack2(v[2], v[3]));
                } else {
                    const uint32_t cf = ldsF + (m * S1_C + c) * 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) s1_stf(cf + 4 * r, s1_ldsf(cf + 4 * r) + acc[mt][nt][r]);
                }
            }
        }
        __syncthreads();                                          // the carry is complete before the next frame's phase A
    }
    if (dh0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = tid + S1_NT * i;
            if (e >= S1_PIX * 16) continue;
            const int m = e >> 4, c = (e & 15) * 4;
            const uint32_t cf = ldsF + (m * S1_C + c) * 4;
            *reinterpret_cast<uint2*>(dh0 + ((size_t)b * S1_PIX + m) * S1_C + c) =
                make_uint2(Elem<H>::pack2(s1_ldsf(cf), s1_ldsf(cf + 4)), Elem<H>::pack2(s1_ldsf(cf + 8), s1_ldsf(cf + 12)));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

constexpr size_t S1_LDS_FWD = (size_t)8 * S1_PLANE + 4 * S1_SLOT;
constexpr size_t S1_LDS_BWD = (size_t)6 * S1_PLANE + 2 * (S1_PIX * S1_C * 4) + 4 * S1_SLOT;

}  // namespace eve

using namespace eve;

// one-sequence-per-workgroup instantiations of eve_cgru_scan_fwd / _bwd (dispatched from cgru_scan.hip)
int eve_cgru_scan1_fwd(int dtype, int B, int T, const void* xs, const void* h0, const void* w1, const float* b1, const void* w2,
                       const float* b2, void* hs, void* hs_tm, void* ru, void* rh, void* og, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)cgru_scan1_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S1_LDS_FWD);
        (void)hipFuncSetAttribute((const void*)cgru_scan1_fwd_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S1_LDS_FWD);
        attr_set = true;
    }
    EVE_DISPATCH_H16(dtype, EVE_LAUNCH(EVE_HNAME(H, "cgru_scan1_fwd_kernel<", ">"), cgru_scan1_fwd_kernel<H>, dim3(B), dim3(S1_NT), S1_LDS_FWD, s, B, T,
                                       (const H*)xs, (const H*)h0, (const H*)w1, b1, (const H*)w2, b2, (H*)hs, (H*)hs_tm, (H*)ru, (H*)rh, (H*)og));
    EVE_CHECK_LAUNCH();
    return 0;
}

int eve_cgru_scan1_bwd(int dtype, int B, int T, const void* dhs_tm, const void* ru, const void* og, const void* hs_tm, const void* h0,
                       const void* w1t, const void* w2t, void* dg1_all, void* dg2_all, void* dxs_tm, void* dh0, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)cgru_scan1_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S1_LDS_BWD);
        (void)hipFuncSetAttribute((const void*)cgru_scan1_bwd_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S1_LDS_BWD);
        attr_set = true;
    }
    EVE_DISPATCH_H16(dtype, EVE_LAUNCH(EVE_HNAME(H, "cgru_scan1_bwd_kernel<", ">"), cgru_scan1_bwd_kernel<H>, dim3(B), dim3(S1_NT), S1_LDS_BWD, s, B, T,
                                       (const H*)dhs_tm, (const H*)ru, (const H*)og, (const H*)hs_tm, (const H*)h0, (const H*)w1t, (const H*)w2t,
                                       (H*)dg1_all, (H*)dg2_all, (H*)dxs_tm, (H*)dh0));
    EVE_CHECK_LAUNCH();
    return 0;
}
