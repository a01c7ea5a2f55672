// RefineNet's conv-GRU bottleneck as ONE persistent kernel over the whole clip (bf16):
//   gates_1 = conv3x3(cat[x_t, h]);  r, u = sigmoid(gates_1);  o = tanh(conv3x3(cat[r * h, x_t]));
//   h' = (1 - u) * o + u * h                         (/root/reference/src/models/common.py:388-415, CGRUCell)
// applied for t = 0 .. T-1 (refine_net.py:132-176 keeps the state across frames).  The per-step path is two conv
// launches and two gate kernels per frame on a 5x8x64 feature map -- 120 launches per clip, each far too small to
// fill the chip.  Here a workgroup owns three sequences: their hidden state lives in LDS for all T steps (written
// back to HBM every step, which is what the backward and the caller read), both gate GEMMs run on MFMA from
// halo-resident LDS tiles with the filter banks streamed through a 4-slot LDS-DMA ring that never drains (it runs
// across the conv1 -> conv2 and the t -> t+1 boundaries), and sigmoid / tanh / blend are the epilogues.
//
// Geometry is the bottleneck's: 5 x 8 pixels, 64 channels (refine_net.py:188-212).  LDS rows are one halo pixel x 32
// channels (64 B) with the chunk swizzle of the halo conv kernel (key = halo-row parity: W = 8 < 16).
#include "common.h"
#include "lds_dma.h"

namespace eve {

constexpr int CG_IMG = 3;                       // sequences per workgroup: 3 x 40 pixels = 120 of the 128-pixel MFMA tile
constexpr int CG_PIX = 40, CG_C = 64, CG_K = 9 * 128;
constexpr int CG_SLICE = 224 * 64;              // bytes per 32-channel halo plane (3 x 7 x 10 = 210 pixels, padded)
constexpr int CG_BSLOT = 128 * 64;              // one weight tile: 128 output channels x 32 k
constexpr int CG_LDS = 6 * CG_SLICE + 4 * CG_BSLOT;

typedef uint32_t cg_u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t cg_u32x4_t __attribute__((ext_vector_type(4)));

// Gate non-linearities of the 16-bit scan.  The results are rounded to 16 bits on the spot, so one-ulp float building blocks
// (v_exp_f32, v_rcp_f32) are exact enough -- and the epilogues are where this kernel's time is: per frame a wave runs 72 MFMA
// steps of ~60 instructions and two epilogues that were ~1 750 + ~3 800 instructions with tanhf and IEEE divisions expanded 64
// times each (round 5; the float32 scan of cell_scan_f32.hip keeps tanhf / the division).
__device__ __forceinline__ float cg_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
// tanh(z) = 1 - 2 / (1 + e^(2z)): saturates correctly (e^(2z) = inf -> 1, 0 -> -1); absolute error ~1e-7
__device__ __forceinline__ float cg_tanh(float z) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * z)); }

template <typename H>
__global__ __launch_bounds__(256) void cgru_scan_fwd_kernel(const int B, const int T, const H* __restrict__ xs,
                                                            const H* __restrict__ h0, const H* __restrict__ w1,
                                                            const float* __restrict__ b1, const H* __restrict__ w2,
                                                            const float* __restrict__ b2, H* __restrict__ hs,
                                                            H* __restrict__ hs_tm, H* __restrict__ ru,
                                                            H* __restrict__ rh, H* __restrict__ og) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // planes: 0,1 = x (channels 0-31, 32-63); 2,3 = h; 4,5 = r*h
    const uint32_t lds0 = lds_addr_of(smem);
    const uint32_t ldsB = lds0 + 6 * CG_SLICE;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int b0 = blockIdx.x * CG_IMG;

    for (int i = tid; i < 6 * CG_SLICE / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

    const eve_int4 rs_w1 = make_rsrc_words(w1, 128 * CG_K * 2);
    const eve_int4 rs_w2 = make_rsrc_words(w2, 64 * CG_K * 2);

    // ---- lane constants -------------------------------------------------------------------------------------
    // interior LDS byte offset (inside a plane) of the lane's pixel for each of its four 16-pixel MFMA column tiles
    int pix_lds[4], pix_glob[4], pix_tm[4];          // pix_glob: (sequence * T) * 40 + pixel, or -1; pix_tm: sequence * 40 + pixel
    int abase[2][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = wm * 64 + mt * 16 + li;
        const bool ok = m < CG_IMG * CG_PIX && b0 + m / CG_PIX < B;
        const int ti = ok ? m / CG_PIX : 0, rem = ok ? m % CG_PIX : 0;
        const int py = rem >> 3, px = rem & 7;
        const int hr0 = ti * 7 + py + 1;
        pix_lds[mt] = ((hr0 * 10 + px + 1) << 6) | ((hr0 & 1) << 16);          // row parity kept in bit 16
        pix_glob[mt] = ok ? (b0 + ti) * T * CG_PIX + rem : -1;
        pix_tm[mt] = (b0 + ti) * CG_PIX + rem;
        // fragment address of tap (dy, dx) = abase[dy & 1][mt] + (dy * 10 + dx) * 64: the chunk key is the halo row's parity
        const int hrt = ti * 7 + py;
#pragma unroll
        for (int q = 0; q < 2; ++q) abase[q][mt] = ((hrt * 10 + px) << 6) + ((lg ^ (((hrt + q) & 1) << 1)) << 4);
    }
    int brow1[4], brow2[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int c1 = wn * 64 + nt * 16 + li, c2 = nt * 16 + li;
        brow1[nt] = (c1 << 6) + ((lg ^ (((c1 >> 2) & 1) << 1)) << 4);
        brow2[nt] = (c2 << 6) + ((lg ^ (((c2 >> 2) & 1) << 1)) << 4);
    }
    int b_rel[2], b_cl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int L = tid + 256 * j;
        b_cl[j] = L >> 2;
        b_rel[j] = (b_cl[j] * CG_K) * 2 + (((L & 3) ^ (((b_cl[j] >> 2) & 1) << 1)) << 4);
    }
    // x-tile staging slots: 120 pixels x 8 chunks of 16 B
    int x_glob[4], x_lds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + 256 * j;
        const int q = e >> 3, part = e & 7;
        const int ti = q / CG_PIX, rem = q % CG_PIX, py = rem >> 3, px = rem & 7;
        const bool ok = e < CG_IMG * CG_PIX * 8 && b0 + ti < B;
        const int hr = ti * 7 + py + 1;
        x_glob[j] = ok ? ((b0 + ti) * T * CG_PIX + rem) * CG_C + part * 8 : -1;
        x_lds[j] = (part >> 2) * CG_SLICE + ((hr * 10 + px + 1) << 6) + ((((part & 3)) ^ ((hr & 1) << 1)) << 4);
    }

    // weight tile of stream position (conv, slice, tap) into ring slot
    auto issue_w = [&](int conv, int sl, int tap, int slot, bool live) {
        const int koff = (tap * 128 + sl * 32) * 2;
        const uint32_t dst = ldsB + slot * CG_BSLOT + wave * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = live && (conv == 0 || b_cl[j] < 64);
            lds_dma16_asm(conv == 0 ? rs_w1 : rs_w2, dst + j * 4096, ok ? b_rel[j] + koff : EVE_OOB);
        }
    };
    // 8 bytes = 4 channels c..c+3 of the lane's pixel mt in plane group `pl` (0 = x, 2 = h, 4 = r*h)
    auto lds_c4 = [&](int pl, int mt, int c) -> uint32_t {
        const int key = (pix_lds[mt] >> 16) & 1;
        return lds0 + (pl + (c >> 5)) * CG_SLICE + (pix_lds[mt] & 0xffff) + (((((c & 31) >> 3)) ^ (key << 1)) << 4) + (c & 7) * 2;
    };

    // ---- initial hidden state, first x tile ------------------------------------------------------------------
    __syncthreads();
    if (h0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j;
            const int q = e >> 3, part = e & 7, ti = q / CG_PIX, rem = q % CG_PIX;
            if (e < CG_IMG * CG_PIX * 8 && b0 + ti < B)
                *reinterpret_cast<uint4*>(smem + 2 * CG_SLICE + x_lds[j]) =
                    *reinterpret_cast<const uint4*>(h0 + ((size_t)(b0 + ti) * CG_PIX + rem) * CG_C + part * 8);
        }
    }
    uint4 xq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        xq[j] = x_glob[j] >= 0 ? *reinterpret_cast<const uint4*>(xs + x_glob[j]) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (x_glob[j] >= 0) *reinterpret_cast<uint4*>(smem + x_lds[j]) = xq[j];

    issue_w(0, 0, 0, 0, true);
    issue_w(0, 0, 1, 1, true);
    issue_w(0, 0, 2, 2, true);
    issue_w(0, 0, 3, 3, true);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");             // tiles 0 and 1 have landed
    __syncthreads();

    // Round 5: the step loop is software-pipelined.  One wave per SIMD means nothing else hides a step's fragment reads
    // (8 x ds_read_b128: ~250 cycles of LDS latency before the first of 16 MFMAs = 272 matrix-pipe cycles; 33 us per frame
    // where the MFMAs need ~12), so step g+1's fragments are read into a second register set under step g's MFMAs.  The
    // filter ring runs one tile further ahead for it (tile g+4 is issued in step g into the slot of tile g, whose reads
    // every wave completed before the barrier of step g-1: each step ends lgkmcnt(0), vmcnt(4), barrier) and tile g+1 is
    // published one step earlier.  The pipeline does not cross the conv1 -> conv2 and frame boundaries (the planes a
    // convolution reads are written by the epilogue before it): each convolution's first step reads its own fragments.
    uint32_t gs = 0;                                  // stream position (ring phase)
    for (int t = 0; t < T; ++t) {
        const bool more_t = t + 1 < T;
        // next frame's x tile: in flight during this frame's GEMMs
#pragma unroll
        for (int j = 0; j < 4; ++j)
            xq[j] = (more_t && x_glob[j] >= 0) ? *reinterpret_cast<const uint4*>(xs + x_glob[j] + (size_t)(t + 1) * CG_PIX * CG_C)
                                               : make_uint4(0, 0, 0, 0);
        f32x4_t acc[4][4];
        uint2 ug[4][4];                                // update gate of the wn = 1 waves, as stored (16-bit pairs)
#pragma unroll
        for (int conv = 0; conv < 2; ++conv) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const bool active = conv == 0 || wn == 1;             // conv2 has 64 output channels: one wave column
            uint4 fxA[4], fwA[4], fxB[4], fwB[4];
            // fragments of (slice sl, tap) from ring position gpos; conv1 reads cat[x, h] = planes 0,1,2,3, conv2 cat[r*h, x] = 4,5,0,1
            auto load = [&](int sl, int tap, uint32_t gpos, uint4 (&fx)[4], uint4 (&fw)[4]) {
                const int plane = conv == 0 ? sl : (sl < 2 ? 4 + sl : sl - 2);
                const uint32_t la = lds0 + plane * CG_SLICE, lb = ldsB + (gpos & 3) * CG_BSLOT;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    fx[mt] = __builtin_bit_cast(uint4, *reinterpret_cast<const EVE_LDS cg_u32x4_t*>((uintptr_t)(la + abase[(tap / 3) & 1][mt] + ((tap / 3) * 10 + tap % 3) * 64)));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    fw[nt] = __builtin_bit_cast(uint4, *reinterpret_cast<const EVE_LDS cg_u32x4_t*>(
                        (uintptr_t)(lb + (conv == 0 ? brow1[nt] : brow2[nt]))));
            };
            auto mma = [&](const uint4 (&fx)[4], const uint4 (&fw)[4]) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        Elem<H>::mfma(acc[mt][nt], fw[nt], fx[mt]);
            };
            if (active) load(0, 0, gs, fxA, fwA);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every wave is done with this tile's slot before step 0 refills it
            __syncthreads();
            for (int sl = 0; sl < 4; sl += 2, gs += 18) {
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    const int s_ = sl + i / 9, tap = i % 9;
                    // weight tile of stream position +4
                    int nsl = s_ + (tap + 4) / 9, nconv = conv, live = 1;
                    if (nsl >= 4) { nsl -= 4; nconv = conv + 1; if (nconv == 2) { nconv = 0; live = more_t; } }
                    issue_w(nconv, nsl, (tap + 4) % 9, (int)((gs + i + 4) & 3), live != 0);
                    if (active) {
                        const bool has_next = i < 17 || sl == 0;
                        if (has_next) {
                            if (i & 1) load(s_ + (tap + 1) / 9, (tap + 1) % 9, gs + i + 1, fxA, fwA);
                            else       load(s_ + (tap + 1) / 9, (tap + 1) % 9, gs + i + 1, fxB, fwB);
                        }
                        __builtin_amdgcn_sched_barrier(0);        // the reads above are issued BEFORE the MFMAs (hipcc would sink them)
                        if (i & 1) mma(fxB, fwB);
                        else       mma(fxA, fwA);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // this wave's reads of tile g+1 are complete; tile g+2 (issued 2 steps ago) has landed
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(4)" ::: "memory");
                    __syncthreads();
                }
            }
            if (conv == 0) {
                // ---- r, u = sigmoid(gates_1 + b);  waves wn = 0 hold r (channels 0..63), wn = 1 hold u ----
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int c = nt * 16 + lg * 4;                       // channel inside the 64-wide half
                    const float4 bv = *reinterpret_cast<const float4*>(b1 + wn * 64 + c);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        float v[4] = {cg_sigmoid(acc[mt][nt][0] + bv.x), cg_sigmoid(acc[mt][nt][1] + bv.y),
                                      cg_sigmoid(acc[mt][nt][2] + bv.z), cg_sigmoid(acc[mt][nt][3] + bv.w)};
                        // the stored (bf16) gate is the one every later stage sees
                        const uint32_t p0 = Elem<H>::pack2(v[0], v[1]), p1 = Elem<H>::pack2(v[2], v[3]);
                        v[0] = Elem<H>::lo(p0); v[1] = Elem<H>::hi(p0);
                        v[2] = Elem<H>::lo(p1); v[3] = Elem<H>::hi(p1);
                        const int pg = pix_glob[mt];
                        const size_t ptm = (size_t)t * B * CG_PIX + pix_tm[mt];           // time-major: what the backward walks
                        if (pg >= 0)
                            *reinterpret_cast<uint2*>(ru + ptm * 128 + wn * 64 + c) = make_uint2(p0, p1);
                        if (wn == 0) {
                            const cg_u32x2_t hq = *reinterpret_cast<const EVE_LDS cg_u32x2_t*>((uintptr_t)lds_c4(2, mt, c));
                            const uint32_t q0 = Elem<H>::pack2(v[0] * Elem<H>::lo(hq.x), v[1] * Elem<H>::hi(hq.x));
                            const uint32_t q1 = Elem<H>::pack2(v[2] * Elem<H>::lo(hq.y), v[3] * Elem<H>::hi(hq.y));
                            if (pg >= 0) {
                                *reinterpret_cast<EVE_LDS cg_u32x2_t*>((uintptr_t)lds_c4(4, mt, c)) = cg_u32x2_t{q0, q1};
                                *reinterpret_cast<uint2*>(rh + ptm * CG_C + c) = make_uint2(q0, q1);
                            }
                        } else {
                            ug[mt][nt] = make_uint2(p0, p1);
                        }
                    }
                }
                __syncthreads();                                  // r*h visible to conv2
            }
        }
        // ---- o = tanh(gate_2 + b);  h' = (1 - u) o + u h   (the wn = 1 waves own u and o for the same lanes) ----
        if (wn == 1) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int c = nt * 16 + lg * 4;
                const float4 bv = *reinterpret_cast<const float4*>(b2 + c);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int pg = pix_glob[mt];
                    if (pg < 0) continue;
                    float o[4] = {cg_tanh(acc[mt][nt][0] + bv.x), cg_tanh(acc[mt][nt][1] + bv.y), cg_tanh(acc[mt][nt][2] + bv.z),
                                  cg_tanh(acc[mt][nt][3] + bv.w)};
                    const uint32_t o0 = Elem<H>::pack2(o[0], o[1]), o1 = Elem<H>::pack2(o[2], o[3]);
                    o[0] = Elem<H>::lo(o0); o[1] = Elem<H>::hi(o0);
                    o[2] = Elem<H>::lo(o1); o[3] = Elem<H>::hi(o1);
                    const uint32_t ha = lds_c4(2, mt, c);
                    const cg_u32x2_t hq = *reinterpret_cast<const EVE_LDS cg_u32x2_t*>((uintptr_t)ha);
                    const float hv[4] = {Elem<H>::lo(hq.x), Elem<H>::hi(hq.x),
                                         Elem<H>::lo(hq.y), Elem<H>::hi(hq.y)};
                    const float uv[4] = {Elem<H>::lo(ug[mt][nt].x), Elem<H>::hi(ug[mt][nt].x), Elem<H>::lo(ug[mt][nt].y), Elem<H>::hi(ug[mt][nt].y)};
                    float hn[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) hn[r] = (1.f - uv[r]) * o[r] + uv[r] * hv[r];
                    const uint32_t n0 = Elem<H>::pack2(hn[0], hn[1]), n1 = Elem<H>::pack2(hn[2], hn[3]);
                    *reinterpret_cast<EVE_LDS cg_u32x2_t*>((uintptr_t)ha) = cg_u32x2_t{n0, n1};
                    const size_t go = ((size_t)pg + (size_t)t * CG_PIX) * CG_C + c;
                    const size_t gtm = ((size_t)t * B * CG_PIX + pix_tm[mt]) * CG_C + c;
                    *reinterpret_cast<uint2*>(og + gtm) = make_uint2(o0, o1);
                    *reinterpret_cast<uint2*>(hs + go) = make_uint2(n0, n1);
                    *reinterpret_cast<uint2*>(hs_tm + gtm) = make_uint2(n0, n1);
                }
            }
        }
        // next frame's x tile (every wave is past its last read of the x planes: the step barrier above)
        if (more_t) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x_glob[j] >= 0) *reinterpret_cast<uint4*>(smem + x_lds[j]) = xq[j];
        }
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // drain the zero-fill DMAs before LDS is released
}


// =================================================================================================
// Backward of the scan above, also ONE persistent launch: frames are walked last to first, the gradient carried into
// the previous hidden state stays in registers (float) for the whole clip, and per frame
//     dhn = d hs[t] + carry
//     dg2 = dhn (1 - u)(1 - o^2)                      d(pre-tanh)          -> LDS planes 0,1 (bf16), dg2_all[t]
//     du  = dhn (h_prev - o);  dg1[64..] = du u (1 - u)                    -> LDS planes 4,5,        dg1_all[t][64..]
//     dcat2 = conv3x3(dg2, W2^T)      = [d(r h) | dx_2]                    (MFMA, filter bank of gate_2 as IHWO)
//     dg1[..64] = d(r h) h_prev r (1 - r)                                  -> LDS planes 2,3,        dg1_all[t][..64]
//     dcat1 = conv3x3(dg1, W1^T)      = [dx_1 | dh_c]
//     carry = dhn u + d(r h) r + dh_c;      dxs[t] = dx_1 + dx_2
// (common.py:400-415 differentiated; the per-frame path was 2 conv launches + 2 gate kernels + 3 adds per frame = ~240
// launches of 10-50 us per clip batch).  The data gradient of a 3x3 / pad 1 convolution is the same convolution with the
// IHWO filter bank and the taps mirrored (tap 8 - t), so both GEMMs reuse the forward's halo tiles, ring and fragment
// addresses.  Output-channel rows are dealt so that the wn = 1 waves own everything that feeds the carry (d(r h), dh_c)
// and the wn = 0 waves both halves of d x: no exchange between waves.  The weight and bias gradients are batched over
// all T*B frames afterwards from dg1_all / dg2_all (eve_conv2d_wgrad), as before.
// =================================================================================================
template <typename H>
__global__ __launch_bounds__(256) void cgru_scan_bwd_kernel(const int B, const int T, const H* __restrict__ dhs_tm,
                                                            const H* __restrict__ ru, const H* __restrict__ og,
                                                            const H* __restrict__ hs_tm, const H* __restrict__ h0,
                                                            const H* __restrict__ w1t, const H* __restrict__ w2t,
                                                            H* __restrict__ dg1_all, H* __restrict__ dg2_all,
                                                            H* __restrict__ dxs_tm, H* __restrict__ dh0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // planes: 0,1 = dg2 (64 channels); 2..5 = dg1 (128 channels: reset part, update part)
    const uint32_t lds0 = lds_addr_of(smem);
    const uint32_t ldsB = lds0 + 6 * CG_SLICE;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int b0 = blockIdx.x * CG_IMG;

    for (int i = tid; i < 6 * CG_SLICE / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

    constexpr int KA = 9 * 64, KB = 9 * 128;                   // row lengths of the IHWO banks (gate_2: 64 outputs, gates_1: 128)
    const eve_int4 rs_a = make_rsrc_words(w2t, 128 * KA * 2);
    const eve_int4 rs_b = make_rsrc_words(w1t, 128 * KB * 2);

    int pix_lds[4], pix_tm[4];
    bool pix_ok[4];
    int abase[2][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = wm * 64 + mt * 16 + li;
        const bool ok = m < CG_IMG * CG_PIX && b0 + m / CG_PIX < B;
        const int ti = ok ? m / CG_PIX : 0, rem = ok ? m % CG_PIX : 0;
        const int py = rem >> 3, px = rem & 7;
        const int hr0 = ti * 7 + py + 1;
        pix_ok[mt] = ok;
        pix_lds[mt] = ((hr0 * 10 + px + 1) << 6) | ((hr0 & 1) << 16);
        pix_tm[mt] = (b0 + ti) * CG_PIX + rem;
        const int hrt = ti * 7 + py;
#pragma unroll
        for (int q = 0; q < 2; ++q) abase[q][mt] = ((hrt * 10 + px) << 6) + ((lg ^ (((hrt + q) & 1) << 1)) << 4);
    }
    int brow[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int c1 = wn * 64 + nt * 16 + li;
        brow[nt] = (c1 << 6) + ((lg ^ (((c1 >> 2) & 1) << 1)) << 4);
    }
    // weight DMA rows: LDS row cl holds input channel (= output row of the data gradient) cl of cat1 for conv B, and
    // (cl + 64) & 127 of cat2 = [r h | x] for conv A, so that wn = 0 gets d x and wn = 1 gets d(r h)
    int b_relA[2], b_relB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int L = tid + 256 * j;
        const int cl = L >> 2, sw = (((L & 3) ^ (((cl >> 2) & 1) << 1)) << 4);
        b_relA[j] = (((cl + 64) & 127) * KA) * 2 + sw;
        b_relB[j] = (cl * KB) * 2 + sw;
    }
    // stream position -> (conv, slice, tap): 18 tiles of conv A (2 slices of dg2), then 36 of conv B (4 slices of dg1)
    auto issue_pos = [&](int pos, int slot, bool live) {
        const bool isA = pos < 18;
        const int q = isA ? pos : pos - 18;
        const int sl = q / 9, tap = q - sl * 9;
        const int koff = ((8 - tap) * (isA ? 64 : 128) + sl * 32) * 2;          // mirrored tap of the IHWO bank
        const uint32_t dst = ldsB + slot * CG_BSLOT + wave * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            lds_dma16_asm(isA ? rs_a : rs_b, dst + j * 4096, live ? (isA ? b_relA[j] : b_relB[j]) + koff : EVE_OOB);
    };
    auto lds_c4 = [&](int pl, int mt, int c) -> uint32_t {
        const int key = (pix_lds[mt] >> 16) & 1;
        return lds0 + (pl + (c >> 5)) * CG_SLICE + (pix_lds[mt] & 0xffff) + (((((c & 31) >> 3)) ^ (key << 1)) << 4) + (c & 7) * 2;
    };
    auto unpack4 = [](const uint2 q, float* f) {
        f[0] = Elem<H>::lo(q.x); f[1] = Elem<H>::hi(q.x);
        f[2] = Elem<H>::lo(q.y); f[3] = Elem<H>::hi(q.y);
    };

    // one register array, two uses that never meet in a wave: the carry of the wn = 1 waves (across frames), dx_2 of the wn = 0
    // waves (from conv A to the end of the frame)
    f32x4_t cy[4][4];
    f32x4_t (&dxk)[4][4] = cy;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) cy[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    issue_pos(0, 0, true);
    issue_pos(1, 1, true);
    issue_pos(2, 2, true);
    issue_pos(3, 3, true);
    __syncthreads();

    uint32_t gs = 0;                                          // stream position (ring phase)
    for (int t = T - 1; t >= 0; --t) {
        const bool more_t = t > 0;
        const size_t frame = (size_t)t * B * CG_PIX;
        // ---- phase A (the waves that own the carry): dg2, the update-gate half of dg1, dh_a ----
        if (wn == 1) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int c = nt * 16 + lg * 4;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    if (!pix_ok[mt]) continue;
                    const size_t pt = frame + pix_tm[mt];
                    float d[4], u[4], o[4], hp[4] = {0.f, 0.f, 0.f, 0.f};
                    unpack4(*reinterpret_cast<const uint2*>(dhs_tm + pt * CG_C + c), d);
                    unpack4(*reinterpret_cast<const uint2*>(ru + pt * 128 + 64 + c), u);
                    unpack4(*reinterpret_cast<const uint2*>(og + pt * CG_C + c), o);
                    if (t > 0) unpack4(*reinterpret_cast<const uint2*>(hs_tm + (pt - (size_t)B * CG_PIX) * CG_C + c), hp);
                    else if (h0) unpack4(*reinterpret_cast<const uint2*>(h0 + (size_t)pix_tm[mt] * CG_C + c), hp);
                    float g2[4], g1u[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dhn = d[r] + cy[mt][nt][r];
                        g2[r] = dhn * (1.f - u[r]) * (1.f - o[r] * o[r]);
                        g1u[r] = dhn * (hp[r] - o[r]) * u[r] * (1.f - u[r]);
                        cy[mt][nt][r] = dhn * u[r];
                    }
                    const uint2 p2 = make_uint2(Elem<H>::pack2(g2[0], g2[1]), Elem<H>::pack2(g2[2], g2[3]));
                    const uint2 p1 = make_uint2(Elem<H>::pack2(g1u[0], g1u[1]), Elem<H>::pack2(g1u[2], g1u[3]));
                    *reinterpret_cast<EVE_LDS cg_u32x2_t*>((uintptr_t)lds_c4(0, mt, c)) = cg_u32x2_t{p2.x, p2.y};
                    *reinterpret_cast<EVE_LDS cg_u32x2_t*>((uintptr_t)lds_c4(4, mt, c)) = cg_u32x2_t{p1.x, p1.y};
                    *reinterpret_cast<uint2*>(dg2_all + pt * CG_C + c) = p2;
                    *reinterpret_cast<uint2*>(dg1_all + pt * 128 + 64 + c) = p1;
                }
            }
        }
        // the weight tiles of this frame's first TWO steps (issued four / three steps ago, or in the prologue): loads return
        // in order, so "at most the two younger tiles outstanding" covers them for every wave, whatever it stored above
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __syncthreads();
        f32x4_t acc[4][4];
#pragma unroll
        for (int conv = 0; conv < 2; ++conv) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const int nsl = conv == 0 ? 2 : 4, base = conv == 0 ? 0 : 18, pl0 = conv == 0 ? 0 : 2;
            // software-pipelined step loop (see cgru_scan_fwd_kernel): step g+1's fragments are read under step g's MFMAs, the
            // filter ring runs four tiles ahead, no pipelining across the conv A -> conv B and frame boundaries
            uint4 fxA[4], fwA[4], fxB[4], fwB[4];
            auto load = [&](int sl, int tap, uint32_t gpos, uint4 (&fx)[4], uint4 (&fw)[4]) {
                const uint32_t la = lds0 + (pl0 + sl) * CG_SLICE, lb = ldsB + (gpos & 3) * CG_BSLOT;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    fx[mt] = __builtin_bit_cast(uint4, *reinterpret_cast<const EVE_LDS cg_u32x4_t*>((uintptr_t)(la + abase[(tap / 3) & 1][mt] + ((tap / 3) * 10 + tap % 3) * 64)));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    fw[nt] = __builtin_bit_cast(uint4, *reinterpret_cast<const EVE_LDS cg_u32x4_t*>((uintptr_t)(lb + brow[nt])));
            };
            auto mma = [&](const uint4 (&fx)[4], const uint4 (&fw)[4]) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        Elem<H>::mfma(acc[mt][nt], fw[nt], fx[mt]);
            };
            load(0, 0, gs, fxA, fwA);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every wave is done with this tile's slot before step 0 refills it
            __syncthreads();
            for (int sl = 0; sl < nsl; sl += 2, gs += 18) {
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    const int s_ = sl + i / 9, tap = i % 9;
                    int npos = base + s_ * 9 + tap + 4;       // weight tile of stream position +4
                    bool live = true;
                    if (npos >= 54) { npos -= 54; live = more_t; }
                    issue_pos(npos, (int)((gs + i + 4) & 3), live);
                    const bool has_next = i < 17 || sl + 2 < nsl;
                    if (has_next) {
                        if (i & 1) load(s_ + (tap + 1) / 9, (tap + 1) % 9, gs + i + 1, fxA, fwA);
                        else       load(s_ + (tap + 1) / 9, (tap + 1) % 9, gs + i + 1, fxB, fwB);
                    }
                    __builtin_amdgcn_sched_barrier(0);            // the reads above are issued BEFORE the MFMAs (hipcc would sink them)
                    if (i & 1) mma(fxB, fwB);
                    else       mma(fxA, fwA);
                    __builtin_amdgcn_sched_barrier(0);
                    // this wave's reads of tile g+1 are complete; tile g+2 (issued 2 steps ago) has landed
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(4)" ::: "memory");
                    __syncthreads();
                }
            }
            if (conv == 0) {
                if (wn == 1) {
                    // ---- d(r h) -> reset-gate half of dg1, dh_b ----
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int c = nt * 16 + lg * 4;
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            if (!pix_ok[mt]) continue;
                            const size_t pt = frame + pix_tm[mt];
                            float rr[4], hp[4] = {0.f, 0.f, 0.f, 0.f}, g1r[4];
                            unpack4(*reinterpret_cast<const uint2*>(ru + pt * 128 + c), rr);
                            if (t > 0) unpack4(*reinterpret_cast<const uint2*>(hs_tm + (pt - (size_t)B * CG_PIX) * CG_C + c), hp);
                            else if (h0) unpack4(*reinterpret_cast<const uint2*>(h0 + (size_t)pix_tm[mt] * CG_C + c), hp);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float drh = acc[mt][nt][r];
                                g1r[r] = drh * hp[r] * rr[r] * (1.f - rr[r]);
                                cy[mt][nt][r] += drh * rr[r];
                            }
                            const uint2 p1 = make_uint2(Elem<H>::pack2(g1r[0], g1r[1]), Elem<H>::pack2(g1r[2], g1r[3]));
                            *reinterpret_cast<EVE_LDS cg_u32x2_t*>((uintptr_t)lds_c4(2, mt, c)) = cg_u32x2_t{p1.x, p1.y};
                            *reinterpret_cast<uint2*>(dg1_all + pt * 128 + c) = p1;
                        }
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) dxk[a][b] = acc[a][b];
                }
                __syncthreads();                              // dg1 complete in LDS for conv B
            }
        }
        // ---- dh_c joins the carry; d x = dx_1 + dx_2 ----
        if (wn == 1) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) cy[a][b] += acc[a][b];
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int c = nt * 16 + lg * 4;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    if (!pix_ok[mt]) continue;
                    const f32x4_t v = dxk[mt][nt] + acc[mt][nt];
                    *reinterpret_cast<uint2*>(dxs_tm + (frame + pix_tm[mt]) * CG_C + c) =
                        make_uint2(Elem<H>::pack2(v[0], v[1]), Elem<H>::pack2(v[2], v[3]));
                }
            }
        }
        // (the step barrier of the last tap separates this frame's reads of the planes from the next frame's phase A)
    }
    if (dh0 && wn == 1) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int c = nt * 16 + lg * 4;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (!pix_ok[mt]) continue;
                *reinterpret_cast<uint2*>(dh0 + (size_t)pix_tm[mt] * CG_C + c) =
                    make_uint2(Elem<H>::pack2(cy[mt][nt][0], cy[mt][nt][1]), Elem<H>::pack2(cy[mt][nt][2], cy[mt][nt][3]));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // drain the zero-fill DMAs before LDS is released
}

}  // namespace eve

using namespace eve;

/* CGRUCell over T frames in one launch (bf16): xs [B][T][5][8][64] NHWC, h0 [B][5][8][64] or NULL (zeros),
   w1 = gates_1 weight OHWI [128][3][3][128] (input channels: x then h), w2 = gate_2 weight OHWI [64][3][3][128]
   (input channels: r*h then x), biases float.  Outputs (bf16): hs [B][T][5][8][64] = hidden states in the caller's
   (sequence, frame) order; and TIME-major [T][B][5][8][.] for the backward, which walks frames: hs_tm, ru = the two
   sigmoid gates (128 channels), rh = r * h_{t-1}, og = tanh output gate. */
// float32 instantiation (cell_scan_f32.hip): one workgroup per sequence, float state in LDS, v_mfma_f32_16x16x4_f32
int eve_cgru_scan_f32_fwd(int B, int T, const float* xs, const float* h0, const float* w1, const float* b1, const float* w2,
                          const float* b2, float* hs, float* hs_tm, float* ru, float* rh, float* og, hipStream_t s);
int eve_cgru_scan_f32_bwd(int B, int T, const float* dhs_tm, const float* ru, const float* og, const float* hs_tm, const float* h0,
                          const float* w1t, const float* w2t, float* dg1_all, float* dg2_all, float* dxs_tm, float* dh0,
                          hipStream_t s);

// one sequence per workgroup (cgru_scan1.hip): the batches this model sees (B <= g_cfg.cgru_seq_max_b sequences)
int eve_cgru_scan1_fwd(int dtype, int B, int T, const void* xs, const void* h0, const void* w1, const float* b1, const void* w2,
                       const float* b2, void* hs, void* hs_tm, void* ru, void* rh, void* og, hipStream_t s);
int eve_cgru_scan1_bwd(int dtype, int B, int T, const void* dhs_tm, const void* ru, const void* og, const void* hs_tm, const void* h0,
                       const void* w1t, const void* w2t, void* dg1_all, void* dg2_all, void* dxs_tm, void* dh0, hipStream_t s);

extern "C" int eve_cgru_scan_fwd(int dtype, int B, int T, const void* xs, const void* h0, const void* w1, const float* b1, const void* w2,
                                 const float* b2, void* hs, void* hs_tm, void* ru, void* rh, void* og, eve_stream_t stream) {
    if (dtype == EVE_DT_F32 && B > 0 && T > 0 && xs && w1 && b1 && w2 && b2 && hs && hs_tm && ru && rh && og)
        return eve_cgru_scan_f32_fwd(B, T, (const float*)xs, (const float*)h0, (const float*)w1, b1, (const float*)w2, b2, (float*)hs,
                                     (float*)hs_tm, (float*)ru, (float*)rh, (float*)og, (hipStream_t)stream);
    if ((dtype != EVE_DT_BF16 && dtype != EVE_DT_F16) || B <= 0 || T <= 0 || !xs || !w1 || !b1 || !w2 || !b2 || !hs || !hs_tm || !ru || !rh || !og)
        return set_error_msg("cgru_scan_fwd: bad arguments");
    if ((long long)B * T * CG_PIX * 128 >= (1ll << 31)) return set_error_msg("cgru_scan_fwd: clip too large for 32-bit offsets");
    if (B <= g_cfg.cgru_seq_max_b) return eve_cgru_scan1_fwd(dtype, B, T, xs, h0, w1, b1, w2, b2, hs, hs_tm, ru, rh, og, (hipStream_t)stream);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)cgru_scan_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)cgru_scan_fwd_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    EVE_DISPATCH_H16(dtype, EVE_LAUNCH(EVE_HNAME(H, "cgru_scan_fwd_kernel<", ">"), cgru_scan_fwd_kernel<H>, dim3((B + CG_IMG - 1) / CG_IMG), dim3(256), CG_LDS,
                                       (hipStream_t)stream, B, T, (const H*)xs, (const H*)h0, (const H*)w1, b1, (const H*)w2, b2, (H*)hs, (H*)hs_tm,
                                       (H*)ru, (H*)rh, (H*)og));
    EVE_CHECK_LAUNCH();
    return 0;
}


/* Backward of eve_cgru_scan_fwd in one launch (bf16).  Inputs, TIME-major [T][B][5][8][.]: dhs_tm = gradient of the hidden
   states, and the forward's ru / og / hs_tm; h0 [B][5][8][64] or NULL; w1t = gates_1 filter bank IHWO [128][3][3][128],
   w2t = gate_2 filter bank IHWO [128][3][3][64].  Outputs (time-major): dg1_all [T][B][5][8][128] and dg2_all [..][64] = the
   gradients of the two pre-activations (what the batched weight / bias gradients read), dxs_tm [..][64] = d xs, and dh0
   [B][5][8][64] (NULL = not wanted).  common.py:400-415 differentiated.                                                   */
extern "C" int eve_cgru_scan_bwd(int dtype, int B, int T, const void* dhs_tm, const void* ru, const void* og, const void* hs_tm, const void* h0,
                                 const void* w1t, const void* w2t, void* dg1_all, void* dg2_all, void* dxs_tm, void* dh0,
                                 eve_stream_t stream) {
    if (dtype == EVE_DT_F32 && B > 0 && T > 0 && dhs_tm && ru && og && hs_tm && w1t && w2t && dg1_all && dg2_all && dxs_tm)
        return eve_cgru_scan_f32_bwd(B, T, (const float*)dhs_tm, (const float*)ru, (const float*)og, (const float*)hs_tm, (const float*)h0,
                                     (const float*)w1t, (const float*)w2t, (float*)dg1_all, (float*)dg2_all, (float*)dxs_tm, (float*)dh0,
                                     (hipStream_t)stream);
    if ((dtype != EVE_DT_BF16 && dtype != EVE_DT_F16) || B <= 0 || T <= 0 || !dhs_tm || !ru || !og || !hs_tm || !w1t || !w2t || !dg1_all || !dg2_all || !dxs_tm)
        return set_error_msg("cgru_scan_bwd: bad arguments");
    if ((long long)B * T * CG_PIX * 128 >= (1ll << 31)) return set_error_msg("cgru_scan_bwd: clip too large for 32-bit offsets");
    if (B <= g_cfg.cgru_seq_max_b)
        return eve_cgru_scan1_bwd(dtype, B, T, dhs_tm, ru, og, hs_tm, h0, w1t, w2t, dg1_all, dg2_all, dxs_tm, dh0, (hipStream_t)stream);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)cgru_scan_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)cgru_scan_bwd_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    EVE_DISPATCH_H16(dtype, EVE_LAUNCH(EVE_HNAME(H, "cgru_scan_bwd_kernel<", ">"), cgru_scan_bwd_kernel<H>, dim3((B + CG_IMG - 1) / CG_IMG), dim3(256), CG_LDS,
                                       (hipStream_t)stream, B, T, (const H*)dhs_tm, (const H*)ru, (const H*)og, (const H*)hs_tm, (const H*)h0,
                                       (const H*)w1t, (const H*)w2t, (H*)dg1_all, (H*)dg2_all, (H*)dxs_tm, (H*)dh0));
    EVE_CHECK_LAUNCH();
    return 0;
}
