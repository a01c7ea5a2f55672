// ResNet stem, fused for gfx950:  conv 7x7/2 (3 -> 64)  ->  InstanceNorm (no affine)  ->  ReLU  ->  max-pool 3x3/2
// in ONE kernel that never materialises the 64x64x64 convolution output.
// Reference: torchvision ResNet._forward_impl conv1/bn1/relu/maxpool as instantiated by
// /root/reference/src/models/eye_net.py:48-50,106 (norm_layer = InstanceNorm2d, 128x128 eye patches).
//
// relu(IN(.)) is monotone per channel (rstd > 0), so the window maximum is taken on the RAW convolution values
// and only the winner is normalised.  One wavefront owns one image: it walks the 64 output rows top to bottom,
// computes each row with 112 MFMAs (4 column tiles x 4 channel tiles x 7 filter rows, K = 8 taps x 4 channels),
// keeps per-channel sum / sum of squares and the running 3x3 window maxima in registers, and writes only the
// pooled 32x32x64 tensor (raw, bf16), the window arg-max and, once the plane statistics are known, normalises
// its own pooled values in place (L2 hits).  HBM traffic per image: 146 KB in, 128 + 64 KB out, instead of
// 146 KB in + 3 x 512 KB for the unfused conv / stats / pool sequence.
//
// Input rows are staged by LDS-DMA into a per-wave 12-row ring two output rows ahead (no barriers: the ring is
// private to the wave, a counted s_waitcnt is the only synchronisation); the 28 KB filter bank is shared by the
// block's 8 waves.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "lds_dma.h"

namespace eve {


constexpr int SF_ROWB = 1280;                    // bytes staged per input row: padded pixels 1..160
constexpr int SF_RING = 12;                      // rows per wave: 7 live + 2 x 2 in flight (+1 spare)
constexpr int SF_WAVES = 8;
constexpr int SF_WBYTES = 7 * 4096;              // filter bank: 7 filter rows x 64 channels x 64 B
constexpr int SF_XROW = 136 * 8;                 // bytes per packed input row ([IW + 8] pixels x 4 channels bf16)
constexpr int SF_KBYTES = 64 * 3 * 4;               // backward: {rstd, B, C} per channel and wave
constexpr uint32_t SF_NEG = 0xff61b1e0u;         // -3.0e38 with the four key bits clear

__device__ __forceinline__ void sf_dma16(const eve_int4& rsrc, uint32_t lds, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void sf_dma4(const eve_int4& rsrc, uint32_t lds, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
typedef uint32_t sf_u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t sf_u32x2_t __attribute__((ext_vector_type(2)));
// A 16-byte fragment stays ONE vector value from the LDS load to the MFMA operand.  (Round 6: as HIP's uint4 -- a struct -- the
// load was split into two 8-byte halves by the middle end and the back end re-merged the filter fragments as ds_read2_b64: twice
// the LDS cycles of ds_read_b128 and banked differently from what the swizzle is built for.  SQ counters of the forward: 59 % of
// the LDS-array cycles were bank conflicts, the array 70 % busy, 23 % of the wave cycles stalled on LDS issue.)
typedef sf_u32x4_t sf_frag_t;
__device__ __forceinline__ sf_frag_t sf_lds_read(uint32_t addr) {
    return *reinterpret_cast<const EVE_LDS sf_u32x4_t*>((uintptr_t)addr);
}
template <typename H>
__device__ __forceinline__ void sf_mfma(f32x4_t& acc, const sf_frag_t& a, const sf_frag_t& b) {
    if constexpr (Elem<H>::IS_BF16)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    else
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}
template <int CTRL>
__device__ __forceinline__ uint32_t sf_dpp(uint32_t old, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ float sf_row_sum16(float v) {          // butterfly over the 16 lanes of a DPP row
    v += __builtin_bit_cast(float, sf_dpp<0xb1>(0u, __builtin_bit_cast(uint32_t, v)));     // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, sf_dpp<0x4e>(0u, __builtin_bit_cast(uint32_t, v)));     // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, sf_dpp<0x141>(0u, __builtin_bit_cast(uint32_t, v)));    // row_half_mirror
    v += __builtin_bit_cast(float, sf_dpp<0x140>(0u, __builtin_bit_cast(uint32_t, v)));    // row_mirror
    return v;
}
__device__ __forceinline__ float sf_fmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// Filter bank fill shared by the forward and backward kernels.  LDS row c_lds = nt*16 + 4*g + r holds output
// channel co = g*16 + nt*4 + r, so that an MFMA lane (rows 4g..4g+3 of tiles nt = 0..3) owns the 16 CONSECUTIVE
// channels g*16 .. g*16+15 of its pixel: 32-byte stores instead of four 8-byte ones.  64-byte rows, chunk-swizzled
// like the halo kernel (conflict-free ds_read_b128).
template <typename H>
__device__ __forceinline__ void sf_fill_weights(char* sW, const H* __restrict__ w8, int tid, int nthreads) {
    for (int e = tid; e < 64 * 7 * 8; e += nthreads) {
        const int kw = e & 7, kh = (e >> 3) % 7, co = e / 56;
        uint2 v = make_uint2(0u, 0u);
        if (kw < 7) v = *reinterpret_cast<const uint2*>(w8 + ((co * 7 + kh) * 7 + kw) * 8);   // channels 0..3
        const int c_lds = ((co >> 2) & 3) * 16 + (co >> 4) * 4 + (co & 3);
        const int chunk = (kw >> 1) ^ (((c_lds >> 2) & 1) << 1);
        *reinterpret_cast<uint2*>(sW + kh * 4096 + c_lds * 64 + chunk * 16 + (kw & 1) * 8) = v;
    }
}

// One output row of the convolution for the wave's image: acc[mt][nt] (mt = 2j + b holds output column
// 2*(li + 16j) + b, so a lane owns the even/odd column pair of pooled column q = li + 16j).
template <typename H, bool FIRST>
__device__ __forceinline__ void sf_conv_tap_row(f32x4_t (&acc)[4][4], uint32_t ring, int slot0, uint32_t xoff,
                                                uint32_t wbase, int kh) {
    int slot = slot0 + kh;
    slot = slot >= SF_RING ? slot - SF_RING : slot;
    const uint32_t xa = ring + slot * SF_ROWB + xoff, wa = wbase + kh * 4096;
    sf_frag_t fx[4], fw[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) fx[mt] = sf_lds_read(xa + (mt & 1) * 16 + (mt >> 1) * 512);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) fw[nt] = sf_lds_read(wa + nt * 1024);
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};             // first filter row: C = 0 is an inline MFMA operand, no zero-fill
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
        {
            if (FIRST) acc[mt][nt] = zero;
            sf_mfma<H>(acc[mt][nt], fw[nt], fx[mt]);
        }
}
// COMPACT keeps the filter-row loop rolled (one set of fragment registers) for the register-hungry backward
template <typename H, bool COMPACT>
__device__ __forceinline__ void sf_conv_row(f32x4_t (&acc)[4][4], uint32_t ring, int slot0, uint32_t xoff,
                                            uint32_t wbase) {
    sf_conv_tap_row<H, true>(acc, ring, slot0, xoff, wbase, 0);
    if (COMPACT) {
#pragma unroll 1
        for (int kh = 1; kh < 7; ++kh) sf_conv_tap_row<H, false>(acc, ring, slot0, xoff, wbase, kh);
    } else {
#pragma unroll
        for (int kh = 1; kh < 7; ++kh) sf_conv_tap_row<H, false>(acc, ring, slot0, xoff, wbase, kh);
    }
}

// stage padded input row `row` of the image at byte offset img_off into its ring slot (rows past the image: zeros)
__device__ __forceinline__ void sf_stage_row(const eve_int4& rs, uint32_t ring, int row, int rows, int img_off, int lane) {
    const int slot = row % SF_RING;
    const bool live = row < rows;
    const int soff = live ? img_off + row * SF_XROW : 0;
    sf_dma16(rs, ring + slot * SF_ROWB, live ? lane * 16 + 8 : EVE_OOB, soff);
    sf_dma4(rs, ring + slot * SF_ROWB + 1024, live ? lane * 4 + 8 + 1024 : EVE_OOB, soff);
}

template <typename H>
__global__ __launch_bounds__(64 * SF_WAVES) void stem_fwd_fused_kernel(const int N, const int IH,
                                                                       const H* __restrict__ xp, const uint32_t xp_bytes,
                                                                       const H* __restrict__ w8, const float eps,
                                                                       H* yp, uint8_t* __restrict__ idx,
                                                                       float* __restrict__ mr, const int halves) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sW = smem + SF_WAVES * SF_RING * SF_ROWB;
    float* const sX = reinterpret_cast<float*>(sW + SF_WBYTES);           // halves == 2: [wave][64 channels][sum, sum of squares]
    const int tid = threadIdx.x;
    sf_fill_weights<H>(sW, w8, tid, 64 * SF_WAVES);
    __syncthreads();

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int OH = IH / 2, PH = OH / 2, rows = IH + 6;
    const uint32_t ring = lds_addr_of(smem) + wave * (SF_RING * SF_ROWB);
    const uint32_t xoff = 16 * (2 * li + lg);
    const uint32_t wbase = lds_addr_of(sW) + li * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4);
    const eve_int4 rs = make_rsrc_words(xp, xp_bytes);
    const float inv_hw = 1.f / (float)(OH * 64);

    // images are dealt round-robin over the workgroups first (wave w of workgroup b takes image w * grid + b): a small batch
    // then puts one or two waves on every CU instead of eight waves on a fraction of them.
    // halves == 2 (fewer images than half the wave slots): waves 2k and 2k + 1 of a workgroup share an image, rows
    // [0, OH/2) and [OH/2, OH); the lower half starts one row early for its first pooling window (that row's statistics
    // belong to the upper half) and the two partial plane sums meet in LDS.  Every wave of the workgroup runs the same
    // number of turns (the exchange is a workgroup barrier); a turn without an image only keeps the barriers.
    const bool fold = (IH & 7) == 0;                              // the image has two equal halves of rows
    const int per_turn = gridDim.x * (SF_WAVES / halves);
    const int turns = (N + per_turn - 1) / per_turn;
    for (int turn = 0; turn < turns; ++turn) {
        const int n = turn * per_turn + (halves == 2 ? (wave >> 1) : wave) * (int)gridDim.x + (int)blockIdx.x;
        const bool live = n < N;
        const int hf = halves == 2 ? (wave & 1) : 0;
        const int oy_first = hf ? OH / 2 - 1 : 0;                 // first convolution row this wave computes
        const int oy_own = hf ? OH / 2 : 0;                       // first row whose statistics / pooled output are its own
        const int oy_end = (halves == 2 && !hf) ? OH / 2 : OH;
        if (!live) {                                              // (uniform per wave)
            if (halves == 2) { __syncthreads(); __syncthreads(); }
            continue;
        }
        const int img_off = n * rows * SF_XROW;
        for (int r = 0; r < 9; ++r) sf_stage_row(rs, ring, 2 * oy_first + r, rows, img_off, lane);
        float S[4][4], Q[4][4];
        uint32_t M[2][4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) { S[a][b] = 0.f; Q[a][b] = 0.f; M[0][a][b] = SF_NEG; M[1][a][b] = SF_NEG; }
        H* yimg = yp + (size_t)n * PH * 32 * 64;
        uint8_t* iimg = idx + (size_t)n * PH * 32 * 64;
        int slot0 = (2 * oy_first) % SF_RING;
        for (int oy = oy_first; oy < oy_end; ++oy) {
            // rows 2oy .. 2oy+6 were issued two iterations ago; younger: 4 DMAs + 6 stores of one of the last two rows (see
            // header); the first rows of a wave's range have issued fewer stores yet: wait for everything but the last 4 DMAs
            if (oy < oy_first + 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else                   asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            sf_stage_row(rs, ring, 2 * oy + 9, rows, img_off, lane);
            sf_stage_row(rs, ring, 2 * oy + 10, rows, img_off, lane);
            f32x4_t acc[4][4];
            sf_conv_row<H, false>(acc, ring, slot0, xoff, wbase);
            slot0 = slot0 + 2 >= SF_RING ? slot0 + 2 - SF_RING : slot0 + 2;
            // ---- plane statistics ----
            // (one wave per image: at the middle row the upper half's sums are reduced and parked in LDS, so that the plane
            //  sums are formed as (upper half) + (lower half) exactly as the two-waves-per-image form does -- an image's
            //  outputs must not depend on the batch it arrives in, bit for bit)
            if (halves == 1 && fold && oy == OH / 2) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float su = sf_row_sum16(S[nt][r]), qu = sf_row_sum16(Q[nt][r]);
                        if (li == 0) {
                            float* o = sX + ((wave * 64) + lg * 16 + nt * 4 + r) * 2;
                            o[0] = su; o[1] = qu;
                        }
                        S[nt][r] = 0.f; Q[nt][r] = 0.f;
                    }
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = 0.f, q = 0.f;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) { const float v = acc[mt][nt][r]; s += v; q += v * v; }
                    if (oy >= oy_own) { S[nt][r] += s; Q[nt][r] += q; }
                }
            // ---- 3x3/2 max-pool on keys = value bits with the low 4 mantissa bits replaced by the window position ----
            const bool odd = oy & 1;
            const uint32_t khbits = odd ? 0u : 4u;                 // filter row 2 (odd rows) or 1 (even rows) of the window
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    uint32_t carry = 0;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        // (copy the vector elements first: __builtin_bit_cast on an element lvalue reads element 0)
                        const float ef = acc[2 * j][nt][r], of = acc[2 * j + 1][nt][r];
                        // keys: (value & ~15) | window position; the window-row bits go in with the column bits
                        const uint32_t e = (__builtin_bit_cast(uint32_t, ef) & 0xfffffff0u) | (khbits | 1u);
                        const uint32_t o = (__builtin_bit_cast(uint32_t, of) & 0xfffffff0u) | khbits;
                        const uint32_t ol = (__builtin_bit_cast(uint32_t, of) & 0xfffffff0u) | (khbits | 2u);
                        // column 2q-1 = the odd column of lane li-1 (lane 0: the last lane of the previous tile / padding)
                        const uint32_t edge = j == 0 ? SF_NEG : sf_dpp<0x121>(0u, carry);          // row_ror:1
                        const uint32_t l = sf_dpp<0x111>(edge, ol);                                  // row_shr:1
                        carry = ol;
                        const float h = sf_fmax3(__builtin_bit_cast(float, l), __builtin_bit_cast(float, e),
                                                 __builtin_bit_cast(float, o));
                        const uint32_t hb = __builtin_bit_cast(uint32_t, h);
                        const float m = fmaxf(__builtin_bit_cast(float, M[j][nt][r]), h);
                        // odd row: it becomes filter row 0 of the next window (khbits is 0 here: just set bit 3)
                        M[j][nt][r] = odd ? (hb | 8u) : __builtin_bit_cast(uint32_t, m);
                        acc[2 * j][nt][r] = m;                                                      // window result (odd rows)
                    }
                }
            if (odd && oy >= oy_own) {
                const int py = oy >> 1;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const size_t o = ((size_t)py * 32 + li + 16 * j) * 64 + lg * 16;
                    uint32_t pk[8], ib[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        uint32_t code[4];
                        float val[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float kf = acc[2 * j][nt][r];
                            const uint32_t k = __builtin_bit_cast(uint32_t, kf);
                            val[r] = __builtin_bit_cast(float, k & 0xfffffff0u);
                            code[r] = (uint32_t)(0x01203450678ull >> ((k & 15u) * 4)) & 15u;       // kh*3 + kw
                        }
                        pk[2 * nt] = Elem<H>::pack2(val[0], val[1]);
                        pk[2 * nt + 1] = Elem<H>::pack2(val[2], val[3]);
                        ib[nt] = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
                    }
                    *reinterpret_cast<uint4*>(yimg + o) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    *reinterpret_cast<uint4*>(yimg + o + 8) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
                    *reinterpret_cast<uint4*>(iimg + o) = make_uint4(ib[0], ib[1], ib[2], ib[3]);
                }
            }
        }
        // ---- plane statistics -> mean / rstd of the lane's 16 channels ----
        float mean[4][4], rstd[4][4];
        if (halves == 1 && fold) {                                // (upper half, parked at the middle row) + (lower half)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* o = sX + ((wave * 64) + lg * 16 + nt * 4 + r) * 2;
                    S[nt][r] = o[0] + sf_row_sum16(S[nt][r]);
                    Q[nt][r] = o[1] + sf_row_sum16(Q[nt][r]);
                }
        }
        if (halves == 2) {                                        // the partner's partial sums (channel lg * 16 + nt * 4 + r)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    S[nt][r] = sf_row_sum16(S[nt][r]); Q[nt][r] = sf_row_sum16(Q[nt][r]);
                    if (li == 0) {
                        float* o = sX + ((wave * 64) + lg * 16 + nt * 4 + r) * 2;
                        o[0] = S[nt][r]; o[1] = Q[nt][r];
                    }
                }
            __syncthreads();
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* o = sX + (((wave ^ 1) * 64) + lg * 16 + nt * 4 + r) * 2;
                    // (the upper half's sum first in both waves: the two agree bit for bit)
                    S[nt][r] = hf ? o[0] + S[nt][r] : S[nt][r] + o[0];
                    Q[nt][r] = hf ? o[1] + Q[nt][r] : Q[nt][r] + o[1];
                }
            __syncthreads();                                      // sX may be rewritten in the next turn
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = ((halves == 2 || fold) ? S[nt][r] : sf_row_sum16(S[nt][r])) * inv_hw, q = ((halves == 2 || fold) ? Q[nt][r] : sf_row_sum16(Q[nt][r])) * inv_hw;
                const float var = fmaxf(q - s * s, 0.f);
                mean[nt][r] = s;
                rstd[nt][r] = rsqrtf(var + eps);
                if (li == 0 && hf == 0) {
                    float* m = mr + ((size_t)n * 64 + lg * 16 + nt * 4 + r) * 2;
                    m[0] = s; m[1] = rstd[nt][r];
                }
            }
        // ---- normalise the lane's own pooled values in place: y = relu((max - mean) * rstd) ----
        for (int py = oy_own >> 1; py < (oy_end >> 1); ++py)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                H* p = yimg + ((size_t)py * 32 + li + 16 * j) * 64 + lg * 16;
                float f[16];
                Elem<H>::unpack(*reinterpret_cast<const uint4*>(p), f);
                Elem<H>::unpack(*reinterpret_cast<const uint4*>(p + 8), f + 8);
#pragma unroll
                for (int c = 0; c < 16; ++c) f[c] = fmaxf((f[c] - mean[c >> 2][c & 3]) * rstd[c >> 2][c & 3], 0.f);
                *reinterpret_cast<uint4*>(p) = Elem<H>::pack(f);
                *reinterpret_cast<uint4*>(p + 8) = Elem<H>::pack(f + 8);
            }
    }
}

// =================================================================================================
// Round 4: the fused forward with TWO waves per image, 32 output channels each (wave h owns co = 16 g + 8 h + 0..7).
// With one wave per image the kernel's duration is the latency of one image (112 MFMAs + ~850 VALU per row in one
// dependent chain, matrix pipe 31 % busy) and the 15 KB input ring per wave caps a CU at 8 waves.  Statistics and the
// pooling keys are per channel, so two waves can share an image's ring (each stages one of the two new rows per conv
// row; one workgroup barrier per row) and split the channels: 16 waves per CU on the same LDS, half the work per wave.
// Per-channel arithmetic is stem_fwd_fused_kernel's (same MFMA sequence, same keys); the plane sums are accumulated over
// all rows per lane and reduced once (the one-wave kernel parks the upper half's sums at the middle row for bit-equality
// with its own row-split mode, which this kernel does not have: every batch size runs the same code).
// =================================================================================================
constexpr int SP_PAIRS = 8;                       // images per workgroup and turn

// Rendezvous of the two waves of an image through an LDS word each (monotone counters): publish mine, wait for the partner's.
// A workgroup barrier per row also re-aligns the OTHER pairs: all 16 waves then issue their fragment reads, their MFMAs and
// their pooling VALU at the same moments (SQ counters of the barrier version: 25 % of the wave cycles stalled on LDS issue,
// 44 % parked).  LDS operations of one wave are served in order, so whatever the partner issued before its publish -- its
// reads of the ring included -- is behind it.
__device__ __forceinline__ void sf_pair_sync(uint32_t my_flag, uint32_t partner_flag, int value) {
    *(volatile EVE_LDS int*)(size_t)my_flag = value;
    while (*(volatile EVE_LDS int*)(size_t)partner_flag < value) __builtin_amdgcn_s_sleep(1);
}

// Round 6: stem_fwd_fused_kernel's pooling arithmetic, bit for bit, in fewer instructions (the kernel is bound by instruction
// issue: per SIMD its MFMA, VALU and SALU issue cycles ADD UP to the measured time, profiles/r06_stem.md; 410 -> ~280 VALU per
// convolution row and wave, 0.372 -> 0.312 ms at N = 1 920 together with the fragment-read fix above) -- the four key bits hold code = 8 - (kh * 3 + kw)
// directly (row part {8, 5, 2} for kh = {0, 1, 2}, minus kw; larger code = earlier in scan order, as before), so the stored arg-max
// byte is 8 - code: one packed subtraction per four channels instead of a 64-bit table shift per value; an odd column's key is
// prepared once (kw = 2 of its own window) and handed to the right-hand neighbour as key + 2 (kw = 0); an odd row's window
// maximum h becomes the next window's top row as h + 6 (kh 2 -> 0) instead of a second masking pass and a select; even / odd
// rows are two straight code paths.  Statistics: S += v, Q = fma(v, v, Q) per value.
template <typename H>
__global__ __launch_bounds__(1024) void stem_fwd_pairs_kernel(const int N, const int IH, const H* __restrict__ xp, const uint32_t xp_bytes,
                                                              const H* __restrict__ w8, const float eps, H* yp, uint8_t* __restrict__ idx,
                                                              float* __restrict__ mr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sW = smem + SP_PAIRS * SF_RING * SF_ROWB;
    int* const sFlag = reinterpret_cast<int*>(sW + SF_WBYTES);
    const int tid = threadIdx.x;
    sf_fill_weights<H>(sW, w8, tid, 1024);
    if (tid < 16) sFlag[tid] = 0;
    __syncthreads();

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave >> 1, h = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int OH = IH / 2, PH = OH / 2, rows = IH + 6;
    const uint32_t ring = lds_addr_of(smem) + pair * (SF_RING * SF_ROWB);
    const uint32_t xoff = 16 * (2 * li + lg);
    const uint32_t wbase = lds_addr_of(sW) + (2 * h) * 1024 + li * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4);
    const eve_int4 rs = make_rsrc_words(xp, xp_bytes);
    const float inv_hw = 1.f / (float)(OH * 64);
    const int ch0 = lg * 16 + 8 * h;                      // the lane's 8 channels
    const uint32_t my_flag = lds_addr_of(sFlag) + wave * 4, partner_flag = lds_addr_of(sFlag) + (wave ^ 1) * 4;
    int tick = 0;
    // stagger the pairs (nothing inside the loop re-aligns them): a quarter of a row per step of the pair index modulo 4
    __builtin_amdgcn_s_sleep(1);
    for (int d = 0; d < (pair & 3); ++d) __builtin_amdgcn_s_sleep(12);

    const int per_turn = gridDim.x * SP_PAIRS;
    const int turns = (N + per_turn - 1) / per_turn;
    for (int turn = 0; turn < turns; ++turn) {
        const int n = turn * per_turn + pair * (int)gridDim.x + (int)blockIdx.x;
        if (n >= N) break;                                        // (uniform per pair; later turns have no image either)
        const bool live = true;
        const int nn = n;
        const int img_off = nn * rows * SF_XROW;
        for (int r = h; r < 9; r += 2) sf_stage_row(rs, ring, r, rows, img_off, lane);
        float S[2][4], Q[2][4];
        uint32_t M[2][2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) { S[a][b] = 0.f; Q[a][b] = 0.f; M[0][a][b] = SF_NEG; M[1][a][b] = SF_NEG; }
        H* yimg = yp + (size_t)nn * PH * 32 * 64;
        uint8_t* iimg = idx + (size_t)nn * PH * 32 * 64;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)yimg, 0, PH * 32 * 64 * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)iimg, 0, PH * 32 * 64, 0x00020000);
        const int lane_off = li * 64 + ch0;                       // the lane's first channel of pooled column li, in elements
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sf_pair_sync(my_flag, partner_flag, ++tick);
        int slot0 = 0;
        for (int oy = 0; oy < OH; ++oy) {
            // this wave's row of two iterations ago has landed once at most the newer operations are outstanding: one row
            // (2 DMAs) and, behind an odd row, its 4 stores (the first rows of an image were waited for above)
            if (oy >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            sf_pair_sync(my_flag, partner_flag, ++tick);
            sf_stage_row(rs, ring, 2 * oy + 9 + h, rows, img_off, lane);
            if (live) {
                f32x4_t acc[4][2];
                {
                    auto conv_frags = [&](int kh, sf_frag_t (&x4)[4], sf_frag_t (&w2)[2]) {
                        int slot = slot0 + kh;
                        slot = slot >= SF_RING ? slot - SF_RING : slot;
                        const uint32_t xa = ring + slot * SF_ROWB + xoff, wa = wbase + kh * 4096;
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) x4[mt] = sf_lds_read(xa + (mt & 1) * 16 + (mt >> 1) * 512);
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) w2[nt] = sf_lds_read(wa + nt * 1024);
                    };
                    auto conv_mfma = [&](bool first, const sf_frag_t (&x4)[4], const sf_frag_t (&w2)[2]) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) {
                                if (first) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                                sf_mfma<H>(acc[mt][nt], w2[nt], x4[mt]);
                            }
                    };
                    sf_frag_t fxa[4], fwa[2], fxb[4], fwb[2];
                    conv_frags(0, fxa, fwa);
                    conv_frags(1, fxb, fwb);
                    conv_mfma(true, fxa, fwa);
#pragma unroll
                    for (int kh = 2; kh < 6; kh += 2) {
                        conv_frags(kh, fxa, fwa);
                        conv_mfma(false, fxb, fwb);
                        conv_frags(kh + 1, fxb, fwb);
                        conv_mfma(false, fxa, fwa);
                    }
                    conv_frags(6, fxa, fwa);
                    conv_mfma(false, fxb, fwb);
                    conv_mfma(false, fxa, fwa);
                }
                // ---- plane statistics ----
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            const float v = acc[mt][nt][r];
                            S[nt][r] += v;
                            Q[nt][r] = __builtin_fmaf(v, v, Q[nt][r]);
                        }
                // ---- 3x3/2 max-pool on keys = value bits with the low 4 mantissa bits replaced by 8 - (window position) ----
                auto pool_row = [&](auto odd_tag) {
                    constexpr bool ODD = decltype(odd_tag)::value;
                    constexpr uint32_t CE = ODD ? 1u : 4u, CO = ODD ? 0u : 3u;      // even column kw = 1, odd column kw = 2 (own window)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e0 = acc[0][nt][r], o0 = acc[1][nt][r], e1 = acc[2][nt][r], o1 = acc[3][nt][r];
                            const uint32_t ke0 = (__builtin_bit_cast(uint32_t, e0) & 0xfffffff0u) | CE, ko0 = (__builtin_bit_cast(uint32_t, o0) & 0xfffffff0u) | CO;
                            const uint32_t ke1 = (__builtin_bit_cast(uint32_t, e1) & 0xfffffff0u) | CE, ko1 = (__builtin_bit_cast(uint32_t, o1) & 0xfffffff0u) | CO;
                            const uint32_t kl0 = ko0 + 2u, kl1 = ko1 + 2u;          // the same columns as kw = 0 of the windows to their right
                            // column 2q - 1 = the odd column of lane li - 1; lane 0 of the first tile: the padding (its own even key
                            // stands in: no maximum changes), of the second tile: the last lane of the first
                            const uint32_t l0 = sf_dpp<0x111>(ke0, kl0);                                   // row_shr:1
                            const uint32_t l1 = sf_dpp<0x111>(sf_dpp<0x121>(0u, kl0), kl1);                // row_ror:1, row_shr:1 over it
                            const float h0 = sf_fmax3(__builtin_bit_cast(float, l0), __builtin_bit_cast(float, ke0), __builtin_bit_cast(float, ko0));
                            const float h1 = sf_fmax3(__builtin_bit_cast(float, l1), __builtin_bit_cast(float, ke1), __builtin_bit_cast(float, ko1));
                            const float m0 = fmaxf(__builtin_bit_cast(float, M[0][nt][r]), h0), m1 = fmaxf(__builtin_bit_cast(float, M[1][nt][r]), h1);
                            if constexpr (ODD) {
                                acc[0][nt][r] = m0;                                                        // the finished windows, parked in
                                acc[2][nt][r] = m1;                                                        // the even columns' registers
                                M[0][nt][r] = __builtin_bit_cast(uint32_t, h0) + 6u;                       // this row as kh = 0 of the next ones
                                M[1][nt][r] = __builtin_bit_cast(uint32_t, h1) + 6u;
                            } else {
                                M[0][nt][r] = __builtin_bit_cast(uint32_t, m0);
                                M[1][nt][r] = __builtin_bit_cast(uint32_t, m1);
                            }
                            if (r & 1) __builtin_amdgcn_sched_barrier(0);            // (two channels' temporaries live at a time, not eight)
                        }
                    if constexpr (ODD) {
                        const int py = oy >> 1;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const size_t o = ((size_t)py * 32 + li + 16 * j) * 64 + ch0;
                            uint32_t pk[4], ib[2];
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt) {
                                const float f0 = acc[2 * j][nt][0], f1 = acc[2 * j][nt][1], f2 = acc[2 * j][nt][2], f3 = acc[2 * j][nt][3];
                                const uint32_t k0 = __builtin_bit_cast(uint32_t, f0), k1 = __builtin_bit_cast(uint32_t, f1);
                                const uint32_t k2 = __builtin_bit_cast(uint32_t, f2), k3 = __builtin_bit_cast(uint32_t, f3);
                                pk[2 * nt] = Elem<H>::pack2(__builtin_bit_cast(float, k0 & 0xfffffff0u), __builtin_bit_cast(float, k1 & 0xfffffff0u));
                                pk[2 * nt + 1] = Elem<H>::pack2(__builtin_bit_cast(float, k2 & 0xfffffff0u), __builtin_bit_cast(float, k3 & 0xfffffff0u));
                                // arg-max bytes kh * 3 + kw = 8 - code, four channels at once
                                ib[nt] = 0x08080808u - ((k0 & 15u) | ((k1 & 15u) << 8) | ((k2 & 15u) << 16) | ((k3 & 15u) << 24));
                            }
                            // (buffer stores: the image's base in scalar registers, one 32-bit lane offset, the pooled row and
                            //  column tile as the scalar offset -- the per-lane 64-bit addresses of the plain stores were spilled)
                            (void)o;
                            const int so = (py * 32 + 16 * j) * 64;
                            sf_u32x4_t pv = {pk[0], pk[1], pk[2], pk[3]};
                            sf_u32x2_t iv = {ib[0], ib[1]};
                            __builtin_amdgcn_raw_buffer_store_b128(pv, ry, lane_off * 2, so * 2, 0);
                            __builtin_amdgcn_raw_buffer_store_b64(iv, ri, lane_off, so, 0);
                        }
                    }
                };
                if (oy & 1) pool_row(std::true_type{});
                else pool_row(std::false_type{});
            }
            slot0 = slot0 + 2 >= SF_RING ? slot0 + 2 - SF_RING : slot0 + 2;
        }
        if (live) {
            // ---- plane statistics -> mean / rstd of the lane's 8 channels; normalise the wave's own pooled values in place ----
            float mean[2][4], rstd[2][4];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = sf_row_sum16(S[nt][r]) * inv_hw, q = sf_row_sum16(Q[nt][r]) * inv_hw;
                    const float var = fmaxf(q - s * s, 0.f);
                    mean[nt][r] = s;
                    rstd[nt][r] = rsqrtf(var + eps);
                    if (li == 0) {
                        float* m = mr + ((size_t)nn * 64 + ch0 + nt * 4 + r) * 2;
                        m[0] = s; m[1] = rstd[nt][r];
                    }
                }
#pragma unroll 4
            for (int py = 0; py < PH; ++py)                       // (four pooled rows of loads in flight)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int so = (py * 32 + 16 * j) * 64 * 2;
                    float f[8];
                    Elem<H>::unpack(__builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ry, lane_off * 2, so, 0)), f);
#pragma unroll
                    for (int c = 0; c < 8; ++c) f[c] = fmaxf((f[c] - mean[c >> 2][c & 3]) * rstd[c >> 2][c & 3], 0.f);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sf_u32x4_t, Elem<H>::pack(f)), ry, lane_off * 2, so, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sf_pair_sync(my_flag, partner_flag, ++tick);              // the ring is rewritten by the next turn
    }
}

// =================================================================================================
// Backward of the fused stem up to the convolution output:  d(conv1 out) from d(pooled output).
// The convolution output was never stored, so the wave recomputes it row by row exactly as the forward did
// (same MFMA sequence, bit-identical values) and applies, per pixel,
//     g  = sum over the (at most 4) pooling windows that contain the pixel, selected it as arg-max (idx) and
//          survived the ReLU (y > 0) of d(pooled)
//     dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),   xhat = (x - mean) * rstd
// with mean(g) = sum d*[y>0] / HW and mean(g*xhat) = sum d*y / HW taken over the POOLED tensors first
// (y = xhat at the arg-max wherever y > 0).  Folded: dx = rstd*g + C - x*B,  B = rstd^2 * mean(g xhat),
// C = mean*B - rstd*mean(g).
// =================================================================================================
struct SfPooledRow {               // one pooled row as the lane sees it: columns q = li + 16j, its 16 channels
    uint32_t eg[2][8];             // d(pooled) where y > 0, else 0 (packed bf16 pairs)
    uint32_t code[2][4];           // arg-max window positions, one byte per channel
};

template <typename H>
__device__ __forceinline__ uint32_t sf_add_pairs(uint32_t a, uint32_t b) {      // two packed 16-bit sums
    return Elem<H>::pack2(Elem<H>::lo(a) + Elem<H>::lo(b), Elem<H>::hi(a) + Elem<H>::hi(b));
}
template <typename H>
__device__ __forceinline__ void sf_load_pooled(SfPooledRow& P, const H* __restrict__ dyp, const H* __restrict__ dyp2,
                                               const H* __restrict__ yp,
                                               const uint8_t* __restrict__ idx, size_t row_base, int li, int lg, bool live) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const size_t o = (row_base + li + 16 * j) * 64 + lg * 16;
        uint4 d0 = make_uint4(0, 0, 0, 0), d1 = d0, y0 = d0, y1 = d0, c = d0;
        if (live) {
            d0 = *reinterpret_cast<const uint4*>(dyp + o); d1 = *reinterpret_cast<const uint4*>(dyp + o + 8);
            y0 = *reinterpret_cast<const uint4*>(yp + o);  y1 = *reinterpret_cast<const uint4*>(yp + o + 8);
            c = *reinterpret_cast<const uint4*>(idx + o);
            if (dyp2) {                                            // gradient delivered as two summands
                const uint4 e0 = *reinterpret_cast<const uint4*>(dyp2 + o), e1 = *reinterpret_cast<const uint4*>(dyp2 + o + 8);
                d0 = make_uint4(sf_add_pairs<H>(d0.x, e0.x), sf_add_pairs<H>(d0.y, e0.y), sf_add_pairs<H>(d0.z, e0.z), sf_add_pairs<H>(d0.w, e0.w));
                d1 = make_uint4(sf_add_pairs<H>(d1.x, e1.x), sf_add_pairs<H>(d1.y, e1.y), sf_add_pairs<H>(d1.z, e1.z), sf_add_pairs<H>(d1.w, e1.w));
            }
        }
        const uint32_t dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        const uint32_t yy[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t m = ((int)(yy[k] << 16) > 0 ? 0xffffu : 0u) | ((int)(yy[k] & 0xffff0000u) > 0 ? 0xffff0000u : 0u);
            P.eg[j][k] = dd[k] & m;
        }
        P.code[j][0] = c.x; P.code[j][1] = c.y; P.code[j][2] = c.z; P.code[j][3] = c.w;
    }
}
// d(pooled) of channel r (0..3) of a 4-channel chunk if the window's arg-max code equals K, else 0
template <typename H>
__device__ __forceinline__ float sf_pick(const uint32_t (&eg)[2], uint32_t code, int r, uint32_t K) {
    const uint32_t w = eg[r >> 1];
    const float t = (r & 1) ? Elem<H>::hi(w) : Elem<H>::lo(w);
    const int sh = 8 * r;
    return (code & (0xffu << sh)) == (K << sh) ? t : 0.f;
}
// channels 4nt..4nt+3 of pooled column q = li + 16j, and of column q + 1 (lane li+1; lane 15 takes lane 0 of
// the next tile, or nothing past the last column)
__device__ __forceinline__ void sf_chunk(const SfPooledRow& P, int j, int nt, uint32_t (&eg)[2], uint32_t& cd,
                                         uint32_t (&neg)[2], uint32_t& ncd) {
    eg[0] = P.eg[j][2 * nt]; eg[1] = P.eg[j][2 * nt + 1]; cd = P.code[j][nt];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t edge = j == 0 ? sf_dpp<0x12f>(0u, P.eg[1][2 * nt + k]) : 0u;    // row_ror:15 = rotate left by one
        neg[k] = sf_dpp<0x101>(edge, eg[k]);                                            // row_shl:1
    }
    const uint32_t edge = j == 0 ? sf_dpp<0x12f>(0u, P.code[1][nt]) : 0u;
    ncd = sf_dpp<0x101>(edge, cd);
}

// (fallback / test reference since round 4: four-wave workgroups, one wave per SIMD -- with eight the 256-register budget spilled)
constexpr int SD_WAVES = 4;
template <typename H>
__global__ __launch_bounds__(64 * SD_WAVES) void stem_bwd_dx_kernel(const int N, const int IH,
                                                                    const H* __restrict__ xp, const uint32_t xp_bytes,
                                                                    const H* __restrict__ w8, const float* __restrict__ mr,
                                                                    const H* __restrict__ dyp, const H* __restrict__ dyp2,
                                                                    const H* __restrict__ yp,
                                                                    const uint8_t* __restrict__ idx, H* __restrict__ dx,
                                                                    const int halves) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sW = smem + SD_WAVES * SF_RING * SF_ROWB;
    const int tid = threadIdx.x;
    sf_fill_weights<H>(sW, w8, tid, 64 * SD_WAVES);
    __syncthreads();

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int OH = IH / 2, PH = OH / 2, rows = IH + 6;
    const uint32_t ring = lds_addr_of(smem) + wave * (SF_RING * SF_ROWB);
    const uint32_t xoff = 16 * (2 * li + lg);
    const uint32_t wbase = lds_addr_of(sW) + li * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4);
    const eve_int4 rs = make_rsrc_words(xp, xp_bytes);
    const float inv_hw = 1.f / (float)(OH * 64);
    char* const sK = sW + SF_WBYTES + wave * SF_KBYTES;         // this wave's per-channel constants

    // images are dealt round-robin over the workgroups first (wave w of workgroup b takes image w * grid + b): a small batch
    // then puts one or two waves on every CU instead of eight waves on a fraction of them
    // halves == 2 (small batches: fewer images than wave slots): an image is TWO work items, the upper and the lower half of its
    // rows.  The halves share nothing but the plane sums of phase A, which are taken over the small pooled tensors and simply
    // computed by both; with one wave per image the kernel's duration is the latency of one image whatever the batch.
    for (int item = wave * gridDim.x + blockIdx.x; item < N * halves; item += gridDim.x * SD_WAVES) {
        const int n = halves == 2 ? item >> 1 : item;
        const int py0 = halves == 2 ? (item & 1) * (PH / 2) : 0, py1 = halves == 2 ? py0 + PH / 2 : PH;
        const int img_off = n * rows * SF_XROW;
        for (int r = 0; r < 9; ++r) sf_stage_row(rs, ring, 4 * py0 + r, rows, img_off, lane);
        const size_t pool_base = (size_t)n * PH * 32;
        // ---- phase A: the two plane sums, over the pooled tensors; the folded per-channel constants go to LDS ----
        {
            float s1[16], s2[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
            for (int py = 0; py < PH; ++py)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const size_t o = (pool_base + (size_t)py * 32 + li + 16 * j) * 64 + lg * 16;
                    float d[16], y[16];
                    Elem<H>::unpack(*reinterpret_cast<const uint4*>(dyp + o), d);
                    Elem<H>::unpack(*reinterpret_cast<const uint4*>(dyp + o + 8), d + 8);
                    if (dyp2) {                                   // same rounding of the sum as sf_load_pooled
                        float d2[16];
                        Elem<H>::unpack(*reinterpret_cast<const uint4*>(dyp2 + o), d2);
                        Elem<H>::unpack(*reinterpret_cast<const uint4*>(dyp2 + o + 8), d2 + 8);
#pragma unroll
                        for (int c = 0; c < 16; ++c) d[c] = Elem<H>::round(d[c] + d2[c]);
                    }
                    Elem<H>::unpack(*reinterpret_cast<const uint4*>(yp + o), y);
                    Elem<H>::unpack(*reinterpret_cast<const uint4*>(yp + o + 8), y + 8);
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const float g = y[c] > 0.f ? d[c] : 0.f;
                        s1[c] += g; s2[c] += g * y[c];
                    }
                }
            const float* m = mr + ((size_t)n * 64 + lg * 16) * 2;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float a = sf_row_sum16(s1[c]) * inv_hw, b = sf_row_sum16(s2[c]) * inv_hw;
                const float mean = m[2 * c], r = m[2 * c + 1];
                const float B = r * r * b, C = mean * B - r * a;
                if (li == 0) {                                   // [lg][nt][{rstd, B, C}][r]
                    float* kc = reinterpret_cast<float*>(sK + ((lg * 4 + (c >> 2)) * 3) * 16) + (c & 3);
                    kc[0] = r; kc[4] = B; kc[8] = C;
                }
            }
        }
        // ---- phase B: recompute the convolution row by row, emit d(conv out) ----
        SfPooledRow P0, P1;
        sf_load_pooled<H>(P0, dyp, dyp2, yp, idx, pool_base + (size_t)py0 * 32, li, lg, true);
        H* dimg = dx + (size_t)n * OH * 64 * 64;
        int slot0 = (4 * py0) % SF_RING;
        for (int py = py0; py < py1; ++py) {
            sf_load_pooled<H>(P1, dyp, dyp2, yp, idx, pool_base + (size_t)(py + 1) * 32, li, lg, py + 1 < PH);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int oy = 2 * py + half;
                // rows 2oy..2oy+6 were issued two iterations ago; younger: >= 4 DMAs + 16 stores (+ pooled loads)
                if (oy == 2 * py0)          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (oy == 2 * py0 + 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else                        asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                sf_stage_row(rs, ring, 2 * oy + 9, rows, img_off, lane);
                sf_stage_row(rs, ring, 2 * oy + 10, rows, img_off, lane);
                f32x4_t acc[4][4];
                sf_conv_row<H, true>(acc, ring, slot0, xoff, wbase);
                slot0 = slot0 + 2 >= SF_RING ? slot0 + 2 - SF_RING : slot0 + 2;
                const uint32_t k0 = half ? 6u : 3u;             // window row of this conv row inside window py
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    uint32_t pe[8], po[8];                       // even / odd column of pooled column q, packed pairs
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {             // four channels at a time keeps the live set small
                        const f32x4_t kr = *reinterpret_cast<const f32x4_t*>(sK + ((lg * 4 + nt) * 3) * 16);
                        const f32x4_t kB = *reinterpret_cast<const f32x4_t*>(sK + ((lg * 4 + nt) * 3 + 1) * 16);
                        const f32x4_t kC = *reinterpret_cast<const f32x4_t*>(sK + ((lg * 4 + nt) * 3 + 2) * 16);
                        uint32_t eg0[2], cd0, ng0[2], nc0, eg1[2], cd1, ng1[2], nc1;
                        sf_chunk(P0, j, nt, eg0, cd0, ng0, nc0);
                        if (half) sf_chunk(P1, j, nt, eg1, cd1, ng1, nc1);
                        float de[4], dd[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // even column 2q: centre column (kw = 1) of window q only
                            float ge = sf_pick<H>(eg0, cd0, r, k0 + 1u);
                            // odd column 2q+1: right column (kw = 2) of window q, left column (kw = 0) of window q+1
                            float go = sf_pick<H>(eg0, cd0, r, k0 + 2u) + sf_pick<H>(ng0, nc0, r, k0);
                            if (half) {                          // odd conv row: also the top row (kh = 0) of window py+1
                                ge += sf_pick<H>(eg1, cd1, r, 1u);
                                go += sf_pick<H>(eg1, cd1, r, 2u) + sf_pick<H>(ng1, nc1, r, 0u);
                            }
                            const float xe = acc[2 * j][nt][r], xo = acc[2 * j + 1][nt][r];
                            de[r] = fmaf(-xe, kB[r], fmaf(kr[r], ge, kC[r]));
                            dd[r] = fmaf(-xo, kB[r], fmaf(kr[r], go, kC[r]));
                        }
                        pe[2 * nt] = Elem<H>::pack2(de[0], de[1]); pe[2 * nt + 1] = Elem<H>::pack2(de[2], de[3]);
                        po[2 * nt] = Elem<H>::pack2(dd[0], dd[1]); po[2 * nt + 1] = Elem<H>::pack2(dd[2], dd[3]);
                    }
                    H* o = dimg + ((size_t)oy * 64 + 2 * (li + 16 * j)) * 64 + lg * 16;
                    *reinterpret_cast<uint4*>(o) = make_uint4(pe[0], pe[1], pe[2], pe[3]);
                    *reinterpret_cast<uint4*>(o + 8) = make_uint4(pe[4], pe[5], pe[6], pe[7]);
                    *reinterpret_cast<uint4*>(o + 64) = make_uint4(po[0], po[1], po[2], po[3]);
                    *reinterpret_cast<uint4*>(o + 72) = make_uint4(po[4], po[5], po[6], po[7]);
                }
            }
            P0 = P1;
        }
    }
}

// =================================================================================================
// Round 4: the stem's backward AND its weight gradient in one kernel -- d(conv1 out) never reaches HBM.
//
// stem_bwd_dx_kernel wrote d(conv1 out) [N][64][64][64] (1 GB at N = 1 920) for wgrad_tr_kernel to read back; the
// stem has no data gradient, so that tensor's only reader was the weight gradient.  Here every conv row's d(conv out)
// goes from the accumulator registers to a 4 KB LDS tile of the wave and is multiplied, on the spot, with the input
// rows that are in the LDS ring anyway:
//     dW[co][kh][kw][c] += sum over the row's 64 pixels x of  dconv[x][co] * patch[2 oy + kh][2 x + kw][c]
// = 56 MFMAs per row and wave (K = pixels: both operands come out of ds_read_b64_tr_b16, exactly as in wgrad_tr_kernel),
// next to the 56 that recompute the row.  What makes the accumulators fit: an image is shared by TWO waves, 32 output
// channels each (wave h owns co = 16 g + 8 h + 0..7): 112 registers of dW + 32 of the recomputed row per lane instead
// of 224 + 64.  The pair shares the input ring (each wave stages one of the two new rows per conv row) and meets at one
// s_barrier per row; statistics, pooling routes and dW are per channel, so the waves exchange nothing else.  The arithmetic
// per channel is stem_bwd_dx_kernel's (same MFMA sequence, same rounding of d(conv out) to the storage format before
// it is multiplied), so dW equals the two-kernel path up to the float summation order.
// Output: dw [64][7][8][4] float, accumulated (the layout eve_stem_wgrad writes: column kw = 7 and channel 3 do not exist).
// =================================================================================================
constexpr int SB_PAIRS = 2;                       // images per workgroup and turn; two workgroups of 4 waves per CU (the launch names say 2)
constexpr int SB_DROW = 64;                       // bytes per pixel of the d(conv out) tile: 32 local channels
constexpr int SB_DTILE = 64 * SB_DROW;            // one conv row of one wave
constexpr int SB_KBYTES = 4 * 2 * 3 * 16;         // [lg][ntl][{rstd, B, C}][r] floats per wave

struct SbPooledRow {               // one pooled row, the lane's 8 channels: columns q = li + 16 j
    uint32_t eg[2][4];
    uint32_t code[2][2];
};
// one pooled row of stem_grad_prep_kernel's output (the summed, ReLU-masked gradient) and of the arg-max codes
template <typename H>
__device__ __forceinline__ void sb_load_pooled(SbPooledRow& P, const H* __restrict__ eg, const uint8_t* __restrict__ idx, size_t row_base,
                                               int li, int ch0, bool live) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const size_t o = (row_base + li + 16 * j) * 64 + ch0;
        uint4 d0 = make_uint4(0, 0, 0, 0);
        uint2 c = make_uint2(0, 0);
        if (live) {
            d0 = *reinterpret_cast<const uint4*>(eg + o);
            c = *reinterpret_cast<const uint2*>(idx + o);
        }
        P.eg[j][0] = d0.x; P.eg[j][1] = d0.y; P.eg[j][2] = d0.z; P.eg[j][3] = d0.w;
        P.code[j][0] = c.x; P.code[j][1] = c.y;
    }
}
__device__ __forceinline__ void sb_chunk(const SbPooledRow& P, int j, int ntl, uint32_t (&eg)[2], uint32_t& cd,
                                         uint32_t (&neg)[2], uint32_t& ncd) {
    eg[0] = P.eg[j][2 * ntl]; eg[1] = P.eg[j][2 * ntl + 1]; cd = P.code[j][ntl];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t edge = j == 0 ? sf_dpp<0x12f>(0u, P.eg[1][2 * ntl + k]) : 0u;   // row_ror:15 = rotate left by one
        neg[k] = sf_dpp<0x101>(edge, eg[k]);                                            // row_shl:1
    }
    const uint32_t edge = j == 0 ? sf_dpp<0x12f>(0u, P.code[1][ntl]) : 0u;
    ncd = sf_dpp<0x101>(edge, cd);
}
__device__ __forceinline__ uint2 sb_tr_read(uint32_t lds_byte_addr) {
    EVE_LDS char* base = (EVE_LDS char*)(size_t)lds_byte_addr;
    bf16x4v_t r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((EVE_LDS bf16x4v_t*)base);
    return __builtin_bit_cast(uint2, r);
}

// The two plane sums of the stem's backward as a streaming pass of their own (round 5).  Inside stem_bwd_wgrad_kernel they are
// "phase A": every wave pair reads its image's three pooled tensors once for the sums and again, row by row, for the routing,
// with 2 048 images in flight (the second read comes from HBM) and nothing else running on the pair meanwhile.  Here one
// workgroup per image reads d(pooled) (+ its second summand) and y once, writes  eg = (d + d2) where y > 0, else 0  (what the
// routing needs) and the folded constants {rstd, B, C} per (image, channel); the fused kernel then reads eg and the arg-max codes
// only.  Thread (column q, channel group cg) walks the pooled rows: a wave's load covers 8 pixels x 128 B = 1 KB of whole lines.
// The sums over the 32 columns: xor-butterfly over the wave's 8, then the four waves' partials in a fixed order through LDS
// (reproducible; not phase A's order -- the constants differ from the one-launch form's in the last float bit).
template <typename H>
__global__ __launch_bounds__(256) void stem_grad_prep_kernel(const int N, const int PH, const H* __restrict__ dyp, const H* __restrict__ dyp2,
                                                             const H* __restrict__ yp, const float* __restrict__ mr,
                                                             H* __restrict__ eg, float* __restrict__ kout) {
    __shared__ float part[4][2][64];
    const int tid = threadIdx.x, n = blockIdx.x;
    const int cg = tid & 7, q = tid >> 3;
    const size_t pool_base = (size_t)n * PH * 32;
    const float inv_hw = 1.f / (float)(PH * 2 * 64);
    float s1[8], s2[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
#pragma unroll 4
    for (int py = 0; py < PH; ++py) {
        const size_t o = (pool_base + (size_t)py * 32 + q) * 64 + cg * 8;
        uint4 d0 = *reinterpret_cast<const uint4*>(dyp + o);
        const uint4 y0 = *reinterpret_cast<const uint4*>(yp + o);
        if (dyp2) {                                       // same rounding of the sum as sb_load_pooled
            const uint4 e0 = *reinterpret_cast<const uint4*>(dyp2 + o);
            d0 = make_uint4(sf_add_pairs<H>(d0.x, e0.x), sf_add_pairs<H>(d0.y, e0.y), sf_add_pairs<H>(d0.z, e0.z), sf_add_pairs<H>(d0.w, e0.w));
        }
        const uint32_t dd[4] = {d0.x, d0.y, d0.z, d0.w}, yy[4] = {y0.x, y0.y, y0.z, y0.w};
        uint32_t gg[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t m = ((int)(yy[k] << 16) > 0 ? 0xffffu : 0u) | ((int)(yy[k] & 0xffff0000u) > 0 ? 0xffff0000u : 0u);
            gg[k] = dd[k] & m;
        }
        const uint4 g4 = make_uint4(gg[0], gg[1], gg[2], gg[3]);
        *reinterpret_cast<uint4*>(eg + o) = g4;
        float g[8], y[8];
        Elem<H>::unpack(g4, g);
        Elem<H>::unpack(y0, y);
#pragma unroll
        for (int c = 0; c < 8; ++c) { s1[c] += g[c]; s2[c] += g[c] * y[c]; }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) { s1[c] += __shfl_xor(s1[c], m); s2[c] += __shfl_xor(s2[c], m); }
    if ((tid & 63) < 8) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { part[tid >> 6][0][cg * 8 + c] = s1[c]; part[tid >> 6][1][cg * 8 + c] = s2[c]; }
    }
    __syncthreads();
    if (tid < 64) {
        const float a = (((part[0][0][tid] + part[1][0][tid]) + part[2][0][tid]) + part[3][0][tid]) * inv_hw;
        const float b = (((part[0][1][tid] + part[1][1][tid]) + part[2][1][tid]) + part[3][1][tid]) * inv_hw;
        const float mean = mr[((size_t)n * 64 + tid) * 2], r = mr[((size_t)n * 64 + tid) * 2 + 1];
        const float B = r * r * b, C = mean * B - r * a;
        float* kc = kout + ((size_t)n * 64 + tid) * 3;
        kc[0] = r; kc[1] = B; kc[2] = C;
    }
}

// dyp = stem_grad_prep_kernel's eg (the summed, masked gradient), mr = its constants [N][64][{rstd, B, C}].  (Rounds 4-5 also had
// a one-launch form that read the three pooled tensors twice and formed the plane sums itself -- "phase A"; it spilled 84-100
// bytes per lane and was only reachable without scratch.  Round 6: the scratch is required, that form is gone.)
template <typename H, int PAIRS>
__global__ __launch_bounds__(128 * PAIRS, 2) void stem_bwd_wgrad_kernel(const int N, const int IH, const H* __restrict__ xp, const uint32_t xp_bytes,
                                                             const H* __restrict__ w8, const float* __restrict__ mr,
                                                             const H* __restrict__ dyp, const uint8_t* __restrict__ idx,
                                                             float* __restrict__ dw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 128 * PAIRS;
    char* const sW = smem + PAIRS * SF_RING * SF_ROWB;
    char* const sKall = sW + SF_WBYTES;
    char* const sDall = sKall + 2 * PAIRS * SB_KBYTES;
    const int tid = threadIdx.x;
    sf_fill_weights<H>(sW, w8, tid, NT);
    __syncthreads();

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave >> 1, h = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int OH = IH / 2, PH = OH / 2, rows = IH + 6;
    const uint32_t ring = lds_addr_of(smem) + pair * (SF_RING * SF_ROWB);
    const uint32_t xoff = 16 * (2 * li + lg);
    // this wave's two weight tiles: LDS rows nt * 16 + 4 g + r with nt = 2 h, 2 h + 1
    const uint32_t wbase = lds_addr_of(sW) + (2 * h) * 1024 + li * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4);
    const eve_int4 rs = make_rsrc_words(xp, xp_bytes);
    const float inv_hw = 1.f / (float)(OH * 64);
    char* const sK = sKall + wave * SB_KBYTES;
    const uint32_t sD = lds_addr_of(sDall) + wave * SB_DTILE;
    const int ch0 = lg * 16 + 8 * h;                      // first of the lane's 8 channels in the pooled tensors

    // transposing-read lane constants (see wgrad_tr_kernel): row 8 g + t / 4 (+ 4 for the second read), 8-byte piece t % 4
    const int trow = 8 * lg + (li >> 2), tq = li & 3;

    f32x4_t dacc[2][14];                                   // dW[co tile][kh * 2 + tap tile]: rows 4 g + r, column t
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 14; ++b) dacc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int per_turn = gridDim.x * PAIRS;
    const int turns = (N + per_turn - 1) / per_turn;
    for (int turn = 0; turn < turns; ++turn) {
        const int n = turn * per_turn + pair * (int)gridDim.x + (int)blockIdx.x;
        const bool live = n < N;                                  // (uniform per wave pair; idle pairs keep the barriers)
        const int nn = live ? n : 0;
        const int img_off = nn * rows * SF_XROW;
        const size_t pool_base = (size_t)nn * PH * 32;
        // rows 0 .. 8 of the image: wave h stages the rows of its parity
        for (int r = h; r < 9; r += 2) sf_stage_row(rs, ring, r, live ? rows : 0, img_off, lane);
        // ---- the folded constants {rstd, B, C} of the lane's channels, from the prep pass: channel ch0 + li ----
        if (live && li < 8) {
            const float* kg = mr + ((size_t)nn * 64 + ch0 + li) * 3;
            float* kc = reinterpret_cast<float*>(sK + ((lg * 2 + (li >> 2)) * 3) * 16) + (li & 3);
            kc[0] = kg[0]; kc[4] = kg[1]; kc[8] = kg[2];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the nine rows (and the constants' loads)
        __syncthreads();                                          // ... and the partner's (sK is this wave's own)
        // ---- phase B: recompute the convolution row by row; d(conv out) -> LDS tile -> weight-gradient MFMAs ----
        SbPooledRow P0, P1;
        sb_load_pooled<H>(P0, dyp, idx, pool_base, li, ch0, live);
        int slot0 = 0;
        for (int py = 0; py < PH; ++py) {
            sb_load_pooled<H>(P1, dyp, idx, pool_base + (size_t)(py + 1) * 32, li, ch0, live && py + 1 < PH);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int oy = 2 * py + half;
                // this wave's row of two iterations ago has landed once at most the newer operations are outstanding:
                // half 0: one row (2 DMAs) + the 4 pooled loads just issued; half 1: one row
                // (the last pooled row issues no pooled loads: one row only there too)
                if (oy >= 2) {
                    if (half == 0 && py + 1 < PH) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                }
                // ... and the partner's.  (A rendezvous of just the two waves through LDS flags, which lets the pairs drift apart so
                //  that the two waves of a SIMD are not in the same phase, was built and measured: with its extra registers the
                //  kernel spilled inside this loop and ran 1.0 ms against 0.76.)
                __builtin_amdgcn_s_barrier();
                sf_stage_row(rs, ring, 2 * oy + 9 + h, live ? rows : 0, img_off, lane);
                if (live) {
                    // -- the row's convolution for this wave's 32 channels (acc[mt][ntl]).  Filter rows in pairs, the second row's
                    //    fragments requested before the first row's MFMAs (a full double buffer over all seven rows spills: the
                    //    112 weight-gradient accumulators leave ~90 registers for everything else) --
                    f32x4_t acc[4][2];
                    auto conv_frags = [&](int kh, sf_frag_t (&x4)[4], sf_frag_t (&w2)[2]) {
                        int slot = slot0 + kh;
                        slot = slot >= SF_RING ? slot - SF_RING : slot;
                        const uint32_t xa = ring + slot * SF_ROWB + xoff, wa = wbase + kh * 4096;
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) x4[mt] = sf_lds_read(xa + (mt & 1) * 16 + (mt >> 1) * 512);
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) w2[nt] = sf_lds_read(wa + nt * 1024);
                    };
                    auto conv_mfma = [&](bool first, const sf_frag_t (&x4)[4], const sf_frag_t (&w2)[2]) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) {
                                if (first) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                                sf_mfma<H>(acc[mt][nt], w2[nt], x4[mt]);
                            }
                    };
                    {
                        sf_frag_t fxa[4], fwa[2], fxb[4], fwb[2];
                        conv_frags(0, fxa, fwa);
                        conv_frags(1, fxb, fwb);
                        conv_mfma(true, fxa, fwa);
#pragma unroll 1
                        for (int kh = 2; kh < 6; kh += 2) {
                            conv_frags(kh, fxa, fwa);
                            conv_mfma(false, fxb, fwb);
                            conv_frags(kh + 1, fxb, fwb);
                            conv_mfma(false, fxa, fwa);
                        }
                        conv_frags(6, fxa, fwa);
                        conv_mfma(false, fxb, fwb);
                        conv_mfma(false, fxa, fwa);
                    }
                    // the first patch fragments of the weight-gradient loop do not depend on d(conv out): requested here, they
                    // arrive under the gradient routing's VALU work instead of in front of the first MFMA
                    auto patch_frag = [&](int q) {               // q = (kc * 7 + kh) * 2 + tt
                        const int tt = q & 1, kh = (q >> 1) % 7, kc = (q >> 1) / 7;
                        int slot = slot0 + kh;
                        slot = slot >= SF_RING ? slot - SF_RING : slot;
                        // patch row of pixel x starts at byte 16 x of the staged input row: taps (kw, c) contiguous
                        const uint32_t xb = ring + slot * SF_ROWB + (kc * 32 + trow) * 16 + tq * 8 + tt * 32;
                        const uint2 b0 = sb_tr_read(xb), b1 = sb_tr_read(xb + 4 * 16);
                        return make_uint4(b0.x, b0.y, b1.x, b1.y);
                    };
                    constexpr int AHEAD = 3;
                    uint4 fb[AHEAD + 1];
#pragma unroll
                    for (int q = 0; q < AHEAD; ++q) fb[q] = patch_frag(q);
                    __builtin_amdgcn_sched_barrier(0);
                    // -- d(conv out) of the lane's 4 pixel columns x 8 channels -> the wave's [pixel][channel] tile --
                    const uint32_t k0 = half ? 6u : 3u;         // window row of this conv row inside window py
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const f32x4_t kr = *reinterpret_cast<const f32x4_t*>(sK + ((lg * 2 + nt) * 3) * 16);
                            const f32x4_t kB = *reinterpret_cast<const f32x4_t*>(sK + ((lg * 2 + nt) * 3 + 1) * 16);
                            const f32x4_t kC = *reinterpret_cast<const f32x4_t*>(sK + ((lg * 2 + nt) * 3 + 2) * 16);
                            uint32_t eg0[2], cd0, ng0[2], nc0, eg1[2], cd1, ng1[2], nc1;
                            sb_chunk(P0, j, nt, eg0, cd0, ng0, nc0);
                            if (half) sb_chunk(P1, j, nt, eg1, cd1, ng1, nc1);
                            float de[4], dd[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float ge = sf_pick<H>(eg0, cd0, r, k0 + 1u);
                                float go = sf_pick<H>(eg0, cd0, r, k0 + 2u) + sf_pick<H>(ng0, nc0, r, k0);
                                if (half) {
                                    ge += sf_pick<H>(eg1, cd1, r, 1u);
                                    go += sf_pick<H>(eg1, cd1, r, 2u) + sf_pick<H>(ng1, nc1, r, 0u);
                                }
                                const float xe = acc[2 * j][nt][r], xo = acc[2 * j + 1][nt][r];
                                de[r] = fmaf(-xe, kB[r], fmaf(kr[r], ge, kC[r]));
                                dd[r] = fmaf(-xo, kB[r], fmaf(kr[r], go, kC[r]));
                            }
                            // pixel x = 2 (li + 16 j) (+ 1), local channels nt * 16 + 4 lg + 0..3: one 8-byte slot, swizzled by
                            // the pixel row so that the 16 lanes of a store / the 32 of a transposing read spread over the banks
                            const int xe_row = 2 * (li + 16 * j), slot8 = nt * 4 + lg;
                            typedef uint32_t sb_u32x2_t __attribute__((ext_vector_type(2)));
                            EVE_LDS sb_u32x2_t* pe = (EVE_LDS sb_u32x2_t*)(size_t)(sD + xe_row * SB_DROW + ((slot8 ^ ((xe_row >> 1) & 7)) << 3));
                            EVE_LDS sb_u32x2_t* po = (EVE_LDS sb_u32x2_t*)(size_t)(sD + (xe_row + 1) * SB_DROW + ((slot8 ^ (((xe_row + 1) >> 1) & 7)) << 3));
                            *pe = sb_u32x2_t{Elem<H>::pack2(de[0], de[1]), Elem<H>::pack2(de[2], de[3])};
                            *po = sb_u32x2_t{Elem<H>::pack2(dd[0], dd[1]), Elem<H>::pack2(dd[2], dd[3])};
                        }
                    }
                    // -- dW += dconv^T x patches: K = the row's 64 pixels in two chunks of 32; 28 patch fragments, each used by two
                    //    MFMAs, requested three fragments (six MFMAs) ahead of their use --
                    uint4 fa[2][2];
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) {
                            const int r0 = kc * 32 + trow, r1 = r0 + 4;
                            const uint2 a0 = sb_tr_read(sD + r0 * SB_DROW + (((ct * 4 + tq) ^ ((r0 >> 1) & 7)) << 3));
                            const uint2 a1 = sb_tr_read(sD + r1 * SB_DROW + (((ct * 4 + tq) ^ ((r1 >> 1) & 7)) << 3));
                            fa[kc][ct] = make_uint4(a0.x, a0.y, a1.x, a1.y);
                        }
#pragma unroll
                    for (int q = 0; q < 28; ++q) {
                        if (q + AHEAD < 28) fb[(q + AHEAD) % (AHEAD + 1)] = patch_frag(q + AHEAD);
                        const int tt = q & 1, kh = (q >> 1) % 7, kc = (q >> 1) / 7;
                        Elem<H>::mfma(dacc[0][kh * 2 + tt], fa[kc][0], fb[q % (AHEAD + 1)]);
                        Elem<H>::mfma(dacc[1][kh * 2 + tt], fa[kc][1], fb[q % (AHEAD + 1)]);
                    }
                }
                slot0 = slot0 + 2 >= SF_RING ? slot0 + 2 - SF_RING : slot0 + 2;
            }
            P0 = P1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the zero-fill DMAs past the image's last rows
        __syncthreads();                                          // the ring is rewritten by the next turn
    }
    // ---- the workgroup's dW: the four pairs' slices summed in LDS (the ring is free now), then one atomic per element ----
    float* const sR = reinterpret_cast<float*>(smem);            // [64 co][224 taps] (over the ring and, with two pairs, the filters)
    static_assert((size_t)PAIRS * SF_RING * SF_ROWB + SF_WBYTES >= 64 * 224 * 4, "the dW reduction reuses the ring and the filter bank");
    for (int e = tid; e < 64 * 224; e += NT) sR[e] = 0.f;
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int tile = 0; tile < 14; ++tile)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = lg * 16 + (2 * h + ct) * 4 + r, tap = tile * 16 + li;
                atomicAdd(sR + co * 224 + tap, dacc[ct][tile][r]);
            }
    __syncthreads();
    for (int e = tid; e < 64 * 224; e += NT) atomicAdd(dw + e, sR[e]);
}

}  // namespace eve

using namespace eve;

/* conv7x7/2 + InstanceNorm + ReLU + maxpool3x3/2 of the packed patches (eve_stem_pack_input layout).
   y_pool [N][IH/4][32][64] bf16, idx uint8 same shape (window position kh*3+kw), mean_rstd [N][64][2]. */
extern "C" int eve_stem_fwd_fused(int dtype, int N, int IH, int IW, const void* x_padded, const void* w_ohwi8, float eps,
                                  void* y_pool, uint8_t* idx, float* mean_rstd, eve_stream_t stream) {
    if ((dtype != EVE_DT_BF16 && dtype != EVE_DT_F16) || N <= 0 || IH <= 0 || (IH & 3) || IW != 128 || !x_padded || !w_ohwi8 || !y_pool || !idx || !mean_rstd)
        return set_error_msg("stem_fwd_fused: needs IW == 128 and IH a multiple of 4");
    const unsigned long long xb = (unsigned long long)N * (IH + 6) * SF_XROW;
    if (xb >= (1ull << 31)) return set_error_msg("stem_fwd_fused: packed input must stay below 2 GiB");
    const size_t lds = (size_t)SF_WAVES * SF_RING * SF_ROWB + SF_WBYTES + SF_WAVES * 64 * 2 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)stem_fwd_fused_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)stem_fwd_fused_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (g_cfg.stem_fwd_pairs) {      // round 4: two waves per image, 32 channels each, 16 waves per CU
        static bool attr2 = false;
        if (!attr2) {
            (void)hipFuncSetAttribute((const void*)stem_fwd_pairs_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)stem_fwd_pairs_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr2 = true;
        }
        const size_t lds2 = (size_t)SP_PAIRS * SF_RING * SF_ROWB + SF_WBYTES + 64;
        const unsigned blocks2 = N < 256 ? (unsigned)N : 256u;
        EVE_DISPATCH_H16(dtype, EVE_LAUNCH(EVE_HNAME(H, "stem_fwd_pairs_kernel<", ">"), stem_fwd_pairs_kernel<H>, dim3(blocks2), dim3(1024), lds2,
                                           (hipStream_t)stream, N, IH, (const H*)x_padded, (uint32_t)xb, (const H*)w_ohwi8, eps, (H*)y_pool, idx, mean_rstd));
        EVE_CHECK_LAUNCH();
        return 0;
    }
    // two waves per image while that still fits the 256 x 8 wave slots (B <= 16 clips per GPU): see the kernel
    const int split = g_cfg.stem_split;
    const int halves = (split && 2 * N <= 256 * SF_WAVES && (IH & 7) == 0) ? 2 : 1;
    unsigned blocks = N < 256 ? (unsigned)N : 256u;              // images are dealt round-robin over the workgroups
    EVE_DISPATCH_H16(dtype, EVE_LAUNCH(EVE_HNAME(H, "stem_fwd_fused_kernel<", ">"), stem_fwd_fused_kernel<H>, dim3(blocks), dim3(64 * SF_WAVES), lds,
                                       (hipStream_t)stream, N, IH, (const H*)x_padded, (uint32_t)xb, (const H*)w_ohwi8, eps, (H*)y_pool, idx, mean_rstd, halves));
    EVE_CHECK_LAUNCH();
    return 0;
}

/* d(conv1 output) [N][IH/2][64][64] bf16 from d(y_pool): the backward of eve_stem_fwd_fused up to the convolution
   output (the weight gradient then runs on it).  Recomputes the convolution from x_padded instead of reading it. */
extern "C" int eve_stem_bwd_dx(int dtype, int N, int IH, int IW, const void* x_padded, const void* w_ohwi8, const float* mean_rstd,
                               const void* dy_pool, const void* dy_pool2, const void* y_pool, const uint8_t* idx, void* dx, eve_stream_t stream) {
    if ((dtype != EVE_DT_BF16 && dtype != EVE_DT_F16) || N <= 0 || IH <= 0 || (IH & 3) || IW != 128 || !x_padded || !w_ohwi8 || !mean_rstd || !dy_pool || !y_pool || !idx || !dx)
        return set_error_msg("stem_bwd_dx: needs IW == 128 and IH a multiple of 4");
    const unsigned long long xb = (unsigned long long)N * (IH + 6) * SF_XROW;
    if (xb >= (1ull << 31)) return set_error_msg("stem_bwd_dx: packed input must stay below 2 GiB");
    const size_t lds = (size_t)SD_WAVES * SF_RING * SF_ROWB + SF_WBYTES + SD_WAVES * SF_KBYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)stem_bwd_dx_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)stem_bwd_dx_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    // two work items per image while that still fits the 256 x 8 wave slots (B <= 16 clips per GPU): see the kernel
    const int split = g_cfg.stem_split;
    const int halves = (split && 2 * N <= 256 * SD_WAVES && (IH & 7) == 0) ? 2 : 1;
    unsigned blocks = N * halves < 256 ? (unsigned)(N * halves) : 256u;
    EVE_DISPATCH_H16(dtype, EVE_LAUNCH(EVE_HNAME(H, "stem_bwd_dx_kernel<", ">"), stem_bwd_dx_kernel<H>, dim3(blocks), dim3(64 * SD_WAVES), lds,
                                       (hipStream_t)stream, N, IH, (const H*)x_padded, (uint32_t)xb, (const H*)w_ohwi8, mean_rstd, (const H*)dy_pool,
                                       (const H*)dy_pool2, (const H*)y_pool, idx, (H*)dx, halves));
    EVE_CHECK_LAUNCH();
    return 0;
}

/* The stem's backward and weight gradient in one launch (round 4): dw [64][7][8][4] float (accumulated; column kw = 7 and
   channel 3 do not exist) from d(y_pool), recomputing the convolution from x_padded; d(conv1 out) is never written.
   Replaces eve_stem_bwd_dx + eve_stem_wgrad (autograd of conv1 / bn1 / relu / maxpool, eye_net.py:48-50,106).             */
extern "C" int eve_stem_bwd_wgrad(int dtype, int N, int IH, int IW, const void* x_padded, const void* w_ohwi8, const float* mean_rstd,
                                  const void* dy_pool, const void* dy_pool2, const void* y_pool, const uint8_t* idx, float* dw,
                                  void* workspace, unsigned long long workspace_bytes, eve_stream_t stream) {
    if ((dtype != EVE_DT_BF16 && dtype != EVE_DT_F16) || N <= 0 || IH <= 0 || (IH & 3) || IW != 128 || !x_padded || !w_ohwi8 || !mean_rstd || !dy_pool || !y_pool || !idx || !dw)
        return set_error_msg("stem_bwd_wgrad: needs IW == 128 and IH a multiple of 4");
    const unsigned long long xb = (unsigned long long)N * (IH + 6) * SF_XROW;
    if (xb >= (1ull << 31)) return set_error_msg("stem_bwd_wgrad: packed input must stay below 2 GiB");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)stem_bwd_wgrad_kernel<bf16_t, SB_PAIRS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)stem_bwd_wgrad_kernel<f16_t, SB_PAIRS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    // Two workgroups of two pairs per CU (2 x 77 KB of LDS): the per-row barrier only ties the four waves of a workgroup, so the two
    // waves of a SIMD are in different phases of the row (fragment reads / MFMAs / gradient routing) most of the time (one workgroup
    // of four pairs: 0.866 ms at N = 1 920, two of two: 0.814).  Images are dealt round-robin over the workgroups first (pair p of
    // workgroup b takes image p * grid + b): a small batch puts one pair on every CU before it puts two on any.
    const size_t lds = (size_t)SB_PAIRS * SF_RING * SF_ROWB + SF_WBYTES + 2 * SB_PAIRS * SB_KBYTES + 2 * SB_PAIRS * SB_DTILE;
    const unsigned blocks = (N + SB_PAIRS - 1) / SB_PAIRS < 512 ? (unsigned)((N + SB_PAIRS - 1) / SB_PAIRS) : 512u;
    // The plane sums and the masked gradient come from a streaming pass of their own into the caller's scratch
    // (eve_stem_bwd_wgrad_workspace(dtype, N, IH) bytes); the fused kernel reads one pooled tensor + the codes, once.
    const int PH = IH / 4;
    const unsigned long long eg_bytes = ((unsigned long long)N * PH * 32 * 64 * 2 + 255) & ~255ull, k_bytes = (unsigned long long)N * 64 * 3 * 4;
    if (!workspace || workspace_bytes < eg_bytes + k_bytes || ((uintptr_t)workspace & 15))
        return set_error_msg("stem_bwd_wgrad: needs 16-byte aligned scratch of eve_stem_bwd_wgrad_workspace(dtype, N, IH) bytes");
    float* const kc = reinterpret_cast<float*>((char*)workspace + eg_bytes);
    EVE_DISPATCH_H16(dtype, EVE_LAUNCH(EVE_HNAME(H, "stem_grad_prep_kernel<", ">"), stem_grad_prep_kernel<H>, dim3(N), dim3(256), 0,
                                       (hipStream_t)stream, N, PH, (const H*)dy_pool, (const H*)dy_pool2, (const H*)y_pool, mean_rstd, (H*)workspace, kc));
    EVE_CHECK_LAUNCH();
    EVE_DISPATCH_H16(dtype, EVE_LAUNCH(EVE_HNAME(H, "stem_bwd_wgrad_kernel<", ", 2>"), (stem_bwd_wgrad_kernel<H, SB_PAIRS>), dim3(blocks), dim3(128 * SB_PAIRS), lds,
                                       (hipStream_t)stream, N, IH, (const H*)x_padded, (uint32_t)xb, (const H*)w_ohwi8, (const float*)kc, (const H*)workspace,
                                       idx, dw));
    EVE_CHECK_LAUNCH();
    return 0;
}
/* bytes of scratch eve_stem_bwd_wgrad needs (masked gradient [N][IH/4][32][64] + constants) */
extern "C" unsigned long long eve_stem_bwd_wgrad_workspace(int dtype, int N, int IH) {
    (void)dtype;
    if (N <= 0 || IH <= 0) return 0;
    return (((unsigned long long)N * (IH / 4) * 32 * 64 * 2 + 255) & ~255ull) + (unsigned long long)N * 64 * 3 * 4;
}
