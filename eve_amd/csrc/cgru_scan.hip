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

__device__ __forceinline__ float cg_sigmoid(float z) { return 1.f / (1.f + __expf(-z)); }

__global__ __launch_bounds__(256) void cgru_scan_fwd_kernel(const int B, const int T, const bf16_t* __restrict__ xs,
                                                            const bf16_t* __restrict__ h0, const bf16_t* __restrict__ w1,
                                                            const float* __restrict__ b1, const bf16_t* __restrict__ w2,
                                                            const float* __restrict__ b2, bf16_t* __restrict__ hs,
                                                            bf16_t* __restrict__ hs_tm, bf16_t* __restrict__ ru,
                                                            bf16_t* __restrict__ rh, bf16_t* __restrict__ og) {
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
    int aaddr[9][4];
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
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hr = ti * 7 + py + t / 3, hx = px + t % 3;
            aaddr[t][mt] = ((hr * 10 + hx) << 6) + ((lg ^ ((hr & 1) << 1)) << 4);
        }
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
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();

    uint32_t gs = 0;                                  // stream position (ring phase)
    for (int t = 0; t < T; ++t) {
        const bool more_t = t + 1 < T;
        // next frame's x tile: in flight during this frame's GEMMs
#pragma unroll
        for (int j = 0; j < 4; ++j)
            xq[j] = (more_t && x_glob[j] >= 0) ? *reinterpret_cast<const uint4*>(xs + x_glob[j] + (size_t)(t + 1) * CG_PIX * CG_C)
                                               : make_uint4(0, 0, 0, 0);
        f32x4_t acc[4][4], ug[4][4];
#pragma unroll
        for (int conv = 0; conv < 2; ++conv) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const bool active = conv == 0 || wn == 1;             // conv2 has 64 output channels: one wave column
            for (int sl = 0; sl < 4; ++sl, gs += 9) {
                // conv1 reads cat[x, h] = planes 0,1,2,3;  conv2 reads cat[r*h, x] = planes 4,5,0,1
                const int plane = conv == 0 ? sl : (sl < 2 ? 4 + sl : sl - 2);
                const uint32_t la = lds0 + plane * CG_SLICE;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    // weight tile of stream position +3
                    int nsl = sl + (tap + 3) / 9, nconv = conv, live = 1;
                    if (nsl == 4) { nsl = 0; nconv = conv + 1; if (nconv == 2) { nconv = 0; live = more_t; } }
                    issue_w(nconv, nsl, (tap + 3) % 9, (int)((gs + tap + 3) & 3), live != 0);
                    if (active) {
                        const uint32_t lb = ldsB + ((gs + tap) & 3) * CG_BSLOT;
                        bf16x8_t fx[4], fw[4];
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            fx[mt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const EVE_LDS cg_u32x4_t*>((uintptr_t)(la + aaddr[tap][mt])));
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            fw[nt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const EVE_LDS cg_u32x4_t*>(
                                (uintptr_t)(lb + (conv == 0 ? brow1[nt] : brow2[nt]))));
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nt], fx[mt], acc[mt][nt], 0, 0, 0);
                    }
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile of the next step (issued 2 steps ago) landed
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
                        const uint32_t p0 = pack2_bf16(v[0], v[1]), p1 = pack2_bf16(v[2], v[3]);
                        v[0] = bf16_bits_to_f32(p0 & 0xffffu); v[1] = __builtin_bit_cast(float, p0 & 0xffff0000u);
                        v[2] = bf16_bits_to_f32(p1 & 0xffffu); v[3] = __builtin_bit_cast(float, p1 & 0xffff0000u);
                        const int pg = pix_glob[mt];
                        const size_t ptm = (size_t)t * B * CG_PIX + pix_tm[mt];           // time-major: what the backward walks
                        if (pg >= 0)
                            *reinterpret_cast<uint2*>(ru + ptm * 128 + wn * 64 + c) = make_uint2(p0, p1);
                        if (wn == 0) {
                            const cg_u32x2_t hq = *reinterpret_cast<const EVE_LDS cg_u32x2_t*>((uintptr_t)lds_c4(2, mt, c));
                            const uint32_t q0 = pack2_bf16(v[0] * bf16_bits_to_f32(hq.x & 0xffffu), v[1] * __builtin_bit_cast(float, hq.x & 0xffff0000u));
                            const uint32_t q1 = pack2_bf16(v[2] * bf16_bits_to_f32(hq.y & 0xffffu), v[3] * __builtin_bit_cast(float, hq.y & 0xffff0000u));
                            if (pg >= 0) {
                                *reinterpret_cast<EVE_LDS cg_u32x2_t*>((uintptr_t)lds_c4(4, mt, c)) = cg_u32x2_t{q0, q1};
                                *reinterpret_cast<uint2*>(rh + ptm * CG_C + c) = make_uint2(q0, q1);
                            }
                        } else {
                            ug[mt][nt] = f32x4_t{v[0], v[1], v[2], v[3]};
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
                    float o[4] = {tanhf(acc[mt][nt][0] + bv.x), tanhf(acc[mt][nt][1] + bv.y), tanhf(acc[mt][nt][2] + bv.z),
                                  tanhf(acc[mt][nt][3] + bv.w)};
                    const uint32_t o0 = pack2_bf16(o[0], o[1]), o1 = pack2_bf16(o[2], o[3]);
                    o[0] = bf16_bits_to_f32(o0 & 0xffffu); o[1] = __builtin_bit_cast(float, o0 & 0xffff0000u);
                    o[2] = bf16_bits_to_f32(o1 & 0xffffu); o[3] = __builtin_bit_cast(float, o1 & 0xffff0000u);
                    const uint32_t ha = lds_c4(2, mt, c);
                    const cg_u32x2_t hq = *reinterpret_cast<const EVE_LDS cg_u32x2_t*>((uintptr_t)ha);
                    const float hv[4] = {bf16_bits_to_f32(hq.x & 0xffffu), __builtin_bit_cast(float, hq.x & 0xffff0000u),
                                         bf16_bits_to_f32(hq.y & 0xffffu), __builtin_bit_cast(float, hq.y & 0xffff0000u)};
                    float hn[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) hn[r] = (1.f - ug[mt][nt][r]) * o[r] + ug[mt][nt][r] * hv[r];
                    const uint32_t n0 = pack2_bf16(hn[0], hn[1]), n1 = pack2_bf16(hn[2], hn[3]);
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

}  // namespace eve

using namespace eve;

/* CGRUCell over T frames in one launch (bf16): xs [B][T][5][8][64] NHWC, h0 [B][5][8][64] or NULL (zeros),
   w1 = gates_1 weight OHWI [128][3][3][128] (input channels: x then h), w2 = gate_2 weight OHWI [64][3][3][128]
   (input channels: r*h then x), biases float.  Outputs (bf16): hs [B][T][5][8][64] = hidden states in the caller's
   (sequence, frame) order; and TIME-major [T][B][5][8][.] for the backward, which walks frames: hs_tm, ru = the two
   sigmoid gates (128 channels), rh = r * h_{t-1}, og = tanh output gate. */
extern "C" int eve_cgru_scan_fwd(int B, int T, const void* xs, const void* h0, const void* w1, const float* b1, const void* w2,
                                 const float* b2, void* hs, void* hs_tm, void* ru, void* rh, void* og, eve_stream_t stream) {
    if (B <= 0 || T <= 0 || !xs || !w1 || !b1 || !w2 || !b2 || !hs || !hs_tm || !ru || !rh || !og)
        return set_error_msg("cgru_scan_fwd: bad arguments");
    if ((long long)B * T * CG_PIX * 128 >= (1ll << 31)) return set_error_msg("cgru_scan_fwd: clip too large for 32-bit offsets");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)cgru_scan_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    EVE_MARK_KERNEL("cgru_scan_fwd_kernel");
    hipLaunchKernelGGL(cgru_scan_fwd_kernel, dim3((B + CG_IMG - 1) / CG_IMG), dim3(256), CG_LDS, (hipStream_t)stream, B, T,
                       (const bf16_t*)xs, (const bf16_t*)h0, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, (bf16_t*)hs, (bf16_t*)hs_tm,
                       (bf16_t*)ru, (bf16_t*)rh, (bf16_t*)og);
    EVE_CHECK_LAUNCH();
    return 0;
}
