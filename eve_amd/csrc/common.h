// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels of the EVE hot path.
// Everything here is written for MI355X only: 64-lane wavefronts, MFMA 16x16 tiles,
// 16-byte vector memory accesses, fp32 accumulation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "../../include/eve_hip.h"

namespace eve {
// f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}): a loop whose index is a compile-time constant
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }


typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---------------------------------------------------------------------------------------------
// element types: `float` and the two 16-bit formats `bf16_t` / `f16_t` (raw 16-bit storage, round-to-nearest-even
// conversions).  Every reduced-precision kernel is a template over the 16-bit format H: the data movement is identical,
// only the conversions and the MFMA opcode (v_mfma_f32_16x16x32_bf16 / _f16) differ.
// ---------------------------------------------------------------------------------------------
struct bf16_t { uint16_t v; };
struct f16_t { uint16_t v; };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t bits16) {
    return __builtin_bit_cast(float, bits16 << 16);
}
// gfx950 converts in hardware (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, round-to-nearest-even, two values per instruction)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {       // lo in bits 0..15
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ uint32_t pack2_f16(float lo, float hi) {
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, f16x2_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;                     // elements per 16-byte vector
    static constexpr int DT = EVE_DT_F32;
    static constexpr bool IS_BF16 = false;
    __device__ static __forceinline__ uint32_t pack2(float, float) { return 0u; }      // (16-bit formats only; keeps dead branches compilable)
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    // unpack a 16-byte vector into VEC floats / pack back
    __device__ static __forceinline__ void unpack(const uint4& q, float* f) {
        f[0] = __builtin_bit_cast(float, q.x); f[1] = __builtin_bit_cast(float, q.y);
        f[2] = __builtin_bit_cast(float, q.z); f[3] = __builtin_bit_cast(float, q.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__builtin_bit_cast(uint32_t, f[0]), __builtin_bit_cast(uint32_t, f[1]),
                          __builtin_bit_cast(uint32_t, f[2]), __builtin_bit_cast(uint32_t, f[3]));
    }
};
template <> struct Elem<bf16_t> {
    static constexpr int VEC = 8;
    static constexpr int DT = EVE_DT_BF16;
    static constexpr bool IS_BF16 = true;
    static constexpr uint32_t ONE2 = 0x3F803F80u;                 // two packed 1.0
    // the two halves of a packed pair / a pair from two floats / one value rounded to the format and back
    __device__ static __forceinline__ float lo(uint32_t w) { return bf16_bits_to_f32(w & 0xffffu); }
    __device__ static __forceinline__ float hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
    __device__ static __forceinline__ uint32_t pack2(float a, float b) { return pack2_bf16(a, b); }
    __device__ static __forceinline__ float round(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_bits_to_f32(p->v); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { p->v = (uint16_t)f32_to_bf16_bits(v); }
    __device__ static __forceinline__ void unpack(const uint4& q, float* f) {
        f[0] = lo(q.x); f[1] = hi(q.x); f[2] = lo(q.y); f[3] = hi(q.y);
        f[4] = lo(q.z); f[5] = hi(q.z); f[6] = lo(q.w); f[7] = hi(q.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
    }
    // acc(16 x 16) += A(16 rows x 32 k) * B(32 k x 16 cols); a / b are the lane's 8 consecutive k
    __device__ static __forceinline__ void mfma(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Elem<f16_t> {
    static constexpr int VEC = 8;
    static constexpr int DT = EVE_DT_F16;
    static constexpr bool IS_BF16 = false;
    static constexpr uint32_t ONE2 = 0x3C003C00u;
    __device__ static __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w).x; }
    __device__ static __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w).y; }
    __device__ static __forceinline__ uint32_t pack2(float a, float b) { return pack2_f16(a, b); }
    __device__ static __forceinline__ float round(float v) { return (float)(_Float16)v; }
    __device__ static __forceinline__ float ld(const f16_t* p) { return (float)__builtin_bit_cast(_Float16, p->v); }
    __device__ static __forceinline__ void st(f16_t* p, float v) { p->v = __builtin_bit_cast(uint16_t, (_Float16)v); }
    __device__ static __forceinline__ void unpack(const uint4& q, float* f) {
        f[0] = lo(q.x); f[1] = hi(q.x); f[2] = lo(q.y); f[3] = hi(q.y);
        f[4] = lo(q.z); f[5] = hi(q.z); f[6] = lo(q.w); f[7] = hi(q.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
    }
    __device__ static __forceinline__ void mfma(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
    }
};
// kernel symbol of a 16-bit instantiation, as rocprofv3 prints it (benchmark attribution)
#define EVE_HNAME(H, pre, post) (Elem<H>::IS_BF16 ? pre "eve::bf16_t" post : pre "eve::f16_t" post)
// run `body` with the 16-bit storage type of a dtype code bound to H (generic lambda taking a value of that type)
#define EVE_DISPATCH_H16(dtype, ...)                                        \
    do {                                                                    \
        if ((dtype) == EVE_DT_BF16) { using H = eve::bf16_t; __VA_ARGS__; } \
        else { using H = eve::f16_t; __VA_ARGS__; }                         \
    } while (0)

// ---------------------------------------------------------------------------------------------
// activations (the set the reference uses: ReLU, LeakyReLU(0.01), SELU, tanh, sigmoid)
// ---------------------------------------------------------------------------------------------
#define EVE_SELU_ALPHA 1.6732632423543772f
#define EVE_SELU_SCALE 1.0507009873554805f

__device__ __forceinline__ float act_fwd(float z, int act) {
    switch (act) {
        case EVE_ACT_RELU:    return z > 0.f ? z : 0.f;
        case EVE_ACT_LEAKY:   return z > 0.f ? z : 0.01f * z;
        case EVE_ACT_SELU:    return EVE_SELU_SCALE * (z > 0.f ? z : EVE_SELU_ALPHA * (__expf(z) - 1.f));
        case EVE_ACT_TANH:    return tanhf(z);
        case EVE_ACT_SIGMOID: return 1.f / (1.f + __expf(-z));
        default:              return z;
    }
}
// Conv epilogue activation on the four values a lane holds.  The selector is wave-uniform; a per-element
// switch costs more scalar branches than the epilogue has vector work and bloats the kernel past the
// instruction cache, so the kernels run the whole epilogue as one of two bodies: FAST (identity / ReLU, the
// ResNet trunk) or the general one.
// bit 8 of the epilogue word: out += result (the data gradient of a residual branch lands on the gradient the
// other branch already wrote)
#define EVE_EPI_ACC EVE_EPI_ACCUMULATE
__device__ __forceinline__ bool act_is_fast(int act) { return (act & 0xff) == EVE_ACT_NONE || (act & 0xff) == EVE_ACT_RELU; }
template <bool FAST>
__device__ __forceinline__ void act_fwd4(float* o, int act) {
    act &= 0xff;
    if (FAST) {
        if (act == EVE_ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = o[r] > 0.f ? o[r] : 0.f;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = act_fwd(o[r], act);
    }
}
// derivative expressed through the OUTPUT y = act(z) (all five are invertible enough for that)
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
    switch (act) {
        case EVE_ACT_RELU:    return y > 0.f ? 1.f : 0.f;
        case EVE_ACT_LEAKY:   return y > 0.f ? 1.f : 0.01f;
        case EVE_ACT_SELU:    return y > 0.f ? EVE_SELU_SCALE : y + EVE_SELU_SCALE * EVE_SELU_ALPHA;
        case EVE_ACT_TANH:    return 1.f - y * y;
        case EVE_ACT_SIGMOID: return y * (1.f - y);
        default:              return 1.f;
    }
}

// ---------------------------------------------------------------------------------------------
// division by a run-time constant (pixel index -> (n, y, x)); valid for dividends < 2^31
// ---------------------------------------------------------------------------------------------
struct FastDiv {
    uint32_t d, mul, shr;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) { f.mul = 0; f.shr = 0; return f; }
    uint32_t lg = 0;
    while ((1u << lg) < d) ++lg;                  // ceil(log2 d)
    uint32_t p = 31 + lg;
    f.mul = (uint32_t)(((1ull << p) + d - 1) / d);
    f.shr = p - 32;
    return f;
}
__device__ __forceinline__ uint32_t fd_div(uint32_t x, const FastDiv& f) {
    return f.d <= 1 ? x : (__umulhi(x, f.mul) >> f.shr);
}

// ---------------------------------------------------------------------------------------------
// wave64 reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware, bijective block-id remap (8 XCDs, block b is dispatched to XCD b % 8): gives each XCD
// a contiguous run of logical tiles so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblk) {
    const uint32_t q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace eve

#define EVE_CHECK_LAUNCH()                                             \
    do {                                                               \
        hipError_t e__ = hipGetLastError();                            \
        if (e__ != hipSuccess) return eve::set_error(e__, __func__);   \
    } while (0)

namespace eve {
int set_error(hipError_t e, const char* where);
int set_error_msg(const char* msg);
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
// 4 x 4 transpose of 16-byte elements across a lane quad (two DPP butterfly stages, per dword: select, quad_perm move, two
// selects): in: r[k] of lane q = element (k, q); out: r[m] of lane q = element (q, m).  Epilogues whose lanes hold four
// 16-byte chunks of ONE pixel use it so that a store instruction writes 64 contiguous bytes per quad (chunk q of the quad's
// pixel m) instead of 16 bytes into 64 different lines.
__device__ __forceinline__ void quad_transpose4x4(u32x4_t (&r)[4], const int lane) {
    const bool q_hi = (lane & 2) != 0, q_lo = (lane & 1) != 0;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t a = r[k][d], b = r[k + 2][d];
            const uint32_t got = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(q_hi ? a : b), 0x4e, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
            r[k][d] = q_hi ? got : a;
            r[k + 2][d] = q_hi ? b : got;
        }
#pragma unroll
    for (int k = 0; k < 4; k += 2)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t a = r[k][d], b = r[k + 1][d];
            const uint32_t got = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(q_lo ? a : b), 0xb1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
            r[k][d] = q_lo ? got : a;
            r[k + 1][d] = q_lo ? b : got;
        }
}

// symbol of the kernel the calling thread launched last (benchmark attribution, see eve_last_kernel)
extern thread_local const char* g_last_kernel;
// kernel selection, resolved once at library load (api.hip; include/eve_hip.h: eve_dispatch_config)
extern eve_dispatch_config g_cfg;
}  // namespace eve
#define EVE_MARK_KERNEL(name) (eve::g_last_kernel = (name))
#define EVE_LAUNCH(name, ...) do { EVE_MARK_KERNEL(name); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
