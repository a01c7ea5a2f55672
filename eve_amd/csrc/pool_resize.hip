// Pooling, resampling and layout kernels (NHWC, 16-byte channel vectors), gfx950.  All HBM-bound.
#include "common.h"

namespace eve {

static inline unsigned rgrid(long long rows) { return (unsigned)(rows < 256 * 32 ? (rows < 1 ? 1 : rows) : 256 * 32); }
static inline unsigned sgrid(long long items) {
    long long b = (items + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ---------------- 3x3 / stride 2 / pad 1 max-pool (torchvision ResNet stem) ----------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ idx, int IH, int IW, int OH,
                                                          int OW, int C, long long items) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = C / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvecs);
        long long t = i / cvecs;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const long long n = t / OH;
        float best[VEC];
        uint32_t bi[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; bi[e] = 0xffu; }
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if (ih < 0 || ih >= IH) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if (iw < 0 || iw >= IW) continue;
                float f[VEC];
                Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + ((n * IH + ih) * IW + iw) * C + cv * VEC), f);
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (f[e] > best[e] || bi[e] == 0xffu || f[e] != f[e]) { best[e] = f[e]; bi[e] = kh * 3 + kw; }
            }
        }
        reinterpret_cast<uint4*>(y)[i] = Elem<T>::pack(best);
        uint8_t* ip = idx + i * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) ip[e] = (uint8_t)bi[e];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          T* __restrict__ dx, int IH, int IW, int OH, int OW,
                                                          int C, long long items) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = C / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvecs);
        long long t = i / cvecs;
        const int iw = (int)(t % IW); t /= IW;
        const int ih = (int)(t % IH);
        const long long n = t / IH;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int oh = ih / 2; oh <= (ih + 1) / 2; ++oh) {
            if (oh >= OH) continue;
            const int kh = ih - (oh * 2 - 1);
            for (int ow = iw / 2; ow <= (iw + 1) / 2; ++ow) {
                if (ow >= OW) continue;
                const int kw = iw - (ow * 2 - 1);
                const uint32_t code = kh * 3 + kw;
                const long long o = ((n * OH + oh) * OW + ow) * C + cv * VEC;
                float g[VEC];
                Elem<T>::unpack(*reinterpret_cast<const uint4*>(dy + o), g);
                const uint8_t* ip = idx + o;
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (ip[e] == code) acc[e] += g[e];
            }
        }
        reinterpret_cast<uint4*>(dx)[i] = Elem<T>::pack(acc);
    }
}

// ---------------- global average pool ----------------
// F32SIDE: the pooled side ([N][C]: y of the forward, dy of the backward) is FLOAT32 memory holding values of format T -- the
// EyeNet trunk hands its features to a float32 tail, and the cast was a launch of its own in each direction (5 us each in a
// 4 ms step).  Bit-identical to pooling in T and casting: the forward rounds to T before widening, the backward rounds dy to T.
template <typename T, bool F32SIDE = false>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, void* __restrict__ y_, int HW,
                                                          int C, long long items) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = C / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvecs);
        const long long n = i / cvecs;
        float s[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] = 0.f;
        for (int p = 0; p < HW; ++p) {
            float f[VEC];
            Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + (n * HW + p) * C + cv * VEC), f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e] += f[e];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] /= (float)HW;
        const uint4 packed = Elem<T>::pack(s);
        if (F32SIDE) {
            float r[VEC];
            Elem<T>::unpack(packed, r);
            float* y = reinterpret_cast<float*>(y_) + i * VEC;
#pragma unroll
            for (int e = 0; e < VEC; e += 4) *reinterpret_cast<float4*>(y + e) = make_float4(r[e], r[e + 1], r[e + 2], r[e + 3]);
        } else {
            reinterpret_cast<uint4*>(y_)[i] = packed;
        }
    }
}
template <typename T, bool F32SIDE = false>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const void* __restrict__ dy_, T* __restrict__ dx, int HW,
                                                          int C, long long items) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = C / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvecs);
        const long long n = i / ((long long)HW * cvecs);
        float g[VEC];
        if (F32SIDE) {
            const float* dy = reinterpret_cast<const float*>(dy_) + n * C + cv * VEC;
            float r[VEC];
#pragma unroll
            for (int e = 0; e < VEC; e += 4) {
                const float4 v = *reinterpret_cast<const float4*>(dy + e);
                r[e] = v.x; r[e + 1] = v.y; r[e + 2] = v.z; r[e + 3] = v.w;
            }
            Elem<T>::unpack(Elem<T>::pack(r), g);             // rounded to T, as the separate cast did
        } else {
            Elem<T>::unpack(*reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(dy_) + n * C + cv * VEC), g);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) g[e] /= (float)HW;
        reinterpret_cast<uint4*>(dx)[i] = Elem<T>::pack(g);
    }
}

// ---------------- adaptive max-pool (window [floor(i*I/O), ceil((i+1)*I/O)) ) ----------------
__device__ __forceinline__ int ad_start(int o, int I, int O) { return (o * I) / O; }
__device__ __forceinline__ int ad_end(int o, int I, int O) { return ((o + 1) * I + O - 1) / O; }

template <typename T>
__global__ __launch_bounds__(256) void adapool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          int32_t* __restrict__ idx, int IH, int IW, int OH,
                                                          int OW, int C, long long items) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = C / VEC;
    // one image ROW per workgroup turn: the 64-bit divisions of the flat index are paid once per row, not per vector
    // (round 4: these four resize kernels ran at ~1.5 TB/s on RefineNet's 72x128 / 36x64 levels)
    const int rowitems = OW * cvecs;
    const long long rows = items / rowitems;
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const int oh = (int)(r % OH);
        const long long n = r / OH;
    for (int j = threadIdx.x; j < rowitems; j += 256) {
        const int cv = j % cvecs, ow = j / cvecs;
        const long long i = r * rowitems + j;
        float best[VEC];
        int32_t bi[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; bi[e] = -1; }
        for (int ih = ad_start(oh, IH, OH); ih < ad_end(oh, IH, OH); ++ih)
            for (int iw = ad_start(ow, IW, OW); iw < ad_end(ow, IW, OW); ++iw) {
                float f[VEC];
                Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + ((n * IH + ih) * IW + iw) * C + cv * VEC), f);
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (f[e] > best[e] || bi[e] < 0 || f[e] != f[e]) { best[e] = f[e]; bi[e] = ih * IW + iw; }
            }
        reinterpret_cast<uint4*>(y)[i] = Elem<T>::pack(best);
        int32_t* ip = idx + i * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) ip[e] = bi[e];
    }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void adapool_bwd_kernel(const T* __restrict__ dy, const int32_t* __restrict__ idx,
                                                          const T* __restrict__ add, T* __restrict__ dx, int IH, int IW, int OH, int OW,
                                                          int C, long long items) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = C / VEC;
    // one image ROW per workgroup turn: the 64-bit divisions of the flat index are paid once per row, not per vector
    // (round 4: these four resize kernels ran at ~1.5 TB/s on RefineNet's 72x128 / 36x64 levels)
    const int rowitems = IW * cvecs;
    const long long rows = items / rowitems;
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const int ih = (int)(r % IH);
        const long long n = r / IH;
    for (int j = threadIdx.x; j < rowitems; j += 256) {
        const int cv = j % cvecs, iw = j / cvecs;
        const long long i = r * rowitems + j;
        const int me = ih * IW + iw;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        const int oh_c = (ih * OH) / IH, ow_c = (iw * OW) / IW;
        for (int oh = max(0, oh_c - 1); oh <= min(OH - 1, oh_c + 1); ++oh) {
            if (ih < ad_start(oh, IH, OH) || ih >= ad_end(oh, IH, OH)) continue;
            for (int ow = max(0, ow_c - 1); ow <= min(OW - 1, ow_c + 1); ++ow) {
                if (iw < ad_start(ow, IW, OW) || iw >= ad_end(ow, IW, OW)) continue;
                const long long o = ((n * OH + oh) * OW + ow) * C + cv * VEC;
                float g[VEC];
                Elem<T>::unpack(*reinterpret_cast<const uint4*>(dy + o), g);
                const int32_t* ip = idx + o;
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (ip[e] == me) acc[e] += g[e];
            }
        }
        if (add) {              // (the pooled gradient is rounded to the storage format first: what the separate add kernel saw)
            float r[VEC], o[VEC];
            Elem<T>::unpack(Elem<T>::pack(acc), r);
            Elem<T>::unpack(reinterpret_cast<const uint4*>(add)[i], o);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = r[e] + o[e];
        }
        reinterpret_cast<uint4*>(dx)[i] = Elem<T>::pack(acc);
    }
    }
}

// ---------------- bilinear resize, align_corners = False ----------------
struct Lerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lerp lerp_src(int o, int I, float scale) {
    float s = scale * ((float)o + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    Lerp l;
    l.i0 = (int)s;
    if (l.i0 > I - 1) l.i0 = I - 1;
    l.i1 = l.i0 + (l.i0 < I - 1 ? 1 : 0);
    l.w1 = s - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int IH,
                                                           int IW, int OH, int OW, int C, long long items) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = C / VEC;
    const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
    // one image ROW per workgroup turn: the 64-bit divisions of the flat index are paid once per row, not per vector
    // (round 4: these four resize kernels ran at ~1.5 TB/s on RefineNet's 72x128 / 36x64 levels)
    const int rowitems = OW * cvecs;
    const long long rows = items / rowitems;
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const int oh = (int)(r % OH);
        const long long n = r / OH;
    for (int j = threadIdx.x; j < rowitems; j += 256) {
        const int cv = j % cvecs, ow = j / cvecs;
        const long long i = r * rowitems + j;
        const Lerp ly = lerp_src(oh, IH, sh), lx = lerp_src(ow, IW, sw);
        const T* b = x + n * IH * IW * C + cv * VEC;
        float a00[VEC], a01[VEC], a10[VEC], a11[VEC], o[VEC];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(b + ((long long)ly.i0 * IW + lx.i0) * C), a00);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(b + ((long long)ly.i0 * IW + lx.i1) * C), a01);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(b + ((long long)ly.i1 * IW + lx.i0) * C), a10);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(b + ((long long)ly.i1 * IW + lx.i1) * C), a11);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            o[e] = ly.w0 * (lx.w0 * a00[e] + lx.w1 * a01[e]) + ly.w1 * (lx.w0 * a10[e] + lx.w1 * a11[e]);
        reinterpret_cast<uint4*>(y)[i] = Elem<T>::pack(o);
    }
    }
}

// adjoint by gathering: for an input pixel, visit the few output pixels whose stencil touches it
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int IH,
                                                           int IW, int OH, int OW, int C, long long items) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = C / VEC;
    const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
    // one image ROW per workgroup turn: the 64-bit divisions of the flat index are paid once per row, not per vector
    // (round 4: these four resize kernels ran at ~1.5 TB/s on RefineNet's 72x128 / 36x64 levels)
    const int rowitems = IW * cvecs;
    const long long rows = items / rowitems;
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const int ih = (int)(r % IH);
        const long long n = r / IH;
    for (int j = threadIdx.x; j < rowitems; j += 256) {
        const int cv = j % cvecs, iw = j / cvecs;
        const long long i = r * rowitems + j;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        const int oh_lo = max(0, (int)floorf(((float)ih - 1.f + 0.5f) / sh - 0.5f) - 1);
        const int oh_hi = min(OH - 1, (int)ceilf(((float)ih + 1.f + 0.5f) / sh - 0.5f) + 1);
        const int ow_lo = max(0, (int)floorf(((float)iw - 1.f + 0.5f) / sw - 0.5f) - 1);
        const int ow_hi = min(OW - 1, (int)ceilf(((float)iw + 1.f + 0.5f) / sw - 0.5f) + 1);
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            const Lerp ly = lerp_src(oh, IH, sh);
            const float wy = (ly.i0 == ih ? ly.w0 : 0.f) + (ly.i1 == ih ? ly.w1 : 0.f);
            if (wy == 0.f) continue;
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const Lerp lx = lerp_src(ow, IW, sw);
                const float wx = (lx.i0 == iw ? lx.w0 : 0.f) + (lx.i1 == iw ? lx.w1 : 0.f);
                if (wx == 0.f) continue;
                float g[VEC];
                Elem<T>::unpack(*reinterpret_cast<const uint4*>(dy + ((n * OH + oh) * OW + ow) * C + cv * VEC), g);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += wy * wx * g[e];
            }
        }
        reinterpret_cast<uint4*>(dx)[i] = Elem<T>::pack(acc);
    }
    }
}

// ---------------- layout / dtype plumbing ----------------
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                           int C, int HW, int Cpad, long long items) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvecs = Cpad / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        // pixel-major so that consecutive threads read consecutive NCHW pixels of one channel
        const long long pix = i % ((long long)HW);
        long long t = i / HW;
        const int cv = (int)(t % cvecs);
        const long long n = t / cvecs;
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int c = cv * VEC + e;
            f[e] = c < C ? src[(n * C + c) * HW + pix] : 0.f;
        }
        *reinterpret_cast<uint4*>(dst + (n * HW + pix) * Cpad + cv * VEC) = Elem<T>::pack(f);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst,
                                                           int C, int HW, int Cpad, long long items) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const long long pix = i % HW;
        long long t = i / HW;
        const int c = (int)(t % C);
        const long long n = t / C;
        dst[i] = Elem<T>::ld(src + (n * HW + pix) * Cpad + c);
    }
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        Elem<D>::st(dst + i, Elem<S>::ld(src + i));
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ ohwi,
                                                           T* __restrict__ ihwo, int Cout, int taps, int Cin,
                                                           long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = w[i];
        if (ohwi) Elem<T>::st(ohwi + i, v);
        if (ihwo) {
            const int ci = (int)(i % Cin);
            long long t = i / Cin;
            const int tap = (int)(t % taps);
            const long long co = t / taps;
            Elem<T>::st(ihwo + ((long long)ci * taps + tap) * Cout + co, v);
        }
    }
}

// all conv weights of a model in one launch (the packs are rebuilt after every optimiser step: 20 launches of a few
// microseconds each were 0.14 ms of a 15 ms step)
struct PackItem { const float* w; void* ohwi; void* ihwo; int Cout, taps, Cin, sCout, sCin; long long start; };   // start = first tile
struct PackTable { PackItem it[EVE_PACK_BATCH_MAX]; int count; long long total; };                   // total tiles

// One workgroup = one 32 (output channels) x 32 (input channels) tile of one filter tap: rows are read and written
// along the input channels (OHWI copy) and, transposed through LDS, along the output channels (IHWO copy), so both
// destinations get contiguous stores (the element-wise version scattered 2-byte writes for the IHWO copy).
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const PackTable tb) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (long long g = blockIdx.x; g < tb.total; g += gridDim.x) {
        int k = 0;
        while (k + 1 < tb.count && g >= tb.it[k + 1].start) ++k;
        const PackItem& q = tb.it[k];
        const int tci = (q.Cin + 31) / 32, tco = (q.Cout + 31) / 32;
        long long t = g - q.start;
        const int ci0 = (int)(t % tci) * 32; t /= tci;
        const int tap = (int)(t % q.taps);
        const int co0 = (int)(t / q.taps) * 32;
        (void)tco;
#pragma unroll
        for (int r = ty; r < 32; r += 8) {
            const int co = co0 + r, ci = ci0 + tx;
            float v = 0.f;
            if (co < q.Cout && ci < q.Cin) {
                // (the source may be narrower than the packed copies: padding channels are written as zeros)
                if (co < q.sCout && ci < q.sCin) v = q.w[((long long)co * q.taps + tap) * q.sCin + ci];
                if (q.ohwi) Elem<T>::st((T*)q.ohwi + ((long long)co * q.taps + tap) * q.Cin + ci, v);
            }
            tile[r][tx] = v;
        }
        __syncthreads();
        if (q.ihwo) {
#pragma unroll
            for (int r = ty; r < 32; r += 8) {
                const int ci = ci0 + r, co = co0 + tx;
                if (ci < q.Cin && co < q.Cout)
                    Elem<T>::st((T*)q.ihwo + ((long long)ci * q.taps + tap) * q.Cout + co, tile[tx][r]);
            }
        }
        __syncthreads();
    }
}

}  // namespace eve

using namespace eve;


static int chk(int dtype, int C, const char* who) {
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    if (((unsigned)dtype > (unsigned)EVE_DT_F16) || C <= 0 || C % vec) return set_error_msg(who);
    return 0;
}

extern "C" int eve_maxpool3x3s2_fwd(int dtype, int N, int IH, int IW, int C, const void* x, void* y, uint8_t* idx,
                                    eve_stream_t stream) {
    if (int e = chk(dtype, C, "maxpool_fwd: bad dtype / C")) return e;
    if (N <= 0 || IH <= 0 || IW <= 0 || !x || !y || !idx) return set_error_msg("maxpool_fwd: bad arguments");
    const int OH = (IH - 1) / 2 + 1, OW = (IW - 1) / 2 + 1;
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    const long long items = (long long)N * OH * OW * (C / vec);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(sgrid(items)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, idx, IH, IW, OH, OW, C, items);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(maxpool_fwd_kernel<f16_t>, dim3(sgrid(items)), dim3(256), 0, s, (const f16_t*)x, (f16_t*)y, idx, IH, IW, OH, OW, C, items);
    else                      hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(sgrid(items)), dim3(256), 0, s, (const float*)x, (float*)y, idx, IH, IW, OH, OW, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_maxpool3x3s2_bwd(int dtype, int N, int IH, int IW, int C, const void* dy, const uint8_t* idx,
                                    void* dx, eve_stream_t stream) {
    if (int e = chk(dtype, C, "maxpool_bwd: bad dtype / C")) return e;
    if (N <= 0 || IH <= 0 || IW <= 0 || !dy || !dx || !idx) return set_error_msg("maxpool_bwd: bad arguments");
    const int OH = (IH - 1) / 2 + 1, OW = (IW - 1) / 2 + 1;
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    const long long items = (long long)N * IH * IW * (C / vec);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(sgrid(items)), dim3(256), 0, s, (const bf16_t*)dy, idx, (bf16_t*)dx, IH, IW, OH, OW, C, items);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(maxpool_bwd_kernel<f16_t>, dim3(sgrid(items)), dim3(256), 0, s, (const f16_t*)dy, idx, (f16_t*)dx, IH, IW, OH, OW, C, items);
    else                      hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(sgrid(items)), dim3(256), 0, s, (const float*)dy, idx, (float*)dx, IH, IW, OH, OW, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_avgpool_fwd(int dtype, int N, int HW, int C, const void* x, void* y, eve_stream_t stream) {
    if (int e = chk(dtype, C, "avgpool_fwd: bad dtype / C")) return e;
    if (N <= 0 || HW <= 0 || !x || !y) return set_error_msg("avgpool_fwd: bad arguments");
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    const long long items = (long long)N * (C / vec);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, dim3(sgrid(items)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, HW, C, items);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(avgpool_fwd_kernel<f16_t>, dim3(sgrid(items)), dim3(256), 0, s, (const f16_t*)x, (f16_t*)y, HW, C, items);
    else                      hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(sgrid(items)), dim3(256), 0, s, (const float*)x, (float*)y, HW, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_avgpool_bwd(int dtype, int N, int HW, int C, const void* dy, void* dx, eve_stream_t stream) {
    if (int e = chk(dtype, C, "avgpool_bwd: bad dtype / C")) return e;
    if (N <= 0 || HW <= 0 || !dy || !dx) return set_error_msg("avgpool_bwd: bad arguments");
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    const long long items = (long long)N * HW * (C / vec);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(sgrid(items)), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, HW, C, items);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(avgpool_bwd_kernel<f16_t>, dim3(sgrid(items)), dim3(256), 0, s, (const f16_t*)dy, (f16_t*)dx, HW, C, items);
    else                      hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(sgrid(items)), dim3(256), 0, s, (const float*)dy, (float*)dx, HW, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
// the same with the pooled side in float32 memory (see the kernels): dtype is the format of the [N][HW][C] side
extern "C" int eve_avgpool_fwd_f32(int dtype, int N, int HW, int C, const void* x, float* y, eve_stream_t stream) {
    if (int e = chk(dtype, C, "avgpool_fwd_f32: bad dtype / C")) return e;
    if (N <= 0 || HW <= 0 || !x || !y) return set_error_msg("avgpool_fwd_f32: bad arguments");
    if (dtype == EVE_DT_F32) return eve_avgpool_fwd(dtype, N, HW, C, x, y, stream);
    const long long items = (long long)N * (C / 8);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL((avgpool_fwd_kernel<bf16_t, true>), dim3(sgrid(items)), dim3(256), 0, s, (const bf16_t*)x, (void*)y, HW, C, items);
    else                      hipLaunchKernelGGL((avgpool_fwd_kernel<f16_t, true>), dim3(sgrid(items)), dim3(256), 0, s, (const f16_t*)x, (void*)y, HW, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_avgpool_bwd_f32(int dtype, int N, int HW, int C, const float* dy, void* dx, eve_stream_t stream) {
    if (int e = chk(dtype, C, "avgpool_bwd_f32: bad dtype / C")) return e;
    if (N <= 0 || HW <= 0 || !dy || !dx) return set_error_msg("avgpool_bwd_f32: bad arguments");
    if (dtype == EVE_DT_F32) return eve_avgpool_bwd(dtype, N, HW, C, dy, dx, stream);
    const long long items = (long long)N * HW * (C / 8);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL((avgpool_bwd_kernel<bf16_t, true>), dim3(sgrid(items)), dim3(256), 0, s, (const void*)dy, (bf16_t*)dx, HW, C, items);
    else                      hipLaunchKernelGGL((avgpool_bwd_kernel<f16_t, true>), dim3(sgrid(items)), dim3(256), 0, s, (const void*)dy, (f16_t*)dx, HW, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_adaptive_maxpool_fwd(int dtype, int N, int IH, int IW, int OH, int OW, int C, const void* x,
                                        void* y, int32_t* idx, eve_stream_t stream) {
    if (int e = chk(dtype, C, "adaptive_maxpool_fwd: bad dtype / C")) return e;
    if (N <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || OH > IH || OW > IW || !x || !y || !idx)
        return set_error_msg("adaptive_maxpool_fwd: bad arguments");
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    const long long items = (long long)N * OH * OW * (C / vec);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(adapool_fwd_kernel<bf16_t>, dim3(rgrid((long long)N * OH)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, idx, IH, IW, OH, OW, C, items);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(adapool_fwd_kernel<f16_t>, dim3(rgrid((long long)N * OH)), dim3(256), 0, s, (const f16_t*)x, (f16_t*)y, idx, IH, IW, OH, OW, C, items);
    else                      hipLaunchKernelGGL(adapool_fwd_kernel<float>, dim3(rgrid((long long)N * OH)), dim3(256), 0, s, (const float*)x, (float*)y, idx, IH, IW, OH, OW, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_adaptive_maxpool_bwd(int dtype, int N, int IH, int IW, int OH, int OW, int C, const void* dy,
                                        const int32_t* idx, const void* add, void* dx, eve_stream_t stream) {
    if (int e = chk(dtype, C, "adaptive_maxpool_bwd: bad dtype / C")) return e;
    if (N <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || OH > IH || OW > IW || !dy || !dx || !idx)
        return set_error_msg("adaptive_maxpool_bwd: bad arguments");
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    const long long items = (long long)N * IH * IW * (C / vec);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(adapool_bwd_kernel<bf16_t>, dim3(rgrid((long long)N * IH)), dim3(256), 0, s, (const bf16_t*)dy, idx, (const bf16_t*)add, (bf16_t*)dx, IH, IW, OH, OW, C, items);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(adapool_bwd_kernel<f16_t>, dim3(rgrid((long long)N * IH)), dim3(256), 0, s, (const f16_t*)dy, idx, (const f16_t*)add, (f16_t*)dx, IH, IW, OH, OW, C, items);
    else                      hipLaunchKernelGGL(adapool_bwd_kernel<float>, dim3(rgrid((long long)N * IH)), dim3(256), 0, s, (const float*)dy, idx, (const float*)add, (float*)dx, IH, IW, OH, OW, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_bilinear_fwd(int dtype, int N, int IH, int IW, int OH, int OW, int C, const void* x, void* y,
                                eve_stream_t stream) {
    if (int e = chk(dtype, C, "bilinear_fwd: bad dtype / C")) return e;
    if (N <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || !x || !y) return set_error_msg("bilinear_fwd: bad arguments");
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    const long long items = (long long)N * OH * OW * (C / vec);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(bilinear_fwd_kernel<bf16_t>, dim3(rgrid((long long)N * OH)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, IH, IW, OH, OW, C, items);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(bilinear_fwd_kernel<f16_t>, dim3(rgrid((long long)N * OH)), dim3(256), 0, s, (const f16_t*)x, (f16_t*)y, IH, IW, OH, OW, C, items);
    else                      hipLaunchKernelGGL(bilinear_fwd_kernel<float>, dim3(rgrid((long long)N * OH)), dim3(256), 0, s, (const float*)x, (float*)y, IH, IW, OH, OW, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_bilinear_bwd(int dtype, int N, int IH, int IW, int OH, int OW, int C, const void* dy, void* dx,
                                eve_stream_t stream) {
    if (int e = chk(dtype, C, "bilinear_bwd: bad dtype / C")) return e;
    if (N <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || !dy || !dx) return set_error_msg("bilinear_bwd: bad arguments");
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    const long long items = (long long)N * IH * IW * (C / vec);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(bilinear_bwd_kernel<bf16_t>, dim3(rgrid((long long)N * IH)), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, IH, IW, OH, OW, C, items);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(bilinear_bwd_kernel<f16_t>, dim3(rgrid((long long)N * IH)), dim3(256), 0, s, (const f16_t*)dy, (f16_t*)dx, IH, IW, OH, OW, C, items);
    else                      hipLaunchKernelGGL(bilinear_bwd_kernel<float>, dim3(rgrid((long long)N * IH)), dim3(256), 0, s, (const float*)dy, (float*)dx, IH, IW, OH, OW, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_nchw_to_nhwc(int dtype_dst, int N, int C, int H, int W, int Cpad, const float* src_nchw,
                                void* dst_nhwc, eve_stream_t stream) {
    if (int e = chk(dtype_dst, Cpad, "nchw_to_nhwc: bad dtype / Cpad")) return e;
    if (N <= 0 || C <= 0 || C > Cpad || H <= 0 || W <= 0 || !src_nchw || !dst_nhwc) return set_error_msg("nchw_to_nhwc: bad arguments");
    const int vec = dtype_dst != EVE_DT_F32 ? 8 : 4;
    const long long items = (long long)N * H * W * (Cpad / vec);
    hipStream_t s = (hipStream_t)stream;
    if (dtype_dst == EVE_DT_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(sgrid(items)), dim3(256), 0, s, src_nchw, (bf16_t*)dst_nhwc, C, H * W, Cpad, items);
    else if (dtype_dst == EVE_DT_F16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16_t>, dim3(sgrid(items)), dim3(256), 0, s, src_nchw, (f16_t*)dst_nhwc, C, H * W, Cpad, items);
    else                          hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(sgrid(items)), dim3(256), 0, s, src_nchw, (float*)dst_nhwc, C, H * W, Cpad, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_nhwc_to_nchw(int dtype_src, int N, int C, int H, int W, int Cpad, const void* src_nhwc,
                                float* dst_nchw, eve_stream_t stream) {
    if ((unsigned)dtype_src > (unsigned)EVE_DT_F16) return set_error_msg("nhwc_to_nchw: bad dtype");
    if (N <= 0 || C <= 0 || C > Cpad || H <= 0 || W <= 0 || !src_nhwc || !dst_nchw) return set_error_msg("nhwc_to_nchw: bad arguments");
    const long long items = (long long)N * C * H * W;
    hipStream_t s = (hipStream_t)stream;
    if (dtype_src == EVE_DT_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(sgrid(items)), dim3(256), 0, s, (const bf16_t*)src_nhwc, dst_nchw, C, H * W, Cpad, items);
    else if (dtype_src == EVE_DT_F16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16_t>, dim3(sgrid(items)), dim3(256), 0, s, (const f16_t*)src_nhwc, dst_nchw, C, H * W, Cpad, items);
    else                          hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(sgrid(items)), dim3(256), 0, s, (const float*)src_nhwc, dst_nchw, C, H * W, Cpad, items);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_cast(int dtype_src, int dtype_dst, long long n, const void* src, void* dst, eve_stream_t stream) {
    if (n <= 0 || !src || !dst) return set_error_msg("cast: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = sgrid(n);
#define EVE_CAST_CASE(DS, TS, DD, TD) \
    if (dtype_src == DS && dtype_dst == DD) hipLaunchKernelGGL((cast_kernel<TS, TD>), dim3(g), dim3(256), 0, s, (const TS*)src, (TD*)dst, n); else
    EVE_CAST_CASE(EVE_DT_F32, float, EVE_DT_BF16, bf16_t)
    EVE_CAST_CASE(EVE_DT_BF16, bf16_t, EVE_DT_F32, float)
    EVE_CAST_CASE(EVE_DT_F32, float, EVE_DT_F16, f16_t)
    EVE_CAST_CASE(EVE_DT_F16, f16_t, EVE_DT_F32, float)
    EVE_CAST_CASE(EVE_DT_F32, float, EVE_DT_F32, float)
    EVE_CAST_CASE(EVE_DT_BF16, bf16_t, EVE_DT_BF16, bf16_t)
    EVE_CAST_CASE(EVE_DT_F16, f16_t, EVE_DT_F16, f16_t)
#undef EVE_CAST_CASE
    return set_error_msg("cast: bad dtype");
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_pack_weights_batch(int dtype_dst, int count, const eve_pack_item* items, eve_stream_t stream) {
    if ((unsigned)dtype_dst > (unsigned)EVE_DT_F16) return set_error_msg("pack_weights_batch: bad dtype");
    if (count <= 0 || count > EVE_PACK_BATCH_MAX || !items) return set_error_msg("pack_weights_batch: 1..EVE_PACK_BATCH_MAX items");
    PackTable tb;
    long long total = 0;
    for (int i = 0; i < count; ++i) {
        const eve_pack_item& q = items[i];
        if (q.Cout <= 0 || q.taps <= 0 || q.Cin <= 0 || !q.w_ohwi || (!q.dst_ohwi && !q.dst_ihwo))
            return set_error_msg("pack_weights_batch: bad item");
        if (q.src_Cout < 0 || q.src_Cout > q.Cout || q.src_Cin < 0 || q.src_Cin > q.Cin) return set_error_msg("pack_weights_batch: bad source shape");
        tb.it[i] = PackItem{q.w_ohwi, q.dst_ohwi, q.dst_ihwo, q.Cout, q.taps, q.Cin, q.src_Cout ? q.src_Cout : q.Cout,
                            q.src_Cin ? q.src_Cin : q.Cin, total};
        total += (long long)((q.Cout + 31) / 32) * q.taps * ((q.Cin + 31) / 32);
    }
    tb.count = count; tb.total = total;
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)(total < 4096 ? total : 4096);
    if (dtype_dst == EVE_DT_BF16) hipLaunchKernelGGL(pack_weights_batch_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, tb);
    else if (dtype_dst == EVE_DT_F16) hipLaunchKernelGGL(pack_weights_batch_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, tb);
    else                          hipLaunchKernelGGL(pack_weights_batch_kernel<float>, dim3(blocks), dim3(256), 0, s, tb);
    EVE_CHECK_LAUNCH();
    return 0;
}
extern "C" int eve_pack_weights(int dtype_dst, int Cout, int taps, int Cin, const float* w_ohwi, void* dst_ohwi,
                                void* dst_ihwo, eve_stream_t stream) {
    if ((unsigned)dtype_dst > (unsigned)EVE_DT_F16) return set_error_msg("pack_weights: bad dtype");
    if (Cout <= 0 || taps <= 0 || Cin <= 0 || !w_ohwi || (!dst_ohwi && !dst_ihwo)) return set_error_msg("pack_weights: bad arguments");
    const long long n = (long long)Cout * taps * Cin;
    hipStream_t s = (hipStream_t)stream;
    if (dtype_dst == EVE_DT_BF16) hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(sgrid(n)), dim3(256), 0, s, w_ohwi, (bf16_t*)dst_ohwi, (bf16_t*)dst_ihwo, Cout, taps, Cin, n);
    else if (dtype_dst == EVE_DT_F16) hipLaunchKernelGGL(pack_weights_kernel<f16_t>, dim3(sgrid(n)), dim3(256), 0, s, w_ohwi, (f16_t*)dst_ohwi, (f16_t*)dst_ihwo, Cout, taps, Cin, n);
    else                          hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(sgrid(n)), dim3(256), 0, s, w_ohwi, (float*)dst_ohwi, (float*)dst_ihwo, Cout, taps, Cin, n);
    EVE_CHECK_LAUNCH();
    return 0;
}
