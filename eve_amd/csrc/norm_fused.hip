// Register-resident InstanceNorm forward / backward for planes that fit one workgroup's registers
// (H*W*C <= 64 Ki elements: every EyeNet stage after the stem, every RefineNet level below 36x64).
// One workgroup owns one image plane: it reads it ONCE from HBM, keeps the packed vectors in VGPRs,
// does the per-channel reductions with wave shuffles + one LDS exchange, and writes the result.
//   forward : 1 read (+1 for the residual) + 1 write     (was: stats pass + apply pass = 2 reads + 1 write)
//   backward: 3 reads (dy, x, y) + 1-2 writes             (was: two passes over all three = 6 reads)
// Statistics are the exact two-pass form (mean, then centred second moment) over registers.
#include <stdlib.h>
#include "common.h"

// Contraction is SYNTACTIC in this file: a * b + c written in one expression is one fma, a product stored by one statement and
// used by another is rounded in between.  hipcc's default (-ffp-contract=fast) fuses across statements as it sees fit per
// kernel -- the branch-free trunk kernels below and the generic kernels then disagree in the last bit (the trunk forward formed
// x - mean as fma(-sum, 1/HW, x), the generic one subtracted the rounded mean: 9 % of the float32 rstd values one ulp apart),
// and tests/test_gpu_kernels.py holds the two families to bit equality.
#pragma clang fp contract(on)

namespace eve {

// Sum s[0..VEC) over all threads of the block that share this thread's channel vector (cv = tid % cvecs).
// On return every thread holds the totals for its channels.
template <int VEC>
__device__ __forceinline__ void plane_allreduce(float* s, float* sh, int tid, int nthreads, int cvecs) {
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    for (int o = 32; o >= cvecs; o >>= 1)
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] += __shfl_xor(s[e], o, 64);
    __syncthreads();                                   // sh may still be read from a previous call
#pragma unroll
    for (int e = 0; e < VEC; ++e) sh[(wave * 64 + lane) * VEC + e] = s[e];
    __syncthreads();
    const int wstep = cvecs > 64 ? cvecs / 64 : 1;     // waves that cover the same channel vectors
    const int src_lane = cvecs >= 64 ? lane : (lane % cvecs);
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    for (int w = wave % wstep; w < nwaves; w += wstep)
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] += sh[(w * 64 + src_lane) * VEC + e];
}

// A plane may be dealt to 2^sl workgroups, each owning C >> sl channels of every pixel (the statistics are per channel, so
// the parts never talk): a 64 Ki-element plane (ResNet layer 1) then runs as two 512-thread workgroups instead of one
// 1 024-thread one, and a CU holds two independent workgroups whose load / reduce / store phases overlap (3.95-4.1 TB/s ->
// the 4.4-5.0 of the 512-thread planes).  The parts of a plane are 8 blocks apart: same XCD under round-robin dispatch, so
// the two halves of every 128-byte line meet in one L2.
struct PlanePart { uint32_t plane, part; };
__device__ __forceinline__ PlanePart plane_part(uint32_t b, int sl) {
    if (sl == 0) return PlanePart{b, 0u};
    const uint32_t per = 8u << sl, grp = b / per, r = b - grp * per;
    return PlanePart{grp * 8u + (r & 7u), r >> 3};
}

template <typename T, int VPT, int ACT>
__global__ __launch_bounds__(1024) void in_fwd_fused_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const T* __restrict__ res,
                                                            int act_rt, T* __restrict__ y, float* __restrict__ mr,
                                                            unsigned char* __restrict__ mask, int N, int HW, int C, int sl,
                                                            float eps) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh[1024 * VEC];
    const int act = ACT >= 0 ? ACT : act_rt;        // ACT >= 0: activation known at compile time
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const PlanePart pp = plane_part(blockIdx.x, sl);
    if (pp.plane >= (uint32_t)N) return;
    const int Cl = C >> sl, cvecs = Cl / VEC, cfull = C / VEC, nvec = HW * cvecs;
    const int cv = tid % cvecs, lc = 31 - __builtin_clz(cvecs);
    const size_t base = (size_t)pp.plane * HW * cfull + pp.part * cvecs + cv;
    const int c0 = pp.part * Cl;                    // first channel of this part
    auto gi = [&](int i) { return base + (size_t)(i >> lc) * cfull; };       // global vector index of local vector i
    uint4 q[VPT];
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = tid + j * nthreads;
        q[j] = i < nvec ? reinterpret_cast<const uint4*>(x)[gi(i)] : make_uint4(0, 0, 0, 0);
        float f[VEC];
        Elem<T>::unpack(q[j], f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] += f[e];
    }
    plane_allreduce<VEC>(s, sh, tid, nthreads, cvecs);
    float mean[VEC], a[VEC], b[VEC];
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mean[e] = s[e] * inv; s[e] = 0.f; }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        if (tid + j * nthreads < nvec) {
            float f[VEC];
            Elem<T>::unpack(q[j], f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = f[e] - mean[e]; s[e] += d * d; }
        }
    }
    plane_allreduce<VEC>(s, sh, tid, nthreads, cvecs);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const float rstd = rsqrtf(s[e] * inv + eps);
        const int c = c0 + cv * VEC + e;
        a[e] = rstd; b[e] = -mean[e] * rstd;
        if (gamma) { a[e] *= gamma[c]; b[e] = b[e] * gamma[c] + beta[c]; }
        if (tid < cvecs) {
            mr[((size_t)pp.plane * C + c) * 2] = mean[e];
            mr[((size_t)pp.plane * C + c) * 2 + 1] = rstd;
        }
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = tid + j * nthreads;
        if (i < nvec) {
            float f[VEC], r[VEC];
            Elem<T>::unpack(q[j], f);
            if (res) Elem<T>::unpack(reinterpret_cast<const uint4*>(res)[gi(i)], r);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float z = fmaf(f[e], a[e], b[e]);         // (explicit: the trunk kernels below must round the same way)
                if (res) z += r[e];
                f[e] = act_fwd(z, act);
            }
            reinterpret_cast<uint4*>(y)[gi(i)] = Elem<T>::pack(f);
            if (mask) {                 // one byte per 16-byte vector: bit e = (output e > 0), all the ReLU backward needs of y
                unsigned m = 0;
#pragma unroll
                for (int e = 0; e < VEC; ++e) m |= (f[e] > 0.f ? 1u : 0u) << e;
                mask[gi(i)] = (unsigned char)m;
            }
        }
    }
}

template <typename T, int VPT, int ACT>
__global__ __launch_bounds__(1024) void in_bwd_fused_kernel(const T* __restrict__ dy, const T* __restrict__ dy2,
                                                            const T* __restrict__ y,
                                                            const T* __restrict__ x, const float* __restrict__ mr,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int act_rt,
                                                            T* __restrict__ dx, T* __restrict__ dres,
                                                            float* __restrict__ sums, const unsigned char* __restrict__ mask,
                                                            const T* __restrict__ dx_add, int N, int HW, int C, int sl) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh[1024 * VEC];
    const int act = ACT >= 0 ? ACT : act_rt;        // ACT >= 0: activation known at compile time (no per-element switch)
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const PlanePart pp = plane_part(blockIdx.x, sl);
    if (pp.plane >= (uint32_t)N) return;
    const int Cl = C >> sl, cvecs = Cl / VEC, cfull = C / VEC, nvec = HW * cvecs;
    const int cv = tid % cvecs, lc = 31 - __builtin_clz(cvecs);
    const size_t base = (size_t)pp.plane * HW * cfull + pp.part * cvecs + cv;
    const int c0 = pp.part * Cl;
    auto gi = [&](int i) { return base + (size_t)(i >> lc) * cfull; };
    float mean[VEC], rstd[VEC];
    {
        const float* m = mr + ((size_t)pp.plane * C + c0 + cv * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { mean[e] = m[2 * e]; rstd[e] = m[2 * e + 1]; }
    }
    // the forward's scale / shift (in_fwd_fused_kernel: same expressions, same roundings), for act' from x when y is not given
    float za[VEC], zb[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        za[e] = rstd[e]; zb[e] = -mean[e] * rstd[e];
        if (gamma && beta) { const int c = c0 + cv * VEC + e; za[e] *= gamma[c]; zb[e] = zb[e] * gamma[c] + beta[c]; }
    }
    uint4 qg[VPT], qx[VPT];          // g = dy * act'(y) re-packed, and x
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = tid + j * nthreads;
        qg[j] = make_uint4(0, 0, 0, 0);
        qx[j] = make_uint4(0, 0, 0, 0);
        if (i < nvec) {
            float g[VEC], xx[VEC];
            Elem<T>::unpack(reinterpret_cast<const uint4*>(dy)[gi(i)], g);
            if (dy2) {                  // the gradient arrives as two summands (residual fork): add on load
                float g2[VEC];
                Elem<T>::unpack(reinterpret_cast<const uint4*>(dy2)[gi(i)], g2);
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[e] += g2[e];
            }
            qx[j] = reinterpret_cast<const uint4*>(x)[gi(i)];
            Elem<T>::unpack(qx[j], xx);
            if (act == EVE_ACT_RELU && mask) {           // sign bits written by the forward instead of the whole of y
                const unsigned m = mask[gi(i)];
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[e] = (m >> e) & 1u ? g[e] : 0.f;
                qg[j] = Elem<T>::pack(g);
                if (dres) reinterpret_cast<uint4*>(dres)[gi(i)] = qg[j];
                Elem<T>::unpack(qg[j], g);
            } else if (act != EVE_ACT_NONE) {
                float yy[VEC];
                if (y) {
                    Elem<T>::unpack(reinterpret_cast<const uint4*>(y)[gi(i)], yy);
                } else if (gamma) { // affine, no residual: y = act(x * za + zb) exactly as the forward formed it
#pragma unroll
                    for (int e = 0; e < VEC; ++e) yy[e] = act_fwd(fmaf(xx[e], za[e], zb[e]), act);
                } else {            // no affine, no residual: y = act(xhat), and sign(xhat) = sign(x - mean)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) yy[e] = act_fwd((xx[e] - mean[e]) * rstd[e], act);
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[e] *= act_grad_from_out(yy[e], act);
                qg[j] = Elem<T>::pack(g);
                if (dres) reinterpret_cast<uint4*>(dres)[gi(i)] = qg[j];
                // keep the arithmetic on the SAME rounded g the two-pass kernel and dres see
                Elem<T>::unpack(qg[j], g);
            } else {
                qg[j] = Elem<T>::pack(g);
                if (dres) reinterpret_cast<uint4*>(dres)[gi(i)] = qg[j];
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                s1[e] += g[e];
                s2[e] += g[e] * (xx[e] - mean[e]) * rstd[e];
            }
        }
    }
    plane_allreduce<VEC>(s1, sh, tid, nthreads, cvecs);
    plane_allreduce<VEC>(s2, sh, tid, nthreads, cvecs);
    if (sums && tid < cvecs) {
        float* o = sums + ((size_t)pp.plane * C + c0 + cv * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { o[2 * e] = s1[e]; o[2 * e + 1] = s2[e]; }
    }
    float k[VEC];
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        s1[e] *= inv; s2[e] *= inv;
        k[e] = rstd[e] * (gamma ? gamma[c0 + cv * VEC + e] : 1.f);
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = tid + j * nthreads;
        if (i < nvec) {
            float g[VEC], xx[VEC];
            Elem<T>::unpack(qg[j], g);
            Elem<T>::unpack(qx[j], xx);
#pragma unroll
            for (int e = 0; e < VEC; ++e) g[e] = k[e] * (g[e] - s1[e] - (xx[e] - mean[e]) * rstd[e] * s2[e]);
            if (dx_add) {           // (eve_instnorm_act_bwd's dx_add: the fork's other gradient, added to the rounded dx)
                float r[VEC], o2[VEC];
                Elem<T>::unpack(Elem<T>::pack(g), r);
                Elem<T>::unpack(reinterpret_cast<const uint4*>(dx_add)[gi(i)], o2);
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[e] = r[e] + o2[e];
            }
            reinterpret_cast<uint4*>(dx)[gi(i)] = Elem<T>::pack(g);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the ResNet trunk's instances (no affine; ReLU or identity; planes that fill their workgroup exactly) as kernels
// whose every optional operand is a TEMPLATE flag.  The generic kernels above decide per vector, at run time, whether there
// is a residual / a second gradient summand / a mask / a dres: hipcc turns each of those into a branch around a load, and a
// load under a branch is followed by `s_waitcnt vmcnt(0)` -- which on gfx9 also waits for every store issued before it.  The
// ISA of in_fwd_fused_kernel<bf16, 8, 1> had its eight residual loads and eight stores as one dependent chain (load, wait,
// store, load, wait, ...: eight exposed HBM round trips per thread); in_bwd_fused_kernel<bf16, 8, 1> waited after EVERY one
// of its 24-32 loads.  Here all loads of a plane part are issued back to back before the first use (8-32 x 16 bytes in flight
// per thread), and the stores go out as one burst.  The arithmetic -- order of operations, rounding points, reduction tree --
// is the generic kernels', so results are bit-identical (tests/test_gpu_kernels.py compares both against the ATen
// restatement and against each other).
// ---------------------------------------------------------------------------------------------------------------------
// (register budget: 128 per lane = two 512-thread workgroups, or one 1 024-thread workgroup, per CU.  What has to stay live is the PACKED data -- 8 x 4 dwords
//  per tensor -- so every phase re-unpacks from the packed registers; `opaque` keeps hipcc from carrying the unpacked floats of
//  one phase into the next, which cost 198-252 registers or, bounded to 128, 270-560 bytes of scratch per lane.)
__device__ __forceinline__ void opaque(uint4& q) { asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w)); }
// ... and a fence on the running sums at the end of an iteration: volatile statements keep their order, so iteration j + 1's
// opaque() -- and with it that iteration's unpacking -- cannot be scheduled before iteration j's arithmetic has produced the
// sums (left alone, the scheduler unpacks all eight vectors first "to hide latency" and spills them)
template <int VEC>
__device__ __forceinline__ void fence_sums(float* s) {
    if constexpr (VEC == 8) asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]));
    else asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]));
}

// Addresses: one buffer resource per tensor and plane part (wave-uniform: four SGPRs), ONE 32-bit lane offset, and the
// vector index j as the instruction's scalar offset j * step -- no address arithmetic per load or store at all.  (nthreads is
// a multiple of cvecs, so vector i = tid + j * nthreads sits nthreads / cvecs pixels below vector tid.)
typedef unsigned int v4u32_t __attribute__((ext_vector_type(4)));
struct TrunkAddr {
    size_t base;            // byte offset of this plane part inside the tensor (uniform)
    uint32_t bytes;         // bytes from there to the end of the plane (buffer bound)
    uint32_t off, step;     // this lane's byte offset; bytes between vectors j and j + 1 (uniform)
};
__device__ __forceinline__ TrunkAddr trunk_addr(const PlanePart pp, int HW, int cfull, int cvecs, int tid, int nthreads) {
    const int lc = 31 - __builtin_clz(cvecs);
    const uint32_t plane = (uint32_t)__builtin_amdgcn_readfirstlane((int)pp.plane), part = (uint32_t)__builtin_amdgcn_readfirstlane((int)pp.part);
    TrunkAddr a;
    a.base = ((size_t)plane * (size_t)HW * cfull + (size_t)part * cvecs) * 16;
    a.bytes = ((uint32_t)HW * cfull - part * cvecs) * 16u;
    a.off = ((uint32_t)(tid >> lc) * cfull + (uint32_t)(tid & (cvecs - 1))) * 16u;
    a.step = (uint32_t)(nthreads >> lc) * cfull * 16u;
    return a;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t trunk_rsrc(const void* tensor, size_t base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)tensor + base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 ldv(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
// Stores take the WHOLE offset in the vector operand (one v_add per store) and an immediate 0 as the scalar offset.  With a
// scalar-REGISTER offset hipcc (clang 22) emits no wait state between `buffer_store_dwordx4 v[58:61], .., s37 offen` and a
// VALU write of v59 in the next instruction -- LLVM's hazard recogniser assumes that form has no store-data hazard -- and on
// gfx950 it has one: dword 1 of ~12 lanes of dres went out with the new value, a few times per launch, run-to-run different
// (tools/dbg_in.py; found by test_instnorm_fwd_bwd's bit-equality of the mask and y paths).  The immediate form is protected.
__device__ __forceinline__ void stv(const uint4& q, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t uniform_off) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32_t, q), r, (int)(voff + uniform_off), 0, 0);
}

template <typename T, int VPT, int ACT, bool RES, bool MASK>
__global__ __launch_bounds__(1024) void in_fwd_trunk_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                              float* __restrict__ mr, unsigned char* __restrict__ mask, int N, int HW,
                                                              int C, int sl, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh[1024 * VEC];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const PlanePart pp = plane_part(blockIdx.x, sl);
    if (pp.plane >= (uint32_t)N) return;
    const int Cl = C >> sl, cvecs = Cl / VEC, cfull = C / VEC;
    const int cv = tid % cvecs;
    const int c0 = pp.part * Cl;
    const TrunkAddr ad = trunk_addr(pp, HW, cfull, cvecs, tid, nthreads);
    const __amdgpu_buffer_rsrc_t rx = trunk_rsrc(x, ad.base, ad.bytes), rr_ = trunk_rsrc(RES ? (const void*)res : (const void*)x, ad.base, ad.bytes);
    uint4 q[VPT], r[RES ? VPT : 1];
#pragma unroll
    for (int j = 0; j < VPT; ++j) q[j] = ldv(rx, ad.off, j * ad.step);
    if (RES) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) r[j] = ldv(rr_, ad.off, j * ad.step);
    }
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        float f[VEC];
        Elem<T>::unpack(q[j], f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] += f[e];
        opaque(q[j]);
    }
    plane_allreduce<VEC>(s, sh, tid, nthreads, cvecs);
    float mean[VEC];
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mean[e] = s[e] * inv; s[e] = 0.f; }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        float f[VEC];
        Elem<T>::unpack(q[j], f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float d = f[e] - mean[e]; s[e] += d * d; }
        opaque(q[j]);
    }
    plane_allreduce<VEC>(s, sh, tid, nthreads, cvecs);
    float a[VEC], b[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const float rstd = rsqrtf(s[e] * inv + eps);
        a[e] = rstd; b[e] = -mean[e] * rstd;
        // (without this hipcc may contract the shift's multiply into the apply -- fma(-mean, rstd, f * a) -- where the generic
        //  kernel, whose b can also carry an affine term, computes fma(f, a, b): one ulp apart in float32)
        asm volatile("" : "+v"(b[e]));
        if (tid < cvecs) {
            const int c = c0 + cv * VEC + e;
            mr[((size_t)pp.plane * C + c) * 2] = mean[e];
            mr[((size_t)pp.plane * C + c) * 2 + 1] = rstd;
        }
    }
    const __amdgpu_buffer_rsrc_t ry = trunk_rsrc(y, ad.base, ad.bytes);
    // (the mask has one byte per 16-byte vector: same indices, 1/16 of the byte offsets)
    const __amdgpu_buffer_rsrc_t rm = trunk_rsrc(MASK ? (const void*)mask : (const void*)y, ad.base >> 4, ad.bytes >> 4);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        float f[VEC], rr[VEC];
        Elem<T>::unpack(q[j], f);
        if (RES) Elem<T>::unpack(r[j], rr);
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float z = fmaf(f[e], a[e], b[e]);             // b is a finished product (fence below): one rounding, as in the generic kernel
            if (RES) z += rr[e];
            f[e] = ACT == EVE_ACT_RELU ? (z > 0.f ? z : 0.f) : z;
            if (MASK) m |= (f[e] > 0.f ? 1u : 0u) << e;
        }
        stv(Elem<T>::pack(f), ry, ad.off, j * ad.step);
        if (MASK) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)m, rm, (int)(ad.off >> 4), (int)(j * (ad.step >> 4)), 0);
    }
}

// MODE 0: dx from (dy, x), ReLU' recomputed from x (mid-block InstanceNorm: no affine, no residual)
// MODE 1: dx and dres from (dy [+ dy2], sign mask, x)  (block-end InstanceNorm + residual + ReLU)
// MODE 2: dx from (dy [+ dy2], x), no activation      (down-sample branch)
template <typename T, int VPT, int MODE, bool DY2>
__global__ __launch_bounds__(1024) void in_bwd_trunk_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ x,
                                                              const float* __restrict__ mr, T* __restrict__ dx, T* __restrict__ dres,
                                                              float* __restrict__ sums, const unsigned char* __restrict__ mask, int N,
                                                              int HW, int C, int sl) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh[1024 * VEC];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const PlanePart pp = plane_part(blockIdx.x, sl);
    if (pp.plane >= (uint32_t)N) return;
    const int Cl = C >> sl, cvecs = Cl / VEC, cfull = C / VEC;
    const int cv = tid % cvecs;
    const int c0 = pp.part * Cl;
    const TrunkAddr ad = trunk_addr(pp, HW, cfull, cvecs, tid, nthreads);
    const __amdgpu_buffer_rsrc_t rg = trunk_rsrc(dy, ad.base, ad.bytes), rx = trunk_rsrc(x, ad.base, ad.bytes);
    const __amdgpu_buffer_rsrc_t rg2 = trunk_rsrc(DY2 ? (const void*)dy2 : (const void*)dy, ad.base, ad.bytes);
    const __amdgpu_buffer_rsrc_t rm = trunk_rsrc(MODE == 1 ? (const void*)mask : (const void*)dy, ad.base >> 4, ad.bytes >> 4);
    // ---- every load of this plane part, back to back ----
    uint4 qg[VPT], qx[VPT], q2[DY2 ? VPT : 1];
    unsigned mk[MODE == 1 ? VPT : 1];
#pragma unroll
    for (int j = 0; j < VPT; ++j) qg[j] = ldv(rg, ad.off, j * ad.step);
    if (DY2) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) q2[j] = ldv(rg2, ad.off, j * ad.step);
    }
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) mk[j] = __builtin_amdgcn_raw_buffer_load_b8(rm, (int)(ad.off >> 4), (int)(j * (ad.step >> 4)), 0);
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) qx[j] = ldv(rx, ad.off, j * ad.step);
    // ---- pass 1 (needs no statistics): g = mask(dy + dy2), rounded to the storage format; dres goes out at once ----
    if (DY2 || MODE == 1) {
        const __amdgpu_buffer_rsrc_t rd = trunk_rsrc(MODE == 1 ? (const void*)dres : (const void*)dy, ad.base, ad.bytes);
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            float g[VEC];
            opaque(qg[j]);
            Elem<T>::unpack(qg[j], g);
            if (DY2) {
                float g2[VEC];
                Elem<T>::unpack(q2[j], g2);
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[e] += g2[e];
            }
            if (MODE == 1) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[e] = (mk[j] >> e) & 1u ? g[e] : 0.f;
            }
            qg[j] = Elem<T>::pack(g);
            if (MODE == 1) stv(qg[j], rd, ad.off, j * ad.step);
            opaque(qg[j]);
        }
    }
    asm volatile("" ::: "memory");           // (the statistics' loads stay behind pass 1: 16 fewer live registers there)
    float mean[VEC], rstd[VEC];
    {
        const float* m = mr + ((size_t)pp.plane * C + c0 + cv * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { mean[e] = m[2 * e]; rstd[e] = m[2 * e + 1]; }
    }
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        float g[VEC], xx[VEC];
        opaque(qg[j]);
        opaque(qx[j]);
        Elem<T>::unpack(qg[j], g);
        Elem<T>::unpack(qx[j], xx);
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float z = (xx[e] - mean[e]) * rstd[e];
                g[e] *= (z > 0.f ? z : 0.f) > 0.f ? 1.f : 0.f;
            }
            qg[j] = Elem<T>::pack(g);
            Elem<T>::unpack(qg[j], g);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            s1[e] += g[e];
            s2[e] += g[e] * (xx[e] - mean[e]) * rstd[e];
        }
        fence_sums<VEC>(s1);
        fence_sums<VEC>(s2);
        opaque(qg[j]);
        opaque(qx[j]);
    }
    plane_allreduce<VEC>(s1, sh, tid, nthreads, cvecs);
    plane_allreduce<VEC>(s2, sh, tid, nthreads, cvecs);
    if (sums && tid < cvecs) {
        float* o = sums + ((size_t)pp.plane * C + c0 + cv * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { o[2 * e] = s1[e]; o[2 * e + 1] = s2[e]; }
    }
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s1[e] *= inv; s2[e] *= inv; }
    const __amdgpu_buffer_rsrc_t ro = trunk_rsrc(dx, ad.base, ad.bytes);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        float g[VEC], xx[VEC];
        opaque(qg[j]);
        opaque(qx[j]);
        Elem<T>::unpack(qg[j], g);
        Elem<T>::unpack(qx[j], xx);
#pragma unroll
        for (int e = 0; e < VEC; ++e) g[e] = rstd[e] * (g[e] - s1[e] - (xx[e] - mean[e]) * rstd[e] * s2[e]);
        qg[j] = Elem<T>::pack(g);
        opaque(qg[j]);
        stv(qg[j], ro, ad.off, j * ad.step);
    }
}

// threads per block and vectors per thread for a plane of nvec 16-byte vectors; false if it does not fit
static bool fused_plan(int nvec, int cvecs, int& threads, int& vpt, int& sl, bool allow_split, bool allow_big) {
    if (nvec <= 0 || cvecs <= 0 || cvecs > 128 || (cvecs & (cvecs - 1))) return false;
    // planes that would need a 1 024-thread workgroup go to two 512-thread ones, half the channels each
    const int split_on = g_cfg.in_split;
    sl = 0;
    // (bf16 only: the float32 instantiation is the parity mode and keeps its summation order)
    if (split_on && allow_split && nvec > 4096 && nvec <= 8192 && cvecs >= 2) { sl = 1; nvec /= 2; cvecs /= 2; }
    // round 4: planes beyond one workgroup's registers are dealt by channels as far as it takes -- the ResNet trunk on 256 x 256
    // patches (BASELINE configs[4]: layer 1 = 64 x 64 x 64 = 32 Ki vectors -> four parts, layer 2 = 16 Ki -> two) then runs on
    // the branch-free trunk kernels instead of the two-pass ones (statistics pass + apply pass: every operand read twice).
    // Only WITHOUT affine parameters (allow_big): RefineNet's planes (72x128 and 36x64 pixels, 16-64 channels: 2^k x 9 216
    // vectors, VPT = 9) were measured through the generic kernels this way and lost to the two-pass kernels -- 16 bytes of
    // every 32-128-byte pixel per workgroup, one 1 024-thread workgroup per CU whose load, reduce and store phases do not
    // overlap: 2.5-3.0 TB/s against 2.9-3.5 (profiles/r04_notes.md); C3 46.2 -> 44.6 ms with it off.
    allow_big = allow_big && g_cfg.in_big_planes && allow_split;
    if (allow_big)
        while (nvec > 8192 && cvecs >= 2) { ++sl; nvec /= 2; cvecs /= 2; }
    if (nvec > 8192) {
        if (!allow_big || nvec > 9216) return false;
        vpt = 9;
        threads = ((nvec + 8) / 9 + 63) / 64 * 64;
        return true;
    }
    // as many vectors per thread as leaves >= `min_threads` threads: fewer, fatter workgroups per plane let several
    // planes share a CU, so one plane's reduction phase overlaps another's loads / stores
    const int min_threads = g_cfg.in_min_threads;
    vpt = 1;
    while (vpt < 8 && ((nvec + vpt - 1) / vpt > 1024 || (nvec + 2 * vpt - 1) / (2 * vpt) >= min_threads)) vpt *= 2;
    if ((nvec + vpt - 1) / vpt > 1024) return false;
    const int q = cvecs > 64 ? cvecs : 64;
    threads = ((nvec + vpt - 1) / vpt + q - 1) / q * q;
    return threads <= 1024;
}

}  // namespace eve

using namespace eve;

// (the kernel symbol is recorded the way rocprofv3 prints it, for the bench's per-kernel attribution)
#define LAUNCH_VPT_A(KERNEL, T, TS, A, AS, ...)                                                                              \
    switch (vpt) {                                                                                                           \
        case 1: EVE_LAUNCH(#KERNEL "<" TS ", 1, " AS ">", (KERNEL<T, 1, A>), dim3(grid), dim3(threads), 0, s, __VA_ARGS__); break;  \
        case 2: EVE_LAUNCH(#KERNEL "<" TS ", 2, " AS ">", (KERNEL<T, 2, A>), dim3(grid), dim3(threads), 0, s, __VA_ARGS__); break;  \
        case 4: EVE_LAUNCH(#KERNEL "<" TS ", 4, " AS ">", (KERNEL<T, 4, A>), dim3(grid), dim3(threads), 0, s, __VA_ARGS__); break;  \
        case 9: EVE_LAUNCH(#KERNEL "<" TS ", 9, " AS ">", (KERNEL<T, 9, A>), dim3(grid), dim3(threads), 0, s, __VA_ARGS__); break;  \
        default: EVE_LAUNCH(#KERNEL "<" TS ", 8, " AS ">", (KERNEL<T, 8, A>), dim3(grid), dim3(threads), 0, s, __VA_ARGS__); break; \
    }
// the two activations of the ResNet trunk get their own instantiation; everything else takes the run-time switch
#define LAUNCH_VPT(KERNEL, T, TS, ...)                                                                    \
    if (act == EVE_ACT_NONE) { LAUNCH_VPT_A(KERNEL, T, TS, EVE_ACT_NONE, "0", __VA_ARGS__) }              \
    else if (act == EVE_ACT_RELU) { LAUNCH_VPT_A(KERNEL, T, TS, EVE_ACT_RELU, "1", __VA_ARGS__) }         \
    else { LAUNCH_VPT_A(KERNEL, T, TS, -1, "-1", __VA_ARGS__) }

/* returns 0 on launch, -1 if the plane does not fit the fused kernel (caller falls back), >0 on error */
extern "C" int eve_instnorm_fwd_fused(int dtype, int N, int HW, int C, const void* x, const float* gamma,
                                      const float* beta, const void* res, int act, float eps, void* y,
                                      float* mean_rstd, unsigned char* sign_mask, eve_stream_t stream) {
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    if (((unsigned)dtype > (unsigned)EVE_DT_F16) || N <= 0 || HW <= 0 || C <= 0 || C % vec || !x || !y ||
        !mean_rstd || ((gamma == nullptr) != (beta == nullptr)))
        return set_error_msg("instnorm_fwd_fused: bad arguments");
    int threads, vpt, sl;
    if (!fused_plan(HW * (C / vec), C / vec, threads, vpt, sl, dtype != EVE_DT_F32, gamma == nullptr)) return -1;
    const unsigned grid = sl ? (unsigned)((N + 7) / 8) * (8u << sl) : (unsigned)N;
    hipStream_t s = (hipStream_t)stream;
    // the trunk's instances: no affine, identity / ReLU, the plane part fills its <= 512-thread workgroup exactly
    if (g_cfg.in_trunk_kernels && vpt <= 8 && !gamma && (long long)threads * vpt == ((long long)HW * (C / vec)) >> sl) {
        const int combo = (act == EVE_ACT_RELU && !res && !sign_mask) ? 0 : (act == EVE_ACT_RELU && res && sign_mask) ? 1
                        : (act == EVE_ACT_NONE && !res && !sign_mask) ? 2 : -1;
#define FWD_TRUNK(T, TS, V)                                                                                                              \
        do {                                                                                                                             \
            if (combo == 0) EVE_LAUNCH("in_fwd_trunk_kernel<" TS ", " #V ", 1, false, false>", (in_fwd_trunk_kernel<T, V, EVE_ACT_RELU, false, false>), dim3(grid), dim3(threads), 0, s, (const T*)x, (const T*)res, (T*)y, mean_rstd, sign_mask, N, HW, C, sl, eps); \
            else if (combo == 1) EVE_LAUNCH("in_fwd_trunk_kernel<" TS ", " #V ", 1, true, true>", (in_fwd_trunk_kernel<T, V, EVE_ACT_RELU, true, true>), dim3(grid), dim3(threads), 0, s, (const T*)x, (const T*)res, (T*)y, mean_rstd, sign_mask, N, HW, C, sl, eps); \
            else EVE_LAUNCH("in_fwd_trunk_kernel<" TS ", " #V ", 0, false, false>", (in_fwd_trunk_kernel<T, V, EVE_ACT_NONE, false, false>), dim3(grid), dim3(threads), 0, s, (const T*)x, (const T*)res, (T*)y, mean_rstd, sign_mask, N, HW, C, sl, eps); \
        } while (0)
#define FWD_TRUNK_V(T, TS)                                                                                                               \
        do { switch (vpt) { case 1: FWD_TRUNK(T, TS, 1); break; case 2: FWD_TRUNK(T, TS, 2); break; case 4: FWD_TRUNK(T, TS, 4); break; default: FWD_TRUNK(T, TS, 8); break; } } while (0)
        if (combo >= 0) {
            if (dtype == EVE_DT_BF16) FWD_TRUNK_V(bf16_t, "eve::bf16_t");
            else if (dtype == EVE_DT_F16) FWD_TRUNK_V(f16_t, "eve::f16_t");
            else FWD_TRUNK_V(float, "float");
            EVE_CHECK_LAUNCH();
            return 0;
        }
#undef FWD_TRUNK_V
#undef FWD_TRUNK
    }
    if (dtype == EVE_DT_BF16) {
        LAUNCH_VPT(in_fwd_fused_kernel, bf16_t, "eve::bf16_t", (const bf16_t*)x, gamma, beta, (const bf16_t*)res, act, (bf16_t*)y,
                   mean_rstd, sign_mask, N, HW, C, sl, eps)
    } else if (dtype == EVE_DT_F16) {
        LAUNCH_VPT(in_fwd_fused_kernel, f16_t, "eve::f16_t", (const f16_t*)x, gamma, beta, (const f16_t*)res, act, (f16_t*)y,
                   mean_rstd, sign_mask, N, HW, C, sl, eps)
    } else {
        LAUNCH_VPT(in_fwd_fused_kernel, float, "float", (const float*)x, gamma, beta, (const float*)res, act, (float*)y,
                   mean_rstd, sign_mask, N, HW, C, sl, eps)
    }
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_instnorm_bwd_fused(int dtype, int N, int HW, int C, const void* dy, const void* dy2, const void* y, const void* x,
                                      const float* mean_rstd, const float* gamma, const float* beta, int act, void* dx, void* dres,
                                      float* sums, const unsigned char* sign_mask, const void* dx_add, eve_stream_t stream) {
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    if (((unsigned)dtype > (unsigned)EVE_DT_F16) || N <= 0 || HW <= 0 || C <= 0 || C % vec || !dy || !x ||
        !mean_rstd || !dx || (act != EVE_ACT_NONE && !y && gamma && !beta && !(act == EVE_ACT_RELU && sign_mask)))
        return set_error_msg("instnorm_bwd_fused: bad arguments");
    int threads, vpt, sl;
    if (!fused_plan(HW * (C / vec), C / vec, threads, vpt, sl, dtype != EVE_DT_F32, gamma == nullptr)) return -1;
    const unsigned grid = sl ? (unsigned)((N + 7) / 8) * (8u << sl) : (unsigned)N;
    hipStream_t s = (hipStream_t)stream;
    if (g_cfg.in_trunk_kernels && vpt <= 8 && !gamma && !dx_add && (long long)threads * vpt == ((long long)HW * (C / vec)) >> sl) {
        // 0: mid-block (ReLU' from x)   1 / 2: block end (mask, dres) without / with a second summand   3: down-sample branch
        const int combo = (act == EVE_ACT_RELU && !y && !sign_mask && !dres && !dy2) ? 0
                        : (act == EVE_ACT_RELU && sign_mask && dres) ? (dy2 ? 2 : 1)
                        : (act == EVE_ACT_NONE && !dres && !dy2) ? 3 : -1;
#define BWD_TRUNK_ARGS(T) (const T*)dy, (const T*)dy2, (const T*)x, mean_rstd, (T*)dx, (T*)dres, sums, sign_mask, N, HW, C, sl
#define BWD_TRUNK(T, TS, V)                                                                                                              \
        do {                                                                                                                             \
            if (combo == 0) EVE_LAUNCH("in_bwd_trunk_kernel<" TS ", " #V ", 0, false>", (in_bwd_trunk_kernel<T, V, 0, false>), dim3(grid), dim3(threads), 0, s, BWD_TRUNK_ARGS(T)); \
            else if (combo == 1) EVE_LAUNCH("in_bwd_trunk_kernel<" TS ", " #V ", 1, false>", (in_bwd_trunk_kernel<T, V, 1, false>), dim3(grid), dim3(threads), 0, s, BWD_TRUNK_ARGS(T)); \
            else if (combo == 2) EVE_LAUNCH("in_bwd_trunk_kernel<" TS ", " #V ", 1, true>", (in_bwd_trunk_kernel<T, V, 1, true>), dim3(grid), dim3(threads), 0, s, BWD_TRUNK_ARGS(T)); \
            else EVE_LAUNCH("in_bwd_trunk_kernel<" TS ", " #V ", 2, false>", (in_bwd_trunk_kernel<T, V, 2, false>), dim3(grid), dim3(threads), 0, s, BWD_TRUNK_ARGS(T)); \
        } while (0)
#define BWD_TRUNK_V(T, TS)                                                                                                               \
        do { switch (vpt) { case 1: BWD_TRUNK(T, TS, 1); break; case 2: BWD_TRUNK(T, TS, 2); break; case 4: BWD_TRUNK(T, TS, 4); break; default: BWD_TRUNK(T, TS, 8); break; } } while (0)
        if (combo >= 0) {
            if (dtype == EVE_DT_BF16) BWD_TRUNK_V(bf16_t, "eve::bf16_t");
            else if (dtype == EVE_DT_F16) BWD_TRUNK_V(f16_t, "eve::f16_t");
            else BWD_TRUNK_V(float, "float");
            EVE_CHECK_LAUNCH();
            return 0;
        }
#undef BWD_TRUNK_V
#undef BWD_TRUNK
#undef BWD_TRUNK_ARGS
    }
    if (dtype == EVE_DT_BF16) {
        LAUNCH_VPT(in_bwd_fused_kernel, bf16_t, "eve::bf16_t", (const bf16_t*)dy, (const bf16_t*)dy2, (const bf16_t*)y, (const bf16_t*)x, mean_rstd, gamma, beta,
                   act, (bf16_t*)dx, (bf16_t*)dres, sums, sign_mask, (const bf16_t*)dx_add, N, HW, C, sl)
    } else if (dtype == EVE_DT_F16) {
        LAUNCH_VPT(in_bwd_fused_kernel, f16_t, "eve::f16_t", (const f16_t*)dy, (const f16_t*)dy2, (const f16_t*)y, (const f16_t*)x, mean_rstd, gamma, beta,
                   act, (f16_t*)dx, (f16_t*)dres, sums, sign_mask, (const f16_t*)dx_add, N, HW, C, sl)
    } else {
        LAUNCH_VPT(in_bwd_fused_kernel, float, "float", (const float*)dy, (const float*)dy2, (const float*)y, (const float*)x, mean_rstd, gamma, beta,
                   act, (float*)dx, (float*)dres, sums, sign_mask, (const float*)dx_add, N, HW, C, sl)
    }
    EVE_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================
// ResNet stem tail: InstanceNorm (no affine) -> ReLU -> max-pool 3x3/2 pad 1 in ONE pass over the conv output
// (torchvision ResNet._forward_impl bn1/relu/maxpool, reference src/models/eye_net.py:48-50,106).
// relu(IN(.)) is monotone in the raw value (rstd > 0), so the window arg-max is taken on raw values and only
// the winner is normalised.  The 64x64x64 normalised tensor (1 GB per step at N=1920) is never materialised.
// Backward uses y = xhat wherever y > 0:  sum g = sum d*[y>0],  sum g*xhat = sum d*y  over the POOLED tensors,
// then one dense pass writes d(conv out) = rstd * (g - mean g - xhat * mean(g xhat)).
// =================================================================================================
namespace eve {

template <typename T>
__global__ __launch_bounds__(256) void in_relu_pool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ mr,
                                                               T* __restrict__ y, uint8_t* __restrict__ idx,
                                                               int IH, int IW, int OH, int OW, int C, long long items) {
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
                    if (f[e] > best[e] || bi[e] == 0xffu) { best[e] = f[e]; bi[e] = kh * 3 + kw; }
            }
        }
        const float* m = mr + ((size_t)n * C + cv * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float z = (best[e] - m[2 * e]) * m[2 * e + 1];
            best[e] = z > 0.f ? z : 0.f;
        }
        reinterpret_cast<uint4*>(y)[i] = Elem<T>::pack(best);
        // the window positions of the VEC channels as ONE store (round 5: eight single-byte stores per thread before)
        uint32_t w0 = 0u, w1 = 0u;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if (e < 4) w0 |= (bi[e] & 0xffu) << (8 * e);
            else       w1 |= (bi[e] & 0xffu) << (8 * (e - 4));
        }
        if (VEC == 8) *reinterpret_cast<uint2*>(idx + i * VEC) = make_uint2(w0, w1);
        else          *reinterpret_cast<uint32_t*>(idx + i * VEC) = w0;
    }
}

// one workgroup per image: pass 1 over the pooled tensors (sums), pass 2 dense over the conv output
template <typename T>
__global__ __launch_bounds__(1024) void in_relu_pool_bwd_kernel(const T* __restrict__ dyp, const T* __restrict__ yp,
                                                               const uint8_t* __restrict__ idx,
                                                               const T* __restrict__ x, const float* __restrict__ mr,
                                                               T* __restrict__ dx, int IH, int IW, int OH, int OW,
                                                               int C) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int NT = 1024;             // 16 waves per image: the dense pass is latency-bound, parallelism pays
    __shared__ float sh[2 * NT * VEC];
    __shared__ float sh_tot[2 * 1024];
    const int cvecs = C / VEC, phases = NT / cvecs;
    const int tid = threadIdx.x, cv = tid % cvecs, ph = tid / cvecs;
    const bool on = ph < phases;
    const size_t n = blockIdx.x;
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (on)
        for (int px = ph; px < OH * OW; px += phases) {
            const size_t o = (n * OH * OW + px) * C + cv * VEC;
            float d[VEC], yy[VEC];
            Elem<T>::unpack(*reinterpret_cast<const uint4*>(dyp + o), d);
            Elem<T>::unpack(*reinterpret_cast<const uint4*>(yp + o), yy);
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (yy[e] > 0.f) { s1[e] += d[e]; s2[e] += d[e] * yy[e]; }
        }
    if (on) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            sh[(ph * cvecs + cv) * VEC + e] = s1[e];
            sh[NT * VEC + (ph * cvecs + cv) * VEC + e] = s2[e];
        }
    }
    __syncthreads();
    if (on && ph == 0) {
        for (int q = 1; q < phases; ++q)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                s1[e] += sh[(q * cvecs + cv) * VEC + e];
                s2[e] += sh[NT * VEC + (q * cvecs + cv) * VEC + e];
            }
#pragma unroll
        for (int e = 0; e < VEC; ++e) { sh_tot[cv * VEC + e] = s1[e]; sh_tot[1024 + cv * VEC + e] = s2[e]; }
    }
    __syncthreads();
    if (!on) return;
    float mean[VEC], rstd[VEC];
    const float inv = 1.f / (float)(IH * IW);
    {
        const float* m = mr + (n * C + cv * VEC) * 2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            mean[e] = m[2 * e]; rstd[e] = m[2 * e + 1];
            s1[e] = sh_tot[cv * VEC + e] * inv; s2[e] = sh_tot[1024 + cv * VEC + e] * inv;
        }
    }
    // dense pass.  Even plane sizes (every stem): 2 x 2 blocks of input pixels (2a + r, 2b + c) -- the block touches exactly the
    // four windows (a + i, b + j), shared by its pixels (pixel (r, c) lies in window (i, j) iff i <= r and j <= c, at filter
    // position (r - 2i + 1, c - 2j + 1)): 4 + 12 loads per four outputs where the per-pixel form below issues 13 per output
    // (round 5: 3.5 ms of a configs[4] step at 2.4 TB/s, bound by load instructions, not bytes)
    if (((IH | IW) & 1) == 0 && OH * 2 == IH && OW * 2 == IW) {
        const int BW = IW >> 1, nblk = (IH >> 1) * BW;
        for (int blk = ph; blk < nblk; blk += phases) {
            const int a = blk / BW, b = blk - a * BW;
            uint4 qx[2][2], qd[2][2], qy[2][2];
            uint2 qi[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    qx[r][c] = *reinterpret_cast<const uint4*>(x + ((n * IH + 2 * a + r) * IW + 2 * b + c) * C + cv * VEC);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int oh = a + i < OH ? a + i : a, ow = b + j < OW ? b + j : b;
                    const size_t o = ((n * OH + oh) * OW + ow) * C + cv * VEC;
                    qd[i][j] = *reinterpret_cast<const uint4*>(dyp + o);
                    qy[i][j] = *reinterpret_cast<const uint4*>(yp + o);
                    if (VEC == 8) qi[i][j] = *reinterpret_cast<const uint2*>(idx + o);
                    else          qi[i][j] = make_uint2(*reinterpret_cast<const uint32_t*>(idx + o), 0u);
                }
            float gd[2][2][VEC];
            uint32_t id[2][2][VEC];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool live = a + i < OH && b + j < OW;
                    float d[VEC], yy[VEC];
                    Elem<T>::unpack(qd[i][j], d);
                    Elem<T>::unpack(qy[i][j], yy);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const uint32_t w32 = e < 4 ? qi[i][j].x : qi[i][j].y;
                        id[i][j][e] = (w32 >> (8 * (e & 3))) & 0xffu;
                        gd[i][j][e] = (live && yy[e] > 0.f) ? d[e] : 0.f;
                    }
                }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float g[VEC], xx[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) g[e] = 0.f;
#pragma unroll
                    for (int i = 0; i <= r; ++i)
#pragma unroll
                        for (int j = 0; j <= c; ++j) {
                            const uint32_t code = (uint32_t)((r - 2 * i + 1) * 3 + (c - 2 * j + 1));
#pragma unroll
                            for (int e = 0; e < VEC; ++e) g[e] += id[i][j][e] == code ? gd[i][j][e] : 0.f;
                        }
                    Elem<T>::unpack(qx[r][c], xx);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) g[e] = rstd[e] * (g[e] - s1[e] - (xx[e] - mean[e]) * rstd[e] * s2[e]);
                    *reinterpret_cast<uint4*>(dx + ((n * IH + 2 * a + r) * IW + 2 * b + c) * C + cv * VEC) = Elem<T>::pack(g);
                }
        }
        return;
    }
    // odd sizes: per pixel, written branch-free so all loads of a pixel are in flight together: an input pixel belongs to
    // 1 (even coordinate) or 2 (odd coordinate) windows per axis
    for (int px = ph; px < IH * IW; px += phases) {
        const int ih = px / IW, iw = px - ih * IW;
        const int oh0 = ih >> 1, oh1 = (ih + 1) >> 1, ow0 = iw >> 1, ow1 = (iw + 1) >> 1;
        const int ohs[2] = {oh0, oh1 < OH ? oh1 : oh0}, ows[2] = {ow0, ow1 < OW ? ow1 : ow0};
        const bool vh[2] = {true, oh1 != oh0 && oh1 < OH}, vw[2] = {true, ow1 != ow0 && ow1 < OW};
        const size_t xo = (n * IH * IW + px) * C + cv * VEC;
        const uint4 qx = *reinterpret_cast<const uint4*>(x + xo);
        uint4 qd[4], qy[4];
        uint2 qi[4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const size_t o = ((n * OH + ohs[a]) * OW + ows[b2]) * C + cv * VEC;
                qd[a * 2 + b2] = *reinterpret_cast<const uint4*>(dyp + o);
                qy[a * 2 + b2] = *reinterpret_cast<const uint4*>(yp + o);
                if (VEC == 8) qi[a * 2 + b2] = *reinterpret_cast<const uint2*>(idx + o);
                else          qi[a * 2 + b2] = make_uint2(*reinterpret_cast<const uint32_t*>(idx + o), 0u);
            }
        float g[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) g[e] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const uint32_t code = (uint32_t)((ih - (ohs[a] * 2 - 1)) * 3 + (iw - (ows[b2] * 2 - 1)));
                const bool live = vh[a] && vw[b2];
                float d[VEC], yy[VEC];
                Elem<T>::unpack(qd[a * 2 + b2], d);
                Elem<T>::unpack(qy[a * 2 + b2], yy);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const uint32_t w32 = e < 4 ? qi[a * 2 + b2].x : qi[a * 2 + b2].y;
                    const uint32_t id = (w32 >> (8 * (e & 3))) & 0xffu;
                    g[e] += (live && id == code && yy[e] > 0.f) ? d[e] : 0.f;
                }
            }
        float xx[VEC];
        Elem<T>::unpack(qx, xx);
#pragma unroll
        for (int e = 0; e < VEC; ++e) g[e] = rstd[e] * (g[e] - s1[e] - (xx[e] - mean[e]) * rstd[e] * s2[e]);
        *reinterpret_cast<uint4*>(dx + xo) = Elem<T>::pack(g);
    }
}

}  // namespace eve

extern "C" int eve_in_relu_maxpool_fwd(int dtype, int N, int IH, int IW, int C, const void* x, const float* mean_rstd,
                                       void* y, uint8_t* idx, eve_stream_t stream) {
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    if (((unsigned)dtype > (unsigned)EVE_DT_F16) || N <= 0 || IH <= 0 || IW <= 0 || C <= 0 || C % vec || !x ||
        !mean_rstd || !y || !idx)
        return set_error_msg("in_relu_maxpool_fwd: bad arguments");
    const int OH = (IH - 1) / 2 + 1, OW = (IW - 1) / 2 + 1;
    const long long items = (long long)N * OH * OW * (C / vec);
    long long blocks = (items + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(in_relu_pool_fwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)x, mean_rstd, (bf16_t*)y, idx, IH, IW, OH, OW, C, items);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(in_relu_pool_fwd_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const f16_t*)x, mean_rstd, (f16_t*)y, idx, IH, IW, OH, OW, C, items);
    else                      hipLaunchKernelGGL(in_relu_pool_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x, mean_rstd, (float*)y, idx, IH, IW, OH, OW, C, items);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_in_relu_maxpool_bwd(int dtype, int N, int IH, int IW, int C, const void* dy_pool, const void* y_pool,
                                       const uint8_t* idx, const void* x, const float* mean_rstd, void* dx,
                                       eve_stream_t stream) {
    const int vec = dtype != EVE_DT_F32 ? 8 : 4;
    if (((unsigned)dtype > (unsigned)EVE_DT_F16) || N <= 0 || IH <= 0 || IW <= 0 || C <= 0 || C % vec ||
        C / vec > 256 || C > 1024 || !dy_pool || !y_pool || !idx || !x || !mean_rstd || !dx)
        return set_error_msg("in_relu_maxpool_bwd: bad arguments");
    const int OH = (IH - 1) / 2 + 1, OW = (IW - 1) / 2 + 1;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EVE_DT_BF16) hipLaunchKernelGGL(in_relu_pool_bwd_kernel<bf16_t>, dim3(N), dim3(1024), 0, s, (const bf16_t*)dy_pool, (const bf16_t*)y_pool, idx, (const bf16_t*)x, mean_rstd, (bf16_t*)dx, IH, IW, OH, OW, C);
    else if (dtype == EVE_DT_F16) hipLaunchKernelGGL(in_relu_pool_bwd_kernel<f16_t>, dim3(N), dim3(1024), 0, s, (const f16_t*)dy_pool, (const f16_t*)y_pool, idx, (const f16_t*)x, mean_rstd, (f16_t*)dx, IH, IW, OH, OW, C);
    else                      hipLaunchKernelGGL(in_relu_pool_bwd_kernel<float>, dim3(N), dim3(1024), 0, s, (const float*)dy_pool, (const float*)y_pool, idx, (const float*)x, mean_rstd, (float*)dx, IH, IW, OH, OW, C);
    EVE_CHECK_LAUNCH();
    return 0;
}
