// Micro-benchmark: the main loop of a 256 x 256 x 32 macro-tile on v_mfma_f32_32x32x16_bf16 -- four waves, ONE per SIMD,
// 128 x 128 per wave = 16 accumulator tiles (256 AGPRs), per K step 16 ds_read_b128 + 32 MFMAs + one s_barrier, optionally
// the LDS-DMA stream of the step (4 weight pieces + 1 halo piece per thread) with a counted wait.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/macro_tile.hip -o tools/probes/macro_tile
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef int int4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LDSAS __attribute__((address_space(3)))

__device__ __forceinline__ void dma16(const int4v& rsrc, uint32_t lds, int voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds), "v"(voff), "s"(rsrc) : "memory", "m0");
}
__device__ __forceinline__ bf16x8 ldsr(uint32_t a) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const LDSAS u32x4*>((uintptr_t)a)); }

template <int MODE>   // 0: reads + MFMA + barrier   1: + DMA stream and counted waits
__global__ __launch_bounds__(256, 1) void tile_kernel(const char* __restrict__ src, uint32_t bytes, int steps, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const uint32_t lds0 = (uint32_t)(uintptr_t)((LDSAS void*)smem);
    for (int i = tid; i < 96 * 1024 / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    int4v rs; const uint64_t a = (uint64_t)src;
    rs.x = (int)(uint32_t)a; rs.y = (int)(uint32_t)((a >> 32) & 0xffff); rs.z = (int)bytes; rs.w = 0x00020000;
    // halo plane: 24 KB at 0; weight ring: 4 x 16 KB at 32 KB
    uint32_t aaddr[4][2], baddr[4][2];
    for (int t = 0; t < 4; ++t)
        for (int kh = 0; kh < 2; ++kh) {
            const int m = wm * 128 + t * 32 + li, c = wn * 128 + t * 32 + li;
            aaddr[t][kh] = lds0 + ((m + 19) << 6) + ((((2 * kh + lh)) ^ ((m >> 2) & 3)) << 4);
            baddr[t][kh] = lds0 + 32768 + (c << 6) + ((((2 * kh + lh)) ^ ((c >> 2) & 3)) << 4);
        }
    int woff[4];
    for (int j = 0; j < 4; ++j) { const int L = tid + 256 * j; woff[j] = (L >> 2) * 4608 + (L & 3) * 16; }
    __syncthreads();
    f32x16 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 fx0[4], fw0[4], fx1[4], fw1[4];
    for (int t = 0; t < 4; ++t) { fx0[t] = ldsr(aaddr[t][0]); fw0[t] = ldsr(baddr[t][0]); }
    if (MODE == 1) for (int pre = 0; pre < 3; ++pre) for (int j = 0; j < 4; ++j) dma16(rs, lds0 + 32768 + pre * 16384 + wave * 1024 + j * 4096, woff[j] + pre * 64);
    for (int s = 0; s < steps; ++s) {
        const int tapoff = (s % 9) * 64;                     // tap shift of the halo reads
        if (MODE == 1) { if (steps & 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (MODE == 1) {
            const uint32_t dst = lds0 + 32768 + ((s + 3) & 3) * 16384 + wave * 1024;
            const int strm = (steps & 2) ? ((s + 3) / 72) * (256 * 4608) + ((s + 3) % 72) * 64 : ((s + 3) % 72) * 64;   // cold: a new 1.2 MB bank every 72 steps
            for (int j = 0; j < 4; ++j) dma16(rs, dst + j * 4096, woff[j] + strm);
            if ((s % 9) < 6) dma16(rs, lds0 + ((s % 9) * 4096 & 0x3fff) + 98304 - 16384 + wave * 1024, woff[0] + 1 << 20);
        }
        const uint32_t bslot = ((s & 3) * 16384);
#pragma unroll
        for (int t = 0; t < 4; ++t) { fx1[t] = ldsr(aaddr[t][1] + tapoff); fw1[t] = ldsr(baddr[t][1] + bslot); }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw0[nt], fx0[mt], acc[mt][nt], 0, 0, 0);
        const uint32_t bslot2 = (((s + 1) & 3) * 16384);
        const int tapoff2 = ((s + 1) % 9) * 64;
#pragma unroll
        for (int t = 0; t < 4; ++t) { fx0[t] = ldsr(aaddr[t][0] + tapoff2); fw0[t] = ldsr(baddr[t][0] + bslot2); }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw1[nt], fx1[mt], acc[mt][nt], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
    out[blockIdx.x * 256 + tid] = sum;
}

int main() {
    const uint32_t bytes = 64u << 20;
    char* d; float* o;
    hipMalloc(&d, bytes); hipMemset(d, 0x3c, bytes); hipMalloc(&o, 1024 * 256 * 4);
    hipFuncSetAttribute((const void*)tile_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)tile_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int steps : {1440, 1441, 1442, 1443})
    for (int mode = 0; mode < 2; ++mode)
        for (int blocks : {256, 512}) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(tile_kernel<0>, dim3(blocks), dim3(256), 98304, 0, d, bytes, steps, o);
                else           hipLaunchKernelGGL(tile_kernel<1>, dim3(blocks), dim3(256), 98304, 0, d, bytes, steps, o);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double flop = 2.0 * 256 * 256 * 32 * (double)steps * blocks;
            printf("macro tile 256x256x32, 4 waves, %s%s: %4d WGs  %.3f ms  %.0f TFLOP/s  (%.0f cycles per step at 2.0 GHz)\n",
                   mode ? "reads + MFMA + barrier + LDS-DMA stream" : "reads + MFMA + barrier", (mode && (steps & 1)) ? ((steps & 2) ? " [ONE tile in flight, COLD stream]" : " [ONE tile in flight]") : (mode ? ((steps & 2) ? " [two tiles in flight, COLD stream]" : " [two tiles in flight]") : ""), blocks, best, flop / best / 1e9,
                   best * 1e-3 * 2.0e9 / steps / (blocks / 256));
        }
    return 0;
}
