// Micro-benchmark: sustained LDS-DMA (buffer_load_dwordx4 ... lds) fill rate per CU from L2-resident data, in the
// access pattern of the conv kernels' weight stream (a tile = 128 rows x 64 B at a row stride of K*2 bytes, or contiguous),
// 2 workgroups of 256 threads per CU, ring of 4 tiles, at most `depth` tiles in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dma_rate.hip -o tools/probes/dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int int4v __attribute__((ext_vector_type(4)));
#define LDSAS __attribute__((address_space(3)))

__device__ __forceinline__ void dma16(const int4v& rsrc, uint32_t lds, int voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds), "v"(voff), "s"(rsrc) : "memory", "m0");
}
template <int MODE>   // 0: LDS-DMA   1: global_load_dwordx4 into registers (sum to keep it alive)
__global__ __launch_bounds__(256) void fill_kernel(const char* __restrict__ src, uint32_t bytes, int row_stride, int tiles,
                                                   int iters, int per_wg_offset, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int4v rs; const uint64_t a = (uint64_t)src;
    rs.x = (int)(uint32_t)a; rs.y = (int)(uint32_t)((a >> 32) & 0xffff); rs.z = (int)bytes; rs.w = 0x00020000;
    const uint32_t lds0 = (uint32_t)(uintptr_t)((LDSAS void*)smem);
    // a tile = 128 rows x 64 B; thread t fetches 16 B: row (t + 256 j) >> 2, chunk & 3  (2 pieces per thread = 8 KB per tile)
    int off[2];
    for (int j = 0; j < 2; ++j) { const int L = tid + 256 * j; off[j] = (L >> 2) * row_stride + (L & 3) * 16; }
    const int base0 = (blockIdx.x * per_wg_offset) % (int)(bytes / 2);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int tile_off = base0 + (it % tiles) * 64;     // next (tap, slice): 64 bytes further along every row
        if (MODE == 0) {
            const uint32_t dst = lds0 + (it & 3) * 8192 + wave * 1024;
            dma16(rs, dst, off[0] + tile_off);
            dma16(rs, dst + 4096, off[1] + tile_off);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // two tiles in flight
        } else {
            const float4 v0 = *reinterpret_cast<const float4*>(src + off[0] + tile_off);
            const float4 v1 = *reinterpret_cast<const float4*>(src + off[1] + tile_off);
            acc += v0.x + v1.y;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 1 || out == nullptr) out[blockIdx.x * 256 + tid] = acc;
}

int main() {
    const uint32_t bytes = 64u << 20;
    char* d; float* o;
    hipMalloc(&d, bytes); hipMemset(d, 1, bytes); hipMalloc(&o, 4096 * 256 * 4);
    hipFuncSetAttribute((const void*)fill_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    struct Cfg { const char* name; int row_stride, tiles, per_wg; } cfgs[] = {
        {"same tile stream for every WG, rows at stride 4608 B (layer-3 filter)", 4608, 72, 0},
        {"same stream, contiguous 8 KB tiles", 64, 1, 0},
        {"4 different channel blocks (WG % 4), stride 4608", 4608, 72, 128 * 4608},
        {"every WG its own 590 KB region (L2 / MALL mix)", 4608, 72, 128 * 4608 + 4096},
    };
    for (int mode = 0; mode < 2; ++mode)
        for (auto& c : cfgs)
            for (int blocks : {256, 512, 1024}) {
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(fill_kernel<0>, dim3(blocks), dim3(256), 32768, 0, d, bytes, c.row_stride, c.tiles, iters, c.per_wg, o);
                    else           hipLaunchKernelGGL(fill_kernel<1>, dim3(blocks), dim3(256), 32768, 0, d, bytes, c.row_stride, c.tiles, iters, c.per_wg, o);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                const double total = 8192.0 * iters * blocks;
                const int resident = blocks < 512 ? blocks : 512;
                printf("%s  %-72s %4d WGs  %.3f ms  %.2f TB/s  ~%.1f B/clk/CU (at 2.0 GHz)\n", mode ? "global_load->VGPR" : "LDS-DMA          ", c.name, blocks,
                       best, total / best / 1e9, total / best / 1e-3 / 256 / 2.0e9);
                (void)resident;
            }
    return 0;
}
