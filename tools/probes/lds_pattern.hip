// LDS read-pattern probe: cycles per ds_read_b128 for a few lane -> address maps (one wave per workgroup, and 16 waves per workgroup).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_pattern lds_pattern.hip && /tmp/lds_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* __restrict__ offs, int npat, unsigned long long* out, u32x4* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 40960 / 4; i += blockDim.x) ((uint32_t*)smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int p = 0; p < npat; ++p) {
        const uint32_t a = (uint32_t)(uintptr_t)smem + offs[p * 64 + lane];
        u32x4 acc = {0, 0, 0, 0};
        __syncthreads();
        const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int it = 0; it < 256; ++it) {
            u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1280\n\tds_read_b128 %2, %8 offset:2560\n\tds_read_b128 %3, %8 offset:3840\n\t"
                         "ds_read_b128 %4, %8 offset:5120\n\tds_read_b128 %5, %8 offset:6400\n\tds_read_b128 %6, %8 offset:7680\n\tds_read_b128 %7, %8 offset:8960\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
            acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) out[blockIdx.x * npat + p] = t1 - t0;
        if (acc.x == 0x12345678u) sink[0] = acc;
    }
}
int main() {
    const int NP = 6;
    int h[NP * 64];
    for (int l = 0; l < 64; ++l) {
        const int li = l & 15, lg = l >> 4;
        h[0 * 64 + l] = l * 16;                                   // linear: conflict-free
        h[1 * 64 + l] = 32 * li + 16 * lg;                        // stem x fragment (overlapping 16-byte pieces)
        h[2 * 64 + l] = li * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4);   // stem weight fragment
        h[3 * 64 + l] = 32 * li + 16 * (lg & 1) + 512 * (lg >> 1);       // x with the two upper K chunks from another row piece
        h[4 * 64 + l] = l * 32;                                   // stride 32: 2-way
        h[5 * 64 + l] = (l & 31) * 64 + (l >> 5) * 16;            // wg8 style
    }
    int* d; unsigned long long* o; u32x4* s;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 8 * NP * 512); hipMalloc(&s, 64);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int threads : {64, 256, 1024}) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(threads), 49152, 0, d, NP, o, s);
        hipDeviceSynchronize();
        unsigned long long r[NP * 256];
        hipMemcpy(r, o, 8 * NP * 256, hipMemcpyDeviceToHost);
        printf("threads %4d:", threads);
        for (int p = 0; p < NP; ++p) printf("  pat%d %.1f", p, (double)r[p] / (256.0 * 8));
        printf("   (clocks of s_memtime-class counter per ds_read_b128, wave 0 of block 0)\n");
    }
    return 0;
}
