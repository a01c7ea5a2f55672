// Micro-benchmark: sustained MFMA rate of the two bf16 shapes in the register arrangement of the conv kernels
// (64 accumulator registers per wave, operands changing every step, random data), 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, bool ROT>
__global__ __launch_bounds__(256) void rate_kernel(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 0xffff]; b[i] = in[(t * 8 + 4 + i) & 0xffff]; }
    if (SHAPE == 16) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            // rotate the operands so that nothing is loop-invariant
            if (ROT) { bf16x8 t0 = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = b[0]; b[0] = b[1]; b[1] = b[2]; b[2] = b[3]; b[3] = t0; }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        out[t] = s;
    } else {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            // the same 64x64x32 step: 2 x 2 tiles of 32x32, two K halves of 16
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * kh + i], b[2 * kh + j], acc[i][j], 0, 0, 0);
            if (ROT) { bf16x8 t0 = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = b[0]; b[0] = b[1]; b[1] = b[2]; b[2] = b[3]; b[3] = t0; }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[i][j][e];
        out[t] = s;
    }
}

int main() {
    const int iters = 4000;
    std::vector<unsigned short> h(65536 * 8);
    srand(1);
    for (auto& v : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    bf16x8* din; float* dout;
    hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 4096 * 256 * 4);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rot : {0, 1})
    for (int shape : {16, 32})
        for (int blocks : {256, 512, 1024, 2048}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (shape == 16 && !rot) hipLaunchKernelGGL((rate_kernel<16, false>), dim3(blocks), dim3(256), 0, 0, din, dout, iters);
                else if (shape == 16)    hipLaunchKernelGGL((rate_kernel<16, true>), dim3(blocks), dim3(256), 0, 0, din, dout, iters);
                else if (!rot)           hipLaunchKernelGGL((rate_kernel<32, false>), dim3(blocks), dim3(256), 0, 0, din, dout, iters);
                else                     hipLaunchKernelGGL((rate_kernel<32, true>), dim3(blocks), dim3(256), 0, 0, din, dout, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double flop = 2.0 * 64 * 64 * 32 * (double)iters * 4 * blocks;
            printf("mfma %dx%d%s: %4d blocks (%.1f waves/SIMD)  %.3f ms  %.0f TFLOP/s\n", shape, shape, rot ? " +36 v_mov/step" : "", blocks, blocks / 256.0, best, flop / best / 1e9);
        }
    return 0;
}
