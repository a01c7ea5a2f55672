// Round-6 experiment: InstanceNorm from the PRODUCER's epilogue for the ResNet layers whose image lies inside one
// wave of conv3x3_wg8_kernel (layer 3: 8 x 8 x 256, layer 4: 4 x 4 x 512), against the shipped pair of launches
//     conv3x3_wg8_kernel  ->  eve_instnorm_fwd_fused (in_fwd_trunk_kernel: 1 read + 1 write of the plane)
// on the same box, same data:   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -I. tools/probes/in_epilogue.hip \
//                                   -Leve_amd/lib -leve_hip -Wl,-rpath,$PWD/eve_amd/lib -o gpurun_out/in_epilogue && gpurun_out/in_epilogue
// Prints ms per launch (HIP events, 50 launches each) and the largest deviation of the fused output / statistics from the pair's.
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

#include "../../eve_amd/csrc/conv_wg8.h"

using namespace eve;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int WM, int WN, int W>
static void run(int N, int C) {
    using G = Wg8Geom<WM, WN, W>;
    const size_t nx = (size_t)N * W * W * C, nw = (size_t)C * 9 * C;
    std::vector<uint16_t> hx(nx), hw(nw);
    srand(7);
    for (auto& v : hx) v = f2bf((float)rand() / RAND_MAX * 2.f - 1.f);
    for (auto& v : hw) v = f2bf(((float)rand() / RAND_MAX * 2.f - 1.f) * sqrtf(2.f / (9.f * C)));
    bf16_t *x, *w, *y, *z_pair, *z_fused;
    float *st_pair, *st_fused;
    CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&y, nx * 2)); CK(hipMalloc(&z_pair, nx * 2)); CK(hipMalloc(&z_fused, nx * 2));
    CK(hipMalloc(&st_pair, (size_t)N * C * 8)); CK(hipMalloc(&st_fused, (size_t)N * C * 8));
    CK(hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice));
    Wg8Params g;
    g.N = N; g.Cin = C; g.Cout = C; g.flip = 0; g.K = 9 * C; g.x_bytes = (uint32_t)(nx * 2); g.w_bytes = (uint32_t)(nw * 2); g.s2_py = 0;
    g.tiles_n = (uint32_t)(C / G::COUT_T);
    const uint32_t tiles = (uint32_t)((N + G::TI - 1) / G::TI) * g.tiles_n;
    auto kplain = conv3x3_wg8_kernel<bf16_t, WM, WN, W, 9, 1, false>;
    auto kfused = conv3x3_wg8_kernel<bf16_t, WM, WN, W, 9, 1, true>;
    CK(hipFuncSetAttribute((const void*)kplain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)kfused, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto pair = [&]() {
        hipLaunchKernelGGL(kplain, dim3(tiles), dim3(512), G::LDS, 0, g, (const bf16_t*)x, (const bf16_t*)w, (const float*)nullptr, EVE_ACT_NONE, y, (float*)nullptr, 1e-5f);
        if (eve_instnorm_fwd_fused(EVE_DT_BF16, N, W * W, C, y, nullptr, nullptr, nullptr, EVE_ACT_RELU, 1e-5f, z_pair, st_pair, nullptr, nullptr) != 0) {
            printf("eve_instnorm_fwd_fused: %s\n", eve_last_error()); exit(1);
        }
    };
    auto conv_only = [&]() {
        hipLaunchKernelGGL(kplain, dim3(tiles), dim3(512), G::LDS, 0, g, (const bf16_t*)x, (const bf16_t*)w, (const float*)nullptr, EVE_ACT_NONE, y, (float*)nullptr, 1e-5f);
    };
    auto fused = [&]() {
        hipLaunchKernelGGL(kfused, dim3(tiles), dim3(512), G::LDS, 0, g, (const bf16_t*)x, (const bf16_t*)w, (const float*)nullptr, EVE_ACT_RELU, z_fused, st_fused, 1e-5f);
    };
    auto timeit = [&](auto f) {
        for (int i = 0; i < 5; ++i) f();
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 50; ++i) f();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / 50.f;
    };
    const float t_conv = timeit(conv_only), t_pair = timeit(pair), t_fused = timeit(fused);
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> a(nx), b(nx);
    std::vector<float> sa((size_t)N * C * 2), sb((size_t)N * C * 2);
    CK(hipMemcpy(a.data(), z_pair, nx * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), z_fused, nx * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sa.data(), st_pair, sa.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(sb.data(), st_fused, sb.size() * 4, hipMemcpyDeviceToHost));
    double dmax = 0, dsum = 0, smax_m = 0, smax_r = 0;
    for (size_t i = 0; i < nx; ++i) { const double d = fabs((double)bf2f(a[i]) - bf2f(b[i])); dmax = d > dmax ? d : dmax; dsum += d; }
    for (size_t i = 0; i < sa.size(); i += 2) {
        smax_m = fmax(smax_m, fabs((double)sa[i] - sb[i]));
        smax_r = fmax(smax_r, fabs((double)sa[i + 1] - sb[i + 1]) / sa[i + 1]);
    }
    const double gf = 2.0 * N * W * W * C * 9.0 * C / 1e9;
    printf("%d x %d x %d x %d (N = %d): conv alone %.4f ms (%.0f TFLOP/s) | conv + InstanceNorm launch %.4f ms | conv with InstanceNorm epilogue %.4f ms"
           " | fused vs pair: max |dz| %.3g, mean %.3g; |d mean| %.3g, rel |d rstd| %.3g\n",
           W, W, C, C, N, t_conv, gf / t_conv, t_pair, t_fused, dmax, dsum / nx, smax_m, smax_r);
    hipFree(x); hipFree(w); hipFree(y); hipFree(z_pair); hipFree(z_fused); hipFree(st_pair); hipFree(st_fused);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1920;
    run<2, 4, 8>(N, 256);       // ResNet layer 3
    run<2, 4, 4>(N, 512);       // ResNet layer 4
    return 0;
}
