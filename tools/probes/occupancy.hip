// Resident workgroups per CU of the convolution / weight-gradient kernels (hipOccupancyMaxActiveBlocksPerMultiprocessor):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I eve_amd/csrc -I include tools/probes/occupancy.hip eve_amd/csrc/api.hip -o tools/probes/occupancy
#include "../../eve_amd/csrc/conv_igemm.hip"
#include <cstdio>

template <typename K>
static void show(const char* name, K kernel, int threads, size_t lds) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int n = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds);
    printf("%-44s threads %4d dyn LDS %6zu -> %d workgroups per CU (%s)\n", name, threads, lds, n, hipGetErrorString(e));
}

int main() {
    using namespace eve;
    show("igemm_dma_kernel<bf16, 2, 2>", igemm_dma_kernel<bf16_t, 2, 2>, 256, 0);
    show("igemm_dma_kernel<bf16, 4, 1>", igemm_dma_kernel<bf16_t, 4, 1>, 256, 0);
    show("igemm_dma_kernel<float, 2, 2>", igemm_dma_kernel<float, 2, 2>, 256, 0);
    show("conv3x3_halo_kernel<2, 2> (l2: 56 KB)", conv3x3_halo_kernel<2, 2>, 256, 2 * 3 * 4096 + 4 * 8192);
    show("conv3x3_halo_kernel<2, 2> (l4: 72 KB)", conv3x3_halo_kernel<2, 2>, 256, 2 * 5 * 4096 + 4 * 8192);
    show("conv3x3_halo_pkernel<4, 1> (60 KB)", conv3x3_halo_pkernel<4, 1>, 256, 60 * 1024);
    show("conv3x3_halo_pkernel<4, 1> (76 KB)", conv3x3_halo_pkernel<4, 1>, 256, 76 * 1024);
    show("wgrad_tr_kernel<2, 2, 1, false, 4> (48 KB)", wgrad_tr_kernel<2, 2, 1, false, 4>, 256, 3 * 32 * 256 * 2);
    show("wgrad_tr_kernel<1, 3, 1, false, 4> (64 KB)", wgrad_tr_kernel<1, 3, 1, false, 4>, 192, 4 * 32 * 256 * 2);
    show("wgrad_tr_kernel<1, 4, 1, false, 4> (80 KB)", wgrad_tr_kernel<1, 4, 1, false, 4>, 256, 4 * 32 * 320 * 2);
    show("wgrad_tr_kernel<1, 4, 2, true, 4> (60 KB)", wgrad_tr_kernel<1, 4, 2, true, 4>, 256, 3 * 32 * 320 * 2);
    show("wgrad_halo64_kernel (149 KB)", wgrad_halo64_kernel, 512, 152576);
    show("wgrad_halo_kernel<1, 4, 3> (52 KB)", wgrad_halo_kernel<1, 4, 3>, 256, 52 * 1024);
    return 0;
}
