// Hardware probes for gfx950 semantics the kernels rely on (run on the GPU box; prints tables).
//   1. ds_read_b64_tr_b16: which (lane, element) of the per-lane 8-byte loads ends up where
//   2. raw.buffer.load.lds (LDS-DMA): lane-linear placement and zero-fill for out-of-range voffset
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_t;
#define LDS_AS __attribute__((address_space(3)))

__global__ void probe_tr(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 4];
    const int l = threadIdx.x;
    // lane l's private 8-byte chunk holds the codes (l << 2 | e), e = 0..3
    for (int e = 0; e < 4; ++e) lds[l * 4 + e] = (uint16_t)((l << 2) | e);
    __syncthreads();
    auto p = (LDS_AS bf16x4_t*)(LDS_AS void*)(&lds[l * 4]);
    bf16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p);
    uint64_t bits = __builtin_bit_cast(uint64_t, r);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (uint16_t)(bits >> (16 * e));
}


__global__ void probe_dma(const uint32_t* src, int nbytes, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * 256];
    const int l = threadIdx.x;
    for (int i = l; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // lanes read 16 B each from a permuted source; odd lanes >= 32 are sent out of range
    int voff = ((l * 7) % 64) * 16;
    if (l >= 32 && (l & 1)) voff = nbytes;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)(&lds[0]), 16, voff, 0, 0, 0);
    // second DMA into the second KiB with a wave-uniform LDS base
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)(&lds[256]), 16, l * 16, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = l; i < 512; i += 64) out[i] = lds[i];
}

int main() {
    uint16_t* d16; hipMalloc(&d16, 256 * 2);
    probe_tr<<<1, 64>>>(d16);
    uint16_t h16[256]; hipMemcpy(h16, d16, 512, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16: result[lane][elem] = (src_lane, src_elem)\n");
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) {
            int sl = h16[l * 4 + e] >> 2, se = h16[l * 4 + e] & 3;
            printf(" (%2d,%d)", sl, se);
            int i = l & 15, g = l >> 4;
            if (sl != g * 16 + 4 * e + (i >> 2) || se != (i & 3)) ok = 0;
        }
        printf("\n");
    }
    printf("TR_MODEL result(i,j)=loaded(16g+4j+i/4, i%%4): %s\n", ok ? "CONFIRMED" : "MISMATCH");

    uint32_t hsrc[256]; for (int i = 0; i < 256; ++i) hsrc[i] = 0x1000 + i;
    uint32_t *dsrc, *dout; hipMalloc(&dsrc, 1024); hipMalloc(&dout, 2048);
    hipMemcpy(dsrc, hsrc, 1024, hipMemcpyHostToDevice);
    probe_dma<<<1, 64>>>(dsrc, 1024, dout);
    uint32_t ho[512]; hipMemcpy(ho, dout, 2048, hipMemcpyDeviceToHost);
    int ok2 = 1;
    for (int l = 0; l < 64; ++l) {
        int srcv = (l * 7) % 64; bool oob = l >= 32 && (l & 1);
        for (int e = 0; e < 4; ++e) {
            uint32_t want = oob ? 0u : 0x1000 + srcv * 4 + e;
            if (ho[l * 4 + e] != want) { ok2 = 0; printf("dma1 lane %d e %d: got %x want %x\n", l, e, ho[l * 4 + e], want); }
            if (ho[256 + l * 4 + e] != 0x1000u + l * 4 + e) { ok2 = 0; printf("dma2 lane %d e %d: got %x\n", l, e, ho[256 + l * 4 + e]); }
        }
    }
    printf("LDS_DMA lane-linear + OOB zero-fill: %s\n", ok2 ? "CONFIRMED" : "MISMATCH");
    return 0;
}
