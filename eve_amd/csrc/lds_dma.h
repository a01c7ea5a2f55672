// LDS-DMA helpers shared by the gfx950 kernels: buffer resources, LDS byte addresses and the
// `buffer_load ... lds` instruction (global -> LDS without passing through VGPRs; lane-linear placement,
// out-of-range lanes are zero-filled).
#pragma once
#include "common.h"

namespace eve {

#define EVE_LDS __attribute__((address_space(3)))
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4v_t;

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, const void* lds_generic_ptr, int voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (EVE_LDS void*)(lds_generic_ptr), 16, voffset, 0, 0, 0);
}
#define EVE_OOB ((int)0x80000000u)   // >= num_records for every tensor we accept (< 2^31 bytes)

// The same LDS-DMA as inline asm.  hipcc treats the builtin as a store to LDS and protects every later ds_read
// with s_waitcnt vmcnt(0), which drains a multi-stage prefetch ring at every step; an asm statement is opaque to
// that bookkeeping, so the waits are exactly the counted s_waitcnt vmcnt(N) the kernel places itself.
// M0 (the LDS destination base) is written inside the statement and declared clobbered.
//
// THE M0 CONTRACT.  hipcc warns "inline asm clobber list contains reserved registers: m0" for these statements: M0 is not
// allocatable, so the clobber is advisory, and what keeps the statements safe is a property of the CODE GENERATOR, not of
// the language: AMD clang (roc-7.2.0, clang 22) never keeps a value live in M0 across other code -- every M0 consumer it
// emits (LDS-DMA builtins, movrel, sendmsg) is preceded by its own `s_mov_b32 m0, ...` in the same basic block -- and none
// of the kernels that use the asm form contains such a compiler-generated consumer (they do not mix the builtin and the
// asm DMA).  Every asm statement here writes M0 in the SAME statement that reads it, so it depends on nothing outside.
// A different compiler must be re-verified (disassemble conv_igemm / stem_fused / cgru_scan and check that no instruction
// outside an ASMSTART/ASMEND pair reads m0 without its own s_mov; then run `pytest -m gpu`, whose parity tests cover every
// LDS-DMA kernel) before this guard is widened; -DEVE_M0_CONTRACT_VERIFIED overrides it for that experiment.
#if !defined(EVE_M0_CONTRACT_VERIFIED)
static_assert(__clang_major__ == 22, "lds_dma.h: the inline-asm M0 contract was verified for AMD clang 22 (ROCm 7.2) only -- "
                                     "re-verify it for this compiler (see the comment above), then extend this guard");
#endif
typedef int eve_int4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ eve_int4 make_rsrc_words(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    eve_int4 r;
    r.x = (int)(uint32_t)a;
    r.y = (int)(uint32_t)((a >> 32) & 0xffffu);      // stride 0
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* generic_ptr) {
    return (uint32_t)(uintptr_t)((EVE_LDS void*)generic_ptr);
}
__device__ __forceinline__ void lds_dma16_asm(const eve_int4& rsrc, uint32_t lds_byte_addr, int voffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(lds_byte_addr), "v"(voffset), "s"(rsrc)
                 : "memory", "m0");
}

}  // namespace eve
