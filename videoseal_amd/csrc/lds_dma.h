// LDS-DMA (global -> LDS without a VGPR round trip) helpers shared by the all-DMA / wave-specialised kernels.
//
// `global_load_lds_dwordx4` takes its LDS destination from M0 (wave-uniform base; lane l lands at base + 16 * l).  Two forms:
//
//  * vs_lds_dma16():            the compiler builtin.  hipcc owns M0 and tracks the transfer as a pending LDS write -- it places
//                               `s_waitcnt vmcnt(0)` in front of the next LDS read of the same array.  Right for kernels whose loop waits are
//                               the compiler's (conv3x3_pl, gemm_pl, conv3x3_patch_pc).
//  * vs_lds_dma16_untracked():  the same instruction written out, for loops whose vmcnt waits are counted by hand (cnx_pipe_kernel,
//                               gemm1x1_pc's consumer-side weight ring): a transfer hipcc cannot see does not get that wait.  hipcc's own
//                               vmcnt waits stay correct with unseen transfers in flight (vmcnt retires in order: an unseen younger transfer
//                               only makes a counted wait stricter).
//
// Round-6 hardening: the first version of the asm form wrote M0 with "m0" on the clobber list -- clang answers "clobber
// list contains reserved registers", i.e. the clobber is IGNORED, and a compiler that keeps a value of its own in M0 across the statement (the
// builtin form's base, a `v_readlane` / `s_movrel` index) would have read ours.  The asm block is now M0-NEUTRAL: it saves M0 into a scalar
// temporary the compiler allocates, sets it, issues the transfer (M0 is read at issue) and restores it -- whatever hipcc believed about M0
// before the statement is true after it, so both forms may be mixed in one kernel.  Two extra SALU moves per transfer.
// tests/test_codegen.py::test_untracked_lds_dma_is_m0_neutral checks the emitted ISA for exactly this bracket around every hand-written DMA.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void vs_lds_dma16(const char* gp, unsigned char* lds_base) {      // lane l: 16 bytes at gp -> lds_base + 16 * l
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (__attribute__((address_space(3))) void*)lds_base,
                                   16, 0, 0);
}

__device__ __forceinline__ void vs_lds_dma16_untracked(const char* gp, unsigned char* lds_base) {
  const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gp), "s"(m)
               : "memory");
}
