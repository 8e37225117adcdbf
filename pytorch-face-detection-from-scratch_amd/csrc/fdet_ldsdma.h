// LDS-DMA pieces issued from inline asm (buffer_load_dwordx4 ... lds: 64 lanes x 16 bytes = 1 KiB, LDS destination =
// M0 + lane*16, global source = descriptor base + scalar offset + per-lane byte offset).
//
// Why asm and not __builtin_amdgcn_global_load_lds: hipcc tracks the builtin as a vector-memory operation that WRITES
// LDS and, unable to prove that a later ds_read does not alias it, puts `s_waitcnt vmcnt(0)` in front of the first LDS
// read after every DMA issue -- the prefetch of the next chunk is then waited for before the current chunk's first
// MFMA (seen in the .s of both PS kernels; it cost the conv kernel its whole DMA / MFMA overlap).  An asm piece is
// invisible to that pass; it has NO register destination, so none of the hazards of asm loads into registers apply.
// Ordering is by hand, as cdna_hip_programming.md 5.7 / MI355X_MICROARCH.md item 7 prescribe: the issuing wave's counted
// `s_waitcnt vmcnt(N)`, then a workgroup barrier, then the ds_read.
//   * M0 is written in the same statement that uses it.  It is NOT on the clobber list: hipcc treats m0 as a reserved
//     register ("clobbering them may lead to undefined behaviour"; the generated code is identical with and without the
//     clobber).  Instead tests/test_asm_audit.py asserts, from the cross-compiled assembly of every kernel that includes
//     this header, that no compiler-generated instruction reads or writes m0;
//   * `s_nop 4` covers the SALU-write -> VMEM-read wait states of M0 / the scalar offset / the descriptor;
//   * compiler-issued loads / stores stay correct beside asm pieces as long as no piece is issued between such a load
//     and its first use (an OLDER piece only makes the compiler's counted wait stricter).
#pragma once
#include <hip/hip_runtime.h>

typedef unsigned dma_u32x4 __attribute__((ext_vector_type(4)));

// buffer descriptor over [base, base + bytes): raw buffer, out-of-range reads return zeros
__device__ __forceinline__ dma_u32x4 dma_rsrc(const void* base, unsigned bytes) {
  const unsigned long long pa = (unsigned long long)base;
  return dma_u32x4{(unsigned)pa, (unsigned)(pa >> 32), bytes, 0x00020000u};
}

// one piece: LDS bytes [lds_addr + 16*lane, +16) <- global bytes [base + soff + voff, +16)   (lds_addr, soff: wave-uniform)
__device__ __forceinline__ void dma_piece(unsigned lds_addr, unsigned voff, const dma_u32x4& rs, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
#endif
}
