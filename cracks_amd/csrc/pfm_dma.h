// pfm_dma.h — global -> LDS transfers without staging registers (global_load_lds_*), used by the z-marching kernels to
// request the next nodal plane ahead of the arithmetic of the current one.
//
// LDS address of lane l = (wave-uniform) lds + 4 l; global address = uniform base (SGPR pair) + per-lane byte offset.
// Written as asm so that the compiler does not track the transfer (it would wait vmcnt(0) at every later LDS access of the
// same __shared__ object): the CONSUMER waits with an explicit s_waitcnt vmcnt before the workgroup barrier.  M0 carries
// a 16-bit LDS offset: the destinations must lie in the first 64 KB of the workgroup's LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pfm
{
  namespace
  {
    __device__ __forceinline__ void dma_b32(const void *base, unsigned byte_off, void *lds)
    {
      const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(byte_off), "s"(base), "s"(l)
                   : "memory");
    }
    // 16 bytes per lane: LDS address of lane l = lds + 16 l.  The global address needs 4-byte alignment only
    // (tools/microbench/dma_align.hip: 8-byte aligned pairs of doubles arrive intact; the 12-byte form lands with a
    // 16-byte lane stride, i.e. with holes)
    __device__ __forceinline__ void dma_b128(const void *base, unsigned byte_off, void *lds)
    {
      const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(byte_off), "s"(base), "s"(l)
                   : "memory");
    }
    // six transfers with one save / restore of M0 (k_cart_residual3d: the six fields of a plane)
    __device__ __forceinline__ void dma_b32x6(const void *const (&base)[6], const unsigned (&byte_off)[6], void *const (&lds)[6])
    {
      unsigned l[6];
#pragma unroll
      for (int i = 0; i < 6; ++i)
        l[i] = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds[i]);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\t"
                   "s_mov_b32 m0, %13\n\ts_nop 0\n\tglobal_load_lds_dword %1, %7\n\t"
                   "s_mov_b32 m0, %14\n\ts_nop 0\n\tglobal_load_lds_dword %2, %8\n\t"
                   "s_mov_b32 m0, %15\n\ts_nop 0\n\tglobal_load_lds_dword %3, %9\n\t"
                   "s_mov_b32 m0, %16\n\ts_nop 0\n\tglobal_load_lds_dword %4, %10\n\t"
                   "s_mov_b32 m0, %17\n\ts_nop 0\n\tglobal_load_lds_dword %5, %11\n\t"
                   "s_mov_b32 m0, %18\n\ts_nop 0\n\tglobal_load_lds_dword %6, %12\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(byte_off[0]), "v"(byte_off[1]), "v"(byte_off[2]), "v"(byte_off[3]), "v"(byte_off[4]), "v"(byte_off[5]),
                     "s"(base[0]), "s"(base[1]), "s"(base[2]), "s"(base[3]), "s"(base[4]), "s"(base[5]),
                     "s"(l[0]), "s"(l[1]), "s"(l[2]), "s"(l[3]), "s"(l[4]), "s"(l[5])
                   : "memory");
    }
    // one byte per lane, zero-extended to the lane's dword in LDS
    __device__ __forceinline__ void dma_u8(const void *base, unsigned byte_off, void *lds)
    {
      const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_ubyte %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(byte_off), "s"(base), "s"(l)
                   : "memory");
    }
  } // namespace
} // namespace pfm
