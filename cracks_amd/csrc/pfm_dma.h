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
