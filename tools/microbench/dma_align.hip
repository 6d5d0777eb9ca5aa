// global_load_lds_dwordx3 / dwordx4 with per-lane addresses that are 8-byte (not 16-byte) aligned: does the data arrive intact?
// build: hipcc --offload-arch=gfx950 -O2 dma_align.hip -o dma_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
__global__ void k(const double *g, double *out, int shift, int mode)
{
  __shared__ double s[256];
  const int lane = threadIdx.x;
  const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&s[0]);
  unsigned keep;
  if (mode == 4)
    {
      // lane reads doubles [shift + 2 lane*3 .. +1]  (stride 3 pairs so addresses are not contiguous), 8-byte aligned
      const unsigned off = 8u * (shift + 3 * lane);
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(off), "s"(g), "s"(l) : "memory");
    }
  else
    {
      const unsigned off = 4u * (shift + 5 * lane); // 4-byte aligned, 12 bytes
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(off), "s"(g), "s"(l) : "memory");
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64)
    out[i] = s[i];
}
int main()
{
  const int N = 4096;
  std::vector<double> h(N);
  for (int i = 0; i < N; ++i) { unsigned w[2] = {0x10000u + 2u * i, 0x10000u + 2u * i + 1u}; memcpy(&h[i], w, 8); }
  double *g, *o;
  hipMalloc(&g, N * 8); hipMalloc(&o, 256 * 8);
  hipMemcpy(g, h.data(), N * 8, hipMemcpyHostToDevice);
  int bad = 0;
  for (int shift = 0; shift < 4; ++shift)
    {
      hipMemset(o, 0, 256 * 8);
      k<<<1, 64>>>(g, o, shift, 4);
      std::vector<double> r(256);
      hipMemcpy(r.data(), o, 256 * 8, hipMemcpyDeviceToHost);
      int b = 0;
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 2; ++e)
          if (r[2 * lane + e] != h[shift + 3 * lane + e]) ++b;
      printf("x4 shift %d (byte offset %% 16 = %d): %d wrong of 128\n", shift, (8 * shift) % 16, b);
      bad += b;
    }
  for (int shift = 0; shift < 4; ++shift)
    {
      hipMemset(o, 0, 256 * 8);
      k<<<1, 64>>>(g, o, shift, 3);
      std::vector<unsigned> r(512);
      hipMemcpy(r.data(), o, 256 * 8, hipMemcpyDeviceToHost);
      const unsigned *hw = reinterpret_cast<const unsigned *>(h.data());
      int b = 0;
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 3; ++e)
          if (r[3 * lane + e] != hw[shift + 5 * lane + e]) ++b;
      printf("x3 shift %d dwords: %d wrong of 192\n", shift, b);
      if (shift == 0)
        for (int i = 0; i < 260; ++i)
          {
            // which source dword (index into hw) landed in LDS dword i?
            int src = -1;
            for (int q = 0; q < 400; ++q) if (hw[q] == r[i] && r[i] != 0) { src = q; break; }
            printf("%d:%d ", i, src);
            if (i % 16 == 15) printf("\n");
          }
      bad += b;
    }
  printf(bad ? "FAIL\n" : "OK\n");
  return bad != 0;
}
