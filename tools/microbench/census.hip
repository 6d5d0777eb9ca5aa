// census.hip — where do the workgroups of two concurrently launched kernels land?  Two kernels of G workgroups each
// (80 KB of LDS: at most two workgroups per CU), on two streams; every workgroup records (XCC, SE, CU) and spins for
// ~200 us so that all of them are resident together.  Prints how many CUs host A+B, A+A, B+B, one or none.
//   hipcc --offload-arch=gfx950 -O2 census.hip -o census && ./census [G=256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void k_census(unsigned *out, long long spin, int tag)
{
  extern __shared__ char lds[];
  lds[threadIdx.x] = (char)tag;
  if (threadIdx.x == 0)
    {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      out[2 * blockIdx.x] = hw;
      out[2 * blockIdx.x + 1] = xcc;
    }
  const long long t0 = clock64();
  while (clock64() - t0 < spin)
    ;
  if (lds[threadIdx.x] == 99)
    out[0] = 0;
}

int main(int argc, char **argv)
{
  const int G = argc > 1 ? atoi(argv[1]) : 256;
  unsigned *dA, *dB;
  hipMalloc(&dA, sizeof(unsigned) * 2 * G);
  hipMalloc(&dB, sizeof(unsigned) * 2 * G);
  hipStream_t sA, sB;
  hipStreamCreateWithFlags(&sA, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&sB, hipStreamNonBlocking);
  hipFuncSetAttribute((const void *)k_census, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
  for (int rep = 0; rep < 3; ++rep)
    {
      hipLaunchKernelGGL(k_census, dim3(G), dim3(256), 80000, sA, dA, 400000LL, 1);
      hipLaunchKernelGGL(k_census, dim3(G), dim3(256), 80000, sB, dB, 400000LL, 2);
      hipDeviceSynchronize();
      std::vector<unsigned> a(2 * G), b(2 * G);
      hipMemcpy(a.data(), dA, sizeof(unsigned) * 2 * G, hipMemcpyDeviceToHost);
      hipMemcpy(b.data(), dB, sizeof(unsigned) * 2 * G, hipMemcpyDeviceToHost);
      std::map<unsigned, std::pair<int, int>> cu; // key: xcc, se, cu
      auto key = [](unsigned hw, unsigned xcc) {
        const unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        return ((xcc & 0xf) << 16) | (se << 8) | (sh << 4) | cu_id;
      };
      for (int i = 0; i < G; ++i)
        {
          cu[key(a[2 * i], a[2 * i + 1])].first++;
          cu[key(b[2 * i], b[2 * i + 1])].second++;
        }
      int ab = 0, aa = 0, bb = 0, one = 0, other = 0;
      for (auto &kv : cu)
        {
          const int x = kv.second.first, y = kv.second.second;
          if (x == 1 && y == 1)
            ++ab;
          else if (x == 2 && y == 0)
            ++aa;
          else if (x == 0 && y == 2)
            ++bb;
          else if (x + y == 1)
            ++one;
          else
            ++other;
        }
      printf("rep %d: G=%d per kernel, distinct CUs seen %zu: A+B %d, A+A %d, B+B %d, single %d, other %d\n", rep, G, cu.size(), ab, aa, bb, one, other);
    }
  return 0;
}
