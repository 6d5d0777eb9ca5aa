// fill.hip — the write floor of the chip for output that nothing reads back: a hand-written store stream over 32 GiB
// with the cache policies gfx950 offers on global stores (default, nt, sc1, sc0 sc1, sc0 sc1 nt), 8- and 16-byte
// stores per lane, 256- and 512-thread workgroups, 1..4 waves per SIMD (occupancy set by dynamic LDS), streaming
// (one chunk per workgroup) and persistent (grid = CUs x occupancy) launches.  The copy-out phases of k_cart_uu3 and
// k_cart_phi4 write 35.4 GB per assembly this way; this table says what the memory system takes at best.
// Build: hipcc --offload-arch=gfx950 -O3 fill.hip -o fill ; run: ./fill [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                      \
  do                                                                                               \
    {                                                                                              \
      hipError_t e_ = (x);                                                                         \
      if (e_ != hipSuccess)                                                                        \
        {                                                                                          \
          printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);            \
          return 1;                                                                                \
        }                                                                                          \
    }                                                                                              \
  while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

// POL 0 default, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt, 5 sc0
template <int POL>
__device__ __forceinline__ void st16(double *p, v2d v)
{
  if (POL == 0)
    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  else if (POL == 1)
    asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else if (POL == 2)
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (POL == 3)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else if (POL == 4)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
  else
    asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}
template <int POL>
__device__ __forceinline__ void st8(double *p, double v)
{
  if (POL == 0)
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  else if (POL == 1)
    asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else if (POL == 2)
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (POL == 3)
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else if (POL == 4)
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
  else
    asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}

// every workgroup writes chunks of `chunk_doubles` (contiguous), chunk index = blockIdx + k * gridDim; within a chunk
// a wave's store instruction covers 1 KiB (16-byte) or 512 B (8-byte) contiguous, as the row copy-outs do.
// ROWS = 1: the chunk is written as rows of 81 doubles (648 B, the (u,u) row of a node) starting at 8-byte aligned
// but not 16-byte aligned addresses for every other row -- the real copy-out's alignment.
template <int POL, int W16>
__global__ void k_fill(double *__restrict__ out, long long n_chunks, int chunk_doubles)
{
  extern __shared__ double s_dummy[];
  const int t = threadIdx.x, nt = blockDim.x;
  const double v = (double)t;
  for (long long c = blockIdx.x; c < n_chunks; c += gridDim.x)
    {
      double *base = out + c * (long long)chunk_doubles;
      if (W16)
        {
          v2d vv = {v, v + 0.5};
          for (int i = 2 * t; i < chunk_doubles; i += 2 * nt)
            st16<POL>(base + i, vv);
        }
      else
        for (int i = t; i < chunk_doubles; i += nt)
          st8<POL>(base + i, v);
    }
  if (s_dummy[0] == 123.456 && out[0] == -1.0)
    out[1] = 0.0;
}

struct Cfg
{
  const char *name;
  int pol, w16;
};

template <int POL, int W16>
static float run(double *d, long long n_doubles, int threads, int lds_bytes, int grid_mode, int chunk_doubles, int reps)
{
  const long long n_chunks = n_doubles / chunk_doubles;
  int grid;
  if (grid_mode == 0)
    grid = (int)(n_chunks > 2000000000LL ? 2000000000LL : n_chunks);
  else
    grid = 256 * grid_mode; // persistent: grid_mode workgroups per CU
  hipFuncSetAttribute((const void *)k_fill<POL, W16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_fill<POL, W16><<<grid, threads, lds_bytes>>>(d, n_chunks, chunk_doubles);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r)
    k_fill<POL, W16><<<grid, threads, lds_bytes>>>(d, n_chunks, chunk_doubles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ms / reps;
}

typedef float (*runfn)(double *, long long, int, int, int, int, int);

int main(int argc, char **argv)
{
  const long long gib = argc > 1 ? atoll(argv[1]) : 32;
  const long long n_doubles = gib * (1LL << 30) / 8;
  double *d = nullptr;
  CK(hipMalloc(&d, n_doubles * 8));
  CK(hipMemset(d, 0, n_doubles * 8));
  CK(hipDeviceSynchronize());
  // library fill for reference
  {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipMemsetAsync(d, 1, n_doubles * 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r)
      hipMemsetAsync(d, 1, n_doubles * 8);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemsetAsync %lld GiB: %.3f ms -> %.2f TB/s\n", gib, ms / 3, n_doubles * 8.0 / (ms / 3 * 1e-3) / 1e12);
  }
  const char *pol_name[6] = {"default", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc0"};
  runfn f16[6] = {run<0, 1>, run<1, 1>, run<2, 1>, run<3, 1>, run<4, 1>, run<5, 1>};
  runfn f8[6] = {run<0, 0>, run<1, 0>, run<2, 0>, run<3, 0>, run<4, 0>, run<5, 0>};
  printf("%-12s %-5s %-8s %-10s %-12s %-9s %9s %8s\n", "policy", "bytes", "threads", "wg_per_cu", "launch", "chunk_KiB", "ms", "TB/s");
  const int chunk_list[2] = {8192, 81 * 8 * 32}; // 64 KiB; 32 tiles' worth of 8 rows of 81 doubles (odd multiple of 8 B rows)
  for (int w16 = 1; w16 >= 0; --w16)
    for (int pol = 0; pol < 6; ++pol)
      for (int threads = 256; threads <= 512; threads *= 2)
        for (int occ = 1; occ <= 8; occ *= 2) // workgroups per CU allowed by LDS (160 KiB per CU)
          {
            const int waves_per_simd = occ * threads / 256;
            if (waves_per_simd > 8)
              continue;
            const int lds = (160 * 1024) / occ - 1024;
            for (int gm = 0; gm <= 1; ++gm)
              {
                const int grid_mode = gm ? occ : 0;
                for (int ci = 0; ci < 2; ++ci)
                  {
                    if (ci == 1 && (gm == 1 || occ != 2))
                      continue;
                    const float ms = (w16 ? f16 : f8)[pol](d, n_doubles, threads, lds, grid_mode, chunk_list[ci], 3);
                    // occupancy is bounded by LDS only when lds is what was asked for; report the asked value
                    printf("%-12s %-5d %-8d %-10d %-12s %-9.1f %9.3f %8.2f\n", pol_name[pol], w16 ? 16 : 8, threads, occ,
                           gm ? "persistent" : "streaming", chunk_list[ci] * 8 / 1024.0, ms, n_doubles * 8.0 / (ms * 1e-3) / 1e12);
                  }
              }
          }
  hipFree(d);
  return 0;
}
