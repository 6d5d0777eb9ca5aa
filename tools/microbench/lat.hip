// lat.hip — latency of a small dependent global read at the start of a tile while the chip streams stores, through the
// vector memory path (global_load) and through the scalar path (s_load: scalar cache -> L2).  Models phase 0 of
// k_cart_uu3 (a 10 x 6 x 3 nodal halo of two fields, then ~61 KB of row stores per tile and workgroup).
// Build: hipcc --offload-arch=gfx950 -O3 lat.hip -o lat ; run: ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
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

constexpr int NX = 217, NY = 217, NZ = 217;
typedef const __attribute__((address_space(4))) double *cptr;

// MODE 0: vector loads (lanes <-> halo nodes), 1: scalar loads (wave <-> rows of 10 doubles), 2: no loads,
// 3: vector loads of the NEXT tile issued before the stores of the current one
template <int MODE>
__global__ __launch_bounds__(512, 4) void k_lat(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out,
                                                 int tiles_per_wg, int stores_per_thread, unsigned long long *__restrict__ ticks, double *sink)
{
  __shared__ double s_a[180], s_b[180];
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  unsigned long long acc = 0;
  double pa = 0.0, pb = 0.0; // MODE 3: values of the next tile, requested before the stores of the current one
  auto node_of = [&](long long tile) {
    const int tx = (int)(tile % 27), ty = (int)((tile / 27) % 54), tz = (int)((tile / (27 * 54)) % 215);
    const int tt = t < 180 ? t : 0;
    const int li = tt % 10, lj = (tt / 10) % 6, lk = tt / 60;
    return (tx * 8 + li) % NX + (long long)NX * ((ty * 4 + lj) % NY + (long long)NY * (tz + lk));
  };
  if (MODE == 3)
    {
      const long long n = node_of(blockIdx.x);
      pa = a[n];
      pb = b[n];
    }
  for (int it = 0; it < tiles_per_wg; ++it)
    {
      const long long tile = (long long)blockIdx.x + (long long)it * gridDim.x;
      const int tx = (int)(tile % 27), ty = (int)((tile / 27) % 54), tz = (int)((tile / (27 * 54)) % 215);
      const long long t0 = wall_clock64();
      if (MODE == 0)
        {
          if (t < 180)
            {
              const int li = t % 10, lj = (t / 10) % 6, lk = t / 60;
              const long long n = (tx * 8 + li) % NX + (long long)NX * ((ty * 4 + lj) % NY + (long long)NY * (tz + lk));
              s_a[t] = a[n];
              s_b[t] = b[n];
            }
        }
      else if (MODE == 1)
        {
          // 18 rows of 10 doubles per field: waves 0..7 take rows wave, wave + 8, wave + 16
          for (int r = wave; r < 18; r += 8)
            {
              const int lj = r % 6, lk = r / 6;
              const long long n0 = (long long)__builtin_amdgcn_readfirstlane(
                (int)((tx * 8) % (NX - 10) + (long long)NX * ((ty * 4 + lj) % NY + (long long)NY * (tz + lk))));
              cptr pa = (cptr)(a + n0), pb = (cptr)(b + n0);
              double va[10], vb[10];
#pragma unroll
              for (int i = 0; i < 10; ++i)
                {
                  va[i] = pa[i];
                  vb[i] = pb[i];
                }
              double ma = 0.0, mb = 0.0;
#pragma unroll
              for (int i = 0; i < 10; ++i)
                if (lane == i)
                  {
                    ma = va[i];
                    mb = vb[i];
                  }
              if (lane < 10)
                {
                  s_a[r * 10 + lane] = ma;
                  s_b[r * 10 + lane] = mb;
                }
            }
        }
      if (MODE == 3 && t < 180)
        {
          s_a[t] = pa;
          s_b[t] = pb;
        }
      __syncthreads();
      const long long t1 = wall_clock64();
      acc += (unsigned long long)(t1 - t0);
      // something that depends on the data, then the tile's row stores
      const double x = (MODE == 2) ? 1.0 : s_a[t % 180] + s_b[(t * 7) % 180];
      if (MODE == 3)
        {
          const long long n = node_of(tile + gridDim.x);
          pa = a[n];
          pb = b[n];
        }
      for (int j = 0; j < stores_per_thread; ++j)
        out[((tile % 4096) * stores_per_thread + j) * 512 + t] = x + j;
      __syncthreads();
    }
  if (t == 0)
    ticks[blockIdx.x] = acc;
  if (acc == 12345)
    sink[0] = 1.0;
}

int main()
{
  const size_t n = (size_t)NX * NY * NZ;
  double *a, *b, *out, *sink;
  unsigned long long *ticks;
  const int NB = 512, TPW = 400;
  CK(hipMalloc(&a, n * 8));
  CK(hipMalloc(&b, n * 8));
  CK(hipMalloc(&out, (size_t)4096 * 16 * 512 * 8));
  CK(hipMalloc(&sink, 8));
  CK(hipMalloc(&ticks, NB * 8));
  CK(hipMemset(a, 0, n * 8));
  CK(hipMemset(b, 0, n * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<unsigned long long> h(NB);
  for (int stores : {0, 15})
    for (int mode = 0; mode < 4; ++mode)
      {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep)
          {
            CK(hipEventRecord(e0));
            if (mode == 0)
              hipLaunchKernelGGL(k_lat<0>, dim3(NB), dim3(512), 0, 0, a, b, out, TPW, stores, ticks, sink);
            else if (mode == 1)
              hipLaunchKernelGGL(k_lat<1>, dim3(NB), dim3(512), 0, 0, a, b, out, TPW, stores, ticks, sink);
            else if (mode == 3)
              hipLaunchKernelGGL(k_lat<3>, dim3(NB), dim3(512), 0, 0, a, b, out, TPW, stores, ticks, sink);
            else
              hipLaunchKernelGGL(k_lat<2>, dim3(NB), dim3(512), 0, 0, a, b, out, TPW, stores, ticks, sink);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
          }
        CK(hipMemcpy(h.data(), ticks, NB * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (auto x : h)
          sum += (double)x;
        const double gb = (double)NB * TPW * stores * 512 * 8 / 1e9;
        printf("stores/thread %2d  %-12s  kernel %7.3f ms  store stream %6.2f TB/s  load phase %7.1f ns per tile\n", stores,
               mode == 0 ? "vector loads" : mode == 1 ? "scalar loads" : mode == 3 ? "prefetched" : "no loads", ms, gb / ms, sum / NB / TPW * 10.0);
      }
  return 0;
}
