// stw.hip — what a vector store instruction costs a CU: 4 waves per workgroup, one workgroup per CU (and two), each wave
// writes NS contiguous pieces per "step" like the copy-out of k_cart_phi4 (8 B or 16 B per lane, 16-byte aligned or off by 8),
// steps separated by ~COMPUTE cycles of FMAs so that the memory is not saturated.  Prints cycles per store instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int W, int OFF>
__global__ __launch_bounds__(256, 2) void k_stw(double *out, long long wg_stride, int steps, int ns, int compute, unsigned long long *ticks, double *sink)
{
  const int t = threadIdx.x;
  double *base = out + (long long)blockIdx.x * wg_stride + OFF;
  double x = 1.0 + t * 1e-9, y = 0.5;
  unsigned long long acc = 0;
  for (int s = 0; s < steps; ++s)
    {
      for (int i = 0; i < compute; ++i)
        x = fma(x, 1.0000001, y);
      const unsigned long long t0 = clock64();
      double *p = base + (long long)s * ns * 256 * W;
      for (int j = 0; j < ns; ++j)
        {
          if constexpr (W == 1)
            p[j * 256 + t] = x;
          else
            {
              double2 v = make_double2(x, x + 1.0);
              __builtin_memcpy(&p[(j * 256 + t) * 2], &v, 16); // 8-byte aligned address: global_store_dwordx4
            }
        }
      acc += clock64() - t0;
    }
  if (t == 0)
    ticks[blockIdx.x] = acc;
  if (x == 123.456)
    *sink = x;
}

int main()
{
  const int NB[2] = {256, 512};
  const int steps = 20, ns = 49;
  double *out, *sink;
  unsigned long long *ticks;
  const long long wg_stride = (long long)steps * ns * 256 * 2 + 64;
  hipMalloc((void **)&out, sizeof(double) * (wg_stride * 512 + 64));
  hipMalloc((void **)&ticks, sizeof(unsigned long long) * 512);
  hipMalloc((void **)&sink, 8);
  for (int nb : NB)
    for (int compute : {0, 3000})
      for (int var = 0; var < 4; ++var)
        {
          for (int rep = 0; rep < 2; ++rep)
            {
              if (var == 0)
                hipLaunchKernelGGL((k_stw<1, 0>), dim3(nb), dim3(256), 0, 0, out, wg_stride, steps, ns, compute, ticks, sink);
              if (var == 1)
                hipLaunchKernelGGL((k_stw<1, 1>), dim3(nb), dim3(256), 0, 0, out, wg_stride, steps, ns, compute, ticks, sink);
              if (var == 2)
                hipLaunchKernelGGL((k_stw<2, 0>), dim3(nb), dim3(256), 0, 0, out, wg_stride, steps, ns / 2, compute, ticks, sink);
              if (var == 3)
                hipLaunchKernelGGL((k_stw<2, 1>), dim3(nb), dim3(256), 0, 0, out, wg_stride, steps, ns / 2, compute, ticks, sink);
              hipDeviceSynchronize();
            }
          std::vector<unsigned long long> h(nb);
          hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * nb, hipMemcpyDeviceToHost);
          double sum = 0;
          for (auto v : h)
            sum += (double)v;
          const int n_st = var < 2 ? ns : ns / 2;
          const char *names[4] = {"8 B/lane aligned", "8 B/lane base+8", "16 B/lane aligned", "16 B/lane base+8"};
          printf("workgroups %3d  compute %4d  %-18s  %8.0f cycles per step of %d stores/wave = %6.1f cycles per store, %6.1f per KB and wave\n", nb,
                 compute, names[var], sum / nb / steps, n_st, sum / nb / steps / n_st, sum / nb / steps / (n_st * (var < 2 ? 0.5 : 1.0)));
        }
  return 0;
}
