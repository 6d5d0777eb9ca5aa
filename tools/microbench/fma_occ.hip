// fma_occ.hip — v_fma_f64 issue rate of one SIMD against the number of resident waves and the number of independent
// dependency chains per wave (gfx950).  One workgroup per CU, W waves per SIMD (4 W waves per workgroup), every wave runs
// `iters` x 64 FMAs arranged as C independent chains; time by s_memtime of wave 0.  Prints FMAs per cycle per SIMD
// (peak 0.25: one wave-instruction per 4 cycles) and the cycles between two FMAs of one wave.
//   hipcc --offload-arch=gfx950 -O3 fma_occ.hip -o fma_occ && ./fma_occ
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int C>
__global__ void k_fma32(double *out, long long *cycles, int iters)
{
  float a[C];
#pragma unroll
  for (int c = 0; c < C; ++c)
    a[c] = threadIdx.x * 1e-3f + c;
  const float m = 1.0f + 1e-7f * threadIdx.x, b = 1e-7f;
  __syncthreads();
  for (int it = 0; it < iters; ++it)
    {
#pragma unroll
      for (int r = 0; r < 64 / C; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c)
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(b));
    }
  float s = 0;
#pragma unroll
  for (int c = 0; c < C; ++c)
    s += a[c];
  if (s == 12345.678f)
    out[0] = s;
  (void)cycles;
}

template <int C>
__global__ void k_fma(double *out, long long *cycles, int iters)
{
  double a[C];
#pragma unroll
  for (int c = 0; c < C; ++c)
    a[c] = threadIdx.x * 1e-3 + c;
  const double m = 1.0 + 1e-9 * threadIdx.x, b = 1e-7;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it)
    {
#pragma unroll
      for (int r = 0; r < 64 / C; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c)
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(b));
    }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < C; ++c)
    s += a[c];
  if (s == 12345.678)
    out[0] = s;
  if (threadIdx.x == 0)
    cycles[blockIdx.x] = t1 - t0;
}

template <int C, bool F32 = false>
void run(int waves_per_simd, double *d_out, long long *d_cyc, int n_cu, double clock_ratio)
{
  const int iters = 20000;
  int threads = 256 * waves_per_simd, wgs_per_cu = 1;
  if (threads > 1024)
    {
      threads /= 2;
      wgs_per_cu = 2;
    }
  auto kern = F32 ? k_fma32<C> : k_fma<C>;
  hipLaunchKernelGGL(kern, dim3(n_cu * wgs_per_cu), dim3(threads), 0, 0, d_out, d_cyc, iters);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(n_cu * wgs_per_cu), dim3(threads), 0, 0, d_out, d_cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> cyc(n_cu * wgs_per_cu);
  hipMemcpy(cyc.data(), d_cyc, sizeof(long long) * n_cu * wgs_per_cu, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto c : cyc)
    mean += (double)c / n_cu;
  const double fmas_per_wave = (double)iters * 64;
  // s_memtime counts at 100 MHz; convert with the event time instead: kernel cycles = ms * clock
  const double per_simd = fmas_per_wave * waves_per_simd; // wave-instructions per SIMD
  printf("%s CUs %3d waves/SIMD %d chains %d: %.3f ms, %.1f ns per FMA of one wave, SIMD rate %.3f wave-FMA/ns (x4 cycles @2.4GHz = %.2f of peak)\n", F32 ? "f32" : "f64", n_cu, waves_per_simd, C,
         ms, ms * 1e6 / fmas_per_wave, per_simd / (ms * 1e6), per_simd / (ms * 1e6) * 4 / 2.4);
  (void)mean;
  (void)clock_ratio;
}

int main()
{
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  double *d_out;
  long long *d_cyc;
  hipMalloc(&d_out, 64);
  hipMalloc(&d_cyc, sizeof(long long) * n_cu * 2);
  printf("%s, %d CUs, clock %.0f MHz\n", p.name, n_cu, p.clockRate / 1e3);
  for (int cus : {1, 32, n_cu})
    for (int w : {1, 2, 3, 4, 8})
      {
        run<1>(w, d_out, d_cyc, cus, 0);
        run<4>(w, d_out, d_cyc, cus, 0);
        run<8>(w, d_out, d_cyc, cus, 0);
        run<8, true>(w, d_out, d_cyc, cus, 0);
      }
  return 0;
}
