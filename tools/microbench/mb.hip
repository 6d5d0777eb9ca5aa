// mb.hip — gfx950 micro-benchmarks that ground the kernel design decisions of DESIGN.md:
//   * v_mfma_f64_16x16x4_f64 issue rate, alone and next to FP64 VALU waves on the same SIMD
//   * v_fma_f64 issue rate
//   * LDS: ds_read_b64, ds_write_b64, ds_add_f64 (no return) wave-instruction throughput
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics mb.hip -o mb ; run: ./mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

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

// mode bit 0: wave does MFMA, bit 1: wave does FMA; roles by wave index parity when both
template <int ROLE_SPLIT>
__global__ __launch_bounds__(512) void k_mfma_fma(double *out, int iters, int mfma_waves_mask, int fma_waves_mask)
{
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = (mfma_waves_mask >> wave) & 1, do_fma = (fma_waves_mask >> wave) & 1;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  d4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
  double f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3, f4 = a + 4, f5 = a + 5, f6 = a + 6, f7 = a + 7;
  const long long t0 = clock64();
  if (do_mfma && !do_fma)
    for (int i = 0; i < iters; ++i)
      {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      }
  else if (do_fma && !do_mfma)
    for (int i = 0; i < iters; ++i)
      {
#pragma unroll
        for (int r = 0; r < 2; ++r)
          {
            f0 = fma(f0, b, a);
            f1 = fma(f1, b, a);
            f2 = fma(f2, b, a);
            f3 = fma(f3, b, a);
            f4 = fma(f4, b, a);
            f5 = fma(f5, b, a);
            f6 = fma(f6, b, a);
            f7 = fma(f7, b, a);
          }
      }
  else if (do_fma && do_mfma) // interleaved in ONE wave: 4 MFMA + 16 FMA per iteration
    for (int i = 0; i < iters; ++i)
      {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        f0 = fma(f0, b, a);
        f1 = fma(f1, b, a);
        f2 = fma(f2, b, a);
        f3 = fma(f3, b, a);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        f4 = fma(f4, b, a);
        f5 = fma(f5, b, a);
        f6 = fma(f6, b, a);
        f7 = fma(f7, b, a);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        f0 = fma(f0, b, a);
        f1 = fma(f1, b, a);
        f2 = fma(f2, b, a);
        f3 = fma(f3, b, a);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        f4 = fma(f4, b, a);
        f5 = fma(f5, b, a);
        f6 = fma(f6, b, a);
        f7 = fma(f7, b, a);
      }
  const long long t1 = clock64();
  double s = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (double)(t1 - t0) * 1e-300;
  if ((threadIdx.x & 63) == 0)
    out[(size_t)gridDim.x * blockDim.x + blockIdx.x * 8 + wave] = (double)(t1 - t0);
}

// LDS ops: OP 0 = ds_read_b64, 1 = ds_write_b64, 2 = ds_add_f64 (no return), 3 = ds_add_rtn_f64, 4 = ds_read_b128
// 16 operations per s_waitcnt, issued from one asm block (no compiler bookkeeping in between).
#define R16(OPSTR)                                                                                                   \
  OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7) OPSTR(8) OPSTR(9) OPSTR(10) OPSTR(11)      \
  OPSTR(12) OPSTR(13) OPSTR(14) OPSTR(15)
template <int OP>
__global__ __launch_bounds__(1024) void k_lds(double *out, int iters, int stride /* in doubles between lanes */)
{
  extern __shared__ double sm[];
  const int t = threadIdx.x;
  for (int i = t; i < 8192; i += blockDim.x)
    sm[i] = i;
  __syncthreads();
  double acc = 1.0;
  const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) double *)(sm + ((t * stride) & 2047));
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i)
    {
      if constexpr (OP == 0)
        {
          double x0, x1, x2, x3;
          asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:4096\n\tds_read_b64 %2, %4 offset:8192\n\tds_read_b64 %3, %4 offset:12288\n\t"
                       "ds_read_b64 %0, %4 offset:16384\n\tds_read_b64 %1, %4 offset:20480\n\tds_read_b64 %2, %4 offset:24576\n\tds_read_b64 %3, %4 offset:28672\n\t"
                       "ds_read_b64 %0, %4 offset:32768\n\tds_read_b64 %1, %4 offset:36864\n\tds_read_b64 %2, %4 offset:40960\n\tds_read_b64 %3, %4 offset:45056\n\t"
                       "ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:4096\n\tds_read_b64 %2, %4 offset:8192\n\tds_read_b64 %3, %4 offset:12288\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                       : "v"(la)
                       : "memory");
          acc += x0 + x1 + x2 + x3;
        }
      else if constexpr (OP == 4)
        {
          typedef double d2 __attribute__((ext_vector_type(2)));
          d2 x0, x1, x2, x3;
          const unsigned lb = (unsigned)(size_t)(__attribute__((address_space(3))) double *)(sm + ((2 * t * stride) & 2047));
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %4 offset:12288\n\t"
                       "ds_read_b128 %0, %4 offset:16384\n\tds_read_b128 %1, %4 offset:20480\n\tds_read_b128 %2, %4 offset:24576\n\tds_read_b128 %3, %4 offset:28672\n\t"
                       "ds_read_b128 %0, %4 offset:32768\n\tds_read_b128 %1, %4 offset:36864\n\tds_read_b128 %2, %4 offset:40960\n\tds_read_b128 %3, %4 offset:45056\n\t"
                       "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %4 offset:12288\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                       : "v"(lb)
                       : "memory");
          acc += x0[0] + x1[1] + x2[0] + x3[1];
        }
      else if constexpr (OP == 1)
        {
#define W(k) "ds_write_b64 %0, %1 offset:" #k "*2048\n\t"
          asm volatile(R16(W) "s_waitcnt lgkmcnt(0)" ::"v"(la), "v"(acc) : "memory");
#undef W
        }
      else if constexpr (OP == 2)
        {
#define W(k) "ds_add_f64 %0, %1 offset:" #k "*2048\n\t"
          asm volatile(R16(W) "s_waitcnt lgkmcnt(0)" ::"v"(la), "v"(acc) : "memory");
#undef W
        }
      else
        {
          double x0, x1, x2, x3;
          asm volatile("ds_add_rtn_f64 %0, %4, %5\n\tds_add_rtn_f64 %1, %4, %5 offset:4096\n\tds_add_rtn_f64 %2, %4, %5 offset:8192\n\tds_add_rtn_f64 %3, %4, %5 offset:12288\n\t"
                       "ds_add_rtn_f64 %0, %4, %5 offset:16384\n\tds_add_rtn_f64 %1, %4, %5 offset:20480\n\tds_add_rtn_f64 %2, %4, %5 offset:24576\n\tds_add_rtn_f64 %3, %4, %5 offset:28672\n\t"
                       "ds_add_rtn_f64 %0, %4, %5 offset:32768\n\tds_add_rtn_f64 %1, %4, %5 offset:36864\n\tds_add_rtn_f64 %2, %4, %5 offset:40960\n\tds_add_rtn_f64 %3, %4, %5 offset:45056\n\t"
                       "ds_add_rtn_f64 %0, %4, %5\n\tds_add_rtn_f64 %1, %4, %5 offset:4096\n\tds_add_rtn_f64 %2, %4, %5 offset:8192\n\tds_add_rtn_f64 %3, %4, %5 offset:12288\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                       : "v"(la), "v"(acc)
                       : "memory");
          acc += x0 + x1 + x2 + x3;
        }
    }
  const long long t1 = clock64();
  __syncthreads();
  out[blockIdx.x * blockDim.x + t] = acc + sm[t];
  if ((t & 63) == 0)
    out[(size_t)gridDim.x * blockDim.x + blockIdx.x * 16 + (t >> 6)] = (double)(t1 - t0);
}

// integer VALU next to an f64 MFMA wave on the same SIMD (address arithmetic beside the matrix pipe)
__global__ __launch_bounds__(512) void k_mfma_int(double *out, int iters, int int_waves_mask)
{
  const int wave = threadIdx.x >> 6;
  const bool do_int = (int_waves_mask >> wave) & 1;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  d4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  unsigned i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
  const long long t0 = clock64();
  if (!do_int)
    for (int i = 0; i < iters; ++i)
      {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      }
  else
    for (int i = 0; i < iters; ++i)
      {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          {
            i0 = i0 * 3u + i7;
            i1 = i1 * 3u + i0;
            i2 = i2 * 3u + i1;
            i3 = i3 * 3u + i2;
            i4 = i4 * 3u + i3;
            i5 = i5 * 3u + i4;
            i6 = i6 * 3u + i5;
            i7 = i7 * 3u + i6;
          }
      }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + (double)(i0 ^ i1 ^ i2 ^ i3 ^ i4 ^ i5 ^ i6 ^ i7);
  if ((threadIdx.x & 63) == 0)
    out[(size_t)gridDim.x * blockDim.x + blockIdx.x * 8 + wave] = (double)(t1 - t0);
}

int main()
{
  double *d;
  const int NB = 256;
  CK(hipMalloc(&d, sizeof(double) * (NB * 1024 + NB * 16)));
  std::vector<double> h(NB * 1024 + NB * 16);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 20000;
  struct Cfg
  {
    const char *name;
    int threads, mm, fm;
  } cfgs[] = {
    {"MFMA f64 16x16x4: 1 wave/SIMD (4 waves/CU)", 256, 0xf, 0},
    {"MFMA f64 16x16x4: 2 waves/SIMD", 512, 0xff, 0},
    {"FMA f64: 1 wave/SIMD", 256, 0, 0xf},
    {"FMA f64: 2 waves/SIMD", 512, 0, 0xff},
    {"MFMA waves 0-3 + FMA waves 4-7 (one of each per SIMD)", 512, 0x0f, 0xf0},
    {"MFMA+FMA interleaved in one wave, 1 wave/SIMD", 256, 0xf, 0xf},
  };
  for (auto &c : cfgs)
    {
      for (int rep = 0; rep < 2; ++rep)
        {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(k_mfma_fma<0>, dim3(NB), dim3(c.threads), 0, 0, d, iters, c.mm, c.fm);
          CK(hipEventRecord(e1));
          CK(hipDeviceSynchronize());
        }
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h.data(), d, sizeof(double) * (NB * c.threads + NB * 8), hipMemcpyDeviceToHost));
      const int nw = c.threads / 64;
      double cyc_m = 0, cyc_f = 0;
      int nm = 0, nf = 0;
      for (int w = 0; w < nw; ++w)
        {
          const double cy = h[(size_t)NB * c.threads + 0 * 8 + w];
          if ((c.mm >> w) & 1)
            cyc_m += cy, ++nm;
          else
            cyc_f += cy, ++nf;
        }
      const bool both = (c.mm & c.fm) != 0;
      printf("%-60s %8.3f ms", c.name, ms);
      if (nm)
        printf("  MFMA-wave cycles/MFMA %.1f", cyc_m / nm / (4.0 * iters));
      if (nf)
        printf("  FMA-wave cycles/FMA %.2f", cyc_f / nf / (16.0 * iters));
      if (both)
        printf("  (the wave also issues 4 FMA per MFMA)");
      // flops
      const double mf = (double)NB * nm * iters * 4.0 * 2048.0, ff = (double)NB * (both ? nm : nf) * iters * 16.0 * 128.0;
      printf("  => MFMA %.1f TF, FMA %.1f TF\n", mf / ms * 1e-9, ff / ms * 1e-9);
    }
  {
    // integer VALU beside the f64 MFMA
    for (int mask : {0x00, 0xf0})
      {
        for (int rep = 0; rep < 2; ++rep)
          {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_mfma_int, dim3(NB), dim3(512), 0, 0, d, iters, mask);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
          }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), d, sizeof(double) * (NB * 512 + NB * 8), hipMemcpyDeviceToHost));
        printf("MFMA f64 waves 0-3%s: %8.3f ms, MFMA wave %.1f cycles/MFMA, waves 4-7: %.2f cycles per int mad (32 per iteration)\n",
               mask ? " + int-VALU waves 4-7" : " + MFMA waves 4-7", ms, h[(size_t)NB * 512] / (2.0 * iters),
               h[(size_t)NB * 512 + 4] / (32.0 * iters));
      }
  }
  const char *opn[5] = {"ds_read_b64", "ds_write_b64", "ds_add_f64", "ds_add_rtn_f64", "ds_read_b128"};
  for (int threads : {256, 512, 1024})
    for (int op = 0; op < 5; ++op)
      for (int stride : {1, 9})
        {
          const int it2 = 2000;
          for (int rep = 0; rep < 2; ++rep)
            {
              CK(hipEventRecord(e0));
              switch (op)
                {
                  case 0: hipLaunchKernelGGL(k_lds<0>, dim3(NB), dim3(threads), 65536, 0, d, it2, stride); break;
                  case 1: hipLaunchKernelGGL(k_lds<1>, dim3(NB), dim3(threads), 65536, 0, d, it2, stride); break;
                  case 2: hipLaunchKernelGGL(k_lds<2>, dim3(NB), dim3(threads), 65536, 0, d, it2, stride); break;
                  case 3: hipLaunchKernelGGL(k_lds<3>, dim3(NB), dim3(threads), 65536, 0, d, it2, stride); break;
                  default: hipLaunchKernelGGL(k_lds<4>, dim3(NB), dim3(threads), 65536, 0, d, it2, stride); break;
                }
              CK(hipEventRecord(e1));
              CK(hipDeviceSynchronize());
            }
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          CK(hipMemcpy(h.data(), d, sizeof(double) * (NB * threads + NB * 16), hipMemcpyDeviceToHost));
          const double cy = h[(size_t)NB * threads];
          const double winstr = (double)(threads / 64) * it2 * 16.0;
          printf("%-16s %4d thr/CU, lane stride %d: %8.3f ms, %.2f cycles per wave-instr per CU (one wave: %.1f cyc/instr)\n",
                 opn[op], threads, stride, ms, cy / winstr, cy / (it2 * 16.0));
        }
  return 0;
}
