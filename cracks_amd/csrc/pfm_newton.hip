// pfm_newton.hip — the other per-iteration sweeps of the Newton / active-set loop, so
// that residual_total, diag_mass and the solution never leave the device between two assemblies:
//
//   pfm_diag_mass_device    assemble_diag_mass_matrix                 cracks.cc:2514-2562
//   pfm_active_set_device   active-set predicate + cycle counter      cracks.cc:2837-2886, 2903-2909
//                           + constraints_hanging_nodes.distribute    cracks.cc:2888-2890
//   pfm_functionals         compute_energy, compute_tcv               cracks.cc:3615-3701, 3553-3611
//
// All three work on any Q1 mesh (MappingQ1 geometry per cell) with the node state / tables of the context.
#include "pfm_internal.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/pfm_newton.h"

namespace pfm
{
  namespace
  {
    template <int dim>
    __device__ __forceinline__ long long dof_of(const DevView &v, int P, int c)
    {
      if (v.layout == PFM_LAYOUT_INTERLEAVED)
        return (long long)P * (dim + 1) + c;
      return c < dim ? (long long)P * dim + c : (long long)v.n_owned * dim + P;
    }

    __device__ __forceinline__ double det2(const double J[2][2]) { return J[0][0] * J[1][1] - J[0][1] * J[1][0]; }
    __device__ __forceinline__ double det3(const double J[3][3])
    {
      return J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) + J[0][1] * (J[1][2] * J[2][0] - J[1][0] * J[2][2]) +
             J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
    }

    // ---- diag_mass: QGaussLobatto(2) = the vertices with weight 2^-dim; the phase-field dof of vertex a of a
    // cell receives det J(vertex a) 2^-dim.  thread <-> (cell, vertex).
    template <int dim>
    __global__ __launch_bounds__(256) void k_diag_mass(DevView v, double *__restrict__ mass)
    {
      constexpr int nv = 1 << dim;
      const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      if (gid >= v.n_cells * nv)
        return;
      const long long cell = gid / nv;
      const int a = (int)(gid - cell * nv);
      double x[nv][dim];
      int node_a = 0;
#pragma unroll
      for (int b = 0; b < nv; ++b)
        {
          const int n = v.conn[(long long)b * v.n_cells + cell];
          if (b == a)
            node_a = n;
#pragma unroll
          for (int d = 0; d < dim; ++d)
            x[b][d] = v.coords[(long long)d * v.n_nodes + n];
        }
      if (node_a >= v.n_owned)
        return; // the owner of the node sums it (owner computes: its cells are all local)
      double J[dim][dim];
#pragma unroll
      for (int i = 0; i < dim; ++i)
#pragma unroll
        for (int j = 0; j < dim; ++j)
          {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < nv; ++b)
              {
                double g = 1.0;
#pragma unroll
                for (int d = 0; d < dim; ++d)
                  {
                    const double xa = (a >> d) & 1;
                    if (d == j)
                      g *= ((b >> d) & 1) ? 1.0 : -1.0;
                    else
                      g *= ((b >> d) & 1) ? xa : (1.0 - xa);
                  }
                s += x[b][i] * g;
              }
            J[i][j] = s;
          }
      double det;
      if constexpr (dim == 2)
        det = det2(J);
      else
        det = det3(J);
      unsafeAtomicAdd(&mass[node_a], det / (double)nv);
    }

    // ---- active set, thread <-> owned node
    template <int dim>
    __global__ __launch_bounds__(256) void k_active_set(DevView v, uint8_t *__restrict__ node_flags,
                                                        const double *__restrict__ residual_total,
                                                        const double *__restrict__ mass, double cconst,
                                                        double *__restrict__ solution, const double *__restrict__ old_solution,
                                                        int32_t *__restrict__ cycle_counter,
                                                        unsigned long long *__restrict__ counts)
    {
      const int P = blockIdx.x * blockDim.x + threadIdx.x;
      unsigned active_now = 0, cycling = 0, changed = 0;
      if (P < v.n_owned)
        {
          const uint8_t f = node_flags[P];
          const bool was = (f >> dim) & 1u;
          const bool hanging = v.hn_index && v.hn_index[P] >= 0; // constraints_hanging_nodes.is_constrained(idx)
          bool now = false;
          if (!hanging)
            {
              const long long idx = dof_of<dim>(v, P, dim);
              const double old_value = old_solution[idx], new_value = solution[idx];
              const double gap = new_value - old_value;
              const int cyc = cycle_counter[P];
              const bool inactive = (residual_total[idx] / mass[P] + cconst * gap <= 0.0) && cyc < 5; // cracks.cc:2868-2872
              if (!inactive)
                {
                  cycling = cyc >= 5;
                  now = true;
                  solution[idx] = old_value; // cracks.cc:2880
                }
            }
          if (was && !now)
            cycle_counter[P] += 1; // cracks.cc:2905-2908
          changed = was != now;
          active_now = now;
          node_flags[P] = (uint8_t)((f & ~(1u << dim)) | ((now ? 1u : 0u) << dim));
        }
      // integer counts: exact and order independent
      const unsigned long long b0 = __ballot(active_now), b1 = __ballot(cycling), b2 = __ballot(changed);
      if ((threadIdx.x & 63) == 0)
        {
          if (b0)
            atomicAdd(&counts[0], (unsigned long long)__popcll(b0));
          if (b1)
            atomicAdd(&counts[1], (unsigned long long)__popcll(b1));
          if (b2)
            atomicAdd(&counts[2], 1ull);
        }
    }

    // constraints_hanging_nodes.distribute(solution): thread <-> (hanging node, component)
    template <int dim>
    __global__ __launch_bounds__(256) void k_distribute_hanging(DevView v, double *__restrict__ solution)
    {
      const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      const int P = (int)(gid / (dim + 1)), c = (int)(gid % (dim + 1));
      if (P >= v.n_owned || v.hn_index[P] < 0)
        return;
      const int k = v.hn_index[P];
      double s = 0.0;
      for (long long j = v.hn_ptr[k]; j < v.hn_ptr[k + 1]; ++j)
        s += v.hn_weights[j] * solution[dof_of<dim>(v, v.hn_parents[j], c)];
      solution[dof_of<dim>(v, P, c)] = s;
    }

    // ---- energies and total crack volume: thread <-> cell, QGauss(3)^dim, MappingQ1; fixed-order reduction
    template <int dim>
    __global__ __launch_bounds__(256) void k_functionals(DevView v, pfm_params prm, const uint8_t *__restrict__ cell_owned,
                                                         const double *__restrict__ lam_over, const double *__restrict__ mu_over,
                                                         double *__restrict__ partial /* [gridDim.x][3] */)
    {
      constexpr int nv = 1 << dim, nq = dim == 2 ? 9 : 27;
      const double gx[3] = {0.5 - 0.5 * 0.7745966692414834, 0.5, 0.5 + 0.5 * 0.7745966692414834};
      const double gw[3] = {5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0};
      const long long cell = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      double acc[3] = {0.0, 0.0, 0.0};
      if (cell < v.n_cells && (!cell_owned || cell_owned[cell]))
        {
          double x[nv][dim], U[nv][dim], PH[nv];
#pragma unroll
          for (int b = 0; b < nv; ++b)
            {
              const int n = v.conn[(long long)b * v.n_cells + cell];
#pragma unroll
              for (int d = 0; d < dim; ++d)
                {
                  x[b][d] = v.coords[(long long)d * v.n_nodes + n];
                  U[b][d] = v.u[d][n];
                }
              PH[b] = v.phi[n];
            }
          const double lam = lam_over ? lam_over[cell] : (v.cell_lambda ? v.cell_lambda[cell] : prm.lambda);
          const double mu = mu_over ? mu_over[cell] : (v.cell_mu ? v.cell_mu[cell] : prm.mu);
#pragma unroll 1
          for (int q = 0; q < nq; ++q)
            {
              const int qi[3] = {q % 3, (q / 3) % 3, q / 9};
              double w = 1.0;
#pragma unroll
              for (int d = 0; d < dim; ++d)
                w *= gw[qi[d]];
              double N[nv], dNr[nv][dim];
#pragma unroll
              for (int b = 0; b < nv; ++b)
                {
                  double val = 1.0;
#pragma unroll
                  for (int d = 0; d < dim; ++d)
                    val *= ((b >> d) & 1) ? gx[qi[d]] : (1.0 - gx[qi[d]]);
                  N[b] = val;
#pragma unroll
                  for (int e = 0; e < dim; ++e)
                    {
                      double g = 1.0;
#pragma unroll
                      for (int d = 0; d < dim; ++d)
                        g *= (d == e) ? (((b >> d) & 1) ? 1.0 : -1.0) : (((b >> d) & 1) ? gx[qi[d]] : (1.0 - gx[qi[d]]));
                      dNr[b][e] = g;
                    }
                }
              double J[dim][dim], inv[dim][dim], det;
#pragma unroll
              for (int i = 0; i < dim; ++i)
#pragma unroll
                for (int j = 0; j < dim; ++j)
                  {
                    double s = 0.0;
#pragma unroll
                    for (int b = 0; b < nv; ++b)
                      s += x[b][i] * dNr[b][j];
                    J[i][j] = s;
                  }
              if constexpr (dim == 2)
                {
                  det = det2(J);
                  const double id = 1.0 / det;
                  inv[0][0] = J[1][1] * id;
                  inv[0][1] = -J[0][1] * id;
                  inv[1][0] = -J[1][0] * id;
                  inv[1][1] = J[0][0] * id;
                }
              else
                {
                  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
                  const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
                  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
                  det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
                  const double id = 1.0 / det;
                  inv[0][0] = c00 * id;
                  inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
                  inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
                  inv[1][0] = c01 * id;
                  inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
                  inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
                  inv[2][0] = c02 * id;
                  inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
                  inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
                }
              double gu[dim][dim], gpf[dim], uq[dim], pf = 0.0;
#pragma unroll
              for (int c = 0; c < dim; ++c)
                {
                  uq[c] = 0.0;
                  gpf[c] = 0.0;
#pragma unroll
                  for (int d = 0; d < dim; ++d)
                    gu[c][d] = 0.0;
                }
#pragma unroll
              for (int b = 0; b < nv; ++b)
                {
                  double g[dim]; // physical gradient of N_b: J^{-T} grad_ref
#pragma unroll
                  for (int d = 0; d < dim; ++d)
                    {
                      double s = 0.0;
#pragma unroll
                      for (int e = 0; e < dim; ++e)
                        s += inv[e][d] * dNr[b][e];
                      g[d] = s;
                    }
                  pf += PH[b] * N[b];
#pragma unroll
                  for (int c = 0; c < dim; ++c)
                    {
                      uq[c] += U[b][c] * N[b];
                      gpf[c] += PH[b] * g[c];
#pragma unroll
                      for (int d = 0; d < dim; ++d)
                        gu[c][d] += U[b][c] * g[d];
                    }
                }
              double trE = 0.0, tr_e_2 = 0.0, gg = 0.0, ug = 0.0;
#pragma unroll
              for (int a = 0; a < dim; ++a)
                {
                  trE += gu[a][a];
                  gg += gpf[a] * gpf[a];
                  ug += uq[a] * gpf[a];
#pragma unroll
                  for (int b = 0; b < dim; ++b)
                    {
                      const double e = 0.5 * (gu[a][b] + gu[b][a]);
                      tr_e_2 += e * e;
                    }
                }
              const double JxW = det * w;
              const double psi_e = 0.5 * lam * trE * trE + mu * tr_e_2;
              acc[0] += ((1 + prm.constant_k) * pf * pf + prm.constant_k) * psi_e * JxW;                               // cracks.cc:3677
              acc[1] += prm.G_c / 2.0 * ((pf - 1) * (pf - 1) / prm.alpha_eps + prm.alpha_eps * gg) * JxW;              // cracks.cc:3679-3680
              acc[2] += ug * JxW;                                                                                      // cracks.cc:3587
            }
        }
      // block reduction in a fixed order: lanes by xor-shuffles, waves through LDS
      __shared__ double s_red[4][3];
#pragma unroll
      for (int k = 0; k < 3; ++k)
        {
          double r = acc[k];
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1)
            r += __shfl_xor(r, off);
          if ((threadIdx.x & 63) == 0)
            s_red[threadIdx.x >> 6][k] = r;
        }
      __syncthreads();
      if (threadIdx.x < 3)
        partial[(long long)blockIdx.x * 3 + threadIdx.x] =
          ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
    }

    __global__ __launch_bounds__(256) void k_reduce3(const double *__restrict__ partial, long long n, double *__restrict__ out)
    {
      // second stage: each thread sums a strided slice, then the same fixed-order block reduction
      __shared__ double s_red[4][3];
      for (int k = 0; k < 3; ++k)
        {
          double r = 0.0;
          for (long long i = threadIdx.x; i < n; i += 256)
            r += partial[i * 3 + k];
          for (int off = 32; off >= 1; off >>= 1)
            r += __shfl_xor(r, off);
          if ((threadIdx.x & 63) == 0)
            s_red[threadIdx.x >> 6][k] = r;
        }
      __syncthreads();
      if (threadIdx.x < 3)
        out[threadIdx.x] = ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
    }


    // ---- norms of a residual vector with the constrained lines zeroed (constraints_update.set_zero, cracks.cc:2791-2794,
    // 2947-2949): thread <-> owned node, grid-stride over a grid that depends on n_owned only; fixed-order reductions
    constexpr int NORM_BLOCKS_MAX = 2048;
    template <int dim>
    __global__ __launch_bounds__(256) void k_residual_norms(DevView v, const double *__restrict__ res, double *__restrict__ partial /* [gridDim.x][2] */)
    {
      double sq = 0.0, mx = 0.0;
      for (long long P = (long long)blockIdx.x * 256 + threadIdx.x; P < v.n_owned; P += (long long)gridDim.x * 256)
        {
          const unsigned f = v.node_flags[P];
          const bool hanging = v.hn_index && v.hn_index[P] >= 0;
#pragma unroll
          for (int c = 0; c <= dim; ++c)
            {
              const double r = (hanging || ((f >> c) & 1u)) ? 0.0 : res[dof_of<dim>(v, (int)P, c)];
              sq = fma(r, r, sq);
              mx = fmax(mx, fabs(r));
            }
        }
      __shared__ double s_red[4][2];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1)
        {
          sq += __shfl_xor(sq, off);
          mx = fmax(mx, __shfl_xor(mx, off));
        }
      if ((threadIdx.x & 63) == 0)
        {
          s_red[threadIdx.x >> 6][0] = sq;
          s_red[threadIdx.x >> 6][1] = mx;
        }
      __syncthreads();
      if (threadIdx.x == 0)
        {
          partial[2 * (long long)blockIdx.x] = ((s_red[0][0] + s_red[1][0]) + s_red[2][0]) + s_red[3][0];
          partial[2 * (long long)blockIdx.x + 1] = fmax(fmax(s_red[0][1], s_red[1][1]), fmax(s_red[2][1], s_red[3][1]));
        }
    }
    __global__ __launch_bounds__(256) void k_reduce_norms(const double *__restrict__ partial, int n, double *__restrict__ out)
    {
      __shared__ double s_red[4][2];
      double sq = 0.0, mx = 0.0;
      for (int i = threadIdx.x; i < n; i += 256)
        {
          sq += partial[2 * i];
          mx = fmax(mx, partial[2 * i + 1]);
        }
      for (int off = 32; off >= 1; off >>= 1)
        {
          sq += __shfl_xor(sq, off);
          mx = fmax(mx, __shfl_xor(mx, off));
        }
      if ((threadIdx.x & 63) == 0)
        {
          s_red[threadIdx.x >> 6][0] = sq;
          s_red[threadIdx.x >> 6][1] = mx;
        }
      __syncthreads();
      if (threadIdx.x == 0)
        {
          const double s2 = ((s_red[0][0] + s_red[1][0]) + s_red[2][0]) + s_red[3][0];
          out[0] = sqrt(s2);
          out[1] = fmax(fmax(s_red[0][1], s_red[1][1]), fmax(s_red[2][1], s_red[3][1]));
          out[2] = s2;
        }
    }

    int fail(pfm_ctx *c, int code, const std::string &msg)
    {
      if (c)
        c->err = msg;
      return code;
    }
  } // namespace
} // namespace pfm

using namespace pfm;

extern "C"
{
  int pfm_diag_mass_device(pfm_ctx *c, double *d_mass)
  {
    if (!c || !d_mass)
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    if (hipMemsetAsync(d_mass, 0, sizeof(double) * (size_t)c->v.n_owned, c->stream) != hipSuccess)
      return fail(c, PFM_ERR_HIP, "diag_mass memset");
    const int nv = 1 << c->v.dim;
    const long long n = c->v.n_cells * nv;
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (nb)
      {
        if (c->v.dim == 2)
          hipLaunchKernelGGL(k_diag_mass<2>, dim3(nb), dim3(256), 0, c->stream, c->v, d_mass);
        else
          hipLaunchKernelGGL(k_diag_mass<3>, dim3(nb), dim3(256), 0, c->stream, c->v, d_mass);
      }
    return hipGetLastError() == hipSuccess ? PFM_OK : fail(c, PFM_ERR_HIP, "k_diag_mass launch");
  }

  int pfm_active_set_device(pfm_ctx *c, const double *d_residual_total, const double *d_mass, double c_const,
                            double *d_solution, const double *d_old_solution, int32_t *d_cycle_counter, int64_t *counts)
  {
    if (!c || !d_residual_total || !d_mass || !d_solution || !d_old_solution || !d_cycle_counter || !counts)
      return PFM_ERR_BAD_ARG;
    if (c->v.n_owned != c->v.n_nodes && c->v.hn_index)
      return fail(c, PFM_ERR_UNSUPPORTED, "hanging nodes on a partitioned mesh");
    (void)hipSetDevice(c->device);
    if (!c->d_counts)
      {
        if (hipMalloc((void **)&c->d_counts, 3 * sizeof(unsigned long long)) != hipSuccess)
          return fail(c, PFM_ERR_NOMEM, "hipMalloc counts");
        c->allocs.push_back(c->d_counts);
      }
    if (hipMemsetAsync(c->d_counts, 0, 3 * sizeof(unsigned long long), c->stream) != hipSuccess)
      return fail(c, PFM_ERR_HIP, "counts memset");
    const unsigned nb = (unsigned)((c->v.n_owned + 255) / 256); // 0 on a rank that owns no node: nothing to launch
    uint8_t *flags = const_cast<uint8_t *>(c->v.node_flags);
    if (nb == 0)
      ;
    else if (c->v.dim == 2)
      hipLaunchKernelGGL(k_active_set<2>, dim3(nb), dim3(256), 0, c->stream, c->v, flags, d_residual_total, d_mass, c_const,
                         d_solution, d_old_solution, d_cycle_counter, c->d_counts);
    else
      hipLaunchKernelGGL(k_active_set<3>, dim3(nb), dim3(256), 0, c->stream, c->v, flags, d_residual_total, d_mass, c_const,
                         d_solution, d_old_solution, d_cycle_counter, c->d_counts);
    if (c->v.hn_index && nb)
      {
        // we might have changed values of the solution, so fix the hanging nodes (cracks.cc:2888-2890)
        const long long n = (long long)c->v.n_owned * (c->v.dim + 1);
        const unsigned nbh = (unsigned)((n + 255) / 256);
        if (c->v.dim == 2)
          hipLaunchKernelGGL(k_distribute_hanging<2>, dim3(nbh), dim3(256), 0, c->stream, c->v, d_solution);
        else
          hipLaunchKernelGGL(k_distribute_hanging<3>, dim3(nbh), dim3(256), 0, c->stream, c->v, d_solution);
      }
    if (hipGetLastError() != hipSuccess)
      return fail(c, PFM_ERR_HIP, "k_active_set launch");
    unsigned long long h[3];
    if (hipMemcpyAsync(h, c->d_counts, sizeof(h), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
      return fail(c, PFM_ERR_HIP, "counts copy");
    counts[0] = (int64_t)h[0];
    counts[1] = (int64_t)h[1];
    counts[2] = h[2] != 0; // number of waves that saw a change -> flag
    return PFM_OK;
  }

  int pfm_get_constraints(pfm_ctx *c, uint8_t *node_flags)
  {
    if (!c || !node_flags)
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    if (hipMemcpyAsync(node_flags, c->v.node_flags, (size_t)c->v.n_nodes, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
      return fail(c, PFM_ERR_HIP, "get_constraints");
    return PFM_OK;
  }


  int pfm_residual_norms(pfm_ctx *c, const double *d_residual, double *out)
  {
    if (!c || !out || (!d_residual && c->v.n_owned > 0))
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    const long long nbl = ((long long)c->v.n_owned + 255) / 256;
    const unsigned nb = (unsigned)std::min<long long>(nbl, NORM_BLOCKS_MAX);
    if (!c->d_norm_partial)
      {
        if (hipMalloc((void **)&c->d_norm_partial, sizeof(double) * (2 * (size_t)NORM_BLOCKS_MAX + 4)) != hipSuccess)
          return fail(c, PFM_ERR_NOMEM, "hipMalloc norm partial sums");
        c->allocs.push_back(c->d_norm_partial);
      }
    double *d_out = c->d_norm_partial + 2 * (size_t)NORM_BLOCKS_MAX;
    if (nb)
      {
        if (c->v.dim == 2)
          hipLaunchKernelGGL(k_residual_norms<2>, dim3(nb), dim3(256), 0, c->stream, c->v, d_residual, c->d_norm_partial);
        else
          hipLaunchKernelGGL(k_residual_norms<3>, dim3(nb), dim3(256), 0, c->stream, c->v, d_residual, c->d_norm_partial);
      }
    hipLaunchKernelGGL(k_reduce_norms, dim3(1), dim3(256), 0, c->stream, c->d_norm_partial, (int)nb, d_out);
    if (hipGetLastError() != hipSuccess)
      return fail(c, PFM_ERR_HIP, "k_residual_norms launch");
    if (hipMemcpyAsync(out, d_out, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
      return fail(c, PFM_ERR_HIP, "residual norms copy");
    return PFM_OK;
  }

  int pfm_functionals(pfm_ctx *c, const uint8_t *cell_owned, double *out)
  {
    return pfm_functionals_material(c, cell_owned, nullptr, nullptr, out);
  }

  int pfm_functionals_material(pfm_ctx *c, const uint8_t *cell_owned, const double *cell_lambda, const double *cell_mu,
                               double *out)
  {
    if (!c || !out || ((cell_lambda == nullptr) != (cell_mu == nullptr)))
      return PFM_ERR_BAD_ARG;
    if (!c->have_params)
      return fail(c, PFM_ERR_BAD_ARG, "pfm_set_params has not been called");
    (void)hipSetDevice(c->device);
    const unsigned nb = (unsigned)((c->v.n_cells + 255) / 256);
    if (!c->d_partial || c->n_partial < (int64_t)nb)
      {
        double *p = nullptr;
        if (hipMalloc((void **)&p, sizeof(double) * 3 * ((size_t)nb + 1)) != hipSuccess)
          return fail(c, PFM_ERR_NOMEM, "hipMalloc partial sums");
        c->allocs.push_back(p);
        c->d_partial = p;
        c->n_partial = nb;
      }
    uint8_t *d_owned = nullptr;
    if (cell_owned)
      {
        if (!c->d_cell_owned)
          {
            if (hipMalloc((void **)&c->d_cell_owned, (size_t)std::max<long long>(c->v.n_cells, 1)) != hipSuccess)
              return fail(c, PFM_ERR_NOMEM, "hipMalloc cell mask");
            c->allocs.push_back(c->d_cell_owned);
          }
        if (hipMemcpyAsync(c->d_cell_owned, cell_owned, (size_t)c->v.n_cells, hipMemcpyHostToDevice, c->stream) != hipSuccess)
          return fail(c, PFM_ERR_HIP, "cell mask upload");
        d_owned = c->d_cell_owned;
      }
    double *d_lam = nullptr, *d_mu = nullptr;
    if (cell_lambda && c->v.n_cells > 0)
      {
        if (!c->d_func_mat)
          {
            if (hipMalloc((void **)&c->d_func_mat, 2 * sizeof(double) * (size_t)c->v.n_cells) != hipSuccess)
              return fail(c, PFM_ERR_NOMEM, "hipMalloc material override");
            c->allocs.push_back(c->d_func_mat);
            c->device_bytes += (int64_t)(2 * sizeof(double) * (size_t)c->v.n_cells);
          }
        d_lam = c->d_func_mat;
        d_mu = c->d_func_mat + c->v.n_cells;
        if (hipMemcpyAsync(d_lam, cell_lambda, sizeof(double) * (size_t)c->v.n_cells, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(d_mu, cell_mu, sizeof(double) * (size_t)c->v.n_cells, hipMemcpyHostToDevice, c->stream) != hipSuccess)
          return fail(c, PFM_ERR_HIP, "material override upload");
      }
    double *d_out = c->d_partial + 3 * (size_t)nb;
    if (nb)
      {
        if (c->v.dim == 2)
          hipLaunchKernelGGL(k_functionals<2>, dim3(nb), dim3(256), 0, c->stream, c->v, c->prm, d_owned, d_lam, d_mu, c->d_partial);
        else
          hipLaunchKernelGGL(k_functionals<3>, dim3(nb), dim3(256), 0, c->stream, c->v, c->prm, d_owned, d_lam, d_mu, c->d_partial);
      }
    hipLaunchKernelGGL(k_reduce3, dim3(1), dim3(256), 0, c->stream, c->d_partial, (long long)nb, d_out);
    if (hipGetLastError() != hipSuccess)
      return fail(c, PFM_ERR_HIP, "k_functionals launch");
    if (hipMemcpyAsync(out, d_out, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
      return fail(c, PFM_ERR_HIP, "functionals copy");
    return PFM_OK;
  }
}
