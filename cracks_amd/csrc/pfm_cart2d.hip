// pfm_cart2d.hip — 2-D row-owner kernel for uniform Cartesian boxes: Jacobian + residual (cracks.cc:2200-2464) of the
// 2-D Sneddon configurations (tests/sneddon_2d_1.prm on a uniform mesh, BASELINE config 2 with the matrix).  Runs with
// the stress split of cracks.cc:2294 active stay on the general family: a row owner would evaluate the linearised split
// of every trial dof 12 times (4 cells around the node x 3 row components to fit the registers) where the general
// family's quad evaluates it once (pfm_kernels.hip); the launcher refuses them and the host routes them there.
//
// Work is assigned by output row: thread <-> owned node.  The thread visits the (up to) 4 cells around its node and
// integrates, per cell, only the 3 rows of its own vertex -- the loop body of cracks.cc:2308-2432 for j = (a, c) -- into
// 81 register accumulators (9 neighbour slots x 3 row components x 3 column components) in a fixed cell order, applies the
// constraints as masks and writes every value of its rows exactly once: no atomics, no zeroing pass, bitwise reproducible.
// Every q-point state is evaluated by the 4 threads around the cell (4x redundant; in 2-D the state is ~60 flops, the
// entries ~400 per q-point) -- the price for not staging anything: the kernel has no LDS and no barrier.
//
// The 3-D family (pfm_cart_uu3/phi4) sum-factorises the element matrix; this kernel integrates the reference's formulas
// directly (9 q-points, the unsplit law written out).
#include "pfm_internal.h"
#include "pfm_cart_common.h"

#include <hip/hip_runtime.h>
#include <type_traits>

namespace pfm
{
  namespace
  {
    struct Vals2
    {
      double *b[4];
    };

    struct Cell2 // nodal data of one cell: [vertex]
    {
      double u[2][4], ph[4], pho[4], phoo[4];
      double lam, mu; // Lame coefficients of the cell (cracks.cc:2207-2216)
    };

    struct Prm2 // resolved scalars
    {
      double lam, mu, kappa, eps, Gc, p, aB1, penal_fac, tfac, ihx, ihy, vol;
      int monolithic, use_old;
    };

    // Rows of vertex A of one cell: the general kernel's loop body (pfm_kernels.hip: k_assemble_general) for constant
    // geometry J = diag(h).  Sink receives uu(b, c, d, x), pu(b, d, x), pp(b, x), r(c, x).
    template <int A, bool FULL, class Sink>
    __device__ __forceinline__ void cell_rows2d(const Cell2 &C, const Prm2 &P, Sink &out)
    {
#pragma unroll 1
      for (int q = 0; q < 9; ++q)
        {
          const int qx = q % 3, qy = q / 3;
          const double nx[2] = {c_g1.n[0][qx], c_g1.n[1][qx]}, ny[2] = {c_g1.n[0][qy], c_g1.n[1][qy]};
          const double JxW = P.vol * (c_g1.w[qx] * c_g1.w[qy]);
          double N[4], gN[4][2];
#pragma unroll
          for (int b = 0; b < 4; ++b)
            {
              N[b] = nx[b & 1] * ny[b >> 1];
              gN[b][0] = ((b & 1) ? P.ihx : -P.ihx) * ny[b >> 1];
              gN[b][1] = ((b >> 1) ? P.ihy : -P.ihy) * nx[b & 1];
            }
          double gu[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, gpf[2] = {0.0, 0.0}, pf = 0.0, pfo = 0.0, pfoo = 0.0;
#pragma unroll
          for (int b = 0; b < 4; ++b)
            {
              pf += C.ph[b] * N[b];
              pfo += C.pho[b] * N[b];
              pfoo += C.phoo[b] * N[b];
#pragma unroll
              for (int d = 0; d < 2; ++d)
                {
                  gpf[d] += C.ph[b] * gN[b][d];
                  gu[0][d] += C.u[0][b] * gN[b][d];
                  gu[1][d] += C.u[1][b] * gN[b][d];
                }
            }
          const double Na = N[A], gNa[2] = {gN[A][0], gN[A][1]};
          // ---- q-point state, cracks.cc:2248-2306
          if (P.monolithic)
            {
              pf = fmax(0.0, pf);
              pfo = fmax(0.0, pfo);
              pfoo = fmax(0.0, pfoo);
            }
          const double pf_minus_old_plus = fmax(0.0, pf - pfo);
          double pfx = pfoo + P.tfac * (pfo - pfoo);
          if (pfx <= 0.0)
            pfx = 0.0;
          if (pfx >= 1.0)
            pfx = 1.0;
          if (P.use_old)
            pfx = pfo;
          const double g = (1 - P.kappa) * pfx * pfx + P.kappa;
          double E[2][2], trE = 0.0, divu = 0.0;
#pragma unroll
          for (int i = 0; i < 2; ++i)
            {
              divu += gu[i][i];
#pragma unroll
              for (int j = 0; j < 2; ++j)
                E[i][j] = 0.5 * (gu[i][j] + gu[j][i]);
              trE += E[i][i];
            }
          double sp[2][2]; // sigma+ = lambda tr(E) I + 2 mu E, sigma- = 0 (no split: cracks.cc:2299-2305)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              sp[i][j] = C.lam * trE * (i == j ? 1.0 : 0.0) + 2 * C.mu * E[i][j];
          double spE = 0.0;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              spE += sp[i][j] * E[i][j];

          // ---- Jacobian rows of vertex A, cracks.cc:2308-2389
          if constexpr (FULL)
            {
              // unsplit law with the tensors written out (see k_assemble_general): the linearised stress of trial dof
              // (b, d) against the test gradient is  lambda gN_b[d] gN_a[c] + mu (gN_b[c] gN_a[d] + delta_cd gN_b.gN_a),
              // and sigma+_LinU : E = sigma+ : E_LinU = sum_k sigma+[d][k] gN_b[k]
              const double gw = g * JxW;
              const double LA[2] = {C.lam * gw * gNa[0], C.lam * gw * gNa[1]}, MA[2] = {C.mu * gw * gNa[0], C.mu * gw * gNa[1]};
              const double mgw = C.mu * gw;
              const double cpu = 2.0 * (1 - P.kappa) * pf * Na * JxW, cdiv = 2.0 * P.aB1 * P.p * pf * Na * JxW;
              const double cpp = ((1 - P.kappa) * spE + P.Gc / P.eps) * Na * JxW, cgg = P.Gc * P.eps * JxW;
              const double cdu = 2.0 * P.aB1 * P.p * divu * Na * JxW, cpen = P.penal_fac * Na * JxW;
              const bool pen_on = !((pf - pfo) < 0.0); // shadowed variable, cracks.cc:2311-2315
              static_for<4>([&](auto Bb) __attribute__((always_inline)) {
                constexpr int b = decltype(Bb)::value;
                const double t = gN[b][0] * gNa[0] + gN[b][1] * gNa[1];
#pragma unroll
                for (int d = 0; d < 2; ++d)
                  {
                    const double sv = sp[d][0] * gN[b][0] + sp[d][1] * gN[b][1];
                    out.pu(std::integral_constant<int, b>{}, d, cpu * sv - cdiv * gN[b][d]);
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                      out.uu(std::integral_constant<int, b>{}, c, d, LA[c] * gN[b][d] + MA[d] * gN[b][c] + (c == d ? mgw * t : 0.0));
                  }
                out.pp(std::integral_constant<int, b>{}, cpen * (pen_on ? N[b] : 0.0) + ((cpp - cdu) * N[b] + cgg * t));
              });
            }
          // ---- residual rows of vertex A, cracks.cc:2393-2432
#pragma unroll
          for (int c = 0; c < 2; ++c)
            {
              double t = 0.0;
#pragma unroll
              for (int k = 0; k < 2; ++k)
                t += g * sp[c][k] * gNa[k];
              out.r(c, -(t - P.aB1 * P.p * pfx * pfx * gNa[c]) * JxW);
            }
          {
            const double gg = gpf[0] * gNa[0] + gpf[1] * gNa[1];
            double x = -P.penal_fac * pf_minus_old_plus * Na * JxW;
            x -= ((1.0 - P.kappa) * spE * pf * Na - P.Gc / P.eps * (1.0 - pf) * Na + P.Gc * P.eps * gg -
                  2.0 * P.aB1 * P.p * pf * divu * Na) *
                 JxW;
            out.r(2, x);
          }
        }
    }

    // sum of |diagonal| of the rows of vertex A (mean |diagonal| of the element matrix: deal.II's placeholder when a
    // constrained row's own diagonal entry vanishes)
    struct DiagSink2
    {
      int a;
      double d[3] = {0.0, 0.0, 0.0};
      template <class B>
      __device__ void uu(B, int c, int dd, double x)
      {
        if (B::value == a && c == dd)
          d[c] += x;
      }
      template <class B>
      __device__ void pu(B, int, double)
      {}
      template <class B>
      __device__ void pp(B, double x)
      {
        if (B::value == a)
          d[2] += x;
      }
      __device__ void r(int, double) {}
    };

    __device__ __forceinline__ double element_mean_abs_diag(const Cell2 &C, const Prm2 &P)
    {
      double s = 0.0;
      static_for<4>([&](auto Aa) __attribute__((always_inline)) {
        DiagSink2 ds;
        ds.a = decltype(Aa)::value;
        cell_rows2d<decltype(Aa)::value, true>(C, P, ds);
        s += fabs(ds.d[0]) + fabs(ds.d[1]) + fabs(ds.d[2]);
      });
      return s / 12.0;
    }

    // accumulates the rows of vertex A of one cell into the node's 9 x 3 x 3 slots: vertex b sits at lattice offset
    // (b_x - a_x, b_y - a_y) of the node
    template <int A>
    struct RowSink2
    {
      static constexpr int ax = A & 1, ay = A >> 1;
      double (*acc)[3][3];
      double *R;
      double kd[3] = {0.0, 0.0, 0.0}; // this cell's own diagonal entries K_e[(A,c),(A,c)]
      template <class B>
      __device__ __forceinline__ void uu(B, int c, int d, double x)
      {
        constexpr int b = B::value, o = ((b & 1) - ax + 1) + 3 * ((b >> 1) - ay + 1);
        acc[o][c][d] += x;
        if (b == A && c == d)
          kd[c] += x;
      }
      template <class B>
      __device__ __forceinline__ void pu(B, int d, double x)
      {
        constexpr int b = B::value, o = ((b & 1) - ax + 1) + 3 * ((b >> 1) - ay + 1);
        acc[o][2][d] += x;
      }
      template <class B>
      __device__ __forceinline__ void pp(B, double x)
      {
        constexpr int b = B::value, o = ((b & 1) - ax + 1) + 3 * ((b >> 1) - ay + 1);
        acc[o][2][2] += x;
        if (b == A)
          kd[2] += x;
      }
      __device__ __forceinline__ void r(int c, double x) { R[c] += x; }
    };

    template <bool FULL>
    __global__ __launch_bounds__(128) void k_cart2d_rows(DevView v, CartView cv, Prm2 P, Vals2 vals, double *__restrict__ res_pde,
                                                         double *__restrict__ res_tot, int write_total, int total_via_update)
    {
      const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1;
      const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      if (id >= (long long)OWX * OWY)
        return;
      const int i = cv.o0[0] + (int)(id % OWX), j = cv.o0[1] + (int)(id / OWX);
      if (cart_tile_skipped(cv, cart_range_has_ghost(cv, 0, i - 1, i + 1) || cart_range_has_ghost(cv, 1, j - 1, j + 1)))
        return; // overlapped assembly: the other launch owns this node's rows
      const int row = cart_local_id(cv, i, j, 0);
      const unsigned fP = v.node_flags[row];

      double acc[FULL ? 9 : 1][3][3]; // [slot o = (ox+1) + 3 (oy+1)][row comp][col comp]
      double R[3] = {0.0, 0.0, 0.0}, dg[3] = {0.0, 0.0, 0.0};
      if constexpr (FULL)
        {
#pragma unroll
          for (int o = 0; o < 9; ++o)
#pragma unroll
            for (int c = 0; c < 3; ++c)
              acc[o][c][0] = acc[o][c][1] = acc[o][c][2] = 0.0;
        }
      // the 4 cells around the node, in the order of a lexicographic cell loop: the node is vertex A = 3, 2, 1, 0 of them
      static_for<4>([&](auto Ee) __attribute__((always_inline)) {
        constexpr int A = 3 - decltype(Ee)::value, ax = A & 1, ay = A >> 1;
        const int ci = i - ax, cj = j - ay;
        if (ci < 0 || ci >= cv.NX - 1 || cj < 0 || cj >= cv.NY - 1)
          return;
        Cell2 C;
        C.lam = P.lam;
        C.mu = P.mu;
        if (cv.cell_lam)
          {
            C.lam = cv.cell_lam[ci + (long long)(cv.NX - 1) * cj];
            C.mu = cv.cell_mu[ci + (long long)(cv.NX - 1) * cj];
          }
#pragma unroll
        for (int b = 0; b < 4; ++b)
          {
            const int n = cart_local_id(cv, ci + (b & 1), cj + (b >> 1), 0);
            C.u[0][b] = v.u[0][n];
            C.u[1][b] = v.u[1][n];
            C.ph[b] = v.phi[n];
            C.pho[b] = v.phi_old[n];
            C.phoo[b] = v.phi_oldold[n];
          }
        RowSink2<A> sink;
        sink.acc = acc;
        sink.R = R;
        cell_rows2d<A, FULL>(C, P, sink);
        if constexpr (FULL)
          {
            if (fP & 7u) // a constrained row needs its placeholder: sum_e (|K_e,aa| != 0 ? |K_e,aa| : mean |diag K_e|)
              {
                const double k0 = fabs(sink.kd[0]), k1 = fabs(sink.kd[1]), k2 = fabs(sink.kd[2]);
                double avg = 0.0;
                if (((fP & 1u) && k0 == 0.0) || ((fP & 2u) && k1 == 0.0) || ((fP & 4u) && k2 == 0.0))
                  avg = element_mean_abs_diag(C, P);
                dg[0] += k0 != 0.0 ? k0 : avg;
                dg[1] += k1 != 0.0 ? k1 : avg;
                dg[2] += k2 != 0.0 ? k2 : avg;
              }
          }
      });
      // ---- constrained scatter as masks (cracks.cc:2439-2464)
      const bool blocked = v.layout == PFM_LAYOUT_BLOCKED;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        {
          const bool con = (fP >> c) & 1u;
          const long long di = blocked ? (c < 2 ? (long long)row * 2 + c : (long long)v.n_owned * 2 + row) : (long long)row * 3 + c;
          res_pde[di] = con ? 0.0 : R[c];
          if (write_total)
            res_tot[di] = (con && total_via_update) ? 0.0 : R[c];
        }
      if constexpr (FULL)
        {
          const unsigned mask = cv.nbr_mask[row];
          const long long off = v.nadj_ptr[row];
          const int deg = __popc(mask & 0x1ffu);
#pragma unroll
          for (int o = 0; o < 9; ++o)
            {
              if (!((mask >> o) & 1u))
                continue;
              int sl = __popc(mask & ((1u << o) - 1u));
              if (mask >> 31)
                sl = cv.row_perm[off + sl];
              const int q = cart_local_id(cv, i + (o % 3) - 1, j + (o / 3) - 1, 0);
              const unsigned fQ = v.node_flags[q];
#pragma unroll
              for (int c = 0; c < 3; ++c)
                {
                  const bool rcon = (fP >> c) & 1u;
#pragma unroll
                  for (int d = 0; d < 3; ++d)
                    {
                      double x = acc[o][c][d];
                      if (c < 2 && d == 2)
                        x = 0.0; // (u,phi) block: structurally zero (cracks.cc:2333-2337)
                      if (rcon)
                        x = (o == 4 && c == d) ? dg[c] : 0.0;
                      else if ((fQ >> d) & 1u)
                        x = 0.0;
                      double *dst;
                      if (!blocked)
                        dst = vals.b[0] + (9 * off + (long long)c * 3 * deg + (long long)sl * 3 + d);
                      else if (c < 2)
                        dst = d < 2 ? vals.b[0] + (4 * off + (long long)c * 2 * deg + (long long)sl * 2 + d)
                                    : vals.b[1] + (2 * off + (long long)c * deg + sl);
                      else
                        dst = d < 2 ? vals.b[2] + (2 * off + (long long)sl * 2 + d) : vals.b[3] + (off + sl);
                      *dst = x;
                    }
                }
            }
        }
    }
  } // namespace

  // 2-D cartesian boxes: Jacobian + residual without the stress split (the plain 2-D residual has its own kernel in
  // pfm_cart.hip)
  int launch_cart2d(const DevView &v, const CartView &cv, const pfm_params &p, int residual_only, double *const *d_values,
                    double *res_pde, double *res_tot, hipStream_t s)
  {
    int rc = ensure_g1();
    if (rc)
      return rc;
    Prm2 P{};
    P.lam = p.lambda;
    P.mu = p.mu;
    P.kappa = p.constant_k;
    P.eps = p.alpha_eps;
    P.Gc = p.G_c;
    P.p = p.pressure;
    P.aB1 = p.alpha_biot - 1.0;
    double gamma = p.gamma_penal;
    if (p.outer_solver == PFM_SOLVER_SIMPLE_MONOLITHIC && p.timestep_number < 1)
      gamma = 0.0; // cracks.cc:2141-2144
    P.penal_fac = gamma / p.timestep * 1.0 / (cv.h[0] * cv.h[0] + cv.h[1] * cv.h[1]); // cell->diameter()^2, cracks.cc:2370
    P.tfac = (p.time - (p.time - p.old_timestep - p.old_old_timestep)) /
             (p.time - p.old_timestep - (p.time - p.old_timestep - p.old_old_timestep));
    P.ihx = 1.0 / cv.h[0];
    P.ihy = 1.0 / cv.h[1];
    P.vol = cv.h[0] * cv.h[1];
    P.monolithic = p.outer_solver == PFM_SOLVER_SIMPLE_MONOLITHIC;
    P.use_old = p.use_old_timestep_pf;
    const int total_via_update = p.outer_solver != PFM_SOLVER_ACTIVE_SET;
    const bool split = p.decompose_stress_matrix > 0 && p.timestep_number > 0; // cracks.cc:2294
    Vals2 vals{};
    if (!residual_only)
      for (int b = 0; b < (v.layout == PFM_LAYOUT_BLOCKED ? 4 : 1); ++b)
        vals.b[b] = d_values[b];
    const long long n = (long long)(cv.o1[0] - cv.o0[0] + 1) * (cv.o1[1] - cv.o0[1] + 1);
    const unsigned nb = (unsigned)((n + 127) / 128);
    if (nb == 0)
      return PFM_OK;
#define PFM_L2D(F) hipLaunchKernelGGL((k_cart2d_rows<F>), dim3(nb), dim3(128), 0, s, v, cv, P, vals, res_pde, res_tot, residual_only, total_via_update)
    if (split)
      return PFM_ERR_UNSUPPORTED; // stress-split runs stay on the general family (header of this file, pfm_host.cpp)
    if (residual_only)
      PFM_L2D(false);
    else
      PFM_L2D(true);
#undef PFM_L2D
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }
} // namespace pfm
