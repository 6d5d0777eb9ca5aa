// pfm_cart.hip — gfx950 kernels of the assembly hot path, "cartesian" family.
//
// Fast path for uniform Cartesian boxes (every BASELINE benchmark mesh: J = diag(h)).
// MI355X-first design, not a translation of the reference's cell loop + scatter:
//
//   ROW OWNER.  Work is assigned by OUTPUT ROW (owned node), not by cell.  Every CSR value
//   and every residual entry is computed completely by one lane and written exactly once:
//   no atomics, no colouring, no zeroing pass, bitwise deterministic, and the result of a
//   row does not depend on how the mesh is cut over GPUs.  The unique-touch traffic model
//   of SURVEY.md §8(d) (3592 B/cell) becomes the actual traffic.
//
//   SUM FACTORISATION.  On a Cartesian cell dN_a/dx_c = (+-1/h_c) * prod_{k!=c} n_{a_k}(xi_k),
//   so the element matrix (cracks.cc:2353-2387) collapses onto a handful of 1-D moment
//   tables of the q-point weights.  The (u,u) block of one hex (576 entries, 27 648 loop
//   bodies in the reference) is determined by 63 numbers:
//       A^c[g_i][g_j]     = sum_q w g(q) m_{g_i}(q_i) m_{g_j}(q_j)            (3 x 9)
//       T^{cd}[al][be][g] = sum_q w g(q) n_al(q_c) n_be(q_d) m_g(q_e)         (3 x 12)
//   with n_0 = 1-xi, n_1 = xi, m_00 = n_0^2, m_01 = n_0 n_1, m_11 = n_1^2 at the 3 Gauss
//   points, g(q) = (1-kappa) pf_extra^2 + kappa.  A workgroup computes these once per cell
//   of its tile into LDS (SoA over cells: conflict-free), then its lanes, one per node,
//   gather K[(a,c),(b,d)] = lambda G^{cd}_{ab} + mu G^{dc}_{ab} + mu delta_cd tr G_ab for all
//   neighbour slots with compile-time LDS offsets, stage the rows in LDS and stream them out
//   with fully coalesced stores.
//
// Arithmetic is the reference's (cracks.cc:2248-2432) re-associated; parity with the CPU
// oracle is at round-off level (tests/test_gpu_parity.py, tolerance 1e-12).
#include "pfm_internal.h"
#include "pfm_poly.h"
#include "pfm_dma.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace pfm
{
  namespace
  {
    // 1-D Gauss(3) data on [0,1]
    struct Tab1D
    {
      double n[2][3]; // n_0 = 1 - xi, n_1 = xi
      double w[3];
      double wn1[3]; // w n_1
    };
    __constant__ Tab1D c_t1;

    Tab1D make_tab1d()
    {
      Tab1D t{};
      const double gx[3] = {0.5 - 0.5 * 0.7745966692414834, 0.5, 0.5 + 0.5 * 0.7745966692414834};
      const double gw[3] = {5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0};
      for (int q = 0; q < 3; ++q)
        {
          t.n[0][q] = 1.0 - gx[q];
          t.n[1][q] = gx[q];
          t.w[q] = gw[q];
          t.wn1[q] = gw[q] * gx[q];
        }
      return t;
    }

    template <int dim>
    __device__ __forceinline__ long long dof_index_c(const DevView &v, int P, int c)
    {
      if (v.layout == PFM_LAYOUT_INTERLEAVED)
        return (long long)P * (dim + 1) + c;
      return c < dim ? (long long)P * dim + c : (long long)v.n_owned * dim + P;
    }

    struct Scal // per-launch scalars derived from pfm_params
    {
      double lam, mu, kappa, eps, Gc, p, aB1, gamma_fac, tfac;
      int monolithic, use_old, total_via_update;
      // uniform constants of residual_cell_poly, formed on the host so that they arrive in scalar registers
      double ih[3], vol, vih[3], geih[3]; // 1 / h_k, h_x h_y h_z, vol / h_k, G_c eps vol / h_k^2
      double c_g, c_pd, c_r2, c_r3;       // 1 - kappa, (alpha_B - 1) p, -2 (alpha_B - 1) p, G_c / eps
      double c_pdg, c_pdk;                // c_pd / c_g, c_pd kappa / c_g: the pressure term through the moments of g (residual_cell_poly)
    };

    Scal make_scal(const pfm_params &prm, const CartView &cv, int dim)
    {
      Scal s{};
      s.lam = prm.lambda;
      s.mu = prm.mu;
      s.kappa = prm.constant_k;
      s.eps = prm.alpha_eps;
      s.Gc = prm.G_c;
      s.p = prm.pressure;
      s.aB1 = prm.alpha_biot - 1.0;
      double gamma = prm.gamma_penal;
      if (prm.outer_solver == PFM_SOLVER_SIMPLE_MONOLITHIC && prm.timestep_number < 1)
        gamma = 0.0; // cracks.cc:2141-2144
      double diam2 = 0.0;
      for (int d = 0; d < dim; ++d)
        diam2 += cv.h[d] * cv.h[d];
      s.gamma_fac = gamma / prm.timestep * 1.0 / diam2; // cracks.cc:2370, 2419
      s.tfac = (prm.time - (prm.time - prm.old_timestep - prm.old_old_timestep)) /
               (prm.time - prm.old_timestep - (prm.time - prm.old_timestep - prm.old_old_timestep));
      s.monolithic = prm.outer_solver == PFM_SOLVER_SIMPLE_MONOLITHIC;
      s.use_old = prm.use_old_timestep_pf;
      s.total_via_update = prm.outer_solver != PFM_SOLVER_ACTIVE_SET;
      s.vol = 1.0;
      for (int d = 0; d < dim; ++d)
        s.vol *= cv.h[d];
      for (int d = 0; d < 3; ++d)
        {
          s.ih[d] = d < dim ? 1.0 / cv.h[d] : 0.0;
          s.vih[d] = s.vol * s.ih[d];
          s.geih[d] = (s.Gc * s.eps) * s.ih[d] * s.vih[d];
        }
      s.c_g = 1.0 - s.kappa;
      s.c_pd = s.aB1 * s.p;
      s.c_pdg = s.c_pd / s.c_g;
      s.c_pdk = s.c_pd * s.kappa / s.c_g;
      s.c_r2 = -2.0 * s.aB1 * s.p;
      s.c_r3 = s.Gc / s.eps;
      return s;
    }

    // local node id of lattice node (i,j,k): arithmetic for lexicographically numbered owned nodes (checked at
    // context creation), table look-up otherwise (ghost layers)
    __device__ __forceinline__ int cart_local_id3(const CartView &cv, int i, int j, int k)
    {
      if (cv.owned_lex && i >= cv.o0[0] && i <= cv.o1[0] && j >= cv.o0[1] && j <= cv.o1[1] && k >= cv.o0[2] && k <= cv.o1[2])
        return (i - cv.o0[0]) + (cv.o1[0] - cv.o0[0] + 1) * ((j - cv.o0[1]) + (cv.o1[1] - cv.o0[1] + 1) * (k - cv.o0[2]));
      return cv.local_of_box[i + (long long)cv.NX * (j + (long long)cv.NY * k)];
    }


    // =====================================================================================
    // 2-D residual, y-marching cell columns (cracks.cc:2393-2432), no LDS and no barrier.
    //
    // A wave (= a workgroup) owns 62 x-consecutive nodes over a chunk of node rows; lane l <-> the cell column between
    // nodes i0 - 1 + l and i0 + l (lane 63 only contributes its node's values).  Marching up in y, a lane loads ONE node
    // per row (the next row is requested before the current cell is evaluated), gets the right-hand vertices from lane
    // l + 1 (DPP shift), evaluates its cell once, keeps the part that belongs to the upper node row for the next step and
    // completes node row j from its own a_x = 0 parts and the a_x = 1 parts of lane l - 1.  Every cell is evaluated once
    // per chunk (the first-generation kernel, removed in round 4, evaluated it for both node rows it touches), every nodal value is
    // read once per tile, and a wave waits for one memory round trip per row that the evaluation of the previous row
    // covers.  LIN: one combined old phase field as in k_cart_residual3.
    // =====================================================================================
    // ---- round 4: one cell of k_cart_residual2m (LIN) without a quadrature loop -- residual_cell_poly (below) in 2-D: bilinear
    // fields, 9 discrete moments of the clamped pfx^2, M[i][j][c] = moments against t^i s^j
    __device__ __forceinline__ void residual_cell_poly2(const double (&lo)[4], const double (&lo1)[4], const double (&up)[4],
                                                        const double (&up1)[4], const Scal &S, double lam, double mu, double (&M)[2][2][3])
    {
      const double c_g = S.c_g, c_pd = S.c_pd, c_r2 = S.c_r2, c_r3 = S.c_r3, vol = S.vol;
      const double(&ih)[3] = S.ih;
      const double(&vih)[3] = S.vih;
      auto Madd = [&](auto Psi, auto Cc, double x) __attribute__((always_inline)) {
        constexpr int psi = decltype(Psi)::value, c = decltype(Cc)::value;
        M[psi & 1][psi >> 1][c] += x;
      };
      double Hg[9], HP4[4];
      {
        double W[4] = {lo[3], lo1[3], up[3], up1[3]}; // LIN: the combined old field
        monomials2(W);
        double Pq[9];
        poly_for<3>([&](auto Qx) __attribute__((always_inline)) {
          constexpr int qx = decltype(Qx)::value;
          const double x0 = fma(GqT<qx>::v, W[1], W[0]), x1 = fma(GqT<qx>::v, W[3], W[2]);
          poly_for<3>([&](auto Qy) __attribute__((always_inline)) {
            constexpr int qy = decltype(Qy)::value;
            double pfx = fma(GqT<qy>::v, x1, x0);
            if (!S.use_old)
              pfx = fmin(fmax(pfx, 0.0), 1.0); // cracks.cc:2270-2277
            Pq[qx + 3 * qy] = pfx * pfx;
          });
        });
        double A1[9];
        poly_for<3>([&](auto Pp) __attribute__((always_inline)) {
          constexpr int pp = decltype(Pp)::value;
#pragma unroll
          for (int qy = 0; qy < 3; ++qy)
            A1[pp + 3 * qy] = GqWt<0, pp>::v * Pq[3 * qy] + GqWt<1, pp>::v * Pq[3 * qy + 1] + GqWt<2, pp>::v * Pq[3 * qy + 2];
        });
        poly_for<9>([&](auto Mm) __attribute__((always_inline)) {
          constexpr int m = decltype(Mm)::value, pp = m % 3, qp = m / 3;
          const double hp = GqWt<0, qp>::v * A1[pp] + GqWt<1, qp>::v * A1[pp + 3] + GqWt<2, qp>::v * A1[pp + 6];
          constexpr double Im = GqMom<pp>::v * GqMom<qp>::v;
          Hg[m] = fma(c_g, hp, S.kappa * Im);
          if constexpr (pp < 2 && qp < 2)
            HP4[pp + 2 * qp] = hp;
        });
      }
      double F[2][4];
#pragma unroll
      for (int f = 0; f < 2; ++f)
        {
          F[f][0] = lo[f], F[f][1] = lo1[f], F[f][2] = up[f], F[f][3] = up1[f];
          monomials2(F[f]);
        }
      auto G = [&](auto Cc, auto Kc, auto Ic) __attribute__((always_inline)) -> double { // d_k u_c, coefficient idx
        constexpr int c = decltype(Cc)::value, k = decltype(Kc)::value, idx = decltype(Ic)::value;
        static_assert(!(idx & (1 << k)), "no such monomial in this derivative");
        return ih[k] * F[c][idx | (1 << k)];
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      constexpr int N2X = 0x5, N2Y = 0x3; // monomials without t: {1, s}; without s: {1, t}
      const double T[4] = {G(I0{}, I0{}, I0{}) + G(I1{}, I1{}, I0{}), G(I1{}, I1{}, I1{}), G(I0{}, I0{}, I2{}), 0.0}; // div u
      const double mu2 = 2.0 * mu;
      poly_for<2>([&](auto Cc) __attribute__((always_inline)) {
        poly_for<2>([&](auto Kc) __attribute__((always_inline)) {
          constexpr int c = decltype(Cc)::value, k = decltype(Kc)::value;
          constexpr int mc = c == 0 ? N2X : N2Y, mk = k == 0 ? N2X : N2Y;
          double sg[4]; // sigma_ck = lambda div u delta_ck + mu (d_k u_c + d_c u_k): coefficients 0, 1, 2
          poly_for<3>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int idx = decltype(Ic)::value;
            if constexpr (c == k)
              {
                if constexpr ((mc >> idx) & 1)
                  sg[idx] = fma(mu2, G(Cc, Cc, Ic), lam * T[idx]);
                else
                  sg[idx] = lam * T[idx];
              }
            else
              {
                constexpr bool hk = (mk >> idx) & 1, hc = (mc >> idx) & 1;
                if constexpr (hk && hc)
                  sg[idx] = mu * (G(Cc, Kc, Ic) + G(Kc, Cc, Ic));
                else if constexpr (hk)
                  sg[idx] = mu * G(Cc, Kc, Ic);
                else
                  sg[idx] = mu * G(Kc, Cc, Ic);
              }
          });
          poly_for<2>([&](auto Jc) __attribute__((always_inline)) {
            constexpr int midx = decltype(Jc)::value << (1 - k); // the monomial 1 or x_other
            double val = sg[0] * Hg[pow2_of(0, midx)];
            val = fma(sg[1], Hg[pow2_of(1, midx)], val);
            val = fma(sg[2], Hg[pow2_of(2, midx)], val);
            if constexpr (c == k)
              val = fma(-c_pd, HP4[midx], val);
            Madd(std::integral_constant<int, (1 << k) | midx>{}, Cc, vih[k] * val);
          });
        });
      });
      // phase-field row
      double Th[9];
#pragma unroll
      for (int m = 0; m < 9; ++m)
        Th[m] = 0.0;
      add_square2<0x7>(T, lam, Th);
      {
        const double P0[4] = {G(I0{}, I0{}, I0{}), 0.0, G(I0{}, I0{}, I2{}), 0.0};
        add_square2<N2X>(P0, mu2, Th);
        const double P1[4] = {G(I1{}, I1{}, I0{}), G(I1{}, I1{}, I1{}), 0.0, 0.0};
        add_square2<N2Y>(P1, mu2, Th);
        const double Sp[4] = {G(I0{}, I1{}, I0{}) + G(I1{}, I0{}, I0{}), G(I0{}, I1{}, I1{}), G(I1{}, I0{}, I2{}), 0.0};
        add_square2<0x7>(Sp, mu, Th); // mu t_01^2
      }
#pragma unroll
      for (int m = 0; m < 9; ++m)
        Th[m] *= c_g;
      Th[0] += c_r3;
      poly_for<3>([&](auto Ic) __attribute__((always_inline)) {
        constexpr int idx = decltype(Ic)::value;
        Th[pow2_of(idx, 0)] = fma(c_r2, T[idx], Th[pow2_of(idx, 0)]);
      });
      double B1[9], HT[9];
      poly_for<3>([&](auto Pp) __attribute__((always_inline)) {
        constexpr int pp = decltype(Pp)::value;
#pragma unroll
        for (int b = 0; b < 3; ++b)
          B1[pp + 3 * b] = GqMom<pp>::v * Th[3 * b] + GqMom<pp + 1>::v * Th[3 * b + 1] + GqMom<pp + 2>::v * Th[3 * b + 2];
      });
      poly_for<3>([&](auto Qp) __attribute__((always_inline)) {
        constexpr int qp = decltype(Qp)::value;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
          HT[pp + 3 * qp] = GqMom<qp>::v * B1[pp] + GqMom<qp + 1>::v * B1[pp + 3] + GqMom<qp + 2>::v * B1[pp + 6];
      });
      double Pf[4] = {lo[2], lo1[2], up[2], up1[2]};
      monomials2(Pf);
      poly_for<4>([&](auto Psi) __attribute__((always_inline)) {
        constexpr int psi = decltype(Psi)::value;
        constexpr double Ipsi = GqMom<(psi & 1)>::v * GqMom<(psi >> 1)>::v;
        double val = -c_r3 * Ipsi;
        poly_for<4>([&](auto Ic) __attribute__((always_inline)) {
          constexpr int idx = decltype(Ic)::value;
          val = fma(Pf[idx], HT[pow2_of(idx, psi)], val);
        });
        Madd(Psi, I2{}, vol * val);
      });
      poly_for<2>([&](auto Kc) __attribute__((always_inline)) {
        constexpr int k = decltype(Kc)::value;
        poly_for<2>([&](auto Jc) __attribute__((always_inline)) {
          constexpr int j = decltype(Jc)::value, midx = j << (1 - k);
          // d_k pf = ih_k (Pf[1 << k] + x_other Pf[3]); integral against the monomial midx
          constexpr double Ia = GqMom<j>::v, Ib = GqMom<j + 1>::v;
          const double val = fma(Pf[3], Ib, Pf[1 << k] * Ia);
          Madd(std::integral_constant<int, (1 << k) | midx>{}, I2{}, S.geih[k] * val);
        });
      });
    }

    constexpr int R2N = 62; // owned nodes per wave
    template <bool LIN>
    __global__ __launch_bounds__(64) void k_cart_residual2m(DevView v, CartView cv, Scal S, double *__restrict__ res_pde,
                                                            double *__restrict__ res_tot, int write_total, int zc)
    {
      constexpr int NF = LIN ? 4 : 5; // u_x u_y phi + (combined old field | phi_old phi_oldold)
      const int lane = threadIdx.x;
      const int OWX = cv.o1[0] - cv.o0[0] + 1;
      const int ntx = (OWX + R2N - 1) / R2N;
      const int tix = (int)(blockIdx.x % ntx), chunk = (int)(blockIdx.x / ntx);
      const int i = cv.o0[0] + tix * R2N - 1 + lane; // this lane's node column = left vertex of its cell column
      const int jA = cv.o0[1] + chunk * zc;
      const int jB = min(jA + zc, cv.o1[1] + 1); // node rows [jA, jB)
      {
        const int iA = cv.o0[0] + tix * R2N;
        if (cart_tile_skipped(cv, cart_range_has_ghost(cv, 0, iA - 1, iA + R2N) || cart_range_has_ghost(cv, 1, jA - 1, jB)))
          return; // overlapped assembly: the other launch owns this wave's chunk
      }
      const bool node_in = i >= 0 && i < cv.NX;
      const bool col_ok = lane < 63 && i >= 0 && i < cv.NX - 1;
      const bool owner = lane >= 1 && lane <= R2N && i <= cv.o1[0];
      const double ihx = 1.0 / cv.h[0], ihy = 1.0 / cv.h[1];
      const double vol = cv.h[0] * cv.h[1];

      auto load_row = [&](int j, double (&val)[NF]) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < NF; ++f)
          val[f] = 0.0;
        if (node_in && j >= 0 && j < cv.NY)
          {
            const int n = cart_local_id3(cv, i, j, 0);
            if (v.fused_solution) // pfm_assemble_nl_residual_device on a single-rank box (see k_cart_residual3)
              {
                const bool il = v.layout == PFM_LAYOUT_INTERLEAVED;
                const double *su = v.fused_solution + (il ? 3LL * n : 2LL * n);
                val[0] = su[0];
                val[1] = su[1];
                val[2] = il ? su[2] : v.fused_solution[2LL * v.n_owned + n];
                v.u[0][n] = val[0];
                v.u[1][n] = val[1];
                v.phi[n] = val[2];
              }
            else
              {
                val[0] = v.u[0][n];
                val[1] = v.u[1][n];
                val[2] = v.phi[n];
              }
            const double po = v.phi_old[n], poo = v.phi_oldold[n];
            if constexpr (LIN)
              val[3] = S.use_old ? po : poo + S.tfac * (po - poo);
            else
              {
                val[3] = po;
                val[4] = poo;
              }
          }
      };
      // Accumulators in the moment basis of the bilinear test functions (see k_cart_residual3): index 0 stands for
      // phi_0 + phi_1 = 1 (gradient 0), index 1 for phi_1; M[x][y][component].  The part of y-vertex 1 is carried to the
      // next cell row, where it is the part of y-vertex 0.
      double lo[NF], up[NF], M[2][2][3];
#pragma unroll
      for (int a = 0; a < 4; ++a)
        M[a & 1][a >> 1][0] = M[a & 1][a >> 1][1] = M[a & 1][a >> 1][2] = 0.0;
      const double c_g = 1.0 - S.kappa, c_pd = S.aB1 * S.p, c_r2 = -2.0 * S.aB1 * S.p, c_r3 = S.Gc / S.eps, c_ge = S.Gc * S.eps;
      load_row(jA - 1, lo);
#pragma unroll 1
      for (int cj = jA - 1; cj < jB; ++cj)
        {
          load_row(cj + 1, up);
          // right-hand vertices: the nodes of lane + 1
          double lo1[NF], up1[NF];
#pragma unroll
          for (int f = 0; f < NF; ++f)
            {
              lo1[f] = __shfl_down(lo[f], 1);
              up1[f] = __shfl_down(up[f], 1);
            }
          if (col_ok && cj >= 0 && cj < cv.NY - 1)
            {
              double lam = S.lam, mu = S.mu;
              if (cv.cell_lam) // heterogeneous material, cracks.cc:2207-2216
                {
                  lam = cv.cell_lam[i + (long long)(cv.NX - 1) * cj];
                  mu = cv.cell_mu[i + (long long)(cv.NX - 1) * cj];
                }
              double Dy0[3], dDy[3]; // d/dy at x-vertex 0 (constant along y) and its x-difference
#pragma unroll
              for (int f = 0; f < 3; ++f)
                {
                  Dy0[f] = (up[f] - lo[f]) * ihy;
                  dDy[f] = (up1[f] - lo1[f]) * ihy - Dy0[f];
                }
              const double mu2 = 2 * mu;
              if constexpr (LIN)
                residual_cell_poly2(lo, lo1, up, up1, S, lam, mu, M);
#pragma unroll 1
              for (int qy = 0; qy < (LIN ? 0 : 3); ++qy)
                {
                  const double eta = c_t1.n[1][qy];
                  const double wy = vol * c_t1.w[qy];
                  double L0[NF], dL[NF];
#pragma unroll
                  for (int f = 0; f < NF; ++f)
                    {
                      L0[f] = fma(eta, up[f] - lo[f], lo[f]);
                      dL[f] = fma(eta, up1[f] - lo1[f], lo1[f]) - L0[f];
                    }
                  const double Dx0 = dL[0] * ihx, Dx1 = dL[1] * ihx, Dxp = dL[2] * ihx;
                  // x-sums of the fluxes against 1 (s) and phi_1 (1); the x-gradient factor needs the plain sum only
                  double A0[3], Bs[3], B1[3], rs, r1;
                  auto xq = [&](const int qx, auto first) __attribute__((always_inline)) {
                    const double xi = c_t1.n[1][qx], wq = c_t1.w[qx], wx = c_t1.wn1[qx];
                    const double g00 = Dx0, g10 = Dx1;
                    const double g01 = fma(xi, dDy[0], Dy0[0]), g11 = fma(xi, dDy[1], Dy0[1]);
                    const double gp0 = Dxp, gp1 = fma(xi, dDy[2], Dy0[2]);
                    double pf = fma(xi, dL[2], L0[2]);
                    double pfo = fma(xi, dL[3], L0[3]); // LIN: the combined field
                    double pen = 0.0, pfx;
                    if constexpr (LIN)
                      {
                        pfx = pfo;
                        if (!S.use_old)
                          pfx = fmin(fmax(pfx, 0.0), 1.0);
                      }
                    else
                      {
                        double pfoo = fma(xi, dL[NF - 1], L0[NF - 1]);
                        if (S.monolithic)
                          {
                            pf = fmax(0.0, pf);
                            pfo = fmax(0.0, pfo);
                            pfoo = fmax(0.0, pfoo);
                          }
                        pen = fmax(0.0, pf - pfo);
                        pfx = pfoo + S.tfac * (pfo - pfoo);
                        if (pfx <= 0.0)
                          pfx = 0.0;
                        if (pfx >= 1.0)
                          pfx = 1.0;
                        if (S.use_old)
                          pfx = pfo;
                      }
                    const double pf2 = pfx * pfx;
                    const double g = fma(c_g, pf2, S.kappa);
                    const double t01 = g01 + g10, trE = g00 + g11;
                    const double lt = lam * trE;
                    const double s00 = fma(mu2, g00, lt), s11 = fma(mu2, g11, lt), s01 = mu * t01;
                    const double spE = fma(s00, g00, fma(s11, g11, s01 * t01));
                    const double pd = c_pd * pf2;
                    const double z00 = fma(g, s00, -pd), z11 = fma(g, s11, -pd), z01 = g * s01;
                    double rq = fma(pf, fma(c_r2, trE, fma(c_g, spE, c_r3)), -c_r3);
                    if constexpr (!LIN)
                      rq = fma(S.gamma_fac, pen, rq);
                    const double F[3][2] = {{z00, z01}, {z01, z11}, {c_ge * gp0, c_ge * gp1}};
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                      if constexpr (decltype(first)::value)
                        {
                          A0[c] = wq * F[c][0];
                          Bs[c] = wq * F[c][1];
                          B1[c] = wx * F[c][1];
                        }
                      else
                        {
                          A0[c] = fma(wq, F[c][0], A0[c]);
                          Bs[c] = fma(wq, F[c][1], Bs[c]);
                          B1[c] = fma(wx, F[c][1], B1[c]);
                        }
                    if constexpr (decltype(first)::value)
                      {
                        rs = wq * rq;
                        r1 = wx * rq;
                      }
                    else
                      {
                        rs = fma(wq, rq, rs);
                        r1 = fma(wx, rq, r1);
                      }
                  };
                  xq(0, std::true_type{});
                  xq(1, std::false_type{});
                  xq(2, std::false_type{});
                  // y factors: M[i][j] += P Y_j + B dY_j with (Y, dY) = (1, 0) | (eta, 1/h_y)
                  const double kx = ihx * wy, ky = ihy * wy;
#pragma unroll
                  for (int c = 0; c < 3; ++c)
                    {
                      double P = kx * A0[c];
                      if (c == 2)
                        P = fma(wy, r1, P);
                      M[1][0][c] += P;
                      M[1][1][c] = fma(eta, P, fma(ky, B1[c], M[1][1][c]));
                      if (c == 2)
                        {
                          const double Ps = wy * rs;
                          M[0][0][c] += Ps;
                          M[0][1][c] = fma(eta, Ps, fma(ky, Bs[c], M[0][1][c]));
                        }
                      else
                        M[0][1][c] = fma(ky, Bs[c], M[0][1][c]);
                    }
                }
            }
          // node row cj: the y-vertex-0 parts; own a_x = 0 part + the a_x = 1 part of the cell column on the left (lane - 1)
          double tot[3];
#pragma unroll
          for (int c = 0; c < 3; ++c)
            {
              const double l1 = M[1][0][c] - M[1][1][c];           // x-vertex 1
              const double l0 = (M[0][0][c] - M[0][1][c]) - l1;    // x-vertex 0
              tot[c] = -(l0 + __shfl_up(l1, 1));
            }
          if (cj >= jA && owner)
            {
              const int row = cart_local_id3(cv, i, cj, 0);
              const unsigned fl = v.node_flags[row];
#pragma unroll
              for (int c = 0; c < 3; ++c)
                {
                  const bool con = (fl >> c) & 1u;
                  const long long di = dof_index_c<2>(v, row, c);
                  res_pde[di] = con ? 0.0 : tot[c]; // constrained scatter = masked store (cracks.cc:2440-2456)
                  if (write_total)
                    res_tot[di] = (con && S.total_via_update) ? 0.0 : tot[c];
                }
            }
          // the y-vertex-1 part becomes the y-vertex-0 part of the next cell row (y index 0 = lower + upper)
#pragma unroll
          for (int c = 0; c < 3; ++c)
            {
              M[0][0][c] = M[0][1][c];
              M[1][0][c] = M[1][1][c];
              M[0][1][c] = M[1][1][c] = 0.0;
            }
#pragma unroll
          for (int f = 0; f < NF; ++f)
            lo[f] = up[f];
        }
    }

    // =====================================================================================
    // 3-D residual, z-marching cell columns (cracks.cc:2393-2432).
    //
    // A workgroup owns a 15 x 15 column of nodes over a chunk of z-planes; thread <-> one of the 16 x 16
    // cell columns touching those nodes.  Marching up in z, a thread evaluates the q-point state of its cell
    // ONCE, integrates it against all 8 test vertices with the x-sum factored out (22 accumulate-ops per
    // q-point instead of 160), keeps the part that belongs to the upper node plane in registers for the
    // next step, and hands the lower part to the 4 nodes of the plane through LDS, where each owned node
    // adds its 4 cell columns in a fixed order and writes its rows exactly once.  Nodal values live in a
    // two-plane LDS ring: every plane is read from HBM/L2 once per tile.
    // =====================================================================================
    constexpr int RTX = 16, RTY = 16;           // cell columns per workgroup = threads
    constexpr int RNX = RTX - 1, RNY = RTY - 1; // owned nodes per tile plane
    constexpr int RHX = RTX + 1, RHY = RTY + 1; // nodal halo per plane
    constexpr int RPL = RHX * RHY + 3;          // padded plane stride (292)

    // LIN: pf_extra is a linear function of the two old phase fields up to its final clamp (no per-q-point clamping of
    // the old fields: not monolithic; no penalisation term that needs phi_old alone): the combination is formed once
    // per node when a plane is loaded and interpolated as ONE field (cracks.cc:2262-2277 are linear until the clamp).
    // value of lane + 1 within a row of 16 lanes (0 for the last lane of the row)
    __device__ __forceinline__ double row_shl1(double x)
    {
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x101, 0xf, 0xf, true);
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x101, 0xf, 0xf, true);
      return __hiloint2double(hi, lo);
    }

    // ---- round 4: one cell of k_cart_residual3 (LIN: staggered scheme without penalisation) WITHOUT a loop over its
    // q-points for anything but the clamped factor.  pfm_poly.h: u and phi are trilinear on the cell, every strain /
    // stress component is a multilinear polynomial with known zeros, and the accumulators M[i][j][k][c] are moments against
    // the monomials psi = t^i s^j r^k (the "moment basis" of the test functions: index 1 = phi_1 = t with gradient 1/h,
    // index 0 = phi_0 + phi_1 = 1 with gradient 0).  Therefore, with H_f[m] = sum_q w f(q) x_q^m the discrete moments of a
    // factor f against the monomials of powers 0..2:
    //   sum_q w g sigma_ck psi        = sum_idx sigma_ck[idx] H_g[idx + psi]         g = (1-kappa) pfx^2 + kappa, pfx CLAMPED at q
    //   sum_q w pf Theta psi          = sum_idx pf[idx] H_Theta[idx + psi]           Theta = (1-kappa) sigma:E - 2(alpha-1)p div u + G_c/eps
    // H_g needs the 27 q-point values of pfx^2 (interpolate, clamp, square: the only thing evaluated at q-points) and a
    // three-stage contraction (243 FMAs); H_Theta comes from the 27 coefficients of Theta and the integrals of t^n (the 3-point
    // rule is exact up to t^5).  ~1500 instead of ~4000 instructions per cell; the sums are the reference's, regrouped.
    // PL: plane stride of a field in LDS.  RAW45: fields 4 and 5 hold the two old phase fields as they lie in memory (the
    // planes of k_cart_residual3d arrive without passing through registers): the combination is formed here, per vertex
    // LY: where the six fields of a nodal plane lie in LDS, relative to the cell's (0,0) vertex -- LY::off(f) doubles from
    // Ulo_u / Uhi_u (displacements) or Ulo_s / Uhi_s (phase fields), LY::xs(f) / LY::ys(f) to the next node along x / y
    template <int PL>
    struct PlaneSoA // k_cart_residual3, k_cart_residual3d: six planes of RHX x RHY doubles, PL apart
    {
      static constexpr int off(int f) { return f * PL; }
      static constexpr int xs(int) { return 1; }
      static constexpr int ys(int) { return RHX; }
    };
    template <class LY, bool RAW45>
    __device__ __forceinline__ void residual_cell_poly(const double *__restrict__ Ulo_u, const double *__restrict__ Uhi_u,
                                                       const double *__restrict__ Ulo_s, const double *__restrict__ Uhi_s, const Scal &S,
                                                       double lam, double mu, double (&M)[2][2][2][4])
    {
      const double c_g = S.c_g, c_r2 = S.c_r2, c_r3 = S.c_r3, vol = S.vol;
      const double(&ih)[3] = S.ih;
      const double(&vih)[3] = S.vih;
      auto load8 = [&](int f, double (&a)[8]) __attribute__((always_inline)) {
        const double *lo = (f < 3 ? Ulo_u : Ulo_s) + LY::off(f), *hi = (f < 3 ? Uhi_u : Uhi_s) + LY::off(f);
        const int xs = LY::xs(f), ys = LY::ys(f);
        a[0] = lo[0], a[1] = lo[xs], a[2] = lo[ys], a[3] = lo[ys + xs];
        a[4] = hi[0], a[5] = hi[xs], a[6] = hi[ys], a[7] = hi[ys + xs];
      };
      auto Madd = [&](auto Psi, auto Cc, double x) __attribute__((always_inline)) {
        constexpr int psi = decltype(Psi)::value, c = decltype(Cc)::value;
        M[psi & 1][(psi >> 1) & 1][psi >> 2][c] += x;
      };
      // ---------------- discrete moments of pfx^2 (HP) and of g (Hg) against t^p s^q r^r, p, q, r = 0..2
      // The pressure term -(alpha_B-1) p pfx^2 delta_ck needs the moments of pfx^2 against the multilinear monomials: they are
      // those of g, H_g[m] = c_g H_P[m] + kappa I_m, so the term is (c_pd / c_g)(kappa I_m - H_g[m]) -- folded into the
      // constant coefficient of sigma_cc below.  (Round 4 kept the eight H_P next to H_g: 16 of the 22 registers the
      // kernel spilled, each reload a scratch load and a vmcnt(0) in the middle of the evaluation.)
      double Hg[27];
      {
        double W[8];
        load8(4, W); // LIN: the combined old field (load_plane)
        if constexpr (RAW45)
          {
            double W5[8];
            load8(5, W5);
#pragma unroll
            for (int i = 0; i < 8; ++i)
              W[i] = S.use_old ? W[i] : W5[i] + S.tfac * (W[i] - W5[i]); // as store_node of k_cart_residual3 forms it
          }
        monomials(W);
        double Pq[27];
        poly_for<3>([&](auto Qx) __attribute__((always_inline)) {
          constexpr int qx = decltype(Qx)::value;
          double X[4];
#pragma unroll
          for (int bc = 0; bc < 4; ++bc)
            X[bc] = fma(GqT<qx>::v, W[2 * bc + 1], W[2 * bc]);
          poly_for<3>([&](auto Qy) __attribute__((always_inline)) {
            constexpr int qy = decltype(Qy)::value;
            const double y0 = fma(GqT<qy>::v, X[1], X[0]), y1 = fma(GqT<qy>::v, X[3], X[2]);
            poly_for<3>([&](auto Qz) __attribute__((always_inline)) {
              constexpr int qz = decltype(Qz)::value;
              double pfx = fma(GqT<qz>::v, y1, y0);
              if (!S.use_old)
                pfx = fmin(fmax(pfx, 0.0), 1.0); // cracks.cc:2270-2277
              Pq[qx + 3 * qy + 9 * qz] = pfx * pfx;
            });
          });
        });
        __builtin_amdgcn_sched_barrier(0); // stage by stage: interleaved, the stages do not fit the registers
        double A1[27], A2[27];
        poly_for<3>([&](auto Pp) __attribute__((always_inline)) {
          constexpr int pp = decltype(Pp)::value;
#pragma unroll
          for (int yz = 0; yz < 9; ++yz)
            A1[pp + 3 * yz] = GqWt<0, pp>::v * Pq[3 * yz] + GqWt<1, pp>::v * Pq[3 * yz + 1] + GqWt<2, pp>::v * Pq[3 * yz + 2];
        });
        __builtin_amdgcn_sched_barrier(0); // stage by stage: interleaved, the stages do not fit the registers
        poly_for<3>([&](auto Qp) __attribute__((always_inline)) {
          constexpr int qp = decltype(Qp)::value;
#pragma unroll
          for (int pp = 0; pp < 3; ++pp)
#pragma unroll
            for (int qz = 0; qz < 3; ++qz)
              A2[pp + 3 * qp + 9 * qz] =
                GqWt<0, qp>::v * A1[pp + 9 * qz] + GqWt<1, qp>::v * A1[pp + 3 + 9 * qz] + GqWt<2, qp>::v * A1[pp + 6 + 9 * qz];
        });
        __builtin_amdgcn_sched_barrier(0); // stage by stage: interleaved, the stages do not fit the registers
        poly_for<27>([&](auto Mm) __attribute__((always_inline)) {
          constexpr int m = decltype(Mm)::value, pp = m % 3, qp = (m / 3) % 3, rp = m / 9;
          const double hp = GqWt<0, rp>::v * A2[pp + 3 * qp] + GqWt<1, rp>::v * A2[pp + 3 * qp + 9] + GqWt<2, rp>::v * A2[pp + 3 * qp + 18];
          constexpr double Im = GqMom<pp>::v * GqMom<qp>::v * GqMom<rp>::v;
          Hg[m] = fma(c_g, hp, S.kappa * Im);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- displacement rows
      double F[3][8];
#pragma unroll
      for (int f = 0; f < 3; ++f)
        {
          load8(f, F[f]);
          monomials(F[f]);
        }
      auto G = [&](auto Cc, auto Kc, auto Ic) __attribute__((always_inline)) -> double { // d_k u_c, coefficient idx
        constexpr int c = decltype(Cc)::value, k = decltype(Kc)::value, idx = decltype(Ic)::value;
        static_assert(!(idx & (1 << k)), "no such monomial in this derivative");
        return ih[k] * F[c][idx | (1 << k)];
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      using I3 = std::integral_constant<int, 3>;
      using I4 = std::integral_constant<int, 4>;
      using I5 = std::integral_constant<int, 5>;
      using I6 = std::integral_constant<int, 6>;
      double T[8]; // div u
      T[0] = (G(I0{}, I0{}, I0{}) + G(I1{}, I1{}, I0{})) + G(I2{}, I2{}, I0{});
      T[1] = G(I1{}, I1{}, I1{}) + G(I2{}, I2{}, I1{});
      T[2] = G(I0{}, I0{}, I2{}) + G(I2{}, I2{}, I2{});
      T[3] = G(I2{}, I2{}, I3{});
      T[4] = G(I0{}, I0{}, I4{}) + G(I1{}, I1{}, I4{});
      T[5] = G(I1{}, I1{}, I5{});
      T[6] = G(I0{}, I0{}, I6{});
      const double mu2 = 2.0 * mu;
      poly_for<3>([&](auto Cc) __attribute__((always_inline)) {
        poly_for<3>([&](auto Kc) __attribute__((always_inline)) {
          constexpr int c = decltype(Cc)::value, k = decltype(Kc)::value;
          constexpr int mc = c == 0 ? NOX : (c == 1 ? NOY : NOZ), mk = k == 0 ? NOX : (k == 1 ? NOY : NOZ);
          constexpr int mask = c == k ? 0x7f : (mc | mk);
          // sigma_ck = lambda div u delta_ck + mu (d_k u_c + d_c u_k)
          double sg[8];
          poly_for<8>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int idx = decltype(Ic)::value;
            if constexpr ((mask >> idx) & 1)
              {
                if constexpr (c == k)
                  {
                    if constexpr ((mc >> idx) & 1)
                      sg[idx] = fma(mu2, G(Cc, Cc, Ic), lam * T[idx]);
                    else
                      sg[idx] = lam * T[idx];
                  }
                else
                  {
                    constexpr bool hk = (mk >> idx) & 1, hc = (mc >> idx) & 1; // d_k u_c lives on the monomials without x_k
                    if constexpr (hk && hc)
                      sg[idx] = mu * (G(Cc, Kc, Ic) + G(Kc, Cc, Ic));
                    else if constexpr (hk)
                      sg[idx] = mu * G(Cc, Kc, Ic);
                    else
                      sg[idx] = mu * G(Kc, Cc, Ic);
                  }
              }
          });
          // flux moments against the 4 monomials on the two other axes; the gradient factor of psi along k is 1/h_k
          constexpr int a1 = k == 0 ? 1 : 0, a2 = k == 2 ? 1 : 2;
          poly_for<4>([&](auto Jl) __attribute__((always_inline)) {
            constexpr int midx = ((decltype(Jl)::value & 1) << a1) | ((decltype(Jl)::value >> 1) << a2);
            double val = 0.0;
            bool any = false;
            poly_for<8>([&](auto Ic) __attribute__((always_inline)) {
              constexpr int idx = decltype(Ic)::value;
              if constexpr ((mask >> idx) & 1)
                {
                  val = any ? fma(sg[idx], Hg[pow_of(idx, midx)], val) : sg[idx] * Hg[pow_of(idx, midx)];
                  any = true;
                }
            });
            if constexpr (c == k)
              {
                // - (alpha_B - 1) p pfx^2 delta_ck = (c_pd / c_g) (kappa I - H_g) against the monomial midx
                constexpr int m0 = pow_of(0, midx);
                constexpr double Im0 = GqMom<m0 % 3>::v * GqMom<(m0 / 3) % 3>::v * GqMom<m0 / 9>::v;
                val = fma(-S.c_pdg, Hg[m0], fma(S.c_pdk, Im0, val));
              }
            Madd(std::integral_constant<int, (1 << k) | midx>{}, Cc, vih[k] * val);
          });
        });
      });
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- phase-field row: value term pf Theta - G_c/eps and the flux G_c eps grad(pf)
      double Th[27];
#pragma unroll
      for (int m = 0; m < 27; ++m)
        Th[m] = 0.0;
      add_square<0x7f>(T, lam, Th);
      poly_for<3>([&](auto Cc) __attribute__((always_inline)) {
        constexpr int c = decltype(Cc)::value;
        constexpr int mc = c == 0 ? NOX : (c == 1 ? NOY : NOZ);
        double P[8];
        poly_for<8>([&](auto Ic) __attribute__((always_inline)) {
          if constexpr ((mc >> decltype(Ic)::value) & 1)
            P[decltype(Ic)::value] = G(Cc, Cc, Ic);
        });
        add_square<mc>(P, mu2, Th);
      });
      poly_for<3>([&](auto Pc) __attribute__((always_inline)) {
        constexpr int pr = decltype(Pc)::value;
        constexpr int c = pr == 2 ? 1 : 0, d = pr == 0 ? 1 : 2; // (0,1), (0,2), (1,2)
        constexpr int mc = c == 0 ? NOX : NOY, md = d == 1 ? NOY : NOZ;
        using IC = std::integral_constant<int, c>;
        using ID = std::integral_constant<int, d>;
        double Sp[8];
        poly_for<8>([&](auto Ic) __attribute__((always_inline)) {
          constexpr int idx = decltype(Ic)::value;
          constexpr bool hd = (md >> idx) & 1, hc = (mc >> idx) & 1;
          if constexpr (hd && hc)
            Sp[idx] = G(IC{}, ID{}, Ic) + G(ID{}, IC{}, Ic);
          else if constexpr (hd)
            Sp[idx] = G(IC{}, ID{}, Ic);
          else if constexpr (hc)
            Sp[idx] = G(ID{}, IC{}, Ic);
        });
        add_square<(mc | md)>(Sp, mu, Th); // mu t_cd^2
      });
#pragma unroll
      for (int m = 0; m < 27; ++m)
        Th[m] *= c_g;
      Th[0] += c_r3;
      poly_for<7>([&](auto Ic) __attribute__((always_inline)) {
        constexpr int idx = decltype(Ic)::value;
        Th[pow_of(idx, 0)] = fma(c_r2, T[idx], Th[pow_of(idx, 0)]);
      });
      __builtin_amdgcn_sched_barrier(0);
      // H_Theta[p + 3 q + 9 r] = sum_abc Theta_abc I(a + p) I(b + q) I(c + r), I(n) = integral of t^n
      double B1[27], B2[27], HT[27];
      poly_for<3>([&](auto Pp) __attribute__((always_inline)) {
        constexpr int pp = decltype(Pp)::value;
#pragma unroll
        for (int bc = 0; bc < 9; ++bc)
          B1[pp + 3 * bc] = GqMom<pp>::v * Th[3 * bc] + GqMom<pp + 1>::v * Th[3 * bc + 1] + GqMom<pp + 2>::v * Th[3 * bc + 2];
      });
      poly_for<3>([&](auto Qp) __attribute__((always_inline)) {
        constexpr int qp = decltype(Qp)::value;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc)
            B2[pp + 3 * qp + 9 * cc] =
              GqMom<qp>::v * B1[pp + 9 * cc] + GqMom<qp + 1>::v * B1[pp + 3 + 9 * cc] + GqMom<qp + 2>::v * B1[pp + 6 + 9 * cc];
      });
      poly_for<3>([&](auto Rp) __attribute__((always_inline)) {
        constexpr int rp = decltype(Rp)::value;
#pragma unroll
        for (int pq = 0; pq < 9; ++pq)
          HT[pq + 9 * rp] = GqMom<rp>::v * B2[pq] + GqMom<rp + 1>::v * B2[pq + 9] + GqMom<rp + 2>::v * B2[pq + 18];
      });
      __builtin_amdgcn_sched_barrier(0);
      double Pf[8];
      load8(3, Pf);
      monomials(Pf);
      poly_for<8>([&](auto Psi) __attribute__((always_inline)) {
        constexpr int psi = decltype(Psi)::value;
        constexpr double Ipsi = GqMom<(psi & 1)>::v * GqMom<((psi >> 1) & 1)>::v * GqMom<(psi >> 2)>::v;
        double val = -c_r3 * Ipsi;
        poly_for<8>([&](auto Ic) __attribute__((always_inline)) {
          constexpr int idx = decltype(Ic)::value;
          val = fma(Pf[idx], HT[pow_of(idx, psi)], val);
        });
        Madd(Psi, I3{}, vol * val);
      });
      poly_for<3>([&](auto Kc) __attribute__((always_inline)) {
        constexpr int k = decltype(Kc)::value;
        constexpr int a1 = k == 0 ? 1 : 0, a2 = k == 2 ? 1 : 2;
        poly_for<4>([&](auto Jl) __attribute__((always_inline)) {
          constexpr int midx = ((decltype(Jl)::value & 1) << a1) | ((decltype(Jl)::value >> 1) << a2);
          double val = 0.0;
          bool any = false;
          poly_for<8>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int idx = decltype(Ic)::value;
            if constexpr (!(idx & (1 << k)))
              {
                constexpr int pw = pow_of(idx, midx);
                constexpr double Ic3 = GqMom<pw % 3>::v * GqMom<(pw / 3) % 3>::v * GqMom<pw / 9>::v;
                val = any ? fma(Pf[idx | (1 << k)], Ic3, val) : Pf[idx | (1 << k)] * Ic3;
                any = true;
              }
          });
          Madd(std::integral_constant<int, (1 << k) | midx>{}, I3{}, S.geih[k] * val);
        });
      });
    }

    template <bool LIN>
    __global__ __launch_bounds__(RTX *RTY, 2) void k_cart_residual3(DevView v, CartView cv, Scal S,
                                                                 double *__restrict__ res_pde,
                                                                 double *__restrict__ res_tot, int write_total,
                                                                 int zc /* node planes per chunk */)
    {
      __shared__ double s_U[2][6][RPL]; // [ring][u_x u_y u_z phi phi_old phi_oldold][halo node]
      __shared__ double s_P[8][RTX * RTY]; // [ay * 4 + component]: the a_x = 0 / 1 parts are merged in registers (DPP)

      const int t = threadIdx.x, cx = t % RTX, cy = t / RTX;
      const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1;
      const int ntx = (OWX + RNX - 1) / RNX, nty = (OWY + RNY - 1) / RNY;
      int bid = (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)); // XCD-aware launch, pfm_internal.h
      const bool listed = cv.tile_sel == 2 && cv.bnd_res3 != nullptr && cv.zc_res3 == zc; // compact launch over the boundary tiles
      if (listed)
        {
          if (bid >= cv.n_bnd_res3)
            return;
          bid = cv.bnd_res3[bid];
        }
      if (bid >= ntx * nty * ((cv.o1[2] - cv.o0[2] + zc) / zc))
        return;
      const int tix = bid % ntx, tiy = (bid / ntx) % nty, chunk = bid / (ntx * nty);
      const int i0 = cv.o0[0] + tix * RNX, j0 = cv.o0[1] + tiy * RNY;
      const int kA = cv.o0[2] + chunk * zc;
      const int kB = min(kA + zc, cv.o1[2] + 1); // node planes [kA, kB)
      if (!listed && cart_tile_skipped(cv, cart_range_has_ghost(cv, 0, i0 - 1, i0 + RNX) || cart_range_has_ghost(cv, 1, j0 - 1, j0 + RNY) ||
                                               cart_range_has_ghost(cv, 2, kA - 1, kB)))
        return; // overlapped assembly: the other launch owns this tile column chunk
      const int ci = i0 - 1 + cx, cj = j0 - 1 + cy;
      const bool col_ok = ci >= 0 && ci < cv.NX - 1 && cj >= 0 && cj < cv.NY - 1;
      const bool node_ok = cx < RNX && cy < RNY && (i0 + cx) <= cv.o1[0] && (j0 + cy) <= cv.o1[1];

      const double ihx = 1.0 / cv.h[0], ihy = 1.0 / cv.h[1], ihz = 1.0 / cv.h[2];
      const double vol = cv.h[0] * cv.h[1] * cv.h[2];

      // nodal plane kz -> ring slot buf.  The 17 x 17 halo nodes are 289 > 256 threads: threads 0..32 fetch a second node.  Both
      // fetches are in flight before the first LDS store (round 5; as a loop the second fetch started after the first one's
      // values had arrived: two global round trips in wave 0 -- behind which the whole workgroup waits at the barrier of
      // every step)
      auto fetch_node = [&](int kz, int idx, double (&val)[6], int &n_pub) __attribute__((always_inline)) {
        const int hx = idx % RHX, hy = idx / RHX;
        const int gi = i0 - 1 + hx, gj = j0 - 1 + hy;
#pragma unroll
        for (int f = 0; f < 6; ++f)
          val[f] = 0.0;
        n_pub = -1;
        const int n = (gi >= 0 && gi < cv.NX && gj >= 0 && gj < cv.NY && kz >= 0 && kz < cv.NZ) ? cart_local_id3(cv, gi, gj, kz) : -1;
        if (n >= 0) // (-1 inside the lattice: a level lattice of the 3-D overlay has no node there)
          {
            if (v.fused_solution) // kernel argument: uniform branch.  Single rank: every node is an owned node
              {
                const bool il = v.layout == PFM_LAYOUT_INTERLEAVED;
                const double *su = v.fused_solution + (il ? 4LL * n : 3LL * n);
                val[0] = su[0];
                val[1] = su[1];
                val[2] = su[2];
                val[3] = il ? su[3] : v.fused_solution[3LL * v.n_owned + n];
                // the node state is what pfm_state_set_solution would have left: every node is written once, by the
                // tile and z-chunk that own it (publish_node, behind the fetches)
                if (hx >= 1 && hx <= RNX && hy >= 1 && hy <= RNY && kz >= kA && kz < kB)
                  n_pub = n;
              }
            else
              {
                val[0] = v.u[0][n];
                val[1] = v.u[1][n];
                val[2] = v.u[2][n];
                val[3] = v.phi[n];
              }
            val[4] = v.phi_old[n];
            val[5] = v.phi_oldold[n];
          }
      };
      auto store_node = [&](int buf, int idx, double (&val)[6], int n_pub) __attribute__((always_inline)) {
        if (n_pub >= 0)
          {
            v.u[0][n_pub] = val[0];
            v.u[1][n_pub] = val[1];
            v.u[2][n_pub] = val[2];
            v.phi[n_pub] = val[3];
          }
        if constexpr (LIN)
          val[4] = S.use_old ? val[4] : val[5] + S.tfac * (val[4] - val[5]);
#pragma unroll
        for (int f = 0; f < (LIN ? 5 : 6); ++f)
          s_U[buf][f][idx] = val[f];
      };
      auto load_plane = [&](int kz, int buf) __attribute__((always_inline)) {
        static_assert(RHX * RHY <= 2 * RTX * RTY, "two nodes per thread at most");
        double va[6], vb[6];
        int na, nb = -1;
        const int ib = t + RTX * RTY;
        fetch_node(kz, t, va, na);
        if (ib < RHX * RHY)
          fetch_node(kz, ib, vb, nb);
        store_node(buf, t, va, na);
        if (ib < RHX * RHY)
          store_node(buf, ib, vb, nb);
      };

      // Accumulators in the moment basis of the trilinear test functions: along every direction index 0 stands for
      // phi_0 + phi_1 = 1 (gradient 0) and index 1 for phi_1 (gradient 1/h).  The part of vertex 0 is a difference that is
      // taken once per layer, and the sum index of a gradient factor contributes nothing: 61 accumulate-ops per (q_y, q_z)
      // instead of 137 in the vertex basis.  M[x][y][z][component].
      double M[2][2][2][4];
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          M[a & 1][(a >> 1) & 1][a >> 2][c] = 0.0;

      // constants of the q-point law with the parameters folded (cracks.cc:2393-2432)
      const double c_g = 1.0 - S.kappa, c_pd = S.aB1 * S.p, c_r2 = -2.0 * S.aB1 * S.p, c_r3 = S.Gc / S.eps, c_ge = S.Gc * S.eps;

      load_plane(kA - 1, 0);
      const int hb = cy * RHX + cx;
#pragma unroll 1
      for (int ck = kA - 1; ck < kB; ++ck)
        {
          const int lo = (ck - (kA - 1)) & 1, hi = lo ^ 1;
          load_plane(ck + 1, hi);
          __syncthreads();
          if (col_ok && ck >= 0 && ck < cv.NZ - 1)
            {
              const double *Ulo = &s_U[lo][0][hb], *Uhi = &s_U[hi][0][hb];
              double lam = S.lam, mu = S.mu;
              if (cv.cell_lam) // heterogeneous material, cracks.cc:2207-2216
                {
                  const long long cidx = ci + (long long)(cv.NX - 1) * (cj + (long long)(cv.NY - 1) * ck);
                  lam = cv.cell_lam[cidx];
                  mu = cv.cell_mu[cidx];
                }
              const double mu2 = 2 * mu;
              if constexpr (LIN)
                residual_cell_poly<PlaneSoA<RPL>, false>(Ulo, Uhi, Ulo, Uhi, S, lam, mu, M);
              double Dy0[4], dDy[4]; // d/dy at x-vertex 0 and its x-difference: depend on the z-level only
#pragma unroll 1
              for (int p = 0; p < (LIN ? 0 : 9); ++p)
                {
                  const int qy = p % 3, qz = p / 3;
                  const double eta = c_t1.n[1][qy], zeta = c_t1.n[1][qz];
                  const double wyz = vol * c_t1.w[qy] * c_t1.w[qz];
                  if (qy == 0)
                    {
#pragma unroll
                      for (int f = 0; f < 4; ++f)
                        {
                          const double s0 = Ulo[f * RPL + RHX] - Ulo[f * RPL], t0 = Uhi[f * RPL + RHX] - Uhi[f * RPL];
                          const double s1 = Ulo[f * RPL + RHX + 1] - Ulo[f * RPL + 1], t1 = Uhi[f * RPL + RHX + 1] - Uhi[f * RPL + 1];
                          const double g0 = fma(zeta, t0 - s0, s0), g1 = fma(zeta, t1 - s1, s1);
                          Dy0[f] = ihy * g0;
                          dDy[f] = ihy * (g1 - g0);
                        }
                    }
                  // values at x-vertex 0 and x-differences of: the field (L0, dL), its z-derivative (Dz0, dDz); Dx = dL / h_x
                  double L0[6], dL[6], Dz0[4], dDz[4], Dx[4];
                  L0[5] = dL[5] = 0.0;
#pragma unroll
                  for (int f = 0; f < (LIN ? 5 : 6); ++f)
                    {
                      const double a00 = Ulo[f * RPL], a10 = Ulo[f * RPL + 1], a01 = Ulo[f * RPL + RHX],
                                   a11 = Ulo[f * RPL + RHX + 1];
                      const double b00 = Uhi[f * RPL], b10 = Uhi[f * RPL + 1], b01 = Uhi[f * RPL + RHX],
                                   b11 = Uhi[f * RPL + RHX + 1];
                      const double lo0 = fma(eta, a01 - a00, a00), lo1 = fma(eta, a11 - a10, a10);
                      const double hi0 = fma(eta, b01 - b00, b00), hi1 = fma(eta, b11 - b10, b10);
                      const double d0 = hi0 - lo0, d1 = hi1 - lo1;
                      L0[f] = fma(zeta, d0, lo0);
                      dL[f] = fma(zeta, d1, lo1) - L0[f];
                      if (f < 4)
                        {
                          Dz0[f] = ihz * d0;
                          dDz[f] = ihz * (d1 - d0);
                          Dx[f] = ihx * dL[f];
                        }
                      if (f & 1) __builtin_amdgcn_sched_barrier(0); // two fields at a time: 16 nodal values in flight
                    }
                  // x-stage: sums over q_x of the fluxes F_c = (Z_c0, Z_c1, Z_c2), c = 3: G_c eps grad phi, and of the value
                  // term rq, against 1 (s) and phi_1 (1); the x-gradient factor needs the plain sum only
                  double A0[4], Bs[4], B1[4], Cs[4], C1[4], rs, r1;
                  auto xq = [&](const int qx, auto first) __attribute__((always_inline)) {
                    const double xi = c_t1.n[1][qx], wq = c_t1.w[qx], wx = c_t1.wn1[qx];
                    // Newton state at q (cracks.cc:2222-2232); along x every interpolated quantity is linear
                    double gu[3][3], gpf[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                      {
                        gu[c][0] = Dx[c];
                        gu[c][1] = fma(xi, dDy[c], Dy0[c]);
                        gu[c][2] = fma(xi, dDz[c], Dz0[c]);
                      }
                    gpf[0] = Dx[3];
                    gpf[1] = fma(xi, dDy[3], Dy0[3]);
                    gpf[2] = fma(xi, dDz[3], Dz0[3]);
                    double pf = fma(xi, dL[3], L0[3]);
                    double pfo = fma(xi, dL[4], L0[4]); // LIN: the combined field
                    double pen = 0.0, pfx;
                    if constexpr (LIN)
                      {
                        pfx = pfo;
                        if (!S.use_old)
                          pfx = fmin(fmax(pfx, 0.0), 1.0);
                      }
                    else
                      {
                        double pfoo = fma(xi, dL[5], L0[5]);
                        if (S.monolithic)
                          {
                            pf = fmax(0.0, pf);
                            pfo = fmax(0.0, pfo);
                            pfoo = fmax(0.0, pfoo);
                          }
                        pen = fmax(0.0, pf - pfo);
                        pfx = pfoo + S.tfac * (pfo - pfoo);
                        if (pfx <= 0.0)
                          pfx = 0.0;
                        if (pfx >= 1.0)
                          pfx = 1.0;
                        if (S.use_old)
                          pfx = pfo;
                      }
                    const double pf2 = pfx * pfx;
                    const double g = fma(c_g, pf2, S.kappa);
                    // sigma+ = lambda tr(E) I + 2 mu E; with t_ab = g_ab + g_ba: sigma_ab = mu t_ab and
                    // sigma : E = sum_a sigma_aa g_aa + sum_{a<b} sigma_ab t_ab
                    const double t01 = gu[0][1] + gu[1][0], t02 = gu[0][2] + gu[2][0], t12 = gu[1][2] + gu[2][1];
                    const double trE = gu[0][0] + gu[1][1] + gu[2][2];
                    const double lt = lam * trE;
                    const double s00 = fma(mu2, gu[0][0], lt), s11 = fma(mu2, gu[1][1], lt), s22 = fma(mu2, gu[2][2], lt);
                    const double s01 = mu * t01, s02 = mu * t02, s12 = mu * t12;
                    const double spE = fma(s00, gu[0][0], fma(s11, gu[1][1], s22 * gu[2][2])) + fma(s01, t01, fma(s02, t02, s12 * t12));
                    // Z = g sigma+ - (alpha_B - 1) p pfx^2 I (symmetric); the weights enter with the sums
                    const double pd = c_pd * pf2;
                    const double z00 = fma(g, s00, -pd), z11 = fma(g, s11, -pd), z22 = fma(g, s22, -pd);
                    const double z01 = g * s01, z02 = g * s02, z12 = g * s12;
                    // gamma pen + (1 - kappa) sigma:E pf - G_c/eps (1 - pf) - 2 (alpha_B - 1) p pf tr(E)
                    double rq = fma(pf, fma(c_r2, trE, fma(c_g, spE, c_r3)), -c_r3);
                    if constexpr (!LIN)
                      rq = fma(S.gamma_fac, pen, rq);
                    const double F[4][3] = {{z00, z01, z02}, {z01, z11, z12}, {z02, z12, z22}, {c_ge * gpf[0], c_ge * gpf[1], c_ge * gpf[2]}};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                      if constexpr (decltype(first)::value)
                        {
                          A0[c] = wq * F[c][0];
                          Bs[c] = wq * F[c][1];
                          B1[c] = wx * F[c][1];
                          Cs[c] = wq * F[c][2];
                          C1[c] = wx * F[c][2];
                        }
                      else
                        {
                          A0[c] = fma(wq, F[c][0], A0[c]);
                          Bs[c] = fma(wq, F[c][1], Bs[c]);
                          B1[c] = fma(wx, F[c][1], B1[c]);
                          Cs[c] = fma(wq, F[c][2], Cs[c]);
                          C1[c] = fma(wx, F[c][2], C1[c]);
                        }
                    if constexpr (decltype(first)::value)
                      {
                        rs = wq * rq;
                        r1 = wx * rq;
                      }
                    else
                      {
                        rs = fma(wq, rq, rs);
                        r1 = fma(wx, rq, r1);
                      }
                  };
                  xq(0, std::true_type{});
                  xq(1, std::false_type{});
                  xq(2, std::false_type{});
                  // y and z factors: M[i][j][k] += P Y_j Z_k + B dY_j Z_k + C Y_j dZ_k with (Y, dY) = (1, 0) | (eta, 1/h_y)
                  const double kx = ihx * wyz, ky = ihy * wyz, kz = ihz * wyz, kze = kz * eta;
#pragma unroll
                  for (int c = 0; c < 4; ++c)
                    {
                      { // x index 1: value factor phi_1 (sums against wx), gradient factor 1/h_x
                        double P = kx * A0[c];
                        if (c == 3)
                          P = fma(wyz, r1, P);
                        const double u = fma(eta, P, ky * B1[c]);
                        M[1][0][0][c] += P;
                        M[1][0][1][c] = fma(kz, C1[c], fma(zeta, P, M[1][0][1][c]));
                        M[1][1][0][c] += u;
                        M[1][1][1][c] = fma(kze, C1[c], fma(zeta, u, M[1][1][1][c]));
                      }
                      if (c == 3) // x index 0 (the sum): the gradient factor vanishes, only rq has a value term
                        {
                          const double P = wyz * rs;
                          const double u = fma(eta, P, ky * Bs[c]);
                          M[0][0][0][c] += P;
                          M[0][0][1][c] = fma(kz, Cs[c], fma(zeta, P, M[0][0][1][c]));
                          M[0][1][0][c] += u;
                          M[0][1][1][c] = fma(kze, Cs[c], fma(zeta, u, M[0][1][1][c]));
                        }
                      else
                        {
                          const double u = ky * Bs[c];
                          M[0][0][1][c] = fma(kz, Cs[c], M[0][0][1][c]);
                          M[0][1][0][c] += u;
                          M[0][1][1][c] = fma(kze, Cs[c], fma(zeta, u, M[0][1][1][c]));
                        }
                    }
                }
            }
          const bool emit = ck >= kA;
          if (emit)
            {
              // lower z-vertex part of the layer, then from the moment basis to the vertices in y and x.  A node's a_x = 1
              // part is this thread's, its a_x = 0 part the next cell column's: lane + 1 of the same 16-lane row
              // (RTX = 16; the last lane of a row owns no node)
#pragma unroll
              for (int c = 0; c < 4; ++c)
                {
                  const double l01 = M[0][1][0][c] - M[0][1][1][c], l11 = M[1][1][0][c] - M[1][1][1][c];
                  const double l00 = (M[0][0][0][c] - M[0][0][1][c]) - l01, l10 = (M[1][0][0][c] - M[1][0][1][c]) - l11;
                  s_P[0 * 4 + c][t] = l10 + row_shl1(l00 - l10);
                  s_P[1 * 4 + c][t] = l11 + row_shl1(l01 - l11);
                }
            }
          __syncthreads();
          // (CartView::row_of_box: the rows of this launch; -1 = the row belongs to the general family)
          const int row = (emit && node_ok) ? (cv.row_of_box ? cv.row_of_box[(i0 + cx) + (long long)cv.NX * ((j0 + cy) + (long long)cv.NY * ck)]
                                                             : cart_local_id3(cv, i0 + cx, j0 + cy, ck))
                                            : -1;
          if (row >= 0)
            {
              const unsigned fl = v.node_flags[row];
#pragma unroll
              for (int c = 0; c < 4; ++c)
                {
                  // cell columns (i-1,j-1), (i,j-1) | (i-1,j), (i,j) of the plane, lower + upper layer already merged
                  const double r = -s_P[4 + c][t] - s_P[c][t + RTX];
                  const bool con = (fl >> c) & 1u;
                  const long long di = dof_index_c<3>(v, row, c);
                  res_pde[di] = con ? 0.0 : r; // constrained scatter = masked store (cracks.cc:2440-2456)
                  if (write_total)
                    res_tot[di] = (con && S.total_via_update) ? 0.0 : r;
                }
            }
          // the upper-vertex part becomes the lower-vertex part of the next layer (z index 0 = lower + upper)
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              {
                M[a & 1][a >> 1][0][c] = M[a & 1][a >> 1][1][c];
                M[a & 1][a >> 1][1][c] = 0.0;
              }
        }
    }


    // =====================================================================================
    // Round 6: k_cart_residual3 <LIN> with the nodal planes requested TWO steps ahead (pfm_dma.h: global -> LDS without
    // staging registers) into a ring of three planes.  Phase clock of k_cart_residual3 at 216^3 (thread 0, cycles per
    // step): load_plane 6357 (two global round trips in front of the barrier, every step), arithmetic 9251, emit + barrier
    // 847, stores 2374.  Here the requests of plane ck + 2 are issued in front of the arithmetic of layer ck and are
    // waited for at the ONE barrier of the step, behind it.  The lattice must be the whole lexicographic box of a single
    // rank (node ids by arithmetic: an id looked up in a table is a load, and the compiler's vmcnt(0) in front of its
    // use would wait for the requests as well) with byte offsets below 4 GiB (launcher).  Sums and their order are those
    // of k_cart_residual3 <true>: the results are the same bits.
    // =====================================================================================
    constexpr int RPD = 320; // plane stride: 10 wave transfers of 64 dwords per field
    struct LdsR3D
    {
      double U[3][6][RPD];       // [ring][field][halo node]; first: the transfers address LDS through M0's 16-bit offset
      double P[2][8][RTX * RTY]; // [parity of the step][a_y * 4 + component]
      unsigned nofs[RPD];        // lexicographic id of the halo node in plane 0 (clamped to the box)
    };
    static_assert(sizeof(LdsR3D) <= 81920, "two workgroups per CU");

    // HET: heterogeneous material.  A template parameter, not a uniform branch: behind a branch the compiler waits for the
    // two loads with vmcnt(0) in the middle of the arithmetic whether they were issued or not -- i.e. for the requests
    template <bool HET>
    __global__ __launch_bounds__(RTX *RTY, 2) void k_cart_residual3d(DevView v, CartView cv, Scal S, double *__restrict__ res_pde,
                                                                  double *__restrict__ res_tot, int write_total, int zc)
    {
      __shared__ LdsR3D s;
      const int t = threadIdx.x, cx = t % RTX, cy = t / RTX;
      const int wv = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
      const int OWX = cv.NX, OWY = cv.NY;
      const int ntx = (OWX + RNX - 1) / RNX, nty = (OWY + RNY - 1) / RNY;
      const int bid = (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)); // XCD-aware launch, pfm_internal.h
      if (bid >= ntx * nty * ((cv.NZ - 1 + zc) / zc))
        return;
      const int tix = bid % ntx, tiy = (bid / ntx) % nty, chunk = bid / (ntx * nty);
      const int i0 = tix * RNX, j0 = tiy * RNY;
      const int kA = chunk * zc;
      const int kB = min(kA + zc, cv.NZ); // node planes [kA, kB)
      const int ci = i0 - 1 + cx, cj = j0 - 1 + cy;
      const bool col_ok = ci >= 0 && ci < cv.NX - 1 && cj >= 0 && cj < cv.NY - 1;
      const bool node_ok = cx < RNX && cy < RNY && (i0 + cx) < cv.NX && (j0 + cy) < cv.NY;
      const unsigned plane = (unsigned)cv.NX * (unsigned)cv.NY;

      for (int idx = t; idx < RPD; idx += RTX * RTY)
        {
          const int q = min(idx, RHX * RHY - 1);
          const int gi = min(max(i0 - 1 + q % RHX, 0), cv.NX - 1), gj = min(max(j0 - 1 + q / RHX, 0), cv.NY - 1);
          s.nofs[idx] = (unsigned)gi + (unsigned)cv.NX * (unsigned)gj;
        }
      // where the six fields lie: base (uniform), byte offset of node n = ((n [* 3]) << sh) + add
      const bool fused = v.fused_solution != nullptr, il = v.layout == PFM_LAYOUT_INTERLEAVED;
      const void *fb[6];
      unsigned ftri[6], fsh[6], fadd[6];
#pragma unroll
      for (int f = 0; f < 6; ++f)
        {
          fb[f] = f == 0 ? v.u[0] : f == 1 ? v.u[1] : f == 2 ? v.u[2] : f == 3 ? v.phi : f == 4 ? v.phi_old : v.phi_oldold;
          ftri[f] = 0u, fsh[f] = 3u, fadd[f] = 0u;
          if (fused && f < 4)
            {
              fb[f] = (f == 3 && !il) ? v.fused_solution + 3LL * v.n_owned : v.fused_solution;
              ftri[f] = (f < 3 && !il) ? 0xffffffffu : 0u;
              fsh[f] = il ? 5u : 3u;
              fadd[f] = (f < 3 || il) ? 8u * f : 0u;
            }
        }
      __syncthreads();
      auto request_plane = [&](int kz, int slot) __attribute__((always_inline)) {
        const unsigned pk = (unsigned)min(max(kz, 0), cv.NZ - 1) * plane;
        int lq = lane;
        asm volatile("" : "+v"(lq)); // recomputed per step, not kept live across the march
#pragma unroll
        for (int r = 0; r < 3; ++r)
          {
            const int jj = wv + 4 * r; // waves 0, 1: transfers {w, w + 4, w + 8} of every field, waves 2, 3: {w, w + 4}
            if (jj < 10)
              {
                const unsigned n = s.nofs[32 * jj + (lq >> 1)] + pk;
                unsigned bo[6];
                void *dst[6];
#pragma unroll
                for (int f = 0; f < 6; ++f)
                  {
                    const unsigned ne = ((n & ftri[f]) << 1) + n;
                    bo[f] = (ne << fsh[f]) + fadd[f] + 4u * (lq & 1);
                    dst[f] = reinterpret_cast<uint32_t *>(&s.U[slot][f][0]) + 64 * jj;
                  }
                dma_b32x6(fb, bo, dst);
              }
          }
      };
      // single rank, solution vector read in place: the node state is what pfm_state_set_solution would have left -- every
      // node is written once, by the tile and z-chunk that own it, from the plane that has just landed
      auto publish_plane = [&](int kz, int slot) __attribute__((always_inline)) {
        if (fused && kz >= kA && kz < kB)
          {
            int tq = t;
            asm volatile("" : "+v"(tq));
#pragma unroll
            for (int r = 0; r < 2; ++r)
              {
                const int idx = tq + r * RTX * RTY;
                const int hx = idx % RHX, hy = idx / RHX;
                if (idx < RHX * RHY && hx >= 1 && hx <= RNX && hy >= 1 && hy <= RNY && i0 - 1 + hx < cv.NX && j0 - 1 + hy < cv.NY)
                  {
                    const unsigned n = s.nofs[idx] + (unsigned)kz * plane;
                    v.u[0][n] = s.U[slot][0][idx];
                    v.u[1][n] = s.U[slot][1][idx];
                    v.u[2][n] = s.U[slot][2][idx];
                    v.phi[n] = s.U[slot][3][idx];
                  }
              }
          }
      };

      double M[2][2][2][4]; // the moment basis of k_cart_residual3
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          M[a & 1][(a >> 1) & 1][a >> 2][c] = 0.0;

      request_plane(kA - 1, 0);
      request_plane(kA, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      publish_plane(kA, 1);
      const int hb = cy * RHX + cx;
      int sl = 0; // ring slot of plane ck
#pragma unroll 1
      for (int ck = kA - 1; ck < kB; ++ck)
        {
          const int sh = sl == 2 ? 0 : sl + 1, sn = sh == 2 ? 0 : sh + 1;
          const int par = (ck - kA) & 1;
          if (ck + 2 <= kB) // (slot sn held plane ck - 1: every wave is past the barrier behind that layer's arithmetic)
            request_plane(ck + 2, sn);
          if (col_ok && ck >= 0 && ck < cv.NZ - 1)
            {
              const double *Ulo = &s.U[sl][0][hb], *Uhi = &s.U[sh][0][hb];
              double lam = S.lam, mu = S.mu;
              if constexpr (HET) // heterogeneous material, cracks.cc:2207-2216
                {
                  const long long cidx = ci + (long long)(cv.NX - 1) * (cj + (long long)(cv.NY - 1) * ck);
                  lam = cv.cell_lam[cidx];
                  mu = cv.cell_mu[cidx];
                }
              residual_cell_poly<PlaneSoA<RPD>, true>(Ulo, Uhi, Ulo, Uhi, S, lam, mu, M);
            }
          const bool emit = ck >= kA;
          if (emit)
            {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                {
                  const double l01 = M[0][1][0][c] - M[0][1][1][c], l11 = M[1][1][0][c] - M[1][1][1][c];
                  const double l00 = (M[0][0][0][c] - M[0][0][1][c]) - l01, l10 = (M[1][0][0][c] - M[1][0][1][c]) - l11;
                  s.P[par][0 * 4 + c][t] = l10 + row_shl1(l00 - l10);
                  s.P[par][1 * 4 + c][t] = l11 + row_shl1(l01 - l11);
                }
            }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // plane ck + 2 has landed (and the stores of the last step are out)
          __syncthreads();
          if (emit && node_ok)
            {
              const unsigned row = (unsigned)(i0 + cx) + (unsigned)cv.NX * (unsigned)(j0 + cy) + (unsigned)ck * plane;
              const unsigned fl = v.node_flags[row];
#pragma unroll
              for (int c = 0; c < 4; ++c)
                {
                  const double r = -s.P[par][4 + c][t] - s.P[par][c][t + RTX];
                  const bool con = (fl >> c) & 1u;
                  const long long di = dof_index_c<3>(v, (int)row, c);
                  res_pde[di] = con ? 0.0 : r; // constrained scatter = masked store (cracks.cc:2440-2456)
                  if (write_total)
                    res_tot[di] = (con && S.total_via_update) ? 0.0 : r;
                }
            }
          publish_plane(ck + 2, sn);
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              {
                M[a & 1][a >> 1][0][c] = M[a & 1][a >> 1][1][c];
                M[a & 1][a >> 1][1][c] = 0.0;
              }
          sl = sh;
        }
    }


    // =====================================================================================
    // k_cart_residual3x: k_cart_residual3d for the solution vector of a single rank in the BLOCKED layout, read in place,
    // with 16-byte transfers.  A transfer instruction costs the wave ~150 cycles whatever its width (phase clock of
    // k_cart_residual3d: 60 dword transfers per plane, 2750 cycles per step in the wave that issues 18 of them), so
    // the plane is fetched in 16 instead of 60:
    //   * the displacements stay interleaved as they lie in the vector: a halo row is 17 nodes x 3 doubles = 51 doubles in
    //     a row of 52 (26 lanes; the last double belongs to the next node), 17 rows = 442 lanes = 7 transfers;
    //   * a phase-field plane is 17 rows of 18 doubles (9 lanes: pairs of x-neighbours; the 18th column is the next
    //     node of the lattice row), 153 lanes = 3 transfers per field.
    // The first / last pair of the whole lattice would reach in front of / behind the arrays: the workgroup and plane
    // that own them fall back to dword transfers, node by node (clamped to the box like every halo node outside it).
    // =====================================================================================
    constexpr int XUR = 52, XSR = 18;                       // row strides (doubles) of the interleaved displacements / a phase field
    constexpr int XUO = 0, XUN = RHY * XUR;                 // 884 doubles = 442 lanes
    constexpr int XSN = RHY * XSR;                          // 306 doubles = 153 lanes
    constexpr int XPL = XUN + 3 * XSN;                      // doubles per ring slot
    struct PlaneX
    {
      static constexpr int off(int f) { return f < 3 ? f : (f - 3) * XSN; }
      static constexpr int xs(int f) { return f < 3 ? 3 : 1; }
      static constexpr int ys(int f) { return f < 3 ? XUR : XSR; }
    };
    struct LdsR3X
    {
      double U[3][XPL];          // [ring][u interleaved | phi | phi_old | phi_oldold]; first: M0 holds a 16-bit LDS offset
      double P[2][8][RTX * RTY]; // [parity of the step][a_y * 4 + component]
      unsigned tu[448];          // byte offset of the lane's 16 bytes of the displacement rows, plane 0
      unsigned ts[192];          // node id of the lane's pair of a phase-field plane, plane 0
    };
    static_assert(sizeof(LdsR3X) <= 81920 && (XUN % 2) == 0 && (XSN % 2) == 0 && (XPL % 2) == 0, "two workgroups per CU; 16-byte lanes");

    template <bool HET>
    __global__ __launch_bounds__(RTX *RTY, 2) void k_cart_residual3x(DevView v, CartView cv, Scal S, double *__restrict__ res_pde,
                                                                  double *__restrict__ res_tot, int write_total, int zc)
    {
      __shared__ LdsR3X s;
      const int t = threadIdx.x, cx = t % RTX, cy = t / RTX;
      const int wv = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
      const int ntx = (cv.NX + RNX - 1) / RNX, nty = (cv.NY + RNY - 1) / RNY;
      const int bid = (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)); // XCD-aware launch, pfm_internal.h
      if (bid >= ntx * nty * ((cv.NZ - 1 + zc) / zc))
        return;
      const int tix = bid % ntx, tiy = (bid / ntx) % nty, chunk = bid / (ntx * nty);
      const int i0 = tix * RNX, j0 = tiy * RNY;
      const int kA = chunk * zc;
      const int kB = min(kA + zc, cv.NZ); // node planes [kA, kB)
      const int ci = i0 - 1 + cx, cj = j0 - 1 + cy;
      const bool col_ok = ci >= 0 && ci < cv.NX - 1 && cj >= 0 && cj < cv.NY - 1;
      const bool node_ok = cx < RNX && cy < RNY && (i0 + cx) < cv.NX && (j0 + cy) < cv.NY;
      const unsigned plane = (unsigned)cv.NX * (unsigned)cv.NY;
      const double *const sol = v.fused_solution, *const solp = v.fused_solution + 3LL * v.n_owned;

      for (int q = t; q < 448; q += RTX * RTY) // (rows outside the box: the nearest row inside; columns: wherever the lattice row leads)
        {
          const int hy = min(q / 26, RHY - 1), pr = q % 26;
          const int gj = min(max(j0 - 1 + hy, 0), cv.NY - 1);
          s.tu[q] = 24u * (unsigned)(i0 - 1 + cv.NX * gj) + 16u * (unsigned)pr;
        }
      for (int q = t; q < 192; q += RTX * RTY)
        {
          const int hy = min(q / 9, RHY - 1), m = q % 9;
          const int gj = min(max(j0 - 1 + hy, 0), cv.NY - 1);
          s.ts[q] = (unsigned)(i0 - 1 + 2 * m + cv.NX * gj);
        }
      __syncthreads();
      // 16 transfers per plane, 4 per wave: job g = wave + 4 r; jobs 0..6: displacement rows, 7..15: field (g - 7) / 3, part (g - 7) % 3
      auto request_plane = [&](int kz, int slot) __attribute__((always_inline)) {
        const int kc = min(max(kz, 0), cv.NZ - 1);
        const unsigned pk = (unsigned)kc * plane;
        int lq = lane;
        asm volatile("" : "+v"(lq)); // recomputed per step, not kept live across the march
        const bool edge = (kc == 0 && i0 == 0 && j0 == 0) || (kc == cv.NZ - 1 && i0 + RHX >= cv.NX - 1 && j0 + RHY - 1 >= cv.NY - 1);
        if (!edge)
          {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              {
                const int g = wv + 4 * r;
                if (g < 7)
                  {
                    const int q = 64 * g + lq;
                    if (q < XUN / 2)
                      dma_b128(sol, s.tu[q] + 24u * pk, &s.U[slot][XUO + 128 * g]);
                  }
                else
                  {
                    const int f = (g - 7) / 3, jj = (g - 7) % 3; // uniform
                    const double *const fb = f == 0 ? solp : f == 1 ? v.phi_old : v.phi_oldold;
                    const int q = 64 * jj + lq;
                    if (q < XSN / 2)
                      dma_b128(fb, 8u * (s.ts[q] + pk), &s.U[slot][XUN + f * XSN + 128 * jj]);
                  }
              }
          }
        else
          {
            // the planes that hold the first / the last node of the lattice, in the workgroups that reach them
#pragma unroll 1
            for (int g = wv; g < 28 + 30; g += 4)
              {
                if (g < 28)
                  {
                    const int d = 64 * g + lq, e = d >> 1; // dword d of the displacement rows
                    const int hy = min(e / XUR, RHY - 1), rem = e % XUR, hx = min(rem / 3, RHX - 1), c = rem % 3;
                    const int gi = min(max(i0 - 1 + hx, 0), cv.NX - 1), gj = min(max(j0 - 1 + hy, 0), cv.NY - 1);
                    const unsigned n = (unsigned)gi + (unsigned)cv.NX * (unsigned)gj + pk;
                    if (d < 2 * XUN)
                      dma_b32(sol, 24u * n + 8u * (unsigned)c + 4u * (d & 1), reinterpret_cast<uint32_t *>(&s.U[slot][XUO]) + 64 * g);
                  }
                else
                  {
                    const int f = (g - 28) / 10, jj = (g - 28) % 10;
                    const double *const fb = f == 0 ? solp : f == 1 ? v.phi_old : v.phi_oldold;
                    const int d = 64 * jj + lq, e = d >> 1;
                    const int hy = min(e / XSR, RHY - 1), hx = min(e % XSR, RHX - 1);
                    const int gi = min(max(i0 - 1 + hx, 0), cv.NX - 1), gj = min(max(j0 - 1 + hy, 0), cv.NY - 1);
                    const unsigned n = (unsigned)gi + (unsigned)cv.NX * (unsigned)gj + pk;
                    if (d < 2 * XSN)
                      dma_b32(fb, 8u * n + 4u * (d & 1), reinterpret_cast<uint32_t *>(&s.U[slot][XUN + f * XSN]) + 64 * jj);
                  }
              }
          }
      };
      // the node state is what pfm_state_set_solution would have left: every node is written once, by the tile and
      // z-chunk that own it, from the plane that has just landed
      auto publish_plane = [&](int kz, int slot) __attribute__((always_inline)) {
        if (kz >= kA && kz < kB)
          {
            int tq = t;
            asm volatile("" : "+v"(tq));
            const int hx = 1 + tq % RNX, hy = 1 + tq / RNX;
            if (tq < RNX * RNY && i0 - 1 + hx < cv.NX && j0 - 1 + hy < cv.NY)
              {
                const unsigned n = (unsigned)(i0 - 1 + hx) + (unsigned)cv.NX * (unsigned)(j0 - 1 + hy) + (unsigned)kz * plane;
                const double *pu = &s.U[slot][XUO + hy * XUR + 3 * hx];
                v.u[0][n] = pu[0];
                v.u[1][n] = pu[1];
                v.u[2][n] = pu[2];
                v.phi[n] = s.U[slot][XUN + hy * XSR + hx];
              }
          }
      };

      double M[2][2][2][4]; // the moment basis of k_cart_residual3
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          M[a & 1][(a >> 1) & 1][a >> 2][c] = 0.0;

      const int hbu = XUO + cy * XUR + 3 * cx, hbs = XUN + cy * XSR + cx;
      int sl = 1; // ring slot of plane ck: plane kA - 1 in slot 0.  The first two steps only request (one call site)
#pragma unroll 1
      for (int ck = kA - 3; ck < kB; ++ck)
        {
          const int sh = sl == 2 ? 0 : sl + 1, sn = sh == 2 ? 0 : sh + 1;
          const int par = (ck - kA) & 1;
          if (ck + 2 <= kB) // (slot sn held plane ck - 1: every wave is past the barrier behind that layer's arithmetic)
            request_plane(ck + 2, sn);
          if (col_ok && ck >= max(kA - 1, 0) && ck < cv.NZ - 1)
            {
              double lam = S.lam, mu = S.mu;
              if constexpr (HET) // heterogeneous material, cracks.cc:2207-2216
                {
                  const long long cidx = ci + (long long)(cv.NX - 1) * (cj + (long long)(cv.NY - 1) * ck);
                  lam = cv.cell_lam[cidx];
                  mu = cv.cell_mu[cidx];
                }
              residual_cell_poly<PlaneX, true>(&s.U[sl][hbu], &s.U[sh][hbu], &s.U[sl][hbs], &s.U[sh][hbs], S, lam, mu, M);
            }
          const bool emit = ck >= kA;
          if (emit)
            {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                {
                  const double l01 = M[0][1][0][c] - M[0][1][1][c], l11 = M[1][1][0][c] - M[1][1][1][c];
                  const double l00 = (M[0][0][0][c] - M[0][0][1][c]) - l01, l10 = (M[1][0][0][c] - M[1][0][1][c]) - l11;
                  s.P[par][0 * 4 + c][t] = l10 + row_shl1(l00 - l10);
                  s.P[par][1 * 4 + c][t] = l11 + row_shl1(l01 - l11);
                }
            }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // plane ck + 2 has landed (and the stores of the last step are out)
          __syncthreads();
          if (emit && node_ok)
            {
              const unsigned row = (unsigned)(i0 + cx) + (unsigned)cv.NX * (unsigned)(j0 + cy) + (unsigned)ck * plane;
              const unsigned fl = v.node_flags[row];
#pragma unroll
              for (int c = 0; c < 4; ++c)
                {
                  const double r = -s.P[par][4 + c][t] - s.P[par][c][t + RTX];
                  const bool con = (fl >> c) & 1u;
                  const long long di = dof_index_c<3>(v, (int)row, c);
                  res_pde[di] = con ? 0.0 : r; // constrained scatter = masked store (cracks.cc:2440-2456)
                  if (write_total)
                    res_tot[di] = (con && S.total_via_update) ? 0.0 : r;
                }
            }
          publish_plane(ck + 2, sn);
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              {
                M[a & 1][a >> 1][0][c] = M[a & 1][a >> 1][1][c];
                M[a & 1][a >> 1][1][c] = 0.0;
              }
          sl = sh;
        }
    }

    bool g_tab_ready[16] = {};
    int ensure_tab()
    {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess)
        return PFM_ERR_HIP;
      if (dev < 16 && g_tab_ready[dev])
        return PFM_OK;
      const Tab1D t = make_tab1d();
      if (hipMemcpyToSymbol(HIP_SYMBOL(c_t1), &t, sizeof(t)) != hipSuccess)
        return PFM_ERR_HIP;
      if (dev < 16)
        g_tab_ready[dev] = true;
      return PFM_OK;
    }
  } // namespace

  int launch_cart_matrix(const DevView &v, const CartView &cv, const pfm_params &p, double *const *d_values,
                         hipStream_t s, void *d_scal, double *res_pde, int phase, hipStream_t s_phi);

  int choose_zchunk(long long tiles, int planes, int zc_min, int zc_max, int per_cu)
  {
    static int n_cu = 0;
    if (!n_cu)
      {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
      }
    // time model: workgroups are dispatched as slots free up, so the launch takes about (wgs / slots + 1/2) workgroup
    // durations, and a workgroup's duration is proportional to its zc + 1 cell layers
    const double slots = (double)per_cu * n_cu;
    const int lo = std::max(1, std::min(zc_min, planes)), hi = std::min(zc_max, planes);
    double tmin = 1e300;
    for (int zc = lo; zc <= hi; ++zc)
      tmin = std::min(tmin, ((double)tiles * ((planes + zc - 1) / zc) / slots + 0.5) * (zc + 1));
    int best = lo;
    for (int zc = lo; zc <= hi; ++zc) // the longest chunk within 1 % of the optimum: fewest redundant layers
      if (((double)tiles * ((planes + zc - 1) / zc) / slots + 0.5) * (zc + 1) <= 1.01 * tmin)
        best = zc;
    return best;
  }

  // the conditions of `rows_residual` in launch_assemble_cart + what the pair needs: the default (u,u) kernel
  bool cart_jacobian_pair(const DevView &v, const CartView &cv, const pfm_params &p, int residual_only, int phase)
  {
    if (v.dim != 3 || residual_only || phase != 0 || cv.cell_lam)
      return false;
    static const bool other_mode = getenv("PFM_RES_KERNEL") || getenv("PFM_UU_CLK") || getenv("PFM_PHI_CLK");
    if ((p.decompose_stress_matrix > 0 && p.timestep_number > 0) || other_mode)
      return false;
    const Scal S = make_scal(p, cv, v.dim);
    return !S.monolithic && S.gamma_fac == 0.0 && S.kappa < 0.5;
  }

  // phase: 0 = the whole assembly; 1 / 2 = the two halves of pfm_assemble_overlapped: 1 launches what reads no ghost
  // node (the "interior" tiles of the first kernel of the sequence), 2 the rest -- the ghost import lands in between
  int launch_assemble_cart(const DevView &v, const CartView &cv_in, const pfm_params &p, int residual_only,
                           double *const *d_values, double *res_pde, double *res_tot, hipStream_t s,
                           hipStream_t s_residual, void *d_scal, int phase)
  {
    // the residual and the Jacobian only read the node state: on different streams they overlap
    // (s_residual == s: plain stream order)
    hipStream_t s_jac = s;
    s = s_residual;
    int rc = ensure_tab();
    if (rc)
      return rc;
    const bool split = p.decompose_stress_matrix > 0 && p.timestep_number > 0; // cracks.cc:2294
    if (split)
      return PFM_ERR_UNSUPPORTED; // the host routes split runs to the general path
    CartView cv = cv_in;
    cv.tile_sel = phase; // 0: all tiles, 1: interior, 2: boundary -- of the FIRST kernel of the sequence only (below)
    if (v.dim == 2 && !residual_only)
      return launch_cart2d(v, cv, p, residual_only, d_values, res_pde, res_tot, s_jac, s); // 2-D Jacobian + residual (s != s_jac: forked by the caller)
    const Scal S = make_scal(p, cv, v.dim);
    // Full 3-D assembly, staggered scheme (no q-point clamps of the phase fields, no penalty term): the unsplit law makes
    // every residual row an exact function of its own matrix row (R_u = pressure part - K_uu u, R_phi = G_c/eps mass - K_phiphi
    // phi), so the Jacobian kernels write the residual as well and the residual kernel is not launched (2.1 of 15.8 ms at
    // 216^3).  PFM_RES_KERNEL=1 keeps the quadrature kernel (A/B runs; tests compare both against the oracle).
    static const bool res_kernel_forced = getenv("PFM_RES_KERNEL") != nullptr;
    const bool rows_residual = v.dim == 3 && !residual_only && !S.monolithic && S.gamma_fac == 0.0 && S.kappa < 0.5 && !res_kernel_forced &&
                               !cv.cell_lam; // (the heterogeneous (u,u) variant has no registers left for it)
    if (rows_residual) // s_residual != s_jac: the caller forked it off for the phase-field kernel (cart_jacobian_pair)
      return launch_cart_matrix(v, cv, p, d_values, s_jac, d_scal, res_pde, phase, (phase == 0 && cv.patch_count) ? s : s_jac);
    const long long OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1, OWZ = v.dim == 3 ? cv.o1[2] - cv.o0[2] + 1 : 1;
    if (v.dim == 2)
      {
        const int ntx = (int)((OWX + R2N - 1) / R2N);
        static const int zc_force = getenv("PFM_RES2_ZC") ? atoi(getenv("PFM_RES2_ZC")) : 0; // tuning only
        const int zc = zc_force > 0 ? zc_force : choose_zchunk(ntx, (int)OWY, 4, 64, 8);
        const unsigned nw = (unsigned)(ntx * ((OWY + zc - 1) / zc));
        if (nw == 0)
          ;
        else if (!S.monolithic && S.gamma_fac == 0.0)
          hipLaunchKernelGGL(k_cart_residual2m<true>, dim3(nw), dim3(64), 0, s, v, cv, S, res_pde, res_tot, residual_only, zc);
        else
          hipLaunchKernelGGL(k_cart_residual2m<false>, dim3(nw), dim3(64), 0, s, v, cv, S, res_pde, res_tot, residual_only, zc);
      }
    else
      {
        const int ntx = (int)((OWX + RNX - 1) / RNX), nty = (int)((OWY + RNY - 1) / RNY);
        // chunks of z-planes: fill the dispatch rounds of the chip (2 workgroups per CU) at few redundant layers
        static const int zc_force = getenv("PFM_RES_ZC") ? atoi(getenv("PFM_RES_ZC")) : 0; // tuning only
        const int zc = zc_force > 0 ? zc_force : choose_zchunk((long long)ntx * nty, (int)OWZ, 4, 24, 2);
        const int nch = (int)((OWZ + zc - 1) / zc);
        const bool listed = cv.tile_sel == 2 && cv.bnd_res3 != nullptr && cv.zc_res3 == zc;
        const unsigned nt = listed ? (unsigned)cv.n_bnd_res3 : (unsigned)(ntx * nty * nch);
        // the whole lexicographic box of a single rank, every byte offset of a node below 4 GiB: planes by transfer
        // (read per launch: the tests compare the three kernels in one process)
        const bool no_transfers = getenv("PFM_RES_NO_TRANSFERS") != nullptr;   // k_cart_residual3 <true>
        const bool wide_off = getenv("PFM_RES_NO_WIDE_TRANSFERS") != nullptr;  // k_cart_residual3d instead of 3x
        const bool whole_lex = cv.owned_lex && cv.o0[0] == 0 && cv.o0[1] == 0 && cv.o0[2] == 0 && cv.o1[0] == cv.NX - 1 &&
                               cv.o1[1] == cv.NY - 1 && cv.o1[2] == cv.NZ - 1 && cv.tile_sel == 0 && !cv.row_of_box &&
                               (long long)v.n_nodes == (long long)cv.NX * cv.NY * cv.NZ && v.n_owned == v.n_nodes &&
                               (long long)v.n_nodes * 32 < (1LL << 32) && !no_transfers;
        if (nt == 0)
          ;
        else if (!S.monolithic && S.gamma_fac == 0.0 && whole_lex && v.fused_solution && v.layout == PFM_LAYOUT_BLOCKED && v.n_nodes >= 64 &&
                 !wide_off)
          {
            if (cv.cell_lam)
              hipLaunchKernelGGL(k_cart_residual3x<true>, dim3(xcd_grid(nt)), dim3(RTX * RTY), 0, s, v, cv, S, res_pde, res_tot, residual_only, zc);
            else
              hipLaunchKernelGGL(k_cart_residual3x<false>, dim3(xcd_grid(nt)), dim3(RTX * RTY), 0, s, v, cv, S, res_pde, res_tot, residual_only, zc);
          }
        else if (!S.monolithic && S.gamma_fac == 0.0 && whole_lex)
          {
            if (cv.cell_lam)
              hipLaunchKernelGGL(k_cart_residual3d<true>, dim3(xcd_grid(nt)), dim3(RTX * RTY), 0, s, v, cv, S, res_pde, res_tot, residual_only, zc);
            else
              hipLaunchKernelGGL(k_cart_residual3d<false>, dim3(xcd_grid(nt)), dim3(RTX * RTY), 0, s, v, cv, S, res_pde, res_tot, residual_only, zc);
          }
        else if (!S.monolithic && S.gamma_fac == 0.0)
          hipLaunchKernelGGL(k_cart_residual3<true>, dim3(xcd_grid(nt)), dim3(RTX * RTY), 0, s, v, cv, S, res_pde, res_tot, residual_only, zc);
        else
          hipLaunchKernelGGL(k_cart_residual3<false>, dim3(xcd_grid(nt)), dim3(RTX * RTY), 0, s, v, cv, S, res_pde, res_tot, residual_only, zc);
      }
    if (hipGetLastError() != hipSuccess)
      return PFM_ERR_HIP;
    // the residual kernel was the first of the sequence (and the one split into interior / boundary tiles): the Jacobian
    // kernels follow it completely, in the second phase of an overlapped assembly
    if (!residual_only && phase != 1)
      return launch_cart_matrix(v, cv_in, p, d_values, s_jac, d_scal, nullptr, 0, s_jac);
    return PFM_OK;
  }
} // namespace pfm
namespace pfm
{
  void cart_res3_boundary_tiles(const CartView &cv, std::vector<int32_t> &out, int &zc)
  {
    out.clear();
    const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1, OWZ = cv.o1[2] - cv.o0[2] + 1;
    const int ntx = (OWX + RNX - 1) / RNX, nty = (OWY + RNY - 1) / RNY;
    static const int zc_force = getenv("PFM_RES_ZC") ? atoi(getenv("PFM_RES_ZC")) : 0; // as launch_assemble_cart
    zc = zc_force > 0 ? zc_force : choose_zchunk((long long)ntx * nty, OWZ, 4, 24, 2);
    const int nch = (OWZ + zc - 1) / zc;
    for (int ch = 0; ch < nch; ++ch)
      for (int tiy = 0; tiy < nty; ++tiy)
        for (int tix = 0; tix < ntx; ++tix)
          {
            const int i0 = cv.o0[0] + tix * RNX, j0 = cv.o0[1] + tiy * RNY, kA = cv.o0[2] + ch * zc;
            const int kB = std::min(kA + zc, cv.o1[2] + 1);
            if (cart_range_has_ghost(cv, 0, i0 - 1, i0 + RNX) || cart_range_has_ghost(cv, 1, j0 - 1, j0 + RNY) ||
                cart_range_has_ghost(cv, 2, kA - 1, kB))
              out.push_back(tix + ntx * (tiy + nty * ch));
          }
  }
} // namespace pfm
