// pfm_cart2d.hip — 2-D Jacobian + residual kernels for uniform Cartesian boxes (cracks.cc:2200-2464): the 2-D Sneddon
// configurations (tests/sneddon_2d_1.prm on a uniform mesh, BASELINE config 2 with the matrix).  Runs with the stress split
// of cracks.cc:2294 active stay on the general family (the linearised split is not a moment of a q-point field; the
// launcher refuses them and the host routes them there).
//
// No atomics, no zeroing pass, every value of a row written exactly once, constraints applied as masks, bitwise
// reproducible:
//   k_cart2d_cells: wave <-> block of 8 x 8 cells, lane <-> cell.  The 9 q-point states of a cell are
//     evaluated ONCE and reduced to 59 moments; the rows of the cell's four vertices are formed from the moments and handed
//     to the lanes that own the nodes (ds_bpermute), which complete their rows in the order of a lexicographic cell loop.
//     Block rows are staged in LDS and stored as whole cache lines.  0.31 ms at 1000^2 (the thread-per-node row owner
//     of round 2, removed in round 4: 0.85 ms).
#include "pfm_internal.h"
#include "pfm_cart_common.h"

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>

namespace pfm
{
  namespace
  {
    struct Vals2
    {
      double *b[4];
    };

    struct Prm2 // resolved scalars
    {
      double lam, mu, kappa, eps, Gc, p, aB1, penal_fac, tfac, ihx, ihy, vol;
      int monolithic, use_old;
    };

    // =====================================================================================
    // One wave <-> a block of 8 x 8 cells, lane <-> cell, sum-factorised.
    //
    // The lattice is uniform and the law unsplit, so every entry of the element matrix is a moment of a q-point field
    // against products of 1-D shape functions: lane (c_x, c_y) evaluates the 9 q-point states of ITS cell once and
    // reduces them to 59 moments (x-sums per q_y line, then the y-factors), forms the rows of vertex A = 3, 2, 1, 0 from
    // them and hands the rows of the three vertices it does not own to the lanes that do (lane + 9, + 8, + 1: the
    // node of lane (c_x, c_y) is the lower-left vertex of its cell; the receiver pulls through ds_bpermute).  A node
    // receives its four cells in the order of a lexicographic cell loop as before; lanes of block row / column 0 only
    // contribute (7 x 7 owned nodes per wave: 1.31 cell evaluations per node instead of 4, ~1.2k instead of ~1.8k
    // operations per evaluation).  Rows c = 0, 1 are completed and stored first, then the phase-field row: the moments
    // and the accumulators of one group fit the registers of two waves per SIMD.  No LDS allocation, no barrier.
    // =====================================================================================
    constexpr int B2 = 8, O2 = B2 - 1; // cells per block side, owned nodes per block side

    struct Cst2 // per-launch constants of the sum-factorised kernel
    {
      double omk2, aB1p2, aB1p, omk, gc_eps, gc_eps_vol_x, gc_eps_vol_y; // 2(1-kappa), 2(aB-1)p, (aB-1)p, 1-kappa, G_c/eps, G_c eps vol/h_x^2, .../h_y^2
      double gce;                                                        // G_c eps
    };

    // GROUP 0: the displacement rows (c = 0, 1) of every node, GROUP 1: the phase-field row, 2: all three.  Round 6: one
    // launch per group.  The 59 moments, the accumulators of a group and the nodal values do not fit 256 registers: the
    // single kernel spilled 30 of them, and at two waves per SIMD under a stream of stores the scratch lines do not
    // stay in the L2 -- counters at 1000^2: 921 MB written, 309 MB fetched for 697 MB of rows and ~90 MB of input.  Each
    // group needs its own moments only (the compiler drops the others); the q-point states are evaluated twice, by
    // SIMDs that were 16 % busy.
#ifndef PFM_C2_OCC0
#define PFM_C2_OCC0 2
#endif
#ifndef PFM_C2_OCC1
#define PFM_C2_OCC1 2
#endif
    template <int GROUP>
    __global__ __launch_bounds__(64, GROUP == 0 ? PFM_C2_OCC0 : GROUP == 1 ? PFM_C2_OCC1 : 2) void k_cart2d_cells(DevView v, CartView cv, Prm2 P, Cst2 K, Vals2 vals, double *__restrict__ res_pde,
                                                            double *__restrict__ res_tot, int write_total, int total_via_update, int n_blocks_y)
    {
      const int lane = threadIdx.x, cx = lane & (B2 - 1), cy = lane >> 3;
      const int OWX = cv.o1[0] - cv.o0[0] + 1;
      const int ntx = (OWX + O2 - 1) / O2;
      // XCD-aware launch (pfm_internal.h): neighbouring blocks read the same nodes and meet in the cache lines at the ends
      // of their block rows, and only blocks of one XCD meet in an L2 (0.297 -> 0.285 ms at 1000^2)
      const int bid = xcd_tile_index();
      if (bid >= ntx * n_blocks_y)
        return;
      const int tix = bid % ntx, tiy = bid / ntx;
      // this lane's cell (i, j) = its lower-left node
      const int i = cv.o0[0] + tix * O2 + cx - 1, j = cv.o0[1] + tiy * O2 + cy - 1;
      const bool owner = cx >= 1 && cy >= 1 && i <= cv.o1[0] && j <= cv.o1[1];
      bool cell_ok = i >= 0 && i < cv.NX - 1 && j >= 0 && j < cv.NY - 1;

      // ---- ids first, then everything that depends on them, in two rounds of independent loads.  (Round 6: as
      // cart_local_id() per vertex / neighbour -- arithmetic or table, decided in a branch -- the compiler waited with
      // vmcnt(0) at every join: 4 + 9 round trips one behind the other in front of the arithmetic, a third of the wave's
      // life at two waves per SIMD.)  The lattice table holds every node of the box: always the table, clamped index.
      const long long NXl = cv.NX;
      int nb[4], q[9];
      // the whole box of a single rank, numbered lexicographically: ids by arithmetic -- one round of loads instead of two.
      // (A uniform branch: the wait the compiler puts at its join finds nothing else in flight.)
      const bool lex = cv.owned_lex && cv.o0[0] == 0 && cv.o0[1] == 0 && cv.o1[0] == cv.NX - 1 && cv.o1[1] == cv.NY - 1;
      if (lex)
        {
#pragma unroll
          for (int b = 0; b < 4; ++b)
            nb[b] = cell_ok ? (i + (b & 1)) + cv.NX * (j + (b >> 1)) : 0;
#pragma unroll
          for (int o = 0; o < 9; ++o)
            {
              const int qi = i + (o % 3) - 1, qj = j + (o / 3) - 1;
              q[o] = (qi >= 0 && qi < cv.NX && qj >= 0 && qj < cv.NY) ? qi + cv.NX * qj : -1;
            }
        }
      else
        {
#pragma unroll
          for (int b = 0; b < 4; ++b)
            nb[b] = cv.local_of_box[cell_ok ? (i + (b & 1)) + NXl * (j + (b >> 1)) : 0];
#pragma unroll
          for (int o = 0; o < 9; ++o)
            {
              const int qi = i + (o % 3) - 1, qj = j + (o / 3) - 1;
              const bool in = qi >= 0 && qi < cv.NX && qj >= 0 && qj < cv.NY;
              const int id = cv.local_of_box[in ? qi + NXl * qj : 0];
              q[o] = in ? id : -1;
            }
        }
      // ---- nodal data of the cell: [field][vertex b = b_x + 2 b_y], fields u_x u_y phi phi_old phi_oldold
      double F[5][4];
      cell_ok = cell_ok && nb[0] >= 0 && nb[1] >= 0 && nb[2] >= 0 && nb[3] >= 0; // a cell this rank does not know completely touches none of its rows
#pragma unroll
      for (int b = 0; b < 4; ++b)
        {
          const int n = cell_ok ? nb[b] : 0;
          F[0][b] = v.u[0][n];
          F[1][b] = v.u[1][n];
          F[2][b] = v.phi[n];
          F[3][b] = v.phi_old[n];
          F[4][b] = v.phi_oldold[n];
        }
      // ---- the node of this lane: row info and the constraint flags of the 9 lattice neighbours
      bool writes = owner;
      if (owner && cart_tile_skipped(cv, cart_range_has_ghost(cv, 0, i - 1, i + 1) || cart_range_has_ghost(cv, 1, j - 1, j + 1)))
        writes = false; // overlapped assembly: the other launch owns this node's rows
      const int row = writes ? q[4] : 0;
      unsigned fN = 0u; // 3 flag bits per lattice offset o
      {
        unsigned fl[9];
#pragma unroll
        for (int o = 0; o < 9; ++o)
          fl[o] = v.node_flags[q[o] >= 0 ? q[o] : 0];
#pragma unroll
        for (int o = 0; o < 9; ++o)
          fN |= ((writes && q[o] >= 0) ? (fl[o] & 7u) : 0u) << (3 * o);
      }
      const unsigned fP_l = v.node_flags[row], mask_l = cv.nbr_mask[row];
      const long long off_l = v.nadj_ptr[row];
      const unsigned fP = writes ? fP_l : 0u, mask = writes ? mask_l : 0u;
      const long long off = writes ? off_l : 0;
      if (!cell_ok)
        {
#pragma unroll
          for (int f = 0; f < 5; ++f)
            F[f][0] = F[f][1] = F[f][2] = F[f][3] = 0.0;
        }
      double lam = P.lam, mu = P.mu;
      if (cv.cell_lam && cell_ok) // heterogeneous material, cracks.cc:2207-2216
        {
          lam = cv.cell_lam[i + (long long)(cv.NX - 1) * j];
          mu = cv.cell_mu[i + (long long)(cv.NX - 1) * j];
        }
      const double mu2 = 2.0 * mu;

      // ---- moments (names: field, then the 1-D functions along x / y: 1, n_a, m_g = (n_0 n_0, n_0 n_1, n_1 n_1))
      double Mxx[3], Mxy[2][2], Myy[3];                 // g w JxW:   [1][m], [n][n], [m][1]
      double Px0[2][3], Px1[2][3], Py0[3][2], Py1[3][2]; // T_d,x: [n][m];  T_d,y: [m][n]
      double Q[3][3];                                   // C: [m][m]
      double RZx0[2], RZx1[2], RZy0[2], RZy1[2];        // Z_c,x: [1][n];  Z_c,y: [n][1]
      double RS[2][2], RHx[2], RHy[2];                  // S: [n][n];  H_x: [1][n];  H_y: [n][1]
#pragma unroll
      for (int a = 0; a < 3; ++a)
        {
          Mxx[a] = Myy[a] = 0.0;
          Q[a][0] = Q[a][1] = Q[a][2] = 0.0;
          Py0[a][0] = Py0[a][1] = Py1[a][0] = Py1[a][1] = 0.0;
        }
#pragma unroll
      for (int a = 0; a < 2; ++a)
        {
          Mxy[a][0] = Mxy[a][1] = 0.0;
          Px0[a][0] = Px0[a][1] = Px0[a][2] = Px1[a][0] = Px1[a][1] = Px1[a][2] = 0.0;
          RZx0[a] = RZx1[a] = RZy0[a] = RZy1[a] = RS[a][0] = RS[a][1] = RHx[a] = RHy[a] = 0.0;
        }
      if (cell_ok)
        {
          double Dy0[3], dDy[3]; // d/dy of u_x u_y phi at x-vertex 0 and its x-difference: constant in the cell
#pragma unroll
          for (int f = 0; f < 3; ++f)
            {
              Dy0[f] = (F[f][2] - F[f][0]) * P.ihy;
              dDy[f] = (F[f][3] - F[f][1]) * P.ihy - Dy0[f];
            }
#pragma unroll 1
          for (int qy = 0; qy < 3; ++qy)
            {
              const double ny0 = c_g1.n[0][qy], ny1 = c_g1.n[1][qy];
              const double wy = P.vol * c_g1.w[qy];
              double L0[5], dL[5];
#pragma unroll
              for (int f = 0; f < 5; ++f)
                {
                  L0[f] = ny0 * F[f][0] + ny1 * F[f][2];
                  dL[f] = (ny0 * F[f][1] + ny1 * F[f][3]) - L0[f];
                }
              const double g00 = dL[0] * P.ihx, g10 = dL[1] * P.ihx, gpx = dL[2] * P.ihx; // d/dx: constant along the line
              double Xg1 = 0.0, Xgn[2] = {0.0, 0.0}, Xgm[3] = {0.0, 0.0, 0.0};
              double XT00n[2] = {0.0, 0.0}, XT01n[2] = {0.0, 0.0}, XT01m[3] = {0.0, 0.0, 0.0}, XT11m[3] = {0.0, 0.0, 0.0};
              double XCm[3] = {0.0, 0.0, 0.0};
              double XZ00 = 0.0, XZ01 = 0.0, XZ01n[2] = {0.0, 0.0}, XZ11n[2] = {0.0, 0.0};
              double XSn[2] = {0.0, 0.0}, XHx = 0.0, XHyn[2] = {0.0, 0.0};
#pragma unroll
              for (int qx = 0; qx < 3; ++qx)
                {
                  const double nx0 = c_g1.n[0][qx], nx1 = c_g1.n[1][qx];
                  const double m0 = c_g1.m[0][qx], m1 = c_g1.m[1][qx], m2 = c_g1.m[2][qx];
                  const double JxW = wy * c_g1.w[qx];
                  // Newton state at q (cracks.cc:2222-2232)
                  const double g01 = fma(nx1, dDy[0], Dy0[0]), g11 = fma(nx1, dDy[1], Dy0[1]), gpy = fma(nx1, dDy[2], Dy0[2]);
                  double pf = fma(nx1, dL[2], L0[2]), pfo = fma(nx1, dL[3], L0[3]), pfoo = fma(nx1, dL[4], L0[4]);
                  // q-point state, cracks.cc:2248-2306
                  if (P.monolithic)
                    {
                      pf = fmax(0.0, pf);
                      pfo = fmax(0.0, pfo);
                      pfoo = fmax(0.0, pfoo);
                    }
                  const double pen_plus = fmax(0.0, pf - pfo);
                  const bool pen_on = !((pf - pfo) < 0.0); // shadowed variable, cracks.cc:2311-2315
                  double pfx = pfoo + P.tfac * (pfo - pfoo);
                  if (pfx <= 0.0)
                    pfx = 0.0;
                  if (pfx >= 1.0)
                    pfx = 1.0;
                  if (P.use_old)
                    pfx = pfo;
                  const double pf2 = pfx * pfx;
                  const double g = fma(K.omk, pf2, P.kappa);
                  const double trE = g00 + g11, t01 = g01 + g10;
                  const double lt = lam * trE;
                  const double s00 = fma(mu2, g00, lt), s11 = fma(mu2, g11, lt), s01 = mu * t01; // sigma+ (no split: cracks.cc:2299-2305)
                  const double spE = fma(s00, g00, fma(s11, g11, s01 * t01));
                  // the integrands (module header): Jacobian
                  const double gw = g * JxW;
                  const double pj = pf * JxW;
                  const double T00 = pj * fma(K.omk2, s00, -K.aB1p2), T11 = pj * fma(K.omk2, s11, -K.aB1p2), T01 = pj * (K.omk2 * s01);
                  const double C = JxW * (fma(K.omk, spE, K.gc_eps) - K.aB1p2 * trE + (pen_on ? P.penal_fac : 0.0));
                  // residual
                  const double pd = K.aB1p * pf2;
                  const double Z00 = JxW * fma(g, s00, -pd), Z11 = JxW * fma(g, s11, -pd), Z01 = gw * s01;
                  const double Sq = JxW * (P.penal_fac * pen_plus + fma(pf, fma(-K.aB1p2, trE, fma(K.omk, spE, K.gc_eps)), -K.gc_eps));
                  const double Hx = (K.gce * JxW) * gpx, Hy = (K.gce * JxW) * gpy;
                  // x-sums
                  Xg1 += gw;
                  Xgn[0] = fma(gw, nx0, Xgn[0]), Xgn[1] = fma(gw, nx1, Xgn[1]);
                  Xgm[0] = fma(gw, m0, Xgm[0]), Xgm[1] = fma(gw, m1, Xgm[1]), Xgm[2] = fma(gw, m2, Xgm[2]);
                  XT00n[0] = fma(T00, nx0, XT00n[0]), XT00n[1] = fma(T00, nx1, XT00n[1]);
                  XT01n[0] = fma(T01, nx0, XT01n[0]), XT01n[1] = fma(T01, nx1, XT01n[1]);
                  XT01m[0] = fma(T01, m0, XT01m[0]), XT01m[1] = fma(T01, m1, XT01m[1]), XT01m[2] = fma(T01, m2, XT01m[2]);
                  XT11m[0] = fma(T11, m0, XT11m[0]), XT11m[1] = fma(T11, m1, XT11m[1]), XT11m[2] = fma(T11, m2, XT11m[2]);
                  XCm[0] = fma(C, m0, XCm[0]), XCm[1] = fma(C, m1, XCm[1]), XCm[2] = fma(C, m2, XCm[2]);
                  XZ00 += Z00;
                  XZ01 += Z01;
                  XZ01n[0] = fma(Z01, nx0, XZ01n[0]), XZ01n[1] = fma(Z01, nx1, XZ01n[1]);
                  XZ11n[0] = fma(Z11, nx0, XZ11n[0]), XZ11n[1] = fma(Z11, nx1, XZ11n[1]);
                  XSn[0] = fma(Sq, nx0, XSn[0]), XSn[1] = fma(Sq, nx1, XSn[1]);
                  XHx += Hx;
                  XHyn[0] = fma(Hy, nx0, XHyn[0]), XHyn[1] = fma(Hy, nx1, XHyn[1]);
                }
              // y-factors of the line
              const double my[3] = {c_g1.m[0][qy], c_g1.m[1][qy], c_g1.m[2][qy]}, ny[2] = {ny0, ny1};
#pragma unroll
              for (int g = 0; g < 3; ++g)
                {
                  Mxx[g] = fma(Xg1, my[g], Mxx[g]);
                  Myy[g] += Xgm[g];
#pragma unroll
                  for (int a = 0; a < 2; ++a)
                    {
                      Px0[a][g] = fma(XT00n[a], my[g], Px0[a][g]);
                      Px1[a][g] = fma(XT01n[a], my[g], Px1[a][g]);
                      Py0[g][a] = fma(XT01m[g], ny[a], Py0[g][a]);
                      Py1[g][a] = fma(XT11m[g], ny[a], Py1[g][a]);
                    }
#pragma unroll
                  for (int h = 0; h < 3; ++h)
                    Q[g][h] = fma(XCm[g], my[h], Q[g][h]);
                }
#pragma unroll
              for (int a = 0; a < 2; ++a)
                {
                  Mxy[a][0] = fma(Xgn[0], ny[a], Mxy[a][0]); // [A_y][b_x]
                  Mxy[a][1] = fma(Xgn[1], ny[a], Mxy[a][1]);
                  RZx0[a] = fma(XZ00, ny[a], RZx0[a]);
                  RZx1[a] = fma(XZ01, ny[a], RZx1[a]);
                  RZy0[a] += XZ01n[a];
                  RZy1[a] += XZ11n[a];
                  RS[0][a] = fma(XSn[0], ny[a], RS[0][a]); // [A_x][A_y]
                  RS[1][a] = fma(XSn[1], ny[a], RS[1][a]);
                  RHx[a] = fma(XHx, ny[a], RHx[a]);
                  RHy[a] += XHyn[a];
                }
            }
        }

      // ---- entries of the element matrix from the moments
      const double ihx2 = P.ihx * P.ihx, ihy2 = P.ihy * P.ihy, ihxy = P.ihx * P.ihy;
      const double kxx_d = (lam + mu2) * ihx2, kxx_o = mu * ihx2, kyy_d = (lam + mu2) * ihy2, kyy_o = mu * ihy2;
      const double kl = lam * ihxy, km = mu * ihxy;
      auto uu_entry = [&](auto Aa, auto Bb, auto Cc, auto Dd) __attribute__((always_inline)) -> double {
        constexpr int A = decltype(Aa)::value, b = decltype(Bb)::value, c = decltype(Cc)::value, d = decltype(Dd)::value;
        constexpr int Ax = A & 1, Ay = A >> 1, bx = b & 1, by = b >> 1;
        constexpr double sxx = ((Ax == bx) ? 1.0 : -1.0), syy = ((Ay == by) ? 1.0 : -1.0);
        if constexpr (c == 0 && d == 0)
          return fma(sxx * kxx_d, Mxx[Ay + by], (syy * kyy_o) * Myy[Ax + bx]);
        else if constexpr (c == 1 && d == 1)
          return fma(syy * kyy_d, Myy[Ax + bx], (sxx * kxx_o) * Mxx[Ay + by]);
        else
          {
            // G_xy = s(A_x) s(b_y) M_xy[A_y][b_x],  G_yx = s(A_y) s(b_x) M_xy[b_y][A_x]
            constexpr double sxy = ((Ax == 1) == (by == 1)) ? 1.0 : -1.0, syx = ((Ay == 1) == (bx == 1)) ? 1.0 : -1.0;
            if constexpr (c == 0) // lambda G_xy + mu G_yx
              return fma(sxy * kl, Mxy[Ay][bx], (syx * km) * Mxy[by][Ax]);
            else // lambda G_yx + mu G_xy
              return fma(syx * kl, Mxy[by][Ax], (sxy * km) * Mxy[Ay][bx]);
          }
      };
      auto pu_entry = [&](auto Aa, auto Bb, auto Dd) __attribute__((always_inline)) -> double {
        constexpr int A = decltype(Aa)::value, b = decltype(Bb)::value, d = decltype(Dd)::value;
        constexpr int Ax = A & 1, Ay = A >> 1, bx = b & 1, by = b >> 1;
        constexpr double sx = bx ? 1.0 : -1.0, sy = by ? 1.0 : -1.0;
        if constexpr (d == 0)
          return fma(sx * P.ihx, Px0[Ax][Ay + by], (sy * P.ihy) * Py0[Ax + bx][Ay]);
        else
          return fma(sx * P.ihx, Px1[Ax][Ay + by], (sy * P.ihy) * Py1[Ax + bx][Ay]);
      };
      auto pp_entry = [&](auto Aa, auto Bb) __attribute__((always_inline)) -> double {
        constexpr int A = decltype(Aa)::value, b = decltype(Bb)::value;
        constexpr int Ax = A & 1, Ay = A >> 1, bx = b & 1, by = b >> 1;
        constexpr double sxx = ((Ax == bx) ? 1.0 : -1.0), syy = ((Ay == by) ? 1.0 : -1.0);
        // G_c eps sum_q JxW grad N_A . grad N_b: the same for every cell of the box
        const double lap = (sxx * K.gc_eps_vol_x) * c_g1.mb[Ay + by] + (syy * K.gc_eps_vol_y) * c_g1.mb[Ax + bx];
        return cell_ok ? Q[Ax + bx][Ay + by] + lap : 0.0;
      };
      auto res_entry = [&](auto Aa, auto Cc) __attribute__((always_inline)) -> double {
        constexpr int A = decltype(Aa)::value, c = decltype(Cc)::value;
        constexpr int Ax = A & 1, Ay = A >> 1;
        constexpr double sx = Ax ? 1.0 : -1.0, sy = Ay ? 1.0 : -1.0;
        if constexpr (c == 0)
          return -fma(sx * P.ihx, RZx0[Ay], (sy * P.ihy) * RZy0[Ax]);
        else if constexpr (c == 1)
          return -fma(sx * P.ihx, RZx1[Ay], (sy * P.ihy) * RZy1[Ax]);
        else
          return -(RS[Ax][Ay] + fma(sx * P.ihx, RHx[Ay], (sy * P.ihy) * RHy[Ax]));
      };
      using std::integral_constant;
      // mean |diagonal| of the element matrix: deal.II's placeholder when a constrained row's own entry vanishes.  Needed
      // in blocks with constrained rows only (!plain, below).  Two launches: the first forms it (it has the registers for the
      // moments of both diagonal blocks) and leaves it in CartView::cell_avg for the second, which then needs none of the
      // (u,u) moments: 256 registers + 14 spilled -> 245, none spilled.
      double avg = 0.0;
      if constexpr (GROUP != 1)
        {
          static_for<4>([&](auto Aa) __attribute__((always_inline)) {
            avg += fabs(uu_entry(Aa, Aa, integral_constant<int, 0>{}, integral_constant<int, 0>{})) +
                   fabs(uu_entry(Aa, Aa, integral_constant<int, 1>{}, integral_constant<int, 1>{})) + fabs(pp_entry(Aa, Aa));
          });
          avg *= 1.0 / 12.0;
        }

      const int deg = __popc(mask & 0x1ffu);
      const bool blocked = v.layout == PFM_LAYOUT_BLOCKED;
      // Blocked layout: the rows of the 7 nodes of a block row are one contiguous piece of each of the four value arrays
      // when the nodes are consecutive CSR rows -- they are staged in LDS and stored as whole cache lines (a lane
      // storing its own 81 values touches 49 lines per store instruction, and the L2 sees 8 partial writes per line:
      // the first version of this kernel spent 81 % of its wave cycles waiting on that).
      __shared__ double s_rows[O2][4 * 9 * O2 + 5]; // 257: block rows start 2 banks apart
      const int lane_first = cy * B2 + 1;                              // first owner lane of this block row
      const int n_row = min(O2, cv.o1[0] - (cv.o0[0] + tix * O2) + 1); // owned nodes in a block row
      const long long off_first = __shfl(off, lane_first);
      const long long off_next = __shfl(off, lane + 1);
      const bool chain_ok = !owner || (writes && (cx == n_row || off_next == off + deg)); // consecutive CSR rows
      const bool staged = blocked && __all(chain_ok);
      // regular interior blocks (all 9 neighbours, no constraint anywhere in the stencils): slots and masks are compile-time
      const bool plain = staged && __all(!writes || (mask == 0x1ffu && fP == 0u && fN == 0u));
      if constexpr (GROUP != 2)
        {
          if (!plain && cell_ok) // (wave-uniform: both launches see the same masks and flags)
            {
              double *const pa = cv.cell_avg + (i + (long long)(cv.NX - 1) * j);
              if constexpr (GROUP == 0)
                *pa = avg;
              else
                avg = *pa;
            }
        }
      const int rel = (int)(off - off_first); // position of this node's row in the block row, in entries
      const long long row_len = __shfl(off + deg, cy * B2 + n_row) - off_first; // entries of the block row
      const bool row_live = cy >= 1 && (cv.o0[1] + tiy * O2 + cy - 1) <= cv.o1[1];
      // stores `len` staged doubles of every live block row to dst + scale * off_first
      auto flush = [&](double *dst, int scale, int base) __attribute__((always_inline)) {
        lds_barrier();
#pragma unroll 1
        for (int r = 0; r < O2; ++r)
          {
            const long long o_f = __shfl(off_first, (r + 1) * B2 + 1);
            const int len = scale * (int)__shfl(row_len, (r + 1) * B2 + 1);
            const bool live = __shfl((int)row_live, (r + 1) * B2 + 1) != 0;
            if (!live)
              continue;
            // all reads of the row, then its stores (one read -> wait -> store per turn was a chain of exposed LDS latencies:
            // 20 k of a wave's 136 k cycles per flush in the phase clock)
            double x[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
              x[qq] = s_rows[r][min(base + lane + 64 * qq, 4 * 9 * O2 + 4)];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
              if (lane + 64 * qq < len)
                dst[scale * o_f + lane + 64 * qq] = x[qq];
          }
        lds_barrier();
      };
      // what the lane of vertex A's node pulls from this cell: lane - 9, - 8, - 1 for A = 3, 2, 1 (own cell: A = 0)
      auto pull = [&](auto Aa, double x) __attribute__((always_inline)) -> double {
        constexpr int A = decltype(Aa)::value;
        if constexpr (A == 0)
          return x;
        else
          return __shfl(x, lane - ((A & 1) + B2 * (A >> 1)));
      };
      auto slot_of = [&](int o) __attribute__((always_inline)) -> int {
        int sl = __popc(mask & ((1u << o) - 1u));
        if (mask >> 31)
          sl = cv.row_perm[off + sl];
        return sl;
      };

      // ---- rows c = 0, 1: (u,u) block, the (u,phi) block is structurally zero (cracks.cc:2333-2337)
      if constexpr (GROUP != 1)
      {
        double acc[9][2][2], R[2] = {0.0, 0.0}, dg[2] = {0.0, 0.0};
#pragma unroll
        for (int o = 0; o < 9; ++o)
          acc[o][0][0] = acc[o][0][1] = acc[o][1][0] = acc[o][1][1] = 0.0;
        static_for<4>([&](auto Ee) __attribute__((always_inline)) {
          constexpr int A = 3 - decltype(Ee)::value, ax = A & 1, ay = A >> 1;
          using IA = integral_constant<int, A>;
          const double avg_e = pull(IA{}, avg);
          static_for<4>([&](auto Bb) __attribute__((always_inline)) {
            constexpr int b = decltype(Bb)::value, o = ((b & 1) - ax + 1) + 3 * ((b >> 1) - ay + 1);
            static_for<2>([&](auto Cc) __attribute__((always_inline)) {
              static_for<2>([&](auto Dd) __attribute__((always_inline)) {
                constexpr int c = decltype(Cc)::value, d = decltype(Dd)::value;
                const double x = pull(IA{}, uu_entry(IA{}, Bb, Cc, Dd));
                acc[o][c][d] += x;
                if constexpr (b == A && c == d)
                  {
                    const double k = fabs(x);
                    dg[c] += k != 0.0 ? k : avg_e;
                  }
              });
            });
          });
          R[0] += pull(IA{}, res_entry(IA{}, integral_constant<int, 0>{}));
          R[1] += pull(IA{}, res_entry(IA{}, integral_constant<int, 1>{}));
        });
        if (plain)
          {
            if (writes)
              {
                res_pde[(long long)row * 2] = R[0];
                res_pde[(long long)row * 2 + 1] = R[1];
                if (write_total)
                  {
                    res_tot[(long long)row * 2] = R[0];
                    res_tot[(long long)row * 2 + 1] = R[1];
                  }
                double *dl = &s_rows[cy - 1][4 * rel];
#pragma unroll
                for (int o = 0; o < 9; ++o)
#pragma unroll
                  for (int c = 0; c < 2; ++c)
                    {
                      dl[c * 18 + o * 2] = acc[o][c][0];
                      dl[c * 18 + o * 2 + 1] = acc[o][c][1];
                    }
              }
          }
        else if (writes)
          {
#pragma unroll
            for (int c = 0; c < 2; ++c)
              {
                const bool con = (fP >> c) & 1u;
                const long long di = blocked ? (long long)row * 2 + c : (long long)row * 3 + c;
                res_pde[di] = con ? 0.0 : R[c]; // constrained scatter as masks (cracks.cc:2439-2464)
                if (write_total)
                  res_tot[di] = (con && total_via_update) ? 0.0 : R[c];
              }
#pragma unroll
            for (int o = 0; o < 9; ++o)
              {
                if (!((mask >> o) & 1u))
                  continue;
                const int sl = slot_of(o);
                const unsigned fQ = (fN >> (3 * o)) & 7u;
#pragma unroll
                for (int c = 0; c < 2; ++c)
                  {
                    const bool rcon = (fP >> c) & 1u;
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                      {
                        double x = d < 2 ? acc[o][c][d < 2 ? d : 0] : 0.0;
                        if (rcon)
                          x = (o == 4 && c == d) ? dg[c] : 0.0;
                        else if ((fQ >> d) & 1u)
                          x = 0.0;
                        if (staged)
                          {
                            if (d < 2)
                              s_rows[cy - 1][4 * rel + c * 2 * deg + sl * 2 + d] = x;
                            continue; // the (u,phi) block is zero-filled below
                          }
                        double *dst;
                        if (!blocked)
                          dst = vals.b[0] + (9 * off + (long long)c * 3 * deg + (long long)sl * 3 + d);
                        else
                          dst = d < 2 ? vals.b[0] + (4 * off + (long long)c * 2 * deg + (long long)sl * 2 + d)
                                      : vals.b[1] + (2 * off + (long long)c * deg + sl);
                        *dst = x;
                      }
                  }
              }
          }
        if (staged)
          {
            flush(vals.b[0], 4, 0);
            // (u,phi): structurally zero (cracks.cc:2333-2337).  CartView::up_block_cleared: a fill in front of the launch
            // has written the block (18 of the 81 doubles of a node, at the rate of a plain stream of stores)
#pragma unroll 1
            for (int r = 0; r < (cv.up_block_cleared ? 0 : O2); ++r)
              {
                const long long o_f = __shfl(off_first, (r + 1) * B2 + 1);
                const int len = 2 * (int)__shfl(row_len, (r + 1) * B2 + 1);
                if (__shfl((int)row_live, (r + 1) * B2 + 1) == 0)
                  continue;
                for (int k = lane; k < len; k += 64)
                  vals.b[1][2 * o_f + k] = 0.0;
              }
          }
      }
      // ---- row c = 2: (phi,u) and (phi,phi) blocks
      if constexpr (GROUP != 0)
      {
        double apu[9][2], app[9], R2 = 0.0, dg2 = 0.0;
#pragma unroll
        for (int o = 0; o < 9; ++o)
          apu[o][0] = apu[o][1] = app[o] = 0.0;
        static_for<4>([&](auto Ee) __attribute__((always_inline)) {
          constexpr int A = 3 - decltype(Ee)::value, ax = A & 1, ay = A >> 1;
          using IA = integral_constant<int, A>;
          const double avg_e = pull(IA{}, avg);
          static_for<4>([&](auto Bb) __attribute__((always_inline)) {
            constexpr int b = decltype(Bb)::value, o = ((b & 1) - ax + 1) + 3 * ((b >> 1) - ay + 1);
            apu[o][0] += pull(IA{}, pu_entry(IA{}, Bb, integral_constant<int, 0>{}));
            apu[o][1] += pull(IA{}, pu_entry(IA{}, Bb, integral_constant<int, 1>{}));
            const double x = pull(IA{}, pp_entry(IA{}, Bb));
            app[o] += x;
            if constexpr (b == A)
              {
                const double k = fabs(x);
                dg2 += k != 0.0 ? k : avg_e;
              }
          });
          R2 += pull(IA{}, res_entry(IA{}, integral_constant<int, 2>{}));
        });
        if (plain)
          {
            if (writes)
              {
                const long long di = (long long)v.n_owned * 2 + row;
                res_pde[di] = R2;
                if (write_total)
                  res_tot[di] = R2;
                double *dl = &s_rows[cy - 1][2 * rel], *dp = &s_rows[cy - 1][2 * 9 * O2 + rel];
#pragma unroll
                for (int o = 0; o < 9; ++o)
                  {
                    dl[o * 2] = apu[o][0];
                    dl[o * 2 + 1] = apu[o][1];
                    dp[o] = app[o];
                  }
              }
          }
        else if (writes)
          {
            const bool rcon = (fP >> 2) & 1u;
            const long long di = blocked ? (long long)v.n_owned * 2 + row : (long long)row * 3 + 2;
            res_pde[di] = rcon ? 0.0 : R2;
            if (write_total)
              res_tot[di] = (rcon && total_via_update) ? 0.0 : R2;
#pragma unroll
            for (int o = 0; o < 9; ++o)
              {
                if (!((mask >> o) & 1u))
                  continue;
                const int sl = slot_of(o);
                const unsigned fQ = (fN >> (3 * o)) & 7u;
#pragma unroll
                for (int d = 0; d < 3; ++d)
                  {
                    double x = d < 2 ? apu[o][d < 2 ? d : 0] : app[o];
                    if (rcon)
                      x = (o == 4 && d == 2) ? dg2 : 0.0;
                    else if ((fQ >> d) & 1u)
                      x = 0.0;
                    if (staged)
                      {
                        if (d < 2)
                          s_rows[cy - 1][2 * rel + sl * 2 + d] = x;
                        else
                          s_rows[cy - 1][2 * 9 * O2 + rel + sl] = x;
                        continue;
                      }
                    double *dst;
                    if (!blocked)
                      dst = vals.b[0] + (9 * off + (long long)2 * 3 * deg + (long long)sl * 3 + d);
                    else
                      dst = d < 2 ? vals.b[2] + (2 * off + (long long)sl * 2 + d) : vals.b[3] + (off + sl);
                    *dst = x;
                  }
              }
          }
        if (staged)
          {
            lds_barrier();
#pragma unroll 1
            for (int r = 0; r < O2; ++r)
              {
                const long long o_f = __shfl(off_first, (r + 1) * B2 + 1);
                const int len = (int)__shfl(row_len, (r + 1) * B2 + 1);
                if (__shfl((int)row_live, (r + 1) * B2 + 1) == 0)
                  continue;
                const double x0 = s_rows[r][lane], x1 = s_rows[r][lane + 64], x2 = s_rows[r][2 * 9 * O2 + lane];
                if (lane < 2 * len)
                  vals.b[2][2 * o_f + lane] = x0;
                if (lane + 64 < 2 * len)
                  vals.b[2][2 * o_f + lane + 64] = x1;
                if (lane < len)
                  vals.b[3][o_f + lane] = x2;
              }
          }
      }
    }
  } // namespace

  // 2-D cartesian boxes: Jacobian + residual without the stress split (the plain 2-D residual has its own kernel in
  // pfm_cart.hip)
  int launch_cart2d(const DevView &v, const CartView &cv, const pfm_params &p, int residual_only, double *const *d_values,
                    double *res_pde, double *res_tot, hipStream_t s, hipStream_t s_phi)
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
    if (split)
      return PFM_ERR_UNSUPPORTED; // stress-split runs are not a moment of a q-point field: they take the general family with the cartesian overlay (pfm_kernels.hip, PATCH; kernel path 3)
    if (residual_only)
      return PFM_ERR_UNSUPPORTED; // residual-only 2-D assemblies: k_cart_residual2m (pfm_cart.hip)
    if (cv.o1[0] < cv.o0[0] || cv.o1[1] < cv.o0[1])
      return PFM_OK;
    Cst2 K{};
    K.omk = 1.0 - P.kappa;
    K.omk2 = 2.0 * (1.0 - P.kappa);
    K.aB1p = P.aB1 * P.p;
    K.aB1p2 = 2.0 * P.aB1 * P.p;
    K.gc_eps = P.Gc / P.eps;
    K.gce = P.Gc * P.eps;
    K.gc_eps_vol_x = P.Gc * P.eps * P.vol * P.ihx * P.ihx;
    K.gc_eps_vol_y = P.Gc * P.eps * P.vol * P.ihy * P.ihy;
    const long long ntx = (cv.o1[0] - cv.o0[0] + 1 + O2 - 1) / O2, nty = (cv.o1[1] - cv.o0[1] + 1 + O2 - 1) / O2;
    const bool one_launch = getenv("PFM_CART2D_ONE_LAUNCH") != nullptr; // A/B runs, tests
    if (one_launch)
      hipLaunchKernelGGL(k_cart2d_cells<2>, dim3(xcd_grid((unsigned)(ntx * nty))), dim3(64), 0, s, v, cv, P, K, vals, res_pde, res_tot, residual_only,
                         total_via_update, (int)nty);
    else
      {
        hipLaunchKernelGGL(k_cart2d_cells<0>, dim3(xcd_grid((unsigned)(ntx * nty))), dim3(64), 0, s, v, cv, P, K, vals, res_pde, res_tot, residual_only,
                           total_via_update, (int)nty);
        hipLaunchKernelGGL(k_cart2d_cells<1>, dim3(xcd_grid((unsigned)(ntx * nty))), dim3(64), 0, s_phi, v, cv, P, K, vals, res_pde, res_tot, residual_only,
                           total_via_update, (int)nty);
      }
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }
} // namespace pfm
