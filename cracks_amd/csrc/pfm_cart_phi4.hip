// pfm_cart_phi4.hip — phase-field rows of the Jacobian (3-D), z-marching "push" kernel:
// the (phi,u) block (cracks.cc:2374-2376, 2381-2382 with a displacement trial function), the (phi,phi)
// block (cracks.cc:2370-2371, 2377-2383) and the placeholder diagonals of constrained rows
// (deal.II distribute_local_to_global).
//
// For these rows a finished CSR row (108 doubles per node) is smaller than the sum-factorised cell
// tables it is built from (191 doubles per cell), so the tables never leave the registers:
//
//   * a workgroup owns a 7 x 7 column of nodes over a chunk of z-planes; its 4 waves are 4 ROLES
//     (column component d = 0,1,2 of the (phi,u) block, and the (phi,phi) block), lane <-> one of the
//     8 x 8 cells of the current layer touching those nodes;
//   * marching up in z, every lane evaluates its cell ONCE per role (quadrature with all three 1-D
//     contractions factored), then pushes the cell's entries into the LDS-staged rows of its 8 vertices:
//     the 4 lower vertices complete the rows of node plane k, the 4 upper ones start the rows of plane k+1.
//     Pushes of one role touch disjoint entries and are issued vertex by vertex in a fixed order, so there
//     are no atomics, no barriers inside the push, and the summation order is that of a lexicographic
//     cell loop (bitwise reproducible);
//   * plane k is then masked (constraints) and streamed out: every CSR value is written exactly once,
//     x-consecutive rows are contiguous in memory.
//
// Staged rows are split by the z-offset of the neighbour slot (oz = -1, 0, +1): only the oz <= 0 parts of
// the next plane are live across steps, which keeps LDS at 80 KB (2 workgroups per CU).
//
// (phi,u) entry of test vertex a and trial dof (b,d):  K = sum_k s(b_k)/h_k C^{dk}[a_k][g_i][g_j],
//   C^{dk}[al][g_i][g_j] = sum_q Phi^{dk}(q) n_al(q_k) m_{g_i}(q_i) m_{g_j}(q_j),   g = a + b per axis,
//   Phi^{dk} = w pf [ (2(1-kappa) lambda trE - 2(alpha_B-1) p) delta_dk + 4(1-kappa) mu E_dk ]
// (phi,phi) entry: M[g_x][g_y][g_z] = sum_q w c(q) m m m + G_c eps (Laplace moments),
//   c(q) = (1-kappa) sigma+:E + G_c/eps - 2(alpha_B-1) p div u + gamma/dt/diam^2 [pf >= pf_old].
#include "pfm_internal.h"
#include "pfm_cart_common.h"
#include "pfm_poly.h"
#include "pfm_dma.h"

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

namespace pfm
{
  namespace
  {
    constexpr int PT = 8, PN = PT - 1, PH = PT + 1; // cells, owned nodes, halo nodes per tile edge
    constexpr int NPN = PN * PN, NPH = PH * PH;     // 49 owned nodes, 81 halo nodes per plane
    constexpr int NT4 = 4 * PT * PT;                // 4 roles x 64 cells
    constexpr int SLAB_PU = NPN * 27, SLAB_PP = NPN * 9;
#ifndef PFM_PHI_ONEPUSH
#define PFM_PHI_ONEPUSH 1
#endif
    constexpr bool ONEPUSH = PFM_PHI_ONEPUSH != 0; // (phi,u) roles: one LDS add per entry and cell (pu_role_poly)


    template <int NF /* 6 with the old phase fields (penalisation term: gamma != 0), else 4 */>
    struct Lds4
    {
      // destinations of the global -> LDS transfers first: M0 carries a 16-bit LDS offset
      double U[2][NF][NPH];  // nodal ring: u_x u_y u_z phi [phi_old phi_oldold]
      long long off[2][NPN]; // node-graph offset of the row, -1 = not an owned node of this tile
      int deg[2][NPN];       // neighbour mask of the row (bit o: lattice offset o exists)
      unsigned flag[4][2 * NPH]; // constraint flag byte of the halo nodes of 4 planes, one dword per requesting lane: node hn at [2 hn]
      int irregular[2];
      int incomplete[2];     // by plane parity: some row of the tile is not a row of this launch or has fewer than 27 neighbours
      int anyflag[4];
      double pu[5][SLAB_PU]; // staged (phi,u) rows: [0,1] oz=-1 ring, [2,3] oz=0 ring, [4] oz=+1; [node][o9][d]
      double pp[5][SLAB_PP]; // staged (phi,phi) rows, same slabs; [node][o9]
      double ex[2][NPN][2];  // per node: placeholder sum, (u,u) placeholder patch
      double rs[2][NPN];     // per node (RES): K_phiphi phi, pushed cell by cell (round 6)
    };


    // (global -> LDS transfers: pfm_dma.h)
    // The 8 vertex values of one nodal field of a cell are read from the nodal ring ONCE per cell and role (the
    // compiler cannot keep LDS values across the LDS-side adds of the pushes by itself), as the 4 values of the lower
    // z-face a[0..3] (index x + 2 y) and the z-derivative (upper - lower) / h_z at the same 4 positions a[4..7].
    // The trilinear interpolation then runs along z first: per z-level 4 FMAs, per (qy,qz) line 4 + 4 operations.
    __device__ __forceinline__ void load_cell_field(const double *__restrict__ lo, const double *__restrict__ hi, double ihz,
                                                    double (&a)[8])
    {
      a[0] = lo[0], a[1] = lo[1], a[2] = lo[PH], a[3] = lo[PH + 1];
      a[4] = (hi[0] - a[0]) * ihz, a[5] = (hi[1] - a[1]) * ihz, a[6] = (hi[PH] - a[2]) * ihz, a[7] = (hi[PH + 1] - a[3]) * ihz;
    }
    // field at the 4 (x,y) vertices of z-level qz: zq = n_1(q_z) h_z
    __device__ __forceinline__ void zlevel_of_field(const double (&a)[8], double zq, double (&Z)[4])
    {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        Z[i] = fma(zq, a[4 + i], a[i]);
    }
    __device__ __forceinline__ void dy_of_level(const double (&Z)[4], double ihy, double (&Dy)[2])
    {
      Dy[0] = (Z[2] - Z[0]) * ihy;
      Dy[1] = (Z[3] - Z[1]) * ihy;
    }
    // values of one nodal field at the (qy,qz) line of a cell: L = value at x-vertex 0/1, Dz = d/dz there
    template <bool DZ>
    __device__ __forceinline__ void line_of_level(const double (&Z)[4], const double (&a)[8], double ny0, double ny1, double (&L)[2],
                                                  double (&Dz)[2])
    {
      L[0] = ny0 * Z[0] + ny1 * Z[2];
      L[1] = ny0 * Z[1] + ny1 * Z[3];
      if constexpr (DZ)
        {
          Dz[0] = ny0 * a[4] + ny1 * a[6];
          Dz[1] = ny0 * a[5] + ny1 * a[7];
        }
    }

    // ---- helpers of the (phi,phi) role: the 8 raw vertex values, index x + 2 y + 4 z
    __device__ __forceinline__ void load_cell_field_raw(const double *__restrict__ lo, const double *__restrict__ hi, double (&a)[8])
    {
      a[0] = lo[0], a[1] = lo[1], a[2] = lo[PH], a[3] = lo[PH + 1];
      a[4] = hi[0], a[5] = hi[1], a[6] = hi[PH], a[7] = hi[PH + 1];
    }
    __device__ __forceinline__ void dy_of_field(const double (&a)[8], double nz0, double nz1, double ihy, double (&Dy)[2])
    {
      Dy[0] = (nz0 * (a[2] - a[0]) + nz1 * (a[6] - a[4])) * ihy;
      Dy[1] = (nz0 * (a[3] - a[1]) + nz1 * (a[7] - a[5])) * ihy;
    }

    // (phi,phi) role: raw vertex values, y-z interpolation per line (fewer live registers next to the 27 moments)
    template <bool DY, bool DZ>
    __device__ __forceinline__ void line_of_field(const double (&a)[8], double ny0,
                                                  double ny1, double nz0, double nz1, double ihy, double ihz, double (&L)[2],
                                                  double (&Dy)[2], double (&Dz)[2])
    {
      const double a00 = a[0], a10 = a[1], a01 = a[2], a11 = a[3];
      const double b00 = a[4], b10 = a[5], b01 = a[6], b11 = a[7];
      const double l0 = ny0 * a00 + ny1 * a01, l1 = ny0 * a10 + ny1 * a11;
      const double h0 = ny0 * b00 + ny1 * b01, h1 = ny0 * b10 + ny1 * b11;
      L[0] = nz0 * l0 + nz1 * h0;
      L[1] = nz0 * l1 + nz1 * h1;
      if constexpr (DZ)
        {
          Dz[0] = (h0 - l0) * ihz;
          Dz[1] = (h1 - l1) * ihz;
        }
      if constexpr (DY)
        {
          Dy[0] = (nz0 * (a01 - a00) + nz1 * (b01 - b00)) * ihy;
          Dy[1] = (nz0 * (a11 - a10) + nz1 * (b11 - b10)) * ihy;
        }
    }

    // (NA ? -a : a) + (NB ? -b : b) as one v_add_f64 that stays where it is written
    template <bool NA, bool NB>
    __device__ __forceinline__ double add_signed(double a, double b)
    {
      double r;
      if constexpr (NA && NB)
        asm volatile("v_add_f64 %0, -%1, -%2" : "=v"(r) : "v"(a), "v"(b));
      else if constexpr (NA)
        asm volatile("v_add_f64 %0, -%1, %2" : "=v"(r) : "v"(a), "v"(b));
      else if constexpr (NB)
        asm volatile("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
      else
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      return r;
    }

    static_assert(sizeof(Lds4<4>) <= 64 * 1280, "two workgroups per CU: 64 allocation granules of LDS each");
    struct PushDst // staged rows the 8 vertices of a cell push into
    {
      double *lo_z0, *lo_p1; // current plane: oz = 0 and oz = +1 slabs (lower vertices)
      double *hi_m1, *hi_z0; // next plane: oz = -1 and oz = 0 slabs (upper vertices)
      bool push_lo, push_hi;
    };

    // ---- role d: (phi,u) entries of column component D of one cell.  The x- and y-contractions are factored
    // (X, Y accumulators); the z-contraction is folded into the push, one push per z-level of q-points, so the
    // 54 numbers C^{dk}[al][g_i][g_j] never have to be held in registers.
    template <int D, bool HET, bool GENQ>
    __device__ __forceinline__ void pu_role(const double *__restrict__ Ulo, const double *__restrict__ Uhi, const MatScal &S,
                                            double cell_muh, double cell_la, bool cell_ok, const PushDst &dst, int nl0, int cx, int cy)
    {
      const double c_muh = HET ? cell_muh : S.c_muh, c_la = HET ? cell_la : S.c_la, cdiag = S.cdiag;
      double V[4][8]; // u_x u_y u_z phi at the cell's vertices
#pragma unroll
      for (int f = 0; f < 4; ++f)
        load_cell_field(Ulo + f * NPH, Uhi + f * NPH, S.ih[2], V[f]);
#pragma unroll 1
      for (int qz = 0; qz < 3; ++qz)
        {
          double Yx[2][3], Yy[3][2], Yz[3][3]; // k=x: [al][g_y], k=y: [g_x][al], k=z: [g_x][g_y]
#pragma unroll
          for (int i = 0; i < 3; ++i)
            {
              Yx[0][i] = Yx[1][i] = Yy[i][0] = Yy[i][1] = 0.0;
              Yz[i][0] = Yz[i][1] = Yz[i][2] = 0.0;
            }
          const double nz0 = c_g1.n[0][qz], nz1 = c_g1.n[1][qz];
          if (cell_ok)
            {
              double Z[4][4], Dy[3][2], dDy[3]; // z-level values; d/dy depends on the z-level only
              const double zq = nz1 * S.hz;
              static_for<4>([&](auto F) __attribute__((always_inline)) {
                constexpr int f = decltype(F)::value;
                zlevel_of_field(V[f], zq, Z[f]);
                if constexpr (f < 3 && ((f == D) || (D == 1) || (f == 1)))
                  {
                    dy_of_level(Z[f], S.ih[1], Dy[f]);
                    dDy[f] = Dy[f][1] - Dy[f][0];
                  }
              });
#pragma unroll 1
              for (int qy = 0; qy < 3; ++qy)
                {
                  const double ny0 = c_g1.n[0][qy], ny1 = c_g1.n[1][qy];
                  const double wyz = S.vww[qy][qz];
                  double L[4][2], Dz[3][2];
                  static_for<3>([&](auto F) __attribute__((always_inline)) {
                    constexpr int f = decltype(F)::value;
                    constexpr bool need_dz = (f == D) || (D == 2) || (f == 2);
                    line_of_level<need_dz>(Z[f], V[f], ny0, ny1, L[f], Dz[f]);
                  });
                  {
                    double dummy[2];
                    line_of_level<false>(Z[3], V[3], ny0, ny1, L[3], dummy);
                  }
                  double Dx[3];
#pragma unroll
                  for (int f = 0; f < 3; ++f)
                    Dx[f] = (L[f][1] - L[f][0]) * S.ih[0];
                  // along x everything is linear: value(q_x) = v0 + n_1(q_x) (v1 - v0), one FMA per q-point
                  const double dpf = L[3][1] - L[3][0];
                  double dDz[3];
#pragma unroll
                  for (int f = 0; f < 3; ++f)
                    dDz[f] = Dz[f][1] - Dz[f][0]; // unused components are dropped by the compiler
                  double Xx[2] = {0.0, 0.0}, Xy[3] = {0.0, 0.0, 0.0}, Xz[3] = {0.0, 0.0, 0.0};
                  if constexpr (!GENQ)
                    {
                      // Round 4: the x-contraction from COEFFICIENTS.  Along the line pf = p0 + t p1 and every requested
                      // gradient a0 + t a1 are linear in t = n_1(q_x), so Phi^{dk} = w pf (A0 + t A1) and
                      //   sum_q Phi W(q) = A0 Q0[W] + A1 Q1[W],  Q0[W] = sum_q w pf W = p0 s0[W] + p1 s1[W],  Q1[W] = p0 s1[W] + p1 s2[W]
                      // with the constants s_j[W] = sum_q w t^j W(q) (G1::sx): the same sums as the q-point loop below (which
                      // stays for the clamped phase field of the monolithic scheme: GENQ), regrouped -- 52 instead of ~100
                      // instructions per line.
                      const double p0 = wyz * L[3][0], p1 = wyz * dpf;
                      double Q0[5], Q1[5];
                      static_for<5>([&](auto Wc) __attribute__((always_inline)) {
                        constexpr int W = decltype(Wc)::value;
                        Q0[W] = p0 * G1Sx<0, W>::v + p1 * G1Sx<1, W>::v;
                        Q1[W] = p0 * G1Sx<1, W>::v + p1 * G1Sx<2, W>::v;
                      });
                      // gradient of u_c along the line: constant part and slope in t (d/dx is constant along x)
                      auto g0 = [&](auto Cc, auto Kk) __attribute__((always_inline)) -> double {
                        constexpr int c = decltype(Cc)::value, k = decltype(Kk)::value;
                        if constexpr (k == 0)
                          return Dx[c];
                        else if constexpr (k == 1)
                          return Dy[c][0];
                        else
                          return Dz[c][0];
                      };
                      auto g1 = [&](auto Cc, auto Kk) __attribute__((always_inline)) -> double {
                        constexpr int c = decltype(Cc)::value, k = decltype(Kk)::value;
                        if constexpr (k == 0)
                          return 0.0;
                        else if constexpr (k == 1)
                          return dDy[c];
                        else
                          return dDz[c];
                      };
                      using I0 = std::integral_constant<int, 0>;
                      using I1 = std::integral_constant<int, 1>;
                      using I2 = std::integral_constant<int, 2>;
                      using ID = std::integral_constant<int, D>;
                      const double d0 = c_la * (Dx[0] + Dy[1][0] + Dz[2][0]) + cdiag, d1 = c_la * (dDy[1] + dDz[2]);
                      double A0[3], A1[3];
                      static_for<3>([&](auto K) __attribute__((always_inline)) {
                        constexpr int k = decltype(K)::value;
                        using IK = std::integral_constant<int, k>;
                        A0[k] = c_muh * (g0(ID{}, IK{}) + g0(IK{}, ID{})) + (k == D ? d0 : 0.0);
                        if constexpr (k == 0 && D == 0)
                          A1[k] = d1;
                        else if constexpr (k == 0)
                          A1[k] = c_muh * g1(IK{}, ID{});
                        else if constexpr (D == 0)
                          A1[k] = c_muh * g1(ID{}, IK{});
                        else
                          A1[k] = c_muh * (g1(ID{}, IK{}) + g1(IK{}, ID{})) + (k == D ? d1 : 0.0);
                      });
                      (void)sizeof(I0), (void)sizeof(I1), (void)sizeof(I2);
                      Xx[0] = A0[0] * Q0[0] + A1[0] * Q1[0];
                      Xx[1] = A0[0] * Q0[1] + A1[0] * Q1[1];
#pragma unroll
                      for (int g = 0; g < 3; ++g)
                        {
                          Xy[g] = A0[1] * Q0[2 + g] + A1[1] * Q1[2 + g];
                          Xz[g] = A0[2] * Q0[2 + g] + A1[2] * Q1[2 + g];
                        }
                    }
                  else
                    {
#pragma unroll
                  for (int qx = 0; qx < 3; ++qx)
                    {
                      const double nx0 = c_g1.n[0][qx], nx1 = c_g1.n[1][qx];
                      double pf = fma(nx1, dpf, L[3][0]);
                      if (S.monolithic)
                        pf = fmax(0.0, pf); // cracks.cc:2251-2256
                      const double wp = (wyz * c_g1.w[qx]) * pf;
                      // d/dx_k u_c at q (only the row D, the column D and the diagonal are ever requested)
                      auto grad = [&](auto Cc, auto Kk) __attribute__((always_inline)) -> double {
                        constexpr int c = decltype(Cc)::value, k = decltype(Kk)::value;
                        if constexpr (k == 0)
                          return Dx[c];
                        else if constexpr (k == 1)
                          return fma(nx1, dDy[c], Dy[c][0]);
                        else
                          return fma(nx1, dDz[c], Dz[c][0]);
                      };
                      using I0 = std::integral_constant<int, 0>;
                      using I1 = std::integral_constant<int, 1>;
                      using I2 = std::integral_constant<int, 2>;
                      using ID = std::integral_constant<int, D>;
                      const double g00 = grad(I0{}, I0{}), g11 = grad(I1{}, I1{}), g22 = grad(I2{}, I2{});
                      const double dterm = c_la * (g00 + g11 + g22) + cdiag;
                      const double gD0 = grad(ID{}, I0{}), gD1 = grad(ID{}, I1{}), gD2 = grad(ID{}, I2{});
                      const double g0D = grad(I0{}, ID{}), g1D = grad(I1{}, ID{}), g2D = grad(I2{}, ID{});
                      const double Phi0 = wp * (c_muh * (gD0 + g0D) + (D == 0 ? dterm : 0.0));
                      const double Phi1 = wp * (c_muh * (gD1 + g1D) + (D == 1 ? dterm : 0.0));
                      const double Phi2 = wp * (c_muh * (gD2 + g2D) + (D == 2 ? dterm : 0.0));
                      Xx[0] += Phi0 * nx0;
                      Xx[1] += Phi0 * nx1;
#pragma unroll
                      for (int g = 0; g < 3; ++g)
                        {
                          Xy[g] += Phi1 * c_g1.m[g][qx];
                          Xz[g] += Phi2 * c_g1.m[g][qx];
                        }
                    }
                    }
#pragma unroll
                  for (int g = 0; g < 3; ++g)
                    {
                      const double my = c_g1.m[g][qy];
                      Yx[0][g] += Xx[0] * my;
                      Yx[1][g] += Xx[1] * my;
#pragma unroll
                      for (int gx = 0; gx < 3; ++gx)
                        Yz[gx][g] += Xz[gx] * my;
                      Yy[g][0] += Xy[g] * ny0;
                      Yy[g][1] += Xy[g] * ny1;
                    }
                }
              // K = sum_k s(b_k)/h_k C^{dk}: fold 1/h_k in
#pragma unroll
              for (int i = 0; i < 3; ++i)
                {
                  Yx[0][i] *= S.ih[0];
                  Yx[1][i] *= S.ih[0];
                  Yy[i][0] *= S.ih[1];
                  Yy[i][1] *= S.ih[1];
#pragma unroll
                  for (int j = 0; j < 3; ++j)
                    Yz[i][j] *= S.ih[2];
                }
            }
          const double mz[3] = {c_g1.m[0][qz], c_g1.m[1][qz], c_g1.m[2][qz]};
          // push this z-level's part of the entries of the 8 vertices, in the order of a lexicographic cell loop
          static_for<8>([&](auto A) __attribute__((always_inline)) {
            constexpr int ax = 1 - (decltype(A)::value & 1), ay = 1 - ((decltype(A)::value >> 1) & 1), az = decltype(A)::value >> 2;
            const int hx = cx + ax, hy = cy + ay;
            if ((az == 0 ? dst.push_lo : dst.push_hi) && hx >= 1 && hx <= PN && hy >= 1 && hy <= PN)
              {
                const int nb = (nl0 + ax + PN * ay) * 27 + D;
                const double nza = az ? nz1 : nz0;
                static_for<8>([&](auto B) __attribute__((always_inline)) {
                  constexpr int bx = decltype(B)::value & 1, by = (decltype(B)::value >> 1) & 1, bz = decltype(B)::value >> 2;
                  constexpr int ox = bx - ax, oy = by - ay, oz = bz - az;
                  constexpr int gx = ax + bx, gy = ay + by, gz = az + bz;
                  constexpr int o9 = (ox + 1) + 3 * (oy + 1);
                  const double t0 = Yx[ax][gy], t1 = Yy[gx][ay], t2 = Yz[gx][gy];
                  const double txy = (bx ? t0 : -t0) + (by ? t1 : -t1);
                  const double e = txy * mz[gz] + (bz ? nza : -nza) * t2;
                  double *slab = (az == 0) ? (oz == 0 ? dst.lo_z0 : dst.lo_p1) : (oz == -1 ? dst.hi_m1 : dst.hi_z0);
                  lds_add(&slab[nb + o9 * 3], e);
                });
              }
          });
        }
    }

    // ---- role 3: (phi,phi) entries of one cell, same scheme; returns the 8 diagonal entries of the element matrix
    template <bool HET, bool OLDF>
    __device__ __forceinline__ void pp_role(const double *__restrict__ Ulo, const double *__restrict__ Uhi, const MatScal &S,
                                            double cell_lam, double cell_mu, bool cell_ok, const PushDst &dst, double *__restrict__ pp_lo_z0,
                                            double *__restrict__ pp_lo_p1, double *__restrict__ pp_hi_m1,
                                            double *__restrict__ pp_hi_z0, int nl0, int cx, int cy, double (&Mdiag)[8])
    {
      const bool use_pen = OLDF && S.gamma_fac != 0.0; // (the launcher instantiates OLDF when gamma != 0 or the scheme is monolithic)
      double M[27]; // M[g_x + 3 g_y + 9 g_z]
#pragma unroll
      for (int m = 0; m < 27; ++m)
        M[m] = 0.0;
      double V[4][8]; // u_x u_y u_z phi at the cell's vertices
#pragma unroll
      for (int f = 0; f < 4; ++f)
        load_cell_field_raw(Ulo + f * NPH, Uhi + f * NPH, V[f]);
#pragma unroll 1
      for (int qz = 0; qz < 3; ++qz)
        {
          double Y[3][3]; // [g_x][g_y]
#pragma unroll
          for (int i = 0; i < 3; ++i)
            Y[i][0] = Y[i][1] = Y[i][2] = 0.0;
          const double nz0 = c_g1.n[0][qz], nz1 = c_g1.n[1][qz];
          if (cell_ok)
            {
              double Dy[3][2]; // d/dy depends on the z-level only
              static_for<3>([&](auto F) __attribute__((always_inline)) {
                constexpr int f = decltype(F)::value;
                dy_of_field(V[f], nz0, nz1, S.ih[1], Dy[f]);
              });
#pragma unroll 1
              for (int qy = 0; qy < 3; ++qy)
                {
                  const double ny0 = c_g1.n[0][qy], ny1 = c_g1.n[1][qy];
                  const double wyz = S.vww[qy][qz];
                  double L[5][2], Dz[3][2], dummy[2];
                  static_for<3>([&](auto F) __attribute__((always_inline)) {
                    constexpr int f = decltype(F)::value;
                    line_of_field<false, true>(V[f], ny0, ny1, nz0, nz1, S.ih[1], S.ih[2], L[f], dummy, Dz[f]);
                    __builtin_amdgcn_sched_barrier(0);
                  });
                  line_of_field<false, false>(V[3], ny0, ny1, nz0, nz1, S.ih[1], S.ih[2], L[3], dummy, dummy);
                  L[4][0] = L[4][1] = 0.0;
                  if constexpr (OLDF)
                    {
                      if (use_pen) // phi_old enters only through the penalisation term (gamma != 0: monolithic runs)
                        {
                          double Vo[8];
                          load_cell_field_raw(Ulo + 4 * NPH, Uhi + 4 * NPH, Vo);
                          line_of_field<false, false>(Vo, ny0, ny1, nz0, nz1, S.ih[1], S.ih[2], L[4], dummy, dummy);
                        }
                    }
                  double Dx[3];
#pragma unroll
                  for (int f = 0; f < 3; ++f)
                    Dx[f] = (L[f][1] - L[f][0]) * S.ih[0];
                  double X[3] = {0.0, 0.0, 0.0};
                  if constexpr (!OLDF)
                    {
                      // Round 4: from coefficients, as in the (phi,u) roles.  Along the line every gradient is a0 + t a1, so
                      // c(q) = (1-kappa) sigma+:E + G_c/eps - 2(alpha_B-1) p div u is a quadratic c0 + c1 t + c2 t^2 and
                      // sum_q w c m_g = c0 s0[m_g] + c1 s1[m_g] + c2 s2[m_g]
                      const double y0 = Dy[1][0], y1 = Dy[1][1] - Dy[1][0], z0 = Dz[2][0], z1 = Dz[2][1] - Dz[2][0];
                      const double u0 = Dy[0][0] + Dx[1], u1 = Dy[0][1] - Dy[0][0];
                      const double v0 = Dz[0][0] + Dx[2], v1 = Dz[0][1] - Dz[0][0];
                      const double w0 = Dz[1][0] + Dy[2][0], w1 = (Dz[1][1] - Dz[1][0]) + (Dy[2][1] - Dy[2][0]);
                      const double a = Dx[0] + y0 + z0, b = y1 + z1; // tr E
                      const double e0 = (Dx[0] * Dx[0] + y0 * y0 + z0 * z0) + 0.5 * (u0 * u0 + v0 * v0 + w0 * w0);
                      const double e1 = 2.0 * (y0 * y1 + z0 * z1) + (u0 * u1 + v0 * v1 + w0 * w1);
                      const double e2 = (y1 * y1 + z1 * z1) + 0.5 * (u1 * u1 + v1 * v1 + w1 * w1);
                      const double la = HET ? cell_lam : S.lam, mu2 = 2 * (HET ? cell_mu : S.mu);
                      const double c0 = wyz * (S.omk * (la * a * a + mu2 * e0) + S.gc_eps - S.aB1p2 * a);
                      const double c1 = wyz * (S.omk * (la * (2.0 * a * b) + mu2 * e1) - S.aB1p2 * b);
                      const double c2 = wyz * (S.omk * (la * b * b + mu2 * e2));
                      static_for<3>([&](auto Gc) __attribute__((always_inline)) {
                        constexpr int g = decltype(Gc)::value;
                        X[g] = c0 * G1Sx<0, 2 + g>::v + c1 * G1Sx<1, 2 + g>::v + c2 * G1Sx<2, 2 + g>::v;
                      });
                    }
                  else
                    {
#pragma unroll
                  for (int qx = 0; qx < 3; ++qx)
                    {
                      const double nx0 = c_g1.n[0][qx], nx1 = c_g1.n[1][qx];
                      double gu[3][3];
#pragma unroll
                      for (int c = 0; c < 3; ++c)
                        {
                          gu[c][0] = Dx[c];
                          gu[c][1] = nx0 * Dy[c][0] + nx1 * Dy[c][1];
                          gu[c][2] = nx0 * Dz[c][0] + nx1 * Dz[c][1];
                        }
                      double pf = nx0 * L[3][0] + nx1 * L[3][1];
                      double pfo = nx0 * L[4][0] + nx1 * L[4][1];
                      if (S.monolithic)
                        {
                          pf = fmax(0.0, pf);
                          pfo = fmax(0.0, pfo);
                        }
                      const double trE = gu[0][0] + gu[1][1] + gu[2][2];
                      // E : E = sum_a E_aa^2 + 2 sum_{a<b} E_ab^2,  2 E_ab^2 = (g_ab + g_ba)^2 / 2
                      const double s01 = gu[0][1] + gu[1][0], s02 = gu[0][2] + gu[2][0], s12 = gu[1][2] + gu[2][1];
                      const double EE = (gu[0][0] * gu[0][0] + gu[1][1] * gu[1][1] + gu[2][2] * gu[2][2]) +
                                        0.5 * (s01 * s01 + s02 * s02 + s12 * s12);
                      const double spE = (HET ? cell_lam : S.lam) * trE * trE + 2 * (HET ? cell_mu : S.mu) * EE;       // sigma+ : E
                      const double pen = (!use_pen || (pf - pfo) < 0.0) ? 0.0 : S.gamma_fac; // cracks.cc:2311-2315, 2370
                      const double cq = S.omk * spE + S.gc_eps - S.aB1p2 * trE + pen;
                      const double wc = (wyz * c_g1.w[qx]) * cq;
                      X[0] += wc * c_g1.m[0][qx];
                      X[1] += wc * c_g1.m[1][qx];
                      X[2] += wc * c_g1.m[2][qx];
                    }
                    }
#pragma unroll
                  for (int gy = 0; gy < 3; ++gy)
#pragma unroll
                    for (int gx = 0; gx < 3; ++gx)
                      Y[gx][gy] += X[gx] * c_g1.m[gy][qy];
                }
            }
          const double mz[3] = {c_g1.m[0][qz], c_g1.m[1][qz], c_g1.m[2][qz]};
#pragma unroll
          for (int gz = 0; gz < 3; ++gz)
#pragma unroll
            for (int gy = 0; gy < 3; ++gy)
#pragma unroll
              for (int gx = 0; gx < 3; ++gx)
                M[gx + 3 * gy + 9 * gz] += Y[gx][gy] * mz[gz];
        }
      // G_c eps sum_q w grad N_a . grad N_b is the same for every cell of the box: host-precomputed (MatScal::lapM)
      if (cell_ok)
        {
#pragma unroll
          for (int m = 0; m < 27; ++m)
            M[m] += S.lapM[m];
        }
      // one push per cell: the 27 moments fit in registers (unlike the 54 numbers of a (phi,u) role)
      static_for<8>([&](auto A) __attribute__((always_inline)) {
        constexpr int ax = 1 - (decltype(A)::value & 1), ay = 1 - ((decltype(A)::value >> 1) & 1), az = decltype(A)::value >> 2;
        const int hx = cx + ax, hy = cy + ay;
        if ((az == 0 ? dst.push_lo : dst.push_hi) && hx >= 1 && hx <= PN && hy >= 1 && hy <= PN)
          {
            const int nb = (nl0 + ax + PN * ay) * 9;
            static_for<8>([&](auto B) __attribute__((always_inline)) {
              constexpr int bx = decltype(B)::value & 1, by = (decltype(B)::value >> 1) & 1, bz = decltype(B)::value >> 2;
              constexpr int ox = bx - ax, oy = by - ay, oz = bz - az;
              constexpr int o9 = (ox + 1) + 3 * (oy + 1);
              double *slab = (az == 0) ? (oz == 0 ? pp_lo_z0 : pp_lo_p1) : (oz == -1 ? pp_hi_m1 : pp_hi_z0);
              lds_add(&slab[nb + o9], M[(ax + bx) + 3 * (ay + by) + 9 * (az + bz)]);
            });
          }
      });
#pragma unroll
      for (int a = 0; a < 8; ++a)
        Mdiag[a] = M[2 * (a & 1) + 3 * 2 * ((a >> 1) & 1) + 9 * 2 * (a >> 2)];
    }

    // ===================================================================================== round 4
    // The roles from POLYNOMIAL COEFFICIENTS (staggered scheme without penalisation: nothing is clamped at the q-points).
    // On a box cell every nodal field is trilinear in the reference coordinates (t, s, r), its gradient is multilinear, so
    //   Phi^{dk} = w pf [c_muh (d_k u_d + d_d u_k) + delta_dk (c_la div u + cdiag)]      (degree <= 2 per variable)
    //   c        = (1-kappa) sigma+:E + G_c/eps - 2 (alpha_B-1) p div u                  (degree <= 2 per variable)
    // are polynomials with 27 coefficients, and the sums over the 27 Gauss points against the weights n_al / m_g are
    // contractions of those coefficients with the 3 x 5 constants s_j[W] = sum_q w t_q^j W(q) (G1Sx; the 3-point rule is exact
    // for every degree that occurs, the values are those of the quadrature up to rounding): vertex values -> 8 monomial
    // coefficients per field (12 subtractions), coefficient products (<= 64 per role field), three contraction stages of 54
    // FMAs per (d,k) -- no q-point is ever visited.  ~1100 instead of ~2500 instructions per cell and (phi,u) role, ~600 for
    // the (phi,phi) role; the z-contraction happens in registers, so a cell pushes ONCE per role (64 instead of 192 LDS adds).
    // the derivative along axis c of a trilinear field has no monomial with t_c: A^{dk} has a coefficient at idx iff ...
    constexpr bool a_nz(int D, int K, int idx) { return K == D ? idx != 7 : (!(idx & (1 << K)) || !(idx & (1 << D))); }
    constexpr bool pair_first(int D, int K, int ia, int ip) // first (ia outer, ip inner) contribution to its power?
    {
      const int pw = pow_of(ia, ip);
      for (int a = 0; a < 8; ++a)
        for (int q = 0; q < 8; ++q)
          {
            if (a == ia && q == ip)
              return true;
            if (a_nz(D, K, a) && pow_of(a, q) == pw)
              return false;
          }
      return true;
    }
    constexpr bool pow_any(int D, int K, int pw)
    {
      for (int a = 0; a < 8; ++a)
        for (int q = 0; q < 8; ++q)
          if (a_nz(D, K, a) && pow_of(a, q) == pw)
            return true;
      return false;
    }
    constexpr int ipow3(int a) { return a == 0 ? 1 : (a == 1 ? 3 : 9); }

    template <int D, bool HET, bool ONEPUSH>
    __device__ __forceinline__ void pu_role_poly(const double *__restrict__ Ulo, const double *__restrict__ Uhi, const MatScal &S,
                                                 double cell_muh, double cell_la, bool cell_ok, const PushDst &dst, int nl0, int cx, int cy)
    {
      const double c_muh = HET ? cell_muh : S.c_muh, c_la = HET ? cell_la : S.c_la;
      // vertices of the cell whose rows this lane pushes into (bit a of the vertex loop below)
      bool vok[8];
      static_for<8>([&](auto A) __attribute__((always_inline)) {
        constexpr int ax = 1 - (decltype(A)::value & 1), ay = 1 - ((decltype(A)::value >> 1) & 1), az = decltype(A)::value >> 2;
        const int hx = cx + ax, hy = cy + ay;
        vok[decltype(A)::value] = cell_ok && (az == 0 ? dst.push_lo : dst.push_hi) && hx >= 1 && hx <= PN && hy >= 1 && hy <= PN;
      });
      if (cell_ok)
        {
          double F0[4][8]; // u_x u_y u_z phi
          if constexpr (!ONEPUSH)
            {
#pragma unroll
              for (int f = 0; f < 4; ++f)
                {
                  load_cell_field_raw(Ulo + f * NPH, Uhi + f * NPH, F0[f]);
                  monomials(F0[f]);
                }
#pragma unroll
              for (int i = 0; i < 8; ++i)
                F0[3][i] *= S.vol; // JxW = vol w w w, and the weights are inside the constants s_j[W]
            }
          // ONEPUSH (round 6): the three k-parts of an entry are added in registers and pushed ONCE -- 64 instead of 192
          // LDS adds per cell and role (the LDS pipe of a CU, shared by two workgroups, was the busiest unit of this kernel:
          // 640 ds_add_f64 per workgroup and step).  The 3 x 18 numbers stay live (no negated copies: the sign is an
          // operand modifier of the add); k = D first, the only part that needs all three displacement fields.
          double Call[ONEPUSH ? 3 : 1][18];
          static_for<3>([&](auto Kc) __attribute__((always_inline)) {
            constexpr int kseq = decltype(Kc)::value;
            constexpr int k = !ONEPUSH ? kseq : (kseq == 0 ? D : (kseq == 1 ? (D == 0 ? 1 : 0) : (D == 2 ? 1 : 2)));
            constexpr int a1 = k == 0 ? 1 : 0, a2 = k == 2 ? 1 : 2; // the other axes, ascending
            // ONEPUSH: the fields a part needs are read from the nodal ring again (volatile reads: not merged with the last
            // part's), so that at most three of the four fields' coefficients are live next to the finished parts
            double Fk[4][8];
            if constexpr (ONEPUSH)
              static_for<4>([&](auto Fc) __attribute__((always_inline)) {
                constexpr int f = decltype(Fc)::value;
                if constexpr (k == D || f == D || f == k || f == 3)
                  {
                    const double *lo = Ulo + f * NPH, *hi = Uhi + f * NPH;
                    Fk[f][0] = lds_read64(lo), Fk[f][1] = lds_read64(lo + 1), Fk[f][2] = lds_read64(lo + PH), Fk[f][3] = lds_read64(lo + PH + 1);
                    Fk[f][4] = lds_read64(hi), Fk[f][5] = lds_read64(hi + 1), Fk[f][6] = lds_read64(hi + PH), Fk[f][7] = lds_read64(hi + PH + 1);
                    monomials(Fk[f]);
                    if constexpr (f == 3)
                      {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                          Fk[3][i] *= S.vol;
                      }
                  }
              });
            double(&F)[4][8] = ONEPUSH ? Fk : F0;
            const double ihk = S.ih[k];
            const double sDk = c_muh * ihk * ihk, skD = c_muh * S.ih[D] * ihk;
            const double sl[3] = {c_la * S.ih[0] * ihk, c_la * S.ih[1] * ihk, c_la * S.ih[2] * ihk};
            double A[8];
            static_for<8>([&](auto Ic) __attribute__((always_inline)) {
              constexpr int idx = decltype(Ic)::value;
              if constexpr (a_nz(D, k, idx))
                {
                  double acc = 0.0;
                  bool any = false;
                  auto add = [&](double x) __attribute__((always_inline)) {
                    acc = any ? acc + x : x;
                    any = true;
                  };
                  if constexpr (k == D)
                    {
                      if constexpr (!(idx & (1 << k)))
                        add((2.0 * sDk + sl[k]) * F[D][idx | (1 << k)]);
                      static_for<3>([&](auto Cc) __attribute__((always_inline)) {
                        constexpr int c = decltype(Cc)::value;
                        if constexpr (c != k && !(idx & (1 << c)))
                          add(sl[c] * F[c][idx | (1 << c)]);
                      });
                      if constexpr (idx == 0)
                        add(S.cdiag * ihk);
                    }
                  else
                    {
                      if constexpr (!(idx & (1 << k)))
                        add(sDk * F[D][idx | (1 << k)]); // d_k u_D
                      if constexpr (!(idx & (1 << D)))
                        add(skD * F[k][idx | (1 << D)]); // d_D u_k
                    }
                  A[idx] = acc;
                }
            });
            // contraction: the special axis k first (2 weights n_al), then a1, a2 (3 weights m_g each)
            double R1[2][3][3], R2[2][3][3];
            if constexpr (!ONEPUSH)
              {
                double c[27];
                static_for<8>([&](auto Ia) __attribute__((always_inline)) {
                  constexpr int ia = decltype(Ia)::value;
                  if constexpr (a_nz(D, k, ia))
                    static_for<8>([&](auto Ip) __attribute__((always_inline)) {
                      constexpr int ip = decltype(Ip)::value;
                      constexpr int pw = pow_of(ia, ip);
                      if constexpr (pair_first(D, k, ia, ip))
                        c[pw] = A[ia] * F[3][ip];
                      else
                        c[pw] = fma(A[ia], F[3][ip], c[pw]);
                    });
                });
                static_for<27>([&](auto Pw) __attribute__((always_inline)) {
                  if constexpr (!pow_any(D, k, decltype(Pw)::value))
                    c[decltype(Pw)::value] = 0.0;
                });
                static_for<2>([&](auto Wc) __attribute__((always_inline)) {
                  constexpr int w = decltype(Wc)::value;
#pragma unroll
                  for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int l = 0; l < 3; ++l)
                      {
                        const int b = j * ipow3(a1) + l * ipow3(a2);
                        R1[w][j][l] = c[b] * G1Sx<0, w>::v + c[b + ipow3(k)] * G1Sx<1, w>::v + c[b + 2 * ipow3(k)] * G1Sx<2, w>::v;
                      }
                });
              }
            else
              {
                // the 27 coefficients of the product polynomial are formed three at a time (the powers 0, 1, 2 of t_k at fixed
                // powers of the other two variables) and contracted along k at once: 3 instead of 27 of them are live
                static_for<9>([&](auto JL) __attribute__((always_inline)) {
                  constexpr int j = decltype(JL)::value % 3, l = decltype(JL)::value / 3;
                  double c3[3] = {0.0, 0.0, 0.0};
                  static_for<3>([&](auto Pk) __attribute__((always_inline)) {
                    constexpr int pk = decltype(Pk)::value;
                    constexpr int pw = j * ipow3(a1) + l * ipow3(a2) + pk * ipow3(k);
                    if constexpr (pow_any(D, k, pw))
                      {
                        bool any = false;
                        static_for<8>([&](auto Ia) __attribute__((always_inline)) {
                          constexpr int ia = decltype(Ia)::value;
                          if constexpr (a_nz(D, k, ia))
                            static_for<8>([&](auto Ip) __attribute__((always_inline)) {
                              constexpr int ip = decltype(Ip)::value;
                              if constexpr (pow_of(ia, ip) == pw)
                                {
                                  c3[pk] = any ? fma(A[ia], F[3][ip], c3[pk]) : A[ia] * F[3][ip];
                                  any = true;
                                }
                            });
                        });
                      }
                  });
                  R1[0][j][l] = c3[0] * G1Sx<0, 0>::v + c3[1] * G1Sx<1, 0>::v + c3[2] * G1Sx<2, 0>::v;
                  R1[1][j][l] = c3[0] * G1Sx<0, 1>::v + c3[1] * G1Sx<1, 1>::v + c3[2] * G1Sx<2, 1>::v;
                  if constexpr (l == 0 || l == 1)
                    __builtin_amdgcn_sched_barrier(0);
                });
                __builtin_amdgcn_sched_barrier(0); // the fields are dead from here on
              }
            static_for<3>([&](auto Gc) __attribute__((always_inline)) {
              constexpr int g = decltype(Gc)::value;
#pragma unroll
              for (int w = 0; w < 2; ++w)
#pragma unroll
                for (int l = 0; l < 3; ++l)
                  R2[w][g][l] = R1[w][0][l] * G1Sx<0, 2 + g>::v + R1[w][1][l] * G1Sx<1, 2 + g>::v + R1[w][2][l] * G1Sx<2, 2 + g>::v;
            });
            // C[al + 2 (g_i + 3 g_j)], i < j the axes other than k (1/h_k and the volume folded in), and its negative: the entry
            // of trial vertex b gets s(b_k) C; the LDS add takes no sign
            double Cp[18], Cn[18];
            static_for<3>([&](auto Gc) __attribute__((always_inline)) {
              constexpr int g2 = decltype(Gc)::value;
#pragma unroll
              for (int w = 0; w < 2; ++w)
#pragma unroll
                for (int g1 = 0; g1 < 3; ++g1)
                  {
                    const double x = R2[w][g1][0] * G1Sx<0, 2 + g2>::v + R2[w][g1][1] * G1Sx<1, 2 + g2>::v + R2[w][g1][2] * G1Sx<2, 2 + g2>::v;
                    if constexpr (ONEPUSH)
                      {
                        double xv = x;
                        asm volatile("" : "+v"(xv)); // finished HERE: not sunk (with its three operands) into the vertex blocks
                        Call[ONEPUSH ? k : 0][w + 2 * (g1 + 3 * g2)] = xv;
                      }
                    else
                      {
                        Cp[w + 2 * (g1 + 3 * g2)] = x;
                        Cn[w + 2 * (g1 + 3 * g2)] = -x;
                      }
                  }
            });
            // the k-part of the cell's entries, vertex by vertex in the order of a lexicographic cell loop (three adds per
            // entry and cell, k = 0, 1, 2: a fixed order)
            if constexpr (ONEPUSH)
              __builtin_amdgcn_sched_barrier(0); // the parts one after the other: their temporaries must not overlap
            if constexpr (!ONEPUSH)
            static_for<8>([&](auto A) __attribute__((always_inline)) {
              constexpr int ax = 1 - (decltype(A)::value & 1), ay = 1 - ((decltype(A)::value >> 1) & 1), az = decltype(A)::value >> 2;
              if (vok[decltype(A)::value])
                {
                  const int nb = (nl0 + ax + PN * ay) * 27 + D;
                  static_for<8>([&](auto B) __attribute__((always_inline)) {
                    constexpr int bx = decltype(B)::value & 1, by = (decltype(B)::value >> 1) & 1, bz = decltype(B)::value >> 2;
                    constexpr int ox = bx - ax, oy = by - ay, oz = bz - az;
                    constexpr int gx = ax + bx, gy = ay + by, gz = az + bz;
                    constexpr int o9 = (ox + 1) + 3 * (oy + 1);
                    constexpr int ci = k == 0 ? ax + 2 * (gy + 3 * gz) : (k == 1 ? ay + 2 * (gx + 3 * gz) : az + 2 * (gx + 3 * gy));
                    constexpr int bk = k == 0 ? bx : (k == 1 ? by : bz);
                    double *slab = (az == 0) ? (oz == 0 ? dst.lo_z0 : dst.lo_p1) : (oz == -1 ? dst.hi_m1 : dst.hi_z0);
#ifdef PFM_PHI_ABLATE_PUSH // measurement only (wrong values): two of the three parts are computed but not pushed
                    const double pushed = bk ? Cp[ci] : Cn[ci];
                    if constexpr (k != 0)
                      asm volatile("" ::"v"(pushed));
                    else
#else
                    const double pushed = bk ? Cp[ci] : Cn[ci];
#endif
                    lds_add(&slab[nb + o9 * 3], pushed);
                  });
                }
            });
          });
          if constexpr (ONEPUSH)
            static_for<8>([&](auto A) __attribute__((always_inline)) {
              constexpr int ax = 1 - (decltype(A)::value & 1), ay = 1 - ((decltype(A)::value >> 1) & 1), az = decltype(A)::value >> 2;
              if (vok[decltype(A)::value])
                {
                  const int nb = (nl0 + ax + PN * ay) * 27 + D;
                  static_for<8>([&](auto B) __attribute__((always_inline)) {
                    constexpr int bx = decltype(B)::value & 1, by = (decltype(B)::value >> 1) & 1, bz = decltype(B)::value >> 2;
                    constexpr int ox = bx - ax, oy = by - ay, oz = bz - az;
                    constexpr int gx = ax + bx, gy = ay + by, gz = az + bz;
                    constexpr int o9 = (ox + 1) + 3 * (oy + 1);
                    const double c0 = Call[0][ax + 2 * (gy + 3 * gz)], c1 = Call[ONEPUSH ? 1 : 0][ay + 2 * (gx + 3 * gz)],
                                 c2 = Call[ONEPUSH ? 2 : 0][az + 2 * (gx + 3 * gy)];
                    // fixed order: x, y, z part.  The adds are opaque to the compiler: as plain expressions the partial sums shared
                    // between vertex blocks are hoisted in front of the first block, all 64 entries at once (128 registers)
                    const double e = add_signed<!bz, false>(c2, add_signed<!bx, !by>(c0, c1));
                    double *slab = (az == 0) ? (oz == 0 ? dst.lo_z0 : dst.lo_p1) : (oz == -1 ? dst.hi_m1 : dst.hi_z0);
                    lds_add(&slab[nb + o9 * 3], e);
                  });
                }
            });
        }
    }

    template <bool HET, bool RESV>
    __device__ __forceinline__ void pp_role_poly(const double *__restrict__ Ulo, const double *__restrict__ Uhi, const MatScal &S,
                                                 double cell_lam, double cell_mu, bool cell_ok, const PushDst &dst, double *__restrict__ pp_lo_z0,
                                                 double *__restrict__ pp_lo_p1, double *__restrict__ pp_hi_m1,
                                                 double *__restrict__ pp_hi_z0, int nl0, int cx, int cy, double (&Mdiag)[8],
                                                 double (&Kphi)[8] /* RESV: sum_b K_phiphi[a][b] phi_b of this cell, by vertex a */,
                                                 double lap_lane /* MatScal::lapM[lane] in lane < 27 */)
    {
      double M[27]; // M[g_x + 3 g_y + 9 g_z]
#pragma unroll
      for (int m = 0; m < 27; ++m)
        M[m] = 0.0;
      if (cell_ok)
        {
          double F[3][8];
#pragma unroll
          for (int f = 0; f < 3; ++f)
            {
              load_cell_field_raw(Ulo + f * NPH, Uhi + f * NPH, F[f]);
              monomials(F[f]);
            }
          // gradient polynomials: d_k u_c = ih_k sum_{idx without bit k} F[c][idx | 1 << k] x^idx, formed where they are used
          auto G = [&](auto Cc, auto Kc, auto Ic) __attribute__((always_inline)) -> double {
            constexpr int c = decltype(Cc)::value, k = decltype(Kc)::value, idx = decltype(Ic)::value;
            static_assert(!(idx & (1 << k)), "no such monomial in this derivative");
            return S.ih[k] * F[c][idx | (1 << k)];
          };
          using I0 = std::integral_constant<int, 0>;
          using I1 = std::integral_constant<int, 1>;
          using I2 = std::integral_constant<int, 2>;
          using I3 = std::integral_constant<int, 3>;
          using I4 = std::integral_constant<int, 4>;
          using I5 = std::integral_constant<int, 5>;
          using I6 = std::integral_constant<int, 6>;
          double Q[27];
#pragma unroll
          for (int m = 0; m < 27; ++m)
            Q[m] = 0.0;
          const double la = HET ? cell_lam : S.lam, mu2 = 2 * (HET ? cell_mu : S.mu);
          // tr E = div u (kept: the linear term of c needs it again)
          double T[8];
          T[0] = (G(I0{}, I0{}, I0{}) + G(I1{}, I1{}, I0{})) + G(I2{}, I2{}, I0{});
          T[1] = G(I1{}, I1{}, I1{}) + G(I2{}, I2{}, I1{});
          T[2] = G(I0{}, I0{}, I2{}) + G(I2{}, I2{}, I2{});
          T[3] = G(I2{}, I2{}, I3{});
          T[4] = G(I0{}, I0{}, I4{}) + G(I1{}, I1{}, I4{});
          T[5] = G(I1{}, I1{}, I5{});
          T[6] = G(I0{}, I0{}, I6{});
          add_square<0x7f>(T, la, Q);
          static_for<3>([&](auto Cc) __attribute__((always_inline)) {
            constexpr int c = decltype(Cc)::value;
            constexpr int mc = c == 0 ? NOX : (c == 1 ? NOY : NOZ);
            double P[8];
            static_for<8>([&](auto Ic) __attribute__((always_inline)) {
              if constexpr ((mc >> decltype(Ic)::value) & 1)
                P[decltype(Ic)::value] = G(Cc, Cc, Ic);
            });
            add_square<mc>(P, mu2, Q);
          });
          // 2 E_cd = d_d u_c + d_c u_d
          static_for<3>([&](auto Pc) __attribute__((always_inline)) {
            constexpr int pr = decltype(Pc)::value;
            constexpr int c = pr == 2 ? 1 : 0, d = pr == 0 ? 1 : 2; // (0,1), (0,2), (1,2)
            constexpr int mc = c == 0 ? NOX : NOY, md = d == 1 ? NOY : NOZ;
            using IC = std::integral_constant<int, c>;
            using ID = std::integral_constant<int, d>;
            double Sp[8];
            static_for<8>([&](auto Ic) __attribute__((always_inline)) {
              constexpr int idx = decltype(Ic)::value;
              constexpr bool hd = (md >> idx) & 1, hc = (mc >> idx) & 1; // d_d u_c lives on the monomials without x_d
              if constexpr (hd && hc)
                Sp[idx] = G(IC{}, ID{}, Ic) + G(ID{}, IC{}, Ic);
              else if constexpr (hd)
                Sp[idx] = G(IC{}, ID{}, Ic);
              else if constexpr (hc)
                Sp[idx] = G(ID{}, IC{}, Ic);
            });
            add_square<(mc | md)>(Sp, 0.5 * mu2, Q);
          });
          // c = vol [ (1-kappa) sigma+:E + G_c/eps - 2 (alpha_B-1) p div u ]
          const double ov = S.omk * S.vol;
          double c[27];
#pragma unroll
          for (int m = 0; m < 27; ++m)
            c[m] = ov * Q[m];
          c[0] += S.gc_eps * S.vol;
          const double av = S.aB1p2 * S.vol;
          static_for<7>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int idx = decltype(Ic)::value;
            c[pow_of(idx, 0)] = fma(-av, T[idx], c[pow_of(idx, 0)]);
          });
          double R1[3][3][3], R2[3][3][3];
          static_for<3>([&](auto Gc) __attribute__((always_inline)) {
            constexpr int g = decltype(Gc)::value;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
              for (int l = 0; l < 3; ++l)
                R1[g][j][l] = c[3 * j + 9 * l] * G1Sx<0, 2 + g>::v + c[1 + 3 * j + 9 * l] * G1Sx<1, 2 + g>::v + c[2 + 3 * j + 9 * l] * G1Sx<2, 2 + g>::v;
          });
          static_for<3>([&](auto Gc) __attribute__((always_inline)) {
            constexpr int g = decltype(Gc)::value;
#pragma unroll
            for (int gx = 0; gx < 3; ++gx)
#pragma unroll
              for (int l = 0; l < 3; ++l)
                R2[gx][g][l] = R1[gx][0][l] * G1Sx<0, 2 + g>::v + R1[gx][1][l] * G1Sx<1, 2 + g>::v + R1[gx][2][l] * G1Sx<2, 2 + g>::v;
          });
          static_for<3>([&](auto Gc) __attribute__((always_inline)) {
            constexpr int g = decltype(Gc)::value;
#pragma unroll
            for (int gx = 0; gx < 3; ++gx)
#pragma unroll
              for (int gy = 0; gy < 3; ++gy)
                {
                  // MatScal::lapM[m] out of lane m of a register that wave 3 loaded once (v_readlane: two VALU instructions,
                  // no memory).  Read as S.lapM[m] these were 27 scalar loads with a wait each -- 27 scalar-cache round trips in
                  // a row, ~3.7k cycles per step in the longest role of the workgroup (phase clock, round 6).  (The 18
                  // constants they are made of in one inline-asm s_load batch: measured, +0.17 ms -- 36 pinned scalar registers.)
                  const int lm = gx + 3 * gy + 9 * g;
                  const double lapm = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(lap_lane), lm),
                                                       __builtin_amdgcn_readlane(__double2loint(lap_lane), lm));
                  double mv = (R2[gx][gy][0] * G1Sx<0, 2 + g>::v + R2[gx][gy][1] * G1Sx<1, 2 + g>::v + R2[gx][gy][2] * G1Sx<2, 2 + g>::v) +
                              lapm;
                  asm volatile("" : "+v"(mv)); // finished here (pu_role_poly: not sunk into the vertex blocks with its operands)
                  M[gx + 3 * gy + 9 * g] = mv;
                }
          });
        }
      static_for<8>([&](auto A) __attribute__((always_inline)) {
        constexpr int ax = 1 - (decltype(A)::value & 1), ay = 1 - ((decltype(A)::value >> 1) & 1), az = decltype(A)::value >> 2;
        const int hx = cx + ax, hy = cy + ay;
        if ((az == 0 ? dst.push_lo : dst.push_hi) && hx >= 1 && hx <= PN && hy >= 1 && hy <= PN)
          {
            const int nb = (nl0 + ax + PN * ay) * 9;
            static_for<8>([&](auto B) __attribute__((always_inline)) {
              constexpr int bx = decltype(B)::value & 1, by = (decltype(B)::value >> 1) & 1, bz = decltype(B)::value >> 2;
              constexpr int ox = bx - ax, oy = by - ay, oz = bz - az;
              constexpr int o9 = (ox + 1) + 3 * (oy + 1);
              double *slab = (az == 0) ? (oz == 0 ? pp_lo_z0 : pp_lo_p1) : (oz == -1 ? pp_hi_m1 : pp_hi_z0);
              lds_add(&slab[nb + o9], M[(ax + bx) + 3 * (ay + by) + 9 * (az + bz)]);
            });
          }
      });
#pragma unroll
      for (int a = 0; a < 8; ++a)
        Mdiag[a] = M[2 * (a & 1) + 3 * 2 * ((a >> 1) & 1) + 9 * 2 * (a >> 2)];
      if constexpr (RESV)
        {
          // the cell's part of K_phiphi phi, from the moments in registers (round 6: the rows used to be read back from LDS
          // behind the pushes -- a wait for the LDS queue of the whole workgroup in the middle of the longest role)
          double Pf[8];
          load_cell_field_raw(Ulo + 3 * NPH, Uhi + 3 * NPH, Pf);
#pragma unroll
          for (int a = 0; a < 8; ++a)
            {
              double acc = 0.0;
#pragma unroll
              for (int b = 0; b < 8; ++b)
                acc = fma(M[((a & 1) + (b & 1)) + 3 * (((a >> 1) & 1) + ((b >> 1) & 1)) + 9 * ((a >> 2) + (b >> 2))], Pf[b], acc);
              Kphi[a] = acc;
            }
        }
    }

    // =====================================================================================
    template <int NCOL, int CLK = 0 /* profiling only: 1 = cycles per phase of thread 0, 2 = cycles per role */,
              bool HET = false /* per-cell Lame coefficients (CartView::cell_lam) */,
              bool RES = false /* also writes the phase-field rows of the residual (res_pde) */,
              bool OLDF = false /* the general form: q-point loops along x (clamped phase field of the monolithic scheme,
                                   penalisation term) and phi_old / phi_oldold in the nodal ring; else the x-contraction from
                                   coefficients, the old fields fetched where the rare placeholder path needs them */>
    __global__ __launch_bounds__(NT4, 2) void k_cart_phi4(DevView v, CartView cv, const MatScal *__restrict__ Sp, double *__restrict__ vals_pu,
                                                          double *__restrict__ vals_pp, double *__restrict__ vals_uu,
                                                          double *__restrict__ vals_up /* blocked layout: structurally zero (u,phi) block, cleared here */,
                                                          int zc_in /* node planes per chunk */,
                                                          unsigned long long *__restrict__ dbg, double *__restrict__ res_pde)
    {
      constexpr int NF = OLDF ? 6 : 4;
      __shared__ Lds4<NF> s;
      // wave priorities per phase (the requests and the copy-out are a few instructions with long latencies: issued ahead
      // of the co-resident workgroup's arithmetic they finish sooner and cost it nothing measurable; -0.2 ms at 216^3).
      // PFM_NO_PRIO=1 switches them off (A/B runs): the launcher then passes the chunk length negated
      const bool PRIO = zc_in > 0;
      const int zc = zc_in < 0 ? -zc_in : zc_in;
      const MatScal &S = *Sp; // per-launch scalars live in device memory: loaded where used, not pinned in SGPRs
      // the 27 Laplace moments, one per lane (pp_role_poly reads lane m with v_readlane): a vector load before any store
      const double lap_lane = Sp->lapM[(threadIdx.x & 63) < 27 ? (threadIdx.x & 63) : 0];
      long long tclk = 0;
      auto stamp = [&](int phase) __attribute__((always_inline)) {
        if constexpr (CLK == 1)
          {
            const long long now = clock64();
            if (threadIdx.x == 0 && phase >= 0)
              dbg[(size_t)blockIdx.x * 16 + phase] += (unsigned long long)(now - tclk); // one slot per workgroup
            tclk = now;
          }
      };
      stamp(-1);
      const int t = threadIdx.x, lane = t & 63;
      const int role = __builtin_amdgcn_readfirstlane(t >> 6); // wave-uniform: scalar branches between the roles
      const int cx = lane % PT, cy = lane / PT;
      const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1;
      const int ntx = (OWX + PN - 1) / PN, nty = (OWY + PN - 1) / PN;
      const int bid = xcd_tile_index();
      if (bid >= ntx * nty * ((cv.o1[2] - cv.o0[2] + zc) / zc))
        return; // padding of the XCD-aware grid
      const int tix = bid % ntx, tiy = (bid / ntx) % nty, chunk = bid / (ntx * nty);
      const int i0 = cv.o0[0] + tix * PN, j0 = cv.o0[1] + tiy * PN;
      const int kA = cv.o0[2] + chunk * zc;
      const int kB = min(kA + zc, cv.o1[2] + 1); // node planes [kA, kB)
      const int ci = i0 - 1 + cx, cj = j0 - 1 + cy;
      const bool col_ok = ci >= 0 && ci < cv.NX - 1 && cj >= 0 && cj < cv.NY - 1;
      const int hb = cy * PH + cx;               // halo index of the cell's (0,0) vertex
      const int nl0 = (cx - 1) + PN * (cy - 1);  // owned-node index of that vertex (may be out of range)

      // Nodal plane / row info of a plane: global memory -> LDS directly (global_load_lds: destination = wave-uniform
      // base + 4 * lane, no staging registers), issued ahead of the copy-out stores of the previous step and waited for
      // at the top of the step that consumes them.  Lanes without a source (outside the mesh / the owned box) write
      // the neutral value themselves.  Waves 0..2: dwords [64 w, 64 w + 64) of each of the 6 nodal fields (81 doubles
      // = 162 dwords), one node id per lane; their even lanes also fetch the node's flag byte, which goes through a
      // register (sub-dword transfers to LDS would still occupy one dword per lane) and is stored at the end of the
      // step.  Wave 3: value offset (64-bit: 98 dwords) and neighbour mask of the 49 rows.
      // (round 4: the flag byte travels global -> LDS like everything else.  As a register load it was live across the
      // march and the compiler waited for it with vmcnt(0) at the entry of every role, right behind the requests -- which
      // made them synchronous -- and behind the copy-out stores -- which made every wave drain its own stores.)
      // node id of the lane's halo dword in plane kz (role < 3) -- SEPARATE from the requests: cart_local_id may look the id up
      // in the lattice table (a load inside a branch, which the compiler waits for at the join with vmcnt(0)); with the
      // transfers of the same step already in flight that wait made them synchronous (round 4: ids of the plane AND of
      // the rows first, then all requests)
      // single-rank box (every lattice node owned, numbered lexicographically): ids by arithmetic, no table, no load -- and
      // therefore no wait in front of the requests (the look-up path waits for its loads there with vmcnt(0), i.e. for the
      // copy-out stores of the previous step as well)
      const bool all_lex = cv.owned_lex && cv.o0[0] == 0 && cv.o0[1] == 0 && cv.o0[2] == 0 && cv.o1[0] == cv.NX - 1 &&
                           cv.o1[1] == cv.NY - 1 && cv.o1[2] == cv.NZ - 1;
      auto lex_id = [&](int gi, int gj, int gk) __attribute__((always_inline)) -> unsigned {
        return (unsigned)(gi + cv.NX * (gj + cv.NY * gk));
      };
      auto plane_node = [&](int kz, bool &ok, bool &inside, int &dw, bool lex) __attribute__((always_inline)) -> unsigned {
        int lq = lane;
        asm volatile("" : "+v"(lq)); // recomputed per step, not kept live across the march
        dw = 64 * role + lq;
        const int hn = dw >> 1;
        const int gi = i0 - 1 + hn % PH, gj = j0 - 1 + hn / PH;
        inside = dw < 2 * NPH;
        ok = role < 3 && inside && kz >= 0 && kz < cv.NZ && gi >= 0 && gi < cv.NX && gj >= 0 && gj < cv.NY;
        if (lex)
          return ok ? lex_id(gi, gj, kz) : 0u;
        const int id = ok ? cart_local_id(cv, gi, gj, kz) : -1;
        ok = id >= 0; // (a level lattice of the 3-D overlay holds -1 where the level has no node: neutral values)
        return ok ? (unsigned)id : 0u;
      };
      auto dma_plane_at = [&](unsigned n, bool ok, bool inside, int dw, int buf, int kz) __attribute__((always_inline)) {
        if (role < 3)
          {
            const unsigned boff = 8u * n + 4u * (dw & 1);
            const double *const fld[6] = {v.u[0], v.u[1], v.u[2], v.phi, v.phi_old, v.phi_oldold};
            if (ok)
              {
#pragma unroll
                for (int c = 0; c < NF; ++c)
                  dma_b32(fld[c], boff, reinterpret_cast<uint32_t *>(&s.U[buf][c][0]) + 64 * role);
                dma_u8(v.node_flags, n, &s.flag[kz & 3][0] + 64 * role); // both lanes of a node fetch its byte: [2 hn] is read
              }
            else if (inside)
              {
#pragma unroll
                for (int c = 0; c < NF; ++c)
                  reinterpret_cast<uint32_t *>(&s.U[buf][c][0])[dw] = 0u;
                s.flag[kz & 3][dw] = 0u;
              }
          }
      };
      auto dma_plane = [&](int kz, int buf) __attribute__((always_inline)) {
        bool ok, inside;
        int dw;
        const unsigned n = plane_node(kz, ok, inside, dw, false);
        dma_plane_at(n, ok, inside, dw, buf, kz);
      };
      auto rows_node = [&](int kz, bool &ok, bool lex) __attribute__((always_inline)) -> unsigned {
        int lq = lane;
        asm volatile("" : "+v"(lq));
        const int gi = i0 + lq % PN, gj = j0 + lq / PN;
        ok = role == 3 && lq < NPN && gi <= cv.o1[0] && gj <= cv.o1[1];
        if (lex)
          return ok ? lex_id(gi, gj, kz) : 0xffffffffu;
        const int id = ok ? cart_row_id(cv, gi, gj, kz) : -1; // (CartView::row_of_box: -1 = not a row of this launch)
        ok = id >= 0;
        return (unsigned)id;
      };
      auto dma_rows_at = [&](unsigned n, bool ok, int kz) __attribute__((always_inline)) {
        const int par = kz & 1;
        if (role == 3)
          {
            int lq = lane;
            asm volatile("" : "+v"(lq));
            if (ok)
              dma_b32(cv.nbr_mask, 4u * n, &s.deg[par][0]);
            else if (lq < NPN)
              s.deg[par][lq] = 0x7ffffff;
            uint32_t *dst = reinterpret_cast<uint32_t *>(&s.off[par][0]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
              {
                const int dw = 64 * j + lq;
                const unsigned nj = (unsigned)__shfl((int)n, dw >> 1); // lanes 2 nl, 2 nl + 1: the two halves of row nl
                if (dw < 2 * NPN && nj != 0xffffffffu)
                  dma_b32(v.nadj_ptr, 8u * nj + 4u * (dw & 1), dst + 64 * j);
                else if (dw < 2 * NPN)
                  dst[dw] = 0xffffffffu; // off = -1: not an owned node
              }
          }
      };
      auto dma_rows = [&](int kz) __attribute__((always_inline)) {
        bool ok;
        const unsigned n = rows_node(kz, ok, false);
        dma_rows_at(n, ok, kz);
      };

      // every push accumulates: staged rows start at zero and the copy-out clears what it has streamed out
      for (int i = t; i < 5 * SLAB_PU; i += NT4)
        (&s.pu[0][0])[i] = 0.0;
      for (int i = t; i < 5 * SLAB_PP; i += NT4)
        (&s.pp[0][0])[i] = 0.0;
      for (int i = t; i < 2 * NPN * 2; i += NT4)
        (&s.ex[0][0][0])[i] = 0.0;
      for (int i = t; i < 2 * NPN; i += NT4)
        (&s.rs[0][0])[i] = 0.0;
      if (t < 4)
        s.anyflag[t] = 0;
      dma_plane(kA - 1, 0);
      dma_plane(kA, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t < NPH && s.flag[(kA - 1) & 3][2 * t])
        s.anyflag[(kA - 1) & 3] = 1;
      int nst = 0;       // lower bound of the vector-memory instructions this wave has issued behind its last requests
#pragma unroll 1
      for (int ck = kA - 1; ck < kB; ++ck)
        {
          const int lo = (ck - (kA - 1)) & 1, hi = lo ^ 1;
          const int cp = ck & 1, np = cp ^ 1;
          stamp(3);
          if constexpr (CLK == 2)
            tclk = clock64();
          // plane ck + 1 and the rows of plane ck were requested before the previous step's copy-out.  vmcnt counts loads
          // and stores in issue order: with at least 28 stores issued behind the requests (fast blocked copy-out: 7 y-lines x
          // 4 unconditional stores per wave), "at most 28 outstanding" means the requests have landed -- the wave does not
          // wait for its own stores to drain, they have the whole next role phase for that
          if (nst >= 28)
            asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
          else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          nst = 0;
          if (t == 0)
            {
              s.anyflag[(ck + 1) & 3] = 0;
              s.irregular[cp] = 0;
              s.incomplete[cp] = 0;
            }
          lds_barrier();
          stamp(0);
          if (t < NPH && s.flag[(ck + 1) & 3][2 * t])
            s.anyflag[(ck + 1) & 3] = 1;
          if (ck >= kA && t >= 128 && t < 128 + NPN && s.off[cp][t - 128] >= 0 && s.deg[cp][t - 128] != 0x7ffffff)
            s.irregular[cp] = 1;
          if (ck >= kA && t >= 128 && t < 128 + NPN && (s.off[cp][t - 128] < 0 || ((unsigned)s.deg[cp][t - 128] & 0x7ffffffu) != 0x7ffffffu))
            s.incomplete[cp] = 1;

          // ---- entries of layer ck, pushed into the rows of planes ck (lower vertices) and ck+1 (upper vertices)
          if (PRIO)
            __builtin_amdgcn_s_setprio(0);
          const bool cell_ok = col_ok && ck >= 0 && ck < cv.NZ - 1;
          PushDst dst;
          dst.push_lo = ck >= kA;
          dst.push_hi = ck + 1 < kB;
          dst.lo_z0 = s.pu[2 + cp];
          dst.lo_p1 = s.pu[4];
          dst.hi_m1 = s.pu[np];
          dst.hi_z0 = s.pu[2 + np];
          const double *Ulo = &s.U[lo][0][hb], *Uhi = &s.U[hi][0][hb];
          double lam = 0.0, mu = 0.0, c_muh = 0.0, c_la = 0.0;
          if (HET && cell_ok) // heterogeneous material, cracks.cc:2207-2216
            {
              const long long cidx = ci + (long long)(cv.NX - 1) * (cj + (long long)(cv.NY - 1) * ck);
              lam = cv.cell_lam[cidx];
              mu = cv.cell_mu[cidx];
              c_muh = 2.0 * (1.0 - S.kappa) * mu;
              c_la = 2.0 * (1.0 - S.kappa) * lam;
            }
          auto pu = [&](auto Dc) __attribute__((always_inline)) {
            constexpr int d = decltype(Dc)::value;
            if constexpr (OLDF)
              pu_role<d, HET, true>(Ulo, Uhi, S, c_muh, c_la, cell_ok, dst, nl0, cx, cy);
            else
              pu_role_poly<d, HET, ONEPUSH>(Ulo, Uhi, S, c_muh, c_la, cell_ok, dst, nl0, cx, cy);
          };
          if (role == 0)
            pu(std::integral_constant<int, 0>{});
          else if (role == 1)
            pu(std::integral_constant<int, 1>{});
          else if (role == 2)
            pu(std::integral_constant<int, 2>{});
          else
            {
              double Mdiag[8], Kphi[8];
              // the flag bytes of the cell's vertices BEFORE the pushes: an LDS read behind them waits for the whole queue
              unsigned anyflag = 0;
              if (cell_ok)
                {
#pragma unroll
                  for (int a = 0; a < 8; ++a)
                    anyflag |= s.flag[(ck + (a >> 2)) & 3][2 * (hb + (a & 1) + PH * ((a >> 1) & 1))];
                }
              if constexpr (OLDF)
                pp_role<HET, true>(Ulo, Uhi, S, lam, mu, cell_ok, dst, s.pp[2 + cp], s.pp[4], s.pp[np], s.pp[2 + np], nl0, cx, cy, Mdiag);
              else
                pp_role_poly<HET, RES>(Ulo, Uhi, S, lam, mu, cell_ok, dst, s.pp[2 + cp], s.pp[4], s.pp[np], s.pp[2 + np], nl0, cx, cy, Mdiag, Kphi,
                                       lap_lane);
              if constexpr (CLK == 2)
                if (lane == 0)
                  dbg[(size_t)blockIdx.x * 16 + 8] += (unsigned long long)(clock64() - tclk); // role 3: moments + pushes
              double avg = 0.0, patch = 0.0;
              if (cell_ok)
                {
                  // mean |diagonal| of the element matrix: deal.II's placeholder for a constrained row whose own
                  // diagonal entry vanishes.  Consumed only by constrained rows; with 0 < kappa <= 1 every
                  // g(q) >= kappa > 0, so the (u,u) diagonal cannot vanish.
                  double dsum = 0.0;
                  bool zero_diag = false;
#pragma unroll
                  for (int a = 0; a < 8; ++a)
                    {
                      const double dg = fabs(Mdiag[a]);
                      dsum += dg;
                      zero_diag = zero_diag || dg == 0.0;
                    }
                  const bool g_positive = S.kappa > 0.0 && S.kappa <= 1.0;
                  if (anyflag != 0 && (zero_diag || !g_positive))
                    {
                      double po[8], poo[8];
                      int cio = ci, cjo = cj;
                      asm volatile("" : "+v"(cio), "+v"(cjo)); // (addresses formed here, not hoisted out of the march and spilled)
#pragma unroll
                      for (int b = 0; b < 8; ++b)
                        {
                          if constexpr (OLDF)
                            {
                              const double *src = (b >> 2) ? Uhi : Ulo;
                              po[b] = src[4 * NPH + (b & 1) + PH * ((b >> 1) & 1)];
                              poo[b] = src[5 * NPH + (b & 1) + PH * ((b >> 1) & 1)];
                            }
                          else
                            {
                              // rare path (a constrained row next to a cell with a vanishing diagonal): straight from memory
                              const int id = cart_local_id(cv, cio + (b & 1), cjo + ((b >> 1) & 1), ck + (b >> 2));
                              po[b] = id >= 0 ? v.phi_old[id] : 0.0; // (-1: no node of this level's lattice, 3-D overlay)
                              poo[b] = id >= 0 ? v.phi_oldold[id] : 0.0;
                            }
                        }
                      // sum_{a,c} K_uu[(a,c),(a,c)] = sum_k (sum_c cA[c][k]) 2 sum_q w g mu(q_i) mu(q_j), mu = m_00 + m_11
                      double gsum = 0.0, gk[3] = {0.0, 0.0, 0.0};
#pragma unroll 1
                      for (int qz = 0; qz < 3; ++qz)
                        {
                          double wg[9];
                          cell_wg_plane(po, poo, S, qz, wg);
                          const double muz = c_g1.m[0][qz] + c_g1.m[2][qz];
#pragma unroll
                          for (int qy = 0; qy < 3; ++qy)
#pragma unroll
                            for (int qx = 0; qx < 3; ++qx)
                              {
                                const double w = wg[qx + 3 * qy];
                                const double mux = c_g1.m[0][qx] + c_g1.m[2][qx], muy = c_g1.m[0][qy] + c_g1.m[2][qy];
                                gsum += w;
                                gk[0] += w * muy * muz;
                                gk[1] += w * mux * muz;
                                gk[2] += w * mux * muy;
                              }
                        }
                      double usum = 0.0;
#pragma unroll
                      for (int k = 0; k < 3; ++k)
                        {
                          double ca = S.cA[0][k] + S.cA[1][k] + S.cA[2][k];
                          if constexpr (HET) // MatScal::cA with this cell's coefficients
                            {
                              ca = 0.0;
#pragma unroll
                              for (int c = 0; c < 3; ++c)
                                ca += (k == c ? lam + 2 * mu : mu) * S.ih[k] * S.ih[k];
                            }
                          usum += ca * 2.0 * gk[k];
                        }
                      avg = (dsum + usum) / 32.0;
                      patch = (gsum == 0.0) ? avg : 0.0;
                    }
                }
              // placeholder of a constrained row: sum_e (|K_e,aa| != 0 ? |K_e,aa| : mean |diag K_e|)
              static_for<8>([&](auto A) __attribute__((always_inline)) {
                constexpr int ax = 1 - (decltype(A)::value & 1), ay = 1 - ((decltype(A)::value >> 1) & 1), az = decltype(A)::value >> 2;
                const int hx = cx + ax, hy = cy + ay;
                if ((az == 0 ? dst.push_lo : dst.push_hi) && hx >= 1 && hx <= PN && hy >= 1 && hy <= PN)
                  {
                    // (the placeholder sums are read for rows with a constraint flag only: a cell none of whose vertices
                    // carries one -- nearly all of them -- has nothing to add)
                    if (anyflag != 0)
                      {
                        const double dg = fabs(Mdiag[ax + 2 * ay + 4 * az]);
                        double *ex = &s.ex[az == 0 ? cp : np][nl0 + ax + PN * ay][0];
                        lds_add(&ex[0], (dg != 0.0) ? dg : avg);
                        lds_add(&ex[1], patch);
                      }
                    if constexpr (RES)
                      if (cell_ok)
                        lds_add(&s.rs[az == 0 ? cp : np][nl0 + ax + PN * ay], Kphi[ax + 2 * ay + 4 * az]);
                  }
              });
              if constexpr (CLK == 2)
                if (lane == 0)
                  dbg[(size_t)blockIdx.x * 16 + 9] += (unsigned long long)(clock64() - tclk); // ... + placeholders
            }
          if constexpr (CLK == 2)
            {
              if (lane == 0)
                dbg[(size_t)blockIdx.x * 16 + 4 + role] += (unsigned long long)(clock64() - tclk);
            }
          stamp(1);
          if (PRIO)
            __builtin_amdgcn_s_setprio(3); // requests and copy-out: few instructions, long latencies -- issue them first
          lds_barrier();
          stamp(2);
          // next step's plane and row info: loads issued ahead of the copy-out stores, consumed after them
          if (ck + 1 < kB)
            {
              // ids first (possible table look-ups and their waits), then every request of the step
              bool okp, inp, okr;
              int dwp;
              unsigned np_, nr_;
              if (all_lex)
                {
                  np_ = plane_node(ck + 2, okp, inp, dwp, true);
                  nr_ = rows_node(ck + 1, okr, true);
                }
              else
                {
                  np_ = plane_node(ck + 2, okp, inp, dwp, false);
                  nr_ = rows_node(ck + 1, okr, false);
                  // a use on every path: a table look-up of cart_local_id left pending where the requests are skipped would be
                  // waited for with vmcnt(0) wherever its register is written next -- at the top of the next step, behind the stores
                  asm volatile("" ::"v"(np_), "v"(nr_));
                }
              dma_plane_at(np_, okp, inp, dwp, lo, ck + 2); // slot lo (plane ck) is dead once the entries of layer ck are done
              dma_rows_at(nr_, okr, ck + 1);
            }
          stamp(10);

          // ---- plane ck is complete: constraints as masks, then stream the rows out
          if (ck >= kA)
            {
              const bool regular = s.irregular[cp] == 0;
              const bool regular_or_permuted = s.incomplete[cp] == 0; // every row of the tile present with its 27 neighbours, in any order
              const bool masked = (s.anyflag[(ck - 1) & 3] | s.anyflag[ck & 3] | s.anyflag[(ck + 1) & 3]) != 0;
              double *pu_m1 = s.pu[cp], *pu_z0 = s.pu[2 + cp], *pu_p1 = s.pu[4];
              double *pp_m1 = s.pp[cp], *pp_z0 = s.pp[2 + cp], *pp_p1 = s.pp[4];
              const bool tile_full = (i0 + PN - 1) <= cv.o1[0] && (j0 + PN - 1) <= cv.o1[1];
              // all rows full, in lattice order, owned and free of constraint flags; the blocked copy-out below writes the
              // 7 rows of a y-line as ONE run, i.e. it also needs x-consecutive owned nodes to have consecutive local ids
              const bool fast = regular && !masked && tile_full && (NCOL != 3 || cv.owned_lex);
              // Rows in lattice order (bit 31 of the mask clear): the CSR slot of lattice offset o is its rank among the
              // offsets that exist (popcount of the row's neighbour mask below bit o); interior rows have all 27.
              if (NCOL == 3 && fast)
                {
                  // Blocked layout, interior plane without constraint flags (the common case): the 7 rows of a
                  // y-line of the tile are ONE contiguous run of 7*81 (phi,u) values, the same run of (u,phi) zeros
                  // and a run of 7*27 (phi,phi) values.  Thread <-> fixed positions of the runs (3 of the 567, the
                  // last one for t < 55; threads 64.. one of the 189); the run starts are wave-uniform (scalar base +
                  // 32-bit lane offset), the LDS strides per y-line are immediates.  (Pairs of elements per lane
                  // with 16-byte stores were measured slower: the runs are only 8-byte aligned.)
                  int tq = t;
                  asm volatile("" : "+v"(tq));
                  const bool act2 = tq < PN * 81 - 2 * NT4;
                  const int gp = tq - 64;
                  const bool actp = gp >= 0 && gp < PN * 27;
                  double *spu[3];
#pragma unroll
                  for (int q = 0; q < 3; ++q)
                    {
                      const int f = (q < 2 || act2) ? tq + NT4 * q : 0;
                      const int nx = f / 81, e = f - nx * 81;
                      const int o = e / 3, d = e - 3 * o;
                      const int oz = o / 9, o9 = o - 9 * oz;
                      spu[q] = (oz == 0 ? pu_m1 : (oz == 1 ? pu_z0 : pu_p1)) + (nx * 27 + o9 * 3 + d);
                    }
                  double *spp;
                  {
                    const int g = actp ? gp : 0;
                    const int nx = g / 27, o = g - nx * 27;
                    const int oz = o / 9, o9 = o - 9 * oz;
                    spp = (oz == 0 ? pp_m1 : (oz == 1 ? pp_z0 : pp_p1)) + (nx * 9 + o9);
                  }
                  double val[PN][4];
                  long long off0[PN];
#pragma unroll
                  for (int ny = 0; ny < PN; ++ny)
                    {
                      const long long o = s.off[cp][ny * PN]; // same address for every lane
                      off0[ny] = ((long long)__builtin_amdgcn_readfirstlane((int)(o >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)o);
                      val[ny][0] = spu[0][ny * (PN * 27)];
                      val[ny][1] = spu[1][ny * (PN * 27)];
                      val[ny][2] = act2 ? spu[2][ny * (PN * 27)] : 0.0;
                      val[ny][3] = actp ? spp[ny * (PN * 9)] : 0.0;
                    }
                  const unsigned uq = (unsigned)tq, up = (unsigned)(actp ? gp : 0);
#pragma unroll
                  for (int ny = 0; ny < PN; ++ny)
                    {
                      double *bpu = vals_pu + 3 * off0[ny], *bup = vals_up + 3 * off0[ny], *bpp = vals_pp + off0[ny];
                      bpu[uq] = val[ny][0];
                      bup[uq] = 0.0;
                      bpu[uq + NT4] = val[ny][1];
                      bup[uq + NT4] = 0.0;
                      spu[0][ny * (PN * 27)] = 0.0; // these slabs are the next planes' accumulators
                      spu[1][ny * (PN * 27)] = 0.0;
                      if (act2)
                        {
                          bpu[uq + 2 * NT4] = val[ny][2];
                          bup[uq + 2 * NT4] = 0.0;
                          spu[2][ny * (PN * 27)] = 0.0;
                        }
                      if (actp)
                        {
                          bpp[up] = val[ny][3];
                          spp[ny * (PN * 9)] = 0.0;
                        }
                    }
                  nst = 4 * PN;
                }
              else if (NCOL == 3 && regular_or_permuted && !masked && tile_full)
                {
                  // Round 6: rows that are COMPLETE (27 neighbours, no constraint flag near the plane) but whose CSR slots are a
                  // permutation of the lattice order and / or not contiguous from node to node -- every interior tile of a level
                  // lattice of the 3-D overlay, and of any box whose pattern the host bound in its own column order (deal.II
                  // numbers its dofs hierarchically: pfm_pattern_bind).  The thread <-> position mapping, the LDS reads and the
                  // store count of the blocked path above; only the destination of a value is looked up: row offset and mask
                  // from LDS, the slot of lattice offset o from CartView::row_perm -- all look-ups of a plane first, then its
                  // stores (a load behind a store waits for the store: vmcnt counts in order).  The generic path below costs
                  // the kernel 31 % when it is forced on every tile.
                  int tq = t;
                  asm volatile("" : "+v"(tq));
                  const bool act2 = tq < PN * 81 - 2 * NT4;
                  const int gp = tq - 64;
                  const bool actp = gp >= 0 && gp < PN * 27;
                  double *spu[3];
                  int nxq[3], oq[3], dq[3];
#pragma unroll
                  for (int q = 0; q < 3; ++q)
                    {
                      const int f = (q < 2 || act2) ? tq + NT4 * q : 0;
                      const int nx = f / 81, e = f - nx * 81;
                      const int o = e / 3, d = e - 3 * o;
                      const int oz = o / 9, o9 = o - 9 * oz;
                      spu[q] = (oz == 0 ? pu_m1 : (oz == 1 ? pu_z0 : pu_p1)) + (nx * 27 + o9 * 3 + d);
                      nxq[q] = nx, oq[q] = o, dq[q] = d;
                    }
                  double *spp;
                  int nxp, op;
                  {
                    const int g = actp ? gp : 0;
                    const int nx = g / 27, o = g - nx * 27;
                    const int oz = o / 9, o9 = o - 9 * oz;
                    spp = (oz == 0 ? pp_m1 : (oz == 1 ? pp_z0 : pp_p1)) + (nx * 9 + o9);
                    nxp = nx, op = o;
                  }
                  double val[PN][4];
                  long long dpu[PN][3], dpp[PN];
#pragma unroll
                  for (int ny = 0; ny < PN; ++ny)
                    {
#pragma unroll
                      for (int q = 0; q < 3; ++q)
                        {
                          const int row = ny * PN + nxq[q];
                          const long long off = s.off[cp][row];
                          int slot = oq[q];
                          if (((unsigned)s.deg[cp][row] >> 31) && (q < 2 || act2))
                            slot = cv.row_perm[off + slot];
                          dpu[ny][q] = 3 * off + 3 * slot + dq[q];
                        }
                      {
                        const int row = ny * PN + nxp;
                        const long long off = s.off[cp][row];
                        int slot = op;
                        if (((unsigned)s.deg[cp][row] >> 31) && actp)
                          slot = cv.row_perm[off + slot];
                        dpp[ny] = off + slot;
                      }
                      val[ny][0] = spu[0][ny * (PN * 27)];
                      val[ny][1] = spu[1][ny * (PN * 27)];
                      val[ny][2] = act2 ? spu[2][ny * (PN * 27)] : 0.0;
                      val[ny][3] = actp ? spp[ny * (PN * 9)] : 0.0;
                    }
#pragma unroll
                  for (int ny = 0; ny < PN; ++ny)
                    {
                      vals_pu[dpu[ny][0]] = val[ny][0];
                      vals_up[dpu[ny][0]] = 0.0;
                      vals_pu[dpu[ny][1]] = val[ny][1];
                      vals_up[dpu[ny][1]] = 0.0;
                      spu[0][ny * (PN * 27)] = 0.0; // these slabs are the next planes' accumulators
                      spu[1][ny * (PN * 27)] = 0.0;
                      if (act2)
                        {
                          vals_pu[dpu[ny][2]] = val[ny][2];
                          vals_up[dpu[ny][2]] = 0.0;
                          spu[2][ny * (PN * 27)] = 0.0;
                        }
                      if (actp)
                        {
                          vals_pp[dpp[ny]] = val[ny][3];
                          spp[ny * (PN * 9)] = 0.0;
                        }
                    }
                  nst = 4 * PN;
                }
              else if (t < 2 * 108)
                {
                  // thread <-> element fe_e of a row (0..80 (phi,u), 81..107 (phi,phi)), two rows at a time;
                  // recomputed per plane rather than kept live across the whole march
                  int tq = t;
                  asm volatile("" : "+v"(tq));
                  const int fe_e = tq % 108, fe_sub = tq / 108;
                  const bool fe_pp = fe_e >= 81;
                  const int fe_o = fe_pp ? fe_e - 81 : fe_e / 3, fe_d = fe_pp ? 3 : fe_e % 3; // lattice offset index, column component
                  const int fe_oz = fe_o / 9, fe_o9 = fe_o % 9;
                  const int fe_stride = fe_pp ? 9 : 27, fe_src = fe_pp ? fe_o9 : fe_o9 * 3 + fe_d;
                  const int fe_nbo = (fe_o9 % 3 - 1) + PH * (fe_o9 / 3 - 1); // halo offset of the neighbour node
                  double *src = (fe_oz == 0) ? (fe_pp ? pp_m1 : pu_m1) : (fe_oz == 1 ? (fe_pp ? pp_z0 : pu_z0) : (fe_pp ? pp_p1 : pu_p1));
                  src += fe_src;
                  if (fast)
                    {
                      // interleaved layout, interior plane: free of control flow so that the LDS reads of several
                      // rows are in flight together
                      const int fe_dst = 3 * 4 * 27 + fe_o * 4 + fe_d;
#pragma unroll 1
                      for (int n0 = fe_sub; n0 < NPN; n0 += 10)
                        {
                          long long offb[5];
                          double valb[5];
#pragma unroll
                          for (int i = 0; i < 5; ++i)
                            {
                              const int nl = min(n0 + 2 * i, NPN - 1); // clamped duplicates are dropped below
                              offb[i] = s.off[cp][nl];
                              valb[i] = src[nl * fe_stride];
                            }
#pragma unroll
                          for (int i = 0; i < 5; ++i)
                            if (n0 + 2 * i < NPN)
                              {
                                vals_uu[16 * offb[i] + fe_dst] = valb[i];
                                src[(n0 + 2 * i) * fe_stride] = 0.0; // these slabs are the next planes' accumulators
                              }
                        }
                    }
                  else
                    {
                      // constraint flags near the plane, partial tiles at the high faces, boundary rows with fewer
                      // than 27 neighbours
                      const unsigned *nfl = &s.flag[(ck + fe_oz - 1) & 3][2 * fe_nbo];
                      const unsigned below = (1u << fe_o) - 1u;
#pragma unroll 1
                      for (int nl = fe_sub; nl < NPN; nl += 2)
                        {
                          const long long off = s.off[cp][nl];
                          const unsigned nmask = (unsigned)s.deg[cp][nl];
                          double val = src[nl * fe_stride];
                          src[nl * fe_stride] = 0.0;
                          if (masked)
                            {
                              const int hn = (nl % PN + 1) + PH * (nl / PN + 1);
                              const unsigned row_flag = s.flag[ck & 3][2 * hn], nflag = nfl[2 * hn];
                              const bool rcon = (row_flag >> 3) & 1u;
                              if (fe_pp)
                                {
                                  if (rcon)
                                    val = (fe_o == 13) ? s.ex[cp][nl][0] : 0.0;
                                  else if ((nflag >> 3) & 1u)
                                    val = 0.0;
                                }
                              else if (rcon || ((nflag >> fe_d) & 1u))
                                val = 0.0; // constrained row (active set) or eliminated column
                            }
                          if (off >= 0 && ((nmask >> fe_o) & 1u)) // owned node, neighbour inside the mesh
                            {
                              int sl = __popc(nmask & below);
                              if (nmask >> 31) // row not in lattice order: permutation of the ranks
                                sl = cv.row_perm[off + sl];
                              asm volatile("" ::"v"(sl));
                              if constexpr (NCOL == 3)
                                {
                                  if (fe_pp)
                                    vals_pp[off + sl] = val;
                                  else
                                    {
                                      vals_pu[3 * off + sl * 3 + fe_d] = val;
                                      vals_up[3 * off + sl * 3 + fe_d] = 0.0; // any bijection onto the node's 3 rows
                                    }
                                }
                              else // interleaved layout: row (node, 3) holds [u_x u_y u_z phi] per neighbour slot
                                vals_uu[16 * off + (long long)3 * 4 * __popc(nmask & 0x7ffffffu) + sl * 4 + fe_d] = val;
                            }
                        }
                    }
                }
              stamp(8);
              lds_barrier(); // placeholders are read above, cleared below
              stamp(9);
              if (t < NPN)
                {
                  const int nl = t;
                  const double patch = s.ex[cp][nl][1];
                  s.ex[cp][nl][0] = 0.0;
                  s.ex[cp][nl][1] = 0.0;
                  const long long off = s.off[cp][nl];
                  if constexpr (RES)
                    {
                      // Phase-field rows of the residual from the (phi,phi) entries: with the unclamped phase field of the
                      // staggered scheme every term of cracks.cc:2412-2431 but -G_c/eps N_a is the matching term of
                      // cracks.cc:2370-2383 times phi_b, i.e.  R_phi = G_c/eps sum_q N_a JxW - K_phiphi phi  (UNMASKED entries),
                      // K_phiphi phi summed cell by cell in the order of the pushes (s.rs)
                      const double sum = s.rs[cp][nl];
                      s.rs[cp][nl] = 0.0;
                      if (off >= 0)
                        {
                          const int nx = nl % PN, ny = nl / PN, hn = (nx + 1) + PH * (ny + 1);
                          const int gi = i0 + nx, gj = j0 + ny;
                          const int ncell = ((gi > 0) + (gi < cv.NX - 1)) * ((gj > 0) + (gj < cv.NY - 1)) * ((ck > 0) + (ck < cv.NZ - 1));
                          const double mass = S.gc_eps * (S.vol * 0.125) * (double)ncell;
                          const bool con = (s.flag[ck & 3][2 * hn] >> 3) & 1u;
                          // (a look-up is waited for inside its arm: cart_local_id_sync)
                          const int row = all_lex ? (int)lex_id(gi, gj, ck) : cart_local_id_sync(cv, gi, gj, ck);
                          const long long di = (v.layout == PFM_LAYOUT_INTERLEAVED) ? (long long)row * 4 + 3 : (long long)v.n_owned * 3 + row;
                          res_pde[di] = con ? 0.0 : mass - sum;
                        }
                    }
                  if (masked && off >= 0 && patch != 0.0)
                    {
                      // constrained displacement rows whose element diagonal vanished in some cell
                      const unsigned row_flag = s.flag[ck & 3][2 * ((nl % PN + 1) + PH * (nl / PN + 1))];
                      const unsigned nmask = (unsigned)s.deg[cp][nl];
                      const int deg = __popc(nmask & 0x7ffffffu);
                      int sself = __popc(nmask & ((1u << 13) - 1u));
                      if (nmask >> 31)
                        sself = cv.row_perm[off + sself];
                      asm volatile("" ::"v"(sself)); // consumed on every path (see the requests: no load may stay pending)
                      for (int c = 0; c < 3; ++c)
                        if ((row_flag >> c) & 1u)
                          {
                            const long long at = (long long)NCOL * NCOL * off + (long long)c * NCOL * deg + sself * NCOL + c;
                            if (cv.patch_count) // the (u,u) kernel may still be writing: deferred (launch_cart_apply_patches)
                              {
                                const int e = atomicAdd(cv.patch_count, 1);
                                if (e < cv.patch_cap)
                                  {
                                    cv.patch_idx[e] = at;
                                    cv.patch_val[e] = patch;
                                  }
                              }
                            else
                              vals_uu[at] += patch;
                          }
                    }
                }
            }
        }
      stamp(3);
    }
  } // namespace

  int upload_mat_scal(const pfm_params &p, const CartView &cv, void *d_scal, hipStream_t s)
  {
    static_assert(sizeof(MatScal) <= PFM_SCAL_BYTES, "scalar buffer too small");
    const MatScal Sh = make_mat_scal(p, cv);
    // pageable source: the runtime stages it before returning, Sh may go out of scope
    return hipMemcpyAsync(d_scal, &Sh, sizeof(MatScal), hipMemcpyHostToDevice, s) == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }

  int launch_cart_phi4(const DevView &v, const CartView &cv, const pfm_params &p, double *const *d_values, hipStream_t s,
                       const void *d_scal, double *res_pde)
  {
    const MatScal *S = static_cast<const MatScal *>(d_scal);
    if (v.dim != 3)
      return PFM_ERR_UNSUPPORTED;
    int rc = ensure_g1();
    if (rc)
      return rc;
    const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1, OWZ = cv.o1[2] - cv.o0[2] + 1;
    const int ntx = (OWX + PN - 1) / PN, nty = (OWY + PN - 1) / PN;
    // z-chunks: one extra cell layer per chunk is recomputed; keep that below ~4 % while filling the chip
    static const int zc_force = getenv("PFM_PHI_ZC") ? atoi(getenv("PFM_PHI_ZC")) : 0; // tuning only
    // (round 6: chunks of up to 48 planes as in k_cart_uu3 -- at 216^3 the model picks 31 = 217 / 7, seven equal chunks per
    // column instead of ten of 22 with a short last one: 10.7 -> 10.4 ms per assembly, profiles/r06/zc_scan.txt)
    const int zc_abs = zc_force > 0 ? zc_force : choose_zchunk((long long)ntx * nty, OWZ, 6, 48, 2);
    const int nch = (OWZ + zc_abs - 1) / zc_abs;
    static const bool no_prio = getenv("PFM_NO_PRIO") != nullptr; // A/B runs only
    const int zc = no_prio ? -zc_abs : zc_abs;
    const unsigned nb = (unsigned)(ntx * nty * nch);
    const bool il = v.layout == PFM_LAYOUT_INTERLEAVED, het = cv.cell_lam != nullptr, res = res_pde != nullptr;
    const dim3 grid(xcd_grid(nb)), block(NT4);
  // the general form (q-point loops, old phase fields in the nodal ring) only where the scheme needs it: clamped phase field
  // (monolithic) or penalisation term (same rules as make_mat_scal)
  const MatScal Sh = make_mat_scal(p, cv);
  const bool oldf = Sh.gamma_fac != 0.0 || Sh.monolithic;
#define PFM_PHI4_(NC, HETV, RESV, OLDV)                                                                                           \
  hipLaunchKernelGGL((k_cart_phi4<NC, 0, HETV, RESV, OLDV>), grid, block, 0, s, v, cv, S, (NC == 3 ? d_values[2] : nullptr),      \
                     (NC == 3 ? d_values[3] : nullptr), d_values[0], (NC == 3 ? d_values[1] : nullptr), zc, nullptr, res_pde)
#define PFM_PHI4(NC, HETV, RESV)                                                                                                  \
  do                                                                                                                              \
    {                                                                                                                             \
      if (oldf && !(RESV))                                                                                                        \
        PFM_PHI4_(NC, HETV, false, true);                                                                                         \
      else                                                                                                                        \
        PFM_PHI4_(NC, HETV, RESV, false);                                                                                         \
    }                                                                                                                             \
  while (0)
    if (il)
      {
        if (het)
          PFM_PHI4(4, true, false); // heterogeneous material: the residual kernel runs
        else
          {
            if (res)
              PFM_PHI4(4, false, true);
            else
              PFM_PHI4(4, false, false);
          }
      }
    else if (het)
      PFM_PHI4(3, true, false); // heterogeneous material: the residual kernel runs
    else if (getenv("PFM_PHI_CLK") && !oldf) // profiling only
      {
        static unsigned long long *d_dbg = nullptr;
        const size_t nd = (size_t)xcd_grid(nb) * 16;
        if (!d_dbg && hipMalloc((void **)&d_dbg, nd * sizeof(unsigned long long)) != hipSuccess)
          return PFM_ERR_HIP;
        (void)hipMemsetAsync(d_dbg, 0, nd * sizeof(unsigned long long), s);
        if (atoi(getenv("PFM_PHI_CLK")) == 2 && res)
          hipLaunchKernelGGL((k_cart_phi4<3, 2, false, true>), dim3(xcd_grid(nb)), dim3(NT4), 0, s, v, cv, S, d_values[2], d_values[3],
                             d_values[0], d_values[1], zc, d_dbg, res_pde);
        else if (atoi(getenv("PFM_PHI_CLK")) == 2)
          hipLaunchKernelGGL((k_cart_phi4<3, 2>), dim3(xcd_grid(nb)), dim3(NT4), 0, s, v, cv, S, d_values[2], d_values[3],
                             d_values[0], d_values[1], zc, d_dbg, nullptr);
        else if (res)
          hipLaunchKernelGGL((k_cart_phi4<3, 1, false, true>), dim3(xcd_grid(nb)), dim3(NT4), 0, s, v, cv, S, d_values[2], d_values[3],
                             d_values[0], d_values[1], zc, d_dbg, res_pde);
        else
          hipLaunchKernelGGL((k_cart_phi4<3, 1>), dim3(xcd_grid(nb)), dim3(NT4), 0, s, v, cv, S, d_values[2], d_values[3],
                             d_values[0], d_values[1], zc, d_dbg, nullptr);
        std::vector<unsigned long long> hall(nd);
        (void)hipMemcpy(hall.data(), d_dbg, nd * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long h[16] = {};
        for (size_t i = 0; i < nd; ++i)
          h[i % 16] += hall[i];
        const char *names[4] = {"load+barrier", "entries+push", "barrier", "copy-out"};
        fprintf(stderr, "[k_cart_phi4 phase clock, wave 0, cycles per workgroup (%d planes)]", zc_abs);
        for (int i = 0; i < 4; ++i)
          fprintf(stderr, " %s=%.0f", names[i], (double)h[i] / nb);
        for (int i = 0; i < 4; ++i)
          fprintf(stderr, " role%d=%.0f", i, (double)h[4 + i] / nb);
        fprintf(stderr, " request-next=%.0f copy-loop=%.0f copy-barrier=%.0f", (double)h[10] / nb, (double)h[8] / nb, (double)h[9] / nb);
        if (atoi(getenv("PFM_PHI_CLK")) == 2)
          fprintf(stderr, " | role 3 up to: moments+pushes=%.0f placeholders=%.0f", (double)h[8] / nb, (double)h[9] / nb);
        fprintf(stderr, "\n");
      }
    else if (res)
      PFM_PHI4(3, false, true);
    else
      PFM_PHI4(3, false, false);
#undef PFM_PHI4
#undef PFM_PHI4_
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }
  bool cart_matrix_supported(int dim) { return dim == 2 || dim == 3; }

  // Jacobian of a cartesian box: (u,u) rows first, k_cart_phi4 patches constrained (u,u) diagonals afterwards
  // (same stream) and clears the structurally zero (u,phi) block (cracks.cc:2333-2337) along with its (phi,u) stores
  // s_phi != s: the two kernels next to each other (the caller has forked s_phi off s and joins them; CartView::patch_*
  // must be set: the phase-field kernel defers its (u,u) patches)
  int launch_cart_matrix(const DevView &v, const CartView &cv_in, const pfm_params &p, double *const *d_values, hipStream_t s,
                         void *d_scal, double *res_pde, int phase, hipStream_t s_phi)
  {
    if (v.dim != 3)
      return PFM_ERR_UNSUPPORTED;
    // phase 1 / 2 of an overlapped assembly: the (u,u) kernel is cut into interior and boundary tiles (CartView::tile_sel),
    // the phase-field kernel follows completely in phase 2 -- it patches (u,u) diagonals of constrained rows and must see
    // every (u,u) tile written, and the ghost import (~0.1 ms) is hidden behind the interior (u,u) tiles alone
    CartView cv = cv_in;
    cv.tile_sel = phase;
    // next to the phase-field kernel: the same LDS allocation as that kernel (64 granules of 1280 B; 79,472 B are 63), so
    // that a slot freed by either kernel takes a workgroup of either
    int rc = launch_cart_uu3(v, cv, p, d_values[0], s, d_scal, res_pde, s_phi != s ? 64 * 1280 : 0);
    cv.tile_sel = 0;
    if (rc || phase == 1)
      return rc;
    if (s_phi != s && !cv.patch_count)
      return PFM_ERR_BAD_ARG; // concurrent kernels need the deferred patch list
    return launch_cart_phi4(v, cv, p, d_values, s_phi, d_scal, res_pde);
  }

  namespace
  {
    __global__ void k_cart_apply_patches(double *__restrict__ vals_uu, const long long *__restrict__ idx, const double *__restrict__ val,
                                         const int *__restrict__ count, int cap, int *__restrict__ status)
    {
      // the list holds one entry per flagged displacement dof at most (capacity from pfm_set_constraints); should that
      // invariant ever break, the dropped patches are reported instead of lost silently
      if (*count > cap && blockIdx.x == 0 && threadIdx.x == 0)
        atomicMax(status, (int)PFM_ERR_INTERNAL);
      const int n = min(*count, cap);
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        vals_uu[idx[i]] += val[i]; // one entry per matrix value at most: no two threads share an address
    }
  } // namespace

  int launch_cart_apply_patches(const CartView &cv, double *vals_uu, hipStream_t s, int *status)
  {
    if (!cv.patch_count)
      return PFM_OK;
    hipLaunchKernelGGL(k_cart_apply_patches, dim3(64), dim3(256), 0, s, vals_uu, cv.patch_idx, cv.patch_val, cv.patch_count, cv.patch_cap,
                       status);
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }
} // namespace pfm
