// pfm_cart_uu6.hip — (u,u) block of the Jacobian (3-D, cracks.cc:2340-2368), z-marching "push" kernel (round 4).
//
// Why another (u,u) kernel.  k_cart_uu3 (pfm_cart_uu3.hip) evaluates every cell 2.8 times (tile of 8 x 4 nodes in one
// plane, both cell layers around it), stores 63 tables per cell in LDS and lets every node PULL its 243 values from
// them through ten barrier-separated, latency-bound phases: 12.9k vector instructions per node, 45 % of them FP64
// (profiles/r03).  The element matrix itself is small -- 576 entries of 2-3 FMAs -- and the phase-field kernel showed
// that long register-resident role phases keep the FP64 pipe busy.  This kernel is that formulation for the (u,u) block:
//
//   * a workgroup owns the rows of ONE row component C of a 7 x 7 column of nodes over a chunk of z-planes; its 3 waves
//     are the 3 column components D (one role = one (C,D) block of the 3 x 3 component blocks), lane <-> one of the
//     8 x 8 cells of the current layer touching those nodes;
//   * marching up in z, every lane evaluates w*g at the 27 q-points of ITS cell (cracks.cc:2262-2306: g = (1-kappa)
//     pf_extra^2 + kappa), reduces them to the moment tables of its role -- A^x, A^y, A^z (27 numbers) for D = C, the
//     pair table T^{CD} (12 numbers) otherwise; header of pfm_cart.hip -- and forms the 64 entries K[(a,C),(b,D)] of
//     the cell from them in registers (3 resp. 2 multiply-adds each, signs and table indices compile-time);
//   * the entries are PUSHED into the LDS-staged rows of the cell's vertices (ds_add_f64): the 32 entries of the lower
//     vertices complete the rows of node plane k, which is then masked (constraints) and streamed out as whole rows;
//     the 32 entries of the upper vertices start plane k + 1 in the same buffer afterwards.  Only ONE plane of rows is
//     staged (the tables stay in registers across the copy-out instead of partial rows in LDS): 31 KB per workgroup,
//     four workgroups = 12 waves per CU.  The first push into a staged value is a plain store, so nothing is zeroed;
//   * every staged value receives its pushes from one wave only (role D owns the entries [.][.][D]) in program order:
//     no atomics between waves, the summation order is fixed, results are bitwise reproducible;
//   * RES: the displacement rows of the residual come out of the same entries, R_u = (alpha_B-1) p sum_q pfx^2 dN/dx_C
//     JxW - K_uu u for the unsplit law (pfm_cart_uu3.hip, uu_row_component), summed per vertex before the push.
//
// Cell evaluations per node: 64/49 = 1.31 (+ one layer per z-chunk) instead of 2.8.  Constraints are masks in the
// copy-out exactly as in k_cart_uu3; constrained rows keep their own diagonal (deal.II: sum of |K_e,aa|, positive here),
// the mean-|diagonal| patch for vanishing diagonals stays with k_cart_phi4.  Heterogeneous material and the cut launches
// of pfm_assemble_overlapped stay on k_cart_uu3.
#include "pfm_internal.h"
#include "pfm_cart_common.h"

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace pfm
{
  namespace
  {
    constexpr int UT = 8, UN = UT - 1, UH = UT + 1; // cells, owned nodes, halo nodes per tile edge
    constexpr int NUN = UN * UN, NUH = UH * UH;     // 49 owned nodes, 81 halo nodes per plane
    constexpr int NT6 = 3 * UT * UT;                // 3 roles x 64 cells
    constexpr int ROW6 = 81;                        // staged row: 27 slots x 3 column components
    constexpr int LINE6 = UN * ROW6;                // the 7 rows of a y-line

    struct Lds6
    {
      double P[2][2][NUH]; // nodal ring: [slot = plane & 1][0: phi_old or the combined old field, 1: phi_oldold][halo node]
      double U[2][3][NUH]; // RES: displacements of the halo nodes, same ring
      double stage[NUN * ROW6]; // rows (node, C) of the current plane: [node][slot o27][D]
      double rpart[3][NUN];     // RES: sum_b K[(a,C),(b,D)] u_(b,D) (- pressure part, D = C) per role D and node
      long long off[2][NUN];    // node-graph offset of the row, -1 = not an owned node of this tile
      unsigned mask[2][NUN];    // neighbour mask of the row
      unsigned char flag[4][NUH];
      int anyflag[4][2]; // per plane & 3: some halo node carries a displacement flag (one entry per storing wave)
      int irregular[2];  // per plane & 1: some row is not a full owned lattice row
    };
    static_assert(sizeof(Lds6) <= 40960, "four workgroups per CU: 32 LDS granules of 1280 B each");

    // 1-D Gauss(3) data on [0,1] as compile-time constants (pfm_cart_common.h: make_g1 -- the same numbers).  Read from the
    // __constant__ table they are 18 + 9 doubles of scalar registers that the march cannot keep (it spilled them into
    // vector lanes: v_readlane / v_writelane in the inner phases); as literals they are rematerialised where used.
    constexpr double GQ6 = 0.7745966692414834; // sqrt(3/5)
    __host__ __device__ constexpr double g6_n1(int q) { return q == 0 ? 0.5 - 0.5 * GQ6 : (q == 1 ? 0.5 : 0.5 + 0.5 * GQ6); }
    __host__ __device__ constexpr double g6_n(int al, int q) { return al ? g6_n1(q) : 1.0 - g6_n1(q); }
    __host__ __device__ constexpr double g6_m(int g, int q)
    {
      return g == 0 ? g6_n(0, q) * g6_n(0, q) : (g == 1 ? g6_n(0, q) * g6_n(1, q) : g6_n(1, q) * g6_n(1, q));
    }
    __host__ __device__ constexpr double g6_w(int q) { return q == 1 ? 8.0 / 18.0 : 5.0 / 18.0; }

    __host__ __device__ constexpr int pair6(int lo, int hi) { return lo == 0 ? (hi == 1 ? 0 : 1) : 2; } // (0,1) (0,2) (1,2)
    __host__ __device__ constexpr int idxA6(int k, int gi, int gj) { return k * 9 + gi * 3 + gj; }
    __host__ __device__ constexpr int idxT6(int al, int be, int g) { return al * 6 + be * 3 + g; }

    // w g / vol at the 27 q-points of one cell from the 8 vertex values of the old phase field(s): z first, then y, then one
    // FMA per point along x (cracks.cc:2262-2277, 2306); the cell volume is folded into the constants of the roles.
    // LIN: one combined field (staggered scheme), clamped to [lo, hi] ([0, 1], or unbounded with use_old_timestep_pf)
    struct Wc6
    {
      double lo, hi, omk, kap, ivol;
    };
    template <bool LIN>
    __device__ __forceinline__ void cell_w27(const double (&pa)[8], const double (&pb)[8], const MatScal &S, const Wc6 &K, bool cell_ok,
                                             double (&w27)[27])
    {
      if constexpr (LIN)
        {
          double dz[4];
#pragma unroll
          for (int v = 0; v < 4; ++v)
            dz[v] = pa[v + 4] - pa[v];
          static_for<3>([&](auto Qz) __attribute__((always_inline)) {
            constexpr int qz = decltype(Qz)::value;
            constexpr double nz1 = g6_n1(qz);
            double z[4];
#pragma unroll
            for (int v = 0; v < 4; ++v)
              z[v] = fma(nz1, dz[v], pa[v]);
            const double dy0 = z[2] - z[0], dy1 = z[3] - z[1];
            static_for<3>([&](auto Qy) __attribute__((always_inline)) {
              constexpr int qy = decltype(Qy)::value;
              constexpr double ny1 = g6_n1(qy);
              const double y0 = fma(ny1, dy0, z[0]), y1 = fma(ny1, dy1, z[1]);
              const double dx = y1 - y0;
              static_for<3>([&](auto Qx) __attribute__((always_inline)) {
                constexpr int qx = decltype(Qx)::value;
                constexpr double nx1 = g6_n1(qx), W = g6_w(qx) * g6_w(qy) * g6_w(qz);
                double pfx = fma(nx1, dx, y0);
                pfx = fmin(fmax(pfx, K.lo), K.hi);
                const double g = fma(K.omk * pfx, pfx, K.kap);
                w27[qx + 3 * qy + 9 * qz] = W * g;
              });
            });
          });
        }
      else
        {
#pragma unroll
          for (int qz = 0; qz < 3; ++qz)
            {
              double wg[9];
              cell_wg_plane(pa, pb, S, qz, wg);
#pragma unroll
              for (int q = 0; q < 9; ++q)
                w27[q + 9 * qz] = cell_ok ? wg[q] * K.ivol : 0.0;
            }
        }
    }

    // A^k[g_i][g_j] = sum_q w g m_{g_i}(q_i) m_{g_j}(q_j), (i, j) = the two other axes ascending: 27 numbers
    __device__ __forceinline__ void tables_A(const double (&w27)[27], double (&A)[27])
    {
      static_for<3>([&](auto Cc) __attribute__((always_inline)) {
        constexpr int c = decltype(Cc)::value;
        constexpr int sc = (c == 0) ? 1 : (c == 1) ? 3 : 9;
        constexpr int si = (c == 0) ? 3 : 1;
        constexpr int sj = (c == 2) ? 3 : 9;
        double s9[3][3]; // [qj][qi]
#pragma unroll
        for (int qj = 0; qj < 3; ++qj)
#pragma unroll
          for (int qi = 0; qi < 3; ++qi)
            {
              const int q0 = qi * si + qj * sj;
              s9[qj][qi] = (w27[q0] + w27[q0 + sc]) + w27[q0 + 2 * sc];
            }
#pragma unroll
        for (int gi = 0; gi < 3; ++gi)
          {
            double tq[3];
#pragma unroll
            for (int qj = 0; qj < 3; ++qj)
              tq[qj] = s9[qj][0] * g6_m(gi, 0) + s9[qj][1] * g6_m(gi, 1) + s9[qj][2] * g6_m(gi, 2);
#pragma unroll
            for (int gj = 0; gj < 3; ++gj)
              A[idxA6(c, gi, gj)] = tq[0] * g6_m(gj, 0) + tq[1] * g6_m(gj, 1) + tq[2] * g6_m(gj, 2);
          }
      });
    }

    // T^p[al][be][g] = sum_q w g n_al(q_lo) n_be(q_hi) m_g(q_e) for the axis pair p = (lo, hi), e = the third axis
    template <int P>
    __device__ __forceinline__ void tables_T(const double (&w27)[27], double (&T)[12])
    {
      constexpr int lo = (P == 2) ? 1 : 0, hi = (P == 0) ? 1 : 2, e = 3 - lo - hi;
      constexpr int slo = (lo == 0) ? 1 : 3, shi = (hi == 1) ? 3 : 9, se = (e == 0) ? 1 : (e == 1) ? 3 : 9;
#pragma unroll
      for (int al = 0; al < 2; ++al)
        {
          const double na0 = g6_n(al, 0), na1 = g6_n(al, 1), na2 = g6_n(al, 2);
          double t1[3][3]; // [q_e][q_hi]
#pragma unroll
          for (int qe = 0; qe < 3; ++qe)
#pragma unroll
            for (int qh = 0; qh < 3; ++qh)
              {
                const int q0 = qh * shi + qe * se;
                t1[qe][qh] = (w27[q0] * na0 + w27[q0 + slo] * na1) + w27[q0 + 2 * slo] * na2;
              }
#pragma unroll
          for (int be = 0; be < 2; ++be)
            {
              double t2[3];
#pragma unroll
              for (int qe = 0; qe < 3; ++qe)
                t2[qe] = t1[qe][0] * g6_n(be, 0) + t1[qe][1] * g6_n(be, 1) + t1[qe][2] * g6_n(be, 2);
#pragma unroll
              for (int g = 0; g < 3; ++g)
                T[idxT6(al, be, g)] = t2[0] * g6_m(g, 0) + t2[1] * g6_m(g, 1) + t2[2] * g6_m(g, 2);
            }
        }
    }

    // element entries K[(a,C),(b,C)] = sum_k sg(a_k) sg(b_k) cA[C][k] A^k[g_i][g_j]   (cA = (k == C ? lambda + 2 mu : mu) vol / h_k^2):
    // sg(a_k) sg(b_k) = -1 exactly where g_k = a_k + b_k = 1, so the 64 entries of the cell take only the 27 values
    // E[g_x + 3 g_y + 9 g_z] -- formed once per cell, then pushed as they are
    __device__ __forceinline__ void entries_A(const double (&A)[27], const double (&cA)[3], double (&E)[27])
    {
#pragma unroll
      for (int gz = 0; gz < 3; ++gz)
#pragma unroll
        for (int gy = 0; gy < 3; ++gy)
#pragma unroll
          for (int gx = 0; gx < 3; ++gx)
            {
              double r = ((gx == 1) ? -cA[0] : cA[0]) * A[idxA6(0, gy, gz)];
              r = fma((gy == 1) ? -cA[1] : cA[1], A[idxA6(1, gx, gz)], r);
              r = fma((gz == 1) ? -cA[2] : cA[2], A[idxA6(2, gx, gy)], r);
              E[gx + 3 * gy + 9 * gz] = r;
            }
    }
    template <int A_, int B_>
    __device__ __forceinline__ double entry_A(const double (&E)[27])
    {
      return E[((A_ & 1) + (B_ & 1)) + 3 * (((A_ >> 1) & 1) + ((B_ >> 1) & 1)) + 9 * ((A_ >> 2) + (B_ >> 2))];
    }
    // element entry K[(a,C),(b,D)], C != D, pair p = (lo, hi) = (min, max): with X = T[b_lo][a_hi][g_e] and
    // Y = T[a_lo][b_hi][g_e] it is sg(a_lo) sg(b_hi) k1 X + sg(a_hi) sg(b_lo) k2 Y, (k1, k2) = (lambda, mu) / (h_lo h_hi)
    // for C < D and (mu, lambda) / (h_lo h_hi) for C > D  (pfm_cart_uu3.hip: uu_acc_visit)
    template <int P, int A_, int B_>
    __device__ __forceinline__ double entry_T(const double (&T)[12], double k1, double k2)
    {
      constexpr int lo = (P == 2) ? 1 : 0, hi = (P == 0) ? 1 : 2, e = 3 - lo - hi;
      constexpr int a[3] = {A_ & 1, (A_ >> 1) & 1, A_ >> 2}, b[3] = {B_ & 1, (B_ >> 1) & 1, B_ >> 2};
      constexpr int ge = a[e] + b[e];
      constexpr bool s1 = (a[lo] == b[hi]), s2 = (a[hi] == b[lo]); // sg(x) sg(y) = +1 iff the bits agree
      const double r = (s1 ? k1 : -k1) * T[idxT6(b[lo], a[hi], ge)];
      return fma(s2 ? k2 : -k2, T[idxT6(a[lo], b[hi], ge)], r);
    }
    // sg(a_C) sum_q w g n_{a_i}(q_i) n_{a_j}(q_j) for the pressure part of the residual row (a, C): the sum of the four
    // moments A^C[a_i + {0,1}][a_j + {0,1}] (n_0 + n_1 = 1)
    template <int CC, int A_>
    __device__ __forceinline__ double pres_moment(const double (&A)[27], double kv4)
    {
      constexpr int a[3] = {A_ & 1, (A_ >> 1) & 1, A_ >> 2};
      constexpr int i = (CC == 0) ? 1 : 0, j = (CC == 2) ? 1 : 2;
      const double s4 = (A[idxA6(CC, a[i], a[j])] + A[idxA6(CC, a[i] + 1, a[j])]) + (A[idxA6(CC, a[i], a[j] + 1)] + A[idxA6(CC, a[i] + 1, a[j] + 1)]);
      const double mom = s4 - kv4; // w pfx^2 = (w g - kappa w) / (1 - kappa), sum_q w n n = 1/4: kv4 = kappa / 4
      return a[CC] ? mom : -mom;
    }

    // Is (vertex i of the pass, b) the FIRST push of its pass into its staged value?  Pass 1 = upper vertices (a_z = 1),
    // after the copy-out: the values of the slots oz <= 0 of the next plane start there; pass 0 = lower vertices
    // (a_z = 0): the slots oz = +1 start there, the slots oz = 0 continue.  Within a pass the in-plane vertices are visited
    // in the order (1,1), (0,1), (1,0), (0,0): the order in which a lexicographic cell loop reaches the node.
    __host__ __device__ constexpr int vord_ax(int i) { return 1 - (i & 1); }
    __host__ __device__ constexpr int vord_ay(int i) { return 1 - ((i >> 1) & 1); }
    __host__ __device__ constexpr bool first_push(int pass_az, int i, int bx, int by, int bz)
    {
      const int ox = bx - vord_ax(i), oy = by - vord_ay(i), oz = bz - pass_az;
      if (pass_az == 0 && oz == 0)
        return false;
      for (int j = 0; j < i; ++j)
        {
          const int cx = vord_ax(j) + ox, cy = vord_ay(j) + oy;
          if (cx >= 0 && cx <= 1 && cy >= 0 && cy <= 1)
            return false; // an earlier vertex of this pass reaches the same slot
        }
      return true;
    }

    // one pass of pushes of one role: AZ = 0 lower vertices (completes the plane), AZ = 1 upper vertices (starts the next).
    // RES: ku = sum_b K[(a,C),(b,D)] u_(b,D) - pressure part of the vertex.  For the upper vertices the part of the lower
    // trial vertices and the pressure part are formed BEFORE the copy-out (kpart, res_prepare below) -- four numbers live
    // across the copy-out instead of the eight displacements and four pressure moments; the displacements of the upper
    // plane are re-read from the ring afterwards.
    template <int KIND /* 0: D == C (A tables), 1..3: pair p = KIND - 1 */, int AZ, bool RES>
    __device__ __forceinline__ void push_pass(const double (&tab)[KIND == 0 ? 27 : 12], const double (&ud)[8], const double (&cst)[3],
                                              const double (&kinit)[4], double *__restrict__ stage_d, double *__restrict__ rpart_d, int cx, int cy)
    {
      static_for<4>([&](auto Ii) __attribute__((always_inline)) {
        constexpr int i = decltype(Ii)::value;
        constexpr int ax = vord_ax(i), ay = vord_ay(i);
        constexpr int A_ = ax + 2 * ay + 4 * AZ;
        const int hx = cx + ax, hy = cy + ay;
        if (hx >= 1 && hx <= UN && hy >= 1 && hy <= UN)
          {
            const int nl = (hx - 1) + UN * (hy - 1);
            double *row = stage_d + nl * ROW6;
            double ku = kinit[i];
            static_for<8>([&](auto Bb) __attribute__((always_inline)) {
              constexpr int B_ = decltype(Bb)::value;
              constexpr int bx = B_ & 1, by = (B_ >> 1) & 1, bz = B_ >> 2;
              constexpr int o27 = (bx - ax + 1) + 3 * (by - ay + 1) + 9 * (bz - AZ + 1);
              double e;
              if constexpr (KIND == 0)
                e = entry_A<A_, B_>(tab);
              else
                e = entry_T<KIND - 1, A_, B_>(tab, cst[0], cst[1]);
              if constexpr (RES && (AZ == 0 || bz == 1))
                ku = fma(e, ud[B_], ku);
              if constexpr (first_push(AZ, i, bx, by, bz))
                row[o27 * 3] = e;
              else
                lds_add(&row[o27 * 3], e);
            });
            if constexpr (RES)
              {
                double *rp = rpart_d + nl;
                if constexpr (AZ == 1 && i == 0)
                  *rp = ku;
                else
                  lds_add(rp, ku);
              }
          }
      });
    }
    // RES: start values of the residual sums of the 4 lower (kinit0) and the 4 upper (kpart) vertices in the order of
    // push_pass: minus the pressure part (pres, role D = C only), for the upper vertices plus the part of the lower trial
    // vertices (the displacements of the lower plane leave the ring during the copy-out)
    template <int KIND>
    __device__ __forceinline__ void res_prepare(const double (&tab)[KIND == 0 ? 27 : 12], const double (&ud)[8], const double (&cst)[3],
                                                const double (&pres)[8], double (&kinit0)[4], double (&kpart)[4])
    {
      static_for<4>([&](auto Ii) __attribute__((always_inline)) {
        constexpr int i = decltype(Ii)::value;
        constexpr int ax = vord_ax(i), ay = vord_ay(i);
        constexpr int A0 = ax + 2 * ay, A1 = A0 + 4;
        kinit0[i] = (KIND == 0) ? -pres[A0] : 0.0;
        double ku = (KIND == 0) ? -pres[A1] : 0.0;
        static_for<4>([&](auto Bb) __attribute__((always_inline)) {
          constexpr int B_ = decltype(Bb)::value; // lower trial vertices
          double e;
          if constexpr (KIND == 0)
            e = entry_A<A1, B_>(tab);
          else
            e = entry_T<KIND - 1, A1, B_>(tab, cst[0], cst[1]);
          ku = fma(e, ud[B_], ku);
        });
        kpart[i] = ku;
      });
    }

    // =====================================================================================
    struct Geo6 // what the march needs of the tile, wave-uniform
    {
      int i0, j0, kA, kB, C, role, abl;
      int cx, cy, hb;
      bool col_ok;
    };

    // The march of one role.  KIND is a template parameter of the WHOLE loop (tables, both push passes, the copy-out in
    // between): the role's tables are a local array of exactly its size whose elements live in registers across the
    // copy-out.  (A table array shared by the kinds and selected by run-time branches around each phase ended up in
    // scratch memory, and every reload waited with vmcnt(0) for the copy-out's global stores: 11.9 ms per launch.)
    template <int KIND /* 0: D == C, 1..3: pair p = KIND - 1 */, int CC /* KIND 0: the row component C = D */, int NCOL, bool RES, bool LIN, bool CLK>
    __device__ __forceinline__ void march6(Lds6 &s, const DevView &v, const CartView &cv, const MatScal &S, const Geo6 &G, double *__restrict__ vals,
                                           double *__restrict__ res_pde, unsigned long long *__restrict__ dbg)
    {
      constexpr int NTAB = (KIND == 0) ? 27 : 12;
      const int t = threadIdx.x;
      long long tclk = 0;
      auto stamp = [&](int phase) __attribute__((always_inline)) { // profiling only: cycles per phase, lane 0 of every wave
        if constexpr (CLK)
          {
            const long long now = clock64();
            if ((threadIdx.x & 63) == 0 && phase >= 0)
              dbg[((size_t)blockIdx.x * 3 + (threadIdx.x >> 6)) * 8 + phase] += (unsigned long long)(now - tclk);
            tclk = now;
          }
      };
      const int i0 = G.i0, j0 = G.j0, kA = G.kA, kB = G.kB, C = G.C, role = G.role, cx = G.cx, cy = G.cy, hb = G.hb;
      constexpr bool lin = LIN;
      // constants of the role, the cell volume folded in (cell_w27 leaves it out)
      double cst[3];
      if constexpr (KIND == 0)
        {
          cst[0] = S.cA[CC][0] * S.vol;
          cst[1] = S.cA[CC][1] * S.vol;
          cst[2] = S.cA[CC][2] * S.vol;
        }
      else
        {
          const double l = S.cTl[KIND - 1] * S.vol, m = S.cTm[KIND - 1] * S.vol;
          cst[0] = (C < role) ? l : m;
          cst[1] = (C < role) ? m : l;
          cst[2] = 0.0;
        }
      const double pscale = (RES && KIND == 0) ? S.aB1 * S.p / (1.0 - S.kappa) * S.ih[CC] * S.vol : 0.0;
      const double kv4 = S.kappa * 0.25;
      Wc6 K;
      K.lo = S.use_old ? -1.0e300 : 0.0;
      K.hi = S.use_old ? 1.0e300 : 1.0;
      K.omk = 1.0 - S.kappa;
      K.kap = S.kappa;
      K.ivol = 1.0 / S.vol;
      double *const stage_d = s.stage + role;
      double *const rpart_d = &s.rpart[role][0];

      // ---- nodal planes: global -> registers (issued early) -> LDS ring (stored behind the copy-out).  The work is split by
      // WAVE (scalar branches, a wave only runs its own part):
      //   wave 0: halo nodes 0..63: the old phase fields and the flag byte;
      //   wave 1: halo nodes 0..63: the displacements (RES); lanes 0..48: row info of the NEXT node plane;
      //   wave 2: halo nodes 64..80: lanes 0..16 the phase fields and flags, lanes 32..48 the displacements (RES).
      // The lattice position of a lane's node is fixed for the whole march: its in-plane parts are formed once.
      int nd_lex = 0, nd_box = 0; // nd_box: bits 30, 31 = inside the lattice / inside the owned box (x, y)
      {
        const int lane = t & 63;
        int hn = lane; // waves 0, 1
        if (role == 2)
          hn = (lane < 32) ? 64 + lane : 64 + lane - 32;
        const bool used = role < 2 || lane < 17 || (lane >= 32 && lane < 49);
        const int gi = i0 - 1 + hn % UH, gj = j0 - 1 + hn / UH;
        const bool in = used && hn < NUH && gi >= 0 && gi < cv.NX && gj >= 0 && gj < cv.NY;
        const bool own = in && cv.owned_lex && gi >= cv.o0[0] && gi <= cv.o1[0] && gj >= cv.o0[1] && gj <= cv.o1[1];
        nd_lex = (gi - cv.o0[0]) + (cv.o1[0] - cv.o0[0] + 1) * (gj - cv.o0[1]);
        nd_box = (in ? (gi + cv.NX * gj) : 0) | (in ? (1 << 30) : 0) | (own ? (1 << 31) : 0);
      }
      const int lex_sz = (cv.o1[0] - cv.o0[0] + 1) * (cv.o1[1] - cv.o0[1] + 1), box_sz = cv.NX * cv.NY;
      auto node_of = [&](int lex, int box, int kz, bool &valid) __attribute__((always_inline)) -> int {
        valid = ((box >> 30) & 1) && kz >= 0 && kz < cv.NZ;
        if (!valid)
          return 0;
        if (box < 0 && kz >= cv.o0[2] && kz <= cv.o1[2])
          return lex + lex_sz * (kz - cv.o0[2]);
        return cv.local_of_box[(box & 0x3fffffff) + (long long)box_sz * kz];
      };
      double d0 = 0.0, d1 = 0.0, d2 = 0.0;
      unsigned ffl = 0u;
      long long roff = -1;
      unsigned rmask = 0u;
      auto fetch_plane = [&](int kz) __attribute__((always_inline)) {
        int lq = t & 63;
        asm volatile("" : "+v"(lq)); // recomputed per step, not kept live across the march
        bool valid;
        const int n = node_of(nd_lex, nd_box, kz, valid);
        d0 = d1 = d2 = 0.0;
        ffl = 0u;
        const bool phase_part = role == 0 || (role == 2 && lq < 32);
        if (valid)
          {
            if (phase_part)
              {
                d0 = v.phi_old[n];
                d1 = v.phi_oldold[n];
                ffl = v.node_flags[n];
              }
            else if (RES)
              {
                d0 = v.u[0][n];
                d1 = v.u[1][n];
                d2 = v.u[2][n];
              }
          }
      };
      auto commit_plane = [&](int kz) __attribute__((always_inline)) {
        int lq = t & 63;
        asm volatile("" : "+v"(lq));
        const int slot = kz & 1;
        if (role != 1) // waves 0 and 2: phase fields + flags
          {
            const int hn = role == 0 ? lq : 64 + lq;
            if (role == 0 || lq < 17)
              {
                double a = d0;
                if (lin) // one combined field is interpolated (cell_w27<true>)
                  a = S.use_old ? d0 : d1 + S.tfac * (d0 - d1);
                s.P[slot][0][hn] = a;
                if (!lin)
                  s.P[slot][1][hn] = d1;
                s.flag[kz & 3][hn] = (unsigned char)ffl;
              }
            const unsigned long long any = __ballot((role == 0 || lq < 17) && (ffl & 7u) != 0);
            if (lq == 0)
              s.anyflag[kz & 3][role >> 1] = any != 0;
          }
        if (RES && (role == 1 || (role == 2 && lq >= 32 && lq < 49)))
          {
            const int hn = role == 1 ? lq : 64 + lq - 32;
            s.U[slot][0][hn] = d0;
            s.U[slot][1][hn] = d1;
            s.U[slot][2][hn] = d2;
          }
      };
      auto fetch_rows = [&](int kz) __attribute__((always_inline)) {
        roff = -1;
        rmask = 0u;
        if (role == 1) // lanes 0..48: the owned node of row nl = lane (recomputed per step: one wave, a dozen instructions)
          {
            int nl = t & 63;
            asm volatile("" : "+v"(nl));
            const int gi = i0 + nl % UN, gj = j0 + nl / UN;
            if (nl < NUN && gi <= cv.o1[0] && gj <= cv.o1[1])
              {
                const int r = cart_local_id(cv, gi, gj, kz);
                roff = v.nadj_ptr[r];
                rmask = cv.nbr_mask[r];
              }
          }
      };
      auto commit_rows = [&](int kz) __attribute__((always_inline)) {
        if (role == 1)
          {
            int lq = t & 63;
            asm volatile("" : "+v"(lq));
            if (lq < NUN)
              {
                s.off[kz & 1][lq] = roff;
                s.mask[kz & 1][lq] = rmask;
              }
            const unsigned long long irr = __ballot(lq < NUN && (roff < 0 || rmask != 0x7ffffffu));
            if (lq == 0)
              s.irregular[kz & 1] = irr != 0;
          }
      };

      fetch_plane(kA - 1);
      commit_plane(kA - 1);
      fetch_plane(kA);
      commit_plane(kA);
      __syncthreads();
      const bool tile_full = (i0 + UN - 1) <= cv.o1[0] && (j0 + UN - 1) <= cv.o1[1];
      const bool fast_ok = NCOL == 3 && cv.owned_lex && tile_full;

#pragma unroll 1
      for (int ck = kA - 1; ck < kB; ++ck)
        {
          const int lo = ck & 1, hi = lo ^ 1;
          const bool more = ck + 1 < kB;
          stamp(-1);
          // requests for the step after this one: their latency is covered by the arithmetic below
          if (more)
            {
              fetch_plane(ck + 2);
              fetch_rows(ck + 1);
            }
          const bool cell_ok = G.col_ok && ck >= 0 && ck < cv.NZ - 1;
          // an absent cell (outside the mesh) contributes zeros: its g is made zero through per-lane kappa terms
          Wc6 Kl = K;
          Kl.omk = cell_ok ? K.omk : 0.0;
          Kl.kap = cell_ok ? K.kap : 0.0;
          const double kv4l = cell_ok ? kv4 : 0.0;
          double tab[NTAB], kpart[4] = {0.0, 0.0, 0.0, 0.0};
          {
            double ud[8], pres[8], kinit0[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int a = 0; a < 8; ++a)
              pres[a] = 0.0;
            double pa[8], pb[8], w27[27];
#pragma unroll
            for (int b = 0; b < 8; ++b)
              {
                const int sl = (b >> 2) ? hi : lo, hn = hb + (b & 1) + UH * ((b >> 1) & 1);
                pa[b] = s.P[sl][0][hn];
                pb[b] = lin ? 0.0 : s.P[sl][1][hn];
              }
            cell_w27<LIN>(pa, pb, S, Kl, cell_ok, w27);
            if constexpr (KIND == 0)
              {
                double A[27];
                tables_A(w27, A);
                if constexpr (RES)
                  {
                    static_for<8>([&](auto Aa) __attribute__((always_inline)) {
                      constexpr int a = decltype(Aa)::value;
                      pres[a] = pscale * pres_moment<CC, a>(A, kv4l);
                    });
                  }
                entries_A(A, cst, tab);
              }
            else
              tables_T<KIND - 1>(w27, tab);
#pragma unroll
            for (int b = 0; b < 8; ++b)
              {
                ud[b] = 0.0;
                if constexpr (RES)
                  ud[b] = s.U[(b >> 2) ? hi : lo][role][hb + (b & 1) + UH * ((b >> 1) & 1)];
              }
            if constexpr (RES)
              res_prepare<KIND>(tab, ud, cst, pres, kinit0, kpart);
            stamp(0); // requests, w*g, tables
            // ---- lower vertices: the rows of plane ck are complete afterwards
            if (ck >= kA)
              push_pass<KIND, 0, RES>(tab, ud, cst, kinit0, stage_d, rpart_d, cx, cy);
          }
          stamp(1); // pushes of the lower vertices
          lds_barrier();
          stamp(2); // barrier
          // the plane after next and the rows of the next plane: ring slot `lo` (plane ck) is dead, every wave has read its
          // vertex values before the barrier above.  (Stored BEFORE the copy-out: the staging registers are free again.)
          if (more)
            {
              commit_plane(ck + 2);
              commit_rows(ck + 1);
            }
          stamp(4); // ring stores

          // ---- plane ck: constraints as masks, then stream the rows out
          if (ck >= kA)
            {
              const int cp = lo;
              // ONE LDS round trip for everything the common case needs: the plane's flags, the 7 line bases and the thread's
              // 21 values (thread <-> fixed positions p = t, t + 192, t + 384 of a y-line's 567 values, the last one
              // for t < 183; on the rare planes that take the generic path below the values are read for nothing).  A
              // chain of dependent reads costs several hundred cycles each while the other waves push.
              int tq = t;
              asm volatile("" : "+v"(tq));
              const double *src = s.stage + tq;
              constexpr int H0 = 4; // lines of the first batch
              int fl[7];
              long long o7[UN];
              double val[H0][3];
#pragma unroll
              for (int i = 0; i < 3; ++i)
                {
                  fl[2 * i] = s.anyflag[(ck - 1 + i) & 3][0];
                  fl[2 * i + 1] = s.anyflag[(ck - 1 + i) & 3][1];
                }
              fl[6] = s.irregular[cp];
#pragma unroll
              for (int ny = 0; ny < UN; ++ny)
                o7[ny] = s.off[cp][ny * UN]; // same address for every lane
#pragma unroll
              for (int ny = 0; ny < H0; ++ny)
                {
                  val[ny][0] = src[ny * LINE6];
                  val[ny][1] = src[ny * LINE6 + NT6];
                  val[ny][2] = src[ny * LINE6 + 2 * NT6]; // t >= 183: a value of the next line, not stored
                }
              __builtin_amdgcn_sched_barrier(0);
              const bool masked = (fl[0] | fl[1] | fl[2] | fl[3] | fl[4] | fl[5]) != 0;
              const bool fast = fast_ok && !masked && fl[6] == 0;
              if (G.abl & 2)
                ;
              else if (fast)
                {
                  // Blocked layout, interior plane without constraint flags: the row (node, C) is one contiguous run of 81
                  // values at 9 off + 81 C; x-consecutive full rows are 243 apart.  Global addresses are a wave-uniform
                  // line base (scalar) + a 32-bit byte offset per position.
                  const bool act2 = tq < LINE6 - 2 * NT6;
                  // (nx, el) of position t + 192 q: 192 = 2 * 81 + 30
                  const int nx0 = (tq >= ROW6 ? 1 : 0) + (tq >= 2 * ROW6 ? 1 : 0), el0 = tq - ROW6 * nx0;
                  const int w1 = (el0 + 30 >= ROW6) ? 1 : 0, el1 = el0 + 30 - ROW6 * w1, nx1 = nx0 + 2 + w1;
                  const int w2 = (el1 + 30 >= ROW6) ? 1 : 0, el2 = el1 + 30 - ROW6 * w2, nx2 = nx1 + 2 + w2;
                  const unsigned b0 = 8u * (unsigned)(nx0 * 243 + el0), b1 = 8u * (unsigned)(nx1 * 243 + el1),
                                 b2 = act2 ? 8u * (unsigned)(nx2 * 243 + el2) : 0u;
                  char *const vbase = reinterpret_cast<char *>(vals + ROW6 * C);
                  long long off0[UN];
#pragma unroll
                  for (int ny = 0; ny < UN; ++ny)
                    off0[ny] = ((long long)__builtin_amdgcn_readfirstlane((int)(o7[ny] >> 32)) << 32) |
                               (unsigned)__builtin_amdgcn_readfirstlane((int)o7[ny]);
                  // second batch of values: requested before the stores of the first
                  double val2[UN - H0][3];
#pragma unroll
                  for (int ny = H0; ny < UN; ++ny)
                    {
                      val2[ny - H0][0] = src[ny * LINE6];
                      val2[ny - H0][1] = src[ny * LINE6 + NT6];
                      val2[ny - H0][2] = src[ny * LINE6 + 2 * NT6];
                    }
                  if (!(G.abl & 1))
                    {
#pragma unroll
                      for (int ny = 0; ny < H0; ++ny)
                        {
                          char *base = vbase + 72 * off0[ny];
                          *reinterpret_cast<double *>(base + b0) = val[ny][0];
                          *reinterpret_cast<double *>(base + b1) = val[ny][1];
                        }
                      if (act2)
                        {
#pragma unroll
                          for (int ny = 0; ny < H0; ++ny)
                            *reinterpret_cast<double *>(vbase + 72 * off0[ny] + b2) = val[ny][2];
                        }
                      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                      for (int ny = H0; ny < UN; ++ny)
                        {
                          char *base = vbase + 72 * off0[ny];
                          *reinterpret_cast<double *>(base + b0) = val2[ny - H0][0];
                          *reinterpret_cast<double *>(base + b1) = val2[ny - H0][1];
                        }
                      if (act2)
                        {
#pragma unroll
                          for (int ny = H0; ny < UN; ++ny)
                            *reinterpret_cast<double *>(vbase + 72 * off0[ny] + b2) = val2[ny - H0][2];
                        }
                    }
                }
              else
                {
                  // rows at the faces of the box, partial tiles, rows next to ghost columns, constraint flags nearby,
                  // interleaved layout: thread <-> (row, lattice offset o, column component); the CSR slot of offset o is
                  // its rank among the offsets that exist, or the row's permutation of that rank
                  constexpr int rowlen = 27 * NCOL;
#pragma unroll 1
                  for (int f = t; f < NUN * rowlen; f += NT6)
                    {
                      const int nl = f / rowlen, e = f - nl * rowlen;
                      const int o = e / NCOL, d = e - o * NCOL;
                      const long long base = s.off[cp][nl];
                      const unsigned nmask = s.mask[cp][nl];
                      if (base < 0 || !((nmask >> o) & 1u))
                        continue;
                      int sl = __popc(nmask & ((1u << o) - 1u));
                      const int deg = __popc(nmask & 0x7ffffffu);
                      if (nmask >> 31) // the row is not in lattice order (ghost columns behind the owned ones, bound pattern)
                        sl = cv.row_perm[base + sl];
                      double val = (d < 3) ? s.stage[nl * ROW6 + o * 3 + d] : 0.0;
                      if (masked && d < 3)
                        {
                          const int hn = (nl % UN + 1) + UH * (nl / UN + 1);
                          const int oz = o / 9, o9 = o - 9 * oz;
                          const unsigned row_flag = s.flag[ck & 3][hn];
                          const unsigned cf = s.flag[(ck + oz - 1) & 3][hn + (o9 % 3 - 1) + UH * (o9 / 3 - 1)];
                          const bool rcon = (row_flag >> C) & 1u;
                          if (rcon || ((cf >> d) & 1u))
                            val = (rcon && o == 13 && d == C) ? val : 0.0; // constrained row: its own diagonal only; eliminated column: 0
                        }
                      vals[(long long)NCOL * NCOL * base + (long long)C * NCOL * deg + sl * NCOL + d] = val;
                    }
                }
              if constexpr (RES)
                {
                  // residual row (node, C): the three roles in a fixed order; constrained rows get 0 (cracks.cc:2440-2456)
                  int tq = t;
                  asm volatile("" : "+v"(tq));
                  if (tq < NUN && s.off[cp][tq] >= 0)
                    {
                      const double sum = (s.rpart[0][tq] + s.rpart[1][tq]) + s.rpart[2][tq];
                      const int nx = tq % UN, ny = tq / UN;
                      const bool con = (s.flag[ck & 3][(nx + 1) + UH * (ny + 1)] >> C) & 1u;
                      const int row = cart_local_id(cv, i0 + nx, j0 + ny, ck);
                      const long long di = (v.layout == PFM_LAYOUT_INTERLEAVED) ? (long long)row * 4 + C : (long long)row * 3 + C;
                      res_pde[di] = con ? 0.0 : -sum;
                    }
                }
            }
          stamp(3); // copy-out
          lds_barrier();
          stamp(5); // barrier
          // ---- upper vertices: start the rows of plane ck + 1 in the same buffer (first pushes are plain stores)
          if (more)
            {
              double ud[8];
#pragma unroll
              for (int b = 0; b < 8; ++b)
                {
                  ud[b] = 0.0;
                  if constexpr (RES)
                    if (b >= 4)
                      ud[b] = s.U[hi][role][hb + (b & 1) + UH * ((b >> 1) & 1)];
                }
              push_pass<KIND, 1, RES>(tab, ud, cst, kpart, stage_d, rpart_d, cx, cy);
            }
          stamp(6); // pushes of the upper vertices
        }
    }

    template <int NCOL /* 3 blocked, 4 interleaved */, bool RES /* also writes the displacement rows of the residual */,
              bool LIN /* staggered scheme: one combined old phase field, no q-point clamps of the old fields */, bool CLK = false>
    __global__ __launch_bounds__(NT6, 3) void k_cart_uu6(DevView v, CartView cv, const MatScal *__restrict__ Sp, double *__restrict__ vals,
                                                         int zc_in /* node planes per chunk */, double *__restrict__ res_pde,
                                                         unsigned long long *__restrict__ dbg)
    {
      __shared__ Lds6 s;
      const MatScal &S = *Sp; // per-launch scalars live in device memory: loaded where used
      const int t = threadIdx.x, lane = t & 63;
      Geo6 G;
      G.role = __builtin_amdgcn_readfirstlane(t >> 6); // column component D, wave-uniform
      G.cx = lane % UT;
      G.cy = lane / UT;
      const int abl = zc_in >> 16, zc = zc_in & 0xffff; // bits 16..: ablations (PFM_UU_ABL, profiling only)
      const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1, OWZ = cv.o1[2] - cv.o0[2] + 1;
      const int ntx = (OWX + UN - 1) / UN, nty = (OWY + UN - 1) / UN, nch = (OWZ + zc - 1) / zc;
      const int bid = xcd_tile_index();
      if (bid >= 3 * ntx * nty * nch)
        return; // padding of the XCD-aware grid
      G.C = bid % 3; // the three row components of a tile are neighbours in the launch: shared planes in L2
      const int tl = bid / 3;
      const int tix = tl % ntx, tiy = (tl / ntx) % nty, chunk = tl / (ntx * nty);
      G.i0 = cv.o0[0] + tix * UN;
      G.j0 = cv.o0[1] + tiy * UN;
      G.abl = abl;
      G.kA = cv.o0[2] + chunk * zc;
      G.kB = min(G.kA + zc, cv.o1[2] + 1); // node planes [kA, kB)
      const int ci = G.i0 - 1 + G.cx, cj = G.j0 - 1 + G.cy;
      G.col_ok = ci >= 0 && ci < cv.NX - 1 && cj >= 0 && cj < cv.NY - 1;
      G.hb = G.cy * UH + G.cx; // halo index of the cell's (0,0) vertex
      // role kind: scalar branches, every wave runs the same sequence of barriers
      const int kind = (G.role == G.C) ? 0 : 1 + pair6(min(G.role, G.C), max(G.role, G.C));
      if (kind == 0 && G.C == 0)
        march6<0, 0, NCOL, RES, LIN, CLK>(s, v, cv, S, G, vals, res_pde, dbg);
      else if (kind == 0 && G.C == 1)
        march6<0, 1, NCOL, RES, LIN, CLK>(s, v, cv, S, G, vals, res_pde, dbg);
      else if (kind == 0)
        march6<0, 2, NCOL, RES, LIN, CLK>(s, v, cv, S, G, vals, res_pde, dbg);
      else if (kind == 1)
        march6<1, 0, NCOL, RES, LIN, CLK>(s, v, cv, S, G, vals, res_pde, dbg);
      else if (kind == 2)
        march6<2, 0, NCOL, RES, LIN, CLK>(s, v, cv, S, G, vals, res_pde, dbg);
      else
        march6<3, 0, NCOL, RES, LIN, CLK>(s, v, cv, S, G, vals, res_pde, dbg);
    }
  } // namespace

  bool cart_uu6_supported(const DevView &v, const CartView &cv)
  {
    return v.dim == 3 && !cv.cell_lam && cv.tile_sel == 0;
  }

  int launch_cart_uu6(const DevView &v, const CartView &cv, const pfm_params &p, double *vals_uu, hipStream_t s, const void *d_scal,
                      double *res_pde)
  {
    int rc = ensure_g1();
    if (rc)
      return rc;
    if (!cart_uu6_supported(v, cv))
      return PFM_ERR_UNSUPPORTED;
    const MatScal *S = static_cast<const MatScal *>(d_scal);
    const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1, OWZ = cv.o1[2] - cv.o0[2] + 1;
    if (OWX <= 0 || OWY <= 0 || OWZ <= 0)
      return PFM_OK;
    const int ntx = (OWX + UN - 1) / UN, nty = (OWY + UN - 1) / UN;
    static const int zc_force = getenv("PFM_UU_ZC") ? atoi(getenv("PFM_UU_ZC")) : 0; // tuning only
    const int zc0 = zc_force > 0 ? zc_force : choose_zchunk(3LL * ntx * nty, OWZ, 6, 25, 4);
    static const int abl = getenv("PFM_UU_ABL") ? atoi(getenv("PFM_UU_ABL")) : 0; // profiling only
    const int zc = zc0 | (abl << 16);
    const int nch = (OWZ + zc0 - 1) / zc0;
    const unsigned nb = (unsigned)(3 * ntx * nty * nch);
    const dim3 grid(xcd_grid(nb)), block(NT6);
    const bool il = v.layout == PFM_LAYOUT_INTERLEAVED, res = res_pde != nullptr;
    const bool lin = p.outer_solver != PFM_SOLVER_SIMPLE_MONOLITHIC;
    if (res && !lin)
      return PFM_ERR_BAD_ARG; // the residual from the rows needs the unclamped staggered scheme (launch_assemble_cart)
#define PFM_UU6(NC, RESV, LINV) hipLaunchKernelGGL((k_cart_uu6<NC, RESV, LINV>), grid, block, 0, s, v, cv, S, vals_uu, zc, res_pde, nullptr)
    if (getenv("PFM_UU_CLK") && !il && res) // profiling only: cycles per phase and role
      {
        static unsigned long long *d_dbg = nullptr;
        const size_t nd = (size_t)xcd_grid(nb) * 3 * 8;
        if (!d_dbg && hipMalloc((void **)&d_dbg, nd * sizeof(unsigned long long)) != hipSuccess)
          return PFM_ERR_HIP;
        (void)hipMemsetAsync(d_dbg, 0, nd * sizeof(unsigned long long), s);
        hipLaunchKernelGGL((k_cart_uu6<3, true, true, true>), grid, block, 0, s, v, cv, S, vals_uu, zc, res_pde, d_dbg);
        std::vector<unsigned long long> hall(nd);
        (void)hipMemcpy(hall.data(), d_dbg, nd * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double h[3][8] = {};
        for (size_t i = 0; i < nd; ++i)
          h[(i / 8) % 3][i % 8] += (double)hall[i];
        const char *names[7] = {"requests+w*g+tables", "push lower", "barrier", "copy-out", "ring stores", "barrier", "push upper"};
        for (int w = 0; w < 3; ++w)
          {
            fprintf(stderr, "[k_cart_uu6 phase clock, wave %d, cycles per workgroup (%d planes)]", w, zc0);
            for (int i = 0; i < 7; ++i)
              fprintf(stderr, " %s=%.0f", names[i], h[w][i] / nb);
            fprintf(stderr, "\n");
          }
      }
    else if (il)
      {
        if (res)
          PFM_UU6(4, true, true);
        else if (lin)
          PFM_UU6(4, false, true);
        else
          PFM_UU6(4, false, false);
      }
    else
      {
        if (res)
          PFM_UU6(3, true, true);
        else if (lin)
          PFM_UU6(3, false, true);
        else
          PFM_UU6(3, false, false);
      }
#undef PFM_UU6
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }
} // namespace pfm
