// pfm_cart_common.h — tile geometry, 1-D Gauss tables and per-launch scalars shared by the
// row-owner Jacobian kernels (pfm_cart_uu3.hip, pfm_cart_phi4.hip).  Everything lives in
// an anonymous namespace: each translation unit owns its copy of the __constant__ table.
#pragma once
#include "pfm_internal.h"

#include <hip/hip_runtime.h>
#include <type_traits>

namespace pfm
{
  namespace
  {
    constexpr int STG = 81; // staged row width (27 slots x 3), odd => conflict-free
    // tile index of this workgroup under the XCD-aware launch (pfm_internal.h: xcd_grid)
    __device__ __forceinline__ int xcd_tile_index() { return (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)); }

    // Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, i.e.
    // every wave would sit out the full HBM write latency of the rows it has just streamed out; the kernels
    // here never exchange data through global memory inside a launch, so outstanding stores may stay in flight.
    __device__ __forceinline__ void lds_barrier()
    {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // local node id of lattice node (i,j,k): arithmetic for owned nodes when they are numbered lexicographically
    // (checked at context creation), table look-up otherwise (ghost layers, arbitrary numberings)
    __device__ __forceinline__ int cart_local_id(const CartView &cv, int i, int j, int k)
    {
      if (cv.owned_lex && i >= cv.o0[0] && i <= cv.o1[0] && j >= cv.o0[1] && j <= cv.o1[1] && k >= cv.o0[2] && k <= cv.o1[2])
        return (i - cv.o0[0]) + (cv.o1[0] - cv.o0[0] + 1) * ((j - cv.o0[1]) + (cv.o1[1] - cv.o0[1] + 1) * (k - cv.o0[2]));
      return cv.local_of_box[i + (long long)cv.NX * (j + (long long)cv.NY * k)];
    }

    // The same with the look-up WAITED FOR INSIDE ITS ARM.  A load whose result is defined in one arm of a branch is waited
    // for at the join with s_waitcnt vmcnt(0), on every path -- and vmcnt counts stores: in a wave that streams rows out,
    // or that has global -> LDS transfers in flight, that is a wait for all of them (k_cart_uu3, round 5: 3600 cycles per
    // plane in the wave that requests the next rows, although every id was arithmetic)
    __device__ __forceinline__ int cart_local_id_sync(const CartView &cv, int i, int j, int k)
    {
      if (cv.owned_lex && i >= cv.o0[0] && i <= cv.o1[0] && j >= cv.o0[1] && j <= cv.o1[1] && k >= cv.o0[2] && k <= cv.o1[2])
        return (i - cv.o0[0]) + (cv.o1[0] - cv.o0[0] + 1) * ((j - cv.o0[1]) + (cv.o1[1] - cv.o0[1] + 1) * (k - cv.o0[2]));
      int id = cv.local_of_box[i + (long long)cv.NX * (j + (long long)cv.NY * k)];
      asm volatile("" : "+v"(id));
      return id;
    }

    // row of lattice node (i,j,k) in this launch, -1: none (CartView::row_of_box)
    __device__ __forceinline__ int cart_row_id(const CartView &cv, int i, int j, int k)
    {
      if (cv.row_of_box)
        return cv.row_of_box[i + (long long)cv.NX * (j + (long long)cv.NY * k)];
      return cart_local_id(cv, i, j, k);
    }
    __device__ __forceinline__ int cart_row_id_sync(const CartView &cv, int i, int j, int k)
    {
      if (cv.row_of_box)
        {
          int id = cv.row_of_box[i + (long long)cv.NX * (j + (long long)cv.NY * k)];
          asm volatile("" : "+v"(id));
          return id;
        }
      return cart_local_id_sync(cv, i, j, k);
    }

    // x += v on an LDS double, done by the LDS unit (ds_add_f64, no return value): one LDS instruction and no
    // read -> wait -> add -> write round trip.  Used where a wave's lanes hit distinct addresses and the order of
    // the adds is the program order of that wave, so the result is the same as a plain read-modify-write.
    __device__ __forceinline__ void lds_add(double *p, double v)
    {
      unsafeAtomicAdd(p, v);
    }

    // One ds_read_b64 (16-bit immediate offset) for an LDS double: a volatile access in the LDS address space is never
    // paired into ds_read2_b64.  The paired form is serviced in 16-lane groups at half the bytes per clock
    // (MI355X_MICROARCH.md, LDS table), needs extra base registers for its 8-bit offsets, and returns register
    // tuples whose halves belong to unrelated values (copies wherever control flow joins).
    typedef __attribute__((address_space(3))) const volatile double lds_cvdouble;
    __device__ __forceinline__ double lds_read64(const double *p) { return *(lds_cvdouble *)p; }

    template <int N, class F>
    __device__ __forceinline__ __attribute__((always_inline)) void static_for(F &&f)
    {
      if constexpr (N > 0)
        {
          static_for<N - 1>(f);
          f(std::integral_constant<int, N - 1>{});
        }
    }

    // tiles of pfm_cart_uu3.hip: 8 x 4 nodes, 512 threads
    constexpr int T3X = 8, T3Y = 4, NT3 = 512;
    constexpr int H3X = T3X + 2, H3Y = T3Y + 2, NH3 = H3X * H3Y * 3; // nodal halo 10 x 6 x 3
    constexpr int C3X = T3X + 1, C3Y = T3Y + 1, CL3 = C3X * C3Y;     // 45 cells per layer
    constexpr int CS3 = 2 * CL3;                                     // 90 cell slots
    constexpr int NN3 = T3X * T3Y;                                   // 32 nodes per tile

    struct G1
    {
      double n[2][3], m[3][3], w[3]; // n_al(q), m_g(q) (g = 0:00, 1:01, 2:11), weights
      double mb[3];                  // 1-D mass moments mbar_g = sum_q w m_g(q)
      double sx[3][5];               // sx[j][W] = sum_q w t_q^j W(q), t = n_1, W = n_0, n_1, m_0, m_1, m_2 (k_cart_phi4: contraction of
                                     // a polynomial in t along a line from its coefficients; 3-point Gauss is exact for these degrees)
    };
    __constant__ G1 c_g1;

    // G1::sx as compile-time constants: literals are materialised where they are used (s_mov pairs), values loaded from
    // c_g1 occupy scalar registers across the whole line loop -- and the role loops of k_cart_phi4 have none to spare
    // (the loaded table cost 76 spill reloads per z-level when it was tried)
    constexpr double g1_sx(int j, int W)
    {
      const double gx[3] = {0.5 - 0.5 * 0.7745966692414834, 0.5, 0.5 + 0.5 * 0.7745966692414834};
      const double gw[3] = {5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0};
      double sum = 0.0;
      for (int q = 0; q < 3; ++q)
        {
          const double n0 = 1.0 - gx[q], n1 = gx[q];
          const double tj = j == 0 ? 1.0 : (j == 1 ? gx[q] : gx[q] * gx[q]);
          const double Wq = W == 0 ? n0 : (W == 1 ? n1 : (W == 2 ? n0 * n0 : (W == 3 ? n0 * n1 : n1 * n1)));
          sum += gw[q] * tj * Wq;
        }
      return sum;
    }
    // the entries of G1 as literals (same arithmetic as make_g1, evaluated by the compiler): inside a loop a value read from
    // c_g1 is a loop invariant that is hoisted and held -- or spilled -- across every phase of the loop body
    constexpr double g1_gx(int q) { return q == 0 ? 0.5 - 0.5 * 0.7745966692414834 : (q == 1 ? 0.5 : 0.5 + 0.5 * 0.7745966692414834); }
    constexpr double g1_w(int q) { return q == 1 ? 8.0 / 18.0 : 5.0 / 18.0; }
    constexpr double g1_n(int a, int q) { return a == 0 ? 1.0 - g1_gx(q) : g1_gx(q); }
    constexpr double g1_m(int g, int q) { return g == 0 ? g1_n(0, q) * g1_n(0, q) : (g == 1 ? g1_n(0, q) * g1_n(1, q) : g1_n(1, q) * g1_n(1, q)); }

    template <int J, int W>
    struct G1Sx
    {
      static constexpr double v = g1_sx(J, W);
    };

    G1 make_g1()
    {
      G1 t{};
      const double gx[3] = {0.5 - 0.5 * 0.7745966692414834, 0.5, 0.5 + 0.5 * 0.7745966692414834};
      const double gw[3] = {5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0};
      for (int q = 0; q < 3; ++q)
        {
          t.n[0][q] = 1.0 - gx[q];
          t.n[1][q] = gx[q];
          t.m[0][q] = t.n[0][q] * t.n[0][q];
          t.m[1][q] = t.n[0][q] * t.n[1][q];
          t.m[2][q] = t.n[1][q] * t.n[1][q];
          t.w[q] = gw[q];
        }
      for (int g = 0; g < 3; ++g)
        t.mb[g] = t.w[0] * t.m[g][0] + t.w[1] * t.m[g][1] + t.w[2] * t.m[g][2];
      for (int j = 0; j < 3; ++j)
        for (int W = 0; W < 5; ++W)
          {
            double sum = 0.0;
            for (int q = 0; q < 3; ++q)
              {
                const double tj = j == 0 ? 1.0 : (j == 1 ? gx[q] : gx[q] * gx[q]);
                sum += gw[q] * tj * (W < 2 ? t.n[W][q] : t.m[W - 2][q]);
              }
            t.sx[j][W] = sum;
          }
      return t;
    }

    struct MatScal
    {
      double lam, mu, kappa, eps, Gc, p, aB1, gamma_fac, tfac;
      double ih[3], vol;
      double hz;       // h_z (the phase-field rows interpolate along z first)
      double cA[3][3]; // cA[c][k] = (k == c ? lam + 2 mu : mu) / h_k^2
      double cT[3];    // 1 / (h_lo h_hi) for the pairs (0,1), (0,2), (1,2)
      double cTl[3], cTm[3]; // cT * lambda, cT * mu: one FMA per table entry in the (u,u) node phase
      int monolithic, use_old;
      // uniform constants of the phase-field rows, precomputed on the host so that they arrive in scalar registers
      double c_muh, c_la, cdiag;       // 2(1-kappa) mu, 2(1-kappa) lambda, -2(alpha_B-1) p
      double omk, gc_eps, aB1p2, lap;  // 1-kappa, G_c/eps, 2(alpha_B-1) p, G_c eps vol
      double vww[3][3];                // vol w(a) w(b)
      double lapP[3][3], lapQ[3][3];   // Laplace moments [g_x][g_y]: in-plane part (times mbar_gz), d/dz part (times s(g_z))
      double lapM[27];                 // G_c eps sum_q w grad N_a . grad N_b by moment index g_x + 3 g_y + 9 g_z
      double pc_res;                   // (alpha_B - 1) p / (1 - kappa): pressure part of the displacement residual (k_cart_uu3<RES>)
    };

    // 1-D Gauss(3) data for a RUN-TIME point index (a per-lane index into the __constant__ table would be a vector
    // load from memory with an exposed cache round trip): same arithmetic as make_g1, bit-identical values
    __device__ __forceinline__ double gauss_n1(int q) { return fma((double)(q - 1), 0.5 * 0.7745966692414834, 0.5); }
    __device__ __forceinline__ double gauss_w(int q) { return (q == 1) ? 8.0 / 18.0 : 5.0 / 18.0; }

    // the same from ONE nodal field: staggered scheme (no clamping of the old fields at the q-points), where pf_extra is
    // linear in (phi_old, phi_oldold) up to its final clamp -- pw = phi_old if use_old_timestep_pf, else
    // phi_oldold + tfac (phi_old - phi_oldold), formed per node by the caller
    template <bool LIT = false /* Gauss data as literals instead of c_g1 */, class MatScalRef>
    __device__ __forceinline__ void cell_wg_plane_lin(const double pw[8], const MatScalRef &S, int qz, double wg[9])
    {
      auto N = [](int a, int q) __attribute__((always_inline)) { return LIT ? g1_n(a, q) : c_g1.n[a][q]; };
      auto Wt = [](int q) __attribute__((always_inline)) { return LIT ? g1_w(q) : c_g1.w[q]; };
      const double nz1 = gauss_n1(qz), nz0 = 1.0 - nz1, wz = gauss_w(qz); // q_z is a per-lane run-time index
      double a[4];
#pragma unroll
      for (int v = 0; v < 4; ++v)
        a[v] = nz0 * pw[v] + nz1 * pw[v + 4];
#pragma unroll
      for (int qy = 0; qy < 3; ++qy)
        {
          const double a0 = N(0, qy) * a[0] + N(1, qy) * a[2];
          const double a1 = N(0, qy) * a[1] + N(1, qy) * a[3];
#pragma unroll
          for (int qx = 0; qx < 3; ++qx)
            {
              double pfx = N(0, qx) * a0 + N(1, qx) * a1;
              if (!S.use_old)
                pfx = fmin(fmax(pfx, 0.0), 1.0);
              const double g = (1 - S.kappa) * pfx * pfx + S.kappa;
              wg[qx + 3 * qy] = S.vol * (Wt(qx) * Wt(qy) * wz) * g;
            }
        }
    }

    // weights w*g(q) of one cell at the 9 q-points of one z-level (cracks.cc:2262-2277)
    template <bool LIT = false, class MatScalRef>
    __device__ __forceinline__ void cell_wg_plane(const double po[8], const double poo[8], const MatScalRef &S, int qz,
                                                  double wg[9])
    {
      auto N = [](int a, int q) __attribute__((always_inline)) { return LIT ? g1_n(a, q) : c_g1.n[a][q]; };
      auto Wt = [](int q) __attribute__((always_inline)) { return LIT ? g1_w(q) : c_g1.w[q]; };
      const double nz1 = gauss_n1(qz), nz0 = 1.0 - nz1, wz = gauss_w(qz);
      double a[4], b[4];
#pragma unroll
      for (int v = 0; v < 4; ++v)
        {
          a[v] = nz0 * po[v] + nz1 * po[v + 4];
          b[v] = nz0 * poo[v] + nz1 * poo[v + 4];
        }
#pragma unroll
      for (int qy = 0; qy < 3; ++qy)
        {
          const double a0 = N(0, qy) * a[0] + N(1, qy) * a[2];
          const double a1 = N(0, qy) * a[1] + N(1, qy) * a[3];
          const double b0 = N(0, qy) * b[0] + N(1, qy) * b[2];
          const double b1 = N(0, qy) * b[1] + N(1, qy) * b[3];
#pragma unroll
          for (int qx = 0; qx < 3; ++qx)
            {
              double pfo = N(0, qx) * a0 + N(1, qx) * a1;
              double pfoo = N(0, qx) * b0 + N(1, qx) * b1;
              if (S.monolithic)
                {
                  pfo = fmax(0.0, pfo);
                  pfoo = fmax(0.0, pfoo);
                }
              double pfx = pfoo + S.tfac * (pfo - pfoo);
              if (pfx <= 0.0)
                pfx = 0.0;
              if (pfx >= 1.0)
                pfx = 1.0;
              if (S.use_old)
                pfx = pfo;
              const double g = (1 - S.kappa) * pfx * pfx + S.kappa;
              wg[qx + 3 * qy] = S.vol * (Wt(qx) * Wt(qy) * wz) * g;
            }
        }
    }

    bool g_g1_ready[16] = {};
    int ensure_g1()
    {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess)
        return PFM_ERR_HIP;
      if (dev < 16 && g_g1_ready[dev])
        return PFM_OK;
      const G1 t = make_g1();
      if (hipMemcpyToSymbol(HIP_SYMBOL(c_g1), &t, sizeof(t)) != hipSuccess)
        return PFM_ERR_HIP;
      if (dev < 16)
        g_g1_ready[dev] = true;
      return PFM_OK;
    }

    MatScal make_mat_scal(const pfm_params &prm, const CartView &cv)
    {
      MatScal s{};
      s.lam = prm.lambda;
      s.mu = prm.mu;
      s.kappa = prm.constant_k;
      s.eps = prm.alpha_eps;
      s.Gc = prm.G_c;
      s.p = prm.pressure;
      s.aB1 = prm.alpha_biot - 1.0;
      double gamma = prm.gamma_penal;
      if (prm.outer_solver == PFM_SOLVER_SIMPLE_MONOLITHIC && prm.timestep_number < 1)
        gamma = 0.0;
      double diam2 = 0.0;
      s.vol = 1.0;
      for (int d = 0; d < 3; ++d)
        {
          diam2 += cv.h[d] * cv.h[d];
          s.ih[d] = 1.0 / cv.h[d];
          s.vol *= cv.h[d];
          if (d == 2)
            s.hz = cv.h[d];
        }
      s.gamma_fac = gamma / prm.timestep * 1.0 / diam2;
      s.tfac = (prm.time - (prm.time - prm.old_timestep - prm.old_old_timestep)) /
               (prm.time - prm.old_timestep - (prm.time - prm.old_timestep - prm.old_old_timestep));
      for (int c = 0; c < 3; ++c)
        for (int k = 0; k < 3; ++k)
          s.cA[c][k] = (k == c ? prm.lambda + 2 * prm.mu : prm.mu) * s.ih[k] * s.ih[k];
      s.cT[0] = s.ih[0] * s.ih[1];
      s.cT[1] = s.ih[0] * s.ih[2];
      s.cT[2] = s.ih[1] * s.ih[2];
      for (int q = 0; q < 3; ++q)
        {
          s.cTl[q] = s.cT[q] * prm.lambda;
          s.cTm[q] = s.cT[q] * prm.mu;
        }
      s.monolithic = prm.outer_solver == PFM_SOLVER_SIMPLE_MONOLITHIC;
      s.use_old = prm.use_old_timestep_pf;
      s.c_muh = 2.0 * (1.0 - s.kappa) * s.mu;
      s.c_la = 2.0 * (1.0 - s.kappa) * s.lam;
      s.cdiag = -2.0 * s.aB1 * s.p;
      s.omk = 1.0 - s.kappa;
      s.gc_eps = s.Gc / s.eps;
      s.aB1p2 = 2.0 * s.aB1 * s.p;
      s.lap = s.Gc * s.eps * s.vol;
      s.pc_res = s.aB1 * s.p / (1.0 - s.kappa);
      const G1 g = make_g1();
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
          {
            s.vww[a][b] = s.vol * g.w[a] * g.w[b];
            const double sa = (a == 1) ? -1.0 : 1.0, sb = (b == 1) ? -1.0 : 1.0;
            s.lapP[a][b] = s.lap * (sa * s.ih[0] * s.ih[0] * g.mb[b] + sb * s.ih[1] * s.ih[1] * g.mb[a]);
            s.lapQ[a][b] = s.lap * s.ih[2] * s.ih[2] * g.mb[a] * g.mb[b];
          }
      for (int gz = 0; gz < 3; ++gz)
        for (int gy = 0; gy < 3; ++gy)
          for (int gx = 0; gx < 3; ++gx) // sign = -1 where a_k != b_k, i.e. where the moment index is 1
            s.lapM[gx + 3 * gy + 9 * gz] = g.mb[gz] * s.lapP[gx][gy] + ((gz == 1) ? -1.0 : 1.0) * s.lapQ[gx][gy];
      return s;
    }
  } // namespace
} // namespace pfm
