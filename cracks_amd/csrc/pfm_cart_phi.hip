// pfm_cart_phi.hip — row-owner kernel for the phase-field rows of the Jacobian (3-D):
// the (phi,u) block (cracks.cc:2374-2376, 2381-2382 with a displacement trial function),
// the (phi,phi) block (cracks.cc:2370-2371, 2377-2383 with a phase-field trial function) and
// the placeholder diagonals of constrained rows (deal.II distribute_local_to_global).
//
// Same tile machinery as k_cart_uu (pfm_cart_matrix.hip).  With the un-split stress
//   sigma_LinU:E + sigma+:E_LinU = 2 (sigma+ grad N_b)_d           (sigma+ = lambda trE I + 2 mu E)
// the (phi,u) entry of trial dof (b,d) and test vertex a is
//   K[a,(b,d)] = sum_k  s(b_k)/h_k * C^{dk}[a_k][g_i][g_j]
//   C^{dk}[al][g_i][g_j] = sum_q Phi^{dk}(q) n_al(q_k) m_{g_i}(q_i) m_{g_j}(q_j)
//   Phi^{dk} = w pf [ (2(1-kappa) lambda trE - 2(alpha_B-1) p) delta_dk + 4(1-kappa) mu E_dk ]
// and the (phi,phi) entry is a single table look-up
//   K[a,b] = M[g_x][g_y][g_z],  M = sum_q w c(q) m m m  +  G_c eps * (Laplace moments),
//   c(q) = (1-kappa) sigma+:E + G_c/eps - 2(alpha_B-1) p div u + gamma/dt/diam^2 [pf >= pf_old].
// One sub-phase per column component d keeps the LDS table at 54 numbers per cell.
#include "pfm_internal.h"
#include "pfm_cart_common.h"

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace pfm
{
  namespace
  {
    constexpr int NNUM_PHI = 56; // 54 (C^{dk}) per sub-phase; phi-phi: 27 M + avg + gzero + 27 partial
    constexpr int STG_PU = 81, STG_PP = 27;
    constexpr int N_M = 0, N_AVG = 27, N_GZERO = 28, N_PART = 29; // phi-phi number layout

    __host__ __device__ constexpr int idxC(int k, int al, int gi, int gj) { return k * 18 + al * 9 + gi * 3 + gj; }
    __host__ __device__ constexpr int sgn_(int bit) { return bit ? 1 : -1; }

    // (phi,u) entries of slot O for the current column component: sum over cells and over k
    template <int OX, int OY, int OZ>
    __device__ __forceinline__ double pu_entry(const double *__restrict__ lane_base, const MatScal &S)
    {
      double r = 0.0;
      auto visit = [&](auto EX, auto EY, auto EZ) {
        constexpr int ex = decltype(EX)::value, ey = decltype(EY)::value, ez = decltype(EZ)::value;
        constexpr int a[3] = {-ex, -ey, -ez};
        constexpr int b[3] = {a[0] + OX, a[1] + OY, a[2] + OZ};
        if constexpr (b[0] >= 0 && b[0] <= 1 && b[1] >= 0 && b[1] <= 1 && b[2] >= 0 && b[2] <= 1)
          {
            constexpr int g[3] = {a[0] + b[0], a[1] + b[1], a[2] + b[2]};
            const double *p = lane_base + (ez * (CX * CY) + ey * CX + ex);
            r += (double)sgn_(b[0]) * S.ih[0] * p[idxC(0, a[0], g[1], g[2]) * CS];
            r += (double)sgn_(b[1]) * S.ih[1] * p[idxC(1, a[1], g[0], g[2]) * CS];
            r += (double)sgn_(b[2]) * S.ih[2] * p[idxC(2, a[2], g[0], g[1]) * CS];
          }
      };
      using M1 = std::integral_constant<int, -1>;
      using Z0 = std::integral_constant<int, 0>;
      visit(M1{}, M1{}, M1{});
      visit(Z0{}, M1{}, M1{});
      visit(M1{}, Z0{}, M1{});
      visit(Z0{}, Z0{}, M1{});
      visit(M1{}, M1{}, Z0{});
      visit(Z0{}, M1{}, Z0{});
      visit(M1{}, Z0{}, Z0{});
      visit(Z0{}, Z0{}, Z0{});
      return r;
    }

    // (phi,phi) entry of slot O; for the centre slot also the constrained-row placeholder
    // sum_e (|K_e,aa| != 0 ? |K_e,aa| : avg_e) and the sum of avg_e over cells whose (u,u) diagonal is 0
    template <int OX, int OY, int OZ>
    __device__ __forceinline__ double pp_entry(const double *__restrict__ lane_base, double &placeholder,
                                               double &uu_patch)
    {
      double r = 0.0;
      auto visit = [&](auto EX, auto EY, auto EZ) {
        constexpr int ex = decltype(EX)::value, ey = decltype(EY)::value, ez = decltype(EZ)::value;
        constexpr int a[3] = {-ex, -ey, -ez};
        constexpr int b[3] = {a[0] + OX, a[1] + OY, a[2] + OZ};
        if constexpr (b[0] >= 0 && b[0] <= 1 && b[1] >= 0 && b[1] <= 1 && b[2] >= 0 && b[2] <= 1)
          {
            constexpr int g[3] = {a[0] + b[0], a[1] + b[1], a[2] + b[2]};
            const double *p = lane_base + (ez * (CX * CY) + ey * CX + ex);
            const double m = p[(N_M + g[0] + 3 * g[1] + 9 * g[2]) * CS];
            r += m;
            if constexpr (OX == 0 && OY == 0 && OZ == 0)
              {
                const double avg = p[N_AVG * CS];
                placeholder += (fabs(m) != 0.0) ? fabs(m) : avg;
                uu_patch += (p[N_GZERO * CS] != 0.0) ? avg : 0.0;
              }
          }
      };
      using M1 = std::integral_constant<int, -1>;
      using Z0 = std::integral_constant<int, 0>;
      visit(M1{}, M1{}, M1{});
      visit(Z0{}, M1{}, M1{});
      visit(M1{}, Z0{}, M1{});
      visit(Z0{}, Z0{}, M1{});
      visit(M1{}, M1{}, Z0{});
      visit(Z0{}, M1{}, Z0{});
      visit(M1{}, Z0{}, Z0{});
      visit(Z0{}, Z0{}, Z0{});
      return r;
    }

    template <int O>
    __device__ __forceinline__ void pu_slot(const double *lane_base, const MatScal &S, double *stage_row, int d,
                                            unsigned row_flag, unsigned col_flag)
    {
      constexpr int OX = O % 3 - 1, OY = (O / 3) % 3 - 1, OZ = O / 9 - 1;
      double val = pu_entry<OX, OY, OZ>(lane_base, S);
      if (((row_flag >> 3) & 1u) || ((col_flag >> d) & 1u))
        val = 0.0; // constrained row (active set) or eliminated column
      stage_row[O * 3 + d] = val;
    }

    template <int O>
    __device__ __forceinline__ void pp_slot(const double *lane_base, double *stage_row, unsigned row_flag,
                                            unsigned col_flag, double &uu_patch)
    {
      constexpr int OX = O % 3 - 1, OY = (O / 3) % 3 - 1, OZ = O / 9 - 1;
      double ph = 0.0;
      double val = pp_entry<OX, OY, OZ>(lane_base, ph, uu_patch);
      const bool rcon = (row_flag >> 3) & 1u;
      if (rcon)
        val = (O == 13) ? ph : 0.0;
      else if ((col_flag >> 3) & 1u)
        val = 0.0;
      stage_row[O] = val;
    }


    // (phi,u) cell tables C^{dk}[al][g1][g2] of one (cell, derivative axis kk) pair for the column component d.
    // Local frame: axis 0 = kk, axes 1 < 2 the two others.  Everything that depends on the role of d in that
    // frame is expressed through per-lane strides and 0/1 masks, so the quadrature loop has no branches;
    // ANYDIAG = some lane of the wave has d == kk (trace and pressure terms present).
    template <bool ANYDIAG>
    __device__ __forceinline__ void pu_cell(const double (*__restrict__ s_u)[NH], const double *__restrict__ s_phi,
                                            int h000, int kk, int d, const MatScal &S, double *__restrict__ out)
    {
      const int ai = (kk == 0) ? 1 : 0, aj = (kk == 2) ? 1 : 2;
      const bool diag = d == kk;
      auto hstride = [](int a) { return a == 0 ? 1 : (a == 1 ? HX : HX * HY); };
      const int st0 = hstride(kk), st1 = hstride(ai), st2 = hstride(aj);
      const double ih0 = kk == 0 ? S.ih[0] : (kk == 1 ? S.ih[1] : S.ih[2]);
      // frame of u_k for d/dx_d u_k: derivative axis d, remaining axis aB (diag lanes: any valid frame, masked below)
      const int aB = diag ? aj : 3 - kk - d;
      const int stD = diag ? st1 : hstride(d), stB = hstride(aB);
      const bool Bis2 = aB == aj; // the weight of axis aB is n(q2), otherwise n(q1)
      const double ihd = S.ih[d];
      const double *uk = s_u[0] + kk * NH + h000, *ud = s_u[d] + h000, *ph = s_phi + h000;

      double dk[4], dd[4], PH[8];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        {
          const int b1 = e & 1, b2 = e >> 1;
          dk[e] = (ud[st0 + b1 * st1 + b2 * st2] - ud[b1 * st1 + b2 * st2]) * ih0;  // d/dx_k u_d at (b1,b2)
          dd[e] = (uk[b1 * st0 + stD + b2 * stB] - uk[b1 * st0 + b2 * stB]) * ihd;   // d/dx_d u_k at (b0,bB)
        }
#pragma unroll
      for (int b = 0; b < 8; ++b)
        PH[b] = ph[(b & 1) * st0 + ((b >> 1) & 1) * st1 + ((b >> 2) & 1) * st2];
      const double c_muh = 2.0 * (1.0 - S.kappa) * S.mu;
      double di[4], dj[4], cl = 0.0, cd = 0.0;
      if constexpr (ANYDIAG)
        {
          const double *ui = s_u[0] + ai * NH + h000, *uj = s_u[0] + aj * NH + h000;
          const double ih1 = ai == 0 ? S.ih[0] : S.ih[1], ih2 = aj == 1 ? S.ih[1] : S.ih[2];
          const double mi = diag ? ih1 : 0.0, mj = diag ? ih2 : 0.0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            {
              const int b0 = e & 1, bo = e >> 1;
              di[e] = (ui[b0 * st0 + st1 + bo * st2] - ui[b0 * st0 + bo * st2]) * mi; // d/dx_i u_i at (b0,b2)
              dj[e] = (uj[b0 * st0 + bo * st1 + st2] - uj[b0 * st0 + bo * st1]) * mj; // d/dx_j u_j at (b0,b1)
            }
          cl = diag ? 2.0 * (1.0 - S.kappa) * S.lam : 0.0;
          cd = diag ? -2.0 * S.aB1 * S.p : 0.0;
        }
      double wn[2][3]; // vol * w(q0) * n_al(q0)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        {
          wn[0][q] = S.vol * c_g1.w[q] * c_g1.n[0][q];
          wn[1][q] = S.vol * c_g1.w[q] * c_g1.n[1][q];
        }
      double t1[2][3][3]; // [al][q2][q1], the q0 contraction done on the fly
#pragma unroll
      for (int q2 = 0; q2 < 3; ++q2)
        {
          const double n2a = c_g1.n[0][q2], n2b = c_g1.n[1][q2];
          double s0 = 0.0, s1 = 0.0;
          if constexpr (ANYDIAG)
            {
              s0 = n2a * di[0] + n2b * di[2];
              s1 = n2a * di[1] + n2b * di[3];
            }
#pragma unroll
          for (int q1 = 0; q1 < 3; ++q1)
            {
              const double n1a = c_g1.n[0][q1], n1b = c_g1.n[1][q1];
              const double w00 = n1a * n2a, w10 = n1b * n2a, w01 = n1a * n2b, w11 = n1b * n2b;
              const double gk = w00 * dk[0] + w10 * dk[1] + w01 * dk[2] + w11 * dk[3];
              const double p0 = w00 * PH[0] + w10 * PH[2] + w01 * PH[4] + w11 * PH[6];
              const double p1 = w00 * PH[1] + w10 * PH[3] + w01 * PH[5] + w11 * PH[7];
              const double nBa = Bis2 ? n2a : n1a, nBb = Bis2 ? n2b : n1b;
              double r0 = nBa * dd[0] + nBb * dd[2], r1 = nBa * dd[1] + nBb * dd[3];
              double A = c_muh * gk, sv0 = 0.0, sv1 = 0.0;
              if constexpr (ANYDIAG)
                {
                  r0 = diag ? gk : r0; // d == k: d/dx_d u_k is d/dx_k u_d
                  r1 = diag ? gk : r1;
                  sv0 = s0 + (n1a * dj[0] + n1b * dj[2]);
                  sv1 = s1 + (n1a * dj[1] + n1b * dj[3]);
                  A += cl * gk + cd;
                }
              double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
              for (int q0 = 0; q0 < 3; ++q0)
                {
                  const double n0a = c_g1.n[0][q0], n0b = c_g1.n[1][q0];
                  double pf = n0a * p0 + n0b * p1;
                  if (S.monolithic)
                    pf = fmax(0.0, pf); // cracks.cc:2251-2256
                  const double gd = n0a * r0 + n0b * r1;
                  double br = c_muh * gd + A; // 4(1-kappa) mu E_dk
                  if constexpr (ANYDIAG)
                    br += cl * (n0a * sv0 + n0b * sv1); // + [2(1-kappa) lambda trE - 2(alpha_B-1) p] delta_dk
                  const double Phi = pf * br;
                  acc0 += Phi * wn[0][q0];
                  acc1 += Phi * wn[1][q0];
                }
              t1[0][q2][q1] = acc0;
              t1[1][q2][q1] = acc1;
            }
        }
      double wm[3][3]; // w(q) m_g(q)
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          wm[g][q] = c_g1.w[q] * c_g1.m[g][q];
#pragma unroll
      for (int al = 0; al < 2; ++al)
#pragma unroll
        for (int g1 = 0; g1 < 3; ++g1)
          {
            double t2[3];
#pragma unroll
            for (int q2 = 0; q2 < 3; ++q2)
              t2[q2] = t1[al][q2][0] * wm[g1][0] + t1[al][q2][1] * wm[g1][1] + t1[al][q2][2] * wm[g1][2];
#pragma unroll
            for (int g2 = 0; g2 < 3; ++g2)
              out[(al * 9 + g1 * 3 + g2) * CS] = t2[0] * wm[g2][0] + t2[1] * wm[g2][1] + t2[2] * wm[g2][2];
          }
    }

#define PFM_FOR_WAVE_SLOTS(W, X) \
  if constexpr (W == 0) { X(13); } \
  else if constexpr (W == 1) { X(4); X(22); } \
  else if constexpr (W == 2) { X(10); X(16); } \
  else if constexpr (W == 3) { X(12); X(14); } \
  else if constexpr (W == 4) { X(1); X(3); X(5); X(7); } \
  else if constexpr (W == 5) { X(9); X(11); X(15); X(17); } \
  else if constexpr (W == 6) { X(19); X(21); X(23); X(25); } \
  else { X(0); X(2); X(6); X(8); X(18); X(20); X(24); X(26); }

    template <int W>
    __device__ __forceinline__ void pu_wave(const double *lane_base, const MatScal &S, double *stage_row, int d,
                                            unsigned row_flag, const unsigned char *nbf)
    {
#define PFM_X(O) pu_slot<O>(lane_base, S, stage_row, d, row_flag, nbf[((O) % 3 - 1) + HX * (((O) / 3) % 3 - 1) + HX * HY * ((O) / 9 - 1)])
      PFM_FOR_WAVE_SLOTS(W, PFM_X)
#undef PFM_X
    }

    template <int W>
    __device__ __forceinline__ void pp_wave(const double *lane_base, double *stage_row, unsigned row_flag,
                                            const unsigned char *nbf, double &uu_patch)
    {
#define PFM_X(O) pp_slot<O>(lane_base, stage_row, row_flag, nbf[((O) % 3 - 1) + HX * (((O) / 3) % 3 - 1) + HX * HY * ((O) / 9 - 1)], uu_patch)
      PFM_FOR_WAVE_SLOTS(W, PFM_X)
#undef PFM_X
    }

    // =====================================================================================
    template <int NCOL, bool CLK = false /* profiling only: per-phase cycle counts of wave 0 */>
    __global__ __launch_bounds__(NTHREADS) void k_cart_phi(DevView v, CartView cv, MatScal S,
                                                           double *__restrict__ vals_pu, double *__restrict__ vals_pp,
                                                           double *__restrict__ vals_uu,
                                                           unsigned long long *__restrict__ dbg)
    {
      long long tclk = 0;
      auto stamp = [&](int phase) __attribute__((always_inline)) {
        if constexpr (CLK)
          {
            const long long now = clock64();
            if (threadIdx.x == 0 && phase >= 0)
              atomicAdd(dbg + phase, (unsigned long long)(now - tclk));
            tclk = now;
          }
      };
      stamp(-1);
      __shared__ double s_buf[NNUM_PHI * CS];
      __shared__ double s_stage_pu[TX * TY * STG_PU];
      __shared__ double s_stage_pp[TX * TY * STG_PP];
      __shared__ double s_u[3][NH], s_phi[NH], s_po[NH], s_poo[NH];
      __shared__ int s_node[NH];
      __shared__ unsigned char s_flag[NH];
      __shared__ long long s_off[TX * TY]; // node-graph offset of the row, -1 = not an owned node of the tile
      __shared__ int s_deg[TX * TY];
      __shared__ unsigned char s_inv[TX * TY * 27];
      __shared__ int s_irregular; // number of tile nodes that are not owned interior nodes with identity slot order

      const int t = threadIdx.x;
      if (t == 0)
        s_irregular = 0;
      __syncthreads();
      const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1;
      const int ntx = (OWX + TX - 1) / TX, nty = (OWY + TY - 1) / TY;
      const int bid = blockIdx.x;
      const int tix = bid % ntx, tiy = (bid / ntx) % nty, tk = bid / (ntx * nty);
      const int i0 = cv.o0[0] + tix * TX, j0 = cv.o0[1] + tiy * TY, k0 = cv.o0[2] + tk;

      // ---- phase 0
      if (t < NH)
        {
          const int li = t % HX, lj = (t / HX) % HY, lk = t / (HX * HY);
          const int gi = i0 - 1 + li, gj = j0 - 1 + lj, gk = k0 - 1 + lk;
          int n = -1;
          double uu[3] = {0, 0, 0}, ph = 0.0, a = 0.0, b = 0.0;
          unsigned char f = 0;
          if (gi >= 0 && gi < cv.NX && gj >= 0 && gj < cv.NY && gk >= 0 && gk < cv.NZ)
            {
              n = cv.local_of_box[gi + (long long)cv.NX * (gj + (long long)cv.NY * gk)];
              uu[0] = v.u[0][n];
              uu[1] = v.u[1][n];
              uu[2] = v.u[2][n];
              ph = v.phi[n];
              a = v.phi_old[n];
              b = v.phi_oldold[n];
              f = v.node_flags[n];
            }
          s_node[t] = n;
          s_u[0][t] = uu[0];
          s_u[1][t] = uu[1];
          s_u[2][t] = uu[2];
          s_phi[t] = ph;
          s_po[t] = a;
          s_poo[t] = b;
          s_flag[t] = f;
        }
      else if (t >= 320 && t < 320 + TX * TY)
        {
          const int nl = t - 320, li = nl % TX, lj = nl / TX;
          const int gi = i0 + li, gj = j0 + lj;
          long long off = -1;
          int deg = 0;
          bool regular = false;
          if (gi <= cv.o1[0] && gj <= cv.o1[1])
            {
              const int r = cv.local_of_box[gi + (long long)cv.NX * (gj + (long long)cv.NY * k0)];
              off = v.nadj_ptr[r];
              deg = (int)(v.nadj_ptr[r + 1] - off);
              regular = deg == 27;
              for (int s = 0; s < 27; ++s)
                {
                  const unsigned char o = cv.inv27[(long long)r * 27 + s];
                  s_inv[nl * 27 + s] = o;
                  regular = regular && o == s;
                }
            }
          s_off[nl] = off;
          s_deg[nl] = deg;
          if (!regular)
            atomicAdd(&s_irregular, 1);
        }
      __syncthreads();
      stamp(0);

      const int wave = t >> 6, lane = t & 63;
      const int ti = lane % TX, tj = lane / TX;
      const int hc = (ti + 1) + HX * ((tj + 1) + HY * 1);
      const bool owned = (i0 + ti) <= cv.o1[0] && (j0 + tj) <= cv.o1[1];
      const unsigned row_flag = s_flag[hc];
      const unsigned char *nbf = s_flag + hc; // flags of the 27 neighbours, read per slot (saves 27 live VGPRs)
      const double *lane_base = s_buf + (CX * CY) + (tj + 1) * CX + (ti + 1);

      // ---- (phi,u): one sub-phase per column component d
#pragma unroll 1
      for (int d = 0; d < 3; ++d)
        {
          // Launder the thread id: everything below is invariant in d and would otherwise be hoisted out of
          // the loop and kept live across the three sub-phases (register spills = scratch traffic to HBM).
          int tq = t;
          asm volatile("" : "+v"(tq));
          if (tq < 3 * CS)
            {
              const int cs = tq % CS, kk = tq / CS; // derivative axis k of this thread
              const int l = cs / (CX * CY), cy = (cs % (CX * CY)) / CX, cx = cs % CX;
              const int h000 = cx + HX * (cy + HY * l);
              const bool valid = s_node[h000] >= 0 && s_node[h000 + 1 + HX + HX * HY] >= 0;
              double *out = s_buf + (kk * 18) * CS + cs;
              if (!valid)
                {
                  for (int m = 0; m < 18; ++m)
                    out[m * CS] = 0.0;
                }
              else if (__any(d == kk))
                pu_cell<true>(s_u, s_phi, h000, kk, d, S, out);
              else
                pu_cell<false>(s_u, s_phi, h000, kk, d, S, out);
            }
          __syncthreads();
          stamp(1);
          if (owned)
            {
              double *stage_row = s_stage_pu + lane * STG_PU;
              switch (wave)
                {
                  case 0: pu_wave<0>(lane_base, S, stage_row, d, row_flag, nbf); break;
                  case 1: pu_wave<1>(lane_base, S, stage_row, d, row_flag, nbf); break;
                  case 2: pu_wave<2>(lane_base, S, stage_row, d, row_flag, nbf); break;
                  case 3: pu_wave<3>(lane_base, S, stage_row, d, row_flag, nbf); break;
                  case 4: pu_wave<4>(lane_base, S, stage_row, d, row_flag, nbf); break;
                  case 5: pu_wave<5>(lane_base, S, stage_row, d, row_flag, nbf); break;
                  case 6: pu_wave<6>(lane_base, S, stage_row, d, row_flag, nbf); break;
                  default: pu_wave<7>(lane_base, S, stage_row, d, row_flag, nbf); break;
                }
            }
          __syncthreads();
          stamp(2);
        }

      // ---- (phi,phi) cell phase, step 1: thread <-> (cell, qz) : partial moments over (qx,qy)
      if (t < 3 * CS)
        {
          const int cs = t % CS, qz = t / CS;
          const int l = cs / (CX * CY), cy = (cs % (CX * CY)) / CX, cx = cs % CX;
          const int h000 = cx + HX * (cy + HY * l);
          const bool valid = s_node[h000] >= 0 && s_node[h000 + 1 + HX + HX * HY] >= 0;
          double *out = s_buf + cs;
          if (!valid)
            {
              for (int m = 0; m < 9; ++m)
                out[(N_PART + qz * 9 + m) * CS] = 0.0;
            }
          else
            {
              double U[6][8]; // ux, uy, uz, phi, phi_old, phi_oldold
#pragma unroll
              for (int b = 0; b < 8; ++b)
                {
                  const int hb = h000 + (b & 1) + HX * ((b >> 1) & 1) + HX * HY * ((b >> 2) & 1);
                  U[0][b] = s_u[0][hb];
                  U[1][b] = s_u[1][hb];
                  U[2][b] = s_u[2][hb];
                  U[3][b] = s_phi[hb];
                  U[4][b] = s_po[hb];
                  U[5][b] = s_poo[hb];
                }
              const double nz0 = c_g1.n[0][qz], nz1 = c_g1.n[1][qz];
              double Pl[6][4], Dz[3][4];
#pragma unroll
              for (int f = 0; f < 6; ++f)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                  Pl[f][b] = nz0 * U[f][b] + nz1 * U[f][b + 4];
#pragma unroll
              for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                  Dz[f][b] = (U[f][b + 4] - U[f][b]) * S.ih[2];
              double part[3][3]; // [gy][gx] partial moments
#pragma unroll
              for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
                  part[a][b] = 0.0;
#pragma unroll
              for (int qy = 0; qy < 3; ++qy)
                {
                  const double ny0 = c_g1.n[0][qy], ny1 = c_g1.n[1][qy];
                  double L[6][2], Dy[3][2], DzL[3][2];
#pragma unroll
                  for (int f = 0; f < 6; ++f)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                      L[f][b] = ny0 * Pl[f][b] + ny1 * Pl[f][b + 2];
#pragma unroll
                  for (int f = 0; f < 3; ++f)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                      {
                        Dy[f][b] = (Pl[f][b + 2] - Pl[f][b]) * S.ih[1];
                        DzL[f][b] = ny0 * Dz[f][b] + ny1 * Dz[f][b + 2];
                      }
                  double Dx[3];
#pragma unroll
                  for (int f = 0; f < 3; ++f)
                    Dx[f] = (L[f][1] - L[f][0]) * S.ih[0];
                  double rowx[3] = {0.0, 0.0, 0.0}; // moments over qx for this qy
#pragma unroll
                  for (int qx = 0; qx < 3; ++qx)
                    {
                      const double nx0 = c_g1.n[0][qx], nx1 = c_g1.n[1][qx];
                      const double JxW = S.vol * c_g1.w[qx] * c_g1.w[qy] * c_g1.w[qz];
                      double gu[3][3];
#pragma unroll
                      for (int c = 0; c < 3; ++c)
                        {
                          gu[c][0] = Dx[c];
                          gu[c][1] = nx0 * Dy[c][0] + nx1 * Dy[c][1];
                          gu[c][2] = nx0 * DzL[c][0] + nx1 * DzL[c][1];
                        }
                      double pf = nx0 * L[3][0] + nx1 * L[3][1];
                      double pfo = nx0 * L[4][0] + nx1 * L[4][1];
                      if (S.monolithic)
                        {
                          pf = fmax(0.0, pf);
                          pfo = fmax(0.0, pfo);
                        }
                      double trE = 0.0, EE = 0.0;
#pragma unroll
                      for (int a = 0; a < 3; ++a)
                        {
                          trE += gu[a][a];
#pragma unroll
                          for (int b = 0; b < 3; ++b)
                            {
                              const double e = 0.5 * (gu[a][b] + gu[b][a]);
                              EE += e * e;
                            }
                        }
                      const double spE = S.lam * trE * trE + 2 * S.mu * EE; // sigma+ : E
                      const double pen = ((pf - pfo) < 0.0) ? 0.0 : S.gamma_fac; // cracks.cc:2311-2315, 2370
                      const double cq = (1 - S.kappa) * spE + S.Gc / S.eps - 2.0 * S.aB1 * S.p * trE + pen;
                      const double wc = JxW * cq;
                      rowx[0] += wc * c_g1.m[0][qx];
                      rowx[1] += wc * c_g1.m[1][qx];
                      rowx[2] += wc * c_g1.m[2][qx];
                    }
#pragma unroll
                  for (int gy = 0; gy < 3; ++gy)
#pragma unroll
                    for (int gx = 0; gx < 3; ++gx)
                      part[gy][gx] += rowx[gx] * c_g1.m[gy][qy];
                }
#pragma unroll
              for (int gy = 0; gy < 3; ++gy)
#pragma unroll
                for (int gx = 0; gx < 3; ++gx)
                  out[(N_PART + qz * 9 + gy * 3 + gx) * CS] = part[gy][gx];
            }
        }
      __syncthreads();
      stamp(3);
      // ---- (phi,phi) cell phase, step 2: thread <-> (cell, gz): finish the z moment, add the Laplace part
      if (t < 3 * CS)
        {
          const int cs = t % CS, gz = t / CS;
          double *out = s_buf + cs;
          const int l = cs / (CX * CY), cy = (cs % (CX * CY)) / CX, cx = cs % CX;
          const int h000 = cx + HX * (cy + HY * l);
          const bool valid = s_node[h000] >= 0 && s_node[h000 + 1 + HX + HX * HY] >= 0;
          // 1-D mass moments mbar_g = sum_q w m_g(q)
          double mb[3];
#pragma unroll
          for (int g = 0; g < 3; ++g)
            mb[g] = c_g1.w[0] * c_g1.m[g][0] + c_g1.w[1] * c_g1.m[g][1] + c_g1.w[2] * c_g1.m[g][2];
          const double lap = S.Gc * S.eps * S.vol;
          // per-thread z factors; an invalid cell gets an all-zero table (its partial moments are zero already)
          const double mbz = valid ? mb[gz] : 0.0;
          const double szz = valid ? ((gz == 1) ? -lap * S.ih[2] * S.ih[2] : lap * S.ih[2] * S.ih[2]) : 0.0;
          const double mz0 = c_g1.m[gz][0], mz1 = c_g1.m[gz][1], mz2 = c_g1.m[gz][2];
#pragma unroll
          for (int gy = 0; gy < 3; ++gy)
#pragma unroll
            for (int gx = 0; gx < 3; ++gx)
              {
                double m = out[(N_PART + 0 * 9 + gy * 3 + gx) * CS] * mz0;
                m += out[(N_PART + 1 * 9 + gy * 3 + gx) * CS] * mz1;
                m += out[(N_PART + 2 * 9 + gy * 3 + gx) * CS] * mz2;
                const double sx = (gx == 1) ? -1.0 : 1.0, sy = (gy == 1) ? -1.0 : 1.0;
                // G_c eps sum_q w grad N_a . grad N_b  (sign = -1 where a_k != b_k):
                //   L = mbar_gz * P[gy][gx] + s_z ih_z^2 * Q[gy][gx], P and Q the same for every thread
                const double P = lap * (sx * S.ih[0] * S.ih[0] * mb[gy] + sy * S.ih[1] * S.ih[1] * mb[gx]);
                const double Q = mb[gx] * mb[gy];
                m += mbz * P;
                m += szz * Q;
                out[(N_M + gx + 3 * gy + 9 * gz) * CS] = m;
              }
        }
      __syncthreads();
      stamp(4);
      // ---- (phi,phi) cell phase, step 3: mean |diagonal| of the element matrix (deal.II uses it as
      // the placeholder of a constrained row whose own diagonal entry is exactly zero)
      if (t < CS)
        {
          const int cs = t;
          double *out = s_buf + cs;
          const int l = cs / (CX * CY), cy = (cs % (CX * CY)) / CX, cx = cs % CX;
          const int h000 = cx + HX * (cy + HY * l);
          const bool valid = s_node[h000] >= 0 && s_node[h000 + 1 + HX + HX * HY] >= 0;
          double avg = 0.0, gzero = 0.0;
          // avg is consumed only by constrained rows (placeholder of a zero diagonal entry), gzero only by
          // constrained displacement rows; with 0 < kappa <= 1 every g(q) >= kappa > 0 so gzero stays 0.
          bool needed = false;
          double dsum = 0.0; // sum_a |K_phiphi[a,a]|
          if (valid)
            {
              unsigned anyflag = 0;
              bool zero_diag = false;
#pragma unroll
              for (int a = 0; a < 8; ++a)
                {
                  const double dg = fabs(out[(N_M + 2 * (a & 1) + 3 * 2 * ((a >> 1) & 1) + 9 * 2 * ((a >> 2) & 1)) * CS]);
                  dsum += dg;
                  zero_diag = zero_diag || dg == 0.0;
                  anyflag |= s_flag[h000 + (a & 1) + HX * ((a >> 1) & 1) + HX * HY * ((a >> 2) & 1)];
                }
              const bool g_positive = S.kappa > 0.0 && S.kappa <= 1.0;
              needed = anyflag != 0 && (zero_diag || !g_positive);
            }
          if (needed)
            {
              // sum_{a,c} K_uu[(a,c),(a,c)] = sum_k (sum_c cA[c][k]) * 2 * sum_q w g mu(q_i) mu(q_j),  mu = m_00 + m_11
              double po[8], poo[8];
#pragma unroll
              for (int b = 0; b < 8; ++b)
                {
                  const int hb = h000 + (b & 1) + HX * ((b >> 1) & 1) + HX * HY * ((b >> 2) & 1);
                  po[b] = s_po[hb];
                  poo[b] = s_poo[hb];
                }
              double gsum = 0.0, gk[3] = {0.0, 0.0, 0.0};
#pragma unroll
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
                usum += (S.cA[0][k] + S.cA[1][k] + S.cA[2][k]) * 2.0 * gk[k];
              avg = (dsum + usum) / 32.0;
              gzero = (gsum == 0.0) ? 1.0 : 0.0;
            }
          out[N_AVG * CS] = avg;
          out[N_GZERO * CS] = gzero;
        }
      __syncthreads();
      stamp(5);

      // ---- (phi,phi) node phase
      double uu_patch = 0.0;
      if (owned)
        {
          double *stage_row = s_stage_pp + lane * STG_PP;
          switch (wave)
            {
              case 0: pp_wave<0>(lane_base, stage_row, row_flag, nbf, uu_patch); break;
              case 1: pp_wave<1>(lane_base, stage_row, row_flag, nbf, uu_patch); break;
              case 2: pp_wave<2>(lane_base, stage_row, row_flag, nbf, uu_patch); break;
              case 3: pp_wave<3>(lane_base, stage_row, row_flag, nbf, uu_patch); break;
              case 4: pp_wave<4>(lane_base, stage_row, row_flag, nbf, uu_patch); break;
              case 5: pp_wave<5>(lane_base, stage_row, row_flag, nbf, uu_patch); break;
              case 6: pp_wave<6>(lane_base, stage_row, row_flag, nbf, uu_patch); break;
              default: pp_wave<7>(lane_base, stage_row, row_flag, nbf, uu_patch); break;
            }
          // wave 0 owns the centre slot: fix the placeholder of constrained displacement rows whose
          // element diagonal vanished in some cell (kernel k_cart_uu wrote the plain sum earlier in the stream)
          if (wave == 0 && (row_flag & 7u) && uu_patch != 0.0)
            {
              const int nl = lane;
              const long long off = s_off[nl];
              const int deg = s_deg[nl];
              int sself = 0;
              for (int s = 0; s < deg; ++s)
                if (s_inv[nl * 27 + s] == 13)
                  sself = s;
              for (int c = 0; c < 3; ++c)
                if ((row_flag >> c) & 1u)
                  vals_uu[(long long)NCOL * NCOL * off + (long long)c * NCOL * deg + sself * NCOL + c] += uu_patch;
            }
        }
      __syncthreads();
      stamp(6);

      // ---- copy-out of the phase-field rows
      if (NCOL == 3 && s_irregular == 0)
        {
          // interior tile: slot order = offset order and every row is full, the staged rows are the CSR rows
#pragma unroll 2
          for (int f = t; f < TX * TY * STG_PU; f += NTHREADS)
            {
              const int nl = f / STG_PU;
              vals_pu[3 * s_off[nl] + (f - nl * STG_PU)] = s_stage_pu[f];
            }
#pragma unroll 2
          for (int f = t; f < TX * TY * STG_PP; f += NTHREADS)
            {
              const int nl = f / STG_PP;
              vals_pp[s_off[nl] + (f - nl * STG_PP)] = s_stage_pp[f];
            }
        }
      else if constexpr (NCOL == 3)
        {
          for (int f = t; f < TX * TY * STG_PU; f += NTHREADS)
            {
              const int nl = f / STG_PU, e = f - nl * STG_PU;
              const int s = e / 3, d = e - s * 3;
              const long long off = s_off[nl];
              const int deg = s_deg[nl];
              if (off < 0 || s >= deg)
                continue;
              const int o = s_inv[nl * 27 + s];
              vals_pu[3 * off + s * 3 + d] = s_stage_pu[nl * STG_PU + o * 3 + d];
            }
          for (int f = t; f < TX * TY * STG_PP; f += NTHREADS)
            {
              const int nl = f / STG_PP, s = f - nl * STG_PP;
              const long long off = s_off[nl];
              const int deg = s_deg[nl];
              if (off < 0 || s >= deg)
                continue;
              const int o = s_inv[nl * 27 + s];
              vals_pp[off + s] = s_stage_pp[nl * STG_PP + o];
            }
        }
      else
        {
          // interleaved layout: row (node, 3) holds [u_x u_y u_z phi] per neighbour slot
          for (int f = t; f < TX * TY * 27 * 4; f += NTHREADS)
            {
              const int nl = f / 108, e = f - nl * 108;
              const int s = e / 4, d = e - s * 4;
              const long long off = s_off[nl];
              const int deg = s_deg[nl];
              if (off < 0 || s >= deg)
                continue;
              const int o = s_inv[nl * 27 + s];
              const double val = (d < 3) ? s_stage_pu[nl * STG_PU + o * 3 + d] : s_stage_pp[nl * STG_PP + o];
              vals_uu[16 * off + (long long)3 * 4 * deg + s * 4 + d] = val;
            }
        }
      if constexpr (CLK)
        {
          __syncthreads();
          stamp(7);
        }
    }
  } // namespace

  bool cart_matrix_supported(int dim) { return dim == 3; }

  int launch_cart_matrix(const DevView &v, const CartView &cv, const pfm_params &p, double *const *d_values, hipStream_t s)
  {
    if (v.dim != 3)
      return PFM_ERR_UNSUPPORTED;
    int rc = ensure_g1();
    if (rc)
      return rc;
    // (u,u) rows first: k_cart_phi patches constrained diagonals afterwards (same stream)
    rc = launch_cart_uu_only(v, cv, p, d_values[0], s);
    if (rc)
      return rc;
    if (!getenv("PFM_PHI_V1"))
      return launch_cart_phi4(v, cv, p, d_values, s); // z-marching push kernel (pfm_cart_phi4.hip)
    const MatScal S = make_mat_scal(p, cv);
    const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1, OWZ = cv.o1[2] - cv.o0[2] + 1;
    const int ntx = (OWX + TX - 1) / TX, nty = (OWY + TY - 1) / TY;
    const unsigned nb = (unsigned)(ntx * nty * OWZ);
    if (v.layout == PFM_LAYOUT_INTERLEAVED)
      hipLaunchKernelGGL(k_cart_phi<4>, dim3(nb), dim3(NTHREADS), 0, s, v, cv, S, nullptr, nullptr, d_values[0], nullptr);
    else if (getenv("PFM_PHI_CLK")) // profiling only
      {
        static unsigned long long *d_dbg = nullptr;
        if (!d_dbg && hipMalloc((void **)&d_dbg, 16 * sizeof(unsigned long long)) != hipSuccess)
          return PFM_ERR_HIP;
        (void)hipMemsetAsync(d_dbg, 0, 16 * sizeof(unsigned long long), s);
        hipLaunchKernelGGL((k_cart_phi<3, true>), dim3(nb), dim3(NTHREADS), 0, s, v, cv, S, d_values[2], d_values[3],
                           d_values[0], d_dbg);
        unsigned long long h[16];
        (void)hipMemcpy(h, d_dbg, sizeof(h), hipMemcpyDeviceToHost);
        const char *names[8] = {"phase0", "pu-cell(x3)", "pu-node(x3)", "pp-step1", "pp-step2", "pp-step3", "pp-node", "copy-out"};
        fprintf(stderr, "[k_cart_phi phase clock, thread 0, cycles per tile]");
        for (int i = 0; i < 8; ++i)
          fprintf(stderr, " %s=%.0f", names[i], (double)h[i] / nb);
        fprintf(stderr, "\n");
      }
    else
      {
        // the structurally zero (u,phi) block (cracks.cc:2333-2337) is cleared by the host side
        hipLaunchKernelGGL(k_cart_phi<3>, dim3(nb), dim3(NTHREADS), 0, s, v, cv, S, d_values[2], d_values[3], d_values[0],
                           nullptr);
      }
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }
} // namespace pfm
