// pfm_cart_matrix.hip — row-owner Jacobian kernels of the cartesian family (3-D).
// See the header of pfm_cart.hip for the design; this file holds
//   k_cart_uu   : the (u,u) block            (cracks.cc:2356-2366, 56 % of the matrix bytes)
//   k_cart_phi  : the (phi,u) and (phi,phi) blocks (cracks.cc:2367-2384) + constrained diagonals
// The (u,phi) block is structurally zero (cracks.cc:2333-2337) and is cleared by a memset.
//
// Tile = 8 x 8 owned nodes of one lattice plane; 512 threads.
//   phase 0  stage the nodal inputs of the 10 x 10 x 3 neighbourhood in LDS
//   phase 1  cell phase: 2 x 9 x 9 cells, 3 threads per cell, moment tables -> LDS (SoA over cells)
//   phase 2  node phase: lane <-> node, wave <-> set of neighbour slots (balanced: every wave
//            visits 8 cell contributions per row), compile-time LDS offsets, rows staged in LDS
//   phase 3  copy-out: flat, coalesced stores of the staged rows into the CSR value array
#include "pfm_internal.h"

#include <hip/hip_runtime.h>
#include <type_traits>

namespace pfm
{
  namespace
  {
    constexpr int TX = 8, TY = 8, NTHREADS = 512;
    constexpr int HX = TX + 2, HY = TY + 2;      // nodal halo (10 x 10 x 3)
    constexpr int NH = HX * HY * 3;              // 300 halo nodes
    constexpr int CX = TX + 1, CY = TY + 1;      // cells per layer (9 x 9)
    constexpr int CS = CX * CY * 2;              // 162 cell slots (two layers)
    constexpr int NNUM_UU = 64;                  // 27 A + 36 T + 1 spare
    constexpr int STG = 81;                      // staged row width (27 slots x 3), odd => conflict-free

    struct G1
    {
      double n[2][3], m[3][3], w[3]; // n_al(q), m_g(q) (g = 0:00, 1:01, 2:11), weights
    };
    __constant__ G1 c_g1;

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
      return t;
    }

    struct MatScal
    {
      double lam, mu, kappa, eps, Gc, p, aB1, gamma_fac, tfac;
      double ih[3], vol;
      double cA[3][3]; // cA[c][k] = (k == c ? lam + 2 mu : mu) / h_k^2
      double cT[3];    // 1 / (h_lo h_hi) for the pairs (0,1), (0,2), (1,2)
      int monolithic, use_old;
    };

    // ---- compile-time index helpers ------------------------------------------------------
    __host__ __device__ constexpr int idxA(int c, int gi, int gj) { return c * 9 + gi * 3 + gj; }
    __host__ __device__ constexpr int pair_of(int lo, int hi) { return lo == 0 ? (hi == 1 ? 0 : 1) : 2; }
    __host__ __device__ constexpr int idxT(int p, int al, int be, int g) { return 27 + p * 12 + al * 6 + be * 3 + g; }
    __host__ __device__ constexpr int third_axis(int c, int d) { return 3 - c - d; }
    __host__ __device__ constexpr int sgn(int bit) { return bit ? 1 : -1; }

    // K_uu[(a,c),(b,d)] of one cell from the moment tables held in LDS.
    // lds = address of number 0 for this lane's cell; numbers are CS doubles apart.
    template <int C, int D, int AX, int AY, int AZ, int BX, int BY, int BZ>
    __device__ __forceinline__ double kuu_from_tables(const double *__restrict__ lds, const MatScal &S)
    {
      constexpr int a[3] = {AX, AY, AZ}, b[3] = {BX, BY, BZ};
      constexpr int g[3] = {AX + BX, AY + BY, AZ + BZ};
      if constexpr (C == D)
        {
          double r = 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k)
            {
              const int i = (k == 0) ? 1 : 0, j = (k == 2) ? 1 : 2; // the two other axes, ascending
              const double t = lds[idxA(k, g[i], g[j]) * CS];
              r += (double)(sgn(a[k]) * sgn(b[k])) * S.cA[C][k] * t;
            }
          return r;
        }
      else
        {
          constexpr int lo = C < D ? C : D, hi = C < D ? D : C, e = third_axis(C, D), p = pair_of(lo, hi);
          // G^{CD}: derivative C on a, D on b;  G^{DC}: derivative D on a, C on b
          // T^p[al][be][g] = sum wg n_al(q_lo) n_be(q_hi) m_g(q_e)
          constexpr int al1 = (C < D) ? b[lo] : a[lo], be1 = (C < D) ? a[hi] : b[hi]; // for G^{CD}
          constexpr int al2 = (C < D) ? a[lo] : b[lo], be2 = (C < D) ? b[hi] : a[hi]; // for G^{DC}
          const double t1 = lds[idxT(p, al1, be1, g[e]) * CS];
          const double t2 = lds[idxT(p, al2, be2, g[e]) * CS];
          return S.cT[p] * (S.lam * (double)(sgn(a[C]) * sgn(b[D])) * t1 + S.mu * (double)(sgn(a[D]) * sgn(b[C])) * t2);
        }
    }

    // sum over the cells shared by the node and its neighbour at offset (OX,OY,OZ)
    template <int C, int D, int OX, int OY, int OZ>
    __device__ __forceinline__ double uu_entry(const double *__restrict__ lane_base, const MatScal &S)
    {
      double r = 0.0;
      auto visit = [&](auto EX, auto EY, auto EZ) {
        constexpr int ex = decltype(EX)::value, ey = decltype(EY)::value, ez = decltype(EZ)::value;
        constexpr int ax = -ex, ay = -ey, az = -ez;
        constexpr int bx = ax + OX, by = ay + OY, bz = az + OZ;
        if constexpr (bx >= 0 && bx <= 1 && by >= 0 && by <= 1 && bz >= 0 && bz <= 1)
          r += kuu_from_tables<C, D, ax, ay, az, bx, by, bz>(lane_base + (ez * (CX * CY) + ey * CX + ex), S);
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

    // all three column components of slot O for row component C, with the constraint mask
    template <int C, int O>
    __device__ __forceinline__ void uu_slot(const double *__restrict__ lane_base, const MatScal &S,
                                            double *__restrict__ stage_row, unsigned row_flag, unsigned col_flag)
    {
      constexpr int OX = O % 3 - 1, OY = (O / 3) % 3 - 1, OZ = O / 9 - 1;
      const bool rcon = (row_flag >> C) & 1u;
      double v0 = uu_entry<C, 0, OX, OY, OZ>(lane_base, S);
      double v1 = uu_entry<C, 1, OX, OY, OZ>(lane_base, S);
      double v2 = uu_entry<C, 2, OX, OY, OZ>(lane_base, S);
      // constrained rows keep only their diagonal (deal.II: |K_ii| summed over the cells; the
      // terms are non-negative here), constrained columns are eliminated
      if (rcon || (col_flag & 1u))
        v0 = (rcon && O == 13 && C == 0) ? v0 : 0.0;
      if (rcon || (col_flag & 2u))
        v1 = (rcon && O == 13 && C == 1) ? v1 : 0.0;
      if (rcon || (col_flag & 4u))
        v2 = (rcon && O == 13 && C == 2) ? v2 : 0.0;
      stage_row[O * 3 + 0] = v0;
      stage_row[O * 3 + 1] = v1;
      stage_row[O * 3 + 2] = v2;
    }

    // balanced slot sets: every wave visits 8 cell contributions per (row, column component)
    template <int C, int W>
    __device__ __forceinline__ void uu_wave(const double *__restrict__ lane_base, const MatScal &S,
                                            double *__restrict__ stage_row, unsigned row_flag,
                                            const unsigned char *__restrict__ nb_flags /* [27] col flags */)
    {
#define PFM_SLOT(O) uu_slot<C, O>(lane_base, S, stage_row, row_flag, nb_flags[O])
      if constexpr (W == 0)
        {
          PFM_SLOT(13);
        }
      else if constexpr (W == 1)
        {
          PFM_SLOT(4);
          PFM_SLOT(22);
        }
      else if constexpr (W == 2)
        {
          PFM_SLOT(10);
          PFM_SLOT(16);
        }
      else if constexpr (W == 3)
        {
          PFM_SLOT(12);
          PFM_SLOT(14);
        }
      else if constexpr (W == 4)
        { // edges with dz = -1
          PFM_SLOT(1);
          PFM_SLOT(3);
          PFM_SLOT(5);
          PFM_SLOT(7);
        }
      else if constexpr (W == 5)
        { // edges with dz = 0
          PFM_SLOT(9);
          PFM_SLOT(11);
          PFM_SLOT(15);
          PFM_SLOT(17);
        }
      else if constexpr (W == 6)
        { // edges with dz = +1
          PFM_SLOT(19);
          PFM_SLOT(21);
          PFM_SLOT(23);
          PFM_SLOT(25);
        }
      else
        { // corners
          PFM_SLOT(0);
          PFM_SLOT(2);
          PFM_SLOT(6);
          PFM_SLOT(8);
          PFM_SLOT(18);
          PFM_SLOT(20);
          PFM_SLOT(24);
          PFM_SLOT(26);
        }
#undef PFM_SLOT
    }

    template <int C>
    __device__ __forceinline__ void uu_dispatch(int wave, const double *lane_base, const MatScal &S, double *stage_row,
                                                unsigned row_flag, const unsigned char *nb_flags)
    {
      switch (wave)
        {
          case 0:
            uu_wave<C, 0>(lane_base, S, stage_row, row_flag, nb_flags);
            break;
          case 1:
            uu_wave<C, 1>(lane_base, S, stage_row, row_flag, nb_flags);
            break;
          case 2:
            uu_wave<C, 2>(lane_base, S, stage_row, row_flag, nb_flags);
            break;
          case 3:
            uu_wave<C, 3>(lane_base, S, stage_row, row_flag, nb_flags);
            break;
          case 4:
            uu_wave<C, 4>(lane_base, S, stage_row, row_flag, nb_flags);
            break;
          case 5:
            uu_wave<C, 5>(lane_base, S, stage_row, row_flag, nb_flags);
            break;
          case 6:
            uu_wave<C, 6>(lane_base, S, stage_row, row_flag, nb_flags);
            break;
          default:
            uu_wave<C, 7>(lane_base, S, stage_row, row_flag, nb_flags);
            break;
        }
    }

    // weights w*g(q) of one cell from the nodal phi_old / phi_oldold (cracks.cc:2262-2277)
    __device__ __forceinline__ void cell_wg(const double po[8], const double poo[8], const MatScal &S, double wg[27])
    {
#pragma unroll
      for (int qz = 0; qz < 3; ++qz)
        {
          double a[4], b[4];
#pragma unroll
          for (int v = 0; v < 4; ++v)
            {
              a[v] = c_g1.n[0][qz] * po[v] + c_g1.n[1][qz] * po[v + 4];
              b[v] = c_g1.n[0][qz] * poo[v] + c_g1.n[1][qz] * poo[v + 4];
            }
#pragma unroll
          for (int qy = 0; qy < 3; ++qy)
            {
              const double a0 = c_g1.n[0][qy] * a[0] + c_g1.n[1][qy] * a[2];
              const double a1 = c_g1.n[0][qy] * a[1] + c_g1.n[1][qy] * a[3];
              const double b0 = c_g1.n[0][qy] * b[0] + c_g1.n[1][qy] * b[2];
              const double b1 = c_g1.n[0][qy] * b[1] + c_g1.n[1][qy] * b[3];
#pragma unroll
              for (int qx = 0; qx < 3; ++qx)
                {
                  double pfo = c_g1.n[0][qx] * a0 + c_g1.n[1][qx] * a1;
                  double pfoo = c_g1.n[0][qx] * b0 + c_g1.n[1][qx] * b1;
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
                  wg[qx + 3 * qy + 9 * qz] = S.vol * (c_g1.w[qx] * c_g1.w[qy] * c_g1.w[qz]) * g;
                }
            }
        }
    }

    // =====================================================================================
    __global__ __launch_bounds__(NTHREADS) void k_cart_uu(DevView v, CartView cv, MatScal S, double *__restrict__ vals,
                                                          int ncol /* 3 blocked, 4 interleaved */)
    {
      __shared__ double s_buf[NNUM_UU * CS];
      __shared__ double s_stage[TX * TY * STG];
      __shared__ double s_po[NH], s_poo[NH];
      __shared__ int s_node[NH];
      __shared__ unsigned char s_flag[NH];

      const int t = threadIdx.x;
      const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1;
      const int ntx = (OWX + TX - 1) / TX, nty = (OWY + TY - 1) / TY;
      const int bid = blockIdx.x;
      const int tix = bid % ntx, tiy = (bid / ntx) % nty, tk = bid / (ntx * nty);
      const int i0 = cv.o0[0] + tix * TX, j0 = cv.o0[1] + tiy * TY, k = cv.o0[2] + tk;

      // ---- phase 0: nodal halo
      if (t < NH)
        {
          const int li = t % HX, lj = (t / HX) % HY, lk = t / (HX * HY);
          const int gi = i0 - 1 + li, gj = j0 - 1 + lj, gk = k - 1 + lk;
          int n = -1;
          double a = 0.0, b = 0.0;
          unsigned char f = 0;
          if (gi >= 0 && gi < cv.NX && gj >= 0 && gj < cv.NY && gk >= 0 && gk < cv.NZ)
            {
              n = cv.local_of_box[gi + (long long)cv.NX * (gj + (long long)cv.NY * gk)];
              a = v.phi_old[n];
              b = v.phi_oldold[n];
              f = v.node_flags[n];
            }
          s_node[t] = n;
          s_po[t] = a;
          s_poo[t] = b;
          s_flag[t] = f;
        }
      __syncthreads();

      // ---- phase 1: cell phase (3 threads per cell)
      if (t < 3 * CS)
        {
          const int cs = t % CS, sub = t / CS;
          const int l = cs / (CX * CY), cy = (cs % (CX * CY)) / CX, cx = cs % CX;
          const int h000 = cx + HX * (cy + HY * l);
          const bool valid = s_node[h000] >= 0 && s_node[h000 + 1 + HX + HX * HY] >= 0;
          double *out = s_buf + cs;
          if (!valid)
            {
              if (sub == 0)
                for (int m = 0; m < 27; ++m)
                  out[m * CS] = 0.0;
              else if (sub == 1)
                for (int m = 27; m < 51; ++m)
                  out[m * CS] = 0.0;
              else
                for (int m = 51; m < 64; ++m)
                  out[m * CS] = 0.0;
            }
          else
            {
              double po[8], poo[8], wg[27];
#pragma unroll
              for (int b = 0; b < 8; ++b)
                {
                  const int hb = h000 + (b & 1) + HX * ((b >> 1) & 1) + HX * HY * ((b >> 2) & 1);
                  po[b] = s_po[hb];
                  poo[b] = s_poo[hb];
                }
              cell_wg(po, poo, S, wg);
              if (sub == 0)
                {
                  // A^c[g_i][g_j]: collapse axis c, then moments over the two others
#pragma unroll
                  for (int c = 0; c < 3; ++c)
                    {
                      double s9[3][3]; // [qj][qi], (i,j) = other axes ascending
#pragma unroll
                      for (int qj = 0; qj < 3; ++qj)
#pragma unroll
                        for (int qi = 0; qi < 3; ++qi)
                          {
                            double acc = 0.0;
#pragma unroll
                            for (int qc = 0; qc < 3; ++qc)
                              {
                                const int q = (c == 0) ? (qc + 3 * qi + 9 * qj) : (c == 1) ? (qi + 3 * qc + 9 * qj) : (qi + 3 * qj + 9 * qc);
                                acc += wg[q];
                              }
                            s9[qj][qi] = acc;
                          }
#pragma unroll
                      for (int gi = 0; gi < 3; ++gi)
                        {
                          double tq[3];
#pragma unroll
                          for (int qj = 0; qj < 3; ++qj)
                            tq[qj] = s9[qj][0] * c_g1.m[gi][0] + s9[qj][1] * c_g1.m[gi][1] + s9[qj][2] * c_g1.m[gi][2];
#pragma unroll
                          for (int gj = 0; gj < 3; ++gj)
                            out[idxA(c, gi, gj) * CS] = tq[0] * c_g1.m[gj][0] + tq[1] * c_g1.m[gj][1] + tq[2] * c_g1.m[gj][2];
                        }
                    }
                }
              else
                {
                  // T^p[al][be][g] = sum wg n_al(q_lo) n_be(q_hi) m_g(q_e)
                  auto do_pair = [&](const int lo, const int hi) {
                    const int e = 3 - lo - hi, p = pair_of(lo, hi);
                    const int st[3] = {1, 3, 9};
#pragma unroll
                    for (int al = 0; al < 2; ++al)
                      {
                        double t1[3][3]; // [q_e][q_hi]
#pragma unroll
                        for (int qe = 0; qe < 3; ++qe)
#pragma unroll
                          for (int qh = 0; qh < 3; ++qh)
                            {
                              double acc = 0.0;
#pragma unroll
                              for (int ql = 0; ql < 3; ++ql)
                                acc += wg[ql * st[lo] + qh * st[hi] + qe * st[e]] * c_g1.n[al][ql];
                              t1[qe][qh] = acc;
                            }
#pragma unroll
                        for (int be = 0; be < 2; ++be)
                          {
                            double t2[3];
#pragma unroll
                            for (int qe = 0; qe < 3; ++qe)
                              t2[qe] = t1[qe][0] * c_g1.n[be][0] + t1[qe][1] * c_g1.n[be][1] + t1[qe][2] * c_g1.n[be][2];
#pragma unroll
                            for (int g = 0; g < 3; ++g)
                              out[idxT(p, al, be, g) * CS] = t2[0] * c_g1.m[g][0] + t2[1] * c_g1.m[g][1] + t2[2] * c_g1.m[g][2];
                          }
                      }
                  };
                  if (sub == 1)
                    {
                      do_pair(0, 1);
                      do_pair(0, 2);
                    }
                  else
                    {
                      do_pair(1, 2);
                      out[63 * CS] = 0.0;
                    }
                }
            }
        }
      __syncthreads();

      // ---- phases 2 + 3 per row component
      const int wave = t >> 6, lane = t & 63;
      const int ti = lane % TX, tj = lane / TX;
      const int hc = (ti + 1) + HX * ((tj + 1) + HY * 1); // halo index of this lane's node
      const int ni = i0 + ti, nj = j0 + tj;
      const bool owned = ni <= cv.o1[0] && nj <= cv.o1[1];
      const unsigned row_flag = s_flag[hc];
      unsigned char nbf[27];
#pragma unroll
      for (int o = 0; o < 27; ++o)
        nbf[o] = s_flag[hc + (o % 3 - 1) + HX * ((o / 3) % 3 - 1) + HX * HY * (o / 9 - 1)];
      const double *lane_base = s_buf + (CX * CY) + (tj + 1) * CX + (ti + 1);
      double *stage_row = s_stage + lane * STG;

#pragma unroll 1
      for (int c = 0; c < 3; ++c)
        {
          if (owned)
            {
              if (c == 0)
                uu_dispatch<0>(wave, lane_base, S, stage_row, row_flag, nbf);
              else if (c == 1)
                uu_dispatch<1>(wave, lane_base, S, stage_row, row_flag, nbf);
              else
                uu_dispatch<2>(wave, lane_base, S, stage_row, row_flag, nbf);
            }
          __syncthreads();
          // copy-out: element f of the tile's row-c data, flat over (node, slot, column comp)
          const int rowlen = 27 * ncol;
          for (int f = t; f < TX * TY * rowlen; f += NTHREADS)
            {
              const int nl = f / rowlen, e = f - nl * rowlen;
              const int s = e / ncol, d = e - s * ncol;
              const int li = nl % TX, lj = nl / TX;
              if (i0 + li > cv.o1[0] || j0 + lj > cv.o1[1])
                continue;
              const int r = s_node[(li + 1) + HX * ((lj + 1) + HY)];
              const long long off = v.nadj_ptr[r];
              const int deg = (int)(v.nadj_ptr[r + 1] - off);
              if (s >= deg)
                continue;
              const int o = cv.inv27[(long long)r * 27 + s];
              const double val = (d < 3) ? s_stage[nl * STG + o * 3 + d] : 0.0;
              vals[(long long)ncol * ncol * off + (long long)c * ncol * deg + (long long)s * ncol + d] = val;
            }
          __syncthreads();
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
      s.monolithic = prm.outer_solver == PFM_SOLVER_SIMPLE_MONOLITHIC;
      s.use_old = prm.use_old_timestep_pf;
      return s;
    }
  } // namespace

  bool cart_matrix_supported(int dim) { return false && dim == 3; }

  int launch_cart_uu_only(const DevView &v, const CartView &cv, const pfm_params &p, double *vals_uu, hipStream_t s)
  {
    int rc = ensure_g1();
    if (rc)
      return rc;
    const MatScal S = make_mat_scal(p, cv);
    const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1, OWZ = cv.o1[2] - cv.o0[2] + 1;
    const int ntx = (OWX + TX - 1) / TX, nty = (OWY + TY - 1) / TY;
    const unsigned nb = (unsigned)(ntx * nty * OWZ);
    const int ncol = v.layout == PFM_LAYOUT_INTERLEAVED ? 4 : 3;
    hipLaunchKernelGGL(k_cart_uu, dim3(nb), dim3(NTHREADS), 0, s, v, cv, S, vals_uu, ncol);
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }

  int launch_cart_matrix(const DevView &, const CartView &, const pfm_params &, double *const *, hipStream_t)
  {
    return PFM_ERR_UNSUPPORTED;
  }
} // namespace pfm
