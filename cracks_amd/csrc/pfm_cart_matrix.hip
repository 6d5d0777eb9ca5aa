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
#include "pfm_cart_common.h"

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace pfm
{
  namespace
  {
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

    // node phase of one row component; MASK = the tile touches constrained dofs
    template <int C, bool MASK>
    __device__ __forceinline__ void uu_rows(int wave, const double *lane_base, const MatScal &S, double *stage_row,
                                            unsigned row_flag, const unsigned char *nbf)
    {
      if constexpr (MASK)
        uu_dispatch<C>(wave, lane_base, S, stage_row, row_flag, nbf);
      else
        {
          const unsigned char zero[27] = {};
          uu_dispatch<C>(wave, lane_base, S, stage_row, 0u, zero);
        }
    }

    // =====================================================================================
    template <int NCOL /* 3 blocked, 4 interleaved */, int ABL = 0 /* ablation bits, profiling only */>
    __global__ __launch_bounds__(NTHREADS) void k_cart_uu(DevView v, CartView cv, MatScal S, double *__restrict__ vals,
                                                          unsigned long long *__restrict__ dbg)
    {
      // phase clock (profiling builds only, ABL & 8): wave 0 accumulates cycles per phase
      long long tclk = 0;
      auto stamp = [&](int phase) {
        if constexpr ((ABL & 8) != 0)
          {
            const long long now = clock64();
            if (threadIdx.x == 0 && phase >= 0)
              atomicAdd(dbg + phase, (unsigned long long)(now - tclk));
            tclk = now;
          }
      };
      stamp(-1);
      __shared__ double s_buf[NNUM_UU * CS];
      __shared__ double s_stage[TX * TY * STG]; // rows being staged; holds w*g(q) [27][CS] during the cell phase
      __shared__ double s_po[NH], s_poo[NH];
      __shared__ int s_node[NH];
      __shared__ unsigned char s_flag[NH];
      __shared__ long long s_rowbase[TX * TY]; // first value of the node's row block, -1 = not an owned node of the tile
      __shared__ int s_deg[TX * TY];
      __shared__ unsigned char s_inv[TX * TY * 27];
      __shared__ int s_info[2]; // [0] any constraint flag in the halo, [1] number of irregular nodes
      static_assert(27 * CS <= TX * TY * STG, "w*g scratch must fit in the staging buffer");

      const int t = threadIdx.x;
      const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1;
      const int ntx = (OWX + TX - 1) / TX, nty = (OWY + TY - 1) / TY;
      const int bid = blockIdx.x;
      const int tix = bid % ntx, tiy = (bid / ntx) % nty, tk = bid / (ntx * nty);
      const int i0 = cv.o0[0] + tix * TX, j0 = cv.o0[1] + tiy * TY, k = cv.o0[2] + tk;

      // ---- phase 0: nodal halo + CSR row info of the tile's nodes
      if (t < 2)
        s_info[t] = 0;
      __syncthreads();
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
          if (f & 7u)
            atomicOr(&s_info[0], 1);
        }
      else if (t >= 320 && t < 320 + TX * TY)
        {
          const int nl = t - 320, li = nl % TX, lj = nl / TX;
          const int gi = i0 + li, gj = j0 + lj;
          long long base = -1;
          int deg = 0;
          bool regular = false;
          if (gi <= cv.o1[0] && gj <= cv.o1[1])
            {
              const int r = cv.local_of_box[gi + (long long)cv.NX * (gj + (long long)cv.NY * k)];
              const long long off = v.nadj_ptr[r];
              deg = (int)(v.nadj_ptr[r + 1] - off);
              base = (long long)NCOL * NCOL * off;
              regular = deg == 27;
              for (int s = 0; s < 27; ++s)
                {
                  const unsigned char o = cv.inv27[(long long)r * 27 + s];
                  s_inv[nl * 27 + s] = o;
                  regular = regular && o == s;
                }
            }
          s_rowbase[nl] = base;
          s_deg[nl] = deg;
          if (!regular)
            atomicAdd(&s_info[1], 1);
        }
      __syncthreads();
      stamp(0);

      // ---- phase 1a: w*g at the quadrature points, one thread per (cell, z-level) -> LDS [q][cell]
      if ((ABL & 1) == 0 && t < 3 * CS)
        {
          const int cs = t % CS, qz = t / CS;
          const int l = cs / (CX * CY), cy = (cs % (CX * CY)) / CX, cx = cs % CX;
          const int h000 = cx + HX * (cy + HY * l);
          const bool valid = s_node[h000] >= 0 && s_node[h000 + 1 + HX + HX * HY] >= 0;
          double wg[9];
          if (valid)
            {
              double po[8], poo[8];
#pragma unroll
              for (int b = 0; b < 8; ++b)
                {
                  const int hb = h000 + (b & 1) + HX * ((b >> 1) & 1) + HX * HY * ((b >> 2) & 1);
                  po[b] = s_po[hb];
                  poo[b] = s_poo[hb];
                }
              cell_wg_plane(po, poo, S, qz, wg);
            }
          else
            {
#pragma unroll
              for (int q = 0; q < 9; ++q)
                wg[q] = 0.0; // cells outside the local box contribute nothing
            }
#pragma unroll
          for (int q = 0; q < 9; ++q)
            s_stage[(qz * 9 + q) * CS + cs] = wg[q];
        }
      __syncthreads();
      stamp(1);

      // ---- phase 1b: moment tables, 3 threads per cell -> LDS [number][cell]
      if ((ABL & 1) == 0 && t < 3 * CS)
        {
          const int cs = t % CS, sub = t / CS;
          double *out = s_buf + cs;
          double wg[27];
#pragma unroll
          for (int q = 0; q < 27; ++q)
            wg[q] = s_stage[q * CS + cs];
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
                do_pair(1, 2);
            }
        }
      __syncthreads();
      stamp(2);

      // ---- phases 2 + 3 per row component
      const int wave = t >> 6, lane = t & 63;
      const int ti = lane % TX, tj = lane / TX;
      const int hc = (ti + 1) + HX * ((tj + 1) + HY * 1); // halo index of this lane's node
      const bool owned = (i0 + ti) <= cv.o1[0] && (j0 + tj) <= cv.o1[1];
      const bool masked = s_info[0] != 0;
      const bool regular_tile = (NCOL == 3) && s_info[1] == 0;
      unsigned row_flag = 0;
      unsigned char nbf[27];
      if (masked)
        {
          row_flag = s_flag[hc];
#pragma unroll
          for (int o = 0; o < 27; ++o)
            nbf[o] = s_flag[hc + (o % 3 - 1) + HX * ((o / 3) % 3 - 1) + HX * HY * (o / 9 - 1)];
        }
      const double *lane_base = s_buf + (CX * CY) + (tj + 1) * CX + (ti + 1);
      double *stage_row = s_stage + lane * STG;
      // fast copy-out of regular tiles: flat element f = nl*81 + e  ->  rowbase[nl] + c*81 + e
      constexpr int NIT = (TX * TY * STG + NTHREADS - 1) / NTHREADS;
      long long dst0[NIT];
      if (regular_tile)
        {
#pragma unroll
          for (int it = 0; it < NIT; ++it)
            {
              const int f = t + it * NTHREADS;
              const int nl = f / STG;
              dst0[it] = (f < TX * TY * STG) ? s_rowbase[nl] + (f - nl * STG) : -1;
            }
        }

#pragma unroll 1
      for (int c = 0; c < 3; ++c)
        {
          if ((ABL & 2) == 0 && owned)
            {
              if (masked)
                {
                  if (c == 0)
                    uu_rows<0, true>(wave, lane_base, S, stage_row, row_flag, nbf);
                  else if (c == 1)
                    uu_rows<1, true>(wave, lane_base, S, stage_row, row_flag, nbf);
                  else
                    uu_rows<2, true>(wave, lane_base, S, stage_row, row_flag, nbf);
                }
              else
                {
                  if (c == 0)
                    uu_rows<0, false>(wave, lane_base, S, stage_row, row_flag, nbf);
                  else if (c == 1)
                    uu_rows<1, false>(wave, lane_base, S, stage_row, row_flag, nbf);
                  else
                    uu_rows<2, false>(wave, lane_base, S, stage_row, row_flag, nbf);
                }
            }
          stamp(3);
          __syncthreads();
          stamp(4);
          if ((ABL & 4) == 0)
            {
              if (regular_tile)
                {
#pragma unroll
                  for (int it = 0; it < NIT; ++it)
                    if (dst0[it] >= 0)
                      vals[dst0[it] + c * STG] = s_stage[t + it * NTHREADS];
                }
              else
                {
                  // general copy-out: element f of the tile's row-c data, flat over (node, slot, column comp)
                  constexpr int rowlen = 27 * NCOL;
                  for (int f = t; f < TX * TY * rowlen; f += NTHREADS)
                    {
                      const int nl = f / rowlen, e = f - nl * rowlen;
                      const int s = e / NCOL, d = e - s * NCOL;
                      const long long base = s_rowbase[nl];
                      const int deg = s_deg[nl];
                      if (base < 0 || s >= deg)
                        continue;
                      const int o = s_inv[nl * 27 + s];
                      const double val = (d < 3) ? s_stage[nl * STG + o * 3 + d] : 0.0;
                      vals[base + (long long)c * NCOL * deg + s * NCOL + d] = val;
                    }
                }
            }
          stamp(5);
          __syncthreads();
          stamp(6);
        }
    }

  } // namespace

  int launch_cart_uu_only(const DevView &v, const CartView &cv, const pfm_params &p, double *vals_uu, hipStream_t s)
  {
    if (!getenv("PFM_CART_V1"))
      return launch_cart_uu3(v, cv, p, vals_uu, s); // third generation (pfm_cart_uu3.hip)
    int rc = ensure_g1();
    if (rc)
      return rc;
    const MatScal S = make_mat_scal(p, cv);
    const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1, OWZ = cv.o1[2] - cv.o0[2] + 1;
    const int ntx = (OWX + TX - 1) / TX, nty = (OWY + TY - 1) / TY;
    const unsigned nb = (unsigned)(ntx * nty * OWZ);
    const char *abl_env = getenv("PFM_UU_ABL"); // profiling only: skip phases (results are then wrong)
    const int abl = abl_env ? atoi(abl_env) : 0;
    unsigned long long *dbg = nullptr;
    if (abl == 8)
      {
        static unsigned long long *d_dbg = nullptr;
        if (!d_dbg && hipMalloc((void **)&d_dbg, 16 * sizeof(unsigned long long)) != hipSuccess)
          return PFM_ERR_HIP;
        hipMemsetAsync(d_dbg, 0, 16 * sizeof(unsigned long long), s);
        hipLaunchKernelGGL((k_cart_uu<3, 8>), dim3(nb), dim3(NTHREADS), 0, s, v, cv, S, vals_uu, d_dbg);
        unsigned long long h[16];
        hipMemcpy(h, d_dbg, sizeof(h), hipMemcpyDeviceToHost);
        const char *names[7] = {"phase0 (halo loads)", "1a (w*g)", "1b (moments)", "node phase", "barrier after node",
                                "copy-out", "barrier after copy"};
        fprintf(stderr, "[k_cart_uu phase clock, wave 0, cycles per tile]");
        for (int i = 0; i < 7; ++i)
          fprintf(stderr, " %s=%.0f", names[i], (double)h[i] / nb);
        fprintf(stderr, "\n");
        return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
      }
    if (v.layout == PFM_LAYOUT_INTERLEAVED)
      hipLaunchKernelGGL(k_cart_uu<4>, dim3(nb), dim3(NTHREADS), 0, s, v, cv, S, vals_uu, dbg);
    else
      switch (abl)
        {
          case 1:
            hipLaunchKernelGGL((k_cart_uu<3, 1>), dim3(nb), dim3(NTHREADS), 0, s, v, cv, S, vals_uu, dbg);
            break;
          case 2:
            hipLaunchKernelGGL((k_cart_uu<3, 2>), dim3(nb), dim3(NTHREADS), 0, s, v, cv, S, vals_uu, dbg);
            break;
          case 3:
            hipLaunchKernelGGL((k_cart_uu<3, 3>), dim3(nb), dim3(NTHREADS), 0, s, v, cv, S, vals_uu, dbg);
            break;
          case 4:
            hipLaunchKernelGGL((k_cart_uu<3, 4>), dim3(nb), dim3(NTHREADS), 0, s, v, cv, S, vals_uu, dbg);
            break;
          case 6:
            hipLaunchKernelGGL((k_cart_uu<3, 6>), dim3(nb), dim3(NTHREADS), 0, s, v, cv, S, vals_uu, dbg);
            break;
          case 7:
            hipLaunchKernelGGL((k_cart_uu<3, 7>), dim3(nb), dim3(NTHREADS), 0, s, v, cv, S, vals_uu, dbg);
            break;
          default:
            hipLaunchKernelGGL((k_cart_uu<3, 0>), dim3(nb), dim3(NTHREADS), 0, s, v, cv, S, vals_uu, dbg);
        }
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }

} // namespace pfm
