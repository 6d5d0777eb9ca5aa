// pfm_cart_uu3.hip — (u,u) block, row-owner kernel, third generation ("mirrored half-waves").
//
// Mathematics: 63 moment tables per cell, see the header of pfm_cart.hip.
//
// Why a third generation.  The first one (k_cart_uu, removed; tile 8x8 nodes, both cell layers + a row
// staging buffer in LDS = 130 KB) ran ONE workgroup per CU; its phases (halo load, cell phase, node phase, copy-out)
// are serialised by barriers and every phase's latency is exposed (profiles/r01: VALU active 38 %
// of the wave cycles, copy-out alone at the HBM write floor).  Forcing the same code to 16 waves
// per CU gave 1.6x; keeping one cell layer resident and the partial sums in registers did not pay
// (spills + twice the barriers).  This version halves the TILE instead (8x4 nodes: 45 KB tables +
// 21 KB staging => two workgroups per CU) and keeps all 64 lanes busy with 32 nodes by splitting
// the two cell LAYERS over the two halves of each wave:
//
//   lanes  0..31  <->  node n, cells below the node plane (a_z = 1)
//   lanes 32..63  <->  node n, cells above the node plane (a_z = 0)
//
// Both halves execute the SAME instruction stream: the tables of the upper layer are stored
// z-mirrored (index permutation gamma_z -> 2-gamma_z, alpha_z -> 1-alpha_z, sign flip where exactly
// one z-derivative is involved), so "upper cell, slot (ox,oy,+oz)" looks like "lower cell, slot
// (ox,oy,-oz)".  A slot with oz = -1 is completed by the lower half while the upper half completes
// its mirror slot oz = +1; slots with oz = 0 need one cross-half add.
#include "pfm_internal.h"
#include "pfm_cart_common.h"

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include <type_traits>

namespace pfm
{
  namespace
  {
    constexpr int NNUM3 = 63;
    // Table storage: tables in groups of eight, a cell ROW of a group = 8 tables x 9 cells = 72 doubles, i.e. the
    // four tile rows of a half-wave (row stride 72 = 8 mod 32 doubles, 8 lanes each) tile the 32 double-banks of a
    // ds_read_b64 lane group exactly: every table read of the node phase is conflict-free (round 5; with the
    // [table][cell] layout of rounds 1-4, row stride 9, rows 0 and 3 of the tile shared three banks)
    constexpr int TROW3 = 72, TLAY3 = 5 * TROW3, TGRP3 = 2 * TLAY3, TABSZ3 = 8 * TGRP3; // 5760 doubles
    __host__ __device__ constexpr int tab_off3(int t) { return (t >> 3) * TGRP3 + (t & 7) * C3X; }

    __host__ __device__ constexpr int idxA3(int c, int gi, int gj) { return c * 9 + gi * 3 + gj; }
    __host__ __device__ constexpr int pair3(int lo, int hi) { return lo == 0 ? (hi == 1 ? 0 : 1) : 2; }
    __host__ __device__ constexpr int idxT3(int p, int al, int be, int g) { return 27 + p * 12 + al * 6 + be * 3 + g; }
    __host__ __device__ constexpr int sg3(int bit) { return bit ? 1 : -1; }

    // ---- node phase: slot sets and their cell visits -------------------------------------------------------------
    // The 18 slots with oz <= 0 are split over the 8 waves in z-symmetric sets (the mirror slots oz = +1 are the same
    // instruction stream run by the upper half-wave); every set needs exactly 4 (slot, cell) visits per half:
    //   W0: (0,0,0)  W1: (0,0,-1)  W2: (0,+-1,0)  W3: (+-1,0,0)  W4: (0,+-1,-1)  W5: (+-1,0,-1)  W6: (+-1,+-1,0)  W7: (+-1,+-1,-1)
    struct Vis
    {
      int ox, oy, oz, ex, ey, slot, first, last; // slot = index within the set; first/last visit of that slot
    };
    __host__ __device__ constexpr Vis visit_of(int W, int v)
    {
      const int oz = (W == 0 || W == 2 || W == 3 || W == 6) ? 0 : -1;
      const int nslot = (W < 2) ? 1 : (W < 6 ? 2 : 4);
      int n = 0;
      for (int sl = 0; sl < nslot; ++sl)
        {
          int ox = 0, oy = 0;
          if (W == 2 || W == 4)
            oy = sl ? 1 : -1;
          else if (W == 3 || W == 5)
            ox = sl ? 1 : -1;
          else if (W >= 6)
            {
              ox = (sl & 1) ? 1 : -1;
              oy = (sl & 2) ? 1 : -1;
            }
          int cnt = 0;
          const int total = (ox == 0 ? 2 : 1) * (oy == 0 ? 2 : 1);
          for (int ey = -1; ey <= 0; ++ey)
            for (int ex = -1; ex <= 0; ++ex)
              {
                const int bx = -ex + ox, by = -ey + oy;
                if (bx < 0 || bx > 1 || by < 0 || by > 1)
                  continue;
                if (n == v)
                  return Vis{ox, oy, oz, ex, ey, sl, cnt == 0, cnt == total - 1};
                ++n;
                ++cnt;
              }
        }
      return Vis{0, 0, 0, 0, 0, -1, 0, 0};
    }
    __host__ __device__ constexpr int nslots_of(int W) { return (W < 2) ? 1 : (W < 6 ? 2 : 4); }

    // the 9 table values one visit needs for ALL nine (row comp, col comp) entries: A^k (k = 0..2), then per pair
    // p = (lo,hi): X_p = T^p[b_lo][a_hi][g_e], Y_p = T^p[a_lo][b_hi][g_e]  (21 FMAs from 9 LDS reads)
    template <int W, int V>
    __device__ __forceinline__ void uu_load_visit(const double *__restrict__ lane_base, double (&tv)[9])
    {
      constexpr Vis vi = visit_of(W, V);
      constexpr int a[3] = {-vi.ex, -vi.ey, 1}, b[3] = {-vi.ex + vi.ox, -vi.ey + vi.oy, 1 + vi.oz};
      constexpr int g[3] = {a[0] + b[0], a[1] + b[1], a[2] + b[2]};
      const double *cell = lane_base + (vi.ey * TROW3 + vi.ex); // lds_read64: single ds_read_b64s, see pfm_cart_common.h
#pragma unroll
      for (int k = 0; k < 3; ++k)
        {
          const int i = (k == 0) ? 1 : 0, j = (k == 2) ? 1 : 2;
          tv[k] = lds_read64(cell + tab_off3(idxA3(k, g[i], g[j])));
        }
#pragma unroll
      for (int p = 0; p < 3; ++p)
        {
          const int lo = (p == 2) ? 1 : 0, hi = (p == 0) ? 1 : 2, e = 3 - lo - hi;
          tv[3 + 2 * p] = lds_read64(cell + tab_off3(idxT3(p, b[lo], a[hi], g[e])));
          tv[4 + 2 * p] = lds_read64(cell + tab_off3(idxT3(p, a[lo], b[hi], g[e])));
        }
    }

    struct UuCoef // uniform constants of the node phase, read once per workgroup
    {
      double cA[3][3], cTl[3], cTm[3];
      double gA[3], cT[3]; // heterogeneous material: the geometric factors alone, 1 / h_k^2 and 1 / (h_lo h_hi)
    };

    // r += entry (C, D) of one visit, same operation order as the reference formulation K = lambda G^{CD} + mu G^{DC} +
    // mu delta_CD tr G (every table value enters through one FMA with a host-precombined constant)
    // HET: lam, mu = Lame coefficients of the visited cell (cracks.cc:2207-2216); the constants of MatScal are formed
    // per visit from the geometric factors
    template <int W, int V, int C, int D, bool HET>
    __device__ __forceinline__ void uu_acc_visit(const double (&tv)[9], const UuCoef &K, double lam, double mu, double &r)
    {
      constexpr Vis vi = visit_of(W, V);
      constexpr int a[3] = {-vi.ex, -vi.ey, 1}, b[3] = {-vi.ex + vi.ox, -vi.ey + vi.oy, 1 + vi.oz};
      if constexpr (HET)
        {
          if constexpr (C == D)
            {
#pragma unroll
              for (int k = 0; k < 3; ++k)
                {
                  const double ca = (k == C ? lam + 2 * mu : mu) * K.gA[k];
                  r = fma((sg3(a[k]) * sg3(b[k]) > 0) ? ca : -ca, tv[k], r);
                }
            }
          else
            {
              constexpr int lo = C < D ? C : D, hi = C < D ? D : C, p = pair3(lo, hi);
              const double t1 = (C < D) ? tv[3 + 2 * p] : tv[4 + 2 * p];
              const double t2 = (C < D) ? tv[4 + 2 * p] : tv[3 + 2 * p];
              const double cl = K.cT[p] * lam, cm = K.cT[p] * mu;
              r = fma((sg3(a[C]) * sg3(b[D]) > 0) ? cl : -cl, t1, r);
              r = fma((sg3(a[D]) * sg3(b[C]) > 0) ? cm : -cm, t2, r);
            }
        }
      else if constexpr (C == D)
        {
#pragma unroll
          for (int k = 0; k < 3; ++k)
            r = fma((sg3(a[k]) * sg3(b[k]) > 0) ? K.cA[C][k] : -K.cA[C][k], tv[k], r);
        }
      else
        {
          constexpr int lo = C < D ? C : D, hi = C < D ? D : C, p = pair3(lo, hi);
          const double t1 = (C < D) ? tv[3 + 2 * p] : tv[4 + 2 * p];
          const double t2 = (C < D) ? tv[4 + 2 * p] : tv[3 + 2 * p];
          r = fma((sg3(a[C]) * sg3(b[D]) > 0) ? K.cTl[p] : -K.cTl[p], t1, r);
          r = fma((sg3(a[D]) * sg3(b[C]) > 0) ? K.cTm[p] : -K.cTm[p], t2, r);
        }
    }

    // x[l] + x[l ^ 32] in every lane, through the VALU (v_permlane32_swap) instead of two LDS bpermutes per double
    __device__ __forceinline__ double add_across_halves(double x)
    {
      const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
      const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      return __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
    }

    // row component C of slot set W for both half-waves, from the cached table values: 28 FMAs, the cross-half adds of
    // the oz = 0 slots, the constraint masks, 3 staged values per slot.  stage_half = the lane's staged row shifted by 18
    // slots for the upper half (a slot with oz = -1 completed by the lower half is slot o_lo, its mirror completed by the
    // upper half is o_lo + 18); oz = 0 slots are summed over the halves and stored by both (same value, same address).
    // RES: also returns (in every lane of the node) this slot set's part of  sum_j K[(node,C),(j,d)] u_(j,d)  over the
    // UNMASKED entries -- the displacement residual is  R_u = (alpha_B-1) p sum_q pfx^2 dN/dx_C JxW - K_uu u  for the
    // unsplit law (sigma+ is linear in u; cracks.cc:2393-2410 against 2340-2368)
    template <int W, int C, bool MASKED, bool HET, bool RES>
    __device__ __forceinline__ double uu_row_component(const double (&tv)[4][9], const UuCoef &K, const double (&lamv)[4],
                                                       const double (&muv)[4], double *__restrict__ stage_row,
                                                       double *__restrict__ stage_half, unsigned row_flag,
                                                       const unsigned char *__restrict__ flag_own,
                                                       const unsigned char *__restrict__ flag_half,
                                                       const double *__restrict__ u_own, const double *__restrict__ u_half)
    {
      double dot_plane = 0.0, dot_half = 0.0;
      constexpr int NS = nslots_of(W);
      double val[NS][3];
#pragma unroll
      for (int sl = 0; sl < NS; ++sl)
        val[sl][0] = val[sl][1] = val[sl][2] = 0.0;
      static_for<4>([&](auto Vv) __attribute__((always_inline)) {
        constexpr int V = decltype(Vv)::value;
        constexpr Vis vi = visit_of(W, V);
        uu_acc_visit<W, V, C, 0, HET>(tv[V], K, lamv[V], muv[V], val[vi.slot][0]);
        uu_acc_visit<W, V, C, 1, HET>(tv[V], K, lamv[V], muv[V], val[vi.slot][1]);
        uu_acc_visit<W, V, C, 2, HET>(tv[V], K, lamv[V], muv[V], val[vi.slot][2]);
      });
      static_for<4>([&](auto Vv) __attribute__((always_inline)) {
        constexpr int V = decltype(Vv)::value;
        constexpr Vis vi = visit_of(W, V);
        if constexpr (vi.last)
          {
            double v0 = val[vi.slot][0], v1 = val[vi.slot][1], v2 = val[vi.slot][2];
            if constexpr (vi.oz == 0)
              {
                v0 = add_across_halves(v0);
                v1 = add_across_halves(v1);
                v2 = add_across_halves(v2);
              }
            if constexpr (RES)
              {
                const double *un = (vi.oz == 0 ? u_own : u_half) + (vi.ox + H3X * vi.oy);
                const double part = fma(v2, un[2 * NH3], fma(v1, un[NH3], v0 * un[0]));
                if constexpr (vi.oz == 0)
                  dot_plane += part; // the same value in both halves
                else
                  dot_half += part;
              }
            if constexpr (MASKED)
              {
                const unsigned cf = (vi.oz == 0 ? flag_own : flag_half)[vi.ox + H3X * vi.oy];
                const bool rcon = (row_flag >> C) & 1u;
                constexpr bool centre = (vi.ox == 0 && vi.oy == 0 && vi.oz == 0);
                if (rcon || (cf & 1u))
                  v0 = (rcon && centre && C == 0) ? v0 : 0.0;
                if (rcon || (cf & 2u))
                  v1 = (rcon && centre && C == 1) ? v1 : 0.0;
                if (rcon || (cf & 4u))
                  v2 = (rcon && centre && C == 2) ? v2 : 0.0;
              }
            constexpr int o_lo = (vi.ox + 1) + 3 * (vi.oy + 1) + 9 * (vi.oz + 1);
            double *dst = (vi.oz == 0 ? stage_row : stage_half) + o_lo * 3;
            dst[0] = v0;
            dst[1] = v1;
            dst[2] = v2;
          }
      });
      if constexpr (RES)
        {
          constexpr bool has_half = !(W == 0 || W == 2 || W == 3 || W == 6);
          return has_half ? add_across_halves(dot_half) : dot_plane;
        }
      else
        return 0.0;
    }

    // =====================================================================================
    template <int NCOL /* 3 blocked, 4 interleaved */, bool CLK = false /* profiling only */,
              bool HET = false /* per-cell Lame coefficients (CartView::cell_lam) */,
              bool RES = false /* also writes the displacement rows of the residual (res_pde) */>
    __global__ __launch_bounds__(NT3, 4) void k_cart_uu3(DevView v, CartView cv, const MatScal *__restrict__ Sp, double *__restrict__ vals,
                                                         unsigned long long *__restrict__ dbg, double *__restrict__ res_pde, int prio)
    {
      const MatScal &S = *Sp; // per-launch scalars in device memory (see pfm_internal.h)
      // wave priorities per phase (prio != 0): the halo loads, w*g and the copy-out are short instruction sequences with
      // long latencies -- issued ahead of the co-resident workgroup's arithmetic
      if (prio & 1)
        __builtin_amdgcn_s_setprio(3);
      long long tclk = 0;
      auto stamp = [&](int phase) __attribute__((always_inline)) {
        if constexpr (CLK)
          {
            const long long now = clock64();
            if (threadIdx.x == 0 && phase >= 0)
              dbg[(size_t)blockIdx.x * 8 + phase] += (unsigned long long)(now - tclk); // one slot per tile: no contention
            tclk = now;
          }
      };
      stamp(-1);
      __shared__ double s_tab[TABSZ3];       // moment tables (layout: tab_off3); layer 1 stored z-mirrored
      __shared__ double s_stage[NN3 * STG];  // staged rows [node][81]; w*g(q) [27][90] during the cell phase
      __shared__ double s_po[NH3], s_poo[NH3];
      __shared__ int s_node[NH3];
      __shared__ unsigned char s_flag[NH3];
      __shared__ long long s_rowbase[NN3];
      __shared__ unsigned s_mask[NN3]; // neighbour mask of the row (bit o: lattice offset o exists)
      __shared__ double s_lam[HET ? CS3 : 1], s_mu[HET ? CS3 : 1]; // Lame coefficients of the tile's cells
      __shared__ double s_u[RES ? 3 * NH3 : 1];                    // displacements of the halo nodes [component][node]
      __shared__ double s_part[RES ? 2 * 8 * NN3 : 1];             // K u per [component & 1][wave = slot set][node]
      __shared__ double s_pres[RES ? 3 * NN3 : 1];                 // pressure part of the residual [component][node]
      __shared__ int s_resrow[RES ? NN3 : 1]; // local id of the tile's nodes (residual rows): looked up ONCE, before any store of the
                                               // workgroup -- a table look-up behind the copy-out stores is a wait for them (round 4)
      __shared__ int s_any[4]; // waves 0..2: some node of the halo carries a displacement flag; [3]: some row is not full
      static_assert(27 * CS3 <= NN3 * STG, "w*g scratch must fit in the staging buffer");

      const int t = threadIdx.x;
      const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1;
      const int ntx = (OWX + T3X - 1) / T3X, nty = (OWY + T3Y - 1) / T3Y;
      int bid = xcd_tile_index();
      const bool listed = cv.tile_sel == 2 && cv.bnd_uu3 != nullptr; // compact launch over the boundary tiles
      if (listed)
        {
          if (bid >= cv.n_bnd_uu3)
            return;
          bid = cv.bnd_uu3[bid];
        }
      if (bid >= ntx * nty * (cv.o1[2] - cv.o0[2] + 1))
        return; // padding of the XCD-aware grid
      const int tix = bid % ntx, tiy = (bid / ntx) % nty, tk = bid / (ntx * nty);
      const int i0 = cv.o0[0] + tix * T3X, j0 = cv.o0[1] + tiy * T3Y, k = cv.o0[2] + tk;
      if (!listed && cart_tile_skipped(cv, cart_range_has_ghost(cv, 0, i0 - 1, i0 + T3X) || cart_range_has_ghost(cv, 1, j0 - 1, j0 + T3Y) ||
                                               cart_range_has_ghost(cv, 2, k - 1, k + 1)))
        return; // overlapped assembly: the other launch owns this tile

      // ---- phase 0: nodal halo + CSR row info (flags of the tile are collected per wave: no atomics, no init barrier)
      stamp(0);
      if (t < NH3)
        {
          const int li = t % H3X, lj = (t / H3X) % H3Y, lk = t / (H3X * H3Y);
          const int gi = i0 - 1 + li, gj = j0 - 1 + lj, gk = k - 1 + lk;
          int n = -1;
          double a = 0.0, b = 0.0, uu[3] = {0.0, 0.0, 0.0};
          unsigned char f = 0;
          if (gi >= 0 && gi < cv.NX && gj >= 0 && gj < cv.NY && gk >= 0 && gk < cv.NZ)
            {
              n = cart_local_id(cv, gi, gj, gk);
              a = v.phi_old[n];
              b = v.phi_oldold[n];
              f = v.node_flags[n];
              if constexpr (RES)
                {
                  uu[0] = v.u[0][n];
                  uu[1] = v.u[1][n];
                  uu[2] = v.u[2][n];
                }
              if (!S.monolithic) // one combined field is interpolated in the cell phase (cell_wg_plane_lin)
                a = S.use_old ? a : b + S.tfac * (a - b);
            }
          if constexpr (RES)
            {
              s_u[t] = uu[0];
              s_u[(RES ? NH3 : 0) + t] = uu[1];
              s_u[(RES ? 2 * NH3 : 0) + t] = uu[2];
            }
          s_node[t] = n;
          s_po[t] = a;
          s_poo[t] = b;
          s_flag[t] = f;
          const unsigned long long any = __ballot((f & 7u) != 0);
          if ((t & 63) == 0)
            s_any[t >> 6] = any != 0; // waves 0..2
        }
      else if (t >= 256 && t < 256 + NN3)
        {
          const int nl = t - 256, li = nl % T3X, lj = nl / T3X;
          const int gi = i0 + li, gj = j0 + lj;
          long long base = -1;
          unsigned mask = 0u;
          if (gi <= cv.o1[0] && gj <= cv.o1[1])
            {
              const int r = cart_local_id(cv, gi, gj, k);
              base = (long long)NCOL * NCOL * v.nadj_ptr[r];
              mask = cv.nbr_mask[r];
              if constexpr (RES)
                s_resrow[nl] = r;
            }
          s_rowbase[nl] = base;
          s_mask[nl] = mask;
          const unsigned long long irr = __ballot(mask != 0x7ffffffu); // fewer than 27 neighbours, or not an owned node
          if (nl == 0)
            s_any[3] = irr != 0;
        }
      else if (HET && t >= 320 && t < 320 + CS3)
        {
          const int cs = t - 320, l = cs / CL3, cy = (cs % CL3) / C3X, cx = cs % C3X;
          const int ci = i0 - 1 + cx, cj = j0 - 1 + cy, ck = k - 1 + l;
          double la = 0.0, mu = 0.0;
          if (ci >= 0 && ci < cv.NX - 1 && cj >= 0 && cj < cv.NY - 1 && ck >= 0 && ck < cv.NZ - 1)
            {
              const long long cidx = ci + (long long)(cv.NX - 1) * (cj + (long long)(cv.NY - 1) * ck);
              la = cv.cell_lam[cidx];
              mu = cv.cell_mu[cidx];
            }
          s_lam[HET ? cs : 0] = la;
          s_mu[HET ? cs : 0] = mu;
        }
      stamp(5); // thread 0: its own halo loads have returned and are stored
      __syncthreads();
      stamp(0);
      if (prio & 2)
        __builtin_amdgcn_s_setprio(2);
      else if (prio)
        __builtin_amdgcn_s_setprio(0);

      // ---- cell phase a: w*g at the quadrature points, thread <-> (cell, z-level) -> LDS [q][cell]
      if (t < 3 * CS3)
        {
          const int cs = t % CS3, qz = t / CS3;
          const int l = cs / CL3, cy = (cs % CL3) / C3X, cx = cs % C3X;
          const int h000 = cx + H3X * (cy + H3Y * l);
          const bool valid = s_node[h000] >= 0 && s_node[h000 + 1 + H3X + H3X * H3Y] >= 0;
          double wg[9];
          if (valid)
            {
              double po[8], poo[8];
              if (!S.monolithic)
                {
#pragma unroll
                  for (int b = 0; b < 8; ++b)
                    po[b] = s_po[h000 + (b & 1) + H3X * ((b >> 1) & 1) + H3X * H3Y * ((b >> 2) & 1)];
                  cell_wg_plane_lin(po, S, qz, wg);
                }
              else
                {
#pragma unroll
                  for (int b = 0; b < 8; ++b)
                    {
                      const int hb = h000 + (b & 1) + H3X * ((b >> 1) & 1) + H3X * H3Y * ((b >> 2) & 1);
                      po[b] = s_po[hb];
                      poo[b] = s_poo[hb];
                    }
                  cell_wg_plane(po, poo, S, qz, wg);
                }
            }
          else
            {
#pragma unroll
              for (int q = 0; q < 9; ++q)
                wg[q] = 0.0;
            }
#pragma unroll
          for (int q = 0; q < 9; ++q)
            s_stage[(qz * 9 + q) * CS3 + cs] = wg[q];
        }
      __syncthreads();
      stamp(1);
      if (prio)
        __builtin_amdgcn_s_setprio(0);

      // ---- cell phase b: moment tables.  Round 3: wave <-> (family f, cell group) with the family WAVE-UNIFORM: waves
      // 0..5 = families 0..2 x cell groups {cells 0..44 = layer 0, cells 45..89 = layer 1}, lane <-> cell.  A family is
      // A^f plus both halves of the pair table that contracts direction f first (T^xy, T^xz for f = 0 ... see below), so
      // that the 27 w*g values of a cell are read ONCE for 198 flops (round 2: thread <-> (cell, task), 9 tasks per cell
      // with run-time strides: 27 reads per 65 flops, and 63 % of the VALU instructions of the phase were address
      // arithmetic -- on this chip an integer VALU instruction costs the same 4-cycle issue slot as an FP64 FMA).
      // Every stride, table number and the z-mirroring of layer 1 are compile-time constants per (family, layer).
      // The arithmetic of each table entry is unchanged (bitwise identical tables).
      // Families: f = 0: A^x, T^xy (lo = x);  f = 1: A^y, T^yz (lo = y);  f = 2: A^z, T^xz (lo = x, hi = z).
      {
        const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
        const int ln = t & 63;
        auto family = [&](auto Ff, auto Ll) __attribute__((always_inline)) {
          constexpr int f = decltype(Ff)::value, l = decltype(Ll)::value;
          constexpr bool mir = l == 1;
          if (ln < CL3)
            {
              const int cs = l * CL3 + ln;
              const double *wq = s_stage + cs;
              const int cyl = ln / C3X;
              double *out = s_tab + l * TLAY3 + cyl * TROW3 + (ln - cyl * C3X);
              double w27[27];
#pragma unroll
              for (int q = 0; q < 27; ++q)
                w27[q] = lds_read64(wq + q * CS3);
              __builtin_amdgcn_sched_barrier(0);
              // ---- A^c, c = f: sum over q_c, then the two moment axes (i, j) = other axes ascending
              {
                constexpr int c = f;
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
                constexpr bool zj = (c != 2); // for c = x or y the second moment axis j is z
#pragma unroll
                for (int gi = 0; gi < 3; ++gi)
                  {
                    double tq[3];
#pragma unroll
                    for (int qj = 0; qj < 3; ++qj)
                      tq[qj] = s9[qj][0] * c_g1.m[gi][0] + s9[qj][1] * c_g1.m[gi][1] + s9[qj][2] * c_g1.m[gi][2];
#pragma unroll
                    for (int gj = 0; gj < 3; ++gj)
                      {
                        const double val = tq[0] * c_g1.m[gj][0] + tq[1] * c_g1.m[gj][1] + tq[2] * c_g1.m[gj][2];
                        const int gjm = (mir && zj) ? 2 - gj : gj; // layer 1 is stored z-mirrored
                        out[tab_off3(c * 9 + gi * 3 + gjm)] = val;
                      }
                  }
              }
              // ---- T^p, both halves al = 0, 1: p = 0 (x,y) for f = 0, p = 2 (y,z) for f = 1, p = 1 (x,z) for f = 2
              {
                constexpr int p = (f == 0) ? 0 : (f == 1) ? 2 : 1;
                constexpr int slo = (p == 2) ? 3 : 1;
                constexpr int shi = (p == 0) ? 3 : 9;
                constexpr int se = (p == 0) ? 9 : (p == 1) ? 3 : 1;
                constexpr bool mz = mir && p == 0, mb = mir && p != 0;
#pragma unroll
                for (int al = 0; al < 2; ++al)
                  {
                    const double na0 = c_g1.n[al][0], na1 = c_g1.n[al][1], na2 = c_g1.n[al][2];
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
                          t2[qe] = t1[qe][0] * c_g1.n[be][0] + t1[qe][1] * c_g1.n[be][1] + t1[qe][2] * c_g1.n[be][2];
                        const int bes = mb ? 1 - be : be;
#pragma unroll
                        for (int g = 0; g < 3; ++g)
                          {
                            const double val = t2[0] * c_g1.m[g][0] + t2[1] * c_g1.m[g][1] + t2[2] * c_g1.m[g][2];
                            const int gm = mz ? 2 - g : g;
                            out[tab_off3(27 + p * 12 + al * 6 + bes * 3 + gm)] = mb ? -val : val;
                          }
                      }
                  }
              }
            }
        };
        using std::integral_constant;
        switch (wv)
          {
            case 0: family(integral_constant<int, 0>{}, integral_constant<int, 0>{}); break;
            case 1: family(integral_constant<int, 0>{}, integral_constant<int, 1>{}); break;
            case 2: family(integral_constant<int, 1>{}, integral_constant<int, 0>{}); break;
            case 3: family(integral_constant<int, 1>{}, integral_constant<int, 1>{}); break;
            case 4: family(integral_constant<int, 2>{}, integral_constant<int, 0>{}); break;
            case 5: family(integral_constant<int, 2>{}, integral_constant<int, 1>{}); break;
            default: break;
          }
      }
      __syncthreads();
      stamp(2);

      // ---- node phase + copy-out.  Every wave first reads the 36 table values of its slot set (4 visits x 9) in ONE
      // batch and keeps them in registers for all three row components: 36 LDS reads per lane instead of 84, one
      // latency instead of a chain of them, and the tables are dead afterwards -- their LDS becomes the second staging
      // buffer, so that the copy-out of component c overlaps the arithmetic of component c + 1 (one barrier per
      // component instead of two).
      const int wave = __builtin_amdgcn_readfirstlane(t >> 6); // wave-uniform: scalar branches between the slot sets
      const int lane = t & 63;
      const bool upper = lane >= 32;
      const int nl_lane = lane & 31;
      const int ti = nl_lane % T3X, tj = nl_lane / T3X;
      const int hc = (ti + 1) + H3X * ((tj + 1) + H3Y * 1);
      const bool masked = (s_any[0] | s_any[1] | s_any[2]) != 0;
      const bool regular_tile = (NCOL == 3) && s_any[3] == 0;
      const unsigned row_flag = s_flag[hc];
      // the cell "below-left" of the node in its layer: lower half -> layer 0, upper half -> layer 1 (mirrored tables)
      const double *lane_base = s_tab + (upper ? TLAY3 : 0) + (tj + 1) * TROW3 + (ti + 1);
      const int lane_cs = (upper ? CL3 : 0) + (tj + 1) * C3X + (ti + 1); // the same cell in [layer][cy][cx] order (s_lam, s_mu)
      const unsigned char *flag_own = s_flag + hc, *flag_half = s_flag + hc + (upper ? H3X * H3Y : -H3X * H3Y);
      static_assert(NN3 * STG <= TABSZ3, "second staging buffer must fit in the table storage");
      UuCoef K;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        {
#pragma unroll
          for (int k = 0; k < 3; ++k)
            K.cA[c][k] = S.cA[c][k];
          K.cTl[c] = S.cTl[c];
          K.cTm[c] = S.cTm[c];
          K.gA[c] = S.ih[c] * S.ih[c];
          K.cT[c] = S.cT[c];
        }

      auto copy_out = [&](int c, const double *__restrict__ stage) __attribute__((always_inline)) {
        if (prio & 4)
          __builtin_amdgcn_s_setprio(3);
        if (regular_tile)
          {
            // thread <-> (node group g, element el) with el fixed: nodes g, g + 6, ..., g + 30 -- no division per
            // position, one row-base read and one value read per node, all of them in flight before the first store (a
            // loop pays two dependent LDS round trips of ~130 cycles per iteration); threads 486..511 idle
            constexpr int NG = 6, NIT = (NN3 + NG - 1) / NG;
            int tq = t;
            asm volatile("" : "+v"(tq)); // (g, el) are recomputed per component, not kept live across the node phases
            const int g = tq / STG, el = tq - g * STG;
            if (g < NG)
              {
                long long rb[NIT];
                double val[NIT];
#pragma unroll
                for (int i = 0; i < NIT; ++i)
                  {
                    const int nl = min(g + NG * i, NN3 - 1);
                    rb[i] = s_rowbase[nl];
                    val[i] = stage[nl * STG + el];
                  }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NIT; ++i)
                  if (g + NG * i < NN3)
                    vals[rb[i] + (c * STG + el)] = val[i];
              }
          }
        else
          {
            // rows at the faces of the box / partial tiles / next to ghost columns: thread <-> (row, lattice offset o,
            // column component); the CSR slot of offset o is its rank among the offsets that exist, or the row's
            // permutation of that rank
            constexpr int rowlen = 27 * NCOL;
            for (int f = t; f < NN3 * rowlen; f += NT3)
              {
                const int nl = f / rowlen, e = f - nl * rowlen;
                const int o = e / NCOL, d = e - o * NCOL;
                const long long base = s_rowbase[nl];
                const unsigned mask = s_mask[nl];
                if (base < 0 || !((mask >> o) & 1u))
                  continue;
                int sl = __popc(mask & ((1u << o) - 1u));
                const int deg = __popc(mask & 0x7ffffffu);
                if (mask >> 31) // the row is not in lattice order (ghost columns behind the owned ones, bound pattern)
                  sl = cv.row_perm[base / (NCOL * NCOL) + sl];
                const double val = (d < 3) ? stage[nl * STG + o * 3 + d] : 0.0;
                vals[base + (long long)c * NCOL * deg + sl * NCOL + d] = val;
              }
          }
        if (prio & 4)
          __builtin_amdgcn_s_setprio(0);
      };

      // RES: pressure part of the displacement residual, (alpha_B-1) p sum_q pfx^2 dN_a/dx_c JxW, from the A^c tables of
      // the 8 cells around the node (slot set 0 visits exactly those): sum_q w g n_ai n_aj is the sum of the four moments
      // A^c[a_i + b_i][a_j + b_j] (n_0 + n_1 = 1), and vol w pfx^2 = (w g - kappa vol w) / (1 - kappa)
      if constexpr (RES)
        {
          if (wave == 0)
            {
              double pres[3] = {0.0, 0.0, 0.0};
              const double kv4 = S.kappa * S.vol * 0.25;
              static_for<4>([&](auto Vv) __attribute__((always_inline)) {
                constexpr Vis vi = visit_of(0, decltype(Vv)::value);
                constexpr int a[3] = {-vi.ex, -vi.ey, 1};
                const double *cell = lane_base + (vi.ey * TROW3 + vi.ex);
#pragma unroll
                for (int k = 0; k < 3; ++k)
                  {
                    const int i = (k == 0) ? 1 : 0, j = (k == 2) ? 1 : 2;
                    const double s4 = (lds_read64(cell + tab_off3(idxA3(k, a[i], a[j]))) + lds_read64(cell + tab_off3(idxA3(k, a[i] + 1, a[j])))) +
                                      (lds_read64(cell + tab_off3(idxA3(k, a[i], a[j] + 1))) + lds_read64(cell + tab_off3(idxA3(k, a[i] + 1, a[j] + 1))));
                    const double mom = s4 - (s4 != 0.0 ? kv4 : 0.0); // absent cell: all tables are zero
                    const bool neg = (k == 2) ? upper : (a[k] == 0);   // sign of dN_a/dx_k (upper layer: z-mirrored tables)
                    pres[k] += neg ? -mom : mom;
                  }
              });
              const double pc = S.aB1 * S.p / (1.0 - S.kappa);
#pragma unroll
              for (int k = 0; k < 3; ++k)
                {
                  const double pk = add_across_halves(pres[k]) * (pc * S.ih[k]);
                  if (!upper)
                    s_pres[RES ? k * NN3 + nl_lane : 0] = pk; // read behind the barrier of component 0 at the earliest
                }
            }
        }
      __builtin_amdgcn_sched_barrier(0); // the pressure part is finished before the table batch is requested
      // the slot set of a wave is selected by scalar branches around the set-specific code only (table reads, the
      // arithmetic of one row component); barriers and the copy-out are shared code (instruction cache: 8 sets x 3
      // components x 2 mask variants of straight-line code)
      double tv[4][9];
      using std::integral_constant;
#define PFM_PER_SET(STMT)                                                                                                    \
  switch (wave)                                                                                                              \
    {                                                                                                                        \
      case 0: { constexpr int W = 0; STMT; } break;                                                                          \
      case 1: { constexpr int W = 1; STMT; } break;                                                                          \
      case 2: { constexpr int W = 2; STMT; } break;                                                                          \
      case 3: { constexpr int W = 3; STMT; } break;                                                                          \
      case 4: { constexpr int W = 4; STMT; } break;                                                                          \
      case 5: { constexpr int W = 5; STMT; } break;                                                                          \
      case 6: { constexpr int W = 6; STMT; } break;                                                                          \
      default: { constexpr int W = 7; STMT; } break;                                                                         \
    }
      double lamv[4] = {0.0, 0.0, 0.0, 0.0}, muv[4] = {0.0, 0.0, 0.0, 0.0}; // HET: coefficients of the 4 visited cells
      PFM_PER_SET(static_for<4>([&](auto Vv) __attribute__((always_inline)) {
        constexpr int V = decltype(Vv)::value;
        uu_load_visit<W, V>(lane_base, tv[V]);
        if constexpr (HET)
          {
            constexpr Vis vi = visit_of(W, V);
            const int cs = lane_cs + (vi.ey * C3X + vi.ex);
            lamv[V] = s_lam[cs];
            muv[V] = s_mu[cs];
          }
      }))
      const double *u_own = s_u + (RES ? hc : 0), *u_half = s_u + (RES ? hc + (upper ? H3X * H3Y : -H3X * H3Y) : 0);
      // buffer 0 = the w*g scratch (free since the moment phase), buffer 1 = the table storage: written after the
      // barrier of component 0, which every wave passes with its table values in registers
      double *st0 = s_stage + nl_lane * STG, *st1 = s_tab + nl_lane * STG;
      const int hs = upper ? 18 * 3 : 0;
      double ku = 0.0;
      // residual row of component c: sum of the 8 slot sets in a fixed order, constrained rows get 0 (cracks.cc:2440-2456)
      auto residual_out = [&](int c) __attribute__((always_inline)) {
        if constexpr (RES)
          {
            if (t < NN3 && s_rowbase[t] >= 0)
              {
                double sum = -s_pres[RES ? c * NN3 + t : 0];
#pragma unroll
                for (int w = 0; w < 8; ++w)
                  sum += s_part[((c & 1) * 8 + w) * NN3 + t]; // component 2 reuses buffer 0 behind the barrier of component 1
                const int li = t % T3X, lj = t / T3X;
                const int row = s_resrow[RES ? t : 0];
                const bool con = (s_flag[(li + 1) + H3X * ((lj + 1) + H3Y * 1)] >> c) & 1u;
                const long long di = (v.layout == PFM_LAYOUT_INTERLEAVED) ? (long long)row * 4 + c : (long long)row * 3 + c;
                res_pde[di] = con ? 0.0 : -sum;
              }
          }
      };
#define PFM_COMPONENT(C, ST)                                                                                                 \
  if (masked)                                                                                                                \
    {                                                                                                                        \
      PFM_PER_SET((ku = uu_row_component<W, C, true, HET, RES>(tv, K, lamv, muv, ST, ST + hs, row_flag, flag_own, flag_half, \
                                                               u_own, u_half)))                                             \
    }                                                                                                                        \
  else                                                                                                                       \
    {                                                                                                                        \
      PFM_PER_SET((ku = uu_row_component<W, C, false, HET, RES>(tv, K, lamv, muv, ST, ST + hs, row_flag, flag_own, flag_half, \
                                                                u_own, u_half)))                                            \
    }                                                                                                                        \
  if (RES && !upper)                                                                                                         \
    s_part[RES ? ((C & 1) * 8 + wave) * NN3 + nl_lane : 0] = ku;
      PFM_COMPONENT(0, st0)
      lds_barrier();
      stamp(3);
      copy_out(0, s_stage);
      residual_out(0);
      PFM_COMPONENT(1, st1)
      lds_barrier();
      stamp(4);
      copy_out(1, s_tab);
      residual_out(1);
      PFM_COMPONENT(2, st0)
      lds_barrier();
      stamp(3);
      copy_out(2, s_stage);
      residual_out(2);
      stamp(4);
#undef PFM_COMPONENT
#undef PFM_PER_SET
    }
  } // namespace

  int launch_cart_uu3(const DevView &v, const CartView &cv, const pfm_params &p, double *vals_uu, hipStream_t s,
                      const void *d_scal, double *res_pde, int lds_total /* LDS bytes per workgroup to pad to (launch_cart_matrix), 0: none */)
  {
    int rc = ensure_g1();
    if (rc)
      return rc;
    (void)p;
    const MatScal *S = static_cast<const MatScal *>(d_scal);
    const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1, OWZ = cv.o1[2] - cv.o0[2] + 1;
    const int ntx = (OWX + T3X - 1) / T3X, nty = (OWY + T3Y - 1) / T3Y;
    const bool listed = cv.tile_sel == 2 && cv.bnd_uu3 != nullptr;
    const unsigned nb = listed ? (unsigned)cv.n_bnd_uu3 : (unsigned)(ntx * nty * OWZ);
    if (nb == 0)
      return PFM_OK;
    const dim3 grid(xcd_grid(nb)), block(NT3);
    const bool il = v.layout == PFM_LAYOUT_INTERLEAVED, het = cv.cell_lam != nullptr, res = res_pde != nullptr;
    static const int prio = getenv("PFM_UU_PRIO") ? atoi(getenv("PFM_UU_PRIO")) : 0; // bit 0: halo loads, 1: w*g, 2: copy-out
    // dynamic LDS on top of the kernel's own: the pair launch asks for the allocation of the phase-field kernel
    auto own_lds = [](const void *fn) {
      hipFuncAttributes at{};
      return hipFuncGetAttributes(&at, fn) == hipSuccess ? (int)at.sharedSizeBytes : 0;
    };
#define PFM_UU3(NC, HETV, RESV)                                                                                              \
  do                                                                                                                         \
    {                                                                                                                        \
      static const int own = own_lds(reinterpret_cast<const void *>(&k_cart_uu3<NC, false, HETV, RESV>));                    \
      const int pad = lds_total > 0 ? std::max(0, lds_total - own) : 0;                                                      \
      hipLaunchKernelGGL((k_cart_uu3<NC, false, HETV, RESV>), grid, block, pad, s, v, cv, S, vals_uu, nullptr, res_pde, prio); \
    }                                                                                                                        \
  while (0)
    if (getenv("PFM_UU_CLK") && !il && !het) // profiling only
      {
        static unsigned long long *d_dbg = nullptr;
        const size_t nd = (size_t)xcd_grid(nb) * 8;
        if (!d_dbg && hipMalloc((void **)&d_dbg, nd * sizeof(unsigned long long)) != hipSuccess)
          return PFM_ERR_HIP;
        (void)hipMemsetAsync(d_dbg, 0, nd * sizeof(unsigned long long), s);
        if (res)
          hipLaunchKernelGGL((k_cart_uu3<3, true, false, true>), grid, block, 0, s, v, cv, S, vals_uu, d_dbg, res_pde, prio);
        else
          hipLaunchKernelGGL((k_cart_uu3<3, true>), grid, block, 0, s, v, cv, S, vals_uu, d_dbg, nullptr, prio);
        std::vector<unsigned long long> hall(nd);
        (void)hipMemcpy(hall.data(), d_dbg, nd * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long h[8] = {};
        for (size_t i = 0; i < nd; ++i)
          h[i % 8] += hall[i];
        const char *names[6] = {"phase0: wait at the barrier", "w*g", "moments", "load+node c0,c2 (+copy c1)", "copy c0,c2 + node c1",
                                "phase0: own loads -> LDS"};
        fprintf(stderr, "[k_cart_uu3 phase clock, thread 0, cycles per tile]");
        for (int i = 0; i < 6; ++i)
          fprintf(stderr, " %s=%.0f", names[i], (double)h[i] / nb);
        fprintf(stderr, "\n");
      }
    else if (il)
      {
        if (het)
          PFM_UU3(4, true, false); // heterogeneous material: the residual kernel runs
        else
          {
            if (res)
              PFM_UU3(4, false, true);
            else
              PFM_UU3(4, false, false);
          }
      }
    else
      {
        if (het)
          PFM_UU3(3, true, false); // heterogeneous material: the residual kernel runs
        else
          {
            if (res)
              PFM_UU3(3, false, true);
            else
              PFM_UU3(3, false, false);
          }
      }
#undef PFM_UU3
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }
  void cart_uu3_boundary_tiles(const CartView &cv, std::vector<int32_t> &out)
  {
    out.clear();
    const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1, OWZ = cv.o1[2] - cv.o0[2] + 1;
    const int ntx = (OWX + T3X - 1) / T3X, nty = (OWY + T3Y - 1) / T3Y;
    for (int tk = 0; tk < OWZ; ++tk)
      for (int tiy = 0; tiy < nty; ++tiy)
        for (int tix = 0; tix < ntx; ++tix)
          {
            const int i0 = cv.o0[0] + tix * T3X, j0 = cv.o0[1] + tiy * T3Y, k = cv.o0[2] + tk;
            if (cart_range_has_ghost(cv, 0, i0 - 1, i0 + T3X) || cart_range_has_ghost(cv, 1, j0 - 1, j0 + T3Y) ||
                cart_range_has_ghost(cv, 2, k - 1, k + 1))
              out.push_back(tix + ntx * (tiy + nty * tk));
          }
  }

  // entry point used by the debug overlay (pfm_ctx_force_path(ctx, 2)) and by launch_cart_matrix
  int launch_cart_uu_only(const DevView &v, const CartView &cv, const pfm_params &p, double *vals_uu, hipStream_t s,
                          void *d_scal)
  {
    return launch_cart_uu3(v, cv, p, vals_uu, s, d_scal, nullptr);
  }
} // namespace pfm
