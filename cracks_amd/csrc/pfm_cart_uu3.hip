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
#include "pfm_dma.h"

#include <hip/hip_runtime.h>
#include <cstddef>
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
    constexpr int NHP3 = H3X * H3Y; // 60 nodes per halo plane
    constexpr int NR3 = 4 * NHP3;   // ring of four nodal planes in LDS (k_cart_uu3)
    // Table storage: tables in groups of eight, a cell ROW of a group = 8 tables x 9 cells = 72 doubles, i.e. the
    // four tile rows of a half-wave (row stride 72 = 8 mod 32 doubles, 8 lanes each) tile the 32 double-banks of a
    // ds_read_b64 lane group exactly: every table read of the node phase is conflict-free (round 5; with the
    // [table][cell] layout of rounds 1-4, row stride 9, rows 0 and 3 of the tile shared three banks)
    constexpr int TROW3 = 72, TLAY3 = 5 * TROW3, TGRP3 = 2 * TLAY3, TABSZ3 = 8 * TGRP3; // 5760 doubles
    // storage slot of table t: the pair tables T^xz (39..50) and T^yz (51..62) one slot up, so that slots 32..39 -- group 4,
    // doubles [4 TGRP3, 5 TGRP3) -- hold only tables every wave has in registers after its first batch (A, T^xy) and the
    // one unused slot: that range takes the partial sums of the residual rows while T^xz / T^yz are still being read
    __host__ __device__ constexpr int tab_off3(int t)
    {
      const int sl = t + (t >= 39 ? 1 : 0);
      return (sl >> 3) * TGRP3 + (sl & 7) * C3X;
    }

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
    // PARTS: bit 0 = A^k and the pair (x,y), bit 1 = pair (x,z), bit 2 = pair (y,z).  Row component c needs the pairs that
    // contain c, so the march reads {A, xy, xz} for c = 0, fetches yz behind the barrier of component 0 and xz AGAIN behind
    // the barrier of component 1 (it is not kept across component 1: 28 instead of 36 table values live, 16 registers of
    // a budget of 128; the tables of the pairs xz, yz lie outside of what is overwritten before those reads, tab_off3)
    template <int W, int V, int PARTS>
    __device__ __forceinline__ void uu_load_visit(const double *__restrict__ lane_base, double (&tv)[9])
    {
      constexpr Vis vi = visit_of(W, V);
      constexpr int a[3] = {-vi.ex, -vi.ey, 1}, b[3] = {-vi.ex + vi.ox, -vi.ey + vi.oy, 1 + vi.oz};
      constexpr int g[3] = {a[0] + b[0], a[1] + b[1], a[2] + b[2]};
      const double *cell = lane_base + (vi.ey * TROW3 + vi.ex); // lds_read64: single ds_read_b64s, see pfm_cart_common.h
      if constexpr (PARTS & 1)
        {
#pragma unroll
          for (int k = 0; k < 3; ++k)
            {
              const int i = (k == 0) ? 1 : 0, j = (k == 2) ? 1 : 2;
              tv[k] = lds_read64(cell + tab_off3(idxA3(k, g[i], g[j])));
            }
        }
#pragma unroll
      for (int p = 0; p < 3; ++p)
        {
          if (!((PARTS >> p) & 1))
            continue;
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

    // lower half-wave: x[l] + x[l + 32]; upper half-wave: y[l - 32] + y[l] -- the sums over the two cell layers of TWO values
    // for the price of one (two lane swaps + one add): each half keeps the sum it stages
    __device__ __forceinline__ double pair_sum_across_halves(double x, double y)
    {
      const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
      return __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
    }
    // staged position (in slots) of slot sl of set W with oz <= 0
    __host__ __device__ constexpr int slot_pos_of(int W, int sl)
    {
      for (int v = 0; v < 4; ++v)
        {
          const Vis vi = visit_of(W, v);
          if (vi.slot == sl)
            return (vi.ox + 1) + 3 * (vi.oy + 1) + 9 * (vi.oz + 1);
        }
      return -1;
    }

    // row component C of slot set W for both half-waves, from the cached table values: 28 FMAs, the cross-half sums of
    // the oz = 0 slots, the constraint masks, 3 staged values per slot.  stage_half = the lane's staged row shifted by 18
    // slots for the upper half (a slot with oz = -1 completed by the lower half is slot o_lo, its mirror completed by the
    // upper half is o_lo + 18).  The entries of the oz = 0 slots are the sums of both halves' parts: they are summed in
    // PAIRS (round 5: pair_sum_across_halves; the phase clock showed the set of the four oz = 0 corner slots -- twelve
    // single sums of five instructions each per component -- 1.7k cycles behind the barrier of a phase whose lightest sets
    // need 0.6k), the lower half stages the first entry of a pair and the upper half the second.
    // RES: also returns (in every lane of the node) this slot set's part of  sum_j K[(node,C),(j,d)] u_(j,d)  over the
    // UNMASKED entries -- the displacement residual is  R_u = (alpha_B-1) p sum_q pfx^2 dN/dx_C JxW - K_uu u  for the
    // unsplit law (sigma+ is linear in u; cracks.cc:2393-2410 against 2340-2368); the products are formed with each half's
    // own part and summed once
    template <int W, int C, bool MASKED, bool HET, bool RES>
    __device__ __forceinline__ double uu_row_component(const double (&tv)[4][9], const UuCoef &K, const double (&lamv)[4],
                                                       const double (&muv)[4], double *__restrict__ stage_row,
                                                       double *__restrict__ stage_half, bool upper, unsigned row_flag,
                                                       const unsigned char *__restrict__ flag_own,
                                                       const unsigned char *__restrict__ flag_half,
                                                       const double *__restrict__ u_own, const double *__restrict__ u_half)
    {
      double dot = 0.0;
      constexpr int NS = nslots_of(W);
      constexpr bool PLANE = (W == 0 || W == 2 || W == 3 || W == 6); // the slots of the set have oz = 0
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
            double &v0 = val[vi.slot][0], &v1 = val[vi.slot][1], &v2 = val[vi.slot][2];
            if constexpr (RES)
              {
                const double *un = (vi.oz == 0 ? u_own : u_half) + (vi.ox + H3X * vi.oy);
                dot += fma(v2, un[2 * NR3], fma(v1, un[NR3], v0 * un[0]));
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
            if constexpr (!PLANE)
              {
                constexpr int o_lo = (vi.ox + 1) + 3 * (vi.oy + 1) + 9 * (vi.oz + 1);
                double *dst = stage_half + o_lo * 3;
                dst[0] = v0;
                dst[1] = v1;
                dst[2] = v2;
              }
          }
      });
      if constexpr (PLANE)
        {
          if constexpr (NS == 1)
            {
              constexpr int o = slot_pos_of(W, 0);
              double *dst = stage_row + o * 3;
              dst[upper ? 1 : 0] = pair_sum_across_halves(val[0][0], val[0][1]);
              dst[2] = add_across_halves(val[0][2]); // both halves: the same value to the same address
            }
          else
            {
              constexpr int delta = slot_pos_of(W, 1) - slot_pos_of(W, 0);
              static_assert(NS == 2 || slot_pos_of(W, NS - 1) - slot_pos_of(W, NS - 2) == delta, "one stride for all slot pairs of a set");
              double *dst = stage_row + (upper ? delta * 3 : 0);
#pragma unroll
              for (int sp = 0; sp < NS / 2; ++sp)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                  dst[slot_pos_of(W, 2 * sp) * 3 + k] = pair_sum_across_halves(val[2 * sp][k], val[2 * sp + 1][k]);
            }
        }
      if constexpr (RES)
        return add_across_halves(dot);
      else
        return 0.0;
    }

    // =====================================================================================
    // Round 5: the workgroup MARCHES over a chunk of z-planes of its 8 x 4 tile.  The phases of a plane are those of
    // rounds 2-4 (w*g -> moment tables of the two cell layers -> node phase / copy-out per row component), but the nodal
    // halo is a ring of three planes in LDS and the one new plane a step needs (60 nodes, 5 fields + the flag byte, and
    // the row offsets of the next 32 rows) is REQUESTED one plane ahead with global -> LDS transfers
    // (global_load_lds_*, pfm_dma.h) by the three waves that have no part in the w*g phase.  The phase clock of round 4
    // showed the halo loads of a tile (~6k cycles of global latency under the store stream + the barrier behind them,
    // 8k of a tile's 25k cycles) fully exposed: with one tile per workgroup nothing else of that workgroup can run.
    //   vmcnt counts loads and stores in order, so the wait for the transfers of a step is vmcnt(#stores the wave has
    //   issued behind them) -- 18 on a regular tile (6 per row component) -- and never drains the copy-out.

    template <bool HET, bool RES>
    struct UuShared
    {
      // landing zone of the requests for the NEXT plane (M0 addresses the first 64 KB of the workgroup's LDS: keep first)
      double raw_po[64], raw_poo[64];  // [hn]: dwords 2 hn, 2 hn + 1 fetched by lanes 2 (hn % 32), 2 (hn % 32) + 1 of wave 6 + hn / 32
      double raw_u[RES ? 3 : 1][64];
      unsigned raw_flag[128];          // [2 hn]: the node's flag byte, zero-extended
      long long raw_row[NN3];          // nadj_ptr of the next plane's rows (wave 7)
      unsigned raw_mask[2 * NN3];      // [2 nl]: both lanes of a row fetch its mask
      int raw_rid[NN3];                // the row ids themselves (-1: no row of this launch)
      double tab[TABSZ3];              // moment tables (layout: tab_off3); layer 1 stored z-mirrored.  Dead once every wave holds
                                       // its table values: [0, NN3 STG) = second staging buffer, behind it the partial sums of
                                       // the residual rows (RES)
      double stage[NN3 * STG];         // staged rows [node][81]; w*g(q) [27][90] during the cell phase
      double po[NR3], poo[RES ? 1 : NR3]; // ring: plane p in slot (p - (kA - 1)) % 4.  RES => staggered: one combined field
      double u[RES ? 3 * NR3 : 1];     // displacements of the halo nodes [component][slot][node]
      double lam[HET ? CS3 : 1], mu[HET ? CS3 : 1];
      long long rowbase[2][NN3];       // by plane parity: written for plane k + 1 while the copy-out of plane k reads its own
      unsigned mask[2][NN3];
      int resrow[2][RES ? NN3 : 1];
      double pres[RES ? 3 * NN3 : 1]; // pressure part of the residual rows of the current plane [component][node]
      unsigned char ok[NR3], flag[NR3];
      int anyflag[4][2];               // per ring slot and request wave: some node of the plane carries a displacement flag
      int irregular[2];                // by plane parity: some row is not a full, lattice-ordered 27-neighbour row
    };
    static_assert(27 * CS3 <= NN3 * STG, "w*g scratch must fit in the staging buffer");
    static_assert(NN3 * STG <= 4 * TGRP3 && 2 * 8 * NN3 <= TGRP3, "second staging buffer in front of group 4, the residual sums inside it");
    using UuSharedRes_ = UuShared<false, true>;
    static_assert(offsetof(UuSharedRes_, tab) < 65536, "landing zone within reach of M0");

    // The kernel's only argument.  Inside the march every phase re-reads what it needs from the kernel-argument segment
    // through a pointer the compiler cannot see through (uu_args): values that are invariant over the plane loop would
    // otherwise be hoisted in front of it and kept -- or spilled -- across all phases of a 128-register budget (the first
    // build of the march spilled 100-230 vector and 50 scalar registers that way).
    struct UuArgs
    {
      DevView v;
      CartView cv;
      MatScal S; // per-launch scalars, by value: one scalar load from the argument segment, not two dependent ones
      double *vals;
      unsigned long long *dbg;
      double *res_pde;
      int zc;
    };
    __device__ __forceinline__ const UuArgs &uu_args()
    {
      auto p = __builtin_amdgcn_kernarg_segment_ptr();
      return *(const UuArgs *)p;
    }
#define UU_ENV()                                                                                                             \
  const UuArgs &A = uu_args();                                                                                               \
  const DevView &v = A.v;                                                                                                    \
  const CartView &cv = A.cv;                                                                                                 \
  const MatScal &S = A.S;                                                                                                    \
  int t = threadIdx.x;                                                                                                       \
  asm volatile("" : "+v"(t));                                                                                                \
  const int lane = t & 63;                                                                                                   \
  (void)v;                                                                                                                   \
  (void)cv;                                                                                                                  \
  (void)S;                                                                                                                   \
  (void)lane

    template <int NCOL /* 3 blocked, 4 interleaved */, bool CLK = false /* profiling only */,
              bool HET = false /* per-cell Lame coefficients (CartView::cell_lam) */,
              bool RES = false /* also writes the displacement rows of the residual (res_pde) */>
    __global__ __launch_bounds__(NT3, 4) void k_cart_uu3(UuArgs args_in_kernarg_segment)
    {
      (void)args_in_kernarg_segment;
      long long tclk = 0;
      unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      unsigned long long acc6[4] = {0, 0, 0, 0}; // wave 7: wait for the requests, landing, requests, -
      unsigned long long accc[4] = {0, 0, 0, 0}; // per wave, component 1: copy-out of component 0 | arithmetic | wait at the barrier
      long long tclkc = 0;
      auto stampc = [&](int phase) __attribute__((always_inline)) {
        if constexpr (CLK)
          {
            const long long now = clock64();
            if (phase >= 0)
              accc[phase] += (unsigned long long)(now - tclkc);
            tclkc = now;
          }
      };
      long long tclk6 = 0;
      auto stamp6 = [&](int phase) __attribute__((always_inline)) {
        if constexpr (CLK)
          {
            const long long now = clock64();
            if (phase >= 0)
              acc6[phase] += (unsigned long long)(now - tclk6);
            tclk6 = now;
          }
      };
      auto stamp = [&](int phase) __attribute__((always_inline)) {
        if constexpr (CLK)
          {
            const long long now = clock64();
            if (phase >= 0)
              acc[phase] += (unsigned long long)(now - tclk); // in registers: a global read-modify-write would measure the store drain
            tclk = now;
          }
      };
      stamp(-1);
      __shared__ UuShared<HET, RES> sh;
      double *const s_tab = sh.tab, *const s_stage = sh.stage;
      double *const s_part = sh.tab + 4 * TGRP3; // K u per [component & 1][wave = slot set][node] (RES): see tab_off3
      double *const s_pres = sh.pres;            // pressure part of the residual [component][node] (RES)

      int i0, j0, kA, kB;
      const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform: scalar branches between the slot sets
      {
        UU_ENV();
        const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1, OWZ = cv.o1[2] - cv.o0[2] + 1;
        const int ntx = (OWX + T3X - 1) / T3X, nty = (OWY + T3Y - 1) / T3Y;
        const int zc = A.zc, nch = (OWZ + zc - 1) / zc;
        int bid = xcd_tile_index();
        const bool listed = cv.tile_sel == 2 && cv.bnd_uu3 != nullptr; // compact launch over the boundary tiles (zc == 1)
        if (listed)
          {
            if (bid >= cv.n_bnd_uu3)
              return;
            bid = cv.bnd_uu3[bid];
          }
        if (bid >= ntx * nty * nch)
          return; // padding of the XCD-aware grid
        const int tix = bid % ntx, tiy = (bid / ntx) % nty, chunk = bid / (ntx * nty);
        i0 = cv.o0[0] + tix * T3X;
        j0 = cv.o0[1] + tiy * T3Y;
        kA = cv.o0[2] + chunk * zc;
        kB = min(kA + zc, cv.o1[2] + 1); // node planes [kA, kB)
        if (!listed && cart_tile_skipped(cv, cart_range_has_ghost(cv, 0, i0 - 1, i0 + T3X) || cart_range_has_ghost(cv, 1, j0 - 1, j0 + T3Y) ||
                                                 cart_range_has_ghost(cv, 2, kA - 1, kA + 1)))
          return; // overlapped assembly (zc == 1): the other launch owns this tile

        // ---- prologue: nodal planes kA - 1 .. kA + 2 and the row info of plane kA through registers (one exposed round
        // trip per chunk)
        stamp(0);
        if (t < NR3)
          {
            const int li = t % H3X, lj = (t / H3X) % H3Y, lk = t / NHP3;
            const int gi = i0 - 1 + li, gj = j0 - 1 + lj, gk = kA - 1 + lk;
            double a = 0.0, b = 0.0, uu[3] = {0.0, 0.0, 0.0};
            unsigned char f = 0;
            bool in = gi >= 0 && gi < cv.NX && gj >= 0 && gj < cv.NY && gk >= 0 && gk < cv.NZ;
            const int n = in ? cart_local_id(cv, gi, gj, gk) : -1;
            in = n >= 0; // (a level lattice of the overlay holds -1 where the level has no node)
            if (in)
              {
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
                sh.u[t] = uu[0];
                sh.u[(RES ? NR3 : 0) + t] = uu[1];
                sh.u[(RES ? 2 * NR3 : 0) + t] = uu[2];
              }
            else
              sh.poo[RES ? 0 : t] = b;
            sh.ok[t] = in;
            sh.po[t] = a;
            sh.flag[t] = f;
            // flags of the four planes, per plane and per wave that holds a part of it (plane q = threads [60 q, 60 q + 60):
            // wave 0 | waves 0, 1 | waves 1, 2 | waves 2, 3): no atomics, no init barrier
            const int lk_ = t / NHP3;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              {
                const unsigned long long any = __ballot((f & 7u) != 0 && lk_ == q);
                const int w_first = (q * NHP3) >> 6, w_last = (q * NHP3 + NHP3 - 1) >> 6;
                if (lane == 0 && (wave == w_first || wave == w_last))
                  sh.anyflag[q][wave == w_first ? 0 : 1] = any != 0;
              }
            if (t == 0)
              sh.anyflag[0][1] = 0; // plane 0 lies in wave 0 alone
          }
        else if (t >= 256 && t < 256 + NN3)
          {
            const int nl = t - 256, li = nl % T3X, lj = nl / T3X;
            const int gi = i0 + li, gj = j0 + lj;
            long long base = -1;
            unsigned mask = 0u;
            const int r = (gi <= cv.o1[0] && gj <= cv.o1[1]) ? cart_row_id(cv, gi, gj, kA) : -1;
            if (r >= 0)
              {
                base = (long long)NCOL * NCOL * v.nadj_ptr[r];
                mask = cv.nbr_mask[r];
                if constexpr (RES)
                  sh.resrow[0][nl] = r;
              }
            sh.rowbase[0][nl] = base;
            sh.mask[0][nl] = mask;
            const unsigned long long irr = __ballot(mask != 0x7ffffffu); // fewer than 27 neighbours, or not an owned node
            if (nl == 0)
              sh.irregular[0] = irr != 0;
          }
        stamp(7); // thread 0: its own halo loads have returned and are stored
      }

      // residual row of component c of a plane: sum of the 8 slot sets in a fixed order, constrained rows get 0
      // (cracks.cc:2440-2456).  The sums of component c are stored BEHIND the barrier of component c (their LDS is table
      // storage until then) and read behind the next barrier; par_row / s_row = row parity / ring slot of that plane
      auto residual_out = [&](int c, int par_row, int s_row) __attribute__((always_inline)) {
        if constexpr (RES)
          {
            UU_ENV();
            if (t < NN3 && sh.rowbase[par_row][t] >= 0)
              {
                double sum = -s_pres[RES ? c * NN3 + t : 0];
#pragma unroll
                for (int w = 0; w < 8; ++w)
                  sum += s_part[((c & 1) * 8 + w) * NN3 + t];
                const int li = t % T3X, lj = t / T3X;
                const int row = sh.resrow[par_row][RES ? t : 0];
                const bool con = (sh.flag[s_row * NHP3 + (li + 1) + H3X * (lj + 1)] >> c) & 1u;
                const long long di = (v.layout == PFM_LAYOUT_INTERLEAVED) ? (long long)row * 4 + c : (long long)row * 3 + c;
                A.res_pde[di] = con ? 0.0 : -sum;
              }
          }
      };

      int it = 0;
      bool regular_prev = false; // the last step's copy-out took the regular path (its store count is known)
#pragma unroll 1
      for (int k = kA; k < kB; ++k, ++it)
        {
          // ring slots of the planes k - 1, k, k + 1 and of plane k + 2 (landing in this step: the slot of plane k - 2)
          const int s0 = it & 3, s1 = (it + 1) & 3, s2 = (it + 2) & 3, s3 = (it + 3) & 3, par = it & 1;
          const bool more = k + 1 < kB;
          lds_barrier(); // (first plane: the prologue's planes; later: the last plane's staged rows and residual sums)
          stamp(0);
          if (it > 0)
            residual_out(2, par ^ 1, s0); // component 2 of the PREVIOUS plane: its sums were stored behind its last barrier
          if constexpr (HET)
            {
              UU_ENV();
              if (t >= 320 && t < 320 + CS3) // (in front of the requests of the same waves: its loads are waited for here)
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
                  sh.lam[HET ? cs : 0] = la;
                  sh.mu[HET ? cs : 0] = mu;
                }
            }
          // ---- waves 5..7 have no part in the w*g phase, waves 6, 7 none in the moment phase either.  During w*g they LAND what
          // they requested one step ago: waves 6, 7 plane k + 2 (nodes [32 (wave - 6), +32) of the plane) into the ring slot
          // of plane k - 2 -- their requests for plane k + 3 follow in the moment phase --, wave 5 the rows of plane k, and
          // requests those of plane k + 1 at once.  vmcnt counts loads and stores in order: since their requests these waves
          // have issued a known number of copy-out stores on a regular tile (copy_out), so vmcnt(that number) waits for the
          // requests and never for the stores of the last step
          if (wave >= 5)
            {
              UU_ENV();
              if (it > 0)
                {
                  stamp6(-1);
                  // stores of this wave since its requests, on a regular tile: at least 2 per row component in waves 5, 6, 7
                  // (their third position lies outside the tile; the single values are wave 0's since round 6)
                  if (!regular_prev)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                  else
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                  stamp6(0);
                }
              if (wave >= 6 && it > 0 && more)
                {
                  const int hn = 32 * (wave - 6) + lane;
                  bool fl = false;
                  if (lane < 32 && hn < NHP3)
                    {
                      // (absent nodes: the requesting lanes left zeros and the marker bit 8 of the flag word)
                      double a = sh.raw_po[hn], b = sh.raw_poo[hn];
                      unsigned f = sh.raw_flag[2 * hn];
                      const bool in = (f & 0x100u) == 0;
                      f &= 0xffu;
                      if (!S.monolithic)
                        a = S.use_old ? a : b + S.tfac * (a - b);
                      const int d = s3 * NHP3 + hn;
                      if constexpr (RES)
                        {
#pragma unroll
                          for (int c = 0; c < 3; ++c)
                            sh.u[(RES ? c * NR3 : 0) + d] = sh.raw_u[RES ? c : 0][hn];
                        }
                      else
                        sh.poo[RES ? 0 : d] = b;
                      sh.po[d] = a;
                      sh.ok[d] = in;
                      sh.flag[d] = (unsigned char)f;
                      fl = (f & 7u) != 0;
                    }
                  const unsigned long long any = __ballot(fl);
                  if (lane == 0)
                    sh.anyflag[s3][wave - 6] = any != 0;
                }
              if (wave == 5)
                {
                  if (it > 0)
                    {
                      unsigned mask = 0x7ffffffu;
                      if (lane < NN3)
                        {
                          const long long off = sh.raw_row[lane];
                          const long long base = off < 0 ? -1 : (long long)NCOL * NCOL * off;
                          mask = sh.raw_mask[2 * lane];
                          if constexpr (RES)
                            sh.resrow[par][lane] = sh.raw_rid[lane];
                          sh.rowbase[par][lane] = base;
                          sh.mask[par][lane] = mask;
                        }
                      const unsigned long long irr = __ballot(mask != 0x7ffffffu);
                      if (lane == 0)
                        sh.irregular[par] = irr != 0;
                      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the landing zone is read out before it is requested into again
                    }
                  if (more)
                    {
                      const int nl = lane >> 1, li = nl % T3X, lj = nl / T3X;
                      const int gi = i0 + li, gj = j0 + lj;
                      const int rid = (gi <= cv.o1[0] && gj <= cv.o1[1]) ? cart_row_id_sync(cv, gi, gj, k + 1) : -1;
                      const unsigned r = (unsigned)rid;
                      if (rid >= 0)
                        {
                          dma_b32(v.nadj_ptr, 8u * r + 4u * (lane & 1), reinterpret_cast<uint32_t *>(sh.raw_row));
                          dma_b32(cv.nbr_mask, 4u * r, sh.raw_mask);
                        }
                      else
                        {
                          reinterpret_cast<uint32_t *>(sh.raw_row)[lane] = 0xffffffffu; // -1: no row of this launch
                          sh.raw_mask[lane] = 0u;
                        }
                      if (!(lane & 1))
                        sh.raw_rid[nl] = rid;
                    }
                }
              if (it > 0)
                stamp6(1);
            }

          // ---- cell phase a: w*g at the quadrature points, thread <-> (cell, z-level) -> LDS [q][cell]
          {
            UU_ENV();
            if (t < 3 * CS3)
              {
                const int cs = t % CS3, qz = t / CS3;
                const int l = cs / CL3, cy = (cs % CL3) / C3X, cx = cs % C3X;
                const int hq = cx + H3X * cy;
                const int pa = (l ? s1 : s0) * NHP3 + hq, pb = (l ? s2 : s1) * NHP3 + hq; // lower / upper nodal plane of the layer
                // all reads of the cell in one batch (the two validity bytes next to the nodal values: one LDS round trip);
                // a cell outside the mesh reads zeros and is zeroed
                const bool valid = sh.ok[pa] && sh.ok[pb + 1 + H3X];
                double wg[9];
                {
                  double po[8], poo[8];
                  if (!S.monolithic)
                    {
#pragma unroll
                      for (int b = 0; b < 4; ++b)
                        {
                          po[b] = sh.po[pa + (b & 1) + H3X * ((b >> 1) & 1)];
                          po[b + 4] = sh.po[pb + (b & 1) + H3X * ((b >> 1) & 1)];
                        }
                      cell_wg_plane_lin<true>(po, S, qz, wg);
                    }
                  else
                    {
#pragma unroll
                      for (int b = 0; b < 4; ++b)
                        {
                          const int ha = pa + (b & 1) + H3X * ((b >> 1) & 1), hb = pb + (b & 1) + H3X * ((b >> 1) & 1);
                          po[b] = sh.po[ha];
                          po[b + 4] = sh.po[hb];
                          poo[b] = sh.poo[RES ? 0 : ha];
                          poo[b + 4] = sh.poo[RES ? 0 : hb];
                        }
                      cell_wg_plane<true>(po, poo, S, qz, wg);
                    }
                }
#pragma unroll
                for (int q = 0; q < 9; ++q)
                  wg[q] = valid ? wg[q] : 0.0;
#pragma unroll
                for (int q = 0; q < 9; ++q)
                  s_stage[(qz * 9 + q) * CS3 + cs] = wg[q];
              }
          }
          lds_barrier();
          stamp(1);

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
            UU_ENV();
            auto family = [&](auto Ff, auto Ll) __attribute__((always_inline)) {
              constexpr int f = decltype(Ff)::value, l = decltype(Ll)::value;
              constexpr bool mir = l == 1;
              if (lane < CL3)
                {
                  const int cs = l * CL3 + lane;
                  const double *wq = s_stage + cs;
                  const int cyl = lane / C3X;
                  double *out = s_tab + l * TLAY3 + cyl * TROW3 + (lane - cyl * C3X);
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
                          tq[qj] = s9[qj][0] * g1_m(gi, 0) + s9[qj][1] * g1_m(gi, 1) + s9[qj][2] * g1_m(gi, 2);
#pragma unroll
                        for (int gj = 0; gj < 3; ++gj)
                          {
                            const double val = tq[0] * g1_m(gj, 0) + tq[1] * g1_m(gj, 1) + tq[2] * g1_m(gj, 2);
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
                        const double na0 = g1_n(al, 0), na1 = g1_n(al, 1), na2 = g1_n(al, 2);
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
                              t2[qe] = t1[qe][0] * g1_n(be, 0) + t1[qe][1] * g1_n(be, 1) + t1[qe][2] * g1_n(be, 2);
                            const int bes = mb ? 1 - be : be;
#pragma unroll
                            for (int g = 0; g < 3; ++g)
                              {
                                const double val = t2[0] * g1_m(g, 0) + t2[1] * g1_m(g, 1) + t2[2] * g1_m(g, 2);
                                const int gm = mz ? 2 - g : g;
                                out[tab_off3(27 + p * 12 + al * 6 + bes * 3 + gm)] = mb ? -val : val;
                              }
                          }
                      }
                  }
                }
            };
            using std::integral_constant;
            switch (wave)
              {
                case 0: family(integral_constant<int, 0>{}, integral_constant<int, 0>{}); break;
                case 1: family(integral_constant<int, 0>{}, integral_constant<int, 1>{}); break;
                case 2: family(integral_constant<int, 1>{}, integral_constant<int, 0>{}); break;
                case 3: family(integral_constant<int, 1>{}, integral_constant<int, 1>{}); break;
                case 4: family(integral_constant<int, 2>{}, integral_constant<int, 0>{}); break;
                case 5: family(integral_constant<int, 2>{}, integral_constant<int, 1>{}); break;
                default:
                  // waves 6, 7 REQUEST plane k + 3 (the upper plane of step k + 2): global -> LDS, no registers; the landing zone
                  // was read out in the w*g phase
                  {
                    stamp6(-1);
                    if (k + 2 < kB)
                      {
                        const int kz = k + 3;
                        const int hn = 32 * (wave - 6) + (lane >> 1);
                        const int gi = i0 - 1 + hn % H3X, gj = j0 - 1 + hn / H3X;
                        bool in = hn < NHP3 && gi >= 0 && gi < cv.NX && gj >= 0 && gj < cv.NY && kz >= 0 && kz < cv.NZ;
                        const int id = in ? cart_local_id_sync(cv, gi, gj, kz) : -1; // (a looked-up id is waited for in its arm)
                        in = id >= 0;
                        const unsigned n = (unsigned)id;
                        if (!in && hn < NHP3)
                          {
                            // no such node (outside the mesh / not a node of this level): the lane writes the neutral values itself
                            const int dw = 64 * (wave - 6) + lane;
                            reinterpret_cast<uint32_t *>(sh.raw_po)[dw] = 0u;
                            reinterpret_cast<uint32_t *>(sh.raw_poo)[dw] = 0u;
                            if constexpr (RES)
                              {
#pragma unroll
                                for (int c = 0; c < 3; ++c)
                                  reinterpret_cast<uint32_t *>(sh.raw_u[RES ? c : 0])[dw] = 0u;
                              }
                            sh.raw_flag[dw] = 0x100u; // bit 8: absent
                          }
                        if (in)
                          {
                            const unsigned boff = 8u * n + 4u * (lane & 1);
                            dma_b32(v.phi_old, boff, reinterpret_cast<uint32_t *>(sh.raw_po) + 64 * (wave - 6));
                            dma_b32(v.phi_oldold, boff, reinterpret_cast<uint32_t *>(sh.raw_poo) + 64 * (wave - 6));
                            if constexpr (RES)
                              {
#pragma unroll
                                for (int c = 0; c < 3; ++c)
                                  dma_b32(v.u[c], boff, reinterpret_cast<uint32_t *>(sh.raw_u[RES ? c : 0]) + 64 * (wave - 6));
                              }
                            dma_u8(v.node_flags, n, sh.raw_flag + 64 * (wave - 6));
                          }
                      }
                    stamp6(2);
                  }
                  break;
              }
          }
          // the uniform constants of the node phase: requested in front of the barrier, so that the scalar loads are under way
          // while the wave waits for the others
          UuCoef K;
          {
            UU_ENV();
#pragma unroll
            for (int c = 0; c < 3; ++c)
              {
#pragma unroll
                for (int kk = 0; kk < 3; ++kk)
                  K.cA[c][kk] = S.cA[c][kk];
                K.cTl[c] = S.cTl[c];
                K.cTm[c] = S.cTm[c];
                K.gA[c] = S.ih[c] * S.ih[c];
                K.cT[c] = S.cT[c];
              }
          }
          lds_barrier();
          stamp(2);

          // ---- node phase + copy-out.  Every wave first reads the 36 table values of its slot set (4 visits x 9) in ONE
          // batch and keeps them in registers for all three row components: 36 LDS reads per lane instead of 84, one
          // latency instead of a chain of them, and the tables are dead afterwards -- their LDS becomes the second staging
          // buffer, so that the copy-out of component c overlaps the arithmetic of component c + 1 (one barrier per
          // component instead of two).
          bool regular_tile;
          {
            UU_ENV();
            const bool upper = lane >= 32;
            const int nl_lane = lane & 31;
            const int ti = nl_lane % T3X, tj = nl_lane / T3X;
            const int hc2 = (ti + 1) + H3X * (tj + 1); // in-plane halo index of the node
            // the cell "below-left" of the node in its layer: lower half -> layer 0, upper half -> layer 1 (mirrored tables)
            const double *lane_base = s_tab + (upper ? TLAY3 : 0) + (tj + 1) * TROW3 + (ti + 1);
            const int lane_cs = (upper ? CL3 : 0) + (tj + 1) * C3X + (ti + 1); // the same cell in [layer][cy][cx] order (s_lam, s_mu)
            const bool masked = (sh.anyflag[s0][0] | sh.anyflag[s0][1] | sh.anyflag[s1][0] | sh.anyflag[s1][1] | sh.anyflag[s2][0] | sh.anyflag[s2][1]) != 0;
            regular_tile = (NCOL == 3) && sh.irregular[par] == 0;
            const int hc = s1 * NHP3 + hc2, hh = (upper ? s2 : s0) * NHP3 + hc2; // the node in its own plane / in the half's other plane
            const unsigned row_flag = sh.flag[hc];
            const unsigned char *flag_own = sh.flag + hc, *flag_half = sh.flag + hh;

            auto copy_out = [&](int c, const double *__restrict__ stage) __attribute__((always_inline)) {
              const long long *rowbase = sh.rowbase[par];
              double *__restrict__ vals = A.vals;
              if (regular_tile)
                {
                  // Round 5: 16-byte stores.  A workgroup's copy-out is bound by how fast the CU's store path takes
                  // instructions (tools/microbench/stw.hip: ~16 cycles per 512-byte instruction, 1.45x the bytes per cycle
                  // with 16 bytes per lane), and since the barrier of a component waits for the copy-out of the last one in
                  // EVERY wave, that time is on the critical path of the plane (phase clock: 1.3k of the 3.4k cycles of a
                  // component).  thread <-> (node group g < 12, element pair ep < 40): nodes g, g + 12, g + 24, the pairs
                  // (2 ep, 2 ep + 1) of their 81 staged values; threads 0..31: the 81st value of node t (the pairs start at thread 32).  No
                  // division per position, every read in flight before the first store.
                  struct __attribute__((packed, aligned(8))) D2
                  {
                    double a, b;
                  };
                  constexpr int NG = 12, NIT = 3;
                  int tq = t;
                  asm volatile("" : "+v"(tq)); // (g, ep) are recomputed per component, not kept live across the node phases
                  // (round 6: the 32 single values go with wave 0, whose slot set is the lightest, instead of wave 7, whose is
                  // the heaviest -- the wave that runs both arms of this branch was the one every barrier waited for)
                  if (tq >= 32)
                    {
                      const int pq = tq - 32, g = pq / 40, ep = pq - g * 40;
                      long long rb[NIT];
                      D2 val[NIT];
#pragma unroll
                      for (int i = 0; i < NIT; ++i)
                        {
                          const int nl = min(g + NG * i, NN3 - 1);
                          rb[i] = rowbase[nl];
                          val[i].a = lds_read64(stage + nl * STG + 2 * ep);
                          val[i].b = lds_read64(stage + nl * STG + 2 * ep + 1);
                        }
                      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                      for (int i = 0; i < NIT; ++i)
                        if (g + NG * i < NN3)
                          *reinterpret_cast<D2 *>(vals + rb[i] + (c * STG + 2 * ep)) = val[i];
                    }
                  else
                    {
                      const int nl = tq;
                      vals[rowbase[nl] + (c * STG + STG - 1)] = stage[nl * STG + STG - 1];
                    }
                }
              else
                {
                  // rows at the faces of the box / partial tiles / next to ghost columns: thread <-> (row, lattice offset o,
                  // column component); the CSR slot of offset o is its rank among the offsets that exist, or the row's
                  // permutation of that rank
                  const unsigned *rowmask = sh.mask[par];
                  constexpr int rowlen = 27 * NCOL;
                  for (int f = t; f < NN3 * rowlen; f += NT3)
                    {
                      const int nl = f / rowlen, e = f - nl * rowlen;
                      const int o = e / NCOL, d = e - o * NCOL;
                      const long long base = rowbase[nl];
                      const unsigned mask = rowmask[nl];
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
                      for (int kk = 0; kk < 3; ++kk)
                        {
                          const int i = (kk == 0) ? 1 : 0, j = (kk == 2) ? 1 : 2;
                          const double s4 = (lds_read64(cell + tab_off3(idxA3(kk, a[i], a[j]))) + lds_read64(cell + tab_off3(idxA3(kk, a[i] + 1, a[j])))) +
                                            (lds_read64(cell + tab_off3(idxA3(kk, a[i], a[j] + 1))) + lds_read64(cell + tab_off3(idxA3(kk, a[i] + 1, a[j] + 1))));
                          const double mom = s4 - (s4 != 0.0 ? kv4 : 0.0); // absent cell: all tables are zero
                          const bool neg = (kk == 2) ? upper : (a[kk] == 0); // sign of dN_a/dx_k (upper layer: z-mirrored tables)
                          pres[kk] += neg ? -mom : mom;
                        }
                    });
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk)
                      {
                        const double pk = add_across_halves(pres[kk]) * (S.pc_res * S.ih[kk]);
                        if (!upper)
                          s_pres[RES ? kk * NN3 + nl_lane : 0] = pk; // read behind the barrier of component 1 at the earliest
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
              uu_load_visit<W, V, 3>(lane_base, tv[V]);
              if constexpr (HET)
                {
                  constexpr Vis vi = visit_of(W, V);
                  const int cs = lane_cs + (vi.ey * C3X + vi.ex);
                  lamv[V] = sh.lam[HET ? cs : 0];
                  muv[V] = sh.mu[HET ? cs : 0];
                }
            }))
            const double *u_own = sh.u + (RES ? hc : 0), *u_half = sh.u + (RES ? hh : 0);
            // buffer 0 = the w*g scratch (free since the moment phase), buffer 1 = the table storage: written after the
            // barrier of component 0, which every wave passes with its table values in registers
            double *st0 = s_stage + nl_lane * STG, *st1 = s_tab + nl_lane * STG;
            const int hs = upper ? 18 * 3 : 0;
            double ku = 0.0;
            auto part_store = [&](int c) __attribute__((always_inline)) {
              if constexpr (RES)
                {
                  if (!upper)
                    s_part[((c & 1) * 8 + wave) * NN3 + nl_lane] = ku;
                }
            };
#define PFM_COMPONENT(C, ST)                                                                                                 \
  if (masked)                                                                                                                \
    {                                                                                                                        \
      PFM_PER_SET((ku = uu_row_component<W, C, true, HET, RES>(tv, K, lamv, muv, ST, ST + hs, upper, row_flag, flag_own, flag_half, \
                                                               u_own, u_half)))                                             \
    }                                                                                                                        \
  else                                                                                                                       \
    {                                                                                                                        \
      PFM_PER_SET((ku = uu_row_component<W, C, false, HET, RES>(tv, K, lamv, muv, ST, ST + hs, upper, row_flag, flag_own, flag_half, \
                                                                u_own, u_half)))                                            \
    }
            PFM_COMPONENT(0, st0)
            lds_barrier();
            stamp(3);
            PFM_PER_SET(static_for<4>([&](auto Vv) __attribute__((always_inline)) {
              uu_load_visit<W, decltype(Vv)::value, 4>(lane_base, tv[decltype(Vv)::value]); // T^yz, behind the copy-out
            }))
            stampc(-1);
            part_store(0);
            copy_out(0, s_stage);
            stampc(0);
            PFM_COMPONENT(1, st1)
            stampc(1);
            lds_barrier();
            stampc(2);
            stamp(4);
            PFM_PER_SET(static_for<4>([&](auto Vv) __attribute__((always_inline)) {
              uu_load_visit<W, decltype(Vv)::value, 2>(lane_base, tv[decltype(Vv)::value]); // T^xz again
            }))
            residual_out(0, par, s1);
            part_store(1);
            copy_out(1, s_tab);
            PFM_COMPONENT(2, st0)
            lds_barrier();
            stamp(5);
            residual_out(1, par, s1);
            part_store(2); // buffer 0 again: the reads of component 0 lie behind the last barrier
            copy_out(2, s_stage);
#undef PFM_COMPONENT
#undef PFM_PER_SET
          }
          regular_prev = regular_tile;
          stamp(6);
        }
      if constexpr (RES)
        {
          lds_barrier();
          residual_out(2, (it - 1) & 1, it & 3); // the last plane k = kB - 1: parity (it - 1) & 1, ring slot ((it - 1) + 1) & 3
        }
      if constexpr (CLK)
        {
          if (threadIdx.x == 0)
            for (int i = 0; i < 8; ++i)
              uu_args().dbg[(size_t)blockIdx.x * 32 + i] = acc[i];
          if (threadIdx.x == 448)
            for (int i = 0; i < 4; ++i)
              uu_args().dbg[(size_t)blockIdx.x * 32 + 8 + i] = acc6[i];
          if (threadIdx.x == 0)
            for (int i = 0; i < 4; ++i)
              uu_args().dbg[(size_t)blockIdx.x * 32 + 12 + i] = accc[i];
          if ((threadIdx.x & 63) == 0) // every wave: copy-out and arithmetic of component 1
            {
              uu_args().dbg[(size_t)blockIdx.x * 32 + 16 + 2 * (threadIdx.x >> 6)] = accc[0];
              uu_args().dbg[(size_t)blockIdx.x * 32 + 17 + 2 * (threadIdx.x >> 6)] = accc[1];
            }
        }
    }
#undef UU_ENV
  } // namespace

  int launch_cart_uu3(const DevView &v, const CartView &cv, const pfm_params &p, double *vals_uu, hipStream_t s,
                      const void *d_scal, double *res_pde, int lds_total /* LDS bytes per workgroup to pad to (launch_cart_matrix), 0: none */)
  {
    int rc = ensure_g1();
    if (rc)
      return rc;
    (void)d_scal; // (the scalar tables travel by value in the kernel's argument segment since round 5)
    const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1, OWZ = cv.o1[2] - cv.o0[2] + 1;
    const int ntx = (OWX + T3X - 1) / T3X, nty = (OWY + T3Y - 1) / T3Y;
    const bool listed = cv.tile_sel == 2 && cv.bnd_uu3 != nullptr;
    // planes per workgroup: the interior / boundary launches of an overlapped assembly select tiles plane by plane (zc = 1:
    // every plane through the prologue's register loads, as in rounds 1-4); otherwise chunks that fill the dispatch rounds
    static const int zc_force = getenv("PFM_UU_ZC") ? atoi(getenv("PFM_UU_ZC")) : 0; // tuning only
    const int zc = cv.tile_sel != 0 ? 1 : (zc_force > 0 ? std::min(zc_force, OWZ) : choose_zchunk((long long)ntx * nty, OWZ, 8, 48, 2));
    const int nch = (OWZ + zc - 1) / zc;
    const unsigned nb = listed ? (unsigned)cv.n_bnd_uu3 : (unsigned)(ntx * nty * nch);
    if (nb == 0)
      return PFM_OK;
    const dim3 grid(xcd_grid(nb)), block(NT3);
    const bool il = v.layout == PFM_LAYOUT_INTERLEAVED, het = cv.cell_lam != nullptr, res = res_pde != nullptr;
    UuArgs ka{v, cv, make_mat_scal(p, cv), vals_uu, nullptr, res_pde, zc};
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
      hipLaunchKernelGGL((k_cart_uu3<NC, false, HETV, RESV>), grid, block, pad, s, ka);                                      \
    }                                                                                                                        \
  while (0)
    if (getenv("PFM_UU_CLK") && !il && !het) // profiling only
      {
        static unsigned long long *d_dbg = nullptr;
        static size_t nd_cap = 0;
        const size_t nd = (size_t)xcd_grid(nb) * 32;
        if (nd > nd_cap)
          {
            if (d_dbg)
              (void)hipFree(d_dbg);
            if (hipMalloc((void **)&d_dbg, nd * sizeof(unsigned long long)) != hipSuccess)
              return PFM_ERR_HIP;
            nd_cap = nd;
          }
        (void)hipMemsetAsync(d_dbg, 0, nd * sizeof(unsigned long long), s);
        ka.dbg = d_dbg;
        if (res)
          hipLaunchKernelGGL((k_cart_uu3<3, true, false, true>), grid, block, 0, s, ka);
        else
          hipLaunchKernelGGL((k_cart_uu3<3, true>), grid, block, 0, s, ka);
        std::vector<unsigned long long> hall(nd);
        (void)hipMemcpy(hall.data(), d_dbg, nd * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long h[32] = {};
        for (size_t i = 0; i < nd; ++i)
          h[i % 32] += hall[i];
        const char *names[8] = {"top barrier", "w*g", "moments", "tables+node c0", "copy c0+node c1", "copy c1+node c2",
                                "copy c2+landing", "prologue loads (per chunk)"};
        const double planes = (double)ntx * nty * OWZ;
        fprintf(stderr, "[k_cart_uu3 phase clock, thread 0, cycles per plane; zc=%d]", zc);
        for (int i = 0; i < 8; ++i)
          fprintf(stderr, " %s=%.0f", names[i], (double)h[i] / (i == 7 ? (double)nb : planes));
        fprintf(stderr, " | wave 7: wait for the requests=%.0f landing=%.0f requests=%.0f\n", (double)h[8] / planes, (double)h[9] / planes,
                (double)h[10] / planes);
        fprintf(stderr, "[k_cart_uu3 phase clock] wave 0, component 1: copy-out of component 0=%.0f arithmetic=%.0f barrier=%.0f\n", (double)h[12] / planes,
                (double)h[13] / planes, (double)h[14] / planes);
        fprintf(stderr, "[k_cart_uu3 phase clock] component 1 per wave (copy-out of component 0 / arithmetic):");
        for (int w = 0; w < 8; ++w)
          fprintf(stderr, " W%d %.0f/%.0f", w, (double)h[16 + 2 * w] / planes, (double)h[17 + 2 * w] / planes);
        fprintf(stderr, "\n");
      }
    else if (il)
      {
        if (het)
          PFM_UU3(4, true, false); // heterogeneous material: the residual kernel runs
        else if (res)
          PFM_UU3(4, false, true);
        else
          PFM_UU3(4, false, false);
      }
    else
      {
        if (het)
          PFM_UU3(3, true, false); // heterogeneous material: the residual kernel runs
        else if (res)
          PFM_UU3(3, false, true);
        else
          PFM_UU3(3, false, false);
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
