// pfm_cart_uu5.hip — (u,u) block, row-owner kernel, fifth generation: z-marching pull, nodal planes by LDS-DMA.
//
// Mathematics: 63 moment tables per cell (header of pfm_cart.hip).  Node phase, table numbering and summation order are
// those of k_cart_uu5 (pfm_cart_uu4.hip: lanes 0..31 <-> node n with the 4 cells below its plane, lanes 32..63 <-> the
// same node with the 4 cells above; 8 waves = 8 z-symmetric slot sets; one new cell layer per plane, ring of two layer
// slots in LDS).  What round 3 changed against it:
//
//   * The nodal planes and the CSR row info arrive by global -> LDS transfers (global_load_lds_dword, no staging
//     registers), requested at the top of step k for plane k + 2 and consumed in step k + 1: a full step (~20k cycles)
//     ahead of their first use, where the register pipeline of k_cart_uu5 consumed them after one cell phase, i.e. before
//     the ~3 us a load takes behind the store stream had passed (8 more live registers did not fit).  Waves 0 and 1 are
//     the loaders; they take no part in the copy-out, so their vmcnt only ever counts loads and the wait for them at the
//     end of the step never waits for a store.
//   * RES: the displacement rows of the residual come out of the matrix rows, as in k_cart_uu3<..., RES>
//     (R_u = (alpha_B-1) p sum_q pfx^2 dN/dx_c JxW - K_uu u for the unsplit law), so that the march is the default path of
//     a full assembly.  The displacements of the three planes around the node plane travel in the same nodal ring.
//   * copy-out by waves 2..7: thread <-> (node group, element) with the element fixed, 8 nodes per thread.
#include "pfm_internal.h"
#include "pfm_cart_common.h"

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

namespace pfm
{
  namespace
  {
    constexpr int NNUM5 = 63;
    constexpr int ZS5 = 22;           // tables per z-digit
    constexpr int NHP5 = H3X * H3Y;   // 60 halo nodes per plane
    constexpr int TABL5 = NNUM5 * CL3; // doubles per layer slot (2835)

    // ---- table numbering: number = 22 z + j for the z-dependent families, 54.. for A^z
    //   A^x[g_y][g_z]       z = g_z, j = g_y                      (A^c[g_i][g_j], (i,j) = other axes ascending)
    //   A^y[g_x][g_z]       z = g_z, j = 3 + g_x
    //   T^xy[al][be][g_z]   z = g_z, j = 6 + 2 al + be
    //   T^xz[al][be][g_y]   z = be,  j = 10 + 3 al + g_y          (z = 0, 1 only)
    //   T^yz[al][be][g_x]   z = be,  j = 16 + 3 al + g_x
    //   A^z[g_x][g_y]       54 + 3 g_x + g_y
    __host__ __device__ constexpr int numA5(int c, int gi, int gj)
    {
      return c == 0 ? ZS5 * gj + gi : (c == 1 ? ZS5 * gj + 3 + gi : 54 + 3 * gi + gj);
    }
    __host__ __device__ constexpr int numT5(int p, int al, int be, int g)
    {
      return p == 0 ? ZS5 * g + 6 + 2 * al + be : (p == 1 ? ZS5 * be + 10 + 3 * al + g : ZS5 * be + 16 + 3 * al + g);
    }
    __host__ __device__ constexpr int pair5(int lo, int hi) { return lo == 0 ? (hi == 1 ? 0 : 1) : 2; }
    __host__ __device__ constexpr int sg5(int bit) { return bit ? 1 : -1; }

    // the 4 (slot, cell) visits of slot set W, in the order k_cart_uu3 summed them
    //   W0: (0,0,0)  W1: (0,0,-1)  W2: (0,+-1,0)  W3: (+-1,0,0)  W4: (0,+-1,-1)  W5: (+-1,0,-1)  W6: (+-1,+-1,0)  W7: (+-1,+-1,-1)
    struct Vis5
    {
      int ox, oy, oz, ex, ey, slot, first, last;
    };
    __host__ __device__ constexpr Vis5 visit5(int W, int v)
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
                  return Vis5{ox, oy, oz, ex, ey, sl, cnt == 0, cnt == total - 1};
                ++n;
                ++cnt;
              }
        }
      return Vis5{0, 0, 0, 0, 0, -1, 0, 0};
    }
    __host__ __device__ constexpr int nslots5(int W) { return (W < 2) ? 1 : (W < 6 ? 2 : 4); }

    // Per-lane pointers to the lane's cell (node offset (0,0)) in its layer slot, shifted by d z-digits for the upper
    // half: b[d + 2], d = -2 .. +1 (lower half: all four equal).
    struct Bases5
    {
      const double *b[4];
    };

    // One table value for both half-waves: the lower half (a_z = 1, b_z = 1 + oz) needs table LO, the upper half
    // (a_z = 0, b_z = -oz) table HI; HI - LO is a multiple of the z-digit stride by construction, so both read
    // base[d] + LO with a compile-time offset.  (Plain loads the compiler tracks: an inline-asm ds_read is "complete" for
    // the register allocator at the end of the statement, and under the register pressure of this kernel it spilled a
    // destination register before the data had landed.)
    template <int LO, int HI, int CELL_OFF>
    __device__ __forceinline__ void tab_read(const Bases5 &B, double &x)
    {
      static_assert((HI - LO) % ZS5 == 0 && (HI - LO) / ZS5 >= -2 && (HI - LO) / ZS5 <= 1, "z-digit shift out of range");
      constexpr int d = (HI - LO) / ZS5;
      x = B.b[d + 2][LO * CL3 + CELL_OFF];
    }

    // the 9 table values one visit needs for all nine (row comp, col comp) entries: A^k (k = 0..2), then per pair
    // p = (lo,hi): X_p = T^p[b_lo][a_hi][g_e], Y_p = T^p[a_lo][b_hi][g_e]
    template <int W, int V>
    __device__ __forceinline__ void uu5_load_visit(const Bases5 &B, double (&tv)[9])
    {
      constexpr Vis5 vi = visit5(W, V);
      constexpr int ax = -vi.ex, ay = -vi.ey, bx = -vi.ex + vi.ox, by = -vi.ey + vi.oy;
      constexpr int co = vi.ey * C3X + vi.ex;           // cell offset relative to the lane's (0,0) cell
      constexpr int aL[3] = {ax, ay, 1}, bL[3] = {bx, by, 1 + vi.oz}; // lower half: cells below the node plane
      constexpr int aU[3] = {ax, ay, 0}, bU[3] = {bx, by, -vi.oz};    // upper half: cells above
      static_for<3>([&](auto Kk) __attribute__((always_inline)) {
        constexpr int k = decltype(Kk)::value;
        constexpr int i = (k == 0) ? 1 : 0, j = (k == 2) ? 1 : 2;
        tab_read<numA5(k, aL[i] + bL[i], aL[j] + bL[j]), numA5(k, aU[i] + bU[i], aU[j] + bU[j]), co>(B, tv[k]);
      });
      static_for<3>([&](auto Pp) __attribute__((always_inline)) {
        constexpr int p = decltype(Pp)::value;
        constexpr int lo = (p == 2) ? 1 : 0, hi = (p == 0) ? 1 : 2, e = 3 - lo - hi;
        tab_read<numT5(p, bL[lo], aL[hi], aL[e] + bL[e]), numT5(p, bU[lo], aU[hi], aU[e] + bU[e]), co>(B, tv[3 + 2 * p]);
        tab_read<numT5(p, aL[lo], bL[hi], aL[e] + bL[e]), numT5(p, aU[lo], bU[hi], aU[e] + bU[e]), co>(B, tv[4 + 2 * p]);
      });
    }

    struct UuCoef5 // uniform constants of the node phase, read once per workgroup
    {
      double cA[3][3], cTl[3], cTm[3];
    };

    // r += entry (C, D) of one visit in the LOWER half's signs (the upper half's differ by one factor -1 for the entries
    // that couple z with x or y: applied to the finished sum)
    template <int W, int V, int C, int D>
    __device__ __forceinline__ void uu5_acc_visit(const double (&tv)[9], const UuCoef5 &K, double &r)
    {
      constexpr Vis5 vi = visit5(W, V);
      constexpr int a[3] = {-vi.ex, -vi.ey, 1}, b[3] = {-vi.ex + vi.ox, -vi.ey + vi.oy, 1 + vi.oz};
      if constexpr (C == D)
        {
#pragma unroll
          for (int k = 0; k < 3; ++k)
            r = fma((sg5(a[k]) * sg5(b[k]) > 0) ? K.cA[C][k] : -K.cA[C][k], tv[k], r);
        }
      else
        {
          constexpr int lo = C < D ? C : D, hi = C < D ? D : C, p = pair5(lo, hi);
          const double t1 = (C < D) ? tv[3 + 2 * p] : tv[4 + 2 * p];
          const double t2 = (C < D) ? tv[4 + 2 * p] : tv[3 + 2 * p];
          r = fma((sg5(a[C]) * sg5(b[D]) > 0) ? K.cTl[p] : -K.cTl[p], t1, r);
          r = fma((sg5(a[D]) * sg5(b[C]) > 0) ? K.cTm[p] : -K.cTm[p], t2, r);
        }
    }

    // x[l] + x[l ^ 32] in every lane, through the VALU (v_permlane32_swap) instead of two LDS bpermutes per double
    __device__ __forceinline__ double add_halves5(double x)
    {
      const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
      const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      return __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
    }
    __device__ __forceinline__ double flip_sign5(double x, unsigned sign_bit)
    {
      return __hiloint2double(__double2hiint(x) ^ (int)sign_bit, __double2loint(x));
    }

    // Row component C of slot set W for both half-waves, from the cached table values: 28 FMAs, the sign of the z-mixed
    // entries for the upper half, the cross-half adds of the oz = 0 slots, 3 staged values per slot.  The staged values
    // are the UNMASKED entries: constraints are applied by the copy-out (only tiles near a constrained node pay for
    // them), and the loader waves form K u for the residual from the staged rows.
    // stage_half = the lane's staged row shifted by 18 slots for the upper half (a slot with oz = -1 completed by the lower
    // half is slot o_lo, its mirror completed by the upper half is o_lo + 18); oz = 0 slots are summed over the halves and
    // stored by both (same value, same address).
    template <int W, int C>
    __device__ __forceinline__ void uu5_row_component(const double (&tv)[4][9], const UuCoef5 &K, unsigned upper_sign,
                                                      double *__restrict__ stage_row, double *__restrict__ stage_half)
    {
      double r0 = 0.0, r1 = 0.0, r2 = 0.0; // the three column components of the current slot
      static_for<4>([&](auto Vv) __attribute__((always_inline)) {
        constexpr int V = decltype(Vv)::value;
        constexpr Vis5 vi = visit5(W, V);
        if constexpr (vi.first)
          r0 = r1 = r2 = 0.0;
        uu5_acc_visit<W, V, C, 0>(tv[V], K, r0);
        uu5_acc_visit<W, V, C, 1>(tv[V], K, r1);
        uu5_acc_visit<W, V, C, 2>(tv[V], K, r2);
        if constexpr (vi.last)
          {
            double v[3] = {r0, r1, r2};
#pragma unroll
            for (int D = 0; D < 3; ++D)
              if ((C == 2) != (D == 2)) // exactly one of the two components is z
                v[D] = flip_sign5(v[D], upper_sign);
            if constexpr (vi.oz == 0)
              {
                v[0] = add_halves5(v[0]);
                v[1] = add_halves5(v[1]);
                v[2] = add_halves5(v[2]);
              }
            constexpr int o_lo = (vi.ox + 1) + 3 * (vi.oy + 1) + 9 * (vi.oz + 1);
            double *dst = (vi.oz == 0 ? stage_row : stage_half) + o_lo * 3;
            dst[0] = v[0];
            dst[1] = v[1];
            dst[2] = v[2];
          }
      });
    }

    // global -> LDS without staging registers (see pfm_cart_phi4.hip): LDS address = M0 + 4 * lane, issued from inline
    // asm so that the compiler does not serialise the requests behind its alias analysis of the __shared__ object
    __device__ __forceinline__ void dma5_b32(const void *base, unsigned byte_off, void *lds)
    {
      const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(byte_off), "s"(base), "s"(l)
                   : "memory");
    }

    template <bool RES>
    struct Lds5
    {
      // destinations of the global -> LDS transfers first: M0 carries a 16-bit LDS offset
      double po[4][NHP5], poo[4][NHP5];  // nodal ring (plane & 3): phi_old (or the combined old field), phi_oldold
      double u[RES ? 3 : 1][4][NHP5];    // RES: displacements [component][plane & 3][halo node]
      long long rowoff[2][NN3];          // plane & 1: node-graph offset of the row (nadj_ptr), -1: not an owned row here
      unsigned mask[2][NN3];
      double stage[NN3 * STG];  // staging buffer 0 [node][81]; w*g scratch [27][45] during the cell phase.  In front of the
                                // tables: the upper half's shifted bases (down to -2 z-digits = -15840 bytes) stay >= 0
      double tab[2][TABL5];     // ring of two cell layers [number][cell]; the dead layer's slot = staging buffer 1
      double kpart[RES ? 2 * 2 * NN3 : 1]; // K u per [component & 1][source: oz = +-1 (wave 0), oz = 0 (wave 1)][node]
      double pres[RES ? 3 * NN3 : 1];     // pressure part of the residual [component][node]
      unsigned char flag[4][NHP5];        // constraint flags; bit 7: the node exists
      int anyflag[4];                     // some node of the plane's halo carries a displacement flag
      int irregular[2];                   // some row of the plane is not a full lattice-ordered row
    };
    static_assert(NN3 * STG * 8 >= 2 * ZS5 * CL3 * 8, "staging buffer must cover the negative base shift");
    static_assert(27 * CL3 <= NN3 * STG, "w*g scratch must fit in the staging buffer");
    static_assert(NN3 * STG <= TABL5, "staging buffer 1 must fit in a layer slot");
    static_assert(sizeof(Lds5<true>) <= 81920, "two workgroups per CU");

    // =====================================================================================
    template <int NCOL /* 3 blocked, 4 interleaved */, bool CLK = false /* profiling only */,
              bool RES = false /* also writes the displacement rows of the residual (res_pde) */>
    __global__ __launch_bounds__(NT3, 4) void k_cart_uu5(DevView v, CartView cv, const MatScal *__restrict__ Sp, double *__restrict__ vals,
                                                         int zc /* node planes per chunk */, unsigned long long *__restrict__ dbg,
                                                         double *__restrict__ res_pde)
    {
      // Per-launch scalars in device memory (see pfm_internal.h), read through the CONSTANT address space: only
      // constant-space loads stay scalar (s_load) inside a loop that also stores to global memory.
      const auto &S = *(const __attribute__((address_space(4))) MatScal *)Sp;
      long long tclk = 0;
      auto stamp = [&](int phase) __attribute__((always_inline)) {
        if constexpr (CLK)
          {
            const long long now = clock64();
            if (threadIdx.x == 0 && phase >= 0)
              dbg[(size_t)blockIdx.x * 8 + phase] += (unsigned long long)(now - tclk); // one slot per workgroup
            tclk = now;
          }
      };
      __shared__ Lds5<RES> s;

      const int t = threadIdx.x;
      const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1;
      const int ntx = (OWX + T3X - 1) / T3X, nty = (OWY + T3Y - 1) / T3Y;
      const int bid = xcd_tile_index();
      if (bid >= ntx * nty * ((cv.o1[2] - cv.o0[2] + zc) / zc))
        return; // padding of the XCD-aware grid
      const int tix = bid % ntx, tiy = (bid / ntx) % nty, chunk = bid / (ntx * nty);
      const int i0 = cv.o0[0] + tix * T3X, j0 = cv.o0[1] + tiy * T3Y;
      const int kA = cv.o0[2] + chunk * zc;
      const int kB = min(kA + zc, cv.o1[2] + 1); // node planes [kA, kB)
      const bool lin = !S.monolithic;           // one combined old field is interpolated (cell_wg_plane_lin)
      const int wave = __builtin_amdgcn_readfirstlane(t >> 6); // wave-uniform: scalar branches between the slot sets

      // ---- loaders (waves 0 and 1).  Lane <-> dword dw = 64 wave + lane of a plane's 60 doubles (halo node dw / 2, half
      // dw & 1); lanes of wave 0 below 32 also fetch the neighbour mask of row `lane`, the lanes of wave 1 the two
      // halves of the row offsets.  Node ids: arithmetic where the owned box is numbered lexicographically, else the
      // lattice table, read ONE STEP AHEAD of the transfer that needs it (idh: halo node, idr: row) so that no transfer
      // waits for a dependent load.
      struct Pos5
      {
        int gi, gj;   // halo node of this lane's dword
        int ri, rj;   // row of this lane
        bool inside, rowlane;
      };
      auto lane_pos = [&](int tl) __attribute__((always_inline)) {
        Pos5 p;
        const int dw = tl; // waves 0, 1: t = 64 wave + lane
        const int hn = dw >> 1;
        p.inside = dw < 2 * NHP5;
        p.gi = i0 - 1 + hn % H3X;
        p.gj = j0 - 1 + hn / H3X;
        const int nl = (tl < 64) ? tl : ((tl - 64) >> 1);
        p.rowlane = (tl < NN3) || (tl >= 64 && tl < 128);
        p.ri = i0 + nl % T3X;
        p.rj = j0 + nl / T3X;
        return p;
      };
      auto node_ok = [&](const Pos5 &p, int kz) __attribute__((always_inline)) {
        return p.inside && p.gi >= 0 && p.gi < cv.NX && p.gj >= 0 && p.gj < cv.NY && kz >= 0 && kz < cv.NZ;
      };
      auto row_ok = [&](const Pos5 &p, int kr) __attribute__((always_inline)) {
        return p.rowlane && p.ri <= cv.o1[0] && p.rj <= cv.o1[1] && kr >= cv.o0[2] && kr < kB;
      };
      auto needs_table = [&](int gi, int gj, int kk) __attribute__((always_inline)) {
        return !(cv.owned_lex && gi >= cv.o0[0] && gi <= cv.o1[0] && gj >= cv.o0[1] && gj <= cv.o1[1] && kk >= cv.o0[2] && kk <= cv.o1[2]);
      };
      auto lex_id = [&](int gi, int gj, int kk) __attribute__((always_inline)) {
        return (gi - cv.o0[0]) + (cv.o1[0] - cv.o0[0] + 1) * ((gj - cv.o0[1]) + (cv.o1[1] - cv.o0[1] + 1) * (kk - cv.o0[2]));
      };
      // stage 1: table ids of halo plane kz and of the rows of plane kr (unconditional loads at clamped indices)
      auto id_request = [&](int tl, int kz, int kr, int &idh, int &idr) __attribute__((always_inline)) {
        const Pos5 p = lane_pos(tl);
        const bool okh = node_ok(p, kz) && needs_table(p.gi, p.gj, kz);
        const bool okr = row_ok(p, kr) && needs_table(p.ri, p.rj, kr);
        idh = cv.local_of_box[okh ? p.gi + (long long)cv.NX * (p.gj + (long long)cv.NY * kz) : 0];
        idr = cv.local_of_box[okr ? p.ri + (long long)cv.NX * (p.rj + (long long)cv.NY * kr) : 0];
      };
      // stage 2: the transfers of halo plane kz and of the row info of plane kr; returns the flag byte in flight
      auto plane_request = [&](int tl, int kz, int kr, int idh, int idr, bool do_rows) __attribute__((always_inline)) -> unsigned {
        const Pos5 p = lane_pos(tl);
        const int dw = tl, w = tl >> 6;
        unsigned pf = 0u;
        const bool okh = node_ok(p, kz);
        const unsigned n = okh ? (unsigned)(needs_table(p.gi, p.gj, kz) ? idh : lex_id(p.gi, p.gj, kz)) : 0u;
        const unsigned boff = 8u * n + 4u * (dw & 1);
        uint32_t *dpo = reinterpret_cast<uint32_t *>(&s.po[kz & 3][0]), *dpoo = reinterpret_cast<uint32_t *>(&s.poo[kz & 3][0]);
        if (okh)
          {
            dma5_b32(v.phi_old, boff, dpo + 64 * w);
            dma5_b32(v.phi_oldold, boff, dpoo + 64 * w);
            if constexpr (RES)
              {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                  dma5_b32(v.u[c], boff, reinterpret_cast<uint32_t *>(&s.u[RES ? c : 0][kz & 3][0]) + 64 * w);
              }
            if ((dw & 1) == 0)
              pf = 0x80u | v.node_flags[n];
          }
        else if (p.inside)
          {
            dpo[dw] = 0u;
            dpoo[dw] = 0u;
            if constexpr (RES)
              {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                  reinterpret_cast<uint32_t *>(&s.u[RES ? c : 0][kz & 3][0])[dw] = 0u;
              }
          }
        if (!do_rows)
          return pf;
        const bool okr = row_ok(p, kr);
        const unsigned nr = okr ? (unsigned)(needs_table(p.ri, p.rj, kr) ? idr : lex_id(p.ri, p.rj, kr)) : 0u;
        if (w == 0)
          {
            if (okr)
              dma5_b32(cv.nbr_mask, 4u * nr, &s.mask[kr & 1][0]);
            else if (p.rowlane)
              s.mask[kr & 1][tl] = 0u;
          }
        else
          {
            uint32_t *dst = reinterpret_cast<uint32_t *>(&s.rowoff[kr & 1][0]);
            if (okr)
              dma5_b32(v.nadj_ptr, 8u * nr + 4u * (tl & 1), dst);
            else if (p.rowlane)
              dst[tl - 64] = 0xffffffffu; // offset -1: not an owned row of this chunk
          }
        return pf;
      };
      // after the transfers have landed (s_waitcnt vmcnt(0) by the loaders): flags, the combined old field, tile summaries
      auto plane_finish = [&](int tl, int kz, int kr, unsigned pf, bool do_rows) __attribute__((always_inline)) {
        const int dw = tl, hn = dw >> 1;
        if ((dw & 1) == 0 && dw < 2 * NHP5)
          {
            s.flag[kz & 3][hn] = (unsigned char)pf;
            if (lin && pf)
              {
                const double a = s.po[kz & 3][hn], b = s.poo[kz & 3][hn];
                s.po[kz & 3][hn] = S.use_old ? a : b + S.tfac * (a - b);
              }
          }
        if (tl == 0)
          s.anyflag[kz & 3] = 0; // raised behind the next barrier by the lanes that see a displacement flag
        if (do_rows && tl < 64)
          {
            const unsigned m = (tl < NN3) ? s.mask[kr & 1][tl] : 0x7ffffffu;
            const unsigned long long irr = __ballot(m != 0x7ffffffu); // not a full lattice-ordered row of an owned node
            if (tl == 0)
              s.irregular[kr & 1] = irr != 0;
          }
      };

      // ---- one cell layer L (cells between node planes L and L + 1): w*g at the q-points, then the 63 moment tables
      auto cell_layer = [&](int L) __attribute__((always_inline)) {
        const int pl = L & 3, pu = (L + 1) & 3;
        int tq = t;
        asm volatile("" : "+v"(tq));
        // (a) w*g at the quadrature points: thread <-> (cell, line (q_y, q_z)), 3 q-points each -> LDS [q][cell]
        if (tq < 9 * CL3)
          {
            const int cs = tq % CL3, ln = tq / CL3, qy = ln % 3, qz = ln / 3;
            const int cy = cs / C3X, cx = cs % C3X;
            const int h00 = cx + H3X * cy;
            const bool valid = (s.flag[pl][h00] & 0x80u) && (s.flag[pu][h00 + 1 + H3X] & 0x80u);
            double wg[3] = {0.0, 0.0, 0.0};
            if (valid)
              {
                const double gz = fma((double)(qz - 1), 0.5 * 0.7745966692414834, 0.5), gy = fma((double)(qy - 1), 0.5 * 0.7745966692414834, 0.5);
                const double nz0 = 1.0 - gz, nz1 = gz, ny0 = 1.0 - gy, ny1 = gy;
                const double wy = (qy == 1) ? 8.0 / 18.0 : 5.0 / 18.0, wz = (qz == 1) ? 8.0 / 18.0 : 5.0 / 18.0;
                double po[8];
#pragma unroll
                for (int b = 0; b < 8; ++b)
                  po[b] = s.po[(b >> 2) ? pu : pl][h00 + (b & 1) + H3X * ((b >> 1) & 1)];
                double a[4];
#pragma unroll
                for (int vtx = 0; vtx < 4; ++vtx)
                  a[vtx] = nz0 * po[vtx] + nz1 * po[vtx + 4];
                const double a0 = ny0 * a[0] + ny1 * a[2], a1 = ny0 * a[1] + ny1 * a[3];
                double b0 = 0.0, b1 = 0.0;
                if (!lin)
                  {
                    double poo[8], bb[4];
#pragma unroll
                    for (int b = 0; b < 8; ++b)
                      poo[b] = s.poo[(b >> 2) ? pu : pl][h00 + (b & 1) + H3X * ((b >> 1) & 1)];
#pragma unroll
                    for (int vtx = 0; vtx < 4; ++vtx)
                      bb[vtx] = nz0 * poo[vtx] + nz1 * poo[vtx + 4];
                    b0 = ny0 * bb[0] + ny1 * bb[2];
                    b1 = ny0 * bb[1] + ny1 * bb[3];
                  }
#pragma unroll
                for (int qx = 0; qx < 3; ++qx)
                  {
                    double pfx = c_g1.n[0][qx] * a0 + c_g1.n[1][qx] * a1;
                    if (lin)
                      {
                        if (!S.use_old)
                          pfx = fmin(fmax(pfx, 0.0), 1.0);
                      }
                    else
                      {
                        double pfo = pfx, pfoo = c_g1.n[0][qx] * b0 + c_g1.n[1][qx] * b1;
                        if (S.monolithic)
                          {
                            pfo = fmax(0.0, pfo);
                            pfoo = fmax(0.0, pfoo);
                          }
                        pfx = pfoo + S.tfac * (pfo - pfoo);
                        if (pfx <= 0.0)
                          pfx = 0.0;
                        if (pfx >= 1.0)
                          pfx = 1.0;
                        if (S.use_old)
                          pfx = pfo;
                      }
                    const double g = (1 - S.kappa) * pfx * pfx + S.kappa;
                    wg[qx] = S.vol * (c_g1.w[qx] * wy * wz) * g;
                  }
              }
#pragma unroll
            for (int qx = 0; qx < 3; ++qx)
              s.stage[(ln * 3 + qx) * CL3 + cs] = wg[qx];
          }
        lds_barrier();
        asm volatile("" : "+v"(tq));
        // (b) moment tables: thread <-> (cell, task), 9 tasks per cell: A^x, A^y, A^z, and the two halves (al = 0, 1) of
        // T^xy, T^xz, T^yz: 405 threads, ~65 flops each, one pass
        if (tq < 9 * CL3)
          {
            const int cs = tq % CL3, task = tq / CL3;
            double *out = s.tab[L & 1] + cs;
            const double *wq = s.stage + cs;
            if (task < 3)
              {
                const int c = task;
                const int sc = (c == 0) ? 1 : (c == 1) ? 3 : 9;
                const int si = (c == 0) ? 3 : 1;
                const int sj = (c == 2) ? 3 : 9;
                const int nB = (c == 0) ? 0 : (c == 1) ? 3 : 54, nP = (c == 2) ? 3 : 1, nQ = (c == 2) ? 1 : ZS5;
                double r27[3][3][3];
#pragma unroll
                for (int qj = 0; qj < 3; ++qj)
#pragma unroll
                  for (int qi = 0; qi < 3; ++qi)
                    {
                      const int q0 = qi * si + qj * sj;
                      r27[qj][qi][0] = wq[q0 * CL3];
                      r27[qj][qi][1] = wq[(q0 + sc) * CL3];
                      r27[qj][qi][2] = wq[(q0 + 2 * sc) * CL3];
                    }
                __builtin_amdgcn_sched_barrier(0);
                double s9[3][3]; // [qj][qi], (i,j) = other axes ascending
#pragma unroll
                for (int qj = 0; qj < 3; ++qj)
#pragma unroll
                  for (int qi = 0; qi < 3; ++qi)
                    s9[qj][qi] = (r27[qj][qi][0] + r27[qj][qi][1]) + r27[qj][qi][2];
#pragma unroll
                for (int gi = 0; gi < 3; ++gi)
                  {
                    double tq3[3];
#pragma unroll
                    for (int qj = 0; qj < 3; ++qj)
                      tq3[qj] = s9[qj][0] * c_g1.m[gi][0] + s9[qj][1] * c_g1.m[gi][1] + s9[qj][2] * c_g1.m[gi][2];
#pragma unroll
                    for (int gj = 0; gj < 3; ++gj)
                      {
                        const double val = tq3[0] * c_g1.m[gj][0] + tq3[1] * c_g1.m[gj][1] + tq3[2] * c_g1.m[gj][2];
                        out[(nB + nP * gi + nQ * gj) * CL3] = val; // numA5(c, gi, gj)
                      }
                  }
              }
            else
              {
                const int p = (task - 3) >> 1, al = (task - 3) & 1; // pair (lo,hi): 0 = (x,y), 1 = (x,z), 2 = (y,z)
                const int slo = (p == 2) ? 3 : 1;
                const int shi = (p == 0) ? 3 : 9;
                const int se = (p == 0) ? 9 : (p == 1) ? 3 : 1;
                const int nB = ((p == 0) ? 6 : (p == 1) ? 10 : 16) + ((p == 0) ? 2 : 3) * al, nQ = (p == 0) ? 1 : ZS5, nR = (p == 0) ? ZS5 : 1;
                const double na0 = al ? c_g1.n[1][0] : c_g1.n[0][0], na1 = al ? c_g1.n[1][1] : c_g1.n[0][1],
                             na2 = al ? c_g1.n[1][2] : c_g1.n[0][2];
                double r27[3][3][3];
#pragma unroll
                for (int qe = 0; qe < 3; ++qe)
#pragma unroll
                  for (int qh = 0; qh < 3; ++qh)
                    {
                      const int q0 = qh * shi + qe * se;
                      r27[qe][qh][0] = wq[q0 * CL3];
                      r27[qe][qh][1] = wq[(q0 + slo) * CL3];
                      r27[qe][qh][2] = wq[(q0 + 2 * slo) * CL3];
                    }
                __builtin_amdgcn_sched_barrier(0);
                double t1[3][3]; // [q_e][q_hi]
#pragma unroll
                for (int qe = 0; qe < 3; ++qe)
#pragma unroll
                  for (int qh = 0; qh < 3; ++qh)
                    t1[qe][qh] = (r27[qe][qh][0] * na0 + r27[qe][qh][1] * na1) + r27[qe][qh][2] * na2;
#pragma unroll
                for (int be = 0; be < 2; ++be)
                  {
                    double t2[3];
#pragma unroll
                    for (int qe = 0; qe < 3; ++qe)
                      t2[qe] = t1[qe][0] * c_g1.n[be][0] + t1[qe][1] * c_g1.n[be][1] + t1[qe][2] * c_g1.n[be][2];
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                      {
                        const double val = t2[0] * c_g1.m[g][0] + t2[1] * c_g1.m[g][1] + t2[2] * c_g1.m[g][2];
                        out[(nB + nQ * be + nR * g) * CL3] = val; // numT5(p, al, be, g)
                      }
                  }
              }
          }
        lds_barrier();
      };

      // copy-out of one staged row component by waves 2..7: thread <-> (node group g of 4, element el) with el fixed --
      // nodes g, g + 4, ..., g + 28: no division per position, one row-offset read and one value read per node, all of
      // them in flight before the first store
      auto copy_out = [&](int c, int k, bool masked, const double *__restrict__ stage) __attribute__((always_inline)) {
        const int par = k & 1;
        int tq = t - 128;
        asm volatile("" : "+v"(tq));
        if (tq < 0)
          return;
        if ((NCOL == 3) && s.irregular[par] == 0 && !masked)
          {
            constexpr int NG = 4, NIT = NN3 / NG, NB = 4; // batches of NB nodes: the 36 cached table values stay live
            const int g = tq / STG, el = tq - g * STG;
            if (g < NG)
              {
#pragma unroll
                for (int i0b = 0; i0b < NIT; i0b += NB)
                  {
                    long long rb[NB];
                    double val[NB];
#pragma unroll
                    for (int i = 0; i < NB; ++i)
                      {
                        const int nl = g + NG * (i0b + i);
                        rb[i] = s.rowoff[par][nl];
                        val[i] = stage[nl * STG + el];
                      }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < NB; ++i)
                      vals[(long long)(NCOL * NCOL) * rb[i] + (c * STG + el)] = val[i];
                    __builtin_amdgcn_sched_barrier(0);
                  }
              }
          }
        else
          {
            // rows at the faces of the box / partial tiles / next to ghost columns / near a constrained node: thread <->
            // (row, lattice offset o, column component); the CSR slot of offset o is its rank among the offsets that
            // exist, or the row's permutation of that rank.  Constraints (cracks.cc:2440-2463): a constrained row keeps
            // its diagonal only, a constrained column is eliminated.
            constexpr int rowlen = 27 * NCOL;
            for (int f = tq; f < NN3 * rowlen; f += NT3 - 128)
              {
                const int nl = f / rowlen, e = f - nl * rowlen;
                const int o = e / NCOL, d = e - o * NCOL;
                const long long off = s.rowoff[par][nl];
                const unsigned mask = s.mask[par][nl];
                if (off < 0 || !((mask >> o) & 1u))
                  continue;
                int sl = __popc(mask & ((1u << o) - 1u));
                const int deg = __popc(mask & 0x7ffffffu);
                if (mask >> 31) // the row is not in lattice order (ghost columns behind the owned ones, bound pattern)
                  sl = cv.row_perm[off + sl];
                double val = (d < 3) ? stage[nl * STG + o * 3 + d] : 0.0;
                if (masked && d < 3)
                  {
                    const int oz = o / 9, oy = (o - 9 * oz) / 3, ox = o - 9 * oz - 3 * oy;
                    const int hc = (nl % T3X + 1) + H3X * (nl / T3X + 1);
                    const unsigned rf = s.flag[k & 3][hc], cf = s.flag[(k + oz - 1) & 3][hc + (ox - 1) + H3X * (oy - 1)];
                    const bool rcon = (rf >> c) & 1u;
                    if (rcon || ((cf >> d) & 1u))
                      val = (rcon && o == 13 && d == c) ? val : 0.0;
                  }
                vals[(long long)(NCOL * NCOL) * off + (long long)c * NCOL * deg + sl * NCOL + d] = val;
              }
          }
      };

      // ---- start-up of the chunk: planes kA - 1, kA, kA + 1, rows of plane kA, cell layer kA - 1
      stamp(-1);
      int idh_next = 0, idr_next = 0; // stage-1 results in flight: ids of halo plane k + 3 / rows of plane k + 2 at the top of step k
      unsigned pf_next = 0u;          // flag byte of plane k + 2 in flight during step k
      if (wave < 2)
        {
          int h0, h1, h2, r0, r1, r2;
          id_request(t, kA - 1, kA - 2, h0, r0);
          id_request(t, kA, kA - 1, h1, r1);
          id_request(t, kA + 1, kA, h2, r2);
          id_request(t, kA + 2, kA + 1, idh_next, idr_next);
          const unsigned f0 = plane_request(t, kA - 1, kA - 2, h0, r0, false); // rows of planes kA - 2, kA - 1: never read
          const unsigned f1 = plane_request(t, kA, kA - 1, h1, r1, false);
          const unsigned f2 = plane_request(t, kA + 1, kA, h2, r2, true);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          plane_finish(t, kA - 1, kA - 2, f0, false);
          plane_finish(t, kA, kA - 1, f1, false);
          plane_finish(t, kA + 1, kA, f2, true);
        }
      __syncthreads();
      // flags of a plane -> "some displacement flag in the halo": any lane of the loaders may raise it
      if (t < NHP5)
        {
#pragma unroll
          for (int q = -1; q <= 1; ++q)
            if (s.flag[(kA + q) & 3][t] & 7u)
              s.anyflag[(kA + q) & 3] = 1;
        }
      stamp(0);
      cell_layer(kA - 1);
      stamp(1);

#pragma unroll 1
      for (int k = kA; k < kB; ++k)
        {
          int tl = t; // opaque per step: nothing derived from the thread index is kept (and spilled) across the march
          asm volatile("" : "+v"(tl));
          // plane k + 2 and the rows of plane k + 1: requested now (ahead of every store of this step), landed and
          // finished at the end of the step, first read in step k + 1
          if (wave < 2)
            {
              pf_next = plane_request(tl, k + 2, k + 1, idh_next, idr_next, true);
              id_request(tl, k + 3, k + 2, idh_next, idr_next);
            }
          stamp(0);
          cell_layer(k); // ends with a barrier: tables of layer k complete, w*g scratch free
          stamp(2);

          asm volatile("" : "+v"(tl));
          // ---- node phase
          const int lane = tl & 63;
          const bool upper = lane >= 32;
          const int nl_lane = lane & 31;
          const int ti = nl_lane % T3X, tj = nl_lane / T3X;
          const int hcp = (ti + 1) + H3X * (tj + 1); // halo index of the node within a plane
          const unsigned upper_sign = upper ? 0x80000000u : 0u;
          const int par = k & 1;
          (void)par;
          UuCoef5 K; // uniform: scalar loads, scalar registers
#pragma unroll
          for (int c = 0; c < 3; ++c)
            {
#pragma unroll
              for (int kk = 0; kk < 3; ++kk)
                K.cA[c][kk] = S.cA[c][kk];
              K.cTl[c] = S.cTl[c];
              K.cTm[c] = S.cTm[c];
            }
          const bool masked = (s.anyflag[(k - 1) & 3] | s.anyflag[k & 3] | s.anyflag[(k + 1) & 3]) != 0;
          // lower half: layer k - 1, cell (ti + 1, tj + 1) is the one whose (1,1,1) vertex is the node;
          // upper half: layer k, same (x,y) cell, z-digit shifts
          Bases5 B;
          {
            const int cellb = (tj + 1) * C3X + (ti + 1);
            const double *lo_b = s.tab[(k - 1) & 1] + cellb, *up_b = s.tab[k & 1] + cellb;
#pragma unroll
            for (int d = -2; d <= 1; ++d)
              B.b[d + 2] = upper ? up_b + d * ZS5 * CL3 : lo_b;
          }
          // RES: pressure part of the displacement residual, (alpha_B-1) p sum_q pfx^2 dN_a/dx_c JxW, from the A^c tables
          // of the 8 cells around the node: sum_q w g n_ai n_aj is the sum of the four moments A^c[a_i + b_i][a_j + b_j]
          // (n_0 + n_1 = 1), and vol w pfx^2 = (w g - kappa vol w) / (1 - kappa)
          if constexpr (RES)
            {
              if (wave == 0)
                {
                  double pres[3] = {0.0, 0.0, 0.0};
                  const double kv4 = S.kappa * S.vol * 0.25;
                  static_for<4>([&](auto Vv) __attribute__((always_inline)) {
                    constexpr Vis5 vi = visit5(0, decltype(Vv)::value);
                    constexpr int co = vi.ey * C3X + vi.ex;
                    constexpr int aL[3] = {-vi.ex, -vi.ey, 1}, aU[3] = {-vi.ex, -vi.ey, 0};
                    static_for<3>([&](auto Kk) __attribute__((always_inline)) {
                      constexpr int kd = decltype(Kk)::value;
                      constexpr int i = (kd == 0) ? 1 : 0, j = (kd == 2) ? 1 : 2;
                      double m00, m10, m01, m11;
                      tab_read<numA5(kd, aL[i], aL[j]), numA5(kd, aU[i], aU[j]), co>(B, m00);
                      tab_read<numA5(kd, aL[i] + 1, aL[j]), numA5(kd, aU[i] + 1, aU[j]), co>(B, m10);
                      tab_read<numA5(kd, aL[i], aL[j] + 1), numA5(kd, aU[i], aU[j] + 1), co>(B, m01);
                      tab_read<numA5(kd, aL[i] + 1, aL[j] + 1), numA5(kd, aU[i] + 1, aU[j] + 1), co>(B, m11);
                      const double s4 = (m00 + m10) + (m01 + m11);
                      const double mom = s4 - (s4 != 0.0 ? kv4 : 0.0); // absent cell: all tables are zero
                      const bool neg = (kd == 2) ? upper : (aL[kd] == 0);  // sign of dN_a/dx_k
                      pres[kd] += neg ? -mom : mom;
                    });
                  });
                  const double pc = S.aB1 * S.p / (1.0 - S.kappa);
#pragma unroll
                  for (int kd = 0; kd < 3; ++kd)
                    {
                      const double pk = add_halves5(pres[kd]) * (pc * S.ih[kd]);
                      if (!upper)
                        s.pres[RES ? kd * NN3 + nl_lane : 0] = pk; // read behind the barrier of component 0 at the earliest
                    }
                }
              __builtin_amdgcn_sched_barrier(0);
            }
          double tv[4][9];
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
          // all 36 reads of the wave's slot set issued before the first use
          PFM_PER_SET(static_for<4>([&](auto Vv) __attribute__((always_inline)) { uu5_load_visit<W, decltype(Vv)::value>(B, tv[decltype(Vv)::value]); });
                      __builtin_amdgcn_sched_barrier(0))
          // buffer 0 = the w*g scratch (free since the moment phase), buffer 1 = the slot of layer k - 1: written after
          // the barrier of component 0, which every wave passes with its table values in registers
          double *st0 = s.stage + nl_lane * STG, *st1 = s.tab[(k - 1) & 1] + nl_lane * STG;
          const int hs = upper ? 18 * 3 : 0;
          // RES: K u of row component c from the staged (unmasked) rows, by the loader waves while the others stream the
          // rows out.  Wave 0: the 27 entries with oz = -1 (lower half) / oz = +1 (upper half: the staged row shifted by
          // 54 entries, the plane above instead of the plane below: one instruction stream); wave 1, lower half: the 27
          // entries with oz = 0.  Partial sums -> LDS, added up and stored by wave 2 behind the next barrier.
          auto ku_partial = [&](int c, const double *__restrict__ stage) __attribute__((always_inline)) {
            if constexpr (RES)
              {
                if (wave < 2)
                  {
                    const double *srow = stage + nl_lane * STG + (wave == 0 ? (upper ? 54 : 0) : 27);
                    const int kp = (wave == 0) ? (upper ? k + 1 : k - 1) : k;
                    const double *up = &s.u[0][kp & 3][hcp];
                    double sum = 0.0;
                    if (wave == 0 || !upper)
                      {
#pragma unroll
                        for (int o9 = 0; o9 < 9; ++o9)
#pragma unroll
                          for (int d = 0; d < 3; ++d)
                            sum = fma(srow[o9 * 3 + d], up[(o9 % 3 - 1) + H3X * (o9 / 3 - 1) + d * 4 * NHP5], sum);
                      }
                    if (wave == 0)
                      sum = add_halves5(sum);
                    if (!upper)
                      s.kpart[RES ? ((c & 1) * 2 + wave) * NN3 + nl_lane : 0] = sum;
                  }
              }
          };
          // residual row of component c (cracks.cc:2440-2456: constrained rows get 0); threads 128..159 (wave 2: the
          // loaders never store to global memory), behind the barrier that follows ku_partial(c)
          auto residual_out = [&](int c) __attribute__((always_inline)) {
            if constexpr (RES)
              {
                const int tr = tl - 128;
                if (tr >= 0 && tr < NN3 && s.rowoff[par][tr] >= 0)
                  {
                    const double sum = (s.kpart[RES ? ((c & 1) * 2 + 0) * NN3 + tr : 0] + s.kpart[RES ? ((c & 1) * 2 + 1) * NN3 + tr : 0]) -
                                       s.pres[RES ? c * NN3 + tr : 0];
                    const int li = tr % T3X, lj = tr / T3X;
                    const int gi = i0 + li, gj = j0 + lj;
                    const int row = needs_table(gi, gj, k) ? cv.local_of_box[gi + (long long)cv.NX * (gj + (long long)cv.NY * k)] : lex_id(gi, gj, k);
                    const bool con = (s.flag[k & 3][(li + 1) + H3X * (lj + 1)] >> c) & 1u;
                    const long long di = (v.layout == PFM_LAYOUT_INTERLEAVED) ? (long long)row * 4 + c : (long long)row * 3 + c;
                    res_pde[di] = con ? 0.0 : -sum;
                  }
              }
          };
#define PFM_COMPONENT(C, ST) PFM_PER_SET((uu5_row_component<W, C>(tv, K, upper_sign, ST, ST + hs)))
          PFM_COMPONENT(0, st0)
          lds_barrier();
          stamp(3);
          copy_out(0, k, masked, s.stage);
          ku_partial(0, s.stage);
          PFM_COMPONENT(1, st1)
          lds_barrier();
          stamp(4);
          copy_out(1, k, masked, s.tab[(k - 1) & 1]);
          residual_out(0);
          ku_partial(1, s.tab[(k - 1) & 1]);
          PFM_COMPONENT(2, st0)
          lds_barrier();
          stamp(3);
          copy_out(2, k, masked, s.stage);
          residual_out(1);
          ku_partial(2, s.stage);
          // the loaders' transfers were issued a full step ago and are the only vector-memory operations of waves 0, 1
          if (wave < 2)
            {
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              plane_finish(tl, k + 2, k + 1, pf_next, true);
            }
          stamp(4);
          lds_barrier(); // staging buffers and the old layer slot are reused by the next step's cell phase
          residual_out(2);
          if (tl < NHP5 && (s.flag[(k + 2) & 3][tl] & 7u))
            s.anyflag[(k + 2) & 3] = 1; // read at the node phase of the next step, behind the barriers of its cell phase
#undef PFM_COMPONENT
#undef PFM_PER_SET
        }
    }
  } // namespace

  int launch_cart_uu5(const DevView &v, const CartView &cv, const pfm_params &p, double *vals_uu, hipStream_t s,
                      const void *d_scal, double *res_pde)
  {
    int rc = ensure_g1();
    if (rc)
      return rc;
    (void)p;
    const MatScal *S = static_cast<const MatScal *>(d_scal);
    const int OWX = cv.o1[0] - cv.o0[0] + 1, OWY = cv.o1[1] - cv.o0[1] + 1, OWZ = cv.o1[2] - cv.o0[2] + 1;
    const int ntx = (OWX + T3X - 1) / T3X, nty = (OWY + T3Y - 1) / T3Y;
    // z-chunks: one extra cell layer per chunk is evaluated at its start
    static const int zc_force = getenv("PFM_UU_ZC") ? atoi(getenv("PFM_UU_ZC")) : 0; // tuning only
    const int zc = zc_force > 0 ? zc_force : choose_zchunk((long long)ntx * nty, OWZ, 8, 32, 2);
    const int nch = (OWZ + zc - 1) / zc;
    const unsigned nb = (unsigned)(ntx * nty * nch);
    const dim3 grid(xcd_grid(nb)), block(NT3);
    const bool il = v.layout == PFM_LAYOUT_INTERLEAVED, res = res_pde != nullptr;
    if (getenv("PFM_UU_CLK") && !il) // profiling only
      {
        static unsigned long long *d_dbg = nullptr;
        const size_t nd = (size_t)xcd_grid(nb) * 8;
        if (!d_dbg && hipMalloc((void **)&d_dbg, nd * sizeof(unsigned long long)) != hipSuccess)
          return PFM_ERR_HIP;
        (void)hipMemsetAsync(d_dbg, 0, nd * sizeof(unsigned long long), s);
        if (res)
          hipLaunchKernelGGL((k_cart_uu5<3, true, true>), grid, block, 0, s, v, cv, S, vals_uu, zc, d_dbg, res_pde);
        else
          hipLaunchKernelGGL((k_cart_uu5<3, true, false>), grid, block, 0, s, v, cv, S, vals_uu, zc, d_dbg, res_pde);
        std::vector<unsigned long long> hall(nd);
        (void)hipMemcpy(hall.data(), d_dbg, nd * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long h[8] = {};
        for (size_t i = 0; i < nd; ++i)
          h[i % 8] += hall[i];
        const char *names[5] = {"start-up loads / requests", "start-up layer", "cell layer", "tables + node c0, copy c1 + node c2",
                                "copy c0 + node c1, copy c2 + finish"};
        fprintf(stderr, "[k_cart_uu5 phase clock, thread 0, cycles per workgroup (%d planes)]", zc);
        for (int i = 0; i < 5; ++i)
          fprintf(stderr, " %s=%.0f", names[i], (double)h[i] / nb);
        fprintf(stderr, "\n");
      }
    else if (il)
      {
        if (res)
          hipLaunchKernelGGL((k_cart_uu5<4, false, true>), grid, block, 0, s, v, cv, S, vals_uu, zc, nullptr, res_pde);
        else
          hipLaunchKernelGGL((k_cart_uu5<4, false, false>), grid, block, 0, s, v, cv, S, vals_uu, zc, nullptr, res_pde);
      }
    else if (res)
      hipLaunchKernelGGL((k_cart_uu5<3, false, true>), grid, block, 0, s, v, cv, S, vals_uu, zc, nullptr, res_pde);
    else
      hipLaunchKernelGGL((k_cart_uu5<3, false, false>), grid, block, 0, s, v, cv, S, vals_uu, zc, nullptr, res_pde);
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }
} // namespace pfm
