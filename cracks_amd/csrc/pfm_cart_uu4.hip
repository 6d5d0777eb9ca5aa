// pfm_cart_uu4.hip — (u,u) block, row-owner kernel, fourth generation: z-marching pull with register-cached tables.
//
// Mathematics: 63 moment tables per cell (header of pfm_cart.hip).  Lanes 0..31 <-> node n with the 4 cells below its
// plane, lanes 32..63 <-> the same node with the 4 cells above; the 8 waves are 8 z-symmetric slot sets.
//
// What changed against k_cart_uu3 (round 1), which evaluated BOTH cell layers around every node plane -- every cell twice
// -- and began every plane with an exposed ~3 us load of its nodal halo:
//
//   * a workgroup owns a column of 8 x 4 nodes and marches up in z.  Per plane it evaluates ONE new cell layer (w*g at the
//     q-points, 63 moments per cell); the layer below is still in LDS from the previous step (ring of two layer slots);
//   * the tables are stored in their NATURAL orientation (no z-mirrored copy for the upper half-wave, which made re-using a
//     layer impossible in round 1).  Both half-waves still run one instruction stream: the table numbering puts the
//     z-dependent index in the major digit (number = 22 z + j), so the entry the upper half needs is the lower half's entry
//     shifted by one of four constants (-2, -1, 0, +1 digits).  Four per-lane base addresses replace the mirrored copy;
//     the LDS offsets stay compile-time immediates.  Entries that couple z with x or y change sign between the halves:
//     one XOR on the finished sum;
//   * every wave reads the 36 table values of its slot set in one batch (plain ds_read_b64 from inline asm: the compiler
//     pairs such reads into ds_read2_b64, which moves half the bytes per LDS cycle) and keeps them in registers for all
//     three row components: 36 LDS reads per lane instead of 84.  The slot of the dead layer becomes the second staging
//     buffer, so the copy-out of component c overlaps the arithmetic of component c + 1 (one barrier per component);
//   * nodal plane k + 2 and the CSR row info of plane k + 1 are requested at the top of step k and land in LDS before the
//     step's first node barrier: no load latency on the critical path after the start-up of a chunk;
//   * cross-half sums through v_permlane32_swap (VALU) instead of ds_bpermute (LDS).
//
// The summation order of every entry is that of k_cart_uu3.
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
    constexpr int NNUM4 = 63;
    constexpr int ZS4 = 22;           // tables per z-digit
    constexpr int NHP = H3X * H3Y;    // 60 halo nodes per plane
    constexpr int TABL = NNUM4 * CL3; // doubles per layer slot (2835)

    // ---- table numbering: number = 22 z + j for the z-dependent families, 54.. for A^z
    //   A^x[g_y][g_z]       z = g_z, j = g_y                      (A^c[g_i][g_j], (i,j) = other axes ascending)
    //   A^y[g_x][g_z]       z = g_z, j = 3 + g_x
    //   T^xy[al][be][g_z]   z = g_z, j = 6 + 2 al + be
    //   T^xz[al][be][g_y]   z = be,  j = 10 + 3 al + g_y          (z = 0, 1 only)
    //   T^yz[al][be][g_x]   z = be,  j = 16 + 3 al + g_x
    //   A^z[g_x][g_y]       54 + 3 g_x + g_y
    __host__ __device__ constexpr int numA4(int c, int gi, int gj)
    {
      return c == 0 ? ZS4 * gj + gi : (c == 1 ? ZS4 * gj + 3 + gi : 54 + 3 * gi + gj);
    }
    __host__ __device__ constexpr int numT4(int p, int al, int be, int g)
    {
      return p == 0 ? ZS4 * g + 6 + 2 * al + be : (p == 1 ? ZS4 * be + 10 + 3 * al + g : ZS4 * be + 16 + 3 * al + g);
    }
    __host__ __device__ constexpr int pair4(int lo, int hi) { return lo == 0 ? (hi == 1 ? 0 : 1) : 2; }
    __host__ __device__ constexpr int sg4(int bit) { return bit ? 1 : -1; }

    // the 4 (slot, cell) visits of slot set W, in the order k_cart_uu3 summed them
    //   W0: (0,0,0)  W1: (0,0,-1)  W2: (0,+-1,0)  W3: (+-1,0,0)  W4: (0,+-1,-1)  W5: (+-1,0,-1)  W6: (+-1,+-1,0)  W7: (+-1,+-1,-1)
    struct Vis4
    {
      int ox, oy, oz, ex, ey, slot, first, last;
    };
    __host__ __device__ constexpr Vis4 visit4(int W, int v)
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
                  return Vis4{ox, oy, oz, ex, ey, sl, cnt == 0, cnt == total - 1};
                ++n;
                ++cnt;
              }
        }
      return Vis4{0, 0, 0, 0, 0, -1, 0, 0};
    }
    __host__ __device__ constexpr int nslots4(int W) { return (W < 2) ? 1 : (W < 6 ? 2 : 4); }

    // Per-lane pointers to the lane's cell (node offset (0,0)) in its layer slot, shifted by d z-digits for the upper
    // half: b[d + 2], d = -2 .. +1 (lower half: all four equal).
    struct Bases4
    {
      const double *b[4];
    };

    // One table value for both half-waves: the lower half (a_z = 1, b_z = 1 + oz) needs table LO, the upper half
    // (a_z = 0, b_z = -oz) table HI; HI - LO is a multiple of the z-digit stride by construction, so both read
    // base[d] + LO with a compile-time offset.  (Plain loads the compiler tracks: an inline-asm ds_read is "complete" for
    // the register allocator at the end of the statement, and under the register pressure of this kernel it spilled a
    // destination register before the data had landed.)
    template <int LO, int HI, int CELL_OFF>
    __device__ __forceinline__ void tab_read(const Bases4 &B, double &x)
    {
      static_assert((HI - LO) % ZS4 == 0 && (HI - LO) / ZS4 >= -2 && (HI - LO) / ZS4 <= 1, "z-digit shift out of range");
      constexpr int d = (HI - LO) / ZS4;
      x = B.b[d + 2][LO * CL3 + CELL_OFF];
    }

    // the 9 table values one visit needs for all nine (row comp, col comp) entries: A^k (k = 0..2), then per pair
    // p = (lo,hi): X_p = T^p[b_lo][a_hi][g_e], Y_p = T^p[a_lo][b_hi][g_e]
    template <int W, int V>
    __device__ __forceinline__ void uu4_load_visit(const Bases4 &B, double (&tv)[9])
    {
      constexpr Vis4 vi = visit4(W, V);
      constexpr int ax = -vi.ex, ay = -vi.ey, bx = -vi.ex + vi.ox, by = -vi.ey + vi.oy;
      constexpr int co = vi.ey * C3X + vi.ex;           // cell offset relative to the lane's (0,0) cell
      constexpr int aL[3] = {ax, ay, 1}, bL[3] = {bx, by, 1 + vi.oz}; // lower half: cells below the node plane
      constexpr int aU[3] = {ax, ay, 0}, bU[3] = {bx, by, -vi.oz};    // upper half: cells above
      static_for<3>([&](auto Kk) __attribute__((always_inline)) {
        constexpr int k = decltype(Kk)::value;
        constexpr int i = (k == 0) ? 1 : 0, j = (k == 2) ? 1 : 2;
        tab_read<numA4(k, aL[i] + bL[i], aL[j] + bL[j]), numA4(k, aU[i] + bU[i], aU[j] + bU[j]), co>(B, tv[k]);
      });
      static_for<3>([&](auto Pp) __attribute__((always_inline)) {
        constexpr int p = decltype(Pp)::value;
        constexpr int lo = (p == 2) ? 1 : 0, hi = (p == 0) ? 1 : 2, e = 3 - lo - hi;
        tab_read<numT4(p, bL[lo], aL[hi], aL[e] + bL[e]), numT4(p, bU[lo], aU[hi], aU[e] + bU[e]), co>(B, tv[3 + 2 * p]);
        tab_read<numT4(p, aL[lo], bL[hi], aL[e] + bL[e]), numT4(p, aU[lo], bU[hi], aU[e] + bU[e]), co>(B, tv[4 + 2 * p]);
      });
    }

    struct UuCoef4 // uniform constants of the node phase, read once per workgroup
    {
      double cA[3][3], cTl[3], cTm[3];
    };

    // r += entry (C, D) of one visit in the LOWER half's signs (the upper half's differ by one factor -1 for the entries
    // that couple z with x or y: applied to the finished sum)
    template <int W, int V, int C, int D>
    __device__ __forceinline__ void uu4_acc_visit(const double (&tv)[9], const UuCoef4 &K, double &r)
    {
      constexpr Vis4 vi = visit4(W, V);
      constexpr int a[3] = {-vi.ex, -vi.ey, 1}, b[3] = {-vi.ex + vi.ox, -vi.ey + vi.oy, 1 + vi.oz};
      if constexpr (C == D)
        {
#pragma unroll
          for (int k = 0; k < 3; ++k)
            r = fma((sg4(a[k]) * sg4(b[k]) > 0) ? K.cA[C][k] : -K.cA[C][k], tv[k], r);
        }
      else
        {
          constexpr int lo = C < D ? C : D, hi = C < D ? D : C, p = pair4(lo, hi);
          const double t1 = (C < D) ? tv[3 + 2 * p] : tv[4 + 2 * p];
          const double t2 = (C < D) ? tv[4 + 2 * p] : tv[3 + 2 * p];
          r = fma((sg4(a[C]) * sg4(b[D]) > 0) ? K.cTl[p] : -K.cTl[p], t1, r);
          r = fma((sg4(a[D]) * sg4(b[C]) > 0) ? K.cTm[p] : -K.cTm[p], t2, r);
        }
    }

    // x[l] + x[l ^ 32] in every lane, through the VALU (v_permlane32_swap) instead of two LDS bpermutes per double
    __device__ __forceinline__ double add_halves4(double x)
    {
      const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
      const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      return __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
    }
    __device__ __forceinline__ double flip_sign4(double x, unsigned sign_bit)
    {
      return __hiloint2double(__double2hiint(x) ^ (int)sign_bit, __double2loint(x));
    }

    // Row component C of slot set W for both half-waves, from the cached table values: 28 FMAs, the sign of the z-mixed
    // entries for the upper half, the cross-half adds of the oz = 0 slots, the constraint masks, 3 staged values per slot.
    // stage_half = the lane's staged row shifted by 18 slots for the upper half (a slot with oz = -1 completed by the lower
    // half is slot o_lo, its mirror completed by the upper half is o_lo + 18); oz = 0 slots are summed over the halves and
    // stored by both (same value, same address).
    template <int W, int C, bool MASKED>
    __device__ __forceinline__ void uu4_row_component(const double (&tv)[4][9], const UuCoef4 &K, unsigned upper_sign,
                                                      double *__restrict__ stage_row, double *__restrict__ stage_half,
                                                      unsigned row_flag, const unsigned char *__restrict__ flag_own,
                                                      const unsigned char *__restrict__ flag_half)
    {
      double r0 = 0.0, r1 = 0.0, r2 = 0.0; // the three column components of the current slot
      static_for<4>([&](auto Vv) __attribute__((always_inline)) {
        constexpr int V = decltype(Vv)::value;
        constexpr Vis4 vi = visit4(W, V);
        if constexpr (vi.first)
          r0 = r1 = r2 = 0.0;
        uu4_acc_visit<W, V, C, 0>(tv[V], K, r0);
        uu4_acc_visit<W, V, C, 1>(tv[V], K, r1);
        uu4_acc_visit<W, V, C, 2>(tv[V], K, r2);
        if constexpr (vi.last)
          {
            double v[3] = {r0, r1, r2};
#pragma unroll
            for (int D = 0; D < 3; ++D)
              if ((C == 2) != (D == 2)) // exactly one of the two components is z
                v[D] = flip_sign4(v[D], upper_sign);
            if constexpr (vi.oz == 0)
              {
                v[0] = add_halves4(v[0]);
                v[1] = add_halves4(v[1]);
                v[2] = add_halves4(v[2]);
              }
            if constexpr (MASKED)
              {
                const unsigned cf = (vi.oz == 0 ? flag_own : flag_half)[vi.ox + H3X * vi.oy];
                const bool rcon = (row_flag >> C) & 1u;
                constexpr bool centre = (vi.ox == 0 && vi.oy == 0 && vi.oz == 0);
                if (rcon || (cf & 1u))
                  v[0] = (rcon && centre && C == 0) ? v[0] : 0.0;
                if (rcon || (cf & 2u))
                  v[1] = (rcon && centre && C == 1) ? v[1] : 0.0;
                if (rcon || (cf & 4u))
                  v[2] = (rcon && centre && C == 2) ? v[2] : 0.0;
              }
            constexpr int o_lo = (vi.ox + 1) + 3 * (vi.oy + 1) + 9 * (vi.oz + 1);
            double *dst = (vi.oz == 0 ? stage_row : stage_half) + o_lo * 3;
            dst[0] = v[0];
            dst[1] = v[1];
            dst[2] = v[2];
          }
      });
    }

    struct Lds4u
    {
      double stage[NN3 * STG];  // staging buffer 0 [node][81]; w*g scratch [27][45] during the cell phase.  First in LDS:
                                // the upper half's shifted bases (down to -2 z-digits = -15840 bytes) must stay >= 0
      double tab[2][TABL];      // ring of two cell layers [number][cell]; the dead layer's slot = staging buffer 1
      double po[4][NHP], poo[4][NHP]; // nodal ring (plane & 3): combined old phase field (or phi_old), phi_oldold
      long long rowbase[2][NN3];      // plane & 1
      unsigned mask[2][NN3];
      unsigned char flag[4][NHP];     // constraint flags; bit 7: the node exists
      int anyflag[4];                 // some node of the plane's halo carries a displacement flag
      int irregular[2];               // some row of the plane is not a full lattice-ordered row
    };
    static_assert(NN3 * STG * 8 >= 2 * ZS4 * CL3 * 8, "staging buffer must cover the negative base shift");
    static_assert(27 * CL3 <= NN3 * STG, "w*g scratch must fit in the staging buffer");
    static_assert(NN3 * STG <= TABL, "staging buffer 1 must fit in a layer slot");

    // =====================================================================================
    // ABL (profiling only, PFM_UU_ABL): 1 = no global stores, 2 = no cell layer (tables stale), 3 = no node arithmetic
    template <int NCOL /* 3 blocked, 4 interleaved */, bool CLK = false /* profiling only */, int ABL = 0>
    __global__ __launch_bounds__(NT3, 4) void k_cart_uu4(DevView v, CartView cv, const MatScal *__restrict__ Sp, double *__restrict__ vals,
                                                         int zc /* node planes per chunk */, unsigned long long *__restrict__ dbg)
    {
      // Per-launch scalars in device memory (see pfm_internal.h), read through the CONSTANT address space: the buffer is
      // not written while a kernel runs, and only constant-space loads stay scalar (s_load) inside a loop that also
      // stores to global memory -- through a generic pointer the compiler falls back to per-lane global_load + vmcnt
      // waits in the middle of the cell phase.
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
      __shared__ Lds4u s;

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

      // ---- loaders: a two-stage, branch-free pipeline through registers.  Stage 1 (step k): the local node id of halo
      // node t of plane k + 3 and of row t - 64 of plane k + 2, where it needs the lattice table (ghost layers,
      // non-lexicographic numberings).  Stage 2 (step k + 1): the nodal values / row info through that id.  Stored to LDS
      // at the end of the cell phase of the step after.  Every load is unconditional at a clamped, always valid index (no
      // control flow around a load: the compiler drains vmcnt at the join of a branch whose arms define the loaded
      // value), validity is a predicate evaluated where the value is stored.
      struct HaloPos
      {
        int gi, gj;
        bool is_node, is_row;
      };
      auto halo_pos = [&](int t) __attribute__((always_inline)) {
        HaloPos hp;
        hp.is_node = t < NHP;
        hp.is_row = t >= 64 && t < 64 + NN3;
        const int nl = t - 64;
        hp.gi = hp.is_row ? i0 + nl % T3X : i0 - 1 + t % H3X;
        hp.gj = hp.is_row ? j0 + nl / T3X : j0 - 1 + t / H3X;
        return hp;
      };
      auto pos_ok = [&](const HaloPos &hp, int kz) __attribute__((always_inline)) {
        const bool node_ok = hp.is_node && hp.gi >= 0 && hp.gi < cv.NX && hp.gj >= 0 && hp.gj < cv.NY && kz >= 0 && kz < cv.NZ;
        const bool row_ok = hp.is_row && hp.gi <= cv.o1[0] && hp.gj <= cv.o1[1] && kz >= cv.o0[2] && kz < kB;
        return node_ok || row_ok;
      };
      // stage 1: halo nodes ask for plane kz, row lanes for plane kz - 1
      auto id_request = [&](int kz, int t) __attribute__((always_inline)) -> int {
        const HaloPos hp = halo_pos(t);
        const int kk = hp.is_row ? kz - 1 : kz;
        const bool ok = pos_ok(hp, kk);
        const long long bidx = ok ? hp.gi + (long long)cv.NX * (hp.gj + (long long)cv.NY * kk) : 0;
        return cv.local_of_box[bidx];
      };
      auto local_id = [&](const HaloPos &hp, int kk, int tabval) __attribute__((always_inline)) -> int {
        const bool arith = cv.owned_lex && hp.gi >= cv.o0[0] && hp.gi <= cv.o1[0] && hp.gj >= cv.o0[1] && hp.gj <= cv.o1[1] &&
                           kk >= cv.o0[2] && kk <= cv.o1[2];
        const int aid = (hp.gi - cv.o0[0]) + (cv.o1[0] - cv.o0[0] + 1) * ((hp.gj - cv.o0[1]) + (cv.o1[1] - cv.o0[1] + 1) * (kk - cv.o0[2]));
        return arith ? aid : tabval;
      };
      struct PlaneReg // a: phi_old or row offset (low), b: phi_oldold, f: flags or neighbour mask
      {
        double a, b;
        long long off;
        unsigned f;
      };
      // stage 2: halo nodes load plane kz, row lanes the rows of plane kz - 1, through the ids of stage 1
      auto plane_request = [&](int kz, PlaneReg &r, int t, int tabval) __attribute__((always_inline)) {
        const HaloPos hp = halo_pos(t);
        const int kk = hp.is_row ? kz - 1 : kz;
        const bool ok = pos_ok(hp, kk);
        const int n = ok ? local_id(hp, kk, tabval) : 0;
        const int nn = hp.is_row ? 0 : n, nr = hp.is_row ? n : 0;
        r.a = v.phi_old[nn];
        r.b = v.phi_oldold[nn];
        r.f = hp.is_row ? cv.nbr_mask[nr] : (unsigned)v.node_flags[nn];
        r.off = v.nadj_ptr[nr];
      };
      auto plane_store = [&](int kz, const PlaneReg &r, int t) __attribute__((always_inline)) {
        const HaloPos hp = halo_pos(t);
        const int kk = hp.is_row ? kz - 1 : kz;
        const bool ok = pos_ok(hp, kk);
        if (hp.is_node)
          {
            double a = r.a;
            if (lin)
              a = S.use_old ? r.a : r.b + S.tfac * (r.a - r.b);
            const unsigned f = ok ? (0x80u | r.f) : 0u;
            s.po[kz & 3][t] = ok ? a : 0.0;
            s.poo[kz & 3][t] = ok ? r.b : 0.0;
            s.flag[kz & 3][t] = (unsigned char)f;
            const unsigned long long any = __ballot((f & 7u) != 0);
            if (t == 0)
              s.anyflag[kz & 3] = any != 0; // the 60 halo nodes live in wave 0
          }
        else if (hp.is_row)
          {
            const int nl = t - 64;
            const unsigned mask = ok ? r.f : 0u;
            s.rowbase[kk & 1][nl] = ok ? (long long)NCOL * NCOL * r.off : -1;
            s.mask[kk & 1][nl] = mask;
            const unsigned long long irr = __ballot(mask != 0x7ffffffu); // not a full lattice-ordered row of an owned node
            if (nl == 0)
              s.irregular[kk & 1] = irr != 0;
          }
      };

      // ---- one cell layer L (cells between node planes L and L + 1): w*g at the q-points, then the 63 moment tables
      auto cell_layer = [&](int L) __attribute__((always_inline)) {
        const int pl = L & 3, pu = (L + 1) & 3;
        // the thread index is made opaque per phase: everything derived from it (LDS offsets of 27 reads and 6..9 writes
        // per lane) is recomputed per step instead of being hoisted out of the march and spilled
        int tq = t;
        asm volatile("" : "+v"(tq));
        // (a) w*g at the quadrature points: thread <-> (cell, line (q_y, q_z)), 3 q-points each -> LDS [q][cell].
        // Short dependency chains on 405 threads instead of 9 q-points on 135: the phase is latency, not issue, bound.
        if (tq < 9 * CL3)
          {
            const int cs = tq % CL3, ln = tq / CL3, qy = ln % 3, qz = ln / 3;
            const int cy = cs / C3X, cx = cs % C3X;
            const int h00 = cx + H3X * cy;
            const bool valid = (s.flag[pl][h00] & 0x80u) && (s.flag[pu][h00 + 1 + H3X] & 0x80u);
            double wg[3] = {0.0, 0.0, 0.0};
            if (valid)
              {
                // 1-D Gauss data of the lane's (q_y, q_z) by arithmetic, bit-identical to the table (make_g1): a per-lane
                // index into constant memory would be a vector load with an exposed L2 round trip in the middle of the phase
                const double gz = fma((double)(qz - 1), 0.5 * 0.7745966692414834, 0.5), gy = fma((double)(qy - 1), 0.5 * 0.7745966692414834, 0.5);
                const double nz0 = 1.0 - gz, nz1 = gz, ny0 = 1.0 - gy, ny1 = gy;
                const double wy = (qy == 1) ? 8.0 / 18.0 : 5.0 / 18.0, wz = (qz == 1) ? 8.0 / 18.0 : 5.0 / 18.0;
                double po[8];
#pragma unroll
                for (int b = 0; b < 8; ++b)
                  po[b] = s.po[(b >> 2) ? pu : pl][h00 + (b & 1) + H3X * ((b >> 1) & 1)];
                // the operation order of cell_wg_plane_lin / cell_wg_plane (pfm_cart_common.h), one line of it
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
                // table number = nB + nP g_i + nQ g_j (numA4), linear per family: per-lane strides, no select chains
                const int nB = (c == 0) ? 0 : (c == 1) ? 3 : 54, nP = (c == 2) ? 3 : 1, nQ = (c == 2) ? 1 : ZS4;
                // all 27 reads in flight before the first use: every s_waitcnt on the LDS is a ~130-cycle round trip, and
                // the compiler otherwise interleaves them with the arithmetic in batches of 2..4
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
                        out[(nB + nP * gi + nQ * gj) * CL3] = val; // numA4(c, gi, gj)
                      }
                  }
              }
            else
              {
                const int p = (task - 3) >> 1, al = (task - 3) & 1; // pair (lo,hi): 0 = (x,y), 1 = (x,z), 2 = (y,z)
                const int slo = (p == 2) ? 3 : 1;
                const int shi = (p == 0) ? 3 : 9;
                const int se = (p == 0) ? 9 : (p == 1) ? 3 : 1;
                const int nB = ((p == 0) ? 6 : (p == 1) ? 10 : 16) + ((p == 0) ? 2 : 3) * al, nQ = (p == 0) ? 1 : ZS4, nR = (p == 0) ? ZS4 : 1;
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
                        out[(nB + nQ * be + nR * g) * CL3] = val; // numT4(p, al, be, g)
                      }
                  }
              }
          }
        lds_barrier();
      };

      const int wave = __builtin_amdgcn_readfirstlane(t >> 6); // wave-uniform: scalar branches between the slot sets

      auto copy_out = [&](int c, int par, const double *__restrict__ stage) __attribute__((always_inline)) {
        int tq = t;
        asm volatile("" : "+v"(tq));
        if ((NCOL == 3) && s.irregular[par] == 0)
          {
            // 6 positions per thread (the last one for tq < 32 only): both LDS reads of all of them in flight before the
            // first store -- a loop would pay two dependent LDS round trips per iteration
            constexpr int NIT = (NN3 * STG + NT3 - 1) / NT3;
            long long rb[NIT];
            double val[NIT];
            int el[NIT];
#pragma unroll
            for (int i = 0; i < NIT; ++i)
              {
                const int f = min(tq + NT3 * i, NN3 * STG - 1);
                const int nl = f / STG;
                el[i] = f - nl * STG;
                rb[i] = s.rowbase[par][nl];
                val[i] = stage[f];
              }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NIT; ++i)
              if (tq + NT3 * i < NN3 * STG)
                {
                  if constexpr (ABL == 1)
                    {
                      if (val[i] == 1.2345e300)
                        vals[rb[i] + c * STG + el[i]] = val[i];
                    }
                  else
                    vals[rb[i] + c * STG + el[i]] = val[i];
                }
          }
        else
          {
            // rows at the faces of the box / partial tiles / next to ghost columns: thread <-> (row, lattice offset o,
            // column component); the CSR slot of offset o is its rank among the offsets that exist, or the row's
            // permutation of that rank
            constexpr int rowlen = 27 * NCOL;
            for (int f = tq; f < NN3 * rowlen; f += NT3)
              {
                const int nl = f / rowlen, e = f - nl * rowlen;
                const int o = e / NCOL, d = e - o * NCOL;
                const long long base = s.rowbase[par][nl];
                const unsigned mask = s.mask[par][nl];
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

      // ---- start-up of the chunk: planes kA - 1, kA, kA + 1, rows of plane kA, cell layer kA - 1
      stamp(-1);
      int id_next; // stage-1 result in flight: ids of plane k + 3 / rows of plane k + 2 at the top of step k
      {
        // planes kA - 1, kA, kA + 1 and the rows of plane kA (the row lanes of a request for plane kz serve plane kz - 1)
        PlaneReg p0, p1, p2;
        const int i0r = id_request(kA - 1, t), i1r = id_request(kA, t), i2r = id_request(kA + 1, t);
        id_next = id_request(kA + 2, t);
        plane_request(kA - 1, p0, t, i0r);
        plane_request(kA, p1, t, i1r);
        plane_request(kA + 1, p2, t, i2r);
        plane_store(kA - 1, p0, t); // row lanes: rows of plane kA - 2, overwritten below
        plane_store(kA, p1, t);     // rows of plane kA - 1 (never read)
        plane_store(kA + 1, p2, t); // rows of plane kA
      }
      __syncthreads();
      stamp(0);
      cell_layer(kA - 1);
      stamp(1);

#pragma unroll 1
      for (int k = kA; k < kB; ++k)
        {
          // plane k + 2 and the rows of plane k + 1: requested now, stored before this step's first node barrier
          PlaneReg pn;
          int tl = t; // opaque per step: nothing derived from the thread index is kept (and spilled) across the march
          asm volatile("" : "+v"(tl));
          plane_request(k + 2, pn, tl, id_next); // plane k + 2, rows of plane k + 1: ids were requested one step ago
          id_next = id_request(k + 3, tl);
          stamp(0);
          if constexpr (ABL != 2)
            cell_layer(k); // ends with a barrier: tables of layer k complete, w*g scratch free
          stamp(2);

          asm volatile("" : "+v"(tl));
          // The requests of the step's top have had the cell phase to land (a global load takes ~3 us while the chip
          // streams 2.8 TB/s of stores: part of that wait is still exposed here; consuming later would keep 8 more
          // registers live through the node phase, which is at the 128-register limit).  No store of this step has been
          // issued yet, so the wait is for the loads alone.  Ring slots (k + 2) & 3 = (k - 2) & 3 and rows (k + 1) & 1 are
          // dead since the previous step and are first read in the next one.
          plane_store(k + 2, pn, tl);

          // ---- node phase
          const int lane = tl & 63;
          const bool upper = lane >= 32;
          const int nl_lane = lane & 31;
          const int ti = nl_lane % T3X, tj = nl_lane / T3X;
          const int hcp = (ti + 1) + H3X * (tj + 1); // halo index of the node within a plane
          const unsigned upper_sign = upper ? 0x80000000u : 0u;
          const int par = k & 1;
          UuCoef4 K; // uniform: scalar loads, scalar registers
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
          const unsigned row_flag = s.flag[k & 3][hcp];
          const unsigned char *flag_own = &s.flag[k & 3][hcp];
          const unsigned char *flag_half = &s.flag[(upper ? k + 1 : k - 1) & 3][hcp];
          // lower half: layer k - 1, cell (ti + 1, tj + 1) is the one whose (1,1,1) vertex is the node;
          // upper half: layer k, same (x,y) cell, z-digit shifts
          Bases4 B;
          {
            const int cellb = (tj + 1) * C3X + (ti + 1);
            const double *lo_b = s.tab[(k - 1) & 1] + cellb, *up_b = s.tab[k & 1] + cellb;
#pragma unroll
            for (int d = -2; d <= 1; ++d)
              B.b[d + 2] = upper ? up_b + d * ZS4 * CL3 : lo_b;
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
          PFM_PER_SET(static_for<4>([&](auto Vv) __attribute__((always_inline)) { uu4_load_visit<W, decltype(Vv)::value>(B, tv[decltype(Vv)::value]); });
                      __builtin_amdgcn_sched_barrier(0))
          // buffer 0 = the w*g scratch (free since the moment phase), buffer 1 = the slot of layer k - 1: written after
          // the barrier of component 0, which every wave passes with its table values in registers
          double *st0 = s.stage + nl_lane * STG, *st1 = s.tab[(k - 1) & 1] + nl_lane * STG;
          const int hs = upper ? 18 * 3 : 0;
#define PFM_COMPONENT(C, ST)                                                                                                 \
  if constexpr (ABL == 3)                                                                                                    \
    ;                                                                                                                        \
  else if (masked)                                                                                                                \
    {                                                                                                                        \
      PFM_PER_SET((uu4_row_component<W, C, true>(tv, K, upper_sign, ST, ST + hs, row_flag, flag_own, flag_half)))            \
    }                                                                                                                        \
  else                                                                                                                       \
    {                                                                                                                        \
      PFM_PER_SET((uu4_row_component<W, C, false>(tv, K, upper_sign, ST, ST + hs, row_flag, flag_own, flag_half)))           \
    }
          PFM_COMPONENT(0, st0)
          lds_barrier();
          stamp(3);
          copy_out(0, par, s.stage);
          PFM_COMPONENT(1, st1)
          lds_barrier();
          stamp(4);
          copy_out(1, par, s.tab[(k - 1) & 1]);
          PFM_COMPONENT(2, st0)
          lds_barrier();
          stamp(3);
          copy_out(2, par, s.stage);
          stamp(4);
          lds_barrier(); // staging buffers and the old layer slot are reused by the next step's cell phase
#undef PFM_COMPONENT
#undef PFM_PER_SET
        }
    }
  } // namespace

  int launch_cart_uu4(const DevView &v, const CartView &cv, const pfm_params &p, double *vals_uu, hipStream_t s,
                      const void *d_scal)
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
    if (v.layout == PFM_LAYOUT_INTERLEAVED)
      hipLaunchKernelGGL(k_cart_uu4<4>, dim3(xcd_grid(nb)), dim3(NT3), 0, s, v, cv, S, vals_uu, zc, nullptr);
    else if (getenv("PFM_UU_CLK")) // profiling only
      {
        static unsigned long long *d_dbg = nullptr;
        const size_t nd = (size_t)xcd_grid(nb) * 8;
        if (!d_dbg && hipMalloc((void **)&d_dbg, nd * sizeof(unsigned long long)) != hipSuccess)
          return PFM_ERR_HIP;
        (void)hipMemsetAsync(d_dbg, 0, nd * sizeof(unsigned long long), s);
        hipLaunchKernelGGL((k_cart_uu4<3, true>), dim3(xcd_grid(nb)), dim3(NT3), 0, s, v, cv, S, vals_uu, zc, d_dbg);
        std::vector<unsigned long long> hall(nd);
        (void)hipMemcpy(hall.data(), d_dbg, nd * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long h[8] = {};
        for (size_t i = 0; i < nd; ++i)
          h[i % 8] += hall[i];
        const char *names[5] = {"start-up loads / requests", "start-up layer", "cell layer", "tables + node c0, copy c1 + node c2",
                                "copy c0 + node c1, copy c2"};
        fprintf(stderr, "[k_cart_uu4 phase clock, thread 0, cycles per workgroup (%d planes)]", zc);
        for (int i = 0; i < 5; ++i)
          fprintf(stderr, " %s=%.0f", names[i], (double)h[i] / nb);
        fprintf(stderr, "\n");
      }
    else if (getenv("PFM_UU_ABL")) // profiling only: ablations
      {
        const int abl = atoi(getenv("PFM_UU_ABL"));
        if (abl == 1)
          hipLaunchKernelGGL((k_cart_uu4<3, false, 1>), dim3(xcd_grid(nb)), dim3(NT3), 0, s, v, cv, S, vals_uu, zc, nullptr);
        else if (abl == 2)
          hipLaunchKernelGGL((k_cart_uu4<3, false, 2>), dim3(xcd_grid(nb)), dim3(NT3), 0, s, v, cv, S, vals_uu, zc, nullptr);
        else
          hipLaunchKernelGGL((k_cart_uu4<3, false, 3>), dim3(xcd_grid(nb)), dim3(NT3), 0, s, v, cv, S, vals_uu, zc, nullptr);
      }
    else
      hipLaunchKernelGGL(k_cart_uu4<3>, dim3(xcd_grid(nb)), dim3(NT3), 0, s, v, cv, S, vals_uu, zc, nullptr);
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }
} // namespace pfm
