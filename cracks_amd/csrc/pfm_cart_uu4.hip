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

    // Per-lane LDS byte addresses of the lane's cell (node offset (0,0)) in its layer slot, shifted by d z-digits for the
    // upper half: b[d + 2], d = -2 .. +1 (lower half: all four equal).
    struct Bases4
    {
      unsigned b[4];
    };

    // One table value for both half-waves: the lower half (a_z = 1, b_z = 1 + oz) needs table LO, the upper half
    // (a_z = 0, b_z = -oz) table HI; HI - LO is a multiple of the z-digit stride by construction.  Issued as asm: plain
    // ds_read_b64 with an immediate offset, not tracked by the compiler (uu4_wait_tables before the first use).
    template <int LO, int HI, int CELL_OFF>
    __device__ __forceinline__ void tab_read(const Bases4 &B, double &x)
    {
      static_assert((HI - LO) % ZS4 == 0 && (HI - LO) / ZS4 >= -2 && (HI - LO) / ZS4 <= 1, "z-digit shift out of range");
      constexpr int d = (HI - LO) / ZS4;
      constexpr int off = (LO * CL3 + CELL_OFF) * 8;
      static_assert(off >= 0 && off < 65536, "ds offset field");
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x) : "v"(B.b[d + 2]), "i"(off) : "memory");
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
        tab_read<numA4(k, aL[i] + bL[i], aL[j] + bL[j]), numA4(k, aU[i] + bU[i], aU[j] + bU[j]), co + 64>(B, tv[k]);
      });
      static_for<3>([&](auto Pp) __attribute__((always_inline)) {
        constexpr int p = decltype(Pp)::value;
        constexpr int lo = (p == 2) ? 1 : 0, hi = (p == 0) ? 1 : 2, e = 3 - lo - hi;
        tab_read<numT4(p, bL[lo], aL[hi], aL[e] + bL[e]), numT4(p, bU[lo], aU[hi], aU[e] + bU[e]), co + 64>(B, tv[3 + 2 * p]);
        tab_read<numT4(p, aL[lo], bL[hi], aL[e] + bL[e]), numT4(p, aU[lo], bU[hi], aU[e] + bU[e]), co + 64>(B, tv[4 + 2 * p]);
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
    template <int NCOL /* 3 blocked, 4 interleaved */, bool CLK = false /* profiling only */>
    __global__ __launch_bounds__(NT3, 4) void k_cart_uu4(DevView v, CartView cv, const MatScal *__restrict__ Sp, double *__restrict__ vals,
                                                         int zc /* node planes per chunk */, unsigned long long *__restrict__ dbg)
    {
      const MatScal &S = *Sp; // per-launch scalars in device memory (see pfm_internal.h)
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

      // ---- loaders: values travel through registers (requested early, stored to LDS later)
      struct PlaneReg
      {
        double a, b;
        unsigned f;
      };
      auto plane_request = [&](int kz, PlaneReg &r, int t) __attribute__((always_inline)) {
        r.a = 0.0, r.b = 0.0, r.f = 0u;
        if (t < NHP)
          {
            const int gi = i0 - 1 + t % H3X, gj = j0 - 1 + t / H3X;
            if (gi >= 0 && gi < cv.NX && gj >= 0 && gj < cv.NY && kz >= 0 && kz < cv.NZ)
              {
                const int n = cart_local_id(cv, gi, gj, kz);
                r.a = v.phi_old[n];
                r.b = v.phi_oldold[n];
                r.f = 0x80u | v.node_flags[n];
              }
          }
      };
      auto plane_store = [&](int kz, const PlaneReg &r, int t) __attribute__((always_inline)) {
        if (t < NHP)
          {
            double a = r.a;
            if (lin)
              a = S.use_old ? r.a : r.b + S.tfac * (r.a - r.b);
            s.po[kz & 3][t] = a;
            s.poo[kz & 3][t] = r.b;
            s.flag[kz & 3][t] = (unsigned char)r.f;
            const unsigned long long any = __ballot((r.f & 7u) != 0);
            if (t == 0)
              s.anyflag[kz & 3] = any != 0; // the 60 halo nodes live in wave 0
          }
      };
      struct RowReg
      {
        long long base;
        unsigned mask;
      };
      auto rows_request = [&](int kz, RowReg &r, int t) __attribute__((always_inline)) {
        r.base = -1, r.mask = 0u;
        if (t >= 64 && t < 64 + NN3)
          {
            const int nl = t - 64, gi = i0 + nl % T3X, gj = j0 + nl / T3X;
            if (gi <= cv.o1[0] && gj <= cv.o1[1] && kz < kB)
              {
                const int row = cart_local_id(cv, gi, gj, kz);
                r.base = (long long)NCOL * NCOL * v.nadj_ptr[row];
                r.mask = cv.nbr_mask[row];
              }
          }
      };
      auto rows_store = [&](int kz, const RowReg &r, int t) __attribute__((always_inline)) {
        if (t >= 64 && t < 64 + NN3)
          {
            const int nl = t - 64;
            s.rowbase[kz & 1][nl] = r.base;
            s.mask[kz & 1][nl] = r.mask;
            const unsigned long long irr = __ballot(r.mask != 0x7ffffffu); // not a full lattice-ordered row of an owned node
            if (nl == 0)
              s.irregular[kz & 1] = irr != 0;
          }
      };

      // ---- one cell layer L (cells between node planes L and L + 1): w*g at the q-points, then the 63 moment tables
      auto cell_layer = [&](int L) __attribute__((always_inline)) {
        const int pl = L & 3, pu = (L + 1) & 3;
        // the thread index is made opaque per phase: everything derived from it (LDS offsets of 27 reads and 9..12 writes
        // per lane) is recomputed per step instead of being hoisted out of the march and spilled
        int tq = t;
        asm volatile("" : "+v"(tq));
        if (tq < 3 * CL3) // thread <-> (cell, z-level) -> LDS [q][cell]
          {
            const int cs = tq % CL3, qz = tq / CL3;
            const int cy = cs / C3X, cx = cs % C3X;
            const int h00 = cx + H3X * cy;
            const bool valid = (s.flag[pl][h00] & 0x80u) && (s.flag[pu][h00 + 1 + H3X] & 0x80u);
            double wg[9];
            if (valid)
              {
                double po[8], poo[8];
#pragma unroll
                for (int b = 0; b < 8; ++b)
                  po[b] = s.po[(b >> 2) ? pu : pl][h00 + (b & 1) + H3X * ((b >> 1) & 1)];
                if (lin)
                  cell_wg_plane_lin(po, S, qz, wg);
                else
                  {
#pragma unroll
                    for (int b = 0; b < 8; ++b)
                      poo[b] = s.poo[(b >> 2) ? pu : pl][h00 + (b & 1) + H3X * ((b >> 1) & 1)];
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
              s.stage[(qz * 9 + q) * CL3 + cs] = wg[q];
          }
        lds_barrier();
        asm volatile("" : "+v"(tq));
        if (tq < 6 * CL3) // thread <-> (cell, {A^x, A^y, A^z, T^xy, T^xz, T^yz})
          {
            const int cs = tq % CL3, sub = tq / CL3;
            double *out = s.tab[L & 1] + cs;
            const double *wq = s.stage + cs;
            if (sub < 3)
              {
                const int c = sub;
                const int sc = (c == 0) ? 1 : (c == 1) ? 3 : 9;
                const int si = (c == 0) ? 3 : 1;
                const int sj = (c == 2) ? 3 : 9;
                // table number = nB + nP g_i + nQ g_j (numA4), linear per family: per-lane strides, no select chains
                const int nB = (c == 0) ? 0 : (c == 1) ? 3 : 54, nP = (c == 2) ? 3 : 1, nQ = (c == 2) ? 1 : ZS4;
                double s9[3][3]; // [qj][qi], (i,j) = other axes ascending
#pragma unroll
                for (int qj = 0; qj < 3; ++qj)
#pragma unroll
                  for (int qi = 0; qi < 3; ++qi)
                    {
                      const int q0 = qi * si + qj * sj;
                      s9[qj][qi] = (wq[q0 * CL3] + wq[(q0 + sc) * CL3]) + wq[(q0 + 2 * sc) * CL3];
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
                      {
                        const double val = tq[0] * c_g1.m[gj][0] + tq[1] * c_g1.m[gj][1] + tq[2] * c_g1.m[gj][2];
                        out[(nB + nP * gi + nQ * gj) * CL3] = val; // numA4(c, gi, gj)
                      }
                  }
              }
            else
              {
                const int p = sub - 3; // pair (lo,hi): 0 = (x,y), 1 = (x,z), 2 = (y,z)
                const int slo = (p == 2) ? 3 : 1;
                const int shi = (p == 0) ? 3 : 9;
                const int se = (p == 0) ? 9 : (p == 1) ? 3 : 1;
                const int nB = (p == 0) ? 6 : (p == 1) ? 10 : 16, nP = (p == 0) ? 2 : 3, nQ = (p == 0) ? 1 : ZS4, nR = (p == 0) ? ZS4 : 1;
#pragma unroll
                for (int al = 0; al < 2; ++al)
                  {
                    double t1[3][3]; // [q_e][q_hi]
#pragma unroll
                    for (int qe = 0; qe < 3; ++qe)
#pragma unroll
                      for (int qh = 0; qh < 3; ++qh)
                        {
                          const int q0 = qh * shi + qe * se;
                          t1[qe][qh] = (wq[q0 * CL3] * c_g1.n[al][0] + wq[(q0 + slo) * CL3] * c_g1.n[al][1]) +
                                       wq[(q0 + 2 * slo) * CL3] * c_g1.n[al][2];
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
                          {
                            const double val = t2[0] * c_g1.m[g][0] + t2[1] * c_g1.m[g][1] + t2[2] * c_g1.m[g][2];
                            out[(nB + nP * al + nQ * be + nR * g) * CL3] = val; // numT4(p, al, be, g)
                          }
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
#pragma unroll 2
            for (int f = tq; f < NN3 * STG; f += NT3)
              {
                const int nl = f / STG;
                vals[s.rowbase[par][nl] + c * STG + (f - nl * STG)] = stage[f];
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
      {
        PlaneReg p0, p1, p2;
        RowReg r0;
        plane_request(kA - 1, p0, t);
        plane_request(kA, p1, t);
        plane_request(kA + 1, p2, t);
        rows_request(kA, r0, t);
        plane_store(kA - 1, p0, t);
        plane_store(kA, p1, t);
        plane_store(kA + 1, p2, t);
        rows_store(kA, r0, t);
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
          RowReg rn;
          int tl = t; // opaque per step: nothing derived from the thread index is kept (and spilled) across the march
          asm volatile("" : "+v"(tl));
          plane_request(k + 2, pn, tl);
          rows_request(k + 1, rn, tl);
          stamp(0);
          cell_layer(k); // ends with a barrier: tables of layer k complete, w*g scratch free
          stamp(2);

          // the requests have had the whole cell phase to land; their ring slots ((k + 2) & 3 = (k - 2) & 3, rows
          // (k + 1) & 1) are dead since the previous step and are first read after this step's barriers
          asm volatile("" : "+v"(tl));
          plane_store(k + 2, pn, tl);
          rows_store(k + 1, rn, tl);

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
            const unsigned cellb = (unsigned)(((tj + 1) * C3X + (ti + 1) - 64) * 8);
            const unsigned lo_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)s.tab[(k - 1) & 1] + cellb;
            const unsigned up_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)s.tab[k & 1] + cellb;
#pragma unroll
            for (int d = -2; d <= 1; ++d)
              B.b[d + 2] = upper ? up_b + (unsigned)(d * ZS4 * CL3 * 8) : lo_b;
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
          // 36 reads in flight, one wait (inside the case: register copies at the join must see landed data)
          PFM_PER_SET(static_for<4>([&](auto Vv) __attribute__((always_inline)) { uu4_load_visit<W, decltype(Vv)::value>(B, tv[decltype(Vv)::value]); });
                      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0))
          // buffer 0 = the w*g scratch (free since the moment phase), buffer 1 = the slot of layer k - 1: written after
          // the barrier of component 0, which every wave passes with its table values in registers
          double *st0 = s.stage + nl_lane * STG, *st1 = s.tab[(k - 1) & 1] + nl_lane * STG;
          const int hs = upper ? 18 * 3 : 0;
#define PFM_COMPONENT(C, ST)                                                                                                 \
  if (masked)                                                                                                                \
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
    else
      hipLaunchKernelGGL(k_cart_uu4<3>, dim3(xcd_grid(nb)), dim3(NT3), 0, s, v, cv, S, vals_uu, zc, nullptr);
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }
} // namespace pfm
