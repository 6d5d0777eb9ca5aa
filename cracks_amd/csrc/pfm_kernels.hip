// pfm_kernels.hip — gfx950 kernels of the assembly hot path, "general" family.
//
// General family = any Q1 quad/hex mesh (MappingQ1 geometry per quadrature point), any
// constraint set (homogeneous lines + hanging nodes), optional 2-D stress split, optional
// per-cell Lame coefficients.  It is the reference-faithful fallback; uniform Cartesian
// meshes (all BASELINE configs) are served by the row-owner kernels in pfm_cart.hip.
//
// Work decomposition: one lane <-> (cell, test vertex a).  The lane integrates the
// (dim+1) x dpc row block of the element matrix that belongs to vertex a
// (cracks.cc:2308-2389, rows j = (a,*), all trial dofs i) and the (dim+1) residual
// entries (cracks.cc:2393-2432), then scatters through the constraints
// (cracks.cc:2439-2464).  Cells are processed in colour classes (no two cells of a class share a node, one launch per
// class): the rows of a vertex are then private to its lane and are updated with plain batched read-modify-writes;
// only the class of the cells with hanging vertices adds with hardware FP64 atomics (device-scope atomics run at
// ~3e10 /s on MI355X and were 90 % of the kernel time when every entry used them).  A 256-thread workgroup covers
// 256/2^dim cells; their nodal inputs are staged once in LDS (SoA over the cell index,
// so the 2^dim lanes of a cell read one broadcast address and neighbouring cells hit
// consecutive banks).
#include "pfm_internal.h"
#include "pfm_split.h"

#include <hip/hip_runtime.h>
#include <algorithm>

namespace pfm
{
  namespace
  {
    // ------------------------------------------------------------ reference element
    // FE_Q(1) shape functions and QGauss(3) on [0,1]^dim, x fastest (cracks.cc:2156-2160).
    struct RefTables
    {
      double N2[9][4], dN2[9][4][2], w2[9];
      double N3[27][8], dN3[27][8][3], w3[27];
    };
    __constant__ RefTables c_ref;

    RefTables make_ref_tables()
    {
      RefTables t{};
      const double gx[3] = {0.5 - 0.5 * 0.7745966692414834, 0.5, 0.5 + 0.5 * 0.7745966692414834};
      const double gw[3] = {5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0};
      for (int dim = 2; dim <= 3; ++dim)
        {
          const int nq = dim == 2 ? 9 : 27, nv = 1 << dim;
          for (int q = 0; q < nq; ++q)
            {
              const int qi[3] = {q % 3, (q / 3) % 3, q / 9};
              double w = 1.0;
              for (int d = 0; d < dim; ++d)
                w *= gw[qi[d]];
              for (int v = 0; v < nv; ++v)
                {
                  double val = 1.0, g[3] = {1.0, 1.0, 1.0};
                  for (int d = 0; d < dim; ++d)
                    {
                      const double x = gx[qi[d]];
                      const double f = ((v >> d) & 1) ? x : 1.0 - x;
                      const double df = ((v >> d) & 1) ? 1.0 : -1.0;
                      val *= f;
                      for (int e = 0; e < dim; ++e)
                        g[e] *= (e == d) ? df : f;
                    }
                  if (dim == 2)
                    {
                      t.N2[q][v] = val;
                      t.dN2[q][v][0] = g[0];
                      t.dN2[q][v][1] = g[1];
                    }
                  else
                    {
                      t.N3[q][v] = val;
                      for (int e = 0; e < 3; ++e)
                        t.dN3[q][v][e] = g[e];
                    }
                }
              if (dim == 2)
                t.w2[q] = w;
              else
                t.w3[q] = w;
            }
        }
      return t;
    }

    template <int dim>
    __device__ __forceinline__ double refN(int q, int v)
    {
      if constexpr (dim == 2)
        return c_ref.N2[q][v];
      else
        return c_ref.N3[q][v];
    }
    template <int dim>
    __device__ __forceinline__ double refdN(int q, int v, int e)
    {
      if constexpr (dim == 2)
        return c_ref.dN2[q][v][e];
      else
        return c_ref.dN3[q][v][e];
    }
    template <int dim>
    __device__ __forceinline__ double refw(int q)
    {
      if constexpr (dim == 2)
        return c_ref.w2[q];
      else
        return c_ref.w3[q];
    }

    struct Vals
    {
      double *b[4];
    };

    // ------------------------------------------------------------ CSR addressing
    // Canonical pattern = node graph (x) component coupling, so the value index of
    // (row node P, row comp c, neighbour slot s, col comp d) is arithmetic.
    template <int dim>
    __device__ __forceinline__ double *val_ptr(const DevView &v, const Vals &vals, int P, int c, int s, int d)
    {
      constexpr int nc = dim + 1;
      const long long off = v.nadj_ptr[P];
      const long long deg = v.nadj_ptr[P + 1] - off;
      if (v.layout == PFM_LAYOUT_INTERLEAVED)
        return vals.b[0] + (nc * nc * off + (long long)c * nc * deg + (long long)s * nc + d);
      if (c < dim)
        {
          if (d < dim)
            return vals.b[0] + (dim * dim * off + (long long)c * dim * deg + (long long)s * dim + d);
          return vals.b[1] + (dim * off + (long long)c * deg + s);
        }
      if (d < dim)
        return vals.b[2] + (dim * off + (long long)s * dim + d);
      return vals.b[3] + (off + s);
    }

    template <int dim>
    __device__ __forceinline__ long long dof_index(const DevView &v, int P, int c)
    {
      if (v.layout == PFM_LAYOUT_INTERLEAVED)
        return (long long)P * (dim + 1) + c;
      return c < dim ? (long long)P * dim + c : (long long)v.n_owned * dim + P;
    }

    // position of node Q in the row of node P; the rows carry the order of the bound pattern (ascending local id
    // unless pfm_pattern_bind gave another one), so the search is linear (used for hanging-node parents only)
    __device__ __forceinline__ int find_slot(const DevView &v, int P, int Q)
    {
      const long long lo = v.nadj_ptr[P], hi = v.nadj_ptr[P + 1];
      long long k = lo;
      while (k < hi && v.nadj[k] != Q)
        ++k;
      return (int)(k - lo);
    }

    // rows of a colour class belong to one cell each: plain read-modify-write; the class of the cells with hanging
    // vertices adds with global_atomic_add_f64
    template <bool ATOMIC>
    __device__ __forceinline__ void add_to(double *p, double x)
    {
      if constexpr (ATOMIC)
        unsafeAtomicAdd(p, x);
      else
        *p += x;
    }

    // value of lane (quad base + b) in all four lanes of a quad (DPP quad_perm: no LDS traffic); b is a constant after
    // unrolling
    __device__ __forceinline__ double quad_bcast(double x, int b)
    {
      int lo = __double2loint(x), hi = __double2hiint(x);
      switch (b)
        {
          case 0:
            lo = __builtin_amdgcn_update_dpp(0, lo, 0x00, 0xf, 0xf, true);
            hi = __builtin_amdgcn_update_dpp(0, hi, 0x00, 0xf, 0xf, true);
            break;
          case 1:
            lo = __builtin_amdgcn_update_dpp(0, lo, 0x55, 0xf, 0xf, true);
            hi = __builtin_amdgcn_update_dpp(0, hi, 0x55, 0xf, 0xf, true);
            break;
          case 2:
            lo = __builtin_amdgcn_update_dpp(0, lo, 0xaa, 0xf, 0xf, true);
            hi = __builtin_amdgcn_update_dpp(0, hi, 0xaa, 0xf, 0xf, true);
            break;
          default:
            lo = __builtin_amdgcn_update_dpp(0, lo, 0xff, 0xf, 0xf, true);
            hi = __builtin_amdgcn_update_dpp(0, hi, 0xff, 0xf, 0xf, true);
            break;
        }
      return __hiloint2double(hi, lo);
    }

    // ------------------------------------------------------------ the cell kernel
    // Thread <-> (cell, test vertex a): the rows of vertex a against all trial vertices of the cell.  3-D Jacobian: the 108
    // row accumulators of a hex vertex need 374 registers, so a hex has 32 lanes there, (a, part): part = 0..3 integrates
    // the trial vertices 2 part, 2 part + 1 (27 accumulators).  Rounds 1-3 split the trial vertices over two LAUNCHES
    // instead; every launch re-read the 128-byte lines of the rows for its 24-byte pieces (44 GB of fetches for 10^6
    // cells, profiles/r04) and re-evaluated the q-point states.  Residual rows: the lanes of the last part.
    // PATCH (2-D, round 4): the workgroup is an 8 x 8 block of one refinement level's lattice (DevView::patch_cells); the
    // rows of the block's regular nodes are completed in LDS -- the four vertex lanes of the cells push in four barrier-
    // separated phases (phase = vertex index: a node receives exactly one cell per phase, the order of the adds is fixed) --
    // masked and written ONCE: no colour classes, no read-modify-write of global memory, no atomics.  Same q-loop, same
    // formulas as every other instantiation.
    template <int dim, bool FULL, bool SPLIT, bool ATOMIC, bool RING = false /* DevView::cell_ring is set */, bool PATCH = false,
              bool SCR = false /* 3-D cells at hanging vertices: reduced matrix / residual into DevView::hs_* instead of the outputs */>
    __global__ __launch_bounds__(256, (dim == 3 && FULL) ? 2 : 1) void k_assemble_general(DevView v, pfm_params prm, Vals vals,
                                                              double *res_pde, double *res_tot,
                                                              int residual_only, long long class_begin, long long class_size)
    {
      static_assert(!PATCH || (dim == 2 && !ATOMIC && !RING), "the patch form exists in 2-D");
      constexpr int PRW = 81 + 3 + 3; // staged row of a node: 3 row components x 9 offsets x 3 column components, residual, placeholders
      __shared__ double s_row[PATCH ? 49 * PRW : 1];
      constexpr int nv = 1 << dim, nc = dim + 1, nq = (dim == 2 ? 9 : 27), dpc = nv * nc;
      constexpr bool Q3 = dim == 3 && FULL && !SPLIT; // the trial vertices of a hex shared out among four lanes per test vertex
      constexpr int NPART = Q3 ? 4 : 1, LPC = nv * NPART /* lanes per cell */, NBL = nv / NPART; // trial vertices [B0, B0 + NBL)
      constexpr int CPB = 256 / LPC; // cells per workgroup
      constexpr int NSLOT = Q3 ? 8 : 4; // q-points per round of the 3-D exchange buffer
      __shared__ double s_x[dim][nv][CPB];
      __shared__ double s_u[dim][nv][CPB];
      __shared__ double s_p[3][nv][CPB]; // phi, phi_old, phi_oldold
      constexpr int QST = 50; // doubles per (q-point slot, cell) of the 3-D exchange buffer: 49 used, stride free of bank conflicts for 16-byte reads
      // KRED (the class of the 3-D cells at hanging vertices): the scatter first forms C^T K C over the cell's constraint-resolved
      // nodes in LDS (below); the cell's part of the exchange buffer is the head of a region that holds its 8 x 8 x 13 matrix
      constexpr bool KRED = Q3 && !PATCH; // (round 5, second step: in the plain classes as well, see the scatter)
      constexpr int KCMP = 13;                      // Kuu 3 x 3, Kpu 3, Kpp
      constexpr int KCELL = KRED ? nv * nv * KCMP : NSLOT * QST; // doubles per cell of s_q
      __shared__ __attribute__((aligned(16))) double s_q[(dim == 3 && !SPLIT) ? (KRED ? CPB * KCELL : NSLOT * CPB * QST) : 2];
      auto sq_at = [&](int slot, int cell_in_wg) { return KRED ? cell_in_wg * KCELL + slot * QST : (slot * CPB + cell_in_wg) * QST; };
      __shared__ double s_rr[(dim == 3 && !PATCH) ? CPB : 1][nv][nc]; // residual entries of a cell at hanging vertices, by vertex
      __shared__ uint8_t s_icnt[KRED ? CPB : 1][16], s_ia[KRED ? CPB : 1][16][8];
      __shared__ double s_iw[KRED ? CPB : 1][16][8];

      const int tid = threadIdx.x;
      const int a = tid % nv, part = Q3 ? (tid / nv) % NPART : 0, cl0 = tid / LPC;
      const int B0 = part * NBL;
      const bool resid_lane = part == NPART - 1;           // scatters the residual rows of vertex a
      const bool diag_lane = (a / NBL) == part;            // holds the diagonal block K[(a,.),(a,.)]
      const int cl = cl0;
      const long long at = (long long)blockIdx.x * CPB + cl;
      bool active = at < class_size;
      long long cell = 0;
      // PATCH: what the copy-out needs of the block's 9 x 9 nodes, fetched while the cell data are on their way
      __shared__ int s_bn[PATCH ? 81 : 1], s_fl[PATCH ? 81 : 1];
      __shared__ long long s_off[PATCH ? 49 : 1];           // first neighbour entry of the row, -1: not a row of this block
      __shared__ unsigned long long s_inv[PATCH ? 49 : 1];  // 4-bit fields: lattice offset (0..8) stored in slot s of the row
      if constexpr (PATCH)
        {
          const int pc = v.patch_cells[(long long)blockIdx.x * CPB + cl];
          active = pc >= 0;
          cell = active ? pc : 0;
          for (int i = tid; i < 49 * PRW; i += 256)
            s_row[i] = 0.0;
          if (tid < 81)
            {
              const int n = v.patch_nodes[(long long)blockIdx.x * 81 + tid];
              const int hx = tid % 9, hy = tid / 9;
              const bool inner = hx >= 1 && hx <= 7 && hy >= 1 && hy <= 7;
              const bool ok = inner && n >= 0 && n < v.n_owned && v.row_patch[n];
              s_bn[tid] = n;
              s_fl[tid] = n >= 0 ? (int)v.node_flags[n] : 0;
              if (inner)
                {
                  const unsigned long long sl = ok ? v.node_slots[n] : 0ull;
                  unsigned long long iv = 0;
#pragma unroll
                  for (int o = 0; o < 9; ++o)
                    iv |= (unsigned long long)o << (4 * (int)((sl >> (4 * o)) & 15ull));
                  s_off[(hx - 1) + 7 * (hy - 1)] = ok ? (long long)v.nadj_ptr[n] : -1ll;
                  s_inv[(hx - 1) + 7 * (hy - 1)] = iv;
                }
            }
        }
      else
        cell = active ? v.color_cells[class_begin + at] : 0;
      int A = 0;
      if (active)
        {
          A = v.conn[(long long)a * v.n_cells + cell];
          if (part == 0)
            {
#pragma unroll
              for (int d = 0; d < dim; ++d)
                {
                  s_x[d][a][cl] = v.coords[(long long)d * v.n_nodes + A];
                  s_u[d][a][cl] = v.u[d][A];
                }
              s_p[0][a][cl] = v.phi[A];
              s_p[1][a][cl] = v.phi_old[A];
              s_p[2][a][cl] = v.phi_oldold[A];
            }
        }
      __syncthreads();
      if (!PATCH && !active)
        return; // (the lanes of an absent cell of a patch block stay for the barriers of the epilogue)

      double lam = prm.lambda, mu = prm.mu;
      if (v.cell_lambda)
        {
          lam = v.cell_lambda[cell];
          mu = v.cell_mu[cell];
        }
      double gamma_penal = prm.gamma_penal;
      if (prm.outer_solver == PFM_SOLVER_SIMPLE_MONOLITHIC && prm.timestep_number < 1)
        gamma_penal = 0.0; // cracks.cc:2141-2144
      const double kappa = prm.constant_k, eps = prm.alpha_eps, Gc = prm.G_c, p = prm.pressure;
      const double aB1 = prm.alpha_biot - 1.0;
      const double d_rhs = prm.decompose_stress_rhs, d_mat = prm.decompose_stress_matrix;
      const bool monolithic = prm.outer_solver == PFM_SOLVER_SIMPLE_MONOLITHIC;
      // time extrapolation factor of pf_extra, cracks.cc:2268-2269
      const double tfac = (prm.time - (prm.time - prm.old_timestep - prm.old_old_timestep)) /
                          (prm.time - prm.old_timestep - (prm.time - prm.old_timestep - prm.old_old_timestep));

      // cell->diameter(): longest vertex-to-opposite-vertex diagonal (only used when gamma != 0)
      double diam2 = 0.0;
#pragma unroll
      for (int vv = 0; vv < nv / 2; ++vv)
        {
          double s = 0.0;
#pragma unroll
          for (int d = 0; d < dim; ++d)
            {
              const double t = s_x[d][vv][cl] - s_x[d][nv - 1 - vv][cl];
              s += t * t;
            }
          diam2 = fmax(diam2, s);
        }
      const double penal_fac = gamma_penal / prm.timestep * 1.0 / diam2;

      // accumulators: rows j = (a, c)
      double R[nc];
      double Kuu[FULL ? NBL : 1][dim][dim]; // [b - B0][c][d]   trial (b,d) -> row (a,c)
      double Kpu[FULL ? NBL : 1][dim];      // [b - B0][d]      trial (b,d) -> row (a,phi)
      double Kpp[FULL ? NBL : 1];           // [b - B0]         trial (b,phi) -> row (a,phi)
#pragma unroll
      for (int c = 0; c < nc; ++c)
        R[c] = 0.0;
      if constexpr (FULL)
        {
#pragma unroll
          for (int b = 0; b < NBL; ++b)
            {
              Kpp[b] = 0.0;
#pragma unroll
              for (int c = 0; c < dim; ++c)
                {
                  Kpu[b][c] = 0.0;
#pragma unroll
                  for (int d = 0; d < dim; ++d)
                    Kuu[b][c][d] = 0.0;
                }
            }
        }
      bool ortho_ok = true;
      double p_ihx = 0.0, p_ihy = 0.0, p_det = 0.0;
      if constexpr (PATCH)
        {
          const double hx = s_x[0][1][cl] - s_x[0][0][cl], hy = s_x[1][2][cl] - s_x[1][0][cl];
          p_ihx = 1.0 / hx;
          p_ihy = 1.0 / hy;
          p_det = hx * hy;
        }

      const int nq_run = active ? nq : 0;
      if constexpr (dim == 2 && SPLIT)
        {
          // ======================= stress split (2-D): the q-point states are shared out among the four lanes of the cell
          // Everything of a q-point that does not depend on the test vertex -- MappingQ1, the Newton state, the eigen
          // split (cracks.cc:1691-1737, 1923-1970) and its linearisation (cracks.cc:1976-2109) -- is the same for the four
          // lanes of a cell and is most of the work.  Round j = 0, 1, 2: lane a evaluates q-point 4 j + a ONCE (phase A) and
          // leaves what the rows need of it, already multiplied by JxW, in NO registers; then the quad goes through the
          // round's q-points together (phase B): the values of the owner lane come by DPP, every lane adds to the rows of
          // its own vertex.  9 q-points in 12 slots (the last round has one) instead of 36 evaluations per cell.
          //
          // The linearised split is linear in the direction E_LinU, and E_LinU of trial dof (b, d) is
          // sym(e_d (x) grad N_b): the owner calls the reference function for the unit directions e0 e0, e1 e1,
          // sym(e0 e1), folds g(phi) and decompose_stress_matrix in (a 3 x 3 tangent D in Voigt order 00, 11, 01) and the
          // rows are the usual B_a^T D B_b.
          constexpr double GX0 = 0.5 - 0.5 * 0.7745966692414834, GX2 = 0.5 + 0.5 * 0.7745966692414834; // make_ref_tables
          constexpr double GW0 = 5.0 / 18.0, GW1 = 8.0 / 18.0;
          constexpr int NO_RES = 6, NO_JAC = 14, NO_INV = PATCH ? 0 : 4;
          constexpr int OR = FULL ? NO_JAC : 0, OI = OR + NO_RES, NO = OI + NO_INV;
          const int ax = a & 1, ay = a >> 1;
#pragma unroll 1
          for (int j = 0; j < (active ? 3 : 0); ++j)
            {
              int cl = cl0;
              asm volatile("" : "+v"(cl)); // the cell's nodal data are re-read from LDS per round, not pinned in registers
              double O[NO];
              {
                // ---------------- phase A: q-point qa of this lane (the idle slots of the last round repeat q-point 8)
                const int qa = min(4 * j + a, nq - 1);
                const int qy = (qa * 11) >> 5, qx = qa - 3 * qy;
                const double x1 = qx == 0 ? GX0 : (qx == 1 ? 0.5 : GX2), y1 = qy == 0 ? GX0 : (qy == 1 ? 0.5 : GX2);
                const double x0 = 1.0 - x1, y0 = 1.0 - y1;
                const double wq = (qx == 1 ? GW1 : GW0) * (qy == 1 ? GW1 : GW0);
                const double Nl[4] = {x0 * y0, x1 * y0, x0 * y1, x1 * y1};
                const double dNl[4][2] = {{-y0, -x0}, {y0, -x1}, {-y1, x0}, {y1, x1}};
                double inv[2][2], det;
                if constexpr (PATCH)
                  {
                    inv[0][0] = p_ihx;
                    inv[1][1] = p_ihy;
                    inv[0][1] = inv[1][0] = 0.0;
                    det = p_det;
                  }
                else
                  {
                    double J[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
                    for (int vv = 0; vv < 4; ++vv)
#pragma unroll
                      for (int i = 0; i < 2; ++i)
                        {
                          const double xi = s_x[i][vv][cl];
                          J[i][0] += xi * dNl[vv][0];
                          J[i][1] += xi * dNl[vv][1];
                        }
                    det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
                    const double id = 1.0 / det;
                    inv[0][0] = J[1][1] * id;
                    inv[0][1] = -J[0][1] * id;
                    inv[1][0] = -J[1][0] * id;
                    inv[1][1] = J[0][0] * id;
                  }
                const double JxW = det * wq;
                double gu[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, gpf[2] = {0.0, 0.0}, pf = 0.0, pfo = 0.0, pfoo = 0.0;
#pragma unroll
                for (int vv = 0; vv < 4; ++vv)
                  {
                    double gv[2];
                    if constexpr (PATCH)
                      {
                        gv[0] = inv[0][0] * dNl[vv][0];
                        gv[1] = inv[1][1] * dNl[vv][1];
                      }
                    else
                      {
                        gv[0] = inv[0][0] * dNl[vv][0] + inv[1][0] * dNl[vv][1];
                        gv[1] = inv[0][1] * dNl[vv][0] + inv[1][1] * dNl[vv][1];
                      }
                    const double ph = s_p[0][vv][cl];
                    pf += ph * Nl[vv];
                    pfo += s_p[1][vv][cl] * Nl[vv];
                    pfoo += s_p[2][vv][cl] * Nl[vv];
#pragma unroll
                    for (int d = 0; d < 2; ++d)
                      {
                        gpf[d] += ph * gv[d];
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                          gu[c][d] += s_u[c][vv][cl] * gv[d];
                      }
                  }
                // q-point state, cracks.cc:2248-2306
                if (monolithic)
                  {
                    pf = fmax(0.0, pf);
                    pfo = fmax(0.0, pfo);
                    pfoo = fmax(0.0, pfoo);
                  }
                const double pf_minus_old_plus = fmax(0.0, pf - pfo);
                double pfx = pfoo + tfac * (pfo - pfoo);
                if (pfx <= 0.0)
                  pfx = 0.0;
                if (pfx >= 1.0)
                  pfx = 1.0;
                if (prm.use_old_timestep_pf)
                  pfx = pfo;
                const double g = (1 - kappa) * pfx * pfx + kappa;
                double E[2][2];
                const double divu = gu[0][0] + gu[1][1];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                  for (int jj = 0; jj < 2; ++jj)
                    E[i][jj] = 0.5 * (gu[i][jj] + gu[jj][i]);
                const double trE = E[0][0] + E[1][1];
                double sp[2][2], sm[2][2];
                ortho_ok &= split_stress(E, trE, lam, mu, sp, sm);
                double spE = 0.0; // scalar_product(stress_term_plus, E)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                  for (int jj = 0; jj < 2; ++jj)
                    spE += sp[i][jj] * E[i][jj];
                if constexpr (FULL)
                  {
                    const double cA = (1 - kappa) * pf * JxW, cB = 2.0 * aB1 * p * pf * JxW;
                    double spL[3][3], smL[3][3];
                    ortho_ok &= split_tangent(E, trE, lam, mu, spL, smL);
                    const double spELk[3] = {sp[0][0], sp[1][1], sp[0][1] * 0.5 + sp[1][0] * 0.5}; // sigma+ : E_LinU of direction k
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                      {
                        const double spLE = spL[k][0] * E[0][0] + spL[k][2] * E[0][1] + spL[k][2] * E[1][0] + spL[k][1] * E[1][1];
#pragma unroll
                        for (int r = 0; r < 3; ++r)
                          O[r * 3 + k] = (g * spL[k][r] + d_mat * smL[k][r]) * JxW;
                        // row (a, phi): (1 - kappa) (sigma+_LinU : E + sigma+ : E_LinU) pf N_a - 2 (alpha - 1) p pf tr(E_LinU) N_a
                        O[9 + k] = cA * (spLE + spELk[k]) - (k < 2 ? cB : 0.0);
                      }
                    const double pen = ((pf - pfo) < 0.0) ? 0.0 : penal_fac; // shadowed variable, cracks.cc:2311-2315
                    O[12] = (pen + (1 - kappa) * spE + Gc / eps - 2.0 * aB1 * p * divu) * JxW;
                    O[13] = Gc * eps * JxW;
                  }
                // residual, cracks.cc:2393-2432: (g sigma+ + decompose_stress_rhs sigma- - (alpha - 1) p pfx^2 I) JxW for the rows
                // (a, c); the factor of N_a and G_c eps grad(phi) JxW for the row (a, phi)
                const double iso = aB1 * p * pfx * pfx;
                O[OR + 0] = (g * sp[0][0] + d_rhs * sm[0][0] - iso) * JxW;
                O[OR + 1] = (g * sp[1][1] + d_rhs * sm[1][1] - iso) * JxW;
                O[OR + 2] = (g * sp[0][1] + d_rhs * sm[0][1]) * JxW;
                O[OR + 3] = (penal_fac * pf_minus_old_plus + (1.0 - kappa) * spE * pf - Gc / eps * (1.0 - pf) - 2.0 * aB1 * p * pf * divu) * JxW;
                O[OR + 4] = Gc * eps * JxW * gpf[0];
                O[OR + 5] = Gc * eps * JxW * gpf[1];
                if constexpr (!PATCH)
                  {
                    O[OI + 0] = inv[0][0];
                    O[OI + 1] = inv[0][1];
                    O[OI + 2] = inv[1][0];
                    O[OI + 3] = inv[1][1];
                  }
              }
              // ---------------- phase B: the q-points 4 j .. 4 j + 3 of the round, one after the other, all four lanes
#pragma unroll
              for (int t = 0; t < 4; ++t)
                {
                  const int q = 4 * j + t;
                  if (q >= nq) // (uniform)
                    continue;
                  double Q[NO];
#pragma unroll
                  for (int i = 0; i < NO; ++i)
                    Q[i] = quad_bcast(O[i], t);
                  // reference gradients at q (scalar loads), the lane's own vertex by its bits
                  const double rx0 = refdN<2>(q, 2, 1), rx1 = refdN<2>(q, 3, 1), ry0 = refdN<2>(q, 1, 0), ry1 = refdN<2>(q, 3, 0);
                  const double fxa = ax ? rx1 : rx0, fya = ay ? ry1 : ry0;
                  const double Na = fxa * fya;
                  const double dax = ax ? fya : -fya, day = ay ? fxa : -fxa; // d N_a / d xi, d eta
                  double gNa[2], gN[4][2];
                  if constexpr (PATCH)
                    {
                      gNa[0] = p_ihx * dax;
                      gNa[1] = p_ihy * day;
#pragma unroll
                      for (int b = 0; b < 4; ++b)
                        {
                          gN[b][0] = p_ihx * refdN<2>(q, b, 0);
                          gN[b][1] = p_ihy * refdN<2>(q, b, 1);
                        }
                    }
                  else
                    {
                      gNa[0] = Q[OI + 0] * dax + Q[OI + 2] * day;
                      gNa[1] = Q[OI + 1] * dax + Q[OI + 3] * day;
#pragma unroll
                      for (int b = 0; b < 4; ++b)
                        {
                          gN[b][0] = Q[OI + 0] * refdN<2>(q, b, 0) + Q[OI + 2] * refdN<2>(q, b, 1);
                          gN[b][1] = Q[OI + 1] * refdN<2>(q, b, 0) + Q[OI + 3] * refdN<2>(q, b, 1);
                        }
                    }
                  if constexpr (FULL)
                    {
                      // G = B_a^T (D JxW), B_a = [gNa_x 0; 0 gNa_y; gNa_y gNa_x]
                      double G[2][3];
#pragma unroll
                      for (int k = 0; k < 3; ++k)
                        {
                          G[0][k] = gNa[0] * Q[0 * 3 + k] + gNa[1] * Q[2 * 3 + k];
                          G[1][k] = gNa[1] * Q[1 * 3 + k] + gNa[0] * Q[2 * 3 + k];
                        }
                      const double m0 = Na * Q[9], m1 = Na * Q[10], m2 = Na * Q[11];
                      const double cpp = Na * Q[12], cgx = Q[13] * gNa[0], cgy = Q[13] * gNa[1];
#pragma unroll
                      for (int b = 0; b < 4; ++b)
                        {
                          const double gbx = gN[b][0], gby = gN[b][1];
                          // trial dofs (b, d): rows (a, c) = sum_k G[c][k] B_b[k][d], B_b[.][0] = (gbx, 0, gby), B_b[.][1] = (0, gby, gbx)
#pragma unroll
                          for (int c = 0; c < 2; ++c)
                            {
                              Kuu[b][c][0] += G[c][0] * gbx + G[c][2] * gby;
                              Kuu[b][c][1] += G[c][1] * gby + G[c][2] * gbx;
                            }
                          Kpu[b][0] += m0 * gbx + m2 * gby;
                          Kpu[b][1] += m1 * gby + m2 * gbx;
                          // trial dof (b, phi): rows (a, c < dim) get exactly 0 (cracks.cc:2333-2337)
                          Kpp[b] += cpp * refN<2>(q, b) + (cgx * gbx + cgy * gby);
                        }
                    }
                  R[0] -= Q[OR + 0] * gNa[0] + Q[OR + 2] * gNa[1];
                  R[1] -= Q[OR + 2] * gNa[0] + Q[OR + 1] * gNa[1];
                  R[2] -= Na * Q[OR + 3] + (Q[OR + 4] * gNa[0] + Q[OR + 5] * gNa[1]);
                }
            }
        }
      else if constexpr (dim == 3 && !SPLIT)
        {
          // ======================= 3-D: the q-point states are shared out among the lanes of the cell (as in the 2-D split
          // form above, through LDS instead of DPP: a hex has 8 lanes, 49 numbers per q-point).  MappingQ1, the shape
          // gradients and the Newton state at a q-point are the same for the eight lanes of a cell and are 60 % of the
          // instructions of the plain loop.  A round is NSLOT q-points: lane (a, part) evaluates q-point NSLOT r + slot
          // (phase A; Jacobian: slot = a, the four parts repeat each other; residual-only: slot = a & 3, lanes 4..7 repeat
          // lanes 0..3 -- eight slots for 32 cells would need 100 KB of LDS) and one lane per slot leaves in s_q what the
          // rows need of it; then the cell's lanes go through the round's q-points together (phase B), each adding to the
          // rows of its own vertex (and, Jacobian, its own two trial vertices).  A cell's lanes sit in one wave and the LDS
          // executes a wave's instructions in order: no barrier between the phases.
          constexpr double GX0 = 0.5 - 0.5 * 0.7745966692414834, GX2 = 0.5 + 0.5 * 0.7745966692414834; // make_ref_tables
          constexpr double GW0 = 5.0 / 18.0, GW1 = 8.0 / 18.0;
          // slot layout (doubles): vertex v at 4 v: grad N_v (3), N_v;  32..: the scalars below;  stride QST
          constexpr int O_GW = 32 /* g JxW, c_pu */, O_CDIV = 34 /* c_div, c_pp */, O_CGG = 36 /* c_gg, c_pen */, O_SP = 38 /* sigma+ (6) */, O_RP = 44 /* r_p, r_phi */, O_GPF = 46 /* grad phi (3) */;
#pragma unroll 1
          for (int q = 0; q < nq_run; ++q)
            {
              int cl = cl0;
              asm volatile("" : "+v"(cl)); // the cell's nodal data are re-read from LDS per round, not pinned in registers
              if ((q & (NSLOT - 1)) == 0)
                {
                  // ---------------- phase A: q-point qa of this lane (the idle slots of the last round repeat q-point 26)
                  const int slot = Q3 ? a : (a & 3);
                  const int qa = min(q + slot, nq - 1);
                  const int qz = (qa * 57) >> 9, qr = qa - 9 * qz, qy = (qr * 11) >> 5, qx = qr - 3 * qy;
                  const double x1 = qx == 0 ? GX0 : (qx == 1 ? 0.5 : GX2), y1 = qy == 0 ? GX0 : (qy == 1 ? 0.5 : GX2),
                               z1 = qz == 0 ? GX0 : (qz == 1 ? 0.5 : GX2);
                  const double wq = (qx == 1 ? GW1 : GW0) * (qy == 1 ? GW1 : GW0) * (qz == 1 ? GW1 : GW0);
                  const double X[2] = {1.0 - x1, x1}, Y[2] = {1.0 - y1, y1}, Z[2] = {1.0 - z1, z1};
                  double yz[2][2], xz[2][2], xy[2][2];
#pragma unroll
                  for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                      {
                        yz[i][k] = Y[i] * Z[k];
                        xz[i][k] = X[i] * Z[k];
                        xy[i][k] = X[i] * Y[k];
                      }
                  auto dNl = [&](int vv, int e) __attribute__((always_inline)) {
                    const int vx = vv & 1, vy = (vv >> 1) & 1, vz = vv >> 2;
                    const double m = e == 0 ? yz[vy][vz] : (e == 1 ? xz[vx][vz] : xy[vx][vy]);
                    return ((vv >> e) & 1) ? m : -m;
                  };
                  double J[3][3];
#pragma unroll
                  for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                      J[i][k] = 0.0;
#pragma unroll
                  for (int vv = 0; vv < 8; ++vv)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                      {
                        const double xi = s_x[i][vv][cl];
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                          J[i][k] += xi * dNl(vv, k);
                      }
                  double inv[3][3];
                  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
                  const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
                  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
                  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
                  const double id = 1.0 / det;
                  inv[0][0] = c00 * id;
                  inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
                  inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
                  inv[1][0] = c01 * id;
                  inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
                  inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
                  inv[2][0] = c02 * id;
                  inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
                  inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
                  const double JxW = det * wq;
                  double *const o = s_q + sq_at(slot, cl);
                  const bool writer = Q3 ? part == 0 : a < 4;
                  double gu[3][3], gpf[3] = {0.0, 0.0, 0.0}, pf = 0.0, pfo = 0.0, pfoo = 0.0;
#pragma unroll
                  for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                      gu[i][k] = 0.0;
#pragma unroll
                  for (int vv = 0; vv < 8; ++vv)
                    {
                      double gv[3];
#pragma unroll
                      for (int d = 0; d < 3; ++d)
                        gv[d] = inv[0][d] * dNl(vv, 0) + inv[1][d] * dNl(vv, 1) + inv[2][d] * dNl(vv, 2);
                      const double Nv = X[vv & 1] * yz[(vv >> 1) & 1][vv >> 2];
                      if (writer)
                        {
                          *reinterpret_cast<double2 *>(o + 4 * vv) = make_double2(gv[0], gv[1]);
                          *reinterpret_cast<double2 *>(o + 4 * vv + 2) = make_double2(gv[2], Nv);
                        }
                      const double ph = s_p[0][vv][cl];
                      pf += ph * Nv;
                      pfo += s_p[1][vv][cl] * Nv;
                      pfoo += s_p[2][vv][cl] * Nv;
#pragma unroll
                      for (int d = 0; d < 3; ++d)
                        {
                          gpf[d] += ph * gv[d];
#pragma unroll
                          for (int c = 0; c < 3; ++c)
                            gu[c][d] += s_u[c][vv][cl] * gv[d];
                        }
                    }
                  // q-point state, cracks.cc:2248-2306
                  if (monolithic)
                    {
                      pf = fmax(0.0, pf);
                      pfo = fmax(0.0, pfo);
                      pfoo = fmax(0.0, pfoo);
                    }
                  const double pf_minus_old_plus = fmax(0.0, pf - pfo);
                  double pfx = pfoo + tfac * (pfo - pfoo);
                  if (pfx <= 0.0)
                    pfx = 0.0;
                  if (pfx >= 1.0)
                    pfx = 1.0;
                  if (prm.use_old_timestep_pf)
                    pfx = pfo;
                  const double g = (1 - kappa) * pfx * pfx + kappa;
                  double E[3][3], trE = 0.0, divu = 0.0;
#pragma unroll
                  for (int i = 0; i < 3; ++i)
                    {
                      divu += gu[i][i];
#pragma unroll
                      for (int k = 0; k < 3; ++k)
                        E[i][k] = 0.5 * (gu[i][k] + gu[k][i]);
                      trE += E[i][i];
                    }
                  double sp[3][3], spE = 0.0; // sigma+ = lambda tr(E) I + 2 mu E, sigma- = 0 (cracks.cc:2284-2292)
#pragma unroll
                  for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                      {
                        sp[i][k] = lam * trE * (i == k ? 1.0 : 0.0) + 2 * mu * E[i][k];
                        spE += sp[i][k] * E[i][k];
                      }
                  if (writer)
                    {
                      const bool pen_on = !((pf - pfo) < 0.0); // shadowed variable, cracks.cc:2311-2315
                      *reinterpret_cast<double2 *>(o + O_GW) = make_double2(g * JxW, 2.0 * (1 - kappa) * pf * JxW);
                      *reinterpret_cast<double2 *>(o + O_CDIV) =
                        make_double2(2.0 * aB1 * p * pf * JxW, (((1 - kappa) * spE + Gc / eps) - 2.0 * aB1 * p * divu) * JxW);
                      *reinterpret_cast<double2 *>(o + O_CGG) = make_double2(Gc * eps * JxW, pen_on ? penal_fac * JxW : 0.0);
                      *reinterpret_cast<double2 *>(o + O_SP) = make_double2(sp[0][0], sp[0][1]);
                      *reinterpret_cast<double2 *>(o + O_SP + 2) = make_double2(sp[0][2], sp[1][1]);
                      *reinterpret_cast<double2 *>(o + O_SP + 4) = make_double2(sp[1][2], sp[2][2]);
                      *reinterpret_cast<double2 *>(o + O_RP) =
                        make_double2(aB1 * p * pfx * pfx * JxW, (penal_fac * pf_minus_old_plus + (1.0 - kappa) * spE * pf - Gc / eps * (1.0 - pf) -
                                                                 2.0 * aB1 * p * pf * divu) *
                                                                  JxW);
                      *reinterpret_cast<double2 *>(o + O_GPF) = make_double2(gpf[0], gpf[1]);
                      o[O_GPF + 2] = gpf[2];
                    }
                  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                  __builtin_amdgcn_wave_barrier();
                  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
              // ---------------- phase B: q-point q, rows of this lane's vertex (cracks.cc:2308-2432)
              const double *const i = s_q + sq_at(q & (NSLOT - 1), cl);
              const double2 ga01 = *reinterpret_cast<const double2 *>(i + 4 * a), ga2n = *reinterpret_cast<const double2 *>(i + 4 * a + 2);
              const double gNa[3] = {ga01.x, ga01.y, ga2n.x}, Na = ga2n.y;
              const double2 s01 = *reinterpret_cast<const double2 *>(i + O_SP), s23 = *reinterpret_cast<const double2 *>(i + O_SP + 2),
                            s45 = *reinterpret_cast<const double2 *>(i + O_SP + 4);
              const double sp[3][3] = {{s01.x, s01.y, s23.x}, {s01.y, s23.y, s45.x}, {s23.x, s45.x, s45.y}};
              const double2 q01 = *reinterpret_cast<const double2 *>(i + O_GW);
              const double gw = q01.x;
              if constexpr (FULL)
                {
                  const double2 q23 = *reinterpret_cast<const double2 *>(i + O_CDIV), q45 = *reinterpret_cast<const double2 *>(i + O_CGG);
                  double LA[3], MA[3];
#pragma unroll
                  for (int c = 0; c < 3; ++c)
                    {
                      LA[c] = lam * gw * gNa[c];
                      MA[c] = mu * gw * gNa[c];
                    }
                  const double mgw = mu * gw;
                  const double cpu = q01.y * Na, cdiv = q23.x * Na, cpp = q23.y * Na, cgg = q45.x, cpen = q45.y * Na;
#pragma unroll
                  for (int bb = 0; bb < NBL; ++bb)
                    {
                      const int b = B0 + bb;
                      const double2 gb01 = *reinterpret_cast<const double2 *>(i + 4 * b), gb2n = *reinterpret_cast<const double2 *>(i + 4 * b + 2);
                      const double gNb[3] = {gb01.x, gb01.y, gb2n.x}, Nb = gb2n.y;
                      double t = 0.0;
#pragma unroll
                      for (int k = 0; k < 3; ++k)
                        t += gNb[k] * gNa[k];
#pragma unroll
                      for (int d = 0; d < 3; ++d)
                        {
                          double sv = 0.0;
#pragma unroll
                          for (int k = 0; k < 3; ++k)
                            sv += sp[d][k] * gNb[k];
                          Kpu[bb][d] += cpu * sv - cdiv * gNb[d];
#pragma unroll
                          for (int c = 0; c < 3; ++c)
                            Kuu[bb][c][d] += LA[c] * gNb[d] + MA[d] * gNb[c] + (c == d ? mgw * t : 0.0);
                        }
                      Kpp[bb] += cpen * Nb;
                      Kpp[bb] += cpp * Nb + cgg * t;
                    }
                }
              {
                  const double2 r01 = *reinterpret_cast<const double2 *>(i + O_RP), g01 = *reinterpret_cast<const double2 *>(i + O_GPF);
                  const double g2 = i[O_GPF + 2];
                  const double cgg_r = i[O_CGG];
#pragma unroll
                  for (int c = 0; c < 3; ++c)
                    {
                      double t = 0.0;
#pragma unroll
                      for (int k = 0; k < 3; ++k)
                        t += sp[c][k] * gNa[k];
                      R[c] -= gw * t - r01.x * gNa[c];
                    }
                  const double gg = g01.x * gNa[0] + g01.y * gNa[1] + g2 * gNa[2];
                  R[3] -= r01.y * Na + cgg_r * gg;
                }
            }
        }
      else
        {
#pragma unroll 1
      for (int q = 0; q < nq_run; ++q)
        {
          int cl = cl0;
          asm volatile("" : "+v"(cl)); // the cell's nodal data are re-read from LDS per q-point, not pinned in registers
          // ---- fe_values.reinit at q: J, J^-1, JxW (MappingQ1)
          double J[dim][dim];
          if constexpr (!PATCH)
            {
#pragma unroll
              for (int i = 0; i < dim; ++i)
#pragma unroll
                for (int j = 0; j < dim; ++j)
                  J[i][j] = 0.0;
#pragma unroll
              for (int vv = 0; vv < nv; ++vv)
#pragma unroll
                for (int i = 0; i < dim; ++i)
                  {
                    const double xi = s_x[i][vv][cl];
#pragma unroll
                    for (int j = 0; j < dim; ++j)
                      J[i][j] += xi * refdN<dim>(q, vv, j);
                  }
            }
          double inv[dim][dim], det;
          if constexpr (PATCH)
            {
              // the cells of a patch block are exact axis-parallel rectangles (pfm_host.cpp: build_patches2d)
              inv[0][0] = p_ihx;
              inv[1][1] = p_ihy;
              inv[0][1] = inv[1][0] = 0.0;
              det = p_det;
            }
          else if constexpr (dim == 2)
            {
              det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
              const double id = 1.0 / det;
              inv[0][0] = J[1][1] * id;
              inv[0][1] = -J[0][1] * id;
              inv[1][0] = -J[1][0] * id;
              inv[1][1] = J[0][0] * id;
            }
          else
            {
              const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
              const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
              const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
              det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
              const double id = 1.0 / det;
              inv[0][0] = c00 * id;
              inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
              inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
              inv[1][0] = c01 * id;
              inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
              inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
              inv[2][0] = c02 * id;
              inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
              inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
            }
          const double JxW = det * refw<dim>(q);

          // ---- shape gradients at q and the Newton state (cracks.cc:2222-2232)
          double gN[nv][dim];
          double gu[dim][dim], gpf[dim], pf = 0.0, pfo = 0.0, pfoo = 0.0;
#pragma unroll
          for (int i = 0; i < dim; ++i)
            {
              gpf[i] = 0.0;
#pragma unroll
              for (int j = 0; j < dim; ++j)
                gu[i][j] = 0.0;
            }
#pragma unroll
          for (int vv = 0; vv < nv; ++vv)
            {
#pragma unroll
              for (int d = 0; d < dim; ++d)
                {
                  double s = 0.0;
                  if constexpr (PATCH)
                    s = inv[d][d] * refdN<dim>(q, vv, d);
                  else
                    {
#pragma unroll
                      for (int e = 0; e < dim; ++e)
                        s += inv[e][d] * refdN<dim>(q, vv, e);
                    }
                  gN[vv][d] = s;
                }
              const double Nv = refN<dim>(q, vv);
              const double ph = s_p[0][vv][cl];
              pf += ph * Nv;
              pfo += s_p[1][vv][cl] * Nv;
              pfoo += s_p[2][vv][cl] * Nv;
#pragma unroll
              for (int d = 0; d < dim; ++d)
                {
                  gpf[d] += ph * gN[vv][d];
#pragma unroll
                  for (int c = 0; c < dim; ++c)
                    gu[c][d] += s_u[c][vv][cl] * gN[vv][d];
                }
            }
          // this lane's test vertex
          double gNa[dim];
#pragma unroll
          for (int d = 0; d < dim; ++d)
            {
              double s = 0.0;
              if constexpr (PATCH)
                s = inv[d][d] * refdN<dim>(q, a, d);
              else
                {
#pragma unroll
                  for (int e = 0; e < dim; ++e)
                    s += inv[e][d] * refdN<dim>(q, a, e);
                }
              gNa[d] = s;
            }
          const double Na = refN<dim>(q, a);

          // ---- q-point state, cracks.cc:2248-2306
          if (monolithic)
            {
              pf = fmax(0.0, pf);
              pfo = fmax(0.0, pfo);
              pfoo = fmax(0.0, pfoo);
            }
          const double pf_minus_old_plus = fmax(0.0, pf - pfo);
          double pfx = pfoo + tfac * (pfo - pfoo);
          if (pfx <= 0.0)
            pfx = 0.0;
          if (pfx >= 1.0)
            pfx = 1.0;
          if (prm.use_old_timestep_pf)
            pfx = pfo;
          const double g = (1 - kappa) * pfx * pfx + kappa;

          double E[dim][dim], trE = 0.0, divu = 0.0;
#pragma unroll
          for (int i = 0; i < dim; ++i)
            {
              divu += gu[i][i];
#pragma unroll
              for (int j = 0; j < dim; ++j)
                E[i][j] = 0.5 * (gu[i][j] + gu[j][i]);
              trE += E[i][i];
            }
          double sp[dim][dim], sm[dim][dim];
          if constexpr (SPLIT)
            {
              ortho_ok &= split_stress(E, trE, lam, mu, sp, sm);
            }
          else
            {
#pragma unroll
              for (int i = 0; i < dim; ++i)
#pragma unroll
                for (int j = 0; j < dim; ++j)
                  {
                    sp[i][j] = lam * trE * (i == j ? 1.0 : 0.0) + 2 * mu * E[i][j];
                    sm[i][j] = 0.0;
                  }
            }
          double spE = 0.0; // scalar_product(stress_term_plus, E)
#pragma unroll
          for (int i = 0; i < dim; ++i)
#pragma unroll
            for (int j = 0; j < dim; ++j)
              spE += sp[i][j] * E[i][j];

          // ---- Jacobian rows of vertex a, cracks.cc:2308-2389
          // Stress split (2-D): decompose_stress(derivative = true) (cracks.cc:1976-2109) is LINEAR in the direction E_LinU
          // for a fixed strain, and E_LinU of trial dof (b, d) is sym(e_d (x) grad N_b): the linearised split of all eight
          // trial dofs of the cell is the 3 x 3 tangent D (Voigt: 00, 11, 01) applied to grad N_b.  The lanes of vertex 0,
          // 1, 2 of the cell evaluate the reference function for the unit directions e_0 e_0, e_1 e_1, sym(e_0 e_1) -- one
          // call per lane and q-point instead of the two of round 3 (the lane of b for its own two trial dofs) -- fold
          // g(phi) and decompose_stress_matrix in, and the quad exchanges 3 x 4 numbers instead of 8 x 9.  The products
          // with grad N_b and grad N_a are then the usual B_a^T D B_b.
          double mineD[SPLIT && FULL ? 4 : 1];
          if constexpr (FULL && SPLIT)
            {
              const int k = a < 2 ? a : 2;
              double EL[dim][dim], spL[dim][dim], smL[dim][dim];
              EL[0][0] = k == 0 ? 1.0 : 0.0;
              EL[1][1] = k == 1 ? 1.0 : 0.0;
              EL[0][1] = EL[1][0] = k == 2 ? 0.5 : 0.0;
              ortho_ok &= split_stress_lin(E, trE, EL, k < 2 ? 1.0 : 0.0, lam, mu, spL, smL);
              double spLE = 0.0, spEL = 0.0;
#pragma unroll
              for (int i = 0; i < dim; ++i)
#pragma unroll
                for (int j = 0; j < dim; ++j)
                  {
                    spLE += spL[i][j] * E[i][j];
                    spEL += sp[i][j] * EL[i][j];
                  }
              mineD[0] = g * spL[0][0] + d_mat * smL[0][0];
              mineD[1] = g * spL[1][1] + d_mat * smL[1][1];
              mineD[2] = g * spL[0][1] + d_mat * smL[0][1];
              mineD[3] = spLE + spEL;
            }
          if constexpr (FULL && !SPLIT)
            {
              // Unsplit law sigma+ = lambda tr(E) I + 2 mu E (sigma- = 0): the linearised stress of trial dof (b, d)
              // contracted with the test gradient is  lambda gN_b[d] gN_a[c] + mu (gN_b[c] gN_a[d] + delta_cd gN_b.gN_a),
              // and sigma+_LinU : E = sigma+ : E_LinU = sum_k sigma+[d][k] gN_b[k]  (cracks.cc:2340-2349, 2357-2376 with
              // the tensors written out: 47 instead of ~190 operations per trial vertex and q-point in 3-D)
              const double gw = g * JxW;
              double LA[dim], MA[dim];
#pragma unroll
              for (int c = 0; c < dim; ++c)
                {
                  LA[c] = lam * gw * gNa[c];
                  MA[c] = mu * gw * gNa[c];
                }
              const double mgw = mu * gw;
              const double cpu = 2.0 * (1 - kappa) * pf * Na * JxW, cdiv = 2.0 * aB1 * p * pf * Na * JxW;
              const double cpp = ((1 - kappa) * spE + Gc / eps) * Na * JxW, cgg = Gc * eps * JxW, cdu = 2.0 * aB1 * p * divu * Na * JxW;
              const bool pen_on = !((pf - pfo) < 0.0); // shadowed variable, cracks.cc:2311-2315
              const double cpen = penal_fac * Na * JxW;
#pragma unroll
              for (int bb = 0; bb < NBL; ++bb)
                {
                  const int b = B0 + bb;
                  const double Nb = refN<dim>(q, b);
                  double t = 0.0;
#pragma unroll
                  for (int k = 0; k < dim; ++k)
                    t += gN[b][k] * gNa[k];
#pragma unroll
                  for (int d = 0; d < dim; ++d)
                    {
                      double sv = 0.0;
#pragma unroll
                      for (int k = 0; k < dim; ++k)
                        sv += sp[d][k] * gN[b][k];
                      Kpu[bb][d] += cpu * sv - cdiv * gN[b][d];
#pragma unroll
                      for (int c = 0; c < dim; ++c)
                        Kuu[bb][c][d] += LA[c] * gN[b][d] + MA[d] * gN[b][c] + (c == d ? mgw * t : 0.0);
                    }
                  Kpp[bb] += cpen * (pen_on ? Nb : 0.0);
                  Kpp[bb] += (cpp - cdu) * Nb + cgg * t;
                }
            }
          if constexpr (FULL && SPLIT)
            {
              double D[3][3], lin[3]; // D[r][k]: stress component r (00, 11, 01) of unit direction k
#pragma unroll
              for (int k = 0; k < 3; ++k)
                {
#pragma unroll
                  for (int r = 0; r < 3; ++r)
                    D[r][k] = quad_bcast(mineD[r], k);
                  lin[k] = quad_bcast(mineD[3], k);
                }
              // G = (B_a^T D) JxW, B_a = [gNa_x 0; 0 gNa_y; gNa_y gNa_x]
              const double gax = gNa[0] * JxW, gay = gNa[1] * JxW;
              double G[dim][3];
#pragma unroll
              for (int k = 0; k < 3; ++k)
                {
                  G[0][k] = gax * D[0][k] + gay * D[2][k];
                  G[1][k] = gay * D[1][k] + gax * D[2][k];
                }
              const double NaW = Na * JxW;
              const double cA = (1 - kappa) * pf * NaW, cB = 2.0 * aB1 * p * pf * NaW;
              const double m0 = cA * lin[0] - cB, m1 = cA * lin[1] - cB, m2 = cA * lin[2];
              const double cpp = ((1 - kappa) * spE + Gc / eps) * NaW - 2.0 * aB1 * p * divu * NaW, cgg = Gc * eps * JxW;
              const double cpen = ((pf - pfo) < 0.0) ? 0.0 : penal_fac * NaW; // shadowed variable, cracks.cc:2311-2315
#pragma unroll
              for (int b = 0; b < nv; ++b)
                {
                  const double gbx = gN[b][0], gby = gN[b][1], Nb = refN<dim>(q, b);
                  // trial dofs (b, d): rows (a, c) = sum_k G[c][k] B_b[k][d], B_b[.][0] = (gbx, 0, gby), B_b[.][1] = (0, gby, gbx)
#pragma unroll
                  for (int c = 0; c < dim; ++c)
                    {
                      Kuu[b][c][0] += G[c][0] * gbx + G[c][2] * gby;
                      Kuu[b][c][1] += G[c][1] * gby + G[c][2] * gbx;
                    }
                  // row (a, phi): (1 - kappa) (sigma+_LinU : E + sigma+ : E_LinU) pf N_a - 2 (alpha - 1) p pf div(u_LinU) N_a
                  Kpu[b][0] += m0 * gbx + m2 * gby;
                  Kpu[b][1] += m1 * gby + m2 * gbx;
                  // trial dof (b, phi): rows (a, c < dim) get exactly 0 (cracks.cc:2333-2337)
                  Kpp[b] += (cpen + cpp) * Nb + cgg * (gbx * gNa[0] + gby * gNa[1]);
                }
            }

            {
          // ---- residual rows of vertex a, cracks.cc:2393-2432
#pragma unroll
          for (int c = 0; c < dim; ++c)
            {
              double t = 0.0, tm = 0.0;
#pragma unroll
              for (int k = 0; k < dim; ++k)
                {
                  t += g * sp[c][k] * gNa[k];
                  tm += sm[c][k] * gNa[k];
                }
              R[c] -= (t + d_rhs * tm - aB1 * p * pfx * pfx * gNa[c]) * JxW;
            }
          {
            double gg = 0.0;
#pragma unroll
            for (int k = 0; k < dim; ++k)
              gg += gpf[k] * gNa[k];
            R[dim] -= penal_fac * pf_minus_old_plus * Na * JxW;
            R[dim] -= ((1.0 - kappa) * spE * pf * Na - Gc / eps * (1.0 - pf) * Na + Gc * eps * gg -
                       2.0 * aB1 * p * pf * divu * Na) *
                      JxW;
          }
            }
        } // q
        }

      if (!ortho_ok)
        atomicMax(v.status, (int)PFM_ERR_NOT_ORTHOGONAL);

      if constexpr (PATCH)
        {
          // =============================== rows of the block's regular nodes, completed in LDS
          const int cx = cl0 % 8, cy = cl0 / 8, ax = a & 1, ay = a >> 1;
          const int nx = cx + ax, ny = cy + ay; // position of the lane's vertex among the block's 9 x 9 nodes
          const bool mine = active && nx >= 1 && nx <= 7 && ny >= 1 && ny <= 7;
          double *row = s_row + ((nx - 1) + 7 * (ny - 1)) * PRW;
          // |K_ii| of this cell, or the mean |diagonal| of its element matrix (deal.II's placeholder rule)
          double diag[nc] = {0.0, 0.0, 0.0};
          if constexpr (FULL)
            {
#pragma unroll
              for (int b = 0; b < nv; ++b)
                if (b == a)
                  {
                    diag[0] = fabs(Kuu[b][0][0]);
                    diag[1] = fabs(Kuu[b][1][1]);
                    diag[2] = fabs(Kpp[b]);
                  }
            }
          double dsum = diag[0] + diag[1] + diag[2];
#pragma unroll
          for (int m = 1; m < nv; m <<= 1)
            dsum += __shfl_xor(dsum, m, nv);
          const double avg = dsum / (double)dpc;
#pragma unroll 1
          for (int phase = 0; phase < nv; ++phase)
            {
              // (ds_add_f64: a node receives one cell per phase, the lanes of a phase hit distinct addresses)
              if (mine && a == phase)
                {
                  if constexpr (FULL)
                    {
#pragma unroll
                      for (int b = 0; b < nv; ++b)
                        {
                          const int o = ((b & 1) - ax + 1) + 3 * ((b >> 1) - ay + 1); // lattice offset of the trial vertex
#pragma unroll
                          for (int c = 0; c < dim; ++c)
#pragma unroll
                            for (int d = 0; d < dim; ++d)
                              unsafeAtomicAdd(&row[(c * 9 + o) * 3 + d], Kuu[b][c][d]);
#pragma unroll
                          for (int d = 0; d < dim; ++d)
                            unsafeAtomicAdd(&row[(dim * 9 + o) * 3 + d], Kpu[b][d]);
                          unsafeAtomicAdd(&row[(dim * 9 + o) * 3 + dim], Kpp[b]);
                        }
#pragma unroll
                      for (int c = 0; c < nc; ++c)
                        unsafeAtomicAdd(&row[84 + c], diag[c] != 0.0 ? diag[c] : avg);
                    }
#pragma unroll
                  for (int c = 0; c < nc; ++c)
                    unsafeAtomicAdd(&row[81 + c], R[c]);
                }
              __syncthreads();
            }
          const bool total_via_update = !(prm.outer_solver == PFM_SOLVER_ACTIVE_SET);
          if constexpr (FULL)
            {
              // in the order of the destination: the 81 values of a row are contiguous (interleaved layout) or four contiguous
              // pieces (blocked layout) -- whole lines, no read.  A thread keeps its position r in the row (component pair and
              // slot, decoded once) and walks over the nodes, three rows per pass of the workgroup.
              const bool il = v.layout == PFM_LAYOUT_INTERLEAVED;
              const int sub = tid / 81, r = tid - sub * 81;
              int c, sl, d, piece, rr, mul;
              if (il)
                {
                  c = r / 27;
                  sl = (r - c * 27) / 3;
                  d = r % 3;
                  piece = 0, rr = r, mul = 9;
                }
              else if (r < 36)
                {
                  c = r / 18;
                  sl = (r - c * 18) / 2;
                  d = r & 1;
                  piece = 0, rr = r, mul = 4;
                }
              else if (r < 54)
                {
                  c = (r - 36) / 9;
                  sl = (r - 36) - c * 9;
                  d = 2;
                  piece = 1, rr = r - 36, mul = 2;
                }
              else if (r < 72)
                {
                  c = 2;
                  sl = (r - 54) / 2;
                  d = (r - 54) & 1;
                  piece = 2, rr = r - 54, mul = 2;
                }
              else
                {
                  c = 2;
                  sl = r - 72;
                  d = 2;
                  piece = 3, rr = r - 72, mul = 1;
                }
              double *const base = vals.b[piece] + rr;
              const unsigned *inv32 = reinterpret_cast<const unsigned *>(s_inv) + (sl >= 8 ? 1 : 0);
              const int sh = 4 * (sl & 7);
              const int src0 = c * 27 + d;
              if (sub < 3)
                for (int nl = sub, hx = sub + 1, hy = 1; nl < 49; nl += 3)
                  {
                    const long long off = s_off[nl];
                    const int hp = hx + 9 * hy;
                    hx += 3;
                    if (hx > 7)
                      hx -= 7, ++hy;
                    if (off < 0)
                      continue;
                    const int o = (int)((inv32[2 * nl] >> sh) & 15u);
                    const int oy = (o * 11) >> 5; // o / 3
                    const unsigned fA = (unsigned)s_fl[hp], fQ = (unsigned)s_fl[hp + o + 6 * oy - 10];
                    const bool rcon = (fA >> c) & 1u, ccon = (fQ >> d) & 1u;
                    double val = s_row[nl * PRW + src0 + 3 * o];
                    if (rcon || ccon)
                      val = (rcon && o == 4 && d == c) ? s_row[nl * PRW + 84 + c] : 0.0;
                    base[mul * off] = val;
                  }
            }
          for (int e = tid; e < 49 * nc; e += 256)
            {
              const int nl = e / nc, c = e - nl * nc;
              if (s_off[nl] < 0)
                continue;
              const int hp = (nl % 7 + 1) + 9 * (nl / 7 + 1);
              const int node = s_bn[hp];
              const bool con = ((unsigned)s_fl[hp] >> c) & 1u;
              const double rv = s_row[nl * PRW + 81 + c];
              const long long di = dof_index<dim>(v, node, c);
              res_pde[di] = con ? 0.0 : rv;
              if (residual_only)
                res_tot[di] = (!con || !total_via_update) ? rv : 0.0;
            }
          return;
        }

      // =============================== scatter through the constraints (cracks.cc:2439-2464)
      // 3-D cell at a hanging vertex with a record in DevView::cres: its matrix and residual are reduced to its distinct
      // constraint-resolved nodes in LDS and every value is added ONCE (below).  Such cells sit in plain colour classes
      // (pfm_host.cpp: greedy_colours over the resolved nodes): plain read-modify-write in a fixed order, bitwise reproducible.
      const uint8_t *rec = nullptr;
      int RN = 0xff;
      if constexpr (dim == 3 && !PATCH)
        if (v.cres)
          {
            const int hc3 = v.hcell[cell];
            if (hc3 >= 0)
              {
                rec = v.cres + (long long)hc3 * PFM_CRES_BYTES;
                RN = (int)rec[PFM_CRES_R];
              }
          }
      const bool red = RN <= 16;
      if (!ATOMIC && !red)
        {
          // Colour class without hanging vertices: the rows of node A are touched by this thread only during this
          // launch.  Every batch of read-modify-writes loads all its old values before the first store (the adds of a
          // batch hit distinct entries; one HBM/L2 round trip per batch instead of one per entry).
          // rows of ghost nodes belong to another rank, rows of regular nodes to the patch kernel (DevView::row_patch)
          const bool owned = A < v.n_owned && !(v.row_patch && v.row_patch[A]);
          const bool total_via_update = !(prm.outer_solver == PFM_SOLVER_ACTIVE_SET);
          const unsigned fA = v.node_flags[A];
          // a cell next to the atomic class (which may be running on another stream): atomic adds, see DevView::cell_ring
          const bool ring = RING && v.cell_ring[cell] != 0;
          if (owned && resid_lane)
            {
            double *pr[nc], *pt[nc], o_r[nc], o_t[nc];
#pragma unroll
            for (int c = 0; c < nc; ++c)
              {
                const long long di = dof_index<dim>(v, A, c);
                pr[c] = res_pde + di;
                pt[c] = res_tot + di;
                o_r[c] = ring ? 0.0 : *pr[c];
                if (residual_only)
                  o_t[c] = ring ? 0.0 : *pt[c];
              }
#pragma unroll
            for (int c = 0; c < nc; ++c)
              {
                const bool con = (fA >> c) & 1u;
                if (!con)
                  {
                    if (ring)
                      add_to<true>(pr[c], R[c]);
                    else
                      *pr[c] = o_r[c] + R[c];
                  }
                if (residual_only && (!con || !total_via_update))
                  {
                    if (ring)
                      add_to<true>(pt[c], R[c]);
                    else
                      *pt[c] = o_t[c] + R[c];
                  }
              }
          }
          if constexpr (FULL)
            {
              const uint8_t *cs = v.cslot + (long long)cell * nv * nv;
              const long long off = owned ? v.nadj_ptr[A] : 0, deg = owned ? v.nadj_ptr[A + 1] - off : 0;
              const bool il = v.layout == PFM_LAYOUT_INTERLEAVED;
              // entry (row comp c, slot s, col comp d) of node A's rows, see val_ptr
              auto entry = [&](int c, int s_, int d) __attribute__((always_inline)) -> double * {
                if (il)
                  return vals.b[0] + (nc * nc * off + (long long)c * nc * deg + (long long)s_ * nc + d);
                if (c < dim)
                  return d < dim ? vals.b[0] + (dim * dim * off + (long long)c * dim * deg + (long long)s_ * dim + d)
                                 : vals.b[1] + (dim * off + (long long)c * deg + s_);
                return d < dim ? vals.b[2] + (dim * off + (long long)s_ * dim + d) : vals.b[3] + (off + s_);
              };
              double diag[nc]; // |diagonal| of this lane's vertex (zero in the lanes whose trial vertices do not include a)
#pragma unroll
              for (int c = 0; c < nc; ++c)
                diag[c] = 0.0;
#pragma unroll
              for (int bb = 0; bb < NBL; ++bb)
                {
                  const int b = B0 + bb;
                  const int B = v.conn[(long long)b * v.n_cells + cell];
                  const unsigned fQ = v.node_flags[B];
                  const int slot = (int)cs[a * nv + b];
                  if (b == a)
                    {
#pragma unroll
                      for (int c = 0; c < dim; ++c)
                        diag[c] = fabs(Kuu[bb][c][c]);
                      diag[dim] = fabs(Kpp[bb]);
                    }
                  if (!owned)
                    continue;
                  constexpr int NE = dim * dim + dim + 1;
                  double *pe[NE], oe[NE], ke[NE];
                  bool on[NE];
#pragma unroll
                  for (int c = 0; c < dim; ++c)
#pragma unroll
                    for (int d = 0; d < dim; ++d)
                      {
                        pe[c * dim + d] = entry(c, slot, d);
                        ke[c * dim + d] = Kuu[bb][c][d];
                        on[c * dim + d] = !((fA >> c) & 1u) && !((fQ >> d) & 1u);
                      }
#pragma unroll
                  for (int d = 0; d < dim; ++d)
                    {
                      pe[dim * dim + d] = entry(dim, slot, d);
                      ke[dim * dim + d] = Kpu[bb][d];
                      on[dim * dim + d] = !((fA >> dim) & 1u) && !((fQ >> d) & 1u);
                    }
                  pe[NE - 1] = entry(dim, slot, dim);
                  ke[NE - 1] = Kpp[bb];
                  on[NE - 1] = !((fA >> dim) & 1u) && !((fQ >> dim) & 1u);
#pragma unroll
                  for (int e = 0; e < NE; ++e)
                    oe[e] = ring ? 0.0 : *pe[e];
#pragma unroll
                  for (int e = 0; e < NE; ++e)
                    if (on[e])
                      {
                        if (ring)
                          add_to<true>(pe[e], ke[e]);
                        else
                          *pe[e] = oe[e] + ke[e];
                      }
                }
              // diagonal of constrained rows (deal.II distribute_local_to_global): |K_ii| or, when that is zero, the
              // mean |diagonal| of the element matrix
              double dsum = 0.0;
#pragma unroll
              for (int c = 0; c < nc; ++c)
                dsum += diag[c];
#pragma unroll
              for (int m = 1; m < LPC; m <<= 1)
                dsum += __shfl_xor(dsum, m, LPC);
              const double avg = dsum / (double)dpc;
              if (fA && owned && diag_lane)
                {
                  const int slot = (int)cs[a * nv + a];
#pragma unroll
                  for (int c = 0; c < nc; ++c)
                    if ((fA >> c) & 1u)
                      {
                        double *pd = entry(c, slot, c);
                        if (ring)
                          add_to<true>(pd, diag[c] != 0.0 ? diag[c] : avg);
                        else
                          *pd += diag[c] != 0.0 ? diag[c] : avg;
                      }
                }
            }
          return;
        }
      const uint8_t *cs = v.cslot + (long long)cell * nv * nv;
      constexpr int MPH = dim == 3 ? 4 : 2;
      bool add_atomically = ATOMIC;
      if constexpr (RING)
        add_atomically = add_atomically || v.cell_ring[cell] != 0;
      auto add_rt = [&](double *p, double x) __attribute__((always_inline)) {
        if (add_atomically)
          unsafeAtomicAdd(p, x);
        else
          *p += x;
      };
      const int hc = v.cslot_h ? v.hcell[cell] : -1;
      const uint8_t *csh = hc >= 0 ? v.cslot_h + (long long)hc * (nv * MPH * nv * MPH) : nullptr;
      const int kA = v.hn_index ? v.hn_index[A] : -1;
      const long long rb = kA < 0 ? 0 : v.hn_ptr[kA];
      const long long re = kA < 0 ? 1 : v.hn_ptr[kA + 1];

      // residual
      const bool total_via_update = !(prm.outer_solver == PFM_SOLVER_ACTIVE_SET);
      if constexpr (dim == 3 && !PATCH)
        if (red)
          {
            // R' = C^T R: the entries by vertex into LDS, lane i takes resolved node i (the lanes of a cell sit in one wave)
            if (resid_lane)
              {
#pragma unroll
                for (int c = 0; c < nc; ++c)
                  s_rr[cl][a][c] = R[c];
              }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int32_t *rnode = reinterpret_cast<const int32_t *>(rec);
            for (int i = a + nv * part; i < RN; i += LPC)
              {
                const int P = rnode[i];
                if (P >= v.n_owned || (v.row_patch && v.row_patch[P]))
                  continue;
                double acc[nc];
#pragma unroll
                for (int c = 0; c < nc; ++c)
                  acc[c] = 0.0;
                for (int a2 = 0; a2 < nv; ++a2)
                  {
                    const int A2 = v.conn[(long long)a2 * v.n_cells + cell];
                    const int k2 = v.hn_index[A2];
                    if (k2 < 0)
                      {
                        if (A2 == P)
#pragma unroll
                          for (int c = 0; c < nc; ++c)
                            acc[c] += s_rr[cl][a2][c];
                      }
                    else
                      for (long long r = v.hn_ptr[k2]; r < v.hn_ptr[k2 + 1]; ++r)
                        if (v.hn_parents[r] == P)
                          {
                            const double w = v.hn_weights[r];
#pragma unroll
                            for (int c = 0; c < nc; ++c)
                              acc[c] = fma(w, s_rr[cl][a2][c], acc[c]);
                          }
                  }
                if constexpr (SCR)
                  {
                    double *o = v.hs_RD + (long long)v.hcell[cell] * PFM_HS_RD + i * 4;
#pragma unroll
                    for (int c = 0; c < nc; ++c)
                      o[c] = acc[c]; // unmasked: k_hanging_gather applies the flags
                    continue;
                  }
                const unsigned fP = v.node_flags[P];
#pragma unroll
                for (int c = 0; c < nc; ++c)
                  {
                    const long long di = dof_index<dim>(v, P, c);
                    const bool con = (fP >> c) & 1u;
                    if (!con)
                      add_rt(res_pde + di, acc[c]);
                    if (residual_only && (!con || !total_via_update))
                      add_rt(res_tot + di, acc[c]);
                  }
              }
          }
      for (long long r = rb; r < ((resid_lane && !red) ? re : rb); ++r)
        {
          const int P = kA < 0 ? A : v.hn_parents[r];
          const double wP = kA < 0 ? 1.0 : v.hn_weights[r];
          if (P >= v.n_owned || (v.row_patch && v.row_patch[P]))
            continue;
          const unsigned fP = v.node_flags[P];
#pragma unroll
          for (int c = 0; c < nc; ++c)
            {
              const long long di = dof_index<dim>(v, P, c);
              const bool con = (fP >> c) & 1u;
              if (!con)
                add_to<ATOMIC>(res_pde + di, wP * R[c]);
              if (residual_only && (!con || !total_via_update))
                add_to<ATOMIC>(res_tot + di, wP * R[c]);
            }
        }

      if constexpr (FULL)
        {
          const unsigned fA = v.node_flags[A];
          double diag[nc];
#pragma unroll
          for (int c = 0; c < nc; ++c)
            diag[c] = 0.0;
          // The class of the 3-D cells at hanging vertices was bound by the rate of the FP64 atomics (a hex at a refined face
          // sends (5 + 2 + 2 + 4)^2 = 169 node pairs x 13 values through its parents, 4.7 ms for 4.2e4 hexes): the pairs fall on
          // only 8 x 8 DISTINCT constraint-resolved nodes.  So the lanes of the hex first form K' = C^T K C in LDS -- the 8 x 8
          // blocks into the cell's region of s_q, lane i < R collects the vertices that feed resolved node i, lane p takes the
          // node pairs p, p + 32, ... -- and add each value of K' once (DevView::cres: resolved nodes and their slots).
          bool reduced = false;
          if constexpr (KRED)
            {
              const int R = RN;
              if (red)
                {
                  reduced = true;
                  const int32_t *rnode = reinterpret_cast<const int32_t *>(rec);
                  double *const Kc = s_q + cl * KCELL; // (the q-loop of this hex is over: its lanes sit in one wave)
                  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                  __builtin_amdgcn_wave_barrier();
#pragma unroll
                  for (int bb = 0; bb < NBL; ++bb)
                    {
                      double *o = Kc + (a * nv + B0 + bb) * KCMP;
#pragma unroll
                      for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int d = 0; d < 3; ++d)
                          o[3 * c + d] = Kuu[bb][c][d];
#pragma unroll
                      for (int d = 0; d < 3; ++d)
                        o[9 + d] = Kpu[bb][d];
                      o[12] = Kpp[bb];
                    }
                  const int l32 = a + nv * part; // lane of the hex
                  if (l32 < R)
                    {
                      const int mine = rnode[l32];
                      int cnt = 0;
                      for (int a2 = 0; a2 < nv; ++a2)
                        {
                          const int A2 = v.conn[(long long)a2 * v.n_cells + cell];
                          const int k2 = v.hn_index[A2];
                          if (k2 < 0)
                            {
                              if (A2 == mine)
                                {
                                  s_ia[cl][l32][cnt] = (uint8_t)a2;
                                  s_iw[cl][l32][cnt++] = 1.0;
                                }
                            }
                          else
                            for (long long r = v.hn_ptr[k2]; r < v.hn_ptr[k2 + 1]; ++r)
                              if (v.hn_parents[r] == mine && cnt < 8)
                                {
                                  s_ia[cl][l32][cnt] = (uint8_t)a2;
                                  s_iw[cl][l32][cnt++] = v.hn_weights[r];
                                }
                        }
                      s_icnt[cl][l32] = (uint8_t)cnt;
                    }
                  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                  __builtin_amdgcn_wave_barrier();
                  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                  for (int pq = l32; pq < R * R; pq += 32)
                    {
                      const int i = pq / R, j = pq - i * R;
                      const int P = rnode[i], Q = rnode[j];
                      const int slot = (int)rec[PFM_CRES_SLOT + 16 * i + j];
                      if (P >= v.n_owned || (v.row_patch && v.row_patch[P]) || slot == 0xff)
                        continue;
                      double acc[KCMP];
#pragma unroll
                      for (int k = 0; k < KCMP; ++k)
                        acc[k] = 0.0;
                      const int ni = s_icnt[cl][i], nj = s_icnt[cl][j];
                      for (int x = 0; x < ni; ++x)
                        for (int y = 0; y < nj; ++y)
                          {
                            const double w = s_iw[cl][i][x] * s_iw[cl][j][y];
                            const double *k = Kc + ((int)s_ia[cl][i][x] * nv + (int)s_ia[cl][j][y]) * KCMP;
#pragma unroll
                            for (int t = 0; t < KCMP; ++t)
                              acc[t] = fma(w, k[t], acc[t]);
                          }
                      if constexpr (SCR)
                        {
                          double *o = v.hs_K + v.hs_off[v.hcell[cell]] + (long long)pq * KCMP;
#pragma unroll
                          for (int t = 0; t < KCMP; ++t)
                            o[t] = acc[t]; // unmasked: k_hanging_gather applies the flags
                          continue;
                        }
                      const unsigned fP = v.node_flags[P], fQ = v.node_flags[Q];
#pragma unroll
                      for (int c = 0; c < 3; ++c)
                        {
                          if ((fP >> c) & 1u)
                            continue;
#pragma unroll
                          for (int d = 0; d < 3; ++d)
                            if (!((fQ >> d) & 1u))
                              add_rt(val_ptr<dim>(v, vals, P, c, slot, d), acc[3 * c + d]);
                        }
                      if (!((fP >> 3) & 1u))
                        {
#pragma unroll
                          for (int d = 0; d < 3; ++d)
                            if (!((fQ >> d) & 1u))
                              add_rt(val_ptr<dim>(v, vals, P, 3, slot, d), acc[9 + d]);
                          if (!((fQ >> 3) & 1u))
                            add_rt(val_ptr<dim>(v, vals, P, 3, slot, 3), acc[12]);
                        }
                    }
                }
            }
          // matrix rows of vertex a
#pragma unroll
          for (int bb = 0; bb < NBL; ++bb)
            {
              const int b = B0 + bb;
              const int B = v.conn[(long long)b * v.n_cells + cell];
              if (b == a)
                {
#pragma unroll
                  for (int c = 0; c < dim; ++c)
                    diag[c] = fabs(Kuu[bb][c][c]);
                  diag[dim] = fabs(Kpp[bb]);
                }
              if (reduced)
                continue;
              const int kB = v.hn_index ? v.hn_index[B] : -1;
              const long long cb = kB < 0 ? 0 : v.hn_ptr[kB];
              const long long ce = kB < 0 ? 1 : v.hn_ptr[kB + 1];
              for (long long r = rb; r < re; ++r)
                {
                  const int P = kA < 0 ? A : v.hn_parents[r];
                  const double wP = kA < 0 ? 1.0 : v.hn_weights[r];
                  if (P >= v.n_owned || (v.row_patch && v.row_patch[P]))
                    continue;
                  const unsigned fP = v.node_flags[P];
                  for (long long s = cb; s < ce; ++s)
                    {
                      const int Q = kB < 0 ? B : v.hn_parents[s];
                      const double w = wP * (kB < 0 ? 1.0 : v.hn_weights[s]);
                      const unsigned fQ = v.node_flags[Q];
                      // a row search per parent pair (find_slot: a chain of dependent loads) made the class of the cells at
                      // hanging vertices take 5.5 ms for 4.2e4 hexes; their slots come from a table now (DevView::cslot_h)
                      const int slot = (kA < 0 && kB < 0) ? (int)cs[a * nv + b]
                                       : csh                ? (int)csh[((a * MPH + (int)(r - rb)) * nv + b) * MPH + (int)(s - cb)]
                                                            : find_slot(v, P, Q);
#pragma unroll
                      for (int c = 0; c < dim; ++c)
                        {
                          if ((fP >> c) & 1u)
                            continue;
#pragma unroll
                          for (int d = 0; d < dim; ++d)
                            if (!((fQ >> d) & 1u))
                              add_to<ATOMIC>(val_ptr<dim>(v, vals, P, c, slot, d), w * Kuu[bb][c][d]);
                        }
                      if (!((fP >> dim) & 1u))
                        {
#pragma unroll
                          for (int d = 0; d < dim; ++d)
                            if (!((fQ >> d) & 1u))
                              add_to<ATOMIC>(val_ptr<dim>(v, vals, P, dim, slot, d), w * Kpu[bb][d]);
                          if (!((fQ >> dim) & 1u))
                            add_to<ATOMIC>(val_ptr<dim>(v, vals, P, dim, slot, dim), w * Kpp[bb]);
                        }
                    }
                }
            }
          // diagonal of constrained rows (deal.II distribute_local_to_global): |K_ii| or,
          // when that is zero, the mean |diagonal| of the element matrix
          double dsum = 0.0;
#pragma unroll
          for (int c = 0; c < nc; ++c)
            dsum += diag[c];
#pragma unroll
          for (int m = 1; m < LPC; m <<= 1)
            dsum += __shfl_xor(dsum, m, LPC);
          const double avg = dsum / (double)dpc;
          if constexpr (SCR)
            {
              if (reduced)
                {
                  if (diag_lane)
                    {
                      double *o = v.hs_RD + (long long)v.hcell[cell] * PFM_HS_RD + 64 + a * 4;
#pragma unroll
                      for (int c = 0; c < nc; ++c)
                        o[c] = diag[c] != 0.0 ? diag[c] : avg;
                    }
                  return;
                }
            }
          if (diag_lane && A < v.n_owned && !(v.row_patch && v.row_patch[A]) && (kA >= 0 || fA))
            {
              const int slot = (int)cs[a * nv + a];
#pragma unroll
              for (int c = 0; c < nc; ++c)
                if (kA >= 0 || ((fA >> c) & 1u))
                  add_rt(val_ptr<dim>(v, vals, A, c, slot, c), diag[c] != 0.0 ? diag[c] : avg);
            }
        }
    }


    // ------------------------------------------------------------ round 6: the cells at hanging vertices, deterministically
    // One wave per destination row (node P).  Its entries -- (cell hc at a hanging vertex, index) in ascending order,
    // pfm_ctx::d_hg_list -- are summed into the row's accumulators in LDS (13 values per neighbour slot + 4 residual entries)
    // and added to the outputs once, behind the plain colour classes (stream order).  Four entries are FETCHED at a time (16
    // lanes each: lane j < R holds the 13 values of the node pair (i, j) of K' = C^T K C, the flags of node j and the slot of
    // j in the row), then ADDED one entry after the other: distinct nodes j are distinct slots, so the lanes of an entry hit
    // distinct accumulators, and the summation order of every value is the order of the list -- no atomics, bitwise
    // reproducible.  Masks (constraint flags of row and column) as in the atomic class; lane 15 of an entry's group adds the
    // placeholder diagonal of a constrained vertex (deal.II distribute_local_to_global) and the residual entries.
    // Rows of more than HG_DEG neighbours (none on 2:1 meshes of hexes) take the entries one by one with plain
    // read-modify-writes of the outputs (same order).
    constexpr int HG_DEG = 64;
    template <bool FULL>
    __global__ __launch_bounds__(256) void k_hanging_gather(DevView v, pfm_params prm, Vals vals, double *__restrict__ res_pde,
                                                            double *__restrict__ res_tot, int residual_only,
                                                            const int32_t *__restrict__ rows, const long long *__restrict__ ptr,
                                                            const HgEntry *__restrict__ list, long long n_rows)
    {
      constexpr int dim = 3, nc = 4, KCMP = 13;
      __shared__ double s_acc[4][FULL ? HG_DEG * KCMP : 1];
      __shared__ double s_res[4][nc];
      const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
      const long long r = (long long)blockIdx.x * 4 + w;
      if (r >= n_rows)
        return;
      const int P = rows[r];
      if (v.row_patch && v.row_patch[P])
        return; // a regular row: the level lattices of the overlay write it
      const unsigned fP = v.node_flags[P];
      const bool total_via_update = !(prm.outer_solver == PFM_SOLVER_ACTIVE_SET);
      const int deg = (int)(v.nadj_ptr[P + 1] - v.nadj_ptr[P]);
      const long long e0 = ptr[r], e1 = ptr[r + 1];
      const bool fast = deg <= HG_DEG;
      double *acc = s_acc[w], *racc = s_res[w];
      if (FULL && fast)
        for (int n = lane; n < deg * KCMP; n += 64)
          acc[n] = 0.0;
      if (lane < nc)
        racc[lane] = 0.0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      auto add_value = [&](int slot, int t, double x) __attribute__((always_inline)) {
        if (fast)
          acc[slot * KCMP + t] += x;
        else
          {
            const int c = t < 9 ? t / 3 : 3, d = t < 9 ? t - 3 * c : t - 9;
            *val_ptr<dim>(v, vals, P, c, slot, d) += x;
          }
      };
      for (long long eb = e0; eb < e1; eb += 4)
        {
          // ---- fetch: group g <-> entry eb + g
          const long long e = eb + g;
          const bool have = e < e1;
          HgEntry en{0, 0, 0};
          if (have)
            en = list[e];
          const int hc = en.code >> 5, idx = en.code & 31, R = en.R;
          const uint8_t *rec = v.cres + (long long)hc * PFM_CRES_BYTES;
          const double *RD = v.hs_RD + (long long)hc * PFM_HS_RD;
          double kv[KCMP];
          int slot = 0xff;
          unsigned fQ = 0;
          const bool pair = FULL && have && idx < 16 && j < R;
          if (pair)
            {
              slot = (int)rec[PFM_CRES_SLOT + 16 * idx + j];
              fQ = v.node_flags[reinterpret_cast<const int32_t *>(rec)[j]];
              const double *src = v.hs_K + en.koff + (long long)j * KCMP;
#pragma unroll
              for (int t = 0; t < KCMP; ++t)
                kv[t] = src[t];
            }
          // lane 15 of the group: residual entries of resolved node idx; placeholder diagonal of the vertex that IS node P
          double rv[nc] = {0.0, 0.0, 0.0, 0.0}, dv[nc] = {0.0, 0.0, 0.0, 0.0};
          int dslot = 0xff;
          unsigned dmask = 0;
          if (have && j == 15)
            {
              if (idx < 16)
                {
#pragma unroll
                  for (int c = 0; c < nc; ++c)
                    rv[c] = RD[idx * 4 + c];
                }
              if (FULL && (idx >= 16 || fP != 0u))
                {
                  const int cell = *reinterpret_cast<const int32_t *>(rec + PFM_CRES_CELL);
                  int a = idx >= 16 ? idx - 16 : -1;
                  if (a < 0)
                    for (int a2 = 0; a2 < 8; ++a2)
                      if (v.conn[(long long)a2 * v.n_cells + cell] == P)
                        a = a2;
                  if (a >= 0)
                    {
                      dslot = (int)v.cslot[(long long)cell * 64 + a * 8 + a];
                      dmask = idx >= 16 ? 0xfu : fP; // the row of a hanging vertex: every component
#pragma unroll
                      for (int c = 0; c < nc; ++c)
                        dv[c] = RD[64 + a * 4 + c];
                    }
                }
            }
          // ---- add: one entry after the other, in list order
#pragma unroll 1
          for (int gg = 0; gg < 4; ++gg)
            {
              if (g == gg)
                {
                  if (pair && slot != 0xff)
                    {
#pragma unroll
                      for (int c = 0; c < 3; ++c)
                        if (!((fP >> c) & 1u))
                          {
#pragma unroll
                            for (int d = 0; d < 3; ++d)
                              if (!((fQ >> d) & 1u))
                                add_value(slot, 3 * c + d, kv[3 * c + d]);
                          }
                      if (!((fP >> 3) & 1u))
                        {
#pragma unroll
                          for (int d = 0; d < 3; ++d)
                            if (!((fQ >> d) & 1u))
                              add_value(slot, 9 + d, kv[9 + d]);
                          if (!((fQ >> 3) & 1u))
                            add_value(slot, 12, kv[12]);
                        }
                    }
                  if (have && j == 15)
                    {
                      if (idx < 16)
                        {
#pragma unroll
                          for (int c = 0; c < nc; ++c)
                            racc[c] += rv[c];
                        }
                    }
                }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
              // the placeholder goes in behind the pairs of its entry (every pair lane skips a constrained row component, so
              // these accumulators receive placeholders only -- in list order)
              if (FULL && g == gg && have && j == 15 && dslot != 0xff)
                {
#pragma unroll
                  for (int c = 0; c < nc; ++c)
                    if ((dmask >> c) & 1u)
                      add_value(dslot, c < 3 ? 4 * c : 12, dv[c]);
                }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
      // ---- the row's sums into the outputs (behind the plain classes: stream order)
      if (FULL && fast)
        for (int n = lane; n < deg * KCMP; n += 64)
          {
            const int slot = n / KCMP, t = n - slot * KCMP;
            const int c = t < 9 ? t / 3 : 3, d = t < 9 ? t - 3 * c : t - 9;
            const double x = acc[n];
            if (x != 0.0)
              *val_ptr<dim>(v, vals, P, c, slot, d) += x;
          }
      if (lane < nc)
        {
          const int c = lane;
          const long long di = dof_index<dim>(v, P, c);
          const bool con = (fP >> c) & 1u;
          const double x = racc[c];
          if (!con)
            res_pde[di] += x;
          if (residual_only && (!con || !total_via_update))
            res_tot[di] += x;
        }
    }

    // ------------------------------------------------------------ small kernels
    template <int dim>
    __global__ void k_state_set(DevView v, const double *__restrict__ sol, const double *__restrict__ old,
                                const double *__restrict__ oldold)
    {
      const int n = blockIdx.x * blockDim.x + threadIdx.x;
      if (n >= v.n_owned)
        return;
#pragma unroll
      for (int d = 0; d < dim; ++d)
        v.u[d][n] = sol[dof_index<dim>(v, n, d)];
      const long long dp = dof_index<dim>(v, n, dim);
      v.phi[n] = sol[dp];
      if (old) // nullptr: pfm_state_set_solution, the old fields keep their values (kernel argument: uniform branch)
        {
          v.phi_old[n] = old[dp];
          v.phi_oldold[n] = oldold[dp];
        }
    }

    template <int dim>
    __global__ void k_halo_pack(DevView v, const int32_t *__restrict__ nodes, long long n, double *__restrict__ buf)
    {
      const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      if (i >= n)
        return;
      const int node = nodes[i];
      // field-major so both sides stay coalesced
#pragma unroll
      for (int d = 0; d < dim; ++d)
        buf[d * n + i] = v.u[d][node];
      buf[(dim + 0) * n + i] = v.phi[node];
      buf[(dim + 1) * n + i] = v.phi_old[node];
      buf[(dim + 2) * n + i] = v.phi_oldold[node];
    }

    template <int dim>
    __global__ void k_halo_unpack(DevView v, const int32_t *__restrict__ nodes, long long n,
                                  const double *__restrict__ buf)
    {
      const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      if (i >= n)
        return;
      const int node = nodes[i];
#pragma unroll
      for (int d = 0; d < dim; ++d)
        v.u[d][node] = buf[d * n + i];
      v.phi[node] = buf[(dim + 0) * n + i];
      v.phi_old[node] = buf[(dim + 1) * n + i];
      v.phi_oldold[node] = buf[(dim + 2) * n + i];
    }

    // all peers in one launch: entry j of the concatenated list belongs to the peer k with ptr[k] <= j < ptr[k+1]
    template <int dim, bool UNPACK>
    __global__ void k_halo_all(DevView v, const int32_t *__restrict__ nodes, const long long *__restrict__ ptr, int n_peers,
                               long long n_total, double *__restrict__ buf)
    {
      const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      if (j >= n_total)
        return;
      int lo = 0, hi = n_peers; // largest k with ptr[k] <= j
      while (hi - lo > 1)
        {
          const int mid = (lo + hi) >> 1;
          if (ptr[mid] <= j)
            lo = mid;
          else
            hi = mid;
        }
      const long long base = ptr[lo], n = ptr[lo + 1] - base, i = j - base;
      double *b = buf + (dim + 3) * base; // this peer's message, field-major
      const int node = nodes[j];
      if constexpr (UNPACK)
        {
#pragma unroll
          for (int d = 0; d < dim; ++d)
            v.u[d][node] = b[d * n + i];
          v.phi[node] = b[(dim + 0) * n + i];
          v.phi_old[node] = b[(dim + 1) * n + i];
          v.phi_oldold[node] = b[(dim + 2) * n + i];
        }
      else
        {
#pragma unroll
          for (int d = 0; d < dim; ++d)
            b[d * n + i] = v.u[d][node];
          b[(dim + 0) * n + i] = v.phi[node];
          b[(dim + 1) * n + i] = v.phi_old[node];
          b[(dim + 2) * n + i] = v.phi_oldold[node];
        }
    }

    // slot table of the general family: cslot[cell][a][b] = position of vertex b's node in the row of vertex a's node
    // (0xff: a is not an owned node, or b is not in the row).  thread <-> (cell, a)
    template <int dim>
    __global__ void k_build_cslot(DevView v, uint8_t *__restrict__ cslot)
    {
      constexpr int nv = 1 << dim;
      const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      if (idx >= v.n_cells * nv)
        return;
      const long long cell = idx / nv;
      const int a = (int)(idx % nv);
      const int A = v.conn[(long long)a * v.n_cells + cell];
      uint8_t *out = cslot + idx * nv;
      if (A >= v.n_owned)
        {
#pragma unroll
          for (int b = 0; b < nv; ++b)
            out[b] = 0xff;
          return;
        }
      const long long lo = v.nadj_ptr[A], hi = v.nadj_ptr[A + 1];
#pragma unroll
      for (int b = 0; b < nv; ++b)
        {
          const int B = v.conn[(long long)b * v.n_cells + cell];
          long long k = lo;
          while (k < hi && v.nadj[k] != B)
            ++k;
          out[b] = k < hi ? (uint8_t)(k - lo) : (uint8_t)0xff;
        }
    }

    // the same for the cells at hanging vertices, per pair of parents (DevView::cslot_h).  thread <-> (cell, a, r)
    template <int dim>
    __global__ void k_build_cslot_h(DevView v, uint8_t *__restrict__ cslot_h)
    {
      constexpr int nv = 1 << dim, MPH = dim == 3 ? 4 : 2;
      const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      if (idx >= v.n_cells * nv * MPH)
        return;
      const long long cell = idx / (nv * MPH);
      const int hc = v.hcell[cell];
      if (hc < 0)
        return;
      const int a = (int)((idx / MPH) % nv), r = (int)(idx % MPH);
      const int A = v.conn[(long long)a * v.n_cells + cell];
      const int kA = v.hn_index[A];
      int P = -1;
      if (kA < 0)
        P = r == 0 ? A : -1;
      else if (v.hn_ptr[kA] + r < v.hn_ptr[kA + 1])
        P = v.hn_parents[v.hn_ptr[kA] + r];
      uint8_t *out = cslot_h + ((long long)hc * nv * MPH + a * MPH + r) * (nv * MPH);
      for (int b = 0; b < nv; ++b)
        {
          const int B = v.conn[(long long)b * v.n_cells + cell];
          const int kB = v.hn_index[B];
          for (int s = 0; s < MPH; ++s)
            {
              int Q = -1;
              if (kB < 0)
                Q = s == 0 ? B : -1;
              else if (v.hn_ptr[kB] + s < v.hn_ptr[kB + 1])
                Q = v.hn_parents[v.hn_ptr[kB] + s];
              uint8_t slot = 0xff;
              if (P >= 0 && P < v.n_owned && Q >= 0)
                {
                  const long long lo = v.nadj_ptr[P], hi = v.nadj_ptr[P + 1];
                  long long k = lo;
                  while (k < hi && v.nadj[k] != Q)
                    ++k;
                  if (k < hi)
                    slot = (uint8_t)(k - lo);
                }
              out[b * MPH + s] = slot;
            }
        }
    }

    // DevView::cres, one record per 3-D cell at a hanging vertex: the distinct constraint-resolved nodes of the cell (a vertex
    // that does not hang is its own, a hanging one brings its parents), their number R (0xff: more than 16), and slot[i][j] =
    // position of node j in the row of node i (0xff: row not owned / not in the row).  16 threads per cell: thread i fills row i.
    __global__ void k_build_cres(DevView v, uint8_t *__restrict__ cres)
    {
      const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      const long long cell = idx >> 4;
      if (cell >= v.n_cells)
        return;
      const int hc = v.hcell[cell], i = (int)(idx & 15);
      if (hc < 0)
        return;
      int node[16], R = 0;
      bool over = false;
      for (int a = 0; a < 8; ++a)
        {
          const int A = v.conn[(long long)a * v.n_cells + cell];
          const int kA = v.hn_index[A];
          const long long rb = kA < 0 ? 0 : v.hn_ptr[kA], re = kA < 0 ? 1 : v.hn_ptr[kA + 1];
          for (long long r = rb; r < re; ++r)
            {
              const int P = kA < 0 ? A : v.hn_parents[r];
              bool known = false;
              for (int t = 0; t < R; ++t)
                known = known || node[t] == P;
              if (known)
                continue;
              if (R == 16)
                over = true;
              else
                node[R++] = P;
            }
        }
      uint8_t *rec = cres + (long long)hc * PFM_CRES_BYTES;
      if (i == 0)
        {
          for (int t = 0; t < 16; ++t)
            reinterpret_cast<int32_t *>(rec)[t] = t < R ? node[t] : -1;
          rec[PFM_CRES_R] = over ? (uint8_t)0xff : (uint8_t)R;
          for (int a = 0; a < 8; ++a)
            {
              const int A = v.conn[(long long)a * v.n_cells + cell];
              reinterpret_cast<int32_t *>(rec + PFM_CRES_HV)[a] = v.hn_index[A] >= 0 ? A : -1;
            }
          *reinterpret_cast<int32_t *>(rec + PFM_CRES_CELL) = (int32_t)cell;
        }
      uint8_t *out = rec + PFM_CRES_SLOT + 16 * i;
      for (int j = 0; j < 16; ++j)
        out[j] = 0xff;
      if (i >= R || over || node[i] >= v.n_owned)
        return;
      const long long lo = v.nadj_ptr[node[i]], hi = v.nadj_ptr[node[i] + 1];
      for (long long k = lo; k < hi; ++k)
        {
          const int q = v.nadj[k];
          for (int j = 0; j < R; ++j)
            if (node[j] == q)
              out[j] = (uint8_t)(k - lo);
        }
    }

    // raises PFM_ERR_NONFINITE in the context's status word if any of the n values is NaN or +-Inf
    __global__ void k_check_finite(const double *__restrict__ x, long long n, int *__restrict__ status)
    {
      bool bad = false;
      for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        bad = bad || !isfinite(x[i]);
      if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0)
        atomicMax(status, (int)PFM_ERR_NONFINITE);
    }

    bool g_tables_ready[16] = {};

    int ensure_tables()
    {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess)
        return PFM_ERR_HIP;
      if (dev < 16 && g_tables_ready[dev])
        return PFM_OK;
      const RefTables t = make_ref_tables();
      if (hipMemcpyToSymbol(HIP_SYMBOL(c_ref), &t, sizeof(t)) != hipSuccess)
        return PFM_ERR_HIP;
      if (dev < 16)
        g_tables_ready[dev] = true;
      return PFM_OK;
    }

    inline int check_launch() { return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP; }
  } // namespace

  int launch_state_set(const DevView &v, const double *sol, const double *old, const double *oldold,
                       hipStream_t s)
  {
    if (v.n_owned == 0)
      return PFM_OK;
    const int bs = 256, nb = (v.n_owned + bs - 1) / bs;
    if (v.dim == 2)
      hipLaunchKernelGGL(k_state_set<2>, dim3(nb), dim3(bs), 0, s, v, sol, old, oldold);
    else
      hipLaunchKernelGGL(k_state_set<3>, dim3(nb), dim3(bs), 0, s, v, sol, old, oldold);
    return check_launch();
  }

  namespace
  {
    __global__ void k_lattice_masks(uint32_t *__restrict__ mask, int NX, int NY, int NZ, int dim)
    {
      const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      if (n >= (long long)NX * NY * NZ)
        return;
      const int i = (int)(n % NX), j = (int)((n / NX) % NY), k = (int)(n / ((long long)NX * NY));
      const int no = dim == 3 ? 27 : 9;
      uint32_t m = 0;
      for (int o = 0; o < no; ++o)
        {
          const int ii = i + (o % 3) - 1, jj = j + ((o / 3) % 3) - 1, kk = k + (dim == 3 ? (o / 9) - 1 : 0);
          if (ii >= 0 && ii < NX && jj >= 0 && jj < NY && kk >= 0 && kk < NZ)
            m |= 1u << o;
        }
      mask[n] = m;
    }
  } // namespace
  int launch_lattice_masks(uint32_t *d_mask, int NX, int NY, int NZ, int dim, hipStream_t s)
  {
    const long long n = (long long)NX * NY * NZ;
    if (n == 0)
      return PFM_OK;
    hipLaunchKernelGGL(k_lattice_masks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_mask, NX, NY, NZ, dim);
    return check_launch();
  }

  int launch_check_finite(const DevView &v, const double *d, int64_t n, hipStream_t s)
  {
    if (n <= 0)
      return PFM_OK;
    const unsigned nb = (unsigned)std::min<int64_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_check_finite, dim3(nb), dim3(256), 0, s, d, (long long)n, v.status);
    return check_launch();
  }

  namespace
  {
    // out[a * n + i] = in[i * w + a]: AoS (host order) -> SoA (device order) of the mesh tables
    template <class T>
    __global__ void k_aos_to_soa(const T *__restrict__ in, T *__restrict__ out, long long n, int w)
    {
      const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      if (idx >= n * w)
        return;
      const long long i = idx / w;
      const int a = (int)(idx - i * w);
      out[(long long)a * n + i] = in[idx];
    }
  } // namespace
  int launch_aos_to_soa_i32(const int32_t *d_in, int32_t *d_out, long long n, int w, hipStream_t s)
  {
    if (n * w == 0)
      return PFM_OK;
    hipLaunchKernelGGL(k_aos_to_soa<int32_t>, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, s, d_in, d_out, n, w);
    return check_launch();
  }
  int launch_aos_to_soa_f64(const double *d_in, double *d_out, long long n, int w, hipStream_t s)
  {
    if (n * w == 0)
      return PFM_OK;
    hipLaunchKernelGGL(k_aos_to_soa<double>, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, s, d_in, d_out, n, w);
    return check_launch();
  }

  int launch_build_cslot(const DevView &v, hipStream_t s)
  {
    const int nv = 1 << v.dim;
    const long long n = v.n_cells * nv;
    if (n == 0)
      return PFM_OK;
    const int bs = 256;
    const unsigned nb = (unsigned)((n + bs - 1) / bs);
    if (v.dim == 2)
      hipLaunchKernelGGL(k_build_cslot<2>, dim3(nb), dim3(bs), 0, s, v, const_cast<uint8_t *>(v.cslot));
    else
      hipLaunchKernelGGL(k_build_cslot<3>, dim3(nb), dim3(bs), 0, s, v, const_cast<uint8_t *>(v.cslot));
    if (v.cslot_h)
      {
        const long long nh = n * (v.dim == 3 ? 4 : 2);
        const unsigned nbh = (unsigned)((nh + bs - 1) / bs);
        if (v.dim == 2)
          hipLaunchKernelGGL(k_build_cslot_h<2>, dim3(nbh), dim3(bs), 0, s, v, const_cast<uint8_t *>(v.cslot_h));
        else
          hipLaunchKernelGGL(k_build_cslot_h<3>, dim3(nbh), dim3(bs), 0, s, v, const_cast<uint8_t *>(v.cslot_h));
      }
    if (v.cres && v.dim == 3)
      hipLaunchKernelGGL(k_build_cres, dim3((unsigned)((v.n_cells * 16 + bs - 1) / bs)), dim3(bs), 0, s, v, const_cast<uint8_t *>(v.cres));
    return check_launch();
  }

  int launch_hanging_gather(const DevView &v, const pfm_params &p, int residual_only, double *const *d_values, double *d_res_pde, double *d_res_tot,
                            const int32_t *rows, const long long *ptr, const HgEntry *list, int64_t n_rows, hipStream_t s)
  {
    if (n_rows == 0)
      return PFM_OK;
    Vals vals{};
    if (!residual_only)
      for (int b = 0; b < (v.layout == PFM_LAYOUT_BLOCKED ? 4 : 1); ++b)
        vals.b[b] = d_values[b];
    const dim3 grid((unsigned)((n_rows + 3) / 4)), block(256);
    if (residual_only)
      hipLaunchKernelGGL(k_hanging_gather<false>, grid, block, 0, s, v, p, vals, d_res_pde, d_res_tot, residual_only, rows, ptr, list, (long long)n_rows);
    else
      hipLaunchKernelGGL(k_hanging_gather<true>, grid, block, 0, s, v, p, vals, d_res_pde, d_res_tot, residual_only, rows, ptr, list, (long long)n_rows);
    return check_launch();
  }

  int launch_halo_pack(const DevView &v, const int32_t *nodes, int64_t n, double *buf, hipStream_t s)
  {
    if (n == 0)
      return PFM_OK;
    const int bs = 256;
    const unsigned nb = (unsigned)((n + bs - 1) / bs);
    if (v.dim == 2)
      hipLaunchKernelGGL(k_halo_pack<2>, dim3(nb), dim3(bs), 0, s, v, nodes, (long long)n, buf);
    else
      hipLaunchKernelGGL(k_halo_pack<3>, dim3(nb), dim3(bs), 0, s, v, nodes, (long long)n, buf);
    return check_launch();
  }

  int launch_halo_unpack(const DevView &v, const int32_t *nodes, int64_t n, const double *buf, hipStream_t s)
  {
    if (n == 0)
      return PFM_OK;
    const int bs = 256;
    const unsigned nb = (unsigned)((n + bs - 1) / bs);
    if (v.dim == 2)
      hipLaunchKernelGGL(k_halo_unpack<2>, dim3(nb), dim3(bs), 0, s, v, nodes, (long long)n, buf);
    else
      hipLaunchKernelGGL(k_halo_unpack<3>, dim3(nb), dim3(bs), 0, s, v, nodes, (long long)n, buf);
    return check_launch();
  }

  int launch_halo_all(const DevView &v, const int32_t *nodes, const long long *ptr, int n_peers, int64_t n_total, double *buf,
                      int unpack, hipStream_t s)
  {
    if (n_total == 0)
      return PFM_OK;
    const int bs = 256;
    const unsigned nb = (unsigned)((n_total + bs - 1) / bs);
    if (v.dim == 2)
      {
        if (unpack)
          hipLaunchKernelGGL((k_halo_all<2, true>), dim3(nb), dim3(bs), 0, s, v, nodes, ptr, n_peers, (long long)n_total, buf);
        else
          hipLaunchKernelGGL((k_halo_all<2, false>), dim3(nb), dim3(bs), 0, s, v, nodes, ptr, n_peers, (long long)n_total, buf);
      }
    else
      {
        if (unpack)
          hipLaunchKernelGGL((k_halo_all<3, true>), dim3(nb), dim3(bs), 0, s, v, nodes, ptr, n_peers, (long long)n_total, buf);
        else
          hipLaunchKernelGGL((k_halo_all<3, false>), dim3(nb), dim3(bs), 0, s, v, nodes, ptr, n_peers, (long long)n_total, buf);
      }
    return check_launch();
  }

  namespace
  {
    // CSR slots of the 9 lattice offsets of every regular row, from the current order of the node-graph rows:
    // thread <-> (block, owned position)
    __global__ void k_patch_slots(DevView v, unsigned long long *__restrict__ slots, int n_blocks)
    {
      const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      if (idx >= (long long)n_blocks * 49)
        return;
      const int blk = (int)(idx / 49), nl = (int)(idx % 49);
      const int hx = nl % 7 + 1, hy = nl / 7 + 1;
      const int32_t *bn = v.patch_nodes + (long long)blk * 81;
      const int node = bn[hx + 9 * hy];
      if (node < 0 || node >= v.n_owned || !v.row_patch[node])
        return;
      const long long off = v.nadj_ptr[node];
      const int deg = (int)(v.nadj_ptr[node + 1] - off);
      unsigned long long packed = 0ull;
      for (int o = 0; o < 9; ++o)
        {
          const int nb = bn[(hx + o % 3 - 1) + 9 * (hy + o / 3 - 1)];
          int sl = 15;
          for (int k = 0; k < deg; ++k)
            if (v.nadj[off + k] == nb)
              sl = k;
          if (sl == 15)
            atomicMax(v.status, (int)PFM_ERR_INTERNAL); // a regular row must hold its 9 lattice neighbours
          packed |= (unsigned long long)sl << (4 * o);
        }
      slots[node] = packed;
    }
  } // namespace

  namespace
  {
    // overlay assemblies: only the rows of the general family are added to and need zeros (one wave per row)
    __global__ void k_zero_rows(DevView v, Vals vals, const int32_t *__restrict__ rows, int n_rows)
    {
      const int w = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
      if (w >= n_rows)
        return;
      const int node = rows[w];
      const long long off = v.nadj_ptr[node], deg = v.nadj_ptr[node + 1] - off;
      const int nc = v.dim + 1, dim = v.dim;
      if (v.layout == PFM_LAYOUT_INTERLEAVED)
        {
          for (long long i = lane; i < nc * nc * deg; i += 64)
            vals.b[0][nc * nc * off + i] = 0.0;
          return;
        }
      for (long long i = lane; i < dim * dim * deg; i += 64)
        vals.b[0][dim * dim * off + i] = 0.0;
      for (long long i = lane; i < dim * deg; i += 64)
        {
          vals.b[1][dim * off + i] = 0.0;
          vals.b[2][dim * off + i] = 0.0;
        }
      for (long long i = lane; i < deg; i += 64)
        vals.b[3][off + i] = 0.0;
    }
  } // namespace

  int launch_zero_rows(const DevView &v, double *const *d_values, const int32_t *rows, int n_rows, hipStream_t s)
  {
    if (n_rows <= 0)
      return PFM_OK;
    Vals vals{};
    for (int b = 0; b < (v.layout == PFM_LAYOUT_BLOCKED ? 4 : 1); ++b)
      vals.b[b] = d_values[b];
    hipLaunchKernelGGL(k_zero_rows, dim3((unsigned)(((long long)n_rows * 64 + 255) / 256)), dim3(256), 0, s, v, vals, rows, n_rows);
    return check_launch();
  }

  int launch_patch_slots(const DevView &v, unsigned long long *d_slots, int n_blocks, hipStream_t s)
  {
    if (n_blocks <= 0)
      return PFM_OK;
    const long long n = (long long)n_blocks * 49;
    hipLaunchKernelGGL(k_patch_slots, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, d_slots, n_blocks);
    return check_launch();
  }

  int launch_assemble_patches(const DevView &v, const pfm_params &p, int residual_only, double *const *d_values, double *res_pde,
                              double *res_tot, int n_blocks, hipStream_t s)
  {
    int rc = ensure_tables();
    if (rc)
      return rc;
    if (n_blocks <= 0)
      return PFM_OK;
    if (v.dim != 2)
      return PFM_ERR_UNSUPPORTED;
    Vals vals{};
    if (!residual_only)
      for (int b = 0; b < (v.layout == PFM_LAYOUT_BLOCKED ? 4 : 1); ++b)
        vals.b[b] = d_values[b];
    const bool split = (p.decompose_stress_matrix > 0 && p.timestep_number > 0);
    const dim3 grid((unsigned)n_blocks), block(256);
#define PFM_PATCH(FULLV, SPLITV)                                                                                              \
  hipLaunchKernelGGL((k_assemble_general<2, FULLV, SPLITV, false, false, true>), grid, block, 0, s, v, p, vals, res_pde, res_tot, \
                     residual_only, 0LL, 0LL)
    if (residual_only)
      {
        if (split)
          PFM_PATCH(false, true);
        else
          PFM_PATCH(false, false);
      }
    else
      {
        if (split)
          PFM_PATCH(true, true);
        else
          PFM_PATCH(true, false);
      }
#undef PFM_PATCH
    return check_launch();
  }

  int launch_assemble_general(const DevView &v, const pfm_params &p, int residual_only,
                              double *const *d_values, double *res_pde, double *res_tot, hipStream_t s,
                              const std::vector<long long> &color_ptr, hipStream_t s_atomic)
  {
    int rc = ensure_tables();
    if (rc)
      return rc;
    if (v.n_cells == 0)
      return PFM_OK;
    Vals vals{};
    if (!residual_only)
      for (int b = 0; b < (v.layout == PFM_LAYOUT_BLOCKED ? 4 : 1); ++b)
        vals.b[b] = d_values[b];
    const bool split = (p.decompose_stress_matrix > 0 && p.timestep_number > 0);
    // The reference gates the split on decompose_stress_matrix only (cracks.cc:2294); a
    // non-zero decompose_stress_rhs without it multiplies a zero stress_term_minus.
    if (v.dim == 3 && split)
      return PFM_ERR_UNSUPPORTED;
    // cells per workgroup: 32 lanes per hex for the 3-D Jacobian (k_assemble_general: Q3), one lane per vertex otherwise
    const int nv = 1 << v.dim, cpb = (v.dim == 3 && !residual_only) ? 8 : 256 / nv;
    const int n_classes = (int)color_ptr.size() - 1;
    // one launch per colour class, in class order (stream order = the summation order of a row: reproducible);
    // the last class (cells with hanging vertices) adds atomically
    // s_atomic: the caller has forked it off s behind the zeroing of the outputs and joins it afterwards; the atomic class
    // goes first, on that stream (needs DevView::cell_ring)
    hipStream_t s_main = s;
    for (int kk = 0; kk < n_classes; ++kk)
      {
        // (DevView::hs_K: that class only writes its scratch -- it may run next to anything)
        const bool side = s_atomic && (v.cell_ring || v.hs_K);
        const int k = side ? (kk == 0 ? n_classes - 1 : kk - 1) : kk;
        const long long c0 = color_ptr[k], cn = color_ptr[k + 1] - c0;
        if (cn == 0)
          continue;
        const bool atomic = k == n_classes - 1;
        s = (atomic && side) ? s_atomic : s_main;
        const dim3 grid((unsigned)((cn + cpb - 1) / cpb)), block(256);
#define PFM_LAUNCH(DIM, FULLV, SPLITV)                                                                                       \
  do                                                                                                                         \
    {                                                                                                                        \
      if (atomic && DIM == 3 && !(SPLITV) && v.hs_K)                                                                         \
        hipLaunchKernelGGL((k_assemble_general<DIM, FULLV, false, true, false, false, DIM == 3>), grid, block, 0, s, v, p,    \
                           vals, res_pde, res_tot, residual_only, c0, cn);                                                   \
      else if (atomic)                                                                                                       \
        hipLaunchKernelGGL((k_assemble_general<DIM, FULLV, SPLITV, true>), grid, block, 0, s, v, p, vals, res_pde, res_tot,  \
                           residual_only, c0, cn);                                                                           \
      else if (v.cell_ring)                                                                                                  \
        hipLaunchKernelGGL((k_assemble_general<DIM, FULLV, SPLITV, false, true>), grid, block, 0, s, v, p, vals, res_pde, \
                           res_tot, residual_only, c0, cn);                                                                  \
      else                                                                                                                   \
        hipLaunchKernelGGL((k_assemble_general<DIM, FULLV, SPLITV, false>), grid, block, 0, s, v, p, vals, res_pde, res_tot, \
                           residual_only, c0, cn);                                                                           \
    }                                                                                                                        \
  while (0)
        if (v.dim == 2)
          {
            if (residual_only)
              {
                if (split)
                  PFM_LAUNCH(2, false, true);
                else
                  PFM_LAUNCH(2, false, false);
              }
            else
              {
                if (split)
                  PFM_LAUNCH(2, true, true);
                else
                  PFM_LAUNCH(2, true, false);
              }
          }
        else
          {
            if (residual_only)
              PFM_LAUNCH(3, false, false);
            else
              {
                PFM_LAUNCH(3, true, false);
              }
          }
#undef PFM_LAUNCH
      }
    return check_launch();
  }
} // namespace pfm
