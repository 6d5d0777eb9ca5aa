// pfm_graph.hip — node graph of a general (non-lattice) mesh on the device.
//
// The rows of the matrix pattern (cracks.cc:1644-1654: make_sparsity_pattern through the constraints) are, at node
// level, "all nodes that share a constraint-resolved cell with the row's node": a hanging vertex stands for its parents.
// pfm_ctx_create needs them after every refine_mesh (cracks.cc:4148); the host build (incidence lists + one sort per
// node) was 12.6 of the 18 ms of a context rebuild on the 2.7e5-cell stand-in of BASELINE config 5 and 7.5 ms threaded.
// Here: count the cells incident to every owned node (atomics), scan, fill the incidence lists, then one thread per owned
// node merges the resolved vertices of its cells into a sorted duplicate-free list in private memory -- once for the row
// length, once (after a scan of the lengths) to write the row.  The incidence lists are in arbitrary order (atomics); the
// rows are sorted, so the result is deterministic: ascending local node id, ghost columns last, as the host build.
#include "pfm_internal.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <mutex>
#include <vector>

namespace pfm
{
  namespace
  {
    struct ScratchEntry
    {
      void *p = nullptr;
      size_t bytes = 0;
      int device = 0;
      bool in_use = false;
    };
    std::mutex g_scratch_mx;
    std::vector<ScratchEntry> g_scratch;
    constexpr size_t SCRATCH_KEEP_ONE = (size_t)32 << 20, SCRATCH_KEEP_ALL = (size_t)64 << 20;
  } // namespace

  hipError_t scratch_acquire(void **p, size_t bytes)
  {
    *p = nullptr;
    bytes = std::max<size_t>(bytes, 16);
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
      std::lock_guard<std::mutex> lk(g_scratch_mx);
      ScratchEntry *best = nullptr;
      size_t kept = 0;
      for (auto &e : g_scratch)
        {
          kept += e.bytes;
          if (!e.in_use && e.device == dev && e.bytes >= bytes && e.bytes <= std::max<size_t>(4 * bytes, (size_t)1 << 20) && (!best || e.bytes < best->bytes))
            best = &e;
        }
      if (best)
        {
          best->in_use = true;
          *p = best->p;
          return hipSuccess;
        }
      if (bytes <= SCRATCH_KEEP_ONE && kept + bytes <= SCRATCH_KEEP_ALL && g_scratch.size() < 64)
        {
          void *q = nullptr;
          const hipError_t e = hipMalloc(&q, bytes);
          if (e != hipSuccess)
            return e;
          g_scratch.push_back(ScratchEntry{q, bytes, dev, true});
          *p = q;
          return hipSuccess;
        }
    }
    return hipMalloc(p, bytes);
  }

  void scratch_release(void *p)
  {
    if (!p)
      return;
    {
      std::lock_guard<std::mutex> lk(g_scratch_mx);
      for (auto &e : g_scratch)
        if (e.p == p)
          {
            e.in_use = false;
            return;
          }
    }
    (void)hipFree(p);
  }

  namespace
  {
    constexpr int MAX_ROW = 254; // the slot tables of the general family hold uint8 positions

    struct GraphIn
    {
      const int32_t *cells; // [NC][nv], host order
      long long NC;
      int nv;
      int32_t NO;
      const int32_t *hn_index; // [N] or nullptr: index into the hanging table, -1 = not hanging
      const long long *hn_ptr;
      const int32_t *hn_parents;
    };

    template <class F>
    __device__ __forceinline__ void for_each_resolved(const GraphIn &g, long long cell, F &&fn)
    {
      for (int a = 0; a < g.nv; ++a)
        {
          const int32_t n = g.cells[cell * g.nv + a];
          fn(n);
          const int32_t k = g.hn_index ? g.hn_index[n] : -1;
          if (k >= 0)
            for (long long j = g.hn_ptr[k]; j < g.hn_ptr[k + 1]; ++j)
              fn(g.hn_parents[j]);
        }
    }

    __global__ void k_graph_count(GraphIn g, int *__restrict__ cnt)
    {
      const long long cell = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      if (cell >= g.NC)
        return;
      for_each_resolved(g, cell, [&](int32_t n) {
        if (n < g.NO)
          atomicAdd(&cnt[n], 1);
      });
    }

    __global__ void k_graph_fill(GraphIn g, const int *__restrict__ inc_ptr, int *__restrict__ fill, int *__restrict__ inc)
    {
      const long long cell = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      if (cell >= g.NC)
        return;
      for_each_resolved(g, cell, [&](int32_t n) {
        if (n < g.NO)
          inc[inc_ptr[n] + atomicAdd(&fill[n], 1)] = (int)cell;
      });
    }

    // sorted duplicate-free neighbour list of node n in private memory; returns its length (MAX_ROW + 1: too long)
    __device__ __forceinline__ int gather_row(const GraphIn &g, const int *__restrict__ inc_ptr, const int *__restrict__ inc, int32_t n,
                                              int32_t (&row)[MAX_ROW + 1])
    {
      int len = 0;
      bool over = false;
      for (int k = inc_ptr[n]; k < inc_ptr[n + 1]; ++k)
        for_each_resolved(g, inc[k], [&](int32_t q) {
          // position of q in the sorted list (binary search), insert if absent
          int lo = 0, hi = len;
          while (lo < hi)
            {
              const int mid = (lo + hi) >> 1;
              if (row[mid] < q)
                lo = mid + 1;
              else
                hi = mid;
            }
          if (lo < len && row[lo] == q)
            return;
          if (len > MAX_ROW)
            {
              over = true;
              return;
            }
          for (int i = len; i > lo; --i)
            row[i] = row[i - 1];
          row[lo] = q;
          ++len;
        });
      return over ? MAX_ROW + 1 : len;
    }

    __global__ void k_graph_degree(GraphIn g, const int *__restrict__ inc_ptr, const int *__restrict__ inc, long long *__restrict__ deg,
                                   int *__restrict__ status)
    {
      const int32_t n = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
      if (n >= g.NO)
        return;
      int32_t row[MAX_ROW + 1];
      const int len = gather_row(g, inc_ptr, inc, n, row);
      if (len > MAX_ROW)
        atomicMax(status, 1);
      deg[n] = len;
    }

    __global__ void k_graph_rows(GraphIn g, const int *__restrict__ inc_ptr, const int *__restrict__ inc, const long long *__restrict__ nadj_ptr,
                                 int32_t *__restrict__ nadj)
    {
      const int32_t n = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
      if (n >= g.NO)
        return;
      int32_t row[MAX_ROW + 1];
      const int len = gather_row(g, inc_ptr, inc, n, row);
      int32_t *out = nadj + nadj_ptr[n];
      for (int i = 0; i < len && i <= MAX_ROW; ++i)
        out[i] = row[i];
    }
  } // namespace

  // Phase 1: everything up to the row pointers.  d_nadj_ptr [NO + 1] receives the exclusive scan of the row lengths; the
  // scratch (incidence lists) is returned for phase 2 and freed by graph_build_finish.  All launches on stream s; the
  // function returns after reading back the total (one 8-byte copy).  status: 0 ok, 1 = a row longer than 254.
  int graph_build_begin(const int32_t *d_cells, long long NC, int nv, int32_t NO, const int32_t *d_hn_index, const long long *d_hn_ptr,
                        const int32_t *d_hn_parents, long long *d_nadj_ptr, GraphScratch &sc, long long &total, hipStream_t s)
  {
    sc = GraphScratch{};
    total = 0;
    if (NO == 0)
      return hipMemsetAsync(d_nadj_ptr, 0, sizeof(long long), s) == hipSuccess ? PFM_OK : PFM_ERR_HIP;
    GraphIn g{d_cells, NC, nv, NO, d_hn_index, d_hn_ptr, d_hn_parents};
    int *cnt = nullptr, *inc_ptr = nullptr, *inc = nullptr, *status = nullptr;
    long long *deg = nullptr;
    void *tmp = nullptr;
    auto fail = [&]() {
      for (void *q : {(void *)cnt, (void *)inc_ptr, (void *)inc, (void *)status, (void *)deg, tmp})
        if (q)
          scratch_release(q);
      return PFM_ERR_HIP;
    };
    if (scratch_acquire((void **)&cnt, sizeof(int) * ((size_t)NO + 1)) != hipSuccess || scratch_acquire((void **)&inc_ptr, sizeof(int) * ((size_t)NO + 1)) != hipSuccess ||
        scratch_acquire((void **)&deg, sizeof(long long) * ((size_t)NO + 1)) != hipSuccess || scratch_acquire((void **)&status, sizeof(int)) != hipSuccess)
      return fail();
    if (hipMemsetAsync(cnt, 0, sizeof(int) * ((size_t)NO + 1), s) != hipSuccess || hipMemsetAsync(status, 0, sizeof(int), s) != hipSuccess ||
        hipMemsetAsync(deg, 0, sizeof(long long) * ((size_t)NO + 1), s) != hipSuccess)
      return fail();
    const int bs = 256;
    const unsigned nbc = (unsigned)((NC + bs - 1) / bs), nbn = (unsigned)((NO + bs - 1) / bs);
    if (NC > 0)
      hipLaunchKernelGGL(k_graph_count, dim3(nbc), dim3(bs), 0, s, g, cnt);
    size_t tb1 = 0, tb2 = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb1, cnt, inc_ptr, NO + 1, s);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, deg, d_nadj_ptr, NO + 1, s);
    const size_t tb = std::max(tb1, tb2);
    if (scratch_acquire(&tmp, std::max<size_t>(tb, 16)) != hipSuccess)
      return fail();
    size_t tbb = tb;
    if (hipcub::DeviceScan::ExclusiveSum(tmp, tbb, cnt, inc_ptr, NO + 1, s) != hipSuccess)
      return fail();
    int n_inc = 0;
    if (hipMemcpyAsync(&n_inc, inc_ptr + NO, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return fail();
    if (scratch_acquire((void **)&inc, sizeof(int) * (size_t)std::max(n_inc, 1)) != hipSuccess)
      return fail();
    if (hipMemsetAsync(cnt, 0, sizeof(int) * ((size_t)NO + 1), s) != hipSuccess)
      return fail();
    if (NC > 0)
      hipLaunchKernelGGL(k_graph_fill, dim3(nbc), dim3(bs), 0, s, g, inc_ptr, cnt, inc);
    hipLaunchKernelGGL(k_graph_degree, dim3(nbn), dim3(bs), 0, s, g, inc_ptr, inc, deg, status);
    tbb = tb;
    if (hipcub::DeviceScan::ExclusiveSum(tmp, tbb, deg, d_nadj_ptr, NO + 1, s) != hipSuccess)
      return fail();
    int st = 0;
    if (hipMemcpyAsync(&total, d_nadj_ptr + NO, sizeof(long long), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(&st, status, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess ||
        hipGetLastError() != hipSuccess)
      return fail();
    scratch_release(cnt);
    scratch_release(deg);
    scratch_release(status);
    scratch_release(tmp);
    sc.inc_ptr = inc_ptr;
    sc.inc = inc;
    if (st != 0)
      {
        graph_build_free(sc);
        return PFM_ERR_UNSUPPORTED; // a node with more than 254 neighbours
      }
    return PFM_OK;
  }

  // Phase 2: the rows into d_nadj (allocated by the caller from the total of phase 1); asynchronous on s
  int graph_build_rows(const int32_t *d_cells, long long NC, int nv, int32_t NO, const int32_t *d_hn_index, const long long *d_hn_ptr,
                       const int32_t *d_hn_parents, const long long *d_nadj_ptr, int32_t *d_nadj, const GraphScratch &sc, hipStream_t s)
  {
    if (NO == 0)
      return PFM_OK;
    GraphIn g{d_cells, NC, nv, NO, d_hn_index, d_hn_ptr, d_hn_parents};
    const int bs = 256;
    hipLaunchKernelGGL(k_graph_rows, dim3((unsigned)((NO + bs - 1) / bs)), dim3(bs), 0, s, g, sc.inc_ptr, sc.inc, d_nadj_ptr, d_nadj);
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }

  void graph_build_free(GraphScratch &sc)
  {
    if (sc.inc_ptr)
      scratch_release(sc.inc_ptr);
    if (sc.inc)
      scratch_release(sc.inc);
    sc = GraphScratch{};
  }

  // ---- uniform box numbered lexicographically (node n at lattice position n): row pointers and colour lists on the device ----
  namespace
  {
    __global__ void k_lattice_degree(long long *__restrict__ deg, int NX, int NY, int NZ)
    {
      const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x, NN = (long long)NX * NY * NZ;
      if (n > NN)
        return;
      long long d = 0;
      if (n < NN)
        {
          const int i = (int)(n % NX), j = (int)((n / NX) % NY), k = (int)(n / ((long long)NX * NY));
          d = (1 + (i > 0) + (i < NX - 1)) * (1 + (j > 0) + (j < NY - 1)) * (1 + (k > 0) + (k < NZ - 1));
        }
      deg[n] = d;
    }
    // colour of a cell = parity of the lattice position of its lower corner (vertex 0 of the cell)
    __global__ void k_lattice_cell_colour(const int32_t *__restrict__ vertex0, const int32_t *__restrict__ box_of_local, long long NC, int NX, int NY,
                                          uint8_t *__restrict__ key, int32_t *__restrict__ cell)
    {
      const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      if (c >= NC)
        return;
      const int32_t n = vertex0[c];
      const long long b = box_of_local ? box_of_local[n] : n;
      const int i = (int)(b % NX), j = (int)((b / NX) % NY), k = (int)(b / ((long long)NX * NY));
      key[c] = (uint8_t)((i & 1) + 2 * (j & 1) + 4 * (k & 1));
      cell[c] = (int32_t)c;
    }
  } // namespace

  // d_ptr [NX NY NZ + 1]: exclusive scan of the row lengths of the lattice node graph (the 1e7 row pointers of a 216^3 box:
  // 6 ms of host scan and an 80 MB upload otherwise)
  int launch_lattice_row_ptr(long long *d_ptr, int NX, int NY, int NZ, hipStream_t s)
  {
    const long long NN = (long long)NX * NY * NZ;
    long long *deg = nullptr;
    void *tmp = nullptr;
    size_t tb = 0;
    auto done = [&](int rc) {
      scratch_release(deg);
      scratch_release(tmp);
      return rc;
    };
    if (scratch_acquire((void **)&deg, sizeof(long long) * (size_t)(NN + 1)) != hipSuccess)
      return done(PFM_ERR_HIP);
    hipLaunchKernelGGL(k_lattice_degree, dim3((unsigned)((NN + 256) / 256)), dim3(256), 0, s, deg, NX, NY, NZ);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, deg, d_ptr, (int)(NN + 1), s);
    if (scratch_acquire(&tmp, std::max<size_t>(tb, 16)) != hipSuccess)
      return done(PFM_ERR_HIP);
    if (hipcub::DeviceScan::ExclusiveSum(tmp, tb, deg, d_ptr, (int)(NN + 1), s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess ||
        hipGetLastError() != hipSuccess)
      return done(PFM_ERR_HIP);
    return done(PFM_OK);
  }

  // d_order [NC]: the cells by colour class (parity of the lower corner), ascending cell number within a class: the list the
  // host's counting sort makes.  d_vertex0: first column of the SoA cell table; d_box_of_local: null when node n sits at
  // lattice position n.
  int launch_lattice_colour_order(int32_t *d_order, const int32_t *d_vertex0, const int32_t *d_box_of_local, long long NC, int NX, int NY, hipStream_t s)
  {
    if (NC == 0)
      return PFM_OK;
    uint8_t *key = nullptr, *key2 = nullptr;
    int32_t *cell = nullptr;
    void *tmp = nullptr;
    size_t tb = 0;
    auto done = [&](int rc) {
      for (void *q : {(void *)key, (void *)key2, (void *)cell, tmp})
        scratch_release(q);
      return rc;
    };
    if (scratch_acquire((void **)&key, (size_t)NC) != hipSuccess || scratch_acquire((void **)&key2, (size_t)NC) != hipSuccess ||
        scratch_acquire((void **)&cell, sizeof(int32_t) * (size_t)NC) != hipSuccess)
      return done(PFM_ERR_HIP);
    hipLaunchKernelGGL(k_lattice_cell_colour, dim3((unsigned)((NC + 255) / 256)), dim3(256), 0, s, d_vertex0, d_box_of_local, NC, NX, NY, key, cell);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, key, key2, cell, d_order, (int)NC, 0, 3, s);
    if (scratch_acquire(&tmp, std::max<size_t>(tb, 16)) != hipSuccess)
      return done(PFM_ERR_HIP);
    if (hipcub::DeviceRadixSort::SortPairs(tmp, tb, key, key2, cell, d_order, (int)NC, 0, 3, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess ||
        hipGetLastError() != hipSuccess)
      return done(PFM_ERR_HIP);
    return done(PFM_OK);
  }

  // bad != 0 afterwards: a marked row does not have `want` entries (2-D overlay: a regular row has the nine nodes of its cells)
  namespace
  {
    __global__ void k_check_row_lengths(const uint8_t *__restrict__ mark, const long long *__restrict__ ptr, int32_t NO, int want, int *__restrict__ bad)
    {
      const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
      if (n < NO && mark[n] && ptr[n + 1] - ptr[n] != want)
        atomicOr(bad, 1);
    }
  } // namespace
  int launch_check_row_lengths(const uint8_t *d_mark, const long long *d_ptr, int32_t NO, int want, int *d_bad, hipStream_t s)
  {
    if (NO > 0)
      hipLaunchKernelGGL(k_check_row_lengths, dim3((unsigned)((NO + 255) / 256)), dim3(256), 0, s, d_mark, d_ptr, NO, want, d_bad);
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }

  // ---- cartesian overlay of a general 3-D mesh (round 5): row tables of one refinement level's lattice --------------------
  // node_at / row_at: node id / row id per lattice point of the level's box (-1: none).  For every row r of the level:
  // nbr_mask[r] = all 27 offsets exist, the row is not in lattice order (bit 31); row_perm[nadj_ptr[r] + o] = CSR slot of
  // the node at lattice offset o in the CURRENT order of the node-graph row (pfm_ctx_create: ascending local id;
  // pfm_pattern_bind: the caller's).  A row that is not a plain 27-neighbour row of the level raises the status word.
  namespace
  {
    __global__ void k_overlay3_rows(const int32_t *__restrict__ node_at, const int32_t *__restrict__ row_at, int NX, int NY, int NZ,
                                    const long long *__restrict__ nadj_ptr, const int32_t *__restrict__ nadj, uint32_t *__restrict__ nbr_mask,
                                    uint8_t *__restrict__ row_perm, int *__restrict__ bad)
    {
      const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
      if (p >= (long long)NX * NY * NZ)
        return;
      const int r = row_at[p];
      if (r < 0)
        return;
      const int i = (int)(p % NX), j = (int)((p / NX) % NY), k = (int)(p / ((long long)NX * NY));
      const long long base = nadj_ptr[r];
      const int deg = (int)(nadj_ptr[r + 1] - base);
      // the offsets that stay inside the level lattice (a node on one of its faces has fewer than 27: the face of a level that
      // reaches the boundary of the domain); every one of them must be a node of the level and sit in the row
      unsigned mask = 0;
      int rank = 0;
      bool ok = true;
      for (int o = 0; o < 27 && ok; ++o)
        {
          const int ii = i + o % 3 - 1, jj = j + (o / 3) % 3 - 1, kk = k + o / 9 - 1;
          if (ii < 0 || ii >= NX || jj < 0 || jj >= NY || kk < 0 || kk >= NZ)
            continue;
          const int q = node_at[ii + (long long)NX * (jj + (long long)NY * kk)];
          int slot = -1;
          for (int t = 0; t < deg; ++t)
            if (nadj[base + t] == q)
              slot = t;
          ok = q >= 0 && slot >= 0 && rank < deg;
          if (ok)
            {
              row_perm[base + rank] = (uint8_t)slot;
              mask |= 1u << o;
              ++rank;
            }
        }
      ok = ok && rank == deg;
      nbr_mask[r] = ok ? (mask | 0x80000000u) : 0u;
      if (!ok)
        atomicAdd(bad, 1);
    }
  } // namespace

  int launch_overlay3_rows(const int32_t *d_node_at, const int32_t *d_row_at, int NX, int NY, int NZ, const long long *d_nadj_ptr,
                           const int32_t *d_nadj, uint32_t *d_nbr_mask, uint8_t *d_row_perm, int *d_bad, hipStream_t s)
  {
    const long long n = (long long)NX * NY * NZ;
    if (n == 0)
      return PFM_OK;
    hipLaunchKernelGGL(k_overlay3_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_node_at, d_row_at, NX, NY, NZ, d_nadj_ptr, d_nadj,
                       d_nbr_mask, d_row_perm, d_bad);
    return hipGetLastError() == hipSuccess ? PFM_OK : PFM_ERR_HIP;
  }
} // namespace pfm
