// pfm_host.cpp — host side of the C ABI (include/pfm_assemble.h): context build
// (node graph, CSR addressing tables, device mirrors), pattern queries, state upload,
// halo registration and the synchronous host-pointer entry point.
//
// What the reference does with deal.II objects before/after the cell loop
// (cracks.cc:2133-2160, 2439-2475) becomes table look-ups prepared here once per
// setup_system() (cracks.cc:1579-1680):
//   * the sparsity pattern of make_sparsity_pattern (cracks.cc:1644-1654) is the node
//     graph of the constraint-resolved mesh (x) full component coupling; its CSR offsets
//     are arithmetic in (node-graph offset, component), so no column search is needed
//     at assembly time;
//   * cell->get_dof_indices + the Trilinos column search (cracks.cc:2439-2463) become a
//     one-byte-per-vertex-pair slot table.
#include "pfm_internal.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>

using namespace pfm;

namespace
{
  struct HipFail
  {
    hipError_t e;
    const char *what;
  };

  template <class T>
  T *dev_alloc(pfm_ctx *c, size_t n)
  {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
      throw HipFail{e, "hipMalloc"};
    c->allocs.push_back(p);
    c->device_bytes += (int64_t)bytes;
    return static_cast<T *>(p);
  }

  template <class T>
  T *dev_upload(pfm_ctx *c, const T *h, size_t n)
  {
    T *d = dev_alloc<T>(c, n);
    if (n)
      {
        hipError_t e = hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess)
          throw HipFail{e, "hipMemcpy H2D"};
      }
    return d;
  }

  int fail(pfm_ctx *c, int code, const std::string &msg)
  {
    if (c)
      c->err = msg;
    return code;
  }

  int hipfail(pfm_ctx *c, hipError_t e, const char *what)
  {
    return fail(c, PFM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  }
  // Lattice of a uniform Cartesian box: node n <-> lattice index box_of_local[n] (x fastest), cells in deal.II
  // vertex order, every lattice cell present exactly once.  false whenever any check fails.
  struct Lattice
  {
    int NX = 0, NY = 0, NZ = 0, nc[3] = {0, 0, 0};
    double h[3] = {1, 1, 1};
    std::vector<int32_t> local_of_box, box_of_local;
  };

  bool detect_lattice(const pfm_mesh_desc *m, Lattice &L)
  {
    const int dim = m->dim, nv = 1 << dim;
    if (m->n_hanging > 0 || m->cell_lambda || m->cell_mu)
      return false;
    int nc[3] = {m->box_cells[0], m->box_cells[1], dim == 3 ? m->box_cells[2] : 1};
    if (nc[0] <= 0 || nc[1] <= 0 || nc[2] <= 0)
      return false;
    const int NX = nc[0] + 1, NY = nc[1] + 1, NZ = dim == 3 ? nc[2] + 1 : 1;
    const int64_t nn = (int64_t)NX * NY * NZ;
    if (nn != m->n_nodes || (int64_t)nc[0] * nc[1] * nc[2] != m->n_cells)
      return false;
    const int32_t N = m->n_nodes;
    double x0[3] = {0, 0, 0}, x1[3] = {0, 0, 0}, h[3] = {1, 1, 1};
    for (int d = 0; d < dim; ++d)
      {
        x0[d] = x1[d] = m->coords[d];
        for (int32_t n = 1; n < N; ++n)
          {
            x0[d] = std::min(x0[d], m->coords[(size_t)n * dim + d]);
            x1[d] = std::max(x1[d], m->coords[(size_t)n * dim + d]);
          }
        h[d] = (x1[d] - x0[d]) / nc[d];
        if (!(h[d] > 0))
          return false;
      }
    std::vector<int32_t> local_of_box((size_t)nn, -1);
    std::vector<int32_t> box_of_local((size_t)N);
    for (int32_t n = 0; n < N; ++n)
      {
        int64_t idx[3] = {0, 0, 0};
        for (int d = 0; d < dim; ++d)
          {
            const double t = (m->coords[(size_t)n * dim + d] - x0[d]) / h[d];
            idx[d] = llround(t);
            if (std::abs(t - (double)idx[d]) > 1e-9 || idx[d] < 0 || idx[d] > nc[d])
              return false;
          }
        const int64_t b = idx[0] + (int64_t)NX * (idx[1] + (int64_t)NY * idx[2]);
        if (local_of_box[b] != -1)
          return false; // duplicated coordinates (e.g. a slit): not a lattice
        local_of_box[b] = n;
        box_of_local[n] = (int32_t)b;
      }
    // every lattice cell must be present exactly once, vertices in deal.II order
    std::vector<uint8_t> seen((size_t)m->n_cells, 0);
    for (int64_t cell = 0; cell < m->n_cells; ++cell)
      {
        const int64_t b0 = box_of_local[m->cell_nodes[cell * nv]];
        const int64_t i = b0 % NX, j = (b0 / NX) % NY, k = b0 / ((int64_t)NX * NY);
        if (i >= nc[0] || j >= nc[1] || (dim == 3 && k >= nc[2]))
          return false;
        for (int a = 0; a < nv; ++a)
          {
            const int64_t b = (i + (a & 1)) + (int64_t)NX * ((j + ((a >> 1) & 1)) + (int64_t)NY * (k + ((a >> 2) & 1)));
            if (local_of_box[b] != m->cell_nodes[cell * nv + a])
              return false;
          }
        const int64_t cid = i + (int64_t)nc[0] * (j + (int64_t)nc[1] * k);
        if (seen[cid])
          return false;
        seen[cid] = 1;
      }
    L.NX = NX;
    L.NY = NY;
    L.NZ = NZ;
    for (int d = 0; d < 3; ++d)
      {
        L.nc[d] = nc[d];
        L.h[d] = h[d];
      }
    L.local_of_box.swap(local_of_box);
    L.box_of_local.swap(box_of_local);
    return true;
  }

  // Build the fast-path tables of a lattice mesh.  Returns false (general path) whenever a
  // check fails; never an error.
  bool build_cart(pfm_ctx *c, const pfm_mesh_desc *m, const Lattice &L)
  {
    const int dim = m->dim;
    const int NX = L.NX, NY = L.NY, NZ = L.NZ;
    const int32_t NO = m->n_owned_nodes;
    const double *h = L.h;
    const std::vector<int32_t> &local_of_box = L.local_of_box, &box_of_local = L.box_of_local;
    // owned nodes must form a sub-box
    int o0[3] = {1 << 30, 1 << 30, 1 << 30}, o1[3] = {-1, -1, -1};
    for (int32_t n = 0; n < NO; ++n)
      {
        const int64_t b = box_of_local[n];
        const int idx[3] = {(int)(b % NX), (int)((b / NX) % NY), (int)(b / ((int64_t)NX * NY))};
        for (int d = 0; d < 3; ++d)
          {
            o0[d] = std::min(o0[d], idx[d]);
            o1[d] = std::max(o1[d], idx[d]);
          }
      }
    if (NO == 0)
      return false;
    if ((int64_t)(o1[0] - o0[0] + 1) * (o1[1] - o0[1] + 1) * (o1[2] - o0[2] + 1) != NO)
      return false;
    // CSR neighbour slot -> lattice offset index (dx+1) + 3*(dy+1) + 9*(dz+1)
    const int no = dim == 3 ? 27 : 9;
    std::vector<uint8_t> inv((size_t)NO * no, 0xff);
    for (int32_t n = 0; n < NO; ++n)
      {
        const int64_t b = box_of_local[n];
        const int i = (int)(b % NX), j = (int)((b / NX) % NY), k = (int)(b / ((int64_t)NX * NY));
        const int32_t *rb = c->h_nadj.data() + c->h_nadj_ptr[n];
        const int32_t *re = c->h_nadj.data() + c->h_nadj_ptr[n + 1];
        int found = 0;
        for (int o = 0; o < no; ++o)
          {
            const int ii = i + (o % 3) - 1, jj = j + ((o / 3) % 3) - 1, kk = k + (dim == 3 ? (o / 9) - 1 : 0);
            if (ii < 0 || ii >= NX || jj < 0 || jj >= NY || kk < 0 || kk >= NZ)
              continue;
            const int32_t q = local_of_box[ii + (int64_t)NX * (jj + (int64_t)NY * kk)];
            const int32_t *p = std::find(rb, re, q);
            if (p == re)
              continue; // neighbour not coupled through a local cell (cannot happen for owned rows)
            inv[(size_t)n * no + (p - rb)] = (uint8_t)o;
            ++found;
          }
        if (found != (int)(re - rb))
          return false;
      }
    std::vector<uint32_t> nbr_mask((size_t)NO, 0); // bit o: the neighbour at lattice offset o exists in the row
    for (int32_t n = 0; n < NO; ++n)
      {
        const int deg = (int)(c->h_nadj_ptr[n + 1] - c->h_nadj_ptr[n]);
        for (int sl = 0; sl < deg; ++sl)
          {
            // rows are in lattice order (pfm_ctx_create): slot sl = rank of its offset among the existing ones
            if (sl > 0 && inv[(size_t)n * no + sl] <= inv[(size_t)n * no + sl - 1])
              return false;
            nbr_mask[n] |= 1u << inv[(size_t)n * no + sl];
          }
      }
    CartView &cv = c->cv;
    cv.NX = NX;
    cv.NY = NY;
    cv.NZ = NZ;
    for (int d = 0; d < 3; ++d)
      {
        cv.o0[d] = o0[d];
        cv.o1[d] = o1[d];
        cv.h[d] = h[d];
      }
    cv.local_of_box = dev_upload(c, local_of_box.data(), local_of_box.size());
    cv.nbr_mask = dev_upload(c, nbr_mask.data(), nbr_mask.size());
    cv.owned_lex = 1;
    {
      const int64_t OWX = o1[0] - o0[0] + 1, OWY = o1[1] - o0[1] + 1;
      for (int32_t n = 0; n < NO && cv.owned_lex; ++n)
        {
          const int64_t b = box_of_local[n];
          const int64_t i = b % NX, j = (b / NX) % NY, k = b / ((int64_t)NX * NY);
          if ((i - o0[0]) + OWX * ((j - o0[1]) + OWY * (k - o0[2])) != n)
            cv.owned_lex = 0;
        }
    }
    return true;
  }
} // namespace

int64_t pfm_ctx::block_rows(int b) const
{
  const int dim = v.dim;
  if (v.layout == PFM_LAYOUT_INTERLEAVED)
    return (int64_t)v.n_owned * (dim + 1);
  return (b == 0 || b == 1) ? (int64_t)v.n_owned * dim : (int64_t)v.n_owned;
}

int64_t pfm_ctx::block_nnz(int b) const
{
  const int dim = v.dim;
  const int64_t g = h_nadj_ptr.empty() ? 0 : (int64_t)h_nadj_ptr.back();
  if (v.layout == PFM_LAYOUT_INTERLEAVED)
    return g * (dim + 1) * (dim + 1);
  switch (b)
    {
      case 0:
        return g * dim * dim;
      case 1:
      case 2:
        return g * dim;
      default:
        return g;
    }
}

extern "C"
{
  int pfm_ctx_create(pfm_ctx **out, const pfm_mesh_desc *m, int device)
  {
    if (!out || !m || (m->dim != 2 && m->dim != 3) ||
        (m->layout != PFM_LAYOUT_INTERLEAVED && m->layout != PFM_LAYOUT_BLOCKED) ||
        m->n_nodes <= 0 || m->n_owned_nodes < 0 || m->n_owned_nodes > m->n_nodes || m->n_cells < 0 ||
        !m->cell_nodes || !m->coords || (m->n_hanging > 0 && (!m->hn_nodes || !m->hn_ptr)))
      return PFM_ERR_BAD_ARG;
    pfm_ctx *c = new (std::nothrow) pfm_ctx;
    if (!c)
      return PFM_ERR_NOMEM;
    *out = c;
    c->device = device;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess)
      return hipfail(c, e, "hipSetDevice");

    const int dim = m->dim, nv = 1 << dim;
    const int32_t N = m->n_nodes, NO = m->n_owned_nodes;
    const int64_t NC = m->n_cells;
    DevView &v = c->v;
    v.dim = dim;
    v.layout = m->layout;
    v.n_nodes = N;
    v.n_owned = NO;
    v.n_cells = NC;
    c->n_blocks = (m->layout == PFM_LAYOUT_BLOCKED) ? 4 : 1;

    for (int64_t i = 0; i < NC * nv; ++i)
      if (m->cell_nodes[i] < 0 || m->cell_nodes[i] >= N)
        return fail(c, PFM_ERR_BAD_ARG, "cell_nodes out of range");

    Lattice lattice;
    bool lattice_ok = false;
    try
      {
        // ---- hanging table: node -> k
        std::vector<int32_t> hn_index;
        if (m->n_hanging > 0)
          {
            hn_index.assign(N, -1);
            for (int32_t k = 0; k < m->n_hanging; ++k)
              {
                if (m->hn_nodes[k] < 0 || m->hn_nodes[k] >= N)
                  return fail(c, PFM_ERR_BAD_ARG, "hn_nodes out of range");
                hn_index[m->hn_nodes[k]] = k;
              }
            for (int64_t j = 0; j < m->hn_ptr[m->n_hanging]; ++j)
              if (m->hn_parents[j] < 0 || m->hn_parents[j] >= N || hn_index[m->hn_parents[j]] >= 0)
                return fail(c, PFM_ERR_BAD_ARG, "hanging table must be closed (parents unconstrained)");
          }

        // ---- node graph over the constraint-resolved cells, rows = owned nodes
        // pass 1: count cells incident to each owned node (through its own vertices and
        // through hanging vertices it is a parent of)
        auto for_each_resolved = [&](int64_t cell, auto &&fn) {
          for (int a = 0; a < nv; ++a)
            {
              const int32_t n = m->cell_nodes[cell * nv + a];
              const int32_t k = hn_index.empty() ? -1 : hn_index[n];
              fn(n);
              if (k >= 0)
                for (int64_t j = m->hn_ptr[k]; j < m->hn_ptr[k + 1]; ++j)
                  fn(m->hn_parents[j]);
            }
        };
        std::vector<int64_t> inc_ptr((size_t)NO + 1, 0);
        for (int64_t cell = 0; cell < NC; ++cell)
          for_each_resolved(cell, [&](int32_t n) {
            if (n < NO)
              ++inc_ptr[n + 1];
          });
        for (int32_t n = 0; n < NO; ++n)
          inc_ptr[n + 1] += inc_ptr[n];
        std::vector<int64_t> inc((size_t)inc_ptr[NO]);
        {
          std::vector<int64_t> fill(inc_ptr.begin(), inc_ptr.end() - 1);
          for (int64_t cell = 0; cell < NC; ++cell)
            for_each_resolved(cell, [&](int32_t n) {
              if (n < NO)
                inc[fill[n]++] = cell;
            });
        }
        // On a lattice the neighbours of a row are ordered by lattice offset (x fastest), not by local node id: a full
        // row then has its 3^dim slots in the order the row-owner kernels produce them on EVERY rank (ghost nodes,
        // which are numbered after the owned ones, would otherwise break the order next to partition faces).
        lattice_ok = detect_lattice(m, lattice);
        c->h_nadj_ptr.assign((size_t)NO + 1, 0);
        std::vector<int32_t> &nadj = c->h_nadj;
        nadj.clear();
        nadj.reserve((size_t)NO * (dim == 2 ? 9 : 27));
        std::vector<int32_t> tmp;
        for (int32_t n = 0; n < NO; ++n)
          {
            tmp.clear();
            for (int64_t k = inc_ptr[n]; k < inc_ptr[n + 1]; ++k)
              for_each_resolved(inc[k], [&](int32_t q) { tmp.push_back(q); });
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            if (lattice_ok)
              {
                const std::vector<int32_t> &bol = lattice.box_of_local;
                std::sort(tmp.begin(), tmp.end(), [&](int32_t p, int32_t q) { return bol[p] < bol[q]; });
              }
            if (tmp.size() > 254)
              return fail(c, PFM_ERR_UNSUPPORTED, "node with more than 254 neighbours");
            nadj.insert(nadj.end(), tmp.begin(), tmp.end());
            c->h_nadj_ptr[n + 1] = (long long)nadj.size();
          }
        std::vector<int64_t>().swap(inc);

        // ---- slot table: position of vertex b's node in the row of vertex a's node
        std::vector<uint8_t> cslot((size_t)NC * nv * nv, 0xff);
        for (int64_t cell = 0; cell < NC; ++cell)
          for (int a = 0; a < nv; ++a)
            {
              const int32_t A = m->cell_nodes[cell * nv + a];
              if (A >= NO)
                continue;
              const int32_t *rb = nadj.data() + c->h_nadj_ptr[A];
              const int32_t *re = nadj.data() + c->h_nadj_ptr[A + 1];
              for (int b = 0; b < nv; ++b)
                {
                  const int32_t B = m->cell_nodes[cell * nv + b];
                  const int32_t *p = std::find(rb, re, B);
                  cslot[(cell * nv + a) * nv + b] = (uint8_t)(p - rb);
                }
            }

        // ---- device mirrors (SoA)
        {
          std::vector<int32_t> conn((size_t)NC * nv);
          for (int64_t cell = 0; cell < NC; ++cell)
            for (int a = 0; a < nv; ++a)
              conn[(size_t)a * NC + cell] = m->cell_nodes[cell * nv + a];
          v.conn = dev_upload(c, conn.data(), conn.size());
        }
        {
          std::vector<double> xs((size_t)N * dim);
          for (int32_t n = 0; n < N; ++n)
            for (int d = 0; d < dim; ++d)
              xs[(size_t)d * N + n] = m->coords[(size_t)n * dim + d];
          v.coords = dev_upload(c, xs.data(), xs.size());
        }
        v.cell_lambda = v.cell_mu = nullptr;
        if (m->cell_lambda && m->cell_mu)
          {
            v.cell_lambda = dev_upload(c, m->cell_lambda, (size_t)NC);
            v.cell_mu = dev_upload(c, m->cell_mu, (size_t)NC);
          }
        v.nadj_ptr = dev_upload(c, c->h_nadj_ptr.data(), c->h_nadj_ptr.size());
        v.nadj = dev_upload(c, nadj.data(), nadj.size());
        v.cslot = dev_upload(c, cslot.data(), cslot.size());
        v.hn_index = nullptr;
        v.hn_ptr = nullptr;
        v.hn_parents = nullptr;
        v.hn_weights = nullptr;
        if (m->n_hanging > 0)
          {
            v.hn_index = dev_upload(c, hn_index.data(), hn_index.size());
            std::vector<long long> hp(m->hn_ptr, m->hn_ptr + m->n_hanging + 1);
            v.hn_ptr = dev_upload(c, hp.data(), hp.size());
            v.hn_parents = dev_upload(c, m->hn_parents, (size_t)hp.back());
            v.hn_weights = dev_upload(c, m->hn_weights, (size_t)hp.back());
          }
        uint8_t *flags = dev_alloc<uint8_t>(c, (size_t)N);
        e = hipMemset(flags, 0, (size_t)N);
        if (e != hipSuccess)
          throw HipFail{e, "hipMemset"};
        v.node_flags = flags;
        for (int d = 0; d < 3; ++d)
          v.u[d] = (d < dim) ? dev_alloc<double>(c, (size_t)N) : nullptr;
        v.phi = dev_alloc<double>(c, (size_t)N);
        v.phi_old = dev_alloc<double>(c, (size_t)N);
        v.phi_oldold = dev_alloc<double>(c, (size_t)N);
        auto zero = [](void *q, size_t bytes) {
          const hipError_t e = hipMemset(q, 0, bytes);
          if (e != hipSuccess)
            throw HipFail{e, "hipMemset"};
        };
        for (int d = 0; d < dim; ++d)
          zero(v.u[d], sizeof(double) * (size_t)N);
        zero(v.phi, sizeof(double) * (size_t)N);
        zero(v.phi_old, sizeof(double) * (size_t)N);
        zero(v.phi_oldold, sizeof(double) * (size_t)N);
        v.status = dev_alloc<int>(c, 1);
        zero(v.status, sizeof(int));
      }
    catch (const HipFail &f)
      {
        return hipfail(c, f.e, f.what);
      }
    catch (const std::bad_alloc &)
      {
        return fail(c, PFM_ERR_NOMEM, "host allocation failed");
      }
    try
      {
        c->cart_ok = lattice_ok && build_cart(c, m, lattice);
        c->d_scal = dev_alloc<unsigned char>(c, PFM_SCAL_BYTES);
      }
    catch (const HipFail &f)
      {
        return hipfail(c, f.e, f.what);
      }
    c->kernel_path = c->cart_ok ? 1 : 0;
    return PFM_OK;
  }

  int pfm_ctx_destroy(pfm_ctx *c)
  {
    if (!c)
      return PFM_OK;
    (void)hipSetDevice(c->device); // a failure surfaces in the next call on the stream
    (void)hipDeviceSynchronize(); // nothing of this context may still be running when its buffers go
    for (void *p : c->allocs)
      (void)hipFree(p);
    for (auto &p : c->peers)
      {
        if (p.d_send)
          (void)hipFree(p.d_send);
        if (p.d_recv)
          (void)hipFree(p.d_recv);
      }
    for (void *q : {(void *)c->d_send_all, (void *)c->d_recv_all, (void *)c->d_send_ptr, (void *)c->d_recv_ptr})
      if (q)
        (void)hipFree(q);
    if (c->side_stream)
      {
        (void)hipStreamDestroy(c->side_stream);
        (void)hipEventDestroy(c->ev_fork);
        (void)hipEventDestroy(c->ev_join);
      }
    for (auto &ev : c->ev_pool)
      {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
      }
    for (double *p : c->d_stage_vec)
      if (p)
        (void)hipFree(p);
    for (double *p : c->d_stage_res)
      if (p)
        (void)hipFree(p);
    for (double *p : c->d_stage_val)
      if (p)
        (void)hipFree(p);
    delete c;
    return PFM_OK;
  }

  const char *pfm_last_error(const pfm_ctx *c) { return c ? c->err.c_str() : "null context"; }

  int pfm_ctx_set_stream(pfm_ctx *c, void *s)
  {
    if (!c)
      return PFM_ERR_BAD_ARG;
    c->stream = static_cast<hipStream_t>(s);
    return PFM_OK;
  }

  int pfm_set_params(pfm_ctx *c, const pfm_params *p)
  {
    if (!c || !p)
      return PFM_ERR_BAD_ARG;
    if (c->v.dim == 3 && (p->decompose_stress_matrix > 0 || p->decompose_stress_rhs > 0) &&
        p->timestep_number > 0)
      return fail(c, PFM_ERR_UNSUPPORTED,
                  "stress split is 2-D only in the reference (cracks.cc:1685-1690)");
    c->prm = *p;
    c->have_params = true;
    c->scal_dirty = true; // the per-launch scalar tables of the cartesian Jacobian kernels follow the parameters
    return PFM_OK;
  }

  int pfm_set_constraints(pfm_ctx *c, const uint8_t *node_flags)
  {
    if (!c || !node_flags)
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    hipError_t e = hipMemcpyAsync(const_cast<uint8_t *>(c->v.node_flags), node_flags,
                                  (size_t)c->v.n_nodes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
      e = hipStreamSynchronize(c->stream); // node_flags is a borrowed host buffer
    return e == hipSuccess ? PFM_OK : hipfail(c, e, "set_constraints copy");
  }

  int pfm_pattern_size(const pfm_ctx *c, int block, int64_t *n_rows, int64_t *nnz)
  {
    if (!c || block < 0 || block >= c->n_blocks)
      return PFM_ERR_BAD_ARG;
    if (n_rows)
      *n_rows = c->block_rows(block);
    if (nnz)
      *nnz = c->block_nnz(block);
    return PFM_OK;
  }

  int pfm_pattern_get(const pfm_ctx *c, int block, int64_t *rowptr, int32_t *colind)
  {
    if (!c || block < 0 || block >= c->n_blocks || !rowptr || !colind)
      return PFM_ERR_BAD_ARG;
    const int dim = c->v.dim;
    int ncr, ncc, coff = 0; // components per node in the row / column space of this block
    bool phi_col = false;
    if (c->v.layout == PFM_LAYOUT_INTERLEAVED)
      ncr = ncc = dim + 1;
    else
      {
        ncr = (block == 0 || block == 1) ? dim : 1;
        ncc = (block == 0 || block == 2) ? dim : 1;
        phi_col = (ncc == 1);
      }
    (void)coff;
    (void)phi_col;
    int64_t pos = 0, row = 0;
    rowptr[0] = 0;
    for (int32_t n = 0; n < c->v.n_owned; ++n)
      {
        const long long b = c->h_nadj_ptr[n], e = c->h_nadj_ptr[n + 1];
        for (int ci = 0; ci < ncr; ++ci)
          {
            for (long long k = b; k < e; ++k)
              for (int d = 0; d < ncc; ++d)
                colind[pos++] = c->h_nadj[k] * ncc + d;
            rowptr[++row] = pos;
          }
      }
    return PFM_OK;
  }

  int pfm_state_set(pfm_ctx *c, const double *sol, const double *old, const double *oldold,
                    int on_device)
  {
    if (!c)
      return PFM_ERR_BAD_ARG;
    if (c->n_owned_dofs() == 0)
      return PFM_OK; // a rank that owns nothing: its node state comes from the ghost import alone
    if (!sol || !old || !oldold)
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    const double *src[3] = {sol, old, oldold};
    const double *d[3] = {sol, old, oldold};
    if (!on_device)
      {
        const size_t bytes = sizeof(double) * (size_t)c->n_owned_dofs();
        for (int k = 0; k < 3; ++k)
          {
            if (!c->d_stage_vec[k])
              {
                hipError_t e = hipMalloc((void **)&c->d_stage_vec[k], std::max<size_t>(bytes, 8));
                if (e != hipSuccess)
                  return hipfail(c, e, "hipMalloc stage");
                c->device_bytes += (int64_t)bytes;
              }
            hipError_t e = hipMemcpyAsync(c->d_stage_vec[k], src[k], bytes, hipMemcpyHostToDevice, c->stream);
            if (e != hipSuccess)
              return hipfail(c, e, "state H2D");
            d[k] = c->d_stage_vec[k];
          }
      }
    int rc = launch_state_set(c->v, d[0], d[1], d[2], c->stream);
    if (rc)
      return fail(c, rc, "state_set launch failed");
    if (!on_device)
      {
        hipError_t e = hipStreamSynchronize(c->stream); // host buffers are borrowed
        if (e != hipSuccess)
          return hipfail(c, e, "state sync");
      }
    return PFM_OK;
  }

  int pfm_halo_register(pfm_ctx *c, int n_peers, const int64_t *send_ptr, const int32_t *send_nodes,
                        const int64_t *recv_ptr, const int32_t *recv_nodes)
  {
    if (!c || n_peers < 0 || (n_peers > 0 && (!send_ptr || !recv_ptr)))
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    for (auto &p : c->peers)
      {
        if (p.d_send)
          (void)hipFree(p.d_send);
        if (p.d_recv)
          (void)hipFree(p.d_recv);
      }
    c->peers.assign((size_t)n_peers, HaloPeer{});
    for (int k = 0; k < n_peers; ++k)
      {
        HaloPeer &p = c->peers[k];
        p.n_send = send_ptr[k + 1] - send_ptr[k];
        p.n_recv = recv_ptr[k + 1] - recv_ptr[k];
        for (int64_t j = send_ptr[k]; j < send_ptr[k + 1]; ++j)
          if (send_nodes[j] < 0 || send_nodes[j] >= c->v.n_owned)
            return fail(c, PFM_ERR_BAD_ARG, "halo send node is not an owned node");
        for (int64_t j = recv_ptr[k]; j < recv_ptr[k + 1]; ++j)
          if (recv_nodes[j] < c->v.n_owned || recv_nodes[j] >= c->v.n_nodes)
            return fail(c, PFM_ERR_BAD_ARG, "halo recv node is not a ghost node");
        hipError_t e = hipMalloc((void **)&p.d_send, std::max<size_t>(4, sizeof(int32_t) * p.n_send));
        if (e == hipSuccess)
          e = hipMalloc((void **)&p.d_recv, std::max<size_t>(4, sizeof(int32_t) * p.n_recv));
        if (e == hipSuccess && p.n_send)
          e = hipMemcpy(p.d_send, send_nodes + send_ptr[k], sizeof(int32_t) * p.n_send, hipMemcpyHostToDevice);
        if (e == hipSuccess && p.n_recv)
          e = hipMemcpy(p.d_recv, recv_nodes + recv_ptr[k], sizeof(int32_t) * p.n_recv, hipMemcpyHostToDevice);
        if (e != hipSuccess)
          return hipfail(c, e, "halo_register");
      }
    // concatenated lists for the one-launch pack / unpack
    for (void *q : {(void *)c->d_send_all, (void *)c->d_recv_all, (void *)c->d_send_ptr, (void *)c->d_recv_ptr})
      if (q)
        (void)hipFree(q);
    c->d_send_all = c->d_recv_all = nullptr;
    c->d_send_ptr = c->d_recv_ptr = nullptr;
    c->n_send_all = n_peers ? send_ptr[n_peers] - send_ptr[0] : 0;
    c->n_recv_all = n_peers ? recv_ptr[n_peers] - recv_ptr[0] : 0;
    if (n_peers > 0)
      {
        std::vector<long long> sp((size_t)n_peers + 1), rp((size_t)n_peers + 1);
        for (int k = 0; k <= n_peers; ++k)
          {
            sp[k] = send_ptr[k] - send_ptr[0];
            rp[k] = recv_ptr[k] - recv_ptr[0];
          }
        hipError_t e = hipMalloc((void **)&c->d_send_all, std::max<size_t>(4, sizeof(int32_t) * c->n_send_all));
        if (e == hipSuccess)
          e = hipMalloc((void **)&c->d_recv_all, std::max<size_t>(4, sizeof(int32_t) * c->n_recv_all));
        if (e == hipSuccess)
          e = hipMalloc((void **)&c->d_send_ptr, sizeof(long long) * sp.size());
        if (e == hipSuccess)
          e = hipMalloc((void **)&c->d_recv_ptr, sizeof(long long) * rp.size());
        if (e == hipSuccess && c->n_send_all)
          e = hipMemcpy(c->d_send_all, send_nodes + send_ptr[0], sizeof(int32_t) * c->n_send_all, hipMemcpyHostToDevice);
        if (e == hipSuccess && c->n_recv_all)
          e = hipMemcpy(c->d_recv_all, recv_nodes + recv_ptr[0], sizeof(int32_t) * c->n_recv_all, hipMemcpyHostToDevice);
        if (e == hipSuccess)
          e = hipMemcpy(c->d_send_ptr, sp.data(), sizeof(long long) * sp.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess)
          e = hipMemcpy(c->d_recv_ptr, rp.data(), sizeof(long long) * rp.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess)
          return hipfail(c, e, "halo_register (concatenated lists)");
      }
    return PFM_OK;
  }

  int pfm_halo_pack_all(pfm_ctx *c, double *d_buf_all)
  {
    if (!c || (!d_buf_all && c->n_send_all))
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    return launch_halo_all(c->v, c->d_send_all, c->d_send_ptr, (int)c->peers.size(), c->n_send_all, d_buf_all, 0, c->stream);
  }

  int pfm_halo_unpack_all(pfm_ctx *c, const double *d_buf_all)
  {
    if (!c || (!d_buf_all && c->n_recv_all))
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    return launch_halo_all(c->v, c->d_recv_all, c->d_recv_ptr, (int)c->peers.size(), c->n_recv_all,
                           const_cast<double *>(d_buf_all), 1, c->stream);
  }

  int pfm_halo_pack(pfm_ctx *c, int peer, double *d_buf)
  {
    if (!c || peer < 0 || peer >= (int)c->peers.size() || !d_buf)
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    return launch_halo_pack(c->v, c->peers[peer].d_send, c->peers[peer].n_send, d_buf, c->stream);
  }

  int pfm_halo_unpack(pfm_ctx *c, int peer, const double *d_buf)
  {
    if (!c || peer < 0 || peer >= (int)c->peers.size() || !d_buf)
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    return launch_halo_unpack(c->v, c->peers[peer].d_recv, c->peers[peer].n_recv, d_buf, c->stream);
  }

  int pfm_assemble_device(pfm_ctx *c, int residual_only, double *const *d_values,
                          double *d_res_pde, double *d_res_tot)
  {
    // a rank may own nothing (empty partition piece): null buffers are fine where there is nothing to write
    const bool no_rows = c && c->n_owned_dofs() == 0;
    if (!c || (!d_res_pde && !no_rows) || (residual_only && !d_res_tot && !no_rows) || (!residual_only && !d_values))
      return PFM_ERR_BAD_ARG;
    if (!c->have_params)
      return fail(c, PFM_ERR_BAD_ARG, "pfm_set_params has not been called");
    (void)hipSetDevice(c->device);
    hipError_t e = hipSuccess;
    const bool split = (c->prm.decompose_stress_matrix > 0 || c->prm.decompose_stress_rhs > 0) &&
                       c->prm.timestep_number > 0;
    const bool cart = c->kernel_path == 1 && !split && (residual_only || cart_matrix_supported(c->v.dim));
    const bool overlay_uu = c->kernel_path == 2 && !residual_only && !split; // debug: general + cart (u,u)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (c->timing)
      {
        if (c->ev_used == c->ev_pool.size())
          {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)
              return fail(c, PFM_ERR_HIP, "hipEventCreate");
            c->ev_pool.emplace_back(a, b);
          }
        ev0 = c->ev_pool[c->ev_used].first;
        ev1 = c->ev_pool[c->ev_used].second;
        ++c->ev_used;
        (void)hipEventRecord(ev0, c->stream);
      }
    // Optional (PFM_SIDE_STREAM=1): residual kernel on a side stream next to the Jacobian kernels.  Measured on MI355X at 216^3: no gain (21.4 vs 21.1 ms per assembly),
    // the kernels do not share CUs usefully; off by default.
    hipStream_t s_res = c->stream;
    const bool fork = cart && !residual_only && getenv("PFM_SIDE_STREAM");
    if (fork)
      {
        if (!c->side_stream)
          {
            if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)
              return fail(c, PFM_ERR_HIP, "side stream");
          }
        s_res = c->side_stream;
        e = hipEventRecord(c->ev_fork, c->stream);
        if (e == hipSuccess)
          e = hipStreamWaitEvent(s_res, c->ev_fork, 0);
        if (e != hipSuccess)
          return hipfail(c, e, "fork");
      }
    // zero the outputs (cracks.cc:2133-2137); the row-owner kernels of the cartesian path
    // write every entry exactly once and need no zeroing pass
    if (!cart)
      e = hipMemsetAsync(d_res_pde, 0, sizeof(double) * (size_t)c->n_owned_dofs(), c->stream);
    if (!cart && e == hipSuccess && residual_only)
      e = hipMemsetAsync(d_res_tot, 0, sizeof(double) * (size_t)c->n_owned_dofs(), c->stream);
    if (e == hipSuccess && !residual_only)
      for (int b = 0; b < c->n_blocks && e == hipSuccess; ++b)
        {
          if (!d_values[b] && c->block_nnz(b) > 0)
            return fail(c, PFM_ERR_BAD_ARG, "null matrix block");
          // the row-owner kernels write every value once, the structurally zero (u,phi) block
          // of the blocked layout included (k_cart_phi4)
          if (!cart)
            e = hipMemsetAsync(d_values[b], 0, sizeof(double) * (size_t)c->block_nnz(b), c->stream);
        }
    if (e != hipSuccess)
      return hipfail(c, e, "zero outputs");
    if (cart && !residual_only && c->scal_dirty)
      {
        // off the hot path: once per pfm_set_params, complete before any kernel of any stream may read it
        int rcs = upload_mat_scal(c->prm, c->cv, c->d_scal, c->stream);
        if (rcs == PFM_OK && hipStreamSynchronize(c->stream) != hipSuccess)
          rcs = PFM_ERR_HIP;
        if (rcs)
          return fail(c, rcs, "scalar tables");
        c->scal_dirty = false;
      }
    int rc = cart ? launch_assemble_cart(c->v, c->cv, c->prm, residual_only, d_values, d_res_pde, d_res_tot, c->stream, s_res, c->d_scal)
                  : launch_assemble_general(c->v, c->prm, residual_only, d_values, d_res_pde, d_res_tot, c->stream);
    if (fork)
      {
        e = hipEventRecord(c->ev_join, s_res);
        if (e == hipSuccess)
          e = hipStreamWaitEvent(c->stream, c->ev_join, 0);
        if (e != hipSuccess)
          return hipfail(c, e, "join");
      }
    if (rc == PFM_OK && overlay_uu && c->scal_dirty)
      {
        rc = upload_mat_scal(c->prm, c->cv, c->d_scal, c->stream);
        c->scal_dirty = rc != PFM_OK;
      }
    if (rc == PFM_OK && overlay_uu)
      rc = launch_cart_uu_only(c->v, c->cv, c->prm, d_values[0], c->stream, c->d_scal);
    if (ev1)
      (void)hipEventRecord(ev1, c->stream);
    if (rc)
      return fail(c, rc, "assemble launch failed");
    return PFM_OK;
  }

  int pfm_sync_status(pfm_ctx *c)
  {
    if (!c)
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    int st = 0;
    hipError_t e = hipMemcpyAsync(&st, c->v.status, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess)
      e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess)
      return hipfail(c, e, "sync_status");
    if (st != 0)
      {
        (void)hipMemsetAsync(c->v.status, 0, sizeof(int), c->stream);
        return fail(c, st, st == PFM_ERR_NOT_ORTHOGONAL
                             ? "eigenvectors not orthogonal (cracks.cc:1732-1736)"
                             : "device-side error");
      }
    return PFM_OK;
  }

  int pfm_assemble(pfm_ctx *c, const double *sol, const double *old, const double *oldold,
                   int residual_only, double *const *values, double *residual_pde,
                   double *residual_total)
  {
    if (!c || !residual_pde || (residual_only && !residual_total) || (!residual_only && !values))
      return PFM_ERR_BAD_ARG;
    if (c->v.n_owned != c->v.n_nodes)
      return fail(c, PFM_ERR_BAD_ARG, "pfm_assemble is single-rank only; use the halo entry points");
    int rc = pfm_state_set(c, sol, old, oldold, 0);
    if (rc)
      return rc;
    const size_t vb = sizeof(double) * (size_t)c->n_owned_dofs();
    for (int k = 0; k < 2; ++k)
      if (!c->d_stage_res[k])
        {
          hipError_t e = hipMalloc((void **)&c->d_stage_res[k], std::max<size_t>(vb, 8));
          if (e != hipSuccess)
            return hipfail(c, e, "hipMalloc stage residual");
          c->device_bytes += (int64_t)vb;
        }
    if (!residual_only)
      for (int b = 0; b < c->n_blocks; ++b)
        if (!c->d_stage_val[b])
          {
            const size_t bytes = sizeof(double) * (size_t)c->block_nnz(b);
            hipError_t e = hipMalloc((void **)&c->d_stage_val[b], std::max<size_t>(bytes, 8));
            if (e != hipSuccess)
              return hipfail(c, e, "hipMalloc stage values");
            c->device_bytes += (int64_t)bytes;
          }
    rc = pfm_assemble_device(c, residual_only, c->d_stage_val, c->d_stage_res[0], c->d_stage_res[1]);
    if (rc)
      return rc;
    hipError_t e = hipMemcpyAsync(residual_pde, c->d_stage_res[0], vb, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && residual_only)
      e = hipMemcpyAsync(residual_total, c->d_stage_res[1], vb, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && !residual_only)
      for (int b = 0; b < c->n_blocks && e == hipSuccess; ++b)
        e = hipMemcpyAsync(values[b], c->d_stage_val[b], sizeof(double) * (size_t)c->block_nnz(b),
                           hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess)
      return hipfail(c, e, "copy back");
    return pfm_sync_status(c);
  }

  int pfm_timing_enable(pfm_ctx *c, int on)
  {
    if (!c)
      return PFM_ERR_BAD_ARG;
    c->timing = on != 0;
    c->ev_used = 0;
    return PFM_OK;
  }

  int pfm_kernel_time_ms(pfm_ctx *c, double *mean_ms, int *n_launches)
  {
    if (!c || !mean_ms)
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess)
      return hipfail(c, e, "kernel_time sync");
    double sum = 0.0;
    for (size_t k = 0; k < c->ev_used; ++k)
      {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, c->ev_pool[k].first, c->ev_pool[k].second);
        if (e != hipSuccess)
          return hipfail(c, e, "hipEventElapsedTime");
        sum += ms;
      }
    *mean_ms = c->ev_used ? sum / (double)c->ev_used : 0.0;
    if (n_launches)
      *n_launches = (int)c->ev_used;
    c->ev_used = 0;
    return PFM_OK;
  }

  int pfm_ctx_kernel_path(const pfm_ctx *c) { return c ? c->kernel_path : -1; }

  int pfm_ctx_force_path(pfm_ctx *c, int path)
  {
    if (!c || path < 0 || path > 2 || (path >= 1 && !c->cart_ok) || (path == 2 && c->v.dim != 3))
      return PFM_ERR_UNSUPPORTED;
    c->kernel_path = path;
    return PFM_OK;
  }

  int64_t pfm_ctx_device_bytes(const pfm_ctx *c) { return c ? c->device_bytes : 0; }
}
