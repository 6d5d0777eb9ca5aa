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
#include <rccl/rccl.h> // types and constants only: the entry points are bound with dlopen (rccl() below)
#include <dlfcn.h>
#include <algorithm>
#include <atomic>
#include <exception>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <mutex>
#include <thread>
#include <condition_variable>
#include <functional>
#include <pthread.h>
#include <execinfo.h>
#include <fcntl.h>
#include <csignal>
#include <unistd.h>
#include <cstdlib>
#include <chrono>

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

  // Host threads for the O(n_nodes) / O(n_cells) loops of the context build (setup_system of the reference runs them
  // after every refine_mesh, cracks.cc:4148): the cores this process may use (affinity mask, cgroup quota), at most 32.
  int host_threads()
  {
    static int n = 0;
    if (n)
      return n;
    unsigned hc = std::thread::hardware_concurrency();
    int t = hc ? (int)hc : 1;
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string q;
    double per = 0;
    if (f >> q >> per && q != "max" && per > 0)
      t = std::min(t, std::max(1, (int)(std::stod(q) / per)));
    if (const char *e = getenv("PFM_HOST_THREADS"))
      t = std::max(1, atoi(e));
    n = std::max(1, std::min(t, 32));
    return n;
  }

  // Worker threads of the context build, kept between calls: a rebuild at 2.7e5 cells runs some twenty parallel loops of
  // 20-200 us each, and starting 16 threads costs more than such a loop.  One parallel region at a time (a second caller --
  // the colour and overlay threads of pfm_ctx_create run next to the main thread -- starts its own threads as before).
  // A forked child has none of the parent's threads: it starts over with an empty pool (pthread_atfork).
  struct HostPool
  {
    std::mutex mx, region;
    std::condition_variable cv, cv_done;
    std::vector<std::thread> th;
    const std::function<void(int)> *job = nullptr;
    std::atomic<uint64_t> gen{0};
    std::atomic<int> left{0};
    int want = 0;
    std::atomic<bool> stop{false};

    // the loops of a rebuild follow one another within microseconds: a worker (and the caller, for the end of a region) polls
    // for ~50 us before it sleeps on the condition variable (a sleep and a wake-up cost 30-50 us each with 16 threads)
    static bool spin_until(const std::function<bool()> &ready)
    {
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0;; ++i)
        {
          if (ready())
            return true;
          __builtin_ia32_pause();
          if ((i & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(50))
            return false;
        }
    }
    void worker(int id, uint64_t seen) // seen: the generation at the worker's creation (it answers the ones after it)
    {
      for (;;)
        {
          if (!spin_until([&] { return stop.load(std::memory_order_acquire) || gen.load(std::memory_order_acquire) != seen; }))
            {
              std::unique_lock<std::mutex> lk(mx);
              cv.wait(lk, [&] { return stop.load() || gen.load() != seen; });
            }
          if (stop.load())
            return;
          seen = gen.load(std::memory_order_acquire);
          // EVERY worker answers every region (those beyond `want` without running anything): want and job are then never
          // rewritten while a worker may still read them
          if (id < want)
            (*job)(id);
          if (left.fetch_sub(1, std::memory_order_acq_rel) == 1)
            {
              std::lock_guard<std::mutex> lk(mx);
              cv_done.notify_one();
            }
        }
    }
    // f(t) for t = 0 .. nt-1, t = 0 on the calling thread; false when the pool is taken (caller falls back)
    bool run(int nt, const std::function<void(int)> &f)
    {
      std::unique_lock<std::mutex> reg(region, std::try_to_lock);
      if (!reg.owns_lock())
        return false;
      while ((int)th.size() + 1 < nt)
        {
          const int id = (int)th.size() + 1;
          const uint64_t born = gen.load(std::memory_order_acquire);
          th.emplace_back([this, id, born] { worker(id, born); });
        }
      job = &f;
      want = nt;
      left.store((int)th.size(), std::memory_order_relaxed);
      gen.fetch_add(1, std::memory_order_release);
      {
        std::lock_guard<std::mutex> lk(mx); // a worker is either before its look at gen or asleep and notified below
      }
      cv.notify_all();
      f(0);
      if (!spin_until([&] { return left.load(std::memory_order_acquire) == 0; }))
        {
          std::unique_lock<std::mutex> lk(mx);
          cv_done.wait(lk, [&] { return left.load() == 0; });
        }
      job = nullptr;
      return true;
    }
    ~HostPool()
    {
      stop.store(true);
      {
        std::lock_guard<std::mutex> lk(mx);
      }
      cv.notify_all();
      for (auto &t : th)
        t.join();
    }
  };
  // Three pools: pfm_ctx_create classifies the cells (overlay) and colours them on two threads next to the one that uploads;
  // each takes the first pool that is free.
  constexpr int N_POOLS = 3;
  std::atomic<HostPool *> g_pools{nullptr};
  HostPool *host_pools()
  {
    HostPool *p = g_pools.load(std::memory_order_acquire);
    if (p)
      return p;
    static std::mutex mk;
    std::lock_guard<std::mutex> lk(mk);
    p = g_pools.load(std::memory_order_acquire);
    if (!p)
      {
        static bool hooked = false;
        if (!hooked)
          {
            hooked = true;
            pthread_atfork(nullptr, nullptr, [] { g_pools.store(nullptr, std::memory_order_release); }); // the child's copy is leaked
            atexit([] { delete[] g_pools.exchange(nullptr); });
          }
        p = new HostPool[N_POOLS];
        g_pools.store(p, std::memory_order_release);
      }
    return p;
  }

  // fn(begin, end) over [0, n) in contiguous chunks, one per thread (at least `grain` items per thread).  An exception
  // thrown by a worker (bad_alloc in a lambda that grows a vector) is carried to the caller instead of ending the
  // process in std::terminate.  parallel_chunks: the same with the chunk index, for loops that keep per-chunk results.
  template <class F>
  void parallel_chunks(int nt, F &&fn)
  {
    if (nt <= 1)
      {
        fn(0);
        return;
      }
    std::vector<std::exception_ptr> err((size_t)nt);
    const std::function<void(int)> body = [&fn, &err](int t) {
      try
        {
          fn(t);
        }
      catch (...)
        {
          err[(size_t)t] = std::current_exception();
        }
    };
    HostPool *pools = host_pools();
    bool done = false;
    for (int q = 0; q < N_POOLS && !done; ++q)
      done = pools[q].run(nt, body);
    if (!done)
      {
        std::vector<std::thread> th;
        th.reserve(nt);
        for (int t = 1; t < nt; ++t)
          th.emplace_back([&body, t] { body(t); });
        body(0);
        for (auto &x : th)
          x.join();
      }
    for (auto &e : err)
      if (e)
        std::rethrow_exception(e);
  }
  inline int chunks_for(int64_t n, int64_t grain) { return (int)std::min<int64_t>(host_threads(), std::max<int64_t>(1, n / std::max<int64_t>(grain, 1))); }
  template <class F>
  void parallel_for(int64_t n, F &&fn, int64_t grain = 16384)
  {
    const int nt = chunks_for(n, grain);
    if (nt <= 1)
      {
        fn((int64_t)0, n);
        return;
      }
    parallel_chunks(nt, [&fn, n, nt](int t) { fn(n * t / nt, n * (t + 1) / nt); });
  }

  // Host -> device copy of caller-owned (pageable) memory.  hipMemcpy from pageable memory ran at ~3.6 GB/s on the
  // test box; tables from 256 KB go through two pinned staging buffers filled by the host threads (parallel memcpy) while
  // the previous piece is on the bus.  On return the caller's buffer has been read.  The copy itself is complete as well,
  // unless the calling thread is inside an H2dBatch scope (pfm_ctx_create: some twenty uploads, one synchronisation at the
  // end, 0.1-0.3 ms each otherwise): then it is ordered on the null stream like a hipMemcpyAsync from pinned memory.
  thread_local int g_h2d_batch = 0;
  struct H2dBatch
  {
    H2dBatch() { ++g_h2d_batch; }
    ~H2dBatch() { --g_h2d_batch; }
  };
  struct Stage // two pinned buffers per device: the events belong to the device that was current when they were made
  {
    static constexpr size_t CHUNK = 32u << 20;
    void *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool pending[2] = {false, false};
    int next = 0;
    bool ok = false, tried = false;
    ~Stage()
    {
      for (int i = 0; i < 2; ++i)
        {
          if (pending[i])
            (void)hipEventSynchronize(ev[i]);
          if (buf[i])
            (void)hipHostFree(buf[i]);
          if (ev[i])
            (void)hipEventDestroy(ev[i]);
        }
    }
  };
  std::mutex g_stage_mx; // one transfer at a time goes through the staging buffers (pfm_ctx_create uploads on a second thread)
  Stage g_stages[16];
  Stage *stage_of_current_device() // call with g_stage_mx held; nullptr: no pinned memory to be had
  {
    int dev = 0;
    (void)hipGetDevice(&dev);
    Stage &st = g_stages[dev & 15];
    if (!st.tried)
      {
        st.tried = true;
        st.ok = hipHostMalloc(&st.buf[0], Stage::CHUNK, hipHostMallocPortable) == hipSuccess &&
                hipHostMalloc(&st.buf[1], Stage::CHUNK, hipHostMallocPortable) == hipSuccess &&
                hipEventCreateWithFlags(&st.ev[0], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&st.ev[1], hipEventDisableTiming) == hipSuccess;
      }
    return st.ok ? &st : nullptr;
  }
  hipError_t h2d(void *d, const void *h, size_t bytes)
  {
    if (bytes < (256u << 10))
      return hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
    std::lock_guard<std::mutex> stage_lock(g_stage_mx);
    Stage *stp = stage_of_current_device();
    if (!stp)
      return hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
    Stage &st = *stp;
    // pieces: a table of a few MB in two halves (the second is copied to the staging buffer while the first is on the bus)
    const size_t piece = bytes <= (1u << 20) ? bytes : std::min(Stage::CHUNK, ((bytes + 1) / 2 + 4095) & ~(size_t)4095);
    for (size_t off = 0; off < bytes; off += piece)
      {
        const int k = st.next;
        st.next ^= 1;
        const size_t nb = std::min(piece, bytes - off);
        if (st.pending[k])
          {
            st.pending[k] = false;
            const hipError_t e = hipEventSynchronize(st.ev[k]);
            if (e != hipSuccess)
              return e;
          }
        const char *src = static_cast<const char *>(h) + off;
        char *dst = static_cast<char *>(st.buf[k]);
        parallel_for((int64_t)nb, [&](int64_t b, int64_t e) { memcpy(dst + b, src + b, (size_t)(e - b)); }, 1 << 18);
        hipError_t e = hipMemcpyAsync(static_cast<char *>(d) + off, st.buf[k], nb, hipMemcpyHostToDevice, nullptr);
        if (e == hipSuccess)
          e = hipEventRecord(st.ev[k], nullptr);
        if (e != hipSuccess)
          return e;
        st.pending[k] = true;
      }
    return g_h2d_batch > 0 ? hipSuccess : hipStreamSynchronize(nullptr);
  }

  // Device -> host copy into pageable memory of the caller through the same buffers, ordered behind the work on `s`,
  // complete on return (results the host reads right after a call: residual vectors, small matrices).
  hipError_t d2h_staged(void *h, const void *d, size_t bytes, hipStream_t s)
  {
    std::lock_guard<std::mutex> stage_lock(g_stage_mx);
    Stage *stp = stage_of_current_device();
    if (!stp)
      {
        const hipError_t e = hipStreamSynchronize(s);
        return e != hipSuccess ? e : hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
      }
    Stage &st = *stp;
    for (int k = 0; k < 2; ++k)
      if (st.pending[k])
        {
          st.pending[k] = false;
          const hipError_t e = hipEventSynchronize(st.ev[k]);
          if (e != hipSuccess)
            return e;
        }
    const size_t piece = Stage::CHUNK;
    size_t prev_off = 0, prev_nb = 0;
    int prev_k = -1;
    auto land = [&]() -> hipError_t { // the previous piece: wait for it, hand it to the caller's buffer
      if (prev_k < 0)
        return hipSuccess;
      const hipError_t e = hipEventSynchronize(st.ev[prev_k]);
      if (e != hipSuccess)
        return e;
      const char *src = static_cast<const char *>(st.buf[prev_k]);
      char *dst = static_cast<char *>(h) + prev_off;
      parallel_for((int64_t)prev_nb, [&](int64_t b, int64_t e2) { memcpy(dst + b, src + b, (size_t)(e2 - b)); }, 1 << 18);
      prev_k = -1;
      return hipSuccess;
    };
    for (size_t off = 0; off < bytes; off += piece)
      {
        const int k = st.next;
        st.next ^= 1;
        const size_t nb = std::min(piece, bytes - off);
        hipError_t e = hipMemcpyAsync(st.buf[k], static_cast<const char *>(d) + off, nb, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess)
          e = hipEventRecord(st.ev[k], s);
        if (e == hipSuccess)
          e = land();
        if (e != hipSuccess)
          return e;
        prev_k = k, prev_off = off, prev_nb = nb;
      }
    return land();
  }

  template <class T>
  T *dev_upload(pfm_ctx *c, const T *h, size_t n)
  {
    T *d = dev_alloc<T>(c, n);
    if (n)
      {
        hipError_t e = h2d(d, h, n * sizeof(T));
        if (e != hipSuccess)
          throw HipFail{e, "hipMemcpy H2D"};
      }
    return d;
  }
  // a[i] += a[i-1] ... in place (row pointers of 1e7 rows: 24 ms on one core)
  template <class T>
  void inclusive_scan_parallel(T *a, int64_t n)
  {
    const int nt = chunks_for(n, 65536);
    if (nt <= 1)
      {
        for (int64_t i = 1; i < n; ++i)
          a[i] += a[i - 1];
        return;
      }
    std::vector<T> tot((size_t)nt);
    parallel_chunks(nt, [&](int t) {
      const int64_t b = n * t / nt, e = n * (t + 1) / nt;
      T s = 0;
      for (int64_t i = b; i < e; ++i)
        {
          s += a[i];
          a[i] = s;
        }
      tot[(size_t)t] = s;
    });
    T run = 0;
    for (int t = 0; t < nt; ++t)
      {
        const T x = tot[(size_t)t];
        tot[(size_t)t] = run;
        run += x;
      }
    parallel_chunks(nt, [&](int t) {
      const T off = tot[(size_t)t];
      if (off == 0)
        return;
      const int64_t b = n * t / nt, e = n * (t + 1) / nt;
      for (int64_t i = b; i < e; ++i)
        a[i] += off;
    });
  }

  // Lattice of a uniform Cartesian box: node n <-> lattice index box_of_local[n] (x fastest), cells in deal.II
  // vertex order, every lattice cell present exactly once.  false whenever any check fails.
  using Lattice = pfm::LatticeHost;

  bool detect_lattice(const pfm_mesh_desc *m, Lattice &L)
  {
    const int dim = m->dim, nv = 1 << dim;
    if (m->n_hanging > 0)
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
    {
      const int nt = chunks_for(N, 16384);
      std::vector<double> lo((size_t)nt * 3), hi((size_t)nt * 3);
      parallel_chunks(nt, [&](int t) {
        const int64_t nb = (int64_t)N * t / nt, ne = (int64_t)N * (t + 1) / nt;
        double a[3], b[3];
        for (int d = 0; d < dim; ++d)
          a[d] = b[d] = m->coords[(size_t)nb * dim + d];
        for (int64_t n = nb; n < ne; ++n)
          for (int d = 0; d < dim; ++d)
            {
              const double x = m->coords[(size_t)n * dim + d];
              a[d] = std::min(a[d], x);
              b[d] = std::max(b[d], x);
            }
        for (int d = 0; d < dim; ++d)
          {
            lo[(size_t)t * 3 + d] = a[d];
            hi[(size_t)t * 3 + d] = b[d];
          }
      });
      for (int d = 0; d < dim; ++d)
        {
          x0[d] = lo[d];
          x1[d] = hi[d];
          for (int t = 1; t < nt; ++t)
            {
              x0[d] = std::min(x0[d], lo[(size_t)t * 3 + d]);
              x1[d] = std::max(x1[d], hi[(size_t)t * 3 + d]);
            }
          h[d] = (x1[d] - x0[d]) / nc[d];
          if (!(h[d] > 0))
            return false;
        }
    }
    pfm::raw_vector<int32_t> local_of_box((size_t)nn); // every entry is written below or the mesh is rejected
    pfm::raw_vector<int32_t> box_of_local((size_t)N);
    std::atomic<bool> ok{true};
    parallel_for(N, [&](int64_t nb, int64_t ne) {
      for (int64_t n = nb; n < ne; ++n)
        {
          int64_t idx[3] = {0, 0, 0};
          for (int d = 0; d < dim; ++d)
            {
              const double t = (m->coords[(size_t)n * dim + d] - x0[d]) / h[d];
              idx[d] = llround(t);
              if (std::abs(t - (double)idx[d]) > 1e-9 || idx[d] < 0 || idx[d] > nc[d])
                {
                  ok = false;
                  return;
                }
            }
          box_of_local[n] = (int32_t)(idx[0] + (int64_t)NX * (idx[1] + (int64_t)NY * idx[2]));
        }
    });
    if (!ok)
      return false;
    // N = nn nodes on nn positions: a bijection unless two nodes share one (duplicated coordinates, e.g. a slit) -- then
    // one of the two does not find itself at its position (racing writers of one entry: either value shows the clash)
    parallel_for(N, [&](int64_t nb, int64_t ne) {
      for (int64_t n = nb; n < ne; ++n)
        __atomic_store_n(&local_of_box[(size_t)box_of_local[n]], (int32_t)n, __ATOMIC_RELAXED);
    });
    parallel_for(N, [&](int64_t nb, int64_t ne) {
      for (int64_t n = nb; n < ne; ++n)
        if (local_of_box[(size_t)box_of_local[n]] != (int32_t)n)
          {
            ok = false;
            return;
          }
    });
    if (!ok)
      return false;
    // every lattice cell must be present exactly once, vertices in deal.II order
    pfm::raw_vector<uint8_t> seen((size_t)m->n_cells);
    parallel_for(m->n_cells, [&](int64_t b, int64_t e) { std::fill(seen.begin() + b, seen.begin() + e, (uint8_t)0); }, 1 << 20);
    parallel_for(m->n_cells, [&](int64_t cb, int64_t ce) {
      for (int64_t cell = cb; cell < ce; ++cell)
        {
          const int64_t b0 = box_of_local[m->cell_nodes[cell * nv]];
          const int64_t i = b0 % NX, j = (b0 / NX) % NY, k = b0 / ((int64_t)NX * NY);
          if (i >= nc[0] || j >= nc[1] || (dim == 3 && k >= nc[2]))
            {
              ok = false;
              return;
            }
          for (int a = 0; a < nv; ++a)
            {
              const int64_t b = (i + (a & 1)) + (int64_t)NX * ((j + ((a >> 1) & 1)) + (int64_t)NY * (k + ((a >> 2) & 1)));
              if (local_of_box[b] != m->cell_nodes[cell * nv + a])
                {
                  ok = false;
                  return;
                }
            }
          seen[i + (int64_t)nc[0] * (j + (int64_t)nc[1] * k)] = 1; // distinct cells write distinct bytes unless duplicated
        }
    });
    if (!ok)
      return false;
    parallel_for(m->n_cells, [&](int64_t cb, int64_t ce) {
      for (int64_t cell = cb; cell < ce; ++cell)
        if (!seen[cell])
          {
            ok = false; // n_cells matches the box, so a missing cell means another one is present twice
            return;
          }
    });
    if (!ok)
      return false;
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

  // Node graph of a lattice mesh, rows = owned nodes, columns ascending by local node id (the canonical order of the
  // ABI, what a host CSR sorted by local column id has).  Arithmetic instead of the generic cell-incidence build: the
  // neighbours of lattice node (i,j,k) are the lattice offsets that stay inside the local box (every cell of the box
  // is local, detect_lattice).  Also fills the host copies of the cartesian row tables (row_order_tables).
  void lattice_host_ptr(pfm_ctx *c, int32_t NO, const Lattice &L)
  {
    const int NX = L.NX, NY = L.NY, NZ = L.NZ;
    auto &ptr = c->h_nadj_ptr;
    ptr.resize((size_t)NO + 1);
    ptr[0] = 0;
    parallel_for(NO, [&](int64_t nb, int64_t ne) {
      for (int64_t n = nb; n < ne; ++n)
        {
          const int64_t b = L.box_of_local[n];
          const int i = (int)(b % NX), j = (int)((b / NX) % NY), k = (int)(b / ((int64_t)NX * NY));
          const int cx = 1 + (i > 0) + (i < NX - 1), cy = 1 + (j > 0) + (j < NY - 1), cz = 1 + (k > 0) + (k < NZ - 1);
          ptr[n + 1] = cx * cy * cz;
        }
    });
    inclusive_scan_parallel(ptr.data() + 1, NO);
  }
  void lattice_graph(pfm_ctx *c, int dim, int32_t NO, const Lattice &L)
  {
    (void)dim;
    // a box without ghost nodes, node n at lattice position n: the row lengths are a function of n alone -- the row pointers
    // are made on the device (launch_lattice_row_ptr) and on the host only when a pattern query asks for them
    std::atomic<bool> positional{(int64_t)NO == (int64_t)L.box_of_local.size()};
    if (positional)
      parallel_for(NO, [&](int64_t nb, int64_t ne) {
        for (int64_t n = nb; n < ne; ++n)
          if (L.box_of_local[n] != (int32_t)n)
            {
              positional = false;
              return;
            }
      });
    c->graph_positional = positional;
    c->h_nadj_ptr.clear();
    if (positional)
      {
        auto span = [](int n) { return n <= 1 ? 1LL : 3LL * n - 2; }; // sum over a line of (1 + left + right)
        c->nadj_total = span(L.NX) * span(L.NY) * span(L.NZ);
      }
    else
      lattice_host_ptr(c, NO, L);
    c->h_nadj.clear();
    c->graph_lazy = true; // rows are materialised by ensure_host_graph
  }

  // neighbours of owned lattice node n in lattice-offset order; returns their number, mask bit o = offset o exists
  int lattice_row(const Lattice &L, int dim, int64_t n, int32_t (&q)[27], uint32_t &mask)
  {
    const int NX = L.NX, NY = L.NY, NZ = L.NZ, no = dim == 3 ? 27 : 9;
    const int64_t b = L.box_of_local[n];
    const int i = (int)(b % NX), j = (int)((b / NX) % NY), k = (int)(b / ((int64_t)NX * NY));
    int deg = 0;
    mask = 0;
    for (int o = 0; o < no; ++o)
      {
        const int ii = i + (o % 3) - 1, jj = j + ((o / 3) % 3) - 1, kk = k + (dim == 3 ? (o / 9) - 1 : 0);
        if (ii < 0 || ii >= NX || jj < 0 || jj >= NY || kk < 0 || kk >= NZ)
          continue;
        mask |= 1u << o;
        q[deg++] = L.local_of_box[ii + (int64_t)NX * (jj + (int64_t)NY * kk)];
      }
    return deg;
  }

  // the canonical rows (ascending local node id) of a lattice context whose host graph has not been materialised
  void ensure_host_graph(pfm_ctx *c)
  {
    if (c->graph_dev_only)
      {
        // general mesh: rows and row pointers were built on the device (pfm_graph.hip); the host copy is a cache for pattern queries
        (void)hipSetDevice(c->device);
        if (c->h_nadj_ptr.empty())
          {
            c->h_nadj_ptr.resize((size_t)c->v.n_owned + 1);
            if (hipMemcpy(c->h_nadj_ptr.data(), c->v.nadj_ptr, sizeof(long long) * c->h_nadj_ptr.size(), hipMemcpyDeviceToHost) != hipSuccess)
              {
                c->h_nadj_ptr.clear();
                throw HipFail{hipGetLastError(), "node graph D2H"};
              }
          }
        if (c->h_nadj.empty() && c->h_nadj_ptr.back() > 0)
          {
            c->h_nadj.resize((size_t)c->h_nadj_ptr.back());
            if (hipMemcpy(c->h_nadj.data(), c->v.nadj, sizeof(int32_t) * c->h_nadj.size(), hipMemcpyDeviceToHost) != hipSuccess)
              {
                c->h_nadj.clear();
                throw HipFail{hipGetLastError(), "node graph D2H"};
              }
          }
        return;
      }
    if (!c->graph_lazy)
      return;
    const int32_t NO = c->v.n_owned;
    const int dim = c->v.dim;
    if (c->h_nadj_ptr.empty())
      lattice_host_ptr(c, NO, c->lat);
    const auto &ptr = c->h_nadj_ptr;
    c->h_nadj.resize((size_t)ptr[NO]);
    parallel_for(NO, [&](int64_t nb, int64_t ne) {
      int32_t q[27];
      uint32_t mask;
      for (int64_t n = nb; n < ne; ++n)
        {
          const int deg = lattice_row(c->lat, dim, n, q, mask);
          int32_t *row = c->h_nadj.data() + ptr[n];
          std::copy(q, q + deg, row);
          if (!std::is_sorted(row, row + deg))
            std::sort(row, row + deg);
        }
    });
    c->graph_lazy = false;
  }

  // device copy of the node graph + slot table of the general cell kernel (lattice contexts: on first use)
  void ensure_general_tables(pfm_ctx *c)
  {
    if (c->general_ready)
      return;
    ensure_host_graph(c);
    DevView &v = c->v;
    if (c->colours_lazy)
      {
        int32_t *d_order = dev_alloc<int32_t>(c, (size_t)std::max<int64_t>(v.n_cells, 1));
        const int32_t *d_box = c->graph_positional ? nullptr : dev_upload(c, c->lat.box_of_local.data(), c->lat.box_of_local.size());
        if (launch_lattice_colour_order(d_order, v.conn, d_box, v.n_cells, c->lat.NX, c->lat.NY, nullptr) != PFM_OK)
          throw HipFail{hipGetLastError(), "colour lists"};
        v.color_cells = d_order;
        c->colours_lazy = false;
      }
    v.nadj = dev_upload(c, c->h_nadj.data(), c->h_nadj.size());
    v.cslot = dev_alloc<uint8_t>(c, (size_t)v.n_cells * (size_t)(1 << v.dim) * (size_t)(1 << v.dim));
    if (launch_build_cslot(v, nullptr) != PFM_OK || hipDeviceSynchronize() != hipSuccess)
      throw HipFail{hipGetLastError(), "cslot kernel"};
    c->general_ready = true;
  }

  // Cartesian row tables from the CURRENT order of the node-graph rows (pfm_ctx_create: ascending local id;
  // pfm_pattern_bind: the caller's order): nbr_mask[n] bit o = lattice offset o exists; the CSR slot of offset o is its
  // rank among the existing offsets when the row is in lattice order, else row_perm[nadj_ptr[n] + rank] (bit 31 of the
  // mask set).  Returns false when a row is not a lattice row (cannot happen for meshes detect_lattice accepts).
  bool row_order_tables(const pfm_ctx *c, int dim, int32_t NO, const Lattice &L, std::vector<uint32_t> &mask,
                        std::vector<uint8_t> &perm, bool &any_perm)
  {
    const int NX = L.NX, NY = L.NY, NZ = L.NZ, no = dim == 3 ? 27 : 9;
    mask.assign((size_t)NO, 0u);
    perm.clear();
    std::vector<uint8_t> flagged((size_t)NO, 0);
    std::atomic<bool> ok{true};
    parallel_for(NO, [&](int64_t nb, int64_t ne) {
      for (int64_t n = nb; n < ne; ++n)
        {
          const int64_t b = L.box_of_local[n];
          const int i = (int)(b % NX), j = (int)((b / NX) % NY), k = (int)(b / ((int64_t)NX * NY));
          const int deg = (int)(c->h_nadj_ptr[n + 1] - c->h_nadj_ptr[n]);
          int32_t canon[27];
          const int32_t *row = canon;
          if (c->graph_lazy) // canonical order = ascending local id, not materialised
            {
              uint32_t m0;
              const int dg = lattice_row(L, dim, n, canon, m0);
              if (!std::is_sorted(canon, canon + dg))
                std::sort(canon, canon + dg);
            }
          else
            row = c->h_nadj.data() + c->h_nadj_ptr[n];
          uint32_t mk = 0;
          int rank = 0;
          bool lattice_order = true;
          for (int o = 0; o < no; ++o)
            {
              const int ii = i + (o % 3) - 1, jj = j + ((o / 3) % 3) - 1, kk = k + (dim == 3 ? (o / 9) - 1 : 0);
              if (ii < 0 || ii >= NX || jj < 0 || jj >= NY || kk < 0 || kk >= NZ)
                continue;
              const int32_t q = L.local_of_box[ii + (int64_t)NX * (jj + (int64_t)NY * kk)];
              mk |= 1u << o;
              if (rank >= deg || row[rank] != q)
                lattice_order = false;
              ++rank;
            }
          if (rank != deg)
            ok = false;
          mask[n] = mk | (lattice_order ? 0u : 0x80000000u);
          flagged[n] = !lattice_order;
        }
    });
    if (!ok)
      return false;
    any_perm = false;
    for (int32_t n = 0; n < NO && !any_perm; ++n)
      any_perm = flagged[n] != 0;
    if (!any_perm)
      return true;
    perm.assign((size_t)c->h_nadj_ptr[NO], 0);
    parallel_for(NO, [&](int64_t nb, int64_t ne) {
      for (int64_t n = nb; n < ne; ++n)
        {
          if (!flagged[n])
            continue;
          const int64_t b = L.box_of_local[n];
          const int i = (int)(b % NX), j = (int)((b / NX) % NY), k = (int)(b / ((int64_t)NX * NY));
          const int deg = (int)(c->h_nadj_ptr[n + 1] - c->h_nadj_ptr[n]);
          int32_t canon[27];
          const int32_t *row = canon;
          if (c->graph_lazy)
            {
              uint32_t m0;
              const int dg = lattice_row(L, dim, n, canon, m0);
              std::sort(canon, canon + dg);
            }
          else
            row = c->h_nadj.data() + c->h_nadj_ptr[n];
          int rank = 0;
          for (int o = 0; o < no; ++o)
            {
              const int ii = i + (o % 3) - 1, jj = j + ((o / 3) % 3) - 1, kk = k + (dim == 3 ? (o / 9) - 1 : 0);
              if (ii < 0 || ii >= NX || jj < 0 || jj >= NY || kk < 0 || kk >= NZ)
                continue;
              const int32_t q = L.local_of_box[ii + (int64_t)NX * (jj + (int64_t)NY * kk)];
              const int32_t *p = std::find(row, row + deg, q);
              if (p == row + deg)
                ok = false;
              perm[(size_t)c->h_nadj_ptr[n] + rank] = (uint8_t)(p - row);
              ++rank;
            }
        }
    });
    return ok;
  }

  // upload (or replace) the device copies of the cartesian row tables
  void upload_row_tables(pfm_ctx *c, const std::vector<uint32_t> &mask, const std::vector<uint8_t> &perm, bool any_perm)
  {
    CartView &cv = c->cv;
    if (!cv.nbr_mask)
      cv.nbr_mask = dev_alloc<uint32_t>(c, mask.size());
    if (!mask.empty() && h2d(const_cast<uint32_t *>(cv.nbr_mask), mask.data(), mask.size() * sizeof(uint32_t)) != hipSuccess)
      throw HipFail{hipGetLastError(), "hipMemcpy nbr_mask"};
    if (any_perm)
      {
        if (!c->d_row_perm)
          c->d_row_perm = dev_alloc<uint8_t>(c, perm.size());
        if (h2d(c->d_row_perm, perm.data(), perm.size()) != hipSuccess)
          throw HipFail{hipGetLastError(), "hipMemcpy row_perm"};
      }
    cv.row_perm = any_perm ? c->d_row_perm : nullptr;
  }

  // Build the fast-path tables of a lattice mesh.  Returns false (general path) whenever a
  // check fails; never an error.
  bool build_cart(pfm_ctx *c, const pfm_mesh_desc *m, const Lattice &L)
  {
    const int NX = L.NX, NY = L.NY;
    const int32_t NO = m->n_owned_nodes;
    const auto &box_of_local = L.box_of_local;
    // owned nodes must form a sub-box
    int o0[3] = {1 << 30, 1 << 30, 1 << 30}, o1[3] = {-1, -1, -1};
    {
      std::mutex mx;
      parallel_for(NO, [&](int64_t nb, int64_t ne) {
        int l0[3] = {1 << 30, 1 << 30, 1 << 30}, l1[3] = {-1, -1, -1};
        for (int64_t n = nb; n < ne; ++n)
          {
            const int64_t b = box_of_local[n];
            const int idx[3] = {(int)(b % NX), (int)((b / NX) % NY), (int)(b / ((int64_t)NX * NY))};
            for (int d = 0; d < 3; ++d)
              {
                l0[d] = std::min(l0[d], idx[d]);
                l1[d] = std::max(l1[d], idx[d]);
              }
          }
        std::lock_guard<std::mutex> lock(mx);
        for (int d = 0; d < 3; ++d)
          {
            o0[d] = std::min(o0[d], l0[d]);
            o1[d] = std::max(o1[d], l1[d]);
          }
      });
    }
    if (NO == 0)
      return false;
    if ((int64_t)(o1[0] - o0[0] + 1) * (o1[1] - o0[1] + 1) * (o1[2] - o0[2] + 1) != NO)
      return false;
    // lexicographic numbering of the owned box?
    bool owned_lex = true;
    {
      const int64_t OWX = o1[0] - o0[0] + 1, OWY = o1[1] - o0[1] + 1;
      std::atomic<bool> lex{true};
      parallel_for(NO, [&](int64_t nb, int64_t ne) {
        for (int64_t n = nb; n < ne && lex; ++n)
          {
            const int64_t b = box_of_local[n];
            const int64_t i = b % NX, j = (b / NX) % NY, k = b / ((int64_t)NX * NY);
            if ((i - o0[0]) + OWX * ((j - o0[1]) + OWY * (k - o0[2])) != n)
              lex = false;
          }
      });
      owned_lex = lex;
    }
    // Row tables.  A box without ghost nodes, numbered lexicographically, in the canonical column order (ascending local
    // id, the lazy lattice graph): every row is in lattice order, the neighbour mask is a function of the lattice
    // position alone -- one trivial kernel instead of a threaded host pass over 27 neighbours per node and a 40 MB upload
    // (55 of the 200 ms of a context rebuild at 216^3).
    const bool positional = owned_lex && NO == m->n_nodes && c->graph_lazy;
    std::vector<uint32_t> mask;
    std::vector<uint8_t> perm;
    bool any_perm = false;
    if (!positional && !row_order_tables(c, m->dim, NO, L, mask, perm, any_perm))
      return false;
    CartView &cv = c->cv;
    cv.NX = NX;
    cv.NY = NY;
    cv.NZ = L.NZ;
    for (int d = 0; d < 3; ++d)
      {
        cv.o0[d] = o0[d];
        cv.o1[d] = o1[d];
        cv.h[d] = L.h[d];
      }
    cv.local_of_box = dev_upload(c, L.local_of_box.data(), L.local_of_box.size());
    if (positional)
      {
        if (!cv.nbr_mask)
          cv.nbr_mask = dev_alloc<uint32_t>(c, (size_t)NO);
        if (launch_lattice_masks(const_cast<uint32_t *>(cv.nbr_mask), NX, NY, L.NZ, m->dim, nullptr) != PFM_OK)
          throw HipFail{hipGetLastError(), "lattice mask kernel"};
        cv.row_perm = nullptr;
      }
    else
      upload_row_tables(c, mask, perm, any_perm);
    cv.cell_lam = cv.cell_mu = nullptr;
    if (m->cell_lambda && m->cell_mu)
      {
        // heterogeneous material: the row-owner kernels address cells by lattice position
        const int nv = 1 << m->dim;
        const int64_t CX = NX - 1, CY = NY - 1;
        std::vector<double> la((size_t)m->n_cells), mu((size_t)m->n_cells);
        parallel_for(m->n_cells, [&](int64_t cb, int64_t ce) {
          for (int64_t cell = cb; cell < ce; ++cell)
            {
              const int64_t b0 = box_of_local[m->cell_nodes[cell * nv]]; // vertex 0 = lower corner (detect_lattice)
              const int64_t i = b0 % NX, j = (b0 / NX) % NY, k = b0 / ((int64_t)NX * NY);
              const int64_t at = i + CX * (j + CY * k);
              la[at] = m->cell_lambda[cell];
              mu[at] = m->cell_mu[cell];
            }
        });
        cv.cell_lam = dev_upload(c, la.data(), la.size());
        cv.cell_mu = dev_upload(c, mu.data(), mu.size());
      }
    cv.owned_lex = owned_lex ? 1 : 0;
    cv.cell_avg = nullptr;
    if (m->dim == 2)
      cv.cell_avg = dev_alloc<double>(c, (size_t)std::max<int64_t>((int64_t)(NX - 1) * (NY - 1), 1));
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
  const int64_t g = nadj_total >= 0 ? (int64_t)nadj_total : (h_nadj_ptr.empty() ? 0 : (int64_t)h_nadj_ptr.back());
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

namespace
{
  // PFM_CTX_TIMING=1: wall time of the phases of pfm_ctx_create on stderr (tuning only)

  // ---- cartesian overlay of a general 2-D mesh (DevView::row_patch ..., pfm_kernels.hip: PATCH) -------------------------
  // Every cell that is an axis-parallel rectangle belongs to the lattice of its size (a refinement level); a node is REGULAR
  // when it is owned, neither hanging nor a parent, has exactly four incident cells, all of one level, and none of the nine
  // lattice nodes around it is hanging -- its row is then the plain 9-point row of that level, completed by one workgroup
  // of the patch kernel.  Blocks of 8 x 8 cells (7 x 7 nodes owned per block) tile each level; the general family keeps
  // the cells that touch any other row (reduced colour lists), and skips the regular rows.
  // host-only part (no context access: it runs on its own thread next to the uploads and the node graph build)
  struct PatchPlan
  {
    std::vector<uint8_t> regular, hang; // per node: row of the patch kernel; hanging or duplicated position
    pfm::raw_vector<int32_t> blk_cells, blk_nodes;
    std::vector<int32_t> rows_general;
    int64_t n_regular = 0;
    int n_blocks = 0;
  };
  // Colour classes of the general cell kernel: cells of one class share no node and are assembled by one launch with plain
  // read-modify-write (device-scope FP64 atomics run at ~3e10 /s on this chip: the scatter of a 2-D Jacobian took 1.1 of
  // 1.2 ms).  Greedy in cell order over the cells with subset[cell] != 0 (null: all cells); cells with a hanging vertex go to
  // the last class, which keeps the atomics.  order: the cells by class, ascending within a class; ptr [n_col + 2].
  struct Colours
  {
    std::vector<long long> ptr;
    pfm::raw_vector<int32_t> order;
    bool overflow = false;  // more than 62 classes were needed: such cells sit in the atomic class although no vertex hangs
    bool cancelled = false; // the caller lost interest (cancel): ptr and order are not filled
  };
  // the distinct constraint-resolved nodes of a cell (a vertex that does not hang is its own, a hanging one brings its
  // parents); returns their number, or max_nodes + 1 when there are more
  template <class NodeOf>
  int resolved_nodes(int64_t cell, int nv, NodeOf node_of, const int32_t *hn_index, const int64_t *hn_ptr, const int32_t *hn_parents, int32_t *out,
                     int max_nodes)
  {
    int R = 0;
    for (int a = 0; a < nv; ++a)
      {
        const int32_t A = node_of(cell, a);
        const int32_t k = hn_index ? hn_index[A] : -1;
        const int64_t rb = k < 0 ? 0 : hn_ptr[k], re = k < 0 ? 1 : hn_ptr[k + 1];
        for (int64_t r = rb; r < re; ++r)
          {
            const int32_t P = k < 0 ? A : hn_parents[r];
            bool known = false;
            for (int t = 0; t < R; ++t)
              known = known || out[t] == P;
            if (known)
              continue;
            if (R == max_nodes)
              return max_nodes + 1;
            out[R++] = P;
          }
      }
    return R;
  }

  // hn_ptr / hn_parents != nullptr (3-D, round 5): a cell at a hanging vertex is coloured like any other, over its
  // constraint-resolved nodes -- the cell kernel reduces its element matrix and residual to those nodes before it adds
  // (K' = C^T K C, DevView::cres), so cells of one class that share no resolved node write disjoint entries and need no
  // atomics: every class adds in a fixed order and the assembly is bitwise reproducible.  Only a cell with more than 16
  // resolved nodes (or one that finds no colour) stays in the last class.
  template <class NodeOf>
  void greedy_colours(int64_t NC, int nv, int32_t N, NodeOf node_of, const int32_t *hn_index, const uint8_t *subset, const std::atomic<bool> *cancel,
                      Colours &out, const int64_t *hn_ptr = nullptr, const int32_t *hn_parents = nullptr)
  {
    constexpr uint8_t NONE = 255, ATOMIC = 254;
    pfm::raw_vector<uint8_t> col((size_t)NC);
    pfm::raw_vector<uint64_t> used((size_t)N);
    parallel_for(N, [&](int64_t b, int64_t e) { std::fill(used.begin() + b, used.begin() + e, (uint64_t)0); });
    int n_col = 0;
    for (int64_t cell = 0; cell < NC; ++cell)
      {
        if (subset && !subset[cell])
          {
            col[cell] = NONE;
            continue;
          }
        if ((cell & 4095) == 0 && cancel && cancel->load(std::memory_order_relaxed))
          {
            out.cancelled = true;
            return;
          }
        uint64_t mask = 0;
        bool hanging = false;
        int32_t nd[16];
        int nn = nv;
        for (int a = 0; a < nv; ++a)
          {
            nd[a] = node_of(cell, a);
            hanging = hanging || (hn_index && hn_index[nd[a]] >= 0);
          }
        bool colourable = !hanging;
        if (hanging && hn_ptr)
          {
            nn = resolved_nodes(cell, nv, node_of, hn_index, hn_ptr, hn_parents, nd, 16);
            colourable = nn <= 16;
          }
        if (colourable)
          for (int a = 0; a < nn; ++a)
            mask |= used[nd[a]];
        int k = 63;
        if (colourable && ~mask != 0)
          k = __builtin_ctzll(~mask);
        if (k < 62)
          {
            for (int a = 0; a < nn; ++a)
              used[nd[a]] |= 1ull << k;
            n_col = std::max(n_col, k + 1);
            col[cell] = (uint8_t)k;
          }
        else
          {
            col[cell] = ATOMIC;
            out.overflow = out.overflow || !hanging || (hn_ptr && colourable); // a cell that should have had a colour and found none
          }
      }
    // counting sort, chunked over the host threads
    const int nk = n_col + 1;
    out.ptr.assign((size_t)n_col + 2, 0);
    const int nt = chunks_for(NC, 65536);
    std::vector<long long> hist((size_t)nt * nk, 0);
    parallel_chunks(nt, [&](int t) {
      long long *hh = hist.data() + (size_t)t * nk;
      for (int64_t cell = NC * t / nt; cell < NC * (t + 1) / nt; ++cell)
        if (col[cell] != NONE)
          ++hh[col[cell] == ATOMIC ? n_col : col[cell]];
    });
    for (int k = 0; k < nk; ++k)
      {
        long long at = out.ptr[k];
        for (int t = 0; t < nt; ++t)
          {
            const long long cnt = hist[(size_t)t * nk + k];
            hist[(size_t)t * nk + k] = at;
            at += cnt;
          }
        out.ptr[(size_t)k + 1] = at;
      }
    out.order.resize((size_t)std::max<long long>(out.ptr.back(), 1));
    out.order[0] = 0;
    parallel_chunks(nt, [&](int t) {
      long long *fill = hist.data() + (size_t)t * nk;
      for (int64_t cell = NC * t / nt; cell < NC * (t + 1) / nt; ++cell)
        if (col[cell] != NONE)
          out.order[(size_t)fill[col[cell] == ATOMIC ? n_col : col[cell]]++] = (int32_t)cell;
    });
  }

  // Cells of the plain classes that share a constraint-resolved node with a cell at a hanging vertex add atomically as well
  // (DevView::cell_ring): every row the atomic class touches then only ever receives atomic adds, and the class -- 68
  // latency-bound waves, 15 % of a Jacobian at 2.7e5 cells -- runs next to the others.  false: no such cell (ring empty).
  template <class NodeOf>
  bool ring_cells(int64_t NC, int nv, int32_t N, NodeOf node_of, const int32_t *hn_index, const int64_t *hn_ptr, const int32_t *hn_parents,
                  std::vector<uint8_t> &ring, bool hanging_coloured = false)
  {
    ring.clear();
    if (!hn_index)
      return false;
    pfm::raw_vector<uint8_t> touched((size_t)N), at_hanging((size_t)NC);
    parallel_for(N, [&](int64_t b, int64_t e) { std::fill(touched.begin() + b, touched.begin() + e, (uint8_t)0); });
    std::atomic<bool> any_atomic{false}, any_ring{false};
    parallel_for(NC, [&](int64_t cb, int64_t ce) {
      bool mine = false;
      for (int64_t cell = cb; cell < ce; ++cell)
        {
          bool hanging = false;
          for (int a = 0; a < nv; ++a)
            hanging = hanging || hn_index[node_of(cell, a)] >= 0;
          if (hanging && hanging_coloured)
            {
              // (greedy_colours: such a cell sits in a plain class unless it has more than 16 resolved nodes)
              int32_t tmp[16];
              hanging = resolved_nodes(cell, nv, node_of, hn_index, hn_ptr, hn_parents, tmp, 16) > 16;
            }
          at_hanging[cell] = hanging ? 1 : 0;
          if (!hanging)
            continue;
          mine = true;
          for (int a = 0; a < nv; ++a)
            {
              const int32_t n = node_of(cell, a);
              touched[n] = 1; // (bytes; racing writers store the same value)
              const int32_t k = hn_index[n];
              if (k >= 0)
                for (long long j = hn_ptr[k]; j < hn_ptr[k + 1]; ++j)
                  touched[hn_parents[j]] = 1;
            }
        }
      if (mine)
        any_atomic = true;
    });
    if (!any_atomic)
      return false;
    ring.resize((size_t)NC);
    parallel_for(NC, [&](int64_t cb, int64_t ce) {
      bool mine = false;
      for (int64_t cell = cb; cell < ce; ++cell)
        {
          bool r = false;
          if (!at_hanging[cell])
            for (int a = 0; a < nv; ++a)
              r = r || touched[node_of(cell, a)] != 0;
          ring[cell] = r ? 1 : 0;
          mine = mine || r;
        }
      if (mine)
        any_ring = true;
    });
    if (!any_ring)
      ring.clear();
    return any_ring;
  }

  // The colour lists of the general family over ALL cells, for a context whose overlay made them unnecessary at pfm_ctx_create
  // (full_colours_lazy): from the device copies of the cell table and the hanging-node index, when that family is first run
  // without the overlay (pfm_ctx_force_path, a stress-split assembly of a 3-D overlay context)
  void ensure_full_colours(pfm_ctx *c)
  {
    if (!c->full_colours_lazy)
      return;
    DevView &v = c->v;
    const int64_t NC = v.n_cells;
    const int nv = 1 << v.dim;
    (void)hipSetDevice(c->device);
    pfm::raw_vector<int32_t> conn((size_t)NC * nv), hn;
    if (NC > 0 && hipMemcpy(conn.data(), v.conn, sizeof(int32_t) * conn.size(), hipMemcpyDeviceToHost) != hipSuccess)
      throw HipFail{hipGetLastError(), "cell table D2H"};
    if (v.hn_index)
      {
        hn.resize((size_t)v.n_nodes);
        if (hipMemcpy(hn.data(), v.hn_index, sizeof(int32_t) * hn.size(), hipMemcpyDeviceToHost) != hipSuccess)
          throw HipFail{hipGetLastError(), "hanging index D2H"};
      }
    pfm::raw_vector<long long> hp;
    pfm::raw_vector<int32_t> hpar;
    if (c->hanging_coloured && v.hn_index && c->n_hanging > 0)
      {
        hp.resize((size_t)c->n_hanging + 1);
        if (hipMemcpy(hp.data(), v.hn_ptr, sizeof(long long) * hp.size(), hipMemcpyDeviceToHost) != hipSuccess)
          throw HipFail{hipGetLastError(), "hanging table D2H"};
        hpar.resize((size_t)hp.back());
        if (!hpar.empty() && hipMemcpy(hpar.data(), v.hn_parents, sizeof(int32_t) * hpar.size(), hipMemcpyDeviceToHost) != hipSuccess)
          throw HipFail{hipGetLastError(), "hanging table D2H"};
      }
    static_assert(sizeof(long long) == sizeof(int64_t), "hanging row pointers");
    Colours col;
    greedy_colours(NC, nv, v.n_nodes, [&](int64_t cell, int a) { return conn[(size_t)a * NC + cell]; }, v.hn_index ? hn.data() : nullptr, nullptr, nullptr, col,
                   hp.empty() ? nullptr : reinterpret_cast<const int64_t *>(hp.data()), hp.empty() ? nullptr : hpar.data());
    v.color_cells = dev_upload(c, col.order.data(), col.order.size());
    c->color_ptr.swap(col.ptr);
    if (col.overflow)
      {
        // (never seen: a node of more than 62 cells) every cell adds atomically
        std::vector<uint8_t> all((size_t)NC, 1);
        v.cell_ring = dev_upload(c, all.data(), all.size());
      }
    c->full_colours_lazy = false;
  }

  // indices i of [b, e) with pred(i), ascending (chunked over the host threads)
  template <class P>
  std::vector<int32_t> parallel_select(int64_t b, int64_t e, P &&pred)
  {
    const int64_t n = e - b;
    const int nt = chunks_for(n, 16384);
    std::vector<std::vector<int32_t>> part((size_t)nt);
    parallel_chunks(nt, [&](int t) {
      std::vector<int32_t> &out = part[(size_t)t];
      for (int64_t i = b + n * t / nt; i < b + n * (t + 1) / nt; ++i)
        if (pred(i))
          out.push_back((int32_t)i);
    });
    if (nt == 1)
      return std::move(part[0]);
    size_t total = 0;
    for (auto &q : part)
      total += q.size();
    std::vector<int32_t> out;
    out.reserve(total);
    for (auto &q : part)
      out.insert(out.end(), q.begin(), q.end());
    return out;
  }

  // Every loop below runs over cells, nodes or blocks in chunks on the host threads (HostPool): the classification is part of
  // the context rebuild after every refine_mesh.  Tables that several cells write (the cell of which a node is vertex a, the
  // node at a lattice position) are filled with compare-and-swap: a second claimant marks the entry instead of racing.
  PatchPlan plan_patches2d(const pfm_mesh_desc *m, const std::vector<int32_t> &hn_index)
  {
    PatchPlan pl;
    const int32_t N = m->n_nodes, NO = m->n_owned_nodes;
    const int64_t NC = m->n_cells;
    if (m->dim != 2 || NC == 0 || getenv("PFM_NO_PATCH"))
      return pl;
    // PFM_CTX_TIMING=1: the phases of this classification on stderr (it runs on a thread of its own)
    const bool ptime = getenv("PFM_CTX_TIMING") != nullptr;
    auto pt0 = std::chrono::steady_clock::now();
    auto pmark = [&](const char *what) {
      if (!ptime)
        return;
      const auto t1 = std::chrono::steady_clock::now();
      fprintf(stderr, "[pfm_ctx_create]     plan: %-22s %6.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - pt0).count());
      pt0 = t1;
    };
    const double *X = m->coords;
    double xmin, ymin, xmax, ymax;
    {
      const int nt = chunks_for(N, 16384);
      std::vector<double> bb((size_t)nt * 4);
      parallel_chunks(nt, [&](int t) {
        const int64_t nb = (int64_t)N * t / nt, ne = (int64_t)N * (t + 1) / nt;
        double a0 = X[2 * nb], a1 = a0, b0 = X[2 * nb + 1], b1 = b0;
        for (int64_t n = nb; n < ne; ++n)
          {
            a0 = std::min(a0, X[2 * n]);
            a1 = std::max(a1, X[2 * n]);
            b0 = std::min(b0, X[2 * n + 1]);
            b1 = std::max(b1, X[2 * n + 1]);
          }
        double *q = bb.data() + (size_t)t * 4;
        q[0] = a0, q[1] = a1, q[2] = b0, q[3] = b1;
      });
      xmin = bb[0], xmax = bb[1], ymin = bb[2], ymax = bb[3];
      for (int t = 1; t < nt; ++t)
        {
          xmin = std::min(xmin, bb[(size_t)t * 4]);
          xmax = std::max(xmax, bb[(size_t)t * 4 + 1]);
          ymin = std::min(ymin, bb[(size_t)t * 4 + 2]);
          ymax = std::max(ymax, bb[(size_t)t * 4 + 3]);
        }
    }
    const double tol = 1e-9 * std::max(xmax - xmin, ymax - ymin);
    pmark("bounds");
    // level of a cell (by its size), lattice position of its lower-left vertex
    struct Level
    {
      double hx, hy;
      long long ix0, iy0, ix1, iy1; // cell index range
    };
    struct Size
    {
      double hx, hy;
    };
    // the size of an EXACT axis-parallel rectangle in deal.II's vertex order (the patch kernel takes J = diag(hx, hy));
    // any other cell stays with the general family
    auto cell_size = [&](int64_t k, double &hx, double &hy, double &x0, double &y0) {
      const int32_t *cn = m->cell_nodes + 4 * k;
      x0 = X[2 * cn[0]], y0 = X[2 * cn[0] + 1];
      const double x1 = X[2 * cn[1]], y1 = X[2 * cn[1] + 1];
      const double x2 = X[2 * cn[2]], y2 = X[2 * cn[2] + 1], x3 = X[2 * cn[3]], y3 = X[2 * cn[3] + 1];
      hx = x1 - x0, hy = y2 - y0;
      return hx > tol && hy > tol && y1 == y0 && x2 == x0 && x3 == x1 && y3 == y2;
    };
    auto same_size = [](const Size &a, double hx, double hy) { return std::fabs(a.hx - hx) <= 1e-9 * hx && std::fabs(a.hy - hy) <= 1e-9 * hy; };
    const int ntc = chunks_for(NC, 8192);
    std::vector<Level> levels;
    {
      // the sizes that occur, in the order of their first cell: per chunk, then merged in chunk order
      std::vector<std::vector<Size>> found((size_t)ntc);
      parallel_chunks(ntc, [&](int t) {
        std::vector<Size> &mine = found[(size_t)t];
        for (int64_t k = NC * t / ntc; k < NC * (t + 1) / ntc; ++k)
          {
            double hx, hy, x0, y0;
            if (!cell_size(k, hx, hy, x0, y0))
              continue;
            bool known = false;
            for (const Size &q : mine)
              known = known || same_size(q, hx, hy);
            if (!known && mine.size() < 100)
              mine.push_back(Size{hx, hy});
          }
      });
      for (const auto &mine : found)
        for (const Size &q : mine)
          {
            bool known = false;
            for (const Level &l : levels)
              known = known || same_size(Size{l.hx, l.hy}, q.hx, q.hy);
            if (!known && levels.size() < 100)
              levels.push_back(Level{q.hx, q.hy, LLONG_MAX, LLONG_MAX, LLONG_MIN, LLONG_MIN});
          }
    }
    if (levels.empty())
      return pl;
    const int NL = (int)levels.size();
    pmark("levels");
    pfm::raw_vector<int8_t> cell_level((size_t)NC); // (written by the loop below: the pages are first touched by its threads)
    pfm::raw_vector<int32_t> cix((size_t)NC), ciy((size_t)NC);
    {
      std::vector<Level> part((size_t)ntc * NL);
      std::atomic<bool> too_far{false};
      parallel_chunks(ntc, [&](int t) {
        Level *mine = part.data() + (size_t)t * NL;
        for (int l = 0; l < NL; ++l)
          mine[l] = Level{0, 0, LLONG_MAX, LLONG_MAX, LLONG_MIN, LLONG_MIN};
        for (int64_t k = NC * t / ntc; k < NC * (t + 1) / ntc; ++k)
          {
            cell_level[k] = -1;
            double hx, hy, x0, y0;
            if (!cell_size(k, hx, hy, x0, y0))
              continue;
            int L = -1;
            for (int l = 0; l < NL; ++l)
              if (same_size(Size{levels[l].hx, levels[l].hy}, hx, hy))
                L = l;
            if (L < 0)
              continue;
            const double fx = (x0 - xmin) / levels[L].hx, fy = (y0 - ymin) / levels[L].hy;
            const long long ix = std::llround(fx), iy = std::llround(fy);
            if (std::fabs(fx - (double)ix) > 1e-6 || std::fabs(fy - (double)iy) > 1e-6)
              continue; // off the level's lattice
            if (ix > (1 << 30) || iy > (1 << 30))
              {
                too_far = true;
                continue;
              }
            cell_level[k] = (int8_t)L;
            cix[k] = (int32_t)ix;
            ciy[k] = (int32_t)iy;
            Level &lv = mine[L];
            lv.ix0 = std::min(lv.ix0, ix);
            lv.iy0 = std::min(lv.iy0, iy);
            lv.ix1 = std::max(lv.ix1, ix);
            lv.iy1 = std::max(lv.iy1, iy);
          }
      });
      (void)too_far; // such cells keep level -1: general family
      for (int t = 0; t < ntc; ++t)
        for (int l = 0; l < NL; ++l)
          {
            const Level &q = part[(size_t)t * NL + l];
            Level &lv = levels[l];
            lv.ix0 = std::min(lv.ix0, q.ix0);
            lv.iy0 = std::min(lv.iy0, q.iy0);
            lv.ix1 = std::max(lv.ix1, q.ix1);
            lv.iy1 = std::max(lv.iy1, q.iy1);
          }
    }
    // incident cells per node: inc[4 n + a] = the cell of which n is vertex a (-1 none, -2 two cells claim the corner)
    pmark("cell levels");
    pfm::raw_vector<int32_t> inc((size_t)N * 4);
    parallel_for((int64_t)N * 4, [&](int64_t b, int64_t e) { std::fill(inc.begin() + b, inc.begin() + e, -1); });
    parallel_for(NC, [&](int64_t cb, int64_t ce) {
      for (int64_t k = cb; k < ce; ++k)
        for (int a = 0; a < 4; ++a)
          {
            int32_t *slot = &inc[(size_t)m->cell_nodes[4 * k + a] * 4 + a];
            int32_t expect = -1;
            if (!__atomic_compare_exchange_n(slot, &expect, (int32_t)k, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED))
              __atomic_store_n(slot, -2, __ATOMIC_RELAXED);
          }
    }, 8192);
    pfm::raw_vector<uint8_t> is_parent((size_t)N), dup((size_t)N);
    pfm::raw_vector<int8_t> node_level((size_t)N); // of regular nodes: the common level of their four cells
    std::vector<uint8_t> regular((size_t)N); // (handed over to the plan)
    parallel_for(N, [&](int64_t b, int64_t e) {
      std::fill(is_parent.begin() + b, is_parent.begin() + e, (uint8_t)0);
      std::fill(dup.begin() + b, dup.begin() + e, (uint8_t)0);
      std::fill(node_level.begin() + b, node_level.begin() + e, (int8_t)-1);
      std::fill(regular.begin() + b, regular.begin() + e, (uint8_t)0);
    });
    if (m->n_hanging > 0)
      for (int64_t j = 0; j < m->hn_ptr[m->n_hanging]; ++j)
        is_parent[m->hn_parents[j]] = 1;
    // two nodes at one lattice position (the lips of a slit, meshes/unit_slit.inp): neither they nor their neighbours are
    // regular.  One table per level (node at a lattice position), all in one array.
    {
      std::vector<long long> off((size_t)NL + 1, 0), Wn((size_t)NL, 0);
      for (int L = 0; L < NL; ++L)
        {
          const Level &lv = levels[L];
          long long sz = 0;
          if (lv.ix0 <= lv.ix1)
            {
              const long long W = lv.ix1 - lv.ix0 + 2, Hh = lv.iy1 - lv.iy0 + 2;
              if ((double)W * (double)Hh <= 4.0e8)
                {
                  sz = W * Hh;
                  Wn[L] = W;
                }
            }
          off[(size_t)L + 1] = off[L] + sz;
        }
      pfm::raw_vector<int32_t> node_at((size_t)off[NL]);
      parallel_for(off[NL], [&](int64_t b, int64_t e) { std::fill(node_at.begin() + b, node_at.begin() + e, -1); });
      parallel_for(NC, [&](int64_t cb, int64_t ce) {
        for (int64_t k = cb; k < ce; ++k)
          {
            const int L = cell_level[k];
            if (L < 0 || Wn[L] == 0)
              continue;
            const Level &lv = levels[L];
            for (int a = 0; a < 4; ++a)
              {
                const int32_t n = m->cell_nodes[4 * k + a];
                int32_t *slot = &node_at[(size_t)(off[L] + (ciy[k] - lv.iy0 + (a >> 1)) * Wn[L] + (cix[k] - lv.ix0 + (a & 1)))];
                int32_t seen = -1;
                if (!__atomic_compare_exchange_n(slot, &seen, n, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED) && seen != n)
                  dup[n] = dup[seen] = 1; // (bytes, racing writers store the same value)
              }
          }
      }, 8192);
    }
    auto hanging = [&](int32_t n) { return (!hn_index.empty() && hn_index[n] >= 0) || dup[n] != 0; };
    pmark("incident cells, duplicates");
    int64_t n_regular = 0;
    {
      const int nt = chunks_for(NO, 8192);
      std::vector<int64_t> cnt((size_t)nt, 0);
      parallel_chunks(nt, [&](int t) {
        int64_t mine = 0;
        for (int64_t n = (int64_t)NO * t / nt; n < (int64_t)NO * (t + 1) / nt; ++n)
          {
            if (hanging((int32_t)n) || is_parent[n])
              continue;
            const int32_t k0 = inc[(size_t)n * 4 + 0], k1 = inc[(size_t)n * 4 + 1], k2 = inc[(size_t)n * 4 + 2], k3 = inc[(size_t)n * 4 + 3];
            if (k0 < 0 || k1 < 0 || k2 < 0 || k3 < 0)
              continue; // not exactly four cells, one at each corner
            const int8_t L = cell_level[k0];
            if (L < 0 || cell_level[k1] != L || cell_level[k2] != L || cell_level[k3] != L)
              continue;
            bool ok = true;
            for (const int32_t k : {k0, k1, k2, k3})
              for (int b = 0; b < 4; ++b)
                ok = ok && !hanging(m->cell_nodes[4 * k + b]);
            // the four cells really are the 2 x 2 cells around n on the level's lattice
            ok = ok && cix[k2] == cix[k3] + 1 && ciy[k2] == ciy[k3] && cix[k1] == cix[k3] && ciy[k1] == ciy[k3] + 1 && cix[k0] == cix[k3] + 1 &&
                 ciy[k0] == ciy[k3] + 1;
            if (ok)
              {
                regular[n] = 1;
                node_level[n] = L;
                ++mine;
              }
          }
        cnt[(size_t)t] = mine;
      });
      for (const int64_t q : cnt)
        n_regular += q;
    }
    if (n_regular == 0)
      return pl;
    pmark("regular nodes");
    // blocks: block (bx, by) of a level owns the lattice nodes [7 bx, 7 bx + 6] x [7 by, 7 by + 6] and holds the cells
    // [7 bx - 1, 7 bx + 6] x [7 by - 1, 7 by + 6]; node (i, j) of the lattice = upper-right vertex of cell (i - 1, j - 1)
    pfm::raw_vector<int32_t> blk_cells, blk_nodes;
    for (int L = 0; L < NL; ++L)
      {
        const Level &lv = levels[L];
        if (lv.ix0 > lv.ix1)
          continue;
        const long long W = lv.ix1 - lv.ix0 + 1, Hh = lv.iy1 - lv.iy0 + 1;
        if ((double)W * (double)Hh > 4.0e8)
          continue; // a level whose bounding box is mostly empty: not worth a dense table
        pfm::raw_vector<int32_t> at((size_t)(W * Hh));
        parallel_for(W * Hh, [&](int64_t b, int64_t e) { std::fill(at.begin() + b, at.begin() + e, -1); });
        parallel_for(NC, [&](int64_t cb, int64_t ce) {
          for (int64_t k = cb; k < ce; ++k)
            if (cell_level[k] == (int8_t)L)
              at[(size_t)((ciy[k] - lv.iy0) * W + (cix[k] - lv.ix0))] = (int32_t)k;
        });
        auto cell_at = [&](long long i, long long j) -> int32_t {
          return (i < lv.ix0 || i > lv.ix1 || j < lv.iy0 || j > lv.iy1) ? -1 : at[(size_t)((j - lv.iy0) * W + (i - lv.ix0))];
        };
        auto floordiv = [](long long a, long long b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
        const long long bx0 = floordiv(lv.ix0, 7), bx1 = floordiv(lv.ix1 + 1, 7), by0 = floordiv(lv.iy0, 7), by1 = floordiv(lv.iy1 + 1, 7);
        const long long nbx = bx1 - bx0 + 1, nblk = nbx * (by1 - by0 + 1);
        const int nt = chunks_for(nblk, 64);
        std::vector<std::vector<int32_t>> pc((size_t)nt), pn((size_t)nt);
        parallel_chunks(nt, [&](int t) {
          for (long long q = nblk * t / nt; q < nblk * (t + 1) / nt; ++q)
            {
              const long long by = by0 + q / nbx, bx = bx0 + q % nbx;
              int32_t cells[64], nodes[81];
              bool any = false;
              for (int i = 0; i < 81; ++i)
                nodes[i] = -1;
              for (int cy = 0; cy < 8; ++cy)
                for (int cx = 0; cx < 8; ++cx)
                  {
                    const int32_t k = cell_at(7 * bx - 1 + cx, 7 * by - 1 + cy);
                    cells[cy * 8 + cx] = k;
                    if (k >= 0)
                      for (int a = 0; a < 4; ++a)
                        nodes[(cx + (a & 1)) + 9 * (cy + (a >> 1))] = m->cell_nodes[4 * k + a];
                  }
              for (int hy = 1; hy <= 7; ++hy)
                for (int hx = 1; hx <= 7; ++hx)
                  {
                    const int32_t n = nodes[hx + 9 * hy];
                    any = any || (n >= 0 && n < NO && regular[n] && node_level[n] == (int8_t)L);
                  }
              if (!any)
                continue;
              // a regular node of ANOTHER level may sit at a position of this block (coinciding lattices): it is not ours
              for (int hy = 1; hy <= 7; ++hy)
                for (int hx = 1; hx <= 7; ++hx)
                  {
                    const int32_t n = nodes[hx + 9 * hy];
                    if (n >= 0 && n < NO && regular[n] && node_level[n] != (int8_t)L)
                      nodes[hx + 9 * hy] = -1;
                  }
              pc[(size_t)t].insert(pc[(size_t)t].end(), cells, cells + 64);
              pn[(size_t)t].insert(pn[(size_t)t].end(), nodes, nodes + 81);
            }
        });
        {
          // the chunks' pieces behind one another, copied by the chunks' threads
          std::vector<size_t> oc((size_t)nt + 1, blk_cells.size()), on((size_t)nt + 1, blk_nodes.size());
          for (int t = 0; t < nt; ++t)
            {
              oc[(size_t)t + 1] = oc[(size_t)t] + pc[(size_t)t].size();
              on[(size_t)t + 1] = on[(size_t)t] + pn[(size_t)t].size();
            }
          blk_cells.resize(oc[(size_t)nt]);
          blk_nodes.resize(on[(size_t)nt]);
          parallel_chunks(nt, [&](int t) {
            std::copy(pc[(size_t)t].begin(), pc[(size_t)t].end(), blk_cells.begin() + (std::ptrdiff_t)oc[(size_t)t]);
            std::copy(pn[(size_t)t].begin(), pn[(size_t)t].end(), blk_nodes.begin() + (std::ptrdiff_t)on[(size_t)t]);
          });
        }
      }
    const int n_blocks = (int)(blk_cells.size() / 64);
    pmark("blocks");
    if (n_blocks == 0)
      return pl;
    // only the rows some block really writes are the patch kernel's (a level without a table above keeps its rows general)
    {
      std::vector<uint8_t> covered((size_t)N);
      parallel_for(N, [&](int64_t b, int64_t e) { std::fill(covered.begin() + b, covered.begin() + e, (uint8_t)0); });
      const int nt = chunks_for(n_blocks, 64);
      std::vector<int64_t> cnt((size_t)nt, 0);
      parallel_chunks(nt, [&](int t) {
        int64_t mine = 0;
        for (int64_t b = (int64_t)n_blocks * t / nt; b < (int64_t)n_blocks * (t + 1) / nt; ++b)
          for (int hy = 1; hy <= 7; ++hy)
            for (int hx = 1; hx <= 7; ++hx)
              {
                const int32_t n = blk_nodes[(size_t)b * 81 + hx + 9 * hy];
                if (n >= 0 && n < NO && regular[n] && !__atomic_exchange_n(&covered[n], (uint8_t)1, __ATOMIC_RELAXED))
                  ++mine;
              }
        cnt[(size_t)t] = mine;
      });
      n_regular = 0;
      for (const int64_t q : cnt)
        n_regular += q;
      regular.swap(covered);
    }
    // the rows of the general family (zeroed before every Jacobian; the patch kernel stores its rows whole)
    pmark("covered");
    std::vector<int32_t> rows_general = parallel_select(0, NO, [&](int64_t n) { return !regular[n]; });
    pl.hang.resize((size_t)N);
    parallel_for(N, [&](int64_t nb, int64_t ne) {
      for (int64_t n = nb; n < ne; ++n)
        pl.hang[n] = hanging((int32_t)n) ? 1 : 0;
    });
    pl.regular.swap(regular);
    pl.blk_cells.swap(blk_cells);
    pl.blk_nodes.swap(blk_nodes);
    pl.rows_general.swap(rows_general);
    pmark("rows, flags");
    pl.n_regular = n_regular;
    pl.n_blocks = n_blocks;
    return pl;
  }

  // context part: reduced colour lists, uploads (after the colour classes and the node graph exist)
  void finish_patches2d(pfm_ctx *c, const pfm_mesh_desc *m, PatchPlan &pl, const int32_t *hn_index)
  {
    DevView &v = c->v;
    const int32_t N = m->n_nodes, NO = m->n_owned_nodes;
    const int64_t NC = m->n_cells;
    if (pl.n_blocks == 0)
      return;
    std::vector<uint8_t> &regular = pl.regular;
    pfm::raw_vector<int32_t> &blk_cells = pl.blk_cells, &blk_nodes = pl.blk_nodes;
    std::vector<int32_t> &rows_general = pl.rows_general;
    const int n_blocks = pl.n_blocks;
    const int64_t n_regular = pl.n_regular;
    // a regular node has the nine nodes of its 2 x 2 cells in its row and nothing else, by construction; checked against the
    // node graph, on the host or (general mesh: the graph was built there) on the device below (a violation would mean a
    // coupling this classification does not know: no overlay then)
    if (c->h_nadj_ptr.size() == (size_t)NO + 1)
      {
        std::atomic<bool> nine{true};
        parallel_for(NO, [&](int64_t nb, int64_t ne) {
          for (int64_t n = nb; n < ne; ++n)
            if (regular[n] && c->h_nadj_ptr[n + 1] - c->h_nadj_ptr[n] != 9)
              nine = false;
        });
        if (!nine)
          return;
      }
    // reduced lists of the general family: the cells that touch a row the patches do not write
    std::vector<uint8_t> need((size_t)NC);
    parallel_for(NC, [&](int64_t cb, int64_t ce) {
      for (int64_t k = cb; k < ce; ++k)
        {
          uint8_t q = 0;
          for (int a = 0; a < 4; ++a)
            {
              const int32_t n = m->cell_nodes[4 * k + a];
              if (pl.hang[n] != 0 || (n < NO && !regular[n]))
                q = 1;
            }
          need[k] = q;
        }
    });
    // its colour classes: greedy over these cells alone (the lists over all cells are made when the general family is first
    // run without the overlay, ensure_full_colours)
    Colours red;
    greedy_colours(NC, 4, N, [&](int64_t cell, int a) { return m->cell_nodes[4 * cell + a]; }, hn_index, need.data(), nullptr, red);
    if (red.overflow)
      return; // (a node of more than 62 such cells: no overlay, the caller colours all cells)
    c->color_ptr_reduced.swap(red.ptr);
    pfm::raw_vector<int32_t> &order_red = red.order;
    c->n_general_cells = (int64_t)c->color_ptr_reduced.back();
    v.row_patch = dev_upload(c, regular.data(), regular.size());
    if (c->graph_dev_only && c->h_nadj_ptr.empty())
      {
        int bad = 1;
        if (launch_check_row_lengths(v.row_patch, v.nadj_ptr, NO, 9, v.status, nullptr) != PFM_OK ||
            hipMemcpy(&bad, v.status, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
          throw HipFail{hipGetLastError(), "row length check"};
        if (bad)
          {
            if (hipMemset(v.status, 0, sizeof(int)) != hipSuccess)
              throw HipFail{hipGetLastError(), "hipMemset"};
            v.row_patch = nullptr;
            return;
          }
      }
    c->n_rows_general = (int32_t)rows_general.size();
    if (rows_general.empty())
      rows_general.push_back(0);
    {
      // the overlay's tables in ONE device allocation (a hipMalloc costs 0.05-0.1 ms; the build made five of them here)
      auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
      const size_t b_cells = al(sizeof(int32_t) * blk_cells.size()), b_nodes = al(sizeof(int32_t) * blk_nodes.size());
      const size_t b_slots = al(sizeof(unsigned long long) * (size_t)std::max<int32_t>(NO, 1));
      const size_t b_order = al(sizeof(int32_t) * order_red.size()), b_rows = al(sizeof(int32_t) * rows_general.size());
      char *slab = dev_alloc<char>(c, b_cells + b_nodes + b_slots + b_order + b_rows);
      auto put = [&](char *d, const void *h, size_t bytes) {
        const hipError_t e = h2d(d, h, bytes);
        if (e != hipSuccess)
          throw HipFail{e, "hipMemcpy H2D"};
      };
      char *at = slab;
      put(at, blk_cells.data(), sizeof(int32_t) * blk_cells.size());
      v.patch_cells = reinterpret_cast<const int32_t *>(at);
      at += b_cells;
      put(at, blk_nodes.data(), sizeof(int32_t) * blk_nodes.size());
      v.patch_nodes = reinterpret_cast<const int32_t *>(at);
      at += b_nodes;
      c->d_node_slots = reinterpret_cast<unsigned long long *>(at);
      v.node_slots = c->d_node_slots;
      at += b_slots;
      put(at, order_red.data(), sizeof(int32_t) * order_red.size());
      c->d_color_cells_reduced = reinterpret_cast<int32_t *>(at);
      at += b_order;
      put(at, rows_general.data(), sizeof(int32_t) * rows_general.size());
      c->d_rows_general = reinterpret_cast<int32_t *>(at);
    }
    c->n_patch_blocks = n_blocks;
    c->n_patch_rows = n_regular;
    c->patch_slots_valid = false;
  }

  // ---- cartesian overlay of a general 3-D mesh (round 5) -------------------------------------------------------------------
  // The same classification as in 2-D (plan_patches2d), with the row-owner kernels of the CARTESIAN family as the "patch
  // kernel": every cell that is an exact axis-parallel box belongs to the lattice of its size (a refinement level); a node
  // is REGULAR when it is owned, neither hanging nor a parent, has exactly eight incident cells, all of one level and at
  // the eight lattice positions around it, and none of the 27 lattice nodes around it is hanging -- its rows are then the
  // plain 27-point rows of a uniform box of that level.  Per level one lattice (CartView) over the bounding box of its
  // regular nodes plus a one-node halo: node_at = the node at each lattice point (-1: none of this level), row_at = the
  // regular nodes.  k_cart_uu3 / k_cart_phi4 / k_cart_residual3 run on it as on a box and write exactly the rows of row_at;
  // cells that do not exist at a lattice position contribute only to rows that are not regular, i.e. to nothing that is
  // written.  The general family keeps the cells that touch any other row (reduced colour lists) and skips the regular rows.
  struct Level3
  {
    double h[3] = {0, 0, 0};
    long long lo[3] = {0, 0, 0}; // lattice position of the box's first node
    int dims[3] = {0, 0, 0};     // nodes of the box
    std::vector<int32_t> node_at, row_at;
    std::vector<double> lam, mu; // per lattice cell (heterogeneous material), else empty
    int64_t n_rows = 0;
  };
  struct PatchPlan3
  {
    std::vector<uint8_t> regular, hang;
    std::vector<int32_t> rows_general;
    std::vector<Level3> levels;
    int64_t n_regular = 0;
  };
  PatchPlan3 plan_patches3d(const pfm_mesh_desc *m, const std::vector<int32_t> &hn_index)
  {
    PatchPlan3 pl;
    const int32_t N = m->n_nodes, NO = m->n_owned_nodes;
    const int64_t NC = m->n_cells;
    if (m->dim != 3 || NC == 0 || getenv("PFM_NO_PATCH"))
      return pl;
    const double *X = m->coords;
    double xmin[3] = {X[0], X[1], X[2]}, xmax[3] = {X[0], X[1], X[2]};
    for (int32_t n = 0; n < N; ++n)
      for (int d = 0; d < 3; ++d)
        {
          xmin[d] = std::min(xmin[d], X[3 * (size_t)n + d]);
          xmax[d] = std::max(xmax[d], X[3 * (size_t)n + d]);
        }
    const double tol = 1e-9 * std::max(xmax[0] - xmin[0], std::max(xmax[1] - xmin[1], xmax[2] - xmin[2]));
    double amax = 0.0;
    for (int d = 0; d < 3; ++d)
      amax = std::max(amax, std::max(std::fabs(xmin[d]), std::fabs(xmax[d])));
    const double geo_tol = 16.0 * 2.220446049250313e-16 * amax;
    struct Lv
    {
      double h[3];
      long long clo[3] = {LLONG_MAX, LLONG_MAX, LLONG_MAX}, chi[3] = {LLONG_MIN, LLONG_MIN, LLONG_MIN}; // box of its cells' positions
    };
    std::vector<Lv> lv;
    std::vector<int8_t> cell_level((size_t)NC, -1);
    std::vector<long long> cpos((size_t)NC * 3, 0);
    for (int64_t k = 0; k < NC; ++k)
      {
        const int32_t *cn = m->cell_nodes + 8 * k;
        const double *p0 = X + 3 * (size_t)cn[0], *p7 = X + 3 * (size_t)cn[7];
        const double h[3] = {p7[0] - p0[0], p7[1] - p0[1], p7[2] - p0[2]};
        bool box = h[0] > tol && h[1] > tol && h[2] > tol;
        // an axis-parallel box in deal.II's vertex order, up to the rounding of the vertex positions (the kernels take
        // J = diag(h)): the midpoints a refinement creates are means of 2, 4 or 8 parents and agree only to a few ulp
        // unless the coordinates are dyadic
        for (int a = 0; a < 8 && box; ++a)
          {
            const double *pa = X + 3 * (size_t)cn[a];
            box = std::fabs(pa[0] - ((a & 1) ? p7[0] : p0[0])) <= geo_tol && std::fabs(pa[1] - ((a & 2) ? p7[1] : p0[1])) <= geo_tol &&
                  std::fabs(pa[2] - ((a & 4) ? p7[2] : p0[2])) <= geo_tol;
          }
        if (!box)
          continue;
        int L = -1;
        for (size_t l = 0; l < lv.size(); ++l)
          if (std::fabs(lv[l].h[0] - h[0]) <= 1e-9 * h[0] && std::fabs(lv[l].h[1] - h[1]) <= 1e-9 * h[1] && std::fabs(lv[l].h[2] - h[2]) <= 1e-9 * h[2])
            L = (int)l;
        if (L < 0)
          {
            if (lv.size() >= 100)
              continue;
            lv.push_back(Lv{});
            for (int d = 0; d < 3; ++d)
              lv.back().h[d] = h[d];
            L = (int)lv.size() - 1;
          }
        bool on = true;
        for (int d = 0; d < 3; ++d)
          {
            const double f = (p0[d] - xmin[d]) / lv[L].h[d];
            const long long i = std::llround(f);
            on = on && std::fabs(f - (double)i) <= 1e-6;
            cpos[3 * (size_t)k + d] = i;
          }
        if (on)
          {
            cell_level[k] = (int8_t)L;
            for (int d = 0; d < 3; ++d)
              {
                lv[L].clo[d] = std::min(lv[L].clo[d], cpos[3 * (size_t)k + d]);
                lv[L].chi[d] = std::max(lv[L].chi[d], cpos[3 * (size_t)k + d]);
              }
          }
      }
    if (lv.empty())
      return pl;
    // incident cells per node: count, common level, the 8 cells by the vertex the node is of them
    std::vector<uint8_t> n_inc((size_t)N, 0), is_parent((size_t)N, 0);
    std::vector<int8_t> node_level((size_t)N, -2);
    std::vector<int32_t> inc((size_t)N * 8, -1);
    for (int64_t k = 0; k < NC; ++k)
      for (int a = 0; a < 8; ++a)
        {
          const int32_t n = m->cell_nodes[8 * k + a];
          if (n_inc[n] < 255)
            ++n_inc[n];
          const int8_t L = cell_level[k];
          node_level[n] = (node_level[n] == -2) ? L : (node_level[n] == L ? L : (int8_t)-1);
          inc[(size_t)n * 8 + a] = (inc[(size_t)n * 8 + a] == -1) ? (int32_t)k : -2;
        }
    if (m->n_hanging > 0)
      for (int64_t j = 0; j < m->hn_ptr[m->n_hanging]; ++j)
        is_parent[m->hn_parents[j]] = 1;
    auto hanging = [&](int32_t n) { return !hn_index.empty() && hn_index[n] >= 0; };
    std::vector<uint8_t> regular((size_t)N, 0);
    std::vector<long long> npos((size_t)N * 3, 0);
    // Regular: every cell around the node that the level's box has room for exists, is of the node's level and has no
    // hanging vertex.  A cell position OUTSIDE the box of the level's cells may be missing: the node then lies on a face of the
    // level lattice -- for a level that reaches the boundary of the domain, on that boundary -- and the row-owner kernels
    // treat it like a face node of a uniform box (fewer than 27 neighbours, cells from index ranges).
    for (int32_t n = 0; n < NO; ++n)
      {
        if (n_inc[n] == 0 || node_level[n] < 0 || hanging(n) || is_parent[n])
          continue;
        const Lv &l = lv[(size_t)node_level[n]];
        bool ok = true;
        long long pn[3] = {0, 0, 0};
        bool have = false;
        for (int a = 0; a < 8 && !have; ++a)
          {
            const int32_t k = inc[(size_t)n * 8 + a];
            if (k >= 0)
              {
                have = true; // the cell of which n is vertex a lies at pn - (a_x, a_y, a_z)
                pn[0] = cpos[3 * (size_t)k] + (a & 1), pn[1] = cpos[3 * (size_t)k + 1] + ((a >> 1) & 1), pn[2] = cpos[3 * (size_t)k + 2] + (a >> 2);
              }
          }
        ok = have;
        for (int a = 0; a < 8 && ok; ++a)
          {
            const int32_t k = inc[(size_t)n * 8 + a];
            const long long cp[3] = {pn[0] - (a & 1), pn[1] - ((a >> 1) & 1), pn[2] - (a >> 2)};
            if (k == -1)
              {
                bool outside = false;
                for (int d = 0; d < 3; ++d)
                  outside = outside || cp[d] < l.clo[d] || cp[d] > l.chi[d];
                ok = outside;
                continue;
              }
            ok = k >= 0 && cpos[3 * (size_t)k] == cp[0] && cpos[3 * (size_t)k + 1] == cp[1] && cpos[3 * (size_t)k + 2] == cp[2];
            for (int b = 0; b < 8 && ok; ++b)
              ok = !hanging(m->cell_nodes[8 * (size_t)k + b]);
          }
        if (ok)
          {
            regular[n] = 1;
            for (int d = 0; d < 3; ++d)
              npos[3 * (size_t)n + d] = pn[d];
          }
      }
    // per level: the lattice of the box of its cells, the tables
    static const long long max_table = getenv("PFM_OVERLAY3_MAX_TABLE") ? atoll(getenv("PFM_OVERLAY3_MAX_TABLE")) : 400000000LL;
    static const long long min_rows = getenv("PFM_OVERLAY3_MIN_ROWS") ? atoll(getenv("PFM_OVERLAY3_MIN_ROWS")) : 64;
    for (size_t L = 0; L < lv.size(); ++L)
      {
        long long lo[3] = {LLONG_MAX, LLONG_MAX, LLONG_MAX}, hi[3] = {LLONG_MIN, LLONG_MIN, LLONG_MIN};
        int64_t cnt = 0;
        for (int32_t n = 0; n < NO; ++n)
          if (regular[n] && node_level[n] == (int8_t)L)
            {
              ++cnt;
              for (int d = 0; d < 3; ++d)
                {
                  lo[d] = std::min(lo[d], npos[3 * (size_t)n + d]);
                  hi[d] = std::max(hi[d], npos[3 * (size_t)n + d]);
                }
            }
        for (int d = 0; d < 3 && cnt; ++d) // the lattice of the level: the nodes of the box of its cells
          {
            lo[d] = lv[L].clo[d];
            hi[d] = lv[L].chi[d] + 1;
          }
        const double vol = cnt ? (double)(hi[0] - lo[0] + 1) * (double)(hi[1] - lo[1] + 1) * (double)(hi[2] - lo[2] + 1) : 0.0;
        if (cnt < min_rows || vol > (double)max_table || vol > 64.0 * (double)cnt)
          {
            // too few rows, or a box that is mostly empty (a thin refined band): these rows stay with the general family
            for (int32_t n = 0; n < NO; ++n)
              if (regular[n] && node_level[n] == (int8_t)L)
                regular[n] = 0;
            continue;
          }
        Level3 lev;
        for (int d = 0; d < 3; ++d)
          {
            lev.h[d] = lv[L].h[d];
            lev.lo[d] = lo[d];
            lev.dims[d] = (int)(hi[d] - lo[d] + 1);
          }
        const long long NX = lev.dims[0], NY = lev.dims[1], NZ = lev.dims[2];
        lev.node_at.assign((size_t)(NX * NY * NZ), -1);
        lev.row_at.assign((size_t)(NX * NY * NZ), -1);
        const bool het = m->cell_lambda && m->cell_mu;
        if (het)
          {
            lev.lam.assign((size_t)((NX - 1) * (NY - 1) * (NZ - 1)), 0.0);
            lev.mu.assign((size_t)((NX - 1) * (NY - 1) * (NZ - 1)), 0.0);
          }
        bool clash = false;
        for (int64_t k = 0; k < NC; ++k)
          if (cell_level[k] == (int8_t)L)
            {
              const long long ci = cpos[3 * (size_t)k] - lev.lo[0], cj = cpos[3 * (size_t)k + 1] - lev.lo[1], ck = cpos[3 * (size_t)k + 2] - lev.lo[2];
              if (ci < 0 || ci >= NX - 1 || cj < 0 || cj >= NY - 1 || ck < 0 || ck >= NZ - 1)
                continue;
              for (int a = 0; a < 8; ++a)
                {
                  int32_t &slot = lev.node_at[(size_t)((ci + (a & 1)) + NX * ((cj + ((a >> 1) & 1)) + NY * (ck + (a >> 2))))];
                  const int32_t n = m->cell_nodes[8 * (size_t)k + a];
                  if (slot == -1)
                    slot = n;
                  else if (slot != n)
                    clash = true; // two nodes at one lattice position (a slit): not a mesh for this overlay
                }
              if (het)
                {
                  lev.lam[(size_t)(ci + (NX - 1) * (cj + (NY - 1) * ck))] = m->cell_lambda[k];
                  lev.mu[(size_t)(ci + (NX - 1) * (cj + (NY - 1) * ck))] = m->cell_mu[k];
                }
            }
        if (clash)
          {
            for (int32_t n = 0; n < NO; ++n)
              if (regular[n] && node_level[n] == (int8_t)L)
                regular[n] = 0;
            continue;
          }
        for (int32_t n = 0; n < NO; ++n)
          if (regular[n] && node_level[n] == (int8_t)L)
            {
              const long long i = npos[3 * (size_t)n] - lev.lo[0], j = npos[3 * (size_t)n + 1] - lev.lo[1], kk = npos[3 * (size_t)n + 2] - lev.lo[2];
              lev.row_at[(size_t)(i + NX * (j + NY * kk))] = n;
              ++lev.n_rows;
            }
        pl.levels.push_back(std::move(lev));
      }
    if (pl.levels.empty())
      return pl;
    for (int32_t n = 0; n < NO; ++n)
      {
        pl.n_regular += regular[n];
        if (!regular[n])
          pl.rows_general.push_back(n);
      }
    pl.hang.assign((size_t)N, 0);
    for (int32_t n = 0; n < N; ++n)
      pl.hang[n] = hanging(n) ? 1 : 0;
    pl.regular.swap(regular);
    return pl;
  }

  // context part: reduced colour lists of the general family, uploads of the level lattices
  void finish_patches3d(pfm_ctx *c, const pfm_mesh_desc *m, PatchPlan3 &pl, const int32_t *hn_index)
  {
    DevView &v = c->v;
    const int32_t NO = m->n_owned_nodes;
    const int64_t NC = m->n_cells;
    if (pl.levels.empty())
      return;
    std::vector<uint8_t> need((size_t)NC);
    parallel_for(NC, [&](int64_t cb, int64_t ce) {
      for (int64_t k = cb; k < ce; ++k)
        {
          uint8_t q = 0;
          for (int a = 0; a < 8; ++a)
            {
              const int32_t n = m->cell_nodes[8 * k + a];
              if (pl.hang[n] || (n < NO && !pl.regular[n]))
                q = 1;
            }
          need[k] = q;
        }
    });
    Colours red; // (see finish_patches2d)
    greedy_colours(NC, 8, m->n_nodes, [&](int64_t cell, int a) { return m->cell_nodes[8 * cell + a]; }, hn_index, need.data(), nullptr, red,
                   c->hanging_coloured ? m->hn_ptr : nullptr, c->hanging_coloured ? m->hn_parents : nullptr);
    if (red.overflow)
      return;
    c->color_ptr_reduced.swap(red.ptr);
    pfm::raw_vector<int32_t> &order_red = red.order;
    c->n_general_cells = (int64_t)c->color_ptr_reduced.back();
    v.row_patch = dev_upload(c, pl.regular.data(), pl.regular.size());
    c->d_color_cells_reduced = dev_upload(c, order_red.data(), order_red.size());
    c->n_rows_general = (int32_t)pl.rows_general.size();
    if (pl.rows_general.empty())
      pl.rows_general.push_back(0);
    c->d_rows_general = dev_upload(c, pl.rows_general.data(), pl.rows_general.size());
    c->n_patch_rows = pl.n_regular;
    for (Level3 &lev : pl.levels)
      {
        pfm_ctx::OverlayLevel ol;
        CartView &cv = ol.cv;
        cv.NX = lev.dims[0];
        cv.NY = lev.dims[1];
        cv.NZ = lev.dims[2];
        for (int d = 0; d < 3; ++d)
          {
            cv.o0[d] = 0; // every node of the level lattice may own a row (CartView::row_of_box says which do)
            cv.o1[d] = lev.dims[d] - 1;
            cv.h[d] = lev.h[d];
          }
        cv.local_of_box = dev_upload(c, lev.node_at.data(), lev.node_at.size());
        cv.row_of_box = dev_upload(c, lev.row_at.data(), lev.row_at.size());
        cv.owned_lex = 0;
        cv.cell_lam = cv.cell_mu = nullptr;
        if (!lev.lam.empty())
          {
            cv.cell_lam = dev_upload(c, lev.lam.data(), lev.lam.size());
            cv.cell_mu = dev_upload(c, lev.mu.data(), lev.mu.size());
          }
        ol.d_scal = dev_alloc<unsigned char>(c, PFM_SCAL_BYTES);
        ol.n_rows = lev.n_rows;
        c->levels3.push_back(ol);
      }
    c->overlay3_rows_valid = false;
  }

  struct PhaseClock
  {
    const bool on = getenv("PFM_CTX_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char *what)
    {
      if (!on)
        return;
      const auto t1 = std::chrono::steady_clock::now();
      fprintf(stderr, "[pfm_ctx_create] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
      t0 = t1;
    }
  };
} // namespace

namespace
{
  // PFM_ABORT_TRACE=<file>: a backtrace of the aborting thread into that file (debugging aid: a test runner that captures
  // stderr hides what libstdc++ or the HIP runtime say before they abort)
  void abort_trace_handler(int)
  {
    const char *path = getenv("PFM_ABORT_TRACE");
    const int fd = path ? open(path, O_WRONLY | O_CREAT | O_APPEND, 0644) : 2;
    void *frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, fd >= 0 ? fd : 2);
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
  }
  void install_abort_trace()
  {
    static std::once_flag once;
    std::call_once(once, [] {
      if (getenv("PFM_ABORT_TRACE"))
        {
          signal(SIGABRT, abort_trace_handler);
          signal(SIGSEGV, abort_trace_handler);
        }
    });
  }
} // namespace

extern "C"
{
  int pfm_ctx_create(pfm_ctx **out, const pfm_mesh_desc *m, int device)
  {
    install_abort_trace();
    PhaseClock clk;
    H2dBatch batch; // uploads are ordered on the null stream; the build ends with a device synchronisation
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

    {
      std::atomic<bool> in_range{true};
      parallel_for(NC * nv, [&](int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i)
          if (m->cell_nodes[i] < 0 || m->cell_nodes[i] >= N)
            {
              in_range = false;
              return;
            }
      });
      if (!in_range)
        return fail(c, PFM_ERR_BAD_ARG, "cell_nodes out of range");
    }

    clk.mark("argument checks");
    Lattice &lattice = c->lat;
    bool lattice_ok = false;
    // The cell table and the coordinates go to the device as the host handed them over (cell-major / node-major) and are
    // transposed to SoA by a kernel (a host transposition of 8e7 + 3e7 entries cost 0.25 s at 1e7 cells).  Large meshes: on
    // a second thread, next to the host passes below that read the same tables (lattice detection: 20 ms at 1e7 cells).
    struct MeshUpload
    {
      int32_t *raw = nullptr, *conn = nullptr;
      double *rawx = nullptr, *xs = nullptr;
      int rct = PFM_OK, rcx = PFM_OK;
      hipError_t err = hipSuccess;
      const char *what = "";
    } up;
    auto upload_mesh = [&]() {
      H2dBatch in_batch;
      auto bad = [&](hipError_t e, const char *what) {
        up.err = e;
        up.what = what;
      };
      if (hipSetDevice(device) != hipSuccess)
        return bad(hipGetLastError(), "hipSetDevice");
      const size_t cb = sizeof(int32_t) * std::max<size_t>((size_t)NC * nv, 1), xb = sizeof(double) * (size_t)N * dim;
      if (scratch_acquire((void **)&up.raw, cb) != hipSuccess || hipMalloc((void **)&up.conn, cb) != hipSuccess ||
          scratch_acquire((void **)&up.rawx, xb) != hipSuccess || hipMalloc((void **)&up.xs, xb) != hipSuccess)
        return bad(hipGetLastError(), "hipMalloc");
      hipError_t e2 = NC > 0 ? h2d(up.raw, m->cell_nodes, sizeof(int32_t) * (size_t)NC * nv) : hipSuccess;
      if (e2 != hipSuccess)
        return bad(e2, "hipMemcpy H2D");
      up.rct = launch_aos_to_soa_i32(up.raw, up.conn, NC, nv, nullptr);
      e2 = h2d(up.rawx, m->coords, xb);
      if (e2 != hipSuccess)
        return bad(e2, "hipMemcpy H2D");
      up.rcx = launch_aos_to_soa_f64(up.rawx, up.xs, N, dim, nullptr);
    };
    std::thread upload_thread;
    struct UploadJoiner // an early return or an exception must not leave the thread running, nor its buffers behind
    {
      std::thread &t;
      MeshUpload &u;
      bool taken = false;
      ~UploadJoiner()
      {
        if (t.joinable())
          t.join();
        if (!taken)
          {
            scratch_release(u.raw);
            scratch_release(u.rawx);
            for (void *q : {(void *)u.conn, (void *)u.xs})
              if (q)
                (void)hipFree(q);
          }
      }
    } upload_joiner{upload_thread, up};
    try
      {
        // constraint bytes, the nodal state and the status word: one allocation, one clear (a hipMalloc costs 0.1-3 ms)
        {
          const size_t sb = ((size_t)N * sizeof(double) + 255) & ~(size_t)255, fb = ((size_t)N + 255) & ~(size_t)255;
          const size_t total = fb + (size_t)(dim + 3) * sb + 256;
          char *slab = dev_alloc<char>(c, total);
          e = hipMemsetAsync(slab, 0, total, nullptr);
          if (e != hipSuccess)
            throw HipFail{e, "hipMemset"};
          v.node_flags = reinterpret_cast<uint8_t *>(slab);
          char *at = slab + fb;
          auto take = [&]() {
            double *q = reinterpret_cast<double *>(at);
            at += sb;
            return q;
          };
          for (int d = 0; d < 3; ++d)
            v.u[d] = (d < dim) ? take() : nullptr;
          v.phi = take();
          v.phi_old = take();
          v.phi_oldold = take();
          v.status = reinterpret_cast<int *>(at);
        }
        clk.mark("state buffers");
        if (NC > (1 << 20))
          upload_thread = std::thread(upload_mesh);
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

        // ---- node graph over the constraint-resolved cells, rows = owned nodes, columns ascending by local node id
        // (ghost nodes, numbered after the owned ones, come last: the order of a host CSR sorted by local column id).
        bool device_graph = false;
        lattice_ok = detect_lattice(m, lattice);
        clk.mark("detect_lattice");
        if (lattice_ok)
          lattice_graph(c, dim, NO, lattice);
        else
          device_graph = true; // general mesh: built on the device below, from the uploaded cell table (pfm_graph.hip)
        std::vector<int32_t> &nadj = c->h_nadj;
        clk.mark("node graph");

        // Colour classes of the general family (greedy_colours) only read host tables: on a general mesh they are computed by a
        // host thread NEXT TO the uploads and the device build of the node graph -- and given up as soon as the overlay plan
        // (below, on a thread of its own) turns out non-empty: the overlay needs the classes of the few cells it leaves to
        // the general family only (finish_patches2d / 3d), the lists over all cells are then made on first use.
        const int32_t *hn_idx = hn_index.empty() ? nullptr : hn_index.data();
        auto node_of = [m, nv](int64_t cell, int a) { return m->cell_nodes[cell * nv + a]; };
        // PFM_HANGING_COLOURED=1 (read at every pfm_ctx_create): the 3-D cells at hanging vertices in plain colour classes
        // (greedy_colours) instead of the one class with FP64 atomics -- every assembly of such a mesh is then bitwise
        // reproducible; it costs some thirty small launches per assembly (1.1e6-cell overlay mesh: Jacobian 4.45 -> 5.1 ms,
        // residual only 0.6 -> 1.3 ms), so the default keeps the atomic class.  Needs the records of DevView::cres, i.e.
        // hanging nodes with at most four parents.
        bool hanging_coloured = dim == 3 && hn_idx && !lattice_ok && getenv("PFM_HANGING_COLOURED") != nullptr;
        for (int32_t k = 0; k < m->n_hanging && hanging_coloured; ++k)
          hanging_coloured = m->hn_ptr[k + 1] - m->hn_ptr[k] <= 4;
        c->hanging_coloured = hanging_coloured;
        c->n_hanging = m->n_hanging;
        const int64_t *col_hn_ptr = hanging_coloured ? m->hn_ptr : nullptr;
        const int32_t *col_hn_parents = hanging_coloured ? m->hn_parents : nullptr;
        Colours full;
        std::atomic<bool> colour_cancel{false};
        std::exception_ptr colour_err;
        auto colour_classes = [&]() {
          try
            {
              if (lattice_ok)
                {
                  // (2-D boxes; a 3-D box sorts its lists on the device when asked, below)  parity of the cell's lattice position
                  const int64_t NX = lattice.NX, NY = lattice.NY;
                  pfm::raw_vector<uint8_t> col((size_t)NC);
                  parallel_for(NC, [&](int64_t cb, int64_t ce) {
                    for (int64_t cell = cb; cell < ce; ++cell)
                      {
                        const int64_t b0 = lattice.box_of_local[m->cell_nodes[cell * nv]];
                        const int64_t i = b0 % NX, j = (b0 / NX) % NY, k = b0 / (NX * NY);
                        col[cell] = (uint8_t)((i & 1) + 2 * (j & 1) + 4 * (k & 1));
                      }
                  });
                  const int n_col = dim == 3 ? 8 : 4, nk = n_col + 1;
                  full.ptr.assign((size_t)n_col + 2, 0);
                  full.order.resize((size_t)std::max<int64_t>(NC, 1));
                  const int nt = chunks_for(NC, 65536);
                  std::vector<long long> hist((size_t)nt * nk, 0);
                  parallel_chunks(nt, [&](int t) {
                    long long *hh = hist.data() + (size_t)t * nk;
                    for (int64_t cell = NC * t / nt; cell < NC * (t + 1) / nt; ++cell)
                      ++hh[col[cell]];
                  });
                  for (int k = 0; k < nk; ++k)
                    {
                      long long at = full.ptr[k];
                      for (int t = 0; t < nt; ++t)
                        {
                          const long long cnt = hist[(size_t)t * nk + k];
                          hist[(size_t)t * nk + k] = at;
                          at += cnt;
                        }
                      full.ptr[(size_t)k + 1] = at;
                    }
                  parallel_chunks(nt, [&](int t) {
                    long long *fill = hist.data() + (size_t)t * nk;
                    for (int64_t cell = NC * t / nt; cell < NC * (t + 1) / nt; ++cell)
                      full.order[(size_t)fill[col[cell]]++] = (int32_t)cell;
                  });
                }
              else
                greedy_colours(NC, nv, N, node_of, hn_idx, nullptr, &colour_cancel, full, col_hn_ptr, col_hn_parents);
            }
          catch (...)
            {
              colour_err = std::current_exception();
            }
        };
        std::thread colour_thread;
        if (!lattice_ok && NC > 65536)
          colour_thread = std::thread(colour_classes);
        // cartesian overlay of a 2-D mesh: the host classification (8 ms at 2.7e5 cells) on its own thread as well
        PatchPlan patch_plan;
        PatchPlan3 patch_plan3;
        std::exception_ptr patch_err;
        auto plan_patches = [&]() {
          try
            {
              if (dim == 2)
                patch_plan = plan_patches2d(m, hn_index);
              else if (!lattice_ok) // (a uniform 3-D box takes the cartesian family as a whole; no stress split in 3-D)
                patch_plan3 = plan_patches3d(m, hn_index);
            }
          catch (...)
            {
              patch_err = std::current_exception();
            }
        };
        std::thread patch_thread;
        if (NC > 65536)
          patch_thread = std::thread(plan_patches);
        struct Joiner
        {
          std::thread &t;
          ~Joiner()
          {
            if (t.joinable())
              t.join();
          }
        } colour_joiner{colour_thread}, patch_joiner{patch_thread}; // an exception on the way must not leave a running thread behind

        // ---- device mirrors (SoA)
        {
          if (upload_thread.joinable())
            upload_thread.join();
          else
            upload_mesh();
          upload_joiner.taken = true;
          clk.mark("  conn + coords h2d");
          int32_t *raw = up.raw, *conn = up.conn;
          double *rawx = up.rawx, *xs = up.xs;
          for (void *q : {(void *)conn, (void *)xs})
            if (q)
              c->allocs.push_back(q);
          c->device_bytes += (int64_t)(sizeof(int32_t) * (size_t)NC * nv + sizeof(double) * (size_t)N * dim);
          if (up.err != hipSuccess)
            {
              scratch_release(raw);
              scratch_release(rawx);
              throw HipFail{up.err, up.what};
            }
          const int rct = up.rct, rcx = up.rcx;
          int rcg = PFM_OK;
          v.hn_index = nullptr;
          v.hn_ptr = nullptr;
          v.hn_parents = nullptr;
          v.hn_weights = nullptr;
          try
            {
              if (m->n_hanging > 0)
                {
                  v.hn_index = dev_upload(c, hn_index.data(), hn_index.size());
                  std::vector<long long> hp(m->hn_ptr, m->hn_ptr + m->n_hanging + 1);
                  v.hn_ptr = dev_upload(c, hp.data(), hp.size());
                  v.hn_parents = dev_upload(c, m->hn_parents, (size_t)hp.back());
                  v.hn_weights = dev_upload(c, m->hn_weights, (size_t)hp.back());
                }
              clk.mark("  hanging tables");
              if (device_graph && rct == PFM_OK)
                {
                  // node graph of a general mesh on the device, from the cell table as the host handed it over
                  long long *d_ptr = dev_alloc<long long>(c, (size_t)NO + 1);
                  struct ScratchGuard // the incidence lists are freed on every way out, a throwing dev_alloc included
                  {
                    GraphScratch sc;
                    ~ScratchGuard() { graph_build_free(sc); }
                  } g;
                  long long total = 0;
                  rcg = graph_build_begin(raw, NC, nv, NO, v.hn_index, v.hn_ptr, v.hn_parents, d_ptr, g.sc, total, nullptr);
                  clk.mark("  graph rows counted");
                  if (rcg == PFM_OK)
                    {
                      int32_t *d_adj = dev_alloc<int32_t>(c, (size_t)std::max<long long>(total, 1));
                      rcg = graph_build_rows(raw, NC, nv, NO, v.hn_index, v.hn_ptr, v.hn_parents, d_ptr, d_adj, g.sc, nullptr);
                      clk.mark("  graph rows filled");
                      if (rcg == PFM_OK)
                        {
                          v.nadj_ptr = d_ptr;
                          v.nadj = d_adj;
                          c->nadj_total = total;
                          c->graph_dev_only = true; // the host copy of rows and row pointers is fetched when a pattern query asks for it
                        }
                    }
                }
            }
          catch (...)
            {
              scratch_release(raw);
              scratch_release(rawx);
              throw;
            }
          clk.mark("  row pointers to the host");
          const hipError_t es = hipDeviceSynchronize();
          clk.mark("  device synchronize");
          scratch_release(raw);
          scratch_release(rawx);
          if (rcg == PFM_ERR_UNSUPPORTED)
            return fail(c, PFM_ERR_UNSUPPORTED, "node with more than 254 neighbours");
          if (rct != PFM_OK || rcx != PFM_OK || rcg != PFM_OK || es != hipSuccess)
            throw HipFail{hipGetLastError(), "mesh table upload / node graph"};
          v.conn = conn;
          v.coords = xs;
        }
        clk.mark("conn + coords upload");
        if (patch_thread.joinable())
          patch_thread.join();
        else
          plan_patches();
        if (patch_err)
          std::rethrow_exception(patch_err);
        clk.mark("overlay plan (wait)");
        std::vector<uint8_t> ring;
        const bool any_ring = !lattice_ok && ring_cells(NC, nv, N, node_of, hn_idx, m->hn_ptr, m->hn_parents, ring, hanging_coloured);
        v.cell_ring = any_ring ? dev_upload(c, ring.data(), ring.size()) : nullptr;
        v.color_cells = nullptr;
        v.hcell = nullptr;
        v.cslot_h = nullptr;
        v.cres = nullptr;
        if (hn_idx && !lattice_ok)
          {
            // slot table of the cells at hanging vertices (DevView::cslot_h), filled next to cslot by launch_build_cslot
            const int MPH = dim == 3 ? 4 : 2;
            bool fits = true;
            for (int32_t k = 0; k < m->n_hanging; ++k)
              fits = fits && m->hn_ptr[k + 1] - m->hn_ptr[k] <= MPH;
            if (fits)
              {
                const std::vector<int32_t> hcells = parallel_select(0, NC, [&](int64_t cell) {
                  bool h = false;
                  for (int a = 0; a < nv; ++a)
                    h = h || hn_idx[m->cell_nodes[cell * nv + a]] >= 0;
                  return h;
                });
                if (!hcells.empty())
                  {
                    pfm::raw_vector<int32_t> hcell((size_t)NC);
                    parallel_for(NC, [&](int64_t b, int64_t e) { std::fill(hcell.begin() + b, hcell.begin() + e, -1); });
                    parallel_for((int64_t)hcells.size(), [&](int64_t b, int64_t e) {
                      for (int64_t i = b; i < e; ++i)
                        hcell[(size_t)hcells[(size_t)i]] = (int32_t)i;
                    });
                    v.hcell = dev_upload(c, hcell.data(), hcell.size());
                    v.cslot_h = dev_alloc<uint8_t>(c, hcells.size() * (size_t)(nv * MPH * nv * MPH));
                    if (dim == 3)
                      {
                        v.cres = dev_alloc<uint8_t>(c, hcells.size() * (size_t)pfm::PFM_CRES_BYTES);
                        c->n_hcells = (int64_t)hcells.size();
                        // round 6 default: scratch + ordered gather instead of FP64 atomics (ensure_hang_gather, on first use);
                        // PFM_HANGING_ATOMIC=1 keeps the atomic class, PFM_HANGING_COLOURED=1 the colour classes of round 5
                        c->hang_gather = !hanging_coloured && getenv("PFM_HANGING_ATOMIC") == nullptr;
                      }
                  }
              }
          }
        clk.mark("ring cells");
        finish_patches2d(c, m, patch_plan, hn_idx);
        finish_patches3d(c, m, patch_plan3, hn_idx);
        const bool overlay = c->n_patch_blocks > 0 || !c->levels3.empty();
        clk.mark("cartesian overlay");
        if (lattice_ok && dim == 3)
          {
            // uniform 3-D box: the cartesian family runs it; the lists of the general family (eight parity classes, their sizes
            // known from the box) are sorted on the device when that family is first asked for (ensure_general_tables)
            c->color_ptr.assign(10, 0);
            for (int k = 0; k < 8; ++k)
              {
                auto half = [](int n, int odd) { return (long long)(odd ? n / 2 : (n + 1) / 2); };
                c->color_ptr[(size_t)k + 1] = c->color_ptr[k] + half(lattice.nc[0], k & 1) * half(lattice.nc[1], (k >> 1) & 1) * half(lattice.nc[2], (k >> 2) & 1);
              }
            c->color_ptr[9] = c->color_ptr[8];
            c->colours_lazy = true;
          }
        else if (overlay)
          {
            colour_cancel = true; // (the thread, if it runs, stops at its next look)
            if (colour_thread.joinable())
              colour_thread.join();
            c->full_colours_lazy = true;
          }
        else
          {
            if (colour_thread.joinable())
              colour_thread.join();
            else
              colour_classes();
            if (colour_err)
              std::rethrow_exception(colour_err);
            v.color_cells = dev_upload(c, full.order.data(), full.order.size());
            c->color_ptr.swap(full.ptr);
            if (full.overflow)
              {
                ring.assign((size_t)NC, 1); // (never seen: a node of more than 62 cells) every cell adds atomically
                v.cell_ring = dev_upload(c, ring.data(), ring.size());
              }
          }
        clk.mark("colour classes");
        v.cell_lambda = v.cell_mu = nullptr;
        if (m->cell_lambda && m->cell_mu)
          {
            v.cell_lambda = dev_upload(c, m->cell_lambda, (size_t)NC);
            v.cell_mu = dev_upload(c, m->cell_mu, (size_t)NC);
          }
        if (c->graph_positional)
          {
            long long *d_ptr = dev_alloc<long long>(c, (size_t)NO + 1);
            if (launch_lattice_row_ptr(d_ptr, lattice.NX, lattice.NY, lattice.NZ, nullptr) != PFM_OK)
              throw HipFail{hipGetLastError(), "lattice row pointers"};
            v.nadj_ptr = d_ptr;
          }
        else if (!c->graph_dev_only)
          v.nadj_ptr = dev_upload(c, c->h_nadj_ptr.data(), c->h_nadj_ptr.size());
        // node graph columns + slot table of the general kernel family (position of vertex b's node in the row of
        // vertex a's node, searched on the device: 64 row searches per hex, 8 s on one host core at 1e7 cells); a
        // lattice context builds them when the general family is first used
        if (!c->graph_dev_only)
          v.nadj = nullptr;
        v.cslot = nullptr;
        c->general_ready = !c->graph_lazy;
        if (c->general_ready)
          {
            if (!c->graph_dev_only)
              v.nadj = dev_upload(c, nadj.data(), nadj.size());
            clk.mark("graph upload");
            v.cslot = dev_alloc<uint8_t>(c, (size_t)NC * nv * nv);
            if (launch_build_cslot(v, nullptr) != PFM_OK)
              throw HipFail{hipGetLastError(), "cslot kernel"};
          }
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
        clk.mark("cslot launch, state buffers");
        c->cart_ok = lattice_ok && build_cart(c, m, lattice);
        clk.mark("build_cart");
        if (!c->cart_ok)
          {
            ensure_general_tables(c); // needs the lattice tables when the graph is still lazy
            c->lat = Lattice{};       // the host lattice tables are only kept on the cartesian path
          }
        if (hipDeviceSynchronize() != hipSuccess)
          throw HipFail{hipGetLastError(), "context build"};
        c->d_scal = dev_alloc<unsigned char>(c, PFM_SCAL_BYTES);
        clk.mark("device synchronize");
      }
    catch (const HipFail &f)
      {
        return hipfail(c, f.e, f.what);
      }
    c->kernel_path = c->cart_ok ? 1 : ((c->n_patch_blocks > 0 || !c->levels3.empty()) ? 3 : 0); // 3: general family + cartesian overlay
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
    for (void *q : {(void *)c->d_send_all, (void *)c->d_recv_all, (void *)c->d_send_ptr, (void *)c->d_recv_ptr,
                    (void *)c->d_halo_send, (void *)c->d_halo_recv, (void *)c->cv.patch_idx, (void *)c->cv.patch_val,
                    (void *)c->cv.patch_count})
      if (q)
        (void)hipFree(q);
    if (c->atomic_stream)
      (void)hipStreamDestroy(c->atomic_stream);
    if (c->ev_atomic)
      (void)hipEventDestroy(c->ev_atomic);
    for (hipStream_t st : c->ov_streams)
      (void)hipStreamDestroy(st);
    for (hipEvent_t ev : c->ov_events)
      (void)hipEventDestroy(ev);
    if (c->ov_fork)
      (void)hipEventDestroy(c->ov_fork);
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
    for (auto &hp : c->host_pins) // the arrays belong to the caller; only the page lock is ours
      (void)hipHostUnregister(hp.p);
    if (c->copy_stream)
      {
        (void)hipStreamDestroy(c->copy_stream);
        (void)hipEventDestroy(c->ev_copy);
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
    // the reference gates the split on decompose_stress_matrix alone (cracks.cc:2294): decompose_stress_rhs > 0
    // without it multiplies a zero stress_term_minus, i.e. is a plain assembly and valid in 3-D
    if (c->v.dim == 3 && p->decompose_stress_matrix > 0 && p->timestep_number > 0)
      return fail(c, PFM_ERR_UNSUPPORTED,
                  "stress split is 2-D only in the reference (cracks.cc:1685-1690)");
    c->prm = *p;
    c->have_params = true;
    c->scal_dirty = true; // the per-launch scalar tables of the cartesian Jacobian kernels follow the parameters
    for (auto &lv : c->levels3)
      lv.scal_dirty = true;
    return PFM_OK;
  }

  static pfm_ctx::HostPin *find_pin(pfm_ctx *c, const void *p, size_t bytes);

  // Host -> device copy of a buffer the CALLER owns.  Page-locked through pfm_host_register: an asynchronous DMA on `s`.
  // Anything else goes through the library's own staging buffers (h2d) and is complete on return: the runtime's path for
  // pageable memory pins the caller's pages on the fly, and on this stack that faulted intermittently ("Memory access fault by
  // GPU ... on address <host heap>") when such pages had been registered and unregistered before (tests/test_gpu_cart.py's
  // host-pointer cases followed by the 216^3 test).
  static hipError_t h2d_user(pfm_ctx *c, void *d, const void *h, size_t bytes, hipStream_t s)
  {
    if (bytes == 0)
      return hipSuccess;
    if (find_pin(c, h, bytes))
      return hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
    // the staged copy runs on the NULL stream, which does not order itself behind a hipStreamNonBlocking stream installed
    // through pfm_ctx_set_stream (torch side streams are of that kind): kernels still queued on `s` may read what this copy
    // overwrites (node_flags of an assembly in flight, the staging vectors of the last state) -- wait for them first
    const hipError_t es = hipStreamSynchronize(s);
    if (es != hipSuccess)
      return es;
    return h2d(d, h, bytes);
  }

  // The other direction.  Page-locked: asynchronous on `s`.  Pageable and up to 64 MB (glibc serves such blocks from the
  // heap, where registered pages may have lived): through the staging buffers, complete on return.  Larger pageable arrays
  // (always mappings of their own) take the runtime's path.
  static hipError_t d2h_user(pfm_ctx *c, void *h, const void *d, size_t bytes, hipStream_t s)
  {
    if (bytes == 0)
      return hipSuccess;
    if (find_pin(c, h, bytes) || bytes > ((size_t)64 << 20))
      return hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
    return d2h_staged(h, d, bytes, s);
  }

  int pfm_set_constraints(pfm_ctx *c, const uint8_t *node_flags)
  {
    if (!c || !node_flags)
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    hipError_t e = h2d_user(c, const_cast<uint8_t *>(c->v.node_flags), node_flags, (size_t)c->v.n_nodes, c->stream);
    // capacity of the deferred patch list of the cartesian 3-D Jacobian (counted while the copy is in flight)
    if (c->cart_ok && c->v.dim == 3)
      {
        std::atomic<int64_t> nf{0};
        parallel_for(c->v.n_owned, [&](int64_t nb, int64_t ne) {
          int64_t k = 0;
          for (int64_t n = nb; n < ne; ++n)
            k += __builtin_popcount(node_flags[n] & 7u);
          nf += k;
        });
        c->n_flag_u = nf;
      }
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
    try
      {
        ensure_host_graph(const_cast<pfm_ctx *>(c)); // a cache: the pattern itself does not change
      }
    catch (const std::bad_alloc &)
      {
        return PFM_ERR_NOMEM;
      }
    catch (const HipFail &)
      {
        return PFM_ERR_HIP;
      }
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

  // Adopt the caller's CSR pattern of one block.  The pattern must be the canonical one up to the order of the
  // entries within a row, and that order must keep the structure (node, component): the ncc column components of a
  // neighbour node adjacent and ascending -- what any CSR sorted by local column id has.  The node order of the rows
  // becomes the order of the context's node graph: every kernel family addresses values through it.
  static int pattern_bind_impl(pfm_ctx *c, int block, const int64_t *rp64, const int32_t *rp32, const int32_t *colind)
  {
    if (!c || block < 0 || block >= c->n_blocks || (!rp64 && !rp32) || !colind)
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    const int dim = c->v.dim;
    int ncr, ncc;
    if (c->v.layout == PFM_LAYOUT_INTERLEAVED)
      ncr = ncc = dim + 1;
    else
      {
        ncr = (block == 0 || block == 1) ? dim : 1;
        ncc = (block == 0 || block == 2) ? dim : 1;
      }
    const int32_t NO = c->v.n_owned;
    auto rp = [&](int64_t r) -> int64_t { return rp64 ? rp64[r] : (int64_t)rp32[r]; };
    if (rp(0) != 0 || rp(c->block_rows(block)) != c->block_nnz(block))
      return fail(c, PFM_ERR_BAD_ARG, "pfm_pattern_bind: row pointers do not describe this block (size mismatch)");
    try
      {
        ensure_host_graph(c);
      }
    catch (const std::bad_alloc &)
      {
        return fail(c, PFM_ERR_NOMEM, "host allocation failed");
      }
    catch (const HipFail &f)
      {
        return hipfail(c, f.e, f.what);
      }
    std::vector<int32_t> order(c->h_nadj.size());
    std::atomic<int> bad{0}; // 1: structure, 2: column set
    std::atomic<bool> changed{false};
    parallel_for(NO, [&](int64_t nb, int64_t ne) {
      std::vector<int32_t> a, b;
      for (int64_t n = nb; n < ne && !bad; ++n)
        {
          const long long off = c->h_nadj_ptr[n];
          const int deg = (int)(c->h_nadj_ptr[n + 1] - off);
          for (int ci = 0; ci < ncr; ++ci)
            {
              const int64_t r = n * ncr + ci, b0 = rp(r);
              if (rp(r + 1) - b0 != (int64_t)deg * ncc)
                {
                  bad = 1;
                  return;
                }
              for (int sl = 0; sl < deg; ++sl)
                {
                  const int32_t col0 = colind[b0 + (int64_t)sl * ncc];
                  const int32_t q = col0 / ncc;
                  for (int d = 0; d < ncc; ++d)
                    if (colind[b0 + (int64_t)sl * ncc + d] != q * ncc + d)
                      {
                        bad = 1;
                        return;
                      }
                  if (ci == 0)
                    order[off + sl] = q;
                  else if (order[off + sl] != q)
                    {
                      bad = 1;
                      return;
                    }
                }
            }
          a.assign(order.begin() + off, order.begin() + off + deg);
          b.assign(c->h_nadj.begin() + off, c->h_nadj.begin() + off + deg);
          if (a != b)
            {
              changed = true;
              std::sort(a.begin(), a.end());
              std::sort(b.begin(), b.end());
              if (a != b)
                {
                  bad = 2;
                  return;
                }
            }
        }
    });
    if (bad == 1)
      return fail(c, PFM_ERR_BAD_ARG, "pfm_pattern_bind: rows must keep the (node, component) structure of the canonical pattern");
    if (bad == 2)
      return fail(c, PFM_ERR_BAD_ARG, "pfm_pattern_bind: a row couples other nodes than the mesh does (see pfm_pattern_get)");
    if (changed)
      {
        for (int b = 0; b < c->n_blocks; ++b)
          if (b != block && c->pattern_bound[b])
            return fail(c, PFM_ERR_UNSUPPORTED, "pfm_pattern_bind: all blocks must order the neighbour nodes of a row alike");
        try
          {
            c->h_nadj.swap(order);
            if (c->general_ready) // else: built from the new order when the general family is first used
              {
                if (!c->h_nadj.empty() &&
                    hipMemcpy(const_cast<int32_t *>(c->v.nadj), c->h_nadj.data(), c->h_nadj.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
                  throw HipFail{hipGetLastError(), "hipMemcpy nadj"};
                c->patch_slots_valid = false; // the regular rows' slots follow the bound order
                c->overlay3_rows_valid = false;
                if (launch_build_cslot(c->v, nullptr) != PFM_OK || hipDeviceSynchronize() != hipSuccess)
                  throw HipFail{hipGetLastError(), "cslot kernel"};
              }
            if (c->cart_ok)
              {
                std::vector<uint32_t> mask;
                std::vector<uint8_t> perm;
                bool any_perm = false;
                if (!row_order_tables(c, dim, NO, c->lat, mask, perm, any_perm))
                  return fail(c, PFM_ERR_BAD_ARG, "pfm_pattern_bind: lattice rows inconsistent");
                upload_row_tables(c, mask, perm, any_perm);
              }
          }
        catch (const HipFail &f)
          {
            return hipfail(c, f.e, f.what);
          }
      }
    c->pattern_bound[block] = true;
    return PFM_OK;
  }

  int pfm_pattern_bind(pfm_ctx *c, int block, const int64_t *rowptr, const int32_t *colind)
  {
    return pattern_bind_impl(c, block, rowptr, nullptr, colind);
  }

  int pfm_pattern_bind_i32(pfm_ctx *c, int block, const int32_t *rowptr, const int32_t *colind)
  {
    return pattern_bind_impl(c, block, nullptr, rowptr, colind);
  }

  // sol / old / oldold -> node state; old == oldold == nullptr: solution only (pfm_state_set_solution)
  static int state_set_impl(pfm_ctx *c, const double *sol, const double *old, const double *oldold, int on_device)
  {
    if (!c)
      return PFM_ERR_BAD_ARG;
    if (c->n_owned_dofs() == 0)
      return PFM_OK; // a rank that owns nothing: its node state comes from the ghost import alone
    const bool sol_only = !old && !oldold;
    if (!sol || (!sol_only && (!old || !oldold)))
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    const int nvec = sol_only ? 1 : 3;
    const double *src[3] = {sol, old, oldold};
    const double *d[3] = {sol, old, oldold};
    if (!on_device)
      {
        const size_t bytes = sizeof(double) * (size_t)c->n_owned_dofs();
        for (int k = 0; k < nvec; ++k)
          {
            if (!c->d_stage_vec[k])
              {
                hipError_t e = hipMalloc((void **)&c->d_stage_vec[k], std::max<size_t>(bytes, 8));
                if (e != hipSuccess)
                  return hipfail(c, e, "hipMalloc stage");
                c->device_bytes += (int64_t)bytes;
              }
            hipError_t e = h2d_user(c, c->d_stage_vec[k], src[k], bytes, c->stream);
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

  int pfm_state_set(pfm_ctx *c, const double *sol, const double *old, const double *oldold, int on_device)
  {
    if (!c || (c->n_owned_dofs() != 0 && (!old || !oldold)))
      return PFM_ERR_BAD_ARG;
    return state_set_impl(c, sol, old, oldold, on_device);
  }

  int pfm_state_set_solution(pfm_ctx *c, const double *sol, int on_device) { return state_set_impl(c, sol, nullptr, nullptr, on_device); }

  // Gather tables and scratch of the cells at hanging vertices (DevView::hs_*, pfm_ctx::d_hg_*), from the records of
  // DevView::cres (built on the device): every (cell, resolved node) and (cell, hanging vertex) pair is listed under its
  // destination row, rows ascending, the entries of a row in ascending (cell, index) order -- the order k_hanging_gather adds in.
  // false: some cell has more than 16 resolved nodes (no record): the context keeps the atomic class.
  static bool ensure_hang_gather(pfm_ctx *c)
  {
    if (c->hang_gather_ready)
      return true;
    if (!c->hang_gather || !c->v.cres || c->n_hcells == 0)
      return false;
    (void)hipSetDevice(c->device);
    const int64_t nh = c->n_hcells;
    pfm::raw_vector<uint8_t> rec((size_t)nh * pfm::PFM_CRES_BYTES);
    if (hipMemcpy(rec.data(), c->v.cres, rec.size(), hipMemcpyDeviceToHost) != hipSuccess)
      throw HipFail{hipGetLastError(), "cres D2H"};
    const int32_t NO = c->v.n_owned;
    std::vector<int32_t> cnt((size_t)NO + 1, 0);
    std::vector<long long> off((size_t)nh + 1, 0);
    for (int64_t hc = 0; hc < nh; ++hc)
      {
        const uint8_t *r = rec.data() + (size_t)hc * pfm::PFM_CRES_BYTES;
        const int R = r[pfm::PFM_CRES_R];
        if (R > 16)
          {
            c->hang_gather = false;
            return false;
          }
        off[(size_t)hc + 1] = off[(size_t)hc] + (long long)R * R * 13;
        const int32_t *node = reinterpret_cast<const int32_t *>(r), *hv = reinterpret_cast<const int32_t *>(r + pfm::PFM_CRES_HV);
        for (int i = 0; i < R; ++i)
          if (node[i] < NO)
            ++cnt[(size_t)node[i]];
        for (int a = 0; a < 8; ++a)
          if (hv[a] >= 0 && hv[a] < NO)
            ++cnt[(size_t)hv[a]];
      }
    std::vector<int32_t> rows, row_of((size_t)NO, -1);
    std::vector<long long> ptr(1, 0);
    for (int32_t n = 0; n < NO; ++n)
      if (cnt[(size_t)n])
        {
          row_of[(size_t)n] = (int32_t)rows.size();
          rows.push_back(n);
          ptr.push_back(ptr.back() + cnt[(size_t)n]);
        }
    std::vector<pfm::HgEntry> list((size_t)std::max<long long>(ptr.back(), 1));
    std::vector<long long> fill(ptr.begin(), ptr.end() - 1);
    for (int64_t hc = 0; hc < nh; ++hc)
      {
        const uint8_t *r = rec.data() + (size_t)hc * pfm::PFM_CRES_BYTES;
        const int R = r[pfm::PFM_CRES_R];
        const int32_t *node = reinterpret_cast<const int32_t *>(r), *hv = reinterpret_cast<const int32_t *>(r + pfm::PFM_CRES_HV);
        for (int i = 0; i < R; ++i)
          if (node[i] < NO)
            list[(size_t)fill[(size_t)row_of[(size_t)node[i]]]++] = pfm::HgEntry{(int32_t)(hc * 32 + i), R, off[(size_t)hc] + (long long)i * R * 13};
        for (int a = 0; a < 8; ++a)
          if (hv[a] >= 0 && hv[a] < NO)
            list[(size_t)fill[(size_t)row_of[(size_t)hv[a]]]++] = pfm::HgEntry{(int32_t)(hc * 32 + 16 + a), R, 0};
      }
    c->n_hg_rows = (int64_t)rows.size();
    if (rows.empty())
      rows.push_back(0);
    c->d_hg_rows = dev_upload(c, rows.data(), rows.size());
    c->d_hg_ptr = dev_upload(c, ptr.data(), ptr.size());
    c->d_hg_list = dev_upload(c, list.data(), list.size());
    c->v.hs_off = dev_upload(c, off.data(), off.size());
    c->v.hs_K = dev_alloc<double>(c, (size_t)std::max<long long>(off.back(), 1));
    c->v.hs_RD = dev_alloc<double>(c, (size_t)nh * pfm::PFM_HS_RD);
    c->hang_gather_ready = true;
    return true;
  }

  static int assemble_impl(pfm_ctx *c, int residual_only, double *const *d_values, double *d_res_pde, double *d_res_tot, int phase);

  // assemble_nl_residual() of the line search (cracks.cc:2942-2957, 2507-2512): solution := d_solution, then the residuals.
  // On a single-rank box (2-D or 3-D lattice) the residual kernel reads d_solution itself (DevView::fused_solution); everywhere else this
  // is pfm_state_set_solution + pfm_assemble_device(residual_only).  Ranks with peers must import ghosts in between: they
  // call the two entry points themselves.
  int pfm_assemble_nl_residual_device(pfm_ctx *c, const double *d_solution, double *d_res_pde, double *d_res_tot)
  {
    if (!c || (!d_solution && c->n_owned_dofs() != 0))
      return PFM_ERR_BAD_ARG;
    if (!c->peers.empty())
      return fail(c, PFM_ERR_BAD_ARG, "pfm_assemble_nl_residual_device: a rank with peers imports ghosts between the scatter and the assembly");
    const bool split = c->have_params && c->prm.decompose_stress_matrix > 0 && c->prm.timestep_number > 0;
    const bool fuse = c->kernel_path == 1 && !split && c->v.n_owned == c->v.n_nodes && c->v.n_owned > 0 &&
                      getenv("PFM_NO_FUSED_SCATTER") == nullptr;
    if (!fuse)
      {
        const int rc = state_set_impl(c, d_solution, nullptr, nullptr, 1);
        return rc ? rc : assemble_impl(c, 1, nullptr, d_res_pde, d_res_tot, 0);
      }
    c->v.fused_solution = d_solution;
    const int rc = assemble_impl(c, 1, nullptr, d_res_pde, d_res_tot, 0);
    c->v.fused_solution = nullptr;
    return rc;
  }

  static int ensure_halo_buffers(pfm_ctx *c);

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
    for (double **q : {&c->d_halo_send, &c->d_halo_recv})
      if (*q)
        {
          (void)hipFree(*q);
          *q = nullptr;
        }
    c->device_bytes -= c->halo_buf_bytes;
    c->halo_buf_bytes = 0;
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
        // the message buffers of pfm_halo_exchange are allocated HERE, a local call whose outcome the ranks can agree on,
        // not inside the exchange: an allocation failing on one rank between its peers' enqueued sends and receives would
        // leave them waiting
        const int rcb = ensure_halo_buffers(c);
        if (rcb)
          return rcb;
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

  // ---- ghost import over RCCL inside the library (include/pfm_assemble.h)
  static_assert(sizeof(ncclUniqueId) == PFM_COMM_ID_BYTES, "PFM_COMM_ID_BYTES must match ncclUniqueId");

  // RCCL is bound at run time: the library loads (and single-rank runs work) on hosts without librccl, and in a
  // process that already carries a copy (torch bundles one under the same SONAME) the calls go to THAT copy instead of
  // a second one with its own state.  The run-time version is checked against the header this file was compiled with
  // (same major: ncclUniqueId / ncclComm_t / the send-receive signatures are stable within a major).
  extern "C++"
  {
  namespace
  {
    struct RcclApi
    {
      ncclResult_t (*GetVersion)(int *) = nullptr;
      ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
      ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
      ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
      ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
      ncclResult_t (*GroupStart)() = nullptr;
      ncclResult_t (*GroupEnd)() = nullptr;
      ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
      ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
      const char *(*GetErrorString)(ncclResult_t) = nullptr;
      ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;    // optional (pfm_comm_info)
      ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr; // optional
      int version = 0;
      bool ok = false;
      std::string why;
    };

    const RcclApi &rccl()
    {
      static const RcclApi api = [] {
        RcclApi a;
        void *h = nullptr;
        // already in the process (torch's bundled copy, or the host application's)?
        for (const char *name : {"librccl.so.1", "librccl.so"})
          if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL)))
            break;
        if (!h)
          {
            const char *rocm = getenv("ROCM_PATH");
            const std::string cand[] = {"librccl.so.1", "librccl.so", std::string(rocm ? rocm : "/opt/rocm") + "/lib/librccl.so.1",
                                        std::string(rocm ? rocm : "/opt/rocm") + "/lib/librccl.so"};
            for (const std::string &name : cand)
              if ((h = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL)))
                break;
          }
        if (!h)
          {
            a.why = "librccl.so not found (dlopen)";
            return a;
          }
        bool all = true;
        auto sym = [&](auto &fp, const char *name) {
          fp = reinterpret_cast<std::remove_reference_t<decltype(fp)>>(dlsym(h, name));
          all = all && fp != nullptr;
        };
        sym(a.GetVersion, "ncclGetVersion");
        sym(a.GetUniqueId, "ncclGetUniqueId");
        sym(a.CommInitRank, "ncclCommInitRank");
        sym(a.CommDestroy, "ncclCommDestroy");
        sym(a.CommAbort, "ncclCommAbort");
        sym(a.GroupStart, "ncclGroupStart");
        sym(a.GroupEnd, "ncclGroupEnd");
        sym(a.Send, "ncclSend");
        sym(a.Recv, "ncclRecv");
        sym(a.GetErrorString, "ncclGetErrorString");
        if (!all)
          {
            a.why = "librccl.so lacks a required symbol";
            return a;
          }
        a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(h, "ncclCommCount"));
        a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
        int ver = 0;
        if (a.GetVersion(&ver) != ncclSuccess || ver / 10000 != NCCL_MAJOR)
          {
            a.why = "RCCL run-time version " + std::to_string(ver) + " does not match the compiled-in major " + std::to_string(NCCL_MAJOR);
            return a;
          }
        a.version = ver;
        a.ok = true;
        return a;
      }();
      return api;
    }
  } // namespace
  } // extern "C++"

  // what a communicator handle of this ABI points to: the RCCL communicator and whether a failed exchange aborted it
  // (ncclCommAbort frees the communicator; the owner still holds the handle and must be able to destroy it)
  constexpr uint64_t PFM_COMM_MAGIC = 0x70666d636f6d6d31ull; // "pfmcomm1"
  struct PfmComm
  {
    uint64_t magic = PFM_COMM_MAGIC; // a raw ncclComm_t passed where a handle is expected is refused, not reinterpreted
    ncclComm_t cm = nullptr;
    bool aborted = false;
    bool owned = true; // false: the host's own communicator (pfm_comm_wrap), never destroyed here
    int n_ranks = 0, rank = 0;
  };

  int pfm_comm_unique_id(uint8_t id[PFM_COMM_ID_BYTES])
  {
    if (!id)
      return PFM_ERR_BAD_ARG;
    const RcclApi &R = rccl();
    if (!R.ok)
      return PFM_ERR_COMM;
    ncclUniqueId u;
    if (R.GetUniqueId(&u) != ncclSuccess)
      return PFM_ERR_COMM;
    memcpy(id, &u, sizeof(u));
    return PFM_OK;
  }

  int pfm_comm_create(void **comm, const uint8_t id[PFM_COMM_ID_BYTES], int n_ranks, int rank, int device)
  {
    if (!comm || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks)
      return PFM_ERR_BAD_ARG;
    const RcclApi &R = rccl();
    if (!R.ok)
      return PFM_ERR_COMM;
    if (hipSetDevice(device) != hipSuccess)
      return PFM_ERR_HIP;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t cm = nullptr;
    if (R.CommInitRank(&cm, n_ranks, u, rank) != ncclSuccess)
      return PFM_ERR_COMM;
    PfmComm *h = new (std::nothrow) PfmComm;
    if (!h)
      {
        (void)R.CommAbort(cm);
        return PFM_ERR_NOMEM;
      }
    h->cm = cm;
    h->n_ranks = n_ranks;
    h->rank = rank;
    *comm = h;
    return PFM_OK;
  }

  // The handle outlives an aborted communicator (a failed exchange aborts it so that no peer waits for a message that
  // will never come): destroying it afterwards only frees the handle, using it again reports PFM_ERR_COMM.
  int pfm_comm_destroy(void *comm)
  {
    if (!comm)
      return PFM_OK;
    PfmComm *h = static_cast<PfmComm *>(comm);
    if (h->magic != PFM_COMM_MAGIC)
      return PFM_ERR_BAD_ARG; // not a handle (e.g. the host's raw ncclComm_t): nothing of it is touched
    int rc = PFM_OK;
    if (!h->aborted && h->owned)
      {
        const RcclApi &R = rccl();
        rc = (R.ok && R.CommDestroy(h->cm) == ncclSuccess) ? PFM_OK : PFM_ERR_COMM;
      }
    h->cm = nullptr;
    h->magic = 0;
    delete h;
    return rc;
  }

  int pfm_comm_wrap(void **comm, void *nccl_comm)
  {
    if (!comm || !nccl_comm)
      return PFM_ERR_BAD_ARG;
    PfmComm *h = new (std::nothrow) PfmComm;
    if (!h)
      return PFM_ERR_NOMEM;
    h->cm = static_cast<ncclComm_t>(nccl_comm);
    h->owned = false;
    *comm = h;
    return PFM_OK;
  }

  static bool comm_handle_ok(const void *comm) { return comm && static_cast<const PfmComm *>(comm)->magic == PFM_COMM_MAGIC; }
  int pfm_comm_aborted(const void *comm) { return comm_handle_ok(comm) ? (static_cast<const PfmComm *>(comm)->aborted ? 1 : 0) : 0; }

  // what RCCL itself says about the communicator behind a handle (the driver's N-GPU record can show that RCCL counted N ranks)
  int pfm_comm_info(const void *comm, int *n_ranks, int *rank, int *rccl_version)
  {
    if (!comm_handle_ok(comm))
      return PFM_ERR_BAD_ARG;
    const PfmComm *h = static_cast<const PfmComm *>(comm);
    const RcclApi &R = rccl();
    if (!R.ok || h->aborted || !h->cm)
      return PFM_ERR_COMM;
    int n = -1, r = -1;
    if (R.CommCount && R.CommCount(h->cm, &n) != ncclSuccess)
      return PFM_ERR_COMM;
    if (R.CommUserRank && R.CommUserRank(h->cm, &r) != ncclSuccess)
      return PFM_ERR_COMM;
    if (n_ranks)
      *n_ranks = n;
    if (rank)
      *rank = r;
    if (rccl_version)
      *rccl_version = R.version;
    return PFM_OK;
  }

  // both staging buffers or none: a half-allocated pair must not survive a failed call
  static int ensure_halo_buffers(pfm_ctx *c)
  {
    if (c->d_halo_send && c->d_halo_recv)
      return PFM_OK;
    const int rec = PFM_HALO_DOUBLES_PER_NODE(c->v.dim);
    if (c->d_halo_send)
      (void)hipFree(c->d_halo_send);
    if (c->d_halo_recv)
      (void)hipFree(c->d_halo_recv);
    c->d_halo_send = c->d_halo_recv = nullptr;
    double *snd = nullptr, *rcv = nullptr;
    hipError_t e = hipMalloc((void **)&snd, std::max<size_t>(8, sizeof(double) * rec * c->n_send_all));
    if (e == hipSuccess)
      e = hipMalloc((void **)&rcv, std::max<size_t>(8, sizeof(double) * rec * c->n_recv_all));
    if (e != hipSuccess)
      {
        if (snd)
          (void)hipFree(snd);
        return hipfail(c, e, "hipMalloc halo buffers");
      }
    c->d_halo_send = snd;
    c->d_halo_recv = rcv;
    c->halo_buf_bytes = (int64_t)sizeof(double) * rec * (c->n_send_all + c->n_recv_all);
    c->device_bytes += c->halo_buf_bytes;
    return PFM_OK;
  }

  // pack -> grouped send/receive -> unpack, on `st` (the context's stream, or its side stream for the overlapped form)
  static int halo_exchange_on(pfm_ctx *c, void *comm, const int *peer_ranks, hipStream_t st)
  {
    const RcclApi &R = rccl();
    if (!R.ok)
      return fail(c, PFM_ERR_COMM, "RCCL unavailable: " + R.why);
    PfmComm *h = static_cast<PfmComm *>(comm);
    if (h->magic != PFM_COMM_MAGIC)
      return fail(c, PFM_ERR_BAD_ARG, "comm is not a handle of pfm_comm_create / pfm_comm_wrap (a raw ncclComm_t must be wrapped)");
    if (h->aborted || !h->cm)
      return fail(c, PFM_ERR_COMM, "communicator was aborted by an earlier failed exchange");
    ncclComm_t cm = h->cm;
    // A rank that fails locally before its sends and receives are enqueued must not leave its peers waiting for them:
    // a communicator the library OWNS is aborted on every failure from here on (the buffers themselves come from
    // pfm_halo_register).  A wrapped communicator belongs to the host: it is never aborted or destroyed here -- the handle is
    // marked (pfm_comm_aborted() == 1, further exchanges refused) and the host decides what to do with its ncclComm_t.
    auto abort_comm = [&](int code, const std::string &msg) {
      if (h->owned)
        (void)R.CommAbort(cm);
      h->aborted = true; // the handle stays valid for pfm_comm_destroy; further exchanges on it are refused
      h->cm = nullptr;
      return fail(c, code, msg + (h->owned ? " (communicator aborted)" : " (wrapped communicator left to the host, handle disabled)"));
    };
    int rc = ensure_halo_buffers(c);
    if (rc)
      return abort_comm(rc, "halo buffers");
    const int rec = PFM_HALO_DOUBLES_PER_NODE(c->v.dim);
    rc = launch_halo_all(c->v, c->d_send_all, c->d_send_ptr, (int)c->peers.size(), c->n_send_all, c->d_halo_send, 0, st);
    if (rc)
      return abort_comm(rc, "halo pack launch failed");
    ncclResult_t r = R.GroupStart();
    int64_t so = 0, ro = 0;
    for (size_t k = 0; k < c->peers.size() && r == ncclSuccess; ++k)
      {
        const HaloPeer &p = c->peers[k];
        if (p.n_recv)
          r = R.Recv(c->d_halo_recv + rec * ro, (size_t)(rec * p.n_recv), ncclDouble, peer_ranks[k], cm, st);
        if (p.n_send && r == ncclSuccess)
          r = R.Send(c->d_halo_send + rec * so, (size_t)(rec * p.n_send), ncclDouble, peer_ranks[k], cm, st);
        so += p.n_send;
        ro += p.n_recv;
      }
    const ncclResult_t r2 = R.GroupEnd();
    if (r == ncclSuccess)
      r = r2;
    if (r != ncclSuccess)
      {
        // work may already be enqueued on the peers: there is no safe fall-back from here -- abort the
        // communicator so that no rank waits for a message that will never come, and report
        return abort_comm(PFM_ERR_COMM, std::string("RCCL: ") + R.GetErrorString(r));
      }
    rc = launch_halo_all(c->v, c->d_recv_all, c->d_recv_ptr, (int)c->peers.size(), c->n_recv_all, c->d_halo_recv, 1, st);
    if (rc)
      return fail(c, rc, "halo unpack launch failed");
    return PFM_OK;
  }

  int pfm_halo_exchange(pfm_ctx *c, void *comm, const int *peer_ranks)
  {
    if (!c || (!c->peers.empty() && (!comm || !peer_ranks)))
      return PFM_ERR_BAD_ARG;
    if (c->peers.empty())
      return PFM_OK;
    (void)hipSetDevice(c->device);
    return halo_exchange_on(c, comm, peer_ranks, c->stream);
  }

  // cartesian overlay: the slots of the regular rows follow the order of the node-graph rows (pfm_pattern_bind may change it)
  static int ensure_patch_ready(pfm_ctx *c)
  {
    if (c->n_patch_blocks == 0 || c->patch_slots_valid)
      return PFM_OK;
    try
      {
        ensure_general_tables(c); // v.nadj (lattice contexts build it on first use)
      }
    catch (const HipFail &f)
      {
        return hipfail(c, f.e, f.what);
      }
    catch (const std::bad_alloc &)
      {
        return fail(c, PFM_ERR_NOMEM, "host allocation failed");
      }
    const int rc = launch_patch_slots(c->v, c->d_node_slots, c->n_patch_blocks, c->stream);
    if (rc)
      return fail(c, rc, "patch slots");
    c->patch_slots_valid = true;
    return PFM_OK;
  }

  // 3-D overlay: neighbour masks and CSR slots of the regular rows from the current order of the node-graph rows
  static int ensure_overlay3_ready(pfm_ctx *c)
  {
    if (c->levels3.empty() || c->overlay3_rows_valid)
      return PFM_OK;
    try
      {
        ensure_general_tables(c); // v.nadj
        if (!c->d_nbr_mask3)
          {
            c->d_nbr_mask3 = dev_alloc<uint32_t>(c, (size_t)std::max<int32_t>(c->v.n_owned, 1));
            c->d_row_perm3 = dev_alloc<uint8_t>(c, (size_t)std::max<long long>(c->nadj_total >= 0 ? c->nadj_total : (c->h_nadj_ptr.empty() ? 1 : c->h_nadj_ptr.back()), 1));
          }
      }
    catch (const HipFail &f)
      {
        return hipfail(c, f.e, f.what);
      }
    catch (const std::bad_alloc &)
      {
        return fail(c, PFM_ERR_NOMEM, "host allocation failed");
      }
    int *d_bad = nullptr;
    if (hipMalloc((void **)&d_bad, sizeof(int)) != hipSuccess)
      return fail(c, PFM_ERR_NOMEM, "overlay row tables");
    hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(int), c->stream);
    if (e == hipSuccess)
      e = hipMemsetAsync(c->d_nbr_mask3, 0, sizeof(uint32_t) * (size_t)std::max<int32_t>(c->v.n_owned, 1), c->stream);
    int rc = e == hipSuccess ? PFM_OK : PFM_ERR_HIP;
    for (auto &lv : c->levels3)
      if (rc == PFM_OK)
        rc = launch_overlay3_rows(lv.cv.local_of_box, lv.cv.row_of_box, lv.cv.NX, lv.cv.NY, lv.cv.NZ, c->v.nadj_ptr, c->v.nadj, c->d_nbr_mask3,
                                  c->d_row_perm3, d_bad, c->stream);
    int bad = 0;
    if (rc == PFM_OK && (hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                         hipStreamSynchronize(c->stream) != hipSuccess))
      rc = PFM_ERR_HIP;
    (void)hipFree(d_bad);
    if (rc)
      return fail(c, rc, "overlay row tables");
    if (bad != 0)
      return fail(c, PFM_ERR_INTERNAL, "3-D overlay: a regular row is not a plain 27-neighbour row of its level");
    for (auto &lv : c->levels3)
      {
        lv.cv.nbr_mask = c->d_nbr_mask3;
        lv.cv.row_perm = c->d_row_perm3;
      }
    c->overlay3_rows_valid = true;
    return PFM_OK;
  }

  // phase 2 of pfm_assemble_overlapped launches only the tiles that read ghost nodes: their indices, once per context
  static int ensure_overlap_lists(pfm_ctx *c)
  {
    if (c->overlap_lists_ready || c->kernel_path != 1 || c->v.dim != 3)
      return PFM_OK;
    std::vector<int32_t> t_uu, t_res;
    int zc = 0;
    cart_uu3_boundary_tiles(c->cv, t_uu);
    cart_res3_boundary_tiles(c->cv, t_res, zc);
    const int n_uu = (int)t_uu.size(), n_res = (int)t_res.size();
    if (t_uu.empty())
      t_uu.push_back(0);
    if (t_res.empty())
      t_res.push_back(0);
    try
      {
        c->cv.bnd_uu3 = dev_upload(c, t_uu.data(), t_uu.size());
        c->cv.bnd_res3 = dev_upload(c, t_res.data(), t_res.size());
      }
    catch (const HipFail &f)
      {
        return hipfail(c, f.e, f.what);
      }
    c->cv.n_bnd_uu3 = n_uu;
    c->cv.n_bnd_res3 = n_res;
    c->cv.zc_res3 = zc;
    c->overlap_lists_ready = true;
    return PFM_OK;
  }

  // phase 0: the whole assembly (pfm_assemble_device).  phases 1, 2: the two halves of pfm_assemble_overlapped -- 1 = what
  // reads no ghost node, 2 = the rest; timing events bracket 1..2 together.
  static int assemble_impl(pfm_ctx *c, int residual_only, double *const *d_values, double *d_res_pde, double *d_res_tot, int phase)
  {
    // a rank may own nothing (empty partition piece): null buffers are fine where there is nothing to write
    const bool no_rows = c && c->n_owned_dofs() == 0;
    if (!c || (!d_res_pde && !no_rows) || (residual_only && !d_res_tot && !no_rows) || (!residual_only && !d_values))
      return PFM_ERR_BAD_ARG;
    if (!c->have_params)
      return fail(c, PFM_ERR_BAD_ARG, "pfm_set_params has not been called");
    (void)hipSetDevice(c->device);
    hipError_t e = hipSuccess;
    const bool split = c->prm.decompose_stress_matrix > 0 && c->prm.timestep_number > 0; // cracks.cc:2294
    const bool cart = c->kernel_path == 1 && !split && (residual_only || cart_matrix_supported(c->v.dim));
    if (!cart && !c->general_ready)
      {
        try
          {
            ensure_general_tables(c);
          }
        catch (const HipFail &f)
          {
            return hipfail(c, f.e, f.what);
          }
        catch (const std::bad_alloc &)
          {
            return fail(c, PFM_ERR_NOMEM, "host allocation failed");
          }
      }
    const bool overlay_uu = c->kernel_path == 2 && !residual_only && !split; // debug: general + cart (u,u)
    // cartesian overlay of a general 2-D mesh (AMR meshes; stress-split runs on a lattice): patch kernel for the regular rows,
    // the general family for the rest (PFM_NO_PATCH=1 at context creation: general family alone)
    const bool overlay3 = !cart && c->v.dim == 3 && !c->levels3.empty() && c->kernel_path != 0 && phase != 1 && !split;
    const bool patches = (!cart && c->v.dim == 2 && c->n_patch_blocks > 0 && c->kernel_path != 0 && phase != 1) || overlay3;
    if (patches)
      {
        const int rcp = overlay3 ? ensure_overlay3_ready(c) : ensure_patch_ready(c);
        if (rcp)
          return rcp;
      }
    else if (!cart && c->full_colours_lazy)
      {
        try
          {
            ensure_full_colours(c); // the general family over all cells of a context that has only run with its overlay so far
          }
        catch (const HipFail &f)
          {
            return hipfail(c, f.e, f.what);
          }
        catch (const std::bad_alloc &)
          {
            return fail(c, PFM_ERR_NOMEM, "host allocation failed");
          }
      }
    // 3-D cells at hanging vertices: scratch + ordered gather instead of atomics (round 6; DevView::hs_*)
    bool gather = false;
    if (!cart && c->v.dim == 3 && c->hang_gather && !split && phase != 1)
      {
        try
          {
            gather = ensure_hang_gather(c);
          }
        catch (const HipFail &f)
          {
            return hipfail(c, f.e, f.what);
          }
        catch (const std::bad_alloc &)
          {
            return fail(c, PFM_ERR_NOMEM, "host allocation failed");
          }
      }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (c->timing && phase == 2)
      ev1 = c->ev_pool[c->ev_used - 1].second; // opened by phase 1
    if (c->timing && phase != 2)
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
    // The two Jacobian kernels of a 3-D box next to each other (default): with equal LDS allocations (64 granules of 1280 B
    // each) any freed slot of a CU takes a workgroup of either kernel, the (u,u) and the phase-field workgroups mix and
    // fill each other's stalls: 14.07 -> 13.75 ms per assembly at 216^3 (PFM_JAC_SEQUENTIAL=1: one after the other).
    static const bool jac_sequential = getenv("PFM_JAC_SEQUENTIAL") != nullptr;
    const bool pair = cart && !jac_sequential && cart_jacobian_pair(c->v, c->cv, c->prm, residual_only, phase);
    if (pair)
      {
        // deferred placeholder patches: one entry per flagged displacement dof at most
        if (c->cv.patch_cap < c->n_flag_u || !c->cv.patch_count)
          {
            for (void *q : {(void *)c->cv.patch_idx, (void *)c->cv.patch_val, (void *)c->cv.patch_count})
              if (q)
                (void)hipFree(q);
            c->device_bytes -= (int64_t)c->cv.patch_cap * 16 + (c->cv.patch_count ? 4 : 0);
            c->cv.patch_idx = nullptr, c->cv.patch_val = nullptr, c->cv.patch_count = nullptr;
            const size_t cap = (size_t)std::max<int64_t>(c->n_flag_u, 1);
            if (hipMalloc((void **)&c->cv.patch_idx, cap * sizeof(long long)) != hipSuccess ||
                hipMalloc((void **)&c->cv.patch_val, cap * sizeof(double)) != hipSuccess ||
                hipMalloc((void **)&c->cv.patch_count, sizeof(int)) != hipSuccess)
              return fail(c, PFM_ERR_NOMEM, "patch list");
            c->cv.patch_cap = (int)std::min<size_t>(cap, 0x7fffffff);
            c->device_bytes += (int64_t)c->cv.patch_cap * 16 + 4;
          }
        if (hipMemsetAsync(c->cv.patch_count, 0, sizeof(int), c->stream) != hipSuccess)
          return fail(c, PFM_ERR_HIP, "patch list reset");
      }
    // 2-D boxes: the two launches of k_cart2d_cells (displacement rows | phase-field rows) next to each other
    // (measured: 0.262 -> 0.259 ms of kernel time at 1000^2, nothing for the caller after the fork and the join: off by default)
    static const bool cart2d_forked = getenv("PFM_CART2D_FORKED") != nullptr && getenv("PFM_CART2D_ONE_LAUNCH") == nullptr;
    const bool fork = cart && !residual_only && phase == 0 && (pair || (c->v.dim == 2 && cart2d_forked) || getenv("PFM_SIDE_STREAM"));
    if (fork)
      {
        if (!c->side_stream)
          {
            // PFM_SIDE_PRIO (A/B runs): -1 = lowest, 1 = highest dispatch priority for the stream of the phase-field kernel
            static const int side_prio = getenv("PFM_SIDE_PRIO") ? atoi(getenv("PFM_SIDE_PRIO")) : 0;
            int prio_lo = 0, prio_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            const hipError_t es = side_prio == 0 ? hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking)
                                                 : hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, side_prio > 0 ? prio_hi : prio_lo);
            if (es != hipSuccess ||
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
    if (!cart && phase == 1)
      return PFM_OK; // the general family is not cut into interior / boundary work: everything in phase 2
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
          if (!cart && !patches)
            e = hipMemsetAsync(d_values[b], 0, sizeof(double) * (size_t)c->block_nnz(b), c->stream);
        }
    if (e == hipSuccess && patches && !residual_only && c->n_rows_general > 0 &&
        launch_zero_rows(c->v, d_values, c->d_rows_general, c->n_rows_general, c->stream) != PFM_OK)
      return fail(c, PFM_ERR_HIP, "zero the rows of the general family");
    if (e != hipSuccess)
      return hipfail(c, e, "zero outputs");
    // general family: the atomic class (cells with hanging vertices) on the side stream next to the colour classes, behind
    // the zeroing of the outputs (DevView::cell_ring; PFM_GENERAL_SEQUENTIAL=1: one after the other)
    static const bool general_sequential = getenv("PFM_GENERAL_SEQUENTIAL") != nullptr;
    // (a residual-only assembly keeps the stream order: its atomic class takes 15 us, less than a fork and a join)
    // overlay: the patch kernel on the stream, every class of the (small) rest of the general family on the side stream
    const bool fork_general = !cart && !general_sequential &&
                              ((!residual_only && c->v.cell_ring != nullptr) || (patches && c->n_general_cells > 0));
    if (fork_general)
      {
        if (!c->side_stream)
          {
            if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)
              return fail(c, PFM_ERR_HIP, "side stream");
          }
        e = hipEventRecord(c->ev_fork, c->stream);
        if (e == hipSuccess)
          e = hipStreamWaitEvent(c->side_stream, c->ev_fork, 0);
        if (e != hipSuccess)
          return hipfail(c, e, "fork (general family)");
      }
    if (cart && !residual_only && c->v.dim == 3 && c->scal_dirty)
      {
        // off the hot path: once per pfm_set_params, complete before any kernel of any stream may read it
        int rcs = upload_mat_scal(c->prm, c->cv, c->d_scal, c->stream);
        if (rcs == PFM_OK && hipStreamSynchronize(c->stream) != hipSuccess)
          rcs = PFM_ERR_HIP;
        if (rcs)
          return fail(c, rcs, "scalar tables");
        c->scal_dirty = false;
      }
    pfm::CartView cv_launch = c->cv;
    cv_launch.up_block_cleared = 0;
    // 2-D box, blocked layout: the structurally zero (u,phi) block (cracks.cc:2333-2337) by a fill in front of the kernels
    // (PFM_CART2D_NO_FILL: the kernel writes the zeros with the rows, A/B runs)
    if (cart && c->v.dim == 2 && !residual_only && phase <= 1 && c->v.layout == PFM_LAYOUT_BLOCKED && c->n_blocks == 4 && d_values[1] &&
        getenv("PFM_CART2D_NO_FILL") == nullptr && getenv("PFM_CART2D_ONE_LAUNCH") == nullptr)
      {
        if (hipMemsetAsync(d_values[1], 0, sizeof(double) * (size_t)c->block_nnz(1), c->stream) != hipSuccess)
          return fail(c, PFM_ERR_HIP, "clear the (u,phi) block");
        cv_launch.up_block_cleared = 1;
      }
    if (cart && c->v.dim == 2 && phase == 2 && c->v.layout == PFM_LAYOUT_BLOCKED && getenv("PFM_CART2D_NO_FILL") == nullptr &&
        getenv("PFM_CART2D_ONE_LAUNCH") == nullptr && !residual_only)
      cv_launch.up_block_cleared = 1; // cleared in phase 1 of this overlapped assembly
    // (round 6, measured and removed: clearing the structurally zero (u,phi) block with a fill on a third stream instead of
    // by 21 of the 49 store instructions of every copy-out of k_cart_phi4 -- 10.95 -> 11.35 ms per assembly at 216^3,
    // profiles/r06/ab_up_fill.txt: the fill takes bandwidth in a burst and dispatch slots from the pair)
    if (!pair)
      cv_launch.patch_idx = nullptr, cv_launch.patch_val = nullptr, cv_launch.patch_count = nullptr, cv_launch.patch_cap = 0;
    int rc;
    if (cart)
      rc = launch_assemble_cart(c->v, cv_launch, c->prm, residual_only, d_values, d_res_pde, d_res_tot, c->stream, pair || fork ? s_res : c->stream, c->d_scal, phase);
    else if (patches)
      {
        // the patch kernel: every regular row is written once, with plain stores; next to it (side stream) the general
        // family over the cells that touch a non-regular row (it skips the regular ones: DevView::row_patch), its classes one
        // after the other
        rc = PFM_OK;
        if (overlay3)
          {
            // the row-owner kernels of the cartesian family on every level lattice: each level on a stream of its own, forked
            // off the context's stream (a level lattice of 6e5 nodes fills a third of the chip's dispatch slots; the levels
            // write disjoint rows) when PFM_OVERLAY3_CONCURRENT=1 is set.  Default: one after the other on the context's
            // stream -- measured the better of the two (profiles/r05/ov3_ab.txt: 4.56 against 4.86 ms per Jacobian)
            static const bool levels_sequential = getenv("PFM_OVERLAY3_CONCURRENT") == nullptr;
            const size_t nl = c->levels3.size();
            for (auto &lv : c->levels3)
              if (rc == PFM_OK && !residual_only && lv.scal_dirty)
                {
                  rc = upload_mat_scal(c->prm, lv.cv, lv.d_scal, c->stream);
                  lv.scal_dirty = rc != PFM_OK;
                }
            const bool par = nl > 1 && !levels_sequential && rc == PFM_OK;
            if (par)
              {
                while (c->ov_streams.size() < nl)
                  {
                    hipStream_t st = nullptr;
                    hipEvent_t ev = nullptr;
                    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
                      return fail(c, PFM_ERR_HIP, "overlay level stream");
                    c->ov_streams.push_back(st);
                    c->ov_events.push_back(ev);
                  }
                if (!c->ov_fork && hipEventCreateWithFlags(&c->ov_fork, hipEventDisableTiming) != hipSuccess)
                  return fail(c, PFM_ERR_HIP, "overlay level event");
                e = hipEventRecord(c->ov_fork, c->stream);
                for (size_t i = 0; i < nl && e == hipSuccess; ++i)
                  e = hipStreamWaitEvent(c->ov_streams[i], c->ov_fork, 0);
                if (e != hipSuccess)
                  return hipfail(c, e, "fork (overlay levels)");
              }
            for (size_t i = 0; i < nl && rc == PFM_OK; ++i)
              {
                hipStream_t st = par ? c->ov_streams[i] : c->stream;
                rc = launch_assemble_cart(c->v, c->levels3[i].cv, c->prm, residual_only, d_values, d_res_pde, d_res_tot, st, st, c->levels3[i].d_scal, 0);
              }
            if (par)
              {
                for (size_t i = 0; i < nl && e == hipSuccess; ++i)
                  {
                    e = hipEventRecord(c->ov_events[i], c->ov_streams[i]);
                    if (e == hipSuccess)
                      e = hipStreamWaitEvent(c->stream, c->ov_events[i], 0);
                  }
                if (e != hipSuccess)
                  return hipfail(c, e, "join (overlay levels)");
              }
          }
        else
          rc = launch_assemble_patches(c->v, c->prm, residual_only, d_values, d_res_pde, d_res_tot, c->n_patch_blocks, c->stream);
        pfm::DevView vg = c->v;
        vg.color_cells = c->d_color_cells_reduced;
        if (gather)
          vg.cell_ring = nullptr; // nobody adds atomically: the cells at hanging vertices only write their scratch
        else
          vg.hs_K = nullptr, vg.hs_off = nullptr, vg.hs_RD = nullptr;
        // 3-D overlay Jacobian: the class of the cells at hanging vertices (FP64 atomics, 4.7 of the general family's 6 ms at
        // 1.1e6 cells) on a third stream next to the plain classes (DevView::cell_ring makes that safe)
        hipStream_t s_atomic = nullptr;
        // (PFM_HANGING_COLOURED: a hanging cell in a plain class adds to its PARENTS' rows without atomics; should a cell with
        // more than 16 resolved nodes ever sit in the atomic class next to it, that class must not run beside the plain ones)
        if (overlay3 && fork_general && !residual_only && (c->v.cell_ring || gather) && c->n_general_cells > 0 && !c->hanging_coloured)
          {
            int prio_lo = 0, prio_hi = 0; // (numerically lower = higher priority) the long pole gets its workgroups dispatched first
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            if (!c->atomic_stream && (hipStreamCreateWithPriority(&c->atomic_stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
                                      hipEventCreateWithFlags(&c->ev_atomic, hipEventDisableTiming) != hipSuccess))
              return fail(c, PFM_ERR_HIP, "atomic-class stream");
            e = hipStreamWaitEvent(c->atomic_stream, c->ev_fork, 0);
            if (e != hipSuccess)
              return hipfail(c, e, "fork (atomic class)");
            s_atomic = c->atomic_stream;
          }
        if (rc == PFM_OK && c->n_general_cells > 0)
          rc = launch_assemble_general(vg, c->prm, residual_only, d_values, d_res_pde, d_res_tot, fork_general ? c->side_stream : c->stream,
                                       c->color_ptr_reduced, s_atomic);
        if (s_atomic)
          {
            e = hipEventRecord(c->ev_atomic, s_atomic);
            if (e == hipSuccess)
              e = hipStreamWaitEvent(c->stream, c->ev_atomic, 0);
            if (e != hipSuccess)
              return hipfail(c, e, "join (atomic class)");
          }
      }
    else
      {
        pfm::DevView vg = c->v;
        vg.row_patch = nullptr; // no overlay in this assembly: the general family writes every row
        if (gather)
          vg.cell_ring = nullptr;
        else
          vg.hs_K = nullptr, vg.hs_off = nullptr, vg.hs_RD = nullptr;
        rc = launch_assemble_general(vg, c->prm, residual_only, d_values, d_res_pde, d_res_tot, c->stream, c->color_ptr,
                                     (fork_general && !c->hanging_coloured) ? c->side_stream : nullptr);
      }
    if (fork_general)
      {
        e = hipEventRecord(c->ev_join, c->side_stream);
        if (e == hipSuccess)
          e = hipStreamWaitEvent(c->stream, c->ev_join, 0);
        if (e != hipSuccess)
          return hipfail(c, e, "join (general family)");
      }
    if (gather && rc == PFM_OK && !cart)
      {
        // behind every class and every level kernel (joined above): the rows the cells at hanging vertices reach, in list order
        pfm::DevView vg = c->v;
        if (!patches)
          vg.row_patch = nullptr;
        rc = launch_hanging_gather(vg, c->prm, residual_only, d_values, d_res_pde, d_res_tot, c->d_hg_rows, c->d_hg_ptr, c->d_hg_list,
                                   c->n_hg_rows, c->stream);
      }
    if (fork)
      {
        e = hipEventRecord(c->ev_join, s_res);
        if (e == hipSuccess)
          e = hipStreamWaitEvent(c->stream, c->ev_join, 0);
        if (e != hipSuccess)
          return hipfail(c, e, "join");
      }
    if (pair && rc == PFM_OK)
      rc = launch_cart_apply_patches(c->cv, d_values[0], c->stream, c->v.status);
    if (phase == 1)
      return rc ? fail(c, rc, "assemble launch failed (interior tiles)") : PFM_OK;
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

  int pfm_assemble_device(pfm_ctx *c, int residual_only, double *const *d_values, double *d_res_pde, double *d_res_tot)
  {
    if (c && c->force_phase) // measurement: one half of the overlapped assembly alone, bracketed like a whole one
      {
        const bool timing = c->timing;
        c->timing = false;
        hipEvent_t a = nullptr, b = nullptr;
        if (timing)
          {
            if (c->ev_used == c->ev_pool.size())
              {
                if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)
                  return fail(c, PFM_ERR_HIP, "hipEventCreate");
                c->ev_pool.emplace_back(a, b);
              }
            a = c->ev_pool[c->ev_used].first;
            b = c->ev_pool[c->ev_used].second;
            ++c->ev_used;
            (void)hipEventRecord(a, c->stream);
          }
        const int rc = assemble_impl(c, residual_only, d_values, d_res_pde, d_res_tot, c->force_phase);
        if (timing)
          (void)hipEventRecord(b, c->stream);
        c->timing = timing;
        return rc;
      }
    return assemble_impl(c, residual_only, d_values, d_res_pde, d_res_tot, 0);
  }

  int pfm_ctx_force_phase(pfm_ctx *c, int phase)
  {
    if (!c || phase < 0 || phase > 2)
      return PFM_ERR_BAD_ARG;
    c->force_phase = phase;
    return phase ? ensure_overlap_lists(c) : PFM_OK;
  }

  // Ghost import NEXT TO the cell work instead of in front of it (cracks.cc:2147-2154 against 2200-2437): after the
  // caller's pfm_state_set, the exchange (pack -> RCCL -> unpack) runs on the context's side stream while the context's
  // stream assembles the tiles that read no ghost node; the stream then waits for the import and assembles the rest.
  int pfm_assemble_overlapped(pfm_ctx *c, void *comm, const int *peer_ranks, int residual_only, double *const *d_values,
                              double *d_res_pde, double *d_res_tot)
  {
    if (!c || (!c->peers.empty() && (!comm || !peer_ranks)))
      return PFM_ERR_BAD_ARG;
    if (c->peers.empty())
      return assemble_impl(c, residual_only, d_values, d_res_pde, d_res_tot, 0);
    (void)hipSetDevice(c->device);
    if (!c->side_stream)
      {
        if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)
          return fail(c, PFM_ERR_HIP, "side stream");
      }
    {
      const int rcl = ensure_overlap_lists(c);
      if (rcl)
        return rcl;
    }
    // fork behind the state scatter; the pack kernel of the exchange reads the owned node state
    hipError_t e = hipEventRecord(c->ev_fork, c->stream);
    if (e == hipSuccess)
      e = hipStreamWaitEvent(c->side_stream, c->ev_fork, 0);
    if (e != hipSuccess)
      return hipfail(c, e, "fork");
    int rc = halo_exchange_on(c, comm, peer_ranks, c->side_stream);
    if (rc)
      return rc;
    e = hipEventRecord(c->ev_join, c->side_stream);
    if (e != hipSuccess)
      return hipfail(c, e, "join event");
    rc = assemble_impl(c, residual_only, d_values, d_res_pde, d_res_tot, 1); // interior: no ghost node is read
    if (rc)
      return rc;
    e = hipStreamWaitEvent(c->stream, c->ev_join, 0);
    if (e != hipSuccess)
      return hipfail(c, e, "join");
    return assemble_impl(c, residual_only, d_values, d_res_pde, d_res_tot, 2);
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
        return fail(c, st, st == PFM_ERR_NOT_ORTHOGONAL ? "eigenvectors not orthogonal (cracks.cc:1732-1736)"
                           : st == PFM_ERR_NONFINITE    ? "non-finite value (pfm_check_finite)"
                                                        : "device-side error");
      }
    return PFM_OK;
  }

  int pfm_check_finite(pfm_ctx *c, const double *d_data, int64_t n)
  {
    if (!c || n < 0 || (n > 0 && !d_data))
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    const int rc = launch_check_finite(c->v, d_data, n, c->stream);
    if (rc)
      return fail(c, rc, "check_finite launch failed");
    return pfm_sync_status(c);
  }

  // ---- host-visible outputs (the caller hands system_pde_matrix to Trilinos right after the call,
  // cracks.cc:2754, 2770, 2918).  35 GB of matrix values cross PCIe per Jacobian at 216^3: what can be done about that is
  // (1) DMA from / to page-locked memory instead of the runtime's pageable path (pfm_host_register), (2) never moving the
  // (u,phi) block, which is identically zero (cracks.cc:2333-2337; placeholders of constrained rows live on the diagonals
  // of the (u,u) and (phi,phi) blocks): 3/16 of the bytes, (3) two copy streams.
  int pfm_host_register(pfm_ctx *c, void *p, int64_t bytes)
  {
    if (!c || !p || bytes <= 0)
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    for (auto &hp : c->host_pins)
      if (hp.p == p)
        {
          if (hp.bytes >= (size_t)bytes)
            return PFM_OK;
          (void)hipHostUnregister(hp.p); // the same array, grown: lock it again
          hp.p = nullptr;
        }
    c->host_pins.erase(std::remove_if(c->host_pins.begin(), c->host_pins.end(), [](const pfm_ctx::HostPin &h) { return h.p == nullptr; }),
                       c->host_pins.end());
    const hipError_t e = hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault);
    if (e != hipSuccess)
      {
        (void)hipGetLastError(); // not sticky: the transfers simply take the pageable path
        return fail(c, PFM_ERR_HIP, std::string("hipHostRegister: ") + hipGetErrorString(e) + " (the array stays pageable)");
      }
    pfm_ctx::HostPin hp;
    hp.p = p;
    hp.bytes = (size_t)bytes;
    c->host_pins.push_back(hp);
    return PFM_OK;
  }

  int pfm_host_unregister(pfm_ctx *c, void *p)
  {
    if (!c)
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream)
      (void)hipStreamSynchronize(c->copy_stream);
    for (auto &hp : c->host_pins)
      if (!p || hp.p == p)
        {
          (void)hipHostUnregister(hp.p);
          hp.p = nullptr;
        }
    c->host_pins.erase(std::remove_if(c->host_pins.begin(), c->host_pins.end(), [](const pfm_ctx::HostPin &h) { return h.p == nullptr; }),
                       c->host_pins.end());
    return PFM_OK;
  }

  static pfm_ctx::HostPin *find_pin(pfm_ctx *c, const void *p, size_t bytes)
  {
    for (auto &hp : c->host_pins)
      if (hp.p == p && hp.bytes >= bytes)
        return &hp;
    return nullptr;
  }

  // matrix values of the last pfm_assemble_device -> the host's arrays, asynchronously: block 0 on the context's stream,
  // the phase-field blocks on a second stream (joined into the first before returning).  Does not synchronise.
  static int values_to_host_async(pfm_ctx *c, double *const *d_values, double *const *h_values)
  {
    (void)hipSetDevice(c->device);
    if (!c->copy_stream)
      {
        if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming) != hipSuccess)
          return fail(c, PFM_ERR_HIP, "copy stream");
      }
    for (int b = 0; b < c->n_blocks; ++b) // (checked before the fork: no exit path leaves copy_stream unjoined)
      if (c->block_nnz(b) > 0 && (!h_values[b] || !d_values[b]))
        return fail(c, PFM_ERR_BAD_ARG, "pfm_values_to_host: null block");
    hipError_t e = hipEventRecord(c->ev_copy, c->stream);
    if (e == hipSuccess)
      e = hipStreamWaitEvent(c->copy_stream, c->ev_copy, 0);
    for (int b = 0; b < c->n_blocks && e == hipSuccess; ++b)
      {
        const size_t bytes = sizeof(double) * (size_t)c->block_nnz(b);
        if (!bytes)
          continue;
        if (c->n_blocks == 4 && b == 1)
          {
            // (u,phi) = 0.  A registered array is ours between the calls (nobody else writes the matrix: the reference only
            // ever fills it through assemble_system): cleared once, by host threads next to the transfers, never copied.
            // An unregistered one is copied like the others (the device block holds the zeros the kernels wrote).
            pfm_ctx::HostPin *hp = find_pin(c, h_values[b], bytes);
            if (hp)
              {
                if (!hp->zeroed)
                  {
                    const int nt = std::max(1, std::min(16, (int)std::thread::hardware_concurrency()));
                    std::vector<std::thread> th;
                    const size_t chunk = ((bytes / (size_t)nt) + 4095) & ~(size_t)4095;
                    for (int t = 0; t < nt; ++t)
                      {
                        const size_t lo = std::min(bytes, (size_t)t * chunk), hi = std::min(bytes, lo + chunk);
                        if (hi > lo)
                          th.emplace_back([=] { std::memset(reinterpret_cast<char *>(h_values[b]) + lo, 0, hi - lo); });
                      }
                    for (auto &t : th)
                      t.join();
                    hp->zeroed = true;
                  }
                continue;
              }
          }
        e = d2h_user(c, h_values[b], d_values[b], bytes, b == 0 ? c->stream : c->copy_stream);
      }
    if (e == hipSuccess)
      e = hipEventRecord(c->ev_copy, c->copy_stream);
    if (e == hipSuccess)
      e = hipStreamWaitEvent(c->stream, c->ev_copy, 0);
    return e == hipSuccess ? PFM_OK : hipfail(c, e, "copy back (matrix values)");
  }

  int pfm_values_to_host(pfm_ctx *c, double *const *d_values, double *const *h_values)
  {
    if (!c || !d_values || !h_values)
      return PFM_ERR_BAD_ARG;
    const int rc = values_to_host_async(c, d_values, h_values);
    if (rc)
      return rc;
    const hipError_t e = hipStreamSynchronize(c->stream);
    return e == hipSuccess ? PFM_OK : hipfail(c, e, "pfm_values_to_host");
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
    // the residual first (the caller's Newton loop reads it at once, cracks.cc:2791-2794), then the matrix blocks
    hipError_t e = d2h_user(c, residual_pde, c->d_stage_res[0], vb, c->stream);
    if (e == hipSuccess && residual_only)
      e = d2h_user(c, residual_total, c->d_stage_res[1], vb, c->stream);
    if (e != hipSuccess)
      return hipfail(c, e, "copy back");
    if (!residual_only)
      {
        rc = values_to_host_async(c, c->d_stage_val, values);
        if (rc)
          return rc;
      }
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

  int pfm_kernel_times_ms(pfm_ctx *c, double *ms, int capacity, int *n_launches)
  {
    if (!c || (capacity > 0 && !ms) || capacity < 0)
      return PFM_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess)
      return hipfail(c, e, "kernel_times sync");
    const int n = (int)std::min<size_t>(c->ev_used, (size_t)capacity);
    for (int k = 0; k < n; ++k)
      {
        float t = 0.f;
        e = hipEventElapsedTime(&t, c->ev_pool[k].first, c->ev_pool[k].second);
        if (e != hipSuccess)
          return hipfail(c, e, "hipEventElapsedTime");
        ms[k] = t;
      }
    if (n_launches)
      *n_launches = (int)c->ev_used;
    return PFM_OK;
  }

  int pfm_ctx_kernel_path(const pfm_ctx *c) { return c ? c->kernel_path : -1; }

  int pfm_ctx_force_path(pfm_ctx *c, int path)
  {
    if (path == 3 && c && !c->cart_ok && (c->n_patch_blocks > 0 || !c->levels3.empty()))
      {
        c->kernel_path = 3;
        return PFM_OK;
      }
    if (!c || path < 0 || path > 2 || (path >= 1 && !c->cart_ok) || (path == 2 && c->v.dim != 3))
      return PFM_ERR_UNSUPPORTED;
    c->kernel_path = path;
    return PFM_OK;
  }

  int pfm_ctx_overlay_info(const pfm_ctx *c, int64_t *n_patch_rows, int64_t *n_general_cells)
  {
    if (!c)
      return PFM_ERR_BAD_ARG;
    if (n_patch_rows)
      *n_patch_rows = (c->n_patch_blocks > 0 || !c->levels3.empty()) ? c->n_patch_rows : 0;
    if (n_general_cells)
      *n_general_cells = (c->n_patch_blocks > 0 || !c->levels3.empty()) ? c->n_general_cells : c->v.n_cells;
    return PFM_OK;
  }

  int64_t pfm_ctx_device_bytes(const pfm_ctx *c) { return c ? c->device_bytes : 0; }
}
