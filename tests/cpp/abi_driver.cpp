// abi_driver.cpp — a C++ host drives the assembly through the C ABI only (no Python, no torch):
//   pfm_ctx_create -> pfm_pattern_get/bind -> pfm_comm_create (1-rank RCCL communicator) -> pfm_state_set ->
//   pfm_halo_exchange (ncclSend/ncclRecv to itself) -> pfm_assemble_device -> compare.
// It is what the deal.II glue of INTEGRATION.md does per assemble_system() call (cracks.cc:2147-2154 + 2200-2475).
//
// Check: context A = a box whose high-x node plane is a GHOST layer that receives, over RCCL, the values of the owned
// plane x = 0 (a self-exchange: the only peer is this rank).  Context B = the same mesh with every node owned and the
// same values written directly.  The owned rows of A (residual + every matrix block) must equal those of B.
//
// With `--ranks R` (R > 1, one GPU per rank; tests/test_cpp_driver.py runs R = 2 wherever the box has two GPUs) the
// process forks R ranks BEFORE any HIP call; rank 0 makes the RCCL id and hands it to the others through pipes (the
// stand-in for the host application's MPI_Bcast).  Every rank holds the same kind of box with its own field values, the
// ghost plane of rank r receives the plane x = 0 of rank (r + 1) % R: a ring of real ncclSend / ncclRecv pairs over
// xGMI.  The reference values of the neighbour are a closed-form function of (rank, node), so each rank checks locally.
//
// Build (tests/test_cpp_driver.py does it): hipcc -std=c++17 abi_driver.cpp -I../../include -L<libdir> -lpfm_hip
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <sys/wait.h>
#include <unistd.h>

#include "pfm_assemble.h"

#define REQUIRE(cond, ...)                                                                                    \
  do                                                                                                          \
    {                                                                                                         \
      if (!(cond))                                                                                            \
        {                                                                                                     \
          fprintf(stderr, "abi_driver: %s:%d: ", __FILE__, __LINE__);                                         \
          fprintf(stderr, __VA_ARGS__);                                                                       \
          fprintf(stderr, "\n");                                                                              \
          return 1;                                                                                           \
        }                                                                                                     \
    }                                                                                                         \
  while (0)
#define PFM(call, ctx)                                                                                        \
  do                                                                                                          \
    {                                                                                                         \
      const int rc_ = (call);                                                                                 \
      REQUIRE(rc_ == PFM_OK, "%s -> %d (%s)", #call, rc_, (ctx) ? pfm_last_error(ctx) : "");                  \
    }                                                                                                         \
  while (0)

static double noise(uint64_t i, uint64_t salt)
{
  uint64_t x = (i + 7919 * salt + 1234) * 0x9E3779B97F4A7C15ull;
  x ^= x >> 29;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 32;
  return (double)(x >> 11) / (double)(1ull << 53);
}

static int run_rank(int nx, int ny, int nz, int rank, int n_ranks, int id_rd, const std::vector<int> &id_wr)
{
  const int dim = 3, NX = nx + 1, NY = ny + 1, NZ = nz + 1;
  const int N = NX * NY * NZ, NO = (NX - 1) * NY * NZ, NG = N - NO;
  // local numbering: owned nodes (i < NX-1) lexicographic first, then the ghost plane i = NX-1
  std::vector<int32_t> id_of((size_t)N);
  {
    int no = 0, ng = 0;
    for (int k = 0; k < NZ; ++k)
      for (int j = 0; j < NY; ++j)
        for (int i = 0; i < NX; ++i)
          id_of[i + NX * (j + NY * k)] = (i < NX - 1) ? no++ : NO + ng++;
  }
  std::vector<double> coords((size_t)N * 3);
  const double h = 20.0 / nx;
  for (int k = 0; k < NZ; ++k)
    for (int j = 0; j < NY; ++j)
      for (int i = 0; i < NX; ++i)
        {
          const int n = id_of[i + NX * (j + NY * k)];
          coords[3 * n + 0] = -10.0 + h * i;
          coords[3 * n + 1] = -10.0 + h * j;
          coords[3 * n + 2] = -10.0 + h * k;
        }
  std::vector<int32_t> cells((size_t)nx * ny * nz * 8);
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i)
        for (int a = 0; a < 8; ++a)
          cells[8 * ((size_t)i + nx * (j + (size_t)ny * k)) + a] = id_of[(i + (a & 1)) + NX * ((j + ((a >> 1) & 1)) + NY * (k + (a >> 2)))];

  // nodal fields of the FULL mesh (what context B sees); ghost node (NX-1,j,k) carries the values of (0,j,k)
  std::vector<double> U((size_t)N * 3), PHI(N), PO(N), POO(N);
  std::vector<uint8_t> flags(N, 0);
  for (int k = 0; k < NZ; ++k)
    for (int j = 0; j < NY; ++j)
      for (int i = 0; i < NX; ++i)
        {
          const int src = (i == NX - 1) ? 0 : i; // image of the plane x = 0 of rank (rank + 1) % n_ranks
          const uint64_t g = src + (uint64_t)NX * (j + (uint64_t)NY * k) +
                             (uint64_t)1000003 * (uint64_t)((i == NX - 1) ? (rank + 1) % n_ranks : rank);
          const int n = id_of[i + NX * (j + NY * k)];
          const bool bnd = j == 0 || j == NY - 1 || k == 0 || k == NZ - 1;
          for (int d = 0; d < 3; ++d)
            U[3 * n + d] = bnd ? 0.0 : 2e-3 * (noise(g, d) - 0.5);
          PHI[n] = 0.2 + 0.6 * noise(g, 10);
          PO[n] = 0.2 + 0.6 * noise(g, 11);
          POO[n] = 0.2 + 0.6 * noise(g, 12);
          flags[n] = bnd ? 0x7 : ((noise(g, 13) < 0.05) ? 0x8 : 0x0); // u = 0 on four faces, a few active-set nodes
        }
  auto pack = [&](int n_own, const std::vector<double> &pf, bool with_u) {
    std::vector<double> v((size_t)n_own * 4, 0.0);
    for (int n = 0; n < n_own; ++n)
      {
        if (with_u)
          for (int d = 0; d < 3; ++d)
            v[(size_t)3 * n + d] = U[3 * n + d];
        v[(size_t)3 * n_own + n] = pf[n];
      }
    return v;
  };

  pfm_params prm{};
  prm.mu = 1.0 / 2.4;
  prm.lambda = 2 * 0.2 * prm.mu / (1.0 - 0.4);
  prm.G_c = 1.0;
  prm.alpha_eps = 2.0 * h * std::sqrt(3.0);
  prm.constant_k = 1e-8;
  prm.pressure = 1e-3;
  prm.timestep = prm.time = prm.old_timestep = prm.old_old_timestep = 1.0;
  prm.outer_solver = PFM_SOLVER_ACTIVE_SET;

  pfm_mesh_desc md{};
  md.dim = dim;
  md.layout = PFM_LAYOUT_BLOCKED;
  md.n_nodes = N;
  md.n_cells = (int64_t)nx * ny * nz;
  md.cell_nodes = cells.data();
  md.coords = coords.data();
  md.box_cells[0] = nx;
  md.box_cells[1] = ny;
  md.box_cells[2] = nz;

  int n_dev = 0;
  REQUIRE(hipGetDeviceCount(&n_dev) == hipSuccess && n_dev > 0, "no HIP device");
  const int device = rank % n_dev;
  REQUIRE(n_ranks == 1 || n_dev >= n_ranks, "%d ranks need %d GPUs, %d visible", n_ranks, n_ranks, n_dev);
  REQUIRE(hipSetDevice(device) == hipSuccess, "hipSetDevice(%d)", device);
  hipStream_t stream;
  REQUIRE(hipStreamCreate(&stream) == hipSuccess, "stream");

  // ---------------- context A: owned + ghost plane, ghost values over RCCL
  pfm_ctx *A = nullptr;
  md.n_owned_nodes = NO;
  PFM(pfm_ctx_create(&A, &md, device), A);
  PFM(pfm_ctx_set_stream(A, stream), A);
  PFM(pfm_set_params(A, &prm), A);
  PFM(pfm_set_constraints(A, flags.data()), A);
  const int path = pfm_ctx_kernel_path(A);
  // the host's own CSR: here a copy of the canonical pattern with 32-bit row pointers, bound back as a host would
  int64_t nnzA[4], rowsA[4];
  std::vector<std::vector<int64_t>> rpA(4);
  for (int b = 0; b < 4; ++b)
    {
      PFM(pfm_pattern_size(A, b, &rowsA[b], &nnzA[b]), A);
      rpA[b].resize(rowsA[b] + 1);
      std::vector<int32_t> ci(nnzA[b]), rp32(rowsA[b] + 1);
      PFM(pfm_pattern_get(A, b, rpA[b].data(), ci.data()), A);
      for (int64_t r = 0; r <= rowsA[b]; ++r)
        rp32[r] = (int32_t)rpA[b][r];
      for (int64_t r = 0; r < rowsA[b]; ++r)
        REQUIRE(std::is_sorted(ci.begin() + rpA[b][r], ci.begin() + rpA[b][r + 1]), "block %d row %lld: columns not ascending", b, (long long)r);
      PFM(pfm_pattern_bind_i32(A, b, rp32.data(), ci.data()), A);
    }
  // halo lists: the owned plane i = 0 goes to rank - 1, the ghost plane comes from rank + 1 (the same rank for
  // n_ranks <= 2: one peer with both lists; otherwise a send-only and a receive-only peer)
  std::vector<int32_t> send_nodes, recv_nodes;
  for (int k = 0; k < NZ; ++k)
    for (int j = 0; j < NY; ++j)
      {
        send_nodes.push_back(id_of[0 + NX * (j + NY * k)]);
        recv_nodes.push_back(id_of[(NX - 1) + NX * (j + NY * k)]);
      }
  REQUIRE((int)recv_nodes.size() == NG, "ghost count");
  const int to = (rank + n_ranks - 1) % n_ranks, from = (rank + 1) % n_ranks;
  const int64_t ns = (int64_t)send_nodes.size(), nr = (int64_t)recv_nodes.size();
  int n_peers = 1;
  int peer_ranks[2] = {to, from};
  int64_t sp[3] = {0, ns, ns}, rp[3] = {0, nr, nr};
  if (to != from)
    {
      n_peers = 2;
      rp[1] = 0; // peer 0 = `to`: send only; peer 1 = `from`: receive only
    }
  PFM(pfm_halo_register(A, n_peers, sp, send_nodes.data(), rp, recv_nodes.data()), A);
  uint8_t uid[PFM_COMM_ID_BYTES];
  if (rank == 0)
    {
      PFM(pfm_comm_unique_id(uid), A);
      for (int fd : id_wr)
        REQUIRE(write(fd, uid, sizeof(uid)) == (ssize_t)sizeof(uid), "id pipe (write)");
    }
  else
    REQUIRE(read(id_rd, uid, sizeof(uid)) == (ssize_t)sizeof(uid), "id pipe (read)");
  void *comm = nullptr;
  PFM(pfm_comm_create(&comm, uid, n_ranks, rank, device), A);

  const std::vector<double> solA = pack(NO, PHI, true), oldA = pack(NO, PO, false), ooA = pack(NO, POO, false);
  double *d_res = nullptr, *d_tot = nullptr, *d_val[4] = {nullptr, nullptr, nullptr, nullptr};
  REQUIRE(hipMalloc((void **)&d_res, sizeof(double) * 4 * NO) == hipSuccess, "hipMalloc");
  REQUIRE(hipMalloc((void **)&d_tot, sizeof(double) * 4 * NO) == hipSuccess, "hipMalloc");
  for (int b = 0; b < 4; ++b)
    REQUIRE(hipMalloc((void **)&d_val[b], sizeof(double) * std::max<int64_t>(nnzA[b], 1)) == hipSuccess, "hipMalloc");
  std::vector<double> resA(4 * (size_t)NO), totA(4 * (size_t)NO);
  std::vector<std::vector<double>> valA(4);
  for (int pass = 0; pass < 2; ++pass) // 0: residual only, 1: Jacobian + residual
    {
      PFM(pfm_state_set(A, solA.data(), oldA.data(), ooA.data(), 0), A);
      PFM(pfm_halo_exchange(A, comm, peer_ranks), A);
      PFM(pfm_assemble_device(A, pass == 0, d_val, d_res, d_tot), A);
      PFM(pfm_sync_status(A), A);
      if (pass == 0)
        {
          REQUIRE(hipMemcpy(totA.data(), d_tot, sizeof(double) * 4 * NO, hipMemcpyDeviceToHost) == hipSuccess, "copy");
          PFM(pfm_check_finite(A, d_tot, 4 * (int64_t)NO), A);
        }
    }
  REQUIRE(hipMemcpy(resA.data(), d_res, sizeof(double) * 4 * NO, hipMemcpyDeviceToHost) == hipSuccess, "copy");
  for (int b = 0; b < 4; ++b)
    {
      valA[b].resize(nnzA[b]);
      REQUIRE(hipMemcpy(valA[b].data(), d_val[b], sizeof(double) * nnzA[b], hipMemcpyDeviceToHost) == hipSuccess, "copy");
      PFM(pfm_check_finite(A, d_val[b], nnzA[b]), A);
    }
  // the same with the ghost import hidden behind the interior tiles (pfm_assemble_overlapped): bit for bit the same rows.
  // The ghost plane is first overwritten with garbage so that a boundary tile assembled before the import shows up.
  {
    std::vector<double> junk(4 * (size_t)NO, 1.0e30);
    for (int pass = 0; pass < 2; ++pass)
      {
        PFM(pfm_state_set(A, junk.data(), junk.data(), junk.data(), 0), A);
        PFM(pfm_halo_exchange(A, comm, peer_ranks), A); // ghosts := 1e30
        PFM(pfm_state_set(A, solA.data(), oldA.data(), ooA.data(), 0), A);
        REQUIRE(hipMemset(d_res, 0xff, sizeof(double) * 4 * NO) == hipSuccess, "memset");
        REQUIRE(hipMemset(d_tot, 0xff, sizeof(double) * 4 * NO) == hipSuccess, "memset");
        PFM(pfm_assemble_overlapped(A, comm, peer_ranks, pass == 0, d_val, d_res, d_tot), A);
        PFM(pfm_sync_status(A), A);
        std::vector<double> r2(4 * (size_t)NO);
        if (pass == 0)
          {
            REQUIRE(hipMemcpy(r2.data(), d_tot, sizeof(double) * 4 * NO, hipMemcpyDeviceToHost) == hipSuccess, "copy");
            REQUIRE(std::equal(r2.begin(), r2.end(), totA.begin()), "overlapped residual-only assembly differs from the sequential one");
          }
        else
          {
            REQUIRE(hipMemcpy(r2.data(), d_res, sizeof(double) * 4 * NO, hipMemcpyDeviceToHost) == hipSuccess, "copy");
            REQUIRE(std::equal(r2.begin(), r2.end(), resA.begin()), "overlapped assembly: residual differs from the sequential one");
            for (int b = 0; b < 4; ++b)
              {
                std::vector<double> v2(nnzA[b]);
                REQUIRE(hipMemcpy(v2.data(), d_val[b], sizeof(double) * nnzA[b], hipMemcpyDeviceToHost) == hipSuccess, "copy");
                REQUIRE(std::equal(v2.begin(), v2.end(), valA[b].begin()), "overlapped assembly: block %d differs from the sequential one", b);
              }
          }
      }
  }

  // ---------------- context B: every node owned, values given directly; synchronous host-pointer call
  pfm_ctx *B = nullptr;
  md.n_owned_nodes = N;
  PFM(pfm_ctx_create(&B, &md, device), B);
  PFM(pfm_set_params(B, &prm), B);
  PFM(pfm_set_constraints(B, flags.data()), B);
  const std::vector<double> solB = pack(N, PHI, true), oldB = pack(N, PO, false), ooB = pack(N, POO, false);
  std::vector<double> resB(4 * (size_t)N), totB(4 * (size_t)N);
  int64_t nnzB[4];
  std::vector<std::vector<double>> valB(4);
  double *pv[4];
  for (int b = 0; b < 4; ++b)
    {
      PFM(pfm_pattern_size(B, b, nullptr, &nnzB[b]), B);
      valB[b].resize(nnzB[b]);
      pv[b] = valB[b].data();
    }
  PFM(pfm_assemble(B, solB.data(), oldB.data(), ooB.data(), 1, nullptr, resB.data(), totB.data()), B);
  PFM(pfm_assemble(B, solB.data(), oldB.data(), ooB.data(), 0, pv, resB.data(), nullptr), B);

  // ---------------- compare the owned rows
  double scale = 1.0, err = 0.0;
  auto cmp = [&](double a, double b) {
    scale = std::max(scale, std::fabs(b));
    err = std::max(err, std::fabs(a - b));
  };
  for (int n = 0; n < NO; ++n)
    {
      for (int d = 0; d < 3; ++d)
        {
          cmp(resA[(size_t)3 * n + d], resB[(size_t)3 * n + d]);
          cmp(totA[(size_t)3 * n + d], totB[(size_t)3 * n + d]);
        }
      cmp(resA[(size_t)3 * NO + n], resB[(size_t)3 * N + n]);
      cmp(totA[(size_t)3 * NO + n], totB[(size_t)3 * N + n]);
    }
  // owned nodes come first in both numberings and have the same neighbours in the same (ascending) order:
  // the values of A's blocks are a prefix of B's
  for (int b = 0; b < 4; ++b)
    {
      REQUIRE(nnzA[b] <= nnzB[b], "block size");
      for (int64_t e = 0; e < nnzA[b]; ++e)
        cmp(valA[b][e], valB[b][e]);
    }
  const double rel = err / scale;
  printf("abi_driver: rank %d of %d on GPU %d, box %dx%dx%d, %d owned + %d ghost nodes, kernel path %d, RCCL %s of %zu bytes: "
         "max |A - B| / max(1,|B|) = %.3e\n",
         rank, n_ranks, device, nx, ny, nz, NO, NG, path, n_ranks == 1 ? "self-exchange" : "ring exchange",
         send_nodes.size() * PFM_HALO_DOUBLES_PER_NODE(3) * sizeof(double), rel);
  REQUIRE(rel < 1e-12, "owned rows differ between the RCCL-fed and the directly fed context");
  double nrm = 0;
  for (double x : resA)
    nrm += x * x;
  REQUIRE(nrm > 0.0, "residual is identically zero");
  PFM(pfm_comm_destroy(comm), A);
  PFM(pfm_ctx_destroy(A), (pfm_ctx *)nullptr);
  PFM(pfm_ctx_destroy(B), (pfm_ctx *)nullptr);
  return 0;
}

int main(int argc, char **argv)
{
  int dims[3] = {17, 9, 11}, nd = 0, n_ranks = 1;
  for (int a = 1; a < argc; ++a)
    {
      if (!strcmp(argv[a], "--ranks") && a + 1 < argc)
        n_ranks = atoi(argv[++a]);
      else if (nd < 3)
        dims[nd++] = atoi(argv[a]);
    }
  if (n_ranks < 1 || n_ranks > 64)
    return 2;
  if (n_ranks == 1)
    {
      const int rc = run_rank(dims[0], dims[1], dims[2], 0, 1, -1, {});
      if (rc == 0)
        printf("abi_driver: OK\n");
      return rc;
    }
  // one process per rank, forked before the first HIP call of this process; the RCCL id travels through pipes
  std::vector<int> rd(n_ranks, -1), wr;
  for (int r = 1; r < n_ranks; ++r)
    {
      int fd[2];
      if (pipe(fd) != 0)
        return 2;
      rd[r] = fd[0];
      wr.push_back(fd[1]);
    }
  std::vector<pid_t> kids;
  for (int r = 0; r < n_ranks; ++r)
    {
      const pid_t pid = fork();
      if (pid < 0)
        return 2;
      if (pid == 0)
        _exit(run_rank(dims[0], dims[1], dims[2], r, n_ranks, rd[r], r == 0 ? wr : std::vector<int>{}));
      kids.push_back(pid);
    }
  int bad = 0;
  for (pid_t pid : kids)
    {
      int st = 0;
      if (waitpid(pid, &st, 0) != pid || !WIFEXITED(st) || WEXITSTATUS(st) != 0)
        ++bad;
    }
  if (bad)
    {
      fprintf(stderr, "abi_driver: %d of %d ranks failed\n", bad, n_ranks);
      return 1;
    }
  printf("abi_driver: OK (%d ranks)\n", n_ranks);
  return 0;
}
