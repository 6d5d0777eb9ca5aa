// glue/cracks_gpu_assemble.cc — the deal.II / Trilinos side of the drop-in: what a maintainer of tjhei/cracks adds to
// cracks.cc so that assemble_system() / assemble_nl_residual() run through include/pfm_assemble.h.
//
//   STATUS (round 4): COMPILED AND RUN AGAINST A MOCK, NEVER AGAINST deal.II.  deal.II, Trilinos and p4est do not exist in
//   the image this repository is built in (cracks.cc needs deal.II >= 9.5 with Trilinos + p4est, CMakeLists.txt:14-36).
//   tests/cpp/mock_dealii/ declares just the types this file touches (test scaffolding, not deal.II);
//   tests/cpp/glue_driver.cpp builds a Problem from a mesh on disk and calls rebuild() + assemble(); tests/test_glue_mock.py
//   compiles it on every CPU run and, on the GPU, compares every matrix entry and both residuals with the oracle for 2-D /
//   3-D hanging-node meshes, the slit mesh and both dof layouts, with the mock's vertex numbering a random permutation of
//   the mesh's (1 rank).  The library calls are additionally driven by tests/cpp/abi_driver.cpp (1 rank, and 2 forked
//   ranks over RCCL wherever two GPUs are visible).
//   What no test here can check is whether REAL deal.II / Epetra behave as this file assumes.  The three assumptions are
//   isolated in one function each and marked "deal.II-knowledge -- verify on a real install":
//     (A) global_dof_of():               the dof numbering of FESystem(FE_Q(1)^(dim+1)) without / with component_wise
//     (B) local_column_of_global_dof():   block-local index <-> vertex rank of the 2x2 block matrices
//     (C) gather_owned()/scatter_owned(): the owned part of a Trilinos vector is contiguous in ascending global index
//
// How to use: add  #include "glue/cracks_gpu_assemble.cc"  behind the class definition in cracks.cc, add the members of
// PfmGlue<dim> (one object `pfm_glue`) to FracturePhaseFieldProblem<dim>, call pfm_glue.rebuild(*this) at the end of
// setup_system() (cracks.cc:1579-1680; called from run() 4174 and refine_mesh() 4148) and replace the body of
// assemble_system(bool) (cracks.cc:2129-2475; the AMG set-up 2477-2497 stays) by pfm_glue.assemble(*this, residual_only).
// The glue is a friend-less template over the problem class: it touches only the members named in SURVEY.md 8(a).
//
// Design notes
//   * Node numbering handed to the library: the phase-field dof of a vertex is the key of a node.  Owned nodes first,
//     in ascending global phase-field index (= the order of the Epetra row map of the phi block and, divided by dim, of
//     the u block: DoFRenumbering::component_wise keeps the relative order of the dofs of a block; deal.II-knowledge),
//     ghost nodes behind them in ascending global index.
//   * Cells handed over: locally owned + ghost cells (owner computes: every owned row is complete locally, no
//     compress(add)) + the cells other ranks SHIP: a cell that reaches a row of this rank only through a HANGING vertex
//     whose parent is owned here need not be in deal.II's one-cell ghost layer (it shares the hanging vertex and the OTHER
//     end of the coarse edge with the coarse neighbour); the reference repairs such rows with compress(add),
//     cracks.cc:2470-2475.  Here the owner of such a cell sends it -- vertex dofs, coordinates, hanging lines -- to the
//     owners of the parents once per setup_system (ship_hanging_closure_cells), the receiver adds the cells it does not
//     hold yet; their vertices become ghost nodes whose values arrive with the ordinary ghost import and whose
//     constraint flags arrive with a few bytes per assemble (exchange_shipped_flags).  Round 3 threw here.  The
//     algorithm is the one of cracks_amd/partition.py: hanging_closure_shipments, tested there on 2-4 ranks (CPU: the
//     missing rows are wrong without it and right with it; GPU: tests/test_gpu_multirank.py "dealii+shipped").
//   * 32-bit limit: Epetra's local CSR has int offsets, pfm_pattern_bind_i32 takes them as they are; a block with more
//     than 2^31-1 entries per rank cannot exist in Epetra either (the 216^3 single-rank (u,u) block has 2.48e9: such a
//     run needs >= 2 ranks, or the library's own 64-bit pattern through pfm_pattern_get).
#ifdef PFM_WITH_DEALII

#include <deal.II/base/index_set.h>
#include <deal.II/base/mpi.h>
#include <deal.II/dofs/dof_handler.h>
#include <deal.II/dofs/dof_tools.h>
#include <deal.II/lac/affine_constraints.h>
#include <deal.II/lac/trilinos_block_sparse_matrix.h>
#include <deal.II/lac/trilinos_parallel_block_vector.h>

#include <Epetra_CrsMatrix.h>
#include <Epetra_Map.h>

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <set>
#include <vector>

extern "C"
{
#include "pfm_assemble.h"
#include "pfm_newton.h"
}

namespace pfm_glue_detail
{
  using namespace dealii;
  using gidx = types::global_dof_index;

#define PFM_CALL(ctx, call)                                                                                     \
  do                                                                                                            \
    {                                                                                                           \
      const int pfm_rc_ = (call);                                                                               \
      AssertThrow(pfm_rc_ == PFM_OK, ExcMessage(std::string(#call) + ": " + ((ctx) ? pfm_last_error(ctx) : "no context")));   \
    }                                                                                                           \
  while (0)

  template <int dim>
  struct PfmGlue
  {
    pfm_ctx *ctx = nullptr;
    void *comm = nullptr; // ncclComm_t, made once per program (collective)
    int device = 0;
    bool blocked = true; // PFM_LAYOUT_BLOCKED (iterative solver) / INTERLEAVED (direct solver), cracks.cc:1587-1590

    // rank-local node numbering
    std::map<gidx, int32_t> node_of_phi_dof; // global phase-field dof -> local node (owned first)
    std::vector<gidx> phi_dof_of_node;       // inverse
    int32_t n_owned = 0, n_nodes = 0;
    // ghost import lists (node level), grouped by peer rank
    std::vector<int> peer_ranks;
    std::vector<int64_t> send_ptr, recv_ptr;
    std::vector<int32_t> send_nodes, recv_nodes;
    // device mirrors of the owned vectors and of the outputs
    double *d_vec[3] = {nullptr, nullptr, nullptr}, *d_res[2] = {nullptr, nullptr}, *d_val[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t nnz[4] = {0, 0, 0, 0};
    std::vector<double> h_vec[3], h_res[2];
    std::vector<uint8_t> flags;
    // cells shipped by other ranks (hanging-node closure) and what is needed to keep their nodes' flags current
    struct ShippedCell
    {
      gidx phi_dof[1u << dim];
      double xyz[1u << dim][dim];
    };
    std::vector<ShippedCell> shipped_cells;
    std::map<gidx, std::vector<std::pair<gidx, double>>> shipped_lines; // hanging lines of their vertices (phase-field dofs)
    std::map<unsigned int, std::vector<gidx>> flags_wanted_from, flags_asked_by; // per assemble: flag bytes of shipped nodes
    bool flags_shipped_anywhere = false; // some rank of the communicator wants such bytes (decided collectively in rebuild)
    IndexSet relevant_dofs;              // locally relevant dofs: the lines deal.II's AffineConstraints objects can be asked about
    bool state_complete = false; // all three vectors have been scattered into this context once
    // Opt-in switches of the host (both off: the behaviour of a plain drop-in).
    //   pin_host_matrix:   rebuild() page-locks Epetra's value arrays (pfm_host_register) so that the 30 GB of a 3-D Jacobian
    //                      travel as DMA at the link rate.  The arrays belong to system_pde_matrix, which setup_system()
    //                      reinitialises (cracks.cc:1583, 1653): a host that switches this on MUST call before_setup_system()
    //                      at the top of setup_system() -- rebuild() refuses to run over a context that still holds locks.
    //   residual_to_host:  false = a residual-only assemble() leaves both residual vectors on the device and the caller asks
    //                      for what the line search reads, residual_l2_norm() (cracks.cc:2946-2949): 24 bytes instead of
    //                      2 x 8 n_dofs per call.
    bool pin_host_matrix = false;
    bool residual_to_host = true;
    bool holds_host_pins = false;

    ~PfmGlue()
    {
      release();
      if (comm)
        pfm_comm_destroy(comm);
    }

    // top of setup_system() (cracks.cc:1579), BEFORE system_pde_matrix.reinit / clear: drops the context and with it every
    // page lock on arrays the matrix is about to free
    void before_setup_system() { release(); }

    void release()
    {
      if (ctx)
        pfm_ctx_destroy(ctx); // (unregisters every host array of the context)
      ctx = nullptr;
      holds_host_pins = false;
      for (double *&p : d_vec)
        {
          if (p)
            (void)hipFree(p);
          p = nullptr;
        }
      for (double *&p : d_res)
        {
          if (p)
            (void)hipFree(p);
          p = nullptr;
        }
      for (double *&p : d_val)
        {
          if (p)
            (void)hipFree(p);
          p = nullptr;
        }
    }

    // ---------------------------------------------------------------------------------------------------------
    // setup_system() epilogue: mesh tables, hanging-node table, pattern bind, halo lists, communicator
    template <class Problem>
    void rebuild(Problem &P)
    {
      // the arrays page-locked by the last rebuild() belonged to the matrix setup_system() has just reinitialised: locked
      // pages would have been freed under the lock and hipHostUnregister would run on dangling pointers
      AssertThrow(!(ctx && holds_host_pins),
                  ExcMessage("PfmGlue::rebuild: the previous context still page-locks the old matrix arrays -- call "
                             "before_setup_system() at the top of setup_system() (INTEGRATION.md), or leave pin_host_matrix off"));
      release();
      state_complete = false;
      const auto &dh = P.dof_handler;
      const auto &fe = dh.get_fe();
      const MPI_Comm mpi = P.mpi_com;
      const unsigned int me = Utilities::MPI::this_mpi_process(mpi), n_ranks = Utilities::MPI::n_mpi_processes(mpi);
      blocked = !P.direct_solver;
      const unsigned int phi_comp = dim; // components 0..dim-1 = u, dim = phi (cracks.cc:980-996)
      AssertThrow(fe.degree == 1 && fe.n_components() == dim + 1, ExcMessage("the GPU assembly is written for Q1/Q1 (FE degree 1)"));
      const IndexSet &owned = dh.locally_owned_dofs();
      relevant_dofs = DoFTools::extract_locally_relevant_dofs(dh);

      // ---- 0. hanging-node closure: cells of other ranks that reach a row owned here through a hanging vertex
      const std::vector<IndexSet> owned_per_rank = Utilities::MPI::all_gather(mpi, owned);
      ship_hanging_closure_cells(P, owned_per_rank, me);

      // ---- 1. nodes: every vertex of a locally owned, ghost or shipped cell, keyed by its phase-field dof
      std::set<gidx> owned_phi, ghost_phi;
      for (const auto &cell : dh.active_cell_iterators())
        if (cell->is_locally_owned() || cell->is_ghost())
          for (unsigned int v = 0; v < GeometryInfo<dim>::vertices_per_cell; ++v)
            {
              const gidx g = cell->vertex_dof_index(v, phi_comp); // FESystem(FE_Q(1)^(dim+1)): vertex dof i = component i
              (owned.is_element(g) ? owned_phi : ghost_phi).insert(g);
            }
      for (const ShippedCell &sc : shipped_cells)
        for (unsigned int v = 0; v < GeometryInfo<dim>::vertices_per_cell; ++v)
          (owned.is_element(sc.phi_dof[v]) ? owned_phi : ghost_phi).insert(sc.phi_dof[v]);
      for (const auto &kv : shipped_lines) // parents of the hanging vertices of shipped cells are local nodes too
        for (const auto &e : kv.second)
          (owned.is_element(e.first) ? owned_phi : ghost_phi).insert(e.first);
      node_of_phi_dof.clear();
      phi_dof_of_node.clear();
      for (const gidx g : owned_phi)
        {
          node_of_phi_dof[g] = (int32_t)phi_dof_of_node.size();
          phi_dof_of_node.push_back(g);
        }
      n_owned = (int32_t)phi_dof_of_node.size();
      for (const gidx g : ghost_phi)
        {
          node_of_phi_dof[g] = (int32_t)phi_dof_of_node.size();
          phi_dof_of_node.push_back(g);
        }
      n_nodes = (int32_t)phi_dof_of_node.size();

      // ---- 2. cells and coordinates (deal.II vertex order = the order pfm_mesh_desc expects)
      constexpr unsigned int nv = GeometryInfo<dim>::vertices_per_cell;
      std::vector<int32_t> cell_nodes;
      std::vector<double> coords((size_t)n_nodes * dim, 0.0);
      std::vector<double> cell_lambda, cell_mu;
      const bool het = P.test_case == Problem::TestCase::multiple_het;
      for (const auto &cell : dh.active_cell_iterators())
        if (cell->is_locally_owned() || cell->is_ghost())
          {
            for (unsigned int v = 0; v < nv; ++v)
              {
                const int32_t n = node_of_phi_dof.at(cell->vertex_dof_index(v, phi_comp));
                cell_nodes.push_back(n);
                for (unsigned int d = 0; d < dim; ++d)
                  coords[(size_t)n * dim + d] = cell->vertex(v)[d];
              }
            if (het)
              {
                // cracks.cc:2207-2216: E from the bitmap at the cell centre, +1.0, nu = poisson_ratio_nu
                const double E = P.func_emodulus->value(cell->center(), 0) + 1.0;
                const double mu = E / (2.0 * (1.0 + P.poisson_ratio_nu));
                cell_mu.push_back(mu);
                cell_lambda.push_back(2.0 * P.poisson_ratio_nu * mu / (1.0 - 2.0 * P.poisson_ratio_nu));
              }
          }
      for (const ShippedCell &sc : shipped_cells)
        {
          Point<dim> centre;
          for (unsigned int v = 0; v < nv; ++v)
            {
              const int32_t n = node_of_phi_dof.at(sc.phi_dof[v]);
              cell_nodes.push_back(n);
              for (unsigned int d = 0; d < dim; ++d)
                {
                  coords[(size_t)n * dim + d] = sc.xyz[v][d];
                  centre[d] += sc.xyz[v][d] / double(nv);
                }
            }
          if (het)
            {
              const double E = P.func_emodulus->value(centre, 0) + 1.0;
              const double mu = E / (2.0 * (1.0 + P.poisson_ratio_nu));
              cell_mu.push_back(mu);
              cell_lambda.push_back(2.0 * P.poisson_ratio_nu * mu / (1.0 - 2.0 * P.poisson_ratio_nu));
            }
        }

      // ---- 3. hanging nodes: constraints_hanging_nodes (cracks.cc:1630-1635) at node level; the lines of all
      // components of a vertex are the same, the phase-field line is taken
      std::vector<int32_t> hn_nodes, hn_parents;
      std::vector<int64_t> hn_ptr(1, 0);
      std::vector<double> hn_w;
      for (int32_t n = 0; n < n_nodes; ++n)
        {
          const gidx g = phi_dof_of_node[n];
          const std::vector<std::pair<gidx, double>> *line = nullptr;
          if (relevant_dofs.is_element(g) && P.constraints_hanging_nodes.is_constrained(g)) // (asked about its local lines only)
            line = P.constraints_hanging_nodes.get_constraint_entries(g);
          else if (shipped_lines.count(g)) // a vertex of a shipped cell outside the locally relevant dofs
            line = &shipped_lines.at(g);
          else
            continue;
          AssertThrow(line != nullptr && !line->empty(), ExcMessage("hanging-node line without entries"));
          hn_nodes.push_back(n);
          for (const auto &e : *line)
            {
              const auto it = node_of_phi_dof.find(e.first);
              AssertThrow(it != node_of_phi_dof.end(), ExcMessage("parent of a hanging node is not a local node"));
              hn_parents.push_back(it->second);
              hn_w.push_back(e.second);
            }
          hn_ptr.push_back((int64_t)hn_parents.size());
        }

      // ---- 4. context
      pfm_mesh_desc m{};
      m.dim = dim;
      m.layout = blocked ? PFM_LAYOUT_BLOCKED : PFM_LAYOUT_INTERLEAVED;
      m.n_nodes = n_nodes;
      m.n_owned_nodes = n_owned;
      m.n_cells = (int64_t)(cell_nodes.size() / nv);
      m.cell_nodes = cell_nodes.data();
      m.coords = coords.data();
      m.cell_lambda = het ? cell_lambda.data() : nullptr;
      m.cell_mu = het ? cell_mu.data() : nullptr;
      m.n_hanging = (int32_t)hn_nodes.size();
      m.hn_nodes = hn_nodes.data();
      m.hn_ptr = hn_ptr.data();
      m.hn_parents = hn_parents.data();
      m.hn_weights = hn_w.data();
      // box_cells stays 0: the library recognises a uniform box from the coordinates by itself
      int n_dev = 0;
      AssertThrow(hipGetDeviceCount(&n_dev) == hipSuccess && n_dev > 0, ExcMessage("no HIP device"));
      device = (int)(me % (unsigned int)n_dev);
      PFM_CALL(ctx, pfm_ctx_create(&ctx, &m, device));

      // ---- 5. the library adopts Trilinos' local CSR (cracks.cc:1644-1654): values[] is then written in Epetra's order
      const unsigned int nb1 = blocked ? 2 : 1;
      for (unsigned int r = 0; r < nb1; ++r)
        for (unsigned int c = 0; c < nb1; ++c)
          {
            const Epetra_CrsMatrix &A = P.system_pde_matrix.block(r, c).trilinos_matrix();
            int *rowptr = nullptr, *colind = nullptr;
            double *values = nullptr;
            AssertThrow(A.ExtractCrsDataPointers(rowptr, colind, values) == 0, ExcMessage("matrix storage is not optimised (FillComplete?)"));
            const int n_rows = A.NumMyRows();
            const int64_t n_entries = rowptr[n_rows];
            // Epetra's local column id = position in ITS column map; translate once to the library's local numbering:
            //   blocked: u column = node * dim + comp, phi column = node;   interleaved: node * (dim + 1) + comp
            std::vector<int32_t> col_lib((size_t)n_entries);
            const Epetra_BlockMap &cmap = A.ColMap();
            std::vector<int32_t> lib_of_lid((size_t)cmap.NumMyElements());
            for (int lid = 0; lid < cmap.NumMyElements(); ++lid)
              lib_of_lid[lid] = local_column_of_global_dof(P, r, c, (gidx)cmap.GID64(lid));
            for (int64_t e = 0; e < n_entries; ++e)
              col_lib[e] = lib_of_lid[colind[e]];
            const int block = blocked ? (int)(2 * r + c) : 0;
            PFM_CALL(ctx, pfm_pattern_bind_i32(ctx, block, rowptr, col_lib.data()));
            PFM_CALL(ctx, pfm_pattern_size(ctx, block, nullptr, &nnz[block]));
            AssertThrow(nnz[block] == n_entries, ExcMessage("pattern size mismatch"));
            AssertThrow(hipMalloc((void **)&d_val[block], sizeof(double) * (size_t)std::max<int64_t>(nnz[block], 1)) == hipSuccess,
                        ExcMessage("hipMalloc (matrix values)"));
            // Epetra's value array receives 35 GB per Jacobian at 1e7 cells: page-locked, the transfer is DMA at the link
            // rate (pageable memory: a fraction of it).  Best effort: if the pages cannot be locked the copy still works.
            // Opt-in (pin_host_matrix): the lock must be gone BEFORE the matrix is reinitialised in setup_system --
            // before_setup_system(), enforced at the top of this function.
            if (pin_host_matrix && n_entries > 0 && pfm_host_register(ctx, values, (int64_t)sizeof(double) * n_entries) == PFM_OK)
              holds_host_pins = true;
          }

      // ---- 6. ghost import lists (cracks.cc:2147-2154 at node level): who owns my ghost nodes, who needs my owned ones
      // owner of a ghost dof: the locally owned ranges of all ranks (deal.II-knowledge: Utilities::MPI::all_gather of
      // the IndexSets, or dh.compute_locally_owned_dofs_per_processor() in older versions)
      std::map<int, std::vector<gidx>> want_from; // rank -> ghost phi dofs I need (ascending)
      for (int32_t n = n_owned; n < n_nodes; ++n)
        {
          const gidx g = phi_dof_of_node[n];
          int owner = -1;
          for (unsigned int p = 0; p < n_ranks; ++p)
            if (owned_per_rank[p].is_element(g))
              {
                owner = (int)p;
                break;
              }
          AssertThrow(owner >= 0 && owner != (int)me, ExcMessage("ghost node without an owner"));
          want_from[owner].push_back(g);
        }
      // tell every owner which of its nodes I need: some_to_some exchange of index lists
      std::map<unsigned int, std::vector<gidx>> requests;
      for (const auto &kv : want_from)
        requests[(unsigned int)kv.first] = kv.second;
      const std::map<unsigned int, std::vector<gidx>> asked = Utilities::MPI::some_to_some(mpi, requests);
      std::set<int> peers;
      for (const auto &kv : want_from)
        peers.insert(kv.first);
      for (const auto &kv : asked)
        peers.insert((int)kv.first);
      peer_ranks.assign(peers.begin(), peers.end());
      send_ptr.assign(1, 0);
      recv_ptr.assign(1, 0);
      send_nodes.clear();
      recv_nodes.clear();
      for (const int p : peer_ranks)
        {
          const auto a = asked.find((unsigned int)p);
          if (a != asked.end())
            for (const gidx g : a->second) // the order the receiver listed them in = the order of its recv list
              send_nodes.push_back(node_of_phi_dof.at(g));
          send_ptr.push_back((int64_t)send_nodes.size());
          const auto w = want_from.find(p);
          if (w != want_from.end())
            for (const gidx g : w->second)
              recv_nodes.push_back(node_of_phi_dof.at(g));
          recv_ptr.push_back((int64_t)recv_nodes.size());
        }
      PFM_CALL(ctx, pfm_halo_register(ctx, (int)peer_ranks.size(), send_ptr.data(), send_nodes.data(), recv_ptr.data(), recv_nodes.data()));
      // constraint flags of ghost nodes that are NOT locally relevant (vertices of shipped cells): deal.II does not know
      // constraints_update there, their owners tell us at every assemble (exchange_shipped_flags)
      {
        flags_wanted_from.clear();
        for (const auto &kv : want_from)
          for (const gidx g : kv.second)
            if (!relevant_dofs.is_element(g))
              flags_wanted_from[(unsigned int)kv.first].push_back(g);
        flags_asked_by = Utilities::MPI::some_to_some(mpi, flags_wanted_from);
        // some_to_some is collective over the whole communicator: whether the per-assemble exchange runs at all is decided
        // HERE, once and by everybody -- a rank that takes no part in the shipping must still make the call (with empty maps)
        flags_shipped_anywhere = Utilities::MPI::max(flags_wanted_from.empty() && flags_asked_by.empty() ? 0u : 1u, mpi) != 0u;
      }

      // ---- 7. one RCCL communicator for the life of the program (collective); the id travels over MPI
      if (!comm && n_ranks > 1)
        {
          uint8_t id[PFM_COMM_ID_BYTES] = {0};
          int ok = 1;
          if (me == 0)
            ok = pfm_comm_unique_id(id) == PFM_OK;
          MPI_Bcast(&ok, 1, MPI_INT, 0, mpi);
          AssertThrow(ok, ExcMessage("pfm_comm_unique_id failed on rank 0 (RCCL unavailable?)"));
          MPI_Bcast(id, PFM_COMM_ID_BYTES, MPI_BYTE, 0, mpi);
          PFM_CALL(ctx, pfm_comm_create(&comm, id, (int)n_ranks, (int)me, device));
        }

      // ---- 8. device mirrors of the owned vectors / outputs
      const size_t nd = (size_t)n_owned * (dim + 1);
      for (double *&p : d_vec)
        AssertThrow(hipMalloc((void **)&p, sizeof(double) * std::max<size_t>(nd, 1)) == hipSuccess, ExcMessage("hipMalloc (vectors)"));
      for (double *&p : d_res)
        AssertThrow(hipMalloc((void **)&p, sizeof(double) * std::max<size_t>(nd, 1)) == hipSuccess, ExcMessage("hipMalloc (residuals)"));
      for (auto &h : h_vec)
        h.assign(nd, 0.0);
      for (auto &h : h_res)
        h.assign(nd, 0.0);
      flags.assign((size_t)n_nodes, 0);
    }

    // library column id (within block (r, c)'s column space) of a global dof
    // deal.II-knowledge (B) -- verify on a real install: for every locally relevant vertex compare
    //   cell->vertex_dof_index(v, comp) with the value this function inverts, e.g. in rebuild():
    //   Assert(local_column_of_global_dof(P, 0, comp < dim ? 0 : 1, block-local index of that dof) == node * dim + comp (or node))
    template <class Problem>
    int32_t local_column_of_global_dof(const Problem &P, unsigned int /*r*/, unsigned int c, gidx g_in_block) const
    {
      // global numbering: blocked = component_wise with blocks {u..u, phi} (cracks.cc:1587-1590): block 0 holds the
      // u dofs node-interleaved (dim per node), block 1 the phi dofs; Epetra's block matrices index each block from 0.
      // The node of a u dof (block 0): its phi dof is found through the vertex -> via the map below.
      if (!blocked)
        {
          // interleaved: dof = (dim + 1) * vertex-rank + comp in the UNrenumbered enumeration; the phi dof of the same
          // vertex is g - comp + dim
          const unsigned int comp = (unsigned int)(g_in_block % (dim + 1));
          const auto it = node_of_phi_dof.find(g_in_block - comp + dim);
          AssertThrow(it != node_of_phi_dof.end(), ExcMessage("matrix column outside the local nodes"));
          return it->second * (dim + 1) + (int32_t)comp;
        }
      if (c == 1) // phi column: block-local index -> global index = offset of block 1 + index
        {
          const auto it = node_of_phi_dof.find(P.n_u_dofs_global() + g_in_block);
          AssertThrow(it != node_of_phi_dof.end(), ExcMessage("matrix column outside the local nodes"));
          return it->second;
        }
      // u column: block-local index = dim * (global vertex rank) + comp; the same vertex rank indexes block 1
      const unsigned int comp = (unsigned int)(g_in_block % dim);
      const auto it = node_of_phi_dof.find(P.n_u_dofs_global() + g_in_block / dim);
      AssertThrow(it != node_of_phi_dof.end(), ExcMessage("matrix column outside the local nodes"));
      return it->second * dim + (int32_t)comp;
    }

    // Hanging-node closure (file header; cracks_amd/partition.py: hanging_closure_shipments is the tested statement of it):
    // every locally OWNED cell with a hanging vertex one of whose parents is owned by another rank is sent to that rank.
    // Record per cell: per vertex the phase-field dof, dim coordinates, the number of entries of its hanging line and the
    // (parent phase-field dof, weight) pairs -- as doubles (dof indices are exact up to 2^53).
    template <class Problem>
    void ship_hanging_closure_cells(const Problem &P, const std::vector<IndexSet> &owned_per_rank, const unsigned int me)
    {
      constexpr unsigned int nv = GeometryInfo<dim>::vertices_per_cell;
      const unsigned int phi_comp = dim;
      auto owner_of = [&](const gidx g) {
        for (unsigned int p = 0; p < owned_per_rank.size(); ++p)
          if (owned_per_rank[p].is_element(g))
            return (int)p;
        return -1;
      };
      std::map<unsigned int, std::vector<double>> outbox;
      for (const auto &cell : P.dof_handler.active_cell_iterators())
        if (cell->is_locally_owned())
          {
            std::set<unsigned int> to;
            for (unsigned int v = 0; v < nv; ++v)
              {
                const gidx g = cell->vertex_dof_index(v, phi_comp);
                if (!P.constraints_hanging_nodes.is_constrained(g))
                  continue;
                const auto *line = P.constraints_hanging_nodes.get_constraint_entries(g);
                if (line != nullptr)
                  for (const auto &e : *line)
                    {
                      const int p = owner_of(e.first);
                      if (p >= 0 && (unsigned int)p != me)
                        to.insert((unsigned int)p);
                    }
              }
            for (const unsigned int p : to)
              {
                std::vector<double> &box = outbox[p];
                for (unsigned int v = 0; v < nv; ++v)
                  {
                    const gidx g = cell->vertex_dof_index(v, phi_comp);
                    box.push_back((double)g);
                    for (unsigned int d = 0; d < dim; ++d)
                      box.push_back(cell->vertex(v)[d]);
                    const auto *line = P.constraints_hanging_nodes.is_constrained(g) ? P.constraints_hanging_nodes.get_constraint_entries(g) : nullptr;
                    box.push_back(line ? (double)line->size() : 0.0);
                    if (line)
                      for (const auto &e : *line)
                        {
                          box.push_back((double)e.first);
                          box.push_back(e.second);
                        }
                  }
              }
          }
      const std::map<unsigned int, std::vector<double>> inbox = Utilities::MPI::some_to_some(P.mpi_com, outbox);
      // cells already held (owned or ghost), by their sorted vertex dofs
      std::set<std::vector<gidx>> have;
      for (const auto &cell : P.dof_handler.active_cell_iterators())
        if (cell->is_locally_owned() || cell->is_ghost())
          {
            std::vector<gidx> key(nv);
            for (unsigned int v = 0; v < nv; ++v)
              key[v] = cell->vertex_dof_index(v, phi_comp);
            std::sort(key.begin(), key.end());
            have.insert(key);
          }
      shipped_cells.clear();
      shipped_lines.clear();
      for (const auto &kv : inbox)
        {
          const std::vector<double> &box = kv.second;
          size_t at = 0;
          while (at < box.size())
            {
              ShippedCell sc;
              std::vector<std::pair<gidx, std::vector<std::pair<gidx, double>>>> lines;
              for (unsigned int v = 0; v < nv; ++v)
                {
                  sc.phi_dof[v] = (gidx)box[at++];
                  for (unsigned int d = 0; d < dim; ++d)
                    sc.xyz[v][d] = box[at++];
                  const size_t n_ent = (size_t)box[at++];
                  std::vector<std::pair<gidx, double>> line;
                  for (size_t e = 0; e < n_ent; ++e, at += 2)
                    line.emplace_back((gidx)box[at], box[at + 1]);
                  if (n_ent)
                    lines.emplace_back(sc.phi_dof[v], line);
                }
              std::vector<gidx> key(sc.phi_dof, sc.phi_dof + nv);
              std::sort(key.begin(), key.end());
              if (!have.insert(key).second)
                continue; // in the ghost layer already, or shipped twice
              shipped_cells.push_back(sc);
              for (const auto &l : lines)
                if (!(relevant_dofs.is_element(l.first) && P.constraints_hanging_nodes.is_constrained(l.first)))
                  shipped_lines[l.first] = l.second;
            }
        }
    }

    // constraint flags (constraints_update minus the hanging lines) of the ghost nodes deal.II knows nothing about: one byte
    // per node from its owner, every assemble (the active set changes with every Newton step, cracks.cc:2826-2911)
    template <class Problem>
    void exchange_shipped_flags(const Problem &P)
    {
      if (!flags_shipped_anywhere)
        return; // the same decision on every rank (rebuild)
      std::map<unsigned int, std::vector<char>> out;
      for (const auto &kv : flags_asked_by)
        for (const gidx g : kv.second)
          out[kv.first].push_back((char)flags[(size_t)node_of_phi_dof.at(g)]);
      const std::map<unsigned int, std::vector<char>> in = Utilities::MPI::some_to_some(P.mpi_com, out);
      for (const auto &kv : flags_wanted_from)
        {
          const auto it = in.find(kv.first);
          AssertThrow(it != in.end() && it->second.size() == kv.second.size(), ExcMessage("flag exchange: answer missing"));
          for (size_t i = 0; i < kv.second.size(); ++i)
            flags[(size_t)node_of_phi_dof.at(kv.second[i])] = (uint8_t)it->second[i];
        }
    }

    // ---------------------------------------------------------------------------------------------------------
    // body of assemble_system(bool residual_only), cracks.cc:2129-2475
    // only_solution_changed: the line search (cracks.cc:2942-2957) and every Newton iteration after the first of a time
    // step call assemble with old_solution / old_old_solution untouched; pass true there and only `solution` is copied
    // and scattered again (pfm_state_set_solution: a third of the traffic)
    template <class Problem>
    void assemble(Problem &P, const bool residual_only, const bool only_solution_changed = false)
    {
      // scalars
      pfm_params p{};
      p.lambda = P.lame_coefficient_lambda;
      p.mu = P.lame_coefficient_mu;
      p.G_c = P.G_c;
      p.alpha_eps = P.alpha_eps;
      p.constant_k = P.constant_k;
      p.pressure = P.func_pressure.value(Point<1>(P.time), 0); // cracks.cc:2145
      p.alpha_biot = P.alpha_biot;
      if (P.outer_solver == Problem::OuterSolverType::simple_monolithic && P.timestep_number < 1)
        P.gamma_penal = 0.0; // the member side effect of cracks.cc:2141-2144
      p.gamma_penal = P.gamma_penal;
      p.timestep = P.timestep;
      p.time = P.time;
      p.old_timestep = P.old_timestep;
      p.old_old_timestep = P.old_old_timestep;
      p.decompose_stress_rhs = P.decompose_stress_rhs;
      p.decompose_stress_matrix = P.decompose_stress_matrix;
      p.timestep_number = (int32_t)P.timestep_number;
      p.outer_solver = P.outer_solver == Problem::OuterSolverType::active_set ? PFM_SOLVER_ACTIVE_SET : PFM_SOLVER_SIMPLE_MONOLITHIC;
      p.use_old_timestep_pf = P.use_old_timestep_pf ? 1 : 0;
      PFM_CALL(ctx, pfm_set_params(ctx, &p));

      // constraints_update minus the hanging-node lines: one flag byte per local node (cracks.cc:2442-2463)
      std::fill(flags.begin(), flags.end(), (uint8_t)0);
      for (int32_t n = 0; n < n_nodes; ++n)
        for (unsigned int comp = 0; comp <= (unsigned int)dim; ++comp)
          {
            const gidx g = global_dof_of(P, n, comp);
            // (vertices of shipped cells lie outside the locally relevant set: their bits arrive from the owner below)
            if (relevant_dofs.is_element(g) && P.constraints_update.is_constrained(g) && !P.constraints_hanging_nodes.is_constrained(g))
              flags[(size_t)n] |= (uint8_t)(1u << comp);
          }
      exchange_shipped_flags(P);
      PFM_CALL(ctx, pfm_set_constraints(ctx, flags.data()));

      // owned values of the three vectors in the context's layout (the Epetra storage of a block is contiguous and in
      // row-map order = ascending global index = the library's owned order; deal.II-knowledge)
      const TrilinosWrappers::MPI::BlockVector *vec[3] = {&P.solution, &P.old_solution, &P.old_old_solution};
      const size_t nd = (size_t)n_owned * (dim + 1);
      const bool sol_only = only_solution_changed && state_complete;
      for (int k = 0; k < (sol_only ? 1 : 3); ++k)
        {
          gather_owned(*vec[k], h_vec[k]);
          AssertThrow(hipMemcpy(d_vec[k], h_vec[k].data(), sizeof(double) * nd, hipMemcpyHostToDevice) == hipSuccess, ExcMessage("H2D"));
        }
      // ghost import (cracks.cc:2147-2154) + cell work; every rank calls the exchange (it is collective among peers)
      if (sol_only)
        PFM_CALL(ctx, pfm_state_set_solution(ctx, d_vec[0], /*on_device=*/1));
      else
        PFM_CALL(ctx, pfm_state_set(ctx, d_vec[0], d_vec[1], d_vec[2], /*on_device=*/1));
      state_complete = true;
      if (comm)
        PFM_CALL(ctx, pfm_halo_exchange(ctx, comm, peer_ranks.data()));
      PFM_CALL(ctx, pfm_assemble_device(ctx, residual_only ? 1 : 0, d_val, d_res[0], d_res[1]));
      const int rc = pfm_sync_status(ctx);
      if (rc == PFM_ERR_NOT_ORTHOGONAL)
        {
          std::cout << "Seems not to be orthogonal" << std::endl; // cracks.cc:1734-1735
          abort();
        }
      AssertThrow(rc == PFM_OK, ExcMessage(pfm_last_error(ctx)));

      // results into the Trilinos objects (owned rows are complete: no compress(add), cracks.cc:2470-2475)
      if (residual_only && !residual_to_host)
        return; // the caller reads residual_l2_norm(): the vectors stay on the device
      AssertThrow(hipMemcpy(h_res[0].data(), d_res[0], sizeof(double) * nd, hipMemcpyDeviceToHost) == hipSuccess, ExcMessage("D2H"));
      scatter_owned(h_res[0], P.system_pde_residual);
      if (residual_only)
        {
          AssertThrow(hipMemcpy(h_res[1].data(), d_res[1], sizeof(double) * nd, hipMemcpyDeviceToHost) == hipSuccess, ExcMessage("D2H"));
          scatter_owned(h_res[1], P.system_total_residual);
        }
      else
        {
          // pfm_values_to_host: blocks (u,u) and (phi,u), (phi,phi) on two streams; the (u,phi) block of a registered array
          // is identically zero, cleared once on the host and never transferred (3/16 of the bytes)
          double *h_val[4] = {nullptr, nullptr, nullptr, nullptr};
          const unsigned int nb1 = blocked ? 2 : 1;
          for (unsigned int r = 0; r < nb1; ++r)
            for (unsigned int c = 0; c < nb1; ++c)
              {
                int *rowptr = nullptr, *colind = nullptr;
                P.system_pde_matrix.block(r, c).trilinos_matrix().ExtractCrsDataPointers(rowptr, colind, h_val[blocked ? 2 * r + c : 0]);
              }
          PFM_CALL(ctx, pfm_values_to_host(ctx, d_val, h_val));
        }
      // the AMG set-up of cracks.cc:2477-2497 follows in the caller, unchanged
    }

    // constraints_update.set_zero(system_pde_residual); system_pde_residual.l2_norm()  (cracks.cc:2791-2794, 2947-2949) of the
    // last assemble(), from the device copy: the rank's sum of squares comes back (24 bytes), the ranks' sums are added
    // as l2_norm() adds them.  total = true: the same for system_total_residual.
    template <class Problem>
    double residual_l2_norm(const Problem &P, const bool total = false)
    {
      double out[3] = {0.0, 0.0, 0.0};
      PFM_CALL(ctx, pfm_residual_norms(ctx, d_res[total ? 1 : 0], out));
      return std::sqrt(Utilities::MPI::sum(out[2], P.mpi_com));
    }

    // global dof of (local node, component)
    // deal.II-knowledge (A) -- verify on a real install: loop over the cells, for every vertex v and component comp
    //   Assert(global_dof_of(P, node_of_phi_dof.at(cell->vertex_dof_index(v, dim)), comp) == cell->vertex_dof_index(v, comp)).
    // It holds if (i) the dim + 1 dofs of a vertex are numbered consecutively by DoFHandler::distribute_dofs for an
    // FESystem of dim + 1 Q1 elements, and (ii) DoFRenumbering::component_wise with the blocks {u..u, phi}
    // (cracks.cc:1587-1590) keeps the vertex order within each block.
    template <class Problem>
    gidx global_dof_of(const Problem &P, int32_t n, unsigned int comp) const
    {
      const gidx gphi = phi_dof_of_node[(size_t)n];
      if (!blocked)
        return gphi - dim + comp; // interleaved: the dim + 1 dofs of a vertex are consecutive, phi last
      if (comp == (unsigned int)dim)
        return gphi;
      return (gphi - P.n_u_dofs_global()) * dim + comp; // block 0: dim * vertex rank + comp
    }

    // owned dofs of a block vector -> the context's layout
    // deal.II-knowledge (C) -- verify on a real install: for i in [0, locally_owned_size)
    //   Assert(v.block(b).locally_owned_elements().nth_index_in_set(i) is ascending and *(v.block(b).begin() + i) == v.block(b)[that index])
    void gather_owned(const TrilinosWrappers::MPI::BlockVector &v, std::vector<double> &out) const
    {
      if (blocked)
        {
          const auto &u = v.block(0), &phi = v.block(1);
          AssertThrow((int64_t)u.locally_owned_size() == (int64_t)n_owned * dim && (int64_t)phi.locally_owned_size() == n_owned,
                      ExcMessage("owned sizes do not match the node count"));
          std::copy(u.begin(), u.end(), out.begin());                                // dim * node + comp
          std::copy(phi.begin(), phi.end(), out.begin() + (size_t)n_owned * dim);  // n_owned * dim + node
        }
      else
        {
          const auto &b = v.block(0);
          AssertThrow((int64_t)b.locally_owned_size() == (int64_t)n_owned * (dim + 1), ExcMessage("owned size does not match the node count"));
          std::copy(b.begin(), b.end(), out.begin()); // (dim + 1) * node + comp
        }
    }
    void scatter_owned(const std::vector<double> &in, TrilinosWrappers::MPI::BlockVector &v) const
    {
      if (blocked)
        {
          std::copy(in.begin(), in.begin() + (size_t)n_owned * dim, v.block(0).begin());
          std::copy(in.begin() + (size_t)n_owned * dim, in.end(), v.block(1).begin());
        }
      else
        std::copy(in.begin(), in.end(), v.block(0).begin());
    }
  };
#undef PFM_CALL
} // namespace pfm_glue_detail

// Two one-line helpers the glue asks of the problem class (add to FracturePhaseFieldProblem<dim>):
//   types::global_dof_index n_u_dofs_global() const { return dof_handler.n_dofs() / (dim + 1) * dim; }   // = n_solid of cracks.cc:1606 (Q1/Q1: dim + 1 dofs per vertex)
//   bool is_phase_field_dof(types::global_dof_index g) const
//   { return direct_solver ? (g % (dim + 1) == dim) : (g >= n_u_dofs_global()); }
//
// and the two call sites:
//   setup_system():            ... diag mass (cracks.cc:1675);  pfm_glue.rebuild(*this);
//   setup_system(), first line: pfm_glue.before_setup_system();   // required with pin_host_matrix, harmless without
//   assemble_system(bool ro):  pfm_glue.assemble(*this, ro);  if (!direct_solver && !ro) { AMG set-up, cracks.cc:2477-2497 }
//   line search (cracks.cc:2946-2949) with pfm_glue.residual_to_host = false:
//                              assemble_nl_residual();  new_newton_residual = pfm_glue.residual_l2_norm(*this);

#endif // PFM_WITH_DEALII
