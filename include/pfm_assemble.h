/* pfm_assemble.h — C ABI of the MI355X-native Newton residual/Jacobian assembly.
 *
 * Drop-in boundary for ONE path of tjhei/cracks:
 *   FracturePhaseFieldProblem<dim>::assemble_system(bool residual_only)   cracks.cc:2129-2498
 *   FracturePhaseFieldProblem<dim>::assemble_nl_residual()                cracks.cc:2507-2512
 * The reference has no plugin/FFI interface for it (assemble_system is a private member,
 * cracks.cc:1039-1040); this header defines the interface a deal.II glue shim binds
 * (INTEGRATION.md).  Semantics = cracks.cc:2133-2475 minus deal.II object handling:
 * zero the outputs, import ghost values, integrate every cell, scatter through the
 * constraints, reduce.  The AMG set-up at cracks.cc:2477-2497 stays with the caller.
 *
 * Plain C, POD only.  Every entry point returns a pfm_status (0 = ok) and never
 * aborts/exits/throws (the reference's abort() at cracks.cc:1735 becomes
 * PFM_ERR_NOT_ORTHOGONAL).  A context is not re-entrant; different contexts are
 * independent.  One context <-> one GPU <-> one reference MPI rank (cracks.cc:4587).
 *
 * Index spaces.  Nodes are numbered per rank: owned nodes [0,n_owned) first, ghost
 * nodes [n_owned,n_nodes) after (the "locally relevant" numbering of cracks.cc:1622-1628).
 * Local dof i of a cell <-> (vertex i/(dim+1), component i%(dim+1)); components
 * 0..dim-1 = displacement, dim = phase field (cracks.cc:980-996).
 *
 * Dof vectors (solution / residual) hold OWNED dofs only, in one of the reference's two
 * numberings (cracks.cc:1587-1590):
 *   PFM_LAYOUT_INTERLEAVED  one block,  dof = node*(dim+1)+comp      (direct solver)
 *   PFM_LAYOUT_BLOCKED      [u | phi],  u dof = node*dim+comp, phi dof = n_owned*dim+node
 * Matrix values are CSR, rows = owned dofs, columns in the rank-local numbering (ghost columns
 * included), ascending within a row unless pfm_pattern_bind() adopted another order from the host;
 * full component coupling (cracks.cc:1644-1654):
 *   INTERLEAVED: block 0 only;   BLOCKED: 0 = (u,u), 1 = (u,phi), 2 = (phi,u), 3 = (phi,phi)
 * The pattern is the node graph of the constraint-resolved mesh tensor the component
 * coupling; pfm_pattern_get() returns it, pfm_pattern_bind() adopts the host's own arrays.
 */
#ifndef PFM_ASSEMBLE_H
#define PFM_ASSEMBLE_H

#include <stdint.h>
#include "pfm_params.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pfm_status
{
  PFM_OK = 0,
  PFM_ERR_BAD_ARG = 1,
  PFM_ERR_HIP = 2,            /* a HIP runtime call failed; pfm_last_error() has the text */
  PFM_ERR_NOT_ORTHOGONAL = 3, /* eigenvector sanity check failed (reference abort(), cracks.cc:1732-1736) */
  PFM_ERR_NONFINITE = 4,      /* non-finite value in an output (checked by pfm_check_finite only) */
  PFM_ERR_UNSUPPORTED = 5,    /* e.g. stress split in 3-D (reference is 2-D only, cracks.cc:1685-1690) */
  PFM_ERR_NOMEM = 6,
  PFM_ERR_COMM = 7,           /* an RCCL call failed; pfm_last_error() has the text */
  PFM_ERR_INTERNAL = 8        /* an invariant of the library was violated (e.g. the deferred-patch list overflowed) */
} pfm_status;

enum
{
  PFM_LAYOUT_INTERLEAVED = 0,
  PFM_LAYOUT_BLOCKED = 1
};

/* per-dof constraint flag bits, one byte per NODE, bit c = component c */
enum
{
  PFM_NODE_ALLCOMP_MASK = 0x0f
};

typedef struct pfm_ctx pfm_ctx;

/* Static description of the rank-local mesh; everything is copied. Replaces the
 * DoFHandler / FEValues / Triangulation inputs of cracks.cc:2156-2203. */
typedef struct pfm_mesh_desc
{
  int32_t dim;               /* 2 or 3 */
  int32_t layout;            /* PFM_LAYOUT_* */
  int32_t n_nodes;           /* owned + ghost */
  int32_t n_owned_nodes;
  int64_t n_cells;           /* all local cells whose contributions reach an owned row:
                                owned cells + one ghost layer ("owner computes") */
  const int32_t *cell_nodes; /* [n_cells][2^dim], deal.II vertex order */
  const double *coords;      /* [n_nodes][dim] */
  const double *cell_lambda; /* [n_cells] or NULL: per-cell Lame override (cracks.cc:2207-2216) */
  const double *cell_mu;
  /* closed hanging-node table (constraints_hanging_nodes, cracks.cc:1630-1635), node level:
   * node hn_nodes[k] = sum_j hn_weights[j]*parent hn_parents[j], j in [hn_ptr[k],hn_ptr[k+1]) */
  int32_t n_hanging;
  const int32_t *hn_nodes;
  const int64_t *hn_ptr;
  const int32_t *hn_parents;
  const double *hn_weights;
  /* structured fast path hint: cells form an nx*ny(*nz) box in lexicographic order with
   * lexicographic node numbering (all zero = unknown; the library verifies the claim). */
  int32_t box_cells[3];
} pfm_mesh_desc;

/* -- life cycle ---------------------------------------------------------------------- */
/* Build a context on HIP device `device`.  Must be rebuilt after every setup_system()
 * (cracks.cc:4148, 4174). */
int pfm_ctx_create(pfm_ctx **out, const pfm_mesh_desc *mesh, int device);
int pfm_ctx_destroy(pfm_ctx *ctx);
const char *pfm_last_error(const pfm_ctx *ctx);
/* all launches/copies go to this hipStream_t (default: the null stream) */
int pfm_ctx_set_stream(pfm_ctx *ctx, void *hip_stream);

/* -- inputs set elsewhere but consumed by assemble_system -------- */
int pfm_set_params(pfm_ctx *ctx, const pfm_params *prm);
/* constraints_update minus the hanging nodes: one byte per local node, bit c set <=> dof
 * (node, c) has a homogeneous line (Dirichlet from set_newton_bc cracks.cc:2711-2714, or
 * active set cracks.cc:2878-2879).  Call whenever constraints_update is rebuilt. */
int pfm_set_constraints(pfm_ctx *ctx, const uint8_t *node_flags /* host, [n_nodes] */);

/* -- matrix pattern ------------------------------------------------------------------- */
int pfm_pattern_size(const pfm_ctx *ctx, int block, int64_t *n_rows, int64_t *nnz);
/* host outputs: rowptr[n_rows+1], colind[nnz]: the pattern the value arrays of pfm_assemble_device follow.  After
 * pfm_ctx_create the columns of every row ascend by local id (owned columns first, ghost columns last): the layout of
 * a host CSR sorted by local column index, e.g. an Epetra_CrsMatrix after FillComplete with optimised storage. */
int pfm_pattern_get(const pfm_ctx *ctx, int block, int64_t *rowptr, int32_t *colind);
/* Bind the HOST's pattern of `block` (make_sparsity_pattern + reinit, cracks.cc:1644-1654; the arrays
 * Epetra_CrsMatrix::ExtractCrsDataPointers returns): from now on the value arrays handed to the assembly are laid
 * out exactly like the host's values[] -- the library borrows the host's CSR, it does not dictate one.  The arrays are
 * read during the call only.  Requirements (checked, PFM_ERR_BAD_ARG otherwise): same rows and the same set of
 * columns per row as pfm_pattern_get (the node graph of the constraint-resolved mesh, full component coupling); within a
 * row the entries of one neighbour node adjacent with ascending component.  Any order of the neighbour nodes is
 * accepted, but all blocks must use the same one (PFM_ERR_UNSUPPORTED).  Binding a pattern that already has the
 * library's order is a pure check.  _i32: 32-bit row pointers (Epetra's int offsets). */
int pfm_pattern_bind(pfm_ctx *ctx, int block, const int64_t *rowptr, const int32_t *colind);
int pfm_pattern_bind_i32(pfm_ctx *ctx, int block, const int32_t *rowptr, const int32_t *colind);

/* -- state: the three vectors read at cracks.cc:2147-2154 ----------------------------- */
/* Scatter the OWNED dofs of solution / old_solution / old_old_solution (dof vectors in the
 * context's layout; host pointers if on_device == 0, device pointers otherwise) into the
 * context's node arrays.  Only the phase-field block of old / old_old is read
 * (cracks.cc:2229, 2232). */
int pfm_state_set(pfm_ctx *ctx, const double *sol, const double *old, const double *oldold,
                  int on_device);
/* The same for `solution` alone: old / old_old keep the values of the last pfm_state_set.  This is the call of the
 * line search (cracks.cc:2942-2957: solution += delta; assemble_nl_residual(), up to max_no_line_search_steps times per
 * Newton step while old_solution / old_old_solution do not change) and of every Newton iteration after the first of a
 * time step: a third of the scatter traffic of pfm_state_set.  Ghost values of solution still come from
 * pfm_halo_exchange. */
int pfm_state_set_solution(pfm_ctx *ctx, const double *sol, int on_device);
/* Ghost import (cracks.cc:2147-2154) as pack -> RCCL send/recv -> unpack; the exchange itself is
 * done by the host side between the two calls.  Registration copies the lists.
 * send_nodes: owned nodes whose values a peer needs; recv_nodes: ghost nodes a peer owns.
 * A packed node record is PFM_HALO_DOUBLES_PER_NODE(dim) doubles: u[dim], phi, phi_old, phi_oldold. */
#define PFM_HALO_DOUBLES_PER_NODE(dim) ((dim) + 3)
int pfm_halo_register(pfm_ctx *ctx, int n_peers, const int64_t *send_ptr, const int32_t *send_nodes,
                      const int64_t *recv_ptr, const int32_t *recv_nodes);
int pfm_halo_pack(pfm_ctx *ctx, int peer, double *d_buf);         /* device buffer */
int pfm_halo_unpack(pfm_ctx *ctx, int peer, const double *d_buf); /* device buffer */
/* All peers with one launch: d_buf_all holds the messages back to back in peer order, the message of peer k starts at
 * PFM_HALO_DOUBLES_PER_NODE(dim) * send_ptr[k] (pack) resp. * recv_ptr[k] (unpack) and has the same layout as the
 * per-peer calls produce. */
int pfm_halo_pack_all(pfm_ctx *ctx, double *d_buf_all);
int pfm_halo_unpack_all(pfm_ctx *ctx, const double *d_buf_all);

/* The whole ghost import inside the library, for hosts without torch (the deal.II application): RCCL point-to-point
 * over xGMI in place of the Trilinos/MPI import of cracks.cc:2147-2154.
 *   pfm_comm_unique_id   ncclGetUniqueId on one rank; the host broadcasts the bytes (MPI_Bcast in the reference's world)
 *   pfm_comm_create      ncclCommInitRank: collective over the n_ranks processes, one GPU each
 *   pfm_halo_exchange    pack (one launch) -> ncclGroupStart, ncclSend/ncclRecv per peer, ncclGroupEnd -> unpack (one
 *                        launch), all asynchronous on the context's stream, into buffers the context owns;
 *                        peer_ranks[k] = communicator rank of peer k of pfm_halo_register.  `comm` is a handle made
 *                        by pfm_comm_create, or by pfm_comm_wrap around the host's own ncclComm_t.  Collective: every
 *                        rank of the communicator that is somebody's peer must call it.
 *                        A raw ncclComm_t is NOT a handle: it is refused with PFM_ERR_BAD_ARG (the handles carry a tag).
 * Error path: a failed exchange on a communicator made by pfm_comm_create aborts it (ncclCommAbort: peers error out
 * instead of waiting for a message that never comes).  A communicator adopted with pfm_comm_wrap belongs to the host: it
 * is neither aborted nor destroyed by the library -- the handle is disabled and the host decides (abort or destroy its
 * ncclComm_t as usual).  Either way the HANDLE stays valid: pfm_comm_aborted() reports 1, every further exchange on it
 * returns PFM_ERR_COMM, and pfm_comm_destroy() only frees the handle then (no double free). */
#define PFM_COMM_ID_BYTES 128
int pfm_comm_unique_id(uint8_t id[PFM_COMM_ID_BYTES]);
int pfm_comm_create(void **comm, const uint8_t id[PFM_COMM_ID_BYTES], int n_ranks, int rank, int device);
int pfm_comm_wrap(void **comm, void *nccl_comm /* the host's ncclComm_t; not destroyed by pfm_comm_destroy */);
int pfm_comm_destroy(void *comm);
int pfm_comm_aborted(const void *comm); /* 1 after a failed exchange aborted the communicator, else 0 */
/* what RCCL itself reports for the communicator behind a handle: ncclCommCount, ncclCommUserRank (-1 where the loaded RCCL has
 * no such entry point) and ncclGetVersion -- diagnostics for multi-GPU records; any of the pointers may be NULL */
int pfm_comm_info(const void *comm, int *n_ranks, int *rank, int *rccl_version);
int pfm_halo_exchange(pfm_ctx *ctx, void *comm, const int *peer_ranks /* host, [n_peers] */);
/* pfm_halo_exchange + pfm_assemble_device with the ghost import HIDDEN behind cell work: after pfm_state_set the exchange
 * runs on a second stream of the context while the tiles that read no ghost node are assembled; the rest follows when the
 * import has landed.  Same results as the two calls in sequence (bit for bit on uniform boxes); same collective rule as
 * pfm_halo_exchange.  Declared here, next to the call it replaces; arguments as pfm_assemble_device. */
int pfm_assemble_overlapped(pfm_ctx *ctx, void *comm, const int *peer_ranks, int residual_only, double *const *d_values,
                            double *d_res_pde, double *d_res_tot);

/* -- the hot path --------------------------------------------------------------------- */
/* assemble_system(residual_only) on the current state.  Output pointers are DEVICE
 * pointers; the call is asynchronous on the context's stream.
 *   residual_only != 0: residual_pde (through constraints_update) and residual_total
 *     (through constraints_hanging_nodes for the active-set solver, constraints_update
 *     otherwise; cracks.cc:2440-2456) are zeroed and assembled; values is ignored.
 *   residual_only == 0: values[block] and residual_pde are zeroed and assembled
 *     (cracks.cc:2457-2464); residual_total is ignored.
 * Rows of ghost nodes are never written: each rank computes its owned rows completely
 * (it also integrates its ghost-layer cells), so no reverse exchange (compress(add),
 * cracks.cc:2470-2475) is needed. */
int pfm_assemble_device(pfm_ctx *ctx, int residual_only, double *const *d_values /* [n_blocks] */,
                        double *d_residual_pde, double *d_residual_total);
/* assemble_nl_residual() after `solution` alone changed -- the call of the line search (cracks.cc:2942-2957: solution +=
 * delta; assemble_nl_residual(), up to max_no_line_search_steps times per Newton step; cracks.cc:2507-2512):
 * = pfm_state_set_solution(d_solution, on_device = 1) + pfm_assemble_device(residual_only = 1) in one call.  On a
 * single-rank 3-D box the residual kernel reads d_solution itself and the scatter launch disappears.  Ranks with halo
 * peers import ghosts between the two steps and keep calling them separately (PFM_ERR_BAD_ARG here). */
int pfm_assemble_nl_residual_device(pfm_ctx *ctx, const double *d_solution, double *d_residual_pde, double *d_residual_total);
/* Blocks until the stream is idle and returns the deferred status of the launches since
 * the last call (PFM_ERR_NOT_ORTHOGONAL, PFM_ERR_HIP, ...). */
int pfm_sync_status(pfm_ctx *ctx);

/* On request: scan n doubles of a device array (an assembled residual or value block) for NaN / Inf.  Returns
 * PFM_ERR_NONFINITE if there is one, else whatever pfm_sync_status() would return.  The assembly itself never checks:
 * like the reference it lets IEEE specials propagate (e.g. the stress split at diagonal or zero strain,
 * cracks.cc:1982-2006). */
int pfm_check_finite(pfm_ctx *ctx, const double *d_data, int64_t n);

/* Host-visible outputs (cracks.cc:2754, 2770, 2918: the caller hands system_pde_matrix to Trilinos right after the call).
 *   pfm_host_register    page-locks a host array the host-pointer entry points read or write at every call -- the value
 *                        arrays of the Epetra_CrsMatrix blocks (Epetra_CrsMatrix::ExtractCrsDataPointers), the owned parts of
 *                        the vectors -- so that their transfers run as DMA at the link rate instead of through the runtime's
 *                        pageable path.  The array stays the caller's; it must be unregistered (pfm_host_unregister, NULL =
 *                        all; pfm_ctx_destroy does it too) BEFORE it is freed or reallocated (setup_system after refine_mesh).
 *                        PFM_ERR_HIP if the pages cannot be locked (ulimit): the array simply stays pageable.
 *   pfm_values_to_host   matrix values of the last pfm_assemble_device -> h_values[block], synchronous.  With the 2x2 block
 *                        layout the (u,phi) block is identically zero (trial phase-field dofs do not enter the displacement
 *                        rows, cracks.cc:2333-2337; the placeholders of constrained rows sit on the diagonals of (u,u) and
 *                        (phi,phi)): 3/16 of the bytes.  If h_values[1] is a REGISTERED array it is cleared once by host
 *                        threads and never transferred -- the library assumes nobody else writes the matrix values between
 *                        assemblies (the reference only fills them through assemble_system); an unregistered array is copied
 *                        like the other blocks.  Blocks 2, 3 travel on a second stream next to block 0.
 * pfm_assemble = pfm_state_set + pfm_assemble_device + these copies + pfm_sync_status: the exact call shape of the
 * reference (outputs complete in host memory on return, cracks.cc:2791-2794, 2918).  Single-rank only (no ghosts). */
int pfm_host_register(pfm_ctx *ctx, void *host_array, int64_t bytes);
int pfm_host_unregister(pfm_ctx *ctx, void *host_array /* NULL: every array of this context */);
int pfm_values_to_host(pfm_ctx *ctx, double *const *d_values /* [n_blocks] */, double *const *h_values /* [n_blocks] */);
int pfm_assemble(pfm_ctx *ctx, const double *sol, const double *old, const double *oldold,
                 int residual_only, double *const *values, double *residual_pde,
                 double *residual_total);

/* -- measurement ---------------------------------------------------------------------- */
/* When enabled, every pfm_assemble_device() brackets its kernel group (output zeroing where
 * the kernel family needs it, residual and Jacobian kernels; not the state scatter, which is
 * pfm_state_set_device) with HIP events on the context's stream;
 * pfm_kernel_time_ms() synchronises, returns the mean duration of the launches recorded
 * since the last call and resets the record. */
int pfm_timing_enable(pfm_ctx *ctx, int on);
int pfm_kernel_time_ms(pfm_ctx *ctx, double *mean_ms, int *n_launches);
/* The individual durations (for a median): the first min(capacity, recorded) ones; *n_launches = recorded.  Does not
 * reset the record (pfm_kernel_time_ms / pfm_timing_enable do). */
int pfm_kernel_times_ms(pfm_ctx *ctx, double *ms, int capacity, int *n_launches);

/* -- introspection -------------------------------------------------------------------- */
/* which kernel family the context selected: 0 = general (any Q1 mesh), 1 = cartesian (uniform box; its 2-D stress-split runs
 * use the overlay below), 3 = general family + cartesian overlay: 2-D meshes with hanging nodes, slits, several refinement
 * levels -- the rows of regular lattice nodes are completed by one workgroup per 8 x 8 block of their level, the general
 * family keeps the cells that touch any other row.  pfm_ctx_force_path(0) selects the general family alone (A/B runs). */
int pfm_ctx_kernel_path(const pfm_ctx *ctx);
int pfm_ctx_force_path(pfm_ctx *ctx, int path);
/* composition of the overlay: rows written by the patch kernel / cells left to the general family (0 / all without it) */
int pfm_ctx_overlay_info(const pfm_ctx *ctx, int64_t *n_patch_rows, int64_t *n_general_cells);
/* measurement only: 1 / 2 make pfm_assemble_device run only the first / second half of pfm_assemble_overlapped (the work
 * that reads no ghost node / the rest), 0 restores the whole assembly: how much work hides the ghost import */
int pfm_ctx_force_phase(pfm_ctx *ctx, int phase);
int64_t pfm_ctx_device_bytes(const pfm_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
