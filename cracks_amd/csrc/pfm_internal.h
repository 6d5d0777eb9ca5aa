// pfm_internal.h — context layout shared by the host side (pfm_host.cpp) and the
// kernel launchers (pfm_kernels.hip).  Not part of the ABI.
#pragma once

#include <hip/hip_runtime_api.h>
#include <cstdint>
#include <string>
#include <vector>
#include <memory>
#include <type_traits>
#include <utility>

#include "../../include/pfm_assemble.h"

namespace pfm
{
  // Device-side view of the static mesh tables and the node state (SoA, HBM resident).
  constexpr int PFM_CRES_SLOT = 64, PFM_CRES_R = 320, PFM_CRES_HV = 336, PFM_CRES_CELL = 368, PFM_CRES_BYTES = 376;
  struct HgEntry // one contribution to a row of the gather tables (pfm_ctx::d_hg_list)
  {
    int32_t code; // hc * 32 + index: index < 16 = resolved node i of cell hc, 16 + a = the row of the cell's hanging vertex a
    int32_t R;    // resolved nodes of the cell
    long long koff; // first double of row i of the cell's K' in DevView::hs_K
  };
  constexpr int PFM_HS_RD = 96; // doubles per cell of DevView::hs_RD: residual of 16 resolved nodes x 4, placeholder of 8 vertices x 4
  struct DevView
  {
    int dim, layout, n_nodes, n_owned;
    long long n_cells;
    const int32_t *conn;   // [nv][n_cells]     cell -> local node, vertex-major (coalesced per vertex)
    const double *coords;  // [dim][n_nodes]
    const double *cell_lambda, *cell_mu; // [n_cells] or nullptr
    const long long *nadj_ptr; // [n_owned+1]  node graph (rows = owned nodes, sorted columns)
    const int32_t *nadj;
    const uint8_t *cslot;      // [n_cells][nv*nv] slot of vertex b's node in the row of vertex a's node
    const int32_t *color_cells; // [n_cells] cell ids sorted by colour class (pfm_ctx::color_ptr): no two cells of a class
                                // share a node, so the general cell kernel adds into the rows without atomics
    const int32_t *hn_index;   // [n_nodes] -> k (hanging table) or -1; nullptr when the mesh is conforming
    const long long *hn_ptr;
    const int32_t *hn_parents;
    const double *hn_weights;
    // cells with a hanging vertex: hcell[cell] = their running number (-1: none hangs), cslot_h[number][a][r][b][s] = slot of
    // parent s of vertex b in the row of parent r of vertex a (r, s < 4 in 3-D, < 2 in 2-D; a vertex that does not hang is
    // its own parent 0; 0xff: no such parent, row not owned, or not in the row).  nullptr: no hanging nodes, or one with more
    // parents than that (the cell kernel then searches the row, find_slot)
    const int32_t *hcell;
    const uint8_t *cslot_h;
    // 3-D: per such cell a record of PFM_CRES_BYTES: int32 node[16] (its distinct constraint-resolved nodes, -1 unused),
    // uint8 slot[16][16] at PFM_CRES_SLOT (slot of node j in the row of node i), uint8 R at PFM_CRES_R (0xff: more than 16).
    // The cell kernel forms C^T K C over these nodes before it adds (k_assemble_general, KRED).  nullptr: no such cells.
    const uint8_t *cres;         // (round 6: + int32 hv[8] at PFM_CRES_HV, the cell's hanging vertices' nodes (-1: does not hang),
                                 // and the cell id at PFM_CRES_CELL)
    // Round 6, the default on 3-D meshes with hanging nodes: the cells at hanging vertices do not add into the outputs at all.
    // Their kernel writes K' = C^T K C, R' = C^T R and the placeholder diagonals into per-cell scratch (hs_K at hs_off[hc]:
    // R x R x 13 doubles; hs_RD: PFM_HS_RD doubles per cell), and k_hanging_gather adds them row by row in a fixed order
    // (pfm_ctx::d_hg_*): no atomics, no colour classes, bitwise reproducible.  nullptr: the atomic class of round 5.
    double *hs_K;
    const long long *hs_off;
    double *hs_RD;
    const uint8_t *node_flags; // [n_nodes] bit c: dof (node,c) has a homogeneous constraint line
    const uint8_t *cell_ring;  // [n_cells] or nullptr: 1 = a cell of a plain colour class that shares a (constraint-resolved) node
                               // with a cell of the atomic class: it adds atomically too, so that the atomic class may run NEXT
                               // TO the colour classes (general family, pfm_kernels.hip)
    // node state (cracks.cc:2147-2154 after the ghost import)
    double *u[3];
    double *phi, *phi_old, *phi_oldold;
    int *status; // device error word (pfm_status)
    // pfm_assemble_nl_residual_device on a single-rank 3-D box: the residual kernel reads `solution` (owned dofs, the
    // context's layout) itself while it fills its nodal planes and writes the node state on the way -- the separate scatter
    // launch of pfm_state_set_solution (0.1 ms at 216^3) disappears.  nullptr everywhere else.
    const double *fused_solution;
    // ---- cartesian overlay of a general 2-D mesh (round 4, pfm_kernels.hip: PATCH): rows of "regular" nodes -- not hanging,
    // not a parent, exactly four incident cells of one refinement level, a plain 9-neighbour row -- are completed inside one
    // workgroup per 8 x 8 block of that level's lattice and written once; the colour classes / the atomic class only run
    // over the cells that touch another row and skip the regular ones.  nullptr: no overlay.
    const uint8_t *row_patch;            // [n_nodes] 1 = the row is written by the patch kernel
    const int32_t *patch_cells;          // [n_patch_blocks][64] cell id of lattice position (cx, cy) of the block, -1 = none
    const int32_t *patch_nodes;          // [n_patch_blocks][81] node id of the block's 9 x 9 nodes, -1 = none
    const unsigned long long *node_slots; // [n_owned] regular rows: CSR slot of lattice offset o in bits 4 o .. 4 o + 3
  };

  // Uniform Cartesian box (fast path): lattice of (NX,NY,NZ) nodes, owned nodes form the
  // sub-box [o0,o1] (inclusive, lattice coordinates); rows are addressed by local node id.
  struct CartView
  {
    int NX, NY, NZ;
    int o0[3], o1[3];
    double h[3];
    const int32_t *local_of_box; // [NX*NY*NZ] lattice index -> local node id
    const uint32_t *nbr_mask;    // [n_owned] bit o (o < 27): lattice offset o exists in the row.  Bit 31 clear: the row
                                 // is in lattice order, the CSR slot of offset o is popcount(mask & ((1 << o) - 1))
    int owned_lex;               // 1: owned node (i,j,k) has local id (i-o0x) + OWX*((j-o0y) + OWY*(k-o0z))
    const uint8_t *row_perm;     // rows whose order is not the lattice order (bit 31 of nbr_mask: e.g. ghost columns
                                 // sorted behind the owned ones): CSR slot of the r-th existing offset =
                                 // row_perm[nadj_ptr[row] + r]; nullptr when no row needs it
    const double *cell_lam, *cell_mu; // per-cell Lame coefficients (cracks.cc:2207-2216) in lattice cell order
    double *cell_avg;                 // 2-D boxes: [(NX-1)*(NY-1)] scratch, mean |diagonal| of the element matrices of the blocks with
                                      // constrained rows -- from the first launch of k_cart2d_cells to the second
                                      // ci + (NX-1) (cj + (NY-1) ck); nullptr: the scalars of pfm_params
    // pfm_assemble_overlapped, phase 2: compact launches over the tiles that read ghost nodes (lists built on first use:
    // an early-exit launch over the full grid costs ~3 ns per skipped workgroup, 0.4 ms at 1.5e5 tiles)
    const int32_t *bnd_uu3, *bnd_res3; // tile indices of k_cart_uu3 / k_cart_residual3 (the latter for its z-chunk length)
    int n_bnd_uu3, n_bnd_res3, zc_res3;
    // Deferred (u,u) placeholder patches of k_cart_phi4 (constrained displacement rows whose element diagonal vanished in
    // some cell, deal.II's mean-|diagonal| rule): when the two Jacobian kernels run next to each other the phase-field kernel
    // must not add to values the (u,u) kernel may still be writing -- it appends (index, value) here and
    // launch_cart_apply_patches adds them after the join.  nullptr: patched in place (the kernels run one after the other).
    long long *patch_idx;
    double *patch_val;
    int *patch_count;
    int patch_cap;
    // Cartesian overlay of a general 3-D mesh (round 5): the lattice is ONE refinement level's, local_of_box holds -1
    // where the level has no node, and the rows this launch writes are those of row_of_box (-1: not a row of this level's
    // launch -- hanging, parent, mixed-level or ghost nodes stay with the general family).  nullptr: every owned node of
    // the box is a row (uniform boxes).
    const int32_t *row_of_box;
    int up_block_cleared;             // 2-D boxes, blocked layout: the caller has cleared the structurally zero (u,phi) block with a fill
    int tile_sel;                     // 0: every tile; 1: only tiles that read no ghost node ("interior"); 2: only the
                                      // others -- the two launches of pfm_assemble_overlapped, between which the ghost
                                      // import lands (cracks.cc:2147-2154 next to the cell loop instead of in front of it)
  };

  // does the node range [lo, hi] (tile + one-node halo, lattice indices along `axis`) contain a ghost node?  Ghost nodes
  // are the lattice nodes outside the owned box [o0, o1].
  __host__ __device__ inline bool cart_range_has_ghost(const CartView &cv, int axis, int lo, int hi)
  {
    const int n = axis == 0 ? cv.NX : (axis == 1 ? cv.NY : cv.NZ);
    return (cv.o0[axis] > 0 && lo < cv.o0[axis]) || (cv.o1[axis] < n - 1 && hi > cv.o1[axis]);
  }
  // tile filter of the overlapped assembly: true = this launch skips the tile
  __host__ __device__ inline bool cart_tile_skipped(const CartView &cv, bool touches_ghost)
  {
    return cv.tile_sel != 0 && touches_ghost != (cv.tile_sel == 2);
  }

  // host copy of the lattice tables of a uniform box (kept for pfm_pattern_bind)
  // std::vector whose resize(n) / vector(n) leaves trivially constructible elements uninitialised: tables of 1e7 entries that
  // are written once by the host threads are not cleared (and their pages not touched) by one thread first
  template <class T>
  struct default_init_allocator : std::allocator<T>
  {
    template <class U>
    struct rebind
    {
      using other = default_init_allocator<U>;
    };
    using std::allocator<T>::allocator;
    template <class U>
    void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value)
    {
      ::new (static_cast<void *>(p)) U;
    }
    template <class U, class... A>
    void construct(U *p, A &&...a)
    {
      ::new (static_cast<void *>(p)) U(std::forward<A>(a)...);
    }
  };
  template <class T>
  using raw_vector = std::vector<T, default_init_allocator<T>>;

  struct LatticeHost
  {
    int NX = 0, NY = 0, NZ = 0, nc[3] = {0, 0, 0};
    double h[3] = {1, 1, 1};
    raw_vector<int32_t> local_of_box, box_of_local;
  };

  struct HaloPeer
  {
    int32_t *d_send = nullptr, *d_recv = nullptr;
    int64_t n_send = 0, n_recv = 0;
  };

  // launchers implemented in pfm_kernels.hip
  int launch_state_set(const DevView &v, const double *d_sol, const double *d_old,
                       const double *d_oldold, hipStream_t s);
  // fills v.cslot from the current order of the node-graph rows (context creation, pfm_pattern_bind)
  int launch_build_cslot(const DevView &v, hipStream_t s);
  // d_out[a * n + i] = d_in[i * w + a] (mesh tables: host AoS -> device SoA)
  int launch_aos_to_soa_i32(const int32_t *d_in, int32_t *d_out, long long n, int w, hipStream_t s);
  int launch_aos_to_soa_f64(const double *d_in, double *d_out, long long n, int w, hipStream_t s);
  int launch_check_finite(const DevView &v, const double *d, int64_t n, hipStream_t s);
  // neighbour masks of a full lexicographic lattice: bit o = lattice offset o lies inside the box (CartView::nbr_mask)
  int launch_lattice_masks(uint32_t *d_mask, int NX, int NY, int NZ, int dim, hipStream_t s);
  int launch_halo_pack(const DevView &v, const int32_t *d_nodes, int64_t n, double *d_buf, hipStream_t s);
  // all peers at once: d_nodes = concatenated lists, d_ptr[n_peers + 1] = their offsets (device)
  int launch_halo_all(const DevView &v, const int32_t *d_nodes, const long long *d_ptr, int n_peers, int64_t n_total,
                      double *d_buf, int unpack, hipStream_t s);
  int launch_halo_unpack(const DevView &v, const int32_t *d_nodes, int64_t n, const double *d_buf,
                         hipStream_t s);
  // d_scal: the context's device buffer (PFM_SCAL_BYTES) for per-launch scalar tables.  Kernels read them through
  // a pointer instead of by-value kernel arguments: a 900-byte argument struct pins so many SGPRs that the
  // compiler spills them into VGPR lanes (12 % of the instructions of the phase-field kernel were such moves).
  int launch_assemble_cart(const DevView &v, const CartView &cv, const pfm_params &p, int residual_only,
                           double *const *d_values, double *d_res_pde, double *d_res_tot, hipStream_t s,
                           hipStream_t s_residual, void *d_scal, int phase = 0);
  // true: this assembly is the pair k_cart_uu3<RES> + k_cart_phi4<RES> and may run them on two streams (s, s_residual of
  // launch_assemble_cart) -- the caller forks / joins and applies the deferred patches (CartView::patch_*)
  bool cart_jacobian_pair(const DevView &v, const CartView &cv, const pfm_params &p, int residual_only, int phase);
  int launch_cart_apply_patches(const CartView &cv, double *vals_uu, hipStream_t s, int *status);
  bool cart_matrix_supported(int dim);
  // 2-D boxes: row-owner Jacobian + residual of runs WITHOUT the stress split (pfm_cart2d.hip; PFM_ERR_UNSUPPORTED otherwise)
  int launch_cart2d(const DevView &v, const CartView &cv, const pfm_params &p, int residual_only, double *const *d_values,
                    double *res_pde, double *res_tot, hipStream_t s, hipStream_t s_phi);
  // z-chunk length of a marching kernel: `tiles` columns, `planes` node planes, one redundant cell layer per chunk,
  // `per_cu` resident workgroups per CU.  Maximises (fill of the last dispatch round) x (useful layers per chunk).
  int choose_zchunk(long long tiles, int planes, int zc_min, int zc_max, int per_cu);
  // XCD-aware launch: workgroup i runs on XCD i % 8.  The kernels are launched with a grid rounded up to a multiple
  // of 8 and map blockIdx to (blockIdx % 8) * (grid / 8) + blockIdx / 8, so that every XCD works on one contiguous
  // range of tiles (neighbouring tiles share their halo in that XCD's L2) and the slow boundary tiles are spread over
  // all XCDs — with the plain mapping a tile count per row that is a multiple of 8 (e.g. the 109-node edge of an
  // 8-rank sub-box) put every boundary tile on the same two XCDs (+30 % kernel time).
  constexpr unsigned xcd_grid(unsigned n_tiles) { return ((n_tiles + 7u) / 8u) * 8u; }
  constexpr size_t PFM_SCAL_BYTES = 4096;
  // the Jacobian launchers expect the MatScal of the current parameters at d_scal (uploaded by pfm_assemble_device
  // after every pfm_set_params, pfm_ctx::scal_dirty)
  int upload_mat_scal(const pfm_params &p, const CartView &cv, void *d_scal, hipStream_t s);
  int launch_cart_phi4(const DevView &v, const CartView &cv, const pfm_params &p, double *const *d_values, hipStream_t s,
                       const void *d_scal, double *res_pde);
  // res_pde != nullptr: the kernel also writes the displacement rows of the residual (from its matrix rows, see the kernel)
  int launch_cart_uu3(const DevView &v, const CartView &cv, const pfm_params &p, double *vals_uu, hipStream_t s,
                      const void *d_scal, double *res_pde, int lds_total = 0);
  // node graph of a general mesh on the device (pfm_graph.hip)
  struct GraphScratch
  {
    int *inc_ptr = nullptr, *inc = nullptr; // incidence lists: cells around every owned node
  };
  int graph_build_begin(const int32_t *d_cells, long long NC, int nv, int32_t NO, const int32_t *d_hn_index, const long long *d_hn_ptr,
                        const int32_t *d_hn_parents, long long *d_nadj_ptr, GraphScratch &sc, long long &total, hipStream_t s);
  int graph_build_rows(const int32_t *d_cells, long long NC, int nv, int32_t NO, const int32_t *d_hn_index, const long long *d_hn_ptr,
                       const int32_t *d_hn_parents, const long long *d_nadj_ptr, int32_t *d_nadj, const GraphScratch &sc, hipStream_t s);
  void graph_build_free(GraphScratch &sc);
  // row tables (CartView::nbr_mask, row_perm) of one level lattice of the 3-D cartesian overlay, from the current row order
  // Device scratch of the context build (raw mesh tables before their transposition, incidence lists and scan space of the
  // node graph): buffers up to 32 MB are kept between builds (64 MB at most) -- a hipMalloc / hipFree pair costs 0.1-0.4 ms and
  // the free synchronises the device, a rebuild at 2.7e5 cells made eight of them.  Users run on the NULL stream, so a buffer
  // handed back while its last kernel is still queued is safe to hand out again.  Contents undefined.
  hipError_t scratch_acquire(void **p, size_t bytes);
  void scratch_release(void *p);
  int launch_lattice_row_ptr(long long *d_ptr, int NX, int NY, int NZ, hipStream_t s);
  int launch_lattice_colour_order(int32_t *d_order, const int32_t *d_vertex0, const int32_t *d_box_of_local, long long NC, int NX, int NY, hipStream_t s);
  int launch_check_row_lengths(const uint8_t *d_mark, const long long *d_ptr, int32_t NO, int want, int *d_bad, hipStream_t s);
  int launch_overlay3_rows(const int32_t *d_node_at, const int32_t *d_row_at, int NX, int NY, int NZ, const long long *d_nadj_ptr,
                           const int32_t *d_nadj, uint32_t *d_nbr_mask, uint8_t *d_row_perm, int *d_bad, hipStream_t s);
  // host: indices of the tiles of k_cart_uu3 / k_cart_residual3 that read a ghost node (pfm_assemble_overlapped, phase 2)
  void cart_uu3_boundary_tiles(const CartView &cv, std::vector<int32_t> &out);
  void cart_res3_boundary_tiles(const CartView &cv, std::vector<int32_t> &out, int &zc);
  int launch_cart_uu_only(const DevView &v, const CartView &cv, const pfm_params &p, double *vals_uu, hipStream_t s,
                          void *d_scal);
  // color_ptr[n_classes + 1]: ranges of DevView::color_cells, one launch per class; the LAST class holds the cells with
  // hanging vertices (their rows are distributed to the parents, which other vertices of the same cell may be: atomics)
  int launch_assemble_general(const DevView &v, const pfm_params &p, int residual_only,
                              double *const *d_values, double *d_res_pde, double *d_res_tot,
                              hipStream_t s, const std::vector<long long> &color_ptr, hipStream_t s_atomic = nullptr);
  // the patch kernel of the cartesian overlay (2-D): one workgroup per block of DevView::patch_cells
  int launch_assemble_patches(const DevView &v, const pfm_params &p, int residual_only, double *const *d_values, double *d_res_pde,
                              double *d_res_tot, int n_blocks, hipStream_t s);
  // fills DevView::node_slots from the current order of the node-graph rows (context creation, pfm_pattern_bind)
  // rows [0, n_rows) of the gather tables: one wave per destination row, its entries in list order (fixed): plain adds
  int launch_hanging_gather(const DevView &v, const pfm_params &p, int residual_only, double *const *d_values, double *d_res_pde, double *d_res_tot,
                            const int32_t *rows, const long long *ptr, const HgEntry *list, int64_t n_rows, hipStream_t s);
  int launch_zero_rows(const DevView &v, double *const *d_values, const int32_t *rows, int n_rows, hipStream_t s);
  int launch_patch_slots(const DevView &v, unsigned long long *d_slots, int n_blocks, hipStream_t s);
} // namespace pfm

struct pfm_ctx
{
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<hipStream_t> ov_streams; // 3-D overlay: one stream per level lattice (forked off `stream` in an assembly)
  std::vector<hipEvent_t> ov_events;
  hipEvent_t ov_fork = nullptr;
  hipStream_t atomic_stream = nullptr; // 3-D overlay: the general family's atomic class next to its plain classes
  hipEvent_t ev_atomic = nullptr;
  hipStream_t side_stream = nullptr;            // residual + clearing of the (u,phi) block, concurrent with the Jacobian
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  pfm::DevView v{};
  pfm_params prm{};
  bool have_params = false;
  int n_blocks = 1;
  int kernel_path = 0;
  int force_phase = 0; // pfm_ctx_force_phase (measurement)
  bool cart_ok = false;
  pfm::CartView cv{};
  // host copies needed for pattern queries
  pfm::raw_vector<long long> h_nadj_ptr;
  long long nadj_total = -1; // entries of the node graph when its row pointers only exist on the device so far (graph_dev_only)
  std::vector<int32_t> h_nadj;
  // Lattice meshes: the 27-wide host node graph (h_nadj), its device copy (v.nadj) and the slot table of the general
  // cell kernel (v.cslot) are 1.1 + 1.1 + 0.65 GB at 1e7 cells and are only needed by pfm_pattern_get / _bind and by the
  // general family: built on first use (ensure_host_graph, ensure_general_tables in pfm_host.cpp)
  bool graph_positional = false; // uniform box, node n at lattice position n, no ghosts: row pointers and colour lists are made on the device
  bool hanging_coloured = false; // 3-D: cells at hanging vertices sit in plain colour classes (reduced scatter, no atomics)
  int32_t n_hanging = 0;
  bool full_colours_lazy = false; // overlay context: the colour lists over ALL cells are made when the general family first runs without the overlay
  bool colours_lazy = false;     // DevView::color_cells is filled when the general family is first used (ensure_general_tables)
  bool graph_lazy = false;    // h_nadj not materialised yet (h_nadj_ptr is)
  bool graph_dev_only = false; // general mesh: v.nadj was built on the device (pfm_graph.hip), h_nadj is fetched on demand
  bool general_ready = true;  // v.nadj and v.cslot exist on the device
  std::vector<long long> color_ptr; // colour classes of the general cell kernel (DevView::color_cells)
  // owned device allocations
  std::vector<void *> allocs;
  int64_t device_bytes = 0;
  // staging for host-pointer entry points
  double *d_stage_vec[3] = {nullptr, nullptr, nullptr};
  double *d_stage_res[2] = {nullptr, nullptr};
  double *d_stage_val[4] = {nullptr, nullptr, nullptr, nullptr};
  // host arrays the caller page-locked through pfm_host_register (DMA at the link rate instead of the pageable path);
  // zeroed: the array is the (u,phi) block of a matrix and has been cleared once (pfm_values_to_host never copies it)
  struct HostPin
  {
    void *p = nullptr;
    size_t bytes = 0;
    bool zeroed = false;
  };
  std::vector<HostPin> host_pins;
  hipStream_t copy_stream = nullptr; // second device -> host stream of pfm_values_to_host
  hipEvent_t ev_copy = nullptr;
  std::vector<pfm::HaloPeer> peers;
  int32_t *d_send_all = nullptr, *d_recv_all = nullptr; // concatenated halo lists and their per-peer offsets
  long long *d_send_ptr = nullptr, *d_recv_ptr = nullptr;
  int64_t n_send_all = 0, n_recv_all = 0;
  double *d_halo_send = nullptr, *d_halo_recv = nullptr; // message buffers of pfm_halo_exchange
  int64_t halo_buf_bytes = 0;                            // their share of device_bytes
  void *d_scal = nullptr; // per-launch scalar tables of the cartesian kernels (PFM_SCAL_BYTES)
  int64_t n_flag_u = 0;   // displacement dofs with a constraint flag (capacity of the deferred patch list, CartView::patch_*)
  uint8_t *d_row_perm = nullptr; // CartView::row_perm storage (in allocs)
  bool overlap_lists_ready = false; // CartView::bnd_uu3 / bnd_res3 built
  pfm::LatticeHost lat;          // host lattice tables (cartesian path only)
  bool pattern_bound[4] = {false, false, false, false};
  bool scal_dirty = true; // d_scal does not hold the tables of the current parameters yet
  // cartesian overlay of a general 2-D mesh (DevView::row_patch ...): blocks, the reduced cell lists of the general family
  int n_patch_blocks = 0;
  int64_t n_patch_rows = 0, n_general_cells = 0;
  unsigned long long *d_node_slots = nullptr;
  bool patch_slots_valid = false;
  // cartesian overlay of a general 3-D mesh (round 5): one level lattice per refinement level with regular rows; the
  // row-owner kernels of the cartesian family (k_cart_uu3, k_cart_phi4, k_cart_residual3) run on each of them and write the
  // regular rows, the general family keeps the rest (same reduced lists as the 2-D overlay)
  struct OverlayLevel
  {
    pfm::CartView cv{};
    void *d_scal = nullptr; // the level's MatScal (its cell size)
    bool scal_dirty = true;
    int64_t n_rows = 0;
  };
  std::vector<OverlayLevel> levels3;
  uint32_t *d_nbr_mask3 = nullptr;
  uint8_t *d_row_perm3 = nullptr;
  bool overlay3_rows_valid = false;
  int32_t *d_rows_general = nullptr;        // owned nodes whose rows the general family writes in an overlay assembly
  int32_t n_rows_general = 0;
  int32_t *d_color_cells_reduced = nullptr; // colour-sorted cells that touch a row the patches do not write
  std::vector<long long> color_ptr_reduced;
  uint8_t *d_cell_ring_reduced = nullptr;
  // scratch of the Newton-side sweeps (pfm_newton.hip)
  unsigned long long *d_counts = nullptr;
  // gather tables of the cells at hanging vertices (DevView::hs_*): destination rows, their entry lists (hc * 32 + index:
  // index < 16 = resolved node i of cell hc, 16 + a = the hanging vertex a's own row), ascending per row
  int32_t *d_hg_rows = nullptr;
  long long *d_hg_ptr = nullptr;
  pfm::HgEntry *d_hg_list = nullptr;
  int64_t n_hg_rows = 0;
  int64_t n_hcells = 0;           // cells at hanging vertices (records of DevView::cres)
  bool hang_gather = false;       // decided at pfm_ctx_create (PFM_HANGING_ATOMIC=1: off)
  bool hang_gather_ready = false; // tables and scratch exist
  double *d_norm_partial = nullptr; // pfm_residual_norms: [2048][2] block partials + the 3 results
  double *d_partial = nullptr;
  int64_t n_partial = 0;
  uint8_t *d_cell_owned = nullptr;
  double *d_func_mat = nullptr; // per-cell Lame override of pfm_functionals_material
  // measurement (pfm_timing_enable)
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  std::string err;

  int64_t n_owned_dofs() const { return (int64_t)v.n_owned * (v.dim + 1); }
  int64_t block_rows(int b) const;
  int64_t block_nnz(int b) const;
};
