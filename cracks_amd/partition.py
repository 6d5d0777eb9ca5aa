"""Owner-computes partition of a uniform box mesh over the GPUs of a node.

Stand-in for the p4est partition the reference runs on (``parallel::distributed::
Triangulation``, cracks.cc:1083; cells with ``cell->is_locally_owned()``, cracks.cc:2201):
the global ``n[0] x n[1](x n[2])`` cell box is cut into ``p[0] x p[1](x p[2])`` sub-boxes,
rank ``r = ix + p0*(iy + p1*iz)``.

* A node belongs to the lowest rank whose sub-box touches it (deal.II's convention), i.e.
  a rank owns the nodes of its sub-box except those on a low face it shares with a
  lower-indexed neighbour.
* Owner computes: a rank also integrates the one layer of cells on its high faces so that
  every owned row is complete locally; the reverse exchange ``compress(add)``
  (cracks.cc:2470-2475) disappears and the only communication is the ghost-value import
  (cracks.cc:2147-2154), done as pairwise RCCL send/recv (cracks_amd/halo.py).
* Local numbering: owned nodes first (ascending global id), then ghost nodes (ascending
  global id).  Results are independent of the number of ranks up to the summation order
  inside one row, which is fixed by the local cell order.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import math

import numpy as np

from .mesh import Mesh


def factor_ranks(world: int, dim: int) -> Tuple[int, ...]:
    """Near-cubic factorisation of the rank count (8 -> 2x2x2, 4 -> 2x2x1, 2 -> 2x1x1)."""
    p = [1] * dim
    w = world
    d = 0
    f = 2
    while w > 1:
        while w % f:
            f += 1
        p[d % dim] *= f
        w //= f
        d += 1
    return tuple(p)


def bench_grid(world: int, dim: int, n: int) -> Tuple[int, ...]:
    """Process grid of the strong-scaling bench.  3-D boxes are cut into z-slabs while a slab keeps >= 16 node planes: the
    row-owner kernels tile (x, y) and march / iterate over z, so a slab keeps the tile efficiency of the whole box (217
    nodes = 31 tiles of 7 exactly) and has two peers instead of seven.  Measured on one MI355X, time of one rank's share
    of 216^3 (tools/bench_extra.py rank --grid): 2 ranks 7.38 (1x1x2) against 7.70 ms (2x1x1), 4 ranks 3.89 (1x1x4) against
    4.12 ms (2x2x1), 8 ranks 2.15 (1x1x8) = 2.16 ms (2x2x2).  PFM_BENCH_GRID=a,b,c overrides."""
    import os

    g = os.environ.get("PFM_BENCH_GRID")
    if g:
        p = tuple(int(x) for x in g.split(","))
        assert len(p) == dim and int(math.prod(p)) == world, f"PFM_BENCH_GRID={g} is not a {dim}-D grid of {world} ranks"
        return p
    if dim == 3 and world > 1 and (n + 1) // world >= 16:
        return (1, 1, world)
    return factor_ranks(world, dim)


def _split(n: int, p: int, i: int) -> Tuple[int, int]:
    base, rem = divmod(n, p)
    lo = i * base + min(i, rem)
    return lo, lo + base + (1 if i < rem else 0)


@dataclass
class LocalBox:
    rank: int
    dim: int
    n_global: Tuple[int, ...]  # global cells per axis
    cells_lo: Tuple[int, ...]  # extended local cell box [lo, hi)
    cells_hi: Tuple[int, ...]
    own_lo: Tuple[int, ...]  # owned node index range [lo, hi] inclusive, global node indices
    own_hi: Tuple[int, ...]
    node_lo: Tuple[int, ...]  # local node box [lo, hi] inclusive
    node_hi: Tuple[int, ...]

    def node_ids(self) -> np.ndarray:
        """Global ids (lexicographic) of the local node box, lexicographic order."""
        npg = [k + 1 for k in self.n_global]
        ax = [np.arange(self.node_lo[d], self.node_hi[d] + 1) for d in range(self.dim)]
        if self.dim == 2:
            return (ax[0][None, :] + npg[0] * ax[1][:, None]).ravel()
        return (ax[0][None, None, :] + npg[0] * (ax[1][None, :, None] + npg[1] * ax[2][:, None, None])).ravel()

    def owned_mask(self) -> np.ndarray:
        ax = [np.arange(self.node_lo[d], self.node_hi[d] + 1) for d in range(self.dim)]
        m = [(ax[d] >= self.own_lo[d]) & (ax[d] <= self.own_hi[d]) for d in range(self.dim)]
        if self.dim == 2:
            return (m[0][None, :] & m[1][:, None]).ravel()
        return (m[0][None, None, :] & m[1][None, :, None] & m[2][:, None, None]).ravel()


def local_box(n: Sequence[int], p: Sequence[int], rank: int) -> LocalBox:
    dim = len(n)
    idx = []
    r = rank
    for d in range(dim):
        idx.append(r % p[d])
        r //= p[d]
    clo, chi, olo, ohi, nlo, nhi = [], [], [], [], [], []
    for d in range(dim):
        c0, c1 = _split(n[d], p[d], idx[d])
        eh = 1 if c1 < n[d] else 0  # ghost cell layer on the high side
        el = 1 if c0 > 0 else 0  # low plane belongs to the lower neighbour
        clo.append(c0)
        chi.append(c1 + eh)
        olo.append(c0 + el)
        ohi.append(c1)
        nlo.append(c0)
        nhi.append(c1 + eh)
    return LocalBox(rank, dim, tuple(n), tuple(clo), tuple(chi), tuple(olo), tuple(ohi), tuple(nlo), tuple(nhi))


def owner_of_nodes(n: Sequence[int], p: Sequence[int], gid: np.ndarray) -> np.ndarray:
    """Rank owning each global node id."""
    dim = len(n)
    npg = [k + 1 for k in n]
    rem = gid.copy()
    rank = np.zeros(gid.shape, np.int64)
    mult = 1
    for d in range(dim):
        i = rem % npg[d]
        rem //= npg[d]
        # node i belongs to the box whose (lo, hi] contains it; node 0 to box 0
        bounds = np.array([_split(n[d], p[d], k)[1] for k in range(p[d])])
        box = np.searchsorted(bounds, i, side="left")
        box = np.minimum(box, p[d] - 1)
        rank += mult * box
        mult *= p[d]
    return rank


@dataclass
class LocalProblem:
    mesh: Mesh  # rank-local mesh, local node numbering (owned first)
    n_owned: int
    global_ids: np.ndarray  # [n_local_nodes] global node id of each local node
    peers: List[int]
    send_ptr: np.ndarray
    send_nodes: np.ndarray  # local (owned) node indices, grouped by peer, ascending global id
    recv_ptr: np.ndarray
    recv_nodes: np.ndarray  # local (ghost) node indices
    box: Optional[LocalBox] = None  # uniform-box partition only
    cell_owned: Optional[np.ndarray] = None  # uint8 per local cell: 1 = cell->is_locally_owned() (functionals)
    global_cells: Optional[np.ndarray] = None  # general partition: global index of each local cell


def build_local_problem(dim: int, n: Sequence[int], p: Sequence[int], rank: int, lo=-10.0, hi=10.0) -> LocalProblem:
    """Rank-local mesh + halo lists, built without ever forming the global mesh."""
    n = tuple(int(k) for k in n)
    p = tuple(int(k) for k in p)
    world = int(np.prod(p))
    box = local_box(n, p, rank)
    lo = np.broadcast_to(np.asarray(lo, float), (dim,))
    hi = np.broadcast_to(np.asarray(hi, float), (dim,))
    gid = box.node_ids()
    owned = box.owned_mask()
    order = np.concatenate([np.nonzero(owned)[0], np.nonzero(~owned)[0]])  # box index -> local id order
    local_of_box = np.empty(gid.size, np.int64)
    local_of_box[order] = np.arange(gid.size)
    global_ids = gid[order]
    n_owned = int(owned.sum())
    # coordinates
    npg = [k + 1 for k in n]
    rem = global_ids.copy()
    coords = np.empty((gid.size, dim))
    for d in range(dim):
        i = rem % npg[d]
        rem //= npg[d]
        coords[:, d] = lo[d] + (hi[d] - lo[d]) * i / n[d]
    # cells of the extended box, lexicographic, vertices in deal.II order
    ln = [box.node_hi[d] - box.node_lo[d] + 1 for d in range(dim)]
    lc = [box.cells_hi[d] - box.cells_lo[d] for d in range(dim)]
    if dim == 2:
        j, i = np.meshgrid(np.arange(lc[1]), np.arange(lc[0]), indexing="ij")
        base = (i + ln[0] * j).ravel()
        offs = np.array([0, 1, ln[0], ln[0] + 1])
    else:
        k, j, i = np.meshgrid(np.arange(lc[2]), np.arange(lc[1]), np.arange(lc[0]), indexing="ij")
        base = (i + ln[0] * (j + ln[1] * k)).ravel()
        sx, sy, sz = 1, ln[0], ln[0] * ln[1]
        offs = np.array([0, sx, sy, sx + sy, sz, sz + sx, sz + sy, sz + sx + sy])
    cells = local_of_box[base[:, None] + offs[None, :]].astype(np.int32)
    # physical boundary nodes (colorized ids) in local numbering
    bn = {}
    rem = global_ids.copy()
    for d in range(dim):
        i = rem % npg[d]
        rem //= npg[d]
        bn[2 * d] = np.nonzero(i == 0)[0].astype(np.int32)
        bn[2 * d + 1] = np.nonzero(i == n[d])[0].astype(np.int32)
    mesh = Mesh(dim=dim, coords=coords, cells=np.ascontiguousarray(cells), boundary_nodes=bn,
                box_shape=tuple(lc))  # the local (extended) cell box is itself a uniform lattice
    # halo lists
    ghost_gid = global_ids[n_owned:]
    ghost_owner = owner_of_nodes(n, p, ghost_gid)
    sorter = np.argsort(global_ids[:n_owned])
    peers_set = set(int(r) for r in np.unique(ghost_owner))
    send_lists = {}
    for r in range(world):
        if r == rank:
            continue
        ob = local_box(n, p, r)
        # quick reject: boxes must touch
        if any(ob.node_lo[d] > box.node_hi[d] or box.node_lo[d] > ob.node_hi[d] for d in range(dim)):
            continue
        og = ob.node_ids()
        their_ghost = og[~ob.owned_mask()]
        mine = np.intersect1d(their_ghost, global_ids[:n_owned])  # sorted ascending
        if mine.size:
            pos = sorter[np.searchsorted(global_ids[:n_owned], mine, sorter=sorter)]
            send_lists[r] = pos.astype(np.int32)
            peers_set.add(r)
    peers = sorted(peers_set)
    send_ptr, recv_ptr = [0], [0]
    send_nodes, recv_nodes = [], []
    for r in peers:
        s = send_lists.get(r, np.zeros(0, np.int32))
        send_nodes.append(s)
        send_ptr.append(send_ptr[-1] + s.size)
        sel = np.nonzero(ghost_owner == r)[0]
        sel = sel[np.argsort(ghost_gid[sel])]
        recv_nodes.append((n_owned + sel).astype(np.int32))
        recv_ptr.append(recv_ptr[-1] + sel.size)
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int32)
    return LocalProblem(mesh, n_owned, global_ids, peers, np.asarray(send_ptr, np.int64), cat(send_nodes),
                        np.asarray(recv_ptr, np.int64), cat(recv_nodes), box)


# ------------------------------------------------------------------------------------------------
# General meshes (adaptively refined, hanging nodes, slits): BASELINE config "Miehe shear with
# predictor-corrector AMR on 4 GPUs".  p4est cuts its space-filling curve into equal pieces
# (cracks.cc:1083, repartitioning inside execute_coarsening_and_refinement, cracks.cc:4137-4148); the
# stand-in orders the active cells along a Morton curve of their centres.


def morton_cell_ranks(mesh: Mesh, world: int) -> np.ndarray:
    """Rank of every active cell: Morton order of the cell centres, cut into ``world`` equal pieces."""
    c = mesh.coords[mesh.cells].mean(axis=1)
    lo, hi = c.min(axis=0), c.max(axis=0)
    q = np.minimum(((c - lo) / np.maximum(hi - lo, 1e-300) * (1 << 20)).astype(np.int64), (1 << 20) - 1)
    key = np.zeros(mesh.n_cells, np.int64)
    for b in range(20):
        for d in range(mesh.dim):
            key |= ((q[:, d] >> b) & 1) << (mesh.dim * b + d)
    order = np.argsort(key, kind="stable")
    rank = np.empty(mesh.n_cells, np.int64)
    rank[order] = (np.arange(mesh.n_cells) * world) // mesh.n_cells
    return rank


def hanging_closure_shipments(mesh: Mesh, cell_rank: np.ndarray, owner: np.ndarray) -> Dict[int, np.ndarray]:
    """The cells each rank must RECEIVE so that owner-computes is complete on a host whose ghost layer is deal.II's.

    deal.II's ghost layer holds the cells that share a vertex with a locally owned cell.  A cell K that reaches an owned
    row only through a HANGING vertex -- ``distribute_local_to_global`` moves K's contributions at the hanging vertex to
    its parents' rows (cracks.cc:2440-2447) -- need not be in it: K has the hanging vertex and one end of the coarse edge
    as vertices, the owned parent may be the other end.  The reference repairs this after the fact with
    ``compress(add)`` (cracks.cc:2470-2475).  Here the OWNER of such a cell ships it instead: rank s = cell_rank[K]
    looks at the constraint lines of K's hanging vertices (locally relevant on s) and sends K -- its vertices, their
    coordinates and constraint lines -- to every rank r != s that owns one of the parents (glue/cracks_gpu_assemble.cc:
    ship_hanging_closure_cells; one some_to_some per setup_system).  The receiver adds the cells it does not have yet to
    its local mesh, their vertices become ghost nodes like any other.  Returns {receiver rank: global cell ids}; the lists
    are what the owners send, duplicates of cells the receiver already holds included."""
    import scipy.sparse as sp

    N, nc_cells, nv = mesh.n_nodes, mesh.n_cells, mesh.nv
    out: Dict[int, List[int]] = {}
    if not mesh.hn_nodes.size:
        return {}
    C0 = sp.csr_matrix((np.ones(nc_cells * nv, np.int32), (np.repeat(np.arange(nc_cells), nv), mesh.cells.ravel())), shape=(nc_cells, N))
    H = sp.csr_matrix((np.ones(mesh.hn_parents.size, np.int32), (np.repeat(mesh.hn_nodes, np.diff(mesh.hn_ptr)), mesh.hn_parents)),
                      shape=(N, N))
    CP = (C0 @ H).tocsr()  # cell -> parents of its hanging vertices
    for k in range(nc_cells):
        par = CP.indices[CP.indptr[k]:CP.indptr[k + 1]]
        for r in np.unique(owner[par]):
            if int(r) != int(cell_rank[k]):
                out.setdefault(int(r), []).append(k)
    return {r: np.asarray(sorted(set(v)), np.int64) for r, v in out.items()}


def partition_general(mesh: Mesh, world: int, cell_rank: Optional[np.ndarray] = None, ghost_layer: str = "closure") -> List[LocalProblem]:
    """Owner-computes partition of an arbitrary Q1 mesh (hanging nodes allowed) into ``world`` rank-local problems.

    ``ghost_layer``: which cells a rank holds besides its own --
      "closure" (default): exactly the cells that contribute to an owned row (below);
      "dealii": the cells sharing a vertex with a locally owned cell, i.e. what a deal.II / p4est host has -- NOT closed
        under hanging nodes (``hanging_closure_shipments``): a model of the hole, for tests;
      "dealii+shipped": that layer plus the cells their owners ship (what glue/cracks_gpu_assemble.cc hands to the library).

    * a node belongs to the lowest rank owning a cell that has it as a vertex (deal.II's rule for dofs);
    * a rank's local cells are all cells that contribute to a row it owns: cells with an owned vertex, and cells
      with a hanging vertex one of whose parents is owned (``distribute_local_to_global`` moves those
      contributions to the parents' rows, cracks.cc:2440-2447) -- every owned row is complete locally and the
      ``compress(add)`` exchange of the reference disappears, as for the box partition;
    * local nodes = vertices of the local cells and the parents of their hanging vertices: owned first, then
      ghosts, each group by ascending global id; the ghost values come from the owners (HaloExchange).
    """
    import scipy.sparse as sp

    N, nc_cells, nv = mesh.n_nodes, mesh.n_cells, mesh.nv
    if cell_rank is None:
        cell_rank = morton_cell_ranks(mesh, world)
    cell_rank = np.asarray(cell_rank, np.int64)
    C = sp.csr_matrix((np.ones(nc_cells * nv, np.int32), (np.repeat(np.arange(nc_cells), nv), mesh.cells.ravel())),
                      shape=(nc_cells, N))
    if mesh.hn_nodes.size:
        H = sp.csr_matrix((np.ones(mesh.hn_parents.size, np.int32),
                           (np.repeat(mesh.hn_nodes, np.diff(mesh.hn_ptr)), mesh.hn_parents)), shape=(N, N))
        C = (C + C @ H).tocsr()  # cell -> vertices and parents of hanging vertices
    owner = np.full(N, world, np.int64)
    np.minimum.at(owner, mesh.cells.ravel(), np.repeat(cell_rank, nv))
    assert (owner < world).all(), "node without a cell"
    hn_index = np.full(N, -1, np.int64)
    hn_index[mesh.hn_nodes] = np.arange(mesh.hn_nodes.size)
    if ghost_layer not in ("closure", "dealii", "dealii+shipped"):
        raise ValueError(ghost_layer)
    shipped = hanging_closure_shipments(mesh, cell_rank, owner) if ghost_layer == "dealii+shipped" else {}
    C0 = sp.csr_matrix((np.ones(nc_cells * nv, np.int32), (np.repeat(np.arange(nc_cells), nv), mesh.cells.ravel())), shape=(nc_cells, N))
    parts = []
    for r in range(world):
        owned_mask = owner == r
        if ghost_layer == "closure":
            lc = np.nonzero(C @ owned_mask.astype(np.int32) > 0)[0]
        else:
            mine = cell_rank == r
            vmine = np.zeros(N, np.int32)
            vmine[mesh.cells[mine].ravel()] = 1
            keep = (C0 @ vmine > 0) | mine
            if r in shipped:
                keep[shipped[r]] = True
            lc = np.nonzero(keep)[0]
        reach = np.zeros(N, bool)
        sub = C[lc]
        reach[sub.indices] = True
        owned_ids = np.nonzero(owned_mask)[0]
        ghost_ids = np.nonzero(reach & ~owned_mask)[0]
        gids = np.concatenate([owned_ids, ghost_ids])
        g2l = np.full(N, -1, np.int64)
        g2l[gids] = np.arange(gids.size)
        cells = g2l[mesh.cells[lc]]
        assert (cells >= 0).all()
        hsel = hn_index[gids]
        hsel = hsel[hsel >= 0]
        hsel.sort()
        cnt = np.diff(mesh.hn_ptr)[hsel]
        hptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        take = np.concatenate([np.arange(mesh.hn_ptr[k], mesh.hn_ptr[k + 1]) for k in hsel]) if hsel.size else np.zeros(0, np.int64)
        hpar = g2l[mesh.hn_parents[take]]
        # a hanging node that is only a ghost neighbour still needs its parents for the column resolution; they are
        # local because the cell that brought the node in also brings its parents
        assert (hpar >= 0).all()
        bn = {b: g2l[nodes[reach[nodes] | owned_mask[nodes]]].astype(np.int32) for b, nodes in mesh.boundary_nodes.items()}
        lmesh = Mesh(dim=mesh.dim, coords=np.ascontiguousarray(mesh.coords[gids]), cells=np.ascontiguousarray(cells.astype(np.int32)),
                     boundary_nodes=bn, hn_nodes=g2l[mesh.hn_nodes[hsel]].astype(np.int32), hn_ptr=hptr,
                     hn_parents=hpar.astype(np.int32), hn_weights=mesh.hn_weights[take].copy())
        parts.append(dict(mesh=lmesh, n_owned=int(owned_ids.size), gids=gids, g2l=g2l, ghost_ids=ghost_ids, lc=lc))
    out = []
    for r, pr in enumerate(parts):
        ghost_owner = owner[pr["ghost_ids"]]
        peers = set(int(k) for k in np.unique(ghost_owner))
        for s, ps in enumerate(parts):
            if s != r and (owner[ps["ghost_ids"]] == r).any():
                peers.add(s)
        peers = sorted(peers)
        send_ptr, recv_ptr, send_nodes, recv_nodes = [0], [0], [], []
        for s in peers:
            theirs = parts[s]["ghost_ids"]
            mine = theirs[owner[theirs] == r]  # ascending global id = the order rank s receives in
            send_nodes.append(pr["g2l"][mine].astype(np.int32))
            send_ptr.append(send_ptr[-1] + mine.size)
            sel = pr["ghost_ids"][ghost_owner == s]
            recv_nodes.append(pr["g2l"][sel].astype(np.int32))
            recv_ptr.append(recv_ptr[-1] + sel.size)
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int32)
        out.append(LocalProblem(pr["mesh"], pr["n_owned"], pr["gids"], peers, np.asarray(send_ptr, np.int64), cat(send_nodes),
                                np.asarray(recv_ptr, np.int64), cat(recv_nodes), None,
                                (cell_rank[pr["lc"]] == r).astype(np.uint8), pr["lc"]))
    return out
