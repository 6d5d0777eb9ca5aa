"""Synthetic Q1 meshes, DoF layouts, constraints and sparsity patterns (host harness).

The reference gets all of this from deal.II/p4est (``setup_mesh`` cracks.cc:1194-1303,
``setup_system`` cracks.cc:1579-1680, ``set_boundary_conditions`` cracks.cc:2567-2697) and
those libraries stay in the host application ("OUT OF SCOPE").  This module
is the stand-in the parity tests and ``bench.py`` use to produce *inputs* for the
assembly hot path in the exact form the reference hands them over:

* cells as vertex lists in deal.II's lexicographic vertex order (v = x + 2y + 4z),
* local dof ``i`` <-> (vertex ``i // (dim+1)``, component ``i % (dim+1)``), components
  ``0..dim-1`` = displacement, ``dim`` = phase field (cracks.cc:980-996),
* the two DoF numberings the reference uses: one interleaved block for the direct solver,
  or component-wise renumbered ``[u | phi]`` blocks for the iterative solver
  (cracks.cc:1587-1590),
* ``AffineConstraints``-like tables (hanging nodes with weights, homogeneous Dirichlet /
  active-set lines, merged and closed as at cracks.cc:1636-1642 and 2909-2911).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

__all__ = [
    "Mesh",
    "DofLayout",
    "ConstraintSet",
    "box_mesh",
    "refine_cells",
    "sneddon_2d_prerefined_mesh",
    "slit_mesh",
    "initial_values_sneddon",
    "hanging_constraints",
    "update_constraints",
    "node_graph",
    "dof_sparsity",
]


@dataclass
class Mesh:
    dim: int
    coords: np.ndarray  # [n_nodes, dim] float64
    cells: np.ndarray  # [n_cells, 2**dim] int32, deal.II vertex order
    # boundary id -> node indices on faces with that id ("colorize" ids of
    # GridGenerator::subdivided_hyper_rectangle: 0/1 = x lo/hi, 2/3 = y, 4/5 = z)
    boundary_nodes: Dict[int, np.ndarray] = field(default_factory=dict)
    # hanging nodes: hn_nodes[k] is constrained to sum_j hn_weights * parents
    hn_nodes: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    hn_ptr: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int64))
    hn_parents: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    hn_weights: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    # structured-box metadata (None for unstructured meshes)
    box_shape: Optional[Tuple[int, ...]] = None  # cells per direction

    @property
    def n_nodes(self) -> int:
        return int(self.coords.shape[0])

    @property
    def n_cells(self) -> int:
        return int(self.cells.shape[0])

    @property
    def nv(self) -> int:
        return 1 << self.dim

    def min_cell_diameter(self) -> float:
        """``min_cell_diameter`` of cracks.cc:3824-3835 (min over active cells)."""
        return float(self.cell_diameters().min())

    def cell_diameters(self) -> np.ndarray:
        x = self.coords[self.cells]  # [n_cells, nv, dim]
        nv = self.nv
        d = np.zeros(self.n_cells)
        for v in range(nv // 2):
            o = nv - 1 - v
            d = np.maximum(d, np.linalg.norm(x[:, v] - x[:, o], axis=1))
        return d


def box_mesh(dim: int, n, lo=-10.0, hi=10.0) -> Mesh:
    """Uniform ``n^dim`` (or ``n[0] x n[1] x ...``) mesh of ``[lo,hi]^dim`` with
    lexicographic vertex numbering and colorized boundary ids — the stand-in for
    ``subdivided_hyper_rectangle(..., colorize=true)`` + ``refine_global``
    (cracks.cc:1248-1253, 1534)."""
    if np.isscalar(n):
        n = (int(n),) * dim
    n = tuple(int(k) for k in n)
    lo = np.broadcast_to(np.asarray(lo, float), (dim,))
    hi = np.broadcast_to(np.asarray(hi, float), (dim,))
    axes = [np.linspace(lo[d], hi[d], n[d] + 1) for d in range(dim)]
    npts = [k + 1 for k in n]
    if dim == 2:
        Y, X = np.meshgrid(axes[1], axes[0], indexing="ij")
        coords = np.stack([X.ravel(), Y.ravel()], axis=1)
        j, i = np.meshgrid(np.arange(n[1]), np.arange(n[0]), indexing="ij")
        base = (i + npts[0] * j).ravel()
        offs = np.array([0, 1, npts[0], npts[0] + 1])
    else:
        Z, Y, X = np.meshgrid(axes[2], axes[1], axes[0], indexing="ij")
        coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        k, j, i = np.meshgrid(np.arange(n[2]), np.arange(n[1]), np.arange(n[0]), indexing="ij")
        base = (i + npts[0] * (j + npts[1] * k)).ravel()
        sx, sy, sz = 1, npts[0], npts[0] * npts[1]
        offs = np.array([0, sx, sy, sx + sy, sz, sz + sx, sz + sy, sz + sx + sy])
    cells = (base[:, None] + offs[None, :]).astype(np.int32)
    bn: Dict[int, np.ndarray] = {}
    idx = np.arange(coords.shape[0])
    grid = np.unravel_index(idx, tuple(reversed(npts)))  # (z,) y, x
    grid = tuple(reversed(grid))  # x, y, (z)
    for d in range(dim):
        bn[2 * d] = idx[grid[d] == 0].astype(np.int32)
        bn[2 * d + 1] = idx[grid[d] == n[d]].astype(np.int32)
    return Mesh(dim=dim, coords=np.ascontiguousarray(coords), cells=np.ascontiguousarray(cells),
                boundary_nodes=bn, box_shape=n)


# reference-cell topology (deal.II ordering)
_EDGES_2D = [(0, 1), (2, 3), (0, 2), (1, 3)]
_EDGES_3D = [(0, 1), (2, 3), (4, 5), (6, 7), (0, 2), (1, 3), (4, 6), (5, 7), (0, 4), (1, 5), (2, 6), (3, 7)]
_FACES_3D = [(0, 2, 4, 6), (1, 3, 5, 7), (0, 1, 4, 5), (2, 3, 6, 7), (0, 1, 2, 3), (4, 5, 6, 7)]


def _refine_cells_loops(mesh: Mesh, flags: np.ndarray) -> Mesh:
    """One level of isotropic refinement of the flagged cells of a conforming mesh
    (``execute_coarsening_and_refinement`` stand-in, cracks.cc:4137-4148).  Midpoints of
    edges/faces shared with an unrefined cell become hanging nodes with weights 1/2
    (edges) and 1/4 (faces), as ``make_hanging_node_constraints`` produces (cracks.cc:1632)."""
    dim, nv = mesh.dim, mesh.nv
    assert mesh.hn_nodes.size == 0, "single-level refinement only"
    flags = np.asarray(flags, bool)
    coords: List[np.ndarray] = [c for c in mesh.coords]
    key2node: Dict[Tuple[int, ...], int] = {}

    def mid(nodes: Tuple[int, ...]) -> int:
        key = tuple(sorted(nodes))
        n = key2node.get(key)
        if n is None:
            n = len(coords)
            coords.append(np.mean([coords[k] for k in key], axis=0))
            key2node[key] = n
        return n

    edges = _EDGES_2D if dim == 2 else _EDGES_3D
    new_cells: List[List[int]] = []
    # edges / faces that belong to an unrefined (still active) cell
    coarse_edges = set()
    coarse_faces = set()
    for c in np.nonzero(~flags)[0]:
        v = mesh.cells[c]
        new_cells.append([int(k) for k in v])
        for a, b in edges:
            coarse_edges.add(tuple(sorted((int(v[a]), int(v[b])))))
        if dim == 3:
            for f in _FACES_3D:
                coarse_faces.add(tuple(sorted(int(v[k]) for k in f)))
    for c in np.nonzero(flags)[0]:
        v = [int(k) for k in mesh.cells[c]]
        # 3^dim lattice of the refined cell
        lat: Dict[Tuple[int, ...], int] = {}
        for p in np.ndindex(*([3] * dim)):
            # p is (x, y(, z)) position on the 3-lattice; the corners involved are those
            # whose bit d equals p[d]/2 when p[d] is even, both when p[d] == 1
            corner_sets = [[0, 1] if p[d] == 1 else [p[d] // 2] for d in range(dim)]
            corners = [0]
            for d in range(dim):
                corners = [cc | (bit << d) for cc in corners for bit in corner_sets[d]]
            nodes = tuple(v[cc] for cc in corners)
            lat[p] = nodes[0] if len(nodes) == 1 else mid(nodes)
        for child in np.ndindex(*([2] * dim)):  # child offset (x, y(, z))
            cell = []
            for vv in range(nv):
                p = tuple(child[d] + ((vv >> d) & 1) for d in range(dim))
                cell.append(lat[p])
            new_cells.append(cell)
    # hanging nodes
    hn_nodes, hn_ptr, hn_par, hn_w = [], [0], [], []
    for key, n in sorted(key2node.items(), key=lambda kv: kv[1]):
        if len(key) == 2 and key in coarse_edges:
            hn_nodes.append(n)
            hn_par += list(key)
            hn_w += [0.5, 0.5]
            hn_ptr.append(len(hn_par))
        elif len(key) == 4 and key in coarse_faces:
            hn_nodes.append(n)
            hn_par += list(key)
            hn_w += [0.25] * 4
            hn_ptr.append(len(hn_par))
    coords_a = np.asarray(coords, float)
    # boundary ids: a new node is on boundary b if all its parents are
    bn = {}
    for b, nodes in mesh.boundary_nodes.items():
        s = set(int(k) for k in nodes)
        extra = [n for key, n in key2node.items() if all(k in s for k in key)]
        bn[b] = np.asarray(sorted(s | set(extra)), np.int32)
    return Mesh(dim=dim, coords=coords_a, cells=np.asarray(new_cells, np.int32), boundary_nodes=bn,
                hn_nodes=np.asarray(hn_nodes, np.int32), hn_ptr=np.asarray(hn_ptr, np.int64),
                hn_parents=np.asarray(hn_par, np.int32), hn_weights=np.asarray(hn_w, float))


def refine_cells(mesh: Mesh, flags: np.ndarray) -> Mesh:
    """``_refine_cells_loops`` with array operations: the same mesh, node for node and cell for cell (new nodes are numbered
    in the order in which the loop version meets them: flagged cells in order, positions of the 3^dim lattice in
    ``np.ndindex`` order; tests/test_oracle_consistency.py compares the two).  The loop version takes 17 s for the 1.1e6-cell
    bench mesh of bench.py's ``overlay_3d``, this one a second."""
    dim, nv = mesh.dim, mesh.nv
    assert mesh.hn_nodes.size == 0, "single-level refinement only"
    flags = np.asarray(flags, bool)
    n0 = mesh.n_nodes
    N = np.int64(n0)
    cf = mesh.cells[flags].astype(np.int64)  # [F, nv]
    F = cf.shape[0]
    P = list(np.ndindex(*([3] * dim)))
    # corner sets of every lattice position, in the loop version's order
    pos_corners = []
    for p in P:
        corner_sets = [[0, 1] if p[d] == 1 else [p[d] // 2] for d in range(dim)]
        corners = [0]
        for d in range(dim):
            corners = [cc | (bit << d) for cc in corners for bit in corner_sets[d]]
        pos_corners.append(corners)
    lat = np.empty((F, len(P)), np.int64)  # node of every lattice position
    groups = {}  # key length -> list of position indices
    for q, corners in enumerate(pos_corners):
        if len(corners) == 1:
            lat[:, q] = cf[:, corners[0]]
        else:
            groups.setdefault(len(corners), []).append(q)
    new_keys, new_first, new_len = [], [], []
    pending = []  # (k, qs, inverse index into the group's unique keys)
    for k, qs in sorted(groups.items()):
        keys = np.stack([np.sort(cf[:, pos_corners[q]], axis=1) for q in qs], axis=1).reshape(-1, k)  # [F * len(qs), k]
        seq = (np.arange(F, dtype=np.int64)[:, None] * len(P) + np.asarray(qs, np.int64)[None, :]).reshape(-1)
        # one int64 per key where that is exact (pairs), rows otherwise
        if k == 2:
            code = keys[:, 0] * N + keys[:, 1]
            uq, inv = np.unique(code, return_inverse=True)
            ukeys = np.stack([uq // N, uq % N], axis=1)
        elif k == nv:  # the centre of a cell: met once
            ukeys, inv = keys, np.arange(keys.shape[0])
        elif k == 4 and float(N) ** 3 < 2.0 ** 62:
            # a face of a conforming hex mesh is known by its two smallest nodes and its largest one
            _, idx, inv = np.unique((keys[:, 0] * N + keys[:, 1]) * N + keys[:, 3], return_index=True, return_inverse=True)
            ukeys = keys[idx]
        else:
            ukeys, inv = np.unique(keys, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        first = np.full(ukeys.shape[0], np.iinfo(np.int64).max, np.int64)
        np.minimum.at(first, inv, seq)
        new_keys.append(ukeys)
        new_first.append(first)
        new_len.append(k)
        pending.append((k, qs, inv))
    # node ids in the order of first encounter over all kinds of midpoints
    all_first = np.concatenate(new_first) if new_first else np.empty(0, np.int64)
    order = np.argsort(all_first, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    offs = np.cumsum([0] + [f.size for f in new_first])
    n_new = int(all_first.size)
    coords_new = np.empty((n_new, dim), float)
    key_len = np.empty(n_new, np.int32)
    key_pad = np.full((n_new, 1 << dim), -1, np.int64)
    for g, (k, qs, inv) in enumerate(pending):
        ids = n0 + rank[offs[g]:offs[g + 1]]  # node id of every unique key of this group
        lat[:, qs] = ids[inv].reshape(F, len(qs))
        ck = mesh.coords[new_keys[g]]  # [U, k, dim]
        acc = ck[:, 0].copy()
        for i in range(1, k):
            acc = acc + ck[:, i]  # the summation order of np.mean over the sorted key
        coords_new[ids - n0] = acc / k
        key_len[ids - n0] = k
        key_pad[ids - n0, :k] = new_keys[g]
    child_pos = []
    for child in np.ndindex(*([2] * dim)):
        child_pos.append([P.index(tuple(child[d] + ((vv >> d) & 1) for d in range(dim))) for vv in range(nv)])
    children = lat[:, np.asarray(child_pos)].reshape(-1, nv)  # [F * 2^dim, nv], children of a cell consecutive
    cells = np.concatenate([mesh.cells[~flags].astype(np.int64), children]).astype(np.int32)
    # hanging nodes: midpoints of edges / faces that also belong to a cell that stays
    cc = mesh.cells[~flags].astype(np.int64)
    edges = _EDGES_2D if dim == 2 else _EDGES_3D
    hang = np.zeros(n_new, bool)
    if cc.size and n_new:
        e = np.concatenate([np.sort(cc[:, list(ab)], axis=1) for ab in edges])
        ecode = np.unique(e[:, 0] * N + e[:, 1])
        m2 = key_len == 2
        hang[m2] = np.isin(key_pad[m2, 0] * N + key_pad[m2, 1], ecode)
        if dim == 3:
            f = np.concatenate([np.sort(cc[:, list(fc)], axis=1) for fc in _FACES_3D])
            m4 = key_len == 4
            if float(N) ** 3 < 2.0 ** 62:
                fcode = np.unique((f[:, 0] * N + f[:, 1]) * N + f[:, 3])
                hang[m4] = np.isin((key_pad[m4, 0] * N + key_pad[m4, 1]) * N + key_pad[m4, 3], fcode)
            else:
                fs = set(map(tuple, np.unique(f, axis=0)))
                hang[m4] = [tuple(r) in fs for r in key_pad[m4, :4]]
    hn = np.nonzero(hang)[0]
    kl = key_len[hn].astype(np.int64)
    hn_ptr = np.concatenate([[0], np.cumsum(kl)]).astype(np.int64)
    hn_par = key_pad[hn][np.arange(1 << dim)[None, :] < kl[:, None]].astype(np.int32)
    hn_w = np.repeat(1.0 / kl, kl)
    bn = {}
    for b, nodes in mesh.boundary_nodes.items():
        on = np.zeros(n0 + 1, bool)  # (slot n0: the padding of short keys counts as "on")
        on[np.asarray(nodes, np.int64)] = True
        on[n0] = True
        kp = np.where(key_pad < 0, n0, key_pad)
        extra = n0 + np.nonzero(on[kp].all(axis=1))[0]
        bn[b] = np.unique(np.concatenate([np.asarray(nodes, np.int64), extra])).astype(np.int32)
    return Mesh(dim=dim, coords=np.concatenate([mesh.coords, coords_new]) if n_new else np.asarray(mesh.coords, float).copy(),
                cells=cells, boundary_nodes=bn, hn_nodes=(n0 + hn).astype(np.int32), hn_ptr=hn_ptr,
                hn_parents=hn_par if hn.size else np.zeros(0, np.int32), hn_weights=hn_w if hn.size else np.zeros(0, float))


def sneddon_2d_prerefined_mesh() -> Mesh:
    """Mesh of tests/sneddon_2d_1.prm: 10x10 cells on [-10,10]^2, one 'fixed preref sneddon'
    step (cracks.cc:3902-3924: refine cells with a vertex in [-2.5,2.5]x[-1.25,1.25]) =>
    124 cells, 151 nodes, 12 hanging nodes (tests/sneddon_2d_1.output:28)."""
    m = box_mesh(2, 10)
    x = m.coords[m.cells]  # [cells, 4, 2]
    inside = (x[..., 0] <= 2.5) & (x[..., 0] >= -2.5) & (x[..., 1] <= 1.25) & (x[..., 1] >= -1.25)
    return refine_cells(m, inside.any(axis=1))


def initial_values_multiple_het(mesh: Mesh, min_cell_diameter: float) -> np.ndarray:
    """Nodal phase field of ``InitialValuesMultipleHet`` (cracks.cc:586-640): 0 inside the initial cracks, else 1.
    3-D: two bars of cross-section ``width = min_cell_diameter`` (cracks.cc:601-612); 2-D: "example 3" (617-624)."""
    p = mesh.coords
    w = 0.5 * min_cell_diameter
    if mesh.dim == 3:
        a = ((p[:, 0] >= 2.6 - w) & (p[:, 0] <= 2.6 + w) & (p[:, 1] >= 3.8 - w) & (p[:, 1] <= 5.5 + w) &
             (p[:, 2] >= 4.0 - w) & (p[:, 2] <= 4.0 + w))
        b = ((p[:, 0] >= 5.5 - w) & (p[:, 0] <= 7.0 + w) & (p[:, 1] >= 4.0 - w) & (p[:, 1] <= 4.0 + w) &
             (p[:, 2] >= 6.0 - w) & (p[:, 2] <= 6.0 + w))
    else:
        a = (p[:, 0] >= 2.5 - w) & (p[:, 0] <= 2.5 + w) & (p[:, 1] >= 0.8) & (p[:, 1] <= 1.5)
        b = (p[:, 0] >= 0.5) & (p[:, 0] <= 1.5) & (p[:, 1] >= 3.0 - w) & (p[:, 1] <= 3.0 + w)
    return np.where(a | b, 0.0, 1.0)


def hetero_3d_prerefined_mesh() -> Mesh:
    """Mesh of tests/hetero_3d_1.prm: ``meshes/unit_cube_10.inp`` ([0,10]^3) refined 3 times globally (512 cells), then one
    'phase field' prerefinement step (cracks.cc:3971-3995: cells with a vertex where phi < 0.4, phi interpolated with the
    coarse h, cracks.cc:4134-4160) => 932 cells, 1322 nodes (tests/hetero_3d_1.mpirun-4.output:3-8, :30)."""
    m = box_mesh(3, 8, 0.0, 10.0)
    phi = initial_values_multiple_het(m, m.min_cell_diameter())
    return refine_cells(m, (phi[m.cells] < 0.4).any(axis=1))


def slit_mesh(n_refine: int = 3) -> Mesh:
    """Unit square with a slit from (0.5,0.5) to (1,0.5): the coarse 2x2 mesh of the
    reference's ``meshes/unit_slit.inp`` (duplicated nodes along the slit, boundary ids
    0 left, 1 right, 2 bottom, 3 top, 4 lower slit face, 7 upper slit face) refined
    ``n_refine`` times globally (tests/miehe_shear_1.prm: 3 => 16x16 cells, 297 nodes)."""
    n = 2 ** (n_refine + 1)
    npt = n + 1
    h = 1.0 / n
    half = n // 2
    ids = -np.ones((npt, npt), np.int64)  # [j (y), i (x)] primary numbering
    coords = []
    for j in range(npt):
        for i in range(npt):
            ids[j, i] = len(coords)
            coords.append((i * h, j * h))
    # duplicated nodes on the slit (x > 0.5, y = 0.5): the *lower* cells use the copies
    dup = {}
    for i in range(half + 1, npt):
        dup[i] = len(coords)
        coords.append((i * h, half * h))
    cells = []
    for j in range(n):
        for i in range(n):
            v = [ids[j, i], ids[j, i + 1], ids[j + 1, i], ids[j + 1, i + 1]]
            if j == half - 1:  # cell just below the slit line: its top vertices use the copies
                if i + 0 > half:
                    v[2] = dup[i]
                if i + 1 > half:
                    v[3] = dup[i + 1]
            cells.append(v)
    coords_a = np.asarray(coords, float)
    bn = {
        0: ids[:, 0].copy(),
        1: np.concatenate([ids[:, n], [dup[n]]]),
        2: ids[0, :].copy(),
        3: ids[n, :].copy(),
        # lower slit face (boundary id 4): node 4 of the .inp (x=0.5) .. copies
        4: np.asarray([ids[half, half]] + [dup[i] for i in range(half + 1, npt)]),
        # upper slit face (boundary id 7)
        7: ids[half, half:].copy(),
    }
    bn = {k: np.asarray(v, np.int32) for k, v in bn.items()}
    return Mesh(dim=2, coords=coords_a, cells=np.asarray(cells, np.int32), boundary_nodes=bn)


def initial_values_sneddon(mesh: Mesh, min_cell_diameter: float) -> np.ndarray:
    """Nodal phase field of ``InitialValuesSneddon`` (cracks.cc:380-406): 0 inside the
    crack ``r^2 <= l0^2 (l0 = 1), |2 y| <= 2 h``, else 1 (displacements are 0)."""
    p = mesh.coords
    l0 = 1.0
    thickness = 2.0 * min_cell_diameter
    r2 = p[:, 0] ** 2 if mesh.dim == 2 else p[:, 0] ** 2 + p[:, 2] ** 2
    crack = (r2 <= l0 * l0) & (np.abs(2.0 * p[:, 1]) <= thickness)
    return np.where(crack, 0.0, 1.0)


class DofLayout:
    """Node/component -> global dof index.

    ``blocked=False``: one block, ``dof = node*(dim+1) + comp`` (direct-solver layout).
    ``blocked=True``: component-wise renumbering with sub-blocks ``[u | phi]``
    (cracks.cc:1587-1590): ``u`` dofs ``node*dim + comp``, then ``phi`` dofs ``dim*N + node``.
    """

    def __init__(self, n_nodes: int, dim: int, blocked: bool):
        self.n_nodes, self.dim, self.blocked = int(n_nodes), int(dim), bool(blocked)
        self.nc = dim + 1
        self.n_dofs = self.n_nodes * self.nc
        self.n_u = self.n_nodes * dim

    def dof(self, node, comp):
        node = np.asarray(node)
        comp = np.asarray(comp)
        if not self.blocked:
            return node * self.nc + comp
        return np.where(comp < self.dim, node * self.dim + comp, self.n_u + node)

    def cell_dofs(self, cells: np.ndarray) -> np.ndarray:
        nv = cells.shape[1]
        comp = np.tile(np.arange(self.nc), nv)[None, :]
        node = np.repeat(cells, self.nc, axis=1)
        return np.ascontiguousarray(self.dof(node, comp).astype(np.int32))

    def node_comp_of_dof(self):
        """Inverse map: arrays (node, comp) indexed by dof."""
        node = np.empty(self.n_dofs, np.int64)
        comp = np.empty(self.n_dofs, np.int64)
        for c in range(self.nc):
            d = self.dof(np.arange(self.n_nodes), c)
            node[d] = np.arange(self.n_nodes)
            comp[d] = c
        return node, comp

    def pack(self, u: np.ndarray, phi: np.ndarray) -> np.ndarray:
        """Build a dof vector from nodal displacement [N,dim] and phase field [N]."""
        v = np.empty(self.n_dofs)
        n = np.arange(self.n_nodes)
        for c in range(self.dim):
            v[self.dof(n, c)] = u[:, c]
        v[self.dof(n, self.dim)] = phi
        return v


@dataclass
class ConstraintSet:
    """Closed ``AffineConstraints`` with zero inhomogeneities, CSR over dofs."""
    flag: np.ndarray  # uint8 [n_dofs]
    ptr: np.ndarray  # int64 [n_dofs+1]
    col: np.ndarray  # int32
    w: np.ndarray  # float64

    @staticmethod
    def from_lines(n_dofs: int, lines: Dict[int, List[Tuple[int, float]]]) -> "ConstraintSet":
        flag = np.zeros(n_dofs, np.uint8)
        counts = np.zeros(n_dofs + 1, np.int64)
        for d, ent in lines.items():
            flag[d] = 1
            counts[d + 1] = len(ent)
        ptr = np.cumsum(counts)
        col = np.zeros(int(ptr[-1]), np.int32)
        w = np.zeros(int(ptr[-1]), np.float64)
        for d, ent in lines.items():
            for k, (c, ww) in enumerate(sorted(ent)):
                col[ptr[d] + k] = c
                w[ptr[d] + k] = ww
        return ConstraintSet(flag, ptr, col, w)

    def lines(self) -> Dict[int, List[Tuple[int, float]]]:
        out = {}
        for d in np.nonzero(self.flag)[0]:
            out[int(d)] = [(int(self.col[k]), float(self.w[k])) for k in range(self.ptr[d], self.ptr[d + 1])]
        return out

    def set_zero(self, v: np.ndarray) -> np.ndarray:
        """``constraints.set_zero(v)`` (cracks.cc:2793)."""
        v = v.copy()
        v[self.flag.astype(bool)] = 0.0
        return v

    def distribute(self, v: np.ndarray) -> np.ndarray:
        """``constraints.distribute(v)`` (cracks.cc:2788) with zero inhomogeneities."""
        v = v.copy()
        for d in np.nonzero(self.flag)[0]:
            s = 0.0
            for k in range(self.ptr[d], self.ptr[d + 1]):
                s += self.w[k] * v[self.col[k]]
            v[d] = s
        return v


def _close(lines: Dict[int, List[Tuple[int, float]]]) -> Dict[int, List[Tuple[int, float]]]:
    """``AffineConstraints::close()``: resolve entries that point at constrained dofs."""
    changed = True
    while changed:
        changed = False
        for d, ent in lines.items():
            new: Dict[int, float] = {}
            for c, w in ent:
                if c in lines:
                    changed = True
                    for c2, w2 in lines[c]:
                        new[c2] = new.get(c2, 0.0) + w * w2
                else:
                    new[c] = new.get(c, 0.0) + w
            lines[d] = sorted(new.items())
    return lines


def hanging_constraints(mesh: Mesh, layout: DofLayout) -> ConstraintSet:
    """``constraints_hanging_nodes`` (cracks.cc:1630-1635): every component of a hanging
    node is constrained to its parents with the node's weights."""
    lines: Dict[int, List[Tuple[int, float]]] = {}
    for k, n in enumerate(mesh.hn_nodes):
        par = mesh.hn_parents[mesh.hn_ptr[k]:mesh.hn_ptr[k + 1]]
        wts = mesh.hn_weights[mesh.hn_ptr[k]:mesh.hn_ptr[k + 1]]
        for c in range(layout.nc):
            lines[int(layout.dof(n, c))] = [(int(layout.dof(p, c)), float(w)) for p, w in zip(par, wts)]
    return ConstraintSet.from_lines(layout.n_dofs, _close(lines))


def update_constraints(mesh: Mesh, layout: DofLayout, dirichlet_dofs: Iterable[int],
                       active_dofs: Iterable[int] = ()) -> ConstraintSet:
    """``constraints_update`` as rebuilt at cracks.cc:1636-1642 / 2826-2911: homogeneous
    lines for the Dirichlet dofs (``set_newton_bc``) and the active-set dofs
    (``add_line`` + zero inhomogeneity, cracks.cc:2878-2879), merged with the hanging-node
    constraints (``right_object_wins``) and closed."""
    lines: Dict[int, List[Tuple[int, float]]] = {}
    for d in list(dirichlet_dofs) + list(active_dofs):
        lines[int(d)] = []
    lines.update(hanging_constraints(mesh, layout).lines())  # right object wins
    return ConstraintSet.from_lines(layout.n_dofs, _close(lines))


def boundary_dofs(mesh: Mesh, layout: DofLayout, spec: Iterable[Tuple[int, Iterable[int]]]) -> np.ndarray:
    """Dofs of the listed (boundary id, components) pairs —
    ``VectorTools::interpolate_boundary_values`` with a component mask."""
    out = []
    for bid, comps in spec:
        nodes = mesh.boundary_nodes[bid]
        for c in comps:
            out.append(layout.dof(nodes, c))
    return np.unique(np.concatenate(out)) if out else np.zeros(0, np.int64)


def sneddon_dirichlet_dofs(mesh: Mesh, layout: DofLayout) -> np.ndarray:
    """cracks.cc:2575-2583 (2-D: ids 0..3) and 2686-2694 (3-D: ids 0..5): all
    displacement components on the whole boundary."""
    return boundary_dofs(mesh, layout, [(b, range(mesh.dim)) for b in range(2 * mesh.dim)])


def miehe_shear_dirichlet_dofs(mesh: Mesh, layout: DofLayout) -> np.ndarray:
    """cracks.cc:2600-2625: u_y on ids 0, 1, 4; u on ids 2 and 3."""
    return boundary_dofs(mesh, layout, [(0, [1]), (1, [1]), (2, [0, 1]), (3, [0, 1]), (4, [1])])


def node_graph(mesh: Mesh):
    """Node adjacency (CSR, sorted, with self loops): nodes coupled through a cell after
    resolving hanging nodes to their parents — the node-level image of
    ``DoFTools::make_sparsity_pattern(dof_handler, csp, constraints, ...)`` (cracks.cc:1647)."""
    import scipy.sparse as sp

    n_cells, nv = mesh.cells.shape
    rows = np.repeat(np.arange(n_cells), nv)
    cols = mesh.cells.ravel()
    C = sp.csr_matrix((np.ones(rows.size, np.int8), (rows, cols)), shape=(n_cells, mesh.n_nodes))
    if mesh.hn_nodes.size:
        # cell -> parents of its hanging nodes
        H = sp.csr_matrix(
            (np.ones(mesh.hn_parents.size, np.int8),
             (np.repeat(mesh.hn_nodes, np.diff(mesh.hn_ptr)), mesh.hn_parents)),
            shape=(mesh.n_nodes, mesh.n_nodes))
        C = C + C @ H
        C.data[:] = 1
    A = (C.T.astype(np.int32) @ C.astype(np.int32)).tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int64), A.indices.astype(np.int32)


def dof_sparsity(mesh: Mesh, layout: DofLayout):
    """Global dof-level CSR pattern (full component coupling, cracks.cc:1644-1654)."""
    import scipy.sparse as sp

    nptr, nadj = node_graph(mesh)
    N, nc = mesh.n_nodes, layout.nc
    G = sp.csr_matrix((np.ones(nadj.size, np.int8), nadj, nptr), shape=(N, N)).tocoo()
    rows, cols = [], []
    for ci in range(nc):
        for cj in range(nc):
            rows.append(layout.dof(G.row, ci))
            cols.append(layout.dof(G.col, cj))
    A = sp.csr_matrix((np.ones(sum(r.size for r in rows), np.int8),
                       (np.concatenate(rows), np.concatenate(cols))),
                      shape=(layout.n_dofs, layout.n_dofs))
    A.sort_indices()
    return A.indptr.astype(np.int64), A.indices.astype(np.int32)
