"""Parity at BENCH size (BASELINE.json configs[2]: 216^3 hexes, the bench.py workload), not only at oracle-sized meshes.

The oracle cannot assemble 10^7 cells in test time, but a row only depends on the 8 cells around its node: a few hundred
rows -- at the mesh faces, at the seams of the (u,u) tiles (8 x 4 nodes), the phase-field tiles (7 x 7), the residual
tiles (15 x 15), around the z-chunk boundaries of the marching kernels, plus random interior nodes -- are compared with the
oracle run on the sub-mesh of exactly those cells.  The CSR offsets of the rows are recomputed here from the lattice
(ascending columns), so the test also pins the pattern the library reports at this size.

PFM_FULLSIZE_N overrides the edge length (default 216)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import oracle_api as O
from cracks_amd import mesh as M

pytestmark = pytest.mark.gpu


def _axis_samples(n):
    s = {0, 1, 6, 7, 8, 13, 14, 15, 16, n // 2, n - 9, n - 8, n - 7, n - 1, n}
    return sorted(k for k in s if 0 <= k <= n)


@pytest.mark.timeout(1500)
def test_rows_of_the_bench_mesh_match_the_oracle():
    import torch
    import bench
    from cracks_amd.assembler import Assembler

    n = int(os.environ.get("PFM_FULLSIZE_N", "216"))
    dim = 3
    g = M.box_mesh(dim, n)
    NP = n + 1
    N = g.n_nodes
    h = (20.0 / n) * np.sqrt(dim)
    u, phi, po, poo, flags = bench.synthetic_state(g, np.arange(N), h, dim)
    prm = bench.sneddon_params(h, dim)
    asm = Assembler(g, blocked=True)
    assert asm.ctx.kernel_path == 1
    asm.set_params(prm)
    asm.set_constraints(flags)
    pack = lambda uu, pp: np.concatenate([uu.reshape(-1), pp])
    asm.set_vectors(pack(u, phi), pack(0 * u, po), pack(0 * u, poo))
    asm.assemble_system(False)
    asm.synchronize()
    res_full = asm.system_pde_residual.clone()
    asm.assemble_nl_residual()
    asm.synchronize()

    # ---- sample nodes
    rng = np.random.default_rng(216)
    ax = _axis_samples(n)
    zc = set(ax)
    for c in (12, 18, 24, 25):  # chunk seams of the marching kernels whatever length the dispatch model chose
        zc |= {k for k in (c - 1, c, c + 1, 2 * c, 2 * c + 1) if k <= n}
    pts = set()
    for _ in range(260):
        pts.add((int(rng.choice(ax)), int(rng.choice(ax)), int(rng.choice(sorted(zc)))))
    for _ in range(120):
        pts.add(tuple(int(x) for x in rng.integers(0, NP, 3)))
    pts = sorted(pts)
    S = np.array([i + NP * (j + NP * k) for i, j, k in pts], np.int64)

    # ---- sub-mesh of the cells around the sampled nodes
    cell_ids = set()
    for i, j, k in pts:
        for dk in (-1, 0):
            for dj in (-1, 0):
                for di in (-1, 0):
                    ci, cj, ck = i + di, j + dj, k + dk
                    if 0 <= ci < n and 0 <= cj < n and 0 <= ck < n:
                        cell_ids.add(ci + n * (cj + n * ck))
    cell_ids = np.array(sorted(cell_ids), np.int64)
    gcells = g.cells[cell_ids].astype(np.int64)
    gnodes, inv = np.unique(gcells, return_inverse=True)
    sub = M.Mesh(dim=dim, coords=np.ascontiguousarray(g.coords[gnodes]), cells=inv.reshape(-1, 8).astype(np.int32))
    lay = M.DofLayout(sub.n_nodes, dim, blocked=True)
    dflag = np.zeros(lay.n_dofs, bool)
    for c in range(dim + 1):
        dflag[lay.dof(np.arange(sub.n_nodes), c)] = (flags[gnodes] >> c) & 1
    cu = M.update_constraints(sub, lay, np.nonzero(dflag)[0])
    ch = M.hanging_constraints(sub, lay)
    oprm = O.PfmParams.from_buffer_copy(bytes(prm))
    rp, ci = M.dof_sparsity(sub, lay)
    sol, old, oo = lay.pack(u[gnodes], phi[gnodes]), lay.pack(0 * u[gnodes], po[gnodes]), lay.pack(0 * u[gnodes], poo[gnodes])
    r = O.assemble(sub, lay, oprm, sol, old, oo, cu, ch, False, rp, ci)
    assert r.err == 0
    r_res = O.assemble(sub, lay, oprm, sol, old, oo, cu, ch, True)
    A_ref = sp.csr_matrix((r.values, ci, rp), shape=(lay.n_dofs,) * 2)
    loc_of = {int(gn): l for l, gn in enumerate(gnodes)}

    # ---- CSR offsets of the full lattice (columns ascending = lexicographic neighbours)
    cnt = lambda i: 1 + (i > 0) + (i < n)
    c1 = np.array([cnt(i) for i in range(NP)], np.int64)
    deg = (c1[None, None, :] * c1[None, :, None] * c1[:, None, None]).reshape(-1)
    off = np.concatenate([[0], np.cumsum(deg)])
    nnz = [9 * off[-1], 3 * off[-1], 3 * off[-1], off[-1]]
    for b in range(4):
        assert asm.ctx.pattern_size(b)[1] == nnz[b]
    vals = asm.system_pde_matrix
    res_d, tot_d = asm.system_pde_residual, asm.system_total_residual

    def fetch(t, idx):
        return t[torch.from_numpy(np.asarray(idx, np.int64)).cuda()].cpu().numpy()

    worst = 0.0
    for (i, j, k), gn in zip(pts, S):
        nb = [ii + NP * (jj + NP * kk) for kk in range(max(k - 1, 0), min(k + 1, n) + 1)
              for jj in range(max(j - 1, 0), min(j + 1, n) + 1) for ii in range(max(i - 1, 0), min(i + 1, n) + 1)]
        d = len(nb)
        assert d == deg[gn]
        o = int(off[gn])
        lnb = np.array([loc_of[q] for q in nb])
        ln = loc_of[int(gn)]
        lu = lambda node, c: lay.dof(node, c)
        got_uu = fetch(vals[0], np.arange(9 * o, 9 * o + 9 * d)).reshape(3, d, 3)
        got_up = fetch(vals[1], np.arange(3 * o, 3 * o + 3 * d)).reshape(3, d)
        got_pu = fetch(vals[2], np.arange(3 * o, 3 * o + 3 * d)).reshape(d, 3)
        got_pp = fetch(vals[3], np.arange(o, o + d))
        for c in range(3):
            row = A_ref[lu(ln, c)].toarray().ravel()
            for dd in range(3):
                worst = max(worst, np.abs(got_uu[c, :, dd] - row[lu(lnb, dd)]).max())
            worst = max(worst, np.abs(got_up[c] - row[lu(lnb, 3)]).max())
        row = A_ref[lu(ln, 3)].toarray().ravel()
        for dd in range(3):
            worst = max(worst, np.abs(got_pu[:, dd] - row[lu(lnb, dd)]).max())
        worst = max(worst, np.abs(got_pp - row[lu(lnb, 3)]).max())
        # residuals: Jacobian call (pde), residual-only call (pde + total)
        gd = [3 * gn + c for c in range(3)] + [3 * N + gn]
        ld = [lu(ln, c) for c in range(4)]
        worst = max(worst, np.abs(fetch(res_full, gd) - r.residual_pde[ld]).max())
        worst = max(worst, np.abs(fetch(res_d, gd) - r_res.residual_pde[ld]).max())
        worst = max(worst, np.abs(fetch(tot_d, gd) - r_res.residual_total[ld]).max())
    scale = max(1.0, float(np.abs(r.values).max()))
    print(f"full size {n}^3: {len(pts)} rows, {cell_ids.size} oracle cells, max |GPU - oracle| = {worst:.3e} (scale {scale:.2e})")
    assert worst < 1e-12 * scale


@pytest.mark.timeout(900)
def test_rows_of_the_config2_mesh_match_the_oracle():
    """BASELINE configs[1] at its real size: 2-D Sneddon, 1000^2 quads, residual-only (k_cart_residual2m: waves of 62
    x-consecutive nodes over chunks of node rows -> 17 wave columns x y-chunks), and the 2-D Jacobian rows of the same
    blocks of 8 x 8 cells (k_cart2d_cells).  Sampled at the mesh edges, the wave-column seams (multiples of 62), the y-chunk seams and
    random interior nodes; oracle on the sub-mesh of the cells around them."""
    import torch
    import bench
    from cracks_amd.assembler import Assembler

    n = int(os.environ.get("PFM_FULLSIZE_N2", "1000"))
    dim = 2
    g = M.box_mesh(dim, n)
    NP = n + 1
    N = g.n_nodes
    h = (20.0 / n) * np.sqrt(dim)
    u, phi, po, poo, flags = bench.synthetic_state(g, np.arange(N), h, dim)
    prm = bench.sneddon_params(h, dim)
    asm = Assembler(g, blocked=True)
    assert asm.ctx.kernel_path == 1
    asm.set_params(prm)
    asm.set_constraints(flags)
    pack = lambda uu, pp: np.concatenate([uu.reshape(-1), pp])
    asm.set_vectors(pack(u, phi), pack(0 * u, po), pack(0 * u, poo))
    asm.assemble_system(False)
    asm.synchronize()
    res_full = asm.system_pde_residual.clone()
    asm.assemble_nl_residual()
    asm.synchronize()

    rng = np.random.default_rng(1000)
    xs = {0, 1, n - 1, n, n // 2}
    for m in range(1, 17):  # wave-column seams of k_cart_residual2m (62 nodes per wave) and of 63-node waves
        xs |= {62 * m - 1, 62 * m, 62 * m + 1, 63 * m - 1, 63 * m, 63 * m + 1}
    xs = sorted(k for k in xs if 0 <= k <= n)
    ys = {0, 1, n - 1, n, n // 2}
    for c in (4, 8, 16, 32, 59, 64, 125, 250):  # y-chunk seams whatever length the dispatch model chose
        ys |= {k for k in (c - 1, c, c + 1, 2 * c - 1, 2 * c, 2 * c + 1) if k <= n}
    ys = sorted(ys)
    pts = set()
    for _ in range(300):
        pts.add((int(rng.choice(xs)), int(rng.choice(ys))))
    for _ in range(150):
        pts.add(tuple(int(x) for x in rng.integers(0, NP, 2)))
    pts = sorted(pts)
    S = np.array([i + NP * j for i, j in pts], np.int64)

    cell_ids = set()
    for i, j in pts:
        for dj in (-1, 0):
            for di in (-1, 0):
                ci, cj = i + di, j + dj
                if 0 <= ci < n and 0 <= cj < n:
                    cell_ids.add(ci + n * cj)
    cell_ids = np.array(sorted(cell_ids), np.int64)
    gcells = g.cells[cell_ids].astype(np.int64)
    gnodes, inv = np.unique(gcells, return_inverse=True)
    sub = M.Mesh(dim=dim, coords=np.ascontiguousarray(g.coords[gnodes]), cells=inv.reshape(-1, 4).astype(np.int32))
    lay = M.DofLayout(sub.n_nodes, dim, blocked=True)
    dflag = np.zeros(lay.n_dofs, bool)
    for c in range(dim + 1):
        dflag[lay.dof(np.arange(sub.n_nodes), c)] = (flags[gnodes] >> c) & 1
    cu = M.update_constraints(sub, lay, np.nonzero(dflag)[0])
    ch = M.hanging_constraints(sub, lay)
    oprm = O.PfmParams.from_buffer_copy(bytes(prm))
    rp, ci = M.dof_sparsity(sub, lay)
    sol, old, oo = lay.pack(u[gnodes], phi[gnodes]), lay.pack(0 * u[gnodes], po[gnodes]), lay.pack(0 * u[gnodes], poo[gnodes])
    r = O.assemble(sub, lay, oprm, sol, old, oo, cu, ch, False, rp, ci)
    assert r.err == 0
    r_res = O.assemble(sub, lay, oprm, sol, old, oo, cu, ch, True)
    A_ref = sp.csr_matrix((r.values, ci, rp), shape=(lay.n_dofs,) * 2)
    loc_of = {int(gn): l for l, gn in enumerate(gnodes)}

    cnt = lambda i: 1 + (i > 0) + (i < n)
    c1 = np.array([cnt(i) for i in range(NP)], np.int64)
    deg = (c1[None, :] * c1[:, None]).reshape(-1)
    off = np.concatenate([[0], np.cumsum(deg)])
    nnz = [4 * off[-1], 2 * off[-1], 2 * off[-1], off[-1]]
    for b in range(4):
        assert asm.ctx.pattern_size(b)[1] == nnz[b]
    vals = asm.system_pde_matrix
    res_d, tot_d = asm.system_pde_residual, asm.system_total_residual

    def fetch(t, idx):
        return t[torch.from_numpy(np.asarray(idx, np.int64)).cuda()].cpu().numpy()

    worst = 0.0
    for (i, j), gn in zip(pts, S):
        nb = [ii + NP * jj for jj in range(max(j - 1, 0), min(j + 1, n) + 1) for ii in range(max(i - 1, 0), min(i + 1, n) + 1)]
        d = len(nb)
        assert d == deg[gn]
        o = int(off[gn])
        lnb = np.array([loc_of[q] for q in nb])
        ln = loc_of[int(gn)]
        lu = lambda node, c: lay.dof(node, c)
        got_uu = fetch(vals[0], np.arange(4 * o, 4 * o + 4 * d)).reshape(2, d, 2)
        got_up = fetch(vals[1], np.arange(2 * o, 2 * o + 2 * d)).reshape(2, d)
        got_pu = fetch(vals[2], np.arange(2 * o, 2 * o + 2 * d)).reshape(d, 2)
        got_pp = fetch(vals[3], np.arange(o, o + d))
        for c in range(2):
            row = A_ref[lu(ln, c)].toarray().ravel()
            for dd in range(2):
                worst = max(worst, np.abs(got_uu[c, :, dd] - row[lu(lnb, dd)]).max())
            worst = max(worst, np.abs(got_up[c] - row[lu(lnb, 2)]).max())
        row = A_ref[lu(ln, 2)].toarray().ravel()
        for dd in range(2):
            worst = max(worst, np.abs(got_pu[:, dd] - row[lu(lnb, dd)]).max())
        worst = max(worst, np.abs(got_pp - row[lu(lnb, 2)]).max())
        gd = [2 * gn + c for c in range(2)] + [2 * N + gn]
        ld = [lu(ln, c) for c in range(3)]
        worst = max(worst, np.abs(fetch(res_full, gd) - r.residual_pde[ld]).max())
        worst = max(worst, np.abs(fetch(res_d, gd) - r_res.residual_pde[ld]).max())
        worst = max(worst, np.abs(fetch(tot_d, gd) - r_res.residual_total[ld]).max())
    scale = max(1.0, float(np.abs(r.values).max()))
    print(f"full size {n}^2: {len(pts)} rows, {cell_ids.size} oracle cells, max |GPU - oracle| = {worst:.3e} (scale {scale:.2e})")
    assert worst < 1e-12 * scale
