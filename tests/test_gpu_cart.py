"""GPU parity tests of the cartesian (row-owner) kernel family against the CPU oracle and
against the general family, on uniform and anisotropic boxes, both dof layouts."""
import numpy as np
import pytest
import scipy.sparse as sp

import cases
import oracle_api as O
from cracks_amd import mesh as M
from gpu_util import blocks_to_global, linf_scaled, make_context

pytestmark = pytest.mark.gpu
TOL = 1e-12


def box_case(dim, n, lo, hi, blocked, seed=3, monolithic=False):
    mesh = M.box_mesh(dim, n, lo, hi)
    h = mesh.min_cell_diameter()
    lay = M.DofLayout(mesh.n_nodes, dim, blocked)
    base = cases.kat_sneddon_3d(4) if dim == 3 else cases.kat_sneddon_2d()
    prm = O.PfmParams.from_buffer_copy(bytes(base.params))
    prm.alpha_eps = 2.0 * h
    prm.constant_k = 1e-8 * h
    phi = M.initial_values_sneddon(mesh, h)
    sol = lay.pack(np.zeros((mesh.n_nodes, dim)), phi)
    ch = M.hanging_constraints(mesh, lay)
    cu = M.update_constraints(mesh, lay, M.sneddon_dirichlet_dofs(mesh, lay))
    c = cases.perturbed(cases.Case("box", mesh, lay, prm, sol, sol.copy(), sol.copy(), cu, ch), seed=seed)
    if monolithic:
        c.params.outer_solver = 1
        c.params.gamma_penal = 7.0
        c.params.timestep_number = 3
        c.params.time, c.params.timestep, c.params.old_timestep, c.params.old_old_timestep = 2.3, 0.5, 0.7, 0.4
    return c


def oracle(c, residual_only):
    rp = ci = None
    if not residual_only:
        rp, ci = M.dof_sparsity(c.mesh, c.layout)
    r = O.assemble(c.mesh, c.layout, c.params, c.sol, c.old, c.oldold, c.cu, c.ch, residual_only, rp, ci,
                   c.cell_lambda, c.cell_mu)
    assert r.err == 0
    return r, rp, ci


# (5, 4, 61): 62 node planes = several z-chunks of the marching kernels (start-up layer, carry across steps,
# chunk seams); (16, 9, 30): more than one tile in x and y with partial tiles at the high faces
BOXES = [(3, (6, 6, 6), -10.0, 10.0), (3, (9, 5, 11), (-1.0, 0.0, 2.0), (2.0, 1.5, 2.7)),
         (3, (5, 4, 61), -10.0, 10.0), (3, (16, 9, 30), (-2.0, 0.0, 0.0), (2.0, 3.0, 5.0)),
         (2, (12, 7), (-3.0, 1.0), (1.0, 2.0)), (2, (17, 17), -10.0, 10.0)]


@pytest.mark.parametrize("blocked", [True, False])
@pytest.mark.parametrize("dim,n,lo,hi", BOXES)
@pytest.mark.parametrize("monolithic", [False, True])
def test_cart_residual_matches_oracle(dim, n, lo, hi, blocked, monolithic):
    c = box_case(dim, n, lo, hi, blocked, monolithic=monolithic)
    ctx = make_context(c)
    assert ctx.kernel_path == 1, "uniform box must select the cartesian family"
    _, res_pde, res_tot = ctx.assemble_host(c.sol, c.old, c.oldold, True)
    r, _, _ = oracle(c, True)
    assert linf_scaled(res_pde, r.residual_pde) < TOL
    assert linf_scaled(res_tot, r.residual_total) < TOL
    ctx.force_path(0)
    _, g_pde, g_tot = ctx.assemble_host(c.sol, c.old, c.oldold, True)
    assert linf_scaled(res_pde, g_pde) < TOL and linf_scaled(res_tot, g_tot) < TOL


def _full(c, path):
    ctx = make_context(c)
    ctx.force_path(path)
    values, res_pde, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
    r, rp, ci = oracle(c, False)
    A_ref = sp.csr_matrix((r.values, ci, rp), shape=(c.layout.n_dofs,) * 2)
    A = blocks_to_global(ctx, c.layout, values)
    A.sort_indices()
    assert (A.indptr == A_ref.indptr).all() and (A.indices == A_ref.indices).all()
    err = linf_scaled(A.data, A_ref.data)
    assert err < TOL, err
    assert linf_scaled(res_pde, r.residual_pde) < TOL


@pytest.mark.parametrize("blocked", [True, False])
@pytest.mark.parametrize("dim,n,lo,hi", [b for b in BOXES if b[0] == 3] + [(3, (17, 10, 3), -10.0, 10.0)])
def test_cart_uu_overlay_matches_oracle(dim, n, lo, hi, blocked):
    _full(box_case(dim, n, lo, hi, blocked), path=2)


def test_cart_uu_overlay_monolithic_and_active_set():
    c = box_case(3, (7, 6, 5), -10.0, 10.0, True, monolithic=True)
    node, comp = c.layout.node_comp_of_dof()
    phi_dofs = np.nonzero(comp == 3)[0]
    c.cu = M.update_constraints(c.mesh, c.layout, M.sneddon_dirichlet_dofs(c.mesh, c.layout), phi_dofs[::5])
    _full(c, path=2)


def test_non_lattice_meshes_use_the_general_family_with_the_cartesian_overlay():
    """2-D meshes that are not one lexicographic box (slit with duplicated nodes, hanging nodes): kernel path 3 = the
    general family for the rows next to the irregularities + the patch kernel for the rows of regular lattice nodes."""
    c = cases.kat_miehe_shear_1()  # duplicated nodes along the slit
    ctx = make_context(c)
    assert ctx.kernel_path == 3
    rows, cells = ctx.overlay_info()
    assert 0 < rows < c.mesh.n_nodes and 0 < cells < c.mesh.n_cells
    c = cases.kat_sneddon_2d()  # hanging nodes
    ctx = make_context(c)
    assert ctx.kernel_path == 3
    rows, cells = ctx.overlay_info()
    assert 0 < rows < c.mesh.n_nodes and 0 < cells < c.mesh.n_cells
    ctx.force_path(0)
    assert ctx.kernel_path == 0


@pytest.mark.parametrize("blocked", [True, False])
@pytest.mark.parametrize("dim,n,lo,hi", [b for b in BOXES if b[0] == 3] + [(3, (17, 10, 3), -10.0, 10.0)])
def test_cart_full_matrix_matches_oracle(dim, n, lo, hi, blocked):
    _full(box_case(dim, n, lo, hi, blocked), path=1)


def test_cart_full_monolithic_and_active_set():
    c = box_case(3, (7, 6, 5), -10.0, 10.0, True, monolithic=True)
    node, comp = c.layout.node_comp_of_dof()
    phi_dofs = np.nonzero(comp == 3)[0]
    c.cu = M.update_constraints(c.mesh, c.layout, M.sneddon_dirichlet_dofs(c.mesh, c.layout), phi_dofs[::5])
    _full(c, path=1)
    c.params.use_old_timestep_pf = 1
    _full(c, path=1)


def test_cart_full_vanishing_degradation_uses_mean_diagonal():
    """kappa = 0 and pf_extra = 0 in whole cells: the (u,u) element diagonal is exactly zero and
    deal.II's placeholder for constrained rows falls back to the mean |diagonal| of the element."""
    c = box_case(3, (6, 5, 4), -10.0, 10.0, True)
    c.params.constant_k = 0.0
    node, comp = c.layout.node_comp_of_dof()
    is_phi = comp == 3
    x = c.mesh.coords[node[is_phi]]
    dead = x[:, 0] < 0.0  # half of the domain fully broken in the two old steps, reaching the boundary
    o = c.old.copy()
    o[np.nonzero(is_phi)[0][dead]] = 0.0
    c.old, c.oldold = o, o.copy()
    _full(c, path=1)


@pytest.mark.parametrize("blocked", [True, False])
def test_general_family_3d_mean_diagonal_monolithic_and_active_set(blocked):
    """The hex kernel of the general family (32 lanes per hex, the q-point states shared out through LDS): the element's mean
    |diagonal| for constrained rows is a sum over the lanes that hold the diagonal blocks; active-set lines, the penalisation
    of the monolithic scheme and a box that is not a cube go through the same lanes."""
    c = box_case(3, (6, 5, 4), -10.0, 10.0, blocked)
    c.params.constant_k = 0.0
    node, comp = c.layout.node_comp_of_dof()
    is_phi = comp == 3
    dead = c.mesh.coords[node[is_phi]][:, 0] < 0.0
    o = c.old.copy()
    o[np.nonzero(is_phi)[0][dead]] = 0.0
    c.old, c.oldold = o, o.copy()
    _full(c, path=0)
    c = box_case(3, (7, 6, 5), (-2.0, 0.0, 0.0), (2.0, 3.0, 5.0), blocked, monolithic=True)
    node, comp = c.layout.node_comp_of_dof()
    c.cu = M.update_constraints(c.mesh, c.layout, M.sneddon_dirichlet_dofs(c.mesh, c.layout), np.nonzero(comp == 3)[0][::5])
    _full(c, path=0)


def test_cart_is_the_default_full_path_for_boxes():
    c = box_case(3, (5, 5, 5), -10.0, 10.0, True)
    ctx = make_context(c)
    assert ctx.kernel_path == 1
    values, res_pde, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
    ctx.force_path(0)
    values_g, res_g, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
    for a, b in zip(values, values_g):
        assert linf_scaled(a, b) < TOL
    assert linf_scaled(res_pde, res_g) < TOL


@pytest.mark.parametrize("dim,n", [(3, (1, 1, 1)), (3, (2, 1, 3)), (3, (1, 8, 1)), (2, (1, 1)), (2, (3, 1))])
def test_cart_degenerate_boxes(dim, n):
    """Smallest lattices: every node is a boundary node, tiles and z-chunks are mostly empty."""
    c = box_case(dim, n, -1.0, 1.0, True)
    ctx = make_context(c)
    _, res_pde, res_tot = ctx.assemble_host(c.sol, c.old, c.oldold, True)
    r, _, _ = oracle(c, True)
    assert linf_scaled(res_pde, r.residual_pde) < TOL
    assert linf_scaled(res_tot, r.residual_total) < TOL
    _full(c, path=ctx.kernel_path)


def test_cart_assembly_is_bitwise_reproducible():
    """Row-owner kernels: no atomics on global memory, fixed summation order => identical bits on every run."""
    c = box_case(3, (16, 9, 30), (-2.0, 0.0, 0.0), (2.0, 3.0, 5.0), True)
    ctx = make_context(c)
    assert ctx.kernel_path == 1
    v0, r0, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
    for _ in range(3):
        v1, r1, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
        assert all(np.array_equal(a, b) for a, b in zip(v0, v1))
        assert np.array_equal(r0, r1)


@pytest.mark.parametrize("blocked", [True, False])
@pytest.mark.parametrize("kappa_zero", [False, True])
def test_cart_random_constraints_stress(blocked, kappa_zero):
    """Random homogeneous constraints on displacement and phase-field dofs (interior ones included), pressure and
    Biot terms on, kappa = 0 (placeholder fall-back reachable): every matrix entry and both residuals."""
    c = box_case(3, (23, 9, 31), (-1.0, 0.0, 0.0), (1.5, 1.0, 4.0), blocked, seed=11)
    rng = np.random.default_rng(5)
    lay = c.layout
    node, comp = lay.node_comp_of_dof()
    pick = np.where(comp == 3, rng.random(lay.n_dofs) < 0.2, rng.random(lay.n_dofs) < 0.05)
    dd = np.union1d(M.sneddon_dirichlet_dofs(c.mesh, lay), np.nonzero(pick)[0])
    c.cu = M.update_constraints(c.mesh, lay, dd)
    c.params.pressure = 3.0e-3
    if kappa_zero:
        c.params.constant_k = 0.0
        # a patch of exactly vanishing old phase field: g(q) = 0 in whole cells
        far = c.mesh.coords[:, 2] > 3.0
        for v in (c.old, c.oldold):
            v[lay.dof(np.nonzero(far)[0], 3)] = 0.0
    ctx = make_context(c)
    assert ctx.kernel_path == 1
    _, res_pde, res_tot = ctx.assemble_host(c.sol, c.old, c.oldold, True)
    r, _, _ = oracle(c, True)
    assert linf_scaled(res_pde, r.residual_pde) < TOL
    assert linf_scaled(res_tot, r.residual_total) < TOL
    _full(c, path=1)


@pytest.mark.parametrize("blocked", [True, False])
def test_cart_overwrites_every_output(blocked):
    """`system_pde_matrix = 0` (cracks.cc:2133-2137) is part of the assembly: every value of every block, the
    structurally zero (u,phi) block included, and both residuals are written whatever the buffers held before."""
    import torch
    from cracks_amd.assembler import Assembler

    c = box_case(3, (16, 9, 30), (-2.0, 0.0, 0.0), (2.0, 3.0, 5.0), blocked)
    asm = Assembler(c.mesh, blocked=blocked)
    assert asm.ctx.kernel_path == 1
    asm.set_params(c.params)
    from cracks_amd.assembler import node_flags_from_dof_flags
    asm.set_constraints(node_flags_from_dof_flags(c.layout, c.cu.flag, c.ch.flag))
    asm.set_vectors(c.sol, c.old, c.oldold)
    asm.allocate_matrix()
    outs = []
    for fill in (float("nan"), 0.0):
        for m in asm.system_pde_matrix:
            m.fill_(fill)
        asm.system_pde_residual.fill_(fill)
        asm.assemble_system(False)
        asm.synchronize()
        outs.append([m.clone() for m in asm.system_pde_matrix] + [asm.system_pde_residual.clone()])
    for a, b in zip(*outs):
        assert not torch.isnan(a).any()
        assert torch.equal(a, b)
    if blocked:
        assert float(outs[0][1].abs().max()) == 0.0


@pytest.mark.parametrize("blocked", [True, False])
def test_cart_matches_general_family_on_a_medium_box(blocked):
    """Two independent implementations (row-owner kernels vs cell-owner kernels with atomics) on a box with many
    tiles, partial tiles on both high faces and several z-chunks: 40 x 33 x 70 cells, every value of every block."""
    c = box_case(3, (40, 33, 70), (-2.0, 0.0, 0.0), (2.0, 3.0, 5.0), blocked, seed=5, monolithic=True)
    ctx = make_context(c)
    assert ctx.kernel_path == 1
    v1, r1, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
    _, rp1, rt1 = ctx.assemble_host(c.sol, c.old, c.oldold, True)
    ctx.force_path(0)
    v0, r0, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
    _, rp0, rt0 = ctx.assemble_host(c.sol, c.old, c.oldold, True)
    for a, b in zip(v1, v0):
        assert linf_scaled(a, b) < TOL
    assert linf_scaled(r1, r0) < TOL and linf_scaled(rp1, rp0) < TOL and linf_scaled(rt1, rt0) < TOL


@pytest.mark.parametrize("dim,n", [(3, (9, 5, 11)), (2, (12, 7))])
def test_cart_residual_with_old_timestep_phase_field(dim, n):
    """use_old_timestep_pf (cracks.cc:2273-2276: pf_extra = phi_old, unclamped) on the staggered (non-monolithic) path,
    where the 3-D residual kernel interpolates one combined old field instead of two."""
    c = box_case(dim, n, -1.0, 2.0, True, seed=9)
    c.params.use_old_timestep_pf = 1
    c.params.time, c.params.timestep, c.params.old_timestep, c.params.old_old_timestep = 2.3, 0.5, 0.7, 0.4
    ctx = make_context(c)
    assert ctx.kernel_path == 1
    _, res_pde, res_tot = ctx.assemble_host(c.sol, c.old, c.oldold, True)
    r, _, _ = oracle(c, True)
    assert linf_scaled(res_pde, r.residual_pde) < TOL and linf_scaled(res_tot, r.residual_total) < TOL
    c.params.use_old_timestep_pf = 0  # extrapolated pf_extra with unequal time steps (tfac != 1), clamped to [0, 1]
    ctx = make_context(c)
    _, res_pde, res_tot = ctx.assemble_host(c.sol, c.old, c.oldold, True)
    r, _, _ = oracle(c, True)
    assert linf_scaled(res_pde, r.residual_pde) < TOL and linf_scaled(res_tot, r.residual_total) < TOL


def test_jacobian_pair_sequential_and_residual_kernel_agree(tmp_path):
    """The same 3-D assembly in its three launch modes: default (the (u,u) and the phase-field kernel next to each other on
    two streams, residual from the matrix rows, deferred placeholder patches), PFM_JAC_SEQUENTIAL=1 (one after the other)
    and PFM_RES_KERNEL=1 (quadrature residual kernel) -- on a box with several tiles, partial tiles, z-chunks, constraint
    flags and both layouts.  The mode is chosen when the library is first used, hence one process each."""
    import os
    import subprocess
    import sys

    script = tmp_path / "run.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path[:0] = [{os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}, {os.path.dirname(os.path.abspath(__file__))!r}]\n"
        "import test_gpu_cart as T\n"
        "from gpu_util import make_context\n"
        "out, rhs = [], []\n"
        "for blocked in (True, False):\n"
        "    for n in ((19, 9, 40), (6, 6, 6)):\n"
        "        c = T.box_case(3, n, -10.0, 10.0, blocked)\n"
        "        ctx = make_context(c)\n"
        "        values, res, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)\n"
        "        out.append(values[0])\n"
        "        rhs.append(res)\n"
        "np.save(sys.argv[1], np.concatenate(out))\n"
        "np.save(sys.argv[2], np.concatenate(rhs))\n")
    val, rhs = {}, {}
    for tag, env in (("rows", {}), ("rows_seq", {"PFM_JAC_SEQUENTIAL": "1"}), ("quad", {"PFM_RES_KERNEL": "1"})):
        f, g = tmp_path / f"{tag}.npy", tmp_path / f"{tag}_rhs.npy"
        e = {k: v for k, v in os.environ.items() if k not in ("PFM_RES_KERNEL", "PFM_JAC_SEQUENTIAL")}
        e.update(env)
        subprocess.run([sys.executable, str(script), str(f), str(g)], check=True, env=e, timeout=600)
        val[tag], rhs[tag] = np.load(f), np.load(g)
    # side by side or one after the other: the same kernels, the same bits (run-to-run reproducibility)
    assert np.array_equal(val["rows"], val["rows_seq"]) and np.array_equal(rhs["rows"], rhs["rows_seq"])
    # the quadrature residual against the residual from the rows: round-off apart; the matrix to round-off as well
    assert val["rows"].shape == val["quad"].shape and linf_scaled(val["rows"], val["quad"]) < TOL
    assert rhs["rows"].shape == rhs["quad"].shape and linf_scaled(rhs["rows"], rhs["quad"]) < TOL


# ---- 2-D row-owner Jacobian (pfm_cart2d.hip): BASELINE config 2 with the matrix, tests/sneddon_2d_1.prm on a uniform mesh
BOXES_2D = [(2, (12, 7), (-3.0, 1.0), (1.0, 2.0)), (2, (17, 17), -10.0, 10.0), (2, (130, 5), -10.0, 10.0), (2, (1, 9), 0.0, 1.0)]


@pytest.mark.parametrize("blocked", [True, False])
@pytest.mark.parametrize("dim,n,lo,hi", BOXES_2D)
@pytest.mark.parametrize("monolithic", [False, True])
def test_cart2d_full_matrix_matches_oracle(dim, n, lo, hi, blocked, monolithic):
    c = box_case(dim, n, lo, hi, blocked, monolithic=monolithic)
    ctx = make_context(c)
    assert ctx.kernel_path == 1
    _full(c, path=1)


@pytest.mark.parametrize("blocked", [True, False])
@pytest.mark.parametrize("kappa_zero", [False, True])
def test_cart2d_random_constraints_and_placeholders(blocked, kappa_zero):
    """Random homogeneous constraints (interior dofs included), pressure on; kappa = 0 with g = 0 in whole cells: the
    constrained rows must take deal.II's mean-|diagonal| placeholder (the slow path of the 2-D kernel)."""
    c = box_case(2, (41, 23), (-1.0, 0.0), (1.5, 1.0), blocked, seed=7)
    rng = np.random.default_rng(9)
    lay = c.layout
    node, comp = lay.node_comp_of_dof()
    pick = np.where(comp == 2, rng.random(lay.n_dofs) < 0.2, rng.random(lay.n_dofs) < 0.08)
    dd = np.union1d(M.sneddon_dirichlet_dofs(c.mesh, lay), np.nonzero(pick)[0])
    c.cu = M.update_constraints(c.mesh, lay, dd)
    c.params.pressure = 3.0e-3
    if kappa_zero:
        c.params.constant_k = 0.0
        far = c.mesh.coords[:, 1] > 0.6
        for v in (c.old, c.oldold):
            v[lay.dof(np.nonzero(far)[0], 2)] = 0.0
    _full(c, path=1)
    # overwrite semantics + bitwise reproducibility of the row-owner kernel
    ctx = make_context(c)
    v0, r0, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
    v1, r1, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
    assert all(np.array_equal(a, b) for a, b in zip(v0, v1)) and np.array_equal(r0, r1)


def test_cart2d_split_runs_stay_on_the_general_family():
    """decompose_stress_matrix > 0 and timestep_number > 0 on a 2-D lattice: correct through the general family."""
    c = box_case(2, (9, 8), 0.0, 1.0, False)
    c.params.decompose_stress_matrix = 1.0
    c.params.decompose_stress_rhs = 1.0
    c.params.timestep_number = 2
    ctx = make_context(c)
    values, res_pde, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
    r, rp, ci = oracle(c, False)
    A_ref = sp.csr_matrix((r.values, ci, rp), shape=(c.layout.n_dofs,) * 2)
    A = blocks_to_global(ctx, c.layout, values)
    A.sort_indices()
    assert linf_scaled(A.data, A_ref.data) < 1e-11 and linf_scaled(res_pde, r.residual_pde) < 1e-11


# ---- heterogeneous material on the row-owner kernels (cracks.cc:2207-2216): per-cell Lame coefficients, cells handed
# over in a shuffled order (the kernels address them by lattice position)
def heterogeneous(c, seed=5, free_ratio=False):
    rng = np.random.default_rng(seed)
    c.mesh.cells = np.ascontiguousarray(c.mesh.cells[rng.permutation(c.mesh.n_cells)])
    E = 1.0 + rng.uniform(1.0, 10.0, c.mesh.n_cells)  # func_emodulus + 1.0 (cracks.cc:2209-2210)
    c.cell_mu = E / (2.0 * (1 + 0.2))
    c.cell_lambda = (2 * 0.2 * c.cell_mu) / (1.0 - 2 * 0.2)
    if free_ratio:  # the ABI takes any pair per cell
        c.cell_lambda = rng.uniform(0.05, 4.0, c.mesh.n_cells)
    return c


HET_BOXES = [(3, (9, 5, 11), (-1.0, 0.0, 2.0), (2.0, 1.5, 2.7)), (3, (16, 9, 30), (-2.0, 0.0, 0.0), (2.0, 3.0, 5.0)),
             (2, (12, 7), (-3.0, 1.0), (1.0, 2.0)), (2, (33, 18), -10.0, 10.0)]


@pytest.mark.parametrize("blocked", [True, False])
@pytest.mark.parametrize("dim,n,lo,hi", HET_BOXES)
def test_cart_heterogeneous_material_full(dim, n, lo, hi, blocked):
    c = heterogeneous(box_case(dim, n, lo, hi, blocked), free_ratio=(n[0] == 9))
    assert make_context(c).kernel_path == 1, "a box with per-cell Lame coefficients stays on the cartesian family"
    _full(c, path=1)


@pytest.mark.parametrize("dim,n,lo,hi", HET_BOXES)
@pytest.mark.parametrize("monolithic", [False, True])
def test_cart_heterogeneous_material_residual(dim, n, lo, hi, monolithic):
    c = heterogeneous(box_case(dim, n, lo, hi, True, monolithic=monolithic), seed=11)
    ctx = make_context(c)
    assert ctx.kernel_path == 1
    _, res_pde, res_tot = ctx.assemble_host(c.sol, c.old, c.oldold, True)
    r, _, _ = oracle(c, True)
    assert linf_scaled(res_pde, r.residual_pde) < TOL
    assert linf_scaled(res_tot, r.residual_total) < TOL


def test_cart_heterogeneous_material_mean_diagonal_and_constraints():
    c = box_case(3, (6, 5, 4), -10.0, 10.0, True, monolithic=True)
    c.params.constant_k = 0.0
    node, comp = c.layout.node_comp_of_dof()
    is_phi = comp == 3
    dead = c.mesh.coords[node[is_phi]][:, 0] < 0.0
    o = c.old.copy()
    o[np.nonzero(is_phi)[0][dead]] = 0.0
    c.old, c.oldold = o, o.copy()
    phi_dofs = np.nonzero(is_phi)[0]
    c.cu = M.update_constraints(c.mesh, c.layout, M.sneddon_dirichlet_dofs(c.mesh, c.layout), phi_dofs[::5])
    _full(heterogeneous(c, free_ratio=True), path=1)


def test_state_set_solution_only_keeps_the_old_fields():
    """pfm_state_set_solution (the line search of cracks.cc:2942-2957: only `solution` changes between two
    assemble_nl_residual calls): the residual after a solution-only scatter equals the one of a context that was given
    all three vectors, bit for bit, and differs from the residual of the previous solution."""
    from cracks_amd import capi

    c = box_case(3, (11, 9, 20), -10.0, 10.0, True)
    ctx = make_context(c)
    _, r0, t0 = ctx.assemble_host(c.sol, c.old, c.oldold, True)
    rng = np.random.default_rng(3)
    sol2 = c.sol + 1e-3 * rng.standard_normal(c.sol.shape)
    a = np.ascontiguousarray(sol2, dtype=np.float64)
    rc = ctx.lib.pfm_state_set_solution(ctx._h, capi.np_ptr(a, np.float64), 0)
    assert rc == capi.PFM_OK
    import torch

    n = c.layout.n_dofs
    bufs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(2)]
    ctx.assemble_device(True, [], bufs[0].data_ptr(), bufs[1].data_ptr())
    ctx.sync_status()
    r1, t1 = bufs[0].cpu().numpy(), bufs[1].cpu().numpy()
    ref = make_context(c)
    _, r2, t2 = ref.assemble_host(sol2, c.old, c.oldold, True)
    assert np.array_equal(r1, r2) and np.array_equal(t1, t2)
    assert not np.array_equal(r1, r0)


@pytest.mark.parametrize("dim,n", [(3, (11, 9, 20)), (2, (70, 33))])
@pytest.mark.parametrize("blocked", [True, False])
def test_assemble_nl_residual_device_reads_the_solution_itself(dim, n, blocked):
    """pfm_assemble_nl_residual_device (cracks.cc:2942-2957, 2507-2512: solution += delta; assemble_nl_residual()): on a
    single-rank box the residual kernel reads `solution` itself instead of a scatter launch in front of it.  Same bits as
    pfm_state_set_solution + pfm_assemble_device, and the node state is left as the scatter would have left it: a
    full assembly that follows WITHOUT another scatter matches the oracle at the new solution."""
    import torch

    c = box_case(dim, n, -10.0, 10.0, blocked)
    ctx = make_context(c)
    ctx.assemble_host(c.sol, c.old, c.oldold, True)  # all three vectors once
    rng = np.random.default_rng(5)
    sol2 = c.sol + 1e-3 * rng.standard_normal(c.sol.shape)
    nd = c.layout.n_dofs
    d_sol = torch.from_numpy(np.ascontiguousarray(sol2)).cuda()
    bufs = [torch.empty(nd, dtype=torch.float64, device="cuda") for _ in range(2)]
    ctx.assemble_nl_residual_device(d_sol.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr())
    ctx.sync_status()
    r1, t1 = bufs[0].cpu().numpy(), bufs[1].cpu().numpy()
    ref = make_context(c)
    _, r2, t2 = ref.assemble_host(sol2, c.old, c.oldold, True)
    assert np.array_equal(r1, r2) and np.array_equal(t1, t2)
    # the state: a Jacobian assembled now (no scatter in between) belongs to sol2
    ctx.set_stream(0)
    nb = ctx.n_blocks
    vals = [torch.empty(ctx.pattern_size(b)[1], dtype=torch.float64, device="cuda") for b in range(nb)]
    ctx.assemble_device(False, [v.data_ptr() for v in vals], bufs[0].data_ptr(), bufs[1].data_ptr())
    ctx.sync_status()
    c2 = c
    c2.sol = sol2
    r, rp, ci = oracle(c2, False)
    A_ref = sp.csr_matrix((r.values, ci, rp), shape=(nd,) * 2)
    A = blocks_to_global(ctx, c.layout, [v.cpu().numpy() for v in vals])
    A.sort_indices()
    assert linf_scaled(A.data, A_ref.data) < TOL and linf_scaled(bufs[0].cpu().numpy(), r.residual_pde) < TOL


@pytest.mark.parametrize("n,het", [((15, 15, 2), False), ((16, 31, 5), False), ((14, 29, 3), False), ((33, 16, 26), False),
                                   ((45, 46, 7), False), ((17, 30, 6), True)])
def test_residual_kernels_with_planes_by_transfer_agree(n, het, monkeypatch):
    """k_cart_residual3x (blocked solution vector read in place, 16-byte global -> LDS transfers two planes ahead),
    k_cart_residual3d (dword transfers, any layout) and k_cart_residual3 <true> (planes through registers) form the same
    sums in the same order: same bits, on boxes whose last tile holds 1, 2 or 15 nodes, with one or several z-chunks, and in
    the workgroups that hold the first / the last node of the lattice (dword fallback).  The node state the 3x kernel
    publishes is the solution: a residual from the state alone (no scatter) is the same again."""
    import torch

    c = box_case(3, n, -10.0, 10.0, True)
    if het:  # per-cell Lame coefficients: the <HET> instances of the two transfer kernels
        c = heterogeneous(c)
    ctx = make_context(c)
    ctx.assemble_host(c.sol, c.old, c.oldold, True)
    rng = np.random.default_rng(11)
    sol2 = c.sol + 1e-3 * rng.standard_normal(c.sol.shape)
    nd = c.layout.n_dofs
    d_sol = torch.from_numpy(np.ascontiguousarray(sol2)).cuda()

    def run():
        bufs = [torch.full((nd,), 7.0, dtype=torch.float64, device="cuda") for _ in range(2)]
        ctx.assemble_nl_residual_device(d_sol.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr())
        ctx.sync_status()
        return bufs[0].cpu().numpy(), bufs[1].cpu().numpy()

    rx, tx = run()
    # the published state: the plain residual-only assembly reads it
    bufs = [torch.empty(nd, dtype=torch.float64, device="cuda") for _ in range(2)]
    ctx.assemble_device(True, [], bufs[0].data_ptr(), bufs[1].data_ptr())
    ctx.sync_status()
    assert np.array_equal(bufs[0].cpu().numpy(), rx) and np.array_equal(bufs[1].cpu().numpy(), tx)
    monkeypatch.setenv("PFM_RES_NO_WIDE_TRANSFERS", "1")
    rd, td = run()
    monkeypatch.setenv("PFM_RES_NO_TRANSFERS", "1")
    ro, to = run()
    assert np.array_equal(rx, rd) and np.array_equal(tx, td)
    assert np.array_equal(rx, ro) and np.array_equal(tx, to)
    c.sol = sol2
    r, _, _ = oracle(c, True)
    assert linf_scaled(rx, r.residual_pde) < TOL and linf_scaled(tx, r.residual_total) < TOL


@pytest.mark.parametrize("blocked", [True, False])
def test_cart2d_launch_variants_agree(blocked, monkeypatch):
    """k_cart2d_cells: one launch per row group (default; the mean |diagonal| of a block with constrained rows goes from
    the first launch to the second through CartView::cell_avg, the structurally zero (u,phi) block is cleared by a fill
    in front of them), the same without the fill, and the single launch of round 5 write the same bits -- on a box with
    Dirichlet lines, an active set and dead zones where diagonal entries vanish (the placeholder needs the mean)."""
    c = box_case(2, (45, 31), -10.0, 10.0, blocked, monolithic=True)
    c.params.constant_k = 0.0
    node, comp = c.layout.node_comp_of_dof()
    is_phi = comp == 2
    dead = c.mesh.coords[node[is_phi]][:, 0] < 0.0
    o = c.old.copy()
    o[np.nonzero(is_phi)[0][dead]] = 0.0
    c.old, c.oldold = o, o.copy()
    phi_dofs = np.nonzero(is_phi)[0]
    c.cu = M.update_constraints(c.mesh, c.layout, M.sneddon_dirichlet_dofs(c.mesh, c.layout), phi_dofs[::5])
    ctx = make_context(c)
    assert ctx.kernel_path == 1

    def run():
        vals, rp, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
        return [np.array(v, copy=True) for v in vals], rp.copy()

    v0, p0 = run()
    for key in ("PFM_CART2D_NO_FILL", "PFM_CART2D_ONE_LAUNCH"):
        monkeypatch.setenv(key, "1")
        v1, p1 = run()
        assert all(np.array_equal(a, b) for a, b in zip(v0, v1)) and np.array_equal(p0, p1), key
    monkeypatch.delenv("PFM_CART2D_NO_FILL")
    monkeypatch.delenv("PFM_CART2D_ONE_LAUNCH")
    _full(c, path=1)


def test_bench_reports_the_config5_standin():
    """bench.py's `extra.config5_standin`: the adaptive stress-split stand-in runs through the general
    family + cartesian overlay (kernel path 3) and the record carries the times, the roofline fraction and the rebuild."""
    import bench

    rec = bench.config5_standin("cuda:0", 0, steps=3, levels=5)
    assert rec["kernel_path"] == 3
    assert rec["overlay"]["rows_of_the_patch_kernel"] > 0.5 * rec["dofs"] / 3
    for key in ("jacobian_ms", "residual_only_ms", "context_rebuild_ms", "jacobian_hbm_frac"):
        assert rec[key] > 0.0
    assert rec["hanging_nodes"] > 0


def test_bench_reports_the_general_family_3d():
    """bench.py's `extra.general_family_3d`: a hex box forced onto the general family (what a 3-D mesh with hanging nodes
    costs) stays on kernel path 0 and reports both assembly modes."""
    import bench

    rec = bench.general_family_3d("cuda:0", 0, steps=4, n=24)
    assert "kernel path 0" in rec["workload"] and rec["cells"] == 24 ** 3
    for key in ("jacobian", "residual_only"):
        assert rec[key]["ms_per_call"] > 0.0 and rec[key]["kernel_ms"] > 0.0 and rec[key]["hbm_frac"] > 0.0


def test_general_family_launch_modes_agree(tmp_path):
    """The same 2-D assembly of a mesh with hanging nodes and the stress split in its launch modes: default (cartesian
    overlay: patch kernel on the stream, the rest of the general family next to it), PFM_NO_PATCH=1 (general family alone, the
    atomic class on the side stream), PFM_NO_PATCH=1 PFM_GENERAL_SEQUENTIAL=1 (one class after the other).  Rows away from
    hanging nodes are bitwise equal between the two general-family modes (one colour class at a time writes them); everything
    agrees to round-off.  The modes are chosen when the library is first used / the context is made: one process each."""
    import os
    import subprocess
    import sys

    script = tmp_path / "run.py"
    here = os.path.dirname(os.path.abspath(__file__))
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path[:0] = [{os.path.dirname(here)!r}, {here!r}]\n"
        "import cases\n"
        "from gpu_util import make_context\n"
        "c = cases.perturbed(cases.kat_miehe_tension())  # the reference's adaptive Miehe tension mesh: hanging nodes, slit, split\n"
        "c.params.timestep_number = 1\n"
        "ctx = make_context(c)\n"
        "values, res, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)\n"
        "np.save(sys.argv[1], np.concatenate([np.ravel(x) for x in values]))\n"
        "np.save(sys.argv[2], res)\n"
        "open(sys.argv[3], 'w').write(str(ctx.kernel_path))\n")
    val, rhs, path = {}, {}, {}
    modes = (("overlay", {}), ("general", {"PFM_NO_PATCH": "1"}), ("general_seq", {"PFM_NO_PATCH": "1", "PFM_GENERAL_SEQUENTIAL": "1"}))
    for tag, env in modes:
        f, g, h = tmp_path / f"{tag}.npy", tmp_path / f"{tag}_rhs.npy", tmp_path / f"{tag}.txt"
        e = {k: v for k, v in os.environ.items() if k not in ("PFM_NO_PATCH", "PFM_GENERAL_SEQUENTIAL")}
        e.update(env)
        subprocess.run([sys.executable, str(script), str(f), str(g), str(h)], check=True, env=e, timeout=600)
        val[tag], rhs[tag], path[tag] = np.load(f), np.load(g), int(open(h).read())
    assert path["general"] == 0 and path["general_seq"] == 0 and path["overlay"] in (0, 3)
    for tag in ("general", "overlay"):
        assert val[tag].shape == val["general_seq"].shape
        assert linf_scaled(val[tag], val["general_seq"]) < TOL and linf_scaled(rhs[tag], rhs["general_seq"]) < TOL
    # side stream or not: only the entries that receive atomic adds (rows next to hanging nodes) may differ in the last bit
    same = val["general"] == val["general_seq"]
    assert same.mean() > 0.5


@pytest.mark.parametrize("blocked", [True, False])
@pytest.mark.parametrize("registered", [True, False])
def test_host_pointer_call_with_the_hosts_own_arrays(blocked, registered):
    """pfm_assemble into arrays the host keeps (the value arrays of its matrix): page-locked through pfm_host_register
    the (u,phi) block is cleared once on the host and never transferred, unregistered it is copied like the others;
    either way the call leaves every entry of every block as the oracle has it -- on the first call, where the arrays hold
    garbage, and on later ones (cracks.cc:2754, 2770, 2918: the caller reads the host storage right after the call)."""
    c = box_case(3, (9, 6, 12), -10.0, 10.0, blocked)
    r, rp, ci = oracle(c, False)
    ctx = make_context(c)
    n = ctx.n_owned_dofs
    values = [np.full(ctx.pattern_size(b)[1], -7.0e77) for b in range(ctx.n_blocks)]
    res_pde, res_tot = np.full(n, -7.0e77), np.full(n, -7.0e77)
    if registered:
        for a in values + [res_pde, res_tot]:
            ctx.host_register(a)
    for rep in range(2):
        sol = c.sol if rep == 0 else c.sol * (1.0 + 1e-3)
        ref = r if rep == 0 else O.assemble(c.mesh, c.layout, c.params, sol, c.old, c.oldold, c.cu, c.ch, False, rp, ci)
        ctx.assemble_host(sol, c.old, c.oldold, False, out=(values, res_pde, None))
        A = blocks_to_global(ctx, c.layout, values)
        A.sort_indices()
        A_ref = sp.csr_matrix((ref.values, ci, rp), shape=A.shape)
        assert (A.indptr == A_ref.indptr).all() and (A.indices == A_ref.indices).all()
        assert np.abs(A.data).max() < 1e70 and linf_scaled(A.data, A_ref.data) < TOL
        assert linf_scaled(res_pde, ref.residual_pde) < TOL
        if blocked:
            assert not values[1].any()  # (u,phi) = 0, cracks.cc:2333-2337
    ctx.assemble_host(c.sol, c.old, c.oldold, True, out=(None, res_pde, res_tot))
    ro = O.assemble(c.mesh, c.layout, c.params, c.sol, c.old, c.oldold, c.cu, c.ch, True)
    assert linf_scaled(res_pde, ro.residual_pde) < TOL and linf_scaled(res_tot, ro.residual_total) < TOL
    if registered:
        ctx.host_unregister(values[0])
        ctx.host_unregister()
    ctx.close()
