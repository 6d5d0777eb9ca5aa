"""Cartesian overlay of general 3-D meshes (round 5; cracks.cc:3895-4163 refine_mesh is dimension independent, the
reference's own 3-D AMR case is tests/hetero_3d_1.prm): the regular rows of every refinement level are written by the
row-owner kernels of the cartesian family running on that level's lattice (k_cart_uu3, k_cart_phi4, k_cart_residual3),
the general family keeps the rows at hanging nodes, level seams and the boundary.  Every entry against the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

import cases
import oracle_api as O
from cracks_amd import mesh as M
from gpu_util import blocks_to_global, linf_scaled, make_context

pytestmark = pytest.mark.gpu
TOL = 1e-12


def refined_block_case(n, blocked, lo=-10.0, hi=10.0, het=False, seed=5, active=True):
    """A box of n coarse cells per direction whose inner half is refined once: hanging nodes on the faces and edges of the
    block, two level lattices with regular rows."""
    g0 = M.box_mesh(3, n, lo, hi)
    c = g0.coords[g0.cells].mean(axis=1)
    mid = 0.5 * (np.asarray(lo, float) + np.asarray(hi, float)) * np.ones(3)
    half = 0.25 * (np.asarray(hi, float) - np.asarray(lo, float)) * np.ones(3)
    mesh = M.refine_cells(g0, (np.abs(c - mid) < half).all(axis=1))
    h = mesh.min_cell_diameter()
    lay = M.DofLayout(mesh.n_nodes, 3, blocked)
    prm = O.PfmParams.from_buffer_copy(bytes(cases.kat_sneddon_3d(4).params))
    prm.alpha_eps = 2.0 * h
    prm.constant_k = 1e-8 * h
    phi = M.initial_values_sneddon(mesh, h)
    sol = lay.pack(np.zeros((mesh.n_nodes, 3)), phi)
    ch = M.hanging_constraints(mesh, lay)
    dirichlet = list(M.sneddon_dirichlet_dofs(mesh, lay))
    if active:  # an active set in the refined block and outside of it (cracks.cc:2878-2879: phase-field dofs with inhomogeneity 0)
        rng = np.random.default_rng(seed)
        node, comp = lay.node_comp_of_dof()
        free_phi = np.nonzero((comp == 3) & ~ch.flag.astype(bool))[0]
        dirichlet += list(rng.choice(free_phi, size=free_phi.size // 9, replace=False))
    cu = M.update_constraints(mesh, lay, dirichlet)
    case = cases.perturbed(cases.Case("refined_block", mesh, lay, prm, sol, sol.copy(), sol.copy(), cu, ch), seed=seed)
    if het:
        rng = np.random.default_rng(seed + 1)
        E = rng.uniform(0.5, 3.0, mesh.n_cells)
        case.cell_mu = E / (2.0 * 1.2)
        case.cell_lambda = 2.0 * 0.2 * case.cell_mu / (1.0 - 0.4)
    return case


def check_against_oracle(c, ctx):
    rp, ci = M.dof_sparsity(c.mesh, c.layout)
    r = O.assemble(c.mesh, c.layout, c.params, c.sol, c.old, c.oldold, c.cu, c.ch, False, rp, ci, c.cell_lambda, c.cell_mu)
    assert r.err == 0
    values, res, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
    A = blocks_to_global(ctx, c.layout, values)
    A.sort_indices()
    A_ref = sp.csr_matrix((r.values, ci, rp), shape=A.shape)
    A_ref.sort_indices()
    assert (A.indptr == A_ref.indptr).all() and (A.indices == A_ref.indices).all()
    assert linf_scaled(A.data, A_ref.data) < TOL
    assert linf_scaled(res, r.residual_pde) < TOL
    ro = O.assemble(c.mesh, c.layout, c.params, c.sol, c.old, c.oldold, c.cu, c.ch, True, None, None, c.cell_lambda, c.cell_mu)
    _, res_pde, res_tot = ctx.assemble_host(c.sol, c.old, c.oldold, True)
    assert linf_scaled(res_pde, ro.residual_pde) < TOL and linf_scaled(res_tot, ro.residual_total) < TOL


@pytest.mark.parametrize("blocked", [True, False])
@pytest.mark.parametrize("n,lo,hi", [((12, 12, 12), -10.0, 10.0), ((20, 8, 12), (-2.0, 0.0, 1.0), (3.0, 2.0, 2.5))])
def test_refined_block_through_the_overlay(n, lo, hi, blocked):
    c = refined_block_case(n, blocked, lo, hi)
    ctx = make_context(c)
    assert ctx.kernel_path == 3
    rows, general_cells = ctx.overlay_info()
    assert rows > 0.2 * c.mesh.n_nodes and 0 < general_cells < c.mesh.n_cells
    check_against_oracle(c, ctx)
    # the same context through the general family alone: the two paths agree with the oracle independently
    ctx.force_path(0)
    check_against_oracle(c, ctx)
    ctx.close()


def test_refined_block_with_heterogeneous_material_and_monolithic_scheme():
    c = refined_block_case((12, 10, 12), True, het=True)
    ctx = make_context(c)
    assert ctx.kernel_path == 3
    check_against_oracle(c, ctx)
    ctx.close()
    c = refined_block_case((12, 10, 12), False)
    c.params.outer_solver = 1
    c.params.gamma_penal = 7.0
    c.params.timestep_number = 3
    c.params.time, c.params.timestep, c.params.old_timestep, c.params.old_old_timestep = 2.3, 0.5, 0.7, 0.4
    ctx = make_context(c)
    assert ctx.kernel_path == 3
    check_against_oracle(c, ctx)
    ctx.close()


def test_complete_rows_in_permuted_slots_take_the_blocked_copy_out():
    """Round 6: interior tiles of a level lattice whose rows are complete (27 neighbours, no constraint flag near the plane)
    but sit in the AMR mesh's own order go through k_cart_phi4's blocked copy-out with looked-up destinations.  No active
    set here, so that such tiles exist (16 coarse cells per direction: the fine lattice has 33 nodes, four full tiles of 7 with
    a margin to the seam); every entry against the oracle, canonical and bound pattern."""
    from test_gpu_pattern import _permuted_patterns

    c = refined_block_case((16, 16, 16), True, active=False)
    ctx = make_context(c)
    assert ctx.kernel_path == 3
    check_against_oracle(c, ctx)
    for b, (rp, ci, _) in enumerate(_permuted_patterns(ctx, 3, True, seed=11)):
        ctx.pattern_bind(b, rp, ci)
    check_against_oracle(c, ctx)
    ctx.close()


def test_overlay_rows_follow_a_bound_pattern():
    """pfm_pattern_bind with shuffled rows: the CSR slots of the regular rows are looked up again in the bound order."""
    from test_gpu_pattern import _permuted_patterns

    c = refined_block_case((12, 12, 12), True, active=False)
    ctx = make_context(c)
    for b, (rp, ci, _) in enumerate(_permuted_patterns(ctx, 3, True, seed=2)):
        ctx.pattern_bind(b, rp, ci)
    assert ctx.kernel_path == 3
    check_against_oracle(c, ctx)
    ctx.close()


@pytest.mark.parametrize("mode", ["default: scratch + ordered gather", "PFM_HANGING_COLOURED", "PFM_HANGING_ATOMIC"])
@pytest.mark.parametrize("blocked", [True, False])
def test_cells_at_hanging_vertices_modes(blocked, mode, monkeypatch):
    """The three ways the hexes at hanging vertices reach the outputs (read by pfm_ctx_create):
    default (round 6): their reduced element matrices / residuals go to per-cell scratch and k_hanging_gather adds them row
    by row in list order; PFM_HANGING_COLOURED=1 (round 5): plain colour classes over the constraint-resolved nodes;
    PFM_HANGING_ATOMIC=1: the class with FP64 atomics.  Same entries as the oracle in every mode, with the overlay and through
    the general family alone, heterogeneous material included -- and in the first two, assemblies of the same state agree in
    every bit, which the atomic class cannot promise."""
    if mode.startswith("PFM_"):
        monkeypatch.setenv(mode, "1")
    bitwise = mode != "PFM_HANGING_ATOMIC"
    c = refined_block_case((12, 10, 12), blocked, het=True)
    ctx = make_context(c)
    assert ctx.kernel_path == 3
    check_against_oracle(c, ctx)
    for path in (None, 0):
        if path is not None:
            ctx.force_path(path)
            check_against_oracle(c, ctx)
        first = ctx.assemble_host(c.sol, c.old, c.oldold, False)
        for rep in range(3):
            again = ctx.assemble_host(c.sol, c.old, c.oldold, False)
            if bitwise:
                for x, y in zip(first[0], again[0]):
                    assert np.array_equal(x, y)
                assert np.array_equal(first[1], again[1])
        r1 = ctx.assemble_host(c.sol, c.old, c.oldold, True)
        r2 = ctx.assemble_host(c.sol, c.old, c.oldold, True)
        if bitwise:
            assert np.array_equal(r1[1], r2[1]) and np.array_equal(r1[2], r2[2])
    ctx.close()
