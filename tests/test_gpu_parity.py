"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI, against the
CPU oracle on the same inputs and against the reference's golden Newton-table values.

Tolerance (north_star): residual l_inf error < 1e-12, scaled by max(1, |R|_inf) so that the
Miehe cases (mu ~ 8e4) are held to the same relative bar; matrices are compared entry by
entry, constrained rows/columns and their placeholder diagonals included."""
import numpy as np
import pytest

import cases
import oracle_api as O
from cracks_amd import mesh as M
from gpu_util import blocks_to_global, linf_scaled, make_context

pytestmark = pytest.mark.gpu

TOL = 1e-12


def _oracle(c, residual_only):
    rowptr = colind = None
    if not residual_only:
        rowptr, colind = M.dof_sparsity(c.mesh, c.layout)
    r = O.assemble(c.mesh, c.layout, c.params, c.sol, c.old, c.oldold, c.cu, c.ch, residual_only,
                   rowptr, colind, c.cell_lambda, c.cell_mu)
    assert r.err == 0
    return r, rowptr, colind


@pytest.mark.parametrize("make", cases.ALL_KATS, ids=lambda f: f.__name__)
def test_step0_residual_golden_through_the_abi(make):
    c = make()
    ctx = make_context(c)
    _, res_pde, res_tot = ctx.assemble_host(c.sol, c.old, c.oldold, residual_only=True)
    norm = np.linalg.norm(c.cu.set_zero(res_pde))
    assert norm == pytest.approx(c.golden_residual0, rel=5e-7)
    r, _, _ = _oracle(c, True)
    assert linf_scaled(res_pde, r.residual_pde) < TOL
    assert linf_scaled(res_tot, r.residual_total) < TOL


def _full_parity(c, tol=TOL):
    import scipy.sparse as sp

    ctx = make_context(c)
    values, res_pde, _ = ctx.assemble_host(c.sol, c.old, c.oldold, residual_only=False)
    r, rowptr, colind = _oracle(c, False)
    A_ref = sp.csr_matrix((r.values, colind, rowptr), shape=(c.layout.n_dofs,) * 2)
    A = blocks_to_global(ctx, c.layout, values)
    # identical pattern (the library's canonical pattern == make_sparsity_pattern stand-in)
    A.sort_indices()
    assert A.nnz == A_ref.nnz and (A.indptr == A_ref.indptr).all() and (A.indices == A_ref.indices).all()
    assert linf_scaled(A.data, A_ref.data) < tol
    assert linf_scaled(res_pde, r.residual_pde) < tol
    _, res_pde2, res_tot2 = ctx.assemble_host(c.sol, c.old, c.oldold, residual_only=True)
    r2, _, _ = _oracle(c, True)
    assert linf_scaled(res_pde2, r2.residual_pde) < tol
    assert linf_scaled(res_tot2, r2.residual_total) < tol
    return ctx


@pytest.mark.parametrize("make", cases.ALL_KATS, ids=lambda f: f.__name__)
def test_full_assembly_matches_oracle(make):
    c = make()
    if c.name == "sneddon_3d":
        c = cases.kat_sneddon_3d(5)
    _full_parity(cases.perturbed(c))


@pytest.mark.parametrize("blocked", [False, True])
@pytest.mark.parametrize("dim", [2, 3])
def test_both_layouts(dim, blocked):
    c = cases.perturbed(cases.kat_sneddon_2d() if dim == 2 else cases.kat_sneddon_3d(4))
    lay = M.DofLayout(c.mesh.n_nodes, dim, blocked)
    node, comp = c.layout.node_comp_of_dof()

    def conv(v):
        out = np.empty_like(v)
        out[lay.dof(node, comp)] = v
        return out

    ch = M.hanging_constraints(c.mesh, lay)
    cu = M.update_constraints(c.mesh, lay, M.sneddon_dirichlet_dofs(c.mesh, lay))
    c2 = cases.Case(c.name, c.mesh, lay, c.params, conv(c.sol), conv(c.old), conv(c.oldold), cu, ch)
    _full_parity(c2)


def test_active_set_lines_and_monolithic_penalty():
    c = cases.perturbed(cases.kat_sneddon_2d())
    # put a few phase-field dofs into the active set (cracks.cc:2878-2879)
    node, comp = c.layout.node_comp_of_dof()
    phi_dofs = np.nonzero((comp == 2) & ~c.ch.flag.astype(bool))[0]
    active = phi_dofs[::7]
    c.cu = M.update_constraints(c.mesh, c.layout, M.sneddon_dirichlet_dofs(c.mesh, c.layout), active)
    _full_parity(c)
    # simple monolithic: penalisation terms + residual_total through constraints_update
    c.params.outer_solver = 1
    c.params.gamma_penal = 10.0
    c.params.timestep_number = 2
    _full_parity(c)


def test_hanging_nodes_3d():
    m = M.box_mesh(3, 4)
    x = m.coords[m.cells].mean(axis=1)
    m = M.refine_cells(m, (np.abs(x) < 5.0).all(axis=1))
    assert m.hn_nodes.size > 0
    lay = M.DofLayout(m.n_nodes, 3, True)
    base = cases.kat_sneddon_3d(4)
    h = m.min_cell_diameter()
    phi = M.initial_values_sneddon(m, h)
    ch = M.hanging_constraints(m, lay)
    cu = M.update_constraints(m, lay, M.sneddon_dirichlet_dofs(m, lay))
    sol = ch.distribute(lay.pack(np.zeros((m.n_nodes, 3)), phi))
    c = cases.Case("hang3d", m, lay, base.params, sol, sol.copy(), sol.copy(), cu, ch)
    c.params.alpha_eps = 2 * h
    _full_parity(cases.perturbed(c))


def test_per_cell_material():
    c = cases.perturbed(cases.kat_sneddon_3d(4))
    rng = np.random.default_rng(5)
    E = 1.0 + rng.uniform(1.0, 10.0, c.mesh.n_cells)  # func_emodulus + 1.0 (cracks.cc:2209-2210)
    c.cell_mu = E / (2.0 * (1 + 0.2))
    c.cell_lambda = (2 * 0.2 * c.cell_mu) / (1.0 - 2 * 0.2)
    _full_parity(c)


def test_non_cartesian_cells():
    c = cases.perturbed(cases.kat_sneddon_3d(4))
    rng = np.random.default_rng(9)
    interior = np.ones(c.mesh.n_nodes, bool)
    for nodes in c.mesh.boundary_nodes.values():
        interior[nodes] = False
    c.mesh.coords[interior] += rng.uniform(-0.6, 0.6, (interior.sum(), 3))
    c.mesh.box_shape = None
    _full_parity(c)


def test_non_orthogonal_eigenvectors_are_reported_not_aborted():
    c = cases.perturbed(cases.kat_miehe_shear_1(), u_amp=2e-3)
    c.params.timestep_number = 1
    r, _, _ = _oracle(c, True)
    assert r.err == 0  # generic data is fine ...
    # ... and the status word path works: a 3-D split is refused up front
    c3 = cases.kat_sneddon_3d(4)
    c3.params.timestep_number = 1
    c3.params.decompose_stress_matrix = 1.0
    from cracks_amd.capi import PfmError
    with pytest.raises(PfmError) as ei:
        make_context(c3)
    assert ei.value.status == 5


def test_device_resident_assembler_matches_host_entry():
    import torch
    from cracks_amd.assembler import Assembler, node_flags_from_dof_flags

    c = cases.perturbed(cases.kat_sneddon_3d(5))
    asm = Assembler(c.mesh, c.layout.blocked)
    asm.set_params(c.params)
    asm.set_constraints(node_flags_from_dof_flags(c.layout, c.cu.flag, c.ch.flag))
    asm.set_vectors(c.sol, c.old, c.oldold)
    asm.assemble_system()
    asm.synchronize()
    ctx = make_context(c)
    values, res_pde, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
    for b in range(4):
        assert linf_scaled(asm.system_pde_matrix[b].cpu().numpy(), values[b]) < 1e-13
    assert linf_scaled(asm.system_pde_residual.cpu().numpy(), res_pde) < 1e-13
    asm.assemble_nl_residual()
    asm.synchronize()
    r, _, _ = _oracle(c, True)
    assert linf_scaled(asm.system_total_residual.cpu().numpy(), r.residual_total) < TOL
    assert torch.isfinite(asm.system_pde_residual).all()


@pytest.mark.parametrize("dim", [2, 3])
def test_rank_without_cells_or_owned_rows(dim):
    """Empty inputs as a partition can produce them: a rank whose local mesh has nodes but no cells (all values and
    residuals zero), and a rank that owns no node at all (nothing to write, every call still succeeds)."""
    from cracks_amd.assembler import Assembler
    import torch

    g = M.box_mesh(dim, (3,) * dim)
    empty = M.Mesh(dim=dim, coords=g.coords.copy(), cells=np.zeros((0, 1 << dim), np.int32), boundary_nodes={})
    a = Assembler(empty, blocked=True)
    a.set_params(cases.kat_sneddon_3d(4).params if dim == 3 else cases.kat_sneddon_2d().params)
    a.set_constraints(np.zeros(empty.n_nodes, np.uint8))
    n = empty.n_nodes * (dim + 1)
    a.set_vectors(np.linspace(0, 1, n), np.zeros(n), np.zeros(n))
    a.system_pde_residual.fill_(float("nan"))
    for ro in (False, True):
        a.assemble_system(ro)
        a.synchronize()
        assert float(a.system_pde_residual.abs().max()) == 0.0
    assert all(m.numel() == 0 or float(m.abs().max()) == 0.0 for m in a.system_pde_matrix)

    b = Assembler(g, blocked=True, n_owned_nodes=0)
    b.set_params(cases.kat_sneddon_3d(4).params if dim == 3 else cases.kat_sneddon_2d().params)
    b.set_constraints(np.zeros(g.n_nodes, np.uint8))
    assert b.solution.numel() == 0
    for ro in (False, True):
        b.assemble_system(ro)
        b.synchronize()
    assert all(m.numel() == 0 for m in b.system_pde_matrix)
    torch.cuda.synchronize()
