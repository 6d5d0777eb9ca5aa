"""SURVEY.md §8 row a10 (decompose_stress + eigen_vectors_and_values, cracks.cc:1691-1737, 1923-2120) closed:

* parity of the split assembly at the 1e-12 bar, widened only by what the reference's own formula amplifies when its
  INPUT moves by one ulp (the derivative branch divides by E_01^2 and by the discriminant: cracks.cc:1982-2006);
* IEEE corner cases: exactly diagonal and exactly zero strain run through both branches; the NaN / Inf / finite pattern
  of every output entry must be the oracle's (the reference divides by E[0][1] and by `diskriminante` there, and its
  orthogonality check lets NaN pass because `NaN > 1e-6` is false);
* the reference's abort() (cracks.cc:1732-1736) is reachable with a SYMMETRIC strain: |E_01| just above the
  1e-10 |E_00| threshold loses the eigenvector components to cancellation; the library must report
  PFM_ERR_NOT_ORTHOGONAL through pfm_sync_status, never abort;
* pfm_check_finite raises PFM_ERR_NONFINITE on request.
"""
import numpy as np
import pytest
import scipy.sparse as sp

import cases
import oracle_api as O
from cracks_amd import capi
from cracks_amd import mesh as M
from cracks_amd.assembler import Context
from gpu_util import blocks_to_global, make_context

pytestmark = pytest.mark.gpu


def _oracle(c, residual_only):
    rp = ci = None
    if not residual_only:
        rp, ci = M.dof_sparsity(c.mesh, c.layout)
    r = O.assemble(c.mesh, c.layout, c.params, c.sol, c.old, c.oldold, c.cu, c.ch, residual_only, rp, ci)
    return r, rp, ci


def test_stress_split_parity_at_1e12_within_the_reference_conditioning():
    c = cases.perturbed(cases.kat_miehe_shear_1(), u_amp=2e-3)
    c.params.timestep_number = 1  # cracks.cc:2294, 2338: split active
    r, rp, ci = _oracle(c, False)
    assert r.err == 0
    # conditioning of the reference formula itself: the same oracle on inputs moved by one ulp
    rng = np.random.default_rng(0)
    sol2 = np.where(rng.uniform(size=c.sol.size) < 0.5, np.nextafter(c.sol, np.inf), np.nextafter(c.sol, -np.inf))
    dmask = c.cu.flag.astype(bool)
    sol2[dmask] = c.sol[dmask]
    r2 = O.assemble(c.mesh, c.layout, c.params, sol2, c.old, c.oldold, c.cu, c.ch, False, rp, ci)
    scale_A, scale_R = max(1.0, np.abs(r.values).max()), max(1.0, np.abs(r.residual_pde).max())
    cond_A = np.abs(r2.values - r.values).max() / scale_A
    cond_R = np.abs(r2.residual_pde - r.residual_pde).max() / scale_R
    ctx = make_context(c)
    values, res, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
    A = blocks_to_global(ctx, c.layout, values)
    A.sort_indices()  # scipy's bmat/COO round trip, not the library's order (tests/test_gpu_pattern.py checks that)
    A_ref = sp.csr_matrix((r.values, ci, rp), shape=A.shape)
    assert (A.indices == A_ref.indices).all()
    err_A = np.abs(A.data - A_ref.data).max() / scale_A
    err_R = np.abs(res - r.residual_pde).max() / scale_R
    print(f"split parity: matrix {err_A:.2e} (one-ulp input sensitivity of the reference {cond_A:.2e}), "
          f"residual {err_R:.2e} ({cond_R:.2e})")
    assert err_R < max(1e-12, 4 * cond_R)
    assert err_A < max(1e-12, 4 * cond_A)
    ctx.close()


def _loose_cells(U):
    """K disconnected unit squares [0,1]^2 (own nodes each, all products with the coordinates exact), nodal
    displacements U[k][vertex][comp]; direct-solver layout like the Miehe runs."""
    K = len(U)
    coords = np.tile(np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0]]), (K, 1))
    cells = np.arange(4 * K, dtype=np.int32).reshape(K, 4)
    mesh = M.Mesh(dim=2, coords=coords, cells=cells)
    lay = M.DofLayout(mesh.n_nodes, 2, blocked=False)
    u = np.asarray(U, float).reshape(4 * K, 2)
    return mesh, lay, u


def _miehe_params(**kw):
    d = dict(mu=80.77e3, G_c=2.7, alpha_eps=0.0884, constant_k=1e-10, pressure=0.0, timestep=1e-3, time=2e-3,
             old_timestep=1e-3, old_old_timestep=1e-3, timestep_number=1, decompose_stress_rhs=1.0,
             decompose_stress_matrix=1.0)
    d.update(kw)
    return O.make_params(**{"lambda": 121.15e3}, **d)


def test_split_at_exactly_diagonal_and_zero_strain_reproduces_the_ieee_pattern():
    a, b, s = 2.0 ** -9, 2.0 ** -11, 2.0 ** -10  # powers of two: every product is exact, fused or not
    lin = lambda exx, exy, eyx, eyy: [[0.0, 0.0], [exx, eyx], [exy, eyy], [exx + exy, eyx + eyy]]  # u = G x at the 4 vertices
    U = [lin(0, 0, 0, 0),        # zero strain: eigen_vectors_and_values computes 0/0 (cracks.cc:1716-1721)
         lin(a, 0, 0, 0),        # diagonal, E_11 = 0
         lin(a, 0, 0, -b),       # diagonal, both entries
         lin(0, 0, 0, b),        # diagonal, E_00 = 0
         lin(-a, 0, 0, -a),      # diagonal, equal negative entries: discriminant 0
         lin(0, s, s, 0),        # pure shear: the regular branch with E_00 = E_11 = 0
         lin(a, s, s, -b)]       # generic control cell
    mesh, lay, u = _loose_cells(U)
    rng = np.random.default_rng(3)
    phi = rng.uniform(0.3, 0.9, mesh.n_nodes)
    sol = lay.pack(u, phi)
    old = lay.pack(0 * u, rng.uniform(0.3, 0.9, mesh.n_nodes))
    oo = lay.pack(0 * u, rng.uniform(0.3, 0.9, mesh.n_nodes))
    cu = M.update_constraints(mesh, lay, [])
    ch = M.hanging_constraints(mesh, lay)
    c = cases.Case("corners", mesh, lay, _miehe_params(), sol, old, oo, cu, ch)
    ctx = make_context(c)
    assert ctx.kernel_path in (0, 3)  # a small mesh may have no regular row at all
    for residual_only in (True, False):
        r, rp, ci = _oracle(c, residual_only)
        assert r.err == 0  # NaN passes the reference's orthogonality check (NaN > 1e-6 is false)
        values, res, res_tot = ctx.assemble_host(c.sol, c.old, c.oldold, residual_only)
        pairs = [(res, r.residual_pde)]
        if residual_only:
            pairs.append((res_tot, r.residual_total))
        else:
            A = sp.csr_matrix((values[0],) + ctx.pattern(0)[::-1], shape=(lay.n_dofs,) * 2)
            A_ref = sp.csr_matrix((r.values, ci, rp), shape=A.shape)
            assert (A.indptr == A_ref.indptr).all() and (A.indices == A_ref.indices).all()
            pairs.append((A.data, A_ref.data))
        seen_nan = False
        for got, want in pairs:
            assert (np.isnan(got) == np.isnan(want)).all(), "NaN pattern differs from the reference arithmetic"
            assert (np.isposinf(got) == np.isposinf(want)).all() and (np.isneginf(got) == np.isneginf(want)).all()
            fin = np.isfinite(want)
            seen_nan |= bool(np.isnan(want).any())
            assert np.abs(got[fin] - want[fin]).max() < 1e-12 * max(1.0, np.abs(want[fin]).max())
        assert seen_nan, "the corner cases were meant to produce NaN in the reference arithmetic"
        ctx.sync_status()  # no error status: the reference does not abort here either
    # on request the library says so
    import torch
    t = torch.from_numpy(res).cuda()
    with pytest.raises(capi.PfmError) as e:
        ctx.check_finite(t.data_ptr(), t.numel())
    assert e.value.status == 4  # PFM_ERR_NONFINITE
    ok = torch.ones(1000, dtype=torch.float64, device="cuda")
    ctx.check_finite(ok.data_ptr(), ok.numel())
    ctx.close()


def test_split_with_subnormal_shear_strain_follows_the_reference_arithmetic():
    """E_01 subnormal next to equally tiny E_00, E_11 (not caught by the 'diagonal' test of cracks.cc:1700-1710).
    The reference DIVIDES (lambda_i - E_00) / E_01 -- an ordinary number -- where a reciprocal of E_01 alone overflows to
    inf and turns the eigenvectors into NaN.  In the reference arithmetic the discriminant underflows to zero here, both
    eigenvalues coincide, the two eigenvectors are equal and the orthogonality check fails (abort(), cracks.cc:1732-1736):
    the library must arrive at the same verdict, PFM_ERR_NOT_ORTHOGONAL -- with NaN eigenvectors it would have passed the
    check (NaN > 1e-6 is false) and returned NaN rows.  A second cell set with a subnormal E_01 but ordinary E_00, E_11 is
    'diagonal' for both arithmetics and must agree entry by entry."""
    s0, a, b = 2.0 ** -1060, 2.0 ** -1062, 2.0 ** -1063
    lin = lambda exx, exy, eyx, eyy: [[0.0, 0.0], [exx, eyx], [exy, eyy], [exx + exy, eyx + eyy]]
    rng = np.random.default_rng(5)
    cu_ch = lambda mesh, lay: (M.update_constraints(mesh, lay, []), M.hanging_constraints(mesh, lay))

    def case(U):
        mesh, lay, u = _loose_cells(U)
        sol = lay.pack(u, rng.uniform(0.3, 0.9, mesh.n_nodes))
        old = lay.pack(0 * u, rng.uniform(0.3, 0.9, mesh.n_nodes))
        cu, ch = cu_ch(mesh, lay)
        return cases.Case("subnormal", mesh, lay, _miehe_params(), sol, old, old.copy(), cu, ch)

    # (i) everything tiny: the reference aborts, the library reports
    c = case([lin(a, s0, s0, b), lin(a, s0, s0, -b)])
    r, _, _ = _oracle(c, True)
    assert r.err != 0, "reference arithmetic: coinciding eigenvalues, equal eigenvectors, abort()"
    ctx = make_context(c)
    with pytest.raises(capi.PfmError) as e:
        ctx.assemble_host(c.sol, c.old, c.oldold, True)
    assert e.value.status == 3  # PFM_ERR_NOT_ORTHOGONAL
    ctx.close()
    # (ii) subnormal shear next to ordinary normal strains: the near-diagonal shortcut in both arithmetics
    c = case([lin(2.0 ** -9, s0, s0, -(2.0 ** -11)), lin(2.0 ** -9, 2.0 ** -10, 2.0 ** -10, -(2.0 ** -11))])
    ctx = make_context(c)
    for residual_only in (True, False):
        r, rp, ci = _oracle(c, residual_only)
        assert r.err == 0
        values, res, _ = ctx.assemble_host(c.sol, c.old, c.oldold, residual_only)
        pairs = [(res, r.residual_pde)]
        if not residual_only:
            A = sp.csr_matrix((values[0],) + ctx.pattern(0)[::-1], shape=(c.layout.n_dofs,) * 2)
            pairs.append((A.data, sp.csr_matrix((r.values, ci, rp), shape=A.shape).data))
        for got, want in pairs:
            assert (np.isnan(got) == np.isnan(want)).all() and (np.isinf(got) == np.isinf(want)).all()
            fin = np.isfinite(want)
            assert np.abs(got[fin] - want[fin]).max() < 1e-12 * max(1.0, np.abs(want[fin]).max())
    ctx.sync_status()
    ctx.close()


def test_not_orthogonal_status_instead_of_abort():
    """|E_01| in (1, 1.3) x 1e-10 |E_00| with E_11 = 0.9 .. 0.95 E_00: the 'not close to diagonal' branch loses
    (lambda - E_00) to cancellation and v_1 . v_2 exceeds 1e-6 in about one q-point out of ten (numpy experiment in
    DESIGN.md); 600 independent cells make both the oracle and the GPU hit it."""
    rng = np.random.default_rng(8)
    K = 600
    e00 = rng.uniform(0.5e-3, 2e-3, K)
    e11 = e00 * (1.0 - 0.1 * rng.uniform(0.5, 1.0, K))
    e01 = e00 * rng.uniform(1.0e-10, 1.3e-10, K)
    U = [[[0.0, 0.0], [e00[k], e01[k]], [e01[k], e11[k]], [e00[k] + e01[k], e01[k] + e11[k]]] for k in range(K)]
    mesh, lay, u = _loose_cells(U)
    ones = np.ones(mesh.n_nodes)
    sol = lay.pack(u, 0.7 * ones)
    old = lay.pack(0 * u, 0.6 * ones)
    cu = M.update_constraints(mesh, lay, [])
    ch = M.hanging_constraints(mesh, lay)
    c = cases.Case("ortho", mesh, lay, _miehe_params(), sol, old, old.copy(), cu, ch)
    r, _, _ = _oracle(c, True)
    assert r.err != 0, "the oracle (reference arithmetic) must hit the abort() of cracks.cc:1732-1736 on this input"
    ctx = make_context(c)
    with pytest.raises(capi.PfmError) as e:
        ctx.assemble_host(c.sol, c.old, c.oldold, True)
    assert e.value.status == 3  # PFM_ERR_NOT_ORTHOGONAL
    # the status is reported once and cleared: the same context assembles a harmless state afterwards
    g = np.array([[0.0, 0.0], [1e-3, 4e-4], [4e-4, -2e-3], [1.4e-3, -1.6e-3]])
    ok_sol = lay.pack(np.tile(g, (K, 1)), 0.7 * ones)
    _, res, _ = ctx.assemble_host(ok_sol, c.old, c.oldold, True)
    assert np.isfinite(res).all()
    ctx.close()
