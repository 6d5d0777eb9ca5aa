"""Pin the CPU oracle against every fixture the reference's own tests hold for the hot
path: the step-0 Newton residual norms of five regression goldens and the
six Catch cases of eigen_vectors_and_values (cracks.cc:1740-1919)."""
import numpy as np
import pytest

import cases
import oracle_api as O


@pytest.mark.parametrize("make", cases.ALL_KATS, ids=lambda f: f.__name__)
def test_step0_residual_matches_reference_golden(make):
    c = make()
    r = O.assemble(c.mesh, c.layout, c.params, c.sol, c.old, c.oldold, c.cu, c.ch, residual_only=True,
                   cell_lambda=c.cell_lambda, cell_mu=c.cell_mu)
    assert r.err == 0
    res = c.cu.set_zero(r.residual_pde)  # cracks.cc:2793
    norm = np.linalg.norm(res)
    # goldens are printed with 7 significant digits (std::scientific, precision 6)
    assert norm == pytest.approx(c.golden_residual0, rel=5e-7), (norm, c.golden_residual0)


def test_kat_mesh_sizes_match_goldens():
    g = cases.golden()
    c = cases.kat_sneddon_2d()
    assert c.mesh.n_cells == g["sneddon_2d_1"]["timesteps"][0]["cells"] == 124
    assert c.layout.n_dofs == g["sneddon_2d_1"]["timesteps"][0]["dofs"] == 453
    assert c.mesh.hn_nodes.size == 12
    c = cases.kat_sneddon_3d()
    assert c.layout.n_dofs == g["sneddon_3d_1.mpirun=4"]["timesteps"][0]["dofs"] == 5324
    c = cases.kat_hetero_3d()
    gh = g["hetero_3d_1.mpirun-4"]
    assert c.mesh.n_cells == gh["timesteps"][0]["cells"] == 932
    assert c.layout.n_dofs == gh["timesteps"][0]["dofs"] == 5288
    assert c.mesh.min_cell_diameter() == pytest.approx(gh["params"]["h (min)"], rel=1e-5)
    c = cases.kat_miehe_shear_1()
    assert c.mesh.n_cells == 256 and c.layout.n_dofs == 891
    assert c.params.alpha_eps == pytest.approx(g["miehe_shear_1"]["params"]["eps"], rel=1e-5)
    assert c.params.constant_k == pytest.approx(g["miehe_shear_1"]["params"]["k"], rel=1e-5)


# ---- the six Catch TEST_CASEs, cracks.cc:1740-1919 -------------------------------------
def _vecs(ev):
    return ev[:, 0], ev[:, 1]


def test_eigen_diagonal():
    err, e1, e2, ev = O.eigen_2x2([[2.0, 0.0], [0.0, 3.0]])
    v1, v2 = _vecs(ev)
    assert err == 0
    assert (e1, e2) == (pytest.approx(2.0), pytest.approx(3.0))
    assert np.allclose(v1, [1, 0]) and np.allclose(v2, [0, 1])


@pytest.mark.parametrize("m00", [-2.0, 5.0])
def test_eigen_11_zero(m00):
    err, e1, e2, ev = O.eigen_2x2([[m00, 0.0], [0.0, 0.0]])
    v1, v2 = _vecs(ev)
    assert err == 0
    assert e1 == pytest.approx(m00) and e2 == pytest.approx(0.0)
    assert np.allclose(v1, [1, 0]) and np.allclose(v2, [0, 1])


def test_eigen_offdiagonal_only():
    err, e1, e2, ev = O.eigen_2x2([[0.0, -2.0], [-2.0, 0.0]])
    v1, v2 = _vecs(ev)
    sq = np.sqrt(2.0)
    assert err == 0
    assert e1 == pytest.approx(2.0) and e2 == pytest.approx(-2.0)
    assert np.allclose(v1, [1 / sq, -1 / sq]) and np.allclose(v2, [1 / sq, 1 / sq])


def test_eigen_full():
    err, e1, e2, ev = O.eigen_2x2([[3.0, 2.0], [2.0, 4.0]])
    v1, v2 = _vecs(ev)
    a, b = 7.0 / 2.0, np.sqrt(17) / 2.0
    assert err == 0
    assert e1 == pytest.approx(a + b) and e2 == pytest.approx(a - b)
    w1 = (-0.5 + b) / 2.0
    l1 = np.sqrt(w1 * w1 + 1.0)
    assert np.allclose(v1, [w1 / l1, 1.0 / l1])
    w2 = (-0.5 - b) / 2.0
    l2 = np.sqrt(w2 * w2 + 1.0)
    assert np.allclose(v2, [-w2 / l2, -1.0 / l2])


def test_eigen_00_zero():
    err, e1, e2, ev = O.eigen_2x2([[0.0, -2.0], [-2.0, 4.0]])
    v1, v2 = _vecs(ev)
    assert err == 0
    assert e1 == pytest.approx(2.0 + 2.0 * np.sqrt(2.0)) and e2 == pytest.approx(2.0 - 2.0 * np.sqrt(2.0))
    w1 = 1.0 - np.sqrt(2.0)
    l1 = np.sqrt(w1 * w1 + 1.0)
    assert np.allclose(v1, [-w1 / l1, -1.0 / l1])
    w2 = 1.0 + np.sqrt(2.0)
    l2 = np.sqrt(w2 * w2 + 1.0)
    assert np.allclose(v2, [w2 / l2, 1.0 / l2])
