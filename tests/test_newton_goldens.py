"""End-to-end parity: the reference's own outer loop, restated in
cracks_amd/newton.py, around the oracle reproduces the Newton tables, energies and loads of
tests/miehe_shear_2.output — the fixtures that pin the Jacobian, the active-set constraints
and the stress split (time steps >= 1)."""
import numpy as np
import pytest

import cases
import newton_cases as NC
from cracks_amd.newton import ActiveSetDriver


def check_against_golden(records, golden_steps, n_steps):
    """What is comparable: the line-0 residual of every step (a function of the previous
    converged state), the first Newton row (physical active set), convergence, energies and
    load.  Rows whose residual is ~1e-13 and the active sets of a converged, still intact
    specimen are decided by round-off in the last bit (phi == old_phi up to 1e-16,
    cracks.cc:2868) and differ between any two correct implementations; the reference's own
    np=1 / np=2 goldens of this case differ for the same reason."""
    for rec, g in zip(records[:n_steps], golden_steps[:n_steps]):
        assert rec.residual0 == pytest.approx(g["residual0"], rel=5e-5)
        g1, r1 = g["newton"][0], rec.newton[0]
        assert r1.residual == pytest.approx(g1["residual"], rel=3e-2 if rec.timestep == 0 else 3e-3)
        if rec.timestep > 0:  # same order of magnitude; the exact count is round-off (see docstring)
            assert abs(r1.active_set - g1["active_set"]) <= 0.2 * g1["active_set"]
        assert rec.newton[-1].residual < 1e-6 and len(rec.newton) <= len(g["newton"]) + 2
        assert rec.bulk_energy == pytest.approx(g["bulk_energy"], rel=5e-5)
        assert rec.crack_energy == pytest.approx(g["crack_energy"], rel=1e-2 if rec.timestep == 1 else 2e-4)


LOAD_X = [32.4555, 64.8685, 97.1969]  # tests/miehe_shear_2.output, "Load x:" of steps 0..2


def test_miehe_shear_2_first_steps_with_oracle():
    setup = NC.miehe_shear_2_setup()
    drv = ActiveSetDriver(setup, NC.OracleAssembler(setup.mesh, setup.layout))
    recs = drv.run(n_steps=6)
    g = cases.golden()["miehe_shear_2"]["timesteps"]
    check_against_golden(recs, g, 6)
    for rec, want in zip(recs, LOAD_X):
        assert rec.load == pytest.approx(want, rel=2e-6)
    # step 1 (stress split active): rows 1-3 of the reference table are reproduced digit for digit
    g1 = g[1]["newton"]
    assert [r.active_set for r in recs[1].newton[:4]] == [x["active_set"] for x in g1[:4]]
    assert recs[1].newton[0].residual == pytest.approx(g1[0]["residual"], rel=2e-6)


@pytest.mark.gpu
def test_miehe_shear_2_first_steps_on_gpu():
    from cracks_amd.newton import GpuAssembler

    setup = NC.miehe_shear_2_setup()
    drv = ActiveSetDriver(setup, GpuAssembler(setup.mesh, setup.layout))
    recs = drv.run(n_steps=4)
    g = cases.golden()["miehe_shear_2"]["timesteps"]
    check_against_golden(recs, g, 4)
    for rec, want in zip(recs, LOAD_X):
        assert rec.load == pytest.approx(want, rel=2e-6)


def _check_sneddon_3d(recs):
    """tests/sneddon_3d_1.mpirun=4.output: energies after every time step.  The reference solves
    the linear systems with GMRES to 1e-8 |r|, the harness exactly, so intermediate Newton
    residuals differ; converged functionals agree."""
    g = cases.golden()["sneddon_3d_1.mpirun=4"]["timesteps"]
    for rec, gg in zip(recs, g):
        assert rec.residual0 == pytest.approx(gg["residual0"], rel=2e-5)
        assert rec.bulk_energy == pytest.approx(gg["bulk_energy"], rel=2e-4)
        assert rec.crack_energy == pytest.approx(gg["crack_energy"], rel=2e-5)
    # the converged active set of step 0 (the three crack nodes) is physical and must match
    assert recs[0].newton[-1].active_set == g[0]["newton"][-1]["active_set"] == 3


def test_sneddon_3d_with_oracle():
    setup = NC.sneddon_3d_setup()
    recs = ActiveSetDriver(setup, NC.OracleAssembler(setup.mesh, setup.layout)).run(n_steps=2)
    _check_sneddon_3d(recs)


@pytest.mark.gpu
def test_sneddon_3d_on_gpu_cartesian_family():
    from cracks_amd.newton import GpuAssembler

    setup = NC.sneddon_3d_setup()
    asm = GpuAssembler(setup.mesh, setup.layout)
    assert asm.ctx.kernel_path == 1
    recs = ActiveSetDriver(setup, asm).run(n_steps=3)
    _check_sneddon_3d(recs)


def _check_sneddon_2d(recs):
    """tests/sneddon_2d_1.output (BASELINE config 0): line-0 residual, energies and the converged active set of every
    time step.  The reference solves with GMRES to a relative tolerance, the harness exactly: intermediate rows differ,
    converged quantities agree to the printed digits."""
    g = cases.golden()["sneddon_2d_1"]["timesteps"]
    assert len(recs) == 3
    for rec, gg in zip(recs, g):
        assert rec.residual0 == pytest.approx(gg["residual0"], rel=2e-6)
        assert rec.bulk_energy == pytest.approx(gg["bulk_energy"], rel=2e-5)
        assert rec.crack_energy == pytest.approx(gg["crack_energy"], rel=2e-6)
        assert rec.newton[-1].residual < 1e-7
        assert rec.newton[-1].active_set == gg["newton"][-1]["active_set"]  # 9, 85, 115


def test_sneddon_2d_with_oracle():
    setup = NC.sneddon_2d_setup()
    _check_sneddon_2d(ActiveSetDriver(setup, NC.OracleAssembler(setup.mesh, setup.layout)).run(n_steps=3))


@pytest.mark.gpu
def test_sneddon_2d_on_gpu_end_to_end():
    """SURVEY.md 8(f) N1 widened: the reference's sneddon_2d_1 run -- hanging nodes, iterative-solver
    layout, active set -- with every assembly on the GPU (general family + cartesian overlay)."""
    from cracks_amd.newton import GpuAssembler

    setup = NC.sneddon_2d_setup()
    asm = GpuAssembler(setup.mesh, setup.layout)
    assert asm.ctx.kernel_path in (0, 3)
    _check_sneddon_2d(ActiveSetDriver(setup, asm).run(n_steps=3))


LOAD_X_SHEAR_1 = [64.911, 129.566, 193.813, 257.017]  # tests/miehe_shear_1.output, "Load x:" of steps 0..3


def _check_miehe_shear_1(recs):
    """Line-0 residual, energies, load and the number of Newton rows of every step; the first Newton row within 2 % (its
    active set is decided by round-off where phi == old_phi to the last bit, see check_against_golden)."""
    g = cases.golden()["miehe_shear_1"]["timesteps"]
    assert len(recs) == 4
    for rec, gg, want in zip(recs, g, LOAD_X_SHEAR_1):
        assert rec.residual0 == pytest.approx(gg["residual0"], rel=5e-6)
        assert rec.bulk_energy == pytest.approx(gg["bulk_energy"], rel=5e-6)
        assert rec.crack_energy == pytest.approx(gg["crack_energy"], rel=5e-6)
        assert rec.load == pytest.approx(want, rel=3e-6)
        assert len(rec.newton) == len(gg["newton"])  # 4, 5, 4, 4 Newton rows as in the reference's table
        assert rec.newton[0].residual == pytest.approx(gg["newton"][0]["residual"], rel=2e-2)
        assert rec.newton[-1].residual < 1e-6


def test_miehe_shear_1_first_steps_with_oracle():
    setup = NC.miehe_shear_1_setup()
    _check_miehe_shear_1(ActiveSetDriver(setup, NC.OracleAssembler(setup.mesh, setup.layout)).run(n_steps=4))


@pytest.mark.gpu
def test_miehe_shear_1_first_steps_on_gpu():
    from cracks_amd.newton import GpuAssembler

    setup = NC.miehe_shear_1_setup()
    _check_miehe_shear_1(ActiveSetDriver(setup, GpuAssembler(setup.mesh, setup.layout)).run(n_steps=4))


def _check_hetero_3d(recs, last_rows_exact=True):
    """tests/hetero_3d_1.mpirun-4.output: per-cell E modulus, 3-D hanging nodes, pressure 1e3 * time.  The reference solves with
    GMRES to a relative tolerance (its Newton residuals stall at 1e-5 ... 1e-8), the harness exactly: line-0 residuals to
    the printed digits, energies to 1e-4, the active sets of step 1 row by row.
    last_rows_exact=False (the overlay path): the last rows of the table are decided by the last bit of a few borderline dofs
    (phi - phi_old ~ 0), and the atomic class of the general family is not bitwise reproducible run to run.  Through the
    general family alone the reference's rows come out in 16 of 16 runs; with the regular rows from the cartesian kernels
    (values that differ by 1e-15) 2 of 16 runs stop one iteration earlier, at 534 instead of 520 active dofs, converged
    (residual < 1e-6) and with the same energies.  The overlay run therefore pins the first three rows and the converged
    state, the general-family run the whole table."""
    g = cases.golden()["hetero_3d_1.mpirun-4"]["timesteps"]
    assert len(recs) == 2
    for rec, gg in zip(recs, g):
        assert rec.residual0 == pytest.approx(gg["residual0"], rel=2e-6)
        assert rec.bulk_energy == pytest.approx(gg["bulk_energy"], rel=1e-4)
        assert rec.crack_energy == pytest.approx(gg["crack_energy"], rel=2e-5)
        assert rec.newton[-1].residual < 1e-6
        if last_rows_exact:
            assert rec.newton[-1].active_set == gg["newton"][-1]["active_set"]  # 36, 520
        else:
            assert abs(rec.newton[-1].active_set - gg["newton"][-1]["active_set"]) <= 0.03 * gg["newton"][-1]["active_set"] + 1
    want = [x["active_set"] for x in g[1]["newton"]]  # 839 639 534 520 520
    got = [r.active_set for r in recs[1].newton]
    if last_rows_exact:
        assert got == want
    else:
        assert got[:3] == want[:3] and len(got) <= len(want)


def test_hetero_3d_with_oracle():
    setup, cl, cm = NC.hetero_3d_setup()
    _check_hetero_3d(ActiveSetDriver(setup, NC.OracleAssembler(setup.mesh, setup.layout, cl, cm)).run(n_steps=2))


@pytest.mark.gpu
def test_hetero_3d_on_gpu_end_to_end():
    """3-D AMR mesh: hanging nodes, per-cell Lame coefficients, time-dependent pressure; energies by pfm_functionals.  Since
    round 5 the regular rows of both refinement levels come from the cartesian row-owner kernels on the level lattices
    (kernel path 3: general family + cartesian overlay), the rest from the general family."""
    from cracks_amd.newton import GpuAssembler

    setup, cl, cm = NC.hetero_3d_setup()
    asm = GpuAssembler(setup.mesh, setup.layout, cl, cm)
    assert asm.ctx.kernel_path == 3 and asm.ctx.overlay_info()[0] > 0
    _check_hetero_3d(ActiveSetDriver(setup, asm).run(n_steps=2), last_rows_exact=False)


@pytest.mark.gpu
def test_hetero_3d_on_gpu_through_the_general_family_alone():
    from cracks_amd.newton import GpuAssembler

    setup, cl, cm = NC.hetero_3d_setup()
    asm = GpuAssembler(setup.mesh, setup.layout, cl, cm)
    asm.ctx.force_path(0)
    _check_hetero_3d(ActiveSetDriver(setup, asm).run(n_steps=2))


def _check_threepoint(recs):
    """tests/threepoint_1.mpirun=2.output, first three time steps: line-0 residuals to the printed digits; energies and the
    Newton table of steps 0 and 1 (step 1 row by row: 317 119 76 47 40 40).  From step 2 on the active sets of the reference
    and of any other correct implementation part ways by round-off (one dof), and the energies follow at 1e-5 ... 1e-4."""
    g = cases.golden()["threepoint_1.mpirun=2"]["timesteps"]
    assert len(recs) == 3
    for rec, gg in zip(recs, g):
        assert rec.residual0 == pytest.approx(gg["residual0"], rel=2e-6)
        assert rec.newton[-1].residual < 1e-6
    for rec, gg in zip(recs[:2], g):
        assert rec.bulk_energy == pytest.approx(gg["bulk_energy"], rel=5e-6)
        assert rec.crack_energy == pytest.approx(gg["crack_energy"], rel=5e-6)
        assert rec.newton[-1].active_set == gg["newton"][-1]["active_set"]
    assert [r.active_set for r in recs[1].newton] == [x["active_set"] for x in g[1]["newton"]]
    assert recs[2].bulk_energy == pytest.approx(g[2]["bulk_energy"], rel=1e-4)
    assert recs[2].crack_energy == pytest.approx(g[2]["crack_energy"], rel=1e-3)


def test_threepoint_first_steps_with_oracle():
    setup = NC.threepoint_setup()
    _check_threepoint(ActiveSetDriver(setup, NC.OracleAssembler(setup.mesh, setup.layout)).run(n_steps=3))


@pytest.mark.gpu
def test_threepoint_first_steps_on_gpu():
    """General family on general quadrilaterals (MappingQ1 per q-point) with the stress split, inhomogeneous point load."""
    from cracks_amd.newton import GpuAssembler

    setup = NC.threepoint_setup()
    _check_threepoint(ActiveSetDriver(setup, GpuAssembler(setup.mesh, setup.layout)).run(n_steps=3))


def _check_miehe_tension(recs):
    """tests/miehe_tension_adaptive_1.output, the time steps before its first refinement: line-0 residual, energies, the
    number of Newton rows (4) and the empty converged active set of every step."""
    g = cases.golden()["miehe_tension_adaptive_1"]["timesteps"]
    assert len(recs) == 4
    for rec, gg in zip(recs, g):
        assert gg["cells"] == 256
        assert rec.residual0 == pytest.approx(gg["residual0"], rel=2e-6)
        assert rec.bulk_energy == pytest.approx(gg["bulk_energy"], rel=5e-6)
        assert rec.crack_energy == pytest.approx(gg["crack_energy"], rel=5e-6)
        assert len(rec.newton) == len(gg["newton"]) and rec.newton[-1].residual < 1e-6
        assert rec.newton[-1].active_set == gg["newton"][-1]["active_set"] == 0
        assert rec.newton[0].residual == pytest.approx(gg["newton"][0]["residual"], rel=2e-2)


def test_miehe_tension_first_steps_with_oracle():
    setup = NC.miehe_tension_setup()
    _check_miehe_tension(ActiveSetDriver(setup, NC.OracleAssembler(setup.mesh, setup.layout)).run(n_steps=4))


@pytest.mark.gpu
def test_miehe_tension_first_steps_on_gpu():
    from cracks_amd.newton import GpuAssembler

    setup = NC.miehe_tension_setup()
    _check_miehe_tension(ActiveSetDriver(setup, GpuAssembler(setup.mesh, setup.layout)).run(n_steps=4))
