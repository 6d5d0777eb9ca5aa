"""Problem set-ups shared by the oracle and GPU parity tests.

Each ``kat_*`` function rebuilds the state the reference is in when it prints line 0 of the
first Newton table (cracks.cc:2787-2799): interpolated initial condition, ``set_initial_bc``,
hanging nodes distributed, ``old = old_old = solution``, ``time = timestep``.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from cracks_amd import mesh as M
from oracle_api import PfmParams, lame_from_E_nu, make_params

HERE = os.path.dirname(os.path.abspath(__file__))


def golden():
    with open(os.path.join(HERE, "golden", "kat.json")) as f:
        return json.load(f)


@dataclass
class Case:
    name: str
    mesh: M.Mesh
    layout: M.DofLayout
    params: PfmParams
    sol: np.ndarray
    old: np.ndarray
    oldold: np.ndarray
    cu: M.ConstraintSet  # constraints_update
    ch: M.ConstraintSet  # constraints_hanging_nodes
    golden_residual0: Optional[float] = None
    cell_lambda: Optional[np.ndarray] = None
    cell_mu: Optional[np.ndarray] = None
    extra: dict = field(default_factory=dict)


def kat_sneddon_3d(n: int = 10) -> Case:
    """tests/sneddon_3d_1.prm -> tests/sneddon_3d_1.mpirun=4.output:29 (6.744161e+01)."""
    mesh = M.box_mesh(3, n)
    h = mesh.min_cell_diameter()
    lay = M.DofLayout(mesh.n_nodes, 3, blocked=True)  # Use Direct Inner Solver = false
    lam, mu = lame_from_E_nu(1.0, 0.2)
    prm = make_params(**{"lambda": lam}, mu=mu, G_c=1.0, alpha_eps=2.0 * h, constant_k=0.0,
                      pressure=1.0e-3, timestep=1.0, time=1.0, old_timestep=1.0,
                      old_old_timestep=1.0, timestep_number=0)
    phi = M.initial_values_sneddon(mesh, h)
    sol = lay.pack(np.zeros((mesh.n_nodes, 3)), phi)
    ch = M.hanging_constraints(mesh, lay)
    cu = M.update_constraints(mesh, lay, M.sneddon_dirichlet_dofs(mesh, lay))
    g = golden()["sneddon_3d_1.mpirun=4"]["timesteps"][0]["residual0"] if n == 10 else None
    return Case("sneddon_3d", mesh, lay, prm, sol, sol.copy(), sol.copy(), cu, ch, g)


def kat_sneddon_2d() -> Case:
    """tests/sneddon_2d_1.prm -> tests/sneddon_2d_1.output:32 (1.491639e+01); 12 hanging nodes."""
    mesh = M.sneddon_2d_prerefined_mesh()
    h = mesh.min_cell_diameter()
    lay = M.DofLayout(mesh.n_nodes, 2, blocked=True)
    lam, mu = lame_from_E_nu(1.0, 0.2)
    prm = make_params(**{"lambda": lam}, mu=mu, G_c=1.0, alpha_eps=2.0 * h, constant_k=1e-8 * h,
                      pressure=1.0e-3, timestep=1.0, time=1.0, old_timestep=1.0,
                      old_old_timestep=1.0, timestep_number=0)
    phi = M.initial_values_sneddon(mesh, h)
    ch = M.hanging_constraints(mesh, lay)
    cu = M.update_constraints(mesh, lay, M.sneddon_dirichlet_dofs(mesh, lay))
    sol = ch.distribute(lay.pack(np.zeros((mesh.n_nodes, 2)), phi))  # cracks.cc:2788
    g = golden()["sneddon_2d_1"]["timesteps"][0]["residual0"]
    return Case("sneddon_2d", mesh, lay, prm, sol, sol.copy(), sol.copy(), cu, ch, g)


def _miehe(name, key, dt, cycles, tension=False, blocked=False, k_factor=1.0e-10) -> Case:
    mesh = M.slit_mesh(3)
    # determine_mesh_dependent_parameters, cracks.cc:3839-3854: coarse diameter * 2^-(global+cycles+local)
    h = 0.5 * np.sqrt(2.0) * 2.0 ** (-(3 + cycles + 0))
    lay = M.DofLayout(mesh.n_nodes, 2, blocked=blocked)
    prm = make_params(**{"lambda": 121.15e3}, mu=80.77e3, G_c=2.7, alpha_eps=2.0 * h,
                      constant_k=k_factor * h, pressure=0.0, timestep=dt, time=dt, old_timestep=dt,
                      old_old_timestep=dt, timestep_number=0,
                      decompose_stress_rhs=0.0 if tension else 1.0,
                      decompose_stress_matrix=0.0 if tension else 1.0)
    u = np.zeros((mesh.n_nodes, 2))
    top = mesh.boundary_nodes[3]
    if tension:
        u[top, 1] = dt  # BoundaryTensionTest, cracks.cc:776-798
        dd = M.boundary_dofs(mesh, lay, [(2, [1]), (3, [0, 1])])  # cracks.cc:2584-2599
    else:
        u[top, 0] = -dt  # BoundaryShearTest, cracks.cc:838-858
        dd = M.miehe_shear_dirichlet_dofs(mesh, lay)
    sol = lay.pack(u, np.ones(mesh.n_nodes))
    ic = lay.pack(np.zeros_like(u), np.ones(mesh.n_nodes))  # old = old_old = interpolated IC
    ch = M.hanging_constraints(mesh, lay)
    cu = M.update_constraints(mesh, lay, dd)
    g = golden()[key]["timesteps"][0]["residual0"]
    return Case(name, mesh, lay, prm, sol, ic, ic.copy(), cu, ch, g)


def kat_miehe_shear_1() -> Case:
    """tests/miehe_shear_1.prm -> tests/miehe_shear_1.output:29 (3.179919e+02)."""
    return _miehe("miehe_shear_1", "miehe_shear_1", 1.0e-3, 1)


def kat_miehe_shear_2() -> Case:
    """tests/miehe_shear_2.prm -> tests/miehe_shear_2.output (1.589959e+02)."""
    return _miehe("miehe_shear_2", "miehe_shear_2", 5.0e-4, 0)


def kat_miehe_tension() -> Case:
    """tests/miehe_tension_adaptive_1.prm -> .output (2.790609e+02); iterative-solver layout."""
    return _miehe("miehe_tension", "miehe_tension_adaptive_1", 2.5e-4, 1, tension=True, blocked=True,
                  k_factor=0.0)


def kat_hetero_3d() -> Case:
    """tests/hetero_3d_1.prm -> tests/hetero_3d_1.mpirun-4.output:31 (2.772590e+01): heterogeneous material
    (cracks.cc:2207-2216, E modulus per cell from the reference's bitmap ``test.pgm`` through ``BitmapFunction``,
    cracks.cc:118-241), 3-D hanging nodes, pressure ``1e3 * time``.  The per-cell values of ``func_emodulus`` are the
    fixture ``golden/hetero_3d_emod.json`` (made by ``golden/make_hetero_emod.py`` from the bitmap)."""
    mesh = M.hetero_3d_prerefined_mesh()
    h = mesh.min_cell_diameter()
    lay = M.DofLayout(mesh.n_nodes, 3, blocked=True)  # Use Direct Inner Solver = false
    E, nu, dt = 1.0e4, 0.2, 0.01
    lam, mu = lame_from_E_nu(E, nu)
    prm = make_params(**{"lambda": lam}, mu=mu, G_c=1.0, alpha_eps=1.5, constant_k=0.0,
                      pressure=1.0e3 * dt, timestep=dt, time=dt, old_timestep=dt,
                      old_old_timestep=dt, timestep_number=0)
    phi = M.initial_values_multiple_het(mesh, h)
    ch = M.hanging_constraints(mesh, lay)
    cu = M.update_constraints(mesh, lay, M.sneddon_dirichlet_dofs(mesh, lay))  # u = 0 on all six faces, cracks.cc:2688-2696
    sol = ch.distribute(lay.pack(np.zeros((mesh.n_nodes, 3)), phi))
    with open(os.path.join(HERE, "golden", "hetero_3d_emod.json")) as f:
        fx = json.load(f)
    table = {tuple(np.round(np.asarray(c[:3]) * 64).astype(int)): c[3] for c in fx["cells"]}
    centres = mesh.coords[mesh.cells].mean(axis=1)
    emod = np.array([table[tuple(np.round(c * 64).astype(int))] for c in centres]) + 1.0  # cracks.cc:2209-2210
    cell_mu = emod / (2.0 * (1 + nu))
    cell_lambda = (2 * nu * cell_mu) / (1.0 - 2 * nu)
    g = golden()["hetero_3d_1.mpirun-4"]["timesteps"][0]["residual0"]
    return Case("hetero_3d", mesh, lay, prm, sol, sol.copy(), sol.copy(), cu, ch, g, cell_lambda, cell_mu)


def kat_threepoint() -> Case:
    """tests/threepoint_1.prm -> tests/threepoint_1.mpirun=2.output (1.210342e+02): the reference's unstructured gmsh
    mesh (``meshes/threepoint.msh``, fixture ``golden/threepoint_mesh.json``) -- general quadrilaterals, i.e. MappingQ1
    geometry per quadrature point; point constraints of cracks.cc:2626-2676."""
    with open(os.path.join(HERE, "golden", "threepoint_mesh.json")) as f:
        fx = json.load(f)
    mesh = M.Mesh(dim=2, coords=np.asarray(fx["coords"], float), cells=np.asarray(fx["cells"], np.int32))
    # determine_mesh_dependent_parameters, cracks.cc:3839-3854: largest coarse diameter * 2^-(global+cycles+local)
    h = mesh.cell_diameters().max() * 2.0 ** (-(0 + 1 + 0))
    lay = M.DofLayout(mesh.n_nodes, 2, blocked=True)  # Use Direct Inner Solver = false
    dt = 5.0e-3
    prm = make_params(**{"lambda": 12.0e3}, mu=8.0e3, G_c=0.25, alpha_eps=2.0 * h, constant_k=1.0e-10,
                      pressure=0.0, timestep=dt, time=dt, old_timestep=dt, old_old_timestep=dt,
                      timestep_number=0, decompose_stress_rhs=1.0, decompose_stress_matrix=1.0)
    x, y = mesh.coords[:, 0], mesh.coords[:, 1]
    u = np.zeros((mesh.n_nodes, 2))
    dd = []
    for n in np.nonzero((np.abs(y) < 1e-10) & ((np.abs(x + 4.0) < 1e-10) | (np.abs(x - 4.0) < 1e-10)))[0]:
        dd.append(lay.dof(n, 1))  # y displacement of both bottom corners
        if abs(x[n] + 4.0) < 1e-10:
            dd.append(lay.dof(n, 0))  # x displacement of the left one
        dd.append(lay.dof(n, 2))  # phase field (inhomogeneity 1.0 = the initial value)
    for n in np.nonzero((np.abs(x) < 1e-10) & (np.abs(y - 2.0) < 1e-10))[0]:
        dd.append(lay.dof(n, 1))
        u[n, 1] = -1.0 * dt  # constraints.set_inhomogeneity(idx, -1.0*time), cracks.cc:2668-2669
    dd = np.asarray(sorted(int(d) for d in dd), np.int64)
    sol = lay.pack(u, np.ones(mesh.n_nodes))  # InitialValuesNoCrack + set_initial_bc
    ic = lay.pack(np.zeros_like(u), np.ones(mesh.n_nodes))
    ch = M.hanging_constraints(mesh, lay)
    cu = M.update_constraints(mesh, lay, dd)
    g = golden()["threepoint_1.mpirun=2"]["timesteps"][0]["residual0"]
    return Case("threepoint", mesh, lay, prm, sol, ic, ic.copy(), cu, ch, g)


ALL_KATS = [kat_threepoint, kat_sneddon_3d, kat_sneddon_2d, kat_miehe_shear_1, kat_miehe_shear_2, kat_miehe_tension, kat_hetero_3d]


def perturbed(case: Case, seed: int = 1234, u_amp: float = 1e-3, phi_amp: float = 0.2) -> Case:
    """Seeded perturbation of a case so that no term of the assembly is identically zero
    (SURVEY.md §8(d) config 2/3 inputs): u ~ U(-u_amp,u_amp), phi clipped to [0,1],
    independent perturbations of old / old_old; constrained dofs re-imposed."""
    rng = np.random.default_rng(seed)
    lay, mesh = case.layout, case.mesh
    node, comp = lay.node_comp_of_dof()
    is_phi = comp == lay.dim

    def pert(v):
        w = v.copy()
        w[~is_phi] += rng.uniform(-u_amp, u_amp, (~is_phi).sum())
        w[is_phi] = np.clip(w[is_phi] + rng.uniform(-phi_amp, phi_amp, is_phi.sum()), 0.0, 1.0)
        return case.ch.distribute(w)

    sol = pert(case.sol)
    # keep the Dirichlet values of the unperturbed solution (set_initial_bc)
    dmask = case.cu.flag.astype(bool) & ~case.ch.flag.astype(bool)
    sol[dmask] = case.sol[dmask]
    return Case(case.name + "_pert", mesh, lay, case.params, sol, pert(case.old), pert(case.oldold),
                case.cu, case.ch, None, case.cell_lambda, case.cell_mu)
