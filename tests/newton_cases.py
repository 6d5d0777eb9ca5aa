"""End-to-end set-ups of reference regression tests for the Newton harness (cracks_amd/newton.py)."""
import numpy as np

import cases
import oracle_api as O
from cracks_amd import mesh as M
from cracks_amd.newton import ProblemSetup


class OracleAssembler:
    """Assembler protocol backed by the CPU oracle (tests only)."""

    def __init__(self, mesh, layout, cell_lambda=None, cell_mu=None):
        self.mesh, self.layout = mesh, layout
        self.cell_lambda, self.cell_mu = cell_lambda, cell_mu
        self.rowptr, self.colind = M.dof_sparsity(mesh, layout)
        if cell_lambda is not None:  # per-cell material: the energies need it too (cracks.cc:3615-3701 with 2207-2216)
            self.functionals = self._functionals

    def _functionals(self, sol, old, oldold, params):
        p = O.PfmParams.from_buffer_copy(bytes(params))
        return O.functionals(self.mesh, self.layout, p, sol, self.cell_lambda, self.cell_mu)

    def assemble(self, residual_only, sol, old, oldold, params, cu, ch):
        import scipy.sparse as sp

        p = O.PfmParams.from_buffer_copy(bytes(params))
        r = O.assemble(self.mesh, self.layout, p, sol, old, oldold, cu, ch, residual_only,
                       None if residual_only else self.rowptr, None if residual_only else self.colind,
                       cell_lambda=self.cell_lambda, cell_mu=self.cell_mu)
        if r.err:
            raise RuntimeError(f"oracle error {r.err}")
        A = None
        if not residual_only:
            A = sp.csr_matrix((r.values, self.colind, self.rowptr), shape=(self.layout.n_dofs,) * 2)
        return A, r.residual_pde, r.residual_total


def miehe_shear_2_setup() -> ProblemSetup:
    """tests/miehe_shear_2.prm (direct solver, stress split, 256 cells, no refinement)."""
    c = cases.kat_miehe_shear_2()
    mesh, lay = c.mesh, c.layout
    top = mesh.boundary_nodes[3]
    dd = M.miehe_shear_dirichlet_dofs(mesh, lay)

    def initial_bc(time):
        vals = {int(d): 0.0 for d in dd}
        for n in top:  # BoundaryShearTest, cracks.cc:838-858
            vals[int(lay.dof(n, 0))] = -1.0 * time
        return vals

    sol0 = lay.pack(np.zeros((mesh.n_nodes, 2)), np.ones(mesh.n_nodes))
    return ProblemSetup(mesh=mesh, layout=lay, params=c.params, dirichlet_dofs=dd, initial_bc=initial_bc,
                        solution0=sol0, E_modulus=1.0e3, timestep=5.0e-4, max_no_timesteps=24,
                        newton_tol=1.0e-6, max_newton_steps=100, max_line_search=10, line_search_damping=0.6,
                        compute_load=True)


def sneddon_3d_setup() -> ProblemSetup:
    """tests/sneddon_3d_1.prm (10^3 cells, iterative solver in the reference; the harness solves exactly)."""
    c = cases.kat_sneddon_3d()
    mesh, lay = c.mesh, c.layout
    dd = M.sneddon_dirichlet_dofs(mesh, lay)
    return ProblemSetup(mesh=mesh, layout=lay, params=c.params, dirichlet_dofs=dd,
                        initial_bc=lambda time: {int(d): 0.0 for d in dd}, solution0=c.sol.copy(), E_modulus=1.0,
                        timestep=1.0, max_no_timesteps=5, newton_tol=1.0e-7, max_newton_steps=60,
                        max_line_search=50, line_search_damping=0.6)


def sneddon_2d_setup() -> ProblemSetup:
    """tests/sneddon_2d_1.prm (BASELINE config 0: 100 cells + one local pre-refinement = 124 cells with hanging nodes,
    iterative solver layout, active set, no stress split)."""
    c = cases.kat_sneddon_2d()
    mesh, lay = c.mesh, c.layout
    dd = M.sneddon_dirichlet_dofs(mesh, lay)
    return ProblemSetup(mesh=mesh, layout=lay, params=c.params, dirichlet_dofs=dd,
                        initial_bc=lambda time: {int(d): 0.0 for d in dd}, solution0=c.sol.copy(), E_modulus=1.0,
                        timestep=1.0, max_no_timesteps=3, newton_tol=1.0e-7, max_newton_steps=50,
                        max_line_search=10, line_search_damping=0.6)


def miehe_shear_1_setup() -> ProblemSetup:
    """tests/miehe_shear_1.prm (direct solver, stress split, dt = 1e-3; its predictor-corrector refinement only starts when the
    phase field drops: the first time steps run on the 256-cell mesh)."""
    c = cases.kat_miehe_shear_1()
    mesh, lay = c.mesh, c.layout
    top = mesh.boundary_nodes[3]
    dd = M.miehe_shear_dirichlet_dofs(mesh, lay)

    def initial_bc(time):
        vals = {int(d): 0.0 for d in dd}
        for n in top:  # BoundaryShearTest, cracks.cc:838-858
            vals[int(lay.dof(n, 0))] = -1.0 * time
        return vals

    sol0 = lay.pack(np.zeros((mesh.n_nodes, 2)), np.ones(mesh.n_nodes))
    return ProblemSetup(mesh=mesh, layout=lay, params=c.params, dirichlet_dofs=dd, initial_bc=initial_bc,
                        solution0=sol0, E_modulus=1.0e3, timestep=1.0e-3, max_no_timesteps=10,
                        newton_tol=1.0e-6, max_newton_steps=100, max_line_search=10, line_search_damping=0.6,
                        compute_load=True)


def hetero_3d_setup():
    """tests/hetero_3d_1.prm (multiple_het: per-cell E modulus, 3-D hanging nodes, pressure 1e3 * time, iterative-solver
    layout).  Returns (setup, cell_lambda, cell_mu)."""
    c = cases.kat_hetero_3d()
    mesh, lay = c.mesh, c.layout
    dd = M.sneddon_dirichlet_dofs(mesh, lay)
    setup = ProblemSetup(mesh=mesh, layout=lay, params=c.params, dirichlet_dofs=dd,
                         initial_bc=lambda time: {int(d): 0.0 for d in dd}, solution0=c.sol.copy(), E_modulus=1.0e4,
                         timestep=0.01, max_no_timesteps=2, newton_tol=1.0e-6, max_newton_steps=20,
                         max_line_search=8, line_search_damping=0.5, pressure_of_time=lambda t: 1.0e3 * t)
    return setup, c.cell_lambda, c.cell_mu


def threepoint_setup() -> ProblemSetup:
    """tests/threepoint_1.prm (three point bending on the reference's unstructured gmsh mesh: MappingQ1 on general
    quadrilaterals, stress split, point constraints of cracks.cc:2626-2676 with the inhomogeneous load u_y = -time at the top
    centre; iterative-solver layout).  Its phase-field refinement starts when phi < 0.5 somewhere: the first steps run on the
    280-cell mesh."""
    c = cases.kat_threepoint()
    mesh, lay = c.mesh, c.layout
    x, y = mesh.coords[:, 0], mesh.coords[:, 1]
    corners = np.nonzero((np.abs(y) < 1e-10) & ((np.abs(x + 4.0) < 1e-10) | (np.abs(x - 4.0) < 1e-10)))[0]
    top = np.nonzero((np.abs(x) < 1e-10) & (np.abs(y - 2.0) < 1e-10))[0]
    dd = np.nonzero(c.cu.flag.astype(bool) & ~c.ch.flag.astype(bool))[0].astype(np.int64)

    def initial_bc(time):
        vals = {int(d): 0.0 for d in dd}
        for n in corners:
            vals[int(lay.dof(n, 2))] = 1.0  # phase field at the supports: inhomogeneity 1.0
        for n in top:
            vals[int(lay.dof(n, 1))] = -1.0 * time  # cracks.cc:2668-2669
        return vals

    sol0 = lay.pack(np.zeros((mesh.n_nodes, 2)), np.ones(mesh.n_nodes))
    return ProblemSetup(mesh=mesh, layout=lay, params=c.params, dirichlet_dofs=dd, initial_bc=initial_bc, solution0=sol0,
                        E_modulus=1.0e3, timestep=5.0e-3, max_no_timesteps=8, newton_tol=1.0e-6, max_newton_steps=30,
                        max_line_search=10, line_search_damping=0.6)


def miehe_tension_setup() -> ProblemSetup:
    """tests/miehe_tension_adaptive_1.prm (tension test, iterative-solver layout, no stress split; its phase-field refinement
    starts when phi < 0.5 somewhere: the first time steps run on the 256-cell slit mesh)."""
    c = cases.kat_miehe_tension()
    mesh, lay = c.mesh, c.layout
    top = mesh.boundary_nodes[3]
    dd = M.boundary_dofs(mesh, lay, [(2, [1]), (3, [0, 1])])  # cracks.cc:2584-2599

    def initial_bc(time):
        vals = {int(d): 0.0 for d in dd}
        for n in top:  # BoundaryTensionTest, cracks.cc:776-798
            vals[int(lay.dof(n, 1))] = 1.0 * time
        return vals

    sol0 = lay.pack(np.zeros((mesh.n_nodes, 2)), np.ones(mesh.n_nodes))
    return ProblemSetup(mesh=mesh, layout=lay, params=c.params, dirichlet_dofs=dd, initial_bc=initial_bc, solution0=sol0,
                        E_modulus=1.0, timestep=2.5e-4, max_no_timesteps=32, newton_tol=1.0e-6, max_newton_steps=50,
                        max_line_search=10, line_search_damping=0.6)
