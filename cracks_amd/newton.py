"""Host harness around the assembly ABI: the reference's callers, restated so that the
converged-state goldens (Newton tables, energies, loads in ``tests/*.output``) can be
reproduced end to end — SURVEY.md §8(f) row N1 (+ the two cheap functionals of N2).

This is NOT part of the accelerated path and not meant to be fast: it is the reference's
outer loop in numpy/scipy, calling an *assembler object* for the two hot-path entry points.

    newton_active_set()            cracks.cc:2781-2994
    solve() (direct branch)        cracks.cc:2744-2759   -> scipy SuperLU
    time loop, active-set branch   cracks.cc:4284-4566
    project_back_phase_field()     cracks.cc:3111-3137
    assemble_diag_mass_matrix()    cracks.cc:2514-2562
    compute_energy()               cracks.cc:3615-3701
    compute_load()                 cracks.cc:3726-3816 (2-D, boundary id 3)

Assembler protocol (duck typed): ``assemble(residual_only, sol, old, oldold, params, cu, ch)``
returns ``(A, res_pde, res_total)`` with ``A`` a scipy CSR over the global dofs (``None`` for
``residual_only``).  ``GpuAssembler`` below drives the HIP library; the tests have an
oracle-backed twin.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import numpy as np

from . import mesh as M


class NoConvergence(RuntimeError):
    """SolverControl::NoConvergence (cracks.cc:2983-2988)."""


@dataclass
class NewtonRow:
    it: int
    active_set: int
    cycling: int
    residual: float
    reduction: float
    line_search: int
    lin_its: int


@dataclass
class StepRecord:
    timestep: int
    time_before: float
    dt: float
    residual0: float = 0.0
    newton: List[NewtonRow] = field(default_factory=list)
    bulk_energy: float = 0.0
    crack_energy: float = 0.0
    load: Optional[float] = None
    tcv: Optional[float] = None  # total crack volume (cracks.cc:3553-3611), device sweep only


def lumped_phase_mass(mesh: M.Mesh, layout: M.DofLayout) -> np.ndarray:
    """``diag_mass`` (cracks.cc:2514-2562): QGaussLobatto(2) puts the quadrature points on the
    vertices, so the phase-field dof of vertex a receives JxW(vertex a) = det J(a) / 2^dim."""
    dim, nv = mesh.dim, mesh.nv
    x = mesh.coords[mesh.cells]  # [cells, nv, dim]
    diag = np.zeros(layout.n_dofs)
    for a in range(nv):
        J = np.zeros((mesh.n_cells, dim, dim))
        for v in range(nv):
            g = np.ones(dim)
            for e in range(dim):
                for d in range(dim):
                    xa = (a >> d) & 1
                    f = xa if ((v >> d) & 1) else 1 - xa
                    df = 1.0 if ((v >> d) & 1) else -1.0
                    g[e] *= df if d == e else f
            J += x[:, v, :, None] * g[None, None, :]
        det = np.linalg.det(J)
        np.add.at(diag, layout.dof(mesh.cells[:, a], dim), det * 0.5 ** dim)
    return diag


def _gauss3(dim):
    gx = np.array([0.5 - 0.5 * 0.7745966692414834, 0.5, 0.5 + 0.5 * 0.7745966692414834])
    gw = np.array([5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0])
    pts, wts = [], []
    for q in range(3 ** dim):
        qi = [q % 3, (q // 3) % 3, q // 9]
        pts.append([gx[qi[d]] for d in range(dim)])
        wts.append(np.prod([gw[qi[d]] for d in range(dim)]))
    return np.array(pts), np.array(wts)


def _shape(dim, xi):
    nv = 1 << dim
    N = np.ones(nv)
    dN = np.ones((nv, dim))
    for v in range(nv):
        for d in range(dim):
            f = xi[d] if ((v >> d) & 1) else 1.0 - xi[d]
            df = 1.0 if ((v >> d) & 1) else -1.0
            N[v] *= f
            for e in range(dim):
                dN[v, e] *= df if e == d else f
    return N, dN


def compute_energy(mesh, layout, sol, lam, mu, G_c, eps, kappa):
    """cracks.cc:3615-3701: bulk = int ((1+k) pf^2 + k) psi(e), crack = G_c/2 int ((pf-1)^2/eps + eps |grad pf|^2)."""
    dim, nv = mesh.dim, mesh.nv
    x = mesh.coords[mesh.cells]
    n = np.arange(mesh.n_nodes)
    U = np.stack([sol[layout.dof(n, c)] for c in range(dim)], axis=1)[mesh.cells]  # [cells, nv, dim]
    PH = sol[layout.dof(n, dim)][mesh.cells]
    pts, wts = _gauss3(dim)
    bulk = crack = 0.0
    for xi, w in zip(pts, wts):
        N, dN = _shape(dim, xi)
        J = np.einsum("cvi,vj->cij", x, dN)
        det = np.linalg.det(J)
        inv = np.linalg.inv(J)
        g = np.einsum("cej,ve->cvj", inv, dN)  # physical gradients [cells, nv, dim]
        gu = np.einsum("cvi,cvj->cij", U, g)
        gpf = np.einsum("cv,cvj->cj", PH, g)
        pf = PH @ N
        E = 0.5 * (gu + np.swapaxes(gu, 1, 2))
        trE = np.trace(E, axis1=1, axis2=2)
        tr_e_2 = np.einsum("cij,cji->c", E, E)
        psi = 0.5 * lam * trE * trE + mu * tr_e_2
        JxW = det * w
        bulk += np.sum(((1 + kappa) * pf * pf + kappa) * psi * JxW)
        crack += np.sum(G_c / 2.0 * ((pf - 1) ** 2 / eps + eps * np.einsum("cj,cj->c", gpf, gpf)) * JxW)
    return float(bulk), float(crack)


def compute_load_2d(mesh, layout, sol, lam, mu, boundary_id=3, component=0):
    """cracks.cc:3726-3816 for 2-D: -int_{id 3} (sigma n)_x (x component negated)."""
    assert mesh.dim == 2
    on_b = np.zeros(mesh.n_nodes, bool)
    on_b[mesh.boundary_nodes[boundary_id]] = True
    faces = {2: (0, 1), 3: (2, 3), 0: (0, 2), 1: (1, 3)}  # deal.II face -> vertices; 0/1: x lo/hi, 2/3: y lo/hi
    n = np.arange(mesh.n_nodes)
    Unod = np.stack([sol[layout.dof(n, c)] for c in range(2)], axis=1)
    gx = np.array([0.5 - 0.5 * 0.7745966692414834, 0.5, 0.5 + 0.5 * 0.7745966692414834])
    gw = np.array([5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0])
    load = np.zeros(2)
    for c in range(mesh.n_cells):
        cell = mesh.cells[c]
        for f, (va, vb) in faces.items():
            if not (on_b[cell[va]] and on_b[cell[vb]]):
                continue
            # the face must be a boundary face of this cell with that id: both vertices carry the id and the
            # face is on the domain boundary (true for the colorized / slit meshes used here)
            x = mesh.coords[cell]
            for q in range(3):
                xi = np.zeros(2)
                if f in (0, 1):
                    xi[0] = float(f == 1)
                    xi[1] = gx[q]
                else:
                    xi[1] = float(f == 3)
                    xi[0] = gx[q]
                N, dN = _shape(2, xi)
                J = x.T @ dN
                inv = np.linalg.inv(J)
                g = dN @ inv
                gu = Unod[cell].T @ g
                E = 0.5 * (gu + gu.T)
                sig = lam * np.trace(E) * np.eye(2) + 2 * mu * E
                tang = x[vb] - x[va]
                length = np.linalg.norm(tang)
                nrm = np.array([tang[1], -tang[0]]) / length
                # orient outward: away from the cell centre
                if np.dot(nrm, 0.5 * (x[va] + x[vb]) - x.mean(axis=0)) < 0:
                    nrm = -nrm
                load += sig @ nrm * (length * gw[q])
    load[0] *= -1.0
    return float(load[component])


@dataclass
class ProblemSetup:
    mesh: M.Mesh
    layout: M.DofLayout
    params: object  # PfmParams-like ctypes struct (mutated per step: time, timestep, ...)
    dirichlet_dofs: np.ndarray  # set_newton_bc lines (cracks.cc:2711-2714)
    initial_bc: Callable[[float], Dict[int, float]]  # set_initial_bc(time): dof -> value (cracks.cc:2699-2707)
    solution0: np.ndarray  # interpolated initial condition
    E_modulus: float  # only its role in the active-set constant c = 10 E (cracks.cc:2859)
    timestep: float
    max_no_timesteps: int
    newton_tol: float = 1e-6
    max_newton_steps: int = 100
    max_line_search: int = 10
    line_search_damping: float = 0.6
    compute_load: bool = False
    pressure_of_time: Optional[Callable[[float], float]] = None  # func_pressure(time), cracks.cc:2145 (constant when None)


class ActiveSetDriver:
    """Time loop + primal-dual active-set Newton of the reference around an assembler."""

    def __init__(self, setup: ProblemSetup, assembler, log: Optional[Callable[[str], None]] = None):
        self.s = setup
        self.asm = assembler
        self.log = log or (lambda msg: None)
        lay, mesh = setup.layout, setup.mesh
        self.ch = M.hanging_constraints(mesh, lay)
        self.diag_mass = lumped_phase_mass(mesh, lay)
        node, comp = lay.node_comp_of_dof()
        self.is_phi = comp == lay.dim
        self.solution = setup.solution0.copy()
        self.old_solution = self.solution.copy()
        self.old_old_solution = self.solution.copy()
        self.time = 0.0
        self.timestep = setup.timestep
        self.old_timestep = setup.timestep
        self.old_old_timestep = setup.timestep
        self.timestep_number = 0
        self.use_old_timestep_pf = False
        self.records: List[StepRecord] = []

    # ---- pieces -------------------------------------------------------------------------
    def _params(self):
        p = self.s.params
        p.time, p.timestep = self.time, self.timestep
        p.old_timestep, p.old_old_timestep = self.old_timestep, self.old_old_timestep
        p.timestep_number = self.timestep_number
        p.use_old_timestep_pf = 1 if self.use_old_timestep_pf else 0
        if self.s.pressure_of_time is not None:
            p.pressure = self.s.pressure_of_time(self.time)
        return p

    def _assemble(self, residual_only, cu):
        return self.asm.assemble(residual_only, self.solution, self.old_solution, self.old_old_solution,
                                 self._params(), cu, self.ch)

    def project_back_phase_field(self):
        self.solution[self.is_phi] = np.clip(self.solution[self.is_phi], 0.0, 1.0)

    def newton_active_set(self, rec: StepRecord) -> float:
        import scipy.sparse.linalg as spla

        s, lay, mesh = self.s, self.s.layout, self.s.mesh
        for d, val in s.initial_bc(self.time).items():  # set_initial_bc
            self.solution[d] = val
        self.solution = self.ch.distribute(self.solution)
        cu = M.update_constraints(mesh, lay, s.dirichlet_dofs)  # state at entry (setup_system / previous call)
        if hasattr(self, "_cu_last"):
            cu = self._cu_last
        _, res_pde, res_tot = self._assemble(True, cu)
        residual_relevant = res_tot.copy()
        newton_residual = float(np.linalg.norm(cu.set_zero(res_pde)))
        old_newton_residual = newton_residual
        rec.residual0 = newton_residual
        self.log(f"0\t\t\t{newton_residual:.6e}")
        active = np.zeros(lay.n_dofs, bool)
        cycle_counter = np.zeros(lay.n_dofs, np.int64)
        old_solution_relevant = self.old_solution.copy()
        hanging = self.ch.flag.astype(bool)
        newton_step = 0
        new_newton_residual = 0.0
        cconst = 1e1 * s.E_modulus
        while True:
            active_old = active.copy()
            cand = self.is_phi & ~hanging
            gap = self.solution - old_solution_relevant
            with np.errstate(divide="ignore", invalid="ignore"):
                crit = residual_relevant / self.diag_mass + cconst * gap
            inactive = (crit <= 0.0) & (cycle_counter < 5)
            active = cand & ~inactive
            n_cycling = int(np.sum(active & (cycle_counter >= 5)))
            self.solution[active] = old_solution_relevant[active]
            self.solution = self.ch.distribute(self.solution)
            cycle_counter[active_old & ~active] += 1
            cu = M.update_constraints(mesh, lay, s.dirichlet_dofs, np.nonzero(active)[0])
            self._cu_last = cu
            num_changed = int(not np.array_equal(active, active_old))
            A, res_pde, _ = self._assemble(False, cu)
            rhs = cu.set_zero(res_pde)
            update = spla.spsolve(A.tocsc(), rhs)
            update = cu.distribute(update)
            saved = self.solution.copy()
            ls = 0
            while ls < s.max_line_search:
                self.solution = self.solution + update
                _, res_pde, res_tot = self._assemble(True, cu)
                residual_relevant = res_tot.copy()
                new_newton_residual = float(np.linalg.norm(cu.set_zero(res_pde)))
                if new_newton_residual < newton_residual:
                    break
                self.solution = saved.copy()
                update = update * s.line_search_damping
                ls += 1
            row = NewtonRow(newton_step + 1, int(active.sum()), n_cycling, new_newton_residual,
                            new_newton_residual / newton_residual if newton_residual else 0.0, ls, 1)
            rec.newton.append(row)
            self.log(f"{row.it}\t{row.active_set}\t{row.cycling}\t{row.residual:.6e}\t{row.reduction:.6e}\t{ls}\t1")
            old_newton_residual = newton_residual
            newton_residual = new_newton_residual
            newton_step += 1
            if newton_residual < s.newton_tol and num_changed == 0:
                break
            if newton_step >= s.max_newton_steps:
                raise NoConvergence(f"Newton iteration did not converge in {newton_step} steps")
        return new_newton_residual / old_newton_residual if old_newton_residual else 0.0

    # ---- time loop (active-set branch) ---------------------------------------------------
    def run(self, n_steps: Optional[int] = None) -> List[StepRecord]:
        s = self.s
        limit = s.max_no_timesteps if n_steps is None else n_steps - 1
        self.project_back_phase_field()  # cracks.cc:4267
        self.old_old_solution = self.solution.copy()
        self.old_solution = self.solution.copy()
        while self.timestep_number <= limit:
            tmp_timestep = self.timestep
            self.old_old_timestep = self.old_timestep
            self.old_timestep = self.timestep
            self.old_old_solution = self.old_solution.copy()
            self.old_solution = self.solution.copy()
            rec = StepRecord(self.timestep_number, self.time, self.timestep)
            self.log(f"Timestep {self.timestep_number}: {self.time:g} ({self.timestep:g})")
            self.time += self.timestep
            while True:
                self.use_old_timestep_pf = False
                try:
                    self.newton_active_set(rec)
                    break
                except NoConvergence:
                    self.log(f"Solver did not converge! Adjusting time step to {self.timestep / 10:g}")
                self.use_old_timestep_pf = True
                self.solution = self.old_solution.copy()
                self.time -= self.timestep
                self.timestep = self.timestep / 10.0
                self.time += self.timestep
                rec = StepRecord(self.timestep_number, self.time - self.timestep, self.timestep)
            self.project_back_phase_field()
            self.solution = self.ch.distribute(self.solution)
            self.timestep = tmp_timestep
            p = s.params
            if hasattr(self.asm, "functionals"):  # device sweep (pfm_functionals), cracks.cc:3553-3701
                rec.bulk_energy, rec.crack_energy, rec.tcv = self.asm.functionals(self.solution, self.old_solution,
                                                                                  self.old_old_solution, self._params())
            else:
                rec.bulk_energy, rec.crack_energy = compute_energy(s.mesh, s.layout, self.solution, p.lambda_, p.mu,
                                                                   p.G_c, p.alpha_eps, p.constant_k)
            if s.compute_load:
                rec.load = compute_load_2d(s.mesh, s.layout, self.solution, p.lambda_, p.mu)
            self.log(f"No {self.timestep_number} time {self.time:g} bulk energy: {rec.bulk_energy:g} "
                     f"crack energy: {rec.crack_energy:g}" + (f"  Load x: {rec.load:g}" if rec.load is not None else ""))
            self.records.append(rec)
            self.timestep_number += 1
        return self.records


class GpuAssembler:
    """Assembler protocol on top of the HIP library (host-pointer entry point ``pfm_assemble``)."""

    def __init__(self, mesh, layout, cell_lambda=None, cell_mu=None):
        from .assembler import Context

        self.mesh, self.layout = mesh, layout
        self.ctx = Context(mesh, layout.blocked, cell_lambda=cell_lambda, cell_mu=cell_mu)
        self._pat = None

    def _global_matrix(self, values):
        import scipy.sparse as sp

        lay = self.layout
        n, dim, N = lay.n_dofs, lay.dim, lay.n_nodes
        if self._pat is None:
            self._pat = [self.ctx.pattern(b) for b in range(self.ctx.n_blocks)]
        if not lay.blocked:
            rp, ci = self._pat[0]
            return sp.csr_matrix((values[0], ci, rp), shape=(n, n))
        mats = []
        for b in range(4):
            rp, ci = self._pat[b]
            rows = (N * dim) if b in (0, 1) else N
            cols = (N * dim) if b in (0, 2) else N
            mats.append(sp.csr_matrix((values[b], ci, rp), shape=(rows, cols)))
        return sp.bmat([[mats[0], mats[1]], [mats[2], mats[3]]], format="csr")

    def assemble(self, residual_only, sol, old, oldold, params, cu, ch):
        from .assembler import node_flags_from_dof_flags

        self.ctx.set_params(params)
        self.ctx.set_constraints(node_flags_from_dof_flags(self.layout, cu.flag, ch.flag))
        values, res_pde, res_tot = self.ctx.assemble_host(sol, old, oldold, residual_only)
        return (None if residual_only else self._global_matrix(values)), res_pde, res_tot

    def functionals(self, sol, old, oldold, params):
        """(bulk energy, crack energy, TCV) on the device (include/pfm_newton.h)."""
        self.ctx.set_params(params)
        self.ctx.state_set_host(sol, old, oldold)
        return self.ctx.functionals()
