"""ctypes driver for the CPU oracle (oracle/liboracle.so).  Test infrastructure only —
imported by tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg, never
by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None


class PfmParams(C.Structure):
    """Mirror of include/pfm_params.h."""
    _fields_ = [
        ("lambda_", C.c_double), ("mu", C.c_double), ("G_c", C.c_double),
        ("alpha_eps", C.c_double), ("constant_k", C.c_double), ("pressure", C.c_double),
        ("alpha_biot", C.c_double), ("gamma_penal", C.c_double), ("timestep", C.c_double),
        ("time", C.c_double), ("old_timestep", C.c_double), ("old_old_timestep", C.c_double),
        ("decompose_stress_rhs", C.c_double), ("decompose_stress_matrix", C.c_double),
        ("timestep_number", C.c_int), ("outer_solver", C.c_int),
        ("use_old_timestep_pf", C.c_int), ("reserved", C.c_int),
    ]


def make_params(**kw) -> PfmParams:
    d = dict(lambda_=0.0, mu=0.0, G_c=1.0, alpha_eps=1.0, constant_k=0.0, pressure=0.0,
             alpha_biot=0.0, gamma_penal=0.0, timestep=1.0, time=1.0, old_timestep=1.0,
             old_old_timestep=1.0, decompose_stress_rhs=0.0, decompose_stress_matrix=0.0,
             timestep_number=0, outer_solver=0, use_old_timestep_pf=0, reserved=0)
    if "lambda" in kw:
        kw["lambda_"] = kw.pop("lambda")
    d.update(kw)
    return PfmParams(**d)


def lame_from_E_nu(E: float, nu: float):
    """cracks.cc:1507-1510."""
    mu = E / (2.0 * (1 + nu))
    lam = (2 * nu * mu) / (1.0 - 2 * nu)
    return lam, mu


def build_oracle(force: bool = False) -> str:
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build_oracle())
        _LIB.oracle_assemble.restype = C.c_int
        _LIB.oracle_assemble_range.restype = C.c_int
        _LIB.oracle_cell_local.restype = C.c_int
        _LIB.oracle_eigen_2x2.restype = C.c_int
        _LIB.oracle_decompose_stress_2d.restype = C.c_int
        for name in ("oracle_diag_mass", "oracle_functionals", "oracle_active_set"):
            getattr(_LIB, name).restype = C.c_int
    return _LIB


def _p(a: Optional[np.ndarray], dtype):
    if a is None:
        return None
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, dtype)
    return a.ctypes.data_as(C.c_void_p)


@dataclass
class OracleResult:
    err: int
    values: Optional[np.ndarray]
    residual_pde: np.ndarray
    residual_total: Optional[np.ndarray]


def assemble(mesh, layout, params: PfmParams, sol, old, oldold, cu, ch, residual_only: bool,
             rowptr=None, colind=None, cell_lambda=None, cell_mu=None,
             cell_range=None, out: Optional[OracleResult] = None) -> OracleResult:
    """One call of the reference's assemble_system(residual_only) on a single rank."""
    L = lib()
    cells = np.ascontiguousarray(mesh.cells, np.int32)
    coords = np.ascontiguousarray(mesh.coords, np.float64)
    cell_dofs = layout.cell_dofs(cells)
    n_dofs = layout.n_dofs
    sol = np.ascontiguousarray(sol, np.float64)
    old = np.ascontiguousarray(old, np.float64)
    oldold = np.ascontiguousarray(oldold, np.float64)
    if out is None:
        res_pde = np.zeros(n_dofs)
        res_tot = np.zeros(n_dofs) if residual_only else None
        values = None
        if not residual_only:
            values = np.zeros(int(rowptr[-1]))
    else:
        res_pde, res_tot, values = out.residual_pde, out.residual_total, out.values
    if rowptr is None:
        rowptr = np.zeros(n_dofs + 1, np.int64)
        colind = np.zeros(0, np.int32)
    args = [C.c_int(mesh.dim), C.c_int64(mesh.n_cells), C.c_int32(n_dofs),
            _p(cells, np.int32), _p(coords, np.float64), _p(cell_dofs, np.int32),
            _p(cell_lambda, np.float64), _p(cell_mu, np.float64), C.byref(params),
            _p(sol, np.float64), _p(old, np.float64), _p(oldold, np.float64),
            _p(cu.flag, np.uint8), _p(cu.ptr, np.int64), _p(cu.col, np.int32), _p(cu.w, np.float64),
            _p(ch.flag, np.uint8), _p(ch.ptr, np.int64), _p(ch.col, np.int32), _p(ch.w, np.float64),
            C.c_int(1 if residual_only else 0),
            _p(np.ascontiguousarray(rowptr, np.int64), np.int64),
            _p(np.ascontiguousarray(colind, np.int32), np.int32),
            _p(values, np.float64), _p(res_pde, np.float64), _p(res_tot, np.float64)]
    if cell_range is None:
        err = L.oracle_assemble(*args)
    else:
        err = L.oracle_assemble_range(*args, C.c_int64(cell_range[0]), C.c_int64(cell_range[1]))
    return OracleResult(err, values, res_pde, res_tot)


def cell_local(dim, vertex_coords, params, lam, mu, U, Uold_pf, Uoldold_pf, residual_only=False):
    nv = 1 << dim
    dpc = nv * (dim + 1)
    lm = np.zeros((dpc, dpc))
    lr = np.zeros(dpc)
    err = lib().oracle_cell_local(
        C.c_int(dim), _p(np.ascontiguousarray(vertex_coords, np.float64), np.float64),
        C.byref(params), C.c_double(lam), C.c_double(mu),
        _p(np.ascontiguousarray(U, np.float64), np.float64),
        _p(np.ascontiguousarray(Uold_pf, np.float64), np.float64),
        _p(np.ascontiguousarray(Uoldold_pf, np.float64), np.float64),
        C.c_int(1 if residual_only else 0), _p(lm, np.float64), _p(lr, np.float64))
    return err, lm, lr


def eigen_2x2(m):
    m = np.ascontiguousarray(m, np.float64).reshape(4)
    e1, e2 = C.c_double(), C.c_double()
    ev = np.zeros(4)
    err = lib().oracle_eigen_2x2(_p(m, np.float64), C.byref(e1), C.byref(e2), _p(ev, np.float64))
    return err, e1.value, e2.value, ev.reshape(2, 2)


def decompose_stress_2d(E, E_LinU, lam, mu, derivative: bool):
    E = np.ascontiguousarray(E, np.float64).reshape(4)
    EL = np.ascontiguousarray(E_LinU, np.float64).reshape(4)
    sp, sm = np.zeros(4), np.zeros(4)
    err = lib().oracle_decompose_stress_2d(_p(E, np.float64), _p(EL, np.float64), C.c_double(lam),
                                           C.c_double(mu), C.c_int(1 if derivative else 0),
                                           _p(sp, np.float64), _p(sm, np.float64))
    return err, sp.reshape(2, 2), sm.reshape(2, 2)


# ---- Newton-side sweeps (oracle.cpp: diag_mass, functionals, oracle_active_set) -----------------------------
def diag_mass(mesh, layout) -> np.ndarray:
    """assemble_diag_mass_matrix, cracks.cc:2514-2562; one entry per dof."""
    cells = np.ascontiguousarray(mesh.cells, np.int32)
    coords = np.ascontiguousarray(mesh.coords, np.float64)
    cell_dofs = layout.cell_dofs(cells)
    diag = np.zeros(layout.n_dofs)
    err = lib().oracle_diag_mass(C.c_int(mesh.dim), C.c_int64(mesh.n_cells), C.c_int32(layout.n_dofs),
                                 _p(cells, np.int32), _p(coords, np.float64), _p(cell_dofs, np.int32), _p(diag, np.float64))
    assert err == 0
    return diag


def functionals(mesh, layout, params: PfmParams, sol, cell_lambda=None, cell_mu=None, cell_owned=None):
    """(bulk energy, crack energy, TCV): compute_energy cracks.cc:3615-3701, compute_tcv cracks.cc:3553-3611."""
    cells = np.ascontiguousarray(mesh.cells, np.int32)
    coords = np.ascontiguousarray(mesh.coords, np.float64)
    cell_dofs = layout.cell_dofs(cells)
    out = np.zeros(3)
    own = None if cell_owned is None else np.ascontiguousarray(cell_owned, np.uint8)
    err = lib().oracle_functionals(C.c_int(mesh.dim), C.c_int64(mesh.n_cells), _p(cells, np.int32), _p(coords, np.float64),
                                   _p(cell_dofs, np.int32), _p(cell_lambda, np.float64), _p(cell_mu, np.float64),
                                   C.byref(params), _p(np.ascontiguousarray(sol, np.float64), np.float64),
                                   _p(own, np.uint8), _p(out, np.float64))
    assert err == 0
    return float(out[0]), float(out[1]), float(out[2])


def active_set(is_phi, hanging, residual_relevant, diag_mass_relevant, c, solution, old_solution, cycle_counter, active):
    """cracks.cc:2837-2886, 2903-2909 over dofs; solution, cycle_counter, active are updated in place.
    Returns (n_active, n_cycling, changed)."""
    n = solution.size
    counts = np.zeros(3, np.int64)
    err = lib().oracle_active_set(C.c_int32(n), _p(np.ascontiguousarray(is_phi, np.uint8), np.uint8),
                                  _p(np.ascontiguousarray(hanging, np.uint8), np.uint8),
                                  _p(np.ascontiguousarray(residual_relevant, np.float64), np.float64),
                                  _p(np.ascontiguousarray(diag_mass_relevant, np.float64), np.float64), C.c_double(c),
                                  _p(solution, np.float64), _p(np.ascontiguousarray(old_solution, np.float64), np.float64),
                                  _p(cycle_counter, np.int32), _p(active, np.uint8), _p(counts, np.int64))
    assert err == 0
    return int(counts[0]), int(counts[1]), int(counts[2])
