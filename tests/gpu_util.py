"""Helpers for the GPU parity tests: run a Case through the C ABI and put the result in
the oracle's index space (one global CSR over all dofs)."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from cracks_amd.assembler import Context, node_flags_from_dof_flags


def make_context(case, **kw) -> Context:
    ctx = Context(case.mesh, case.layout.blocked, cell_lambda=case.cell_lambda, cell_mu=case.cell_mu, **kw)
    ctx.set_params(case.params)
    ctx.set_constraints(node_flags_from_dof_flags(case.layout, case.cu.flag, case.ch.flag))
    return ctx


def blocks_to_global(ctx: Context, layout, values) -> sp.csr_matrix:
    """Assemble the per-block CSR value arrays into one matrix over the global dofs."""
    n, dim, N = layout.n_dofs, layout.dim, layout.n_nodes
    if not layout.blocked:
        rp, ci = ctx.pattern(0)
        return sp.csr_matrix((values[0], ci, rp), shape=(n, n))
    mats = []
    for b in range(4):
        rp, ci = ctx.pattern(b)
        rows = (N * dim) if b in (0, 1) else N
        cols = (N * dim) if b in (0, 2) else N
        mats.append(sp.csr_matrix((values[b], ci, rp), shape=(rows, cols)))
    return sp.bmat([[mats[0], mats[1]], [mats[2], mats[3]]], format="csr")


def linf_scaled(a, b) -> float:
    """l_inf error scaled by max(1, |reference|_inf)."""
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
