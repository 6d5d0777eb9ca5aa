"""Host-side mirror of the reference's assembler interface for the hot path.

``Assembler`` plays the role of the slice of ``FracturePhaseFieldProblem<dim>`` that
owns ``assemble_system(bool residual_only)`` / ``assemble_nl_residual()``
(cracks.cc:2129-2512): same member names (``solution``, ``old_solution``,
``old_old_solution``, ``system_pde_matrix``, ``system_pde_residual``,
``system_total_residual``), same call shapes, same error behaviour (exceptions where the
reference throws / aborts).  All arithmetic happens in the HIP library behind the C ABI
(``include/pfm_assemble.h``); PyTorch only provides device memory, streams and — in
``cracks_amd.halo`` — ``torch.distributed`` (RCCL).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from . import capi
from .capi import PfmError, PfmParams


def node_flags_from_dof_flags(layout, update_flag: np.ndarray, hanging_flag: Optional[np.ndarray] = None) -> np.ndarray:
    """One byte per node, bit c = dof (node, c) carries a homogeneous line of
    ``constraints_update`` that is not a hanging-node line (those travel in the mesh
    description)."""
    n = np.arange(layout.n_nodes)
    flags = np.zeros(layout.n_nodes, np.uint8)
    for c in range(layout.nc):
        d = layout.dof(n, c)
        f = update_flag[d].astype(bool)
        if hanging_flag is not None:
            f &= ~hanging_flag[d].astype(bool)
        flags |= (f.astype(np.uint8) << c).astype(np.uint8)
    return flags


class Context:
    """RAII wrapper of ``pfm_ctx``."""

    def __init__(self, mesh, blocked: bool, device: int = 0, n_owned_nodes: Optional[int] = None,
                 cell_lambda: Optional[np.ndarray] = None, cell_mu: Optional[np.ndarray] = None):
        self.lib = capi.load()
        self.dim = mesh.dim
        self.blocked = bool(blocked)
        self.n_nodes = mesh.n_nodes
        self.n_owned = mesh.n_nodes if n_owned_nodes is None else int(n_owned_nodes)
        self.n_blocks = 4 if blocked else 1
        d = capi.PfmMeshDesc()
        d.dim = mesh.dim
        d.layout = capi.LAYOUT_BLOCKED if blocked else capi.LAYOUT_INTERLEAVED
        d.n_nodes = mesh.n_nodes
        d.n_owned_nodes = self.n_owned
        d.n_cells = mesh.n_cells
        keep = []

        def arr(a, dt):
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return a.ctypes.data

        d.cell_nodes = arr(mesh.cells, np.int32)
        d.coords = arr(mesh.coords, np.float64)
        if cell_lambda is not None:
            d.cell_lambda = arr(cell_lambda, np.float64)
            d.cell_mu = arr(cell_mu, np.float64)
        d.n_hanging = int(mesh.hn_nodes.size)
        if d.n_hanging:
            d.hn_nodes = arr(mesh.hn_nodes, np.int32)
            d.hn_ptr = arr(mesh.hn_ptr, np.int64)
            d.hn_parents = arr(mesh.hn_parents, np.int32)
            d.hn_weights = arr(mesh.hn_weights, np.float64)
        if getattr(mesh, "box_shape", None):
            for k, v in enumerate(mesh.box_shape):
                d.box_cells[k] = int(v)
        self._h = C.c_void_p()
        import time as _time
        _t0 = _time.perf_counter()
        rc = self.lib.pfm_ctx_create(C.byref(self._h), C.byref(d), int(device))
        self.create_seconds = _time.perf_counter() - _t0  # pfm_ctx_create alone (synchronous): the cost after every refine_mesh
        if rc != capi.PFM_OK:
            msg = self.lib.pfm_last_error(self._h).decode() if self._h else ""
            if self._h:
                self.lib.pfm_ctx_destroy(self._h)
                self._h = C.c_void_p()
            raise PfmError(rc, "pfm_ctx_create", msg)
        self.n_owned_dofs = self.n_owned * (self.dim + 1)

    # -- helpers ---------------------------------------------------------------------
    def _check(self, rc: int, where: str):
        if rc != capi.PFM_OK:
            raise PfmError(rc, where, self.lib.pfm_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self.lib.pfm_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- C ABI -----------------------------------------------------------------------
    def set_stream(self, stream_handle: int):
        self._check(self.lib.pfm_ctx_set_stream(self._h, C.c_void_p(stream_handle)), "pfm_ctx_set_stream")

    def set_params(self, prm):
        p = prm if isinstance(prm, PfmParams) else PfmParams.from_any(prm)
        self._check(self.lib.pfm_set_params(self._h, C.byref(p)), "pfm_set_params")

    def set_constraints(self, node_flags: np.ndarray):
        f = np.ascontiguousarray(node_flags, np.uint8)
        assert f.size == self.n_nodes
        self._check(self.lib.pfm_set_constraints(self._h, capi.np_ptr(f, np.uint8)), "pfm_set_constraints")

    def pattern_size(self, block: int):
        r, z = C.c_int64(), C.c_int64()
        self._check(self.lib.pfm_pattern_size(self._h, block, C.byref(r), C.byref(z)), "pfm_pattern_size")
        return r.value, z.value

    def pattern(self, block: int):
        rows, nnz = self.pattern_size(block)
        rowptr = np.zeros(rows + 1, np.int64)
        colind = np.zeros(nnz, np.int32)
        self._check(self.lib.pfm_pattern_get(self._h, block, capi.np_ptr(rowptr, np.int64),
                                             capi.np_ptr(colind, np.int32)), "pfm_pattern_get")
        return rowptr, colind

    def pattern_bind(self, block: int, rowptr: np.ndarray, colind: np.ndarray):
        """``pfm_pattern_bind``: adopt the host's CSR arrays of one block (int64 or int32 row pointers)."""
        ci = np.ascontiguousarray(colind, np.int32)
        if np.asarray(rowptr).dtype == np.int32:
            rp = np.ascontiguousarray(rowptr, np.int32)
            rc = self.lib.pfm_pattern_bind_i32(self._h, block, capi.np_ptr(rp, np.int32), capi.np_ptr(ci, np.int32))
        else:
            rp = np.ascontiguousarray(rowptr, np.int64)
            rc = self.lib.pfm_pattern_bind(self._h, block, capi.np_ptr(rp, np.int64), capi.np_ptr(ci, np.int32))
        self._check(rc, "pfm_pattern_bind")

    def halo_exchange(self, comm_handle: int, peer_ranks):
        """``pfm_halo_exchange``: pack -> RCCL send/recv -> unpack inside the library (context's stream)."""
        pr = np.ascontiguousarray(peer_ranks, np.int32)
        self._check(self.lib.pfm_halo_exchange(self._h, C.c_void_p(comm_handle),
                                               capi.np_ptr(pr, np.int32) if pr.size else None), "pfm_halo_exchange")

    def check_finite(self, data_ptr: int, n: int):
        self._check(self.lib.pfm_check_finite(self._h, C.c_void_p(data_ptr), C.c_int64(n)), "pfm_check_finite")

    def state_set_device(self, sol_ptr: int, old_ptr: int, oldold_ptr: int):
        self._check(self.lib.pfm_state_set(self._h, C.c_void_p(sol_ptr), C.c_void_p(old_ptr),
                                           C.c_void_p(oldold_ptr), 1), "pfm_state_set")

    def state_set_solution_device(self, sol_ptr: int):
        """``pfm_state_set_solution``: old / old_old keep the values of the last full scatter (line search)."""
        self._check(self.lib.pfm_state_set_solution(self._h, C.c_void_p(sol_ptr), 1), "pfm_state_set_solution")

    def state_set_host(self, sol: np.ndarray, old: np.ndarray, oldold: np.ndarray):
        a = [np.ascontiguousarray(x, np.float64) for x in (sol, old, oldold)]
        self._check(self.lib.pfm_state_set(self._h, capi.np_ptr(a[0], np.float64), capi.np_ptr(a[1], np.float64),
                                           capi.np_ptr(a[2], np.float64), 0), "pfm_state_set")

    def halo_register(self, send_ptr, send_nodes, recv_ptr, recv_nodes):
        sp = np.ascontiguousarray(send_ptr, np.int64)
        sn = np.ascontiguousarray(send_nodes, np.int32)
        rp = np.ascontiguousarray(recv_ptr, np.int64)
        rn = np.ascontiguousarray(recv_nodes, np.int32)
        self._check(self.lib.pfm_halo_register(self._h, sp.size - 1, capi.np_ptr(sp, np.int64),
                                               capi.np_ptr(sn, np.int32), capi.np_ptr(rp, np.int64),
                                               capi.np_ptr(rn, np.int32)), "pfm_halo_register")
        self.n_peers = int(sp.size - 1)  # contexts with peers have ghost nodes: no fused line-search call for them

    def halo_pack(self, peer: int, buf_ptr: int):
        self._check(self.lib.pfm_halo_pack(self._h, peer, C.c_void_p(buf_ptr)), "pfm_halo_pack")

    def halo_unpack(self, peer: int, buf_ptr: int):
        self._check(self.lib.pfm_halo_unpack(self._h, peer, C.c_void_p(buf_ptr)), "pfm_halo_unpack")

    def halo_pack_all(self, buf_ptr: int):
        self._check(self.lib.pfm_halo_pack_all(self._h, C.c_void_p(buf_ptr)), "pfm_halo_pack_all")

    def halo_unpack_all(self, buf_ptr: int):
        self._check(self.lib.pfm_halo_unpack_all(self._h, C.c_void_p(buf_ptr)), "pfm_halo_unpack_all")

    def assemble_device(self, residual_only: bool, value_ptrs: Sequence[int], res_pde_ptr: int, res_tot_ptr: int):
        arr = (C.c_void_p * 4)(*[C.c_void_p(p) for p in list(value_ptrs) + [0] * (4 - len(value_ptrs))])
        self._check(self.lib.pfm_assemble_device(self._h, 1 if residual_only else 0, arr,
                                                 C.c_void_p(res_pde_ptr), C.c_void_p(res_tot_ptr)),
                    "pfm_assemble_device")

    def assemble_nl_residual_device(self, sol_ptr: int, res_pde_ptr: int, res_tot_ptr: int):
        """``pfm_assemble_nl_residual_device``: solution := sol, then both residuals (the line-search call, cracks.cc:2942-2957);
        single-rank contexts only."""
        self._check(self.lib.pfm_assemble_nl_residual_device(self._h, C.c_void_p(sol_ptr), C.c_void_p(res_pde_ptr), C.c_void_p(res_tot_ptr)),
                    "pfm_assemble_nl_residual_device")

    def assemble_overlapped(self, comm_handle: int, peer_ranks, residual_only: bool, value_ptrs: Sequence[int], res_pde_ptr: int,
                            res_tot_ptr: int):
        """``pfm_assemble_overlapped``: ghost import on the context's side stream next to the interior tiles."""
        pr = np.ascontiguousarray(peer_ranks, np.int32)
        arr = (C.c_void_p * 4)(*[C.c_void_p(p) for p in list(value_ptrs) + [0] * (4 - len(value_ptrs))])
        self._check(self.lib.pfm_assemble_overlapped(self._h, C.c_void_p(comm_handle), capi.np_ptr(pr, np.int32) if pr.size else None,
                                                     1 if residual_only else 0, arr, C.c_void_p(res_pde_ptr),
                                                     C.c_void_p(res_tot_ptr)), "pfm_assemble_overlapped")

    def sync_status(self):
        self._check(self.lib.pfm_sync_status(self._h), "pfm_sync_status")

    def host_register(self, arr: np.ndarray):
        """``pfm_host_register``: page-lock a host array that ``assemble_host(..., out=...)`` reads or writes at every call.
        The caller keeps the array alive until ``host_unregister`` / ``close``."""
        self._check(self.lib.pfm_host_register(self._h, C.c_void_p(arr.ctypes.data), arr.nbytes), "pfm_host_register")

    def host_unregister(self, arr: np.ndarray = None):
        self._check(self.lib.pfm_host_unregister(self._h, C.c_void_p(arr.ctypes.data) if arr is not None else None), "pfm_host_unregister")

    def assemble_host(self, sol, old, oldold, residual_only: bool, out=None):
        """``pfm_assemble``: synchronous, host numpy in / host numpy out (single rank).  ``out`` = (values, res_pde,
        res_tot): arrays of an earlier call to write into again (what a host with its own matrix storage does; with
        ``host_register`` on them the transfers are DMA from page-locked memory)."""
        sol = np.ascontiguousarray(sol, np.float64)
        old = np.ascontiguousarray(old, np.float64)
        oldold = np.ascontiguousarray(oldold, np.float64)
        n = self.n_owned_dofs
        assert sol.size == old.size == oldold.size == n
        res_pde = out[1] if out is not None else np.zeros(n)
        res_tot = (out[2] if out is not None and out[2] is not None else np.zeros(n)) if residual_only else None
        values: List[np.ndarray] = []
        ptrs = (C.c_void_p * 4)()
        if not residual_only:
            for b in range(self.n_blocks):
                values.append(out[0][b] if out is not None else np.zeros(self.pattern_size(b)[1]))
                ptrs[b] = values[b].ctypes.data
        rc = self.lib.pfm_assemble(self._h, capi.np_ptr(sol, np.float64), capi.np_ptr(old, np.float64),
                                   capi.np_ptr(oldold, np.float64), 1 if residual_only else 0,
                                   None if residual_only else ptrs, capi.np_ptr(res_pde, np.float64),
                                   capi.np_ptr(res_tot, np.float64) if residual_only else None)
        self._check(rc, "pfm_assemble")
        return values, res_pde, res_tot

    def timing_enable(self, on: bool = True):
        self._check(self.lib.pfm_timing_enable(self._h, 1 if on else 0), "pfm_timing_enable")

    def kernel_time_ms(self):
        ms, n = C.c_double(), C.c_int()
        self._check(self.lib.pfm_kernel_time_ms(self._h, C.byref(ms), C.byref(n)), "pfm_kernel_time_ms")
        return ms.value, n.value

    def kernel_times_ms(self, capacity: int = 4096) -> np.ndarray:
        """Durations of the recorded launches (call before ``kernel_time_ms``, which resets the record)."""
        buf = np.zeros(capacity)
        n = C.c_int()
        self._check(self.lib.pfm_kernel_times_ms(self._h, capi.np_ptr(buf, np.float64), capacity, C.byref(n)), "pfm_kernel_times_ms")
        return buf[:min(n.value, capacity)]

    @property
    def kernel_path(self) -> int:
        return self.lib.pfm_ctx_kernel_path(self._h)

    def force_path(self, path: int):
        self._check(self.lib.pfm_ctx_force_path(self._h, path), "pfm_ctx_force_path")

    def overlay_info(self):
        """(rows written by the patch kernel of the cartesian overlay, cells left to the general family)."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.pfm_ctx_overlay_info(self._h, C.byref(a), C.byref(b)), "pfm_ctx_overlay_info")
        return int(a.value), int(b.value)

    def force_phase(self, phase: int):
        """measurement: 1 / 2 = only the first / second half of the overlapped assembly, 0 = all."""
        self._check(self.lib.pfm_ctx_force_phase(self._h, phase), "pfm_ctx_force_phase")

    @property
    def device_bytes(self) -> int:
        return int(self.lib.pfm_ctx_device_bytes(self._h))

    # ---- include/pfm_newton.h
    def diag_mass_device(self, mass_ptr: int):
        """cracks.cc:2514-2562 into a device vector over the owned nodes."""
        self._check(self.lib.pfm_diag_mass_device(self._h, C.c_void_p(mass_ptr)), "pfm_diag_mass_device")

    def active_set_device(self, res_tot_ptr: int, mass_ptr: int, c: float, sol_ptr: int, old_ptr: int, cycle_ptr: int):
        """cracks.cc:2837-2909; returns (active, cycling, changed)."""
        counts = (C.c_int64 * 3)()
        self._check(self.lib.pfm_active_set_device(self._h, C.c_void_p(res_tot_ptr), C.c_void_p(mass_ptr), C.c_double(c),
                                                   C.c_void_p(sol_ptr), C.c_void_p(old_ptr), C.c_void_p(cycle_ptr), counts),
                    "pfm_active_set_device")
        return int(counts[0]), int(counts[1]), int(counts[2])

    def get_constraints(self) -> np.ndarray:
        flags = np.empty(self.n_nodes, np.uint8)
        self._check(self.lib.pfm_get_constraints(self._h, capi.np_ptr(flags, np.uint8)), "pfm_get_constraints")
        return flags

    def residual_norms(self, res_ptr: int):
        """(l2, linf, sum of squares) of a device residual vector with the constrained lines zeroed -- what the Newton loop
        and the line search read after every assembly (cracks.cc:2791-2794, 2947-2949); this rank's owned dofs."""
        out = (C.c_double * 3)()
        self._check(self.lib.pfm_residual_norms(self._h, C.c_void_p(res_ptr), out), "pfm_residual_norms")
        return float(out[0]), float(out[1]), float(out[2])

    def functionals(self, cell_owned: Optional[np.ndarray] = None, cell_lambda=None, cell_mu=None):
        """(bulk energy, crack energy, TCV) of the node state in the context (cracks.cc:3553-3701); optional per-cell
        Lame coefficients for the energy (the reference's heterogeneous case uses other ones than the assembly)."""
        out = (C.c_double * 3)()
        mask = None if cell_owned is None else np.ascontiguousarray(cell_owned, np.uint8)
        mp = None if mask is None else capi.np_ptr(mask, np.uint8)
        if cell_lambda is None:
            self._check(self.lib.pfm_functionals(self._h, mp, out), "pfm_functionals")
        else:
            la = np.ascontiguousarray(cell_lambda, np.float64)
            mu = np.ascontiguousarray(cell_mu, np.float64)
            self._check(self.lib.pfm_functionals_material(self._h, mp, capi.np_ptr(la, np.float64),
                                                          capi.np_ptr(mu, np.float64), out), "pfm_functionals_material")
        return float(out[0]), float(out[1]), float(out[2])


class Assembler:
    """Device-resident counterpart of the reference's assembly members.

    ``solution`` / ``old_solution`` / ``old_old_solution`` are torch device vectors over
    the owned dofs; ``assemble_system`` / ``assemble_nl_residual`` fill
    ``system_pde_matrix`` (list of CSR value vectors, one per block),
    ``system_pde_residual`` and ``system_total_residual`` on the device, asynchronously
    on torch's current stream."""

    def __init__(self, mesh, blocked: bool, device: int = 0, n_owned_nodes: Optional[int] = None,
                 cell_lambda=None, cell_mu=None, halo=None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("cracks_amd.Assembler needs a ROCm GPU (there is no CPU fallback)")
        self.torch = torch
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.ctx = Context(mesh, blocked, device, n_owned_nodes, cell_lambda, cell_mu)
        self.ctx.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        n = self.ctx.n_owned_dofs
        z = lambda: torch.zeros(n, dtype=torch.float64, device=self.dev)
        self.solution, self.old_solution, self.old_old_solution = z(), z(), z()
        self.system_pde_residual, self.system_total_residual = z(), z()
        self.system_pde_matrix: List = []
        self.halo = halo

    def allocate_matrix(self):
        if not self.system_pde_matrix:
            t = self.torch
            self.system_pde_matrix = [t.empty(self.ctx.pattern_size(b)[1], dtype=t.float64, device=self.dev)
                                      for b in range(self.ctx.n_blocks)]

    def set_params(self, prm):
        self.ctx.set_params(prm)

    def set_constraints(self, node_flags: np.ndarray):
        self.ctx.set_constraints(node_flags)

    def set_vectors(self, sol: np.ndarray, old: np.ndarray, oldold: np.ndarray):
        t = self.torch
        self.solution.copy_(t.from_numpy(np.ascontiguousarray(sol)))
        self.old_solution.copy_(t.from_numpy(np.ascontiguousarray(old)))
        self.old_old_solution.copy_(t.from_numpy(np.ascontiguousarray(oldold)))

    def assemble_system(self, residual_only: bool = False, solution_only: bool = False):
        """cracks.cc:2129-2475 (without the AMG set-up that follows it).  ``solution_only``: only ``solution`` changed
        since the last call (the line search of cracks.cc:2942-2957 and the Newton iterations within a time step):
        old_solution / old_old_solution are not scattered again."""
        self.ctx.set_stream(self.torch.cuda.current_stream(self.dev).cuda_stream)
        if solution_only and residual_only and self.halo is None and getattr(self.ctx, "n_peers", 0) == 0:
            # one library call; on a single-rank box the residual kernel reads `solution` itself (no scatter launch).  The
            # context's own state decides, not this wrapper's: a caller may register peers with ctx.halo_register and move
            # the ghosts itself (pack / unpack), and pfm_assemble_nl_residual_device refuses such contexts
            self.ctx.assemble_nl_residual_device(self.solution.data_ptr(), self.system_pde_residual.data_ptr(),
                                                 self.system_total_residual.data_ptr())
            return
        if solution_only:
            self.ctx.state_set_solution_device(self.solution.data_ptr())
        else:
            self.ctx.state_set_device(self.solution.data_ptr(), self.old_solution.data_ptr(),
                                      self.old_old_solution.data_ptr())
        if not residual_only:
            self.allocate_matrix()
        ptrs = [m.data_ptr() for m in self.system_pde_matrix] if not residual_only else []
        # PFM_OVERLAP=1: the ghost import on the context's side stream next to the interior tiles.  Off by default: on one
        # node the import is ~0.05 ms (pack + unpack 0.02 ms measured, ~7 messages of <= 0.6 MB over xGMI) while cutting
        # the first kernel into an interior and a boundary launch costs 0.12 ms at 8 ranks (profiles/r03/rank_share_w8.json)
        lib_comm = self.halo.library_comm(self.ctx) if (self.halo is not None and os.environ.get("PFM_OVERLAP") == "1") else None
        if lib_comm is not None:
            # one library call: exchange on the side stream || interior tiles, then the tiles that read ghost nodes
            self.ctx.assemble_overlapped(lib_comm, self.halo.peers, residual_only, ptrs, self.system_pde_residual.data_ptr(),
                                         self.system_total_residual.data_ptr())
            return
        if self.halo is not None:
            self.halo.exchange(self.ctx)
        self.ctx.assemble_device(residual_only, ptrs, self.system_pde_residual.data_ptr(), self.system_total_residual.data_ptr())

    def assemble_nl_residual(self, solution_only: bool = False):
        """cracks.cc:2507-2512."""
        self.assemble_system(True, solution_only)

    def residual_norm(self, total: bool = False) -> float:
        """constraints_update.set_zero(residual); residual.l2_norm() (cracks.cc:2791-2794, 2947-2949) of the last assembly, without
        moving the vector: 24 bytes come back.  Synchronous (as the reference's call)."""
        r = self.system_total_residual if total else self.system_pde_residual
        return self.ctx.residual_norms(r.data_ptr())[0]

    def synchronize(self):
        """Wait for the stream and raise what the reference would have thrown/aborted on."""
        self.ctx.sync_status()
