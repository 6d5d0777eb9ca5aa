#!/usr/bin/env python3
"""bench.py — assembled DoFs/s of the Newton Jacobian+residual assembly (BASELINE.json metric).

One "step" = one ``assemble_system()`` (cracks.cc:2133-2475: zero outputs, ghost import,
cell integration, constrained scatter) on a synthetic uniformly refined 3-D Sneddon mesh
(BASELINE.json configs[2]: ~1e7 hexes, Q1/Q1, full Jacobian + residual into the 2x2 block
CSR).  Inputs are resident in HBM when the timed region starts; outputs stay on the device.

  python bench.py                       # N=1, 216^3 cells
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N   # strong scaling: the same mesh cut into N sub-boxes,
                                             # ghost values exchanged over RCCL (cracks_amd/halo.py)

Prints ONE JSON line on rank 0 (contract in the task description) with the extra objects
"roofline" (dominant kernel, HIP-event timed, algorithmic bytes of SURVEY.md §8(d)) and
"cpu_baseline" (the CPU oracle = loop-for-loop port of the reference, timed on this host).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL peer-to-peer between the ranks of a node

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
N_SIMD = 1024  # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4  # max clock (the chip sustains ~1.6-2.1 GHz under FP64 load: profiles/r02/microbench_mfma_f64_lds.txt)


def kernel_source_hash() -> str:
    """sha1 over the kernel sources: profiles/ summaries carry the hash they were measured with, and bench.py refuses to
    quote counters of other kernels (a stale profile goes into the JSON as null, with a warning on stderr)."""
    import hashlib

    h = hashlib.sha1()
    d = os.path.join(ROOT, "cracks_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def _profile_record(key: str):
    import glob

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", key)))
    if not found:
        return None, None
    rec = json.load(open(found[-1]))
    src = os.path.relpath(found[-1], ROOT)
    want = rec.get("kernel_source_hash")
    if want is not None and want != kernel_source_hash():
        print(f"bench.py: WARNING {src} was measured with other kernel sources (hash {want}): not quoted; "
              f"re-run tools/profile_round.sh", file=sys.stderr)
        return None, src + " (STALE: kernels changed since)"
    return rec, src


def measured_valu_instructions(dim: int, n: int, residual_only: bool):
    """Second bound, from counters instead of a flop estimate: VALU wave-instructions per assembly and how many of them
    are FP64 arithmetic (SQ_INSTS_VALU, SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 of a separate rocprofv3 PMC pass of this very
    command, tools/profile_round.sh; summary committed under profiles/).  On gfx950 EVERY VALU instruction occupies its
    SIMD for a 4-cycle issue slot (measured: an integer or move instruction costs what an FP64 FMA costs), so
    N_valu x 4 cycles / (4 SIMDs x CUs) is the issue floor of the instruction stream as it is, and N_fp64 x 4 cycles
    the floor of its arithmetic alone.  None if this workload was not profiled with the current kernels."""
    rec, src = _profile_record(f"instruction_mix_{dim}d_{n}{'_residual' if residual_only else ''}.json")
    if rec is None:
        return None, src
    tot = fp64 = 0.0
    for name, d in rec["per_launch"].items():
        if not name.startswith(("k_cart", "k_state_set", "k_assemble_general")):
            continue  # context creation (k_aos_to_soa, k_lattice_masks, k_graph_*, k_build_cslot) is not part of an assembly
        tot += d.get("SQ_INSTS_VALU", 0.0)
        fp64 += d.get("SQ_INSTS_VALU_ADD_F64", 0.0) + d.get("SQ_INSTS_VALU_MUL_F64", 0.0) + d.get("SQ_INSTS_VALU_FMA_F64", 0.0)
    return {"valu": tot, "fp64": fp64}, src


def algorithmic_bytes_per_cell(dim: int, residual_only: bool) -> float:
    """Unique-touch model of SURVEY.md §8(d): every global datum read once, every output
    written once, ~1 node per cell."""
    if dim == 3:
        return 168.0 if residual_only else 3592.0
    return 120.0 if residual_only else 744.0


def measured_hbm_traffic(dim: int, n: int, residual_only: bool):
    """HBM bytes per assembly from the PMC counters (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes of this very
    command, tools/profile_round.sh; summaries committed under profiles/).  None if this workload was not profiled with
    the current kernels."""
    rec, src = _profile_record(f"hbm_traffic_{dim}d_{n}{'_residual' if residual_only else ''}.json")
    if rec is None:
        return None, src
    total = 0.0
    for name, d in rec["per_launch"].items():
        if residual_only and name not in ("k_cart_residual3", "k_cart_residual3d", "k_cart_residual3x", "k_cart_residual2m", "k_state_set", "k_state_set_solution"):
            continue
        total += d.get("write_bytes", 0.0) + d.get("fetch_bytes", 0.0)
    return total, src


def sneddon_params(h: float, dim: int):
    from cracks_amd.capi import PfmParams

    E, nu = 1.0, 0.2
    mu = E / (2.0 * (1 + nu))
    lam = (2 * nu * mu) / (1.0 - 2 * nu)
    kappa = 1.0e-8 * h if dim == 2 else 0.0  # parameters_sneddon_2d.prm / _3d.prm
    return PfmParams(lambda_=lam, mu=mu, G_c=1.0, alpha_eps=2.0 * h, constant_k=kappa, pressure=1.0e-3,
                     alpha_biot=0.0, gamma_penal=0.0, timestep=1.0, time=1.0, old_timestep=1.0,
                     old_old_timestep=1.0, decompose_stress_rhs=0.0, decompose_stress_matrix=0.0,
                     timestep_number=0, outer_solver=0, use_old_timestep_pf=0, reserved=0)


def synthetic_state(mesh, global_ids, h, dim, seed=1234):
    """Interpolated InitialValuesSneddon + seeded perturbation; values are a
    function of the global node id so that every rank count sees the same field."""
    from cracks_amd.mesh import initial_values_sneddon

    def noise(salt, lo, hi):
        x = (global_ids.astype(np.uint64) + np.uint64(seed + 7919 * salt)) * np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(29)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(32)
        return lo + (hi - lo) * (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)

    phi0 = initial_values_sneddon(mesh, h)
    u = np.stack([noise(d, -1e-3, 1e-3) for d in range(dim)], axis=1)
    on_boundary = np.zeros(mesh.n_nodes, bool)
    for nodes in mesh.boundary_nodes.values():
        on_boundary[nodes] = True
    u[on_boundary] = 0.0
    phi = np.clip(phi0 + noise(10, -0.2, 0.2), 0.0, 1.0)
    phi_old = np.clip(phi0 + noise(11, -0.2, 0.2), 0.0, 1.0)
    phi_oldold = np.clip(phi0 + noise(12, -0.2, 0.2), 0.0, 1.0)
    flags = np.where(on_boundary, (1 << dim) - 1, 0).astype(np.uint8)  # u = 0 on the boundary (cracks.cc:2686-2694)
    return u, phi, phi_old, phi_oldold, flags


def cpu_baseline(dim: int, residual_only: bool, target_seconds: float = 15.0):
    """Time the CPU oracle (port of cracks.cc:2200-2467) on a bounded sample of the same
    workload, one thread = one reference MPI rank (cracks.cc:4587)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import subprocess

    import oracle_api as O
    from cracks_amd import mesh as M

    # -march=native must mean THIS host: rebuild the oracle here (the .so in the tree comes from the build container)
    build_note = "prebuilt liboracle.so (no compiler on this host)"
    try:
        O.build_oracle(force=True)
        mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
        cxxflags = [ln.split("?=", 1)[1].strip() for ln in mk.splitlines() if ln.startswith("CXXFLAGS")][0]
        cxx = subprocess.run(["g++", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
        build_note = f"rebuilt on this host: {cxx}, {cxxflags}"
    except Exception as e:  # pragma: no cover
        build_note += f" [{type(e).__name__}]"

    n = 20 if dim == 3 else 160
    if residual_only and dim == 3:
        n = 40
    mesh = M.box_mesh(dim, n)
    h = mesh.min_cell_diameter()
    lay = M.DofLayout(mesh.n_nodes, dim, blocked=True)
    u, phi, po, poo, flags = synthetic_state(mesh, np.arange(mesh.n_nodes), h, dim)
    sol = lay.pack(u, phi)
    old = lay.pack(np.zeros_like(u), po)
    oo = lay.pack(np.zeros_like(u), poo)
    dd = M.sneddon_dirichlet_dofs(mesh, lay)
    cu = M.update_constraints(mesh, lay, dd)
    ch = M.hanging_constraints(mesh, lay)
    prm = O.PfmParams.from_buffer_copy(bytes(sneddon_params(h, dim)))
    rowptr = colind = None
    if not residual_only:
        rowptr, colind = M.dof_sparsity(mesh, lay)
    O.assemble(mesh, lay, prm, sol, old, oo, cu, ch, residual_only, rowptr, colind)  # warm
    times = []
    t_start = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_start < target_seconds and len(times) < 50):
        t0 = time.perf_counter()
        r = O.assemble(mesh, lay, prm, sol, old, oo, cu, ch, residual_only, rowptr, colind)
        times.append(time.perf_counter() - t0)
        assert r.err == 0
    t = float(np.median(times))
    # (ii) all cores: one thread per core, each assembling its own copy of the sample -- the stand-in for
    # `mpirun -n N` of the reference, whose ranks assemble their own cells independently; the oracle
    # call releases the GIL
    import threading

    try:
        ncore = len(os.sched_getaffinity(0))
    except AttributeError:
        ncore = os.cpu_count() or 1
    quota_note = ""
    try:  # container CPU quota (cgroup v2): the cores this process may actually use
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            ncore = max(1, min(ncore, int(float(q) / float(per))))
            quota_note = f" (cgroup quota {q}/{per})"
    except (OSError, ValueError):
        pass
    reps = max(1, int(round(6.0 / max(t, 1e-3))))
    barrier = threading.Barrier(ncore + 1)
    errs = []

    def worker():
        barrier.wait()
        for _ in range(reps):
            rr = O.assemble(mesh, lay, prm, sol, old, oo, cu, ch, residual_only, rowptr, colind)
            if rr.err != 0:
                errs.append(rr.err)
        barrier.wait()

    threads = [threading.Thread(target=worker) for _ in range(ncore)]
    for th in threads:
        th.start()
    barrier.wait()
    t_all0 = time.perf_counter()
    barrier.wait()
    t_all = time.perf_counter() - t_all0
    for th in threads:
        th.join()
    assert not errs
    # (iii) a second sample size on the all-cores leg ("time the CPU on 1e5-1e6 cells and report
    # DoFs/s, size-independent"): the same mesh kind at 46^3 = 97 336 cells (3-D) / 320^2 (2-D) SHARED by the threads is not
    # what the oracle offers (one call = one mesh, serial), so every thread assembles its own copy once -- the DoFs/s of
    # that leg next to the small sample's shows whether the figure depends on the working set (cache-resident 20^3
    # against 46^3: 37 k against 4.1e5 DoFs, 2x2 block CSR of 44 MB per thread)
    second = None
    try:
        n2 = (46 if not residual_only else 64) if dim == 3 else 320
        mesh2 = M.box_mesh(dim, n2)
        h2 = mesh2.min_cell_diameter()
        lay2 = M.DofLayout(mesh2.n_nodes, dim, blocked=True)
        u2, phi2, po2, poo2, _ = synthetic_state(mesh2, np.arange(mesh2.n_nodes), h2, dim)
        sol2, old2, oo2 = lay2.pack(u2, phi2), lay2.pack(np.zeros_like(u2), po2), lay2.pack(np.zeros_like(u2), poo2)
        cu2 = M.update_constraints(mesh2, lay2, M.sneddon_dirichlet_dofs(mesh2, lay2))
        ch2 = M.hanging_constraints(mesh2, lay2)
        prm2 = O.PfmParams.from_buffer_copy(bytes(sneddon_params(h2, dim)))
        rp2 = ci2 = None
        if not residual_only:
            rp2, ci2 = M.dof_sparsity(mesh2, lay2)
        nth2 = min(ncore, 32)  # (0.35 GB of CSR values per thread)
        barrier2 = threading.Barrier(nth2 + 1)
        errs2 = []

        def worker2():
            barrier2.wait()
            rr = O.assemble(mesh2, lay2, prm2, sol2, old2, oo2, cu2, ch2, residual_only, rp2, ci2)
            if rr.err != 0:
                errs2.append(rr.err)
            barrier2.wait()

        th2 = [threading.Thread(target=worker2) for _ in range(nth2)]
        for th in th2:
            th.start()
        barrier2.wait()
        t20 = time.perf_counter()
        barrier2.wait()
        t2 = time.perf_counter() - t20
        for th in th2:
            th.join()
        assert not errs2
        second = {"value": nth2 * lay2.n_dofs / t2, "unit": "DoFs/s", "cores": nth2,
                  "sample": f"{nth2} threads x 1 assembly of {n2}^{dim} cells ({lay2.n_dofs} DoFs) each, wall {t2:.1f} s"}
    except Exception as e:  # the second size must not take the baseline away
        second = {"error": f"{type(e).__name__}: {e}"}
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": lay.n_dofs / t, "unit": "DoFs/s", "cores": 1, "kind": "port",
            "all_cores": {"value": ncore * reps * lay.n_dofs / t_all, "unit": "DoFs/s", "cores": ncore,
                          "sample": f"{ncore} threads{quota_note} x {reps} assemblies of the same sample each, wall {t_all:.1f} s"},
            "all_cores_second_size": second,
            "sample": f"{n}^{dim} cells ({lay.n_dofs} DoFs), median of {len(times)} assemblies, "
                      f"{build_note}, 1 thread of {os.cpu_count()} ({model}); excludes Trilinos "
                      f"insertion overhead the real reference pays"}


def time_mode(asm, dev, residual_only: bool, steps: int, warmup: int):
    """ms per call (wall, barrier-free single rank) and mean kernel-group ms (HIP events) of one assembly mode."""
    import torch

    first = [True]

    def call():
        asm.assemble_system(residual_only, solution_only=residual_only and not first[0])
        first[0] = False

    for _ in range(warmup):
        call()
    asm.synchronize()
    # two passes: the wall time per call WITHOUT the library's event pair around every kernel group (two event records
    # cost ~10 us per call: nothing at 12 ms, a quarter of a 35 us residual call), then the kernel-group time with it
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        call()
    torch.cuda.synchronize(dev)
    wall = (time.perf_counter() - t0) / steps * 1e3
    asm.synchronize()
    # GPU time per call: ONE event pair around the whole batch on the stream the calls are enqueued on, divided by the
    # number of calls -- it cannot exceed the wall time per call by more than the timer noise, which a pair per call can
    # (two event records cost ~10 us: more than a quarter of a 35 us kernel)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        call()
    e1.record()
    e1.synchronize()
    asm.synchronize()
    k_ms = min(e0.elapsed_time(e1) / steps, wall)
    return wall, k_ms


def extra_lines(asm3, dev, local_rank, n3: int, steps: int):
    """The other lines of SURVEY.md 8(d) in the same run (a few seconds): residual-only at the headline size (the
    line-search call of cracks.cc:2942-2957), BASELINE config 2 (2-D 1000^2 residual-only) and the 2-D Jacobian, each
    with its fraction of the HBM roofline and, where the kernels were profiled, the counted FP64 bound."""
    from cracks_amd import partition as P
    from cracks_amd.assembler import Assembler

    out = {}

    def record(key, dim, n, residual_only, wall, k_ms, n_cells, n_dofs):
        ab = algorithmic_bytes_per_cell(dim, residual_only) * n_cells
        rec = {"ms_per_call": wall, "kernel_ms": k_ms, "DoFs_per_s": n_dofs / (wall * 1e-3),
               "hbm_frac": ab / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else None,
               "algorithmic_bytes_per_cell": algorithmic_bytes_per_cell(dim, residual_only)}
        if residual_only:
            # the line-search call (cracks.cc:2942-2957): only `solution` is scattered again between two residuals
            # pfm_assemble_nl_residual_device: on a single rank the residual kernel reads `solution` itself
            rec["state_scatter"] = "solution only, read by the residual kernel (pfm_assemble_nl_residual_device)"
            # host time per call beyond the GPU time per call (kernel_ms: one event pair around the batch, see time_mode)
            rec["launch_overhead_ms"] = max(0.0, wall - k_ms)
        vi, src = measured_valu_instructions(dim, n, residual_only)
        if vi is not None:
            to_ms = 4.0 / N_SIMD / (CLOCK_GHZ * 1e9) * 1e3
            rec["fp64_bound"] = {"fp64_wave_instructions": vi["fp64"], "valu_wave_instructions": vi["valu"],
                                 "fp64_share": vi["fp64"] / vi["valu"] if vi["valu"] else None,
                                 "fp64_floor_ms_at_max_clock": vi["fp64"] * to_ms,
                                 "issue_floor_ms_at_max_clock": vi["valu"] * to_ms, "source": src}
        out[key] = rec

    n_nodes3 = (n3 + 1) ** 3
    wall, k_ms = time_mode(asm3, dev, True, steps, 2)
    record("residual_only_3d", 3, n3, True, wall, k_ms, n3 ** 3, 4 * n_nodes3)
    # what the line search actually consumes (cracks.cc:2946-2949): ||set_zero(residual)||_2.  pfm_residual_norms leaves the
    # vector on the device and returns 24 bytes; the call is synchronous like the reference's l2_norm()
    asm3.assemble_nl_residual(solution_only=True)
    nrm = asm3.residual_norm()
    torch_sync = __import__("torch").cuda.synchronize
    torch_sync(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        asm3.assemble_nl_residual(solution_only=True)
        asm3.residual_norm()
    torch_sync(dev)
    per = (time.perf_counter() - t0) / steps
    out["residual_only_3d"]["line_search_call"] = {
        "what": "pfm_assemble_nl_residual_device + pfm_residual_norms, synchronous; residual stays on the device",
        "ms_per_call": per * 1e3, "bytes_to_host": 24, "residual_l2": nrm, "DoFs_per_s": 4 * n_nodes3 / per}
    n2 = 1000
    lp = P.build_local_problem(2, (n2, n2), P.factor_ranks(1, 2), 0)
    h = (20.0 / n2) * np.sqrt(2)
    u, phi, po, poo, flags = synthetic_state(lp.mesh, lp.global_ids, h, 2)
    a2 = Assembler(lp.mesh, blocked=True, device=local_rank, n_owned_nodes=lp.n_owned)
    a2.set_params(sneddon_params(h, 2))
    a2.set_constraints(flags)
    no = lp.n_owned

    def pack(uu, pp):
        v = np.empty(no * 3)
        v[:no * 2] = uu[:no].reshape(-1)
        v[no * 2:] = pp[:no]
        return v

    a2.set_vectors(pack(u, phi), pack(np.zeros_like(u), po), pack(np.zeros_like(u), poo))
    wall, k_ms = time_mode(a2, dev, True, 4 * steps, 3)
    record("config2_residual_only_2d", 2, n2, True, wall, k_ms, n2 * n2, 3 * (n2 + 1) ** 2)
    wall, k_ms = time_mode(a2, dev, False, steps, 2)
    record("jacobian_2d", 2, n2, False, wall, k_ms, n2 * n2, 3 * (n2 + 1) ** 2)
    a2.ctx.close()
    try:
        out["general_family_3d"] = general_family_3d(dev, local_rank, steps)
    except Exception as e:
        out["general_family_3d"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        out["overlay_3d"] = overlay_3d(dev, local_rank, steps)
    except Exception as e:
        out["overlay_3d"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        out["config5_standin"] = config5_standin(dev, local_rank, steps)
    except Exception as e:  # a missing mesh helper must not take the other lines away
        out["config5_standin"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def general_family_3d(dev, local_rank, steps: int, n: int = 100):
    """What a 3-D mesh with hanging nodes or general hexes costs (cracks.cc:2200-2203 with MappingQ1 at every q-point,
    scatter through colour classes): the Sneddon box of `n`^3 hexes FORCED onto the general family
    (`pfm_ctx_force_path(0)`; the same line as `python bench.py --n 100 --path general`)."""
    from cracks_amd import partition as P
    from cracks_amd.assembler import Assembler

    lp = P.build_local_problem(3, (n, n, n), P.factor_ranks(1, 3), 0)
    h = (20.0 / n) * np.sqrt(3)
    u, phi, po, poo, flags = synthetic_state(lp.mesh, lp.global_ids, h, 3)
    a = Assembler(lp.mesh, blocked=True, device=local_rank, n_owned_nodes=lp.n_owned)
    a.ctx.force_path(0)
    a.set_params(sneddon_params(h, 3))
    a.set_constraints(flags)
    no = lp.n_owned

    def pack(uu, pp):
        v = np.empty(no * 4)
        v[:no * 3] = uu[:no].reshape(-1)
        v[no * 3:] = pp[:no]
        return v

    a.set_vectors(pack(u, phi), pack(np.zeros_like(u), po), pack(np.zeros_like(u), poo))
    n_cells, n_dofs = n ** 3, 4 * (n + 1) ** 3
    rec = {"workload": f"Sneddon 3D, {n}^3 hexes forced onto the general family (kernel path {a.ctx.kernel_path})", "cells": n_cells}
    for key, ro in (("jacobian", False), ("residual_only", True)):
        wall, k_ms = time_mode(a, dev, ro, max(3, steps // 2), 2)
        rec[key] = {"ms_per_call": wall, "kernel_ms": k_ms, "DoFs_per_s": n_dofs / (wall * 1e-3),
                    "hbm_frac": algorithmic_bytes_per_cell(3, ro) * n_cells / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else None}
    a.ctx.close()
    return rec


def overlay_3d(dev, local_rank, steps: int, n: int = 84):
    """A 3-D AMR mesh (cracks.cc:3895-4163 refine_mesh is dimension independent; tests/hetero_3d_1.prm): the Sneddon box
    of `n`^3 hexes with its inner half refined once -- hanging nodes on the faces and edges of the block.  The regular rows
    of both refinement levels are written by the cartesian row-owner kernels on the level lattices (kernel path 3), the
    general family keeps the cells at hanging nodes, level seams and the boundary; next to it the same mesh through the
    general family alone (`pfm_ctx_force_path(0)`)."""
    from cracks_amd import mesh as M
    from cracks_amd.assembler import Assembler

    t0 = time.perf_counter()
    g0 = M.box_mesh(3, (n,) * 3)
    cc = g0.coords[g0.cells].mean(axis=1)
    mesh = M.refine_cells(g0, (np.abs(cc) < 5.0).all(axis=1))
    t_mesh = time.perf_counter() - t0
    h = (20.0 / n) * np.sqrt(3.0) / 2
    u, phi, po, poo, flags = synthetic_state(mesh, np.arange(mesh.n_nodes), h, 3)
    flags[mesh.hn_nodes] = 0
    a = Assembler(mesh, blocked=True, device=local_rank)
    a.set_params(sneddon_params(h, 3))
    a.set_constraints(flags)
    pack = lambda uu, pp: np.concatenate([uu.reshape(-1), pp])
    a.set_vectors(pack(u, phi), pack(0 * u, po), pack(0 * u, poo))
    rows, general_cells = a.ctx.overlay_info()
    rec = {"workload": f"Sneddon 3D, {n}^3 hexes with the inner half refined once: {mesh.n_cells} cells, {mesh.n_nodes} nodes, "
                       f"{mesh.hn_nodes.size} hanging nodes", "cells": mesh.n_cells, "kernel_path": a.ctx.kernel_path,
           "regular_rows": rows, "regular_row_fraction": rows / mesh.n_nodes, "cells_left_to_the_general_family": general_cells,
           "ctx_create_s": round(a.ctx.create_seconds, 3), "mesh_build_s_python": round(t_mesh, 1)}
    n_dofs = 4 * mesh.n_nodes
    for tag, path in (("overlay", None), ("general_family_alone", 0)):
        if path is not None:
            a.ctx.force_path(path)
        for key, ro in (("jacobian", False), ("residual_only", True)):
            wall, k_ms = time_mode(a, dev, ro, max(3, steps // 2), 2)
            rec[f"{tag}_{key}"] = {"ms_per_call": wall, "kernel_ms": k_ms, "ns_per_cell": 1e6 * k_ms / mesh.n_cells, "DoFs_per_s": n_dofs / (wall * 1e-3),
                                   "hbm_frac": algorithmic_bytes_per_cell(3, ro) * mesh.n_cells / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else None}
    a.ctx.close()
    return rec


def config5_problem(levels: int, step: int):
    """Stand-in for BASELINE config 5 (Miehe shear with adaptive refinement, cracks.cc:4137-4174): unit-slit square,
    2^(levels+1) cells per edge, a band of refined cells with hanging nodes along the crack that grows with `step`,
    one-block (direct solver) layout, stress split active.  Returns mesh, layout, parameters, node flags and the three
    vectors."""
    from cracks_amd import mesh as M
    from cracks_amd.assembler import node_flags_from_dof_flags
    from cracks_amd.capi import PfmParams

    base = M.slit_mesh(levels)
    n = 2 ** (levels + 1)
    h = 1.0 / n
    tip = 0.5 - 0.08 * step  # the band follows a crack growing to the left
    cc = base.coords[base.cells].mean(axis=1)
    flags = (np.abs(cc[:, 1] - 0.5) < 6 * h) & (cc[:, 0] > tip - 4 * h)
    t0 = time.perf_counter()
    mesh = M.refine_cells(base, flags)
    t_refine = time.perf_counter() - t0
    lay = M.DofLayout(mesh.n_nodes, 2, blocked=False)
    hfine = 0.5 * h
    dt = 1.0e-4
    prm = PfmParams(lambda_=121.15e3, mu=80.77e3, G_c=2.7, alpha_eps=2.0 * hfine * np.sqrt(2.0),
                    constant_k=1.0e-10 * hfine, pressure=0.0, alpha_biot=0.0, gamma_penal=0.0, timestep=dt,
                    time=5 * dt, old_timestep=dt, old_old_timestep=dt, decompose_stress_rhs=1.0,
                    decompose_stress_matrix=1.0, timestep_number=5, outer_solver=0, use_old_timestep_pf=0,
                    reserved=0)
    ch = M.hanging_constraints(mesh, lay)
    cu = M.update_constraints(mesh, lay, M.miehe_shear_dirichlet_dofs(mesh, lay))
    # state: shear ramp + noise, phase field with a smeared crack along the slit line
    rng = np.random.default_rng(1234 + step)
    x, y = mesh.coords[:, 0], mesh.coords[:, 1]
    u = np.stack([-5 * dt * y + 1e-6 * rng.standard_normal(x.size), 1e-6 * rng.standard_normal(x.size)], axis=1)
    phi = np.clip(1.0 - np.exp(-np.abs(y - 0.5) / (4 * hfine)) * (x > tip), 0.0, 1.0)
    sol = ch.distribute(lay.pack(u, phi))
    old = ch.distribute(lay.pack(0.9 * u, np.clip(phi + 0.01 * rng.random(x.size), 0, 1)))
    oldold = ch.distribute(lay.pack(0.8 * u, np.clip(phi + 0.02 * rng.random(x.size), 0, 1)))
    return {"mesh": mesh, "layout": lay, "params": prm, "cu": cu, "ch": ch, "refine_host_s": t_refine,
            "node_flags": node_flags_from_dof_flags(lay, cu.flag, ch.flag), "vectors": (sol, old, oldold)}


def config5_standin(dev, local_rank, steps: int, levels: int = 8):
    """The config-5 stand-in in the driver's run: two meshes of the adaptive sequence (the first context of a process pays
    one-time module loads), per-call times of the second one: Jacobian + residual, residual only, context rebuild."""
    import torch

    from cracks_amd.assembler import Assembler

    rec = {}
    for step in range(2):
        pb = config5_problem(levels, step)
        mesh, lay = pb["mesh"], pb["layout"]
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        asm = Assembler(mesh, blocked=False, device=local_rank)
        asm.allocate_matrix()
        asm.set_params(pb["params"])
        asm.set_constraints(pb["node_flags"])
        torch.cuda.synchronize(dev)
        t_ctx = time.perf_counter() - t0
        asm.set_vectors(*pb["vectors"])
        if step == 1:
            w_j, k_j = time_mode(asm, dev, False, steps, 3)
            w_r, k_r = time_mode(asm, dev, True, 2 * steps, 3)
            rows, cells = asm.ctx.overlay_info()
            ab_j, ab_r = algorithmic_bytes_per_cell(2, False), algorithmic_bytes_per_cell(2, True)
            rec = {"workload": f"unit-slit square, {2 ** (levels + 1)}^2 base cells + refined band, stress split on, one-block layout",
                   "cells": int(mesh.n_cells), "dofs": int(lay.n_dofs), "hanging_nodes": int(mesh.hn_nodes.size),
                   "kernel_path": int(asm.ctx.kernel_path),
                   "overlay": {"rows_of_the_patch_kernel": int(rows), "cells_left_to_the_general_family": int(cells)},
                   "jacobian_ms": w_j, "jacobian_kernel_ms": k_j, "residual_only_ms": w_r, "residual_only_kernel_ms": k_r,
                   "jacobian_hbm_frac": ab_j * mesh.n_cells / (w_j * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "residual_only_hbm_frac": ab_r * mesh.n_cells / (w_r * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "algorithmic_bytes_per_cell": {"jacobian": ab_j, "residual_only": ab_r},
                   "context_rebuild_ms": 1e3 * t_ctx,
                   "context_rebuild_note": "pfm_ctx_create + pattern + pfm_set_params + pfm_set_constraints after refine_mesh, second mesh of the sequence",
                   "DoFs_per_s_jacobian": lay.n_dofs / (w_j * 1e-3)}
        asm.ctx.close()
    return rec


def host_pointer_call(asm, dev, n_dofs: int):
    """MEASURED wall time of the reference's call shape (outputs complete in host-visible storage on
    return, cracks.cc:2754, 2770, 2918): pfm_assemble -- state H2D, the kernels, every matrix value and the residual in
    the host's own arrays -- on the bench problem itself.  The arrays are page-locked once through pfm_host_register (what
    the glue does with Epetra's value arrays at setup_system); the (u,phi) block, identically zero, is cleared once on the
    host and never transferred.  One call from pageable arrays is timed first for comparison."""
    import psutil
    import torch

    ctx = asm.ctx
    sizes = [ctx.pattern_size(b)[1] for b in range(ctx.n_blocks)]
    need = 8.0 * (sum(sizes) + 5 * ctx.n_owned_dofs)
    avail = float(psutil.virtual_memory().available)
    if avail < 1.25 * need + (8 << 30):
        return {"skipped": f"host memory: {need / 1e9:.1f} GB of arrays needed, {avail / 1e9:.1f} GB available"}
    sol, old, oo = (t.cpu().numpy().copy() for t in (asm.solution, asm.old_solution, asm.old_old_solution))
    # the device copy of the matrix that torch holds is not needed here: the library stages in its own buffers
    asm.system_pde_matrix = None
    torch.cuda.empty_cache()
    values = [np.empty(k) for k in sizes]
    res = np.empty(ctx.n_owned_dofs)
    rec = {"bytes_host_arrays": need, "blocks_transferred": "(u,u), (phi,u), (phi,phi); (u,phi) = 0 cleared once on the host"}

    def call():
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ctx.assemble_host(sol, old, oo, False, out=(values, res, None))
        return time.perf_counter() - t0

    call()  # first touch of the pageable arrays, staging buffers of the library
    rec["seconds_pageable"] = call()
    t0 = time.perf_counter()
    pinned = True
    try:
        for a in values + [res, sol, old, oo]:
            ctx.host_register(a)
    except Exception as e:
        pinned = False
        rec["register_error"] = str(e)
    rec["register_seconds"] = time.perf_counter() - t0
    rec["pinned"] = pinned
    call()  # clears the (u,phi) block on the host (once per registration)
    ts = [call() for _ in range(3)]
    rec["seconds_per_assembly"] = float(np.median(ts))
    moved = 8.0 * (sum(k for b, k in enumerate(sizes) if not (ctx.n_blocks == 4 and b == 1)) + 4 * ctx.n_owned_dofs)
    rec["bytes_over_pcie"] = moved
    rec["GBps_effective"] = moved / rec["seconds_per_assembly"] / 1e9
    rec["DoFs_per_s"] = n_dofs / rec["seconds_per_assembly"]
    # the residual-only call in the same shape (cracks.cc:2946-2949: the line search reads system_pde_residual on the host
    # after every assemble_nl_residual(), 10-50 times per Newton step): three vectors in, two residual vectors out
    if pinned:
        res_tot = np.empty(ctx.n_owned_dofs)
        ctx.host_register(res_tot)

        def call_res():
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ctx.assemble_host(sol, old, oo, True, out=(None, res, res_tot))
            return time.perf_counter() - t0

        call_res()
        tr = float(np.median([call_res() for _ in range(5)]))
        moved_r = 8.0 * 5 * ctx.n_owned_dofs
        rec["residual_only"] = {"seconds_per_call": tr, "bytes_over_pcie": moved_r, "GBps_effective": moved_r / tr / 1e9,
                                "DoFs_per_s": n_dofs / tr,
                                "note": "pfm_assemble(residual_only=1), page-locked host vectors: 3 in, 2 out; a caller that only needs "
                                        "the norm uses extra.residual_only_3d.line_search_call (24 bytes)"}
    ctx.host_unregister()
    return rec


def self_launch_command(n_gpus: int, argv):
    """argv of `python -m torch.distributed.run ... bench.py <argv>` for a single-node run with one rank per GPU."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # torch.distributed.run's own parser trips over the abbreviation --n: pass the long form on
    fwd = ["--cells" if a == "--n" else ("--cells=" + a[4:] if a.startswith("--n=") else a) for a in argv]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + fwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--n", "--cells", dest="n", type=int, default=0,
                    help="cells per direction (default 216 in 3-D, 1000 in 2-D); --cells under torch.distributed.run, whose parser trips over --n")
    ap.add_argument("--residual-only", action="store_true")
    ap.add_argument("--path", choices=["auto", "general", "cart", "overlay"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra lines (residual-only, 2-D configurations, PCIe-inclusive value)")
    ap.add_argument("--checksum", action="store_true",
                    help="add partition-independent sums of the assembled values (outside the timed region): the "
                         "N-rank run must reproduce the 1-rank numbers up to round-off")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from cracks_amd import partition as P
    from cracks_amd.assembler import Assembler
    from cracks_amd.halo import HaloExchange

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as typed: become the launcher of N ranks of this very script (one process per GPU,
        # rendezvous on 127.0.0.1 at a free port); rank 0 of the children prints the JSON line
        os.execv(sys.executable, self_launch_command(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} does not match WORLD_SIZE={world} of the launcher")
    # PFM_BENCH_SMOKE_GLOO=1: all ranks on cuda:0, ghost import staged through the host over gloo -- exercises the
    # multi-process flow on a single-GPU box; never a measurement
    smoke_gloo = os.environ.get("PFM_BENCH_SMOKE_GLOO") == "1"
    if smoke_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if smoke_gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    dim = args.dim
    n = args.n or (216 if dim == 3 else 1000)
    ncell = (n,) * dim
    p = P.bench_grid(world, dim, n)
    t0 = time.perf_counter()
    lp = P.build_local_problem(dim, ncell, p, rank)
    h = (20.0 / n) * np.sqrt(dim)
    u, phi, po, poo, flags = synthetic_state(lp.mesh, lp.global_ids, h, dim)
    halo = None
    if world > 1:
        halo = HaloExchange(dim, lp.peers, lp.send_ptr, lp.send_nodes, lp.recv_ptr, lp.recv_nodes, dev)
    asm = Assembler(lp.mesh, blocked=True, device=local_rank, n_owned_nodes=lp.n_owned, halo=halo)
    if args.path == "general":
        asm.ctx.force_path(0)
    elif args.path == "cart":
        asm.ctx.force_path(1)
    elif args.path == "overlay":
        asm.ctx.force_path(2)
    asm.set_params(sneddon_params(h, dim))
    asm.set_constraints(flags)
    no = lp.n_owned
    # owned dof vectors, blocked layout [u | phi]
    def pack(uu, pp):
        v = np.empty(no * (dim + 1))
        v[:no * dim] = uu[:no].reshape(-1)
        v[no * dim:] = pp[:no]
        return v
    asm.set_vectors(pack(u, phi), pack(np.zeros_like(u), po), pack(np.zeros_like(u), poo))
    t_setup = time.perf_counter() - t0
    # a second context on the same mesh: pfm_ctx_create without the process's one-time costs (first large host->device
    # copy, pinned staging buffers) = what the rebuild after a refine_mesh costs in a running program
    ctx_rebuild_s = None
    if world == 1 and not smoke_gloo:
        again = Assembler(lp.mesh, blocked=True, device=local_rank, n_owned_nodes=lp.n_owned)
        ctx_rebuild_s = round(again.ctx.create_seconds, 3)
        again.ctx.close()
        del again
    residual_only = args.residual_only

    # residual-only runs measure the line search (cracks.cc:2942-2957): between two assemble_nl_residual() calls only
    # `solution` has changed, so only it is scattered again (pfm_state_set_solution); the first call scatters all three
    first_call = [True]

    def step():
        asm.assemble_system(residual_only, solution_only=residual_only and not first_call[0])
        first_call[0] = False

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    asm.synchronize()
    asm.ctx.timing_enable(True)
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t1
    asm.synchronize()
    k_all = asm.ctx.kernel_times_ms()
    k_med = float(np.median(k_all)) if k_all.size else 0.0
    k_ms, k_n = asm.ctx.kernel_time_ms()
    asm.ctx.timing_enable(False)

    # the ghost import alone (cracks.cc:2147-2154 as pack -> RCCL group -> unpack): events around halo.exchange, median of 20
    exchange_ms = 0.0
    if world > 1 and halo is not None:
        xs = []
        for _ in range(3):
            halo.exchange(asm.ctx)
        fence()
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            halo.exchange(asm.ctx)
            e1.record()
            e1.synchronize()
            xs.append(e0.elapsed_time(e1))
        exchange_ms = float(np.median(xs))
    # what RCCL itself counted on the communicator the exchange used (None: no in-library RCCL transport in this run)
    rccl = halo.rccl_info() if (world > 1 and halo is not None) else None
    tt = torch.tensor([elapsed, k_ms, k_med, exchange_ms], dtype=torch.float64, device="cpu" if smoke_gloo else dev)
    per_rank = None
    if world > 1:
        # every rank's own numbers (outside the timed region): compute imbalance shows as max - min of the kernel time,
        # exchange cost as elapsed - kernel time
        allr = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allr, tt)
        rows = np.array([[float(x) for x in r.cpu()] for r in allr])
        per_rank = {"kernel_ms": {"min": float(rows[:, 1].min()), "max": float(rows[:, 1].max()), "all": [round(float(x), 4) for x in rows[:, 1]]},
                    "ms_per_step": {"min": float(rows[:, 0].min() * 1e3 / args.steps), "max": float(rows[:, 0].max() * 1e3 / args.steps)},
                    "exchange_ms": {"min": float(rows[:, 3].min()), "max": float(rows[:, 3].max())},
                    "owned_nodes": None}
        nn = torch.tensor([float(lp.n_owned), float(lp.mesh.n_cells)], dtype=torch.float64, device="cpu" if smoke_gloo else dev)
        alln = [torch.zeros_like(nn) for _ in range(world)]
        dist.all_gather(alln, nn)
        per_rank["owned_nodes"] = [int(x[0]) for x in alln]
        per_rank["local_cells"] = [int(x[1]) for x in alln]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, k_ms, k_med, exchange_ms = float(tt[0]), float(tt[1]), float(tt[2]), float(tt[3])
    checksum = None
    if args.checksum:
        # sums over the owned rows: every global row is owned by exactly one rank
        parts = [asm.system_pde_residual.sum(), asm.system_pde_residual.abs().sum()]
        if residual_only:
            parts += [asm.system_total_residual.sum(), asm.system_total_residual.abs().sum()]
        else:
            for m in asm.system_pde_matrix:
                parts += [m.sum(), m.abs().sum()]
        cs = torch.stack(parts).to("cpu" if smoke_gloo else dev)
        if world > 1:
            dist.all_reduce(cs, op=dist.ReduceOp.SUM)
        checksum = [float(x) for x in cs.cpu()]
    n_nodes_global = int(np.prod([k + 1 for k in ncell]))
    n_cells_global = int(np.prod(ncell))
    n_dofs = n_nodes_global * (dim + 1)
    ms_per_step = 1e3 * elapsed / args.steps

    # The headline cut is z-slabs (2 peers, whole x-y tiles); a deal.II host hands over p4est's Morton sub-cubes (up to 7
    # peers, partial tiles).  The same steps on the near-cubic grid, in the same run, so that the SCALE record shows both.
    cubic = None
    p_cubic = P.factor_ranks(world, dim)
    if world > 1 and dim == 3 and tuple(p_cubic) != tuple(p) and not args.no_extras and not residual_only:
        asm.ctx.close()
        del asm
        if halo is not None:
            halo.close()
        torch.cuda.empty_cache()
        lp2 = P.build_local_problem(dim, ncell, p_cubic, rank)
        u2, phi2, po2, poo2, flags2 = synthetic_state(lp2.mesh, lp2.global_ids, h, dim)
        halo2 = HaloExchange(dim, lp2.peers, lp2.send_ptr, lp2.send_nodes, lp2.recv_ptr, lp2.recv_nodes, dev)
        asm2 = Assembler(lp2.mesh, blocked=True, device=local_rank, n_owned_nodes=lp2.n_owned, halo=halo2)
        asm2.set_params(sneddon_params(h, dim))
        asm2.set_constraints(flags2)
        no2 = lp2.n_owned

        def pack2(uu, pp):
            v = np.empty(no2 * (dim + 1))
            v[:no2 * dim] = uu[:no2].reshape(-1)
            v[no2 * dim:] = pp[:no2]
            return v
        asm2.set_vectors(pack2(u2, phi2), pack2(np.zeros_like(u2), po2), pack2(np.zeros_like(u2), poo2))
        for _ in range(args.warmup):
            asm2.assemble_system(False)
        asm2.synchronize()
        asm2.ctx.timing_enable(True)
        fence()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            asm2.assemble_system(False)
        fence()
        el2 = time.perf_counter() - t2
        asm2.synchronize()
        k2_ms, _ = asm2.ctx.kernel_time_ms()
        asm2.ctx.timing_enable(False)
        xs = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            halo2.exchange(asm2.ctx)
            e1.record()
            e1.synchronize()
            xs.append(e0.elapsed_time(e1))
        t3 = torch.tensor([el2, float(np.median(xs)), float(len(lp2.peers))], dtype=torch.float64, device="cpu" if smoke_gloo else dev)
        k2 = torch.tensor([k2_ms], dtype=torch.float64, device="cpu" if smoke_gloo else dev)
        k2all = [torch.zeros_like(k2) for _ in range(world)]
        dist.all_gather(k2all, k2)
        k2v = [float(x[0]) for x in k2all]
        dist.all_reduce(t3, op=dist.ReduceOp.MAX)
        cubic = {"partition": "x".join(str(k) for k in p_cubic), "ms_per_step": 1e3 * float(t3[0]) / args.steps,
                 "value": n_dofs / (float(t3[0]) / args.steps), "exchange_ms": float(t3[1]), "max_peers": int(t3[2]),
                 "per_rank_kernel_ms": {"min": min(k2v), "max": max(k2v), "all": [round(x, 4) for x in k2v]}}
        asm = asm2

    if rank == 0:
        abytes = algorithmic_bytes_per_cell(dim, residual_only) * lp.mesh.n_cells  # this rank's launch
        achieved = abytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic, traffic_src = measured_hbm_traffic(dim, n, residual_only) if world == 1 else (None, None)
        vi, vi_src = measured_valu_instructions(dim, n, residual_only) if world == 1 else (None, None)
        valu = None
        if vi is not None:
            to_ms = 4.0 / N_SIMD / (CLOCK_GHZ * 1e9) * 1e3
            valu = {"wave_instructions_per_launch": vi["valu"], "fp64_wave_instructions_per_launch": vi["fp64"],
                    "fp64_share": vi["fp64"] / vi["valu"] if vi["valu"] else None,
                    "issue_floor_ms_at_max_clock": vi["valu"] * to_ms, "fp64_floor_ms_at_max_clock": vi["fp64"] * to_ms,
                    "issue_floor_frac_of_kernel_time": vi["valu"] * to_ms / k_ms if k_ms > 0 else None, "source": vi_src}
        elif vi_src:
            valu = {"source": vi_src}
        out = {
            "metric": "assembled DoFs/sec (residual+Jacobian) on 3D Sneddon" if (dim == 3 and not residual_only)
            else f"assembled DoFs/sec ({'residual-only' if residual_only else 'residual+Jacobian'}) on {dim}D Sneddon",
            "value": n_dofs / (elapsed / args.steps),
            "unit": "DoFs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "parity": "GPU vs CPU oracle: |x - x_ref|_inf < 1e-12 * max(1, |x_ref|_inf), every matrix entry and residual (tests/, 216^3 sample in tests/test_gpu_fullsize.py)",
            "data": "synthetic (uniform hex mesh on [-10,10]^d, interpolated Sneddon crack + seeded perturbation)" +
                    (" -- SMOKE RUN: all ranks on one GPU, gloo, not a measurement" if smoke_gloo else ""),
            "config": {"workload": f"Sneddon {dim}D, {n}^{dim} = {n_cells_global} Q1 cells, {n_dofs} DoFs, "
                                   f"{'residual-only' if residual_only else 'full Jacobian+residual, 2x2 block CSR (%d nnz/row-node-comp)' % (4 * 3 ** dim)}",
                       "partition": "x".join(str(k) for k in p), "peers": len(lp.peers), "kernel_path": asm.ctx.kernel_path,
                       "state_scatter": ("solution only, read by the residual kernel" if world == 1 else "solution only") if residual_only else "all three vectors, every step",
                       "setup_s": round(t_setup, 2),  # mesh + synthetic state in numpy + context
                       # pfm_ctx_create alone: what a setup_system() after refine_mesh costs (cracks.cc:4148)
                       "ctx_create_s": round(asm.ctx.create_seconds, 3), "ctx_rebuild_s": ctx_rebuild_s},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes": abytes,
                         # second bound: VALU issue, from counted instructions (not a flop estimate)
                         "valu": valu,
                         "kernel_ms": k_ms, "kernel_ms_median": k_med, "launches": k_n,
                         "algorithmic_bytes_per_cell": algorithmic_bytes_per_cell(dim, residual_only)},
        }
        if world > 1:
            out["exchange_ms"] = exchange_ms  # the ghost import alone, max over ranks of the per-rank median
            # ncclCommCount / ncclGetVersion of the communicator the ghost exchange ran on (rank 0's view; null in gloo runs)
            out["rccl_nranks"] = rccl["rccl_nranks"] if rccl else None
            out["rccl_version"] = rccl["rccl_version"] if rccl else None
            out["per_rank"] = per_rank
            out["per_rank_kernel_ms"] = per_rank["kernel_ms"] if per_rank else None
        if cubic is not None:
            out["cubic_partition"] = cubic
        if checksum is not None:
            out["checksum"] = checksum
        if world == 1 and not smoke_gloo and not args.no_extras and dim == 3 and not residual_only and args.path == "auto":
            try:
                out["extra"] = extra_lines(asm, dev, local_rank, n, max(5, args.steps // 2))
                # PCIe-inclusive figure of SURVEY 8(d), MEASURED: pfm_assemble into the host's own (page-locked) arrays
                hp = host_pointer_call(asm, dev, n_dofs)
                out["host_pointer_call"] = hp
                out["value_incl_d2h"] = hp.get("DoFs_per_s")  # never the headline value
            except Exception as e:  # the headline line must not depend on the extras
                out["extra_error"] = f"{type(e).__name__}: {e}"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dim, residual_only)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
