#!/usr/bin/env python3
"""Measurements of SURVEY.md §8(d) that are not the headline bench line (run on the GPU box):

  config5  Miehe-shear-like adaptive sequence (cracks.cc:4137-4174 rebuilds everything after refine_mesh):
           unit-slit squares, one-block (direct solver) layout, stress split active
           (decompose_stress_rhs = decompose_stress_matrix = 1, timestep_number > 0), a refined band with hanging
           nodes that grows with the crack.  Per mesh: context rebuild time (pfm_ctx_create + pattern +
           pfm_set_constraints) and assembly times.  Single GPU: the sub-box partition of cracks_amd/partition.py
           covers uniform boxes only.
  pcie     what the host round trip of today's Trilinos solve would add to config 3: device->host copy of the
           CSR values and residuals over PCIe (never part of bench.py's value).

  python tools/bench_extra.py config5 [--levels 7] [--world 4] [--out profiles/rNN/config5_miehe_amr.json]
  python -m torch.distributed.run --nproc-per-node 4 tools/bench_extra.py config5 --dist     (4 GPUs, RCCL ghost import)
  python tools/bench_extra.py pcie    [--n 216]    [--out profiles/rNN/pcie_216cube.json]
  python tools/bench_extra.py rank    [--n 216 --world 8]   one rank's share of the strong-scaling run, no exchange
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def config5(args):
    import torch

    from cracks_amd import mesh as M
    from cracks_amd.assembler import Assembler, node_flags_from_dof_flags
    from cracks_amd.capi import PfmParams

    import bench as B

    rows = []
    for step in range(args.meshes):
        pb = B.config5_problem(args.levels, step)
        mesh, lay, prm, cu, ch = pb["mesh"], pb["layout"], pb["params"], pb["cu"], pb["ch"]
        t_refine = pb["refine_host_s"]
        sol, old, oldold = pb["vectors"]

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        asm = Assembler(mesh, blocked=False)
        asm.allocate_matrix()
        t_ctx = time.perf_counter() - t0
        t0 = time.perf_counter()
        asm.set_params(prm)
        asm.set_constraints(node_flags_from_dof_flags(lay, cu.flag, ch.flag))
        t_con = time.perf_counter() - t0
        asm.set_vectors(sol, old, oldold)

        def timed(residual_only, reps=20):
            for _ in range(3):
                asm.assemble_system(residual_only)
            asm.synchronize()
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t = time.perf_counter()
                asm.assemble_system(residual_only)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t)
            asm.synchronize()
            return 1e3 * float(np.median(ts))

        t_jac, t_res = timed(False), timed(True)
        rank_rows = []
        if args.world > 1:
            # the same mesh on args.world ranks (general partition, cracks_amd/partition.py): every rank's share is
            # timed on this GPU, one after the other, without the ghost import
            from cracks_amd import partition as P

            gflags = node_flags_from_dof_flags(lay, cu.flag, ch.flag)
            fields_g = np.stack([sol[0::3], sol[1::3], sol[2::3], old[2::3], oldold[2::3]], axis=1)  # one-block layout
            t0 = time.perf_counter()
            lps = P.partition_general(mesh, args.world)
            t_part = time.perf_counter() - t0
            for r, lp in enumerate(lps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                a = Assembler(lp.mesh, blocked=False, n_owned_nodes=lp.n_owned)
                a.allocate_matrix()
                a.set_params(prm)
                a.set_constraints(gflags[lp.global_ids])
                torch.cuda.synchronize()
                t_rctx = time.perf_counter() - t0
                gd = (lp.global_ids[:lp.n_owned, None] * 3 + np.arange(3)[None, :]).ravel()  # one-block layout
                a.set_vectors(sol[gd], old[gd], oldold[gd])
                # ghost values: what the import would deliver (packed per peer, field-major), through the HIP unpack
                a.ctx.state_set_device(a.solution.data_ptr(), a.old_solution.data_ptr(), a.old_old_solution.data_ptr())
                a.ctx.halo_register(lp.send_ptr, lp.send_nodes, lp.recv_ptr, lp.recv_nodes)
                loc = fields_g[lp.global_ids]
                msgs = [loc[lp.recv_nodes[lp.recv_ptr[k]:lp.recv_ptr[k + 1]]].T.ravel() for k in range(len(lp.peers))]
                if msgs:
                    recv_all = torch.from_numpy(np.concatenate(msgs)).cuda()
                    a.ctx.halo_unpack_all(recv_all.data_ptr())
                ts = {}
                for ro in (False, True):
                    for _ in range(3):
                        a.ctx.assemble_device(ro, [m.data_ptr() for m in a.system_pde_matrix] if not ro else [],
                                              a.system_pde_residual.data_ptr(), a.system_total_residual.data_ptr())
                    a.synchronize()
                    tt = []
                    for _ in range(20):
                        torch.cuda.synchronize()
                        t = time.perf_counter()
                        a.ctx.assemble_device(ro, [m.data_ptr() for m in a.system_pde_matrix] if not ro else [],
                                              a.system_pde_residual.data_ptr(), a.system_total_residual.data_ptr())
                        torch.cuda.synchronize()
                        tt.append(time.perf_counter() - t)
                    ts[ro] = 1e3 * float(np.median(tt))
                rank_rows.append({"rank": r, "local_cells": int(lp.mesh.n_cells), "owned_cells": int(lp.cell_owned.sum()),
                                  "owned_nodes": int(lp.n_owned), "ghost_nodes": int(lp.mesh.n_nodes - lp.n_owned),
                                  "peers": lp.peers, "halo_bytes_sent": int(lp.send_ptr[-1]) * 5 * 8,
                                  "context_rebuild_s": round(t_rctx, 4), "assemble_jacobian_ms": round(ts[False], 4),
                                  "assemble_residual_ms": round(ts[True], 4)})
                print(rank_rows[-1], flush=True)
                del a
            rank_rows = {"world": args.world, "partition_host_s": round(t_part, 3), "ranks": rank_rows,
                         "max_assemble_jacobian_ms": max(x["assemble_jacobian_ms"] for x in rank_rows),
                         "max_context_rebuild_s": max(x["context_rebuild_s"] for x in rank_rows)}
        rows.append({"mesh": step, "cells": int(mesh.n_cells), "nodes": int(mesh.n_nodes), "dofs": int(lay.n_dofs),
                     **({"multi_rank": rank_rows} if rank_rows else {}),
                     "hanging_nodes": int(mesh.hn_nodes.size), "refine_host_s": round(t_refine, 3),
                     "context_rebuild_s": round(t_ctx, 4), "set_params_constraints_s": round(t_con, 4),
                     "assemble_jacobian_ms": round(t_jac, 4), "assemble_residual_ms": round(t_res, 4),
                     "dofs_per_s_jacobian": lay.n_dofs / (t_jac * 1e-3), "kernel_path": asm.ctx.kernel_path})
        print(rows[-1], flush=True)
        del asm
    out = {"config": "SURVEY §8(d) config 5 stand-in: unit slit, %d^2 base cells, refined band, one-block layout, "
                     "stress split active, single MI355X" % (2 ** (args.levels + 1)), "rows": rows}
    if args.out:
        json.dump(out, open(os.path.join(ROOT, args.out), "w"), indent=1)


def config5_dist(args):
    """BASELINE config 'Miehe shear with predictor-corrector AMR, 4 x MI355X': the config5 mesh sequence on
    WORLD_SIZE ranks (python -m torch.distributed.run --nproc-per-node 4 tools/bench_extra.py config5 --dist).
    Per mesh: general partition (cracks_amd/partition.py), context rebuild, then timed reassemblies INCLUDING the
    ghost import over RCCL; times are the max over ranks."""
    import torch
    import torch.distributed as dist

    from cracks_amd import mesh as M
    from cracks_amd import partition as P
    from cracks_amd.assembler import Assembler, node_flags_from_dof_flags
    from cracks_amd.capi import PfmParams
    from cracks_amd.halo import HaloExchange

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    rows = []
    base = M.slit_mesh(args.levels)
    n = 2 ** (args.levels + 1)
    h = 1.0 / n
    for step in range(args.meshes):
        tip = 0.5 - 0.08 * step
        cc = base.coords[base.cells].mean(axis=1)
        mesh = M.refine_cells(base, (np.abs(cc[:, 1] - 0.5) < 6 * h) & (cc[:, 0] > tip - 4 * h))
        lay = M.DofLayout(mesh.n_nodes, 2, blocked=False)
        hfine, dt = 0.5 * h, 1.0e-4
        prm = PfmParams(lambda_=121.15e3, mu=80.77e3, G_c=2.7, alpha_eps=2.0 * hfine * np.sqrt(2.0),
                        constant_k=1.0e-10 * hfine, pressure=0.0, alpha_biot=0.0, gamma_penal=0.0, timestep=dt,
                        time=5 * dt, old_timestep=dt, old_old_timestep=dt, decompose_stress_rhs=1.0,
                        decompose_stress_matrix=1.0, timestep_number=5, outer_solver=0, use_old_timestep_pf=0,
                        reserved=0)
        ch = M.hanging_constraints(mesh, lay)
        cu = M.update_constraints(mesh, lay, M.miehe_shear_dirichlet_dofs(mesh, lay))
        rng = np.random.default_rng(1234 + step)
        x, y = mesh.coords[:, 0], mesh.coords[:, 1]
        u = np.stack([-5 * dt * y + 1e-6 * rng.standard_normal(x.size), 1e-6 * rng.standard_normal(x.size)], axis=1)
        phi = np.clip(1.0 - np.exp(-np.abs(y - 0.5) / (4 * hfine)) * (x > tip), 0.0, 1.0)
        sol = ch.distribute(lay.pack(u, phi))
        old = ch.distribute(lay.pack(0.9 * u, np.clip(phi + 0.01 * rng.random(x.size), 0, 1)))
        oldold = ch.distribute(lay.pack(0.8 * u, np.clip(phi + 0.02 * rng.random(x.size), 0, 1)))
        gflags = node_flags_from_dof_flags(lay, cu.flag, ch.flag)

        t0 = time.perf_counter()
        lp = P.partition_general(mesh, world)[rank]
        t_part = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        halo = HaloExchange(2, lp.peers, lp.send_ptr, lp.send_nodes, lp.recv_ptr, lp.recv_nodes, dev) if world > 1 else None
        asm = Assembler(lp.mesh, blocked=False, device=local_rank, n_owned_nodes=lp.n_owned, halo=halo)
        asm.allocate_matrix()
        asm.set_params(prm)
        asm.set_constraints(gflags[lp.global_ids])
        torch.cuda.synchronize(dev)
        t_ctx = time.perf_counter() - t0
        gd = (lp.global_ids[:lp.n_owned, None] * 3 + np.arange(3)[None, :]).ravel()
        asm.set_vectors(sol[gd], old[gd], oldold[gd])

        def fence():
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)

        out = {}
        for ro in (False, True):
            for _ in range(3):
                asm.assemble_system(ro)
            asm.synchronize()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                asm.assemble_system(ro)
            fence()
            out[ro] = (time.perf_counter() - t0) / args.steps
        asm.synchronize()
        tt = torch.tensor([out[False], out[True], t_ctx, t_part], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if rank == 0:
            rows.append({"mesh": step, "cells": int(mesh.n_cells), "dofs": int(lay.n_dofs), "hanging_nodes": int(mesh.hn_nodes.size),
                         "n_gpus": world, "reassemble_jacobian_ms": 1e3 * float(tt[0]), "reassemble_residual_ms": 1e3 * float(tt[1]),
                         "context_rebuild_s": float(tt[2]), "partition_host_s": float(tt[3]),
                         "dofs_per_s_jacobian": lay.n_dofs / float(tt[0])})
            print(json.dumps(rows[-1]), flush=True)
        del asm
    if rank == 0 and args.out:
        json.dump({"config": "Miehe-shear-like AMR sequence on %d GPU(s), ghost import over RCCL included" % world, "rows": rows},
                  open(os.path.join(ROOT, args.out), "w"), indent=1)
    if world > 1:
        dist.destroy_process_group()


def pcie(args):
    import torch

    n = args.n
    nodes = (n + 1) ** 3
    nbytes_values = nodes * 27 * 16 * 8
    nbytes_res = nodes * 4 * 8 * 2
    dev = torch.device("cuda", 0)
    chunk = 1 << 30  # copy 1 GiB pieces through one pinned buffer: what a host solver hand-over would do
    src = torch.empty(chunk // 8, dtype=torch.float64, device=dev).normal_()
    dst = torch.empty(chunk // 8, dtype=torch.float64).pin_memory()
    for _ in range(2):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        t = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    gbs = chunk / float(np.median(ts)) / 1e9
    out = {"workload": f"Sneddon 3D {n}^3: {nbytes_values / 1e9:.2f} GB of CSR values + {nbytes_res / 1e9:.3f} GB of residuals",
           "d2h_pinned_GBps": gbs, "d2h_seconds_for_one_jacobian": (nbytes_values + nbytes_res) / (gbs * 1e9)}
    print(out, flush=True)
    if args.out:
        json.dump(out, open(os.path.join(ROOT, args.out), "w"), indent=1)


def rank_share(args):
    """Assembly time of ONE rank's sub-box of an N-rank run (no exchange): what a GPU of the strong-scaling run
    computes per step.  Lets the per-rank efficiency be measured on a single-GPU box."""
    import torch

    import bench
    from cracks_amd import partition as P
    from cracks_amd.assembler import Assembler

    n, world = args.n, args.world
    p = tuple(int(x) for x in args.grid.split(",")) if args.grid else P.bench_grid(world, 3, n)
    assert int(np.prod(p)) == world
    rows = []
    for rank in ([0, world - 1] if world > 1 else [0]):
        lp = P.build_local_problem(3, (n,) * 3, p, rank)
        h = (20.0 / n) * np.sqrt(3.0)
        u, phi, po, poo, flags = bench.synthetic_state(lp.mesh, lp.global_ids, h, 3)
        asm = Assembler(lp.mesh, blocked=True, n_owned_nodes=lp.n_owned)
        asm.set_params(bench.sneddon_params(h, 3))
        asm.set_constraints(flags)
        no = lp.n_owned

        def pack(uu, pp):
            v = np.empty(no * 4)
            v[:no * 3] = uu[:no].reshape(-1)
            v[no * 3:] = pp[:no]
            return v
        asm.set_vectors(pack(u, phi), pack(np.zeros_like(u), po), pack(np.zeros_like(u), poo))
        for _ in range(3):
            asm.assemble_system(False)
        asm.synchronize()
        ts = []
        for _ in range(20):
            torch.cuda.synchronize()
            t = time.perf_counter()
            asm.assemble_system(False)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        # the two halves of pfm_assemble_overlapped alone (pfm_ctx_force_phase): phase 1 = the tiles that read no ghost
        # node = the work that runs NEXT TO the ghost import; phase 2 = the rest.  And what the import itself costs on
        # this GPU without the wire: the pack + unpack launches of the registered lists.
        phase_ms = {}
        for ph in (1, 2):
            asm.ctx.force_phase(ph)
            for _ in range(2):
                asm.assemble_system(False)
            tp = []
            for _ in range(10):
                torch.cuda.synchronize()
                t = time.perf_counter()
                asm.assemble_system(False)
                torch.cuda.synchronize()
                tp.append(time.perf_counter() - t)
            phase_ms[ph] = 1e3 * float(np.median(tp))
        asm.ctx.force_phase(0)
        pack_ms = None
        if world > 1 and lp.send_nodes.size:
            asm.ctx.halo_register(lp.send_ptr, lp.send_nodes, lp.recv_ptr, lp.recv_nodes)
            rec = 6
            sbuf = torch.empty(rec * lp.send_nodes.size, dtype=torch.float64, device="cuda")
            rbuf = torch.zeros(rec * lp.recv_nodes.size, dtype=torch.float64, device="cuda")
            tp = []
            for _ in range(12):
                torch.cuda.synchronize()
                t = time.perf_counter()
                asm.ctx.halo_pack_all(sbuf.data_ptr())
                asm.ctx.halo_unpack_all(rbuf.data_ptr())
                torch.cuda.synchronize()
                tp.append(time.perf_counter() - t)
            pack_ms = 1e3 * float(np.median(tp[2:]))
        msg_bytes = 8 * 6 * int(lp.send_nodes.size)
        rows.append({"rank": rank, "of": world, "partition": "x".join(map(str, p)), "local_cells": int(lp.mesh.n_cells),
                     "owned_nodes": int(no), "assemble_ms": round(1e3 * float(np.median(ts)), 4),
                     "interior_phase_ms": round(phase_ms[1], 4), "boundary_phase_ms": round(phase_ms[2], 4),
                     "halo_pack_unpack_ms": None if pack_ms is None else round(pack_ms, 4),
                     "halo_send_bytes": msg_bytes, "n_peers": int(len(lp.peers)),
                     # exchange = pack + RCCL group (7 messages of <= 0.6 MB on 7 xGMI links: ~5 us of wire + ~25 us of
                     # group latency) + unpack; hidden if it is shorter than the interior phase, which runs beside it
                     "exchange_estimate_ms": None if pack_ms is None else round(pack_ms + 0.03, 4),
                     "exposed_exchange_ms_sequential": None if pack_ms is None else round(pack_ms + 0.03, 4),
                     "exposed_exchange_ms_overlapped": None if pack_ms is None else round(max(0.0, pack_ms + 0.03 - phase_ms[1]), 4)})
        print(rows[-1], flush=True)
        del asm
    if args.out:
        json.dump({"config": f"one rank of a {world}-rank strong-scaling run of Sneddon 3D {n}^3, no exchange", "rows": rows},
                  open(os.path.join(ROOT, args.out), "w"), indent=1)


def bound_pattern(args):
    """What a host with its own column order pays (deal.II numbers its dofs hierarchically; Epetra's local CSR is sorted by ITS
    local column ids): the uniform box of `--n`^3 cells assembled into the library's canonical pattern, then into the same
    pattern with the neighbour nodes of every row shuffled (pfm_pattern_bind).  Rows are complete but their slots are a
    permutation of the lattice order: k_cart_phi4 takes its blocked copy-out with looked-up destinations (round 6),
    k_cart_uu3 its generic one.  The sums of the values of every block must agree (a permutation within rows)."""
    import torch

    import bench
    from cracks_amd import partition as P
    from cracks_amd.assembler import Assembler

    n = args.n
    lp = P.build_local_problem(3, (n,) * 3, (1, 1, 1), 0)
    h = (20.0 / n) * np.sqrt(3.0)
    u, phi, po, poo, flags = bench.synthetic_state(lp.mesh, lp.global_ids, h, 3)
    asm = Assembler(lp.mesh, blocked=True, n_owned_nodes=lp.n_owned)
    asm.set_params(bench.sneddon_params(h, 3))
    asm.set_constraints(flags)
    no = lp.n_owned
    pack = lambda uu, pp: np.concatenate([uu[:no].reshape(-1), pp[:no]])
    asm.set_vectors(pack(u, phi), pack(np.zeros_like(u), po), pack(np.zeros_like(u), poo))
    dev = asm.dev

    def measure():
        wall, k_ms = bench.time_mode(asm, dev, False, args.steps, 3)
        sums = [float(m.sum()) for m in asm.system_pde_matrix] + [float(m.abs().sum()) for m in asm.system_pde_matrix]
        return wall, k_ms, sums

    w0, k0, s0 = measure()
    rng = np.random.default_rng(5)
    ctx = asm.ctx
    order = None
    t0 = time.perf_counter()
    canonical = [ctx.pattern(b) for b in range(4)]  # (all four before the first bind: a bound order shows in every block)
    bound = []
    for b in range(4):
        ncr, ncc = (3 if b in (0, 1) else 1), (3 if b in (0, 2) else 1)
        rp, ci = canonical[b]
        deg = (np.diff(rp)[::ncr] // ncc).astype(np.int64)
        node_ptr = np.concatenate([[0], np.cumsum(deg)])
        if order is None:  # one shuffle of the neighbour nodes per row node, the same in every block and row component
            node_of_slot = np.repeat(np.arange(no, dtype=np.int64), deg)
            order = np.lexsort((rng.random(node_ptr[-1]), node_of_slot)) - node_ptr[node_of_slot]  # source slot of new slot j
        new_ci = np.empty_like(ci)
        for c in range(ncr):
            start = rp[np.arange(no) * ncr + c].astype(np.int64)  # first entry of row (node, c)
            src = (np.repeat(start, deg) + order * ncc)  # first entry of the source slot, per (node, new slot)
            dst = (np.repeat(start, deg) + (np.arange(node_ptr[-1]) - node_ptr[np.repeat(np.arange(no), deg)]) * ncc)
            for k in range(ncc):
                new_ci[dst + k] = ci[src + k]
        bound.append((rp, new_ci))
    for b, (rp, new_ci) in enumerate(bound):
        ctx.pattern_bind(b, rp, new_ci)
    t_bind = time.perf_counter() - t0
    w1, k1, s1 = measure()
    ok = all(abs(a - b) <= 1e-9 * max(1.0, abs(a), abs(b)) for a, b in zip(s0, s1))
    rec = {"workload": f"Sneddon 3D, {n}^3 cells, Jacobian + residual, canonical pattern against a bound pattern with shuffled rows",
           "canonical": {"ms_per_call": w0, "kernel_ms": k0}, "bound_shuffled": {"ms_per_call": w1, "kernel_ms": k1},
           "ratio": k1 / k0, "block_sums_agree": ok, "shuffle_and_bind_s_python": round(t_bind, 1)}
    print(json.dumps(rec))
    if args.out:
        json.dump(rec, open(args.out, "w"), indent=1)
    assert ok, (s0, s1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["config5", "pcie", "rank", "bound"])
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--levels", type=int, default=7)
    ap.add_argument("--meshes", type=int, default=5)
    ap.add_argument("--n", type=int, default=216)
    ap.add_argument("--out", default=None)
    ap.add_argument("--dist", action="store_true", help="config5: one process per GPU under torch.distributed.run")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--grid", default=None, help="rank: process grid instead of the near-cubic one, e.g. 1,1,8")
    a = ap.parse_args()
    if a.what == "config5" and (a.dist or int(os.environ.get("WORLD_SIZE", "1")) > 1):
        config5_dist(a)
    else:
        {"config5": config5, "pcie": pcie, "rank": rank_share, "bound": bound_pattern}[a.what](a)
