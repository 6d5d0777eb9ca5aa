"""Single-GPU emulation of the multi-rank path: every rank's context lives on cuda:0, the
RCCL exchange is replaced by handing the packed device buffers from the owner's context to
the receiver's.  Checks the HIP pack/unpack kernels, the owned-first local numbering, the
cartesian family on sub-boxes with ghost layers and the owner-computes rule: the rows a rank
owns must equal the rows of the single-rank assembly."""
import numpy as np
import pytest

import oracle_api as O
from cracks_amd import mesh as M
from cracks_amd import partition as P
from gpu_util import linf_scaled

pytestmark = pytest.mark.gpu


def _global_problem(dim, n):
    import bench
    g = M.box_mesh(dim, n)
    h = g.min_cell_diameter()
    u, phi, po, poo, flags = bench.synthetic_state(g, np.arange(g.n_nodes), h, dim)
    return g, h, u, phi, po, poo, flags


@pytest.mark.parametrize("dim,n,world,path", [(3, (12, 11, 10), 2, 1), (3, (12, 11, 10), 8, 1), (3, (9, 9, 9), 4, 0),
                                              (3, (17, 9, 70), 8, 1),  # ranks with several z-chunks and partial tiles
                                              (2, (20, 14), 4, 1)])
def test_owned_rows_match_single_rank(dim, n, world, path):
    import torch
    import bench
    from cracks_amd.assembler import Assembler

    g, h, u, phi, po, poo, flags = _global_problem(dim, n)
    prm = bench.sneddon_params(h, dim)
    # single-rank reference on the GPU (general family) and its residual
    ref = Assembler(g, blocked=True)
    ref.ctx.force_path(0)
    ref.set_params(prm)
    ref.set_constraints(flags)
    N = g.n_nodes

    def pack(no, uu, pp):
        v = np.empty(no * (dim + 1))
        v[:no * dim] = uu[:no].reshape(-1)
        v[no * dim:] = pp[:no]
        return v

    ref.set_vectors(pack(N, u, phi), pack(N, 0 * u, po), pack(N, 0 * u, poo))
    ref.assemble_system()
    ref.synchronize()
    ref_vals = [m.cpu().numpy() for m in ref.system_pde_matrix]
    ref_res = ref.system_pde_residual.cpu().numpy()
    ref_pat = [ref.ctx.pattern(b) for b in range(4)]

    p = P.factor_ranks(world, dim)
    lps = [P.build_local_problem(dim, n, p, r) for r in range(world)]
    asms = []
    for lp in lps:
        a = Assembler(lp.mesh, blocked=True, n_owned_nodes=lp.n_owned)
        assert a.ctx.kernel_path == 1, "sub-boxes must stay on the cartesian family"
        if path == 0:
            a.ctx.force_path(0)
        a.set_params(prm)
        a.set_constraints(flags[lp.global_ids])
        gi = lp.global_ids
        no = lp.n_owned
        a.set_vectors(pack(no, u[gi], phi[gi]), pack(no, 0 * u[gi], po[gi]), pack(no, 0 * u[gi], poo[gi]))
        a.ctx.state_set_device(a.solution.data_ptr(), a.old_solution.data_ptr(), a.old_old_solution.data_ptr())
        a.ctx.halo_register(lp.send_ptr, lp.send_nodes, lp.recv_ptr, lp.recv_nodes)
        asms.append(a)
    # "exchange": pack on the sender, unpack on the receiver (same device).  Senders pack all peers with one launch
    # (pfm_halo_pack_all), which must equal the per-peer packs; receivers unpack everything with one launch.
    rec = dim + 3
    recv_all = [torch.zeros(int(lp.recv_ptr[-1] - lp.recv_ptr[0]) * rec, dtype=torch.float64, device="cuda") for lp in lps]
    for r, lp in enumerate(lps):
        send_all = torch.empty(int(lp.send_ptr[-1] - lp.send_ptr[0]) * rec, dtype=torch.float64, device="cuda")
        if send_all.numel():
            asms[r].ctx.halo_pack_all(send_all.data_ptr())
        for k, s in enumerate(lp.peers):
            nsend = int(lp.send_ptr[k + 1] - lp.send_ptr[k])
            if nsend == 0:
                continue
            buf = torch.empty(nsend * rec, dtype=torch.float64, device="cuda")
            asms[r].ctx.halo_pack(k, buf.data_ptr())
            torch.cuda.synchronize()
            o = int(lp.send_ptr[k] - lp.send_ptr[0]) * rec
            assert torch.equal(buf, send_all[o:o + nsend * rec])
            ko = lps[s].peers.index(r)
            assert int(lps[s].recv_ptr[ko + 1] - lps[s].recv_ptr[ko]) == nsend
            oo = int(lps[s].recv_ptr[ko] - lps[s].recv_ptr[0]) * rec
            recv_all[s][oo:oo + nsend * rec] = buf
    for s, lp in enumerate(lps):
        if recv_all[s].numel():
            asms[s].ctx.halo_unpack_all(recv_all[s].data_ptr())
        torch.cuda.synchronize()
    for r, (lp, a) in enumerate(zip(lps, asms)):
        for residual_only in (False, True):
            a.ctx.assemble_device(residual_only, [m.data_ptr() for m in a.system_pde_matrix] if not residual_only
                                  else [], a.system_pde_residual.data_ptr(), a.system_total_residual.data_ptr()) \
                if a.system_pde_matrix or residual_only else None
            if not residual_only:
                a.allocate_matrix()
                a.ctx.assemble_device(False, [m.data_ptr() for m in a.system_pde_matrix],
                                      a.system_pde_residual.data_ptr(), a.system_total_residual.data_ptr())
            a.synchronize()
            gi = lp.global_ids
            no = lp.n_owned
            res = a.system_pde_residual.cpu().numpy()
            gd_u = (gi[:no, None] * dim + np.arange(dim)[None, :]).ravel()
            assert linf_scaled(res[:no * dim], ref_res[gd_u]) < 1e-12
            assert linf_scaled(res[no * dim:], ref_res[N * dim + gi[:no]]) < 1e-12
            if residual_only:
                continue
            # matrix rows: map local columns to global and compare value by value
            for b in range(4):
                rp, ci = a.ctx.pattern(b)
                vals = a.system_pde_matrix[b].cpu().numpy()
                grp, gci = ref_pat[b]
                ncr = dim if b in (0, 1) else 1
                ncc = dim if b in (0, 2) else 1
                rows = no * ncr
                for lr in np.linspace(0, rows - 1, min(rows, 400)).astype(int):
                    node, c = divmod(lr, ncr)
                    grow = gi[node] * ncr + c
                    lcols = ci[rp[lr]:rp[lr + 1]]
                    gcols = gi[lcols // ncc] * ncc + lcols % ncc
                    want = dict(zip(gci[grp[grow]:grp[grow + 1]], ref_vals[b][grp[grow]:grp[grow + 1]]))
                    got = dict(zip(gcols, vals[rp[lr]:rp[lr + 1]]))
                    assert set(got) == set(want)
                    scale = max(1.0, max(abs(x) for x in want.values()))
                    assert max(abs(got[k] - want[k]) for k in want) < 1e-12 * scale


@pytest.mark.parametrize("kind,world,ghost_layer", [("slit2d", 4, "closure"), ("slit2d", 3, "closure"), ("box3d", 2, "closure"),
                                                   ("slit2d", 4, "dealii+shipped"), ("slit2d", 2, "dealii+shipped"),
                                                   ("box3d", 3, "dealii+shipped"), ("box3d", 4, "dealii+shipped")])
def test_general_partition_owned_rows_match_single_rank(kind, world, ghost_layer):
    """BASELINE config 'Miehe shear with AMR on 4 GPUs': general partition (hanging nodes, slit), general kernel
    family, stress split active in 2-D; every rank's context on cuda:0, ghost import through the HIP pack/unpack."""
    import torch
    from cracks_amd.assembler import Assembler, node_flags_from_dof_flags
    from cracks_amd.capi import PfmParams
    from test_partition_halo import amr_mesh, _amr_fields
    import cases

    g = amr_mesh(kind)
    dim = g.dim
    N = g.n_nodes
    f = _amr_fields(g, dim)
    for k, n in enumerate(g.hn_nodes):  # nodal fields with the hanging-node constraints distributed
        sl = slice(g.hn_ptr[k], g.hn_ptr[k + 1])
        f[n] = (g.hn_weights[sl, None] * f[g.hn_parents[sl]]).sum(axis=0)
    base = cases.kat_sneddon_3d(4) if dim == 3 else cases.kat_miehe_shear_1()
    prm = PfmParams.from_buffer_copy(bytes(base.params))
    if dim == 2:
        prm.decompose_stress_rhs = prm.decompose_stress_matrix = 1.0
        prm.timestep_number = 3
    glay = M.DofLayout(N, dim, blocked=True)
    dirichlet = M.sneddon_dirichlet_dofs if dim == 3 else M.miehe_shear_dirichlet_dofs
    gcu = M.update_constraints(g, glay, dirichlet(g, glay))
    gch = M.hanging_constraints(g, glay)
    gflags = node_flags_from_dof_flags(glay, gcu.flag, gch.flag)

    def pack(no, ff):
        v = np.empty(no * (dim + 1))
        v[:no * dim] = ff[:no, :dim].reshape(-1)
        return v

    def vectors(no, ff):
        sol, old, oo = pack(no, ff), pack(no, 0 * ff), pack(no, 0 * ff)
        sol[no * dim:], old[no * dim:], oo[no * dim:] = ff[:no, dim], ff[:no, dim + 1], ff[:no, dim + 2]
        return sol, old, oo

    ref = Assembler(g, blocked=True)
    assert ref.ctx.kernel_path == (3 if dim == 2 else 0)  # 2-D: general family + cartesian overlay
    ref.set_params(prm)
    ref.set_constraints(gflags)
    ref.set_vectors(*vectors(N, f))
    ref.assemble_system()
    ref.synchronize()
    ref_vals = [m.cpu().numpy() for m in ref.system_pde_matrix]
    ref_res = ref.system_pde_residual.cpu().numpy()
    ref_pat = [ref.ctx.pattern(b) for b in range(4)]

    # "dealii+shipped": the cell set a deal.II host hands over (its one-cell ghost layer + the cells that reach an owned
    # row through a hanging vertex, shipped by their owners: partition.hanging_closure_shipments, the glue)
    lps = P.partition_general(g, world, ghost_layer=ghost_layer)
    asms = []
    for lp in lps:
        a = Assembler(lp.mesh, blocked=True, n_owned_nodes=lp.n_owned)
        a.set_params(prm)
        a.set_constraints(gflags[lp.global_ids])
        a.set_vectors(*vectors(lp.n_owned, f[lp.global_ids]))
        a.ctx.state_set_device(a.solution.data_ptr(), a.old_solution.data_ptr(), a.old_old_solution.data_ptr())
        a.ctx.halo_register(lp.send_ptr, lp.send_nodes, lp.recv_ptr, lp.recv_nodes)
        asms.append(a)
    rec = dim + 3
    recv_all = [torch.zeros(int(lp.recv_ptr[-1]) * rec, dtype=torch.float64, device="cuda") for lp in lps]
    for r, lp in enumerate(lps):
        send_all = torch.empty(int(lp.send_ptr[-1]) * rec, dtype=torch.float64, device="cuda")
        if send_all.numel():
            asms[r].ctx.halo_pack_all(send_all.data_ptr())
        torch.cuda.synchronize()
        for k, s in enumerate(lp.peers):
            ko = lps[s].peers.index(r)
            o0, o1 = int(lp.send_ptr[k]) * rec, int(lp.send_ptr[k + 1]) * rec
            q0 = int(lps[s].recv_ptr[ko]) * rec
            recv_all[s][q0:q0 + (o1 - o0)] = send_all[o0:o1]
    for s, lp in enumerate(lps):
        if recv_all[s].numel():
            asms[s].ctx.halo_unpack_all(recv_all[s].data_ptr())
    torch.cuda.synchronize()
    for lp, a in zip(lps, asms):
        a.allocate_matrix()
        a.ctx.assemble_device(False, [m.data_ptr() for m in a.system_pde_matrix], a.system_pde_residual.data_ptr(),
                              a.system_total_residual.data_ptr())
        a.synchronize()
        gi, no = lp.global_ids, lp.n_owned
        res = a.system_pde_residual.cpu().numpy()
        gd_u = (gi[:no, None] * dim + np.arange(dim)[None, :]).ravel()
        assert linf_scaled(res[:no * dim], ref_res[gd_u]) < 1e-12
        assert linf_scaled(res[no * dim:], ref_res[N * dim + gi[:no]]) < 1e-12
        for b in range(4):
            rp, ci = a.ctx.pattern(b)
            vals = a.system_pde_matrix[b].cpu().numpy()
            grp, gci = ref_pat[b]
            ncr = dim if b in (0, 1) else 1
            ncc = dim if b in (0, 2) else 1
            for lr in range(no * ncr):
                node, c = divmod(lr, ncr)
                grow = gi[node] * ncr + c
                lcols = ci[rp[lr]:rp[lr + 1]]
                gcols = gi[lcols // ncc] * ncc + lcols % ncc
                want = dict(zip(gci[grp[grow]:grp[grow + 1]], ref_vals[b][grp[grow]:grp[grow + 1]]))
                got = dict(zip(gcols, vals[rp[lr]:rp[lr + 1]]))
                assert set(got) == set(want)
                scale = max(1.0, max(abs(x) for x in want.values()))
                assert max(abs(got[k] - want[k]) for k in want) < 1e-12 * scale


@pytest.mark.parametrize("world,cells,extra", [(2, 40, []), (8, 36, []), (4, 300, ["--dim", "2", "--residual-only"])])
def test_bench_multiprocess_flow_reproduces_single_rank_sums(world, cells, extra):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one process per rank), with all
    ranks on this GPU and the ghost import over gloo (PFM_BENCH_SMOKE_GLOO=1): partition-independent sums of every
    matrix block and residual must agree with the single-process run."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--cells", str(cells), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--checksum"] + extra
    env = dict(os.environ, PFM_BENCH_SMOKE_GLOO="1")

    def run(cmd):
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)

    one = run([sys.executable, "bench.py", "--gpus", "1"] + common)
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))  # a free port for the rendezvous
    port = sock.getsockname()[1]
    sock.close()
    many = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", str(world)] + common)
    assert many["n_gpus"] == world and one["n_gpus"] == 1
    # gloo smoke run: no RCCL communicator exists, and the line says so (on an N-GPU box the fields carry ncclCommCount /
    # ncclGetVersion of the communicator the exchange ran on: test_bench_two_gpus_over_rccl)
    assert "rccl_nranks" in many and many["rccl_nranks"] is None and many["rccl_version"] is None
    assert "rccl_nranks" not in one
    # what the driver's SCALE record needs beside the value: the cut, the measured ghost import, and -- where the headline
    # cut is z-slabs -- the same steps on the near-cubic grid a p4est host would hand over
    assert many["config"]["partition"] and many["config"]["peers"] >= 1 and many["exchange_ms"] > 0.0
    # every rank's own kernel time (gathered outside the timed region): compute imbalance apart from exchange cost
    pr = many["per_rank_kernel_ms"]
    assert len(pr["all"]) == world and 0.0 < pr["min"] <= pr["max"]
    assert len(many["per_rank"]["owned_nodes"]) == world and sum(many["per_rank"]["owned_nodes"]) == (cells + 1) ** (2 if extra else 3)
    if world == 2 and not extra:
        assert many["config"]["partition"] == "1x1x2"
        cp = many["cubic_partition"]
        assert cp["partition"] == "2x1x1" and cp["ms_per_step"] > 0 and cp["exchange_ms"] > 0 and cp["max_peers"] == 1
        assert len(cp["per_rank_kernel_ms"]["all"]) == 2 and cp["per_rank_kernel_ms"]["min"] > 0
    a, b = np.array(one["checksum"]), np.array(many["checksum"])
    assert a.shape == b.shape and np.abs(a).max() > 0
    scale = np.abs(a).reshape(-1, 2)[:, 1].repeat(2)  # each pair is (sum, sum of absolute values)
    assert (np.abs(a - b) <= 1e-11 * np.maximum(scale, 1e-300)).all(), (a, b)


def _bench_json(cmd, env, timeout=900):
    import json
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_gpus_n_as_typed_self_launches():
    """`python bench.py --gpus 2` exactly as typed (no torch.distributed.run in front): the script re-executes itself
    under the launcher and rank 0 prints the line.  All ranks on this GPU, ghost import over gloo (smoke switch)."""
    import os
    import sys

    env = dict(os.environ, PFM_BENCH_SMOKE_GLOO="1")
    env.pop("WORLD_SIZE", None)
    common = ["--n", "24", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--checksum"]
    one = _bench_json([sys.executable, "bench.py", "--gpus", "1"] + common, env)
    two = _bench_json([sys.executable, "bench.py", "--gpus", "2"] + common, env)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    a, b = np.array(one["checksum"]), np.array(two["checksum"])
    scale = np.abs(a).reshape(-1, 2)[:, 1].repeat(2)
    assert (np.abs(a - b) <= 1e-11 * np.maximum(scale, 1e-300)).all(), (a, b)


def test_bench_two_gpus_over_rccl():
    """The same over real RCCL, one GPU per rank: runs wherever `pytest -m gpu` finds two GPUs."""
    import os
    import sys

    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("PFM_BENCH_SMOKE_GLOO", None)
    common = ["--n", "40", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--checksum"]
    one = _bench_json([sys.executable, "bench.py", "--gpus", "1"] + common, env)
    two = _bench_json([sys.executable, "bench.py", "--gpus", "2"] + common, env)
    assert two["n_gpus"] == 2
    assert two["rccl_nranks"] == 2 and two["rccl_version"] >= 20000  # what RCCL itself counted on the exchange's communicator
    a, b = np.array(one["checksum"]), np.array(two["checksum"])
    scale = np.abs(a).reshape(-1, 2)[:, 1].repeat(2)
    assert (np.abs(a - b) <= 1e-11 * np.maximum(scale, 1e-300)).all(), (a, b)


@pytest.mark.parametrize("world,grid", [(4, "2,2,1"), (8, "2,2,2"), (8, "1,1,8")])
def test_bench_more_gpus_over_rccl_subcubes(world, grid):
    """4 and 8 ranks over real RCCL with p4est-like sub-cube partitions (up to 7 peers per rank: faces, edges, corners in
    the exchange lists) and with the z-slabs of the headline run; partition-independent sums against the 1-rank ones."""
    import os
    import sys

    import torch

    if torch.cuda.device_count() < world:
        pytest.skip(f"needs >= {world} GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("PFM_BENCH_SMOKE_GLOO", None)
    common = ["--n", "40", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--checksum"]
    one = _bench_json([sys.executable, "bench.py", "--gpus", "1"] + common, env)
    many = _bench_json([sys.executable, "bench.py", "--gpus", str(world)] + common, dict(env, PFM_BENCH_GRID=grid))
    assert many["n_gpus"] == world and many["config"]["partition"] == grid.replace(",", "x") and many["exchange_ms"] > 0.0
    assert many["rccl_nranks"] == world
    a, b = np.array(one["checksum"]), np.array(many["checksum"])
    scale = np.abs(a).reshape(-1, 2)[:, 1].repeat(2)
    assert (np.abs(a - b) <= 1e-11 * np.maximum(scale, 1e-300)).all(), (a, b)


def test_library_rccl_binds_inside_a_torch_process():
    """The in-library transport binds RCCL with dlopen at first use.  In a torch process the copy torch bundles is already
    loaded; the library must find a usable one (same major version as the header it was compiled with), make an id and a
    1-rank communicator on this GPU and destroy it again -- what every rank of `bench.py --gpus N` does first."""
    import ctypes as C

    import torch  # noqa: F401  (the bundled RCCL is in the process)
    from cracks_amd import capi

    lib = capi.load()
    uid = np.zeros(capi.COMM_ID_BYTES, np.uint8)
    assert lib.pfm_comm_unique_id(capi.np_ptr(uid, np.uint8)) == capi.PFM_OK
    assert uid.any()
    h = C.c_void_p()
    assert lib.pfm_comm_create(C.byref(h), capi.np_ptr(uid, np.uint8), 1, 0, 0) == capi.PFM_OK
    assert h.value
    assert lib.pfm_comm_destroy(h) == capi.PFM_OK
