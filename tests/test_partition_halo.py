"""Owner-computes partition + halo lists (CPU): structural checks for 1/2/4/8 ranks and a
world_size-2 gloo run in which every rank assembles its sub-mesh (with the oracle standing
in for the device kernels) and must reproduce the single-rank rows it owns."""
import os
import socket

import numpy as np
import pytest

import cases
import oracle_api as O
from cracks_amd import mesh as M
from cracks_amd import partition as P


@pytest.mark.parametrize("dim,n,world", [(2, (7, 5), 2), (2, (6, 6), 4), (3, (5, 4, 6), 2), (3, (4, 4, 4), 8),
                                         (3, (5, 6, 4), 4), (3, (3, 3, 3), 1)])
def test_partition_structure(dim, n, world):
    p = P.factor_ranks(world, dim)
    assert int(np.prod(p)) == world
    g = M.box_mesh(dim, n)
    probs = [P.build_local_problem(dim, n, p, r) for r in range(world)]
    owner_count = np.zeros(g.n_nodes, int)
    for r, lp in enumerate(probs):
        owner_count[lp.global_ids[:lp.n_owned]] += 1
        assert (P.owner_of_nodes(n, p, lp.global_ids[:lp.n_owned]) == r).all()
        assert np.allclose(lp.mesh.coords, g.coords[lp.global_ids])
        # every global cell that touches an owned node is present locally
        owned = np.zeros(g.n_nodes, bool)
        owned[lp.global_ids[:lp.n_owned]] = True
        need = {tuple(sorted(c)) for c in g.cells[owned[g.cells].any(axis=1)]}
        have = {tuple(sorted(lp.global_ids[c])) for c in lp.mesh.cells}
        assert need <= have
        # local cell vertex order is deal.II's
        x = lp.mesh.coords[lp.mesh.cells]
        assert (x[:, 1, 0] > x[:, 0, 0]).all() and (x[:, 2, 1] > x[:, 0, 1]).all()
    assert (owner_count == 1).all()
    # send/recv lists are mirror images in global ids
    for r, lp in enumerate(probs):
        for k, s in enumerate(lp.peers):
            sent = lp.global_ids[lp.send_nodes[lp.send_ptr[k]:lp.send_ptr[k + 1]]]
            other = probs[s]
            ko = other.peers.index(r)
            recvd = other.global_ids[other.recv_nodes[other.recv_ptr[ko]:other.recv_ptr[ko + 1]]]
            assert (sent == recvd).all()
        ghosts = set(range(lp.n_owned, lp.mesh.n_nodes))
        assert set(int(k) for k in lp.recv_nodes) == ghosts


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, dim, n, q):
    import torch
    import torch.distributed as dist

    from cracks_amd.halo import HaloExchange

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = P.factor_ranks(world, dim)
        lp = P.build_local_problem(dim, n, p, rank)
        nloc = lp.mesh.n_nodes
        rec = dim + 3
        # global fields, function of the global node id only
        gmesh = M.box_mesh(dim, n)
        rng = np.random.default_rng(11)
        gfield = rng.uniform(0.1, 0.9, (gmesh.n_nodes, rec))
        gfield[:, :dim] = (gfield[:, :dim] - 0.5) * 2e-3
        state = np.full((nloc, rec), np.nan)
        state[:lp.n_owned] = gfield[lp.global_ids[:lp.n_owned]]
        hx = HaloExchange(dim, lp.peers, lp.send_ptr, lp.send_nodes, lp.recv_ptr, lp.recv_nodes, torch.device("cpu"))

        def pack(k, nodes):
            return torch.from_numpy(np.ascontiguousarray(state[nodes].T))  # field-major like the HIP kernel

        def unpack(k, nodes, buf):
            state[nodes] = buf.numpy().reshape(rec, nodes.size).T

        hx.exchange_with(pack, unpack)
        assert np.array_equal(state, gfield[lp.global_ids]), "ghost import wrong"

        # owner-computes assembly of the local sub-mesh == owned rows of the global assembly
        base = cases.kat_sneddon_3d(4) if dim == 3 else cases.kat_sneddon_2d()
        prm = base.params
        lay = M.DofLayout(nloc, dim, blocked=False)
        glay = M.DofLayout(gmesh.n_nodes, dim, blocked=False)

        def vecs(layout, f):
            sol = layout.pack(f[:, :dim], f[:, dim])
            old = layout.pack(np.zeros((f.shape[0], dim)), f[:, dim + 1])
            oo = layout.pack(np.zeros((f.shape[0], dim)), f[:, dim + 2])
            return sol, old, oo

        def constraints(mesh, layout):
            return (M.update_constraints(mesh, layout, M.sneddon_dirichlet_dofs(mesh, layout)),
                    M.hanging_constraints(mesh, layout))

        cu, ch = constraints(lp.mesh, lay)
        rp, ci = M.dof_sparsity(lp.mesh, lay)
        loc = O.assemble(lp.mesh, lay, prm, *vecs(lay, state), cu, ch, False, rp, ci)
        gcu, gch = constraints(gmesh, glay)
        grp, gci = M.dof_sparsity(gmesh, glay)
        glob = O.assemble(gmesh, glay, prm, *vecs(glay, gfield), gcu, gch, False, grp, gci)
        assert loc.err == 0 and glob.err == 0
        import scipy.sparse as sp
        A = sp.csr_matrix((loc.values, ci, rp), shape=(lay.n_dofs,) * 2)
        G = sp.csr_matrix((glob.values, gci, grp), shape=(glay.n_dofs,) * 2)
        nc = dim + 1
        gdof = (lp.global_ids[:, None] * nc + np.arange(nc)[None, :]).ravel()  # local dof -> global dof
        own = np.arange(lp.n_owned * nc)
        Aown = A[own].tocoo()
        Gsub = G[gdof[own]][:, gdof].tocoo()
        d = (sp.csr_matrix((Aown.data, (Aown.row, Aown.col)), shape=(own.size, lay.n_dofs)) -
             sp.csr_matrix((Gsub.data, (Gsub.row, Gsub.col)), shape=(own.size, lay.n_dofs)))
        scale = max(1.0, np.abs(G.data).max())
        assert (np.abs(d.data).max() if d.nnz else 0.0) < 1e-13 * scale
        assert np.abs(loc.residual_pde[own] - glob.residual_pde[gdof[own]]).max() < 1e-13
        # the owned rows of G have no columns outside the local node set
        assert G[gdof[own]].nnz == Gsub.nnz
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dim,n", [(3, (5, 4, 4)), (2, (8, 6))])
def test_world2_gloo_halo_and_owner_computes(dim, n):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, dim, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
