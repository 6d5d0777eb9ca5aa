"""Round 6: launch variants of the 2-D Jacobian over random boxes, against each other and the oracle (`python tools/fuzz_cart2d.py [seed]`)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import test_gpu_cart as T
M = T.M
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
shapes = [(1, 1), (2, 3), (7, 7), (8, 8), (6, 20), (21, 6), (14, 1), (1, 14), (49, 50)]
shapes += [tuple(int(x) for x in rng.integers(1, 70, 2)) for _ in range(25)]
for n in shapes:
    for blocked in (True, False):
        for k in ("PFM_CART2D_NO_FILL", "PFM_CART2D_ONE_LAUNCH"):
            os.environ.pop(k, None)
        mono = bool(rng.integers(0, 2))
        c = T.box_case(2, n, -10.0, 10.0, blocked, monolithic=mono)
        if rng.integers(0, 2):
            c.params.constant_k = 0.0
            node, comp = c.layout.node_comp_of_dof()
            is_phi = comp == 2
            dead = c.mesh.coords[node[is_phi]][:, 0] < 0.0
            o = c.old.copy(); o[np.nonzero(is_phi)[0][dead]] = 0.0
            c.old, c.oldold = o, o.copy()
            phi_dofs = np.nonzero(is_phi)[0]
            c.cu = M.update_constraints(c.mesh, c.layout, M.sneddon_dirichlet_dofs(c.mesh, c.layout), phi_dofs[::5])
        if rng.integers(0, 3) == 0:
            c = T.heterogeneous(c)
        ctx = T.make_context(c)
        if ctx.kernel_path != 1:
            print(n, "path", ctx.kernel_path); continue
        def run():
            vals, rp, _ = ctx.assemble_host(c.sol, c.old, c.oldold, False)
            return [np.array(v, copy=True) for v in vals], rp.copy()
        v0, p0 = run()
        for key in ("PFM_CART2D_NO_FILL", "PFM_CART2D_ONE_LAUNCH"):
            os.environ[key] = "1"
            v1, p1 = run()
            if not (all(np.array_equal(a, b) for a, b in zip(v0, v1)) and np.array_equal(p0, p1)):
                bad += 1
                print("MISMATCH", n, blocked, key, [float(np.abs(a - b).max()) for a, b in zip(v0, v1)])
        for k in ("PFM_CART2D_NO_FILL", "PFM_CART2D_ONE_LAUNCH"):
            os.environ.pop(k, None)
        r, rp, ci = T.oracle(c, False)
        import scipy.sparse as sp
        nd = c.layout.n_dofs
        A_ref = sp.csr_matrix((r.values, ci, rp), shape=(nd,) * 2)
        A = T.blocks_to_global(ctx, c.layout, v0); A.sort_indices()
        e = T.linf_scaled(A.data, A_ref.data)
        if not e < T.TOL:
            bad += 1; print("ORACLE", n, blocked, e)
print("shapes", len(shapes), "bad", bad)
