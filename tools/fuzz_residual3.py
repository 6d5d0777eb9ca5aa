"""Round 6: the three 3-D residual kernels over random box shapes (run on the GPU box: `python tools/fuzz_residual3.py [seed]`)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import test_gpu_cart as T
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
shapes = [(1, 1, 1), (2, 1, 1), (1, 1, 40), (3, 3, 1), (14, 14, 1), (15, 1, 2), (1, 15, 2), (30, 1, 1), (1, 30, 3), (16, 16, 16)]
shapes += [tuple(int(x) for x in rng.integers(1, 48, 3)) for _ in range(30)]
for n in shapes:
    for het in (False, True) if sum(n) % 5 == 0 else (False,):
        for k in ("PFM_RES_NO_WIDE_TRANSFERS", "PFM_RES_NO_TRANSFERS"):
            os.environ.pop(k, None)
        c = T.box_case(3, n, -10.0, 10.0, True)
        if het:
            c = T.heterogeneous(c)
        ctx = T.make_context(c)
        if ctx.kernel_path != 1:
            print(n, "path", ctx.kernel_path); continue
        ctx.assemble_host(c.sol, c.old, c.oldold, True)
        sol2 = c.sol + 1e-3 * rng.standard_normal(c.sol.shape)
        nd = c.layout.n_dofs
        d_sol = torch.from_numpy(np.ascontiguousarray(sol2)).cuda()
        def run():
            bufs = [torch.full((nd,), 7.0, dtype=torch.float64, device="cuda") for _ in range(2)]
            ctx.assemble_nl_residual_device(d_sol.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr())
            ctx.sync_status()
            return bufs[0].cpu().numpy(), bufs[1].cpu().numpy()
        rx, tx = run()
        b2 = [torch.empty(nd, dtype=torch.float64, device="cuda") for _ in range(2)]
        ctx.assemble_device(True, [], b2[0].data_ptr(), b2[1].data_ptr()); ctx.sync_status()
        ok_state = np.array_equal(b2[0].cpu().numpy(), rx)
        os.environ["PFM_RES_NO_WIDE_TRANSFERS"] = "1"
        rd, td = run()
        os.environ["PFM_RES_NO_TRANSFERS"] = "1"
        ro, to = run()
        ok = np.array_equal(rx, rd) and np.array_equal(rx, ro) and np.array_equal(tx, td) and np.array_equal(tx, to) and ok_state
        if not ok:
            bad += 1
            print("MISMATCH", n, het, np.abs(rx - ro).max(), np.abs(rd - ro).max(), ok_state)
print("shapes", len(shapes), "bad", bad)
