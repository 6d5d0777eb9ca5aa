#!/usr/bin/env python3
"""One-off robustness run on the GPU box: (1) 200 Jacobian assemblies of the 3-D Sneddon 216^3 bench problem must
leave bit-identical outputs and a stable amount of free device memory; (2) 60 context create/destroy cycles on a
40^3 box must give the memory back.  `python tools/stress.py 64 general`: the same box forced onto the general family.
`python tools/stress.py 24 hanging`: a 3-D box with a refined block (hanging nodes on its faces and edges) -- 100 assemblies,
reports whether the outputs are bitwise equal run to run and, if not, the largest deviation relative to the row's largest
entry.  Default since round 6: the cells at hanging vertices write scratch, k_hanging_gather adds in list order (bitwise).
`PFM_HANGING_ATOMIC=1 python tools/stress.py 24 hanging`: the class with FP64 atomic adds of rounds 4-5 (no fixed order);
`PFM_HANGING_COLOURED=1 ...`: those cells in plain colour classes (round 5, bitwise, some thirty small launches)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import bench
    from cracks_amd import mesh as M
    from cracks_amd.assembler import Assembler

    def problem(n):
        g = M.box_mesh(3, (n,) * 3)
        h = (20.0 / n) * np.sqrt(3.0)
        u, phi, po, poo, flags = bench.synthetic_state(g, np.arange(g.n_nodes), h, 3)
        a = Assembler(g, blocked=True)
        a.set_params(bench.sneddon_params(h, 3))
        a.set_constraints(flags)
        N = g.n_nodes
        pack = lambda uu, pp: np.concatenate([uu.reshape(-1), pp])
        a.set_vectors(pack(u, phi), pack(0 * u, po), pack(0 * u, poo))
        return a

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
    if len(sys.argv) > 2 and sys.argv[2] == "hanging":
        g0 = M.box_mesh(3, (n,) * 3)
        c = g0.coords[g0.cells].mean(axis=1)
        g = M.refine_cells(g0, (np.abs(c) < 5.0).all(axis=1))  # the inner block one level down
        h = (20.0 / n) * np.sqrt(3.0) / 2
        u, phi, po, poo, flags = bench.synthetic_state(g, np.arange(g.n_nodes), h, 3)
        flags[g.hn_nodes] = 0
        a = Assembler(g, blocked=True)
        a.set_params(bench.sneddon_params(h, 3))
        a.set_constraints(flags)
        pack = lambda uu, pp: np.concatenate([uu.reshape(-1), pp])
        a.set_vectors(pack(u, phi), pack(0 * u, po), pack(0 * u, poo))
        a.assemble_system(False)
        a.synchronize()
        ref = [m.clone() for m in a.system_pde_matrix] + [a.system_pde_residual.clone()]
        worst, differing = 0.0, 0
        for it in range(100):
            for m in a.system_pde_matrix:
                m.fill_(float(it))
            a.assemble_system(False)
            a.synchronize()
            for x, y in zip(ref, list(a.system_pde_matrix) + [a.system_pde_residual]):
                if not torch.equal(x, y):
                    differing += 1
                    worst = max(worst, float(((x - y).abs().max() / x.abs().max().clamp_min(1e-300)).item()))
        print(f"general family with hanging nodes ({g.n_cells} cells, {g.hn_nodes.size} hanging nodes, kernel path {a.ctx.kernel_path}): "
              f"100 assemblies, {differing} output arrays differed from the first run, largest deviation {worst:.2e} of the array's "
              f"largest entry" + (" (bitwise reproducible)" if differing == 0 else " (atomic class: bounded, not bitwise)")
              + (" [PFM_HANGING_COLOURED=1]" if os.environ.get("PFM_HANGING_COLOURED") else ""))
        assert worst < 1e-13
        return
    a = problem(n)
    if len(sys.argv) > 2 and sys.argv[2] == "general":  # python tools/stress.py 64 general: the colour classes of the general family
        a.ctx.force_path(0)
    a.assemble_system(False)
    a.synchronize()
    ref = [m.clone() for m in a.system_pde_matrix] + [a.system_pde_residual.clone()]
    free0 = torch.cuda.mem_get_info()[0]
    for it in range(200):
        for m in a.system_pde_matrix:
            m.fill_(float(it))  # whatever was there must be overwritten
        a.assemble_system(False)
        if it % 50 == 49:
            a.synchronize()
            print(f"  after {it + 1}: free {torch.cuda.mem_get_info()[0] / 2**30:.3f} GiB", flush=True)
    a.synchronize()
    free1 = torch.cuda.mem_get_info()[0]  # before the comparison below, whose temporaries torch keeps cached
    assert abs(free0 - free1) < (64 << 20), "device memory grew during the assemblies"
    now = list(a.system_pde_matrix) + [a.system_pde_residual]
    assert all(torch.equal(x, y) for x, y in zip(ref, now)), "outputs changed between assemblies"
    print(f"200 assemblies at {n}^3: bit-identical; free memory {free0 / 2**30:.2f} -> {free1 / 2**30:.2f} GiB")
    del a, ref, now
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    for it in range(60):
        b = problem(40)
        b.assemble_system(False)
        b.synchronize()
        del b
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    print(f"60 context create/destroy cycles: free memory {free0 / 2**30:.2f} -> {free1 / 2**30:.2f} GiB")
    assert abs(free0 - free1) < (64 << 20)


if __name__ == "__main__":
    main()
