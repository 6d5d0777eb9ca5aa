#!/usr/bin/env python3
"""Heterogeneous material on the cartesian family: time per full 3-D assembly with and without per-cell Lame
coefficients on a 120^3 box (run on the GPU box: python tools/het_bench.py)."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cracks_amd import mesh as M
from cracks_amd.assembler import Assembler
n = 120
g = M.box_mesh(3, (n,)*3)
h = (20.0/n)*np.sqrt(3.0)
u, phi, po, poo, flags = bench.synthetic_state(g, np.arange(g.n_nodes), h, 3)
rng = np.random.default_rng(1)
E = 1.0 + rng.uniform(1.0, 10.0, g.n_cells)
mu = E/2.4; lam = 0.4*mu/0.6
pack = lambda uu, pp: np.concatenate([uu.reshape(-1), pp])
for het in (False, True):
    a = Assembler(g, blocked=True, cell_lambda=lam if het else None, cell_mu=mu if het else None)
    a.set_params(bench.sneddon_params(h, 3)); a.set_constraints(flags)
    a.set_vectors(pack(u, phi), pack(0*u, po), pack(0*u, poo))
    for _ in range(3): a.assemble_system(False)
    a.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): a.assemble_system(False)
    a.synchronize(); torch.cuda.synchronize()
    print('het' if het else 'hom', 'path', a.ctx.kernel_path, 'ms per assembly', (time.perf_counter()-t0)/10*1e3)
    a.ctx.close()
