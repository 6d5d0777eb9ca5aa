"""Six Jacobian assemblies of bench.py's 3-D overlay mesh (84^3, inner half refined) for rocprofv3 --kernel-trace:
`PFM_GENERAL_SEQUENTIAL=1 rocprofv3 --kernel-trace --stats ... -- python tools/overlay3d_profile.py` gives the sequential
durations of the level kernels and of the general family's classes (profiles/r05/ov3_kst*.txt, ov3_timeline.txt)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from cracks_amd import mesh as M
from cracks_amd.assembler import Assembler
n = 84
g0 = M.box_mesh(3, (n,) * 3)
cc = g0.coords[g0.cells].mean(axis=1)
mesh = M.refine_cells(g0, (np.abs(cc) < 5.0).all(axis=1))
h = (20.0 / n) * np.sqrt(3.0) / 2
u, phi, po, poo, flags = bench.synthetic_state(mesh, np.arange(mesh.n_nodes), h, 3)
flags[mesh.hn_nodes] = 0
a = Assembler(mesh, blocked=True, device=0)
a.set_params(bench.sneddon_params(h, 3))
a.set_constraints(flags)
pack = lambda uu, pp: np.concatenate([uu.reshape(-1), pp])
a.set_vectors(pack(u, phi), pack(0 * u, po), pack(0 * u, poo))
for it in range(6):
    a.assemble_system(False)
a.synchronize()
