"""Wall time of the phases of pfm_ctx_create (PFM_CTX_TIMING=1 prints them): `python tools/ctx_timing.py c5` -- the meshes
of the config-5 stand-in, two contexts each (the first of a process pays one-time costs); `python tools/ctx_timing.py 216` --
three contexts on the 216^3 box.  profiles/r05/ctx_timing_*.txt."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from cracks_amd.assembler import Assembler
from cracks_amd import mesh as M
which = sys.argv[1]
if which == "c5":
    for step in range(3):
        pb = bench.config5_problem(8, min(step, 1))
        for rep in range(2):
            print(f"--- c5 step {step} rep {rep}", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            asm = Assembler(pb["mesh"], blocked=False, device=0)
            t1 = time.perf_counter()
            asm.allocate_matrix()
            t2 = time.perf_counter()
            asm.set_params(pb["params"]); asm.set_constraints(pb["node_flags"])
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            print(f"Assembler {1e3*(t1-t0):.2f} (create {1e3*asm.ctx.create_seconds:.2f}) allocate_matrix {1e3*(t2-t1):.2f} params+constraints {1e3*(t3-t2):.2f} total {1e3*(t3-t0):.2f} ms", file=sys.stderr, flush=True)
            asm.set_vectors(*pb["vectors"]); asm.assemble_system(False); asm.synchronize()
            asm.ctx.close()
else:
    n = int(which)
    g = M.box_mesh(3, (n,)*3)
    for rep in range(3):
        print(f"--- box {n} rep {rep}", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        asm = Assembler(g, blocked=True, device=0)
        print(f"Assembler {1e3*(time.perf_counter()-t0):.2f} create {1e3*asm.ctx.create_seconds:.2f}", file=sys.stderr, flush=True)
        asm.ctx.close()
