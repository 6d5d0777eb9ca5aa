import sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from cracks_amd import partition as P
from cracks_amd.assembler import Assembler
def run(n, p, rank, label):
    lp = P.build_local_problem(3, n, p, rank)
    h = (20.0 / 216) * np.sqrt(3.0)
    u, phi, po, poo, flags = bench.synthetic_state(lp.mesh, lp.global_ids, h, 3)
    asm = Assembler(lp.mesh, blocked=True, n_owned_nodes=lp.n_owned)
    asm.set_params(bench.sneddon_params(h, 3)); asm.set_constraints(flags)
    no = lp.n_owned
    def pack(uu, pp):
        v = np.empty(no * 4); v[:no * 3] = uu[:no].reshape(-1); v[no * 3:] = pp[:no]; return v
    asm.set_vectors(pack(u, phi), pack(np.zeros_like(u), po), pack(np.zeros_like(u), poo))
    asm.ctx.timing_enable(True)
    for _ in range(3): asm.assemble_system(False)
    asm.synchronize(); asm.ctx.kernel_time_ms()
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t = time.perf_counter(); asm.assemble_system(False); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(label, lp.mesh.n_cells, no, "ms", round(1e3 * float(np.median(ts)), 3), flush=True)
run((216, 216, 216), (2, 1, 1), 0, "rank0of2 (x split, ghost plane on high x)")
run((108, 216, 216), (1, 1, 1), 0, "box 108x216x216 single")
run((216, 216, 108), (1, 1, 1), 0, "box 216x216x108 single")
run((216, 216, 216), (1, 1, 2), 0, "rank0 of z-split")
