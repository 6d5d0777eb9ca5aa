"""Compile the HIP extension (C ABI of include/pfm_assemble.h) for gfx950, in-tree.

Every source of csrc/ is compiled to its own object (in parallel, rebuilt only when it or a header is newer), then
linked into csrc/libpfm_hip.so.  Objects live in csrc/_obj/ (git-ignored; they need not travel to the GPU box)."""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(CSRC, "libpfm_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]
# Per-source flags.  The marching kernels run their phases inside a plane loop with a 128-register budget: LLVM's
# machine-level loop-invariant code motion hoists constant materialisations and address arithmetic in front of the
# loop and keeps -- or spills -- them across every phase (k_cart_uu3: 128 registers + spills with it, 116 without).
EXTRA_FLAGS = {
    "pfm_cart_uu3.hip": ["-mllvm", "-disable-machine-licm"],
    "pfm_cart.hip": ["-mllvm", "-disable-machine-licm"],  # k_cart_residual3: 15 -> 13 spilled registers, 1.045 -> 1.014 ms at 216^3
}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _sources():
    return [s for s in sorted(os.listdir(CSRC)) if s.endswith((".hip", ".cpp"))]


def _headers():
    inc = os.path.join(HERE, "..", "include")
    hs = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")]
    hs += [os.path.join(inc, h) for h in sorted(os.listdir(inc)) if h.endswith(".h")]
    return hs + [os.path.abspath(__file__)]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in _sources()] + _headers()
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cc = hipcc()
    hdr_t = max(os.path.getmtime(h) for h in _headers() if os.path.exists(h))
    jobs = []
    objs = []
    for s in _sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([cc] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-I" + os.path.join(rocm, "include"), "-x", "hip", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    # RCCL (the in-library ghost exchange, pfm_halo_exchange) is resolved at run time (dlopen in pfm_host.cpp): the
    # library loads on hosts without RCCL and binds to whatever copy the process already carries (torch's, or ROCm's)
    run([cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB, "-ldl"])
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
