"""Compile the HIP extension (C ABI of include/pfm_assemble.h) for gfx950, in-tree."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libpfm_hip.so")
SOURCES = ["pfm_host.cpp", "pfm_kernels.hip"]
HEADERS = ["pfm_internal.h", "pfm_cart_common.h", "pfm_split.h", os.path.join("..", "..", "include", "pfm_assemble.h"),
           os.path.join("..", "..", "include", "pfm_params.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_native(force: bool = False, verbose: bool = False) -> str:
    srcs = [s for s in sorted(os.listdir(CSRC)) if s.endswith((".hip", ".cpp"))]
    global SOURCES
    SOURCES = srcs
    if not force and not needs_build():
        return LIB
    # RCCL (the in-library ghost exchange, pfm_halo_exchange): the ROCm copy; in a process that has imported torch the
    # loader resolves the same SONAME (librccl.so.1) to torch's bundled build
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    link = ["-L" + os.path.join(rocm, "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    cmd = [hipcc()] + FLAGS + ["-I" + os.path.join(rocm, "include"), "-x", "hip"] + \
          [os.path.join(CSRC, s) for s in srcs] + ["-o", LIB] + link
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
