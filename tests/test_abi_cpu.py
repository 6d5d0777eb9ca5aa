"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads and exports
every symbol include/pfm_assemble.h declares; the ctypes mirrors match the C structs;
argument validation works without touching a GPU."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from cracks_amd import build, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_native()
    return capi.load()


def _declared_symbols():
    names = set()
    for header in ("pfm_assemble.h", "pfm_newton.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(pfm_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_header_symbols_are_exported(lib):
    names = _declared_symbols()
    assert len(names) >= 15
    assert sorted(capi.EXPORTS) == names
    for n in names:
        assert hasattr(lib, n), n


def test_struct_layout_matches_c():
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "pfm_assemble.h"
int main(void){
  printf("%zu %zu %zu %zu\n", sizeof(pfm_params), offsetof(pfm_params, timestep_number),
         offsetof(pfm_params, decompose_stress_matrix), offsetof(pfm_params, reserved));
  printf("%zu %zu %zu %zu\n", sizeof(pfm_mesh_desc), offsetof(pfm_mesh_desc, n_cells),
         offsetof(pfm_mesh_desc, hn_nodes), offsetof(pfm_mesh_desc, box_cells));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])  # plain C: header is C-clean
        out = subprocess.check_output([exe]).decode().split()
    P, M = capi.PfmParams, capi.PfmMeshDesc
    assert [int(x) for x in out[:4]] == [C.sizeof(P), P.timestep_number.offset,
                                         P.decompose_stress_matrix.offset, P.reserved.offset]
    assert [int(x) for x in out[4:]] == [C.sizeof(M), M.n_cells.offset, M.hn_nodes.offset, M.box_cells.offset]


def test_bad_arguments_are_rejected_without_a_gpu(lib):
    h = C.c_void_p()
    assert lib.pfm_ctx_create(C.byref(h), None, 0) == 1  # PFM_ERR_BAD_ARG
    d = capi.PfmMeshDesc()
    d.dim = 4
    assert lib.pfm_ctx_create(C.byref(h), C.byref(d), 0) == 1
    assert lib.pfm_set_params(None, None) == 1
    assert lib.pfm_assemble_device(None, 0, None, None, None) == 1
    assert lib.pfm_ctx_destroy(None) == 0
    assert lib.pfm_last_error(None) == b"null context"


def test_raw_pointers_are_not_taken_for_communicator_handles(lib):
    """a host that passes its own ncclComm_t (or anything else) where a pfm_comm_* handle is expected gets
    PFM_ERR_BAD_ARG -- the handles carry a tag -- instead of having its memory reinterpreted."""
    fake = (C.c_uint8 * 256)()  # something that is not a handle, large enough to be read
    assert lib.pfm_comm_aborted(C.cast(fake, C.c_void_p)) == 0
    assert lib.pfm_comm_destroy(C.cast(fake, C.c_void_p)) == 1  # PFM_ERR_BAD_ARG, nothing freed
    n = C.c_int32(0)
    assert lib.pfm_comm_info(C.cast(fake, C.c_void_p), C.byref(n), None, None) == 1
    assert lib.pfm_comm_info(None, None, None, None) == 1
    assert lib.pfm_comm_destroy(None) == 0


def test_no_oracle_in_product_package():
    """The product path must not import, link or call the oracle."""
    pkg = os.path.join(ROOT, "cracks_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in text and "oracle_api" not in text and "oracle/" not in text.replace(
                    "oracle/)", ""), f


def test_bench_self_launch_command_line():
    """`python bench.py --gpus N` re-executes itself under torch.distributed.run (bench.py: self_launch_command);
    the multi-process run itself is a GPU test (tests/test_gpu_multirank.py)."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    cmd = bench.self_launch_command(8, ["--gpus", "8", "--n", "216", "--steps", "5"])
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    script = cmd.index(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--cells", "216", "--steps", "5"]
