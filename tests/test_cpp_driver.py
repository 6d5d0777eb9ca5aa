"""A C++ host (no Python, no torch in the process) drives create -> pattern bind -> RCCL halo exchange -> assemble through
the C ABI (tests/cpp/abi_driver.cpp).  The CPU part checks that the driver compiles and links against the library and
header alone; the GPU part runs it: 1-rank RCCL communicator, ncclSend/ncclRecv of the ghost plane to itself."""
import os
import shutil
import subprocess

import pytest

from cracks_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "abi_driver.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "abi_driver")


def build_driver(force=False):
    lib = build.build_native()
    if not force and os.path.exists(EXE) and os.path.getmtime(EXE) > max(os.path.getmtime(SRC), os.path.getmtime(lib)):
        return EXE
    libdir = os.path.dirname(lib)
    cmd = [build.hipcc(), "-std=c++17", "-O1", SRC, "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-lpfm_hip",
           "-Wl,-rpath," + libdir, "-o", EXE]
    subprocess.check_call(cmd)
    return EXE


def test_driver_compiles_against_the_public_header_only():
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    exe = build_driver(force=True)
    assert os.path.exists(exe)
    text = open(SRC).read()
    assert "pfm_internal.h" not in text and "torch" not in text.replace("no torch", "")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(17, 9, 11), (8, 29, 5)])
def test_cpp_host_create_halo_assemble(shape):
    exe = build_driver()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe] + [str(k) for k in shape], capture_output=True, text=True, timeout=600, env=env)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_driver: OK" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [4, 8])
def test_cpp_host_more_ranks_over_rccl(ranks):
    """The ring exchange of the C++ host with 4 and 8 forked ranks (one GPU each) wherever the box has them."""
    import torch

    if torch.cuda.device_count() < ranks:
        pytest.skip(f"needs >= {ranks} GPUs")
    exe = build_driver()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, "13", "9", "10", "--ranks", str(ranks)], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"abi_driver: OK ({ranks} ranks)" in r.stdout


@pytest.mark.gpu
def test_cpp_host_two_ranks_over_rccl():
    """Two forked ranks, one GPU each, RCCL id over a pipe, ring exchange of the ghost planes through
    pfm_comm_create / pfm_halo_exchange: the first box with >= 2 GPUs that runs `pytest -m gpu` exercises the in-library
    transport with more than one rank.  Skipped on single-GPU boxes."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the 1-rank self-exchange above covers the call sequence)")
    exe = build_driver()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, "13", "9", "10", "--ranks", "2"], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_driver: OK (2 ranks)" in r.stdout
