"""ctypes binding of the C ABI declared in include/pfm_assemble.h (libpfm_hip.so).

The library is the product; this module only marshals pointers.  There is no CPU
fallback: if the extension is missing or no GPU is present the calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libpfm_hip.so")

PFM_OK = 0
STATUS_NAMES = {0: "PFM_OK", 1: "PFM_ERR_BAD_ARG", 2: "PFM_ERR_HIP", 3: "PFM_ERR_NOT_ORTHOGONAL",
                4: "PFM_ERR_NONFINITE", 5: "PFM_ERR_UNSUPPORTED", 6: "PFM_ERR_NOMEM", 7: "PFM_ERR_COMM", 8: "PFM_ERR_INTERNAL"}
COMM_ID_BYTES = 128
LAYOUT_INTERLEAVED, LAYOUT_BLOCKED = 0, 1

# every symbol include/pfm_assemble.h declares
EXPORTS = [
    "pfm_ctx_create", "pfm_ctx_destroy", "pfm_last_error", "pfm_ctx_set_stream", "pfm_set_params",
    "pfm_set_constraints", "pfm_pattern_size", "pfm_pattern_get", "pfm_pattern_bind", "pfm_pattern_bind_i32",
    "pfm_state_set", "pfm_state_set_solution", "pfm_comm_unique_id", "pfm_comm_create", "pfm_comm_wrap", "pfm_comm_destroy", "pfm_comm_aborted", "pfm_comm_info", "pfm_halo_exchange", "pfm_assemble_overlapped",
    "pfm_check_finite",
    "pfm_halo_register", "pfm_halo_pack", "pfm_halo_unpack", "pfm_halo_pack_all", "pfm_halo_unpack_all",
    "pfm_assemble_device", "pfm_assemble_nl_residual_device",
    "pfm_sync_status", "pfm_assemble", "pfm_host_register", "pfm_host_unregister", "pfm_values_to_host", "pfm_ctx_kernel_path", "pfm_ctx_force_path", "pfm_ctx_overlay_info", "pfm_ctx_force_phase",
    "pfm_ctx_device_bytes", "pfm_timing_enable", "pfm_kernel_time_ms", "pfm_kernel_times_ms",
    # include/pfm_newton.h
    "pfm_diag_mass_device", "pfm_active_set_device", "pfm_get_constraints", "pfm_functionals",
    "pfm_functionals_material", "pfm_residual_norms",
]


class PfmParams(C.Structure):
    """include/pfm_params.h"""
    _fields_ = [
        ("lambda_", C.c_double), ("mu", C.c_double), ("G_c", C.c_double),
        ("alpha_eps", C.c_double), ("constant_k", C.c_double), ("pressure", C.c_double),
        ("alpha_biot", C.c_double), ("gamma_penal", C.c_double), ("timestep", C.c_double),
        ("time", C.c_double), ("old_timestep", C.c_double), ("old_old_timestep", C.c_double),
        ("decompose_stress_rhs", C.c_double), ("decompose_stress_matrix", C.c_double),
        ("timestep_number", C.c_int), ("outer_solver", C.c_int),
        ("use_old_timestep_pf", C.c_int), ("reserved", C.c_int),
    ]

    @classmethod
    def from_any(cls, other) -> "PfmParams":
        """Copy from any ctypes structure with the same field names (e.g. the test mirror)."""
        return cls(**{name: getattr(other, name) for name, _ in cls._fields_})


class PfmMeshDesc(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("layout", C.c_int32), ("n_nodes", C.c_int32), ("n_owned_nodes", C.c_int32),
        ("n_cells", C.c_int64), ("cell_nodes", C.c_void_p), ("coords", C.c_void_p),
        ("cell_lambda", C.c_void_p), ("cell_mu", C.c_void_p), ("n_hanging", C.c_int32),
        ("hn_nodes", C.c_void_p), ("hn_ptr", C.c_void_p), ("hn_parents", C.c_void_p),
        ("hn_weights", C.c_void_p), ("box_cells", C.c_int32 * 3),
    ]


class PfmError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: {STATUS_NAMES.get(status, status)} {detail}".strip())


_LIB = None


def load():
    """Load libpfm_hip.so (building it first if the sources are newer and hipcc exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        from . import build
        build.build_native()
    # PyTorch wheels bundle their own HIP runtime (same SONAME as /opt/rocm's).  Import torch
    # first so that exactly one runtime lives in the process and torch's device pointers /
    # streams are valid for our launches; a pure C/C++ host just uses the system ROCm.
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover
        pass
    # PFM_LIB: another build of the library (A/B runs of kernel changes in one gpurun call; tools/ab.sh)
    lib = C.CDLL(os.environ.get("PFM_LIB") or LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.pfm_ctx_create.argtypes = [C.POINTER(vp), C.POINTER(PfmMeshDesc), i32]
    lib.pfm_ctx_destroy.argtypes = [vp]
    lib.pfm_last_error.argtypes = [vp]
    lib.pfm_last_error.restype = C.c_char_p
    lib.pfm_ctx_set_stream.argtypes = [vp, vp]
    lib.pfm_set_params.argtypes = [vp, C.POINTER(PfmParams)]
    lib.pfm_set_constraints.argtypes = [vp, vp]
    lib.pfm_pattern_size.argtypes = [vp, i32, C.POINTER(i64), C.POINTER(i64)]
    lib.pfm_pattern_get.argtypes = [vp, i32, vp, vp]
    lib.pfm_pattern_bind.argtypes = [vp, i32, vp, vp]
    lib.pfm_pattern_bind_i32.argtypes = [vp, i32, vp, vp]
    lib.pfm_comm_unique_id.argtypes = [vp]
    lib.pfm_comm_create.argtypes = [C.POINTER(vp), vp, i32, i32, i32]
    lib.pfm_comm_destroy.argtypes = [vp]
    lib.pfm_comm_wrap.argtypes = [C.POINTER(vp), vp]
    lib.pfm_comm_aborted.argtypes = [vp]
    lib.pfm_host_register.argtypes = [vp, vp, i64]
    lib.pfm_host_unregister.argtypes = [vp, vp]
    lib.pfm_values_to_host.argtypes = [vp, vp, vp]
    lib.pfm_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.pfm_halo_exchange.argtypes = [vp, vp, vp]
    lib.pfm_assemble_overlapped.argtypes = [vp, vp, vp, i32, vp, vp, vp]
    lib.pfm_check_finite.argtypes = [vp, vp, i64]
    lib.pfm_functionals_material.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_double)]
    lib.pfm_state_set.argtypes = [vp, vp, vp, vp, i32]
    lib.pfm_state_set_solution.argtypes = [vp, vp, i32]
    lib.pfm_halo_register.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.pfm_halo_pack.argtypes = [vp, i32, vp]
    lib.pfm_halo_unpack.argtypes = [vp, i32, vp]
    lib.pfm_halo_pack_all.argtypes = [vp, vp]
    lib.pfm_halo_unpack_all.argtypes = [vp, vp]
    lib.pfm_assemble_device.argtypes = [vp, i32, vp, vp, vp]
    lib.pfm_assemble_nl_residual_device.argtypes = [vp, vp, vp, vp]
    lib.pfm_sync_status.argtypes = [vp]
    lib.pfm_assemble.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
    lib.pfm_ctx_kernel_path.argtypes = [vp]
    lib.pfm_ctx_force_path.argtypes = [vp, i32]
    lib.pfm_ctx_overlay_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    lib.pfm_timing_enable.argtypes = [vp, i32]
    lib.pfm_kernel_time_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.pfm_kernel_times_ms.argtypes = [vp, vp, i32, C.POINTER(C.c_int)]
    lib.pfm_ctx_device_bytes.argtypes = [vp]
    lib.pfm_ctx_device_bytes.restype = i64
    lib.pfm_diag_mass_device.argtypes = [vp, vp]
    lib.pfm_active_set_device.argtypes = [vp, vp, vp, C.c_double, vp, vp, vp, C.POINTER(i64)]
    lib.pfm_get_constraints.argtypes = [vp, vp]
    lib.pfm_functionals.argtypes = [vp, vp, C.POINTER(C.c_double)]
    lib.pfm_residual_norms.argtypes = [vp, vp, C.POINTER(C.c_double)]
    _LIB = lib
    return lib


def np_ptr(a: np.ndarray, dtype) -> C.c_void_p:
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, dtype, a.flags)
    return C.c_void_p(a.ctypes.data)
