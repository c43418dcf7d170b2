"""ctypes binding of libfeasthip.so (include/feasthip.h).  No torch types cross the ABI:
device buffers are passed as raw integer addresses (``tensor.data_ptr()``)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FEASTHIP_LIB: another build of the same library (kernel experiments: tools/ scripts compare variants in one GPU session)
LIB_PATH = os.environ.get("FEASTHIP_LIB") or os.path.join(_HERE, "libfeasthip.so")


class FeastHipStats(C.Structure):
    _fields_ = [
        ("seconds_total", C.c_double),
        ("seconds_solve", C.c_double),
        ("krylov_iterations", C.c_int64),
        ("spmm_calls", C.c_int64),
        ("factorizations", C.c_int64),
        ("max_rel_residual", C.c_double),
    ]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class FeastHipPolicy(C.Structure):
    """feasthip_policy (include/feasthip.h): state of the inexact-mode host policy for one solve."""
    _fields_ = [("Emin", C.c_double), ("Emax", C.c_double), ("inner_rtol", C.c_double), ("outer_tol", C.c_double),
                ("ne", C.c_int), ("quadrature", C.c_int), ("steer", C.c_int), ("aspect", C.c_int), ("cap", C.c_int),
                ("inner_cap", C.c_int), ("base_cap", C.c_int), ("n_hist", C.c_int), ("eps_prev", C.c_double),
                ("eps_hist", C.c_double * 3), ("next_rtol", C.c_double), ("last_reach", C.c_double)]


# every symbol include/feasthip.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
_pi, _pd, _pi64 = C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int64)
_ps = C.POINTER(FeastHipStats)
SYMBOLS = {
    "feasthip_version": (_i, [_pi, _pi]),
    "feasthip_create": (_i, [C.POINTER(_vp), _i]),
    "feasthip_destroy": (_i, [_vp]),
    "feasthip_last_error": (C.c_char_p, [_vp]),
    "feasthip_set_stream": (_i, [_vp, _vp]),
    "feasthip_synchronize": (_i, [_vp]),
    "feasthip_comm_unique_id": (_i, [C.c_char_p]),
    "feasthip_comm_init_rank": (_i, [_vp, _i, _i, C.c_char_p, _i]),
    "feasthip_comm_destroy": (_i, [_vp]),
    "feasthip_comm_info": (_i, [_vp, _pi, _pi, _pi]),
    "feasthip_allreduce_sum_dev": (_i, [_vp, _vp, _i64]),
    "feasthip_set_column_block": (_i, [_vp, _i64, _i64]),
    "feasthip_set_dense": (_i, [_vp, _i64, _i, _vp, _i64, _vp, _i64]),
    "feasthip_set_csr": (_i, [_vp, _i64, _i, _i, _i, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "feasthip_set_contour": (_i, [_vp, _i, _vp, _vp, _d]),
    "feasthip_set_real_projection": (_i, [_vp, _i]),
    "feasthip_set_node_range": (_i, [_vp, _i, _i]),
    "feasthip_set_node_list": (_i, [_vp, _i, _vp]),
    "feasthip_set_solver": (_i, [_vp, _i, _d, _d, _i, _i, _i, _i]),
    "feasthip_set_column_mask": (_i, [_vp, _i64, _vp]),
    "feasthip_rayleigh_ritz_dev": (_i, [_vp, _i64, _vp, _d, _d, _i, _vp, _vp, _vp, _vp]),
    "feasthip_contour_apply": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _ps]),
    "feasthip_contour_apply_dev": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _ps]),
    "feasthip_contour_apply_resident": (_i, [_vp, _i64, _vp, _vp, _vp, _ps]),
    "feasthip_rr_reduce_resident": (_i, [_vp, _i64, _d, _i, _pi, _vp, _vp]),
    "feasthip_rr_ritz_resident": (_i, [_vp, _i64, _vp, _vp, _i64, _i, _i, _vp]),
    "feasthip_resident_export": (_i, [_vp, _i, _i64, _vp]),
    "feasthip_resident_import": (_i, [_vp, _i, _i64, _vp]),
    "feasthip_orthonormalize": (_i, [_vp, _i64, _vp, _d, _pi]),
    "feasthip_orthonormalize_dev": (_i, [_vp, _i64, _vp, _d, _pi]),
    "feasthip_project": (_i, [_vp, _i64, _vp, _i, _i, _vp, _vp]),
    "feasthip_project_dev": (_i, [_vp, _i64, _vp, _i, _i, _vp, _vp]),
    "feasthip_ritz_residual": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp]),
    "feasthip_ritz_residual_dev": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp]),
    "feasthip_matmul": (_i, [_vp, _i, _i64, _vp, _vp]),
    "feasthip_matmul_dev": (_i, [_vp, _i, _i64, _vp, _vp]),
    "feasthip_band_plan": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "feasthip_direct_plan_flops": (_i, [_vp, _vp]),
    "feasthip_release_factors": (_i, [_vp]),
    "feasthip_shifted_solve": (_i, [_vp, _d, _d, _i64, _vp, _vp, _ps]),
    "feasthip_shifted_solve_dev": (_i, [_vp, _d, _d, _i64, _vp, _vp, _ps]),
    "feasthip_last_node_iterations": (_i, [_vp, _vp, _i]),
    "feasthip_last_column_iterations": (_i, [_vp, _vp, _i]),
    "feasthip_last_global_node_iterations": (_i, [_vp, _vp, _i]),
    "feasthip_profile_enable": (_i, [_vp, _i]),
    "feasthip_profile_reset": (_i, [_vp]),
    "feasthip_profile_get": (_i, [_vp, C.c_char_p, _pd, _pi64]),
    "feasthip_profile_set_period": (_i, [_vp, _i]),
    "feasthip_profile_get_work": (_i, [_vp, C.c_char_p, _pd]),
    "feasthip_policy_init": (_i, [_vp, _d, _d, _i, _i, _d, _d, _i, _i, _i]),
    "feasthip_policy_update": (_i, [_vp, _d, _i, _i, _vp, _i]),
    "feasthip_policy_set_aside": (_i, [_vp, _i, _vp]),
    "feasthip_policy_filter_ratio": (_d, [_d, _d, _i, _i, _i, _d, _vp, _i]),
    "feasthip_policy_reach": (_d, [_vp, _i, _d, _d, _d]),
}

UNIQUE_ID_BYTES = 128      # FEASTHIP_UNIQUE_ID_BYTES

_lib = None


class FeastHipUnavailable(RuntimeError):
    """libfeasthip.so is missing or no MI355X is visible.  There is no CPU fallback."""


def load_library(path: str | None = None):
    """dlopen libfeasthip.so and type every entry point.  Needs no GPU."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FeastHipUnavailable(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  feastkit.jl_amd has no CPU fallback.")
    # PyTorch bundles its own HIP runtime (libamdhip64); load torch FIRST so that this library's
    # dependency resolves to the runtime already in the process.  Loading ours first puts two
    # HIP runtimes side by side and feasthip_create then fails (observed: code 7).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib
