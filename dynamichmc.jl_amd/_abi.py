"""ctypes binding of the C ABI in include/dhmc.h (libdhmc_amd.so, built in-tree under lib/).

This is the whole Python<->HIP surface: plain pointers and sizes, no torch types.  There is no
CPU fallback — if the library is missing or no HIP device is present, calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DHMC_LIB_PATH") or os.path.join(PKG_DIR, "lib", "libdhmc_amd.so")   # override: A/B builds only

OK, ERR_INVALID_ARGUMENT, ERR_HIP, ERR_UNSUPPORTED, ERR_CHAIN_FAILURE, ERR_NO_DEVICE, ERR_CALLBACK = range(7)
ST_NONFINITE_POSITION, ST_INVALID_INITIAL, ST_STEPSIZE_SEARCH_FAILED, ST_NONFINITE_START_DENSITY = 1, 2, 4, 8
TARGET_STD_NORMAL, TARGET_DIAG_NORMAL, TARGET_TRIDIAG_NORMAL, TARGET_FUNNEL, TARGET_LOGISTIC, TARGET_ALWAYS_DIVERGENT, TARGET_DENSE_NORMAL, TARGET_EXTERNAL = range(8)
TARGET_USER_BASE = 1000     # + the handle of dhmc_register_target_source: a run-time compiled device functor
METRIC_DIAG, METRIC_DENSE = 0, 1

ERROR_NAMES = {ERR_INVALID_ARGUMENT: "invalid argument", ERR_HIP: "HIP runtime error",
               ERR_UNSUPPORTED: "unsupported configuration", ERR_CHAIN_FAILURE: "chain failure",
               ERR_NO_DEVICE: "no HIP device (there is no CPU fallback)",
               ERR_CALLBACK: "the external log-density callback is missing or failed"}


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("dim", C.c_int32), ("chains", C.c_int32),
                ("chain_offset", C.c_int32), ("metric", C.c_int32), ("target", C.c_int32),
                ("target_params", C.c_void_p), ("target_params_bytes", C.c_uint64),
                ("max_depth", C.c_int32), ("dense_per_chain", C.c_int32), ("min_delta", C.c_double),
                ("seed", C.c_uint64)]


class StepsizeSearch(C.Structure):
    _fields_ = [("initial_eps", C.c_double), ("log_threshold", C.c_double),
                ("maxiter_crossing", C.c_int32), ("reserved", C.c_int32)]


class DualAveragingABI(C.Structure):
    _fields_ = [("delta", C.c_double), ("gamma", C.c_double), ("kappa", C.c_double),
                ("t0", C.c_int32), ("init", C.c_int32), ("finalize", C.c_int32),
                ("reserved", C.c_int32)]


class Outputs(C.Structure):
    _fields_ = [("on_device", C.c_int32), ("reserved", C.c_int32), ("draws", C.c_void_p),
                ("logdensities", C.c_void_p), ("eps", C.c_void_p), ("pi", C.c_void_p),
                ("acceptance_rate", C.c_void_p), ("steps", C.c_void_p), ("term_left", C.c_void_p),
                ("term_right", C.c_void_p), ("depth", C.c_void_p), ("directions", C.c_void_p)]


class TreeStatisticsSummaryABI(C.Structure):
    _fields_ = [("n", C.c_int64), ("a_mean", C.c_double), ("a_quantiles", C.c_double * 5), ("max_depth", C.c_int64),
                ("divergence", C.c_int64), ("turning", C.c_int64), ("depth_counts", C.c_int64 * 33)]


# int fn(void* user, const double* q, int64 chains, int64 ld, int64 dim, double* lq, double* grad, void* stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
LOGDENSITY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p)

OUTPUT_FIELDS = [("draws", np.float64), ("logdensities", np.float64), ("eps", np.float64),
                 ("pi", np.float64), ("acceptance_rate", np.float64), ("steps", np.int64),
                 ("term_left", np.int64), ("term_right", np.int64), ("depth", np.int32),
                 ("directions", np.uint32)]

# every symbol include/dhmc.h declares
SYMBOLS = ["dhmc_create", "dhmc_destroy", "dhmc_set_stream", "dhmc_last_error", "dhmc_version", "dhmc_detmath_version",
           "dhmc_init", "dhmc_set_position", "dhmc_get_position", "dhmc_set_metric_diag", "dhmc_get_metric_diag",
           "dhmc_set_metric_dense", "dhmc_get_metric_dense", "dhmc_get_metric_dense_chain", "dhmc_set_stepsize", "dhmc_get_stepsize", "dhmc_get_status",
           "dhmc_find_initial_stepsize", "dhmc_run", "dhmc_update_metric_diag", "dhmc_update_metric_dense", "dhmc_state_bytes",
           "dhmc_export_state", "dhmc_import_state", "dhmc_last_run_kernel_ms",
           "dhmc_last_run_leapfrogs", "dhmc_last_run_rounds", "dhmc_workspace_bytes",
           "dhmc_leapfrog_trajectory", "dhmc_explore_log_acceptance_ratios", "dhmc_ess_rhat",
           "dhmc_set_logdensity_callback", "dhmc_ess_bulk", "dhmc_ess_tail", "dhmc_summarize_tree_statistics",
           "dhmc_set_dense_products", "dhmc_get_dense_products", "dhmc_host_alloc", "dhmc_host_free",
           "dhmc_register_target_source", "dhmc_check_target_source", "dhmc_target_source_log", "dhmc_detmath_selftest", "dhmc_set_metric_allreduce",
           "dhmc_metric_window_begin", "dhmc_metric_window_count", "dhmc_update_metric_diag_window", "dhmc_metric_window_end"]

_lib = None


class LibraryMissing(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  If this library pulled in the
    system copy first, a later `import torch` would bring a SECOND HIP runtime into the process, and that one sees
    no GPUs.  Loading torch's copy first (when torch is installed; torch itself is not imported) makes both sides
    resolve to one runtime whatever the import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libdhmc_amd.so from the package tree.  Never falls back to anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LibraryMissing(
                f"{LIB_PATH} not found: build it with `make -C dynamichmc.jl_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        for s in SYMBOLS:
            getattr(L, s)
        L.dhmc_last_error.restype = C.c_char_p
        L.dhmc_version.restype = C.c_char_p
        L.dhmc_target_source_log.restype = C.c_char_p
        L.dhmc_last_run_kernel_ms.restype = C.c_double
        L.dhmc_last_run_leapfrogs.restype = C.c_uint64
        L.dhmc_last_run_rounds.restype = C.c_uint64
        L.dhmc_workspace_bytes.restype = C.c_uint64
        L.dhmc_metric_window_count.restype = C.c_int64
        for name in SYMBOLS:
            if name.startswith("dhmc_") and name not in ("dhmc_last_error", "dhmc_version", "dhmc_target_source_log", "dhmc_last_run_kernel_ms",
                                                         "dhmc_last_run_leapfrogs", "dhmc_last_run_rounds", "dhmc_workspace_bytes", "dhmc_metric_window_count"):
                getattr(L, name).restype = C.c_int
        _lib = L
    return _lib
