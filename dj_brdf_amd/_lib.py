"""ctypes binding of libdjb_hip.so (include/djb_hip.h).

The library is built in-tree by ``dj_brdf_amd/csrc/Makefile`` (``__graft_entry__.build()``) into
``dj_brdf_amd/lib/libdjb_hip.so``.  There is no fallback: if the shared object is missing this
module raises, and if no GPU is present every compute entry point returns DJB_ERR_NO_DEVICE.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DJB_LIB_PATH: an alternative build of the same library (kernel-tuning experiments: tools/variants.sh)
LIB_PATH = os.environ.get("DJB_LIB_PATH") or os.path.join(_HERE, "lib", "libdjb_hip.so")

DJB_OK = 0
ABI_VERSION = 235          # include/djb_hip.h: DJB_HIP_VERSION (the major digit must match the loaded library)
STATUS_NAMES = {
    0: "DJB_OK", 1: "DJB_ERR_INVALID_ARGUMENT", 2: "DJB_ERR_OPEN_FAILED", 3: "DJB_ERR_BAD_HEADER",
    4: "DJB_ERR_READ_FAILED", 5: "DJB_ERR_NOT_IMPLEMENTED", 6: "DJB_ERR_HIP", 7: "DJB_ERR_NO_DEVICE",
    8: "DJB_ERR_UNKNOWN_MATERIAL", 9: "DJB_ERR_OUT_OF_MEMORY", 10: "DJB_ERR_INTERNAL",
}
MEM_DEVICE, MEM_HOST = 0, 1


class Vec3View(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("stride", C.c_int64)]


class Params(C.Structure):
    _fields_ = [("kind", C.c_int), ("v", C.c_float * 5)]


class ParamsResolved(C.Structure):
    _fields_ = [("n", C.c_float * 3), ("a1", C.c_float), ("a2", C.c_float), ("phi_a", C.c_float),
                ("ax", C.c_float), ("ay", C.c_float), ("rho", C.c_float),
                ("sqrt_one_minus_rho_sqr", C.c_float), ("tx_n", C.c_float), ("ty_n", C.c_float)]


class FresnelDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("a", C.c_float * 3), ("b", C.c_float * 3),
                ("points", C.c_void_p), ("npoints", C.c_int)]


class exc(RuntimeError):
    """Mirror of ``djb::exc`` (reference dj_brdf.h:54-59): carries the djb_error message."""

    def __init__(self, status: int, message: str):
        super().__init__(message.rstrip("\n") or STATUS_NAMES.get(status, str(status)))
        self.status = status
        self.status_name = STATUS_NAMES.get(status, str(status))


# every symbol include/djb_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "djb_last_error", "djb_version", "djb_device_count", "djb_ctx_create", "djb_ctx_create_on_stream", "djb_ctx_destroy",
    "djb_ctx_synchronize", "djb_ctx_set_stream", "djb_ctx_set_option", "djb_merl_guard_stats", "djb_merl_guard_attack", "djb_ctx_stream", "djb_timer_start", "djb_timer_stop_ms",
    "djb_brdf_create_beckmann", "djb_brdf_create_ggx", "djb_brdf_create_merl_from_file",
    "djb_brdf_create_merl_from_memory", "djb_brdf_create_utia_from_file",
    "djb_brdf_create_utia_from_memory", "djb_brdf_create_lambert", "djb_brdf_create_tabular",
    "djb_fit_merl_files", "djb_fit_merl_files_multi", "djb_merl_bin_keys_batch", "djb_selftest_guarded_math", "djb_selftest_fast_trig", "djb_selftest_model_fast", "djb_selftest_contract", "djb_selftest_contract_sample", "djb_contract_sample_attack", "djb_ctx_libm_matches_host", "djb_host_libm_mode", "djb_host_atan_log_kat", "djb_selftest_libm", "djb_selftest_trig_sweep", "djb_brdf_create_tabular_anisotropic", "djb_tabular_anisotropic_get", "djb_tabular_anisotropic_fit",
    "djb_eval_pp_batch", "djb_eval_lean_batch", "djb_sample_pp_batch", "djb_sample_lean_batch", "djb_lrep_op", "djb_params_to_lrep", "djb_lrep_to_params",
    "djb_brdf_create_sgd", "djb_brdf_create_abc", "djb_brdf_create_sgd_from_params",
    "djb_brdf_create_abc_from_params",
    "djb_brdf_destroy", "djb_brdf_kind", "djb_brdf_get_samples", "djb_brdf_get_shadow", "djb_brdf_set_shadow", "djb_brdf_set_fresnel", "djb_brdf_get_fresnel", "djb_eval_batch", "djb_evalp_batch",
    "djb_pdf_batch", "djb_eval_pdf_batch", "djb_sample_batch", "djb_sample_rng_batch",
    "djb_evalp_is_batch", "djb_io_to_hd_batch", "djb_hd_to_io_batch", "djb_merl_index_batch", "djb_query_batch",
    "djb_params_resolve", "djb_tabular_get", "djb_tabular_fit", "djb_fit_merl_batch", "djb_fit_brdf_batch",
    "djb_gen_directions", "djb_gen_uniforms", "djb_histogram_xy", "djb_helper",
    "djb_set_file_map_observer", "djb_brdf_create_user_microfacet", "djb_fit_query_dirs", "djb_fit_aniso_query_dirs", "djb_brdf_create_tabular_from_samples",
    "djb_brdf_create_tabular_anisotropic_from_samples",
]

_lib = None


def load() -> C.CDLL:
    """Load libdjb_hip.so (once).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C dj_brdf_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "dj_brdf_amd has no CPU/PyTorch fallback path.")
    try:  # share torch's HIP runtime (same soname libamdhip64.so.7) when torch is in the process
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional plumbing
        pass
    lib = C.CDLL(LIB_PATH)
    lib.djb_last_error.restype = C.c_char_p
    lib.djb_ctx_stream.restype = C.c_void_p
    lib.djb_ctx_stream.argtypes = [C.c_void_p]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("djb_last_error", "djb_ctx_stream"):
            fn.restype = C.c_int
    if lib.djb_version() // 100 != ABI_VERSION // 100:
        raise ImportError(f"{LIB_PATH} has ABI version {lib.djb_version()}, this binding was written against {ABI_VERSION} "
                          "(include/djb_hip.h: DJB_HIP_VERSION) -- rebuild the library")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != DJB_OK:
        raise exc(status, load().djb_last_error().decode(errors="replace"))
