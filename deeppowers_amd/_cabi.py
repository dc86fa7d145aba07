"""ctypes binding of libdpfhe_hip.so (include/dpfhe.h).  Fails LOUDLY when the HIP library is missing:
there is no CPU fallback on the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdpfhe_hip.so")
_U64P = C.c_void_p  # device pointers travel as integers

SYMBOLS = {
    "dpfhe_ctx_create": ([C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int], C.c_int),
    "dpfhe_ctx_destroy": ([C.c_void_p], C.c_int),
    "dpfhe_ctx_log2n": ([C.c_void_p], C.c_uint32),
    "dpfhe_ctx_limbs": ([C.c_void_p], C.c_uint32),
    "dpfhe_ctx_uses_fold": ([C.c_void_p], C.c_int),
    "dpfhe_ctx_limb_class": ([C.c_void_p, C.c_uint32], C.c_int),
    "dpfhe_ctx_release_scratch": ([C.c_void_p, C.c_void_p, C.c_uint32], C.c_int),
    "dpfhe_ctx_scratch_bytes": ([C.c_void_p], C.c_size_t),
    "dpfhe_tune_cache_clear": ([], None),
    "dpfhe_ctx_set_scratch_limit": ([C.c_void_p, C.c_size_t], C.c_int),
    "dpfhe_ctx_autotune": ([C.c_void_p, _U64P, C.c_size_t, C.c_uint32, C.c_void_p], C.c_int),
    "dpfhe_ctx_tune_info": ([C.c_void_p, C.c_void_p], C.c_int),
    "dpfhe_ctx_set_ct_mul_variant": ([C.c_void_p, C.c_int], C.c_int),
    "dpfhe_ct_mul_variant_name": ([C.c_int], C.c_char_p),
    "dpfhe_debug_ct_mul_trace": ([C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, _U64P, C.c_void_p], C.c_int),
    "dpfhe_ntt_fwd": ([C.c_void_p, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_ntt_inv": ([C.c_void_p, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_ntt_fwd_oop": ([C.c_void_p, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_ntt_inv_oop": ([C.c_void_p, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_dyadic_mul": ([C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_dyadic_mul_add": ([C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_add": ([C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_sub": ([C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_negate": ([C.c_void_p, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_multiply_plain": ([C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_ct_mul": ([C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_uint32, C.c_void_p], C.c_int),
    "dpfhe_relinearize": ([C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_relinearize_hybrid": ([C.c_void_p, _U64P, _U64P, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_switch_key_hybrid": ([C.c_void_p, _U64P, _U64P, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_rotate_hybrid_batch": ([C.c_void_p, _U64P, _U64P, C.c_size_t, C.POINTER(C.c_uint32), _U64P, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_rotate_hybrid_hoisted": ([C.c_void_p, _U64P, _U64P, C.c_size_t, C.POINTER(C.c_uint32), _U64P, _U64P, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_rotate_hybrid_grouped": ([C.c_void_p, _U64P, _U64P, C.POINTER(C.c_uint32), C.c_size_t, C.c_size_t, _U64P, _U64P, _U64P, C.c_void_p], C.c_int),
    "dpfhe_rotate_hoisted_qp": ([C.c_void_p, _U64P, _U64P, C.c_size_t, C.POINTER(C.c_uint32), _U64P, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_ntt_inv_galois": ([C.c_void_p, _U64P, _U64P, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_switch_key_qp": ([C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_rescale_bsgs": ([C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_base_extend": ([C.c_void_p, _U64P, C.c_size_t, _U64P, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_scale_round": ([C.c_void_p, _U64P, C.c_size_t, _U64P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_matvec_plain_multi": ([C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_rescale": ([C.c_void_p, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_apply_galois": ([C.c_void_p, _U64P, _U64P, C.c_size_t, C.c_uint32, C.c_void_p], C.c_int),
    "dpfhe_switch_key": ([C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_matvec_plain": ([C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_matvec_scalar": ([C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_reduce_sum": ([C.c_void_p, _U64P, _U64P, C.c_size_t, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_copy": ([C.c_void_p, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_comm_unique_id": ([C.POINTER(C.c_uint8)], C.c_int),
    "dpfhe_comm_create": ([C.POINTER(C.c_void_p), C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int], C.c_int),
    "dpfhe_comm_destroy": ([C.c_void_p], C.c_int),
    "dpfhe_comm_allgather": ([C.c_void_p, _U64P, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_comm_allreduce_sum": ([C.c_void_p, C.c_void_p, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_canonicalize_sum": ([C.c_void_p, _U64P, C.c_size_t, C.c_void_p], C.c_int),
    "dpfhe_strerror": ([C.c_int], C.c_char_p),
    "dpfhe_last_error": ([], C.c_char_p),
}

IN_NTT, OUT_NTT = 1, 2


class TuneInfo(C.Structure):
    """dpfhe_tune_info (include/dpfhe.h)"""
    _fields_ = [("chosen", C.c_int32), ("n_variants", C.c_int32), ("source", C.c_int32), ("probe_pairs", C.c_uint32),
                ("probe_reps", C.c_uint32), ("probe_us", C.c_float * 8)]


TUNE_SOURCES = ("default", "retired (1)", "dpfhe_ctx_autotune", "forced", "cached dpfhe_ctx_autotune of this shape")   # include/dpfhe.h DPFHE_TUNE_*


class DpfheError(RuntimeError):
    """Mirrors deeppowers::common::Exception{ErrorCode} (/root/reference/src/common/error.hpp:42-53)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[{code}] {message}")
        self.code = code


_lib = None


def load():
    """Loads the HIP library.  torch is imported first so that ONE HIP runtime (the one torch bundles,
    same soname libamdhip64.so.7) serves both torch's allocator/streams and our kernels."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the FHE hot path.")
    import torch  # noqa: F401  (loads libamdhip64 before our library resolves it)
    lib = C.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted from include/dpfhe.h
        fn.argtypes, fn.restype = argtypes, restype
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        lib = load()
        detail = lib.dpfhe_last_error().decode() or lib.dpfhe_strerror(rc).decode()
        raise DpfheError(rc, f"{what}: {detail}" if what else detail)
