"""ctypes declarations for libpvw_hip.so (include/pvw_hip.h).

The product has no CPU fallback: if the library is missing this module raises at
import of the symbols, and every device entry point returns PVW_ERR_INTERNAL
without a gfx950 device."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpvw_hip.so")

PVW_OK = 0
ERROR_NAMES = {
    1: "InvalidParameters", 2: "SamplingError", 3: "EncryptionError", 4: "DecryptionError",
    5: "KeyGenerationError", 6: "CrsError", 7: "SerializationError", 8: "DeserializationError",
    9: "EncodingError", 10: "DecodingError", 11: "ValidationError", 12: "ContextError",
    13: "PolynomialError", 14: "MatrixError", 15: "DimensionMismatch", 16: "IndexOutOfBounds",
    17: "InsufficientData", 18: "InvalidFormat", 19: "InternalError",
}
REPR_POWER, REPR_NTT = 0, 1
RND_SEED, RND_EXPLICIT = 0, 1
DOM_R, DOM_E1, DOM_E2, DOM_SK, DOM_EKEY, DOM_CRS, DOM_GAUSS, DOM_PK = range(8)
PREPARE_PACKED, PREPARE_MFMA = 1, 2


class pvw_params_t(C.Structure):
    _fields_ = [
        ("n", C.c_uint32), ("k", C.c_uint32), ("l", C.c_uint32), ("num_moduli", C.c_uint32),
        ("moduli", C.POINTER(C.c_uint64)), ("secret_variance", C.c_float),
        ("error_bound_1", C.c_uint64), ("error_bound_2", C.c_uint64), ("device", C.c_int32),
        ("party_lo", C.c_uint32), ("party_hi", C.c_uint32), ("c1_lo", C.c_uint32), ("c1_hi", C.c_uint32),
    ]


class pvw_randomness_t(C.Structure):
    _fields_ = [
        ("mode", C.c_uint32), ("seed", C.c_uint8 * 32),
        ("r", C.c_void_p), ("e1", C.c_void_p), ("e2", C.c_void_p),
    ]


_P = C.c_void_p
_SIGNATURES = {
    "pvw_last_error": [C.c_char_p, C.c_size_t],
    "pvw_device_available": [],
    "pvw_ctx_create": [C.POINTER(pvw_params_t), C.POINTER(_P)],
    "pvw_ctx_destroy": [_P],
    "pvw_ctx_get_roots": [_P, _P],
    "pvw_ctx_set_roots": [_P, _P],
    "pvw_ctx_delta": [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)],
    "pvw_ctx_delta_power_l_minus_1": [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)],
    "pvw_ctx_q_total": [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)],
    "pvw_ctx_gadget": [_P, _P, C.c_uint32],
    "pvw_ctx_verify_correctness_condition": [_P, C.POINTER(C.c_int32)],
    "pvw_suggest_error_bounds": [C.c_uint32, C.c_uint32, C.c_uint32, _P, C.c_uint32, C.c_float,
                                 C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)],
    "pvw_encode_scalar": [_P, C.c_int64, _P, C.c_uint32],
    "pvw_load_crs": [_P, _P, C.c_uint32],
    "pvw_load_crs_device": [_P, _P, C.c_uint32, _P],
    "pvw_crs_generate": [_P, _P],
    "pvw_get_crs": [_P, _P, C.c_uint32],
    "pvw_load_pk": [_P, C.c_uint32, C.c_uint32, _P, C.c_uint32],
    "pvw_load_pk_device": [_P, C.c_uint32, C.c_uint32, _P, C.c_uint32, _P],
    "pvw_pk_fill_uniform": [_P, _P],
    "pvw_get_pk": [_P, C.c_uint32, C.c_uint32, _P, C.c_uint32],
    "pvw_num_public_keys": [_P, C.POINTER(C.c_uint32)],
    "pvw_is_full": [_P, C.POINTER(C.c_int32)],
    "pvw_keygen": [_P, C.c_uint32, C.c_uint32, _P, _P, _P],
    "pvw_sample_secret_keys": [_P, _P, C.c_uint32, C.c_uint32, _P],
    "pvw_encrypt": [_P, _P, C.c_size_t, C.POINTER(pvw_randomness_t), _P, _P, C.c_uint32],
    "pvw_encrypt_device": [_P, _P, C.c_size_t, C.POINTER(pvw_randomness_t), _P, _P, C.c_uint32, _P],
    "pvw_encrypt_multi": [_P, _P, C.c_size_t, C.c_size_t, _P, _P, _P, C.c_uint32],
    "pvw_encrypt_multi_device": [_P, _P, C.c_size_t, C.c_size_t, _P, _P, _P, C.c_uint32, _P],
    "pvw_decrypt_batch": [_P, _P, _P, _P, C.c_size_t, C.c_uint32, _P, _P],
    "pvw_decrypt_noisy_device": [_P, _P, _P, _P, C.c_size_t, C.c_uint32, _P, _P],
    "pvw_decrypt_batch_device": [_P, _P, _P, _P, C.c_size_t, C.c_uint32, _P, _P, _P],
    "pvw_sk_load": [_P, _P, C.POINTER(C.c_void_p)],
    "pvw_sk_free": [_P],
    "pvw_decrypt_batch_device_sk": [_P, _P, _P, _P, C.c_size_t, C.c_uint32, _P, _P, _P],
    "pvw_decode": [_P, _P, C.c_size_t, _P],
    "pvw_decode_host": [_P, _P, C.c_size_t, _P],
    "pvw_decode_device": [_P, _P, C.c_size_t, _P, _P],
    "pvw_selftest_decode_fixed": [_P, _P, C.c_size_t, _P],
    "pvw_selftest_mfma_i8": [_P, _P, _P, _P],
    "pvw_selftest_secret_residue": [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "pvw_selftest_siphash": [_P, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)],
    "pvw_selftest_decode_tables": [_P, C.POINTER(C.c_uint32)],
    "pvw_selftest_decode_shortcuts": [_P, _P, C.c_size_t, _P, _P],
    "pvw_build_is_tuning": [],
    "pvw_crs_seed_from_tag": [C.c_char_p, _P],
    "pvw_ntt_forward": [_P, _P, C.c_size_t],
    "pvw_ntt_inverse": [_P, _P, C.c_size_t],
    "pvw_small_to_poly": [_P, _P, C.c_size_t, _P, C.c_uint32],
    "pvw_sample_cbd": [_P, _P, C.c_uint32, C.c_uint32, C.c_size_t, C.c_float, _P],
    "pvw_sample_uniform": [_P, _P, C.c_uint32, C.c_uint32, C.c_size_t, C.c_uint64, _P],
    "pvw_sample_gaussian": [_P, _P, C.c_uint32, C.c_size_t, C.c_uint64, _P],
    "pvw_ctx_set_profiling": [_P, C.c_int32],
    "pvw_ctx_kernel_time": [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)],
    "pvw_ctx_reset_profiling": [_P],
    "pvw_host_alloc": [C.c_size_t, C.POINTER(_P)],
    "pvw_host_free": [_P],
    "pvw_ctx_resident_bytes": [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "pvw_ctx_derived_bytes": [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "pvw_prepare": [_P, C.c_uint32, _P, C.POINTER(C.c_uint64)],
    "pvw_ctx_packed_active": [_P, C.POINTER(C.c_uint32)],
    "pvw_ctx_synchronize": [_P],
}

# include/pvw_hip_tuning.h: exported by the measurement build only
_TUNING_SIGNATURES = {
    "pvw_selftest_read_bandwidth": [_P, C.c_uint32, _P, _P],
    "pvw_tuning_read_stamps": [_P, _P, _P, C.c_uint32],
    "pvw_tuning_read_probe": [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P],
}

LIB_TUNING_PATH = os.path.join(HERE, "libpvw_hip_tuning.so")
_libs = {}
# which build `lib()` hands out: always the shipped library unless CODE asks for the measurement build with
# `select("tuning")` (tools/*.py, bench.py --tuning-library, tests/test_gpu_tuning.py).  Nothing in the environment can
# redirect the package to it -- that build honours switches that change what encrypt computes.  Every PvwParameters
# remembers the library it was created with.
_selected = "default"


def _load(which: str) -> C.CDLL:
    if which not in _libs:
        path = LIB_TUNING_PATH if which == "tuning" else LIB_PATH
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: the PVW hot path is HIP-only, run "
                "`python -c 'import __graft_entry__ as g; g.build()'` first")
        L = C.CDLL(path)
        sigs = dict(_SIGNATURES)
        if which == "tuning":
            sigs.update(_TUNING_SIGNATURES)
        for name, args in sigs.items():
            fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
            fn.argtypes = args
            fn.restype = C.c_int32
        _libs[which] = L
    return _libs[which]


def lib() -> C.CDLL:
    """The selected HIP library (libpvw_hip.so unless the tuning build was selected)."""
    return _load(_selected)


def tuning_lib() -> C.CDLL:
    """libpvw_hip_tuning.so (-DPVW_TUNING=1): schedule selectors, timing ablations, bandwidth probe."""
    return _load("tuning")


def select(which: str) -> str:
    """Choose the build later `lib()` calls return ("default" | "tuning"); returns the previous choice."""
    global _selected
    if which not in ("default", "tuning"):
        raise ValueError(which)
    prev, _selected = _selected, which
    return prev


def last_error(L: C.CDLL = None) -> str:
    buf = C.create_string_buffer(512)
    (L or lib()).pvw_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")
