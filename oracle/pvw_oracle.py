"""ctypes binding of oracle/libpvw_oracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY -- see pvw_oracle.c.  Arrays are numpy, API layout:
polynomial = [L][l] u64 (limb-major, parameters.rs:433-458), matrices row-major.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpvw_oracle.so")
_PLAIN_PATH = os.path.join(_HERE, "libpvw_oracle_plain.so")   # -DPVW_ORACLE_PLAIN_MOD: 128-bit remainder, no Barrett
_lib = None
_plain = None

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pvw_oracle.c")
    for path in (_LIB_PATH, _PLAIN_PATH):
        if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(path)], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib(plain: bool = False):
    """The C restatement; plain=True: the build whose every reduction is a 128-bit `%` (no Barrett step)."""
    global _lib, _plain
    if (_plain if plain else _lib) is None:
        build()
        L = C.CDLL(_PLAIN_PATH if plain else _LIB_PATH)
        L.pvwo_min_primitive_root.restype = C.c_uint64
        L.pvwo_min_primitive_root.argtypes = [C.c_uint64, C.c_uint32]
        L.pvwo_ctx_create.restype = C.c_void_p
        L.pvwo_ctx_create.argtypes = [u64p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.pvwo_ctx_destroy.argtypes = [C.c_void_p]
        L.pvwo_ntt_forward.argtypes = [C.c_void_p, u64p, C.c_size_t]
        L.pvwo_ntt_inverse.argtypes = [C.c_void_p, u64p, C.c_size_t]
        L.pvwo_small_to_ntt.argtypes = [C.c_void_p, i64p, C.c_size_t, u64p]
        L.pvwo_mac_rows.argtypes = [C.c_void_p, u64p, u64p, C.c_size_t, C.c_uint32, u64p, C.c_int]
        L.pvwo_encrypt.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u64p, u64p, u64p, u64p,
                                   i64p, i64p, i64p, u64p, u64p, C.c_int]
        L.pvwo_keygen.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u64p, i64p, i64p, u64p]
        L.pvwo_decrypt_noisy.argtypes = [C.c_void_p, C.c_uint32, i64p, u64p, u64p, C.c_size_t, u64p]
        L.pvwo_sample_cbd.restype = C.c_int
        L.pvwo_sample_cbd.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_size_t, C.c_uint32, C.c_float, i64p]
        L.pvwo_sample_uniform.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_size_t, C.c_uint32, C.c_uint64, i64p]
        L.pvwo_fill_uniform_residues.argtypes = [C.c_void_p, u8p, C.c_uint32, C.c_uint32, C.c_size_t, u64p]
        L.pvwo_num_threads.restype = C.c_int
        if plain:
            _plain = L
        else:
            _lib = L
    return _plain if plain else _lib


def _seed(seed: bytes) -> np.ndarray:
    assert len(seed) == 32
    return np.frombuffer(seed, dtype=np.uint8).copy()


class Oracle:
    """One (moduli, l) context of the C restatement."""

    def __init__(self, moduli: Sequence[int], l: int, psi: Optional[Sequence[int]] = None, plain: bool = False):
        self._L = lib(plain)
        self.moduli = np.asarray(list(moduli), dtype=np.uint64)
        self.L, self.l = len(self.moduli), l
        self.psi = (np.asarray(list(psi), dtype=np.uint64) if psi is not None else
                    np.asarray([self._L.pvwo_min_primitive_root(int(q), 2 * l) for q in self.moduli],
                               dtype=np.uint64))
        self._h = self._L.pvwo_ctx_create(self.moduli, self.psi.ctypes.data, self.L, l)
        if not self._h:
            raise ValueError("pvwo_ctx_create failed")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.pvwo_ctx_destroy(self._h)
            self._h = None

    @property
    def poly(self):
        return self.L * self.l

    def ntt_forward(self, polys: np.ndarray) -> np.ndarray:
        out = np.ascontiguousarray(polys, dtype=np.uint64).copy()
        self._L.pvwo_ntt_forward(self._h, out.reshape(-1), out.size // self.poly)
        return out

    def ntt_inverse(self, polys: np.ndarray) -> np.ndarray:
        out = np.ascontiguousarray(polys, dtype=np.uint64).copy()
        self._L.pvwo_ntt_inverse(self._h, out.reshape(-1), out.size // self.poly)
        return out

    def small_to_ntt(self, coeffs: np.ndarray) -> np.ndarray:
        coeffs = np.ascontiguousarray(coeffs, dtype=np.int64)
        count = coeffs.size // self.l
        out = np.empty(coeffs.shape[:-1] + (self.L, self.l), dtype=np.uint64)
        self._L.pvwo_small_to_ntt(self._h, coeffs.reshape(-1), count, out.reshape(-1))
        return out

    def mac_rows(self, M: np.ndarray, v: np.ndarray, parallel: bool = True) -> np.ndarray:
        rows, k = M.shape[0], M.shape[1]
        out = np.empty((rows, self.L, self.l), dtype=np.uint64)
        self._L.pvwo_mac_rows(self._h, np.ascontiguousarray(M).reshape(-1), np.ascontiguousarray(v).reshape(-1),
                            rows, k, out.reshape(-1), int(parallel))
        return out

    def encrypt(self, a_hat, b_hat, g_hat, scalars, r, e1, e2, serial_c1: bool = False):
        n, k = b_hat.shape[0], b_hat.shape[1]
        c1 = np.empty((k, self.L, self.l), dtype=np.uint64)
        c2 = np.empty((n, self.L, self.l), dtype=np.uint64)
        self._L.pvwo_encrypt(self._h, n, k, np.ascontiguousarray(a_hat).reshape(-1),
                           np.ascontiguousarray(b_hat).reshape(-1),
                           np.ascontiguousarray(g_hat, dtype=np.uint64).reshape(-1),
                           np.ascontiguousarray(scalars, dtype=np.uint64),
                           np.ascontiguousarray(r, dtype=np.int64).reshape(-1),
                           np.ascontiguousarray(e1, dtype=np.int64).reshape(-1),
                           np.ascontiguousarray(e2, dtype=np.int64).reshape(-1),
                           c1.reshape(-1), c2.reshape(-1), int(serial_c1))
        return c1, c2

    def keygen(self, a_hat, sk, ek):
        n, k = sk.shape[0], sk.shape[1]
        b_hat = np.empty((n, k, self.L, self.l), dtype=np.uint64)
        self._L.pvwo_keygen(self._h, n, k, np.ascontiguousarray(a_hat).reshape(-1),
                          np.ascontiguousarray(sk, dtype=np.int64).reshape(-1),
                          np.ascontiguousarray(ek, dtype=np.int64).reshape(-1), b_hat.reshape(-1))
        return b_hat

    def decrypt_noisy(self, sk, c1s, c2col):
        D, k = c1s.shape[0], c1s.shape[1]
        noisy = np.empty((D, self.L, self.l), dtype=np.uint64)
        self._L.pvwo_decrypt_noisy(self._h, k, np.ascontiguousarray(sk, dtype=np.int64).reshape(-1),
                                 np.ascontiguousarray(c1s).reshape(-1),
                                 np.ascontiguousarray(c2col).reshape(-1), D, noisy.reshape(-1))
        return noisy

    def fill_uniform(self, seed: bytes, domain: int, index0: int, count: int) -> np.ndarray:
        out = np.empty((count, self.L, self.l), dtype=np.uint64)
        self._L.pvwo_fill_uniform_residues(self._h, _seed(seed), domain, index0, count, out.reshape(-1))
        return out


def sample_cbd(seed: bytes, domain: int, index0: int, count: int, l: int, variance: float) -> np.ndarray:
    out = np.empty((count, l), dtype=np.int64)
    rc = lib().pvwo_sample_cbd(_seed(seed), domain, index0, count, l, variance, out.reshape(-1))
    if rc:
        raise ValueError("The variance should be between 0.5 and 16")
    return out


def sample_uniform(seed: bytes, domain: int, index0: int, count: int, l: int, bound: int) -> np.ndarray:
    out = np.empty((count, l), dtype=np.int64)
    lib().pvwo_sample_uniform(_seed(seed), domain, index0, count, l, bound, out.reshape(-1))
    return out


def num_threads() -> int:
    return lib().pvwo_num_threads()
