"""Host-side mirror of `pvw::{params, crs, keys, crypto}` over the C ABI (include/pvw_hip.h).

Same names, argument meaning and error behaviour as the reference's Rust API so that the
parity tests read like the reference's own tests (tests/crypto.rs, tests/params.rs,
tests/keys.rs).  Everything heavy happens in libpvw_hip.so on the GPU; this file only
marshals numpy arrays.  Differences forced by the boundary:

  * randomness is an explicit input (a 32-byte seed or explicit small polynomials): the
    reference draws from thread_rng() (src/crypto/encryption.rs:138,164,180);
  * one PvwParameters object owns one device context, which holds at most one CRS and one
    GlobalPublicKey (the device-resident A-hat / B-hat);
  * polynomials are numpy arrays [L][l] uint64 (power basis unless stated otherwise).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from ._ffi import REPR_NTT, REPR_POWER


class PvwError(Exception):
    """PvwError (src/errors.rs:13-70): `.variant` is the Rust variant name."""

    def __init__(self, code: int, message: str):
        self.code = code
        self.variant = _ffi.ERROR_NAMES.get(code, f"Unknown({code})")
        super().__init__(f"{self.variant}: {message}")


def _check(rc: int, lib=None) -> None:
    if rc != _ffi.PVW_OK:
        raise PvwError(rc, _ffi.last_error(lib))


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _i64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int64)


def _seed(seed: bytes) -> np.ndarray:
    if len(seed) != 32:
        raise PvwError(1, "seed must be 32 bytes")
    return np.frombuffer(bytes(seed), dtype=np.uint8).copy()


def device_available() -> bool:
    return bool(_ffi.lib().pvw_device_available())


# ------------------------------------------------------------------------------------
# params (src/params/parameters.rs)
# ------------------------------------------------------------------------------------
class PvwParametersBuilder:
    """parameters.rs:44-201."""

    def __init__(self):
        self._n = self._k = self._l = self._moduli = None
        self._variance = self._b1 = self._b2 = None
        self._device = -1
        self._shard = (0, 0, 0, 0)

    @staticmethod
    def new() -> "PvwParametersBuilder":
        return PvwParametersBuilder()

    def set_parties(self, n): self._n = n; return self
    def set_dimension(self, k): self._k = k; return self
    def set_l(self, l): self._l = l; return self
    def set_moduli(self, moduli): self._moduli = list(moduli); return self
    def set_secret_variance(self, v): self._variance = float(v); return self
    def set_error_bound_1(self, b): self._b1 = int(b); return self
    def set_error_bound_2(self, b): self._b2 = int(b); return self
    def set_error_bounds(self, b1, b2): self._b1, self._b2 = int(b1), int(b2); return self
    def set_error_bounds_u32(self, b1, b2): return self.set_error_bounds(b1, b2)

    # not in the reference: device placement and the party shard of a multi-GPU job
    def set_device(self, ordinal): self._device = int(ordinal); return self

    def set_shard(self, party_lo, party_hi, c1_lo, c1_hi):
        self._shard = (party_lo, party_hi, c1_lo, c1_hi)
        return self

    def build(self) -> "PvwParameters":
        for name, v in (("n", self._n), ("k", self._k), ("l", self._l), ("moduli", self._moduli)):
            if v is None:
                raise PvwError(1, f"{name} not set")                      # parameters.rs:118-129
        b1 = 100 if self._b1 is None else self._b1                        # :167
        b2 = 200 if self._b2 is None else self._b2                        # :168
        if b1 <= 0:
            raise PvwError(1, "error_bound_1 must be positive")          # :172
        if b2 <= 0:
            raise PvwError(1, "error_bound_2 must be positive")          # :177
        if b1 >= 1 << 62 or b2 >= 1 << 62:
            raise PvwError(1, "error bounds must be below 2^62")
        variance = 0.5 if self._variance is None else self._variance     # :166
        return PvwParameters(self._n, self._k, self._l, self._moduli, variance, b1, b2,
                             self._device, self._shard)

    build_arc = build


class PvwParameters:
    """PvwParameters (parameters.rs:19-40) + the device context behind it."""

    def __init__(self, n, k, l, moduli, secret_variance, error_bound_1, error_bound_2,
                 device=-1, shard=(0, 0, 0, 0)):
        for name, v in (("n", n), ("k", k), ("l", l)):
            if not (0 <= int(v) < 1 << 32):
                raise PvwError(1, f"{name} out of range")
        self.n, self.k, self.l = int(n), int(k), int(l)
        self._moduli = _u64(list(moduli))
        self.secret_variance = float(secret_variance)
        self.error_bound_1, self.error_bound_2 = int(error_bound_1), int(error_bound_2)
        p = _ffi.pvw_params_t()
        p.n, p.k, p.l, p.num_moduli = self.n, self.k, self.l, len(self._moduli)
        p.moduli = self._moduli.ctypes.data_as(C.POINTER(C.c_uint64))
        p.secret_variance = self.secret_variance
        p.error_bound_1, p.error_bound_2 = self.error_bound_1, self.error_bound_2
        p.device = device
        p.party_lo, p.party_hi, p.c1_lo, p.c1_hi = shard
        h = C.c_void_p()
        self._h = None
        self._lib = _ffi.lib()              # the build selected now serves this context for its whole life
        _check(self._lib.pvw_ctx_create(C.byref(p), C.byref(h)), self._lib)
        self._h = h
        self.t = (self.n - 1) // 2                                        # :169
        self.party_lo, self.party_hi = (shard[0], shard[1]) if (shard[0] or shard[1]) else (0, self.n)
        self.c1_lo, self.c1_hi = (shard[2], shard[3]) if (shard[2] or shard[3]) else (0, self.k)

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                self._lib.pvw_ctx_destroy(self._h)
            except Exception:
                pass
            self._h = None

    @staticmethod
    def builder() -> PvwParametersBuilder:
        return PvwParametersBuilder()

    @staticmethod
    def new_with_u32_bounds(n, k, l, moduli, variance, b1, b2) -> "PvwParameters":   # :231-249
        return (PvwParametersBuilder().set_parties(n).set_dimension(k).set_l(l).set_moduli(moduli)
                .set_secret_variance(variance).set_error_bounds_u32(b1, b2).build())

    # -- accessors -----------------------------------------------------------------
    @property
    def L(self) -> int:
        return len(self._moduli)

    def moduli(self) -> List[int]:
        return [int(q) for q in self._moduli]

    def _call(self, name: str, *args) -> None:
        """one C-ABI call on this context, through the library the context was created with"""
        rc = getattr(self._lib, name)(self._h, *args)
        if rc != _ffi.PVW_OK:
            raise PvwError(rc, _ffi.last_error(self._lib))

    def _big(self, name) -> int:
        n = C.c_size_t()
        self._call(name, None, 0, C.byref(n))
        w = np.zeros(max(n.value, 1), dtype=np.uint64)
        self._call(name, _ptr(w), len(w), C.byref(n))
        return sum(int(w[i]) << (64 * i) for i in range(n.value))

    def delta(self) -> int:
        return self._big("pvw_ctx_delta")

    def delta_power_l_minus_1(self) -> int:
        return self._big("pvw_ctx_delta_power_l_minus_1")

    def q_total(self) -> int:
        return self._big("pvw_ctx_q_total")

    def roots(self) -> List[int]:
        w = np.zeros(self.L, dtype=np.uint64)
        self._call("pvw_ctx_get_roots", _ptr(w))
        return [int(x) for x in w]

    def set_roots(self, psi: Sequence[int]) -> None:
        w = _u64(list(psi))
        if len(w) != self.L:
            raise PvwError(15, f"expected {self.L}, got {len(w)}")
        self._call("pvw_ctx_set_roots", _ptr(w))

    def gadget_polynomial(self, repr: int = REPR_POWER) -> np.ndarray:   # :288-308
        out = np.zeros((self.L, self.l), dtype=np.uint64)
        self._call("pvw_ctx_gadget", _ptr(out), repr)
        return out

    def gadget_vector(self) -> List[int]:                                 # :311-324
        d = self.delta()
        return [d ** j for j in range(self.l)]

    def encode_scalar(self, scalar: int, repr: int = REPR_POWER) -> np.ndarray:   # :346-367
        out = np.zeros((self.L, self.l), dtype=np.uint64)
        self._call("pvw_encode_scalar", C.c_int64(scalar), _ptr(out), repr)
        return out

    def verify_correctness_condition(self) -> bool:                       # :510-551
        ok = C.c_int32()
        self._call("pvw_ctx_verify_correctness_condition", C.byref(ok))
        return bool(ok.value)

    @staticmethod
    def suggest_error_bounds(n, k, l, moduli, variance) -> Tuple[int, int]:   # :554-603
        m = _u64(list(moduli))
        b1, b2 = C.c_uint32(), C.c_uint32()
        _check(_ffi.lib().pvw_suggest_error_bounds(n, k, l, _ptr(m), len(m), variance,
                                                   C.byref(b1), C.byref(b2)))
        return b1.value, b2.value

    # -- ring primitives (fhe-math call sites) ----------------------------------------
    def bigints_to_poly(self, bigints: Sequence[int]) -> np.ndarray:     # :420-474 (host marshalling)
        if len(bigints) != self.l:
            raise PvwError(1, f"Expected {self.l} coefficients, got {len(bigints)}")
        return np.array([[int(c) % int(q) for c in bigints] for q in self._moduli], dtype=np.uint64)

    def poly_to_bigints(self, poly: np.ndarray) -> List[int]:
        """Vec<BigUint>::from(&Poly): CRT lift to [0, Q)."""
        Q = self.q_total()
        out = [0] * self.l
        for i, q in enumerate(self.moduli()):
            Qi = Q // q
            inv = pow(Qi, -1, q)
            for c in range(self.l):
                out[c] = (out[c] + int(poly[i, c]) * inv % q * Qi) % Q
        return out

    def from_coefficients(self, coeffs, repr: int = REPR_NTT) -> np.ndarray:
        """Poly::from_coefficients(&[i64]) (+ change_representation(Ntt)) on the device."""
        a = _i64(coeffs)
        count = a.size // self.l
        out = np.zeros(a.shape[:-1] + (self.L, self.l), dtype=np.uint64)
        self._call("pvw_small_to_poly", _ptr(a), count, _ptr(out), repr)
        return out

    def ntt_forward(self, polys) -> np.ndarray:
        a = _u64(polys).copy()
        self._call("pvw_ntt_forward", _ptr(a), a.size // (self.L * self.l))
        return a

    def ntt_inverse(self, polys) -> np.ndarray:
        a = _u64(polys).copy()
        self._call("pvw_ntt_inverse", _ptr(a), a.size // (self.L * self.l))
        return a

    # -- samplers (src/sampling) --------------------------------------------------------
    def sample_vec_cbd(self, seed: bytes, domain: int, index0: int, count: int, variance=None) -> np.ndarray:
        out = np.zeros((count, self.l), dtype=np.int64)
        v = self.secret_variance if variance is None else variance
        self._call("pvw_sample_cbd", _ptr(_seed(seed)), domain, index0, count, v, _ptr(out))
        return out

    def sample_uniform_coefficients(self, seed: bytes, domain: int, index0: int, count: int, bound: int) -> np.ndarray:
        out = np.zeros((count, self.l), dtype=np.int64)
        self._call("pvw_sample_uniform", _ptr(_seed(seed)), domain, index0, count, bound, _ptr(out))
        return out

    def sample_discrete_gaussian_vec(self, seed: bytes, bound: int, n: int, index0: int = 0) -> np.ndarray:
        out = np.zeros(n, dtype=np.int64)
        self._call("pvw_sample_gaussian", _ptr(_seed(seed)), index0, n, bound, _ptr(out))
        return out

    # -- measurement ----------------------------------------------------------------------
    def set_profiling(self, on: bool) -> None:
        self._call("pvw_ctx_set_profiling", int(on))

    def reset_profiling(self) -> None:
        self._call("pvw_ctx_reset_profiling")

    def kernel_time(self, name: str) -> Tuple[float, int]:
        ms, cnt = C.c_double(), C.c_uint64()
        self._call("pvw_ctx_kernel_time", name.encode(), C.byref(ms), C.byref(cnt))
        return ms.value, cnt.value

    def resident_bytes(self) -> Tuple[int, int]:
        a, b = C.c_uint64(), C.c_uint64()
        self._call("pvw_ctx_resident_bytes", C.byref(a), C.byref(b))
        return a.value, b.value

    def prepare(self, flags: int = _ffi.PREPARE_PACKED | _ffi.PREPARE_MFMA, stream=None) -> int:
        """pvw_prepare: build the derived copies of the resident matrices (and `stream`'s workspace) now; returns the
        bytes allocated for them.  After it, *_device calls on that stream neither allocate nor synchronise."""
        taken = C.c_uint64(0)
        self._call("pvw_prepare", int(flags), C.c_void_p(stream) if stream else None, C.byref(taken))
        return int(taken.value)

    def packed_active(self) -> int:
        """bits per residue of the packed stream single-dealer encrypt would use right now (0 = the tiled matrices)"""
        w = C.c_uint32(0)
        self._call("pvw_ctx_packed_active", C.byref(w))
        return int(w.value)

    def derived_bytes(self) -> Tuple[int, int]:
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._call("pvw_ctx_derived_bytes", C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def synchronize(self) -> None:
        self._call("pvw_ctx_synchronize")


# ------------------------------------------------------------------------------------
# CRS (src/params/crs.rs)
# ------------------------------------------------------------------------------------
class PvwCrs:
    """PvwCrs (crs.rs:12-17): the k x k matrix lives on the device of `params`."""

    def __init__(self, params: PvwParameters):
        self.params = params

    @staticmethod
    def new(params: PvwParameters, rng=None) -> "PvwCrs":                    # crs.rs:24-39
        """Fresh random CRS: `rng` is anything with `.bytes(32)` / `.randbytes(32)` (default: the OS entropy
        source, the reference takes a CryptoRng); the polynomials come from this library's seeded streams."""
        if rng is None:
            import os
            seed = os.urandom(32)
        else:
            seed = bytes(rng.bytes(32) if hasattr(rng, "bytes") else rng.randbytes(32))
        return PvwCrs.new_deterministic(params, seed)

    @staticmethod
    def seed_from_tag(tag: str) -> bytes:
        """The 32-byte seed PvwCrs::new_from_tag derives (crs.rs:75-87): DefaultHasher(tag + "CRS") as 8
        little-endian bytes, repeated four times."""
        out = np.zeros(32, dtype=np.uint8)
        _check(_ffi.lib().pvw_crs_seed_from_tag(tag.encode("utf-8"), _ptr(out)))
        return out.tobytes()

    @staticmethod
    def new_from_tag(params: PvwParameters, tag: str) -> "PvwCrs":           # crs.rs:74-90
        return PvwCrs.new_deterministic(params, PvwCrs.seed_from_tag(tag))

    @staticmethod
    def new_deterministic(params: PvwParameters, seed: bytes) -> "PvwCrs":   # crs.rs:45-67
        params._call("pvw_crs_generate", _ptr(_seed(seed)))
        return PvwCrs(params)

    @staticmethod
    def from_polynomials(params: PvwParameters, a: np.ndarray, repr: int = REPR_POWER) -> "PvwCrs":
        a = _u64(a)
        want = (params.k, params.k, params.L, params.l)
        if a.shape != want:
            raise PvwError(15, f"expected {want}, got {a.shape}")
        params._call("pvw_load_crs", _ptr(a), repr)
        return PvwCrs(params)

    def dimensions(self) -> Tuple[int, int]:
        return self.params.k, self.params.k

    def matrix(self, repr: int = REPR_POWER) -> np.ndarray:
        p = self.params
        out = np.zeros((p.k, p.k, p.L, p.l), dtype=np.uint64)
        p._call("pvw_get_crs", _ptr(out), repr)
        return out

    def get(self, i: int, j: int, repr: int = REPR_POWER) -> Optional[np.ndarray]:   # crs.rs:93
        if not (0 <= i < self.params.k and 0 <= j < self.params.k):
            return None
        return self.matrix(repr)[i, j]

    def validate(self) -> None:
        return None


# ------------------------------------------------------------------------------------
# keys (src/keys)
# ------------------------------------------------------------------------------------
class SecretKey:
    """SecretKey (secret_key.rs:14-18): k x l CBD coefficients."""

    def __init__(self, params: PvwParameters, secret_coeffs: np.ndarray):
        self.params = params
        self.secret_coeffs = np.array(secret_coeffs, dtype=np.int64, order="C", copy=True)   # owned: zeroized on drop

    @staticmethod
    def random(params: PvwParameters, seed: bytes, party_index: int = 0) -> "SecretKey":   # :45-63
        out = np.zeros((params.k, params.l), dtype=np.int64)
        params._call("pvw_sample_secret_keys", _ptr(_seed(seed)), party_index, 1, _ptr(out))
        key = SecretKey(params, out)
        out.fill(0)
        return key

    @staticmethod
    def from_coefficients(params: PvwParameters, coeffs) -> "SecretKey":
        a = _i64(coeffs)
        if a.shape != (params.k, params.l):
            raise PvwError(1, f"Secret key has shape {a.shape} but expected {(params.k, params.l)}")
        return SecretKey(params, a)

    def coefficients(self) -> np.ndarray:
        """A COPY of the coefficients (secret_key.rs:280-292 hands out a slice tied to the key's lifetime; numpy cannot
        express that, and a view would silently turn to zeros when the key is dropped).  The caller owns the copy and
        wipes it.  `secret_coeffs` is the key's own storage: cleared by zeroize() / on drop."""
        return self.secret_coeffs.copy()

    def zeroize(self) -> None:
        """Zeroize / ZeroizeOnDrop (secret_key.rs:20-30): overwrite the coefficients in place (the device side
        clears its own copies before every call returns, see pvw_selftest_secret_residue)."""
        if getattr(self, "secret_coeffs", None) is not None and self.secret_coeffs.flags.writeable:
            self.secret_coeffs.fill(0)

    def __del__(self):
        try:
            self.zeroize()
        except Exception:
            pass

    def load_device(self) -> "DeviceSecretKey":
        """The key in the form the inner products of decrypt read (NTT(sk[j]), secret_key.rs:98-112), resident on the
        device until the returned handle is freed: pvw_decrypt_batch_device_sk then neither transforms nor wipes per call."""
        return DeviceSecretKey(self)

    def get_polynomial(self, index: int) -> np.ndarray:                   # :98-112 (NTT form)
        if not 0 <= index < len(self.secret_coeffs):
            raise PvwError(1, f"Index {index} out of bounds for {len(self.secret_coeffs)} polynomials")
        return self.params.from_coefficients(self.secret_coeffs[index], REPR_NTT)

    def __len__(self):
        return len(self.secret_coeffs)


class Party:
    """Party (public_key.rs:17-22)."""

    def __init__(self, index: int, secret_key: SecretKey):
        self.index, self.secret_key = index, secret_key

    @staticmethod
    def new(index: int, params: PvwParameters, seed: bytes) -> "Party":   # :62-79
        if index >= params.n:
            raise PvwError(1, f"Party index {index} exceeds maximum {params.n - 1}")
        return Party(index, SecretKey.random(params, seed, index))


class GlobalPublicKey:
    """GlobalPublicKey (public_key.rs:43-54): the n x k matrix B lives on the device."""

    def __init__(self, crs: PvwCrs):
        self.crs = crs
        self.params = crs.params

    @staticmethod
    def new(crs: PvwCrs) -> "GlobalPublicKey":
        return GlobalPublicKey(crs)

    def add_public_key(self, index: int, key_polynomials: np.ndarray, repr: int = REPR_POWER) -> None:   # :214-250
        p = self.params
        b = _u64(key_polynomials)
        if index >= p.n:
            raise PvwError(1, f"Party index {index} exceeds maximum {p.n - 1}")
        if b.shape != (p.k, p.L, p.l):
            raise PvwError(1, f"Public key dimension {b.shape[0]} doesn't match parameter k={p.k}")
        p._call("pvw_load_pk", index, index + 1, _ptr(b), repr)

    def load_rows(self, party_lo: int, rows: np.ndarray, repr: int = REPR_POWER) -> None:
        p = self.params
        b = _u64(rows)
        p._call("pvw_load_pk", party_lo, party_lo + b.shape[0], _ptr(b), repr)

    def generate_and_add_party(self, party: Party, seed: bytes) -> None:   # :256-263
        self._keygen(party.index, party.index + 1, party.secret_key.secret_coeffs[None], None, seed)

    def generate_all_party_keys(self, parties: Sequence[Party], seed: bytes) -> None:   # :376-401
        if len(parties) > self.params.n:
            raise PvwError(1, f"Too many parties: {len(parties)} > {self.params.n}")
        # the reference generates the keys in parallel and adds them in order (:387-399); here every run of
        # consecutive party indices is ONE batched device call (the matrix-core path from 8 parties up)
        i = 0
        while i < len(parties):
            j = i + 1
            while j < len(parties) and parties[j].index == parties[j - 1].index + 1:
                j += 1
            sk = np.stack([pt.secret_key.secret_coeffs for pt in parties[i:j]])
            try:
                self._keygen(parties[i].index, parties[i].index + (j - i), sk, None, seed)
            finally:
                sk.fill(0)                                 # the stacked copy does not outlive the call, whatever happened
            i = j

    def generate_with_errors(self, party_lo: int, sk: np.ndarray, ek: np.ndarray) -> None:
        """b_i = s_i*A + e_i with explicit key errors (public_key.rs:111-147)."""
        self._keygen(party_lo, party_lo + len(sk), sk, ek, None)

    def _keygen(self, lo, hi, sk, ek, seed):
        p = self.params
        sk = _i64(sk)
        ekp = None if ek is None else _i64(ek)
        sd = None if seed is None else _seed(seed)
        p._call("pvw_keygen", lo, hi, _ptr(sk), _ptr(ekp), _ptr(sd))

    def fill_uniform(self, seed: bytes) -> None:
        self.params._call("pvw_pk_fill_uniform", _ptr(_seed(seed)))

    def matrix(self, party_lo: int = 0, party_hi: Optional[int] = None, repr: int = REPR_POWER) -> np.ndarray:
        p = self.params
        hi = p.n if party_hi is None else party_hi
        out = np.zeros((hi - party_lo, p.k, p.L, p.l), dtype=np.uint64)
        p._call("pvw_get_pk", party_lo, hi, _ptr(out), repr)
        return out

    def get_polynomial(self, i: int, j: int, repr: int = REPR_POWER) -> Optional[np.ndarray]:   # :334-336
        p = self.params
        if not (0 <= i < p.n and 0 <= j < p.k):
            return None
        return self.matrix(i, i + 1, repr)[0, j]

    def dimensions(self) -> Tuple[int, int]:
        return self.params.n, self.params.k

    def num_public_keys(self) -> int:                                      # :344
        out = C.c_uint32()
        self.params._call("pvw_num_public_keys", C.byref(out))
        return out.value

    def is_full(self) -> bool:                                             # :349
        out = C.c_int32()
        self.params._call("pvw_is_full", C.byref(out))
        return bool(out.value)


# ------------------------------------------------------------------------------------
# crypto (src/crypto)
# ------------------------------------------------------------------------------------
class PvwCiphertext:
    """PvwCiphertext (encryption.rs:15-24): c1 [k][L][l], c2 [n][L][l] in `repr`."""

    def __init__(self, c1: np.ndarray, c2: np.ndarray, params: PvwParameters, repr: int):
        self.c1, self.c2, self.params, self.repr = c1, c2, params, repr

    def __len__(self):
        return len(self.c2)

    def is_empty(self) -> bool:
        return len(self.c1) == 0 and len(self.c2) == 0

    def validate(self) -> None:                                            # :41-76
        p = self.params
        if len(self.c1) != p.k:
            raise PvwError(1, f"c1 has {len(self.c1)} components but should have k={p.k}")
        if len(self.c2) != p.n:
            raise PvwError(1, f"c2 has {len(self.c2)} components but should have n={p.n}")

    def get_party_ciphertext(self, party_index: int) -> Optional[np.ndarray]:   # :82-84
        return self.c2[party_index] if 0 <= party_index < len(self.c2) else None

    def c1_components(self):
        return self.c1

    def c2_components(self):
        return self.c2


def _randomness(params: PvwParameters, seed, r, e1, e2):
    rnd = _ffi.pvw_randomness_t()
    keep = []
    if r is not None or e1 is not None or e2 is not None:
        if r is None or e1 is None or e2 is None:
            raise PvwError(1, "explicit randomness needs r, e1 and e2")
        r, e1, e2 = _i64(r), _i64(e1), _i64(e2)
        if r.shape != (params.k, params.l) or e1.shape != (params.k, params.l) or e2.shape != (params.n, params.l):
            raise PvwError(15, "explicit randomness has the wrong shape")
        rnd.mode = _ffi.RND_EXPLICIT
        rnd.r, rnd.e1, rnd.e2 = r.ctypes.data, e1.ctypes.data, e2.ctypes.data
        keep = [r, e1, e2]
    else:
        if seed is None:
            raise PvwError(1, "encrypt needs a 32-byte seed or explicit randomness")
        rnd.mode = _ffi.RND_SEED
        sd = _seed(seed)
        C.memmove(rnd.seed, sd.ctypes.data, 32)
    return rnd, keep


def encrypt(scalars: Sequence[int], global_pk: GlobalPublicKey, seed: Optional[bytes] = None, *,
            r=None, e1=None, e2=None, repr: int = REPR_NTT) -> PvwCiphertext:
    """encrypt (encryption.rs:105-214)."""
    p = global_pk.params
    sc = np.array([int(s) & 0xFFFFFFFFFFFFFFFF for s in scalars], dtype=np.uint64)
    rnd, keep = _randomness(p, seed, r, e1, e2)
    c1 = np.zeros((p.k, p.L, p.l), dtype=np.uint64)
    c2 = np.zeros((p.n, p.L, p.l), dtype=np.uint64)
    p._call("pvw_encrypt", _ptr(sc), len(sc), C.byref(rnd), _ptr(c1), _ptr(c2), repr)
    del keep
    ct = PvwCiphertext(c1, c2, p, repr)
    ct.validate()                                                          # :204-211
    return ct


def encrypt_party_shares(party_shares: Sequence[int], party_index: int, global_pk: GlobalPublicKey,
                         seed: Optional[bytes] = None, **kw) -> PvwCiphertext:
    """encryption.rs:221-245."""
    n = global_pk.params.n
    if party_index >= n:
        raise PvwError(1, f"Party index {party_index} exceeds maximum {n - 1}")
    if len(party_shares) != n:
        raise PvwError(1, f"Party must provide {n} shares, got {len(party_shares)}")
    return encrypt(party_shares, global_pk, seed, **kw)


def _dealer_seed(seed: bytes, dealer: int) -> bytes:
    """Per-dealer seed for encrypt_all_party_shares: the base seed with the dealer index
    folded into its last four bytes (the reference gives every dealer a fresh thread_rng)."""
    s = bytearray(seed)
    for i in range(4):
        s[28 + i] ^= (dealer >> (8 * i)) & 0xFF
    return bytes(s)


def encrypt_all_party_shares(all_shares: Sequence[Sequence[int]], global_pk: GlobalPublicKey,
                             seed: bytes, repr: int = REPR_NTT) -> List[PvwCiphertext]:
    """encrypt_all_party_shares (encryption.rs:253-286): one batched device call; dealers share
    passes over the public key four at a time (pvw_encrypt_multi).  Dealer d uses
    `_dealer_seed(seed, d)`, so the result equals d separate `encrypt_party_shares` calls."""
    p = global_pk.params
    n = p.n
    if len(all_shares) != n:
        raise PvwError(1, f"Must provide shares for all {n} parties")
    for dealer_idx, dealer_shares in enumerate(all_shares):
        if len(dealer_shares) != n:
            raise PvwError(1, f"Dealer {dealer_idx} provided {len(dealer_shares)} shares but needs {n}")
    return encrypt_many(all_shares, global_pk, [_dealer_seed(seed, d) for d in range(len(all_shares))], repr)


def encrypt_many(all_scalars: Sequence[Sequence[int]], global_pk: GlobalPublicKey, seeds: Sequence[bytes],
                 repr: int = REPR_NTT) -> List[PvwCiphertext]:
    """D independent encrypts (any D) batched through pvw_encrypt_multi."""
    p = global_pk.params
    D = len(all_scalars)
    if len(seeds) != D:
        raise PvwError(15, f"expected {D} seeds, got {len(seeds)}")
    sc = np.array([[int(s) & 0xFFFFFFFFFFFFFFFF for s in row] for row in all_scalars], dtype=np.uint64)
    if sc.ndim != 2:
        raise PvwError(1, "ragged scalar rows")
    sd = np.concatenate([_seed(s) for s in seeds]) if D else np.zeros(0, dtype=np.uint8)
    c1 = np.zeros((D, p.k, p.L, p.l), dtype=np.uint64)
    c2 = np.zeros((D, p.n, p.L, p.l), dtype=np.uint64)
    p._call("pvw_encrypt_multi", _ptr(sc), D, sc.shape[1] if D else 0, _ptr(sd), _ptr(c1), _ptr(c2), repr)
    return [PvwCiphertext(c1[d], c2[d], p, repr) for d in range(D)]


def encrypt_broadcast(scalar: int, global_pk: GlobalPublicKey, seed: bytes, **kw) -> PvwCiphertext:
    """encryption.rs:292-296."""
    return encrypt([scalar] * global_pk.params.n, global_pk, seed, **kw)


def decrypt_party_shares(all_ciphertexts: Sequence[PvwCiphertext], secret_key: SecretKey, party_index: int,
                         return_noisy: bool = False):
    """decrypt_party_shares (decryption.rs:281-325): one batched device pass over all dealers."""
    if len(all_ciphertexts) == 0:
        raise PvwError(1, "No ciphertexts provided")
    p = all_ciphertexts[0].params
    if len(all_ciphertexts) != p.n:
        raise PvwError(1, f"Expected {p.n} ciphertexts, got {len(all_ciphertexts)}")
    if party_index >= p.n:
        raise PvwError(1, f"Party index {party_index} exceeds maximum {p.n - 1}")
    for d, ct in enumerate(all_ciphertexts):
        try:
            ct.validate()
        except PvwError as e:
            raise PvwError(1, f"Ciphertext {d} invalid: {e}")
    return _decrypt_batch(p, all_ciphertexts, secret_key, party_index, return_noisy)


class DeviceSecretKey:
    """pvw_sk: a SecretKey's NTT form on the device (pvw_sk_load); cleared by free() / on drop (pvw_sk_free), as the
    reference's SecretKey is ZeroizeOnDrop (secret_key.rs:20-30).  Use as a context manager or call free()."""

    def __init__(self, secret_key: SecretKey):
        self.params = secret_key.params
        h = C.c_void_p()
        sk = _i64(secret_key.secret_coeffs)
        self.params._call("pvw_sk_load", _ptr(sk), C.byref(h))
        self._h = h

    def free(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            _check(_ffi.lib().pvw_sk_free(h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _decrypt_batch(p, cts, secret_key, party_index, return_noisy=False):
    repr = cts[0].repr
    c1s = np.ascontiguousarray(np.stack([ct.c1 for ct in cts]), dtype=np.uint64)
    c2col = np.ascontiguousarray(np.stack([ct.c2[party_index] for ct in cts]), dtype=np.uint64)
    sk = _i64(secret_key.secret_coeffs)
    out = np.zeros(len(cts), dtype=np.uint64)
    noisy = np.zeros((len(cts), p.L, p.l), dtype=np.uint64) if return_noisy else None
    p._call("pvw_decrypt_batch", _ptr(sk), _ptr(c1s), _ptr(c2col), len(cts), repr,
                                        _ptr(out), _ptr(noisy))
    vals = [int(v) for v in out]
    return (vals, noisy) if return_noisy else vals


def decrypt_party_value(ciphertext: PvwCiphertext, secret_key: SecretKey, party_index: int) -> int:
    """decrypt_party_value (decryption.rs:249-278)."""
    return _decrypt_batch(ciphertext.params, [ciphertext], secret_key, party_index)[0]


def _secret_residue(params: PvwParameters) -> Tuple[int, int]:
    """(non-zero words, scanned words) of the device regions the last key-bearing calls declared secret."""
    nz, sc = C.c_uint64(), C.c_uint64()
    params._call("pvw_selftest_secret_residue", C.byref(nz), C.byref(sc))
    return nz.value, sc.value


def _selftest_decode_fixed(params: PvwParameters, noisy: np.ndarray) -> List[int]:
    """Host run of the fixed-width decode the GPU executes (self-test hook, see pvw_hip.h)."""
    a = _u64(noisy).reshape(-1, params.L, params.l)
    out = np.zeros(len(a), dtype=np.uint64)
    params._call("pvw_selftest_decode_fixed", _ptr(a), len(a), _ptr(out))
    return [int(v) for v in out]


def decode_scalar_pvw(params: PvwParameters, noisy: np.ndarray) -> List[int]:
    """decode_scalar_pvw_rns (decryption.rs:10-58) on power-basis noisy polynomials [D][L][l], on the device."""
    a = _u64(noisy).reshape(-1, params.L, params.l)
    out = np.zeros(len(a), dtype=np.uint64)
    params._call("pvw_decode", _ptr(a), len(a), _ptr(out))
    return [int(v) for v in out]


def decode_scalar_pvw_host(params: PvwParameters, noisy: np.ndarray) -> List[int]:
    """The same decode with host big integers (no GPU): cross-check of the device algorithm."""
    a = _u64(noisy).reshape(-1, params.L, params.l)
    out = np.zeros(len(a), dtype=np.uint64)
    params._call("pvw_decode_host", _ptr(a), len(a), _ptr(out))
    return [int(v) for v in out]
