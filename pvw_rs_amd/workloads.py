"""The synthetic workloads bench.py and the full-size parity tests run: BASELINE.json's configs as
geometries, the deterministic modulus chain of SURVEY.md 8(d) and the fixed seeds.  Product-side
definitions -- nothing here touches oracle/ (the checker has its own copy of the modulus rule, and
tests/test_host_logic.py asserts the two agree)."""
from __future__ import annotations

from typing import Dict, List, Tuple

# the first 34 primes found descending from 2^61 in steps of 64 with p = 1 (mod 64) (valid for l <= 32):
# 17 limbs = 1037-bit Q (configs[0..2]), 34 limbs = 2074-bit Q (configs[3..4]).  The reference's own largest chain
# is 4 x 56 bits (examples/pvw_valid_dec.rs:40-45); BASELINE.json asks for 1024 / 2048 bits.
_CHAIN_2_61_STEP_64 = [
    0x1ffffffffffffb41, 0x1ffffffffffff8c1, 0x1fffffffffffef01, 0x1fffffffffffed01, 0x1fffffffffffe601,
    0x1fffffffffffe281, 0x1fffffffffffdf41, 0x1fffffffffffdec1, 0x1fffffffffffde81, 0x1fffffffffffdd41,
    0x1fffffffffffd801, 0x1fffffffffffd741, 0x1fffffffffffd581, 0x1fffffffffffd401, 0x1fffffffffffd081,
    0x1fffffffffffca81, 0x1fffffffffffca41, 0x1fffffffffffc8c1, 0x1fffffffffffc681, 0x1fffffffffffbf41,
    0x1fffffffffffb901, 0x1fffffffffffb101, 0x1fffffffffffab81, 0x1fffffffffffaac1, 0x1fffffffffffa641,
    0x1fffffffffff9f81, 0x1fffffffffff9a81, 0x1fffffffffff9781, 0x1fffffffffff9301, 0x1fffffffffff9081,
    0x1fffffffffff8c41, 0x1fffffffffff8901, 0x1fffffffffff7d01, 0x1fffffffffff7281,
]


def _is_prime(n: int) -> bool:
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in small:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d, s = d // 2, s + 1
    for a in small:                      # deterministic below 2^64
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


# the reference's own "128-bit" chain: 4 x 56 bits (examples/pvw_valid_dec.rs:40-45)
REFERENCE_128_MODULI = [0x800000022A0001, 0x800000021A0001, 0x80000002120001, 0x80000001F60001]


def config_moduli(name: str, limbs: int) -> List[int]:
    """modulus chain of a bench configuration: SURVEY 8d's 61-bit chain unless the configuration names its own"""
    return list(REFERENCE_128_MODULI[:limbs]) if name == "ref128x" else bench_moduli(limbs)


def bench_moduli(count: int) -> List[int]:
    """First `count` primes of the chain (the table above; continued by search beyond it)."""
    out = list(_CHAIN_2_61_STEP_64[:count])
    p = out[-1] if out else (1 << 61) + 1
    while len(out) < count:
        p -= 64
        if _is_prime(p):
            out.append(p)
    return out


# (rows per GPU, k, l, RNS limbs, description).  rows = parties (encrypt / keygen) or dealer ciphertexts (decrypt)
ENCRYPT_CONFIGS: Dict[str, Tuple[int, int, int, int, str]] = {
    "c1": (16, 256, 8, 17, "BASELINE configs[0]: n=16, k=256, l=8, 1037-bit q (plumbing)"),
    "c2": (1024, 256, 8, 17, "BASELINE configs[1]: n=1024, k=256, l=8, 1037-bit q"),
    "c3": (4096, 256, 8, 17, "BASELINE configs[2] / north-star target: n=4096, k=256, l=8, 1037-bit q (17 limbs)"),
    "c3x4": (16384, 256, 8, 17, "sizing experiment: config 3 geometry with n=16384 parties on one GPU"),
    "c4full": (16384, 512, 16, 34, "BASELINE configs[3] in full on ONE GPU: n=16384, k=512, l=16, 2074-bit q (B-hat 36.5 GB)"),
    "c4shard": (2048, 512, 16, 34, "BASELINE configs[3] per-GPU shard: n=16384/8, k=512, l=16, 2074-bit q"),
    "ref128x": (4096, 1024, 8, 4, "the reference's own 128-bit set (examples/pvw_valid_dec.rs:40-52: k=1024, l=8, 4 x 56-bit q) at n=4096 parties"),
}
DECRYPT_CONFIGS: Dict[str, Tuple[int, int, int, int, str]] = {
    "c5shard": (1024, 512, 16, 34, "BASELINE configs[4] per-GPU shard: D=8192/8 dealer ciphertexts, k=512, l=16, 2074-bit q"),
    "c5full": (8192, 512, 16, 34, "BASELINE configs[4] in full on ONE GPU: D=8192 dealer ciphertexts, k=512, l=16, 2074-bit q (18.3 GB)"),
    "d3": (2048, 256, 8, 17, "decrypt of D=2048 dealer ciphertexts at the config-3 geometry: k=256, l=8, 1037-bit q"),
    "c5one": (1, 512, 16, 34, "one decrypt_party_value at the config-5 geometry (latency): k=512, l=16, 2074-bit q"),
    "c5x16": (16, 512, 16, 34, "16 dealer ciphertexts at the config-5 geometry: k=512, l=16, 2074-bit q"),
}

# synthetic inputs: CRS / public-key / encrypt seeds, builder defaults (parameters.rs:166-168)
SEED_A, SEED_B, SEED_ENC = bytes([0xA]) * 32, bytes([0xB]) * 32, bytes([0x2A]) * 32
SECRET_VARIANCE, ERROR_BOUND_1, ERROR_BOUND_2 = 0.5, 100, 200


def scalars(n: int) -> List[int]:
    """m_i = (i * 1000 + 1) mod 2^32 (the pattern of examples/pvw.rs:98-100)."""
    return [(i * 1000 + 1) % (1 << 32) for i in range(n)]
