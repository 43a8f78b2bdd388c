"""Pins oracle/pvw_model.py against the properties the reference's own tests state
(the reference holds no golden vectors -- SURVEY.md 8c).  CPU only."""
import random
import struct

import pytest

import pvw_model as M
from _util import EXAMPLE_MODULI, SEED, TEST_MODULI, make_system


def test_chacha_known_answers():
    # ChaCha20 block 0, zero key/nonce (RFC 7539 2.3.2 family, widely published) and
    # the ChaCha8 zero-key/zero-IV keystream of the eSTREAM test-vector set.
    b20 = struct.pack("<16I", *M.chacha_block([0] * 8, 0, 0, rounds=20)).hex()
    assert b20.startswith("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7")
    b8 = struct.pack("<16I", *M.chacha_block([0] * 8, 0, 0, rounds=8)).hex()
    assert b8 == ("3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e"
                  "984ce172b9216f419f445367456d5619314a42a3da86b001387bfdb80e0cfe42")


def test_bench_moduli_match_survey():
    mods = M.bench_moduli(34)
    assert mods[0] == 0x1FFFFFFFFFFFFB41 and mods[16] == 0x1FFFFFFFFFFFCA41 and mods[33] == 0x1FFFFFFFFFFF7281
    Q = 1
    for q in mods[:17]:
        Q *= q
    assert Q.bit_length() == 1037


def test_builder_validation():
    # parameters.rs:131-181, tests/params.rs:677-697, tests/keys.rs:541-576
    for bad in [dict(n=0, k=4, l=8), dict(n=3, k=0, l=8), dict(n=3, k=4, l=4), dict(n=3, k=4, l=12)]:
        with pytest.raises(M.PvwError):
            M.Params(bad["n"], bad["k"], bad["l"], TEST_MODULI)
    with pytest.raises(M.PvwError):
        M.Params(3, 4, 8, TEST_MODULI, 0.5, 0, 10)
    p = M.Params(7, 4, 8, TEST_MODULI)
    assert (p.secret_variance, p.error_bound_1, p.error_bound_2, p.t) == (0.5, 100, 200, 3)
    assert p.delta ** 8 <= p.Q < (p.delta + 1) ** 8


def test_gadget_structure():
    # tests/crypto.rs:17-37, tests/params.rs:638-674
    p = M.Params(3, 4, 8, TEST_MODULI)
    g = M.from_rns(M.to_rns(p.gadget_vector(), p.moduli), p.moduli)
    assert g == [p.delta ** j for j in range(8)]
    assert g[-1] == p.delta_power_l_minus_1


def test_bigints_to_poly_round_trips():
    # tests/params.rs:485-635
    p = M.Params(3, 64, 8, TEST_MODULI, 0.5, 100, 200)
    cases = [
        [0] * 8,
        list(range(1, 9)),
        [p.delta * (i + 1) for i in range(8)],
        [-(i * 100) for i in range(1, 9)],
        [42, -123, p.delta // 2, 0, 1, -1, 999999, -888888],
    ]
    for coeffs in cases:
        lifted = M.from_rns(M.to_rns(coeffs, p.moduli), p.moduli)
        assert lifted == [c % p.Q for c in coeffs]
    # from_coefficients(i64) == bigints_to_poly  (tests/params.rs:733-767)
    small = [5, -3, 0, 7, -1, 2, -9, 4]
    assert M.to_rns(small, p.moduli) == [[c % q for c in small] for q in p.moduli]


def test_rounding_division_table():
    # tests/crypto.rs:308-330
    for dividend, divisor, expected in [(7, 3, 2), (8, 3, 3), (-7, 3, -2), (-8, 3, -3)]:
        tw = 2 * dividend
        got = M.tdiv(tw - divisor, 2 * divisor) if dividend < 0 else M.tdiv(tw + divisor, 2 * divisor)
        assert got == expected
    assert M.trem(-7, 3) == -1 and M.trem(7, -3) == 1


def test_cbd_statistics():
    # tests/sampling.rs:198-274, tests/keys.rs:275-307,431-459
    xs = []
    for p in range(1250):
        xs += M.sample_vec_cbd(8, 0.5, M.ChaChaRng(SEED, M.DOM_R, p))
    assert set(xs) <= {-1, 0, 1}
    mean = sum(xs) / len(xs)
    var = sum((x - mean) ** 2 for x in xs) / len(xs)
    assert abs(mean) < 0.1 and abs(var - 0.5) < 0.1
    ys = []
    for p in range(500):
        ys += M.sample_vec_cbd(16, 1.0, M.ChaChaRng(SEED, M.DOM_SK, p))
    assert min(ys) >= -2 and max(ys) <= 2
    vy = sum(y * y for y in ys) / len(ys)
    assert abs(vy - 1.0) < 0.15
    zs = M.sample_vec_cbd(4096, 10.0, M.ChaChaRng(SEED, M.DOM_SK, 7))
    vz = sum(z * z for z in zs) / len(zs)
    assert abs(vz - 10.0) < 1.0 and max(abs(z) for z in zs) <= 20
    with pytest.raises(M.PvwError):
        M.sample_vec_cbd(8, 0.3, M.ChaChaRng(SEED, 0, 0))
    with pytest.raises(M.PvwError):
        M.sample_vec_cbd(8, 17.0, M.ChaChaRng(SEED, 0, 0))


def test_uniform_bounds():
    for bound in (1, 50, 200, 1172385, (1 << 40) + 12345):
        xs = M.sample_uniform_coefficients(bound, 2000, M.ChaChaRng(SEED, M.DOM_E2, bound & 0xFFFF))
        assert min(xs) >= -bound and max(xs) <= bound
        if bound >= 50:
            assert min(xs) < -bound // 2 and max(xs) > bound // 2


def test_gaussian_bound_respected():
    # tests/sampling.rs:181-195
    for bound in (1, 4, 100, 10 ** 6):
        rng = M.ChaChaRng(SEED, M.DOM_GAUSS, bound & 0xFFFF)
        xs = [M.sample_single_gaussian(bound, rng) for _ in range(500)]
        assert all(-bound <= x <= bound for x in xs)
    assert M.sample_single_gaussian(0, M.ChaChaRng(SEED, M.DOM_GAUSS, 0)) == 0


def test_correctness_gate():
    # tests/params.rs:277-299,463-481
    b1, b2 = M.Params.suggest_error_bounds(3, 4, 8, TEST_MODULI, 0.5)
    assert M.Params(3, 4, 8, TEST_MODULI, 0.5, b1, b2).verify_correctness_condition()
    b1, b2 = M.Params.suggest_error_bounds(30, 64, 32, TEST_MODULI, 0.5)
    assert M.Params(30, 64, 32, TEST_MODULI, 0.5, b1, b2).verify_correctness_condition()
    # 2074-bit Q, l=16: D^(l-1) overflows f64 -> +inf -> gate passes
    big = M.Params(16, 8, 16, M.bench_moduli(34))
    assert M.big_to_f64(big.delta_power_l_minus_1) == float("inf")
    assert big.verify_correctness_condition()


@pytest.mark.parametrize("n,k,l,moduli", [
    (3, 4, 8, TEST_MODULI),          # tests/crypto.rs:56-70
    (10, 4, 16, TEST_MODULI),        # tests/crypto.rs:237-305
    (5, 8, 8, EXAMPLE_MODULI),       # examples/pvw_valid_dec.rs:40-52 at reduced k
])
def test_encrypt_decrypt_round_trip(n, k, l, moduli):
    s = make_system(n, k, l, moduli)
    P = s["P"]
    ok = 0
    for i in range(n):
        ok += M.decrypt_party_value(P, s["c1"], s["c2"][i], s["sk"][i]) == s["scalars"][i]
    assert ok >= 0.95 * n          # the reference's own bar (tests/crypto.rs:301-304)
    assert ok == n                 # and in fact exact for these seeds


def test_decode_quirks():
    P = M.Params(3, 4, 8, TEST_MODULI)
    Q, D = P.Q, P.delta
    def noisy_for(m, noise):
        return [(-(m * D ** j) + noise[j]) % Q for j in range(8)]
    rnd = random.Random(1)
    noise = [rnd.randint(-50, 50) for _ in range(8)]
    assert M.decode_scalar_pvw(noisy_for(12345, noise), P) == 12345
    assert M.decode_scalar_pvw(noisy_for(0, noise), P) == 0
    # small negative result -> 0 (decryption.rs:233)
    assert M.decode_scalar_pvw(noisy_for(-5, noise), P) == 0
    # large negative -> (v+Q)%Q does not fit u64 -> 0 (decryption.rs:238-240)
    assert M.decode_scalar_pvw(noisy_for(-5000, noise), P) == 0
    # `scalars[i] as i64` wrap: m >= 2^63 encodes a negative number (encryption.rs:195)
    assert M.u64_as_i64((1 << 64) - 5) == -5
