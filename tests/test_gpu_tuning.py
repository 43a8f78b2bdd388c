"""GPU tests of the MEASUREMENT build libpvw_hip_tuning.so (-DPVW_TUNING=1, include/pvw_hip_tuning.h): every
kernel schedule / launch shape / decode form that the environment selectors reach computes what the oracle
computes, and the timing switches exist there -- and only there.  The shipped library picks its schedules by
shape and reads no environment variable (tests/test_gpu_parity.py runs on it)."""
import ctypes as C

import numpy as np
import pytest

import pvw_model as M
import pvw_rs_amd as P
from pvw_rs_amd import _ffi
from _util import SEED, TEST_MODULI, primes_1mod
import test_gpu_parity as T

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def tuning_library():
    """contexts created inside these tests live in libpvw_hip_tuning.so (a context keeps its library for life)"""
    prev = _ffi.select("tuning")
    yield
    _ffi.select(prev)


def test_tuning_build_is_selected():
    p = T.build_params(3, 4, 8, TEST_MODULI)
    assert p._lib.pvw_build_is_tuning() == 1 and _ffi._load("default").pvw_build_is_tuning() == 0


@pytest.mark.parametrize("n,k,l,L", T.MAC_SCHEDULE_CASES + [(20, 512, 8, 2), (10, 256, 16, 2), (3000, 256, 16, 4)])
def test_mac_rows_schedules_agree_with_c_oracle(n, k, l, L, monkeypatch):
    # the schedules of the tiled-stream mac_rows that are left (PVW_MAC_VARIANT: 0 by shape -- which streams the packed copy
    # where it can --, 17 not interleaved, 40 interleaved + time stamps): the same c1, c2 (encryption.rs:158,177-200)
    # each with the addends in compact form (what ships for l <= 16) and as full polynomials from the prologue (PVW_MAC_COMPACT=0)
    run, c1o, c2o = T.mac_rows_case(n, k, l, L)
    for compact in ("1", "0"):
        monkeypatch.setenv("PVW_MAC_COMPACT", compact)
        for variant in (0, 17, 40):
            monkeypatch.setenv("PVW_MAC_VARIANT", str(variant))
            ct = run()
            assert np.array_equal(ct.c1, c1o) and np.array_equal(ct.c2, c2o), (compact, variant)


@pytest.mark.parametrize("n,k,l,L", T.MAC_PACKED_CASES)
def test_mac_rows_packed_and_unpacked_streams_agree(n, k, l, L, monkeypatch):
    # PVW_MAC_PACKED=0: the geometries the shipped library streams from the 61-bit packed copy, served by the
    # unpacked mac_rows_kernel instead; 44: the stamped packed kernel -- same ciphertexts, all equal to the oracle's
    run, c1o, c2o = T.mac_rows_case(n, k, l, L)
    for packed, variant, compact, width in (("1", "0", "1", 61), ("0", "0", "1", 0), ("1", "44", "1", 61), ("1", "0", "0", 61), ("0", "0", "0", 0)):
        monkeypatch.setenv("PVW_MAC_PACKED", packed)
        monkeypatch.setenv("PVW_MAC_VARIANT", variant)
        monkeypatch.setenv("PVW_MAC_COMPACT", compact)
        ct = run()
        assert np.array_equal(ct.c1, c1o) and np.array_equal(ct.c2, c2o), (packed, variant, compact)
        if packed == "1":
            assert run.params.packed_active() == width


@pytest.mark.parametrize("n,k,l,moduli,width", T.MAC_PACKED_WIDTH_CASES[:7])
def test_mac_rows_width_streams_stamped_and_unpacked(n, k, l, moduli, width, monkeypatch):
    # the 40 / 48 / 56-bit streams with per-workgroup time stamps (44), and the same geometry from the tiled matrices
    run, c1o, c2o = T.mac_rows_case(n, k, l, None, moduli)
    for packed, variant, compact in (("1", "44", "1"), ("0", "0", "1"), ("1", "0", "0"), ("1", "0", "1")):
        monkeypatch.setenv("PVW_MAC_PACKED", packed)
        monkeypatch.setenv("PVW_MAC_VARIANT", variant)
        monkeypatch.setenv("PVW_MAC_COMPACT", compact)
        ct = run()
        assert np.array_equal(ct.c1, c1o) and np.array_equal(ct.c2, c2o), (packed, variant, compact)
    assert run.params.packed_active() == width


@pytest.mark.parametrize("k,l,L,D", T.DECRYPT_SHAPE_CASES)
def test_decrypt_mac_launch_shapes_agree_with_c_oracle(k, l, L, D, monkeypatch):
    # every launch shape of decrypt_party_value's <sk, c1> (decryption.rs:257-274) gives the oracle's noisy
    # polynomials: the shape-selected default, the dealer-grouped form and the full-width form, each with 1, 2, 3
    # j-replicas
    run, want = T.decrypt_mac_case(k, l, L, D)
    for variant, c in [(0, 0), (10, 0), (10, 2), (10, 3), (60, 0), (60, 1), (60, 2), (60, 3)]:
        monkeypatch.setenv("PVW_DEC_VARIANT", str(variant))
        monkeypatch.setenv("PVW_DEC_C", str(c))
        assert np.array_equal(run(), want), (variant, c)


@pytest.mark.parametrize("D", [1, 2, 4, 7])
def test_multi_dealer_encrypt_on_the_integer_valu(D, monkeypatch):
    # PVW_GEMM_MIN_DEALERS=0: mac_rows_multi (up to four dealers per pass over B-hat) at every dealer count
    monkeypatch.setenv("PVW_GEMM_MIN_DEALERS", "0")
    T.multi_dealer_case(D)


@pytest.mark.parametrize("l,moduli", [(8, TEST_MODULI), (8, M.bench_moduli(17)), (16, M.bench_moduli(34)), (64, primes_1mod(128, 5))])
def test_device_decode_forms_match_model(l, moduli, monkeypatch):
    # decode_scalar_pvw_rns (decryption.rs:10-58): the lifted chain (four waves per ciphertext) and one thread per ciphertext
    # -- and the chain with every lift in full (PVW_DECODE_SMALL=0: no short cut for noise-sized chain inputs)
    def select(v):
        monkeypatch.setenv("PVW_DECODE_VARIANT", "1" if v == 1 else "0")
        monkeypatch.setenv("PVW_DECODE_SMALL", "0" if v == "full lifts" else "1")
    T.device_decode_case(l, moduli, select, (0, 1, "full lifts"))


@pytest.mark.parametrize("D", [5, 70])
def test_multi_dealer_encrypt_with_e2_from_the_prologue(D, monkeypatch):
    # PVW_FUSED_E2=0: e2 + m g-hat come from the prologue as an addend of the c2 finish pass (the round-1 form) instead
    # of being drawn inside it -- same ciphertexts (encryption.rs:195-196)
    monkeypatch.setenv("PVW_FUSED_E2", "0")
    T.multi_dealer_case(D)


@pytest.mark.parametrize("n,k,l,L", [(150, 256, 8, 2), (70, 9, 8, 3)])
def test_batched_keygen_transposed_crs_form(n, k, l, L, monkeypatch):
    # PVW_KEYGEN_SWAP=0: the transposed CRS is the streamed operand and super-groups of secret keys are digitised
    monkeypatch.setenv("PVW_KEYGEN_SWAP", "0")
    T.batched_keygen_case(n, k, l, L)


def test_read_bandwidth_probe_runs():
    # the measurement aid bench.py reports beside mac_rows (same loads, no arithmetic): runs and gives a sane rate
    p = T.build_params(2048, 256, 8, M.bench_moduli(17))
    gpk = P.GlobalPublicKey.new(P.PvwCrs.new_deterministic(p, SEED))
    gpk.fill_uniform(SEED)
    sec, nbytes = C.c_double(0.0), C.c_uint64(0)
    p._call("pvw_selftest_read_bandwidth", 5, C.byref(sec), C.byref(nbytes))
    assert nbytes.value == 2048 // 16 * 17 * 256 * 1024
    assert 500.0 < nbytes.value / sec.value / 1e9 < 8000.0


def test_switches_exist_only_in_the_tuning_build(monkeypatch):
    # PVW_MAC_PACKED=0 makes the TUNING build stream the tiled matrices; the shipped library has neither the lookup nor
    # the branch, so no variable can change what it runs (the reference samples and computes unconditionally,
    # encryption.rs:135-167) -- and the ciphertexts are the same either way
    def once(lib_name):
        prev = _ffi.select(lib_name)
        try:
            p = T.build_params(40, 256, 8, M.bench_moduli(3))
            gpk = P.GlobalPublicKey.new(P.PvwCrs.new_deterministic(p, SEED))
            gpk.fill_uniform(SEED)
        finally:
            _ffi.select(prev)
        ct = P.encrypt([(i * 1000 + 1) % (1 << 32) for i in range(40)], gpk, SEED)
        return ct.c1.copy(), ct.c2.copy(), p.packed_active()
    want = once("default")
    assert want[2] == 61 and once("tuning")[2] == 61
    for name, val in (("PVW_MAC_PACKED", "0"), ("PVW_MAC_COMPACT", "0"), ("PVW_DECODE_TIMING", "1"), ("PVW_GEMM_ZERO_OPERANDS", "1"), ("PVW_MAC_VARIANT", "17")):
        monkeypatch.setenv(name, val)
    got = once("default")
    assert all(np.array_equal(a, b) for a, b in zip(got[:2], want[:2])) and got[2] == 61, "the shipped library reacted to a tuning variable"
    got_t = once("tuning")
    assert all(np.array_equal(a, b) for a, b in zip(got_t[:2], want[:2])) and got_t[2] == 0


def test_digit_gemm_round1_form_agrees(monkeypatch):
    # PVW_GEMM_WIDE=0: more than 16 vectors through gemm_digits_kernel (128 rows x 16 vectors per workgroup) instead of the
    # wide ping-pong form that ships and that tests/test_gpu_parity.py runs: same ciphertexts, k = 256 and k = 512
    monkeypatch.setenv("PVW_GEMM_WIDE", "0")
    T.test_digit_gemm_multi_dealer_equals_separate_encrypts(100, 256, 8, 3, 40)
    T.test_digit_gemm_multi_dealer_equals_separate_encrypts(40, 512, 16, 2, 17)


@pytest.mark.parametrize("n,k,l,L,D", [(37, 64, 8, 4, 5), (70, 128, 8, 3, 33)])
def test_digit_gemm_eight_byte_contraction_on_short_moduli(n, k, l, L, D, monkeypatch):
    # PVW_GEMM_BYTES=8: moduli below 2^56 through the general 8-byte contraction (what ran before the 7-byte form existed)
    from _util import EXAMPLE_MODULI
    monkeypatch.setenv("PVW_GEMM_BYTES", "8")
    T.digit_gemm_case(n, k, l, EXAMPLE_MODULI[:L], D)
