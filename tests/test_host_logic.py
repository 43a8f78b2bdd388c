"""CPU-only checks of the product's host side: the C-ABI library loads and exports every
symbol include/pvw_hip.h declares, parameter arithmetic / validation / decode agree with the
oracle and the golden fixtures.  No device compute is called here."""
import glob
import os
import re

import numpy as np
import pytest

import pvw_model as M
import pvw_rs_amd as P
from pvw_rs_amd import _ffi
from _util import TEST_MODULI, EXAMPLE_MODULI

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "pvw_hip.h")).read()
    declared = re.findall(r"^PVW_API int32_t (pvw_\w+)\(", header, flags=re.M)
    assert len(declared) >= 40
    lib = _ffi.lib()
    for name in declared:
        assert hasattr(lib, name), f"libpvw_hip.so does not export {name}"
    assert set(declared) == set(_ffi._SIGNATURES), "ctypes table and header disagree"


def test_tuning_library_is_a_superset_and_the_shipped_one_has_no_switches():
    # the measurement build exports everything the shipped one does plus include/pvw_hip_tuning.h; the shipped
    # library imports no getenv and carries none of the switch names: no environment variable can make it skip
    # sampling (the reference samples unconditionally, encryption.rs:135-167)
    import subprocess
    tuning_h = open(os.path.join(ROOT, "include", "pvw_hip_tuning.h")).read()
    extra = re.findall(r"^PVW_API int32_t (pvw_\w+)\(", tuning_h, flags=re.M)
    assert extra == ["pvw_selftest_read_bandwidth", "pvw_tuning_read_probe", "pvw_tuning_read_stamps"] and set(extra) == set(_ffi._TUNING_SIGNATURES)
    dflt, tun = _ffi._load("default"), _ffi.tuning_lib()
    assert dflt.pvw_build_is_tuning() == 0 and tun.pvw_build_is_tuning() == 1
    for name in extra:
        assert hasattr(tun, name) and not hasattr(dflt, name)
    for name in _ffi._SIGNATURES:
        assert hasattr(tun, name)
    def strings(path):
        return subprocess.run(["strings", "-a", path], capture_output=True, text=True, check=True).stdout
    def undefined(path):
        return subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
    switches = ("PVW_GEMM_ZERO_OPERANDS", "PVW_DECODE_TIMING", "PVW_DECODE_SMALL", "PVW_GEMM_BYTES", "PVW_MAC_VARIANT", "PVW_MAC_PACKED", "PVW_MAC_COMPACT", "PVW_DEC_VARIANT",
                "PVW_GEMM_MIN_DEALERS", "PVW_KEYGEN_SWAP")
    s_def, s_tun = strings(_ffi.LIB_PATH), strings(_ffi.LIB_TUNING_PATH)
    for name in switches:
        assert name not in s_def, f"the shipped library mentions {name}"
        assert name in s_tun
    assert "getenv" not in undefined(_ffi.LIB_PATH)
    assert "getenv" in undefined(_ffi.LIB_TUNING_PATH)


def test_workload_definition_agrees_with_the_checker():
    # bench.py's workload lives in the product (pvw_rs_amd/workloads.py); the oracle keeps its own copy of the rule
    from pvw_rs_amd import workloads as W
    assert W.bench_moduli(34) == M.bench_moduli(34) and W.bench_moduli(36) == M.bench_moduli(36)
    assert all(q % 64 == 1 and q < 1 << 61 for q in W.bench_moduli(34))
    assert (_ffi.DOM_R, _ffi.DOM_E1, _ffi.DOM_E2, _ffi.DOM_SK, _ffi.DOM_EKEY, _ffi.DOM_CRS, _ffi.DOM_GAUSS, _ffi.DOM_PK) == \
        (M.DOM_R, M.DOM_E1, M.DOM_E2, M.DOM_SK, M.DOM_EKEY, M.DOM_CRS, M.DOM_GAUSS, M.DOM_PK)
    assert W.ENCRYPT_CONFIGS["c3"][:4] == (4096, 256, 8, 17) and W.DECRYPT_CONFIGS["c5shard"][:4] == (1024, 512, 16, 34)


def test_crs_seed_from_tag_is_siphash13_of_tag_crs():
    # PvwCrs::new_from_tag (crs.rs:74-90): DefaultHasher = SipHash-1-3 with a zero key over tag + "CRS" + 0xFF.
    # The hash is pinned by the published SipHash-2-4 vector through the same code (round counts are parameters).
    import ctypes as C
    lib = _ffi.lib()
    msg = bytes(range(15))
    out = C.c_uint64()
    assert lib.pvw_selftest_siphash(msg, len(msg), 0x0706050403020100, 0x0F0E0D0C0B0A0908, 2, 4, C.byref(out)) == 0
    assert out.value == 0xA129CA6149BE45E5                        # SipHash paper, appendix A
    seed = P.PvwCrs.seed_from_tag("pvss-session-1")
    assert len(seed) == 32 and seed[:8] == seed[8:16] == seed[16:24] == seed[24:]
    m = b"pvss-session-1" + b"CRS" + b"\xff"
    assert lib.pvw_selftest_siphash(m, len(m), 0, 0, 1, 3, C.byref(out)) == 0
    assert seed[:8] == out.value.to_bytes(8, "little")
    assert P.PvwCrs.seed_from_tag("other") != seed


def test_secret_key_zeroize_on_drop():
    # secret_key.rs:20-30 / tests/keys.rs:515-538: the mirror owns its coefficients and clears them
    p = _builder().build()
    src = np.arange(4 * 8, dtype=np.int64).reshape(4, 8) - 7
    key = P.SecretKey.from_coefficients(p, src)
    view = key.secret_coeffs
    key.zeroize()
    assert not view.any() and src.any()                           # the caller's array is not the key's storage


def test_bench_self_launches_its_ranks():
    # `python bench.py --gpus 2` must start torch.distributed.run itself (as a child, before any GPU call);
    # on a box without a GPU both ranks then stop at the device check -- not at "must be launched with ..."
    import subprocess
    import sys
    if P.device_available():
        pytest.skip("GPU present: covered by tests/test_dist_gloo.py::test_bench_two_ranks_on_one_gpu")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a gfx950 GPU") >= 1, r.stderr[-2000:]
    assert "must be launched with" not in r.stderr


def test_error_codes_follow_pvw_error_order():
    header = open(os.path.join(ROOT, "include", "pvw_hip.h")).read()
    codes = dict((int(v), k) for k, v in re.findall(r"(PVW_ERR_\w+) = (\d+)", header))
    assert len(codes) == 19 and codes[1] == "PVW_ERR_INVALID_PARAMETERS" and codes[19] == "PVW_ERR_INTERNAL"
    assert _ffi.ERROR_NAMES[15] == "DimensionMismatch" and _ffi.ERROR_NAMES[16] == "IndexOutOfBounds"


def _builder(n=3, k=4, l=8, moduli=TEST_MODULI):
    return P.PvwParametersBuilder().set_parties(n).set_dimension(k).set_l(l).set_moduli(moduli)


def test_builder_validation_and_defaults():
    # parameters.rs:117-195; tests/params.rs:677-697; tests/keys.rs:541-576
    p = _builder().build()
    assert (p.secret_variance, p.error_bound_1, p.error_bound_2, p.t) == (0.5, 100, 200, 1)
    for bad in (_builder(n=0), _builder(k=0), _builder(l=4), _builder(l=12), _builder(moduli=[]),
                _builder(moduli=[0xFFFFEE001, 0xFFFFEE001]), _builder(moduli=[15]), _builder(moduli=[(1 << 62) + 1]),
                _builder(moduli=[41]),                    # prime but not 1 mod 16
                _builder().set_error_bounds_u32(0, 5), _builder().set_error_bounds_u32(5, 0)):
        with pytest.raises(P.PvwError) as e:
            bad.build()
        assert e.value.variant == "InvalidParameters"
    with pytest.raises(P.PvwError):
        P.PvwParametersBuilder().set_parties(3).build()     # k not set


@pytest.mark.parametrize("l,moduli", [(8, TEST_MODULI), (16, TEST_MODULI), (32, TEST_MODULI), (8, EXAMPLE_MODULI),
                                      (8, M.bench_moduli(17)), (16, M.bench_moduli(34))])
def test_delta_gadget_roots_match_model(l, moduli):
    p = _builder(l=l, moduli=moduli).build()
    m = M.Params(3, 4, l, moduli)
    assert p.q_total() == m.Q and p.delta() == m.delta and p.delta_power_l_minus_1() == m.delta_power_l_minus_1
    assert p.roots() == [M.minimal_primitive_root(q, 2 * l) for q in moduli]
    g = p.gadget_polynomial(P.REPR_POWER)
    assert p.poly_to_bigints(g) == m.gadget_vector()            # tests/crypto.rs:17-37
    # NTT-domain gadget = evaluation of the gadget at psi^(2 bitrev(s)+1)
    ghat = p.gadget_polynomial(P.REPR_NTT)
    for i, q in enumerate(moduli[:3]):
        assert ghat[i].tolist() == M.ntt_eval([int(v) for v in g[i]], q, p.roots()[i])
    assert p.verify_correctness_condition() == m.verify_correctness_condition()


def test_correctness_gate_and_suggested_bounds():
    for (n, k, l) in [(3, 4, 8), (10, 4, 16), (30, 64, 32), (5, 1024, 8)]:
        assert P.PvwParameters.suggest_error_bounds(n, k, l, TEST_MODULI, 0.5) == \
            M.Params.suggest_error_bounds(n, k, l, TEST_MODULI, 0.5)
    # failing gate: huge bounds on a small modulus (tests/crypto.rs:209-234 shape)
    p = _builder(moduli=[0xFFFFEE001]).set_error_bounds((1 << 50), (1 << 50)).build()
    assert not p.verify_correctness_condition()
    assert not M.Params(3, 4, 8, [0xFFFFEE001], 0.5, 1 << 50, 1 << 50).verify_correctness_condition()
    with pytest.raises(P.PvwError):
        P.PvwParameters.suggest_error_bounds(1 << 26, 1 << 12, 8, [0xFFFFEE001], 0.5)
    with pytest.raises(M.PvwError):
        M.Params.suggest_error_bounds(1 << 26, 1 << 12, 8, [0xFFFFEE001], 0.5)


def test_set_roots_validation():
    p = _builder().build()
    with pytest.raises(P.PvwError):
        p.set_roots([1, 1, 1])
    alt = []
    for q, psi in zip(TEST_MODULI, p.roots()):
        alt.append(pow(psi, 3, q))                               # another primitive 16th root
    p.set_roots(alt)
    assert p.roots() == alt
    g = p.gadget_polynomial(P.REPR_POWER)
    ghat = p.gadget_polynomial(P.REPR_NTT)
    assert ghat[0].tolist() == M.ntt_eval([int(v) for v in g[0]], TEST_MODULI[0], alt[0])


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(g)[:-4] for g in GOLDEN])
def test_decode_matches_golden(path):
    z = np.load(path)
    p = (P.PvwParametersBuilder().set_parties(int(z["n"])).set_dimension(int(z["k"])).set_l(int(z["l"]))
         .set_moduli([int(q) for q in z["moduli"]]).set_secret_variance(float(z["variance"]))
         .set_error_bounds(int(z["bound1"]), int(z["bound2"])).build())
    assert p.delta() == sum(int(w) << (64 * i) for i, w in enumerate(z["delta_words"]))
    assert P.decode_scalar_pvw_host(p, z["noisy_pb"]) == [int(v) for v in z["decoded"]]
    # the fixed-width algorithm the GPU runs (pvw_decode.h), executed on the host
    assert P.api._selftest_decode_fixed(p, z["noisy_pb"]) == [int(v) for v in z["decoded"]]


def test_decode_quirks_match_model():
    # decryption.rs:226-247: small negatives -> 0, big negatives -> 0, >= 2^64 -> 0
    rng = np.random.default_rng(7)
    for l, moduli in [(8, TEST_MODULI), (16, M.bench_moduli(34))]:
        p = _builder(l=l, moduli=moduli).build()
        m = M.Params(3, 4, l, moduli)
        cases = []
        for msg in (0, 1, 12345, -5, -1000, -1001, -5000, (1 << 63) - 1, (1 << 64) - 1, 1 << 64, (1 << 70) + 3):
            noise = [int(x) for x in rng.integers(-300, 300, size=l)]
            cases.append([(-(msg * m.delta ** j) + noise[j]) % m.Q for j in range(l)])
        for _ in range(20):   # random garbage polynomials must decode identically as well
            cases.append([int.from_bytes(rng.bytes(m.Q.bit_length() // 8 + 8), "little") % m.Q for _ in range(l)])
        arr = np.array([[[c % q for c in z] for q in moduli] for z in cases], dtype=np.uint64)
        want = [M.decode_scalar_pvw(z, m) for z in cases]
        assert P.decode_scalar_pvw_host(p, arr) == want
        assert P.api._selftest_decode_fixed(p, arr) == want


@pytest.mark.parametrize("l,moduli", [(8, [0xFFFFEE001]), (8, TEST_MODULI), (32, TEST_MODULI), (8, EXAMPLE_MODULI),
                                      (8, M.bench_moduli(17)), (16, M.bench_moduli(34))])
def test_fixed_width_decode_matches_model_on_random_and_edge_inputs(l, moduli):
    rng = np.random.default_rng(l * 1000 + len(moduli))
    p = _builder(l=l, moduli=moduli).build()
    m = M.Params(3, 4, l, moduli)
    Q, D = m.Q, m.delta
    cases = []
    for _ in range(60):      # uniformly random polynomials: exercises every branch of the integer decode
        cases.append([int.from_bytes(rng.bytes(Q.bit_length() // 8 + 8), "little") % Q for _ in range(l)])
    for msg in (0, 1, 7, 1000, 1001, -1, -1000, -1001, 2 ** 32, 2 ** 63, 2 ** 64 - 1, 2 ** 64):
        for amp in (0, 1, 50, 10 ** 4):
            noise = [int(x) for x in rng.integers(-amp, amp + 1, size=l)]
            cases.append([(-(msg * D ** j) + noise[j]) % Q for j in range(l)])
    # boundary values of the centring / halving comparisons
    half = Q // 2
    for v in (0, 1, half - 1, half, half + 1, Q - 1, m.delta_power_l_minus_1 // 2, m.delta_power_l_minus_1 // 2 + 1,
              m.delta_power_l_minus_1, D, D // 2, D // 2 + 1):
        cases.append([v % Q] * l)
        cases.append([(v * (j + 1)) % Q for j in range(l)])
    arr = np.array([[[c % q for c in z] for q in moduli] for z in cases], dtype=np.uint64)
    want = [M.decode_scalar_pvw(z, m) for z in cases]
    assert P.decode_scalar_pvw_host(p, arr) == want
    assert P.api._selftest_decode_fixed(p, arr) == want


def test_device_entry_points_fail_loudly_without_gpu():
    if P.device_available():
        pytest.skip("GPU present")
    p = _builder().build()
    with pytest.raises(P.PvwError) as e:
        P.PvwCrs.new_deterministic(p, bytes(32))
    assert e.value.variant == "InternalError" and "no CPU fallback" in str(e.value)
    with pytest.raises(P.PvwError):
        p.ntt_forward(np.zeros((1, 3, 8), dtype=np.uint64))


def test_decode_short_cut_tables_hold_their_identities():
    # The short cuts of the device gadget decode (decode_scalar_pvw_rns, decryption.rs:10-247) rest on per-context
    # constants: mixed-radix inverses and partial products of the leading moduli, the normalised 2*Delta and its
    # reciprocal, Delta^(l-1) mod q_i and its inverses.  Host-only self-test of their defining identities, and which short
    # cuts each parameter set reaches: (leading moduli, digits reduce by one subtraction, short chain, proven top noise)
    import ctypes as C
    from _util import primes_1mod
    from pvw_rs_amd import workloads as W
    lib = _ffi.lib()
    wide, narrow = primes_1mod(64, 6), primes_1mod(64, 6, top=1 << 40)
    mixed = [wide[i // 2] if i % 2 == 0 else narrow[i // 2] for i in range(12)]
    cases = [
        (8, [0xFFFFEE001], (0, 0, 0, 0)),                     # one modulus: nothing to confirm a candidate against
        (8, TEST_MODULI, (2, 1, 0, 0)),                       # Q of 109 bits: candidates only
        (8, W.bench_moduli(17), (3, 1, 1, 1)),                # configs[1..2]: everything
        (16, W.bench_moduli(34), (3, 1, 1, 1)),               # configs[3..4]
        (16, W.bench_moduli(17), (2, 1, 1, 1)),               # Delta of 65 bits: a two-word divisor
        (64, primes_1mod(128, 5), (2, 1, 1, 0)),              # Delta of 5 bits: no room for the proof of noise_{l-1}
        (8, EXAMPLE_MODULI, (2, 1, 1, 0)),                    # the reference's 4 x 56-bit chain, Delta of 28 bits
        (8, mixed, (2, 0, 1, 1)),                             # 61- and 40-bit moduli alternating
    ]
    for l, moduli, want in cases:
        p = (P.PvwParametersBuilder().set_parties(3).set_dimension(4).set_l(l).set_moduli(moduli)
             .set_secret_variance(0.5).set_error_bounds(100, 200).build())
        info = (C.c_uint32 * 4)()
        rc = lib.pvw_selftest_decode_tables(p._h, info)
        assert rc == 0, _ffi.last_error()
        assert tuple(info) == want, (l, len(moduli), tuple(info))


@pytest.mark.parametrize("l,L,kind", [(8, 17, "bench"), (16, 34, "bench"), (16, 17, "bench"), (8, 12, "mixed"), (8, 3, "test")])
def test_decode_short_path_restated_on_the_host_matches_the_model(l, L, kind):
    # The short path of the device decode (candidates from a few residues confirmed on every limb, noise_{l-1} proven from
    # the residues, the chain to a fixed point) restated sequentially on the host with the very arithmetic the kernel
    # uses (pvw_selftest_decode_shortcuts): against the big-integer model (decode_scalar_pvw_rns, decryption.rs:10-247)
    # on inputs on and around every bound those proofs rely on; ciphertext-shaped inputs must have taken the short path,
    # uniform ones must not have
    import ctypes as C
    from _util import decode_cases, primes_1mod
    from pvw_rs_amd import workloads as W
    if kind == "bench":
        moduli = W.bench_moduli(L)
    elif kind == "mixed":
        wide, narrow = primes_1mod(64, L // 2), primes_1mod(64, L // 2, top=1 << 40)
        moduli = [wide[i // 2] if i % 2 == 0 else narrow[i // 2] for i in range(L)]
    else:
        moduli = TEST_MODULI
    p = (P.PvwParametersBuilder().set_parties(3).set_dimension(4).set_l(l).set_moduli(moduli)
         .set_secret_variance(0.5).set_error_bounds(100, 200).build())
    m = M.Params(3, 4, l, moduli)
    cases = decode_cases(l, moduli)
    arr = np.ascontiguousarray(np.array([[[c % q for c in z] for q in moduli] for z in cases], dtype=np.uint64))
    out = np.zeros(len(cases), dtype=np.uint64)
    took = np.zeros(len(cases), dtype=np.uint8)
    rc = _ffi.lib().pvw_selftest_decode_shortcuts(p._h, arr.ctypes.data_as(C.c_void_p), len(cases), out.ctypes.data_as(C.c_void_p),
                                                  took.ctypes.data_as(C.c_void_p))
    assert rc == 0, _ffi.last_error()
    want = [M.decode_scalar_pvw(z, m) for z in cases]
    assert [int(x) for x in out] == want
    if kind == "test":
        assert not took.any()                       # Q of 109 bits: the chain's short form is off, nothing takes the whole short path
    else:
        assert not took[:150].any()                 # the uniform inputs
        assert took[150:183].all()                  # message * Delta^j + small noise, 33 of them
