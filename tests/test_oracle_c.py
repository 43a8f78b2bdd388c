"""C restatement (oracle/pvw_oracle.c) vs the big-integer model, in the power basis.  CPU only."""
import numpy as np
import pytest

import pvw_model as M
import pvw_oracle as O
from _util import EXAMPLE_MODULI, SEED, TEST_MODULI, make_system, ntt_rows, psi_list, ring_to_rns


def test_min_primitive_root_matches_model():
    for q in TEST_MODULI + EXAMPLE_MODULI + M.bench_moduli(4):
        for l in (8, 16, 32):
            psi = O.lib().pvwo_min_primitive_root(q, 2 * l)
            assert psi == M.minimal_primitive_root(q, 2 * l)
            assert pow(psi, l, q) == q - 1


@pytest.mark.parametrize("l", [8, 16, 32])
def test_ntt_is_evaluation_at_odd_powers(l):
    moduli = M.bench_moduli(3)
    orc = O.Oracle(moduli, l)
    rng = np.random.default_rng(l)
    polys = np.stack([rng.integers(0, q, size=(5, l), dtype=np.uint64) for q in moduli], axis=1)
    fwd = orc.ntt_forward(polys)
    assert np.array_equal(fwd, ntt_rows(polys, moduli, l))
    assert np.array_equal(orc.ntt_inverse(fwd), polys)


def test_samplers_match_model():
    for var in (0.5, 1.0, 3.0, 10.0, 16.0):
        got = O.sample_cbd(SEED, M.DOM_R, 5, 6, 16, var)
        want = [M.sample_vec_cbd(16, var, M.ChaChaRng(SEED, M.DOM_R, 5 + p)) for p in range(6)]
        assert got.tolist() == want
    for bound in (1, 50, 200, 1172385, (1 << 31) - 1, (1 << 32), (1 << 40) + 12345, (1 << 61)):
        got = O.sample_uniform(SEED, M.DOM_E2, 3, 4, 8, bound)
        want = [M.sample_uniform_coefficients(bound, 8, M.ChaChaRng(SEED, M.DOM_E2, 3 + p)) for p in range(4)]
        assert got.tolist() == want
    with pytest.raises(ValueError):
        O.sample_cbd(SEED, 0, 0, 1, 8, 0.7)
    moduli = M.bench_moduli(2)
    orc = O.Oracle(moduli, 8)
    got = orc.fill_uniform(SEED, M.DOM_CRS, 2, 3)
    for p in range(3):
        for i, q in enumerate(moduli):
            want = M.sample_uniform_residues(q, 8, M.ChaChaRng(SEED, M.DOM_CRS, (2 + p) * 2 + i))
            assert got[p, i].tolist() == want


@pytest.mark.parametrize("n,k,l,moduli", [
    (3, 4, 8, TEST_MODULI),
    (10, 4, 16, TEST_MODULI),
    (6, 5, 32, TEST_MODULI),
    (5, 8, 8, EXAMPLE_MODULI),
    (4, 6, 8, M.bench_moduli(17)),
])
def test_encrypt_keygen_decrypt_match_model(n, k, l, moduli):
    s = make_system(n, k, l, moduli)
    P = s["P"]
    orc = O.Oracle(moduli, l)
    assert orc.psi.tolist() == psi_list(moduli, l)
    a_hat = orc.ntt_forward(s["A_pb"])
    # keygen: B = S*A + E
    b_hat = orc.keygen(a_hat, np.array(s["sk"], dtype=np.int64), np.array(s["ek"], dtype=np.int64))
    assert np.array_equal(orc.ntt_inverse(b_hat), s["B_pb"])
    g_hat = orc.ntt_forward(s["g_pb"][None])[0]
    for serial in (False, True):
        c1, c2 = orc.encrypt(a_hat, b_hat, g_hat, np.array(s["scalars"], dtype=np.uint64),
                             np.array(s["r"]), np.array(s["e1"]), np.array(s["e2"]), serial_c1=serial)
        assert np.array_equal(orc.ntt_inverse(c1), s["c1_pb"])
        assert np.array_equal(orc.ntt_inverse(c2), s["c2_pb"])
    # decrypt: party p over the single ciphertext
    for p in range(n):
        noisy = orc.decrypt_noisy(np.array(s["sk"][p]), c1[None], c2[p][None])[0]
        want = M.decrypt_noisy(P, s["c1"], s["c2"][p], s["sk"][p])
        assert np.array_equal(noisy, ring_to_rns(want, moduli))
        assert M.decode_scalar_pvw(want, P) == s["scalars"][p]


def test_i64_wrap_of_scalars():
    # m >= 2^63 is encoded as a negative number (encryption.rs:195)
    moduli, l, n, k = TEST_MODULI, 8, 3, 4
    scalars = [(1 << 64) - 5, 1 << 63, (1 << 63) - 1]
    s = make_system(n, k, l, moduli, scalars=scalars)
    orc = O.Oracle(moduli, l)
    a_hat, b_hat = orc.ntt_forward(s["A_pb"]), orc.ntt_forward(s["B_pb"])
    g_hat = orc.ntt_forward(s["g_pb"][None])[0]
    c1, c2 = orc.encrypt(a_hat, b_hat, g_hat, np.array(scalars, dtype=np.uint64),
                         np.array(s["r"]), np.array(s["e1"]), np.array(s["e2"]))
    assert np.array_equal(orc.ntt_inverse(c2), s["c2_pb"])


@pytest.mark.parametrize("n,k,l,moduli", [(9, 7, 8, TEST_MODULI), (6, 12, 16, M.bench_moduli(5)), (5, 4, 32, EXAMPLE_MODULI)])
def test_barrett_build_agrees_with_plain_remainder_build(n, k, l, moduli):
    # libpvw_oracle.so reduces with a 128-bit Barrett ratio -- the same SHAPE of step as the product's pvw_arith.h;
    # libpvw_oracle_plain.so is the same restatement with every reduction done by the compiler's 128-bit `%`.  The
    # two must agree on everything the GPU parity tests use the oracle for (transforms, key generation, encrypt,
    # the decrypt inner products), on residues drawn from the whole range [0, q) including its ends.
    fast, plain = O.Oracle(moduli, l), O.Oracle(moduli, l, plain=True)
    assert fast._L is not plain._L
    rng = np.random.default_rng(n * 1000 + k)
    L = len(moduli)
    q = np.array(moduli, dtype=np.uint64)[None, None, :, None]

    def residues(rows, cols):
        m = np.stack([rng.integers(0, int(qq), size=(rows, cols, l), dtype=np.uint64) for qq in moduli], axis=2)
        m[0, 0] = 0
        m[-1, -1] = (q - np.uint64(1))[0, 0]
        return m

    a_hat, b_hat = residues(k, k), residues(n, k)
    polys = residues(3, 2).reshape(6, L, l)
    assert np.array_equal(fast.ntt_forward(polys), plain.ntt_forward(polys))
    assert np.array_equal(fast.ntt_inverse(polys), plain.ntt_inverse(polys))
    sk = O.sample_cbd(SEED, M.DOM_SK, 0, n * k, l, 0.5).reshape(n, k, l)
    ek = O.sample_uniform(SEED, M.DOM_EKEY, 0, n * k, l, 100).reshape(n, k, l)
    assert np.array_equal(fast.keygen(a_hat, sk, ek), plain.keygen(a_hat, sk, ek))
    r = O.sample_cbd(SEED, M.DOM_R, 0, k, l, 0.5)
    e1 = O.sample_uniform(SEED, M.DOM_E1, 0, k, l, 100)
    e2 = O.sample_uniform(SEED, M.DOM_E2, 0, n, l, 200)
    g_hat = fast.ntt_forward(np.stack([np.array([pow(int(M.Params(n, k, l, list(moduli)).delta), j, int(qq)) for j in range(l)],
                                                dtype=np.uint64) for qq in moduli])[None])[0]
    scalars = rng.integers(0, 1 << 64, size=n, dtype=np.uint64)
    c1f, c2f = fast.encrypt(a_hat, b_hat, g_hat, scalars, r, e1, e2)
    c1p, c2p = plain.encrypt(a_hat, b_hat, g_hat, scalars, r, e1, e2)
    assert np.array_equal(c1f, c1p) and np.array_equal(c2f, c2p)
    c1s = np.stack([c1f, c1p[::-1].copy()])
    c2col = np.stack([c2f[0], c2f[-1]])
    assert np.array_equal(fast.decrypt_noisy(sk[0], c1s, c2col), plain.decrypt_noisy(sk[0], c1s, c2col))
