"""Shared helpers for the tests: build small PVW systems with the big-integer model
and hand them over as RNS numpy arrays in the API layout ([..][L][l] u64)."""
import numpy as np

import pvw_model as M

TEST_MODULI = [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001]          # tests/crypto.rs:52
EXAMPLE_MODULI = [0x800000022A0001, 0x800000021A0001, 0x80000002120001, 0x80000001F60001]  # examples/pvw_valid_dec.rs:40-45
SEED = bytes([0x2A]) * 32                                         # tests/params.rs:91


def primes_1mod(step, count, top=(1 << 61)):
    """`count` primes p = 1 (mod step) descending from `top` (for ring degrees l > 32)."""
    out, p = [], top + 1
    while len(out) < count:
        p -= step
        if M.is_prime(p):
            out.append(p)
    return out


def ring_to_rns(poly_ints, moduli):
    """list of l ints mod Q -> [L][l] u64"""
    return np.array([[c % q for c in poly_ints] for q in moduli], dtype=np.uint64)


def rns_to_ring(arr, moduli):
    return M.from_rns([[int(v) for v in row] for row in arr], list(moduli))


def psi_list(moduli, l):
    return [M.minimal_primitive_root(q, 2 * l) for q in moduli]


def ntt_rows(arr_pb, moduli, l, psi=None):
    """[..][L][l] power-basis residues -> NTT domain by direct evaluation (model)."""
    psi = psi or psi_list(moduli, l)
    a = np.asarray(arr_pb, dtype=np.uint64)
    flat = a.reshape(-1, len(moduli), l)
    out = np.empty_like(flat)
    for p in range(flat.shape[0]):
        for i, q in enumerate(moduli):
            out[p, i] = M.ntt_eval([int(v) for v in flat[p, i]], q, psi[i])
    return out.reshape(a.shape)


def make_system(n, k, l, moduli, variance=0.5, bounds=None, seed=SEED, scalars=None):
    """A complete small system from the model, seeded ChaCha randomness.
    Returns dict with Params, ring-level (int) objects and RNS power-basis arrays."""
    if bounds is None:
        bounds = M.Params.suggest_error_bounds(n, k, l, moduli, variance)
    P = M.Params(n, k, l, moduli, variance, bounds[0], bounds[1])
    L = len(moduli)
    A = [[M.from_rns([M.sample_uniform_residues(q, l, M.ChaChaRng(seed, M.DOM_CRS, (i * k + j) * L + li))
                      for li, q in enumerate(moduli)], moduli) for j in range(k)] for i in range(k)]
    sk = [[M.sample_vec_cbd(l, variance, M.ChaChaRng(seed, M.DOM_SK, p * k + j)) for j in range(k)] for p in range(n)]
    ek = [[M.sample_uniform_coefficients(bounds[0], l, M.ChaChaRng(seed, M.DOM_EKEY, p * k + j)) for j in range(k)]
          for p in range(n)]
    B = [M.public_key(P, A, sk[p], ek[p]) for p in range(n)]
    r = [M.sample_vec_cbd(l, variance, M.ChaChaRng(seed, M.DOM_R, j)) for j in range(k)]
    e1 = [M.sample_uniform_coefficients(bounds[0], l, M.ChaChaRng(seed, M.DOM_E1, j)) for j in range(k)]
    e2 = [M.sample_uniform_coefficients(bounds[1], l, M.ChaChaRng(seed, M.DOM_E2, i)) for i in range(n)]
    if scalars is None:
        scalars = [(i * 1000 + 1) % (1 << 32) for i in range(n)]    # examples/pvw.rs:98-100 pattern
    c1, c2 = M.encrypt(P, A, B, scalars, r, e1, e2)
    return dict(
        P=P, A=A, B=B, sk=sk, ek=ek, r=r, e1=e1, e2=e2, scalars=scalars, c1=c1, c2=c2,
        A_pb=np.array([[ring_to_rns(A[i][j], moduli) for j in range(k)] for i in range(k)], dtype=np.uint64),
        B_pb=np.array([[ring_to_rns(B[i][j], moduli) for j in range(k)] for i in range(n)], dtype=np.uint64),
        c1_pb=np.array([ring_to_rns(c, moduli) for c in c1], dtype=np.uint64),
        c2_pb=np.array([ring_to_rns(c, moduli) for c in c2], dtype=np.uint64),
        g_pb=ring_to_rns(P.gadget_vector(), moduli),
    )
