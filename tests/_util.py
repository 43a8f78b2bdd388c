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


def decode_cases(l, moduli):
    """Inputs of the gadget decode (ring form: l integers mod Q each) for one parameter set: uniform values, ciphertext-shaped
    ones, and values on and around every bound the device decode's general path and its short cuts rely on."""
    rng = np.random.default_rng(l + len(moduli))
    m = M.Params(3, 4, l, moduli)
    Q, D = m.Q, m.delta
    cases = [[int.from_bytes(rng.bytes(Q.bit_length() // 8 + 8), "little") % Q for _ in range(l)] for _ in range(150)]
    for msg in (0, 1, 1000, 1001, -1, -1000, -1001, 2 ** 32, 2 ** 63, 2 ** 64 - 1, 2 ** 64):
        for amp in (0, 50, 10 ** 4):
            noise = [int(x) for x in rng.integers(-amp, amp + 1, size=l)]
            cases.append([(-(msg * D ** j) + noise[j]) % Q for j in range(l)])
    half = Q // 2
    for v in (0, 1, half - 1, half, half + 1, Q - 1, m.delta_power_l_minus_1 // 2, m.delta_power_l_minus_1 // 2 + 1,
              m.delta_power_l_minus_1, D, D // 2, D // 2 + 1):
        cases.append([v % Q] * l)
        cases.append([(v * (j + 1)) % Q for j in range(l)])
    # the short cut for noise-sized chain inputs (small_candidates / small_confirm) settles tmp_i = z_i*Delta - z_{i+1}
    # (and z_0) from the first 2..4 residues when the value fits half the product of those moduli, and confirms it against
    # every limb: inputs whose tmp_i sit on and around that bound for each possible count, either sign, in every / the
    # first / the last / alternating positions (a refused short cut sends the rest of that wave's inputs the long way)
    for nl in range(2, min(4, len(moduli) - 1) + 1):
        Pn = 1
        for q in moduli[:nl]:
            Pn *= q
        # ... and on the rounding boundaries of the division by Delta and the operand bound (2^191) of the chain's
        # short step (small_chain_step), whose quotient must fit one word (it does not for the larger of these values
        # when Delta is short: refused, the general step takes over)
        for v in (Pn // 2 - 1, Pn // 2, Pn // 2 + 1, Pn - 1, Pn, Pn + 1, 1, 2 ** 64, 2 ** 128 + 5, D // 2 - 1, D // 2, D // 2 + 1,
                  D, 3 * D // 2, 3 * D // 2 + 1, D * (2 ** 64 - 1), D * 2 ** 64 - D // 2 - 1, D * 2 ** 64 - D // 2, 2 ** 191 - 1, 2 ** 191, 2 ** 190):
            for sign in (1, -1):
                for where in ("all", "first", "last", "alternate", "z0"):
                    small = [int(x) for x in rng.integers(-1000, 1001, size=l)]
                    tm = [sign * v if where == "all" or (where == "first" and i == 0) or (where == "last" and i == l - 2) or
                          (where == "alternate" and i % 8 >= 4) else small[i] for i in range(l - 1)]
                    z = [(sign * v if where == "z0" else small[l - 1]) % Q]
                    for i in range(l - 1):
                        z.append((z[i] * D - tm[i]) % Q)
                    cases.append(z)
    # noise_{l-1} without the Horner value's lift (small_top): ciphertext-shaped inputs z_j = -m Delta^j + n_j whose top
    # noise n_{l-1} sits on the rounding boundaries of Delta, and whose n_0 (the multiple of Delta^(l-1) the proof finds on
    # every limb) sits on and beyond what one limb can carry
    q0 = moduli[0]
    for top_noise in (0, 1, -1, D // 2 - 1, D // 2, D // 2 + 1, -(D // 2 - 1), -(D // 2), -(D // 2 + 1), 2 ** 100, -(2 ** 100)):
        for n0 in (0, 5, -7, q0 // 2 - 1, q0 // 2, q0 // 2 + 1, -(q0 // 2), -(q0 // 2 + 1), q0, 2 ** 61, 2 ** 64 + 3):
            msg = int(rng.integers(0, 2 ** 63))
            noise = [int(x) for x in rng.integers(-5000, 5001, size=l)]
            noise[0], noise[l - 1] = n0, top_noise
            cases.append([(-(msg * D ** j) + noise[j]) % Q for j in range(l)])
    return cases
