"""PVW multi-receiver encrypt/decrypt -- independent big-integer model.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.

This is a CPU restatement of the hot path of gnosisguild/pvw-rs (citations are
``file:line`` under the reference checkout) written from the mathematics:
every ring element is a list of ``l`` Python integers mod Q, the ring product is
a plain negacyclic schoolbook convolution in Z_Q[X]/(X^l+1) -- no NTT, no RNS --
so it is independent of any choice of root of unity or slot order and of the
C oracle / HIP kernels, which work limb-wise in the NTT domain.

PARITY STATUS.  The reference's arithmetic lives in fhe-math / fhe-util /
fhe-traits 0.1.0-beta.7 (git dependency gnosisguild/fhe.rs @ 3643350, branch
refactor/pvw-compat, Cargo.lock:294-335) which is NOT vendored in the reference
checkout, and no Rust toolchain exists in this pipeline, so the reference cannot
be run.  Its tests hold no golden vectors; they hold properties.  This model is
pinned against those properties (tests/test_oracle_model.py):
  * encrypt -> decrypt round trip            tests/crypto.rs:237-305
  * gadget = [1, D, ..., D^(l-1)]             tests/crypto.rs:17-37, tests/params.rs:638-674
  * bigints_to_poly <-> CRT lift round trips  tests/params.rs:485-635
  * from_coefficients(i64) == bigints_to_poly tests/params.rs:733-767
  * rounding-division truth table             tests/crypto.rs:308-330
  * CBD support / mean / variance             tests/sampling.rs:198-274
Ring-level (PowerBasis) results are therefore pinned by the reference's own
test properties.  PARITY UNPINNED for: the NTT-domain slot order and the 2l-th
root psi that fhe-math picks, the bytes of Poly::random_from_seed, and every
sampled stream (the reference draws from thread_rng(), encryption.rs:138,164,180).
"""
from __future__ import annotations

import math
import struct
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

MASK32 = 0xFFFFFFFF
MASK64 = 0xFFFFFFFFFFFFFFFF


# --------------------------------------------------------------------------
# number theory helpers
# --------------------------------------------------------------------------
def is_prime(n: int) -> bool:
    """Deterministic Miller-Rabin for n < 2^64."""
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in small:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in small:
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


def bitrev(i: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def minimal_primitive_root(q: int, order: int) -> int:
    """Smallest primitive ``order``-th root of unity mod prime q (order = 2l, a power of two).

    This is THIS BUILD's deterministic rule for psi (the reference delegates the
    choice to fhe-math, whose source is not available: parity unpinned)."""
    assert (q - 1) % order == 0
    exp = (q - 1) // order
    g = 2
    while True:
        w = pow(g, exp, q)
        if pow(w, order // 2, q) == q - 1:
            break
        g += 1
    # all primitive roots are the odd powers of w; take the smallest
    best, cur, w2 = w, w, w * w % q
    for _ in range(order // 2 - 1):
        cur = cur * w2 % q
        if cur < best:
            best = cur
    return best


def iroot(x: int, n: int) -> int:
    """floor(x ** (1/n)) -- BigUint::nth_root (parameters.rs:156)."""
    if x < 2:
        return x
    hi = 1 << ((x.bit_length() + n - 1) // n)
    lo = 0
    while lo < hi:  # invariant: lo^n <= x < (hi+1)^n
        mid = (lo + hi + 1) >> 1
        if mid ** n <= x:
            lo = mid
        else:
            hi = mid - 1
    return lo


def tdiv(a: int, b: int) -> int:
    """num-bigint BigInt '/' : truncation toward zero (decryption.rs:159,191,194)."""
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def trem(a: int, b: int) -> int:
    """num-bigint BigInt '%' : remainder with the sign of the dividend."""
    return a - b * tdiv(a, b)


def big_to_f64(x: int) -> float:
    """BigUint/BigInt::to_f64: correctly rounded, saturating to +-inf (parameters.rs:518,547)."""
    try:
        return float(x)
    except OverflowError:
        return math.inf if x > 0 else -math.inf


def bench_moduli(count: int) -> List[int]:
    """The synthetic modulus chain of SURVEY.md 8(d): the first ``count`` primes found
    descending from 2^61 in steps of 64 with p = 1 (mod 64)."""
    out, p = [], (1 << 61) + 1
    while len(out) < count:
        p -= 64
        if is_prime(p):
            out.append(p)
    return out


# --------------------------------------------------------------------------
# parameters (src/params/parameters.rs)
# --------------------------------------------------------------------------
class PvwError(Exception):
    pass


@dataclass
class Params:
    """PvwParameters (parameters.rs:19-40) built with the builder's rules (:117-195)."""
    n: int
    k: int
    l: int
    moduli: Sequence[int]
    secret_variance: float = 0.5      # builder default, parameters.rs:166
    error_bound_1: int = 100          # :167
    error_bound_2: int = 200          # :168
    Q: int = field(init=False)
    delta: int = field(init=False)
    delta_power_l_minus_1: int = field(init=False)
    t: int = field(init=False)

    def __post_init__(self):
        if self.n == 0:
            raise PvwError("n must be > 0")                      # :132
        if self.k == 0:
            raise PvwError("k must be > 0")                      # :135
        l = self.l
        if l < 8 or (l & (l - 1)) != 0:
            raise PvwError("l must be power of 2 and >= 8")      # :140
        # Context::new_arc (:147) lives in fhe-math; this build states its own
        # conditions: distinct odd primes < 2^62 with q = 1 (mod 2l).
        if len(self.moduli) == 0:
            raise PvwError("moduli not set")
        if len(set(self.moduli)) != len(self.moduli):
            raise PvwError("moduli must be distinct")
        for q in self.moduli:
            if q >= (1 << 62) or q < 3 or not is_prime(q) or (q - 1) % (2 * l) != 0:
                raise PvwError(f"modulus {q:#x} unsupported")
        if self.error_bound_1 <= 0:
            raise PvwError("error_bound_1 must be positive")     # :172
        if self.error_bound_2 <= 0:
            raise PvwError("error_bound_2 must be positive")     # :177
        self.moduli = list(self.moduli)
        Q = 1
        for q in self.moduli:
            Q *= q
        self.Q = Q
        self.delta = iroot(Q, l)                                  # :156
        self.delta_power_l_minus_1 = self.delta ** (l - 1)        # :159-163
        self.t = (self.n - 1) // 2                                # :169

    # -- gadget / encode -------------------------------------------------
    def gadget_vector(self) -> List[int]:
        """[1, D, ..., D^(l-1)] (parameters.rs:288-324)."""
        return [self.delta ** j for j in range(self.l)]

    def encode_scalar(self, scalar_i64: int) -> List[int]:
        """Power-basis coefficients scalar*D^j mod Q (parameters.rs:346-367)."""
        return [(scalar_i64 * self.delta ** j) % self.Q for j in range(self.l)]

    # -- correctness gate ---------------------------------------------------
    def correctness_bound(self) -> float:
        n, k, l = float(self.n), float(self.k), float(self.l)
        b1 = big_to_f64(self.error_bound_1)
        b2 = big_to_f64(self.error_bound_2)
        first = b2 * math.sqrt(n * l) * (1.0 + math.sqrt(n))
        second = 2.0 * b1 * k * l
        third = 14.0 * b1 * math.sqrt(n * k * l)
        return first + second + third

    def verify_correctness_condition(self) -> bool:
        """parameters.rs:510-551 (f64 arithmetic, D^(l-1) saturating to +inf)."""
        return big_to_f64(self.delta_power_l_minus_1) > self.correctness_bound()

    @staticmethod
    def suggest_error_bounds(n, k, l, moduli, variance) -> Tuple[int, int]:
        """parameters.rs:554-603."""
        tmp = Params(n, k, l, moduli, variance, 1, 1)
        dp = big_to_f64(tmp.delta_power_l_minus_1)
        nf, kf, lf = float(n), float(k), float(l)
        c1 = 2.0 * kf * lf + 14.0 * math.sqrt(nf * kf * lf)
        c2 = math.sqrt(nf * lf) * (1.0 + math.sqrt(nf))
        for b1 in (50, 100, 200, 500, 1000, 2000):
            for b2 in (50, 100, 200, 500, 1000, 2000):
                if dp > b1 * c1 + b2 * c2:
                    return b1, b2
        raise PvwError("Cannot find suitable error bounds")


def u64_as_i64(x: int) -> int:
    """`scalars[i] as i64` wrap (encryption.rs:195)."""
    x &= MASK64
    return x - (1 << 64) if x >> 63 else x


# --------------------------------------------------------------------------
# RNS <-> integer (parameters.rs:420-474; fhe-math Vec<BigUint>::from(&Poly))
# --------------------------------------------------------------------------
def to_rns(coeffs: Sequence[int], moduli: Sequence[int]) -> List[List[int]]:
    """bigints_to_poly: row = limb, col = coefficient, residue ((c % q) + q) % q."""
    return [[c % q for c in coeffs] for q in moduli]


def from_rns(rows: Sequence[Sequence[int]], moduli: Sequence[int]) -> List[int]:
    """CRT lift to [0, Q)."""
    Q = 1
    for q in moduli:
        Q *= q
    out = [0] * len(rows[0])
    for q, row in zip(moduli, rows):
        Qi = Q // q
        inv = pow(Qi, -1, q)
        for c, v in enumerate(row):
            out[c] = (out[c] + v * inv % q * Qi) % Q
    return out


# --------------------------------------------------------------------------
# ring arithmetic in Z_Q[X]/(X^l + 1), power basis
# --------------------------------------------------------------------------
def ring_mul(a: Sequence[int], b: Sequence[int], Q: int) -> List[int]:
    l = len(a)
    out = [0] * l
    for i, ai in enumerate(a):
        if ai == 0:
            continue
        for j, bj in enumerate(b):
            if i + j < l:
                out[i + j] += ai * bj
            else:
                out[i + j - l] -= ai * bj
    return [v % Q for v in out]


def ring_add(a, b, Q):
    return [(x + y) % Q for x, y in zip(a, b)]


def ring_sub(a, b, Q):
    return [(x - y) % Q for x, y in zip(a, b)]


def ntt_eval(a: Sequence[int], q: int, psi: int) -> List[int]:
    """Negacyclic NTT of one limb by direct evaluation: slot s holds a(psi^(2*br(s)+1)).

    This is the slot convention of THIS build (bit-reversed output of the usual
    in-place Cooley-Tukey network); it is used to produce NTT-domain fixtures."""
    l = len(a)
    bits = l.bit_length() - 1
    out = []
    for s in range(l):
        x = pow(psi, 2 * bitrev(s, bits) + 1, q)
        acc, xp = 0, 1
        for c in a:
            acc = (acc + c * xp) % q
            xp = xp * x % q
        out.append(acc)
    return out


# --------------------------------------------------------------------------
# ChaCha counter-based RNG (this build's reproducible randomness)
# --------------------------------------------------------------------------
def _rotl(x, n):
    return ((x << n) | (x >> (32 - n))) & MASK32


def chacha_block(key_words: Sequence[int], counter: int, stream: int, rounds: int = 8) -> List[int]:
    """One 64-byte ChaCha block as 16 little-endian u32 words.  State layout of
    rand_chacha: constants | key(8) | 64-bit block counter (words 12,13) |
    64-bit stream id (words 14,15)."""
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [
        counter & MASK32, (counter >> 32) & MASK32, stream & MASK32, (stream >> 32) & MASK32]
    x = list(st)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & MASK32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & MASK32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & MASK32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & MASK32; x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & MASK32 for a, b in zip(x, st)]


# stream-id domains: stream = (domain << 32) | polynomial index
DOM_R, DOM_E1, DOM_E2, DOM_SK, DOM_EKEY, DOM_CRS, DOM_GAUSS, DOM_PK = 0, 1, 2, 3, 4, 5, 6, 7


class ChaChaRng:
    """ChaCha8 word stream for one polynomial: key = 32-byte seed, stream id =
    (domain << 32) | index, block counter from 0.  next_u32 / next_u64 hand out
    the block words in order (u64 = low word first), like rand_chacha's BlockRng."""

    def __init__(self, seed: bytes, domain: int, index: int, rounds: int = 8):
        assert len(seed) == 32
        self.key = struct.unpack("<8I", seed)
        self.stream = ((domain & MASK32) << 32) | (index & MASK32)
        self.rounds = rounds
        self.counter = 0
        self.buf: List[int] = []

    def next_u32(self) -> int:
        if not self.buf:
            self.buf = chacha_block(self.key, self.counter, self.stream, self.rounds)
            self.counter += 1
        return self.buf.pop(0)

    def next_u64(self) -> int:
        lo = self.next_u32()
        hi = self.next_u32()
        return lo | (hi << 32)


def sample_vec_cbd(size: int, variance: float, rng: ChaChaRng) -> List[int]:
    """Centered binomial samples, bit consumption as in sampling/uniform.rs:27-70."""
    if not (0.5 <= variance <= 16.0):
        raise PvwError("The variance should be between 0.5 and 16")       # :32
    out = []
    if abs(variance - 0.5) < 1.1920929e-07:                              # f32::EPSILON, :38
        for _ in range(size):
            b1 = rng.next_u32() & 1
            b2 = rng.next_u32() & 1
            out.append(b1 - b2)
        return out
    v = int(variance)                                                     # `as usize`, :47
    if v < 1:
        # 0.5 < variance < 1 truncates to 0 bits in the reference (shift overflow);
        # this build rejects it.
        raise PvwError("non-integer variance below 1 is not supported")
    nbits = 4 * v
    mask_add = ((MASK64 >> (64 - nbits)) >> (2 * v))
    mask_sub = mask_add << (2 * v)
    pool, pool_n = 0, 0
    for _ in range(size):
        if pool_n < nbits:
            pool |= rng.next_u64() << pool_n
            pool_n += 64
        out.append(bin(pool & mask_add).count("1") - bin(pool & mask_sub).count("1"))
        pool >>= nbits
        pool_n -= nbits
    return out


def sample_uniform_coefficients(bound: int, count: int, rng: ChaChaRng) -> List[int]:
    """Uniform integers in [-bound, bound] (sampling/uniform.rs:5-22): rejection
    sampling of a bit_length(2*bound+1)-bit value built from u32 words, the top
    word shifted down -- the shape of num-bigint's gen_biguint_below."""
    rng_range = 2 * bound + 1
    bits = rng_range.bit_length()
    digits, rem = divmod(bits, 32)
    out = []
    for _ in range(count):
        while True:
            words = [rng.next_u32() for _ in range(digits + (1 if rem else 0))]
            if rem:
                words[-1] >>= 32 - rem
            v = 0
            for i, w in enumerate(words):
                v |= w << (32 * i)
            if v < rng_range:
                break
        out.append(v - bound)
    return out


def sample_uniform_residues(q: int, count: int, rng: ChaChaRng) -> List[int]:
    """Uniform residues in [0, q) by 64-bit rejection (used for synthetic A-hat / B-hat)."""
    bits = q.bit_length()
    out = []
    while len(out) < count:
        v = rng.next_u64() >> (64 - bits)
        if v < q:
            out.append(v)
    return out


# --------------------------------------------------------------------------
# keys (src/keys/public_key.rs:111-147, src/params/crs.rs:138-171)
# --------------------------------------------------------------------------
def small_to_ring(coeffs: Sequence[int], Q: int) -> List[int]:
    """Poly::from_coefficients(&[i64]) == bigints_to_poly: non-negative residue."""
    return [c % Q for c in coeffs]


def public_key(params: Params, A, sk: Sequence[Sequence[int]], e: Sequence[Sequence[int]]):
    """b[c] = sum_j sk[j] * A[j][c] + e[c]  (crs.rs:152-168, public_key.rs:134-139)."""
    Q, k = params.Q, params.k
    skq = [small_to_ring(s, Q) for s in sk]
    out = []
    for c in range(k):
        acc = [0] * params.l
        for j in range(k):
            acc = ring_add(acc, ring_mul(skq[j], A[j][c], Q), Q)
        out.append(ring_add(acc, small_to_ring(e[c], Q), Q))
    return out


# --------------------------------------------------------------------------
# encrypt (src/crypto/encryption.rs:105-214) with explicit randomness
# --------------------------------------------------------------------------
def encrypt(params: Params, A, B, scalars: Sequence[int], r, e1, e2):
    """c1[i] = sum_j A[i][j]*r[j] + e1[i]   (crs.rs:188-201, encryption.rs:171-173)
       c2[i] = sum_j B[i][j]*r[j] + encode(scalars[i] as i64) + e2[i]  (:177-200)."""
    if len(scalars) != params.n:
        raise PvwError(f"Must provide exactly n={params.n} scalars, got {len(scalars)}")  # :109
    if len(B) < params.n:
        raise PvwError("Global public key is not complete")                                # :117
    if not params.verify_correctness_condition():
        raise PvwError("Parameters do not satisfy correctness condition")                  # :124
    Q, k, l = params.Q, params.k, params.l
    rq = [small_to_ring(x, Q) for x in r]
    c1 = []
    for i in range(k):
        acc = [0] * l
        for j in range(k):
            acc = ring_add(acc, ring_mul(A[i][j], rq[j], Q), Q)
        c1.append(ring_add(acc, small_to_ring(e1[i], Q), Q))
    c2 = []
    for i in range(params.n):
        acc = [0] * l
        for j in range(k):
            acc = ring_add(acc, ring_mul(B[i][j], rq[j], Q), Q)
        acc = ring_add(acc, params.encode_scalar(u64_as_i64(scalars[i])), Q)
        c2.append(ring_add(acc, small_to_ring(e2[i], Q), Q))
    return c1, c2


# --------------------------------------------------------------------------
# decrypt + decode (src/crypto/decryption.rs)
# --------------------------------------------------------------------------
def center(v: int, Q: int) -> int:
    """center_coefficient_with_precision (decryption.rs:140-152)."""
    return v - Q if v > Q // 2 else v


def decode_scalar_pvw(noisy: Sequence[int], params: Params) -> int:
    """decode_scalar_pvw_rns (decryption.rs:10-58).  Every polynomial in the
    reference routine is a constant, so this is integer arithmetic mod Q with a
    centred lift wherever the reference calls extract_constant_term_bigint."""
    Q, l, D = params.Q, params.l, params.delta
    z = [center(v % Q, Q) for v in noisy]                       # :109-137
    tmp = [(z[i] * D - z[i + 1]) % Q for i in range(l - 1)]     # :19-27
    last = tmp[0]
    for i in range(1, l - 1):                                   # :30-33 Horner
        last = (last * D + tmp[i]) % Q
    # reduce_modulo_poly (:154-178)
    poly_const = center(last, Q)
    mod_const = center(params.delta_power_l_minus_1 % Q, Q)
    reduced = trem(poly_const, mod_const)
    half = tdiv(mod_const, 2)
    if reduced > half:
        reduced -= mod_const
    elif reduced < -half:
        reduced += mod_const
    tmp.append(reduced % Q)
    noise = [0] * l
    noise[l - 1] = tmp[l - 1]
    delta_const = center(D % Q, Q)
    for i in range(l - 2, -1, -1):                              # :44-48
        p = center((noise[i + 1] - tmp[i]) % Q, Q)              # divide_by_delta_rns :180-207
        if delta_const == 0:
            quo = 0
        elif p < 0:
            quo = tdiv(2 * p - delta_const, 2 * delta_const)
        else:
            quo = tdiv(2 * p + delta_const, 2 * delta_const)
        noise[i] = quo % Q
    plain = center((-z[0] - noise[0]) % Q, Q)                   # :51-53
    # extract_constant_term_as_u64 (:226-247)
    if plain < 0:
        if -plain <= 1000:
            return 0
        pos = trem(plain + Q, Q)
        return pos if 0 <= pos < (1 << 64) else 0
    return plain if plain < (1 << 64) else 0


def decrypt_noisy(params: Params, c1, c2_i, sk) -> List[int]:
    """noisy = <sk, c1> - c2[i]  (decryption.rs:257-274)."""
    Q = params.Q
    acc = [0] * params.l
    for j in range(params.k):
        acc = ring_add(acc, ring_mul(small_to_ring(sk[j], Q), c1[j], Q), Q)
    return ring_sub(acc, c2_i, Q)


def decrypt_party_value(params: Params, c1, c2_i, sk) -> int:
    """decrypt_party_value (decryption.rs:249-278)."""
    return decode_scalar_pvw(decrypt_noisy(params, c1, c2_i, sk), params)


# --------------------------------------------------------------------------
# truncated discrete Gaussian (src/sampling/normal.rs:136-190) -- off the encrypt path
# --------------------------------------------------------------------------
TAIL_STDDEV_MULTIPLIER = 16.96


def _unit_f64(rng: ChaChaRng) -> float:
    """53-bit uniform in [0,1) from one u64 (the standard rand 'Standard' f64 shape)."""
    return (rng.next_u64() >> 11) * (1.0 / (1 << 53))


def box_muller(rng: ChaChaRng) -> float:
    """normal.rs:186-190 with u1 in [EPSILON, 1), u2 in [0, 1)."""
    eps = 2.220446049250313e-16
    u1 = eps + (1.0 - eps) * _unit_f64(rng)
    u2 = _unit_f64(rng)
    return math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * math.pi * u2)


def sample_single_gaussian(bound: int, rng: ChaChaRng) -> int:
    """normal.rs:136-162 (ratio ~ N(0, sigma^2) truncated to [-1,1], sigma = bound/16.96
    taken as an absolute number; sigma > 0.3 => uniform ratio; bound > 1e15 => +-[0,1e6])."""
    if bound == 0:
        return 0
    bf = big_to_f64(bound)
    if bf > 1e15:
        sign = 1 if (rng.next_u32() >> 31) else -1
        return sign * (rng.next_u32() % 1000001)
    sigma = bf / TAIL_STDDEV_MULTIPLIER
    ratio = None
    if sigma > 0.3:
        ratio = 2.0 * _unit_f64(rng) - 1.0
    else:
        for _ in range(1000):
            r = box_muller(rng) * sigma
            if -1.0 <= r <= 1.0:
                ratio = r
                break
        if ratio is None:
            ratio = 2.0 * _unit_f64(rng) - 1.0
    fx = ratio * bf                      # f64::round = half away from zero (normal.rs:201)
    x = int(math.floor(abs(fx) + 0.5)) * (1 if fx >= 0 else -1)
    return max(-bound, min(bound, x))
