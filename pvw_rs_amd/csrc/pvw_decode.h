// pvw_decode.h -- the PVW gadget decode (src/crypto/decryption.rs:10-247) as fixed-width
// integer arithmetic that runs per thread on the GPU (and, for self-tests, on the host).
//
// The reference routine dresses an integer algorithm in constant polynomials.  Everything it
// does "mod Q" is done here limb-wise in RNS straight from the noisy residues (centring does not
// change a residue mod q_i), and a positional big integer is materialised only where the
// reference calls extract_constant_term_bigint (decryption.rs:209-224) to compare or divide:
//   * once for the Horner value that is reduced modulo Delta^(l-1)        (:30-37, :154-178)
//   * once per noise component for the rounded division by Delta          (:44-48, :180-207)
//   * once for the final plaintext                                        (:51-55, :226-247)
// i.e. l+1 CRT lifts and l conversions back to RNS per ciphertext, O(L*W) word operations each.
//
// Big integers are W 64-bit words (W = words(Q) + 1) stored with a stride, so that on the device
// word j of thread t sits at lds[j*64 + t] (bank-conflict free) and on the host stride = 1.
#pragma once
#include "pvw_arith.h"

namespace pvw {

struct BN {
  u64* p;
  int stride;
  PVW_HD u64& operator[](int i) const { return p[(size_t)i * stride]; }
};

// per-context constants of the decode, all device pointers in kernels
struct DecodeTables {
  u32 W;              // words per big integer
  u32 L, ell;
  const Mod* mods;    // [L]
  const u64* Q;       // [W]
  const u64* halfQ;   // [W] floor(Q/2)
  const u64* qi;      // [L][W]  Q / q_i
  const u64* inv;     // [L]     (Q/q_i)^-1 mod q_i
  const u64* invp;    // [L]     Shoup companion
  const u64* pow64;   // [L][W]  2^(64 j) mod q_i
  const u64* dmod;    // [L]     Delta mod q_i
  const u64* dmodp;   // [L]     Shoup companion
  const u64* delta;   // [W]     Delta
  const u64* dpow;    // [W]     Delta^(l-1)
  const u64* half_dpow;  // [W]  floor(Delta^(l-1) / 2)
  // normalised divisors for Knuth D: value << shift, nwords significant words
  const u64* dpow_n;  u32 dpow_nw, dpow_sh;
  const u64* td_n;    u32 td_nw, td_sh;      // 2*Delta
  // wave-cooperative kernel (one lane per word / per limb): transposed power table and Barrett-style
  // reciprocals floor(B^(W+1) / d), B = 2^64, for d = 2*Delta and d = Delta^(l-1)
  const u64* pow64T;  // [W][L]
  const u64* td;      // [W+2]  2*Delta
  const u64* mu_td;   // [W+2]
  const u64* mu_dp;   // [W+2]
  // short-cut lift of small values (pvw_decode_wave.h: wave_lift_small): Garner constants over the first gar_n <= 4 moduli
  // (0 = not used).  inv[j][i] = q_i^-1 mod q_j (i < j) at [4 j + i] | their Shoup companions at [16 + 4 j + i] |
  // prod[j] = q_0 .. q_{j-1} as 4 words at [32 + 4 j], j = 0..4 | floor(prod[gar_n] / 2) at [52]
  u32 gar_n;
  u32 gar_close;      // q_i < 2 q_j for all i < j < gar_n: a mixed-radix digit reduces to the next modulus with one subtraction
  const u64* gar;     // [56]
  // chain steps on noise-sized operands (pvw_decode_wave.h: small_chain_step), used when sc_on: Q has at least 194 bits
  // and 2*Delta at most three words.  [0..2] 2*Delta shifted left until bit 191 is set | [3] Moeller-Granlund reciprocal
  // of its top word, floor((2^128 - 1) / d2) - 2^64 | [4] whole words of that shift | [5] bits of it | [6..8] Delta
  u32 sc_on;
  const u64* sc;      // [9]
  // Delta^(l-1) mod q_i and its inverse, each with its Shoup companion: [4][L] = value | companion | inverse | companion.
  // The Horner value telescopes, sum_i tmp_i Delta^(l-2-i) = z_0 Delta^(l-1) - z_{l-1}; the inverse serves the short cut
  // for noise_{l-1} (hs_on: every limb's inverse exists, Delta has at least 64 bits, sc_on)
  u32 hs_on;
  const u64* dpm;     // [4][L]
};

PVW_HD void bn_zero(BN a, int W) { for (int i = 0; i < W; ++i) a[i] = 0; }
PVW_HD void bn_load(BN a, const u64* src, int W) { for (int i = 0; i < W; ++i) a[i] = src[i]; }
PVW_HD int bn_cmp_c(BN a, const u64* b, int W) {   // compare with a constant
  for (int i = W - 1; i >= 0; --i) {
    u64 x = a[i], y = b[i];
    if (x != y) return x < y ? -1 : 1;
  }
  return 0;
}
PVW_HD bool bn_is_zero(BN a, int W) {
  u64 o = 0;
  for (int i = 0; i < W; ++i) o |= a[i];
  return o == 0;
}
PVW_HD void bn_sub_c(BN a, const u64* b, int W) {   // a -= b  (a >= b)
  u64 borrow = 0;
  for (int i = 0; i < W; ++i) {
    u64 x = a[i], y = b[i];
    u64 d = x - y - borrow;
    borrow = (x < y) || (x == y && borrow) ? 1 : 0;
    a[i] = d;
  }
}
PVW_HD void bn_rsub_c(BN a, const u64* b, int W) {  // a = b - a  (b >= a)
  u64 borrow = 0;
  for (int i = 0; i < W; ++i) {
    u64 x = b[i], y = a[i];
    u64 d = x - y - borrow;
    borrow = (x < y) || (x == y && borrow) ? 1 : 0;
    a[i] = d;
  }
}
// a += x * t   (x constant, t a word); W words, the top carry is dropped (callers size W for it)
PVW_HD void bn_addmul_c(BN a, const u64* x, u64 t, int W) {
  u64 carry = 0;
  for (int i = 0; i < W; ++i) {
    u128 s = (u128)x[i] * t + a[i] + carry;
    a[i] = (u64)s;
    carry = (u64)(s >> 64);
  }
}
// |a| mod q via the table pow64[j] = 2^(64 j) mod q  (one lazy accumulation, one reduction)
PVW_HD u64 bn_mod_small(BN a, const u64* pow64, const Mod& m, int W) {
  Acc acc;
  acc_zero(acc);
  for (int j = 0; j < W; ++j) acc_mac(acc, a[j], pow64[j]);
  return acc_reduce(acc, m);
}

// Knuth algorithm D.  u holds W+1 words (u[W] is scratch for the normalisation shift): on entry the
// W-word dividend, on exit the remainder.  vn = divisor << sh with its top bit set, nw >= 2
// significant words.  Quotient -> q (W words).
PVW_HD void bn_divmod_knuth(BN u, BN q, const u64* vn, int nw, int sh, int W) {
  u[W] = 0;
  if (sh) {
    for (int i = W; i >= 0; --i) {
      u64 lo = i ? u[i - 1] : 0;
      u[i] = (u[i] << sh) | (lo >> (64 - sh));
    }
  }
  const u64 v1 = vn[nw - 1], v2 = vn[nw - 2];
  for (int i = 0; i < W; ++i) q[i] = 0;
  for (int j = W - nw; j >= 0; --j) {
    const u64 uh = u[j + nw], u1 = u[j + nw - 1], u0 = u[j + nw - 2];
    const u128 num = ((u128)uh << 64) | u1;
    u128 qhat, rhat;
    if (uh >= v1) {          // quotient digit would not fit a word: start from B-1
      qhat = ~(u64)0;
      rhat = num - qhat * v1;
    } else {
      qhat = num / v1;
      rhat = num - qhat * v1;
    }
    while ((rhat >> 64) == 0 && qhat * v2 > ((rhat << 64) | u0)) {
      --qhat;
      rhat += v1;
    }
    u64 borrow = 0, carry = 0;
    for (int i = 0; i < nw; ++i) {      // u[j .. j+nw-1] -= qhat * v
      u128 p = (u128)(u64)qhat * vn[i] + carry;
      carry = (u64)(p >> 64);
      u64 x = u[j + i], y = (u64)p;
      u[j + i] = x - y - borrow;
      borrow = (x < y) || (x == y && borrow) ? 1 : 0;
    }
    {
      u64 x = uh, y = carry;
      u[j + nw] = x - y - borrow;
      borrow = (x < y) || (x == y && borrow) ? 1 : 0;
    }
    if (borrow) {            // qhat was one too large: add the divisor back
      --qhat;
      u64 c = 0;
      for (int i = 0; i < nw; ++i) {
        u128 t = (u128)u[j + i] + vn[i] + c;
        u[j + i] = (u64)t;
        c = (u64)(t >> 64);
      }
      u[j + nw] += c;
    }
    q[j] = (u64)qhat;
  }
  if (sh) {
    for (int i = 0; i < W; ++i) u[i] = (u[i] >> sh) | (u[i + 1] << (64 - sh));
  }
  u[W] = 0;
}
// short division by a single word d (tiny parameter sets): u -> remainder in u[0], quotient -> q
PVW_HD void bn_divmod_word(BN u, BN q, u64 d, int W) {
  u64 rem = 0;
  for (int i = W - 1; i >= 0; --i) {
    u128 cur = ((u128)rem << 64) | u[i];
    u64 qq = (u64)(cur / d);
    rem = (u64)(cur - (u128)qq * d);
    q[i] = qq;
  }
  for (int i = 1; i < W; ++i) u[i] = 0;
  u[0] = rem;
}
PVW_HD void bn_divmod(BN u, BN q, const u64* vn, int nw, int sh, int W) {
  if (nw == 1) bn_divmod_word(u, q, vn[0] >> sh, W);   // vn is stored normalised: undo the shift
  else bn_divmod_knuth(u, q, vn, nw, sh, W);
}

// CRT lift of the residues res(limb) to x in [0, Q), then centring: returns true if the centred
// value is negative, with |value| left in x.
template <class ResidueFn>
PVW_HD bool lift_centered(const DecodeTables& t, BN x, ResidueFn res) {
  const int W = (int)t.W;
  bn_zero(x, W);
  for (u32 i = 0; i < t.L; ++i) {
    const Mod m = t.mods[i];
    u64 ti = mulmod_shoup(res(i), t.inv[i], t.invp[i], m.q);
    bn_addmul_c(x, t.qi + (size_t)i * W, ti, W);
  }
  while (bn_cmp_c(x, t.Q, W) >= 0) bn_sub_c(x, t.Q, W);          // sum < L*Q
  if (bn_cmp_c(x, t.halfQ, W) > 0) {                               // decryption.rs:145-151
    bn_rsub_c(x, t.Q, W);
    return true;
  }
  return false;
}

// decode_scalar_pvw_rns for one ciphertext.  noisy: [L][l] power-basis residues of this dealer.
// x: (W+1)-word and y: W-word big-integer scratch; nres: L-word scratch (residues of the current noise).
PVW_HD u64 decode_one_fixed(const DecodeTables& t, const u64* noisy, BN x, BN y, BN nres) {
  const int W = (int)t.W;
  const u32 L = t.L, l = t.ell;
  // tmp_i = z_i * Delta - z_{i+1}  (mod q_limb), straight from the noisy residues (:19-27)
  auto tmp = [&](u32 limb, u32 i) -> u64 {
    const Mod m = t.mods[limb];
    const u64* z = noisy + (size_t)limb * l;
    return submod(mulmod_shoup(z[i], t.dmod[limb], t.dmodp[limb], m.q), z[i + 1], m.q);
  };
  // Horner over tmp_0 .. tmp_{l-2} (:30-33), lifted and centred
  bool neg = lift_centered(t, x, [&](u32 limb) -> u64 {
    const Mod m = t.mods[limb];
    u64 r = tmp(limb, 0);
    for (u32 i = 1; i + 1 < l; ++i) r = addmod(mulmod_shoup(r, t.dmod[limb], t.dmodp[limb], m.q), tmp(limb, i), m.q);
    return r;
  });
  // reduce_modulo_poly (:154-178): truncated remainder by Delta^(l-1), then re-centre
  bn_divmod(x, y, t.dpow_n, (int)t.dpow_nw, (int)t.dpow_sh, W);   // x = |poly_const| % mod_const
  if (bn_is_zero(x, W)) neg = false;
  if (bn_cmp_c(x, t.half_dpow, W) > 0) {     // reduced > half (positive)  or  reduced < -half (negative)
    bn_rsub_c(x, t.dpow, W);                 // magnitude becomes mod_const - |reduced| ...
    neg = !neg;                              // ... with the opposite sign
  }
  // residues of noise[l-1] = reduced
  for (u32 limb = 0; limb < L; ++limb) {
    const Mod m = t.mods[limb];
    u64 r = bn_mod_small(x, t.pow64 + (size_t)limb * W, m, W);
    nres[(int)limb] = (neg && r) ? m.q - r : r;
  }
  // noise[i] = round((noise[i+1] - tmp[i]) / Delta), i = l-2 .. 0   (:44-48, :180-207)
  for (u32 i = l - 1; i-- > 0;) {
    bool pneg = lift_centered(t, x, [&](u32 limb) -> u64 { return submod(nres[(int)limb], tmp(limb, i), t.mods[limb].q); });
    // |quotient| = floor((2|p| + Delta) / (2 Delta)); sign = sign(p)   (truncating BigInt division)
    u64 carry = 0;
    for (int w = 0; w < W; ++w) {            // x = 2x + Delta
      u64 v = x[w];
      u128 s = (u128)(v << 1) + (carry) + t.delta[w];
      carry = (v >> 63) + (u64)(s >> 64);
      x[w] = (u64)s;
    }
    bn_divmod(x, y, t.td_n, (int)t.td_nw, (int)t.td_sh, W);       // y = quotient
    const bool qzero = bn_is_zero(y, W);
    for (u32 limb = 0; limb < L; ++limb) {
      const Mod m = t.mods[limb];
      u64 r = bn_mod_small(y, t.pow64 + (size_t)limb * W, m, W);
      nres[(int)limb] = (pneg && !qzero && r) ? m.q - r : r;
    }
  }
  // plaintext = -z_0 - noise_0 (:51-53), then extract_constant_term_as_u64 (:226-247)
  bool vneg = lift_centered(t, x, [&](u32 limb) -> u64 {
    const Mod m = t.mods[limb];
    u64 z0 = noisy[(size_t)limb * l];
    u64 a = z0 ? m.q - z0 : 0;
    return submod(a, nres[(int)limb], m.q);
  });
  if (vneg && !bn_is_zero(x, W)) {
    bool hi = false;
    for (int w = 1; w < W; ++w) hi |= x[w] != 0;
    if (!hi && x[0] <= 1000) return 0;       // small negative -> 0 (:233-235)
    bn_rsub_c(x, t.Q, W);                    // (v + Q) % Q = Q - |v|
  }
  for (int w = 1; w < W; ++w)
    if (x[w] != 0) return 0;                 // does not fit u64 (:240,:243)
  return x[0];
}

}  // namespace pvw
