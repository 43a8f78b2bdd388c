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

// ---- arithmetic of the short cuts of the wave-cooperative decode (pvw_decode_wave.h), host + device ----
// Garner's mixed-radix digits over the first n = gar_n moduli: from the residues r[0..n) of a value to its centred
// representative modulo P = q_0..q_{n-1}, as four words + sign in out[0..4] (out[4] bit 0 = negative).  A candidate only:
// it IS the value's centred representative modulo Q exactly when it reproduces the residue on every limb (|v| <= P/2 < Q/2).
PVW_HD void garner_small(const DecodeTables& t, const u64 (&r)[4], u64* out) {
  const u32 NL = t.gar_n;
  const u64* g = t.gar;
  u64 xm[4] = {0, 0, 0, 0};
#pragma unroll
  for (u32 j = 0; j < 4; ++j) {
    if (j < NL) {
      const Mod mj = t.mods[j];
      u64 u = r[j];
#pragma unroll
      for (u32 i = 0; i < j; ++i) {
        u = submod(u, t.gar_close ? (xm[i] >= mj.q ? xm[i] - mj.q : xm[i]) : reduce128(xm[i], 0, mj), mj.q);
        u = mulmod_shoup(u, g[4 * j + i], g[16 + 4 * j + i], mj.q);
      }
      xm[j] = u;
    }
  }
  // v = x_0 + x_1 q_0 + x_2 q_0 q_1 + x_3 q_0 q_1 q_2  < P
  u64 v0, v1, v2 = 0, v3 = 0;
  {
    u128 p = (u128)xm[1] * g[36] + xm[0];
    v0 = (u64)p;
    v1 = (u64)(p >> 64);
  }
  if (NL > 2) {
    u128 p = (u128)xm[2] * g[40] + v0;
    v0 = (u64)p;
    p = (u128)xm[2] * g[41] + v1 + (u64)(p >> 64);
    v1 = (u64)p;
    v2 = (u64)(p >> 64);
  }
  if (NL > 3) {
    u128 p = (u128)xm[3] * g[44] + v0;
    v0 = (u64)p;
    p = (u128)xm[3] * g[45] + v1 + (u64)(p >> 64);
    v1 = (u64)p;
    p = (u128)xm[3] * g[46] + v2 + (u64)(p >> 64);
    v2 = (u64)p;
    v3 = (u64)(p >> 64);
  }
  // centre modulo P
  const u128 vlo = ((u128)v1 << 64) | v0, vhi = ((u128)v3 << 64) | v2;
  const u128 hlo = ((u128)g[53] << 64) | g[52], hhi = ((u128)g[55] << 64) | g[54];
  const bool ng = vhi != hhi ? vhi > hhi : vlo > hlo;
  if (ng) {
    const u64* P = g + 32 + 4 * NL;
    const u128 plo = ((u128)P[1] << 64) | P[0], phi = ((u128)P[3] << 64) | P[2];
    const u128 dlo = plo - vlo, dhi = phi - vhi - (plo < vlo ? 1 : 0);
    v0 = (u64)dlo; v1 = (u64)(dlo >> 64); v2 = (u64)dhi; v3 = (u64)(dhi >> 64);
  }
  out[0] = v0; out[1] = v1; out[2] = v2; out[3] = v3; out[4] = ng ? 1 : 0;
}
// One step of the chain, noise_i = round((noise_{i+1} - tmp_i) / Delta) (decryption.rs:44-48, :180-207), on noise-sized
// operands held by ONE LANE: noise_{i+1} = a and tmp_i = the confirmed candidate cb, both below 2^191 in magnitude, so
// the step is a few dozen word operations (the general step below spends ~20 ballots and lane shifts on W-word integers
// that are almost all zeros).  With Q >= 2^193 the integer a - b is the centred difference mod Q; round(p / Delta) =
// sign(p) * floor((2|p| + Delta) / (2 Delta)) as the general step computes it, here by one Knuth step (4 words by 3,
// one-word quotient, trial digit from the top words by Moeller-Granlund's reciprocal).  Returns false when an operand or
// the quotient does not fit (q, qneg are then meaningless).
struct SmallVal {
  u64 w0, w1, w2;
  bool neg;
};
PVW_HD bool small_chain_step(const u64* sc, const SmallVal& a, const u64* cb, u64& q, bool& qneg) {
  const u64 b0 = cb[0], b1 = cb[1], b2 = cb[2], b3 = cb[3], bf = cb[4];
  bool ok = (bf & 2) != 0 && b3 == 0 && (b2 >> 63) == 0 && (a.w2 >> 63) == 0;
  const bool bneg = (bf & 1) != 0;
  u64 p0, p1, p2, p3;                                      // |p|, p = a - b
  bool pneg;
  if (a.neg != bneg) {
    u128 s = (u128)a.w0 + b0;
    p0 = (u64)s;
    s = (u128)a.w1 + b1 + (u64)(s >> 64);
    p1 = (u64)s;
    s = (u128)a.w2 + b2 + (u64)(s >> 64);
    p2 = (u64)s;
    p3 = (u64)(s >> 64);
    pneg = a.neg;
  } else {
    const bool age = a.w2 != b2 ? a.w2 > b2 : (a.w1 != b1 ? a.w1 > b1 : a.w0 >= b0);
    const u64 x0 = age ? a.w0 : b0, x1 = age ? a.w1 : b1, x2 = age ? a.w2 : b2;
    const u64 y0 = age ? b0 : a.w0, y1 = age ? b1 : a.w1, y2 = age ? b2 : a.w2;
    const u128 xl = ((u128)x1 << 64) | x0, yl = ((u128)y1 << 64) | y0, dl = xl - yl;
    p0 = (u64)dl;
    p1 = (u64)(dl >> 64);
    p2 = x2 - y2 - (xl < yl ? 1 : 0);
    p3 = 0;
    pneg = age ? a.neg : !a.neg;
  }
  // 2|p| + Delta
  u64 n0 = p0 << 1, n1 = (p1 << 1) | (p0 >> 63), n2 = (p2 << 1) | (p1 >> 63), n3 = (p3 << 1) | (p2 >> 63);
  {
    u128 s = (u128)n0 + sc[6];
    n0 = (u64)s;
    s = (u128)n1 + sc[7] + (u64)(s >> 64);
    n1 = (u64)s;
    s = (u128)n2 + sc[8] + (u64)(s >> 64);
    n2 = (u64)s;
    n3 += (u64)(s >> 64);
  }
  // the shift that normalised the divisor; anything pushed out means a quotient of more than one word
  const u32 ws = (u32)sc[4], bs = (u32)sc[5];              // wave-uniform
  u64 m0, m1, m2, m3;
  if (ws == 0) {
    m0 = n0; m1 = n1; m2 = n2; m3 = n3;
  } else if (ws == 1) {
    ok = ok && n3 == 0;
    m0 = 0; m1 = n0; m2 = n1; m3 = n2;
  } else {
    ok = ok && (n3 | n2) == 0;
    m0 = 0; m1 = 0; m2 = n0; m3 = n1;
  }
  if (bs) {
    ok = ok && (m3 >> (64 - bs)) == 0;
    m3 = (m3 << bs) | (m2 >> (64 - bs));
    m2 = (m2 << bs) | (m1 >> (64 - bs));
    m1 = (m1 << bs) | (m0 >> (64 - bs));
    m0 <<= bs;
  }
  const u64 d0 = sc[0], d1 = sc[1], d2 = sc[2], v = sc[3];
  ok = ok && (m3 != d2 ? m3 < d2 : (m2 != d1 ? m2 < d1 : m1 < d0));              // one-word quotient
  u64 qh = ~0ULL;                                          // trial digit, at most 2 too large (Knuth D3)
  if (m3 < d2) {
    const u128 qq = (u128)v * m3 + (((u128)m3 << 64) | m2);
    u64 q1 = (u64)(qq >> 64) + 1;
    const u64 q0 = (u64)qq;
    u64 r = m2 - q1 * d2;
    if (r > q0) { --q1; r += d2; }
    if (r >= d2) { ++q1; r -= d2; }
    qh = q1;
  }
  // m - qh * d; below zero: the trial was too large
  const u128 t0 = (u128)qh * d0, t1 = (u128)qh * d1 + (u64)(t0 >> 64), t2 = (u128)qh * d2 + (u64)(t1 >> 64);
  const u128 ml = ((u128)m1 << 64) | m0, mh = ((u128)m3 << 64) | m2;
  const u128 tl = ((u128)(u64)t1 << 64) | (u64)t0;
  u128 rl = ml - tl;
  const u128 th_b = t2 + (ml < tl ? 1 : 0);                // t2 <= 2^128 - 2^64: the borrow cannot wrap it
  bool below = ok && mh < th_b;
  u128 rh = mh - th_b;
  const u128 dlw = ((u128)d1 << 64) | d0;
  for (int fix = 0; fix < 2 && below; ++fix) {
    --qh;
    const u128 nl = rl + dlw;
    const u128 nh = rh + d2 + (nl < rl ? 1 : 0);
    below = nh >= rh;                                      // no wrap past 2^128: still below zero (d2 + carry > 0)
    rl = nl;
    rh = nh;
  }
  q = qh;
  qneg = pneg && qh != 0;
  return ok && !below;
}
// noise_{l-1} without the Horner lift, the parts that are plain arithmetic (pvw_decode_wave.h: small_top has the story).
// The guess g = b + round(-b/Delta) Delta for b = tmp_{l-2} (cb: its confirmed candidate), qv = |round(-b/Delta)|.
PVW_HD bool small_top_guess(const u64* sc, const u64* cb, u64 qv, SmallVal& g) {
  const u64 b0 = cb[0], b1 = cb[1], b2 = cb[2];
  const bool bneg = (cb[4] & 1) != 0;
  const u128 t0 = (u128)qv * sc[6], t1 = (u128)qv * sc[7] + (u64)(t0 >> 64), t2 = (u128)qv * sc[8] + (u64)(t1 >> 64);
  const u128 pl = ((u128)(u64)t1 << 64) | (u64)t0, ph = t2;                  // qv * Delta, 256 bits
  const u128 bl = ((u128)b1 << 64) | b0, bh = b2;
  const bool bge = bh != ph ? bh > ph : bl >= pl;
  const u128 xl = bge ? bl : pl, xh = bge ? bh : ph, yl = bge ? pl : bl, yh = bge ? ph : bh;
  const u128 gl = xl - yl, gh = xh - yh - (xl < yl ? 1 : 0);
  if ((u64)(gh >> 63) != 0) return false;                           // |g| must stay below 2^191
  g.w0 = (u64)gl; g.w1 = (u64)(gl >> 64); g.w2 = (u64)gh;
  g.neg = (g.w0 | g.w1 | g.w2) != 0 && (bge ? bneg : !bneg);
  return true;
}
// one limb's e_i = (H_i - g_i) * (Delta^(l-1))^-1 mod q, H_i = z_0 Delta^(l-1) - z_{l-1} (the Horner sum telescopes);
// pw0..2 = 2^(64 w) mod q
PVW_HD u64 small_top_quotient(const SmallVal& g, u64 z0, u64 zl1, u64 dp, u64 dpp, u64 dpinv, u64 dpinvp, u64 pw0, u64 pw1, u64 pw2, const Mod& m) {
  const u64 hi = submod(mulmod_shoup(z0, dp, dpp, m.q), zl1, m.q);
  const u128 gs = (u128)g.w0 * pw0 + (u128)g.w1 * pw1 + (u128)g.w2 * pw2;
  u64 gi = reduce128((u64)gs, (u64)(gs >> 64), m);
  if (g.neg && gi) gi = m.q - gi;
  return mulmod_shoup(submod(hi, gi, m.q), dpinv, dpinvp, m.q);
}
// what e_i must be on this limb if limb 0 says e = e0 (centred modulo q_0)
PVW_HD u64 small_top_expected(u64 e0, u64 q0, const Mod& m) {
  const bool eneg = e0 > (q0 >> 1);
  const u64 emag = eneg ? q0 - e0 : e0;
  u64 want = reduce128(emag, 0, m);
  if (eneg && want) want = m.q - want;
  return want;
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
