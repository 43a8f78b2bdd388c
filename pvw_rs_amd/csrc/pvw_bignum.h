// pvw_bignum.h -- small host-side arbitrary-precision integer (sign + magnitude).
//
// The reference leans on num-bigint for the parameter set-up (Delta = floor(Q^(1/l)),
// src/params/parameters.rs:150-163), the correctness gate (:510-551) and the PVW gadget
// decode (src/crypto/decryption.rs:10-247).  Those are host-side, per-call-tiny integer
// computations on 100..2100-bit numbers; this is the minimal arithmetic they need.
// There is no GMP/Boost header in the build image, hence an in-repo implementation.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <vector>

namespace pvw {

class BigInt {
 public:
  typedef unsigned __int128 u128;
  std::vector<uint64_t> mag;  // little-endian, no leading zero words
  bool neg = false;

  BigInt() {}
  BigInt(uint64_t v) { if (v) mag.push_back(v); }
  static BigInt from_i64(int64_t v) {
    BigInt r(v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v);
    r.neg = v < 0;
    return r;
  }
  static BigInt from_words(const uint64_t* w, size_t n) {
    BigInt r;
    r.mag.assign(w, w + n);
    r.trim();
    return r;
  }
  bool is_zero() const { return mag.empty(); }
  bool is_negative() const { return neg && !mag.empty(); }
  size_t bits() const {
    if (mag.empty()) return 0;
    return 64 * (mag.size() - 1) + (64 - __builtin_clzll(mag.back()));
  }
  void trim() {
    while (!mag.empty() && mag.back() == 0) mag.pop_back();
    if (mag.empty()) neg = false;
  }
  bool fits_u64() const { return !is_negative() && mag.size() <= 1; }
  uint64_t low_u64() const { return mag.empty() ? 0 : mag[0]; }

  static int cmp_abs(const BigInt& a, const BigInt& b) {
    if (a.mag.size() != b.mag.size()) return a.mag.size() < b.mag.size() ? -1 : 1;
    for (size_t i = a.mag.size(); i-- > 0;)
      if (a.mag[i] != b.mag[i]) return a.mag[i] < b.mag[i] ? -1 : 1;
    return 0;
  }
  static int cmp(const BigInt& a, const BigInt& b) {
    bool an = a.is_negative(), bn = b.is_negative();
    if (an != bn) return an ? -1 : 1;
    int c = cmp_abs(a, b);
    return an ? -c : c;
  }
  friend bool operator<(const BigInt& a, const BigInt& b) { return cmp(a, b) < 0; }
  friend bool operator>(const BigInt& a, const BigInt& b) { return cmp(a, b) > 0; }
  friend bool operator<=(const BigInt& a, const BigInt& b) { return cmp(a, b) <= 0; }
  friend bool operator>=(const BigInt& a, const BigInt& b) { return cmp(a, b) >= 0; }
  friend bool operator==(const BigInt& a, const BigInt& b) { return cmp(a, b) == 0; }
  friend bool operator!=(const BigInt& a, const BigInt& b) { return cmp(a, b) != 0; }

  static void add_abs(std::vector<uint64_t>& r, const std::vector<uint64_t>& a,
                      const std::vector<uint64_t>& b) {
    const std::vector<uint64_t>& x = a.size() >= b.size() ? a : b;
    const std::vector<uint64_t>& y = a.size() >= b.size() ? b : a;
    std::vector<uint64_t> out(x.size() + 1);
    uint64_t carry = 0;
    for (size_t i = 0; i < x.size(); ++i) {
      u128 s = (u128)x[i] + (i < y.size() ? y[i] : 0) + carry;
      out[i] = (uint64_t)s;
      carry = (uint64_t)(s >> 64);
    }
    out[x.size()] = carry;
    r.swap(out);
  }
  // r = a - b, requires |a| >= |b|
  static void sub_abs(std::vector<uint64_t>& r, const std::vector<uint64_t>& a,
                      const std::vector<uint64_t>& b) {
    std::vector<uint64_t> out(a.size());
    uint64_t borrow = 0;
    for (size_t i = 0; i < a.size(); ++i) {
      uint64_t bi = i < b.size() ? b[i] : 0;
      u128 d = (u128)a[i] - bi - borrow;
      out[i] = (uint64_t)d;
      borrow = (uint64_t)(d >> 64) & 1;
    }
    r.swap(out);
  }
  BigInt operator-() const {
    BigInt r = *this;
    if (!r.mag.empty()) r.neg = !r.neg;
    return r;
  }
  friend BigInt operator+(const BigInt& a, const BigInt& b) {
    BigInt r;
    if (a.is_negative() == b.is_negative()) {
      add_abs(r.mag, a.mag, b.mag);
      r.neg = a.is_negative();
    } else {
      int c = cmp_abs(a, b);
      if (c >= 0) { sub_abs(r.mag, a.mag, b.mag); r.neg = a.is_negative(); }
      else { sub_abs(r.mag, b.mag, a.mag); r.neg = b.is_negative(); }
    }
    r.trim();
    return r;
  }
  friend BigInt operator-(const BigInt& a, const BigInt& b) { return a + (-b); }
  friend BigInt operator*(const BigInt& a, const BigInt& b) {
    BigInt r;
    if (a.mag.empty() || b.mag.empty()) return r;
    r.mag.assign(a.mag.size() + b.mag.size(), 0);
    for (size_t i = 0; i < a.mag.size(); ++i) {
      uint64_t carry = 0;
      for (size_t j = 0; j < b.mag.size(); ++j) {
        u128 t = (u128)a.mag[i] * b.mag[j] + r.mag[i + j] + carry;
        r.mag[i + j] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
      }
      r.mag[i + b.mag.size()] += carry;
    }
    r.neg = a.is_negative() != b.is_negative();
    r.trim();
    return r;
  }
  BigInt shl(size_t s) const {
    BigInt r;
    if (mag.empty()) return r;
    size_t ws = s / 64, bs = s % 64;
    r.mag.assign(mag.size() + ws + 1, 0);
    for (size_t i = 0; i < mag.size(); ++i) {
      r.mag[i + ws] |= mag[i] << bs;
      if (bs) r.mag[i + ws + 1] |= mag[i] >> (64 - bs);
    }
    r.neg = neg;
    r.trim();
    return r;
  }
  BigInt shr(size_t s) const {  // magnitude shift (truncates toward zero)
    BigInt r;
    size_t ws = s / 64, bs = s % 64;
    if (ws >= mag.size()) return r;
    r.mag.assign(mag.size() - ws, 0);
    for (size_t i = 0; i < r.mag.size(); ++i) {
      r.mag[i] = mag[i + ws] >> bs;
      if (bs && i + ws + 1 < mag.size()) r.mag[i] |= mag[i + ws + 1] << (64 - bs);
    }
    r.neg = neg;
    r.trim();
    return r;
  }
  // |this| mod m for a 64-bit m
  uint64_t mod_small_abs(uint64_t m) const {
    u128 r = 0;
    for (size_t i = mag.size(); i-- > 0;) r = ((r << 64) | mag[i]) % m;
    return (uint64_t)r;
  }
  // non-negative residue of the signed value: ((c % m) + m) % m
  uint64_t mod_small(uint64_t m) const {
    uint64_t r = mod_small_abs(m);
    return (is_negative() && r) ? m - r : r;
  }

  // truncating division (num-bigint BigInt '/' and '%': quotient toward zero, remainder
  // with the sign of the dividend).  Knuth algorithm D on 64-bit digits.
  static void divmod_trunc(const BigInt& a, const BigInt& b, BigInt& q, BigInt& r) {
    q = BigInt();
    r = BigInt();
    if (b.mag.empty()) return;  // caller guards division by zero
    if (cmp_abs(a, b) < 0) { r = a; return; }
    if (b.mag.size() == 1) {
      uint64_t d = b.mag[0];
      q.mag.assign(a.mag.size(), 0);
      u128 rem = 0;
      for (size_t i = a.mag.size(); i-- > 0;) {
        u128 cur = (rem << 64) | a.mag[i];
        q.mag[i] = (uint64_t)(cur / d);
        rem = cur % d;
      }
      if (rem) r.mag.push_back((uint64_t)rem);
    } else {
      int s = __builtin_clzll(b.mag.back());
      BigInt u = a.shl(s), v = b.shl(s);
      u.neg = v.neg = false;
      size_t n = v.mag.size(), m = u.mag.size() >= n ? u.mag.size() - n : 0;
      u.mag.resize(std::max(u.mag.size(), a.mag.size() + 1), 0);
      if (u.mag.size() < n + m + 1) u.mag.resize(n + m + 1, 0);
      m = u.mag.size() - n - 1;
      q.mag.assign(m + 1, 0);
      for (size_t j = m + 1; j-- > 0;) {
        u128 num = ((u128)u.mag[j + n] << 64) | u.mag[j + n - 1];
        u128 qhat = num / v.mag[n - 1], rhat = num % v.mag[n - 1];
        while ((qhat >> 64) || (uint64_t)qhat * (u128)v.mag[n - 2] > ((rhat << 64) | u.mag[j + n - 2])) {
          --qhat;
          rhat += v.mag[n - 1];
          if (rhat >> 64) break;
        }
        // multiply and subtract
        uint64_t borrow = 0, carry = 0;
        for (size_t i = 0; i < n; ++i) {
          u128 p = (u128)(uint64_t)qhat * v.mag[i] + carry;
          carry = (uint64_t)(p >> 64);
          u128 d = (u128)u.mag[i + j] - (uint64_t)p - borrow;
          u.mag[i + j] = (uint64_t)d;
          borrow = (uint64_t)(d >> 64) & 1;
        }
        u128 d = (u128)u.mag[j + n] - carry - borrow;
        u.mag[j + n] = (uint64_t)d;
        if ((uint64_t)(d >> 64) & 1) {  // qhat was one too large: add back
          --qhat;
          uint64_t c = 0;
          for (size_t i = 0; i < n; ++i) {
            u128 t = (u128)u.mag[i + j] + v.mag[i] + c;
            u.mag[i + j] = (uint64_t)t;
            c = (uint64_t)(t >> 64);
          }
          u.mag[j + n] += c;
        }
        q.mag[j] = (uint64_t)qhat;
      }
      u.mag.resize(n);
      u.trim();
      r = u.shr(s);
    }
    q.neg = a.is_negative() != b.is_negative();
    r.neg = a.is_negative();
    q.trim();
    r.trim();
  }
  friend BigInt operator/(const BigInt& a, const BigInt& b) {
    BigInt q, r;
    divmod_trunc(a, b, q, r);
    return q;
  }
  friend BigInt operator%(const BigInt& a, const BigInt& b) {
    BigInt q, r;
    divmod_trunc(a, b, q, r);
    return r;
  }
  // non-negative residue mod a positive m (Euclidean)
  BigInt mod_floor(const BigInt& m) const {
    BigInt r = *this % m;
    if (r.is_negative()) r = r + m;
    return r;
  }
  BigInt pow(uint32_t e) const {
    BigInt r(1), b = *this;
    while (e) {
      if (e & 1) r = r * b;
      e >>= 1;
      if (e) b = b * b;
    }
    return r;
  }
  // floor(x^(1/n)) for x >= 0 -- BigUint::nth_root (parameters.rs:156)
  BigInt nth_root(uint32_t n) const {
    if (mag.empty()) return BigInt();
    BigInt lo, hi = BigInt(1).shl((bits() + n - 1) / n);
    while (lo < hi) {  // lo^n <= x < (hi+1)^n
      BigInt mid = (lo + hi + BigInt(1)).shr(1);
      if (mid.pow(n) <= *this) lo = mid;
      else hi = mid - BigInt(1);
    }
    return lo;
  }
  // correctly rounded conversion, saturating to +-inf (num-bigint to_f64)
  double to_double() const {
    if (mag.empty()) return 0.0;
    size_t b = bits();
    if (b > 1024) return neg ? -INFINITY : INFINITY;
    // top 64 bits with a sticky bit for everything below
    uint64_t top;
    long exp;
    if (b <= 64) { top = mag[0]; exp = 0; }
    else {
      BigInt t = shr(b - 64);
      top = t.mag[0];
      exp = (long)(b - 64);
      // sticky
      BigInt back = t.shl(b - 64);
      back.neg = false;
      BigInt self = *this;
      self.neg = false;
      if (back != self) top |= 1;
    }
    double d = std::ldexp((double)top, (int)exp);  // (double)top rounds to nearest even with sticky
    return neg ? -d : d;
  }
};

}  // namespace pvw
