// pvw_arith.h -- 64-bit modular arithmetic shared by host and gfx950 device code.
//
// Everything the kernels compute is unsigned 64-bit modular integer arithmetic
// over RNS limbs q_i < 2^62 (what fhe-math's zq::Modulus does for the reference;
// its source is not vendored, so this is written from the mathematics).
//
//   * Mod / reduce128 / mulmod: Barrett reduction of a full 128-bit value with the
//     128-bit ratio floor(2^128/q).  In the kernels q and the ratio are wave-uniform
//     (one limb per workgroup), so they sit in SGPRs: "wavefront-wide Barrett".
//   * Acc: lazy accumulator for sum_j x_j*y_j of up to 2^32 products of 64-bit
//     operands with a single reduction at the end.  The four 32x32->64 partial
//     products are accumulated separately (v_mad_u64_u32 with the carry-out added
//     into a 32-bit overflow counter), so one MAC is 4 multiply-adds + 4 adds and
//     there is no carry chain between the partial sums.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PVW_HD __host__ __device__ __forceinline__
#define PVW_D __device__ __forceinline__
#else
#define PVW_HD inline
#endif

namespace pvw {

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef uint32_t u32;
typedef int64_t i64;

struct Mod {
  u64 q;
  u64 ratio_lo, ratio_hi;  // floor(2^128 / q)
};

inline Mod make_mod(u64 q) {
  Mod m;
  m.q = q;
  u128 r = (~(u128)0) / q;  // == floor(2^128/q) for q not a power of two
  m.ratio_lo = (u64)r;
  m.ratio_hi = (u64)(r >> 64);
  return m;
}

PVW_HD u64 mulhi64(u64 a, u64 b) { return (u64)(((u128)a * b) >> 64); }

// (hi:lo) mod q for ANY 128-bit value, q < 2^62.  quo = floor(x*ratio/2^128) mod 2^64 is
// at most 2 below the true quotient, so x - quo*q (mod 2^64) lies in [0, 3q) < 2^64.
PVW_HD u64 reduce128(u64 lo, u64 hi, const Mod& m) {
  u128 b = (u128)lo * m.ratio_hi;
  u128 c = (u128)hi * m.ratio_lo;
  u128 mid = (u128)mulhi64(lo, m.ratio_lo) + (u64)b + (u64)c;
  u64 quo = (u64)(mid >> 64) + (u64)(b >> 64) + (u64)(c >> 64) + hi * m.ratio_hi;
  u64 r = lo - quo * m.q;
  if (r >= m.q) r -= m.q;
  if (r >= m.q) r -= m.q;
  return r;
}
PVW_HD u64 mulmod(u64 a, u64 b, const Mod& m) {
  u128 p = (u128)a * b;
  return reduce128((u64)p, (u64)(p >> 64), m);
}
// Shoup / Harvey multiplication by a fixed w < q with wp = floor(w * 2^64 / q):
// a*w - hi64(a*wp)*q lies in [0, 2q) for ANY 64-bit a.  Used for every twiddle / table multiply.
PVW_HD u64 mulmod_shoup(u64 a, u64 w, u64 wp, u64 q) {
  u64 r = a * w - mulhi64(a, wp) * q;
  return r >= q ? r - q : r;
}
inline u64 shoup_precompute(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }
PVW_HD u64 addmod(u64 a, u64 b, u64 q) {
  u64 s = a + b;
  return s >= q ? s - q : s;
}
PVW_HD u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }

// non-negative residue of a signed value: ((c % q) + q) % q  (parameters.rs:440-443,
// the rule Poly::from_coefficients(&[i64]) follows as well, tests/params.rs:733-767)
PVW_HD u64 signed_residue(i64 c, const Mod& m) {
  u64 a = c < 0 ? (u64)0 - (u64)c : (u64)c;
  u64 r = a < m.q ? a : reduce128(a, 0, m);   // small coefficients (the common case) need no reduction
  return (c < 0 && r != 0) ? m.q - r : r;
}

inline u64 powmod(u64 b, u64 e, const Mod& m) {
  u64 r = 1 % m.q;
  b = reduce128(b, 0, m);
  while (e) {
    if (e & 1) r = mulmod(r, b, m);
    b = mulmod(b, b, m);
    e >>= 1;
  }
  return r;
}

// ---------------------------------------------------------------- lazy accumulator
struct Acc {
  u64 ll, lh, hl, hh;      // sums of xl*yl, xl*yh, xh*yl, xh*yh (mod 2^64)
  u32 cll, clh, chl, chh;  // how many times each sum wrapped
};
PVW_HD void acc_zero(Acc& a) {
  a.ll = a.lh = a.hl = a.hh = 0;
  a.cll = a.clh = a.chl = a.chh = 0;
}
PVW_HD void acc_part(u64& s, u32& c, u32 x, u32 y) {
  u64 p = (u64)x * y;
  u64 t = s + p;
  c += (t < p);
  s = t;
}
// portable form (host tests, and the reference the asm form is checked against)
PVW_HD void acc_mac(Acc& a, u64 x, u64 y) {
  u32 xl = (u32)x, xh = (u32)(x >> 32), yl = (u32)y, yh = (u32)(y >> 32);
  acc_part(a.ll, a.cll, xl, yl);
  acc_part(a.lh, a.clh, xl, yh);
  acc_part(a.hl, a.chl, xh, yl);
  acc_part(a.hh, a.chh, xh, yh);
}
#if defined(__HIPCC__)
// gfx950 form: 4 x v_mad_u64_u32 (carry-out to an SGPR pair) + 4 x v_addc_co_u32.
// Each carry is consumed >= 3 instructions after it is produced, which covers the
// 2 wait states gfx950 wants between a VALU write of an SGPR and a VALU read of it
// as a carry-in (hipcc pads nothing inside an asm statement).
PVW_D void acc_mac_dev(Acc& a, u64 x, u64 y) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 xl = (u32)x, xh = (u32)(x >> 32), yl = (u32)y, yh = (u32)(y >> 32);
  u64 c0, c1, c2, c3;
  asm("v_mad_u64_u32 %0, %8, %12, %14, %0\n\t"
      "v_mad_u64_u32 %1, %9, %12, %15, %1\n\t"
      "v_mad_u64_u32 %2, %10, %13, %14, %2\n\t"
      "v_mad_u64_u32 %3, %11, %13, %15, %3\n\t"
      "v_addc_co_u32_e64 %4, %8, 0, %4, %8\n\t"
      "v_addc_co_u32_e64 %5, %9, 0, %5, %9\n\t"
      "v_addc_co_u32_e64 %6, %10, 0, %6, %10\n\t"
      "v_addc_co_u32_e64 %7, %11, 0, %7, %11"
      : "+v"(a.ll), "+v"(a.lh), "+v"(a.hl), "+v"(a.hh), "+v"(a.cll), "+v"(a.clh), "+v"(a.chl),
        "+v"(a.chh), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
      : "v"(xl), "v"(xh), "v"(yl), "v"(yh));
#else
  acc_mac(a, x, y);
#endif
}
#endif
PVW_HD void acc_add(Acc& a, const Acc& b) {  // a += b  (cross-wave reduction)
  u64 t;
  t = a.ll + b.ll; a.cll += b.cll + (t < b.ll); a.ll = t;
  t = a.lh + b.lh; a.clh += b.clh + (t < b.lh); a.lh = t;
  t = a.hl + b.hl; a.chl += b.chl + (t < b.hl); a.hl = t;
  t = a.hh + b.hh; a.chh += b.chh + (t < b.hh); a.hh = t;
}
// value = (ll + cll*2^64) + ((lh + clh*2^64) + (hl + chl*2^64))*2^32 + (hh + chh*2^64)*2^64,
// reduced mod q.  The total is < 2^160 for <= 2^32 terms, held in three 64-bit words.
PVW_HD void acc_words(const Acc& a, u64& t0, u64& t1, u64& t2) {   // total = t2:t1:t0
  u128 mid = (u128)a.lh + a.hl;                                  // < 2^65
  u128 midc = (u128)a.clh + a.chl + (u64)(mid >> 64);            // weight 2^96
  u64 midlo = (u64)mid;                                          // weight 2^32
  u128 w0 = (u128)a.ll + ((u128)(midlo & 0xffffffffULL) << 32);
  t0 = (u64)w0;
  u128 w1 = (w0 >> 64) + (u128)a.cll + (midlo >> 32) + (midc << 32) + a.hh;
  t1 = (u64)w1;
  t2 = (u64)(w1 >> 64) + a.chh;
}
PVW_HD u64 acc_reduce(const Acc& a, const Mod& m) {
  u64 t0, t1, t2;
  acc_words(a, t0, t1, t2);
  u64 h = reduce128(t1, t2, m);                                  // (t2:t1) mod q
  return reduce128(t0, h, m);                                    // (h*2^64 + t0) mod q
}

PVW_HD u32 bitrev32(u32 i, u32 bits) {
  u32 r = 0;
  for (u32 b = 0; b < bits; ++b) {
    r = (r << 1) | (i & 1);
    i >>= 1;
  }
  return r;
}

// ---------------------------------------------------------------- l-point negacyclic NTT
// In-register transform of one limb, L_ a compile-time 8/16/32/64: natural order in,
// bit-reversed order out (slot s holds a(psi^(2*bitrev(s)+1))).  tw[i] = psi^bitrev(i),
// twp[i] = floor(tw[i]*2^64/q) (Shoup).
template <int L_>
PVW_HD void ntt_forward(u64 (&a)[L_], const u64* tw, const u64* twp, const Mod& m) {
  int step = L_;
#pragma unroll
  for (int mm = 1; mm < L_; mm <<= 1) {
    step >>= 1;
#pragma unroll
    for (int i = 0; i < mm; ++i) {
      u64 w = tw[mm + i], wp = twp[mm + i];
#pragma unroll
      for (int j = 2 * i * step; j < 2 * i * step + step; ++j) {
        u64 u = a[j], v = mulmod_shoup(a[j + step], w, wp, m.q);
        a[j] = addmod(u, v, m.q);
        a[j + step] = submod(u, v, m.q);
      }
    }
  }
}
// inverse: bit-reversed in, natural out; itw[i] = psi^-bitrev(i); linv = l^-1
template <int L_>
PVW_HD void ntt_inverse(u64 (&a)[L_], const u64* itw, const u64* itwp, u64 linv, u64 linvp, const Mod& m) {
  int step = 1;
#pragma unroll
  for (int mm = L_ >> 1; mm >= 1; mm >>= 1) {
#pragma unroll
    for (int i = 0; i < mm; ++i) {
      u64 w = itw[mm + i], wp = itwp[mm + i];
#pragma unroll
      for (int j = 2 * i * step; j < 2 * i * step + step; ++j) {
        u64 u = a[j], v = a[j + step];
        a[j] = addmod(u, v, m.q);
        a[j + step] = mulmod_shoup(submod(u, v, m.q), w, wp, m.q);
      }
    }
    step <<= 1;
  }
#pragma unroll
  for (int j = 0; j < L_; ++j) a[j] = mulmod_shoup(a[j], linv, linvp, m.q);
}

}  // namespace pvw
