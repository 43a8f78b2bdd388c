// pvw_chacha.h -- counter-based randomness: one ChaCha8 word stream per polynomial.
//
// The reference samples from rand::thread_rng() inside rayon closures
// (src/crypto/encryption.rs:138,164,180) and cannot be replayed.  This build keys
// every polynomial's randomness by (32-byte seed, domain, polynomial index): the
// ChaCha state is laid out as rand_chacha does (constants | key | 64-bit block
// counter | 64-bit stream id) with stream id = (domain << 32) | index, and the
// samplers consume next_u32/next_u64 in the order the reference's samplers do
// (src/sampling/uniform.rs), so a host holding the same seed reproduces them.
#pragma once
#include "pvw_arith.h"

namespace pvw {

struct ChaChaKey {
  u32 w[8];
};

inline ChaChaKey make_key(const uint8_t seed[32]) {
  ChaChaKey k;
  for (int i = 0; i < 8; ++i)
    k.w[i] = (u32)seed[4 * i] | (u32)seed[4 * i + 1] << 8 | (u32)seed[4 * i + 2] << 16 |
             (u32)seed[4 * i + 3] << 24;
  return k;
}

PVW_HD u32 rotl32(u32 x, int n) { return (x << n) | (x >> (32 - n)); }

#define PVW_QR(a, b, c, d)                                                                   \
  a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12);                       \
  a += b; d ^= a; d = rotl32(d, 8);  c += d; b ^= c; b = rotl32(b, 7);

struct ChaChaRng {
  u32 key[8];
  u32 s0, s1;   // stream id words (14, 15)
  u64 counter;  // next block
  u32 buf[16];
  int pos;

  PVW_HD void init(const ChaChaKey& k, u32 domain, u32 index) {
#pragma unroll
    for (int i = 0; i < 8; ++i) key[i] = k.w[i];
    s0 = index;
    s1 = domain;
    counter = 0;
    pos = 16;
  }
  PVW_HD void refill() {
    u32 x0 = 0x61707865, x1 = 0x3320646e, x2 = 0x79622d32, x3 = 0x6b206574;
    u32 x4 = key[0], x5 = key[1], x6 = key[2], x7 = key[3];
    u32 x8 = key[4], x9 = key[5], x10 = key[6], x11 = key[7];
    u32 x12 = (u32)counter, x13 = (u32)(counter >> 32), x14 = s0, x15 = s1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      PVW_QR(x0, x4, x8, x12) PVW_QR(x1, x5, x9, x13)
      PVW_QR(x2, x6, x10, x14) PVW_QR(x3, x7, x11, x15)
      PVW_QR(x0, x5, x10, x15) PVW_QR(x1, x6, x11, x12)
      PVW_QR(x2, x7, x8, x13) PVW_QR(x3, x4, x9, x14)
    }
    buf[0] = x0 + 0x61707865; buf[1] = x1 + 0x3320646e; buf[2] = x2 + 0x79622d32; buf[3] = x3 + 0x6b206574;
    buf[4] = x4 + key[0]; buf[5] = x5 + key[1]; buf[6] = x6 + key[2]; buf[7] = x7 + key[3];
    buf[8] = x8 + key[4]; buf[9] = x9 + key[5]; buf[10] = x10 + key[6]; buf[11] = x11 + key[7];
    buf[12] = x12 + (u32)counter; buf[13] = x13 + (u32)(counter >> 32); buf[14] = x14 + s0; buf[15] = x15 + s1;
    ++counter;
    pos = 0;
  }
  PVW_HD u32 next_u32() {
    if (pos == 16) refill();
    // select without dynamic register indexing (runtime-indexed arrays go to scratch)
    u32 v = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) v = (i == pos) ? buf[i] : v;
    ++pos;
    return v;
  }
  PVW_HD u64 next_u64() {
    u64 lo = next_u32();
    u64 hi = next_u32();
    return lo | (hi << 32);
  }
};

// sample_vec_cbd (src/sampling/uniform.rs:27-70): `half` selects the variance-0.5 special
// case (:38-44); otherwise v = (usize)variance >= 1 and the u128 bit pool of :45-68.
// Writes l coefficients through `emit(index, value)`.
template <class Emit>
PVW_HD void sample_cbd_poly(ChaChaRng& g, u32 l, bool half, u32 v, Emit emit) {
  if (half && g.pos == 16) {
    // block-aligned fast path (same words as the sequential draws below): coefficient 8b+i of a
    // stream uses words 2i, 2i+1 of block b -- statically indexed, no per-draw bookkeeping
    for (u32 s0 = 0; s0 < l; s0 += 8) {
      g.refill();
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (s0 + i < l) emit(s0 + i, (i64)(g.buf[2 * i] & 1) - (i64)(g.buf[2 * i + 1] & 1));
      const u32 used = (l - s0) < 8 ? 2 * (l - s0) : 16;
      g.pos = (int)used;
    }
    return;
  }
  if (half) {
    for (u32 s = 0; s < l; ++s) {
      i64 b1 = g.next_u32() & 1;
      i64 b2 = g.next_u32() & 1;
      emit(s, b1 - b2);
    }
    return;
  }
  u32 nbits = 4 * v;  // <= 64
  u64 mask_add = ((~(u64)0) >> (64 - nbits)) >> (2 * v);
  // pool kept as two 64-bit halves
  u64 p_lo = 0, p_hi = 0;
  u32 pool_n = 0;
  for (u32 s = 0; s < l; ++s) {
    if (pool_n < nbits) {
      u64 w = g.next_u64();
      // pool |= w << pool_n   (pool_n < 64 here because nbits <= 64)
      if (pool_n == 0) {
        p_lo |= w;
      } else {
        p_lo |= w << pool_n;
        p_hi |= w >> (64 - pool_n);
      }
      pool_n += 64;
    }
    u64 add = p_lo & mask_add;
    // mask_sub = mask_add << 2v spans into the high half only when nbits == 64 (never: 4v<=64 keeps it in the low word)
    u64 sub = (2 * v == 64) ? 0 : ((p_lo >> (2 * v)) & mask_add);
#if defined(__HIP_DEVICE_COMPILE__)
    i64 val = (i64)__popcll(add) - (i64)__popcll(sub);
#else
    i64 val = (i64)__builtin_popcountll(add) - (i64)__builtin_popcountll(sub);
#endif
    emit(s, val);
    // pool >>= nbits
    if (nbits == 64) {
      p_lo = p_hi;
      p_hi = 0;
    } else {
      p_lo = (p_lo >> nbits) | (p_hi << (64 - nbits));
      p_hi >>= nbits;
    }
    pool_n -= nbits;
  }
}

// sample_uniform_coefficients (src/sampling/uniform.rs:5-22): uniform in [-bound, bound],
// rejection sampling of a bit_length(2*bound+1)-bit value assembled from u32 words with
// the top word shifted down (the shape of num-bigint's gen_biguint_below).  bound < 2^62.
template <class Emit>
PVW_HD void sample_uniform_poly(ChaChaRng& g, u32 l, u64 bound, Emit emit) {
  u64 range = 2 * bound + 1;
#if defined(__HIP_DEVICE_COMPILE__)
  u32 bits = 64 - (u32)__clzll((long long)range);
#else
  u32 bits = 64 - (u32)__builtin_clzll(range);
#endif
  u32 digits = bits / 32, rem = bits % 32;
  u32 nwords = digits + (rem ? 1 : 0);
  if (nwords == 1) {
    // one word per draw (bounds below 2^31): walk the block with static word indices; rejected
    // draws are skipped exactly as the sequential loop below would skip them
    const u32 sh = rem ? 32 - rem : 0;
    u32 s = 0;
    while (s < l) {
      if (g.pos == 16) g.refill();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i >= g.pos && s < l) {
          const u32 w = g.buf[i] >> sh;
          g.pos = i + 1;
          if (w < (u32)range) {
            emit(s, (i64)w - (i64)bound);
            ++s;
          }
        }
      }
    }
    return;
  }
  for (u32 s = 0; s < l; ++s) {
    u64 v;
    do {
      u32 w0 = g.next_u32(), w1 = 0;
      if (nwords > 1) w1 = g.next_u32();
      if (digits == 0) w0 >>= 32 - rem;
      else if (digits == 1 && rem) w1 >>= 32 - rem;
      v = (u64)w0 | ((u64)w1 << 32);
    } while (v >= range);
    emit(s, (i64)v - (i64)bound);
  }
}

// l uniform residues in [0, q): 64-bit draws shifted to bit_length(q) bits, rejection.
template <class Emit>
PVW_HD void sample_residues_poly(ChaChaRng& g, u32 l, u64 q, Emit emit) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 sh = (u32)__clzll((long long)q);
#else
  u32 sh = (u32)__builtin_clzll(q);
#endif
  for (u32 s = 0; s < l;) {
    u64 v = g.next_u64() >> sh;
    if (v < q) emit(s++, v);
  }
}

}  // namespace pvw
