// pvw_decode_wave.h -- device code of the wave-cooperative gadget decode (decode_scalar_pvw_rns, decryption.rs:10-247):
// big integers one 64-bit word per lane, used by decode_chain_kernel (pvw_decode_kernels.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "pvw_arith.h"
#include "pvw_decode.h"
#include "pvw_dev.h"
#include "pvw_kernels.h"

namespace pvw {

// ------------------------------------------------------------------------------------
// decode, wave-cooperative form: ONE WAVE per ciphertext.  A big integer lives one 64-bit word
// per lane (word w in lane w), an RNS value one limb per lane; every step of pvw_decode.h's
// algorithm becomes "per-lane column sums + a short cross-lane carry loop":
//   lift      x = sum_i t_i * (Q/q_i) - kq*Q          L broadcast steps, columns of 3 words
//   to RNS    r_limb = sum_j x_j * 2^(64 j) mod q      W broadcast steps, lazy accumulator
//   divide    q^ = floor(N * floor(B^(W+1)/d) / B^(W+1)) from the top W+2 columns only, then
//             at most two corrections against the exact remainder (no digit-serial long division)
// Needs L <= 64 and W + 2 <= 63 (Q up to ~3900 bits); otherwise launch_decode uses decode_kernel.
// ------------------------------------------------------------------------------------
struct WaveBN {
  u64 x;   // this lane's word
};
__device__ __forceinline__ u64 shfl_up_u64(u64 v, int delta, u32 lane) {
  u32 lo = __shfl_up((u32)v, delta), hi = __shfl_up((u32)(v >> 32), delta);
  u64 r = ((u64)hi << 32) | lo;
  return lane >= (u32)delta ? r : 0;
}
// word of lane i (i wave-uniform) as a scalar broadcast: no LDS round trip
__device__ __forceinline__ u64 readlane_u64(u64 v, u32 i) {
  u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, (int)i), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), (int)i);
  return ((u64)hi << 32) | lo;
}
// whole-wave shifts by one lane as DPP moves (gfx9 wave_shr / wave_shl): no LDS crossbar round trip
__device__ __forceinline__ u32 lane_up1(u32 v) {      // lane i <- lane i-1, lane 0 <- 0
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ u32 lane_down1(u32 v) {    // lane i <- lane i+1, lane 63 <- 0
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ u64 lane_up1_u64(u64 v) { return ((u64)lane_up1((u32)(v >> 32)) << 32) | lane_up1((u32)v); }
__device__ __forceinline__ u64 lane_down1_u64(u64 v) { return ((u64)lane_down1((u32)(v >> 32)) << 32) | lane_down1((u32)v); }
// columns (c0 + c1*B + c2*B^2 at weight lane) -> one word per lane
__device__ __forceinline__ u64 wave_normalize(u64 c0, u64 c1, u64 c2, u32 lane) {
  u64 b = lane_up1_u64(c1), c = lane_up1_u64(lane_up1_u64(c2));
  u64 s = c0 + b;
  u32 k = s < b;
  s += c;
  k += s < c;
  while (__ballot(k != 0)) {
    u32 kin = lane_up1(k);
    s += kin;
    k = s < kin;
  }
  return s;
}
// x - y for x >= y (both one word per lane)
__device__ __forceinline__ u64 wave_sub(u64 x, u64 y, u32 lane) {
  u64 d = x - y;
  u32 b = x < y;
  while (__ballot(b != 0)) {
    u32 bin = lane_up1(b);
    b = d < bin;
    d -= bin;
  }
  return d;
}
// three-way compare of two lane-distributed integers: >0, 0, <0
__device__ __forceinline__ int wave_cmp(u64 x, u64 y) {
  unsigned long long g = __ballot(x > y), l = __ballot(x < y);
  return g > l ? 1 : (g == l ? 0 : -1);
}
__device__ __forceinline__ void col_mac(u64& c0, u64& c1, u64& c2, u64 a, u64 b) {
  u128 p = (u128)a * b;
  u64 lo = (u64)p, hi = (u64)(p >> 64);
  c0 += lo;
  u64 k = c0 < lo;
  hi += k;               // hi <= 2^64 - 2, cannot wrap
  c1 += hi;
  c2 += c1 < hi;
}

// acc += sum_{i<n} sc(i) * ld(i): groups of four with the next group's table words already in flight
// (the compiler does not unroll a loop around the asm MAC by itself, and a lone wave would eat the
// full LDS latency on every term)
template <typename LoadF, typename ScalF>
__device__ __forceinline__ void mac_loop4(Acc& acc, u32 n, LoadF ld, ScalF sc) {
  u32 i = 0;
  if (n >= 4) {
    u64 m0 = ld(0), m1 = ld(1), m2 = ld(2), m3 = ld(3);
    for (; i + 8 <= n; i += 4) {
      const u64 n0 = ld(i + 4), n1 = ld(i + 5), n2 = ld(i + 6), n3 = ld(i + 7);
      acc_mac_dev(acc, sc(i), m0);
      acc_mac_dev(acc, sc(i + 1), m1);
      acc_mac_dev(acc, sc(i + 2), m2);
      acc_mac_dev(acc, sc(i + 3), m3);
      m0 = n0; m1 = n1; m2 = n2; m3 = n3;
    }
    acc_mac_dev(acc, sc(i), m0);
    acc_mac_dev(acc, sc(i + 1), m1);
    acc_mac_dev(acc, sc(i + 2), m2);
    acc_mac_dev(acc, sc(i + 3), m3);
    i += 4;
  }
  for (; i < n; ++i) acc_mac_dev(acc, sc(i), ld(i));
}

struct WaveDecodeCtx {
  const DecodeTables& t;
  const u64* qiL;    // LDS copy of t.qi    [L][W]
  const u64* powL;   // LDS copy of t.pow64T [W][L]
  u64* xs;     // per-wave LDS scratch, 64 words
  u32 lane;
  u32 W, L;
  Mod m;       // this lane's limb modulus (lanes >= L: limb 0, masked out by `limb_on`)
  bool limb_on, word_on;
  u64 Qw, halfQw;
};

// CRT lift of one residue per limb-lane to x in [0, Q), centred: |value| one word per lane, sign returned
template <bool CENTRE = true>
__device__ __forceinline__ u64 wave_lift_centered(const WaveDecodeCtx& c, u64 res, bool& neg) {
  const DecodeTables& t = c.t;
  const u32 lane = c.lane;
  u64 ti = c.limb_on ? mulmod_shoup(res, t.inv[lane], t.invp[lane], c.m.q) : 0;
  // fixed-point t_i / q_i (error < 2 ulp, from below) to predict how many multiples of Q the sum holds
  u64 f = c.limb_on ? ti * c.m.ratio_hi + mulhi64(ti, c.m.ratio_lo) : 0;
  // wave sum of the 64-bit fractions as a 128-bit value
  u64 flo = f, fhi = 0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    u64 olo = ((u64)__shfl_xor((u32)(flo >> 32), d) << 32) | __shfl_xor((u32)flo, d);
    u64 ohi = ((u64)__shfl_xor((u32)(fhi >> 32), d) << 32) | __shfl_xor((u32)fhi, d);
    flo += olo;
    fhi += ohi + (flo < olo);
  }
  const u64 kq = fhi;   // floor(sum t_i/q_i) or one less
  u64 c0, c1, c2;
  {
    const u64* qp = c.qiL + (c.word_on ? lane : 0);      // lanes >= W compute a discarded column
    Acc acc;
    acc_zero(acc);
    const u32 Wq = c.W;
    mac_loop4(acc, c.L, [&](u32 i) { return qp[i * Wq]; }, [&](u32 i) { return readlane_u64(ti, i); });
    acc_words(acc, c0, c1, c2);
    if (!c.word_on) c0 = c1 = c2 = 0;
  }
  u64 x = wave_normalize(c0, c1, c2, lane);
  // subtract kq * Q
  {
    u128 p = c.word_on ? (u128)kq * c.Qw : 0;
    u64 y = wave_normalize((u64)p, (u64)(p >> 64), 0, lane);
    x = wave_sub(x, y, lane);
  }
  while (wave_cmp(x, c.Qw) >= 0) x = wave_sub(x, c.Qw, lane);
  neg = false;
  if (CENTRE && wave_cmp(x, c.halfQw) > 0) {       // decryption.rs:145-151
    x = wave_sub(c.Qw, x, lane);
    neg = true;
  }
  return x;
}
// floor(N / d) and N mod d via the reciprocal mu = floor(B^(W+1)/d): N and d (one word per lane, dw; dn = its
// significant words) are read lane-to-lane (v_readlane); only the reciprocal is a table, W+2 words zero-padded to
// 2W+2 so that the column loop has no bounds test.  N < B^W.  Returns quotient in q, remainder in r.
__device__ __forceinline__ void wave_divmod2(const WaveDecodeCtx& c, u64 n, const u64* muP, u64 dw, u32 dn,
                                             u64& q, u64& r) {
  const u32 lane = c.lane, W = c.W;
  // top W+2 columns of N * mu: lane v holds column W-1+v = sum_i N_i * mu[W-1+v-i]; idle lanes walk the zero pad
  const u64* mp = muP + (lane < W + 2 ? W - 1 + lane : 2 * W + 1);
  u64 c0, c1, c2;
  Acc acc;
  acc_zero(acc);
  mac_loop4(acc, W, [&](u32 i) { return *(mp - i); }, [&](u32 i) { return readlane_u64(n, i); });
  acc_words(acc, c0, c1, c2);
  u64 p = wave_normalize(c0, c1, c2, lane);
  {
    u64 pd = lane_down1_u64(lane_down1_u64(p));
    q = lane < W ? pd : 0;
  }
  // remainder N - q*d (q <= true quotient): column `lane` = sum_{j < dn} q[lane-j] * d[j]
  acc_zero(acc);
  u64 qj = q;
  for (u32 j = 0; j < dn; ++j) {
    acc_mac_dev(acc, qj, readlane_u64(dw, j));
    qj = lane_up1_u64(qj);
  }
  acc_words(acc, c0, c1, c2);
  u64 qd = wave_normalize(c0, c1, c2, lane);
  qd = lane < W ? qd : 0;
  r = wave_sub(n, qd, lane);
  while (wave_cmp(r, dw) >= 0) {                   // at most two corrections
    r = wave_sub(r, dw, lane);
    u64 one = lane == 0 ? 1 : 0;
    q = wave_normalize(q + one, (q + one) < one ? 1 : 0, 0, lane);
  }
}

// decode, lifted-chain form: WPC waves per ciphertext, CPW ciphertexts per workgroup.
// The reference's chain noise_i = round((noise_{i+1} - tmp_i) / Delta) (decryption.rs:44-48) is exact
// integer arithmetic mod Q, so it can be carried in big-integer form throughout: the l+1 CRT lifts it
// needs (tmp_0..tmp_{l-2}, the Horner value, z_0) do not depend on the chain and are spread over the
// WPC waves; after one barrier wave 0 walks the chain with one short-divisor division per step and NO
// conversion back to RNS.  Serial big steps per ciphertext: l divisions (instead of l+1 lifts +
// l divisions + l RNS conversions of the first, one-wave-per-ciphertext form: 0.20 -> 0.10 ms per 1024 ciphertexts at
// 2074-bit Q; two or eight waves per ciphertext for the lifts measured no better).
// blk = the workgroup's index among the decode workgroups; dws = its dynamic LDS.
// (Measured and dropped in round 3: the inverse transform of the noisy polynomial folded in here -- one limb per lane of the
// ciphertext's first wave -- instead of the INTT launch between the inner products and the decode.  The step did not
// move (499.8 vs 497 us at the config-5 shard: the decode grew by what the launch had cost) and the extra registers
// ended the co-residency with decrypt_mac that the overlapped batch path lives on (config 5 in full: 4.05 vs 3.10 ms).)
template <int WPC>
__device__ __forceinline__ void decode_chain_body(const u64* __restrict__ noisy, u64* __restrict__ out,
                                                  u32 count, u32 cpw_dbg, const DecodeTables& t, u32 blk, u64* dws) {
  const u32 cpw = cpw_dbg & 0xffff;
  const u32 dbg = PVW_TUNING ? (cpw_dbg >> 16) : 0;        // tuning build, dbg != 0: timing experiment, out[] = cycle counts
  const u64 tk0 = dbg ? clock64() : 0;
  // LDS: CRT table [L][W] | two reciprocals, 2W+2 words each | per ciphertext: lifts [l+1][64] + signs, residues [L][l]
  //      | per wave: 64-word scratch
  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32 W = t.W, L = t.L, l = t.ell;
  const u32 nw = cpw * WPC;
  u64* qiL = dws;
  u64* smallL = dws + (size_t)L * W;                     // mu_dp | mu_td, W+2 words each, zero-padded to 2W+2
  u64* ctbase = smallL + (size_t)2 * (2 * W + 2);
  const size_t ct_words = (size_t)(l + 1) * 64 + (size_t)L * l;
  const u32 cw = wave / WPC, wsub = wave % WPC;          // ciphertext within the workgroup, wave within it
  u64* Tl = ctbase + (size_t)cw * ct_words;              // [l+1][64]
  u64* zs = Tl + (size_t)(l + 1) * 64;                   // [L][l]
  u64* xs = ctbase + (size_t)cpw * ct_words + (size_t)wave * 64;
  for (u32 x = threadIdx.x; x < L * W; x += nw * 64) qiL[x] = t.qi[x];
  for (u32 x = threadIdx.x; x < 2 * W + 2; x += nw * 64) {
    smallL[x] = x < W + 2 ? t.mu_dp[x] : 0;
    smallL[(2 * W + 2) + x] = x < W + 2 ? t.mu_td[x] : 0;
  }
  const u32 d = blk * cpw + cw;
  const bool live = d < count;                           // uniform over the ciphertext's waves
  if (live)
    for (u32 x = wsub * 64 + lane; x < L * l; x += WPC * 64) zs[x] = noisy[(size_t)d * L * l + x];
  __syncthreads();
  WaveDecodeCtx c{t, qiL, nullptr, xs, lane, W, L, t.mods[lane < L ? lane : 0], lane < L, lane < W,
                  lane < W ? t.Q[lane] : 0, lane < W ? t.halfQ[lane] : 0};
  const u64* z = zs + (size_t)(c.limb_on ? lane : 0) * l;
  const u64 dm = t.dmod[c.limb_on ? lane : 0], dmp = t.dmodp[c.limb_on ? lane : 0];
  const u64 q = c.m.q;
  auto tmp = [&](u32 i) -> u64 { return submod(mulmod_shoup(z[i], dm, dmp, q), z[i + 1], q); };   // :19-27
  // ---- phase 1: the l+1 lifts, item = 0..l-2: tmp_i in [0,Q); l-1: Horner value, centred; l: z_0 in [0,Q)
  bool hneg = false;
  if (live) {
    for (u32 item = wsub; item <= l; item += WPC) {
      bool ng = false;
      u64 x;
      if (item + 1 < l) {
        x = wave_lift_centered<false>(c, tmp(item), ng);
      } else if (item == l - 1) {
        u64 h = tmp(0);                                  // Horner over tmp_0 .. tmp_{l-2} (:30-33)
        for (u32 i = 1; i + 1 < l; ++i) h = addmod(mulmod_shoup(h, dm, dmp, q), tmp(i), q);
        x = wave_lift_centered<true>(c, h, ng);
        if (lane == 0) Tl[(size_t)(l - 1) * 64 + 63] = ng ? 1 : 0;    // word 63 is never a value word (W + 2 <= 64)
      } else {
        x = wave_lift_centered<false>(c, z[0], ng);
      }
      if (lane < 63 || item != l - 1) Tl[(size_t)item * 64 + lane] = (lane < W) ? x : 0;
    }
  }
  __syncthreads();
  const u64 tk1 = dbg ? clock64() : 0;
  if (!live || wsub != 0) return;
  // ---- phase 2: the chain, one wave
  const u64 Qw = c.Qw;
  const u64 dpw = lane < W ? t.dpow[lane] : 0, hdw = lane < W ? t.half_dpow[lane] : 0;
  const u64 tdw = lane < W ? t.td[lane] : 0, dlw = lane < W ? t.delta[lane] : 0;
  const unsigned long long bdp = __ballot(dpw != 0), btd = __ballot(tdw != 0);
  const u32 dn_dp = bdp ? 64 - __builtin_clzll(bdp) : 1, dn_td = btd ? 64 - __builtin_clzll(btd) : 1;
  hneg = Tl[(size_t)(l - 1) * 64 + 63] != 0;
  u64 x = lane < 63 ? Tl[(size_t)(l - 1) * 64 + lane] : 0;
  // reduce_modulo_poly (:154-178): noise_{l-1} = (nm, nneg)
  u64 qq, r;
  wave_divmod2(c, x, smallL, dpw, dn_dp, qq, r);
  bool nneg = hneg;
  if (__ballot(r != 0) == 0) nneg = false;
  if (wave_cmp(r, hdw) > 0) {
    r = wave_sub(dpw, r, lane);
    nneg = !nneg;
  }
  u64 nm = r;
  const u64 tk2 = dbg ? clock64() : 0;
  // (a - b) mod Q, centred, for a given as signed magnitude (am, aneg), |a| < Q, and b in [0, Q)
  auto sub_centre = [&](u64 am, bool aneg, u64 b, bool& vneg) -> u64 {
    const bool azero = __ballot(am != 0) == 0;
    u64 a = (aneg && !azero) ? wave_sub(Qw, am, lane) : am;       // a mod Q
    u64 v = wave_cmp(a, b) >= 0 ? wave_sub(a, b, lane) : wave_sub(Qw, wave_sub(b, a, lane), lane);
    vneg = false;
    if (wave_cmp(v, c.halfQw) > 0) {                              // decryption.rs:145-151
      v = wave_sub(Qw, v, lane);
      vneg = true;
    }
    return v;
  };
  // noise_i = round((noise_{i+1} - tmp_i) / Delta), i = l-2 .. 0   (:44-48, :180-207)
  for (u32 i = l - 1; i-- > 0;) {
    bool pneg;
    const u64 ta = dbg >= 4 ? clock64() : 0;
    u64 p = sub_centre(nm, nneg, Tl[(size_t)i * 64 + lane], pneg);
    const u64 tb = dbg >= 4 ? clock64() : 0;
    u64 hi = p >> 63, lo2 = p << 1;                     // 2|p| + Delta
    u64 sm = lo2 + dlw;
    u64 num = wave_normalize(sm, hi + (sm < dlw), 0, lane);
    const u64 tc = dbg >= 4 ? clock64() : 0;
    wave_divmod2(c, num, smallL + (2 * W + 2), tdw, dn_td, qq, r);
    if (dbg >= 4 && i == l - 3) {                       // timing experiment: one step of the chain in three parts
      const u64 td2 = clock64();
      if (lane == 0) out[d] = dbg == 4 ? (tb - ta) : (dbg == 5 ? (tc - tb) : (td2 - tc));
      return;
    }
    const bool qzero = __ballot(qq != 0) == 0;
    nm = qq;
    nneg = pneg && !qzero;
  }
  // plaintext = -z_0 - noise_0 (:51-53) = ((Q - z_0) mod Q) - noise_0, then extract_constant_term_as_u64 (:226-247)
  bool vneg;
  u64 v;
  {
    // -(z_0 + noise_0): first s = (noise_0 + z_0) centred as (noise_0 - (Q - z_0 mod Q)), then negate
    u64 z0 = Tl[(size_t)l * 64 + lane];
    const bool z0zero = __ballot(z0 != 0) == 0;
    u64 mz0 = z0zero ? 0 : wave_sub(Qw, z0, lane);      // (-z_0) mod Q
    // (-z_0 - noise_0) mod Q = ((-z_0) - noise_0) mod Q: swap roles: a = -z_0 in [0,Q), b = noise_0 mod Q
    const bool nzero = __ballot(nm != 0) == 0;
    u64 b = (nneg && !nzero) ? wave_sub(Qw, nm, lane) : nm;
    v = wave_cmp(mz0, b) >= 0 ? wave_sub(mz0, b, lane) : wave_sub(Qw, wave_sub(b, mz0, lane), lane);
    vneg = false;
    if (wave_cmp(v, c.halfQw) > 0) {
      v = wave_sub(Qw, v, lane);
      vneg = true;
    }
  }
  if (dbg) {
    const u64 tk3 = clock64();
    if (lane == 0) out[d] = dbg == 1 ? (tk1 - tk0) : (dbg == 2 ? (tk2 - tk1) : (tk3 - tk2));
    return;
  }
  const bool vzero = __ballot(v != 0) == 0;
  u64 result;
  if (vneg && !vzero) {
    const bool hiw = __ballot(lane > 0 && v != 0) != 0;
    const u64 v0 = ((u64)__shfl((u32)(v >> 32), 0) << 32) | (u32)__shfl((u32)v, 0);
    if (!hiw && v0 <= 1000) {                       // small negative -> 0 (:233-235)
      if (lane == 0) out[d] = 0;
      return;
    }
    v = wave_sub(Qw, v, lane);                      // (v + Q) % Q = Q - |v|
  }
  const bool hiw2 = __ballot(lane > 0 && v != 0) != 0;
  result = hiw2 ? 0 : v;
  if (lane == 0) out[d] = result;
}

}  // namespace pvw
