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
// decode, wave-cooperative form.  A big integer lives one 64-bit word per lane (word w in lane w), an RNS value one limb
// per lane; every step of pvw_decode.h's algorithm becomes "per-lane column sums + a short cross-lane carry loop":
//   lift      x = sum_i t_i * (Q/q_i) - kq*Q          L broadcast steps, columns of 3 words
//   divide    q^ = floor(N * floor(B^(W+1)/d) / B^(W+1)) from the top W+2 columns only, then
//             at most two corrections against the exact remainder (no digit-serial long division)
// Needs L <= 64 and W + 2 <= 63 (Q up to ~3900 bits); otherwise launch_decode uses decode_kernel.
// That is the GENERAL path.  In front of it sit short cuts for what a decrypt actually produces -- values that are
// noise-sized against Q -- each a guess from a few residues that is proven on every limb before it is used
// (small_candidates / small_confirm, small_top, small_chain below); an input for which a proof fails takes the general
// path, so the result is the same for every input.
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
  const u64* powL;   // LDS copy of the first four rows of t.pow64T, [4][64]: 2^(64 w) mod q_lane
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
// The same lift for values that are SMALL against Q.  For a well-formed ciphertext the chain inputs tmp_i = z_i*Delta -
// z_{i+1} and z_0 are noise-sized (a few words: noise times Delta) while Q has W words, and the full lift above spends
// L broadcast steps and a W-word reduction on each.  A value whose centred representative v satisfies |v| <= P/2,
// P = q_0..q_{n-1} the product of the first n = gar_n <= 4 moduli, is determined by its first n residues alone:
//   small_candidates  one lane per chain input of this wave: Garner's mixed-radix digits from the n residues, v centred
//                     modulo P as four words + sign, parked in the wave's LDS scratch (five words per input)
//   small_confirm     per input, one limb per lane: |v| mod q_limb from the power rows, sign applied, must equal the
//                     residue on EVERY limb -- then v = x mod Q centred (|v| <= P/2 < Q/2, the representative is
//                     unique) and x in [0, Q) is returned one word per lane; otherwise false, nothing assumed, and the
//                     caller takes the full lift.  Results are identical either way.
__device__ __forceinline__ void small_candidates(const WaveDecodeCtx& c, const u64* zs, u32 l, u32 item, u64* park) {
  const DecodeTables& t = c.t;
  u64 r[4] = {0, 0, 0, 0};
#pragma unroll
  for (u32 j = 0; j < 4; ++j) {
    if (j < t.gar_n) {
      const u64 q = t.mods[j].q;
      const u64* z = zs + (size_t)j * l;
      // this input's residue at limb j: tmp_item (decryption.rs:19-27) or z_0 (item == l)
      r[j] = item < l ? submod(mulmod_shoup(z[item], t.dmod[j], t.dmodp[j], q), z[item + 1], q) : z[0];
    }
  }
  garner_small(t, r, park);
}
__device__ __forceinline__ bool small_confirm(const WaveDecodeCtx& c, u64 res, u64* park, u64& x) {
  const u32 lane = c.lane;
  const u64 v0 = park[0], v1 = park[1], v2 = park[2], v3 = park[3];
  const bool ng = park[4] != 0;
  const u64* pw = c.powL + (c.limb_on ? lane : 0);
  u128 sum = (u128)v0 * pw[0] + (u128)v1 * pw[64];        // at most four terms < 2^126 each (q < 2^62)
  if (v2) sum += (u128)v2 * pw[128];
  if (v3) sum += (u128)v3 * pw[192];
  u64 sres = reduce128((u64)sum, (u64)(sum >> 64), c.m);
  if (ng && sres) sres = c.m.q - sres;
  if (__ballot(c.limb_on && sres != res)) return false;
  if (lane == 0) park[4] = (ng ? 1 : 0) | 2;               // confirmed: the chain may use the signed magnitude as it stands
  x = lane == 0 ? v0 : (lane == 1 ? v1 : (lane == 2 ? v2 : (lane == 3 ? v3 : 0)));
  if (ng) x = wave_sub(c.Qw, x, lane);                     // the representative in [0, Q)
  return true;
}

// noise_{l-1} without lifting the Horner value.  The reference lifts H = sum_i tmp_i Delta^(l-2-i) (= z_0 Delta^(l-1) -
// z_{l-1}, the sum telescopes) to (-Q/2, Q/2] and reduces it modulo Delta^(l-1), centred (decryption.rs:30-37, :154-178):
// a W-word lift and a W-word division for a result that, in a well-formed ciphertext, is a noise value.  Here: GUESS
// g = the centred remainder of tmp_{l-2} modulo Delta (the same noise value when the ciphertext is well formed), then
// PROVE it: e_i = (H - g) / Delta^(l-1) mod q_i on every limb must be one small integer e (|e| <= q_0/2, taken from limb
// 0 and compared on all the others).  Then H = e Delta^(l-1) + g modulo Q, the right-hand side is below Q/2 in magnitude
// (the host checked 2^61 Delta^(l-1) + Delta < Q/2 before setting hs_on), so it IS the centred H, and with |g| <= Delta/2 <
// Delta^(l-1)/2 the reference's reduction gives exactly g.  false: nothing assumed, the caller lifts and divides.
// qv = |round(-tmp_{l-2} / Delta)| comes from the caller (the top lane's chain step on a zero input).
__device__ __forceinline__ bool small_top(const WaveDecodeCtx& c, const u64* z, const u64* cand, u32 l, u64 qv, SmallVal& top) {
  const DecodeTables& t = c.t;
  const u32 lane = c.lane, L = c.L;
  if (!small_top_guess(t.sc, cand + (size_t)(l - 2) * 5, qv, top)) return false;
  // every limb: e_i = (H_i - g_i) / Delta^(l-1) mod q_i; all of them one small e
  const u32 li = c.limb_on ? lane : 0;
  const u64* pw = c.powL + li;
  const u64 ei = small_top_quotient(top, z[0], z[l - 1], t.dpm[li], t.dpm[L + li], t.dpm[2 * L + li], t.dpm[3 * L + li], pw[0], pw[64], pw[128], c.m);
  const u64 want = small_top_expected(readlane_u64(ei, 0), t.mods[0].q, c.m);
  return __ballot(c.limb_on && want != ei) == 0;
}
// The whole chain for noise-sized values.  The steps are serial by definition -- noise_i needs noise_{i+1} -- but for
// a well-formed ciphertext noise_{i+1} is ~2^-100 of tmp_i and moves round((noise_{i+1} - tmp_i)/Delta) only on a
// rounding boundary.  So lane i takes step i with a guess for its input (first pass: 0; later passes: what the lane
// above produced in the pass before; the top lane has the true noise_{l-1}), all lanes at once, until a pass
// reproduces the one before.  That fixed point IS the chain: the top lane's input is exact, hence its output, hence the
// input the next lane used in the confirming pass, and so on down.  The top input is either given (top_known: the
// caller reduced the lifted Horner value) or found here from the first pass (small_top; the top lane's output on a
// zero input is then already its true output, since noise_{l-1} - tmp_{l-2} is an exact multiple of Delta).
// false (nothing assumed) when a step does not fit the short form, the top input cannot be proven or the passes do not
// settle in `max_pass`; `top` then still holds a proven noise_{l-1} if *top_known came back true.
__device__ __forceinline__ bool small_chain(const WaveDecodeCtx& c, const u64* z, const u64* cand, u32 l, bool* top_known,
                                            SmallVal& top, SmallVal& out) {
  const u32 lane = c.lane;
  const u64* sc = c.t.sc;
  const bool mine = lane + 1 < l;                          // steps 0 .. l-2
  const u64* cb = cand + (size_t)(mine ? lane : 0) * 5;
  u64 q = 0;
  bool qneg = false;
  const int max_pass = 4;
  for (int pass = 0; pass < max_pass; ++pass) {
    SmallVal a;
    a.w0 = lane_down1_u64(q);                              // lane i <- lane i+1
    a.w1 = a.w2 = 0;
    a.neg = lane_down1((u32)qneg) != 0;
    if (lane + 2 == l && *top_known) a = top;
    u64 nq;
    bool nneg;
    const bool ok = small_chain_step(sc, a, cb, nq, nneg);
    if (__ballot(mine && !ok)) return false;
    const bool same = nq == q && nneg == qneg;
    q = nq;
    qneg = nneg;
    if (!*top_known) {                                     // first pass, every input zero
      if (!c.t.hs_on || !small_top(c, z, cand, l, readlane_u64(q, l - 2), top)) return false;
      *top_known = true;
      continue;
    }
    if (pass > 0 && __ballot(mine && !same) == 0) {
      out.w0 = readlane_u64(q, 0);
      out.w1 = out.w2 = 0;
      out.neg = __builtin_amdgcn_readlane((int)qneg, 0) != 0;
      return true;
    }
  }
  return false;
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
  // only N's significant words contribute: the chain's numerators are a few words long for a well-formed ciphertext
  const unsigned long long nzw = __ballot(n != 0);
  const u32 ns = nzw ? 64 - (u32)__builtin_clzll(nzw) : 0;
  mac_loop4(acc, ns, [&](u32 i) { return *(mp - i); }, [&](u32 i) { return readlane_u64(n, i); });
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
// integer arithmetic mod Q, so it can be carried in big-integer form throughout: the inputs it needs
// (tmp_0..tmp_{l-2}, z_0, the Horner value) do not depend on the chain and are settled by the WPC waves
// (drawn from a counter: a confirmed short-cut candidate or a full CRT lift each); after a barrier wave 0
// takes noise_{l-1} and the chain -- at once on noise-sized values (small_chain), else one short-divisor
// division per step on W-word integers -- with NO conversion back to RNS.
// Timings per 1024 ciphertexts at 2074-bit Q: the first, one-wave-per-ciphertext form with lifts, divisions and RNS
// conversions in series 0.20 ms; lifts spread over four waves 0.10-0.12 ms (two or eight waves no better); with the
// short cuts 0.02 ms on well-formed ciphertexts, 0.10 ms on uniform residues.
// blk = the workgroup's index among the decode workgroups; dws = its dynamic LDS.
// (Round 3, first attempt at the inverse transform of decrypt inside this kernel: one limb per lane of the
// ciphertext's first wave -- the step did not move and the extra registers ended the co-residency with decrypt_mac that
// the overlapped batch path lives on (config 5 in full: 4.05 vs 3.10 ms).  What ships is stage_inverse in an instance of
// its own, used only where the decode does not share the chip: pvw_decode_kernels.hip.)
// tables of the inverse transform when the decode does it itself (itw == nullptr: the input is in power basis already)
struct InverseTables {
  const u64* itw;    // [L][l]  psi^-bitrev(i)
  const u64* itwp;
  const u64* linv;   // [L]     l^-1 mod q
  const u64* linvp;
};
// Staging of one ciphertext's residues when they arrive in the NTT domain: the inverse transform of decrypt
// (decryption.rs:116) on the way into LDS.  ntt_inverse's butterflies (pvw_arith.h), one per thread and stage, l/2
// consecutive threads per polynomial: they load it, share a wave (l/2 divides 64), and a wave's LDS accesses execute in
// order, so neither the load nor the stages need a barrier.  The power-basis polynomial replaces the input in memory.
__device__ __forceinline__ void stage_inverse(const Mod* mods, const InverseTables& xf, u64* zs, u64* dst, u32 L, u32 l,
                                                        u32 first, u32 stride) {
  const u32 half = l >> 1, npair = L * half, lh = (u32)__builtin_ctz(half);       // l is a power of two
#pragma unroll 1
  for (u32 x = first; x < ((npair + 63) & ~63u); x += stride) {
    const bool on = x < npair;
    const u32 limb = on ? x >> lh : 0, b = x & (half - 1);
    const u64 q = mods[limb].q;
    u64* a = zs + (size_t)limb * l;
    if (on) {                                              // the polynomial's l/2 threads bring it in themselves
      const v2u64 in = reinterpret_cast<const v2u64*>(dst)[x];
      a[2 * b] = in.x;
      a[2 * b + 1] = in.y;
    }
    __builtin_amdgcn_wave_barrier();
    const u64* tw = xf.itw + (size_t)limb * l;
    const u64* twp = xf.itwp + (size_t)limb * l;
    u32 ls = 0;                                            // step = 1 << ls, mm = half >> ls
#pragma unroll 1
    for (u32 mm = half; mm >= 1; mm >>= 1, ++ls) {
      const u32 i = b >> ls, j = (i << (ls + 1)) + (b & ((1u << ls) - 1));
      if (on) {
        const u64 w = tw[mm + i], wp = twp[mm + i];
        const u64 u = a[j], v = a[j + (1u << ls)];
        a[j] = addmod(u, v, q);
        a[j + (1u << ls)] = mulmod_shoup(submod(u, v, q), w, wp, q);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (on) {
      const u64 li = xf.linv[limb], lip = xf.linvp[limb];
      const u64 r0 = mulmod_shoup(a[2 * b], li, lip, q), r1 = mulmod_shoup(a[2 * b + 1], li, lip, q);
      a[2 * b] = r0;
      a[2 * b + 1] = r1;
      reinterpret_cast<v2u64*>(dst)[x] = v2u64{r0, r1};
    }
  }
}
template <int WPC>
__device__ __forceinline__ void decode_chain_body(u64* __restrict__ noisy, u64* __restrict__ out, u32 count, u32 cpw_dbg,
                                                  const DecodeTables& t, const InverseTables& xf, u32 blk, u64* dws) {
  const u32 cpw = cpw_dbg & 0xffff;
  const u32 dbg = PVW_TUNING ? ((cpw_dbg >> 16) & 0xff) : 0;
  const bool no_small = PVW_TUNING && (cpw_dbg >> 31);     // tuning build: PVW_DECODE_SMALL=0, every lift in full        // tuning build, dbg != 0: timing experiment, out[] = cycle counts
  const u64 tk0 = dbg ? clock64() : 0;
  // LDS: CRT table [L][W] | two reciprocals, 2W+2 words each | power rows [4][64] | per ciphertext: lifts [l+1][64] + signs,
  //      residues [L][l], short-cut candidates [l][5] + the work counter
  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32 W = t.W, L = t.L, l = t.ell;
  const u32 nw = cpw * WPC;
  u64* qiL = dws;
  u64* smallL = dws + (size_t)L * W;                     // mu_dp | mu_td, W+2 words each, zero-padded to 2W+2
  u64* gpowL = smallL + (size_t)2 * (2 * W + 2);         // 2^(64 w) mod q_lane, w < 4
  u64* ctbase = gpowL + 256;
  const size_t ct_words = (size_t)(l + 1) * 64 + (size_t)L * l + (size_t)5 * l + 2;
  const u32 cw = wave / WPC, wsub = wave % WPC;          // ciphertext within the workgroup, wave within it
  u64* Tl = ctbase + (size_t)cw * ct_words;              // [l+1][64]
  u64* zs = Tl + (size_t)(l + 1) * 64;                   // [L][l]
  u64* cand = zs + (size_t)L * l;                        // [l][5]: four words + sign per chain input
  u32* next_input = reinterpret_cast<u32*>(cand + (size_t)5 * l);
  // the residues and the power rows first: all the candidates need.  The tables of the full lifts and the general
  // divisions are staged by the other waves while each ciphertext's first wave works the candidates out.
  for (u32 x = threadIdx.x; x < 256; x += nw * 64) gpowL[x] = ((x & 63) < L && (x >> 6) < W) ? t.pow64T[(size_t)(x >> 6) * L + (x & 63)] : 0;
  const u32 d = blk * cpw + cw;
  const bool live = d < count;                           // uniform over the ciphertext's waves
  if (live) {
    if (xf.itw) stage_inverse(t.mods, xf, zs, noisy + (size_t)d * L * l, L, l, wsub * 64 + lane, WPC * 64);
    else
      for (u32 x = wsub * 64 + lane; x < L * l; x += WPC * 64) zs[x] = noisy[(size_t)d * L * l + x];
  }
  __syncthreads();
  const u64 tk0b = dbg ? clock64() : 0;
  WaveDecodeCtx c{t, qiL, gpowL, lane, W, L, t.mods[lane < L ? lane : 0], lane < L, lane < W,
                  lane < W ? t.Q[lane] : 0, lane < W ? t.halfQ[lane] : 0};
  const u64* z = zs + (size_t)(c.limb_on ? lane : 0) * l;
  const u64 dm = t.dmod[c.limb_on ? lane : 0], dmp = t.dmodp[c.limb_on ? lane : 0];
  const u64 q = c.m.q;
  auto tmp = [&](u32 i) -> u64 { return submod(mulmod_shoup(z[i], dm, dmp, q), z[i + 1], q); };   // :19-27
  // ---- phase 1: the l+1 lifts, item = 0..l-2: tmp_i in [0,Q); l-1: Horner value, centred; l: z_0 in [0,Q)
  // The l chain inputs (tmp_0..tmp_{l-2} and z_0) are noise-sized for a well-formed ciphertext: the ciphertext's first
  // wave works out a short-cut candidate for each, one lane per input; after a barrier the waves draw inputs from a
  // counter -- confirm the candidate or lift in full -- the last wave joining once it has lifted the Horner value (always
  // in full: it is of the order of Delta^(l-1)).
  bool hneg = false;
  const bool small_on = t.gar_n != 0 && !no_small;
  if (wsub != 0) {
    const u32 nst = (WPC - 1) * cpw * 64, me = (cw * (WPC - 1) + wsub - 1) * 64 + lane;
    for (u32 x = me; x < L * W; x += nst) qiL[x] = t.qi[x];
    for (u32 x = me; x < 2 * W + 2; x += nst) {
      smallL[x] = x < W + 2 ? t.mu_dp[x] : 0;
      smallL[(2 * W + 2) + x] = x < W + 2 ? t.mu_td[x] : 0;
    }
  }
  if (live) {
    if (wsub == 0) {
      const u64 tc0 = dbg == 9 ? clock64() : 0;
      if (lane == 0) { next_input[0] = 0; next_input[1] = 0; }
      if (small_on && lane < l) small_candidates(c, zs, l, lane + 1 < l ? lane : l, cand + (size_t)lane * 5);
      if (dbg == 9) {                                    // timing experiment: the candidates
        const u64 tc1 = clock64();
        if (lane == 0) out[d] = tc1 - tc0;
      }
    }
  }
  __syncthreads();
  u32* h_done = next_input + 1;                          // the Horner value sits lifted in its slot
  // Horner value (:30-33): sum_i tmp_i Delta^(l-2-i) telescopes to z_0 Delta^(l-1) - z_{l-1}; lifted in full, centred
  auto lift_horner = [&]() {
    const u64 tl0 = dbg == 10 ? clock64() : 0;
    bool ng = false;
    const u32 li = c.limb_on ? lane : 0;
    const u64 h = submod(mulmod_shoup(z[0], t.dpm[li], t.dpm[L + li], q), z[l - 1], q);
    const u64 x = wave_lift_centered<true>(c, h, ng);
    if (lane == 0) {
      Tl[(size_t)(l - 1) * 64 + 63] = ng ? 1 : 0;        // word 63 is never a value word (W + 2 <= 64)
      *h_done = 1;
    }
    if (lane < 63) Tl[(size_t)(l - 1) * 64 + lane] = (lane < W) ? x : 0;
    if (dbg == 10) {                                     // timing experiment: the Horner value's full lift
      const u64 tl1 = clock64();
      if (lane == 0) out[d] = tl1 - tl0;
    }
  };
  if (live) {
    bool try_small = small_on;
    bool first = true;
    // settle the next chain input: 1 = by its confirmed candidate, 0 = by a full lift, -1 = none left
    auto settle_next = [&]() -> int {
      u32 idx = 0;
      if (lane == 0) idx = atomicAdd(next_input, 1u);
      idx = (u32)__builtin_amdgcn_readfirstlane((int)idx);
      if (idx >= l) return -1;
      const u32 item = idx + 1 < l ? idx : l;            // slot l-1 of the lifts is the Horner value's
      const u64 tl0 = dbg == 8 ? clock64() : 0;
      const u64 res = item < l ? tmp(item) : z[0];
      bool done = false, ng = false;
      u64 x;
      if (try_small) {
        done = small_confirm(c, res, cand + (size_t)idx * 5, x);
        try_small = done;                                // one refusal: the rest of this wave's inputs go the long way
      }
      if (!done) x = wave_lift_centered<false>(c, res, ng);
      Tl[(size_t)item * 64 + lane] = (lane < W) ? x : 0;
      if (dbg == 8 && first && wsub == 0) {              // timing experiment: the first input the first wave settles
        const u64 tl1 = clock64();
        if (lane == 0) out[d] = tl1 - tl0;
      }
      first = false;
      return done ? 1 : 0;
    };
    if (wsub == WPC - 1) {
      // the last wave lifts the Horner value now -- unless the ciphertext looks well formed (its first input was settled
      // by its candidate) and noise_{l-1} can be had without it (small_top); should that fail, the first wave lifts it later
      const int r = (small_on && t.hs_on) ? settle_next() : 0;
      if (r != 1) lift_horner();
    }
    while (settle_next() >= 0) {}
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
  u64 qq, r;
  u64 nm = 0;
  bool nneg = false;
  // noise_{l-1} and the chain noise_i = round((noise_{i+1} - tmp_i) / Delta), i = l-2 .. 0 (:30-48, :154-207).
  // First on noise-sized values throughout: noise_{l-1} proven from the residues (small_top), all steps at once
  // (small_chain).  Whatever part of that does not apply is done the long way: the Horner value lifted (by the last wave
  // earlier, or here) and reduced modulo Delta^(l-1); the steps one by one on W-word integers.
  bool chain_done = false, top_known = false;
  SmallVal top{0, 0, 0, false}, n0v{0, 0, 0, false};
  if (small_on && t.hs_on) chain_done = small_chain(c, z, cand, l, &top_known, top, n0v);
  if (!chain_done) {
    if (top_known) {
      nm = lane == 0 ? top.w0 : (lane == 1 ? top.w1 : (lane == 2 ? top.w2 : 0));
      nneg = top.neg;
    } else {
      if (*h_done == 0) lift_horner();
      hneg = Tl[(size_t)(l - 1) * 64 + 63] != 0;
      u64 x = lane < 63 ? Tl[(size_t)(l - 1) * 64 + lane] : 0;
      // reduce_modulo_poly (:154-178): noise_{l-1} = (nm, nneg)
      wave_divmod2(c, x, smallL, dpw, dn_dp, qq, r);
      nneg = hneg;
      if (__ballot(r != 0) == 0) nneg = false;
      if (wave_cmp(r, hdw) > 0) {
        r = wave_sub(dpw, r, lane);
        nneg = !nneg;
      }
      nm = r;
      if (small_on && t.sc_on && l >= 2 && __ballot(lane >= 3 && nm != 0) == 0) {
        top = SmallVal{readlane_u64(nm, 0), readlane_u64(nm, 1), readlane_u64(nm, 2), nneg};
        top_known = true;
        chain_done = small_chain(c, z, cand, l, &top_known, top, n0v);
      }
    }
  }
  if (chain_done) {
    nm = lane == 0 ? n0v.w0 : 0;
    nneg = n0v.neg;
  }
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
  for (u32 i = l - 1; !chain_done && i-- > 0;) {
    bool pneg;
    const u64 ta = dbg >= 4 && dbg <= 6 ? clock64() : 0;
    u64 p = sub_centre(nm, nneg, Tl[(size_t)i * 64 + lane], pneg);
    const u64 tb = dbg >= 4 && dbg <= 6 ? clock64() : 0;
    u64 hi = p >> 63, lo2 = p << 1;                     // 2|p| + Delta
    u64 sm = lo2 + dlw;
    u64 num = wave_normalize(sm, hi + (sm < dlw), 0, lane);
    const u64 tc = dbg >= 4 && dbg <= 6 ? clock64() : 0;
    wave_divmod2(c, num, smallL + (2 * W + 2), tdw, dn_td, qq, r);
    if (dbg >= 4 && dbg <= 6 && i == l - 3) {                       // timing experiment: one step of the chain in three parts
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
    if (lane == 0 && dbg < 8) out[d] = dbg == 1 ? (tk1 - tk0) : (dbg == 2 ? (tk2 - tk1) : (dbg == 7 ? (tk1 - tk0b) : (tk3 - tk2)));
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
