// pvw_mac.hip -- the streamed inner products of encrypt on gfx950 (MI355X / CDNA4): mac_rows over the tiled matrix,
// mac_rows_packed over its bit-packed copy (what single-dealer encrypt runs), mac_rows_multi (<= 4 vectors per pass)
// and, in the measurement build, the read-bandwidth probes.
//
// Data layout in HBM
//   "tiled matrix" M (A-hat rows followed by B-hat rows) -- the streamed operand of
//   encrypt (c1 = A r + e1, c2 = B r + e2 + m g; src/crypto/encryption.rs:158,177-200,
//   src/params/crs.rs:188-201):
//        M[row_block][limb][j][rho][slot]        rho < R = 128/l, slot < l
//   One (row_block, limb, j) tile is 128 u64 = 1 KiB = exactly one 16-byte-per-lane
//   wave64 load; a (row_block, limb) pair is k contiguous tiles.  A lane owns the same
//   (row, slot pair) for every j, so the k-term inner product needs no cross-lane step.
//   r-hat is stored [limb][j][slot] so the slice a wave needs is contiguous.
//   Everything that crosses the C ABI uses the reference's layout, [..][limb][slot]
//   (src/params/parameters.rs:433-458).
//
// Roofline: HBM-bound integer work (no dense contraction, no MFMA): 8 B read per modular MAC
// (1 MAC = 4 v_mad_u64_u32 + 4 v_addc), W/8 B when the W-bit packed copy is streamed.
#include <hip/hip_runtime.h>

#include "pvw_arith.h"
#include "pvw_chacha.h"
#include "pvw_dev.h"
#include "pvw_kernels.h"

namespace pvw {

#if PVW_TUNING
// per-workgroup time stamps of one stamped launch (tuning build, PVW_MAC_VARIANT 40 / 44): [2b] = first instruction,
// [2b+1] = last store issued, in ticks of the constant 100 MHz counter (s_memrealtime); hw[b] = XCC_ID << 28 | HW_ID
#define PVW_STAMP_MAX 65536
__device__ u64 g_stamp_buf[2 * PVW_STAMP_MAX];
__device__ u32 g_stamp_hw[PVW_STAMP_MAX];
#endif
template <bool STAMP>
__device__ __forceinline__ void stamp_begin(u32 item) {
#if PVW_TUNING
  if constexpr (STAMP) {
    if (threadIdx.x == 0 && item < PVW_STAMP_MAX) {
      g_stamp_buf[2 * item] = __builtin_amdgcn_s_memrealtime();
      // XCC_ID (hardware register 20, low 4 bits) in bits 28..31 of the word, the CU / SE fields of HW_ID below it
      // ... bits 24..26: the block id modulo 8 (which dispatcher share the block came from)
      g_stamp_hw[item] = (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 28) | ((blockIdx.x & 7u) << 24) |
                         (__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) & 0x00ffffffu);
    }
  }
#endif
}
template <bool STAMP>
__device__ __forceinline__ void stamp_end(u32 item) {
#if PVW_TUNING
  if constexpr (STAMP) {
    if (threadIdx.x == 0 && item < PVW_STAMP_MAX) g_stamp_buf[2 * item + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// what a MAC workgroup needs to know about its output rows; shared by the three streaming kernels
struct MacItem {
  const u64* M;
  const u64* addend;
  u64* out;
  const i64* e_small;      // compact addends (MacSection): small coefficients per row ...
  const u64* scalars;      // ... and the row's scalar
  u32 nrows, rb, limb;
  __device__ __forceinline__ bool has_addend() const { return addend != nullptr || e_small != nullptr; }
};
__device__ __forceinline__ MacItem mac_item(const MacSection& sa, const MacSection& sb, u32 item, u32 L) {
  // section a = A-hat rows (c1), section b = B-hat rows (c2): one launch covers both
  MacItem it;
  it.limb = item % L;
  const u32 rbg = item / L;
  const bool in_a = rbg < sa.row_blocks;
  it.rb = in_a ? rbg : rbg - sa.row_blocks;
  it.M = in_a ? sa.M : sb.M;
  it.addend = in_a ? sa.addend : sb.addend;
  it.out = in_a ? sa.out : sb.out;
  it.e_small = in_a ? sa.e_small : sb.e_small;
  it.scalars = in_a ? sa.scalars : sb.scalars;
  it.nrows = in_a ? sa.nrows : sb.nrows;
  return it;
}
// Wave 0 of a MAC workgroup makes its rows' addends from the compact form (MacSection::e_small / scalars): lane i < R
// fetches row i's l small coefficients and scalar at the top of the kernel (mac_small_fetch: the loads ride with the first
// tile loads) and, AFTER its share of the inner products, reduces them mod this workgroup's modulus, transforms them
// (l-point NTT in registers) and adds m_i g-hat (encryption.rs:161-167, :195-196; encode_scalar parameters.rs:346-367) --
// ~250 instructions on R lanes -- then the wave redistributes through `adl` (R * ELL words of LDS) and every lane gets its
// (row rho, slot pair sp) pair.  Bit-identical to what the prologue launch would have written as the addend.  (Doing it at
// the top instead, before wave 0's first MAC, cost 2.5 us per round of workgroups: the coefficients arrive no earlier than
// the tiles, and the transform then stands between the tiles and their MACs.)
template <int ELL>
struct SmallAddend {
  v2u64 e[ELL / 2];
  u64 m;
};
template <int ELL>
__device__ __forceinline__ void mac_small_fetch(const MacItem& it, u32 lane, SmallAddend<ELL>& sa) {
  constexpr int R = 128 / ELL;
  const u32 row = it.rb * R + lane;
  sa.m = 0;
#pragma unroll
  for (int s = 0; s < ELL / 2; ++s) sa.e[s] = (v2u64){0, 0};
  if (lane < (u32)R && row < it.nrows) {
    const v2u64* src = reinterpret_cast<const v2u64*>(it.e_small + (size_t)row * ELL);
#pragma unroll
    for (int s = 0; s < ELL / 2; ++s) sa.e[s] = src[s];
    if (it.scalars) sa.m = it.scalars[row];
  }
}
template <int ELL>
__device__ __forceinline__ v2u64 mac_small_make(const MacItem& it, const DevTables& t, const Mod& m, const SmallAddend<ELL>& sa, u32 lane,
                                                u32 rho, u32 sp, u64* adl) {
  constexpr int R = 128 / ELL;
  if (lane < (u32)R) {
    u64 a[ELL];
#pragma unroll
    for (int s = 0; s < ELL / 2; ++s) {
      a[2 * s] = signed_residue((i64)sa.e[s].x, m);
      a[2 * s + 1] = signed_residue((i64)sa.e[s].y, m);
    }
    ntt_forward<ELL>(a, t.tw + (size_t)it.limb * ELL, t.twp + (size_t)it.limb * ELL, m);
    if (it.scalars) {
      const u64 mr = signed_residue((i64)sa.m, m);                       // `as i64` wrap, encryption.rs:195
      const u64* g = t.ghat + (size_t)it.limb * ELL;
      const u64* gp = t.ghatp + (size_t)it.limb * ELL;
#pragma unroll
      for (int s = 0; s < ELL; ++s) a[s] = addmod(a[s], mulmod_shoup(mr, g[s], gp[s], m.q), m.q);
    }
#pragma unroll
    for (int s = 0; s < ELL; s += 2) *reinterpret_cast<v2u64*>(adl + lane * ELL + s) = (v2u64){a[s], a[s + 1]};
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return *reinterpret_cast<const v2u64*>(adl + rho * ELL + 2 * sp);      // (rows past the section's end are never stored)
}
// cross-wave sum of the four wave partials, addend, store (wave 0); `lds` holds at least 256 v2u64
__device__ __forceinline__ void mac_epilogue(v2u64* lds, const v2u64& part, const Mod& m, const MacItem& it, u32 wave, u32 lane,
                                             u32 out_row, size_t out_o, v2u64 add_pf) {
  __syncthreads();  // all waves are done with their r-hat slices
  lds[wave * 64 + lane] = part;
  __syncthreads();
  if (wave == 0) {
    if (out_row < it.nrows) {
      v2u64 s = lds[lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const v2u64 tq = lds[w * 64 + lane];
        s.x = addmod(s.x, tq.x, m.q);
        s.y = addmod(s.y, tq.y, m.q);
      }
      if (it.has_addend()) {
        s.x = addmod(s.x, add_pf.x, m.q);
        s.y = addmod(s.y, add_pf.y, m.q);
      }
      reinterpret_cast<v2u64*>(it.out)[out_o] = s;
    }
  }
}

// ------------------------------------------------------------------------------------
// mac_rows: out[row][limb][slot] = sum_j M[row][j][limb][slot] * rhat[j][limb][slot] + addend
// grid = row_blocks * L workgroups of 256 threads; the 4 waves split the j range, each streaming
// 1-KiB tiles with non-temporal loads, U in flight + U prefetched per wave; r-hat slices are staged in wave-private
// LDS.  ILV (k % 4U == 0): the waves interleave groups of U tiles, so the workgroup reads ONE contiguous stream
// (profiles/r01_variant_sweep.txt, r01d_mac_ilv_sweep.txt hold the sweeps that chose these schedules).
// ------------------------------------------------------------------------------------
template <int ELL, int U, bool ILV = false, bool STAMP = false>
__global__ __launch_bounds__(256) void mac_rows_kernel(MacSection sa, MacSection sb, const u64* __restrict__ rhat, DevTables t, u32 k, u32 L) {
  constexpr int NW = 4;
  constexpr int HALF = ELL / 2;   // 16-byte slot pairs per polynomial limb
  constexpr int R = 128 / ELL;    // rows per tile
  constexpr int JC = ELL <= 16 ? 64 : (ELL == 32 ? 32 : 16);  // j per staged r-hat chunk (LDS <= 32 KiB)
  __shared__ v2u64 lds[NW * JC * HALF];
  __shared__ u64 adl[ELL <= 16 ? 128 : 1];                    // compact addends of the workgroup's rows (mac_small_addend)
  static_assert(JC * HALF >= 64, "the wave partials reuse the r-hat slabs");
  const u32 item = blockIdx.x;
  stamp_begin<STAMP>(item);
  const MacItem it = mac_item(sa, sb, item, L);
  const u32 limb = it.limb;
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const u32 sp = lane % HALF, rho = lane / HALF;
  // ILV: local tile t of a wave is global tile (t / U) * NW*U + wave * U + t % U
  const u32 kq = ILV ? k / NW : (k + NW - 1) / NW;
  const u32 j0 = ILV ? 0 : (wave * kq < k ? wave * kq : k);
  const u32 j1 = ILV ? kq : ((j0 + kq) < k ? (j0 + kq) : k);
  auto gmap = [&](u32 tt) -> u32 { return ILV ? (tt / U) * NW * U + wave * U + tt % U : tt; };

  const v2u64* Mp = reinterpret_cast<const v2u64*>(it.M + ((size_t)it.rb * L + limb) * (size_t)k * 128) + lane;
  const v2u64* rp = reinterpret_cast<const v2u64*>(rhat + (size_t)limb * k * ELL);
  v2u64* lw = lds + wave * (JC * HALF);

  const u32 out_row = it.rb * R + rho;
  const size_t out_o = (((size_t)out_row * L + limb) * ELL) / 2 + sp;
  v2u64 add_pf = (v2u64){0, 0};
  SmallAddend<(ELL <= 16 ? ELL : 2)> small;

  Acc a0, a1;
  acc_zero(a0);
  acc_zero(a1);
  for (u32 jc = j0; jc < j1; jc += JC) {
    const u32 cnt = (j1 - jc) < (u32)JC ? (j1 - jc) : (u32)JC;
    auto ld = [&](u32 tile) -> v2u64 { return __builtin_nontemporal_load(Mp + (size_t)gmap(jc + tile) * 64); };
    __builtin_amdgcn_wave_barrier();
    // the chunk's first U matrix tiles are requested first; the r-hat elements of this lane behind them (all of them
    // before the first is awaited: a load per loop trip would pay one L2 round trip each)
    v2u64 x[U];
    const bool full = cnt >= U;
    if (full) {
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = ld(u);
    }
    // the addend of this lane's output is made (compact form, l <= 16) or requested (e1 / e2 + m*g written by the prologue) now,
    // by the wave that will write the result: at the end it would cost the workgroup one more exposed memory latency
    if (jc == j0 && wave == 0) {
      bool fetched = false;
      if constexpr (ELL <= 16) {
        if (it.e_small) { mac_small_fetch<ELL>(it, lane, small); fetched = true; }
      }
      if (!fetched && it.addend && out_row < it.nrows) add_pf = reinterpret_cast<const v2u64*>(it.addend)[out_o];
    }
    constexpr int RN = JC * HALF / 64;
    v2u64 rv[RN];
#pragma unroll
    for (int xx = 0; xx < RN; ++xx) {
      const u32 idx = lane + 64 * xx;
      const u32 ic = idx < cnt * HALF ? idx : 0;
      rv[xx] = rp[(size_t)gmap(jc + ic / HALF) * HALF + ic % HALF];
    }
#pragma unroll
    for (int xx = 0; xx < RN; ++xx) {
      const u32 idx = lane + 64 * xx;
      if (idx < cnt * HALF) lw[idx] = rv[xx];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    u32 jj = 0;
    if (full) {
      // two register buffers: the next U tiles are in flight while the current U are consumed
      v2u64 xn[U];
      for (; jj + 2 * U <= cnt; jj += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) xn[u] = ld(jj + U + u);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const v2u64 y = lw[(jj + u) * HALF + sp];
          acc_mac_dev(a0, x[u].x, y.x);
          acc_mac_dev(a1, x[u].y, y.y);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = xn[u];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const v2u64 y = lw[(jj + u) * HALF + sp];
        acc_mac_dev(a0, x[u].x, y.x);
        acc_mac_dev(a1, x[u].y, y.y);
      }
      jj += U;
    }
    for (; jj < cnt; ++jj) {
      const v2u64 xv = ld(jj);
      const v2u64 y = lw[jj * HALF + sp];
      acc_mac_dev(a0, xv.x, y.x);
      acc_mac_dev(a1, xv.y, y.y);
    }
  }
  // one Barrett reduction per wave partial ("wavefront-wide": q, ratio are SGPRs)
  const Mod m = t.mods[limb];
  if constexpr (ELL <= 16) {
    if (wave == 0 && it.e_small) add_pf = mac_small_make<ELL>(it, t, m, small, lane, rho, sp, adl);
  }
  v2u64 part;
  part.x = acc_reduce(a0, m);
  part.y = acc_reduce(a1, m);
  mac_epilogue(lds, part, m, it, wave, lane, out_row, out_o, add_pf);
  stamp_end<STAMP>(item);
}

// ------------------------------------------------------------------------------------
// mac_rows over a PACKED copy of the tiled matrix.  mac_rows is bound by the bytes it streams, and a residue of a
// W-bit modulus carries W bits in an 8-byte word: the packed copy stores, for every (row block, limb, lane), the
// lane's residue pairs (x_j, y_j), j = 0..k-1, as ONE bit stream of 2W bits per j, cut into 16-byte chunks;
// chunk c of the 64 lanes is 1 KiB contiguous, so the loads are exactly mac_rows' (global_load_dwordx4 nt, 1 KiB
// per wave-instruction) -- there are just W of them per 64 j instead of 64.  Unpacking is two funnel shifts and a
// mask per residue on a VALU that the quarter-rate v_mad_u64_u32 stream leaves half idle.  Same lazy accumulation,
// same epilogue, same results as mac_rows_kernel.  Built by the C ABI (pack_kernel) next to the tiled matrix, which
// every other consumer keeps using.  Widths: the modulus chain's widest modulus rounded up to 40 / 48 / 56 / 61 --
// the reference's own parameter sets are 36/37-bit (tests/crypto.rs:52) and 56-bit (examples/pvw_valid_dec.rs:40-45)
// chains; the bench chain of SURVEY 8d is 61-bit.
//
// W = 61 (mac_rows_packed61_kernel): 64 j are 61 chunks exactly, so with k a multiple of 256 every wave owns whole
// periods (j in [w k/4, (w+1) k/4)) and every shift amount is a compile-time constant: the period is unrolled as 4
// groups of 16 j, each living in 16 chunks (the last chunk of a group is the first of the next and is carried in
// registers, not loaded again), 15-16 chunks in flight per wave while the previous group is multiplied.
// W = 40 / 48 / 56 (mac_rows_packedw_kernel): a group of 16 j is W/4 chunks EXACTLY, so waves own whole groups and
// any k that is a multiple of 64 qualifies; two windows of W/4 chunks in turn.
// ------------------------------------------------------------------------------------
#ifndef PVW_PACKED_WPC
#define PVW_PACKED_WPC 2                                  // workgroups per CU the register allocation aims at
#endif
template <int N>
__device__ __forceinline__ u64 pk_word(const v2u64 (&a)[N], int idx) { return (idx & 1) ? a[idx >> 1].y : a[idx >> 1].x; }
// the W bits at bit offset `bit` of the window a[] (bit is a constant after unrolling)
template <int W, int N>
__device__ __forceinline__ u64 pk_get(const v2u64 (&a)[N], int bit) {
  const int idx = bit >> 6, sh = bit & 63;
  u64 v = pk_word(a, idx) >> sh;
  if (sh + W > 64) v |= pk_word(a, idx + 1) << (64 - sh);
  return v & ((1ull << W) - 1);
}
template <int ELL, bool STAMP = false>
__global__ __launch_bounds__(256, PVW_PACKED_WPC) void mac_rows_packed61_kernel(MacSection sa, MacSection sb, const u64* __restrict__ rhat,
                                                                               DevTables t, u32 k, u32 L) {
  constexpr int HALF = ELL / 2, R = 128 / ELL, JC = 64, NW = 4, W = 61;
  static_assert(ELL <= 16, "one period of 64 j per r-hat slab");
  __shared__ v2u64 lds[NW * JC * HALF];
  __shared__ u64 adl[128];                                    // compact addends of the workgroup's rows (mac_small_addend)
  const u32 item = blockIdx.x;
  stamp_begin<STAMP>(item);
  const MacItem it = mac_item(sa, sb, item, L);
  const u32 limb = it.limb;
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const u32 sp = lane % HALF, rho = lane / HALF;
  const u32 kq = k / NW, periods = kq / 64;                   // the launcher guarantees k % 256 == 0
  const u32 chunks = k / 64 * W;                              // per (row block, limb)
  const v2u64* Pp = reinterpret_cast<const v2u64*>(it.M) + (((size_t)it.rb * L + limb) * chunks + (size_t)wave * periods * W) * 64 + lane;
  const v2u64* rp = reinterpret_cast<const v2u64*>(rhat + (size_t)limb * k * ELL) + (size_t)wave * kq * HALF;
  v2u64* lw = lds + wave * (JC * HALF);
  const u32 out_row = it.rb * R + rho;
  const size_t out_o = (((size_t)out_row * L + limb) * ELL) / 2 + sp;
  v2u64 add_pf = (v2u64){0, 0};
  Acc a0, a1;
  acc_zero(a0);
  acc_zero(a1);
  auto ldc = [&](u32 c) -> v2u64 { return __builtin_nontemporal_load(Pp + (size_t)c * 64); };
  v2u64 xa[16], xb[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) xa[u] = ldc(u);                // group 0 of the first period
  // the addend of this lane's output is made (compact form) or requested (e1 / e2 + m*g written by the prologue) now, by the
  // wave that will write the result: at the end it would cost the workgroup one more exposed memory latency
  SmallAddend<ELL> small;
  if (wave == 0) {
    if (it.e_small) mac_small_fetch<ELL>(it, lane, small);
    else if (it.addend && out_row < it.nrows) add_pf = reinterpret_cast<const v2u64*>(it.addend)[out_o];
  }
  for (u32 pd = 0; pd < periods; ++pd) {
    const u32 cb = pd * W;
    // this period's r-hat slab: 64 j x HALF sixteen-byte elements, HALF per lane
    __builtin_amdgcn_wave_barrier();
    constexpr int RN = JC * HALF / 64;
    v2u64 rv[RN];
#pragma unroll
    for (int x = 0; x < RN; ++x) rv[x] = rp[(size_t)pd * 64 * HALF + lane + 64 * x];
#pragma unroll
    for (int x = 0; x < RN; ++x) lw[lane + 64 * x] = rv[x];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // group g: j = 16 g .. 16 g + 15 of the period, bits 32 g + 122 jj of the window cur[] = chunks 15 g .. 15 g + 15;
    // nxt[1..15] = chunks 15 g + 16 .. 15 g + 30 are requested first, nxt[0] is cur[15]
    auto group = [&](const int g, v2u64 (&cur)[16], v2u64 (&nxt)[16]) {
      if (g < 3) {
#pragma unroll
        for (int u = 1; u < 16; ++u) nxt[u] = ldc(cb + 15 * (g + 1) + u);
      } else if (pd + 1 < periods) {
#pragma unroll
        for (int u = 0; u < 16; ++u) nxt[u] = ldc(cb + W + u);   // group 0 of the next period
      }
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int bit = 32 * g + 2 * W * jj;
        const u64 xv = pk_get<W>(cur, bit), yv = pk_get<W>(cur, bit + W);
        const v2u64 r = lw[(16 * g + jj) * HALF + sp];
        acc_mac_dev(a0, xv, r.x);
        acc_mac_dev(a1, yv, r.y);
      }
      if (g < 3) nxt[0] = cur[15];
    };
    group(0, xa, xb);
    group(1, xb, xa);
    group(2, xa, xb);
    group(3, xb, xa);                                          // leaves the next period's group 0 in xa
  }
  const Mod m = t.mods[limb];
  if (wave == 0 && it.e_small) add_pf = mac_small_make<ELL>(it, t, m, small, lane, rho, sp, adl);
  v2u64 part;
  part.x = acc_reduce(a0, m);
  part.y = acc_reduce(a1, m);
  mac_epilogue(lds, part, m, it, wave, lane, out_row, out_o, add_pf);
  stamp_end<STAMP>(item);
}

template <int ELL, int W, bool STAMP = false>
__global__ __launch_bounds__(256, PVW_PACKED_WPC) void mac_rows_packedw_kernel(MacSection sa, MacSection sb, const u64* __restrict__ rhat,
                                                                              DevTables t, u32 k, u32 L) {
  constexpr int HALF = ELL / 2, R = 128 / ELL, JC = 64, NW = 4, CG = W / 4;   // CG chunks per group of 16 j
  static_assert(ELL <= 16 && W % 4 == 0 && W < 64, "whole chunks per group");
  __shared__ v2u64 lds[NW * JC * HALF];
  __shared__ u64 adl[128];                                    // compact addends of the workgroup's rows (mac_small_addend)
  const u32 item = blockIdx.x;
  stamp_begin<STAMP>(item);
  const MacItem it = mac_item(sa, sb, item, L);
  const u32 limb = it.limb;
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const u32 sp = lane % HALF, rho = lane / HALF;
  const u32 kq = k / NW, gw = kq / 16;                        // the launcher guarantees k % 64 == 0: gw groups per wave
  const u32 chunks = k / 16 * CG;                             // per (row block, limb)
  const v2u64* Pp = reinterpret_cast<const v2u64*>(it.M) + (((size_t)it.rb * L + limb) * chunks + (size_t)wave * gw * CG) * 64 + lane;
  const v2u64* rp = reinterpret_cast<const v2u64*>(rhat + (size_t)limb * k * ELL) + (size_t)wave * kq * HALF;
  v2u64* lw = lds + wave * (JC * HALF);
  const u32 out_row = it.rb * R + rho;
  const size_t out_o = (((size_t)out_row * L + limb) * ELL) / 2 + sp;
  v2u64 add_pf = (v2u64){0, 0};
  Acc a0, a1;
  acc_zero(a0);
  acc_zero(a1);
  auto ldc = [&](u32 c) -> v2u64 { return __builtin_nontemporal_load(Pp + (size_t)c * 64); };
  v2u64 xa[CG], xb[CG];
#pragma unroll
  for (int u = 0; u < CG; ++u) xa[u] = ldc(u);                // group 0
  // the addend of this lane's output is made (compact form) or requested (e1 / e2 + m*g written by the prologue) now, by the
  // wave that will write the result: at the end it would cost the workgroup one more exposed memory latency
  SmallAddend<ELL> small;
  if (wave == 0) {
    if (it.e_small) mac_small_fetch<ELL>(it, lane, small);
    else if (it.addend && out_row < it.nrows) add_pf = reinterpret_cast<const v2u64*>(it.addend)[out_o];
  }
  // group g of this wave: j = 16 g .. 16 g + 15 of its range = chunks CG g .. CG g + CG - 1, residue i of the group at
  // bit W i.  Every fourth group starts a slab of (up to) 64 j of r-hat: its loads go out first, the next group's
  // chunks behind them, and only then are the slab's elements awaited and written to LDS.
  auto group = [&](const u32 g, v2u64 (&cur)[CG], v2u64 (&nxt)[CG]) {
    constexpr int RN = JC * HALF / 64;
    const bool slab = (g & 3) == 0;
    v2u64 rv[RN];
    if (slab) {
      const u32 cnt = (gw - g) < 4 ? (gw - g) * 16 : 64;       // j in this slab
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int x = 0; x < RN; ++x) {
        const u32 idx = lane + 64 * x;
        rv[x] = rp[(size_t)g * 16 * HALF + (idx < cnt * HALF ? idx : 0)];
      }
    }
    if (g + 1 < gw) {
#pragma unroll
      for (int u = 0; u < CG; ++u) nxt[u] = ldc((g + 1) * CG + u);
    }
    if (slab) {
#pragma unroll
      for (int x = 0; x < RN; ++x) lw[lane + 64 * x] = rv[x];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    const v2u64* lg = lw + (g & 3) * 16 * HALF + sp;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const u64 xv = pk_get<W>(cur, 2 * W * jj), yv = pk_get<W>(cur, 2 * W * jj + W);
      const v2u64 r = lg[jj * HALF];
      acc_mac_dev(a0, xv, r.x);
      acc_mac_dev(a1, yv, r.y);
    }
  };
  for (u32 g = 0; g < gw; g += 2) {
    group(g, xa, xb);
    if (g + 1 < gw) group(g + 1, xb, xa);
  }
  const Mod m = t.mods[limb];
  if (wave == 0 && it.e_small) add_pf = mac_small_make<ELL>(it, t, m, small, lane, rho, sp, adl);
  v2u64 part;
  part.x = acc_reduce(a0, m);
  part.y = acc_reduce(a1, m);
  mac_epilogue(lds, part, m, it, wave, lane, out_row, out_o, add_pf);
  stamp_end<STAMP>(item);
}

// tiled matrix -> W-bit packed copy: one thread per (row block, limb, lane) walks its k residue pairs and emits the
// bit stream in 16-byte chunks (reads and writes are both 1 KiB per wave and step; load-time only)
template <int W>
__global__ __launch_bounds__(256) void pack_kernel(const u64* __restrict__ M, u64* __restrict__ P, u32 k, size_t items,
                                                   u32* __restrict__ wide_flag) {
  const size_t tt = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t item = tt >> 6;
  const u32 lane = (u32)(tt & 63);
  if (item >= items) return;
  const v2u64* src = reinterpret_cast<const v2u64*>(M) + item * (size_t)k * 64 + lane;
  v2u64* dst = reinterpret_cast<v2u64*>(P) + item * (size_t)(k / 64 * W) * 64 + lane;
  u64 lo = 0, hi = 0, pend = 0;      // bit buffer (lo, hi), `nb` bits used; pend = the even word of the chunk being filled
  u32 nb = 0, words = 0;
  constexpr u64 MASK = (1ull << W) - 1;
  auto push = [&](u64 v) {
    lo |= v << nb;
    if (nb + W > 64) hi |= v >> (64 - nb);
    nb += W;
    if (nb >= 64) {
      if (words & 1) dst[(size_t)(words >> 1) * 64] = (v2u64){pend, lo};
      else pend = lo;
      ++words;
      lo = hi;
      hi = 0;
      nb -= 64;
    }
  };
  u64 seen = 0;
  for (u32 j = 0; j < k; ++j) {
    const v2u64 v = src[(size_t)j * 64];
    seen |= v.x | v.y;
    push(v.x & MASK);
    push(v.y & MASK);
  }
  // a word that does not fit W bits (a caller loaded unreduced data): the copy must not be used
  if (seen >> W) atomicOr(wide_flag, 1u);
}

// ------------------------------------------------------------------------------------
// mac_rows_multi: NV vectors against one pass over the tiled matrix,
//     out_v[row] = sum_j M[row][j] * vhat_v[j] + addend_v[row],   v < NV.
// Every 16-byte tile element is loaded once and used for 2*NV modular MACs, so the kernel moves
// from the HBM roofline (NV = 1: mac_rows) towards the integer-VALU roofline.  Serves
//   * multi-dealer encrypt (encrypt_all_party_shares, encryption.rs:253-286): vectors = r-hat of
//     NV dealers, matrix = [A-hat; B-hat];
//   * batched key generation (public_key.rs:111-147 over crs.rs:138-171): vectors = s-hat of NV
//     parties, matrix = transposed CRS.
// ------------------------------------------------------------------------------------
template <int ELL, int NV>
__global__ __launch_bounds__(256) void mac_rows_multi_kernel(MacSection sa, MacSection sb, MultiVec mv,
                                                              const Mod* __restrict__ mods, u32 k, u32 L) {
  constexpr int HALF = ELL / 2;
  constexpr int R = 128 / ELL;
  constexpr int JC = ELL <= 8 ? 16 : (ELL == 16 ? 8 : (ELL == 32 ? 8 : 4));   // LDS = 4*NV*JC*HALF*16 B <= 32 KiB
  constexpr int U = JC < 8 ? JC : 8;
  __shared__ v2u64 lds[4 * NV * JC * HALF];

  const u32 limb = blockIdx.x % L;
  const u32 rbg = blockIdx.x / L;
  const bool in_a = rbg < sa.row_blocks;
  const u32 rb = in_a ? rbg : rbg - sa.row_blocks;
  const u64* __restrict__ M = in_a ? sa.M : sb.M;
  const u64* addend = in_a ? sa.addend : sb.addend;
  u64* out = in_a ? sa.out : sb.out;
  const u32 nrows = in_a ? sa.nrows : sb.nrows;
  const size_t ostride = in_a ? mv.out_stride_a : mv.out_stride_b;

  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const u32 sp = lane % HALF, rho = lane / HALF;
  const u32 kq = (k + 3) / 4;
  const u32 j0 = wave * kq < k ? wave * kq : k;
  const u32 j1 = (j0 + kq) < k ? (j0 + kq) : k;

  const v2u64* Mp = reinterpret_cast<const v2u64*>(M + ((size_t)rb * L + limb) * (size_t)k * 128) + lane;
  v2u64* lw = lds + wave * (NV * JC * HALF);

  Acc a0[NV], a1[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) { acc_zero(a0[v]); acc_zero(a1[v]); }

  for (u32 jc = j0; jc < j1; jc += JC) {
    const u32 cnt = (j1 - jc) < (u32)JC ? (j1 - jc) : (u32)JC;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const u32 vv = (u32)v < mv.nv ? (u32)v : mv.nv - 1;
      const v2u64* rp = reinterpret_cast<const v2u64*>(mv.vhat + (size_t)vv * mv.vstride + (size_t)limb * k * ELL);
      for (u32 idx = lane; idx < cnt * HALF; idx += 64) lw[v * (JC * HALF) + idx] = rp[(size_t)jc * HALF + idx];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const v2u64* mp = Mp + (size_t)jc * 64;
    u32 jj = 0;
    for (; jj + U <= cnt; jj += U) {
      v2u64 x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(mp + (size_t)(jj + u) * 64);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          v2u64 y = lw[v * (JC * HALF) + (jj + u) * HALF + sp];
          acc_mac_dev(a0[v], x[u].x, y.x);
          acc_mac_dev(a1[v], x[u].y, y.y);
        }
      }
    }
    for (; jj < cnt; ++jj) {
      v2u64 xv = mp[(size_t)jj * 64];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        v2u64 y = lw[v * (JC * HALF) + jj * HALF + sp];
        acc_mac_dev(a0[v], xv.x, y.x);
        acc_mac_dev(a1[v], xv.y, y.y);
      }
    }
  }

  const Mod m = mods[limb];
  const u32 row = rb * R + rho;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    v2u64 part;
    part.x = acc_reduce(a0[v], m);
    part.y = acc_reduce(a1[v], m);
    __syncthreads();
    lds[wave * 64 + lane] = part;
    __syncthreads();
    if (wave == 0 && row < nrows && (u32)v < mv.nv) {
      v2u64 s = lds[lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        v2u64 tq = lds[w * 64 + lane];
        s.x = addmod(s.x, tq.x, m.q);
        s.y = addmod(s.y, tq.y, m.q);
      }
      const size_t o = ((size_t)v * ostride + ((size_t)row * L + limb) * ELL) / 2 + sp;
      if (addend) {
        v2u64 e = reinterpret_cast<const v2u64*>(addend)[o];
        s.x = addmod(s.x, e.x, m.q);
        s.y = addmod(s.y, e.y, m.q);
      }
      reinterpret_cast<v2u64*>(out)[o] = s;
    }
  }
}

#if PVW_TUNING
// ------------------------------------------------------------------------------------
// read-bandwidth probe (measurement aid): the loads of mac_rows -- 1-KiB tiles, 16 bytes per lane,
// non-temporal, 16 in flight per wave, four waves per workgroup on one contiguous run -- with the arithmetic
// replaced by an xor, so that the ceiling the memory system offers this access pattern is measured, not assumed
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void read_probe_kernel(const u64* __restrict__ M, size_t total_tiles, u32 tiles_per_wave,
                                                          u64* __restrict__ sink) {
  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t run0 = (size_t)blockIdx.x * 4 * tiles_per_wave;          // the workgroup's contiguous run
  const v2u64* p = reinterpret_cast<const v2u64*>(M) + lane;
  v2u64 acc = (v2u64){0, 0};
  // waves interleave groups of 16 tiles, as the default mac_rows schedule does
  for (u32 g = 0; g + 16 <= tiles_per_wave; g += 16) {
    v2u64 x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const size_t tile = run0 + (size_t)(g / 16) * 64 + wave * 16 + u;
      x[u] = tile < total_tiles ? __builtin_nontemporal_load(p + tile * 64) : (v2u64){0, 0};
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc ^= x[u];
  }
  if ((acc.x ^ acc.y) == 0x9e3779b97f4a7c15ULL) sink[blockIdx.x] = acc.x;   // keeps the loads alive
}

// the same with the number of tiles in flight per wave (U, and U + U when DBUF) and the workgroups resident per CU
// (through a dynamic LDS allocation that is never read) as parameters: maps delivered bandwidth against bytes in flight
template <int U, bool DBUF>
__global__ __launch_bounds__(256) void read_probe2_kernel(const u64* __restrict__ M, size_t total_tiles, u32 tiles_per_wave,
                                                           u64* __restrict__ sink, u32 xmap) {
  extern __shared__ u64 probe_pad[];
  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  u32 item = blockIdx.x;
  if (xmap) {                                                 // every XCD a contiguous eighth of the runs
    const u32 per = gridDim.x >> 3, tail = gridDim.x & 7;
    if (item < gridDim.x - tail) item = (item & 7) * per + (item >> 3);
  }
  const size_t run0 = (size_t)item * 4 * tiles_per_wave;
  const v2u64* p = reinterpret_cast<const v2u64*>(M) + lane;
  v2u64 acc = (v2u64){0, 0};
  const u32 G = tiles_per_wave / U;
  auto ld = [&](u32 g, v2u64 (&x)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      size_t tile = run0 + (size_t)g * (4 * U) + wave * U + u;
      tile = tile < total_tiles ? tile : total_tiles - 1;
      x[u] = __builtin_nontemporal_load(p + tile * 64);
    }
  };
  if constexpr (DBUF) {
    v2u64 x[U], xn[U];
    if (G) ld(0, x);
    for (u32 g = 0; g < G; ++g) {
      if (g + 1 < G) ld(g + 1, xn);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= x[u];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = xn[u];
    }
  } else {
    for (u32 g = 0; g < G; ++g) {
      v2u64 x[U];
      ld(g, x);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= x[u];
    }
  }
  if ((acc.x ^ acc.y) == 0x9e3779b97f4a7c15ULL) { sink[blockIdx.x] = acc.x; probe_pad[threadIdx.x] = acc.y; }
}
#endif  // PVW_TUNING

// ------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------
// one workgroup per (row block, limb) of the two sections; false: nothing to launch
static bool mac_grid(MacSection& sa, MacSection& sb, u32 L, u32 ell, u32& blocks) {
  const u32 R = 128 / ell;
  sa.row_blocks = (sa.nrows + R - 1) / R;
  sb.row_blocks = (sb.nrows + R - 1) / R;
  blocks = (sa.row_blocks + sb.row_blocks) * L;
  return blocks != 0;
}

// PVW_MAC_VARIANT (tuning build only): 0 (default) by shape | 17 the non-interleaved schedule | 40 default + per-workgroup
// time stamps | 44 (launch_mac_rows_packed) the packed kernel + stamps.  The sweeps that chose the defaults
// (double-buffered non-temporal loads; the four waves interleave groups of 16 tiles when k allows it: +5 % at l = 16,
// k = 512, +2 % at n = 16384) are profiles/r01_variant_sweep.txt and r01d_mac_ilv_sweep.txt; the persistent / work-queue,
// XCD-contiguous, 8- and 16-wave and single-buffer forms they and profiles/r02_mac_rows_timeline.txt closed are gone.
hipError_t launch_mac_rows(const MacSection& a, const MacSection& b, const u64* rhat, const DevTables& t, u32 k, u32 L, u32 ell,
                           hipStream_t s) {
  MacSection sa = a, sb = b;
  u32 blocks;
  if (!mac_grid(sa, sb, L, ell, blocks)) return hipSuccess;
  [[maybe_unused]] const int variant = (int)PVW_ENV_INT("PVW_MAC_VARIANT", 0);
  const dim3 grid(blocks);
  switch (ell) {
    case 8:
    case 16:
#if PVW_TUNING
      if (variant == 40 && k % 64 == 0) {
        if (ell == 8) mac_rows_kernel<8, 16, true, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);
        else mac_rows_kernel<16, 16, true, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);
        break;
      }
      if (variant == 17) {
        if (ell == 8) mac_rows_kernel<8, 8><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);
        else mac_rows_kernel<16, 16><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);
        break;
      }
#endif
      if (k % 64 == 0) {
        if (ell == 8) mac_rows_kernel<8, 16, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);
        else mac_rows_kernel<16, 16, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);
      } else if (ell == 8) {
        mac_rows_kernel<8, 8><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);
      } else {
        mac_rows_kernel<16, 16><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);
      }
      break;
    case 32: mac_rows_kernel<32, 8><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L); break;
    case 64: mac_rows_kernel<64, 8><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

u32 packed_width(u32 max_q_bits, u32 k, u32 ell) {
  if (ell > 16 || max_q_bits == 0) return 0;
  const u32 w = max_q_bits <= 40 ? 40 : (max_q_bits <= 48 ? 48 : (max_q_bits <= 56 ? 56 : (max_q_bits <= 61 ? 61 : 0)));
  if (w == 0) return 0;
  if (w == 61) return k % 256 == 0 ? 61 : 0;
  return k % 64 == 0 ? w : 0;
}

hipError_t launch_mac_rows_packed(const MacSection& a, const MacSection& b, const u64* rhat, const DevTables& t, u32 k, u32 L,
                                  u32 ell, u32 width, hipStream_t s) {
  if (ell > 16 || width == 0 || (width == 61 ? k % 256 != 0 : (k % 64 != 0 || width % 4 != 0))) return hipErrorInvalidValue;
  MacSection sa = a, sb = b;
  u32 blocks;
  if (!mac_grid(sa, sb, L, ell, blocks)) return hipSuccess;
  const dim3 grid(blocks);
#define PVW_PACKEDW(Wv)                                                                                   \
  do {                                                                                                    \
    if (stamp) {                                                                                          \
      if (ell == 8) mac_rows_packedw_kernel<8, Wv, PVW_TUNING != 0><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);   \
      else mac_rows_packedw_kernel<16, Wv, PVW_TUNING != 0><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);           \
    } else if (ell == 8) mac_rows_packedw_kernel<8, Wv><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);   \
    else mac_rows_packedw_kernel<16, Wv><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);                  \
  } while (0)
  const bool stamp = PVW_TUNING && PVW_ENV_INT("PVW_MAC_VARIANT", 0) == 44;   // per-workgroup time stamps (tools/mac_timeline.py c3 44)
  switch (width) {
    case 61:
      if (stamp) {
        if (ell == 8) mac_rows_packed61_kernel<8, PVW_TUNING != 0><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);
        else mac_rows_packed61_kernel<16, PVW_TUNING != 0><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);
      } else if (ell == 8) mac_rows_packed61_kernel<8><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);
      else mac_rows_packed61_kernel<16><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, t, k, L);
      break;
    case 56: PVW_PACKEDW(56); break;
    case 48: PVW_PACKEDW(48); break;
    case 40: PVW_PACKEDW(40); break;
    default: return hipErrorInvalidValue;
  }
#undef PVW_PACKEDW
  return hipGetLastError();
}

hipError_t launch_pack(const u64* M, u64* P, u32 rows, u32 k, u32 L, u32 ell, u32 width, u32* wide_flag, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  if (k % 64 != 0) return hipErrorInvalidValue;
  const u32 R = 128 / ell;
  const size_t items = (size_t)((rows + R - 1) / R) * L;
  const dim3 grid((u32)((items * 64 + 255) / 256)), blk(256);
  switch (width) {
    case 61: pack_kernel<61><<<grid, blk, 0, s>>>(M, P, k, items, wide_flag); break;
    case 56: pack_kernel<56><<<grid, blk, 0, s>>>(M, P, k, items, wide_flag); break;
    case 48: pack_kernel<48><<<grid, blk, 0, s>>>(M, P, k, items, wide_flag); break;
    case 40: pack_kernel<40><<<grid, blk, 0, s>>>(M, P, k, items, wide_flag); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_mac_rows_multi(const MacSection& a, const MacSection& b, const MultiVec& mv,
                                 const DevTables& t, u32 k, u32 L, u32 ell, hipStream_t s) {
  const u32 R = 128 / ell;
  MacSection sa = a, sb = b;
  sa.row_blocks = (sa.nrows + R - 1) / R;
  sb.row_blocks = (sb.nrows + R - 1) / R;
  const u32 blocks = (sa.row_blocks + sb.row_blocks) * L;
  if (blocks == 0 || mv.nv == 0) return hipSuccess;
  if (mv.nv > 4) return hipErrorInvalidValue;
  if (mv.nv <= 2) {
    PVW_DISPATCH_ELL(ell, mac_rows_multi_kernel<E, 2><<<dim3(blocks), dim3(256), 0, s>>>(sa, sb, mv, t.mods, k, L));
  } else {
    PVW_DISPATCH_ELL(ell, mac_rows_multi_kernel<E, 4><<<dim3(blocks), dim3(256), 0, s>>>(sa, sb, mv, t.mods, k, L));
  }
  return hipGetLastError();
}

#if PVW_TUNING
hipError_t read_stamps(u64* out, u32* hw, u32 count) {
  if (count > PVW_STAMP_MAX) return hipErrorInvalidValue;
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp_buf), (size_t)count * 16, 0, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return e;
  return hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_stamp_hw), (size_t)count * 4, 0, hipMemcpyDeviceToHost);
}
hipError_t launch_read_probe(const u64* M, size_t total_tiles, u32 tiles_per_wave, u64* sink, hipStream_t s) {
  if (total_tiles == 0 || tiles_per_wave < 16) return hipErrorInvalidValue;
  const size_t per_wg = (size_t)4 * tiles_per_wave;
  const u32 blocks = (u32)((total_tiles + per_wg - 1) / per_wg);
  read_probe_kernel<<<dim3(blocks), dim3(256), 0, s>>>(M, total_tiles, tiles_per_wave, sink);
  return hipGetLastError();
}
hipError_t launch_read_probe2(const u64* M, size_t total_tiles, u32 tiles_per_wave, u64* sink, u32 U, bool dbuf, u32 lds_bytes,
                              hipStream_t s, u32 xmap) {
  if (total_tiles == 0 || tiles_per_wave < U || lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  const size_t per_wg = (size_t)4 * tiles_per_wave;
  const u32 blocks = (u32)((total_tiles + per_wg - 1) / per_wg);
#define PVW_PROBE2(Uv, Dv) read_probe2_kernel<Uv, Dv><<<dim3(blocks), dim3(256), lds_bytes, s>>>(M, total_tiles, tiles_per_wave, sink, xmap)
  if (U == 8) { if (dbuf) PVW_PROBE2(8, true); else PVW_PROBE2(8, false); }
  else if (U == 16) { if (dbuf) PVW_PROBE2(16, true); else PVW_PROBE2(16, false); }
  else if (U == 32) { if (dbuf) return hipErrorInvalidValue; else PVW_PROBE2(32, false); }
  else if (U == 4) { if (dbuf) PVW_PROBE2(4, true); else PVW_PROBE2(4, false); }
  else return hipErrorInvalidValue;
#undef PVW_PROBE2
  return hipGetLastError();
}
hipError_t init_probe_attributes() {
  const int big = 160 * 1024;
  hipFuncSetAttribute((const void*)read_probe2_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)read_probe2_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)read_probe2_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)read_probe2_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)read_probe2_kernel<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)read_probe2_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  return hipFuncSetAttribute((const void*)read_probe2_kernel<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
}
#endif  // PVW_TUNING

}  // namespace pvw
