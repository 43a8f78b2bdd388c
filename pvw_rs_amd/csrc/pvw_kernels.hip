// pvw_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels of the PVW hot path.
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
// Roofline: every kernel here is HBM-bound integer work (no dense contraction, no MFMA):
//   mac_rows      8 B read per modular MAC (1 MAC = 4 v_mad_u64_u32 + 4 v_addc)
//   decrypt_mac   same, on the ciphertext layout as it arrives
//   the rest      O(n + k) polynomials, launch-latency sized
#include <hip/hip_runtime.h>

#include "pvw_arith.h"
#include "pvw_chacha.h"
#include "pvw_decode.h"
#include "pvw_kernels.h"

namespace pvw {

typedef u64 v2u64 __attribute__((ext_vector_type(2)));  // one 16-byte lane access

// ------------------------------------------------------------------------------------
// mac_rows: out[row][limb][slot] = sum_j M[row][j][limb][slot] * rhat[j][limb][slot] + addend
// grid = row_blocks * L workgroups of 256 threads; the 4 waves split the j range, each
// streaming a contiguous run of 1-KiB tiles; r-hat slices are staged in wave-private LDS.
// ------------------------------------------------------------------------------------
#if PVW_TUNING
// per-workgroup time stamps of one stamped launch (tuning build, PVW_MAC_VARIANT 40 / 41): [2b] = first instruction,
// [2b+1] = last store issued, in ticks of the constant 100 MHz counter (s_memrealtime); hw[b] = HW_ID of wave 0
#define PVW_STAMP_MAX 65536
__device__ u64 g_stamp_buf[2 * PVW_STAMP_MAX];
__device__ u32 g_stamp_hw[PVW_STAMP_MAX];
__device__ u64 g_stamp_wg[2 * 4096];       // persistent form: [2b] = kernel entry of workgroup b, [2b+1] = its HW_ID
#endif
// WPE: minimum waves per SIMD the register allocation must leave room for (1 = unconstrained).  ARITH (tuning build,
// timing experiments only, wrong results): 2 = the modular MAC, 1 = one of the two MACs per 16 bytes, 0 = an xor.
// STAMP (tuning build): record start / end times of every workgroup.
// XMAP: which item a block takes.  0: item = block id (the hardware deals blocks round-robin over the XCDs, so an XCD
// walks the matrix with a stride of 8 items = 2 MiB at k = 256).  1: every XCD gets a contiguous eighth of the items.
// 2: eighths interleaved in pairs of items.
template <int ELL, int U, bool NT, bool DBUF, bool ILV = false, int NW = 4, int WPE = 1, int ARITH = 2, bool STAMP = false, int XMAP = 0>
__global__ __launch_bounds__(NW * 64, WPE) void mac_rows_kernel(MacSection sa, MacSection sb,
                                                        const u64* __restrict__ rhat,
                                                        const Mod* __restrict__ mods, u32 k, u32 L) {
#if PVW_TUNING
  if constexpr (STAMP) {
    if (threadIdx.x == 0 && blockIdx.x < PVW_STAMP_MAX) {
      g_stamp_buf[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
      // XCC_ID (hardware register 20, low 4 bits) in bits 28..31 of the word, the CU / SE fields of HW_ID below it
      g_stamp_hw[blockIdx.x] = (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 28) |
                               (__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) & 0x0fffffffu);
    }
  }
#endif
  constexpr int HALF = ELL / 2;   // 16-byte slot pairs per polynomial limb
  constexpr int R = 128 / ELL;    // rows per tile
  constexpr int JC = ELL <= 16 ? 64 : (ELL == 32 ? 32 : 16);  // j per staged r-hat chunk (LDS <= 32 KiB)
  __shared__ v2u64 lds[NW * JC * HALF];
  static_assert(JC * HALF >= 64, "the wave partials reuse the r-hat slabs");

  // section a = A-hat rows (c1), section b = B-hat rows (c2): one launch covers both
  u32 item = blockIdx.x;
  if constexpr (XMAP == 1) {
    const u32 per = gridDim.x >> 3, tail = gridDim.x & 7;       // blocks beyond a multiple of 8 keep their id
    if (item < gridDim.x - tail) item = (item & 7) * per + (item >> 3);
  } else if constexpr (XMAP == 2) {
    const u32 per = (gridDim.x >> 4) << 1, lim = per << 3;
    if (item < lim) { const u32 q = item >> 1, e = item & 1; item = (q & 7) * per + ((q >> 3) << 1) + e; }
  }
  const u32 limb = item % L;
  const u32 rbg = item / L;
  const bool in_a = rbg < sa.row_blocks;
  const u32 rb = in_a ? rbg : rbg - sa.row_blocks;
  const u64* __restrict__ M = in_a ? sa.M : sb.M;
  const u64* addend = in_a ? sa.addend : sb.addend;
  u64* out = in_a ? sa.out : sb.out;
  const u32 nrows = in_a ? sa.nrows : sb.nrows;
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const u32 sp = lane % HALF, rho = lane / HALF;
  // ILV (needs k % (NW*U) == 0): the NW waves interleave groups of U tiles, so the workgroup reads ONE
  // contiguous stream; local tile t of a wave is global tile (t / U) * NW*U + wave * U + t % U
  const u32 kq = ILV ? k / NW : (k + NW - 1) / NW;
  const u32 j0 = ILV ? 0 : (wave * kq < k ? wave * kq : k);
  const u32 j1 = ILV ? kq : ((j0 + kq) < k ? (j0 + kq) : k);
  auto gmap = [&](u32 t) -> u32 { return ILV ? (t / U) * NW * U + wave * U + t % U : t; };

  const v2u64* Mp =
      reinterpret_cast<const v2u64*>(M + ((size_t)rb * L + limb) * (size_t)k * 128) + lane;
  const v2u64* rp = reinterpret_cast<const v2u64*>(rhat + (size_t)limb * k * ELL);
  v2u64* lw = lds + wave * (JC * HALF);

  // the addend of this lane's output (e1 / e2 + m*g, written by the prologue) is requested now by the wave that
  // will write the result: at the end it would cost the workgroup one more exposed memory latency
  const u32 out_row = rb * R + rho;
  const size_t out_o = (((size_t)out_row * L + limb) * ELL) / 2 + sp;
  v2u64 add_pf = (v2u64){0, 0};
  if (wave == 0 && addend && out_row < nrows) add_pf = reinterpret_cast<const v2u64*>(addend)[out_o];

  Acc a0, a1;
  acc_zero(a0);
  acc_zero(a1);
  auto mac2 = [&](const v2u64& xv, const v2u64& yv) {
    if constexpr (ARITH == 2) {
      acc_mac_dev(a0, xv.x, yv.x);
      acc_mac_dev(a1, xv.y, yv.y);
    } else if constexpr (ARITH == 1) {
      acc_mac_dev(a0, xv.x, yv.x);
      a1.ll ^= xv.y ^ yv.y;
    } else {
      a0.ll ^= xv.x ^ yv.x;
      a1.ll ^= xv.y ^ yv.y;
    }
  };

  for (u32 jc = j0; jc < j1; jc += JC) {
    const u32 cnt = (j1 - jc) < (u32)JC ? (j1 - jc) : (u32)JC;
    auto ld = [&](u32 tile) -> v2u64 {
      const v2u64* p = Mp + (size_t)gmap(jc + tile) * 64;
      if constexpr (NT) return __builtin_nontemporal_load(p);
      else return *p;
    };
    __builtin_amdgcn_wave_barrier();
    // all of this lane's r-hat elements are requested before the first is awaited (a load per loop trip
    // would pay one L2 round trip each); indices past the chunk's end are clamped and not stored
    constexpr int RN = JC * HALF / 64;
    v2u64 rv[RN];
#pragma unroll
    for (int x = 0; x < RN; ++x) {
      const u32 idx = lane + 64 * x;
      const u32 ic = idx < cnt * HALF ? idx : 0;
      rv[x] = rp[(size_t)gmap(jc + ic / HALF) * HALF + ic % HALF];
    }
    // (DBUF) the chunk's first U matrix tiles are requested behind them, before the wave waits for r-hat
    v2u64 x[DBUF ? U : 1];
    const bool full = DBUF && cnt >= U;
    if constexpr (DBUF) {
      if (full) {
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = ld(u);
      }
    }
#pragma unroll
    for (int xx = 0; xx < RN; ++xx) {
      const u32 idx = lane + 64 * xx;
      if (idx < cnt * HALF) lw[idx] = rv[xx];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    u32 jj = 0;
    if constexpr (DBUF) {
      // two register buffers: the next U tiles are in flight while the current U are consumed
      if (full) {
        v2u64 xn[U];
        for (; jj + 2 * U <= cnt; jj += U) {
#pragma unroll
          for (int u = 0; u < U; ++u) xn[u] = ld(jj + U + u);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            v2u64 y = lw[(jj + u) * HALF + sp];
            mac2(x[u], y);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) x[u] = xn[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          v2u64 y = lw[(jj + u) * HALF + sp];
          mac2(x[u], y);
        }
        jj += U;
      }
    } else {
      for (; jj + U <= cnt; jj += U) {
        v2u64 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = ld(jj + u);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          v2u64 y = lw[(jj + u) * HALF + sp];
          mac2(x[u], y);
        }
      }
    }
    for (; jj < cnt; ++jj) {
      v2u64 xv = ld(jj);
      v2u64 y = lw[jj * HALF + sp];
      mac2(xv, y);
    }
  }

  // one Barrett reduction per wave partial ("wavefront-wide": q, ratio are SGPRs)
  const Mod m = mods[limb];
  v2u64 part;
  part.x = acc_reduce(a0, m);
  part.y = acc_reduce(a1, m);
  __syncthreads();  // all waves are done with their r-hat slices
  lds[wave * 64 + lane] = part;
  __syncthreads();
  if (wave == 0) {
    if (out_row < nrows) {
      v2u64 s = lds[lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        v2u64 t = lds[w * 64 + lane];
        s.x = addmod(s.x, t.x, m.q);
        s.y = addmod(s.y, t.y, m.q);
      }
      if (addend) {
        s.x = addmod(s.x, add_pf.x, m.q);
        s.y = addmod(s.y, add_pf.y, m.q);
      }
      reinterpret_cast<v2u64*>(out)[out_o] = s;
    }
  }
#if PVW_TUNING
  if constexpr (STAMP) {
    if (threadIdx.x == 0 && blockIdx.x < PVW_STAMP_MAX) g_stamp_buf[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ------------------------------------------------------------------------------------
// mac_rows over the PACKED copy of the tiled matrix.  mac_rows is bound by the bytes it streams, and a residue of a
// 61-bit modulus carries 61 bits in an 8-byte word: the packed copy stores, for every (row block, limb, lane), the
// lane's residue pairs (x_j, y_j), j = 0..k-1, as ONE bit stream of 122 bits per j, cut into 16-byte chunks;
// chunk c of the 64 lanes is 1 KiB contiguous, so the loads are exactly mac_rows' (global_load_dwordx4 nt, 1 KiB
// per wave-instruction) -- there are just 61 of them per 64 j instead of 64 (-4.7 % bytes).  64 j are 61 chunks
// exactly, so with k a multiple of 256 every wave owns whole periods (j in [w k/4, (w+1) k/4)) and every shift
// amount is a compile-time constant: the period is unrolled as 4 groups of 16 j, each living in 16 chunks (the last
// chunk of a group is the first of the next and is carried in registers, not loaded again), 15-16 chunks in
// flight per wave while the previous group is multiplied.  Unpacking is two funnel shifts and a mask per residue on
// a VALU that the quarter-rate v_mad_u64_u32 stream leaves half idle.  Same lazy accumulation, same epilogue, same
// results as mac_rows_kernel.  Built lazily by the C ABI (pack61_kernel) next to the tiled matrix, which every
// other consumer keeps using.
// ------------------------------------------------------------------------------------
constexpr u64 PVW_MASK61 = (1ull << 61) - 1;
#ifndef PVW_PACKED_WPC
#define PVW_PACKED_WPC 2                                  // workgroups per CU the register allocation aims at
#endif
__device__ __forceinline__ u64 pk_word(const v2u64 (&a)[16], int idx) { return (idx & 1) ? a[idx >> 1].y : a[idx >> 1].x; }
// the 61 bits at bit offset `bit` of the 2048-bit window a[0..15] (bit is a constant after unrolling)
__device__ __forceinline__ u64 pk_get61(const v2u64 (&a)[16], int bit) {
  const int idx = bit >> 6, sh = bit & 63;
  u64 v = pk_word(a, idx) >> sh;
  if (sh > 3) v |= pk_word(a, idx + 1) << (64 - sh);
  return v & PVW_MASK61;
}
template <int ELL, bool STAMP = false, bool DEEP = false>
__global__ __launch_bounds__(256, PVW_PACKED_WPC) void mac_rows_packed_kernel(MacSection sa, MacSection sb, const u64* __restrict__ rhat,
                                                               const Mod* __restrict__ mods, u32 k, u32 L) {
#if PVW_TUNING
  if constexpr (STAMP) {
    if (threadIdx.x == 0 && blockIdx.x < PVW_STAMP_MAX) {
      g_stamp_buf[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
      g_stamp_hw[blockIdx.x] = (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 28) | (blockIdx.x & 0x0fffffffu);
    }
  }
#endif
  constexpr int HALF = ELL / 2, R = 128 / ELL, JC = 64, NW = 4;
  static_assert(ELL <= 16, "one period of 64 j per r-hat slab");
  __shared__ v2u64 lds[NW * JC * HALF];
  const u32 item = blockIdx.x;
  const u32 limb = item % L, rbg = item / L;
  const bool in_a = rbg < sa.row_blocks;
  const u32 rb = in_a ? rbg : rbg - sa.row_blocks;
  const u64* __restrict__ Pk = in_a ? sa.M : sb.M;           // the PACKED copy of the section
  const u64* addend = in_a ? sa.addend : sb.addend;
  u64* out = in_a ? sa.out : sb.out;
  const u32 nrows = in_a ? sa.nrows : sb.nrows;
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const u32 sp = lane % HALF, rho = lane / HALF;
  const u32 kq = k / NW, periods = kq / 64;                   // the launcher guarantees k % 256 == 0
  const u32 chunks = k / 64 * 61;                             // per (row block, limb)
  const v2u64* Pp = reinterpret_cast<const v2u64*>(Pk) + (((size_t)rb * L + limb) * chunks + (size_t)wave * periods * 61) * 64 + lane;
  const v2u64* rp = reinterpret_cast<const v2u64*>(rhat + (size_t)limb * k * ELL) + (size_t)wave * kq * HALF;
  v2u64* lw = lds + wave * (JC * HALF);
  const u32 out_row = rb * R + rho;
  const size_t out_o = (((size_t)out_row * L + limb) * ELL) / 2 + sp;
  v2u64 add_pf = (v2u64){0, 0};
  if (wave == 0 && addend && out_row < nrows) add_pf = reinterpret_cast<const v2u64*>(addend)[out_o];
  Acc a0, a1;
  acc_zero(a0);
  acc_zero(a1);
  auto ldc = [&](u32 c) -> v2u64 { return __builtin_nontemporal_load(Pp + (size_t)c * 64); };
  if constexpr (DEEP) {
    // k = 256 (one period per wave): three windows, the chunks of group g + 2 requested before group g is multiplied --
    // a group is ~1.2 us of MACs, less than a loaded HBM round trip, so one group of lookahead leaves the wave waiting
    v2u64 wa[16], wb[16], wc[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) wa[u] = ldc(u);               // group 0: chunks 0..15
#pragma unroll
    for (int u = 1; u < 16; ++u) wb[u] = ldc(15 + u);          // group 1: chunks 16..30 (+ carry 15)
    __builtin_amdgcn_wave_barrier();
    constexpr int RN = JC * HALF / 64;
    {
      v2u64 rv[RN];
#pragma unroll
      for (int x = 0; x < RN; ++x) rv[x] = rp[lane + 64 * x];
#pragma unroll
      for (int x = 0; x < RN; ++x) lw[lane + 64 * x] = rv[x];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    auto macs = [&](const int g, const v2u64 (&cur)[16]) {
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int bit = 32 * g + 122 * jj;
        const u64 xv = pk_get61(cur, bit), yv = pk_get61(cur, bit + 61);
        const v2u64 r = lw[(16 * g + jj) * HALF + sp];
        acc_mac_dev(a0, xv, r.x);
        acc_mac_dev(a1, yv, r.y);
      }
    };
#pragma unroll
    for (int u = 1; u < 16; ++u) wc[u] = ldc(30 + u);          // group 2: chunks 31..45 (+ carry 30)
    __builtin_amdgcn_sched_barrier(0);                         // (the loads are to be ISSUED here, not where hipcc finds room)
    macs(0, wa);
    wb[0] = wa[15];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 1; u < 16; ++u) wa[u] = ldc(45 + u);          // group 3: chunks 46..60 (+ carry 45)
    __builtin_amdgcn_sched_barrier(0);
    macs(1, wb);
    wc[0] = wb[15];
    macs(2, wc);
    wa[0] = wc[15];
    macs(3, wa);
  } else {
  v2u64 xa[16], xb[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) xa[u] = ldc(u);                // group 0 of the first period
  for (u32 pd = 0; pd < periods; ++pd) {
    const u32 cb = pd * 61;
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
        for (int u = 0; u < 16; ++u) nxt[u] = ldc(cb + 61 + u);   // group 0 of the next period
      }
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int bit = 32 * g + 122 * jj;
        const u64 xv = pk_get61(cur, bit), yv = pk_get61(cur, bit + 61);
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
  }
  const Mod m = mods[limb];
  v2u64 part;
  part.x = acc_reduce(a0, m);
  part.y = acc_reduce(a1, m);
  __syncthreads();
  lds[wave * 64 + lane] = part;
  __syncthreads();
  if (wave == 0 && out_row < nrows) {
    v2u64 sres = lds[lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const v2u64 t = lds[w * 64 + lane];
      sres.x = addmod(sres.x, t.x, m.q);
      sres.y = addmod(sres.y, t.y, m.q);
    }
    if (addend) {
      sres.x = addmod(sres.x, add_pf.x, m.q);
      sres.y = addmod(sres.y, add_pf.y, m.q);
    }
    reinterpret_cast<v2u64*>(out)[out_o] = sres;
  }
#if PVW_TUNING
  if constexpr (STAMP) {
    if (threadIdx.x == 0 && blockIdx.x < PVW_STAMP_MAX) g_stamp_buf[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// tiled matrix -> packed copy: one thread per (row block, limb, lane) walks its k residue pairs and emits the bit
// stream in 16-byte chunks (reads and writes are both 1 KiB per wave and step; load-time only)
__global__ __launch_bounds__(256) void pack61_kernel(const u64* __restrict__ M, u64* __restrict__ P, u32 k, size_t items,
                                                      u32* __restrict__ wide_flag) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t item = t >> 6;
  const u32 lane = (u32)(t & 63);
  if (item >= items) return;
  const v2u64* src = reinterpret_cast<const v2u64*>(M) + item * (size_t)k * 64 + lane;
  v2u64* dst = reinterpret_cast<v2u64*>(P) + item * (size_t)(k / 64 * 61) * 64 + lane;
  u64 lo = 0, hi = 0, pend = 0;      // bit buffer (lo, hi), `nb` bits used; pend = the even word of the chunk being filled
  u32 nb = 0, words = 0;
  auto push = [&](u64 v) {
    lo |= v << nb;
    if (nb > 3) hi |= v >> (64 - nb);
    nb += 61;
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
    push(v.x & PVW_MASK61);
    push(v.y & PVW_MASK61);
  }
  // a word that does not fit 61 bits (a caller loaded unreduced data): the copy must not be used
  if (seen >> 61) atomicOr(wide_flag, 1u);
}

// ------------------------------------------------------------------------------------
// mac_rows, persistent form.  The grid is sized to what the chip holds at once (workgroups per CU x CUs) and every
// workgroup walks work items (row block, limb), so a slot never idles while the dispatcher tears one workgroup
// down and sets the next one up, the epilogue of an item (Barrett, cross-wave sum, addend, store) runs UNDER the
// first tile loads of the next one, and -- the point of the exercise -- the eight XCDs, which stream at rates up to
// 8 % apart (profiles/r02_mac_timeline.txt: the hardware deals workgroups round-robin over them, so the one-
// workgroup-per-item grid ends 13 us after its fastest XCD has run dry), share the tail of the work dynamically.
//
// Items are cut into NS = 8 contiguous shards; workgroup b belongs to shard b % NS (in practice: its XCD), rank
// b / NS.  It first walks its STATIC share -- item lo + round * W + rank for the first two rounds, no communication
// at all, so every workgroup starts at once and knows its first successor -- and then takes items from its shard's counter, and when that is exhausted from the other shards' counters
// (one u32 per 128-byte line; a single global counter does not work: one address takes ~20 returning atomics
// per microsecond under this load, which is the whole kernel's item rate).  The successor of an item is known
// one item ahead, so the last tile group of item i prefetches the first group and the r-hat slice of item i+1.
// Partial sums alternate between two small LDS buffers and the one barrier per item waits for LDS traffic only
// (s_waitcnt lgkmcnt(0) + s_barrier: __syncthreads() would also drain the prefetched tiles); the chunk count NC
// is a template parameter so that an item is straight-line code and hipcc counts the outstanding loads exactly.
// Same tiling, same lazy accumulation, same per-item arithmetic as mac_rows_kernel -- bit-identical results.
// Termination: a workgroup makes at most (its static rounds) + (pops that return an item) + NS failing pops.
// counters: [s * 32] shard s, [NS * 32] workgroups finished; the last one to finish re-arms them all for the next
// launch on this stream (per-workspace counters: stream order makes that safe).
// ------------------------------------------------------------------------------------
#define PVW_PERSIST_SHARDS 8
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <int ELL, int U, int WPE, int NC, int NW = 4, bool STAMP = false>
__global__ __launch_bounds__(NW * 64, WPE) void mac_rows_persist_kernel(MacSection sa, MacSection sb,
                                                                const u64* __restrict__ rhat,
                                                                const Mod* __restrict__ mods, u32 k, u32 L,
                                                                u32 items, u32 static_cap, u32* __restrict__ counters) {
#if PVW_TUNING
  if constexpr (STAMP) {
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
      g_stamp_wg[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
      g_stamp_wg[2 * blockIdx.x + 1] = ((u64)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32) |
                                       __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    }
  }
#endif
  constexpr int HALF = ELL / 2;
  constexpr int R = 128 / ELL;
  constexpr int JC = 256 / HALF;                 // tiles per r-hat chunk: 4 KiB slabs (64 at l = 8, 32 at l = 16, ...)
  constexpr int RN = JC * HALF / 64;             // = 4 sixteen-byte r-hat elements per lane per chunk
  constexpr int GPC = JC / U;                    // tile groups per chunk
  constexpr int NS = PVW_PERSIST_SHARDS;
  constexpr u32 NONE = 0xffffffffu;
  static_assert(JC % U == 0 && GPC >= 2 && GPC % 2 == 0, "groups are processed in pairs");
  __shared__ v2u64 slab[NW][JC * HALF];          // wave-private r-hat slices (a wave's LDS operations execute in order,
                                                 // so the next chunk's slice can overwrite the slab once the last
                                                 // multiply of the current chunk has been issued)
  __shared__ v2u64 part[2][NW * 64];             // wave partials, two generations
  __shared__ u32 nextslot[2];

  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const u32 sp = lane % HALF, rho = lane / HALF;

  // ---- work distribution (everything here is workgroup-uniform) ----
  const u32 W = gridDim.x / NS;                  // workgroups per shard (the launcher makes the grid a multiple of NS)
  const u32 my_shard = blockIdx.x % NS, rank = blockIdx.x / NS;
  auto shard_lo = [&](u32 sh) -> u32 { return (u32)(((u64)items * sh) / NS); };
  auto static_rounds = [&](u32 sh) -> u32 {      // rounds of W items every workgroup of the shard takes unasked
    const u32 full = (shard_lo(sh + 1) - shard_lo(sh)) / W;
    return full < static_cap ? full : static_cap;
  };
  u32 round = 0;                                 // static rounds taken so far
  u32 steal = 0;                                 // shards found empty so far (thread 0 only meaningful)
  const u32 my_static = static_rounds(my_shard);
  // thread 0: next item from the counters, own shard first, then the others in turn
  auto pop = [&]() -> u32 {
    while (steal < (u32)NS) {
      const u32 sh = (my_shard + steal) % NS;
      const u32 base = shard_lo(sh) + static_rounds(sh) * W, hi = shard_lo(sh + 1);
      const u32 v = atomicAdd(&counters[sh * 32], 1u);
      if (base + v < hi) return base + v;
      ++steal;
    }
    return NONE;
  };
  // next item of this workgroup: arithmetic while the static share lasts, the counters afterwards
  // (gen = parity of the LDS word the dynamic answer travels through)
  auto next_static = [&]() -> u32 { return shard_lo(my_shard) + (round++) * W + rank; };

  // an item's wave-uniform description, kept in separate scalars (a struct copied in the loop goes through the
  // stack, and a kernel with a scratch segment ramps up an order of magnitude more slowly): the wave's first tile,
  // the r-hat of the item's limb, limb, row block, section
#define PVW_ITEM_DECL(P) const v2u64* P##mp; const v2u64* P##rp; u32 P##limb, P##rb; bool P##in_a
#define PVW_ITEM_SETUP(P, it)                                                                                          \
  do {                                                                                                                 \
    P##limb = (it) % L;                                                                                                \
    const u32 rbg_ = (it) / L;                                                                                         \
    P##in_a = rbg_ < sa.row_blocks;                                                                                    \
    P##rb = P##in_a ? rbg_ : rbg_ - sa.row_blocks;                                                                     \
    const u64* M_ = P##in_a ? sa.M : sb.M;                                                                             \
    P##mp = reinterpret_cast<const v2u64*>(M_ + ((size_t)P##rb * L + P##limb) * (size_t)k * 128) + (size_t)wave * U * 64; \
    P##rp = reinterpret_cast<const v2u64*>(rhat + (size_t)P##limb * k * ELL);                                          \
  } while (0)
  // local tile t of this wave is global tile (t / U) * NW * U + wave * U + t % U (the waves interleave groups of U)
  auto ld_group = [&](const v2u64* mp, u32 g, v2u64 (&dst)[U]) {
    const v2u64* p = mp + (size_t)g * (NW * U * 64);        // scalar base; the lane offset is the only vector part
#pragma unroll
    for (int u = 0; u < U; ++u) dst[u] = __builtin_nontemporal_load(p + u * 64 + lane);
  };
  // this lane's RN r-hat elements of a chunk: element idx = lane + 64 x of the chunk, i.e. slot pair idx % HALF of
  // local tile chunk * JC + idx / HALF
  u32 roff[RN];
#pragma unroll
  for (int x = 0; x < RN; ++x) {
    const u32 idx = lane + 64 * x, t = idx / HALF;            // t < JC: position inside a chunk
    roff[x] = ((t / U) * (NW * U) + wave * U + t % U) * HALF + idx % HALF;
  }
  auto fetch_r = [&](const v2u64* rp, u32 chunk, v2u64 (&rv)[RN]) {
    const v2u64* p = rp + (size_t)chunk * (JC / U) * (NW * U) * HALF;     // whole groups per chunk (JC % U == 0)
#pragma unroll
    for (int x = 0; x < RN; ++x) rv[x] = p[roff[x]];
  };

  // ---- the first two items ----
  u32 cur, nxt;
  if (my_static >= 2) {
    cur = next_static();
    nxt = next_static();
  } else {
    if (threadIdx.x == 0) {
      u32 a = my_static >= 1 ? next_static() : pop();
      u32 b2 = a == NONE ? NONE : pop();
      nextslot[0] = a;
      nextslot[1] = b2;
    }
    __syncthreads();
    cur = __builtin_amdgcn_readfirstlane(nextslot[0]);
    nxt = __builtin_amdgcn_readfirstlane(nextslot[1]);
    round = my_static;                           // the static share (0 or 1 item) is used up
    __syncthreads();
  }
  u32 gen = 0;                                   // item generation: partial buffer / nextslot parity
  if (cur != NONE) {
    PVW_ITEM_DECL(c_);
    PVW_ITEM_DECL(n_);
    PVW_ITEM_SETUP(c_, cur);
    v2u64 xa[U], xb[U], rv[RN];
    fetch_r(c_rp, 0, rv);
    ld_group(c_mp, 0, xa);
    for (;;) {
#if PVW_TUNING
      if constexpr (STAMP) {
        if (threadIdx.x == 0 && cur < PVW_STAMP_MAX) {
          g_stamp_buf[2 * cur] = __builtin_amdgcn_s_memrealtime();
          g_stamp_hw[cur] = (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 28) | (blockIdx.x & 0x0fffffffu);
        }
      }
#endif
      // the item after `nxt`: arithmetic in the static phase; otherwise thread 0 asks the counters now and the
      // answer is published at this item's barrier
      const bool nn_static = round < my_static;  // uniform
      u32 nn = NONE;
      if (nn_static) nn = next_static();
      else if (threadIdx.x == 0 && nxt != NONE) nn = pop();
      const bool have_next = nxt != NONE;        // workgroup-uniform
      PVW_ITEM_SETUP(n_, have_next ? nxt : cur);
      // the addend of this lane's output (e1 / e2 + m*g, written by the prologue), requested before the tile stream
      // so that at the end it is the oldest load in flight
      const u32 nrows = c_in_a ? sa.nrows : sb.nrows;
      const u64* addend = c_in_a ? sa.addend : sb.addend;
      u64* out = c_in_a ? sa.out : sb.out;
      const u32 out_row = c_rb * R + rho;
      const size_t out_o = (((size_t)out_row * L + c_limb) * ELL) / 2 + sp;
      v2u64 add_pf = (v2u64){0, 0};
      if (wave == 0 && addend && out_row < nrows) add_pf = reinterpret_cast<const v2u64*>(addend)[out_o];
      Acc a0, a1;
      acc_zero(a0);
      acc_zero(a1);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        v2u64* lw = slab[wave];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int x = 0; x < RN; ++x) lw[lane + 64 * x] = rv[x];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // one group: prefetch what comes next into `nx`, multiply `cx` against the slab
        auto group = [&](int g, v2u64 (&cx)[U], v2u64 (&nx)[U]) {
          if (g + 1 < GPC) {
            ld_group(c_mp, c * GPC + g + 1, nx);
          } else if (c + 1 < NC) {
            fetch_r(c_rp, c + 1, rv);
            ld_group(c_mp, (c + 1) * GPC, nx);
          } else if (have_next) {
            fetch_r(n_rp, 0, rv);
            ld_group(n_mp, 0, nx);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const v2u64 y = lw[(g * U + u) * HALF + sp];
            acc_mac_dev(a0, cx[u].x, y.x);
            acc_mac_dev(a1, cx[u].y, y.y);
          }
        };
#pragma unroll
        for (int g = 0; g < GPC; g += 2) {
          group(g, xa, xb);
          group(g + 1, xb, xa);
        }
      }
      // epilogue of item `cur` (its successor's first loads are in flight): one Barrett per wave partial, cross-wave
      // sum through LDS, addend, store
      const Mod m = mods[c_limb];
      v2u64 pt;
      pt.x = acc_reduce(a0, m);
      pt.y = acc_reduce(a1, m);
      part[gen][wave * 64 + lane] = pt;
      if (!nn_static && threadIdx.x == 0) nextslot[gen] = nn;
      lds_barrier();
      if (!nn_static) nn = __builtin_amdgcn_readfirstlane(nextslot[gen]);
      if (wave == 0 && out_row < nrows) {
        v2u64 sres = pt;
#pragma unroll
        for (int w = 1; w < NW; ++w) {
          const v2u64 t = part[gen][w * 64 + lane];
          sres.x = addmod(sres.x, t.x, m.q);
          sres.y = addmod(sres.y, t.y, m.q);
        }
        if (addend) {
          sres.x = addmod(sres.x, add_pf.x, m.q);
          sres.y = addmod(sres.y, add_pf.y, m.q);
        }
        reinterpret_cast<v2u64*>(out)[out_o] = sres;
      }
#if PVW_TUNING
      if constexpr (STAMP) {
        if (threadIdx.x == 0 && cur < PVW_STAMP_MAX) g_stamp_buf[2 * cur + 1] = __builtin_amdgcn_s_memrealtime();
      }
#endif
      gen ^= 1;
      if (!have_next) break;
      cur = nxt;
      nxt = nn;
      c_mp = n_mp; c_rp = n_rp; c_limb = n_limb; c_rb = n_rb; c_in_a = n_in_a;
    }
  }
#undef PVW_ITEM_DECL
#undef PVW_ITEM_SETUP
  if (threadIdx.x == 0) {
    const u32 done = atomicAdd(&counters[NS * 32], 1u);
    if (done == gridDim.x - 1) {
#pragma unroll
      for (int sh = 0; sh <= NS; ++sh) counters[sh * 32] = 0;
    }
  }
}

#if PVW_TUNING
// mac_rows, continuous-stream schedule: the B-hat tile stream of a wave never drains at an r-hat chunk
// boundary.  The r-hat slice of the NEXT chunk is fetched into registers one group ahead (so its loads
// sit in front of the tile prefetch in the in-order vmcnt queue) and dropped into the wave-private LDS
// slab when the current chunk has been consumed.
template <int ELL, int U, bool NT>
__global__ __launch_bounds__(256) void mac_rows_stream_kernel(MacSection sa, MacSection sb,
                                                               const u64* __restrict__ rhat,
                                                               const Mod* __restrict__ mods, u32 k, u32 L) {
  constexpr int HALF = ELL / 2;
  constexpr int R = 128 / ELL;
  constexpr int JC = ELL <= 16 ? 64 : (ELL == 32 ? 32 : 16);
  constexpr int RN = JC * HALF / 64;          // 16-byte r-hat elements per lane per chunk
  static_assert(JC % U == 0, "a group of U tiles must not straddle an r-hat chunk");
  __shared__ v2u64 lds[4 * JC * HALF];

  const u32 limb = blockIdx.x % L;
  const u32 rbg = blockIdx.x / L;
  const bool in_a = rbg < sa.row_blocks;
  const u32 rb = in_a ? rbg : rbg - sa.row_blocks;
  const u64* __restrict__ M = in_a ? sa.M : sb.M;
  const u64* addend = in_a ? sa.addend : sb.addend;
  u64* out = in_a ? sa.out : sb.out;
  const u32 nrows = in_a ? sa.nrows : sb.nrows;

  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const u32 sp = lane % HALF, rho = lane / HALF;
  const u32 kq = (k + 3) / 4;
  const u32 j0 = wave * kq < k ? wave * kq : k;
  const u32 j1 = (j0 + kq) < k ? (j0 + kq) : k;
  const u32 total = j1 - j0;

  const v2u64* mp = reinterpret_cast<const v2u64*>(M + ((size_t)rb * L + limb) * (size_t)k * 128) + lane + (size_t)j0 * 64;
  const v2u64* rp = reinterpret_cast<const v2u64*>(rhat + (size_t)limb * k * ELL) + (size_t)j0 * HALF;
  v2u64* lw = lds + wave * (JC * HALF);
  auto ld = [&](size_t tile) -> v2u64 {
    if constexpr (NT) return __builtin_nontemporal_load(mp + tile * 64);
    else return mp[tile * 64];
  };
  // r-hat chunk starting at local tile `base` -> registers (clamped reads past the end are never used)
  auto fetch_r = [&](u32 base, v2u64 (&rn)[RN]) {
#pragma unroll
    for (int x = 0; x < RN; ++x) {
      u32 idx = base * HALF + lane + 64 * x;
      u32 lim = total * HALF;
      rn[x] = rp[idx < lim ? idx : (lim ? lim - 1 : 0)];
    }
  };
  auto store_r = [&](const v2u64 (&rn)[RN]) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int x = 0; x < RN; ++x) lw[lane + 64 * x] = rn[x];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  Acc a0, a1;
  acc_zero(a0);
  acc_zero(a1);
  const u32 G = total / U;
  v2u64 x[U], xn[U], rn[RN];
  if (total) fetch_r(0, rn);
  if (G) {
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = ld(u);
  }
  if (total) store_r(rn);
  u32 chunk_base = 0;
  for (u32 g = 0; g < G; ++g) {
    const u32 t0 = g * U;
    const bool last_of_chunk = ((t0 + U) % JC) == 0 && (t0 + U) < total;
    if (last_of_chunk) fetch_r(t0 + U, rn);            // ahead of the tile prefetch in the load queue
    if (g + 1 < G) {
#pragma unroll
      for (int u = 0; u < U; ++u) xn[u] = ld(t0 + U + u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v2u64 y = lw[(t0 - chunk_base + u) * HALF + sp];
      acc_mac_dev(a0, x[u].x, y.x);
      acc_mac_dev(a1, x[u].y, y.y);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = xn[u];
    if (last_of_chunk) {
      store_r(rn);
      chunk_base = t0 + U;
    }
  }
  for (u32 jj = G * U; jj < total; ++jj) {                // ragged tail (k not a multiple of 4*U)
    if (jj - chunk_base >= (u32)JC) {
      fetch_r(jj, rn);
      store_r(rn);
      chunk_base = jj;
    }
    v2u64 xv = mp[(size_t)jj * 64];
    v2u64 y = lw[(jj - chunk_base) * HALF + sp];
    acc_mac_dev(a0, xv.x, y.x);
    acc_mac_dev(a1, xv.y, y.y);
  }

  const Mod m = mods[limb];
  v2u64 part;
  part.x = acc_reduce(a0, m);
  part.y = acc_reduce(a1, m);
  __syncthreads();
  lds[wave * 64 + lane] = part;
  __syncthreads();
  if (wave == 0) {
    const u32 row = rb * R + rho;
    if (row < nrows) {
      v2u64 s = lds[lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        v2u64 t = lds[w * 64 + lane];
        s.x = addmod(s.x, t.x, m.q);
        s.y = addmod(s.y, t.y, m.q);
      }
      const size_t o = (((size_t)row * L + limb) * ELL) / 2 + sp;
      if (addend) {
        v2u64 e = reinterpret_cast<const v2u64*>(addend)[o];
        s.x = addmod(s.x, e.x, m.q);
        s.y = addmod(s.y, e.y, m.q);
      }
      reinterpret_cast<v2u64*>(out)[o] = s;
    }
  }
}

#endif  // PVW_TUNING

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
        v2u64 t = lds[w * 64 + lane];
        s.x = addmod(s.x, t.x, m.q);
        s.y = addmod(s.y, t.y, m.q);
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

// ------------------------------------------------------------------------------------
// prep: small signed coefficients -> RNS -> l-point NTT (+ scalar * g-hat), one thread per
// (polynomial, limb).  Serves r-hat, the e1/e2 addends, encode_scalar
// (src/params/parameters.rs:346-367) and Poly::from_coefficients + NTT
// (encryption.rs:147-154, secret_key.rs:98-112).
// ------------------------------------------------------------------------------------
template <int ELL>
__global__ __launch_bounds__(64) void prep_kernel(const i64* __restrict__ coeffs,
                                                   const u64* __restrict__ scalars,
                                                   u64* __restrict__ out, size_t stride_poly,
                                                   size_t stride_limb, u32 count, u32 L,
                                                   u32 do_ntt, DevTables t, u32 group, size_t stride_group) {
  const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= count * L) return;
  const u32 p = tid / L, limb = tid % L;
  const Mod m = t.mods[limb];
  u64 a[ELL];
#pragma unroll
  for (int s = 0; s < ELL; ++s) a[s] = signed_residue(coeffs[(size_t)p * ELL + s], m);
  if (do_ntt) ntt_forward<ELL>(a, t.tw + (size_t)limb * ELL, t.twp + (size_t)limb * ELL, m);
  if (scalars) {
    // `scalars[i] as i64` wrap (encryption.rs:195), then scalar * g  (parameters.rs:346-367)
    const u64 mr = signed_residue((i64)scalars[p], m);
    const u64* g = (do_ntt ? t.ghat : t.gpow) + (size_t)limb * ELL;
    const u64* gp = (do_ntt ? t.ghatp : t.gpowp) + (size_t)limb * ELL;
#pragma unroll
    for (int s = 0; s < ELL; ++s) a[s] = addmod(a[s], mulmod_shoup(mr, g[s], gp[s], m.q), m.q);
  }
  u64* o = out + (group ? (size_t)(p / group) * stride_group + (size_t)(p % group) * stride_poly : (size_t)p * stride_poly) +
           (size_t)limb * stride_limb;
#pragma unroll
  for (int s = 0; s < ELL; s += 2)
    *reinterpret_cast<v2u64*>(o + s) = (v2u64){a[s], a[s + 1]};
}

// dst[c][j] = src[j][c] over a k x k matrix of polynomials (`words` u64 each): key generation walks the
// CRS by columns (crs.rs:152-168)
__global__ __launch_bounds__(256) void transpose_polys_kernel(const u64* __restrict__ src, u64* __restrict__ dst,
                                                               u32 k, u32 words) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)k * k * words) return;
  const u32 x = idx % words;
  const size_t pj = idx / words;
  const u32 j = pj / k, c = pj % k;
  dst[((size_t)c * k + j) * words + x] = src[idx];
}

// in-place change_representation on [count][L][l] polynomials
template <int ELL>
__global__ __launch_bounds__(64) void ntt_kernel(u64* __restrict__ polys, u32 count, u32 L,
                                                  u32 inverse, DevTables t) {
  const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= count * L) return;
  const u32 limb = tid % L;
  const Mod m = t.mods[limb];
  u64* p = polys + (size_t)tid * ELL;
  u64 a[ELL];
#pragma unroll
  for (int s = 0; s < ELL; s += 2) {
    v2u64 v = *reinterpret_cast<const v2u64*>(p + s);
    a[s] = v.x;
    a[s + 1] = v.y;
  }
  if (inverse) ntt_inverse<ELL>(a, t.itw + (size_t)limb * ELL, t.itwp + (size_t)limb * ELL, t.linv[limb], t.linvp[limb], m);
  else ntt_forward<ELL>(a, t.tw + (size_t)limb * ELL, t.twp + (size_t)limb * ELL, m);
#pragma unroll
  for (int s = 0; s < ELL; s += 2)
    *reinterpret_cast<v2u64*>(p + s) = (v2u64){a[s], a[s + 1]};
}

// ------------------------------------------------------------------------------------
// tile / untile: API layout [row][j][L][l] <-> tiled M, optional NTT on the way.
// One thread per (row, j, limb).
// ------------------------------------------------------------------------------------
template <int ELL>
__global__ __launch_bounds__(256) void tile_kernel(const u64* __restrict__ src, u64* __restrict__ M,
                                                    u32 rows, u32 row0_tiled, u32 k, u32 L,
                                                    u32 ntt_first, DevTables t) {
  constexpr int R = 128 / ELL;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (size_t)rows * k * L) return;
  const u32 limb = tid % L;
  const u32 j = (tid / L) % k;
  const u32 row = tid / ((size_t)L * k);
  const u64* p = src + tid * ELL;
  u64 a[ELL];
#pragma unroll
  for (int s = 0; s < ELL; s += 2) {
    v2u64 v = *reinterpret_cast<const v2u64*>(p + s);
    a[s] = v.x;
    a[s + 1] = v.y;
  }
  if (ntt_first) ntt_forward<ELL>(a, t.tw + (size_t)limb * ELL, t.twp + (size_t)limb * ELL, t.mods[limb]);
  const u32 trow = row0_tiled + row;
  u64* o = M + (((size_t)(trow / R) * L + limb) * k + j) * 128 + (trow % R) * ELL;
#pragma unroll
  for (int s = 0; s < ELL; s += 2)
    *reinterpret_cast<v2u64*>(o + s) = (v2u64){a[s], a[s + 1]};
}

template <int ELL>
__global__ __launch_bounds__(256) void untile_kernel(const u64* __restrict__ M, u64* __restrict__ dst,
                                                      u32 rows, u32 row0_tiled, u32 k, u32 L,
                                                      u32 intt_after, DevTables t) {
  constexpr int R = 128 / ELL;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (size_t)rows * k * L) return;
  const u32 limb = tid % L;
  const u32 j = (tid / L) % k;
  const u32 row = tid / ((size_t)L * k);
  const u32 trow = row0_tiled + row;
  const u64* p = M + (((size_t)(trow / R) * L + limb) * k + j) * 128 + (trow % R) * ELL;
  u64 a[ELL];
#pragma unroll
  for (int s = 0; s < ELL; s += 2) {
    v2u64 v = *reinterpret_cast<const v2u64*>(p + s);
    a[s] = v.x;
    a[s + 1] = v.y;
  }
  if (intt_after) ntt_inverse<ELL>(a, t.itw + (size_t)limb * ELL, t.itwp + (size_t)limb * ELL, t.linv[limb], t.linvp[limb], t.mods[limb]);
  u64* o = dst + tid * ELL;
#pragma unroll
  for (int s = 0; s < ELL; s += 2)
    *reinterpret_cast<v2u64*>(o + s) = (v2u64){a[s], a[s + 1]};
}

// uniform residues straight into the tiled matrix: polynomial (grow, j), limb i uses ChaCha8
// stream (domain << 32) | ((grow*k + j)*L + i)   (grow = global row index)
template <int ELL>
__global__ __launch_bounds__(256) void fill_uniform_tiled_kernel(u64* __restrict__ M, ChaChaKey key,
                                                                  u32 domain, u32 rows,
                                                                  u32 row0_tiled, u32 grow0, u32 k,
                                                                  u32 L, DevTables t) {
  constexpr int R = 128 / ELL;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (size_t)rows * k * L) return;
  const u32 limb = tid % L;
  const u32 j = (tid / L) % k;
  const u32 row = tid / ((size_t)L * k);
  ChaChaRng g;
  g.init(key, domain, (u32)((((size_t)(grow0 + row)) * k + j) * L + limb));
  u64 a[ELL];
  const u64 q = t.mods[limb].q;
  const u32 sh = (u32)__clzll((long long)q);
#pragma unroll
  for (int s = 0; s < ELL; ++s) {
    u64 v;
    do { v = g.next_u64() >> sh; } while (v >= q);
    a[s] = v;
  }
  const u32 trow = row0_tiled + row;
  u64* o = M + (((size_t)(trow / R) * L + limb) * k + j) * 128 + (trow % R) * ELL;
#pragma unroll
  for (int s = 0; s < ELL; s += 2)
    *reinterpret_cast<v2u64*>(o + s) = (v2u64){a[s], a[s + 1]};
}

// ------------------------------------------------------------------------------------
// samplers: one thread per polynomial, coefficient order and word consumption as the
// reference's samplers (src/sampling/uniform.rs).  out [count][l] i64.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sample_kernel(i64* __restrict__ out, ChaChaKey key, u32 l,
                                                     SampleJob j0, SampleJob j1, SampleJob j2) {
  const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  SampleJob job;
  u32 local;
  if (tid < j0.count) { job = j0; local = tid; }
  else if (tid < j0.count + j1.count) { job = j1; local = tid - j0.count; }
  else if (tid < j0.count + j1.count + j2.count) { job = j2; local = tid - j0.count - j1.count; }
  else return;
  ChaChaRng g;
  g.init(key, job.domain, job.index0 + local);
  i64* o = out + ((size_t)job.out_poly0 + local) * l;
  auto emit = [o](u32 s, i64 v) { o[s] = v; };
  if (job.kind == SAMPLE_CBD) sample_cbd_poly(g, l, job.cbd_half != 0, job.cbd_v, emit);
  else sample_uniform_poly(g, l, job.bound, emit);
}

// ------------------------------------------------------------------------------------
// prologue: everything encrypt needs before the streamed MAC, in ONE launch
// (encryption.rs:135-154 r, :161-167 e1, :195-196 encode + e2): each block takes PB <= 64
// polynomials, samples (or copies) their small coefficients into LDS with one thread per
// polynomial, then one thread per (polynomial, limb) reduces, transforms and stores.
// ------------------------------------------------------------------------------------
// Latency is what this kernel is made of (one encrypt's worth is 4608 polynomials: a launch that cannot fill the chip
// for long), so the dependent memory round trips are counted: the batch descriptor travels in the kernel-argument
// segment (host memory behind PCIe unless the runtime keeps kernel arguments on the device -- every dependent read of
// it costs microseconds) and is therefore read ONCE, by one wide load per workgroup into LDS; job look-ups after
// that are LDS reads.  The inputs of the transform phase that live in device memory (the party's scalar, the limb's
// modulus) are requested before the sampling phase and arrive under it.
//   hop 1 scalar header (implicit)  ->  hop 2 descriptor -> LDS  ->  [ sampling || table staging, scalar / modulus
//   loads ]  ->  transform  ->  store
template <int ELL>
__global__ __launch_bounds__(256) void prologue_kernel(PrologueBatch b, u32 L, u32 PB, u32 stage_tables, DevTables t) {
  extern __shared__ u64 psm[];
  i64* sc = reinterpret_cast<i64*>(psm);              // [PB][ELL] sampled coefficients
  u64* tab = psm + (size_t)PB * ELL;                  // [4][L][ELL] tw | twp | ghat | ghatp (if staged)
  constexpr u32 JOB_WORDS = sizeof(PrologueJob) / 4, KEY_WORDS = sizeof(ChaChaKey) / 4;
  static_assert(sizeof(PrologueJob) % 8 == 0, "descriptor copy is word-wise");
  u32* jobw = reinterpret_cast<u32*>(tab + (stage_tables ? (size_t)4 * L * ELL : 0));   // [njobs] PrologueJob
  u32* keyw = jobw + PVW_MAX_PROLOGUE_JOBS * JOB_WORDS;                                  // [key_window] ChaChaKey
  const PrologueJob* jobs = reinterpret_cast<const PrologueJob*>(jobw);
  const ChaChaKey* keys = reinterpret_cast<const ChaChaKey*>(keyw);
  const u32 gp0 = blockIdx.x * PB;
  const u32 tid = threadIdx.x;
  const u32 rep = blockIdx.y;                         // replica (dealer / party) of the template jobs
  // ---- the descriptor: one coalesced read of the jobs and of this replica's key window ----
  {
    const u32* src = reinterpret_cast<const u32*>(&b.job[0]);
    const u32 nw = b.njobs * JOB_WORDS;
    for (u32 w = tid; w < nw; w += 256) jobw[w] = src[w];
    const u32* ksrc = reinterpret_cast<const u32*>(&b.key[rep * b.key_rep]);
    const u32 kw = b.key_window * KEY_WORDS;
    for (u32 w = tid; w < kw; w += 256) keyw[w] = ksrc[w];
  }
  if (stage_tables && tid >= 64) {
    // the three waves that do not sample bring the twiddle / gadget tables into LDS
    const u32 n = L * ELL;
    for (u32 x = tid - 64; x < n; x += 192) {
      tab[x] = t.tw[x];
      tab[n + x] = t.twp[x];
      tab[2 * n + x] = t.ghat[x];
      tab[3 * n + x] = t.ghatp[x];
    }
  }
  __syncthreads();
  // locate (job, local polynomial) of global polynomial gp: jobs are laid end to end
  auto locate = [&](u32 gp, u32& ji, u32& local) {
    ji = 0;
    local = gp;
#pragma unroll
    for (u32 x = 0; x + 1 < PVW_MAX_PROLOGUE_JOBS; ++x)
      if (ji == x && x + 1 < b.njobs && local >= jobs[x].sj.count) { local -= jobs[x].sj.count; ji = x + 1; }
  };
  // ---- this thread's (polynomial, limb) of the transform phase (first trip): request what it needs from device
  // memory now, so that it arrives while wave 0 samples ----
  const u32 p0 = tid / L, limb0 = tid % L;
  const bool work0 = tid < PB * L && gp0 + p0 < b.total;
  u32 ji0 = 0, local0 = 0;
  Mod m0 = Mod{1, 0, 0};
  u64 scalar0 = 0;
  if (work0) {
    locate(gp0 + p0, ji0, local0);
    m0 = t.mods[limb0];
    if (jobs[ji0].scalars) scalar0 = jobs[ji0].scalars[(size_t)rep * jobs[ji0].rep_scalars + local0];
  }
  if (tid < PB && tid < 64 && gp0 + tid < b.total && !PVW_PDBG(b, 1)) {
    u32 ji, local;
    locate(gp0 + tid, ji, local);
    const PrologueJob& job = jobs[ji];
    i64* o = sc + tid * ELL;
    if (job.explicit_coeffs) {
      const i64* ec = job.explicit_coeffs + (size_t)rep * job.rep_coeffs;
#pragma unroll
      for (int s = 0; s < ELL; ++s) o[s] = ec[(size_t)local * ELL + s];
    } else {
      ChaChaRng g;
      g.init(keys[job.key_idx], job.sj.domain, job.sj.index0 + rep * job.rep_index0 + local);
      auto emit = [o](u32 s, i64 v) { o[s] = v; };
      if (job.sj.kind == SAMPLE_CBD) sample_cbd_poly(g, ELL, job.sj.cbd_half != 0, job.sj.cbd_v, emit);
      else sample_uniform_poly(g, ELL, job.sj.bound, emit);
    }
  }
  __syncthreads();
  if (PVW_PDBG(b, 2)) return;
  const u32 n = L * ELL;
  // one thread per (polynomial, limb); a block of PB <= 64 polynomials takes ceil(PB * L / 256) trips
  for (u32 idx = tid; idx < PB * L; idx += 256) {
    const u32 p = idx / L, limb = idx % L;
    if (gp0 + p >= b.total) break;
    u32 ji = ji0, local = local0;
    Mod m = m0;
    u64 scalar = scalar0;
    if (idx != tid) {                                    // later trips (PB * L > 256): the same look-ups, not prefetched
      locate(gp0 + p, ji, local);
      m = t.mods[limb];
      scalar = jobs[ji].scalars ? jobs[ji].scalars[(size_t)rep * jobs[ji].rep_scalars + local] : 0;
    }
    const PrologueJob& job = jobs[ji];
    const u64* tw = stage_tables ? tab + (size_t)limb * ELL : t.tw + (size_t)limb * ELL;
    const u64* twp = stage_tables ? tab + n + (size_t)limb * ELL : t.twp + (size_t)limb * ELL;
    u64 a[ELL];
#pragma unroll
    for (int s = 0; s < ELL; ++s) a[s] = signed_residue(sc[p * ELL + s], m);
    ntt_forward<ELL>(a, tw, twp, m);
    if (job.scalars) {
      const u64 mr = signed_residue((i64)scalar, m);     // `as i64` wrap, encryption.rs:195
      const u64* g = stage_tables ? tab + 2 * n + (size_t)limb * ELL : t.ghat + (size_t)limb * ELL;
      const u64* gp = stage_tables ? tab + 3 * n + (size_t)limb * ELL : t.ghatp + (size_t)limb * ELL;
#pragma unroll
      for (int s = 0; s < ELL; ++s) a[s] = addmod(a[s], mulmod_shoup(mr, g[s], gp[s], m.q), m.q);
    }
    u64* o = job.out + (size_t)rep * job.rep_out + (size_t)local * job.stride_poly + (size_t)limb * job.stride_limb;
#pragma unroll
    for (int s = 0; s < ELL; s += 2) *reinterpret_cast<v2u64*>(o + s) = (v2u64){a[s], a[s + 1]};
  }
}

// truncated discrete Gaussian (src/sampling/normal.rs:136-190): one thread per sample.
__device__ __forceinline__ double unit_f64(ChaChaRng& g) {
  return (double)(g.next_u64() >> 11) * (1.0 / 9007199254740992.0);
}
__global__ __launch_bounds__(64) void gaussian_kernel(i64* __restrict__ out, ChaChaKey key,
                                                       u32 index0, u32 count, u64 bound) {
  const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= count) return;
  ChaChaRng g;
  g.init(key, DOM_GAUSS, index0 + tid);
  if (bound == 0) { out[tid] = 0; return; }            // normal.rs:137-139
  const double bf = (double)bound;
  if (bf > 1e15) {                                      // :144-149
    const i64 sign = (g.next_u32() >> 31) ? 1 : -1;
    out[tid] = sign * (i64)(g.next_u32() % 1000001u);
    return;
  }
  const double sigma = bf / 16.96;                      // :8,:151
  double ratio = 0.0;
  bool have = false;
  if (sigma > 0.3) {                                    // :168-170
    ratio = 2.0 * unit_f64(g) - 1.0;
    have = true;
  } else {
    for (int it = 0; it < 1000 && !have; ++it) {        // :173-179
      const double eps = 2.220446049250313e-16;
      const double u1 = eps + (1.0 - eps) * unit_f64(g);
      const double u2 = unit_f64(g);
      const double z = sqrt(-2.0 * log(u1)) * cos(2.0 * 3.14159265358979323846 * u2);  // :186-190
      const double r = z * sigma;
      if (r >= -1.0 && r <= 1.0) { ratio = r; have = true; }
    }
    if (!have) ratio = 2.0 * unit_f64(g) - 1.0;         // :182
  }
  const double fx = ratio * bf;                         // ratio_to_bigint fast path :199-204
  i64 x = (i64)floor(fabs(fx) + 0.5);
  if (fx < 0) x = -x;
  const i64 b = (i64)bound;
  out[tid] = x > b ? b : (x < -b ? -b : x);             // :156-160
}

// ------------------------------------------------------------------------------------
// decrypt_mac: noisy[d] = sum_j shat[j] (.) c1s[d][j] - c2col[d]   (decryption.rs:257-274)
// on the ciphertext layout as it arrives, [d][j][L][l].  One workgroup per dealer; thread
// (g, e) owns slot pair e of the polynomial and the j = g, g+c, g+2c, ... terms.
// ------------------------------------------------------------------------------------
template <int U, bool DBUF>
__global__ __launch_bounds__(1024) void decrypt_mac_kernel(const u64* __restrict__ c1s,
                                                            const u64* __restrict__ shat,
                                                            const u64* __restrict__ c2col,
                                                            u64* __restrict__ noisy,
                                                            const Mod* __restrict__ mods, u32 k,
                                                            u32 ell, u32 pairs, u32 c,
                                                            u32 pair0_step) {
  extern __shared__ v2u64 dl[];
  const u32 d = blockIdx.x;
  const u32 pair_base = blockIdx.y * pair0_step;
  const u32 chunk = (pairs - pair_base) < pair0_step ? (pairs - pair_base) : pair0_step;
  const u32 g = threadIdx.x / chunk, el = threadIdx.x % chunk;
  const bool active = threadIdx.x < c * chunk;
  const u32 e = pair_base + el;
  const v2u64* cp = reinterpret_cast<const v2u64*>(c1s) + (size_t)d * k * pairs + e;
  const v2u64* sp = reinterpret_cast<const v2u64*>(shat) + e;
  Acc a0, a1;
  acc_zero(a0);
  acc_zero(a1);
  if (active) {
    u32 j = g;
    const size_t stride = (size_t)c * pairs;       // one step of this thread through j
    if constexpr (DBUF) {
      if (j + (U - 1) * c < k) {
        v2u64 x[U], y[U], xn[U], yn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          x[u] = __builtin_nontemporal_load(cp + (size_t)j * pairs + u * stride);
          y[u] = sp[(size_t)j * pairs + u * stride];
        }
        for (; j + (2 * U - 1) * c < k; j += U * c) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            xn[u] = __builtin_nontemporal_load(cp + (size_t)(j + U * c) * pairs + u * stride);
            yn[u] = sp[(size_t)(j + U * c) * pairs + u * stride];
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            acc_mac_dev(a0, x[u].x, y[u].x);
            acc_mac_dev(a1, x[u].y, y[u].y);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) { x[u] = xn[u]; y[u] = yn[u]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc_mac_dev(a0, x[u].x, y[u].x);
          acc_mac_dev(a1, x[u].y, y[u].y);
        }
        j += U * c;
      }
    } else {
      for (; j + (U - 1) * c < k; j += U * c) {
        v2u64 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          x[u] = __builtin_nontemporal_load(cp + (size_t)j * pairs + u * stride);
          y[u] = sp[(size_t)j * pairs + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc_mac_dev(a0, x[u].x, y[u].x);
          acc_mac_dev(a1, x[u].y, y[u].y);
        }
      }
    }
    for (; j < k; j += c) {
      v2u64 x0 = cp[(size_t)j * pairs];
      v2u64 y0 = sp[(size_t)j * pairs];
      acc_mac_dev(a0, x0.x, y0.x); acc_mac_dev(a1, x0.y, y0.y);
    }
  }
  const u32 limb = active ? (2 * e) / ell : 0;
  const Mod m = mods[limb];
  v2u64 part;
  part.x = acc_reduce(a0, m);
  part.y = acc_reduce(a1, m);
  if (active) dl[threadIdx.x] = part;
  __syncthreads();
  if (active && g == 0) {
    v2u64 s = part;
    for (u32 w = 1; w < c; ++w) {
      v2u64 t = dl[w * chunk + el];
      s.x = addmod(s.x, t.x, m.q);
      s.y = addmod(s.y, t.y, m.q);
    }
    const size_t o = (size_t)d * pairs + e;
    v2u64 c2 = reinterpret_cast<const v2u64*>(c2col)[o];
    s.x = submod(s.x, c2.x, m.q);
    s.y = submod(s.y, c2.y, m.q);
    reinterpret_cast<v2u64*>(noisy)[o] = s;
  }
}

// decrypt_mac, dealer-grouped form: one workgroup serves DG dealers, so every s-hat pair fetched
// (through L2) is used DG times and the vector-memory instruction count per streamed byte drops
// from 2 to 1 + 1/DG.  Thread (g, e) as above; UJ j-steps are issued together.
// gridDim.y > 1 (both forms): the k terms are cut into gridDim.y ranges; a workgroup then leaves the partial sum of
// its range in partial[range][dealer] (no c2) and decrypt_finish_kernel adds the ranges up -- small batches
// (a single decrypt_party_value is ONE workgroup otherwise) then spread over the chip.
template <int DG, int UJ, int MAXT, bool NO_S = false>
__global__ __launch_bounds__(MAXT) void decrypt_mac_grouped_kernel(const u64* __restrict__ c1s,
                                                                    const u64* __restrict__ shat,
                                                                    const u64* __restrict__ c2col,
                                                                    u64* __restrict__ noisy,
                                                                    const Mod* __restrict__ mods, u32 k_all,
                                                                    u32 ell, u32 pairs, u32 c, u32 dealers, u64* __restrict__ partial) {
  extern __shared__ v2u64 dl[];
  const u32 kq = (k_all + gridDim.y - 1) / gridDim.y, jlo = blockIdx.y * kq;
  const u32 k = (jlo + kq) < k_all ? (jlo + kq) : k_all;       // this workgroup's terms: [jlo, k)
  const u32 d0 = blockIdx.x * DG;
  const u32 g = threadIdx.x / pairs, e = threadIdx.x % pairs;
  const bool active = threadIdx.x < c * pairs;
  const v2u64* sp = reinterpret_cast<const v2u64*>(shat) + e;
  const v2u64* cp[DG];
#pragma unroll
  for (int dd = 0; dd < DG; ++dd) {
    const u32 d = (d0 + dd) < dealers ? (d0 + dd) : (dealers - 1);   // clamp: tail group re-reads the last dealer
    cp[dd] = reinterpret_cast<const v2u64*>(c1s) + (size_t)d * k_all * pairs + e;
  }
  Acc a0[DG], a1[DG];
#pragma unroll
  for (int dd = 0; dd < DG; ++dd) { acc_zero(a0[dd]); acc_zero(a1[dd]); }
  if (active) {
    u32 j = jlo + g;
    for (; j + (UJ - 1) * c < k; j += UJ * c) {
      v2u64 y[UJ], x[UJ][DG];
#pragma unroll
      for (int u = 0; u < UJ; ++u) {
        if constexpr (NO_S) y[u] = (v2u64){(u64)j + 3, (u64)j + 5};    // timing experiment: no s-hat traffic
        else y[u] = sp[(size_t)(j + u * c) * pairs];
#pragma unroll
        for (int dd = 0; dd < DG; ++dd) x[u][dd] = __builtin_nontemporal_load(cp[dd] + (size_t)(j + u * c) * pairs);
      }
#pragma unroll
      for (int u = 0; u < UJ; ++u)
#pragma unroll
        for (int dd = 0; dd < DG; ++dd) {
          acc_mac_dev(a0[dd], x[u][dd].x, y[u].x);
          acc_mac_dev(a1[dd], x[u][dd].y, y[u].y);
        }
    }
    for (; j < k; j += c) {
      v2u64 y0 = sp[(size_t)j * pairs];
#pragma unroll
      for (int dd = 0; dd < DG; ++dd) {
        v2u64 x0 = cp[dd][(size_t)j * pairs];
        acc_mac_dev(a0[dd], x0.x, y0.x);
        acc_mac_dev(a1[dd], x0.y, y0.y);
      }
    }
  }
  const u32 limb = active ? (2 * e) / ell : 0;
  const Mod m = mods[limb];
#pragma unroll
  for (int dd = 0; dd < DG; ++dd) {
    v2u64 part;
    part.x = acc_reduce(a0[dd], m);
    part.y = acc_reduce(a1[dd], m);
    __syncthreads();
    if (active) dl[threadIdx.x] = part;
    __syncthreads();
    if (active && g == 0 && d0 + dd < dealers) {
      v2u64 sres = part;
      for (u32 w = 1; w < c; ++w) {
        v2u64 t = dl[w * pairs + e];
        sres.x = addmod(sres.x, t.x, m.q);
        sres.y = addmod(sres.y, t.y, m.q);
      }
      const size_t o = (size_t)(d0 + dd) * pairs + e;
      if (gridDim.y > 1) {
        reinterpret_cast<v2u64*>(partial)[(size_t)blockIdx.y * dealers * pairs + o] = sres;
      } else {
        v2u64 c2 = reinterpret_cast<const v2u64*>(c2col)[o];
        sres.x = submod(sres.x, c2.x, m.q);
        sres.y = submod(sres.y, c2.y, m.q);
        reinterpret_cast<v2u64*>(noisy)[o] = sres;
      }
    }
  }
}

// decrypt_mac, full-width form.  A wave-wide load moves at most 1 KiB and the chip sustains a fixed
// number of them per second, so a polynomial of L*l/2 = 64*FW + rem slot pairs is split into FW waves
// that own 64 pairs each (every load full width) plus ONE remainder wave whose lanes cover
// G = 64/rem' consecutive j at once (rem' = rem rounded up to a power of two): its loads are G
// segments of rem' pairs, again (nearly) full width.  The grouped form above leaves 15 % (L*l/2 = 272)
// to 47 % (68) of the lanes of its last wave idle on every load.  Two dealers per workgroup.
// (88 VGPRs is a budget, not an accident: 13 of these waves and the 8 waves of a decode workgroup share a CU when
// pvw_decrypt_batch_device overlaps the two; prefetching c2 at the top costs 12 registers, gains 1.5 % alone and
// loses 20 % overlapped.)
template <int DG, int UJ>
__global__ __launch_bounds__(1024) void decrypt_mac_fw_kernel(const u64* __restrict__ c1s,
                                                               const u64* __restrict__ shat,
                                                               const u64* __restrict__ c2col,
                                                               u64* __restrict__ noisy,
                                                               const Mod* __restrict__ mods, u32 k_all, u32 ell,
                                                               u32 pairs, u32 FW, u32 cfull, u32 remp, u32 crem,
                                                               u32 dealers, u64* __restrict__ partial) {
  extern __shared__ v2u64 dl[];                        // [DG][waves*64] partial sums
  const u32 kq = (k_all + gridDim.y - 1) / gridDim.y, jlo = blockIdx.y * kq;
  const u32 k = (jlo + kq) < k_all ? (jlo + kq) : k_all;       // this workgroup's terms: [jlo, k)
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const u32 nfull = cfull * FW, rem = pairs - FW * 64;
  const bool is_full = wave < nfull;                   // wave-uniform
  const u32 d0 = blockIdx.x * DG;
  u32 e, jstart, jstep;
  bool active;
  if (is_full) {
    const u32 g = wave / FW, fw = wave % FW;
    e = fw * 64 + lane;
    jstart = g;
    jstep = cfull;
    active = true;
  } else {
    const u32 rw = wave - nfull;                       // replica among the remainder waves
    const u32 G = 64 / remp, jsub = lane / remp, er = lane % remp;
    e = FW * 64 + er;
    jstart = rw * G + jsub;
    jstep = crem * G;
    active = er < rem;
  }
  const v2u64* sp = reinterpret_cast<const v2u64*>(shat) + (active ? e : 0);
  const v2u64* cp[DG];
#pragma unroll
  for (int dd = 0; dd < DG; ++dd) {
    const u32 d = (d0 + dd) < dealers ? (d0 + dd) : (dealers - 1);
    cp[dd] = reinterpret_cast<const v2u64*>(c1s) + (size_t)d * k_all * pairs + (active ? e : 0);
  }
  Acc a0[DG], a1[DG];
#pragma unroll
  for (int dd = 0; dd < DG; ++dd) { acc_zero(a0[dd]); acc_zero(a1[dd]); }
  if (active) {
    u32 j = jlo + jstart;
    for (; j + (UJ - 1) * jstep < k; j += UJ * jstep) {
      v2u64 y[UJ], x[UJ][DG];
#pragma unroll
      for (int u = 0; u < UJ; ++u) {
        y[u] = sp[(size_t)(j + u * jstep) * pairs];
#pragma unroll
        for (int dd = 0; dd < DG; ++dd) x[u][dd] = __builtin_nontemporal_load(cp[dd] + (size_t)(j + u * jstep) * pairs);
      }
#pragma unroll
      for (int u = 0; u < UJ; ++u)
#pragma unroll
        for (int dd = 0; dd < DG; ++dd) {
          acc_mac_dev(a0[dd], x[u][dd].x, y[u].x);
          acc_mac_dev(a1[dd], x[u][dd].y, y[u].y);
        }
    }
    for (; j < k; j += jstep) {
      v2u64 y0 = sp[(size_t)j * pairs];
#pragma unroll
      for (int dd = 0; dd < DG; ++dd) {
        v2u64 x0 = cp[dd][(size_t)j * pairs];
        acc_mac_dev(a0[dd], x0.x, y0.x);
        acc_mac_dev(a1[dd], x0.y, y0.y);
      }
    }
  }
  const u32 limb = active ? (2 * e) / ell : 0;
  const Mod m = mods[limb];
  const u32 T = blockDim.x;
#pragma unroll
  for (int dd = 0; dd < DG; ++dd) {
    v2u64 part;
    part.x = acc_reduce(a0[dd], m);
    part.y = acc_reduce(a1[dd], m);
    dl[dd * T + threadIdx.x] = part;
  }
  __syncthreads();
  // one owner per pair sums the partials of its replicas and finishes: full waves of replica 0 and
  // the lanes with jsub == 0 of remainder replica 0
  bool owner;
  if (is_full) owner = wave < FW;
  else owner = active && wave == nfull && lane < remp;
  if (owner) {
#pragma unroll
    for (int dd = 0; dd < DG; ++dd) {
      if (d0 + dd >= dealers) continue;
      v2u64 sres = (v2u64){0, 0};
      if (is_full) {
        for (u32 g = 0; g < cfull; ++g) {
          v2u64 tq = dl[dd * T + (g * FW + wave) * 64 + lane];
          sres.x = addmod(sres.x, tq.x, m.q);
          sres.y = addmod(sres.y, tq.y, m.q);
        }
      } else {
        const u32 G = 64 / remp;
        for (u32 rw = 0; rw < crem; ++rw)
          for (u32 js = 0; js < G; ++js) {
            v2u64 tq = dl[dd * T + (nfull + rw) * 64 + js * remp + lane];
            sres.x = addmod(sres.x, tq.x, m.q);
            sres.y = addmod(sres.y, tq.y, m.q);
          }
      }
      const size_t o = (size_t)(d0 + dd) * pairs + e;
      if (gridDim.y > 1) {
        reinterpret_cast<v2u64*>(partial)[(size_t)blockIdx.y * dealers * pairs + o] = sres;
      } else {
        v2u64 c2 = reinterpret_cast<const v2u64*>(c2col)[o];
        sres.x = submod(sres.x, c2.x, m.q);
        sres.y = submod(sres.y, c2.y, m.q);
        reinterpret_cast<v2u64*>(noisy)[o] = sres;
      }
    }
  }
}

// decrypt_finish: noisy[d] = INTT( sum_r partial[r][d] - c2col[d] )   (decryption.rs:268-274 and the
// change_representation(PowerBasis) of :116), one thread per (dealer, limb): the range sums of a split decrypt_mac
// are added up where the inverse transform reads them anyway
template <int ELL>
__global__ __launch_bounds__(64) void decrypt_finish_kernel(const u64* __restrict__ partial, u32 nsplit,
                                                             const u64* __restrict__ c2col, u64* __restrict__ noisy,
                                                             u32 dealers, u32 L, DevTables t) {
  const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= dealers * L) return;
  const u32 limb = tid % L;
  const Mod m = t.mods[limb];
  const size_t o = (size_t)tid * ELL, plane = (size_t)dealers * L * ELL;
  u64 a[ELL];
#pragma unroll
  for (int s = 0; s < ELL; s += 2) {
    const v2u64 c2 = *reinterpret_cast<const v2u64*>(c2col + o + s);
    v2u64 acc = *reinterpret_cast<const v2u64*>(partial + o + s);
    for (u32 r = 1; r < nsplit; ++r) {
      const v2u64 p = *reinterpret_cast<const v2u64*>(partial + r * plane + o + s);
      acc.x = addmod(acc.x, p.x, m.q);
      acc.y = addmod(acc.y, p.y, m.q);
    }
    a[s] = submod(acc.x, c2.x, m.q);
    a[s + 1] = submod(acc.y, c2.y, m.q);
  }
  ntt_inverse<ELL>(a, t.itw + (size_t)limb * ELL, t.itwp + (size_t)limb * ELL, t.linv[limb], t.linvp[limb], m);
#pragma unroll
  for (int s = 0; s < ELL; s += 2) *reinterpret_cast<v2u64*>(noisy + o + s) = (v2u64){a[s], a[s + 1]};
}

// ------------------------------------------------------------------------------------
// decode: decode_scalar_pvw_rns (decryption.rs:10-58) on the device, one thread per ciphertext.
// Big integers live in LDS with the thread index as the fast axis (word j of thread t at [j][t]).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void decode_kernel(const u64* __restrict__ noisy, u64* __restrict__ out,
                                                     u32 count, DecodeTables t) {
  extern __shared__ u64 dsm[];
  const u32 d = blockIdx.x * 64 + threadIdx.x;
  u64* base = dsm + threadIdx.x;
  BN x{base, 64};
  BN y{base + (size_t)(t.W + 1) * 64, 64};
  BN nres{base + (size_t)(2 * t.W + 1) * 64, 64};
  if (d >= count) return;
  out[d] = decode_one_fixed(t, noisy + (size_t)d * t.L * t.ell, x, y, nres);
}

// ------------------------------------------------------------------------------------
// decode, wave-cooperative form: ONE WAVE per ciphertext.  A big integer lives one 64-bit word
// per lane (word w in lane w), an RNS value one limb per lane; every step of pvw_decode.h's
// algorithm becomes "per-lane column sums + a short cross-lane carry loop":
//   lift      x = sum_i t_i * (Q/q_i) - kq*Q          L broadcast steps, columns of 3 words
//   to RNS    r_limb = sum_j x_j * 2^(64 j) mod q      W broadcast steps, lazy accumulator
//   divide    q^ = floor(N * floor(B^(W+1)/d) / B^(W+1)) from the top W+2 columns only, then
//             at most two corrections against the exact remainder (no digit-serial long division)
// Needs L <= 64 and W + 2 <= 64 (Q up to ~3900 bits); otherwise launch_decode uses decode_kernel.
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
// |x| (one word per lane) mod q_limb for every limb-lane, sign applied
__device__ __forceinline__ u64 wave_to_rns(const WaveDecodeCtx& c, u64 x, bool neg) {
  c.xs[c.lane] = x;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  u64 r = 0;
  if (c.limb_on) {
    Acc acc;
    acc_zero(acc);
    for (u32 j = 0; j < c.W; ++j) acc_mac_dev(acc, c.xs[j], c.powL[j * c.L + c.lane]);
    r = acc_reduce(acc, c.m);
    if (neg && r) r = c.m.q - r;
  }
  __builtin_amdgcn_wave_barrier();
  return r;
}
// floor(N / d) and N mod d via the reciprocal mu = floor(B^(W+1)/d); d given one word per lane (dw)
// and as a table (dtab, W+2 words); mu as a table of W+2 words.  N < B^W.  Returns quotient in q, remainder in r.
// dn = number of significant words of d (the q*d columns have only dn terms each)
__device__ __forceinline__ void wave_divmod(const WaveDecodeCtx& c, u64 n, const u64* mu, const u64* dtab, u64 dw,
                                            u64& q, u64& r, u32 dn = 64) {
  const u32 lane = c.lane, W = c.W;
  c.xs[lane] = n;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // top W+2 columns of N * mu: lane v holds column W-1+v
  u64 c0 = 0, c1 = 0, c2 = 0;
  if (lane < W + 2) {
    const u32 col = W - 1 + lane;
    for (u32 i = 0; i < W; ++i) {
      const u32 j = col - i;                       // index into mu
      if (j < W + 2) col_mac(c0, c1, c2, c.xs[i], mu[j]);
    }
  }
  u64 p = wave_normalize(c0, c1, c2, lane);        // lane v: word W-1+v of the (truncated) product
  // quotient estimate = words W+1 .. 2W  -> lanes 2 .. W+1, moved down to lanes 0 .. W-1
  {
    u32 lo = __shfl_down((u32)p, 2), hi = __shfl_down((u32)(p >> 32), 2);
    q = lane < W ? (((u64)hi << 32) | lo) : 0;
  }
  __builtin_amdgcn_wave_barrier();
  // remainder N - q*d (q <= true quotient, so this is >= 0)
  c.xs[lane] = q;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  c0 = c1 = c2 = 0;
  if (lane < W) {
    const u32 jn = (lane + 1) < dn ? (lane + 1) : dn;
    for (u32 j = 0; j < jn; ++j) col_mac(c0, c1, c2, c.xs[lane - j], dtab[j]);   // j indexes d
  }
  __builtin_amdgcn_wave_barrier();
  u64 qd = wave_normalize(c0, c1, c2, lane);
  qd = lane < W ? qd : 0;                          // the product fits W words (<= N)
  r = wave_sub(n, qd, lane);
  while (wave_cmp(r, dw) >= 0) {                   // at most two corrections
    r = wave_sub(r, dw, lane);
    u64 one = lane == 0 ? 1 : 0;
    q = wave_normalize(q + one, (q + one) < one ? 1 : 0, 0, lane);
  }
}

// wave_divmod without LDS round trips: N and d are read lane-to-lane (v_readlane / bpermute); only the
// reciprocal is a table, zero-padded to 2W+2 words so that the column loop has no bounds test.
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

__global__ __launch_bounds__(256) void decode_wave_kernel(const u64* __restrict__ noisy, u64* __restrict__ out,
                                                           u32 count, u32 stage_z, DecodeTables t) {
  // LDS: CRT table [L][W] | power table [W][L] | per wave: 64-word scratch (+ this ciphertext's residues)
  extern __shared__ u64 dws[];
  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32 W = t.W, L = t.L, l = t.ell;
  u64* qiL = dws;
  u64* powL = dws + (size_t)L * W;
  u64* smallL = dws + (size_t)2 * L * W;                 // mu_dp | dpow (padded) | mu_td | td, W+2 words each
  u64* wbase = smallL + (size_t)4 * (W + 2) + (size_t)wave * (64 + (stage_z ? L * l : 0));
  for (u32 x = threadIdx.x; x < L * W; x += 256) {
    qiL[x] = t.qi[x];
    powL[x] = t.pow64T[x];
  }
  for (u32 x = threadIdx.x; x < W + 2; x += 256) {
    smallL[x] = t.mu_dp[x];
    smallL[(W + 2) + x] = x < W ? t.dpow[x] : 0;
    smallL[2 * (W + 2) + x] = t.mu_td[x];
    smallL[3 * (W + 2) + x] = t.td[x];
  }
  const u32 d = blockIdx.x * 4 + wave;
  const bool live = d < count;                      // wave-uniform
  u64* zs = wbase + 64;
  if (live && stage_z)
    for (u32 x = lane; x < L * l; x += 64) zs[x] = noisy[(size_t)d * L * l + x];
  __syncthreads();
  if (!live) return;
  WaveDecodeCtx c{t, qiL, powL, wbase, lane, W, L, t.mods[lane < L ? lane : 0], lane < L, lane < W,
                  lane < W ? t.Q[lane] : 0, lane < W ? t.halfQ[lane] : 0};
  // this limb's l residues
  const u64* z = (stage_z ? zs : noisy + (size_t)d * L * l) + (size_t)(c.limb_on ? lane : 0) * l;
  const u64 dm = t.dmod[c.limb_on ? lane : 0], dmp = t.dmodp[c.limb_on ? lane : 0];
  const u64 q = c.m.q;
  auto tmp = [&](u32 i) -> u64 { return submod(mulmod_shoup(z[i], dm, dmp, q), z[i + 1], q); };   // :19-27
  // Horner over tmp_0 .. tmp_{l-2} (:30-33)
  u64 h = tmp(0);
  for (u32 i = 1; i + 1 < l; ++i) h = addmod(mulmod_shoup(h, dm, dmp, q), tmp(i), q);
  bool neg;
  u64 x = wave_lift_centered(c, h, neg);
  // reduce_modulo_poly (:154-178)
  const u64 dpw = lane < W ? t.dpow[lane] : 0, hdw = lane < W ? t.half_dpow[lane] : 0;
  u64 qq, r;
  wave_divmod(c, x, smallL, smallL + (W + 2), dpw, qq, r);
  if (__ballot(r != 0) == 0) neg = false;
  if (wave_cmp(r, hdw) > 0) {
    r = wave_sub(dpw, r, lane);
    neg = !neg;
  }
  u64 nres = wave_to_rns(c, r, neg);
  // noise[i] = round((noise[i+1] - tmp[i]) / Delta), i = l-2 .. 0   (:44-48, :180-207)
  const u64 tdw = lane < W ? t.td[lane] : 0, dlw = lane < W ? t.delta[lane] : 0;
  for (u32 i = l - 1; i-- > 0;) {
    bool pneg;
    u64 p = wave_lift_centered(c, submod(nres, tmp(i), q), pneg);
    // 2|p| + Delta
    u64 hi = p >> 63, lo2 = p << 1;
    u64 s = lo2 + dlw;
    u64 num = wave_normalize(s, hi + (s < dlw), 0, lane);
    wave_divmod(c, num, smallL + 2 * (W + 2), smallL + 3 * (W + 2), tdw, qq, r);
    const bool qzero = __ballot(qq != 0) == 0;
    nres = wave_to_rns(c, qq, pneg && !qzero);
  }
  // plaintext = -z_0 - noise_0 (:51-53), then extract_constant_term_as_u64 (:226-247)
  bool vneg;
  u64 z0 = z[0];
  u64 v = wave_lift_centered(c, submod(z0 ? q - z0 : 0, nres, q), vneg);
  const bool vzero = __ballot(v != 0) == 0;
  u64 result;
  if (vneg && !vzero) {
    const bool hiw = __ballot(lane > 0 && v != 0) != 0;
    const u64 v0 = ((u64)__shfl((u32)(v >> 32), 0) << 32) | (u32)__shfl((u32)v, 0);
    if (!hiw && v0 <= 1000) {                       // small negative -> 0 (:233-235)
      if (lane == 0) out[d] = 0;
      return;
    }
    v = wave_sub(c.Qw, v, lane);                    // (v + Q) % Q = Q - |v|
  }
  const bool hiw2 = __ballot(lane > 0 && v != 0) != 0;
  result = hiw2 ? 0 : v;
  if (lane == 0) out[d] = result;
}

// decode, lifted-chain form (default): WPC waves per ciphertext, CPW ciphertexts per workgroup.
// The reference's chain noise_i = round((noise_{i+1} - tmp_i) / Delta) (decryption.rs:44-48) is exact
// integer arithmetic mod Q, so it can be carried in big-integer form throughout: the l+1 CRT lifts it
// needs (tmp_0..tmp_{l-2}, the Horner value, z_0) do not depend on the chain and are spread over the
// WPC waves; after one barrier wave 0 walks the chain with one short-divisor division per step and NO
// conversion back to RNS.  Serial big steps per ciphertext: l divisions (instead of l+1 lifts +
// l divisions + l RNS conversions in decode_wave_kernel).
template <int WPC>
__global__ __launch_bounds__(512) void decode_chain_kernel(const u64* __restrict__ noisy, u64* __restrict__ out,
                                                            u32 count, u32 cpw_dbg, DecodeTables t) {
  const u32 cpw = cpw_dbg & 0xffff;
  const u32 dbg = PVW_TUNING ? (cpw_dbg >> 16) : 0;        // tuning build, dbg != 0: timing experiment, out[] = cycle counts
  const u64 tk0 = dbg ? clock64() : 0;
  // LDS: CRT table [L][W] | two reciprocals, 2W+2 words each | per ciphertext: lifts [l+1][64] + signs, residues [L][l]
  //      | per wave: 64-word scratch
  extern __shared__ u64 dws[];
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
  const u32 d = blockIdx.x * cpw + cw;
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

// ------------------------------------------------------------------------------------
// i8 MFMA operand-map probe (self-test): C[32][32] = A[32][32] * B[32][32] with ONE
// v_mfma_i32_32x32x32_i8, operands fetched with the lane maps the digit-GEMM kernels rely on:
//   A fragment of lane l (r = l & 31, h = l >> 5): A[r][16h .. 16h+15]     (16 consecutive K)
//   B fragment:                                   B[16h .. 16h+15][r]
//   C/D register g of lane l:                      C[(g & 3) + 8 (g >> 2) + 4 h][r]
// ------------------------------------------------------------------------------------
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(64) void mfma_i8_probe_kernel(const signed char* __restrict__ A,
                                                            const signed char* __restrict__ B, int* __restrict__ Cm) {
  const u32 l = threadIdx.x, r = l & 31, h = l >> 5;
  union { v4i32 v; signed char b[16]; } fa, fb;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    fa.b[j] = A[r * 32 + 16 * h + j];
    fb.b[j] = B[(16 * h + j) * 32 + r];
  }
  v16i32 acc = {};
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa.v, fb.v, acc, 0, 0, 0);
#pragma unroll
  for (int g = 0; g < 16; ++g) Cm[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = acc[g];
}

// ====================================================================================
// Digit GEMM on the matrix cores: NV (up to 16) vectors against one pass over the matrix.
//
// For many vectors the k-term inner products are a genuine GEMM per (limb, slot),
//     out[row][v] = sum_j X[row][j] * Y[j][v]   (mod q),
// and the integer VALU (4 v_mad_u64_u32 per MAC, ~1.6e12 MAC/s) is the bound.  On the matrix
// cores a 64x64-bit modular product is folded into i8 MFMAs like this:
//     x * y = sum_a x_a 2^(8a) * y  ==  sum_a x_a * y^(a)   (mod q),   y^(a) = 2^(8a) y mod q
// so the sum over the 8 bytes x_a of x joins the contraction index: K = (j, a).  The A operand is
// then the RAW little-endian u64 data (16 bytes per lane = 2 consecutive j of one row) and each
// vector element contributes 8 shifted copies, written as 8 balanced base-256 digits
// y^(a) = sum_b d_b 2^(8b), d_b in [-128, 127]:  B[(j,a)][(v,b)] = d_b(y_v^(a)[j]).
// One v_mfma_i32_32x32x32_i8 = 32 rows x 4 vectors x 4 j = 512 modular MACs, and only 8 partial
// sums per output remain to be recombined: out = sum_b C[row][(v,b)] 2^(8b) mod q.
// The raw bytes are unsigned: they are offset by -128 (xor 0x80) and 128 * colsum(B) is added back.
// Used by multi-dealer encrypt (>= 8 dealers) and batched key generation; the single-vector
// encrypt stays on mac_rows (a GEMV: HBM-bound, no matrix-core shape).
// ====================================================================================

// tiled matrix (or API-layout rows) -> MFMA-tiled copy XM[limb][slot][rt][jb][h*32+m][2]
template <int ELL>
__global__ __launch_bounds__(256) void mftile_kernel(const u64* __restrict__ src, u32 src_is_tiled,
                                                      u64* __restrict__ XM, u32 rows, u32 k, u32 L) {
  constexpr int R = 128 / ELL;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (size_t)rows * k * L) return;
  const u32 limb = tid % L;
  const u32 j = (tid / L) % k;
  const u32 row = tid / ((size_t)L * k);
  const u64* p = src_is_tiled ? src + (((size_t)(row / R) * L + limb) * k + j) * 128 + (row % R) * ELL
                              : src + tid * ELL;     // API layout [row][j][limb][slot]
  const u32 RT = ((rows + PVW_GEMM_ROWS_PER_WG - 1) / PVW_GEMM_ROWS_PER_WG) * (PVW_GEMM_ROWS_PER_WG / 32), JB = (k + 3) / 4;   // row tiles padded to whole workgroups
  const u32 rt = row >> 5, m = row & 31, jb = j >> 2, h = (j >> 1) & 1, e = j & 1;
#pragma unroll
  for (int sl = 0; sl < ELL; ++sl) {
    const size_t tile = (((size_t)limb * ELL + sl) * RT + rt) * JB + jb;
    XM[tile * 128 + (h * 32 + m) * 2 + e] = p[sl] ^ 0x8080808080808080ULL;   // bytes stored signed-offset (x - 128): the i8 MFMA operand as is
  }
}

// the same from API-layout rows [row][j][limb][slot], through LDS: one block = (limb, row tile of 32, JBG j-blocks);
// 64-byte runs in (the l slots of one (row, j)), whole 1-KiB tiles out; padding rows / j are written as the
// offset-zero byte pattern, so XM needs no clearing beforehand.  (mftile_kernel scatters 8-byte words: 1.9 TB/s.)
template <int ELL>
__global__ __launch_bounds__(256) void mftile_rows_kernel(const u64* __restrict__ src, u64* __restrict__ XM, u32 rows, u32 k, u32 L) {
  constexpr int JBG = ELL <= 8 ? 4 : (ELL == 16 ? 2 : 1);       // j-blocks (of 4 j) per block; LDS stays at 33 KB
  constexpr int PLANE = JBG * 128 + 2;                          // u64 per slot plane (+2: bank spread)
  __shared__ u64 lt[ELL * PLANE];
  const u32 JB = (k + 3) / 4, JG = (JB + JBG - 1) / JBG;
  const u32 RT = ((rows + PVW_GEMM_ROWS_PER_WG - 1) / PVW_GEMM_ROWS_PER_WG) * (PVW_GEMM_ROWS_PER_WG / 32);
  const u32 jg = blockIdx.x % JG, rt = (blockIdx.x / JG) % RT, limb = blockIdx.x / (JG * RT);
  const u32 row0 = rt * 32, j0 = jg * JBG * 4;
  const size_t P = (size_t)L * ELL;
  // in: pieces of 16 bytes (two slots): piece q = ((row * NJ + jj) * (ELL / 2) + sp)
  constexpr int NJ = 4 * JBG, NPIECE = 32 * NJ * (ELL / 2);
  for (int q = threadIdx.x; q < NPIECE; q += 256) {
    const u32 sp = q % (ELL / 2), jj = (q / (ELL / 2)) % NJ, row = q / ((ELL / 2) * NJ);
    v2u64 v = (v2u64){0, 0};
    if (row0 + row < rows && j0 + jj < k)
      v = *reinterpret_cast<const v2u64*>(src + ((size_t)(row0 + row) * k + (j0 + jj)) * P + (size_t)limb * ELL + 2 * sp);
    const u32 jbl = jj >> 2, h = (jj >> 1) & 1, e = jj & 1;
    const u32 w = (jbl * 64 + h * 32 + row) * 2 + e;
    lt[(2 * sp) * PLANE + w] = v.x ^ 0x8080808080808080ULL;      // bytes stored signed-offset, as mftile_kernel does
    lt[(2 * sp + 1) * PLANE + w] = v.y ^ 0x8080808080808080ULL;
  }
  __syncthreads();
  // out: ELL * JBG tiles of 64 lanes x 16 bytes
  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (u32 tq = wave; tq < ELL * JBG; tq += 4) {
    const u32 slot = tq / JBG, jbl = tq % JBG;
    if (jg * JBG + jbl >= JB) continue;
    const v2u64 v = (v2u64){lt[slot * PLANE + (jbl * 64 + lane) * 2], lt[slot * PLANE + (jbl * 64 + lane) * 2 + 1]};
    const size_t tile = (((size_t)limb * ELL + slot) * RT + rt) * JB + (jg * JBG + jbl);
    *reinterpret_cast<v2u64*>(XM + tile * 128 + lane * 2) = v;
  }
}

// small coefficients [row][j][l] (secret keys: secret_key.rs:98-112) -> reduced, transformed and written straight
// into the MFMA-tiled raw operand XM -- the prologue + mftile_rows pair of key generation in one pass, without the
// API-layout rows in between.  One block = (32 rows, NJ = 4 JBG consecutive j, every gridDim.z-th limb): a thread
// keeps ONE polynomial's coefficients in registers and per limb transforms them into an LDS image of the l x JBG
// tiles (two images in turn: one barrier per limb), which leave as whole 1-KiB tiles.  Padding rows / j are written
// as the offset-zero byte pattern, as mftile_rows_kernel does.
template <int ELL>
__global__ __launch_bounds__(256) void shat_mftile_kernel(const i64* __restrict__ coeffs, u64* __restrict__ XM, u32 rows, u32 k, u32 L, DevTables t) {
  constexpr int JBG = 2, NJ = 4 * JBG;                          // 32 rows x 8 j = 256 polynomials: one per thread
  constexpr int PLANE = JBG * 128 + 2;                          // u64 per slot plane of the tile image (+2: bank spread)
  __shared__ u64 lt[2][ELL * PLANE];
  const u32 JB = (k + 3) / 4, jg = blockIdx.x, rt = blockIdx.y, RT = gridDim.y;
  const u32 row = threadIdx.x & 31, jj = threadIdx.x >> 5;       // consecutive lanes: consecutive rows of one j
  const u32 grow = rt * 32 + row, gj = jg * NJ + jj;
  i64 c[ELL];
  if (grow < rows && gj < k) {
    const v2u64* src = reinterpret_cast<const v2u64*>(coeffs + ((size_t)grow * k + gj) * ELL);
#pragma unroll
    for (int sl = 0; sl < ELL; sl += 2) {
      const v2u64 v = src[sl / 2];
      c[sl] = (i64)v.x;
      c[sl + 1] = (i64)v.y;
    }
  } else {
#pragma unroll
    for (int sl = 0; sl < ELL; ++sl) c[sl] = 0;
  }
  const u32 w = ((jj >> 2) * 64 + ((jj >> 1) & 1) * 32 + row) * 2 + (jj & 1);
  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  u32 buf = 0;
  for (u32 limb = blockIdx.z; limb < L; limb += gridDim.z, buf ^= 1) {
    const Mod m = t.mods[limb];
    u64 a[ELL];
#pragma unroll
    for (int sl = 0; sl < ELL; ++sl) a[sl] = signed_residue(c[sl], m);
    ntt_forward<ELL>(a, t.tw + (size_t)limb * ELL, t.twp + (size_t)limb * ELL, m);
#pragma unroll
    for (int sl = 0; sl < ELL; ++sl) lt[buf][sl * PLANE + w] = a[sl] ^ 0x8080808080808080ULL;   // bytes stored signed-offset, as mftile_kernel does
    __syncthreads();       // image `buf` is complete.  It is next written two limbs on, by waves that have passed the
                           // barrier in between, which every wave reaches only after the reads below
    for (u32 tq = wave; tq < ELL * JBG; tq += 4) {
      const u32 slot = tq / JBG, jbl = tq % JBG;
      if (jg * JBG + jbl >= JB) continue;
      const v2u64 v = (v2u64){lt[buf][slot * PLANE + (jbl * 64 + lane) * 2], lt[buf][slot * PLANE + (jbl * 64 + lane) * 2 + 1]};
      const size_t tile = (((size_t)limb * ELL + slot) * RT + rt) * JB + (jg * JBG + jbl);
      *reinterpret_cast<v2u64*>(XM + tile * 128 + lane * 2) = v;
    }
  }
}

// vector elements -> digit tiles YD[vg][limb][slot][jb][h*32+col][16] and the offset correction SY.
// One wave per (v, limb, slot); lane = j-block: each lane turns 4 consecutive j into the 8 shifted
// copies y*2^(8a) mod q and writes their balanced digits as 16 16-byte runs.
// The raw matrix bytes enter the MFMA offset by -128 (signed), so every output lacks 128 * sum_{j,a,b} d_b 2^(8b)
// = 128 * sum_{j,a} (2^(8a) y_j mod q): a constant per (vector, limb, slot), because the balanced digits represent
// each shifted copy exactly.  It is accumulated here from the copies themselves (wave-reduced, no atomics), reduced
// mod q and left in SY as one u64 per vector (record of four per vector group); gemm_finish adds it.
// 4x4 byte transpose of four dwords (y_i byte j = x_j byte i) with v_perm_b32
__device__ __forceinline__ void transpose4x4_bytes(u32 x0, u32 x1, u32 x2, u32 x3, u32& y0, u32& y1, u32& y2, u32& y3) {
  const u32 t0 = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t1 = __builtin_amdgcn_perm(x1, x0, 0x07030602u);
  const u32 t2 = __builtin_amdgcn_perm(x3, x2, 0x05010400u), t3 = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
  y0 = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
  y1 = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
  y2 = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
  y3 = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}
// digit GEMM with biased accumulators (gemm_recombine_biased): the geometries it is exact for
constexpr u32 PVW_GEMM_BIASED_MAX_K = 512;               // |half| <= 8 k 2^14 (2^24 + 2^16 + 2^8 + 1) < 2^51
__host__ __device__ inline bool gemm_biased(u32 k) { return k <= PVW_GEMM_BIASED_MAX_K; }
template <int ELL, bool STAGE>
__global__ __launch_bounds__(64) void vec_digits_kernel(const u64* __restrict__ vhat, size_t vstride,
                                                         signed char* __restrict__ YD, int* __restrict__ SY,
                                                         u32 nv, u32 k, u32 L, DevTables t, size_t lstride, size_t jstride) {
  const u32 lane = threadIdx.x;
  const u32 slot = blockIdx.x % ELL;
  const u32 limb = (blockIdx.x / ELL) % L;
  const u32 v = blockIdx.x / (ELL * L);
  const Mod m = t.mods[limb];
  const u32 JB = (k + 3) / 4;
  const u32 vg = v >> 2, v4 = v & 3;
  const u64 w256p = (m.ratio_hi << 8) | (m.ratio_lo >> 56);   // floor(256 * 2^64 / q) = floor(2^128 / q) >> 56
  const u64* y = vhat + (size_t)v * vstride + (size_t)limb * lstride + slot;
  signed char* tiles = YD + (((size_t)vg * L + limb) * ELL + slot) * (size_t)JB * 1024;
  u64 csum_lo = 0;                                                // sum of the shifted copies (each < 2^62), 96 bits
  u32 csum_hi = 0;
  __shared__ v4i32 st[STAGE ? 32 * 16 : 1];                     // [tile of this round][piece], 8 KiB
  const u32 jb_end = STAGE ? ((JB + 63) & ~63u) : JB;           // STAGE: whole passes, every lane takes part in the staging
  for (u32 jb = lane; jb < jb_end; jb += 64) {
    // digit[b][kappa], kappa = 8*jj + a  (32 bytes per digit column b = four u64, one per jj)
    union { u64 d[8][4]; v4i32 q[8][2]; } dg;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const u32 j = 4 * jb + jj;
      u64 cur = j < k ? y[(size_t)j * jstride] : 0;
      // the 8 balanced base-256 digits of w < 2^62 are the bytes of (w + 0x80..80) with their top bits
      // flipped: adding 128 to every byte position propagates exactly the carries of "digit > 127"
      const u64 C = 0x8080808080808080ULL;
      u32 rl[8], rh[8];                                          // rows a: digits b = 0..3 | 4..7
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const u64 dgt = (cur + C) ^ C;
        rl[a] = (u32)dgt;
        rh[a] = (u32)(dgt >> 32);
        csum_lo += cur;
        csum_hi += csum_lo < cur;
        cur = mulmod_shoup(cur, 256, w256p, m.q);
      }
      // 8x8 byte transpose: column b gets the bytes a = 0..7
      u32 cl[8], ch[8];
      transpose4x4_bytes(rl[0], rl[1], rl[2], rl[3], cl[0], cl[1], cl[2], cl[3]);
      transpose4x4_bytes(rl[4], rl[5], rl[6], rl[7], ch[0], ch[1], ch[2], ch[3]);
      transpose4x4_bytes(rh[0], rh[1], rh[2], rh[3], cl[4], cl[5], cl[6], cl[7]);
      transpose4x4_bytes(rh[4], rh[5], rh[6], rh[7], ch[4], ch[5], ch[6], ch[7]);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        dg.d[b][jj] = ((u64)ch[b] << 32) | cl[b];
      }
    }
    if constexpr (STAGE) {
      // this lane's 16 runs of 16 bytes (piece = 8 h + b) go through LDS so that every global store
      // instruction writes whole 128-byte lines (lanes 8x..8x+7 = the 8 digit columns of one (tile, h)).
      // The piece index is XORed with the tile index so that neither side has bank conflicts.
      // two rounds of 32 tiles each keep the staging buffer at 8 KiB per wave (20 waves per CU instead of 10)
      const u32 jb0 = jb - lane;                                   // first tile of this pass
#pragma unroll
      for (u32 rnd = 0; rnd < 2; ++rnd) {
        __builtin_amdgcn_wave_barrier();
        if ((lane >> 5) == rnd) {
          const u32 tq = lane & 31;
#pragma unroll
          for (int b = 0; b < 8; ++b) {
            st[tq * 16 + ((0 * 8 + b) ^ (tq & 15))] = dg.q[b][0];   // h = 0: kappa 0..15
            st[tq * 16 + ((1 * 8 + b) ^ (tq & 15))] = dg.q[b][1];   // h = 1: kappa 16..31
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (u32 it = 0; it < 8; ++it) {
          const u32 tl = it * 4 + (lane >> 4), piece = lane & 15;   // tile within the round
          const v4i32 val = st[tl * 16 + (piece ^ (tl & 15))];
          const u32 hh = piece >> 3, bb = piece & 7;
          const u32 tg = jb0 + rnd * 32 + tl;
          if (tg < JB)
            *reinterpret_cast<v4i32*>(tiles + (size_t)tg * 1024 + (size_t)(hh * 32 + v4 * 8 + bb) * 16) = val;
        }
      }
    } else {
      signed char* tile = tiles + (size_t)jb * 1024;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        *reinterpret_cast<v4i32*>(tile + (size_t)(0 * 32 + v4 * 8 + b) * 16) = dg.q[b][0];   // h = 0: kappa 0..15
        *reinterpret_cast<v4i32*>(tile + (size_t)(1 * 32 + v4 * 8 + b) * 16) = dg.q[b][1];   // h = 1: kappa 16..31
      }
    }
  }
  // wave sum of the 96-bit lane sums, then 128 * sum mod q
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const u64 olo = ((u64)__shfl_xor((u32)(csum_lo >> 32), d) << 32) | __shfl_xor((u32)csum_lo, d);
    const u32 ohi = __shfl_xor(csum_hi, d);
    csum_lo += olo;
    csum_hi += ohi + (csum_lo < olo);
  }
  if (lane == 0) {
    const u64 r = reduce128(csum_lo, (u64)csum_hi, m);
    u64 corr = mulmod(r, 128, m);
    if (gemm_biased(k)) {                                    // minus the constant the biased accumulators leave: 2^51 + 2^83
      const u64 c51 = (1ull << 51) % m.q;
      const u64 cb = addmod(c51, mulmod(c51, (1ull << 32) % m.q, m), m.q);
      corr = corr >= cb ? corr - cb : corr + m.q - cb;
    }
    reinterpret_cast<u64*>(SY + (((size_t)vg * L + limb) * ELL + slot) * 32)[v4] = corr;
  }
}

// One 32x32 accumulator of the digit GEMM (digit tile as first operand: register 4 v4 + bb of lane (h, rr) holds
// digit b = 4 h + bb of vector v4 for matrix row rr) -> the two finished sums this lane owns,
//   res[pr] = sum_b C[(v, b)][row] 2^(8b) mod q   for v = pr + 2 h  (pr = 0, 1).
// Four registers give a 52-bit half-sum per vector by shifts and adds inside the lane, the lower half of the wave
// ending up with the LOW halves (digits 0-3) of four vectors and the upper half with their HIGH halves (digits 4-7).
// One exchange between the halves (v_permlane32_swap: upper half of one register <-> lower half of another) pairs
// them up: the lower lanes finish vectors 0 and 1, the upper lanes vectors 2 and 3.
//
// BIASED (k <= PVW_GEMM_BIASED_MAX_K): the accumulators start at gemm_acc_init(), 2^27 in every digit-3 register, so a
// lane's half-sum comes out as half + 2^51 in (0, 2^52) -- no signs -- and is formed exactly in f64 (4 conversions and
// 3 fused multiply-adds; the integer form needs 13 two-dword shift/add instructions).  The finished sum carries the
// constant 2^51 + 2^83, which vec_digits_kernel has taken out of the offset correction that gemm_finish adds.
// FASTQ (every q wider than 54 bits): (lo' + hi' 2^32) mod q through a quotient estimate in f64 that is never above
// the true quotient and at most one below it (inv32 = 2^32 / q (1 - 2^-40)), so x - qhat q lies in [0, 2q), plus
// lo' < 2^52 in [0, 3q): two conditional subtractions.
__device__ __forceinline__ v16i32 gemm_acc_init(bool biased) {
  const int b = biased ? (1 << 27) : 0;
  return (v16i32){0, 0, 0, b, 0, 0, 0, b, 0, 0, 0, b, 0, 0, 0, b};
}
__device__ __forceinline__ double gemm_inv32(const Mod& m, bool biased) {
  const double inv = 4294967296.0 / (double)m.q;
  return biased ? inv * (1.0 - 0x1p-40) : inv;
}
template <bool FASTQ>
__device__ __forceinline__ void gemm_recombine_biased(const v16i32& a, const Mod& m, double inv32, u64 (&res)[2]) {
  double half[4];
#pragma unroll
  for (int v4 = 0; v4 < 4; ++v4)
    half[v4] = __builtin_fma((double)a[4 * v4 + 3], 0x1p24,
                             __builtin_fma((double)a[4 * v4 + 2], 0x1p16, __builtin_fma((double)a[4 * v4 + 1], 0x1p8, (double)a[4 * v4])));
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    const u64 P = (u64)__double_as_longlong(half[pr]), Q = (u64)__double_as_longlong(half[pr + 2]);
    const auto slo = __builtin_amdgcn_permlane32_swap((u32)P, (u32)Q, false, false);
    const auto shi = __builtin_amdgcn_permlane32_swap((u32)(P >> 32), (u32)(Q >> 32), false, false);
    const double lo4 = __longlong_as_double((long long)(((u64)shi[0] << 32) | slo[0]));   // low half-sum + 2^51
    const double hi4 = __longlong_as_double((long long)(((u64)shi[1] << 32) | slo[1]));   // high half-sum + 2^51
    // an integer below 2^52 plus 2^52 has that integer as its mantissa
    const u64 lo_i = (u64)__double_as_longlong(lo4 + 0x1p52) & 0x000fffffffffffffull;
    const u64 hi_b = (u64)__double_as_longlong(hi4 + 0x1p52);
    if constexpr (FASTQ) {
      const u32 qhat = (u32)(hi4 * inv32);                                     // < 2^30
      u64 s = ((u64)(u32)hi_b << 32) - (u64)qhat * m.q + lo_i;                 // mod 2^64; the true value is in [0, 3q)
      if (s >= 2 * m.q) s -= 2 * m.q;
      if (s >= m.q) s -= m.q;
      res[pr] = s;
    } else {
      const unsigned __int128 tot = (unsigned __int128)lo_i + ((unsigned __int128)(hi_b & 0x000fffffffffffffull) << 32);
      res[pr] = reduce128((u64)tot, (u64)(tot >> 64), m);
    }
  }
}
template <bool FASTQ>
__device__ __forceinline__ void gemm_recombine(const v16i32& a, const Mod& m, double inv32, u64 (&res)[2]) {
  long long half[4];                                     // this lane's half-sum of the four vectors, |.| < 2^51
#pragma unroll
  for (int v4 = 0; v4 < 4; ++v4)
    half[v4] = (long long)a[4 * v4] + ((long long)a[4 * v4 + 1] << 8) + ((long long)a[4 * v4 + 2] << 16) +
               ((long long)a[4 * v4 + 3] << 24);
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    // P = half[pr] (lower lanes keep their low part of vector pr, upper lanes give up their high part of it),
    // Q = half[pr + 2] (lower lanes give up their low part of vector pr + 2, upper lanes keep their high part):
    // after the swap every lane reads (low, high) = (P, Q) of the vector it finishes
    const u64 P = (u64)half[pr], Q = (u64)half[pr + 2];
    const auto slo = __builtin_amdgcn_permlane32_swap((u32)P, (u32)Q, false, false);
    const auto shi = __builtin_amdgcn_permlane32_swap((u32)(P >> 32), (u32)(Q >> 32), false, false);
    const long long lo4 = (long long)(((u64)shi[0] << 32) | slo[0]);
    const long long hi4 = (long long)(((u64)shi[1] << 32) | slo[1]);
    if constexpr (FASTQ) {
      // (lo4 + hi4 * 2^32) mod q for q > 2^53: |lo4| < q already; hi4 * 2^32 through a quotient estimated
      // in f64 (|hi4| < 2^52 is exact, the estimate is off by at most one) and two corrections
      const u64 ah = (u64)(hi4 < 0 ? -hi4 : hi4), al = (u64)(lo4 < 0 ? -lo4 : lo4);
      const u64 qhat = (u64)((double)ah * inv32);
      long long rem = (long long)((ah << 32) - qhat * m.q);
      if (rem < 0) rem += (long long)m.q;
      if (rem >= (long long)m.q) rem -= (long long)m.q;
      u64 rh = (u64)rem;
      if (hi4 < 0 && rh) rh = m.q - rh;
      const u64 rl = (lo4 < 0 && al) ? m.q - al : al;
      res[pr] = addmod(rh, rl, m.q);
    } else {
      const __int128 tot = (__int128)lo4 + ((__int128)hi4 << 32);
      u64 lo = (u64)tot, hi = (u64)((unsigned __int128)tot >> 64);
      const bool neg = (long long)hi < 0;
      if (neg) { lo = ~lo + 1; hi = ~hi + (lo == 0); }
      u64 r = reduce128(lo, hi, m);
      if (neg && r) r = m.q - r;
      res[pr] = r;
    }
  }
}

#if !PVW_TUNING
#undef PVW_GEMM_ABLATE
#endif
#ifndef PVW_GEMM_ABLATE
#define PVW_GEMM_ABLATE 0
#endif
#define PVW_ABL(bit) ((PVW_GEMM_ABLATE & (bit)) != 0)
template <int ELL, int NVG, int RPW, int NCH = 0, bool FASTQ = false>
__global__ __launch_bounds__(256) void gemm_digits_kernel(GemmSection sa, GemmSection sb, const signed char* __restrict__ YD,
                                                           const int* __restrict__ SY, const Mod* __restrict__ mods,
                                                           u32 k, u32 L, u32 nv_total, u32 nv_pad, u32 dbg_arg, u32 vbn,
                                                           size_t yd_b16, size_t sy_b16) {
  const u32 dbg = PVW_TUNING ? dbg_arg : 0;                // the shipped build has no timing branches
  // dbg (tuning build: PVW_GEMM_DEBUG, timing experiments only, results wrong): 1 = no K loop, 2 = no epilogue;
  // compile-time ablations -DPVW_GEMM_ABLATE=bits: 8 = no A loads, 16 = no B loads, 32 = no MFMA, 64 = no epilogue (wide form)
  // block = (limb, slot, group of 4*RPW row tiles); the 4 waves share the vector-digit tiles through
  // LDS (CJ j-blocks at a time); each wave owns RPW row tiles of 32 rows and streams their raw u64 tiles.
  constexpr int CJ = 8;                                    // j-blocks per staged chunk (32 MFMAs per wave per barrier)
  constexpr int BSH = NVG * CJ * 64 / 256;                 // 16-byte B elements each thread stages per chunk
  __shared__ v4i32 bl[2][NVG * CJ * 64];                   // two chunks of NVG*CJ KiB
  // NCH != 0: the launcher guarantees k == 4 * NCH * CJ, so every bounds test below folds away
  const u32 JB = NCH ? (u32)(NCH * CJ) : ((dbg & 1) ? 0 : (k + 3) / 4);
  const bool biased = gemm_biased(k);                      // uniform: see gemm_recombine_biased
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const u32 rtg_total = sa.rt_groups + sb.rt_groups;
  // XCD-aware order: blocks b and b+8 share an XCD (and its L2), so give each XCD a contiguous run of
  // block ids -- workgroups that share the vector-digit tiles of one (limb, slot) then hit in L2
  u32 bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  // batches of 16 vectors are the fastest-varying part of the block id: the vbn workgroups that stream the
  // same matrix tiles sit next to each other on one XCD and share them through its L2
  const u32 vb = bid % vbn;
  bid /= vbn;
  const u32 nv = (nv_total - 16 * vb) < 16 ? (nv_total - 16 * vb) : 16;
  YD += vb * yd_b16;
  SY += vb * sy_b16;
  const u32 ls = bid / rtg_total, rtg = bid % rtg_total;
  const u32 limb = ls / ELL, slot = ls % ELL;
  const bool in_a = rtg < sa.rt_groups;
  const GemmSection& sec = in_a ? sa : sb;
  const u32 rt0 = ((in_a ? rtg : rtg - sa.rt_groups) * 4 + wave) * RPW;
  const u32 RT = sec.rt_groups * 4 * RPW;
  const u32 rows_pad = sec.rt_groups * PVW_GEMM_ROWS_PER_WG;
  const v4i32* ap[RPW];
  bool live[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    ap[r] = reinterpret_cast<const v4i32*>(sec.XM) + ((((size_t)limb * ELL + slot) * RT + rt0 + r) * JB) * 64 + lane;
    live[r] = ((rt0 + r) * 32) < sec.nrows;              // wave-uniform
  }
  const v4i32* ybase = reinterpret_cast<const v4i32*>(YD);
  v16i32 acc[RPW][NVG];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int g = 0; g < NVG; ++g) acc[r][g] = gemm_acc_init(biased);
  const v4i32 zero4 = (v4i32){0, 0, 0, 0};
  // software pipeline over chunks of CJ j-blocks: the A tiles and this thread's share of the B tiles
  // of chunk c+1 are in flight (registers) while chunk c is multiplied out of LDS
  // Every load below is UNCONDITIONAL (clamped index, select afterwards): a load inside a branch makes hipcc
  // fall back to s_waitcnt vmcnt(0) at the loop header, which drains the whole prefetch pipeline on every
  // chunk.  Rows past the section's end read the zeroed padding of XM (their results are never stored) and
  // tiles past JB re-read the last tile against zero B digits.
  const u32 jlast = JB ? JB - 1 : 0;
  auto fetch_a = [&](u32 jc, v4i32 (&an)[RPW][CJ]) {
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int u = 0; u < CJ; ++u) {
        const u32 j = (jc + u) < JB ? (jc + u) : jlast;
        an[r][u] = PVW_ABL(8) ? zero4 : __builtin_nontemporal_load(ap[r] + (size_t)j * 64);
      }
  };
  auto fetch_b = [&](u32 jc, v4i32 (&bn)[BSH]) {
#pragma unroll
    for (int x = 0; x < BSH; ++x) {
      const u32 e = threadIdx.x + 256 * x;                  // element of the [NVG][CJ][64] chunk
      const u32 g = e / (CJ * 64), rem = e % (CJ * 64), u = rem / 64;
      const bool in = (jc + u) < JB;
      const u32 jb = in ? (jc + u) : jlast;
      const v4i32 val = PVW_ABL(16) ? zero4 : ybase[((((size_t)g * L + limb) * ELL + slot) * JB + jb) * 64 + (rem & 63)];
      bn[x] = in ? val : zero4;
    }
  };
  // A tiles are prefetched TWO chunks ahead through three register sets used in rotation, B one chunk ahead
  // (bn -> LDS).  NCH != 0: the chunk loop is fully unrolled (JB == NCH * CJ), so the rotation is plain
  // renaming and hipcc can count the outstanding loads exactly; around a loop back-edge it falls back to
  // s_waitcnt vmcnt(0), which cuts the lead to one chunk (the NCH == 0 form, kept for other k).
  v4i32 aset[3][RPW][CJ], bn[BSH];
  fetch_b(0, bn);
  fetch_a(0, aset[0]);
  fetch_a(CJ, aset[1]);
#pragma unroll
  for (int x = 0; x < BSH; ++x) bl[0][threadIdx.x + 256 * x] = bn[x];
  __syncthreads();
  // one chunk: issue the loads for later chunks, multiply chunk jc out of `ac` and bl[cur], stage B of chunk jc+CJ
  auto step = [&](u32 jc, u32 cur, v4i32 (&ac)[RPW][CJ], v4i32 (&aload)[RPW][CJ], bool load_a) {
    fetch_b(jc + CJ, bn);                                   // past the end: clamped re-reads (cache hits), unused
    if (load_a) fetch_a(jc + 2 * CJ, aload);
    // B fragments of step u+1 are read from LDS while the MFMAs of step u run (two register sets); the
    // sched_barriers keep hipcc from sinking each read next to its use, which exposes the LDS latency
    // before every other MFMA
    v4i32 bf[2][NVG];
#pragma unroll
    for (int g = 0; g < NVG; ++g) bf[0][g] = bl[cur][g * (CJ * 64) + lane];
#pragma unroll
    for (int u = 0; u < CJ; ++u) {
      if (u + 1 < CJ) {
#pragma unroll
        for (int g = 0; g < NVG; ++g) bf[(u + 1) & 1][g] = bl[cur][g * (CJ * 64) + (u + 1) * 64 + lane];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const v4i32 ax = ac[r][u];                          // XM holds the bytes already offset by -128
#pragma unroll
        for (int g = 0; g < NVG; ++g) {
          // the DIGIT tile is the first operand and the raw tile the second: the product comes out transposed,
          // C[(v, b)][row], so a lane holds (for one matrix row) four digits b = 4h .. 4h+3 of four vectors in
          // consecutive registers and the recombination below needs ONE exchange between the wave's halves
          if (PVW_ABL(32)) acc[r][g][0] += ax[0] ^ bf[u & 1][g][0];   // timing experiment: no MFMA
          else acc[r][g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[u & 1][g], ax, acc[r][g], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the other buffer was last read in the previous step, which every wave left through the barrier below
#pragma unroll
    for (int x = 0; x < BSH; ++x) bl[cur ^ 1][threadIdx.x + 256 * x] = bn[x];
    __syncthreads();
  };
  if constexpr (NCH != 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) step(c * CJ, c & 1, aset[c % 3], aset[(c + 2) % 3], c + 2 < NCH);
  } else {
    // rotation by register moves (each move waits for the loads it copies: one chunk of lead)
    u32 cur = 0;
    for (u32 jc = 0; jc < JB; jc += CJ) {
      step(jc, cur, aset[0], aset[2], true);
#pragma unroll
      for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int u = 0; u < CJ; ++u) { aset[0][r][u] = aset[1][r][u]; aset[1][r][u] = aset[2][r][u]; }
      cur ^= 1;
    }
  }
  // recombine (gemm_recombine): out[row][v] = sum_b C[(v, b)][row] 2^(8b) mod q; the offset correction is added by gemm_finish
  const Mod m = mods[limb];
  const u32 h = lane >> 5, rr = lane & 31;
  const double inv32 = gemm_inv32(m, biased);                // FASTQ: every modulus is wider than 54 bits
  if (dbg & 2) {
    int keep = 0;
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int g = 0; g < NVG; ++g)
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) keep ^= acc[r][g][tt];
    if (keep == 0x7fffffff && k == 0xffffffffu) sec.tmp[0] = 1;
    return;
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    if (!live[r]) continue;
    const u32 row = (rt0 + r) * 32 + rr;
#pragma unroll
    for (int g = 0; g < NVG; ++g) {
      u64 res[2];
      if (biased) gemm_recombine_biased<FASTQ>(acc[r][g], m, inv32, res);
      else gemm_recombine<FASTQ>(acc[r][g], m, inv32, res);
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const u32 v = g * 4 + pr + 2 * h;
        // intermediate [limb][slot][v][row]: the 32 lanes of a half write 32 consecutive rows of one vector
        if (row < sec.nrows && v < nv)
          sec.tmp[vb * sec.tmp_bstride + (((size_t)limb * ELL + slot) * nv_pad + v) * rows_pad + row] = res[pr];
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Digit GEMM, wide form (more than 16 vectors): a workgroup of 8 waves computes 256 rows x 32 vectors of one
// (limb, slot) -- wave (wr, wv) owns two row tiles and one batch of 16 vectors (8 accumulators) -- and BOTH operands
// go through LDS: a raw tile is used by the two waves of its row pair, a digit tile by the four waves of its batch.
// Per MFMA that is 256 bytes through L2 / L1 instead of the 512 of gemm_digits_kernel (128 rows x 16 vectors, raw
// tiles straight to registers), and the ablations of round 2 (profiles/r02_gemm_ablations.txt) say the loads, not the
// matrix pipe, set that kernel's time: no MFMA at all saves 8 % of it, no loads 36 %.
// Staging is LDS-DMA (global_load_lds_dwordx4: one 1-KiB tile per wave-instruction, lane-linear in both memories --
// XM and YD are stored as the MFMA fragments lie), four 32-KiB buffers of 2 j-blocks each; two stages stay in
// flight across every barrier (counted s_waitcnt vmcnt + raw s_barrier: __syncthreads() would drain them):
//     wait for my DMAs of stage s+1 | barrier | issue the DMAs of stage s+3 | 16 MFMAs per wave on stage s, the
//     fragment reads of the next j-block (of stage s or s+1) issued ahead of each group of 8
// The epilogue is gemm_digits_kernel's (gemm_recombine, intermediate [limb][slot][v][row], gemm_finish).
// Needs k % 16 == 0 (whole stages); the launcher falls back to gemm_digits_kernel otherwise and for <= 16 vectors.
// ------------------------------------------------------------------------------------
template <int ELL, bool FASTQ, int WRN, bool PP = false>
__global__ __launch_bounds__(128 * WRN, 2) void gemm_digits_wide_kernel(GemmSection sa, GemmSection sb, const signed char* __restrict__ YD,
                                                                         const Mod* __restrict__ mods, u32 k, u32 L, u32 nv_total,
                                                                         u32 nv_pad, u32 vbn, size_t yd_b16) {
  static_assert(PVW_GEMM_RPW == 1, "XM is padded to groups of four row tiles");
  static_assert(WRN == 2 || WRN == 4, "4 or 8 waves");
  constexpr int NWV = 2 * WRN;                             // waves: WRN along the rows x 2 batches of 16 vectors
  constexpr int CJ = 2;                                    // j-blocks per stage
  constexpr int NB = WRN == 4 ? 4 : 3;                     // stage buffers (8 waves: 128 KiB, one workgroup per CU; 4 waves: 72 KiB, two)
  constexpr int AHEAD = NB - 2;                            // stages still in flight when a barrier is passed
  constexpr bool CROSS = WRN == 4;                         // the barrier of iteration s certifies stage s + 1 (else stage s)
  constexpr int RTW = 2 * WRN, NG = 8;                     // row tiles / vector groups (of 4) per workgroup
  constexpr int STAGE = (RTW + NG) * CJ * 64;              // 16-byte elements per stage: 32 / 24 KiB
  constexpr int GPS = (RTW + NG) * CJ / NWV;               // LDS-DMA instructions per wave per stage: 4 / 6
  __shared__ v4i32 stage[NB * STAGE];                      // ONE array (a second __shared__ object next to LDS-DMA
                                                           // destinations makes hipcc drain the DMAs early)
  const u32 JB = k / 4, NST = JB / CJ;
  const bool biased = gemm_biased(k);                      // uniform: see gemm_recombine_biased
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const u32 wr = wave >> 1, wv = wave & 1;
  const u32 RTa = sa.rt_groups * 4, RTb = sb.rt_groups * 4;                  // row tiles of the two sections (padded)
  const u32 ga = (RTa + RTW - 1) / RTW, gb = (RTb + RTW - 1) / RTW;          // workgroups along the rows
  const u32 vbpn = (vbn + 1) / 2;                                            // pairs of 16-vector batches
  u32 bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous (see gemm_digits_kernel)
  const u32 vbp = bid % vbpn;
  bid /= vbpn;
  const u32 ls = bid / (ga + gb), rg = bid % (ga + gb);
  const u32 limb = ls / ELL, slot = ls % ELL;
  const bool in_a = rg < ga;
  const GemmSection& sec = in_a ? sa : sb;
  const u32 RT = in_a ? RTa : RTb;
  const u32 rtbase = (in_a ? rg : rg - ga) * RTW;
  const u32 rows_pad = sec.rt_groups * PVW_GEMM_ROWS_PER_WG;
  // ---- what this wave stages: tiles GPS wave .. GPS wave + GPS - 1 of a stage's (RTW raw + NG digit tiles) x CJ ----
  const v4i32* src[GPS];
  {
    const v4i32* xm = reinterpret_cast<const v4i32*>(sec.XM);
    const v4i32* yd = reinterpret_cast<const v4i32*>(YD);
#pragma unroll
    for (int x = 0; x < GPS; ++x) {
      const u32 t = wave * GPS + x, jb_i = t % CJ;
      u32 rt = rtbase + t / CJ;
      rt = rt < RT ? rt : RT - 1;                            // past the section: re-read its last tile (never stored)
      const v4i32* sraw = xm + ((((size_t)limb * ELL + slot) * RT + rt) * JB + jb_i) * 64 + lane;
      const u32 g_i = (t >= RTW * CJ ? t - RTW * CJ : 0) / CJ;   // 0..7: batch (g_i >> 2) of the pair, group (g_i & 3)
      u32 vbq = 2 * vbp + (g_i >> 2);
      vbq = vbq < vbn ? vbq : vbn - 1;
      const v4i32* sdig = yd + (vbq * yd_b16) / 16 + (((((size_t)(g_i & 3)) * L + limb) * ELL + slot) * JB + jb_i) * 64 + lane;
      src[x] = t < RTW * CJ ? sraw : sdig;
    }
  }
  auto issue = [&](u32 st) {
    const u32 b = st % NB;
    if (PVW_ABL(24)) return;                                 // timing experiment: nothing staged
#pragma unroll
    for (int x = 0; x < GPS; ++x)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[x] + (size_t)st * CJ * 64),
                                       (__attribute__((address_space(3))) void*)&stage[b * STAGE + (wave * GPS + x) * 64], 16, 0, 0);
  };
  v16i32 acc[2][4];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[r][g] = gemm_acc_init(biased);
  // Fragment reads and MFMAs must overlap (per stage a wave reads 12 KiB for its 16 MFMAs): two fragment sets, the
  // reads of j-block i+1 issued under the MFMAs of j-block i.  The 8-wave form (all of a SIMD's waves in one
  // workgroup, in step) carries that across the stage boundary: the barrier of iteration s certifies stage s + 1.
  // The 4-wave form leaves the first read of a stage exposed -- the other workgroup on the CU fills the gap.
  v4i32 f0[6], f1[6];                                        // [0..1] raw tiles of the two row tiles, [2..5] digit tiles
  auto read = [&](v4i32 (&f)[6], u32 st, int jb_i) {
    const v4i32* raw = &stage[(st % NB) * STAGE + (2 * wr) * CJ * 64 + lane];
    const v4i32* dig = &stage[(st % NB) * STAGE + (RTW + 4 * wv) * CJ * 64 + lane];
#pragma unroll
    for (int r = 0; r < 2; ++r) f[r] = raw[(r * CJ + jb_i) * 64];
#pragma unroll
    for (int g = 0; g < 4; ++g) f[2 + g] = dig[(g * CJ + jb_i) * 64];
  };
  // 8 MFMAs on one fragment set.  Everything else a wave has to issue rides in the gaps between them, a little per
  // gap (MI355X_MICROARCH.md, LDS: two ds_read_b128 per gap are free, a third per wave saturates the LDS array and the
  // MFMA behind it waits for the ISSUE of the reads; an LDS-DMA instruction costs 25-60 cycles in a quiet gap, 100+
  // next to a burst of reads): the six fragment reads into the OTHER set behind MFMAs 1-3 (not earlier: hipcc puts a
  // full lgkmcnt(0) before a set's first use), this wave's DMAs for stage st + NB - 1 behind MFMAs 4 onwards.
  auto issue_one = [&](u32 st, int x) {
    if (PVW_ABL(24)) return;                                 // timing experiment: nothing staged
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[x] + (size_t)st * CJ * 64),
                                     (__attribute__((address_space(3))) void*)&stage[(st % NB) * STAGE + (wave * GPS + x) * 64], 16, 0, 0);
  };
  auto mac_and_read = [&](const v4i32 (&f)[6], v4i32 (&fn)[6], u32 st_n, int jb_n, bool rd, u32 st_dma, bool dma, int x0) {
    const v4i32* raw = &stage[(st_n % NB) * STAGE + (2 * wr) * CJ * 64 + lane];
    const v4i32* dig = &stage[(st_n % NB) * STAGE + (RTW + 4 * wv) * CJ * 64 + lane];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = i >> 2, g = i & 3;
      if (PVW_ABL(32)) acc[r][g][0] += f[2 + g][0] ^ f[r][0];   // timing experiment: no MFMA
      else acc[r][g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[2 + g], f[r], acc[r][g], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (rd && i == 0 && !PVW_ABL(256)) { fn[0] = raw[(0 * CJ + jb_n) * 64]; fn[1] = raw[(1 * CJ + jb_n) * 64]; }
      if (rd && i == 1 && !PVW_ABL(256)) { fn[2] = dig[(0 * CJ + jb_n) * 64]; fn[3] = dig[(1 * CJ + jb_n) * 64]; }
      if (rd && i == 2 && !PVW_ABL(256)) { fn[4] = dig[(2 * CJ + jb_n) * 64]; fn[5] = dig[(3 * CJ + jb_n) * 64]; }
      if (dma && i >= 3 && i - 3 < GPS / 2) issue_one(st_dma, x0 + i - 3);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // s_waitcnt vmcnt(n stages x GPS): the immediate must be a literal
  auto wait_stages = [&](u32 n) {
    if (n >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GPS) : "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(GPS % 2 == 0 && GPS / 2 <= 5 && CJ == 2 && AHEAD <= 2, "two j-blocks per stage; wait_stages knows 0, 1 and 2 stages");
#pragma unroll
  for (int i = 0; i < NB - 1; ++i)
    if ((u32)i < NST) issue(i);
  if constexpr (PP) {
    // Ping-pong (8 waves): the waves of a SIMD, w and w + 4, run half a stage apart -- while one issues its 16 MFMAs
    // the other reads its 12 fragments and sits out the waits, so the matrix pipe is not left idle
    // by instructions that cost issue time.  Phases are separated by barriers every wave executes; group B (waves
    // 4-7) starts one barrier late and group A takes one extra at the end.
    //   A:  L0 | M0 | L1 | M1 | ...        L_s: fragments of stage s -> registers, lgkmcnt(0)
    //   B:     | L0 | M0 | L1 | ...        M_s: 16 MFMAs, the wave's DMAs of stage s + 3 in their gaps
    // Stage s is read first by A's L_s; every wave waits for its own DMAs of stage s in the phase before that
    // (A: end of M_{s-1}, B: end of L_{s-1}; two younger stages stay in flight).  The buffer of stage s - 1 is
    // refilled (DMAs of stage s + 3) only in M_s, after the barrier that follows B's L_{s-1}, whose lgkmcnt(0)
    // retired the last reads of it.
    static_assert(!PP || (WRN == 4 && NB == 4), "ping-pong: the 8-wave form");
    const bool grp_b = wave >= 4;
    wait_stages(NST - 1 < 2u ? NST - 1 : 2u);                // stage 0 has landed (mine; the barrier: everyone's)
    __builtin_amdgcn_s_barrier();
    if (grp_b) __builtin_amdgcn_s_barrier();
    for (u32 st = 0; st < NST; ++st) {
      // stages that may stay in flight while a wave waits for its DMAs of stage st + 1: B waits at the end of L_st
      // (newest issued: st + 2), A at the end of M_st (newest: st + 3)
      const u32 last = NST - 1;
      const u32 ahead_b = (st + 2 < last ? st + 2 : last) > st + 1 ? (st + 2 < last ? st + 2 : last) - (st + 1) : 0;
      const u32 ahead_a = (st + 3 < last ? st + 3 : last) > st + 1 ? (st + 3 < last ? st + 3 : last) - (st + 1) : 0;
      const bool dma = st + 3 < NST;
      read(f0, st, 0);
      read(f1, st, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (grp_b && st + 1 < NST) wait_stages(ahead_b);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = (i >> 2) & 1, g = i & 3;
        const v4i32(&f)[6] = i < 8 ? f0 : f1;
        if (PVW_ABL(32)) acc[r][g][0] += f[2 + g][0] ^ f[r][0];
        else acc[r][g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[2 + g], f[r], acc[r][g], 0, 0, 0);
        // this wave's four DMAs of stage st + 3 ride in the gaps of its own MFMAs (a DMA costs ~60 cycles of issue
        // against the 32 of the MFMA in front of it; in the L phase it would lengthen the phase the partner waits on)
        if ((i & 3) == 3 && dma) {
          __builtin_amdgcn_sched_barrier(0);
          issue_one(st + 3, i >> 2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (!grp_b && st + 1 < NST) wait_stages(ahead_a);
      __builtin_amdgcn_s_barrier();
    }
    if (!grp_b) __builtin_amdgcn_s_barrier();
  } else {
  if constexpr (CROSS) {
    wait_stages(NST - 1 < (u32)(NB - 2) ? NST - 1 : (u32)(NB - 2));    // stage 0 has landed (mine; the barrier: everyone's)
    __builtin_amdgcn_s_barrier();
    read(f0, 0, 0);
  }
  for (u32 st = 0; st < NST; ++st) {
    // CROSS: my DMAs of stage st + 1 (else: of stage st) have landed once only the AHEAD younger stages' are outstanding;
    // the barrier extends that to every wave's and says that every wave has consumed stage st - 1
    const u32 newest = st + NB - 2 < NST - 1 ? st + NB - 2 : NST - 1;   // newest stage issued so far
    const u32 need = CROSS ? st + 1 : st;
    wait_stages(newest > need ? newest - need : 0);
    if (!PVW_ABL(128)) __builtin_amdgcn_s_barrier();          // (128, 256: timing experiments -- no barrier, no fragment reads)
    const bool dma = st + NB - 1 < NST;                       // into the buffer stage st - 1 has just left
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!CROSS) read(f0, st, 0);
    mac_and_read(f0, f1, st, 1, true, st + NB - 1, dma, 0);
    mac_and_read(f1, f0, st + 1 < NST ? st + 1 : st, 0, CROSS, st + NB - 1, dma, GPS / 2);   // past the end: a harmless re-read
  }
  }
  if (PVW_ABL(64)) {                                         // timing experiment: no epilogue
    int keep = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) keep ^= acc[r][g][tt];
    if (keep == 0x7fffffff && k == 0xffffffffu) sec.tmp[0] = 1;
    return;
  }
  // ---- epilogue ----
  const Mod m = mods[limb];
  const u32 h = lane >> 5, rr = lane & 31;
  const double inv32 = gemm_inv32(m, biased);
  const u32 vb = 2 * vbp + wv;
  if (vb >= vbn) return;
  const u32 nv = (nv_total - 16 * vb) < 16 ? (nv_total - 16 * vb) : 16;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const u32 rt = rtbase + 2 * wr + r;
    if (rt >= RT || rt * 32 >= sec.nrows) continue;          // wave-uniform
    const u32 row = rt * 32 + rr;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u64 res[2];
      if (biased) gemm_recombine_biased<FASTQ>(acc[r][g], m, inv32, res);
      else gemm_recombine<FASTQ>(acc[r][g], m, inv32, res);
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const u32 v = g * 4 + pr + 2 * h;
        if (row < sec.nrows && v < nv)
          sec.tmp[vb * sec.tmp_bstride + (((size_t)limb * ELL + slot) * nv_pad + v) * rows_pad + row] = res[pr];
      }
    }
  }
}

// gemm_finish: intermediate [limb][slot][v][row] -> API layout out[v][row][limb][slot] (+ addend),
// 256-byte runs in, 8*l-byte runs out, through LDS tiles of 32 rows x l slots; one block takes VPB = 4
// vectors so that four tiles' worth of loads are in flight per thread.
#define PVW_FINISH_VPB (ELL >= 64 ? 2 : 4)
template <int ELL>
__global__ __launch_bounds__(256) void gemm_finish_kernel(GemmSection sec, const Mod* __restrict__ mods, u32 L,
                                                           u32 nv, u32 nv_pad, u32 rows_pad, size_t ostride,
                                                           const int* __restrict__ SY, size_t sy_b16) {
  constexpr int VPB = PVW_FINISH_VPB, PT = 32 * ELL / 256 ? 32 * ELL / 256 : 1;   // elements per thread per tile
  static_assert(sizeof(u64) * VPB * ELL * 33 <= 48 * 1024 || ELL > 32, "finish tiles");
  __shared__ u64 tile[VPB][ELL][33];
  const u32 rb = blockIdx.x, v0 = blockIdx.y * VPB, limb = blockIdx.z;
  const u32 row0 = rb * 32;
  u64 in[VPB][PT];
#pragma unroll
  for (int vi = 0; vi < VPB; ++vi) {
    const u32 v = (v0 + vi) < nv ? (v0 + vi) : (nv - 1);
    const u64* tp = sec.tmp + (v >> 4) * sec.tmp_bstride + (((size_t)limb * ELL) * nv_pad + (nv_pad == 16 ? (v & 15) : v)) * rows_pad + row0;
#pragma unroll
    for (int x = 0; x < PT; ++x) {
      const u32 e = threadIdx.x + 256 * x;
      const u32 row = e & 31, slot = (e >> 5) % ELL;
      in[vi][x] = tp[(size_t)slot * nv_pad * rows_pad + row];
    }
  }
#pragma unroll
  for (int vi = 0; vi < VPB; ++vi)
#pragma unroll
    for (int x = 0; x < PT; ++x) {
      const u32 e = threadIdx.x + 256 * x;
      if (e < 32 * ELL) tile[vi][(e >> 5) % ELL][e & 31] = in[vi][x];
    }
  __syncthreads();
  const u64 q = mods[limb].q;
  u64 add[VPB][PT];
  // offset correction of vector v at (limb, slot): SY record of its group of four (vec_digits_kernel)
  auto corr_of = [&](u32 v, u32 slot) -> u64 {
    return reinterpret_cast<const u64*>(SY + (size_t)(v >> 4) * sy_b16 + ((((size_t)((v & 15) >> 2)) * L + limb) * ELL + slot) * 32)[v & 3];
  };
  const bool has_add = sec.addend != nullptr;
  const size_t rstride = sec.row_stride ? sec.row_stride : (size_t)L * ELL;
#pragma unroll
  for (int vi = 0; vi < VPB; ++vi)
#pragma unroll
    for (int x = 0; x < PT; ++x) {
      const u32 e = threadIdx.x + 256 * x;
      const u32 slot = e % ELL, row = (e / ELL) & 31;
      const u32 v = (v0 + vi) < nv ? (v0 + vi) : (nv - 1);
      const u32 rr = (row0 + row) < sec.nrows ? (row0 + row) : 0;
      const size_t o = (size_t)v * ostride + (size_t)rr * rstride + (size_t)limb * ELL + slot;
      add[vi][x] = addmod(has_add ? sec.addend[o] : 0, corr_of(v, slot), q);
    }
#pragma unroll
  for (int vi = 0; vi < VPB; ++vi)
#pragma unroll
    for (int x = 0; x < PT; ++x) {
      const u32 e = threadIdx.x + 256 * x;
      const u32 slot = e % ELL, row = e / ELL;
      if (e < 32 * ELL && row0 + row < sec.nrows && v0 + vi < nv) {
        const u64 val = addmod(tile[vi][slot][row], add[vi][x], q);
        if (sec.tiled_out) {
          // M[row_block][limb][j][rho][slot] with the party as the matrix row and the GEMM row as j
          constexpr u32 R = 128 / ELL;
          const u32 prow = sec.tiled_row0 + (sec.tiled_swap ? row0 + row : v0 + vi);     // party
          const u32 pcol = sec.tiled_swap ? v0 + vi : row0 + row;                        // column of B
          const u32 ncol = sec.tiled_swap ? nv : sec.nrows;
          sec.tiled_out[(((size_t)(prow / R) * L + limb) * ncol + pcol) * 128 + (prow % R) * ELL + slot] = val;
        } else {
          const size_t o = (size_t)(v0 + vi) * ostride + (size_t)(row0 + row) * rstride + (size_t)limb * ELL + slot;
          sec.out[o] = val;
        }
      }
    }
}

// gemm_finish with the error term made on the spot instead of read as an addend:
//   key generation   b_p[col] = (s_p A)[col] + e_p[col]                    (public_key.rs:128-147), into the tiled B-hat;
//   multi-dealer c2  c2_d[i]  = (B r_d)[i] + e2_d[i] + m_{d,i} g-hat        (encryption.rs:177-200), into the API planes.
// The error polynomial of (GEMM row, vector) is drawn here -- the ChaCha stream and rejection sampler of the prologue,
// uniform in [-bound, bound], key and stream index from GemmErrSource -- or read as explicit small coefficients; no
// transformed error rows in memory, no prologue work for them.  One THREAD = (row, vector): it makes the small
// coefficients once, then for each of its limbs (a contiguous range per blockIdx.z) transforms them, adds the
// intermediate's l values (for a fixed slot the 32 rows of a half-wave are 256 contiguous bytes), the offset
// correction and the encoded scalar.  The finished l slots (8 l bytes per thread) go through the wave's own LDS rows
// so that l / 2 neighbouring lanes write ONE row's 8 l contiguous bytes per store instead of 16 bytes each of l / 2
// rows (API layout: rows are 8 L l bytes apart).  No block-wide barriers.
template <int ELL>
__global__ __launch_bounds__(256) void gemm_finish_err_kernel(GemmSection sec, DevTables t, u32 L, u32 nv, u32 nv_pad, u32 rows_pad,
                                                               size_t ostride, const int* __restrict__ SY, size_t sy_b16, GemmErrSource es,
                                                               u32 v_lo, u32 v_hi) {
  constexpr int VPB = 8;                                         // vectors per block (x 32 rows)
  constexpr int G = ELL / 2;                                     // lanes that share out one another's 8 l bytes (16 each)
  constexpr int CST = ELL + 2;                                   // LDS words per thread (16-byte aligned, bank spread)
  __shared__ i64 coef[256 * CST];
  const u32 tid = threadIdx.x, lane = tid & 63;
  const u32 row_raw = blockIdx.x * 32 + (tid & 31), v_raw = v_lo + blockIdx.y * VPB + (tid >> 5);   // this launch: vectors [v_lo, v_hi)
  // every lane stays: a lane past the end still carries 16-byte pieces of its neighbours' rows to memory
  const bool v_ok = v_raw < v_hi;
  const u32 row = row_raw < sec.nrows ? row_raw : sec.nrows - 1, v = v_ok ? v_raw : v_hi - 1;
  i64 c[ELL];
  {
    i64* o = coef + tid * CST;
    if (es.explicit_coeffs) {
      const i64* ec = es.explicit_coeffs + ((size_t)row * es.coef_row + (size_t)v * es.coef_v) * ELL;
#pragma unroll
      for (int sl = 0; sl < ELL; sl += 2) *reinterpret_cast<v2u64*>(o + sl) = *reinterpret_cast<const v2u64*>(ec + sl);
    } else {
      ChaChaRng g;
      g.init(es.key[(v - v_lo) * es.key_v], es.domain, es.index0 + row * es.index_row + v * es.index_v);
      auto emit = [o](u32 sl, i64 val) { o[sl] = val; };       // (dynamic index: through LDS, then into registers)
      sample_uniform_poly(g, ELL, es.bound, emit);
    }
#pragma unroll
    for (int sl = 0; sl < ELL; ++sl) c[sl] = o[sl];             // own row of the array: no barrier needed
  }
  const i64 scalar = es.scalars ? (i64)es.scalars[(size_t)v * es.scalar_v + row] : 0;   // `as i64` wrap, encryption.rs:195
  const u64* tbase = sec.tmp + (v >> 4) * sec.tmp_bstride + (size_t)(nv_pad == 16 ? (v & 15) : v) * rows_pad + row;
  const size_t sstride = (size_t)nv_pad * rows_pad;             // words between consecutive slots of the intermediate
  const u64* cbase = reinterpret_cast<const u64*>(SY + (size_t)(v >> 4) * sy_b16 + ((size_t)((v & 15) >> 2) * L * ELL) * 32) + (v & 3);
  constexpr u32 R = 128 / ELL;
  const size_t rstride = sec.row_stride ? sec.row_stride : (size_t)L * ELL;
  u64 in[ELL];
  auto load_tmp = [&](u32 limb) {
#pragma unroll
    for (int sl = 0; sl < ELL; ++sl) in[sl] = __builtin_nontemporal_load(tbase + ((size_t)limb * ELL + sl) * sstride);
  };
  // this block's limbs: a contiguous range (a thread's stores then fill neighbouring lines one after the other)
  const u32 per = (L + gridDim.z - 1) / gridDim.z, limb_end = (blockIdx.z + 1) * per < L ? (blockIdx.z + 1) * per : L;
  u32 limb = blockIdx.z * per;
  if (limb < limb_end) load_tmp(limb);
  // the wave's staging rows (the coefficient rows, free once c[] is loaded): lane x's finished l slots at stg + x * CST
  u64* stg = reinterpret_cast<u64*>(coef) + (size_t)(tid - lane) * CST;
  const u32 gi = lane & (G - 1), gb = lane & ~(u32)(G - 1);      // piece this lane carries; first lane of its group
  for (; limb < limb_end; ++limb) {
    const Mod m = t.mods[limb];
    u64 a[ELL], x[ELL];
#pragma unroll
    for (int sl = 0; sl < ELL; ++sl) { a[sl] = signed_residue(c[sl], m); x[sl] = in[sl]; }
    if (limb + 1 < limb_end) load_tmp(limb + 1);                 // the next limb's intermediate arrives under this transform
    ntt_forward<ELL>(a, t.tw + (size_t)limb * ELL, t.twp + (size_t)limb * ELL, m);
    if (es.scalars) {                                            // encode_scalar (parameters.rs:346-367): + m g-hat
      const u64 mr = signed_residue(scalar, m);
      const u64* g = t.ghat + (size_t)limb * ELL;
      const u64* gp = t.ghatp + (size_t)limb * ELL;
#pragma unroll
      for (int sl = 0; sl < ELL; ++sl) a[sl] = addmod(a[sl], mulmod_shoup(mr, g[sl], gp[sl], m.q), m.q);
    }
#pragma unroll
    for (int sl = 0; sl < ELL; sl += 2) {
      const u64 c0 = cbase[((size_t)limb * ELL + sl) * 16], c1 = cbase[((size_t)limb * ELL + sl + 1) * 16];   // SY records are 32 ints
      *reinterpret_cast<v2u64*>(stg + lane * CST + sl) =
          (v2u64){addmod(addmod(x[sl], a[sl], m.q), c0, m.q), addmod(addmod(x[sl + 1], a[sl + 1], m.q), c1, m.q)};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // store j: the G lanes of a group write the 8 l contiguous bytes of the group's j-th row, 16 bytes each
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const v2u64 pv = *reinterpret_cast<const v2u64*>(stg + (gb + j) * CST + 2 * gi);
      const u32 rj = row_raw - gi + j;                           // the row lane gb + j works on (before clamping)
      if (rj < sec.nrows && v_ok) {
        const u32 prow = sec.tiled_row0 + rj;                    // tiled form: the party; the column of B is the GEMM vector
        u64* o = sec.tiled_out ? sec.tiled_out + (((size_t)(prow / R) * L + limb) * nv + v) * 128 + (prow % R) * ELL
                               : sec.out + (size_t)v * ostride + (size_t)rj * rstride + (size_t)limb * ELL;
        *reinterpret_cast<v2u64*>(o + 2 * gi) = pv;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                             // the staging rows are rewritten for the next limb
  }
}

#if PVW_TUNING
// ------------------------------------------------------------------------------------
// read-bandwidth probe (self-test / measurement aid): the loads of mac_rows -- 1-KiB tiles, 16 bytes per lane,
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
  if (xmap) {                                                 // every XCD a contiguous eighth of the runs (see mac_rows XMAP)
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
#define PVW_DISPATCH_ELL(ell, ...)                       \
  switch (ell) {                                         \
    case 8:  { constexpr int E = 8;  __VA_ARGS__; } break;   \
    case 16: { constexpr int E = 16; __VA_ARGS__; } break;   \
    case 32: { constexpr int E = 32; __VA_ARGS__; } break;   \
    case 64: { constexpr int E = 64; __VA_ARGS__; } break;   \
    default: return hipErrorInvalidValue;                \
  }

// PVW_MAC_VARIANT (tuning build only): selects the streaming schedule of mac_rows for l = 8 / 16
//   0 (default) by shape, see below | 17 U=8 (l=8) / 16 (l=16) double-buffered nt, not interleaved | 1 same, default cache policy | 2 U=4 dbuf nt | 3 U=16 dbuf nt
//   4 U=16 single buffer nt | 5 U=8 single buffer nt | 6 U=16 single buffer, default policy
//   7 continuous stream U=8 nt | 8 continuous stream U=16 nt | 9/10/11 waves interleave groups of U=8/16/4 tiles | 18/19 interleaved, single buffer, U=16/8
// The shipped library has the shape-selected schedule only (and no environment lookup).
static int mac_variant() {   // read per launch: the parity tests walk the variants in one process
  return (int)PVW_ENV_INT("PVW_MAC_VARIANT", 0);
}
// persistent form: grid = what the chip holds at once (occupancy query x CUs, a multiple of the shard count).
// Returns false when the shape does not qualify (the caller then uses the one-workgroup-per-item kernel): k must
// give every wave whole r-hat chunks (1, 2 or 4 of them), and there must be at least two items per resident
// workgroup -- below that there is nothing to balance and the plain grid starts faster.
template <int E, int U, int WPE, bool STAMP = false>
static bool launch_mac_persist(dim3 grid, hipStream_t s, const MacSection& sa, const MacSection& sb, const u64* rhat,
                               const Mod* mods, u32 k, u32 L, u32* counters) {
  constexpr int HALF = E / 2, JC = 256 / HALF;
  if constexpr (JC % U != 0 || (JC / U) % 2 != 0) return false;
  else {
    if (!counters || k % (4 * U) != 0 || (k / 4) % JC != 0) return false;
    const u32 nc = (k / 4) / JC;
    if (nc != 1 && nc != 2 && nc != 4) return false;
    static const int resident = [] {
      int dev = 0, cus = 0, per_cu = 0;
      if (hipGetDevice(&dev) != hipSuccess) return 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mac_rows_persist_kernel<E, U, WPE, 1, 4, STAMP>, 256, 0) != hipSuccess) return 0;
      return (cus * per_cu / PVW_PERSIST_SHARDS) * PVW_PERSIST_SHARDS;
    }();
    const u32 items = grid.x;
    if (resident <= 0 || items < 2 * (u32)resident) return false;
    // rounds every workgroup takes without asking (instant start, first successor known); everything after that
    // comes from the shard counters: the XCDs stream at rates 10-20 % apart and which ones are slow changes from
    // launch to launch, so nearly all of the work has to be up for grabs
    const u32 dyn_rounds = (u32)PVW_ENV_INT("PVW_MAC_STATIC", 2);
    const dim3 g((u32)resident);
    if (nc == 1) mac_rows_persist_kernel<E, U, WPE, 1, 4, STAMP><<<g, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L, items, dyn_rounds, counters);
    else if (nc == 2) mac_rows_persist_kernel<E, U, WPE, 2, 4, STAMP><<<g, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L, items, dyn_rounds, counters);
    else mac_rows_persist_kernel<E, U, WPE, 4, 4, STAMP><<<g, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L, items, dyn_rounds, counters);
    return true;
  }
}

template <int E>
static void launch_mac_variant(int variant, dim3 grid, hipStream_t s, const MacSection& sa, const MacSection& sb,
                               const u64* rhat, const Mod* mods, u32 k, u32 L, u32* counters) {
#if PVW_TUNING
  if constexpr (E <= 16) {
    switch (variant) {
      case 20: if (k % 64 == 0) { mac_rows_kernel<E, 16, true, true, true, 4, 4><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;
      case 21: if (k % 64 == 0) { mac_rows_kernel<E, 16, true, true, true, 4, 1, 0><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;   // timing only
      case 22: if (k % 64 == 0) { mac_rows_kernel<E, 16, true, true, true, 4, 1, 1><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;   // timing only
      case 23: if (k % 32 == 0) { mac_rows_kernel<E, 8, true, true, true, 4, 4><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;
      case 24: if (k % 32 == 0) { mac_rows_kernel<E, 16, true, true, true, 2><<<grid, dim3(128), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;   // two waves per item
      case 25: if (k % 16 == 0) { mac_rows_kernel<E, 8, true, true, true, 2><<<grid, dim3(128), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;
      case 50: if (k % 64 == 0) { mac_rows_kernel<E, 16, true, true, true, 4, 1, 2, false, 1><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;   // XCD-contiguous items
      case 51: if (k % 64 == 0) { mac_rows_kernel<E, 16, true, true, true, 4, 1, 2, true, 1><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;    // + stamps
      case 40: if (k % 64 == 0) { mac_rows_kernel<E, 16, true, true, true, 4, 1, 2, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;   // default schedule + stamps
      case 41: if (k % 32 == 0) { mac_rows_kernel<E, 8, true, true, true, 4, 4, 2, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;    // U = 8, four waves per SIMD + stamps
      case 42: if (launch_mac_persist<E, 16, 2, true>(grid, s, sa, sb, rhat, mods, k, L, counters)) return; break;   // + per-item stamps
      case 43: if (launch_mac_persist<E, 8, 3, true>(grid, s, sa, sb, rhat, mods, k, L, counters)) return; break;
      case 30: if (launch_mac_persist<E, 16, 2>(grid, s, sa, sb, rhat, mods, k, L, counters)) return; break;
      case 31: if (launch_mac_persist<E, 8, 3>(grid, s, sa, sb, rhat, mods, k, L, counters)) return; break;
      case 1: mac_rows_kernel<E, 8, false, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return;
      case 2: mac_rows_kernel<E, 4, true, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return;
      case 3: mac_rows_kernel<E, 16, true, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return;
      case 4: mac_rows_kernel<E, 16, true, false><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return;
      case 5: mac_rows_kernel<E, 8, true, false><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return;
      case 6: mac_rows_kernel<E, 16, false, false><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return;
      case 7: mac_rows_stream_kernel<E, 8, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return;
      case 8: mac_rows_stream_kernel<E, 16, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return;
      case 9: if (k % 32 == 0) { mac_rows_kernel<E, 8, true, true, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;
      case 10: if (k % 64 == 0) { mac_rows_kernel<E, 16, true, true, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;
      case 11: if (k % 16 == 0) { mac_rows_kernel<E, 4, true, true, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;
      case 12: if (k % 64 == 0) { mac_rows_kernel<E, 8, true, true, true, 8><<<grid, dim3(512), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;
      case 13: if (k % 128 == 0) { mac_rows_kernel<E, 8, true, true, true, 16><<<grid, dim3(1024), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;
      case 14: if (k % 128 == 0) { mac_rows_kernel<E, 16, true, true, true, 8><<<grid, dim3(512), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;
      case 15: if (k % 64 == 0) { mac_rows_kernel<E, 4, true, true, true, 16><<<grid, dim3(1024), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;
      case 16: if (k % 64 == 0) { mac_rows_kernel<E, 8, true, false, true, 8><<<grid, dim3(512), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;
      case 18: if (k % 64 == 0) { mac_rows_kernel<E, 16, true, false, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;
      case 19: if (k % 32 == 0) { mac_rows_kernel<E, 8, true, false, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L); return; } break;
      default: break;
    }
  }
#endif
  // defaults from the round-1 sweeps (profiles/r01_variant_sweep.txt, r01d_mac_ilv_sweep.txt): always
  // double-buffered non-temporal loads; when k allows it the four waves interleave groups of 16 tiles so the
  // workgroup reads one contiguous stream (+5 % at l = 16, k = 512; +2 % at n = 16384; within noise at config 3)
  if constexpr (E <= 16) {
    if (variant != 0) {
      // an explicit schedule that does not apply to this k falls back to the non-interleaved default
    } else if (k % 64 == 0) {
      mac_rows_kernel<E, 16, true, true, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L);
      return;
    }
  }
  if constexpr (E == 16) {
    mac_rows_kernel<E, 16, true, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L);
    return;
  }
  mac_rows_kernel<E, 8, true, true><<<grid, dim3(256), 0, s>>>(sa, sb, rhat, mods, k, L);
}

hipError_t launch_mac_rows(const MacSection& a, const MacSection& b, const u64* rhat,
                           const DevTables& t, u32 k, u32 L, u32 ell, hipStream_t s, u32* counters) {
  const u32 R = 128 / ell;
  MacSection sa = a, sb = b;
  sa.row_blocks = (sa.nrows + R - 1) / R;
  sb.row_blocks = (sb.nrows + R - 1) / R;
  const u32 blocks = (sa.row_blocks + sb.row_blocks) * L;
  if (blocks == 0) return hipSuccess;
  PVW_DISPATCH_ELL(ell, launch_mac_variant<E>(mac_variant(), dim3(blocks), s, sa, sb, rhat, t.mods, k, L, counters));
  return hipGetLastError();
}

hipError_t launch_mac_rows_packed(const MacSection& a, const MacSection& b, const u64* rhat, const DevTables& t, u32 k, u32 L,
                                  u32 ell, hipStream_t s) {
  if (ell > 16 || k % 256 != 0) return hipErrorInvalidValue;
  const u32 R = 128 / ell;
  MacSection sa = a, sb = b;
  sa.row_blocks = (a.nrows + R - 1) / R;
  sb.row_blocks = (b.nrows + R - 1) / R;
  const u32 blocks = (sa.row_blocks + sb.row_blocks) * L;
  if (blocks == 0) return hipSuccess;
#if PVW_TUNING
  if (PVW_ENV_INT("PVW_MAC_VARIANT", 0) == 44) {          // per-workgroup time stamps (tools/mac_timeline.py c3 44)
    if (ell == 8) mac_rows_packed_kernel<8, true><<<dim3(blocks), dim3(256), 0, s>>>(sa, sb, rhat, t.mods, k, L);
    else mac_rows_packed_kernel<16, true><<<dim3(blocks), dim3(256), 0, s>>>(sa, sb, rhat, t.mods, k, L);
    return hipGetLastError();
  }
#endif
  // tuning build, PVW_PACKED_DEEP=1 (k = 256): three windows per wave, the chunks of two groups in flight.  Measured equal to the
  // two-window form (profiles/r02_mac_rows_packed.txt: both land on ~177 or ~185 us at config 3 depending on the launch, not on
  // the form), so the shipped library keeps the smaller kernel
  const bool deep = PVW_TUNING && k == 256 && PVW_ENV_INT("PVW_PACKED_DEEP", 0) != 0;
  if (deep) {
    if (ell == 8) mac_rows_packed_kernel<8, false, true><<<dim3(blocks), dim3(256), 0, s>>>(sa, sb, rhat, t.mods, k, L);
    else mac_rows_packed_kernel<16, false, true><<<dim3(blocks), dim3(256), 0, s>>>(sa, sb, rhat, t.mods, k, L);
    return hipGetLastError();
  }
  if (ell == 8) mac_rows_packed_kernel<8><<<dim3(blocks), dim3(256), 0, s>>>(sa, sb, rhat, t.mods, k, L);
  else mac_rows_packed_kernel<16><<<dim3(blocks), dim3(256), 0, s>>>(sa, sb, rhat, t.mods, k, L);
  return hipGetLastError();
}
hipError_t launch_pack61(const u64* M, u64* P, u32 rows, u32 k, u32 L, u32 ell, u32* wide_flag, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  if (k % 64 != 0) return hipErrorInvalidValue;
  const u32 R = 128 / ell;
  const size_t items = (size_t)((rows + R - 1) / R) * L;
  pack61_kernel<<<dim3((u32)((items * 64 + 255) / 256)), dim3(256), 0, s>>>(M, P, k, items, wide_flag);
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

hipError_t launch_prep(const i64* coeffs, const u64* scalars, u64* out, size_t stride_poly,
                       size_t stride_limb, u32 count, bool do_ntt, const DevTables& t, u32 L,
                       u32 ell, hipStream_t s, u32 group, size_t stride_group) {
  if (count == 0) return hipSuccess;
  const u32 threads = count * L;
  PVW_DISPATCH_ELL(ell, prep_kernel<E><<<dim3((threads + 63) / 64), dim3(64), 0, s>>>(coeffs, scalars, out, stride_poly, stride_limb, count, L,
                                            do_ntt ? 1u : 0u, t, group, stride_group));
  return hipGetLastError();
}

hipError_t launch_transpose_polys(const u64* src, u64* dst, u32 k, u32 words, hipStream_t s) {
  const size_t total = (size_t)k * k * words;
  if (total == 0) return hipSuccess;
  transpose_polys_kernel<<<dim3((u32)((total + 255) / 256)), dim3(256), 0, s>>>(src, dst, k, words);
  return hipGetLastError();
}

hipError_t launch_ntt(u64* polys, size_t count, bool inverse, const DevTables& t, u32 L, u32 ell,
                      hipStream_t s) {
  // keep each launch below 2^31 threads
  const size_t step = (size_t)1 << 24;
  for (size_t off = 0; off < count; off += step) {
    const u32 cnt = (u32)((count - off) < step ? (count - off) : step);
    const u32 threads = cnt * L;
    u64* p = polys + off * L * ell;
    PVW_DISPATCH_ELL(ell, ntt_kernel<E><<<dim3((threads + 63) / 64), dim3(64), 0, s>>>(p, cnt, L, inverse ? 1u : 0u, t));
  }
  return hipGetLastError();
}

hipError_t launch_tile(const u64* src, u64* M, u32 rows, u32 row0_tiled, u32 k, u32 L, u32 ell,
                       bool ntt_first, const DevTables& t, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  const size_t threads = (size_t)rows * k * L;
  PVW_DISPATCH_ELL(ell, tile_kernel<E><<<dim3((u32)((threads + 255) / 256)), dim3(256), 0, s>>>(src, M, rows, row0_tiled, k, L, ntt_first ? 1u : 0u, t));
  return hipGetLastError();
}

hipError_t launch_untile(const u64* M, u64* dst, u32 rows, u32 row0_tiled, u32 k, u32 L, u32 ell,
                         bool intt_after, const DevTables& t, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  const size_t threads = (size_t)rows * k * L;
  PVW_DISPATCH_ELL(ell, untile_kernel<E><<<dim3((u32)((threads + 255) / 256)), dim3(256), 0, s>>>(M, dst, rows, row0_tiled, k, L, intt_after ? 1u : 0u, t));
  return hipGetLastError();
}

hipError_t launch_fill_uniform_tiled(u64* M, const ChaChaKey& key, u32 domain, u32 rows,
                                     u32 row0_tiled, u32 grow0, u32 k, u32 L, u32 ell,
                                     const DevTables& t, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  const size_t threads = (size_t)rows * k * L;
  PVW_DISPATCH_ELL(ell, fill_uniform_tiled_kernel<E><<<dim3((u32)((threads + 255) / 256)), dim3(256), 0, s>>>(M, key, domain, rows, row0_tiled, grow0, k, L, t));
  return hipGetLastError();
}

hipError_t launch_sample(i64* out, const ChaChaKey& key, u32 ell, const SampleJob& j0,
                         const SampleJob& j1, const SampleJob& j2, hipStream_t s) {
  const u32 threads = j0.count + j1.count + j2.count;
  if (threads == 0) return hipSuccess;
  sample_kernel<<<dim3((threads + 63) / 64), dim3(64), 0, s>>>(out, key, ell, j0, j1, j2);
  return hipGetLastError();
}

hipError_t launch_prologue(const PrologueBatch& batch, const DevTables& t, u32 L, u32 ell, hipStream_t s) {
  PrologueBatch b = batch;
  b.total = 0;
  b.debug = (u32)PVW_ENV_INT("PVW_PROLOGUE_DEBUG", 0);   // tuning build only (timing experiments: 1 = skip sampling, 2 = skip transform); the shipped kernel has neither branch
  if (b.njobs > PVW_MAX_PROLOGUE_JOBS) return hipErrorInvalidValue;
  if (b.reps == 0) b.reps = 1;
  if (b.reps > 65535) return hipErrorInvalidValue;
  b.key_window = 1;
  b.key_rep = b.njobs ? b.job[0].rep_key : 0;
  for (u32 i = 0; i < b.njobs; ++i) {
    b.total += b.job[i].sj.count;
    if (b.job[i].key_idx + (b.reps - 1) * b.job[i].rep_key >= PVW_MAX_PROLOGUE_KEYS) return hipErrorInvalidValue;
    if (b.job[i].rep_key != b.key_rep) return hipErrorInvalidValue;     // one key policy per batch: shared, or one per replica
    if (b.job[i].key_idx + 1 > b.key_window) b.key_window = b.job[i].key_idx + 1;
  }
  if (b.key_window > 8) return hipErrorInvalidValue;
  if (b.total == 0) return hipSuccess;
  if (L > 256) return hipErrorInvalidValue;
  // polynomials per block: 256/L (one trip of the transform loop, lowest latency) for one encrypt's worth of
  // work; a whole wave of samplers (64) when the launch is large enough to fill the chip anyway
  u32 PB = 256 / L;
  if (PB > 64) PB = 64;
  if ((size_t)b.total * b.reps >= 65536) PB = 64;
  const u32 blocks = (b.total + PB - 1) / PB;
  const size_t sc_bytes = (size_t)PB * ell * 8, tab_bytes = (size_t)4 * L * ell * 8;
  const size_t desc_bytes = (size_t)PVW_MAX_PROLOGUE_JOBS * sizeof(PrologueJob) + 8 * sizeof(ChaChaKey);
  const u32 stage = (sc_bytes + tab_bytes + desc_bytes <= 64 * 1024) ? 1u : 0u;
  const size_t lds = sc_bytes + (stage ? tab_bytes : 0) + desc_bytes;
  PVW_DISPATCH_ELL(ell, prologue_kernel<E><<<dim3(blocks, b.reps), dim3(256), lds, s>>>(b, L, PB, stage, t));
  return hipGetLastError();
}

hipError_t launch_gaussian(i64* out, const ChaChaKey& key, u32 index0, u32 count, u64 bound,
                           hipStream_t s) {
  if (count == 0) return hipSuccess;
  gaussian_kernel<<<dim3((count + 63) / 64), dim3(64), 0, s>>>(out, key, index0, count, bound);
  return hipGetLastError();
}

// how many ranges of j a decrypt over `dealers` ciphertexts is cut into (1 = no split): enough workgroups to put one
// on every CU when the batch alone does not, never ranges shorter than 64 terms
u32 decrypt_split(u32 k, u32 L, u32 ell, size_t dealers) {
  const u32 pairs = L * ell / 2;
  if (pairs > 1024 || dealers == 0) return 1;            // the generic form does not split
  u32 ns = (u32)PVW_ENV_INT("PVW_DEC_SPLIT", 0);          // tuning build: forced split
  if (ns == 0) {
    // measured at config 5 (profiles/r02_decrypt_split.txt): once every CU has a workgroup, cutting the ranges only
    // costs (356 -> 370 / 377 / 387 us at 2 / 4 / 8 ranges); the split is for small batches -- a single
    // decrypt_party_value is one workgroup streaming k polynomials alone otherwise
    const size_t wgs = (dealers + 1) / 2;
    ns = wgs >= 256 ? 1 : (u32)((256 + wgs - 1) / wgs);
    if (ns > 8) ns = 8;
  }
  while (ns > 1 && (k + ns - 1) / ns < 64) --ns;
  return ns ? ns : 1;
}

hipError_t launch_decrypt_finish(const u64* partial, u32 nsplit, const u64* c2col, u64* noisy, const DevTables& t, u32 L, u32 ell,
                                 size_t dealers, hipStream_t s) {
  if (dealers == 0) return hipSuccess;
  const u32 threads = (u32)dealers * L;
  PVW_DISPATCH_ELL(ell, decrypt_finish_kernel<E><<<dim3((threads + 63) / 64), dim3(64), 0, s>>>(partial, nsplit, c2col, noisy, (u32)dealers, L, t));
  return hipGetLastError();
}

hipError_t launch_decrypt_mac(const u64* c1s, const u64* shat, const u64* c2col, u64* noisy,
                              const DevTables& t, u32 k, u32 L, u32 ell, size_t dealers,
                              hipStream_t s, u64* partial, u32 nsplit) {
  if (dealers == 0) return hipSuccess;
  if (nsplit == 0 || !partial) nsplit = 1;
  const u32 pairs = L * ell / 2;
  u32 step, c, threads, ny;
  if (pairs <= 1024) {
    step = pairs;
    ny = 1;
    c = pairs >= 256 ? 1 : 256 / pairs;
    if (c > k) c = k;
    threads = ((c * pairs + 63) / 64) * 64;
  } else {
    step = 1024;
    ny = (pairs + 1023) / 1024;
    c = 1;
    threads = 1024;
  }
  // variant 0 (default) picks by shape: the full-width form from 128 slot pairs per polynomial up, the
  // dealer-grouped form below that (profiles/r01_variant_sweep.txt, profiles/r01d_decrypt_sweep.txt).
  // The environment is read on every launch so that the tests can walk the variants in one process.
  int variant = (int)PVW_ENV_INT("PVW_DEC_VARIANT", 0), cenv = (int)PVW_ENV_INT("PVW_DEC_C", 0);   // tuning build only
  if (variant == 0) variant = pairs >= 128 && pairs <= 1024 ? 61 : 10;
  if (variant < 60 && pairs <= 1024 && cenv > 0 && (u32)cenv * pairs <= 1024 && (u32)cenv <= k) {
    c = (u32)cenv;
    threads = ((c * pairs + 63) / 64) * 64;
  }
  const size_t lds = (size_t)threads * sizeof(v2u64);
  if (variant >= 60 && pairs <= 1024) {
    const u32 FW = pairs / 64, rem = pairs % 64;
    u32 remp = 0;
    if (rem) { remp = 1; while (remp < rem) remp <<= 1; }
    // replicas of the full waves (each takes every cfull-th j): the largest ODD count that fits 16 waves.
    // Measured at l=16, L=34 (272 pairs): 1, 2 replicas 398-402 us, 3 replicas 357 us; even counts lose on
    // every shape tried, and one 13-wave workgroup per CU beats two 5-wave ones.
    const u32 has_rem = rem ? 1u : 0u;
    u32 cfull = 0;
    if (FW) {
      cfull = (16 - has_rem) / FW;
      if (cfull > 1 && cfull % 2 == 0) --cfull;
      if (cfull > 7) cfull = 7;
      if (cenv > 0 && (u32)cenv * FW + has_rem <= 16) cfull = (u32)cenv;
      if (cfull > k) cfull = k;
    }
    const u32 crem = rem ? (FW ? 1 : 4) : 0;
    const u32 waves = cfull * FW + crem;
    if (waves >= 1 && waves <= 16) {
      const u32 thr = waves * 64;
#define PVW_DEC_FW(DGv, UJv)                                                                                       \
  decrypt_mac_fw_kernel<DGv, UJv><<<dim3((u32)((dealers + DGv - 1) / DGv), nsplit), dim3(thr), (size_t)DGv * thr * sizeof(v2u64), s>>>( \
      c1s, shat, c2col, noisy, t.mods, k, ell, pairs, FW, cfull, remp, crem, (u32)dealers, partial)
#if PVW_TUNING
      switch (variant) {
        case 61: PVW_DEC_FW(2, 4); break;
        case 62: PVW_DEC_FW(1, 4); break;
        case 63: PVW_DEC_FW(1, 8); break;
        case 64: PVW_DEC_FW(3, 2); break;
        default: PVW_DEC_FW(2, 2); break;
      }
#else
      PVW_DEC_FW(2, 4);
#endif
#undef PVW_DEC_FW
      return hipGetLastError();
    }
  }
  if (variant >= 10 && ny == 1) {
    // dealer-grouped kernels: variant 1x = DG 2, 2x = DG 4; x = 0: UJ 2, 1: UJ 4 (DG 2) / UJ 1 (DG 4)
#define PVW_DEC_GROUPED(DGv, UJv)                                                                         \
  do {                                                                                                    \
    if (threads <= 512)                                                                                   \
      decrypt_mac_grouped_kernel<DGv, UJv, 512><<<dim3((u32)((dealers + DGv - 1) / DGv), nsplit), dim3(threads), lds, s>>>( \
          c1s, shat, c2col, noisy, t.mods, k, ell, pairs, c, (u32)dealers, partial);                       \
    else                                                                                                  \
      decrypt_mac_grouped_kernel<DGv, UJv, 1024><<<dim3((u32)((dealers + DGv - 1) / DGv), nsplit), dim3(threads), lds, s>>>( \
          c1s, shat, c2col, noisy, t.mods, k, ell, pairs, c, (u32)dealers, partial);                       \
  } while (0)
#if PVW_TUNING
    switch (variant) {
      case 10: PVW_DEC_GROUPED(2, 2); break;
      case 11: PVW_DEC_GROUPED(2, 4); break;
      case 20: PVW_DEC_GROUPED(4, 2); break;
      case 21: PVW_DEC_GROUPED(4, 1); break;
      case 30: PVW_DEC_GROUPED(1, 8); break;
      case 40: PVW_DEC_GROUPED(3, 2); break;
      case 41: PVW_DEC_GROUPED(4, 2); break;
      case 50:   // timing experiment only (wrong results): the c1 stream without the s-hat loads
        decrypt_mac_grouped_kernel<2, 2, 512, true><<<dim3((u32)((dealers + 1) / 2), nsplit), dim3(threads), lds, s>>>(
            c1s, shat, c2col, noisy, t.mods, k, ell, pairs, c, (u32)dealers, partial);
        break;
      default: PVW_DEC_GROUPED(2, 2); break;
    }
#else
    PVW_DEC_GROUPED(2, 2);
#endif
#undef PVW_DEC_GROUPED
    return hipGetLastError();
  }
  for (size_t off = 0; off < dealers; off += 32768) {
    const u32 nd = (u32)((dealers - off) < 32768 ? (dealers - off) : 32768);
    const u64* c1p = c1s + off * (size_t)k * L * ell;
    const u64* c2p = c2col + off * (size_t)L * ell;
    u64* np = noisy + off * (size_t)L * ell;
#if PVW_TUNING
    switch (variant) {
      case 1: decrypt_mac_kernel<4, false><<<dim3(nd, ny), dim3(threads), lds, s>>>(c1p, shat, c2p, np, t.mods, k, ell, pairs, c, step); break;
      case 2: decrypt_mac_kernel<8, false><<<dim3(nd, ny), dim3(threads), lds, s>>>(c1p, shat, c2p, np, t.mods, k, ell, pairs, c, step); break;
      case 3: decrypt_mac_kernel<8, true><<<dim3(nd, ny), dim3(threads), lds, s>>>(c1p, shat, c2p, np, t.mods, k, ell, pairs, c, step); break;
      case 4: decrypt_mac_kernel<2, true><<<dim3(nd, ny), dim3(threads), lds, s>>>(c1p, shat, c2p, np, t.mods, k, ell, pairs, c, step); break;
      default: decrypt_mac_kernel<4, true><<<dim3(nd, ny), dim3(threads), lds, s>>>(c1p, shat, c2p, np, t.mods, k, ell, pairs, c, step); break;
    }
#else
    decrypt_mac_kernel<4, true><<<dim3(nd, ny), dim3(threads), lds, s>>>(c1p, shat, c2p, np, t.mods, k, ell, pairs, c, step);
#endif
  }
  return hipGetLastError();
}

hipError_t launch_mftile(const u64* src, bool src_is_tiled, u64* XM, u32 rows, u32 k, u32 L, u32 ell, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  if (!src_is_tiled && ell <= 32) {      // API-layout rows: the LDS-transposing form (writes every tile, padding included)
    const u32 jbg = ell <= 8 ? 4 : (ell == 16 ? 2 : 1);
    const u32 JB = (k + 3) / 4, JG = (JB + jbg - 1) / jbg;
    const u32 RT = ((rows + PVW_GEMM_ROWS_PER_WG - 1) / PVW_GEMM_ROWS_PER_WG) * (PVW_GEMM_ROWS_PER_WG / 32);
    switch (ell) {
      case 8: mftile_rows_kernel<8><<<dim3(JG * RT * L), dim3(256), 0, s>>>(src, XM, rows, k, L); break;
      case 16: mftile_rows_kernel<16><<<dim3(JG * RT * L), dim3(256), 0, s>>>(src, XM, rows, k, L); break;
      default: mftile_rows_kernel<32><<<dim3(JG * RT * L), dim3(256), 0, s>>>(src, XM, rows, k, L); break;
    }
    return hipGetLastError();
  }
  const size_t threads = (size_t)rows * k * L;
  PVW_DISPATCH_ELL(ell, mftile_kernel<E><<<dim3((u32)((threads + 255) / 256)), dim3(256), 0, s>>>(src, src_is_tiled ? 1u : 0u, XM, rows, k, L));
  return hipGetLastError();
}

hipError_t launch_shat_mftile(const i64* coeffs, u64* XM, u32 rows, u32 k, u32 L, u32 ell, const DevTables& t, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  if (ell > 32) return hipErrorInvalidValue;                 // callers keep the prologue + launch_mftile pair for l = 64
  const u32 jbg = 2;
  const u32 JB = (k + 3) / 4, JG = (JB + jbg - 1) / jbg;
  const u32 RT = ((rows + PVW_GEMM_ROWS_PER_WG - 1) / PVW_GEMM_ROWS_PER_WG) * (PVW_GEMM_ROWS_PER_WG / 32);
  u32 ls = 1;                                                 // limb interleave: enough blocks for several rounds on the chip
  while (ls < L && (size_t)JG * RT * ls < 4096) ls *= 2;
  if (ls > L) ls = L;
  switch (ell) {
    case 8: shat_mftile_kernel<8><<<dim3(JG, RT, ls), dim3(256), 0, s>>>(coeffs, XM, rows, k, L, t); break;
    case 16: shat_mftile_kernel<16><<<dim3(JG, RT, ls), dim3(256), 0, s>>>(coeffs, XM, rows, k, L, t); break;
    default: shat_mftile_kernel<32><<<dim3(JG, RT, ls), dim3(256), 0, s>>>(coeffs, XM, rows, k, L, t); break;
  }
  return hipGetLastError();
}

hipError_t launch_vec_digits(const u64* vhat, size_t vstride, signed char* YD, int* SY, u32 nv, u32 k, u32 L, u32 ell,
                             const DevTables& t, hipStream_t s, size_t lstride, size_t jstride) {
  if (nv == 0) return hipSuccess;
  if (lstride == 0 && jstride == 0) { lstride = (size_t)k * ell; jstride = ell; }
  // unused vector slots of the last group must read as zero digits / zero sums
  if (nv % 4) {
    const u32 NVG = (nv + 3) / 4, JB = (k + 3) / 4;
    hipError_t e = hipMemsetAsync(YD + (size_t)(NVG - 1) * L * ell * JB * 1024, 0, (size_t)L * ell * JB * 1024, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(SY + (size_t)(NVG - 1) * L * ell * 32, 0, (size_t)L * ell * 32 * sizeof(int), s);
    if (e != hipSuccess) return e;
  }
  // default: stores staged through LDS (whole 128-byte lines per instruction, 16 KiB per wave); PVW_VEC_DIGITS_STAGE=0: direct
#if PVW_TUNING
  static const int stage = (int)PVW_ENV_INT("PVW_VEC_DIGITS_STAGE", 1);
  if (!stage) {
    PVW_DISPATCH_ELL(ell, vec_digits_kernel<E, false><<<dim3(nv * L * ell), dim3(64), 0, s>>>(vhat, vstride, YD, SY, nv, k, L, t, lstride, jstride));
    return hipGetLastError();
  }
#endif
  PVW_DISPATCH_ELL(ell, vec_digits_kernel<E, true><<<dim3(nv * L * ell), dim3(64), 0, s>>>(vhat, vstride, YD, SY, nv, k, L, t, lstride, jstride));
  return hipGetLastError();
}

hipError_t launch_gemm_digits(const GemmSection& a, const GemmSection& b, const signed char* YD, const int* SY,
                              const DevTables& t, u32 k, u32 L, u32 ell, u32 nv, size_t ostride_a, size_t ostride_b,
                              hipStream_t s, const GemmErrSource* es_a, const GemmErrSource* es_b) {
  GemmSection sa = a, sb = b;
  sa.rt_groups = (sa.nrows + PVW_GEMM_ROWS_PER_WG - 1) / PVW_GEMM_ROWS_PER_WG;
  sb.rt_groups = (sb.nrows + PVW_GEMM_ROWS_PER_WG - 1) / PVW_GEMM_ROWS_PER_WG;
  const u32 blocks = (sa.rt_groups + sb.rt_groups) * L * ell;
  if (blocks == 0 || nv == 0) return hipSuccess;
  const u32 vbn = (nv + 15) / 16;
  const u32 NVG = vbn > 1 ? 4 : (nv + 3) / 4;
  const u32 nv_pad = NVG * 4;
  sa.tmp_bstride = (size_t)L * ell * 16 * sa.rt_groups * PVW_GEMM_ROWS_PER_WG;
  sb.tmp_bstride = (size_t)L * ell * 16 * sb.rt_groups * PVW_GEMM_ROWS_PER_WG;
  const size_t yd_b16 = yd_bytes(16, k, L, ell), sy_b16 = sy_bytes(16, L, ell) / sizeof(int);
  const u32 dbg = (u32)PVW_ENV_INT("PVW_GEMM_DEBUG", 0);   // tuning build only; the shipped kernel ignores the argument
#define PVW_GEMM_LAUNCH(G, N)                                                                                              \
  do {                                                                                                                    \
    if (t.min_q_bits >= 55) { PVW_DISPATCH_ELL(ell, gemm_digits_kernel<E, G, PVW_GEMM_RPW, N, true><<<dim3(blocks * vbn), dim3(256), 0, s>>>(sa, sb, YD, SY, t.mods, k, L, nv, nv_pad, dbg, vbn, yd_b16, sy_b16)); } \
    else { PVW_DISPATCH_ELL(ell, gemm_digits_kernel<E, G, PVW_GEMM_RPW, N, false><<<dim3(blocks * vbn), dim3(256), 0, s>>>(sa, sb, YD, SY, t.mods, k, L, nv, nv_pad, dbg, vbn, yd_b16, sy_b16)); } \
  } while (0)
#if PVW_TUNING
  // timing experiment (results wrong): all-zero operand bytes, to separate the schedule from the data-dependent power draw
  if (PVW_ENV_INT("PVW_GEMM_ZERO_OPERANDS", 0)) {
    if (sa.nrows) (void)hipMemsetAsync(const_cast<u64*>(sa.XM), 0, xm_words(sa.nrows, k, L, ell) * 8, s);
    if (sb.nrows) (void)hipMemsetAsync(const_cast<u64*>(sb.XM), 0, xm_words(sb.nrows, k, L, ell) * 8, s);
    (void)hipMemsetAsync(const_cast<signed char*>(YD), 0, yd_b16 * vbn, s);
  }
#endif
  // more than 16 vectors and whole stages of 16 terms: the wide form (256 rows x 32 vectors per workgroup, both
  // operands through LDS).  PVW_GEMM_WIDE=0 in the tuning build selects gemm_digits_kernel everywhere.
  const bool wide = vbn >= 2 && k % 16 == 0 && k >= 16 && PVW_ENV_INT("PVW_GEMM_WIDE", 1) != 0;
  if (wide) {
    // shipped: 8 waves in ping-pong (256 rows x 32 vectors).  Tuning build, PVW_GEMM_WIDE: 1 = 8 waves in step, 2 = 4 waves
    // (128 x 32, two workgroups per CU: one's epilogue under the other's MFMAs), 3 = the shipped form
    [[maybe_unused]] const int wform = (int)PVW_ENV_INT("PVW_GEMM_WIDE", 3);
#define PVW_GEMM_WIDE_LAUNCH(WRN, PP)                                                                                      \
  do {                                                                                                                    \
    const u32 ga = (sa.rt_groups * 4 + 2 * WRN - 1) / (2 * WRN), gb2 = (sb.rt_groups * 4 + 2 * WRN - 1) / (2 * WRN);       \
    const u32 wblocks = (ga + gb2) * L * ell * ((vbn + 1) / 2);                                                           \
    if (t.min_q_bits >= 55) { PVW_DISPATCH_ELL(ell, gemm_digits_wide_kernel<E, true, WRN, PP><<<dim3(wblocks), dim3(128 * WRN), 0, s>>>(sa, sb, YD, t.mods, k, L, nv, nv_pad, vbn, yd_b16)); } \
    else { PVW_DISPATCH_ELL(ell, gemm_digits_wide_kernel<E, false, WRN, PP><<<dim3(wblocks), dim3(128 * WRN), 0, s>>>(sa, sb, YD, t.mods, k, L, nv, nv_pad, vbn, yd_b16)); } \
  } while (0)
#if PVW_TUNING
    if (wform == 1) PVW_GEMM_WIDE_LAUNCH(4, false);
    else if (wform == 2) PVW_GEMM_WIDE_LAUNCH(2, false);
    else
#endif
      PVW_GEMM_WIDE_LAUNCH(4, true);
#undef PVW_GEMM_WIDE_LAUNCH
  } else {
  // fully unrolled chunk loops for the BASELINE geometries (k = 256: 8 chunks of 8 j-blocks, k = 512: 16), full vector groups
  static const int unroll_ok = (int)PVW_ENV_INT("PVW_GEMM_UNROLL", 1);
  if (NVG == 4 && unroll_ok && !(dbg & 1) && k == 256) { PVW_GEMM_LAUNCH(4, 8); }
  else if (NVG == 4 && unroll_ok && !(dbg & 1) && k == 512) { PVW_GEMM_LAUNCH(4, 16); }
  else switch (NVG) {
    case 1: PVW_GEMM_LAUNCH(1, 0); break;
    case 2: PVW_GEMM_LAUNCH(2, 0); break;
    case 3: PVW_GEMM_LAUNCH(3, 0); break;
    case 4: PVW_GEMM_LAUNCH(4, 0); break;
    default: return hipErrorInvalidValue;
  }
#undef PVW_GEMM_LAUNCH
  }
  // a section with an error source: gemm_finish_err_kernel (l <= 32; the tiled form only with tiled_swap)
  auto finish = [&](const GemmSection& sec, size_t ostride, const GemmErrSource* es) -> hipError_t {
    if (!sec.nrows) return hipSuccess;
    const u32 rows_pad = sec.rt_groups * PVW_GEMM_ROWS_PER_WG;
    if (es) {
      if (ell > 32 || (sec.tiled_out && !sec.tiled_swap) || (!sec.tiled_out && !sec.out)) return hipErrorInvalidValue;
      // one launch per GemmErrSource of the array: es[i] covers the next es[i].span vectors (0: all that are left)
      for (u32 v_lo = 0; v_lo < nv; ++es) {
        const u32 span = es->span && es->span < nv - v_lo ? es->span : nv - v_lo, v_hi = v_lo + span;
        const u32 gx = (sec.nrows + 31) / 32, gy = (span + 7) / 8;
        u32 lz = 1;                                          // limb interleave: enough blocks for several rounds on the chip
        const size_t want = (size_t)PVW_ENV_INT("PVW_FINISH_BLOCKS", 4096);   // tuning build: blocks the limb split aims at (1024 .. 16384 measured: 4096)
        while (lz < L && (size_t)gx * gy * lz < want) lz *= 2;
        if (lz > L) lz = L;
        const dim3 grid(gx, gy, lz);
        switch (ell) {
          case 8: gemm_finish_err_kernel<8><<<grid, dim3(256), 0, s>>>(sec, t, L, nv, nv_pad, rows_pad, ostride, SY, sy_b16, *es, v_lo, v_hi); break;
          case 16: gemm_finish_err_kernel<16><<<grid, dim3(256), 0, s>>>(sec, t, L, nv, nv_pad, rows_pad, ostride, SY, sy_b16, *es, v_lo, v_hi); break;
          default: gemm_finish_err_kernel<32><<<grid, dim3(256), 0, s>>>(sec, t, L, nv, nv_pad, rows_pad, ostride, SY, sy_b16, *es, v_lo, v_hi); break;
        }
        v_lo = v_hi;
      }
      return hipGetLastError();
    }
    PVW_DISPATCH_ELL(ell, gemm_finish_kernel<E><<<dim3((sec.nrows + 31) / 32, (nv + (E >= 64 ? 2 : 4) - 1) / (E >= 64 ? 2 : 4), L), dim3(256), 0, s>>>(
                              sec, t.mods, L, nv, nv_pad, rows_pad, ostride, SY, sy_b16));
    return hipGetLastError();
  };
  hipError_t fe = finish(sa, ostride_a, es_a);
  if (fe != hipSuccess) return fe;
  return finish(sb, ostride_b, es_b);
}

#if PVW_TUNING
hipError_t read_wg_stamps(u64* out, u32 count) {
  if (count > 4096) return hipErrorInvalidValue;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp_wg), (size_t)count * 16, 0, hipMemcpyDeviceToHost);
}
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
#endif  // PVW_TUNING

hipError_t launch_mfma_probe(const signed char* A, const signed char* B, int* C, hipStream_t s) {
  mfma_i8_probe_kernel<<<dim3(1), dim3(64), 0, s>>>(A, B, C);
  return hipGetLastError();
}

// Kernels that may ask for more than the default 64 KiB of dynamic LDS.  The attribute is per device and per
// code object, so it is set once per CONTEXT while the context initialises its device (ensure_device, under the
// context's init mutex, after hipSetDevice) -- not lazily behind process-wide flags.
hipError_t init_kernel_attributes() {
  const int big = 160 * 1024;
  hipError_t e = hipFuncSetAttribute((const void*)decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)decode_chain_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)decode_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e != hipSuccess) return e;
#if PVW_TUNING
  hipFuncSetAttribute((const void*)read_probe2_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)read_probe2_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)read_probe2_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)read_probe2_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)read_probe2_kernel<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)read_probe2_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)read_probe2_kernel<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  e = hipFuncSetAttribute((const void*)decode_chain_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)decode_chain_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e != hipSuccess) return e;
#endif
  return hipSuccess;
}

hipError_t launch_decode(const u64* noisy, u64* out, size_t count, const DecodeTables& t, hipStream_t s) {
  if (count == 0) return hipSuccess;
  const size_t lds = (size_t)(2 * t.W + 1 + t.L) * 64 * sizeof(u64);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  // by shape: the lifted chain (4 waves per ciphertext) while L <= 64 and W + 2 <= 63 (Q up to ~3900 bits) and its
  // tables fit the LDS; one wave per ciphertext below W + 2 <= 64; one thread per ciphertext beyond.
  // tuning build: PVW_DECODE_VARIANT 1 thread per ciphertext | 2 one wave per ciphertext (RNS round trip per step)
  //   | 3 / 4 lifted chain with 2 / 8 waves per ciphertext
  const int variant = (int)PVW_ENV_INT("PVW_DECODE_VARIANT", 0);
  if ((variant == 0 || variant >= 3) && t.L <= 64 && t.W + 2 <= 63) {
    const u32 wpc = variant == 3 ? 2 : (variant == 4 ? 8 : 4);
    const u32 cpw = 8 / wpc ? 8 / wpc : 1;
    const size_t bytes = ((size_t)t.L * t.W + 2 * (2 * t.W + 2) + (size_t)cpw * ((size_t)(t.ell + 1) * 64 + (size_t)t.L * t.ell) +
                          (size_t)cpw * wpc * 64) * 8;
    if (bytes <= 160 * 1024) {
      const dim3 grid((u32)((count + cpw - 1) / cpw)), block(cpw * wpc * 64);
      // tuning build only: PVW_DECODE_TIMING=1|2|3: out[] = cycles of phase 1 | first division | chain (results are NOT values)
      const u32 dbg = (u32)PVW_ENV_INT("PVW_DECODE_TIMING", 0);
      const u32 arg = cpw | (dbg << 16);
#if PVW_TUNING
      if (wpc == 2) { decode_chain_kernel<2><<<grid, block, bytes, s>>>(noisy, out, (u32)count, arg, t); return hipGetLastError(); }
      if (wpc == 8) { decode_chain_kernel<8><<<grid, block, bytes, s>>>(noisy, out, (u32)count, arg, t); return hipGetLastError(); }
#endif
      decode_chain_kernel<4><<<grid, block, bytes, s>>>(noisy, out, (u32)count, arg, t);
      return hipGetLastError();
    }
  }
  if (variant != 1 && t.L <= 64 && t.W + 2 <= 64) {
    const size_t tab = ((size_t)2 * t.L * t.W + 4 * (t.W + 2)) * 8, scratch = (size_t)4 * 64 * 8, zb = (size_t)4 * t.L * t.ell * 8;
    const u32 stage_z = (tab + scratch + zb <= 96 * 1024) ? 1u : 0u;
    const size_t wl = tab + scratch + (stage_z ? zb : 0);
    decode_wave_kernel<<<dim3((u32)((count + 3) / 4)), dim3(256), wl, s>>>(noisy, out, (u32)count, stage_z, t);
    return hipGetLastError();
  }
  decode_kernel<<<dim3((u32)((count + 63) / 64)), dim3(64), lds, s>>>(noisy, out, (u32)count, t);
  return hipGetLastError();
}

}  // namespace pvw
static_assert(sizeof(pvw::PrologueBatch) <= 4000, "PrologueBatch must fit the kernel-argument segment");
