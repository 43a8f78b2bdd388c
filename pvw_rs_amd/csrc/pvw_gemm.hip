// pvw_gemm.hip -- the many-vector rows of the PVW path on the gfx950 matrix cores: multi-dealer encrypt
// (encrypt_all_party_shares, encryption.rs:253-286) and batched key generation (public_key.rs:111-147 over
// crs.rs:138-171) as a modular GEMM per (limb, slot) folded into v_mfma_i32_32x32x32_i8.
#include <hip/hip_runtime.h>

#include "pvw_arith.h"
#include "pvw_chacha.h"
#include "pvw_decode.h"
#include "pvw_kernels.h"
#include "pvw_dev.h"

namespace pvw {

// ------------------------------------------------------------------------------------
// i8 MFMA operand-map probe (self-test): C[32][32] = A[32][32] * B[32][32] with ONE
// v_mfma_i32_32x32x32_i8, operands fetched with the lane maps the digit-GEMM kernels rely on:
//   A fragment of lane l (r = l & 31, h = l >> 5): A[r][16h .. 16h+15]     (16 consecutive K)
//   B fragment:                                   B[16h .. 16h+15][r]
//   C/D register g of lane l:                      C[(g & 3) + 8 (g >> 2) + 4 h][r]
// ------------------------------------------------------------------------------------
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

// ---- 7-byte contraction (every modulus below 2^56; gemm_ktiles) ----
// Byte 7 of every matrix element is zero, so K = (j, a) needs a < 7 only: 7/8 of the MFMAs, of the LDS traffic and of the
// operand bytes.  The K rows are regrouped so that both producers stay simple: K tile (g, a) = byte position a of the 32
// consecutive j of group g; row j % 32 of the tile.  A fragment of lane (h, m): byte a of x[row m][32 g + 16 h + p],
// p = 0..15; digit fragment of lane (h, col = 8 v4 + b): digit b of the shifted copy a of y_v[32 g + 16 h + p].
// tiled matrix (or API-layout rows) -> XM7[limb][slot][row tile][g*7 + a][lane][16 bytes]; one thread per (row, limb,
// slot, g, h) reads its 16 elements and writes 7 fragments.  A one-off per matrix (pvw_prepare / first multi-dealer call).
template <int ELL>
__global__ __launch_bounds__(256) void mftile7_kernel(const u64* __restrict__ src, u32 src_is_tiled, u64* __restrict__ XM,
                                                       u32 rows, u32 k, u32 L) {
  constexpr int R = 128 / ELL;
  const u32 G = k / 32, KT = 7 * G;
  const u32 RT = ((rows + PVW_GEMM_ROWS_PER_WG - 1) / PVW_GEMM_ROWS_PER_WG) * (PVW_GEMM_ROWS_PER_WG / 32);
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)RT * 32 * L * ELL * G * 2;      // padding rows included: they must hold the offset-zero pattern
  if (tid >= total) return;
  const u32 slot = tid % ELL;
  size_t r = tid / ELL;
  const u32 row = r % (RT * 32);
  r /= (RT * 32);
  const u32 limb = r % L;
  r /= L;
  const u32 h = r & 1, g = (u32)(r >> 1);
  u64 x[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const u32 j = 32 * g + 16 * h + p;
    u64 v = 0;
    if (row < rows)
      v = src_is_tiled ? src[(((size_t)(row / R) * L + limb) * k + j) * 128 + (row % R) * ELL + slot]
                       : src[(((size_t)row * k + j) * L + limb) * ELL + slot];
    x[p] = v;
  }
  const u32 rt = row >> 5, m = row & 31;
  u64* base = XM + ((((size_t)limb * ELL + slot) * RT + rt) * KT + (size_t)g * 7) * 128 + (h * 32 + m) * 2;
#pragma unroll
  for (int a = 0; a < 7; ++a) {
    u64 lo = 0, hi = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      lo |= ((x[p] >> (8 * a)) & 0xff) << (8 * p);
      hi |= ((x[8 + p] >> (8 * a)) & 0xff) << (8 * p);
    }
    *reinterpret_cast<v2u64*>(base + (size_t)a * 128) = (v2u64){lo ^ 0x8080808080808080ULL, hi ^ 0x8080808080808080ULL};   // signed-offset, as mftile_kernel
  }
}
// vector elements -> digit tiles YD[vg][limb][slot][g*7 + a][h*32 + 8 v4 + b][16] and the offset correction SY (as
// vec_digits_kernel; the sum runs over the 7 copies that take part).  One wave per (v, limb, slot); lane = block of 16 j.
template <int ELL>
__global__ __launch_bounds__(64) void vec_digits7_kernel(const u64* __restrict__ vhat, size_t vstride,
                                                          signed char* __restrict__ YD, int* __restrict__ SY,
                                                          u32 nv, u32 k, u32 L, DevTables t, size_t lstride, size_t jstride) {
  const u32 lane = threadIdx.x;
  const u32 slot = blockIdx.x % ELL;
  const u32 limb = (blockIdx.x / ELL) % L;
  const u32 v = blockIdx.x / (ELL * L);
  const Mod m = t.mods[limb];
  const u32 KT = 7 * (k / 32);
  const u32 vg = v >> 2, v4 = v & 3;
  const u64 w256p = (m.ratio_hi << 8) | (m.ratio_lo >> 56);   // floor(256 * 2^64 / q)
  const u64* y = vhat + (size_t)v * vstride + (size_t)limb * lstride + slot;
  signed char* tiles = YD + (((size_t)vg * L + limb) * ELL + slot) * (size_t)KT * 1024;
  u64 csum_lo = 0;
  u32 csum_hi = 0;
  __shared__ v4i32 st[64 * 8];                                  // one byte position's runs of the wave, 8 KiB
  const u32 nblk = k / 16;
  for (u32 jb0 = 0; jb0 < nblk; jb0 += 64) {                    // whole passes: every lane takes part in the staging
    const u32 jb = jb0 + lane;
    const bool on = jb < nblk;
    u64 cur[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) cur[p] = on ? y[(size_t)(16 * jb + p) * jstride] : 0;
#pragma unroll 1
    for (u32 a = 0; a < 7; ++a) {
      const u64 C = 0x8080808080808080ULL;
      u32 lo[16], hi[16];
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const u64 dgt = (cur[p] + C) ^ C;                       // the 8 balanced digits as bytes (vec_digits_kernel)
        lo[p] = (u32)dgt;
        hi[p] = (u32)(dgt >> 32);
        csum_lo += cur[p];
        csum_hi += csum_lo < cur[p];
        cur[p] = mulmod_shoup(cur[p], 256, w256p, m.q);
      }
      // digit b of the 16 elements = one 16-byte run: byte transposes of four elements at a time
      u32 run[8][4];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        transpose4x4_bytes(lo[4 * qd], lo[4 * qd + 1], lo[4 * qd + 2], lo[4 * qd + 3], run[0][qd], run[1][qd], run[2][qd], run[3][qd]);
        transpose4x4_bytes(hi[4 * qd], hi[4 * qd + 1], hi[4 * qd + 2], hi[4 * qd + 3], run[4][qd], run[5][qd], run[6][qd], run[7][qd]);
      }
      // this lane's 8 runs are 128 contiguous bytes of its tile (rows h*32 + 8 v4 .. + 7); through LDS so that lanes
      // 8x .. 8x+7 of one store instruction write them as one line (the piece index XORed against bank conflicts)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int b = 0; b < 8; ++b)
        st[lane * 8 + (b ^ (lane & 7))] = (v4i32){(int)run[b][0], (int)run[b][1], (int)run[b][2], (int)run[b][3]};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (u32 it = 0; it < 8; ++it) {
        const u32 sl = it * 8 + (lane >> 3), b = lane & 7, sjb = jb0 + sl;       // source lane, its block of 16 j
        const v4i32 val = st[sl * 8 + (b ^ (sl & 7))];
        if (sjb < nblk)
          *reinterpret_cast<v4i32*>(tiles + (size_t)((sjb >> 1) * 7 + a) * 1024 + (size_t)((sjb & 1) * 32 + v4 * 8 + b) * 16) = val;
      }
    }
  }
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

template <int ELL, int NVG, int RPW, int NCH = 0, bool FASTQ = false>
__global__ __launch_bounds__(256) void gemm_digits_kernel(GemmSection sa, GemmSection sb, const signed char* __restrict__ YD,
                                                           const int* __restrict__ SY, const Mod* __restrict__ mods,
                                                           u32 k, u32 L, u32 nv_total, u32 nv_pad, u32 vbn,
                                                           size_t yd_b16, size_t sy_b16, u32 kt) {
  // block = (limb, slot, group of 4*RPW row tiles); the 4 waves share the vector-digit tiles through
  // LDS (CJ j-blocks at a time); each wave owns RPW row tiles of 32 rows and streams their raw u64 tiles.
  constexpr int CJ = 8;                                    // j-blocks per staged chunk (32 MFMAs per wave per barrier)
  constexpr int BSH = NVG * CJ * 64 / 256;                 // 16-byte B elements each thread stages per chunk
  __shared__ v4i32 bl[2][NVG * CJ * 64];                   // two chunks of NVG*CJ KiB
  // NCH != 0: the launcher guarantees k == 4 * NCH * CJ, so every bounds test below folds away
  const u32 JB = NCH ? (u32)(NCH * CJ) : kt;               // K tiles (gemm_ktiles)
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
        an[r][u] = __builtin_nontemporal_load(ap[r] + (size_t)j * 64);
      }
  };
  auto fetch_b = [&](u32 jc, v4i32 (&bn)[BSH]) {
#pragma unroll
    for (int x = 0; x < BSH; ++x) {
      const u32 e = threadIdx.x + 256 * x;                  // element of the [NVG][CJ][64] chunk
      const u32 g = e / (CJ * 64), rem = e % (CJ * 64), u = rem / 64;
      const bool in = (jc + u) < JB;
      const u32 jb = in ? (jc + u) : jlast;
      const v4i32 val = ybase[((((size_t)g * L + limb) * ELL + slot) * JB + jb) * 64 + (rem & 63)];
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
          acc[r][g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[u & 1][g], ax, acc[r][g], 0, 0, 0);
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
template <int ELL, bool FASTQ>
__global__ __launch_bounds__(512, 2) void gemm_digits_wide_kernel(GemmSection sa, GemmSection sb, const signed char* __restrict__ YD,
                                                                         const Mod* __restrict__ mods, u32 k, u32 L, u32 nv_total,
                                                                         u32 nv_pad, u32 vbn, size_t yd_b16, u32 kt) {
  static_assert(PVW_GEMM_RPW == 1, "XM is padded to groups of four row tiles");
  constexpr int WRN = 4, NWV = 2 * WRN;                    // waves: WRN along the rows x 2 batches of 16 vectors
  constexpr int CJ = 2;                                    // j-blocks per stage
  constexpr int NB = 4;                                    // stage buffers (128 KiB: one workgroup per CU)
  constexpr int RTW = 2 * WRN, NG = 8;                     // row tiles / vector groups (of 4) per workgroup
  constexpr int STAGE = (RTW + NG) * CJ * 64;              // 16-byte elements per stage: 32 / 24 KiB
  constexpr int GPS = (RTW + NG) * CJ / NWV;               // LDS-DMA instructions per wave per stage: 4 / 6
  __shared__ v4i32 stage[NB * STAGE];                      // ONE array (a second __shared__ object next to LDS-DMA
                                                           // destinations makes hipcc drain the DMAs early)
  const u32 JB = kt, NST = JB / CJ;                        // K tiles (gemm_ktiles; even)
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
  auto issue_one = [&](u32 st, int x) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[x] + (size_t)st * CJ * 64),
                                     (__attribute__((address_space(3))) void*)&stage[(st % NB) * STAGE + (wave * GPS + x) * 64], 16, 0, 0);
  };
  // s_waitcnt vmcnt(n stages x GPS): the immediate must be a literal
  auto wait_stages = [&](u32 n) {
    if (n >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GPS) : "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(GPS == 4 && CJ == 2, "two j-blocks per stage, four DMAs per wave and stage; wait_stages knows 0, 1 and 2 stages");
#pragma unroll
  for (int i = 0; i < NB - 1; ++i)
    if ((u32)i < NST) issue(i);
  {
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
        acc[r][g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[2 + g], f[r], acc[r][g], 0, 0, 0);
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
template <int ELL, int VPB>
__global__ __launch_bounds__(32 * VPB) void gemm_finish_err_kernel(GemmSection sec, DevTables t, u32 L, u32 nv, u32 nv_pad, u32 rows_pad,
                                                               size_t ostride, const int* __restrict__ SY, size_t sy_b16, GemmErrSource es,
                                                               u32 v_lo, u32 v_hi) {
  // VPB vectors per block (x 32 rows): 8 with the limbs cut into gridDim.z ranges (every range repeats the sampling),
  // or 2 -- one wave -- with all limbs in one block, so that each polynomial is drawn ONCE (the launcher picks)
  constexpr int G = ELL / 2;                                     // lanes that share out one another's 8 l bytes (16 each)
  constexpr int CST = ELL + 2;                                   // LDS words per thread (16-byte aligned, bank spread)
  __shared__ i64 coef[32 * VPB * CST];
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
hipError_t launch_mftile(const u64* src, bool src_is_tiled, u64* XM, u32 rows, u32 k, u32 L, u32 ell, hipStream_t s, u32 bytes) {
  if (rows == 0) return hipSuccess;
  if (bytes == 7) {
    const u32 RT = ((rows + PVW_GEMM_ROWS_PER_WG - 1) / PVW_GEMM_ROWS_PER_WG) * (PVW_GEMM_ROWS_PER_WG / 32);
    const size_t threads = (size_t)RT * 32 * L * ell * (k / 32) * 2;
    PVW_DISPATCH_ELL(ell, mftile7_kernel<E><<<dim3((u32)((threads + 255) / 256)), dim3(256), 0, s>>>(src, src_is_tiled ? 1u : 0u, XM, rows, k, L));
    return hipGetLastError();
  }
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
                             const DevTables& t, hipStream_t s, size_t lstride, size_t jstride, u32 bytes) {
  if (nv == 0) return hipSuccess;
  if (lstride == 0 && jstride == 0) { lstride = (size_t)k * ell; jstride = ell; }
  // unused vector slots of the last group must read as zero digits / zero sums
  if (nv % 4) {
    const u32 NVG = (nv + 3) / 4, JB = gemm_ktiles(k, bytes);
    hipError_t e = hipMemsetAsync(YD + (size_t)(NVG - 1) * L * ell * JB * 1024, 0, (size_t)L * ell * JB * 1024, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(SY + (size_t)(NVG - 1) * L * ell * 32, 0, (size_t)L * ell * 32 * sizeof(int), s);
    if (e != hipSuccess) return e;
  }
  if (bytes == 7) {
    PVW_DISPATCH_ELL(ell, vec_digits7_kernel<E><<<dim3(nv * L * ell), dim3(64), 0, s>>>(vhat, vstride, YD, SY, nv, k, L, t, lstride, jstride));
    return hipGetLastError();
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
                              hipStream_t s, const GemmErrSource* es_a, const GemmErrSource* es_b, u32 bytes) {
  GemmSection sa = a, sb = b;
  const u32 kt = gemm_ktiles(k, bytes);
  sa.rt_groups = (sa.nrows + PVW_GEMM_ROWS_PER_WG - 1) / PVW_GEMM_ROWS_PER_WG;
  sb.rt_groups = (sb.nrows + PVW_GEMM_ROWS_PER_WG - 1) / PVW_GEMM_ROWS_PER_WG;
  const u32 blocks = (sa.rt_groups + sb.rt_groups) * L * ell;
  if (blocks == 0 || nv == 0) return hipSuccess;
  const u32 vbn = (nv + 15) / 16;
  const u32 NVG = vbn > 1 ? 4 : (nv + 3) / 4;
  const u32 nv_pad = NVG * 4;
  sa.tmp_bstride = (size_t)L * ell * 16 * sa.rt_groups * PVW_GEMM_ROWS_PER_WG;
  sb.tmp_bstride = (size_t)L * ell * 16 * sb.rt_groups * PVW_GEMM_ROWS_PER_WG;
  const size_t yd_b16 = (size_t)4 * L * ell * kt * 1024, sy_b16 = sy_bytes(16, L, ell) / sizeof(int);
#define PVW_GEMM_LAUNCH(G, N)                                                                                              \
  do {                                                                                                                    \
    if (t.min_q_bits >= 55) { PVW_DISPATCH_ELL(ell, gemm_digits_kernel<E, G, PVW_GEMM_RPW, N, true><<<dim3(blocks * vbn), dim3(256), 0, s>>>(sa, sb, YD, SY, t.mods, k, L, nv, nv_pad, vbn, yd_b16, sy_b16, kt)); } \
    else { PVW_DISPATCH_ELL(ell, gemm_digits_kernel<E, G, PVW_GEMM_RPW, N, false><<<dim3(blocks * vbn), dim3(256), 0, s>>>(sa, sb, YD, SY, t.mods, k, L, nv, nv_pad, vbn, yd_b16, sy_b16, kt)); } \
  } while (0)
#if PVW_TUNING
  // timing experiment (results wrong): all-zero operand bytes, to separate the schedule from the data-dependent power draw
  if (PVW_ENV_INT("PVW_GEMM_ZERO_OPERANDS", 0)) {
    if (sa.nrows) (void)hipMemsetAsync(const_cast<u64*>(sa.XM), 0, xm_words(sa.nrows, k, L, ell) * 8, s);
    if (sb.nrows) (void)hipMemsetAsync(const_cast<u64*>(sb.XM), 0, xm_words(sb.nrows, k, L, ell) * 8, s);
    (void)hipMemsetAsync(const_cast<signed char*>(YD), 0, yd_b16 * vbn, s);
  }
#endif
  // more than 16 vectors and whole stages of 16 terms: the wide form (256 rows x 32 vectors per workgroup of 8 waves in
  // ping-pong, both operands through LDS).  PVW_GEMM_WIDE=0 in the tuning build selects gemm_digits_kernel everywhere.
  // (8 waves in step and 4 waves x two workgroups per CU were the other forms measured: profiles/r02_gemm_wide.txt.)
  const bool wide = vbn >= 2 && k % 16 == 0 && k >= 16 && kt % 2 == 0 && PVW_ENV_INT("PVW_GEMM_WIDE", 1) != 0;
  if (wide) {
    const u32 ga = (sa.rt_groups * 4 + 7) / 8, gb2 = (sb.rt_groups * 4 + 7) / 8;
    const u32 wblocks = (ga + gb2) * L * ell * ((vbn + 1) / 2);
    if (t.min_q_bits >= 55) { PVW_DISPATCH_ELL(ell, gemm_digits_wide_kernel<E, true><<<dim3(wblocks), dim3(512), 0, s>>>(sa, sb, YD, t.mods, k, L, nv, nv_pad, vbn, yd_b16, kt)); }
    else { PVW_DISPATCH_ELL(ell, gemm_digits_wide_kernel<E, false><<<dim3(wblocks), dim3(512), 0, s>>>(sa, sb, YD, t.mods, k, L, nv, nv_pad, vbn, yd_b16, kt)); }
  } else {
  // fully unrolled chunk loops for the BASELINE geometries (k = 256: 8 chunks of 8 j-blocks, k = 512: 16), full vector groups
  static const int unroll_ok = (int)PVW_ENV_INT("PVW_GEMM_UNROLL", 1);
  if (NVG == 4 && unroll_ok && kt == 64) { PVW_GEMM_LAUNCH(4, 8); }
  else if (NVG == 4 && unroll_ok && kt == 128) { PVW_GEMM_LAUNCH(4, 16); }
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
        // 8 vectors per block and the limbs cut into ranges (each range repeats the sampling).  Tuning build, PVW_FINISH_VPB=2:
        // one wave per block (32 rows x 2 vectors) sweeping ALL limbs, so that every error polynomial is drawn ONCE -- measured
        // no better (64 dealers 0.946 vs 0.928 ms per step, key generation 4.17 vs 4.10 ms: profiles/r03_finish_ab.txt): the pass
        // is bound by the 0.6 GB it moves (intermediate in, output out), not by the repeated ChaCha blocks.
        const u32 gx = (sec.nrows + 31) / 32;
        bool one_wave = false;
#if PVW_TUNING
        one_wave = PVW_ENV_INT("PVW_FINISH_VPB", 8) == 2;
        if (one_wave) {
          const dim3 grid(gx, (span + 1) / 2, 1);
          switch (ell) {
            case 8: gemm_finish_err_kernel<8, 2><<<grid, dim3(64), 0, s>>>(sec, t, L, nv, nv_pad, rows_pad, ostride, SY, sy_b16, *es, v_lo, v_hi); break;
            case 16: gemm_finish_err_kernel<16, 2><<<grid, dim3(64), 0, s>>>(sec, t, L, nv, nv_pad, rows_pad, ostride, SY, sy_b16, *es, v_lo, v_hi); break;
            default: gemm_finish_err_kernel<32, 2><<<grid, dim3(64), 0, s>>>(sec, t, L, nv, nv_pad, rows_pad, ostride, SY, sy_b16, *es, v_lo, v_hi); break;
          }
        }
#endif
        if (!one_wave) {
          const u32 gy = (span + 7) / 8;
          u32 lz = 1;                                          // limb ranges: enough blocks for several rounds on the chip
          const size_t want = (size_t)PVW_ENV_INT("PVW_FINISH_BLOCKS", 4096);   // tuning build: blocks the limb split aims at (1024 .. 16384 measured: 4096)
          while (lz < L && (size_t)gx * gy * lz < want) lz *= 2;
          if (lz > L) lz = L;
          const dim3 grid(gx, gy, lz);
          switch (ell) {
            case 8: gemm_finish_err_kernel<8, 8><<<grid, dim3(256), 0, s>>>(sec, t, L, nv, nv_pad, rows_pad, ostride, SY, sy_b16, *es, v_lo, v_hi); break;
            case 16: gemm_finish_err_kernel<16, 8><<<grid, dim3(256), 0, s>>>(sec, t, L, nv, nv_pad, rows_pad, ostride, SY, sy_b16, *es, v_lo, v_hi); break;
            default: gemm_finish_err_kernel<32, 8><<<grid, dim3(256), 0, s>>>(sec, t, L, nv, nv_pad, rows_pad, ostride, SY, sy_b16, *es, v_lo, v_hi); break;
          }
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
hipError_t launch_mfma_probe(const signed char* A, const signed char* B, int* C, hipStream_t s) {
  mfma_i8_probe_kernel<<<dim3(1), dim3(64), 0, s>>>(A, B, C);
  return hipGetLastError();
}

}  // namespace pvw
