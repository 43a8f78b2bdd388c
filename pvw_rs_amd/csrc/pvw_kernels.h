// pvw_kernels.h -- launch interface of the gfx950 kernels (pvw_mac / pvw_poly / pvw_decrypt / pvw_decode_kernels / pvw_gemm .hip).
#pragma once
#include <hip/hip_runtime.h>

#include "pvw_arith.h"
#include "pvw_chacha.h"
#include "pvw_decode.h"

// PVW_TUNING selects the MEASUREMENT build (libpvw_hip_tuning.so, pvw_rs_amd/build.py): only there are the
// timing switches (PVW_DECODE_TIMING, PVW_GEMM_ZERO_OPERANDS), the schedule selectors (PVW_MAC_VARIANT, PVW_MAC_PACKED,
// PVW_DEC_C, PVW_DECODE_VARIANT, ...) and the read-bandwidth probe compiled in.  The shipped library (PVW_TUNING 0) holds the
// shape-selected schedules only and reads NO environment variable: nothing outside the arguments of a call can
// change what it computes (the reference samples unconditionally, src/crypto/encryption.rs:135-167).
#ifndef PVW_TUNING
#define PVW_TUNING 0
#endif
#if PVW_TUNING
#include <cstdlib>
#define PVW_ENV_INT(name, dflt) ([]() -> long { const char* e_ = getenv(name); return e_ ? atol(e_) : (long)(dflt); }())
#else
#define PVW_ENV_INT(name, dflt) ((long)(dflt))
#endif

namespace pvw {

enum { DOM_R = 0, DOM_E1 = 1, DOM_E2 = 2, DOM_SK = 3, DOM_EKEY = 4, DOM_CRS = 5, DOM_GAUSS = 6, DOM_PK = 7 };

// per-context device tables (all device pointers)
struct DevTables {
  const Mod* mods;   // [L]
  const u64* tw;     // [L][l]  psi^bitrev(i)
  const u64* itw;    // [L][l]  psi^-bitrev(i)
  const u64* linv;   // [L]     l^-1 mod q
  const u64* ghat;   // [L][l]  NTT(gadget)              (parameters.rs:288-308)
  const u64* gpow;   // [L][l]  gadget residues D^j mod q (power basis)
  // Shoup companions floor(x * 2^64 / q) of the five tables above
  const u64* twp;
  const u64* itwp;
  const u64* linvp;
  const u64* ghatp;
  const u64* gpowp;
  u32 min_q_bits;    // bit length of the smallest modulus (kernels with a fast path for wide moduli test it)
  u32 max_q_bits;    // ... of the widest one (packed stream width; byte count of the digit GEMM)
};

enum { SAMPLE_CBD = 0, SAMPLE_UNIFORM = 1 };
struct SampleJob {
  u32 kind;       // SAMPLE_CBD | SAMPLE_UNIFORM
  u32 domain;     // ChaCha stream domain
  u32 index0;     // first polynomial index (stream id low word)
  u32 count;      // polynomials
  u32 out_poly0;  // first output polynomial slot
  u32 cbd_half;   // variance == 0.5 special case
  u32 cbd_v;      // (usize)variance otherwise
  u64 bound;      // uniform bound
};

// one section of the streamed tiled matrix: out[row] = sum_j M[row][j]*rhat[j] + addend[row]
// (addend may alias out; NULL = none).  row_blocks is filled in by the launcher.
struct MacSection {
  const u64* M;
  const u64* addend;
  u64* out;
  u32 nrows;
  u32 row_blocks;
  // l <= 16: the addend of row i in COMPACT form instead of `addend` -- the l small coefficients of its error polynomial
  // (e_small[i][l], as sampled or as the caller supplied them) and, for c2 rows, the scalar m_i (scalars[i]; NULL: none).
  // The kernel transforms them itself (NTT(e_i) + m_i g-hat for its limb: encryption.rs:161-167, :195-196) -- 8 l + 8
  // bytes per row cross memory instead of 8 L l written by the prologue and read back here.
  const i64* e_small = nullptr;
  const u64* scalars = nullptr;
};
// group != 0: polynomial p goes to out + (p / group) * stride_group + (p % group) * stride_poly
hipError_t launch_prep(const i64* coeffs, const u64* scalars, u64* out, size_t stride_poly,
                       size_t stride_limb, u32 count, bool do_ntt, const DevTables& t, u32 L,
                       u32 ell, hipStream_t s, u32 group = 0, size_t stride_group = 0);
// dst[c][j] = src[j][c] for a k x k matrix of polynomials of `words` u64 each
hipError_t launch_transpose_polys(const u64* src, u64* dst, u32 k, u32 words, hipStream_t s);
hipError_t launch_ntt(u64* polys, size_t count, bool inverse, const DevTables& t, u32 L, u32 ell,
                      hipStream_t s);
hipError_t launch_tile(const u64* src, u64* M, u32 rows, u32 row0_tiled, u32 k, u32 L, u32 ell,
                       bool ntt_first, const DevTables& t, hipStream_t s);
hipError_t launch_untile(const u64* M, u64* dst, u32 rows, u32 row0_tiled, u32 k, u32 L, u32 ell,
                         bool intt_after, const DevTables& t, hipStream_t s);
hipError_t launch_fill_uniform_tiled(u64* M, const ChaChaKey& key, u32 domain, u32 rows,
                                     u32 row0_tiled, u32 grow0, u32 k, u32 L, u32 ell,
                                     const DevTables& t, hipStream_t s);
// one family of small polynomials of the encrypt prologue: sampled (explicit_coeffs == NULL) or
// supplied, reduced + transformed into out[p*stride_poly + limb*stride_limb + slot],
// optionally with scalars[p] * g-hat added (encode_scalar, parameters.rs:346-367)
struct PrologueJob {
  SampleJob sj;
  const i64* explicit_coeffs;
  const u64* scalars;
  u64* out;
  size_t stride_poly, stride_limb;
  u32 key_idx;   // which key of the batch seeds this family (replica r uses key_idx + r * rep_key)
  // replication (PrologueBatch::reps > 1): what replica r adds to the fields above
  u32 rep_key;        // 0: every replica shares the key, 1: one key per replica
  u32 rep_index0;     // stream index offset per replica
  size_t rep_out, rep_scalars, rep_coeffs;   // element offsets per replica
  // not NULL: the family is only SAMPLED -- polynomial p's l coefficients go to raw_out[p * l ..] as they are (no reduction, no
  // transform; `out`, `scalars` and the strides are not used).  Single replica only.
  i64* raw_out;
};
#define PVW_MAX_PROLOGUE_JOBS 8
#define PVW_MAX_PROLOGUE_KEYS 64
// The polynomial families of ONE encrypt (r, e1, e2) or ONE key generation (s, e), replicated `reps` times
// with regular strides (up to 64 dealers / parties per launch); the whole batch travels in the
// kernel-argument segment (< 4 KiB)
struct PrologueBatch {   // sizeof must stay below the 4 KiB kernel-argument limit
  PrologueJob job[PVW_MAX_PROLOGUE_JOBS];
  ChaChaKey key[PVW_MAX_PROLOGUE_KEYS];
  u32 njobs;
  u32 reps;    // 0 is read as 1
  u32 total;   // filled in by the launcher (polynomials per replica)
  u32 key_window, key_rep;   // filled in by the launcher: replica r reads keys [r * key_rep, r * key_rep + key_window)
};
hipError_t launch_prologue(const PrologueBatch& batch, const DevTables& t, u32 L, u32 ell, hipStream_t s);

// c1 (section a: A-hat rows) and c2 (section b: B-hat rows) in a single launch
hipError_t launch_mac_rows(const MacSection& a, const MacSection& b, const u64* rhat, const DevTables& t, u32 k, u32 L, u32 ell,
                           hipStream_t s);
// the same over the PACKED copy of the sections (`width` bits per residue; MacSection::M = the packed copy).
// packed_width: the stream width for a modulus chain whose widest modulus has max_q_bits bits -- 40 / 48 / 56 (k a
// multiple of 64) or 61 (k a multiple of 256), l <= 16 -- or 0 when the geometry does not qualify.  launch_pack
// builds the copy from the tiled matrix: packed_words(rows, ..., width) u64 per section.
u32 packed_width(u32 max_q_bits, u32 k, u32 ell);
hipError_t launch_mac_rows_packed(const MacSection& a, const MacSection& b, const u64* rhat, const DevTables& t, u32 k, u32 L,
                                  u32 ell, u32 width, hipStream_t s);
// *wide_flag (device word, zeroed by the caller) is set when a matrix word does not fit `width` bits: the copy is then unusable
hipError_t launch_pack(const u64* M, u64* P, u32 rows, u32 k, u32 L, u32 ell, u32 width, u32* wide_flag, hipStream_t s);
inline size_t packed_words(u32 rows, u32 k, u32 L, u32 ell, u32 width) {
  const u32 R = 128 / ell;
  return (size_t)((rows + R - 1) / R) * L * (k / 64 * width) * 128;
}
// NV (<= 4) vectors sharing one pass over the tiled matrix (mac_rows_multi)
struct MultiVec {
  const u64* vhat;      // [nv][L][k][l]
  size_t vstride;       // words between consecutive vectors
  size_t out_stride_a;  // words between consecutive vectors' outputs / addends, section a
  size_t out_stride_b;  // ... section b
  u32 nv;
};
hipError_t launch_mac_rows_multi(const MacSection& a, const MacSection& b, const MultiVec& mv,
                                 const DevTables& t, u32 k, u32 L, u32 ell, hipStream_t s);
hipError_t launch_sample(i64* out, const ChaChaKey& key, u32 ell, const SampleJob& j0,
                         const SampleJob& j1, const SampleJob& j2, hipStream_t s);
hipError_t launch_gaussian(i64* out, const ChaChaKey& key, u32 index0, u32 count, u64 bound,
                           hipStream_t s);
// nsplit > 1 (from decrypt_split; `partial` then holds nsplit x dealers polynomials): the k terms are cut into nsplit
// ranges whose sums go to `partial`; launch_decrypt_finish adds them up, subtracts c2 and transforms back.
// nsplit <= 1: noisy = <s-hat, c1> - c2 in the NTT domain, as before (follow with launch_ntt(inverse)).
// alone = false: another kernel (the decode of the previous chunk) is meant to share the CUs with this launch: the
// register-lean instance is used (pvw_decrypt.hip).
u32 decrypt_split(u32 k, u32 L, u32 ell, size_t dealers);
hipError_t launch_decrypt_mac(const u64* c1s, const u64* shat, const u64* c2col, u64* noisy,
                              const DevTables& t, u32 k, u32 L, u32 ell, size_t dealers,
                              hipStream_t s, u64* partial = nullptr, u32 nsplit = 1, bool alone = true);
hipError_t launch_decrypt_finish(const u64* partial, u32 nsplit, const u64* c2col, u64* noisy, const DevTables& t, u32 L, u32 ell,
                                 size_t dealers, hipStream_t s);

// ---- digit GEMM on the matrix cores (see pvw_gemm.hip) ----
#ifndef PVW_GEMM_RPW
#define PVW_GEMM_RPW 1                                   // row tiles (of 32 rows) per wave
#endif
#define PVW_GEMM_ROWS_PER_WG (4 * PVW_GEMM_RPW * 32)     // 4 waves per workgroup
// XM = MFMA-tiled copy of a matrix section: [limb][slot][row tile of 32][j block of 4][64 lanes][2 u64],
// row tiles padded to whole workgroups (4 waves x PVW_GEMM_RPW row tiles).
struct GemmSection {
  const u64* XM;
  const u64* addend;   // per-vector planes, same indexing as out (may alias out; NULL = none)
  u64* out;
  u64* tmp;            // intermediate [limb][slot][16 vectors][rows padded to whole workgroups]
  u32 nrows;
  u32 rt_groups;       // filled in by the launcher
  size_t tmp_bstride;  // words of `tmp` per batch of 16 vectors (filled in by the launcher)
  // key generation: the finish pass writes vector v, GEMM row c straight into the TILED public-key matrix as
  // element (party tiled_row0 + v, column c) instead of out[] (which is then only the addend); NULL = API layout
  u64* tiled_out = nullptr;
  u32 tiled_row0 = 0;
  // tiled_swap: the GEMM ROW is the party (tiled_row0 + row) and the VECTOR the column instead
  u32 tiled_swap = 0;
  // element stride between consecutive GEMM rows in out / addend (0 = one polynomial, L * l)
  size_t row_stride = 0;
};
inline size_t gemm_tmp_words(u32 rows, u32 L, u32 ell) {
  return (size_t)L * ell * 16 * (((rows + PVW_GEMM_ROWS_PER_WG - 1) / PVW_GEMM_ROWS_PER_WG) * PVW_GEMM_ROWS_PER_WG);
}
inline size_t xm_words(u32 rows, u32 k, u32 L, u32 ell) {
  return (size_t)L * ell * (((rows + PVW_GEMM_ROWS_PER_WG - 1) / PVW_GEMM_ROWS_PER_WG) * (PVW_GEMM_ROWS_PER_WG / 32)) *
         ((k + 3) / 4) * 128;
}
inline size_t yd_bytes(u32 nv, u32 k, u32 L, u32 ell) { return (size_t)((nv + 3) / 4) * L * ell * ((k + 3) / 4) * 1024; }
// K tiles (32 contraction rows each) of the digit GEMM.  bytes = 8: a tile is 4 consecutive j x the 8 bytes of the matrix
// element.  bytes = 7 (every modulus below 2^56, k a multiple of 64): byte 7 of every element is zero and is left out of
// the contraction -- a tile is one byte position a < 7 of 32 consecutive j, 7 tiles per 32 j instead of 8 (gemm7_ok).
inline u32 gemm_ktiles(u32 k, u32 bytes) { return bytes == 7 ? 7 * (k / 32) : (k + 3) / 4; }
inline bool gemm7_ok(u32 max_q_bits, u32 k) { return max_q_bits <= 56 && k % 64 == 0 && k >= 64; }
inline size_t sy_bytes(u32 nv, u32 L, u32 ell) { return (size_t)((nv + 3) / 4) * L * ell * 32 * sizeof(int); }
hipError_t launch_mftile(const u64* src, bool src_is_tiled, u64* XM, u32 rows, u32 k, u32 L, u32 ell, hipStream_t s, u32 bytes = 8);
// small coefficients [row][j][l] -> NTT -> MFMA-tiled raw operand in one pass (l <= 32); padding included
hipError_t launch_shat_mftile(const i64* coeffs, u64* XM, u32 rows, u32 k, u32 L, u32 ell, const DevTables& t, hipStream_t s);
// element j of vector v at (limb, slot): vhat[v * vstride + limb * lstride + j * jstride + slot];
// lstride = jstride = 0 selects the r-hat layout [limb][j][slot] (lstride = k * l, jstride = l)
hipError_t launch_vec_digits(const u64* vhat, size_t vstride, signed char* YD, int* SY, u32 nv, u32 k, u32 L, u32 ell,
                             const DevTables& t, hipStream_t s, size_t lstride = 0, size_t jstride = 0, u32 bytes = 8);
// nv may exceed 16: batches of 16 vectors then run as extra workgroups of ONE launch (adjacent in dispatch
// order, so they share the streamed matrix tiles through L2); tmp must hold ceil(nv/16) batches.
// es_a / es_b != NULL (l <= 32): that section's finish pass adds an error term it draws or reads itself, plus the
// encoded scalar, instead of an addend from memory (gemm_finish_err_kernel).  (row, v) = (GEMM row, vector).
// The pointer is to an ARRAY: element i describes the next `span` vectors (its keys travel as kernel arguments,
// 64 at most); span == 0 covers all that are left.
struct GemmErrSource {
  u32 span;
  const i64* explicit_coeffs;       // small coefficients of (row, v) at (row * coef_row + v * coef_v) * l, or NULL: drawn
  size_t coef_row, coef_v;
  ChaChaKey key[PVW_MAX_PROLOGUE_KEYS];   // key of vector v: key[(v - first vector of this element) * key_v]
  u32 key_v;
  u32 domain, index0, index_row, index_v;   // ChaCha stream of (row, v): index0 + row * index_row + v * index_v
  u64 bound;                        // uniform in [-bound, bound]
  const u64* scalars;               // NULL, or m of (row, v) at scalars[v * scalar_v + row]: + m g-hat (encode_scalar)
  size_t scalar_v;
};
hipError_t launch_gemm_digits(const GemmSection& a, const GemmSection& b, const signed char* YD, const int* SY,
                              const DevTables& t, u32 k, u32 L, u32 ell, u32 nv, size_t ostride_a, size_t ostride_b,
                              hipStream_t s, const GemmErrSource* es_a = nullptr, const GemmErrSource* es_b = nullptr, u32 bytes = 8);
// read-only probe: every wave streams `tiles` consecutive 1-KiB tiles (16 in flight), grid as mac_rows
#if PVW_TUNING
// time stamps (100 MHz ticks, [2b] start / [2b+1] end) and HW_ID words of the workgroups of the last stamped mac_rows launch
hipError_t read_stamps(u64* out, u32* hw, u32 count);
hipError_t init_probe_attributes();
hipError_t launch_read_probe(const u64* M, size_t total_tiles, u32 tiles_per_wave, u64* sink, hipStream_t s);
hipError_t launch_read_probe2(const u64* M, size_t total_tiles, u32 tiles_per_wave, u64* sink, u32 U, bool dbuf, u32 lds_bytes,
                              hipStream_t s, u32 xmap = 0);
#endif
// per-device kernel attributes (dynamic-LDS limits); call once per context after hipSetDevice
hipError_t init_kernel_attributes();
hipError_t launch_mfma_probe(const signed char* A, const signed char* B, int* C, hipStream_t s);
// xf == nullptr: noisy holds power-basis polynomials (read only).  xf != nullptr: noisy holds them in the NTT domain; the
// decode transforms each ciphertext back as it stages it and stores the power-basis polynomial over it (the inverse
// transform of decrypt, decryption.rs:116, without a launch of its own).
// wipe / wipe_bytes (a multiple of 16): a region the decode launch clears as well (NTT(sk) of the decrypt it closes; every
// launch in front of it on `s` must be done with it); *wiped says whether this launch took that on (the fixed-width
// fallback does not).
hipError_t launch_decode(u64* noisy, u64* out, size_t count, const DecodeTables& t, hipStream_t s, const DevTables* xf = nullptr,
                         u64* wipe = nullptr, size_t wipe_bytes = 0, bool* wiped = nullptr);

}  // namespace pvw
