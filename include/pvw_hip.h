/* pvw_hip.h -- C ABI of the MI355X-native PVW multi-receiver encrypt/decrypt path.
 *
 * This is the drop-in boundary for the hot path of gnosisguild/pvw-rs: a Rust
 * host that keeps the `pvw::{params,crs,keys,crypto}` API binds these symbols
 * (see INTEGRATION.md for the `extern "C"` block) and every bulk polynomial
 * operation becomes one call into hand-written HIP for gfx950.  The reference
 * has no FFI of its own (no `extern`, no `unsafe`); each entry point below
 * cites the reference routine (file:line under the reference checkout) whose
 * work it replaces.
 *
 * Conventions
 *   - plain pointers and sizes; every pointer is caller-owned unless returned
 *     by a *_create; no C++/torch types.
 *   - every function returns int32_t: 0 = PVW_OK, otherwise one of the
 *     PVW_ERR_* codes, which map 1:1 onto the PvwError variants of
 *     src/errors.rs:13-70.  pvw_last_error() returns the thread-local message.
 *   - polynomial = [L][l] uint64_t, limb-major (the Array2<u64> (num_moduli,
 *     degree) of src/params/parameters.rs:433-458), residues in [0, q_i).
 *     Matrices are row-major arrays of polynomials: A is [k][k], B is [n][k].
 *   - `repr`: PVW_REPR_POWER = coefficients (fhe-math Representation::PowerBasis),
 *     PVW_REPR_NTT = this library's NTT domain (slot s of limb i holds the
 *     evaluation at psi_i^(2*bitrev(s)+1); psi_i from pvw_ctx_get_roots).  The
 *     NTT-domain layout of fhe-math is not pinned by the reference's tests, so
 *     data exchanged with an fhe-math host should cross in PVW_REPR_POWER unless
 *     pvw_ctx_set_roots() has been given fhe-math's roots.
 *   - host-buffer calls are synchronous (results are in the buffers on return)
 *     and safe to call concurrently on one context, as rayon does with
 *     `encrypt` (src/crypto/encryption.rs:277-283): device tensors are
 *     read-only after load and every call takes a stream + workspace from a pool.
 *   - *_device calls take device pointers and a hipStream_t (as void*; NULL =
 *     the context's own stream) and enqueue asynchronously.  They do not
 *     synchronise or allocate ONCE pvw_prepare() has run for that stream since
 *     the matrices last changed.  Without pvw_prepare() the first call on a
 *     stream allocates that stream's workspace, and the first encrypt after a
 *     CRS / public-key change builds the derived copies of the matrices it
 *     streams from (a bit-packed copy for pvw_encrypt_device, an MFMA-tiled copy
 *     for pvw_encrypt_multi_device): that call allocates up to a second copy of
 *     the resident matrices and waits for the build.  A call made while its
 *     stream is being captured into a graph never builds anything: it uses the
 *     copies that are valid (pvw_prepare first) or the plain tiled matrices.
 *     The context's own stream is created non-blocking: it is NOT ordered against
 *     the legacy default stream, so a caller that prepares or consumes the buffers
 *     on the default stream (stream 0 -- also what a framework's "current stream"
 *     usually is) must either pass a stream of its own or synchronise the device
 *     between its work and the call (pvw_ctx_synchronize waits for the context's
 *     stream).
 */
#ifndef PVW_HIP_H
#define PVW_HIP_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define PVW_API __attribute__((visibility("default")))
#else
#define PVW_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pvw_ctx pvw_ctx;

/* ---- status codes <-> PvwError (src/errors.rs:13-70), in declaration order ---- */
enum {
  PVW_OK = 0,
  PVW_ERR_INVALID_PARAMETERS = 1,  /* errors.rs:15 */
  PVW_ERR_SAMPLING = 2,            /* :18 */
  PVW_ERR_ENCRYPTION = 3,          /* :21 */
  PVW_ERR_DECRYPTION = 4,          /* :24 */
  PVW_ERR_KEY_GENERATION = 5,      /* :27 */
  PVW_ERR_CRS = 6,                 /* :30 */
  PVW_ERR_SERIALIZATION = 7,       /* :33 */
  PVW_ERR_DESERIALIZATION = 8,     /* :36 */
  PVW_ERR_ENCODING = 9,            /* :39 */
  PVW_ERR_DECODING = 10,           /* :42 */
  PVW_ERR_VALIDATION = 11,         /* :45 */
  PVW_ERR_CONTEXT = 12,            /* :48 */
  PVW_ERR_POLYNOMIAL = 13,         /* :51 */
  PVW_ERR_MATRIX = 14,             /* :54 */
  PVW_ERR_DIMENSION_MISMATCH = 15, /* :57  {expected, actual} in the message */
  PVW_ERR_INDEX_OUT_OF_BOUNDS = 16,/* :60  {index, bound} in the message */
  PVW_ERR_INSUFFICIENT_DATA = 17,  /* :63 */
  PVW_ERR_INVALID_FORMAT = 18,     /* :66 */
  PVW_ERR_INTERNAL = 19            /* :69  also: HIP runtime failures, no device */
};

enum { PVW_REPR_POWER = 0, PVW_REPR_NTT = 1 };
enum { PVW_RND_SEED = 0, PVW_RND_EXPLICIT = 1 };

/* ChaCha8 stream-id domains of the counter-based sampler: stream = (domain<<32)|poly index */
enum {
  PVW_DOM_R = 0, PVW_DOM_E1 = 1, PVW_DOM_E2 = 2, PVW_DOM_SK = 3, PVW_DOM_EKEY = 4,
  PVW_DOM_CRS = 5, PVW_DOM_GAUSS = 6, PVW_DOM_PK = 7
};

/* PvwParametersBuilder fields (src/params/parameters.rs:44-52).  The builder's
 * defaults (variance 0.5, bounds 100/200, :166-168) are applied by the host
 * mirror; this struct always carries explicit values. */
typedef struct {
  uint32_t n;               /* set_parties   :61  (global party count)            */
  uint32_t k;               /* set_dimension :67                                  */
  uint32_t l;               /* set_l         :73  ring degree, power of two >= 8  */
  uint32_t num_moduli;      /* set_moduli    :79                                  */
  const uint64_t* moduli;
  float secret_variance;    /* set_secret_variance :85                            */
  uint64_t error_bound_1;   /* set_error_bound_1   :91  (fits u64; reference is BigInt) */
  uint64_t error_bound_2;   /* set_error_bound_2   :97                            */
  int32_t device;           /* HIP device ordinal, -1 = current device            */
  /* party shard held by this context (one process per GPU): rows [party_lo, party_hi)
   * of B / c2 and rows [c1_lo, c1_hi) of A / c1.  All zero = everything.        */
  uint32_t party_lo, party_hi;
  uint32_t c1_lo, c1_hi;
} pvw_params_t;

/* Randomness of one encrypt call.  The reference draws from thread_rng() inside
 * rayon closures (encryption.rs:138,164,180) and cannot be replayed; this ABI
 * makes the randomness an input.  SEED: r ~ CBD(secret_variance), e1/e2 uniform
 * in [-bound, bound], each polynomial from its own ChaCha8 stream.  EXPLICIT:
 * small signed coefficients supplied by the caller. */
typedef struct {
  uint32_t mode;            /* PVW_RND_SEED | PVW_RND_EXPLICIT */
  uint8_t seed[32];
  const int64_t* r;         /* [k][l] */
  const int64_t* e1;        /* [k][l] */
  const int64_t* e2;        /* [n][l]  (global n; a sharded context reads its rows) */
} pvw_randomness_t;

/* message of the calling thread's last failure (NUL-terminated, truncated to len) */
PVW_API int32_t pvw_last_error(char* buf, size_t len);
/* 1 if a gfx950 device is usable from this process, 0 otherwise (never fails) */
PVW_API int32_t pvw_device_available(void);

/* ---- parameters: PvwParametersBuilder::build (parameters.rs:117-195) ------------
 * Validation mirrors :131-181 (n>0, k>0, l power of two >= 8, bounds > 0) plus what
 * the reference delegates to fhe-math Context::new_arc (:147): moduli distinct odd
 * primes < 2^62 with q = 1 (mod 2l). */
PVW_API int32_t pvw_ctx_create(const pvw_params_t* params, pvw_ctx** out);
PVW_API int32_t pvw_ctx_destroy(pvw_ctx* ctx);
/* psi_i (primitive 2l-th root per limb).  Default: the smallest one.  set_roots must
 * precede any load/keygen call; it lets a host align this NTT domain with another library's. */
PVW_API int32_t pvw_ctx_get_roots(const pvw_ctx* ctx, uint64_t* psi_out /*[L]*/);
PVW_API int32_t pvw_ctx_set_roots(pvw_ctx* ctx, const uint64_t* psi /*[L]*/);
/* big integers as little-endian 64-bit words; *nwords receives the count (cap = capacity) */
PVW_API int32_t pvw_ctx_delta(const pvw_ctx* ctx, uint64_t* words, size_t cap, size_t* nwords);            /* delta()  :370 */
PVW_API int32_t pvw_ctx_delta_power_l_minus_1(const pvw_ctx* ctx, uint64_t* words, size_t cap, size_t* nwords); /* :375 */
PVW_API int32_t pvw_ctx_q_total(const pvw_ctx* ctx, uint64_t* words, size_t cap, size_t* nwords);          /* q_total() :380 */
/* gadget_polynomial (parameters.rs:288-308): [1, D, ..., D^(l-1)] as one polynomial */
PVW_API int32_t pvw_ctx_gadget(const pvw_ctx* ctx, uint64_t* poly_out /*[L][l]*/, uint32_t repr);
/* verify_correctness_condition (parameters.rs:510-551) */
PVW_API int32_t pvw_ctx_verify_correctness_condition(const pvw_ctx* ctx, int32_t* ok_out);
/* suggest_error_bounds (parameters.rs:554-603) */
PVW_API int32_t pvw_suggest_error_bounds(uint32_t n, uint32_t k, uint32_t l, const uint64_t* moduli,
                                 uint32_t num_moduli, float variance, uint32_t* bound1_out,
                                 uint32_t* bound2_out);
/* encode_scalar (parameters.rs:346-367): scalar * gadget as one polynomial */
PVW_API int32_t pvw_encode_scalar(const pvw_ctx* ctx, int64_t scalar, uint64_t* poly_out, uint32_t repr);

/* ---- CRS: PvwCrs.matrix (src/params/crs.rs:12-17) --------------------------------
 * a: host [k][k][L][l] in `repr`.  A sharded context keeps rows [c1_lo, c1_hi). */
PVW_API int32_t pvw_load_crs(pvw_ctx* ctx, const uint64_t* a, uint32_t repr);
PVW_API int32_t pvw_load_crs_device(pvw_ctx* ctx, const uint64_t* d_a, uint32_t repr, void* stream);
/* PvwCrs::new_deterministic analogue (crs.rs:45-67): uniform NTT-domain polynomials from a
 * 32-byte seed with this library's ChaCha8 streams (PVW_DOM_CRS) -- not fhe-math's bytes. */
PVW_API int32_t pvw_crs_generate(pvw_ctx* ctx, const uint8_t seed[32]);
/* PvwCrs::new_from_tag (crs.rs:74-90): the 32-byte seed the reference derives from a string tag -- the 64-bit
 * std DefaultHasher (SipHash-1-3, zero key) of tag + "CRS", little-endian, repeated four times.  Feed it to
 * pvw_crs_generate.  (PvwCrs::new, crs.rs:24-39, is pvw_crs_generate with a seed from the host's entropy source.) */
PVW_API int32_t pvw_crs_seed_from_tag(const char* tag, uint8_t seed_out[32]);
/* download: a_out host [k][k][L][l] (only rows held by this context are written) */
PVW_API int32_t pvw_get_crs(pvw_ctx* ctx, uint64_t* a_out, uint32_t repr);

/* ---- global public key: GlobalPublicKey.matrix (src/keys/public_key.rs:43-54) ----
 * add_public_key (:214-250) for parties [party_lo, party_hi): b host [count][k][L][l].
 * Bookkeeping as :245: num_keys = max(num_keys, party_hi). */
PVW_API int32_t pvw_load_pk(pvw_ctx* ctx, uint32_t party_lo, uint32_t party_hi, const uint64_t* b,
                    uint32_t repr);
PVW_API int32_t pvw_load_pk_device(pvw_ctx* ctx, uint32_t party_lo, uint32_t party_hi,
                           const uint64_t* d_b, uint32_t repr, void* stream);
/* synthetic uniform B-hat (benchmarks; statistically what Poly::random gives, crs.rs:32) */
PVW_API int32_t pvw_pk_fill_uniform(pvw_ctx* ctx, const uint8_t seed[32]);
PVW_API int32_t pvw_get_pk(pvw_ctx* ctx, uint32_t party_lo, uint32_t party_hi, uint64_t* b_out,
                   uint32_t repr);
PVW_API int32_t pvw_num_public_keys(const pvw_ctx* ctx, uint32_t* out);   /* num_public_keys :344 */
PVW_API int32_t pvw_is_full(const pvw_ctx* ctx, int32_t* out);            /* is_full :349 */

/* ---- key generation: PublicKey::generate (public_key.rs:111-147) over
 * PvwCrs::multiply_by_secret_key (crs.rs:138-171), batched as generate_all_keys
 * (public_key.rs:407-434).  b_i = s_i * A + e_i for parties [party_lo, party_hi).
 * sk: [count][k][l] CBD coefficients (SecretKey.secret_coeffs, secret_key.rs:14-18).
 * ek: [count][k][l] explicit key errors, or NULL to sample uniform[-bound1, bound1]
 * from `seed` (PVW_DOM_EKEY).  Result is stored as rows of B on the device. */
PVW_API int32_t pvw_keygen(pvw_ctx* ctx, uint32_t party_lo, uint32_t party_hi, const int64_t* sk,
                   const int64_t* ek, const uint8_t seed[32]);
/* SecretKey::random (secret_key.rs:45-63) for `count` parties from a seed (PVW_DOM_SK) */
PVW_API int32_t pvw_sample_secret_keys(const pvw_ctx* ctx, const uint8_t seed[32], uint32_t party_lo,
                               uint32_t count, int64_t* sk_out /*[count][k][l]*/);

/* ---- encrypt (src/crypto/encryption.rs:105-214) -----------------------------------
 * Checks mirror :109 (scalar count), :117 (key fullness), :124 (correctness gate).
 * scalars: n values (global).  c1_out: [k][L][l], c2_out: [n][L][l]; a sharded context
 * writes only its rows [c1_lo,c1_hi) / [party_lo,party_hi) at their global positions.
 * `scalars[i] as i64` wraps as the reference does (:195). */
PVW_API int32_t pvw_encrypt(pvw_ctx* ctx, const uint64_t* scalars, size_t num_scalars,
                    const pvw_randomness_t* rnd, uint64_t* c1_out, uint64_t* c2_out,
                    uint32_t out_repr);
/* device-resident variant: d_scalars [n]; d_c1 [c1 rows held][L][l]; d_c2 [parties held][L][l]
 * (LOCAL row numbering); explicit randomness pointers, if used, are device pointers too. */
PVW_API int32_t pvw_encrypt_device(pvw_ctx* ctx, const uint64_t* d_scalars, size_t num_scalars,
                           const pvw_randomness_t* rnd, uint64_t* d_c1, uint64_t* d_c2,
                           uint32_t out_repr, void* stream);

/* ---- multi-dealer encrypt: encrypt_all_party_shares (src/crypto/encryption.rs:253-286) ----
 * Dealer d encrypts scalars[d][0..n) with its own randomness (seeds + 32*d, PVW_RND_SEED
 * semantics).  Dealers are processed four at a time against ONE pass over A-hat / B-hat, so
 * the public key is streamed D/4 times instead of D times.
 * scalars [D][n]; c1_out [D][k][L][l]; c2_out [D][n][L][l].  scalars_per_dealer must be n (:264-274). */
PVW_API int32_t pvw_encrypt_multi(pvw_ctx* ctx, const uint64_t* scalars, size_t num_dealers,
                                  size_t scalars_per_dealer, const uint8_t* seeds /*[D][32]*/,
                                  uint64_t* c1_out, uint64_t* c2_out, uint32_t out_repr);
/* device-resident variant: d_scalars [D][n]; d_c1 [D][c1 rows held][L][l]; d_c2 [D][parties held][L][l];
 * seeds stays a HOST pointer */
PVW_API int32_t pvw_encrypt_multi_device(pvw_ctx* ctx, const uint64_t* d_scalars, size_t num_dealers,
                                         size_t scalars_per_dealer, const uint8_t* seeds,
                                         uint64_t* d_c1, uint64_t* d_c2, uint32_t out_repr, void* stream);

/* ---- decrypt (src/crypto/decryption.rs:249-325) ------------------------------------
 * One secret key against D dealer ciphertexts (decrypt_party_shares :281-325):
 *   noisy_d = sum_j NTT(sk[j]) * c1s[d][j] - c2col[d]      (:257-274)
 *   out[d]  = decode_scalar_pvw_rns(noisy_d)               (:10-58)
 * sk [k][l]; c1s [D][k][L][l] and c2col [D][L][l] in `in_repr`; out_u64 [D];
 * noisy_out optional [D][L][l], power basis. */
PVW_API int32_t pvw_decrypt_batch(pvw_ctx* ctx, const int64_t* sk, const uint64_t* c1s,
                          const uint64_t* c2col, size_t num_dealers, uint32_t in_repr,
                          uint64_t* out_u64, uint64_t* noisy_out);
/* device-resident first half: d_noisy [D][L][l] power basis */
PVW_API int32_t pvw_decrypt_noisy_device(pvw_ctx* ctx, const int64_t* d_sk, const uint64_t* d_c1s,
                                 const uint64_t* d_c2col, size_t num_dealers, uint32_t in_repr,
                                 uint64_t* d_noisy, void* stream);
/* decrypt_party_shares (decryption.rs:281-325) with device pointers end to end: d_noisy [D][L][l] is scratch /
 * optional output (power basis), d_out [D] the decoded values.  Asynchronous on `stream`; internally the decode
 * of one chunk of dealers overlaps the inner products of the next. */
PVW_API int32_t pvw_decrypt_batch_device(pvw_ctx* ctx, const int64_t* d_sk, const uint64_t* d_c1s,
                                 const uint64_t* d_c2col, size_t num_dealers, uint32_t in_repr,
                                 uint64_t* d_noisy, uint64_t* d_out, void* stream);
/* A secret key kept on the device as the inner products read it -- NTT(sk[j]) in the ciphertext layout, what
 * SecretKey::get_polynomial computes k times per decrypt_party_value (src/keys/secret_key.rs:98-112, called at
 * src/crypto/decryption.rs:260).  decrypt_party_shares (decryption.rs:281-325) decrypts n ciphertexts under ONE key: load it
 * once, decrypt with pvw_decrypt_batch_device_sk as often as needed (no transform of the key, no wipe per call), free it when
 * the SecretKey is dropped -- pvw_sk_free clears the device copy (Zeroize + ZeroizeOnDrop, secret_key.rs:20-30).
 * sk: k x l coefficients on the host.  The handle belongs to `ctx` and must be freed before it. */
typedef struct pvw_sk pvw_sk;
PVW_API int32_t pvw_sk_load(pvw_ctx* ctx, const int64_t* sk /*[k][l]*/, pvw_sk** out);
PVW_API int32_t pvw_sk_free(pvw_sk* key);
PVW_API int32_t pvw_decrypt_batch_device_sk(pvw_ctx* ctx, const pvw_sk* key, const uint64_t* d_c1s,
                                            const uint64_t* d_c2col, size_t num_dealers, uint32_t in_repr,
                                            uint64_t* d_noisy, uint64_t* d_out, void* stream);
/* decode_scalar_pvw_rns alone, on the device: noisy [D][L][l] power basis (host) -> out_u64 [D] */
PVW_API int32_t pvw_decode(pvw_ctx* ctx, const uint64_t* noisy, size_t count, uint64_t* out_u64);
/* the same with host big integers on the host cores (no GPU needed): an independent
 * implementation kept as a cross-check of the device algorithm and for GPU-less tooling */
PVW_API int32_t pvw_decode_host(const pvw_ctx* ctx, const uint64_t* noisy, size_t count, uint64_t* out_u64);
/* device pointers: d_noisy -> d_out [D].  pvw_decrypt_batch uses this, so only D x u64 leave the GPU. */
PVW_API int32_t pvw_decode_device(pvw_ctx* ctx, const uint64_t* d_noisy, size_t count, uint64_t* d_out,
                                  void* stream);
/* SELF-TEST hook: runs the device decode algorithm (pvw_decode.h) on the host so it can be checked
 * without a GPU.  No product path calls it. */
PVW_API int32_t pvw_selftest_decode_fixed(const pvw_ctx* ctx, const uint64_t* noisy, size_t count,
                                          uint64_t* out_u64);

/* SELF-TEST: C[32][32] (int32) = A[32][32] * B[32][32] (int8, row-major) with one i8 MFMA fetched
 * through the lane maps the digit-GEMM kernels assume. */
PVW_API int32_t pvw_selftest_mfma_i8(pvw_ctx* ctx, const int8_t* a, const int8_t* b, int32_t* out);
/* SELF-TEST: key hygiene.  SecretKey is Zeroize + ZeroizeOnDrop in the reference (src/keys/secret_key.rs:20-30);
 * here every device region that held key material during pvw_keygen / pvw_decrypt_* / pvw_sample_secret_keys
 * (uploaded coefficients, NTT(sk), key errors, their tiled / digitised copies) is cleared on the call's stream
 * before the call returns its workspace.  Reports how many 64-bit words of those regions (as declared by the
 * last such call on each pooled workspace) are not zero, and how many were scanned. */
PVW_API int32_t pvw_selftest_secret_residue(pvw_ctx* ctx, uint64_t* nonzero_words, uint64_t* scanned_words);
/* SELF-TEST: SipHash-c-d of msg under (k0, k1) -- the hash behind pvw_crs_seed_from_tag, exposed so that it can be
 * pinned against the published SipHash-2-4 vector */
PVW_API int32_t pvw_selftest_siphash(const uint8_t* msg, size_t len, uint64_t k0, uint64_t k1, int32_t c_rounds,
                                     int32_t d_rounds, uint64_t* out);
/* SELF-TEST (host only, no GPU): the short path of the device gadget decode (decode_scalar_pvw_rns,
 * src/crypto/decryption.rs:10-247) restated sequentially with the same arithmetic -- candidates confirmed on every limb,
 * noise_{l-1} proven from the residues, the chain to a fixed point.  short_path[d] (may be NULL) = 1 where every proof
 * held; elsewhere the value comes from pvw_selftest_decode_fixed's algorithm.  No product path calls it. */
PVW_API int32_t pvw_selftest_decode_shortcuts(const pvw_ctx* ctx, const uint64_t* noisy, size_t count, uint64_t* out_u64,
                                              uint8_t* short_path);
/* SELF-TEST (host only, no GPU): the per-context constants behind the short cuts of the device gadget decode
 * (decode_scalar_pvw_rns, src/crypto/decryption.rs:10-247: mixed-radix inverses and partial products of the leading
 * moduli, the normalised 2*Delta with its reciprocal, Delta^(l-1) mod q_i with its inverses) checked against their
 * defining identities.  info_out[0..3] = leading moduli used (0 = none), whether their mixed-radix digits reduce with one
 * subtraction, whether the chain runs on short operands, whether noise_{l-1} is proven without the Horner lift. */
PVW_API int32_t pvw_selftest_decode_tables(const pvw_ctx* ctx, uint32_t info_out[4]);
/* 1 for the measurement build libpvw_hip_tuning.so (include/pvw_hip_tuning.h: environment-selected kernel
 * schedules, timing ablations, bandwidth probe), 0 for the shipped library, which reads no environment variable */
PVW_API int32_t pvw_build_is_tuning(void);

/* ---- ring primitives (fhe-math call sites) -----------------------
 * change_representation(Ntt / PowerBasis) on `count` polynomials, host buffers, in place */
PVW_API int32_t pvw_ntt_forward(pvw_ctx* ctx, uint64_t* polys, size_t count);
PVW_API int32_t pvw_ntt_inverse(pvw_ctx* ctx, uint64_t* polys, size_t count);
/* Poly::from_coefficients(&[i64]) + NTT: coeffs [count][l] -> polys [count][L][l] */
PVW_API int32_t pvw_small_to_poly(pvw_ctx* ctx, const int64_t* coeffs, size_t count, uint64_t* polys,
                          uint32_t repr);

/* ---- samplers (src/sampling) on the device, counter-based, written to host ----------
 * polynomial p of the call uses stream (domain<<32) | (index0+p); out [count][l]. */
PVW_API int32_t pvw_sample_cbd(pvw_ctx* ctx, const uint8_t seed[32], uint32_t domain, uint32_t index0,
                       size_t count, float variance, int64_t* out);      /* uniform.rs:27-70 */
PVW_API int32_t pvw_sample_uniform(pvw_ctx* ctx, const uint8_t seed[32], uint32_t domain,
                           uint32_t index0, size_t count, uint64_t bound, int64_t* out); /* uniform.rs:5-22 */
PVW_API int32_t pvw_sample_gaussian(pvw_ctx* ctx, const uint8_t seed[32], uint32_t index0, size_t count,
                            uint64_t bound, int64_t* out /*[count]*/);    /* normal.rs:12-20,136-162 */

/* ---- measurement hooks -------------------------------------------------------------
 * With profiling on, every kernel launch of the context is bracketed by HIP events on
 * its stream; pvw_ctx_kernel_time returns the accumulated device time of kernel `name`
 * ("mac_rows", "prep", "sample", "intt", "decrypt_mac", ...) and resets nothing. */
PVW_API int32_t pvw_ctx_set_profiling(pvw_ctx* ctx, int32_t on);
PVW_API int32_t pvw_ctx_kernel_time(pvw_ctx* ctx, const char* name, double* total_ms, uint64_t* launches);
PVW_API int32_t pvw_ctx_reset_profiling(pvw_ctx* ctx);
/* Host memory the device can write directly (pinned + mapped).  pvw_encrypt recognises output buffers that live in
 * such memory -- from here, or pinned / registered by the caller (hipHostMalloc, hipHostRegister) -- and, for
 * PVW_REPR_NTT output, has the kernel store c1 / c2 straight into them while it runs: the ciphertexts (4.7 MB at
 * n = 4096, k = 256, 1037-bit q) cross PCIe under the kernel instead of in a copy after it.  Pageable buffers work as
 * before.  (encryption.rs:105 returns host objects; a Rust host allocates the Vec it converts from with this.) */
PVW_API int32_t pvw_host_alloc(size_t bytes, void** out);
PVW_API int32_t pvw_host_free(void* p);

/* geometry of the resident tensors (bytes) for roofline bookkeeping: the tiled A-hat / B-hat sections ... */
PVW_API int32_t pvw_ctx_resident_bytes(const pvw_ctx* ctx, uint64_t* crs_bytes, uint64_t* pk_bytes);
/* ... and the derived copies held next to them at the moment (0 when not built) */
PVW_API int32_t pvw_ctx_derived_bytes(const pvw_ctx* ctx, uint64_t* packed_bytes, uint64_t* mfma_tiled_bytes);

/* ---- derived copies as an explicit step (GlobalPublicKey's mutators take &mut self, src/keys/public_key.rs:214-263:
 * in the reference a key change and an encrypt never overlap, so there is a well-defined moment for this) -------------
 * pvw_prepare builds, NOW and on `stream` (NULL = the context's own), what later *_device calls on that stream would
 * otherwise build lazily: the stream's workspace, and for
 *   PVW_PREPARE_PACKED  the bit-packed copies of A-hat / B-hat that single-dealer encrypt streams (encryption.rs:177-200):
 *                       40 / 48 / 56 / 61 bits per residue by the widest modulus; l <= 16, k a multiple of 64 (256 at
 *                       61 bits); skipped -- not an error -- when the geometry does not qualify or memory is short
 *                       (pvw_ctx_packed_active tells; the tiled matrices are streamed then, and the allocation is
 *                       retried by later encrypts every so often);
 *   PVW_PREPARE_MFMA    the MFMA-tiled copies and digit buffers of multi-dealer encrypt (encryption.rs:253-286).
 * It allocates, waits for the builds, and returns the bytes it allocated for the copies in *bytes_out (may be NULL).
 * Any later pvw_load_crs* / pvw_load_pk* / pvw_keygen / pvw_*_generate / fill invalidates the copies of the matrix it
 * touched; call pvw_prepare again (only that matrix's copies are rebuilt; nothing is reallocated). */
enum { PVW_PREPARE_PACKED = 1, PVW_PREPARE_MFMA = 2 };
PVW_API int32_t pvw_prepare(pvw_ctx* ctx, uint32_t flags, void* stream, uint64_t* bytes_out);
/* the stream single-dealer encrypt would use right now: *width_out = bits per residue of the valid packed copies,
 * 0 = the tiled matrices (copies not built, invalidated, geometry not eligible, or no room) */
PVW_API int32_t pvw_ctx_packed_active(const pvw_ctx* ctx, uint32_t* width_out);
PVW_API int32_t pvw_ctx_synchronize(pvw_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PVW_HIP_H */
