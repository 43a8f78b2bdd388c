/* PVW encrypt / decrypt hot path -- CPU restatement in plain C (RNS + NTT domain).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing here is linked into or called by the
 * product (pvw_rs_amd/): only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load this library, and only as the checker / the reported
 * CPU baseline.
 *
 * It restates, limb-wise in the NTT domain and with the reference's loop
 * structure (parallel over parties, serial over the k-term inner product,
 * one modular multiply + one modular add per element), these reference
 * routines (file:line under the reference checkout):
 *   encrypt                         src/crypto/encryption.rs:105-214
 *   PvwCrs::multiply_by_randomness  src/params/crs.rs:177-205
 *   encode_scalar                   src/params/parameters.rs:346-367
 *   bigints_to_poly residue rule    src/params/parameters.rs:437-451
 *   decrypt_party_value (to noisy)  src/crypto/decryption.rs:249-274
 *   SecretKey::get_polynomial       src/keys/secret_key.rs:98-112
 *   PublicKey::generate             src/keys/public_key.rs:111-147, crs.rs:138-171
 *   sample_vec_cbd / uniform        src/sampling/uniform.rs:5-70
 * The ring arithmetic itself (fhe-math Poly, NttOperator, Modulus) is a git
 * dependency that is not vendored in the reference checkout, so it is restated
 * from the mathematics: R_Q = Z_Q[X]/(X^l+1) in RNS form, residues in [0,q).
 *
 * PARITY STATUS: pinned, in the power basis, against oracle/pvw_model.py (an
 * independent big-integer schoolbook model which is itself pinned against the
 * properties the reference's tests state -- see its header).  PARITY UNPINNED
 * for the NTT-domain slot order / choice of psi of fhe-math and for all
 * sampled streams (the reference uses thread_rng()).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;

/* ---------------------------------------------------------------- modular */
typedef struct {
  u64 q;
  u64 ratio_lo, ratio_hi; /* floor(2^128 / q) */
} mod_t;

static mod_t mod_make(u64 q) {
  mod_t m;
  m.q = q;
  /* floor((2^128 - 1) / q) == floor(2^128 / q) for q not a power of two */
  u128 r = (~(u128)0) / q;
  m.ratio_lo = (u64)r;
  m.ratio_hi = (u64)(r >> 64);
  return m;
}

/* x mod q for any 128-bit x (Barrett with the 128-bit ratio; q < 2^62).
 * -DPVW_ORACLE_PLAIN_MOD (libpvw_oracle_plain.so): the compiler's own 128-bit remainder instead -- slow, and shares
 * nothing with the product's Barrett step (pvw_arith.h); tests/test_oracle_c.py holds the two builds together. */
static inline u64 reduce128(u128 x, const mod_t *m) {
#ifdef PVW_ORACLE_PLAIN_MOD
  return (u64)(x % (u128)m->q);
#endif
  u64 x0 = (u64)x, x1 = (u64)(x >> 64);
  /* low 64 bits of floor(x * ratio / 2^128) */
  u128 a = (u128)x0 * m->ratio_lo;
  u128 b = (u128)x0 * m->ratio_hi;
  u128 c = (u128)x1 * m->ratio_lo;
  u128 mid = (a >> 64) + (u64)b + (u64)c;
  u64 quo = (u64)(mid >> 64) + (u64)(b >> 64) + (u64)(c >> 64) + x1 * m->ratio_hi;
  u64 r = x0 - quo * m->q;
  while (r >= m->q) r -= m->q;
  return r;
}
static inline u64 mulmod(u64 a, u64 b, const mod_t *m) { return reduce128((u128)a * b, m); }
static inline u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
static inline u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
static u64 powmod(u64 b, u64 e, const mod_t *m) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = mulmod(r, b, m);
    b = mulmod(b, b, m);
    e >>= 1;
  }
  return r;
}
/* non-negative residue of a signed 64-bit integer: ((c % q) + q) % q, parameters.rs:440-443 */
static inline u64 signed_residue(i64 c, u64 q) {
  i64 r = c % (i64)q;
  return r < 0 ? (u64)(r + (i64)q) : (u64)r;
}

static u32 bitrev(u32 i, u32 bits) {
  u32 r = 0;
  for (u32 b = 0; b < bits; ++b) { r = (r << 1) | (i & 1); i >>= 1; }
  return r;
}
static u32 ilog2(u32 x) { u32 b = 0; while ((1u << b) < x) ++b; return b; }

/* Smallest primitive `order`-th root of unity mod q (order = 2l): this build's rule for psi. */
u64 pvwo_min_primitive_root(u64 q, u32 order) {
  mod_t m = mod_make(q);
  u64 e = (q - 1) / order, w = 0;
  for (u64 g = 2;; ++g) {
    w = powmod(g, e, &m);
    if (powmod(w, order / 2, &m) == q - 1) break;
  }
  u64 best = w, cur = w, w2 = mulmod(w, w, &m);
  for (u32 i = 1; i < order / 2; ++i) {
    cur = mulmod(cur, w2, &m);
    if (cur < best) best = cur;
  }
  return best;
}

/* ------------------------------------------------------------------- NTT */
typedef struct {
  mod_t m;
  u32 l;
  u64 tw[128];  /* tw[i]  = psi^bitrev(i)     , i in [1,l) */
  u64 itw[128]; /* itw[i] = psi^-bitrev(i)                 */
  u64 linv;     /* l^-1 mod q                               */
} ntt_t;

static void ntt_make(ntt_t *t, u64 q, u64 psi, u32 l) {
  t->m = mod_make(q);
  t->l = l;
  u32 bits = ilog2(l);
  u64 ipsi = powmod(psi, q - 2, &t->m);
  for (u32 i = 0; i < l; ++i) {
    t->tw[i] = powmod(psi, bitrev(i, bits), &t->m);
    t->itw[i] = powmod(ipsi, bitrev(i, bits), &t->m);
  }
  t->linv = powmod(l, q - 2, &t->m);
}
/* forward negacyclic NTT, natural order in, bit-reversed order out:
 * slot s holds a(psi^(2*bitrev(s)+1)). */
static void ntt_fwd(u64 *a, const ntt_t *t) {
  u32 l = t->l, step = l;
  u64 q = t->m.q;
  for (u32 m = 1; m < l; m <<= 1) {
    step >>= 1;
    for (u32 i = 0; i < m; ++i) {
      u64 w = t->tw[m + i];
      u32 j1 = 2 * i * step, j2 = j1 + step;
      for (u32 j = j1; j < j2; ++j) {
        u64 u = a[j], v = mulmod(a[j + step], w, &t->m);
        a[j] = addmod(u, v, q);
        a[j + step] = submod(u, v, q);
      }
    }
  }
}
static void ntt_inv(u64 *a, const ntt_t *t) {
  u32 l = t->l, step = 1;
  u64 q = t->m.q;
  for (u32 m = l >> 1; m >= 1; m >>= 1) {
    for (u32 i = 0; i < m; ++i) {
      u64 w = t->itw[m + i];
      u32 j1 = 2 * i * step, j2 = j1 + step;
      for (u32 j = j1; j < j2; ++j) {
        u64 u = a[j], v = a[j + step];
        a[j] = addmod(u, v, q);
        a[j + step] = mulmod(submod(u, v, q), w, &t->m);
      }
    }
    step <<= 1;
  }
  for (u32 j = 0; j < l; ++j) a[j] = mulmod(a[j], t->linv, &t->m);
}

/* context: moduli + per-limb NTT tables */
typedef struct {
  u32 L, l;
  ntt_t *t;
} pvwo_ctx;

pvwo_ctx *pvwo_ctx_create(const u64 *moduli, const u64 *psi, u32 L, u32 l) {
  if (l > 128) return NULL;
  pvwo_ctx *c = (pvwo_ctx *)malloc(sizeof(pvwo_ctx));
  c->L = L;
  c->l = l;
  c->t = (ntt_t *)malloc(sizeof(ntt_t) * L);
  for (u32 i = 0; i < L; ++i) {
    u64 p = psi ? psi[i] : pvwo_min_primitive_root(moduli[i], 2 * l);
    ntt_make(&c->t[i], moduli[i], p, l);
  }
  return c;
}
void pvwo_ctx_destroy(pvwo_ctx *c) {
  if (c) { free(c->t); free(c); }
}

/* polys: [count][L][l] in place */
void pvwo_ntt_forward(const pvwo_ctx *c, u64 *polys, size_t count) {
#pragma omp parallel for schedule(static)
  for (long long p = 0; p < (long long)count; ++p)
    for (u32 i = 0; i < c->L; ++i) ntt_fwd(polys + ((size_t)p * c->L + i) * c->l, &c->t[i]);
}
void pvwo_ntt_inverse(const pvwo_ctx *c, u64 *polys, size_t count) {
#pragma omp parallel for schedule(static)
  for (long long p = 0; p < (long long)count; ++p)
    for (u32 i = 0; i < c->L; ++i) ntt_inv(polys + ((size_t)p * c->L + i) * c->l, &c->t[i]);
}

/* Poly::from_coefficients(&[i64]) then change_representation(Ntt)
 * (encryption.rs:148-152, secret_key.rs:107-110): coeffs [count][l] -> out [count][L][l] */
void pvwo_small_to_ntt(const pvwo_ctx *c, const i64 *coeffs, size_t count, u64 *out) {
#pragma omp parallel for schedule(static)
  for (long long p = 0; p < (long long)count; ++p)
    for (u32 i = 0; i < c->L; ++i) {
      u64 *o = out + ((size_t)p * c->L + i) * c->l;
      for (u32 s = 0; s < c->l; ++s) o[s] = signed_residue(coeffs[(size_t)p * c->l + s], c->t[i].m.q);
      ntt_fwd(o, &c->t[i]);
    }
}

/* out[row] = sum_j M[row][j] (.) v[j]   -- the k-term inner product of
 * encryption.rs:185-192 / crs.rs:188-201: one mulmod + one addmod per element,
 * serial over j, rows in parallel (rayon over parties).  M: [rows][k][L][l]. */
static void mac_row(const pvwo_ctx *c, const u64 *Mrow, const u64 *v, u32 k, u64 *acc) {
  size_t poly = (size_t)c->L * c->l;
  memset(acc, 0, poly * sizeof(u64));
  for (u32 j = 0; j < k; ++j) {
    const u64 *b = Mrow + (size_t)j * poly, *r = v + (size_t)j * poly;
    for (u32 i = 0; i < c->L; ++i) {
      const mod_t *m = &c->t[i].m;
      for (u32 s = 0; s < c->l; ++s) {
        size_t e = (size_t)i * c->l + s;
        acc[e] = addmod(acc[e], mulmod(b[e], r[e], m), m->q);
      }
    }
  }
}
void pvwo_mac_rows(const pvwo_ctx *c, const u64 *M, const u64 *v, size_t rows, u32 k, u64 *out,
                   int parallel) {
  size_t poly = (size_t)c->L * c->l;
#pragma omp parallel for schedule(static) if (parallel)
  for (long long row = 0; row < (long long)rows; ++row)
    mac_row(c, M + (size_t)row * k * poly, v, k, out + (size_t)row * poly);
}

/* encrypt with explicit randomness, NTT-domain inputs and outputs.
 *   a_hat [k][k][L][l], b_hat [n][k][L][l], g_hat [L][l] = NTT(gadget residues)
 *   r, e1 [k][l], e2 [n][l] small signed; scalars [n]
 *   c1 [k][L][l], c2 [n][L][l]
 * serial_c1 != 0 runs the c1 double loop on one thread, as crs.rs:188 does. */
void pvwo_encrypt(const pvwo_ctx *c, u32 n, u32 k, const u64 *a_hat, const u64 *b_hat,
                  const u64 *g_hat, const u64 *scalars, const i64 *r, const i64 *e1,
                  const i64 *e2, u64 *c1, u64 *c2, int serial_c1) {
  size_t poly = (size_t)c->L * c->l;
  u64 *r_hat = (u64 *)malloc(sizeof(u64) * poly * k);
  pvwo_small_to_ntt(c, r, k, r_hat);                                /* encryption.rs:147-154 */
  pvwo_mac_rows(c, a_hat, r_hat, k, k, c1, !serial_c1);             /* :158 */
  u64 *e_hat = (u64 *)malloc(sizeof(u64) * poly * (k > n ? k : n));
  pvwo_small_to_ntt(c, e1, k, e_hat);                               /* :161-167 */
  for (size_t x = 0; x < poly * k; ++x) {                           /* :171-173 */
    u32 i = (u32)((x % poly) / c->l);
    c1[x] = addmod(c1[x], e_hat[x], c->t[i].m.q);
  }
  pvwo_small_to_ntt(c, e2, n, e_hat);                               /* :196 */
#pragma omp parallel for schedule(static)
  for (long long p = 0; p < (long long)n; ++p) {                    /* :177-200 */
    u64 *out = c2 + (size_t)p * poly;
    mac_row(c, b_hat + (size_t)p * k * poly, r_hat, k, out);
    i64 m = (i64)scalars[p];                                        /* `as i64`, :195 */
    for (u32 i = 0; i < c->L; ++i) {
      const mod_t *md = &c->t[i].m;
      u64 mr = signed_residue(m, md->q);
      for (u32 s = 0; s < c->l; ++s) {
        size_t e = (size_t)i * c->l + s;
        u64 enc = mulmod(mr, g_hat[e], md);                         /* parameters.rs:346-367 */
        out[e] = addmod(addmod(out[e], enc, md->q), e_hat[(size_t)p * poly + e], md->q); /* :198 */
      }
    }
  }
  free(r_hat);
  free(e_hat);
}

/* b[p][c] = sum_j sk[p][j] (.) A[j][c] + e[p][c]  (crs.rs:152-168, public_key.rs:134-139)
 *   a_hat [k][k][L][l]; sk, ek [n][k][l]; b_hat out [n][k][L][l] */
void pvwo_keygen(const pvwo_ctx *c, u32 n, u32 k, const u64 *a_hat, const i64 *sk, const i64 *ek,
                 u64 *b_hat) {
  size_t poly = (size_t)c->L * c->l;
#pragma omp parallel for schedule(static)
  for (long long p = 0; p < (long long)n; ++p) {
    u64 *s_hat = (u64 *)malloc(sizeof(u64) * poly * k);
    u64 *e_hat = (u64 *)malloc(sizeof(u64) * poly * k);
    pvwo_small_to_ntt(c, sk + (size_t)p * k * c->l, k, s_hat);
    pvwo_small_to_ntt(c, ek + (size_t)p * k * c->l, k, e_hat);
    for (u32 col = 0; col < k; ++col) {
      u64 *out = b_hat + ((size_t)p * k + col) * poly;
      memcpy(out, e_hat + (size_t)col * poly, poly * sizeof(u64));
      for (u32 j = 0; j < k; ++j) {
        const u64 *a = a_hat + ((size_t)j * k + col) * poly, *s = s_hat + (size_t)j * poly;
        for (u32 i = 0; i < c->L; ++i) {
          const mod_t *m = &c->t[i].m;
          for (u32 x = 0; x < c->l; ++x) {
            size_t e = (size_t)i * c->l + x;
            out[e] = addmod(out[e], mulmod(s[e], a[e], m), m->q);
          }
        }
      }
    }
    free(s_hat);
    free(e_hat);
  }
}

/* noisy[d] = INTT( sum_j NTT(sk[j]) (.) c1s[d][j] - c2col[d] )   (decryption.rs:257-274,
 * then the PowerBasis view decode works on, :116).  c1s [D][k][L][l], c2col [D][L][l]
 * NTT domain; noisy [D][L][l] power-basis residues. */
void pvwo_decrypt_noisy(const pvwo_ctx *c, u32 k, const i64 *sk, const u64 *c1s, const u64 *c2col,
                        size_t D, u64 *noisy) {
  size_t poly = (size_t)c->L * c->l;
  u64 *s_hat = (u64 *)malloc(sizeof(u64) * poly * k);
  pvwo_small_to_ntt(c, sk, k, s_hat);                               /* secret_key.rs:98-112 */
#pragma omp parallel for schedule(static)
  for (long long d = 0; d < (long long)D; ++d) {
    u64 *out = noisy + (size_t)d * poly;
    mac_row(c, c1s + (size_t)d * k * poly, s_hat, k, out);
    for (u32 i = 0; i < c->L; ++i) {
      u64 q = c->t[i].m.q;
      for (u32 s = 0; s < c->l; ++s) {
        size_t e = (size_t)i * c->l + s;
        out[e] = submod(out[e], c2col[(size_t)d * poly + e], q);
      }
      ntt_inv(out + (size_t)i * c->l, &c->t[i]);
    }
  }
  free(s_hat);
}

/* ------------------------------------------------- ChaCha8 counter RNG */
#define ROTL(x, n) (((x) << (n)) | ((x) >> (32 - (n))))
#define QR(a, b, c, d)                                                                            \
  a += b; d ^= a; d = ROTL(d, 16); c += d; b ^= c; b = ROTL(b, 12);                                \
  a += b; d ^= a; d = ROTL(d, 8);  c += d; b ^= c; b = ROTL(b, 7);

static void chacha8_block(const u32 key[8], u64 counter, u64 stream, u32 out[16]) {
  u32 st[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574, key[0], key[1], key[2], key[3],
                key[4], key[5], key[6], key[7], (u32)counter, (u32)(counter >> 32), (u32)stream,
                (u32)(stream >> 32)};
  u32 x[16];
  memcpy(x, st, sizeof x);
  for (int i = 0; i < 4; ++i) {
    QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13])
    QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
    QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12])
    QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
  }
  for (int i = 0; i < 16; ++i) out[i] = x[i] + st[i];
}
typedef struct {
  u32 key[8];
  u64 stream, counter;
  u32 buf[16];
  int pos;
} rng_t;
static void rng_init(rng_t *g, const uint8_t seed[32], u32 domain, u32 index) {
  for (int i = 0; i < 8; ++i)
    g->key[i] = (u32)seed[4 * i] | (u32)seed[4 * i + 1] << 8 | (u32)seed[4 * i + 2] << 16 |
                (u32)seed[4 * i + 3] << 24;
  g->stream = ((u64)domain << 32) | index;
  g->counter = 0;
  g->pos = 16;
}
static u32 rng_u32(rng_t *g) {
  if (g->pos == 16) {
    chacha8_block(g->key, g->counter++, g->stream, g->buf);
    g->pos = 0;
  }
  return g->buf[g->pos++];
}
static u64 rng_u64(rng_t *g) {
  u64 lo = rng_u32(g);
  u64 hi = rng_u32(g);
  return lo | (hi << 32);
}

/* sample_vec_cbd (uniform.rs:27-70) for polynomials index0 .. index0+count-1; out [count][l].
 * returns 0 ok, 1 bad variance */
int pvwo_sample_cbd(const uint8_t seed[32], u32 domain, u32 index0, size_t count, u32 l,
                    float variance, i64 *out) {
  if (!(variance >= 0.5f && variance <= 16.0f)) return 1;
  int half = (variance - 0.5f < 1.1920929e-07f) && (0.5f - variance < 1.1920929e-07f);
  u32 v = (u32)variance;
  if (!half && v < 1) return 1;
#pragma omp parallel for schedule(static)
  for (long long p = 0; p < (long long)count; ++p) {
    rng_t g;
    rng_init(&g, seed, domain, index0 + (u32)p);
    i64 *o = out + (size_t)p * l;
    if (half) {
      for (u32 s = 0; s < l; ++s) {
        i64 b1 = rng_u32(&g) & 1;
        i64 b2 = rng_u32(&g) & 1;
        o[s] = b1 - b2;
      }
    } else {
      u32 nbits = 4 * v;
      u128 mask_add = (u128)((~(u64)0 >> (64 - nbits)) >> (2 * v));
      u128 mask_sub = mask_add << (2 * v);
      u128 pool = 0;
      u32 pool_n = 0;
      for (u32 s = 0; s < l; ++s) {
        if (pool_n < nbits) {
          pool |= (u128)rng_u64(&g) << pool_n;
          pool_n += 64;
        }
        u128 pa = pool & mask_add, ps = pool & mask_sub;
        o[s] = (i64)(__builtin_popcountll((u64)pa) + __builtin_popcountll((u64)(pa >> 64))) -
               (i64)(__builtin_popcountll((u64)ps) + __builtin_popcountll((u64)(ps >> 64)));
        pool >>= nbits;
        pool_n -= nbits;
      }
    }
  }
  return 0;
}

/* sample_uniform_coefficients (uniform.rs:5-22) in [-bound, bound], bound < 2^62 */
void pvwo_sample_uniform(const uint8_t seed[32], u32 domain, u32 index0, size_t count, u32 l,
                         u64 bound, i64 *out) {
  u64 range = 2 * bound + 1;
  u32 bits = 64 - (u32)__builtin_clzll(range);
  u32 digits = bits / 32, rem = bits % 32;
#pragma omp parallel for schedule(static)
  for (long long p = 0; p < (long long)count; ++p) {
    rng_t g;
    rng_init(&g, seed, domain, index0 + (u32)p);
    for (u32 s = 0; s < l; ++s) {
      u64 v;
      do {
        u32 w0 = rng_u32(&g), w1 = 0;
        if (digits + (rem ? 1 : 0) > 1) w1 = rng_u32(&g);
        if (digits == 0) { w0 >>= 32 - rem; }
        else if (digits == 1 && rem) { w1 >>= 32 - rem; }
        v = (u64)w0 | ((u64)w1 << 32);
      } while (v >= range);
      out[(size_t)p * l + s] = (i64)v - (i64)bound;
    }
  }
}

/* uniform residues in [0,q): polynomial `index` of domain `domain` gets, for limb i,
 * stream index (index0+p)*L + i; out [count][L][l] */
void pvwo_fill_uniform_residues(const pvwo_ctx *c, const uint8_t seed[32], u32 domain, u32 index0,
                                size_t count, u64 *out) {
#pragma omp parallel for schedule(static)
  for (long long p = 0; p < (long long)count; ++p)
    for (u32 i = 0; i < c->L; ++i) {
      rng_t g;
      rng_init(&g, seed, domain, (index0 + (u32)p) * c->L + i);
      u64 q = c->t[i].m.q;
      u32 sh = (u32)__builtin_clzll(q);
      u64 *o = out + ((size_t)p * c->L + i) * c->l;
      for (u32 s = 0; s < c->l;) {
        u64 v = rng_u64(&g) >> sh;
        if (v < q) o[s++] = v;
      }
    }
}

int pvwo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
