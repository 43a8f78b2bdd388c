// pvw_capi.hip -- C ABI (include/pvw_hip.h) over the gfx950 kernels: context, parameter
// arithmetic, device residency of A-hat / B-hat, encrypt / decrypt orchestration, decode.
//
// The product path has no CPU fallback: every bulk polynomial operation runs in a HIP
// kernel and every device entry point fails with PVW_ERR_INTERNAL when no gfx950 device
// is usable.  Host-side code here is what the reference also does on the host with
// num-bigint: parameter set-up (src/params/parameters.rs:117-195), the f64 correctness
// gate (:510-551) and the integer gadget decode (src/crypto/decryption.rs:10-247).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pvw_hip.h"
#include "pvw_arith.h"
#include "pvw_bignum.h"
#include "pvw_chacha.h"
#include "pvw_decode.h"
#include "pvw_kernels.h"
#if PVW_TUNING
#include "../../include/pvw_hip_tuning.h"
#endif

using namespace pvw;

// ------------------------------------------------------------------------ errors
static thread_local std::string g_last_error;

static int32_t fail(int32_t code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
#define PVW_HIP(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(PVW_ERR_INTERNAL, std::string("HIP error: ") + hipGetErrorString(e_) +  \
                                        " at " #expr);                                     \
  } while (0)
#define PVW_TRY(expr)              \
  do {                             \
    int32_t rc_ = (expr);          \
    if (rc_ != PVW_OK) return rc_; \
  } while (0)

// ------------------------------------------------------------------------ number theory (host)
static bool is_prime_u64(u64 n) {
  if (n < 2) return false;
  static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (u64 p : small)
    if (n % p == 0) return n == p;
  u64 d = n - 1;
  int s = 0;
  while ((d & 1) == 0) { d >>= 1; ++s; }
  for (u64 a : small) {
    u128 x = 1, b = a % n;
    for (u64 e = d; e; e >>= 1) {
      if (e & 1) x = x * b % n;
      b = b * b % n;
    }
    if (x == 1 || x == n - 1) continue;
    bool comp = true;
    for (int i = 1; i < s; ++i) {
      x = x * x % n;
      if (x == n - 1) { comp = false; break; }
    }
    if (comp) return false;
  }
  return true;
}
static u32 ilog2(u32 x) { u32 b = 0; while ((1u << b) < x) ++b; return b; }

// smallest primitive `order`-th root of unity mod q: this build's deterministic psi
static u64 min_primitive_root(const Mod& m, u32 order) {
  u64 e = (m.q - 1) / order, w = 0;
  for (u64 g = 2;; ++g) {
    w = powmod(g, e, m);
    if (powmod(w, order / 2, m) == m.q - 1) break;
  }
  u64 best = w, cur = w, w2 = mulmod(w, w, m);
  for (u32 i = 1; i < order / 2; ++i) {
    cur = mulmod(cur, w2, m);
    if (cur < best) best = cur;
  }
  return best;
}
// runtime-l host NTT (table building, gadget): natural in, bit-reversed out
static void host_ntt_forward(u64* a, u32 l, const u64* tw, const Mod& m) {
  u32 step = l;
  for (u32 mm = 1; mm < l; mm <<= 1) {
    step >>= 1;
    for (u32 i = 0; i < mm; ++i) {
      u64 w = tw[mm + i];
      for (u32 j = 2 * i * step; j < 2 * i * step + step; ++j) {
        u64 u = a[j], v = mulmod(a[j + step], w, m);
        a[j] = addmod(u, v, m.q);
        a[j + step] = submod(u, v, m.q);
      }
    }
  }
}

// ------------------------------------------------------------------------ context
struct ProfRec {
  std::string name;
  hipEvent_t a, b;
};
struct Workspace {
  hipStream_t stream = nullptr;
  bool own_stream = false;
  u64* rhat = nullptr;       // [L][k][l]  (x4: up to four r-hat / s-hat vectors)
  size_t rhat_bytes = 0;
  i64* esmall = nullptr;     // [rowsA + rowsB][l] sampled e1 | e2 coefficients of one encrypt (compact addends, l <= 16)
  u64* dpart = nullptr;      // range sums of a split decrypt_mac [nsplit][dealers][L][l]
  size_t dpart_bytes = 0;
  u64* scalars = nullptr;    // [n]
  u64* c1 = nullptr;         // [rowsA][L][l]
  u64* c2 = nullptr;         // [rowsB][L][l]
  void* scratch = nullptr;   // growable staging
  size_t scratch_bytes = 0;
  // digit-GEMM operands for up to 16 vectors (allocated on first use)
  u64* vhat16 = nullptr;     // [16][L][k][l]
  signed char* yd = nullptr; // vector digit tiles
  int* sy = nullptr;         // their column sums
  u64* gtmpA = nullptr;      // GEMM intermediates [limb][slot][v][row]
  u64* gtmpB = nullptr;
  u64* gtmpK = nullptr;      // ... for key generation (k rows)
  // helper stream + events for pvw_decrypt_batch_device (decode of chunk i under the MAC of chunk i+1)
  hipStream_t aux = nullptr;
  std::vector<hipEvent_t> events;
  // device regions that hold secret-key material during the current call (sk coefficients, NTT(sk), key errors,
  // their MFMA-tiled / digitised copies): cleared on the call's stream before the workspace goes back to the pool
  // (the reference's SecretKey is Zeroize + ZeroizeOnDrop, src/keys/secret_key.rs:20-30).  `wiped` remembers what
  // the last call cleared, for pvw_selftest_secret_residue.
  struct Span { void* p; size_t bytes; };
  std::vector<Span> secrets, wiped;
};
static void ws_mark_secret(Workspace* w, void* p, size_t bytes) {
  if (p && bytes) w->secrets.push_back(Workspace::Span{p, bytes});
}
// enqueue the wipes on `s` (call after the last kernel that reads the regions has been enqueued on `s`)
static hipError_t ws_wipe_secrets(Workspace* w, hipStream_t s) {
  hipError_t rc = hipSuccess;
  for (const Workspace::Span& sp : w->secrets) {
    hipError_t e = hipMemsetAsync(sp.p, 0, sp.bytes, s);
    if (e != hipSuccess) rc = e;
  }
  w->wiped = w->secrets;
  w->secrets.clear();
  return rc;
}

struct pvw_ctx {
  u32 n, k, l, L;
  std::vector<u64> moduli, psi;
  float variance;
  u64 b1, b2;
  u32 num_cus = 256;               // of the context's device (set by ensure_device)
  u32 xm_bytes = 8;                // bytes per element in the MFMA-tiled copies xmA / xmB (set when they are built)
  int device;
  u32 party_lo, party_hi, c1_lo, c1_hi;
  BigInt Q, halfQ, delta, delta_pow;
  std::vector<BigInt> crt_qi;    // Q / q_i
  std::vector<u64> crt_inv;      // (Q/q_i)^-1 mod q_i
  std::vector<Mod> mods;
  std::vector<u64> tw, itw, linv, ghat, gpow;
  bool roots_locked = false;
  // fixed-width decode tables (host copies; pvw_decode.h)
  std::vector<u64> dec_words;      // all u64 tables back to back
  DecodeTables dec_host{};         // pointers into dec_words
  DecodeTables dec_dev{};          // pointers into d_dec
  void* d_dec = nullptr;

  // device state
  std::atomic<bool> dev_ready{false};
  void* d_tables = nullptr;
  DevTables dt{};
  u64* dA = nullptr;  // tiled A-hat rows [c1_lo, c1_hi)
  u64* dB = nullptr;  // tiled B-hat rows [party_lo, party_hi)
  // MFMA-tiled copies for the digit GEMM (built lazily by the first many-dealer encrypt, dropped
  // whenever A or B changes)
  u64* xmA = nullptr;
  u64* xmB = nullptr;
  bool xm_valid = false;
  // packed copies (pk_width = 40 / 48 / 56 / 61 bits per residue, by the widest modulus) for the single-dealer
  // mac_rows: built by pvw_prepare, or lazily by the first encrypt after the matrices changed, when the geometry
  // allows it and the memory is there (launch_pack); a section is rebuilt when ITS matrix changed
  u64* pkA = nullptr;
  u64* pkB = nullptr;
  u32* pk_flag = nullptr;   // device word: set by pack_kernel when a matrix word exceeds the stream width
  u32 pk_width = 0;         // 0: the geometry does not qualify (decided at context creation)
  bool pkA_valid = false, pkB_valid = false;
  bool pk_wide = false;     // the resident matrices hold unreduced words: no packed stream until they change
  u32 pk_nomem_calls = 0;   // encrypts since the copies last failed to fit (the allocation is retried now and then)
  bool crs_loaded = false;
  u32 num_keys = 0;
  hipStream_t stream = nullptr;

  std::mutex mu, init_mu;
  std::vector<Workspace*> pool;
  std::map<void*, Workspace*> async_ws;

  bool profiling = false;
  std::vector<ProfRec> prof;
  std::map<std::string, std::pair<double, uint64_t>> prof_acc;

  u32 rowsA() const { return c1_hi - c1_lo; }
  u32 rowsB() const { return party_hi - party_lo; }
  u32 R() const { return 128 / l; }
  size_t poly() const { return (size_t)L * l; }
  size_t tiled_words(u32 rows) const { return (size_t)((rows + R() - 1) / R()) * L * k * 128; }
};

struct ProfScope {
  pvw_ctx* c;
  hipStream_t s;
  ProfRec rec;
  bool on;
  ProfScope(pvw_ctx* c_, const char* name, hipStream_t s_) : c(c_), s(s_), on(c_->profiling) {
    if (on) {
      rec.name = name;
      hipEventCreate(&rec.a);
      hipEventCreate(&rec.b);
      hipEventRecord(rec.a, s);
    }
  }
  ~ProfScope() {
    if (on) {
      hipEventRecord(rec.b, s);
      std::lock_guard<std::mutex> g(c->mu);
      c->prof.push_back(rec);
    }
  }
};

static void build_tables(pvw_ctx* c) {
  const u32 L = c->L, l = c->l, bits = ilog2(l);
  c->tw.assign((size_t)L * l, 0);
  c->itw.assign((size_t)L * l, 0);
  c->linv.assign(L, 0);
  c->gpow.assign((size_t)L * l, 0);
  c->ghat.assign((size_t)L * l, 0);
  for (u32 i = 0; i < L; ++i) {
    const Mod& m = c->mods[i];
    u64 psi = c->psi[i], ipsi = powmod(psi, m.q - 2, m);
    for (u32 x = 0; x < l; ++x) {
      c->tw[(size_t)i * l + x] = powmod(psi, bitrev32(x, bits), m);
      c->itw[(size_t)i * l + x] = powmod(ipsi, bitrev32(x, bits), m);
    }
    c->linv[i] = powmod(l, m.q - 2, m);
    // gadget residues D^j mod q_i (parameters.rs:288-308) and their NTT
    u64 dm = c->delta.mod_small(m.q), p = 1;
    for (u32 j = 0; j < l; ++j) {
      c->gpow[(size_t)i * l + j] = p;
      p = mulmod(p, dm, m);
    }
    std::vector<u64> g(c->gpow.begin() + (size_t)i * l, c->gpow.begin() + (size_t)(i + 1) * l);
    host_ntt_forward(g.data(), l, &c->tw[(size_t)i * l], m);
    std::copy(g.begin(), g.end(), c->ghat.begin() + (size_t)i * l);
  }
}

static double correctness_bound(double n, double k, double l, double b1, double b2) {  // parameters.rs:510-544
  double first = b2 * std::sqrt(n * l) * (1.0 + std::sqrt(n));
  double second = 2.0 * b1 * k * l;
  double third = 14.0 * b1 * std::sqrt(n * k * l);
  return first + second + third;
}

// ---- decode tables (pvw_decode.h): big constants as W little-endian words ----
static void bn_words(const BigInt& v, size_t W, u64* out) {
  for (size_t i = 0; i < W; ++i) out[i] = i < v.mag.size() ? v.mag[i] : 0;
}
static void build_decode_tables(pvw_ctx* c) {
  const size_t L = c->L;
  const size_t W = (c->Q.bits() + 8 + 63) / 64;
  // layout (u64 words): Q | halfQ | qi[L][W] | inv[L] | invp[L] | pow64[L][W] | dmod[L] | dmodp[L] |
  //                     delta | dpow | half_dpow | dpow_n | td_n
  const size_t total = 2 * W + L * W + 2 * L + L * W + 2 * L + 5 * W + W * L + 3 * (W + 2) + 56 + 9 + 4 * L;
  c->dec_words.assign(total, 0);
  u64* p = c->dec_words.data();
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += n; return o; };
  const size_t oQ = take(W), oH = take(W), oQi = take(L * W), oInv = take(L), oInvp = take(L), oPow = take(L * W),
               oDm = take(L), oDmp = take(L), oDelta = take(W), oDpow = take(W), oHalfD = take(W), oDpn = take(W), oTdn = take(W),
               oPowT = take(W * L), oTd = take(W + 2), oMuTd = take(W + 2), oMuDp = take(W + 2), oGar = take(56), oSc = take(9), oDpm = take(4 * L);
  bn_words(c->Q, W, p + oQ);
  bn_words(c->halfQ, W, p + oH);
  for (size_t i = 0; i < L; ++i) {
    const Mod& m = c->mods[i];
    bn_words(c->crt_qi[i], W, p + oQi + i * W);
    p[oInv + i] = c->crt_inv[i];
    p[oInvp + i] = shoup_precompute(c->crt_inv[i], m.q);
    u64 b = powmod(2, 64, m), cur = 1;
    for (size_t j = 0; j < W; ++j) { p[oPow + i * W + j] = cur; cur = mulmod(cur, b, m); }
    u64 dm = c->delta.mod_small(m.q);
    p[oDm + i] = dm;
    p[oDmp + i] = shoup_precompute(dm, m.q);
  }
  for (size_t i = 0; i < L; ++i)
    for (size_t j = 0; j < W; ++j) p[oPowT + j * L + i] = p[oPow + i * W + j];
  {
    BigInt td = c->delta * BigInt(2), bw1 = BigInt(1).shl(64 * (W + 1));
    bn_words(td, W + 2, p + oTd);
    bn_words(bw1 / td, W + 2, p + oMuTd);
    bn_words(bw1 / c->delta_pow, W + 2, p + oMuDp);
  }
  bn_words(c->delta, W, p + oDelta);
  bn_words(c->delta_pow, W, p + oDpow);
  bn_words(c->delta_pow.shr(1), W, p + oHalfD);
  auto norm = [&](const BigInt& v, u64* dst, u32& nw, u32& sh) {
    nw = (u32)v.mag.size();
    sh = (u32)__builtin_clzll(v.mag.back());
    bn_words(v.shl(sh), W, dst);
  };
  DecodeTables t{};
  t.W = (u32)W; t.L = c->L; t.ell = c->l;
  norm(c->delta_pow, p + oDpn, t.dpow_nw, t.dpow_sh);
  norm(c->delta * BigInt(2), p + oTdn, t.td_nw, t.td_sh);
  // Garner constants of the short-cut lift (pvw_decode_wave.h).  A well-formed ciphertext's chain inputs are at most about
  // Delta times the decryption noise; the noise is below the bound of the correctness gate (parameters.rs:510-544).  Use as
  // many of the leading moduli as make half their product cover that with a few bits to spare: at least 2, at most 4, and
  // at least one limb must remain to confirm a candidate against.  (Whatever does not fit takes the full lift: the choice
  // only decides how often the short cut applies.)
  t.gar_n = 0;
  if (L >= 3) {
    const double bound = correctness_bound((double)c->n, (double)c->k, (double)c->l, (double)c->b1, (double)c->b2);
    const size_t need = c->delta.bits() + (size_t)std::ceil(std::log2(bound + 2.0)) + 4;
    const size_t most = std::min<size_t>(4, L - 1);
    u64* g = p + oGar;
    BigInt prod(1);
    size_t nl = 0;
    while (nl < most && (nl < 2 || prod.bits() < need + 1)) prod = prod * BigInt(c->moduli[nl++]);
    t.gar_n = (u32)nl;
    t.gar_close = 1;
    prod = BigInt(1);
    for (size_t j = 0; j <= nl; ++j) {
      bn_words(prod, 4, g + 32 + 4 * j);
      if (j == nl) break;
      for (size_t i = 0; i < j; ++i) {
        const u64 v = powmod(c->moduli[i] % c->moduli[j], c->moduli[j] - 2, c->mods[j]);
        g[4 * j + i] = v;
        g[16 + 4 * j + i] = shoup_precompute(v, c->moduli[j]);
        if (c->moduli[i] >= 2 * c->moduli[j]) t.gar_close = 0;
      }
      prod = prod * BigInt(c->moduli[j]);
    }
    bn_words(prod.shr(1), 4, g + 52);
  }
  // constants of the chain step on noise-sized operands: an operand below 2^191 on either side keeps a - b inside
  // (-Q/2, Q/2) once Q has 194 bits, so the integer difference IS the centred difference the reference takes mod Q
  {
    const BigInt td = c->delta * BigInt(2);
    t.sc_on = (t.gar_n != 0 && c->Q.bits() >= 194 && td.mag.size() <= 3) ? 1 : 0;
    if (t.sc_on) {
      u64* sc = p + oSc;
      const size_t sh = 192 - td.bits();
      const BigInt td3 = td.shl(sh);
      bn_words(td3, 3, sc);
      const BigInt recip = (BigInt(1).shl(128) - BigInt(1)) / BigInt(sc[2]) - BigInt(1).shl(64);
      bn_words(recip, 1, sc + 3);
      sc[4] = sh / 64;
      sc[5] = sh % 64;
      bn_words(c->delta, 3, sc + 6);
    }
  }
  // Delta^(l-1) mod q_i, its inverse
  {
    bool all_inv = true;
    for (size_t i = 0; i < L; ++i) {
      const u64 q = c->moduli[i], v = c->delta_pow.mod_small(q);
      p[oDpm + i] = v;
      p[oDpm + L + i] = shoup_precompute(v, q);
      const u64 inv = v ? powmod(v, q - 2, c->mods[i]) : 0;
      if (!v) all_inv = false;
      p[oDpm + 2 * L + i] = inv;
      p[oDpm + 3 * L + i] = shoup_precompute(inv, q);
    }
    // e Delta^(l-1) + g must stay inside (-Q/2, Q/2) for every |e| <= q_0/2 < 2^61, |g| <= Delta/2
    const bool room = BigInt::cmp(c->halfQ, c->delta_pow.shl(61) + c->delta) > 0;
    t.hs_on = (t.sc_on && all_inv && room && c->l >= 3 && c->moduli[0] < (1ULL << 62)) ? 1 : 0;
  }
  auto bind = [&](DecodeTables& d, const u64* base, const Mod* mods) {
    d = t;
    d.mods = mods;
    d.Q = base + oQ; d.halfQ = base + oH; d.qi = base + oQi; d.inv = base + oInv; d.invp = base + oInvp;
    d.pow64 = base + oPow; d.dmod = base + oDm; d.dmodp = base + oDmp; d.delta = base + oDelta;
    d.dpow = base + oDpow; d.half_dpow = base + oHalfD; d.dpow_n = base + oDpn; d.td_n = base + oTdn;
    d.pow64T = base + oPowT; d.td = base + oTd; d.mu_td = base + oMuTd; d.mu_dp = base + oMuDp; d.gar = base + oGar; d.sc = base + oSc; d.dpm = base + oDpm;
  };
  bind(c->dec_host, p, c->mods.data());
  c->dec_dev = t;   // pointers bound at upload
}

static int32_t upload_tables(pvw_ctx* c) {
  const size_t L = c->L, l = c->l;
  const size_t bytes = L * sizeof(Mod) + 8 * L * l * 8 + 2 * L * 8;
  if (!c->d_tables) PVW_HIP(hipMalloc(&c->d_tables, bytes));
  char* p = (char*)c->d_tables;
  auto put = [&](const void* src, size_t n) -> const void* {
    const void* at = p;
    hipMemcpy(p, src, n, hipMemcpyHostToDevice);
    p += n;
    return at;
  };
  c->dt.mods = (const Mod*)put(c->mods.data(), L * sizeof(Mod));
  c->dt.min_q_bits = 64;
  c->dt.max_q_bits = 0;
  for (u32 i = 0; i < L; ++i) {
    u32 bits = 0;
    for (u64 q = c->mods[i].q; q; q >>= 1) ++bits;
    if (bits < c->dt.min_q_bits) c->dt.min_q_bits = bits;
    if (bits > c->dt.max_q_bits) c->dt.max_q_bits = bits;
  }
  c->dt.tw = (const u64*)put(c->tw.data(), L * l * 8);
  c->dt.itw = (const u64*)put(c->itw.data(), L * l * 8);
  c->dt.ghat = (const u64*)put(c->ghat.data(), L * l * 8);
  c->dt.gpow = (const u64*)put(c->gpow.data(), L * l * 8);
  c->dt.linv = (const u64*)put(c->linv.data(), L * 8);
  // Shoup companions floor(x * 2^64 / q)
  auto shoup = [&](const std::vector<u64>& src, size_t per_limb) {
    std::vector<u64> out(src.size());
    for (size_t i = 0; i < src.size(); ++i) out[i] = shoup_precompute(src[i], c->moduli[i / per_limb]);
    return out;
  };
  std::vector<u64> twp = shoup(c->tw, l), itwp = shoup(c->itw, l), ghatp = shoup(c->ghat, l),
                   gpowp = shoup(c->gpow, l), linvp = shoup(c->linv, 1);
  c->dt.twp = (const u64*)put(twp.data(), L * l * 8);
  c->dt.itwp = (const u64*)put(itwp.data(), L * l * 8);
  c->dt.ghatp = (const u64*)put(ghatp.data(), L * l * 8);
  c->dt.gpowp = (const u64*)put(gpowp.data(), L * l * 8);
  c->dt.linvp = (const u64*)put(linvp.data(), L * 8);
  // decode tables
  if (!c->d_dec) PVW_HIP(hipMalloc(&c->d_dec, c->dec_words.size() * 8));
  PVW_HIP(hipMemcpy(c->d_dec, c->dec_words.data(), c->dec_words.size() * 8, hipMemcpyHostToDevice));
  {
    const u64* hb = c->dec_words.data();
    const u64* db = (const u64*)c->d_dec;
    DecodeTables d = c->dec_host;
    auto mv = [&](const u64* hp) { return db + (hp - hb); };
    d.mods = c->dt.mods;
    d.Q = mv(d.Q); d.halfQ = mv(d.halfQ); d.qi = mv(d.qi); d.inv = mv(d.inv); d.invp = mv(d.invp); d.pow64 = mv(d.pow64);
    d.dmod = mv(d.dmod); d.dmodp = mv(d.dmodp); d.delta = mv(d.delta); d.dpow = mv(d.dpow); d.half_dpow = mv(d.half_dpow);
    d.dpow_n = mv(d.dpow_n); d.td_n = mv(d.td_n);
    d.pow64T = mv(d.pow64T); d.td = mv(d.td); d.mu_td = mv(d.mu_td); d.mu_dp = mv(d.mu_dp); d.gar = mv(d.gar); d.sc = mv(d.sc); d.dpm = mv(d.dpm);
    c->dec_dev = d;
  }
  PVW_HIP(hipDeviceSynchronize());
  return PVW_OK;
}

static int32_t ensure_device(pvw_ctx* c) {
  if (c->dev_ready) {
    PVW_HIP(hipSetDevice(c->device));
    return PVW_OK;
  }
  std::lock_guard<std::mutex> init_guard(c->init_mu);   // concurrent first calls initialise once
  if (c->dev_ready) {
    PVW_HIP(hipSetDevice(c->device));
    return PVW_OK;
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    return fail(PVW_ERR_INTERNAL, "no HIP device available: the PVW hot path has no CPU fallback");
  if (c->device < 0) {
    int cur = 0;
    PVW_HIP(hipGetDevice(&cur));
    c->device = cur;
  }
  if (c->device >= count) return fail(PVW_ERR_INTERNAL, "device ordinal out of range");
  PVW_HIP(hipSetDevice(c->device));
  hipDeviceProp_t prop;
  PVW_HIP(hipGetDeviceProperties(&prop, c->device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(PVW_ERR_INTERNAL, std::string("kernels are built for gfx950 only, device is ") +
                                      prop.gcnArchName);
  c->num_cus = (u32)prop.multiProcessorCount;
  PVW_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  PVW_HIP(init_kernel_attributes());          // per device: dynamic-LDS limits of the decode kernels
  PVW_TRY(upload_tables(c));
  c->dev_ready = true;
  c->roots_locked = true;
  return PVW_OK;
}

static int32_t ws_alloc(pvw_ctx* c, Workspace* w) {
  const size_t k = c->k, P = c->poly();
  PVW_HIP(hipMalloc((void**)&w->rhat, 4 * k * P * 8));   // up to 4 r-hat / s-hat vectors (mac_rows_multi)
  w->rhat_bytes = 4 * k * P * 8;
  PVW_HIP(hipMemset(w->rhat, 0, w->rhat_bytes));          // recycled device memory may hold an earlier owner's data
  if (c->l <= 16) PVW_HIP(hipMalloc((void**)&w->esmall, ((size_t)c->rowsA() + c->rowsB()) * c->l * 8 + 16));
  return PVW_OK;
}
static int32_t ws_host_buffers(pvw_ctx* c, Workspace* w) {
  const size_t P = c->poly();
  if (!w->scalars) PVW_HIP(hipMalloc((void**)&w->scalars, (size_t)c->n * 8 + 16));
  if (!w->c1) PVW_HIP(hipMalloc((void**)&w->c1, (size_t)c->rowsA() * P * 8 + 16));
  if (!w->c2) PVW_HIP(hipMalloc((void**)&w->c2, (size_t)c->rowsB() * P * 8 + 16));
  return PVW_OK;
}
static int32_t ws_scratch(Workspace* w, size_t bytes) {
  if (w->scratch_bytes >= bytes) return PVW_OK;
  // callers grow the scratch before they enqueue anything that uses it; whatever an earlier call left there is
  // cleared before the block goes back to the allocator
  if (w->scratch) { hipMemset(w->scratch, 0, w->scratch_bytes); hipFree(w->scratch); }
  w->wiped.clear();
  w->scratch = nullptr;
  w->scratch_bytes = 0;
  PVW_HIP(hipMalloc(&w->scratch, bytes));
  w->scratch_bytes = bytes;
  return PVW_OK;
}
static void ws_free(Workspace* w) {
  if (!w) return;
  // r-hat / s-hat vectors and the staging block may hold key material of the last call
  if (w->rhat && w->rhat_bytes) hipMemset(w->rhat, 0, w->rhat_bytes);
  if (w->scratch) hipMemset(w->scratch, 0, w->scratch_bytes);
  hipFree(w->rhat);
  hipFree(w->esmall);
  hipFree(w->dpart);
  hipFree(w->scalars);
  hipFree(w->c1);
  hipFree(w->c2);
  hipFree(w->scratch);
  hipFree(w->vhat16);
  hipFree(w->yd);
  hipFree(w->sy);
  hipFree(w->gtmpA);
  hipFree(w->gtmpB);
  hipFree(w->gtmpK);
  for (hipEvent_t e : w->events) hipEventDestroy(e);
  if (w->aux) hipStreamDestroy(w->aux);
  if (w->own_stream && w->stream) hipStreamDestroy(w->stream);
  delete w;
}
// synchronous host-buffer calls take a private workspace + stream from the pool
static int32_t ws_acquire(pvw_ctx* c, Workspace** out) {
  {
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->pool.empty()) {
      *out = c->pool.back();
      c->pool.pop_back();
      return PVW_OK;
    }
  }
  Workspace* w = new Workspace();
  if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) {
    delete w;
    return fail(PVW_ERR_INTERNAL, "hipStreamCreate failed");
  }
  w->own_stream = true;
  int32_t rc = ws_alloc(c, w);
  if (rc != PVW_OK) { ws_free(w); return rc; }
  *out = w;
  return PVW_OK;
}
static void ws_release(pvw_ctx* c, Workspace* w) {
  std::lock_guard<std::mutex> g(c->mu);
  c->pool.push_back(w);
}
// asynchronous device-pointer calls keep one workspace per stream (stream order protects it)
static int32_t ws_for_stream(pvw_ctx* c, hipStream_t s, Workspace** out) {
  std::lock_guard<std::mutex> g(c->mu);
  auto it = c->async_ws.find((void*)s);
  if (it != c->async_ws.end()) { *out = it->second; return PVW_OK; }
  Workspace* w = new Workspace();
  w->stream = s;
  int32_t rc = ws_alloc(c, w);
  if (rc != PVW_OK) { ws_free(w); return rc; }
  c->async_ws[(void*)s] = w;
  *out = w;
  return PVW_OK;
}

// ------------------------------------------------------------------------ parameters
static int32_t validate_params(const pvw_params_t* p) {
  if (!p) return fail(PVW_ERR_INVALID_PARAMETERS, "params is NULL");
  if (p->n == 0) return fail(PVW_ERR_INVALID_PARAMETERS, "n must be > 0");                 // parameters.rs:132
  if (p->k == 0) return fail(PVW_ERR_INVALID_PARAMETERS, "k must be > 0");                 // :135
  if (p->l < 8 || (p->l & (p->l - 1)) != 0)                                               // :140
    return fail(PVW_ERR_INVALID_PARAMETERS, "l must be power of 2 and >= 8 (fhe.rs Context requirement)");
  if (p->l > 64) return fail(PVW_ERR_INVALID_PARAMETERS, "l > 64 is not supported by the gfx950 kernels");
  if (p->num_moduli == 0 || !p->moduli) return fail(PVW_ERR_INVALID_PARAMETERS, "moduli not set");  // :129
  for (u32 i = 0; i < p->num_moduli; ++i) {
    u64 q = p->moduli[i];
    char buf[96];
    snprintf(buf, sizeof buf, "Context creation failed: modulus %#llx ", (unsigned long long)q);
    if (q >= (1ull << 62) || q < 3) return fail(PVW_ERR_INVALID_PARAMETERS, std::string(buf) + "must be in [3, 2^62)");
    if (!is_prime_u64(q)) return fail(PVW_ERR_INVALID_PARAMETERS, std::string(buf) + "is not prime");
    if ((q - 1) % (2 * (u64)p->l) != 0)
      return fail(PVW_ERR_INVALID_PARAMETERS, std::string(buf) + "is not 1 mod 2l (no NTT of size l)");
    for (u32 j = 0; j < i; ++j)
      if (p->moduli[j] == q) return fail(PVW_ERR_INVALID_PARAMETERS, std::string(buf) + "is repeated");
  }
  if (p->error_bound_1 == 0) return fail(PVW_ERR_INVALID_PARAMETERS, "error_bound_1 must be positive");  // :172
  if (p->error_bound_2 == 0) return fail(PVW_ERR_INVALID_PARAMETERS, "error_bound_2 must be positive");  // :177
  if (p->error_bound_1 >= (1ull << 62) || p->error_bound_2 >= (1ull << 62))
    return fail(PVW_ERR_INVALID_PARAMETERS, "error bounds must be below 2^62");
  return PVW_OK;
}

static void compute_q_delta(const u64* moduli, u32 L, u32 l, BigInt& Q, BigInt& delta, BigInt& dpow) {
  Q = BigInt(1);
  for (u32 i = 0; i < L; ++i) Q = Q * BigInt(moduli[i]);
  delta = Q.nth_root(l);            // parameters.rs:156
  dpow = delta.pow(l - 1);          // :159-163
}


extern "C" {

int32_t pvw_last_error(char* buf, size_t len) {
  if (!buf || len == 0) return PVW_ERR_INVALID_PARAMETERS;
  snprintf(buf, len, "%s", g_last_error.c_str());
  return PVW_OK;
}

int32_t pvw_device_available(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return 0;
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

int32_t pvw_ctx_create(const pvw_params_t* p, pvw_ctx** out) {
  if (!out) return fail(PVW_ERR_INVALID_PARAMETERS, "out is NULL");
  *out = nullptr;
  PVW_TRY(validate_params(p));
  pvw_ctx* c = new pvw_ctx();
  c->n = p->n; c->k = p->k; c->l = p->l; c->L = p->num_moduli;
  c->moduli.assign(p->moduli, p->moduli + p->num_moduli);
  c->variance = p->secret_variance;
  c->b1 = p->error_bound_1; c->b2 = p->error_bound_2;
  c->device = p->device;
  c->party_lo = p->party_lo; c->party_hi = p->party_hi;
  c->c1_lo = p->c1_lo; c->c1_hi = p->c1_hi;
  if (c->party_lo == 0 && c->party_hi == 0) c->party_hi = c->n;
  if (c->c1_lo == 0 && c->c1_hi == 0) c->c1_hi = c->k;
  if (c->party_lo > c->party_hi || c->party_hi > c->n || c->c1_lo > c->c1_hi || c->c1_hi > c->k) {
    delete c;
    return fail(PVW_ERR_INVALID_PARAMETERS, "party / c1 shard out of range");
  }
  compute_q_delta(c->moduli.data(), c->L, c->l, c->Q, c->delta, c->delta_pow);
  c->halfQ = c->Q.shr(1);
  for (u32 i = 0; i < c->L; ++i) {
    c->mods.push_back(make_mod(c->moduli[i]));
    c->psi.push_back(min_primitive_root(c->mods[i], 2 * c->l));
    BigInt qi = c->Q / BigInt(c->moduli[i]);
    c->crt_qi.push_back(qi);
    c->crt_inv.push_back(powmod(qi.mod_small(c->moduli[i]), c->moduli[i] - 2, c->mods[i]));
  }
  build_tables(c);
  build_decode_tables(c);
  {
    u32 maxbits = 0;
    for (u64 q : c->moduli) {
      u32 bits = 0;
      for (; q; q >>= 1) ++bits;
      if (bits > maxbits) maxbits = bits;
    }
    c->pk_width = packed_width(maxbits, c->k, c->l);
  }
  *out = c;
  return PVW_OK;
}

int32_t pvw_ctx_destroy(pvw_ctx* c) {
  if (!c) return PVW_OK;
  if (c->dev_ready) {
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (auto& r : c->prof) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (Workspace* w : c->pool) ws_free(w);
    for (auto& kv : c->async_ws) ws_free(kv.second);
    hipFree(c->dA);
    hipFree(c->dB);
    hipFree(c->xmA);
    hipFree(c->xmB);
    hipFree(c->pkA);
    hipFree(c->pkB);
    hipFree(c->pk_flag);
    hipFree(c->d_tables);
    hipFree(c->d_dec);
    if (c->stream) hipStreamDestroy(c->stream);
  }
  delete c;
  return PVW_OK;
}

int32_t pvw_ctx_get_roots(const pvw_ctx* c, uint64_t* psi_out) {
  if (!c || !psi_out) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  memcpy(psi_out, c->psi.data(), c->L * 8);
  return PVW_OK;
}

int32_t pvw_ctx_set_roots(pvw_ctx* c, const uint64_t* psi) {
  if (!c || !psi) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  if (c->roots_locked)
    return fail(PVW_ERR_CONTEXT, "roots must be set before any device operation");
  for (u32 i = 0; i < c->L; ++i)
    if (psi[i] >= c->moduli[i] || powmod(psi[i], c->l, c->mods[i]) != c->moduli[i] - 1)
      return fail(PVW_ERR_INVALID_PARAMETERS, "psi is not a primitive 2l-th root of unity");
  c->psi.assign(psi, psi + c->L);
  build_tables(c);
  return PVW_OK;
}

static int32_t export_big(const BigInt& v, uint64_t* words, size_t cap, size_t* nwords) {
  if (!nwords) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  *nwords = v.mag.size();
  if (words) {
    if (cap < v.mag.size()) return fail(PVW_ERR_INSUFFICIENT_DATA, "word buffer too small");
    memcpy(words, v.mag.data(), v.mag.size() * 8);
  }
  return PVW_OK;
}
int32_t pvw_ctx_delta(const pvw_ctx* c, uint64_t* w, size_t cap, size_t* n) {
  if (!c) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL context");
  return export_big(c->delta, w, cap, n);
}
int32_t pvw_ctx_delta_power_l_minus_1(const pvw_ctx* c, uint64_t* w, size_t cap, size_t* n) {
  if (!c) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL context");
  return export_big(c->delta_pow, w, cap, n);
}
int32_t pvw_ctx_q_total(const pvw_ctx* c, uint64_t* w, size_t cap, size_t* n) {
  if (!c) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL context");
  return export_big(c->Q, w, cap, n);
}

int32_t pvw_ctx_gadget(const pvw_ctx* c, uint64_t* poly_out, uint32_t repr) {
  if (!c || !poly_out) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  const std::vector<u64>& src = repr == PVW_REPR_NTT ? c->ghat : c->gpow;
  memcpy(poly_out, src.data(), src.size() * 8);
  return PVW_OK;
}

int32_t pvw_ctx_verify_correctness_condition(const pvw_ctx* c, int32_t* ok) {
  if (!c || !ok) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  double bound = correctness_bound((double)c->n, (double)c->k, (double)c->l, (double)c->b1, (double)c->b2);
  *ok = c->delta_pow.to_double() > bound ? 1 : 0;   // parameters.rs:547-550 (to_f64 saturates to +inf)
  return PVW_OK;
}

int32_t pvw_suggest_error_bounds(uint32_t n, uint32_t k, uint32_t l, const uint64_t* moduli,
                                 uint32_t num_moduli, float variance, uint32_t* b1o, uint32_t* b2o) {
  if (!b1o || !b2o) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  pvw_params_t p{};
  p.n = n; p.k = k; p.l = l; p.moduli = moduli; p.num_moduli = num_moduli;
  p.secret_variance = variance; p.error_bound_1 = 1; p.error_bound_2 = 1; p.device = -1;   // parameters.rs:562-569
  PVW_TRY(validate_params(&p));
  BigInt Q, d, dp;
  compute_q_delta(moduli, num_moduli, l, Q, d, dp);
  const double dpf = dp.to_double();
  const double nf = n, kf = k, lf = l;
  const double c1 = 2.0 * kf * lf + 14.0 * std::sqrt(nf * kf * lf);      // :578-580
  const double c2 = std::sqrt(nf * lf) * (1.0 + std::sqrt(nf));          // :583-585
  static const uint32_t cand[] = {50, 100, 200, 500, 1000, 2000};
  for (uint32_t e1 : cand)
    for (uint32_t e2 : cand)
      if (dpf > (double)e1 * c1 + (double)e2 * c2) { *b1o = e1; *b2o = e2; return PVW_OK; }   // :588-598
  char buf[160];
  snprintf(buf, sizeof buf, "Cannot find suitable error bounds for variance %g with the correctness condition", variance);
  return fail(PVW_ERR_INVALID_PARAMETERS, buf);
}

int32_t pvw_ctx_resident_bytes(const pvw_ctx* c, uint64_t* crs, uint64_t* pk) {
  if (!c) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL context");
  if (crs) *crs = c->tiled_words(c->rowsA()) * 8;
  if (pk) *pk = c->tiled_words(c->rowsB()) * 8;
  return PVW_OK;
}

int32_t pvw_ctx_derived_bytes(const pvw_ctx* c, uint64_t* packed, uint64_t* mfma_tiled) {
  if (!c) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL context");
  if (packed) *packed = ((c->pkA ? packed_words(c->rowsA(), c->k, c->L, c->l, c->pk_width) : 0) +
                         (c->pkB ? packed_words(c->rowsB(), c->k, c->L, c->l, c->pk_width) : 0)) * 8;
  if (mfma_tiled) *mfma_tiled = ((c->xmA ? xm_words(c->rowsA(), c->k, c->L, c->l) : 0) + (c->xmB ? xm_words(c->rowsB(), c->k, c->L, c->l) : 0)) * 8;
  return PVW_OK;
}

int32_t pvw_ctx_packed_active(const pvw_ctx* c, uint32_t* width) {
  if (!c || !width) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  const bool valid = c->pk_width && !c->pk_wide && (c->pkA_valid || c->rowsA() == 0) && (c->pkB_valid || c->rowsB() == 0);
  *width = valid ? c->pk_width : 0;
  return PVW_OK;
}

int32_t pvw_ctx_synchronize(pvw_ctx* c) {
  if (!c) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL context");
  PVW_TRY(ensure_device(c));
  PVW_HIP(hipDeviceSynchronize());
  return PVW_OK;
}

// ------------------------------------------------------------------------ profiling
static void prof_resolve(pvw_ctx* c) {
  hipDeviceSynchronize();
  std::lock_guard<std::mutex> g(c->mu);
  for (auto& r : c->prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      auto& acc = c->prof_acc[r.name];
      acc.first += ms;
      acc.second += 1;
    }
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  c->prof.clear();
}
int32_t pvw_ctx_set_profiling(pvw_ctx* c, int32_t on) {
  if (!c) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL context");
  c->profiling = on != 0;
  return PVW_OK;
}
int32_t pvw_ctx_kernel_time(pvw_ctx* c, const char* name, double* total_ms, uint64_t* launches) {
  if (!c || !name) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  if (c->dev_ready) { hipSetDevice(c->device); prof_resolve(c); }
  std::lock_guard<std::mutex> g(c->mu);
  auto it = c->prof_acc.find(name);
  if (total_ms) *total_ms = it == c->prof_acc.end() ? 0.0 : it->second.first;
  if (launches) *launches = it == c->prof_acc.end() ? 0 : it->second.second;
  return PVW_OK;
}
int32_t pvw_ctx_reset_profiling(pvw_ctx* c) {
  if (!c) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL context");
  if (c->dev_ready) { hipSetDevice(c->device); prof_resolve(c); }
  std::lock_guard<std::mutex> g(c->mu);
  c->prof_acc.clear();
  return PVW_OK;
}

// ------------------------------------------------------------------------ CRS / public key residency
static int32_t ensure_matrix(pvw_ctx* c, u64** M, u32 rows) {
  if (*M || rows == 0) return PVW_OK;
  const size_t bytes = c->tiled_words(rows) * 8;
  PVW_HIP(hipMalloc((void**)M, bytes));
  PVW_HIP(hipMemsetAsync(*M, 0, bytes, c->stream));   // padding rows must read as zero
  PVW_HIP(hipStreamSynchronize(c->stream));
  return PVW_OK;
}

// rows [lo, hi) (global numbering) of a matrix whose shard is [shard_lo, shard_hi); d_src holds
// [hi-lo][k][L][l].  Rows outside the shard are skipped.
static int32_t load_rows_device(pvw_ctx* c, u64* M, u32 shard_lo, u32 shard_hi, u32 lo, u32 hi,
                                const u64* d_src, uint32_t repr, hipStream_t s) {
  const u32 a = lo > shard_lo ? lo : shard_lo, b = hi < shard_hi ? hi : shard_hi;
  if (a >= b) return PVW_OK;
  const size_t rowwords = (size_t)c->k * c->poly();
  ProfScope ps(c, "tile", s);
  PVW_HIP(launch_tile(d_src + (size_t)(a - lo) * rowwords, M, b - a, a - shard_lo, c->k, c->L, c->l,
                      repr == PVW_REPR_POWER, c->dt, s));
  return PVW_OK;
}
// host source, staged through a bounded device buffer
static int32_t load_rows_host(pvw_ctx* c, u64* M, u32 shard_lo, u32 shard_hi, u32 lo, u32 hi,
                              const u64* src, uint32_t repr) {
  const u32 a = lo > shard_lo ? lo : shard_lo, b = hi < shard_hi ? hi : shard_hi;
  if (a >= b) return PVW_OK;
  const size_t rowwords = (size_t)c->k * c->poly();
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  size_t chunk = ((size_t)256 << 20) / (rowwords * 8);
  if (chunk == 0) chunk = 1;
  int32_t rc = ws_scratch(w, (chunk < (size_t)(b - a) ? chunk : (size_t)(b - a)) * rowwords * 8);
  for (u32 r0 = a; rc == PVW_OK && r0 < b; r0 += (u32)chunk) {
    const u32 cnt = (b - r0) < chunk ? (b - r0) : (u32)chunk;
    if (hipMemcpyAsync(w->scratch, src + (size_t)(r0 - lo) * rowwords, cnt * rowwords * 8,
                       hipMemcpyHostToDevice, w->stream) != hipSuccess) {
      rc = fail(PVW_ERR_INTERNAL, "H2D copy failed");
      break;
    }
    rc = load_rows_device(c, M, shard_lo, shard_hi, r0, r0 + cnt, (const u64*)w->scratch, repr, w->stream);
    if (rc == PVW_OK && hipStreamSynchronize(w->stream) != hipSuccess) rc = fail(PVW_ERR_INTERNAL, "stream sync failed");
  }
  ws_release(c, w);
  return rc;
}
// dealers per gemm_digits launch: PVW_GEMM_VB batches of 16 (they share one pass over the matrix through L2)
static u32 gemm_vb() {   // tuning build: PVW_GEMM_VB
  static const u32 v = [] {
    const long x = PVW_ENV_INT("PVW_GEMM_VB", 8);
    return (u32)(x < 1 ? 1 : (x > 8 ? 8 : x));
  }();
  return v;
}
static int32_t ws_gemm_buffers(pvw_ctx* c, Workspace* w) {
  const u32 vb = gemm_vb();
  if (!w->vhat16) PVW_HIP(hipMalloc((void**)&w->vhat16, (size_t)16 * vb * c->k * c->poly() * 8));
  if (!w->yd) PVW_HIP(hipMalloc((void**)&w->yd, yd_bytes(16 * vb, c->k, c->L, c->l)));
  if (!w->sy) PVW_HIP(hipMalloc((void**)&w->sy, sy_bytes(16 * vb, c->L, c->l)));
  if (!w->gtmpA && c->rowsA()) PVW_HIP(hipMalloc((void**)&w->gtmpA, (size_t)vb * gemm_tmp_words(c->rowsA(), c->L, c->l) * 8));
  if (!w->gtmpB && c->rowsB()) PVW_HIP(hipMalloc((void**)&w->gtmpB, (size_t)vb * gemm_tmp_words(c->rowsB(), c->L, c->l) * 8));
  return PVW_OK;
}
// MFMA-tiled copies of the resident A-hat / B-hat sections
static int32_t ensure_xm(pvw_ctx* c, hipStream_t s) {
  std::lock_guard<std::mutex> g(c->init_mu);
  if (c->xm_valid) return PVW_OK;
  const u32 rA = c->rowsA(), rB = c->rowsB();
  const size_t wa = xm_words(rA, c->k, c->L, c->l), wb = xm_words(rB, c->k, c->L, c->l);
  if (!c->xmA && wa) PVW_HIP(hipMalloc((void**)&c->xmA, wa * 8));
  if (!c->xmB && wb) PVW_HIP(hipMalloc((void**)&c->xmB, wb * 8));
  if (wa) PVW_HIP(hipMemsetAsync(c->xmA, 0, wa * 8, s));
  if (wb) PVW_HIP(hipMemsetAsync(c->xmB, 0, wb * 8, s));
  // every modulus below 2^56 and k a multiple of 64: the copy holds 7 bytes per element (gemm_ktiles), and multi-dealer
  // encrypt contracts over 7/8 of the terms.  Tuning build: PVW_GEMM_BYTES=8 keeps all 8.
  c->xm_bytes = gemm7_ok(c->dt.max_q_bits, c->k) && PVW_ENV_INT("PVW_GEMM_BYTES", 7) != 8 ? 7 : 8;
  ProfScope ps(c, "mftile", s);
  PVW_HIP(launch_mftile(c->dA, true, c->xmA, rA, c->k, c->L, c->l, s, c->xm_bytes));
  PVW_HIP(launch_mftile(c->dB, true, c->xmB, rB, c->k, c->L, c->l, s, c->xm_bytes));
  PVW_HIP(hipStreamSynchronize(s));
  c->xm_valid = true;
  return PVW_OK;
}

// packed copies of the resident A-hat / B-hat sections for mac_rows (pvw_mac.hip, mac_rows_packed*_kernel): needs
// l <= 16, k a multiple of 64 (256 at 61 bits), every modulus below 2^61, and room for a second copy of the matrices.
// Returns the stream width, or 0 when the plain tiled matrices are to be streamed instead.  may_build = false (a
// stream capture is in progress): only says what is valid already, touches nothing.
static u32 ensure_packed(pvw_ctx* c, hipStream_t s, bool may_build = true, size_t* bytes_taken = nullptr) {
  // tuning build: A/B runs against the unpacked stream, and an explicit schedule of the unpacked kernel is honoured
  if (PVW_ENV_INT("PVW_MAC_PACKED", 1) == 0 || (PVW_ENV_INT("PVW_MAC_VARIANT", 0) != 0 && PVW_ENV_INT("PVW_MAC_VARIANT", 0) != 44)) return 0;
  if (c->pk_width == 0) return 0;
  std::lock_guard<std::mutex> g(c->init_mu);
  const u32 rA = c->rowsA(), rB = c->rowsB();
  if (c->pk_wide) return 0;
  if ((c->pkA_valid || rA == 0) && (c->pkB_valid || rB == 0)) return c->pk_width;
  if (!may_build) return 0;
  const size_t wa = packed_words(rA, c->k, c->L, c->l, c->pk_width), wb = packed_words(rB, c->k, c->L, c->l, c->pk_width);
  const size_t need = (c->pkA || !wa ? 0 : wa * 8) + (c->pkB || !wb ? 0 : wb * 8);
  if (need) {
    // memory was short the last time: look again every 64th encrypt, not on every call
    if (c->pk_nomem_calls && (c->pk_nomem_calls++ & 63) != 0) return 0;
    size_t free_b = 0, total_b = 0;
    bool ok = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= need + ((size_t)4 << 30);   // leave room for the callers' own buffers
    if (ok && !c->pkA && wa) ok = hipMalloc((void**)&c->pkA, wa * 8) == hipSuccess;
    if (ok && !c->pkB && wb) ok = hipMalloc((void**)&c->pkB, wb * 8) == hipSuccess;
    if (!ok) {
      // nothing half-built stays behind, and a failed hipMalloc does not poison the next launch's hipGetLastError()
      if (!c->pkA_valid) { hipFree(c->pkA); c->pkA = nullptr; }
      if (!c->pkB_valid) { hipFree(c->pkB); c->pkB = nullptr; }
      (void)hipGetLastError();
      if (c->pk_nomem_calls == 0) c->pk_nomem_calls = 1;
      return 0;
    }
    c->pk_nomem_calls = 0;
    if (bytes_taken) *bytes_taken += need;
  }
  if (!c->pk_flag && hipMalloc((void**)&c->pk_flag, sizeof(u32)) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (hipMemsetAsync(c->pk_flag, 0, sizeof(u32), s) != hipSuccess) return 0;
  {
    ProfScope ps(c, "pack", s);
    if (!c->pkA_valid && launch_pack(c->dA, c->pkA, rA, c->k, c->L, c->l, c->pk_width, c->pk_flag, s) != hipSuccess) return 0;
    if (!c->pkB_valid && launch_pack(c->dB, c->pkB, rB, c->k, c->L, c->l, c->pk_width, c->pk_flag, s) != hipSuccess) return 0;
  }
  u32 wide = 1;
  if (hipMemcpyAsync(&wide, c->pk_flag, sizeof(u32), hipMemcpyDeviceToHost, s) != hipSuccess) return 0;
  if (hipStreamSynchronize(s) != hipSuccess) return 0;
  if (wide) {                                             // residues were loaded unreduced: stream the tiled matrices as they are
    c->pk_wide = true;
    return 0;
  }
  c->pkA_valid = c->pkB_valid = true;
  return c->pk_width;
}

// The derived copies and the calling stream's workspace, built NOW: after this, pvw_encrypt_device /
// pvw_encrypt_multi_device on `stream` neither allocate nor synchronise until a matrix changes again (GlobalPublicKey's
// mutators take &mut self, public_key.rs:214-263: a change and a use never overlap).
int32_t pvw_prepare(pvw_ctx* c, uint32_t flags, void* stream, uint64_t* bytes_out) {
  if (!c) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL context");
  if (flags & ~(uint32_t)(PVW_PREPARE_PACKED | PVW_PREPARE_MFMA)) return fail(PVW_ERR_INVALID_PARAMETERS, "unknown prepare flag");
  PVW_TRY(ensure_device(c));
  if (!c->crs_loaded) return fail(PVW_ERR_CRS, "CRS not loaded");
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  Workspace* w;
  PVW_TRY(ws_for_stream(c, s, &w));
  size_t taken = 0;
  if (flags & PVW_PREPARE_PACKED) (void)ensure_packed(c, s, true, &taken);       // 0 = does not qualify / no room: pvw_ctx_packed_active tells
  if (flags & PVW_PREPARE_MFMA) {
    const bool had_a = c->xmA != nullptr, had_b = c->xmB != nullptr;
    PVW_TRY(ws_gemm_buffers(c, w));
    PVW_TRY(ensure_xm(c, s));
    if (!had_a && c->xmA) taken += xm_words(c->rowsA(), c->k, c->L, c->l) * 8;
    if (!had_b && c->xmB) taken += xm_words(c->rowsB(), c->k, c->L, c->l) * 8;
  }
  PVW_HIP(hipStreamSynchronize(s));
  if (bytes_out) *bytes_out = taken;
  return PVW_OK;
}

static int32_t check_repr(uint32_t repr) {
  if (repr != PVW_REPR_POWER && repr != PVW_REPR_NTT) return fail(PVW_ERR_INVALID_FORMAT, "unknown representation");
  return PVW_OK;
}

int32_t pvw_load_crs(pvw_ctx* c, const uint64_t* a, uint32_t repr) {
  if (!c || !a) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(check_repr(repr));
  PVW_TRY(ensure_device(c));
  c->xm_valid = false; c->pkA_valid = false; c->pk_wide = false;
  PVW_TRY(ensure_matrix(c, &c->dA, c->rowsA()));
  PVW_TRY(load_rows_host(c, c->dA, c->c1_lo, c->c1_hi, 0, c->k, a, repr));
  c->crs_loaded = true;
  return PVW_OK;
}
int32_t pvw_load_crs_device(pvw_ctx* c, const uint64_t* d_a, uint32_t repr, void* stream) {
  if (!c || !d_a) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(check_repr(repr));
  PVW_TRY(ensure_device(c));
  c->xm_valid = false; c->pkA_valid = false; c->pk_wide = false;
  PVW_TRY(ensure_matrix(c, &c->dA, c->rowsA()));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  PVW_TRY(load_rows_device(c, c->dA, c->c1_lo, c->c1_hi, 0, c->k, d_a, repr, s));
  c->crs_loaded = true;
  return PVW_OK;
}
int32_t pvw_crs_generate(pvw_ctx* c, const uint8_t seed[32]) {
  if (!c || !seed) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(ensure_device(c));
  c->xm_valid = false; c->pkA_valid = false; c->pk_wide = false;
  PVW_TRY(ensure_matrix(c, &c->dA, c->rowsA()));
  {
    ProfScope ps(c, "fill_uniform", c->stream);
    PVW_HIP(launch_fill_uniform_tiled(c->dA, make_key(seed), DOM_CRS, c->rowsA(), 0, c->c1_lo, c->k, c->L,
                                      c->l, c->dt, c->stream));
  }
  PVW_HIP(hipStreamSynchronize(c->stream));
  c->crs_loaded = true;
  return PVW_OK;
}
// PvwCrs::new_from_tag (crs.rs:74-90): seed = the 8 little-endian bytes of DefaultHasher(tag + "CRS") repeated
// four times.  std's DefaultHasher is SipHash-1-3 with a zero key; `str::hash` feeds the bytes followed by 0xFF.
static u64 siphash(const uint8_t* m, size_t len, u64 k0, u64 k1, int c_rounds, int d_rounds) {
  u64 v0 = k0 ^ 0x736f6d6570736575ULL, v1 = k1 ^ 0x646f72616e646f6dULL, v2 = k0 ^ 0x6c7967656e657261ULL,
      v3 = k1 ^ 0x7465646279746573ULL;
  auto rotl = [](u64 x, int b) { return (x << b) | (x >> (64 - b)); };
  auto round = [&]() {
    v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
    v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
    v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
    v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
  };
  size_t i = 0;
  for (; i + 8 <= len; i += 8) {
    u64 w = 0;
    for (int b = 0; b < 8; ++b) w |= (u64)m[i + b] << (8 * b);
    v3 ^= w;
    for (int r = 0; r < c_rounds; ++r) round();
    v0 ^= w;
  }
  u64 w = (u64)(len & 0xff) << 56;
  for (int b = 0; i + b < len; ++b) w |= (u64)m[i + b] << (8 * b);
  v3 ^= w;
  for (int r = 0; r < c_rounds; ++r) round();
  v0 ^= w;
  v2 ^= 0xff;
  for (int r = 0; r < d_rounds; ++r) round();
  return v0 ^ v1 ^ v2 ^ v3;
}
int32_t pvw_selftest_siphash(const uint8_t* msg, size_t len, uint64_t k0, uint64_t k1, int32_t c_rounds, int32_t d_rounds,
                             uint64_t* out) {
  if ((!msg && len) || !out || c_rounds < 1 || d_rounds < 1) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  *out = siphash(msg, len, k0, k1, c_rounds, d_rounds);
  return PVW_OK;
}
int32_t pvw_crs_seed_from_tag(const char* tag, uint8_t seed_out[32]) {
  if (!tag || !seed_out) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  std::string m = std::string(tag) + "CRS";
  m.push_back((char)0xFF);                                   // the terminator `impl Hash for str` writes
  const u64 h = siphash((const uint8_t*)m.data(), m.size(), 0, 0, 1, 3);
  for (int i = 0; i < 32; ++i) seed_out[i] = (uint8_t)(h >> (8 * (i % 8)));
  return PVW_OK;
}

// SELF-TEST (host only): the constants behind the short cuts of the device decode (DecodeTables::gar / sc / dpm, built in
// build_decode_tables) against their defining identities, recomputed here with the host big integers.  info_out[0..3] =
// gar_n, gar_close, sc_on, hs_on -- which short cuts the parameter set reaches.
int32_t pvw_selftest_decode_tables(const pvw_ctx* c, uint32_t info_out[4]) {
  if (!c || !info_out) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  const DecodeTables& t = c->dec_host;
  info_out[0] = t.gar_n; info_out[1] = t.gar_close; info_out[2] = t.sc_on; info_out[3] = t.hs_on;
  auto words = [](const u64* w, size_t n) { return BigInt::from_words(w, n); };
  if (t.gar_n) {
    if (t.gar_n < 2 || t.gar_n > 4 || t.gar_n >= c->L) return fail(PVW_ERR_INTERNAL, "decode tables: gar_n out of range");
    BigInt prod(1);
    bool close = true;
    for (u32 j = 0; j <= t.gar_n; ++j) {
      if (BigInt::cmp(words(t.gar + 32 + 4 * j, 4), prod) != 0) return fail(PVW_ERR_INTERNAL, "decode tables: partial product");
      if (j == t.gar_n) break;
      const u64 qj = c->moduli[j];
      for (u32 i = 0; i < j; ++i) {
        const u64 inv = t.gar[4 * j + i];
        if (inv >= qj || mulmod(inv, c->moduli[i] % qj, c->mods[j]) != 1) return fail(PVW_ERR_INTERNAL, "decode tables: mixed-radix inverse");
        if (t.gar[16 + 4 * j + i] != shoup_precompute(inv, qj)) return fail(PVW_ERR_INTERNAL, "decode tables: Shoup companion");
        if (c->moduli[i] >= 2 * qj) close = false;
      }
      prod = prod * BigInt(qj);
    }
    if (BigInt::cmp(words(t.gar + 52, 4), prod.shr(1)) != 0) return fail(PVW_ERR_INTERNAL, "decode tables: half product");
    if ((t.gar_close != 0) != close) return fail(PVW_ERR_INTERNAL, "decode tables: gar_close");
  }
  const BigInt td = c->delta * BigInt(2);
  if (t.sc_on) {
    if (!t.gar_n || c->Q.bits() < 194 || td.mag.size() > 3) return fail(PVW_ERR_INTERNAL, "decode tables: sc_on without its conditions");
    const size_t sh = 192 - td.bits();
    if (t.sc[4] != sh / 64 || t.sc[5] != sh % 64 || !(t.sc[2] >> 63)) return fail(PVW_ERR_INTERNAL, "decode tables: divisor shift");
    if (BigInt::cmp(words(t.sc, 3), td.shl(sh)) != 0) return fail(PVW_ERR_INTERNAL, "decode tables: normalised divisor");
    const BigInt recip = (BigInt(1).shl(128) - BigInt(1)) / BigInt(t.sc[2]) - BigInt(1).shl(64);
    if (BigInt::cmp(words(t.sc + 3, 1), recip) != 0) return fail(PVW_ERR_INTERNAL, "decode tables: reciprocal");
    if (BigInt::cmp(words(t.sc + 6, 3), c->delta) != 0) return fail(PVW_ERR_INTERNAL, "decode tables: Delta words");
  }
  bool all_inv = true;
  for (u32 i = 0; i < c->L; ++i) {
    const u64 q = c->moduli[i], v = c->delta_pow.mod_small(q);
    if (t.dpm[i] != v || t.dpm[c->L + i] != shoup_precompute(v, q)) return fail(PVW_ERR_INTERNAL, "decode tables: Delta^(l-1) residue");
    if (!v) { all_inv = false; continue; }
    const u64 inv = t.dpm[2 * c->L + i];
    if (mulmod(inv, v, c->mods[i]) != 1 || t.dpm[3 * c->L + i] != shoup_precompute(inv, q)) return fail(PVW_ERR_INTERNAL, "decode tables: Delta^(l-1) inverse");
  }
  if (t.hs_on) {
    const bool room = BigInt::cmp(c->halfQ, c->delta_pow.shl(61) + c->delta) > 0;
    if (!t.sc_on || !all_inv || !room || c->l < 3 || c->moduli[0] >= (1ULL << 62)) return fail(PVW_ERR_INTERNAL, "decode tables: hs_on without its conditions");
  }
  return PVW_OK;
}

int32_t pvw_build_is_tuning(void) { return PVW_TUNING; }

// SELF-TEST: 64-bit words that are not zero in the regions the last key-bearing call on each pooled workspace
// declared secret (and cleared), plus every workspace's r-hat / s-hat block.  0 after pvw_keygen / pvw_decrypt_*.
int32_t pvw_selftest_secret_residue(pvw_ctx* c, uint64_t* nonzero_words, uint64_t* scanned_words) {
  if (!c || !nonzero_words || !scanned_words) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  *nonzero_words = *scanned_words = 0;
  if (!c->dev_ready) return PVW_OK;
  PVW_HIP(hipSetDevice(c->device));
  PVW_HIP(hipDeviceSynchronize());
  std::vector<Workspace*> all;
  {
    std::lock_guard<std::mutex> g(c->mu);
    all = c->pool;
    for (auto& kv : c->async_ws) all.push_back(kv.second);
  }
  std::vector<u64> host;
  for (Workspace* w : all) {
    std::vector<Workspace::Span> spans = w->wiped;
    if (w->rhat) spans.push_back(Workspace::Span{w->rhat, w->rhat_bytes});
    for (const Workspace::Span& sp : spans) {
      const size_t step = (size_t)64 << 20;
      for (size_t off = 0; off < sp.bytes; off += step) {
        const size_t nb = (sp.bytes - off) < step ? (sp.bytes - off) : step;
        host.resize((nb + 7) / 8);
        host.back() = 0;
        PVW_HIP(hipMemcpy(host.data(), (const char*)sp.p + off, nb, hipMemcpyDeviceToHost));
        for (u64 v : host) *nonzero_words += v != 0;
        *scanned_words += host.size();
      }
    }
  }
  return PVW_OK;
}

static int32_t get_rows(pvw_ctx* c, const u64* M, u32 shard_lo, u32 shard_hi, u32 lo, u32 hi,
                        uint64_t* dst, uint32_t repr) {
  const u32 a = lo > shard_lo ? lo : shard_lo, b = hi < shard_hi ? hi : shard_hi;
  if (a >= b) return PVW_OK;
  if (!M) return fail(PVW_ERR_INVALID_PARAMETERS, "matrix not loaded");
  const size_t rowwords = (size_t)c->k * c->poly();
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  size_t chunk = ((size_t)256 << 20) / (rowwords * 8);
  if (chunk == 0) chunk = 1;
  int32_t rc = ws_scratch(w, (chunk < (size_t)(b - a) ? chunk : (size_t)(b - a)) * rowwords * 8);
  for (u32 r0 = a; rc == PVW_OK && r0 < b; r0 += (u32)chunk) {
    const u32 cnt = (b - r0) < chunk ? (b - r0) : (u32)chunk;
    if (launch_untile(M, (u64*)w->scratch, cnt, r0 - shard_lo, c->k, c->L, c->l, repr == PVW_REPR_POWER,
                      c->dt, w->stream) != hipSuccess ||
        hipMemcpyAsync(dst + (size_t)(r0 - lo) * rowwords, w->scratch, cnt * rowwords * 8,
                       hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
        hipStreamSynchronize(w->stream) != hipSuccess)
      rc = fail(PVW_ERR_INTERNAL, "untile / D2H failed");
  }
  ws_release(c, w);
  return rc;
}
int32_t pvw_get_crs(pvw_ctx* c, uint64_t* a_out, uint32_t repr) {
  if (!c || !a_out) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(check_repr(repr));
  PVW_TRY(ensure_device(c));
  return get_rows(c, c->dA, c->c1_lo, c->c1_hi, 0, c->k, a_out, repr);
}

static int32_t check_party_range(const pvw_ctx* c, u32 lo, u32 hi) {
  if (lo > hi) return fail(PVW_ERR_INVALID_PARAMETERS, "party_lo > party_hi");
  if (hi > c->n) {                                                                 // public_key.rs:216-222
    char buf[96];
    snprintf(buf, sizeof buf, "Party index %u exceeds maximum %u", hi - 1, c->n - 1);
    return fail(PVW_ERR_INVALID_PARAMETERS, buf);
  }
  return PVW_OK;
}
int32_t pvw_load_pk(pvw_ctx* c, uint32_t lo, uint32_t hi, const uint64_t* b, uint32_t repr) {
  if (!c || !b) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(check_repr(repr));
  PVW_TRY(check_party_range(c, lo, hi));
  PVW_TRY(ensure_device(c));
  c->xm_valid = false; c->pkB_valid = false; c->pk_wide = false;
  PVW_TRY(ensure_matrix(c, &c->dB, c->rowsB()));
  PVW_TRY(load_rows_host(c, c->dB, c->party_lo, c->party_hi, lo, hi, b, repr));
  if (hi > c->num_keys) c->num_keys = hi;                                          // public_key.rs:245-247
  return PVW_OK;
}
int32_t pvw_load_pk_device(pvw_ctx* c, uint32_t lo, uint32_t hi, const uint64_t* d_b, uint32_t repr, void* stream) {
  if (!c || !d_b) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(check_repr(repr));
  PVW_TRY(check_party_range(c, lo, hi));
  PVW_TRY(ensure_device(c));
  c->xm_valid = false; c->pkB_valid = false; c->pk_wide = false;
  PVW_TRY(ensure_matrix(c, &c->dB, c->rowsB()));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  PVW_TRY(load_rows_device(c, c->dB, c->party_lo, c->party_hi, lo, hi, d_b, repr, s));
  if (hi > c->num_keys) c->num_keys = hi;
  return PVW_OK;
}
int32_t pvw_pk_fill_uniform(pvw_ctx* c, const uint8_t seed[32]) {
  if (!c || !seed) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(ensure_device(c));
  c->xm_valid = false; c->pkB_valid = false; c->pk_wide = false;
  PVW_TRY(ensure_matrix(c, &c->dB, c->rowsB()));
  {
    ProfScope ps(c, "fill_uniform", c->stream);
    PVW_HIP(launch_fill_uniform_tiled(c->dB, make_key(seed), DOM_PK, c->rowsB(), 0, c->party_lo, c->k, c->L,
                                      c->l, c->dt, c->stream));
  }
  PVW_HIP(hipStreamSynchronize(c->stream));
  c->num_keys = c->party_hi;
  return PVW_OK;
}
int32_t pvw_get_pk(pvw_ctx* c, uint32_t lo, uint32_t hi, uint64_t* b_out, uint32_t repr) {
  if (!c || !b_out) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(check_repr(repr));
  PVW_TRY(check_party_range(c, lo, hi));
  PVW_TRY(ensure_device(c));
  return get_rows(c, c->dB, c->party_lo, c->party_hi, lo, hi, b_out, repr);
}
int32_t pvw_num_public_keys(const pvw_ctx* c, uint32_t* out) {
  if (!c || !out) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  *out = c->num_keys;
  return PVW_OK;
}
int32_t pvw_is_full(const pvw_ctx* c, int32_t* out) {
  if (!c || !out) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  *out = c->num_keys >= c->party_hi ? 1 : 0;                                        // public_key.rs:349-351
  return PVW_OK;
}

// ------------------------------------------------------------------------ samplers
static int32_t cbd_job(float variance, SampleJob& j) {
  if (!(variance >= 0.5f && variance <= 16.0f))
    return fail(PVW_ERR_SAMPLING, "The variance should be between 0.5 and 16");          // uniform.rs:32-34
  j.kind = SAMPLE_CBD;
  j.cbd_half = std::fabs(variance - 0.5f) < 1.1920929e-07f ? 1 : 0;                       // :38
  j.cbd_v = (u32)variance;                                                                // :47
  if (!j.cbd_half && j.cbd_v < 1)
    return fail(PVW_ERR_SAMPLING, "non-integer variance below 1 is not supported (the reference's bit pool is empty there)");
  j.bound = 0;
  return PVW_OK;
}

int32_t pvw_sample_cbd(pvw_ctx* c, const uint8_t seed[32], uint32_t domain, uint32_t index0, size_t count,
                       float variance, int64_t* out) {
  if (!c || !seed || (!out && count)) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  SampleJob j{}, z{};
  PVW_TRY(cbd_job(variance, j));
  if (count == 0) return PVW_OK;
  PVW_TRY(ensure_device(c));
  j.domain = domain; j.index0 = index0; j.count = (u32)count; j.out_poly0 = 0;
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  int32_t rc = ws_scratch(w, count * c->l * 8);
  if (rc == PVW_OK) {
    ProfScope ps(c, "sample", w->stream);
    if (launch_sample((i64*)w->scratch, make_key(seed), c->l, j, z, z, w->stream) != hipSuccess)
      rc = fail(PVW_ERR_INTERNAL, "sample launch failed");
  }
  if (rc == PVW_OK && hipMemcpyAsync(out, w->scratch, count * c->l * 8, hipMemcpyDeviceToHost, w->stream) != hipSuccess)
    rc = fail(PVW_ERR_INTERNAL, "D2H failed");
  if (rc == PVW_OK && domain == PVW_DOM_SK) {            // SecretKey::random: the staged coefficients are key material
    ws_mark_secret(w, w->scratch, count * c->l * 8);
    if (ws_wipe_secrets(w, w->stream) != hipSuccess) rc = fail(PVW_ERR_INTERNAL, "wipe failed");
  }
  if (hipStreamSynchronize(w->stream) != hipSuccess && rc == PVW_OK) rc = fail(PVW_ERR_INTERNAL, "D2H failed");
  ws_release(c, w);
  return rc;
}

int32_t pvw_sample_uniform(pvw_ctx* c, const uint8_t seed[32], uint32_t domain, uint32_t index0, size_t count,
                           uint64_t bound, int64_t* out) {
  if (!c || !seed || (!out && count)) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  if (bound >= (1ull << 62)) return fail(PVW_ERR_SAMPLING, "bound must be below 2^62");
  if (count == 0) return PVW_OK;
  PVW_TRY(ensure_device(c));
  SampleJob j{}, z{};
  j.kind = SAMPLE_UNIFORM; j.domain = domain; j.index0 = index0; j.count = (u32)count; j.bound = bound;
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  int32_t rc = ws_scratch(w, count * c->l * 8);
  if (rc == PVW_OK) {
    ProfScope ps(c, "sample", w->stream);
    if (launch_sample((i64*)w->scratch, make_key(seed), c->l, j, z, z, w->stream) != hipSuccess)
      rc = fail(PVW_ERR_INTERNAL, "sample launch failed");
  }
  if (rc == PVW_OK && (hipMemcpyAsync(out, w->scratch, count * c->l * 8, hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
                       hipStreamSynchronize(w->stream) != hipSuccess))
    rc = fail(PVW_ERR_INTERNAL, "D2H failed");
  ws_release(c, w);
  return rc;
}

int32_t pvw_sample_gaussian(pvw_ctx* c, const uint8_t seed[32], uint32_t index0, size_t count, uint64_t bound,
                            int64_t* out) {
  if (!c || !seed || (!out && count)) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  if (bound >= (1ull << 62)) return fail(PVW_ERR_SAMPLING, "bound must be below 2^62");
  if (count == 0) return PVW_OK;
  PVW_TRY(ensure_device(c));
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  int32_t rc = ws_scratch(w, count * 8);
  if (rc == PVW_OK) {
    ProfScope ps(c, "gaussian", w->stream);
    if (launch_gaussian((i64*)w->scratch, make_key(seed), index0, (u32)count, bound, w->stream) != hipSuccess)
      rc = fail(PVW_ERR_INTERNAL, "gaussian launch failed");
  }
  if (rc == PVW_OK && (hipMemcpyAsync(out, w->scratch, count * 8, hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
                       hipStreamSynchronize(w->stream) != hipSuccess))
    rc = fail(PVW_ERR_INTERNAL, "D2H failed");
  ws_release(c, w);
  return rc;
}

int32_t pvw_sample_secret_keys(const pvw_ctx* cc, const uint8_t seed[32], uint32_t party_lo, uint32_t count,
                               int64_t* sk_out) {
  pvw_ctx* c = const_cast<pvw_ctx*>(cc);
  if (!c) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL context");
  // party p, polynomial j uses stream index p*k + j
  return pvw_sample_cbd(c, seed, PVW_DOM_SK, party_lo * c->k, (size_t)count * c->k, c->variance, sk_out);
}

// ------------------------------------------------------------------------ ring primitives on host buffers
int32_t pvw_ntt_forward(pvw_ctx* c, uint64_t* polys, size_t count) {
  if (!c || (!polys && count)) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  if (count == 0) return PVW_OK;
  PVW_TRY(ensure_device(c));
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  const size_t bytes = count * c->poly() * 8;
  int32_t rc = ws_scratch(w, bytes);
  if (rc == PVW_OK) {
    if (hipMemcpyAsync(w->scratch, polys, bytes, hipMemcpyHostToDevice, w->stream) != hipSuccess) rc = fail(PVW_ERR_INTERNAL, "H2D failed");
    if (rc == PVW_OK) {
      ProfScope ps(c, "ntt", w->stream);
      if (launch_ntt((u64*)w->scratch, count, false, c->dt, c->L, c->l, w->stream) != hipSuccess) rc = fail(PVW_ERR_INTERNAL, "ntt launch failed");
    }
    if (rc == PVW_OK && (hipMemcpyAsync(polys, w->scratch, bytes, hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
                         hipStreamSynchronize(w->stream) != hipSuccess)) rc = fail(PVW_ERR_INTERNAL, "D2H failed");
  }
  ws_release(c, w);
  return rc;
}
int32_t pvw_ntt_inverse(pvw_ctx* c, uint64_t* polys, size_t count) {
  if (!c || (!polys && count)) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  if (count == 0) return PVW_OK;
  PVW_TRY(ensure_device(c));
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  const size_t bytes = count * c->poly() * 8;
  int32_t rc = ws_scratch(w, bytes);
  if (rc == PVW_OK) {
    if (hipMemcpyAsync(w->scratch, polys, bytes, hipMemcpyHostToDevice, w->stream) != hipSuccess) rc = fail(PVW_ERR_INTERNAL, "H2D failed");
    if (rc == PVW_OK) {
      ProfScope ps(c, "intt", w->stream);
      if (launch_ntt((u64*)w->scratch, count, true, c->dt, c->L, c->l, w->stream) != hipSuccess) rc = fail(PVW_ERR_INTERNAL, "intt launch failed");
    }
    if (rc == PVW_OK && (hipMemcpyAsync(polys, w->scratch, bytes, hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
                         hipStreamSynchronize(w->stream) != hipSuccess)) rc = fail(PVW_ERR_INTERNAL, "D2H failed");
  }
  ws_release(c, w);
  return rc;
}

static int32_t small_to_poly_impl(pvw_ctx* c, const int64_t* coeffs, const uint64_t* scalar, size_t count,
                                  uint64_t* polys, uint32_t repr) {
  PVW_TRY(check_repr(repr));
  if (count == 0) return PVW_OK;
  PVW_TRY(ensure_device(c));
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  const size_t inb = count * c->l * 8, outb = count * c->poly() * 8, scb = scalar ? count * 8 : 0;
  const size_t off_out = (inb + 255) & ~(size_t)255, off_sc = off_out + ((outb + 255) & ~(size_t)255);
  int32_t rc = ws_scratch(w, off_sc + scb + 256);
  if (rc == PVW_OK) {
    char* base = (char*)w->scratch;
    if (hipMemcpyAsync(base, coeffs, inb, hipMemcpyHostToDevice, w->stream) != hipSuccess) rc = fail(PVW_ERR_INTERNAL, "H2D failed");
    if (rc == PVW_OK && scalar && hipMemcpyAsync(base + off_sc, scalar, scb, hipMemcpyHostToDevice, w->stream) != hipSuccess) rc = fail(PVW_ERR_INTERNAL, "H2D failed");
    if (rc == PVW_OK) {
      ProfScope ps(c, "prep", w->stream);
      if (launch_prep((const i64*)base, scalar ? (const u64*)(base + off_sc) : nullptr, (u64*)(base + off_out),
                      c->poly(), c->l, (u32)count, repr == PVW_REPR_NTT, c->dt, c->L, c->l, w->stream) != hipSuccess)
        rc = fail(PVW_ERR_INTERNAL, "prep launch failed");
    }
    if (rc == PVW_OK && (hipMemcpyAsync(polys, base + off_out, outb, hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
                         hipStreamSynchronize(w->stream) != hipSuccess)) rc = fail(PVW_ERR_INTERNAL, "D2H failed");
  }
  ws_release(c, w);
  return rc;
}
int32_t pvw_small_to_poly(pvw_ctx* c, const int64_t* coeffs, size_t count, uint64_t* polys, uint32_t repr) {
  if (!c || ((!coeffs || !polys) && count)) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  return small_to_poly_impl(c, coeffs, nullptr, count, polys, repr);
}
int32_t pvw_encode_scalar(const pvw_ctx* cc, int64_t scalar, uint64_t* poly_out, uint32_t repr) {
  pvw_ctx* c = const_cast<pvw_ctx*>(cc);
  if (!c || !poly_out) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  std::vector<int64_t> zero(c->l, 0);
  uint64_t s = (uint64_t)scalar;
  return small_to_poly_impl(c, zero.data(), &s, 1, poly_out, repr);
}

// the device's address for [p, p + bytes) if p is host memory the GPU can write (hipHostMalloc / hipHostRegister), else NULL
static void* device_alias(void* p, size_t bytes) {
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }   // pageable memory: not an error
  if (at.type != hipMemoryTypeHost || !at.devicePointer) return nullptr;
  hipPointerAttribute_t last{};
  if (hipPointerGetAttributes(&last, (char*)p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (last.type != hipMemoryTypeHost || (char*)last.devicePointer - (char*)at.devicePointer != (ptrdiff_t)(bytes - 1)) return nullptr;
  return at.devicePointer;
}

// host memory the device can read and write directly (pinned, mapped): output buffers placed here receive pvw_encrypt's
// ciphertexts without a copy
int32_t pvw_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return fail(PVW_ERR_INTERNAL, "no HIP device available");
  PVW_HIP(hipHostMalloc(out, bytes, hipHostMallocMapped | hipHostMallocPortable));
  return PVW_OK;
}
int32_t pvw_host_free(void* p) {
  if (p) PVW_HIP(hipHostFree(p));
  return PVW_OK;
}

// ------------------------------------------------------------------------ encrypt
static int32_t encrypt_checks(pvw_ctx* c, size_t num_scalars, const pvw_randomness_t* rnd, uint32_t out_repr) {
  PVW_TRY(check_repr(out_repr));
  if (!rnd) return fail(PVW_ERR_INVALID_PARAMETERS, "randomness is NULL");
  if (num_scalars != c->n) {                                                        // encryption.rs:109-115
    char buf[96];
    snprintf(buf, sizeof buf, "Must provide exactly n=%u scalars, got %zu", c->n, num_scalars);
    return fail(PVW_ERR_INVALID_PARAMETERS, buf);
  }
  if (c->num_keys < c->party_hi)                                                    // :117-121
    return fail(PVW_ERR_INVALID_PARAMETERS, "Global public key is not complete (missing party keys)");
  if (!c->crs_loaded) return fail(PVW_ERR_CRS, "CRS not loaded");
  int32_t ok = 0;
  pvw_ctx_verify_correctness_condition(c, &ok);
  if (!ok)                                                                          // :124-128
    return fail(PVW_ERR_INVALID_PARAMETERS, "Parameters do not satisfy correctness condition - decryption may fail");
  if (rnd->mode == PVW_RND_EXPLICIT) {
    if (!rnd->r || (!rnd->e1 && c->rowsA()) || (!rnd->e2 && c->rowsB()))
      return fail(PVW_ERR_INVALID_PARAMETERS, "explicit randomness pointers are NULL");
  } else if (rnd->mode != PVW_RND_SEED) {
    return fail(PVW_ERR_INVALID_PARAMETERS, "unknown randomness mode");
  }
  return PVW_OK;
}

// the three polynomial families of one encrypt (encryption.rs:135-154 r, :161-167 e1, :195-196
// encode + e2) as prologue jobs [3*slot, 3*slot+3) seeded by key `slot` of the batch:
// r -> r-hat [L][k][l]; NTT(e1) -> c1 rows; NTT(e2) + scalar*g-hat -> c2 rows (the MAC adds onto them)
static int32_t fill_encrypt_jobs(pvw_ctx* c, PrologueBatch& pb, u32 slot, u32 /*unused*/, const pvw_randomness_t* rnd,
                                 const u64* d_scalars, u64* rhat, u64* d_c1, u64* d_c2) {
  const u32 k = c->k, l = c->l;
  const size_t P = c->poly();
  PrologueJob& jr = pb.job[3 * slot];
  PrologueJob& j1 = pb.job[3 * slot + 1];
  PrologueJob& j2 = pb.job[3 * slot + 2];
  jr = PrologueJob{}; j1 = PrologueJob{}; j2 = PrologueJob{};
  PVW_TRY(cbd_job(c->variance, jr.sj));
  jr.sj.domain = DOM_R; jr.sj.index0 = 0; jr.sj.count = k;
  j1.sj.kind = SAMPLE_UNIFORM; j1.sj.domain = DOM_E1; j1.sj.index0 = c->c1_lo; j1.sj.count = c->rowsA(); j1.sj.bound = c->b1;
  j2.sj.kind = SAMPLE_UNIFORM; j2.sj.domain = DOM_E2; j2.sj.index0 = c->party_lo; j2.sj.count = c->rowsB(); j2.sj.bound = c->b2;
  if (rnd->mode == PVW_RND_EXPLICIT) {
    jr.explicit_coeffs = rnd->r;
    j1.explicit_coeffs = rnd->e1 + (size_t)c->c1_lo * l;
    j2.explicit_coeffs = rnd->e2 + (size_t)c->party_lo * l;
  }
  jr.out = rhat; jr.stride_poly = l; jr.stride_limb = (size_t)k * l;
  j1.out = d_c1; j1.stride_poly = P; j1.stride_limb = l;
  j2.out = d_c2; j2.stride_poly = P; j2.stride_limb = l; j2.scalars = d_scalars + c->party_lo;
  jr.key_idx = j1.key_idx = j2.key_idx = slot;
  pb.key[slot] = make_key(rnd->seed);
  return PVW_OK;
}

// all pointers are device pointers; explicit r/e1/e2 are GLOBAL arrays ([k][l], [k][l], [n][l])
// out_c1 / out_c2 != NULL: the MAC stores its results there (device-visible HOST memory of a caller whose buffers are
// pinned) while the addends stay in d_c1 / d_c2; NTT-domain output only
static int32_t encrypt_enqueue(pvw_ctx* c, Workspace* w, const u64* d_scalars, const pvw_randomness_t* rnd,
                               u64* d_c1, u64* d_c2, uint32_t out_repr, hipStream_t s, u64* out_c1 = nullptr, u64* out_c2 = nullptr) {
  const u32 k = c->k, l = c->l, L = c->L, rA = c->rowsA(), rB = c->rowsB();
  if (!out_c1) out_c1 = d_c1;
  if (!out_c2) out_c2 = d_c2;
  if (out_repr == PVW_REPR_POWER && (out_c1 != d_c1 || out_c2 != d_c2)) return fail(PVW_ERR_INTERNAL, "direct output is NTT-domain only");
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  (void)hipGetLastError();
  // first call after a matrix change (and no pvw_prepare since): builds the packed copies -- allocates and synchronises
  const u32 width = ensure_packed(c, s, !capturing);
  PrologueBatch pb{};
  PVW_TRY(fill_encrypt_jobs(c, pb, 0, 0, rnd, d_scalars, w->rhat, d_c1, d_c2));
  // l <= 16: the addends travel in COMPACT form -- the prologue transforms r only and leaves the sampled e1 / e2 coefficients as
  // they are (8 l bytes per row instead of 8 L l written and read back); the MAC workgroups transform their own rows' e and add
  // m g-hat (MacSection::e_small, mac_small_addend).  Explicit randomness: the caller's e1 / e2 arrays ARE the compact form.
  // (Round 3 also measured five forms of making r-hat and / or the addends inside the MAC launch instead of in a launch in front
  // of it: none was faster -- profiles/r03_front_ab.txt.)  Tuning build: PVW_MAC_COMPACT=0 the round-2 form (full addends).
  const bool compact = l <= 16 && w->esmall && PVW_ENV_INT("PVW_MAC_COMPACT", 1) != 0;
  const i64 *es1 = nullptr, *es2 = nullptr;
  if (compact) {
    if (rnd->mode == PVW_RND_EXPLICIT) {
      es1 = pb.job[1].explicit_coeffs;
      es2 = pb.job[2].explicit_coeffs;
      pb.njobs = 1;                                   // r only
    } else {
      pb.job[1].raw_out = w->esmall;
      pb.job[2].raw_out = w->esmall + (size_t)rA * l;
      es1 = pb.job[1].raw_out;
      es2 = pb.job[2].raw_out;
      pb.njobs = 3;
    }
  } else {
    pb.njobs = 3;
  }
  {
    ProfScope ps(c, "prologue", s);
    PVW_HIP(launch_prologue(pb, c->dt, L, l, s));
  }
  {
    ProfScope ps(c, "mac_rows", s);
    MacSection a{width ? c->pkA : c->dA, compact ? nullptr : d_c1, out_c1, rA, 0}, b{width ? c->pkB : c->dB, compact ? nullptr : d_c2, out_c2, rB, 0};
    if (compact) {
      a.e_small = es1;
      b.e_small = es2;
      b.scalars = d_scalars + c->party_lo;
    }
    if (width) PVW_HIP(launch_mac_rows_packed(a, b, w->rhat, c->dt, k, L, l, width, s));   // the same sums over the packed copy
    else PVW_HIP(launch_mac_rows(a, b, w->rhat, c->dt, k, L, l, s));                       // crs.rs:188-201, encryption.rs:177-200
  }
  if (out_repr == PVW_REPR_POWER) {
    ProfScope ps(c, "intt", s);
    PVW_HIP(launch_ntt(d_c1, rA, true, c->dt, L, l, s));
    PVW_HIP(launch_ntt(d_c2, rB, true, c->dt, L, l, s));
  }
  return PVW_OK;
}

// NOTE (measured, round 1): overlapping the prologue of call i with the MAC of call i-1 on a second
// stream was tried and reverted -- the cross-stream event dependencies cost more (step 242 us vs
// 207 us at config 3) than the ~14 us prologue they hide.  Single-stream, in-order is the fast path.
// Round 3, again, with nothing but the seed feeding the prologue (compact addends: it no longer reads the scalars), two
// sets of (r-hat, addends) and one wait per stream and call: the prologue does run under the previous MAC (26 us there
// instead of 11), but the MAC behind the cross-stream wait starts as late as it did behind the prologue: step = MAC +
// 7.6-7.9 us against MAC + 7.8-8.3 us in order (same box, 183-184 us MACs).  Not kept.
int32_t pvw_encrypt_device(pvw_ctx* c, const uint64_t* d_scalars, size_t num_scalars, const pvw_randomness_t* rnd,
                           uint64_t* d_c1, uint64_t* d_c2, uint32_t out_repr, void* stream) {
  if (!c || !d_scalars || (!d_c1 && c->rowsA()) || (!d_c2 && c->rowsB())) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(encrypt_checks(c, num_scalars, rnd, out_repr));
  PVW_TRY(ensure_device(c));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  Workspace* w;
  PVW_TRY(ws_for_stream(c, s, &w));
  return encrypt_enqueue(c, w, d_scalars, rnd, d_c1, d_c2, out_repr, s);
}

int32_t pvw_encrypt(pvw_ctx* c, const uint64_t* scalars, size_t num_scalars, const pvw_randomness_t* rnd,
                    uint64_t* c1_out, uint64_t* c2_out, uint32_t out_repr) {
  if (!c || !scalars || !c1_out || !c2_out) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(encrypt_checks(c, num_scalars, rnd, out_repr));
  PVW_TRY(ensure_device(c));
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  int32_t rc = ws_host_buffers(c, w);
  const size_t l = c->l, k = c->k, P = c->poly();
  pvw_randomness_t dr = *rnd;
  if (rc == PVW_OK && rnd->mode == PVW_RND_EXPLICIT) {
    // upload the explicit small polynomials as global arrays
    const size_t words = (2 * k + c->n) * l;
    rc = ws_scratch(w, words * 8);
    if (rc == PVW_OK) {
      i64* base = (i64*)w->scratch;
      if (hipMemcpyAsync(base, rnd->r, k * l * 8, hipMemcpyHostToDevice, w->stream) != hipSuccess ||
          hipMemcpyAsync(base + k * l, rnd->e1, k * l * 8, hipMemcpyHostToDevice, w->stream) != hipSuccess ||
          hipMemcpyAsync(base + 2 * k * l, rnd->e2, (size_t)c->n * l * 8, hipMemcpyHostToDevice, w->stream) != hipSuccess)
        rc = fail(PVW_ERR_INTERNAL, "H2D failed");
      dr.r = base;
      dr.e1 = base + k * l;
      dr.e2 = base + 2 * k * l;
    }
  }
  if (rc == PVW_OK && hipMemcpyAsync(w->scalars, scalars, (size_t)c->n * 8, hipMemcpyHostToDevice, w->stream) != hipSuccess)
    rc = fail(PVW_ERR_INTERNAL, "H2D failed");
  // Output buffers the device can write (pvw_host_alloc, or memory the caller pinned / registered): the MAC stores c1 / c2
  // straight into them, 64 bytes per (row, limb) as its workgroups finish -- the 4.7 MB of config 3 cross PCIe under the
  // kernel instead of after it.  Pageable buffers take the copy.
  u64 *dir1 = nullptr, *dir2 = nullptr;
  if (rc == PVW_OK && out_repr == PVW_REPR_NTT) {
    dir1 = (u64*)device_alias(c1_out, (size_t)c->k * P * 8);
    dir2 = (u64*)device_alias(c2_out, (size_t)c->n * P * 8);
    if (!dir1 || !dir2) dir1 = dir2 = nullptr;
  }
  if (rc == PVW_OK) rc = encrypt_enqueue(c, w, w->scalars, &dr, w->c1, w->c2, out_repr, w->stream,
                                         dir1 ? dir1 + (size_t)c->c1_lo * P : nullptr, dir2 ? dir2 + (size_t)c->party_lo * P : nullptr);
  if (rc == PVW_OK && !dir1 &&
      (hipMemcpyAsync(c1_out + (size_t)c->c1_lo * P, w->c1, (size_t)c->rowsA() * P * 8, hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
       hipMemcpyAsync(c2_out + (size_t)c->party_lo * P, w->c2, (size_t)c->rowsB() * P * 8, hipMemcpyDeviceToHost, w->stream) != hipSuccess))
    rc = fail(PVW_ERR_INTERNAL, "D2H failed");
  if (hipStreamSynchronize(w->stream) != hipSuccess && rc == PVW_OK) rc = fail(PVW_ERR_INTERNAL, "stream sync failed");
  ws_release(c, w);
  return rc;
}

// ------------------------------------------------------------------------ multi-dealer encrypt
// encrypt_all_party_shares (encryption.rs:253-286): dealer d encrypts scalars[d][0..n) with its own
// randomness (seed d).  Groups of 4 dealers share one pass over A-hat / B-hat (mac_rows_multi).
// Device layout: d_scalars [D][n]; d_c1 [D][rowsA][L][l]; d_c2 [D][rowsB][L][l].
static int32_t encrypt_multi_enqueue(pvw_ctx* c, Workspace* w, const u64* d_scalars, const uint8_t* seeds,
                                     size_t D, u64* d_c1, u64* d_c2, uint32_t out_repr, hipStream_t s) {
  const u32 k = c->k, l = c->l, L = c->L, rA = c->rowsA(), rB = c->rowsB();
  const size_t P = c->poly();
  const int gemm_min = (int)PVW_ENV_INT("PVW_GEMM_MIN_DEALERS", 3);   // tuning build: read per call (tests switch it); measured at config 3: 2 dealers 0.23 ms on the VALU vs 0.25 here, 4 dealers 0.43 vs 0.26, 6 dealers 0.67 vs 0.31
  const bool use_gemm = gemm_min > 0 && D >= (size_t)gemm_min;
  if (use_gemm) {
    PVW_TRY(ws_gemm_buffers(c, w));
    PVW_TRY(ensure_xm(c, s));
  }
  const size_t group = use_gemm ? (size_t)16 * gemm_vb() : 4;
  // matrix-core passes: the c2 finish pass draws e2 and encodes the scalars itself (tuning build: PVW_FUSED_E2=0 the prologue does)
  const bool fused_e2 = use_gemm && l <= 32 && PVW_ENV_INT("PVW_FUSED_E2", 1) != 0;
  u64* vh = use_gemm ? w->vhat16 : w->rhat;
  for (size_t d0 = 0; d0 < D; d0 += group) {
    const u32 nv = (u32)((D - d0) < group ? (D - d0) : group);
    // prologue: the (r, e1, e2) families of dealer d0 replicated over the nv dealers of this pass (up to 64 keys
    // per launch): r-hat_d -> vh[v], NTT(e1), NTT(e2) + m*g-hat -> output planes
    for (u32 v0 = 0; v0 < nv; v0 += PVW_MAX_PROLOGUE_KEYS) {
      const u32 cnt = (nv - v0) < PVW_MAX_PROLOGUE_KEYS ? (nv - v0) : PVW_MAX_PROLOGUE_KEYS;
      const size_t d = d0 + v0;
      PrologueBatch pb{};
      pvw_randomness_t rnd{};
      rnd.mode = PVW_RND_SEED;
      memcpy(rnd.seed, seeds + d * 32, 32);
      PVW_TRY(fill_encrypt_jobs(c, pb, 0, 0, &rnd, d_scalars + d * c->n, vh + (size_t)v0 * k * P,
                                d_c1 + d * rA * P, d_c2 + d * rB * P));
      for (u32 x = 0; x < cnt; ++x) pb.key[x] = make_key(seeds + (d + x) * 32);
      pb.job[0].rep_key = pb.job[1].rep_key = pb.job[2].rep_key = 1;
      pb.job[0].rep_out = (size_t)k * P;                       // r-hat vectors
      pb.job[1].rep_out = (size_t)rA * P;                      // c1 planes
      pb.job[2].rep_out = (size_t)rB * P;                      // c2 planes
      pb.job[2].rep_scalars = c->n;
      pb.njobs = fused_e2 ? 2 : 3;                              // fused: e2 + m g-hat are made by the c2 finish pass
      pb.reps = cnt;
      ProfScope ps(c, "prologue", s);
      PVW_HIP(launch_prologue(pb, c->dt, L, l, s));
    }
    u64* c1g = d_c1 + d0 * rA * P;
    u64* c2g = d_c2 + d0 * rB * P;
    if (use_gemm) {
      {
        ProfScope ps(c, "vec_digits", s);
        PVW_HIP(launch_vec_digits(vh, (size_t)k * P, w->yd, w->sy, nv, k, L, l, c->dt, s, 0, 0, c->xm_bytes));
      }
      ProfScope ps(c, "gemm_digits", s);
      GemmSection a{c->xmA, c1g, c1g, w->gtmpA, rA, 0, 0}, b{c->xmB, c2g, c2g, w->gtmpB, rB, 0, 0};
      std::vector<GemmErrSource> es;
      if (fused_e2) {
        // e2_d[i] (encryption.rs:195-196): dealer d's key, stream DOM_E2 / party index, uniform in [-b2, b2]; + m_{d,i} g-hat
        for (u32 v0 = 0; v0 < nv; v0 += PVW_MAX_PROLOGUE_KEYS) {
          GemmErrSource e{};
          e.span = (nv - v0) < PVW_MAX_PROLOGUE_KEYS ? (nv - v0) : PVW_MAX_PROLOGUE_KEYS;
          for (u32 x = 0; x < e.span; ++x) e.key[x] = make_key(seeds + (d0 + v0 + x) * 32);
          e.key_v = 1;
          e.domain = DOM_E2; e.index0 = c->party_lo; e.index_row = 1; e.index_v = 0; e.bound = c->b2;
          e.scalars = d_scalars + d0 * c->n + c->party_lo; e.scalar_v = c->n;
          es.push_back(e);
        }
        b.addend = nullptr;
      }
      PVW_HIP(launch_gemm_digits(a, b, w->yd, w->sy, c->dt, k, L, l, nv, (size_t)rA * P, (size_t)rB * P, s, nullptr, fused_e2 ? es.data() : nullptr, c->xm_bytes));
    } else {
      ProfScope ps(c, "mac_rows_multi", s);
      MacSection a{c->dA, c1g, c1g, rA, 0}, b{c->dB, c2g, c2g, rB, 0};
      MultiVec mv{vh, (size_t)k * P, (size_t)rA * P, (size_t)rB * P, nv};
      PVW_HIP(launch_mac_rows_multi(a, b, mv, c->dt, k, L, l, s));
    }
  }
  if (out_repr == PVW_REPR_POWER) {
    ProfScope ps(c, "intt", s);
    PVW_HIP(launch_ntt(d_c1, D * rA, true, c->dt, L, l, s));
    PVW_HIP(launch_ntt(d_c2, D * rB, true, c->dt, L, l, s));
  }
  return PVW_OK;
}

static int32_t encrypt_multi_checks(pvw_ctx* c, size_t D, size_t per_dealer, uint32_t out_repr) {
  PVW_TRY(check_repr(out_repr));
  if (D == 0) return fail(PVW_ERR_INVALID_PARAMETERS, "no dealers");
  if (per_dealer != c->n) {                                                          // encryption.rs:264-274
    char buf[96];
    snprintf(buf, sizeof buf, "Dealer provided %zu shares but needs %u", per_dealer, c->n);
    return fail(PVW_ERR_INVALID_PARAMETERS, buf);
  }
  if (c->num_keys < c->party_hi)
    return fail(PVW_ERR_INVALID_PARAMETERS, "Global public key is not complete (missing party keys)");
  if (!c->crs_loaded) return fail(PVW_ERR_CRS, "CRS not loaded");
  int32_t ok = 0;
  pvw_ctx_verify_correctness_condition(c, &ok);
  if (!ok) return fail(PVW_ERR_INVALID_PARAMETERS, "Parameters do not satisfy correctness condition - decryption may fail");
  return PVW_OK;
}

int32_t pvw_encrypt_multi_device(pvw_ctx* c, const uint64_t* d_scalars, size_t num_dealers, size_t scalars_per_dealer,
                                 const uint8_t* seeds, uint64_t* d_c1, uint64_t* d_c2, uint32_t out_repr, void* stream) {
  if (!c || !d_scalars || !seeds || (!d_c1 && c->rowsA()) || (!d_c2 && c->rowsB())) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(encrypt_multi_checks(c, num_dealers, scalars_per_dealer, out_repr));
  PVW_TRY(ensure_device(c));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  Workspace* w;
  PVW_TRY(ws_for_stream(c, s, &w));
  return encrypt_multi_enqueue(c, w, d_scalars, seeds, num_dealers, d_c1, d_c2, out_repr, s);
}

int32_t pvw_encrypt_multi(pvw_ctx* c, const uint64_t* scalars, size_t num_dealers, size_t scalars_per_dealer,
                          const uint8_t* seeds, uint64_t* c1_out, uint64_t* c2_out, uint32_t out_repr) {
  if (!c || !scalars || !seeds || !c1_out || !c2_out) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(encrypt_multi_checks(c, num_dealers, scalars_per_dealer, out_repr));
  PVW_TRY(ensure_device(c));
  const size_t P = c->poly(), rA = c->rowsA(), rB = c->rowsB(), n = c->n;
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  // dealers per pass: bounded staging (<= ~512 MiB of ciphertext)
  size_t per = ((size_t)512 << 20) / ((rA + rB) * P * 8 + n * 8);
  if (per < 4) per = 4;
  per &= ~(size_t)3;
  if (per > num_dealers) per = num_dealers;
  const size_t b_sc = (per * n * 8 + 255) & ~(size_t)255;
  const size_t b_c1 = (per * rA * P * 8 + 255) & ~(size_t)255;
  const size_t b_c2 = (per * rB * P * 8 + 255) & ~(size_t)255;
  int32_t rc = ws_scratch(w, b_sc + b_c1 + b_c2);
  for (size_t d0 = 0; rc == PVW_OK && d0 < num_dealers; d0 += per) {
    const size_t cnt = (num_dealers - d0) < per ? (num_dealers - d0) : per;
    char* base = (char*)w->scratch;
    u64* d_sc = (u64*)base;
    u64* d_c1 = (u64*)(base + b_sc);
    u64* d_c2 = (u64*)(base + b_sc + b_c1);
    if (hipMemcpyAsync(d_sc, scalars + d0 * n, cnt * n * 8, hipMemcpyHostToDevice, w->stream) != hipSuccess) { rc = fail(PVW_ERR_INTERNAL, "H2D failed"); break; }
    rc = encrypt_multi_enqueue(c, w, d_sc, seeds + d0 * 32, cnt, d_c1, d_c2, out_repr, w->stream);
    for (size_t d = 0; rc == PVW_OK && d < cnt; ++d) {
      // a sharded context writes its rows at their global positions inside each dealer's block
      if (hipMemcpyAsync(c1_out + ((d0 + d) * c->k + c->c1_lo) * P, d_c1 + d * rA * P, rA * P * 8, hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
          hipMemcpyAsync(c2_out + ((d0 + d) * n + c->party_lo) * P, d_c2 + d * rB * P, rB * P * 8, hipMemcpyDeviceToHost, w->stream) != hipSuccess)
        rc = fail(PVW_ERR_INTERNAL, "D2H failed");
    }
    if (rc == PVW_OK && hipStreamSynchronize(w->stream) != hipSuccess) rc = fail(PVW_ERR_INTERNAL, "stream sync failed");
  }
  ws_release(c, w);
  return rc;
}

// ------------------------------------------------------------------------ decode (host, integers)
static BigInt center(const BigInt& v, const pvw_ctx* c) {              // decryption.rs:140-152
  return v > c->halfQ ? v - c->Q : v;
}
static BigInt crt_lift(const pvw_ctx* c, const uint64_t* poly, u32 coeff) {
  BigInt acc;
  for (u32 i = 0; i < c->L; ++i) {
    u64 t = mulmod(poly[(size_t)i * c->l + coeff], c->crt_inv[i], c->mods[i]);
    acc = acc + c->crt_qi[i] * BigInt(t);
  }
  return acc % c->Q;
}
static uint64_t decode_one(const pvw_ctx* c, const uint64_t* noisy) {   // decryption.rs:10-58
  const u32 l = c->l;
  const BigInt &Q = c->Q, &D = c->delta;
  std::vector<BigInt> z(l), tmp(l), noise(l);
  for (u32 j = 0; j < l; ++j) z[j] = center(crt_lift(c, noisy, j), c);                  // :109-137
  for (u32 i = 0; i + 1 < l; ++i) tmp[i] = (z[i] * D - z[i + 1]).mod_floor(Q);          // :19-27
  BigInt last = tmp[0];
  for (u32 i = 1; i + 1 < l; ++i) last = (last * D + tmp[i]).mod_floor(Q);             // :30-33
  {                                                                                     // reduce_modulo_poly :154-178
    BigInt poly_const = center(last, c);
    BigInt mod_const = center(c->delta_pow.mod_floor(Q), c);
    BigInt reduced = poly_const % mod_const;
    BigInt half = mod_const / BigInt(2);
    if (reduced > half) reduced = reduced - mod_const;
    else if (reduced < -half) reduced = reduced + mod_const;
    tmp[l - 1] = reduced.mod_floor(Q);
  }
  noise[l - 1] = tmp[l - 1];
  const BigInt delta_const = center(D.mod_floor(Q), c);
  const BigInt two_delta = delta_const * BigInt(2);
  for (u32 i = l - 1; i-- > 0;) {                                                       // :44-48, divide_by_delta_rns :180-207
    BigInt p = center((noise[i + 1] - tmp[i]).mod_floor(Q), c);
    BigInt quo;
    if (!delta_const.is_zero()) {
      BigInt twice = p * BigInt(2);
      quo = p.is_negative() ? (twice - delta_const) / two_delta : (twice + delta_const) / two_delta;
    }
    noise[i] = quo.mod_floor(Q);
  }
  BigInt plain = center((-z[0] - noise[0]).mod_floor(Q), c);                            // :51-53
  if (plain.is_negative()) {                                                            // :226-247
    BigInt abs = -plain;
    if (abs <= BigInt(1000)) return 0;
    BigInt pos = (plain + Q) % Q;
    return pos.fits_u64() ? pos.low_u64() : 0;
  }
  return plain.fits_u64() ? plain.low_u64() : 0;
}

int32_t pvw_decode_host(const pvw_ctx* c, const uint64_t* noisy, size_t count, uint64_t* out) {
  if (!c || ((!noisy || !out) && count)) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  const size_t P = c->poly();
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > 32) nt = 32;
  if (count < 64) nt = 1;
  if (nt == 1) {
    for (size_t d = 0; d < count; ++d) out[d] = decode_one(c, noisy + d * P);
    return PVW_OK;
  }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([=]() {
      for (size_t d = t; d < count; d += nt) out[d] = decode_one(c, noisy + d * P);
    });
  for (auto& x : th) x.join();
  return PVW_OK;
}

#if PVW_TUNING
// MEASUREMENT AID (tuning build only, include/pvw_hip_tuning.h): seconds per pass of a read-only kernel with mac_rows' access pattern over the resident public
// key section (B-hat, tiled): what the memory system delivers to this pattern, next to what mac_rows achieves
int32_t pvw_selftest_read_bandwidth(pvw_ctx* c, uint32_t reps, double* seconds_per_pass, uint64_t* bytes_per_pass) {
  if (!c || !seconds_per_pass || !bytes_per_pass || reps == 0) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(ensure_device(c));
  if (!c->dB || c->rowsB() == 0) return fail(PVW_ERR_INVALID_PARAMETERS, "no public key section resident");
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  const size_t tiles = c->tiled_words(c->rowsB()) / 128;
  const u32 tpw = c->k >= 64 ? (c->k / 4 / 16) * 16 : 16;           // a workgroup covers k tiles, as in mac_rows
  int32_t rc = ws_scratch(w, ((tiles + 63) / 64 + 1) * 8);
  hipEvent_t a = nullptr, b = nullptr;
  if (rc == PVW_OK && (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)) rc = fail(PVW_ERR_INTERNAL, "event");
  if (rc == PVW_OK) {
    bool ok = launch_read_probe(c->dB, tiles, tpw, (u64*)w->scratch, w->stream) == hipSuccess;   // warm-up
    ok = ok && hipEventRecord(a, w->stream) == hipSuccess;
    for (uint32_t i = 0; ok && i < reps; ++i) ok = launch_read_probe(c->dB, tiles, tpw, (u64*)w->scratch, w->stream) == hipSuccess;
    ok = ok && hipEventRecord(b, w->stream) == hipSuccess && hipEventSynchronize(b) == hipSuccess;
    float ms = 0;
    ok = ok && hipEventElapsedTime(&ms, a, b) == hipSuccess;
    if (!ok) rc = fail(PVW_ERR_INTERNAL, "read probe failed");
    *seconds_per_pass = (double)ms * 1e-3 / reps;
    *bytes_per_pass = (uint64_t)tiles * 1024;
  }
  if (a) hipEventDestroy(a);
  if (b) hipEventDestroy(b);
  ws_release(c, w);
  return rc;
}
// MEASUREMENT AID: the read probe with U tiles (2U when dbuf) in flight per wave and lds_bytes of dead LDS per
// workgroup (caps the workgroups resident per CU): bandwidth against bytes in flight
int32_t pvw_tuning_read_probe(pvw_ctx* c, uint32_t reps, uint32_t u, uint32_t dbuf, uint32_t lds_bytes, double* seconds_per_pass,
                              uint64_t* bytes_per_pass) {
  if (!c || !seconds_per_pass || !bytes_per_pass || reps == 0) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(ensure_device(c));
  if (!c->dB || c->rowsB() == 0) return fail(PVW_ERR_INVALID_PARAMETERS, "no public key section resident");
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  const size_t tiles = c->tiled_words(c->rowsB()) / 128;
  const u32 tpw = c->k >= 128 ? (c->k / 4 / 32) * 32 : 32;
  int32_t rc = ws_scratch(w, ((tiles + 63) / 64 + 1) * 8);
  hipEvent_t a = nullptr, b = nullptr;
  if (rc == PVW_OK && (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)) rc = fail(PVW_ERR_INTERNAL, "event");
  if (rc == PVW_OK) {
    bool ok = launch_read_probe2(c->dB, tiles, tpw, (u64*)w->scratch, u, (dbuf & 1) != 0, lds_bytes, w->stream, (dbuf >> 1) & 1) == hipSuccess;
    ok = ok && hipEventRecord(a, w->stream) == hipSuccess;
    for (uint32_t i = 0; ok && i < reps; ++i)
      ok = launch_read_probe2(c->dB, tiles, tpw, (u64*)w->scratch, u, (dbuf & 1) != 0, lds_bytes, w->stream, (dbuf >> 1) & 1) == hipSuccess;
    ok = ok && hipEventRecord(b, w->stream) == hipSuccess && hipEventSynchronize(b) == hipSuccess;
    float ms = 0;
    ok = ok && hipEventElapsedTime(&ms, a, b) == hipSuccess;
    if (!ok) rc = fail(PVW_ERR_INTERNAL, "read probe failed");
    *seconds_per_pass = (double)ms * 1e-3 / reps;
    *bytes_per_pass = (uint64_t)tiles * 1024;
  }
  if (a) hipEventDestroy(a);
  if (b) hipEventDestroy(b);
  ws_release(c, w);
  return rc;
}
// MEASUREMENT AID: start / end stamps of the workgroups of the last mac_rows launch made with PVW_MAC_VARIANT=40 / 41
int32_t pvw_tuning_read_stamps(pvw_ctx* c, uint64_t* stamps, uint32_t* hw_id, uint32_t count) {
  if (!c || !stamps || !hw_id) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(ensure_device(c));
  PVW_HIP(hipDeviceSynchronize());
  PVW_HIP(read_stamps(stamps, hw_id, count));
  return PVW_OK;
}
#endif  // PVW_TUNING

// SELF-TEST: one i8 MFMA through the operand maps the digit-GEMM kernels assume (exact integer data)
int32_t pvw_selftest_mfma_i8(pvw_ctx* c, const int8_t* a, const int8_t* b, int32_t* out) {
  if (!c || !a || !b || !out) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(ensure_device(c));
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  int32_t rc = ws_scratch(w, 8192);
  if (rc == PVW_OK) {
    char* base = (char*)w->scratch;
    if (hipMemcpyAsync(base, a, 1024, hipMemcpyHostToDevice, w->stream) != hipSuccess ||
        hipMemcpyAsync(base + 1024, b, 1024, hipMemcpyHostToDevice, w->stream) != hipSuccess ||
        launch_mfma_probe((const signed char*)base, (const signed char*)base + 1024, (int*)(base + 2048), w->stream) != hipSuccess ||
        hipMemcpyAsync(out, base + 2048, 4096, hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
        hipStreamSynchronize(w->stream) != hipSuccess)
      rc = fail(PVW_ERR_INTERNAL, "mfma probe failed");
  }
  ws_release(c, w);
  return rc;
}

// host execution of the fixed-width decode that the GPU runs (pvw_decode.h) -- a SELF-TEST hook so
// the device algorithm can be checked on a machine without a GPU; not used by any product path.
int32_t pvw_selftest_decode_fixed(const pvw_ctx* c, const uint64_t* noisy, size_t count, uint64_t* out) {
  if (!c || ((!noisy || !out) && count)) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  const DecodeTables& t = c->dec_host;
  std::vector<u64> x(t.W + 1), y(t.W), nres(t.L);
  for (size_t d = 0; d < count; ++d)
    out[d] = decode_one_fixed(t, noisy + d * c->poly(), BN{x.data(), 1}, BN{y.data(), 1}, BN{nres.data(), 1});
  return PVW_OK;
}

// SELF-TEST (host only): the short path of decode_chain_kernel restated sequentially with the SAME arithmetic (pvw_decode.h:
// garner_small, small_chain_step, small_top_*): candidates for every chain input, each confirmed on every limb; noise_{l-1}
// guessed and proven; the chain as passes to a fixed point; the plaintext from the two small values.  true + *out when every
// proof holds (the device then never touches its W-word path for this ciphertext); false when any does not (the device
// settles that part the general way; here the caller takes decode_one_fixed for the whole ciphertext -- same result).
static bool decode_one_short(const pvw_ctx* c, const u64* noisy, u64* out) {
  const DecodeTables& t = c->dec_host;
  const u32 L = t.L, l = t.ell, W = t.W;
  if (!t.gar_n || !t.sc_on || !t.hs_on || l < 3 || W < 4) return false;
  auto z = [&](u32 limb, u32 i) -> u64 { return noisy[(size_t)limb * l + i]; };
  auto res = [&](u32 limb, u32 item) -> u64 {               // tmp_item (decryption.rs:19-27) or z_0 (item == l) on one limb
    const u64 q = t.mods[limb].q;
    return item < l ? submod(mulmod_shoup(z(limb, item), t.dmod[limb], t.dmodp[limb], q), z(limb, item + 1), q) : z(limb, 0);
  };
  std::vector<u64> cand((size_t)5 * l);
  for (u32 idx = 0; idx < l; ++idx) {
    const u32 item = idx + 1 < l ? idx : l;
    u64 r[4] = {0, 0, 0, 0};
    for (u32 j = 0; j < t.gar_n; ++j) r[j] = res(j, item);
    u64* cv = &cand[(size_t)5 * idx];
    garner_small(t, r, cv);
    const bool ng = (cv[4] & 1) != 0;
    for (u32 limb = 0; limb < L; ++limb) {                    // small_confirm
      const Mod& m = t.mods[limb];
      const u64* pw = t.pow64 + (size_t)limb * W;
      u128 sum = (u128)cv[0] * pw[0] + (u128)cv[1] * pw[1];
      if (cv[2]) sum += (u128)cv[2] * pw[2];
      if (cv[3]) sum += (u128)cv[3] * pw[3];
      u64 sres = reduce128((u64)sum, (u64)(sum >> 64), m);
      if (ng && sres) sres = m.q - sres;
      if (sres != res(limb, item)) return false;
    }
    cv[4] |= 2;
  }
  // first pass of the chain: every step on a zero input
  std::vector<u64> q(l - 1), nq(l - 1);
  std::vector<char> ng(l - 1), nng(l - 1);
  const SmallVal zero{0, 0, 0, false};
  for (u32 i = 0; i + 1 < l; ++i) {
    bool n;
    if (!small_chain_step(t.sc, zero, &cand[(size_t)5 * i], q[i], n)) return false;
    ng[i] = n;
  }
  // small_top: the guess and its proof on every limb
  SmallVal top;
  if (!small_top_guess(t.sc, &cand[(size_t)5 * (l - 2)], q[l - 2], top)) return false;
  std::vector<u64> e(L);
  for (u32 limb = 0; limb < L; ++limb) {
    const u64* pw = t.pow64 + (size_t)limb * W;
    e[limb] = small_top_quotient(top, z(limb, 0), z(limb, l - 1), t.dpm[limb], t.dpm[L + limb], t.dpm[2 * L + limb], t.dpm[3 * L + limb],
                                 pw[0], pw[1], pw[2], t.mods[limb]);
  }
  for (u32 limb = 0; limb < L; ++limb)
    if (small_top_expected(e[0], t.mods[0].q, t.mods[limb]) != e[limb]) return false;
  // further passes: step i on the previous output of step i + 1 (the top step on the proven noise_{l-1})
  bool settled = false;
  for (int pass = 1; pass < 4 && !settled; ++pass) {
    settled = true;
    for (u32 i = 0; i + 1 < l; ++i) {
      const SmallVal a = i + 2 == l ? top : SmallVal{q[i + 1], 0, 0, ng[i + 1] != 0};
      bool n;
      if (!small_chain_step(t.sc, a, &cand[(size_t)5 * i], nq[i], n)) return false;
      nng[i] = n;
      if (nq[i] != q[i] || nng[i] != ng[i]) settled = false;
    }
    q = nq;
    ng = nng;
  }
  if (!settled) return false;
  // plaintext = -z_0 - noise_0 (:51-53), extract_constant_term_as_u64 (:226-247): both small, so the centred value is the integer
  const u64* z0 = &cand[(size_t)5 * (l - 1)];
  BigInt zv = BigInt::from_words(z0, 4);
  if (z0[4] & 1) zv = -zv;
  BigInt n0(q[0]);
  if (ng[0]) n0 = -n0;
  const BigInt v = -(zv + n0);
  *out = (v.neg || v.mag.size() > 1) ? 0 : (v.mag.empty() ? 0 : v.mag[0]);
  return true;
}
int32_t pvw_selftest_decode_shortcuts(const pvw_ctx* c, const uint64_t* noisy, size_t count, uint64_t* out, uint8_t* short_path) {
  if (!c || ((!noisy || !out) && count)) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  const DecodeTables& t = c->dec_host;
  std::vector<u64> x(t.W + 1), y(t.W), nres(t.L);
  for (size_t d = 0; d < count; ++d) {
    const u64* nz = noisy + d * c->poly();
    const bool took = decode_one_short(c, nz, out + d);
    if (!took) out[d] = decode_one_fixed(t, nz, BN{x.data(), 1}, BN{y.data(), 1}, BN{nres.data(), 1});
    if (short_path) short_path[d] = took ? 1 : 0;
  }
  return PVW_OK;
}

// decode_scalar_pvw_rns on the device, host buffers in and out
int32_t pvw_decode(pvw_ctx* c, const uint64_t* noisy, size_t count, uint64_t* out) {
  if (!c || ((!noisy || !out) && count)) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  if (count == 0) return PVW_OK;
  PVW_TRY(ensure_device(c));
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  const size_t inb = (count * c->poly() * 8 + 255) & ~(size_t)255;
  int32_t rc = ws_scratch(w, inb + count * 8);
  if (rc == PVW_OK) {
    char* base = (char*)w->scratch;
    if (hipMemcpyAsync(base, noisy, count * c->poly() * 8, hipMemcpyHostToDevice, w->stream) != hipSuccess) rc = fail(PVW_ERR_INTERNAL, "H2D failed");
    if (rc == PVW_OK) {
      ProfScope ps(c, "decode", w->stream);
      if (launch_decode((u64*)base, (u64*)(base + inb), count, c->dec_dev, w->stream) != hipSuccess) rc = fail(PVW_ERR_INTERNAL, "decode launch failed");
    }
    if (rc == PVW_OK && (hipMemcpyAsync(out, base + inb, count * 8, hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
                         hipStreamSynchronize(w->stream) != hipSuccess)) rc = fail(PVW_ERR_INTERNAL, "D2H failed");
  }
  ws_release(c, w);
  return rc;
}

int32_t pvw_decode_device(pvw_ctx* c, const uint64_t* d_noisy, size_t count, uint64_t* d_out, void* stream) {
  if (!c || ((!d_noisy || !d_out) && count)) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  if (count == 0) return PVW_OK;
  PVW_TRY(ensure_device(c));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  ProfScope ps(c, "decode", s);
  PVW_HIP(launch_decode(const_cast<u64*>(d_noisy), d_out, count, c->dec_dev, s));    // power basis in: read only
  return PVW_OK;
}

// ------------------------------------------------------------------------ decrypt
// noisy[d] = sum_j s-hat[j] (.) c1s[d][j] - c2col[d]  for D ciphertexts (decryption.rs:257-274): the inner products as one
// launch, cut into ranges of j when that gives the launch enough short workgroups (decrypt_split).  Returns in
// *ntt_domain whether d_noisy still has to be transformed back (change_representation(PowerBasis), :116): with ranges
// the pass that adds them up (decrypt_finish) does it; without, the consumer does -- the decode kernel itself
// (launch_decode with the transform tables) or launch_ntt.
static int32_t decrypt_mac_only(pvw_ctx* c, Workspace* w, const u64* d_c1s, const u64* d_c2col, size_t D, u64* d_noisy,
                                hipStream_t s, bool* ntt_domain, bool alone = true, const u64* shat = nullptr) {
  if (!shat) shat = w->rhat;                           // NTT(sk) made by this call (launch_prep); else a resident key's
  const u32 k = c->k, l = c->l, L = c->L;
  const u32 ns = decrypt_split(k, L, l, D);
  if (ns > 1) {
    const size_t need = (size_t)ns * D * c->poly() * 8;
    if (w->dpart_bytes < need) {
      if (w->dpart) { PVW_HIP(hipStreamSynchronize(s)); hipFree(w->dpart); w->dpart = nullptr; w->dpart_bytes = 0; }
      PVW_HIP(hipMalloc((void**)&w->dpart, need));
      w->dpart_bytes = need;
    }
  }
  {
    ProfScope ps(c, "decrypt_mac", s);
    PVW_HIP(launch_decrypt_mac(d_c1s, shat, d_c2col, d_noisy, c->dt, k, L, l, D, s, w->dpart, ns, alone));
  }
  *ntt_domain = ns <= 1;
  if (ns > 1) {
    ProfScope ps(c, "intt", s);
    PVW_HIP(launch_decrypt_finish(w->dpart, ns, d_c2col, d_noisy, c->dt, L, l, D, s));
  }
  return PVW_OK;
}
static int32_t decrypt_mac_intt(pvw_ctx* c, Workspace* w, const u64* d_c1s, const u64* d_c2col, size_t D, u64* d_noisy,
                                hipStream_t s) {
  bool ntt_domain = false;
  PVW_TRY(decrypt_mac_only(c, w, d_c1s, d_c2col, D, d_noisy, s, &ntt_domain));
  if (ntt_domain) {
    ProfScope ps(c, "intt", s);
    PVW_HIP(launch_ntt(d_noisy, D, true, c->dt, c->L, c->l, s));
  }
  return PVW_OK;
}

// ntt_domain == NULL: d_noisy comes back in power basis; otherwise *ntt_domain says whether the caller (the decode) still has to
// transform it back
static int32_t decrypt_enqueue(pvw_ctx* c, Workspace* w, const i64* d_sk, u64* d_c1s, u64* d_c2col, size_t D,
                               uint32_t in_repr, u64* d_noisy, hipStream_t s, bool inputs_mutable, bool* ntt_domain = nullptr) {
  const u32 k = c->k, l = c->l, L = c->L;
  const size_t P = c->poly();
  {
    ProfScope ps(c, "prep", s);
    // NTT(sk[j]) in the ciphertext layout [k][L][l]   (secret_key.rs:98-112, once per call)
    PVW_HIP(launch_prep(d_sk, nullptr, w->rhat, P, l, k, true, c->dt, L, l, s));
  }
  if (in_repr == PVW_REPR_POWER) {
    if (!inputs_mutable) return fail(PVW_ERR_INVALID_FORMAT, "power-basis ciphertexts need a mutable device buffer");
    ProfScope ps(c, "ntt", s);
    PVW_HIP(launch_ntt(d_c1s, D * k, false, c->dt, L, l, s));
    PVW_HIP(launch_ntt(d_c2col, D, false, c->dt, L, l, s));
  }
  if (ntt_domain) return decrypt_mac_only(c, w, d_c1s, d_c2col, D, d_noisy, s, ntt_domain);
  return decrypt_mac_intt(c, w, d_c1s, d_c2col, D, d_noisy, s);
}
// NTT(sk) sits in w->rhat while a decrypt runs: cleared on the call's stream behind the last kernel that read it
static int32_t wipe_shat(pvw_ctx* c, Workspace* w, hipStream_t s) {
  ws_mark_secret(w, w->rhat, (size_t)c->k * c->poly() * 8);
  PVW_HIP(ws_wipe_secrets(w, s));
  return PVW_OK;
}

int32_t pvw_decrypt_noisy_device(pvw_ctx* c, const int64_t* d_sk, const uint64_t* d_c1s, const uint64_t* d_c2col,
                                 size_t D, uint32_t in_repr, uint64_t* d_noisy, void* stream) {
  if (!c || !d_sk || ((!d_c1s || !d_c2col || !d_noisy) && D)) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(check_repr(in_repr));
  if (in_repr != PVW_REPR_NTT) return fail(PVW_ERR_INVALID_FORMAT, "device decrypt takes NTT-domain ciphertexts");
  PVW_TRY(ensure_device(c));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  Workspace* w;
  PVW_TRY(ws_for_stream(c, s, &w));
  int32_t rc = decrypt_enqueue(c, w, d_sk, const_cast<u64*>(d_c1s), const_cast<u64*>(d_c2col), D, in_repr, d_noisy, s, false);
  int32_t rw = wipe_shat(c, w, s);
  return rc != PVW_OK ? rc : rw;
}

// decrypt_party_shares with device pointers end to end: <sk, c1> - c2, INTT and gadget decode for D dealer
// ciphertexts; only D x u64 are produced.  Large batches are cut into chunks of about 2 GiB and the
// decode of chunk i (integer-ALU work, a few waves per CU) runs on a helper stream under the HBM-bound MAC of
// chunk i+1; the caller's stream waits for the last decode before the call's work counts as complete.
// A secret key kept on the device in the form the inner products read (NTT(sk[j]) in the ciphertext layout,
// secret_key.rs:98-112): decrypt calls that take one skip the transform of the key and the wipe behind it.  The reference's
// SecretKey lives as long as its owner does and is ZeroizeOnDrop (secret_key.rs:20-30); so does this: pvw_sk_free clears it.
struct pvw_sk {
  pvw_ctx* ctx;
  u64* shat;       // [k][L][l]
  size_t bytes;
};
static int32_t decrypt_batch_core(pvw_ctx* c, const int64_t* d_sk, const u64* key_shat, const uint64_t* d_c1s, const uint64_t* d_c2col,
                                  size_t D, uint32_t in_repr, uint64_t* d_noisy, uint64_t* d_out, void* stream);
int32_t pvw_decrypt_batch_device(pvw_ctx* c, const int64_t* d_sk, const uint64_t* d_c1s, const uint64_t* d_c2col,
                                 size_t D, uint32_t in_repr, uint64_t* d_noisy, uint64_t* d_out, void* stream) {
  if (!c || !d_sk) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  return decrypt_batch_core(c, d_sk, nullptr, d_c1s, d_c2col, D, in_repr, d_noisy, d_out, stream);
}
int32_t pvw_decrypt_batch_device_sk(pvw_ctx* c, const pvw_sk* key, const uint64_t* d_c1s, const uint64_t* d_c2col,
                                    size_t D, uint32_t in_repr, uint64_t* d_noisy, uint64_t* d_out, void* stream) {
  if (!c || !key || !key->shat) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  if (key->ctx != c) return fail(PVW_ERR_INVALID_PARAMETERS, "the key was loaded for another context");
  return decrypt_batch_core(c, nullptr, key->shat, d_c1s, d_c2col, D, in_repr, d_noisy, d_out, stream);
}
int32_t pvw_sk_load(pvw_ctx* c, const int64_t* sk, pvw_sk** out) {
  if (!c || !sk || !out) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  *out = nullptr;
  PVW_TRY(ensure_device(c));
  const size_t k = c->k, l = c->l, P = c->poly();
  pvw_sk* key = new pvw_sk{c, nullptr, k * P * 8};
  i64* stage = nullptr;
  hipError_t e = hipMalloc((void**)&key->shat, key->bytes);
  if (e == hipSuccess) e = hipMalloc((void**)&stage, k * l * 8);
  if (e == hipSuccess) e = hipMemcpyAsync(stage, sk, k * l * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = launch_prep(stage, nullptr, key->shat, P, l, (u32)k, true, c->dt, c->L, c->l, c->stream);
  if (stage) {                                               // the uploaded coefficients do not outlive the call
    hipError_t e2 = hipMemsetAsync(stage, 0, k * l * 8, c->stream);
    if (e == hipSuccess) e = e2;
  }
  hipError_t e3 = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = e3;
  if (stage) hipFree(stage);
  if (e != hipSuccess) {
    if (key->shat) { hipMemset(key->shat, 0, key->bytes); hipFree(key->shat); }
    delete key;
    (void)hipGetLastError();
    return fail(PVW_ERR_INTERNAL, "loading the secret key failed");
  }
  *out = key;
  return PVW_OK;
}
int32_t pvw_sk_free(pvw_sk* key) {
  if (!key) return PVW_OK;
  int32_t rc = PVW_OK;
  if (key->shat) {
    if (key->ctx) (void)hipSetDevice(key->ctx->device);
    if (hipDeviceSynchronize() != hipSuccess || hipMemset(key->shat, 0, key->bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
      rc = fail(PVW_ERR_INTERNAL, "clearing the secret key failed");
    hipFree(key->shat);
  }
  delete key;
  return rc;
}
static int32_t decrypt_batch_core(pvw_ctx* c, const int64_t* d_sk, const u64* key_shat, const uint64_t* d_c1s, const uint64_t* d_c2col,
                                  size_t D, uint32_t in_repr, uint64_t* d_noisy, uint64_t* d_out, void* stream) {
  if (!c || ((!d_c1s || !d_c2col || !d_noisy || !d_out) && D)) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(check_repr(in_repr));
  if (in_repr != PVW_REPR_NTT) return fail(PVW_ERR_INVALID_FORMAT, "device decrypt takes NTT-domain ciphertexts");
  if (D == 0) return fail(PVW_ERR_INVALID_PARAMETERS, "No ciphertexts provided");          // decryption.rs:286-290
  PVW_TRY(ensure_device(c));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  Workspace* w;
  PVW_TRY(ws_for_stream(c, s, &w));
  const u32 k = c->k, l = c->l, L = c->L;
  const size_t P = c->poly();
  // chunks of about 2 GiB of ciphertext (measured: at config 5 in full, 18 GB, overlapping the decode is -7 %;
  // with 0.3 GB chunks the cross-stream events cost more than the decode they hide, +29 %): below 3 GiB in all,
  // one pass on the caller's stream.  PVW_DECRYPT_CHUNK=<dealers> overrides.
  const long chunk_env = PVW_ENV_INT("PVW_DECRYPT_CHUNK", 0);   // tuning build only (read per call)
  const double total_gib = (double)D * k * P * 8 / (double)((size_t)1 << 30);
  size_t chunk = D;
  if (chunk_env >= 64) chunk = (size_t)chunk_env;
  else if (total_gib >= 3.0) chunk = (D + (size_t)(total_gib / 2.0) - 1) / (size_t)(total_gib / 2.0);
  const size_t nch = (D + chunk - 1) / chunk;
  if (!key_shat) {
    ProfScope ps(c, "prep", s);
    PVW_HIP(launch_prep(d_sk, nullptr, w->rhat, P, l, k, true, c->dt, L, l, s));   // NTT(sk[j]) once per call (secret_key.rs:98-112)
  }
  const bool overlap = nch >= 2;
  if (overlap) {
    if (!w->aux) PVW_HIP(hipStreamCreateWithFlags(&w->aux, hipStreamNonBlocking));
    while (w->events.size() < nch + 1) {
      hipEvent_t e;
      PVW_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      w->events.push_back(e);
    }
  }
  bool shat_wiped = false;
  auto chunks = [&]() -> int32_t {
  for (size_t i = 0; i < nch; ++i) {
    const size_t d0 = i * chunk, cnt = (D - d0) < chunk ? (D - d0) : chunk;
    u64* nz = d_noisy + d0 * P;
    bool ntt_domain = false;
    PVW_TRY(decrypt_mac_only(c, w, d_c1s + d0 * k * P, d_c2col + d0 * P, cnt, nz, s, &ntt_domain, !overlap, key_shat));   // decryption.rs:257-274
    hipStream_t ds = s;
    if (overlap) {
      PVW_HIP(hipEventRecord(w->events[i], s));
      PVW_HIP(hipStreamWaitEvent(w->aux, w->events[i], 0));
      ds = w->aux;
    }
    // The decode can transform back while it stages its input (no launch for decryption.rs:116), at the price of 94
    // instead of 79 registers.  Taken while its workgroups (two ciphertexts each) are resident all at once anyway; beyond,
    // and when the decode shares the chip with the next chunk's inner products, the 79-register decode runs behind a
    // transform launch of its own (on the helper stream when there is one).
    if (ntt_domain && (overlap || (cnt + 1) / 2 > (size_t)2 * c->num_cus)) {
      ProfScope pi(c, "intt", ds);
      PVW_HIP(launch_ntt(nz, cnt, true, c->dt, L, l, ds));
      ntt_domain = false;
    }
    ProfScope ps(c, "decode", ds);
    // one pass on one stream: the decode is the last launch to follow the inner products, and clears NTT(sk) on its way
    bool by_decode = false;
    PVW_HIP(launch_decode(nz, d_out + d0, cnt, c->dec_dev, ds, ntt_domain ? &c->dt : nullptr,                    // :116, :10-58
                          overlap || key_shat ? nullptr : w->rhat, overlap || key_shat ? 0 : (size_t)k * P * 8, &by_decode));
    if (by_decode) shat_wiped = true;
  }
  if (overlap) {
    PVW_HIP(hipEventRecord(w->events[nch], w->aux));
    PVW_HIP(hipStreamWaitEvent(s, w->events[nch], 0));
  }
  return PVW_OK;
  };
  int32_t rc = chunks();
  if (rc != PVW_OK) {                       // nothing of a failed call stays queued behind the caller's back
    if (w->aux) hipStreamSynchronize(w->aux);
    hipStreamSynchronize(s);
  }
  int32_t rw = PVW_OK;
  if (key_shat) return rc;                  // a resident key: nothing of it was copied anywhere
  if (rc == PVW_OK && shat_wiped) {         // cleared by the decode launch: recorded as this call's wiped region
    ws_mark_secret(w, w->rhat, (size_t)k * P * 8);
    w->wiped = w->secrets;
    w->secrets.clear();
  } else {
    rw = wipe_shat(c, w, s);                // every decrypt_mac launch above is on `s`
  }
  return rc != PVW_OK ? rc : rw;
}

int32_t pvw_decrypt_batch(pvw_ctx* c, const int64_t* sk, const uint64_t* c1s, const uint64_t* c2col, size_t D,
                          uint32_t in_repr, uint64_t* out_u64, uint64_t* noisy_out) {
  if (!c || !sk) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  if (D == 0) return fail(PVW_ERR_INVALID_PARAMETERS, "No ciphertexts provided");          // decryption.rs:286-290
  if (!c1s || !c2col || !out_u64) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  PVW_TRY(check_repr(in_repr));
  PVW_TRY(ensure_device(c));
  const size_t k = c->k, l = c->l, P = c->poly();
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  // dealers per pass: bounded staging (<= ~1 GiB of ciphertext)
  size_t per = ((size_t)1 << 30) / (k * P * 8);
  if (per == 0) per = 1;
  if (per > D) per = D;
  const size_t b_sk = (k * l * 8 + 255) & ~(size_t)255;
  const size_t b_c1 = (per * k * P * 8 + 255) & ~(size_t)255;
  const size_t b_c2 = (per * P * 8 + 255) & ~(size_t)255;
  const size_t b_out = (per * 8 + 255) & ~(size_t)255;
  int32_t rc = ws_scratch(w, b_sk + b_c1 + 2 * b_c2 + b_out);
  if (rc == PVW_OK) {
    char* base = (char*)w->scratch;
    i64* d_sk = (i64*)base;
    u64* d_c1 = (u64*)(base + b_sk);
    u64* d_c2 = (u64*)(base + b_sk + b_c1);
    u64* d_nz = (u64*)(base + b_sk + b_c1 + b_c2);
    u64* d_out = (u64*)(base + b_sk + b_c1 + 2 * b_c2);
    if (hipMemcpyAsync(d_sk, sk, k * l * 8, hipMemcpyHostToDevice, w->stream) != hipSuccess) rc = fail(PVW_ERR_INTERNAL, "H2D failed");
    for (size_t d0 = 0; rc == PVW_OK && d0 < D; d0 += per) {
      const size_t cnt = (D - d0) < per ? (D - d0) : per;
      if (hipMemcpyAsync(d_c1, c1s + d0 * k * P, cnt * k * P * 8, hipMemcpyHostToDevice, w->stream) != hipSuccess ||
          hipMemcpyAsync(d_c2, c2col + d0 * P, cnt * P * 8, hipMemcpyHostToDevice, w->stream) != hipSuccess) {
        rc = fail(PVW_ERR_INTERNAL, "H2D failed");
        break;
      }
      bool ntt_domain = false;
      rc = decrypt_enqueue(c, w, d_sk, d_c1, d_c2, cnt, in_repr, d_nz, w->stream, true, &ntt_domain);
      if (rc == PVW_OK && ntt_domain && (cnt + 1) / 2 > (size_t)2 * c->num_cus) {      // as in pvw_decrypt_batch_device
        ProfScope pi(c, "intt", w->stream);
        if (launch_ntt(d_nz, cnt, true, c->dt, c->L, c->l, w->stream) != hipSuccess) rc = fail(PVW_ERR_INTERNAL, "intt launch failed");
        ntt_domain = false;
      }
      if (rc == PVW_OK) {
        ProfScope ps(c, "decode", w->stream);
        if (launch_decode(d_nz, d_out, cnt, c->dec_dev, w->stream, ntt_domain ? &c->dt : nullptr) != hipSuccess)
          rc = fail(PVW_ERR_INTERNAL, "decode launch failed");   // decryption.rs:116, :277
      }
      if (rc == PVW_OK && (hipMemcpyAsync(out_u64 + d0, d_out, cnt * 8, hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
                           (noisy_out && hipMemcpyAsync(noisy_out + d0 * P, d_nz, cnt * P * 8, hipMemcpyDeviceToHost, w->stream) != hipSuccess) ||
                           hipStreamSynchronize(w->stream) != hipSuccess))
        rc = fail(PVW_ERR_INTERNAL, "D2H failed");
    }
    // the uploaded coefficients and NTT(sk) do not outlive the call (secret_key.rs:20-30)
    ws_mark_secret(w, d_sk, k * l * 8);
    ws_mark_secret(w, w->rhat, k * P * 8);
    if ((ws_wipe_secrets(w, w->stream) != hipSuccess || hipStreamSynchronize(w->stream) != hipSuccess) && rc == PVW_OK)
      rc = fail(PVW_ERR_INTERNAL, "wipe failed");
  }
  ws_release(c, w);
  return rc;
}

// ------------------------------------------------------------------------ key generation
// b_i = s_i * A + e_i: for every party the k-term inner products over A's COLUMNS, i.e. one
// mac_rows pass over the transposed CRS per party (public_key.rs:111-147, crs.rs:138-171).
// The CRS is transposed once per call into a temporary tiled matrix; groups of 4 parties then
// share one pass over it (mac_rows_multi with s-hat_i in the role of r-hat).
// key generation on the matrix cores with the roles chosen so that the SHARED operand is digitised once:
//   b_p[col] = sum_j s-hat_p[j] * A-hat[j][col] + e_p[col]
// GEMM rows = parties (their s-hat rows are the raw streamed operand, MFMA-tiled per chunk), GEMM vectors = the k
// columns of A-hat (digit tiles built ONCE per call, straight from the CRS in API layout: no transpose), the
// finish pass adds e_p and writes into the tiled B-hat.  Chunks of up to 1024 parties.
static int32_t keygen_gemm_swapped(pvw_ctx* c, Workspace* w, u32 a, u32 b, u32 lo, const int64_t* sk, const int64_t* ek,
                                   const uint8_t* seed) {
  const u32 k = c->k, l = c->l, L = c->L;
  const size_t P = c->poly();
  hipStream_t s = w->stream;
  const u32 chunk = (b - a) < 1024 ? (b - a) : 1024;
  const u32 nb = (k + 15) / 16;                                      // batches of 16 column-vectors
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t b_api = al((size_t)k * k * P * 8);
  const size_t b_yd = al(yd_bytes(16 * nb, k, L, l)), b_sy = al(sy_bytes(16 * nb, L, l));   // whole batches: the last one is read in full
  const size_t b_small = al((size_t)2 * chunk * k * l * 8);                     // sk | ek of one chunk; two of these (double buffer)
  const bool direct = l <= 32;                                        // see the chunk loop
  const size_t b_rows = direct ? 0 : al((size_t)chunk * k * P * 8);   // l = 64 only: s-hat rows | e rows, API layout [p][j or col][P]
  const size_t b_xm = al(xm_words(chunk, k, L, l) * 8);
  const size_t b_tmp = al((size_t)nb * gemm_tmp_words(chunk, L, l) * 8);
  PVW_TRY(ws_scratch(w, b_api + b_yd + b_sy + 2 * b_small + 2 * b_rows + b_xm + b_tmp));
  char* base = (char*)w->scratch;
  u64* d_api = (u64*)base;
  signed char* d_yd = (signed char*)(base + b_api);
  int* d_sy = (int*)(base + b_api + b_yd);
  i64* d_small2[2] = {(i64*)(base + b_api + b_yd + b_sy), (i64*)(base + b_api + b_yd + b_sy + b_small)};
  u64* d_srow = (u64*)(base + b_api + b_yd + b_sy + 2 * b_small);
  u64* d_erow = (u64*)(base + b_api + b_yd + b_sy + 2 * b_small + b_rows);
  u64* d_xm = (u64*)(base + b_api + b_yd + b_sy + 2 * b_small + 2 * b_rows);
  u64* d_tmp = (u64*)(base + b_api + b_yd + b_sy + 2 * b_small + 2 * b_rows + b_xm);
  // everything derived from the secret keys: their coefficients (and explicit key errors), the MFMA-tiled copy of
  // NTT(s), the GEMM intermediate (s A without the error) and, for l = 64, the rows of NTT(s) and of the transformed
  // errors -- cleared by pvw_keygen when the call ends
  ws_mark_secret(w, d_small2[0], 2 * b_small + 2 * b_rows + b_xm + b_tmp);
  // the secret keys of chunk i+1 are uploaded on a helper stream while chunk i computes
  const u32 nchunks = (b - a + chunk - 1) / chunk;
  if (!w->aux) PVW_HIP(hipStreamCreateWithFlags(&w->aux, hipStreamNonBlocking));
  while (w->events.size() < 2 * (size_t)nchunks + 2) {
    hipEvent_t e;
    PVW_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    w->events.push_back(e);
  }
  auto upload = [&](u32 ci, hipStream_t st) -> int32_t {
    const u32 q0 = a + ci * chunk, cn = (b - q0) < chunk ? (b - q0) : chunk;
    const size_t wd = (size_t)cn * k * l;
    i64* dst = d_small2[ci & 1];
    PVW_HIP(hipMemcpyAsync(dst, sk + (size_t)(q0 - lo) * k * l, wd * 8, hipMemcpyHostToDevice, st));
    if (ek) PVW_HIP(hipMemcpyAsync(dst + (size_t)chunk * k * l, ek + (size_t)(q0 - lo) * k * l, wd * 8, hipMemcpyHostToDevice, st));
    return PVW_OK;
  };
  PVW_TRY(upload(0, s));
  // A-hat -> API layout [j][col][limb][slot]; vector `col` is its column: element j at col * P + limb * l + j * (k * P)
  PVW_HIP(launch_untile(c->dA, d_api, k, 0, k, L, l, false, c->dt, s));
  PVW_HIP(launch_vec_digits(d_api, P, d_yd, d_sy, k, k, L, l, c->dt, s, l, (size_t)k * P));
  for (u32 ci = 0; ci < nchunks; ++ci) {
    const u32 p0 = a + ci * chunk;
    const u32 cnt = (b - p0) < chunk ? (b - p0) : chunk;
    i64* d_small = d_small2[ci & 1];
    if (ci > 0) PVW_HIP(hipStreamWaitEvent(s, w->events[2 * ci], 0));          // this chunk's keys have arrived
    ProfScope ps(c, "keygen", s);
    // l <= 32: s-hat_p (secret_key.rs:98-112) goes straight from the uploaded coefficients into the MFMA-tiled raw
    // operand and the key errors e_p (public_key.rs:128-132) are drawn (or read) by the finish pass of the GEMM --
    // no transformed rows of either in memory.  l = 64: both through API-layout rows from one prologue launch.
    GemmSection ga{d_xm, d_erow, d_erow, d_tmp, cnt, 0, 0}, gb{nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    ga.tiled_out = c->dB;
    ga.tiled_row0 = p0 - c->party_lo;
    ga.tiled_swap = 1;
    ga.row_stride = (size_t)k * P;
    if (direct) {
      PVW_HIP(launch_shat_mftile(d_small, d_xm, cnt, k, L, l, c->dt, s));
      GemmErrSource es{};
      if (ek) { es.explicit_coeffs = d_small + (size_t)chunk * k * l; es.coef_row = k; es.coef_v = 1; }
      else es.key[0] = make_key(seed);
      es.domain = DOM_EKEY; es.index0 = p0 * k; es.index_row = k; es.index_v = 1; es.bound = c->b1;
      ga.addend = nullptr;
      ga.out = nullptr;
      // all k columns in one launch (crs.rs:152-168)
      PVW_HIP(launch_gemm_digits(ga, gb, d_yd, d_sy, c->dt, k, L, l, k, P, 0, s, &es));
    } else {
      PrologueBatch pb{};
      if (seed) pb.key[0] = make_key(seed);
      PrologueJob& js = pb.job[0];
      PrologueJob& je = pb.job[1];
      js.sj.count = k; js.explicit_coeffs = d_small; js.rep_coeffs = (size_t)k * l;
      js.out = d_srow; js.stride_poly = P; js.stride_limb = l; js.rep_out = (size_t)k * P;
      je.sj.kind = SAMPLE_UNIFORM; je.sj.domain = DOM_EKEY; je.sj.index0 = p0 * k; je.sj.count = k; je.sj.bound = c->b1;
      je.rep_index0 = k;
      if (ek) { je.explicit_coeffs = d_small + (size_t)chunk * k * l; je.rep_coeffs = (size_t)k * l; }
      je.out = d_erow; je.stride_poly = P; je.stride_limb = l; je.rep_out = (size_t)k * P;
      pb.njobs = 2;
      pb.reps = cnt;
      PVW_HIP(launch_prologue(pb, c->dt, L, l, s));
      PVW_HIP(hipMemsetAsync(d_xm, 0, xm_words(cnt, k, L, l) * 8, s));
      PVW_HIP(launch_mftile(d_srow, false, d_xm, cnt, k, L, l, s));
      // out / addend element (v = col, row = p) at p * k * P + col * P
      PVW_HIP(launch_gemm_digits(ga, gb, d_yd, d_sy, c->dt, k, L, l, k, P, 0, s));
    }
    PVW_HIP(hipEventRecord(w->events[2 * ci + 1], s));                         // this chunk's key buffer is free again
    if (ci + 1 < nchunks) {
      // issued after this chunk's launches so that the host-side staging of a pageable copy overlaps the GPU's work;
      // the buffer of chunk i+1 was last read by the prologue of chunk i-1
      if (ci >= 1) PVW_HIP(hipStreamWaitEvent(w->aux, w->events[2 * (ci - 1) + 1], 0));
      PVW_TRY(upload(ci + 1, w->aux));
      PVW_HIP(hipEventRecord(w->events[2 * ci + 2], w->aux));
    }
  }
  return PVW_OK;
}

// single exit of a key generation: both streams are drained whatever happened (an early return must not leave
// copies from the caller's sk / ek buffers or launches on pooled scratch in flight), the secret-bearing
// regions are cleared, and only then does the workspace go back to the pool
static int32_t keygen_finish(Workspace* w, int32_t rc) {
  if (w->aux && hipStreamSynchronize(w->aux) != hipSuccess && rc == PVW_OK) rc = fail(PVW_ERR_INTERNAL, "stream sync failed");
  if (hipStreamSynchronize(w->stream) != hipSuccess && rc == PVW_OK) rc = fail(PVW_ERR_INTERNAL, "stream sync failed");
  if ((ws_wipe_secrets(w, w->stream) != hipSuccess || hipStreamSynchronize(w->stream) != hipSuccess) && rc == PVW_OK)
    rc = fail(PVW_ERR_INTERNAL, "wipe failed");
  return rc;
}

int32_t pvw_keygen(pvw_ctx* c, uint32_t lo, uint32_t hi, const int64_t* sk, const int64_t* ek, const uint8_t seed[32]) {
  if (!c || !sk) return fail(PVW_ERR_INVALID_PARAMETERS, "NULL argument");
  if (!ek && !seed) return fail(PVW_ERR_INVALID_PARAMETERS, "either explicit key errors or a seed is required");
  PVW_TRY(check_party_range(c, lo, hi));
  PVW_TRY(ensure_device(c));
  c->xm_valid = false; c->pkB_valid = false; c->pk_wide = false;
  if (!c->crs_loaded) return fail(PVW_ERR_CRS, "CRS not loaded");
  if (c->rowsA() != c->k) return fail(PVW_ERR_KEY_GENERATION, "key generation needs the full CRS on this context");
  PVW_TRY(ensure_matrix(c, &c->dB, c->rowsB()));
  const u32 a = lo > c->party_lo ? lo : c->party_lo, b = hi < c->party_hi ? hi : c->party_hi;
  const u32 k = c->k, l = c->l, L = c->L;
  const size_t P = c->poly();
  Workspace* w;
  PVW_TRY(ws_acquire(c, &w));
  hipStream_t s = w->stream;
  // >= 8 parties: the matrix cores (gemm_digits, 16 parties per pass over A^T, everything around it batched
  // over super-groups of up to 128 parties); fewer: 4 per pass on the VALU
  const int gemm_min = (int)PVW_ENV_INT("PVW_GEMM_MIN_DEALERS", 8);   // tuning build only
  const bool use_gemm = gemm_min > 0 && (b - a) >= (u32)gemm_min;
  // default matrix-core form: parties as GEMM rows, the CRS columns digitised once (PVW_KEYGEN_SWAP=0: the earlier
  // form with the transposed CRS as the streamed operand and the secret keys digitised per super-group)
  const int swap_roles = (int)PVW_ENV_INT("PVW_KEYGEN_SWAP", 1);   // tuning build, per call: the tests walk both
  if (use_gemm && swap_roles && (b - a) >= 64) {
    int32_t rc2 = keygen_finish(w, keygen_gemm_swapped(c, w, a, b, lo, sk, ek, seed));
    ws_release(c, w);
    if (rc2 == PVW_OK && hi > c->num_keys) c->num_keys = hi;
    return rc2;
  }
  const u32 group = use_gemm ? 16 : 4;
  // super-group: parties whose sampling, NTTs, digit tiles and final tiling are single launches; bounded so
  // that the digit tiles stay below ~2 GiB
  u32 sg = group;
  if (use_gemm) {
    const size_t per16 = yd_bytes(16, k, L, l);
    size_t m = ((size_t)2 << 30) / (per16 ? per16 : 1);
    if (m < 1) m = 1;
    if (m > 8) m = 8;
    sg = 16 * (u32)m;
  }
  // scratch: A in API layout [k][k][P] | A^T in API layout | A^T tiled or MFMA-tiled | sk,ek of one chunk |
  //          rows of B for one super-group | gemm intermediate | (gemm) s-hat vectors, digit tiles, column sums
  const size_t b_api = ((size_t)k * k * P * 8 + 255) & ~(size_t)255;
  const size_t b_tt = ((use_gemm ? xm_words(k, k, L, l) : c->tiled_words(k)) * 8 + 255) & ~(size_t)255;
  u32 chunk = (b - a) < 1024 ? (b - a) : 1024;         // parties whose sk / ek are uploaded together
  if (chunk > sg) chunk -= chunk % sg;                  // whole super-groups per chunk
  const size_t b_small = ((size_t)2 * chunk * k * l * 8 + 255) & ~(size_t)255;
  const size_t b_row = ((size_t)sg * k * P * 8 + 255) & ~(size_t)255;
  const size_t b_tmp = use_gemm ? (((size_t)(sg / 16) * gemm_tmp_words(k, L, l) * 8 + 255) & ~(size_t)255) : 0;
  const size_t b_vh = use_gemm ? b_row : 0;
  const size_t b_yd = use_gemm ? ((yd_bytes(sg, k, L, l) + 255) & ~(size_t)255) : 0;
  const size_t b_sy = use_gemm ? ((sy_bytes(sg, L, l) + 255) & ~(size_t)255) : 0;
  int32_t rc = ws_scratch(w, 2 * b_api + b_tt + b_small + b_row + b_tmp + b_vh + b_yd + b_sy);
  if (rc == PVW_OK) {
    char* base = (char*)w->scratch;
    u64* d_api = (u64*)base;
    u64* d_apiT = (u64*)(base + b_api);
    u64* d_tt = (u64*)(base + 2 * b_api);
    i64* d_small = (i64*)(base + 2 * b_api + b_tt);
    u64* d_row = (u64*)(base + 2 * b_api + b_tt + b_small);
    u64* d_tmp = (u64*)(base + 2 * b_api + b_tt + b_small + b_row);
    char* gb0 = base + 2 * b_api + b_tt + b_small + b_row + b_tmp;
    u64* d_vh = (u64*)gb0;
    signed char* d_yd = (signed char*)(gb0 + b_vh);
    int* d_sy = (int*)(gb0 + b_vh + b_yd);
    // secret-bearing regions (cleared by keygen_finish): uploaded sk / ek coefficients; NTT(s) vectors, their digit
    // tiles and column sums (matrix-core form) or the s-hat vectors in w->rhat (VALU form).  d_row holds e only
    // until the product is added onto it, then rows of the public key.
    ws_mark_secret(w, d_small, b_small);
    if (use_gemm) {
      ws_mark_secret(w, d_tmp, b_tmp);                       // s A^T without the error
      ws_mark_secret(w, d_vh, b_vh + b_yd + b_sy);
    }
    else ws_mark_secret(w, w->rhat, w->rhat_bytes);
    // A -> API layout -> transpose polynomials (A^T[c][j] = A[j][c]) -> tiled / MFMA-tiled
    bool okk = launch_untile(c->dA, d_api, k, 0, k, L, l, false, c->dt, s) == hipSuccess;
    okk = okk && launch_transpose_polys(d_api, d_apiT, k, (u32)P, s) == hipSuccess;
    okk = okk && hipMemsetAsync(d_tt, 0, b_tt, s) == hipSuccess;
    if (use_gemm) okk = okk && launch_mftile(d_apiT, false, d_tt, k, k, L, l, s) == hipSuccess;
    else okk = okk && launch_tile(d_apiT, d_tt, k, 0, k, L, l, false, c->dt, s) == hipSuccess;
    if (!okk) rc = fail(PVW_ERR_INTERNAL, "CRS transpose failed");
    for (u32 p0 = a; rc == PVW_OK && p0 < b; p0 += sg) {
      const u32 nv = (b - p0) < sg ? (b - p0) : sg;
      const u32 in_chunk = (p0 - a) % chunk;                       // position inside the uploaded chunk
      if (in_chunk == 0) {
        const u32 cn = (b - p0) < chunk ? (b - p0) : chunk;
        const size_t words = (size_t)cn * k * l;
        if (hipMemcpyAsync(d_small, sk + (size_t)(p0 - lo) * k * l, words * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
            (ek && hipMemcpyAsync(d_small + (size_t)chunk * k * l, ek + (size_t)(p0 - lo) * k * l, words * 8, hipMemcpyHostToDevice, s) != hipSuccess)) {
          rc = fail(PVW_ERR_INTERNAL, "H2D failed");
          break;
        }
      }
      ProfScope ps(c, "keygen", s);
      bool ok2 = true;
      if (use_gemm) {
        // ONE prologue launch for the super-group: the (s, e) families of party p0 replicated over its nv parties.
        // s-hat_p (secret_key.rs:98-112) goes to the vector layout [party][limb][j][slot]; e_p (public_key.rs:128-132),
        // sampled or explicit, to NTT form in the row buffer [nv][k][P]
        {
          PrologueBatch pb{};
          if (seed) pb.key[0] = make_key(seed);
          PrologueJob& js = pb.job[0];
          PrologueJob& je = pb.job[1];
          js.sj.count = k; js.explicit_coeffs = d_small + (size_t)in_chunk * k * l;
          js.out = d_vh; js.stride_poly = l; js.stride_limb = (size_t)k * l;
          js.rep_coeffs = (size_t)k * l; js.rep_out = (size_t)k * P;
          je.sj.kind = SAMPLE_UNIFORM; je.sj.domain = DOM_EKEY; je.sj.index0 = p0 * k; je.sj.count = k; je.sj.bound = c->b1;
          je.rep_index0 = k;
          if (ek) { je.explicit_coeffs = d_small + (size_t)(chunk + in_chunk) * k * l; je.rep_coeffs = (size_t)k * l; }
          je.out = d_row; je.stride_poly = P; je.stride_limb = l; je.rep_out = (size_t)k * P;
          pb.njobs = 2;
          pb.reps = nv;
          ok2 = launch_prologue(pb, c->dt, L, l, s) == hipSuccess;
        }
        ok2 = ok2 && launch_vec_digits(d_vh, (size_t)k * P, d_yd, d_sy, nv, k, L, l, c->dt, s) == hipSuccess;
        // all batches of 16 parties in one launch (crs.rs:152-168): they share A^T through L2
        // ... and the finish pass writes b_p = s_p*A + e_p straight into the tiled B-hat (no re-tiling launch)
        GemmSection ga{d_tt, d_row, d_row, d_tmp, k, 0, 0}, gb{nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
        ga.tiled_out = c->dB;
        ga.tiled_row0 = p0 - c->party_lo;
        ok2 = ok2 && launch_gemm_digits(ga, gb, d_yd, d_sy, c->dt, k, L, l, nv, (size_t)k * P, 0, s) == hipSuccess;
      } else {
        u64* vh = w->rhat;
        PrologueBatch pb{};
        if (seed) pb.key[0] = make_key(seed);
        {
          PrologueJob& js = pb.job[0];
          PrologueJob& je = pb.job[1];
          js.sj.count = k; js.explicit_coeffs = d_small + (size_t)in_chunk * k * l;              // secret_key.rs:98-112
          js.out = vh; js.stride_poly = l; js.stride_limb = (size_t)k * l;
          js.rep_coeffs = (size_t)k * l; js.rep_out = (size_t)k * P;
          je.sj.kind = SAMPLE_UNIFORM; je.sj.domain = DOM_EKEY; je.sj.index0 = p0 * k; je.sj.count = k; je.sj.bound = c->b1;  // public_key.rs:128-132
          je.rep_index0 = k;
          if (ek) { je.explicit_coeffs = d_small + (size_t)(chunk + in_chunk) * k * l; je.rep_coeffs = (size_t)k * l; }
          je.out = d_row; je.stride_poly = P; je.stride_limb = l; je.rep_out = (size_t)k * P;
        }
        pb.njobs = 2;
        pb.reps = nv;
        ok2 = launch_prologue(pb, c->dt, L, l, s) == hipSuccess;
        MacSection sa{d_tt, d_row, d_row, k, 0}, sb{nullptr, nullptr, nullptr, 0, 0};
        MultiVec mv{vh, (size_t)k * P, (size_t)k * P, 0, nv};
        ok2 = ok2 && launch_mac_rows_multi(sa, sb, mv, c->dt, k, L, l, s) == hipSuccess;
      }
      // VALU path: d_row is [nv][k polys][P] = nv rows of B in API layout -> tile into B
      if (!use_gemm) ok2 = ok2 && launch_tile(d_row, c->dB, nv, p0 - c->party_lo, k, L, l, false, c->dt, s) == hipSuccess;
      if (!ok2) rc = fail(PVW_ERR_KEY_GENERATION, "keygen launch failed");
    }
  }
  rc = keygen_finish(w, rc);
  ws_release(c, w);
  if (rc == PVW_OK && hi > c->num_keys) c->num_keys = hi;
  return rc;
}

}  // extern "C"
