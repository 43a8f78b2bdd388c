// pvw.hpp -- header-only C++ mirror of the reference's `pvw::{params, crs, keys, crypto}`
// interface over the C ABI (include/pvw_hip.h).
//
// The reference is a Rust crate; no Rust toolchain exists in the build image, so the host
// side above the C ABI is written in C++ with the same names, argument meaning and error
// behaviour (citations are file:line under the reference checkout).  Everything heavy runs
// in libpvw_hip.so on the GPU; this header only owns handles and flat buffers.
//   polynomial  = std::vector<uint64_t> of L*l residues, limb-major (parameters.rs:433-458)
//   randomness  = a 32-byte seed (the reference uses thread_rng(), encryption.rs:138,164,180)
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pvw_hip.h"

namespace pvw_host {

using Seed = std::array<uint8_t, 32>;

// PvwError (src/errors.rs:13-70)
class PvwError : public std::runtime_error {
 public:
  int32_t code;
  PvwError(int32_t c, const std::string& m) : std::runtime_error(variant_name(c) + ": " + m), code(c) {}
  static std::string variant_name(int32_t c) {
    static const char* names[] = {"Ok", "InvalidParameters", "SamplingError", "EncryptionError", "DecryptionError",
                                  "KeyGenerationError", "CrsError", "SerializationError", "DeserializationError",
                                  "EncodingError", "DecodingError", "ValidationError", "ContextError",
                                  "PolynomialError", "MatrixError", "DimensionMismatch", "IndexOutOfBounds",
                                  "InsufficientData", "InvalidFormat", "InternalError"};
    return (c >= 0 && c <= 19) ? names[c] : "Unknown";
  }
  std::string variant() const { return variant_name(code); }
};
inline void check(int32_t rc) {
  if (rc == PVW_OK) return;
  char buf[512];
  pvw_last_error(buf, sizeof buf);
  throw PvwError(rc, buf);
}

// PvwParameters (src/params/parameters.rs:19-40) + the device context behind it
class PvwParameters {
 public:
  uint32_t n, k, l, t;
  std::vector<uint64_t> moduli_;
  float secret_variance;
  uint64_t error_bound_1, error_bound_2;
  pvw_ctx* ctx = nullptr;

  ~PvwParameters() { pvw_ctx_destroy(ctx); }
  PvwParameters(const PvwParameters&) = delete;
  PvwParameters& operator=(const PvwParameters&) = delete;
  PvwParameters() = default;

  size_t L() const { return moduli_.size(); }
  size_t poly_words() const { return L() * l; }
  const std::vector<uint64_t>& moduli() const { return moduli_; }                    // :389
  bool verify_correctness_condition() const {                                        // :510-551
    int32_t ok = 0;
    check(pvw_ctx_verify_correctness_condition(ctx, &ok));
    return ok != 0;
  }
  std::vector<uint64_t> delta() const { return big(pvw_ctx_delta); }                  // :370 (LE 64-bit words)
  std::vector<uint64_t> q_total() const { return big(pvw_ctx_q_total); }              // :380
  std::vector<uint64_t> gadget_polynomial(uint32_t repr = PVW_REPR_POWER) const {     // :288-308
    std::vector<uint64_t> g(poly_words());
    check(pvw_ctx_gadget(ctx, g.data(), repr));
    return g;
  }
  std::vector<uint64_t> encode_scalar(int64_t scalar, uint32_t repr = PVW_REPR_POWER) const {   // :346-367
    std::vector<uint64_t> g(poly_words());
    check(pvw_encode_scalar(ctx, scalar, g.data(), repr));
    return g;
  }
  static std::pair<uint32_t, uint32_t> suggest_error_bounds(uint32_t n, uint32_t k, uint32_t l,
                                                            const std::vector<uint64_t>& moduli, float variance) {  // :554-603
    uint32_t b1 = 0, b2 = 0;
    check(pvw_suggest_error_bounds(n, k, l, moduli.data(), (uint32_t)moduli.size(), variance, &b1, &b2));
    return {b1, b2};
  }

 private:
  template <class F>
  std::vector<uint64_t> big(F fn) const {
    size_t nw = 0;
    check(fn(ctx, nullptr, 0, &nw));
    std::vector<uint64_t> w(nw);
    check(fn(ctx, w.data(), w.size(), &nw));
    return w;
  }
};

// PvwParametersBuilder (parameters.rs:44-201)
class PvwParametersBuilder {
  std::optional<uint32_t> n_, k_, l_;
  std::optional<std::vector<uint64_t>> moduli_;
  std::optional<float> variance_;
  std::optional<uint64_t> b1_, b2_;
  int32_t device_ = -1;
  uint32_t shard_[4] = {0, 0, 0, 0};

 public:
  PvwParametersBuilder& set_parties(uint32_t n) { n_ = n; return *this; }
  PvwParametersBuilder& set_dimension(uint32_t k) { k_ = k; return *this; }
  PvwParametersBuilder& set_l(uint32_t l) { l_ = l; return *this; }
  PvwParametersBuilder& set_moduli(const std::vector<uint64_t>& m) { moduli_ = m; return *this; }
  PvwParametersBuilder& set_secret_variance(float v) { variance_ = v; return *this; }
  PvwParametersBuilder& set_error_bound_1(uint64_t b) { b1_ = b; return *this; }
  PvwParametersBuilder& set_error_bound_2(uint64_t b) { b2_ = b; return *this; }
  PvwParametersBuilder& set_error_bounds_u32(uint32_t a, uint32_t b) { b1_ = a; b2_ = b; return *this; }
  PvwParametersBuilder& set_device(int32_t d) { device_ = d; return *this; }
  PvwParametersBuilder& set_shard(uint32_t plo, uint32_t phi, uint32_t clo, uint32_t chi) {
    shard_[0] = plo; shard_[1] = phi; shard_[2] = clo; shard_[3] = chi;
    return *this;
  }
  std::shared_ptr<PvwParameters> build_arc() const {                                  // :117-201
    if (!n_) throw PvwError(1, "n not set");
    if (!k_) throw PvwError(1, "k not set");
    if (!l_) throw PvwError(1, "l not set");
    if (!moduli_) throw PvwError(1, "moduli not set");
    auto p = std::make_shared<PvwParameters>();
    p->n = *n_; p->k = *k_; p->l = *l_; p->moduli_ = *moduli_;
    p->secret_variance = variance_.value_or(0.5f);                                    // :166
    p->error_bound_1 = b1_.value_or(100);                                             // :167
    p->error_bound_2 = b2_.value_or(200);                                             // :168
    p->t = *n_ ? (*n_ - 1) / 2 : 0;                                                   // :169
    pvw_params_t c{};
    c.n = p->n; c.k = p->k; c.l = p->l; c.num_moduli = (uint32_t)p->moduli_.size(); c.moduli = p->moduli_.data();
    c.secret_variance = p->secret_variance; c.error_bound_1 = p->error_bound_1; c.error_bound_2 = p->error_bound_2;
    c.device = device_; c.party_lo = shard_[0]; c.party_hi = shard_[1]; c.c1_lo = shard_[2]; c.c1_hi = shard_[3];
    check(pvw_ctx_create(&c, &p->ctx));
    return p;
  }
};

// PvwCrs (src/params/crs.rs:12-17): the k x k matrix is resident on the device of `params`
class PvwCrs {
 public:
  std::shared_ptr<PvwParameters> params;
  static PvwCrs new_deterministic(const std::shared_ptr<PvwParameters>& p, const Seed& seed) {   // crs.rs:45-67
    check(pvw_crs_generate(p->ctx, seed.data()));
    return PvwCrs{p};
  }
  // PvwCrs::new (crs.rs:24-39): a fresh random CRS; `rng` is any callable returning random 32-bit words
  // (default: std::random_device, the OS entropy source -- the reference takes a CryptoRng)
  template <class Rng>
  static PvwCrs create(const std::shared_ptr<PvwParameters>& p, Rng& rng) {
    Seed seed;
    for (size_t i = 0; i < 32; i += 4) {
      const uint32_t w = (uint32_t)rng();
      for (size_t b = 0; b < 4; ++b) seed[i + b] = (uint8_t)(w >> (8 * b));
    }
    return new_deterministic(p, seed);
  }
  static PvwCrs create(const std::shared_ptr<PvwParameters>& p) {
    std::random_device rd;
    return create(p, rd);
  }
  // PvwCrs::new_from_tag (crs.rs:74-90): the seed is DefaultHasher(tag + "CRS"), 8 LE bytes repeated four times
  static Seed seed_from_tag(const std::string& tag) {
    Seed seed;
    check(pvw_crs_seed_from_tag(tag.c_str(), seed.data()));
    return seed;
  }
  static PvwCrs new_from_tag(const std::shared_ptr<PvwParameters>& p, const std::string& tag) {
    return new_deterministic(p, seed_from_tag(tag));
  }
  static PvwCrs from_polynomials(const std::shared_ptr<PvwParameters>& p, const std::vector<uint64_t>& a,
                                 uint32_t repr = PVW_REPR_POWER) {
    if (a.size() != (size_t)p->k * p->k * p->poly_words()) throw PvwError(15, "CRS size mismatch");
    check(pvw_load_crs(p->ctx, a.data(), repr));
    return PvwCrs{p};
  }
  std::pair<uint32_t, uint32_t> dimensions() const { return {params->k, params->k}; }
};

// SecretKey (src/keys/secret_key.rs:14-18): k x l CBD coefficients
class SecretKey {
 public:
  std::shared_ptr<PvwParameters> params;
  std::vector<int64_t> secret_coeffs;   // [k][l]
  SecretKey(std::shared_ptr<PvwParameters> p, std::vector<int64_t> c) : params(std::move(p)), secret_coeffs(std::move(c)) {}
  SecretKey(const SecretKey&) = default;
  SecretKey(SecretKey&&) = default;
  // assignment wipes what the key held before the vector lets go of it (Zeroize, secret_key.rs:20-30)
  SecretKey& operator=(const SecretKey& o) {
    if (this != &o) { zeroize(); params = o.params; secret_coeffs = o.secret_coeffs; }
    return *this;
  }
  SecretKey& operator=(SecretKey&& o) noexcept {
    if (this != &o) { zeroize(); params = std::move(o.params); secret_coeffs = std::move(o.secret_coeffs); }
    return *this;
  }
  ~SecretKey() { zeroize(); }                                   // ZeroizeOnDrop (secret_key.rs:20-30)
  static SecretKey random(const std::shared_ptr<PvwParameters>& p, const Seed& seed, uint32_t party_index) {   // :45-63
    SecretKey s{p, std::vector<int64_t>((size_t)p->k * p->l)};
    check(pvw_sample_secret_keys(p->ctx, seed.data(), party_index, 1, s.secret_coeffs.data()));
    return s;
  }
  // Zeroize: the coefficients are overwritten through a volatile pointer (not elided as a dead store); the device
  // side clears its own copies before every key-bearing call returns (pvw_selftest_secret_residue)
  void zeroize() {
    volatile int64_t* p = secret_coeffs.data();
    for (size_t i = 0; i < secret_coeffs.size(); ++i) p[i] = 0;
  }
  size_t len() const { return params->k; }
};

// A SecretKey kept on the device in the form the inner products of decrypt read (pvw_sk_load: NTT(sk[j]),
// secret_key.rs:98-112) for pvw_decrypt_batch_device_sk; cleared when the handle goes (pvw_sk_free), as the reference's
// SecretKey is ZeroizeOnDrop (secret_key.rs:20-30).  Keeps its parameters alive.
class DeviceSecretKey {
 public:
  explicit DeviceSecretKey(const SecretKey& sk) : params_(sk.params) { check(pvw_sk_load(params_->ctx, sk.secret_coeffs.data(), &key_)); }
  DeviceSecretKey(const DeviceSecretKey&) = delete;
  DeviceSecretKey& operator=(const DeviceSecretKey&) = delete;
  DeviceSecretKey(DeviceSecretKey&& o) noexcept : params_(std::move(o.params_)), key_(o.key_) { o.key_ = nullptr; }
  ~DeviceSecretKey() { if (key_) pvw_sk_free(key_); }
  const pvw_sk* raw() const { return key_; }

 private:
  std::shared_ptr<PvwParameters> params_;
  pvw_sk* key_ = nullptr;
};

// Party (src/keys/public_key.rs:17-22)
class Party {
 public:
  uint32_t index;
  SecretKey secret_key;
  static Party create(uint32_t index, const std::shared_ptr<PvwParameters>& p, const Seed& seed) {   // Party::new :62-79
    if (index >= p->n)
      throw PvwError(1, "Party index " + std::to_string(index) + " exceeds maximum " + std::to_string(p->n - 1));
    return Party{index, SecretKey::random(p, seed, index)};
  }
};

// GlobalPublicKey (public_key.rs:43-54): the n x k matrix B is resident on the device
class GlobalPublicKey {
 public:
  PvwCrs crs;
  std::shared_ptr<PvwParameters> params;
  explicit GlobalPublicKey(const PvwCrs& c) : crs(c), params(c.params) {}
  void add_public_key(uint32_t index, const std::vector<uint64_t>& key_polynomials, uint32_t repr = PVW_REPR_POWER) {   // :214-250
    if (key_polynomials.size() != (size_t)params->k * params->poly_words()) throw PvwError(1, "Public key dimension mismatch");
    check(pvw_load_pk(params->ctx, index, index + 1, key_polynomials.data(), repr));
  }
  void generate_and_add_party(const Party& party, const Seed& seed) {                                  // :256-263
    check(pvw_keygen(params->ctx, party.index, party.index + 1, party.secret_key.secret_coeffs.data(), nullptr, seed.data()));
  }
  void generate_all_party_keys(const std::vector<Party>& parties, const Seed& seed) {                  // :376-401
    if (parties.size() > params->n) throw PvwError(1, "Too many parties");
    // every run of consecutive party indices is ONE batched device call (the reference generates in parallel and
    // adds in order, :387-399)
    for (size_t i = 0; i < parties.size();) {
      size_t j = i + 1;
      while (j < parties.size() && parties[j].index == parties[j - 1].index + 1) ++j;
      std::vector<int64_t> sk;
      for (size_t x = i; x < j; ++x) sk.insert(sk.end(), parties[x].secret_key.secret_coeffs.begin(), parties[x].secret_key.secret_coeffs.end());
      const int32_t rc = pvw_keygen(params->ctx, parties[i].index, parties[i].index + (uint32_t)(j - i), sk.data(), nullptr, seed.data());
      volatile int64_t* w = sk.data();
      for (size_t x = 0; x < sk.size(); ++x) w[x] = 0;
      check(rc);
      i = j;
    }
  }
  uint32_t num_public_keys() const { uint32_t v = 0; check(pvw_num_public_keys(params->ctx, &v)); return v; }   // :344
  bool is_full() const { int32_t v = 0; check(pvw_is_full(params->ctx, &v)); return v != 0; }                   // :349
  std::pair<uint32_t, uint32_t> dimensions() const { return {params->n, params->k}; }
};

// PvwCiphertext (src/crypto/encryption.rs:15-24)
class PvwCiphertext {
 public:
  std::vector<uint64_t> c1, c2;   // [k][L][l], [n][L][l]
  std::shared_ptr<PvwParameters> params;
  uint32_t repr;
  size_t len() const { return c2.size() / params->poly_words(); }
  void validate() const {                                                                            // :41-76
    if (c1.size() != (size_t)params->k * params->poly_words()) throw PvwError(1, "c1 has the wrong number of components");
    if (c2.size() != (size_t)params->n * params->poly_words()) throw PvwError(1, "c2 has the wrong number of components");
  }
};

// encrypt (encryption.rs:105-214)
inline PvwCiphertext encrypt(const std::vector<uint64_t>& scalars, const GlobalPublicKey& gpk, const Seed& seed,
                             uint32_t repr = PVW_REPR_NTT) {
  const auto& p = gpk.params;
  PvwCiphertext ct{std::vector<uint64_t>((size_t)p->k * p->poly_words()), std::vector<uint64_t>((size_t)p->n * p->poly_words()), p, repr};
  pvw_randomness_t rnd{};
  rnd.mode = PVW_RND_SEED;
  for (int i = 0; i < 32; ++i) rnd.seed[i] = seed[i];
  check(pvw_encrypt(p->ctx, scalars.data(), scalars.size(), &rnd, ct.c1.data(), ct.c2.data(), repr));
  ct.validate();                                                                                     // :204-211
  return ct;
}
inline Seed dealer_seed(Seed s, uint32_t dealer) {
  for (int i = 0; i < 4; ++i) s[28 + i] ^= (uint8_t)(dealer >> (8 * i));
  return s;
}
// encrypt_party_shares (encryption.rs:221-245)
inline PvwCiphertext encrypt_party_shares(const std::vector<uint64_t>& shares, uint32_t party_index,
                                          const GlobalPublicKey& gpk, const Seed& seed) {
  if (party_index >= gpk.params->n) throw PvwError(1, "Party index exceeds maximum");
  if (shares.size() != gpk.params->n) throw PvwError(1, "Party must provide n shares");
  return encrypt(shares, gpk, seed);
}
// encrypt_all_party_shares (encryption.rs:253-286): ONE batched call for all dealers (pvw_encrypt_multi: the
// dealers share passes over the public key; from 8 dealers up on the matrix cores), dealer d seeded with
// dealer_seed(seed, d) -- the same ciphertexts as n separate encrypt_party_shares calls
inline std::vector<PvwCiphertext> encrypt_all_party_shares(const std::vector<std::vector<uint64_t>>& all_shares,
                                                           const GlobalPublicKey& gpk, const Seed& seed,
                                                           uint32_t repr = PVW_REPR_NTT) {
  const auto& p = gpk.params;
  if (all_shares.size() != p->n) throw PvwError(1, "Must provide shares for all parties");
  const size_t D = all_shares.size(), n = p->n, P = p->poly_words();
  std::vector<uint64_t> scalars(D * n), c1(D * p->k * P), c2(D * n * P);
  std::vector<uint8_t> seeds(D * 32);
  for (size_t d = 0; d < D; ++d) {
    if (all_shares[d].size() != n) throw PvwError(1, "Party must provide n shares");
    std::copy(all_shares[d].begin(), all_shares[d].end(), scalars.begin() + d * n);
    const Seed sd = dealer_seed(seed, (uint32_t)d);
    std::copy(sd.begin(), sd.end(), seeds.begin() + d * 32);
  }
  check(pvw_encrypt_multi(p->ctx, scalars.data(), D, n, seeds.data(), c1.data(), c2.data(), repr));
  std::vector<PvwCiphertext> out;
  for (size_t d = 0; d < D; ++d) {
    PvwCiphertext ct{std::vector<uint64_t>(c1.begin() + d * p->k * P, c1.begin() + (d + 1) * p->k * P),
                     std::vector<uint64_t>(c2.begin() + d * n * P, c2.begin() + (d + 1) * n * P), p, repr};
    ct.validate();
    out.push_back(std::move(ct));
  }
  return out;
}
// encrypt_broadcast (encryption.rs:292-296)
inline PvwCiphertext encrypt_broadcast(uint64_t scalar, const GlobalPublicKey& gpk, const Seed& seed) {
  return encrypt(std::vector<uint64_t>(gpk.params->n, scalar), gpk, seed);
}
// decrypt_party_shares (decryption.rs:281-325): one batched device pass over all dealers
inline std::vector<uint64_t> decrypt_party_shares(const std::vector<PvwCiphertext>& cts, const SecretKey& sk, uint32_t party_index) {
  if (cts.empty()) throw PvwError(1, "No ciphertexts provided");
  const auto& p = cts[0].params;
  if (cts.size() != p->n) throw PvwError(1, "Expected n ciphertexts");
  if (party_index >= p->n) throw PvwError(1, "Party index exceeds maximum");
  const size_t P = p->poly_words();
  std::vector<uint64_t> c1s, c2col, out(cts.size());
  for (const auto& ct : cts) {
    ct.validate();
    c1s.insert(c1s.end(), ct.c1.begin(), ct.c1.end());
    c2col.insert(c2col.end(), ct.c2.begin() + (size_t)party_index * P, ct.c2.begin() + (size_t)(party_index + 1) * P);
  }
  check(pvw_decrypt_batch(p->ctx, sk.secret_coeffs.data(), c1s.data(), c2col.data(), cts.size(), cts[0].repr, out.data(), nullptr));
  return out;
}
// decrypt_party_value (decryption.rs:249-278)
inline uint64_t decrypt_party_value(const PvwCiphertext& ct, const SecretKey& sk, uint32_t party_index) {
  const auto& p = ct.params;
  const size_t P = p->poly_words();
  uint64_t out = 0;
  check(pvw_decrypt_batch(p->ctx, sk.secret_coeffs.data(), ct.c1.data(), ct.c2.data() + (size_t)party_index * P, 1, ct.repr, &out, nullptr));
  return out;
}

}  // namespace pvw_host
