// Reads like the reference's tests/crypto.rs::test_decryption_l16 (:237-305), through the C++ host
// mirror (pvw_rs_amd/host/pvw.hpp) over the C ABI.  Built and run by tests/test_cpp_host.py on the GPU box.
#include <cstdio>

#include "../../pvw_rs_amd/host/pvw.hpp"

using namespace pvw_host;

int main() {
  try {
    const std::vector<uint64_t> moduli = {0xffffee001ULL, 0xffffc4001ULL, 0x1ffffe0001ULL};
    const uint32_t num_parties = 10;
    auto [bound1, bound2] = PvwParameters::suggest_error_bounds(num_parties, 4, 16, moduli, 0.5f);
    auto params = PvwParametersBuilder().set_parties(num_parties).set_dimension(4).set_l(16).set_moduli(moduli)
                      .set_secret_variance(0.5f).set_error_bounds_u32(bound1, bound2).build_arc();
    if (!params->verify_correctness_condition()) { printf("gate failed\n"); return 1; }
    Seed seed;
    seed.fill(0x2A);
    PvwCrs crs = PvwCrs::new_deterministic(params, seed);
    GlobalPublicKey global_pk(crs);
    std::vector<Party> parties;
    for (uint32_t i = 0; i < num_parties; ++i) {
      parties.push_back(Party::create(i, params, seed));
      global_pk.generate_and_add_party(parties.back(), seed);
    }
    if (!global_pk.is_full()) { printf("not full\n"); return 1; }
    std::vector<std::vector<uint64_t>> all(num_parties);
    for (uint32_t d = 0; d < num_parties; ++d)
      for (uint32_t j = 1; j <= num_parties; ++j) all[d].push_back(d * 100 + j);
    auto cts = encrypt_all_party_shares(all, global_pk, seed);
    // the batched call gives the ciphertexts of n separate per-dealer calls (encryption.rs:277-283)
    for (uint32_t d = 0; d < num_parties; ++d) {
      auto one = encrypt_party_shares(all[d], d, global_pk, dealer_seed(seed, d));
      if (one.c1 != cts[d].c1 || one.c2 != cts[d].c2) { printf("batched/separate mismatch at dealer %u\n", d); return 1; }
    }
    uint32_t correct = 0, total = 0;
    for (uint32_t i = 0; i < num_parties; ++i) {
      auto shares = decrypt_party_shares(cts, parties[i].secret_key, i);
      for (uint32_t d = 0; d < num_parties; ++d) { correct += shares[d] == all[d][i]; ++total; }
      if (decrypt_party_value(cts[0], parties[i].secret_key, i) != shares[0]) { printf("single/batch mismatch\n"); return 1; }
    }
    printf("success %u/%u\n", correct, total);
    // error behaviour: wrong number of scalars -> InvalidParameters (tests/crypto.rs:181-207)
    try {
      encrypt({1, 2}, global_pk, seed);
      printf("missing error\n");
      return 1;
    } catch (const PvwError& e) {
      if (e.variant() != "InvalidParameters") { printf("wrong variant %s\n", e.what()); return 1; }
    }
    try {
      Party::create(num_parties, params, seed);
      return 1;
    } catch (const PvwError&) {}
    if (correct * 100 < total * 95) return 1;
    // PvwCrs::new_from_tag (crs.rs:74-90) + generate_all_party_keys as ONE batched call (public_key.rs:376-401):
    // same tag, same CRS; the batched keys decrypt what was encrypted to them
    {
      auto params2 = PvwParametersBuilder().set_parties(num_parties).set_dimension(4).set_l(16).set_moduli(moduli)
                         .set_secret_variance(0.5f).set_error_bounds_u32(bound1, bound2).build_arc();
      if (PvwCrs::seed_from_tag("host-roundtrip") != PvwCrs::seed_from_tag("host-roundtrip") ||
          PvwCrs::seed_from_tag("host-roundtrip") == PvwCrs::seed_from_tag("another tag")) { printf("tag seed\n"); return 1; }
      GlobalPublicKey gpk2(PvwCrs::new_from_tag(params2, "host-roundtrip"));
      std::vector<Party> parties2;
      for (uint32_t i = 0; i < num_parties; ++i) parties2.push_back(Party::create(i, params2, seed));
      gpk2.generate_all_party_keys(parties2, seed);
      if (!gpk2.is_full()) { printf("batched keygen: not full\n"); return 1; }
      auto ct = encrypt_party_shares(all[3], 3, gpk2, seed);
      uint32_t good = 0;
      for (uint32_t i = 0; i < num_parties; ++i) good += decrypt_party_value(ct, parties2[i].secret_key, i) == all[3][i];
      if (good * 100 < num_parties * 95) { printf("batched keygen round trip %u/%u\n", good, num_parties); return 1; }
      SecretKey gone = parties2[0].secret_key;
      gone.zeroize();
      for (int64_t v : gone.secret_coeffs) if (v != 0) { printf("zeroize\n"); return 1; }
    }
    printf("CPP_HOST_OK\n");
    return 0;
  } catch (const std::exception& e) {
    printf("exception: %s\n", e.what());
    return 2;
  }
}
