// Exercises include/barnett_smart.hpp (the C++ mirror of the reference's trait surface) against libmpshuffle.so.
// usage: mirror_smoke <case.bin> ; the case file is written by tests/test_gpu_parity.py::test_cpp_mirror:
//   u32 m, u32 n, params[(n+3)*64], pk[64], deck[N*128], rho[N*32], perm[N*4], seed[32], exp_deck[N*128], exp_proof[psz]
// Reads like the reference's test_shuffle [REF barnett-smart-card-protocol/src/discrete_log_cards/tests.rs:175-227].
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>

#include "barnett_smart.hpp"

using namespace barnett_smart;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream f(argv[1], std::ios::binary);
  std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  const uint8_t* p = buf.data();
  uint32_t m, n;
  memcpy(&m, p, 4); memcpy(&n, p + 4, 4); p += 8;
  const size_t N = (size_t)m * n, psz = mp_proof_size(m, n);
  Parameters pp; pp.m = m; pp.n = n; pp.raw.assign(p, p + 64 * (n + 3)); p += 64 * (n + 3);
  PublicKey pk; memcpy(pk.data(), p, 64); p += 64;
  std::vector<MaskedCard> deck(N); memcpy(deck[0].data(), p, N * 128); p += N * 128;
  std::vector<Scalar> rho(N); memcpy(rho[0].data(), p, N * 32); p += N * 32;
  Permutation perm; perm.mapping.resize(N); memcpy(perm.mapping.data(), p, N * 4); p += N * 4;
  std::array<uint8_t, 32> seed; memcpy(seed.data(), p, 32); p += 32;
  const uint8_t* exp_deck = p; p += N * 128;
  const uint8_t* exp_proof = p;

  DLCards cards(MP_CURVE_STARK, 0);
  auto res = cards.shuffle_and_remask(seed, pp, pk, deck, rho, perm);
  if (memcmp(res.first[0].data(), exp_deck, N * 128) != 0) { printf("FAIL: shuffled deck differs\n"); return 1; }
  if (res.second.size() != psz || memcmp(res.second.data(), exp_proof, psz) != 0) { printf("FAIL: proof differs\n"); return 1; }
  cards.verify_shuffle(pp, pk, deck, res.first, res.second);          // Ok(())
  {                                                                     // serialise -> deserialise round trip of the proof
    const std::vector<uint8_t> ser = cards.serialize(pp, res.second);
    if (ser.size() != cards.serialized_size(pp) || cards.deserialize_proof(pp, ser) != res.second) {
      std::fprintf(stderr, "serialisation round trip failed\n");
      return 3;
    }
  }
  std::vector<MaskedCard> wrong(res.first.rbegin(), res.first.rend());  // some other deck
  try {
    cards.verify_shuffle(pp, pk, deck, wrong, res.second);
    printf("FAIL: wrong deck accepted\n");
    return 1;
  } catch (const CryptoError& e) {
    if (e.check != "Hadamard Product (5.1)") { printf("FAIL: wrong check name %s\n", e.check.c_str()); return 1; }
  }
  Permutation bad = perm; bad.mapping[0] = bad.mapping[1];
  try {
    cards.shuffle_and_remask(seed, pp, pk, deck, rho, bad);
    printf("FAIL: bad permutation accepted\n");
    return 1;
  } catch (const CardProtocolError&) {
  }
  printf("mirror_smoke ok\n");
  return 0;
}
