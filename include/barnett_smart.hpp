// barnett_smart.hpp -- header-only C++ mirror of the reference's trait surface for the hot path, over the C ABI of
// libmpshuffle.so.  Same names, argument meaning and error behaviour as
//   trait BarnettSmartProtocol { fn setup; fn shuffle_and_remask; fn verify_shuffle; }
//   [REF barnett-smart-card-protocol/src/lib.rs:74-78, 181-197] as implemented by DLCards<C>
//   [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:105-121, 380-443].
// Result<T, E> becomes: return T, throw E.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mpshuffle.h"

namespace barnett_smart {

// proof_essentials::error::CryptoError::ProofVerificationError(String)   [REF src/discrete_log_cards/tests.rs:223-225]
struct CryptoError : std::runtime_error {
  std::string check;
  explicit CryptoError(const std::string& name) : std::runtime_error("ProofVerificationError(" + name + ")"), check(name) {}
};
// CardProtocolError::{ProofVerificationError(CryptoError), IoError(String)}   [REF src/error.rs:6-12]
struct CardProtocolError : std::runtime_error {
  explicit CardProtocolError(const std::string& io) : std::runtime_error("IoError: " + io) {}
};

typedef std::array<uint8_t, 32> Scalar;        // C::ScalarField, little-endian canonical
typedef std::vector<uint8_t> ZKProofShuffle;   // shuffle::proof::Proof

struct Permutation {                            // utils::permutation::Permutation
  std::vector<uint32_t> mapping;                // permute_array(v)[i] = v[mapping[i]]
};

struct Parameters {                             // discrete_log_cards::Parameters [REF mod.rs:37-61]
  uint32_t m = 0, n = 0;
  std::vector<uint8_t> raw;                     // G | ck_0..ck_{n-1} | H | gen
};

// PB = wire bytes of an affine point on the curve: 64 for the 256-bit curves, 96 for BLS12-377 (mp_point_size)
template <size_t PB>
class DLCardsT {
 public:
  typedef std::array<uint8_t, PB> PublicKey;        // el_gamal::PublicKey (affine point)
  typedef std::array<uint8_t, 2 * PB> MaskedCard;   // el_gamal::Ciphertext(pub Affine, pub Affine)

  explicit DLCardsT(int curve_id = MP_CURVE_STARK, int device = 0) : curve_(curve_id) {
    if (mp_point_size(curve_id) != PB) throw CardProtocolError("curve / point size mismatch");
    if (mp_ctx_create(curve_id, device, &ctx_) != MP_OK) throw CardProtocolError(mp_last_error());
  }
  ~DLCardsT() {
    if (table_) mp_table_destroy(table_);
    mp_ctx_destroy(ctx_);
  }
  DLCardsT(const DLCardsT&) = delete;
  DLCardsT& operator=(const DLCardsT&) = delete;

  // fn setup<R: Rng>(rng, m, n) -> Result<Parameters, CardProtocolError>
  Parameters setup(const std::array<uint8_t, 32>& rng_seed, uint32_t m, uint32_t n) {
    Parameters pp;
    pp.m = m;
    pp.n = n;
    pp.raw.resize(mp_params_size_curve(curve_, n));
    if (mp_setup(ctx_, m, n, rng_seed.data(), pp.raw.data()) != MP_OK) throw CardProtocolError(mp_last_error());
    return pp;
  }

  // fn shuffle_and_remask<R: Rng>(rng, pp, shared_key, deck, masking_factors, permutation)
  //     -> Result<(Vec<MaskedCard>, ZKProofShuffle), CardProtocolError>
  std::pair<std::vector<MaskedCard>, ZKProofShuffle> shuffle_and_remask(const std::array<uint8_t, 32>& rng_seed, const Parameters& pp,
                                                                        const PublicKey& shared_key, const std::vector<MaskedCard>& deck,
                                                                        const std::vector<Scalar>& masking_factors,
                                                                        const Permutation& permutation) {
    const size_t N = (size_t)pp.m * pp.n;
    if (deck.size() != N || masking_factors.size() != N || permutation.mapping.size() != N)
      throw CardProtocolError("deck, masking factors and permutation must have m*n entries");
    bind(pp, shared_key);
    std::vector<MaskedCard> out(N);
    ZKProofShuffle proof(mp_proof_size_curve(curve_, pp.m, pp.n));
    int rc = mp_shuffle_and_remask(table_, deck[0].data(), masking_factors[0].data(), permutation.mapping.data(), rng_seed.data(),
                                   out[0].data(), proof.data());
    if (rc != MP_OK) throw CardProtocolError(mp_last_error());
    return {std::move(out), std::move(proof)};
  }

  // fn verify_shuffle(pp, shared_key, original_deck, shuffled_deck, proof) -> Result<(), CryptoError>
  void verify_shuffle(const Parameters& pp, const PublicKey& shared_key, const std::vector<MaskedCard>& original_deck,
                      const std::vector<MaskedCard>& shuffled_deck, const ZKProofShuffle& proof) {
    const size_t N = (size_t)pp.m * pp.n;
    if (original_deck.size() != N || shuffled_deck.size() != N) throw CardProtocolError("decks must have m*n entries");
    bind(pp, shared_key);
    int rc = mp_verify_shuffle(table_, original_deck[0].data(), shuffled_deck[0].data(), proof.data(), proof.size());
    if (rc > 0) throw CryptoError(mp_check_name(rc));
    if (rc < 0) throw CardProtocolError(mp_last_error());
  }

  // CanonicalSerialize / CanonicalDeserialize of ZKProofShuffle and `proof.serialized_size()`
  // [REF src/lib.rs:71; examples/parameter_selection.rs:95]
  size_t serialized_size(const Parameters& pp) const { return mp_serialized_proof_size(curve_, pp.m, pp.n); }
  std::vector<uint8_t> serialize(const Parameters& pp, const ZKProofShuffle& proof) const {
    std::vector<uint8_t> out(serialized_size(pp));
    if (proof.size() != mp_proof_size_curve(curve_, pp.m, pp.n) || mp_proof_serialize(curve_, pp.m, pp.n, proof.data(), out.data()) != MP_OK)
      throw CardProtocolError(mp_last_error());
    return out;
  }
  ZKProofShuffle deserialize_proof(const Parameters& pp, const std::vector<uint8_t>& bytes) const {
    ZKProofShuffle proof(mp_proof_size_curve(curve_, pp.m, pp.n));
    if (mp_proof_deserialize(curve_, pp.m, pp.n, bytes.data(), bytes.size(), proof.data()) != MP_OK) throw CardProtocolError(mp_last_error());
    return proof;
  }

  mp_table* table() const { return table_; }   // for the batched / device-resident entry points of mpshuffle.h

 private:
  void bind(const Parameters& pp, const PublicKey& pk) {
    if (table_ && pp.raw == bound_params_ && pk == bound_pk_ && pp.m == bound_m_) return;
    if (table_) mp_table_destroy(table_);
    table_ = nullptr;
    if (mp_table_create(ctx_, pp.m, pp.n, pp.raw.data(), pk.data(), &table_) != MP_OK) throw CardProtocolError(mp_last_error());
    bound_params_ = pp.raw;
    bound_pk_ = pk;
    bound_m_ = pp.m;
  }
  int curve_;
  mp_ctx* ctx_ = nullptr;
  mp_table* table_ = nullptr;
  std::vector<uint8_t> bound_params_;
  PublicKey bound_pk_{};
  uint32_t bound_m_ = 0;
};
typedef DLCardsT<64> DLCards;              // DLCards<starknet_curve / bn254 / secp256k1>
typedef DLCardsT<96> DLCardsBls12_377;     // DLCards<ark_bls12_377::G1Projective> [REF examples/parameter_selection.rs:25-29]
typedef DLCards::PublicKey PublicKey;
typedef DLCards::MaskedCard MaskedCard;

}  // namespace barnett_smart
