// Batched sigma protocols: Schnorr identification (1 base) and Chaum-Pedersen discrete-log equality
// (2 bases) -- the proofs behind DLCards::{prove,verify}_key_ownership, mask, remask, compute_reveal_token and their
// verifiers [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:132-357].  One lane = one proof; the group
// work (commitments r*g_i; checks z*g_i - c*a_i - A_i == O) goes through the same table / Straus kernels as the
// shuffle argument.  "sigma transcript v1" (oracle/py/mp_oracle.py): absorb(g.., a.., A..) -> c ; z = r + c*x.
#pragma once
#include "kernels_proto.hpp"

namespace mp {

struct SigmaLay {
  uint32_t nb;
  // S slots
  uint32_t x, r, c, z, negc, minus_one;
  // P slots: bases [g, g+nb), publics [a, a+nb), commitments [A, A+nb); J check slots [chk, chk+nb)
  uint32_t g, a, A, chk;
};
MP_HD SigmaLay make_sigma_lay(uint32_t nb) {
  SigmaLay l;
  l.nb = nb;
  l.x = 0; l.r = 1; l.c = 2; l.z = 3; l.negc = 4; l.minus_one = 5;
  l.g = 0; l.a = nb; l.A = 2 * nb; l.chk = 3 * nb;
  return l;
}

// The nonce is HEDGED ("sigma transcript v2"): r = Fr::rand(ChaCha20Rng(s2)) with
//   s1 = Blake2s(witness (32 B canonical) || fs_init (32 B) || prover_seed),  s2 = Blake2s(ToBytes(g.., a..) || s1)
// so that a repeated seed does not repeat the nonce unless witness AND statement repeat as well (the reference's `rng: &mut R`
// advances by itself; an explicit seed does not -- with the bare seed two proofs for one secret would reveal it).
struct SigmaInitArgs {
  uint32_t* S;
  const uint8_t* seeds;   // [B][32]
  SigmaLay l;
  uint32_t Bpad;
  FsDev f;
  const uint32_t* P;
  const uint8_t* fs_init; // [B][32]
};
template <class C>
MP_HD void body_sigma_init(const SigmaInitArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  uint32_t key[8];
  const uint32_t* sw = reinterpret_cast<const uint32_t*>(a.seeds + (size_t)b * 32);
#pragma unroll
  for (int i = 0; i < 8; ++i) key[i] = sw[i];
  {
    StageWriter w = stage_begin(a.f.stage, a.f.Bpad, b);
    uint32_t k[8];
    fe_to_canonical<R>(ld_fe<R>(a.S + s_off(a.l.x, a.Bpad, b)), k);
#pragma unroll
    for (int i = 0; i < 8; ++i) stage_word(w, k[i]);
    const uint32_t* fw = reinterpret_cast<const uint32_t*>(a.fs_init + (size_t)b * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) stage_word(w, fw[i]);
    fs_finish_absorb(w, key);                                   // s1
  }
  fs_absorb_points<C>(a.f, a.P, b, key, a.l.g, 2 * a.l.nb);     // s2: bases and publics are consecutive P slots
  FrStream st;
  frstream_init(st, key);
  st_fe<R>(a.S + s_off(a.l.r, a.Bpad, b), frstream_next<R>(st));
}
MP_KERNEL(k_sigma_init, SigmaInitArgs, body_sigma_init)

struct SigmaFsArgs {
  FsDev f;
  uint32_t* S;
  const uint32_t* P;
  const uint8_t* fs_init;   // [B][32]: Blake2s digest the FiatShamirRng is seeded with
  SigmaLay l;
  uint32_t prove;           // 1: z = r + c x ; 0: -c, -1 for the verification MSMs
};
template <class C>
MP_HD void body_sigma_fs(const SigmaFsArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  const SigmaLay& l = a.l;
  uint32_t seed[8];
  const uint32_t* sw = reinterpret_cast<const uint32_t*>(a.fs_init + (size_t)b * 32);
#pragma unroll
  for (int i = 0; i < 8; ++i) seed[i] = sw[i];
  fs_absorb_points<C>(a.f, a.P, b, seed, 0, 3 * l.nb);     // g.., a.., A.. are consecutive P slots
  fs_challenges<C>(seed, a.S, a.f.Bpad, b, l.c, NO_SLOT);
  const Fe<R> c = ld_fe<R>(a.S + s_off(l.c, a.f.Bpad, b));
  if (a.prove) {
    const Fe<R> z = fe_add<R>(ld_fe<R>(a.S + s_off(l.r, a.f.Bpad, b)), fe_mul<R>(c, ld_fe<R>(a.S + s_off(l.x, a.f.Bpad, b))));
    st_fe<R>(a.S + s_off(l.z, a.f.Bpad, b), z);
  } else {
    st_fe<R>(a.S + s_off(l.negc, a.f.Bpad, b), fe_neg<R>(c));
    st_fe<R>(a.S + s_off(l.minus_one, a.f.Bpad, b), fe_neg<R>(fe_one<R>()));
  }
}
MP_KERNEL(k_sigma_fs, SigmaFsArgs, body_sigma_fs)

struct SigmaIoArgs {
  uint8_t* proofs;          // [B][nb * point bytes + 32]
  uint32_t* S;
  uint32_t* P;
  int32_t* status;
  SigmaLay l;
  uint32_t Bpad;
};
// y < nb: commitment A_y ; y == nb: response z
template <class C>
MP_HD void body_sigma_store(const SigmaIoArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  uint8_t* dst = a.proofs + (size_t)b * (a.l.nb * Geo<C>::PB + 32);
  if (y < a.l.nb)
    aff_to_wire<C>(ld_aff<C>(a.P + p_off<C>(a.l.A + y, a.Bpad, b)), dst + Geo<C>::PB * y);
  else
    fe_to_wire<R>(ld_fe<R>(a.S + s_off(a.l.z, a.Bpad, b)), dst + Geo<C>::PB * a.l.nb);
}
MP_KERNEL(k_sigma_store, SigmaIoArgs, body_sigma_store)
template <class C>
MP_HD void body_sigma_load(const SigmaIoArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  const uint8_t* src = a.proofs + (size_t)b * (a.l.nb * Geo<C>::PB + 32);
  if (y < a.l.nb) {
    Aff<C> pt;
    if (!wire_to_aff<C>(src + Geo<C>::PB * y, pt)) {
      status_fail(a.status, b, ST_BAD_ENCODING);
      pt = aff_inf<C>();
    }
    st_aff<C>(a.P + p_off<C>(a.l.A + y, a.Bpad, b), pt);
  } else {
    Fe<R> v;
    if (!wire_to_fe<R>(src + Geo<C>::PB * a.l.nb, v)) {
      status_fail(a.status, b, ST_BAD_ENCODING);
      v = fe_zero<R>();
    }
    st_fe<R>(a.S + s_off(a.l.z, a.Bpad, b), v);
  }
}
MP_KERNEL(k_sigma_load, SigmaIoArgs, body_sigma_load)

struct SigmaVerdictArgs {
  const uint32_t* J;
  int32_t* status;
  SigmaLay l;
  uint32_t Bpad;
  int32_t fail_code;        // 5 "Schnorr Identification" / 6 "Chaum-Pedersen"
};
template <class C>
MP_HD void body_sigma_verdict(const SigmaVerdictArgs& a, uint32_t b, uint32_t y) {
  if (a.status[b] < 0) return;
  bool ok = true;
  for (uint32_t i = 0; i < a.l.nb; ++i) ok &= fe_is_zero(ld_fe<typename C::FqP>(a.J + j_off<C>(a.l.chk + i, a.Bpad, b) + 2 * Geo<C>::FW));
  a.status[b] = ok ? 0 : a.fail_code;
}
MP_KERNEL(k_sigma_verdict, SigmaVerdictArgs, body_sigma_verdict)

}  // namespace mp
