// Protocol-side kernels of the shuffle engine: everything of the Bayer-Groth prover / verifier that is NOT
// group arithmetic.  One lane = one proof (64 proofs per wave, all doing identical control flow).
//   * wire <-> arena conversion with validation (canonical range, on-curve, permutation)
//   * prover randomness: ChaCha20Rng(prover_seed) -> Fr::rand draws in transcript-v1 order
//   * Fiat-Shamir rounds: FiatShamirRng<Blake2s> absorb / squeeze on device (hash.hpp)
//   * the Fr vector algebra between rounds (powers, Hadamard products, bilinear map, responses)
//   * verifier: challenge recomputation, MSM coefficients, direct checks and the verdict
// Restates what `shuffle::ShuffleArgument::prove / verify` do between their group operations
// [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:409-415, 437-442]; the algebra follows
// Bayer-Groth sections 4-5.3 as frozen in oracle/py/mp_oracle.py (transcript v1).
#pragma once
#include "hash.hpp"
#include "kernels_msm.hpp"

namespace mp {

enum : int32_t { ST_OK = 0, ST_BAD_ENCODING = -1, ST_BAD_PERMUTATION = -2 };

MP_HD void status_fail(int32_t* status, uint32_t b, int32_t code) {
  // several lanes may report the same proof; any of the (negative) codes is acceptable
  status[b] = code;
}

// wire field element (4 F::NW bytes little-endian canonical: 32, or 48 for BLS12-377 Fq) -> Montgomery; false if >= modulus
template <class F>
MP_HD bool wire_to_fe(const uint8_t* p, Fe<F>& out) {
  uint32_t k[F::NW];
  const uint32_t* q = reinterpret_cast<const uint32_t*>(p);   // wire buffers are 4-byte aligned (API contract)
#pragma unroll
  for (int i = 0; i < F::NW; ++i) k[i] = q[i];
  const bool ok = fe_canonical_in_range<F>(k);
  out = fe_from_canonical<F>(k);
  return ok;
}
template <class F>
MP_HD void fe_to_wire(const Fe<F>& a, uint8_t* p) {
  uint32_t k[F::NW];
  fe_to_canonical<F>(a, k);
  uint32_t* q = reinterpret_cast<uint32_t*>(p);
#pragma unroll
  for (int i = 0; i < F::NW; ++i) q[i] = k[i];
}
// wire point: x || y (Geo<C>::PB = 64 B; 96 B on BLS12-377); infinity = all-zero bytes.  false if a coordinate is out of range or the point
// is not on the curve.
template <class C>
MP_HD bool wire_to_aff(const uint8_t* p, Aff<C>& out) {
  typedef typename C::FqP F;
  bool ok = wire_to_fe<F>(p, out.x);
  ok &= wire_to_fe<F>(p + Geo<C>::FB, out.y);
  if (aff_is_inf<C>(out)) return ok;
  return ok && aff_on_curve<C>(out);
}
template <class C>
MP_HD void aff_to_wire(const Aff<C>& a, uint8_t* p) {
  typedef typename C::FqP F;
  fe_to_wire<F>(a.x, p);       // (0,0) Montgomery = (0,0) canonical = the infinity encoding
  fe_to_wire<F>(a.y, p + Geo<C>::FB);
}

// ---- load / store -------------------------------------------------------------------------------------
struct LoadPointsArgs {
  const uint8_t* src;    // [B][count][point bytes]
  uint32_t* P;
  int32_t* status;
  uint32_t Bpad, count, p_slot;
  // optional: keep the wire words of every point, [w_slot + y][Bpad][words] (the layout of the P arena: 16-byte vector accesses).  A valid wire point IS its canonical coordinates, i.e.
  // what the transcript hashes: the Fiat-Shamir lane then stages these words instead of taking every coordinate out of Montgomery
  // form again (one field multiplication each on the one lane a proof's transcript has)
  uint32_t* W = nullptr;
  uint32_t w_slot = 0;
};
// x = proof, y = point index (a deck of N cards is 2N points: c0, c1 of card i at 2i, 2i+1)
template <class C>
MP_HD void body_load_points(const LoadPointsArgs& a, uint32_t b, uint32_t y) {
  Aff<C> pt;
  const uint8_t* src = a.src + ((size_t)b * a.count + y) * Geo<C>::PB;
  const bool ok = wire_to_aff<C>(src, pt);
  if (!ok) {
    status_fail(a.status, b, ST_BAD_ENCODING);
    pt = aff_inf<C>();
  }
  st_aff<C>(a.P + p_off<C>(a.p_slot + y, a.Bpad, b), pt);
  if (a.W) {
    constexpr uint32_t WW = Geo<C>::PB / 4;
    const uint32_t* q = reinterpret_cast<const uint32_t*>(src);
    uint32_t k[WW];
#pragma unroll
    for (uint32_t i = 0; i < WW; ++i) k[i] = q[i];
    st_words<WW>(a.W + ((size_t)(a.w_slot + y) * a.Bpad + b) * WW, k);
  }
}
MP_KERNEL(k_load_points, LoadPointsArgs, body_load_points)

struct LoadScalarsArgs {
  const uint8_t* src;    // [B][count][32]
  uint32_t* S;
  int32_t* status;
  uint32_t Bpad, count, s_slot;
};
template <class C>
MP_HD void body_load_scalars(const LoadScalarsArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  Fe<R> v;
  if (!wire_to_fe<R>(a.src + ((size_t)b * a.count + y) * 32, v)) {
    status_fail(a.status, b, ST_BAD_ENCODING);
    v = fe_zero<R>();
  }
  st_fe<R>(a.S + s_off(a.s_slot + y, a.Bpad, b), v);
}
MP_KERNEL(k_load_scalars, LoadScalarsArgs, body_load_scalars)

struct ProofIoArgs {
  uint8_t* proof;        // [B][proof_bytes]
  uint32_t* S;
  uint32_t* P;
  int32_t* status;
  const ProofElem* map;
  uint32_t Bpad, proof_bytes;
};
template <class C>
MP_HD void body_load_proof(const ProofIoArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  const ProofElem e = a.map[y];
  const uint8_t* src = a.proof + (size_t)b * a.proof_bytes + e.offset;
  if (e.is_point) {
    Aff<C> pt;
    if (!wire_to_aff<C>(src, pt)) {
      status_fail(a.status, b, ST_BAD_ENCODING);
      pt = aff_inf<C>();
    }
    st_aff<C>(a.P + p_off<C>(e.slot, a.Bpad, b), pt);
  } else {
    Fe<R> v;
    if (!wire_to_fe<R>(src, v)) {
      status_fail(a.status, b, ST_BAD_ENCODING);
      v = fe_zero<R>();
    }
    st_fe<R>(a.S + s_off(e.slot, a.Bpad, b), v);
  }
}
MP_KERNEL(k_load_proof, ProofIoArgs, body_load_proof)
template <class C>
MP_HD void body_store_proof(const ProofIoArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  const ProofElem e = a.map[y];
  uint8_t* dst = a.proof + (size_t)b * a.proof_bytes + e.offset;
  if (e.is_point)
    aff_to_wire<C>(ld_aff<C>(a.P + p_off<C>(e.slot, a.Bpad, b)), dst);
  else
    fe_to_wire<R>(ld_fe<R>(a.S + s_off(e.slot, a.Bpad, b)), dst);
}
MP_KERNEL(k_store_proof, ProofIoArgs, body_store_proof)

struct StorePointsArgs {
  uint8_t* dst;          // [B][count][point bytes]
  const uint32_t* P;
  uint32_t Bpad, count, p_slot;
};
template <class C>
MP_HD void body_store_points(const StorePointsArgs& a, uint32_t b, uint32_t y) {
  aff_to_wire<C>(ld_aff<C>(a.P + p_off<C>(a.p_slot + y, a.Bpad, b)), a.dst + ((size_t)b * a.count + y) * Geo<C>::PB);
}
MP_KERNEL(k_store_points, StorePointsArgs, body_store_points)

// ---- prover: permutation check, a = pi + 1, randomness ---------------------------------------------
struct ProveInitArgs {
  uint32_t* S;
  int32_t* status;
  const uint32_t* perm;         // [B][N]
  const uint8_t* seeds;         // [B][32]
  const uint32_t* draw_slots;   // [n_draws]
  ProveLay l;
  uint32_t Bpad;
};
template <class C>
MP_HD void body_prove_init(const ProveInitArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  const ProveLay& l = a.l;
  // permutation: every value < N and distinct.  N <= 4096: 128-word bitmap in private memory.
  uint32_t seen[128];
  for (uint32_t i = 0; i < 128; ++i) seen[i] = 0;
  bool ok = true;
  for (uint32_t i = 0; i < l.N; ++i) {
    uint32_t v = a.perm[(size_t)b * l.N + i];
    if (v >= l.N) {
      ok = false;
      v = 0;
    } else {
      if (seen[v >> 5] & (1u << (v & 31))) ok = false;
      seen[v >> 5] |= 1u << (v & 31);
    }
    st_fe<R>(a.S + s_off(l.a + i, a.Bpad, b), fe_from_u32<R>(v + 1));
  }
  if (!ok) status_fail(a.status, b, ST_BAD_PERMUTATION);
  // prover randomness
  uint32_t key[8];
  const uint32_t* sw = reinterpret_cast<const uint32_t*>(a.seeds + (size_t)b * 32);
#pragma unroll
  for (int i = 0; i < 8; ++i) key[i] = sw[i];
  // Rejection sampling, lane-efficiently: loop over CANDIDATES, not draws -- a lane whose candidate is rejected
  // simply keeps its draw index, so no lane waits for the unluckiest one of the wave at every draw
  // (Fr::rand accepts ~1/2 of the candidates on the STARK curve).
  FrStream st;
  frstream_init(st, key);
  uint32_t got = 0;
  while (got < l.n_draws) {
    Fe<R> v;
    if (frstream_try<R>(st, v)) {
      st_fe<R>(a.S + s_off(a.draw_slots[got], a.Bpad, b), v);
      ++got;
    }
  }
  // fixed values of transcript v1
  st_fe<R>(a.S + s_off(l.zt + l.m + 1, a.Bpad, b), fe_zero<R>());
  st_fe<R>(a.S + s_off(l.meb + l.m, a.Bpad, b), fe_zero<R>());
  st_fe<R>(a.S + s_off(l.mes + l.m, a.Bpad, b), fe_zero<R>());
  st_fe<R>(a.S + s_off(l.svdelta + l.n - 1, a.Bpad, b), fe_zero<R>());
}
MP_KERNEL(k_prove_init, ProveInitArgs, body_prove_init)
// The same with one WAVE per proof, for small batches (a lone lane spends 0.45 ms of a 52-card proof's 4.7 here, nearly all of it
// in ~300 sequential ChaCha20 blocks): lane i checks and converts cards i, i + 64, ...; the rejection sampler generates 64 blocks
// (128 candidates) at a time and a prefix sum over the accept flags gives every accepted candidate its place in the draw order --
// the values and their order are those of the sequential sampler, the surplus candidates of the last round are dropped.
// LDS: N counters (how often each card index occurs in the permutation).
template <class C, class W>
MP_HD void body_prove_init_w(const ProveInitArgs& a, uint32_t b, W& wv) {
  typedef typename C::FrP R;
  const ProveLay& l = a.l;
  uint32_t* cnt = wv.lds;
  wv.lanes([&](uint32_t lane) {
    for (uint32_t i = lane; i < l.N; i += 64) cnt[i] = 0;
  });
  wv.sync();
  PerLane<uint32_t> bad;
  wv.lanes([&](uint32_t lane) {
    bad[lane] = 0;
    for (uint32_t i = lane; i < l.N; i += 64) {
      uint32_t v = a.perm[(size_t)b * l.N + i];
      if (v >= l.N) {
        bad[lane] = 1;
        v = 0;
      } else {
        wv.atomic_add(&cnt[v], 1u);
      }
      st_fe<R>(a.S + s_off(l.a + i, a.Bpad, b), fe_from_u32<R>(v + 1));
    }
  });
  wv.sync();
  wv.lanes([&](uint32_t lane) {
    for (uint32_t i = lane; i < l.N; i += 64)
      if (cnt[i] > 1) bad[lane] = 1;
  });
  if (wv.any(bad)) wv.lanes([&](uint32_t lane) {
    if (lane == 0) status_fail(a.status, b, ST_BAD_PERMUTATION);
  });
  const uint32_t* sw = reinterpret_cast<const uint32_t*>(a.seeds + (size_t)b * 32);
  uint32_t got = 0;
  uint64_t counter = 0;
  while (got < l.n_draws) {
    PerLane<Fe<R>> c0, c1;
    PerLane<uint32_t> ok0, ok1, pos, incl;
    wv.lanes([&](uint32_t lane) {
      uint32_t key[8], blk[16];
#pragma unroll
      for (int i = 0; i < 8; ++i) key[i] = sw[i];
      chacha20_block(key, counter + lane, blk);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        c0[lane].v[i] = blk[i];
        c1[lane].v[i] = blk[8 + i];
      }
      if (R::BITS < 256) {
        c0[lane].v[7] &= 0xFFFFFFFFu >> (256 - R::BITS);
        c1[lane].v[7] &= 0xFFFFFFFFu >> (256 - R::BITS);
      }
      ok0[lane] = fe_canonical_in_range<R>(c0[lane].v) ? 1u : 0u;
      ok1[lane] = fe_canonical_in_range<R>(c1[lane].v) ? 1u : 0u;
      pos[lane] = ok0[lane] + ok1[lane];
      incl[lane] = pos[lane];
    });
    wv.excl_scan(pos);
    wv.lanes([&](uint32_t lane) {
      incl[lane] += pos[lane];
      const uint32_t i0 = got + pos[lane], i1 = i0 + ok0[lane];
      if (ok0[lane] && i0 < l.n_draws) st_fe<R>(a.S + s_off(a.draw_slots[i0], a.Bpad, b), c0[lane]);
      if (ok1[lane] && i1 < l.n_draws) st_fe<R>(a.S + s_off(a.draw_slots[i1], a.Bpad, b), c1[lane]);
    });
    got += wv.max(incl);
    counter += 64;
  }
  wv.lanes([&](uint32_t lane) {
    if (lane != 0) return;
    st_fe<R>(a.S + s_off(l.zt + l.m + 1, a.Bpad, b), fe_zero<R>());
    st_fe<R>(a.S + s_off(l.meb + l.m, a.Bpad, b), fe_zero<R>());
    st_fe<R>(a.S + s_off(l.mes + l.m, a.Bpad, b), fe_zero<R>());
    st_fe<R>(a.S + s_off(l.svdelta + l.n - 1, a.Bpad, b), fe_zero<R>());
  });
}
MP_WAVE_KERNEL(k_prove_init_w, ProveInitArgs, body_prove_init_w)

// ---- Fiat-Shamir helpers ------------------------------------------------------------------------------
struct FsDev {
  uint32_t* stage;       // [stage_words][Bpad]
  uint32_t* seed;        // [8][Bpad]
  uint32_t Bpad;
};
template <class C>
MP_HD void fs_put_point(StageWriter& w, const Aff<C>& p) {   // ark ToBytes: x || y (8 bytes per limb) || infinity flag
  typedef typename C::FqP F;
  constexpr int FW = F::NW;
  uint32_t k[FW];
  if (aff_is_inf<C>(p)) {   // GroupAffine::zero() = (0, 1, infinity = true)
    for (int i = 0; i < FW; ++i) stage_word(w, 0);
    stage_word(w, 1);
    for (int i = 1; i < FW; ++i) stage_word(w, 0);
    stage_byte(w, 1);
    return;
  }
  fe_to_canonical<F>(p.x, k);
#pragma unroll
  for (int i = 0; i < FW; ++i) stage_word(w, k[i]);
  fe_to_canonical<F>(p.y, k);
#pragma unroll
  for (int i = 0; i < FW; ++i) stage_word(w, k[i]);
  stage_byte(w, 0);
}
// the same bytes from the kept wire words of a loaded point (LoadPointsArgs::W): all-zero words are the wire form of infinity
template <class C>
MP_HD void fs_put_wire_point(StageWriter& w, const uint32_t* W, uint32_t slot, uint32_t Bpad, uint32_t b) {
  constexpr int FW = C::FqP::NW, WW = 2 * FW;
  uint32_t k[WW], nz = 0;
  ld_words<WW>(W + ((size_t)slot * Bpad + b) * WW, k);
#pragma unroll
  for (int i = 0; i < WW; ++i) nz |= k[i];
  if (nz == 0) {
    for (int i = 0; i < FW; ++i) stage_word(w, 0);
    stage_word(w, 1);
    for (int i = 1; i < FW; ++i) stage_word(w, 0);
    stage_byte(w, 1);
    return;
  }
#pragma unroll
  for (int i = 0; i < WW; ++i) stage_word(w, k[i]);
  stage_byte(w, 0);
}
MP_HD void fs_load_seed(const FsDev& f, uint32_t b, uint32_t seed[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) seed[i] = f.seed[(size_t)i * f.Bpad + b];
}
MP_HD void fs_store_seed(const FsDev& f, uint32_t b, const uint32_t seed[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f.seed[(size_t)i * f.Bpad + b] = seed[i];
}
// finish an absorb: append the old seed, hash, new seed
MP_HD void fs_finish_absorb(StageWriter& w, uint32_t seed[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) stage_word(w, seed[i]);
  blake2s_staged(w, seed);
}
// absorb the P slots [first, first+count)
template <class C>
MP_HD void fs_absorb_points(const FsDev& f, const uint32_t* P, uint32_t b, uint32_t seed[8], uint32_t first, uint32_t count) {
  StageWriter w = stage_begin(f.stage, f.Bpad, b);
  for (uint32_t i = 0; i < count; ++i) fs_put_point<C>(w, ld_aff<C>(P + p_off<C>(first + i, f.Bpad, b)));
  fs_finish_absorb(w, seed);
}
template <class C>
MP_HD void fs_challenges(const uint32_t seed[8], uint32_t* S, uint32_t Bpad, uint32_t b, uint32_t slot0, uint32_t slot1) {
  typedef typename C::FrP R;
  FrStream st;
  frstream_init(st, seed);
  st_fe<R>(S + s_off(slot0, Bpad, b), frstream_next<R>(st));
  if (slot1 != NO_SLOT) st_fe<R>(S + s_off(slot1, Bpad, b), frstream_next<R>(st));
}

// statement absorb + c_A absorb -> x        (shared by prover and verifier)
struct FsStatementArgs {
  FsDev f;
  uint32_t* S;
  const uint32_t* P;
  const uint32_t* fbpts;       // fixed base points (affine): ck.., H, G, pk, gen, gsum
  uint32_t init_seed[8];       // Blake2s("Shuffle Proof")
  uint32_t m, n, N;
  uint32_t p_deck, p_shuf, p_cA, s_x;
  uint32_t p_pk;               // keyed batches: P slot of the per-proof aggregate key (NO_SLOT: the table's fixed base)
  const uint32_t* W;           // kept wire words of the loaded decks (LoadPointsArgs::W), or null
  uint32_t w_deck, w_shuf;     // their first W slots (NO_SLOT: take the deck from its P slots)
};
// The statement message: G, pk, gen, ck_0 .. ck_{n-1}, H, the 2N points of the deck, the 2N of the shuffled deck (then u64 m, u64 n
// and the old seed).  Point i of that list:
MP_HD uint32_t fs_statement_points(const FsStatementArgs& a) { return 4 + a.n + 4 * a.N; }
template <class C>
MP_HD void fs_put_statement_point(StageWriter& w, const FsStatementArgs& a, uint32_t b, uint32_t i) {
  const FixedBases fb{a.n};
  const uint32_t nfix = 4 + a.n;
  if (i < nfix) {
    if (i == 1 && a.p_pk != NO_SLOT) {
      fs_put_point<C>(w, ld_aff<C>(a.P + p_off<C>(a.p_pk, a.f.Bpad, b)));
      return;
    }
    const uint32_t base = i == 0 ? fb.G() : i == 1 ? fb.pk() : i == 2 ? fb.gen() : i == nfix - 1 ? fb.H() : fb.ck(i - 3);
    fs_put_point<C>(w, ld_aff<C>(a.fbpts + (size_t)base * Geo<C>::PW));
    return;
  }
  i -= nfix;
  const bool second = i >= 2 * a.N;
  if (second) i -= 2 * a.N;
  const uint32_t wslot = second ? a.w_shuf : a.w_deck;
  if (a.W && wslot != NO_SLOT)
    fs_put_wire_point<C>(w, a.W, wslot + i, a.f.Bpad, b);
  else
    fs_put_point<C>(w, ld_aff<C>(a.P + p_off<C>((second ? a.p_shuf : a.p_deck) + i, a.f.Bpad, b)));
}
template <class C>
MP_HD void fs_statement_and_x(const FsStatementArgs& a, uint32_t b, uint32_t seed[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) seed[i] = a.init_seed[i];
  StageWriter w = stage_begin(a.f.stage, a.f.Bpad, b);
  const uint32_t npts = fs_statement_points(a);
  for (uint32_t i = 0; i < npts; ++i) fs_put_statement_point<C>(w, a, b, i);
  stage_word(w, a.m); stage_word(w, 0); stage_word(w, a.n); stage_word(w, 0);   // u64 m, u64 n
  fs_finish_absorb(w, seed);
  fs_absorb_points<C>(a.f, a.P, b, seed, a.p_cA, a.m);
  fs_challenges<C>(seed, a.S, a.f.Bpad, b, a.s_x, NO_SLOT);
}
template <class C>
MP_HD void body_fs_round1(const FsStatementArgs& a, uint32_t b, uint32_t y) {
  uint32_t seed[8];
  fs_statement_and_x<C>(a, b, seed);
  fs_store_seed(a.f, b, seed);
}
MP_KERNEL(k_fs_round1, FsStatementArgs, body_fs_round1)

// generic later rounds: up to 3 consecutive (absorb slot range -> challenges) steps in one kernel
struct FsStep {
  uint32_t first, count;       // P slots absorbed
  uint32_t first2, count2;     // optional second range absorbed in the same message
  uint32_t slot0, slot1;       // challenge destinations (slot0 == NO_SLOT: absorb only)
};
struct FsRoundArgs {
  FsDev f;
  uint32_t* S;
  uint32_t* P;
  FsStep step[4];
  uint32_t nsteps;
  uint32_t copy_from, copy_to;   // optional P-slot copy done first (NO_SLOT = none)
};
template <class C>
MP_HD void fs_run_steps(const FsRoundArgs& a, uint32_t b, uint32_t seed[8]) {
  for (uint32_t s = 0; s < a.nsteps; ++s) {
    const FsStep st = a.step[s];
    StageWriter w = stage_begin(a.f.stage, a.f.Bpad, b);
    for (uint32_t i = 0; i < st.count; ++i) fs_put_point<C>(w, ld_aff<C>(a.P + p_off<C>(st.first + i, a.f.Bpad, b)));
    for (uint32_t i = 0; i < st.count2; ++i) fs_put_point<C>(w, ld_aff<C>(a.P + p_off<C>(st.first2 + i, a.f.Bpad, b)));
    fs_finish_absorb(w, seed);
    if (st.slot0 != NO_SLOT) fs_challenges<C>(seed, a.S, a.f.Bpad, b, st.slot0, st.slot1);
  }
}
template <class C>
MP_HD void body_fs_round(const FsRoundArgs& a, uint32_t b, uint32_t y) {
  if (a.copy_from != NO_SLOT)
    st_aff<C>(a.P + p_off<C>(a.copy_to, a.f.Bpad, b), ld_aff<C>(a.P + p_off<C>(a.copy_from, a.f.Bpad, b)));
  uint32_t seed[8];
  fs_load_seed(a.f, b, seed);
  fs_run_steps<C>(a, b, seed);
  fs_store_seed(a.f, b, seed);
}
MP_KERNEL(k_fs_round, FsRoundArgs, body_fs_round)

// ---- prover scalar programs -----------------------------------------------------------------------------
struct ProveScalArgs {
  uint32_t* S;
  const uint32_t* perm;
  ProveLay l;
  uint32_t Bpad;
  const LinJob* lin;          // scalar-row sums of the Karatsuba plan (layout.hpp), evaluated at the end of scal1
  const uint32_t* lin_src;
  uint32_t n_lin;
  uint32_t d_parts;           // k_prove_scal3d: column ranges whose partial sums k_prove_scal3d_part left behind (0 = none: sum everything)
};
#define MP_LD(slot) ld_fe<R>(a.S + s_off((slot), a.Bpad, b))
#define MP_ST(slot, val) st_fe<R>(a.S + s_off((slot), a.Bpad, b), (val))

// The prover's scalar programs (Fr vector algebra between the Fiat-Shamir rounds).  They were one lane per proof; for every batch that
// does not fill the chip with such lanes that lane is what the proof waits for -- ~1 ms of a single 52-card proof's 3.0, 7 % of a
// 4 096-proof step -- and most of it is dependent loads, not arithmetic.  Each program is now cut along its data flow into lanes that
// need nothing from each other inside a launch (y = entry of a vector / column of the m x n matrices / one of the scalar results);
// sums of powers are Horner recurrences on the lane that owns them and single powers square-and-multiply, so no table of powers goes
// through memory.  Where a result needs every lane's output (rho_hat, the prefix products of the single-value product) a second
// small launch follows.  Same field elements, same bytes.
template <class R>
MP_HD Fe<R> fe_pow_small(const Fe<R>& x, uint32_t e) {      // x^e, e >= 1 (a card index / vector position: <= 13 bits)
  int top = 31;
  while (!((e >> top) & 1u)) --top;
  Fe<R> acc = x;
  for (int i = top - 1; i >= 0; --i) {
    acc = fe_sqr<R>(acc);
    if ((e >> i) & 1u) acc = fe_mul<R>(acc, x);
  }
  return acc;
}
// after x, first launch: y = i < N: b_i = x^(pi(i)+1) and the product rho_i b_i (tmp[i])
template <class C>
MP_HD void body_prove_scal1(const ProveScalArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  const ProveLay& l = a.l;
  uint32_t pi = a.perm[(size_t)b * l.N + y];
  if (pi >= l.N) pi = 0;
  const Fe<R> bi = fe_pow_small<R>(MP_LD(l.x), pi + 1);
  MP_ST(l.b + y, bi);
  MP_ST(l.tmp + y, fe_mul<R>(MP_LD(l.rho + y), bi));
}
MP_KERNEL(k_prove_scal1, ProveScalArgs, body_prove_scal1)
// second launch: y = j < n: column j of rho_hat's sum (sum_k rho_i b_i over the m rows, i = k n + j -> tmp[N + j]) and the halved
// evaluation scalars of the m = 2 Toom-Cook diagonals (layout.hpp);  y = n + q n + j: entry j of scalar-row sum q of the Karatsuba
// plan (m >= 3).  Third launch (k_prove_scal1c): rho_hat = -sum_j of the column sums -> tau[m].  (Until round 4 one lane added up all
// N products: 1 024 dependent loads on 256 waves -- 64 ms of a (16,64) step at 16 384 proofs.)
template <class C>
MP_HD void body_prove_scal1b(const ProveScalArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  const ProveLay& l = a.l;
  if (y < l.n) {
    const uint32_t j = y;
    Fe<R> col = MP_LD(l.tmp + j);
    for (uint32_t k = 1; k < l.m; ++k) col = fe_add<R>(col, MP_LD(l.tmp + k * l.n + j));
    MP_ST(l.tmp + l.N + j, col);
    if (!l.toom) return;
    const Fe<R> a0 = MP_LD(l.mea0 + j), a1 = MP_LD(l.b + j), a2 = MP_LD(l.b + l.n + j);
    const Fe<R> e = fe_add<R>(a0, a2);
    MP_ST(l.tsp + j, fe_half<R>(fe_add<R>(e, a1)));
    MP_ST(l.tsm + j, fe_half<R>(fe_sub<R>(e, a1)));
    return;
  }
  y -= l.n;
  const LinJob lj = a.lin[y / l.n];
  const uint32_t j = y % l.n;
  Fe<R> acc = MP_LD(a.lin_src[lj.begin] + j);
  for (uint32_t i = 1; i < lj.count; ++i) acc = fe_add<R>(acc, MP_LD(a.lin_src[lj.begin + i] + j));
  MP_ST(lj.dst + j, acc);
}
template <class C>
MP_HD void body_prove_scal1c(const ProveScalArgs& a, uint32_t b, uint32_t) {
  typedef typename C::FrP R;
  const ProveLay& l = a.l;
  Fe<R> rho_hat = fe_zero<R>();
  for (uint32_t j = 0; j < l.n; ++j) rho_hat = fe_sub<R>(rho_hat, MP_LD(l.tmp + l.N + j));
  MP_ST(l.metau + l.m, rho_hat);
}
MP_KERNEL(k_prove_scal1c, ProveScalArgs, body_prove_scal1c)
MP_KERNEL(k_prove_scal1b, ProveScalArgs, body_prove_scal1b)

// ---- Toom-Cook operands (3 <= m <= 16, layout.hpp ToomPlan) ---------------------------------------------------------------
// scalar side: S[dst + t] = sum_i consts[coef_i] * S[src_i + t]   (x = proof, y = job * n + t: one lane per output scalar)
struct LinCombArgs {
  uint32_t* S;
  const LinJob* lin;
  const uint32_t* lin_src;
  const uint32_t* lin_coef;
  const uint32_t* consts;     // Fr constants, Montgomery, 8 words each
  uint32_t Bpad, n;
};
template <class C>
MP_HD void body_lin_comb(const LinCombArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  const LinJob lj = a.lin[y / a.n];
  const uint32_t t = y % a.n;
  Fe<R> acc = fe_zero<R>();
  for (uint32_t i = 0; i < lj.count; ++i) {
    const Fe<R> c = ld_fe<R>(a.consts + (size_t)a.lin_coef[lj.begin + i] * 8);
    acc = fe_add<R>(acc, fe_mul<R>(c, ld_fe<R>(a.S + s_off(a.lin_src[lj.begin + i] + t, a.Bpad, b))));
  }
  st_fe<R>(a.S + s_off(lj.dst + t, a.Bpad, b), acc);
}
MP_KERNEL(k_lin_comb, LinCombArgs, body_lin_comb)
// constants into S slots (the interpolation matrix as ordinary MSM scalars): x = proof, y = constant
struct FillConstArgs {
  uint32_t* S;
  const uint32_t* consts;
  uint32_t Bpad, s_first, c_first;
};
template <class C>
MP_HD void body_fill_consts(const FillConstArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  st_fe<R>(a.S + s_off(a.s_first + y, a.Bpad, b), ld_fe<R>(a.consts + (size_t)(a.c_first + y) * 8));
}
MP_KERNEL(k_fill_consts, FillConstArgs, body_fill_consts)

// after y, z, first launch: y < n: column j of d - z and of the Hadamard partial products;  y = n: t and the Hadamard blinders
template <class C>
MP_HD void body_prove_scal2(const ProveScalArgs& a, uint32_t b, uint32_t y_) {
  typedef typename C::FrP R;
  const ProveLay& l = a.l;
  const uint32_t m = l.m, n = l.n;
  const Fe<R> y = MP_LD(l.y);
  if (y_ == n) {
    for (uint32_t k = 0; k < m; ++k) {
      const Fe<R> tk = fe_add<R>(fe_mul<R>(y, MP_LD(l.r + k)), MP_LD(l.s + k));
      MP_ST(l.t + k, tk);
      if (k == 0) MP_ST(l.hs + 0, tk);
    }
    MP_ST(l.hs + m - 1, MP_LD(l.sb));
    return;
  }
  const uint32_t j = y_;
  const Fe<R> z = MP_LD(l.z);
  Fe<R> acc;
  for (uint32_t k = 0; k < m; ++k) {
    const uint32_t i = k * n + j;
    const Fe<R> dz = fe_sub<R>(fe_add<R>(fe_mul<R>(y, MP_LD(l.a + i)), MP_LD(l.b + i)), z);
    MP_ST(l.dz + i, dz);
    acc = k == 0 ? dz : fe_mul<R>(acc, dz);
    MP_ST(l.bp + i, acc);
  }
}
MP_KERNEL(k_prove_scal2, ProveScalArgs, body_prove_scal2)
// second launch, single value product on a = bvec = bp[m-1] (randomness sb): y = i < n: the prefix product b_i = a_0 .. a_i (every
// lane multiplies its own -- n products deep like the one chain was, but nobody waits for it) and the first-message entries i
template <class C>
MP_HD void body_prove_scal2b(const ProveScalArgs& a, uint32_t b, uint32_t i) {
  typedef typename C::FrP R;
  const ProveLay& l = a.l;
  const uint32_t n = l.n, av = l.bp + (l.m - 1) * n;
  Fe<R> pref = MP_LD(av);
  for (uint32_t t = 1; t <= i; ++t) pref = fe_mul<R>(pref, MP_LD(av + t));
  MP_ST(l.svbp + i, pref);
  Fe<R> deli;
  if (i == 0) {
    deli = MP_LD(l.svd + 0);
    MP_ST(l.svdelta + 0, deli);
  } else {
    deli = MP_LD(l.svdelta + i);
  }
  if (i + 1 < n) {
    const Fe<R> di1 = MP_LD(l.svd + i + 1);
    MP_ST(l.svv1 + i, fe_neg<R>(fe_mul<R>(deli, di1)));
    Fe<R> v2 = fe_sub<R>(MP_LD(l.svdelta + i + 1), fe_mul<R>(MP_LD(av + i + 1), deli));
    v2 = fe_sub<R>(v2, fe_mul<R>(pref, di1));
    MP_ST(l.svv2 + i, v2);
  }
}
MP_KERNEL(k_prove_scal2b, ProveScalArgs, body_prove_scal2b)

// zero-argument witness accessors (rows a_0..a_m and b_0..b_m of section 5.2)
template <class C>
MP_HD Fe<typename C::FrP> zero_Aa(const ProveScalArgs& a, uint32_t b, uint32_t i, uint32_t j) {
  typedef typename C::FrP R;
  const ProveLay& l = a.l;
  if (i == 0) return MP_LD(l.za0 + j);
  if (i == l.m) return fe_neg<R>(fe_one<R>());
  return MP_LD(l.dz + i * l.n + j);
}
template <class C>
MP_HD Fe<typename C::FrP> zero_Bb(const ProveScalArgs& a, uint32_t b, uint32_t i, uint32_t j) {
  typedef typename C::FrP R;
  const ProveLay& l = a.l;
  if (i == l.m) return MP_LD(l.zbm + j);
  return MP_LD(l.zB + i * l.n + j);
}

// after the Hadamard challenges (hx, hy): zero-argument statement witness.  y = j < n: column j of the rows zB and of the weighted rows
// wb[jj][j] = Bb[jj][j] hy^(j+1) (tmp + m + 1 + n: read by k_prove_scal3d, which computes the d_k one lane per k);  y = n: zs
template <class C>
MP_HD void body_prove_scal3(const ProveScalArgs& a, uint32_t b, uint32_t y_) {
  typedef typename C::FrP R;
  const ProveLay& l = a.l;
  const uint32_t m = l.m, n = l.n;
  const Fe<R> hx = MP_LD(l.hx);
  if (y_ == n) {
    Fe<R> last = fe_zero<R>(), xi = hx;                    // xi = hx^(i+1)
    for (uint32_t i = 0; i + 1 < m; ++i) {
      MP_ST(l.zs + i, fe_mul<R>(xi, MP_LD(l.hs + i)));
      last = fe_add<R>(last, fe_mul<R>(xi, MP_LD(l.hs + i + 1)));
      xi = fe_mul<R>(xi, hx);
    }
    MP_ST(l.zs + m - 1, last);
    return;
  }
  const uint32_t j = y_, t_wb = l.tmp + m + 1 + n;
  const Fe<R> yp = fe_pow_small<R>(MP_LD(l.hy), j + 1);
  Fe<R> last = fe_zero<R>(), xi = hx;
  for (uint32_t i = 0; i + 1 < m; ++i) {
    const Fe<R> v = fe_mul<R>(xi, MP_LD(l.bp + i * n + j));
    MP_ST(l.zB + i * n + j, v);
    MP_ST(t_wb + i * n + j, fe_mul<R>(v, yp));
    last = fe_add<R>(last, fe_mul<R>(xi, MP_LD(l.bp + (i + 1) * n + j)));
    xi = fe_mul<R>(xi, hx);
  }
  MP_ST(l.zB + (m - 1) * n + j, last);
  MP_ST(t_wb + (m - 1) * n + j, fe_mul<R>(last, yp));
  MP_ST(t_wb + m * n + j, fe_mul<R>(MP_LD(l.zbm + j), yp));
}
MP_KERNEL(k_prove_scal3, ProveScalArgs, body_prove_scal3)

// d_k = sum_{i,jj : k = m - jj + i} Aa[i] . wb[jj]   (y = k in [0, 2m]): O(m n) per lane instead of O(m^2 n) in one lane.
// Small batches (round 5): the (m + 1) n products of a d_k in one lane were 0.1 ms of a single 52-card proof -- 78 dependent
// multiply-adds on five lanes; k_prove_scal3d_part first leaves the sums over d_parts column ranges (y = k d_parts + p) behind the wb
// rows in tmp, and the lane of d_k adds those up.  (Field addition is exact: the value does not depend on the association.)
MP_HD uint32_t scal3d_parts(uint32_t m, uint32_t n) {      // column ranges that fit the free tail of tmp (layout.hpp make_prove_lay)
  const uint32_t room = (n + m + m * n + 7u) / (2u * m + 1u);
  const uint32_t p = n < 8u ? n : 8u;
  return p < room ? p : room;
}
template <class C>
MP_HD Fe<typename C::FrP> scal3d_range(const ProveScalArgs& a, uint32_t b, uint32_t k, uint32_t j0, uint32_t j1) {
  typedef typename C::FrP R;
  const ProveLay& l = a.l;
  const uint32_t m = l.m, n = l.n;
  const uint32_t t_wb = l.tmp + m + 1 + n;
  Fe<R> d = fe_zero<R>();
  for (uint32_t i = 0; i <= m; ++i) {
    const int64_t jj = (int64_t)m + (int64_t)i - (int64_t)k;
    if (jj < 0 || jj > (int64_t)m) continue;
    for (uint32_t j = j0; j < j1; ++j) d = fe_add<R>(d, fe_mul<R>(zero_Aa<C>(a, b, i, j), MP_LD(t_wb + (uint32_t)jj * n + j)));
  }
  return d;
}
template <class C>
MP_HD void body_prove_scal3d_part(const ProveScalArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  const ProveLay& l = a.l;
  const uint32_t k = y / a.d_parts, p = y % a.d_parts;
  const uint32_t t_part = l.tmp + l.m + 1 + l.n + (l.m + 1) * l.n;
  MP_ST(t_part + y, scal3d_range<C>(a, b, k, p * l.n / a.d_parts, (p + 1) * l.n / a.d_parts));
}
MP_KERNEL(k_prove_scal3d_part, ProveScalArgs, body_prove_scal3d_part)
template <class C>
MP_HD void body_prove_scal3d(const ProveScalArgs& a, uint32_t b, uint32_t k) {
  typedef typename C::FrP R;
  const ProveLay& l = a.l;
  if (a.d_parts) {
    const uint32_t t_part = l.tmp + l.m + 1 + l.n + (l.m + 1) * l.n;
    Fe<R> d = MP_LD(t_part + k * a.d_parts);
    for (uint32_t p = 1; p < a.d_parts; ++p) d = fe_add<R>(d, MP_LD(t_part + k * a.d_parts + p));
    MP_ST(l.zd + k, d);
    return;
  }
  MP_ST(l.zd + k, scal3d_range<C>(a, b, k, 0, l.n));
}
MP_KERNEL(k_prove_scal3d, ProveScalArgs, body_prove_scal3d)

// after the last challenges: all responses.  y < n: entry y of the five response vectors; y = n, n + 1, n + 2: the scalar responses of
// the zero argument, the single-value product and the multi-exponentiation argument.  Every sum sum_i x^i v_i is a Horner recurrence
// on the lane that owns it (no table of powers shared through memory, so the lanes of a proof need nothing from each other): a
// response costs one lane ~3m + 4 products instead of one lane per proof ~(3m + 4)(n + 3) in a row -- 0.35 ms of a single proof.
template <class C>
MP_HD void body_prove_scal4(const ProveScalArgs& a, uint32_t b, uint32_t y_) {
  typedef typename C::FrP R;
  const ProveLay& l = a.l;
  const uint32_t m = l.m, n = l.n;
  if (y_ < n) {
    const uint32_t j = y_;
    {   // zero argument: abar_j = sum_i x^i A_i[j], bbar_j = sum_i x^(m-i) B_i[j]
      const Fe<R> x = MP_LD(l.zx);
      Fe<R> ab = zero_Aa<C>(a, b, m, j), bb = zero_Bb<C>(a, b, 0, j);
      for (uint32_t i = m; i-- > 0;) ab = fe_add<R>(fe_mul<R>(ab, x), zero_Aa<C>(a, b, i, j));
      for (uint32_t i = 1; i <= m; ++i) bb = fe_add<R>(fe_mul<R>(bb, x), zero_Bb<C>(a, b, i, j));
      MP_ST(l.zabar + j, ab);
      MP_ST(l.zbbar + j, bb);
    }
    {   // single value product
      const Fe<R> x = MP_LD(l.svx);
      MP_ST(l.svat + j, fe_add<R>(fe_mul<R>(x, MP_LD(l.bp + (m - 1) * n + j)), MP_LD(l.svd + j)));
      MP_ST(l.svbt + j, fe_add<R>(fe_mul<R>(x, MP_LD(l.svbp + j)), MP_LD(l.svdelta + j)));
    }
    {   // multi-exponentiation: abar_j = a0_j + sum_{i=1..m} x^i b_{i-1}[j]
      const Fe<R> x = MP_LD(l.mx);
      Fe<R> ab = MP_LD(l.b + (m - 1) * n + j);
      for (uint32_t i = m - 1; i >= 1; --i) ab = fe_add<R>(fe_mul<R>(ab, x), MP_LD(l.b + (i - 1) * n + j));
      MP_ST(l.meabar + j, fe_add<R>(fe_mul<R>(ab, x), MP_LD(l.mea0 + j)));
    }
    return;
  }
  if (y_ == n) {   // zero argument: rbar = r0 + sum_{i=1..m-1} x^i t_i  (r' = (t_1..t_{m-1}, 0)), sbar = sum_{j<m} x^(m-j) s_j + s_m, tbar = sum_k x^k t_k
    const Fe<R> x = MP_LD(l.zx);
    Fe<R> rb = fe_zero<R>(), sbar = fe_zero<R>(), tb = fe_zero<R>();
    for (uint32_t i = m - 1; i >= 1; --i) rb = fe_mul<R>(fe_add<R>(rb, MP_LD(l.t + i)), x);
    for (uint32_t j = 0; j < m; ++j) sbar = fe_mul<R>(fe_add<R>(sbar, MP_LD(l.zs + j)), x);
    for (uint32_t k = 2 * m + 1; k-- > 0;) tb = fe_add<R>(fe_mul<R>(tb, x), MP_LD(l.zt + k));
    MP_ST(l.zrbar, fe_add<R>(rb, MP_LD(l.zr0)));
    MP_ST(l.zsbar, fe_add<R>(sbar, MP_LD(l.zsm)));
    MP_ST(l.ztbar, tb);
    return;
  }
  if (y_ == n + 1) {
    const Fe<R> x = MP_LD(l.svx);
    MP_ST(l.svrt, fe_add<R>(fe_mul<R>(x, MP_LD(l.sb)), MP_LD(l.svrd)));
    MP_ST(l.svst, fe_add<R>(fe_mul<R>(x, MP_LD(l.svsx)), MP_LD(l.svs1)));
    return;
  }
  {   // multi-exponentiation: rbar = r0 + sum_{i=1..m} x^i s_{i-1}; bbar, sbar, taubar = sum_{k<2m} x^k (b_k, s_k, tau_k)
    const Fe<R> x = MP_LD(l.mx);
    Fe<R> rb = fe_zero<R>(), bb = fe_zero<R>(), sbar = fe_zero<R>(), tb = fe_zero<R>();
    for (uint32_t i = m; i >= 1; --i) rb = fe_mul<R>(fe_add<R>(rb, MP_LD(l.s + i - 1)), x);
    for (uint32_t k = 2 * m; k-- > 0;) {
      bb = fe_add<R>(fe_mul<R>(bb, x), MP_LD(l.meb + k));
      sbar = fe_add<R>(fe_mul<R>(sbar, x), MP_LD(l.mes + k));
      tb = fe_add<R>(fe_mul<R>(tb, x), MP_LD(l.metau + k));
    }
    MP_ST(l.merbar, fe_add<R>(rb, MP_LD(l.mer0)));
    MP_ST(l.mebbar, bb);
    MP_ST(l.mesbar, sbar);
    MP_ST(l.metaubar, tb);
  }
}
MP_KERNEL(k_prove_scal4, ProveScalArgs, body_prove_scal4)

// ---- verifier -------------------------------------------------------------------------------------------
struct VerifyFsArgs {
  FsStatementArgs st;
  VerifyLay l;
  uint32_t merge;      // 1: also derive the weights r_k of the merged equation (verifier-only randomness)
};
template <class C>
MP_HD void body_verify_fs(const VerifyFsArgs& a, uint32_t b, uint32_t y) {
  const VerifyLay& l = a.l;
  const uint32_t m = l.m;
  uint32_t seed[8];
  fs_statement_and_x<C>(a.st, b, seed);
  const FsDev& f = a.st.f;
  uint32_t* S = a.st.S;
  const uint32_t* P = a.st.P;
  fs_absorb_points<C>(f, P, b, seed, l.cB, m);
  fs_challenges<C>(seed, S, f.Bpad, b, l.y, l.z);
  fs_absorb_points<C>(f, P, b, seed, l.cb, 1);
  fs_absorb_points<C>(f, P, b, seed, l.hB, m);
  fs_challenges<C>(seed, S, f.Bpad, b, l.hx, l.hy);
  fs_absorb_points<C>(f, P, b, seed, l.zcA0, 2 + 2 * m + 1);       // zcA0, zcBm, zcD[0..2m] are consecutive slots
  fs_challenges<C>(seed, S, f.Bpad, b, l.zx, NO_SLOT);
  fs_absorb_points<C>(f, P, b, seed, l.svcd, 3);
  fs_challenges<C>(seed, S, f.Bpad, b, l.svx, NO_SLOT);
  fs_absorb_points<C>(f, P, b, seed, l.mecA0, 1 + 2 * m + 4 * m);  // mecA0, mecB[2m], meE[4m] consecutive
  fs_challenges<C>(seed, S, f.Bpad, b, l.mx, NO_SLOT);
  if (a.merge) {
    // Weights of the merged equation: they must depend on EVERY proof element, including the final responses that the
    // transcript itself never absorbs (a prover who knew r could trade errors between equations through them): absorb
    // the 5n+9 response scalars (slots 0 .. 5n+8, wire order) and squeeze one weight per check id.
    typedef typename C::FrP R;
    StageWriter w = stage_begin(f.stage, f.Bpad, b);
    for (uint32_t i = 0; i < 5 * l.n + 9; ++i) {
      uint32_t k[8];
      fe_to_canonical<R>(ld_fe<R>(S + s_off(l.zabar + i, f.Bpad, b)), k);
#pragma unroll
      for (int t = 0; t < 8; ++t) stage_word(w, k[t]);
    }
    fs_finish_absorb(w, seed);
    FrStream st;
    frstream_init(st, seed);
    for (uint32_t k = 0; k < (uint32_t)VC_COUNT; ++k) st_fe<R>(S + s_off(l.mr + k, f.Bpad, b), frstream_next<R>(st));
#pragma unroll
    for (int t = 0; t < 8; ++t) f.seed[(size_t)t * f.Bpad + b] = seed[t];     // the transcript's last state: chain verification hashes it
  }
}
MP_KERNEL(k_verify_fs, VerifyFsArgs, body_verify_fs)

// ---- the same transcripts with FOUR LANES per hash (hash.hpp: blake2s_compress_quad) ---------------------------------------------
// Used for batches that cannot fill the chip with one-lane transcripts (FSQ_MAX_BATCH, engine_core.hpp): a single 52-card proof
// waits 1.2 ms for its verifier's transcript lane and 1.1 ms for the prover's, a batch of 16 384 1024-card decks 20 ms per pass.
// A wave serves 64 / lpp proofs: the lpp lanes of a proof (4 .. 64, a power of two) share the staging of a message -- aligned groups
// of four points (32 FW + 4 bytes = 8 FW + 1 words) go round the lanes, the unaligned rest, the trailing words and the old seed to
// one lane of the first quad -- and the first quad hashes.  Same bytes, same digests, same challenges as the one-lane kernels.
struct FsqGeom {
  uint32_t lpp, B;      // lanes per proof; proofs in the batch
};
struct FsqLane {
  uint32_t b, sub;      // proof (clamped into the batch), lane within the proof
  bool live;            // the proof exists
};
MP_HD FsqLane fsq_lane(const FsqGeom& g, uint32_t wid, uint32_t l) {
  FsqLane q;
  const uint32_t b = wid * (64u / g.lpp) + l / g.lpp;
  q.live = b < g.B;
  q.b = q.live ? b : g.B - 1;
  q.sub = l & (g.lpp - 1u);
  return q;
}
// absorb: npts points (put_point(w, b, i) appends point i), `tail(w, b)` appends whole words, then the old seed; seed <- digest
template <class C, class W, class PutPoint, class PutTail>
MP_HD void fsq_absorb(W& wv, const FsDev& f, const FsqGeom& g, uint32_t wid, uint32_t npts, PutPoint put_point, uint32_t tail_words,
                      PutTail tail, PerLane<B2sSeed>& seed) {
  constexpr uint32_t GW = 8 * C::FqP::NW + 1;     // words of four points
  const uint32_t ngroups = npts / 4;
  PerLane<const uint32_t*> base;
  wv.lanes([&](uint32_t l) {
    const FsqLane q = fsq_lane(g, wid, l);
    base[l] = f.stage + q.b;
    if (!q.live) return;
    for (uint32_t gi = q.sub; gi < ngroups; gi += g.lpp) {
      StageWriter w = stage_begin_at(f.stage, f.Bpad, q.b, gi * GW);
      for (uint32_t k = 0; k < 4; ++k) put_point(w, q.b, 4 * gi + k);
    }
    if (q.sub == (ngroups & 3u)) {                 // the next lane of the first quad: it has the seed
      StageWriter w = stage_begin_at(f.stage, f.Bpad, q.b, ngroups * GW);
      for (uint32_t i = 4 * ngroups; i < npts; ++i) put_point(w, q.b, i);
      tail(w, q.b);
#pragma unroll
      for (int i = 0; i < 8; ++i) stage_word(w, seed[l].s[i]);
      stage_flush(w);
    }
  });
  wv.sync_global();
  blake2s_staged_quad(wv, base, f.Bpad, npts * GW + 4 * tail_words + 32, seed);   // (a point is GW BYTES)
  wv.sync_global();        // the buffer is free for the next message only when every lane has read this one
}
template <class C, class W>
MP_HD void fsq_absorb_points(W& wv, const FsDev& f, const FsqGeom& g, uint32_t wid, const uint32_t* P, uint32_t first, uint32_t count,
                             uint32_t first2, uint32_t count2, PerLane<B2sSeed>& seed) {
  fsq_absorb<C>(wv, f, g, wid, count + count2,
                [&](StageWriter& w, uint32_t b, uint32_t i) {
                  fs_put_point<C>(w, ld_aff<C>(P + p_off<C>(i < count ? first + i : first2 + (i - count), f.Bpad, b)));
                },
                0, [](StageWriter&, uint32_t) {}, seed);
}
template <class C, class W>
MP_HD void fsq_challenges(W& wv, const FsDev& f, const FsqGeom& g, uint32_t wid, uint32_t* S, const PerLane<B2sSeed>& seed, uint32_t slot0,
                          uint32_t slot1) {
  wv.lanes([&](uint32_t l) {
    const FsqLane q = fsq_lane(g, wid, l);
    if (q.live && q.sub == 0) fs_challenges<C>(seed[l].s, S, f.Bpad, q.b, slot0, slot1);
  });
}
template <class W>
MP_HD void fsq_store_seed(W& wv, const FsDev& f, const FsqGeom& g, uint32_t wid, const PerLane<B2sSeed>& seed) {
  wv.lanes([&](uint32_t l) {
    const FsqLane q = fsq_lane(g, wid, l);
    if (q.live && q.sub == 0) fs_store_seed(f, q.b, seed[l].s);
  });
}
template <class C, class W>
MP_HD void fsq_statement_and_x(W& wv, const FsStatementArgs& a, const FsqGeom& g, uint32_t wid, PerLane<B2sSeed>& seed) {
  wv.lanes([&](uint32_t l) {
#pragma unroll
    for (int i = 0; i < 8; ++i) seed[l].s[i] = a.init_seed[i];
  });
  fsq_absorb<C>(wv, a.f, g, wid, fs_statement_points(a),
                [&](StageWriter& w, uint32_t b, uint32_t i) { fs_put_statement_point<C>(w, a, b, i); }, 4,
                [&](StageWriter& w, uint32_t) { stage_word(w, a.m); stage_word(w, 0); stage_word(w, a.n); stage_word(w, 0); }, seed);
  fsq_absorb_points<C>(wv, a.f, g, wid, a.P, a.p_cA, a.m, 0, 0, seed);
  fsq_challenges<C>(wv, a.f, g, wid, a.S, seed, a.s_x, NO_SLOT);
}
struct FsqStatementArgs {
  FsStatementArgs st;
  FsqGeom g;
};
template <class C, class W>
MP_HD void body_fsq_round1(const FsqStatementArgs& a, uint32_t wid, W& wv) {
  PerLane<B2sSeed> seed;
  fsq_statement_and_x<C>(wv, a.st, a.g, wid, seed);
  fsq_store_seed(wv, a.st.f, a.g, wid, seed);
}
MP_WAVE_KERNEL(k_fsq_round1, FsqStatementArgs, body_fsq_round1)
struct FsqRoundArgs {
  FsRoundArgs r;
  FsqGeom g;
};
template <class C, class W>
MP_HD void body_fsq_round(const FsqRoundArgs& a, uint32_t wid, W& wv) {
  const FsDev& f = a.r.f;
  PerLane<B2sSeed> seed;
  wv.lanes([&](uint32_t l) {
    const FsqLane q = fsq_lane(a.g, wid, l);
    if (a.r.copy_from != NO_SLOT && q.live && q.sub == 0)
      st_aff<C>(a.r.P + p_off<C>(a.r.copy_to, f.Bpad, q.b), ld_aff<C>(a.r.P + p_off<C>(a.r.copy_from, f.Bpad, q.b)));
    fs_load_seed(f, q.b, seed[l].s);
  });
  if (a.r.copy_from != NO_SLOT) wv.sync_global();
  for (uint32_t s = 0; s < a.r.nsteps; ++s) {
    const FsStep st = a.r.step[s];
    fsq_absorb_points<C>(wv, f, a.g, wid, a.r.P, st.first, st.count, st.first2, st.count2, seed);
    if (st.slot0 != NO_SLOT) fsq_challenges<C>(wv, f, a.g, wid, a.r.S, seed, st.slot0, st.slot1);
  }
  fsq_store_seed(wv, f, a.g, wid, seed);
}
MP_WAVE_KERNEL(k_fsq_round, FsqRoundArgs, body_fsq_round)
struct FsqVerifyArgs {
  VerifyFsArgs v;
  FsqGeom g;
};
template <class C, class W>
MP_HD void body_fsq_verify(const FsqVerifyArgs& a, uint32_t wid, W& wv) {
  typedef typename C::FrP R;
  const VerifyLay& l = a.v.l;
  const uint32_t m = l.m;
  const FsDev& f = a.v.st.f;
  uint32_t* S = a.v.st.S;
  const uint32_t* P = a.v.st.P;
  const FsqGeom& g = a.g;
  PerLane<B2sSeed> seed;
  fsq_statement_and_x<C>(wv, a.v.st, g, wid, seed);
  fsq_absorb_points<C>(wv, f, g, wid, P, l.cB, m, 0, 0, seed);
  fsq_challenges<C>(wv, f, g, wid, S, seed, l.y, l.z);
  fsq_absorb_points<C>(wv, f, g, wid, P, l.cb, 1, 0, 0, seed);
  fsq_absorb_points<C>(wv, f, g, wid, P, l.hB, m, 0, 0, seed);
  fsq_challenges<C>(wv, f, g, wid, S, seed, l.hx, l.hy);
  fsq_absorb_points<C>(wv, f, g, wid, P, l.zcA0, 2 + 2 * m + 1, 0, 0, seed);
  fsq_challenges<C>(wv, f, g, wid, S, seed, l.zx, NO_SLOT);
  fsq_absorb_points<C>(wv, f, g, wid, P, l.svcd, 3, 0, 0, seed);
  fsq_challenges<C>(wv, f, g, wid, S, seed, l.svx, NO_SLOT);
  fsq_absorb_points<C>(wv, f, g, wid, P, l.mecA0, 1 + 2 * m + 4 * m, 0, 0, seed);
  fsq_challenges<C>(wv, f, g, wid, S, seed, l.mx, NO_SLOT);
  if (a.v.merge) {      // the weights of the merged equation (body_verify_fs): the 5n + 9 response scalars, 8 aligned words each
    const uint32_t nsc = 5 * l.n + 9;
    PerLane<const uint32_t*> base;
    wv.lanes([&](uint32_t ln) {
      const FsqLane q = fsq_lane(g, wid, ln);
      base[ln] = f.stage + q.b;
      if (!q.live) return;
      for (uint32_t i = q.sub; i < nsc; i += g.lpp) {
        uint32_t k[8];
        fe_to_canonical<R>(ld_fe<R>(S + s_off(l.zabar + i, f.Bpad, q.b)), k);
        StageWriter w = stage_begin_at(f.stage, f.Bpad, q.b, 8 * i);
#pragma unroll
        for (int t = 0; t < 8; ++t) stage_word(w, k[t]);
      }
      if (q.sub == 0) {
        StageWriter w = stage_begin_at(f.stage, f.Bpad, q.b, 8 * nsc);
#pragma unroll
        for (int t = 0; t < 8; ++t) stage_word(w, seed[ln].s[t]);
      }
    });
    wv.sync_global();
    blake2s_staged_quad(wv, base, f.Bpad, 32 * nsc + 32, seed);
    wv.lanes([&](uint32_t ln) {
      const FsqLane q = fsq_lane(g, wid, ln);
      if (!q.live || q.sub != 0) return;
      FrStream st;
      frstream_init(st, seed[ln].s);
      for (uint32_t k = 0; k < (uint32_t)VC_COUNT; ++k) st_fe<R>(S + s_off(l.mr + k, f.Bpad, q.b), frstream_next<R>(st));
      fs_store_seed(f, q.b, seed[ln].s);      // the transcript's last state: chain verification hashes it
    });
  }
}
MP_WAVE_KERNEL(k_fsq_verify, FsqVerifyArgs, body_fsq_verify)

// merged scalars: S[dst] = sum_{pairs} S[r] * S[coef]      (lane = (proof, merge job))
struct VerifyMergeArgs {
  uint32_t* S;
  const MergeJob* jobs;
  const MergePair* pairs;
  uint32_t Bpad;
};
template <class C>
MP_HD void body_verify_merge(const VerifyMergeArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  const MergeJob job = a.jobs[y];
  Fe<R> acc = fe_zero<R>();
  for (uint32_t i = 0; i < job.count; ++i) {
    const MergePair pr = a.pairs[job.begin + i];
    acc = fe_add<R>(acc, fe_mul<R>(ld_fe<R>(a.S + s_off(pr.r, a.Bpad, b)), ld_fe<R>(a.S + s_off(pr.coef, a.Bpad, b))));
  }
  st_fe<R>(a.S + s_off(job.dst, a.Bpad, b), acc);
}
MP_KERNEL(k_verify_merge, VerifyMergeArgs, body_verify_merge)

struct VerifyScalArgs {
  uint32_t* S;
  const uint32_t* P;
  uint32_t* direct;      // [2][Bpad] bit i set = direct check i failed (two lanes of k_verify_scal report, one word each)
  VerifyLay l;
  VCoefMap c;
  uint32_t Bpad;
};
// y = 0: the O(m) coefficients, the bilinear value and the direct checks on points (failure bits -> direct[b]);  y = 1: the product
// value prod_{i=1..N} (y i + x^i - z) with the deck coefficients x^i (the one chain of length N) and the scalar checks of the
// single-value product (-> direct[Bpad + b]);  y = 2 + j, j < n: entry j of the commitment-key coefficient vectors
template <class C>
MP_HD void body_verify_scal(const VerifyScalArgs& a, uint32_t b, uint32_t y_) {
  typedef typename C::FrP R;
  const VerifyLay& l = a.l;
  const VCoefMap& c = a.c;
  const uint32_t m = l.m, n = l.n, N = l.N;
  const Fe<R> one = fe_one<R>();
  if (y_ >= 2) {
    const uint32_t j = y_ - 2;
    MP_ST(c.za_ck + j, fe_neg<R>(MP_LD(l.zabar + j)));
    MP_ST(c.zb_ck + j, fe_neg<R>(MP_LD(l.zbbar + j)));
    const Fe<R> sx = MP_LD(l.svx), at = MP_LD(l.svat + j);
    MP_ST(c.sa_ck + j, fe_neg<R>(at));
    if (j + 1 < n)      // -(x bt_{j+1} - bt_j at_{j+1})
      MP_ST(c.sd_ck + j, fe_sub<R>(fe_mul<R>(MP_LD(l.svbt + j), MP_LD(l.svat + j + 1)), fe_mul<R>(sx, MP_LD(l.svbt + j + 1))));
    const Fe<R> ab = MP_LD(l.meabar + j), mx = MP_LD(l.mx);
    MP_ST(c.ma_ck + j, fe_neg<R>(ab));
    Fe<R> co = fe_neg<R>(one);                       // -mx^(m-i), i = m .. 1
    for (uint32_t i = m; i >= 1; --i) {
      MP_ST(c.me_c + (i - 1) * n + j, fe_mul<R>(co, ab));
      co = fe_mul<R>(co, mx);
    }
    return;
  }
  const Fe<R> x = MP_LD(l.x), y = MP_LD(l.y), z = MP_LD(l.z);
  if (y_ == 1) {
    const Fe<R> sx = MP_LD(l.svx);
    Fe<R> prod = one, xi = one, yi = fe_zero<R>();
    for (uint32_t i = 1; i <= N; ++i) {
      xi = fe_mul<R>(xi, x);
      yi = fe_add<R>(yi, y);
      prod = fe_mul<R>(prod, fe_sub<R>(fe_add<R>(yi, xi), z));
      MP_ST(c.em_x + i - 1, xi);                      // x^i: coefficient of deck[i-1] in Cx
    }
    uint32_t fail = 0;
    if (!fe_eq(MP_LD(l.svbt + 0), MP_LD(l.svat + 0)) || !fe_eq(MP_LD(l.svbt + n - 1), fe_mul<R>(sx, prod))) fail |= 1u << VC_SVP_SCALARS;
    a.direct[(size_t)a.Bpad + b] = fail;
    return;
  }
  uint32_t fail = 0;
  MP_ST(l.one, one);
  MP_ST(c.minus_one, fe_neg<R>(one));
  // --- Hadamard: first commitment, last commitment
  MP_ST(c.had_y, y);
  MP_ST(c.had_mz, fe_neg<R>(z));
  if (!aff_eq<C>(ld_aff<C>(a.P + p_off<C>(l.hB + m - 1, a.Bpad, b)), ld_aff<C>(a.P + p_off<C>(l.cb, a.Bpad, b)))) fail |= 1u << VC_HAD_BM;
  // --- zero argument
  {
    const Fe<R> hx = MP_LD(l.hx), hy = MP_LD(l.hy), zx = MP_LD(l.zx);
    if (!aff_is_inf<C>(ld_aff<C>(a.P + p_off<C>(l.zcD + m + 1, a.Bpad, b)))) fail |= 1u << VC_ZERO_DM1;
    const uint32_t t_zx = l.tmp, t_hx = l.tmp + 2 * m + 1;     // zx^0..zx^2m ; hx^0..hx^m
    Fe<R> acc = one;
    for (uint32_t i = 0; i <= 2 * m; ++i) {
      MP_ST(t_zx + i, acc);
      acc = fe_mul<R>(acc, zx);
    }
    acc = one;
    for (uint32_t i = 0; i <= m; ++i) {
      MP_ST(t_hx + i, acc);
      acc = fe_mul<R>(acc, hx);
    }
    // VC_ZERO_A
    Fe<R> gs = fe_zero<R>();
    for (uint32_t i = 1; i < m; ++i) {
      const Fe<R> zi = MP_LD(t_zx + i);
      MP_ST(c.za_cA + i, fe_mul<R>(zi, y));
      MP_ST(c.za_cB + i, zi);
      gs = fe_add<R>(gs, zi);
    }
    gs = fe_neg<R>(fe_add<R>(fe_mul<R>(gs, z), MP_LD(t_zx + m)));
    MP_ST(c.za_gsum, gs);
    MP_ST(c.za_H, fe_neg<R>(MP_LD(l.zrbar)));
    // VC_ZERO_B: hB[j] gets [j <= m-2] zx^(m-j) hx^(j+1) + [j >= 1] zx hx^j
    for (uint32_t j = 0; j < m; ++j) {
      Fe<R> co = fe_zero<R>();
      if (j + 2 <= m) co = fe_mul<R>(MP_LD(t_zx + m - j), MP_LD(t_hx + j + 1));
      if (j >= 1) co = fe_add<R>(co, fe_mul<R>(zx, MP_LD(t_hx + j)));
      MP_ST(c.zb_hB + j, co);
    }
    MP_ST(c.zb_H, fe_neg<R>(MP_LD(l.zsbar)));
    // VC_ZERO_D
    for (uint32_t k = 0; k <= 2 * m; ++k) MP_ST(c.zd_cD + k, MP_LD(t_zx + k));
    Fe<R> bil = fe_zero<R>(), yp = hy;
    for (uint32_t j = 0; j < n; ++j) {
      bil = fe_add<R>(bil, fe_mul<R>(fe_mul<R>(MP_LD(l.zabar + j), MP_LD(l.zbbar + j)), yp));
      yp = fe_mul<R>(yp, hy);
    }
    MP_ST(c.zd_ck0, fe_neg<R>(bil));
    MP_ST(c.zd_H, fe_neg<R>(MP_LD(l.ztbar)));
  }
  // --- single value product
  {
    const Fe<R> sx = MP_LD(l.svx);
    MP_ST(c.sa_x, sx);
    MP_ST(c.sa_H, fe_neg<R>(MP_LD(l.svrt)));
    MP_ST(c.sd_x, sx);
    MP_ST(c.sd_H, fe_neg<R>(MP_LD(l.svst)));
  }
  // --- multi-exponentiation
  {
    const Fe<R> mx = MP_LD(l.mx);
    if (!aff_is_inf<C>(ld_aff<C>(a.P + p_off<C>(l.mecB + m, a.Bpad, b)))) fail |= 1u << VC_ME_BM;
    Fe<R> acc = one;          // mx^0 .. mx^(2m-1)
    for (uint32_t k = 0; k < 2 * m; ++k) {
      MP_ST(c.mb_x + k, acc);
      MP_ST(c.me_x + k, acc);
      if (k <= m) MP_ST(c.ma_x + k, acc);
      acc = fe_mul<R>(acc, mx);
    }
    MP_ST(c.ma_H, fe_neg<R>(MP_LD(l.merbar)));
    MP_ST(c.mb_ck0, fe_neg<R>(MP_LD(l.mebbar)));
    MP_ST(c.mb_H, fe_neg<R>(MP_LD(l.mesbar)));
    const Fe<R> ntau = fe_neg<R>(MP_LD(l.metaubar));
    MP_ST(c.me_tauG, ntau);
    MP_ST(c.me_taupk, ntau);
    MP_ST(c.me_bgen, fe_neg<R>(MP_LD(l.mebbar)));
  }
  a.direct[b] = fail;
}
MP_KERNEL(k_verify_scal, VerifyScalArgs, body_verify_scal)

struct VerdictArgs {
  const uint32_t* J;
  const uint32_t* direct;
  int32_t* status;
  uint32_t Bpad, chk_first;
};
template <class C>
MP_HD void body_verdict(const VerdictArgs& a, uint32_t b, uint32_t y) {
  if (a.status[b] < 0) return;   // usage error already recorded
  const uint32_t direct = a.direct[b] | a.direct[(size_t)a.Bpad + b];
  int32_t code = 0;
  for (int cidx = 0; cidx < (int)VC_COUNT && code == 0; ++cidx) {
    bool failed;
    if (vcheck_is_msm(cidx))
      failed = !fe_is_zero(ld_fe<typename C::FqP>(a.J + j_off<C>(a.chk_first + cidx, a.Bpad, b) + 2 * Geo<C>::FW));
    else
      failed = (direct >> cidx) & 1u;
    if (failed) code = vcheck_code(cidx);
  }
  a.status[b] = code;
}
MP_KERNEL(k_verdict, VerdictArgs, body_verdict)

// merged verification: 0 = every equation holds (up to the 2^-250 chance that random weights hide an error) and every
// direct check passed; otherwise the proof is marked 1 and the batch flag is raised -- the engine then evaluates the
// equations one by one to name the first failing check [REF tests.rs:223-225 expects the name]
struct VerdictMergedArgs {
  const uint32_t* J;
  const uint32_t* direct;
  int32_t* status;
  uint32_t* flag;        // [1] raised when any proof of the batch needs the per-equation pass
  uint32_t Bpad, chk_merged;
};
template <class C>
MP_HD void body_verdict_merged(const VerdictMergedArgs& a, uint32_t b, uint32_t y) {
  if (a.status[b] < 0) return;   // usage error already recorded
  const bool bad = (a.direct[b] | a.direct[(size_t)a.Bpad + b]) != 0 ||
                   !fe_is_zero(ld_fe<typename C::FqP>(a.J + j_off<C>(a.chk_merged, a.Bpad, b) + 2 * Geo<C>::FW));
  a.status[b] = bad ? 1 : 0;
  if (bad) a.flag[0] = 1u;       // same value from every failing lane: no atomic needed
}
MP_KERNEL(k_verdict_merged, VerdictMergedArgs, body_verdict_merged)

// ---- chain verification: the L links of one card table's shuffle chain verified as ONE equation per table -------------------------
// [REF barnett-smart-card-protocol/examples/round.rs:268-350: every player shuffles the deck the previous player produced, and
// every shuffle is verified].  Lane layout: link j of table t is lane j T + t; its merged scalars (k_verify_merge) are already in S.
// (1) weights: rho_j = Fr::rand of ChaCha20(Blake2s(final transcript seeds of all L links of the table)) -- they depend on every
//     byte of every proof of the chain;
//     Round 6: in two levels, so that a group of 1 024 proofs does not hash 1 024 seeds and draw 1 024 weights on ONE lane (19 ms at 256
//     groups): k_chain_digest -- one lane per (table, block of CW_BLOCK = 64 links) -- hashes the block's seeds; k_chain_weights -- the
//     same lanes -- hashes the table's block digests into the table's key (every lane of a table does that: L / 128 compressions),
//     derives the block's own key = Blake2s(table key || block number) and draws the block's 64 weights from ChaCha20(block key).
//     A weight is still a function of every byte of every proof of its table and of nothing an adversary controls afterwards.
static const uint32_t CW_BLOCK = 64;
struct ChainWeightsArgs {
  const uint32_t* seed;    // [8][Bpad]: last transcript state per link (k_verify_fs)
  uint32_t* CW;            // [L][Tpad] Fr
  uint32_t* dig;           // [blocks][8][Tpad]: the blocks' digests
  uint32_t Bpad, Tpad, T, L;
};
// x = table, y = block
template <class C>
MP_HD void body_chain_digest(const ChainWeightsArgs& a, uint32_t t, uint32_t k) {
  Blake2sState st;
  blake2s_init(st);
  uint32_t m[16];
  const uint32_t j0 = k * CW_BLOCK, j1 = j0 + CW_BLOCK < a.L ? j0 + CW_BLOCK : a.L;
  for (uint32_t j = j0; j < j1; j += 2) {
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      m[w] = a.seed[(size_t)w * a.Bpad + (size_t)j * a.T + t];
      m[8 + w] = j + 1 < j1 ? a.seed[(size_t)w * a.Bpad + (size_t)(j + 1) * a.T + t] : 0u;
    }
    const bool last = j + 2 >= j1;
    blake2s_compress(st, m, last ? 32ull * (j1 - j0) : 32ull * (j + 2 - j0), last);
  }
#pragma unroll
  for (int w = 0; w < 8; ++w) a.dig[((size_t)k * 8 + w) * a.Tpad + t] = st.h[w];
}
MP_KERNEL(k_chain_digest, ChainWeightsArgs, body_chain_digest)
template <class C>
MP_HD void body_chain_weights(const ChainWeightsArgs& a, uint32_t t, uint32_t k) {
  typedef typename C::FrP R;
  const uint32_t nb = (a.L + CW_BLOCK - 1) / CW_BLOCK;
  Blake2sState st;
  blake2s_init(st);
  uint32_t m[16];
  for (uint32_t i = 0; i < nb; i += 2) {           // the table's key: the digests of all its blocks
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      m[w] = a.dig[((size_t)i * 8 + w) * a.Tpad + t];
      m[8 + w] = i + 1 < nb ? a.dig[((size_t)(i + 1) * 8 + w) * a.Tpad + t] : 0u;
    }
    const bool last = i + 2 >= nb;
    blake2s_compress(st, m, last ? 32ull * nb : 32ull * (i + 2), last);
  }
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    m[w] = st.h[w];
    m[8 + w] = 0u;
  }
  m[8] = k;                                        // the block's key: Blake2s(table key || block number)
  blake2s_init(st);
  blake2s_compress(st, m, 36ull, true);
  FrStream fs;
  frstream_init(fs, st.h);
  const uint32_t j0 = k * CW_BLOCK, j1 = j0 + CW_BLOCK < a.L ? j0 + CW_BLOCK : a.L;
  for (uint32_t j = j0; j < j1; ++j) st_fe<R>(a.CW + ((size_t)j * a.Tpad + t) * 8, frstream_next<R>(fs));
}
MP_KERNEL(k_chain_weights, ChainWeightsArgs, body_chain_weights)
// (2) scalars of the chain equation: CS[i][t] = sum_{k < cnt} rho_j S[s][lane(j,t)], j = j0 + k step  (+ rho_{j0-step} S[s2][lane(j0-step,t)]):
//     a deck between two links carries the scalar of its role as "shuffled deck" of the earlier link plus that as "deck" of the later.
//     step = 1: one table per equation.  step = G > 1 (round 5): the chains of G tables share one equation -- link j of member g is
//     "link" j G + g of the equation, so a chain's consecutive links lie G apart and the lane formula j T + t stays what it is.
struct ChainTerm {
  uint32_t s, j0, cnt, s2, step;
};
struct ChainScalArgs {
  const uint32_t* S;
  const uint32_t* CW;
  uint32_t* CS;            // [terms][Tpad] Fr
  const ChainTerm* terms;
  uint32_t Bpad, Tpad, T;
};
template <class C>
MP_HD void body_chain_scalars(const ChainScalArgs& a, uint32_t t, uint32_t y) {
  typedef typename C::FrP R;
  const ChainTerm ct = a.terms[y];
  Fe<R> acc = fe_zero<R>();
  for (uint32_t k = 0, j = ct.j0; k < ct.cnt; ++k, j += ct.step)
    acc = fe_add<R>(acc, fe_mul<R>(ld_fe<R>(a.CW + ((size_t)j * a.Tpad + t) * 8), ld_fe<R>(a.S + s_off(ct.s, a.Bpad, j * a.T + t))));
  if (ct.s2 != NO_SLOT)
    acc = fe_add<R>(acc, fe_mul<R>(ld_fe<R>(a.CW + ((size_t)(ct.j0 - ct.step) * a.Tpad + t) * 8),
                                   ld_fe<R>(a.S + s_off(ct.s2, a.Bpad, (ct.j0 - ct.step) * a.T + t))));
  st_fe<R>(a.CS + ((size_t)y * a.Tpad + t) * 8, acc);
}
MP_KERNEL(k_chain_scalars, ChainScalArgs, body_chain_scalars)
// (3) verdict per table: every link's direct checks passed, no encoding error, and the chain equation holds
struct ChainVerdictArgs {
  const uint32_t* J;
  const uint32_t* direct;
  int32_t* status;         // [L T] (lane order)
  uint32_t* flag;
  uint32_t* gbad;          // [T] (may be null): 1 = this table / group needs a closer look -- only ITS members are re-verified (round 5)
  uint32_t Bpad, T, L, j_final;
  // keyed chains: the chain equation carries ONE key term per table (link 0's key with the summed key scalars of all links), which
  // is only the sum of the per-link equations if every link of the table was given the same key.  A table whose links name
  // different keys takes the per-link path, where every link is checked against its own key.
  const uint32_t* P;
  uint32_t p_pk;           // NO_SLOT: not keyed
  uint32_t kstep;          // tables per equation: link j belongs to member j % kstep, whose first link is j % kstep (1: every link to link 0)
  // the CALLER's status words (lane order; may be null): 0 for every member of an equation that holds, MP_ERR_INTERNAL for the members of
  // one that does not -- until the finer passes have given each of them its own word.  A call that fails on its way there (out of memory
  // in a refinement) then leaves no word that reads "accepted" for a proof nobody cleared
  int32_t* out;
  uint32_t* part;          // [T] scratch, zero at launch: what the links of equation t found on their own
};
// Three launches since round 6 (an equation of 64 tables x 32 links is 2 048 links: one lane walking them took 10 ms per step):
// k_chain_check -- x = lane of (link, equation): the link's own findings, OR-ed into the equation's word part[t]; k_chain_verdict -- x =
// equation: the equation's value and part[t] give gbad[t] and the flag; k_chain_mark -- x = lane: the caller's status words.
template <class C>
MP_HD void body_chain_check(const ChainVerdictArgs& a, uint32_t x, uint32_t) {
  const uint32_t t = x % a.T, j = x / a.T;
  bool bad = a.status[x] != 0 || (a.direct[x] | a.direct[(size_t)a.Bpad + x]) != 0;
  if (a.p_pk != NO_SLOT && j >= a.kstep) {
    uint32_t k0[Geo<C>::PW], kj[Geo<C>::PW];
    ld_words<Geo<C>::PW>(a.P + p_off<C>(a.p_pk, a.Bpad, (j % a.kstep) * a.T + t), k0);
    ld_words<Geo<C>::PW>(a.P + p_off<C>(a.p_pk, a.Bpad, x), kj);
    uint32_t d = 0;
#pragma unroll
    for (uint32_t i = 0; i < Geo<C>::PW; ++i) d |= k0[i] ^ kj[i];
    bad = bad || d != 0;
  }
  if (bad) a.part[t] = 1u;       // (the same value from every lane that writes: no atomic needed)
}
MP_KERNEL(k_chain_check, ChainVerdictArgs, body_chain_check)
template <class C>
MP_HD void body_chain_verdict(const ChainVerdictArgs& a, uint32_t t, uint32_t y) {
  const bool bad = a.part[t] != 0 || !fe_is_zero(ld_fe<typename C::FqP>(a.J + j_off<C>(a.j_final, a.Bpad, t) + 2 * Geo<C>::FW));
  a.part[t] = bad ? 1u : 0u;
  if (a.gbad) a.gbad[t] = bad ? 1u : 0u;
  if (bad) a.flag[0] = 1u;
}
MP_KERNEL(k_chain_verdict, ChainVerdictArgs, body_chain_verdict)
template <class C>
MP_HD void body_chain_mark(const ChainVerdictArgs& a, uint32_t x, uint32_t) {
  a.out[x] = a.part[x % a.T] ? -5 : 0;
}
MP_KERNEL(k_chain_mark, ChainVerdictArgs, body_chain_mark)

// ---- re-verification of the proofs a screen could not clear (engine_core.hpp verify_subset, round 5): a failing screen -- the merged
// equation of one proof, the equation of a group, of a chain -- says WHICH proofs need a closer look; their inputs are gathered into a
// contiguous sub-batch, the sub-batch goes through the next finer pass, and its status words are scattered back.  Everybody else's
// verdict stands: what a rejected proof costs does not depend on how many honest proofs shared its batch
// [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:420-443: one call, one proof].
// ---- group verification: the points of a group equation gathered ONCE into a contiguous run per group (round 5).  In the P arena the
// 128 members of a group lie T lanes apart in each of 238 slots -- 30 464 pieces of 64 bytes scattered over 4 GB, each sharing its
// 128-byte line with a stranger; the bucket kernel read every one of them 26 times (once per window), in bucket order.  As ONE run of
// K x 64 bytes (2 MB) the lines are fully used, the run is a page or two, and the windows of the group find it in the L2 of their XCD.
// x = proof lane b = j T + t, y = P slot; term index in the run = j `per` + slot (the order the group plan lists its points in)
struct GroupTileArgs {
  const uint32_t* P;
  uint32_t* tile;          // [T][K][PW]
  uint32_t Bpad, T, per, K;
};
template <class C>
MP_HD void body_group_tile(const GroupTileArgs& a, uint32_t b, uint32_t y) {
  const uint32_t t = b % a.T, j = b / a.T;
  uint32_t w[Geo<C>::PW];
  ld_words<Geo<C>::PW>(a.P + p_off<C>(y, a.Bpad, b), w);
  st_words<Geo<C>::PW>(a.tile + ((size_t)t * a.K + (size_t)j * a.per + y) * Geo<C>::PW, w);
}
MP_KERNEL(k_group_tile, GroupTileArgs, body_group_tile)

// ... and the same for a chain equation (round 6): its distinct points -- the L + 1 decks of each member table, the proofs' points, the
// members' keys -- lie in the P arena at (slot, lane of the link that brought them); term i of the equation's plan names them as
// src[i] = slot | link << 20.  Copied once into the run [e][i], the sorted entries of the bucket kernels index the run, a chain equation
// may span more than the 1 022 "links" ten bits of a sorted entry could name (64 tables x 32 links), and its windows gather from one
// contiguous page instead of 4 424 x 64 pieces of the arena.  x = equation, y = term
struct ChainTileArgs {
  const uint32_t* P;
  uint32_t* tile;          // [T][K][PW]
  const uint32_t* src;     // [K]
  uint32_t Bpad, T, K;
};
template <class C>
MP_HD void body_chain_tile(const ChainTileArgs& a, uint32_t x, uint32_t) {
  const uint32_t e = x % a.T, y = x / a.T;       // (flat: a pass may hold a few hundred equations -- a workgroup per term would be half empty)
  const uint32_t s = a.src[y];
  uint32_t w[Geo<C>::PW];
  ld_words<Geo<C>::PW>(a.P + p_off<C>(s & 0xFFFFFu, a.Bpad, e + (s >> 20) * a.T), w);
  st_words<Geo<C>::PW>(a.tile + ((size_t)e * a.K + y) * Geo<C>::PW, w);
}
MP_KERNEL(k_chain_tile, ChainTileArgs, body_chain_tile)

struct GatherRowsArgs {
  const uint32_t* src;     // rows of `words` 32-bit words
  uint32_t* dst;           // dst row i = src row idx[i]
  const uint32_t* idx;
  uint32_t words;
};
template <class C>
MP_HD void body_gather_rows(const GatherRowsArgs& a, uint32_t x, uint32_t y) {
  const uint32_t row = x / a.words, c = x % a.words;
  a.dst[(size_t)row * a.words + c] = a.src[(size_t)a.idx[row] * a.words + c];
}
MP_KERNEL(k_gather_rows, GatherRowsArgs, body_gather_rows)
struct ScatterStatusArgs {
  const int32_t* src;
  int32_t* dst;            // dst[idx[i]] = src[i]
  const uint32_t* idx;
};
template <class C>
MP_HD void body_scatter_status(const ScatterStatusArgs& a, uint32_t i, uint32_t y) {
  a.dst[a.idx[i]] = a.src[i];
}
MP_KERNEL(k_scatter_status, ScatterStatusArgs, body_scatter_status)

#undef MP_LD
#undef MP_ST

}  // namespace mp
