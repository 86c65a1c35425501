// Group-arithmetic kernels of the shuffle engine (gfx950).  One lane = one (proof, job): the 64 lanes of
// a wave run the SAME job of 64 consecutive proofs, so control flow is wave-uniform (only the window
// digits differ per lane) and every arena access is a contiguous run (slot-major arenas, layout.hpp).
//
//   body_fixed_msm   sum_t k_t * B_t over bases shared by the whole batch (commit key, G, pk, gen):
//                    8/16/20-bit windows over precomputed tables -> 32/16/13 mixed additions per term, no doublings.
//                    Replaces the Pedersen commits / ElGamal encrypt scalar-muls inside the reference's
//                    prover and verifier [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:409-415,437-442]
//   body_remask      out[i] = deck[pi(i)] + rho_i * (G, pk)          [REF mod.rs:388-395, remasking.rs:16-18]
//   body_var_msm     sum_t k_t * P_t over per-proof bases (the decks, proof elements): Straus interleaving
//                    with signed 5-bit windows over per-proof 16-entry affine tables (body_table; ark-ec's
//                    VariableBaseMSM bucket method is hopeless at 26..52 terms on a SIMT machine:
//                    SURVEY.md App. D), doubling chain shared by the terms of a job.  XYZZ accumulators.
//   body_table / body_recode / body_combine / body_normalize: their supporting passes.
//   body_key_windows keyed batches: window bases 2^(5w) pk of a per-proof aggregate key (their tables: body_table).
#pragma once
#include "curve.hpp"
#include "layout.hpp"

namespace mp {

// ---- arena access (16-byte vector loads/stores; arenas are 256-byte aligned) ----------------------------------
// Sizes in 32-bit words of the arena elements of curve C: a base-field element is FW packed words (8; 12 on BLS12-377),
// an affine point 2 FW, a Jacobian point 3 FW; a scalar (Fr) is 8 words on every supported curve.
template <class C>
struct Geo {
  // waves per SIMD the group-arithmetic kernels are compiled for: 4 (128 VGPRs) on the 256-bit curves; the 14-limb base field of
  // BLS12-377 needs ~200 registers for an XYZZ accumulator plus the temporaries of one addition, i.e. 2 waves -- asking for 4
  // there only made the compiler spill and warn
  static constexpr int OCC4 = C::FqP::NW > 8 ? 2 : 4;
  static constexpr int OCC3 = C::FqP::NW > 8 ? 2 : 3;
#ifdef MP_OCC_VAR      // experiment hook (tools/ab_build.py): waves per SIMD k_var_msm is compiled for
  static constexpr int OCC_VAR = MP_OCC_VAR;
#else
  static constexpr int OCC_VAR = OCC4;
#endif
  // k_table: with the division-step inversion inlined the 29-bit fields need 144 registers; capped at 128 the compiler spills 15-20
  // of them and the fourth wave still wins (-1.7 % in an A/B)
  static constexpr int OCC_TABLE = C::FqP::NW > 8 ? 2 : (C::FqP::L29 ? 4 : 3);
  static constexpr uint32_t FW = C::FqP::NW;
  static constexpr uint32_t PW = 2 * FW;
  static constexpr uint32_t JW = 3 * FW;
  static constexpr uint32_t FB = 4 * FW;   // wire bytes of a coordinate
  static constexpr uint32_t PB = 8 * FW;   // wire bytes of a point (x || y)
};
template <int N>
MP_HD void ld_words(const uint32_t* p, uint32_t* v) {
  static_assert(N % 4 == 0, "elements are whole 16-byte vectors");
  const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    const uint4 t = q[i];
    v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
  }
}
template <int N>
MP_HD void st_words(uint32_t* p, const uint32_t* v) {
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < N / 4; ++i) q[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}
// memory format of a field element = F::NW packed words (canonical Montgomery residue, field.hpp)
template <class F>
MP_HD Fe<F> ld_fe(const uint32_t* p) {
  uint32_t w[F::NW];
  ld_words<F::NW>(p, w);
  return fe_unpack<F>(w);
}
template <class F>
MP_HD void st_fe(uint32_t* p, const Fe<F>& a) {
  uint32_t w[F::NW];
  fe_pack<F>(a, w);
  st_words<F::NW>(p, w);
}
// A lazily reduced element stored WITHOUT canonicalising it (pack only: 16 instructions instead of 81).  Allowed where every reader
// unpacks it into the lazy invariant again and nobody compares or hashes the words: the window tables and the prefix-product scratch
// of k_table.  The value must fit the packed words: < 4p < 2^254 on the sparse STARK prime, < 2p on the dense ones; the pseudo-
// Mersenne form (4p ~ 2^258 > 2^256) keeps the canonical store.
template <class F>
MP_HD void st_fe_lazy(uint32_t* p, const Fe<F>& a) {
  if constexpr (F::L29 && !F::PM29) {
    uint32_t w[F::NW];
    pack29<F::NW, F::NL29>(a.v, w);
    st_words<F::NW>(p, w);
  } else {
    st_fe<F>(p, a);
  }
}
template <class C>
MP_HD void st_aff_lazy(uint32_t* p, const Aff<C>& a) {
  st_fe_lazy<typename C::FqP>(p, a.x);
  st_fe_lazy<typename C::FqP>(p + Geo<C>::FW, a.y);
}
template <class C>
MP_HD Aff<C> ld_aff(const uint32_t* p) {
  Aff<C> a;
  a.x = ld_fe<typename C::FqP>(p);
  a.y = ld_fe<typename C::FqP>(p + Geo<C>::FW);
  return a;
}
template <class C>
MP_HD void st_aff(uint32_t* p, const Aff<C>& a) {
  st_fe<typename C::FqP>(p, a.x);
  st_fe<typename C::FqP>(p + Geo<C>::FW, a.y);
}
template <class C>
MP_HD Jac<C> ld_jac(const uint32_t* p) {
  Jac<C> j;
  j.X = ld_fe<typename C::FqP>(p);
  j.Y = ld_fe<typename C::FqP>(p + Geo<C>::FW);
  j.Z = ld_fe<typename C::FqP>(p + 2 * Geo<C>::FW);
  return j;
}
template <class C>
MP_HD void st_jac(uint32_t* p, const Jac<C>& j) {
  st_fe<typename C::FqP>(p, j.X);
  st_fe<typename C::FqP>(p + Geo<C>::FW, j.Y);
  st_fe<typename C::FqP>(p + 2 * Geo<C>::FW, j.Z);
}

MP_HD size_t s_off(uint32_t slot, uint32_t Bpad, uint32_t b) { return ((size_t)slot * Bpad + b) * 8; }
template <class C>   // scratch arenas of base-field elements (prefix products of the batched inversions)
MP_HD size_t f_off(uint32_t slot, uint32_t Bpad, uint32_t b) { return ((size_t)slot * Bpad + b) * Geo<C>::FW; }
template <class C>
MP_HD size_t p_off(uint32_t slot, uint32_t Bpad, uint32_t b) { return ((size_t)slot * Bpad + b) * Geo<C>::PW; }
template <class C>
MP_HD size_t j_off(uint32_t slot, uint32_t Bpad, uint32_t b) { return ((size_t)slot * Bpad + b) * Geo<C>::JW; }

// ---- fixed-base MSM ---------------------------------------------------------------------------------
// geometry of the fixed-base tables of a table context: `bits`-wide unsigned windows (8, 16, 20 or 21)
struct FbGeom {
  uint32_t bits, windows, entries;   // windows = ceil(scalar bits / bits), entries = 2^bits - 1
};
struct FixedArgs {
  const uint32_t* S;
  uint32_t* J;
  const uint32_t* FB;   // [base][window][entry(1..entries)] affine, Geo<C>::PW words each
  const Job* jobs;
  const Term* terms;
  uint32_t Bpad;
  FbGeom g;
  uint32_t Sbpad;       // lane stride of the scalar array (= Bpad except for the compact scalars of chain verification)
};
template <class C>
MP_HD const uint32_t* fb_entry(const uint32_t* FB, const FbGeom& g, uint32_t base, uint32_t w, uint32_t d) {
  return FB + (((size_t)base * g.windows + w) * g.entries + (d - 1)) * Geo<C>::PW;
}
// window w of the canonical scalar k (bits <= 24; a window may straddle two words)
MP_HD uint32_t fb_digit(const uint32_t k[8], const FbGeom& g, uint32_t w) {
  const uint32_t bit = w * g.bits, word = bit >> 5, off = bit & 31;
  uint32_t v = k[word] >> off;
  if (off + g.bits > 32 && word < 7) v |= k[word + 1] << (32 - off);
  return v & ((1u << g.bits) - 1u);
}
template <class C>
MP_HD void body_fixed_msm(const FixedArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  const Job job = a.jobs[y];
  Xyzz<C> acc = xyzz_inf<C>();     // no doublings here: every operation is the 8M+2S mixed addition
  for (uint32_t t = 0; t < job.count; ++t) {
    const Term term = a.terms[job.begin + t];
    uint32_t k[8];
    fe_to_canonical<R>(ld_fe<R>(a.S + s_off(term.s, a.Sbpad, b)), k);
#pragma unroll 1
    for (uint32_t w = 0; w < a.g.windows; ++w) {
      const uint32_t d = fb_digit(k, a.g, w);
      if (d) xyzz_madd_ip<C>(acc, ld_aff<C>(fb_entry<C>(a.FB, a.g, term.b, w, d)));
    }
  }
  st_jac<C>(a.J + j_off<C>(job.out, a.Bpad, b), xyzz_to_jac<C>(acc));
}
MP_KERNEL_OCC(k_fixed_msm, FixedArgs, body_fixed_msm, Geo<C>::OCC4)

// ---- re-encryption (remask) -------------------------------------------------------------------------
struct RemaskArgs {
  const uint32_t* S;
  const uint32_t* P;
  uint32_t* J;
  const uint32_t* FB;
  const uint32_t* perm;   // [B][N] (caller's layout), may be null = identity
  uint32_t Bpad, N;
  uint32_t s_rho, p_deck, j_out;
  uint32_t base_G, base_pk;
  FbGeom g;
  // keyed batches: the aggregate key is per proof -- its multiples come from the proof's own window tables
  // (entries 1..16 of 2^(5w) pk, built by k_key_windows + k_table) and the signed digits of rho
  uint32_t keyed;
  const int8_t* D;      // [dslot][window][Bpad]
  const uint32_t* T;    // [tslot][entry][Bpad]
  uint32_t d_first, t_first, nwin;   // digit slot of rho_0, table slot of window 0
  // keyed == 2: the proof's key is member kidx[b] of a key set (mp_keyset_create): fixed-base tables of every key of the
  // set, geometry kg -- the keys of the card tables a server runs are known long before the shuffles are
  const uint32_t* KFB;  // [key][window][entry] affine
  const uint32_t* kidx; // [B]
  FbGeom kg;
  uint32_t nkeys;
};
// y = 2*i + component
template <class C>
MP_HD void body_remask(const RemaskArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  const uint32_t i = y >> 1, comp = y & 1;
  uint32_t src = a.perm ? a.perm[(size_t)b * a.N + i] : i;
  if (src >= a.N) src = 0;  // an invalid permutation is reported through the status word; stay in bounds
  uint32_t k[8];
  fe_to_canonical<R>(ld_fe<R>(a.S + s_off(a.s_rho + i, a.Bpad, b)), k);
  const uint32_t base = comp ? a.base_pk : a.base_G;
  Xyzz<C> acc = xyzz_inf<C>();
  if (a.keyed == 2 && comp) {
    uint32_t key = a.kidx[b];
    if (key >= a.nkeys) key = 0;      // reported through the status word (k_gather_keys)
#pragma unroll 1
    for (uint32_t w = 0; w < a.kg.windows; ++w) {
      const uint32_t d = fb_digit(k, a.kg, w);
      if (d) xyzz_madd_ip<C>(acc, ld_aff<C>(fb_entry<C>(a.KFB, a.kg, key, w, d)));
    }
    xyzz_madd_ip<C>(acc, ld_aff<C>(a.P + p_off<C>(a.p_deck + 2 * src + comp, a.Bpad, b)));
    st_jac<C>(a.J + j_off<C>(a.j_out + y, a.Bpad, b), xyzz_to_jac<C>(acc));
    return;
  }
  if (a.keyed && comp) {
#pragma unroll 1
    for (uint32_t w = 0; w < a.nwin; ++w) {
      const int d = a.D[((size_t)(a.d_first + i) * a.nwin + w) * a.Bpad + b];
      if (d != 0) {
        const uint32_t e = (uint32_t)(d < 0 ? -d : d) - 1;
        const Aff<C> q = ld_aff<C>(a.T + p_off<C>((a.t_first + w) * VB_ENTRIES + e, a.Bpad, b));
        xyzz_madd_signed_ip<C>(acc, q, d < 0);
      }
    }
    xyzz_madd_ip<C>(acc, ld_aff<C>(a.P + p_off<C>(a.p_deck + 2 * src + comp, a.Bpad, b)));
    st_jac<C>(a.J + j_off<C>(a.j_out + y, a.Bpad, b), xyzz_to_jac<C>(acc));
    return;
  }
#pragma unroll 1
  for (uint32_t w = 0; w < a.g.windows; ++w) {
    const uint32_t d = fb_digit(k, a.g, w);
    if (d) xyzz_madd_ip<C>(acc, ld_aff<C>(fb_entry<C>(a.FB, a.g, base, w, d)));
  }
  xyzz_madd_ip<C>(acc, ld_aff<C>(a.P + p_off<C>(a.p_deck + 2 * src + comp, a.Bpad, b)));
  st_jac<C>(a.J + j_off<C>(a.j_out + y, a.Bpad, b), xyzz_to_jac<C>(acc));
}
MP_KERNEL_OCC(k_remask, RemaskArgs, body_remask, Geo<C>::OCC4)

// ---- keyed batches: the products tau_k pk of the multi-exponentiation diagonals E_k (round 5) ---------------------------------
// A keyed proof's aggregate key is a per-proof point.  Its 2m products tau_k pk used to be one-term Straus jobs -- a 16-entry table and
// 250 doublings each for 51 additions, + 7 % on k_var_msm of every keyed batch (chain32, --keyed).  The key's multiples are there
// already: the proof's own window tables (entries 1 .. 16 of 2^(5w) pk, built for the re-encryption) or the key set's fixed-base
// tables.  y = k; the sum goes to J slot j_first + k, which the plan adds to E_k as a finished partial sum (layout.hpp, jkey).
struct KeyTermsArgs {
  const uint32_t* S;
  uint32_t* J;
  uint32_t Bpad, s_first, j_first;
  uint32_t keyed;        // 1: the proof's own window tables T[t_first + w]; 2: the key set's tables KFB[kidx[b]]
  const uint32_t* T;
  uint32_t t_first, nwin;
  const uint32_t* KFB;
  const uint32_t* kidx;
  FbGeom kg;
  uint32_t nkeys;
};
template <class C>
MP_HD void body_key_terms(const KeyTermsArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  uint32_t k[8];
  fe_to_canonical<R>(ld_fe<R>(a.S + s_off(a.s_first + y, a.Bpad, b)), k);
  Xyzz<C> acc = xyzz_inf<C>();
  if (a.keyed == 2) {
    uint32_t key = a.kidx[b];
    if (key >= a.nkeys) key = 0;      // reported through the status word (k_gather_keys)
#pragma unroll 1
    for (uint32_t w = 0; w < a.kg.windows; ++w) {
      const uint32_t d = fb_digit(k, a.kg, w);
      if (d) xyzz_madd_ip<C>(acc, ld_aff<C>(fb_entry<C>(a.KFB, a.kg, key, w, d)));
    }
  } else {
    // signed 5-bit digits on the fly: v = window + carry in [0, 32]; v > 16 becomes v - 32 with a carry into the next window
    uint32_t carry = 0;
#pragma unroll 1
    for (uint32_t w = 0; w < a.nwin; ++w) {
      const uint32_t bit = w * VB_WINDOW_BITS, word = bit >> 5, off = bit & 31;
      uint32_t v = word < 8 ? k[word] >> off : 0u;
      if (off + VB_WINDOW_BITS > 32 && word < 7) v |= k[word + 1] << (32 - off);
      v = (v & ((1u << VB_WINDOW_BITS) - 1u)) + carry;
      carry = v > (1u << (VB_WINDOW_BITS - 1)) ? 1u : 0u;
      const int d = carry ? (int)v - (1 << VB_WINDOW_BITS) : (int)v;
      if (d != 0) {
        const uint32_t e = (uint32_t)(d < 0 ? -d : d) - 1;
        xyzz_madd_signed_ip<C>(acc, ld_aff<C>(a.T + p_off<C>((a.t_first + w) * VB_ENTRIES + e, a.Bpad, b)), d < 0);
      }
    }
  }
  st_jac<C>(a.J + j_off<C>(a.j_first + y, a.Bpad, b), xyzz_to_jac<C>(acc));
}
MP_KERNEL_OCC(k_key_terms, KeyTermsArgs, body_key_terms, Geo<C>::OCC4)

// ---- key sets: the wire bytes of key kidx[b] for proof b (x = proof, y = 32-bit word of the point)
struct GatherKeysArgs {
  const uint32_t* wire;   // [nkeys][PB / 4]
  const uint32_t* kidx;   // [B]
  uint32_t* out;          // [B][PB / 4]
  int32_t* status;
  uint32_t nkeys;
};
template <class C>
MP_HD void body_gather_keys(const GatherKeysArgs& a, uint32_t b, uint32_t y) {
  constexpr uint32_t W = Geo<C>::PB / 4;
  uint32_t key = a.kidx[b];
  if (key >= a.nkeys) {
    if (y == 0) a.status[b] = -3;      // MP_ERR_BAD_ARGUMENT: no such key in the set
    key = 0;
  }
  a.out[(size_t)b * W + y] = a.wire[(size_t)key * W + y];
}
MP_KERNEL(k_gather_keys, GatherKeysArgs, body_gather_keys)

// ---- window bases of a per-proof key: W_w = 2^(5w) * pk, w < nwin (lane = proof; Jacobian out -> normalise -> k_table)
struct KeyWinArgs {
  const uint32_t* P;
  uint32_t* J;
  uint32_t Bpad, p_pk, j_first, nwin;
};
template <class C>
MP_HD void body_key_windows(const KeyWinArgs& a, uint32_t b, uint32_t y) {
  Jac<C> acc = jac_from_aff<C>(ld_aff<C>(a.P + p_off<C>(a.p_pk, a.Bpad, b)));
  st_jac<C>(a.J + j_off<C>(a.j_first, a.Bpad, b), acc);
#pragma unroll 1
  for (uint32_t w = 1; w < a.nwin; ++w) {
#pragma unroll 1
    for (int q = 0; q < VB_WINDOW_BITS; ++q) jac_dbl_ip<C>(acc);
    st_jac<C>(a.J + j_off<C>(a.j_first + w, a.Bpad, b), acc);
  }
}
MP_KERNEL(k_key_windows, KeyWinArgs, body_key_windows)

// ---- signed-window recoding of the variable-base scalars -------------------------------------------
struct RecodeArgs {
  const uint32_t* S;
  int8_t* D;            // [dslot][window][Bpad]
  const Term* list;     // {S slot, digit slot}
  uint32_t Bpad, nwin;
};
template <class C>
MP_HD void body_recode(const RecodeArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FrP R;
  const Term t = a.list[y];
  uint32_t k[9];
  fe_to_canonical<R>(ld_fe<R>(a.S + s_off(t.s, a.Bpad, b)), k);
  k[8] = 0;
  int carry = 0;
  for (uint32_t w = 0; w < a.nwin; ++w) {
    const uint32_t bit = w * VB_WINDOW_BITS;
    const uint32_t lo = k[bit >> 5] >> (bit & 31);
    const uint32_t hi = (bit & 31) > 27 ? k[(bit >> 5) + 1] << (32 - (bit & 31)) : 0u;
    int d = (int)((lo | hi) & 31u) + carry;
    carry = 0;
    if (d > 16) {
      d -= 32;
      carry = 1;
    }
    a.D[((size_t)t.b * a.nwin + w) * a.Bpad + b] = (int8_t)d;
  }
}
MP_KERNEL(k_recode, RecodeArgs, body_recode)

// ---- per-proof window tables: multiples 1P..16P of every variable base, AFFINE, built with batched affine
// additions.  One lane owns a group of up to TABLE_GROUP bases of one proof.  The multiples are produced in four
// rounds -- {2P}, {3P,4P}, {5P..8P}, {9P..16P} -- from the PREVIOUS round's outputs only:
//      even t:  tP = 2 * (t/2)P                      odd t:  tP = aP + (a+1)P,  a = (t-1)/2
// and ALL slopes of a round (up to 8 x TABLE_GROUP) share ONE Fermat inversion (Montgomery's trick; prefix products
// through HBM scratch).  ~5M + 1S per entry plus 4/15 of 1/64 of an inversion, against 7M+4S (Jacobian chain) + ~13M
// (normalisation).
// The kernel is HBM-bound, so the passes are arranged to touch memory as little as possible:
//   * the denominators of round r+1 (2 y_s, x_{s+1} - x_s) are functions of round r's outputs: their running product is
//     accumulated WHILE round r writes those outputs (no separate pass that re-reads them);
//   * a round's operands are consecutive entries of the previous round, consumed in order: each is loaded once and slides
//     through two registers (lo, hi) instead of being re-read per target;
//   * Montgomery's trick needs the second pass in the exact reverse order of the products, so the rounds alternate
//     direction (bases and targets descending, then ascending, ...) -- which is also the order the sliding window wants.
// Per base: 128 B copy + 15 x (32 + 32) B prefix products + ~15 x 64 B operand reads + 15 x 64 B results
// = ~3.0 KB instead of 4.2 KB with a separate denominator pass and per-target operand loads.
static const uint32_t TABLE_GROUP = 64;
struct TableArgs {
  const uint32_t* P;
  uint32_t* T;           // [tslot][entry][Bpad] affine
  uint32_t* scratch;     // [tslot][8][Bpad] field elements (prefix products)
  const Term* list;      // {P slot, table slot}; table slots are 0..n_tables-1 in list order
  uint32_t Bpad, n_tables, group;   // group = bases per lane (<= TABLE_GROUP)
};
// Scratch slot (of the 8 per table) of the prefix product for target t of the round with base e0.  A round reads its own
// slots while it writes the next round's: with this placement (round 2 in the upper half) every slot is read before the
// next round overwrites it, in both directions -- no second buffer.
MP_HD uint32_t table_key(uint32_t e0, uint32_t t) { return (e0 == 2 ? 4u : 0u) + (t - e0 - 1); }
// one prefix-product step of the NEXT round's batch: remember the product so far for target t (multiple t of table ts,
// round base e0n: targets e0n+1 .. 2 e0n), then fold its denominator in (a zero denominator -- P = infinity on a
// prime-order group -- is skipped here and in the consuming pass alike)
template <class C>
MP_HD void table_emit(const TableArgs& a, uint32_t b, uint32_t ts, uint32_t e0n, uint32_t t, Fe<typename C::FqP>& prod,
                      const Fe<typename C::FqP>& den) {
  typedef typename C::FqP F;
  st_fe_lazy<F>(a.scratch + f_off<C>(ts * 8 + table_key(e0n, t), a.Bpad, b), prod);      // (a product: < 2p)
  if (!fe_is_zero(den)) prod = fe_mul<F>(prod, den);
}
// target t of the current round from its operands (doubling of lo, or lo + hi); consumes one step of the running inverse
template <class C>
MP_HD Aff<C> table_step(const TableArgs& a, uint32_t b, uint32_t ts, uint32_t e0, uint32_t t, Fe<typename C::FqP>& inv,
                        const Aff<C>& lo, const Aff<C>& hi, bool dbl) {
  typedef typename C::FqP F;
  const Fe<F> den = dbl ? fe_dbl<F>(lo.y) : fe_sub_wide<F>(hi.x, lo.x);      // only multiplied and tested for zero (entries are < 2p here: loaded canonical or lazily stored products' differences)
  Aff<C> out = aff_inf<C>();
  if (!fe_is_zero(den)) {
    const Fe<F> dinv = fe_mul<F>(inv, ld_fe<F>(a.scratch + f_off<C>(ts * 8 + table_key(e0, t), a.Bpad, b)));
    inv = fe_mul<F>(inv, den);
    Fe<F> num;
    if (dbl) {
      const Fe<F> xx = fe_sqr<F>(lo.x);
      num = C::A == 1 ? fe_triple_add<F>(xx, fe_one<F>()) : fe_add<F>(fe_dbl<F>(xx), xx);      // 3 x^2 + a, one carry pass (R mod p < p)
    } else {
      num = fe_sub<F>(hi.y, lo.y);
    }
    const Fe<F> lam = fe_mul<F>(num, dinv);
    const Fe<F>& x2 = dbl ? lo.x : hi.x;
    out.x = fe_sub<F>(fe_sub<F>(fe_sqr<F>(lam), lo.x), x2);
    out.y = fe_sub<F>(fe_mul<F>(lam, fe_sub_lazy<F>(lo.x, out.x)), lo.y);      // the inner difference only feeds the product
  }
  st_aff_lazy<C>(a.T + p_off<C>(ts * VB_ENTRIES + t - 1, a.Bpad, b), out);
  return out;
}
template <class C>
MP_HD void body_table(const TableArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FqP F;
  const uint32_t g0 = y * a.group;
  const uint32_t g1 = g0 + a.group < a.n_tables ? g0 + a.group : a.n_tables;
  // entry 1 = P; denominators of round 1 (target 2 = 2P): bases ascending
  Fe<F> prod = fe_one<F>();
  for (uint32_t g = g0; g < g1; ++g) {
    const Term t = a.list[g];
    const Aff<C> p1 = ld_aff<C>(a.P + p_off<C>(t.s, a.Bpad, b));
    st_aff<C>(a.T + p_off<C>(t.b * VB_ENTRIES, a.Bpad, b), p1);
    table_emit<C>(a, b, t.b, 1, 2, prod, fe_dbl<F>(p1.y));
  }
  bool desc = true;     // direction of this round = reverse of the order its denominators were multiplied in
#pragma unroll 1
  for (uint32_t e0 = 1; e0 < (uint32_t)VB_ENTRIES; e0 *= 2, desc = !desc) {   // targets: multiples e0+1 .. 2 e0
    Fe<F> inv = fe_inv<F>(prod);
    prod = fe_one<F>();
    const bool emit = 2 * e0 < (uint32_t)VB_ENTRIES;
    const uint32_t e0n = 2 * e0;
    for (uint32_t gi = 0; gi < g1 - g0; ++gi) {
      const uint32_t g = desc ? g1 - 1 - gi : g0 + gi;
      const uint32_t ts = a.list[g].b;
      const uint32_t* tb = a.T + p_off<C>(ts * VB_ENTRIES, a.Bpad, b);
      const size_t estride = (size_t)a.Bpad * Geo<C>::PW;          // words between consecutive entries of a table
      if (e0 == 1) {            // single target 2 = 2 * 1; next round: 4 = 2*2, 3 = 1 + 2 (descending emission)
        const Aff<C> lo = ld_aff<C>(tb);
        const Aff<C> o2 = table_step<C>(a, b, ts, 1, 2, inv, lo, lo, true);
        table_emit<C>(a, b, ts, e0n, 4, prod, fe_dbl<F>(o2.y));
        table_emit<C>(a, b, ts, e0n, 3, prod, fe_sub<F>(o2.x, lo.x));
        continue;
      }
      const uint32_t h = e0 / 2;                                    // operands: multiples h .. e0 (entries h-1 .. e0-1)
      if (desc) {
        // t = 2e0, 2e0-1, ..., e0+1 : dbl(e0), add(e0-1,e0), dbl(e0-1), ..., add(h, h+1)
        Aff<C> lo = ld_aff<C>(tb + (size_t)(e0 - 1) * estride), hi = lo;
        Fe<F> prevx = lo.x;                                         // x of the previous output (t + 1)
#pragma unroll 1
        for (uint32_t t = 2 * e0; t > e0; --t) {
          const bool dbl = (t & 1u) == 0;
          if (!dbl) {                                               // operands (t-1)/2, (t+1)/2: slide down by one
            hi = lo;
            lo = ld_aff<C>(tb + (size_t)((t - 1) / 2 - 1) * estride);
          }
          const Aff<C> out = table_step<C>(a, b, ts, e0, t, inv, lo, hi, dbl);
          if (emit) {
            if (t < 2 * e0) table_emit<C>(a, b, ts, e0n, 2 * t + 1, prod, fe_sub<F>(prevx, out.x));   // (t) + (t+1)
            table_emit<C>(a, b, ts, e0n, 2 * t, prod, fe_dbl<F>(out.y));
            prevx = out.x;
          }
        }
        if (emit)                                                   // (e0) + (e0+1); x of multiple e0 is re-read (32 B)
          table_emit<C>(a, b, ts, e0n, 2 * e0 + 1, prod, fe_sub<F>(prevx, ld_fe<F>(tb + (size_t)(e0 - 1) * estride)));
      } else {
        // t = e0+1, ..., 2e0 : add(h, h+1), dbl(h+1), add(h+1, h+2), ..., dbl(e0)
        Aff<C> lo = ld_aff<C>(tb + (size_t)(h - 1) * estride), hi = lo;
        Fe<F> prevx = lo.x;
        if (emit) prevx = ld_fe<F>(tb + (size_t)(e0 - 1) * estride);    // x of multiple e0: first operand of (e0)+(e0+1)
#pragma unroll 1
        for (uint32_t t = e0 + 1; t <= 2 * e0; ++t) {
          const bool dbl = (t & 1u) == 0;
          if (!dbl) {                                               // operands a = (t-1)/2 (current hi) and a+1 (loaded)
            if (t != e0 + 1) lo = hi;
            hi = ld_aff<C>(tb + (size_t)((t + 1) / 2 - 1) * estride);
          } else {
            lo = hi;                                                // 2 * (t/2): the operand loaded last
          }
          const Aff<C> out = table_step<C>(a, b, ts, e0, t, inv, lo, hi, dbl);
          if (emit) {
            table_emit<C>(a, b, ts, e0n, 2 * t - 1, prod, fe_sub<F>(out.x, prevx));                    // (t-1) + (t)
            table_emit<C>(a, b, ts, e0n, 2 * t, prod, fe_dbl<F>(out.y));
            prevx = out.x;
          }
        }
      }
    }
  }
}
MP_KERNEL_OCC(k_table, TableArgs, body_table, Geo<C>::OCC_TABLE)

// ---- variable-base MSM (Straus) -----------------------------------------------------------------------
struct VarArgs {
  const int8_t* D;
  const uint32_t* T;     // [tslot][entry][Bpad] affine
  uint32_t* J;
  const Job* jobs;
  const Term* terms;     // {digit slot, table slot}
  uint32_t Bpad, nwin;
  uint32_t split;        // lanes per job (layout.hpp vsplit_lo): y = job * split + r, lane r runs the windows [lo(r), lo(r + 1)) and
                         // writes J slot job.out + r; 1 = the whole chain in one lane
};
template <class C>
MP_HD void body_var_msm(const VarArgs& a, uint32_t b, uint32_t y) {
  const uint32_t jidx = a.split > 1 ? y / a.split : y, r = y - jidx * a.split;
  const Job job = a.jobs[jidx];
  const int w_top = a.split > 1 ? (int)vsplit_lo(r + 1, a.split, a.nwin) - 1 : (int)a.nwin - 1;
  const int w_low = a.split > 1 ? (int)vsplit_lo(r, a.split, a.nwin) : 0;
  Xyzz<C> acc = xyzz_inf<C>();     // XYZZ accumulator: a job is ~25..60 mixed additions per 5 doublings
#pragma unroll 1
  for (int w = w_top; w >= w_low; --w) {
    if (w != w_top) {
#pragma unroll 1
      for (int q = 0; q < VB_WINDOW_BITS; ++q) xyzz_dbl_ip<C>(acc);
    }
#pragma unroll 1
    for (uint32_t t = 0; t < job.count; ++t) {
      const Term term = a.terms[job.begin + t];
      const int d = a.D[((size_t)term.s * a.nwin + w) * a.Bpad + b];
      if (d != 0) {
        const uint32_t e = (uint32_t)(d < 0 ? -d : d) - 1;
        const Aff<C> q = ld_aff<C>(a.T + p_off<C>(term.b * VB_ENTRIES + e, a.Bpad, b));
        xyzz_madd_signed_ip<C>(acc, q, d < 0);
      }
    }
  }
  st_jac<C>(a.J + j_off<C>(job.out + r, a.Bpad, b), xyzz_to_jac<C>(acc));
}
MP_KERNEL_OCC(k_var_msm, VarArgs, body_var_msm, Geo<C>::OCC_VAR)

// ---- combine partial sums ---------------------------------------------------------------------------
struct CombineArgs {
  uint32_t* J;
  const uint32_t* P;
  const Job* jobs;
  const Term* terms;
  uint32_t Bpad;
};
template <class C>
MP_HD void body_combine(const CombineArgs& a, uint32_t b, uint32_t y) {
  const Job job = a.jobs[y];
  Jac<C> acc = jac_inf<C>();
  for (uint32_t t = 0; t < job.count; ++t) {
    const uint32_t s = a.terms[job.begin + t].s;
    if (s & AFF_FLAG) {
      Aff<C> q = ld_aff<C>(a.P + p_off<C>(s & SLOT_MASK, a.Bpad, b));
      if (s & NEG_FLAG) q = aff_neg<C>(q);
      jac_madd_ip<C>(acc, q);
    } else {
      Jac<C> q = ld_jac<C>(a.J + j_off<C>(s & SLOT_MASK, a.Bpad, b));
      if (s & NEG_FLAG) q.Y = fe_neg<typename C::FqP>(q.Y);
      jac_add_ip<C>(acc, q);
    }
  }
  st_jac<C>(a.J + j_off<C>(job.out, a.Bpad, b), acc);
}
MP_KERNEL_OCC(k_combine, CombineArgs, body_combine, Geo<C>::OCC4)

// ---- batch normalisation Jacobian -> affine (Montgomery's trick, one inversion per `chunk` points) ----
// Works on FLAT arrays: element e of the source is 24 words at src + 24 e; a slot range of an arena is
// such an array.  Thread x owns elements x, x + nthreads, x + 2 nthreads, ... so accesses stay coalesced;
// prefix products go through `scratch` (8 words per element, same indexing).
struct NormArgs {
  const uint32_t* src;
  uint32_t* dst;
  uint32_t* scratch;
  uint32_t count, nthreads, chunk;
};
template <class C>
MP_HD void body_normalize(const NormArgs& a, uint32_t x, uint32_t y) {
  typedef typename C::FqP F;
  Fe<F> prod = fe_one<F>();
  uint32_t nmine = 0;
  for (uint32_t i = 0; i < a.chunk; ++i) {
    const size_t e = (size_t)x + (size_t)i * a.nthreads;
    if (e >= a.count) break;
    st_fe<F>(a.scratch + e * Geo<C>::FW, prod);               // product of the Z's before this element
    Fe<F> z = ld_fe<F>(a.src + e * Geo<C>::JW + 2 * Geo<C>::FW);
    if (!fe_is_zero(z)) prod = fe_mul<F>(prod, z);
    nmine = i + 1;
  }
  Fe<F> inv = fe_inv<F>(prod);
  for (uint32_t i = nmine; i-- > 0;) {
    const size_t e = (size_t)x + (size_t)i * a.nthreads;
    Jac<C> j = ld_jac<C>(a.src + e * Geo<C>::JW);
    Aff<C> out = aff_inf<C>();
    if (!fe_is_zero(j.Z)) {
      Fe<F> before = ld_fe<F>(a.scratch + e * Geo<C>::FW);
      Fe<F> zinv = fe_mul<F>(inv, before);
      inv = fe_mul<F>(inv, j.Z);
      out = jac_to_aff_with_zinv<C>(j, zinv);
    }
    st_aff<C>(a.dst + e * Geo<C>::PW, out);
  }
}
MP_KERNEL(k_normalize, NormArgs, body_normalize)
// several slot ranges of the J arena in ONE launch (the ranges a phase normalises are independent; launched one after the other
// each of them is a full inversion deep, which is what a small batch waits for): thread x belongs to range r with
// tstart[r] <= x < tstart[r + 1] and does there what body_normalize does
static const uint32_t NORM_MAX_RANGES = 4;
struct NormMultiArgs {
  const uint32_t* J;
  uint32_t* P;
  uint32_t* scratch;
  uint32_t nr, chunk;
  uint32_t first[NORM_MAX_RANGES];     // first element (slot * Bpad) of the range in both arenas
  uint32_t count[NORM_MAX_RANGES];     // elements
  uint32_t tstart[NORM_MAX_RANGES + 1];   // first thread of the range
  uint32_t sstart[NORM_MAX_RANGES];    // first scratch element of the range
};
template <class C>
MP_HD void body_normalize_multi(const NormMultiArgs& a, uint32_t x, uint32_t y) {
  uint32_t r = 0;
  while (r + 1 < a.nr && x >= a.tstart[r + 1]) ++r;
  NormArgs n;
  n.src = a.J + (size_t)a.first[r] * Geo<C>::JW;
  n.dst = a.P + (size_t)a.first[r] * Geo<C>::PW;
  n.scratch = a.scratch + (size_t)a.sstart[r] * Geo<C>::FW;
  n.count = a.count[r];
  n.nthreads = a.tstart[r + 1] - a.tstart[r];
  n.chunk = a.chunk;
  body_normalize<C>(n, x - a.tstart[r], y);
}
MP_KERNEL(k_normalize_multi, NormMultiArgs, body_normalize_multi)

// ---- Toom-Cook, ciphertext side: C(x) = sum_s x^s c_s at x = +-1 .. +-(m-1) for one column point of the shuffled deck
// (c_s = row m - s; x = proof, y = 2 t + component).  Even / odd split: C(+-x) = Ce(x^2) +- x Co(x^2), Horner in x^2; the small
// integer multiples are double-and-add chains on Jacobian points (x <= 8 with the reciprocal points: x^2 <= 64).
// Jacobian out -> k_normalize -> the operand vectors of the 2m - 2 products.
struct ToomPointsArgs {
  const uint32_t* P;
  uint32_t* J;
  uint32_t Bpad, m, n, p_shuf, cv_first;
};
template <class C>
MP_HD void jac_mul_small_ip(Jac<C>& p, uint32_t k) {      // p <- k p, 1 <= k < 256
  if (k == 1) return;
  const Jac<C> base = p;
  int top = 7;
  while (!((k >> top) & 1u)) --top;
  for (int i = top - 1; i >= 0; --i) {
    jac_dbl_ip<C>(p);
    if ((k >> i) & 1u) jac_add_ip<C>(p, base);
  }
}
template <class C>
MP_HD void body_toom_points(const ToomPointsArgs& a, uint32_t b, uint32_t y) {
  const uint32_t t = y >> 1, comp = y & 1u, m = a.m;
  const uint32_t top_even = (m - 1) & ~1u, top_odd = ((m - 1) & 1u) ? m - 1 : m - 2;      // m >= 3: both exist
#pragma unroll 1
  for (uint32_t p = 0; p + 1 < m; ++p) {            // pair p: +-x, direct or reversed coefficient order (layout.hpp ToomPlan)
    const uint32_t x = p == 0 ? 1u : (p + 1) / 2 + 1, yy = x * x;
    const bool rev = p != 0 && (p & 1u) == 0;
    // coefficient s of the (possibly reversed) polynomial: c_s = row m - s, at P slot p_shuf + 2 ((m - 1 - s) n + t) + comp
    auto cs = [&](uint32_t s) {
      const uint32_t row = rev ? s : m - 1 - s;
      return ld_aff<C>(a.P + p_off<C>(a.p_shuf + 2 * (row * a.n + t) + comp, a.Bpad, b));
    };
    Jac<C> ce = jac_from_aff<C>(cs(top_even));
#pragma unroll 1
    for (int s = (int)top_even - 2; s >= 0; s -= 2) {
      jac_mul_small_ip<C>(ce, yy);
      jac_madd_ip<C>(ce, cs((uint32_t)s));
    }
    Jac<C> co = jac_from_aff<C>(cs(top_odd));
#pragma unroll 1
    for (int s = (int)top_odd - 2; s >= 1; s -= 2) {
      jac_mul_small_ip<C>(co, yy);
      jac_madd_ip<C>(co, cs((uint32_t)s));
    }
    jac_mul_small_ip<C>(co, x);
    Jac<C> plus = ce;
    jac_add_ip<C>(plus, co);
    co.Y = fe_neg<typename C::FqP>(co.Y);
    jac_add_ip<C>(ce, co);
    // e = 2 + 2p (+x) and 3 + 2p (-x); vectors of 2n points each
    st_jac<C>(a.J + j_off<C>(a.cv_first + (2 * p) * 2 * a.n + y, a.Bpad, b), plus);
    st_jac<C>(a.J + j_off<C>(a.cv_first + (2 * p + 1) * 2 * a.n + y, a.Bpad, b), ce);
  }
}
MP_KERNEL_OCC(k_toom_points, ToomPointsArgs, body_toom_points, 2)

// ---- subgroup membership of wire points on a curve with a cofactor (BLS12-377 G1): [q]P == O, q = the prime group order.
// wire_to_aff only proves "on the curve"; the batched-affine tables and the completeness rules of the group law assume the
// prime-order subgroup, and an off-subgroup component of small order would pass an equation with probability ~1/order.  This is
// the check ark-ec 0.3 performs when it deserialises a point.  253 doublings + ~126 additions per point: dearer than the
// verifier's own work per point, so it only exists on curves that need it and can be switched off by callers whose points come
// from a validating deserialiser (mp_set_subgroup_check).
struct SubgroupArgs {
  const uint32_t* P;
  int32_t* status;
  uint32_t Bpad, p_first;
};
// is the affine point p (on the curve, not the identity) in the prime-order subgroup?
template <class C>
MP_HD bool aff_in_subgroup_dev(const Aff<C>& p) {
  typedef typename C::FrP R;
  if constexpr (C::ENDO_SUBGROUP) {
    // The same verdict from 126 doublings + 12 additions instead of 253 + 126 (round 3: this kernel was half of a BLS12-377 step).
    // phi(x, y) = (beta x, y) acts on G1 as -u^2 (u = the curve's seed, r = u^4 - u^2 + 1), and psi = phi + [u^2] has degree
    // Norm(u^2 + omega) = r, so its kernel IS the subgroup of order r:  P in G1  <=>  phi(P) = -[u^2] P  (curve_params.hpp;
    // checked against [r]P = O by tools/gen_curve_params.py's note and tests/test_gpu_canonical.py's off-subgroup points).
    typedef typename C::FqP F;
    Jac<C> q1 = jac_from_aff<C>(p);
#pragma unroll 1
    for (int i = 62; i >= 0; --i) {              // [u] P  (u has bit 63 set)
      jac_dbl_ip<C>(q1);
      if ((C::SEED >> i) & 1u) jac_madd_ip<C>(q1, p);
    }
    Jac<C> q2 = q1;
#pragma unroll 1
    for (int i = 62; i >= 0; --i) {              // [u] ([u] P)
      jac_dbl_ip<C>(q2);
      if ((C::SEED >> i) & 1u) jac_add_ip<C>(q2, q1);
    }
    bool ok = !jac_is_inf<C>(q2);                // (X : Y : Z) = -(beta x, y):  beta x Z^2 = X,  y Z^3 = -Y
    const Fe<F> zz = fe_sqr<F>(q2.Z);
    ok = ok && fe_is_zero(fe_sub<F>(fe_mul<F>(fe_mul<F>(fe_unpack<F>(C::BETA_MONT), p.x), zz), q2.X));
    ok = ok && fe_is_zero(fe_add<F>(fe_mul<F>(fe_mul<F>(p.y, zz), q2.Z), q2.Y));
    return ok;
  } else {
    Jac<C> acc = jac_from_aff<C>(p);
#pragma unroll 1
    for (int i = R::BITS - 2; i >= 0; --i) {
      jac_dbl_ip<C>(acc);
      if ((R::MOD[i >> 5] >> (i & 31)) & 1u) jac_madd_ip<C>(acc, p);
    }
    return jac_is_inf<C>(acc);
  }
}
template <class C>
MP_HD void body_subgroup_check(const SubgroupArgs& a, uint32_t b, uint32_t y) {
  const Aff<C> p = ld_aff<C>(a.P + p_off<C>(a.p_first + y, a.Bpad, b));
  if (aff_is_inf<C>(p)) return;
  if (!aff_in_subgroup_dev<C>(p)) a.status[b] = -1;      // ST_BAD_ENCODING (kernels_proto.hpp)
}
MP_KERNEL_OCC(k_subgroup_check, SubgroupArgs, body_subgroup_check, 2)

// ---- construction of the fixed-base tables (setup time, once per table context) -------------------------
// pass 1: window bases W_w = 2^(8w) * B for every base (thread = base), Jacobian out -> normalise
struct FbWinArgs {
  const uint32_t* bases;   // [nbases] affine
  uint32_t* WJ;            // [base][window] Jacobian
  FbGeom g;
  uint32_t bits2;          // width of the odd-numbered windows (= g.bits unless a wide window splits unevenly)
};
template <class C>
MP_HD void body_fb_windows(const FbWinArgs& a, uint32_t x, uint32_t y) {
  Jac<C> acc = jac_from_aff<C>(ld_aff<C>(a.bases + (size_t)x * Geo<C>::PW));
  for (uint32_t w = 0; w < a.g.windows; ++w) {
    st_jac<C>(a.WJ + ((size_t)x * a.g.windows + w) * Geo<C>::JW, acc);
    for (uint32_t q = 0; q < ((w & 1u) ? a.bits2 : a.g.bits); ++q) jac_dbl_ip<C>(acc);
  }
}
MP_KERNEL(k_fb_windows, FbWinArgs, body_fb_windows)
// pass 2: entries e * W_w, e = 1..255 (thread = (base, window)), Jacobian out -> normalise
struct FbFillArgs {
  const uint32_t* W;       // [base*window] affine
  uint32_t* EJ;            // [base*window][entries] Jacobian
  FbGeom g;
};
template <class C>
MP_HD void body_fb_fill(const FbFillArgs& a, uint32_t x, uint32_t y) {
  const Aff<C> w = ld_aff<C>(a.W + (size_t)x * Geo<C>::PW);
  Jac<C> acc = jac_from_aff<C>(w);
  uint32_t* out = a.EJ + (size_t)x * a.g.entries * Geo<C>::JW;
  st_jac<C>(out, acc);
  jac_dbl_ip<C>(acc);
  st_jac<C>(out + Geo<C>::JW, acc);
  for (uint32_t e = 2; e < a.g.entries; ++e) {
    jac_madd_ip<C>(acc, w);
    st_jac<C>(out + (size_t)e * Geo<C>::JW, acc);
  }
}
MP_KERNEL(k_fb_fill, FbFillArgs, body_fb_fill)

// pass 3 (wide windows): entry d = hi * 2^h + lo of the 2h-bit window w is Th[2w+1][hi] + Th[2w][lo], Th = the table
// with h-bit windows (thread = (base * windows + w) * entries + d - 1), Jacobian out -> normalise
struct FbWidenArgs {
  const uint32_t* Th;      // [base][gh.windows][gh.entries] affine
  uint32_t* EJ;            // [base][g.windows][g.entries] Jacobian
  FbGeom gh, g;
};
template <class C>
MP_HD void body_fb_widen(const FbWidenArgs& a, uint32_t x, uint32_t y) {
  const uint32_t d = x % a.g.entries + 1u;
  const uint32_t bw = x / a.g.entries;           // base * g.windows + w
  const uint32_t base = bw / a.g.windows, w = bw % a.g.windows;
  const uint32_t hi = d >> a.gh.bits, lo = d & a.gh.entries;
  const uint32_t* tlo = a.Th + ((size_t)base * a.gh.windows + 2 * w) * a.gh.entries * Geo<C>::PW;
  Jac<C> acc = jac_inf<C>();
  if (lo) acc = jac_from_aff<C>(ld_aff<C>(tlo + (size_t)(lo - 1) * Geo<C>::PW));
  if (hi && 2 * w + 1 < a.gh.windows) jac_madd_ip<C>(acc, ld_aff<C>(tlo + ((size_t)a.gh.entries + hi - 1) * Geo<C>::PW));
  st_jac<C>(a.EJ + (size_t)x * Geo<C>::JW, acc);
}
MP_KERNEL(k_fb_widen, FbWidenArgs, body_fb_widen)

// The group-arithmetic kernels are compiled in their own translation unit per curve (curve_<name>_msm.hip instantiates them,
// curve_<name>.hip only declares them): it halves the build time of the slowest units.
#define MP_MSM_KERNELS(X, C) \
  MP_KERNEL_INST(X, k_fixed_msm, FixedArgs, C) \
  MP_KERNEL_INST(X, k_remask, RemaskArgs, C) \
  MP_KERNEL_INST(X, k_key_windows, KeyWinArgs, C) \
  MP_KERNEL_INST(X, k_key_terms, KeyTermsArgs, C) \
  MP_KERNEL_INST(X, k_gather_keys, GatherKeysArgs, C) \
  MP_KERNEL_INST(X, k_recode, RecodeArgs, C) \
  MP_KERNEL_INST(X, k_table, TableArgs, C) \
  MP_KERNEL_INST(X, k_var_msm, VarArgs, C) \
  MP_KERNEL_INST(X, k_combine, CombineArgs, C) \
  MP_KERNEL_INST(X, k_normalize, NormArgs, C) \
  MP_KERNEL_INST(X, k_normalize_multi, NormMultiArgs, C) \
  MP_KERNEL_INST(X, k_fb_windows, FbWinArgs, C) \
  MP_KERNEL_INST(X, k_fb_fill, FbFillArgs, C) \
  MP_KERNEL_INST(X, k_fb_widen, FbWidenArgs, C) \
  MP_KERNEL_INST(X, k_subgroup_check, SubgroupArgs, C) \
  MP_KERNEL_INST(X, k_toom_points, ToomPointsArgs, C)

}  // namespace mp
