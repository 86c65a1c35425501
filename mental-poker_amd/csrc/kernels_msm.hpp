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
//                    SURVEY.md App. D), doubling chain shared by the terms of a job.
//   body_table / body_recode / body_combine / body_normalize: their supporting passes.
#pragma once
#include "curve.hpp"
#include "layout.hpp"

namespace mp {

// ---- arena access (16-byte vector loads/stores; arenas are 256-byte aligned) ----------------------------------
// Sizes in 32-bit words of the arena elements of curve C: a base-field element is FW packed words (8; 12 on BLS12-377),
// an affine point 2 FW, a Jacobian point 3 FW; a scalar (Fr) is 8 words on every supported curve.
template <class C>
struct Geo {
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
// geometry of the fixed-base tables of a table context: `bits`-wide unsigned windows (8, 16 or 20)
struct FbGeom {
  uint32_t bits, windows, entries;   // windows = ceil(256 / bits), entries = 2^bits - 1
};
struct FixedArgs {
  const uint32_t* S;
  uint32_t* J;
  const uint32_t* FB;   // [base][window][entry(1..entries)] affine, Geo<C>::PW words each
  const Job* jobs;
  const Term* terms;
  uint32_t Bpad;
  FbGeom g;
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
  Jac<C> acc = jac_inf<C>();
  for (uint32_t t = 0; t < job.count; ++t) {
    const Term term = a.terms[job.begin + t];
    uint32_t k[8];
    fe_to_canonical<R>(ld_fe<R>(a.S + s_off(term.s, a.Bpad, b)), k);
#pragma unroll 1
    for (uint32_t w = 0; w < a.g.windows; ++w) {
      const uint32_t d = fb_digit(k, a.g, w);
      if (d) jac_madd_ip<C>(acc, ld_aff<C>(fb_entry<C>(a.FB, a.g, term.b, w, d)));
    }
  }
  st_jac<C>(a.J + j_off<C>(job.out, a.Bpad, b), acc);
}
MP_KERNEL_OCC(k_fixed_msm, FixedArgs, body_fixed_msm, 4)

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
  Jac<C> acc = jac_inf<C>();
#pragma unroll 1
  for (uint32_t w = 0; w < a.g.windows; ++w) {
    const uint32_t d = fb_digit(k, a.g, w);
    if (d) jac_madd_ip<C>(acc, ld_aff<C>(fb_entry<C>(a.FB, a.g, base, w, d)));
  }
  jac_madd_ip<C>(acc, ld_aff<C>(a.P + p_off<C>(a.p_deck + 2 * src + comp, a.Bpad, b)));
  st_jac<C>(a.J + j_off<C>(a.j_out + y, a.Bpad, b), acc);
}
MP_KERNEL_OCC(k_remask, RemaskArgs, body_remask, 4)

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
// rounds -- {2P}, {3P,4P}, {5P..8P}, {9P..16P}: tP = 2*(t/2)P for even t, tP = 2^k P + (t-2^k)P for odd t, so
// every operand comes from an earlier round -- and ALL slopes of a round (up to 8 x TABLE_GROUP) share ONE
// Fermat inversion (Montgomery's trick; prefix products through HBM scratch).
// ~5M + 1S per entry plus 4/15 of 1/64 of an inversion, against 7M+4S (Jacobian chain) + ~13M (normalisation).
static const uint32_t TABLE_GROUP = 64;
struct TableArgs {
  const uint32_t* P;
  uint32_t* T;           // [tslot][entry][Bpad] affine
  uint32_t* scratch;     // [tslot][8][Bpad] field elements (prefix products)
  const Term* list;      // {P slot, table slot}; table slots are 0..n_tables-1 in list order
  uint32_t Bpad, n_tables, group;   // group = bases per lane (<= TABLE_GROUP)
};
// operands of target multiple t (entry index t-1) in the round that starts at multiple e0+1 (e0 = 1, 2, 4, 8)
MP_HD void table_operands(uint32_t t, uint32_t e0, uint32_t& ia, uint32_t& ib, bool& dbl) {
  dbl = (t & 1u) == 0;
  if (dbl) {
    ia = ib = t / 2 - 1;
  } else {
    ia = e0 - 1;          // (2^k) P
    ib = t - e0 - 1;      // (t - 2^k) P
  }
}
template <class C>
MP_HD void body_table(const TableArgs& a, uint32_t b, uint32_t y) {
  typedef typename C::FqP F;
  const uint32_t g0 = y * a.group;
  const uint32_t g1 = g0 + a.group < a.n_tables ? g0 + a.group : a.n_tables;
  for (uint32_t g = g0; g < g1; ++g) {   // entry 0 = P
    const Term t = a.list[g];
    st_aff<C>(a.T + p_off<C>(t.b * VB_ENTRIES, a.Bpad, b), ld_aff<C>(a.P + p_off<C>(t.s, a.Bpad, b)));
  }
#pragma unroll 1
  for (uint32_t e0 = 1; e0 < (uint32_t)VB_ENTRIES; e0 *= 2) {   // targets: entries e0 .. 2*e0-1
    // pass 1: denominators (2y for a doubling, x_b - x_a for an addition) and their running product.  Every addition of
    // the round has the same first operand (2^k P, entry e0 - 1): loaded once per base, not once per target.
    Fe<F> prod = fe_one<F>();
    for (uint32_t g = g0; g < g1; ++g) {
      const uint32_t ts = a.list[g].b;
      const Fe<F> xa = ld_fe<F>(a.T + p_off<C>(ts * VB_ENTRIES + e0 - 1, a.Bpad, b));
#pragma unroll 1
      for (uint32_t i = 0; i < e0; ++i) {
        uint32_t ia, ib;
        bool dbl;
        table_operands(e0 + i + 1, e0, ia, ib, dbl);
        Fe<F> den;
        if (dbl)
          den = fe_dbl<F>(ld_fe<F>(a.T + p_off<C>(ts * VB_ENTRIES + ia, a.Bpad, b) + Geo<C>::FW));
        else
          den = fe_sub<F>(ld_fe<F>(a.T + p_off<C>(ts * VB_ENTRIES + ib, a.Bpad, b)), xa);
        st_fe<F>(a.scratch + f_off<C>(ts * 8 + i, a.Bpad, b), prod);
        if (!fe_is_zero(den)) prod = fe_mul<F>(prod, den);   // zero only for P = infinity (prime-order group)
      }
    }
    Fe<F> inv = fe_inv<F>(prod);
    // pass 2 (exact reverse order): slope, new point
    for (uint32_t g = g1; g-- > g0;) {
      const uint32_t ts = a.list[g].b;
      const Aff<C> pe = ld_aff<C>(a.T + p_off<C>(ts * VB_ENTRIES + e0 - 1, a.Bpad, b));   // 2^k P
#pragma unroll 1
      for (uint32_t i = e0; i-- > 0;) {
        uint32_t ia, ib;
        bool dbl;
        table_operands(e0 + i + 1, e0, ia, ib, dbl);
        Aff<C> pa = pe, pb;
        Fe<F> den;
        if (dbl) {
          if (ia != e0 - 1) pa = ld_aff<C>(a.T + p_off<C>(ts * VB_ENTRIES + ia, a.Bpad, b));
          pb = pa;
          den = fe_dbl<F>(pa.y);
        } else {
          pb = ld_aff<C>(a.T + p_off<C>(ts * VB_ENTRIES + ib, a.Bpad, b));
          den = fe_sub<F>(pb.x, pa.x);
        }
        Aff<C> out = aff_inf<C>();
        if (!fe_is_zero(den)) {
          const Fe<F> dinv = fe_mul<F>(inv, ld_fe<F>(a.scratch + f_off<C>(ts * 8 + i, a.Bpad, b)));
          inv = fe_mul<F>(inv, den);
          Fe<F> num;
          if (dbl) {
            const Fe<F> xx = fe_sqr<F>(pa.x);
            num = fe_add<F>(fe_dbl<F>(xx), xx);
            if (C::A == 1) num = fe_add<F>(num, fe_one<F>());
          } else {
            num = fe_sub<F>(pb.y, pa.y);
          }
          const Fe<F> lam = fe_mul<F>(num, dinv);
          out.x = fe_sub<F>(fe_sub<F>(fe_sqr<F>(lam), pa.x), pb.x);
          out.y = fe_sub<F>(fe_mul<F>(lam, fe_sub<F>(pa.x, out.x)), pa.y);
        }
        st_aff<C>(a.T + p_off<C>(ts * VB_ENTRIES + e0 + i, a.Bpad, b), out);
      }
    }
  }
}
MP_KERNEL_OCC(k_table, TableArgs, body_table, 4)

// ---- variable-base MSM (Straus) -----------------------------------------------------------------------
struct VarArgs {
  const int8_t* D;
  const uint32_t* T;     // [tslot][entry][Bpad] affine
  uint32_t* J;
  const Job* jobs;
  const Term* terms;     // {digit slot, table slot}
  uint32_t Bpad, nwin;
};
template <class C>
MP_HD void body_var_msm(const VarArgs& a, uint32_t b, uint32_t y) {
  const Job job = a.jobs[y];
  Jac<C> acc = jac_inf<C>();
#pragma unroll 1
  for (int w = (int)a.nwin - 1; w >= 0; --w) {
    if (w != (int)a.nwin - 1) {
#pragma unroll 1
      for (int q = 0; q < VB_WINDOW_BITS; ++q) jac_dbl_ip<C>(acc);
    }
#pragma unroll 1
    for (uint32_t t = 0; t < job.count; ++t) {
      const Term term = a.terms[job.begin + t];
      const int d = a.D[((size_t)term.s * a.nwin + w) * a.Bpad + b];
      if (d != 0) {
        const uint32_t e = (uint32_t)(d < 0 ? -d : d) - 1;
        Aff<C> q = ld_aff<C>(a.T + p_off<C>(term.b * VB_ENTRIES + e, a.Bpad, b));
        if (d < 0) q = aff_neg<C>(q);
        jac_madd_ip<C>(acc, q);
      }
    }
  }
  st_jac<C>(a.J + j_off<C>(job.out, a.Bpad, b), acc);
}
MP_KERNEL_OCC(k_var_msm, VarArgs, body_var_msm, 4)

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
MP_KERNEL_OCC(k_combine, CombineArgs, body_combine, 4)

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

// ---- construction of the fixed-base tables (setup time, once per table context) -------------------------
// pass 1: window bases W_w = 2^(8w) * B for every base (thread = base), Jacobian out -> normalise
struct FbWinArgs {
  const uint32_t* bases;   // [nbases] affine
  uint32_t* WJ;            // [base][window] Jacobian
  FbGeom g;
};
template <class C>
MP_HD void body_fb_windows(const FbWinArgs& a, uint32_t x, uint32_t y) {
  Jac<C> acc = jac_from_aff<C>(ld_aff<C>(a.bases + (size_t)x * Geo<C>::PW));
  for (uint32_t w = 0; w < a.g.windows; ++w) {
    st_jac<C>(a.WJ + ((size_t)x * a.g.windows + w) * Geo<C>::JW, acc);
    for (uint32_t q = 0; q < a.g.bits; ++q) jac_dbl_ip<C>(acc);
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

}  // namespace mp
