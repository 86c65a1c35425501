// engine_core.hpp -- the per-curve engine: batch workspace, plans on the device, prove / verify pipelines.
// Included by one translation unit per curve (curve_*.hip).
#pragma once
#include "engine_base.hpp"
#include "kernels_bucket.hpp"
#include "kernels_decompress.hpp"
#include "kernels_sigma.hpp"
#include "serialize_host.hpp"
#include "setup_host.hpp"

namespace mp {

static const uint32_t FCHUNK = 8;     // fixed-base terms per sub-job (8 x 13..32 mixed additions, by window width)
static const uint32_t VCHUNK = 64;    // variable-base terms per sub-job: the terms of a job share one 250-doubling chain
static const uint32_t NORM_CHUNK = 64;   // points per Fermat inversion in k_normalize
static const uint32_t BUCKET_MIN = 2048; // variable-base terms from which an MSM runs on the bucket kernel (kernels_bucket.hpp)
static const uint32_t BUCKET_MIN_SMALL_BATCH = 512;   // ... in the finest split (latency, not throughput): the merged equation of a 300-card proof (1 239 terms), not that of a 52-card one (239: Straus verifies one proof in 1.4 ms, the rebuilt bucket kernel in 1.6; profiles/r05d_latency.txt)

struct PhaseDev {
  DevBuf<Term> recode, tables, fterms, vterms, cterms, cterms2, cterms0;
  DevBuf<Job> fjobs, vjobs, cjobs, cjobs2, cjobs0;
  DevBuf<BJob> bjobs;
  DevBuf<Term> bterms;
  DevBuf<BTermPos> bpos;
  DevBuf<Job> wcjobs;       // window-split Straus jobs (layout.hpp Phase::vsplit): range sums over an MSM's sub-jobs,
  DevBuf<Term> wcterms;
  DevBuf<BJob> wjobs;       // and the fold of an MSM's range sums
  uint32_t vsplit = 1, n_wc = 0, n_w = 0;
  uint32_t n_b = 0, n_bterms = 0, b_dig_bytes = 0, b_kpad_max = 0, b_bits = 8;
  uint32_t n_recode = 0, n_tables = 0, n_f = 0, n_v = 0, n_c = 0, n_c2 = 0, n_c0 = 0, n_tslots = 0, n_dslots = 0;
  std::vector<std::pair<uint32_t, uint32_t>> normalize;
  void upload(const Phase& ph, rt::Stream s) {
    recode.upload(ph.recode, s);
    tables.upload(ph.tables, s);
    fterms.upload(ph.fterms, s);
    vterms.upload(ph.vterms, s);
    cterms.upload(ph.cterms, s);
    fjobs.upload(ph.fjobs, s);
    vjobs.upload(ph.vjobs, s);
    cjobs.upload(ph.cjobs, s);
    cterms2.upload(ph.cterms2, s);
    cjobs2.upload(ph.cjobs2, s);
    n_c2 = (uint32_t)ph.cjobs2.size();
    cterms0.upload(ph.cterms0, s);
    cjobs0.upload(ph.cjobs0, s);
    n_c0 = (uint32_t)ph.cjobs0.size();
    wcjobs.upload(ph.wcjobs, s);
    wcterms.upload(ph.wcterms, s);
    wjobs.upload(ph.wjobs, s);
    vsplit = ph.vsplit;
    n_wc = (uint32_t)ph.wcjobs.size();
    n_w = (uint32_t)ph.wjobs.size();
    bjobs.upload(ph.bjobs, s);
    bterms.upload(ph.bterms, s);
    bpos.upload(ph.bpos, s);
    n_b = (uint32_t)ph.bjobs.size();
    n_bterms = (uint32_t)ph.bterms.size();
    b_dig_bytes = ph.b_dig_bytes;
    b_kpad_max = ph.b_kpad_max;
    b_bits = ph.b_bits;
    n_recode = (uint32_t)ph.recode.size();
    n_tables = (uint32_t)ph.tables.size();
    n_f = (uint32_t)ph.fjobs.size();
    n_v = (uint32_t)ph.vjobs.size();
    n_c = (uint32_t)ph.cjobs.size();
    n_tslots = ph.n_tslots;
    n_dslots = ph.n_dslots;
    normalize = ph.normalize;
  }
};

// per-batch arenas
struct Workspace {
  uint32_t fw = 8;      // words of a base-field element (Geo<C>::FW of the owning table: 8, or 12 on BLS12-377)
  uint32_t Bpad = 0;
  uint32_t nS = 0, nP = 0, nJ = 0, nD = 0, nT = 0, nwin = 0, stage_words = 0;
  DevBuf<uint32_t> S, P, J, T, NS, NS2, stage, seed, direct;      // NS2: inversion scratch of the main stream while `side` has NS (prove_dev: m points per proof, allocated by reserve_ws for batches that fork)
  DevBuf<uint32_t> W;       // wire words of the loaded decks, [slot][word][Bpad] (what the transcript hashes; LoadPointsArgs::W)
  DevBuf<int8_t> D;
  DevBuf<int16_t> D16;      // bucket-method digits, proof-major: [b][d8_bytes] (d8_bytes counts digits)
  uint32_t d8_bytes = 0;
  // scratch rows of the bucket kernel's persistent waves (kernels_bucket.hpp): the sorted point references of the window a wave is
  // working on, and its parked bucket sums -- at most 2 048 x (128 KB + 64 KB), whatever the batch
  DevBuf<uint32_t> bk_sorted, bk_park, bk_counter, bk_timing, bk_quarters;
  DevBuf<uint16_t> bk_offs;           // the split pipeline's bucket offsets per (item, chunk)
  void ensure_bucket(uint32_t nslots, uint32_t kpad_max, uint32_t bits, uint32_t xw, rt::Stream s) {
    bk_sorted.alloc((size_t)nslots * kpad_max, s, false);
    bk_park.alloc((size_t)nslots * bk_buckets(bits) * xw, s, false);
    bk_counter.alloc(8, s);
  }
  DevBuf<int32_t> status;
  // The arenas are slot-major with the lane stride Bpad.  Bpad follows the batch (B rounded up to a wave), whatever the arenas were
  // allocated for (`cap` lanes): a 1 024-proof batch laid out with the stride of an earlier 262 144-proof one touches 32 KB out of
  // every 8 MB -- a TLB miss per slot -- and ran at 100 k proofs/s instead of 160 k (round 4).  Lanes [B, Bpad) of the J arena are
  // read by the flat normalisation kernels and must hold zeros (Z = 0: skipped), so a change of stride clears the part in use.
  uint32_t cap = 0;
  void ensure(uint32_t B, uint32_t nS_, uint32_t nP_, uint32_t nJ_, uint32_t nD_, uint32_t nT_, uint32_t nwin_,
              uint32_t stage_words_, rt::Stream s, uint32_t d8_bytes_ = 0) {
    const uint32_t need = (B + 63u) & ~63u;
    const bool grow = !(need <= cap && nS_ <= nS && nP_ <= nP && nJ_ <= nJ && nD_ <= nD && nT_ <= nT && stage_words_ <= stage_words &&
                        d8_bytes_ <= d8_bytes);
    if (grow) {
      cap = std::max(cap, need);
      nS = std::max(nS, nS_); nP = std::max(nP, nP_); nJ = std::max(nJ, nJ_); nD = std::max(nD, nD_); nT = std::max(nT, nT_);
      nwin = nwin_;
      stage_words = std::max(stage_words, stage_words_);
      // re-allocate everything (capacity grows monotonically); zero-filled so padding lanes hold valid data
      S.n = P.n = J.n = T.n = NS.n = stage.n = seed.n = direct.n = 0;
      D.n = 0;
      D16.n = 0;
      d8_bytes = std::max(d8_bytes, d8_bytes_);
      D16.alloc((size_t)std::max(d8_bytes, 4u) * cap, s);      // zero-filled: the padding digits of an MSM stay zero
      status.n = 0;
      S.alloc((size_t)nS * cap * 8, s);
      P.alloc((size_t)nP * cap * 2 * fw, s);
      J.alloc((size_t)nJ * cap * 3 * fw, s);
      D.alloc((size_t)std::max(nD, 1u) * nwin * cap, s);
      T.alloc((size_t)std::max(nT, 1u) * VB_ENTRIES * cap * 2 * fw, s);
      size_t norm_max = std::max((size_t)nT * 8, (size_t)nJ) * cap;   // k_table prefix products (8 per base), k_normalize ranges
      NS.alloc(norm_max * fw, s);
      stage.alloc((size_t)stage_words * cap, s);
      seed.alloc((size_t)8 * cap, s);
      direct.alloc((size_t)2 * cap, s);
      status.alloc(cap, s);
      Bpad = need;
      return;
    }
    if (Bpad != need) {
      Bpad = need;
      rt::dzero(J.p, (size_t)nJ * Bpad * 3 * fw * sizeof(uint32_t), s);      // (the bucket digits D16 are proof-major: no stride in them)
    }
  }
};

template <class C>
struct Table : mp_table {
  typedef typename C::FqP F;
  typedef typename C::FrP R;
  typedef Geo<C> G_;      // element sizes of this curve (words in the arenas, bytes on the wire)

  // One static plan per work split (below): the same results from sub-jobs of different sizes.
  struct PlanSet {
    ProvePlan pplan;
    VerifyPlan vplan;
    PhaseDev pph[6], vph, vmph;      // vmph: the merged verification plan (one MSM for all equations)
    DevBuf<MergeJob> mjobs;
    DevBuf<MergePair> mpairs;
    DevBuf<uint32_t> draws, lin_src, lin_coef, consts;      // consts: Fr constants of the Toom-Cook plan (8 words each)
    DevBuf<LinJob> lin;
    DevBuf<ProofElem> pwire, vwire;
    uint32_t table_group = TABLE_GROUP;
    uint32_t norm_chunk = NORM_CHUNK;
  };
  // Six static work splits per table, identical results.  Sub-job sizes (PlanParams below): fixed-base / variable-base terms per lane,
  // bases per table lane, points per shared inversion, and -- since round 4 -- lanes per variable-base sub-job (window split,
  // layout.hpp vsplit_lo: the windows of a sub-job dealt to k lanes, one fold per MSM).  Measured on an MI355X, 52 cards, proofs/s with
  // the verify calls pipelined (profiles/r04_plan_sweep.txt):
  //   [3] finest      1 / 1 / 2 / 4 / 1     up to ~128 proofs in flight (plus the bucket kernel for the merged verifier equation)
  //   [5] small       1 / 8 / 2 / 4 / 8     up to ~768      (256: 83 k against 55 k on the finest split, 512: 137 k against 90 k)
  //   [1] latency     1 / 16 / 4 / 8 / 16   up to ~2 560    (1 024: 168 k serial, 200 k pipelined; 2 048: 253 k; round 4's 2 / 16 / 4 / 8 / 8: 240 k at 2 048)
  //   [2] medium      4 / 32 / 8 / 16 / 4   up to ~6 144    (4 096: 338 k; round 3's 4 / 16 / 16 / 32 / 1: 242-255 k)
  //   [4] wide        4 / 64 / 16 / 32 / 16 up to ~49 152   (16 384: 462 k, 32 768: 483 k; round 3's 4 / 32 / 16 / 32 / 1: 402 k, 440 k)
  //   [0] throughput  8 / 64 / 64 / 64 / 1  beyond: fewest operations
  // Window lanes instead of ever smaller sub-jobs: a sub-job of T terms costs 250 doublings + 51 T additions whatever T is, so cutting
  // 64-term jobs into 4-term jobs for 16x the lanes multiplied the doublings by 16; dealing the 51 windows to 16 lanes gives the same
  // lanes for one extra fold (< 250 doublings) per MSM output.
  static const int N_PLANS = 6;
  struct PlanParams {
    uint32_t fch, vch, grp, nch, vsp;
  };
  PlanParams pprm[N_PLANS];
  void default_plan_params() {
    // ([3] on decks of up to 128 cards: one variable-base term per lane and two bases per table lane -- the partial sums go through
    // the two-level combine; measured on one MI355X: one 52-card proof 6.6 -> 6.1 ms against 2 terms / 4 bases, same or better up
    // to 256 proofs.  A single 300-card proof already brings more lanes than the chip holds: there the finer split only adds work,
    // BLS12-377 (30,10) 90 -> 117 ms.)
    const uint32_t tiny_v = N <= 128 ? 1u : 2u, tiny_g = N <= 128 ? 2u : 4u;
    pprm[0] = PlanParams{FCHUNK, VCHUNK, TABLE_GROUP, NORM_CHUNK, 1};
    pprm[1] = PlanParams{1, 16, 4, 8, 16};      // (round 5 re-sweep, profiles/r05k_plan_sweep_small.txt: 2 048 proofs 240 k -> 253 k/s against 2 / 16 / 4 / 8 / 8, 1 024 unchanged)
    pprm[2] = PlanParams{4, 32, 8, 16, 4};
    pprm[3] = PlanParams{1, tiny_v, tiny_g, 4, 1};
    pprm[4] = PlanParams{4, 64, 16, 32, 16};
    pprm[5] = PlanParams{1, 8, 2, 4, 8};
  }
  int set_plan_params(int plan, uint32_t fch, uint32_t vch, uint32_t grp, uint32_t nch, uint32_t vsp) override {
    if (plan < 0 || plan >= N_PLANS || !fch || !vch || !grp || grp > TABLE_GROUP || !nch || !vsp || vsp > VSPLIT_MAX) return MP_ERR_BAD_ARGUMENT;
    pprm[plan] = PlanParams{fch, vch, grp, nch, vsp};
    flush();
    rt::stream_sync(ctx->stream);
    build_plans(ps, false);
    psk_ready = false;
    chain.L = 0;
    gplans.clear();
    rt::stream_sync(ctx->stream);
    return MP_OK;
  }
  void set_plan_thresholds(size_t tiny, size_t small, size_t latency, size_t medium, size_t wide) override {
    auto cap = [](size_t v) { return (uint32_t)std::min<size_t>(v, 0x40000000u); };
    tiny_batch = cap(tiny); small_batch = cap(small); latency_batch = cap(latency); medium_batch = cap(medium); wide_batch = cap(wide);
  }
  PlanSet ps[N_PLANS];        // plans with the table's own aggregate key as a fixed base
  PlanSet psk[N_PLANS];       // plans for keyed batches (per-proof aggregate key): built on first use
  bool psk_ready = false;
  DevBuf<Term> key_recode, key_tables;          // static job lists of the per-proof key tables
  uint32_t key_d_first = 0, key_t_first = 0;     // first digit slot (rho_0) / table slot (window 0) of the key machinery
  // crossovers on 52-card decks (round 4, with window lanes; round 3 had 600 / 3 840 / 12 288 / 49 152 for its four finer splits)
  uint32_t tiny_batch = 128;                     // up to this size the finest split,
  uint32_t small_batch = 768;                    // the small split,
  uint32_t latency_batch = 2560;                 // the latency plan (mp_set_latency_batch scales all of them),
  uint32_t medium_batch = 6144;                  // the medium plan,
  uint32_t wide_batch = 49152;                   // the wide plan; larger batches: throughput
  int plan_of(uint32_t B) const {
    if (forced_split >= 0 && forced_split < N_PLANS) return forced_split;
    return B <= tiny_batch ? 3 : (B <= small_batch ? 5 : (B <= latency_batch ? 1 : (B <= medium_batch ? 2 : (B <= wide_batch ? 4 : 0))));
  }
  PlanSet& pick(uint32_t B, bool keyed = false) { return (keyed ? psk : ps)[plan_of(B)]; }
  void set_latency_batch(size_t b) override {
    latency_batch = (uint32_t)std::min<size_t>(b, 0x02000000u);
    tiny_batch = latency_batch / 20;
    small_batch = (uint32_t)((uint64_t)latency_batch * 3 / 10);
    medium_batch = (uint32_t)((uint64_t)latency_batch * 12 / 5);
    wide_batch = (uint32_t)((uint64_t)latency_batch * 96 / 5);
  }
  uint32_t bucket_min = BUCKET_MIN;              // MSMs of at least this many variable-base terms use the bucket kernel (0 = never)
  bool toom_cook = true;        // 3 <= m <= 16: Toom-Cook instead of Karatsuba for the multi-exponentiation diagonals
  void set_toom_cook(bool on) override {
    if (on == toom_cook) return;
    toom_cook = on;
    flush();
    rt::stream_sync(ctx->stream);
    build_plans(ps, false);
    psk_ready = false;
    rt::stream_sync(ctx->stream);
  }
  // Fr constants of a Toom-Cook plan: powers x_e^j of the evaluation points and the inverse Vandermonde matrix W (host)
  std::vector<uint32_t> toom_constants(const ToomPlan& T) {
    std::vector<uint32_t> out((size_t)std::max<uint32_t>(T.n_consts, 1) * 8, 0u);
    if (!T.E) return out;
    const uint32_t E = T.E, mm = E / 2;
    auto put = [&](uint32_t idx, const Fe<R>& v) { fe_pack<R>(v, &out[(size_t)idx * 8]); };
    std::vector<Fe<R>> X(E);
    auto powers = [&](const Fe<R>& x, uint32_t cnt) {                 // x^0 .. x^(cnt-1)
      std::vector<Fe<R>> p(cnt);
      p[0] = fe_one<R>();
      for (uint32_t k = 1; k < cnt; ++k) p[k] = fe_mul<R>(p[k - 1], x);
      return p;
    };
    for (uint32_t e = 2; e < E; ++e) {
      const int32_t x = T.x_of(e);
      X[e] = x >= 0 ? fe_from_u32<R>((uint32_t)x) : fe_neg<R>(fe_from_u32<R>((uint32_t)(-x)));
    }
    // scalar operand of point e: sum_j coef_j a_j with coef_j = x^j, or x^(m - j) for a reversed point
    for (uint32_t e = 2; e < E; ++e) {
      const std::vector<Fe<R>> pw = powers(X[e], mm + 1);
      for (uint32_t j = 0; j <= mm; ++j) put(e * (mm + 1) + j, T.rev_of(e) ? pw[mm - j] : pw[j]);
    }
    // V[e][k]: the product at point e is sum_k V[e][k] E_k -- x^k, or x^(2m - 1 - k) for a reversed point (reversed operands of
    // degrees m and m - 1); e = 0: (1, 0, ...), e = 1 (infinity): (0, ..., 0, 1).  W = V^-1 by Gauss-Jordan
    std::vector<Fe<R>> V((size_t)E * E, fe_zero<R>()), W((size_t)E * E, fe_zero<R>());
    for (uint32_t e = 0; e < E; ++e) {
      W[(size_t)e * E + e] = fe_one<R>();
      if (e == 0) {
        V[0] = fe_one<R>();
        continue;
      }
      if (e == 1) {
        V[(size_t)e * E + (E - 1)] = fe_one<R>();
        continue;
      }
      const std::vector<Fe<R>> pw = powers(X[e], E);
      for (uint32_t k = 0; k < E; ++k) V[(size_t)e * E + k] = T.rev_of(e) ? pw[E - 1 - k] : pw[k];
    }
    for (uint32_t col = 0; col < E; ++col) {
      uint32_t piv = col;
      while (piv < E && fe_is_zero(V[(size_t)piv * E + col])) ++piv;
      if (piv == E) throw std::logic_error("Toom-Cook: singular Vandermonde matrix");
      for (uint32_t k = 0; k < E; ++k) {
        std::swap(V[(size_t)piv * E + k], V[(size_t)col * E + k]);
        std::swap(W[(size_t)piv * E + k], W[(size_t)col * E + k]);
      }
      const Fe<R> inv = fe_inv<R>(V[(size_t)col * E + col]);
      for (uint32_t k = 0; k < E; ++k) {
        V[(size_t)col * E + k] = fe_mul<R>(V[(size_t)col * E + k], inv);
        W[(size_t)col * E + k] = fe_mul<R>(W[(size_t)col * E + k], inv);
      }
      for (uint32_t r = 0; r < E; ++r) {
        if (r == col) continue;
        const Fe<R> f = V[(size_t)r * E + col];
        if (fe_is_zero(f)) continue;
        for (uint32_t k = 0; k < E; ++k) {
          V[(size_t)r * E + k] = fe_sub<R>(V[(size_t)r * E + k], fe_mul<R>(f, V[(size_t)col * E + k]));
          W[(size_t)r * E + k] = fe_sub<R>(W[(size_t)r * E + k], fe_mul<R>(f, W[(size_t)col * E + k]));
        }
      }
    }
    // W now maps the products (index e) to the coefficients (index k): E_k = sum_e W[k][e] P_e
    for (uint32_t k = 0; k < E; ++k)
      for (uint32_t e = 0; e < E; ++e) put(T.w_const_first + k * E + e, W[(size_t)k * E + e]);
    return out;
  }
  void set_bucket_min(uint32_t terms) override {
    if (terms == bucket_min) return;
    bucket_min = terms;
    flush();
    rt::stream_sync(ctx->stream);
    build_plans(ps, false);          // the split between Straus sub-jobs and bucket jobs is part of the static plans
    psk_ready = false;
    rt::stream_sync(ctx->stream);
  }
  void set_bucket_bits(uint32_t bits) override {
    if (bits == bucket_bits) return;
    bucket_bits = bits;
    flush();
    rt::stream_sync(ctx->stream);
    build_plans(ps, false);
    psk_ready = false;
    chain.L = 0;                     // (the chain and group equations are rebuilt with the new width on their next use)
    gplans.clear();
    rt::stream_sync(ctx->stream);
  }
  uint32_t cur_table_group = TABLE_GROUP;
  uint32_t cur_norm_chunk = NORM_CHUNK;          // points per inversion in k_normalize: a property of the plan in use
  DevBuf<uint32_t> fbpts;    // (n+5) affine base points
  DevBuf<uint32_t> FB;       // fixed-base window tables
  Workspace ws;
  bool merged_verify = true;   // verify_dev screens the batch with the merged equation first (mp_set_merged_verify)
  void set_merged_verify(bool on) override { merged_verify = on; }
  // curves with a cofactor: every wire point of a call is tested for [q]P == O (kernels_msm.hpp k_subgroup_check)
  bool subgroup_check = !Cofactor<C>::ONE;
  void set_subgroup_check(bool on) override { subgroup_check = on && !Cofactor<C>::ONE; }
  void check_subgroup(Workspace& w, uint32_t B, uint32_t first, uint32_t count) {
    if (!subgroup_check || !count) return;
    SubgroupArgs a{w.P.p, w.status.p, w.Bpad, first};
    MP_RUN(k_subgroup_check, C, B, count, a);
  }
  // The wire points of a verify call: input deck, shuffled deck, the proof's points [, the proof's key] -- the P slots [0, pk (+ 1)).
  // What the caller has validated already (mp_set_validated: a deck that came through mp_deck_deserialize_dev or mp_deck_validate_dev,
  // as the reference's typed points came through CanonicalDeserialize [REF examples/parameter_selection.rs:78-91]) is not tested again:
  // a deck that passes along a chain of shuffles used to be tested in every call that touched it (round 6).
  void check_verify_inputs(Workspace& w, uint32_t B, const VerifyLay& l, bool keyed) {
    if (!validated) {
      check_subgroup(w, B, 0, l.pk + (keyed ? 1u : 0u));
      return;
    }
    if (!(validated & MP_VALIDATED_DECKS)) check_subgroup(w, B, l.deck, 2 * N);
    if (!(validated & MP_VALIDATED_SHUFFLED)) check_subgroup(w, B, l.shuf, 2 * N);
    const uint32_t rest = std::max(l.deck, l.shuf) + 2 * N;      // (decks first, then the proof's points up to pk)
    if (!(validated & MP_VALIDATED_PROOFS)) check_subgroup(w, B, rest, l.pk - rest);
    if (keyed) check_subgroup(w, B, l.pk, 1);
  }
  // wire-v1 decks in HBM -> one status word per deck: 0, or MP_ERR_BAD_ENCODING if some point is not canonical, not on the curve or
  // (curves with a cofactor) outside the prime-order subgroup -- whatever mp_set_subgroup_check says: this IS the validation
  Workspace dws;
  void validate_decks_dev(size_t count, const uint8_t* decks, int32_t* status) override {
    rt::Stream s = ctx->stream;
    const uint32_t B = (uint32_t)count;
    Workspace& w = dws;
    w.fw = G_::FW;
    w.ensure(B, 1, 2 * N, 8, 0, 0, nwin, 4, s, 0);
    rt::dzero(w.status.p, (size_t)w.Bpad * 4, s);
    LoadPointsArgs a{decks, w.P.p, w.status.p, w.Bpad, 2 * N, 0};
    MP_RUN(k_load_points, C, B, 2 * N, a);
    if (!Cofactor<C>::ONE) {
      SubgroupArgs sa{w.P.p, w.status.p, w.Bpad, 0};
      MP_RUN(k_subgroup_check, C, B, 2 * N, sa);
    }
    rt::d2d(status, w.status.p, (size_t)B * 4, s);
    rt::stream_sync(s);
  }
  DevBuf<uint32_t> vflag;      // [2] "some proof of the batch needs a closer look" (word 0: the caller's lane, word 1: the verify lane)
  uint32_t* flag_word(bool vlane, rt::Stream s) {
    if (!vflag.n) vflag.alloc(2, s);
    rt::dzero(vflag.p + (vlane ? 1 : 0), 4, s);
    return vflag.p + (vlane ? 1 : 0);
  }
  bool read_flag(bool vlane) {
    uint32_t flag = 0;
    rt::d2h(&flag, vflag.p + (vlane ? 1 : 0), 4, ctx->stream);
    rt::stream_sync(ctx->stream);
    return flag != 0;
  }
  uint32_t nwin = 0;
  FbGeom fbg{8, 32, 255};
  uint32_t init_seed[8];
  // staging for the host-buffer API
  DevBuf<uint8_t> io_a, io_b, io_c, io_d, io_e, io_f;
  DevBuf<int32_t> io_status;

  // ---------------------------------------------------------------- construction
  static bool wire_point_host(const uint8_t* p, Aff<C>& out) {
    // host-side use of the same MP_HD conversion (alignment: copy to an aligned temp)
    alignas(8) uint8_t tmp[G_::PB];
    memcpy(tmp, p, G_::PB);
    return wire_to_aff<C>(tmp, out);
  }

  void build_plans(PlanSet* set, bool keyed) {
    rt::Stream s = ctx->stream;
    for (int k = 0; k < N_PLANS; ++k) {
      PlanSet& q = set[k];
      const PlanParams& pp = pprm[k];
      // the finest split serves batches too small to fill the chip with one lane per Straus job: there the bucket kernel
      // (windows x 64 lanes per MSM) pays from ~500 terms on -- the merged equation of a 300-card proof (1 239 terms), no longer that of a
      // 52-card one (239: since the quad-lane Straus chains of round 3 and the kernel's rebuild in round 4, Straus verifies one proof in 1.4 ms,
      // the bucket kernel in 1.6, 64 proofs in 1.7 against 2.1; profiles/r05d_latency.txt).  The latency split had it too until the end of round 3: from ~1 000
      // proofs on the fixed 14-addition reduction per window is 3x the work of Straus (2 048 proofs: 164 k -> 184 k/s without it)
      const uint32_t bmin = (bucket_min && k == 3) ? std::min(bucket_min, BUCKET_MIN_SMALL_BATCH) : bucket_min;
      // Toom-Cook adds two dependent stages (operand evaluation, interpolation): a win when the batch fills the chip (throughput and
      // medium plans), a loss for a handful of proofs, where the small-batch plans keep Karatsuba (BLS12-377 (6,50), one proof: 78 vs 94 ms)
      const uint32_t pbits = bucket_bits_of(4 * N + 11 * m + 9);      // (the merged equation: the largest MSM a proof brings)
      q.pplan = make_prove_plan(m, n, pp.fch, pp.vch, G_::PB, keyed, bmin, bk_windows(R::BITS, pbits), toom_cook && (k == 0 || k == 2 || k == 4), pp.vsp, pbits);
      q.vplan = make_verify_plan(m, n, pp.fch, pp.vch, G_::PB, keyed, bmin, bk_windows(R::BITS, pbits), pp.vsp, pbits);
      q.table_group = pp.grp;
      q.norm_chunk = pp.nch;       // fewer points per serial inversion chain when lanes are idle
      for (int i = 0; i < 6; ++i) q.pph[i].upload(q.pplan.ph[i], s);
      q.vph.upload(q.vplan.ph, s);
      q.vmph.upload(q.vplan.mph, s);
      q.mjobs.upload(q.vplan.mjobs, s);
      q.mpairs.upload(q.vplan.mpairs, s);
      q.draws.upload(q.pplan.draws, s);
      q.lin.upload(q.pplan.lin, s);
      q.lin_src.upload(q.pplan.lin_src, s);
      q.lin_coef.upload(q.pplan.lin_coef, s);
      q.consts.upload(toom_constants(q.pplan.toom), s);
      q.pwire.upload(q.pplan.wire, s);
      q.vwire.upload(q.vplan.wire, s);
    }
  }
  // keyed batches: plans + the static lists that turn a proof's key into window tables (recode rho, tables of 2^(5w) pk)
  void ensure_keyed() {
    if (psk_ready) return;
    build_plans(psk, true);
    const ProveLay& l = psk[0].pplan.lay;
    // the key's digit / table slots live behind those of every phase of either plan
    key_d_first = key_t_first = 0;
    for (int k = 0; k < N_PLANS; ++k)
      for (int i = 0; i < 6; ++i) {
        key_d_first = std::max(key_d_first, psk[k].pplan.ph[i].n_dslots);
        key_t_first = std::max(key_t_first, psk[k].pplan.ph[i].n_tslots);
      }
    std::vector<Term> rec, tab;
    for (uint32_t i = 0; i < N; ++i) rec.push_back(Term{l.rho + i, key_d_first + i});
    for (uint32_t w = 0; w < nwin; ++w) tab.push_back(Term{l.kw + w, key_t_first + w});
    key_recode.upload(rec, ctx->stream);
    key_tables.upload(tab, ctx->stream);
    psk_ready = true;
  }

  int init(mp_ctx* c, uint32_t m_, uint32_t n_, const uint8_t* params, const uint8_t* pk, uint32_t fb_bits) {
    ctx = c;
    if (fb_bits != 8 && fb_bits != 16 && fb_bits != 20 && fb_bits != 21)
      return fail(MP_ERR_BAD_ARGUMENT, "fixed-base window width must be 8, 16, 20 or 21 bits");
    // windows cover the scalar field's bit length (252 bits on the STARK curve: 12 windows of 21 bits instead of 13 of 20)
    fbg = FbGeom{fb_bits, ((uint32_t)R::BITS + fb_bits - 1u) / fb_bits, (1u << fb_bits) - 1u};
    this->fb_bits = fb_bits;
    m = m_; n = n_; N = m * n;
    point_bytes = G_::PB;
    if (G_::FW > 8) bucket_split_bits = 11;      // (BLS12-377: the one-wave-per-window kernel spills on the 14-limb field, the split pipeline's hot loop does not)
    // plan thresholds count lanes, and a proof of N cards brings ~N/52 times the lanes of a 52-card proof
    set_latency_batch(std::max<size_t>(40, (size_t)2560 * 52 / N));
    nwin = (uint32_t)vb_windows(R::BITS);
    default_plan_params();
    FixedBases fb{n};
    std::vector<Aff<C>> bases(fb.count());
    bool ok = true;
    Aff<C> G, H, gen, pkp;
    ok &= wire_point_host(params, G);
    for (uint32_t j = 0; j < n; ++j) ok &= wire_point_host(params + G_::PB * (1 + j), bases[fb.ck(j)]);
    ok &= wire_point_host(params + G_::PB * (1 + n), H);
    ok &= wire_point_host(params + G_::PB * (2 + n), gen);
    ok &= wire_point_host(pk, pkp);
    if (!ok) return fail(MP_ERR_BAD_ENCODING, "parameters / shared key: bad point encoding");
    ok = aff_in_subgroup_host<C>(G) && aff_in_subgroup_host<C>(H) && aff_in_subgroup_host<C>(gen) && aff_in_subgroup_host<C>(pkp);
    for (uint32_t j = 0; j < n; ++j) ok = ok && aff_in_subgroup_host<C>(bases[fb.ck(j)]);
    if (!ok) return fail(MP_ERR_BAD_ENCODING, "parameters / shared key: a point is not in the prime-order subgroup");
    bases[fb.H()] = H; bases[fb.G()] = G; bases[fb.pk()] = pkp; bases[fb.gen()] = gen;
    Jac<C> gs = jac_inf<C>();
    for (uint32_t j = 0; j < n; ++j) gs = jac_madd<C>(gs, bases[fb.ck(j)]);
    if (jac_is_inf<C>(gs)) {
      bases[fb.gsum()] = aff_inf<C>();
    } else {
      bases[fb.gsum()] = jac_to_aff_with_zinv<C>(gs, fe_inv<F>(gs.Z));
    }
    for (uint32_t i = 0; i < fb.count(); ++i)
      if (i != fb.gsum() && aff_is_inf<C>(bases[i])) return fail(MP_ERR_BAD_ENCODING, "parameters: a base is the point at infinity");

    rt::Stream s = ctx->stream;
    std::vector<uint32_t> flat(fb.count() * G_::PW);
    for (uint32_t i = 0; i < fb.count(); ++i) {
      fe_pack<F>(bases[i].x, &flat[i * G_::PW]);
      fe_pack<F>(bases[i].y, &flat[i * G_::PW + G_::FW]);
    }
    fbpts.upload(flat, s);
    build_fixed_tables(fb.count());

    build_plans(ps, false);
    // Blake2s("Shuffle Proof")  [REF mod.rs:84]
    {
      Blake2sState st;
      blake2s_init(st);
      uint32_t mblk[16] = {0};
      memcpy(mblk, "Shuffle Proof", 13);
      blake2s_compress(st, mblk, 13, true);
      memcpy(init_seed, st.h, 32);
    }
    rt::stream_sync(s);
    return MP_OK;
  }

  void normalize_flat(const uint32_t* src, uint32_t* dst, uint32_t* scratch, size_t count) {
    if (!count) return;
    const uint32_t ch = cur_norm_chunk;
    NormArgs a{src, dst, scratch, (uint32_t)count, (uint32_t)((count + ch - 1) / ch), ch};
    MP_RUN(k_normalize, C, a.nthreads, 1, a);
  }

  // tables with narrow windows, built by chains: out[base][gh.windows][gh.entries] affine (window widths alternate gh.bits / h2)
  void build_narrow_tables(const uint32_t* pts, uint32_t nb, const FbGeom& gh, uint32_t h2, uint32_t* out) {
    rt::Stream s = ctx->stream;
    DevBuf<uint32_t> WJ, W, EJ, scratch;
    const size_t nwinpts = (size_t)nb * gh.windows, nent = nwinpts * gh.entries;
    WJ.alloc(nwinpts * G_::JW, s);
    W.alloc(nwinpts * G_::PW, s);
    EJ.alloc(nent * G_::JW, s);
    scratch.alloc(nent * G_::FW, s);
    FbWinArgs wa{pts, WJ.p, gh, h2};
    MP_RUN(k_fb_windows, C, nb, 1, wa);
    normalize_flat(WJ.p, W.p, scratch.p, nwinpts);
    FbFillArgs fa{W.p, EJ.p, gh};
    MP_RUN(k_fb_fill, C, (uint32_t)nwinpts, 1, fa);
    normalize_flat(EJ.p, out, scratch.p, nent);
    rt::stream_sync(s);
  }
  void build_fixed_tables(uint32_t nb) {
    rt::Stream s = ctx->stream;
    // narrow table first (h-bit windows: h = 8 for 8- and 16-bit tables, 10 for 20-bit tables)
    // (a 21-bit window splits 11 + 10: the narrow windows alternate between the two widths, all of them with 2^h - 1 entries)
    const uint32_t h = fbg.bits == 8 ? 8u : (fbg.bits + 1u) / 2u, h2 = fbg.bits == 8 ? 8u : fbg.bits - h;
    const FbGeom gh{h, fbg.bits == 8 ? fbg.windows : 2u * fbg.windows, (1u << h) - 1u};
    DevBuf<uint32_t> Th;
    Th.alloc((size_t)nb * gh.windows * gh.entries * G_::PW, s);
    build_narrow_tables(fbpts.p, nb, gh, h2, Th.p);
    if (fbg.bits == h) {
      std::swap(FB.p, Th.p);
      std::swap(FB.n, Th.n);
      return;
    }
    // widen: windows of 2h bits; every entry is one affine + affine addition of two narrow entries
    // (16-bit: 2 GB at n = 26; 20-bit: 27 GB -- sized for a 288 GB part)
    const size_t nentw = (size_t)nb * fbg.windows * fbg.entries;
    if (nentw >= ((size_t)1 << 32)) throw std::runtime_error("fixed-base table too large for one launch: use narrower windows");
    DevBuf<uint32_t> EJw, scratchw;
    EJw.alloc(nentw * G_::JW, s, false);
    scratchw.alloc(nentw * G_::FW, s, false);
    FB.alloc(nentw * G_::PW, s, false);
    FbWidenArgs ww{Th.p, EJw.p, gh, fbg};
    MP_RUN(k_fb_widen, C, (uint32_t)nentw, 1, ww);
    const size_t per = (size_t)1 << 26;                     // normalise in slices of 64 M points
    for (size_t off = 0; off < nentw; off += per) {
      const size_t cnt = std::min(per, nentw - off);
      normalize_flat(EJw.p + off * G_::JW, FB.p + off * G_::PW, scratchw.p + off * G_::FW, cnt);
    }
    rt::stream_sync(s);
  }

  // ---- key sets: 8-bit fixed-base tables of K aggregate keys (32 windows x 255 entries = 8 160 points per key: 0.5 MB on the
  // 256-bit curves), built in slices of 2 048 keys so that the Jacobian intermediates stay below 2 GB
  int keyset_build(mp_keyset& ks, size_t K, const uint8_t* keys_host) override {
    rt::Stream s = ctx->stream;
    const FbGeom g8{8u, ((uint32_t)R::BITS + 7u) / 8u, 255u};
    if (K == 0 || K * (size_t)g8.windows * g8.entries >= ((size_t)1 << 32)) return fail(MP_ERR_BAD_ARGUMENT, "mp_keyset_create: 1 .. 500 000 keys");
    std::vector<uint32_t> flat(K * G_::PW), wire(K * (G_::PB / 4));
    for (size_t i = 0; i < K; ++i) {
      Aff<C> p;
      if (!wire_point_host(keys_host + i * G_::PB, p) || aff_is_inf<C>(p) || !aff_in_subgroup_host<C>(p))
        return fail(MP_ERR_BAD_ENCODING, "mp_keyset_create: key " + std::to_string(i) + " is not a point of the prime-order group");
      fe_pack<F>(p.x, &flat[i * G_::PW]);
      fe_pack<F>(p.y, &flat[i * G_::PW + G_::FW]);
    }
    memcpy(wire.data(), keys_host, K * G_::PB);
    DevBuf<uint32_t> pts;
    pts.upload(flat, s);
    ks.wire.upload(wire, s);
    ks.K = K;
    ks.bits = g8.bits; ks.windows = g8.windows; ks.entries = g8.entries;
    const size_t per_key = (size_t)g8.windows * g8.entries * G_::PW;
    ks.FB.alloc(K * per_key, s, false);
    const size_t slice = 2048;
    for (size_t k0 = 0; k0 < K; k0 += slice) {
      const uint32_t kc = (uint32_t)std::min(slice, K - k0);
      build_narrow_tables(pts.p + k0 * G_::PW, kc, g8, 8u, ks.FB.p + k0 * per_key);
    }
    rt::stream_sync(s);
    return MP_OK;
  }
  DevBuf<uint32_t> ks_keys_main, ks_keys_vlane;      // the wire keys of a batch, gathered from a key set (one buffer per lane)
  const uint8_t* gather_keys(uint32_t B, const mp_keyset* ks, const uint32_t* kidx, int32_t* status, bool vlane = false) {
    DevBuf<uint32_t>& ks_keys = vlane ? ks_keys_vlane : ks_keys_main;
    ks_keys.alloc((size_t)B * (G_::PB / 4), ctx->stream, false);
    GatherKeysArgs ga{ks->wire.p, kidx, ks_keys.p, status, (uint32_t)ks->K};
    MP_RUN(k_gather_keys, C, B, G_::PB / 4, ga);
    return reinterpret_cast<const uint8_t*>(ks_keys.p);
  }

  uint32_t stage_words_needed() const {
    const size_t tb = G_::PB + 1;     // ark ToBytes of a point: x || y || flag
    size_t bytes = (size_t)(3 + n + 1 + 4 * N) * tb + 16 + 32;
    bytes = std::max(bytes, (size_t)(1 + 6 * m) * tb + 32);
    return (uint32_t)(bytes / 4 + 4);
  }

  void reserve(size_t B) override { reserve_for(B, false); }
  void reserve_for(size_t B, bool keyed) { reserve_ws(ws, B, keyed); }
  void reserve_ws(Workspace& ws, size_t B, bool keyed) {
    if (keyed) ensure_keyed();
    PlanSet& q = pick((uint32_t)B, keyed);
    uint32_t nS = std::max(q.pplan.lay.nS, q.vplan.lay.nS), nP = std::max(q.pplan.lay.nP, q.vplan.lay.nP);
    uint32_t nJ = std::max(q.pplan.nJ, q.vplan.nJ), nD = std::max(q.vph.n_dslots, q.vmph.n_dslots),
             nT = std::max(q.vph.n_tslots, q.vmph.n_tslots);
    uint32_t d8 = std::max(q.vph.b_dig_bytes, q.vmph.b_dig_bytes);
    for (int i = 0; i < 6; ++i) {
      nD = std::max(nD, q.pph[i].n_dslots);
      nT = std::max(nT, q.pph[i].n_tslots);
      d8 = std::max(d8, q.pph[i].b_dig_bytes);
    }
    if (keyed) {
      nD = std::max(nD, key_d_first + N);
      nT = std::max(nT, key_t_first + nwin);
    }
    ws.fw = G_::FW;
    // NS2 = the main stream's inversion scratch while `side` has NS (prove_dev: only c_A's m points are normalised there, and only
    // batches up to overlap_max fork); the verify lane's, chain and ad-hoc workspaces never need it
    ws.ensure((uint32_t)B, nS, nP, nJ, nD, nT, nwin, stage_words_needed(), ctx->stream, d8);
    if (&ws == &this->ws && overlap_max && B <= overlap_max) ws.NS2.alloc((size_t)m * ws.Bpad * G_::FW, ctx->stream, false);
  }

  // ---------------------------------------------------------------- one dependency level of group work
  // parts: which of (recode, window tables, MSMs + combines, normalisations) to run; norm_only / norm_skip: the normalisation range
  // that starts at this slot alone / every range but it; scratch: the inversion scratch to use (default: w.NS)
  enum : uint32_t { PH_RECODE = 1, PH_TABLES = 2, PH_MSM = 4, PH_NORM = 8, PH_ALL = 15 };
  void run_phase(PhaseDev& ph, Workspace& w, uint32_t B, uint32_t parts = PH_ALL, uint32_t norm_only = NO_SLOT, uint32_t norm_skip = NO_SLOT,
                 uint32_t* scratch = nullptr) {
    if (!scratch) scratch = w.NS.p;
    if (parts & PH_TABLES) run_tables(ph, w, B, scratch);
    if (parts & PH_RECODE) run_recode(ph, w, B);
    if (parts & PH_MSM) run_msms(ph, w, B);
    if (parts & PH_NORM) run_normalize(ph, w, B, norm_only, norm_skip, scratch);
  }
  // kernels_quad.hpp: up to this many lanes' worth of (proof, job) pairs run their group operations on four lanes each
#ifdef MP_EXP_QUAD_MAX      // experiment hook (tools/ab_build.py): 0 = never
  uint32_t quad_max_lanes = MP_EXP_QUAD_MAX;
#else
  uint32_t quad_max_lanes = 65536;       // (per launch; 262 144 was tried: the many short chains of a 4 096-proof batch -- combines, fixed-base sums -- then go four lanes wide too: 239 k -> 161 k/s)
#endif
  // (the quad kernels index their items -- (proof, job) pairs -- with 32 bits: a forced mp_set_group_lanes(t, 4) on a launch with more
  // items than that falls back to one lane per chain)
  bool quad_ops(uint32_t B, uint32_t njobs) const {
    const uint64_t items = (uint64_t)B * njobs;
    if (items > 0xFFFFFFF0ull) return false;
    return group_lanes == 4 || (group_lanes == 0 && items * 4u <= quad_max_lanes);
  }
  static uint32_t quad_waves(uint32_t B, uint32_t njobs) { return (uint32_t)(((uint64_t)B * njobs + 15u) / 16u); }
  void run_combine(const CombineArgs& a, uint32_t B, uint32_t njobs) {
    if (quad_ops(B, njobs)) {
      CombineQuadArgs qa{a, B, njobs};
      MP_WAVE_RUN(k_combine_q, C, quad_waves(B, njobs), 0, qa);
    } else {
      MP_RUN(k_combine, C, B, njobs, a);
    }
  }
  void run_recode(PhaseDev& ph, Workspace& w, uint32_t B) {
    if (ph.n_recode) {
      RecodeArgs a{w.S.p, w.D.p, ph.recode.p, w.Bpad, nwin};
      MP_RUN(k_recode, C, B, ph.n_recode, a);
    }
  }
  void run_tables(PhaseDev& ph, Workspace& w, uint32_t B, uint32_t* scratch) {
    if (ph.n_tables) {
      TableArgs a{w.P.p, w.T.p, scratch, ph.tables.p, w.Bpad, ph.n_tables, cur_table_group};
      MP_RUN(k_table, C, B, (ph.n_tables + cur_table_group - 1) / cur_table_group, a);
    }
  }

  // the bucket method over a phase's large MSMs: digits, the persistent wave kernel (one wave per (equation, MSM, window) at a time),
  // the fold of the window results.  `count` equations -- proofs, or chain / group equations -- whose scalars lie in S with lane stride
  // sstride and whose digits go to D (dstride per equation)
  uint32_t bk_pass_eqs = 0;           // the split pipeline's equations per pass and what they were sized for (run_bucket)
  size_t bk_pass_bytes = 0;
  uint32_t bucket_slots() {
    if (!ctx->bk_slots) ctx->bk_slots = BK_WAVES_PER_CU * rt::cu_count();
    return ctx->bk_slots;
  }
  // tile / tile_K: group verification -- the points of equation e as one contiguous run (k_group_tile), terms carry their index in it
  void run_bucket(Workspace& w, PhaseDev& ph, const uint32_t* S, uint32_t sstride, int16_t* D, size_t dstride, uint32_t count,
                  uint32_t link_stride, const char* too_large, const uint32_t* tile = nullptr, uint32_t tile_K = 0) {
    rt::Stream s = ctx->stream;
    const uint32_t c = ph.b_bits, bw = bk_windows(R::BITS, c);
    // (the item counter runs past the last item by one draw per persistent wave)
    if ((uint64_t)count * ph.n_bterms >= ((uint64_t)1 << 32) || (uint64_t)count * ph.n_b * bw + 8ull * bucket_slots() >= ((uint64_t)1 << 32)) throw std::runtime_error(too_large);
    BRecodeArgs ra{S, D, ph.bterms.p, ph.bpos.p, sstride, bw, ph.n_bterms, dstride, c};
    MP_RUN(k_bucket_recode, C, count * ph.n_bterms, 1, ra);
    if (c >= bucket_split_bits) {
      // windows of 12 bits and more: sort / additions / reduction as three kernels (kernels_bucket.hpp, round 6).  Their scratch
      // (sorted runs, offsets, parked sums: ~1.7 MB per (equation, window) at 13 bits and 243 712 points) holds all items of a PASS of
      // equations: as many as fit a quarter of the free memory, 12 GB at the most
      const uint32_t NBK = bk_buckets(c), XW = XyzzWords<C>::N, gmax = bk_chunks(ph.b_kpad_max), units = bk_units(c);
      if (gmax > BK_CHUNKS_MAX) throw std::runtime_error("bucket method: more than 589 824 terms in one multi-scalar multiplication with windows of 12 bits or more");
      const size_t per_item = ((size_t)ph.b_kpad_max + (size_t)NBK * XW) * 4 + (size_t)gmax * bk_offs_row(c) * 2, per_eq = per_item * ph.n_b * bw;
      if (!bk_pass_eqs || bk_pass_bytes != per_eq) {
        size_t free_b = 0, total_b = 0;
        rt::mem_info(&free_b, &total_b);
        // (what the scratch already holds counts as free; the emulator reports no memory: one gigabyte)
        const size_t have = (w.bk_sorted.n + w.bk_park.n) * 4 + w.bk_offs.n * 2;
        const size_t budget = total_b ? std::min<size_t>((size_t)12 << 30, (free_b + have) / 4) : (size_t)1 << 30;
        bk_pass_eqs = (uint32_t)std::max<size_t>(1, std::min<size_t>(budget / per_eq, 0x7FFFFFFFu));
        bk_pass_bytes = per_eq;
      }
      const uint32_t pass = std::min(count, std::max(bk_pass_eqs, 1u));
      const size_t items_max = (size_t)pass * ph.n_b * bw;
      if (items_max * units >= ((uint64_t)1 << 32)) throw std::runtime_error(too_large);
      w.bk_sorted.alloc(items_max * ph.b_kpad_max, s, false);
      w.bk_offs.alloc(items_max * gmax * bk_offs_row(c), s, false);
      w.bk_park.alloc(items_max * NBK * XW, s, false);
      w.bk_quarters.alloc(items_max * 8 * XW, s, false);
      for (uint32_t e0 = 0; e0 < count; e0 += pass) {
        const uint32_t ne = std::min(pass, count - e0), items = ne * ph.n_b * bw;
        const uint32_t acc_lds = bk_acc_lds_words(c, gmax, XW, G_::PW), wpb = rt::waves_per_block(acc_lds), wgs = (items * units + wpb - 1) / wpb;
        BSplitArgs sa{D, w.P.p, w.J.p, ph.bjobs.p, ph.bterms.p, w.Bpad, bw, ph.n_b, dstride, link_stride, c, e0, ne, tile, tile_K,
                      w.bk_sorted.p, w.bk_offs.p, w.bk_park.p, ph.b_kpad_max, gmax, units, wpb, wgs, w.bk_quarters.p};
        ctx->prof.begin("k_bucket_sort", s, (uint64_t)items * gmax);
        MP_BLOCK_LAUNCH(k_bucket_sort, C, s, items * gmax, bk_sort_lds_words(c), sa);
        ctx->prof.end(s);
        MP_WAVE_RUN(k_bucket_acc, C, (wgs + 7u) / 8u * 8u * wpb, acc_lds, sa);      // (8 XCDs x their share of the workgroups: bk_unit_of_wave)
        MP_WAVE_RUN(k_bucket_list, C, (wgs + 7u) / 8u * 8u * wpb, acc_lds, sa);     // (the windows whose digits crowd into a few buckets: nothing to do for the others)
        MP_WAVE_RUN(k_bucket_reduce, C, items * 4u, 64u * XW, sa);
        MP_RUN(k_bucket_final, C, items, 1, sa);
      }
      BFoldArgs fa{w.J.p, ph.bjobs.p, w.Bpad, bw, 0u, c};
      if (quad_ops(count, ph.n_b)) {
        BFoldQuadArgs qa{fa, count, ph.n_b};
        MP_WAVE_RUN(k_bucket_fold_q, C, quad_waves(count, ph.n_b), 0, qa);
      } else {
        MP_RUN(k_bucket_fold, C, count, ph.n_b, fa);
      }
      return;
    }
    const uint32_t nitems = count * ph.n_b * bw, nslots = std::min(nitems, bucket_slots());
    w.ensure_bucket(bucket_slots(), ph.b_kpad_max, c, XyzzWords<C>::N, s);
    rt::dzero(w.bk_counter.p, 8 * 4, s);
    BucketArgs ba{D, w.P.p, w.J.p, ph.bjobs.p, ph.bterms.p, w.Bpad, bw, ph.n_b, dstride, link_stride, c, nitems, nslots, w.bk_counter.p,
                  count, bk_xcd_affine && count >= 8u ? 8u : 1u, tile, tile_K, bk_stage ? 1u : 0u,
                  w.bk_sorted.p, w.bk_park.p, ph.b_kpad_max, nullptr};
#ifdef MP_EXP_BK_TIMING
    w.bk_timing.alloc((size_t)bucket_slots() * 8, s);
    ba.timing = w.bk_timing.p;
#endif
    ctx->prof.begin("k_bucket_msm", s, nitems);
    MP_WAVE_LAUNCH(k_bucket_msm, C, s, nslots, bk_lds_words(c, XyzzWords<C>::N), ba);
    ctx->prof.end(s);
#ifdef MP_EXP_BK_TIMING     // experiment (tools/ab_build.py --units=curve_stark_msm.hip,curve_stark.hip): mean cycles per phase and wave
    {
      std::vector<unsigned long long> tm((size_t)nslots * 4);
      rt::d2h(tm.data(), w.bk_timing.p, tm.size() * 8, s);
      rt::stream_sync(s);
      double sum[4] = {0, 0, 0, 0};
      for (size_t i = 0; i < tm.size(); ++i) sum[i & 3] += (double)tm[i];
      fprintf(stderr, "k_bucket_msm items %u slots %u bits %u kpad %u: cycles/wave sort %.0f ranks %.0f additions %.0f reduction %.0f\n", nitems, nslots, c,
              ph.b_kpad_max, sum[0] / nslots, sum[1] / nslots, sum[2] / nslots, sum[3] / nslots);
    }
#endif
    BFoldArgs fa{w.J.p, ph.bjobs.p, w.Bpad, bw, 0u, c};
    if (quad_ops(count, ph.n_b)) {          // few equations: the fold's ~256 dependent doublings on four lanes each
      BFoldQuadArgs qa{fa, count, ph.n_b};
      MP_WAVE_RUN(k_bucket_fold_q, C, quad_waves(count, ph.n_b), 0, qa);
    } else {
      MP_RUN(k_bucket_fold, C, count, ph.n_b, fa);
    }
  }
  void run_msms(PhaseDev& ph, Workspace& w, uint32_t B) {
    if (ph.n_f) {
      FixedArgs a{w.S.p, w.J.p, FB.p, ph.fjobs.p, ph.fterms.p, w.Bpad, fbg, w.Bpad};
      if (quad_ops(B, ph.n_f)) {
        FixedQuadArgs qa{a, B, ph.n_f};
        MP_WAVE_RUN(k_fixed_msm_q, C, quad_waves(B, ph.n_f), 0, qa);
      } else {
        MP_RUN(k_fixed_msm, C, B, ph.n_f, a);
      }
    }
    if (ph.n_v) {
      VarArgs a{w.D.p, w.T.p, w.J.p, ph.vjobs.p, ph.vterms.p, w.Bpad, nwin, ph.vsplit};
      const uint32_t nvl = ph.n_v * ph.vsplit;      // lanes per proof: (job, window range) pairs
#ifdef MP_EXP_VAR_LDS      // experiment hook (tools/ab_build.py): unused dynamic LDS per workgroup caps the waves per SIMD of k_var_msm
      ctx->prof.begin("k_var_msm", ctx->stream);
      hipLaunchKernelGGL((k_var_msm<C>), dim3((B + 255u) / 256u, nvl), dim3(256), MP_EXP_VAR_LDS, ctx->stream, a, (uint32_t)B);
      ctx->prof.end(ctx->stream);
#else
      if (quad_ops(B, nvl)) {      // a handful of proofs: four lanes per group operation, 3-4 products deep instead of 10
        VarQuadArgs qa{a, B, nvl};
        MP_WAVE_RUN(k_var_msm_q, C, quad_waves(B, nvl), 0, qa);
      } else {
        MP_RUN(k_var_msm, C, B, nvl, a);
      }
#endif
      if (ph.n_wc) {      // window-split jobs: the range sums of an MSM's sub-jobs, range by range ...
        CombineArgs ca{w.J.p, w.P.p, ph.wcjobs.p, ph.wcterms.p, w.Bpad};
        run_combine(ca, B, ph.n_wc);
      }
      if (ph.n_w) {       // ... and the fold R = sum_r 2^(5 lo(r)) S_r, once per MSM
        BFoldArgs fa{w.J.p, ph.wjobs.p, w.Bpad, 0u, nwin, 0u};
        if (quad_ops(B, ph.n_w)) {
          BFoldQuadArgs qa{fa, B, ph.n_w};
          ctx->prof.begin("k_wfold_q", ctx->stream);
          MP_WAVE_LAUNCH(k_bucket_fold_q, C, ctx->stream, quad_waves(B, ph.n_w), 0, qa);
          ctx->prof.end(ctx->stream);
        } else {
          ctx->prof.begin("k_wfold", ctx->stream);
          MP_LAUNCH(k_bucket_fold, C, ctx->stream, B, ph.n_w, fa);
          ctx->prof.end(ctx->stream);
        }
      }
    }
    if (ph.n_b) run_bucket(w, ph, w.S.p, w.Bpad, w.D16.p, (size_t)w.d8_bytes, B, 0u, "bucket recode: batch too large for one launch");
    if (ph.n_c0) {   // group sums of MSMs with many partials
      CombineArgs a{w.J.p, w.P.p, ph.cjobs0.p, ph.cterms0.p, w.Bpad};
      run_combine(a, B, ph.n_c0);
    }
    if (ph.n_c) {
      CombineArgs a{w.J.p, w.P.p, ph.cjobs.p, ph.cterms.p, w.Bpad};
      run_combine(a, B, ph.n_c);
    }
    if (ph.n_c2) {   // second stage: consumers of first-stage combine outputs
      CombineArgs a{w.J.p, w.P.p, ph.cjobs2.p, ph.cterms2.p, w.Bpad};
      run_combine(a, B, ph.n_c2);
    }
  }
  void run_normalize(PhaseDev& ph, Workspace& w, uint32_t B, uint32_t norm_only, uint32_t norm_skip, uint32_t* scratch) {
    std::vector<std::pair<uint32_t, uint32_t>> ranges;
    for (auto& r : ph.normalize)
      if ((norm_only == NO_SLOT || r.first == norm_only) && r.first != norm_skip) ranges.push_back(r);
    // the ranges of a phase in as few launches as possible (NORM_MAX_RANGES per launch; element offsets must fit 32 bits)
    for (size_t i = 0; i < ranges.size();) {
      const size_t nr = std::min<size_t>(NORM_MAX_RANGES, ranges.size() - i);
      size_t last_elem = 0;
      for (size_t k = 0; k < nr; ++k)
        last_elem = std::max(last_elem, ((size_t)ranges[i + k].first + ranges[i + k].second) * w.Bpad);
      if (nr == 1 || last_elem >= ((size_t)1 << 32)) {
        auto& r = ranges[i];
        normalize_flat(w.J.p + j_off<C>(r.first, w.Bpad, 0), w.P.p + p_off<C>(r.first, w.Bpad, 0), scratch, (size_t)r.second * w.Bpad);
        ++i;
        continue;
      }
      NormMultiArgs a{};
      a.J = w.J.p; a.P = w.P.p; a.scratch = scratch;
      a.nr = (uint32_t)nr; a.chunk = cur_norm_chunk;
      uint32_t threads = 0, selem = 0;
      for (size_t k = 0; k < nr; ++k) {
        auto& r = ranges[i + k];
        const uint32_t cnt = r.second * w.Bpad;
        a.first[k] = r.first * w.Bpad;
        a.count[k] = cnt;
        a.tstart[k] = threads;
        a.sstart[k] = selem;
        threads += (cnt + cur_norm_chunk - 1) / cur_norm_chunk;
        selem += cnt;
      }
      a.tstart[nr] = threads;
      MP_RUN(k_normalize_multi, C, threads, 1, a);
      i += nr;
    }
  }

  FsStatementArgs statement_args(Workspace& w, uint32_t p_deck, uint32_t p_shuf, uint32_t p_cA, uint32_t s_x, uint32_t p_pk = NO_SLOT) {
    FsStatementArgs a{};
    a.f = FsDev{w.stage.p, w.seed.p, w.Bpad};
    a.S = w.S.p;
    a.P = w.P.p;
    a.fbpts = fbpts.p;
    memcpy(a.init_seed, init_seed, 32);
    a.m = m; a.n = n; a.N = N;
    a.p_deck = p_deck; a.p_shuf = p_shuf; a.p_cA = p_cA; a.s_x = s_x;
    a.p_pk = p_pk;
    a.W = nullptr;
    a.w_deck = a.w_shuf = NO_SLOT;
    return a;
  }
  // The transcript kernels: one lane per proof, or four lanes per hash (k_fsq_*: the lanes of a wave dealt to 64 / lpp proofs)
  // for batches that would leave the chip to a few hundred lone lanes.  Crossover: a lone wave issues every 8-10 cycles, four
  // waves per SIMD every ~4.3, and the quad kernels run ~2.7x fewer instructions per hash on 16x more waves.
  FsqGeom fsq_geom(uint32_t B) const {
    uint32_t lpp = 64;
    while (lpp > 4 && (uint64_t)B * lpp > 65536u) lpp >>= 1;
    return FsqGeom{lpp, B};
  }
  static uint32_t fsq_waves(const FsqGeom& g) { return (g.B + 64u / g.lpp - 1) / (64u / g.lpp); }
  bool fs_quad(uint32_t B) const { return fs_lanes == 4 || (fs_lanes == 0 && B <= FSQ_MAX_BATCH); }
  void run_fs_round1(const FsStatementArgs& a, uint32_t B) {
    if (fs_quad(B)) {
      FsqStatementArgs q{a, fsq_geom(B)};
      MP_WAVE_RUN(k_fsq_round1, C, fsq_waves(q.g), 0, q);
    } else {
      MP_RUN(k_fs_round1, C, B, 1, a);
    }
  }
  void run_fs_round(const FsRoundArgs& a, uint32_t B) {
    if (fs_quad(B)) {
      FsqRoundArgs q{a, fsq_geom(B)};
      MP_WAVE_RUN(k_fsq_round, C, fsq_waves(q.g), 0, q);
    } else {
      MP_RUN(k_fs_round, C, B, 1, a);
    }
  }
  void run_verify_fs(const VerifyFsArgs& a, uint32_t B) {
    if (fs_quad(B)) {
      FsqVerifyArgs q{a, fsq_geom(B)};
      MP_WAVE_RUN(k_fsq_verify, C, fsq_waves(q.g), 0, q);
    } else {
      MP_RUN(k_verify_fs, C, B, 1, a);
    }
  }
  // room for the wire words of `decks` decks per proof (their transcript bytes, kept by k_load_points) -- for the small-batch
  // plans only: there the transcript lane is what a proof waits for (a 300-card BLS12-377 verification: 24 -> 19 ms, a 52-card one
  // 3.5 -> 3.3 ms), while a full batch hides that lane behind thousands of others and would only pay for the extra 64 bytes
  // written per point (-0.3 % on the default bench, A/B)
  uint32_t* wire_words(Workspace& w, uint32_t B, uint32_t decks) {
    const int plan = plan_of(B);
    if (plan != 1 && plan != 3 && plan != 5) return nullptr;
    w.W.alloc((size_t)decks * 2 * N * (G_::PB / 4) * w.Bpad, ctx->stream, false);
    return w.W.p;
  }

  void prove_init(const ProveInitArgs& ia, uint32_t B) {
    if (B <= PROVE_INIT_WAVE_MAX)      // a wave per proof: 64 ChaCha20 blocks at a time instead of one lane's ~300 in a row
      MP_WAVE_RUN(k_prove_init_w, C, B, N, ia);
    else
      MP_RUN(k_prove_init, C, B, 1, ia);
  }
  // the challenge-independent group work behind the re-encryption (after the shuffled deck is normalised): operand sums of its
  // rows (m = 2 Toom-Cook / Karatsuba), their Toom-Cook evaluations (3 <= m <= 16); with_tables: also the window tables of level B
#ifdef MP_EXP_OVERLAP_MAX      // experiment hook (tools/ab_build.py): 0 = never
  uint32_t overlap_max = MP_EXP_OVERLAP_MAX;
#else
  uint32_t overlap_max = OVERLAP_MAX_BATCH;
#endif
  // work forked onto ctx->side: until the main stream has been made to wait for it (joined = true), an exception on the way out
  // must not leave kernels of this batch running on the shared arenas behind the caller's back
  struct SideGuard {
    mp_ctx* c;
    bool joined = false;
    ~SideGuard() {
      if (!joined) {
        try {
          rt::stream_sync(c->side);
        } catch (...) {
        }
      }
    }
  };
  void prove_side_work(PlanSet& q, Workspace& w, uint32_t B, bool with_tables) {
    const ProveLay& l = q.pplan.lay;
    run_phase(q.pph[4], w, B);      // Toom-Cook (m = 2) / Karatsuba operand sums (empty when unused)
    const ToomPlan& tk = q.pplan.toom;
    if (tk.E) {                   // Toom-Cook, 3 <= m <= 16: the ciphertext polynomial at +-1 .. +-(m-1)
      ToomPointsArgs ta{w.P.p, w.J.p, w.Bpad, m, n, l.shuf, tk.cv_first};
      MP_RUN(k_toom_points, C, B, 2 * n, ta);
      normalize_flat(w.J.p + j_off<C>(tk.cv_first, w.Bpad, 0), w.P.p + p_off<C>(tk.cv_first, w.Bpad, 0), w.NS.p,
                     (size_t)(tk.E - 2) * 2 * n * w.Bpad);
    }
    if (with_tables) run_phase(q.pph[1], w, B, PH_TABLES);
  }

  // ---------------------------------------------------------------- prove
  // keys != nullptr: keyed batch -- proof b is made under the aggregate key keys[b] (one wire point each) instead of the
  // table's own key [REF mod.rs:380-418 takes shared_key per call; tables of different card tables differ in nothing else]
  void prove_dev(size_t B_, const uint8_t* decks, const uint8_t* rho, const uint32_t* perm, const uint8_t* seeds,
                 uint8_t* out_decks, uint8_t* out_proofs, int32_t* status, const uint8_t* keys, const mp_keyset* kset,
                 const uint32_t* kidx) override {
    const uint32_t B = (uint32_t)B_;
    const bool keyed = keys != nullptr || kset != nullptr;
    reserve_for(B, keyed);
    Workspace& w = ws;
    PlanSet& q = pick(B, keyed);
    cur_table_group = q.table_group;
    cur_norm_chunk = q.norm_chunk;
    const ProveLay& l = q.pplan.lay;
    PhaseDev* pph = q.pph;
    rt::Stream s = ctx->stream;
    FixedBases fb{n};
    const bool overlap = overlap_max && B <= overlap_max;      // two streams for the stretch before the first challenge
    SideGuard side_guard{ctx, !overlap};
    rt::dzero(w.status.p, (size_t)w.Bpad * 4, s);
    {
      uint32_t* const ww = wire_words(w, B, 1);
      LoadPointsArgs a{decks, w.P.p, w.status.p, w.Bpad, 2 * N, l.deck, ww, 0};
      MP_RUN(k_load_points, C, B, 2 * N, a);
      LoadScalarsArgs sa{rho, w.S.p, w.status.p, w.Bpad, N, l.rho};
      MP_RUN(k_load_scalars, C, B, N, sa);
      ProveInitArgs ia{w.S.p, w.status.p, perm, seeds, q.draws.p, l, w.Bpad};
      if (!overlap) prove_init(ia, B);
      RemaskArgs ra{w.S.p, w.P.p, w.J.p, FB.p, perm, w.Bpad, N, l.rho, l.deck, l.shuf, fb.G(), fb.pk(), fbg,
                    0, w.D.p, w.T.p, key_d_first, key_t_first, nwin, nullptr, nullptr, FbGeom{8, 32, 255}, 0};
      if (!(validated & MP_VALIDATED_DECKS)) check_subgroup(w, B, l.deck, 2 * N);
      if (kset) {
        // the proof's key is a member of a key set: its multiples come from the set's tables, the key itself (transcript, its
        // own terms in the argument) from the set's copy of the wire bytes (validated when the set was built)
        keys = gather_keys(B, kset, kidx, w.status.p);
        LoadPointsArgs ka{keys, w.P.p, w.status.p, w.Bpad, 1, l.pk};
        MP_RUN(k_load_points, C, B, 1, ka);
        ra.keyed = 2;
        ra.KFB = kset->FB.p;
        ra.kidx = kidx;
        ra.kg = FbGeom{kset->bits, kset->windows, kset->entries};
        ra.nkeys = (uint32_t)kset->K;
      } else if (keyed) {
        // the proof's key -> window bases 2^(5w) pk -> their 16-entry tables; signed digits of the masking factors
        LoadPointsArgs ka{keys, w.P.p, w.status.p, w.Bpad, 1, l.pk};
        MP_RUN(k_load_points, C, B, 1, ka);
        check_subgroup(w, B, l.pk, 1);
        KeyWinArgs kw{w.P.p, w.J.p, w.Bpad, l.pk, l.kw, nwin};
        MP_RUN(k_key_windows, C, B, 1, kw);
        normalize_flat(w.J.p + j_off<C>(l.kw, w.Bpad, 0), w.P.p + p_off<C>(l.kw, w.Bpad, 0), w.NS.p, (size_t)nwin * w.Bpad);
        RecodeArgs rc{w.S.p, w.D.p, key_recode.p, w.Bpad, nwin};
        MP_RUN(k_recode, C, B, N, rc);
        TableArgs ta{w.P.p, w.T.p, w.NS.p, key_tables.p, w.Bpad, nwin, cur_table_group};
        MP_RUN(k_table, C, B, (nwin + cur_table_group - 1) / cur_table_group, ta);
        ra.keyed = 1;
      }
      // Everything up to the first challenge falls into two independent halves.  On `side`: the re-encryption, the shuffled deck
      // in affine form, the operand sums / Toom-Cook evaluations of its rows and the window tables of all of them -- none of it
      // depends on the prover's randomness or on a challenge.  On the main stream: the randomness, c_A, then (once the shuffled
      // deck is there) the statement hash and the scalar program behind it.  A batch that fills the chip gains nothing from
      // running them side by side; a small one hides its transcript lanes and ChaCha draws behind the group work, or the other way.
      if (overlap) {
        struct Restore {      // (kernel launches take their stream from the context)
          mp_ctx* c;
          rt::Stream keep;
          ~Restore() { c->stream = keep; }
        } restore{ctx, s};
        rt::event_record(ctx->ev_fork, s);
        ctx->stream = ctx->side;
        rt::stream_wait(ctx->side, ctx->ev_fork);
        MP_RUN(k_remask, C, B, 2 * N, ra);
        run_phase(pph[0], w, B, PH_NORM, l.shuf);
        rt::event_record(ctx->ev_shuf, ctx->side);
        prove_side_work(q, w, B, true);
        rt::event_record(ctx->ev_tab, ctx->side);
      } else {
        MP_RUN(k_remask, C, B, 2 * N, ra);
      }
      if (overlap) {
        prove_init(ia, B);
        run_phase(pph[0], w, B, PH_ALL, NO_SLOT, l.shuf, w.NS2.p);
        rt::stream_wait(s, ctx->ev_shuf);
      } else {
        run_phase(pph[0], w, B);
        prove_side_work(q, w, B, false);
      }
    }
    const ToomPlan& tk = q.pplan.toom;
    {
      FsStatementArgs a = statement_args(w, l.deck, l.shuf, l.cA, l.x, keyed ? l.pk : NO_SLOT);
      a.W = wire_words(w, B, 1);
      a.w_deck = 0;             // the input deck as it came; the shuffled deck was computed here and is taken from its P slots
      run_fs_round1(a, B);
    }
    ProveScalArgs sc{w.S.p, perm, l, w.Bpad, q.lin.p, q.lin_src.p, tk.E ? 0u : (uint32_t)q.pplan.lin.size()};
    MP_RUN(k_prove_scal1, C, B, N, sc);
    MP_RUN(k_prove_scal1b, C, B, n + sc.n_lin * n, sc);
    MP_RUN(k_prove_scal1c, C, B, 1, sc);
    if (tk.E) {                   // the scalar polynomial at the same points; the interpolation matrix as MSM scalars
      LinCombArgs la{w.S.p, q.lin.p, q.lin_src.p, q.lin_coef.p, q.consts.p, w.Bpad, n};
      MP_RUN(k_lin_comb, C, B, (uint32_t)q.pplan.lin.size() * n, la);
      FillConstArgs fa{w.S.p, q.consts.p, w.Bpad, tk.w_first, tk.w_const_first};
      MP_RUN(k_fill_consts, C, B, tk.E * tk.E, fa);
    }
    if (keyed) {      // tau_k pk, k < 2m, from the key's own tables (the proof's, or its key set's): finished partial sums of the E_k
      KeyTermsArgs ka{w.S.p, w.J.p, w.Bpad, l.metau, q.pplan.jkey, kset ? 2u : 1u, w.T.p, key_t_first, nwin,
                      kset ? kset->FB.p : nullptr, kidx, kset ? FbGeom{kset->bits, kset->windows, kset->entries} : FbGeom{8, 32, 255},
                      kset ? (uint32_t)kset->K : 0u};
      MP_RUN(k_key_terms, C, B, 2 * m, ka);
    }
    if (overlap) {
      rt::stream_wait(s, ctx->ev_tab);
      side_guard.joined = true;
      run_phase(pph[1], w, B, PH_ALL & ~PH_TABLES);
    } else {
      run_phase(pph[1], w, B);
    }
    run_phase(pph[5], w, B);      // Toom-Cook interpolation: the diagonals E_k from the 2m products (empty otherwise)
    const FsDev f{w.stage.p, w.seed.p, w.Bpad};
    {
      FsRoundArgs a{};
      a.f = f; a.S = w.S.p; a.P = w.P.p;
      a.step[0] = FsStep{l.cB, m, 0, 0, l.y, l.z};
      a.nsteps = 1;
      a.copy_from = NO_SLOT; a.copy_to = NO_SLOT;
      run_fs_round(a, B);
    }
    MP_RUN(k_prove_scal2, C, B, n + 1, sc);
    MP_RUN(k_prove_scal2b, C, B, n, sc);
    run_phase(pph[2], w, B);
    {
      FsRoundArgs a{};
      a.f = f; a.S = w.S.p; a.P = w.P.p;
      a.copy_from = l.cb; a.copy_to = l.hB + m - 1;
      a.step[0] = FsStep{l.cb, 1, 0, 0, NO_SLOT, NO_SLOT};
      a.step[1] = FsStep{l.hB, m, 0, 0, l.hx, l.hy};
      a.nsteps = 2;
      run_fs_round(a, B);
    }
    MP_RUN(k_prove_scal3, C, B, n + 1, sc);
    {      // small batches: the d_k as partial sums over column ranges first (kernels_proto.hpp; 0.1 -> 0.03 ms of a single proof)
      ProveScalArgs sd = sc;
      sd.d_parts = B <= SCAL3D_SPLIT_MAX_BATCH ? scal3d_parts(m, n) : 0u;
      if (sd.d_parts >= 2) {
        MP_RUN(k_prove_scal3d_part, C, B, (2 * m + 1) * sd.d_parts, sd);
      } else {
        sd.d_parts = 0;
      }
      MP_RUN(k_prove_scal3d, C, B, 2 * m + 1, sd);
    }
    run_phase(pph[3], w, B);
    {
      FsRoundArgs a{};
      a.f = f; a.S = w.S.p; a.P = w.P.p;
      a.copy_from = NO_SLOT; a.copy_to = NO_SLOT;
      a.step[0] = FsStep{l.zcA0, 2 + 2 * m + 1, 0, 0, l.zx, NO_SLOT};
      a.step[1] = FsStep{l.svcd, 3, 0, 0, l.svx, NO_SLOT};
      a.step[2] = FsStep{l.mecA0, 1 + 6 * m, 0, 0, l.mx, NO_SLOT};
      a.nsteps = 3;
      run_fs_round(a, B);
    }
    MP_RUN(k_prove_scal4, C, B, n + 3, sc);
    {
      StorePointsArgs a{out_decks, w.P.p, w.Bpad, 2 * N, l.shuf};
      MP_RUN(k_store_points, C, B, 2 * N, a);
      ProofIoArgs pa{out_proofs, w.S.p, w.P.p, w.status.p, q.pwire.p, w.Bpad, (uint32_t)proof_size_bytes(m, n, G_::PB)};
      MP_RUN(k_store_proof, C, B, (uint32_t)q.pplan.wire.size(), pa);
    }
    rt::d2d(status, w.status.p, (size_t)B * 4, s);
  }

  // ---------------------------------------------------------------- verify
  // Two passes.  (1) Screening: all group equations merged with random weights into ONE multi-scalar multiplication
  // (n + 5 fixed-base terms instead of one per equation and base, a quarter of the doubling chains) -- every honest batch
  // ends here.  (2) Only if some proof failed the screen: the equations one by one, to report the FIRST failing check by
  // name as the reference does [REF tests.rs:223-225].  A proof that passes (1) satisfies every equation except with
  // probability ~2^-250 over weights that depend on the whole proof.
  struct VArgs {
    uint32_t B;
    const uint8_t *decks, *shuf, *proofs;
    int32_t* status;
    const uint8_t* keys;
    const mp_keyset* kset;
    const uint32_t* kidx;
  };
  // a pipelined verify call whose screening verdict has not been looked at yet: k_verdict_merged writes the flag straight into a
  // page-locked host word (zero-copy; no copy call on the lane, which would make the runtime wait); it is valid once `ev` has passed
  struct Pending {
    VArgs v{};
    uint32_t* h_flag = nullptr;      // host address
    uint32_t* d_flag = nullptr;      // the same word as the device sees it
    rt::Event ev = nullptr;
    uint32_t gl = 0;                 // > 0: the screen was the equation of groups of gl proofs; their verdicts:
    uint32_t *h_gbad = nullptr, *d_gbad = nullptr;      // [gbad_cap] page-locked words the device writes directly (k_chain_verdict)
    size_t gbad_cap = 0;
  };
  std::deque<Pending> pend;          // oldest first; at most `pipeline` of them stay unexamined when a verify call returns
  std::vector<Pending> pend_pool;    // flag words and events for reuse
  Workspace vws;                     // the verify lane's arenas (pipelined mode: a prove call uses `ws` at the same time)
  static void release(Pending& p_) {
    if (p_.ev) rt::event_destroy(p_.ev);
    rt::host_free(p_.h_flag);
    rt::host_free(p_.h_gbad);
  }
  ~Table() {
    for (auto& p_ : pend_pool) release(p_);
    for (auto& p_ : pend) release(p_);
  }
  // one pass over a batch on the context's CURRENT lane: merged = the screening equation, else equation by equation
  void verify_pass(Workspace& w, const VArgs& v, bool merged, bool vlane, uint32_t* host_flag = nullptr) {
    const uint32_t B = v.B;
    const bool keyed = v.keys != nullptr || v.kset != nullptr;
    const uint8_t* keys = v.keys;
    PlanSet& q = pick(B, keyed);
    cur_table_group = q.table_group;
    cur_norm_chunk = q.norm_chunk;
    const VerifyLay& l = q.vplan.lay;
    rt::Stream s = ctx->stream;
    rt::dzero(w.status.p, (size_t)w.Bpad * 4, s);
    {
      uint32_t* const ww = wire_words(w, B, 2);
      LoadPointsArgs a{v.decks, w.P.p, w.status.p, w.Bpad, 2 * N, l.deck, ww, 0};
      MP_RUN(k_load_points, C, B, 2 * N, a);
      LoadPointsArgs b{v.shuf, w.P.p, w.status.p, w.Bpad, 2 * N, l.shuf, ww, 2 * N};
      MP_RUN(k_load_points, C, B, 2 * N, b);
      ProofIoArgs pa{const_cast<uint8_t*>(v.proofs), w.S.p, w.P.p, w.status.p, q.vwire.p, w.Bpad, (uint32_t)proof_size_bytes(m, n, G_::PB)};
      MP_RUN(k_load_proof, C, B, (uint32_t)q.vplan.wire.size(), pa);
      if (v.kset) keys = gather_keys(B, v.kset, v.kidx, w.status.p, vlane);
      if (keyed) {
        LoadPointsArgs ka{keys, w.P.p, w.status.p, w.Bpad, 1, l.pk};
        MP_RUN(k_load_points, C, B, 1, ka);
      }
      // decks and proof points are the P slots [0, pk); the key follows
      check_verify_inputs(w, B, l, keyed);
    }
    // The window tables of the verifier's bases need the loaded points and nothing else: batches that do not fill the chip build
    // them on `side` while the main stream hashes the transcript and derives the scalars (the table kernel leans on HBM, the
    // transcript lanes on latency: they do not compete)
    PhaseDev& ph = merged ? q.vmph : q.vph;
    const bool vtab_forked = overlap_max && B <= overlap_max && ph.n_tables != 0;
    SideGuard vguard{ctx, !vtab_forked};
    if (vtab_forked) {
      struct Restore {
        mp_ctx* c;
        rt::Stream keep;
        ~Restore() { c->stream = keep; }
      } restore{ctx, s};
      rt::event_record(ctx->ev_fork, s);
      ctx->stream = ctx->side;
      rt::stream_wait(ctx->side, ctx->ev_fork);
      run_tables(ph, w, B, w.NS.p);
      rt::event_record(ctx->ev_tab, ctx->side);
    }
    {
      VerifyFsArgs a{};
      a.st = statement_args(w, l.deck, l.shuf, l.cA, l.x, keyed ? l.pk : NO_SLOT);
      a.st.W = wire_words(w, B, 2);
      a.st.w_deck = 0;
      a.st.w_shuf = 2 * N;
      a.l = l;
      a.merge = merged ? 1u : 0u;
      run_verify_fs(a, B);
      VerifyScalArgs sa{w.S.p, w.P.p, w.direct.p, l, q.vplan.cm, w.Bpad};
      MP_RUN(k_verify_scal, C, B, n + 2, sa);
    }
    if (merged) {
      VerifyMergeArgs ma{w.S.p, q.mjobs.p, q.mpairs.p, w.Bpad};
      MP_RUN(k_verify_merge, C, B, (uint32_t)q.vplan.mjobs.size(), ma);
    }
    if (vtab_forked) {
      rt::stream_wait(s, ctx->ev_tab);
      vguard.joined = true;
    }
    run_phase(ph, w, B, vtab_forked ? (PH_ALL & ~PH_TABLES) : PH_ALL);
    if (merged) {
      uint32_t* fl = host_flag;      // (already zero)
      if (!fl) fl = flag_word(vlane, s);
      VerdictMergedArgs a{w.J.p, w.direct.p, w.status.p, fl, w.Bpad, l.chk_merged};
      MP_RUN(k_verdict_merged, C, B, 1, a);
    } else {
      VerdictArgs a{w.J.p, w.direct.p, w.status.p, w.Bpad, l.chk_first};
      MP_RUN(k_verdict, C, B, 1, a);
    }
    rt::d2d(v.status, w.status.p, (size_t)B * 4, s);
  }
  // whether a batch of this size is screened with the merged equation first: small batches (the two finest splits) go straight to
  // the per-equation pass -- with an idle chip the merged MSM is one long dependency chain and its flag read-back a round trip; it
  // pays from the medium plan on (+3 % there, measured) -- unless the merged equation runs on the bucket kernel, which spreads ONE
  // MSM over windows x 64 lanes
  bool screens(uint32_t B, bool keyed) {
    const int plan = plan_of(B);
    return merged_verify && (plan != 3 || pick(B, keyed).vmph.n_b);
  }
  // ---- round 5: a failing screen costs what the rejected proofs cost, not what the batch costs.  Every screen says WHICH proofs it
  // could not clear -- k_verdict_merged marks them 1, k_chain_verdict writes one word per group / chain --, their inputs are gathered
  // into a contiguous sub-batch (k_gather_rows), the sub-batch takes the next finer pass and its status words are scattered back:
  //   level 0  the suspects of failing groups: equations of sub-groups an eighth the size (16 proofs of a 52-card deck) when there are
  //            enough of them to fill the bucket kernel (>= 128 sub-groups), else straight to level 1;
  //   level 1  equation by equation: the FIRST failing check by name, exactly as the reference reports it [REF tests.rs:223-225].
  // Everybody else's verdict stands.  One tampered proof among 262 144 re-verifies 128 proofs, not 262 144.
  struct SubBatch {
    DevBuf<uint32_t> decks, shuf, proofs, keys, kidx, idx;
    DevBuf<int32_t> status;
  };
  SubBatch sub[2][3];                 // [lane][depth of the refinement]
  Workspace rws[2];                   // [lane] arenas of the per-equation passes over sub-batches (sized by the sub-batch, not by the batch)
  DevBuf<uint32_t> gbad[2];           // per-group verdicts of a group / chain pass on the caller's lane, on the verify lane
  uint32_t refine_points = 0;         // points per sub-group equation (mp_set_group_refine; 0 = an eighth of the group equation's)
  uint32_t refine_min = 128;          // fewer sub-groups than this: straight to the per-equation pass
  uint64_t n_reverified = 0;          // proofs that went through a per-equation pass because a screen could not clear them
  void set_group_refine(uint32_t points, uint32_t min_groups) override {
    refine_points = points;
    refine_min = min_groups ? min_groups : 128u;
  }
  uint64_t reverified() const override { return n_reverified; }
  // proofs per sub-group for the nsub suspects a screen with groups of l1 proofs left (l1 = 0: no groups above, e.g. a chain)
  // `crowded`: (nearly) every group of the screen above failed -- the rate of bad proofs is beyond what an eighth of such a group could
  // clear (round 6: 1 % of bad proofs fails every group of 1 024 and three in four of 128): a sixty-fourth instead
  uint32_t subgroup_size(size_t nsub, bool keyed, uint32_t l1, bool crowded = false) const {
    const uint32_t per = 4 * N + 11 * m + 8 + (keyed ? 1u : 0u);
    if (!group_points || !merged_verify) return 0;
    // (a sub-group equation costs 0.55 us per proof at 32 proofs, 0.73 at 16, 0.85 at 8, 1.1 at 4 -- the per-equation pass 1.2: an eighth
    // of a group of fewer than 64 proofs is not worth an equation of its own; profiles/r05h_rejection_strategies.txt)
    if (!refine_points && l1 && l1 < 64) return 0;
    const uint32_t pts = refine_points ? refine_points : (l1 ? l1 * per : group_points) / (crowded && l1 >= 512 ? 64u : 8u);
    const uint32_t want = (uint32_t)std::min<uint64_t>((pts + per / 2) / per, 1023u);
    if (want < 2 || (l1 && want >= l1) || (uint64_t)want * per + n + 5 > BUCKET_TERMS_MAX || nsub < (uint64_t)refine_min * want) return 0;
    return want;
  }
  void gather_rows(const void* src, DevBuf<uint32_t>& dst, const uint32_t* d_idx, size_t rows, size_t row_bytes) {
    const uint32_t words = (uint32_t)(row_bytes / 4);
    dst.alloc(rows * words, ctx->stream, false);
    const size_t slice = std::max<size_t>(1, ((size_t)1 << 30) / words);      // (thread index: 32 bits)
    for (size_t r0 = 0; r0 < rows; r0 += slice) {
      const size_t rc = std::min(slice, rows - r0);
      GatherRowsArgs ga{reinterpret_cast<const uint32_t*>(src), dst.p + r0 * words, d_idx + r0, words};
      MP_RUN(k_gather_rows, C, (uint32_t)(rc * words), 1, ga);
    }
  }
  // the positions of the non-zero words among `count` device (or mapped host) words, on the context's current lane
  void read_words(const void* d_words, size_t count, std::vector<uint32_t>& host) {
    host.resize(count);
    rt::d2h(host.data(), d_words, count * 4, ctx->stream);
    rt::stream_sync(ctx->stream);
  }
  // members of the failing groups (lane of (member j, group t) = j T + t)
  static void group_members(const uint32_t* bad, uint32_t T, uint32_t L, std::vector<uint32_t>& idx) {
    idx.clear();
    for (uint32_t t = 0; t < T; ++t)
      if (bad[t])
        for (uint32_t j = 0; j < L; ++j) idx.push_back(j * T + t);
    std::sort(idx.begin(), idx.end());
  }
  // the proofs idx[] of batch v (ascending) through the finer passes; their status words replace the screen's marks
  // own_arenas: the caller's arenas (ws) are laid out for batch v (verify_dev); a chain call's are not (its lean workspace is cws)
  // depth: how many sub-batches lie above this one (each has gather buffers of its own: sub[lane][depth])
  void verify_subset(const VArgs& v, std::vector<uint32_t>& idx, int level, bool vlane, uint32_t l1 = 0, bool own_arenas = true, uint32_t depth = 0,
                     int crowd = -1) {
    if (depth >= 2) level = 1;
    // (the suspects are whole groups of l1: if they are nearly the whole batch, nearly every group failed -- judged before any slicing)
    const bool crowded = crowd >= 0 ? crowd != 0 : (l1 != 0 && idx.size() * 20 >= (size_t)v.B * 19);
    if (idx.empty()) return;
    const bool keyed = v.keys != nullptr || v.kset != nullptr;
    rt::Stream s = ctx->stream;
    // Bounded memory whatever the number of suspects (one bad link per chain equation sends all T x L links of a call here):
    // at most 131 072 52-card proofs' worth of them are gathered and looked at at a time (2.6 GB of decks and proofs, arenas of the passes
    // below for that many lanes); the slices are sub-batches of their own, any partition of the suspects gives the same status words
    const size_t sub_max = std::max<size_t>(4096, ((size_t)131072 * 52u / N) & ~(size_t)1023);
    if (!(level == 1 && idx.size() == v.B && (vlane || own_arenas)) && idx.size() > sub_max) {
      for (size_t o = 0; o < idx.size(); o += sub_max) {
        std::vector<uint32_t> part(idx.begin() + o, idx.begin() + std::min(idx.size(), o + sub_max));
        verify_subset(v, part, level, vlane, l1, own_arenas, depth, crowded ? 1 : 0);
      }
      return;
    }
    const uint32_t L2 = level == 0 ? subgroup_size(idx.size(), keyed, l1, crowded) : 0u;
    if (!L2) level = 1;
    // The equations of a sub-batch run on arenas of their OWN (rws), in slices of at most 32 768 52-card proofs: the work split follows
    // the size of the sub-batch (128 suspects take the finest split), and a split with more table / digit slots per proof than the
    // caller's must not grow the caller's arenas -- those are laid out for 262 144 lanes, and re-allocating them cost the first rejected
    // proof of a session 3.5 s (measured).  Bounded memory (~20 GB at the most), cost proportional to the suspects.
    auto per_equation = [&](const VArgs& a) {
      Workspace& w = rws[vlane ? 1 : 0];
      const uint32_t slice = std::max<uint32_t>(64u, (uint32_t)(((uint64_t)32768 * 52u / N) & ~63ull));
      const size_t dsz = (size_t)2 * N * G_::PB, psz = proof_size_bytes(m, n, G_::PB);
      n_reverified += a.B;
      for (uint32_t s0 = 0; s0 < a.B; s0 += slice) {
        const uint32_t sc = std::min(slice, a.B - s0);
        const VArgs ss{sc, a.decks + (size_t)s0 * dsz, a.shuf + (size_t)s0 * dsz, a.proofs + (size_t)s0 * psz, a.status + s0,
                       a.keys ? a.keys + (size_t)s0 * G_::PB : nullptr, a.kset, a.kidx ? a.kidx + s0 : nullptr};
        reserve_ws(w, sc, keyed);
        verify_pass(w, ss, false, vlane);
      }
    };
    if (level == 1 && idx.size() == v.B) {      // (everybody: no gather)
      if (vlane || !own_arenas) {               // (no arenas of the batch's size for the equations on the verify lane or under a chain call: slices)
        per_equation(v);
      } else {                                  // (the caller's arenas are laid out for this batch and its work split already)
        reserve_ws(ws, v.B, keyed);
        n_reverified += v.B;
        verify_pass(ws, v, false, false);
      }
      return;
    }
    const uint32_t distinct = (uint32_t)idx.size();
    if (L2)
      while (idx.size() % L2) idx.push_back(idx[0]);      // (a suspect twice: the same verdict written twice)
    const uint32_t nsub = (uint32_t)idx.size();
    SubBatch& sb = sub[vlane ? 1 : 0][depth];
    sb.idx.upload(idx, s);
    gather_rows(v.decks, sb.decks, sb.idx.p, nsub, (size_t)2 * N * G_::PB);
    gather_rows(v.shuf, sb.shuf, sb.idx.p, nsub, (size_t)2 * N * G_::PB);
    gather_rows(v.proofs, sb.proofs, sb.idx.p, nsub, proof_size_bytes(m, n, G_::PB));
    if (v.kset) gather_rows(v.kidx, sb.kidx, sb.idx.p, nsub, 4);
    else if (v.keys) gather_rows(v.keys, sb.keys, sb.idx.p, nsub, G_::PB);
    sb.status.alloc(nsub, s, false);
    const VArgs sv{nsub, reinterpret_cast<const uint8_t*>(sb.decks.p), reinterpret_cast<const uint8_t*>(sb.shuf.p),
                   reinterpret_cast<const uint8_t*>(sb.proofs.p), sb.status.p,
                   v.keys && !v.kset ? reinterpret_cast<const uint8_t*>(sb.keys.p) : nullptr, v.kset, v.kset ? sb.kidx.p : nullptr};
    if (L2) {
      const uint32_t T2 = nsub / L2;
      DevBuf<uint32_t>& gb = gbad[vlane ? 1 : 0];
      gb.alloc(T2, s, false);
      verify_group_pass(sv, L2, vlane, nullptr, gb.p);
      if (read_flag(vlane)) {
        std::vector<uint32_t> bad, idx2;
        read_words(gb.p, T2, bad);
        group_members(bad.data(), T2, L2, idx2);
        // (the members of the failing sub-groups: through sub-groups an eighth the size once more if there are enough of them -- 1 024 ->
        // 128 -> 16 --, else equation by equation)
        verify_subset(sv, idx2, L2 >= 64 ? 0 : 1, vlane, L2, true, depth + 1);
        for (uint32_t i2 : idx2)                          // (mp_reverified_count counts proofs, not the copies that fill the last sub-group)
          if (i2 >= distinct) n_reverified -= 1;
      }
    } else {
      per_equation(sv);
    }
    ScatterStatusArgs sa{sb.status.p, v.status, sb.idx.p};
    MP_RUN(k_scatter_status, C, nsub, 1, sa);
    rt::stream_sync(s);      // (idx[] was uploaded from pageable memory and sb is reused by the next failing batch)
  }
  // the proofs a per-proof screen marked (status word 1) through the per-equation pass
  void refine_marked(const VArgs& v, bool vlane) {
    std::vector<uint32_t> st, idx;
    read_words(v.status, v.B, st);
    for (uint32_t b = 0; b < v.B; ++b)
      if ((int32_t)st[b] == 1) idx.push_back(b);
    verify_subset(v, idx, 1, vlane);
  }
  void verify_dev(size_t B_, const uint8_t* decks, const uint8_t* shuf, const uint8_t* proofs, int32_t* status,
                  const uint8_t* keys, const mp_keyset* kset, const uint32_t* kidx) override {
    const VArgs v{(uint32_t)B_, decks, shuf, proofs, status, keys, kset, kidx};
    const bool keyed = keys != nullptr || kset != nullptr;
    if (keyed) ensure_keyed();
    const uint32_t gl = group_size(v.B, keyed);      // > 0: the screen is one equation per group of gl proofs (bucket kernel)
    if (pipeline) {
      // Pipelined mode: the call runs on the verify lane with arenas of its own and does NOT wait for its screening verdict, so the
      // caller's next prove call overlaps it on the chip.  The verdict of call k is looked at when call k + 1 comes in (or at
      // mp_sync); only then -- and only for the proofs the screen could not clear -- do the finer passes run.
      resolve_pending((size_t)pipeline - 1);      // this call makes it `pipeline` unexamined ones
      rt::event_record(ctx->ev_vin, ctx->stream);
      LaneSwap lane(ctx);
      rt::stream_wait(ctx->stream, ctx->ev_vin);
      // (the verify lane's arenas are sized ON the verify lane: a zero-fill of theirs must queue behind the verify work in flight)
      if (!gl) reserve_ws(vws, v.B, keyed);
      if (!gl && !screens(v.B, keyed)) {
        verify_pass(vws, v, false, true);
        return;
      }
      Pending pn;
      if (!pend_pool.empty()) {
        pn = pend_pool.back();
        pend_pool.pop_back();
      } else {
        void* dp = nullptr;
        pn.h_flag = (uint32_t*)rt::host_alloc_mapped(4, &dp);
        pn.d_flag = (uint32_t*)dp;
        pn.ev = rt::event_create();
      }
      pn.v = v;
      pn.gl = gl;
      *pn.h_flag = 0;
      if (gl && pn.gbad_cap < v.B / gl) {
        rt::host_free(pn.h_gbad);
        pn.h_gbad = nullptr;
        pn.gbad_cap = 0;
        void* dp = nullptr;
        pn.h_gbad = (uint32_t*)rt::host_alloc_mapped((size_t)(v.B / gl) * 4, &dp);
        pn.d_gbad = (uint32_t*)dp;
        pn.gbad_cap = v.B / gl;
      }
      pend.push_back(pn);                   // (before the launches: an exception on the way still leaves the slot owned)
      if (gl) verify_group_pass(v, gl, true, pn.d_flag, pn.d_gbad);
      else verify_pass(vws, v, true, true, pn.d_flag);
      rt::event_record(pn.ev, ctx->stream);
      return;
    }
    if (gl) {
      const uint32_t T = v.B / gl;
      gbad[0].alloc(T, ctx->stream, false);
      verify_group_pass(v, gl, false, nullptr, gbad[0].p);
      if (!read_flag(false)) {              // every group's equation holds
        note_group_verdicts(T, 0, gl);
        return;
      }
      std::vector<uint32_t> bad, idx;
      read_words(gbad[0].p, T, bad);
      group_members(bad.data(), T, gl, idx);
      note_group_verdicts(T, (uint32_t)(idx.size() / gl), gl);
      verify_subset(v, idx, 0, false, gl);  // the members of the failing groups, nobody else
      return;
    }
    reserve_for(v.B, keyed);
    // Two passes.  (1) Screening; every honest batch ends here.  (2) The proofs the screen marked, equation by equation
    if (!screens(v.B, keyed)) {
      verify_pass(ws, v, false, false);
      return;
    }
    verify_pass(ws, v, true, false);
    if (read_flag(false)) refine_marked(v, false);
  }
  // look at the screening verdicts of all but the `keep` most recent pipelined verify calls
  void resolve_pending(size_t keep) {
    while (pend.size() > keep) {
      Pending pn = pend.front();
      pend.pop_front();
      pend_pool.push_back(pn);
      rt::event_sync(pn.ev);
      if (!*pn.h_flag) {
        if (pn.gl) note_group_verdicts(pn.v.B / pn.gl, 0, pn.gl);
        continue;
      }
      LaneSwap lane(ctx);
      if (pn.gl) {
        std::vector<uint32_t> idx;
        group_members(pn.h_gbad, pn.v.B / pn.gl, pn.gl, idx);
        note_group_verdicts(pn.v.B / pn.gl, (uint32_t)(idx.size() / pn.gl), pn.gl);
        verify_subset(pn.v, idx, 0, true, pn.gl);
      } else {
        refine_marked(pn.v, true);          // the proofs the screen marked: the first failing check of each
      }
      rt::stream_sync(ctx->stream);
    }
  }
  void flush() override {
    resolve_pending(0);
    if (ctx->vstream) rt::stream_sync(ctx->vstream);
  }


  // ---------------------------------------------------------------- chain verification
  // The L shuffles of one card table form a chain (deck_{j+1} = output of link j [REF examples/round.rs:268-350]) and every one of
  // them is verified.  Checked one by one, each of the L - 1 inner decks is a base in TWO verifier equations (as the shuffled deck
  // of link j - 1 and as the input deck of link j).  Here the merged equations of all links of a table are added up with weights
  // rho_j that depend on every proof of the chain: every distinct point appears ONCE -- (L + 1) 2N deck points + L (11m + 8) proof
  // points instead of L (4N + 11m + 8) -- the n + 5 fixed bases appear once per table instead of once per link, and the resulting
  // MSM (4 424 terms for 32 links of a 52-card deck) is long enough for the bucket kernel.  A table whose chain fails is re-verified
  // link by link, so the per-link status words are exactly those of verify_dev.  T tables, lane of (link j, table t) = j T + t.
  struct ChainPlan {
    uint32_t L = 0, G = 0;
    bool keyed = false;
    Phase ph;
    PhaseDev dev;
    std::vector<ChainTerm> cterms;
    DevBuf<ChainTerm> dterms;
    uint32_t K = 0, nfix = 0, nJ = 0;
    std::vector<uint32_t> tile_src;     // chain equations (round 6): slot | link << 20 of term i -- where k_chain_tile finds the point of entry i of the equation's run
    DevBuf<uint32_t> dtile_src;
  };
  ChainPlan chain;
  Workspace cws;                      // lean workspace of chain verification: no window tables, no digit planes
  DevBuf<uint32_t> chain_cw, chain_cs, chain_dig, chain_part;
  // the verdicts of T chain / group equations of L links (kernels_proto.hpp: per link, per equation, then the caller's status words)
  void run_chain_verdict(ChainVerdictArgs va) {
    chain_part.alloc(va.T, ctx->stream, false);
    rt::dzero(chain_part.p, (size_t)va.T * 4, ctx->stream);
    va.part = chain_part.p;
    MP_RUN(k_chain_check, C, va.T * va.L, 1, va);
    MP_RUN(k_chain_verdict, C, va.T, 1, va);
    if (va.out) MP_RUN(k_chain_mark, C, va.T * va.L, 1, va);
  }
  // the weights of T chain / group equations of L links each (kernels_proto.hpp: block digests, then table key, block key, 64 weights per lane)
  void run_chain_weights(Workspace& w, uint32_t Tpad, uint32_t T, uint32_t L) {
    const uint32_t nb = (L + CW_BLOCK - 1) / CW_BLOCK;
    chain_dig.alloc((size_t)nb * 8 * Tpad, ctx->stream, false);
    ChainWeightsArgs wa{w.seed.p, chain_cw.p, chain_dig.p, w.Bpad, Tpad, T, L};
    MP_RUN(k_chain_digest, C, T, nb, wa);
    MP_RUN(k_chain_weights, C, T, nb, wa);
  }
  DevBuf<int16_t> chain_d8;
  // Tables per chain equation (round 5).  One table's chain is 4 424 points for 32 links of a 52-card deck: 8-bit windows (32 per
  // scalar) and a wave-wide reduction per window that is a sixth of its additions.  The chains of G tables added up with the same kind
  // of weights (they depend on every proof of every member) are ONE equation of G x 4 424 points: 10-bit windows (26 per scalar), the
  // reduction spread over G times the terms, the fixed bases once per G tables -- what group verification does for independent proofs.
  // Member g of equation e is table g (T / G) + e, so that link j of member g is lane (j G + g)(T / G) + e: the plan below is the
  // chain plan with "links" j G + g and a chain's consecutive links G apart; kernels and lane formula are unchanged.  A failing
  // equation sends ITS G x L links through the per-link verifier.  G = the divisor of T that brings the equation nearest to the
  // group equation's size (mp_set_group_verify), with enough equations left to keep the persistent waves busy; 1 = a table on its own.
  uint32_t chain_points_per_table(uint32_t L, bool keyed) const { return (L + 1) * 2 * N + L * (11 * m + 8) + (keyed ? 1u : 0u); }
  uint32_t chain_group_of(uint32_t T, uint32_t L, bool keyed) const {
    // (an explicit mp_set_chain_group is honoured whatever mp_set_group_verify says; the automatic rule follows the group equations')
    if (chain_group == 1 || (!chain_group && !group_points && !group_points_wg)) return 1;
    const uint32_t per = chain_points_per_table(L, keyed);
    uint32_t want = chain_group;
    if (!want) {
      const uint32_t eq_min = std::max<uint32_t>(1u, (uint32_t)(((uint64_t)group_min_batch * 2u) / 13u));
      want = std::min<uint32_t>((group_points + per / 2) / per, T / eq_min);
      // (round 6, as group_size: equations for the split pipeline if at least min_batch / 48 of them, of at least 50 000 points, are left
      // -- 64 tables x 4 392 points for 32 links of a 52-card deck)
      const uint32_t want_wg = std::min<uint32_t>((group_points_wg + per / 2) / per, T / std::max<uint32_t>(1u, group_min_batch / 512u));
      if (group_points_wg && (uint64_t)want_wg * per >= GROUP_WG_POINTS_MIN) want = std::max(want, want_wg);
    }
    if (want < 2) return 1;
    for (uint32_t d = 0; d <= want; ++d)
      for (int sgn = 1; sgn >= -1; sgn -= 2) {
        const int64_t G = (int64_t)want + sgn * (int64_t)d;
        // (12 bits of link in a term's tile source: k_chain_tile)
        if (G < 2 || 2 * G < (int64_t)want || G > 2 * (int64_t)want || (uint64_t)G * per + n + 5 > BUCKET_TERMS_MAX || (uint64_t)G * (L + 1) > 4094) continue;
        if (T % (uint32_t)G == 0) return (uint32_t)G;
      }
    return 1;
  }
  void build_chain_plan(uint32_t L, bool keyed, uint32_t G = 1) {
    if (chain.L == L && chain.keyed == keyed && chain.G == G) return;
    if (keyed) ensure_keyed();
    PlanSet& q = (keyed ? psk : ps)[0];
    const VerifyLay& l = q.vplan.lay;
    chain.ph = Phase();
    chain.cterms.clear();
    chain.tile_src.clear();
    chain.L = L;
    chain.G = G;
    chain.keyed = keyed;
    uint32_t next_partial = 1;                  // J slot 0 = the chain equation's value
    const uint32_t cbits = bucket_bits_of(G * ((L + 1) * 2 * N + L * (l.pk - l.cA) + (keyed ? 1u : 0u)));
    PhaseBuilder pb(chain.ph, next_partial, FCHUNK, VCHUNK, 1u, bk_windows(R::BITS, cbits), 1u, cbits);
    pb.begin(0);
    // (a term's point: its index in the equation's contiguous run -- k_chain_tile copies it there from slot | link << 20)
    auto var = [&](ChainTerm ct, uint32_t pslot, uint32_t link) {
      pb.var((uint32_t)chain.cterms.size(), (uint32_t)chain.cterms.size());
      chain.tile_src.push_back(pslot | (link << 20));
      chain.cterms.push_back(ct);
    };
    // ("link" j G + g of the equation = link j of member g)
    for (uint32_t g = 0; g < G; ++g)
      for (uint32_t j = 0; j <= L; ++j)
        for (uint32_t i = 0; i < 2 * N; ++i) {
          if (j == 0) var(ChainTerm{l.mvar + l.deck + i, g, 1, NO_SLOT, G}, l.deck + i, g);
          else if (j < L) var(ChainTerm{l.mvar + l.deck + i, j * G + g, 1, l.mvar + l.shuf + i, G}, l.deck + i, j * G + g);
          else var(ChainTerm{l.mvar + l.shuf + i, (L - 1) * G + g, 1, NO_SLOT, G}, l.shuf + i, (L - 1) * G + g);
        }
    for (uint32_t j = 0; j < L * G; ++j)
      for (uint32_t slot = l.cA; slot < l.pk; ++slot) var(ChainTerm{l.mvar + slot, j, 1, NO_SLOT, 1}, slot, j);
    if (keyed)
      for (uint32_t g = 0; g < G; ++g) var(ChainTerm{l.mvar + l.pk, g, L, NO_SLOT, G}, l.pk, g);      // one key term per member
    chain.K = (uint32_t)chain.cterms.size();
    // (the scalar of a fixed base: a sum over the equation's L G links, in runs of at most 64 -- as in the group plan below)
    FixedBases fb{n};
    for (uint32_t f = 0; f < fb.count(); ++f) {
      if (keyed && f == fb.pk()) continue;
      for (uint32_t j0 = 0; j0 < L * G; j0 += 64u) {
        pb.fixed((uint32_t)chain.cterms.size(), f);
        chain.cterms.push_back(ChainTerm{l.mfix + f, j0, std::min(64u, L * G - j0), NO_SLOT, 1});
      }
    }
    chain.nfix = (uint32_t)chain.cterms.size() - chain.K;
    pb.end();
    chain.nJ = next_partial;
    chain.dev.upload(chain.ph, ctx->stream);
    chain.dterms.upload(chain.cterms, ctx->stream);
    chain.dtile_src.upload(chain.tile_src, ctx->stream);
  }
  size_t chain_lane_bytes(uint32_t, bool keyed) override {
    if (keyed) ensure_keyed();
    const VerifyLay& l = (keyed ? psk : ps)[0].vplan.lay;
    // J slots of a chain plan: the equation's value, the bucket job's result and one partial sum per window (at most 33 of 8 bits), the
    // fixed-base part -- whatever the number of links or of tables per equation (build_chain_plan; Workspace::ensure takes at least 8)
    const size_t fwb = (size_t)G_::FW * 4, nJ = 4 + bk_windows(R::BITS, 8);
    // (Workspace::ensure: S, P, J, one window-table row and digit plane it always keeps, inversion scratch, stage, seed, direct, status, digits)
    return (size_t)l.nS * 32 + (size_t)l.nP * 2 * fwb + nJ * 3 * fwb + nwin + (size_t)VB_ENTRIES * 2 * fwb + nJ * fwb + (size_t)stage_words_needed() * 4 + 32 + 8 + 4 + 8;
  }
  size_t chain_lanes_held() const override { return cws.cap; }
  uint32_t chain_group_size(size_t T, uint32_t L, bool keyed) const override { return T < 0x7FFFFFFFu ? chain_group_of((uint32_t)T, L, keyed) : 1u; }
  void verify_chain_dev(size_t T_, uint32_t L, const uint8_t* decks, const uint8_t* proofs, int32_t* status, const uint8_t* keys) override {
    const uint32_t T = (uint32_t)T_, B = T * L;
    const bool keyed = keys != nullptr;
    // G tables per equation: Tq equations of Lq "links" each, lane of (link, equation) = link Tq + equation as before
    const uint32_t G = chain_group_of(T, L, keyed), Tq = T / G, Lq = L * G, Tpad = (Tq + 63u) & ~63u;
    build_chain_plan(L, keyed, G);
    PlanSet& q = (keyed ? psk : ps)[0];
    const VerifyLay& l = q.vplan.lay;
    rt::Stream s = ctx->stream;
    Workspace& w = cws;
    w.fw = G_::FW;
    w.ensure(B, l.nS, l.nP, std::max(chain.nJ, 8u), 0, 0, nwin, stage_words_needed(), s, 0);
    const size_t deck_bytes = (size_t)2 * N * G_::PB;
    rt::dzero(w.status.p, (size_t)w.Bpad * 4, s);
    {
      LoadPointsArgs a{decks, w.P.p, w.status.p, w.Bpad, 2 * N, l.deck};
      MP_RUN(k_load_points, C, B, 2 * N, a);
      LoadPointsArgs b{decks + (size_t)T * deck_bytes, w.P.p, w.status.p, w.Bpad, 2 * N, l.shuf};
      MP_RUN(k_load_points, C, B, 2 * N, b);
      ProofIoArgs pa{const_cast<uint8_t*>(proofs), w.S.p, w.P.p, w.status.p, q.vwire.p, w.Bpad, (uint32_t)proof_size_bytes(m, n, G_::PB)};
      MP_RUN(k_load_proof, C, B, (uint32_t)q.vplan.wire.size(), pa);
      if (keyed) {
        LoadPointsArgs ka{keys, w.P.p, w.status.p, w.Bpad, 1, l.pk};
        MP_RUN(k_load_points, C, B, 1, ka);
      }
      check_verify_inputs(w, B, l, keyed);
    }
    {
      VerifyFsArgs a{};
      a.st = statement_args(w, l.deck, l.shuf, l.cA, l.x, keyed ? l.pk : NO_SLOT);
      a.l = l;
      a.merge = 1u;
      run_verify_fs(a, B);
      VerifyScalArgs sa{w.S.p, w.P.p, w.direct.p, l, q.vplan.cm, w.Bpad};
      MP_RUN(k_verify_scal, C, B, n + 2, sa);
      VerifyMergeArgs ma{w.S.p, q.mjobs.p, q.mpairs.p, w.Bpad};
      MP_RUN(k_verify_merge, C, B, (uint32_t)q.vplan.mjobs.size(), ma);
    }
    // the chain equation: weights, one scalar per distinct point / fixed base, ONE bucket MSM + fixed-base part per table
    const uint32_t nterms = chain.K + chain.nfix;
    chain_cw.alloc((size_t)Lq * Tpad * 8, s, false);
    chain_cs.alloc((size_t)nterms * Tpad * 8, s);
    chain_d8.alloc((size_t)chain.dev.b_dig_bytes * Tpad, s);
    run_chain_weights(w, Tpad, Tq, Lq);
    ChainScalArgs ca{w.S.p, chain_cw.p, chain_cs.p, chain.dterms.p, w.Bpad, Tpad, Tq};
    MP_RUN(k_chain_scalars, C, Tq, nterms, ca);
    PhaseDev& ph = chain.dev;
    DevBuf<uint32_t>& gt = gtile[0];
    gt.alloc((size_t)Tq * chain.K * G_::PW, s, false);
    ChainTileArgs ta{w.P.p, gt.p, chain.dtile_src.p, w.Bpad, Tq, chain.K};
    if ((uint64_t)Tq * chain.K >= ((uint64_t)1 << 32)) throw std::runtime_error("chain verification: too many tables for one launch");
    MP_RUN(k_chain_tile, C, Tq * chain.K, 1, ta);
    run_bucket(w, ph, chain_cs.p, Tpad, chain_d8.p, (size_t)ph.b_dig_bytes, Tq, Tq, "chain verification: too many tables for one launch", gt.p, chain.K);
    FixedArgs fx{chain_cs.p, w.J.p, FB.p, ph.fjobs.p, ph.fterms.p, w.Bpad, fbg, Tpad};
    MP_RUN(k_fixed_msm, C, Tq, ph.n_f, fx);
    if (ph.n_c0) {
      CombineArgs cb0{w.J.p, w.P.p, ph.cjobs0.p, ph.cterms0.p, w.Bpad};
      MP_RUN(k_combine, C, Tq, ph.n_c0, cb0);
    }
    CombineArgs cb{w.J.p, w.P.p, ph.cjobs.p, ph.cterms.p, w.Bpad};
    MP_RUN(k_combine, C, Tq, ph.n_c, cb);
    gbad[0].alloc(Tq, s, false);
    // (the caller's words: 0 for every link of every table whose chain equation holds, MP_ERR_INTERNAL for the others until the per-link
    // verifier below has looked at them)
    ChainVerdictArgs va{w.J.p, w.direct.p, w.status.p, flag_word(false, s), gbad[0].p, w.Bpad, Tq, Lq, 0u, w.P.p, keyed ? l.pk : NO_SLOT, G, status, nullptr};
    run_chain_verdict(va);
    if (!read_flag(false)) return;
    // some equation failed: the per-link verifier gives every link of ITS tables its exact status (link j of table t: deck row j T + t,
    // shuffled deck row (j + 1) T + t -- the same index into the array one deck further on); the other tables' verdicts stand
    std::vector<uint32_t> bad, idx;
    read_words(gbad[0].p, Tq, bad);
    group_members(bad.data(), Tq, Lq, idx);
    const VArgs cv{B, decks, decks + (size_t)T * deck_bytes, proofs, status, keys, nullptr, nullptr};
    verify_subset(cv, idx, 0, false, 0, false);
  }

  // ---------------------------------------------------------------- group verification (round 4)
  // The screening pass of a LARGE batch, one level up from the merged equation of one proof: the merged equations of L independent proofs
  // are added up with weights rho_j that depend on every proof of the group (the machinery of chain verification, without its shared
  // decks), and the resulting multi-scalar multiplication -- L (4N + 11m + 8) points, 7 616 for 32 proofs of a 52-card deck -- runs on
  // the bucket-method kernel: 33 additions per term and window-sorted, wave-reduced buckets instead of 51 additions, 15 window-table
  // entries and a share of four 250-doubling chains per term on the Straus path; the n + 5 fixed bases appear once per GROUP.  The
  // verifier's half of a 52-card step is the north star's kernel from here on (Pippenger: LDS-staged digits, counting sort by wavefront
  // prefix sum).  Soundness as for the merged equation (weights = Fr::rand of ChaCha20(Blake2s(final transcript states of the group's
  // proofs)): an error in one proof cannot cancel against another's except with probability ~2^-250).  A batch in which some group
  // fails is re-verified equation by equation, so the status words are exactly those of the other paths.  Lane of (member j, group t)
  // = j T + t with T = B / L groups: the members of a group are T proofs apart.
  // round 5, the bucket kernel's memory side (kernels_bucket.hpp): contiguous point runs per group, XCD-affine items, LDS-staged points
  // (A/B hooks of round 5, compile-time like the others -- tools/ab_build.py -DMP_EXP_BK_TILE=0 ...; the product library reads no
  // environment variable.  profiles/r05b_bucket_ab.txt has what each bought)
#ifndef MP_EXP_BK_TILE
#define MP_EXP_BK_TILE 1
#endif
#ifndef MP_EXP_BK_XCD
#define MP_EXP_BK_XCD 1
#endif
#ifndef MP_EXP_BK_STAGE
#define MP_EXP_BK_STAGE 1
#endif
  static constexpr bool bk_tile = MP_EXP_BK_TILE != 0, bk_xcd_affine = MP_EXP_BK_XCD != 0, bk_stage = MP_EXP_BK_STAGE != 0;
  DevBuf<uint32_t> gtile[2];          // [lane] the group equations' point runs: T x K x 64 bytes (4 GB at 262 144 proofs of 52 cards)
  std::map<std::pair<uint32_t, bool>, std::unique_ptr<ChainPlan>> gplans;      // by (proofs per group, keyed): a failing batch alternates between two sizes
  Workspace gws;                      // lean workspace of the pipelined group pass: no window tables, no digit planes (68 KB per proof)
  // Group size: the wave-wide reduction of a window costs ~40 additions whatever the equation holds, and 10-bit windows (26 per scalar
  // instead of 32) want ~60 terms per bucket -- ~30 000 points per equation when the batch is large
  // (128 proofs of a 52-card deck, 8 of a 1 024-card one); but the kernel's 2 048 persistent waves want a dozen (equation, window) items
  // each, so a smaller batch takes smaller groups: no fewer than 2/13 of the minimum batch (945 at the default 6 144: 12 x 2 048 / 26).
  // window width of the bucket method for an MSM of K terms (mp_set_bucket_bits; 0 = by size)
  uint32_t bucket_bits_of(uint32_t K) const { return bucket_bits ? bucket_bits : bk_bits_for(K); }
  // (BLS12-377: the bucket kernel's additions over a 377-bit field spill ~200 registers at two waves per SIMD and only draw level with the
  // Straus screen -- 13.1 k against 13.1 k proofs/s at (10,30) --, so groups are off there unless mp_set_group_verify asks for them)
  // Round 6: two sizes.  group_points (30 464: 128 proofs of a 52-card deck, 10-bit windows on one wave per window, 26 additions per
  // point) is what a batch of fewer than 16 384 proofs takes under the rule below; a larger one takes equations of up to group_points_wg
  // points (243 712: 1 024 proofs of a 52-card deck) instead -- 14-bit windows on the split pipeline (kernels_bucket.hpp k_bucket_sort /
  // _acc / _reduce: 18 additions per point; 12-bit ones, 21, from 50 000 points on).
  // BLS12-377: the one-wave-per-window kernel spills 200 registers on the 14-limb field and only drew level with the Straus screen
  // (round 4), so group_points stays 0 there; the split pipeline's hot loop is spill-free (213 registers, no scratch), takes 11-bit
  // windows too on this curve (mp_set_bucket_split 11) and equations from 40 000 points on: 32 proofs of a 300-card deck at 4 096 in
  // flight, 13.3 -> 14.5 k pairs/s with the subgroup test of every point still in the call (profiles/r06o_bls_groups.txt).
  static const uint32_t GROUP_POINTS_WAVE = 30464u, GROUP_POINTS_WG = 243712u, GROUP_WG_POINTS_MIN = G_::FW > 8 ? 40000u : 50000u;
  uint32_t group_points = G_::FW > 8 ? 0u : GROUP_POINTS_WAVE;      // points per group equation aimed at (mp_set_group_verify; 0 = off)
  uint32_t group_points_wg = GROUP_POINTS_WG;                       // ... by the batches large enough for the split pipeline (0 = never)
  uint32_t group_min_batch = 6144;    // smaller batches (in 52-card proofs: x 52 / N) keep the per-proof screen
  void set_group_verify(uint32_t points, size_t min_batch) override {
    // (one knob: up to 65 535 points = equations of that size at the most, under the rule of rounds 4-5; more = the split pipeline's
    // equations for the batches large enough, the default size below them)
    group_points = std::min(points, points > 65535u ? GROUP_POINTS_WAVE : 65535u);
    group_points_wg = points > 65535u ? points : 0u;
    group_min_batch = (uint32_t)std::min<size_t>(min_batch, 0x7FFFFFFFu);
  }
  // proofs per group for a batch of B: the divisor of B nearest to the wanted size (between half and twice it) whose equation
  // fits one bucket job (65 535 points); 0 = this batch takes the per-proof screen
  uint32_t group_size_of(size_t B) const override { return B < 0x7FFFFFFFu ? group_size((uint32_t)B, false) : 0; }
  uint32_t group_size(uint32_t B, bool keyed) const {
    const uint32_t per = 4 * N + 11 * m + 8 + (keyed ? 1u : 0u);
    // (the minimum counts lanes like the work-split thresholds: a proof of N cards brings N / 52 times the points of a 52-card one)
    if ((!group_points && !group_points_wg) || !merged_verify || (uint64_t)B * N < (uint64_t)group_min_batch * 52u) return 0;
    // as many proofs per group as bring its equation nearest to group_points points, but no fewer groups than keep the persistent
    // waves busy
    const uint32_t groups_min = std::max<uint32_t>(1u, (uint32_t)(((uint64_t)group_min_batch * 2u) / 13u));
    uint32_t want = std::min<uint32_t>((group_points + per / 2) / per, B / groups_min);
    // (batches of 8/3 min_batch -- 16 384 -- 52-card proofs and more: the split pipeline's equations, if at least min_batch / 512 (12)
    // of at least GROUP_WG_POINTS_MIN points are left.  Its units are a thirty-second of a window, so sixteen equations of 1 024 proofs
    // keep the chip busy: 482 -> 504 k proofs/s at 16 384 in flight, 550 -> 586 k at 32 768, 610 -> 661 k at 65 536, 644 -> 702 k at
    // 131 072 (profiles/r06v_mid_batches.txt); below that the equations of rounds 4-5 or the per-proof screen are as fast or faster)
    const uint32_t want_wg = std::min<uint32_t>((group_points_wg + per / 2) / per, B / std::max<uint32_t>(1u, group_min_batch / 512u));
    if (group_points_wg && (uint64_t)want_wg * per >= GROUP_WG_POINTS_MIN && (uint64_t)B * N * 3u >= (uint64_t)group_min_batch * 52u * 8u)
      want = std::max(want, want_wg);
    // (under sustained rejection the groups shrink: note_group_verdicts below)
    if (want >= 4) want = std::max<uint32_t>(want >> adapt_shift, 4u);
    if (want < 2) return 0;
    for (uint32_t d = 0; d <= want; ++d)
      for (int sgn = 1; sgn >= -1; sgn -= 2) {
        const int64_t L = (int64_t)want + sgn * (int64_t)d;
        // (without the contiguous run of k_group_tile a sorted entry names member and slot: 10 bits of link -- kernels_bucket.hpp)
        if (L < 2 || 2 * L < (int64_t)want || L > 2 * (int64_t)want || (uint64_t)L * per + n + 5 > BUCKET_TERMS_MAX || (!bk_tile && L > 1023)) continue;
        if (B % (uint32_t)L == 0) return (uint32_t)L;
      }
    return 0;
  }
  // Groups that adapt to the rejection rate (round 5).  A group of L proofs fails if ANY member does: with a fraction p of bad proofs
  // spread over the batch, 1 - (1 - p)^L of the groups fail -- 72 % of the groups of 128 at p = 1 % -- and every member of a failing group
  // pays a finer pass (0.55-0.85 us per proof for a sub-group equation, 1.2 us equation by equation: profiles/r05h_rejection_strategies.txt)
  // on top of the screen that told it nothing.  So the table remembers what the last screens saw: if more than a fifth of the groups of a
  // call fail, the next call takes groups of half the size (down to 8: the screen of groups of 16 costs 0.16 us per proof more than that of
  // 128, and 15 % of them fail at p = 1 %); if fewer than 4 % fail, the size goes back up, one step per call.  Honest traffic never
  // leaves the default; under sustained 1 % rejection the step settles at ~0.8 x the honest rate instead of ~0.68 x.  Only calls with at
  // least 64 groups count (a fraction of three groups says nothing).  mp_set_group_adapt(t, 0) pins the default size.
  uint32_t adapt_shift = 0;
  bool group_adapt = true;
  void set_group_adapt(bool on) override {
    group_adapt = on;
    adapt_shift = 0;
  }
  void note_group_verdicts(uint32_t T, uint32_t failing, uint32_t L) {
    // (a fraction of three groups says nothing -- but sixteen equations of 1 024 proofs that ALL fail do: from eight groups on, and a
    // step down only on the evidence of at least four failing groups)
    if (!group_adapt || T < 8) return;
    if (T < 64 && failing > 0 && failing < 4) return;
    // (round 6: the default group is 1 024 proofs -- seven halvings down to 8.  When nineteen groups in twenty fail, the rate is far
    // beyond what the next size could clear: three halvings at once; and a call in which no group fails at all grows back two steps)
    if ((uint64_t)failing * 5 > T) {
      const uint32_t by = (uint64_t)failing * 20 >= (uint64_t)T * 19 && L >= 128 ? 3u : 1u;
      if (L >= 16) adapt_shift = std::min<uint32_t>(7u, adapt_shift + by);
    } else if ((uint64_t)failing * 25 < T && adapt_shift > 0) {
      adapt_shift -= failing == 0 && adapt_shift >= 2 ? 2u : 1u;
    }
  }
  ChainPlan& build_group_plan(uint32_t L, bool keyed) {
    const std::pair<uint32_t, bool> key(L, keyed);
    auto it = gplans.find(key);
    if (it != gplans.end()) return *it->second;
    if (gplans.size() >= 8) {      // (sizes come and go with the batch size: keep a handful)
      rt::stream_sync(ctx->stream);
      if (ctx->vstream) rt::stream_sync(ctx->vstream);
      gplans.clear();
    }
    std::unique_ptr<ChainPlan>& fresh = gplans[key];
    fresh.reset(new ChainPlan());
    ChainPlan& gplan = *fresh;
    if (keyed) ensure_keyed();
    PlanSet& q = (keyed ? psk : ps)[0];
    const VerifyLay& l = q.vplan.lay;
    gplan.ph = Phase();
    gplan.cterms.clear();
    gplan.L = L;
    gplan.keyed = keyed;
    uint32_t next_partial = 1;                  // J slot 0 = the group equation's value
    const uint32_t gbits = bucket_bits_of(L * (l.pk + (keyed ? 1u : 0u)));
    PhaseBuilder pb(gplan.ph, next_partial, FCHUNK, VCHUNK, 1u, bk_windows(R::BITS, gbits), 1u, gbits);
    pb.begin(0);
    for (uint32_t j = 0; j < L; ++j)
      for (uint32_t slot = 0; slot < l.pk + (keyed ? 1u : 0u); ++slot) {      // decks, proof points [, the proof's own key]
        // (a term's point: its index in the group's contiguous run, or the P slot | member whose lane holds it)
        pb.var((uint32_t)gplan.cterms.size(), bk_tile ? (uint32_t)gplan.cterms.size() : (slot | (j << 20)));
        gplan.cterms.push_back(ChainTerm{l.mvar + slot, j, 1, NO_SLOT, 1});
      }
    gplan.K = (uint32_t)gplan.cterms.size();
    // (the scalar of a fixed base is a sum over the members, one lane adding them up in k_chain_scalars: in runs of at most 64 members --
    // a base then appears ceil(L / 64) times in the fixed-base part -- so that no lane of that kernel runs 1 024 products in a row)
    FixedBases fb{n};
    for (uint32_t f = 0; f < fb.count(); ++f) {
      if (keyed && f == fb.pk()) continue;
      for (uint32_t j0 = 0; j0 < L; j0 += 64u) {
        pb.fixed((uint32_t)gplan.cterms.size(), f);
        gplan.cterms.push_back(ChainTerm{l.mfix + f, j0, std::min(64u, L - j0), NO_SLOT, 1});
      }
    }
    gplan.nfix = (uint32_t)gplan.cterms.size() - gplan.K;
    pb.end();
    gplan.nJ = next_partial;
    gplan.dev.upload(gplan.ph, ctx->stream);
    gplan.dterms.upload(gplan.cterms, ctx->stream);
    return gplan;
  }
  // the group pass on the context's CURRENT lane; the flag word (host_flag, or vflag) is raised if some group needs a closer look
  // and gbad[t] says which (null: nobody asks)
  void verify_group_pass(const VArgs& v, uint32_t L, bool vlane, uint32_t* host_flag, uint32_t* gbad_out) {
    const uint32_t B = v.B, T = B / L, Tpad = (T + 63u) & ~63u;
    const bool keyed = v.keys != nullptr || v.kset != nullptr;
    const uint8_t* keys = v.keys;
    ChainPlan& gplan = build_group_plan(L, keyed);
    PlanSet& q = (keyed ? psk : ps)[0];
    const VerifyLay& l = q.vplan.lay;
    rt::Stream s = ctx->stream;
    // (on the caller's stream the pass borrows the prover's arenas, as the per-proof screen does: nothing of a prove call outlives it;
    // the verify lane has lean arenas of its own)
    Workspace& w = vlane ? gws : ws;
    w.fw = G_::FW;
    w.ensure(B, l.nS, l.nP, std::max(gplan.nJ, 8u), 0, 0, nwin, stage_words_needed(), s, 0);
    rt::dzero(w.status.p, (size_t)w.Bpad * 4, s);
    {
      LoadPointsArgs a{v.decks, w.P.p, w.status.p, w.Bpad, 2 * N, l.deck};
      MP_RUN(k_load_points, C, B, 2 * N, a);
      LoadPointsArgs b{v.shuf, w.P.p, w.status.p, w.Bpad, 2 * N, l.shuf};
      MP_RUN(k_load_points, C, B, 2 * N, b);
      ProofIoArgs pa{const_cast<uint8_t*>(v.proofs), w.S.p, w.P.p, w.status.p, q.vwire.p, w.Bpad, (uint32_t)proof_size_bytes(m, n, G_::PB)};
      MP_RUN(k_load_proof, C, B, (uint32_t)q.vplan.wire.size(), pa);
      if (v.kset) keys = gather_keys(B, v.kset, v.kidx, w.status.p, vlane);
      if (keyed) {
        LoadPointsArgs ka{keys, w.P.p, w.status.p, w.Bpad, 1, l.pk};
        MP_RUN(k_load_points, C, B, 1, ka);
      }
      check_verify_inputs(w, B, l, keyed);
    }
    {
      VerifyFsArgs a{};
      a.st = statement_args(w, l.deck, l.shuf, l.cA, l.x, keyed ? l.pk : NO_SLOT);
      a.l = l;
      a.merge = 1u;
      run_verify_fs(a, B);
    }
    const uint32_t nterms = gplan.K + gplan.nfix;
    chain_cw.alloc((size_t)L * Tpad * 8, s, false);
    chain_cs.alloc((size_t)nterms * Tpad * 8, s);
    chain_d8.alloc((size_t)gplan.dev.b_dig_bytes * Tpad, s);
    {
      // the group weights need the transcripts' final states and nothing else, and one lane per GROUP hashes them (2 048 lanes, 64
      // BLAKE2s blocks in a row: 2.5 ms of latency at 262 144 proofs): on `side`, beside the coefficient programs of the proofs
      SideGuard wguard{ctx, false};
      rt::event_record(ctx->ev_fork, s);
      rt::stream_wait(ctx->side, ctx->ev_fork);
      {
        struct Restore {
          mp_ctx* c;
          rt::Stream keep;
          ~Restore() { c->stream = keep; }
        } restore{ctx, s};
        ctx->stream = ctx->side;
        run_chain_weights(w, Tpad, T, L);
        rt::event_record(ctx->ev_tab, ctx->side);
      }
      VerifyScalArgs sa{w.S.p, w.P.p, w.direct.p, l, q.vplan.cm, w.Bpad};
      MP_RUN(k_verify_scal, C, B, n + 2, sa);
      VerifyMergeArgs ma{w.S.p, q.mjobs.p, q.mpairs.p, w.Bpad};
      MP_RUN(k_verify_merge, C, B, (uint32_t)q.vplan.mjobs.size(), ma);
      rt::stream_wait(s, ctx->ev_tab);
      wguard.joined = true;
    }
    ChainScalArgs ca{w.S.p, chain_cw.p, chain_cs.p, gplan.dterms.p, w.Bpad, Tpad, T};
    MP_RUN(k_chain_scalars, C, T, nterms, ca);
    PhaseDev& ph = gplan.dev;
    const uint32_t* tile = nullptr;
    if (bk_tile) {
      const uint32_t per = l.pk + (keyed ? 1u : 0u);
      DevBuf<uint32_t>& gt = gtile[vlane ? 1 : 0];
      gt.alloc((size_t)T * gplan.K * G_::PW, s, false);
      GroupTileArgs ta{w.P.p, gt.p, w.Bpad, T, per, gplan.K};
      MP_RUN(k_group_tile, C, B, per, ta);
      tile = gt.p;
    }
    run_bucket(w, ph, chain_cs.p, Tpad, chain_d8.p, (size_t)ph.b_dig_bytes, T, T, "group verification: too many groups for one launch", tile, gplan.K);
    FixedArgs fx{chain_cs.p, w.J.p, FB.p, ph.fjobs.p, ph.fterms.p, w.Bpad, fbg, Tpad};
    MP_RUN(k_fixed_msm, C, T, ph.n_f, fx);
    if (ph.n_c0) {
      CombineArgs cb0{w.J.p, w.P.p, ph.cjobs0.p, ph.cterms0.p, w.Bpad};
      MP_RUN(k_combine, C, T, ph.n_c0, cb0);
    }
    CombineArgs cb{w.J.p, w.P.p, ph.cjobs.p, ph.cterms.p, w.Bpad};
    MP_RUN(k_combine, C, T, ph.n_c, cb);
    uint32_t* fl = host_flag;
    if (!fl) fl = flag_word(vlane, s);
    // (the caller's words: zeros for the members of the groups whose equation holds -- final --, MP_ERR_INTERNAL for those of the others
    // until the finer passes have given each its own word)
    ChainVerdictArgs va{w.J.p, w.direct.p, w.status.p, fl, gbad_out, w.Bpad, T, L, 0u, w.P.p, NO_SLOT, 1u, v.status, nullptr};
    run_chain_verdict(va);
  }

  // ---------------------------------------------------------------- building blocks (ad-hoc plans)
  struct Adhoc {
    Phase ph;
    PhaseDev dev;
    Workspace w;
  };
  void run_adhoc(Adhoc& ad, uint32_t B, uint32_t nS, uint32_t nP, uint32_t nJ) {
    ad.dev.upload(ad.ph, ctx->stream);
    ad.w.fw = G_::FW;
    ad.w.ensure(B, nS, nP, nJ, ad.ph.n_dslots, ad.ph.n_tslots, nwin, 4, ctx->stream, ad.ph.b_dig_bytes);
  }

  void remask_host(size_t count, const uint8_t* cards, const uint8_t* rho, uint8_t* out) override {
    // `count` cards = `count` batch lanes with N = 1
    rt::Stream s = ctx->stream;
    const uint32_t B = (uint32_t)count;
    Adhoc ad;
    run_adhoc(ad, B, 1, 4, 4);
    Workspace& w = ad.w;
    DevBuf<uint8_t> din, drho, dout;
    din.alloc(count * 2 * G_::PB, s, false); drho.alloc(count * 32, s, false); dout.alloc(count * 2 * G_::PB, s, false);
    rt::h2d(din.p, cards, count * 2 * G_::PB, s);
    rt::h2d(drho.p, rho, count * 32, s);
    rt::dzero(w.status.p, (size_t)w.Bpad * 4, s);
    FixedBases fb{n};
    LoadPointsArgs a{din.p, w.P.p, w.status.p, w.Bpad, 2, 0};
    MP_RUN(k_load_points, C, B, 2, a);
    LoadScalarsArgs sa{drho.p, w.S.p, w.status.p, w.Bpad, 1, 0};
    MP_RUN(k_load_scalars, C, B, 1, sa);
    RemaskArgs ra{w.S.p, w.P.p, w.J.p, FB.p, nullptr, w.Bpad, 1, 0, 0, 2, fb.G(), fb.pk(), fbg, 0, nullptr, nullptr, 0, 0, 0};
    MP_RUN(k_remask, C, B, 2, ra);
    normalize_flat(w.J.p + j_off<C>(2, w.Bpad, 0), w.P.p + p_off<C>(2, w.Bpad, 0), w.NS.p, (size_t)2 * w.Bpad);
    StorePointsArgs st{dout.p, w.P.p, w.Bpad, 2, 2};
    MP_RUN(k_store_points, C, B, 2, st);
    std::vector<int32_t> hs(B);
    rt::d2h(out, dout.p, count * 2 * G_::PB, s);
    rt::d2h(hs.data(), w.status.p, (size_t)B * 4, s);
    rt::stream_sync(s);
    for (auto v : hs)
      if (v < 0) throw std::invalid_argument("remask: bad encoding in input");
  }

  void msm_host(size_t n_msm, size_t k, const uint8_t* scalars, const uint8_t* points, uint8_t* out) override {
    rt::Stream s = ctx->stream;
    const uint32_t B = (uint32_t)n_msm, K = (uint32_t)k;
    Adhoc ad;
    uint32_t next_partial = K + 1;
    {
      const uint32_t abits = bucket_bits_of(K);
      PhaseBuilder pb(ad.ph, next_partial, FCHUNK, VCHUNK, bucket_min, bk_windows(R::BITS, abits), 1u, abits);
      pb.begin(K);
      for (uint32_t t = 0; t < K; ++t) pb.var(t, t);
      pb.end();
      pb.normalize(K, 1);
    }
    run_adhoc(ad, B, K, K + 1, next_partial);
    Workspace& w = ad.w;
    DevBuf<uint8_t> dsc, dpt, dout;
    dsc.alloc(n_msm * k * 32, s, false); dpt.alloc(n_msm * k * G_::PB, s, false); dout.alloc(n_msm * G_::PB, s, false);
    rt::h2d(dsc.p, scalars, n_msm * k * 32, s);
    rt::h2d(dpt.p, points, n_msm * k * G_::PB, s);
    rt::dzero(w.status.p, (size_t)w.Bpad * 4, s);
    LoadPointsArgs la{dpt.p, w.P.p, w.status.p, w.Bpad, K, 0};
    MP_RUN(k_load_points, C, B, K, la);
    LoadScalarsArgs sa{dsc.p, w.S.p, w.status.p, w.Bpad, K, 0};
    MP_RUN(k_load_scalars, C, B, K, sa);
    run_phase(ad.dev, w, B);
    StorePointsArgs so{dout.p, w.P.p, w.Bpad, 1, K};
    MP_RUN(k_store_points, C, B, 1, so);
    std::vector<int32_t> hs(B);
    rt::d2h(out, dout.p, n_msm * G_::PB, s);
    rt::d2h(hs.data(), w.status.p, (size_t)B * 4, s);
    rt::stream_sync(s);
    for (auto v : hs)
      if (v < 0) throw std::invalid_argument("msm: bad encoding in input");
  }

  void commit_host(size_t count, size_t len, const uint8_t* values, const uint8_t* r, uint8_t* out) override {
    rt::Stream s = ctx->stream;
    const uint32_t B = (uint32_t)count, L = (uint32_t)len;
    FixedBases fb{n};
    Adhoc ad;
    uint32_t next_partial = 1;
    {
      PhaseBuilder pb(ad.ph, next_partial, FCHUNK, VCHUNK);
      pb.begin(0);
      for (uint32_t t = 0; t < L; ++t) pb.fixed(t, fb.ck(t));
      pb.fixed(L, fb.H());
      pb.end();
      pb.normalize(0, 1);
    }
    run_adhoc(ad, B, L + 1, 1, next_partial);
    Workspace& w = ad.w;
    DevBuf<uint8_t> dv, dr, dout;
    dv.alloc(std::max<size_t>(count * len * 32, 4), s, false); dr.alloc(count * 32, s, false); dout.alloc(count * G_::PB, s, false);
    if (len) rt::h2d(dv.p, values, count * len * 32, s);
    rt::h2d(dr.p, r, count * 32, s);
    rt::dzero(w.status.p, (size_t)w.Bpad * 4, s);
    if (L) {
      LoadScalarsArgs sa{dv.p, w.S.p, w.status.p, w.Bpad, L, 0};
      MP_RUN(k_load_scalars, C, B, L, sa);
    }
    LoadScalarsArgs sr{dr.p, w.S.p, w.status.p, w.Bpad, 1, L};
    MP_RUN(k_load_scalars, C, B, 1, sr);
    run_phase(ad.dev, w, B);
    StorePointsArgs so{dout.p, w.P.p, w.Bpad, 1, 0};
    MP_RUN(k_store_points, C, B, 1, so);
    std::vector<int32_t> hs(B);
    rt::d2h(out, dout.p, count * G_::PB, s);
    rt::d2h(hs.data(), w.status.p, (size_t)B * 4, s);
    rt::stream_sync(s);
    for (auto v : hs)
      if (v < 0) throw std::invalid_argument("commit: bad encoding in input");
  }

  void plan_stats(uint64_t out[16]) override {
    for (int i = 0; i < 16; ++i) out[i] = 0;
    uint64_t bucket_terms = 0, bucket_jobs = 0;
    auto add = [&](const Phase& ph, uint64_t* o) {
      o[0] += ph.fterms.size(); o[1] += ph.vterms.size(); o[2] += ph.fjobs.size(); o[3] += ph.vjobs.size();
      o[4] += ph.tables.size(); o[5] += ph.cterms.size() + ph.cterms2.size() + ph.cterms0.size();
      bucket_terms += ph.bterms.size();
      bucket_jobs += ph.bjobs.size();
    };
    for (int i = 0; i < 6; ++i) add(ps[0].pplan.ph[i], out);
    add(merged_verify ? ps[0].vplan.mph : ps[0].vplan.ph, out + 6);    // what an honest batch executes
    out[12] = nwin; out[13] = fbg.windows;
    out[14] = N | ((uint64_t)(ps[0].pplan.toom.E ? m : 0u) << 32);      // high word: m if k_toom_points runs in this plan
    out[15] = bucket_terms | (bucket_jobs << 32);      // variable-base terms / MSMs on the bucket kernel (prove + verify)
  }
  // ---------------------------------------------------------------- sigma protocols
  void sigma_host(bool prove, size_t B_, uint32_t nb, const uint8_t* bases, const uint8_t* publics, const uint8_t* witness,
                  const uint8_t* fs_init, const uint8_t* seeds, uint8_t* proofs, int32_t* status) override {
    rt::Stream s = ctx->stream;
    const uint32_t B = (uint32_t)B_;
    const SigmaLay l = make_sigma_lay(nb);
    const size_t psz = (size_t)nb * G_::PB + 32;
    Adhoc ad;
    uint32_t next_partial = l.chk + nb;
    {
      PhaseBuilder pb(ad.ph, next_partial, FCHUNK, VCHUNK);
      if (prove) {
        for (uint32_t i = 0; i < nb; ++i) {
          pb.begin(l.A + i);
          pb.var(l.r, l.g + i);
          pb.end();
        }
        pb.normalize(l.A, nb);
      } else {
        for (uint32_t i = 0; i < nb; ++i) {
          pb.begin(l.chk + i);
          pb.var(l.z, l.g + i);
          pb.var(l.negc, l.a + i);
          pb.var(l.minus_one, l.A + i);
          pb.end();
        }
      }
    }
    ad.dev.upload(ad.ph, s);
    ad.w.fw = G_::FW;
    ad.w.ensure(B, 6, 3 * nb, next_partial, ad.ph.n_dslots, ad.ph.n_tslots, nwin, (3 * nb * (G_::PB + 1) + 32) / 4 + 4, s);
    Workspace& w = ad.w;
    DevBuf<uint8_t> dg, da, dx, dfs, dseed, dpf;
    DevBuf<int32_t> dst;
    dg.alloc((size_t)B * nb * G_::PB, s, false); da.alloc((size_t)B * nb * G_::PB, s, false); dfs.alloc((size_t)B * 32, s, false);
    dpf.alloc((size_t)B * psz, s, false);
    rt::h2d(dg.p, bases, (size_t)B * nb * G_::PB, s);
    rt::h2d(da.p, publics, (size_t)B * nb * G_::PB, s);
    rt::h2d(dfs.p, fs_init, (size_t)B * 32, s);
    rt::dzero(w.status.p, (size_t)w.Bpad * 4, s);
    LoadPointsArgs lg{dg.p, w.P.p, w.status.p, w.Bpad, nb, l.g};
    MP_RUN(k_load_points, C, B, nb, lg);
    LoadPointsArgs la{da.p, w.P.p, w.status.p, w.Bpad, nb, l.a};
    MP_RUN(k_load_points, C, B, nb, la);
    const FsDev f{w.stage.p, w.seed.p, w.Bpad};
    if (prove) {
      dx.alloc((size_t)B * 32, s, false); dseed.alloc((size_t)B * 32, s, false);
      rt::h2d(dx.p, witness, (size_t)B * 32, s);
      rt::h2d(dseed.p, seeds, (size_t)B * 32, s);
      LoadScalarsArgs lx{dx.p, w.S.p, w.status.p, w.Bpad, 1, l.x};
      MP_RUN(k_load_scalars, C, B, 1, lx);
      SigmaInitArgs ia{w.S.p, dseed.p, l, w.Bpad, f, w.P.p, dfs.p};
      MP_RUN(k_sigma_init, C, B, 1, ia);
      run_phase(ad.dev, w, B);
      SigmaFsArgs fa{f, w.S.p, w.P.p, dfs.p, l, 1};
      MP_RUN(k_sigma_fs, C, B, 1, fa);
      SigmaIoArgs io{dpf.p, w.S.p, w.P.p, w.status.p, l, w.Bpad};
      MP_RUN(k_sigma_store, C, B, nb + 1, io);
      rt::d2h(proofs, dpf.p, (size_t)B * psz, s);
    } else {
      rt::h2d(dpf.p, proofs, (size_t)B * psz, s);
      SigmaIoArgs io{dpf.p, w.S.p, w.P.p, w.status.p, l, w.Bpad};
      MP_RUN(k_sigma_load, C, B, nb + 1, io);
      SigmaFsArgs fa{f, w.S.p, w.P.p, dfs.p, l, 0};
      MP_RUN(k_sigma_fs, C, B, 1, fa);
      run_phase(ad.dev, w, B);
      SigmaVerdictArgs va{w.J.p, w.status.p, l, w.Bpad, nb == 1 ? 5 : 6};
      MP_RUN(k_sigma_verdict, C, B, 1, va);
    }
    rt::d2h(status, w.status.p, (size_t)B * 4, s);
    rt::stream_sync(s);
  }

  void census(uint64_t* pt, uint64_t* vt, uint64_t* po, uint64_t* vo) override {
    auto count = [&](const Phase& ph, uint64_t& terms, uint64_t& ops) {
      terms += ph.fterms.size() + ph.vterms.size();
      ops += (uint64_t)ph.fterms.size() * fbg.windows;                      // mixed additions
      ops += (uint64_t)ph.vterms.size() * nwin;                             // mixed additions
      ops += (uint64_t)ph.vjobs.size() * (nwin - 1) * VB_WINDOW_BITS;       // doublings
      ops += (uint64_t)ph.tables.size() * (VB_ENTRIES - 1);                 // table construction (affine additions)
      ops += ph.cterms.size() + ph.cterms2.size() + ph.cterms0.size();      // combines
      terms += ph.bterms.size();
      const uint32_t bwin = bk_windows(R::BITS, ph.b_bits), bnb = bk_buckets(ph.b_bits) / 64;
      ops += (uint64_t)ph.bterms.size() * bwin;                             // bucket method: one mixed addition per term and window
      ops += (uint64_t)ph.bjobs.size() * bwin * (13 + 2 * bnb - 3 + ph.b_bits + 1);   // wave-wide reduction + fold
    };
    uint64_t t = 0, o = 0;
    for (int i = 0; i < 6; ++i) count(ps[0].pplan.ph[i], t, o);
    t += 2 * N;
    o += (uint64_t)2 * N * (fbg.windows + 1);  // remask
    *pt = t; *po = o;
    t = 0; o = 0;
    count(merged_verify ? ps[0].vplan.mph : ps[0].vplan.ph, t, o);
    *vt = t; *vo = o;
  }
};

// DLCards::setup [REF mod.rs:105-121]: n + 3 independent `C::rand` points (setup_host.hpp) -- host work, once per table
template <class C>
static void aff_to_wire_host(const Aff<C>& a, uint8_t* out) {
  alignas(8) uint8_t tmp[Geo<C>::PB];
  aff_to_wire<C>(a, tmp);
  memcpy(out, tmp, Geo<C>::PB);
}
template <class C>
static int setup_device(mp_ctx* ctx, uint32_t m, uint32_t n, const uint8_t seed[32], uint8_t* out) {
  (void)ctx;
  (void)m;
  setup_points_host<C>(n, seed, out, &aff_to_wire_host<C>, Geo<C>::PB);
  return MP_OK;
}

// ---- on-device point decompression (kernels_decompress.hpp): square-root tables of the base field, once per context (host work with
// the kernels' own field code), then one lane per point
template <class C>
static void build_sqrt_tables(mp_ctx* ctx) {
  typedef typename C::FqP F;
  constexpr int W = F::NW;
  uint32_t q[W], half[W], e[W];
  for (int i = 0; i < W; ++i) q[i] = F::MOD[i];
  q[0] -= 1u;                                                // p - 1
  auto shr1 = [](uint32_t* r, const uint32_t* x) {
    for (int i = 0; i < W; ++i) r[i] = (x[i] >> 1) | (i + 1 < W ? x[i + 1] << 31 : 0u);
  };
  shr1(half, q);                                             // (p - 1) / 2
  uint32_t S = 0;
  while (!(q[0] & 1u)) {
    shr1(q, q);
    ++S;
  }
  for (int i = 0; i < W; ++i) e[i] = q[i];
  e[0] -= 1u;                                                // q odd
  shr1(e, e);                                                // (q - 1) / 2
  const uint32_t w = S % 4 == 0 ? 4u : (S % 2 == 0 ? 2u : 1u), k = S / w, nd = 1u << w;
  const Fe<F> one = fe_one<F>(), minus_one = fe_neg<F>(one);
  Fe<F> z = fe_from_u32<F>(2);
  for (uint32_t c = 2; !fe_eq<F>(fe_pow_host<F>(z, half, W), minus_one); ++c) z = fe_from_u32<F>(c + 1);      // smallest non-residue
  const Fe<F> g = fe_pow_host<F>(z, q, W), ginv = fe_inv<F>(g);
  std::vector<uint32_t> t_ghalf((size_t)k * nd * W, 0u), t_hh((size_t)nd * W);
  Fe<F> gi = ginv, gh = ginv;                                // gi = ginv^(2^(w i)); gh = ginv^(2^(w i - 1)) for i >= 1
  for (uint32_t i = 0; i < k; ++i) {
    if (i) {
      gh = gi;                                               // still ginv^(2^(w (i-1)))
      for (uint32_t j = 0; j + 1 < w; ++j) gh = fe_sqr<F>(gh);
      gi = fe_sqr<F>(gh);
    }
    Fe<F> ph = one;
    for (uint32_t d = 0; d < nd; ++d) {
      if (i) {
        fe_pack<F>(ph, &t_ghalf[((size_t)i * nd + d) * W]);
        ph = fe_mul<F>(ph, gh);
      } else if (!(d & 1u)) {
        fe_pack<F>(ph, &t_ghalf[(size_t)d * W]);             // ginv^(d / 2)
        ph = fe_mul<F>(ph, ginv);
      }
    }
  }
  // R[c][d] = ginv^(d 2^(S - w c)), c = 1 .. k (c = k: ginv itself; row 1 is h^-d, unused; row 0 empty)
  std::vector<uint32_t> t_rr((size_t)(k + 1) * nd * W, 0u);
  {
    Fe<F> base = ginv;                                       // ginv^(2^(S - w c)) for c = k, k-1, ..
    for (uint32_t c = k; c >= 1; --c) {
      Fe<F> pr = one;
      for (uint32_t d = 0; d < nd; ++d) {
        fe_pack<F>(pr, &t_rr[((size_t)c * nd + d) * W]);
        pr = fe_mul<F>(pr, base);
      }
      for (uint32_t j = 0; j < w; ++j) base = fe_sqr<F>(base);
    }
  }
  ctx->sq_rr.upload(t_rr, ctx->stream);
  Fe<F> h = g;
  for (uint32_t j = 0; j < S - w; ++j) h = fe_sqr<F>(h);      // order 2^w
  Fe<F> hp = one;
  for (uint32_t d = 0; d < nd; ++d) {
    fe_pack<F>(hp, &t_hh[(size_t)d * W]);
    hp = fe_mul<F>(hp, h);
  }
  ctx->sq_ghalf.upload(t_ghalf, ctx->stream);
  ctx->sq_hh.upload(t_hh, ctx->stream);
  uint32_t ebits = 32 * W;
  while (ebits && !((e[(ebits - 1) >> 5] >> ((ebits - 1) & 31)) & 1u)) --ebits;
  ctx->sq_geom[0] = S; ctx->sq_geom[1] = w; ctx->sq_geom[2] = k; ctx->sq_geom[3] = ebits;
  for (int i = 0; i < 12; ++i) ctx->sq_exp[i] = i < W ? e[i] : 0u;
  rt::stream_sync(ctx->stream);
}
template <class C>
static int decompress_device(mp_ctx* ctx, size_t groups, uint32_t per_group, uint32_t prefix, const uint8_t* d_in, uint8_t* d_out,
                             int32_t* d_status) {
  if (!ctx->sq_geom[0]) build_sqrt_tables<C>(ctx);
  if ((uint64_t)groups * per_group >= ((uint64_t)1 << 32)) return fail(MP_ERR_BAD_ARGUMENT, "point decompression: too many points for one call");
  rt::dzero(d_status, groups * sizeof(int32_t), ctx->stream);
  DecompressArgs a{};
  a.in = d_in; a.out = d_out; a.status = d_status;
  a.ghalf = ctx->sq_ghalf.p; a.hh = ctx->sq_hh.p; a.rr = ctx->sq_rr.p;
  a.per_group = per_group; a.prefix = prefix;
  a.g = SqrtGeom{ctx->sq_geom[0], ctx->sq_geom[1], ctx->sq_geom[2], ctx->sq_geom[3]};
  for (int i = 0; i < 12; ++i) a.e[i] = ctx->sq_exp[i];
  // launches of up to 2^20 points: the scratch column of a point (its chain of k - 1 powers) is 1.5 KB on the STARK prime
  const size_t total = groups * per_group, chunk = std::min<size_t>(total, (size_t)1 << 20);
  if (a.g.k > 1) ctx->sq_chain.alloc((size_t)(a.g.k - 1) * chunk * C::FqP::NW, ctx->stream, false);
  a.chain = ctx->sq_chain.p;
  for (size_t off = 0; off < total; off += chunk) {
    a.first = (uint32_t)off;
    a.lanes = (uint32_t)std::min(chunk, total - off);
    MP_RUN(k_decompress, C, a.lanes, 1, a);
  }
  return MP_OK;
}

}  // namespace mp

#define MP_DEFINE_CURVE(NAME)                                                                                  \
  namespace mp {                                                                                               \
  mp_table* make_table_##NAME(mp_ctx* ctx, uint32_t m, uint32_t n, const uint8_t* params, const uint8_t* pk,  \
                              uint32_t fb_bits, int* rc) {                                                     \
    std::unique_ptr<Table<NAME>> p(new Table<NAME>());      /* (an init that throws -- out of memory -- gives everything back) */ \
    *rc = p->init(ctx, m, n, params, pk, fb_bits);                                                             \
    return p.release();                                                                                        \
  }                                                                                                            \
  int setup_##NAME(mp_ctx* ctx, uint32_t m, uint32_t n, const uint8_t seed[32], uint8_t* out) {                \
    return setup_device<NAME>(ctx, m, n, seed, out);                                                           \
  }                                                                                                            \
  long ser_points_##NAME(bool de, size_t count, const uint8_t* in, uint8_t* out) {                             \
    return Ser<NAME>::points(de, count, in, out);                                                              \
  }                                                                                                            \
  bool ser_scalars_ok_##NAME(size_t count, const uint8_t* in) { return Ser<NAME>::scalars_ok(count, in); }     \
  int decompress_dev_##NAME(mp_ctx* ctx, size_t groups, uint32_t per_group, uint32_t prefix, const uint8_t* d_in,  \
                            uint8_t* d_out, int32_t* d_status) {                                               \
    return decompress_device<NAME>(ctx, groups, per_group, prefix, d_in, d_out, d_status);                     \
  }                                                                                                            \
  }
