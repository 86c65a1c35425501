// Data layout of one batch of B independent shuffle proofs in HBM, and the static "job tables" that
// describe every multi-scalar multiplication of the Bayer-Groth prover / verifier for given (m, n).
//
// Arenas are SLOT-MAJOR: element `slot` of proof `b` lives at arena[(slot * Bpad + b) * words], so the 64
// lanes of a wave (64 consecutive proofs, same slot) read one contiguous 2-6 KB run -- every global access
// of the MSM kernels is coalesced.  Three arenas: S (Fr scalars, 8 words, Montgomery), P (affine points,
// 16 words), J (Jacobian points, 24 words; slots [0, nP) mirror P, slots >= nP are partial sums).
//
// The reference computes these quantities inside `shuffle::ShuffleArgument::prove/verify`
// [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:409-415, 437-442]; naming follows
// oracle/py/mp_oracle.py (transcript v1).
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <vector>

#include "rt.hpp"

namespace mp {

// ---- device-visible descriptors -------------------------------------------------------------------
struct Job {
  uint32_t out;    // J slot written
  uint32_t begin;  // first term
  uint32_t count;  // number of terms
};
struct Term {
  uint32_t s;  // fixed: S slot of the scalar        | var: digit slot        | combine: J slot or (AFF_FLAG | P slot)
  uint32_t b;  // fixed: index of the fixed base     | var: table slot        | combine: unused
};
// bucket-method MSM (kernels_bucket.hpp): one job = one large MSM; its terms are {S slot, P slot} in Phase::bterms
struct BJob {
  uint32_t out;        // J slot of the folded result
  uint32_t win_first;  // J slots [win_first, win_first + windows): per-window results
  uint32_t begin;      // first term
  uint32_t count;      // number of terms K
  uint32_t kpad;       // K rounded up to a multiple of 64
  uint32_t dig_off;    // offset (in digits) of the job's digits inside a proof's digit block (window w at + w * kpad)
};
struct BTermPos {
  uint32_t pos;        // dig_off of the job + index of the term in the job
  uint32_t kpad;
};
static const uint32_t AFF_FLAG = 0x80000000u;   // combine term: P (affine) slot instead of J slot
static const uint32_t NEG_FLAG = 0x40000000u;   // combine term: subtract instead of add
static const uint32_t SLOT_MASK = 0x3FFFFFFFu;
static const uint32_t NO_SLOT = 0xFFFFFFFFu;

// fixed bases of one table context: ck_0..ck_{n-1}, H, G, pk, gen, gsum
struct FixedBases {
  uint32_t n;
  MP_HD uint32_t ck(uint32_t j) const { return j; }
  MP_HD uint32_t H() const { return n; }
  MP_HD uint32_t G() const { return n + 1; }
  MP_HD uint32_t pk() const { return n + 2; }
  MP_HD uint32_t gen() const { return n + 3; }
  MP_HD uint32_t gsum() const { return n + 4; }
  MP_HD uint32_t count() const { return n + 5; }
};

// fixed-base tables: window width is a property of the table context (FbGeom in kernels_msm.hpp: 8, 16, 20 or 21 bits)
static const int VB_WINDOW_BITS = 5;     // variable-base (Straus) signed windows: digits in [-15, 16]
static const int VB_ENTRIES = 16;
static const uint32_t KEY_WINDOWS = 52;  // >= vb_windows(scalar bits) of every curve: window bases of a per-proof key
static inline int vb_windows(int scalar_bits) { return (scalar_bits + 1 + VB_WINDOW_BITS - 1) / VB_WINDOW_BITS; }
// Window-split Straus jobs (round 4).  The nwin windows of a variable-base job are dealt to k lanes: lane r runs the windows
// [vsplit_lo(r), vsplit_lo(r + 1)) as a Straus chain of its own (5 (windows - 1) doublings instead of 5 (nwin - 1)) and a fold
// R = sum_r 2^(5 vsplit_lo(r)) S_r -- once per MSM, after the partial sums of the MSM's sub-jobs have been added up range by
// range -- puts the pieces together (k_bucket_fold with Straus geometry).  k times the lanes of a job for < 250 extra doublings per
// MSM output: what a batch too small to fill the chip wants instead of ever smaller sub-jobs with a full doubling chain each.
MP_HD uint32_t vsplit_lo(uint32_t r, uint32_t k, uint32_t nwin) { return r * nwin / k; }
static const uint32_t VSPLIT_MAX = 16;

// One phase = everything that can run between two Fiat-Shamir squeeze points.
struct Phase {
  std::vector<Term> recode;   // {S slot, digit slot}
  std::vector<Term> tables;   // {P slot, table slot}
  std::vector<Job> fjobs, vjobs, cjobs, cjobs2;     // cjobs2: combine jobs that consume outputs of cjobs (run after them)
  std::vector<Term> fterms, vterms, cterms, cterms2;
  std::vector<Job> cjobs0;                          // group sums of MSMs with many partials (run before cjobs: a two-level tree)
  std::vector<Term> cterms0;
  std::vector<std::pair<uint32_t, uint32_t>> normalize;  // [first slot, count) J -> P
  uint32_t n_dslots = 0, n_tslots = 0;
  // window-split Straus jobs (vsplit > 1): job c writes its vsplit range sums to the J slots [out, out + vsplit); wcjobs add up the
  // range sums of an MSM's sub-jobs range by range; wjobs fold the vsplit range sums of an MSM (BJob: out, win_first, count = vsplit)
  uint32_t vsplit = 1;
  std::vector<Job> wcjobs;
  std::vector<Term> wcterms;
  std::vector<BJob> wjobs;
  std::vector<BJob> bjobs;    // MSMs large enough for the bucket method
  std::vector<Term> bterms;   // {S slot, P slot}
  std::vector<BTermPos> bpos;
  uint32_t b_dig_bytes = 0;   // bucket digits (int16 each) per proof
  uint32_t b_kpad_max = 0;
  uint32_t b_bits = 8;        // window width of the phase's bucket jobs (kernels_bucket.hpp: 8 .. 11)
};

static const size_t COMBINE_TREE_MIN = 12;
static const size_t BUCKET_TERMS_MAX = 589824;      // terms of one bucket job: 24 sorted runs of 24 576 terms (kernels_bucket.hpp BK_CHUNKS_MAX x BK_CHUNK; a sorted entry holds 24 bits of term index); larger MSMs stay on the Straus kernel
// Host-side builder: msm(out) { fixed(..) var(..) addend(..) } -> chunked sub-jobs + one combine job.
class PhaseBuilder {
 public:
  // bucket_min: MSMs with at least this many variable-base terms go to the bucket kernel (0 = never); bwin = its windows of bbits bits
  // vsplit: lanes per variable-base (Straus) sub-job, each with a share of the windows (1 = one lane runs all of them)
  PhaseBuilder(Phase& ph, uint32_t& next_partial, uint32_t fchunk, uint32_t vchunk, uint32_t bucket_min = 0, uint32_t bwin = 0,
               uint32_t vsplit = 1, uint32_t bbits = 8)
      : ph_(ph), next_partial_(next_partial), fchunk_(fchunk), vchunk_(vchunk), bucket_min_(bucket_min), bwin_(bwin),
        vsplit_(vsplit < 1 ? 1 : (vsplit > VSPLIT_MAX ? VSPLIT_MAX : vsplit)) {
    ph_.vsplit = vsplit_;
    ph_.b_bits = bbits;
  }
  void begin(uint32_t out_slot) {
    out_ = out_slot;
    f_.clear();
    v_.clear();
    a_.clear();
  }
  void fixed(uint32_t sslot, uint32_t base) { f_.push_back(Term{sslot, base}); }
  // (digit / table slots are only created when the MSM is closed as a Straus job: a bucket job needs neither)
  void var(uint32_t sslot, uint32_t pslot) { v_.push_back(Term{sslot, pslot}); }
  void addend(uint32_t pslot, bool negate = false) { a_.push_back(AFF_FLAG | (negate ? NEG_FLAG : 0u) | pslot); }
  // add (or subtract) a Jacobian result of this phase: the output of a fixed / var job, or -- when the consuming msm is
  // closed with end(late = true) -- the output of a first-stage combine job
  void addend_j(uint32_t jslot, bool negate = false) { a_.push_back((negate ? NEG_FLAG : 0u) | jslot); }
  uint32_t new_partial() { return next_partial_++; }
  void end(bool late = false) {
    // One lane adds up the partial sums of an MSM one after the other (k_combine): with the fine splits a long MSM
    // (n = 150 commitments, 4N-term verifier equations) would turn into hundreds of partials and a serial chain longer than
    // the one the split was meant to shorten -- cap the partials per MSM and kind.
    const size_t MAXP = 128;
    size_t fchunk_ = this->fchunk_, vchunk_ = this->vchunk_;
    const bool bucket = bucket_min_ && v_.size() >= bucket_min_ && v_.size() <= BUCKET_TERMS_MAX;
    if (!bucket)
      for (Term& t : v_) t = Term{dslot(t.s), tslot(t.b)};
    if ((f_.size() + fchunk_ - 1) / fchunk_ > MAXP) fchunk_ = (f_.size() + MAXP - 1) / MAXP;
    if ((v_.size() + vchunk_ - 1) / vchunk_ > MAXP) vchunk_ = (v_.size() + MAXP - 1) / MAXP;
    size_t nf = (f_.size() + fchunk_ - 1) / fchunk_, nv = bucket ? 1 : (v_.size() + vchunk_ - 1) / vchunk_;
    const bool wsplit = !bucket && vsplit_ > 1 && nv > 0;      // the Straus part comes back as ONE folded piece
    size_t pieces = nf + (wsplit ? 1 : nv) + a_.size();
    if (pieces == 0) throw std::logic_error("empty msm");
    bool direct = pieces == 1 && a_.empty();
    std::vector<uint32_t> parts;
    for (size_t c = 0; c < nf; ++c) {
      uint32_t out = direct ? out_ : next_partial_++;
      size_t b = c * fchunk_, e = std::min(f_.size(), b + fchunk_);
      ph_.fjobs.push_back(Job{out, (uint32_t)ph_.fterms.size(), (uint32_t)(e - b)});
      ph_.fterms.insert(ph_.fterms.end(), f_.begin() + b, f_.begin() + e);
      parts.push_back(out);
    }
    if (bucket) {
      const uint32_t out = direct ? out_ : next_partial_++;
      const uint32_t kpad = (uint32_t)((v_.size() + 63) / 64 * 64);
      BJob bj{out, next_partial_, (uint32_t)ph_.bterms.size(), (uint32_t)v_.size(), kpad, ph_.b_dig_bytes};
      next_partial_ += bwin_;
      for (size_t i = 0; i < v_.size(); ++i) ph_.bpos.push_back(BTermPos{ph_.b_dig_bytes + (uint32_t)i, kpad});
      ph_.bterms.insert(ph_.bterms.end(), v_.begin(), v_.end());
      ph_.b_dig_bytes += bwin_ * kpad;
      ph_.b_kpad_max = std::max(ph_.b_kpad_max, kpad);
      ph_.bjobs.push_back(bj);
      parts.push_back(out);
    }
    if (wsplit) {
      std::vector<uint32_t> first(nv);
      for (size_t c = 0; c < nv; ++c) {
        first[c] = next_partial_;
        next_partial_ += vsplit_;
        size_t b = c * vchunk_, e = std::min(v_.size(), b + vchunk_);
        ph_.vjobs.push_back(Job{first[c], (uint32_t)ph_.vterms.size(), (uint32_t)(e - b)});
        ph_.vterms.insert(ph_.vterms.end(), v_.begin() + b, v_.begin() + e);
      }
      uint32_t sums = first[0];
      if (nv > 1) {
        sums = next_partial_;
        next_partial_ += vsplit_;
        for (uint32_t r = 0; r < vsplit_; ++r) {
          ph_.wcjobs.push_back(Job{sums + r, (uint32_t)ph_.wcterms.size(), (uint32_t)nv});
          for (size_t c = 0; c < nv; ++c) ph_.wcterms.push_back(Term{first[c] + r, 0});
        }
      }
      const uint32_t out = direct ? out_ : next_partial_++;
      ph_.wjobs.push_back(BJob{out, sums, 0, vsplit_, 0, 0});
      parts.push_back(out);
    }
    for (size_t c = 0; !bucket && !wsplit && c < nv; ++c) {
      uint32_t out = direct ? out_ : next_partial_++;
      size_t b = c * vchunk_, e = std::min(v_.size(), b + vchunk_);
      ph_.vjobs.push_back(Job{out, (uint32_t)ph_.vterms.size(), (uint32_t)(e - b)});
      ph_.vterms.insert(ph_.vterms.end(), v_.begin() + b, v_.begin() + e);
      parts.push_back(out);
    }
    if (!direct) {
      std::vector<Job>& cj = late ? ph_.cjobs2 : ph_.cjobs;
      std::vector<Term>& ct = late ? ph_.cterms2 : ph_.cterms;
      parts.insert(parts.end(), a_.begin(), a_.end());
      // the fine splits leave an MSM with up to ~250 partial sums and one lane to add them up: from COMBINE_TREE_MIN pieces on,
      // lanes of a first pass add groups of ~sqrt(pieces) and the MSM's own job adds the group sums (2 sqrt(n) additions deep
      // instead of n; the throughput plans never get here)
      if (!late && parts.size() >= COMBINE_TREE_MIN) {
        size_t g = 1;
        while (g * g < parts.size()) ++g;
        std::vector<uint32_t> sums;
        for (size_t b = 0; b < parts.size(); b += g) {
          const size_t e = std::min(parts.size(), b + g);
          const uint32_t out = next_partial_++;
          ph_.cjobs0.push_back(Job{out, (uint32_t)ph_.cterms0.size(), (uint32_t)(e - b)});
          for (size_t i = b; i < e; ++i) ph_.cterms0.push_back(Term{parts[i], 0});
          sums.push_back(out);
        }
        parts.swap(sums);
      }
      cj.push_back(Job{out_, (uint32_t)ct.size(), (uint32_t)parts.size()});
      for (uint32_t p : parts) ct.push_back(Term{p, 0});
    }
  }
  void normalize(uint32_t first, uint32_t count) {
    if (count) ph_.normalize.push_back({first, count});
  }

 private:
  uint32_t dslot(uint32_t sslot) {
    auto it = dmap_.find(sslot);
    if (it != dmap_.end()) return it->second;
    uint32_t d = ph_.n_dslots++;
    dmap_[sslot] = d;
    ph_.recode.push_back(Term{sslot, d});
    return d;
  }
  uint32_t tslot(uint32_t pslot) {
    auto it = tmap_.find(pslot);
    if (it != tmap_.end()) return it->second;
    uint32_t t = ph_.n_tslots++;
    tmap_[pslot] = t;
    ph_.tables.push_back(Term{pslot, t});
    return t;
  }
  Phase& ph_;
  uint32_t& next_partial_;
  uint32_t fchunk_, vchunk_, bucket_min_, bwin_, vsplit_;
  uint32_t out_ = 0;
  std::vector<Term> f_, v_;
  std::vector<uint32_t> a_;
  std::map<uint32_t, uint32_t> dmap_, tmap_;
};

// ---- slot maps (plain data: passed by value to the protocol kernels) --------------------------------
struct ProveLay {
  uint32_t m, n, N;
  // S arena
  uint32_t rho, a, b, tmp, r, s, sb, hs, dz, t, bp, zB, zs, za0, zbm, zr0, zsm, zt, zd;
  uint32_t svbp, svd, svrd, svdelta, svs1, svsx, svv1, svv2;
  uint32_t mea0, mer0, meb, mes, metau;
  uint32_t tsp, tsm;          // Toom-Cook (m = 2): halved scalar vectors (a0+a1+a2)/2, (a0-a1+a2)/2
  uint32_t x, y, z, hx, hy, zx, svx, mx;
  uint32_t zabar, zbbar, zrbar, zsbar, ztbar, svat, svbt, svrt, svst, meabar, merbar, mebbar, mesbar, metaubar;
  uint32_t nS, tmp_len;
  // P arena (J arena mirrors [0, nP))
  uint32_t deck, shuf, cA, cB, cb, hB, zcA0, zcBm, zcD, svcd, svcdelta, svcDelta, mecA0, mecB, meE;
  uint32_t tDp, tDm;          // Toom-Cook (m = 2): C'_1 + C'_2 and C'_2 - C'_1 (2n points each)
  uint32_t pk, kw;            // per-proof aggregate key (keyed batches) and its window bases 2^(5w) pk, w < KEY_WINDOWS
  uint32_t nP;
  uint32_t n_draws;
  uint32_t toom;              // 1: the multi-exponentiation diagonals use 4 evaluation points instead of 6 row products
};

struct VerifyLay {
  uint32_t m, n, N;
  // proof scalars
  uint32_t zabar, zbbar, zrbar, zsbar, ztbar, svat, svbt, svrt, svst, meabar, merbar, mebbar, mesbar, metaubar;
  // challenges, scratch, coefficients
  uint32_t x, y, z, hx, hy, zx, svx, mx, one, tmp, coef;
  uint32_t nS, tmp_len, n_coef;
  // points
  uint32_t deck, shuf, cA, cB, cb, hB, zcA0, zcBm, zcD, svcd, svcdelta, svcDelta, mecA0, mecB, meE;
  uint32_t pk;                // per-proof aggregate key (keyed batches)
  uint32_t nP;
  // result slots of the "== O" checks (J arena, >= nP) and their codes
  uint32_t chk_first, n_chk;
  // merged ("screening") verification: random weights r_k of the MSM checks, the merged scalar of every P slot and of
  // every fixed base, the J slot of the single merged check
  uint32_t mr, mvar, mfix, chk_merged;
};

// wire order of the proof: element = (is_point, slot); identical for prover (source) and verifier (destination)
struct ProofElem {
  uint32_t is_point;
  uint32_t slot;
  uint32_t offset;  // byte offset in the wire proof
};

template <class L>
static inline std::vector<ProofElem> proof_wire_map(const L& l, uint32_t point_bytes) {
  std::vector<ProofElem> v;
  uint32_t off = 0;
  auto P = [&](uint32_t s) { v.push_back(ProofElem{1, s, off}); off += point_bytes; };
  auto S = [&](uint32_t s) { v.push_back(ProofElem{0, s, off}); off += 32; };
  const uint32_t m = l.m, n = l.n;
  for (uint32_t k = 0; k < m; ++k) P(l.cA + k);
  for (uint32_t k = 0; k < m; ++k) P(l.cB + k);
  P(l.cb);
  for (uint32_t k = 0; k < m; ++k) P(l.hB + k);
  P(l.zcA0); P(l.zcBm);
  for (uint32_t k = 0; k < 2 * m + 1; ++k) P(l.zcD + k);
  for (uint32_t i = 0; i < n; ++i) S(l.zabar + i);
  for (uint32_t i = 0; i < n; ++i) S(l.zbbar + i);
  S(l.zrbar); S(l.zsbar); S(l.ztbar);
  P(l.svcd); P(l.svcdelta); P(l.svcDelta);
  for (uint32_t i = 0; i < n; ++i) S(l.svat + i);
  for (uint32_t i = 0; i < n; ++i) S(l.svbt + i);
  S(l.svrt); S(l.svst);
  P(l.mecA0);
  for (uint32_t k = 0; k < 2 * m; ++k) P(l.mecB + k);
  for (uint32_t k = 0; k < 4 * m; ++k) P(l.meE + k);
  for (uint32_t i = 0; i < n; ++i) S(l.meabar + i);
  S(l.merbar); S(l.mebbar); S(l.mesbar); S(l.metaubar);
  return v;
}

// point_bytes: 64 for the 256-bit curves, 96 for BLS12-377 (Geo<C>::PB)
static inline size_t proof_size_bytes(uint32_t m, uint32_t n, uint32_t point_bytes = 64) {
  return (size_t)(11 * m + 8) * point_bytes + (size_t)(5 * n + 9) * 32;
}

static inline ProveLay make_prove_lay(uint32_t m, uint32_t n) {
  ProveLay l{};
  l.m = m; l.n = n; l.N = m * n;
  const uint32_t N = l.N;
  uint32_t s = 0;
  auto A = [&](uint32_t cnt) { uint32_t r = s; s += cnt; return r; };
  l.rho = A(N); l.a = A(N); l.b = A(N);
  l.tmp_len = (m + 3) * n + 2 * m + N + 8;
  l.tmp = A(l.tmp_len);
  l.r = A(m); l.s = A(m); l.sb = A(1); l.hs = A(m);
  l.dz = A(N); l.t = A(m); l.bp = A(N); l.zB = A(N); l.zs = A(m);
  l.za0 = A(n); l.zbm = A(n); l.zr0 = A(1); l.zsm = A(1); l.zt = A(2 * m + 1); l.zd = A(2 * m + 1);
  l.svbp = A(n); l.svd = A(n); l.svrd = A(1); l.svdelta = A(n); l.svs1 = A(1); l.svsx = A(1); l.svv1 = A(n); l.svv2 = A(n);
  l.mea0 = A(n); l.mer0 = A(1); l.meb = A(2 * m); l.mes = A(2 * m); l.metau = A(2 * m);
  l.tsp = A(n); l.tsm = A(n);
  l.x = A(1); l.y = A(1); l.z = A(1); l.hx = A(1); l.hy = A(1); l.zx = A(1); l.svx = A(1); l.mx = A(1);
  l.zabar = A(n); l.zbbar = A(n); l.zrbar = A(1); l.zsbar = A(1); l.ztbar = A(1);
  l.svat = A(n); l.svbt = A(n); l.svrt = A(1); l.svst = A(1);
  l.meabar = A(n); l.merbar = A(1); l.mebbar = A(1); l.mesbar = A(1); l.metaubar = A(1);
  l.nS = s;
  uint32_t p = 0;
  auto Pn = [&](uint32_t cnt) { uint32_t r = p; p += cnt; return r; };
  l.deck = Pn(2 * N); l.shuf = Pn(2 * N); l.cA = Pn(m); l.cB = Pn(m); l.cb = Pn(1); l.hB = Pn(m);
  l.zcA0 = Pn(1); l.zcBm = Pn(1); l.zcD = Pn(2 * m + 1); l.svcd = Pn(1); l.svcdelta = Pn(1); l.svcDelta = Pn(1);
  l.mecA0 = Pn(1); l.mecB = Pn(2 * m); l.meE = Pn(4 * m);
  l.tDp = Pn(2 * n); l.tDm = Pn(2 * n);
  l.pk = Pn(1); l.kw = Pn(KEY_WINDOWS);
  l.nP = p;
  return l;
}

// prover randomness: the i-th `Fr::rand` draw of the prover stream goes to slot draws[i] (transcript v1
// order: r, s | sb | hadamard s_2..s_{m-1} | zero a0, bm, r0, sm, t | svp d, rd, delta_2..n-1, s1, sx |
// mexp a0, r0, b, s, tau)
static inline std::vector<uint32_t> prove_draw_slots(const ProveLay& l) {
  std::vector<uint32_t> d;
  const uint32_t m = l.m, n = l.n;
  auto R = [&](uint32_t first, uint32_t cnt) { for (uint32_t i = 0; i < cnt; ++i) d.push_back(first + i); };
  R(l.r, m); R(l.s, m);
  R(l.sb, 1);
  if (m > 2) R(l.hs + 1, m - 2);
  R(l.za0, n); R(l.zbm, n); R(l.zr0, 1); R(l.zsm, 1); R(l.zt, 2 * m + 1);
  R(l.svd, n); R(l.svrd, 1);
  if (n > 2) R(l.svdelta + 1, n - 2);
  R(l.svs1, 1); R(l.svsx, 1);
  R(l.mea0, n); R(l.mer0, 1); R(l.meb, 2 * m); R(l.mes, 2 * m); R(l.metau, 2 * m);
  return d;
}

// scalar-vector sums needed by the Karatsuba evaluation of the multi-exponentiation diagonals (kernel scal1):
// S[dst + t] = sum_{i < count} S[src[begin + i] + t], t < n
struct LinJob {
  uint32_t dst, begin, count;
};

// Toom-Cook evaluation of the multi-exponentiation diagonals for 3 <= m <= TOOM_MAX_M (see make_prove_plan): 2m evaluation
// points -- e = 0: X = 0, e = 1: X = infinity, then m - 1 pairs +-x: pair p = (e - 2) / 2 has the integer x = pair_x(p) and is either
// DIRECT (the polynomials at +-x) or REVERSED (the reversed polynomials at +-x, i.e. the originals at +-1/x scaled by a power of x --
// the same Horner recurrence on the coefficients in opposite order).  Pairs: (1), (2), (1/2), (3), (1/3), (4), (1/4), ...: with
// reciprocals the largest integer at m = 16 is 8 instead of 15 and the Horner multipliers x^2 stay <= 64.
struct ToomPlan {
  uint32_t E = 0;                 // number of evaluation points (2m), 0 = not used
  uint32_t sv_first = 0;          // S slots of the evaluated scalar vectors, e >= 2: sv_first + (e - 2) n
  uint32_t cv_first = 0;          // P slots of the evaluated ciphertext vectors, e >= 2: cv_first + (e - 2) 2n
  uint32_t pp_first = 0;          // P slots of the 2m products (2 components each): pp_first + 2 e + c
  uint32_t w_first = 0;           // S slots of the interpolation matrix W[k][e] (constants): w_first + k E + e
  // constants the engine computes in Fr (layout.hpp has no field arithmetic): index e (m + 1) + j = coefficient of a_j in the
  // scalar operand of point e, then W row-major
  uint32_t n_consts = 0, w_const_first = 0;
  static uint32_t pair_x(uint32_t p) { return p == 0 ? 1u : (p + 1) / 2 + 1; }        // 1, 2, 2, 3, 3, 4, 4, ...
  static bool pair_rev(uint32_t p) { return p != 0 && (p & 1u) == 0; }                // .., direct, reversed, direct, reversed
  int32_t x_of(uint32_t e) const {                                                    // signed integer of point e >= 2
    const int32_t x = (int32_t)pair_x((e - 2) / 2);
    return (e & 1u) ? -x : x;
  }
  bool rev_of(uint32_t e) const { return pair_rev((e - 2) / 2); }
};
static const uint32_t TOOM_MAX_M = 16;
// S[dst + t] = sum_i C_i S[src_i + t] with constant coefficients C_i = consts[lin_coef_i] (kernel k_lin_comb)
struct ProvePlan {
  ProveLay lay;
  std::vector<LinJob> lin;
  std::vector<uint32_t> lin_src;
  std::vector<uint32_t> lin_coef;     // per source: index into the constant table (Toom-Cook), empty = plain sums (Karatsuba)
  ToomPlan toom;
  Phase ph[6];          // A (cA), B (cB + multi-exp first message), C (product first messages), D (zero argument),
                        // [4] = A2: Toom-Cook / Karatsuba operand sums, run between A and B; [5] = B2: Toom-Cook interpolation
  uint32_t nJ;          // J arena slots (nP + partial sums)
  uint32_t jkey = NO_SLOT;   // keyed plans: J slots of tau_k pk, k < 2m (k_key_terms, before phase B)
  std::vector<uint32_t> draws;
  std::vector<ProofElem> wire;
};

// ---- Karatsuba evaluation of E(X) = (sum_j a_j X^j) (sum_t c_t X^t), c_t = row (m - t) of the shuffled deck, whose
// coefficients are the 2m diagonals E_k of the multi-exponentiation argument (Bayer-Groth section 4; the reference's
// dependency uses the schoolbook m(m+1) row products, examples/parameter_selection.rs:3-5).  A "product" is a row MSM
// <scalar vector, ciphertext vector>; Karatsuba only needs sums of scalar rows, sums of ciphertext rows and +-
// combinations of the products, so the E_k come out as the same group elements with 13 products instead of 20 at m = 4,
// 35 instead of 72 at m = 8.  Handles are sorted lists of original row indices (sums of disjoint row sets).
struct KLeaf {
  std::vector<uint32_t> a, c;
  std::map<uint32_t, int> contrib;   // coefficient k -> +-1
};
typedef std::vector<std::vector<uint32_t>> KPoly;
static inline std::vector<uint32_t> k_union(const std::vector<uint32_t>& x, const std::vector<uint32_t>& y) {
  std::vector<uint32_t> r(x);
  r.insert(r.end(), y.begin(), y.end());
  std::sort(r.begin(), r.end());
  return r;
}
static inline void k_append(std::vector<KLeaf>& dst, const std::vector<KLeaf>& src, uint32_t shift, int sign) {
  for (const KLeaf& s : src) {
    KLeaf t;
    t.a = s.a;
    t.c = s.c;
    for (auto& kv : s.contrib) t.contrib[kv.first + shift] = sign * kv.second;
    dst.push_back(t);
  }
}
static inline std::vector<KLeaf> k_mul(const KPoly& A, const KPoly& C) {
  std::vector<KLeaf> out;
  if (A.empty() || C.empty()) return out;
  if (A.size() == 1 || C.size() == 1) {
    for (uint32_t j = 0; j < A.size(); ++j)
      for (uint32_t t = 0; t < C.size(); ++t) {
        KLeaf lf;
        lf.a = A[j];
        lf.c = C[t];
        lf.contrib[j + t] = 1;
        out.push_back(lf);
      }
    return out;
  }
  const uint32_t h = (uint32_t)(std::max(A.size(), C.size()) + 1) / 2;
  const KPoly A0(A.begin(), A.begin() + std::min<size_t>(h, A.size())), A1(A.begin() + std::min<size_t>(h, A.size()), A.end());
  const KPoly C0(C.begin(), C.begin() + std::min<size_t>(h, C.size())), C1(C.begin() + std::min<size_t>(h, C.size()), C.end());
  if (A1.empty() || C1.empty()) {
    k_append(out, k_mul(A0, C0), 0, 1);
    if (!C1.empty()) k_append(out, k_mul(A0, C1), h, 1);
    if (!A1.empty()) k_append(out, k_mul(A1, C0), h, 1);
    return out;
  }
  KPoly As(A0), Cs(C0);
  for (size_t i = 0; i < A1.size(); ++i) As[i] = k_union(A0[i], A1[i]);
  for (size_t i = 0; i < C1.size(); ++i) Cs[i] = k_union(C0[i], C1[i]);
  const std::vector<KLeaf> P0 = k_mul(A0, C0), P2 = k_mul(A1, C1), P1 = k_mul(As, Cs);
  k_append(out, P0, 0, 1);
  k_append(out, P2, 2 * h, 1);
  k_append(out, P1, h, 1);
  k_append(out, P0, h, -1);
  k_append(out, P2, h, -1);
  return out;
}
// merge leaves with identical operands
static inline std::vector<KLeaf> k_merge(const std::vector<KLeaf>& in) {
  std::map<std::pair<std::vector<uint32_t>, std::vector<uint32_t>>, std::map<uint32_t, int>> acc;
  std::vector<std::pair<std::vector<uint32_t>, std::vector<uint32_t>>> order;
  for (const KLeaf& lf : in) {
    auto key = std::make_pair(lf.a, lf.c);
    if (!acc.count(key)) order.push_back(key);
    for (auto& kv : lf.contrib) acc[key][kv.first] += kv.second;
  }
  std::vector<KLeaf> out;
  for (auto& key : order) {
    KLeaf lf;
    lf.a = key.first;
    lf.c = key.second;
    for (auto& kv : acc[key])
      if (kv.second != 0) lf.contrib[kv.first] = kv.second;
    if (!lf.contrib.empty()) out.push_back(lf);
  }
  return out;
}

// keyed: the aggregate key is a per-proof point (P slot lay.pk) instead of the table's fixed base.  The re-encryption uses the key's
// own window tables (kernels_msm.hpp body_remask), and so do the 2m products tau_k pk of the multi-exponentiation diagonals since
// round 5 (k_key_terms writes them to the J slots jkey + k before phase B; until then each was a one-term Straus job with a
// 250-doubling chain of its own: + 7 % on k_var_msm of every keyed batch)
static inline ProvePlan make_prove_plan(uint32_t m, uint32_t n, uint32_t fchunk, uint32_t vchunk, uint32_t point_bytes = 64,
                                        bool keyed = false, uint32_t bucket_min = 0, uint32_t bwin = 0, bool toom_cook = true,
                                        uint32_t vsplit = 1, uint32_t bbits = 8) {
  ProvePlan pl;
  pl.lay = make_prove_lay(m, n);
  pl.lay.toom = m == 2 ? 1u : 0u;
  // 3 <= m <= 16: Toom-Cook with the 2m points 0, inf, +-1 .. +-(m-1): 2m row products instead of Karatsuba's 13 (m = 4) / 35
  // (m = 8) / 97 (m = 16).  (At m = 32 the evaluation of the ciphertext polynomial at +-31 would cost more than the products it saves.)  The ciphertext polynomial is evaluated with doublings and additions only (small integer points), the scalar
  // polynomial in Fr, and the coefficients E_k come back through the inverse Vandermonde matrix over Fr (phase B2: one
  // 2m-term MSM per diagonal over the 2m normalised products) -- the same group elements, so the proof bytes do not change.
  const bool toomk = toom_cook && m >= 3 && m <= TOOM_MAX_M;
  const bool karatsuba = m >= 3 && !toomk;
  if (toomk) {
    ToomPlan& T = pl.toom;
    T.E = 2 * m;
    T.sv_first = pl.lay.nS;
    pl.lay.nS += (T.E - 2) * n;
    T.w_first = pl.lay.nS;
    pl.lay.nS += T.E * T.E;
    T.cv_first = pl.lay.nP;
    pl.lay.nP += (T.E - 2) * 2 * n;
    T.pp_first = pl.lay.nP;
    pl.lay.nP += 2 * T.E;
    T.w_const_first = T.E * (m + 1);
    T.n_consts = T.w_const_first + T.E * T.E;
    for (uint32_t e = 2; e < T.E; ++e) {          // A(x_e) = sum_j x_e^j a_j
      pl.lin.push_back(LinJob{T.sv_first + (e - 2) * n, (uint32_t)pl.lin_src.size(), m + 1});
      for (uint32_t j = 0; j <= m; ++j) {
        pl.lin_src.push_back(j == 0 ? pl.lay.mea0 : pl.lay.b + (j - 1) * n);
        pl.lin_coef.push_back(e * (m + 1) + j);
      }
    }
  }
  // Karatsuba plan: leaves, operand vectors (new S / P slots appended to the layout)
  std::vector<KLeaf> leaves;
  std::map<std::vector<uint32_t>, uint32_t> svec, cvec;     // handle -> first S slot / first P slot
  uint32_t kP0 = pl.lay.nP;
  if (karatsuba) {
    KPoly A, Cc;
    for (uint32_t j = 0; j <= m; ++j) A.push_back({j});
    for (uint32_t t = 0; t < m; ++t) Cc.push_back({m - t - 1});    // X^t <-> row m - t (0-based m - t - 1)
    leaves = k_merge(k_mul(A, Cc));
    for (const KLeaf& lf : leaves) {
      for (auto& kv : lf.contrib)
        if (kv.second > 8 || kv.second < -8) throw std::logic_error("Karatsuba: coefficient out of range");
      if (!svec.count(lf.a)) {
        if (lf.a.size() == 1) {
          svec[lf.a] = lf.a[0] == 0 ? pl.lay.mea0 : pl.lay.b + (lf.a[0] - 1) * n;
        } else {
          svec[lf.a] = pl.lay.nS;
          pl.lin.push_back(LinJob{pl.lay.nS, (uint32_t)pl.lin_src.size(), (uint32_t)lf.a.size()});
          for (uint32_t j : lf.a) pl.lin_src.push_back(j == 0 ? pl.lay.mea0 : pl.lay.b + (j - 1) * n);
          pl.lay.nS += n;
        }
      }
      if (!cvec.count(lf.c)) {
        if (lf.c.size() == 1) {
          cvec[lf.c] = pl.lay.shuf + 2 * lf.c[0] * n;
        } else {
          cvec[lf.c] = pl.lay.nP;
          pl.lay.nP += 2 * n;
        }
      }
    }
  }
  const ProveLay& l = pl.lay;
  FixedBases fb{n};
  uint32_t next_partial = l.nP;
  if (keyed) {
    pl.jkey = next_partial;
    next_partial += 2 * m;
  }
  auto commit = [&](PhaseBuilder& B, uint32_t out, uint32_t vec, uint32_t len, uint32_t rslot) {
    B.begin(out);
    for (uint32_t j = 0; j < len; ++j) B.fixed(vec + j, fb.ck(j));
    B.fixed(rslot, fb.H());
    B.end();
  };
  {  // phase A: c_A (the re-encryption itself is the dedicated remask kernel)
    PhaseBuilder B(pl.ph[0], next_partial, fchunk, vchunk, bucket_min, bwin, vsplit, bbits);
    for (uint32_t k = 0; k < m; ++k) commit(B, l.cA + k, l.a + k * n, n, l.r + k);
    B.normalize(l.shuf, 2 * l.N);
    B.normalize(l.cA, m);
  }
  if (pl.lay.toom) {  // phase A2 (m = 2): D+ = C'_1 + C'_2, D- = C'_2 - C'_1 (affine + affine, then normalised)
    PhaseBuilder B(pl.ph[4], next_partial, fchunk, vchunk, bucket_min, bwin, vsplit, bbits);
    for (uint32_t t = 0; t < n; ++t)
      for (uint32_t c = 0; c < 2; ++c) {
        B.begin(l.tDp + 2 * t + c);
        B.addend(l.shuf + 2 * t + c);
        B.addend(l.shuf + 2 * (n + t) + c);
        B.end();
        B.begin(l.tDm + 2 * t + c);
        B.addend(l.shuf + 2 * (n + t) + c);
        B.addend(l.shuf + 2 * t + c, true);
        B.end();
      }
    B.normalize(l.tDp, 4 * n);
  }
  if (karatsuba && l.nP > kP0) {  // phase A2 (m >= 3): sums of ciphertext rows used as Karatsuba operands
    PhaseBuilder B(pl.ph[4], next_partial, fchunk, vchunk, bucket_min, bwin, vsplit, bbits);
    for (auto& kv : cvec) {
      if (kv.first.size() == 1) continue;
      for (uint32_t t = 0; t < n; ++t)
        for (uint32_t c = 0; c < 2; ++c) {
          B.begin(kv.second + 2 * t + c);
          for (uint32_t r : kv.first) B.addend(l.shuf + 2 * (r * n + t) + c);
          B.end();
        }
    }
    B.normalize(kP0, l.nP - kP0);
  }
  {  // phase B: c_B, multi-exponentiation first message
    PhaseBuilder B(pl.ph[1], next_partial, fchunk, vchunk, bucket_min, bwin, vsplit, bbits);
    for (uint32_t k = 0; k < m; ++k) commit(B, l.cB + k, l.b + k * n, n, l.s + k);
    commit(B, l.mecA0, l.mea0, n, l.mer0);
    for (uint32_t k = 0; k < 2 * m; ++k) commit(B, l.mecB + k, l.meb + k, 1, l.mes + k);
    const bool toom = pl.lay.toom != 0;
    if (toom) {
      // m = 2.  E(X) = (a0 + a1 X + a2 X^2)(C'_2 + C'_1 X): evaluate at 0, inf, 1, -1 with the scalars of the last two
      // already halved (kernel scal1), then E_0 = V0, E_3 = Vinf, E_1 = V1h - Vm1h - Vinf, E_2 = V1h + Vm1h - V0:
      // 4 row products of n terms per component instead of 6, same group elements.
      for (uint32_t c = 0; c < 2; ++c) {
        const uint32_t v0 = B.new_partial(), vinf = B.new_partial(), v1 = B.new_partial(), vm1 = B.new_partial();
        B.begin(v0);
        for (uint32_t t = 0; t < n; ++t) B.var(l.mea0 + t, l.shuf + 2 * (n + t) + c);          // a0 . C'_2
        B.end();
        B.begin(vinf);
        for (uint32_t t = 0; t < n; ++t) B.var(l.b + n + t, l.shuf + 2 * t + c);               // a2 . C'_1
        B.end();
        B.begin(v1);
        for (uint32_t t = 0; t < n; ++t) B.var(l.tsp + t, l.tDp + 2 * t + c);                  // (a0+a1+a2)/2 . (C'_1+C'_2)
        B.end();
        B.begin(vm1);
        for (uint32_t t = 0; t < n; ++t) B.var(l.tsm + t, l.tDm + 2 * t + c);                  // (a0-a1+a2)/2 . (C'_2-C'_1)
        B.end();
        for (uint32_t k = 0; k < 4; ++k) {
          B.begin(l.meE + 2 * k + c);
          if (c == 0) {
            B.fixed(l.metau + k, fb.G());
          } else {
            B.fixed(l.meb + k, fb.gen());
            if (keyed) B.addend_j(pl.jkey + k); else B.fixed(l.metau + k, fb.pk());
          }
          if (k == 0) B.addend_j(v0);
          if (k == 3) B.addend_j(vinf);
          if (k == 1) { B.addend_j(v1); B.addend_j(vm1, true); B.addend_j(vinf, true); }
          if (k == 2) { B.addend_j(v1); B.addend_j(vm1); B.addend_j(v0, true); }
          B.end(true);
        }
      }
    } else if (toomk) {
      // the 2m products <A(x_e), C(x_e)> (two components each), normalised at the end of this phase; the diagonals: phase B2
      const ToomPlan& T = pl.toom;
      for (uint32_t e = 0; e < T.E; ++e)
        for (uint32_t c = 0; c < 2; ++c) {
          const uint32_t sv = e == 0 ? l.mea0 : (e == 1 ? l.b + (m - 1) * n : T.sv_first + (e - 2) * n);
          const uint32_t cv = e == 0 ? l.shuf + 2 * (m - 1) * n : (e == 1 ? l.shuf : T.cv_first + (e - 2) * 2 * n);
          B.begin(T.pp_first + 2 * e + c);
          for (uint32_t t = 0; t < n; ++t) B.var(sv + t, cv + 2 * t + c);
          B.end();
        }
      B.normalize(T.pp_first, 2 * T.E);
    } else if (karatsuba) {
      for (uint32_t c = 0; c < 2; ++c) {
        std::vector<uint32_t> part(leaves.size());
        for (size_t i = 0; i < leaves.size(); ++i) {
          part[i] = B.new_partial();
          const uint32_t sv = svec[leaves[i].a], cv = cvec[leaves[i].c];
          B.begin(part[i]);
          for (uint32_t t = 0; t < n; ++t) B.var(sv + t, cv + 2 * t + c);
          B.end();
        }
        for (uint32_t k = 0; k < 2 * m; ++k) {
          B.begin(l.meE + 2 * k + c);
          if (c == 0) {
            B.fixed(l.metau + k, fb.G());
          } else {
            B.fixed(l.meb + k, fb.gen());
            if (keyed) B.addend_j(pl.jkey + k); else B.fixed(l.metau + k, fb.pk());
          }
          for (size_t i = 0; i < leaves.size(); ++i) {
            auto it = leaves[i].contrib.find(k);
            if (it == leaves[i].contrib.end()) continue;
            // merged leaves can carry a small integer coefficient: add the product that many times
            for (int rep = 0; rep < (it->second < 0 ? -it->second : it->second); ++rep) B.addend_j(part[i], it->second < 0);
          }
          B.end(true);
        }
      }
    } else {
    for (uint32_t k = 0; k < 2 * m; ++k) {
      for (uint32_t c = 0; c < 2; ++c) {
        B.begin(l.meE + 2 * k + c);
        if (c == 0) {
          B.fixed(l.metau + k, fb.G());
        } else {
          B.fixed(l.meb + k, fb.gen());
          if (keyed) B.addend_j(pl.jkey + k); else B.fixed(l.metau + k, fb.pk());
        }
        for (uint32_t i = 1; i <= m; ++i) {
          int64_t j = (int64_t)k - (int64_t)m + (int64_t)i;
          if (j < 0 || j > (int64_t)m) continue;
          uint32_t vec = j == 0 ? l.mea0 : l.b + (uint32_t)(j - 1) * n;   // a_j: a_0 random, a_j = row j of b
          for (uint32_t t = 0; t < n; ++t) B.var(vec + t, l.shuf + 2 * ((i - 1) * n + t) + c);
        }
        B.end();
      }
    }
    }
    B.normalize(l.cB, m);
    B.normalize(l.mecA0, toomk ? 1 + 2 * m : 1 + 2 * m + 4 * m);
  }
  if (toomk) {  // phase B2: E_k = E(b_k gen; tau_k) + sum_e W[k][e] P_e  (E_0 = P_0 and E_{2m-1} = P_inf exactly)
    const ToomPlan& T = pl.toom;
    PhaseBuilder B(pl.ph[5], next_partial, fchunk, vchunk, bucket_min, bwin, vsplit, bbits);
    for (uint32_t k = 0; k < 2 * m; ++k)
      for (uint32_t c = 0; c < 2; ++c) {
        B.begin(l.meE + 2 * k + c);
        if (c == 0) {
          B.fixed(l.metau + k, fb.G());
        } else {
          B.fixed(l.meb + k, fb.gen());
          if (keyed) B.addend_j(pl.jkey + k); else B.fixed(l.metau + k, fb.pk());
        }
        if (k == 0) {
          B.addend(T.pp_first + c);
        } else if (k == 2 * m - 1) {
          B.addend(T.pp_first + 2 + c);
        } else {
          for (uint32_t e = 0; e < T.E; ++e) B.var(T.w_first + k * T.E + e, T.pp_first + 2 * e + c);
        }
        B.end();
      }
    B.normalize(l.meE, 4 * m);
  }
  {  // phase C: product-argument first messages that do not depend on later challenges
    PhaseBuilder B(pl.ph[2], next_partial, fchunk, vchunk, bucket_min, bwin, vsplit, bbits);
    commit(B, l.cb, l.bp + (m - 1) * n, n, l.sb);
    commit(B, l.hB + 0, l.dz, n, l.t);                       // = c_A[0] of the product statement (c_D0 + c_{-z})
    for (uint32_t i = 1; i + 1 < m; ++i) commit(B, l.hB + i, l.bp + i * n, n, l.hs + i);
    commit(B, l.svcd, l.svd, n, l.svrd);
    commit(B, l.svcdelta, l.svv1, n - 1, l.svs1);
    commit(B, l.svcDelta, l.svv2, n - 1, l.svsx);
    B.normalize(l.cb, 1);
    B.normalize(l.hB, m - 1);
    B.normalize(l.svcd, 3);
  }
  {  // phase D: zero-argument first message
    PhaseBuilder B(pl.ph[3], next_partial, fchunk, vchunk, bucket_min, bwin, vsplit, bbits);
    commit(B, l.zcA0, l.za0, n, l.zr0);
    commit(B, l.zcBm, l.zbm, n, l.zsm);
    for (uint32_t k = 0; k < 2 * m + 1; ++k) commit(B, l.zcD + k, l.zd + k, 1, l.zt + k);
    B.normalize(l.zcA0, 2 + 2 * m + 1);
  }
  pl.nJ = next_partial;
  pl.draws = prove_draw_slots(l);
  pl.lay.n_draws = (uint32_t)pl.draws.size();
  pl.wire = proof_wire_map(l, point_bytes);
  return pl;
}

// ---- verifier ------------------------------------------------------------------------------------
// Every group equation is evaluated as one MSM that must equal the identity.  Check order (= the order in
// which the oracle's verifier would fail) and the code reported:
enum VCheck {
  VC_HAD_B1 = 0,      // 1  c_B1 == c_A1 of the product statement          (MSM)
  VC_HAD_BM,          // 1  c_Bm == c_b                                    (direct)
  VC_ZERO_DM1,        // 2  c_D[m+1] == O                                  (direct)
  VC_ZERO_A,          // 2  sum x^i c_Ai == com(abar; rbar)                (MSM)
  VC_ZERO_B,          // 2  sum x^(m-j) c_Bj == com(bbar; sbar)            (MSM)
  VC_ZERO_D,          // 2  sum x^k c_Dk == com(abar*bbar; tbar)           (MSM)
  VC_SVP_A,           // 3  x c_a + c_d == com(at; rt)                     (MSM)
  VC_SVP_D,           // 3  x c_Delta + c_delta == com(..; st)             (MSM)
  VC_SVP_SCALARS,     // 3  bt_1 == at_1, bt_n == x b                      (direct)
  VC_ME_BM,           // 4  c_B[m] == O                                    (direct)
  VC_ME_EM0,          // 4  E[m].c0 == Cx.c0                               (MSM)
  VC_ME_EM1,          // 4  E[m].c1 == Cx.c1                               (MSM)
  VC_ME_A,            // 4  c_A0 + sum x^j c_Aj == com(abar; rbar)         (MSM)
  VC_ME_B,            // 4  sum x^k c_Bk == com(bbar; sbar)                (MSM)
  VC_ME_E0,           // 4  ciphertext equation, component 0               (MSM)
  VC_ME_E1,           // 4  ciphertext equation, component 1               (MSM)
  VC_COUNT
};
MP_HD int vcheck_code(int c) {
  if (c <= VC_HAD_BM) return 1;
  if (c <= VC_ZERO_D) return 2;
  if (c <= VC_SVP_SCALARS) return 3;
  return 4;
}
MP_HD bool vcheck_is_msm(int c) {
  return !(c == VC_HAD_BM || c == VC_ZERO_DM1 || c == VC_SVP_SCALARS || c == VC_ME_BM);
}

static inline VerifyLay make_verify_lay(uint32_t m, uint32_t n) {
  VerifyLay l{};
  l.m = m; l.n = n; l.N = m * n;
  const uint32_t N = l.N;
  uint32_t s = 0;
  auto A = [&](uint32_t cnt) { uint32_t r = s; s += cnt; return r; };
  l.zabar = A(n); l.zbbar = A(n); l.zrbar = A(1); l.zsbar = A(1); l.ztbar = A(1);
  l.svat = A(n); l.svbt = A(n); l.svrt = A(1); l.svst = A(1);
  l.meabar = A(n); l.merbar = A(1); l.mebbar = A(1); l.mesbar = A(1); l.metaubar = A(1);
  l.x = A(1); l.y = A(1); l.z = A(1); l.hx = A(1); l.hy = A(1); l.zx = A(1); l.svx = A(1); l.mx = A(1); l.one = A(1);
  l.tmp_len = N + 2 * n + 4 * m + 8;
  l.tmp = A(l.tmp_len);
  // coefficient block: sized generously; exact use is fixed by make_verify_plan / body_verify_scalars
  l.n_coef = 2 * N + 6 * n + 16 * m + 32;
  l.coef = A(l.n_coef);
  l.mr = A(VC_COUNT);
  l.mvar = A(4 * N + 11 * m + 9);      // one per P slot (= nP below)
  l.mfix = A(n + 5);                   // one per fixed base (FixedBases::count())
  l.nS = s;
  uint32_t p = 0;
  auto Pn = [&](uint32_t cnt) { uint32_t r = p; p += cnt; return r; };
  l.deck = Pn(2 * N); l.shuf = Pn(2 * N); l.cA = Pn(m); l.cB = Pn(m); l.cb = Pn(1); l.hB = Pn(m);
  l.zcA0 = Pn(1); l.zcBm = Pn(1); l.zcD = Pn(2 * m + 1); l.svcd = Pn(1); l.svcdelta = Pn(1); l.svcDelta = Pn(1);
  l.mecA0 = Pn(1); l.mecB = Pn(2 * m); l.meE = Pn(4 * m);
  l.pk = Pn(1);
  l.nP = p;
  if (l.nP != 4 * N + 11 * m + 9) throw std::logic_error("verify layout: P slot count");
  l.chk_first = l.nP;
  l.n_chk = VC_COUNT + 1;
  l.chk_merged = l.chk_first + VC_COUNT;
  return l;
}

// Coefficient slots.  body_verify_scalars (kernels_proto.hpp) fills them; make_verify_plan consumes them in
// the SAME order through this little allocator, so both sides are generated from one description:
struct VCoef {
  uint32_t base, next;
  explicit VCoef(uint32_t b) : base(b), next(b) {}
  uint32_t take(uint32_t cnt = 1) {
    uint32_t r = next;
    next += cnt;
    return r;
  }
};
// Layout of the coefficient block (c = coef base); all values are the scalars of the "== O" MSMs.
struct VCoefMap {
  uint32_t had_y, had_m1, had_mz;                        // VC_HAD_B1: y*cA0 + 1*cB0 - 1*hB0 - z*gsum
  uint32_t za_cA, za_cB, za_gsum, za_ck, za_H;           // VC_ZERO_A: 1*zcA0 + sum_{i=1}^{m-1} zx^i (y cA_i + cB_i) + gsum*(..) - abar.ck - rbar H
  uint32_t zb_hB, zb_ck, zb_H;                           // VC_ZERO_B: coefficients of hB[0..m-1], 1*zcBm, -bbar.ck, -sbar H
  uint32_t zd_cD, zd_ck0, zd_H;                          // VC_ZERO_D: zx^k (k=0..2m), -(abar*bbar), -tbar
  uint32_t sa_x, sa_ck, sa_H;                            // VC_SVP_A: x*cb + 1*cd - at.ck - rt H
  uint32_t sd_x, sd_ck, sd_H;                            // VC_SVP_D: x*cDelta + 1*cdelta - v.ck - st H
  uint32_t em_x;                                         // VC_ME_EM*: x^{i+1} (N values); -1*E[m]
  uint32_t ma_x, ma_ck, ma_H;                            // VC_ME_A: 1*mecA0 + mx^j cB_{j-1} - abar.ck - rbar H
  uint32_t mb_x, mb_ck0, mb_H;                           // VC_ME_B: mx^k mecB_k - bbar ck0 - sbar H
  uint32_t me_x, me_c, me_tauG, me_bgen, me_taupk;       // VC_ME_E*: mx^k E_k ; -(mx^{m-i} abar_l) (N) ; -taubar G ; -bbar gen ; -taubar pk
  uint32_t minus_one;
  uint32_t end;
};
static inline VCoefMap make_vcoef_map(const VerifyLay& l) {
  VCoefMap c{};
  const uint32_t m = l.m, n = l.n, N = l.N;
  VCoef a(l.coef);
  c.minus_one = a.take();
  c.had_y = a.take(); c.had_mz = a.take();
  c.za_cA = a.take(m); c.za_cB = a.take(m); c.za_gsum = a.take(); c.za_ck = a.take(n); c.za_H = a.take();
  c.zb_hB = a.take(m); c.zb_ck = a.take(n); c.zb_H = a.take();
  c.zd_cD = a.take(2 * m + 1); c.zd_ck0 = a.take(); c.zd_H = a.take();
  c.sa_x = a.take(); c.sa_ck = a.take(n); c.sa_H = a.take();
  c.sd_x = a.take(); c.sd_ck = a.take(n); c.sd_H = a.take();
  c.em_x = a.take(N);
  c.ma_x = a.take(m + 1); c.ma_ck = a.take(n); c.ma_H = a.take();
  c.mb_x = a.take(2 * m); c.mb_ck0 = a.take(); c.mb_H = a.take();
  c.me_x = a.take(2 * m); c.me_c = a.take(N); c.me_tauG = a.take(); c.me_bgen = a.take(); c.me_taupk = a.take();
  c.end = a.next;
  if (c.end - l.coef > l.n_coef) throw std::logic_error("coefficient block too small");
  return c;
}

// merged scalar dst = sum over pairs S[r] * S[coef]   (kernel k_verify_merge)
struct MergeJob {
  uint32_t dst, begin, count;
};
struct MergePair {
  uint32_t r, coef;
};
struct VerifyPlan {
  VerifyLay lay;
  VCoefMap cm;
  Phase ph;             // one MSM per equation: names the first failing check
  Phase mph;            // all equations merged with random weights into ONE MSM: accept / "look closer"
  std::vector<MergeJob> mjobs;
  std::vector<MergePair> mpairs;
  uint32_t nJ;
  std::vector<ProofElem> wire;
};

// The verifier's group equations, each "sum of scalar * point == O" (one description, two consumers: the per-equation plan
// and the merged plan).  Sink: begin(check id) / var(coef slot, P slot) / fixed(coef slot, base) / end().
template <class Sink>
static inline void describe_verify(const VerifyLay& l, const VCoefMap& c, Sink& B, bool keyed) {
  const uint32_t m = l.m, n = l.n, N = l.N;
  FixedBases fb{n};
  // VC_HAD_B1
  B.begin(VC_HAD_B1);
  B.var(c.had_y, l.cA + 0); B.var(l.one, l.cB + 0); B.var(c.minus_one, l.hB + 0); B.fixed(c.had_mz, fb.gsum());
  B.end();
  // VC_ZERO_A
  B.begin(VC_ZERO_A);
  B.var(l.one, l.zcA0);
  for (uint32_t i = 1; i < m; ++i) { B.var(c.za_cA + i, l.cA + i); B.var(c.za_cB + i, l.cB + i); }
  B.fixed(c.za_gsum, fb.gsum());
  for (uint32_t j = 0; j < n; ++j) B.fixed(c.za_ck + j, fb.ck(j));
  B.fixed(c.za_H, fb.H());
  B.end();
  // VC_ZERO_B
  B.begin(VC_ZERO_B);
  for (uint32_t j = 0; j < m; ++j) B.var(c.zb_hB + j, l.hB + j);
  B.var(l.one, l.zcBm);
  for (uint32_t j = 0; j < n; ++j) B.fixed(c.zb_ck + j, fb.ck(j));
  B.fixed(c.zb_H, fb.H());
  B.end();
  // VC_ZERO_D
  B.begin(VC_ZERO_D);
  for (uint32_t k = 0; k < 2 * m + 1; ++k) B.var(c.zd_cD + k, l.zcD + k);
  B.fixed(c.zd_ck0, fb.ck(0)); B.fixed(c.zd_H, fb.H());
  B.end();
  // VC_SVP_A
  B.begin(VC_SVP_A);
  B.var(c.sa_x, l.cb); B.var(l.one, l.svcd);
  for (uint32_t j = 0; j < n; ++j) B.fixed(c.sa_ck + j, fb.ck(j));
  B.fixed(c.sa_H, fb.H());
  B.end();
  // VC_SVP_D
  B.begin(VC_SVP_D);
  B.var(c.sd_x, l.svcDelta); B.var(l.one, l.svcdelta);
  for (uint32_t j = 0; j + 1 < n; ++j) B.fixed(c.sd_ck + j, fb.ck(j));
  B.fixed(c.sd_H, fb.H());
  B.end();
  // VC_ME_EM0/1 : sum x^{i+1} deck_i - E[m] == O
  for (uint32_t comp = 0; comp < 2; ++comp) {
    B.begin(VC_ME_EM0 + comp);
    for (uint32_t i = 0; i < N; ++i) B.var(c.em_x + i, l.deck + 2 * i + comp);
    B.var(c.minus_one, l.meE + 2 * m + comp);
    B.end();
  }
  // VC_ME_A
  B.begin(VC_ME_A);
  B.var(l.one, l.mecA0);
  for (uint32_t j = 1; j <= m; ++j) B.var(c.ma_x + j, l.cB + (j - 1));
  for (uint32_t j = 0; j < n; ++j) B.fixed(c.ma_ck + j, fb.ck(j));
  B.fixed(c.ma_H, fb.H());
  B.end();
  // VC_ME_B
  B.begin(VC_ME_B);
  for (uint32_t k = 0; k < 2 * m; ++k) B.var(c.mb_x + k, l.mecB + k);
  B.fixed(c.mb_ck0, fb.ck(0)); B.fixed(c.mb_H, fb.H());
  B.end();
  // VC_ME_E0/1
  for (uint32_t comp = 0; comp < 2; ++comp) {
    B.begin(VC_ME_E0 + comp);
    for (uint32_t k = 0; k < 2 * m; ++k) B.var(c.me_x + k, l.meE + 2 * k + comp);
    for (uint32_t i = 0; i < N; ++i) B.var(c.me_c + i, l.shuf + 2 * i + comp);
    if (comp == 0) {
      B.fixed(c.me_tauG, fb.G());
    } else {
      B.fixed(c.me_bgen, fb.gen());
      if (keyed) B.var(c.me_taupk, l.pk); else B.fixed(c.me_taupk, fb.pk());
    }
    B.end();
  }
}

// Sink 1: one MSM per equation, result in the equation's check slot
struct PerCheckSink {
  PhaseBuilder& B;
  uint32_t chk_first;
  void begin(int id) { B.begin(chk_first + (uint32_t)id); }
  void var(uint32_t coef, uint32_t pslot) { B.var(coef, pslot); }
  void fixed(uint32_t coef, uint32_t base) { B.fixed(coef, base); }
  void end() { B.end(); }
};
// Sink 2: records (check, coefficient) per point / base for the merged equation sum_k r_k * (equation k) == O
struct MergeSink {
  uint32_t mr;
  int cur = 0;
  std::map<uint32_t, std::vector<MergePair>> by_p, by_base;
  void begin(int id) { cur = id; }
  void var(uint32_t coef, uint32_t pslot) { by_p[pslot].push_back(MergePair{mr + (uint32_t)cur, coef}); }
  void fixed(uint32_t coef, uint32_t base) { by_base[base].push_back(MergePair{mr + (uint32_t)cur, coef}); }
  void end() {}
};

static inline VerifyPlan make_verify_plan(uint32_t m, uint32_t n, uint32_t fchunk, uint32_t vchunk, uint32_t point_bytes = 64,
                                          bool keyed = false, uint32_t bucket_min = 0, uint32_t bwin = 0, uint32_t vsplit = 1,
                                          uint32_t bbits = 8) {
  VerifyPlan pl;
  pl.lay = make_verify_lay(m, n);
  const VerifyLay& l = pl.lay;
  pl.cm = make_vcoef_map(l);
  const VCoefMap& c = pl.cm;
  {
    uint32_t next_partial = l.chk_first + l.n_chk;
    PhaseBuilder B(pl.ph, next_partial, fchunk, vchunk, bucket_min, bwin, vsplit, bbits);
    PerCheckSink sink{B, l.chk_first};
    describe_verify(l, c, sink, keyed);
    pl.nJ = next_partial;
  }
  {
    // merged plan: every distinct point / base once, with the scalar sum_k r_k * coef_k (k_verify_merge).  Fewer
    // fixed-base terms (n + 5 instead of one per equation and base) and fewer doubling chains (the variable-base terms
    // fill whole sub-jobs) -- and a single result to test.
    MergeSink ms{l.mr};
    describe_verify(l, c, ms, keyed);
    uint32_t next_partial = l.chk_first + l.n_chk;
    PhaseBuilder B(pl.mph, next_partial, fchunk, vchunk, bucket_min, bwin, vsplit, bbits);
    B.begin(l.chk_merged);
    auto job = [&](uint32_t dst, const std::vector<MergePair>& v) {
      pl.mjobs.push_back(MergeJob{dst, (uint32_t)pl.mpairs.size(), (uint32_t)v.size()});
      pl.mpairs.insert(pl.mpairs.end(), v.begin(), v.end());
    };
    for (auto& kv : ms.by_p) {
      job(l.mvar + kv.first, kv.second);
      B.var(l.mvar + kv.first, kv.first);
    }
    for (auto& kv : ms.by_base) {
      job(l.mfix + kv.first, kv.second);
      B.fixed(l.mfix + kv.first, kv.first);
    }
    B.end();
    pl.nJ = std::max(pl.nJ, next_partial);
  }
  pl.wire = proof_wire_map(l, point_bytes);
  return pl;
}

}  // namespace mp
