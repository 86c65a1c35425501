// TEST INFRASTRUCTURE ONLY (oracle).  CPU restatement of the Bayer-Groth shuffle argument and the
// ElGamal re-encryption behind
//   DLCards::shuffle_and_remask  [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:380-418]
//   DLCards::verify_shuffle      [REF .../discrete_log_cards/mod.rs:420-443]
//   Remask / Mask                [REF .../remasking.rs:9-22, .../masking.rs:10-20]
// The argument itself (`proof_essentials::zkp::arguments::shuffle`, imported at mod.rs:20-24) is a git
// dependency that is not under /root/reference: this follows Bayer-Groth (EUROCRYPT 2012) sections 4-5.3
// exactly as frozen in oracle/py/mp_oracle.py ("mpshuffle transcript v1") and must stay byte-identical
// to it (tests/test_oracle_golden.py).  MSMs use the arkworks bucket method, single scalar-muls the
// arkworks double-and-add, single-threaded -- the algorithms of the reference's CPU path.
// PARITY UNPINNED (see oracle/README.md).
#pragma once
#include <cstring>
#include <stdexcept>
#include <vector>

#include "curve.hpp"
#include "hash.hpp"

namespace mpo {

enum Check { OK = 0, HADAMARD = 1, ZERO_ARG = 2, SVP = 3, MULTIEXP = 4 };

template <class Cv>
struct Shuffle {
  typedef typename Cv::Fq Fq;
  typedef typename Cv::Fr Fr;
  typedef Affine<Cv> Pt;
  typedef Jac<Cv> J;
  typedef std::vector<Fr> FrVec;
  typedef std::vector<Pt> PtVec;
  struct Ct {
    Pt c0, c1;
    bool operator==(const Ct& o) const { return c0 == o.c0 && c1 == o.c1; }
    bool operator!=(const Ct& o) const { return !(*this == o); }
  };
  typedef std::vector<Ct> Deck;

  struct Params {
    uint32_t m, n;
    Pt G;
    PtVec ck;  // n generators
    Pt H, gen;
    Pt gsum;   // sum of ck (cached; derived)
  };

  // ---------------------------------------------------------------- encodings
  static const size_t FQB = Fq::BYTES;   // bytes of a base-field coordinate (32; 48 on BLS12-377)
  static const size_t PW = 2 * FQB;      // wire size of a point
  static void pt_tobytes(const Pt& P, std::vector<uint8_t>& out) {  // ark ToBytes: x || y || inf
    uint8_t b[2 * Fq::BYTES + 1];
    if (P.inf) {
      memset(b, 0, PW + 1);
      b[FQB] = 1;
      b[PW] = 1;
    } else {
      P.x.to_bytes(b);
      P.y.to_bytes(b + FQB);
      b[PW] = 0;
    }
    out.insert(out.end(), b, b + PW + 1);
  }
  static void pts_tobytes(const PtVec& v, std::vector<uint8_t>& out) {
    for (auto& P : v) pt_tobytes(P, out);
  }
  static void pt_wire(const Pt& P, uint8_t* out) {
    if (P.inf) {
      memset(out, 0, PW);  // infinity = (0, 0): on none of the curves (b != 0)
    } else {
      P.x.to_bytes(out);
      P.y.to_bytes(out + FQB);
    }
  }
  static bool pt_from_wire(const uint8_t* in, Pt& P) {
    bool allzero = true;
    for (size_t i = 0; i < PW; ++i) allzero &= in[i] == 0;
    if (allzero) {
      P = Pt::infinity();
      return true;
    }
    P.inf = false;
    return Fq::from_bytes(in, P.x) && Fq::from_bytes(in + FQB, P.y);
  }

  // ---------------------------------------------------------------- group helpers
  static Pt msm(const FrVec& s, const PtVec& p) { return msm_pippenger<Cv>(s.data(), p.data(), s.size()).to_affine(); }
  static J msm_j(const FrVec& s, const PtVec& p) { return msm_pippenger<Cv>(s.data(), p.data(), s.size()); }
  static Pt mul(const Fr& k, const Pt& P) { return scalar_mul<Cv>(k, P).to_affine(); }
  static Pt add(const Pt& a, const Pt& b) { return J::from_affine(a).add_mixed(b).to_affine(); }

  static Pt commit(const Params& pp, const FrVec& v, const Fr& r) {
    FrVec s(v);
    s.push_back(r);
    PtVec b(pp.ck.begin(), pp.ck.begin() + v.size());
    b.push_back(pp.H);
    return msm(s, b);
  }
  static Pt commit1(const Params& pp, const Fr& v, const Fr& r) { return commit(pp, FrVec{v}, r); }
  static Ct encrypt(const Params& pp, const Pt& pk, const Pt& M, const Fr& r) {  // (r*G, M + r*pk)
    Ct c;
    c.c0 = mul(r, pp.G);
    c.c1 = scalar_mul<Cv>(r, pk).add_mixed(M).to_affine();
    return c;
  }
  static Ct ct_add(const Ct& a, const Ct& b) { return Ct{add(a.c0, b.c0), add(a.c1, b.c1)}; }
  static Ct remask(const Params& pp, const Pt& pk, const Ct& c, const Fr& alpha) {
    return ct_add(c, encrypt(pp, pk, Pt::infinity(), alpha));
  }
  static Ct ct_msm(const FrVec& s, const Ct* cts, size_t n) {
    PtVec a(n), b(n);
    for (size_t i = 0; i < n; ++i) {
      a[i] = cts[i].c0;
      b[i] = cts[i].c1;
    }
    return Ct{msm(s, a), msm(s, b)};
  }
  static Pt compute_gsum(const PtVec& ck) {
    J acc = J::infinity();
    for (auto& P : ck) acc = acc.add_mixed(P);
    return acc.to_affine();
  }

  // ---------------------------------------------------------------- Fr helpers
  static FrVec powers(const Fr& x, size_t count) {  // x^0 .. x^(count-1)
    FrVec p(count);
    Fr acc = Fr::one();
    for (size_t i = 0; i < count; ++i) {
      p[i] = acc;
      acc = acc * x;
    }
    return p;
  }
  static Fr bilinear(const FrVec& a, const FrVec& b, const FrVec& ypow) {
    Fr acc = Fr::zero();
    for (size_t j = 0; j < a.size(); ++j) acc = acc + a[j] * b[j] * ypow[j];
    return acc;
  }
  template <class Rng>
  static FrVec rand_vec(Rng& rng, size_t k) {
    FrVec v(k);
    for (auto& x : v) x = field_rand<Fr>(rng);
    return v;
  }

  // ---------------------------------------------------------------- proof structures
  struct ZeroProof {
    Pt cA0, cBm;
    PtVec cD;
    FrVec abar, bbar;
    Fr rbar, sbar, tbar;
  };
  struct HadamardProof {
    PtVec cB;
    ZeroProof zero;
  };
  struct SvpProof {
    Pt cd, cdelta, cDelta;
    FrVec at, bt;
    Fr rt, st;
  };
  struct ProductProof {
    Pt cb;
    HadamardProof had;
    SvpProof svp;
  };
  struct MexpProof {
    Pt cA0;
    PtVec cB;
    Deck E;
    FrVec abar;
    Fr rbar, bbar, sbar, taubar;
  };
  struct Proof {
    PtVec cA, cB;
    ProductProof product;
    MexpProof mexp;
  };

  static size_t proof_size(uint32_t m, uint32_t n) { return (size_t)(11 * m + 8) * PW + (size_t)(5 * n + 9) * 32; }

  static void proof_to_bytes(const Proof& pf, uint8_t* out) {
    uint8_t* o = out;
    auto P = [&](const Pt& p) { pt_wire(p, o); o += PW; };
    auto S = [&](const Fr& s) { s.to_bytes(o); o += 32; };
    for (auto& p : pf.cA) P(p);
    for (auto& p : pf.cB) P(p);
    P(pf.product.cb);
    for (auto& p : pf.product.had.cB) P(p);
    const ZeroProof& z = pf.product.had.zero;
    P(z.cA0); P(z.cBm);
    for (auto& p : z.cD) P(p);
    for (auto& s : z.abar) S(s);
    for (auto& s : z.bbar) S(s);
    S(z.rbar); S(z.sbar); S(z.tbar);
    const SvpProof& sv = pf.product.svp;
    P(sv.cd); P(sv.cdelta); P(sv.cDelta);
    for (auto& s : sv.at) S(s);
    for (auto& s : sv.bt) S(s);
    S(sv.rt); S(sv.st);
    const MexpProof& me = pf.mexp;
    P(me.cA0);
    for (auto& p : me.cB) P(p);
    for (auto& e : me.E) { P(e.c0); P(e.c1); }
    for (auto& s : me.abar) S(s);
    S(me.rbar); S(me.bbar); S(me.sbar); S(me.taubar);
  }

  static bool proof_from_bytes(const uint8_t* in, uint32_t m, uint32_t n, Proof& pf) {
    const uint8_t* o = in;
    bool ok = true;
    auto P = [&](Pt& p) { ok &= pt_from_wire(o, p); o += PW; };
    auto S = [&](Fr& s) { ok &= Fr::from_bytes(o, s); o += 32; };
    pf.cA.resize(m); pf.cB.resize(m);
    for (auto& p : pf.cA) P(p);
    for (auto& p : pf.cB) P(p);
    P(pf.product.cb);
    pf.product.had.cB.resize(m);
    for (auto& p : pf.product.had.cB) P(p);
    ZeroProof& z = pf.product.had.zero;
    P(z.cA0); P(z.cBm);
    z.cD.resize(2 * m + 1);
    for (auto& p : z.cD) P(p);
    z.abar.resize(n); z.bbar.resize(n);
    for (auto& s : z.abar) S(s);
    for (auto& s : z.bbar) S(s);
    S(z.rbar); S(z.sbar); S(z.tbar);
    SvpProof& sv = pf.product.svp;
    P(sv.cd); P(sv.cdelta); P(sv.cDelta);
    sv.at.resize(n); sv.bt.resize(n);
    for (auto& s : sv.at) S(s);
    for (auto& s : sv.bt) S(s);
    S(sv.rt); S(sv.st);
    MexpProof& me = pf.mexp;
    P(me.cA0);
    me.cB.resize(2 * m);
    for (auto& p : me.cB) P(p);
    me.E.resize(2 * m);
    for (auto& e : me.E) { P(e.c0); P(e.c1); }
    me.abar.resize(n);
    for (auto& s : me.abar) S(s);
    S(me.rbar); S(me.bbar); S(me.sbar); S(me.taubar);
    return ok;
  }

  // ---------------------------------------------------------------- zero argument (5.2)
  static ZeroProof zero_prove(const Params& pp, ChaChaRng& prng, FsRng& fs, const std::vector<FrVec>& A,
                              const FrVec& r, const std::vector<FrVec>& B, const FrVec& s, const FrVec& ypow) {
    const size_t m = A.size(), n = pp.n;
    FrVec a0 = rand_vec(prng, n), bm = rand_vec(prng, n);
    Fr r0 = field_rand<Fr>(prng), sm = field_rand<Fr>(prng);
    FrVec t = rand_vec(prng, 2 * m + 1);
    t[m + 1] = Fr::zero();
    std::vector<FrVec> Aa, Bb;
    Aa.push_back(a0);
    for (auto& v : A) Aa.push_back(v);
    for (auto& v : B) Bb.push_back(v);
    Bb.push_back(bm);
    FrVec ra{r0}, sb(s);
    for (auto& v : r) ra.push_back(v);
    sb.push_back(sm);
    FrVec d(2 * m + 1, Fr::zero());
    for (size_t i = 0; i <= m; ++i)
      for (size_t j = 0; j <= m; ++j) d[m - j + i] = d[m - j + i] + bilinear(Aa[i], Bb[j], ypow);
    if (!d[m + 1].is_zero()) throw std::runtime_error("zero-argument witness does not satisfy the statement");
    ZeroProof pf;
    pf.cA0 = commit(pp, a0, r0);
    pf.cBm = commit(pp, bm, sm);
    for (size_t k = 0; k <= 2 * m; ++k) pf.cD.push_back(commit1(pp, d[k], t[k]));
    std::vector<uint8_t> buf;
    pt_tobytes(pf.cA0, buf);
    pt_tobytes(pf.cBm, buf);
    pts_tobytes(pf.cD, buf);
    fs.absorb(buf);
    Fr x = field_rand<Fr>(fs);
    FrVec xp = powers(x, 2 * m + 1);
    pf.abar.assign(n, Fr::zero());
    pf.bbar.assign(n, Fr::zero());
    pf.rbar = pf.sbar = pf.tbar = Fr::zero();
    for (size_t i = 0; i <= m; ++i) {
      for (size_t l = 0; l < n; ++l) pf.abar[l] = pf.abar[l] + xp[i] * Aa[i][l];
      pf.rbar = pf.rbar + xp[i] * ra[i];
    }
    for (size_t j = 0; j <= m; ++j) {
      for (size_t l = 0; l < n; ++l) pf.bbar[l] = pf.bbar[l] + xp[m - j] * Bb[j][l];
      pf.sbar = pf.sbar + xp[m - j] * sb[j];
    }
    for (size_t k = 0; k <= 2 * m; ++k) pf.tbar = pf.tbar + xp[k] * t[k];
    return pf;
  }

  static int zero_verify(const Params& pp, FsRng& fs, const PtVec& cA, const PtVec& cB, const FrVec& ypow,
                         const ZeroProof& pf) {
    const size_t m = cA.size();
    std::vector<uint8_t> buf;
    pt_tobytes(pf.cA0, buf);
    pt_tobytes(pf.cBm, buf);
    pts_tobytes(pf.cD, buf);
    fs.absorb(buf);
    Fr x = field_rand<Fr>(fs);
    FrVec xp = powers(x, 2 * m + 1);
    if (pf.cD.size() != 2 * m + 1 || pf.abar.size() != pp.n || pf.bbar.size() != pp.n) return ZERO_ARG;
    if (!pf.cD[m + 1].inf) return ZERO_ARG;
    {
      PtVec b{pf.cA0};
      b.insert(b.end(), cA.begin(), cA.end());
      FrVec s(xp.begin(), xp.begin() + m + 1);
      if (msm(s, b) != commit(pp, pf.abar, pf.rbar)) return ZERO_ARG;
    }
    {
      PtVec b(cB);
      b.push_back(pf.cBm);
      FrVec s(m + 1);
      for (size_t j = 0; j <= m; ++j) s[j] = xp[m - j];
      if (msm(s, b) != commit(pp, pf.bbar, pf.sbar)) return ZERO_ARG;
    }
    if (msm(xp, pf.cD) != commit1(pp, bilinear(pf.abar, pf.bbar, ypow), pf.tbar)) return ZERO_ARG;
    return OK;
  }

  // ---------------------------------------------------------------- Hadamard product argument (5.1)
  static HadamardProof hadamard_prove(const Params& pp, ChaChaRng& prng, FsRng& fs, const PtVec& cA, const Pt& cb,
                                      const std::vector<FrVec>& A, const FrVec& r, const FrVec& bvec, const Fr& sb) {
    const size_t m = A.size(), n = pp.n;
    std::vector<FrVec> Bp{A[0]};
    for (size_t i = 1; i < m; ++i) {
      FrVec v(n);
      for (size_t l = 0; l < n; ++l) v[l] = Bp.back()[l] * A[i][l];
      Bp.push_back(v);
    }
    FrVec s{r[0]};
    for (size_t i = 0; i + 2 < m; ++i) s.push_back(field_rand<Fr>(prng));
    s.push_back(sb);
    HadamardProof pf;
    pf.cB.push_back(cA[0]);
    for (size_t i = 1; i + 1 < m; ++i) pf.cB.push_back(commit(pp, Bp[i], s[i]));
    pf.cB.push_back(cb);
    std::vector<uint8_t> buf;
    pts_tobytes(pf.cB, buf);
    fs.absorb(buf);
    Fr x = field_rand<Fr>(fs);
    Fr y = field_rand<Fr>(fs);
    FrVec xp = powers(x, m + 1);
    FrVec ypow = powers(y, n + 1);
    ypow.erase(ypow.begin());  // y^1..y^n
    std::vector<FrVec> zA(A.begin() + 1, A.end());
    zA.push_back(FrVec(n, Fr::one().neg()));
    FrVec zr(r.begin() + 1, r.end());
    zr.push_back(Fr::zero());
    std::vector<FrVec> zB;
    FrVec zs;
    for (size_t i = 0; i + 1 < m; ++i) {
      FrVec v(n);
      for (size_t l = 0; l < n; ++l) v[l] = xp[i + 1] * Bp[i][l];
      zB.push_back(v);
      zs.push_back(xp[i + 1] * s[i]);
    }
    FrVec last(n, Fr::zero());
    Fr slast = Fr::zero();
    for (size_t i = 0; i + 1 < m; ++i) {
      for (size_t l = 0; l < n; ++l) last[l] = last[l] + xp[i + 1] * Bp[i + 1][l];
      slast = slast + xp[i + 1] * s[i + 1];
    }
    zB.push_back(last);
    zs.push_back(slast);
    pf.zero = zero_prove(pp, prng, fs, zA, zr, zB, zs, ypow);
    return pf;
  }

  static int hadamard_verify(const Params& pp, FsRng& fs, const PtVec& cA, const Pt& cb, const HadamardProof& pf) {
    const size_t m = cA.size(), n = pp.n;
    if (pf.cB.size() != m || pf.cB[0] != cA[0] || pf.cB[m - 1] != cb) return HADAMARD;
    std::vector<uint8_t> buf;
    pts_tobytes(pf.cB, buf);
    fs.absorb(buf);
    Fr x = field_rand<Fr>(fs);
    Fr y = field_rand<Fr>(fs);
    FrVec xp = powers(x, m + 1);
    FrVec ypow = powers(y, n + 1);
    ypow.erase(ypow.begin());
    PtVec zcA(cA.begin() + 1, cA.end());
    zcA.push_back(pp.gsum.neg());
    PtVec zcB;
    for (size_t i = 0; i + 1 < m; ++i) zcB.push_back(mul(xp[i + 1], pf.cB[i]));
    {
      FrVec s;
      PtVec b;
      for (size_t i = 0; i + 1 < m; ++i) {
        s.push_back(xp[i + 1]);
        b.push_back(pf.cB[i + 1]);
      }
      zcB.push_back(msm(s, b));
    }
    return zero_verify(pp, fs, zcA, zcB, ypow, pf.zero);
  }

  // ---------------------------------------------------------------- single value product argument (5.3)
  static SvpProof svp_prove(const Params& pp, ChaChaRng& prng, FsRng& fs, const Fr& b, const FrVec& a, const Fr& r) {
    const size_t n = pp.n;
    FrVec bp{a[0]};
    for (size_t i = 1; i < n; ++i) bp.push_back(bp.back() * a[i]);
    if (bp.back() != b) throw std::runtime_error("svp witness does not satisfy the statement");
    FrVec d = rand_vec(prng, n);
    Fr rd = field_rand<Fr>(prng);
    FrVec delta{d[0]};
    for (size_t i = 0; i + 2 < n; ++i) delta.push_back(field_rand<Fr>(prng));
    delta.push_back(Fr::zero());
    Fr s1 = field_rand<Fr>(prng), sx = field_rand<Fr>(prng);
    SvpProof pf;
    pf.cd = commit(pp, d, rd);
    FrVec v1(n - 1), v2(n - 1);
    for (size_t i = 0; i + 1 < n; ++i) {
      v1[i] = (delta[i] * d[i + 1]).neg();
      v2[i] = delta[i + 1] - a[i + 1] * delta[i] - bp[i] * d[i + 1];
    }
    pf.cdelta = commit(pp, v1, s1);
    pf.cDelta = commit(pp, v2, sx);
    std::vector<uint8_t> buf;
    pt_tobytes(pf.cd, buf);
    pt_tobytes(pf.cdelta, buf);
    pt_tobytes(pf.cDelta, buf);
    fs.absorb(buf);
    Fr x = field_rand<Fr>(fs);
    pf.at.resize(n);
    pf.bt.resize(n);
    for (size_t i = 0; i < n; ++i) {
      pf.at[i] = x * a[i] + d[i];
      pf.bt[i] = x * bp[i] + delta[i];
    }
    pf.rt = x * r + rd;
    pf.st = x * sx + s1;
    return pf;
  }

  static int svp_verify(const Params& pp, FsRng& fs, const Pt& ca, const Fr& b, const SvpProof& pf) {
    const size_t n = pp.n;
    std::vector<uint8_t> buf;
    pt_tobytes(pf.cd, buf);
    pt_tobytes(pf.cdelta, buf);
    pt_tobytes(pf.cDelta, buf);
    fs.absorb(buf);
    Fr x = field_rand<Fr>(fs);
    if (pf.at.size() != n || pf.bt.size() != n) return SVP;
    if (msm(FrVec{x, Fr::one()}, PtVec{ca, pf.cd}) != commit(pp, pf.at, pf.rt)) return SVP;
    FrVec v(n - 1);
    for (size_t i = 0; i + 1 < n; ++i) v[i] = x * pf.bt[i + 1] - pf.bt[i] * pf.at[i + 1];
    if (msm(FrVec{x, Fr::one()}, PtVec{pf.cDelta, pf.cdelta}) != commit(pp, v, pf.st)) return SVP;
    if (pf.bt[0] != pf.at[0] || pf.bt[n - 1] != x * b) return SVP;
    return OK;
  }

  // ---------------------------------------------------------------- product argument (5)
  static ProductProof product_prove(const Params& pp, ChaChaRng& prng, FsRng& fs, const PtVec& cA, const Fr& b,
                                    const std::vector<FrVec>& A, const FrVec& r) {
    const size_t m = A.size(), n = pp.n;
    FrVec bvec(A[0]);
    for (size_t i = 1; i < m; ++i)
      for (size_t l = 0; l < n; ++l) bvec[l] = bvec[l] * A[i][l];
    Fr sb = field_rand<Fr>(prng);
    ProductProof pf;
    pf.cb = commit(pp, bvec, sb);
    std::vector<uint8_t> buf;
    pt_tobytes(pf.cb, buf);
    fs.absorb(buf);
    pf.had = hadamard_prove(pp, prng, fs, cA, pf.cb, A, r, bvec, sb);
    pf.svp = svp_prove(pp, prng, fs, b, bvec, sb);
    return pf;
  }
  static int product_verify(const Params& pp, FsRng& fs, const PtVec& cA, const Fr& b, const ProductProof& pf) {
    std::vector<uint8_t> buf;
    pt_tobytes(pf.cb, buf);
    fs.absorb(buf);
    int rc = hadamard_verify(pp, fs, cA, pf.cb, pf.had);
    if (rc) return rc;
    return svp_verify(pp, fs, pf.cb, b, pf.svp);
  }

  // ---------------------------------------------------------------- multi-exponentiation argument (4)
  static MexpProof mexp_prove(const Params& pp, const Pt& pk, ChaChaRng& prng, FsRng& fs, const Deck& Cp, const Ct& C,
                              const std::vector<FrVec>& A, const FrVec& r, const Fr& rho) {
    const size_t m = A.size(), n = pp.n;
    FrVec a0 = rand_vec(prng, n);
    Fr r0 = field_rand<Fr>(prng);
    FrVec b = rand_vec(prng, 2 * m), s = rand_vec(prng, 2 * m), tau = rand_vec(prng, 2 * m);
    b[m] = Fr::zero();
    s[m] = Fr::zero();
    tau[m] = rho;
    std::vector<FrVec> Aa{a0};
    for (auto& v : A) Aa.push_back(v);
    MexpProof pf;
    pf.cA0 = commit(pp, a0, r0);
    for (size_t k = 0; k < 2 * m; ++k) pf.cB.push_back(commit1(pp, b[k], s[k]));
    for (size_t k = 0; k < 2 * m; ++k) {
      Ct acc;
      acc.c0 = mul(tau[k], pp.G);
      acc.c1 = add(mul(b[k], pp.gen), mul(tau[k], pk));
      for (size_t i = 1; i <= m; ++i) {
        long j = (long)k - (long)m + (long)i;
        if (j < 0 || j > (long)m) continue;
        acc = ct_add(acc, ct_msm(Aa[j], &Cp[(i - 1) * n], n));
      }
      pf.E.push_back(acc);
    }
    if (pf.E[m] != C) throw std::runtime_error("multi-exp witness does not open the statement");
    std::vector<uint8_t> buf;
    pt_tobytes(pf.cA0, buf);
    pts_tobytes(pf.cB, buf);
    for (auto& e : pf.E) {
      pt_tobytes(e.c0, buf);
      pt_tobytes(e.c1, buf);
    }
    fs.absorb(buf);
    Fr x = field_rand<Fr>(fs);
    FrVec xp = powers(x, 2 * m);
    FrVec ra{r0};
    for (auto& v : r) ra.push_back(v);
    pf.abar.assign(n, Fr::zero());
    pf.rbar = pf.bbar = pf.sbar = pf.taubar = Fr::zero();
    for (size_t j = 0; j <= m; ++j) {
      for (size_t l = 0; l < n; ++l) pf.abar[l] = pf.abar[l] + xp[j] * Aa[j][l];
      pf.rbar = pf.rbar + xp[j] * ra[j];
    }
    for (size_t k = 0; k < 2 * m; ++k) {
      pf.bbar = pf.bbar + xp[k] * b[k];
      pf.sbar = pf.sbar + xp[k] * s[k];
      pf.taubar = pf.taubar + xp[k] * tau[k];
    }
    return pf;
  }

  static int mexp_verify(const Params& pp, const Pt& pk, FsRng& fs, const Deck& Cp, const Ct& C, const PtVec& cA,
                         const MexpProof& pf) {
    const size_t m = cA.size(), n = pp.n;
    std::vector<uint8_t> buf;
    pt_tobytes(pf.cA0, buf);
    pts_tobytes(pf.cB, buf);
    for (auto& e : pf.E) {
      pt_tobytes(e.c0, buf);
      pt_tobytes(e.c1, buf);
    }
    fs.absorb(buf);
    Fr x = field_rand<Fr>(fs);
    FrVec xp = powers(x, 2 * m);
    if (pf.cB.size() != 2 * m || pf.E.size() != 2 * m || pf.abar.size() != n) return MULTIEXP;
    if (!pf.cB[m].inf) return MULTIEXP;
    if (pf.E[m] != C) return MULTIEXP;
    {
      PtVec b{pf.cA0};
      b.insert(b.end(), cA.begin(), cA.end());
      FrVec s(xp.begin(), xp.begin() + m + 1);
      if (msm(s, b) != commit(pp, pf.abar, pf.rbar)) return MULTIEXP;
    }
    if (msm(xp, pf.cB) != commit1(pp, pf.bbar, pf.sbar)) return MULTIEXP;
    Ct lhs = ct_msm(xp, pf.E.data(), 2 * m);
    Ct rhs;
    rhs.c0 = mul(pf.taubar, pp.G);
    rhs.c1 = add(mul(pf.bbar, pp.gen), mul(pf.taubar, pk));
    for (size_t i = 1; i <= m; ++i) {
      FrVec s(n);
      for (size_t l = 0; l < n; ++l) s[l] = xp[m - i] * pf.abar[l];
      rhs = ct_add(rhs, ct_msm(s, &Cp[(i - 1) * n], n));
    }
    if (lhs != rhs) return MULTIEXP;
    return OK;
  }

  // ---------------------------------------------------------------- shuffle argument
  static void statement_bytes(const Params& pp, const Pt& pk, const Deck& deck, const Deck& shuffled,
                              std::vector<uint8_t>& out) {
    pt_tobytes(pp.G, out);
    pt_tobytes(pk, out);
    pt_tobytes(pp.gen, out);
    pts_tobytes(pp.ck, out);
    pt_tobytes(pp.H, out);
    for (auto& c : deck) { pt_tobytes(c.c0, out); pt_tobytes(c.c1, out); }
    for (auto& c : shuffled) { pt_tobytes(c.c0, out); pt_tobytes(c.c1, out); }
    uint64_t mn[2] = {pp.m, pp.n};
    const uint8_t* p = (const uint8_t*)mn;
    out.insert(out.end(), p, p + 16);
  }

  static Fr product_value(const Fr& x, const Fr& y, const Fr& z, size_t N) {
    Fr prod = Fr::one(), xi = Fr::one(), yi = Fr::zero();
    for (size_t i = 1; i <= N; ++i) {
      xi = xi * x;
      yi = yi + y;
      prod = prod * (yi + xi - z);
    }
    return prod;
  }

  static Proof prove(const Params& pp, const Pt& pk, const Deck& deck, const Deck& shuffled,
                     const std::vector<uint32_t>& perm, const FrVec& rho, ChaChaRng& prng) {
    const size_t m = pp.m, n = pp.n, N = m * n;
    static const uint8_t seed[] = "Shuffle Proof";  // [REF mod.rs:84]
    FsRng fs(seed, 13);
    std::vector<uint8_t> buf;
    statement_bytes(pp, pk, deck, shuffled, buf);
    fs.absorb(buf);
    FrVec r = rand_vec(prng, m), s = rand_vec(prng, m);
    FrVec a(N);
    for (size_t i = 0; i < N; ++i) a[i] = Fr::from_u64(perm[i] + 1);
    Proof pf;
    for (size_t k = 0; k < m; ++k) pf.cA.push_back(commit(pp, FrVec(a.begin() + k * n, a.begin() + (k + 1) * n), r[k]));
    buf.clear();
    pts_tobytes(pf.cA, buf);
    fs.absorb(buf);
    Fr x = field_rand<Fr>(fs);
    FrVec xp = powers(x, N + 1);
    FrVec b(N);
    for (size_t i = 0; i < N; ++i) b[i] = xp[perm[i] + 1];
    for (size_t k = 0; k < m; ++k) pf.cB.push_back(commit(pp, FrVec(b.begin() + k * n, b.begin() + (k + 1) * n), s[k]));
    buf.clear();
    pts_tobytes(pf.cB, buf);
    fs.absorb(buf);
    Fr y = field_rand<Fr>(fs);
    Fr z = field_rand<Fr>(fs);
    std::vector<FrVec> dz(m, FrVec(n));
    FrVec t(m);
    PtVec cDz;
    for (size_t k = 0; k < m; ++k) {
      for (size_t l = 0; l < n; ++l) dz[k][l] = y * a[k * n + l] + b[k * n + l] - z;
      t[k] = y * r[k] + s[k];
      cDz.push_back(msm(FrVec{y, Fr::one(), z.neg()}, PtVec{pf.cA[k], pf.cB[k], pp.gsum}));
    }
    pf.product = product_prove(pp, prng, fs, cDz, product_value(x, y, z, N), dz, t);
    Fr rho_hat = Fr::zero();
    for (size_t i = 0; i < N; ++i) rho_hat = rho_hat - rho[i] * b[i];
    Ct Cx = ct_msm(FrVec(xp.begin() + 1, xp.end()), deck.data(), N);
    std::vector<FrVec> brows;
    for (size_t k = 0; k < m; ++k) brows.push_back(FrVec(b.begin() + k * n, b.begin() + (k + 1) * n));
    pf.mexp = mexp_prove(pp, pk, prng, fs, shuffled, Cx, brows, s, rho_hat);
    return pf;
  }

  static int verify(const Params& pp, const Pt& pk, const Deck& deck, const Deck& shuffled, const Proof& pf) {
    const size_t m = pp.m, n = pp.n, N = m * n;
    static const uint8_t seed[] = "Shuffle Proof";
    FsRng fs(seed, 13);
    std::vector<uint8_t> buf;
    statement_bytes(pp, pk, deck, shuffled, buf);
    fs.absorb(buf);
    buf.clear();
    pts_tobytes(pf.cA, buf);
    fs.absorb(buf);
    Fr x = field_rand<Fr>(fs);
    buf.clear();
    pts_tobytes(pf.cB, buf);
    fs.absorb(buf);
    Fr y = field_rand<Fr>(fs);
    Fr z = field_rand<Fr>(fs);
    PtVec cDz;
    for (size_t k = 0; k < m; ++k) cDz.push_back(msm(FrVec{y, Fr::one(), z.neg()}, PtVec{pf.cA[k], pf.cB[k], pp.gsum}));
    int rc = product_verify(pp, fs, cDz, product_value(x, y, z, N), pf.product);
    if (rc) return rc;
    FrVec xp = powers(x, N + 1);
    Ct Cx = ct_msm(FrVec(xp.begin() + 1, xp.end()), deck.data(), N);
    return mexp_verify(pp, pk, fs, shuffled, Cx, pf.cB, pf.mexp);
  }

  // ---------------------------------------------------------------- boundary functions
  static void shuffle_and_remask(const Params& pp, const Pt& pk, const Deck& deck, const FrVec& rho,
                                 const std::vector<uint32_t>& perm, const uint8_t prover_seed[32], Deck& out,
                                 Proof& proof) {
    const size_t N = deck.size();
    out.resize(N);
    for (size_t i = 0; i < N; ++i) out[i] = remask(pp, pk, deck[perm[i]], rho[i]);  // permute_array then remask
    ChaChaRng prng(prover_seed);
    proof = prove(pp, pk, deck, out, perm, rho, prng);
  }

  // ---------------------------------------------------------------- sigma protocols
  // Schnorr identification (1 base) / Chaum-Pedersen DL equality (2 bases) behind
  // DLCards::{prove,verify}_key_ownership, mask, remask, compute_reveal_token and their verifiers
  // [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:132-357]; "sigma transcript v1" as frozen in
  // oracle/py/mp_oracle.py: A_i = r g_i ; absorb(g.., a.., A..) ; c ; z = r + c x ; check z g_i == A_i + c a_i.
  struct SigmaProof {
    PtVec A;
    Fr z;
  };
  static SigmaProof sigma_prove(const PtVec& bases, const PtVec& publics, const Fr& x, const uint8_t* fs_init,
                                size_t fs_init_len, const uint8_t prover_seed[32]) {
    // hedged nonce ("sigma transcript v2"): s1 = Blake2s(x || Blake2s(fs_init) || seed), s2 = Blake2s(ToBytes(bases, publics) || s1)
    uint8_t s1[32], s2[32], fsd[32], xb[32];
    Blake2s::digest(fs_init, fs_init_len, fsd);
    x.to_bytes(xb);
    {
      Blake2s b;
      b.update(xb, 32);
      b.update(fsd, 32);
      b.update(prover_seed, 32);
      b.final(s1);
    }
    {
      std::vector<uint8_t> sb;
      pts_tobytes(bases, sb);
      pts_tobytes(publics, sb);
      Blake2s b;
      b.update(sb.data(), sb.size());
      b.update(s1, 32);
      b.final(s2);
    }
    ChaChaRng prng(s2);
    Fr r = field_rand<Fr>(prng);
    SigmaProof pf;
    for (auto& g : bases) pf.A.push_back(mul(r, g));
    FsRng fs(fs_init, fs_init_len);
    std::vector<uint8_t> buf;
    pts_tobytes(bases, buf);
    pts_tobytes(publics, buf);
    pts_tobytes(pf.A, buf);
    fs.absorb(buf);
    Fr c = field_rand<Fr>(fs);
    pf.z = r + c * x;
    return pf;
  }
  static bool sigma_verify(const PtVec& bases, const PtVec& publics, const SigmaProof& pf, const uint8_t* fs_init,
                           size_t fs_init_len) {
    if (pf.A.size() != bases.size() || publics.size() != bases.size()) return false;
    FsRng fs(fs_init, fs_init_len);
    std::vector<uint8_t> buf;
    pts_tobytes(bases, buf);
    pts_tobytes(publics, buf);
    pts_tobytes(pf.A, buf);
    fs.absorb(buf);
    Fr c = field_rand<Fr>(fs);
    for (size_t i = 0; i < bases.size(); ++i)
      if (mul(pf.z, bases[i]) != add(pf.A[i], mul(c, publics[i]))) return false;
    return true;
  }

  // ---------------------------------------------------------------- synthetic inputs
  // ---- `C::rand(rng)` of ark-ec 0.3 [UPSTREAM-RECALL]: x = Fq::rand, greatest = rng.gen::<bool>(), lift, scale by the cofactor.
  // No discrete logarithm of the result is known to anybody ("setup v2": round 1 used k*G_std, a commitment trapdoor).
  static Fq fq_rand(ChaChaRng& rng) {
    const int NL = Fq::N, shave = 64 * NL - Fq::C().bits;
    for (;;) {
      u64 a[Fq::N];
      for (int i = 0; i < NL; ++i) a[i] = rng.next_u64();
      if (shave) a[NL - 1] &= (~(u64)0) >> shave;
      if (cmpN(a, Fq::modulus(), NL) >= 0) continue;
      Fq f;                                   // the accepted limbs ARE the Montgomery representation
      memcpy(f.v, a, 8 * NL);
      return f;
    }
  }
  // Tonelli-Shanks; false if `a` is not a square
  static bool fq_sqrt(const Fq& a, Fq& out) {
    const int NL = Fq::N;
    if (a.is_zero()) { out = a; return true; }
    u64 pm1[Fq::N], half[Fq::N], t[Fq::N], t1h[Fq::N], one[Fq::N] = {1};
    subN(pm1, Fq::modulus(), one, NL);
    auto shr1 = [&](u64* r, const u64* x) { for (int i = 0; i < NL; ++i) r[i] = (x[i] >> 1) | (i + 1 < NL ? x[i + 1] << 63 : 0); };
    shr1(half, pm1);
    const Fq minus_one = Fq::one().neg();
    if (a.pow(half) != Fq::one()) return false;
    int s = 0;
    memcpy(t, pm1, 8 * NL);
    while (!(t[0] & 1)) { shr1(t, t); ++s; }
    addN(t1h, t, one, NL);
    shr1(t1h, t1h);
    Fq z = Fq::from_u64(2);
    for (u64 k = 2; z.pow(half) != minus_one; ) z = Fq::from_u64(++k);
    Fq c = z.pow(t), r = a.pow(t1h), tt = a.pow(t);
    int M = s;
    while (tt != Fq::one()) {
      int i = 0;
      Fq u = tt;
      while (u != Fq::one()) { u = u.sqr(); ++i; }
      Fq b = c;
      for (int k = 0; k < M - i - 1; ++k) b = b.sqr();
      r = r * b;
      c = b.sqr();
      tt = tt * c;
      M = i;
    }
    out = r;
    return true;
  }
  static Pt point_rand(ChaChaRng& rng) {
    for (;;) {
      Fq x = fq_rand(rng);
      const bool greatest = (rng.next_u32() >> 31) & 1;
      Fq rhs = x.sqr() * x + Fq::from_u256(Cv::B);
      if (Cv::A == 1) rhs = rhs + x;
      Fq y;
      if (!fq_sqrt(rhs, y)) continue;
      Fq ny = y.neg();
      u64 yi[Fq::N], nyi[Fq::N];
      y.to_u256(yi);
      ny.to_u256(nyi);
      const bool y_is_larger = cmpN(yi, nyi, Fq::N) > 0;
      Pt p;
      p.x = x;
      p.y = (y_is_larger == greatest) ? y : ny;
      p.inf = false;
      if (Cv::ID == 3) {                       // BLS12-377 G1: scale by the cofactor 0x170b5d44300000000000000000000000
        const u64 h[2] = {0x0000000000000000ull, 0x170b5d4430000000ull};
        Jac<Cv> acc = Jac<Cv>::infinity();
        for (int i = 127; i >= 0; --i) {
          acc = acc.dbl();
          if ((h[i / 64] >> (i % 64)) & 1) acc = acc.add_mixed(p);
        }
        p = acc.to_affine();
      }
      return p;
    }
  }

  // ---------------------------------------------------------------- synthetic inputs
  // DLCards::setup [REF mod.rs:105-121]: G, ck_0..ck_{n-1}, H, gen -- independent `C::rand` points in that order
  static Params setup(uint32_t m, uint32_t n, ChaChaRng& rng) {
    Params pp;
    pp.m = m;
    pp.n = n;
    pp.G = point_rand(rng);
    for (uint32_t i = 0; i < n; ++i) pp.ck.push_back(point_rand(rng));
    pp.H = point_rand(rng);
    pp.gen = point_rand(rng);
    pp.gsum = compute_gsum(pp.ck);
    return pp;
  }
};

}  // namespace mpo
