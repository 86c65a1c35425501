// TEST INFRASTRUCTURE ONLY (oracle).  Short-Weierstrass group law in Jacobian coordinates plus the two
// multiplication algorithms the reference's CPU path bottoms out in (arkworks 0.3):
//   * `ProjectiveCurve::mul`  = MSB-first double-and-add  (the 2N re-encryption scalar-muls of
//     [REF barnett-smart-card-protocol/src/discrete_log_cards/remasking.rs:16-18] / masking.rs:17)
//   * `VariableBaseMSM::multi_scalar_mul` = bucket method, c = 3 if size < 32 else ln_without_floats+2
// Curve constants: SURVEY.md App. C.  PARITY UNPINNED (see oracle/README.md).
#pragma once
#include <vector>

#include "field.hpp"

namespace mpo {

struct StarkFq { static const int NL = 4; static const u64 MOD[4]; };
struct StarkFr { static const int NL = 4; static const u64 MOD[4]; };
struct Bn254Fq { static const int NL = 4; static const u64 MOD[4]; };
struct Bn254Fr { static const int NL = 4; static const u64 MOD[4]; };
struct SecpFq { static const int NL = 4; static const u64 MOD[4]; };
struct SecpFr { static const int NL = 4; static const u64 MOD[4]; };
struct Bls377Fq { static const int NL = 6; static const u64 MOD[6]; };   // 377-bit base field: ark-ff Fp384
struct Bls377Fr { static const int NL = 4; static const u64 MOD[4]; };

struct Stark {
  typedef Fp<StarkFq> Fq;
  typedef Fp<StarkFr> Fr;
  static const int A = 1;
  static const u64 B[4], GX[4], GY[4];
  static const int ID = 0;
};
struct Bn254 {
  typedef Fp<Bn254Fq> Fq;
  typedef Fp<Bn254Fr> Fr;
  static const int A = 0;
  static const u64 B[4], GX[4], GY[4];
  static const int ID = 1;
};
struct Secp256k1 {
  typedef Fp<SecpFq> Fq;
  typedef Fp<SecpFr> Fr;
  static const int A = 0;
  static const u64 B[4], GX[4], GY[4];
  static const int ID = 2;
};

// BLS12-377 G1 [REF barnett-smart-card-protocol/examples/parameter_selection.rs:25]; cofactor != 1 -- the protocol only
// ever sees multiples of the generator (prime-order subgroup)
struct Bls12_377 {
  typedef Fp<Bls377Fq> Fq;
  typedef Fp<Bls377Fr> Fr;
  static const int A = 0;
  static const u64 B[6], GX[6], GY[6];
  static const int ID = 3;
};

template <class Cv>
struct Affine {
  typename Cv::Fq x, y;
  bool inf;
  static Affine infinity() {
    Affine a;
    a.x = Cv::Fq::zero();
    a.y = Cv::Fq::zero();
    a.inf = true;
    return a;
  }
  static Affine generator() {
    Affine a;
    a.x = Cv::Fq::from_u256(Cv::GX);
    a.y = Cv::Fq::from_u256(Cv::GY);
    a.inf = false;
    return a;
  }
  bool operator==(const Affine& o) const {
    if (inf || o.inf) return inf == o.inf;
    return x == o.x && y == o.y;
  }
  bool operator!=(const Affine& o) const { return !(*this == o); }
  Affine neg() const {
    Affine a = *this;
    if (!inf) a.y = y.neg();
    return a;
  }
  bool on_curve() const {
    if (inf) return true;
    typename Cv::Fq rhs = x.sqr() * x + Cv::Fq::from_u256(Cv::B);
    if (Cv::A == 1) rhs = rhs + x;
    return y.sqr() == rhs;
  }
};

template <class Cv>
struct Jac {
  typedef typename Cv::Fq Fq;
  Fq X, Y, Z;
  static Jac infinity() {
    Jac j;
    j.X = Fq::one();
    j.Y = Fq::one();
    j.Z = Fq::zero();
    return j;
  }
  static Jac from_affine(const Affine<Cv>& a) {
    if (a.inf) return infinity();
    Jac j;
    j.X = a.x;
    j.Y = a.y;
    j.Z = Fq::one();
    return j;
  }
  bool is_inf() const { return Z.is_zero(); }

  Jac dbl() const {
    if (is_inf() || Y.is_zero()) return infinity();
    Fq XX = X.sqr(), YY = Y.sqr();
    Fq S = (X * YY).dbl().dbl();
    Fq M = XX.dbl() + XX;
    if (Cv::A == 1) M = M + Z.sqr().sqr();
    Jac r;
    r.X = M.sqr() - S.dbl();
    Fq YYYY8 = YY.sqr().dbl().dbl().dbl();
    r.Y = M * (S - r.X) - YYYY8;
    r.Z = (Y * Z).dbl();
    return r;
  }
  Jac add(const Jac& o) const {
    if (is_inf()) return o;
    if (o.is_inf()) return *this;
    Fq Z1Z1 = Z.sqr(), Z2Z2 = o.Z.sqr();
    Fq U1 = X * Z2Z2, U2 = o.X * Z1Z1;
    Fq S1 = Y * o.Z * Z2Z2, S2 = o.Y * Z * Z1Z1;
    if (U1 == U2) {
      if (S1 == S2) return dbl();
      return infinity();
    }
    Fq H = U2 - U1, R = S2 - S1;
    Fq HH = H.sqr(), HHH = H * HH, V = U1 * HH;
    Jac r;
    r.X = R.sqr() - HHH - V.dbl();
    r.Y = R * (V - r.X) - S1 * HHH;
    r.Z = Z * o.Z * H;
    return r;
  }
  Jac add_mixed(const Affine<Cv>& o) const {  // `add_assign_mixed`
    if (o.inf) return *this;
    if (is_inf()) return from_affine(o);
    Fq Z1Z1 = Z.sqr();
    Fq U2 = o.x * Z1Z1, S2 = o.y * Z * Z1Z1;
    if (X == U2) {
      if (Y == S2) return dbl();
      return infinity();
    }
    Fq H = U2 - X, R = S2 - Y;
    Fq HH = H.sqr(), HHH = H * HH, V = X * HH;
    Jac r;
    r.X = R.sqr() - HHH - V.dbl();
    r.Y = R * (V - r.X) - Y * HHH;
    r.Z = Z * H;
    return r;
  }
  Affine<Cv> to_affine() const {
    if (is_inf()) return Affine<Cv>::infinity();
    Fq zi = Z.inverse(), zi2 = zi.sqr();
    Affine<Cv> a;
    a.x = X * zi2;
    a.y = Y * zi2 * zi;
    a.inf = false;
    return a;
  }
};

// canonical scalar bits helpers
static inline int bit_at(const u64 k[4], int i) { return (int)((k[i / 64] >> (i % 64)) & 1); }

// ark-ec `mul`: MSB-first double-and-add on the canonical scalar.
template <class Cv>
Jac<Cv> scalar_mul(const typename Cv::Fr& k, const Affine<Cv>& P) {
  u64 e[4];
  k.to_u256(e);
  Jac<Cv> acc = Jac<Cv>::infinity();
  bool started = false;
  for (int i = 255; i >= 0; --i) {
    if (started) acc = acc.dbl();
    if (bit_at(e, i)) {
      acc = acc.add_mixed(P);
      started = true;
    }
  }
  return acc;
}

static inline int log2_ceil(size_t x) {
  int l = 0;
  while (((size_t)1 << l) < x) ++l;
  return l;
}

// ark-ec 0.3 `VariableBaseMSM::multi_scalar_mul` (sequential form), restated.
template <class Cv>
Jac<Cv> msm_pippenger(const typename Cv::Fr* scalars, const Affine<Cv>* bases, size_t size) {
  typedef Jac<Cv> J;
  if (size == 0) return J::infinity();
  const int c = size < 32 ? 3 : (log2_ceil(size) * 69 / 100) + 2;
  const int num_bits = Cv::Fr::C().bits;
  std::vector<U256> ks(size);
  for (size_t i = 0; i < size; ++i) scalars[i].to_u256(ks[i].l);
  const u64 one[4] = {1, 0, 0, 0};
  std::vector<J> window_sums;
  std::vector<J> buckets(((size_t)1 << c) - 1);
  for (int w_start = 0; w_start < num_bits; w_start += c) {
    J res = J::infinity();
    for (auto& b : buckets) b = J::infinity();
    for (size_t i = 0; i < size; ++i) {
      const u64* k = ks[i].l;
      if ((k[0] | k[1] | k[2] | k[3]) == 0) continue;
      if (cmpN(k, one, 4) == 0) {
        if (w_start == 0) res = res.add_mixed(bases[i]);
        continue;
      }
      // (k >> w_start) mod 2^c
      u64 d = 0;
      for (int bpos = 0; bpos < c; ++bpos) {
        int bit = w_start + bpos;
        if (bit < 256) d |= (u64)bit_at(k, bit) << bpos;
      }
      if (d) buckets[d - 1] = buckets[d - 1].add_mixed(bases[i]);
    }
    J running = J::infinity();
    for (size_t b = buckets.size(); b-- > 0;) {
      running = running.add(buckets[b]);
      res = res.add(running);
    }
    window_sums.push_back(res);
  }
  J total = J::infinity();
  for (size_t w = window_sums.size(); w-- > 1;) {
    total = total.add(window_sums[w]);
    for (int i = 0; i < c; ++i) total = total.dbl();
  }
  return window_sums[0].add(total);
}

}  // namespace mpo
