// Short-Weierstrass group law for the gfx950 engine.  Two accumulator forms, affine operands:
//   * XYZZ (X, Y, ZZ, ZZZ) in the MSM loops: mixed addition 8M+2S, doubling 6M+4S, each with one fused product pair;
//   * Jacobian for partial sums and their combination: mixed addition 8M+3S, doubling 3M+6S, full addition 11M+5S.
// Replaces ark-ec 0.3 `GroupProjective` add_assign_mixed / double_in_place / add_assign used by the
// reference's ElGamal, Pedersen and shuffle-argument calls
// [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:7, 380-443].
// Infinity: Jacobian Z = 0; affine (0, 0) (on none of the supported curves because b != 0).
// Every operation is complete on the prime-order group: P + P, P + (-P) and infinity operands take
// the rare branches below, so results are bit-exact for adversarial inputs (duplicate cards, rho = 0).
#pragma once
#include "field.hpp"

namespace mp {

template <class C>
struct Aff {
  Fe<typename C::FqP> x, y;
};
template <class C>
struct Jac {
  Fe<typename C::FqP> X, Y, Z;
};

// cofactor of the curve's group (ONE = prime order): BLS12-377 G1 has h = 0x170b5d44300000000000000000000000
template <class C>
struct Cofactor {
  static constexpr bool ONE = C::ID != 3;
  static constexpr uint32_t H[4] = {0x00000000u, 0x00000000u, 0x30000000u, 0x170b5d44u};
};

template <class C>
MP_HD bool aff_is_inf(const Aff<C>& a) {
  return fe_is_zero(a.x) && fe_is_zero(a.y);
}
template <class C>
MP_HD Aff<C> aff_inf() {
  Aff<C> a;
  a.x = fe_zero<typename C::FqP>();
  a.y = fe_zero<typename C::FqP>();
  return a;
}
template <class C>
MP_HD Aff<C> aff_neg(const Aff<C>& a) {
  Aff<C> r;
  r.x = a.x;
  r.y = fe_neg<typename C::FqP>(a.y);  // -0 = 0 keeps infinity
  return r;
}
template <class C>
MP_HD bool aff_eq(const Aff<C>& a, const Aff<C>& b) {
  return fe_eq(a.x, b.x) && fe_eq(a.y, b.y);
}
template <class C>
MP_HD bool aff_on_curve(const Aff<C>& a) {
  typedef typename C::FqP F;
  if (aff_is_inf<C>(a)) return true;
  const Fe<F> b = fe_unpack<F>(C::B_MONT);
  Fe<F> rhs = fe_add<F>(fe_mul<F>(fe_sqr<F>(a.x), a.x), b);
  if (C::A == 1) rhs = fe_add<F>(rhs, a.x);
  return fe_eq(fe_sqr<F>(a.y), rhs);
}

template <class C>
MP_HD Jac<C> jac_inf() {
  Jac<C> j;
  j.X = fe_one<typename C::FqP>();
  j.Y = fe_one<typename C::FqP>();
  j.Z = fe_zero<typename C::FqP>();
  return j;
}
template <class C>
MP_HD bool jac_is_inf(const Jac<C>& j) {
  return fe_is_zero(j.Z);
}
template <class C>
MP_HD Jac<C> jac_from_aff(const Aff<C>& a) {
  if (aff_is_inf<C>(a)) return jac_inf<C>();
  Jac<C> j;
  j.X = a.x;
  j.Y = a.y;
  j.Z = fe_one<typename C::FqP>();
  return j;
}

// The group law updates the accumulator IN PLACE and has a single exit per special case: returning the struct
// by value from several paths made hipcc keep two field elements on the stack (scratch traffic in the MSM loops).

// p <- 2p.  (Y = 0 cannot happen on a prime-order group of odd order, but is handled.)
template <class C>
MP_HD void jac_dbl_ip(Jac<C>& p) {
  typedef typename C::FqP F;
  if (fe_is_zero(p.Z)) return;
  if (fe_is_zero(p.Y)) {
    p.Z = fe_zero<F>();
    return;
  }
  const Fe<F> XX = fe_sqr<F>(p.X), YY = fe_sqr<F>(p.Y);
  const Fe<F> S = fe_dbl<F>(fe_dbl<F>(fe_mul<F>(p.X, YY)));
  Fe<F> M = fe_add<F>(fe_dbl<F>(XX), XX);
  if (C::A == 1) M = fe_add<F>(M, fe_sqr<F>(fe_sqr<F>(p.Z)));
  const Fe<F> X3 = fe_sub<F>(fe_sqr<F>(M), fe_dbl<F>(S));
  const Fe<F> Y8 = fe_dbl<F>(fe_dbl<F>(fe_dbl<F>(fe_sqr<F>(YY))));
  p.Z = fe_dbl<F>(fe_mul<F>(p.Y, p.Z));
  p.Y = fe_sub<F>(fe_mul<F>(M, fe_sub<F>(S, X3)), Y8);
  p.X = X3;
}
template <class C>
MP_HD Jac<C> jac_dbl(const Jac<C>& p) {
  Jac<C> r = p;
  jac_dbl_ip<C>(r);
  return r;
}

// p <- p + q, q affine
template <class C>
MP_HD void jac_madd_ip(Jac<C>& p, const Aff<C>& q) {
  typedef typename C::FqP F;
  if (aff_is_inf<C>(q)) return;
  if (fe_is_zero(p.Z)) {
    p.X = q.x;
    p.Y = q.y;
    p.Z = fe_one<F>();
    return;
  }
  const Fe<F> Z1Z1 = fe_sqr<F>(p.Z);
  const Fe<F> U2 = fe_mul<F>(q.x, Z1Z1);
  const Fe<F> S2 = fe_mul<F>(fe_mul<F>(q.y, p.Z), Z1Z1);
  const Fe<F> H = fe_sub<F>(U2, p.X);
  const Fe<F> Rr = fe_sub<F>(S2, p.Y);
  if (fe_is_zero(H)) {
    if (fe_is_zero(Rr))
      jac_dbl_ip<C>(p);          // P + P
    else
      p.Z = fe_zero<F>();        // P + (-P)
    return;
  }
  const Fe<F> HH = fe_sqr<F>(H);
  const Fe<F> HHH = fe_mul<F>(H, HH);
  const Fe<F> V = fe_mul<F>(p.X, HH);
  const Fe<F> X3 = fe_sub<F>(fe_sub<F>(fe_sqr<F>(Rr), HHH), fe_dbl<F>(V));
  p.Y = fe_sub<F>(fe_mul<F>(Rr, fe_sub<F>(V, X3)), fe_mul<F>(p.Y, HHH));
  p.Z = fe_mul<F>(p.Z, H);
  p.X = X3;
}
template <class C>
MP_HD Jac<C> jac_madd(const Jac<C>& p, const Aff<C>& q) {
  Jac<C> r = p;
  jac_madd_ip<C>(r, q);
  return r;
}

// p <- p + q, both Jacobian
template <class C>
MP_HD void jac_add_ip(Jac<C>& p, const Jac<C>& q) {
  typedef typename C::FqP F;
  if (fe_is_zero(q.Z)) return;
  if (fe_is_zero(p.Z)) {
    p = q;
    return;
  }
  const Fe<F> Z1Z1 = fe_sqr<F>(p.Z), Z2Z2 = fe_sqr<F>(q.Z);
  const Fe<F> U1 = fe_mul<F>(p.X, Z2Z2), U2 = fe_mul<F>(q.X, Z1Z1);
  const Fe<F> S1 = fe_mul<F>(fe_mul<F>(p.Y, q.Z), Z2Z2), S2 = fe_mul<F>(fe_mul<F>(q.Y, p.Z), Z1Z1);
  const Fe<F> H = fe_sub<F>(U2, U1);
  const Fe<F> Rr = fe_sub<F>(S2, S1);
  if (fe_is_zero(H)) {
    if (fe_is_zero(Rr))
      jac_dbl_ip<C>(p);
    else
      p.Z = fe_zero<F>();
    return;
  }
  const Fe<F> HH = fe_sqr<F>(H);
  const Fe<F> HHH = fe_mul<F>(H, HH);
  const Fe<F> V = fe_mul<F>(U1, HH);
  const Fe<F> X3 = fe_sub<F>(fe_sub<F>(fe_sqr<F>(Rr), HHH), fe_dbl<F>(V));
  p.Y = fe_sub<F>(fe_mul<F>(Rr, fe_sub<F>(V, X3)), fe_mul<F>(S1, HHH));
  p.Z = fe_mul<F>(fe_mul<F>(p.Z, q.Z), H);
  p.X = X3;
}
template <class C>
MP_HD Jac<C> jac_add(const Jac<C>& p, const Jac<C>& q) {
  Jac<C> r = p;
  jac_add_ip<C>(r, q);
  return r;
}

// ---- XYZZ accumulator (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity: ZZ = 0): mixed addition 8M+2S instead of the
// Jacobian 8M+3S, doubling 6M+4S instead of 3M+6S -- pays where mixed additions outnumber doublings 5:1 or more
// (the Straus loop of k_var_msm).  Same completeness rules as the Jacobian law above.
template <class C>
struct Xyzz {
  Fe<typename C::FqP> X, Y, ZZ, ZZZ;
};
template <class C>
MP_HD Xyzz<C> xyzz_inf() {
  Xyzz<C> p;
  p.X = fe_one<typename C::FqP>();
  p.Y = fe_one<typename C::FqP>();
  p.ZZ = fe_zero<typename C::FqP>();
  p.ZZZ = fe_zero<typename C::FqP>();
  return p;
}
template <class C>
MP_HD void xyzz_dbl_ip(Xyzz<C>& p) {
  typedef typename C::FqP F;
  if (fe_is_zero(p.ZZ)) return;
  if (fe_is_zero(p.Y)) {
    p.ZZ = fe_zero<F>();
    p.ZZZ = fe_zero<F>();
    return;
  }
  const Fe<F> U = fe_dbl<F>(p.Y);
  const Fe<F> V = fe_sqr<F>(U);
  const Fe<F> W = fe_mul<F>(U, V);
  const Fe<F> S = fe_mul<F>(p.X, V);
  const Fe<F> XX = fe_sqr<F>(p.X);
  const Fe<F> M = C::A == 1 ? fe_triple_add<F>(XX, fe_sqr<F>(p.ZZ)) : fe_add<F>(fe_dbl<F>(XX), XX);     // 3 X^2 + a ZZ^2, one carry pass
  const Fe<F> X3 = fe_sub_dbl<F>(fe_sqr<F>(M), S);                                                     // M^2 - 2 S, one carry pass
  p.Y = fe_mulsub<F>(M, fe_sub_lazy<F>(S, X3), W, p.Y);      // the difference only feeds this product: no carry pass (field.hpp LazySub)
  p.X = X3;
  p.ZZ = fe_mul<F>(V, p.ZZ);
  p.ZZZ = fe_mul<F>(W, p.ZZZ);
}
// PROBE = true (tools/madprobe only): the P + P case does not expand the doubling, so that the compiled function is the
// main path plus the tests -- what a lane executes per mixed addition
// p <- p + q or p - q (neg).  The negated y only feeds one product, so on the sparse prime it is formed without a carry pass
// (field.hpp LazySub); the rare paths that keep y normalise it.
template <class C, bool PROBE = false>
MP_HD void xyzz_madd_signed_ip(Xyzz<C>& p, const Aff<C>& q, bool neg) {
  typedef typename C::FqP F;
  if (aff_is_inf<C>(q)) return;
  if (fe_is_zero(p.ZZ)) {
    p.X = q.x;
    p.Y = neg ? fe_neg<F>(q.y) : q.y;
    p.ZZ = fe_one<F>();
    p.ZZZ = fe_one<F>();
    return;
  }
  Fe<F> qy = q.y;
  if constexpr (LazySub<F>::ON) {
    const Fe<F> ny = fe_neg_lazy<F>(q.y);
#pragma unroll
    for (int i = 0; i < 9; ++i) qy.v[i] = neg ? ny.v[i] : q.y.v[i];
  } else {
    if (neg) qy = fe_neg<F>(q.y);
  }
  const Fe<F> Pd = fe_sub_wide<F>(fe_mul<F>(q.x, p.ZZ), p.X);      // only squared, multiplied and tested for zero: no weak reduction
  const Fe<F> Rr = fe_sub_wide<F>(fe_mul<F>(qy, p.ZZZ), p.Y);
  if (fe_is_zero(Pd)) {
    if (!PROBE && fe_is_zero(Rr)) {
      xyzz_dbl_ip<C>(p);         // P + P
    } else {
      p.ZZ = fe_zero<F>();       // P + (-P)
      p.ZZZ = fe_zero<F>();
    }
    return;
  }
  const Fe<F> PP = fe_sqr<F>(Pd);
  const Fe<F> PPP = fe_mul<F>(Pd, PP);
  const Fe<F> Q = fe_mul<F>(p.X, PP);
  const Fe<F> X3 = fe_sub_sub_dbl<F>(fe_sqr<F>(Rr), PPP, Q);                                            // R^2 - PPP - 2 Q, one carry pass
  p.Y = fe_mulsub<F>(Rr, fe_sub_lazy<F>(Q, X3), p.Y, PPP);     // one reduction for the two products; the difference skips its carry pass
  p.X = X3;
  p.ZZ = fe_mul<F>(p.ZZ, PP);
  p.ZZZ = fe_mul<F>(p.ZZZ, PPP);
}
template <class C, bool PROBE = false>
MP_HD void xyzz_madd_ip(Xyzz<C>& p, const Aff<C>& q) {
  xyzz_madd_signed_ip<C, PROBE>(p, q, false);
}
// p <- p + q, both XYZZ (12M + 2S): partial sums of the bucket method (kernels_bucket.hpp)
template <class C, bool PROBE = false>
MP_HD void xyzz_add_ip(Xyzz<C>& p, const Xyzz<C>& q) {
  typedef typename C::FqP F;
  if (fe_is_zero(q.ZZ)) return;
  if (fe_is_zero(p.ZZ)) {
    p = q;
    return;
  }
  const Fe<F> U1 = fe_mul<F>(p.X, q.ZZ), U2 = fe_mul<F>(q.X, p.ZZ);
  const Fe<F> S1 = fe_mul<F>(p.Y, q.ZZZ), S2 = fe_mul<F>(q.Y, p.ZZZ);
  const Fe<F> Pd = fe_sub<F>(U2, U1);
  const Fe<F> Rr = fe_sub<F>(S2, S1);
  if (fe_is_zero(Pd)) {
    if (!PROBE && fe_is_zero(Rr)) {
      xyzz_dbl_ip<C>(p);         // P + P
    } else {
      p.ZZ = fe_zero<F>();       // P + (-P)
      p.ZZZ = fe_zero<F>();
    }
    return;
  }
  const Fe<F> PP = fe_sqr<F>(Pd);
  const Fe<F> PPP = fe_mul<F>(Pd, PP);
  const Fe<F> Q = fe_mul<F>(U1, PP);
  const Fe<F> X3 = fe_sub_sub_dbl<F>(fe_sqr<F>(Rr), PPP, Q);
  p.Y = fe_mulsub<F>(Rr, fe_sub_lazy<F>(Q, X3), S1, PPP);
  p.X = X3;
  p.ZZ = fe_mul<F>(fe_mul<F>(p.ZZ, q.ZZ), PP);
  p.ZZZ = fe_mul<F>(fe_mul<F>(p.ZZZ, q.ZZZ), PPP);
}
// the same point in Jacobian coordinates with Z = ZZ: (X ZZ, Y ZZZ, ZZ)
template <class C>
MP_HD Jac<C> xyzz_to_jac(const Xyzz<C>& p) {
  typedef typename C::FqP F;
  Jac<C> j;
  if (fe_is_zero(p.ZZ)) return jac_inf<C>();
  j.X = fe_mul<F>(p.X, p.ZZ);
  j.Y = fe_mul<F>(p.Y, p.ZZZ);
  j.Z = p.ZZ;
  return j;
}

// affine from Jacobian given zinv = 1/Z
template <class C>
MP_HD Aff<C> jac_to_aff_with_zinv(const Jac<C>& j, const Fe<typename C::FqP>& zinv) {
  typedef typename C::FqP F;
  Fe<F> zi2 = fe_sqr<F>(zinv);
  Aff<C> a;
  a.x = fe_mul<F>(j.X, zi2);
  a.y = fe_mul<F>(fe_mul<F>(j.Y, zi2), zinv);
  return a;
}

}  // namespace mp
