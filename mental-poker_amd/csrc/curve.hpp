// Short-Weierstrass group law for the gfx950 engine: Jacobian accumulators, affine operands
// (mixed addition 7M+4S, doubling 1M+8S for a = 1 / 2M+5S-style for a = 0, full addition 11M+5S).
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

// 2P.  (Y = 0 cannot happen on a prime-order group of odd order, but is handled.)
template <class C>
MP_HD Jac<C> jac_dbl(const Jac<C>& p) {
  typedef typename C::FqP F;
  if (fe_is_zero(p.Z) || fe_is_zero(p.Y)) return jac_inf<C>();
  Fe<F> XX = fe_sqr<F>(p.X), YY = fe_sqr<F>(p.Y);
  Fe<F> S = fe_dbl<F>(fe_dbl<F>(fe_mul<F>(p.X, YY)));
  Fe<F> M = fe_add<F>(fe_dbl<F>(XX), XX);
  if (C::A == 1) M = fe_add<F>(M, fe_sqr<F>(fe_sqr<F>(p.Z)));
  Jac<C> r;
  r.X = fe_sub<F>(fe_sqr<F>(M), fe_dbl<F>(S));
  Fe<F> Y8 = fe_dbl<F>(fe_dbl<F>(fe_dbl<F>(fe_sqr<F>(YY))));
  r.Z = fe_dbl<F>(fe_mul<F>(p.Y, p.Z));
  r.Y = fe_sub<F>(fe_mul<F>(M, fe_sub<F>(S, r.X)), Y8);
  return r;
}
// the rare P + P branch inside additions.  Inlined on purpose: an out-of-line call makes every value that is
// live across it pay the call ABI (196 VGPRs / 2 waves per SIMD in k_var_msm instead of 128 / 4).
template <class C>
MP_HD Jac<C> jac_dbl_rare(const Jac<C>& p) {
  return jac_dbl<C>(p);
}

// P + Q, Q affine
template <class C>
MP_HD Jac<C> jac_madd(const Jac<C>& p, const Aff<C>& q) {
  typedef typename C::FqP F;
  if (aff_is_inf<C>(q)) return p;
  if (fe_is_zero(p.Z)) return jac_from_aff<C>(q);
  Fe<F> Z1Z1 = fe_sqr<F>(p.Z);
  Fe<F> U2 = fe_mul<F>(q.x, Z1Z1);
  Fe<F> S2 = fe_mul<F>(fe_mul<F>(q.y, p.Z), Z1Z1);
  Fe<F> H = fe_sub<F>(U2, p.X);
  Fe<F> Rr = fe_sub<F>(S2, p.Y);
  if (fe_is_zero(H)) {
    if (fe_is_zero(Rr)) return jac_dbl_rare<C>(p);
    return jac_inf<C>();
  }
  Fe<F> HH = fe_sqr<F>(H);
  Fe<F> HHH = fe_mul<F>(H, HH);
  Fe<F> V = fe_mul<F>(p.X, HH);
  Jac<C> r;
  r.X = fe_sub<F>(fe_sub<F>(fe_sqr<F>(Rr), HHH), fe_dbl<F>(V));
  r.Y = fe_sub<F>(fe_mul<F>(Rr, fe_sub<F>(V, r.X)), fe_mul<F>(p.Y, HHH));
  r.Z = fe_mul<F>(p.Z, H);
  return r;
}

// P + Q, both Jacobian
template <class C>
MP_HD Jac<C> jac_add(const Jac<C>& p, const Jac<C>& q) {
  typedef typename C::FqP F;
  if (fe_is_zero(p.Z)) return q;
  if (fe_is_zero(q.Z)) return p;
  Fe<F> Z1Z1 = fe_sqr<F>(p.Z), Z2Z2 = fe_sqr<F>(q.Z);
  Fe<F> U1 = fe_mul<F>(p.X, Z2Z2), U2 = fe_mul<F>(q.X, Z1Z1);
  Fe<F> S1 = fe_mul<F>(fe_mul<F>(p.Y, q.Z), Z2Z2), S2 = fe_mul<F>(fe_mul<F>(q.Y, p.Z), Z1Z1);
  Fe<F> H = fe_sub<F>(U2, U1);
  Fe<F> Rr = fe_sub<F>(S2, S1);
  if (fe_is_zero(H)) {
    if (fe_is_zero(Rr)) return jac_dbl_rare<C>(p);
    return jac_inf<C>();
  }
  Fe<F> HH = fe_sqr<F>(H);
  Fe<F> HHH = fe_mul<F>(H, HH);
  Fe<F> V = fe_mul<F>(U1, HH);
  Jac<C> r;
  r.X = fe_sub<F>(fe_sub<F>(fe_sqr<F>(Rr), HHH), fe_dbl<F>(V));
  r.Y = fe_sub<F>(fe_mul<F>(Rr, fe_sub<F>(V, r.X)), fe_mul<F>(S1, HHH));
  r.Z = fe_mul<F>(fe_mul<F>(p.Z, q.Z), H);
  return r;
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
