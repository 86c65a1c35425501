// 256-bit prime-field arithmetic on 8 x 32-bit limbs, Montgomery form (R = 2^256), written for the
// gfx950 vector ALU: every limb product is one v_mad_u64_u32 (32x32+64 -> 64), all loops are fully
// unrolled over compile-time moduli so that zero limbs of a sparse modulus (STARK p = 2^251 + 17*2^192
// + 1, secp256k1 p) cost nothing.  Replaces ark-ff 0.3 `Fp256` (4x64 limbs) used by every reference
// call on the hot path [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:7-8].
// Values are always fully reduced (in [0, p)), so equality is limb equality.
#pragma once
#include <cstdint>

#include "curve_params.hpp"
#include "rt.hpp"

namespace mp {

template <class P>
struct Fe {
  uint32_t v[8];
};

template <class P>
MP_HD Fe<P> fe_zero() {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0;
  return r;
}
template <class P>
MP_HD Fe<P> fe_one() {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = P::R1[i];
  return r;
}
template <class P>
MP_HD bool fe_is_zero(const Fe<P>& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.v[i];
  return o == 0;
}
template <class P>
MP_HD bool fe_eq(const Fe<P>& a, const Fe<P>& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
  return o == 0;
}

// r = a - MOD if a >= MOD (a given with an extra top carry word `hi`), branch-free
template <class P>
MP_HD void fe_cond_sub(uint32_t r[8], const uint32_t a[8], uint32_t hi) {
  uint32_t d[8];
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t t = (uint64_t)a[i] - P::MOD[i] - br;
    d[i] = (uint32_t)t;
    br = (t >> 32) & 1;
  }
  // subtract when no borrow, or when the carry word absorbs it
  bool ge = (br == 0) || (hi != 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = ge ? d[i] : a[i];
}

template <class P>
MP_HD Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
  uint32_t s[8];
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)a.v[i] + b.v[i];
    s[i] = (uint32_t)c;
    c >>= 32;
  }
  Fe<P> r;
  fe_cond_sub<P>(r.v, s, P::SPARE ? 0u : (uint32_t)c);
  return r;
}
template <class P>
MP_HD Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
  uint32_t d[8];
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t t = (uint64_t)a.v[i] - b.v[i] - br;
    d[i] = (uint32_t)t;
    br = (t >> 32) & 1;
  }
  // add MOD back if we borrowed
  uint32_t mask = (uint32_t)0 - (uint32_t)br;
  Fe<P> r;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)d[i] + (P::MOD[i] & mask);
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  return r;
}
template <class P>
MP_HD Fe<P> fe_neg(const Fe<P>& a) {
  return fe_sub<P>(fe_zero<P>(), a);
}
template <class P>
MP_HD Fe<P> fe_dbl(const Fe<P>& a) {
  return fe_add<P>(a, a);
}

// CIOS Montgomery product.  Per outer round: 8 mads for a*b[i], then m = t0 * INV and one mad per
// NON-ZERO modulus limb (the compiler drops `m * 0`): 3 for STARK p, 8 for a dense modulus.
template <class P>
MP_HD Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t c = 0;
    const uint32_t bi = b.v[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      c += (uint64_t)a.v[j] * bi + t[j];
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[8] = (uint32_t)c;
    uint32_t t9 = (uint32_t)(c >> 32);
    const uint32_t m = t[0] * P::INV;
    c = (uint64_t)m * P::MOD[0] + t[0];
    c >>= 32;
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      c += (uint64_t)m * P::MOD[j] + t[j];
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[7] = (uint32_t)c;
    t[8] = t9 + (uint32_t)(c >> 32);
  }
  Fe<P> r;
  fe_cond_sub<P>(r.v, t, t[8]);
  return r;
}
template <class P>
MP_HD Fe<P> fe_sqr(const Fe<P>& a) {
  return fe_mul<P>(a, a);
}

// canonical integer (8 x u32, little-endian limbs) <-> Montgomery form
template <class P>
MP_HD Fe<P> fe_from_canonical(const uint32_t a[8]) {
  Fe<P> t, r2;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    t.v[i] = a[i];
    r2.v[i] = P::R2[i];
  }
  return fe_mul<P>(t, r2);
}
template <class P>
MP_HD void fe_to_canonical(const Fe<P>& a, uint32_t out[8]) {
  Fe<P> one;
#pragma unroll
  for (int i = 0; i < 8; ++i) one.v[i] = i == 0 ? 1u : 0u;
  Fe<P> r = fe_mul<P>(a, one);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = r.v[i];
}
template <class P>
MP_HD Fe<P> fe_from_u32(uint32_t x) {
  uint32_t a[8] = {x, 0, 0, 0, 0, 0, 0, 0};
  return fe_from_canonical<P>(a);
}
// is the canonical integer a < MOD ?
template <class P>
MP_HD bool fe_canonical_in_range(const uint32_t a[8]) {
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t t = (uint64_t)a[i] - P::MOD[i] - br;
    br = (t >> 32) & 1;
  }
  return br != 0;
}

// a^(p-2) by left-to-right square-and-multiply over the compile-time exponent (inverse of 0 is 0).
// Not inlined: it is called once per batch of points, and the body is 256 squarings long.
template <class P>
MP_HD_NOINLINE Fe<P> fe_inv(const Fe<P>& a) {
  Fe<P> acc = fe_one<P>();
  for (int i = P::BITS - 1; i >= 0; --i) {
    acc = fe_sqr<P>(acc);
    if ((P::PM2[i >> 5] >> (i & 31)) & 1u) acc = fe_mul<P>(acc, a);
  }
  return acc;
}

}  // namespace mp
