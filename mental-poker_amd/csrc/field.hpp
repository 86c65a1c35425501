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

// Montgomery product, host / reference form: CIOS in plain C (also what the development emulator runs).
template <class P>
MP_HD Fe<P> fe_mul_cios(const Fe<P>& a, const Fe<P>& b) {
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

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 form: product scanning (Comba).  Each column sum lives in a 96-bit accumulator {acc2 : acc}; one limb
// product is ONE v_mad_u64_u32 (32x32+64 with carry-out to VCC) plus one v_addc_co_u32 that banks the carry in
// the third word -- hipcc cannot express the carry-out of the mad from C (the CIOS loop above compiles to ~580
// VALU instructions per product for the same 72 multiplies; this form to ~290).  Montgomery reduction is
// interleaved per column; reduction products with zero modulus limbs vanish at compile time.
// Measured on MI355X (tools/microbench/fmul.hip, STARK Fq): 101 -> 171 G products/s.
MP_HD void fe_mac96(uint64_t& acc, uint32_t& acc2, uint32_t x, uint32_t y) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(acc2) : "v"(x), "v"(y) : "vcc");
}
template <class P>
MP_HD Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
  uint64_t acc = 0;
  uint32_t acc2 = 0;
  uint32_t t[8], m[8];
#pragma unroll
  for (int k = 0; k < 15; ++k) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = k - i;
      if (j < 0 || j > 7) continue;
      fe_mac96(acc, acc2, a.v[i], b.v[j]);
    }
    if (k < 8) {
#pragma unroll
      for (int i = 0; i < k; ++i) {
        if (P::MOD[k - i] == 0) continue;
        fe_mac96(acc, acc2, m[i], P::MOD[k - i]);
      }
      if (P::INV == 0xFFFFFFFFu && P::MOD[0] == 1u) {
        // m = -acc_lo; adding m * 1 clears the low word and carries iff it was non-zero
        m[k] = 0u - (uint32_t)acc;
        const uint32_t c = (uint32_t)acc != 0;
        acc = (uint64_t)(uint32_t)(acc >> 32) + c + ((uint64_t)acc2 << 32);
      } else {
        m[k] = (uint32_t)acc * P::INV;
        fe_mac96(acc, acc2, m[k], P::MOD[0]);
        acc = (acc >> 32) | ((uint64_t)acc2 << 32);
      }
      acc2 = 0;
    } else {
#pragma unroll
      for (int i = k - 7; i < 8; ++i) {
        if (P::MOD[k - i] == 0) continue;
        fe_mac96(acc, acc2, m[i], P::MOD[k - i]);
      }
      t[k - 8] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)acc2 << 32);
      acc2 = 0;
    }
  }
  t[7] = (uint32_t)acc;
  Fe<P> r;
  fe_cond_sub<P>(r.v, t, (uint32_t)(acc >> 32));
  return r;
}
#else
template <class P>
MP_HD Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
  return fe_mul_cios<P>(a, b);
}
#endif
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

// a^(p-2) (inverse of 0 is 0).  The exponents p-2 of the supported fields contain long runs of one bits (STARK:
// 192 of them, secp256k1: 223), so the chain consumes up to 5 one-bits at a time with the precomputed powers
// a^(2^L - 1), L = 1..5: ~256 squarings + ~45..75 products instead of one product per one-bit.
// Not inlined: it is called once per batch of points.
template <class P>
MP_HD_NOINLINE Fe<P> fe_inv(const Fe<P>& a) {
  Fe<P> run[5];                       // run[L-1] = a^(2^L - 1)
  run[0] = a;
  for (int l = 1; l < 5; ++l) run[l] = fe_mul<P>(fe_sqr<P>(run[l - 1]), a);
  Fe<P> acc = fe_one<P>();
  int i = P::BITS - 1;
  while (i >= 0) {
    if (!((P::PM2[i >> 5] >> (i & 31)) & 1u)) {
      acc = fe_sqr<P>(acc);
      --i;
      continue;
    }
    int len = 1;
    while (len < 5 && i - len >= 0 && ((P::PM2[(i - len) >> 5] >> ((i - len) & 31)) & 1u)) ++len;
    for (int q = 0; q < len; ++q) acc = fe_sqr<P>(acc);
    acc = fe_mul<P>(acc, run[len - 1]);
    i -= len;
  }
  return acc;
}

}  // namespace mp
