// Prime-field arithmetic for the gfx950 vector ALU.  Two representations, chosen per modulus at compile time (curve_params.hpp,
// `P::L29`); everything else in the engine goes through the fe_* functions below and the packed memory format, never through the limbs.
//
//  (1) NW x 32-bit limbs, Montgomery R = 2^(32 NW), always fully reduced.  Product = product scanning (Comba) with a 96-bit column
//      accumulator: one v_mad_u64_u32 (32x32+64 -> 64, carry-out to VCC) + one v_addc_co_u32 per limb product (inline asm: hipcc
//      cannot express the carry-out of the mad from C).  Used for the scalar fields (`Fr::rand` defines their Montgomery form w.r.t.
//      2^256; moving them to (2) was measured at +-0: they are 1.4 % of a step).
//
//  (2) NL29 x 29-bit limbs (9; 14 for BLS12-377 Fq), Montgomery R = 2^(29 NL29), LAZILY reduced: limbs < 2^29, value in [0, 4p)
//      ([0, 2p) for dense primes).  A column sum of limb products fits a 64-bit accumulator, so the whole product is a chain of plain
//      `acc = a*b + acc` multiply-adds with NO carry handling; additions / subtractions fold a weak reduction into their single signed
//      carry pass.  Three flavours of the modulus: sparse (STARK 2^251 + 17*2^192 + 1: 2 reduction multiplies per column), signed
//      sparse (secp256k1 2^256 - 2^32 - 977, `PM29`), dense (bn254, BLS12-377, `DENSE29`: full Montgomery reduction, quotient estimate
//      by one multiply).  Plain C, identical on host and device.  All base fields use it: every VALU instruction costs a 4-cycle issue
//      slot on gfx950 whatever it is, so the form with the fewest instructions wins, and (1) pays one v_addc per mad.
//      Checked against a schoolbook big-integer reference: tests/cpp/field_check.cpp.
//
// Replaces ark-ff 0.3 `Fp256` (4x64 limbs) used by every reference call on the hot path
// [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:7-8].
// Memory format of a field element: 8 little-endian 32-bit words holding the CANONICAL (< p) Montgomery residue.
#pragma once
#include <cstdint>

#include "curve_params.hpp"
#include "rt.hpp"

namespace mp {

template <class P>
struct Fe {
  uint32_t v[P::L29 ? P::NL29 : P::NW];   // 29-bit form: NL29 = 9 limbs (256-bit fields) or 14 (BLS12-377 Fq); else NW = 8 packed words
};

static constexpr uint32_t M29 = (1u << 29) - 1;

// =====================================================================================================
// helpers of representation (1)
// =====================================================================================================
// r = a - MOD if a >= MOD (a given with an extra top carry word `hi`), branch-free
template <class P>
MP_HD void fe_cond_sub(uint32_t* r, const uint32_t* a, uint32_t hi) {
  constexpr int N = P::NW;
  uint32_t d[N];
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    uint64_t t = (uint64_t)a[i] - P::MOD[i] - br;
    d[i] = (uint32_t)t;
    br = (t >> 32) & 1;
  }
  bool ge = (br == 0) || (hi != 0);
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = ge ? d[i] : a[i];
}

// Montgomery product on 8x32 limbs, CIOS in plain C (host / development emulator form)
template <class P>
MP_HD void mul32_cios(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  constexpr int N = P::NW;
  uint32_t t[N + 1];
#pragma unroll
  for (int i = 0; i < N + 1; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    uint64_t c = 0;
    const uint32_t bi = b[i];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      c += (uint64_t)a[j] * bi + t[j];
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    c += t[N];
    t[N] = (uint32_t)c;
    uint32_t t9 = (uint32_t)(c >> 32);
    const uint32_t m = t[0] * P::INV;
    c = (uint64_t)m * P::MOD[0] + t[0];
    c >>= 32;
#pragma unroll
    for (int j = 1; j < N; ++j) {
      c += (uint64_t)m * P::MOD[j] + t[j];
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += t[N];
    t[N - 1] = (uint32_t)c;
    t[N] = t9 + (uint32_t)(c >> 32);
  }
  fe_cond_sub<P>(r, t, t[N]);
}

#if defined(__HIP_DEVICE_COMPILE__)
MP_HD void fe_mac96(uint64_t& acc, uint32_t& acc2, uint32_t x, uint32_t y) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(acc2) : "v"(x), "v"(y) : "vcc");
}
// gfx950 form of the 8x32 product (see header comment)
template <class P>
MP_HD void mul32(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  constexpr int N = P::NW;
  uint64_t acc = 0;
  uint32_t acc2 = 0;
  uint32_t t[N], m[N];
#pragma unroll
  for (int k = 0; k < 2 * N - 1; ++k) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int j = k - i;
      if (j < 0 || j > N - 1) continue;
      fe_mac96(acc, acc2, a[i], b[j]);
    }
    if (k < N) {
#pragma unroll
      for (int i = 0; i < k; ++i) {
        if (P::MOD[k - i] == 0) continue;
        fe_mac96(acc, acc2, m[i], P::MOD[k - i]);
      }
      if (P::INV == 0xFFFFFFFFu && P::MOD[0] == 1u) {
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
      for (int i = k - (N - 1); i < N; ++i) {
        if (P::MOD[k - i] == 0) continue;
        fe_mac96(acc, acc2, m[i], P::MOD[k - i]);
      }
      t[k - N] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)acc2 << 32);
      acc2 = 0;
    }
  }
  t[N - 1] = (uint32_t)acc;
  fe_cond_sub<P>(r, t, (uint32_t)(acc >> 32));
}
#else
template <class P>
MP_HD void mul32(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  mul32_cios<P>(r, a, b);
}
#endif

// =====================================================================================================
// helpers of representation (2)
// =====================================================================================================
// one signed carry pass: s (each |s_i| < 2^31) -> limbs < 2^29, the top limb keeps the rest (must be >= 0)
template <int NL>
MP_HD void carry29(int32_t* s, uint32_t* out) {
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) {
    const int32_t c = s[i] >> 29;
    out[i] = (uint32_t)s[i] & M29;
    s[i + 1] += c;
  }
  out[NL - 1] = (uint32_t)s[NL - 1];
}
// weak reduction folded into the carry pass: subtract max(q - 2, 0) * p with q = s_8 >> TOP29 (an estimate of
// floor(value / 2^(232+TOP29)) that is off by at most one either way before the carries are propagated):
// any value in [0, 8p) comes out in [0, 4p).
template <class P>
MP_HD void reduce_carry29(int32_t* s, uint32_t* out) {
  constexpr int NL = P::NL29;
  if constexpr (P::DENSE29) {
    // A prime without structure (bn254, BLS12-377): values are kept in [0, 2p), so sums and differences come here in [0, 4p).
    // t = the top limbs before the carries, aligned so that p / 2^QBIT is ~2^21 (bn254: s_8 alone; BLS12-377: s_13 2^21 + s_12 / 2^8),
    // is within (-1.01, +3.01) of v / 2^QBIT (every lower s_i is in (-2^29, 3 * 2^29)), so with the padding of 4 the estimate
    // q = floor((t + 4) QREC / 2^32), QREC = floor(2^(32 + QBIT) / p) + 1, satisfies  floor(v/p) <= q <= floor(v/p) + 1
    // (p / 2^QBIT > 10^6 absorbs both the padding and the excess of QREC).  Subtracting max(q - 1, 0) p leaves [p, 2p) if the
    // estimate was exact and [0, p) if it was one high: always [0, 2p), never negative.  k p_i <= 3 * 2^29 keeps s_i - k p_i
    // above -2^31.
    int32_t top = s[NL - 1] << P::QHI;
    if constexpr (P::QLO < 29) top += s[NL - 2] >> P::QLO;
    const uint32_t t = (uint32_t)(top + 4);
    const uint32_t q = (uint32_t)(((uint64_t)t * P::QREC) >> 32);
    const uint32_t k = q > 0 ? q - 1 : 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) s[i] -= (int32_t)(k * P::MOD29[i]);
    carry29<NL>(s, out);
  } else {
    int32_t q = (s[NL - 1] >> P::TOP29) - 2;
    q = q < 0 ? 0 : q;
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (P::SMOD29[i] != 0) s[i] -= q * P::SMOD29[i];
    carry29<NL>(s, out);
  }
}
// bring a lazily reduced value (< 8p, normalised limbs) to the canonical residue in [0, p)
template <class P>
MP_HD void canonical29(const uint32_t* a, uint32_t* out) {
  constexpr int NL = P::NL29;
  if constexpr (P::DENSE29) {       // a in [0, 2p) with normalised limbs: one exact conditional subtraction of p
    int32_t d[NL];
    uint32_t t[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] = (int32_t)a[i] - (int32_t)P::MOD29[i];
    carry29<NL>(d, t);
    const bool neg = (int32_t)t[NL - 1] < 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) out[i] = neg ? a[i] : t[i];
    return;
  }
  int32_t s[NL];
  const int32_t q = (int32_t)(a[NL - 1] >> (P::DENSE29 ? 0 : P::TOP29));
#pragma unroll
  for (int i = 0; i < NL; ++i) s[i] = (int32_t)a[i] - (P::SMOD29[i] != 0 ? q * P::SMOD29[i] : 0);
  uint32_t t[NL];
  carry29<NL>(s, t);
  // now in (-p, p) (pseudo-Mersenne p = 2^256 - c: in [0, p + 8c)): add p back if negative
  const int32_t neg = (int32_t)t[NL - 1] < 0 ? 1 : 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) s[i] = (int32_t)t[i] + (P::SMOD29[i] != 0 ? neg * P::SMOD29[i] : 0);
  carry29<NL>(s, out);
  if constexpr (P::PM29) {
    // 2^256 > p: the remainder below 2^256 may still be >= p.  t >= p  <=>  t + c >= 2^256
#pragma unroll
    for (int i = 0; i < 9; ++i) s[i] = (int32_t)out[i];
    s[0] += (int32_t)P::G0;
    s[1] += (int32_t)P::G1;
    uint32_t u[9];
    carry29<9>(s, u);
    const bool ge = (u[8] >> 24) != 0;
    u[8] &= 0x00FFFFFFu;
#pragma unroll
    for (int i = 0; i < 9; ++i) out[i] = ge ? u[i] : out[i];
  }
}
// NL limbs of 29 bits (normalised, value < 2^(32 NW)) <-> NW packed 32-bit words
template <int NW, int NL>
MP_HD void pack29(const uint32_t* l, uint32_t* w) {
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int k = (32 * j) / 29, off = (32 * j) % 29;
    uint32_t x = l[k] >> off;
    if (k + 1 < NL) x |= l[k + 1] << (29 - off);
    if (off > 26 && k + 2 < NL) x |= l[k + 2] << (58 - off);
    w[j] = x;
  }
}
template <int NW, int NL>
MP_HD void unpack29(const uint32_t* w, uint32_t* l) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int k = (29 * i) / 32, off = (29 * i) % 32;
    uint32_t x = k < NW ? w[k] >> off : 0u;
    if (off > 3 && k + 1 < NW) x |= w[k + 1] << (32 - off);
    l[i] = i < NL - 1 ? (x & M29) : x;
  }
}
// Montgomery product (R = 2^261): inputs with limbs < 2^29 and value < 4p, output in (0, 2p) with limbs < 2^29.
// Product scanning with ONE rolling accumulator and the subtractive reduction T - M p (M = T p^-1 mod R) + R p:
//   column k:  acc += sum_{i+j=k} a_i b_j - sum_{i+j=k, j>=1} m_i p_j ;  m_k = acc p^-1 mod 2^29 ;  acc = (acc - m_k p_0) >> 29
// The accumulator is signed (two's complement in a uint64_t; |acc| < 9 * 2^58 + 2^59 < 2^62).  For p = 1 (mod 2^29)
// (STARK) m_k is just the low limb and the subtraction of m_k p_0 is the shift itself, so a column costs its mads, one
// 64-bit shift and one mask: 99 mads + 38 other VALU instructions per product on gfx950 (was 90 + 94 with independent
// column sums, a negation and a 64-bit add per column).  The kernels are bound by VALU issue slots, not by the mads alone
// (DESIGN.md "instruction mix"), so the instruction count is what matters.  MP_CHAIN pins the association order
// (mad addend = running accumulator); without it LLVM rebuilds independent column sums and adds the carries separately.
#if defined(__HIP_DEVICE_COMPILE__)
#define MP_CHAIN(x) asm("" : "+v"(x))
#define MP_OPAQUE(x) asm("" : "+v"(x))      // stops LLVM from re-deriving x's shifted copies with further multiplies
MP_HD void mont_sub_step(uint64_t& acc, uint32_t m, uint32_t pj) {
  uint32_t negp = 0u - pj;
  asm("" : "+s"(negp));   // keeps it a v_mad_i64_i32 (one instruction) instead of a multiply/shift and a 64-bit subtract
  acc = (uint64_t)((int64_t)acc + (int64_t)(int32_t)m * (int64_t)(int32_t)negp);
}
#else
#define MP_CHAIN(x) ((void)0)
#define MP_OPAQUE(x) ((void)0)
MP_HD void mont_sub_step(uint64_t& acc, uint32_t m, uint32_t pj) { acc -= (uint64_t)((int64_t)m * (int64_t)(int32_t)pj); }
#endif
// acc -= m * pj for a limb pj of the (signed sparse) modulus.  Pseudo-Mersenne primes have pj = +-2^s there: a 64-bit shift
// and add / subtract instead of a multiply-add keeps the product at the 99 multiply-adds of the sparse STARK prime
// (81 limb products + m_k = acc p^-1 + m_k p_0 per column).
template <class P>
MP_HD void mont_sub_limb(uint64_t& acc, uint32_t m, int32_t pj) {
  if (P::PM29 && pj > 0 && (pj & (pj - 1)) == 0) {
    acc -= (uint64_t)m << __builtin_ctz((unsigned)pj);
  } else if (P::PM29 && pj < 0 && ((-pj) & (-pj - 1)) == 0) {
    acc += (uint64_t)m << __builtin_ctz((unsigned)(-pj));
  } else {
    mont_sub_step(acc, m, (uint32_t)pj);      // one v_mad_i64_i32 with the constant -pj (either sign)
  }
}
template <class P, bool SQR>
MP_HD void mont29(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  constexpr int NL = P::NL29;
  constexpr uint32_t PINV = (0u - P::INV29) & M29;   // +p^-1 mod 2^29
  uint32_t m[NL], a2[NL];
  if (SQR) {
#pragma unroll
    for (int i = 0; i < NL; ++i) a2[i] = a[i] << 1;
  }
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int j = k - i;
      if (j < 0 || j > NL - 1) continue;
      if (SQR) {
        if (j < i) continue;
        acc += (uint64_t)(i == j ? a[i] : a2[i]) * a[j];
      } else {
        acc += (uint64_t)a[i] * b[j];
      }
      MP_CHAIN(acc);
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int j = k - i;
      if (j < 1 || j > NL - 1 || i >= k) continue;
      if (P::SMOD29[j] != 0) {
        mont_sub_limb<P>(acc, m[i], P::SMOD29[j]);
        MP_CHAIN(acc);
      }
    }
    if (k >= NL && P::SMOD29[k - NL] != 0) acc += (uint64_t)(int64_t)P::SMOD29[k - NL];   // + R p (limbs 0..7; limb 8 below)
    if (k < NL) {
      m[k] = ((uint32_t)acc * PINV) & M29;
      if (P::PM29) MP_OPAQUE(m[k]);
      if (P::SMOD29[0] != 1) mont_sub_limb<P>(acc, m[k], P::SMOD29[0]);
    } else {
      r[k - NL] = (uint32_t)acc & M29;
    }
    acc = (uint64_t)((int64_t)acc >> 29);
  }
  r[NL - 1] = (uint32_t)acc + (uint32_t)P::SMOD29[NL - 1];
}
template <class P>
MP_HD void mul29(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  mont29<P, false>(r, a, b);
}
// (a b + c d) / R with ONE reduction: both products go into the same column sums (18 limb products per column:
// |acc| < 18 * 2^58 + 2^59 < 2^63), so the pair costs 162 + 18 mads instead of 2 x 99.  Inputs as for mul29, output in (0, 2p).
template <class P>
MP_HD void muladd29(uint32_t* r, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d) {
  constexpr int NL = P::NL29;
  static_assert(NL <= 9, "two products per column overflow the 64-bit accumulator beyond 9 limbs");
  constexpr uint32_t PINV = (0u - P::INV29) & M29;
  uint32_t m[NL];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int j = k - i;
      if (j < 0 || j > NL - 1) continue;
      acc += (uint64_t)a[i] * b[j];
      MP_CHAIN(acc);
      acc += (uint64_t)c[i] * d[j];
      MP_CHAIN(acc);
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int j = k - i;
      if (j < 1 || j > NL - 1 || i >= k) continue;
      if (P::SMOD29[j] != 0) {
        mont_sub_limb<P>(acc, m[i], P::SMOD29[j]);
        MP_CHAIN(acc);
      }
    }
    if (k >= NL && P::SMOD29[k - NL] != 0) acc += (uint64_t)(int64_t)P::SMOD29[k - NL];
    if (k < NL) {
      m[k] = ((uint32_t)acc * PINV) & M29;
      if (P::PM29) MP_OPAQUE(m[k]);
      if (P::SMOD29[0] != 1) mont_sub_limb<P>(acc, m[k], P::SMOD29[0]);
    } else {
      r[k - NL] = (uint32_t)acc & M29;
    }
    acc = (uint64_t)((int64_t)acc >> 29);
  }
  r[NL - 1] = (uint32_t)acc + (uint32_t)P::SMOD29[NL - 1];
}
template <class P>
MP_HD void sqr29(uint32_t* r, const uint32_t* a) {
  mont29<P, true>(r, a, a);
}

// =====================================================================================================
// the field API
// =====================================================================================================
template <class P>
MP_HD Fe<P> fe_zero() {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < (P::L29 ? P::NL29 : P::NW); ++i) r.v[i] = 0;
  return r;
}
template <class P>
MP_HD Fe<P> fe_one() {
  Fe<P> r;
  if constexpr (P::L29) {
#pragma unroll
    for (int i = 0; i < P::NL29; ++i) r.v[i] = P::R1_29[i];
  } else {
#pragma unroll
    for (int i = 0; i < P::NW; ++i) r.v[i] = P::R1[i];
  }
  return r;
}
// a == 0 (mod p)
template <class P>
MP_HD bool fe_is_zero(const Fe<P>& a) {
  if constexpr (P::PM29) {
    // a is one of 0, p, 2p, 3p; k p = k 2^256 - k c has the limbs (2^29 - k G0, 2^29 - 1 - k G1, 2^29 - 1, ..., k 2^24 - 1)
    const uint32_t k = (a.v[8] + 1u) >> 24;
    if (k == 0) {
      uint32_t o = 0;
#pragma unroll
      for (int i = 0; i < 9; ++i) o |= a.v[i];
      return o == 0;
    }
    uint32_t o = (a.v[0] ^ ((1u << 29) - k * P::G0)) | (a.v[1] ^ (M29 - k * P::G1)) | (a.v[8] ^ ((k << 24) - 1u));
#pragma unroll
    for (int i = 2; i < 8; ++i) o |= a.v[i] ^ M29;
    return o == 0;
  } else if constexpr (P::DENSE29) {
    // a in [0, 2p) with normalised limbs: zero mod p means a = 0 or a = p.  The low limb decides almost always.
    if (a.v[0] != 0u && a.v[0] != P::MOD29[0]) return false;
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < P::NL29; ++i) {
      z |= a.v[i];
      e |= a.v[i] ^ P::MOD29[i];
    }
    return z == 0 || e == 0;
  } else if constexpr (P::L29) {
    // lazily reduced: a is one of 0, p, 2p, ... ; the limbs of k*p are k*MOD29[i] (no carries: sparse p).
    // Fast path: where p has a zero limb, so has k*p -- almost every non-zero value is rejected by one OR chain.
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < P::NL29; ++i)
      if (P::MOD29[i] == 0) z |= a.v[i];
    if (z != 0) return false;
    const uint32_t k = a.v[P::NL29 - 1] >> P::TOP29;
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::NL29; ++i) o |= a.v[i] ^ (k * P::MOD29[i]);
    return o == 0;
  } else {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::NW; ++i) o |= a.v[i];
    return o == 0;
  }
}
template <class P>
MP_HD Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
  if constexpr (P::L29) {
    int32_t s[P::NL29];
#pragma unroll
    for (int i = 0; i < P::NL29; ++i) s[i] = (int32_t)(a.v[i] + b.v[i]);
    reduce_carry29<P>(s, r.v);
  } else {
    uint32_t s[P::NW];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < P::NW; ++i) {
      c += (uint64_t)a.v[i] + b.v[i];
      s[i] = (uint32_t)c;
      c >>= 32;
    }
    fe_cond_sub<P>(r.v, s, P::SPARE ? 0u : (uint32_t)c);
  }
  return r;
}
template <class P>
MP_HD Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
  if constexpr (P::L29) {
    // a - b + 4p in (0, 8p)
    int32_t s[P::NL29];
#pragma unroll
    for (int i = 0; i < P::NL29; ++i) s[i] = (int32_t)a.v[i] - (int32_t)b.v[i] + (P::DENSE29 ? 2 : 4) * P::SMOD29[i];   // + 4p (dense primes: values < 2p, + 2p)
    reduce_carry29<P>(s, r.v);
  } else {
    uint32_t d[P::NW];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < P::NW; ++i) {
      uint64_t t = (uint64_t)a.v[i] - b.v[i] - br;
      d[i] = (uint32_t)t;
      br = (t >> 32) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)br;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < P::NW; ++i) {
      c += (uint64_t)d[i] + (P::MOD[i] & mask);
      r.v[i] = (uint32_t)c;
      c >>= 32;
    }
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
// ---- short linear combinations of PRODUCTS with one carry pass (sparse and pseudo-Mersenne 29-bit forms) -------------------------
// Montgomery products come out in (0, 2p).  Combinations of them that stay inside [0, 8p) after adding a multiple of p need ONE
// reducing carry pass instead of one per addition / subtraction:
//   fe_sub_sub_dbl(a, u, v) = a - u - 2v    (+ 6p: (-6p, 2p) -> (0, 8p))      the x coordinate of an addition: R^2 - PPP - 2Q
//   fe_sub_dbl(a, v)        = a - 2v        (+ 4p: (-4p, 2p) -> (0, 6p))      the x coordinate of a doubling: M^2 - 2S
//   fe_triple_add(a, b)     = 3a + b        ((0, 8p))                          the slope numerator 3 X^2 + ZZ^2 (a = 1 curves)
// Every argument MUST be the direct result of fe_mul / fe_sqr / fe_mulsub.  Limb sums stay inside (-2^31, 2^31) (|.| <= 4 * 2^29 + 6 p_i).
// The dense form keeps values below 2p and reduces [0, 4p) only: it composes the plain functions.
template <class P>
MP_HD Fe<P> fe_sub_sub_dbl(const Fe<P>& a, const Fe<P>& u, const Fe<P>& v) {
  if constexpr (P::L29 && !P::DENSE29) {
    Fe<P> r;
    int32_t s[P::NL29];
#pragma unroll
    for (int i = 0; i < P::NL29; ++i) s[i] = (int32_t)a.v[i] - (int32_t)u.v[i] - 2 * (int32_t)v.v[i] + 6 * P::SMOD29[i];
    reduce_carry29<P>(s, r.v);
    return r;
  } else {
    return fe_sub<P>(fe_sub<P>(a, u), fe_dbl<P>(v));
  }
}
template <class P>
MP_HD Fe<P> fe_sub_dbl(const Fe<P>& a, const Fe<P>& v) {
  if constexpr (P::L29 && !P::DENSE29) {
    Fe<P> r;
    int32_t s[P::NL29];
#pragma unroll
    for (int i = 0; i < P::NL29; ++i) s[i] = (int32_t)a.v[i] - 2 * (int32_t)v.v[i] + 4 * P::SMOD29[i];
    reduce_carry29<P>(s, r.v);
    return r;
  } else {
    return fe_sub<P>(a, fe_dbl<P>(v));
  }
}
template <class P>
MP_HD Fe<P> fe_triple_add(const Fe<P>& a, const Fe<P>& b) {
  if constexpr (P::L29 && !P::DENSE29) {
    Fe<P> r;
    int32_t s[P::NL29];
#pragma unroll
    for (int i = 0; i < P::NL29; ++i) s[i] = 3 * (int32_t)a.v[i] + (int32_t)b.v[i];
    reduce_carry29<P>(s, r.v);
    return r;
  } else {
    return fe_add<P>(fe_add<P>(fe_dbl<P>(a), a), b);
  }
}
// ---- carry-free differences for operands that ONLY feed a product (sparse 29-bit form) ------------------------------------------
// On the sparse prime (STARK) a subtraction whose result is used as ONE multiplicand and nothing else can skip its carry pass:
//   r_i = a_i - b_i + K_i     with K = 8p written so that every limb but the top one is >= 2^29 - 1
//                             (K_0 = 8 p_0 + 2^29, K_i = 8 p_i + 2^29 - 1 for 0 < i < 8, K_8 = 8 p_8 - 1 = 2^22 - 1: the same integer 8p).
// For normalised a, b < 4p (limbs < 2^29, top limbs <= 2^21) every r_i is >= 0 and < 2^30.05, the value is a - b + 8p < 12p.  A product
// with a NORMALISED second operand then has column sums < 9 * 2^29 * 2^30.05 = 2^62.2, and the fused pair a b - c d with b = such a
// difference and c = such a negation (limbs <= K_i < 2^29.1) < 9 * 2^29 * (2^30.05 + 2^29.1) = 2^62.82 < 2^63: the signed 64-bit
// accumulator of mont29 / muladd29 holds (the subtracted reduction terms are < 2^52 on this prime).  Results: (12 * 4 + 8 * 4) p^2 / R
// + p < 1.08 p.  NOT a field element in the lazy invariant: never store it, never add or subtract it, never test it for zero.
// Saves 23 (difference) / 32 (negation) of the 41 instructions of a normalising subtraction.  Other primes: the plain functions
// (pseudo-Mersenne and dense limbs of 8p are ~2^29 each: the fused pair would reach 2^63.1).
template <class P>
struct LazySub {
  static constexpr bool ON = P::L29 && !P::PM29 && !P::DENSE29 && P::NL29 == 9;
  static constexpr int32_t k(int i) { return 8 * P::SMOD29[i] + (i == 0 ? (1 << 29) : i == 8 ? -1 : (1 << 29) - 1); }
};
template <class P>
MP_HD Fe<P> fe_sub_lazy(const Fe<P>& a, const Fe<P>& b) {
  if constexpr (LazySub<P>::ON) {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] - b.v[i] + (uint32_t)LazySub<P>::k(i);
    return r;
  } else {
    return fe_sub<P>(a, b);
  }
}
// a - b + 4p with the carry pass but WITHOUT the weak reduction: for a < 2p (a product) and b < 4p the value is in (0, 6p), limbs
// normalised.  Good as an operand of products / squares (36 p^2 / R < 0.04 p on this prime) and for zero tests (k p is recognised for
// any small k); not to be added or subtracted again.  Sparse prime only; elsewhere the plain subtraction.
template <class P>
MP_HD Fe<P> fe_sub_wide(const Fe<P>& a, const Fe<P>& b) {
  if constexpr (LazySub<P>::ON) {
    Fe<P> r;
    int32_t s[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) s[i] = (int32_t)a.v[i] - (int32_t)b.v[i] + 4 * P::SMOD29[i];
    carry29<9>(s, r.v);
    return r;
  } else {
    return fe_sub<P>(a, b);
  }
}
template <class P>
MP_HD Fe<P> fe_neg_lazy(const Fe<P>& a) {
  if constexpr (LazySub<P>::ON) {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = (uint32_t)LazySub<P>::k(i) - a.v[i];
    return r;
  } else {
    return fe_neg<P>(a);
  }
}
// a / 2 (8x32 representation only: the scalar fields).  Works on the Montgomery residue: (a + (a odd ? p : 0)) >> 1
template <class P>
MP_HD Fe<P> fe_half(const Fe<P>& a) {
  static_assert(!P::L29 && P::NW == 8, "fe_half is only provided for the 8x32 representation");
  const uint32_t mask = (uint32_t)0 - (a.v[0] & 1u);
  uint32_t s[9];
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)a.v[i] + (P::MOD[i] & mask);
    s[i] = (uint32_t)c;
    c >>= 32;
  }
  s[8] = (uint32_t)c;
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (s[i] >> 1) | (s[i + 1] << 31);
  return r;
}
template <class P>
MP_HD bool fe_eq(const Fe<P>& a, const Fe<P>& b) {
  if constexpr (P::L29) {
    return fe_is_zero<P>(fe_sub<P>(a, b));
  } else {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::NW; ++i) o |= a.v[i] ^ b.v[i];
    return o == 0;
  }
}
// The 12-limb product (BLS12-377 Fq: 288 multiply-adds) is kept out of line: inlined into every call site of the group law
// it multiplies the compile time of that curve's translation unit by ~5 for a curve that is not the throughput target.
template <class P>
MP_HD_NOINLINE void mul32_call(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  mul32<P>(r, a, b);
}
// (the 14-limb product of the 29-bit form -- BLS12-377 Fq, 196 + 168 multiply-adds -- is inlined like the 9-limb ones: with the
// loops fully unrolled its translation unit compiles in 90 s, and kept out of line it cost the group law its registers: 812 bytes of
// stack per lane in every kernel against none inlined)
template <class P>
MP_HD Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
  if constexpr (P::L29)
    mul29<P>(r.v, a.v, b.v);
  else if constexpr (P::NW > 8)
    mul32_call<P>(r.v, a.v, b.v);
  else
    mul32<P>(r.v, a.v, b.v);
  return r;
}
// a b - c d (one reduction for the pair in the 29-bit form)
template <class P>
MP_HD Fe<P> fe_mulsub(const Fe<P>& a, const Fe<P>& b, const Fe<P>& c, const Fe<P>& d) {
  if constexpr (P::L29 && P::NL29 <= 9) {
    Fe<P> r;
    const Fe<P> nc = fe_neg_lazy<P>(c);          // c, a and d normalised; b normalised or a fe_sub_lazy difference (bounds above)
    muladd29<P>(r.v, a.v, b.v, nc.v, d.v);
    return r;
  } else {
    return fe_sub<P>(fe_mul<P>(a, b), fe_mul<P>(c, d));
  }
}
template <class P>
MP_HD Fe<P> fe_sqr(const Fe<P>& a) {
  Fe<P> r;
  if constexpr (P::L29)
    sqr29<P>(r.v, a.v);
  else if constexpr (P::NW > 8)
    mul32_call<P>(r.v, a.v, a.v);
  else
    mul32<P>(r.v, a.v, a.v);
  return r;
}

// ---- memory format: P::NW packed words, canonical Montgomery residue ---------------------------------------
template <class P>
MP_HD Fe<P> fe_unpack(const uint32_t* w) {
  Fe<P> r;
  if constexpr (P::L29) {
    unpack29<P::NW, P::NL29>(w, r.v);
  } else {
#pragma unroll
    for (int i = 0; i < P::NW; ++i) r.v[i] = w[i];
  }
  return r;
}
template <class P>
MP_HD void fe_pack(const Fe<P>& a, uint32_t* w) {
  if constexpr (P::L29) {
    uint32_t c[P::NL29];
    canonical29<P>(a.v, c);
    pack29<P::NW, P::NL29>(c, w);
  } else {
#pragma unroll
    for (int i = 0; i < P::NW; ++i) w[i] = a.v[i];
  }
}

// ---- canonical integer (8 x u32, little-endian) <-> Montgomery form ---------------------------------------------
template <class P>
MP_HD Fe<P> fe_from_canonical(const uint32_t* a) {
  Fe<P> t = fe_unpack<P>(a), r2;
  if constexpr (P::L29) {
#pragma unroll
    for (int i = 0; i < P::NL29; ++i) r2.v[i] = P::R2_29[i];
  } else {
#pragma unroll
    for (int i = 0; i < P::NW; ++i) r2.v[i] = P::R2[i];
  }
  return fe_mul<P>(t, r2);
}
template <class P>
MP_HD void fe_to_canonical(const Fe<P>& a, uint32_t* out) {
  Fe<P> one = fe_zero<P>();
  one.v[0] = 1u;
  fe_pack<P>(fe_mul<P>(a, one), out);
}
template <class P>
MP_HD Fe<P> fe_from_u32(uint32_t x) {
  uint32_t a[P::NW] = {x};
  return fe_from_canonical<P>(a);
}
// is the canonical integer a < MOD ?
template <class P>
MP_HD bool fe_canonical_in_range(const uint32_t* a) {
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < P::NW; ++i) {
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
MP_HD_NOINLINE Fe<P> fe_inv_fermat(const Fe<P>& a) {
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

// ---- inversion by division steps --------------------------------------------------------------------------------------------
// Bernstein-Yang "safegcd" [PAPER: Bernstein, Yang, "Fast constant-time gcd computation and modular inversion", CHES 2019], in the
// half-delta form with batched 2x2 transition matrices that libsecp256k1's modinv32 made standard, restated for NL signed limbs of
// 29 bits (9 limbs up to 256 bits, 14 for the 377-bit base field of BLS12-377): batches of 29 division steps on the low words, then
// the matrix applied to (f, g) (4 NL multiply-adds) and to (d, e) modulo p (6 NL).  256 bits: 21 batches (609 >= the 590 steps that
// suffice there), ~1 900 multiply-adds and ~14 000 single-cycle operations instead of the ~20 600 dependent multiply-adds of the
// Fermat ladder above -- a 2.5-4x shorter chain on the lane that every batch of points waits for (k_normalize, k_table).
// Constant number of steps, no data-dependent branches: the lanes of a wave stay together.  Beyond 256 bits the step count is the
// (larger) bound of the original delta = 1 analysis, (49 bits + 80) / 17, and the result is CHECKED (g = 0 at the end, which
// makes d the inverse whatever the count was): should it ever fail, the Fermat ladder answers instead.
MP_HD int32_t divsteps29(int32_t zeta, uint32_t f0, uint32_t g0, int32_t t[4]) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 1
  for (int i = 0; i < 29; ++i) {
    uint32_t m1 = (uint32_t)(zeta >> 31);          // zeta < 0
    const uint32_t m2 = 0u - (g & 1u);              // g odd
    const uint32_t x = (f ^ m1) - m1, y = (u ^ m1) - m1, z = (v ^ m1) - m1;   // (f, u, v), negated if zeta < 0
    g += x & m2; q += y & m2; r += z & m2;
    m1 &= m2;
    zeta = (zeta ^ (int32_t)m1) - 1;                // -zeta - 2 or zeta - 1
    f += g & m1; u += q & m1; v += r & m1;
    g >>= 1; u <<= 1; v <<= 1;
  }
  t[0] = (int32_t)u; t[1] = (int32_t)v; t[2] = (int32_t)q; t[3] = (int32_t)r;
  return zeta;
}
// (f, g) <- t (f, g) / 2^29 (exact)
template <int NL>
MP_HD void divsteps_update_fg(int32_t* f, int32_t* g, const int32_t t[4]) {
  const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
  int64_t cf = u * f[0] + v * g[0], cg = q * f[0] + r * g[0];
  cf >>= 29; cg >>= 29;
#pragma unroll
  for (int i = 1; i < NL; ++i) {
    cf += u * f[i] + v * g[i];
    cg += q * f[i] + r * g[i];
    f[i - 1] = (int32_t)cf & (int32_t)M29; cf >>= 29;
    g[i - 1] = (int32_t)cg & (int32_t)M29; cg >>= 29;
  }
  f[NL - 1] = (int32_t)cf;
  g[NL - 1] = (int32_t)cg;
}
// (d, e) <- t (d, e) / 2^29 mod p, both kept in (-2p, p); pinv = p^-1 mod 2^29
template <int NL>
MP_HD void divsteps_update_de(int32_t* d, int32_t* e, const int32_t t[4], const int32_t* p, uint32_t pinv) {
  const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
  const int32_t sd = d[NL - 1] >> 31, se = e[NL - 1] >> 31;
  int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
  int64_t cd = (int64_t)u * d[0] + (int64_t)v * e[0], ce = (int64_t)q * d[0] + (int64_t)r * e[0];
  md -= (int32_t)((pinv * (uint32_t)cd + (uint32_t)md) & M29);
  me -= (int32_t)((pinv * (uint32_t)ce + (uint32_t)me) & M29);
  cd += (int64_t)p[0] * md;
  ce += (int64_t)p[0] * me;
  cd >>= 29; ce >>= 29;
#pragma unroll
  for (int i = 1; i < NL; ++i) {
    cd += (int64_t)u * d[i] + (int64_t)v * e[i] + (int64_t)p[i] * md;
    ce += (int64_t)q * d[i] + (int64_t)r * e[i] + (int64_t)p[i] * me;
    d[i - 1] = (int32_t)cd & (int32_t)M29; cd >>= 29;
    e[i - 1] = (int32_t)ce & (int32_t)M29; ce >>= 29;
  }
  d[NL - 1] = (int32_t)cd;
  e[NL - 1] = (int32_t)ce;
}
// NW packed 32-bit words <-> NL limbs of 29 bits (the top limb takes what is left)
template <int NW, int NL>
MP_HD void limbs29_from_words(const uint32_t* w, int32_t* l) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int k = (29 * i) / 32, off = (29 * i) % 32;
    uint32_t x = k < NW ? w[k] >> off : 0u;
    if (off > 3 && k + 1 < NW) x |= w[k + 1] << (32 - off);
    l[i] = (int32_t)(x & M29);
  }
}
template <int NW, int NL>
MP_HD void words_from_limbs29(const int32_t* l, uint32_t* w) {      // limbs normalised, value < 2^(32 NW)
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int k = (32 * j) / 29, off = (32 * j) % 29;
    uint32_t x = (uint32_t)l[k] >> off;
    if (k + 1 < NL) x |= (uint32_t)l[k + 1] << (29 - off);
    if (off > 0 && 29 - off + 29 < 32 && k + 2 < NL) x |= (uint32_t)l[k + 2] << (58 - off);
    w[j] = x;
  }
}
// inlined into its kernels: a function of its own is compiled without the caller's register budget (it came out at 248 VGPRs
// and took k_table and k_normalize down to 2 waves per SIMD)
template <class P>
MP_HD Fe<P> fe_inv_divsteps(const Fe<P>& a) {
  constexpr int NW = P::NW, NL = (P::BITS + 2 + 28) / 29;      // room for values in (-2p, p)
  constexpr int BATCHES = P::BITS <= 256 ? 21 : ((49 * P::BITS + 80) / 17 + 28) / 29;
  uint32_t xw[NW];
  if constexpr (P::L29) {
    uint32_t c[P::NL29];
    canonical29<P>(a.v, c);                           // the residue a R mod p as an integer in [0, p)
    pack29<NW, P::NL29>(c, xw);
  } else {
#pragma unroll
    for (int i = 0; i < NW; ++i) xw[i] = a.v[i];      // canonical Montgomery residue already
  }
  int32_t p[NL], f[NL], g[NL], d[NL], e[NL];
  limbs29_from_words<NW, NL>(P::MOD, p);
  limbs29_from_words<NW, NL>(xw, g);
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    f[i] = p[i]; d[i] = 0; e[i] = 0;
  }
  e[0] = 1;
  uint32_t pinv = (uint32_t)p[0];                     // Newton: p^-1 mod 2^32 from p (odd) in 4 steps, then mod 2^29
#pragma unroll
  for (int i = 0; i < 4; ++i) pinv *= 2u - (uint32_t)p[0] * pinv;
  pinv &= M29;
  int32_t zeta = -1;
#pragma unroll 1
  for (int it = 0; it < BATCHES; ++it) {
    int32_t t[4];
    zeta = divsteps29(zeta, (uint32_t)f[0], (uint32_t)g[0], t);
    divsteps_update_de<NL>(d, e, t, p, pinv);
    divsteps_update_fg<NL>(f, g, t);
  }
  if constexpr (P::BITS > 256) {
    // beyond 256 bits the step count is not the published bound of this variant: check the outcome (g = 0 makes d the inverse
    // whatever the count was).  Up to 256 bits the 590-step bound is proven and the kernels stay free of an out-of-line call.
    int32_t gnz = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) gnz |= g[i];
    if (gnz != 0) {                                   // never observed
#ifdef MP_DIVSTEPS_COUNT_FALLBACKS                     // (tests/cpp/inv_check.cpp counts how often this branch is taken: 0)
      ++MP_DIVSTEPS_COUNT_FALLBACKS;
#endif
      return fe_inv_fermat<P>(a);
    }
  }
  // g = 0 and f = +-gcd = +-1 (or f = +-p for a = 0, with d = 0): the inverse is sign(f) d, brought into [0, p)
  const int32_t neg = f[NL - 1] >> 31;
  int32_t add = d[NL - 1] >> 31;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    d[i] += p[i] & add;
    d[i] = (d[i] ^ neg) - neg;
  }
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) {
    d[i + 1] += d[i] >> 29;
    d[i] &= (int32_t)M29;
  }
  add = d[NL - 1] >> 31;
#pragma unroll
  for (int i = 0; i < NL; ++i) d[i] += p[i] & add;
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) {
    d[i + 1] += d[i] >> 29;
    d[i] &= (int32_t)M29;
  }
  // d = (a R)^-1 as an integer; a^-1 R = d R^2
  if constexpr (P::L29) {
    Fe<P> y, r2;
    static_assert(!P::L29 || NL == P::NL29, "the division steps work on the field's own limb count");
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      y.v[i] = (uint32_t)d[i];
      r2.v[i] = P::R2_29[i];
    }
    return fe_mul<P>(y, fe_mul<P>(r2, r2));           // montmul(d, R^3), R^3 = montmul(R^2, R^2)
  } else {
    Fe<P> y, r2;
    words_from_limbs29<NW, NL>(d, y.v);
#pragma unroll
    for (int i = 0; i < NW; ++i) r2.v[i] = P::R2[i];
    return fe_mul<P>(fe_mul<P>(y, r2), r2);           // montmul(montmul(d, R^2), R^2) = d R^2
  }
}
template <class P>
MP_HD Fe<P> fe_inv(const Fe<P>& a) {
  return fe_inv_divsteps<P>(a);
}

}  // namespace mp
