// kernels_decompress.hpp -- arkworks-0.3 compressed points -> wire v1, ON THE DEVICE (round 4).
//
// Every associated type of the trait is CanonicalSerialize + CanonicalDeserialize [REF barnett-smart-card-protocol/src/lib.rs:45-71]:
// a Rust caller holds decks as compressed points (x + two flag bits).  Until round 3 the conversion to the engine's wire format was
// host work (serialize_host.hpp: one Tonelli-Shanks per point -- ~10^4 squarings on the STARK prime, whose p - 1 = 2^192 (2^59 + 17)
// has 2-adicity 192 -- three times the cost of proving and verifying the deck).  Here one lane decompresses one point and the square
// root is a windowed Pohlig-Hellman walk through the 2-Sylow subgroup with uniform control flow:
//
//   p - 1 = 2^S q (q odd), z a non-residue, g = z^q (order 2^S), w | S, k = S / w windows
//   v = a^((q-1)/2);  x = a v  (x^2 = a t),  t = a v^2 = a^q  in <g>
//   for i = 0 .. k-1:   u = acc^(2^(S - w (i+1)))            acc = t g^-(e_0 + .. + e_{i-1} 2^(w(i-1))), so u lies in <h>, h = g^(2^(S-w))
//                       d = the index with H[d] = h^d = u     (2^w candidates, compared one by one: w = 4 on the STARK prime)
//                       acc *= g^(-d 2^(wi));                 y *= Ghalf[i][d] = g^(-d 2^(wi) / 2)
//   a is a square  <=>  e_0 even;  then sqrt(a) = x * prod_i Ghalf[i][e_i]
//
// Computed that way each window costs S - w (i+1) squarings of its own: S k - w k (k+1) / 2 = 4 512 on the STARK prime (w = 4).  But
//   u_i = acc_i^(2^(S - w(i+1))) = t^(2^(S - w(i+1))) * prod_{j<i} R[i+1-j][e_j],      R[c][d] = g^(-d 2^(S - w c)),
// and the powers T_j = t^(2^(w j)) all lie on ONE chain of S - w squarings: the lane writes them to a scratch column once
// ([j][lane]: coalesced) and window i costs i table multiplications instead -- S - w squarings + k (k-1) / 2 products (188 + 1 128 on
// the STARK prime) against ~9 000 squarings on average (18 000 at worst, with the lanes of a wave waiting for the worst) for
// bit-by-bit Tonelli-Shanks.  529 -> 45 + 253 on BLS12-377's base field (S = 46, w = 2); p = 3 (mod 4) (bn254, secp256k1: S = 1)
// degenerates to the single exponentiation a^((p+1)/4).  The tables (Ghalf: k 2^w entries, R: k 2^w, H: 2^w -- 50 KB
// on the STARK prime) are built once per context on the host with the same field code.
// Validation as ark-ec's deserialiser does it (and serialize_host.hpp, the host version of the same conversion): canonical x (< p),
// spare bits clear, infinity flag only with x = 0 and the sign flag clear, x on the curve, prime-order subgroup on curves with a
// cofactor; the sign flag selects y > -y (as canonical integers).
#pragma once
#include "kernels_msm.hpp"

namespace mp {

struct SqrtGeom {
  uint32_t S, w, k;          // 2-adicity, window width (divides S), windows
  uint32_t ebits;            // bit length of (q - 1) / 2
};
struct DecompressArgs {
  const uint8_t* in;         // groups of `per_group` compressed points, `prefix` bytes in front of each group (8: a Vec's u64 length)
  uint8_t* out;              // wire points, groups back to back
  int32_t* status;           // one word per group: 0 or -1 (MP_ERR_BAD_ENCODING); written only on failure (caller zeroes)
  const uint32_t* ghalf;     // [k][2^w] packed Montgomery words: g^(-d 2^(wi) / 2)
  const uint32_t* hh;        // [2^w]
  const uint32_t* rr;        // [k + 1][2^w]: R[c][d] = g^(-d 2^(S - w c)), rows 2 .. k used
  uint32_t* chain;           // scratch [k][lanes] field elements: T_j = t^(2^(w j)) of the launch's lanes (k > 1 only)
  uint32_t lanes, first;     // lanes of this launch (scratch stride) and the index of its first point
  uint32_t per_group, prefix;
  SqrtGeom g;
  uint32_t e[12];            // (q - 1) / 2, little-endian words
};

// sqrt(a) by the windowed walk above; false if a is not a square.  a != 0.  lane = this point's column of the scratch.
template <class F>
MP_HD bool fe_sqrt_windowed(const DecompressArgs& a, const Fe<F>& v, Fe<F>& out, uint32_t lane) {
  Fe<F> pw = fe_one<F>();
  for (int i = (int)a.g.ebits - 1; i >= 0; --i) {          // v^((q-1)/2), fixed exponent: uniform across lanes
    pw = fe_sqr<F>(pw);
    if ((a.e[i >> 5] >> (i & 31)) & 1u) pw = fe_mul<F>(pw, v);
  }
  Fe<F> x = fe_mul<F>(v, pw);
  Fe<F> t = fe_mul<F>(x, pw);                               // v^q
  const uint32_t nd = 1u << a.g.w, k = a.g.k;
  // the chain T_j = t^(2^(w j)), j < k, once (T_{k-1} stays in a register: window 0 wants it first)
  for (uint32_t j = 0; j + 1 < k; ++j) {
    st_fe_lazy<F>(a.chain + ((size_t)j * a.lanes + lane) * F::NW, t);
#pragma unroll 1
    for (uint32_t q = 0; q < a.g.w; ++q) t = fe_sqr<F>(t);
  }
  bool square = true;
  uint32_t dig[48];                                         // e_j, low window first (k <= 48: S <= 192 at w = 4)
#pragma unroll 1
  for (uint32_t i = 0; i < k; ++i) {
    Fe<F> u = i == 0 ? t : ld_fe<F>(a.chain + ((size_t)(k - 1 - i) * a.lanes + lane) * F::NW);
#pragma unroll 1
    for (uint32_t j = 0; j < i; ++j) u = fe_mul<F>(u, ld_fe<F>(a.rr + ((size_t)(i + 1 - j) * nd + dig[j]) * F::NW));
    uint32_t d = 0;
    bool found = false;
#pragma unroll 1
    for (uint32_t c = 0; c < nd; ++c) {
      const bool eq = fe_eq<F>(u, ld_fe<F>(a.hh + (size_t)c * F::NW));
      d = eq ? c : d;
      found = found || eq;
    }
    if (!found) square = false;                             // (cannot happen for an element of the field: u lies in <h>)
    if (i == 0 && (d & 1u)) square = false;                 // odd discrete logarithm: not a square
    dig[i] = d;
    x = fe_mul<F>(x, ld_fe<F>(a.ghalf + ((size_t)i * nd + d) * F::NW));
  }
  out = x;
  return square && fe_eq<F>(fe_sqr<F>(x), v);
}

// thread x = point index over all groups
template <class C>
MP_HD void body_decompress(const DecompressArgs& a, uint32_t lane, uint32_t) {
  typedef typename C::FqP F;
  const uint32_t idx = a.first + lane;
  constexpr uint32_t CB = (F::BITS + 2 + 7) / 8, FB = 4 * F::NW, PB = 8 * F::NW;
  const uint32_t grp = idx / a.per_group, j = idx - grp * a.per_group;
  const uint8_t* src = a.in + (size_t)grp * (a.prefix + (size_t)a.per_group * CB) + a.prefix + (size_t)j * CB;
  uint32_t* dst = reinterpret_cast<uint32_t*>(a.out + ((size_t)grp * a.per_group + j) * PB);
  bool ok = true;
  if (a.prefix == 8 && j == 0) {                            // Vec<T>: u64 little-endian length in front of the group
    uint64_t len = 0;
    for (int b = 0; b < 8; ++b) len |= (uint64_t)src[b - 8] << (8 * b);
    if (len != (uint64_t)a.per_group / 2) ok = false;       // (a deck: `cards` ciphertexts = 2 points each)
  }
  uint32_t xw[F::NW + 1];
  for (uint32_t i = 0; i <= F::NW; ++i) xw[i] = 0;
  for (uint32_t b = 0; b < CB; ++b) xw[b >> 2] |= (uint32_t)src[b] << (8 * (b & 3));
  const uint32_t fl_word = (CB - 1) >> 2, fl_shift = 8 * ((CB - 1) & 3);
  const uint32_t flags = (xw[fl_word] >> fl_shift) & 0xC0u;
  xw[fl_word] &= ~(0xC0u << fl_shift);
  if (xw[F::NW] != 0) ok = false;                           // (33-byte secp256k1 encoding: the spare byte carries only flags)
  uint32_t yw[F::NW];
  for (uint32_t i = 0; i < F::NW; ++i) yw[i] = 0;
  if (flags & 0x40u) {                                      // infinity: x = 0, sign flag clear -> all-zero wire point
    bool zero = (flags & 0x80u) == 0;
    for (uint32_t i = 0; i < F::NW; ++i) zero = zero && xw[i] == 0;
    if (!zero) ok = false;
  } else if (ok && fe_canonical_in_range<F>(xw)) {
    const Fe<F> x = fe_from_canonical<F>(xw);
    Fe<F> rhs = fe_add<F>(fe_mul<F>(fe_sqr<F>(x), x), fe_unpack<F>(C::B_MONT));
    if (C::A == 1) rhs = fe_add<F>(rhs, x);
    Fe<F> y = fe_zero<F>();
    if (!fe_is_zero(rhs) && !fe_sqrt_windowed<F>(a, rhs, y, lane)) ok = false;
    uint32_t nyw[F::NW];
    fe_to_canonical<F>(y, yw);
    fe_to_canonical<F>(fe_neg<F>(y), nyw);
    bool greater = false, differ = false;
    for (int i = F::NW - 1; i >= 0; --i)
      if (!differ && yw[i] != nyw[i]) {
        differ = true;
        greater = yw[i] > nyw[i];
      }
    if (greater != ((flags & 0x80u) != 0))
      for (uint32_t i = 0; i < F::NW; ++i) yw[i] = nyw[i];
    if (ok && !Cofactor<C>::ONE) {
      Aff<C> p;
      p.x = x;
      p.y = fe_from_canonical<F>(yw);
      if (!aff_is_inf<C>(p) && !aff_in_subgroup_dev<C>(p)) ok = false;      // (kernels_msm.hpp: the endomorphism test on BLS12-377)
    }
  } else {
    ok = false;
  }
  if (!ok) {
    a.status[grp] = -1;                                     // MP_ERR_BAD_ENCODING (same value from every failing lane)
    for (uint32_t i = 0; i < F::NW; ++i) xw[i] = yw[i] = 0;
  }
  for (uint32_t i = 0; i < F::NW; ++i) {
    dst[i] = xw[i];
    dst[F::NW + i] = yw[i];
  }
}
MP_KERNEL_OCC(k_decompress, DecompressArgs, body_decompress, Geo<C>::OCC3)

#define MP_DECOMPRESS_KERNELS(X, C) MP_KERNEL_INST(X, k_decompress, DecompressArgs, C)

}  // namespace mp
