// setup_host.hpp -- DLCards::setup: the shared parameters of a card table, sampled WITHOUT a known discrete logarithm.
//
// The reference calls `Enc::setup(rng)`, `Comm::setup(rng, n)` and `Enc::generator(rng)`
// [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:105-121]; each of them draws its group elements with
// `C::rand(rng)` (ark-ec 0.3 `GroupAffine::rand` [UPSTREAM-RECALL]: x = Fq::rand, greatest = rng.gen::<bool>(), lift x to the
// curve or retry, scale by the cofactor).  Round 1 of this engine derived the n + 3 generators as k_i * G_std from the seed,
// which made the seed a trapdoor of the Pedersen key (knowing the k_i breaks binding, hence the soundness of the shuffle
// argument).  "setup v2" below follows the reference: nobody, including the holder of the seed, learns a relation between
// G, ck_0..ck_{n-1}, H and gen.  Runs on the host (n + 3 square roots, once per table); the executable definition is
// oracle/py/mp_oracle.py::setup.
#pragma once
#include <vector>

#include "curve.hpp"
#include "hash.hpp"

namespace mp {

// ChaCha20Rng as a stream of 32-bit words (rand_chacha's BlockRng: next_u64 = two consecutive words, low word first)
struct WordStream {
  uint32_t key[8];
  uint32_t blk[16];
  uint64_t counter = 0;
  uint32_t idx = 16;
  explicit WordStream(const uint8_t seed[32]) { memcpy(key, seed, 32); }
  uint32_t next_u32() {
    if (idx >= 16) {
      chacha20_block(key, counter++, blk);
      idx = 0;
    }
    return blk[idx++];
  }
};

template <class F>
static Fe<F> fe_pow_host(const Fe<F>& a, const uint32_t* e, int words) {
  Fe<F> acc = fe_one<F>();
  for (int i = 32 * words - 1; i >= 0; --i) {
    acc = fe_sqr<F>(acc);
    if ((e[i >> 5] >> (i & 31)) & 1u) acc = fe_mul<F>(acc, a);
  }
  return acc;
}

// `Fq::rand`: NW words (= NW/2 u64 limbs, limb 0 first), top bits beyond the modulus cleared, accepted if < p; the accepted
// limbs are the Montgomery representation with R = 2^(32 NW) -- i.e. the element limbs / 2^(32 NW) mod p
template <class F>
static Fe<F> fq_rand_host(WordStream& rng, const Fe<F>& rinv) {
  for (;;) {
    uint32_t w[F::NW];
    for (int i = 0; i < F::NW; ++i) w[i] = rng.next_u32();
    if (F::BITS < 32 * F::NW) w[F::NW - 1] &= 0xFFFFFFFFu >> (32 * F::NW - F::BITS);
    if (!fe_canonical_in_range<F>(w)) continue;
    return fe_mul<F>(fe_from_canonical<F>(w), rinv);
  }
}

// Tonelli-Shanks square root; false if `a` is not a square
template <class F>
static bool fe_sqrt_host(const Fe<F>& a, Fe<F>& out) {
  constexpr int W = F::NW;
  if (fe_is_zero(a)) {
    out = a;
    return true;
  }
  uint32_t pm1[W], half[W], t[W], t1h[W];
  for (int i = 0; i < W; ++i) pm1[i] = F::MOD[i];
  pm1[0] -= 1u;                                              // p odd
  auto shr1 = [](uint32_t* r, const uint32_t* x) {
    for (int i = 0; i < W; ++i) r[i] = (x[i] >> 1) | (i + 1 < W ? x[i + 1] << 31 : 0u);
  };
  shr1(half, pm1);
  const Fe<F> one = fe_one<F>(), minus_one = fe_neg<F>(one);
  if (!fe_eq<F>(fe_pow_host<F>(a, half, W), one)) return false;
  int s = 0;
  for (int i = 0; i < W; ++i) t[i] = pm1[i];
  while (!(t[0] & 1u)) {
    shr1(t, t);
    ++s;
  }
  {
    uint64_t c = 1;
    for (int i = 0; i < W; ++i) {
      c += t[i];
      t1h[i] = (uint32_t)c;
      c >>= 32;
    }
    shr1(t1h, t1h);
  }
  Fe<F> z = fe_from_u32<F>(2);
  for (uint32_t k = 2; !fe_eq<F>(fe_pow_host<F>(z, half, W), minus_one);) z = fe_from_u32<F>(++k);
  Fe<F> c = fe_pow_host<F>(z, t, W), r = fe_pow_host<F>(a, t1h, W), tt = fe_pow_host<F>(a, t, W);
  int M = s;
  while (!fe_eq<F>(tt, one)) {
    int i = 0;
    Fe<F> u = tt;
    while (!fe_eq<F>(u, one)) {
      u = fe_sqr<F>(u);
      ++i;
    }
    Fe<F> b = c;
    for (int k = 0; k < M - i - 1; ++k) b = fe_sqr<F>(b);
    r = fe_mul<F>(r, b);
    c = fe_sqr<F>(b);
    tt = fe_mul<F>(tt, c);
    M = i;
  }
  out = r;
  return true;
}

// [k]P for a small multi-word integer k (host)
template <class C>
static Aff<C> aff_mul_words_host(const Aff<C>& p, const uint32_t* k, int words) {
  typedef typename C::FqP F;
  Jac<C> acc = jac_inf<C>();
  for (int i = 32 * words - 1; i >= 0; --i) {
    jac_dbl_ip<C>(acc);
    if ((k[i >> 5] >> (i & 31)) & 1u) jac_madd_ip<C>(acc, p);
  }
  if (jac_is_inf<C>(acc)) return aff_inf<C>();
  return jac_to_aff_with_zinv<C>(acc, fe_inv<F>(acc.Z));
}

// is [q]P the identity (q = the prime group order)?  Trivially true on prime-order curves.
template <class C>
static bool aff_in_subgroup_host(const Aff<C>& p) {
  if (Cofactor<C>::ONE || aff_is_inf<C>(p)) return true;
  return aff_is_inf<C>(aff_mul_words_host<C>(p, C::FrP::MOD, C::FrP::NW));
}

template <class C>
static Aff<C> point_rand_host(WordStream& rng, const Fe<typename C::FqP>& rinv) {
  typedef typename C::FqP F;
  const Fe<F> b = fe_unpack<F>(C::B_MONT);
  for (;;) {
    const Fe<F> x = fq_rand_host<F>(rng, rinv);
    const bool greatest = (rng.next_u32() >> 31) & 1u;
    Fe<F> rhs = fe_add<F>(fe_mul<F>(fe_sqr<F>(x), x), b);
    if (C::A == 1) rhs = fe_add<F>(rhs, x);
    Fe<F> y;
    if (!fe_sqrt_host<F>(rhs, y)) continue;
    const Fe<F> ny = fe_neg<F>(y);
    uint32_t yi[F::NW], nyi[F::NW];
    fe_to_canonical<F>(y, yi);
    fe_to_canonical<F>(ny, nyi);
    bool y_larger = false;
    for (int i = F::NW - 1; i >= 0; --i)
      if (yi[i] != nyi[i]) {
        y_larger = yi[i] > nyi[i];
        break;
      }
    Aff<C> p;
    p.x = x;
    p.y = (y_larger == greatest) ? y : ny;
    if (!Cofactor<C>::ONE) p = aff_mul_words_host<C>(p, Cofactor<C>::H, 4);
    return p;
  }
}

// G | ck_0 .. ck_{n-1} | H | gen as wire points
template <class C>
static void setup_points_host(uint32_t n, const uint8_t seed[32], uint8_t* out, void (*to_wire)(const Aff<C>&, uint8_t*), uint32_t pb) {
  typedef typename C::FqP F;
  WordStream rng(seed);
  // 1 / 2^(32 NW) mod p in the engine's own representation
  Fe<F> two_k = fe_one<F>();
  for (int i = 0; i < 32 * F::NW; ++i) two_k = fe_dbl<F>(two_k);
  const Fe<F> rinv = fe_inv<F>(two_k);
  for (uint32_t i = 0; i < n + 3; ++i) to_wire(point_rand_host<C>(rng, rinv), out + (size_t)i * pb);
}

}  // namespace mp
