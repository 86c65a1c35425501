// host check of the base-field arithmetic (field.hpp: 9x29 sparse / signed-sparse / dense and 14x29 dense lazy limbs, 8x32 words) against an
// independent schoolbook big-integer reference (multi-word product, remainder by shift-and-subtract): products, squares, fused
// a b - c d, sums and differences of LAZILY reduced operands (chains of additions and subtractions that leave values anywhere in the
// representation's allowed range), zero tests on every representative of zero a chain can produce, pack / unpack round trips.
// Built and run by tests/test_cabi_and_host.py::test_field_arithmetic_matches_bigint_reference
//   g++ -O2 -std=c++17 -include tools/hostemu/rt.hpp -Itools/hostemu -Imental-poker_amd/csrc tests/cpp/field_check.cpp
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "curve.hpp"

using namespace mp;

static uint64_t rng_state = 0xD1B54A32D192ED03ull;
static uint32_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 16);
}

// ---- reference: little-endian 32-bit words -------------------------------------------------------------------------
template <int N>
struct Big {
  uint32_t w[N];
};
template <int N>
static int cmp(const uint32_t* a, const uint32_t* b) {
  for (int i = N - 1; i >= 0; --i)
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
template <int N>
static void sub_in_place(uint32_t* a, const uint32_t* b) {
  uint64_t br = 0;
  for (int i = 0; i < N; ++i) {
    const uint64_t t = (uint64_t)a[i] - b[i] - br;
    a[i] = (uint32_t)t;
    br = (t >> 32) & 1;
  }
}
// r = x mod p for a 2N-word x (bitwise long division; N + 1 working words)
template <int N>
static void mod_wide(const uint32_t* x, const uint32_t* p, uint32_t* r) {
  uint32_t rem[N + 1], pp[N + 1];
  memset(rem, 0, sizeof rem);
  memcpy(pp, p, 4 * N);
  pp[N] = 0;
  for (int bit = 64 * N - 1; bit >= 0; --bit) {
    for (int i = N; i > 0; --i) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 31);
    rem[0] = (rem[0] << 1) | ((x[bit >> 5] >> (bit & 31)) & 1u);
    if (cmp<N + 1>(rem, pp) >= 0) sub_in_place<N + 1>(rem, pp);
  }
  memcpy(r, rem, 4 * N);
}
template <int N>
static void ref_mul(const uint32_t* a, const uint32_t* b, const uint32_t* p, uint32_t* r) {
  uint32_t x[2 * N];
  memset(x, 0, sizeof x);
  for (int i = 0; i < N; ++i) {
    uint64_t c = 0;
    for (int j = 0; j < N; ++j) {
      c += (uint64_t)a[i] * b[j] + x[i + j];
      x[i + j] = (uint32_t)c;
      c >>= 32;
    }
    x[i + N] = (uint32_t)c;
  }
  mod_wide<N>(x, p, r);
}
template <int N>
static void ref_add(const uint32_t* a, const uint32_t* b, const uint32_t* p, uint32_t* r) {
  uint32_t x[2 * N];
  memset(x, 0, sizeof x);
  uint64_t c = 0;
  for (int i = 0; i < N; ++i) {
    c += (uint64_t)a[i] + b[i];
    x[i] = (uint32_t)c;
    c >>= 32;
  }
  x[N] = (uint32_t)c;
  mod_wide<N>(x, p, r);
}
template <int N>
static void ref_sub(const uint32_t* a, const uint32_t* b, const uint32_t* p, uint32_t* r) {   // a, b < p
  uint32_t nb[N];
  memcpy(nb, p, 4 * N);
  sub_in_place<N>(nb, b);                       // p - b
  ref_add<N>(a, nb, p, r);
}

template <class F>
static long check(const char* name, long count) {
  constexpr int N = F::NW;
  long bad = 0, done = 0;
  auto rand_canon = [&](uint32_t* w, long i) {
    for (;;) {
      for (int j = 0; j < N; ++j) w[j] = rnd();
      if (i % 7 == 0)
        for (int j = 1 + (int)(rnd() % (N - 1)); j < N; ++j) w[j] = 0;
      if (i % 11 == 0)
        for (int j = 0; j < N; ++j) w[j] = (rnd() & 1u) ? 0xFFFFFFFFu : 0u;
      if (i % 13 == 0) {                         // p - small
        memcpy(w, F::MOD, 4 * N);
        w[0] -= 1 + (rnd() & 7u);
      }
      if (F::BITS < 32 * N) w[N - 1] &= 0xFFFFFFFFu >> (32 * N - F::BITS);
      if (fe_canonical_in_range<F>(w)) return;
    }
  };
  auto expect = [&](const Fe<F>& got, const uint32_t* want, const char* what) {
    uint32_t g[N];
    fe_to_canonical<F>(got, g);
    if (memcmp(g, want, 4 * N) != 0) {
      if (bad < 5) printf("  %s: %s mismatch\n", name, what);
      ++bad;
    }
    // the packed memory format round-trips and holds a canonical residue
    uint32_t pk[N];
    fe_pack<F>(got, pk);
    if (!fe_canonical_in_range<F>(pk)) ++bad;
    uint32_t g2[N];
    fe_to_canonical<F>(fe_unpack<F>(pk), g2);
    if (memcmp(g2, want, 4 * N) != 0) ++bad;
    ++done;
  };
  for (long i = 0; i < count; ++i) {
    uint32_t a[N], b[N], c[N], d[N], r[N], t1[N], t2[N];
    rand_canon(a, i); rand_canon(b, i + 1); rand_canon(c, i + 2); rand_canon(d, i + 3);
    const Fe<F> A = fe_from_canonical<F>(a), B = fe_from_canonical<F>(b), C = fe_from_canonical<F>(c), D = fe_from_canonical<F>(d);
    ref_mul<N>(a, b, F::MOD, r);
    expect(fe_mul<F>(A, B), r, "mul");
    ref_mul<N>(a, a, F::MOD, r);
    expect(fe_sqr<F>(A), r, "sqr");
    ref_mul<N>(a, b, F::MOD, t1); ref_mul<N>(c, d, F::MOD, t2); ref_sub<N>(t1, t2, F::MOD, r);
    expect(fe_mulsub<F>(A, B, C, D), r, "mulsub");
    ref_add<N>(a, b, F::MOD, r);
    expect(fe_add<F>(A, B), r, "add");
    ref_sub<N>(a, b, F::MOD, r);
    expect(fe_sub<F>(A, B), r, "sub");
    // lazily reduced operands: x = ((a - b) + c) - d + (a - b), y = (c + c) - (b + d); then x y, x^2, x y - (a - b) c
    const Fe<F> AB = fe_sub<F>(A, B);
    const Fe<F> X = fe_add<F>(fe_sub<F>(fe_add<F>(AB, C), D), AB);
    const Fe<F> Y = fe_sub<F>(fe_dbl<F>(C), fe_add<F>(B, D));
    uint32_t ab[N], x[N], y[N], u[N];
    ref_sub<N>(a, b, F::MOD, ab);
    ref_add<N>(ab, c, F::MOD, u); ref_sub<N>(u, d, F::MOD, x); ref_add<N>(x, ab, F::MOD, x);
    ref_add<N>(c, c, F::MOD, u); ref_add<N>(b, d, F::MOD, t1); ref_sub<N>(u, t1, F::MOD, y);
    expect(X, x, "lazy chain x");
    expect(Y, y, "lazy chain y");
    ref_mul<N>(x, y, F::MOD, r);
    expect(fe_mul<F>(X, Y), r, "mul of lazy operands");
    ref_mul<N>(x, x, F::MOD, r);
    expect(fe_sqr<F>(X), r, "sqr of a lazy operand");
    ref_mul<N>(x, y, F::MOD, t1); ref_mul<N>(ab, c, F::MOD, t2); ref_sub<N>(t1, t2, F::MOD, r);
    expect(fe_mulsub<F>(X, Y, AB, C), r, "mulsub of lazy operands");
    ref_sub<N>(y, x, F::MOD, r);
    expect(fe_sub<F>(Y, X), r, "sub of lazy operands");
    expect(fe_neg<F>(fe_neg<F>(X)), x, "double negation");
    // one-pass combinations of products (fe_sub_sub_dbl, fe_sub_dbl, fe_triple_add: arguments are direct products, as in curve.hpp)
    {
      const Fe<F> P1 = fe_mul<F>(X, Y), P2 = fe_sqr<F>(X), P3 = fe_mulsub<F>(X, Y, AB, C);
      uint32_t p1[N], p2[N], p3[N], w1[N], w2[N];
      ref_mul<N>(x, y, F::MOD, p1); ref_mul<N>(x, x, F::MOD, p2);
      ref_mul<N>(ab, c, F::MOD, w1); ref_sub<N>(p1, w1, F::MOD, p3);
      ref_sub<N>(p2, p1, F::MOD, w1); ref_add<N>(p3, p3, F::MOD, w2); ref_sub<N>(w1, w2, F::MOD, r);
      expect(fe_sub_sub_dbl<F>(P2, P1, P3), r, "a - u - 2v");
      ref_add<N>(p1, p1, F::MOD, w1); ref_sub<N>(p2, w1, F::MOD, r);
      expect(fe_sub_dbl<F>(P2, P1), r, "a - 2v");
      ref_add<N>(p2, p2, F::MOD, w1); ref_add<N>(w1, p2, F::MOD, w1); ref_add<N>(w1, p3, F::MOD, r);
      expect(fe_triple_add<F>(P2, P3), r, "3a + b");
      // product minus a lazily reduced value without the weak reduction: as a multiplicand, squared, and as a zero
      const Fe<F> Wd = fe_sub_wide<F>(P1, X);
      ref_sub<N>(p1, x, F::MOD, w1); ref_mul<N>(w1, y, F::MOD, r);
      expect(fe_mul<F>(Wd, Y), r, "(product - x) y, wide difference");
      ref_mul<N>(w1, w1, F::MOD, r);
      expect(fe_sqr<F>(Wd), r, "(product - x)^2, wide difference");
      if (!fe_is_zero(fe_sub_wide<F>(P1, fe_add<F>(P1, fe_sub<F>(X, X)))) || !fe_is_zero(fe_sub_wide<F>(P2, fe_sqr<F>(fe_neg<F>(X))))) {
        if (bad < 5) printf("  %s: a wide zero is not recognised\n", name);
        ++bad;
      }
    }
    // representatives of zero
    const Fe<F> Z1 = fe_sub<F>(X, X), Z2 = fe_add<F>(X, fe_neg<F>(X)), Z3 = fe_sub<F>(fe_add<F>(A, B), fe_add<F>(B, A));
    const Fe<F> Z4 = fe_add<F>(fe_add<F>(Z1, Z2), fe_add<F>(Z3, Z2));
    if (!fe_is_zero(Z1) || !fe_is_zero(Z2) || !fe_is_zero(Z3) || !fe_is_zero(Z4) || !fe_eq<F>(fe_add<F>(X, Z4), X)) {
      if (bad < 5) printf("  %s: a representative of zero is not recognised\n", name);
      ++bad;
    }
    bool xz = true;
    for (int j = 0; j < N; ++j) xz = xz && x[j] == 0;
    if (fe_is_zero(X) != xz) ++bad;
  }
  // the fused a (b - c) - d e with a carry-free difference and negation (field.hpp LazySub; plain functions on the other forms) against
  // the unfused, fully normalising evaluation -- on random operands and on the limb patterns that maximise the column sums
  // (every limb 2^29 - 1 under the largest top limb a value below 4p / 2p can have)
  if constexpr (F::L29) {
    constexpr int NL = F::NL29;
    auto extreme = [&](int kind) {
      Fe<F> e;
      for (int i = 0; i < NL - 1; ++i) e.v[i] = kind == 2 ? 0u : M29;
      uint32_t top = (uint32_t)((F::DENSE29 ? 2 : 4) * (uint64_t)F::MOD29[NL - 1]);
      if (F::PM29) top = (4u << 24) - 1u;
      e.v[NL - 1] = kind == 1 ? 0u : (top > 0 ? top - 1 : 0);
      return e;
    };
    long lazy_bad = 0, lazy_done = 0;
    for (long i = 0; i < count + 81; ++i) {
      Fe<F> o[5];
      for (int j = 0; j < 5; ++j) {
        uint32_t w[N];
        rand_canon(w, i + j);
        o[j] = fe_from_canonical<F>(w);
        if (i < 81) {                       // all 3^4 combinations of the extreme patterns in b, c, d, e (a = the largest)
          int sel = (int)(i / (j == 0 ? 1 : j == 1 ? 1 : j == 2 ? 3 : j == 3 ? 9 : 27)) % 3;
          o[j] = extreme(j == 0 ? 0 : sel);
        } else if ((i + j) % 5 == 0) {
          o[j] = fe_sub<F>(o[j], fe_from_canonical<F>(w));      // a lazily reduced zero
        }
      }
      const Fe<F> fused = fe_mulsub<F>(o[0], fe_sub_lazy<F>(o[1], o[2]), o[3], o[4]);
      const Fe<F> plain = fe_sub<F>(fe_mul<F>(o[0], fe_sub<F>(o[1], o[2])), fe_mul<F>(o[3], o[4]));
      const Fe<F> single = fe_mul<F>(o[0], fe_neg_lazy<F>(o[3]));
      const Fe<F> single_plain = fe_mul<F>(o[0], fe_neg<F>(o[3]));
      if (!fe_eq<F>(fused, plain) || !fe_eq<F>(single, single_plain)) {
        if (lazy_bad < 5) printf("  %s: fused lazy product differs (case %ld)\n", name, i);
        ++lazy_bad;
      }
      for (int l = 0; l < NL - 1; ++l)
        if (fused.v[l] > M29) ++lazy_bad;                        // outputs are normalised
      ++lazy_done;
    }
    printf("  %s: %ld fused products with carry-free operands, %ld mismatches\n", name, lazy_done, lazy_bad);
    bad += lazy_bad;
  }
  printf("%s: %ld checks, %ld mismatches\n", name, done, bad);
  return bad;
}

int main(int argc, char** argv) {
  const long count = argc > 1 ? atol(argv[1]) : 3000;
  long bad = 0;
  bad += check<Stark::FqP>("stark Fq (9x29 sparse)", count);
  bad += check<Secp256k1::FqP>("secp256k1 Fq (9x29 signed sparse)", count);
  bad += check<Bn254::FqP>("bn254 Fq (9x29 dense)", count);
  bad += check<Bls12_377::FqP>("bls12-377 Fq (14x29 dense)", count / 2);
  bad += check<Stark::FrP>("stark Fr (8x32)", count);
  bad += check<Bn254::FrP>("bn254 Fr (8x32)", count);
  return bad ? 1 : 0;
}
