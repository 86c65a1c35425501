// host check of the division-step inversion (field.hpp fe_inv_divsteps) against the Fermat ladder it replaces, on the base
// fields of the four curves and two scalar fields; built and run by tests/test_cabi_and_host.py::test_division_step_inversion_matches_fermat
//   g++ -O2 -std=c++17 -include tools/hostemu/rt.hpp -Itools/hostemu -Imental-poker_amd/csrc tests/cpp/inv_check.cpp
#include <cstdio>
#include <cstdlib>

static long fallbacks = 0;
#define MP_DIVSTEPS_COUNT_FALLBACKS fallbacks
#include "curve.hpp"

using namespace mp;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 16);
}

template <class F>
static long check(const char* name, long count) {
  long bad = 0, done = 0;
  auto one = [&](const Fe<F>& a) {
    const Fe<F> x = fe_inv_divsteps<F>(a), y = fe_inv_fermat<F>(a);
    if (!fe_eq<F>(x, y)) ++bad;
    if (!fe_is_zero(a) && !fe_eq<F>(fe_mul<F>(a, x), fe_one<F>())) ++bad;
    ++done;
  };
  one(fe_zero<F>());
  one(fe_one<F>());
  one(fe_neg<F>(fe_one<F>()));
  for (uint32_t k = 2; k < 200; ++k) {
    one(fe_from_u32<F>(k));
    one(fe_neg<F>(fe_from_u32<F>(k)));
    // lazily reduced representatives: sums and differences leave values in [0, 4p) that are not canonical
    Fe<F> s = fe_from_u32<F>(k);
    for (int j = 0; j < 5; ++j) s = fe_sub<F>(fe_add<F>(s, fe_neg<F>(fe_from_u32<F>(3))), fe_neg<F>(fe_from_u32<F>(3)));
    one(s);
  }
  for (long i = 0; i < count; ++i) {
    uint32_t w[F::NW];
    for (int j = 0; j < F::NW; ++j) w[j] = rnd();
    if (i % 7 == 0)
      for (int j = 1 + (int)(rnd() % 7); j < F::NW; ++j) w[j] = 0;            // short values
    if (i % 11 == 0)
      for (int j = 0; j < F::NW; ++j) w[j] = (rnd() & 1u) ? 0xFFFFFFFFu : 0u;   // runs of ones / zeros
    if (F::BITS < 32 * F::NW) w[F::NW - 1] &= 0xFFFFFFFFu >> (32 * F::NW - F::BITS);
    if (!fe_canonical_in_range<F>(w)) continue;
    one(fe_from_canonical<F>(w));
  }
  printf("%s: %ld inversions, %ld mismatches, %ld answered by the fallback\n", name, done, bad, fallbacks);
  return bad + fallbacks;
}

int main(int argc, char** argv) {
  const long count = argc > 1 ? atol(argv[1]) : 20000;
  long bad = 0;
  bad += check<Stark::FqP>("stark Fq", count);
  bad += check<Secp256k1::FqP>("secp256k1 Fq", count);
  bad += check<Bn254::FqP>("bn254 Fq", count);
  bad += check<Bls12_377::FqP>("bls12_377 Fq", count / 4);
  bad += check<Stark::FrP>("stark Fr", count / 4);
  bad += check<Bls12_377::FrP>("bls12_377 Fr", count / 4);
  return bad ? 1 : 0;
}
