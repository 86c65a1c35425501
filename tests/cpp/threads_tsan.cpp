// Threading contract of the C ABI (include/mpshuffle.h) under ThreadSanitizer, on the development emulator (kernel bodies as plain CPU
// loops, compiled WITHOUT OpenMP here so that every access is one the sanitizer follows).
//   (a) four host threads, each with a context and a table of its own: setup, prove, verify (one of them verifies a tampered proof, one
//       runs pipelined verification) -- the bytes every thread gets are the bytes a single-threaded run gets;
//   (b) two host threads on ONE table: the library serialises the calls on the context's lock; same bytes, same status words.
// The reference's trait members are associated functions without `self` or global state [REF barnett-smart-card-protocol/src/lib.rs:74-197].
// Built and run by tests/test_threads_tsan.py; exit code 0 and no "WARNING: ThreadSanitizer" on stderr = pass.
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mpshuffle.h"

static const uint32_t M = 2, N_ = 3, B = 6;
#define CHECK(x)                                                                   \
  do {                                                                             \
    if (!(x)) {                                                                    \
      fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #x, mp_last_error()); \
      std::abort();                                                                \
    }                                                                              \
  } while (0)

struct Inputs {
  std::vector<uint8_t> params, pk, decks, rho, seeds;
  std::vector<uint32_t> perm;
};
struct Outputs {
  std::vector<uint8_t> decks, proofs;
  std::vector<int32_t> st_bad;
};

static Inputs make_inputs() {
  Inputs in;
  mp_ctx* ctx = nullptr;
  CHECK(mp_ctx_create(MP_CURVE_STARK, 0, &ctx) == 0);
  const uint32_t cards = M * N_;
  in.params.resize(mp_params_size(N_));
  uint8_t seed[32];
  memset(seed, 7, 32);
  CHECK(mp_setup(ctx, M, N_, seed, in.params.data()) == 0);
  std::vector<uint8_t> more(mp_params_size(2 * cards * B));      // (2 cards B + 3 points: one key, the decks)
  memset(seed, 9, 32);
  CHECK(mp_setup(ctx, M, 2 * cards * B, seed, more.data()) == 0);
  in.pk.assign(more.begin(), more.begin() + 64);
  in.decks.assign(more.begin() + 64, more.begin() + 64 + (size_t)B * cards * 128);
  in.rho.resize((size_t)B * cards * 32);
  for (size_t i = 0; i < in.rho.size(); ++i) in.rho[i] = (uint8_t)((i * 37 + 11) & 0xFF);
  for (size_t i = 31; i < in.rho.size(); i += 32) in.rho[i] &= 3;
  in.perm.resize((size_t)B * cards);
  for (uint32_t b = 0; b < B; ++b)
    for (uint32_t i = 0; i < cards; ++i) in.perm[b * cards + i] = (i * 5 + b) % cards;      // (5 and 6 coprime)
  in.seeds.resize((size_t)B * 32);
  for (size_t i = 0; i < in.seeds.size(); ++i) in.seeds[i] = (uint8_t)(i * 13 + 5);
  mp_ctx_destroy(ctx);
  return in;
}

static void prove_verify(mp_table* t, const Inputs& in, Outputs& out, bool check_bad, int iterations) {
  const uint32_t cards = M * N_;
  const size_t psz = mp_proof_size(M, N_);
  for (int it = 0; it < iterations; ++it) {
    std::vector<uint8_t> od((size_t)B * cards * 128), op(B * psz);
    std::vector<int32_t> st(B, 55);
    CHECK(mp_shuffle_and_remask_batch(t, B, in.decks.data(), in.rho.data(), in.perm.data(), in.seeds.data(), od.data(), op.data(), st.data()) == 0);
    for (int32_t v : st) CHECK(v == 0);
    if (out.decks.empty()) {
      out.decks = od;
      out.proofs = op;
    }
    CHECK(od == out.decks && op == out.proofs);
    CHECK(mp_verify_shuffle_batch(t, B, in.decks.data(), od.data(), op.data(), st.data()) == 0);
    for (int32_t v : st) CHECK(v == 0);
    if (check_bad) {
      op[3 * psz + psz - 31] ^= 2;
      CHECK(mp_verify_shuffle_batch(t, B, in.decks.data(), od.data(), op.data(), st.data()) == 0);
      if (out.st_bad.empty()) out.st_bad = st;
      CHECK(st == out.st_bad && st[3] > 0 && st[0] == 0);
    }
  }
}

int main() {
  const Inputs in = make_inputs();
  Outputs ref;
  {
    mp_ctx* ctx = nullptr;
    mp_table* t = nullptr;
    CHECK(mp_ctx_create(MP_CURVE_STARK, 0, &ctx) == 0);
    CHECK(mp_table_create_ex(ctx, M, N_, in.params.data(), in.pk.data(), 8, &t) == 0);
    prove_verify(t, in, ref, true, 1);
    mp_table_destroy(t);
    mp_ctx_destroy(ctx);
  }
  // (a) four threads, four contexts
  {
    std::vector<std::thread> th;
    for (int k = 0; k < 4; ++k)
      th.emplace_back([&, k] {
        mp_ctx* ctx = nullptr;
        mp_table* t = nullptr;
        CHECK(mp_ctx_create(MP_CURVE_STARK, 0, &ctx) == 0);
        CHECK(mp_table_create_ex(ctx, M, N_, in.params.data(), in.pk.data(), 8, &t) == 0);
        if (k == 2) CHECK(mp_set_group_verify(t, 3 * (4 * M * N_ + 11 * M + 8), 0) == 0);      // the group screen on the bucket kernels
        if (k == 3) CHECK(mp_set_work_split(t, 0) == 0);
        Outputs mine = ref;
        prove_verify(t, in, mine, k == 1 || k == 2, 2);
        mp_table_destroy(t);
        mp_ctx_destroy(ctx);
      });
    for (auto& x : th) x.join();
  }
  // (b) two threads, one table (and the setters from a third)
  {
    mp_ctx* ctx = nullptr;
    mp_table* t = nullptr;
    CHECK(mp_ctx_create(MP_CURVE_STARK, 0, &ctx) == 0);
    CHECK(mp_table_create_ex(ctx, M, N_, in.params.data(), in.pk.data(), 8, &t) == 0);
    std::vector<std::thread> th;
    for (int k = 0; k < 2; ++k)
      th.emplace_back([&, k] {
        Outputs mine = ref;
        prove_verify(t, in, mine, k == 0, 3);
      });
    th.emplace_back([&] {
      for (int i = 0; i < 6; ++i) {
        CHECK(mp_set_work_split(t, i % 2 ? -1 : 0) == 0);
        CHECK(mp_set_group_adapt(t, i % 2) == 0);
        CHECK(mp_sync(ctx) == 0);
      }
    });
    for (auto& x : th) x.join();
    mp_table_destroy(t);
    mp_ctx_destroy(ctx);
  }
  printf("threads ok\n");
  return 0;
}
