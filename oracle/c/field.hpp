// TEST INFRASTRUCTURE ONLY (oracle).  N x 64-bit-limb Montgomery prime field, the representation
// ark-ff 0.3 `Fp256` / `Fp384` use ("Fp repr: 4xu64 little-endian limbs in Montgomery form,
// R = 2^256 (6 limbs, R = 2^384 for BLS12-377 Fq)").  The arithmetic behind every reference call on the hot path
// [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:7-8 -- ark_ec / ark_ff imports]
// is in arkworks 0.3.0 (Cargo.toml:11-12), not under /root/reference: this restates the published
// CIOS Montgomery algorithm.  PARITY UNPINNED (see oracle/README.md).
#pragma once
#include <cstdint>
#include <cstring>

namespace mpo {
typedef uint64_t u64;
typedef unsigned __int128 u128;

struct U256 {
  u64 l[4];
};

static inline int cmpN(const u64* a, const u64* b, int n) {
  for (int i = n - 1; i >= 0; --i) {
    if (a[i] < b[i]) return -1;
    if (a[i] > b[i]) return 1;
  }
  return 0;
}
static inline u64 addN(u64* r, const u64* a, const u64* b, int n) {
  u128 c = 0;
  for (int i = 0; i < n; ++i) {
    c += (u128)a[i] + b[i];
    r[i] = (u64)c;
    c >>= 64;
  }
  return (u64)c;
}
static inline u64 subN(u64* r, const u64* a, const u64* b, int n) {
  u64 borrow = 0;
  for (int i = 0; i < n; ++i) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (u64)d;
    borrow = (u64)(d >> 64) & 1;
  }
  return borrow;
}

// One instance per modulus; Tag supplies `static const int NL` (limbs: 4 or 6) and `static const u64 MOD[NL]`.
template <class Tag>
struct Fp {
  static const int N = Tag::NL;
  static const int BYTES = 8 * Tag::NL;   // canonical little-endian encoding
  u64 v[N];  // Montgomery form

  struct Consts {
    u64 inv;      // -MOD^{-1} mod 2^64
    u64 r1[Tag::NL];    // R mod MOD
    u64 r2[Tag::NL];    // R^2 mod MOD
    int bits;     // modulus bit length
    Consts() {
      const u64* M = Tag::MOD;
      u64 x = 1;  // Newton: x = MOD^{-1} mod 2^64
      for (int i = 0; i < 6; ++i) x *= 2 - M[0] * x;
      inv = (u64)0 - x;
      // r1 = 2^(64 N) mod M by 64 N modular doublings of 1
      u64 t[Tag::NL] = {1};
      auto dbl = [&](u64* a) {
        u64 c = addN(a, a, a, Tag::NL);
        if (c || cmpN(a, M, Tag::NL) >= 0) subN(a, a, M, Tag::NL);
      };
      for (int i = 0; i < 64 * Tag::NL; ++i) dbl(t);
      memcpy(r1, t, 8 * Tag::NL);
      for (int i = 0; i < 64 * Tag::NL; ++i) dbl(t);
      memcpy(r2, t, 8 * Tag::NL);
      bits = 0;
      for (int i = 64 * Tag::NL - 1; i >= 0; --i)
        if ((M[i / 64] >> (i % 64)) & 1) {
          bits = i + 1;
          break;
        }
    }
  };
  static const Consts& C() {
    static const Consts c;
    return c;
  }

  static const u64* modulus() { return Tag::MOD; }
  static Fp zero() {
    Fp r;
    memset(r.v, 0, BYTES);
    return r;
  }
  static Fp one() {
    Fp r;
    memcpy(r.v, C().r1, BYTES);
    return r;
  }
  bool is_zero() const {
    u64 o = 0;
    for (int i = 0; i < N; ++i) o |= v[i];
    return o == 0;
  }
  bool operator==(const Fp& o) const { return memcmp(v, o.v, BYTES) == 0; }
  bool operator!=(const Fp& o) const { return !(*this == o); }

  static Fp mont_mul(const Fp& a, const Fp& b) {
    const u64* M = Tag::MOD;
    const u64 inv = C().inv;
    u64 t[N + 2];
    memset(t, 0, sizeof(t));
    for (int i = 0; i < N; ++i) {
      u128 c = 0;
      for (int j = 0; j < N; ++j) {
        c += (u128)t[j] + (u128)a.v[j] * b.v[i];
        t[j] = (u64)c;
        c >>= 64;
      }
      c += t[N];
      t[N] = (u64)c;
      t[N + 1] = (u64)(c >> 64);
      u64 m = t[0] * inv;
      c = (u128)t[0] + (u128)m * M[0];
      c >>= 64;
      for (int j = 1; j < N; ++j) {
        c += (u128)t[j] + (u128)m * M[j];
        t[j - 1] = (u64)c;
        c >>= 64;
      }
      c += t[N];
      t[N - 1] = (u64)c;
      t[N] = t[N + 1] + (u64)(c >> 64);
    }
    Fp r;
    memcpy(r.v, t, BYTES);
    if (t[N] || cmpN(r.v, M, N) >= 0) subN(r.v, r.v, M, N);
    return r;
  }
  Fp operator*(const Fp& o) const { return mont_mul(*this, o); }
  Fp sqr() const { return mont_mul(*this, *this); }
  Fp operator+(const Fp& o) const {
    Fp r;
    u64 c = addN(r.v, v, o.v, N);
    if (c || cmpN(r.v, Tag::MOD, N) >= 0) subN(r.v, r.v, Tag::MOD, N);
    return r;
  }
  Fp operator-(const Fp& o) const {
    Fp r;
    if (subN(r.v, v, o.v, N)) addN(r.v, r.v, Tag::MOD, N);
    return r;
  }
  Fp neg() const { return is_zero() ? *this : zero() - *this; }
  Fp dbl() const { return *this + *this; }

  // canonical integer <-> Montgomery
  static Fp from_u256(const u64* a) {   // N limbs (the name dates from the 4-limb fields)
    Fp t, r2;
    memcpy(t.v, a, BYTES);
    memcpy(r2.v, C().r2, BYTES);
    return mont_mul(t, r2);
  }
  static Fp from_u64(u64 x) {
    u64 a[N] = {x};
    return from_u256(a);
  }
  void to_u256(u64* out) const {        // N limbs
    Fp o = zero();
    o.v[0] = 1;
    Fp r = mont_mul(*this, o);
    memcpy(out, r.v, BYTES);
  }
  // BYTES-byte little-endian canonical encoding; from_bytes requires value < MOD (returns false otherwise)
  void to_bytes(uint8_t* out) const {
    u64 a[N];
    to_u256(a);
    memcpy(out, a, BYTES);  // host is little-endian
  }
  static bool from_bytes(const uint8_t* in, Fp& out) {
    u64 a[N];
    memcpy(a, in, BYTES);
    if (cmpN(a, Tag::MOD, N) >= 0) return false;
    out = from_u256(a);
    return true;
  }

  Fp pow(const u64* e) const {            // N-limb exponent
    Fp acc = one();
    for (int i = 64 * N - 1; i >= 0; --i) {
      acc = acc.sqr();
      if ((e[i / 64] >> (i % 64)) & 1) acc = acc * *this;
    }
    return acc;
  }
  Fp inverse() const {  // Fermat; inverse of 0 is 0
    u64 e[N], two[N] = {2};
    subN(e, Tag::MOD, two, N);
    return pow(e);
  }
};

}  // namespace mpo
