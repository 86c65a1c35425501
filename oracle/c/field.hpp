// TEST INFRASTRUCTURE ONLY (oracle).  4x64-bit-limb Montgomery prime field, the representation
// ark-ff 0.3 `Fp256` uses ("Fp repr: 4xu64 little-endian limbs in Montgomery form,
// R = 2^256").  The arithmetic behind every reference call on the hot path
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

static inline int cmp256(const u64* a, const u64* b) {
  for (int i = 3; i >= 0; --i) {
    if (a[i] < b[i]) return -1;
    if (a[i] > b[i]) return 1;
  }
  return 0;
}
static inline u64 add256(u64* r, const u64* a, const u64* b) {
  u128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (u128)a[i] + b[i];
    r[i] = (u64)c;
    c >>= 64;
  }
  return (u64)c;
}
static inline u64 sub256(u64* r, const u64* a, const u64* b) {
  u64 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (u64)d;
    borrow = (u64)(d >> 64) & 1;
  }
  return borrow;
}

// One instance per modulus; Tag supplies `static const u64 MOD[4]`.
template <class Tag>
struct Fp {
  u64 v[4];  // Montgomery form

  struct Consts {
    u64 inv;      // -MOD^{-1} mod 2^64
    u64 r1[4];    // R mod MOD
    u64 r2[4];    // R^2 mod MOD
    int bits;     // modulus bit length
    Consts() {
      const u64* M = Tag::MOD;
      u64 x = 1;  // Newton: x = MOD^{-1} mod 2^64
      for (int i = 0; i < 6; ++i) x *= 2 - M[0] * x;
      inv = (u64)0 - x;
      // r1 = 2^256 mod M by 256 modular doublings of 1
      u64 t[4] = {1, 0, 0, 0};
      auto dbl = [&](u64* a) {
        u64 c = add256(a, a, a);
        if (c || cmp256(a, M) >= 0) sub256(a, a, M);
      };
      for (int i = 0; i < 256; ++i) dbl(t);
      memcpy(r1, t, 32);
      for (int i = 0; i < 256; ++i) dbl(t);
      memcpy(r2, t, 32);
      bits = 0;
      for (int i = 255; i >= 0; --i)
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
    memset(r.v, 0, 32);
    return r;
  }
  static Fp one() {
    Fp r;
    memcpy(r.v, C().r1, 32);
    return r;
  }
  bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
  bool operator==(const Fp& o) const { return memcmp(v, o.v, 32) == 0; }
  bool operator!=(const Fp& o) const { return !(*this == o); }

  static Fp mont_mul(const Fp& a, const Fp& b) {
    const u64* M = Tag::MOD;
    const u64 inv = C().inv;
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
      u128 c = 0;
      for (int j = 0; j < 4; ++j) {
        c += (u128)t[j] + (u128)a.v[j] * b.v[i];
        t[j] = (u64)c;
        c >>= 64;
      }
      c += t[4];
      t[4] = (u64)c;
      t[5] = (u64)(c >> 64);
      u64 m = t[0] * inv;
      c = (u128)t[0] + (u128)m * M[0];
      c >>= 64;
      for (int j = 1; j < 4; ++j) {
        c += (u128)t[j] + (u128)m * M[j];
        t[j - 1] = (u64)c;
        c >>= 64;
      }
      c += t[4];
      t[3] = (u64)c;
      t[4] = t[5] + (u64)(c >> 64);
    }
    Fp r;
    memcpy(r.v, t, 32);
    if (t[4] || cmp256(r.v, M) >= 0) sub256(r.v, r.v, M);
    return r;
  }
  Fp operator*(const Fp& o) const { return mont_mul(*this, o); }
  Fp sqr() const { return mont_mul(*this, *this); }
  Fp operator+(const Fp& o) const {
    Fp r;
    u64 c = add256(r.v, v, o.v);
    if (c || cmp256(r.v, Tag::MOD) >= 0) sub256(r.v, r.v, Tag::MOD);
    return r;
  }
  Fp operator-(const Fp& o) const {
    Fp r;
    if (sub256(r.v, v, o.v)) add256(r.v, r.v, Tag::MOD);
    return r;
  }
  Fp neg() const { return is_zero() ? *this : zero() - *this; }
  Fp dbl() const { return *this + *this; }

  // canonical integer <-> Montgomery
  static Fp from_u256(const u64 a[4]) {
    Fp t, r2;
    memcpy(t.v, a, 32);
    memcpy(r2.v, C().r2, 32);
    return mont_mul(t, r2);
  }
  static Fp from_u64(u64 x) {
    u64 a[4] = {x, 0, 0, 0};
    return from_u256(a);
  }
  void to_u256(u64 out[4]) const {
    Fp o;
    o.v[0] = 1;
    o.v[1] = o.v[2] = o.v[3] = 0;
    Fp r = mont_mul(*this, o);
    memcpy(out, r.v, 32);
  }
  // 32-byte little-endian canonical encoding; from_bytes requires value < MOD (returns false otherwise)
  void to_bytes(uint8_t out[32]) const {
    u64 a[4];
    to_u256(a);
    memcpy(out, a, 32);  // host is little-endian
  }
  static bool from_bytes(const uint8_t in[32], Fp& out) {
    u64 a[4];
    memcpy(a, in, 32);
    if (cmp256(a, Tag::MOD) >= 0) return false;
    out = from_u256(a);
    return true;
  }

  Fp pow(const u64 e[4]) const {
    Fp acc = one();
    for (int i = 255; i >= 0; --i) {
      acc = acc.sqr();
      if ((e[i / 64] >> (i % 64)) & 1) acc = acc * *this;
    }
    return acc;
  }
  Fp inverse() const {  // Fermat; inverse of 0 is 0
    u64 e[4], two[4] = {2, 0, 0, 0};
    sub256(e, Tag::MOD, two);
    return pow(e);
  }
};

}  // namespace mpo
