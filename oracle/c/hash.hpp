// TEST INFRASTRUCTURE ONLY (oracle).  BLAKE2s-256 (RFC 7693), ChaCha20 block function as used by
// rand_chacha::ChaCha20Rng (64-bit block counter, stream 0) and ark-marlin 0.3's
// `FiatShamirRng<Blake2s>` built from them -- the RNG the reference seeds with b"Shuffle Proof"
// [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:84,408,436].  These crates are
// dependencies (Cargo.toml:13,15), not under /root/reference; restated from their published algorithms
// (SURVEY.md App. B).  KATs: RFC 7539 2.3.2, Blake2s("Shuffle Proof").
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace mpo {

struct Blake2s {
  uint32_t h[8];
  uint64_t t;
  uint8_t buf[64];
  size_t buflen;

  static uint32_t rotr(uint32_t x, int c) { return (x >> c) | (x << (32 - c)); }
  static const uint32_t* IV() {
    static const uint32_t iv[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A,
                                   0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
    return iv;
  }
  Blake2s() {
    for (int i = 0; i < 8; ++i) h[i] = IV()[i];
    h[0] ^= 0x01010020;  // digest length 32, no key, fanout = depth = 1
    t = 0;
    buflen = 0;
  }
  void compress(const uint8_t* block, bool last) {
    static const uint8_t S[10][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
    uint32_t m[16], v[16];
    for (int i = 0; i < 16; ++i) memcpy(&m[i], block + 4 * i, 4);
    for (int i = 0; i < 8; ++i) {
      v[i] = h[i];
      v[i + 8] = IV()[i];
    }
    v[12] ^= (uint32_t)t;
    v[13] ^= (uint32_t)(t >> 32);
    if (last) v[14] = ~v[14];
#define MPO_G(a, b, c, d, x, y)      \
  v[a] = v[a] + v[b] + (x);          \
  v[d] = rotr(v[d] ^ v[a], 16);      \
  v[c] = v[c] + v[d];                \
  v[b] = rotr(v[b] ^ v[c], 12);      \
  v[a] = v[a] + v[b] + (y);          \
  v[d] = rotr(v[d] ^ v[a], 8);       \
  v[c] = v[c] + v[d];                \
  v[b] = rotr(v[b] ^ v[c], 7);
    for (int r = 0; r < 10; ++r) {
      const uint8_t* s = S[r];
      MPO_G(0, 4, 8, 12, m[s[0]], m[s[1]]);
      MPO_G(1, 5, 9, 13, m[s[2]], m[s[3]]);
      MPO_G(2, 6, 10, 14, m[s[4]], m[s[5]]);
      MPO_G(3, 7, 11, 15, m[s[6]], m[s[7]]);
      MPO_G(0, 5, 10, 15, m[s[8]], m[s[9]]);
      MPO_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
      MPO_G(2, 7, 8, 13, m[s[12]], m[s[13]]);
      MPO_G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef MPO_G
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
  }
  void update(const uint8_t* in, size_t len) {
    while (len) {
      if (buflen == 64) {  // buffer full and more input follows: not the last block
        t += 64;
        compress(buf, false);
        buflen = 0;
      }
      size_t take = 64 - buflen;
      if (take > len) take = len;
      memcpy(buf + buflen, in, take);
      buflen += take;
      in += take;
      len -= take;
    }
  }
  void final(uint8_t out[32]) {
    t += buflen;
    memset(buf + buflen, 0, 64 - buflen);
    compress(buf, true);
    memcpy(out, h, 32);  // little-endian host
  }
  static void digest(const uint8_t* in, size_t len, uint8_t out[32]) {
    Blake2s b;
    b.update(in, len);
    b.final(out);
  }
};

static inline void chacha20_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
  uint32_t st[16] = {0x61707865, 0x3320646E, 0x79622D32, 0x6B206574};
  for (int i = 0; i < 8; ++i) st[4 + i] = key[i];
  st[12] = (uint32_t)counter;
  st[13] = (uint32_t)(counter >> 32);
  st[14] = 0;
  st[15] = 0;
  uint32_t x[16];
  memcpy(x, st, 64);
  auto rotl = [](uint32_t v, int c) { return (v << c) | (v >> (32 - c)); };
#define MPO_QR(a, b, c, d)                   \
  x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16); \
  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12); \
  x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);  \
  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
  for (int r = 0; r < 10; ++r) {
    MPO_QR(0, 4, 8, 12) MPO_QR(1, 5, 9, 13) MPO_QR(2, 6, 10, 14) MPO_QR(3, 7, 11, 15)
    MPO_QR(0, 5, 10, 15) MPO_QR(1, 6, 11, 12) MPO_QR(2, 7, 8, 13) MPO_QR(3, 4, 9, 14)
  }
#undef MPO_QR
  for (int i = 0; i < 16; ++i) out[i] = x[i] + st[i];
}

// ChaCha20Rng::from_seed; only u64 draws are ever made, so the stream is consecutive word pairs.
struct ChaChaRng {
  uint32_t key[8];
  uint64_t counter;
  uint32_t buf[16];
  int pos;
  ChaChaRng() : counter(0), pos(16) { memset(key, 0, 32); }
  explicit ChaChaRng(const uint8_t seed[32]) : counter(0), pos(16) { memcpy(key, seed, 32); }
  uint32_t next_u32() {
    if (pos == 16) {
      chacha20_block(key, counter++, buf);
      pos = 0;
    }
    return buf[pos++];
  }
  uint64_t next_u64() {
    uint64_t lo = next_u32();
    uint64_t hi = next_u32();
    return lo | (hi << 32);
  }
};

// ark_marlin::rng::FiatShamirRng<Blake2s> (0.3): seed = H(init); absorb: seed = H(new || seed).
struct FsRng {
  uint8_t seed[32];
  ChaChaRng r;
  FsRng(const uint8_t* init, size_t len) {
    Blake2s::digest(init, len, seed);
    r = ChaChaRng(seed);
  }
  void absorb(const std::vector<uint8_t>& data) {
    Blake2s b;
    b.update(data.data(), data.size());
    b.update(seed, 32);
    b.final(seed);
    r = ChaChaRng(seed);
  }
  uint64_t next_u64() { return r.next_u64(); }
};

// arkworks-0.3 `Fp::rand`: 4 u64 limbs, top REPR_SHAVE_BITS of limb 3 cleared, accept if < modulus;
// the accepted limbs are the Montgomery representation.
template <class F, class Rng>
F field_rand(Rng& rng) {
  const int shave = 256 - F::C().bits;
  for (;;) {
    F f;
    for (int i = 0; i < 4; ++i) f.v[i] = rng.next_u64();
    if (shave) f.v[3] &= (~(uint64_t)0) >> shave;
    if (cmpN(f.v, F::modulus(), 4) < 0) return f;   // scalar fields only (4 limbs)
  }
}

}  // namespace mpo
