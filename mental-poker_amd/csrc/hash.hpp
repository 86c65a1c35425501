// On-device Fiat-Shamir machinery: BLAKE2s-256 (RFC 7693), the ChaCha20 block function as used by
// rand_chacha::ChaCha20Rng (64-bit counter, stream 0) and ark-marlin 0.3 `FiatShamirRng<Blake2s>`
// semantics (seed = H(init); absorb: seed = H(new || seed); challenges = `Fp::rand` on the re-seeded
// ChaCha20 stream).  Replaces the host-side RNG the reference constructs at
// [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:408,436] so that a whole batch of
// proofs runs without a host round-trip per challenge.
// One lane = one proof.  Transcript bytes are first serialised into a per-proof staging buffer in HBM
// with a word-interleaved layout (word w of proof b at stage[w * stride + b]: lanes of a wave touch
// consecutive dwords), then compressed 64 bytes at a time with the message block held in registers.
#pragma once
#include <cstdint>

#include "field.hpp"

namespace mp {

MP_HD uint32_t rotr32(uint32_t x, int c) { return (x >> c) | (x << (32 - c)); }
MP_HD uint32_t rotl32(uint32_t x, int c) { return (x << c) | (x >> (32 - c)); }

struct Blake2sState {
  uint32_t h[8];
};

MP_HD void blake2s_init(Blake2sState& s) {
  const uint32_t iv[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
#pragma unroll
  for (int i = 0; i < 8; ++i) s.h[i] = iv[i];
  s.h[0] ^= 0x01010020u;
}

#define MP_B2S_G(a, b, c, d, x, y) \
  v[a] = v[a] + v[b] + (x);        \
  v[d] = rotr32(v[d] ^ v[a], 16);  \
  v[c] = v[c] + v[d];              \
  v[b] = rotr32(v[b] ^ v[c], 12);  \
  v[a] = v[a] + v[b] + (y);        \
  v[d] = rotr32(v[d] ^ v[a], 8);   \
  v[c] = v[c] + v[d];              \
  v[b] = rotr32(v[b] ^ v[c], 7);

#define MP_B2S_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
  MP_B2S_G(0, 4, 8, 12, m[s0], m[s1])                                                      \
  MP_B2S_G(1, 5, 9, 13, m[s2], m[s3])                                                      \
  MP_B2S_G(2, 6, 10, 14, m[s4], m[s5])                                                     \
  MP_B2S_G(3, 7, 11, 15, m[s6], m[s7])                                                     \
  MP_B2S_G(0, 5, 10, 15, m[s8], m[s9])                                                     \
  MP_B2S_G(1, 6, 11, 12, m[s10], m[s11])                                                   \
  MP_B2S_G(2, 7, 8, 13, m[s12], m[s13])                                                    \
  MP_B2S_G(3, 4, 9, 14, m[s14], m[s15])

// t = total bytes hashed including this block; last = final block flag
MP_HD void blake2s_compress(Blake2sState& s, const uint32_t m[16], uint64_t t, bool last) {
  const uint32_t iv[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
  uint32_t v[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = s.h[i];
    v[i + 8] = iv[i];
  }
  v[12] ^= (uint32_t)t;
  v[13] ^= (uint32_t)(t >> 32);
  if (last) v[14] = ~v[14];
  MP_B2S_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  MP_B2S_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
  MP_B2S_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
  MP_B2S_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
  MP_B2S_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
  MP_B2S_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
  MP_B2S_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
  MP_B2S_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
  MP_B2S_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
  MP_B2S_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
#pragma unroll
  for (int i = 0; i < 8; ++i) s.h[i] ^= v[i] ^ v[i + 8];
}

// Byte-stream writer into the word-interleaved staging buffer: a 64-bit shift register turns the
// 65-byte point encodings (x || y || flag) into aligned dword stores without dynamic register indexing.
struct StageWriter {
  uint32_t* base;   // stage + b
  uint32_t stride;  // dwords between consecutive words of one proof
  uint32_t widx;    // next word index
  uint64_t acc;     // pending bytes (low `nb` bytes valid)
  uint32_t nb;      // 0..3
};
MP_HD StageWriter stage_begin(uint32_t* stage, uint32_t stride, uint32_t b) {
  StageWriter w;
  w.base = stage + b;
  w.stride = stride;
  w.widx = 0;
  w.acc = 0;
  w.nb = 0;
  return w;
}
// a writer that starts at word `widx` of the proof's buffer (several lanes staging word-aligned pieces of one message)
MP_HD StageWriter stage_begin_at(uint32_t* stage, uint32_t stride, uint32_t b, uint32_t widx) {
  StageWriter w = stage_begin(stage, stride, b);
  w.widx = widx;
  return w;
}
MP_HD void stage_flush(StageWriter& w) {      // the partial last word, zero padded
  if (w.nb) w.base[(size_t)w.widx * w.stride] = (uint32_t)w.acc;
}
MP_HD void stage_word(StageWriter& w, uint32_t x) {
  w.acc |= (uint64_t)x << (8 * w.nb);
  w.base[(size_t)w.widx * w.stride] = (uint32_t)w.acc;
  w.widx++;
  w.acc >>= 32;
}
MP_HD void stage_byte(StageWriter& w, uint32_t x) {
  w.acc |= (uint64_t)(x & 0xFFu) << (8 * w.nb);
  w.nb++;
  if (w.nb == 4) {
    w.base[(size_t)w.widx * w.stride] = (uint32_t)w.acc;
    w.widx++;
    w.acc = 0;
    w.nb = 0;
  }
}
MP_HD uint32_t stage_len(const StageWriter& w) { return w.widx * 4u + w.nb; }

// Hash the staged bytes [0, len): pads the tail word(s) with zeros itself.
MP_HD void blake2s_staged(StageWriter& w, uint32_t out[8]) {
  const uint32_t len = stage_len(w);
  // flush the partial word (zero padded)
  if (w.nb) w.base[(size_t)w.widx * w.stride] = (uint32_t)w.acc;
  const uint32_t nwords = (len + 3u) / 4u;
  Blake2sState s;
  blake2s_init(s);
  uint32_t nblocks = (len + 63u) / 64u;
  if (nblocks == 0) nblocks = 1;
  for (uint32_t blk = 0; blk < nblocks; ++blk) {
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      uint32_t wi = blk * 16u + i;
      m[i] = wi < nwords ? w.base[(size_t)wi * w.stride] : 0u;
    }
    const bool last = blk + 1 == nblocks;
    const uint64_t t = last ? (uint64_t)len : (uint64_t)(blk + 1) * 64u;
    blake2s_compress(s, m, t, last);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = s.h[i];
}

// ---- four lanes on one BLAKE2s state -----------------------------------------------------------------------------------------
// The transcript of a proof is ONE hash chain: with a lane per proof a small batch waits ~5 us per 64-byte block (a lone wave issues
// one instruction every 8-10 cycles) and a batch of a few thousand 1024-card decks fills a quarter of the SIMDs with such lanes.
// The four G functions of a half-round are independent, so a QUAD (four adjacent lanes) keeps the 4 x 4 working matrix one column
// per lane: the column step is lane-local, the diagonal step is the same code after rotating rows b, c, d by 1, 2, 3 lanes inside
// the quad (one DPP move each, WaveCtx::quad_rot) -- ~42 instructions per round instead of ~112.  Every lane of the quad holds the
// whole message block (16 redundant loads, prefetched one block ahead) and picks its two words of a half-round by lane number.
// Written against the wave interface of rt.hpp, so the development emulator runs the same source.
struct B2sBlock {
  uint32_t w[16];
};
struct B2sSeed {
  uint32_t s[8];
};
MP_HD uint32_t sel4(uint32_t j, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
  const uint32_t lo = (j & 1u) ? w1 : w0, hi = (j & 1u) ? w3 : w2;
  return (j & 2u) ? hi : lo;
}
#define MP_B2Q_HALF(x, y)                                                             \
  wv.lanes([&](uint32_t l) {                                                          \
    const uint32_t j = l & 3u;                                                        \
    const B2sBlock& q = m[l];                                                         \
    uint32_t a = va[l], b = vb[l], c = vc[l], d = vd[l];                              \
    a = a + b + (x);                                                                  \
    d = rotr32(d ^ a, 16);                                                            \
    c = c + d;                                                                        \
    b = rotr32(b ^ c, 12);                                                            \
    a = a + b + (y);                                                                  \
    d = rotr32(d ^ a, 8);                                                             \
    c = c + d;                                                                        \
    b = rotr32(b ^ c, 7);                                                             \
    va[l] = a; vb[l] = b; vc[l] = c; vd[l] = d;                                       \
  });
#define MP_B2Q_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                                   \
  MP_B2Q_HALF(sel4(j, q.w[s0], q.w[s2], q.w[s4], q.w[s6]), sel4(j, q.w[s1], q.w[s3], q.w[s5], q.w[s7]))                     \
  wv.template quad_rot<1>(vb); wv.template quad_rot<2>(vc); wv.template quad_rot<3>(vd);                                    \
  MP_B2Q_HALF(sel4(j, q.w[s8], q.w[s10], q.w[s12], q.w[s14]), sel4(j, q.w[s9], q.w[s11], q.w[s13], q.w[s15]))               \
  wv.template quad_rot<3>(vb); wv.template quad_rot<2>(vc); wv.template quad_rot<1>(vd);
// state: lane j of the quad holds h[j] in ha and h[4 + j] in hb; t, last as in blake2s_compress (the same for every lane)
template <class W>
MP_HD void blake2s_compress_quad(W& wv, PerLane<uint32_t>& ha, PerLane<uint32_t>& hb, const PerLane<B2sBlock>& m, uint64_t t, bool last) {
  PerLane<uint32_t> va, vb, vc, vd;
  wv.lanes([&](uint32_t l) {
    const uint32_t j = l & 3u;
    va[l] = ha[l];
    vb[l] = hb[l];
    vc[l] = sel4(j, 0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au);
    vd[l] = sel4(j, 0x510E527Fu ^ (uint32_t)t, 0x9B05688Cu ^ (uint32_t)(t >> 32), last ? ~0x1F83D9ABu : 0x1F83D9ABu, 0x5BE0CD19u);
  });
  MP_B2Q_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  MP_B2Q_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
  MP_B2Q_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
  MP_B2Q_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
  MP_B2Q_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
  MP_B2Q_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
  MP_B2Q_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
  MP_B2Q_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
  MP_B2Q_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
  MP_B2Q_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
  wv.lanes([&](uint32_t l) {
    ha[l] ^= va[l] ^ vc[l];
    hb[l] ^= vb[l] ^ vd[l];
  });
}
// BLAKE2s-256 of the staged bytes [0, len) of each lane's proof: word w at base[l][w * stride] (StageWriter layout), zero-padded by
// the writer up to the next word.  len is the same for every lane.  Every lane of a quad ends up with the whole digest.
template <class W>
MP_HD void blake2s_staged_quad(W& wv, const PerLane<const uint32_t*>& base, uint32_t stride, uint32_t len, PerLane<B2sSeed>& out) {
  const uint32_t nwords = (len + 3u) / 4u;
  uint32_t nblocks = (len + 63u) / 64u;
  if (nblocks == 0) nblocks = 1;
  PerLane<uint32_t> ha, hb;
  wv.lanes([&](uint32_t l) {
    const uint32_t j = l & 3u;
    ha[l] = sel4(j, 0x6A09E667u ^ 0x01010020u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au);
    hb[l] = sel4(j, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u);
  });
  PerLane<B2sBlock> m, mn;
  auto load = [&](PerLane<B2sBlock>& dst, uint32_t blk) {
    wv.lanes([&](uint32_t l) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t wi = blk * 16u + i;
        dst[l].w[i] = wi < nwords ? base[l][(size_t)wi * stride] : 0u;
      }
    });
  };
  load(m, 0);
#pragma unroll 1
  for (uint32_t blk = 0; blk < nblocks; ++blk) {
    const bool last = blk + 1 == nblocks;
    if (!last) load(mn, blk + 1);            // one block ahead: the loads fly while this block is compressed
    blake2s_compress_quad(wv, ha, hb, m, last ? (uint64_t)len : (uint64_t)(blk + 1) * 64u, last);
    if (!last) wv.lanes([&](uint32_t l) { m[l] = mn[l]; });
  }
  PerLane<uint32_t> t;
  wv.template quad_bcast<0>(ha, t); wv.lanes([&](uint32_t l) { out[l].s[0] = t[l]; });
  wv.template quad_bcast<1>(ha, t); wv.lanes([&](uint32_t l) { out[l].s[1] = t[l]; });
  wv.template quad_bcast<2>(ha, t); wv.lanes([&](uint32_t l) { out[l].s[2] = t[l]; });
  wv.template quad_bcast<3>(ha, t); wv.lanes([&](uint32_t l) { out[l].s[3] = t[l]; });
  wv.template quad_bcast<0>(hb, t); wv.lanes([&](uint32_t l) { out[l].s[4] = t[l]; });
  wv.template quad_bcast<1>(hb, t); wv.lanes([&](uint32_t l) { out[l].s[5] = t[l]; });
  wv.template quad_bcast<2>(hb, t); wv.lanes([&](uint32_t l) { out[l].s[6] = t[l]; });
  wv.template quad_bcast<3>(hb, t); wv.lanes([&](uint32_t l) { out[l].s[7] = t[l]; });
}

MP_HD void chacha20_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
  uint32_t x[16];
  x[0] = 0x61707865u; x[1] = 0x3320646Eu; x[2] = 0x79622D32u; x[3] = 0x6B206574u;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[4 + i] = key[i];
  x[12] = (uint32_t)counter;
  x[13] = (uint32_t)(counter >> 32);
  x[14] = 0;
  x[15] = 0;
#define MP_QR(a, b, c, d)                          \
  x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16);    \
  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);    \
  x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);     \
  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
  for (int r = 0; r < 10; ++r) {
    MP_QR(0, 4, 8, 12) MP_QR(1, 5, 9, 13) MP_QR(2, 6, 10, 14) MP_QR(3, 7, 11, 15)
    MP_QR(0, 5, 10, 15) MP_QR(1, 6, 11, 12) MP_QR(2, 7, 8, 13) MP_QR(3, 4, 9, 14)
  }
#undef MP_QR
  out[0] = x[0] + 0x61707865u; out[1] = x[1] + 0x3320646Eu; out[2] = x[2] + 0x79622D32u; out[3] = x[3] + 0x6B206574u;
#pragma unroll
  for (int i = 0; i < 8; ++i) out[4 + i] = x[4 + i] + key[i];
  out[12] = x[12] + (uint32_t)counter;
  out[13] = x[13] + (uint32_t)(counter >> 32);
  out[14] = x[14];
  out[15] = x[15];
}

// `Fp::rand` over a ChaCha20 stream (arkworks 0.3): candidates are 8 consecutive stream words (= 4 u64,
// limb 0 first); the top 256-BITS bits are cleared; accepted iff < modulus; the accepted limbs ARE the
// Montgomery representation.  A 64-byte block holds exactly two candidates, so the stream position is
// (block counter, half).
struct FrStream {
  uint32_t key[8];
  uint32_t blk[16];
  uint64_t counter;  // next block to generate
  uint32_t half;     // 0,1 = next candidate inside blk; 2 = blk exhausted
};
MP_HD void frstream_init(FrStream& s, const uint32_t key[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) s.key[i] = key[i];
  s.counter = 0;
  s.half = 2;
}
// one candidate: true (and the value) if it is accepted
template <class P>
MP_HD bool frstream_try(FrStream& s, Fe<P>& f) {
  if (s.half >= 2) {
    chacha20_block(s.key, s.counter, s.blk);
    s.counter++;
    s.half = 0;
  }
  if (s.half == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f.v[i] = s.blk[i];
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) f.v[i] = s.blk[8 + i];
  }
  s.half++;
  if (P::BITS < 256) f.v[7] &= 0xFFFFFFFFu >> (256 - P::BITS);
  return fe_canonical_in_range<P>(f.v);
}
template <class P>
MP_HD Fe<P> frstream_next(FrStream& s) {
  for (;;) {
    if (s.half >= 2) {
      chacha20_block(s.key, s.counter, s.blk);
      s.counter++;
      s.half = 0;
    }
    Fe<P> f;
    if (s.half == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) f.v[i] = s.blk[i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) f.v[i] = s.blk[8 + i];
    }
    s.half++;
    if (P::BITS < 256) f.v[7] &= 0xFFFFFFFFu >> (256 - P::BITS);
    if (fe_canonical_in_range<P>(f.v)) return f;
  }
}

}  // namespace mp
