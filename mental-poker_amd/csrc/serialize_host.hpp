// serialize_host.hpp -- arkworks-0.3 `CanonicalSerialize` / `CanonicalDeserialize` (compressed) for the associated types of
// the trait: every one of them must implement both [REF barnett-smart-card-protocol/src/lib.rs:45-71], and the reference's
// harness measures `proof.serialized_size()` [REF examples/parameter_selection.rs:95].  This is the part of the boundary a Rust
// caller holding serialised values binds (include/mpshuffle.h "canonical serialisation"); the engine's own batch entry points stay on
// the fixed-width wire v1 (x || y), because decompression costs a square root per point (Tonelli-Shanks with 2-adicity 192 on the
// STARK prime: ~10^4 squarings) -- more than proving and verifying the proof the point belongs to.
//   Fr      32 B little-endian canonical, must be < q
//   point   x little-endian in ceil((bits + 2) / 8) bytes; flags in the top bits of the last byte: bit 7 = (y > -y), bit 6 = infinity
//   Vec<T>  u64 little-endian length + elements; usize = u64
// Deserialisation validates like ark-ec: canonical x, x on the curve, subgroup membership on curves with a cofactor.
// Host code (same field / curve routines as the kernels); points of a call are spread over the host's hardware threads.
#pragma once
#include <thread>
#include <vector>

#include "setup_host.hpp"

namespace mp {

template <class C>
struct Ser {
  typedef typename C::FqP F;
  static constexpr uint32_t FB = 4 * F::NW;                    // wire bytes of a coordinate
  static constexpr uint32_t PB = 8 * F::NW;                    // wire bytes of a point
  static constexpr uint32_t CB = (F::BITS + 2 + 7) / 8;        // compressed bytes of a point

  static bool words_lt_mod(const uint32_t* w) { return fe_canonical_in_range<F>(w); }

  // wire -> compressed; false if a coordinate is out of range
  static bool compress_one(const uint8_t* wire, uint8_t* out) {
    uint32_t xw[F::NW + 1], yw[F::NW];
    memcpy(xw, wire, FB);
    memcpy(yw, wire + FB, FB);
    xw[F::NW] = 0;
    bool zero = true;
    for (uint32_t i = 0; i < F::NW; ++i) zero = zero && xw[i] == 0 && yw[i] == 0;
    memset(out, 0, CB);
    if (zero) {
      out[CB - 1] |= 0x40;
      return true;
    }
    if (!words_lt_mod(xw) || !words_lt_mod(yw)) return false;
    memcpy(out, xw, CB <= FB ? CB : FB);
    // y > p - y  <=>  2 y > p  (y < p)
    uint32_t ny[F::NW];
    uint64_t br = 0;
    for (uint32_t i = 0; i < F::NW; ++i) {
      const uint64_t t = (uint64_t)F::MOD[i] - yw[i] - br;
      ny[i] = (uint32_t)t;
      br = (t >> 32) & 1;
    }
    bool greater = false;
    for (int i = F::NW - 1; i >= 0; --i)
      if (yw[i] != ny[i]) {
        greater = yw[i] > ny[i];
        break;
      }
    if (greater) out[CB - 1] |= 0x80;
    return true;
  }
  // compressed -> wire; false on a non-canonical encoding, an x that is not on the curve or a point outside the subgroup
  static bool decompress_one(const uint8_t* in, uint8_t* wire, const Fe<F>& bmont) {
    uint8_t buf[4 * (F::NW + 1)];
    memset(buf, 0, sizeof(buf));
    memcpy(buf, in, CB);
    const uint8_t flags = buf[CB - 1] & 0xC0;
    buf[CB - 1] &= 0x3F;
    uint32_t xw[F::NW + 1];
    memcpy(xw, buf, 4 * (F::NW + 1));
    if (xw[F::NW] != 0) return false;                          // (33-byte secp256k1 encoding: the spare byte carries only flags)
    if (flags & 0x40) {
      bool zero = (flags & 0x80) == 0;
      for (uint32_t i = 0; i < F::NW; ++i) zero = zero && xw[i] == 0;
      if (!zero) return false;
      memset(wire, 0, PB);
      return true;
    }
    if (!words_lt_mod(xw)) return false;
    const Fe<F> x = fe_from_canonical<F>(xw);
    Fe<F> rhs = fe_add<F>(fe_mul<F>(fe_sqr<F>(x), x), bmont);
    if (C::A == 1) rhs = fe_add<F>(rhs, x);
    Fe<F> y;
    if (!fe_sqrt_host<F>(rhs, y)) return false;
    uint32_t yw[F::NW], nyw[F::NW];
    fe_to_canonical<F>(y, yw);
    fe_to_canonical<F>(fe_neg<F>(y), nyw);
    bool greater = false;
    for (int i = F::NW - 1; i >= 0; --i)
      if (yw[i] != nyw[i]) {
        greater = yw[i] > nyw[i];
        break;
      }
    if (greater != ((flags & 0x80) != 0)) memcpy(yw, nyw, sizeof(yw));
    if (!Cofactor<C>::ONE) {
      Aff<C> p;
      p.x = x;
      p.y = fe_from_canonical<F>(yw);
      if (!aff_in_subgroup_host<C>(p)) return false;
    }
    memcpy(wire, xw, FB);
    memcpy(wire + FB, yw, FB);
    return true;
  }
  // `count` points; returns the index of the first bad one or -1
  static long points(bool de, size_t count, const uint8_t* in, uint8_t* out) {
    const Fe<F> b = fe_unpack<F>(C::B_MONT);
    const size_t in_sz = de ? CB : PB, out_sz = de ? PB : CB;
    unsigned nt = std::thread::hardware_concurrency();
    nt = (!de || count < 16 || nt < 2) ? 1 : (nt > 32 ? 32 : nt);
    std::vector<long> bad(nt, -1);
    auto work = [&](unsigned t) {
      for (size_t i = t; i < count; i += nt) {
        const bool ok = de ? decompress_one(in + i * in_sz, out + i * out_sz, b) : compress_one(in + i * in_sz, out + i * out_sz);
        if (!ok && (bad[t] < 0 || (long)i < bad[t])) bad[t] = (long)i;
      }
    };
    if (nt == 1) {
      work(0);
    } else {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, t);
      for (auto& x : th) x.join();
    }
    long first = -1;
    for (long v : bad)
      if (v >= 0 && (first < 0 || v < first)) first = v;
    return first;
  }
  // are `count` 32-byte little-endian scalars canonical (< q)?
  static bool scalars_ok(size_t count, const uint8_t* in) {
    for (size_t i = 0; i < count; ++i) {
      uint32_t w[8];
      memcpy(w, in + 32 * i, 32);
      if (!fe_canonical_in_range<typename C::FrP>(w)) return false;
    }
    return true;
  }
};

}  // namespace mp
