// Bucket-method (Pippenger) multi-scalar multiplication for the LARGE per-proof MSMs -- the verifier's products over a
// 1024-card deck (4N + 11m + 9 = 4193 terms in one equation at m = 8, n = 128) -- as a WAVE-COOPERATIVE kernel: one 64-lane
// wave owns one (proof, MSM, window) and computes sum_t d_t P_t for the window's signed 8-bit digits d_t:
//
//   A  the window's K digits are read as words of four from the proof-major digit array (twice: B and D; 4 KB that stay in cache);
//   B  histogram of |d| over the 128 buckets (LDS atomics);
//   C  bucket offsets: an exclusive wavefront prefix sum over the pair counts {2l+1, 2l+2} gives the offsets of a counting sort;
//   D  scatter: term indices (sign in bit 15) sorted by bucket into LDS;
//   E  accumulation: every lane sums two buckets, run += +-P_t (XYZZ + affine, 8M+2S; the points come from the P arena, once
//      per window and L2-resident across the windows of a proof).  WHICH two is a matter of balance -- the wave waits for its
//      slowest lane: the r-th fullest odd bucket goes with the r-th emptiest even one (ranks by counting, round 4), so the lanes'
//      shares differ by a term or two; a window whose digits crowd into a few buckets (the top window of a 252-bit scalar has 8)
//      is cut into equal shares of the sorted list instead;
//   F  bucket reduction across the wave: the sums travel through LDS to the lanes that own the pairs {2l+1, 2l+2}; there
//      (2l+2) S_{2l+2} + (2l+1) S_{2l+1} = (2l+1) run_l + acc_l with run_l = S_{2l+2} + S_{2l+1}, acc_l = S_{2l+2} (one addition, NO
//      multiplication by a bucket number, no bucket array in memory) and
//      sum_l [(2l+1) run_l + acc_l] = sum_l acc_l + Suf_0 + 2 sum_{l>=1} Suf_l with the inclusive suffix sums
//      Suf_l = sum_{l'>=l} run_l' -- a 6-step wavefront suffix scan and a 6-step tree reduction of XYZZ points exchanged through
//      LDS (15 point additions per window instead of the 2 x 128 of the serial running sum).
//
// The W window results of an MSM are folded (8 doublings + 1 addition per window) by k_bucket_fold; its output slot joins
// the MSM's other partial sums in k_combine exactly like a Straus sub-job's.  Per term and window this is one mixed
// addition plus (load imbalance + 15 wave-wide additions) / terms-per-lane -- 33 windows instead of Straus' 51 and no
// per-base window tables at all (no k_table work, no 1 KB of table per base): it wins from ~2 000 terms per MSM on
// (DESIGN.md "bucket MSM"); below that the Straus kernel (kernels_msm.hpp) stays.
// Replaces ark-ec 0.3 `VariableBaseMSM::multi_scalar_mul` (the bucket method, sequential on the CPU) inside the
// reference's verifier [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:437-442].
#pragma once
#include "kernels_msm.hpp"

namespace mp {

static const int BK_BITS = 8;                    // signed windows: digits in [-128, 127]
static const uint32_t BK_BUCKETS = 128;          // |d| = 1 .. 128: two buckets per lane
static const uint32_t BK_HDR = 272;              // LDS words in front of the sort arrays: counts[132] + cursors[132] (+ pad)
static inline uint32_t bk_windows(int scalar_bits) { return (uint32_t)(scalar_bits + BK_BITS) / BK_BITS; }
// LDS words one wave needs for an MSM of kpad (multiple of 64) terms on a curve whose XYZZ point is xw words
static inline uint32_t bk_lds_words(uint32_t kpad, uint32_t xw) { return BK_HDR + std::max(kpad / 2u, 64u * xw); }

// ---- digits: canonical scalar -> W signed bytes, d_w in [-128, 127] (top window non-negative), proof-major:
// D8[b * dstride + pos + w * kpad]  (pos = digit offset of the term inside the proof's block, kpad = padded terms of its MSM)
struct BRecodeArgs {
  const uint32_t* S;
  int8_t* D8;
  const Term* bterms;      // {S slot, P slot}
  const BTermPos* bpos;
  uint32_t Bpad, nwin, nterms;
  size_t dstride;
};
template <class C>
MP_HD void body_bucket_recode(const BRecodeArgs& a, uint32_t xx, uint32_t) {
  typedef typename C::FrP R;
  const uint32_t x = xx % a.nterms, b = xx / a.nterms;
  const Term t = a.bterms[x];
  const BTermPos ps = a.bpos[x];
  uint32_t k[9];
  fe_to_canonical<R>(ld_fe<R>(a.S + s_off(t.s, a.Bpad, b)), k);
  k[8] = 0;
  int8_t* out = a.D8 + (size_t)b * a.dstride + ps.pos;
  uint32_t carry = 0;
  for (uint32_t w = 0; w < a.nwin; ++w) {
    uint32_t raw = ((k[w >> 2] >> (8 * (w & 3))) & 0xFFu) + carry;
    carry = 0;
    if (w + 1 < a.nwin && raw >= 128u) {
      raw -= 256u;
      carry = 1;
    }
    out[(size_t)w * ps.kpad] = (int8_t)(int32_t)raw;
  }
}
// thread = proof * nterms + bucket term of the phase (the term is the fast axis: the digit stores of a wave are contiguous)
MP_KERNEL(k_bucket_recode, BRecodeArgs, body_bucket_recode)

// ---- the bucket kernel ----------------------------------------------------------------------------------------
struct BucketArgs {
  const int8_t* D8;
  const uint32_t* P;       // affine points, slot-major arena
  uint32_t* J;
  const BJob* jobs;
  const Term* bterms;
  uint32_t Bpad, nwin, njobs;
  size_t dstride;
  uint32_t link_stride;    // chain verification: term.b = P slot | link << 20, the point lives in lane b + link * link_stride
};
static const uint32_t BK_SLOT_MASK = 0xFFFFFu;
template <class C>
MP_HD void xyzz_to_words(const Xyzz<C>& p, uint32_t* w) {
  constexpr int L = sizeof(p.X.v) / 4;
#pragma unroll
  for (int i = 0; i < L; ++i) {
    w[i] = p.X.v[i];
    w[L + i] = p.Y.v[i];
    w[2 * L + i] = p.ZZ.v[i];
    w[3 * L + i] = p.ZZZ.v[i];
  }
}
template <class C>
MP_HD Xyzz<C> xyzz_from_words(const uint32_t* w) {
  Xyzz<C> p;
  constexpr int L = sizeof(p.X.v) / 4;
#pragma unroll
  for (int i = 0; i < L; ++i) {
    p.X.v[i] = w[i];
    p.Y.v[i] = w[L + i];
    p.ZZ.v[i] = w[2 * L + i];
    p.ZZZ.v[i] = w[3 * L + i];
  }
  return p;
}
template <class C>
struct XyzzWords {
  static constexpr uint32_t N = 4 * sizeof(Fe<typename C::FqP>) / 4;
};

template <class C, class W>
MP_HD void body_bucket_msm(const BucketArgs& a, uint32_t wid, W& wv) {
  constexpr uint32_t XW = XyzzWords<C>::N;
  const uint32_t w = wid % a.nwin, jb = (wid / a.nwin) % a.njobs, b = wid / (a.nwin * a.njobs);
  const BJob job = a.jobs[jb];
  const uint32_t K = job.count, kpad = job.kpad;
  uint32_t* cnt = wv.lds;                         // [0 .. 128]: terms per |digit|
  uint32_t* cur = wv.lds + 132;                   // scatter cursors; after the scatter: the lanes' bucket assignment
  uint16_t* ix = reinterpret_cast<uint16_t*>(wv.lds + BK_HDR);              // sorted term indices | sign << 15
  uint32_t* xch = wv.lds + BK_HDR;                                            // point exchange (after the sort arrays are dead)
  // the window's digits: K bytes (kpad a multiple of 64), read twice as words of four -- 4 KB that stay in the L1/L2
  const uint32_t* src = reinterpret_cast<const uint32_t*>(a.D8 + (size_t)b * a.dstride + job.dig_off + (size_t)w * kpad);

  // A: clear the histogram
  wv.lanes([&](uint32_t lane) {
    for (uint32_t i = lane; i < 132; i += 64) cnt[i] = 0;
  });
  wv.sync();
  // B: histogram of |d|
  wv.lanes([&](uint32_t lane) {
    for (uint32_t i = lane; i < kpad / 4; i += 64) {
      const uint32_t four = src[i];
#pragma unroll
      for (uint32_t q = 0; q < 4; ++q) {
        const int d = (int8_t)(four >> (8 * q));
        if (4 * i + q < K) wv.atomic_add(&cnt[d < 0 ? -d : d], 1u);
      }
    }
  });
  wv.sync();
  // C: offsets of the counting sort (wavefront prefix sum over the lanes' two-bucket counts)
  PerLane<uint32_t> lo1, lo2, end, pair;
  wv.lanes([&](uint32_t lane) { pair[lane] = cnt[2 * lane + 1] + cnt[2 * lane + 2]; });
  wv.excl_scan(pair);
  wv.lanes([&](uint32_t lane) {
    const uint32_t c1 = cnt[2 * lane + 1], c2 = cnt[2 * lane + 2];
    lo1[lane] = pair[lane];
    lo2[lane] = pair[lane] + c1;
    end[lane] = pair[lane] + c1 + c2;
  });
  wv.sync();
  wv.lanes([&](uint32_t lane) {            // cnt[] becomes off[]: first sorted position of every bucket, off[129] = total
    cnt[2 * lane + 1] = cur[2 * lane + 1] = lo1[lane];
    cnt[2 * lane + 2] = cur[2 * lane + 2] = lo2[lane];
    if (lane == 63) cnt[129] = end[lane];
  });
  wv.sync();
  const uint32_t* off = cnt;
  const uint32_t T = off[129];
  // D: scatter (zero digits take no part)
  wv.lanes([&](uint32_t lane) {
    for (uint32_t i = lane; i < kpad / 4; i += 64) {
      const uint32_t four = src[i];
#pragma unroll
      for (uint32_t q = 0; q < 4; ++q) {
        const int d = (int8_t)(four >> (8 * q));
        const uint32_t t = 4 * i + q;
        if (t < K && d != 0) {
          const uint32_t pos = wv.atomic_add(&cur[d < 0 ? -d : d], 1u);
          ix[pos] = (uint16_t)(t | (d < 0 ? 0x8000u : 0u));
        }
      }
    }
  });
  wv.sync();
  // E: accumulation, one mixed addition per term: run += +-P_t.
  //   pair mode     every lane takes one odd and one even bucket -- the r-th fullest odd one with the r-th emptiest even one, so
  //                 that the lanes' shares differ by a term or two instead of by the +-2.4 sigma of the fullest pair {2l+1, 2l+2}
  //                 (78 against 59.5 terms on average at 3 808 terms: the wave waits for its slowest lane) -- and sums them one
  //                 after the other (acc = the odd bucket's sum, run = the even one's); the sums then change lanes through LDS
  //                 (stage F) to where the reduction wants them;
  //   balanced mode (a window whose digits crowd into few buckets -- the top window of a 252-bit scalar has 8): equal shares of
  //                 the sorted list, walked from the top; whenever a lane crosses into the next lower bucket it adds the running
  //                 sum to acc, so that at the end  sum_t d_t P_t over the share = lo * run + acc  with lo the lowest bucket reached
  //                 (the buckets are long there, so a share crosses at most one boundary or so).
  PerLane<uint32_t> n, nA, oA, oB, s1, cb;
  wv.lanes([&](uint32_t lane) {              // ranks by counting: 2 x 64 broadcast reads per lane
    const uint32_t co = lo2[lane] - lo1[lane], ce = end[lane] - lo2[lane];
    uint32_t ro = 0, re = 0;
    for (uint32_t l = 0; l < 64; ++l) {
      const uint32_t o = off[2 * l + 2] - off[2 * l + 1], e = off[2 * l + 3] - off[2 * l + 2];
      ro += (o > co || (o == co && l < lane)) ? 1u : 0u;
      re += (e > ce || (e == ce && l < lane)) ? 1u : 0u;
    }
    cur[ro] = lane;                          // cur[r] = the lane whose odd bucket is the r-th fullest,
    cur[64 + 63 - re] = lane;                // cur[64 + r] = the lane whose even bucket is the r-th emptiest
  });
  wv.sync();
  wv.lanes([&](uint32_t lane) {
    const uint32_t A = 2 * cur[lane] + 1, B2 = 2 * cur[64 + lane] + 2;
    oA[lane] = off[A];
    nA[lane] = off[A + 1] - off[A];
    oB[lane] = off[B2];
    n[lane] = nA[lane] + off[B2 + 1] - off[B2];
  });
#ifdef MP_EXP_BK_BALANCED  // experiment hook (tools/ab_build.py): every window in balanced mode
  const bool balanced = true;
#else
  const bool balanced = wv.max(n) > T / 64 + T / 128 + 32;
#endif
  PerLane<Xyzz<C>> run, acc;
  wv.lanes([&](uint32_t lane) {
    run[lane] = xyzz_inf<C>();
    acc[lane] = xyzz_inf<C>();
    if (balanced) {
      const uint32_t s0 = (uint32_t)(((uint64_t)T * lane) >> 6);
      s1[lane] = (uint32_t)(((uint64_t)T * (lane + 1)) >> 6);
      uint32_t lo_b = 1, hi_b = 128;                    // largest bucket whose first position is <= s1 - 1
      const uint32_t last = s1[lane] ? s1[lane] - 1 : 0;
      while (lo_b < hi_b) {
        const uint32_t mid = (lo_b + hi_b + 1) >> 1;
        if (off[mid] <= last) lo_b = mid; else hi_b = mid - 1;
      }
      n[lane] = s1[lane] - s0;
      cb[lane] = lo_b;
    }
  });
  const uint32_t iters = wv.max(n);
  for (uint32_t i = 0; i < iters; ++i) {
    wv.lanes([&](uint32_t lane) {
      if (i < n[lane]) {
        uint32_t p;
        if (balanced) {
          p = s1[lane] - 1 - i;
          while (p < off[cb[lane]]) {                   // into the next lower bucket
            xyzz_add_ip<C>(acc[lane], run[lane]);
            cb[lane] -= 1;
          }
        } else {
          if (i == nA[lane]) {                          // the odd bucket is done
            acc[lane] = run[lane];
            run[lane] = xyzz_inf<C>();
          }
          p = i < nA[lane] ? oA[lane] + i : oB[lane] + (i - nA[lane]);
        }
        const uint32_t e = ix[p];
        const uint32_t tb = a.bterms[job.begin + (e & 0x7FFFu)].b;
        const Aff<C> q = ld_aff<C>(a.P + p_off<C>(tb & BK_SLOT_MASK, a.Bpad, b + (tb >> 20) * a.link_stride));
        xyzz_madd_signed_ip<C>(run[lane], q, (e & 0x8000u) != 0);
      }
    });
  }
  wv.sync();                                                          // the sort arrays are dead from here on
  // F: sum over the buckets of k S_k
  if (!balanced) {
    // the sums go home: lane l wants S_{2l+2} and S_{2l+1} (two rounds through the 64 exchange slots, even buckets first)
    wv.lanes([&](uint32_t lane) {
      if (n[lane] <= nA[lane]) {                                      // (an empty even bucket: the loop never switched)
        acc[lane] = run[lane];
        run[lane] = xyzz_inf<C>();
      }
      xyzz_to_words<C>(run[lane], xch + cur[64 + lane] * XW);
    });
    wv.sync();
    wv.lanes([&](uint32_t lane) { run[lane] = xyzz_from_words<C>(xch + lane * XW); });
    wv.sync();
    wv.lanes([&](uint32_t lane) { xyzz_to_words<C>(acc[lane], xch + cur[lane] * XW); });
    wv.sync();
    wv.lanes([&](uint32_t lane) {                                     // acc = S_{2l+2}, run = S_{2l+2} + S_{2l+1}:
      acc[lane] = run[lane];                                          // (2l+2) S_{2l+2} + (2l+1) S_{2l+1} = (2l+1) run + acc
      xyzz_add_ip<C>(run[lane], xyzz_from_words<C>(xch + lane * XW));
    });
    wv.sync();
    // sum_l (2l+1) run_l = Suf_0 + 2 sum_{l>=1} Suf_l with the inclusive suffix sums Suf_l = sum_{l'>=l} run_l'
    for (uint32_t s = 1; s < 64; s <<= 1) {
      wv.lanes([&](uint32_t lane) { xyzz_to_words<C>(run[lane], xch + lane * XW); });
      wv.sync();
      wv.lanes([&](uint32_t lane) {
        if (lane + s < 64) xyzz_add_ip<C>(run[lane], xyzz_from_words<C>(xch + (lane + s) * XW));
      });
      wv.sync();
    }
    wv.lanes([&](uint32_t lane) {
      Xyzz<C> t = run[lane];
      if (lane >= 1) xyzz_dbl_ip<C>(t);
      xyzz_add_ip<C>(acc[lane], t);
    });
  } else {
    // arbitrary small lo_l: lo_l * run_l by double-and-add over the bits of the largest lo (wave-uniform trip count)
    const uint32_t lomax = wv.max(cb);
    int nb = 0;
    while ((lomax >> nb) != 0) ++nb;
    PerLane<Xyzz<C>> prod;
    wv.lanes([&](uint32_t lane) { prod[lane] = xyzz_inf<C>(); });
    for (int bit = nb - 1; bit >= 0; --bit) {
      wv.lanes([&](uint32_t lane) {
        xyzz_dbl_ip<C>(prod[lane]);
        if ((cb[lane] >> bit) & 1u) xyzz_add_ip<C>(prod[lane], run[lane]);
      });
    }
    wv.lanes([&](uint32_t lane) { xyzz_add_ip<C>(acc[lane], prod[lane]); });
  }
  for (uint32_t s = 32; s >= 1; s >>= 1) {                            // tree reduction
    wv.lanes([&](uint32_t lane) { xyzz_to_words<C>(acc[lane], xch + lane * XW); });
    wv.sync();
    wv.lanes([&](uint32_t lane) {
      if (lane < s) xyzz_add_ip<C>(acc[lane], xyzz_from_words<C>(xch + (lane + s) * XW));
    });
    wv.sync();
  }
  wv.lanes([&](uint32_t lane) {
    if (lane == 0) st_jac<C>(a.J + j_off<C>(job.win_first + w, a.Bpad, b), xyzz_to_jac<C>(acc[lane]));
  });
}
MP_WAVE_KERNEL(k_bucket_msm, BucketArgs, body_bucket_msm)

// ---- fold the window results: R = sum_w 2^(8w) R_w (x = proof, y = bucket job)
// The same kernel folds the range sums of window-split Straus jobs (layout.hpp vsplit_lo; vb_nwin != 0): job.count parts, part w
// starts at the 5-bit window vsplit_lo(w, job.count, vb_nwin), R = sum_w 2^(5 vsplit_lo(w)) R_w.
struct BFoldArgs {
  uint32_t* J;
  const BJob* jobs;
  uint32_t Bpad, nwin;
  uint32_t vb_nwin;      // 0: bucket windows (nwin parts, BK_BITS apart); else the Straus windows a split job's parts share
};
MP_HD uint32_t fold_parts(const BFoldArgs& a, const BJob& job) { return a.vb_nwin ? job.count : a.nwin; }
// doublings between part w + 1 and part w
MP_HD uint32_t fold_bits(const BFoldArgs& a, const BJob& job, uint32_t w) {
  return a.vb_nwin ? (uint32_t)VB_WINDOW_BITS * (vsplit_lo(w + 1, job.count, a.vb_nwin) - vsplit_lo(w, job.count, a.vb_nwin)) : (uint32_t)BK_BITS;
}
template <class C>
MP_HD void body_bucket_fold(const BFoldArgs& a, uint32_t b, uint32_t y) {
  const BJob job = a.jobs[y];
  const uint32_t parts = fold_parts(a, job);
  Jac<C> acc = ld_jac<C>(a.J + j_off<C>(job.win_first + parts - 1, a.Bpad, b));
#pragma unroll 1
  for (int w = (int)parts - 2; w >= 0; --w) {
    const uint32_t nd = fold_bits(a, job, (uint32_t)w);
#pragma unroll 1
    for (uint32_t q = 0; q < nd; ++q) jac_dbl_ip<C>(acc);
    jac_add_ip<C>(acc, ld_jac<C>(a.J + j_off<C>(job.win_first + (uint32_t)w, a.Bpad, b)));
  }
  st_jac<C>(a.J + j_off<C>(job.out, a.Bpad, b), acc);
}
MP_KERNEL_OCC(k_bucket_fold, BFoldArgs, body_bucket_fold, Geo<C>::OCC4)

}  // namespace mp
#include "kernels_quad.hpp"
#define MP_BUCKET_KERNELS(X, C)                          \
  MP_KERNEL_INST(X, k_bucket_recode, BRecodeArgs, C)     \
  MP_WAVE_KERNEL_INST(X, k_bucket_msm, BucketArgs, C)    \
  MP_KERNEL_INST(X, k_bucket_fold, BFoldArgs, C)         \
  MP_WAVE_KERNEL_INST(X, k_var_msm_q, VarQuadArgs, C)    \
  MP_WAVE_KERNEL_INST(X, k_bucket_fold_q, BFoldQuadArgs, C) \
  MP_WAVE_KERNEL_INST(X, k_fixed_msm_q, FixedQuadArgs, C)   \
  MP_WAVE_KERNEL_INST(X, k_combine_q, CombineQuadArgs, C)
