// Bucket-method (Pippenger) multi-scalar multiplication for the LARGE MSMs -- the verifier's products over a 1024-card deck
// (4N + 11m + 9 = 4193 terms in one equation at m = 8, n = 128), a chain's equation, the ONE equation a group of up to 128 proofs is
// screened by (30 464 terms at 52 cards) -- as a WAVE-COOPERATIVE kernel: one 64-lane wave owns one (proof, MSM, window) at a time and
// computes sum_t d_t P_t for the window's signed c-bit digits d_t (c = 8 .. 11: 2^(c-1) buckets |d|, NB = 2^(c-1) / 64 per lane):
//
//   A  the window's K digits (int16) are read as words of two from the proof-major digit array (twice: B and D; they stay in cache);
//   B  histogram of |d| over the buckets (LDS atomics);
//   C  bucket offsets: an exclusive wavefront prefix sum over the lanes' NB-bucket totals gives the offsets of a counting sort;
//   D  scatter: the terms' point references (sign in bit 31) sorted by bucket -- into a scratch row in GLOBAL memory that belongs to
//      the wave (round 4: with the list in LDS a wave could sort ~7 600 terms before the CU ran out of LDS for two waves per SIMD;
//      the equation of 128 proofs spreads the wave-wide reduction F over four times as many terms and affords 10-bit windows,
//      26 instead of 32 per scalar).  The waves are persistent -- 8 per CU, each with one scratch row, drawing items from a counter --
//      so the scratch is 2 048 rows whatever the batch, written and read back by the same CU;
//   E  accumulation: every lane sums NB buckets one after the other, run += +-P_t (XYZZ + affine, 8M+2S), and parks every finished
//      sum in the wave's second scratch row (2^(c-1) XYZZ slots, global).  WHICH buckets is a matter of balance -- the wave waits
//      for its slowest lane: of the 64 buckets j mod NB the lane takes the r-th fullest for even j and the r-th emptiest for odd j
//      (ranks by counting), so the lanes' shares differ by a term or two instead of by the +-2.4 sigma of a fixed assignment; a
//      window whose digits crowd into a few buckets (the top window of a 252-bit scalar has 4 or 8) is cut into equal shares of
//      the sorted list instead;
//   F  bucket reduction across the wave: lane l reads the sums of ITS buckets NB l + 1 .. NB l + NB back, top first:
//      sum_j (NB l + j) S_j = (NB l + 1) R_l + A_l with R_l the sum of all NB and A_l the sum of the first NB - 1 running sums
//      (2 NB - 3 additions, NO multiplication by a bucket number), and
//      sum_l [(NB l + 1) R_l + A_l] = sum_l A_l + Suf_0 + NB sum_{l>=1} Suf_l with the inclusive suffix sums Suf_l = sum_{l'>=l} R_l'
//      -- a 6-step wavefront suffix scan, log2 NB doublings and a 6-step tree reduction of XYZZ points exchanged through LDS
//      (13 + 2 NB - 3 point additions per window instead of the 2 x 2^(c-1) of the serial running sum).
//
// Round 5 (the kernel was at 0.855 of its issue slots with 3.7 TB/s of 64-byte gathers beside it): (1) the points of
// a group equation are gathered ONCE into a contiguous run per equation (k_group_tile: 2 MB for 128 proofs of a 52-card deck, two points
// per 128-byte line, one page instead of 238 x 128 pieces of a 4 GB arena) and the sorted entries index that run; (2) the (equation,
// window) items are handed out XCD-affine -- one counter per XCD over whole equations, so that the 26 windows of an equation meet in ONE
// L2; (3) the point of term i + 1 travels global -> LDS (global_load_lds_dwordx4, no registers) while term i is added.
// The W window results of an MSM are folded (c doublings + 1 addition per window) by k_bucket_fold; its output slot joins
// the MSM's other partial sums in k_combine exactly like a Straus sub-job's.  Per term and window this is one mixed
// addition plus (load imbalance + the wave-wide additions of F) / terms-per-lane -- 26 to 33 windows instead of Straus' 51 and no
// per-base window tables at all (no k_table work, no 1 KB of table per base): it wins from ~2 000 terms per MSM on
// (DESIGN.md "bucket MSM"); below that the Straus kernel (kernels_msm.hpp) stays.
// Replaces ark-ec 0.3 `VariableBaseMSM::multi_scalar_mul` (the bucket method, sequential on the CPU) inside the
// reference's verifier [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:437-442].
#pragma once
#include "kernels_msm.hpp"

namespace mp {

static const uint32_t BK_BITS_MIN = 8, BK_BITS_MAX = 14;     // signed windows: digits in [-2^(c-1), 2^(c-1)]
// (windows of 12 bits and more: the split pipeline at the end of this file, round 6)
#ifdef MP_EXP_BK_OCC
static const uint32_t BK_WAVES_PER_CU = 4 * MP_EXP_BK_OCC;   // persistent waves: MP_EXP_BK_OCC workgroups of 4 per CU
#else
static const uint32_t BK_WAVES_PER_CU = 8;                   // persistent waves (MP_WAVE_KERNEL: 2 workgroups of 4 per CU)
#endif
// Round 6: a scalar k is recoded as min(k, q - k) with the signs of its digits flipped in the second case: the value is below
// 2^(bits - 1), the top window never carries out, and ceil(bits / c) windows do -- 18 of 14 bits, 21 of 12 for a 252-bit order instead of
// 19 and 22 (rounds 2-5 recoded k itself: (bits + c) / c windows)
static inline uint32_t bk_windows(int scalar_bits, uint32_t c) { return ((uint32_t)scalar_bits + c - 1u) / c; }
MP_HD uint32_t bk_buckets(uint32_t c) { return 1u << (c - 1); }
// window width for an MSM of K terms: the reduction F costs ~(20 + 4 NB) additions per window, a narrower window K / (c (c + 1)) more
// (12 bits and more: the split pipeline at the end of this file; measured on whole steps, profiles/r06i_batches.txt: 256 proofs of a
// 52-card deck -- 60 928 points -- are better off with 12 bits there than with 11 on one wave per window.  From 200 000 points on 14
// bits: 18 windows for a 252-bit order, all of them full -- 13-bit windows are 20, the top one holding 17 values and summed in list mode
// --: 66.4 against 72.1 ms per launch for the 243 712-point equations of 262 144 proofs, profiles/r06r_bits14.txt)
static inline uint32_t bk_bits_for(uint32_t K) {
  return K >= 200000u ? 14u : (K >= 50000u ? 12u : (K >= 40000u ? 11u : (K >= 12000u ? 10u : (K >= 6000u ? 9u : 8u))));
}
// LDS words one item's lanes need: counts + cursors of the buckets, one XYZZ exchange slot of xw words per lane
static inline uint32_t bk_lds_words(uint32_t c, uint32_t xw, uint32_t lanes = 64u) { return 2u * (bk_buckets(c) + 4u) + lanes * xw; }

// ---- digits: canonical scalar -> W signed digits, d_w in [-2^(c-1), 2^(c-1) - 1] (top window non-negative), proof-major:
// D16[b * dstride + pos + w * kpad]  (pos = digit offset of the term inside the proof's block, kpad = padded terms of its MSM)
struct BRecodeArgs {
  const uint32_t* S;
  int16_t* D16;
  const Term* bterms;      // {S slot, P slot}
  const BTermPos* bpos;
  uint32_t Bpad, nwin, nterms;
  size_t dstride;
  uint32_t bits;
};
template <class C>
MP_HD void body_bucket_recode(const BRecodeArgs& a, uint32_t xx, uint32_t) {
  typedef typename C::FrP R;
  const uint32_t x = xx % a.nterms, b = xx / a.nterms;
  const Term t = a.bterms[x];
  const BTermPos ps = a.bpos[x];
  uint32_t k[10];
  fe_to_canonical<R>(ld_fe<R>(a.S + s_off(t.s, a.Bpad, b)), k);
  k[8] = k[9] = 0;
  // k or q - k, whichever is smaller (k = 0 stays: q - 0 is not smaller)
  uint32_t u[8];
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t d = (uint64_t)R::MOD[i] - k[i] - br;
    u[i] = (uint32_t)d;
    br = (uint32_t)(d >> 63);
  }
  bool flip = false, decided = false;
#pragma unroll
  for (int i = 7; i >= 0; --i)
    if (!decided && u[i] != k[i]) {
      flip = u[i] < k[i];
      decided = true;
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) k[i] = flip ? u[i] : k[i];
  int16_t* out = a.D16 + (size_t)b * a.dstride + ps.pos;
  const uint32_t c = a.bits, half = 1u << (c - 1), mask = (1u << c) - 1u;
  uint32_t carry = 0;
  for (uint32_t w = 0; w < a.nwin; ++w) {
    const uint32_t bit = c * w, wd = bit >> 5, sh = bit & 31u;
    const uint64_t two = (uint64_t)k[wd] | ((uint64_t)k[wd + 1] << 32);
    uint32_t raw = ((uint32_t)(two >> sh) & mask) + carry;
    carry = 0;
    if (w + 1 < a.nwin && raw >= half) {
      raw -= 1u << c;
      carry = 1;
    }
    const int32_t d = (int32_t)raw;              // (the top window: 0 .. 2^(c-1), the last bucket included)
    out[(size_t)w * ps.kpad] = (int16_t)(flip ? -d : d);
  }
}
// thread = proof * nterms + bucket term of the phase (the term is the fast axis: the digit stores of a wave are contiguous)
MP_KERNEL(k_bucket_recode, BRecodeArgs, body_bucket_recode)

// ---- the bucket kernel ----------------------------------------------------------------------------------------
struct BucketArgs {
  const int16_t* D16;
  const uint32_t* P;       // affine points, slot-major arena
  uint32_t* J;
  const BJob* jobs;
  const Term* bterms;
  uint32_t Bpad, nwin, njobs;
  size_t dstride;
  uint32_t link_stride;    // chain verification: term.b = P slot | link << 20, the point lives in lane b + link * link_stride
  uint32_t bits;           // window width c
  uint32_t nitems, nslots; // (proof, MSM, window) items for nslots persistent waves
  uint32_t* counter;       // [8] the next item to hand out, per partition of the equations (zero at launch)
  uint32_t count;          // equations (proofs, chains, groups)
  uint32_t parts;          // 8: equation e belongs to XCD e mod 8 -- its windows are drawn by waves of that XCD first, so that its points
                           // stay in ONE L2 (the L2s are per XCD); 1: one counter for everybody
  const uint32_t* tile;    // group verification: the points of equation e as ONE contiguous run [e][term] (k_group_tile); a sorted
  uint32_t tile_K;         // entry then is the term's index in the run instead of a P slot | link.  null: gather from the P arena
  uint32_t stage;          // != 0: the point of term i + 1 travels global -> LDS while term i is added (pair mode)
  uint32_t* sorted;        // scratch [nslots][kpad_max]: the window's point references (term.b | sign << 31) sorted by bucket
  uint32_t* park;          // scratch [nslots][2^(c-1)][XYZZ words]: the bucket sums on their way to the lanes that reduce them
  uint32_t kpad_max;
  uint32_t* timing;        // MP_EXP_BK_TIMING builds: [nslots][4] cycle counts (64-bit)
};
struct alignas(16) Quad32 {
  uint32_t v[4];
};
static const uint32_t BK_SLOT_MASK = 0xFFFFFu;
static const uint32_t BK_LINK_MASK = 0x3FFu;     // (bit 31 of a sorted entry is the sign: 1 023 links)
static const uint32_t BK_TILE_MASK = 0xFFFFFFu;  // ... or the term's index in the equation's contiguous run (k_group_tile)
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
MP_HD void body_bucket_msm(const BucketArgs& a, uint32_t slot, W& wv) {
  constexpr uint32_t XW = XyzzWords<C>::N, NL = W::NL, LOG_NL = W::LOG_NL;      // NL lanes share the item: a wave (64) or a workgroup (256)
  const uint32_t NBK = bk_buckets(a.bits), NB = NBK >> LOG_NL, LOGNB = a.bits - 1u - LOG_NL, HW = NBK + 4u;
  uint32_t* cnt = wv.lds;                         // [0 .. NBK]: terms per |digit|; then off[]: first sorted position per bucket
  uint32_t* cur = wv.lds + HW;                    // scatter cursors; after the scatter: the lanes' bucket assignment [class][lane]
  uint32_t* xch = wv.lds + 2 * HW;                // point exchange of the reduction
  uint32_t* ix = a.sorted + (size_t)slot * a.kpad_max;
  uint32_t* park = a.park + (size_t)slot * NBK * XW;
#ifdef MP_EXP_BK_TIMING     // experiment: cycles per phase (sort, ranks, additions, reduction), summed per wave into the tail of its park row
  unsigned long long tm_[5] = {0, 0, 0, 0, 0}, t0_ = 0;
#define MP_BK_T0() t0_ = __builtin_readcyclecounter()
#define MP_BK_T(i) do { const unsigned long long t1_ = __builtin_readcyclecounter(); tm_[i] += t1_ - t0_; t0_ = t1_; } while (0)
#else
#define MP_BK_T0() do {} while (0)
#define MP_BK_T(i) do {} while (0)
#endif
  // (items are handed out by a counter, not dealt in advance: the windows of an MSM do not take the same time -- the top one is short --
  // and neither do the XCDs; with 2 048 x 256 items dealt in advance the last wave finished 15 % behind the average)
  // XCD-affine since round 5: the equations are dealt to a.parts partitions (e mod parts), one counter each; a wave draws from the
  // partition of its own XCD until that is empty and then helps the next one.  The windows of an equation are consecutive items of
  // one partition: they run at about the same time on one XCD and find each other's points in its L2
  const uint32_t per_eq = a.nwin * a.njobs;
  uint32_t part = a.parts > 1 ? wv.xcd() % a.parts : 0u, tried = 0;
#pragma unroll 1
  for (;;) {
    const uint32_t li = wv.next_item(a.counter + part);
    if (li >= ((a.count + a.parts - 1u - part) / a.parts) * per_eq) {      // this partition is handed out: on to the next
      if (++tried >= a.parts) break;
      part = part + 1 == a.parts ? 0u : part + 1;
      continue;
    }
    MP_BK_T0();
    const uint32_t w = li % a.nwin, jb = (li / a.nwin) % a.njobs, b = (li / per_eq) * a.parts + part;
    const BJob job = a.jobs[jb];
    const uint32_t K = job.count, kpad = job.kpad;
    // (kpad and every offset in front of a window's digits are multiples of 64 digits: 16-byte loads of eight digits)
    const Quad32* src = reinterpret_cast<const Quad32*>(a.D16 + (size_t)b * a.dstride + job.dig_off + (size_t)w * kpad);

    const uint32_t* off = cnt;
    uint32_t T = 0;
#ifdef MP_EXP_BK_SORT2     // experiment: the sort phases twice (their share of the kernel)
    for (int rep_ = 0; rep_ < 2; ++rep_) {
#endif
    // A: clear the histogram
    wv.lanes([&](uint32_t lane) {
      for (uint32_t i = lane; i < HW; i += NL) cnt[i] = 0;
    });
    wv.sync();
    // B: histogram of |d|.  B and D are latency, not work: their loops are branch-free (the padding of the row counts as zero digits,
    // sorted behind every bucket; the last iteration loads its own words once more) with the next iteration's loads in flight over the atomics
    const uint32_t nq = kpad / 8;
    wv.lanes([&](uint32_t lane) {
      Quad32 next = src[lane < nq ? lane : nq - 1];
      for (uint32_t i = lane; i < nq; i += NL) {
        const Quad32 eight = next;
        next = src[i + NL < nq ? i + NL : nq - 1];
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) {
          const int d = 8 * i + q < K ? (int16_t)(eight.v[q >> 1] >> (16 * (q & 1))) : 0;
          wv.atomic_add(&cnt[d < 0 ? -d : d], 1u);
        }
      }
    });
    wv.sync();
    // C: offsets of the counting sort (wavefront prefix sum over the lanes' NB-bucket totals)
    typename W::template PL<uint32_t> tot;
    wv.lanes([&](uint32_t lane) {
      uint32_t t = 0;
      for (uint32_t j = 0; j < NB; ++j) t += cnt[NB * lane + 1 + j];
      tot[lane] = t;
    });
    wv.excl_scan(tot);
    wv.sync();
    wv.lanes([&](uint32_t lane) {            // cnt[] becomes off[]: first sorted position of every bucket, off[NBK + 1] = total
      uint32_t o = tot[lane];
      for (uint32_t j = 0; j < NB; ++j) {
        const uint32_t k = NB * lane + 1 + j, c = cnt[k];
        cnt[k] = cur[k] = o;
        o += c;
      }
      if (lane == NL - 1) cnt[NBK + 1] = cur[0] = o;      // (the zero digits: sorted behind every bucket, never visited)
    });
    wv.sync();
    T = off[NBK + 1];
    // D: scatter: every term of the padded row goes to its place (one store each: the waits can count them)
    wv.lanes([&](uint32_t lane) {
      const Term* terms = a.bterms + job.begin;
      Quad32 next = src[lane < nq ? lane : nq - 1];
      uint32_t nref[8];
#pragma unroll
      for (uint32_t q = 0; q < 8; ++q) nref[q] = terms[8 * lane + q < K ? 8 * lane + q : K - 1].b;
      for (uint32_t i = lane; i < nq; i += NL) {
        const Quad32 eight = next;
        uint32_t ref[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) ref[q] = nref[q];
        const uint32_t j = i + NL < nq ? i + NL : nq - 1;
        next = src[j];
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) nref[q] = terms[8 * j + q < K ? 8 * j + q : K - 1].b;
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) {
          const int d = 8 * i + q < K ? (int16_t)(eight.v[q >> 1] >> (16 * (q & 1))) : 0;
          const uint32_t at = wv.atomic_add(&cur[d < 0 ? -d : d], 1u);
          ix[at] = ref[q] | (d < 0 ? 0x80000000u : 0u);
        }
      }
    });
    wv.sync();
#ifdef MP_EXP_BK_SORT2
    }
#endif
    MP_BK_T(0);
    // E: accumulation, one mixed addition per term: run += +-P_t.  The lane's walk through the sorted list runs two terms ahead of
    // the additions: the entry of term i + 2 is requested while term i is added, so that only the point itself is waited for.
    typename W::template PL<uint32_t> n, seg, pos, rem;        // the fetch side (balanced mode: pos = one past the share's top position, rem = the bucket the additions are in)
    typename W::template PL<uint32_t> mseg, e0, e1, nb0, nb1;  // the additions' side: class of the bucket being summed; entries of terms i and i + 1, and whether
                                               // they open a new bucket (kept apart from the entry: nothing may wait for the load before its turn)
    wv.lanes([&](uint32_t lane) {              // ranks by counting: NB x NL broadcast reads per lane
      for (uint32_t j = 0; j < NB; ++j) {
        const uint32_t mine = off[NB * lane + 2 + j] - off[NB * lane + 1 + j];
        uint32_t r = 0;
        for (uint32_t l = 0; l < NL; ++l) {
          const uint32_t o = off[NB * l + 2 + j] - off[NB * l + 1 + j];
          r += (o > mine || (o == mine && l < lane)) ? 1u : 0u;
        }
        cur[j * NL + ((j & 1u) ? NL - 1u - r : r)] = lane;      // worker lane -> the home lane of its bucket of class j
      }
    });
    wv.sync();
    wv.sync_global();                          // (the sorted list is read by other lanes than wrote it)
    wv.lanes([&](uint32_t lane) {
      uint32_t t = 0;
      for (uint32_t j = 0; j < NB; ++j) {
        const uint32_t k = NB * cur[j * NL + lane] + 1 + j;
        t += off[k + 1] - off[k];
      }
      n[lane] = t;
    });
#ifdef MP_EXP_BK_BALANCED  // experiment hook (tools/ab_build.py): every window in balanced mode
    const bool balanced = true;
#else
    const bool balanced = wv.max(n) > T / NL + T / (2 * NL) + 32;
#endif
    // the position of the next entry of the lane's walk (only called while terms remain); fresh: it is the first term of a new bucket
    auto advance = [&](uint32_t lane, uint32_t j, uint32_t& fresh) -> uint32_t {
      fresh = 0;
      if (balanced) return pos[lane] - 1 - j;
      while (rem[lane] == 0) {                 // on to the lane's bucket of the next class (empty ones are passed over)
        seg[lane] += 1;
        const uint32_t k = NB * cur[seg[lane] * NL + lane] + 1 + seg[lane];
        pos[lane] = off[k];
        rem[lane] = off[k + 1] - off[k];
        fresh = 1;
      }
      pos[lane] += 1;
      rem[lane] -= 1;
      return pos[lane] - 1;
    };
    const uint32_t* const tile = a.tile ? a.tile + (size_t)b * a.tile_K * Geo<C>::PW : nullptr;
    auto point_of = [&](uint32_t e) -> const uint32_t* {
      if (tile) return tile + (size_t)(e & BK_TILE_MASK) * Geo<C>::PW;
      return a.P + p_off<C>(e & BK_SLOT_MASK, a.Bpad, b + ((e >> 20) & BK_LINK_MASK) * a.link_stride);
    };
    // pair mode: the exchange area is idle during the additions -- it stages the next point (4 KB of its 9)
    const bool staged = a.stage != 0 && !balanced;
    // (acc is not needed while the pair-mode loop runs; the balanced walk keeps it in the lane's exchange slot in LDS -- 32 registers less
    // across the mixed addition)
    typename W::template PL<Xyzz<C>> run;
    wv.lanes([&](uint32_t lane) {
      run[lane] = xyzz_inf<C>();
      seg[lane] = 0;
      mseg[lane] = 0;
      if (balanced) {
        xyzz_to_words<C>(run[lane], xch + lane * XW);
        // equal shares of the sorted list, walked from the top; whenever a lane crosses into the next lower bucket it adds the running
        // sum to acc, so that at the end  sum_t d_t P_t over the share = lo * run + acc  with lo the lowest bucket reached
        const uint32_t s0 = (uint32_t)(((uint64_t)T * lane) >> LOG_NL), s1 = (uint32_t)(((uint64_t)T * (lane + 1)) >> LOG_NL);
        uint32_t lo_b = 1, hi_b = NBK;                    // largest bucket whose first position is <= s1 - 1
        const uint32_t last = s1 ? s1 - 1 : 0;
        while (lo_b < hi_b) {
          const uint32_t mid = (lo_b + hi_b + 1) >> 1;
          if (off[mid] <= last) lo_b = mid; else hi_b = mid - 1;
        }
        n[lane] = s1 - s0;
        pos[lane] = s1;
        rem[lane] = lo_b;
      } else {
        const uint32_t k = NB * cur[lane] + 1;
        pos[lane] = off[k];
        rem[lane] = off[k + 1] - off[k];
      }
      e0[lane] = e1[lane] = nb1[lane] = 0;
      uint32_t first = 0;
      if (n[lane] > 0) e0[lane] = ix[advance(lane, 0, first)];            // (empty classes in front of the first term: mseg below)
      if (n[lane] > 1) e1[lane] = ix[advance(lane, 1, nb1[lane])];
      nb0[lane] = 0;
      if (!balanced && n[lane] > 0)
        while (off[NB * cur[mseg[lane] * NL + lane] + 2 + mseg[lane]] == off[NB * cur[mseg[lane] * NL + lane] + 1 + mseg[lane]]) mseg[lane] += 1;
      if (staged && n[lane] > 0) wv.template stage<Geo<C>::PW>(xch, point_of(e0[lane]), lane);
    });
    const uint32_t iters = wv.max(n);
    MP_BK_T(1);
#pragma unroll 1
    for (uint32_t i = 0; i < iters; ++i) {
      wv.lanes([&](uint32_t lane) {
        if (i < n[lane]) {
          const uint32_t e = e0[lane];
          const uint32_t fresh = nb0[lane];
          if (balanced) {
            const uint32_t p = pos[lane] - 1 - i;
            while (p < off[rem[lane]]) {                  // into the next lower bucket
              Xyzz<C> t = xyzz_from_words<C>(xch + lane * XW);
              xyzz_add_ip<C>(t, run[lane]);
              xyzz_to_words<C>(t, xch + lane * XW);
              rem[lane] -= 1;
            }
          } else if (fresh) {                             // the bucket before this term is done: park its sum
            xyzz_to_words<C>(run[lane], park + (size_t)(NB * cur[mseg[lane] * NL + lane] + mseg[lane]) * XW);
            run[lane] = xyzz_inf<C>();
            do mseg[lane] += 1;
            while (off[NB * cur[mseg[lane] * NL + lane] + 2 + mseg[lane]] == off[NB * cur[mseg[lane] * NL + lane] + 1 + mseg[lane]]);
          }
#ifdef MP_EXP_BK_NOPOINT   // experiment: every point out of 256 cache-resident ones (wrong sums; how much of the kernel is memory latency)
          const Aff<C> q = ld_aff<C>(a.P + p_off<C>(e & 0xFFu, a.Bpad, b));
#else
          Aff<C> q;
          if (staged) {                                   // requested one addition ago: it has landed
            uint32_t pw[Geo<C>::PW];
            wv.template take<Geo<C>::PW>(xch, pw, lane);
            q.x = fe_unpack<typename C::FqP>(pw);
            q.y = fe_unpack<typename C::FqP>(pw + Geo<C>::FW);
          } else {
            q = ld_aff<C>(point_of(e));
          }
#endif
          // (the entry of term i + 2 is requested BEHIND the point of term i: loads come back in order, and the addition waits for its
          // point with this one still in flight)
          // (... and by ONE load instruction on every path: the wait for the point counts the loads behind it)
          e0[lane] = e1[lane];
          nb0[lane] = nb1[lane];
          uint32_t at = 0;
          if (i + 2 < n[lane]) at = advance(lane, i + 2, nb1[lane]);
          e1[lane] = ix[at];
          if (staged && i + 1 < n[lane]) wv.template stage<Geo<C>::PW>(xch, point_of(e0[lane]), lane);      // the point of term i + 1
          xyzz_madd_signed_ip<C>(run[lane], q, (e >> 31) != 0);
        }
      });
    }
    wv.sync();
    MP_BK_T(2);
    // F: sum over the buckets of k S_k
    typename W::template PL<Xyzz<C>> acc;
    if (!balanced) {
      wv.lanes([&](uint32_t lane) {                                     // the last sum (empty buckets have no parked sum: the reader knows)
        if (n[lane] > 0) xyzz_to_words<C>(run[lane], park + (size_t)(NB * cur[mseg[lane] * NL + lane] + mseg[lane]) * XW);
      });
      wv.sync_global();
      wv.lanes([&](uint32_t lane) {                                     // the sums come home: R = sum of the lane's NB buckets,
        const uint32_t* mine = park + (size_t)NB * lane * XW;           // A = sum of the first NB - 1 running sums from the top
        const uint32_t* o = off + NB * lane + 1;
        run[lane] = o[NB] != o[NB - 1] ? xyzz_from_words<C>(mine + (size_t)(NB - 1) * XW) : xyzz_inf<C>();
        for (int j = (int)NB - 2; j >= 0; --j) {
          if (j == (int)NB - 2) acc[lane] = run[lane]; else xyzz_add_ip<C>(acc[lane], run[lane]);
          if (o[j + 1] != o[j]) xyzz_add_ip<C>(run[lane], xyzz_from_words<C>(mine + (size_t)j * XW));
        }
      });
      // sum_l (NB l + 1) R_l = Suf_0 + NB sum_{l>=1} Suf_l with the inclusive suffix sums Suf_l = sum_{l'>=l} R_l'
      for (uint32_t s = 1; s < NL; s <<= 1) {
        wv.lanes([&](uint32_t lane) { xyzz_to_words<C>(run[lane], xch + lane * XW); });
        wv.sync();
        wv.lanes([&](uint32_t lane) {
          if (lane + s < NL) xyzz_add_ip<C>(run[lane], xyzz_from_words<C>(xch + (lane + s) * XW));
        });
        wv.sync();
      }
      wv.lanes([&](uint32_t lane) {
        Xyzz<C> t = run[lane];
        if (lane >= 1)
          for (uint32_t q = 0; q < LOGNB; ++q) xyzz_dbl_ip<C>(t);
        xyzz_add_ip<C>(acc[lane], t);
      });
    } else {
      // arbitrary small lo_l: lo_l * run_l by double-and-add over the bits of the largest lo (wave-uniform trip count)
      wv.lanes([&](uint32_t lane) { acc[lane] = xyzz_from_words<C>(xch + lane * XW); });
      wv.sync();
      const uint32_t lomax = wv.max(rem);
      int nb = 0;
      while ((lomax >> nb) != 0) ++nb;
      typename W::template PL<Xyzz<C>> prod;
      wv.lanes([&](uint32_t lane) { prod[lane] = xyzz_inf<C>(); });
      for (int bit = nb - 1; bit >= 0; --bit) {
        wv.lanes([&](uint32_t lane) {
          xyzz_dbl_ip<C>(prod[lane]);
          if ((rem[lane] >> bit) & 1u) xyzz_add_ip<C>(prod[lane], run[lane]);
        });
      }
      wv.lanes([&](uint32_t lane) { xyzz_add_ip<C>(acc[lane], prod[lane]); });
    }
    for (uint32_t s = NL / 2; s >= 1; s >>= 1) {                        // tree reduction
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
    wv.sync_global();                          // (the next item's scatter and parking overwrite what other lanes have just read)
    MP_BK_T(3);
  }
#ifdef MP_EXP_BK_TIMING
  wv.lanes([&](uint32_t lane) {
    if (lane == 0)
      for (int i = 0; i < 4; ++i) reinterpret_cast<unsigned long long*>(a.timing)[(size_t)slot * 4 + i] = tm_[i];
  });
#endif
}
// waves per SIMD the kernel is compiled for: 2 (256 registers); MP_EXP_BK_OCC=3 (168 registers) on the 256-bit curves is an A/B hook
#ifdef MP_EXP_BK_OCC
#define MP_BK_OCC(C) (Geo<C>::FW > 8 ? 2 : MP_EXP_BK_OCC)
#else
#define MP_BK_OCC(C) 2
#endif
MP_WAVE_KERNEL_OCC(k_bucket_msm, BucketArgs, body_bucket_msm, MP_BK_OCC(C))

// ================================================================================================================================
// Round 6: the SPLIT pipeline -- windows of 12 bits and more (mp_set_bucket_split; equations of >= 50 000 points: the screen of 256 ..
// 2 048 proofs of a 52-card deck, of 16 .. 128 proofs of a 1 024-card one, the chains of 16 .. 128 tables; 11 bits too on BLS12-377).
//
// One wave per (equation, window) caps the kernel above at 11 bits: 2^(c-1) buckets on 64 lanes are 16 per lane and a 42-addition
// reduction per window at c = 11, which cancels the gain of 23 windows over 26.  Giving the window to a 256-lane workgroup instead
// (measured first, profiles/r06a_wg_kernel_sweep.txt, r06b_wg_kernel_phases.txt) executed 22 % fewer instructions and was 10 % faster,
// not 25 %: an item of 243 712 points is a tenth of what a persistent workgroup does in the launch, so the launch waited 9 % of its time
// for the last items; the sort of an item ran at two waves per SIMD (the register budget of the additions) and took 11 % of it -- its
// scatter writes four bytes at a time into a 1 MB row, 512 rows in flight: every store a read-modify-write of a 64-byte line in HBM
// (20 ms per launch when it first ran as a kernel of its own, r06c) --; and the ranks by counting cost 256 comparisons per bucket.  So
// the phases are kernels of their own, each with the parallelism and the registers IT wants, the sorted list leaves the chip in whole
// lines, and the additions are handed out in pieces a thirtieth the size:
//
//   k_bucket_sort    one 256-lane workgroup per (item, CHUNK of BK_CHUNK = 24 576 terms), ~32 registers, two workgroups per CU:
//                    histogram of |d| (LDS atomics), exclusive scan (wavefront prefix sums, the four waves joined through LDS),
//                    counting sort of the chunk's terms INSIDE LDS (16-bit indices), then one contiguous copy to the chunk's run of
//                    `sorted`; the bucket offsets inside the run go to `offs` as 16-bit words, with the largest bucket of the chunk in
//                    front.  A bucket's terms are then G = ceil(K / 24 576) short runs, one per chunk, instead of one long one
//                    (5.6 ms per launch for the 4 608 items of 262 144 52-card proofs);
//   k_bucket_acc     one WAVE per unit = (item, one of P bucket ranges of 64 x 4 buckets): the lanes take four buckets each, dealt by
//                    rank inside the unit (64 comparisons per bucket), and walk a bucket's G runs one after the other -- one mixed
//                    addition per term with the next point staged global -> LDS as above, every bucket sum parked in the item's row of
//                    `park` (the point at infinity for an empty bucket: the reduction reads no offsets).  A unit is ~120 additions per
//                    lane at 14 bits: 147 456 units for 262 144 proofs, 70 per wave slot -- the hardware's own workgroup dispatcher
//                    hands them out, XCD r taking the r-th eighth of them (bk_unit_of_wave), and the launch ends within one unit of
//                    its average (55 ms per launch, 0.9 of the issue slots);
//   k_bucket_list    the same launch shape for the items in LIST mode -- a window whose digits crowd into a few buckets (the top window
//                    of a 252-bit scalar holds 9 values at c = 13; an MSM whose scalars are all equal): unit p takes the runs p, p + P,
//                    ... in 64 equal shares each and leaves ONE sum.  An item is in list mode if a chunk's largest bucket exceeds eight
//                    times its share (+ 32).  A kernel of its own: the full additions of the list walk would otherwise set the register
//                    budget of k_bucket_acc's hot loop (92 spilled registers on BLS12-377; none now);
//   k_bucket_reduce  four waves per item, a quarter of the 2^(c-1) parked sums each: sum_j j S_j and sum_j S_j over the quarter as in F
//                    above with NB = 2^(c-1) / 256 buckets per lane; k_bucket_final puts the quarters together (or takes the list sums).
//
// 14-bit windows (the default from 200 000 points on) are 18 for a 252-bit order, all of them full -- no list-mode item in the
// verifier's screen --, their 8 192 counters per chunk are 16-bit halves of LDS words (bk_sort_packed).  `sorted`, `offs` and `park` hold
// every item of a pass at once (3 MB per item at c = 14 and 243 712 points: the engine cuts a call into passes of equations that fit its
// scratch budget).  Results are canonical group elements: bit-identical to the kernel above.
static const uint32_t BK_SPLIT_BITS = 12;
#ifndef MP_EXP_BK_CHUNK     // (A/B hook, tools/ab_build.py)
#define MP_EXP_BK_CHUNK 24576
#endif
static const uint32_t BK_CHUNK = MP_EXP_BK_CHUNK;       // terms per sorted run (48 KB of LDS -- 16-bit local indices -- beside the 16 KB of counters at c = 13; < 32 768: sign in bit 15)
static const uint32_t BK_CHUNKS_MAX = 24;     // runs per bucket (their offsets sit in the accumulating wave's LDS): 589 824 terms per job
static_assert((size_t)BK_CHUNKS_MAX * MP_EXP_BK_CHUNK >= BUCKET_TERMS_MAX || MP_EXP_BK_CHUNK != 24576, "layout.hpp BUCKET_TERMS_MAX");
#ifndef MP_EXP_BK_UNIT_NB
#define MP_EXP_BK_UNIT_NB 4
#endif
MP_HD uint32_t bk_unit_nb(uint32_t c) { return bk_buckets(c) >= 64u * MP_EXP_BK_UNIT_NB ? (uint32_t)MP_EXP_BK_UNIT_NB : bk_buckets(c) / 64u; }      // buckets per lane and unit
MP_HD uint32_t bk_units(uint32_t c) { return bk_buckets(c) / (64u * bk_unit_nb(c)); }                    // bucket ranges per item (P <= 64)
MP_HD uint32_t bk_chunks(uint32_t kpad) { return (kpad + BK_CHUNK - 1u) / BK_CHUNK; }
// offs row of (item, chunk), 16-bit words: [0] largest bucket of the chunk, [k] first position of bucket k inside the run (k = 1 ..
// NBK), [NBK + 1] the chunk's terms with a non-zero digit; rows of NBK + 2 words
MP_HD uint32_t bk_offs_row(uint32_t c) { return bk_buckets(c) + 2u; }
MP_HD bool bk_crowded(uint32_t largest, uint32_t c) { return largest > 8u * (BK_CHUNK / bk_buckets(c)) + 32u; }
struct BSplitArgs {
  const int16_t* D16;
  const uint32_t* P;
  uint32_t* J;
  const BJob* jobs;
  const Term* bterms;
  uint32_t Bpad, nwin, njobs;
  size_t dstride;
  uint32_t link_stride, bits;
  uint32_t eq0, neq;       // this pass: equations [eq0, eq0 + neq) of the call; item = ((e - eq0) njobs + job) nwin + window
  const uint32_t* tile;
  uint32_t tile_K;
  uint32_t* sorted;        // [items][kpad_max]: chunk g's run starts at g BK_CHUNK
  uint16_t* offs;          // [items][gmax][2^(c-1) + 2]
  uint32_t* park;          // [items][2^(c-1)][XYZZ words] bucket sums (list mode: the first G slots hold the chunks' sums)
  uint32_t kpad_max, gmax; // gmax = bk_chunks(kpad_max)
  uint32_t units;          // units per item: its bucket ranges, bk_units(bits)
  uint32_t wpb, wgs;       // k_bucket_acc: waves per workgroup of the launch, workgroups that hold units
  uint32_t* quarters;      // [items][4][2][XYZZ words]: k_bucket_reduce's (A, R) per quarter of the buckets, for k_bucket_final
};
// k_bucket_acc: which unit a wave of the launch takes.  Workgroup i of a launch runs on XCD i mod 8 (observed; the hardware deals
// workgroups round-robin and statically), and consecutive units belong to the same item: dealt as they come, the units of every other
// window -- and with them all the list-mode windows, which are slower -- met on the same half of the XCDs, and the launch waited for
// that half (-10 %: profiles/r06g_xcd_placement_pmc.txt, r06h_ab_xcd_remap.txt).  So XCD r gets the r-th EIGHTH of the units, whole equations: an equation's
// windows share one L2 again (the XCD-affine items of round 5), and every XCD holds the same mix of windows.
MP_HD uint32_t bk_unit_of_wave(const BSplitArgs& a, uint32_t wave, uint32_t nunits) {
  const uint32_t i = wave / a.wpb, lane_wave = wave % a.wpb, per = (a.wgs + 7u) / 8u;
  const uint32_t j = (i % 8u) * per + i / 8u;
  if (j >= a.wgs) return 0xFFFFFFFFu;
  const uint32_t u = j * a.wpb + lane_wave;
  return u < nunits ? u : 0xFFFFFFFFu;
}
struct BItem {
  uint32_t w, jb, b;
};
MP_HD BItem bk_item(const BSplitArgs& a, uint32_t it) {
  BItem r;
  r.w = it % a.nwin;
  r.jb = (it / a.nwin) % a.njobs;
  r.b = a.eq0 + it / (a.nwin * a.njobs);
  return r;
}
// is the item summed in list mode?  (every unit of the item and its reduction ask the same question of the same words)
MP_HD bool bk_list_mode(const BSplitArgs& a, uint32_t it, uint32_t G) {
  const uint16_t* og = a.offs + (size_t)it * a.gmax * bk_offs_row(a.bits);
  uint32_t largest = 0;
  for (uint32_t g = 0; g < G; ++g) {
    const uint32_t m = og[(size_t)g * bk_offs_row(a.bits)];
    largest = m > largest ? m : largest;
  }
  return bk_crowded(largest, a.bits);
}

// ---- k_bucket_sort: W = BlockCtx (256 lanes), x = item * gmax + chunk; wv.lds = bk_sort_lds_words(c) words.
// The counters of more than 4 096 buckets (14-bit windows) are 16-bit halves of LDS words -- a chunk holds 24 576 terms, so neither a
// count nor a cursor reaches 65 536, and an atomic add of 1 << 16 never carries out of its half: 16 KB of counters instead of 32, two
// workgroups per CU as at 13 bits (one per CU took 8.6 ms per launch against 5.7)
MP_HD bool bk_sort_packed(uint32_t c) { return bk_buckets(c) > 4096u; }
MP_HD uint32_t bk_sort_counter_words(uint32_t c) { return bk_sort_packed(c) ? bk_buckets(c) / 2u + 2u : bk_buckets(c) + 2u; }
static inline uint32_t bk_sort_lds_words(uint32_t c) { return bk_sort_counter_words(c) + BK_CHUNK / 2u; }
template <class C, class W>
MP_HD void body_bucket_sort(const BSplitArgs& a, uint32_t x, W& wv) {
  constexpr uint32_t NL = W::NL;
  const uint32_t NBK = bk_buckets(a.bits), ROW = bk_offs_row(a.bits), CW = bk_sort_counter_words(a.bits);
  const bool packed = bk_sort_packed(a.bits);
  const uint32_t it = x / a.gmax, g = x % a.gmax;
  const BItem id = bk_item(a, it);
  const BJob job = a.jobs[id.jb];
  if (g >= bk_chunks(job.kpad)) return;            // (a shorter job of the phase)
  const uint32_t K = job.count, t0 = g * BK_CHUNK, t1 = job.kpad < t0 + BK_CHUNK ? job.kpad : t0 + BK_CHUNK;      // the chunk's terms [t0, t1) of the padded row
  const uint32_t q0 = t0 / 8, nq = (t1 - t0) / 8;
  const Quad32* src = reinterpret_cast<const Quad32*>(a.D16 + (size_t)id.b * a.dstride + job.dig_off + (size_t)id.w * job.kpad) + q0;
  uint32_t* cnt = wv.lds;                          // terms per |digit| (bucket k: word k, or half (k - 1) & 1 of word (k - 1) / 2), then the scatter cursors; the last word: the chunk's total
  uint16_t* buf = reinterpret_cast<uint16_t*>(wv.lds + CW);      // the chunk's sorted run: index of the term inside the chunk, sign in bit 15
  uint16_t* og = a.offs + ((size_t)it * a.gmax + g) * ROW;
  uint32_t* ix = a.sorted + (size_t)it * a.kpad_max + t0;
  // one more term in bucket k: its position (scatter pass) or nothing of interest (histogram pass)
  auto bump = [&](uint32_t k) -> uint32_t {
    if (!packed) return wv.atomic_add(&cnt[k], 1u);
    const uint32_t sh = ((k - 1u) & 1u) << 4;
    return (wv.atomic_add(&cnt[(k - 1u) >> 1], 1u << sh) >> sh) & 0xFFFFu;
  };
  auto get = [&](uint32_t k) -> uint32_t { return packed ? (cnt[(k - 1u) >> 1] >> (((k - 1u) & 1u) << 4)) & 0xFFFFu : cnt[k]; };
  wv.lanes([&](uint32_t lane) {
    for (uint32_t i = lane; i < CW; i += NL) cnt[i] = 0;
  });
  wv.sync();
  wv.lanes([&](uint32_t lane) {
    Quad32 next = src[lane < nq ? lane : nq - 1];
    for (uint32_t i = lane; i < nq; i += NL) {
      const Quad32 eight = next;
      next = src[i + NL < nq ? i + NL : nq - 1];
#pragma unroll
      for (uint32_t q = 0; q < 8; ++q) {
        const int d = t0 + 8 * i + q < K ? (int16_t)(eight.v[q >> 1] >> (16 * (q & 1))) : 0;
        if (d != 0) bump((uint32_t)(d < 0 ? -d : d));
      }
    }
  });
  wv.sync();
  // offsets: lane l owns the buckets [lo, hi) of 1 .. NBK (a share of NBK / NL -- whole words of packed counters: NBK / NL = 32 --, or
  // none when there are fewer buckets than lanes)
  typename W::template PL<uint32_t> tot, big;
  wv.lanes([&](uint32_t lane) {
    const uint32_t lo = 1u + (uint32_t)(((uint64_t)NBK * lane) / NL), hi = 1u + (uint32_t)(((uint64_t)NBK * (lane + 1)) / NL);
    uint32_t t = 0, m = 0;
    for (uint32_t k = lo; k < hi; ++k) {
      const uint32_t c = get(k);
      t += c;
      m = c > m ? c : m;
    }
    tot[lane] = t;
    big[lane] = m;
  });
  wv.excl_scan(tot);
  const uint32_t maxc = wv.max(big);
  wv.lanes([&](uint32_t lane) {
    const uint32_t lo = 1u + (uint32_t)(((uint64_t)NBK * lane) / NL), hi = 1u + (uint32_t)(((uint64_t)NBK * (lane + 1)) / NL);
    uint32_t o = tot[lane];
    if (packed) {
      for (uint32_t k = lo; k < hi; k += 2) {      // (lo - 1 is even and the share is even: the lane's own words)
        const uint32_t w = cnt[(k - 1u) >> 1], c0 = w & 0xFFFFu, c1 = w >> 16;
        og[k] = (uint16_t)o;
        og[k + 1] = (uint16_t)(o + c0);
        cnt[(k - 1u) >> 1] = o | ((o + c0) << 16);
        o += c0 + c1;
      }
    } else {
      for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t c = cnt[k];
        cnt[k] = o;
        og[k] = (uint16_t)o;
        o += c;
      }
    }
    if (lane == NL - 1) {
      og[0] = (uint16_t)maxc;
      og[NBK + 1] = (uint16_t)o;                   // the chunk's terms with a non-zero digit
      cnt[CW - 1] = o;
    }
  });
  wv.sync();
  wv.lanes([&](uint32_t lane) {
    Quad32 next = src[lane < nq ? lane : nq - 1];
    for (uint32_t i = lane; i < nq; i += NL) {
      const Quad32 eight = next;
      next = src[i + NL < nq ? i + NL : nq - 1];
#pragma unroll
      for (uint32_t q = 0; q < 8; ++q) {
        const int d = t0 + 8 * i + q < K ? (int16_t)(eight.v[q >> 1] >> (16 * (q & 1))) : 0;
        if (d != 0) {
          const uint32_t at = bump((uint32_t)(d < 0 ? -d : d));
          buf[at] = (uint16_t)((8 * i + q) | (d < 0 ? 0x8000u : 0u));
        }
      }
    }
  });
  wv.sync();
  const uint32_t Tg = cnt[CW - 1];
  wv.lanes([&](uint32_t lane) {                    // the run leaves in whole lines, as point references (the terms' table stays in the L2;
    const Term* terms = a.bterms + job.begin + t0; // eight lookups in flight per lane: one at a time the loop waited 2 us per entry)
    for (uint32_t i = lane; i < Tg; i += 8 * NL) {
      uint32_t v[8], r[8];
#pragma unroll
      for (uint32_t q = 0; q < 8; ++q) v[q] = i + q * NL < Tg ? buf[i + q * NL] : 0u;
#pragma unroll
      for (uint32_t q = 0; q < 8; ++q) r[q] = terms[v[q] & 0x7FFFu].b;
#pragma unroll
      for (uint32_t q = 0; q < 8; ++q)
        if (i + q * NL < Tg) ix[i + q * NL] = r[q] | ((v[q] & 0x8000u) << 16);
    }
  });
}
MP_BLOCK_KERNEL_OCC(k_bucket_sort, BSplitArgs, body_bucket_sort, 4)

// ---- k_bucket_acc: W = WaveCtx; wv.lds = bk_acc_lds_words(c, gmax, XW, PW)
static inline uint32_t bk_acc_lds_words(uint32_t c, uint32_t gmax, uint32_t xw, uint32_t pw) {
  const uint32_t nbp = 64u * bk_unit_nb(c);
  // offsets of the unit's buckets in every chunk (16-bit, rows of nbp + 2), bucket totals, assignment, staging / exchange
  // (list mode needs none of the tables: its 64 XYZZ slots lie over them)
  return std::max(gmax * (nbp + 2u) / 2u + nbp + nbp + 64u * pw, 64u * xw);
}
// (bucket mode and list mode are two kernels: the full additions of the list walk would otherwise set the register budget of the hot loop
// -- 92 spilled registers on the 14-limb field of BLS12-377; every wave of the launch that finds its item in the other mode leaves at once)
template <class C, class W>
MP_HD void body_bucket_acc(const BSplitArgs& a, uint32_t wave, W& wv) {
  constexpr uint32_t XW = XyzzWords<C>::N;
  const uint32_t NBK = bk_buckets(a.bits), ROW = bk_offs_row(a.bits), NB = bk_unit_nb(a.bits), NBP = 64u * NB, OR = NBP + 2u;
  const uint32_t unit = bk_unit_of_wave(a, wave, a.neq * a.njobs * a.nwin * a.units);
  if (unit == 0xFFFFFFFFu) return;
  const uint32_t it = unit / a.units, p = unit % a.units;
  const BItem id = bk_item(a, it);
  const uint32_t b = id.b;
  const uint32_t G = bk_chunks(a.jobs[id.jb].kpad);
  const uint16_t* og = a.offs + (size_t)it * a.gmax * ROW;
  const uint32_t* ix = a.sorted + (size_t)it * a.kpad_max;
  uint32_t* park = a.park + (size_t)it * NBK * XW;
  uint16_t* o16 = reinterpret_cast<uint16_t*>(wv.lds);      // [G][OR]: o16[g][i] = first position of the unit's bucket i inside run g, i = 0 .. NBP
  uint32_t* tot = wv.lds + a.gmax * OR / 2u;                // [NBP]: terms per bucket (all runs)
  uint32_t* cur = tot + NBP;                                // [NB][64]: worker lane -> home lane of its bucket of class j
  uint32_t* xch = cur + NBP;                                // the staged point (bucket mode)
  const bool list = bk_list_mode(a, it, G);
  const uint32_t* const tile = a.tile ? a.tile + (size_t)b * a.tile_K * Geo<C>::PW : nullptr;
  auto point_of = [&](uint32_t e) -> const uint32_t* {
    if (tile) return tile + (size_t)(e & BK_TILE_MASK) * Geo<C>::PW;
    return a.P + p_off<C>(e & BK_SLOT_MASK, a.Bpad, b + ((e >> 20) & BK_LINK_MASK) * a.link_stride);
  };
  if (list) return;                                // (k_bucket_list)
  typename W::template PL<Xyzz<C>> run;
  {
    // ---- bucket mode: the buckets base .. base + NBP - 1, NB per lane
    const uint32_t base = 1u + p * NBP;
    wv.lanes([&](uint32_t lane) {
      for (uint32_t g = 0; g < G; ++g)
        for (uint32_t i = lane; i <= NBP; i += 64) o16[g * OR + i] = og[(size_t)g * ROW + base + i];
    });
    wv.sync();
    wv.lanes([&](uint32_t lane) {
      for (uint32_t k = lane; k < NBP; k += 64) {
        uint32_t t = 0;
        for (uint32_t g = 0; g < G; ++g) t += (uint32_t)o16[g * OR + k + 1] - (uint32_t)o16[g * OR + k];
        tot[k] = t;
      }
    });
    wv.sync();
    wv.lanes([&](uint32_t lane) {                  // ranks by counting inside the unit; the empty buckets' sums are written right away
      for (uint32_t j = 0; j < NB; ++j) {
        const uint32_t mine = tot[NB * lane + j];
        uint32_t r = 0;
        for (uint32_t l = 0; l < 64; ++l) {
          const uint32_t o = tot[NB * l + j];
          r += (o > mine || (o == mine && l < lane)) ? 1u : 0u;
        }
        cur[j * 64 + ((j & 1u) ? 63u - r : r)] = lane;
        if (mine == 0) xyzz_to_words<C>(xyzz_inf<C>(), park + (size_t)(base - 1u + NB * lane + j) * XW);
      }
    });
    wv.sync();
    typename W::template PL<uint32_t> n, seg, gi, pos, rem, mseg, e0, e1, nb0, nb1;
    auto bucket_of = [&](uint32_t lane, uint32_t j) -> uint32_t { return NB * cur[j * 64 + lane] + j; };      // the lane's local bucket of class j
    wv.lanes([&](uint32_t lane) {
      uint32_t t = 0;
      for (uint32_t j = 0; j < NB; ++j) t += tot[bucket_of(lane, j)];
      n[lane] = t;
    });
    // the position of the next entry of the lane's walk (only called while terms remain): run after run of a bucket, bucket after bucket;
    // fresh: it is the first term of a new bucket
    auto advance = [&](uint32_t lane, uint32_t& fresh) -> uint32_t {
      fresh = 0;
      while (rem[lane] == 0) {
        gi[lane] += 1;
        if (gi[lane] == G) {
          gi[lane] = 0;
          seg[lane] += 1;
          fresh = 1;
        }
        const uint32_t k = bucket_of(lane, seg[lane]), lo = o16[gi[lane] * OR + k];
        pos[lane] = gi[lane] * BK_CHUNK + lo;
        rem[lane] = (uint32_t)o16[gi[lane] * OR + k + 1] - lo;
      }
      pos[lane] += 1;
      rem[lane] -= 1;
      return pos[lane] - 1;
    };
    wv.lanes([&](uint32_t lane) {
      run[lane] = xyzz_inf<C>();
      seg[lane] = 0;
      gi[lane] = 0;
      mseg[lane] = 0;
      {
        const uint32_t k = bucket_of(lane, 0), lo = o16[k];
        pos[lane] = lo;
        rem[lane] = (uint32_t)o16[k + 1] - lo;
      }
      e0[lane] = e1[lane] = nb1[lane] = 0;
      uint32_t first = 0;
      if (n[lane] > 0) e0[lane] = ix[advance(lane, first)];
      if (n[lane] > 0) mseg[lane] = seg[lane];     // the class of the first non-empty bucket
      if (n[lane] > 1) e1[lane] = ix[advance(lane, nb1[lane])];
      nb0[lane] = 0;
      if (n[lane] > 0) wv.template stage<Geo<C>::PW>(xch, point_of(e0[lane]), lane);
    });
    const uint32_t iters = wv.max(n);
#pragma unroll 1
    for (uint32_t i = 0; i < iters; ++i) {
      wv.lanes([&](uint32_t lane) {
        if (i < n[lane]) {
          const uint32_t e = e0[lane];
          if (nb0[lane]) {                         // the bucket before this term is done: park its sum
            xyzz_to_words<C>(run[lane], park + (size_t)(base - 1u + bucket_of(lane, mseg[lane])) * XW);
            run[lane] = xyzz_inf<C>();
            do mseg[lane] += 1;
            while (tot[bucket_of(lane, mseg[lane])] == 0);
          }
          Aff<C> q;
          {
            uint32_t pw[Geo<C>::PW];
            wv.template take<Geo<C>::PW>(xch, pw, lane);
            q.x = fe_unpack<typename C::FqP>(pw);
            q.y = fe_unpack<typename C::FqP>(pw + Geo<C>::FW);
          }
          e0[lane] = e1[lane];
          nb0[lane] = nb1[lane];
          uint32_t at = 0;
          if (i + 2 < n[lane]) at = advance(lane, nb1[lane]);
          e1[lane] = ix[at];
          if (i + 1 < n[lane]) wv.template stage<Geo<C>::PW>(xch, point_of(e0[lane]), lane);      // the point of term i + 1
          xyzz_madd_signed_ip<C>(run[lane], q, (e >> 31) != 0);
        }
      });
    }
    wv.lanes([&](uint32_t lane) {                  // the last sum
      if (n[lane] > 0) xyzz_to_words<C>(run[lane], park + (size_t)(base - 1u + bucket_of(lane, mseg[lane])) * XW);
    });
    return;
  }
}
MP_WAVE_KERNEL_OCC(k_bucket_acc, BSplitArgs, body_bucket_acc, 2)
template <class C, class W>
MP_HD void body_bucket_list(const BSplitArgs& a, uint32_t wave, W& wv) {
  constexpr uint32_t XW = XyzzWords<C>::N;
  const uint32_t NBK = bk_buckets(a.bits), ROW = bk_offs_row(a.bits), NB = bk_unit_nb(a.bits), NBP = 64u * NB, OR = NBP + 2u;
  const uint32_t unit = bk_unit_of_wave(a, wave, a.neq * a.njobs * a.nwin * a.units);
  if (unit == 0xFFFFFFFFu) return;
  const uint32_t it = unit / a.units, p = unit % a.units;
  const BItem id = bk_item(a, it);
  const uint32_t b = id.b;
  const uint32_t G = bk_chunks(a.jobs[id.jb].kpad);
  const uint16_t* og = a.offs + (size_t)it * a.gmax * ROW;
  const uint32_t* ix = a.sorted + (size_t)it * a.kpad_max;
  uint32_t* park = a.park + (size_t)it * NBK * XW;
  uint16_t* o16 = reinterpret_cast<uint16_t*>(wv.lds);      // [G][OR]: o16[g][i] = first position of the unit's bucket i inside run g, i = 0 .. NBP
  uint32_t* tot = wv.lds + a.gmax * OR / 2u;                // [NBP]: terms per bucket (all runs)
  uint32_t* cur = tot + NBP;                                // [NB][64]: worker lane -> home lane of its bucket of class j
  uint32_t* xch = cur + NBP;                                // the staged point (bucket mode)
  const bool list = bk_list_mode(a, it, G);
  const uint32_t* const tile = a.tile ? a.tile + (size_t)b * a.tile_K * Geo<C>::PW : nullptr;
  auto point_of = [&](uint32_t e) -> const uint32_t* {
    if (tile) return tile + (size_t)(e & BK_TILE_MASK) * Geo<C>::PW;
    return a.P + p_off<C>(e & BK_SLOT_MASK, a.Bpad, b + ((e >> 20) & BK_LINK_MASK) * a.link_stride);
  };
  if (!list) return;                               // (k_bucket_acc)
  typename W::template PL<Xyzz<C>> run;
  // ---- list mode: the runs p, p + P, ... of the item, each in 64 equal shares walked from the top; whenever a lane crosses into the
  // next lower bucket it adds the running sum to acc (kept in LDS), so that  sum_t d_t P_t over the share = lo * run + acc.  The entry
  // of term i + 2 and the point of term i + 1 are requested while term i is added
  xch = wv.lds;                                    // the lanes' second accumulator and the final exchange
  wv.lanes([&](uint32_t lane) { xyzz_to_words<C>(xyzz_inf<C>(), xch + lane * XW); });
  for (uint32_t g = p; g < G; g += a.units) {
    const uint16_t* oc = og + (size_t)g * ROW;     // (read where it lies: one item in twenty takes this path)
    const uint32_t* ixc = ix + (size_t)g * BK_CHUNK;
    const uint32_t Tu = oc[NBK + 1];
    typename W::template PL<uint32_t> n, pos, rem, e0, e1;
    typename W::template PL<Aff<C>> q0;
    wv.lanes([&](uint32_t lane) {
      run[lane] = xyzz_inf<C>();
      const uint32_t s0 = (uint32_t)(((uint64_t)Tu * lane) >> 6), s1 = (uint32_t)(((uint64_t)Tu * (lane + 1)) >> 6);
      uint32_t lo_b = 1, hi_b = NBK;               // largest bucket whose first position is <= s1 - 1
      const uint32_t last = s1 ? s1 - 1 : 0;
      while (lo_b < hi_b) {
        const uint32_t mid = (lo_b + hi_b + 1) >> 1;
        if (oc[mid] <= last) lo_b = mid; else hi_b = mid - 1;
      }
      n[lane] = s1 - s0;
      pos[lane] = s1;
      rem[lane] = s1 > s0 ? lo_b : 0u;
      e0[lane] = n[lane] > 0 ? ixc[s1 - 1] : 0u;
      e1[lane] = n[lane] > 1 ? ixc[s1 - 2] : 0u;
      q0[lane] = ld_aff<C>(point_of(e0[lane]));
    });
    const uint32_t iters = wv.max(n);
#pragma unroll 1
    for (uint32_t i = 0; i < iters; ++i) {
      wv.lanes([&](uint32_t lane) {
        if (i < n[lane]) {
          const uint32_t at = pos[lane] - 1 - i, e = e0[lane];
          const Aff<C> q = q0[lane];
          e0[lane] = e1[lane];
          e1[lane] = i + 2 < n[lane] ? ixc[at - 2] : 0u;
          if (i + 1 < n[lane]) q0[lane] = ld_aff<C>(point_of(e0[lane]));
          while (at < oc[rem[lane]]) {             // into the next lower bucket
            Xyzz<C> t = xyzz_from_words<C>(xch + lane * XW);
            xyzz_add_ip<C>(t, run[lane]);
            xyzz_to_words<C>(t, xch + lane * XW);
            rem[lane] -= 1;
          }
          xyzz_madd_signed_ip<C>(run[lane], q, (e >> 31) != 0);
        }
      });
    }
    // acc += lo * run (double-and-add over the bits of the largest lo: wave-uniform trip count)
    const uint32_t lomax = wv.max(rem);
    int nbits = 0;
    while ((lomax >> nbits) != 0) ++nbits;
    typename W::template PL<Xyzz<C>> prod;
    wv.lanes([&](uint32_t lane) { prod[lane] = xyzz_inf<C>(); });
    for (int bit = nbits - 1; bit >= 0; --bit) {
      wv.lanes([&](uint32_t lane) {
        xyzz_dbl_ip<C>(prod[lane]);
        if ((rem[lane] >> bit) & 1u) xyzz_add_ip<C>(prod[lane], run[lane]);
      });
    }
    wv.lanes([&](uint32_t lane) {
      Xyzz<C> t = xyzz_from_words<C>(xch + lane * XW);
      xyzz_add_ip<C>(t, prod[lane]);
      xyzz_to_words<C>(t, xch + lane * XW);
    });
  }
  typename W::template PL<Xyzz<C>> acc;
  wv.sync();
  wv.lanes([&](uint32_t lane) { acc[lane] = xyzz_from_words<C>(xch + lane * XW); });
  wv.sync();
  for (uint32_t s = 32; s >= 1; s >>= 1) {
    wv.lanes([&](uint32_t lane) { xyzz_to_words<C>(acc[lane], xch + lane * XW); });
    wv.sync();
    wv.lanes([&](uint32_t lane) {
      if (lane < s) xyzz_add_ip<C>(acc[lane], xyzz_from_words<C>(xch + (lane + s) * XW));
    });
    wv.sync();
  }
  wv.lanes([&](uint32_t lane) {
    if (lane == 0) xyzz_to_words<C>(acc[lane], park + (size_t)p * XW);
  });
}
MP_WAVE_KERNEL_OCC(k_bucket_list, BSplitArgs, body_bucket_list, 2)

// ---- k_bucket_reduce: W = WaveCtx, FOUR waves per item (x = 4 item + quarter); wv.lds = 64 XW words.
// sum_k k S_k over the item's 2^(c-1) parked sums.  Each wave takes a QUARTER of the buckets, base + 1 .. base + Q, and computes
// A = sum_j j S_(base + j)  and  R = sum_j S_(base + j)  exactly as phase F of the wave kernel does with NB = Q / 64 buckets per lane
// (2 NB - 3 additions, a 6-step suffix scan, log2 NB doublings, a 6-step tree); k_bucket_final then adds up
//   sum_k k S_k = A_0 + A_1 + A_2 + A_3 + Q (R_1 + 2 R_2 + 3 R_3)
// -- five additions, log2 Q doublings and four additions on one lane per item.  (Round 6 first ran one wave per item: 128 buckets per
// lane at 14 bits, 5.5 ms per launch.  The empty buckets hold the point at infinity, so no offsets are read.)
template <class C, class W>
MP_HD void body_bucket_reduce(const BSplitArgs& a, uint32_t x, W& wv) {
  constexpr uint32_t XW = XyzzWords<C>::N;
  const uint32_t NBK = bk_buckets(a.bits), Q = NBK >> 2, NB = Q >> 6;      // (NBK >= 512 on this path: NB >= 2)
  uint32_t LOGNB = 0;
  while ((1u << LOGNB) < NB) ++LOGNB;
  const uint32_t it = x >> 2, quarter = x & 3u;
  const BItem id = bk_item(a, it);
  const uint32_t G = bk_chunks(a.jobs[id.jb].kpad);
  const uint32_t* park = a.park + (size_t)it * NBK * XW;
  uint32_t* out = a.quarters + ((size_t)it * 4 + quarter) * 2 * XW;
  uint32_t* xch = wv.lds;
  typename W::template PL<Xyzz<C>> run, acc;
  const bool list = bk_list_mode(a, it, G);
  if (list) {                                      // list mode: the sums of the item's units, on the first wave
    wv.lanes([&](uint32_t lane) {
      acc[lane] = quarter == 0 && lane < a.units ? xyzz_from_words<C>(park + (size_t)lane * XW) : xyzz_inf<C>();
      run[lane] = xyzz_inf<C>();
    });
  } else {
    wv.lanes([&](uint32_t lane) {                  // R = sum of the lane's NB buckets, A = sum of the first NB - 1 running sums from the top
      const uint32_t* mine = park + ((size_t)quarter * Q + (size_t)NB * lane) * XW;
      run[lane] = xyzz_from_words<C>(mine + (size_t)(NB - 1) * XW);
      for (int j = (int)NB - 2; j >= 0; --j) {
        if (j == (int)NB - 2) acc[lane] = run[lane]; else xyzz_add_ip<C>(acc[lane], run[lane]);
        xyzz_add_ip<C>(run[lane], xyzz_from_words<C>(mine + (size_t)j * XW));
      }
    });
    for (uint32_t s = 1; s < 64; s <<= 1) {        // suffix sums
      wv.lanes([&](uint32_t lane) { xyzz_to_words<C>(run[lane], xch + lane * XW); });
      wv.sync();
      wv.lanes([&](uint32_t lane) {
        if (lane + s < 64) xyzz_add_ip<C>(run[lane], xyzz_from_words<C>(xch + (lane + s) * XW));
      });
      wv.sync();
    }
    // sum_l (NB l + 1) R_l = Suf_0 + NB sum_(l >= 1) Suf_l; Suf_0 = the quarter's R stays in run[] of lane 0
    wv.lanes([&](uint32_t lane) {
      Xyzz<C> t = run[lane];
      if (lane >= 1)
        for (uint32_t q = 0; q < LOGNB; ++q) xyzz_dbl_ip<C>(t);
      xyzz_add_ip<C>(acc[lane], t);
    });
  }
  for (uint32_t s = 32; s >= 1; s >>= 1) {         // tree reduction
    wv.lanes([&](uint32_t lane) { xyzz_to_words<C>(acc[lane], xch + lane * XW); });
    wv.sync();
    wv.lanes([&](uint32_t lane) {
      if (lane < s) xyzz_add_ip<C>(acc[lane], xyzz_from_words<C>(xch + (lane + s) * XW));
    });
    wv.sync();
  }
  wv.lanes([&](uint32_t lane) {
    if (lane == 0) {
      xyzz_to_words<C>(acc[lane], out);
      xyzz_to_words<C>(run[lane], out + XW);
    }
  });
}
MP_WAVE_KERNEL_OCC(k_bucket_reduce, BSplitArgs, body_bucket_reduce, 2)
// x = item: the four quarters -> the window's sum
template <class C>
MP_HD void body_bucket_final(const BSplitArgs& a, uint32_t it, uint32_t) {
  constexpr uint32_t XW = XyzzWords<C>::N;
  const uint32_t Q = bk_buckets(a.bits) >> 2;
  uint32_t LOGQ = 0;
  while ((1u << LOGQ) < Q) ++LOGQ;
  const BItem id = bk_item(a, it);
  const BJob job = a.jobs[id.jb];
  const uint32_t* qs = a.quarters + (size_t)it * 8 * XW;
  Xyzz<C> tot = xyzz_from_words<C>(qs);
  Xyzz<C> r3 = xyzz_from_words<C>(qs + 7 * XW), r2 = xyzz_from_words<C>(qs + 5 * XW), r1 = xyzz_from_words<C>(qs + 3 * XW);
  xyzz_add_ip<C>(r2, r3);                          // R_2 + R_3
  xyzz_add_ip<C>(r1, r2);                          // R_1 + R_2 + R_3
  xyzz_add_ip<C>(r1, r2);
  xyzz_add_ip<C>(r1, r3);                          // R_1 + 2 R_2 + 3 R_3  (list mode: every R is the point at infinity)
#pragma unroll 1
  for (uint32_t q = 0; q < LOGQ; ++q) xyzz_dbl_ip<C>(r1);
  xyzz_add_ip<C>(tot, r1);
#pragma unroll 1
  for (uint32_t q = 1; q < 4; ++q) xyzz_add_ip<C>(tot, xyzz_from_words<C>(qs + 2 * q * XW));
  st_jac<C>(a.J + j_off<C>(job.win_first + id.w, a.Bpad, id.b), xyzz_to_jac<C>(tot));
}
MP_KERNEL_OCC(k_bucket_final, BSplitArgs, body_bucket_final, Geo<C>::OCC4)

// ---- fold the window results: R = sum_w 2^(c w) R_w (x = proof, y = bucket job)
// The same kernel folds the range sums of window-split Straus jobs (layout.hpp vsplit_lo; vb_nwin != 0): job.count parts, part w
// starts at the 5-bit window vsplit_lo(w, job.count, vb_nwin), R = sum_w 2^(5 vsplit_lo(w)) R_w.
struct BFoldArgs {
  uint32_t* J;
  const BJob* jobs;
  uint32_t Bpad, nwin;
  uint32_t vb_nwin;      // 0: bucket windows (nwin parts, `bits` apart); else the Straus windows a split job's parts share
  uint32_t bits;         // the bucket method's window width
};
MP_HD uint32_t fold_parts(const BFoldArgs& a, const BJob& job) { return a.vb_nwin ? job.count : a.nwin; }
// doublings between part w + 1 and part w
MP_HD uint32_t fold_bits(const BFoldArgs& a, const BJob& job, uint32_t w) {
  return a.vb_nwin ? (uint32_t)VB_WINDOW_BITS * (vsplit_lo(w + 1, job.count, a.vb_nwin) - vsplit_lo(w, job.count, a.vb_nwin)) : a.bits;
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
  MP_WAVE_KERNEL_INST(X, k_bucket_sort, BSplitArgs, C)   \
  MP_WAVE_KERNEL_INST(X, k_bucket_acc, BSplitArgs, C)    \
  MP_WAVE_KERNEL_INST(X, k_bucket_list, BSplitArgs, C)   \
  MP_WAVE_KERNEL_INST(X, k_bucket_reduce, BSplitArgs, C) \
  MP_KERNEL_INST(X, k_bucket_final, BSplitArgs, C)       \
  MP_KERNEL_INST(X, k_bucket_fold, BFoldArgs, C)         \
  MP_WAVE_KERNEL_INST(X, k_var_msm_q, VarQuadArgs, C)    \
  MP_WAVE_KERNEL_INST(X, k_bucket_fold_q, BFoldQuadArgs, C) \
  MP_WAVE_KERNEL_INST(X, k_fixed_msm_q, FixedQuadArgs, C)   \
  MP_WAVE_KERNEL_INST(X, k_combine_q, CombineQuadArgs, C)
