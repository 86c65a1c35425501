// Group operations on FOUR LANES, for batches too small to fill the chip (a single proof above all).
//
// A lone wave issues one instruction every 8-10 cycles, so a chain of ~250 dependent doublings -- every multi-scalar
// multiplication has one -- costs a single proof ~1.3 ms however few terms it has: the prover's one variable-base level and the
// verifier's window fold were 1.26 ms of 3.8 and 1.3 ms of 2.4.  The field products INSIDE a group operation are independent in
// layers (doubling: 3 + 4 + 3 products, mixed addition: 2 + 2 + 3 + 3, full addition: 4 + 4 + 3 + 3), so a QUAD (four adjacent
// lanes) holds the same accumulator in every lane, each lane computes one product of a layer, and the results go round with DPP
// moves (WaveCtx::quad_read, 9 per field element): 3-4 products deep instead of 10-14.  Plain products and differences only -- no
// fused forms: this path is about depth, the throughput kernels (kernels_msm.hpp) about instruction count.  Results are the same
// group elements, so everything downstream (normalisation, transcript, proof bytes) is unchanged.
//
// Written against the wave interface of rt.hpp (sections of per-lane code; a section may read what OTHER lanes wrote in an earlier
// section, never in its own), so the development emulator runs the same source.  `on` flags are the same in the four lanes of a quad.
#pragma once
#include "kernels_bucket.hpp"

namespace mp {

template <class F>
MP_HD Fe<F> fe_pick(bool c, const Fe<F>& a, const Fe<F>& b) {      // c ? a : b, limb by limb (no branch: the lanes of a quad differ in c)
  Fe<F> r;
  constexpr int N = sizeof(r.v) / sizeof(r.v[0]);
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
template <class F>
MP_HD Fe<F> fe_pick4(uint32_t j, const Fe<F>& a0, const Fe<F>& a1, const Fe<F>& a2, const Fe<F>& a3) {
  return fe_pick<F>((j & 2u) != 0, fe_pick<F>((j & 1u) != 0, a3, a2), fe_pick<F>((j & 1u) != 0, a1, a0));
}

// p <- 2 p where on
template <class C, class W>
MP_HD void xyzz_dbl_quad(W& wv, PerLane<Xyzz<C>>& p, const PerLane<uint32_t>& on) {
  typedef typename C::FqP F;
  PerLane<uint32_t> go;
  PerLane<Fe<F>> U, V, M, X3, t;
  wv.lanes([&](uint32_t l) {
    go[l] = on[l] && !fe_is_zero(p[l].ZZ);
    if (go[l] && fe_is_zero(p[l].Y)) {
      p[l].ZZ = fe_zero<F>();
      p[l].ZZZ = fe_zero<F>();
      go[l] = 0;
    }
    if (!go[l]) return;
    const uint32_t j = l & 3u;
    U[l] = fe_dbl<F>(p[l].Y);
    t[l] = fe_sqr<F>(fe_pick4<F>(j, U[l], p[l].X, p[l].ZZ, U[l]));              // U^2 | X^2 | ZZ^2 | -
  });
  wv.lanes([&](uint32_t l) {
    if (!go[l]) return;
    const uint32_t j = l & 3u;
    V[l] = wv.template quad_read<0>(t, l);
    const Fe<F> XX = wv.template quad_read<1>(t, l);
    M[l] = fe_add<F>(fe_dbl<F>(XX), XX);
    if (C::A == 1) M[l] = fe_add<F>(M[l], wv.template quad_read<2>(t, l));      // 3 X^2 + a ZZ^2
  });
  wv.lanes([&](uint32_t l) {
    if (!go[l]) return;
    const uint32_t j = l & 3u;
    t[l] = fe_mul<F>(fe_pick4<F>(j, U[l], p[l].X, V[l], M[l]), fe_pick4<F>(j, V[l], V[l], p[l].ZZ, M[l]));   // W | S | ZZ' | M^2
  });
  PerLane<Fe<F>> t2, Wv;
  wv.lanes([&](uint32_t l) {
    if (!go[l]) return;
    const uint32_t j = l & 3u;
    Wv[l] = wv.template quad_read<0>(t, l);
    const Fe<F> S = wv.template quad_read<1>(t, l);
    X3[l] = fe_sub<F>(fe_sub<F>(wv.template quad_read<3>(t, l), S), S);
    const Fe<F> D = fe_sub<F>(S, X3[l]);
    t2[l] = fe_mul<F>(fe_pick4<F>(j, Wv[l], Wv[l], M[l], Wv[l]), fe_pick4<F>(j, p[l].Y, p[l].ZZZ, D, p[l].Y));   // W Y | ZZZ' | M (S - X3) | -
  });
  wv.lanes([&](uint32_t l) {
    if (!go[l]) return;
    p[l].X = X3[l];
    p[l].Y = fe_sub<F>(wv.template quad_read<2>(t2, l), wv.template quad_read<0>(t2, l));
    p[l].ZZ = wv.template quad_read<2>(t, l);
    p[l].ZZZ = wv.template quad_read<1>(t2, l);
  });
}

// p <- p + q (on = 1) or p - q (on = 2), q affine; 0: nothing
template <class C, class W>
MP_HD void xyzz_madd_quad(W& wv, PerLane<Xyzz<C>>& p, const PerLane<Aff<C>>& q, const PerLane<uint32_t>& on) {
  typedef typename C::FqP F;
  PerLane<uint32_t> go, same;
  PerLane<Fe<F>> Pd, Rr, t, t2, t3, X3;
  wv.lanes([&](uint32_t l) {
    same[l] = 0;
    go[l] = on[l] != 0 && !aff_is_inf<C>(q[l]);
    if (!go[l]) return;
    const Fe<F> qy = on[l] == 2 ? fe_neg<F>(q[l].y) : q[l].y;
    if (fe_is_zero(p[l].ZZ)) {
      p[l].X = q[l].x;
      p[l].Y = qy;
      p[l].ZZ = fe_one<F>();
      p[l].ZZZ = fe_one<F>();
      go[l] = 0;
      return;
    }
    const bool odd = (l & 1u) != 0;
    t[l] = fe_mul<F>(fe_pick<F>(odd, qy, q[l].x), fe_pick<F>(odd, p[l].ZZZ, p[l].ZZ));         // U2 | S2 | U2 | S2
  });
  wv.lanes([&](uint32_t l) {
    if (!go[l]) return;
    Pd[l] = fe_sub<F>(wv.template quad_read<0>(t, l), p[l].X);
    Rr[l] = fe_sub<F>(wv.template quad_read<1>(t, l), p[l].Y);
  });
  wv.lanes([&](uint32_t l) {
    if (!go[l]) return;
    if (fe_is_zero(Pd[l])) {
      if (fe_is_zero(Rr[l])) {
        same[l] = 1;             // P + P
      } else {
        p[l].ZZ = fe_zero<F>();  // P + (-P)
        p[l].ZZZ = fe_zero<F>();
      }
      go[l] = 0;
      return;
    }
    t2[l] = fe_sqr<F>(fe_pick<F>((l & 1u) != 0, Rr[l], Pd[l]));                               // PP | RR | PP | RR
  });
  if (wv.any(same)) xyzz_dbl_quad<C>(wv, p, same);
  wv.lanes([&](uint32_t l) {
    if (!go[l]) return;
    const uint32_t j = l & 3u;
    const Fe<F> PP = wv.template quad_read<0>(t2, l);
    t3[l] = fe_mul<F>(fe_pick4<F>(j, Pd[l], p[l].X, p[l].ZZ, Pd[l]), PP);                        // PPP | Q | ZZ' | -
  });
  wv.lanes([&](uint32_t l) {
    if (!go[l]) return;
    const uint32_t j = l & 3u;
    const Fe<F> PPP = wv.template quad_read<0>(t3, l), Q = wv.template quad_read<1>(t3, l);
    X3[l] = fe_sub<F>(fe_sub<F>(fe_sub<F>(wv.template quad_read<1>(t2, l), PPP), Q), Q);
    const Fe<F> D = fe_sub<F>(Q, X3[l]);
    t[l] = fe_mul<F>(fe_pick4<F>(j, p[l].Y, p[l].ZZZ, Rr[l], p[l].Y), fe_pick4<F>(j, PPP, PPP, D, PPP));   // Y PPP | ZZZ' | R (Q - X3) | -
  });
  wv.lanes([&](uint32_t l) {
    if (!go[l]) return;
    p[l].X = X3[l];
    p[l].Y = fe_sub<F>(wv.template quad_read<2>(t, l), wv.template quad_read<0>(t, l));
    p[l].ZZ = wv.template quad_read<2>(t3, l);
    p[l].ZZZ = wv.template quad_read<1>(t, l);
  });
}

// p <- p + q where on, both XYZZ
template <class C, class W>
MP_HD void xyzz_add_quad(W& wv, PerLane<Xyzz<C>>& p, const PerLane<Xyzz<C>>& q, const PerLane<uint32_t>& on) {
  typedef typename C::FqP F;
  PerLane<uint32_t> go, same;
  PerLane<Fe<F>> Pd, Rr, t, t2, t3, X3;
  wv.lanes([&](uint32_t l) {
    same[l] = 0;
    go[l] = on[l] && !fe_is_zero(q[l].ZZ);
    if (!go[l]) return;
    if (fe_is_zero(p[l].ZZ)) {
      p[l] = q[l];
      go[l] = 0;
      return;
    }
    const uint32_t j = l & 3u;
    t[l] = fe_mul<F>(fe_pick4<F>(j, p[l].X, q[l].X, p[l].Y, q[l].Y), fe_pick4<F>(j, q[l].ZZ, p[l].ZZ, q[l].ZZZ, p[l].ZZZ));   // U1 | U2 | S1 | S2
  });
  wv.lanes([&](uint32_t l) {
    if (!go[l]) return;
    Pd[l] = fe_sub<F>(wv.template quad_read<1>(t, l), wv.template quad_read<0>(t, l));
    Rr[l] = fe_sub<F>(wv.template quad_read<3>(t, l), wv.template quad_read<2>(t, l));
  });
  wv.lanes([&](uint32_t l) {
    if (!go[l]) return;
    if (fe_is_zero(Pd[l])) {
      if (fe_is_zero(Rr[l])) {
        same[l] = 1;
      } else {
        p[l].ZZ = fe_zero<F>();
        p[l].ZZZ = fe_zero<F>();
      }
      go[l] = 0;
      return;
    }
    const uint32_t j = l & 3u;
    t2[l] = fe_mul<F>(fe_pick4<F>(j, Pd[l], Rr[l], p[l].ZZ, p[l].ZZZ), fe_pick4<F>(j, Pd[l], Rr[l], q[l].ZZ, q[l].ZZZ));   // PP | RR | ZZ1 ZZ2 | ZZZ1 ZZZ2
  });
  if (wv.any(same)) xyzz_dbl_quad<C>(wv, p, same);
  wv.lanes([&](uint32_t l) {
    if (!go[l]) return;
    const uint32_t j = l & 3u;
    const Fe<F> PP = wv.template quad_read<0>(t2, l);
    t3[l] = fe_mul<F>(fe_pick4<F>(j, Pd[l], wv.template quad_read<0>(t, l), wv.template quad_read<2>(t2, l), Pd[l]), PP);   // PPP | Q | ZZ' | -
  });
  PerLane<Fe<F>> t4;
  wv.lanes([&](uint32_t l) {
    if (!go[l]) return;
    const uint32_t j = l & 3u;
    const Fe<F> PPP = wv.template quad_read<0>(t3, l), Q = wv.template quad_read<1>(t3, l);
    X3[l] = fe_sub<F>(fe_sub<F>(fe_sub<F>(wv.template quad_read<1>(t2, l), PPP), Q), Q);
    const Fe<F> D = fe_sub<F>(Q, X3[l]);
    t4[l] = fe_mul<F>(fe_pick4<F>(j, wv.template quad_read<2>(t, l), wv.template quad_read<3>(t2, l), Rr[l], PPP),
                      fe_pick4<F>(j, PPP, PPP, D, PPP));                                                                  // S1 PPP | ZZZ' | R (Q - X3) | -
  });
  wv.lanes([&](uint32_t l) {
    if (!go[l]) return;
    p[l].X = X3[l];
    p[l].Y = fe_sub<F>(wv.template quad_read<2>(t4, l), wv.template quad_read<0>(t4, l));
    p[l].ZZ = wv.template quad_read<2>(t3, l);
    p[l].ZZZ = wv.template quad_read<1>(t4, l);
  });
}

// ---- Straus on quads: lanes 4 k .. 4 k + 3 of a wave run job (item / B) of proof (item % B), item = 16 wave + k ------------------
struct VarQuadArgs {
  VarArgs v;
  uint32_t B, njobs;
};
template <class C, class W>
MP_HD void body_var_msm_q(const VarQuadArgs& a, uint32_t wid, W& wv) {
  const uint32_t nitems = a.B * a.njobs;
  PerLane<Xyzz<C>> acc;
  PerLane<Aff<C>> q;
  PerLane<uint32_t> live, on;
  uint32_t maxcount = 0;
  wv.lanes([&](uint32_t l) {
    const uint32_t item = wid * 16u + (l >> 2);
    live[l] = item < nitems;
    acc[l] = xyzz_inf<C>();
  });
  // (the jobs of a wave may differ in length: every quad runs the longest, idle where it has no term)
  // (window-split jobs, a.v.split > 1: njobs counts (job, range) pairs; the quads of a wave step through their ranges together --
  // step i is window lo(r + 1) - 1 - i of the quad's own range r, ranges differ in length by at most one window)
  const uint32_t sp = a.v.split > 1 ? a.v.split : 1u;
  uint32_t maxwin = 0;
  for (uint32_t k = 0; k < 16; ++k) {
    const uint32_t item = wid * 16u + k;
    if (item >= nitems) continue;
    const uint32_t jr = item / a.B, r = jr % sp;
    maxcount = a.v.jobs[jr / sp].count > maxcount ? a.v.jobs[jr / sp].count : maxcount;
    const uint32_t nw = sp > 1 ? vsplit_lo(r + 1, sp, a.v.nwin) - vsplit_lo(r, sp, a.v.nwin) : a.v.nwin;
    maxwin = nw > maxwin ? nw : maxwin;
  }
  PerLane<uint32_t> wtop, wcnt, dblon;
  wv.lanes([&](uint32_t l) {
    wtop[l] = 0;
    wcnt[l] = 0;
    if (!live[l]) return;
    const uint32_t jr = (wid * 16u + (l >> 2)) / a.B, r = jr % sp;
    const uint32_t lo = sp > 1 ? vsplit_lo(r, sp, a.v.nwin) : 0u, hi = sp > 1 ? vsplit_lo(r + 1, sp, a.v.nwin) : a.v.nwin;
    wtop[l] = hi - 1;
    wcnt[l] = hi - lo;
  });
#pragma unroll 1
  for (uint32_t i = 0; i < maxwin; ++i) {
    if (i != 0) {
      wv.lanes([&](uint32_t l) { dblon[l] = live[l] && i < wcnt[l]; });
#pragma unroll 1
      for (int d = 0; d < VB_WINDOW_BITS; ++d) xyzz_dbl_quad<C>(wv, acc, dblon);
    }
#pragma unroll 1
    for (uint32_t t = 0; t < maxcount; ++t) {
      wv.lanes([&](uint32_t l) {
        on[l] = 0;
        if (!live[l] || i >= wcnt[l]) return;
        const uint32_t item = wid * 16u + (l >> 2), b = item % a.B;
        const Job job = a.v.jobs[(item / a.B) / sp];
        if (t >= job.count) return;
        const Term term = a.v.terms[job.begin + t];
        const uint32_t w = wtop[l] - i;
        const int d = a.v.D[((size_t)term.s * a.v.nwin + w) * a.v.Bpad + b];
        if (d == 0) return;
        const uint32_t e = (uint32_t)(d < 0 ? -d : d) - 1;
        q[l] = ld_aff<C>(a.v.T + p_off<C>(term.b * VB_ENTRIES + e, a.v.Bpad, b));
        on[l] = d < 0 ? 2u : 1u;
      });
      if (wv.any(on)) xyzz_madd_quad<C>(wv, acc, q, on);
    }
  }
  wv.lanes([&](uint32_t l) {
    if (!live[l] || (l & 3u) != 0) return;
    const uint32_t item = wid * 16u + (l >> 2), b = item % a.B, jr = item / a.B;
    st_jac<C>(a.v.J + j_off<C>(a.v.jobs[jr / sp].out + jr % sp, a.v.Bpad, b), xyzz_to_jac<C>(acc[l]));
  });
}
MP_WAVE_KERNEL(k_var_msm_q, VarQuadArgs, body_var_msm_q)

// ---- fixed-base sums and the sums of partial results on quads (same items: job * B + proof) ----------------------------------------
struct FixedQuadArgs {
  FixedArgs f;
  uint32_t B, njobs;
};
template <class C, class W>
MP_HD void body_fixed_msm_q(const FixedQuadArgs& a, uint32_t wid, W& wv) {
  typedef typename C::FrP R;
  const uint32_t nitems = a.B * a.njobs;
  PerLane<Xyzz<C>> acc;
  PerLane<Aff<C>> q;
  PerLane<uint32_t> live, on;
  struct Words {
    uint32_t k[8];
  };
  PerLane<Words> sc;
  uint32_t maxcount = 0;
  wv.lanes([&](uint32_t l) {
    live[l] = wid * 16u + (l >> 2) < nitems;
    acc[l] = xyzz_inf<C>();
  });
  for (uint32_t k = 0; k < 16; ++k) {
    const uint32_t item = wid * 16u + k;
    if (item < nitems) maxcount = a.f.jobs[item / a.B].count > maxcount ? a.f.jobs[item / a.B].count : maxcount;
  }
#pragma unroll 1
  for (uint32_t t = 0; t < maxcount; ++t) {
    wv.lanes([&](uint32_t l) {
      if (!live[l]) return;
      const uint32_t item = wid * 16u + (l >> 2);
      const Job job = a.f.jobs[item / a.B];
      if (t >= job.count) return;
      fe_to_canonical<R>(ld_fe<R>(a.f.S + s_off(a.f.terms[job.begin + t].s, a.f.Sbpad, item % a.B)), sc[l].k);
    });
#pragma unroll 1
    for (uint32_t w = 0; w < a.f.g.windows; ++w) {
      wv.lanes([&](uint32_t l) {
        on[l] = 0;
        if (!live[l]) return;
        const uint32_t item = wid * 16u + (l >> 2);
        const Job job = a.f.jobs[item / a.B];
        if (t >= job.count) return;
        const uint32_t d = fb_digit(sc[l].k, a.f.g, w);
        if (!d) return;
        q[l] = ld_aff<C>(fb_entry<C>(a.f.FB, a.f.g, a.f.terms[job.begin + t].b, w, d));
        on[l] = 1;
      });
      if (wv.any(on)) xyzz_madd_quad<C>(wv, acc, q, on);
    }
  }
  wv.lanes([&](uint32_t l) {
    if (!live[l] || (l & 3u) != 0) return;
    const uint32_t item = wid * 16u + (l >> 2);
    st_jac<C>(a.f.J + j_off<C>(a.f.jobs[item / a.B].out, a.f.Bpad, item % a.B), xyzz_to_jac<C>(acc[l]));
  });
}
MP_WAVE_KERNEL(k_fixed_msm_q, FixedQuadArgs, body_fixed_msm_q)

template <class C>
MP_HD Xyzz<C> xyzz_from_jac(const Jac<C>& j) {      // (X, Y, Z) -> (X, Y, Z^2, Z^3); Z = 0 stays infinity
  typedef typename C::FqP F;
  Xyzz<C> p;
  p.X = j.X;
  p.Y = j.Y;
  p.ZZ = fe_sqr<F>(j.Z);
  p.ZZZ = fe_mul<F>(p.ZZ, j.Z);
  return p;
}
struct CombineQuadArgs {
  CombineArgs c;
  uint32_t B, njobs;
};
template <class C, class W>
MP_HD void body_combine_q(const CombineQuadArgs& a, uint32_t wid, W& wv) {
  typedef typename C::FqP F;
  const uint32_t nitems = a.B * a.njobs;
  PerLane<Xyzz<C>> acc, r;
  PerLane<Aff<C>> q;
  PerLane<uint32_t> live, ona, onj;
  uint32_t maxcount = 0;
  wv.lanes([&](uint32_t l) {
    live[l] = wid * 16u + (l >> 2) < nitems;
    acc[l] = xyzz_inf<C>();
  });
  for (uint32_t k = 0; k < 16; ++k) {
    const uint32_t item = wid * 16u + k;
    if (item < nitems) maxcount = a.c.jobs[item / a.B].count > maxcount ? a.c.jobs[item / a.B].count : maxcount;
  }
#pragma unroll 1
  for (uint32_t t = 0; t < maxcount; ++t) {
    wv.lanes([&](uint32_t l) {
      ona[l] = 0;
      onj[l] = 0;
      if (!live[l]) return;
      const uint32_t item = wid * 16u + (l >> 2), b = item % a.B;
      const Job job = a.c.jobs[item / a.B];
      if (t >= job.count) return;
      const uint32_t s = a.c.terms[job.begin + t].s;
      if (s & AFF_FLAG) {
        q[l] = ld_aff<C>(a.c.P + p_off<C>(s & SLOT_MASK, a.c.Bpad, b));
        ona[l] = (s & NEG_FLAG) ? 2u : 1u;
      } else {
        Jac<C> j = ld_jac<C>(a.c.J + j_off<C>(s & SLOT_MASK, a.c.Bpad, b));
        if (s & NEG_FLAG) j.Y = fe_neg<F>(j.Y);
        r[l] = xyzz_from_jac<C>(j);
        onj[l] = 1;
      }
    });
    if (wv.any(ona)) xyzz_madd_quad<C>(wv, acc, q, ona);
    if (wv.any(onj)) xyzz_add_quad<C>(wv, acc, r, onj);
  }
  wv.lanes([&](uint32_t l) {
    if (!live[l] || (l & 3u) != 0) return;
    const uint32_t item = wid * 16u + (l >> 2);
    st_jac<C>(a.c.J + j_off<C>(a.c.jobs[item / a.B].out, a.c.Bpad, item % a.B), xyzz_to_jac<C>(acc[l]));
  });
}
MP_WAVE_KERNEL(k_combine_q, CombineQuadArgs, body_combine_q)

// ---- the window fold of the bucket method on quads: R = sum_w 2^(8w) R_w, item = bucket job * B + proof ---------------------------
struct BFoldQuadArgs {
  BFoldArgs f;
  uint32_t B, njobs;
};
template <class C, class W>
MP_HD void body_bucket_fold_q(const BFoldQuadArgs& a, uint32_t wid, W& wv) {
  const uint32_t nitems = a.B * a.njobs;
  PerLane<Xyzz<C>> acc, r;
  PerLane<uint32_t> live;
  wv.lanes([&](uint32_t l) {
    const uint32_t item = wid * 16u + (l >> 2);
    live[l] = item < nitems;
    acc[l] = xyzz_inf<C>();
    if (!live[l]) return;
    const BJob job = a.f.jobs[item / a.B];
    acc[l] = xyzz_from_jac<C>(ld_jac<C>(a.f.J + j_off<C>(job.win_first + fold_parts(a.f, job) - 1, a.f.Bpad, item % a.B)));
  });
  // (the jobs of a launch have the same number of parts and the same spacing: bucket windows, or the ranges of one split factor)
  const BJob job0 = a.f.jobs[0];
  const uint32_t parts = fold_parts(a.f, job0);
#pragma unroll 1
  for (int w = (int)parts - 2; w >= 0; --w) {
    const uint32_t nd = fold_bits(a.f, job0, (uint32_t)w);
#pragma unroll 1
    for (uint32_t d = 0; d < nd; ++d) xyzz_dbl_quad<C>(wv, acc, live);
    wv.lanes([&](uint32_t l) {
      if (!live[l]) return;
      const uint32_t item = wid * 16u + (l >> 2);
      const BJob job = a.f.jobs[item / a.B];
      r[l] = xyzz_from_jac<C>(ld_jac<C>(a.f.J + j_off<C>(job.win_first + (uint32_t)w, a.f.Bpad, item % a.B)));
    });
    xyzz_add_quad<C>(wv, acc, r, live);
  }
  wv.lanes([&](uint32_t l) {
    if (!live[l] || (l & 3u) != 0) return;
    const uint32_t item = wid * 16u + (l >> 2);
    st_jac<C>(a.f.J + j_off<C>(a.f.jobs[item / a.B].out, a.f.Bpad, item % a.B), xyzz_to_jac<C>(acc[l]));
  });
}
MP_WAVE_KERNEL(k_bucket_fold_q, BFoldQuadArgs, body_bucket_fold_q)

}  // namespace mp
