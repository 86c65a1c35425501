// TEST INFRASTRUCTURE ONLY (oracle).  C entry points (ctypes) over the CPU restatement in shuffle.hpp.
// Nothing in the product path (mental-poker_amd/, include/) may link or call this library; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg do (as the checker / the timed CPU baseline).
// Boundary byte formats are those of include/mpshuffle.h.  Curve constants: SURVEY.md App. C.
#include <chrono>
#include <cstdio>

#include "shuffle.hpp"

namespace mpo {
// p, q little-endian 64-bit limbs
const u64 StarkFq::MOD[4] = {0x0000000000000001ULL, 0x0000000000000000ULL, 0x0000000000000000ULL, 0x0800000000000011ULL};
const u64 StarkFr::MOD[4] = {0x1e66a241adc64d2fULL, 0xb781126dcae7b232ULL, 0xffffffffffffffffULL, 0x0800000000000010ULL};
const u64 Stark::B[4] = {0xf4cdfcb99cee9e89ULL, 0x609ad26c15c915c1ULL, 0x150e596d72f7a8c5ULL, 0x06f21413efbe40deULL};
const u64 Stark::GX[4] = {0x3d723d8bc943cfcaULL, 0xdeacfd9b0d1819e0ULL, 0x7beced415a40f0c7ULL, 0x01ef15c18599971bULL};
const u64 Stark::GY[4] = {0x2873000c36e8dc1fULL, 0xde53ecd11abe43a3ULL, 0xb7be4801df46ec62ULL, 0x005668060aa49730ULL};

const u64 Bn254Fq::MOD[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
const u64 Bn254Fr::MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
const u64 Bn254::B[4] = {3, 0, 0, 0};
const u64 Bn254::GX[4] = {1, 0, 0, 0};
const u64 Bn254::GY[4] = {2, 0, 0, 0};

const u64 SecpFq::MOD[4] = {0xfffffffefffffc2fULL, 0xffffffffffffffffULL, 0xffffffffffffffffULL, 0xffffffffffffffffULL};
const u64 SecpFr::MOD[4] = {0xbfd25e8cd0364141ULL, 0xbaaedce6af48a03bULL, 0xfffffffffffffffeULL, 0xffffffffffffffffULL};
const u64 Secp256k1::B[4] = {7, 0, 0, 0};
const u64 Secp256k1::GX[4] = {0x59f2815b16f81798ULL, 0x029bfcdb2dce28d9ULL, 0x55a06295ce870b07ULL, 0x79be667ef9dcbbacULL};
const u64 Secp256k1::GY[4] = {0x9c47d08ffb10d4b8ULL, 0xfd17b448a6855419ULL, 0x5da4fbfc0e1108a8ULL, 0x483ada7726a3c465ULL};
const u64 Bls377Fq::MOD[6] = {0x8508c00000000001ULL, 0x170b5d4430000000ULL, 0x1ef3622fba094800ULL, 0x1a22d9f300f5138fULL, 0xc63b05c06ca1493bULL, 0x01ae3a4617c510eaULL};
const u64 Bls377Fr::MOD[4] = {0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL};
const u64 Bls12_377::B[6] = {1, 0, 0, 0, 0, 0};
const u64 Bls12_377::GX[6] = {0xeab9b16eb21be9efULL, 0xd5481512ffcd394eULL, 0x188282c8bd37cb5cULL, 0x85951e2caa9d41bbULL, 0xc8fc6225bf87ff54ULL, 0x008848defe740a67ULL};
const u64 Bls12_377::GY[6] = {0xfd82de55559c8ea6ULL, 0xc2fe3d3634a9591aULL, 0x6d182ad44fb82305ULL, 0xbd7fb348ca3e52d9ULL, 0x1f674f5d30afeec4ULL, 0x01914a69c5102effULL};
}  // namespace mpo

using namespace mpo;

namespace {

template <class Cv>
struct Api {
  typedef Shuffle<Cv> S;
  typedef typename S::Fr Fr;
  typedef typename S::Pt Pt;
  typedef typename S::Ct Ct;
  typedef typename S::Deck Deck;
  typedef typename S::Params Params;
  static const size_t PW = S::PW;   // wire bytes of a point (64; 96 on BLS12-377)

  static bool load_params(uint32_t m, uint32_t n, const uint8_t* p, Params& pp) {
    pp.m = m;
    pp.n = n;
    bool ok = S::pt_from_wire(p, pp.G);
    pp.ck.resize(n);
    for (uint32_t i = 0; i < n; ++i) ok &= S::pt_from_wire(p + PW * (1 + i), pp.ck[i]);
    ok &= S::pt_from_wire(p + PW * (1 + n), pp.H);
    ok &= S::pt_from_wire(p + PW * (2 + n), pp.gen);
    pp.gsum = S::compute_gsum(pp.ck);
    return ok;
  }
  static void store_params(const Params& pp, uint8_t* p) {
    S::pt_wire(pp.G, p);
    for (uint32_t i = 0; i < pp.n; ++i) S::pt_wire(pp.ck[i], p + PW * (1 + i));
    S::pt_wire(pp.H, p + PW * (1 + pp.n));
    S::pt_wire(pp.gen, p + PW * (2 + pp.n));
  }
  static bool load_deck(const uint8_t* p, size_t N, Deck& d) {
    d.resize(N);
    bool ok = true;
    for (size_t i = 0; i < N; ++i) {
      ok &= S::pt_from_wire(p + 2 * PW * i, d[i].c0);
      ok &= S::pt_from_wire(p + 2 * PW * i + PW, d[i].c1);
    }
    return ok;
  }
  static void store_deck(const Deck& d, uint8_t* p) {
    for (size_t i = 0; i < d.size(); ++i) {
      S::pt_wire(d[i].c0, p + 2 * PW * i);
      S::pt_wire(d[i].c1, p + 2 * PW * i + PW);
    }
  }
  static bool load_scalars(const uint8_t* p, size_t k, std::vector<Fr>& v) {
    v.resize(k);
    bool ok = true;
    for (size_t i = 0; i < k; ++i) ok &= Fr::from_bytes(p + 32 * i, v[i]);
    return ok;
  }

  static int gen_inputs(uint32_t m, uint32_t n, uint64_t seed, uint8_t* params, uint8_t* pk, uint8_t* deck,
                        uint8_t* rho, uint32_t* perm, uint8_t* prover_seed) {
    uint8_t key[32] = {0};
    memcpy(key, &seed, 8);
    ChaChaRng rng(key);
    Params pp = S::setup(m, n, rng);
    store_params(pp, params);
    Fr sk = field_rand<Fr>(rng);
    S::pt_wire(S::mul(sk, pp.G), pk);
    const size_t N = (size_t)m * n;
    Pt g = Pt::generator();
    Deck d(N);
    for (size_t i = 0; i < N; ++i) {
      Fr k1 = field_rand<Fr>(rng), k2 = field_rand<Fr>(rng);
      d[i].c0 = S::mul(k1, g);
      d[i].c1 = S::mul(k2, g);
    }
    store_deck(d, deck);
    for (size_t i = 0; i < N; ++i) field_rand<Fr>(rng).to_bytes(rho + 32 * i);
    for (size_t i = 0; i < N; ++i) perm[i] = (uint32_t)i;
    for (size_t i = N - 1; i >= 1; --i) {
      size_t j = rng.next_u64() % (i + 1);
      uint32_t t = perm[i];
      perm[i] = perm[j];
      perm[j] = t;
    }
    for (int i = 0; i < 4; ++i) {
      uint64_t w = rng.next_u64();
      memcpy(prover_seed + 8 * i, &w, 8);
    }
    return 0;
  }

  static int shuffle_and_remask(uint32_t m, uint32_t n, const uint8_t* params, const uint8_t* pk, const uint8_t* deck,
                                const uint8_t* rho, const uint32_t* perm, const uint8_t* seed, uint8_t* out_deck,
                                uint8_t* out_proof) {
    Params pp;
    Pt pkp;
    Deck d, out;
    std::vector<Fr> rh;
    const size_t N = (size_t)m * n;
    if (!load_params(m, n, params, pp) || !S::pt_from_wire(pk, pkp) || !load_deck(deck, N, d) ||
        !load_scalars(rho, N, rh))
      return -1;
    std::vector<uint32_t> pm(perm, perm + N);
    std::vector<uint8_t> seen(N, 0);
    for (auto v : pm) {
      if (v >= N || seen[v]) return -2;
      seen[v] = 1;
    }
    typename S::Proof pf;
    try {
      S::shuffle_and_remask(pp, pkp, d, rh, pm, seed, out, pf);
    } catch (const std::exception&) {
      return -3;
    }
    store_deck(out, out_deck);
    S::proof_to_bytes(pf, out_proof);
    return 0;
  }

  static int verify_shuffle(uint32_t m, uint32_t n, const uint8_t* params, const uint8_t* pk, const uint8_t* deck,
                            const uint8_t* shuffled, const uint8_t* proof) {
    Params pp;
    Pt pkp;
    Deck d, sh;
    const size_t N = (size_t)m * n;
    if (!load_params(m, n, params, pp) || !S::pt_from_wire(pk, pkp) || !load_deck(deck, N, d) ||
        !load_deck(shuffled, N, sh))
      return -1;
    typename S::Proof pf;
    if (!S::proof_from_bytes(proof, m, n, pf)) return -1;
    return S::verify(pp, pkp, d, sh, pf);
  }

  static int remask_deck(const uint8_t* G, const uint8_t* pk, const uint8_t* deck, size_t N, const uint8_t* rho,
                         const uint32_t* perm, uint8_t* out) {
    Params pp;
    pp.m = pp.n = 0;
    Pt pkp;
    Deck d, o(N);
    std::vector<Fr> rh;
    if (!S::pt_from_wire(G, pp.G) || !S::pt_from_wire(pk, pkp) || !load_deck(deck, N, d) || !load_scalars(rho, N, rh))
      return -1;
    for (size_t i = 0; i < N; ++i) o[i] = S::remask(pp, pkp, d[perm ? perm[i] : i], rh[i]);
    store_deck(o, out);
    return 0;
  }

  static int msm(const uint8_t* scalars, const uint8_t* points, size_t n, int algo, uint8_t* out) {
    std::vector<Fr> s;
    std::vector<Pt> p(n);
    bool ok = load_scalars(scalars, n, s);
    for (size_t i = 0; i < n; ++i) ok &= S::pt_from_wire(points + PW * i, p[i]);
    if (!ok) return -1;
    Jac<Cv> acc = Jac<Cv>::infinity();
    if (algo == 0) {
      acc = msm_pippenger<Cv>(s.data(), p.data(), n);
    } else {
      for (size_t i = 0; i < n; ++i) acc = acc.add(scalar_mul<Cv>(s[i], p[i]));
    }
    S::pt_wire(acc.to_affine(), out);
    return 0;
  }

  static int commit(uint32_t n, const uint8_t* params, const uint8_t* v, size_t len, const uint8_t* r, uint8_t* out) {
    Params pp;
    std::vector<Fr> vs, rs;
    if (!load_params(1, n, params, pp) || !load_scalars(v, len, vs) || !load_scalars(r, 1, rs) || len > n) return -1;
    S::pt_wire(S::commit(pp, vs, rs[0]), out);
    return 0;
  }

  static int fs_challenges(const uint8_t* init, size_t init_len, const uint8_t* absorb, size_t absorb_len, size_t count,
                           uint8_t* out) {
    FsRng fs(init, init_len);
    if (absorb) fs.absorb(std::vector<uint8_t>(absorb, absorb + absorb_len));
    for (size_t i = 0; i < count; ++i) field_rand<Fr>(fs).to_bytes(out + 32 * i);
    return 0;
  }

  static int on_curve(const uint8_t* pt) {
    Pt p;
    if (!S::pt_from_wire(pt, p)) return -1;
    return p.on_curve() ? 1 : 0;
  }

  static int sigma_prove(uint32_t nb, const uint8_t* bases, const uint8_t* publics, const uint8_t* x, const uint8_t* fs_init,
                         size_t fs_len, const uint8_t* seed, uint8_t* out) {
    std::vector<Pt> g(nb), a(nb);
    bool ok = true;
    for (uint32_t i = 0; i < nb; ++i) {
      ok &= S::pt_from_wire(bases + PW * i, g[i]);
      ok &= S::pt_from_wire(publics + PW * i, a[i]);
    }
    Fr xs;
    ok &= Fr::from_bytes(x, xs);
    if (!ok) return -1;
    auto pf = S::sigma_prove(g, a, xs, fs_init, fs_len, seed);
    for (uint32_t i = 0; i < nb; ++i) S::pt_wire(pf.A[i], out + PW * i);
    pf.z.to_bytes(out + PW * nb);
    return 0;
  }
  static int sigma_verify(uint32_t nb, const uint8_t* bases, const uint8_t* publics, const uint8_t* proof, const uint8_t* fs_init,
                          size_t fs_len) {
    std::vector<Pt> g(nb), a(nb);
    typename S::SigmaProof pf;
    pf.A.resize(nb);
    bool ok = true;
    for (uint32_t i = 0; i < nb; ++i) {
      ok &= S::pt_from_wire(bases + PW * i, g[i]);
      ok &= S::pt_from_wire(publics + PW * i, a[i]);
      ok &= S::pt_from_wire(proof + PW * i, pf.A[i]);
    }
    ok &= Fr::from_bytes(proof + PW * nb, pf.z);
    if (!ok) return -1;
    return S::sigma_verify(g, a, pf, fs_init, fs_len) ? 0 : (nb == 1 ? 5 : 6);
  }

  // timed CPU baseline: `iters` prove+verify pairs on inputs gen_inputs(seed + it)
  static int bench(uint32_t m, uint32_t n, uint64_t seed, int iters, double* prove_s, double* verify_s) {
    const size_t N = (size_t)m * n;
    std::vector<uint8_t> params(PW * (n + 3)), pk(PW), deck(2 * PW * N), rho(32 * N), pseed(32), outd(2 * PW * N),
        proof(S::proof_size(m, n));
    std::vector<uint32_t> perm(N);
    *prove_s = *verify_s = 0;
    for (int it = 0; it < iters; ++it) {
      gen_inputs(m, n, seed + it, params.data(), pk.data(), deck.data(), rho.data(), perm.data(), pseed.data());
      auto t0 = std::chrono::steady_clock::now();
      int rc = shuffle_and_remask(m, n, params.data(), pk.data(), deck.data(), rho.data(), perm.data(), pseed.data(),
                                  outd.data(), proof.data());
      auto t1 = std::chrono::steady_clock::now();
      int vc = verify_shuffle(m, n, params.data(), pk.data(), deck.data(), outd.data(), proof.data());
      auto t2 = std::chrono::steady_clock::now();
      if (rc != 0 || vc != 0) return -1;
      *prove_s += std::chrono::duration<double>(t1 - t0).count();
      *verify_s += std::chrono::duration<double>(t2 - t1).count();
    }
    return 0;
  }
};

}  // namespace

#define DISPATCH(curve, call)                   \
  switch (curve) {                              \
    case 0: return Api<Stark>::call;            \
    case 1: return Api<Bn254>::call;            \
    case 2: return Api<Secp256k1>::call;        \
    case 3: return Api<Bls12_377>::call;        \
    default: return -100;                       \
  }

extern "C" {

size_t mpo_proof_size(uint32_t m, uint32_t n) { return Shuffle<Stark>::proof_size(m, n); }   // the 256-bit curves
size_t mpo_point_size(int curve) { return curve == 3 ? Shuffle<Bls12_377>::PW : Shuffle<Stark>::PW; }
size_t mpo_proof_size_curve(int curve, uint32_t m, uint32_t n) {
  return curve == 3 ? Shuffle<Bls12_377>::proof_size(m, n) : Shuffle<Stark>::proof_size(m, n);
}

int mpo_gen_inputs(int curve, uint32_t m, uint32_t n, uint64_t seed, uint8_t* params, uint8_t* pk, uint8_t* deck,
                   uint8_t* rho, uint32_t* perm, uint8_t* prover_seed) {
  DISPATCH(curve, gen_inputs(m, n, seed, params, pk, deck, rho, perm, prover_seed));
}
int mpo_shuffle_and_remask(int curve, uint32_t m, uint32_t n, const uint8_t* params, const uint8_t* pk,
                           const uint8_t* deck, const uint8_t* rho, const uint32_t* perm, const uint8_t* seed,
                           uint8_t* out_deck, uint8_t* out_proof) {
  DISPATCH(curve, shuffle_and_remask(m, n, params, pk, deck, rho, perm, seed, out_deck, out_proof));
}
int mpo_verify_shuffle(int curve, uint32_t m, uint32_t n, const uint8_t* params, const uint8_t* pk,
                       const uint8_t* deck, const uint8_t* shuffled, const uint8_t* proof) {
  DISPATCH(curve, verify_shuffle(m, n, params, pk, deck, shuffled, proof));
}
int mpo_remask_deck(int curve, const uint8_t* G, const uint8_t* pk, const uint8_t* deck, size_t N, const uint8_t* rho,
                    const uint32_t* perm, uint8_t* out) {
  DISPATCH(curve, remask_deck(G, pk, deck, N, rho, perm, out));
}
int mpo_msm(int curve, const uint8_t* scalars, const uint8_t* points, size_t n, int algo, uint8_t* out) {
  DISPATCH(curve, msm(scalars, points, n, algo, out));
}
int mpo_commit(int curve, uint32_t n, const uint8_t* params, const uint8_t* v, size_t len, const uint8_t* r,
               uint8_t* out) {
  DISPATCH(curve, commit(n, params, v, len, r, out));
}
int mpo_fs_challenges(int curve, const uint8_t* init, size_t init_len, const uint8_t* absorb, size_t absorb_len,
                      size_t count, uint8_t* out) {
  DISPATCH(curve, fs_challenges(init, init_len, absorb, absorb_len, count, out));
}
int mpo_on_curve(int curve, const uint8_t* pt) { DISPATCH(curve, on_curve(pt)); }
int mpo_bench(int curve, uint32_t m, uint32_t n, uint64_t seed, int iters, double* prove_s, double* verify_s) {
  DISPATCH(curve, bench(m, n, seed, iters, prove_s, verify_s));
}
int mpo_sigma_prove(int curve, uint32_t nb, const uint8_t* bases, const uint8_t* publics, const uint8_t* x,
                    const uint8_t* fs_init, size_t fs_len, const uint8_t* seed, uint8_t* out) {
  DISPATCH(curve, sigma_prove(nb, bases, publics, x, fs_init, fs_len, seed, out));
}
int mpo_sigma_verify(int curve, uint32_t nb, const uint8_t* bases, const uint8_t* publics, const uint8_t* proof,
                     const uint8_t* fs_init, size_t fs_len) {
  DISPATCH(curve, sigma_verify(nb, bases, publics, proof, fs_init, fs_len));
}
void mpo_blake2s(const uint8_t* in, size_t len, uint8_t out[32]) { Blake2s::digest(in, len, out); }
void mpo_chacha20_block(const uint8_t key[32], uint64_t counter, uint32_t out[16]) {
  uint32_t k[8];
  memcpy(k, key, 32);
  chacha20_block(k, counter, out);
}

}  // extern "C"
