#pragma once
// engine_base.hpp -- shared declarations of libmpshuffle.so
// libmpshuffle.so: host orchestration of the gfx950 shuffle-proof engine and its C ABI (include/mpshuffle.h).
//
// A batch of B independent proofs goes through a fixed sequence of kernels on one HIP stream (two for the prover's first stretch
// of batches up to 32 768 proofs: engine_core.hpp prove_dev) with NO host round-trip inside a batch: Fiat-Shamir challenges are derived on the device.  The prover's group work is
// packed into four dependency levels (everything that can be computed between two squeeze points runs in one
// launch of each kernel class), the verifier's into one.
//
//   prove : load -> init(rand, perm) -> remask -> [A: c_A] -> FS x -> scal1 -> [B: c_B, multi-exp msg] -> FS y,z
//           -> scal2 -> [C: c_b, Hadamard, SVP msgs] -> FS hx,hy -> scal3 -> [D: zero-arg msgs] -> FS zx,svx,mx
//           -> scal4 (responses) -> store
//   verify: load -> FS (all challenges) -> scalars (MSM coefficients, direct checks) -> [MSMs == O] -> verdict
//
// Mirrors DLCards::{setup, shuffle_and_remask, verify_shuffle}
// [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:105-121, 380-418, 420-443].
#include <algorithm>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/mpshuffle.h"
#include "hash.hpp"
#include "layout.hpp"

namespace mp {

std::string& last_error();
inline int fail(int code, const std::string& msg) {
  last_error() = msg;
  return code;
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() {}
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { rt::dfree(p); }
  void alloc(size_t count, rt::Stream s, bool zero = true) {
    if (count <= n) return;
    rt::dfree(p);
    p = nullptr;
    p = (T*)rt::dmalloc(count * sizeof(T));
    n = count;
    if (zero) rt::dzero(p, count * sizeof(T), s);
  }
  void upload(const std::vector<T>& v, rt::Stream s) {
    alloc(v.size() ? v.size() : 1, s, false);
    if (!v.empty()) rt::h2d(p, v.data(), v.size() * sizeof(T), s);
  }
};

struct Profiler {
  bool on = false;
  struct Rec {
    const char* name;
    rt::Event a, b;
    uint64_t items;      // threads (waves, for the wave kernels) of the launch
  };
  std::vector<Rec> recs;
  std::vector<rt::Event> pool;
  rt::Event get() {
    if (!pool.empty()) {
      rt::Event e = pool.back();
      pool.pop_back();
      return e;
    }
    return rt::event_create();
  }
  void begin(const char* name, rt::Stream s, uint64_t items = 0) {
    if (!on) return;
    Rec r{name, get(), get(), items};
    rt::event_record(r.a, s);
    recs.push_back(r);
  }
  void end(rt::Stream s) {
    if (!on) return;
    rt::event_record(recs.back().b, s);
  }
  std::string report() {
    std::map<std::string, std::pair<long, double>> acc;
    std::map<std::string, uint64_t> items;
    std::vector<std::string> order;
    for (auto& r : recs) {
      float ms = rt::event_ms(r.a, r.b);
      if (!acc.count(r.name)) order.push_back(r.name);
      acc[r.name].first += 1;
      acc[r.name].second += ms;
      items[r.name] += r.items;
      pool.push_back(r.a);
      pool.push_back(r.b);
    }
    recs.clear();
    std::ostringstream os;
    for (auto& k : order) os << k << " " << acc[k].first << " " << acc[k].second << " " << items[k] << "\n";
    return os.str();
  }
  ~Profiler() {
    for (auto& r : recs) {
      rt::event_destroy(r.a);
      rt::event_destroy(r.b);
    }
    for (auto e : pool) rt::event_destroy(e);
  }
};

}  // namespace mp

struct mp_ctx {
  // Round 6, the threading contract (include/mpshuffle.h): every entry point that takes this context or one of its tables holds this
  // lock for the length of the call, so calls from several host threads on ONE context run one after the other; different contexts
  // share nothing (streams, arenas, tables, profiler are all per context; mp_last_error is per thread) and run side by side.
  // Recursive: the host-buffer entry points call the device-pointer ones.
  std::recursive_mutex mu;
  int curve = 0;
  int device = 0;
  mp::rt::Stream stream{};      // all kernels
  mp::rt::Stream h2d{}, d2h{};  // host-buffer API: uploads and downloads of neighbouring chunks overlap the kernels (= vstream / vside: capi.hip)
  // small and medium batches: the prover's challenge-independent group work (re-encryption, operand sums, window tables) runs on
  // `side` next to the randomness, c_A and the statement hash on `stream` (engine_core.hpp: prove_dev)
  mp::rt::Stream side{};
  mp::rt::Event ev_fork{}, ev_shuf{}, ev_tab{};
  // pipelined verification (mp_set_pipeline): verify calls run on a lane of their own -- stream, side stream, events -- next to the
  // prove calls on the lane above; ev_vin orders a verify call behind everything the main stream held when it was issued
  mp::rt::Stream vstream{}, vside{};
  mp::rt::Event ev_vfork{}, ev_vshuf{}, ev_vtab{}, ev_vin{};
  std::vector<mp_table*> tables;      // the tables of this context (mp_sync completes their deferred verification passes)
  bool dying = false;                 // mp_ctx_destroy was called while tables were alive: the last mp_table_destroy releases the context
  // square-root tables of the curve's base field for on-device point decompression (kernels_decompress.hpp), built on first use
  mp::DevBuf<uint32_t> sq_ghalf, sq_hh, sq_rr, sq_chain;
  uint32_t sq_geom[4] = {0, 0, 0, 0};     // S, w, k, bits of the fixed exponent; S = 0: not built yet
  uint32_t sq_exp[12] = {0};
  mp::Profiler prof;
  // persistent waves of the bucket kernel (kernels_bucket.hpp): 8 per CU -- two workgroups of four, what its registers allow
  uint32_t bk_slots = 0;              // (engine_core.hpp bucket_slots(): BK_WAVES_PER_CU x CUs, queried once)
};
namespace mp {
// for the duration of a pipelined verify call the context's stream / side stream / events ARE the verify lane's
struct LaneSwap {
  mp_ctx* c;
  explicit LaneSwap(mp_ctx* ctx) : c(ctx) { swap(); }
  ~LaneSwap() { swap(); }
  LaneSwap(const LaneSwap&) = delete;
  LaneSwap& operator=(const LaneSwap&) = delete;
  void swap() {
    std::swap(c->stream, c->vstream);
    std::swap(c->side, c->vside);
    std::swap(c->ev_fork, c->ev_vfork);
    std::swap(c->ev_shuf, c->ev_vshuf);
    std::swap(c->ev_tab, c->ev_vtab);
  }
};
}  // namespace mp

// batches of up to this many proofs hash their transcripts with four lanes per BLAKE2s state (kernels_proto.hpp: k_fsq_*): up to
// 512 waves of lone transcript lanes leave half the SIMDs idle and the others waiting 8-10 cycles per instruction
static const uint32_t FSQ_MAX_BATCH = 32768;
// batches of up to this many proofs draw the prover's randomness with a wave per proof (kernels_proto.hpp: k_prove_init_w)
static const uint32_t PROVE_INIT_WAVE_MAX = 32768;      // (the kernel itself: 0.67 -> 0.19 ms at 4 096 proofs, 0.70 -> 0.23 ms at 32 768)
// batches of up to this many proofs overlap the two halves of the prover's first stretch on two streams (prove_dev)
static const uint32_t OVERLAP_MAX_BATCH = 32768;
static const uint32_t SCAL3D_SPLIT_MAX_BATCH = 4096;     // up to here the zero argument's d_k are summed in column ranges first (k_prove_scal3d_part)
#define MP_WAVE_RUN(NAME, C, nwaves, lds_words, args)                             \
  do {                                                                            \
    ctx->prof.begin(#NAME, ctx->stream, (uint64_t)(nwaves));                      \
    MP_WAVE_LAUNCH(NAME, C, ctx->stream, (nwaves), (lds_words), (args));          \
    ctx->prof.end(ctx->stream);                                                   \
  } while (0)
#define MP_RUN(NAME, C, nx, ny, args)                      \
  do {                                                     \
    ctx->prof.begin(#NAME, ctx->stream, (uint64_t)(nx) * (ny)); \
    MP_LAUNCH(NAME, C, ctx->stream, (nx), (ny), (args));   \
    ctx->prof.end(ctx->stream);                            \
  } while (0)

// device staging of the host-buffer entry points (mp_*_batch): two chunks in flight
struct mp_io_stage {
  mp::DevBuf<uint8_t> in0, in1, in2, in3, out0, out1, keys;
  mp::DevBuf<uint32_t> perm;
  mp::DevBuf<int32_t> status;
  mp::rt::Event up = nullptr, done = nullptr, down = nullptr;   // upload finished / kernels finished / download finished
  bool used = false;
};
// A key set (mp_keyset_create): fixed-base window tables of many aggregate keys, built once -- a card server knows the keys
// of its tables long before the shuffles arrive.  Proofs refer to a key by its index in the set.
struct mp_keyset {
  mp_table* owner = nullptr;
  size_t K = 0;
  mp::DevBuf<uint32_t> wire;     // [K][point_bytes / 4]: the keys as they came (gathered per proof for the transcript / the key's own terms)
  mp::DevBuf<uint32_t> FB;       // [K][windows][entries] affine points (the layout of the table context's own fixed-base tables)
  uint32_t bits = 0, windows = 0, entries = 0;
};
struct mp_table {
  mp_ctx* ctx = nullptr;
  mp_io_stage io[2];
  size_t io_chunk = 0;         // proofs per pipelined chunk of the host-buffer entry points (0 = default, mp_set_io_chunk)
  uint32_t m = 0, n = 0, N = 0;
  uint32_t point_bytes = 64;   // wire size of a point on this table's curve (Geo<C>::PB)
  uint32_t fb_bits = 8;        // window width of the fixed-base tables (mp_table_window_bits)
  bool keyless = false;        // created from the parameters alone (mp_table_create_params): keyed entry points only
  uint32_t bucket_bits = 0;       // window width of the bucket method (0 = by the size of the MSM: kernels_bucket.hpp bk_bits_for; mp_set_bucket_bits)
  uint32_t validated = 0;         // MP_VALIDATED_* (mp_set_validated): inputs the caller has validated once already -- their subgroup test is not repeated
  uint32_t bucket_split_bits = 12;   // windows of at least this many bits run sort / additions / reduction as three kernels (kernels_bucket.hpp, round 6; mp_set_bucket_split)
  uint32_t chain_max_links = 0;   // links per chain equation (0 = as many as fit 32 767 points; mp_set_chain_max_links)
  size_t chain_slice = 0;         // tables per pass of chain verification (0 = as many as the free memory holds; mp_set_chain_slice)
  uint32_t chain_group = 0;       // tables per chain equation (0 = by size, as the groups of mp_set_group_verify; 1 = one table each; mp_set_chain_group)
  uint32_t fs_lanes = 0;          // lanes per transcript hash: 1, 4, or 0 = by batch size (mp_set_transcript_lanes)
  uint32_t group_lanes = 0;       // lanes per group operation of the MSM chains: 1, 4, or 0 = by batch size (mp_set_group_lanes)
  int forced_split = -1;          // work split every batch takes: 0 throughput, 1 latency, 2 medium, 3 finest, 4 wide, 5 small; -1 = by batch size (mp_set_work_split)
  int pipeline = 0;               // > 0: device-resident verify calls run on the context's second lane, next to the prove calls, and up
                                  // to this many of their screening verdicts stay unexamined when a call returns (mp_set_pipeline)
  virtual ~mp_table() {}
  virtual void flush() = 0;       // complete deferred verification passes and wait for the verify lane
  virtual void reserve(size_t B) = 0;
  virtual void set_latency_batch(size_t B) = 0;
  virtual void set_merged_verify(bool on) = 0;
  virtual void set_subgroup_check(bool on) = 0;
  virtual void set_bucket_min(uint32_t terms) = 0;
  virtual void set_bucket_bits(uint32_t bits) = 0;
  virtual void set_toom_cook(bool on) = 0;
  virtual void set_group_verify(uint32_t links, size_t min_batch) = 0;
  virtual uint32_t group_size_of(size_t B) const = 0;
  virtual void set_group_refine(uint32_t points, uint32_t min_groups) = 0;
  virtual void set_group_adapt(bool on) = 0;
  virtual uint64_t reverified() const = 0;
  virtual int set_plan_params(int plan, uint32_t fch, uint32_t vch, uint32_t grp, uint32_t nch, uint32_t vsp) = 0;
  virtual void set_plan_thresholds(size_t tiny, size_t small, size_t latency, size_t medium, size_t wide) = 0;
  // keys: nullptr = the table's own aggregate key; otherwise one wire point per proof (device memory).
  // ks / kidx: proof b is made under key kidx[b] (device array) of the key set instead (keys is ignored)
  virtual void prove_dev(size_t B, const uint8_t* decks, const uint8_t* rho, const uint32_t* perm, const uint8_t* seeds,
                         uint8_t* out_decks, uint8_t* out_proofs, int32_t* status, const uint8_t* keys = nullptr,
                         const mp_keyset* ks = nullptr, const uint32_t* kidx = nullptr) = 0;
  virtual void verify_dev(size_t B, const uint8_t* decks, const uint8_t* shuf, const uint8_t* proofs, int32_t* status,
                          const uint8_t* keys = nullptr, const mp_keyset* ks = nullptr, const uint32_t* kidx = nullptr) = 0;
  virtual int keyset_build(mp_keyset& ks, size_t K, const uint8_t* keys_host) = 0;
  // chain verification: T tables x L links, decks [(L + 1)][T], proofs / status / keys [L][T] (device memory)
  virtual void verify_chain_dev(size_t T, uint32_t L, const uint8_t* decks, const uint8_t* proofs, int32_t* status,
                                const uint8_t* keys = nullptr) = 0;
  // bytes of chain workspace a link in flight needs, and the links the workspace holds already (mp_verify_shuffle_chain_dev sizes its passes by them)
  virtual size_t chain_lane_bytes(uint32_t L, bool keyed) = 0;
  virtual size_t chain_lanes_held() const = 0;
  virtual void validate_decks_dev(size_t count, const uint8_t* decks, int32_t* status) = 0;
  virtual uint32_t chain_group_size(size_t T, uint32_t L, bool keyed) const = 0;
  size_t chain_last_slice = 0;    // tables per pass of the last mp_verify_shuffle_chain_dev call (mp_chain_plan)
  virtual void remask_host(size_t count, const uint8_t* cards, const uint8_t* rho, uint8_t* out) = 0;
  virtual void msm_host(size_t n_msm, size_t k, const uint8_t* scalars, const uint8_t* points, uint8_t* out) = 0;
  virtual void commit_host(size_t count, size_t len, const uint8_t* values, const uint8_t* r, uint8_t* out) = 0;
  virtual void sigma_host(bool prove, size_t B, uint32_t nb, const uint8_t* bases, const uint8_t* publics, const uint8_t* witness,
                          const uint8_t* fs_init, const uint8_t* seeds, uint8_t* proofs, int32_t* status) = 0;
  virtual void census(uint64_t* pt, uint64_t* vt, uint64_t* po, uint64_t* vo) = 0;
  virtual void plan_stats(uint64_t out[16]) = 0;
};


namespace mp {
// per-curve factories (one translation unit per curve so that the curves compile in parallel)
#define MP_DECLARE_CURVE(NAME)                                                                                          \
  mp_table* make_table_##NAME(mp_ctx* ctx, uint32_t m, uint32_t n, const uint8_t* params, const uint8_t* pk,           \
                              uint32_t fb_bits, int* rc);                                                              \
  int setup_##NAME(mp_ctx* ctx, uint32_t m, uint32_t n, const uint8_t seed[32], uint8_t* out);                                   \
  long ser_points_##NAME(bool de, size_t count, const uint8_t* in, uint8_t* out);                                                \
  int decompress_dev_##NAME(mp_ctx* ctx, size_t groups, uint32_t per_group, uint32_t prefix, const uint8_t* d_in, uint8_t* d_out, \
                            int32_t* d_status);                                                                                   \
  bool ser_scalars_ok_##NAME(size_t count, const uint8_t* in);
MP_DECLARE_CURVE(Stark)
MP_DECLARE_CURVE(Bn254)
MP_DECLARE_CURVE(Secp256k1)
MP_DECLARE_CURVE(Bls12_377)
}  // namespace mp
