// capi.hip -- the C ABI of libmpshuffle.so (include/mpshuffle.h): argument checking, error mapping and
// dispatch to the per-curve engines (curve_*.hip).  Mirrors DLCards::{setup, shuffle_and_remask, verify_shuffle}
// [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:105-121, 380-418, 420-443].
#include <algorithm>
#include <cstring>

#include "engine_base.hpp"

namespace mp {
std::string& last_error() {
  static thread_local std::string e;
  return e;
}
static inline size_t proof_size_bytes_(uint32_t m, uint32_t n) { return proof_size_bytes(m, n); }
}  // namespace mp

using namespace mp;

// the context's lock for the length of the call + its device for this host thread (engine_base.hpp mp_ctx::mu)
#define MP_ENTER(CTX)                                          \
  std::lock_guard<std::recursive_mutex> mp_lock_((CTX)->mu);   \
  rt::set_device((CTX)->device)
#define MP_TRY try {
#define MP_CATCH                                                                     \
  }                                                                                  \
  catch (const std::invalid_argument& e) { return fail(MP_ERR_BAD_ENCODING, e.what()); } \
  catch (const std::exception& e) { return fail(MP_ERR_INTERNAL, e.what()); }

// ---- canonical (arkworks-0.3 compressed) serialisation of the trait's associated types: host work, no context ---------------
namespace {
long ser_points(int curve, bool de, size_t count, const uint8_t* in, uint8_t* out) {
  switch (curve) {
    case 0: return ser_points_Stark(de, count, in, out);
    case 1: return ser_points_Bn254(de, count, in, out);
    case 3: return ser_points_Bls12_377(de, count, in, out);
    default: return ser_points_Secp256k1(de, count, in, out);
  }
}
bool ser_scalars_ok(int curve, size_t count, const uint8_t* in) {
  switch (curve) {
    case 0: return ser_scalars_ok_Stark(count, in);
    case 1: return ser_scalars_ok_Bn254(count, in);
    case 3: return ser_scalars_ok_Bls12_377(count, in);
    default: return ser_scalars_ok_Secp256k1(count, in);
  }
}
const int kCurveBits[4] = {252, 254, 256, 377};
// element groups of the shuffle proof in wire order: {points?, count, is_vec, vec_len}
struct Group {
  bool point;
  uint32_t count;
  bool vec;
  uint32_t vec_len;
};
std::vector<Group> proof_schema(uint32_t m, uint32_t n) {
  return {
      {true, m, true, m}, {true, m, true, m}, {true, 1, false, 0}, {true, m, true, m},
      {true, 1, false, 0}, {true, 1, false, 0}, {true, 2 * m + 1, true, 2 * m + 1},
      {false, n, true, n}, {false, n, true, n}, {false, 1, false, 0}, {false, 1, false, 0}, {false, 1, false, 0},
      {true, 1, false, 0}, {true, 1, false, 0}, {true, 1, false, 0},
      {false, n, true, n}, {false, n, true, n}, {false, 1, false, 0}, {false, 1, false, 0},
      {true, 1, false, 0}, {true, 2 * m, true, 2 * m}, {true, 4 * m, true, 2 * m},      // E: 2m ciphertexts = 4m points
      {false, n, true, n}, {false, 1, false, 0}, {false, 1, false, 0}, {false, 1, false, 0}, {false, 1, false, 0}};
}
void put_u64(uint8_t* p, uint64_t v) { memcpy(p, &v, 8); }
uint64_t get_u64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
bool curve_ok(int c) { return c >= 0 && c <= 3; }
// the host-buffer entry points and chain verification order their own transfers around verify_dev: not pipelined
struct NoPipeline {
  mp_table* t;
  int keep;
  explicit NoPipeline(mp_table* tt) : t(tt), keep(tt->pipeline) {
    if (keep) t->flush();
    t->pipeline = 0;
  }
  ~NoPipeline() { t->pipeline = keep; }
};
}  // namespace

extern "C" {

const char* mp_last_error(void) { return mp::last_error().c_str(); }
const char* mp_check_name(int code) {
  switch (code) {
    case 0: return "Ok";
    case 1: return "Hadamard Product (5.1)";
    case 2: return "Zero Argument (5.2)";
    case 3: return "Single Value Product (5.3)";
    case 4: return "Multi-Exponentiation Argument (4)";
    case 5: return "Schnorr Identification";
    case 6: return "Chaum-Pedersen";
    case MP_ERR_BAD_ENCODING: return "IoError: bad encoding";
    case MP_ERR_BAD_PERMUTATION: return "IoError: not a permutation";
    case MP_ERR_BAD_ARGUMENT: return "IoError: bad argument";
    case MP_ERR_NO_DEVICE: return "IoError: no MI355X device";
    default: return "IoError: internal";
  }
}
size_t mp_proof_size(uint32_t m, uint32_t n) { return proof_size_bytes(m, n); }
size_t mp_params_size(uint32_t n) { return (size_t)(n + 3) * 64; }
size_t mp_point_size(int curve_id) { return curve_id == MP_CURVE_BLS12_377 ? 96 : 64; }
size_t mp_proof_size_curve(int curve_id, uint32_t m, uint32_t n) { return proof_size_bytes(m, n, (uint32_t)mp_point_size(curve_id)); }
size_t mp_params_size_curve(int curve_id, uint32_t n) { return (size_t)(n + 3) * mp_point_size(curve_id); }

int mp_ctx_create(int curve_id, int device, mp_ctx** out) {
  if (!out) return fail(MP_ERR_BAD_ARGUMENT, "null out pointer");
  if (curve_id < 0 || curve_id > 3) return fail(MP_ERR_BAD_ARGUMENT, "unknown curve id");
  MP_TRY
  int ndev = rt::device_count();
  if (ndev <= 0 || device < 0 || device >= ndev)
    return fail(MP_ERR_NO_DEVICE, "no HIP device: libmpshuffle has no CPU path (runtime " MP_RT_NAME ")");
  rt::set_device(device);
  mp_ctx* c = new mp_ctx();
  c->curve = curve_id;
  c->device = device;
  // HIP streams become hardware queues in the order they are created, the queues sit on the compute pipes round robin (four of them), and
  // two streams overlap fully only on different pipes -- measured, not documented: with the verify lane's streams created fifth and sixth
  // (GPU_MAX_HW_QUEUES=8) or on first use, a pipelined 1 024-proof step ran at 172-176 k proofs/s; created second and fourth, at 198-202 k
  // with 4 or 8 queues.  So a context has exactly four streams, the two lanes interleaved; the host-buffer entry points -- which do not
  // pipeline their verify calls -- run their uploads and downloads on the verify lane's two streams (with copy streams of their own,
  // fifth and sixth, they lost 2-9 %).
  c->stream = rt::stream_create();
  c->vstream = rt::stream_create();
  c->side = rt::stream_create();
  c->vside = rt::stream_create();
  c->h2d = c->vstream;
  c->d2h = c->vside;
  c->ev_fork = rt::event_create();
  c->ev_shuf = rt::event_create();
  c->ev_tab = rt::event_create();
  c->ev_vfork = rt::event_create();
  c->ev_vshuf = rt::event_create();
  c->ev_vtab = rt::event_create();
  c->ev_vin = rt::event_create();
  *out = c;
  return MP_OK;
  MP_CATCH
}
static void ctx_release(mp_ctx* ctx) {
  for (rt::Stream st : {ctx->stream, ctx->side, ctx->vstream, ctx->vside})      // (h2d / d2h are the verify lane's)
    if (st) rt::stream_destroy(st);
  for (rt::Event e : {ctx->ev_fork, ctx->ev_shuf, ctx->ev_tab, ctx->ev_vfork, ctx->ev_vshuf, ctx->ev_vtab, ctx->ev_vin})
    if (e) rt::event_destroy(e);
  delete ctx;
}
void mp_ctx_destroy(mp_ctx* ctx) {
  if (!ctx) return;
  // tables hold a pointer to their context (streams, events, the registry mp_sync walks): a context that still has tables is only
  // marked and goes with the last of them (mp_table_destroy), so the two destroy calls may come in either order
  {
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    if (!ctx->tables.empty()) {
      ctx->dying = true;
      return;
    }
  }
  ctx_release(ctx);
}
int mp_setup(mp_ctx* ctx, uint32_t m, uint32_t n, const uint8_t seed[32], uint8_t* out_params) {
  if (!ctx || !seed || !out_params || m < 2 || n < 2) return fail(MP_ERR_BAD_ARGUMENT, "mp_setup: bad argument");
  MP_TRY
  MP_ENTER(ctx);
  switch (ctx->curve) {
    case 0: return setup_Stark(ctx, m, n, seed, out_params);
    case 1: return setup_Bn254(ctx, m, n, seed, out_params);
    case 3: return setup_Bls12_377(ctx, m, n, seed, out_params);
    default: return setup_Secp256k1(ctx, m, n, seed, out_params);
  }
  MP_CATCH
}

// mp_table_create: the widest fixed-base windows whose tables fit the HBM that is free right now.  A table of b-bit windows holds
// (n + 5) bases x ceil(scalar bits / b) windows x (2^b - 1) affine points; building it needs ~3.3x that for a moment (Jacobian
// entries and inversion scratch before the normalisation).  Rule: table <= 30 % of the free memory and the build <= 85 % of it.
static uint32_t auto_window_bits(int curve, uint32_t n) {
  size_t free_b = 0, total_b = 0;
  rt::mem_info(&free_b, &total_b);
  const uint32_t sbits[4] = {252, 254, 256, 253};
  const size_t pbytes = curve == MP_CURVE_BLS12_377 ? 96 : 64;
  for (uint32_t bits : {21u, 20u, 16u}) {
    if (bits == 21 && sbits[curve] > 252) continue;      // 21 bits only save a window on the 252-bit STARK scalars
    const double tbl = (double)(n + 5) * ((sbits[curve] + bits - 1) / bits) * (double)((1u << bits) - 1u) * (double)pbytes;
    if (tbl <= 0.30 * (double)free_b && 3.3 * tbl <= 0.85 * (double)free_b) return bits;
  }
  return 8;
}
int mp_table_create(mp_ctx* ctx, uint32_t m, uint32_t n, const uint8_t* params, const uint8_t* shared_key, mp_table** out) {
  return mp_table_create_ex(ctx, m, n, params, shared_key, 0, out);
}
int mp_table_create_ex(mp_ctx* ctx, uint32_t m, uint32_t n, const uint8_t* params, const uint8_t* shared_key,
                       uint32_t fb_window_bits, mp_table** out) {
  if (!ctx || !params || !shared_key || !out) return fail(MP_ERR_BAD_ARGUMENT, "mp_table_create: null pointer");
  if (m < 2 || n < 2 || (uint64_t)m * n > 4096) return fail(MP_ERR_BAD_ARGUMENT, "mp_table_create: need m >= 2, n >= 2, m*n <= 4096");
  MP_TRY
  MP_ENTER(ctx);
  const bool auto_bits = fb_window_bits == 0;
  if (auto_bits) fb_window_bits = auto_window_bits(ctx->curve, n);
  int rc = MP_OK;
  mp_table* t = nullptr;
  auto make = [&](uint32_t bits) {
    switch (ctx->curve) {
      case 0: return make_table_Stark(ctx, m, n, params, shared_key, bits, &rc);
      case 1: return make_table_Bn254(ctx, m, n, params, shared_key, bits, &rc);
      case 3: return make_table_Bls12_377(ctx, m, n, params, shared_key, bits, &rc);
      default: return make_table_Secp256k1(ctx, m, n, params, shared_key, bits, &rc);
    }
  };
  if (!auto_bits) {
    t = make(fb_window_bits);
  } else {
    // the width was chosen from ONE snapshot of the free memory: other ranks / contexts on this GPU may have taken it since (the
    // build needs ~3.3x the table for a moment).  A build that runs out of memory is retried with the next narrower windows
    for (;;) {
      try {
        t = make(fb_window_bits);
        break;
      } catch (const std::exception& e) {
        rt::clear_error();
        // (only a failed allocation is worth narrower windows: anything else would fail again)
        if (fb_window_bits <= 8 || (!strstr(e.what(), "hipMalloc") && !strstr(e.what(), "memory"))) throw;
        fb_window_bits = fb_window_bits > 20 ? 20 : (fb_window_bits > 16 ? 16 : 8);
      }
    }
  }
  if (rc != MP_OK) {
    delete t;
    return rc;
  }
  ctx->tables.push_back(t);
  *out = t;
  return MP_OK;
  MP_CATCH
}
int mp_table_create_params(mp_ctx* ctx, uint32_t m, uint32_t n, const uint8_t* params, uint32_t fb_window_bits, mp_table** out) {
  // a table of the shared parameters only, for keyed batches: the enc generator G stands in for the (unused) fixed key
  int rc = mp_table_create_ex(ctx, m, n, params, params, fb_window_bits, out);
  if (rc == MP_OK) (*out)->keyless = true;
  return rc;
}
void mp_table_destroy(mp_table* t) {
  if (!t) return;
  try {
    MP_ENTER(t->ctx);
    t->flush();
  } catch (...) {
  }
  mp_ctx* ctx = t->ctx;
  bool last = false;
  {
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    auto& reg = ctx->tables;
    reg.erase(std::remove(reg.begin(), reg.end(), t), reg.end());
    for (auto& st : t->io) {
      if (st.up) rt::event_destroy(st.up);
      if (st.done) rt::event_destroy(st.done);
      if (st.down) rt::event_destroy(st.down);
    }
    delete t;
    last = ctx->dying && reg.empty();
  }
  if (last) ctx_release(ctx);      // mp_ctx_destroy came first (the lock is not held: it goes with the context)
}
void* mp_host_alloc(size_t bytes) {
  try {
    return rt::host_alloc(bytes);
  } catch (const std::exception& e) {
    fail(MP_ERR_INTERNAL, e.what());
    return nullptr;
  }
}
void mp_host_free(void* p) { rt::host_free(p); }

uint32_t mp_table_window_bits(const mp_table* t) { return t ? t->fb_bits : 0; }
int mp_set_latency_batch(mp_table* t, size_t B) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_latency_batch: null table");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->set_latency_batch(B);
  return MP_OK;
}
int mp_set_io_chunk(mp_table* t, size_t proofs) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_io_chunk: null table");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->io_chunk = proofs;
  return MP_OK;
}
int mp_set_merged_verify(mp_table* t, int on) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_merged_verify: null table");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->set_merged_verify(on != 0);
  return MP_OK;
}
int mp_set_group_refine(mp_table* t, uint32_t points_per_subgroup, uint32_t min_subgroups) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_group_refine: null table");
  if (points_per_subgroup > BUCKET_TERMS_MAX - 4096) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_group_refine: at most 585 728 points per equation");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->set_group_refine(points_per_subgroup, min_subgroups);
  return MP_OK;
}
uint64_t mp_reverified_count(const mp_table* t) { return t ? t->reverified() : 0; }
int mp_set_group_adapt(mp_table* t, int on) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_group_adapt: null table");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->set_group_adapt(on != 0);
  return MP_OK;
}
int mp_set_bucket_min(mp_table* t, size_t terms) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_bucket_min: null table");
  MP_TRY
  MP_ENTER(t->ctx);
  t->set_bucket_min((uint32_t)std::min<size_t>(terms, 0x7FFFFFFFu));
  return MP_OK;
  MP_CATCH
}
int mp_set_bucket_bits(mp_table* t, uint32_t bits) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_bucket_bits: null table");
  if (bits != 0 && (bits < 8 || bits > 14)) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_bucket_bits: 0 (by size) or 8 .. 14");
  MP_TRY
  MP_ENTER(t->ctx);
  t->set_bucket_bits(bits);
  return MP_OK;
  MP_CATCH
}
int mp_set_bucket_split(mp_table* t, uint32_t min_bits) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_bucket_split: null table");
  if (min_bits < 10 || min_bits > 15) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_bucket_split: 10 .. 15 (none)");
  MP_TRY
  MP_ENTER(t->ctx);
  t->flush();
  t->bucket_split_bits = min_bits;
  return MP_OK;
  MP_CATCH
}
int mp_set_validated(mp_table* t, uint32_t what) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_validated: null table");
  if (what & ~(MP_VALIDATED_DECKS | MP_VALIDATED_SHUFFLED | MP_VALIDATED_PROOFS)) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_validated: unknown flag");
  MP_TRY
  MP_ENTER(t->ctx);
  t->flush();
  t->validated = what;
  return MP_OK;
  MP_CATCH
}
int mp_deck_validate_dev(mp_table* t, size_t decks, const void* d_wire_decks, void* d_status) {
  if (!t || !decks || !d_wire_decks || !d_status || decks >= ((size_t)1 << 31)) return fail(MP_ERR_BAD_ARGUMENT, "mp_deck_validate_dev: bad argument");
  MP_TRY
  MP_ENTER(t->ctx);
  t->validate_decks_dev(decks, (const uint8_t*)d_wire_decks, (int32_t*)d_status);
  return MP_OK;
  MP_CATCH
}
int mp_set_chain_max_links(mp_table* t, uint32_t links) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_chain_max_links: null table");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->chain_max_links = links;
  return MP_OK;
}
int mp_set_chain_slice(mp_table* t, size_t tables_per_pass) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_chain_slice: null table");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->chain_slice = tables_per_pass;
  return MP_OK;
}
int mp_set_chain_group(mp_table* t, uint32_t tables_per_equation) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_chain_group: null table");
  if (tables_per_equation > 4094) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_chain_group: at most 4 094 tables per equation");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->chain_group = tables_per_equation;
  return MP_OK;
}
int mp_set_transcript_lanes(mp_table* t, uint32_t lanes) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_transcript_lanes: null table");
  if (lanes != 0 && lanes != 1 && lanes != 4) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_transcript_lanes: 0 (by batch size), 1 or 4");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->fs_lanes = lanes;
  return MP_OK;
}
int mp_set_group_lanes(mp_table* t, uint32_t lanes) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_group_lanes: null table");
  if (lanes != 0 && lanes != 1 && lanes != 4) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_group_lanes: 0 (by batch size), 1 or 4");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->group_lanes = lanes;
  return MP_OK;
}
int mp_set_work_split(mp_table* t, int split) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_work_split: null table");
  if (split < -1 || split > 5) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_work_split: -1 (by batch size) or 0 .. 5");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->forced_split = split;
  return MP_OK;
}
int mp_set_group_verify(mp_table* t, uint32_t points_per_group, size_t min_batch) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_group_verify: null table");
  if (points_per_group > BUCKET_TERMS_MAX - 4096) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_group_verify: 0 (off) or up to 585 728 points per group equation");
  MP_TRY
  MP_ENTER(t->ctx);
  t->flush();
  t->set_group_verify(points_per_group, min_batch);
  return MP_OK;
  MP_CATCH
}
uint32_t mp_group_size(const mp_table* t, size_t B) { return t ? t->group_size_of(B) : 0; }
uint32_t mp_chain_group_size(const mp_table* t, size_t tables, uint32_t links, int keyed) {
  if (!t || !tables || !links) return 0;
  // (as mp_verify_shuffle_chain_dev cuts the chain: the sub-chains of a long chain are equations of their own)
  const size_t per_link = (size_t)2 * t->N + 11 * t->m + 8, fixed_part = (size_t)2 * t->N + 1;
  const size_t eq_cap = fixed_part + per_link > 32767 ? (size_t)65535 : 32767;
  if (fixed_part + per_link > eq_cap) return 0;
  uint32_t lmax = std::min<uint32_t>((uint32_t)((eq_cap - fixed_part) / per_link), 1022u);
  if (t->chain_max_links) lmax = std::max(1u, std::min(lmax, t->chain_max_links));
  return t->chain_group_size(tables, std::min(lmax, links), keyed != 0);
}
size_t mp_chain_last_slice(const mp_table* t) { return t ? t->chain_last_slice : 0; }
int mp_set_pipeline(mp_table* t, int depth) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_pipeline: null table");
  MP_TRY
  MP_ENTER(t->ctx);
  if (depth < 0 || depth > 8) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_pipeline: depth 0 (off) .. 8");
  t->flush();
  t->pipeline = depth;
  return MP_OK;
  MP_CATCH
}
int mp_set_plan_params(mp_table* t, int split, uint32_t fixed_terms, uint32_t var_terms, uint32_t table_group, uint32_t norm_chunk,
                       uint32_t window_lanes) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_plan_params: null table");
  MP_TRY
  MP_ENTER(t->ctx);
  if (t->set_plan_params(split, fixed_terms, var_terms, table_group, norm_chunk, window_lanes) != MP_OK)
    return fail(MP_ERR_BAD_ARGUMENT, "mp_set_plan_params: split 0 .. 5, sizes >= 1, table_group <= 64, window_lanes <= 16");
  return MP_OK;
  MP_CATCH
}
int mp_set_plan_thresholds(mp_table* t, size_t finest, size_t small, size_t latency, size_t medium, size_t wide) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_plan_thresholds: null table");
  if (finest > small || small > latency || latency > medium || medium > wide)
    return fail(MP_ERR_BAD_ARGUMENT, "mp_set_plan_thresholds: finest <= small <= latency <= medium <= wide");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->set_plan_thresholds(finest, small, latency, medium, wide);
  return MP_OK;
}
int mp_set_toom_cook(mp_table* t, int on) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_toom_cook: null table");
  MP_TRY
  MP_ENTER(t->ctx);
  t->set_toom_cook(on != 0);
  return MP_OK;
  MP_CATCH
}
int mp_set_subgroup_check(mp_table* t, int on) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_set_subgroup_check: null table");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->set_subgroup_check(on != 0);
  return MP_OK;
}
int mp_reserve(mp_table* t, size_t B) {
  if (!t || !B) return fail(MP_ERR_BAD_ARGUMENT, "mp_reserve: bad argument");
  MP_TRY
  MP_ENTER(t->ctx);
  t->reserve(B);
  rt::stream_sync(t->ctx->stream);
  return MP_OK;
  MP_CATCH
}

int mp_shuffle_and_remask_batch_dev(mp_table* t, size_t B, const void* d_decks, const void* d_masking_factors,
                                    const void* d_permutations, const void* d_prover_seeds, void* d_out_decks,
                                    void* d_out_proofs, void* d_status) {
  if (!t || !B || !d_decks || !d_masking_factors || !d_permutations || !d_prover_seeds || !d_out_decks || !d_out_proofs || !d_status)
    return fail(MP_ERR_BAD_ARGUMENT, "mp_shuffle_and_remask_batch_dev: bad argument");
  if (t->keyless) return fail(MP_ERR_BAD_ARGUMENT, "this table has no aggregate key: use the _keys entry points");
  MP_TRY
  MP_ENTER(t->ctx);
  t->prove_dev(B, (const uint8_t*)d_decks, (const uint8_t*)d_masking_factors, (const uint32_t*)d_permutations,
               (const uint8_t*)d_prover_seeds, (uint8_t*)d_out_decks, (uint8_t*)d_out_proofs, (int32_t*)d_status);
  return MP_OK;
  MP_CATCH
}
int mp_verify_shuffle_batch_dev(mp_table* t, size_t B, const void* d_decks, const void* d_shuffled_decks,
                                const void* d_proofs, void* d_status) {
  if (!t || !B || !d_decks || !d_shuffled_decks || !d_proofs || !d_status)
    return fail(MP_ERR_BAD_ARGUMENT, "mp_verify_shuffle_batch_dev: bad argument");
  if (t->keyless) return fail(MP_ERR_BAD_ARGUMENT, "this table has no aggregate key: use the _keys entry points");
  MP_TRY
  MP_ENTER(t->ctx);
  t->verify_dev(B, (const uint8_t*)d_decks, (const uint8_t*)d_shuffled_decks, (const uint8_t*)d_proofs, (int32_t*)d_status);
  return MP_OK;
  MP_CATCH
}
int mp_shuffle_and_remask_batch_keys_dev(mp_table* t, size_t B, const void* d_keys, const void* d_decks, const void* d_masking_factors,
                                         const void* d_permutations, const void* d_prover_seeds, void* d_out_decks,
                                         void* d_out_proofs, void* d_status) {
  if (!t || !B || !d_keys || !d_decks || !d_masking_factors || !d_permutations || !d_prover_seeds || !d_out_decks || !d_out_proofs || !d_status)
    return fail(MP_ERR_BAD_ARGUMENT, "mp_shuffle_and_remask_batch_keys_dev: bad argument");
  MP_TRY
  MP_ENTER(t->ctx);
  t->prove_dev(B, (const uint8_t*)d_decks, (const uint8_t*)d_masking_factors, (const uint32_t*)d_permutations,
               (const uint8_t*)d_prover_seeds, (uint8_t*)d_out_decks, (uint8_t*)d_out_proofs, (int32_t*)d_status, (const uint8_t*)d_keys);
  return MP_OK;
  MP_CATCH
}
int mp_verify_shuffle_batch_keys_dev(mp_table* t, size_t B, const void* d_keys, const void* d_decks, const void* d_shuffled_decks,
                                     const void* d_proofs, void* d_status) {
  if (!t || !B || !d_keys || !d_decks || !d_shuffled_decks || !d_proofs || !d_status)
    return fail(MP_ERR_BAD_ARGUMENT, "mp_verify_shuffle_batch_keys_dev: bad argument");
  MP_TRY
  MP_ENTER(t->ctx);
  t->verify_dev(B, (const uint8_t*)d_decks, (const uint8_t*)d_shuffled_decks, (const uint8_t*)d_proofs, (int32_t*)d_status,
                (const uint8_t*)d_keys);
  return MP_OK;
  MP_CATCH
}
int mp_keyset_create(mp_table* t, size_t n_keys, const uint8_t* keys, mp_keyset** out) {
  if (!t || !n_keys || !keys || !out) return fail(MP_ERR_BAD_ARGUMENT, "mp_keyset_create: bad argument");
  MP_TRY
  MP_ENTER(t->ctx);
  std::unique_ptr<mp_keyset> ks(new mp_keyset());
  ks->owner = t;
  const int rc = t->keyset_build(*ks, n_keys, keys);
  if (rc != MP_OK) return rc;
  *out = ks.release();
  return MP_OK;
  MP_CATCH
}
void mp_keyset_destroy(mp_keyset* ks) {
  if (!ks) return;
  try {
    MP_ENTER(ks->owner->ctx);
    delete ks;
  } catch (...) {
  }
}
size_t mp_keyset_size(const mp_keyset* ks) { return ks ? ks->K : 0; }
int mp_shuffle_and_remask_batch_keyset_dev(mp_table* t, const mp_keyset* ks, size_t B, const void* d_key_index, const void* d_decks,
                                           const void* d_masking_factors, const void* d_permutations, const void* d_prover_seeds,
                                           void* d_out_decks, void* d_out_proofs, void* d_status) {
  if (!t || !ks || ks->owner != t || !B || !d_key_index || !d_decks || !d_masking_factors || !d_permutations || !d_prover_seeds || !d_out_decks ||
      !d_out_proofs || !d_status)
    return fail(MP_ERR_BAD_ARGUMENT, "mp_shuffle_and_remask_batch_keyset_dev: bad argument (the key set must belong to this table)");
  MP_TRY
  MP_ENTER(t->ctx);
  t->prove_dev(B, (const uint8_t*)d_decks, (const uint8_t*)d_masking_factors, (const uint32_t*)d_permutations,
               (const uint8_t*)d_prover_seeds, (uint8_t*)d_out_decks, (uint8_t*)d_out_proofs, (int32_t*)d_status, nullptr, ks,
               (const uint32_t*)d_key_index);
  return MP_OK;
  MP_CATCH
}
int mp_verify_shuffle_batch_keyset_dev(mp_table* t, const mp_keyset* ks, size_t B, const void* d_key_index, const void* d_decks,
                                       const void* d_shuffled_decks, const void* d_proofs, void* d_status) {
  if (!t || !ks || ks->owner != t || !B || !d_key_index || !d_decks || !d_shuffled_decks || !d_proofs || !d_status)
    return fail(MP_ERR_BAD_ARGUMENT, "mp_verify_shuffle_batch_keyset_dev: bad argument (the key set must belong to this table)");
  MP_TRY
  MP_ENTER(t->ctx);
  t->verify_dev(B, (const uint8_t*)d_decks, (const uint8_t*)d_shuffled_decks, (const uint8_t*)d_proofs, (int32_t*)d_status, nullptr, ks,
                (const uint32_t*)d_key_index);
  return MP_OK;
  MP_CATCH
}
int mp_verify_shuffle_chain_dev(mp_table* t, size_t tables, uint32_t links, const void* d_keys, const void* d_decks, const void* d_proofs,
                                void* d_status) {
  if (!t || !tables || !links || !d_decks || !d_proofs || !d_status || links > 4095)
    return fail(MP_ERR_BAD_ARGUMENT, "mp_verify_shuffle_chain_dev: bad argument");
  if (t->keyless && !d_keys) return fail(MP_ERR_BAD_ARGUMENT, "this table has no aggregate key: pass one key per link");
  if ((uint64_t)tables * links >= ((uint64_t)1 << 31)) return fail(MP_ERR_BAD_ARGUMENT, "mp_verify_shuffle_chain_dev: too many proofs for one call");
  MP_TRY
  MP_ENTER(t->ctx);
  NoPipeline nopipe(t);
  // one chain equation holds at most 32 767 distinct points ((L + 1) 2N decks + L (11m + 8) proof elements + key): longer chains are
  // verified as consecutive sub-chains (the decks array is link-major, so a sub-chain is a contiguous slice)
  const size_t per_link = (size_t)2 * t->N + 11 * t->m + 8, fixed_part = (size_t)2 * t->N + 1;
  const size_t pb = t->point_bytes, deck_bytes = (size_t)2 * t->N * pb, psz = proof_size_bytes(t->m, t->n, (uint32_t)pb);
  // (one link does not always fit 32 767 points -- m = 2048, n = 2: 4N + 11m + 9 = 38 921 -- but it fits one bucket job: the kernel sorts
  // up to BUCKET_TERMS_MAX = 65 535 points per window, and 4N + 11m + 9 <= 38 921 for every table mp_table_create accepts, m n <= 4096)
  const size_t eq_cap = fixed_part + per_link > 32767 ? (size_t)65535 : 32767;
  if (fixed_part + per_link > eq_cap) return fail(MP_ERR_INTERNAL, "mp_verify_shuffle_chain_dev: deck too large for one chain equation");
  uint32_t lmax = std::min<uint32_t>((uint32_t)((eq_cap - fixed_part) / per_link), 1022u);      // (links 0 .. L in 10 bits of a sorted entry: kernels_bucket.hpp)
  if (t->chain_max_links) lmax = std::max(1u, std::min(lmax, t->chain_max_links));
  auto sub_chains = [&](size_t T, const uint8_t* decks, const uint8_t* proofs, int32_t* status, const uint8_t* keys) {
    for (uint32_t j0 = 0; j0 < links; j0 += lmax) {
      const uint32_t lc = std::min(lmax, links - j0);
      t->verify_chain_dev(T, lc, decks + (size_t)j0 * T * deck_bytes, proofs + (size_t)j0 * T * psz, status + (size_t)j0 * T, keys ? keys + (size_t)j0 * T * pb : nullptr);
    }
  };
  // Tables per pass (round 5).  The chain workspace is ~68 KB per link in flight (52 cards): 49 152 tables x 32 links are 107 GB, and the
  // tables a caller holds are bounded by that long before its decks and proofs fill the HBM -- while the PROVER wants many tables per
  // launch (a link of every table is one batch).  So a call whose links do not fit the memory that is free (or the workspace that is
  // there already) is verified in passes of `slice` tables: their rows of the link-major arrays are gathered into the staging buffers
  // of the host-buffer calls (one copy per link and array: 13 KB per proof, ~10 ns at HBM rates), verified as a call of `slice` tables,
  // and their status words scattered back.  mp_set_chain_slice pins the number; verdicts do not depend on it.
  size_t slice = tables;
  const uint32_t lcmax = std::min(lmax, links);
  if (t->chain_slice) {
    slice = std::min(tables, t->chain_slice);
  } else if ((size_t)tables * lcmax > t->chain_lanes_held()) {
    size_t free_b = 0, total_b = 0;
    rt::mem_info(&free_b, &total_b);
    const size_t lane = t->chain_lane_bytes(lcmax, d_keys != nullptr);
    // (what is free now plus what the workspace gives back when it is re-allocated; 40 % of it: the bucket kernel's rows, the equation's
    // scalars and digits, the staging buffers below and the per-link fallback of a failing equation come on top, and smaller passes
    // cost nothing -- 65 536 tables in passes of 16 384 or 32 768: 677 k proofs/s either way, 195 GB against 241 GB in use)
    const size_t room = (size_t)(0.4 * (double)(free_b + t->chain_lanes_held() * lane));
    if (total_b && (size_t)tables * lcmax * lane > room) {
      slice = std::max<size_t>(1024, room / ((size_t)lcmax * lane) / 1024 * 1024);
      // (no sliver at the end: equal passes, whole multiples of 1 024 tables where that is possible)
      const size_t passes = (tables + slice - 1) / slice;
      slice = std::min(tables, ((tables + passes - 1) / passes + 1023) / 1024 * 1024);
    }
  }
  t->chain_last_slice = std::min(slice, tables);
  if (slice >= tables) {
    sub_chains(tables, (const uint8_t*)d_decks, (const uint8_t*)d_proofs, (int32_t*)d_status, (const uint8_t*)d_keys);
    return MP_OK;
  }
  rt::Stream s = t->ctx->stream;
  mp_io_stage& st = t->io[0];
  st.in0.alloc((size_t)(links + 1) * slice * deck_bytes, s, false);
  st.out1.alloc((size_t)links * slice * psz, s, false);
  st.status.alloc((size_t)links * slice, s, false);
  if (d_keys) st.keys.alloc((size_t)links * slice * pb, s, false);
  for (size_t t0 = 0; t0 < tables; t0 += slice) {
    const size_t ts = std::min(slice, tables - t0);
    for (uint32_t j = 0; j <= links; ++j)
      rt::d2d(st.in0.p + (size_t)j * ts * deck_bytes, (const uint8_t*)d_decks + ((size_t)j * tables + t0) * deck_bytes, ts * deck_bytes, s);
    for (uint32_t j = 0; j < links; ++j) {
      rt::d2d(st.out1.p + (size_t)j * ts * psz, (const uint8_t*)d_proofs + ((size_t)j * tables + t0) * psz, ts * psz, s);
      if (d_keys) rt::d2d(st.keys.p + (size_t)j * ts * pb, (const uint8_t*)d_keys + ((size_t)j * tables + t0) * pb, ts * pb, s);
    }
    sub_chains(ts, st.in0.p, st.out1.p, st.status.p, d_keys ? st.keys.p : nullptr);
    for (uint32_t j = 0; j < links; ++j) rt::d2d((int32_t*)d_status + (size_t)j * tables + t0, st.status.p + (size_t)j * ts, ts * 4, s);
  }
  return MP_OK;
  MP_CATCH
}
int mp_verify_shuffle_chain(mp_table* t, size_t tables, uint32_t links, const uint8_t* shared_keys, const uint8_t* decks, const uint8_t* proofs,
                            int32_t* status) {
  if (!t || !tables || !links || !decks || !proofs || !status) return fail(MP_ERR_BAD_ARGUMENT, "mp_verify_shuffle_chain: bad argument");
  MP_TRY
  MP_ENTER(t->ctx);
  rt::Stream s = t->ctx->stream;
  const size_t B = tables * links, pb = t->point_bytes, deck_bytes = (size_t)2 * t->N * pb, psz = proof_size_bytes(t->m, t->n, (uint32_t)pb);
  DevBuf<uint8_t> dd, dp, dk;
  DevBuf<int32_t> ds;
  dd.alloc((B + tables) * deck_bytes, s, false);
  dp.alloc(B * psz, s, false);
  ds.alloc(B, s, false);
  rt::h2d(dd.p, decks, (B + tables) * deck_bytes, s);
  rt::h2d(dp.p, proofs, B * psz, s);
  if (shared_keys) {
    dk.alloc(B * pb, s, false);
    rt::h2d(dk.p, shared_keys, B * pb, s);
  }
  const int rc = mp_verify_shuffle_chain_dev(t, tables, links, shared_keys ? dk.p : nullptr, dd.p, dp.p, ds.p);
  if (rc != MP_OK) return rc;
  rt::d2h(status, ds.p, B * 4, s);
  rt::stream_sync(s);
  return MP_OK;
  MP_CATCH
}
int mp_sync(mp_ctx* ctx) {
  if (!ctx) return fail(MP_ERR_BAD_ARGUMENT, "mp_sync: null");
  MP_TRY
  MP_ENTER(ctx);
  rt::stream_sync(ctx->stream);
  for (mp_table* t : ctx->tables) t->flush();      // pipelined verify calls: deferred per-equation passes, then the verify lane
  return MP_OK;
  MP_CATCH
}

// ---- host-buffer entry points: the batch is cut into chunks and pipelined -- while the kernels of chunk k run on the
// context's stream, chunk k+1 is uploaded on a second stream and the results of chunk k-1 are downloaded on a third
// (PCIe is full duplex).  Staging buffers are persistent (two chunks in flight).  With page-locked caller buffers
// (mp_host_alloc) every copy is an asynchronous DMA; with pageable buffers the runtime stages them and the overlap is partial.
static const size_t IO_CHUNK = 65536;   // default proofs per chunk: the kernels need ~64 k lanes to run at full rate ...
static const size_t IO_CHUNK_LARGE = 131072;   // ... and calls of 262 144 proofs or more take chunks of twice that (round 5: the copies are
// hidden either way; what the chunked call loses against one batch is kernel efficiency -- 474 ms of kernels against 385 with 65 536-proof
// chunks, profiles/r04_pcie_inclusive.txt -- the screen of a 131 072-proof chunk runs on groups of 128 proofs, that of 65 536 on groups of 64)
static size_t io_chunk_of(const mp_table* t, size_t B) { return std::min(B, t->io_chunk ? t->io_chunk : (B >= 2 * IO_CHUNK_LARGE ? IO_CHUNK_LARGE : IO_CHUNK)); }
// The chunks of one call: full chunks in the middle, a ramp of C/8 and 3C/8 at either end -- the first upload and the last download
// are the only transfers nothing overlaps, so the first and the last chunk are small (262 144 proofs: 22 + 34 ms of exposed copies per
// prove call with four equal chunks, 3 + 4 ms with the ramp, for ~15 ms of less efficient small-batch kernels).  Calls of less
// than three chunks are cut evenly as before.  tail_ramp = false (verification: all that comes back is a status word per proof): see below
static std::vector<size_t> io_schedule(size_t B, size_t C, bool tail_ramp = true) {
  std::vector<size_t> v;
  if (B < 2 * C || C < 8) {
    for (size_t o = 0; o < B; o += C) v.push_back(std::min(C, B - o));
    return v;
  }
#ifndef MP_EXP_IO_RAMP      // (A/B hook of round 5, compile-time: profiles/r05g_ab_verify_upload_ramp.txt; no environment variable steers the library)
#define MP_EXP_IO_RAMP 1
#endif
  constexpr bool slow_ramp = MP_EXP_IO_RAMP != 0;
  if (!tail_ramp && !slow_ramp) {      // round 4's schedule without its tail: C/8, 3C/8, then full chunks
    v.push_back(C / 8);
    v.push_back(3 * C / 8);
    for (size_t left = B - C / 2; left;) {
      const size_t c = std::min(C, left);
      v.push_back(c);
      left -= c;
    }
    return v;
  }
  if (!tail_ramp) {
    // Verification uploads 19.7 KB per proof (both decks and the proof) for ~0.5 us of kernels: at PCIe rates the copy of chunk k + 1
    // takes about as long as the kernels of chunk k, so the chunks can only grow slowly -- by half from one to the next, starting at C / 8;
    // whatever is left when the next step would overshoot is the last chunk (nothing of a verify call is exposed at its end)
    size_t left = B;
    for (size_t c = C / 8; left; c = std::min(C, (c + c / 2 + 1023) / 1024 * 1024)) {
      // (no sliver at the end: what is left below 1.5 c goes as one chunk, or as two halves if that would exceed C)
      const size_t take = left >= c + c / 2 ? c : (left <= C ? left : (left + 1) / 2);
      v.push_back(take);
      left -= take;
    }
    return v;
  }
  const size_t r0 = C / 8, r1 = 3 * C / 8;
  v.push_back(r0);
  v.push_back(r1);
  for (size_t left = B - 2 * (r0 + r1); left;) {
    const size_t c = std::min(C, left);
    v.push_back(c);
    left -= c;
  }
  v.push_back(r1);
  v.push_back(r0);
  return v;
}
static void io_events(mp_io_stage& st) {
  if (!st.up) {
    st.up = rt::event_create();
    st.done = rt::event_create();
    st.down = rt::event_create();
  }
}

static int prove_batch_host(mp_table* t, size_t B, const uint8_t* keys, const uint8_t* decks, const uint8_t* masking_factors,
                            const uint32_t* permutations, const uint8_t* prover_seeds, uint8_t* out_decks,
                            uint8_t* out_proofs, int32_t* status) {
  if (!t || !B || !decks || !masking_factors || !permutations || !prover_seeds || !out_decks || !out_proofs || !status)
    return fail(MP_ERR_BAD_ARGUMENT, "mp_shuffle_and_remask_batch: bad argument");
  if (t->keyless && !keys) return fail(MP_ERR_BAD_ARGUMENT, "this table has no aggregate key: use the _keys entry points");
  MP_TRY
  MP_ENTER(t->ctx);
  rt::Stream s = t->ctx->stream, up = t->ctx->h2d, down = t->ctx->d2h;
  const size_t N = t->N, psz = proof_size_bytes(t->m, t->n, t->point_bytes), dsz = N * 2 * t->point_bytes;
  const size_t chunk = io_chunk_of(t, B);
  const std::vector<size_t> sched = io_schedule(B, chunk);
  std::vector<size_t> first(sched.size() + 1, 0);
  for (size_t k = 0; k < sched.size(); ++k) first[k + 1] = first[k] + sched[k];
  const size_t nchunks = sched.size();
  for (auto& st : t->io) {
    io_events(st);
    st.used = false;
    st.in0.alloc(chunk * dsz, s, false); st.in1.alloc(chunk * N * 32, s, false); st.perm.alloc(chunk * N, s, false);
    st.in2.alloc(chunk * 32, s, false); st.out0.alloc(chunk * dsz, s, false); st.out1.alloc(chunk * psz, s, false);
    st.status.alloc(chunk, s, false);
    if (keys) st.keys.alloc(chunk * t->point_bytes, s, false);
  }
  auto upload = [&](size_t k) {
    mp_io_stage& st = t->io[k & 1];
    const size_t o = first[k], c = sched[k];
    if (st.used) rt::stream_wait(up, st.done);          // the kernels of chunk k-2 have consumed these buffers
    rt::h2d(st.in0.p, decks + o * dsz, c * dsz, up);
    rt::h2d(st.in1.p, masking_factors + o * N * 32, c * N * 32, up);
    rt::h2d(st.perm.p, permutations + o * N, c * N * 4, up);
    rt::h2d(st.in2.p, prover_seeds + o * 32, c * 32, up);
    if (keys) rt::h2d(st.keys.p, keys + o * t->point_bytes, c * t->point_bytes, up);
    rt::event_record(st.up, up);
  };
  upload(0);
  for (size_t k = 0; k < nchunks; ++k) {
    mp_io_stage& st = t->io[k & 1];
    const size_t o = first[k], c = sched[k];
    rt::stream_wait(s, st.up);
    if (st.used) rt::stream_wait(s, st.down);            // results of chunk k-2 have left the output buffers
    t->prove_dev(c, st.in0.p, st.in1.p, st.perm.p, st.in2.p, st.out0.p, st.out1.p, st.status.p, keys ? st.keys.p : nullptr);
    rt::event_record(st.done, s);
    if (k + 1 < nchunks) upload(k + 1);
    rt::stream_wait(down, st.done);
    rt::d2h(out_decks + o * dsz, st.out0.p, c * dsz, down);
    rt::d2h(out_proofs + o * psz, st.out1.p, c * psz, down);
    rt::d2h(status + o, st.status.p, c * 4, down);
    rt::event_record(st.down, down);
    st.used = true;
  }
  rt::stream_sync(down);
  rt::stream_sync(s);
  return MP_OK;
  MP_CATCH
}
static int verify_batch_host(mp_table* t, size_t B, const uint8_t* keys, const uint8_t* decks, const uint8_t* shuffled_decks,
                             const uint8_t* proofs, int32_t* status) {
  if (!t || !B || !decks || !shuffled_decks || !proofs || !status) return fail(MP_ERR_BAD_ARGUMENT, "mp_verify_shuffle_batch: bad argument");
  if (t->keyless && !keys) return fail(MP_ERR_BAD_ARGUMENT, "this table has no aggregate key: use the _keys entry points");
  MP_TRY
  MP_ENTER(t->ctx);
  NoPipeline nopipe(t);
  rt::Stream s = t->ctx->stream, up = t->ctx->h2d, down = t->ctx->d2h;
  const size_t N = t->N, psz = proof_size_bytes(t->m, t->n, t->point_bytes), dsz = N * 2 * t->point_bytes;
  const size_t chunk = io_chunk_of(t, B);
  const std::vector<size_t> sched = io_schedule(B, chunk, false);
  std::vector<size_t> first(sched.size() + 1, 0);
  for (size_t k = 0; k < sched.size(); ++k) first[k + 1] = first[k] + sched[k];
  const size_t nchunks = sched.size();
  for (auto& st : t->io) {
    io_events(st);
    st.used = false;
    st.in0.alloc(chunk * dsz, s, false); st.in3.alloc(chunk * dsz, s, false); st.out1.alloc(chunk * psz, s, false);
    st.status.alloc(chunk, s, false);
    if (keys) st.keys.alloc(chunk * t->point_bytes, s, false);
  }
  auto upload = [&](size_t k) {
    mp_io_stage& st = t->io[k & 1];
    const size_t o = first[k], c = sched[k];
    if (st.used) rt::stream_wait(up, st.done);
    rt::h2d(st.in0.p, decks + o * dsz, c * dsz, up);
    rt::h2d(st.in3.p, shuffled_decks + o * dsz, c * dsz, up);
    rt::h2d(st.out1.p, proofs + o * psz, c * psz, up);
    if (keys) rt::h2d(st.keys.p, keys + o * t->point_bytes, c * t->point_bytes, up);
    rt::event_record(st.up, up);
  };
  upload(0);
  for (size_t k = 0; k < nchunks; ++k) {
    mp_io_stage& st = t->io[k & 1];
    const size_t o = first[k], c = sched[k];
    if (k + 1 < nchunks) upload(k + 1);                  // before verify_dev: it ends with a read-back of the screening flag
    rt::stream_wait(s, st.up);
    if (st.used) rt::stream_wait(s, st.down);
    t->verify_dev(c, st.in0.p, st.in3.p, st.out1.p, st.status.p, keys ? st.keys.p : nullptr);
    rt::event_record(st.done, s);
    rt::stream_wait(down, st.done);
    rt::d2h(status + o, st.status.p, c * 4, down);
    rt::event_record(st.down, down);
    st.used = true;
  }
  rt::stream_sync(down);
  rt::stream_sync(s);
  return MP_OK;
  MP_CATCH
}

int mp_shuffle_and_remask_batch(mp_table* t, size_t B, const uint8_t* decks, const uint8_t* masking_factors,
                                const uint32_t* permutations, const uint8_t* prover_seeds, uint8_t* out_decks,
                                uint8_t* out_proofs, int32_t* status) {
  return prove_batch_host(t, B, nullptr, decks, masking_factors, permutations, prover_seeds, out_decks, out_proofs, status);
}
int mp_verify_shuffle_batch(mp_table* t, size_t B, const uint8_t* decks, const uint8_t* shuffled_decks,
                            const uint8_t* proofs, int32_t* status) {
  return verify_batch_host(t, B, nullptr, decks, shuffled_decks, proofs, status);
}
int mp_shuffle_and_remask_batch_keys(mp_table* t, size_t B, const uint8_t* shared_keys, const uint8_t* decks,
                                     const uint8_t* masking_factors, const uint32_t* permutations, const uint8_t* prover_seeds,
                                     uint8_t* out_decks, uint8_t* out_proofs, int32_t* status) {
  if (!shared_keys) return fail(MP_ERR_BAD_ARGUMENT, "mp_shuffle_and_remask_batch_keys: null keys");
  return prove_batch_host(t, B, shared_keys, decks, masking_factors, permutations, prover_seeds, out_decks, out_proofs, status);
}
int mp_verify_shuffle_batch_keys(mp_table* t, size_t B, const uint8_t* shared_keys, const uint8_t* decks,
                                 const uint8_t* shuffled_decks, const uint8_t* proofs, int32_t* status) {
  if (!shared_keys) return fail(MP_ERR_BAD_ARGUMENT, "mp_verify_shuffle_batch_keys: null keys");
  return verify_batch_host(t, B, shared_keys, decks, shuffled_decks, proofs, status);
}
int mp_shuffle_and_remask(mp_table* t, const uint8_t* deck, const uint8_t* masking_factors, const uint32_t* permutation,
                          const uint8_t prover_seed[32], uint8_t* out_deck, uint8_t* out_proof) {
  int32_t st = 0;
  int rc = mp_shuffle_and_remask_batch(t, 1, deck, masking_factors, permutation, prover_seed, out_deck, out_proof, &st);
  if (rc != MP_OK) return rc;
  if (st < 0) return fail(st, mp_check_name(st));
  return st;
}
int mp_verify_shuffle(mp_table* t, const uint8_t* deck, const uint8_t* shuffled_deck, const uint8_t* proof, size_t proof_len) {
  if (!t) return fail(MP_ERR_BAD_ARGUMENT, "mp_verify_shuffle: null table");
  if (proof_len != proof_size_bytes(t->m, t->n, t->point_bytes)) return fail(MP_ERR_BAD_ARGUMENT, "mp_verify_shuffle: wrong proof length");
  int32_t st = 0;
  int rc = mp_verify_shuffle_batch(t, 1, deck, shuffled_deck, proof, &st);
  if (rc != MP_OK) return rc;
  if (st < 0) return fail(st, mp_check_name(st));
  return st;
}

int mp_remask_batch(mp_table* t, size_t count, const uint8_t* cards, const uint8_t* masking_factors, uint8_t* out) {
  if (!t || !count || !cards || !masking_factors || !out) return fail(MP_ERR_BAD_ARGUMENT, "mp_remask_batch: bad argument");
  MP_TRY
  MP_ENTER(t->ctx);
  t->remask_host(count, cards, masking_factors, out);
  return MP_OK;
  MP_CATCH
}
int mp_msm(mp_table* t, size_t n_msm, size_t k, const uint8_t* scalars, const uint8_t* points, uint8_t* out) {
  if (!t || !n_msm || !k || !scalars || !points || !out) return fail(MP_ERR_BAD_ARGUMENT, "mp_msm: bad argument");
  MP_TRY
  MP_ENTER(t->ctx);
  t->msm_host(n_msm, k, scalars, points, out);
  return MP_OK;
  MP_CATCH
}
int mp_commit_batch(mp_table* t, size_t count, size_t len, const uint8_t* values, const uint8_t* r, uint8_t* out) {
  if (!t || !count || !r || !out || (len && !values) || len > t->n) return fail(MP_ERR_BAD_ARGUMENT, "mp_commit_batch: bad argument");
  MP_TRY
  MP_ENTER(t->ctx);
  t->commit_host(count, len, values, r, out);
  return MP_OK;
  MP_CATCH
}

int mp_profile_enable(mp_ctx* ctx, int on) {
  if (!ctx) return fail(MP_ERR_BAD_ARGUMENT, "null ctx");
  MP_TRY
  std::lock_guard<std::recursive_mutex> mp_lock_(ctx->mu);
  rt::stream_sync(ctx->stream);
  ctx->prof.report();
  ctx->prof.on = on != 0;
  return MP_OK;
  MP_CATCH
}
int mp_profile_report(mp_ctx* ctx, char* buf, size_t buf_len) {
  if (!ctx || !buf || !buf_len) return fail(MP_ERR_BAD_ARGUMENT, "mp_profile_report: bad argument");
  MP_TRY
  std::lock_guard<std::recursive_mutex> mp_lock_(ctx->mu);
  rt::stream_sync(ctx->stream);
  std::string r = ctx->prof.report();
  if (r.size() + 1 > buf_len) r.resize(buf_len - 1);
  memcpy(buf, r.c_str(), r.size() + 1);
  return MP_OK;
  MP_CATCH
}
int mp_work_census(mp_table* t, uint64_t* prove_terms, uint64_t* verify_terms, uint64_t* prove_point_ops, uint64_t* verify_point_ops) {
  if (!t || !prove_terms || !verify_terms || !prove_point_ops || !verify_point_ops) return fail(MP_ERR_BAD_ARGUMENT, "mp_work_census: bad argument");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->census(prove_terms, verify_terms, prove_point_ops, verify_point_ops);
  return MP_OK;
}

int mp_blake2s(const uint8_t* in, size_t len, uint8_t out[32]) {
  if ((!in && len) || !out) return fail(MP_ERR_BAD_ARGUMENT, "mp_blake2s: bad argument");
  Blake2sState st;
  blake2s_init(st);
  size_t off = 0;
  uint32_t m[16];
  while (len - off > 64) {
    memcpy(m, in + off, 64);
    off += 64;
    blake2s_compress(st, m, off, false);
  }
  memset(m, 0, 64);
  if (len - off) memcpy(m, in + off, len - off);
  blake2s_compress(st, m, len, true);
  memcpy(out, st.h, 32);
  return MP_OK;
}
int mp_sigma_prove_batch(mp_table* t, size_t B, uint32_t nbases, const uint8_t* bases, const uint8_t* publics,
                         const uint8_t* witness, const uint8_t* fs_init, const uint8_t* prover_seeds, uint8_t* out_proofs,
                         int32_t* status) {
  if (!t || !B || (nbases != 1 && nbases != 2) || !bases || !publics || !witness || !fs_init || !prover_seeds || !out_proofs || !status)
    return fail(MP_ERR_BAD_ARGUMENT, "mp_sigma_prove_batch: bad argument");
  MP_TRY
  MP_ENTER(t->ctx);
  t->sigma_host(true, B, nbases, bases, publics, witness, fs_init, prover_seeds, out_proofs, status);
  return MP_OK;
  MP_CATCH
}
int mp_sigma_verify_batch(mp_table* t, size_t B, uint32_t nbases, const uint8_t* bases, const uint8_t* publics,
                          const uint8_t* proofs, const uint8_t* fs_init, int32_t* status) {
  if (!t || !B || (nbases != 1 && nbases != 2) || !bases || !publics || !proofs || !fs_init || !status)
    return fail(MP_ERR_BAD_ARGUMENT, "mp_sigma_verify_batch: bad argument");
  MP_TRY
  MP_ENTER(t->ctx);
  t->sigma_host(false, B, nbases, bases, publics, nullptr, fs_init, nullptr, const_cast<uint8_t*>(proofs), status);
  return MP_OK;
  MP_CATCH
}

int mp_plan_stats(mp_table* t, uint64_t out[16]) {
  if (!t || !out) return fail(MP_ERR_BAD_ARGUMENT, "mp_plan_stats: bad argument");
  std::lock_guard<std::recursive_mutex> mp_lock_(t->ctx->mu);
  t->plan_stats(out);
  return MP_OK;
}


// ---- canonical serialisation entry points (helpers: above the extern "C" block)

size_t mp_serialized_point_size(int curve_id) { return curve_ok(curve_id) ? (size_t)(kCurveBits[curve_id] + 2 + 7) / 8 : 0; }
size_t mp_serialized_deck_size(int curve_id, size_t cards) { return 8 + 2 * cards * mp_serialized_point_size(curve_id); }
size_t mp_serialized_params_size(int curve_id, uint32_t n) { return 24 + (size_t)(n + 3) * mp_serialized_point_size(curve_id); }
size_t mp_serialized_proof_size(int curve_id, uint32_t m, uint32_t n) {
  size_t sz = 0;
  for (const Group& g : proof_schema(m, n)) sz += (g.vec ? 8 : 0) + (size_t)g.count * (g.point ? mp_serialized_point_size(curve_id) : 32);
  return sz;
}
int mp_points_serialize(int curve_id, size_t count, const uint8_t* wire_points, uint8_t* out) {
  if (!curve_ok(curve_id) || (count && (!wire_points || !out))) return fail(MP_ERR_BAD_ARGUMENT, "mp_points_serialize: bad argument");
  MP_TRY
  if (ser_points(curve_id, false, count, wire_points, out) >= 0) return fail(MP_ERR_BAD_ENCODING, "point: coordinate out of range");
  return MP_OK;
  MP_CATCH
}
int mp_points_deserialize(int curve_id, size_t count, const uint8_t* data, uint8_t* out_wire_points) {
  if (!curve_ok(curve_id) || (count && (!data || !out_wire_points))) return fail(MP_ERR_BAD_ARGUMENT, "mp_points_deserialize: bad argument");
  MP_TRY
  if (ser_points(curve_id, true, count, data, out_wire_points) >= 0)
    return fail(MP_ERR_BAD_ENCODING, "point: non-canonical encoding, x not on the curve, or not in the prime-order subgroup");
  return MP_OK;
  MP_CATCH
}
int mp_deck_serialize(int curve_id, size_t cards, const uint8_t* wire_deck, uint8_t* out) {
  if (!curve_ok(curve_id) || !out || (cards && !wire_deck)) return fail(MP_ERR_BAD_ARGUMENT, "mp_deck_serialize: bad argument");
  put_u64(out, cards);
  return mp_points_serialize(curve_id, 2 * cards, wire_deck, out + 8);
}
int mp_deck_deserialize(int curve_id, const uint8_t* data, size_t len, size_t max_cards, uint8_t* out_wire_deck, size_t* out_cards) {
  if (!curve_ok(curve_id) || !data || !out_cards) return fail(MP_ERR_BAD_ARGUMENT, "mp_deck_deserialize: bad argument");
  if (len < 8) return fail(MP_ERR_BAD_ENCODING, "deck: not enough data");
  const uint64_t k = get_u64(data);
  if (k > max_cards || k > ((uint64_t)1 << 40)) return fail(MP_ERR_BAD_ARGUMENT, "deck: more cards than the output buffer holds");
  if (len != mp_serialized_deck_size(curve_id, (size_t)k)) return fail(MP_ERR_BAD_ENCODING, "deck: length does not match the card count");
  *out_cards = (size_t)k;
  return mp_points_deserialize(curve_id, 2 * (size_t)k, data + 8, out_wire_deck);
}
// Parameters { m, n, enc_parameters { generator }, commit_parameters { g: Vec, h }, generator } [REF mod.rs:37-43]; raw = G | ck | H | gen
int mp_params_serialize(int curve_id, uint32_t m, uint32_t n, const uint8_t* raw_params, uint8_t* out) {
  if (!curve_ok(curve_id) || !raw_params || !out) return fail(MP_ERR_BAD_ARGUMENT, "mp_params_serialize: bad argument");
  const size_t pb = mp_point_size(curve_id), cb = mp_serialized_point_size(curve_id);
  put_u64(out, m);
  put_u64(out + 8, n);
  int rc = mp_points_serialize(curve_id, 1, raw_params, out + 16);
  if (rc) return rc;
  put_u64(out + 16 + cb, n);
  rc = mp_points_serialize(curve_id, n, raw_params + pb, out + 24 + cb);
  if (rc) return rc;
  return mp_points_serialize(curve_id, 2, raw_params + pb * (n + 1), out + 24 + cb * (n + 1));
}
int mp_params_deserialize(int curve_id, const uint8_t* data, size_t len, size_t max_n, uint32_t* m, uint32_t* n, uint8_t* out_raw_params) {
  if (!curve_ok(curve_id) || !data || !m || !n || !out_raw_params) return fail(MP_ERR_BAD_ARGUMENT, "mp_params_deserialize: bad argument");
  if (len < 16) return fail(MP_ERR_BAD_ENCODING, "parameters: not enough data");
  const uint64_t mm = get_u64(data), nn = get_u64(data + 8);
  if (nn > max_n || nn > 0x00FFFFFFull || mm > 0xFFFFFFFFull) return fail(MP_ERR_BAD_ARGUMENT, "parameters: n exceeds the output buffer");
  if (len != mp_serialized_params_size(curve_id, (uint32_t)nn)) return fail(MP_ERR_BAD_ENCODING, "parameters: length does not match n");
  const size_t pb = mp_point_size(curve_id), cb = mp_serialized_point_size(curve_id);
  if (get_u64(data + 16 + cb) != nn) return fail(MP_ERR_BAD_ENCODING, "parameters: commit key length != n");
  int rc = mp_points_deserialize(curve_id, 1, data + 16, out_raw_params);
  if (rc) return rc;
  rc = mp_points_deserialize(curve_id, (size_t)nn, data + 24 + cb, out_raw_params + pb);
  if (rc) return rc;
  rc = mp_points_deserialize(curve_id, 2, data + 24 + cb * (nn + 1), out_raw_params + pb * (nn + 1));
  if (rc) return rc;
  *m = (uint32_t)mm;
  *n = (uint32_t)nn;
  return MP_OK;
}
// ---- on-device decompression: compressed arkworks points in HBM -> wire v1 in HBM (kernels_decompress.hpp)
static int decompress_dev(mp_ctx* ctx, size_t groups, uint32_t per_group, uint32_t prefix, const void* d_in, void* d_out, void* d_status) {
  MP_TRY
  MP_ENTER(ctx);
  switch (ctx->curve) {
    case 0: return decompress_dev_Stark(ctx, groups, per_group, prefix, (const uint8_t*)d_in, (uint8_t*)d_out, (int32_t*)d_status);
    case 1: return decompress_dev_Bn254(ctx, groups, per_group, prefix, (const uint8_t*)d_in, (uint8_t*)d_out, (int32_t*)d_status);
    case 3: return decompress_dev_Bls12_377(ctx, groups, per_group, prefix, (const uint8_t*)d_in, (uint8_t*)d_out, (int32_t*)d_status);
    default: return decompress_dev_Secp256k1(ctx, groups, per_group, prefix, (const uint8_t*)d_in, (uint8_t*)d_out, (int32_t*)d_status);
  }
  MP_CATCH
}
int mp_points_deserialize_dev(mp_ctx* ctx, size_t count, const void* d_data, void* d_out_wire_points, void* d_status) {
  if (!ctx || !count || !d_data || !d_out_wire_points || !d_status) return fail(MP_ERR_BAD_ARGUMENT, "mp_points_deserialize_dev: bad argument");
  return decompress_dev(ctx, count, 1, 0, d_data, d_out_wire_points, d_status);
}
int mp_deck_deserialize_dev(mp_ctx* ctx, size_t decks, size_t cards, const void* d_data, void* d_out_wire_decks, void* d_status) {
  if (!ctx || !decks || !cards || cards > 4096 || !d_data || !d_out_wire_decks || !d_status)
    return fail(MP_ERR_BAD_ARGUMENT, "mp_deck_deserialize_dev: bad argument");
  return decompress_dev(ctx, decks, (uint32_t)(2 * cards), 8, d_data, d_out_wire_decks, d_status);
}
int mp_proof_serialize(int curve_id, uint32_t m, uint32_t n, const uint8_t* proof_wire, uint8_t* out) {
  if (!curve_ok(curve_id) || !proof_wire || !out || m < 2 || n < 2) return fail(MP_ERR_BAD_ARGUMENT, "mp_proof_serialize: bad argument");
  const size_t pb = mp_point_size(curve_id), cb = mp_serialized_point_size(curve_id);
  for (const Group& g : proof_schema(m, n)) {
    if (g.vec) {
      put_u64(out, g.vec_len);
      out += 8;
    }
    if (g.point) {
      const int rc = mp_points_serialize(curve_id, g.count, proof_wire, out);
      if (rc) return rc;
      proof_wire += pb * g.count;
      out += cb * g.count;
    } else {
      memcpy(out, proof_wire, 32 * (size_t)g.count);
      proof_wire += 32 * (size_t)g.count;
      out += 32 * (size_t)g.count;
    }
  }
  return MP_OK;
}
int mp_proof_deserialize(int curve_id, uint32_t m, uint32_t n, const uint8_t* data, size_t len, uint8_t* out_proof_wire) {
  if (!curve_ok(curve_id) || !data || !out_proof_wire || m < 2 || n < 2) return fail(MP_ERR_BAD_ARGUMENT, "mp_proof_deserialize: bad argument");
  if (len != mp_serialized_proof_size(curve_id, m, n)) return fail(MP_ERR_BAD_ENCODING, "shuffle proof: wrong length");
  const size_t pb = mp_point_size(curve_id), cb = mp_serialized_point_size(curve_id);
  for (const Group& g : proof_schema(m, n)) {
    if (g.vec) {
      if (get_u64(data) != g.vec_len) return fail(MP_ERR_BAD_ENCODING, "shuffle proof: a vector has the wrong length");
      data += 8;
    }
    if (g.point) {
      const int rc = mp_points_deserialize(curve_id, g.count, data, out_proof_wire);
      if (rc) return rc;
      data += cb * g.count;
      out_proof_wire += pb * g.count;
    } else {
      if (!ser_scalars_ok(curve_id, g.count, data)) return fail(MP_ERR_BAD_ENCODING, "shuffle proof: scalar out of range");
      memcpy(out_proof_wire, data, 32 * (size_t)g.count);
      data += 32 * (size_t)g.count;
      out_proof_wire += 32 * (size_t)g.count;
    }
  }
  return MP_OK;
}

}
