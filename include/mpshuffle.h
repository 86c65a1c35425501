/*
 * mpshuffle.h -- C ABI of libmpshuffle.so: the MI355X (gfx950) shuffle-proof engine that replaces, for the
 * hot path only, the `BarnettSmartProtocol` implementation of geometryxyz/mental-poker:
 *
 *     DLCards::setup               [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:105-121]
 *     DLCards::shuffle_and_remask  [REF .../discrete_log_cards/mod.rs:380-418, trait decl src/lib.rs:181-188]
 *     DLCards::verify_shuffle      [REF .../discrete_log_cards/mod.rs:420-443, trait decl src/lib.rs:191-197]
 *
 * The reference has no FFI seam -- its seam is the Rust trait (a second `impl BarnettSmartProtocol`); these are
 * the entry points such an impl binds through `extern "C"` (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain pointers and sizes; the CALLER allocates every buffer (sizes: mp_proof_size, mp_params_size);
 *   - all multi-byte buffers must be 4-byte aligned;
 *   - return value / status word: 0 = Ok; > 0 = proof rejected, value = code of the first failing check
 *     (mp_check_name: 1 "Hadamard Product (5.1)" [REF tests.rs:223-225], 2 "Zero Argument (5.2)",
 *     3 "Single Value Product (5.3)", 4 "Multi-Exponentiation Argument (4)") <-> CryptoError::ProofVerificationError;
 *     < 0 = usage / encoding error <-> CardProtocolError::IoError [REF src/error.rs:6-12]
 *     (MP_ERR_*; text via mp_last_error);
 *   - wire encodings ("mpshuffle wire v1"):
 *       scalar (Fr)      32 B little-endian canonical integer, must be < group order
 *       point            mp_point_size(curve) B: x LE || y LE (canonical, affine), 8 bytes per ark-ff limb of the base
 *                        field: 64 B on the 256-bit curves, 96 B on BLS12-377; point at infinity = all-zero bytes
 *       ciphertext/card  two points: c0 || c1                             (el_gamal::Ciphertext(pub Affine, pub Affine))
 *       deck             N ciphertexts back to back, N = m*n
 *       parameters       (n+3) points: G | ck_0 .. ck_{n-1} | H | gen     (enc generator, Pedersen key, extra generator)
 *       permutation      N uint32, out[i] = in[perm[i]]                   (Permutation::permute_array)
 *       proof            mp_proof_size(m, n) bytes, element order in DESIGN.md ("proof wire order")
 *   - prover randomness is an explicit 32-byte seed: the prover draws `Fr::rand` from
 *     ChaCha20Rng::from_seed(seed) in the documented order (the reference takes `rng: &mut R`, mod.rs:381).
 *   - thread-safety (round 6; the reference's trait members are associated functions without `self` or global state [REF src/lib.rs:74-197],
 *     so any number of host threads may prove and verify at once): every entry point that takes a context, a table or a key set holds
 *     the CONTEXT's lock for the length of the call.  Calls from several host threads on one context (or on tables of one context) are
 *     therefore safe and run one after the other, in the order the threads get the lock -- a prove or verify call of 2 048 proofs is a few
 *     milliseconds; calls on DIFFERENT contexts share nothing (streams, arenas, tables and profiler are per context) and run side by
 *     side on the device.  mp_last_error is per thread.  What stays with the caller: a buffer handed to a call is not written by
 *     another thread until the call has returned (pipelined verification, mp_set_pipeline: until mp_sync or `depth` further verify
 *     calls have), and no call is in flight on a context or table that is being destroyed.  The library reads no environment variable
 *     and keeps no mutable global.  tests/test_gpu_round6.py::test_four_host_threads_* and tests/test_threads_tsan.py hold it to that.
 */
#ifndef MPSHUFFLE_H
#define MPSHUFFLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MP_CURVE_STARK 0      /* starknet_curve::Projective: every reference test + examples/round.rs */
#define MP_CURVE_BN254 1
#define MP_CURVE_SECP256K1 2
#define MP_CURVE_BLS12_377 3  /* ark_bls12_377::G1Projective: examples/parameter_selection.rs [REF :25] (377-bit base field) */

#define MP_OK 0
#define MP_ERR_BAD_ENCODING (-1)     /* non-canonical scalar / coordinate, point not on the curve */
#define MP_ERR_BAD_PERMUTATION (-2)
#define MP_ERR_BAD_ARGUMENT (-3)     /* null pointer, m < 2, n < 2, N > 4096, wrong proof length ... */
#define MP_ERR_NO_DEVICE (-4)        /* no usable MI355X / HIP runtime error: the engine never falls back to a CPU path */
#define MP_ERR_INTERNAL (-5)

typedef struct mp_ctx mp_ctx;       /* one GPU + stream + curve */
typedef struct mp_table mp_table;   /* shared parameters + shared key of one card table, with their device tables */

/* ---- context ---------------------------------------------------------------------------------------- */
int mp_ctx_create(int curve_id, int device, mp_ctx** out);
void mp_ctx_destroy(mp_ctx* ctx);                 /* tables of the context that are still alive keep it alive: it goes with the last of them */
const char* mp_last_error(void);                 /* thread-local text of the last error */
const char* mp_check_name(int code);             /* "Ok", "Hadamard Product (5.1)", ... */
size_t mp_proof_size(uint32_t m, uint32_t n);    /* (11m+8) points + (5n+9) scalars, 64-byte points (the 256-bit curves) */
size_t mp_params_size(uint32_t n);               /* (n+3) * 64 */
size_t mp_point_size(int curve_id);                              /* wire bytes of one point: 64, or 96 on MP_CURVE_BLS12_377 */
size_t mp_proof_size_curve(int curve_id, uint32_t m, uint32_t n); /* same counts with mp_point_size(curve_id) per point */
size_t mp_params_size_curve(int curve_id, uint32_t n);            /* (n+3) * mp_point_size(curve_id) */

/* Page-locked host memory for the host-buffer entry points (mp_*_batch): with buffers from mp_host_alloc every transfer is
 * an asynchronous DMA that overlaps the kernels of the neighbouring chunks; ordinary (pageable) buffers work too, at the
 * runtime's staging speed.  NULL on failure (mp_last_error). */
void* mp_host_alloc(size_t bytes);
void mp_host_free(void* p);
/* proofs per pipelined chunk of the host-buffer entry points (0 restores the default: 65 536, and 131 072 for calls of 262 144 proofs or
 * more; the first chunks of a call are smaller -- the first upload and the last download are the copies nothing overlaps -- and the chunks
 * of a verify call grow by half from one to the next, because the copy of chunk k + 1 takes about as long as the kernels of chunk k) */
int mp_set_io_chunk(mp_table* t, size_t proofs);

/* ---- DLCards::setup [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:105-121] ---------------------------
 * "setup v2": G, ck_0..ck_{n-1}, H, gen -- n + 3 INDEPENDENT curve points sampled the way ark-ec's `C::rand` does from
 * ChaCha20Rng::from_seed(seed), in that order: x = Fq::rand, a sign bit, the square root of x^3 + a x + b (retry if there is none),
 * cofactor cleared.  Nobody, the holder of the seed included, knows a discrete logarithm between two of them (round 1 derived
 * them as k * G_std, which made the seed a trapdoor of the Pedersen key).  Host work, once per table.  out_params: n + 3 wire points. */
int mp_setup(mp_ctx* ctx, uint32_t m, uint32_t n, const uint8_t seed[32], uint8_t* out_params);

/* ---- table context: Parameters + aggregate public key -> fixed-base window tables in HBM ------------------
 * mp_table_create sizes the tables for the GPU it runs on: the widest windows -- 21 bits (48 GB at n = 26, 12 additions per term on the
 * 252-bit STARK scalars), 20 (27 GB, 13 additions), 16 (2 GB, 16 additions), 8 (16 MB, 32 additions) -- whose tables take at most 30 % of
 * the HBM that is free at the call and whose construction (~3.3x the table for a moment) fits 85 % of it.  On an otherwise empty
 * MI355X that is the configuration bench.py measures; a caller that keeps many tables alive (one per aggregate key) should rather
 * create ONE table of the shared parameters and name the keys through a key set (below).  Construction takes 0.1 s (8 / 16 bits) to
 * ~3 s (21 bits, first touch of 150 GB). */
int mp_table_create(mp_ctx* ctx, uint32_t m, uint32_t n, const uint8_t* params, const uint8_t* shared_key,
                    mp_table** out);
/* same with an explicit fixed-base window width: 8, 16, 20 or 21 bits (21: ceil(scalar bits / 21) windows -- 12 on the STARK curve, 13
 * on the other curves, where 20 bits give the same count for half the memory); 0 = choose as mp_table_create does */
int mp_table_create_ex(mp_ctx* ctx, uint32_t m, uint32_t n, const uint8_t* params, const uint8_t* shared_key,
                       uint32_t fb_window_bits, mp_table** out);
uint32_t mp_table_window_bits(const mp_table* t);      /* the width in use */
void mp_table_destroy(mp_table* t);

/* ---- DLCards::shuffle_and_remask / verify_shuffle (one proof; host buffers) --------------------------------- */
/* prover_seed: 32 FRESH uniformly random bytes per proof.  It stands for the reference's `rng: &mut R` [REF src/lib.rs:181-188]
 * and keys the ChaCha20 stream behind every blinding value of the argument; like any sigma-type proof, two proofs made from
 * one seed with different witnesses reveal the witnesses (permutation and masking factors).  (The sigma protocols further
 * down hedge their single nonce with witness and statement; the shuffle prover draws its O(N) blinders from the seed alone.) */
int mp_shuffle_and_remask(mp_table* t, const uint8_t* deck, const uint8_t* masking_factors,
                          const uint32_t* permutation, const uint8_t prover_seed[32], uint8_t* out_deck,
                          uint8_t* out_proof);
int mp_verify_shuffle(mp_table* t, const uint8_t* deck, const uint8_t* shuffled_deck, const uint8_t* proof,
                      size_t proof_len);

/* ---- batched forms: B independent proofs, arrays of the single-proof buffers back to back -------------------
 * status[b] receives the per-proof result; the return value is < 0 only for call-level errors. */
int mp_shuffle_and_remask_batch(mp_table* t, size_t B, const uint8_t* decks, const uint8_t* masking_factors,
                                const uint32_t* permutations, const uint8_t* prover_seeds, uint8_t* out_decks,
                                uint8_t* out_proofs, int32_t* status);
int mp_verify_shuffle_batch(mp_table* t, size_t B, const uint8_t* decks, const uint8_t* shuffled_decks,
                            const uint8_t* proofs, int32_t* status);

/* ---- device-resident forms: every pointer is a DEVICE pointer (HBM).  All kernels are enqueued on the context's stream
 * (mp_sync waits for them).  The prover returns without waiting.  The verifier waits ONCE inside the call when merged
 * verification is on (the default): after the screening pass it reads one 4-byte flag back to decide whether the per-equation
 * pass has to run (it does only if some proof failed), so the call returns when the screening kernels have finished;
 * with mp_set_merged_verify(t, 0) it does not wait either.  d_status is written by the stream in both cases: read it after
 * mp_sync.  These are what bench.py times (inputs already in HBM). */
int mp_shuffle_and_remask_batch_dev(mp_table* t, size_t B, const void* d_decks, const void* d_masking_factors,
                                    const void* d_permutations, const void* d_prover_seeds, void* d_out_decks,
                                    void* d_out_proofs, void* d_status);
int mp_verify_shuffle_batch_dev(mp_table* t, size_t B, const void* d_decks, const void* d_shuffled_decks,
                                const void* d_proofs, void* d_status);
/* ---- keyed batches: one aggregate public key PER PROOF ----------------------------------------------------------------
 * A card server runs many tables at once; the tables share the public parameters (one mp_table) and differ only in their
 * aggregate key, which the reference passes per call (`shared_key`, [REF mod.rs:380-386, 420-426]).  shared_keys /
 * d_keys: B wire points.  The key's terms become per-proof work: the verifier gains one variable-base term, the prover
 * builds the key's own window tables (2^(5w) pk, w < 51) for the N re-encryptions rho_i * pk -- about 9 % more work per
 * prove+verify than under the table's fixed key.  Results are byte-identical to a table created with that key. */
/* a table of the shared parameters alone (no aggregate key): accepts only the _keys entry points */
int mp_table_create_params(mp_ctx* ctx, uint32_t m, uint32_t n, const uint8_t* params, uint32_t fb_window_bits, mp_table** out);
int mp_shuffle_and_remask_batch_keys(mp_table* t, size_t B, const uint8_t* shared_keys, const uint8_t* decks,
                                     const uint8_t* masking_factors, const uint32_t* permutations, const uint8_t* prover_seeds,
                                     uint8_t* out_decks, uint8_t* out_proofs, int32_t* status);
int mp_verify_shuffle_batch_keys(mp_table* t, size_t B, const uint8_t* shared_keys, const uint8_t* decks,
                                 const uint8_t* shuffled_decks, const uint8_t* proofs, int32_t* status);
int mp_shuffle_and_remask_batch_keys_dev(mp_table* t, size_t B, const void* d_keys, const void* d_decks,
                                         const void* d_masking_factors, const void* d_permutations, const void* d_prover_seeds,
                                         void* d_out_decks, void* d_out_proofs, void* d_status);
int mp_verify_shuffle_batch_keys_dev(mp_table* t, size_t B, const void* d_keys, const void* d_decks, const void* d_shuffled_decks,
                                     const void* d_proofs, void* d_status);
/* ---- key sets: the aggregate keys of many card tables, prepared once ---------------------------------------------------------
 * A card table keeps its aggregate key for as long as its players stay [REF examples/round.rs:228-262: the key is computed once,
 * before the shuffles], so a server that runs many tables can hand their keys over ahead of time: mp_keyset_create builds
 * fixed-base window tables for every key (8-bit windows: 8 160 points = 0.5 MB per key on the 256-bit curves, in HBM) and the
 * _keyset entry points name a proof's key by its index in the set (d_key_index: B uint32, device memory).  The N
 * re-encryptions rho_i * pk then cost 32 table additions each and no per-proof table construction (keyed batches without a
 * set: 51 additions plus the key's own window tables per proof).  Results are byte-identical to the _keys entry points and to
 * a table created with that key.  keys: n_keys wire points in HOST memory; each must be a point of the prime-order group other
 * than the identity (MP_ERR_BAD_ENCODING otherwise).  An index >= n_keys gives status MP_ERR_BAD_ARGUMENT for that proof.
 * A key set belongs to the table it was created for and must be destroyed before it. */
typedef struct mp_keyset mp_keyset;
int mp_keyset_create(mp_table* t, size_t n_keys, const uint8_t* keys, mp_keyset** out);
void mp_keyset_destroy(mp_keyset* ks);
size_t mp_keyset_size(const mp_keyset* ks);
int mp_shuffle_and_remask_batch_keyset_dev(mp_table* t, const mp_keyset* ks, size_t B, const void* d_key_index, const void* d_decks,
                                           const void* d_masking_factors, const void* d_permutations, const void* d_prover_seeds,
                                           void* d_out_decks, void* d_out_proofs, void* d_status);
int mp_verify_shuffle_batch_keyset_dev(mp_table* t, const mp_keyset* ks, size_t B, const void* d_key_index, const void* d_decks,
                                       const void* d_shuffled_decks, const void* d_proofs, void* d_status);
/* ---- chain verification: the shuffle chain of a card table verified as one equation -------------------------------------------
 * A table's deck passes through `links` shuffles (deck_{j+1} = output of link j [REF examples/round.rs:268-350]) and every one of
 * them is verified.  These entry points verify `tables` chains at once; decks: (links + 1) x tables decks, deck j of table t at index
 * j * tables + t; proofs / status / keys: links x tables, link j of table t at j * tables + t (keys: the table's aggregate key repeated
 * per link; NULL = the mp_table's own key).  The merged equations of a chain's links are added up with weights that depend on every
 * proof of the chain, so every inner deck is a base once instead of twice and the n + 5 fixed bases appear once per table; the MSM runs on
 * the bucket kernel.  A table whose chain fails is re-verified link by link: status words are identical to mp_verify_shuffle_batch*. */
int mp_verify_shuffle_chain(mp_table* t, size_t tables, uint32_t links, const uint8_t* shared_keys, const uint8_t* decks,
                            const uint8_t* proofs, int32_t* status);
int mp_verify_shuffle_chain_dev(mp_table* t, size_t tables, uint32_t links, const void* d_keys, const void* d_decks, const void* d_proofs,
                                void* d_status);
int mp_sync(mp_ctx* ctx);
int mp_reserve(mp_table* t, size_t B);           /* pre-allocate the batch workspace for B proofs */
/* Every table holds six static work splits with identical results: a throughput plan (large sub-jobs, fewest operations) and five
 * finer ones -- wide, medium, latency, small, finest -- that give a proof more lanes (smaller sub-jobs; the windows of a variable-base
 * sub-job dealt to several lanes).  Batches of at most B / 20 proofs use the finest split, up to 0.3 B the small one, up to `B` the
 * latency plan, up to 2.4 B the medium plan, up to 19.2 B the wide plan, larger ones the throughput plan (default B = 2560 * 52 / N, at
 * least 40: 128 / 768 / 2 560 / 6 144 / 49 152 proofs of 52 cards, the crossovers measured on an MI355X; 0 = always throughput). */
int mp_set_latency_batch(mp_table* t, size_t B);
/* Every batch takes work split `split` whatever its size: 0 throughput, 1 latency, 2 medium, 3 finest, 4 wide, 5 small; -1 (default) =
 * by batch size as above.  For tests and measurements: the results do not depend on it. */
int mp_set_work_split(mp_table* t, int split);
/* Pipelined verification (depth 0 = off, the default).  A card server proves the next batch while the previous one is verified;
 * with depth >= 1 the device-resident verify calls (mp_verify_shuffle_batch[_keys|_keyset]_dev) run on a second lane of the context --
 * streams and arenas of their own -- ordered behind everything the context held when they were issued, and return WITHOUT waiting for
 * their screening verdict, so that the caller's next mp_shuffle_and_remask_batch*_dev call runs beside them on the chip (batches that
 * do not fill it alone).  The verdict of verify call k is looked at when verify call k + depth comes in, or at
 * mp_sync; only then -- and only if some proof failed the screen -- does the per-equation pass of call k run.  The caller's side of
 * the contract: the buffers a verify call reads (decks, shuffled decks, proofs, keys) stay untouched until `depth` further verify
 * calls on that table, or mp_sync, have returned (rotate depth + 1 sets of prover outputs); d_status is final after mp_sync, as
 * before.  Status words are identical in every mode.  The host-buffer entry points and chain verification are unaffected.
 * (A context owns four HIP streams -- two lanes of two -- which the runtime's default of four hardware queues serves.) */
int mp_set_pipeline(mp_table* t, int depth);
/* The sizes behind work split `split` (0 .. 5 as above): fixed-base / variable-base terms per sub-job, bases per window-table lane,
 * points per shared inversion, and `window_lanes` = lanes per variable-base sub-job: the Straus windows of a sub-job are dealt to that
 * many lanes (each runs its share as a chain of its own) and one fold per multi-scalar multiplication puts the range sums together --
 * k times the lanes for < 250 extra doublings per MSM instead of k times the doubling chains (1 = one lane runs all windows).
 * mp_set_plan_thresholds: the largest batch that takes the finest / small / latency / medium / wide split.  For measurements and tuning on
 * other parts: results do not depend on either.  Both rebuild nothing a running batch uses (call between batches). */
int mp_set_plan_params(mp_table* t, int split, uint32_t fixed_terms, uint32_t var_terms, uint32_t table_group, uint32_t norm_chunk,
                       uint32_t window_lanes);
int mp_set_plan_thresholds(mp_table* t, size_t finest, size_t small, size_t latency, size_t medium, size_t wide);
/* Verification strategy.  on (default): the verifier first evaluates ALL group equations of a proof merged into one
 * multi-scalar multiplication with random weights derived from the whole proof (soundness loss ~2^-250); a batch in which
 * every proof passes ends there.  The proofs that fail it -- they and nobody else (round 5, mp_set_group_refine below) -- are evaluated
 * equation by equation, so that the FIRST failing check is reported by name exactly as the reference does [REF tests.rs:223-225].
 * off: always evaluate the equations one by one.  Results (status words) are identical in both modes. */
int mp_set_merged_verify(mp_table* t, int on);
/* Group verification (round 4; on by default).  The screening pass of a batch of at least `min_batch` x 52 / N proofs (default 6 144 for
 * 52-card decks) adds the merged equations of a GROUP of proofs with weights derived from every proof of the group and evaluates the
 * sum as ONE multi-scalar multiplication by the bucket method: 18 to 32 additions per point instead of 51 plus window tables and
 * doubling chains, and the fixed bases once per group.  Round 6: from 8/3 `min_batch` (16 384) proofs on, the equations hold up to
 * `points_per_group` points (default 243 712: a proof brings 4N + 11m + 8, so 1 024 proofs of a 52-card deck, 64 of a 1 024-card one)
 * and run on the split pipeline (mp_set_bucket_split: 14-bit windows, 18 additions per point) as long as min_batch / 512 (12)
 * equations of at least 50 000 points are left; smaller batches take the rule of rounds 4-5 -- at most 30 464 points (128 proofs) and no
 * fewer than 2/13 `min_batch` (945) groups: 8 proofs per group at 8 192 in flight, one wave per window.  points_per_group <= 65 535
 * asks for that rule at every batch size (30 464 = the default of rounds 4-5).  The group size is the divisor of the batch size nearest
 * to the target (between half and twice it; a batch without one keeps the per-proof screen).  BLS12-377: the split pipeline only (its
 * hot loop is spill-free on the 14-limb field; equations from 40 000 points on, 11-bit windows included).  A group whose equation
 * fails is looked at more closely -- ITS members only (below): status words are identical to every other strategy.  points_per_group
 * = 0 switches it off.  Needs merged verification on.  mp_group_size: the group size a batch of B proofs takes under the table's own
 * key (0: per-proof screen).  (52-card decks at 262 144 in flight: 524 k proofs/s without groups, 672 k with the groups of round 5,
 * 717 k with those of round 6 on the same kind of box.) */
int mp_set_group_verify(mp_table* t, uint32_t points_per_group, size_t min_batch);
uint32_t mp_group_size(const mp_table* t, size_t B);
/* What a rejected proof costs (round 5).  A screen that fails -- the merged equation of one proof, the equation of a group of proofs,
 * of a chain -- names the proofs it could not clear; only THEY are looked at again: their inputs are gathered into a contiguous
 * sub-batch on the device, the sub-batch takes the next finer pass and its status words are written back over the screen's marks.
 * Everybody else's verdict stands, as in the reference, where one call verifies one proof [REF src/discrete_log_cards/mod.rs:420-443].
 * The members of failing groups go through equations of sub-groups of `points_per_subgroup` points (0 = default: an eighth of the group
 * equation's -- 128 proofs of a 52-card deck under an equation of 1 024, and an eighth of THAT, 16, for the members of failing
 * sub-groups; a sixty-fourth at once when nearly every group of the call failed) when there are at least `min_subgroups` of them (0 =
 * default 128: enough to fill the bucket kernel), otherwise -- and after two such levels always -- through the per-equation pass that
 * names the first failing check.  One tampered proof among 262 144 sends the 1 024 members of its equation through that pass: +2.7 ms
 * on a 90 ms verify call.  The suspects are gathered and looked at in slices of at most 131 072 52-card proofs (bounded memory whatever
 * their number), and until a suspect has its own word its status reads MP_ERR_INTERNAL, never 0.  mp_reverified_count: proofs that
 * have taken a per-equation pass on this table because a screen could not clear them. */
int mp_set_group_refine(mp_table* t, uint32_t points_per_subgroup, uint32_t min_subgroups);
uint64_t mp_reverified_count(const mp_table* t);
/* Groups that adapt to the rejection rate (default on).  A group fails if any member does, so with a fraction p of bad proofs in the traffic
 * 1 - (1 - p)^L of the groups of L fail (72 % of the groups of 128 at p = 1 %) and their members pay a finer pass on top of a screen that
 * cleared nobody.  The table remembers the last screens: when more than a fifth of a call's groups fail the next call takes groups of half
 * the size (down to 8; by three halvings at once when nineteen groups in twenty fail), when fewer than 4 % fail the size goes back up,
 * one step per call, two when no group failed (calls with at least 8 groups count, a step down needs four failing groups).  Honest
 * traffic never leaves the default size; verdicts do not depend on it.  on = 0 pins the default (and resets the memory); mp_group_size
 * reports the size the NEXT call of B proofs takes. */
int mp_set_group_adapt(mp_table* t, int on);
/* Variable-base MSMs with at least `terms` terms (default 2048: the verifier's products over a 1024-card deck) run on the
 * wave-cooperative bucket-method kernel (counting sort by wavefront prefix sum, balanced bucket shares, wave-wide bucket reduction),
 * smaller ones on the Straus kernel with per-proof window tables; 0 = never.  Results are identical; the split is a property of
 * the table's static plans, which this call rebuilds. */
int mp_set_bucket_min(mp_table* t, size_t terms);
/* Window width of the bucket method: 8 to 14 bits (128 to 8 192 buckets per window; 32, 28, 26, 23, 21, 20 or 18 windows per 252-bit
 * scalar -- the smaller of k and q - k is recoded, signs flipped: ceil(bits / c) windows), or 0 (default) = by the size of the MSM (8
 * bits below 6 000 terms, 9 below 12 000, 10 below 40 000, 11 below 50 000, 12 below 200 000, 14 from there on -- the equation of a
 * group of 1 024 52-card proofs, mp_set_group_verify).  Results are identical; rebuilds the static plans like mp_set_bucket_min. */
int mp_set_bucket_bits(mp_table* t, uint32_t bits);
/* Round 6: inputs the caller has validated ONCE are not validated again in every call that touches them.  The reference's trait takes
 * typed arkworks points, validated when they were deserialised [REF examples/parameter_selection.rs:78-91]; this engine takes wire
 * bytes and by default tests every point of every call for membership of the prime-order subgroup (curves with a cofactor: BLS12-377)
 * -- a deck that passes along a chain of shuffles was tested in the call that proved it, the call that verified it and the call that
 * shuffled it on.  `what` = OR of MP_VALIDATED_*: the table's following calls skip the subgroup test of those inputs (range, curve
 * equation and permutation checks stay).  A deck is "validated" if it came out of mp_deck_deserialize_dev / mp_deck_deserialize with
 * status 0, out of mp_deck_validate_dev with status 0, or out of this engine's own prover.  0 (default) = test everything. */
#define MP_VALIDATED_DECKS 1u     /* the input decks of prove and verify calls */
#define MP_VALIDATED_SHUFFLED 2u  /* the shuffled decks of verify calls */
#define MP_VALIDATED_PROOFS 4u    /* the points of the proofs (only if they came through mp_proof_deserialize) */
int mp_set_validated(mp_table* t, uint32_t what);
/* `decks` wire-v1 decks of the table's size in DEVICE memory -> one int32 per deck in d_status: 0, or MP_ERR_BAD_ENCODING if a point is
 * not canonical, not on the curve or outside the prime-order subgroup.  The once-per-deck validation that mp_set_validated relies on. */
int mp_deck_validate_dev(mp_table* t, size_t decks, const void* d_wire_decks, void* d_status);
/* Round 6.  Bucket jobs whose windows are at least `min_bits` wide (default 12: equations of 50 000 points and more; 11 on BLS12-377)
 * run as a pipeline of kernels instead of one wave per (equation, window): k_bucket_sort (a workgroup per 24 576 terms: counting sort
 * inside LDS, the sorted run written in whole lines), k_bucket_acc (a wave per range of 256 buckets, four per lane dealt by rank, one
 * mixed addition per term; k_bucket_list -- equal shares of the sorted list -- where a window's digits crowd into a few buckets),
 * k_bucket_reduce (four waves per window, a quarter of the buckets each) and k_bucket_final.  10 .. 15 (= none).  Results are identical. */
int mp_set_bucket_split(mp_table* t, uint32_t min_bits);
/* Chain verification (mp_verify_shuffle_chain*): at most `links` links share one chain equation; longer chains are verified as
 * consecutive sub-chains.  0 (default) = as many as fit the 32 767 points of one equation (293 links of a 52-card deck; one link of a deck too large for that gets an equation of up to 65 535 points).  A smaller
 * value bounds the work that is repeated link by link when a chain fails.  Verdicts are the same. */
int mp_set_chain_max_links(mp_table* t, uint32_t links);
/* Chain verification of many tables at once: the chain equations of `tables_per_equation` tables are added up (with weights that
 * depend on every proof of every member, as in mp_set_group_verify) into ONE equation -- 64 tables x 4 392 points for 32 links of a
 * 52-card deck (round 6; 8 tables in round 5): 14-bit windows on the split pipeline, the points of an equation copied once into a
 * contiguous run (k_chain_tile), the fixed bases once per 64 tables.  0 (default) = by size: the divisor of `tables` that brings the
 * equation nearest to the group equation's points (mp_set_group_verify, under the same rule of how many equations must be left); 1 =
 * every table on its own (rounds 2-4); other values (up to 4 094, tables x (links + 1) <= 4 094) = the divisor of `tables` nearest to
 * it, whatever mp_set_group_verify says.  Tables g (tables / G) + e, g < G, share equation e.  If an equation fails, the links of ITS
 * tables are re-verified one by one: status words are identical in every setting. */
int mp_set_chain_group(mp_table* t, uint32_t tables_per_equation);
/* Chain verification in passes of `tables_per_pass` tables.  The workspace of chain verification is ~68 KB per link in flight (52-card
 * decks: 107 GB for 49 152 tables x 32 links), far more than the 13 KB of deck and proof a link occupies, while the prover wants as many
 * tables per call as there are (link j of every table is one batch).  0 (default) = one pass if the workspace fits the free device
 * memory, equal passes of whole thousands of tables otherwise; the rows of a pass are gathered from the link-major arrays on the device
 * (one copy per link and array).  Verdicts do not depend on it. */
int mp_set_chain_slice(mp_table* t, size_t tables_per_pass);
/* What a chain call does under the current settings: tables per equation for a pass of `tables` tables x `links` links (1 = every table
 * on its own), and the tables per pass the last mp_verify_shuffle_chain[_dev] call on this table took (0 = none yet). */
uint32_t mp_chain_group_size(const mp_table* t, size_t tables, uint32_t links, int keyed);
size_t mp_chain_last_slice(const mp_table* t);
/* Lanes per Fiat-Shamir transcript.  A proof's transcript is one BLAKE2s chain (13.6 KB of statement for a 52-card deck): 1 = one
 * lane per proof (what a batch that fills the chip wants), 4 = the four G functions of a half-round on four adjacent lanes (2.7x
 * fewer instructions in the chain: what a single proof or a few thousand large decks wait for), 0 (default) = 4 for batches of up
 * to 32 768 proofs, 1 above.  Digests, challenges and proofs are the same. */
int mp_set_transcript_lanes(mp_table* t, uint32_t lanes);
/* Lanes per group operation in the dependency chains of the multi-scalar multiplications (the Straus accumulators, the window fold
 * of the bucket method).  1 = one lane per chain (fewest instructions: what a batch that fills the chip wants); 4 = the independent
 * field products of a doubling / addition on four adjacent lanes, 3-4 products deep instead of 10-14 (what a single proof waits for:
 * every MSM has ~250 dependent doublings); 0 (default) = 4 while the chains of a launch need at most 65 536 lanes that way (a few
 * dozen 52-card proofs), 1 above.  The results are the same group elements. */
int mp_set_group_lanes(mp_table* t, uint32_t lanes);
/* How the prover evaluates the multi-exponentiation diagonals E_k (a polynomial product of the scalar rows with the ciphertext
 * rows) for 3 <= m <= 16.  on (default): Toom-Cook with the 2m points 0, inf, +-1 .. +-(m-1) -- 2m row products; off: recursive
 * Karatsuba (13 products at m = 4, 35 at m = 8; what m > 16 always uses).  m = 2 always uses its 4-point Toom-Cook form.  The
 * E_k are the same group elements either way: proof bytes do not change.  Rebuilds the table's static plans. */
int mp_set_toom_cook(mp_table* t, int on);
/* Curves with a cofactor (MP_CURVE_BLS12_377): every wire point of a call -- decks, keys, proof elements -- is tested for
 * membership in the prime-order subgroup ([q]P == O), as ark-ec's validating deserialiser does; failures give
 * MP_ERR_BAD_ENCODING for that proof.  on by default; a caller whose points were already validated (e.g. deserialised by
 * arkworks with checks) may switch it off -- it costs about three verifications per proof.  No effect on prime-order curves. */
int mp_set_subgroup_check(mp_table* t, int on);

/* ---- building blocks (host buffers) ------------------------------------------------------------------------
 * mp_remask_batch: out[i] = in[i] + (rho_i * G, rho_i * pk)        [REF remasking.rs:16-18]
 * mp_msm:          n_msm independent variable-base MSMs of k terms: out[j] = sum_t scalars[j][t] * points[j][t]
 * mp_commit_batch: Pedersen com(v; r) = r*H + sum v_l * ck_l, `count` commitments of `len` <= n values each */
int mp_remask_batch(mp_table* t, size_t count, const uint8_t* cards, const uint8_t* masking_factors, uint8_t* out);
int mp_msm(mp_table* t, size_t n_msm, size_t k, const uint8_t* scalars, const uint8_t* points, uint8_t* out);
int mp_commit_batch(mp_table* t, size_t count, size_t len, const uint8_t* values, const uint8_t* r, uint8_t* out);

/* ---- sigma protocols behind the rest of DLCards (host buffers) -------------------------------------
 * nbases = 1: Schnorr identification (prove/verify_key_ownership, [REF mod.rs:132-165]);
 * nbases = 2: Chaum-Pedersen DL equality (mask / remask / reveal proofs and their verifiers, [REF mod.rs:182-357]).
 * Per proof: bases g_i, publics a_i = x * g_i (nbases points each), witness x, `fs_init` = Blake2s digest of the bytes the
 * reference seeds its FiatShamirRng with (mp_blake2s of e.g. b"Masking Proof"), prover seed; proof = A_1..A_nb || z.
 * verify status: 0 Ok, 5 "Schnorr Identification", 6 "Chaum-Pedersen" [REF tests.rs:74-76,120,170], < 0 usage error.
 * The nonce is hedged ("sigma transcript v2"): r = Fr::rand(ChaCha20Rng(Blake2s(ToBytes(bases, publics) || Blake2s(witness || fs_init ||
 * prover_seed)))), so a repeated seed repeats the nonce only if witness and statement repeat too (the reference's `rng: &mut R` advances
 * by itself; an explicit seed does not).  Seeds should still be fresh CSPRNG output per call (`rng.fill_bytes`). */
int mp_sigma_prove_batch(mp_table* t, size_t B, uint32_t nbases, const uint8_t* bases, const uint8_t* publics,
                         const uint8_t* witness, const uint8_t* fs_init, const uint8_t* prover_seeds, uint8_t* out_proofs,
                         int32_t* status);
int mp_sigma_verify_batch(mp_table* t, size_t B, uint32_t nbases, const uint8_t* bases, const uint8_t* publics,
                          const uint8_t* proofs, const uint8_t* fs_init, int32_t* status);
int mp_blake2s(const uint8_t* in, size_t len, uint8_t out[32]);   /* host helper: BLAKE2s-256 */

/* ---- canonical serialisation (arkworks-0.3 `CanonicalSerialize` / `CanonicalDeserialize`, compressed) ----------------------
 * Every associated type of the trait is CanonicalSerialize + CanonicalDeserialize [REF src/lib.rs:45-71], and the reference's
 * harness measures `proof.serialized_size()` [REF examples/parameter_selection.rs:95]: these are the conversions between those
 * byte strings and the engine's wire v1.  Host work (no context, no GPU); deserialisation validates like ark-ec (canonical x, x on
 * the curve, prime-order subgroup on curves with a cofactor, scalars < q) and costs one square root per point.
 *   Fr 32 B LE | point: x LE in ceil((bits+2)/8) bytes, bit 7 of the last byte = (y > -y), bit 6 = infinity | Vec<T>: u64 LE length
 *   MaskedCard = 2 points; deck = Vec<MaskedCard>; Parameters { m, n, enc { G }, commit { Vec ck, H }, gen } [REF mod.rs:37-43];
 *   proof: elements in wire-v1 order, one Vec per vector-valued element (upstream's struct lives in the un-vendored dependency,
 *   so sizes are exact for this grouping; byte-compatibility with upstream's layout is not claimed).
 * Sizes: caller allocates; errors: MP_ERR_BAD_ENCODING (invalid data) / MP_ERR_BAD_ARGUMENT. */
size_t mp_serialized_point_size(int curve_id);                               /* 32 (STARK, bn254), 33 (secp256k1), 48 (BLS12-377) */
size_t mp_serialized_deck_size(int curve_id, size_t cards);
size_t mp_serialized_params_size(int curve_id, uint32_t n);
size_t mp_serialized_proof_size(int curve_id, uint32_t m, uint32_t n);      /* = ZKProofShuffle::serialized_size() */
int mp_points_serialize(int curve_id, size_t count, const uint8_t* wire_points, uint8_t* out);
int mp_points_deserialize(int curve_id, size_t count, const uint8_t* data, uint8_t* out_wire_points);
int mp_deck_serialize(int curve_id, size_t cards, const uint8_t* wire_deck, uint8_t* out);
int mp_deck_deserialize(int curve_id, const uint8_t* data, size_t len, size_t max_cards, uint8_t* out_wire_deck, size_t* out_cards);
int mp_params_serialize(int curve_id, uint32_t m, uint32_t n, const uint8_t* raw_params, uint8_t* out);
int mp_params_deserialize(int curve_id, const uint8_t* data, size_t len, size_t max_n, uint32_t* m, uint32_t* n, uint8_t* out_raw_params);
int mp_proof_serialize(int curve_id, uint32_t m, uint32_t n, const uint8_t* proof_wire, uint8_t* out);
int mp_proof_deserialize(int curve_id, uint32_t m, uint32_t n, const uint8_t* data, size_t len, uint8_t* out_proof_wire);
/* The same conversion ON THE DEVICE for the bulk data (round 4): decks that arrive as arkworks bytes go to HBM as they are -- half the
 * bytes of wire v1 over PCIe -- and are decompressed there, one lane per point, the square root as a windowed walk through the 2-Sylow
 * subgroup of the base field (4 512 squarings on the STARK prime, whose p - 1 has 2-adicity 192; the host path needs ~10^4), with the
 * validation of the host functions above: canonical x, flags, x on the curve, prime-order subgroup where there is a cofactor.  All
 * pointers are DEVICE pointers; the kernel is enqueued on the context's stream (mp_sync, or any later call on the context, orders
 * behind it).  d_status receives 0 or MP_ERR_BAD_ENCODING: one int32 per point (mp_points_deserialize_dev: `count` compressed points
 * back to back) / per deck (mp_deck_deserialize_dev: `decks` serialised Vec<MaskedCard> of `cards` cards each, back to back, every
 * one mp_serialized_deck_size bytes with its u64 length in front); a failing point leaves an all-zero wire point.  The output of
 * mp_deck_deserialize_dev is what mp_shuffle_and_remask_batch_dev / mp_verify_shuffle_batch_dev take as d_decks. */
int mp_points_deserialize_dev(mp_ctx* ctx, size_t count, const void* d_data, void* d_out_wire_points, void* d_status);
int mp_deck_deserialize_dev(mp_ctx* ctx, size_t decks, size_t cards, const void* d_data, void* d_out_wire_decks, void* d_status);

/* ---- measurement hooks ---------------------------------------------------------------------------------------
 * With profiling on, every kernel launch is bracketed by HIP events on the context's stream;
 * mp_profile_report writes "name count total_ms items\n" lines (and resets) -- bench.py's roofline source; items = threads launched (waves
 * for the wave-cooperative kernels, (equation, window) items for k_bucket_msm), summed over the launches of that name. */
int mp_profile_enable(mp_ctx* ctx, int on);
int mp_profile_report(mp_ctx* ctx, char* buf, size_t buf_len);
/* static work census of one prove / verify for this table: number of scalar*point terms and point operations */
int mp_work_census(mp_table* t, uint64_t* prove_terms, uint64_t* verify_terms, uint64_t* prove_point_ops,
                   uint64_t* verify_point_ops);

/* static plan statistics for this table (what one proof costs, per kernel class); out[16]:
 *  [0..5]  prove : fixed terms, var terms, fixed jobs, var jobs, table bases, combine terms
 *  [6..11] verify: same six
 *  [12] variable-base windows per scalar  [13] fixed-base windows per scalar  [14] N  [15] reserved */
int mp_plan_stats(mp_table* t, uint64_t out[16]);

#ifdef __cplusplus
}
#endif
#endif /* MPSHUFFLE_H */
