"""CPU tests (-m "not gpu"): the C-ABI library loads and exports every declared symbol (no compute without a
GPU), the engine refuses to run without a device (no CPU fallback), the host-side planners are sane, and the
development emulator (kernel bodies as CPU loops, tools/hostemu) reproduces the oracle byte for byte."""
import ctypes
import importlib
import os
import re
import subprocess

import pytest

from conftest import GOLDEN, ROOT, load_json


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "mpshuffle.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mp_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def native(mp):
    mp.build()
    return mp


def test_library_exports_every_declared_symbol(native):
    lib = native.load()
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(native._native.SYMBOLS) == syms
    assert lib.mp_proof_size(2, 26) == (11 * 2 + 8) * 64 + (5 * 26 + 9) * 32 == 6368
    assert lib.mp_params_size(26) == 29 * 64
    assert lib.mp_check_name(1) == b"Hadamard Product (5.1)"
    assert lib.mp_check_name(0) == b"Ok"


def test_no_cpu_fallback_without_a_device(native):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(native.NoDeviceError):
        native.Engine("stark", 0)
    with pytest.raises(native.NoDeviceError):
        native.DLCards("stark")


def test_mirror_types(native):
    p = native.Permutation([2, 0, 1])
    assert p.permute_array(["a", "b", "c"]) == ["c", "a", "b"]
    rng = native.ChaCha20Rng(bytes(32))
    assert rng.next_u64().to_bytes(8, "little").hex() == "76b8e0ada0f13d90"
    q = native.Permutation.new(native.ChaCha20Rng(bytes(32)), 52)
    assert sorted(q.mapping) == list(range(52))
    assert native.CryptoError("x") == native.CryptoError("x") != native.CryptoError("y")
    with pytest.raises(native.CardProtocolError):
        native.Parameters(2, 3, b"\0" * 5)


@pytest.fixture(scope="module")
def emu(native):
    d = os.path.join(ROOT, "tools", "hostemu")
    subprocess.check_call(["make", "-s", "-j8", "-C", d])
    lib = native._native.bind(ctypes.CDLL(os.path.join(d, "libmpemu.so")))
    return lambda curve: native._native.Engine(curve, 0, lib=lib)


@pytest.mark.parametrize("name", ["shuffle_stark_m2_n3_s1.json", "shuffle_stark_m3_n4_s11.json", "shuffle_bn254_m2_n4_s3.json",
                                  "shuffle_secp256k1_m3_n3_s5.json", "shuffle_stark_m4_n13_s9.json",
                                  "shuffle_bls12_377_m2_n3_s13.json"])
def test_kernel_bodies_under_emulation_match_golden(emu, native, name):
    g = load_json(os.path.join(GOLDEN, name))
    eng = emu(g["curve"])
    m, n = g["m"], g["n"]
    t = eng.table(m, n, bytes.fromhex(g["params"]), bytes.fromhex(g["pk"]))
    # a single proof takes the four-lane transcripts and group operations by default (k_fsq_*, kernels_quad.hpp); lanes = 1 forces
    # the kernels a full batch runs (one lane per proof, one lane per chain)
    for latency_batch, lanes in ((8192, 0), (8, 0), (0, 0), (8192, 1), (0, 1), (-4, 0)):   # finest split, latency plan, throughput plan (large sub-jobs, Toom-Cook for m = 2); -4: the wide split forced
        t.set_work_split(4 if latency_batch < 0 else -1)
        t.set_latency_batch(max(latency_batch, 0))
        t.set_transcript_lanes(lanes)
        t.set_group_lanes(lanes)
        eng.profile_enable(True)
        deck, proof = t.shuffle_and_remask(bytes.fromhex(g["deck"]), bytes.fromhex(g["rho"]), g["perm"], bytes.fromhex(g["prover_seed"]))
        assert deck.hex() == g["shuffled"]
        assert proof.hex() == g["proof"]
        assert t.verify_shuffle(bytes.fromhex(g["deck"]), deck, proof) == 0
        rep = eng.profile_report()
        eng.profile_enable(False)
        assert ("k_fsq_verify" in rep) == (lanes == 0) and ("k_verify_fs" in rep) == (lanes == 1), sorted(rep)
        assert ("k_fixed_msm_q" in rep) == (lanes == 0) and ("k_fixed_msm" in rep) == (lanes == 1), sorted(rep)
    with pytest.raises(native.NativeError):
        t.set_transcript_lanes(3)
    with pytest.raises(native.NativeError):
        t.set_group_lanes(2)
    with pytest.raises(native.NativeError):
        t.set_work_split(5)
    t.set_work_split(-1)
    t.set_transcript_lanes(0)
    t.set_group_lanes(0)
    bad = bytearray(proof)
    bad[-1] ^= 0          # unchanged copy still verifies
    cb = 2 * eng.point_bytes          # one card = two points
    swapped = deck[cb:2 * cb] + deck[0:cb] + deck[2 * cb:]
    assert eng.check_name(t.verify_shuffle(bytes.fromhex(g["deck"]), swapped, proof)) == "Hadamard Product (5.1)"
    t.close()


def test_emulated_chunked_host_pipeline(emu, coracle):
    """the host-buffer entry points cut a batch into chunks (upload / kernels / download pipelined): 5 proofs in chunks of 2
    give the same bytes as one chunk, with and without per-proof keys"""
    cv, m, n = "stark", 2, 3
    eng = emu(cv)
    ins = [coracle.gen_inputs(cv, m, n, 800 + b) for b in range(5)]
    g0 = ins[0]
    t = eng.table(m, n, g0["params"], g0["pk"])
    args = (b"".join(g["deck"] for g in ins), b"".join(g["rho"] for g in ins), [v for g in ins for v in g["perm"]],
            b"".join(g["prover_seed"] for g in ins))
    keys = b"".join(g["pk"] for g in ins)
    ref = t.shuffle_and_remask_batch(*args)
    refk = t.shuffle_and_remask_batch_keys(keys, *args)
    t.set_io_chunk(2)
    assert t.shuffle_and_remask_batch(*args) == ref
    assert t.shuffle_and_remask_batch_keys(keys, *args) == refk
    assert t.verify_shuffle_batch(args[0], ref[0], ref[1]) == [0] * 5
    assert t.verify_shuffle_batch_keys(keys, args[0], refk[0], refk[1]) == [0] * 5
    ps = t.proof_bytes
    swapped = ref[1][ps:2 * ps] + ref[1][:ps] + ref[1][2 * ps:]     # proofs 0 and 1 exchanged: both fail, the other chunks pass
    st = t.verify_shuffle_batch(args[0], ref[0], swapped)
    assert st[0] > 0 and st[1] > 0 and st[2:] == [0] * 3
    t.set_io_chunk(0)
    t.close()


def test_emulated_keyed_batch(emu, coracle, native):
    """keyed batches (one aggregate key per proof, mp_*_batch_keys): byte-identical to the oracle run under each proof's key"""
    for cv, m, n in (("stark", 2, 3), ("bls12_377", 2, 3)):
        eng = emu(cv)
        ins = [coracle.gen_inputs(cv, m, n, 500 + b) for b in range(3)]
        g0 = ins[0]
        t = eng.table(m, n, g0["params"], g0["pk"])
        keys = b"".join(g["pk"] for g in ins)
        decks = b"".join(g["deck"] for g in ins)
        for lb in (8192, 8, 2, 0):         # finest, latency, medium (2 < B = 3 <= 7) and throughput plans
            t.set_latency_batch(lb)
            d, p, st = t.shuffle_and_remask_batch_keys(keys, decks, b"".join(g["rho"] for g in ins), [v for g in ins for v in g["perm"]],
                                                       b"".join(g["prover_seed"] for g in ins))
            assert st == [0, 0, 0]
            cb, ps = len(g0["deck"]), t.proof_bytes
            for b, g in enumerate(ins):
                ed, ep = coracle.shuffle_and_remask(cv, m, n, g0["params"], g["pk"], g["deck"], g["rho"], g["perm"], g["prover_seed"])
                assert d[b * cb:(b + 1) * cb] == ed and p[b * ps:(b + 1) * ps] == ep
            assert t.verify_shuffle_batch_keys(keys, decks, d, p) == [0, 0, 0]
            wrong = ins[0]["pk"] + ins[0]["pk"] + ins[2]["pk"]          # proof 1 checked under another table's key
            assert t.verify_shuffle_batch_keys(wrong, decks, d, p) == [0, 1, 0]
        t.close()
    # a table made from the parameters alone serves keyed batches and refuses the fixed-key entry points
    eng = emu("stark")
    g = coracle.gen_inputs("stark", 2, 3, 500)
    t = eng.table(2, 3, g["params"], None)
    d, p, st = t.shuffle_and_remask_batch_keys(g["pk"], g["deck"], g["rho"], g["perm"], g["prover_seed"])
    assert st == [0] and (d, p) == coracle.shuffle_and_remask("stark", 2, 3, g["params"], g["pk"], g["deck"], g["rho"], g["perm"], g["prover_seed"])
    assert t.verify_shuffle_batch_keys(g["pk"], g["deck"], d, p) == [0]
    with pytest.raises(native.NativeError):
        t.verify_shuffle_batch(g["deck"], d, p)
    t.close()


def test_emulated_key_sets(emu, coracle, native):
    """key sets (mp_keyset_create + the _keyset entry points): the keys of several card tables prepared once, proofs name their
    key by index -- byte-identical to the oracle under each proof's key, and to the _keys entry points.  The emulator's "device"
    memory is host memory, so ctypes buffers stand in for the device arrays."""
    import numpy as np
    for cv, m, n in (("stark", 2, 3), ("bls12_377", 2, 3)):
        eng = emu(cv)
        ins = [coracle.gen_inputs(cv, m, n, 520 + b) for b in range(3)]
        g0 = ins[0]
        t = eng.table(m, n, g0["params"], None)
        keys = [coracle.gen_inputs(cv, m, n, 530 + k)["pk"] for k in range(2)]
        ks = t.keyset(b"".join(keys))
        assert t.lib.mp_keyset_size(ks.h) == 2
        kidx = np.array([1, 0, 1], dtype=np.uint32)
        B, N, cb, ps = 3, m * n, len(g0["deck"]), t.proof_bytes
        buf = lambda b: np.frombuffer(b, dtype=np.uint8).copy()
        decks, rho = buf(b"".join(g["deck"] for g in ins)), buf(b"".join(g["rho"] for g in ins))
        perms = np.array([v for g in ins for v in g["perm"]], dtype=np.uint32)
        seeds = buf(b"".join(g["prover_seed"] for g in ins))
        ptr = lambda a: a.ctypes.data
        for lb in (8192, 8, 0):
            t.set_latency_batch(lb)
            out_d, out_p = np.zeros(B * cb, dtype=np.uint8), np.zeros(B * ps, dtype=np.uint8)
            st = np.full(B, 77, dtype=np.int32)
            t.shuffle_and_remask_batch_keyset_dev(ks, B, ptr(kidx), ptr(decks), ptr(rho), ptr(perms), ptr(seeds), ptr(out_d), ptr(out_p), ptr(st))
            eng.sync() if hasattr(eng, "sync") else None
            assert st.tolist() == [0, 0, 0]
            for b, g in enumerate(ins):
                ed, ep = coracle.shuffle_and_remask(cv, m, n, g0["params"], keys[kidx[b]], g["deck"], g["rho"], g["perm"], g["prover_seed"])
                assert out_d[b * cb:(b + 1) * cb].tobytes() == ed and out_p[b * ps:(b + 1) * ps].tobytes() == ep
            sv = np.full(B, 77, dtype=np.int32)
            t.verify_shuffle_batch_keyset_dev(ks, B, ptr(kidx), ptr(decks), ptr(out_d), ptr(out_p), ptr(sv))
            assert sv.tolist() == [0, 0, 0]
            other = np.array([1, 1, 1], dtype=np.uint32)           # proof 1 checked under the other table's key
            t.verify_shuffle_batch_keyset_dev(ks, B, ptr(other), ptr(decks), ptr(out_d), ptr(out_p), ptr(sv))
            assert sv.tolist() == [0, 1, 0]
            assert t.verify_shuffle_batch_keys(b"".join(keys[i] for i in kidx), decks.tobytes(), out_d.tobytes(), out_p.tobytes()) == [0, 0, 0]
        bad = np.array([1, 2, 0], dtype=np.uint32)                 # no key 2 in the set
        t.shuffle_and_remask_batch_keyset_dev(ks, B, ptr(bad), ptr(decks), ptr(rho), ptr(perms), ptr(seeds), ptr(out_d), ptr(out_p), ptr(st))
        assert st.tolist() == [0, native._native.MP_ERR_BAD_ARGUMENT, 0]
        with pytest.raises(native.NativeError):
            t.keyset(bytes(len(keys[0])))                          # the identity / not a curve point
        t2 = eng.table(m, n, g0["params"], g0["pk"])
        with pytest.raises(native.NativeError):                    # a key set belongs to its table
            t2.verify_shuffle_batch_keyset_dev(ks, B, ptr(kidx), ptr(decks), ptr(out_d), ptr(out_p), ptr(sv))
        t2.close()
        ks.close()
        t.close()


def test_division_step_inversion_matches_fermat():
    """field.hpp fe_inv_divsteps (Bernstein-Yang division steps on 29-bit limbs, what every field inverts with) against the Fermat
    ladder it replaced: edge values, lazily reduced representatives and random residues on the four base fields and two scalar fields"""
    exe = os.path.join(ROOT, "tests", "cpp", "_inv_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-include", os.path.join(ROOT, "tools", "hostemu", "rt.hpp"),
                           "-I" + os.path.join(ROOT, "tools", "hostemu"), "-I" + os.path.join(ROOT, "mental-poker_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "inv_check.cpp"), "-o", exe])
    out = subprocess.run([exe, "30000"], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert out.count(" 0 mismatches, 0 answered by the fallback") == 6, out


@pytest.mark.parametrize("cvn", ["stark", "bn254", "secp256k1", "bls12_377"])
def test_emulated_edge_inputs(emu, cvn):
    """identity permutation, rho in {0, 1, q-1}, duplicate cards, a card and its negative, a point-at-infinity component: the kernel
    bodies under emulation give the Python oracle's bytes on every base-field form (equal points meet as P + P and P - P in the window
    tables and accumulators, where a lazily reduced zero has to be recognised) in all three work splits"""
    import mp_oracle as po
    m, n = 2, 3
    cv = po.CURVES[cvn]
    with po.curve_ctx(cv):
        pp, pk, deck, rho, perm, ps = po.gen_inputs(cv, m, n, 42)
        deck[1] = deck[0]
        deck[2] = (None, deck[2][1])
        deck[4] = (deck[3][0], po.pt_neg(cv, deck[3][1]))
        rho = [0, 1, cv.q - 1, rho[3], rho[4], 0]
        perm = list(range(m * n))
        sh, pf = po.shuffle_and_remask(pp, pk, deck, rho, perm, ps)
        want_deck, want_proof, wdeck, wparams, wpk = po.deck_to_bytes(sh), po.proof_to_bytes(pf), po.deck_to_bytes(deck), po.params_to_bytes(pp), po.pt_wire(pk)
    eng = emu(cvn)
    t = eng.table(m, n, wparams, wpk)
    rho_b = b"".join(r.to_bytes(32, "little") for r in rho)
    for latency_batch in (8192, 8, 0):
        t.set_latency_batch(latency_batch)
        d, p = t.shuffle_and_remask(wdeck, rho_b, perm, ps)
        assert d == want_deck and p == want_proof
        assert t.verify_shuffle(wdeck, d, p) == 0


def test_field_arithmetic_matches_bigint_reference():
    """field.hpp (9x29 sparse / signed-sparse / dense lazy limbs, 8x32, 12x32) against an independent schoolbook big-integer reference:
    products, squares, fused a b - c d, chains of lazily reduced sums and differences, zero tests, packed round trips"""
    exe = os.path.join(ROOT, "tests", "cpp", "_field_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-include", os.path.join(ROOT, "tools", "hostemu", "rt.hpp"),
                           "-I" + os.path.join(ROOT, "tools", "hostemu"), "-I" + os.path.join(ROOT, "mental-poker_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "field_check.cpp"), "-o", exe])
    out = subprocess.run([exe, "6000"], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert out.count(" 0 mismatches") == 10 and "fused products with carry-free operands, 0 mismatches" in out, out


def test_emulated_batch_and_status(emu, coracle):
    cv, m, n = "stark", 2, 3
    eng = emu(cv)
    ins = [coracle.gen_inputs(cv, m, n, 300 + b) for b in range(3)]
    t = eng.table(m, n, ins[0]["params"], ins[0]["pk"])
    perms = [v for g in ins for v in g["perm"]]
    perms[6:12] = [0, 0, 1, 2, 3, 4]          # proof 1: invalid permutation
    d, p, st = t.shuffle_and_remask_batch(b"".join(g["deck"] for g in ins), b"".join(g["rho"] for g in ins), perms,
                                          b"".join(g["prover_seed"] for g in ins))
    assert st == [0, -2, 0]
    ps = t.proof_bytes
    for b in (0, 2):
        g = ins[b]
        ed, ep = coracle.shuffle_and_remask(cv, m, n, ins[0]["params"], ins[0]["pk"], g["deck"], g["rho"], g["perm"], g["prover_seed"])
        assert d[b * 6 * 128:(b + 1) * 6 * 128] == ed and p[b * ps:(b + 1) * ps] == ep
    st = t.verify_shuffle_batch(b"".join(g["deck"] for g in ins), d, p)
    assert st[0] == 0 and st[2] == 0 and st[1] != 0
    t.close()


def test_sigma_oracle_roundtrip():
    """SURVEY 8f1 oracle: keygen, Schnorr, Chaum-Pedersen mask / remask / reveal, unmask (Python big-int spec)"""
    import mp_oracle as po
    cv = po.STARK
    pp = po.setup(cv, 2, 3, po.ChaCha20Rng(bytes(range(32))))
    rng = po.ChaCha20Rng(b"\x01" * 32)
    players = [po.player_keygen(pp, rng) for _ in range(3)]
    infos = [b"p%d" % i for i in range(3)]
    proofs = [po.prove_key_ownership(pp, pk, sk, info, bytes([i]) * 32) for i, ((pk, sk), info) in enumerate(zip(players, infos))]
    agg = po.compute_aggregate_key(pp, [(pk, pr, info) for (pk, sk), pr, info in zip(players, proofs, infos)])
    with pytest.raises(po.VerifyError) as e:
        po.compute_aggregate_key(pp, [(players[0][0], proofs[1], infos[0])])
    assert str(e.value) == "Schnorr Identification"
    card = po.pt_mul(cv, 4242, cv.G)
    masked, mpf = po.mask(pp, agg, card, 777, bytes(32))
    assert po.verify_mask(pp, agg, card, masked, mpf) and not po.verify_mask(pp, agg, po.pt_mul(cv, 2, card), masked, mpf)
    rem, rpf = po.remask_with_proof(pp, agg, masked, 999, b"\x02" * 32)
    assert po.verify_remask(pp, agg, masked, rem, rpf)
    toks = []
    for i, (pk, sk) in enumerate(players):
        t, pf = po.compute_reveal_token(pp, sk, pk, rem, bytes([9 + i]) * 32)
        assert po.verify_reveal(pp, pk, t, rem, pf)
        toks.append((t, pf, pk))
    assert po.unmask(pp, toks, rem) == card
    toks[1] = (po.pt_mul(cv, 5, cv.G), toks[1][1], toks[1][2])
    with pytest.raises(po.VerifyError) as e:
        po.unmask(pp, toks, rem)
    assert str(e.value) == "Chaum-Pedersen"
    assert po.sigma_proof_from_bytes(po.sigma_proof_bytes(mpf), 2) == mpf


def test_emulated_sigma_engine_matches_oracle(emu):
    import hashlib
    import mp_oracle as po
    cv = po.STARK
    eng = emu("stark")
    for k in (0, 1, 63, 64, 65, 200):
        assert eng.blake2s(bytes(range(k))) == hashlib.blake2s(bytes(range(k))).digest()
    pp, pk0, _, _, _, _ = po.gen_inputs(cv, 2, 3, 5)
    t = eng.table(2, 3, po.params_to_bytes(pp), po.pt_wire(pk0))
    rng = po.ChaCha20Rng(bytes(range(32)))
    bases = pubs = wit = seeds = exp = b""
    for i in range(3):
        x = po.fr_rand(cv, rng)
        g, h = po.pt_mul(cv, po.fr_rand(cv, rng), cv.G), po.pt_mul(cv, po.fr_rand(cv, rng), cv.G)
        a = [po.pt_mul(cv, x, g), po.pt_mul(cv, x, h)]
        seed = bytes([i + 1]) * 32
        bases += po.pt_wire(g) + po.pt_wire(h)
        pubs += po.pt_wire(a[0]) + po.pt_wire(a[1])
        wit += po.fe_bytes(x)
        seeds += seed
        exp += po.sigma_proof_bytes(po.sigma_prove(cv, [g, h], a, x, po.MASKING_RNG_SEED, seed))
    fsi = eng.blake2s(po.MASKING_RNG_SEED) * 3
    got, st = t.sigma_prove_batch(2, bases, pubs, wit, fsi, seeds)
    assert got == exp and st == [0, 0, 0]
    assert t.sigma_verify_batch(2, bases, pubs, got, fsi) == [0, 0, 0]
    bad = bytearray(got)
    bad[-1] ^= 1
    assert t.sigma_verify_batch(2, bases, pubs, bytes(bad), fsi) == [0, 0, 6]
    assert eng.check_name(5) == "Schnorr Identification" and eng.check_name(6) == "Chaum-Pedersen"
    t.close()


@pytest.mark.parametrize("cvn", ["stark", "bn254", "secp256k1", "bls12_377"])
def test_setup_has_no_trapdoor_and_matches_oracle(emu, cvn):
    """mp_setup ("setup v2"): n + 3 independent `C::rand` points (x from the stream, lifted, cofactor-cleared) -- the same bytes as
    the oracle, on the curve, in the prime-order subgroup, and NOT the round-1 construction k * G_std"""
    import mp_oracle as po
    cv = po.CURVES[cvn]
    seed = bytes(range(7, 39))
    raw = emu(cvn).setup(2, 4, seed)
    with po.curve_ctx(cv):
        pp = po.setup(cv, 2, 4, po.ChaCha20Rng(seed))
        pts = [pp.G] + pp.ck + [pp.H, pp.gen]
        assert raw == b"".join(po.pt_wire(P) for P in pts)
        for P in pts:
            assert cv.is_on_curve(P) and po.pt_mul_raw(cv, cv.q, P) is None and po.pt_mul_raw(cv, cv.q - 1, P) == po.pt_neg(cv, P)
        old = po.pt_mul(cv, po.fr_rand(cv, po.ChaCha20Rng(seed)), cv.G)
        assert pts[0] != old


def test_single_shot_entry_points_validate_lengths(emu, native):
    """peer-supplied buffers of the wrong size are refused in the binding (never handed to the C ABI as short buffers)"""
    g = load_json(os.path.join(GOLDEN, "shuffle_stark_m2_n3_s1.json"))
    eng = emu("stark")
    t = eng.table(2, 3, bytes.fromhex(g["params"]), bytes.fromhex(g["pk"]))
    deck, shuf, proof = bytes.fromhex(g["deck"]), bytes.fromhex(g["shuffled"]), bytes.fromhex(g["proof"])
    assert t.verify_shuffle(deck, shuf, proof) == 0
    for bad in ((deck[:-128], shuf, proof), (deck, shuf[:-1], proof), (deck, shuf, proof[:-32]), (deck + deck, shuf, proof)):
        with pytest.raises(native.NativeError):
            t.verify_shuffle(*bad)
    rho, perm, seed = bytes.fromhex(g["rho"]), g["perm"], bytes.fromhex(g["prover_seed"])
    for bad in ((deck[:-64], rho, perm, seed), (deck, rho[:-32], perm, seed), (deck, rho, perm[:-1], seed), (deck, rho, perm, seed[:16])):
        with pytest.raises(native.NativeError):
            t.shuffle_and_remask(*bad)
    with pytest.raises(native.NativeError):
        t.verify_shuffle_batch(deck, shuf[:-1], proof)


def test_points_outside_the_prime_order_subgroup_are_rejected(emu, native):
    """BLS12-377 G1 has a cofactor: a point ON the curve but outside the prime-order subgroup must be refused as a key, as a
    deck component (status MP_ERR_BAD_ENCODING) and by the canonical decoder -- what ark-ec's validating deserialiser does"""
    import mp_oracle as po
    cv = po.CURVES["bls12_377"]
    g = load_json(os.path.join(GOLDEN, "shuffle_bls12_377_m2_n3_s13.json"))
    with po.curve_ctx(cv):
        x = 5
        while True:                      # a curve point that was NOT multiplied by the cofactor ...
            y = po.fq_sqrt(cv, (x ** 3 + cv.b) % cv.p)
            if y is not None and po.pt_mul_raw(cv, cv.q, (x, y)) is not None:
                break
            x += 1
        low = po.pt_mul_raw(cv, cv.q, (x, y))      # ... times q: non-trivial, of order dividing the cofactor
        assert cv.is_on_curve(low) and po.pt_mul_raw(cv, po.COFACTOR["bls12_377"], low) is None
        bad = po.pt_wire(low)
        # the device's test is phi(P) = -[u^2]P (kernels_msm.hpp): also the untouched curve point and a subgroup point plus `low`
        others = [po.pt_wire((x, y)), po.pt_wire(po.pt_add(cv, low, cv.G))]
    eng = emu("bls12_377")
    params, pk = bytes.fromhex(g["params"]), bytes.fromhex(g["pk"])
    with pytest.raises(native.NativeError):
        eng.table(2, 3, params, bad)
    with pytest.raises(native.NativeError):
        eng.table(2, 3, bad + params[96:], pk)
    t = eng.table(2, 3, params, pk)
    deck, shuf, proof = bytes.fromhex(g["deck"]), bytes.fromhex(g["shuffled"]), bytes.fromhex(g["proof"])
    assert t.verify_shuffle_batch(deck, shuf, proof) == [0]
    tampered = bad + deck[96:]
    assert t.verify_shuffle_batch(tampered, shuf, proof) == [native._native.MP_ERR_BAD_ENCODING]
    for o in others:
        assert t.verify_shuffle_batch(o + deck[96:], shuf, proof) == [native._native.MP_ERR_BAD_ENCODING]
    _, _, st = t.shuffle_and_remask_batch(tampered, bytes.fromhex(g["rho"]), g["perm"], bytes.fromhex(g["prover_seed"]))
    assert st == [native._native.MP_ERR_BAD_ENCODING]
    t.set_subgroup_check(False)                     # the caller vouches for its points: the same input is now merely a wrong statement
    assert t.verify_shuffle_batch(tampered, shuf, proof)[0] > 0
    comp = native.canonical.point_compress("bls12_377", bad)
    with pytest.raises(native.canonical.SerializationError):
        native.canonical.point_decompress("bls12_377", comp)
    assert native.canonical.point_decompress("bls12_377", native.canonical.point_compress("bls12_377", pk)) == pk


@pytest.mark.parametrize("name", ["shuffle_stark_m2_n26_s7.json", "shuffle_stark_m3_n4_s11.json", "shuffle_secp256k1_m3_n3_s5.json",
                                  "shuffle_bls12_377_m2_n3_s13.json"])
def test_bucket_method_kernel_under_emulation_matches_golden(emu, name):
    """mp_set_bucket_min forces every variable-base MSM of >= 4 terms through the wave-cooperative bucket kernel
    (kernels_bucket.hpp: LDS-staged digits, counting sort, two buckets per lane, wave-wide reduction): same bytes, same verdicts"""
    g = load_json(os.path.join(GOLDEN, name))
    eng = emu(g["curve"])
    m, n = g["m"], g["n"]
    t = eng.table(m, n, bytes.fromhex(g["params"]), bytes.fromhex(g["pk"]))
    t.set_bucket_min(4)
    for latency_batch, merged in (((0, True),) if n > 8 else ((0, True), (0, False), (8192, True))):
        t.set_latency_batch(latency_batch)
        t.set_merged_verify(merged)
        eng.profile_enable(True)
        deck, proof = t.shuffle_and_remask(bytes.fromhex(g["deck"]), bytes.fromhex(g["rho"]), g["perm"], bytes.fromhex(g["prover_seed"]))
        assert deck.hex() == g["shuffled"] and proof.hex() == g["proof"]
        assert t.verify_shuffle(bytes.fromhex(g["deck"]), deck, proof) == 0
        bad = bytearray(proof)
        bad[-1] ^= 1
        assert t.verify_shuffle(bytes.fromhex(g["deck"]), deck, bytes(bad)) != 0
        rep = eng.profile_report()
        eng.profile_enable(False)
        # (a single proof runs the fold and the remaining Straus jobs on four lanes per group operation: kernels_quad.hpp)
        assert rep["k_bucket_msm"][0] >= 2 and ("k_bucket_fold" in rep or "k_bucket_fold_q" in rep) and "k_bucket_recode" in rep
        if m == 2 and latency_batch == 0 and merged:
            nvar = rep.get("k_var_msm", (0,))[0] + rep.get("k_var_msm_q", (0,))[0]
            assert nvar < rep["k_bucket_msm"][0] + 3


def test_bucket_msm_edge_scalars_under_emulation(emu, coracle):
    """0, 1, q - 1, +-128 boundaries of the signed 8-bit recoding, repeated and opposite points, the point at infinity"""
    import random
    cvn = "stark"
    q = 0x0800000000000010ffffffffffffffffb781126dcae7b2321e66a241adc64d2f
    eng = emu(cvn)
    gi = coracle.gen_inputs(cvn, 2, 3, 5)
    t = eng.table(2, 3, gi["params"], gi["pk"])
    t.set_bucket_min(16)
    random.seed(4)
    K = 150
    pts = bytearray(eng.setup(2, K - 3, bytes([9] * 32)))
    pts[64 * 7:64 * 8] = pts[64 * 6:64 * 7]            # the same point twice
    pts[64 * 9:64 * 10] = bytes(64)                    # infinity
    sc = [random.randrange(q) for _ in range(K)]
    sc[0:12] = [0, 1, q - 1, 128, 127, 129, 2 ** 248, 2 ** 251, 255, 256 * 128, q - 128, q - 129]
    sc[6], sc[7] = 5, q - 5                            # P and -P cancel inside one bucket
    scb = b"".join(s.to_bytes(32, "little") for s in sc)
    assert t.msm(1, K, scb, bytes(pts)) == coracle.msm(cvn, scb, bytes(pts))
    same = b"".join((77).to_bytes(32, "little") for _ in range(K))      # every term in ONE bucket of one window
    assert t.msm(1, K, same, bytes(pts)) == coracle.msm(cvn, same, bytes(pts))
    zero = bytes(32 * K)
    assert t.msm(1, K, zero, bytes(pts)) == bytes(64)


@pytest.mark.parametrize("name", ["shuffle_stark_m3_n4_s11.json", "shuffle_stark_m4_n13_s9.json", "shuffle_secp256k1_m3_n3_s5.json"])
def test_toom_cook_and_karatsuba_give_the_same_proof(emu, name):
    """3 <= m <= 8: the diagonals of the multi-exponentiation argument through Toom-Cook (2m products, evaluation at small integer
    points, interpolation over Fr) and through recursive Karatsuba are the same group elements -- identical proof bytes"""
    g = load_json(os.path.join(GOLDEN, name))
    eng = emu(g["curve"])
    m, n = g["m"], g["n"]
    t = eng.table(m, n, bytes.fromhex(g["params"]), bytes.fromhex(g["pk"]))
    args = (bytes.fromhex(g["deck"]), bytes.fromhex(g["rho"]), g["perm"], bytes.fromhex(g["prover_seed"]))
    for on in (True, False):
        t.set_toom_cook(on)
        for latency_batch in (0, 8192):
            t.set_latency_batch(latency_batch)
            eng.profile_enable(True)
            deck, proof = t.shuffle_and_remask(*args)
            rep = eng.profile_report()
            eng.profile_enable(False)
            assert deck.hex() == g["shuffled"] and proof.hex() == g["proof"]
            uses_toom = on and latency_batch == 0         # the small-batch plans keep Karatsuba (fewer dependent stages)
            assert ("k_toom_points" in rep) == uses_toom and ("k_lin_comb" in rep) == uses_toom
            assert t.verify_shuffle(args[0], deck, proof) == 0


@pytest.mark.parametrize("cvn,m,n,L,T,keyed", [("stark", 2, 3, 4, 3, False), ("stark", 3, 2, 3, 2, True), ("secp256k1", 2, 3, 2, 2, False)])
def test_chain_verification_under_emulation(emu, coracle, cvn, m, n, L, T, keyed):
    """mp_verify_shuffle_chain: T tables x L dependent shuffles verified as one equation per table (every inner deck a base once);
    honest chains pass, a chain with one bad link gets exactly the per-link verifier's status words"""
    eng = emu(cvn)
    N, pb = m * n, eng.point_bytes
    g0 = coracle.gen_inputs(cvn, m, n, 50)
    params = g0["params"]
    keys_t = [coracle.gen_inputs(cvn, m, n, 60 + t)["pk"] for t in range(T)] if keyed else [g0["pk"]] * T
    table = eng.table(m, n, params, None if keyed else g0["pk"])
    chain = [[coracle.gen_inputs(cvn, m, n, 70 + t)["deck"] for t in range(T)]]     # deck 0 of every table
    proofs = []
    for j in range(L):
        nxt, prf = [], []
        for t in range(T):
            gi = coracle.gen_inputs(cvn, m, n, 100 + 10 * j + t)
            d, p = coracle.shuffle_and_remask(cvn, m, n, params, keys_t[t], chain[j][t], gi["rho"], gi["perm"], gi["prover_seed"])
            nxt.append(d)
            prf.append(p)
        chain.append(nxt)
        proofs.append(prf)
    decks = b"".join(b"".join(row) for row in chain)
    pf = b"".join(b"".join(row) for row in proofs)
    keys = b"".join(keys_t[t] for j in range(L) for t in range(T)) if keyed else None
    eng.profile_enable(True)
    assert table.verify_shuffle_chain(T, L, decks, pf, keys) == [0] * (L * T)
    rep = eng.profile_report()
    table.set_chain_max_links(2)                     # long chains are cut into sub-chains: same verdicts
    try:
        assert table.verify_shuffle_chain(T, L, decks, pf, keys) == [0] * (L * T)
        eng.profile_report()
    finally:
        table.set_chain_max_links(0)
    eng.profile_enable(False)
    assert rep["k_chain_scalars"][0] == 1 and rep["k_bucket_msm"][0] == 1 and "k_var_msm" not in rep and "k_table" not in rep
    # break link 1 of table 0 (swap in another table's proof) and one deck point of the last link of table T-1
    bad = [row[:] for row in proofs]
    bad[1 % L][0] = proofs[1 % L][(0 + 1) % T]
    st = table.verify_shuffle_chain(T, L, decks, b"".join(b"".join(row) for row in bad), keys)
    exp = []
    for j in range(L):
        ks = b"".join(keys_t) if keyed else None
        row = (table.verify_shuffle_batch_keys(ks, b"".join(chain[j]), b"".join(chain[j + 1]), b"".join(bad[j])) if keyed else
               table.verify_shuffle_batch(b"".join(chain[j]), b"".join(chain[j + 1]), b"".join(bad[j])))
        exp += row
    assert st == exp and st[(1 % L) * T + 0] > 0 and sum(1 for v in st if v) == 1
    # errors that cancel between two links under equal weights are caught: the weights depend on every proof of the chain
    tam = bytearray(decks)
    tam[(1 * T + 0) * N * 2 * pb] ^= 1                 # first byte of deck 1 of table 0: not a curve point any more (or another one)
    st2 = table.verify_shuffle_chain(T, L, bytes(tam), pf, keys)
    assert st2[0 * T + 0] != 0 and st2[1 * T + 0] != 0 and all(v == 0 for i, v in enumerate(st2) if i % T != 0)
    if keyed and L >= 2:
        # the links of a table need not name the same key.  (i) the verifier is told another key for link 1 of table 0 than the one the
        # link was proven under: exactly that link fails, as it does link by link; (ii) link 1 of table 0 really IS proven under
        # another key and the verifier is told so: every link passes, although the chain equation's single key term does not apply
        other = coracle.gen_inputs(cvn, m, n, 99)["pk"]
        klist = [[keys_t[t] for t in range(T)] for j in range(L)]
        klist[1][0] = other
        mixed = b"".join(b"".join(row) for row in klist)
        st3 = table.verify_shuffle_chain(T, L, decks, pf, mixed)
        exp3 = []
        for j in range(L):
            exp3 += table.verify_shuffle_batch_keys(b"".join(klist[j]), b"".join(chain[j]), b"".join(chain[j + 1]), b"".join(proofs[j]))
        assert st3 == exp3 and st3[1 * T + 0] != 0 and sum(1 for v in st3 if v) == 1
        ch2 = [row[:] for row in chain]
        pf2 = [row[:] for row in proofs]
        for j in range(1, L):                          # table 0 from link 1 on: link 1 under `other`, the rest under the table's key again
            gi = coracle.gen_inputs(cvn, m, n, 123 + j)
            ch2[j + 1][0], pf2[j][0] = coracle.shuffle_and_remask(cvn, m, n, params, other if j == 1 else keys_t[0], ch2[j][0], gi["rho"],
                                                                   gi["perm"], gi["prover_seed"])
        st4 = table.verify_shuffle_chain(T, L, b"".join(b"".join(r) for r in ch2), b"".join(b"".join(r) for r in pf2), mixed)
        assert st4 == [0] * (L * T)
        # (iii) a cheating prover: link 1 of table 0 is made under the table's key but its transcript absorbs `other`, and the
        # verifier is told `other` for that link.  Link by link the multi-exponentiation check fails (the algebra runs under
        # `other`); a chain equation that took link 0's key for every link would accept it.
        import mp_oracle as po
        cv = po.CURVES[cvn]
        with po.curve_ctx(cv):
            w = po.point_bytes()
            pts = [po.pt_from_wire(params[i:i + w]) for i in range(0, len(params), w)]
            pp = po.Params(cv, m, n, pts[0], pts[1:1 + n], pts[1 + n], pts[2 + n])
            gi = coracle.gen_inputs(cvn, m, n, 777)
            rho = [int.from_bytes(gi["rho"][i:i + 32], "little") for i in range(0, 32 * N, 32)]
            honest_statement = po.statement_bytes
            po.statement_bytes = lambda pp_, pk_, d_, s_: honest_statement(pp_, po.pt_from_wire(other), d_, s_)
            try:
                sh, prf = po.shuffle_and_remask(pp, po.pt_from_wire(keys_t[0]), po.deck_from_bytes(chain[1][0]), rho, list(gi["perm"]), gi["prover_seed"])
            finally:
                po.statement_bytes = honest_statement
            ch3 = [row[:] for row in chain]
            pf3 = [row[:] for row in proofs]
            ch3[2][0], pf3[1][0] = po.deck_to_bytes(sh), po.proof_to_bytes(prf)
        for j in range(2, L):
            gj = coracle.gen_inputs(cvn, m, n, 800 + j)
            ch3[j + 1][0], pf3[j][0] = coracle.shuffle_and_remask(cvn, m, n, params, keys_t[0], ch3[j][0], gj["rho"], gj["perm"], gj["prover_seed"])
        st5 = table.verify_shuffle_chain(T, L, b"".join(b"".join(r) for r in ch3), b"".join(b"".join(r) for r in pf3), mixed)
        exp5 = []
        for j in range(L):
            exp5 += table.verify_shuffle_batch_keys(b"".join(klist[j]), b"".join(ch3[j]), b"".join(ch3[j + 1]), b"".join(pf3[j]))
        assert st5 == exp5 and st5[1 * T + 0] != 0 and sum(1 for v in st5 if v) == 1
