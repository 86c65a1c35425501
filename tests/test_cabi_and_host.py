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
    # finest split, latency plan, throughput plan (large sub-jobs, Toom-Cook for m = 2); negative: split -latency_batch forced (wide: 16
    # window lanes per variable-base sub-job, small: 8; medium: 4)
    for latency_batch, lanes in ((8192, 0), (1, 0), (0, 0), (8192, 1), (0, 1), (-4, 0), (-5, 1), (-2, 1), (-5, 0)):
        t.set_work_split(-latency_batch if latency_batch < 0 else -1)
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
        t.set_work_split(6)
    with pytest.raises(native.NativeError):
        t.set_plan_params(2, 4, 16, 65, 8, 1)          # more than 64 bases per table lane
    with pytest.raises(native.NativeError):
        t.set_plan_thresholds(10, 5, 20, 30, 40)       # not ascending
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
    # a call of three chunks and more ramps up and down (C/8, 3C/8, C ..., 3C/8, C/8): 27 proofs in chunks of 8 = 1 3 8 8 3 1 + ragged 3
    ins2 = [coracle.gen_inputs(cv, m, n, 900 + b % 3) for b in range(27)]
    args2 = (b"".join(g["deck"] for g in ins2), b"".join(g["rho"] for g in ins2), [v for g in ins2 for v in g["perm"]],
             b"".join(g["prover_seed"] for g in ins2))
    t.set_io_chunk(0)
    ref2 = t.shuffle_and_remask_batch(*args2)
    t.set_io_chunk(8)
    assert t.shuffle_and_remask_batch(*args2) == ref2
    assert t.verify_shuffle_batch(args2[0], ref2[0], ref2[1]) == [0] * 27
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
    (kernels_bucket.hpp: LDS histogram, counting sort, two buckets per lane dealt by rank, wave-wide reduction): same bytes, same verdicts"""
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
    if name == "shuffle_stark_m3_n4_s11.json":
        # round 6: the same proof with every one of those MSMs on the SPLIT pipeline (12-bit windows: k_bucket_sort / _acc / _reduce /
        # _final, several bucket jobs of different lengths per phase) -- same bytes, same verdicts
        t.set_latency_batch(0)
        t.set_merged_verify(True)
        t.set_bucket_bits(12)
        eng.profile_enable(True)
        deck, proof = t.shuffle_and_remask(bytes.fromhex(g["deck"]), bytes.fromhex(g["rho"]), g["perm"], bytes.fromhex(g["prover_seed"]))
        assert deck.hex() == g["shuffled"] and proof.hex() == g["proof"]
        assert t.verify_shuffle(bytes.fromhex(g["deck"]), deck, proof) == 0
        bad = bytearray(proof)
        bad[-1] ^= 1
        assert t.verify_shuffle(bytes.fromhex(g["deck"]), deck, bytes(bad)) != 0
        rep = eng.profile_report()
        eng.profile_enable(False)
        assert rep["k_bucket_acc"][0] >= 2 and "k_bucket_final" in rep and "k_bucket_msm" not in rep
        t.set_bucket_bits(0)


def test_bucket_msm_edge_scalars_under_emulation(emu, coracle):
    """0, 1, q - 1, +-128 boundaries of the signed 8-bit recoding, repeated and opposite points, the point at infinity; the same through
    9- and 10-bit windows (mp_set_bucket_bits)"""
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
    want = coracle.msm(cvn, scb, bytes(pts))
    same = b"".join((77).to_bytes(32, "little") for _ in range(K))      # every term in ONE bucket of one window
    want_same = coracle.msm(cvn, same, bytes(pts))
    zero = bytes(32 * K)
    for bits in (0, 8, 9, 10, 11, 12, 13, 14):             # window widths of the bucket method (0: by size = 8 here); 2, 4, 8, 16 buckets per lane of a wave, the split pipeline from 12 bits on (round 6)
        t.set_bucket_bits(bits)
        assert t.msm(1, K, scb, bytes(pts)) == want, bits
        assert t.msm(1, K, same, bytes(pts)) == want_same, bits
        assert t.msm(1, K, zero, bytes(pts)) == bytes(64), bits
    # the boundaries of the 9- and 10-bit signed recodings, and more MSMs than the emulator has persistent waves (items wrap around)
    t.set_bucket_bits(10)
    sc[12:20] = [511, 512, 513, 2 ** 250 + 511, (1 << 252) - 1 - (q - (1 << 251)) % 7, q - 512, q - 513, 256]
    nm = 3
    many = b"".join(((s * (i + 1)) % q).to_bytes(32, "little") for i in range(nm) for s in sc)
    got = t.msm(nm, K, many, bytes(pts) * nm)
    for i in range(nm):
        assert got[64 * i:64 * (i + 1)] == coracle.msm(cvn, many[32 * K * i:32 * K * (i + 1)], bytes(pts)), i
    # ... of the 12- and 13-bit ones (256 lanes per window)
    for bits, edge in ((12, 2048), (13, 4096), (14, 8192)):
        t.set_bucket_bits(bits)
        sc[12:20] = [edge - 1, edge, edge + 1, 2 ** 250 + edge - 1, (1 << 252) - 1 - (q - (1 << 251)) % 7, q - edge, q - edge - 1, 2 * edge]
        many = b"".join(((s * (i + 1)) % q).to_bytes(32, "little") for i in range(nm) for s in sc)
        got = t.msm(nm, K, many, bytes(pts) * nm)
        for i in range(nm):
            assert got[64 * i:64 * (i + 1)] == coracle.msm(cvn, many[32 * K * i:32 * K * (i + 1)], bytes(pts)), (bits, i)
    with pytest.raises(Exception):
        t.set_bucket_bits(15)
    # the width the engine picks by size: 9 bits from 6 000 terms, 10 from 12 000 (the points repeat: the oracle folds the scalars)
    base = bytes(pts)
    # (... 12 and 13 bits on the split pipeline with three sorted runs per bucket: 24 576 terms per run; the last scalars all equal --
    # they do not make the window crowded, the 13-bit top window of a 252-bit scalar is: both modes in one MSM)
    for big, bits in ((6100, 0), (12100, 0), (60000, 12), (52000, 13), (52000, 14)):
        t.set_bucket_bits(bits)
        scs = [random.randrange(q) for _ in range(big)]
        if bits:
            scs[-40:] = [77] * 40
        folded = [0] * K
        for i, s_ in enumerate(scs):
            folded[i % K] = (folded[i % K] + s_) % q
        got = t.msm(1, big, b"".join(s_.to_bytes(32, "little") for s_ in scs), (base * (big // K + 1))[:64 * big])
        assert got == coracle.msm(cvn, b"".join(s_.to_bytes(32, "little") for s_ in folded), base), big
    t.set_bucket_bits(0)


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


@pytest.mark.parametrize("cvn,m,n,L,T,keyed,group", [("stark", 2, 3, 4, 3, False, 0), ("stark", 3, 2, 3, 2, True, 0), ("secp256k1", 2, 3, 2, 2, False, 0),
                                                     ("stark", 2, 3, 3, 4, True, 2), ("stark", 2, 3, 4, 3, False, 3)])
def test_chain_verification_under_emulation(emu, coracle, cvn, m, n, L, T, keyed, group):
    """mp_verify_shuffle_chain: T tables x L dependent shuffles verified as one equation per table (every inner deck a base once);
    honest chains pass, a chain with one bad link gets exactly the per-link verifier's status words; the same with the chains of
    `group` tables in one equation (mp_set_chain_group)"""
    from chain_cases import run_chain_cases
    run_chain_cases(emu(cvn), coracle, cvn, m, n, L, T, keyed, group)


def test_emulated_window_lanes_and_pipelined_verification(emu, coracle):
    """round 4: (a) the windows of a variable-base sub-job dealt to k lanes + one fold per MSM give the same bytes for every k;
    (b) pipelined verify calls (mp_set_pipeline: second lane, verdict looked at `depth` calls later) give the same status words as
    the waiting ones, including a batch whose per-equation pass is deferred, and mp_sync completes everything outstanding"""
    import ctypes
    cv, m, n, B = "stark", 2, 3, 3
    eng = emu(cv)
    ins = [coracle.gen_inputs(cv, m, n, 4100 + b) for b in range(B)]
    g0 = ins[0]
    t = eng.table(m, n, g0["params"], g0["pk"])
    args = (b"".join(g["deck"] for g in ins), b"".join(g["rho"] for g in ins), [v for g in ins for v in g["perm"]],
            b"".join(g["prover_seed"] for g in ins))
    ref = t.shuffle_and_remask_batch(*args)
    for g, d, p in zip(ins, [ref[0][i * len(g0["deck"]):(i + 1) * len(g0["deck"])] for i in range(B)],
                       [ref[1][i * t.proof_bytes:(i + 1) * t.proof_bytes] for i in range(B)]):
        assert (d, p) == coracle.shuffle_and_remask(cv, m, n, g0["params"], g0["pk"], g["deck"], g["rho"], g["perm"], g["prover_seed"])
    for split, prm in ((2, (4, 16, 16, 32, 3)), (0, (8, 64, 64, 64, 16)), (1, (1, 2, 2, 4, 5)), (5, (1, 1, 2, 4, 2))):
        t.set_plan_params(split, *prm)
        t.set_work_split(split)
        for lanes in (1, 0):
            t.set_group_lanes(lanes)
            assert t.shuffle_and_remask_batch(*args) == ref
            assert t.verify_shuffle_batch(args[0], ref[0], ref[1]) == [0] * B
    t.set_group_lanes(0)
    # ---- pipelined verify calls through the device-pointer entry points (the emulator's "device" memory is host memory)
    t.set_work_split(2)                                  # a split that screens with the merged equation
    dsz, psz = len(g0["deck"]), t.proof_bytes
    buf = lambda b: (ctypes.c_uint8 * len(b)).from_buffer_copy(b)
    decks, good_d, good_p = buf(args[0]), buf(ref[0]), buf(ref[1])
    bad_p = bytearray(ref[1])
    bad_p[psz + 40] ^= 1                                 # proof 1: one bit of c_A
    bad_p = buf(bytes(bad_p))
    rot_d = buf(ref[0][dsz:] + ref[0][:dsz])             # every proof against its neighbour's deck
    expect = {}
    for name, d, p in (("good", good_d, good_p), ("badproof", good_d, bad_p), ("rotated", rot_d, good_p)):
        expect[name] = t.verify_shuffle_batch(args[0], bytes(d), bytes(p))
    assert expect["good"] == [0] * B and expect["badproof"][0] == 0 and expect["badproof"][1] != 0 and all(expect["rotated"])
    addr = ctypes.addressof
    for depth in (1, 2):
        t.set_pipeline(depth)
        sts = [(ctypes.c_int32 * B)(*([77] * B)) for _ in range(4)]
        order = ["good", "badproof", "rotated", "good"]
        srcs = {"good": (good_d, good_p), "badproof": (good_d, bad_p), "rotated": (rot_d, good_p)}
        for st, name in zip(sts, order):
            d, p = srcs[name]
            t.verify_shuffle_batch_dev(B, addr(decks), addr(d), addr(p), addr(st))
        eng.sync()                                       # deferred per-equation passes run here at the latest
        for st, name in zip(sts, order):
            assert list(st) == expect[name], (depth, name, list(st))
        # the host-buffer entry points are not pipelined and may be mixed in
        assert t.verify_shuffle_batch(args[0], ref[0], ref[1]) == [0] * B
    t.set_pipeline(0)
    with pytest.raises(Exception):
        t.set_pipeline(9)
    t.close()


class _HostMem:
    """the emulator's "device" memory is host memory: ctypes buffers stand in for HBM"""

    def put(self, b):
        import ctypes
        buf = (ctypes.c_uint8 * max(len(b), 1)).from_buffer_copy(b if b else b"\0")
        return buf, ctypes.addressof(buf)

    def new(self, nbytes):
        import ctypes
        buf = (ctypes.c_uint8 * max(nbytes, 1))()
        return buf, ctypes.addressof(buf)

    def get(self, h, nbytes):
        return bytes(h)[:nbytes]


@pytest.mark.parametrize("name", ["shuffle_stark_m2_n3_s1.json", "shuffle_bn254_m2_n4_s3.json", "shuffle_secp256k1_m3_n3_s5.json",
                                  "shuffle_bls12_377_m2_n3_s13.json"])
def test_emulated_device_decompression_matches_oracle(emu, name):
    """mp_deck_deserialize_dev / mp_points_deserialize_dev (kernel body under emulation): arkworks-compressed points -> wire v1 with the
    windowed square root, against the oracle's encoder and decoder -- golden decks, random points, every malformed case"""
    from decompress_cases import run_decompress_cases
    g = load_json(os.path.join(GOLDEN, name))
    run_decompress_cases(emu(g["curve"]), _HostMem(), g["curve"], g, n_random=6 if g["curve"] != "bls12_377" else 3)


@pytest.mark.parametrize("cv,keyed", [("stark", False), ("stark", True), ("secp256k1", False)])
def test_emulated_group_verification(emu, coracle, cv, keyed):
    """round 4, group verification: the screening pass of a batch as one equation per group of proofs on the bucket kernel -- same status
    words as the per-proof screen for honest batches, one bad proof, a bad input encoding, every proof bad; with and without pipelining"""
    import ctypes
    m, n, B = 2, 3, 6
    eng = emu(cv)
    ins = [coracle.gen_inputs(cv, m, n, 6100 + b) for b in range(B)]
    g0 = ins[0]
    keys = b"".join(coracle.gen_inputs(cv, m, n, 6200 + b % 2)["pk"] for b in range(B))
    t = eng.table(m, n, g0["params"], None if keyed else g0["pk"])
    args = (b"".join(g["deck"] for g in ins), b"".join(g["rho"] for g in ins), [v for g in ins for v in g["perm"]],
            b"".join(g["prover_seed"] for g in ins))
    prove = (lambda: t.shuffle_and_remask_batch_keys(keys, *args)) if keyed else (lambda: t.shuffle_and_remask_batch(*args))
    verify = (lambda d, s_, p: t.verify_shuffle_batch_keys(keys, d, s_, p)) if keyed else (lambda d, s_, p: t.verify_shuffle_batch(d, s_, p))
    t.set_work_split(0)
    t.set_group_verify(0, 0)
    ref = prove()
    dsz, psz = len(g0["deck"]), t.proof_bytes
    bad_p = bytearray(ref[1])
    bad_p[5 * psz - 31] ^= 2                              # proof 4: one bit of its last response scalar
    bad_d = bytearray(ref[0])
    bad_d[1 * dsz + 5] ^= 1                                # deck 1: a coordinate that is not on the curve any more
    rot = ref[0][dsz:] + ref[0][:dsz]
    cases = {"good": (ref[0], ref[1]), "badproof": (ref[0], bytes(bad_p)), "badpoint": (bytes(bad_d), ref[1]), "rotated": (rot, ref[1])}
    want = {k: verify(args[0], d, p) for k, (d, p) in cases.items()}
    assert want["good"] == [0] * B and want["badproof"][4] > 0 and want["badpoint"][1] < 0 and all(v > 0 for v in want["rotated"])
    # (the group equation through 8-, 10-, 9- and 11-bit windows on the wave kernel, 12- and 13-bit ones on the split pipeline (k_bucket_sort / _acc / _reduce): mp_set_bucket_bits)
    for links, bits in ((3, 0), (2, 10), (6, 9), (3, 11), (3, 12), (6, 13), (3, 14)):
        t.set_bucket_bits(bits)
        t.set_group_verify(links * (4 * m * n + 11 * m + 8 + (1 if keyed else 0)), 0)
        eng.profile_enable(True)
        for k, (d, p) in cases.items():
            assert verify(args[0], d, p) == want[k], (links, k)
        rep = eng.profile_report()
        eng.profile_enable(False)
        assert "k_chain_scalars" in rep and ("k_bucket_acc" if bits >= 12 else "k_bucket_msm") in rep
    t.set_bucket_bits(0)
    if not keyed:                                          # pipelined: the group pass is the deferred screen
        buf = lambda b: (ctypes.c_uint8 * len(b)).from_buffer_copy(b)
        addr = ctypes.addressof
        decks = buf(args[0])
        t.set_group_verify(3 * (4 * m * n + 11 * m + 8), 0)
        t.set_pipeline(1)
        held = []
        for k, (d, p) in cases.items():
            st = (ctypes.c_int32 * B)(*([55] * B))
            db, pb_ = buf(d), buf(p)
            held.append((k, st, db, pb_))
            t.verify_shuffle_batch_dev(B, addr(decks), addr(db), addr(pb_), addr(st))
        eng.sync()
        for k, st, _, _ in held:
            assert list(st) == want[k], k
        t.set_pipeline(0)
    t.set_group_verify(30464, 6144)
    t.set_work_split(-1)
    t.close()


def test_emulated_rejection_costs_only_the_rejected(emu, coracle):
    """round 5: a failing screen re-verifies the proofs it could not clear and nobody else -- group -> sub-group -> per-equation pass,
    the per-proof screen -> per-equation pass, waiting and pipelined (depth 3, batch sizes that alternate, so the verify lane's arenas
    change their stride under calls in flight); status words as without any screen; the launch sizes say who was looked at"""
    import ctypes
    cv, m, n, B = "stark", 2, 3, 8
    per = 4 * m * n + 11 * m + 8
    eng = emu(cv)
    ins = [coracle.gen_inputs(cv, m, n, 7300 + b) for b in range(B)]
    g0 = ins[0]
    t = eng.table(m, n, g0["params"], g0["pk"])
    args = (b"".join(g["deck"] for g in ins), b"".join(g["rho"] for g in ins), [v for g in ins for v in g["perm"]],
            b"".join(g["prover_seed"] for g in ins))
    t.set_work_split(0)
    t.set_group_verify(0, 0)
    t.set_merged_verify(False)
    out_d, out_p, st0 = t.shuffle_and_remask_batch(*args)
    assert st0 == [0] * B
    dsz, psz = len(g0["deck"]), t.proof_bytes
    bad_p = bytearray(out_p)
    bad_p[6 * psz - 31] ^= 2                               # proof 5: one bit of its last response scalar
    bad_p = bytes(bad_p)
    want = t.verify_shuffle_batch(args[0], out_d, bad_p)   # equation by equation, no screen at all
    assert [w != 0 for w in want] == [b == 5 for b in range(B)]
    assert coracle.verify_shuffle(cv, m, n, g0["params"], g0["pk"], ins[5]["deck"], out_d[5 * dsz:6 * dsz], bad_p[5 * psz:6 * psz]) == want[5]
    t.set_merged_verify(True)

    def looked_at(fn):
        before = t.reverified_count()
        eng.profile_enable(True)
        got = fn()
        rep = eng.profile_report()
        eng.profile_enable(False)
        return got, t.reverified_count() - before, rep, dict(eng.last_profile_items)

    # (a) per-proof screen: only the marked proof takes the per-equation pass
    got, n_re, rep, items = looked_at(lambda: t.verify_shuffle_batch(args[0], out_d, bad_p))
    assert got == want and n_re == 1 and items["k_verdict"] == 1 and items["k_verdict_merged"] == B
    got, n_re, rep, items = looked_at(lambda: t.verify_shuffle_batch(args[0], out_d, out_p))
    assert got == [0] * B and n_re == 0 and "k_verdict" not in rep
    # (b) groups of 4: the failing group's 4 members, straight to the equations (too few suspects for sub-groups)
    t.set_group_verify(4 * per, 0)
    assert t.group_size(B) == 4
    got, n_re, rep, items = looked_at(lambda: t.verify_shuffle_batch(args[0], out_d, bad_p))
    assert got == want and n_re == 4 and items["k_verdict"] == 4 and items["k_gather_rows"] > 0 and rep["k_bucket_msm"][0] == 1
    # (c) ... through sub-groups of 2 first: the failing sub-group's 2 members reach the equations
    t.set_group_refine(2 * per, 1)
    got, n_re, rep, items = looked_at(lambda: t.verify_shuffle_batch(args[0], out_d, bad_p))
    assert got == want and n_re == 2 and items["k_verdict"] == 2 and rep["k_bucket_msm"][0] == 2 and rep["k_chain_verdict"][0] == 2
    # 16 proofs (the 8 twice) in 8 groups of 2: the (group, window) items of the bucket kernel are dealt to 8 partitions, one per XCD
    t.set_group_verify(2 * per, 0)
    t.set_group_refine(0, 0)
    assert t.group_size(16) == 2
    got, n_re, rep, items = looked_at(lambda: t.verify_shuffle_batch(args[0] * 2, out_d * 2, bad_p + out_p))
    assert got == want + [0] * B and n_re == 2 and rep["k_group_tile"][0] == 1
    t.set_group_verify(4 * per, 0)
    t.set_group_refine(2 * per, 1)
    # a bad point encoding in another group: that proof keeps its usage error, its group is looked at, the rest is not
    bad_d = bytearray(out_d)
    bad_d[2 * dsz + 5] ^= 1                                # (lane of (member j, group t) = 2 j + t: proof 2 is in group 0, proof 5 in group 1)
    want2 = list(want)
    want2[2] = t.verify_shuffle_batch(args[0][2 * dsz:3 * dsz], bytes(bad_d[2 * dsz:3 * dsz]), out_p[2 * psz:3 * psz])[0]
    assert want2[2] < 0
    got, n_re, rep, items = looked_at(lambda: t.verify_shuffle_batch(args[0], bytes(bad_d), bad_p))
    assert got == want2 and n_re == 4                      # both groups fail; sub-groups {2, 6} and {1, 5} reach the equations
    # (d) pipelined, depth 3, batch sizes 8 / 4 / 8 / 4 / 8 with the bad proof in the 8s
    buf = lambda b: (ctypes.c_uint8 * len(b)).from_buffer_copy(b)
    addr = ctypes.addressof
    decks, good_d, good_p, badp = buf(args[0]), buf(out_d), buf(out_p), buf(bad_p)
    t.set_pipeline(3)
    calls = []
    for k, (nb, pbuf) in enumerate(((8, badp), (4, good_p), (8, good_p), (4, badp), (8, badp))):
        st = (ctypes.c_int32 * nb)(*([66] * nb))
        calls.append((nb, pbuf is badp, st))
        t.verify_shuffle_batch_dev(nb, addr(decks), addr(good_d), addr(pbuf), addr(st))
    eng.sync()
    for nb, bad, st in calls:
        assert list(st) == (want[:nb] if bad else [0] * nb), (nb, bad, list(st))
    t.set_pipeline(0)
    # the two destroy calls in either order (the context goes with its last table)
    eng.close()
    t.close()
