"""-m gpu: parity of the HIP engine (through the C ABI) with the oracle.  Bit-exact everywhere: the path is
integer arithmetic, so proofs, decks and verdicts must be byte-identical to the oracle's.

Mirrors the reference's `test_shuffle` [REF barnett-smart-card-protocol/src/discrete_log_cards/tests.rs:175-227]
(accept honest / reject a wrong deck with "Hadamard Product (5.1)") and adds what the reference lacks:
fixed vectors, edge inputs, tamper cases per check, usage errors, batch-position independence and
full-size round trips."""
import copy
import os
import random

import pytest

from conftest import GOLDEN, golden_cases, load_json

import mp_oracle as po

pytestmark = pytest.mark.gpu


def _split(b, sz):
    return [b[i:i + sz] for i in range(0, len(b), sz)]


@pytest.fixture(scope="module")
def engines(mp):
    cache = {}

    def get(curve):
        if curve not in cache:
            cache[curve] = mp.DLCards(curve, device=0)
        return cache[curve]
    return get


def test_native_library_is_the_one_running(mp, engines):
    engines("stark")
    maps = open("/proc/self/maps").read()
    assert "libmpshuffle.so" in maps
    assert "libmpemu" not in maps


@pytest.mark.parametrize("path", golden_cases(), ids=os.path.basename)
def test_golden_vectors(mp, engines, path):
    g = load_json(path)
    cv, m, n = g["curve"], g["m"], g["n"]
    cards = engines(cv)
    pp = mp.Parameters(m, n, bytes.fromhex(g["params"]))
    pk = bytes.fromhex(g["pk"])
    cb = 2 * cards.engine.point_bytes          # one card = two points (128 B; 192 B on BLS12-377)
    deck = _split(bytes.fromhex(g["deck"]), cb)
    rho = [int.from_bytes(x, "little") for x in _split(bytes.fromhex(g["rho"]), 32)]
    shuffled, proof = cards.shuffle_and_remask(bytes.fromhex(g["prover_seed"]), pp, pk, deck, rho, mp.Permutation(g["perm"]))
    assert b"".join(shuffled).hex() == g["shuffled"]
    assert proof.hex() == g["proof"]
    assert cards.verify_shuffle(pp, pk, deck, shuffled, proof) is None
    with po.curve_ctx(po.CURVES[cv]):
        wrong = _split(po.deck_to_bytes(po.gen_inputs(po.CURVES[cv], m, n, g["seed"] + 1000)[2]), cb)
    with pytest.raises(mp.CryptoError) as ei:
        cards.verify_shuffle(pp, pk, deck, wrong, proof)
    assert ei.value == mp.CryptoError("Hadamard Product (5.1)")


@pytest.mark.parametrize("curve,m,n,B", [("stark", 2, 26, 6), ("stark", 4, 13, 5), ("stark", 6, 5, 3),
                                         ("bn254", 3, 5, 3), ("secp256k1", 2, 7, 3), ("bls12_377", 2, 5, 3)])
@pytest.mark.parametrize("plan", ["tiny", "latency", "medium", "wide", "throughput"])
def test_batch_matches_oracle(mp, engines, coracle, curve, m, n, B, plan):
    cards = engines(curve)
    g0 = coracle.gen_inputs(curve, m, n, 100)
    pp = mp.Parameters(m, n, g0["params"])
    pk = g0["pk"]
    # finest split up to 3/16 L proofs, latency plan up to L, medium plan up to 3.5 L, throughput beyond (L = 0: always
    # throughput); B is 3..6 here
    cards.table(pp, pk).set_latency_batch({"tiny": 8192, "latency": 8, "medium": 2, "wide": 8192, "throughput": 0}[plan])
    cards.table(pp, pk).set_work_split(4 if plan == "wide" else -1)      # (the wide split starts at 4x the medium one's batch)
    ins = []
    for b in range(B):
        g = coracle.gen_inputs(curve, m, n, 200 + b)     # same draw order => same params? no: own params per seed
        ins.append(g)
    # all proofs of a batch share the table's (params, pk): take decks / rho / perm / seeds from each input set
    res = cards.shuffle_and_remask_batch([g["prover_seed"] for g in ins], pp, pk,
                                         [_split(g["deck"], 2 * cards.engine.point_bytes) for g in ins],
                                         [[int.from_bytes(x, "little") for x in _split(g["rho"], 32)] for g in ins],
                                         [mp.Permutation(g["perm"]) for g in ins])
    decks, shufs, proofs = [], [], []
    for g, r in zip(ins, res):
        assert not isinstance(r, Exception), r
        exp_deck, exp_proof = coracle.shuffle_and_remask(curve, m, n, g0["params"], pk, g["deck"], g["rho"], g["perm"], g["prover_seed"])
        assert b"".join(r[0]) == exp_deck
        assert r[1] == exp_proof
        decks.append(_split(g["deck"], 2 * cards.engine.point_bytes)); shufs.append(r[0]); proofs.append(r[1])
    assert cards.verify_shuffle_batch(pp, pk, decks, shufs, proofs) == [None] * B
    # mixed batch: proof b checked against the deck of proof b+1 must fail by name, the others pass
    rot = shufs[1:] + shufs[:1]
    out = cards.verify_shuffle_batch(pp, pk, decks, rot, proofs)
    assert all(o == mp.CryptoError("Hadamard Product (5.1)") for o in out)
    cards.table(pp, pk).set_latency_batch(8192)
    cards.table(pp, pk).set_work_split(-1)


@pytest.mark.parametrize("fb_bits", [16, 20, 21])
def test_wide_fixed_base_windows_match_oracle(mp, coracle, fb_bits):
    """the throughput configurations (16-bit fixed-base windows: 2 GB of tables; 20-bit: 27 GB; 21-bit: 48 GB and one window fewer on the 252-bit STARK scalars) are
    bit-identical too"""
    cv, m, n, B = "stark", 2, 26, 4
    cards = mp.DLCards(cv, device=0, fb_bits=fb_bits)
    g0 = coracle.gen_inputs(cv, m, n, 100)
    pp = mp.Parameters(m, n, g0["params"])
    ins = [coracle.gen_inputs(cv, m, n, 700 + b) for b in range(B)]
    res = cards.shuffle_and_remask_batch([g["prover_seed"] for g in ins], pp, g0["pk"], [_split(g["deck"], 2 * cards.engine.point_bytes) for g in ins],
                                         [[int.from_bytes(x, "little") for x in _split(g["rho"], 32)] for g in ins],
                                         [mp.Permutation(g["perm"]) for g in ins])
    for g, r in zip(ins, res):
        exp_deck, exp_proof = coracle.shuffle_and_remask(cv, m, n, g0["params"], g0["pk"], g["deck"], g["rho"], g["perm"], g["prover_seed"])
        assert b"".join(r[0]) == exp_deck and r[1] == exp_proof
    assert cards.verify_shuffle_batch(pp, g0["pk"], [_split(g["deck"], 2 * cards.engine.point_bytes) for g in ins], [r[0] for r in res], [r[1] for r in res]) == [None] * B
    wrong = [res[(b + 1) % B][0] for b in range(B)]
    out = cards.verify_shuffle_batch(pp, g0["pk"], [_split(g["deck"], 2 * cards.engine.point_bytes) for g in ins], wrong, [r[1] for r in res])
    assert all(o == mp.CryptoError("Hadamard Product (5.1)") for o in out)


@pytest.mark.parametrize("curve,m,n,B", [("stark", 8, 128, 2), ("stark", 16, 64, 1), ("stark", 32, 32, 1),
                                         ("secp256k1", 2, 26, 3), ("bn254", 2, 26, 2), ("stark", 10, 30, 1),
                                         ("bls12_377", 2, 150, 1), ("bls12_377", 6, 50, 1), ("bls12_377", 10, 30, 1),
                                         ("bls12_377", 12, 25, 1), ("bls12_377", 30, 10, 1)])
def test_baseline_config_shapes(mp, engines, coracle, curve, m, n, B):
    """the other shapes BASELINE.json / the reference name: 1024-card decks as (8,128), (16,64), (32,32); secp256k1 and
    bn254 at 52 cards; the five (m,n) pairs of the 300-card sweep on BLS12-377 G1 [REF examples/parameter_selection.rs:25-57]
    -- bit-exact against the oracle"""
    cards = engines(curve)
    g0 = coracle.gen_inputs(curve, m, n, 900)
    pp = mp.Parameters(m, n, g0["params"])
    ins = [g0] + [coracle.gen_inputs(curve, m, n, 901 + b) for b in range(B - 1)]
    res = cards.shuffle_and_remask_batch([g["prover_seed"] for g in ins], pp, g0["pk"], [_split(g["deck"], 2 * cards.engine.point_bytes) for g in ins],
                                         [[int.from_bytes(x, "little") for x in _split(g["rho"], 32)] for g in ins],
                                         [mp.Permutation(g["perm"]) for g in ins])
    for g, r in zip(ins, res):
        exp_deck, exp_proof = coracle.shuffle_and_remask(curve, m, n, g0["params"], g0["pk"], g["deck"], g["rho"], g["perm"], g["prover_seed"])
        assert b"".join(r[0]) == exp_deck and r[1] == exp_proof
    decks = [_split(g["deck"], 2 * cards.engine.point_bytes) for g in ins]
    assert cards.verify_shuffle_batch(pp, g0["pk"], decks, [r[0] for r in res], [r[1] for r in res]) == [None] * B
    bad = list(res[0][0])
    bad[0], bad[1] = bad[1], bad[0]
    assert cards.verify_shuffle_batch(pp, g0["pk"], decks[:1], [bad], [res[0][1]]) == [mp.CryptoError("Hadamard Product (5.1)")]


def test_cpp_mirror(mp, coracle, tmp_path):
    """include/barnett_smart.hpp compiled with g++ against libmpshuffle.so"""
    import struct
    import subprocess
    from conftest import ROOT
    m, n = 2, 5
    g = coracle.gen_inputs("stark", m, n, 4242)
    exp_deck, exp_proof = coracle.shuffle_and_remask("stark", m, n, **g)
    case = tmp_path / "case.bin"
    case.write_bytes(struct.pack("<II", m, n) + g["params"] + g["pk"] + g["deck"] + g["rho"] +
                     struct.pack("<%dI" % (m * n), *g["perm"]) + g["prover_seed"] + exp_deck + exp_proof)
    exe = tmp_path / "mirror_smoke"
    libdir = os.path.join(ROOT, "mental-poker_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "mirror_smoke.cpp"),
                           "-L", libdir, "-lmpshuffle", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    out = subprocess.run([str(exe), str(case)], capture_output=True, text=True)
    assert out.returncode == 0 and "mirror_smoke ok" in out.stdout, out.stdout + out.stderr


def test_tampering_names_the_failing_check(mp, engines):
    g = load_json(os.path.join(GOLDEN, "shuffle_stark_m3_n4_s11.json"))
    m, n = g["m"], g["n"]
    cards = engines("stark")
    pp = mp.Parameters(m, n, bytes.fromhex(g["params"]))
    pk = bytes.fromhex(g["pk"])
    deck, shuf = _split(bytes.fromhex(g["deck"]), 128), _split(bytes.fromhex(g["shuffled"]), 128)
    pf = po.proof_from_bytes(bytes.fromhex(g["proof"]), m, n)
    q = po.STARK.q
    G = po.STARK.G

    def name(mut):
        p2 = copy.deepcopy(pf)
        mut(p2)
        try:
            cards.verify_shuffle(pp, pk, deck, shuf, po.proof_to_bytes(p2))
            return "Ok"
        except mp.CryptoError as e:
            return e.check

    def bump(d, k, i=None):
        if i is None:
            d[k] = (d[k] + 1) % q
        else:
            d[k][i] = (d[k][i] + 1) % q

    def setpt(d, k, i=None):
        if i is None:
            d[k] = G
        else:
            d[k][i] = G

    assert name(lambda p: None) == "Ok"
    assert name(lambda p: setpt(p["product"]["had"], "cB", 0)) == "Hadamard Product (5.1)"
    assert name(lambda p: setpt(p["product"]["had"], "cB", m - 1)) == "Hadamard Product (5.1)"
    assert name(lambda p: bump(p["product"]["had"]["zero"], "tbar")) == "Zero Argument (5.2)"
    assert name(lambda p: bump(p["product"]["had"]["zero"], "abar", 2)) == "Zero Argument (5.2)"
    assert name(lambda p: bump(p["product"]["had"]["zero"], "bbar", 0)) == "Zero Argument (5.2)"
    assert name(lambda p: setpt(p["product"]["had"]["zero"], "cD", m + 1)) == "Zero Argument (5.2)"
    assert name(lambda p: bump(p["product"]["svp"], "rt")) == "Single Value Product (5.3)"
    assert name(lambda p: bump(p["product"]["svp"], "st")) == "Single Value Product (5.3)"
    assert name(lambda p: bump(p["product"]["svp"], "bt", 0)) == "Single Value Product (5.3)"
    assert name(lambda p: bump(p["mexp"], "taubar")) == "Multi-Exponentiation Argument (4)"
    assert name(lambda p: bump(p["mexp"], "abar", 1)) == "Multi-Exponentiation Argument (4)"
    assert name(lambda p: bump(p["mexp"], "sbar")) == "Multi-Exponentiation Argument (4)"
    assert name(lambda p: setpt(p["mexp"], "cB", m)) == "Multi-Exponentiation Argument (4)"
    # the oracle names the same check for every one of these
    for mut in (lambda p: bump(p["mexp"], "rbar"), lambda p: bump(p["product"]["svp"], "at", 1)):
        p2 = copy.deepcopy(pf)
        mut(p2)
        exp = po.CHECK_NAMES[po.verify_shuffle(*_po_args(g), po.deck_from_bytes(bytes.fromhex(g["shuffled"])), p2)]
        assert name(mut) == exp


@pytest.mark.parametrize("curve,m,n,B", [("stark", 2, 26, 5), ("stark", 4, 13, 3), ("secp256k1", 2, 7, 3), ("bls12_377", 2, 5, 2)])
@pytest.mark.parametrize("plan", ["tiny", "latency", "throughput"])
def test_keyed_batches_match_oracle(mp, engines, coracle, curve, m, n, B, plan):
    """mp_*_batch_keys: every proof of the batch under its own aggregate key (tables of a card server share the parameters
    and differ in the key [REF mod.rs:380-386]) -- byte-identical to the oracle run with that key; a proof checked under
    another table's key is rejected"""
    cards = engines(curve)
    g0 = coracle.gen_inputs(curve, m, n, 700)
    pp = mp.Parameters(m, n, g0["params"])
    t = cards.table(pp, g0["pk"])
    t.set_latency_batch({"tiny": 8192, "latency": 8, "throughput": 0}[plan])
    ins = [coracle.gen_inputs(curve, m, n, 701 + b) for b in range(B)]
    keys, decks = b"".join(g["pk"] for g in ins), b"".join(g["deck"] for g in ins)
    d, p, st = t.shuffle_and_remask_batch_keys(keys, decks, b"".join(g["rho"] for g in ins), [v for g in ins for v in g["perm"]],
                                               b"".join(g["prover_seed"] for g in ins))
    assert st == [0] * B
    cb, ps = len(g0["deck"]), t.proof_bytes
    for b, g in enumerate(ins):
        ed, ep = coracle.shuffle_and_remask(curve, m, n, g0["params"], g["pk"], g["deck"], g["rho"], g["perm"], g["prover_seed"])
        assert d[b * cb:(b + 1) * cb] == ed and p[b * ps:(b + 1) * ps] == ep
    assert t.verify_shuffle_batch_keys(keys, decks, d, p) == [0] * B
    rot = b"".join(g["pk"] for g in ins[1:] + ins[:1])
    assert all(v > 0 for v in t.verify_shuffle_batch_keys(rot, decks, d, p))
    t.set_latency_batch(8192)


def test_chunked_host_pipeline(mp, engines, coracle):
    """mp_*_batch with a batch cut into several pipelined chunks (upload, kernels and download on three streams): 11 proofs in
    chunks of 4 equal the single-chunk result and the oracle"""
    cv, m, n, B = "stark", 2, 26, 11
    cards = engines(cv)
    ins = [coracle.gen_inputs(cv, m, n, 900 + b) for b in range(B)]
    g0 = ins[0]
    t = cards.table(mp.Parameters(m, n, g0["params"]), g0["pk"])
    args = (b"".join(g["deck"] for g in ins), b"".join(g["rho"] for g in ins), [v for g in ins for v in g["perm"]],
            b"".join(g["prover_seed"] for g in ins))
    ref = t.shuffle_and_remask_batch(*args)
    t.set_io_chunk(4)
    got = t.shuffle_and_remask_batch(*args)
    assert got == ref and got[2] == [0] * B
    cb, ps = len(g0["deck"]), t.proof_bytes
    for b in (0, 3, 4, 10):
        g = ins[b]
        ed, ep = coracle.shuffle_and_remask(cv, m, n, g0["params"], g0["pk"], g["deck"], g["rho"], g["perm"], g["prover_seed"])
        assert got[0][b * cb:(b + 1) * cb] == ed and got[1][b * ps:(b + 1) * ps] == ep
    assert t.verify_shuffle_batch(args[0], got[0], got[1]) == [0] * B
    swapped = got[1][ps:2 * ps] + got[1][:ps] + got[1][2 * ps:]
    st = t.verify_shuffle_batch(args[0], got[0], swapped)
    assert st[0] > 0 and st[1] > 0 and st[2:] == [0] * (B - 2)
    t.set_io_chunk(0)


def test_verifier_fuzz_against_oracle(mp, engines, coracle):
    """~200 random single-element corruptions of an honest proof (a scalar replaced by a random scalar, a point by
    another valid point, or two elements swapped), verified in ONE batch: every status word -- accept, or the code of the first failing
    check -- equals the C++ oracle's verdict, under both verification strategies"""
    g = load_json(os.path.join(GOLDEN, "shuffle_stark_m4_n13_s9.json"))
    m, n = g["m"], g["n"]
    cards = engines("stark")
    pp = mp.Parameters(m, n, bytes.fromhex(g["params"]))
    pk, deck, shuf = bytes.fromhex(g["pk"]), bytes.fromhex(g["deck"]), bytes.fromhex(g["shuffled"])
    good = bytes.fromhex(g["proof"])
    npts, nsc = 11 * m + 8, 5 * n + 9
    # element boundaries of wire v1: (offset, size) of every element
    with po.curve_ctx(po.STARK):
        pf = po.proof_from_bytes(good, m, n)
    sizes = []

    def walk(x):
        if isinstance(x, dict):
            for k in x:
                walk(x[k])
        elif isinstance(x, list):
            for v in x:
                walk(v)
        elif x is None:
            sizes.append(64)                                    # point at infinity (c_D[m+1], c_B[m] of the honest proof)
        elif isinstance(x, tuple):
            if len(x) == 2 and all(isinstance(v, int) for v in x):
                sizes.append(64)
            else:
                for v in x:
                    walk(v)
        else:
            sizes.append(32)
    walk(pf)            # dict order of proof_from_bytes == wire order (asserted below)
    assert sum(sizes) == len(good) and sizes.count(64) == npts and sizes.count(32) == nsc
    offs = [sum(sizes[:i]) for i in range(len(sizes))]
    rng = mp.ChaCha20Rng(bytes([9] * 32))
    q = po.STARK.q
    pts = [deck[64 * i:64 * i + 64] for i in range(8)]
    proofs = [good]
    for trial in range(199):
        b = bytearray(good)
        e = rng.next_u64() % len(sizes)
        kind = rng.next_u64() % 3
        if kind == 2:                                           # swap two elements of the same size
            f = next(i for i in range(e + 1, e + len(sizes)) if sizes[i % len(sizes)] == sizes[e]) % len(sizes)
            o1, o2, sz = offs[e], offs[f], sizes[e]
            b[o1:o1 + sz], b[o2:o2 + sz] = b[o2:o2 + sz], b[o1:o1 + sz]
        elif sizes[e] == 32:
            b[offs[e]:offs[e] + 32] = (mp.fr_rand("stark", rng) if kind else (int.from_bytes(b[offs[e]:offs[e] + 32], "little") + 1) % q).to_bytes(32, "little")
        else:
            b[offs[e]:offs[e] + 64] = pts[rng.next_u64() % len(pts)]
        proofs.append(bytes(b))
    B = len(proofs)
    exp = [coracle.verify_shuffle("stark", m, n, bytes.fromhex(g["params"]), pk, deck, shuf, p) for p in proofs]
    assert exp[0] == 0 and sum(1 for v in exp if v) > 150 and len(set(exp)) == 5      # every check code occurs
    t = cards.table(pp, pk)
    t.set_latency_batch(0)
    for mode in (True, False):
        t.set_merged_verify(mode)
        assert t.verify_shuffle_batch(deck * B, shuf * B, b"".join(proofs)) == exp
    t.set_merged_verify(True)
    t.set_latency_batch(8192)


def test_merged_and_per_equation_verification_agree(mp, engines):
    """mp_set_merged_verify: the merged screening pass and the equation-by-equation pass give the same status words on a
    mixed batch (honest proofs, one tampering per sub-argument, a bad encoding), and an all-honest batch passes in both"""
    g = load_json(os.path.join(GOLDEN, "shuffle_stark_m3_n4_s11.json"))
    m, n = g["m"], g["n"]
    cards = engines("stark")
    pp = mp.Parameters(m, n, bytes.fromhex(g["params"]))
    pk = bytes.fromhex(g["pk"])
    deck, shuf = bytes.fromhex(g["deck"]), bytes.fromhex(g["shuffled"])
    pf = po.proof_from_bytes(bytes.fromhex(g["proof"]), m, n)
    q = po.STARK.q

    def tampered(path, key, idx=None):
        p2 = copy.deepcopy(pf)
        d = p2
        for k in path:
            d = d[k]
        if idx is None:
            d[key] = (d[key] + 1) % q
        else:
            d[key][idx] = (d[key][idx] + 1) % q
        return po.proof_to_bytes(p2)

    good = bytes.fromhex(g["proof"])
    proofs = [good, tampered(["product", "had", "zero"], "tbar"), good, tampered(["product", "svp"], "rt"),
              tampered(["mexp"], "taubar"), tampered(["mexp"], "abar", 1), good]
    garbage = bytearray(good)
    garbage[0:32] = b"\xff" * 32                       # x coordinate >= p: bad encoding, reported as a usage error
    proofs.append(bytes(garbage))
    # errors that cancel under EQUAL weights: +1 on the H-coefficient of one equation, -1 on the H-coefficient of another
    # (zero argument, equations A and B) -- the merged equation must still see them (its weights are random)
    p3 = copy.deepcopy(pf)
    z = p3["product"]["had"]["zero"]
    z["rbar"], z["sbar"] = (z["rbar"] + 1) % q, (z["sbar"] - 1) % q
    proofs.append(po.proof_to_bytes(p3))
    B = len(proofs)
    t = cards.table(pp, pk)
    t.set_latency_batch(0)          # small batches skip the screening pass unless the throughput plan is forced
    out = {}
    for mode in (True, False):
        t.set_merged_verify(mode)
        out[mode] = t.verify_shuffle_batch(deck * B, shuf * B, b"".join(proofs))
        assert t.verify_shuffle_batch(deck * 3, shuf * 3, good * 3) == [0, 0, 0]
    t.set_merged_verify(True)
    t.set_latency_batch(8192)
    assert out[True] == out[False]
    assert out[True] == [0, 2, 0, 3, 4, 4, 0, -1, 2]


def _po_args(g):
    cv = po.CURVES[g["curve"]]
    pp, pk, deck, rho, perm, ps = po.gen_inputs(cv, g["m"], g["n"], g["seed"])
    return pp, pk, deck


@pytest.mark.parametrize("cvn", ["stark", "bn254", "secp256k1", "bls12_377"])
def test_edge_inputs(mp, engines, cvn):
    """identity permutation, rho in {0, 1, q-1}, duplicate cards, a point-at-infinity component -- on every base-field
    form (9x29 sparse, 9x29 dense, 9x29 signed sparse, 14x29 dense): equal points meet in the window tables and in the accumulators
    (P + P, P - P), which is where a lazily reduced zero has to be recognised"""
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
        cards = engines(cvn)
        P = mp.Parameters(m, n, po.params_to_bytes(pp))
        cb = 2 * cards.engine.point_bytes
        wdeck = _split(po.deck_to_bytes(deck), cb)
        shuffled, proof = cards.shuffle_and_remask(ps, P, po.pt_wire(pk), wdeck, rho, mp.Permutation(perm))
        assert b"".join(shuffled) == po.deck_to_bytes(sh)
        assert proof == po.proof_to_bytes(pf)
        assert cards.verify_shuffle(P, po.pt_wire(pk), wdeck, shuffled, proof) is None


def test_usage_errors_are_io_errors(mp, engines, coracle):
    cv, m, n = "stark", 2, 3
    g = coracle.gen_inputs(cv, m, n, 5)
    cards = engines(cv)
    pp = mp.Parameters(m, n, g["params"])
    deck = _split(g["deck"], 2 * cards.engine.point_bytes)
    rho = [int.from_bytes(x, "little") for x in _split(g["rho"], 32)]
    good = cards.shuffle_and_remask(g["prover_seed"], pp, g["pk"], deck, rho, mp.Permutation(g["perm"]))
    with pytest.raises(mp.CardProtocolError) as e:
        cards.shuffle_and_remask(g["prover_seed"], pp, g["pk"], deck, rho, mp.Permutation([0, 0, 1, 2, 3, 4]))
    assert e.value.kind == "IoError"
    with pytest.raises(mp.CardProtocolError):
        cards.shuffle_and_remask(g["prover_seed"], pp, g["pk"], deck, [po.STARK.q] + rho[1:], mp.Permutation(g["perm"]))
    bad_card = bytearray(deck[0])
    bad_card[0] ^= 1   # x changed: not on the curve any more
    with pytest.raises(mp.CardProtocolError):
        cards.shuffle_and_remask(g["prover_seed"], pp, g["pk"], [bytes(bad_card)] + deck[1:], rho, mp.Permutation(g["perm"]))
    pf = bytearray(good[1])
    pf[-32:] = b"\xff" * 32          # non-canonical scalar
    with pytest.raises(mp.CardProtocolError):
        cards.verify_shuffle(pp, g["pk"], deck, good[0], bytes(pf))
    with pytest.raises(mp.CardProtocolError):
        cards.verify_shuffle(pp, g["pk"], deck, good[0], good[1][:-1])   # wrong length


def test_setup_matches_oracle(mp, engines):
    for cvn in ("stark", "bn254", "secp256k1"):
        seed = bytes(range(7, 39))
        pp = engines(cvn).setup(seed, 3, 5)
        exp = po.setup(po.CURVES[cvn], 3, 5, po.ChaCha20Rng(seed))
        assert pp.raw == po.params_to_bytes(exp)


def test_building_blocks(mp, engines, coracle):
    cv, m, n = "stark", 2, 6
    g = coracle.gen_inputs(cv, m, n, 77)
    t = engines(cv).table(mp.Parameters(m, n, g["params"]), g["pk"])
    # remask (K3)
    assert t.remask_batch(g["deck"], g["rho"]) == coracle.remask_deck(cv, g["params"][:64], g["pk"], g["deck"], g["rho"])
    # variable-base MSM (K5): 5 MSMs of 7 terms, 3 of 33 terms, with zero / one / q-1 scalars
    pts = [g["deck"][i * 64:(i + 1) * 64] for i in range(2 * m * n)]
    for n_msm, k in ((5, 7), (3, 33), (2, 1)):
        sc, pt, exp = b"", b"", b""
        for j in range(n_msm):
            s_j = [int.from_bytes(g["rho"][((j + t_) % (m * n)) * 32:((j + t_) % (m * n) + 1) * 32], "little") for t_ in range(k)]
            s_j[0] = 0
            if k > 2:
                s_j[1] = 1
                s_j[2] = po.STARK.q - 1
            p_j = [pts[(3 * j + t_) % len(pts)] for t_ in range(k)]
            sb = b"".join(v.to_bytes(32, "little") for v in s_j)
            sc += sb
            pt += b"".join(p_j)
            exp += coracle.msm(cv, sb, b"".join(p_j), 0)
        assert t.msm(n_msm, k, sc, pt) == exp
    # Pedersen commitments (K4)
    vals = g["rho"][:32 * n]
    r = g["rho"][32 * n:32 * (n + 1)]
    for length in (n, n - 1, 1):
        exp = coracle.commit(cv, n, g["params"], vals[:32 * length], r)
        assert t.commit_batch(1, length, vals[:32 * length], r) == exp


def test_full_size_properties(mp, engines, coracle):
    """BASELINE size (52 cards, m=2, n=26), a few hundred proofs: every honest proof verifies, outputs do not
    depend on the position in the batch, a chain of dependent shuffles (deck_{j+1} = output_j) verifies, and
    spot proofs equal the oracle's."""
    cv, m, n, B = "stark", 2, 26, 640
    N = m * n
    g = coracle.gen_inputs(cv, m, n, 31337)
    cards = engines(cv)
    pp = mp.Parameters(m, n, g["params"])
    cards.table(pp, g["pk"]).set_latency_batch(0)       # the throughput plan (and the merged verifier), as at full batch sizes
    deck = _split(g["deck"], 2 * cards.engine.point_bytes)
    rng = mp.ChaCha20Rng(b"\x05" * 32)
    q = po.STARK.q
    seeds, rhos, perms = [], [], []
    for b in range(B):
        seeds.append(b"".join((rng.next_u64()).to_bytes(8, "little") for _ in range(4)))
        rhos.append([(rng.next_u64() | (rng.next_u64() << 64) | (rng.next_u64() << 128) | ((rng.next_u64() >> 6) << 192)) % q for _ in range(N)])
        perms.append(mp.Permutation.new(rng, N))
    # duplicate entry 0 at the end: same inputs at a different batch position
    seeds.append(seeds[0]); rhos.append(rhos[0]); perms.append(perms[0])
    res = cards.shuffle_and_remask_batch(seeds, pp, g["pk"], [deck] * (B + 1), rhos, perms)
    assert all(not isinstance(r, Exception) for r in res)
    assert res[0] == res[B]
    for b in (0, B // 2, B - 1):
        exp_deck, exp_proof = coracle.shuffle_and_remask(cv, m, n, g["params"], g["pk"], g["deck"],
                                                         b"".join(v.to_bytes(32, "little") for v in rhos[b]), perms[b].mapping, seeds[b])
        assert b"".join(res[b][0]) == exp_deck and res[b][1] == exp_proof
    out = cards.verify_shuffle_batch(pp, g["pk"], [deck] * (B + 1), [r[0] for r in res], [r[1] for r in res])
    assert out == [None] * (B + 1)
    # chain: each player shuffles the previous output [REF examples/round.rs:263-350]
    cur = deck
    for j in range(3):
        nxt, proof = cards.shuffle_and_remask(seeds[j], pp, g["pk"], cur, rhos[j], perms[j])
        assert cards.verify_shuffle(pp, g["pk"], cur, nxt, proof) is None
        with pytest.raises(mp.CryptoError):
            cards.verify_shuffle(pp, g["pk"], deck if j else nxt, nxt if j else cur, proof)
        cur = nxt
    cards.table(pp, g["pk"]).set_latency_batch(8192)


# ---- the bucket-method kernel (kernels_bucket.hpp): large MSMs run on it by default (>= 2048 terms: the merged verifier
# equation of the 1024-card shapes in test_baseline_config_shapes); here it is forced onto small shapes and exercised directly
@pytest.mark.parametrize("curve,m,n,B", [("stark", 2, 26, 70), ("stark", 4, 13, 5), ("secp256k1", 2, 7, 3), ("bls12_377", 2, 5, 3)])
def test_bucket_kernel_matches_oracle(mp, coracle, curve, m, n, B):
    cards = mp.DLCards(curve, device=0)
    g0 = coracle.gen_inputs(curve, m, n, 100)
    pp, pk = mp.Parameters(m, n, g0["params"]), g0["pk"]
    t = cards.table(pp, pk)
    t.set_bucket_min(4)
    t.set_latency_batch(0)
    cards.engine.profile_enable(True)
    ins = [coracle.gen_inputs(curve, m, n, 900 + b) for b in range(B)]
    cb = 2 * cards.engine.point_bytes
    res = cards.shuffle_and_remask_batch([g["prover_seed"] for g in ins], pp, pk, [_split(g["deck"], cb) for g in ins],
                                         [[int.from_bytes(x, "little") for x in _split(g["rho"], 32)] for g in ins],
                                         [mp.Permutation(g["perm"]) for g in ins])
    decks, shufs, proofs = [], [], []
    for i, (g, r) in enumerate(zip(ins, res)):
        assert not isinstance(r, Exception), r
        if i < 4:
            exp_deck, exp_proof = coracle.shuffle_and_remask(curve, m, n, g0["params"], pk, g["deck"], g["rho"], g["perm"], g["prover_seed"])
            assert b"".join(r[0]) == exp_deck and r[1] == exp_proof
        decks.append(_split(g["deck"], cb)); shufs.append(r[0]); proofs.append(r[1])
    for merged in (True, False):
        t.set_merged_verify(merged)
        assert cards.verify_shuffle_batch(pp, pk, decks, shufs, proofs) == [None] * B
        out = cards.verify_shuffle_batch(pp, pk, decks, shufs[1:] + shufs[:1], proofs)
        assert all(o == mp.CryptoError("Hadamard Product (5.1)") for o in out)
    rep = cards.engine.profile_report()
    assert rep["k_bucket_msm"][0] >= 3


@pytest.mark.parametrize("curve,m,n,B,keyed", [("stark", 2, 26, 70, False), ("stark", 4, 13, 1500, False), ("stark", 2, 3, 20000, False),
                                               ("secp256k1", 2, 7, 3, True), ("bls12_377", 2, 5, 3, False), ("bn254", 3, 5, 1, True)])
def test_transcript_lane_modes_agree(mp, coracle, curve, m, n, B, keyed):
    """mp_set_transcript_lanes / mp_set_group_lanes: one lane per proof and four lanes per BLAKE2s state (k_fsq_*: 64, 32 and 4 lanes
    of a wave per proof at these batch sizes), one lane and four lanes per group operation of the MSM chains (k_var_msm_q,
    k_bucket_fold_q) produce the same proofs -- the oracle's -- and the same verdicts, honest and tampered"""
    cards = mp.DLCards(curve, device=0)
    g0 = coracle.gen_inputs(curve, m, n, 100)
    pp, pk = mp.Parameters(m, n, g0["params"]), g0["pk"]
    t = cards.table(pp, pk)
    cards.engine.profile_enable(True)
    distinct = min(B, 6)
    ins = [coracle.gen_inputs(curve, m, n, 700 + b) for b in range(distinct)]
    rep = (B + distinct - 1) // distinct
    cat = lambda key: (b"".join(g[key] for g in ins) * rep)[:B * len(ins[0][key])]
    decks, rho, seeds = cat("deck"), cat("rho"), cat("prover_seed")
    perms = ([x for g in ins for x in g["perm"]] * rep)[:B * m * n]
    keys = (b"".join(coracle.gen_inputs(curve, m, n, 800 + b)["pk"] for b in range(distinct)) * rep)[:B * cards.engine.point_bytes]
    out = {}
    for lanes in (1, 4):
        t.set_transcript_lanes(lanes)
        t.set_group_lanes(lanes if B <= 1500 else 1)         # (kernels_quad.hpp: four lanes per group operation)
        if keyed:
            sh, pf, st = t.shuffle_and_remask_batch_keys(keys, decks, rho, perms, seeds)
        else:
            sh, pf, st = t.shuffle_and_remask_batch(decks, rho, perms, seeds)
        assert st == [0] * B
        bad = bytearray(pf)
        bad[-96] ^= 1                                        # low byte of a response scalar of the last proof
        ver = (lambda d, s_, p: t.verify_shuffle_batch_keys(keys, d, s_, p)) if keyed else t.verify_shuffle_batch
        verdicts = []
        for merged in (True, False):
            t.set_merged_verify(merged)
            verdicts.append((ver(decks, sh, pf), ver(decks, sh, bytes(bad))))
        t.set_merged_verify(True)
        out[lanes] = (sh, pf, verdicts)
    t.set_transcript_lanes(0)
    t.set_group_lanes(0)
    assert out[1] == out[4]
    sh, pf, verdicts = out[4]
    assert verdicts[0][0] == [0] * B and verdicts[0] == verdicts[1] and verdicts[0][1][:-1] == [0] * (B - 1) and verdicts[0][1][-1] > 0
    if not keyed:
        psz = len(pf) // B
        for b in range(distinct):
            g = ins[b]
            exp_deck, exp_proof = coracle.shuffle_and_remask(curve, m, n, g0["params"], pk, g["deck"], g["rho"], g["perm"], g["prover_seed"])
            assert sh[b * len(exp_deck):(b + 1) * len(exp_deck)] == exp_deck and pf[b * psz:(b + 1) * psz] == exp_proof
    rp = cards.engine.profile_report()
    for k in ("k_fsq_round1", "k_fsq_round", "k_fsq_verify", "k_fs_round1", "k_fs_round", "k_verify_fs", "k_var_msm") + \
            (("k_var_msm_q",) if B <= 1500 else ()):
        assert rp[k][0] >= 1, k


def test_alternating_batch_sizes_are_consistent(mp, coracle):
    """one table, batches of very different sizes back to back (they take different kernels -- four-lane or one-lane transcripts and
    group operations, one or two prover streams, the five static work splits -- and share the context's streams, events and arenas):
    proof 0 is the oracle's every time, every batch verifies, and a batch verified right after a larger one gets its own verdicts"""
    curve, m, n = "stark", 2, 4
    cards = mp.DLCards(curve, device=0)
    g0 = coracle.gen_inputs(curve, m, n, 100)
    pp, pk = mp.Parameters(m, n, g0["params"]), g0["pk"]
    t = cards.table(pp, pk)
    ins = [coracle.gen_inputs(curve, m, n, 300 + b) for b in range(4)]
    exp = [coracle.shuffle_and_remask(curve, m, n, g0["params"], pk, g["deck"], g["rho"], g["perm"], g["prover_seed"]) for g in ins]
    dsz, psz = len(exp[0][0]), len(exp[0][1])
    for rnd, B in enumerate([1, 40000, 3, 70, 1500, 2, 33000, 5, 1, 9000]):
        rep = (B + 3) // 4
        cat = lambda key: (b"".join(g[key] for g in ins) * rep)[:B * len(ins[0][key])]
        decks, rho, seeds = cat("deck"), cat("rho"), cat("prover_seed")
        perms = ([x for g in ins for x in g["perm"]] * rep)[:B * m * n]
        sh, pf, st = t.shuffle_and_remask_batch(decks, rho, perms, seeds)
        assert st == [0] * B, (B, [x for x in st if x][:3])
        for b in sorted({0, B // 2, B - 1}):
            assert sh[b * dsz:(b + 1) * dsz] == exp[b % 4][0] and pf[b * psz:(b + 1) * psz] == exp[b % 4][1], (B, b)
        bad = bytearray(pf)
        bad[(B - 1) * psz + psz - 96] ^= 1
        assert t.verify_shuffle_batch(decks, sh, pf) == [0] * B
        v = t.verify_shuffle_batch(decks, sh, bytes(bad))
        assert v[:-1] == [0] * (B - 1) and v[-1] > 0, (B, v[-3:])


def test_four_lane_group_law_matches_one_lane(mp):
    """kernels_quad.hpp on the device: doubling, mixed addition / subtraction and full addition (incl. P + P) on four lanes per
    operation against curve.hpp's one-lane forms, 16 different points per curve, with and without divergence between the quads of a
    wave (tools/quadcheck/quad_check.hip, built with the library)"""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "quadcheck", "quad_check")
    if not os.path.exists(exe):
        mp._native.build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600).stdout
    lines = [l for l in out.splitlines() if "fails mask" in l]
    assert len(lines) == 4, out
    for l in lines:
        assert "fails mask 0x0 " in l, l


def test_bucket_msm_large_and_edge_scalars(mp, coracle):
    cv = "stark"
    q = 0x0800000000000010ffffffffffffffffb781126dcae7b2321e66a241adc64d2f
    eng = mp.Engine(cv, device=0)
    gi = coracle.gen_inputs(cv, 2, 3, 5)
    t = eng.table(2, 3, gi["params"], gi["pk"])
    rnd = random.Random(11)
    for K, n_msm in ((2500, 3), (4193, 2), (150, 5)):
        t.set_bucket_min(2048 if K > 2048 else 16)
        pts = bytearray(eng.setup(2, K - 3, bytes([9] * 32)))
        pts[64 * 7:64 * 8] = pts[64 * 6:64 * 7]
        pts[64 * 9:64 * 10] = bytes(64)
        sc, allp = b"", b""
        for j in range(n_msm):
            s = [rnd.randrange(q) for _ in range(K)]
            s[0:12] = [0, 1, q - 1, 128, 127, 129, 2 ** 248, 2 ** 251, 255, 256 * 128, q - 128, q - 129]
            if j == 1:
                s = [77] * K                      # every term of every window in one bucket
            sc += b"".join(v.to_bytes(32, "little") for v in s)
            allp += bytes(pts)
        got = t.msm(n_msm, K, sc, allp)
        for j in range(n_msm):
            assert got[64 * j:64 * j + 64] == coracle.msm(cv, sc[32 * K * j:32 * K * (j + 1)], bytes(pts)), (K, j)


@pytest.mark.parametrize("cvn,m,n,L,T,keyed", [("stark", 2, 26, 8, 40, True), ("stark", 4, 13, 3, 5, False), ("secp256k1", 2, 7, 4, 3, True),
                                                ("stark", 8, 128, 2, 2, True),
                                                ("stark", 2, 26, 32, 3, True)])     # BASELINE config 3 at its stated length: 32 players
def test_chain_verification_matches_per_link_verifier(mp, coracle, cvn, m, n, L, T, keyed):
    """mp_verify_shuffle_chain (one equation per table over all links of its shuffle chain) accepts honest chains and, when a link
    is bad, returns exactly the status words of the per-link verifier"""
    eng = mp.Engine(cvn, device=0)
    N, pb = m * n, eng.point_bytes
    g0 = coracle.gen_inputs(cvn, m, n, 50)
    params = g0["params"]
    keys_t = [coracle.gen_inputs(cvn, m, n, 60 + t)["pk"] for t in range(T)] if keyed else [g0["pk"]] * T
    table = eng.table(m, n, params, None if keyed else g0["pk"])
    decks = [[coracle.gen_inputs(cvn, m, n, 70 + t)["deck"] for t in range(T)]]
    proofs = []
    for j in range(L):
        ins = [coracle.gen_inputs(cvn, m, n, 1000 + 50 * j + t) for t in range(T)]
        args = (b"".join(decks[j]), b"".join(g["rho"] for g in ins), [v for g in ins for v in g["perm"]], b"".join(g["prover_seed"] for g in ins))
        d, p, st = (table.shuffle_and_remask_batch_keys(b"".join(keys_t), *args) if keyed else table.shuffle_and_remask_batch(*args))
        assert not any(st)
        decks.append(_split(d, N * 2 * pb))
        proofs.append(_split(p, table.proof_bytes))
    flat_d = b"".join(b"".join(r) for r in decks)
    keys = b"".join(keys_t) * L if keyed else None
    assert table.verify_shuffle_chain(T, L, flat_d, b"".join(b"".join(r) for r in proofs), keys) == [0] * (L * T)
    bad = [r[:] for r in proofs]
    bad[L - 1][1] = proofs[L - 1][0]
    bad[0][T - 1] = proofs[0][0]
    st = table.verify_shuffle_chain(T, L, flat_d, b"".join(b"".join(r) for r in bad), keys)
    exp = []
    for j in range(L):
        a = (b"".join(decks[j]), b"".join(decks[j + 1]), b"".join(bad[j]))
        exp += table.verify_shuffle_batch_keys(b"".join(keys_t), *a) if keyed else table.verify_shuffle_batch(*a)
    assert st == exp and sum(1 for v in st if v) == 2


@pytest.mark.parametrize("cvn,m,n,K,B", [("stark", 2, 26, 5, 64), ("secp256k1", 2, 26, 3, 7), ("bls12_377", 2, 5, 2, 3), ("stark", 8, 16, 4, 6000)])
def test_key_sets_match_oracle_and_keyed_batches(mp, coracle, cvn, m, n, K, B):
    """key sets (mp_keyset_create, _keyset_dev entry points): proofs made under key key_index[b] of the set are the oracle's bytes
    under that key (spot-checked) and the bytes of the per-call keyed path (all of them); verification by index agrees with
    verification under explicit keys, a wrong index is a wrong statement, an index outside the set a usage error.  The device
    arrays are page-locked host buffers from mp_host_alloc (device-addressable; keeps torch out of this process)."""
    import ctypes
    import numpy as np
    eng = mp.Engine(cvn, device=0)
    N, pb = m * n, eng.point_bytes
    g0 = coracle.gen_inputs(cvn, m, n, 50)
    table = eng.table(m, n, g0["params"], None)
    lib = table.lib
    held = []

    def dev(raw=None, dtype=np.uint8, count=None):
        nbytes = len(raw) if raw is not None else count * np.dtype(dtype).itemsize
        p = lib.mp_host_alloc(nbytes)
        assert p
        held.append(p)
        a = np.frombuffer((ctypes.c_uint8 * nbytes).from_address(p), dtype=dtype)
        if raw is not None:
            a[...] = np.frombuffer(raw, dtype=dtype)
        return a

    ptr = lambda a: a.ctypes.data
    keys = [coracle.gen_inputs(cvn, m, n, 60 + k)["pk"] for k in range(K)]
    ks = table.keyset(b"".join(keys))
    distinct = [coracle.gen_inputs(cvn, m, n, 3000 + i) for i in range(min(B, 11))]
    ins = [distinct[b % 11] for b in range(B)]
    decks, rho = dev(b"".join(g["deck"] for g in ins)), dev(b"".join(g["rho"] for g in ins))
    perms = dev(np.array([v for g in ins for v in g["perm"]], dtype=np.uint32).tobytes(), np.uint32)
    seeds = dev(b"".join(g["prover_seed"] for g in ins))
    kidx = dev((np.arange(B, dtype=np.uint32) * 7 % K).astype(np.uint32).tobytes(), np.uint32)
    out_d, out_p = dev(count=B * N * 2 * pb), dev(count=B * table.proof_bytes)
    st, sv = dev(count=B, dtype=np.int32), dev(count=B, dtype=np.int32)
    st[...] = 77
    table.shuffle_and_remask_batch_keyset_dev(ks, B, ptr(kidx), ptr(decks), ptr(rho), ptr(perms), ptr(seeds), ptr(out_d), ptr(out_p), ptr(st))
    eng.sync()
    assert not st.any()
    per_proof_keys = b"".join(keys[int(i)] for i in kidx)
    d2, p2, st2 = table.shuffle_and_remask_batch_keys(per_proof_keys, decks.tobytes(), rho.tobytes(), [int(v) for v in perms], seeds.tobytes())
    assert not any(st2) and out_d.tobytes() == d2 and out_p.tobytes() == p2
    for b in (0, B // 2, B - 1):
        g = ins[b]
        ed, ep = coracle.shuffle_and_remask(cvn, m, n, g0["params"], keys[int(kidx[b])], g["deck"], g["rho"], g["perm"], g["prover_seed"])
        assert d2[b * N * 2 * pb:(b + 1) * N * 2 * pb] == ed and p2[b * table.proof_bytes:(b + 1) * table.proof_bytes] == ep
    sv[...] = 77
    table.verify_shuffle_batch_keyset_dev(ks, B, ptr(kidx), ptr(decks), ptr(out_d), ptr(out_p), ptr(sv))
    eng.sync()
    assert not sv.any()
    good = kidx.copy()
    kidx[B - 1] = (int(kidx[B - 1]) + 1) % K
    table.verify_shuffle_batch_keyset_dev(ks, B, ptr(kidx), ptr(decks), ptr(out_d), ptr(out_p), ptr(sv))
    eng.sync()
    assert int((sv != 0).sum()) == 1 and int(sv[B - 1]) > 0
    kidx[...] = good
    kidx[0] = K
    table.shuffle_and_remask_batch_keyset_dev(ks, B, ptr(kidx), ptr(decks), ptr(rho), ptr(perms), ptr(seeds), ptr(out_d), ptr(out_p), ptr(st))
    eng.sync()
    assert int(st[0]) == -3 and not st[1:].any()
    ks.close()
    table.close()
    for p in held:
        lib.mp_host_free(p)


@pytest.mark.parametrize("m,n", [(3, 2), (5, 3), (7, 4), (9, 3), (12, 2), (16, 2), (17, 2), (16, 5)])
def test_every_plan_family_matches_oracle(mp, coracle, m, n):
    """Toom-Cook (3 <= m <= 16, direct and reciprocal points), Karatsuba (m = 17) and the bucket kernel forced onto these small shapes:
    proof bytes equal the CPU oracle's, both verification strategies accept, a rotated batch is rejected by name"""
    cv, B = "stark", 3
    cards = mp.DLCards(cv, device=0)
    g0 = coracle.gen_inputs(cv, m, n, 4242)
    pp, pk = mp.Parameters(m, n, g0["params"]), g0["pk"]
    t = cards.table(pp, pk)
    ins = [coracle.gen_inputs(cv, m, n, 4300 + b) for b in range(B)]
    cb = 2 * cards.engine.point_bytes
    for latency_batch, bucket_min, toom in ((0, 2048, True), (0, 4, True), (0, 2048, False), (8192, 2048, True)):
        t.set_latency_batch(latency_batch)
        t.set_bucket_min(bucket_min)
        t.set_toom_cook(toom)
        res = cards.shuffle_and_remask_batch([g["prover_seed"] for g in ins], pp, pk, [_split(g["deck"], cb) for g in ins],
                                             [[int.from_bytes(x, "little") for x in _split(g["rho"], 32)] for g in ins],
                                             [mp.Permutation(g["perm"]) for g in ins])
        decks, shufs, proofs = [], [], []
        for g, r in zip(ins, res):
            assert not isinstance(r, Exception), r
            exp_deck, exp_proof = coracle.shuffle_and_remask(cv, m, n, g0["params"], pk, g["deck"], g["rho"], g["perm"], g["prover_seed"])
            assert b"".join(r[0]) == exp_deck and r[1] == exp_proof, (latency_batch, bucket_min, toom)
            decks.append(_split(g["deck"], cb)); shufs.append(r[0]); proofs.append(r[1])
        for merged in (True, False):
            t.set_merged_verify(merged)
            assert cards.verify_shuffle_batch(pp, pk, decks, shufs, proofs) == [None] * B
            out = cards.verify_shuffle_batch(pp, pk, decks, shufs[1:] + shufs[:1], proofs)
            assert all(o == mp.CryptoError("Hadamard Product (5.1)") for o in out)


def test_chain_verification_edge_shapes(mp, coracle):
    """one link, one table; and a chain whose LAST deck is tampered"""
    cv, m, n = "stark", 2, 3
    eng = mp.Engine(cv, device=0)
    g0 = coracle.gen_inputs(cv, m, n, 9)
    t = eng.table(m, n, g0["params"], g0["pk"])
    d1, p1 = coracle.shuffle_and_remask(cv, m, n, g0["params"], g0["pk"], g0["deck"], g0["rho"], g0["perm"], g0["prover_seed"])
    assert t.verify_shuffle_chain(1, 1, g0["deck"] + d1, p1) == [0]
    g1 = coracle.gen_inputs(cv, m, n, 10)
    d2, p2 = coracle.shuffle_and_remask(cv, m, n, g0["params"], g0["pk"], d1, g1["rho"], g1["perm"], g1["prover_seed"])
    assert t.verify_shuffle_chain(1, 2, g0["deck"] + d1 + d2, p1 + p2) == [0, 0]
    st = t.verify_shuffle_chain(1, 2, g0["deck"] + d1 + g1["deck"], p1 + p2)
    assert st[0] == 0 and st[1] > 0
    with pytest.raises(mp.NativeError):
        t.verify_shuffle_chain(1, 2, g0["deck"] + d1, p1 + p2)          # a deck short
