"""CPU tests (-m "not gpu"): pin the oracles.

The reference holds no golden vectors and cannot be run here, so the pins are:
 * published constants / KATs: secp256k1 2G, RFC 7539 ChaCha20 block, the ChaCha20 zero-key keystream,
   SURVEY App. C values (checked there with independent arithmetic);
 * the committed fixtures in tests/golden (made by tests/golden/gen_golden.py from the Python oracle);
 * agreement of two independent restatements (Python big-int vs C++ 4x64 Montgomery + Pippenger);
 * the reference's behavioural pins: accept honest / reject a wrong deck as "Hadamard Product (5.1)"
   [REF barnett-smart-card-protocol/src/discrete_log_cards/tests.rs:175-227].
"""
import copy
import hashlib
import os
import struct

import pytest

from conftest import GOLDEN, golden_cases, load_json

import mp_oracle as po

K = 0x0123456789abcdef0fedcba9876543210123456789abcdef0fedcba987654321


def test_curve_constants_and_kats_python():
    for cv in po.CURVES.values():
        assert cv.is_on_curve(cv.G)
        assert po.pt_mul(cv, cv.q, cv.G) is None
        assert po.pt_mul(cv, cv.q - 1, cv.G) == po.pt_neg(cv, cv.G)
    # universally published secp256k1 2G
    assert po.pt_mul(po.SECP256K1, 2, po.SECP256K1.G) == (
        0xc6047f9441ed7d6d3045406e95c07cd85c778e4b8cef3ca7abac09b95c709ee5,
        0x1ae168fea63dc339a3c58419466ceaeef7f632653266d0e1236431a950cfe52a)
    # SURVEY App. C
    assert po.pt_mul(po.STARK, K, po.STARK.G) == (
        0x56a347111c423fb2deff8678925ded9c8ba03b0f577a589cef9f3d3936877c1,
        0x2d58166ea0e2c7447739de2ba33a84aa8729176f2ee470c3c5b6526e8cae8c1)
    assert po.pt_mul(po.BN254, K, po.BN254.G) == (
        0x15b653a46cb336794237e1cdcf6cf7e899315229adf895acb32cb7013887531,
        0x28506bd1fd0c2ffafeab89b1fe80056432a5977ab1f03cea6568591ca9576d15)


def test_hash_kats_python():
    key = struct.unpack("<8I", bytes(range(32)))
    out = po.chacha20_block(key, 1, w13=0x09000000, w14=0x4a000000, w15=0)  # RFC 7539 2.3.2
    assert out[:4] == [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3]
    assert out[12:] == [0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]
    assert po.blake2s(b"Shuffle Proof").hex() == "99df86eeefd21867b5ea2a0194c5e8dd819aa01221dcdcbc5ff6b16f9303b656"


def test_curve_kats_c(coracle):
    kats = load_json(os.path.join(GOLDEN, "curve_kats.json"))
    for name, k in kats.items():
        G = bytes.fromhex(k["G"])
        assert coracle.on_curve(name, G) == 1
        one = (1).to_bytes(32, "little")
        two = (2).to_bytes(32, "little")
        for algo in (0, 1):
            assert coracle.msm(name, two, G, algo).hex() == k["twoG"]
            assert coracle.msm(name, bytes.fromhex(k["k"]), G, algo).hex() == k["kG"]
            assert coracle.msm(name, one + one, G + G, algo).hex() == k["twoG"]
        qm1 = (int(k["q"], 16) - 1).to_bytes(32, "little")
        assert coracle.msm(name, qm1, G, 0).hex() == k["qm1G"]
        # G + (q-1)G = infinity (all-zero wire encoding)
        assert coracle.msm(name, one + qm1, G + G, 0) == bytes(coracle.point_size(name))
        r = k["remask"]
        out = coracle.remask_deck(name, G, bytes.fromhex(r["pk"]), bytes.fromhex(r["ct"]), bytes.fromhex(r["alpha"]))
        assert out.hex() == r["out"]


def test_fs_kats_c(coracle):
    kats = load_json(os.path.join(GOLDEN, "fs_kats.json"))
    assert coracle.blake2s(b"Shuffle Proof").hex() == kats["blake2s_shuffle_proof"]
    for n in (0, 1, 63, 64, 65, 127, 128, 129, 1000):
        data = bytes((i * 7 + 3) & 0xFF for i in range(n))
        assert coracle.blake2s(data) == hashlib.blake2s(data).digest()
    ks = b"".join(struct.pack("<16I", *coracle.chacha20_block(bytes(32), c)) for c in range(1))
    assert ks.hex() == kats["chacha20_zero_key_first64"]
    for name in po.CURVES:
        k = kats["challenges_" + name]
        assert [hex(v) for v in coracle.fs_challenges(name, b"Shuffle Proof", None, 3)] == k["after_seed"]
        assert [hex(v) for v in coracle.fs_challenges(name, b"Shuffle Proof", bytes(range(200)), 3)] == k["after_absorb_0_199"]


@pytest.mark.parametrize("path", golden_cases(), ids=os.path.basename)
def test_c_oracle_matches_golden(coracle, path):
    g = load_json(path)
    cv, m, n = g["curve"], g["m"], g["n"]
    gi = coracle.gen_inputs(cv, m, n, g["seed"])
    for key in ("params", "pk", "deck", "rho", "prover_seed"):
        assert gi[key].hex() == g[key], key
    assert gi["perm"] == g["perm"]
    sh, pf = coracle.shuffle_and_remask(cv, m, n, **gi)
    assert sh.hex() == g["shuffled"]
    assert pf.hex() == g["proof"]
    with po.curve_ctx(po.CURVES[cv]):
        assert len(pf) == coracle.proof_size(m, n, cv) == po.proof_size(m, n)
    assert coracle.verify_shuffle(cv, m, n, gi["params"], gi["pk"], gi["deck"], sh, pf) == 0
    # [REF tests.rs:213-226] a random wrong deck is rejected by name
    wrong = coracle.gen_inputs(cv, m, n, g["seed"] + 1000)["deck"]
    rc = coracle.verify_shuffle(cv, m, n, gi["params"], gi["pk"], gi["deck"], wrong, pf)
    assert coracle.CHECK_NAMES[rc] == "Hadamard Product (5.1)"


def test_c_oracle_matches_the_chain_fixture(coracle):
    """chain_stark_m2_n3_L3_s21.json: three dependent shuffles of one table under one key [REF examples/round.rs:268-350] -- the C++ restatement
    reproduces every deck and proof of the Python oracle's chain and verifies every link"""
    g = load_json(os.path.join(GOLDEN, "chain_stark_m2_n3_L3_s21.json"))
    cv, m, n = g["curve"], g["m"], g["n"]
    params, pk = bytes.fromhex(g["params"]), bytes.fromhex(g["pk"])
    assert len(g["decks"]) == g["links"] + 1 == len(g["chain"]) + 1
    for j, link in enumerate(g["chain"]):
        deck = bytes.fromhex(g["decks"][j])
        sh, pf = coracle.shuffle_and_remask(cv, m, n, params, pk, deck, bytes.fromhex(link["rho"]), link["perm"], bytes.fromhex(link["prover_seed"]))
        assert sh.hex() == g["decks"][j + 1] and pf.hex() == link["proof"], j
        assert coracle.verify_shuffle(cv, m, n, params, pk, deck, sh, pf) == 0
        if j:       # a link verified against the wrong input deck (the chain's first) is rejected by name [REF tests.rs:213-226]
            assert coracle.CHECK_NAMES[coracle.verify_shuffle(cv, m, n, params, pk, bytes.fromhex(g["decks"][0]), sh, pf)] == "Hadamard Product (5.1)"


@pytest.mark.parametrize("name", ["shuffle_stark_m2_n3_s1.json", "shuffle_bn254_m2_n4_s3.json",
                                  "shuffle_secp256k1_m3_n3_s5.json", "shuffle_bls12_377_m2_n3_s13.json"])
def test_python_oracle_matches_golden(name):
    g = load_json(os.path.join(GOLDEN, name))
    cv = po.CURVES[g["curve"]]
    pp, pk, deck, rho, perm, ps = po.gen_inputs(cv, g["m"], g["n"], g["seed"])
    assert po.params_to_bytes(pp).hex() == g["params"]
    sh, pf = po.shuffle_and_remask(pp, pk, deck, rho, perm, ps)
    with po.curve_ctx(cv):
        assert po.deck_to_bytes(sh).hex() == g["shuffled"]
        assert po.proof_to_bytes(pf).hex() == g["proof"]
    assert po.verify_shuffle(pp, pk, deck, sh, pf) == 0


def test_tampering_names_the_failing_check(coracle):
    g = load_json(os.path.join(GOLDEN, "shuffle_stark_m3_n4_s11.json"))
    cv, m, n = g["curve"], g["m"], g["n"]
    b = {k: bytes.fromhex(g[k]) for k in ("params", "pk", "deck", "shuffled", "proof")}
    pf = po.proof_from_bytes(b["proof"], m, n)
    q = po.CURVES[cv].q

    def check(mut):
        p2 = copy.deepcopy(pf)
        mut(p2)
        return coracle.CHECK_NAMES[coracle.verify_shuffle(cv, m, n, b["params"], b["pk"], b["deck"], b["shuffled"],
                                                          po.proof_to_bytes(p2))]

    def bump(d, k, i=None):
        if i is None:
            d[k] = (d[k] + 1) % q
        else:
            d[k][i] = (d[k][i] + 1) % q

    assert check(lambda p: None) == "Ok"
    assert check(lambda p: bump(p["mexp"], "taubar")) == "Multi-Exponentiation Argument (4)"
    assert check(lambda p: bump(p["mexp"], "abar", 1)) == "Multi-Exponentiation Argument (4)"
    assert check(lambda p: bump(p["product"]["svp"], "rt")) == "Single Value Product (5.3)"
    assert check(lambda p: bump(p["product"]["svp"], "bt", 0)) == "Single Value Product (5.3)"
    assert check(lambda p: bump(p["product"]["had"]["zero"], "tbar")) == "Zero Argument (5.2)"
    assert check(lambda p: bump(p["product"]["had"]["zero"], "abar", 2)) == "Zero Argument (5.2)"
    # swapping two shuffled cards breaks the statement hash -> first check to fail is Hadamard's
    sh = bytearray(b["shuffled"])
    sh[0:128], sh[128:256] = sh[128:256], sh[0:128]
    rc = coracle.verify_shuffle(cv, m, n, b["params"], b["pk"], b["deck"], bytes(sh), b["proof"])
    assert coracle.CHECK_NAMES[rc] == "Hadamard Product (5.1)"


def test_edge_inputs_c_vs_python(coracle):
    """identity permutation, rho in {0, 1, q-1}, duplicate cards, an infinity component."""
    cvn, m, n = "stark", 2, 3
    cv = po.CURVES[cvn]
    pp, pk, deck, rho, perm, ps = po.gen_inputs(cv, m, n, 42)
    deck[1] = deck[0]                       # duplicate card
    deck[2] = (None, deck[2][1])            # infinity component
    rho = [0, 1, cv.q - 1, rho[3], rho[4], 0]
    perm = list(range(m * n))               # identity permutation
    sh, pf = po.shuffle_and_remask(pp, pk, deck, rho, perm, ps)
    assert po.verify_shuffle(pp, pk, deck, sh, pf) == 0
    csh, cpf = coracle.shuffle_and_remask(cvn, m, n, po.params_to_bytes(pp), po.pt_wire(pk), po.deck_to_bytes(deck),
                                          b"".join(po.fe_bytes(r) for r in rho), perm, ps)
    assert csh == po.deck_to_bytes(sh)
    assert cpf == po.proof_to_bytes(pf)
    assert coracle.verify_shuffle(cvn, m, n, po.params_to_bytes(pp), po.pt_wire(pk), po.deck_to_bytes(deck), csh, cpf) == 0


def test_c_oracle_rejects_bad_usage(coracle):
    g = load_json(os.path.join(GOLDEN, "shuffle_stark_m2_n3_s1.json"))
    gi = coracle.gen_inputs("stark", 2, 3, 1)
    bad = dict(gi)
    bad["perm"] = [0, 0, 1, 2, 3, 4]  # not a permutation
    with pytest.raises(ValueError):
        coracle.shuffle_and_remask("stark", 2, 3, **bad)
    # non-canonical scalar in the proof (>= q) is a usage error, not a verification failure
    pf = bytearray(bytes.fromhex(g["proof"]))
    pf[-32:] = b"\xff" * 32
    assert coracle.verify_shuffle("stark", 2, 3, gi["params"], gi["pk"], gi["deck"], bytes.fromhex(g["shuffled"]), bytes(pf)) < 0


def test_sigma_oracles_agree(coracle):
    """the C++ and the Python restatements of the sigma protocols agree byte for byte, on all curves"""
    for cvn in ("stark", "bn254", "secp256k1"):
        cv = po.CURVES[cvn]
        rng = po.ChaCha20Rng(bytes(range(32)))
        for nb, fs_init in ((1, po.KEY_OWN_RNG_SEED + b"player"), (2, po.REVEAL_RNG_SEED)):
            x = po.fr_rand(cv, rng)
            g = [po.pt_mul(cv, po.fr_rand(cv, rng), cv.G) for _ in range(nb)]
            a = [po.pt_mul(cv, x, gi) for gi in g]
            seed = bytes([nb]) * 32
            exp = po.sigma_proof_bytes(po.sigma_prove(cv, g, a, x, fs_init, seed))
            gb, ab = b"".join(po.pt_wire(p) for p in g), b"".join(po.pt_wire(p) for p in a)
            got = coracle.sigma_prove(cvn, nb, gb, ab, po.fe_bytes(x), fs_init, seed)
            assert got == exp
            assert coracle.sigma_verify(cvn, nb, gb, ab, got, fs_init) == 0
            bad = bytearray(got)
            bad[-1] ^= 1
            assert coracle.CHECK_NAMES_ALL[coracle.sigma_verify(cvn, nb, gb, ab, bytes(bad), fs_init)] == po.SIGMA_NAMES[nb]
            assert coracle.sigma_verify(cvn, nb, gb, ab, got, fs_init + b"x") != 0


# ---- published vectors for the two primitives under the transcript (RFC 7693 BLAKE2s, RFC 7539 / rand_chacha ChaCha20) -----------
RFC7693_ABC = "508c5e8c327c14e2e1a72ba34eeb452f37458b209ed63a294d999b4c86675982"        # RFC 7693 Appendix B: BLAKE2s-256("abc")
BLAKE2S_EMPTY = "69217a3079908094e11121d042354a7c1f55b6482ca1a51e1b250dfd1ed0eef9"      # BLAKE2s-256("")
# rand_chacha `test_chacha_true_values_a` (ChaCha20Rng::from_seed([0; 32]), the first 32 next_u32 results) = RFC 7539 A.1 vectors
# #1 and #2: the zero-key keystream blocks 0 and 1
CHACHA20_ZERO_SEED_U32 = [
    0xade0b876, 0x903df1a0, 0xe56a5d40, 0x28bd8653, 0xb819d2bd, 0x1aed8da0, 0xccef36a8, 0xc70d778b,
    0x7c5941da, 0x8d485751, 0x3fe02477, 0x374ad8b8, 0xf4b8436a, 0x1ca11815, 0x69b687c3, 0x8665eeb2,
    0xbee7079f, 0x7a385155, 0x7c97ba98, 0x0d082d73, 0xa0290fcb, 0x6965e348, 0x3e53c612, 0xed7aee32,
    0x7621b729, 0x434ee69c, 0xb03371d5, 0xd539d874, 0x281fed31, 0x45fb0a51, 0x1f0ae1ac, 0x6f4d794b]


def test_published_blake2s_vectors(coracle, mp):
    for msg, exp in ((b"abc", RFC7693_ABC), (b"", BLAKE2S_EMPTY)):
        assert po.blake2s(msg).hex() == exp
        assert coracle.blake2s(msg).hex() == exp
    # multi-block input: both restatements against hashlib (which carries the reference implementation of RFC 7693)
    import hashlib
    for n in (63, 64, 65, 127, 128, 129, 1000):
        msg = bytes((7 * i + 3) & 0xFF for i in range(n))
        assert coracle.blake2s(msg) == hashlib.blake2s(msg).digest() == po.blake2s(msg)
    # the engine's own host helper (include/mpshuffle.h mp_blake2s): callable without a device
    import ctypes
    lib = mp.load()
    for msg, exp in ((b"abc", RFC7693_ABC), (b"Shuffle Proof", "99df86eeefd21867b5ea2a0194c5e8dd819aa01221dcdcbc5ff6b16f9303b656")):
        out = (ctypes.c_uint8 * 32)()
        assert lib.mp_blake2s((ctypes.c_uint8 * len(msg)).from_buffer_copy(msg), len(msg), out) == 0
        assert bytes(out).hex() == exp


def test_published_chacha20rng_vectors(coracle, mp):
    r = po.ChaCha20Rng(bytes(32))
    assert [r.next_u32() for _ in range(32)] == CHACHA20_ZERO_SEED_U32
    r = po.ChaCha20Rng(bytes(32))
    w = CHACHA20_ZERO_SEED_U32
    assert [r.next_u64() for _ in range(16)] == [w[2 * i] | (w[2 * i + 1] << 32) for i in range(16)]   # BlockRng: low word first
    assert coracle.chacha20_block(bytes(32), 0) == w[:16] and coracle.chacha20_block(bytes(32), 1) == w[16:]
    m = mp.ChaCha20Rng(bytes(32))                    # the host mirror used by protocol.py
    assert [m.next_u64() for _ in range(16)] == [w[2 * i] | (w[2 * i + 1] << 32) for i in range(16)]


def test_sigma_nonce_is_hedged(coracle):
    """the same prover seed gives the same proof only for the same witness AND statement: two proofs for one secret under a reused
    seed (different statements) have different commitments, so z1 - z2 no longer reveals the secret"""
    cv = po.STARK
    with po.curve_ctx(cv):
        seed, x = bytes(range(32)), 0x1234567
        g1, g2 = cv.G, po.pt_mul(cv, 5, cv.G)
        A1, z1 = po.sigma_prove(cv, [g1], [po.pt_mul(cv, x, g1)], x, b"Masking Proof", seed)
        A1b, z1b = po.sigma_prove(cv, [g1], [po.pt_mul(cv, x, g1)], x, b"Masking Proof", seed)
        A2, z2 = po.sigma_prove(cv, [g2], [po.pt_mul(cv, x, g2)], x, b"Masking Proof", seed)
        A3, z3 = po.sigma_prove(cv, [g1], [po.pt_mul(cv, x, g1)], x, b"Reveal Proof", seed)
        assert (A1, z1) == (A1b, z1b)
        r1 = po.pt_mul(cv, pow(5, -1, cv.q), A2[0])          # A2 = r2 * 5G  ->  r2 * G
        assert r1 != A1[0] and A3[0] != A1[0]
