"""-m gpu: SURVEY.md section 8 row f2 on the GPU box.  The arkworks-canonical (de)serialisation of the C ABI
(`mp_*_serialize` / `mp_*_deserialize`, include/mpshuffle.h) is checked against the ORACLE's encoder
(oracle/py/ark_canonical.py -- no code shared with the package), and what it decodes goes through the device:
compressed bytes -> wire v1 -> `mp_verify_shuffle` on the MI355X.
[REF barnett-smart-card-protocol/src/lib.rs:45-71 (every associated type is CanonicalSerialize + CanonicalDeserialize),
 examples/parameter_selection.rs:95 (proof.serialized_size())]"""
import os

import pytest

from conftest import golden_cases, load_json

import ark_canonical as ac
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


def _oracle_objects(g):
    cv = po.CURVES[g["curve"]]
    m, n = g["m"], g["n"]
    with po.curve_ctx(cv):
        w = po.point_bytes()
        pts = [po.pt_from_wire(x) for x in _split(bytes.fromhex(g["params"]), w)]
        pp = po.Params(cv, m, n, pts[0], pts[1:1 + n], pts[1 + n], pts[2 + n])
        deck = po.deck_from_bytes(bytes.fromhex(g["deck"]))
        shuffled = po.deck_from_bytes(bytes.fromhex(g["shuffled"]))
        proof = po.proof_from_bytes(bytes.fromhex(g["proof"]), m, n)
        pk = po.pt_from_wire(bytes.fromhex(g["pk"]))
    return cv, pp, pk, deck, shuffled, proof


@pytest.mark.parametrize("path", golden_cases(), ids=os.path.basename)
def test_golden_vectors_through_serialisation_and_the_device(mp, engines, path):
    g = load_json(path)
    cvn, m, n = g["curve"], g["m"], g["n"]
    cv, pp, pk, deck, shuffled, proof = _oracle_objects(g)
    ser = mp.Serializer(cvn)
    wire = {k: bytes.fromhex(g[k]) for k in ("params", "pk", "deck", "shuffled", "proof")}
    # (b) the C ABI's bytes are the oracle encoder's bytes, element for element
    enc = {"params": ser.params_serialize(m, n, wire["params"]), "deck": ser.deck_serialize(wire["deck"]),
           "shuffled": ser.deck_serialize(wire["shuffled"]), "proof": ser.proof_serialize(m, n, wire["proof"]),
           "pk": ser.points_serialize(wire["pk"])}
    assert enc["params"] == ac.enc_parameters(pp)
    assert enc["deck"] == ac.enc_deck(cv, deck)
    assert enc["shuffled"] == ac.enc_deck(cv, shuffled)
    assert enc["proof"] == ac.enc_proof(cv, proof)
    assert enc["pk"] == ac.enc_point(cv, pk)
    assert len(enc["proof"]) == ser.proof_serialized_size(m, n) == ac.proof_serialized_size(cv, m, n)
    # ... and the oracle decodes them to the objects the wire bytes stand for
    assert ac.dec_deck(cv, enc["shuffled"]) == shuffled
    assert ac.dec_proof(cv, m, n, enc["proof"]) == proof
    # (a) compressed bytes -> C ABI deserialise -> wire v1 -> verification ON THE DEVICE
    m2, n2, params2 = ser.params_deserialize(enc["params"])
    assert (m2, n2, params2) == (m, n, wire["params"])
    deck2, shuf2, proof2 = ser.deck_deserialize(enc["deck"]), ser.deck_deserialize(enc["shuffled"]), ser.proof_deserialize(m, n, enc["proof"])
    pk2 = ser.points_deserialize(enc["pk"])
    assert (deck2, shuf2, proof2, pk2) == (wire["deck"], wire["shuffled"], wire["proof"], wire["pk"])
    cards = engines(cvn)
    P = mp.Parameters(m2, n2, params2)
    cb = 2 * cards.engine.point_bytes
    assert cards.verify_shuffle(P, pk2, _split(deck2, cb), _split(shuf2, cb), proof2) is None
    # the same through the trait-shaped helpers of the mirror (serialize_proof / deserialize_proof)
    assert cards.deserialize_proof(P, cards.serialize_proof(P, proof2)) == proof2
    # a compressed proof whose first commitment has its sign bit flipped still decodes (it is -c_A0, a valid point) and is
    # rejected by the device with the reference's check name
    L = ac.compressed_len(cv)
    bad = bytearray(enc["proof"])
    bad[8 + L - 1] ^= 0x80
    bad_wire = ser.proof_deserialize(m, n, bytes(bad))
    assert bad_wire != proof2
    with pytest.raises(mp.CryptoError) as ei:
        cards.verify_shuffle(P, pk2, _split(deck2, cb), _split(shuf2, cb), bad_wire)
    assert ei.value.check in ("Hadamard Product (5.1)", "Zero Argument (5.2)", "Single Value Product (5.3)", "Multi-Exponentiation Argument (4)")
    # the oracle's verifier names the same check for the same bytes
    with po.curve_ctx(cv):
        code = po.verify_shuffle(pp, pk, deck, shuffled, po.proof_from_bytes(bad_wire, m, n))
    assert code != 0 and po.CHECK_NAMES[code] == ei.value.check


def test_survey_known_answer_through_the_c_abi(mp):
    # SURVEY.md Appendix B/C [CHECKED-HERE with independent big-int code]: a STARK-curve point and its compressed bytes
    x = 0x56a347111c423fb2deff8678925ded9c8ba03b0f577a589cef9f3d3936877c1
    y = 0x2d58166ea0e2c7447739de2ba33a84aa8729176f2ee470c3c5b6526e8cae8c1
    kat = "c1776893d3f3f9ce89a577f5b003bac8d9de258967f8ef2dfb23c41171346a05"
    wire = x.to_bytes(32, "little") + y.to_bytes(32, "little")
    ser = mp.Serializer("stark")
    assert ser.points_serialize(wire).hex() == kat
    assert ser.points_deserialize(bytes.fromhex(kat)) == wire
    assert ac.enc_point(po.STARK, (x, y)).hex() == kat
    assert ac.dec_point(po.STARK, bytes.fromhex(kat)) == (x, y)


def _small_order_point(cv):
    """a point of the curve group whose order divides the cofactor (BLS12-377 G1): q * R for a curve point R"""
    x = 1
    while True:
        x += 1
        yy = po.fq_sqrt(cv, (x * x * x + cv.a * x + cv.b) % cv.p)
        if yy is None:
            continue
        Q = po.pt_mul_raw(cv, cv.q, (x, yy))
        if Q is not None:
            return Q


def test_points_outside_the_prime_order_subgroup_are_rejected(mp, engines):
    cv = po.BLS12_377
    Q = _small_order_point(cv)
    assert cv.is_on_curve(Q) and po.pt_mul_raw(cv, cv.q, Q) is not None
    enc = ac.enc_point(cv, Q)
    with pytest.raises(ac.DecodeError):
        ac.dec_point(cv, enc)
    ser = mp.Serializer("bls12_377")
    with pytest.raises(mp.NativeError) as ei:
        ser.points_deserialize(enc)
    assert ei.value.code == -1                     # MP_ERR_BAD_ENCODING
    # the device's own screen (k_subgroup_check) refuses the same point when it arrives as wire bytes inside a deck
    g = load_json([p for p in golden_cases() if "bls12_377" in p][0])
    m, n = g["m"], g["n"]
    cards = engines("bls12_377")
    P = mp.Parameters(m, n, bytes.fromhex(g["params"]))
    cb = 2 * cards.engine.point_bytes
    with po.curve_ctx(cv):
        qwire = po.pt_wire(Q)
    deck = _split(bytes.fromhex(g["deck"]), cb)
    shuf = _split(bytes.fromhex(g["shuffled"]), cb)
    shuf[0] = qwire + shuf[0][len(qwire):]
    with pytest.raises((mp.CardProtocolError, mp.NativeError)):
        cards.verify_shuffle(P, bytes.fromhex(g["pk"]), deck, shuf, bytes.fromhex(g["proof"]))
    # (the device tests phi(P) = -[u^2]P: also a subgroup point plus that low-order point, in every position of a batch)
    with po.curve_ctx(cv):
        mixed = po.pt_wire(po.pt_add(cv, Q, po.pt_mul(cv, 12345, cv.G)))
    t = cards.table(P, bytes.fromhex(g["pk"]))
    good_d, good_s, pf = bytes.fromhex(g["deck"]), bytes.fromhex(g["shuffled"]), bytes.fromhex(g["proof"])
    bad_s = mixed + good_s[len(mixed):]
    assert t.verify_shuffle_batch(good_d * 5, good_s + bad_s + good_s * 2 + bad_s, pf * 5) == [0, -1, 0, 0, -1]


def test_malformed_encodings(mp):
    for cvn in ("stark", "bn254", "secp256k1", "bls12_377"):
        cv = po.CURVES[cvn]
        ser = mp.Serializer(cvn)
        L = ac.compressed_len(cv)
        inf = ac.enc_point(cv, None)
        with po.curve_ctx(cv):
            assert ser.points_deserialize(inf) == bytes(po.point_bytes())
            assert ser.points_serialize(bytes(po.point_bytes())) == inf
        # x >= p
        big = bytearray((cv.p).to_bytes(L, "little"))
        for bad in (bytes(big), bytes(inf[:-1]) + bytes([inf[-1] | 0x80]), bytes([1]) + inf[1:]):
            with pytest.raises(ac.DecodeError):
                ac.dec_point(cv, bad)
            with pytest.raises(mp.NativeError):
                ser.points_deserialize(bad)


def test_decks_validated_once_are_not_validated_again(mp, engines):
    """round 6: mp_deck_validate_dev is the once-per-deck validation -- it refuses a deck with a point outside the
    prime-order subgroup, as k_subgroup_check inside a verify call does --, and a table told that its decks are validated
    (mp_set_validated) does not test them again: the same bad deck then reaches the equations (and fails THEM: the statement is wrong),
    while the proofs' points are still tested.  Honest inputs give 0 in every mode."""
    import torch
    cv = po.BLS12_377
    Q = _small_order_point(cv)
    g = load_json([p for p in golden_cases() if "bls12_377" in p][0])
    m, n = g["m"], g["n"]
    cards = engines("bls12_377")
    P = mp.Parameters(m, n, bytes.fromhex(g["params"]))
    t = cards.table(P, bytes.fromhex(g["pk"]))
    with po.curve_ctx(cv):
        mixed = po.pt_wire(po.pt_add(cv, Q, po.pt_mul(cv, 12345, cv.G)))
    good_d, good_s, pf = bytes.fromhex(g["deck"]), bytes.fromhex(g["shuffled"]), bytes.fromhex(g["proof"])
    bad_s = mixed + good_s[len(mixed):]
    gpu = torch.device("cuda", 0)
    dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(gpu)
    decks = dev(good_s + bad_s + good_s)
    st = torch.full((3,), 55, dtype=torch.int32, device=gpu)
    t.deck_validate_dev(3, decks.data_ptr(), st.data_ptr())
    assert st.cpu().tolist() == [0, -1, 0]
    assert t.verify_shuffle_batch(good_d * 3, good_s + bad_s + good_s, pf * 3) == [0, -1, 0]
    t.set_validated(t.VALIDATED_DECKS | t.VALIDATED_SHUFFLED)
    got = t.verify_shuffle_batch(good_d * 3, good_s + bad_s + good_s, pf * 3)
    assert got[0] == 0 and got[2] == 0 and got[1] > 0          # not refused as an encoding error any more: an equation fails instead
    # the proof's points are still tested: an off-subgroup point in the proof is an encoding error
    pb = cards.engine.point_bytes
    bad_pf = mixed + pf[pb:]
    assert t.verify_shuffle_batch(good_d, good_s, bad_pf) == [-1]
    t.set_validated(t.VALIDATED_DECKS | t.VALIDATED_SHUFFLED | t.VALIDATED_PROOFS)
    assert t.verify_shuffle_batch(good_d, good_s, bad_pf)[0] > 0
    t.set_validated(0)
    assert t.verify_shuffle_batch(good_d, good_s, bad_pf) == [-1]
    with pytest.raises(mp.NativeError):
        t.set_validated(8)
