"""-m gpu: the rest of the trait on the GPU, mirroring the reference's six non-shuffle tests
[REF barnett-smart-card-protocol/src/discrete_log_cards/tests.rs:48-173; masking.rs:64-107; remasking.rs:65-114;
reveal.rs:43-84] (same accept / reject behaviour and error names), with byte-for-byte parity against the oracle."""
import pytest

import mp_oracle as po

pytestmark = pytest.mark.gpu

CURVE = "stark"
M, N_ = 4, 13          # the reference's test parameters [REF tests.rs:52-53]


@pytest.fixture(scope="module")
def env(mp):
    cards = mp.DLCards(CURVE, device=0)
    pp = cards.setup(bytes(range(32)), M, N_)
    cv = po.CURVES[CURVE]
    opp = po.setup(cv, M, N_, po.ChaCha20Rng(bytes(range(32))))
    assert pp.raw == po.params_to_bytes(opp)
    return cards, pp, cv, opp


def setup_players(mp, cards, pp, opp, cv, num):
    rng, orng = mp.ChaCha20Rng(b"\x11" * 32), po.ChaCha20Rng(b"\x11" * 32)
    players, agg = [], None
    for _ in range(num):
        pk, sk = cards.player_keygen(rng, pp)
        opk, osk = po.player_keygen(opp, orng)
        assert sk == osk and pk == po.pt_wire(opk)
        info = mp.fr_rand(CURVE, rng).to_bytes(32, "little")
        assert info == po.fe_bytes(po.fr_rand(cv, orng))
        players.append((pk, sk, info))
        agg = po.pt_add(cv, agg, opk)
    return players, po.pt_wire(agg)


def test_generate_and_verify_key(mp, env):
    cards, pp, cv, opp = env
    rng = mp.ChaCha20Rng(b"\x21" * 32)
    pk, sk = cards.player_keygen(rng, pp)
    info = b"player public info"
    proof = cards.prove_key_ownership(b"\x05" * 32, pp, pk, sk, info)
    assert proof == po.sigma_proof_bytes(po.prove_key_ownership(opp, po.pt_from_wire(pk), sk, info, b"\x05" * 32))
    assert cards.verify_key_ownership(pp, pk, info, proof) is None
    wrong_sk = mp.fr_rand(CURVE, rng)
    wrong_proof = cards.prove_key_ownership(b"\x05" * 32, pp, pk, wrong_sk, info)
    with pytest.raises(mp.CryptoError) as e:
        cards.verify_key_ownership(pp, pk, info, wrong_proof)
    assert e.value == mp.CryptoError("Schnorr Identification")


def test_aggregate_keys(mp, env):
    cards, pp, cv, opp = env
    players, expected = setup_players(mp, cards, pp, opp, cv, 10)
    triples = [(pk, cards.prove_key_ownership(bytes([i]) * 32, pp, pk, sk, info), info) for i, (pk, sk, info) in enumerate(players)]
    assert cards.compute_aggregate_key(pp, triples) == expected
    bad = list(triples)
    bad[3] = (bytes(64), bad[3][1], bad[3][2])          # a zeroed key [REF tests.rs:108-115]
    with pytest.raises(mp.CardProtocolError) as e:
        cards.compute_aggregate_key(pp, bad)
    assert e.value == mp.CardProtocolError("ProofVerificationError", mp.CryptoError("Schnorr Identification"))


def test_verify_masking_remasking_reveal_unmask(mp, env):
    cards, pp, cv, opp = env
    players, agg = setup_players(mp, cards, pp, opp, cv, 10)
    oagg = po.pt_from_wire(agg)
    rng = mp.ChaCha20Rng(b"\x31" * 32)
    card_pt = po.pt_mul(cv, mp.fr_rand(CURVE, rng), cv.G)
    card = po.pt_wire(card_pt)
    r = mp.fr_rand(CURVE, rng)
    # test_verify_masking [REF masking.rs:64-107]
    masked, proof = cards.mask(b"\x41" * 32, pp, agg, card, r)
    omasked, oproof = po.mask(opp, oagg, card_pt, r, b"\x41" * 32)
    assert masked == po.deck_to_bytes([omasked]) and proof == po.sigma_proof_bytes(oproof)
    assert cards.verify_mask(pp, agg, card, masked, proof) is None
    wrong_masked = po.deck_to_bytes([(po.pt_mul(cv, 3, cv.G), po.pt_mul(cv, 5, cv.G))])
    with pytest.raises(mp.CryptoError) as e:
        cards.verify_mask(pp, agg, card, wrong_masked, proof)
    assert e.value == mp.CryptoError("Chaum-Pedersen")
    # test_verify_remasking [REF remasking.rs:65-114]
    alpha = mp.fr_rand(CURVE, rng)
    remasked, rproof = cards.remask(b"\x42" * 32, pp, agg, masked, alpha)
    oremasked, orproof = po.remask_with_proof(opp, oagg, omasked, alpha, b"\x42" * 32)
    assert remasked == po.deck_to_bytes([oremasked]) and rproof == po.sigma_proof_bytes(orproof)
    assert cards.verify_remask(pp, agg, masked, remasked, rproof) is None
    with pytest.raises(mp.CryptoError) as e:
        cards.verify_remask(pp, agg, masked, wrong_masked, rproof)
    assert e.value == mp.CryptoError("Chaum-Pedersen")
    # test_verify_reveal [REF reveal.rs:43-84] and test_unmask [REF tests.rs:125-173]
    tokens = []
    for i, (pk, sk, _) in enumerate(players):
        token, tproof = cards.compute_reveal_token(bytes([0x50 + i]) * 32, pp, sk, pk, remasked)
        otoken, otproof = po.compute_reveal_token(opp, sk, po.pt_from_wire(pk), oremasked, bytes([0x50 + i]) * 32)
        assert token == po.pt_wire(otoken) and tproof == po.sigma_proof_bytes(otproof)
        assert cards.verify_reveal(pp, pk, token, remasked, tproof) is None
        tokens.append((token, tproof, pk))
    with pytest.raises(mp.CryptoError) as e:
        cards.verify_reveal(pp, players[0][0], po.pt_wire(po.pt_mul(cv, 99, cv.G)), remasked, tokens[0][1])
    assert e.value == mp.CryptoError("Chaum-Pedersen")
    assert cards.unmask(pp, tokens, remasked) == card
    bad = list(tokens)
    bad[0] = (po.pt_wire(po.pt_mul(cv, 7, cv.G)), bad[0][1], bad[0][2])
    with pytest.raises(mp.CardProtocolError) as e:
        cards.unmask(pp, bad, remasked)
    assert e.value == mp.CardProtocolError("ProofVerificationError", mp.CryptoError("Chaum-Pedersen"))


def test_sigma_batch_parity(mp, env):
    """a batch of 64 Chaum-Pedersen proofs through the C ABI equals the oracle; one tampered proof fails alone"""
    cards, pp, cv, opp = env
    t = cards.table(pp, pp.enc_parameters)
    rng = po.ChaCha20Rng(b"\x61" * 32)
    B = 64
    bases = pubs = wit = seeds = exp = b""
    for i in range(B):
        x = po.fr_rand(cv, rng)
        g, h = po.pt_mul(cv, po.fr_rand(cv, rng), cv.G), po.pt_mul(cv, po.fr_rand(cv, rng), cv.G)
        a = [po.pt_mul(cv, x, g), po.pt_mul(cv, x, h)]
        seed = bytes([i]) * 32
        bases += po.pt_wire(g) + po.pt_wire(h)
        pubs += po.pt_wire(a[0]) + po.pt_wire(a[1])
        wit += po.fe_bytes(x)
        seeds += seed
        exp += po.sigma_proof_bytes(po.sigma_prove(cv, [g, h], a, x, po.REVEAL_RNG_SEED, seed))
    fsi = cards.engine.blake2s(po.REVEAL_RNG_SEED) * B
    got, st = t.sigma_prove_batch(2, bases, pubs, wit, fsi, seeds)
    assert got == exp and st == [0] * B
    assert t.sigma_verify_batch(2, bases, pubs, got, fsi) == [0] * B
    bad = bytearray(got)
    bad[17 * 160 + 159] ^= 1
    st = t.sigma_verify_batch(2, bases, pubs, bytes(bad), fsi)
    assert st[17] == 6 and sum(1 for v in st if v) == 1
