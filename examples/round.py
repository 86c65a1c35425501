#!/usr/bin/env python3
"""A round of cards on the MI355X engine -- the flow of the reference's example
[REF barnett-smart-card-protocol/examples/round.rs:228-436]: 4 players, 52-card deck (m=2, n=26), every player shuffles
and re-masks the deck in turn with a proof that everyone verifies, cards are dealt, peeked at privately and opened.
Everything cryptographic runs in libmpshuffle.so (HIP); this script is plumbing (BASELINE config 1)."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mp = importlib.import_module("mental-poker_amd")

SUITS, VALUES = ["Club", "Diamond", "Heart", "Spade"], ["2", "3", "4", "5", "6", "7", "8", "9", "10", "J", "Q", "K", "A"]


def main():
    m, n, num_cards = 2, 26, 52
    cards = mp.DLCards("stark", device=0)
    rng = mp.ChaCha20Rng(b"round example seed".ljust(32, b"\0"))
    fresh = lambda: b"".join(rng.next_u64().to_bytes(8, "little") for _ in range(4))     # noqa: E731
    pp = cards.setup(fresh(), m, n)

    # players: key pair + proof of key ownership [REF round.rs:141-152]
    names = [b"Andrija", b"Kobi", b"Nico", b"Tom"]
    players = []
    for name in names:
        pk, sk = cards.player_keygen(rng, pp)
        players.append(dict(name=name, pk=pk, sk=sk, proof=cards.prove_key_ownership(fresh(), pp, pk, sk, name)))
    joint_pk = cards.compute_aggregate_key(pp, [(p["pk"], p["proof"], p["name"]) for p in players])

    # open deck: 52 distinct points <-> classic cards, masked with r = 1 [REF round.rs:253-256]
    card_points = [pp.raw[64 * (1 + i):64 * (2 + i)] for i in range(n)]          # any 52 distinct group elements will do:
    extra = cards.setup(fresh(), m, num_cards)                                      # take them from a second setup
    card_points = [extra.raw[64 * i:64 * (i + 1)] for i in range(num_cards)]
    mapping = {pt: "%s of %ss" % (v, s) for pt, (s, v) in zip(card_points, [(s, v) for s in SUITS for v in VALUES])}
    deck = []
    for pt in card_points:
        masked, proof = cards.mask(fresh(), pp, joint_pk, pt, 1)
        cards.verify_mask(pp, joint_pk, pt, masked, proof)
        deck.append(masked)

    # every player shuffles + remasks; everyone verifies [REF round.rs:263-350]
    t0 = time.time()
    for p in players:
        perm = mp.Permutation.new(rng, num_cards)
        factors = [mp.fr_rand("stark", rng) for _ in range(num_cards)]
        shuffled, proof = cards.shuffle_and_remask(fresh(), pp, joint_pk, deck, factors, perm)
        cards.verify_shuffle(pp, joint_pk, deck, shuffled, proof)
        deck = shuffled
    print("4 shuffles proved and verified in %.2f s (batch of one each: latency, not throughput)" % (time.time() - t0))

    # deal one card each; everyone else publishes reveal tokens; the owner peeks [REF round.rs:355-385]
    dealt = deck[:4]
    for i, p in enumerate(players):
        tokens = []
        for q in players:
            tok, pf = cards.compute_reveal_token(fresh(), pp, q["sk"], q["pk"], dealt[i])
            tokens.append((tok, pf, q["pk"]))
        opened = cards.unmask(pp, tokens, dealt[i])
        print("%s holds the %s" % (p["name"].decode(), mapping[opened]))
    print("round ok")


if __name__ == "__main__":
    main()
