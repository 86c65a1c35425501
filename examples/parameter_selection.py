#!/usr/bin/env python3
"""How prover time and proof size depend on the split N = m x n -- the reference's only benchmark harness
[REF barnett-smart-card-protocol/examples/parameter_selection.rs:25-96] on the MI355X engine: BLS12-377 G1, a 300-card
deck, (m, n) in {(2,150), (6,50), (10,30), (12,25), (30,10)}; the same deck, shared key, blinding factors and permutation
for every pair, fresh parameters per pair [REF :33-57, :80], prover time around shuffle_and_remask [REF :82-93] and
`proof.serialized_size()` [REF :95] (arkworks-canonical compressed points: mental-poker_amd/canonical.py).

Printed per pair: the GPU prover's latency for ONE proof (what the reference times), its throughput with `--batch` proofs in
flight, the size of the proof, and -- with --cpu -- the single-threaded CPU port (oracle/, test infrastructure) on the
same inputs beside it.  SURVEY.md section 8 row f3."""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mp = importlib.import_module("mental-poker_amd")

NUMBER_OF_CARDS = 300
PAIRS = [(2, 150), (6, 50), (10, 30), (12, 25), (30, 10)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="bls12_377")
    ap.add_argument("--batch", type=int, default=256, help="proofs in flight for the throughput column")
    ap.add_argument("--cpu", action="store_true", help="also time the CPU port (oracle) on the same inputs")
    ap.add_argument("--subgroup-check", action="store_true",
                    help="keep the engine's per-call subgroup test of every wire point on (BLS12-377 has a cofactor).  The reference times "
                         "shuffle_and_remask on values that arkworks validated when they were deserialised, i.e. outside its timed region; "
                         "here the inputs come from the engine's own setup, so the harness switches the test off like a caller with validated points would")
    ap.add_argument("--karatsuba", action="store_true", help="3 <= m <= 8: recursive Karatsuba also in the throughput plans (the small-batch plans always use it)")
    args = ap.parse_args()
    cards = mp.DLCards(args.curve, device=0)
    can = mp.canonical
    cb = 2 * cards.engine.point_bytes
    rng = mp.ChaCha20Rng(b"parameter selection".ljust(32, b"\0"))
    fresh = lambda: b"".join(rng.next_u64().to_bytes(8, "little") for _ in range(4))     # noqa: E731

    # deck, shared key, blinding factors, permutation: sampled once [REF :36-39].  Random group elements = k * G.
    pp0 = cards.setup(fresh(), 2, NUMBER_OF_CARDS)             # 300 + 3 random points from the engine's own setup
    pts = [pp0.raw[cards.engine.point_bytes * i:cards.engine.point_bytes * (i + 1)] for i in range(NUMBER_OF_CARDS + 3)]
    shared_key = pts[NUMBER_OF_CARDS]
    pp1 = cards.setup(fresh(), 2, NUMBER_OF_CARDS)
    pts1 = [pp1.raw[cards.engine.point_bytes * i:cards.engine.point_bytes * (i + 1)] for i in range(NUMBER_OF_CARDS)]
    deck = [pts[i] + pts1[i] for i in range(NUMBER_OF_CARDS)]
    factors = [mp.fr_rand(args.curve, rng) for _ in range(NUMBER_OF_CARDS)]
    permutation = mp.Permutation.new(rng, NUMBER_OF_CARDS)

    print("%d cards on %s, one MI355X; proof size = serialized_size() with compressed points" % (NUMBER_OF_CARDS, args.curve))
    print("%4s %4s | %12s %12s %14s | %10s %10s%s" % ("m", "n", "prove 1 (ms)", "verify 1 (ms)", "prove/s @B=%d" % args.batch,
                                                    "proof (B)", "wire (B)", " | CPU prove (s)" if args.cpu else ""))
    for m, n in PAIRS:
        pp = cards.setup(fresh(), m, n)
        t = cards.table(pp, shared_key)                                                      # builds the table
        t.set_subgroup_check(args.subgroup_check)
        t.set_toom_cook(not args.karatsuba)
        cards.shuffle_and_remask(fresh(), pp, shared_key, deck, factors, permutation)        # warms up
        seed = fresh()
        t0 = time.perf_counter()
        shuffled, proof = cards.shuffle_and_remask(seed, pp, shared_key, deck, factors, permutation)
        t1 = time.perf_counter()
        assert cards.verify_shuffle(pp, shared_key, deck, shuffled, proof) is None
        t2 = time.perf_counter()
        B = args.batch
        seeds = [fresh() for _ in range(B)]
        t3 = time.perf_counter()
        res = cards.shuffle_and_remask_batch(seeds, pp, shared_key, [deck] * B, [factors] * B, [permutation] * B)
        t4 = time.perf_counter()
        assert not any(isinstance(r, Exception) for r in res)
        size = cards.proof_serialized_size(pp)
        assert len(cards.serialize_proof(pp, proof)) == size
        line = "%4d %4d | %12.1f %12.1f %14.0f | %10d %10d" % (m, n, 1e3 * (t1 - t0), 1e3 * (t2 - t1), B / (t4 - t3), size, len(proof))
        if args.cpu:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import coracle
            rho = b"".join(int(f).to_bytes(32, "little") for f in factors)
            c0 = time.perf_counter()
            exp_deck, exp_proof = coracle.shuffle_and_remask(args.curve, m, n, pp.raw, shared_key, b"".join(deck), rho,
                                                             permutation.mapping, seed)
            c1 = time.perf_counter()
            assert exp_proof == proof and exp_deck == b"".join(shuffled), "GPU and CPU port disagree"
            line += " | %13.2f" % (c1 - c0)
        print(line)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
