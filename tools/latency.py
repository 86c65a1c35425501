#!/usr/bin/env python3
"""single-proof latency of shuffle_and_remask + verify_shuffle (host-buffer API, B = 1) with the latency plan and with the
throughput plan forced -- reported in DESIGN.md."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
mp = importlib.import_module("mental-poker_amd")
import coracle as co
m, n = 2, 26
g = co.gen_inputs("stark", m, n, 7)
eng = mp.Engine("stark", 0)
t = eng.table(m, n, g["params"], g["pk"], fb_bits=16)
for name, lb, bm, lanes in (("finest split, Straus only (the default for 52 cards since round 5)", 8192, 2048, 0), ("finest split, one-lane transcripts", 8192, 2048, 1),
                            ("finest split, bucket kernel from 128 terms (the default of rounds 2-4)", 8192, 128, 0),
                            ("throughput plan", 0, 2048, 0)):
    t.set_latency_batch(lb)
    t.set_bucket_min(bm)
    t.set_transcript_lanes(lanes)
    for B in (1, 4, 64):
        decks, rho, perm, seeds = g["deck"] * B, g["rho"] * B, g["perm"] * B, g["prover_seed"] * B
        d, p, st = t.shuffle_and_remask_batch(decks, rho, perm, seeds)
        t0 = time.perf_counter(); K = 5
        for _ in range(K):
            d, p, st = t.shuffle_and_remask_batch(decks, rho, perm, seeds)
        tp = (time.perf_counter() - t0) / K
        t0 = time.perf_counter()
        for _ in range(K):
            sv = t.verify_shuffle_batch(decks, d, p)
        tv = (time.perf_counter() - t0) / K
        assert not any(st) and not any(sv)
        print("%-80s B=%3d  prove %.1f ms  verify %.1f ms" % (name, B, 1e3 * tp, 1e3 * tv))
