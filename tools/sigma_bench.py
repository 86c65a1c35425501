#!/usr/bin/env python3
"""Throughput of the batched sigma protocols behind the rest of the trait: Chaum-Pedersen proofs of
discrete-log equality (mask / remask / reveal) and Schnorr identification (key ownership) through mp_sigma_*_batch
(host-buffer API), beside the single-threaded C++ oracle on the same statements.  Reported in DESIGN.md."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
mp = importlib.import_module("mental-poker_amd")
import coracle as co  # noqa: E402

curve, m, n = "stark", 2, 26
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
eng = mp.Engine(curve, 0)
params = eng.setup(m, n, bytes([1] * 32))
t = eng.table(m, n, params, params[64:128])
pts = eng.setup(m, 2 * 64, bytes([5] * 32))           # 131 random points
P = [pts[64 * i:64 * i + 64] for i in range(128)]
rng = mp.ChaCha20Rng(bytes([6] * 32))
fs = eng.blake2s(b"Reveal Proof")
for nb, name in ((2, "Chaum-Pedersen"), (1, "Schnorr")):
    xs = [mp.fr_rand(curve, rng) for _ in range(64)]
    # 64 distinct statements, tiled over the batch: publics_i = x * bases_i computed with the engine's own MSM
    bases = [b"".join(P[(2 * k + i) % 128] for i in range(nb)) for k in range(64)]
    pubs = []
    for k in range(64):
        xb = xs[k].to_bytes(32, "little")
        pubs.append(b"".join(t.msm(1, 1, xb, bases[k][64 * i:64 * i + 64]) for i in range(nb)))
    rep = B // 64
    gb, gp = b"".join(bases) * rep, b"".join(pubs) * rep
    wit = b"".join(x.to_bytes(32, "little") for x in xs) * rep
    seeds = bytes(range(32)) * B
    fsb = fs * B
    proofs, st = t.sigma_prove_batch(nb, gb, gp, wit, fsb, seeds)
    t0 = time.perf_counter()
    proofs, st = t.sigma_prove_batch(nb, gb, gp, wit, fsb, seeds)
    tp = time.perf_counter() - t0
    sv = t.sigma_verify_batch(nb, gb, gp, proofs, fsb)
    t0 = time.perf_counter()
    sv = t.sigma_verify_batch(nb, gb, gp, proofs, fsb)
    tv = time.perf_counter() - t0
    assert not any(st) and not any(sv)
    # CPU oracle on the first 64 statements
    psz = nb * 64 + 32
    t0 = time.perf_counter()
    for k in range(64):
        exp = co.sigma_prove(curve, nb, bases[k], pubs[k], xs[k].to_bytes(32, "little"), b"Reveal Proof", bytes(range(32)))
        assert exp == proofs[k * psz:(k + 1) * psz]
    cp = (time.perf_counter() - t0) / 64
    t0 = time.perf_counter()
    for k in range(64):
        assert co.sigma_verify(curve, nb, bases[k], pubs[k], proofs[k * psz:(k + 1) * psz], b"Reveal Proof") == 0
    cv = (time.perf_counter() - t0) / 64
    print("%-15s B=%d: prove %.0f/s verify %.0f/s on the GPU (host-buffer API, PCIe included); CPU oracle %.0f/s and %.0f/s per core; "
          "first 64 proofs byte-identical" % (name, B, B / tp, B / tv, 1 / cp, 1 / cv))
