#!/usr/bin/env python3
"""Many card tables at once -- BASELINE config 3 on one MI355X: T tables of P players share the public
parameters and differ in their aggregate key; at every table the players shuffle-and-remask the 52-card deck in turn
[REF barnett-smart-card-protocol/examples/round.rs:263-341: deck_{j+1} = player j's output], and every shuffle is
verified.  The dependency is along a table's chain, the parallelism across tables: step j is ONE keyed batch of T proofs
(`mp_shuffle_and_remask_batch_keys_dev`, one aggregate key per proof), its output decks stay in HBM and are step j+1's
input; the P*T proofs are verified as they are produced (`mp_verify_shuffle_batch_keys_dev`).

Reports proofs/s for the whole tournament (prove + verify of every shuffle), and spot-checks one table's chain against the
CPU oracle (test infrastructure) when --check is given."""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mp = importlib.import_module("mental-poker_amd")


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--tables", type=int, default=65536)
    ap.add_argument("--players", type=int, default=4)
    ap.add_argument("--check", action="store_true", help="re-run table 0's chain on the CPU oracle and compare bytes")
    ap.add_argument("--chain-verify", action="store_true",
                    help="keep every deck and proof in HBM and verify all tables' chains at the end with ONE equation per table "
                         "(mp_verify_shuffle_chain_dev) instead of link by link as the proofs are produced")
    ap.add_argument("--fb-bits", type=int, default=20, help="window width of the fixed-base tables of the shared parameters (8, 16, 20, 21)")
    ap.add_argument("--keyset", action="store_true",
                    help="hand the tables' aggregate keys over once as a key set (mp_keyset_create) and name them by index, instead of "
                         "passing one key per proof with every call")
    args = ap.parse_args()
    m, n, curve = 2, 26, "stark"
    N, T, P = m * n, args.tables, args.players
    gpu = torch.device("cuda:0")
    eng = mp.Engine(curve, device=0)
    params = eng.setup(m, n, bytes([1] * 32))
    t = eng.table(m, n, params, None, fb_bits=args.fb_bits)          # parameters only: every proof brings its own aggregate key
    gen = torch.Generator(device=gpu)
    gen.manual_seed(1)

    def rand_bytes(*shape):
        return torch.randint(0, 256, shape, dtype=torch.uint8, device=gpu, generator=gen)

    # aggregate keys: 4096 distinct random group elements spread over the tables; initial decks: random ciphertexts
    K = min(T, 4096)
    key_bytes = eng.setup(m, max(K, 2), bytes([4] * 32))[:64 * K]
    kpts = torch.frombuffer(bytearray(key_bytes), dtype=torch.uint8).to(gpu).view(K, 64)
    keys = kpts[torch.arange(T, device=gpu) % K].contiguous()
    kset = t.keyset(key_bytes) if args.keyset else None
    kidx = (torch.arange(T, device=gpu) % K).to(torch.int32).contiguous()

    def prove(*a):
        if kset is not None:
            return t.shuffle_and_remask_batch_keyset_dev(kset, T, kidx.data_ptr(), *a)
        return t.shuffle_and_remask_batch_keys_dev(T, keys.data_ptr(), *a)
    base = torch.frombuffer(bytearray(eng.setup(m, 2 * N - 3, bytes([3] * 32))), dtype=torch.uint8).to(gpu)
    deck = base.repeat(T, 1).contiguous()
    nxt = torch.empty_like(deck)
    proofs = torch.empty(T, t.proof_bytes, dtype=torch.uint8, device=gpu)
    st_p = torch.empty(T, dtype=torch.int32, device=gpu)
    st_v = torch.empty(T, dtype=torch.int32, device=gpu)
    t.reserve(T)
    trace = []
    if args.chain_verify:
        chain = torch.empty(P + 1, T, N * 128, dtype=torch.uint8, device=gpu)
        chain[0] = deck
        all_proofs = torch.empty(P, T, t.proof_bytes, dtype=torch.uint8, device=gpu)
        st_c = torch.empty(P, T, dtype=torch.int32, device=gpu)
    # warm-up on scratch outputs: the first keyed call builds the keyed plans and grows the batch workspace
    w_rho = rand_bytes(T, N, 32)
    w_rho[:, :, 31] &= 0x07
    w_perm = torch.argsort(torch.rand(T, N, device=gpu, generator=gen), dim=1).to(torch.int32).contiguous()
    prove(deck.data_ptr(), w_rho.data_ptr(), w_perm.data_ptr(), rand_bytes(T, 32).data_ptr(), nxt.data_ptr(), proofs.data_ptr(), st_p.data_ptr())
    eng.sync()
    del w_rho, w_perm
    torch.cuda.synchronize()
    busy = 0.0
    for j in range(P):
        rho = rand_bytes(T, N, 32)
        rho[:, :, 31] &= 0x07
        perms = torch.argsort(torch.rand(T, N, device=gpu, generator=gen), dim=1).to(torch.int32).contiguous()
        seeds = rand_bytes(T, 32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()          # the players' random choices above are input generation, not timed
        prove(deck.data_ptr(), rho.data_ptr(), perms.data_ptr(), seeds.data_ptr(), nxt.data_ptr(), proofs.data_ptr(), st_p.data_ptr())
        if args.chain_verify:
            st_v.zero_()
        else:
            t.verify_shuffle_batch_keys_dev(T, keys.data_ptr(), deck.data_ptr(), nxt.data_ptr(), proofs.data_ptr(), st_v.data_ptr())
        eng.sync()
        busy += time.perf_counter() - t0
        assert int(st_p.abs().sum().item()) == 0 and int(st_v.abs().sum().item()) == 0, "a shuffle failed"
        if args.chain_verify:
            chain[j + 1] = nxt
            all_proofs[j] = proofs
        if args.check:
            trace.append(tuple(bytes(x[0].cpu().numpy().tobytes()) for x in (deck, rho, seeds, nxt, proofs)) + ([int(v) for v in perms[0].tolist()],))
        deck, nxt = nxt, deck
    if args.chain_verify:
        kk = keys.repeat(P, 1).contiguous()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t.verify_shuffle_chain_dev(T, P, kk.data_ptr(), chain.data_ptr(), all_proofs.data_ptr(), st_c.data_ptr())
        eng.sync()
        busy += time.perf_counter() - t0
        assert int(st_c.abs().sum().item()) == 0, "a chain failed"
    print("%d tables x %d players, 52 cards, %d distinct aggregate keys: %d shuffles proved and verified in %.2f s of engine time "
          "= %.0f proofs/s" % (T, P, K, T * P, busy, T * P / busy))
    if args.check:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import coracle
        pk0 = bytes(keys[0].cpu().numpy().tobytes())
        for j, (d_in, rho, seed, d_out, proof, perm) in enumerate(trace):
            ed, ep = coracle.shuffle_and_remask(curve, m, n, params, pk0, d_in, rho, perm, seed)
            assert ed == d_out and ep == proof, "table 0, player %d: GPU and CPU oracle disagree" % j
            assert coracle.verify_shuffle(curve, m, n, params, pk0, d_in, d_out, proof) == 0
        print("table 0: all %d shuffles byte-identical to the CPU oracle under that table's key" % P)


if __name__ == "__main__":
    main()
