"""GPU tests of round 5 (run on the MI355X box): the `round` flow end to end [REF examples/round.rs:228-436], fixed-base tables sized
under memory pressure (co-resident ranks), and the host-buffer entry points at their new chunking."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_round_example_runs_end_to_end():
    """BASELINE config 1 as a script: 4 players, 52 cards, keygen + key proofs, aggregate key, masked deck, one shuffle per player (proved
    and verified), one private card each opened with the others' reveal tokens -- four DISTINCT cards and `round ok`
    [REF examples/round.rs:430-433]"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "round.py")], cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    text = out.stdout.decode()
    holds = re.findall(r"^(\S+) holds the (.+)$", text, flags=re.M)
    assert len(holds) == 4 and len({c for _, c in holds}) == 4 and len({p for p, _ in holds}) == 4, text
    assert "4 shuffles proved and verified" in text and text.strip().endswith("round ok")


def test_table_create_under_memory_pressure_takes_narrower_windows(mp, coracle):
    """mp_table_create sizes the fixed-base windows by the free HBM; with most of the memory taken (8 ranks on one GPU, another context's
    tables) it must come back with narrower windows -- and the same proofs -- instead of failing"""
    import torch
    cv, m, n = "stark", 2, 26
    g = coracle.gen_inputs(cv, m, n, 8100)
    eng = mp._native.Engine(cv, 0)
    free_b, total_b = torch.cuda.mem_get_info(0)
    wide = eng.table(m, n, g["params"], g["pk"])
    bits_free = wide.fb_bits
    wide.close()
    torch.cuda.empty_cache()
    free_b, _ = torch.cuda.mem_get_info(0)
    hog = torch.empty(int(free_b * 0.86), dtype=torch.uint8, device="cuda:0")      # 86 % of what is free: < 41 GB left on a 288 GB part
    t = eng.table(m, n, g["params"], g["pk"])
    assert t.fb_bits < bits_free or bits_free == 8, (t.fb_bits, bits_free)
    assert t.fb_bits in (8, 16, 20)
    perm = list(g["perm"])
    d, p, st = t.shuffle_and_remask_batch(g["deck"], g["rho"], perm, g["prover_seed"])
    assert st == [0] and (d, p) == coracle.shuffle_and_remask(cv, m, n, g["params"], g["pk"], g["deck"], g["rho"], perm, g["prover_seed"])
    assert t.verify_shuffle_batch(g["deck"], d, p) == [0]
    del hog
    t.close()
    eng.close()
    torch.cuda.empty_cache()      # (the caching allocator would keep the 86 % for this process: later tests launch other processes)


def test_chain_fixture_on_the_hip_engine(mp):
    """tests/golden/chain_stark_m2_n3_L3_s21.json (one table, three dependent shuffles, one key): the HIP engine proves every link byte
    for byte, verifies the chain with ONE equation (mp_verify_shuffle_chain) and link by link; a chain whose middle proof is replaced is
    rejected at exactly that link"""
    from conftest import GOLDEN, load_json
    g = load_json(os.path.join(GOLDEN, "chain_stark_m2_n3_L3_s21.json"))
    cv, m, n, L = g["curve"], g["m"], g["n"], g["links"]
    eng = mp._native.Engine(cv, 0)
    t = eng.table(m, n, bytes.fromhex(g["params"]), bytes.fromhex(g["pk"]))
    decks = [bytes.fromhex(d) for d in g["decks"]]
    proofs = []
    for j, link in enumerate(g["chain"]):
        d, p, st = t.shuffle_and_remask_batch(decks[j], bytes.fromhex(link["rho"]), link["perm"], bytes.fromhex(link["prover_seed"]))
        assert st == [0] and d == decks[j + 1] and p.hex() == link["proof"], j
        assert t.verify_shuffle_batch(decks[j], d, p) == [0]
        proofs.append(p)
    assert t.verify_shuffle_chain(1, L, b"".join(decks), b"".join(proofs), None) == [0] * L
    bad = list(proofs)
    bad[1] = proofs[2]
    st = t.verify_shuffle_chain(1, L, b"".join(decks), b"".join(bad), None)
    assert st[0] == 0 and st[1] > 0 and st[2] == 0
    t.close()
    eng.close()


def test_groups_adapt_to_sustained_rejection(mp, coracle):
    """mp_set_group_adapt (default on): 8 192 proofs in 64 groups of 128; with 1 % of the traffic tampered (random positions) 72 % of the
    groups fail, and the table halves its groups from call to call -- 128, 64, 32, 16 -- until fewer than a fifth of them fail; every call
    rejects exactly the tampered proofs; honest traffic restores the size (one step per call, two when no group fails at all); with adaptation off the size stays"""
    import torch
    cv, m, n, B = "stark", 2, 26, 8192
    eng = mp._native.Engine(cv, 0)
    g0 = coracle.gen_inputs(cv, m, n, 5990)
    t = eng.table(m, n, g0["params"], g0["pk"], fb_bits=16)
    t.set_group_verify(30464, 0)
    gpu = torch.device("cuda", 0)
    gen = torch.Generator(device=gpu)
    gen.manual_seed(17)
    N = m * n
    decks = torch.frombuffer(bytearray(g0["deck"]), dtype=torch.uint8).to(gpu).repeat(B, 1).contiguous()
    rho = torch.randint(0, 256, (B, N, 32), dtype=torch.uint8, device=gpu, generator=gen)
    rho[:, :, 31] &= 7
    perm = torch.argsort(torch.rand(B, N, device=gpu, generator=gen), dim=1).to(torch.int32).contiguous()
    seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device=gpu, generator=gen)
    od = torch.empty(B, len(g0["deck"]), dtype=torch.uint8, device=gpu)
    op = torch.empty(B, t.proof_bytes, dtype=torch.uint8, device=gpu)
    sp = torch.empty(B, dtype=torch.int32, device=gpu)
    sv = torch.empty(B, dtype=torch.int32, device=gpu)
    t.shuffle_and_remask_batch_dev(B, decks.data_ptr(), rho.data_ptr(), perm.data_ptr(), seeds.data_ptr(), od.data_ptr(), op.data_ptr(), sp.data_ptr())
    eng.sync()
    good = op.clone()
    idx = torch.randperm(B, generator=torch.Generator().manual_seed(5))[:82].to(gpu)
    bad = good.clone()
    bad[idx, t.proof_bytes - 31] ^= 2
    want = torch.zeros(B, dtype=torch.bool, device=gpu)
    want[idx] = True

    def verify(proofs, expect):
        sv.fill_(55)
        t.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), proofs.data_ptr(), sv.data_ptr())
        eng.sync()
        assert torch.equal(sv != 0, expect)
    nobody = torch.zeros(B, dtype=torch.bool, device=gpu)
    sizes = []
    for _ in range(6):
        sizes.append(t.group_size(B))
        verify(bad, want)
    assert sizes[:4] == [128, 64, 32, 16] and sizes[4] in (16, 8) and sizes[5] == sizes[4], sizes
    for _ in range(5):
        verify(good, nobody)
    assert t.group_size(B) == 128
    t.set_group_adapt(False)
    for _ in range(3):
        verify(bad, want)
        assert t.group_size(B) == 128
    t.close()
    eng.close()


@pytest.mark.parametrize("cv,m,n,B", [("stark", 2, 26, 3000), ("secp256k1", 2, 7, 2304), ("stark", 4, 13, 1536)])
def test_rejection_fuzz_every_strategy_gives_the_per_equation_verdicts(mp, coracle, cv, m, n, B):
    """randomised rejection: a batch with a random mix of tampered response scalars, swapped decks and broken point encodings is
    verified (a) equation by equation with no screen at all -- the reference's order of evaluation, the expected status words --, then
    with the per-proof screen, with groups of several sizes with and without sub-groups, with adapting groups, waiting and pipelined
    (depth 1 and 2, three calls each).  Every strategy must return exactly the status words of (a); a sample of them is checked against
    the CPU oracle proof by proof."""
    import random
    import torch
    rnd = random.Random(1234 + B)
    eng = mp._native.Engine(cv, 0)
    g0 = coracle.gen_inputs(cv, m, n, 8800)
    t = eng.table(m, n, g0["params"], g0["pk"], fb_bits=8)
    gpu = torch.device("cuda", 0)
    gen = torch.Generator(device=gpu)
    gen.manual_seed(B)
    N, pb = m * n, eng.point_bytes
    per = 4 * N + 11 * m + 8
    decks = torch.frombuffer(bytearray(g0["deck"]), dtype=torch.uint8).to(gpu).repeat(B, 1).contiguous()
    rho = torch.randint(0, 256, (B, N, 32), dtype=torch.uint8, device=gpu, generator=gen)
    rho[:, :, 31] &= 7
    perm = torch.argsort(torch.rand(B, N, device=gpu, generator=gen), dim=1).to(torch.int32).contiguous()
    seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device=gpu, generator=gen)
    od = torch.empty(B, len(g0["deck"]), dtype=torch.uint8, device=gpu)
    op = torch.empty(B, t.proof_bytes, dtype=torch.uint8, device=gpu)
    sp = torch.empty(B, dtype=torch.int32, device=gpu)
    t.shuffle_and_remask_batch_dev(B, decks.data_ptr(), rho.data_ptr(), perm.data_ptr(), seeds.data_ptr(), od.data_ptr(), op.data_ptr(), sp.data_ptr())
    eng.sync()
    assert int(sp.abs().sum().item()) == 0
    # tamper: ~4 % of the proofs, three kinds
    victims = rnd.sample(range(B), max(6, B // 25))
    kinds = {}
    for b in victims:
        kind = rnd.choice(("scalar", "deck", "encoding"))
        kinds[b] = kind
        if kind == "scalar":
            op[b, (11 * m + 8) * pb + 32 * rnd.randrange(5 * n + 9) + rnd.randrange(8)] ^= 1 << rnd.randrange(8)
        elif kind == "deck":                                   # another proof's output deck: valid points, wrong statement
            od[b] = od[(b + 1) % B].clone()
        else:                                                  # a coordinate that is not on the curve (or not canonical) any more
            od[b, rnd.randrange(2 * N) * pb + rnd.randrange(pb)] ^= 1 << rnd.randrange(8)
    torch.cuda.synchronize()

    def verify(depth=0, calls=1):
        t.set_pipeline(depth)
        outs = [torch.full((B,), 55, dtype=torch.int32, device=gpu) for _ in range(calls)]
        for sv in outs:
            t.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv.data_ptr())
        eng.sync()
        t.set_pipeline(0)
        return [sv.cpu().tolist() for sv in outs]
    t.set_merged_verify(False)
    want = verify()[0]
    t.set_merged_verify(True)
    bad = [b for b, v in enumerate(want) if v]
    assert sorted(bad) == sorted(victims)
    assert all(want[b] < 0 for b in victims if kinds[b] == "encoding") and any(want[b] > 0 for b in victims)
    # (a swapped deck is > 0 unless the neighbour it came from had its encoding broken; a flipped scalar > 0, or < 0 if no longer canonical)
    deck_b = bytes(g0["deck"])
    for b in rnd.sample(bad, 4) + rnd.sample(range(B), 4):
        if want[b] >= 0:                                       # (the oracle has no encoding errors: it takes points as they come)
            got = coracle.verify_shuffle(cv, m, n, g0["params"], g0["pk"], deck_b, od[b].cpu().numpy().tobytes(), op[b].cpu().numpy().tobytes())
            assert got == want[b], (b, kinds.get(b))
    configs = [("per-proof screen", (0, 0), (0, 0), True)]
    for L in (4, 16, 64):
        configs.append(("groups of %d" % L, (L * per, 0), (0, 1 << 30), False))
        configs.append(("groups of %d, sub-groups of %d" % (L, max(2, L // 4)), (L * per, 0), (max(2, L // 4) * per, 1), False))
    configs.append(("groups of 64, adapting", (64 * per, 0), (0, 0), True))
    for name, gv, gr, adapt in configs:
        t.set_group_verify(*gv)
        t.set_group_refine(*gr)
        t.set_group_adapt(adapt)
        for depth, calls in ((0, 2), (1, 3), (2, 3)):
            for got in verify(depth, calls):
                assert got == want, (name, depth, [(b, got[b], want[b]) for b in range(B) if got[b] != want[b]][:5])
    t.close()
    eng.close()


@pytest.mark.parametrize("cvn,m,n,L,T,keyed,group", [("stark", 2, 3, 3, 4, True, 2), ("stark", 2, 3, 4, 3, False, 3), ("secp256k1", 2, 3, 2, 4, False, 4)])
def test_grouped_chain_cases_on_the_hip_engine(mp, coracle, cvn, m, n, L, T, keyed, group):
    """mp_set_chain_group: the chains of `group` tables in ONE equation -- every case of the chain tests (honest chains, a replaced proof,
    a tampered inner deck, links of a table under different keys, the cheating prover whose transcript absorbs another key) gives the
    per-link verifier's status words, and a failing equation sends the links of ITS tables, nobody else's, through the per-link pass"""
    from chain_cases import run_chain_cases
    eng = mp._native.Engine(cvn, 0)
    run_chain_cases(eng, coracle, cvn, m, n, L, T, keyed, group)
    eng.close()


def test_grouped_chains_of_52_card_tables(mp, coracle):
    """96 tables x 8 dependent shuffles of a 52-card deck, one key per table, the engine's own prover: chain equations of 6 tables each
    (5 640 points: 8-bit windows), of 24 (22 560: 10-bit windows), by size (nothing to group at 96 tables) and table by table give the
    same verdicts as the per-link verifier -- all zero for the honest chains, and exactly the replaced proof / the tampered deck's two
    links otherwise; one link of one table is checked against the oracle"""
    import random
    cv, m, n, L, T = "stark", 2, 26, 8, 96
    N = m * n
    g0 = coracle.gen_inputs(cv, m, n, 9300)
    eng = mp._native.Engine(cv, 0)
    pb = eng.point_bytes
    t = eng.table(m, n, g0["params"], None)
    rnd = random.Random(93)
    keys_t = [coracle.gen_inputs(cv, 2, 2, 9400 + i)["pk"] for i in range(T)]
    keys = b"".join(keys_t)
    chain = [b"".join(coracle.gen_inputs(cv, m, n, 9500 + (i % 4))["deck"] for i in range(T))]
    proofs, wit = [], []
    for j in range(L):
        rho = b"".join((rnd.getrandbits(248)).to_bytes(32, "little") for _ in range(T * N))
        perm = []
        for _ in range(T):
            p = list(range(N))
            rnd.shuffle(p)
            perm += p
        seeds = bytes(rnd.getrandbits(8) for _ in range(32 * T))
        d, p, st = t.shuffle_and_remask_batch_keys(keys, chain[j], rho, perm, seeds)
        assert st == [0] * T
        chain.append(d)
        proofs.append(p)
        wit.append((rho, perm, seeds))
    dsz, psz = N * 2 * pb, len(proofs[0]) // T
    tt, jj = 37, 5
    rho, perm, seeds = wit[jj]
    exp = coracle.shuffle_and_remask(cv, m, n, g0["params"], keys_t[tt], chain[jj][tt * dsz:(tt + 1) * dsz], rho[tt * N * 32:(tt + 1) * N * 32],
                                     perm[tt * N:(tt + 1) * N], seeds[tt * 32:(tt + 1) * 32])
    assert exp == (chain[jj + 1][tt * dsz:(tt + 1) * dsz], proofs[jj][tt * psz:(tt + 1) * psz])
    decks, pf, kk = b"".join(chain), b"".join(proofs), keys * L
    bad = bytearray(pf)
    o = (3 * T + 50) * psz
    bad[o:o + psz] = pf[(3 * T + 51) * psz:(3 * T + 52) * psz]          # link 3 of table 50 carries table 51's proof
    tam = bytearray(decks)
    tam[(2 * T + 7) * dsz + 40] ^= 4                                     # deck 2 of table 7: links 1 and 2 of that table
    per_link = []
    for j in range(L):
        per_link += t.verify_shuffle_batch_keys(keys, bytes(tam[j * T * dsz:(j + 1) * T * dsz]), bytes(tam[(j + 1) * T * dsz:(j + 2) * T * dsz]),
                                                bytes(bad[j * T * psz:(j + 1) * T * psz]))
    assert per_link[3 * T + 50] > 0 and per_link[1 * T + 7] != 0 and per_link[2 * T + 7] != 0 and sum(1 for v in per_link if v) == 3
    for group, tables_looked_at in ((6, 12), (24, 48), (0, 2), (1, 2)):
        t.set_chain_group(group)
        eng.profile_enable(True)
        assert t.verify_shuffle_chain(T, L, decks, pf, kk) == [0] * (T * L), group
        rep = eng.profile_report()
        eng.profile_enable(False)
        assert rep["k_bucket_msm"][0] == 1 and "k_var_msm" not in rep, (group, rep)
        looked = t.reverified_count()
        assert t.verify_shuffle_chain(T, L, bytes(tam), bytes(bad), kk) == per_link, group
        assert t.reverified_count() - looked == tables_looked_at * L, (group, t.reverified_count() - looked)
    t.close()
    eng.close()


@pytest.mark.parametrize("extra", [[], ["--keyset"], ["--chain-verify", "--keyset"]])
def test_tournament_example_runs(extra):
    """BASELINE config 3 as a script (examples/tournament.py): 384 card tables x 4 players, one aggregate key per table (passed with
    every call, or prepared once as a key set), every shuffle proved and verified -- link by link, or the tables' chains at the end
    (one chain equation per table at this size: groups of tables from ~2 000 tables on) -- and table 0's chain byte-identical to the CPU oracle under that table's key"""
    import torch
    torch.cuda.empty_cache()      # (this process's cached device memory is not available to the script's process)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "tournament.py"), "--tables", "384", "--players", "4", "--check",
                          "--fb-bits", "16"] + extra,
                         cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    text = out.stdout.decode()
    assert "1536 shuffles proved and verified" in text and "table 0: all 4 shuffles byte-identical to the CPU oracle" in text, text


def test_parameter_selection_example_runs():
    """the reference's only benchmark harness [REF examples/parameter_selection.rs:25-96] as a script on the engine (row f3): BLS12-377,
    300 cards, the five (m, n) pairs; every pair proves and verifies, and the proof is smallest at (10, 30) -- 10 840 bytes of
    compressed points -- as the reference's doc comment predicts [REF parameter_selection.rs:10]"""
    import torch
    torch.cuda.empty_cache()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "parameter_selection.py"), "--batch", "32"], cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    rows = [ln.split() for ln in out.stdout.decode().splitlines() if re.match(r"^\s*\d+\s+\d+\s+\|", ln)]
    sizes = {(int(r[0]), int(r[1])): int(r[7]) for r in rows}
    assert sorted(sizes) == [(2, 150), (6, 50), (10, 30), (12, 25), (30, 10)], rows
    assert min(sizes, key=sizes.get) == (10, 30) and sizes[(10, 30)] == 10840, sizes
