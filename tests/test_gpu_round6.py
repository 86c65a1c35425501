"""GPU tests of round 6 (run on the MI355X box): the split bucket pipeline (k_bucket_sort / k_bucket_acc / k_bucket_reduce, windows of 12
and 13 bits) under the verifier's screen of 256 .. 1 024 proofs per equation -- against the ORACLE at production size, against the
per-proof screen's status words, on single multi-scalar multiplications up to the new term limit -- and the threading contract of the
C ABI (contexts on host threads of their own; one table from two threads)."""
import random
import threading

import pytest

pytestmark = pytest.mark.gpu


def _batch(mp, coracle, torch, cv, m, n, B, seed, fb_bits=16):
    eng = mp._native.Engine(cv, 0)
    g0 = coracle.gen_inputs(cv, m, n, seed)
    t = eng.table(m, n, g0["params"], g0["pk"], fb_bits=fb_bits)
    gpu = torch.device("cuda", 0)
    gen = torch.Generator(device=gpu)
    gen.manual_seed(seed)
    N = m * n
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
    return eng, t, g0, decks, od, op


def test_default_group_sizes(mp, coracle):
    """what the screen of a batch takes by default (round 6): the split pipeline's equations from 16 384 52-card proofs on"""
    cv, m, n = "stark", 2, 26
    eng = mp._native.Engine(cv, 0)
    g0 = coracle.gen_inputs(cv, m, n, 6001)
    t = eng.table(m, n, g0["params"], g0["pk"], fb_bits=8)
    assert [t.group_size(B) for B in (4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288)] == [0, 8, 1024, 1024, 1024, 1024, 1024, 1024]
    t.set_group_verify(30464, 6144)                        # rounds 4-5: at most 128 proofs per equation
    assert [t.group_size(B) for B in (16384, 65536, 262144)] == [16, 64, 128]
    t.close()
    eng.close()


def test_one_equation_of_1024_proofs_against_the_oracle(mp, coracle):
    """the screen at its round-6 production size against the ORACLE: 8 192 proofs of a 52-card deck in 8 equations of 1 024 (243 712 points,
    14-bit windows, ten sorted runs per bucket); 256 members of one equation -- one of them tampered -- are verified by the CPU oracle
    proof by proof next to the engine's verdicts: same accept / reject, same check name"""
    import torch
    cv, m, n, B, L = "stark", 2, 26, 8192, 1024
    eng, t, g0, decks, od, op = _batch(mp, coracle, torch, cv, m, n, B, 6950)
    t.set_group_verify(243712, 0)
    assert t.group_size(B) == L
    sv = torch.full((B,), 55, dtype=torch.int32, device=decks.device)
    eng.profile_enable(True)
    t.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv.data_ptr())
    eng.sync()
    rep = eng.profile_report()
    eng.profile_enable(False)
    assert int(sv.abs().sum().item()) == 0
    assert rep["k_bucket_sort"][0] == 1 and rep["k_bucket_acc"][0] == 1 and rep["k_bucket_reduce"][0] == 1 and "k_bucket_msm" not in rep
    assert dict(eng.last_profile_items)["k_bucket_sort"] == 8 * 18 * 10      # 8 equations x 18 windows of 14 bits x 10 runs of 24 576 terms
    T, grp = B // L, 5
    members = [j * T + grp for j in range(L)]                # lane of (member j, group t) = j T + t
    bad = members[700]
    op[bad, (11 * m + 8) * 64 + 10 * 32 + 2] ^= 1            # one byte of a response scalar
    before = t.reverified_count()
    t.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv.data_ptr())
    eng.sync()
    st = sv.cpu().tolist()
    assert [i for i, v in enumerate(st) if v] == [bad]
    assert t.reverified_count() - before == L                # the members of the failing equation, nobody else
    deck_b = bytes(g0["deck"])
    random.seed(3)
    sample = sorted(set(random.sample(members, 255) + [bad]))
    od_c, op_c = od[sample].cpu().numpy(), op[sample].cpu().numpy()
    for k, b in enumerate(sample):
        assert coracle.verify_shuffle(cv, m, n, g0["params"], g0["pk"], deck_b, od_c[k].tobytes(), op_c[k].tobytes()) == st[b], (k, b)
    t.close()
    eng.close()


@pytest.mark.parametrize("cv,m,n,B", [("stark", 2, 26, 4096), ("secp256k1", 2, 26, 2048), ("stark", 8, 128, 128), ("bls12_377", 2, 5, 512)])
def test_split_pipeline_status_words(mp, coracle, cv, m, n, B):
    """equations of 256 .. 1 024 proofs (12- and 13-bit windows, and 10-bit ones forced through the split pipeline) give the status words of
    the per-proof screen: all accepted; a bad response scalar, a swapped deck and a point off the curve named exactly as without groups"""
    import torch
    eng, t, g0, decks, od, op = _batch(mp, coracle, torch, cv, m, n, B, 6960, fb_bits=8)
    per = 4 * m * n + 11 * m + 8
    pb = eng.point_bytes
    sv = torch.full((B,), 55, dtype=torch.int32, device=decks.device)

    def verify():
        sv.fill_(55)
        t.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv.data_ptr())
        eng.sync()
        return sv.cpu().tolist()
    t.set_group_verify(0, 0)
    assert verify() == [0] * B
    a, b_, c = B // 7, B // 2 + 3, B - 2
    op[a, (11 * m + 8) * pb + 10 * 32 + 2] ^= 1              # a response scalar of proof a
    od[b_, 0:pb] = od[b_ + 1, 0:pb]                          # card 0 of deck b replaced by a neighbour's: a valid point, a wrong statement
    od[c, 5] ^= 1                                            # a coordinate of deck c that is not on the curve any more
    want = verify()
    assert sorted(i for i, v in enumerate(want) if v) == [a, b_, c] and want[c] < 0 and (want[a] > 0 or want[b_] > 0), (want[a], want[b_], want[c])
    for proofs_per_eq, bits, split in ((B // 4, 0, 12), (B // 8, 12, 12), (B // 4, 13, 12), (B // 16, 10, 10), (B // 2, 0, 12)):
        t.set_bucket_split(split)
        t.set_bucket_bits(bits)
        t.set_group_verify(proofs_per_eq * per, 0)
        assert t.group_size(B) == proofs_per_eq
        eng.profile_enable(True)
        got = verify()
        rep = eng.profile_report()
        eng.profile_enable(False)
        assert got == want, (proofs_per_eq, bits)
        assert ("k_bucket_acc" in rep) == (bits == 0 and proofs_per_eq * per >= 50000 or bits >= split), (proofs_per_eq, bits, sorted(rep))
    t.close()
    eng.close()


def test_bucket_msm_up_to_the_new_term_limit(mp, coracle):
    """ONE multi-scalar multiplication of 100 000 (12-bit windows), 300 000 and 589 824 terms (13-bit, 13 and 24 sorted runs per bucket: the
    most a job takes) against the oracle -- the points repeat with period 509, so the oracle's MSM of 509 terms with the summed scalars is
    the same group element -- with scalars that crowd single windows into one bucket"""
    cv = "stark"
    q = 0x0800000000000010ffffffffffffffffb781126dcae7b2321e66a241adc64d2f
    eng = mp._native.Engine(cv, 0)
    g0 = coracle.gen_inputs(cv, 2, 3, 6200)
    t = eng.table(2, 3, g0["params"], g0["pk"])
    per = 509
    base = eng.setup(2, per - 3, bytes([11] * 32))[:64 * per]
    random.seed(66)
    for K in (100000, 300000, 589824):
        sc = [random.randrange(q) for _ in range(K)]
        sc[:4] = [0, 1, q - 1, (1 << 251) + 1]
        sc[K // 2:K // 2 + 30000] = [(random.randrange(q) >> 26 << 26) | 77 for _ in range(30000)]      # two low windows of 30 000 terms in ONE bucket each
        pts = (base * (K // per + 1))[:64 * K]
        folded = [0] * per
        for i, s in enumerate(sc):
            folded[i % per] = (folded[i % per] + s) % q
        want = coracle.msm(cv, b"".join(s.to_bytes(32, "little") for s in folded), base)
        eng.profile_enable(True)
        got = t.msm(1, K, b"".join(s.to_bytes(32, "little") for s in sc), pts)
        rep = eng.profile_report()
        eng.profile_enable(False)
        assert got == want, K
        assert "k_bucket_acc" in rep and "k_bucket_msm" not in rep, K
    # one run more than that: the job stays on the Straus kernel (as jobs of more than 65 535 terms did before round 6)
    K = 589824 + 64
    sc = [random.randrange(q) for _ in range(per)] + [0] * (K - per)
    eng.profile_enable(True)
    got = t.msm(1, K, b"".join(s.to_bytes(32, "little") for s in sc), (base * (K // per + 1))[:64 * K])
    rep = eng.profile_report()
    eng.profile_enable(False)
    assert got == coracle.msm(cv, b"".join(s.to_bytes(32, "little") for s in sc[:per]), base) and "k_bucket_acc" not in rep
    t.close()
    eng.close()


# ---- the threading contract (include/mpshuffle.h): the reference's trait members are associated functions without `self` or global
# state [REF barnett-smart-card-protocol/src/lib.rs:74-197]
def _thread_inputs(coracle, cv, m, n, B, seed0, distinct=8):
    ins = [coracle.gen_inputs(cv, m, n, seed0 + b) for b in range(distinct)]
    rep = B // distinct
    args = (b"".join(g["deck"] for g in ins) * rep, b"".join(g["rho"] for g in ins) * rep, [v for g in ins for v in g["perm"]] * rep,
            b"".join(g["prover_seed"] for g in ins) * rep)
    return ins, args


def test_four_host_threads_four_contexts(mp, coracle):
    """four host threads, each with a context and a table of its own on device 0, each proving and verifying 2 048 proofs at once for
    several iterations -- one of them with a tampered proof in every call, one with pipelined verification: bytes and status words equal
    the single-threaded run's, whose first proofs equal the oracle's"""
    import torch
    cv, m, n, B, ITER = "stark", 2, 26, 2048, 4
    ins, args = _thread_inputs(coracle, cv, m, n, B, 6600)
    g0 = ins[0]
    eng0 = mp._native.Engine(cv, 0)
    t0 = eng0.table(m, n, g0["params"], g0["pk"], fb_bits=16)
    ref_d, ref_p, ref_st = t0.shuffle_and_remask_batch(*args)
    dsz, psz = len(g0["deck"]), t0.proof_bytes
    assert ref_st == [0] * B
    for k, g in enumerate(ins):
        assert (ref_d[k * dsz:(k + 1) * dsz], ref_p[k * psz:(k + 1) * psz]) == coracle.shuffle_and_remask(
            cv, m, n, g0["params"], g0["pk"], g["deck"], g["rho"], g["perm"], g["prover_seed"])
    bad_p = bytearray(ref_p)
    bad_p[1234 * psz + psz - 31] ^= 2
    bad_p = bytes(bad_p)
    want_bad = t0.verify_shuffle_batch(args[0], ref_d, bad_p)
    assert [i for i, v in enumerate(want_bad) if v] == [1234]
    t0.close()
    eng0.close()
    errors = []

    def worker(kind):
        try:
            eng = mp._native.Engine(cv, 0)
            t = eng.table(m, n, g0["params"], g0["pk"], fb_bits=16)
            for _ in range(ITER):
                d, p, st = t.shuffle_and_remask_batch(*args)
                assert st == [0] * B and d == ref_d and p == ref_p, kind
                if kind == "tampered":
                    assert t.verify_shuffle_batch(args[0], d, bad_p) == want_bad
                elif kind == "pipelined":                  # mp_set_pipeline(1): two device-resident verify calls in flight, verdicts at mp_sync
                    gpu = torch.device("cuda", 0)
                    dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(gpu)
                    dk, dd, dp, db = dev(args[0]), dev(d), dev(p), dev(bad_p)
                    s1 = torch.full((B,), 55, dtype=torch.int32, device=gpu)
                    s2 = torch.full((B,), 55, dtype=torch.int32, device=gpu)
                    torch.cuda.synchronize()
                    t.set_pipeline(1)
                    t.verify_shuffle_batch_dev(B, dk.data_ptr(), dd.data_ptr(), dp.data_ptr(), s1.data_ptr())
                    t.verify_shuffle_batch_dev(B, dk.data_ptr(), dd.data_ptr(), db.data_ptr(), s2.data_ptr())
                    eng.sync()
                    assert s1.cpu().tolist() == [0] * B and s2.cpu().tolist() == want_bad
                    t.set_pipeline(0)
                else:
                    assert t.verify_shuffle_batch(args[0], d, p) == [0] * B
            t.close()
            eng.close()
        except BaseException as e:      # noqa: BLE001 -- reported to the main thread
            errors.append((kind, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in ("plain", "tampered", "pipelined", "plain2")]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_two_host_threads_one_table(mp, coracle):
    """two host threads on ONE table: the library serialises the calls on the context's lock (include/mpshuffle.h) -- every call returns
    the bytes and status words of the single-threaded run, whatever the interleaving"""
    cv, m, n, B, ITER = "stark", 2, 26, 1024, 6
    ins, args = _thread_inputs(coracle, cv, m, n, B, 6700)
    g0 = ins[0]
    eng = mp._native.Engine(cv, 0)
    t = eng.table(m, n, g0["params"], g0["pk"], fb_bits=16)
    ref_d, ref_p, ref_st = t.shuffle_and_remask_batch(*args)
    psz = t.proof_bytes
    half = (args[0][:len(args[0]) // 2], args[1][:len(args[1]) // 2], args[2][:len(args[2]) // 2], args[3][:len(args[3]) // 2])
    ref_half = t.shuffle_and_remask_batch(*half)
    bad_p = bytearray(ref_p)
    bad_p[77 * psz + psz - 31] ^= 2
    bad_p = bytes(bad_p)
    want_bad = t.verify_shuffle_batch(args[0], ref_d, bad_p)
    assert [i for i, v in enumerate(want_bad) if v] == [77]
    errors = []

    def worker(kind):
        try:
            for _ in range(ITER):
                if kind == "full":
                    d, p, st = t.shuffle_and_remask_batch(*args)
                    assert (d, p, st) == (ref_d, ref_p, ref_st)
                    assert t.verify_shuffle_batch(args[0], d, bad_p) == want_bad
                else:                                       # another batch size: the arenas change their stride between the calls of the other thread
                    assert t.shuffle_and_remask_batch(*half) == ref_half
                    assert t.verify_shuffle_batch(half[0], ref_half[0], ref_half[1]) == [0] * (B // 2)
                    t.set_work_split(-1)
        except BaseException as e:      # noqa: BLE001
            errors.append((kind, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in ("full", "half")]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    t.close()
    eng.close()
