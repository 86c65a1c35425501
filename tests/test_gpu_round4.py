"""-m gpu: round 4 -- window lanes, pipelined verification, tables sized by free HBM, the cheating-prover chain case on the HIP build.
Every comparison is byte-for-byte against the C++ oracle (coracle) through the C ABI; nothing here reads /root/reference."""
import hashlib

import pytest

pytestmark = pytest.mark.gpu


def _inputs(coracle, cv, m, n, B, seed0):
    ins = [coracle.gen_inputs(cv, m, n, seed0 + b) for b in range(B)]
    args = (b"".join(g["deck"] for g in ins), b"".join(g["rho"] for g in ins), [v for g in ins for v in g["perm"]],
            b"".join(g["prover_seed"] for g in ins))
    return ins, args


def _expected(coracle, cv, m, n, g0, ins):
    d, p = [], []
    for g in ins:
        ed, ep = coracle.shuffle_and_remask(cv, m, n, g0["params"], g0["pk"], g["deck"], g["rho"], g["perm"], g["prover_seed"])
        d.append(ed)
        p.append(ep)
    return b"".join(d), b"".join(p)


@pytest.mark.parametrize("cv,m,n,B", [("stark", 2, 26, 5), ("stark", 4, 13, 3), ("secp256k1", 2, 7, 3), ("bls12_377", 2, 5, 2)])
def test_window_lanes_match_oracle(mp, coracle, cv, m, n, B):
    """the windows of a variable-base sub-job dealt to k lanes (+ one fold per MSM): same bytes as the oracle for every k, on every work
    split, with one and with four lanes per group operation, merged and per-equation verification"""
    eng = mp._native.Engine(cv, 0)
    ins, args = _inputs(coracle, cv, m, n, B, 5100)
    g0 = ins[0]
    t = eng.table(m, n, g0["params"], g0["pk"])
    exp = _expected(coracle, cv, m, n, g0, ins)
    dsz = len(g0["deck"])
    rot = exp[0][dsz:] + exp[0][:dsz]
    for split, prm in ((2, (4, 16, 16, 32, 3)), (0, (8, 64, 64, 64, 16)), (1, (1, 2, 2, 4, 5)), (5, (1, 1, 2, 4, 2)), (4, (4, 64, 32, 32, 7)),
                       (3, (1, 1, 2, 4, 4))):
        t.set_plan_params(split, *prm)
        t.set_work_split(split)
        for lanes in (1, 0, 4):
            t.set_group_lanes(lanes)
            eng.profile_enable(True)
            out = t.shuffle_and_remask_batch(*args)
            rep = eng.profile_report()
            eng.profile_enable(False)
            assert (out[0], out[1]) == exp, (split, prm, lanes)
            assert any(k.startswith("k_wfold") for k in rep), sorted(rep)
            for merged in (True, False):
                t.set_merged_verify(merged)
                assert t.verify_shuffle_batch(args[0], out[0], out[1]) == [0] * B
                st = t.verify_shuffle_batch(args[0], rot, out[1])
                assert [eng.check_name(v) for v in st] == ["Hadamard Product (5.1)"] * B
            t.set_merged_verify(True)
    t.close()
    eng.close()


def test_default_plans_by_batch_size_match_oracle(mp, coracle):
    """the engine's own choice of split by batch size (finest / small / latency / medium thresholds scaled down so that 2 .. 7 proofs
    cross all of them): byte-identical outputs"""
    cv, m, n, B = "stark", 2, 26, 7
    eng = mp._native.Engine(cv, 0)
    ins, args = _inputs(coracle, cv, m, n, B, 5200)
    g0 = ins[0]
    t = eng.table(m, n, g0["params"], g0["pk"])
    exp = _expected(coracle, cv, m, n, g0, ins)
    dsz, psz = len(g0["deck"]), t.proof_bytes
    t.set_plan_thresholds(1, 2, 3, 4, 6)      # 1: finest, 2: small, 3: latency, 4: medium, 5-6: wide, 7: throughput
    for k in range(1, B + 1):
        sub = (args[0][:k * dsz], args[1][:k * 32 * m * n], args[2][:k * m * n], args[3][:k * 32])
        out = t.shuffle_and_remask_batch(*sub)
        assert out[0] == exp[0][:k * dsz] and out[1] == exp[1][:k * psz], k
        assert t.verify_shuffle_batch(sub[0], out[0], out[1]) == [0] * k
    t.close()
    eng.close()


def test_pipelined_verification_status_words(mp, coracle):
    """mp_set_pipeline: verify calls on the second lane, verdicts examined `depth` calls later -- status words identical to the waiting
    calls, for honest batches, a batch with one bad proof (deferred per-equation pass) and a batch in which every proof fails; prove calls
    issued in between are unaffected; mp_sync completes everything"""
    import torch
    cv, m, n, B = "stark", 2, 26, 6
    eng = mp._native.Engine(cv, 0)
    ins, args = _inputs(coracle, cv, m, n, B, 5300)
    g0 = ins[0]
    t = eng.table(m, n, g0["params"], g0["pk"])
    exp = _expected(coracle, cv, m, n, g0, ins)
    dsz, psz = len(g0["deck"]), t.proof_bytes
    gpu = torch.device("cuda", 0)
    dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(gpu)
    bad = bytearray(exp[1])
    bad[2 * psz + 70] ^= 4
    cases = {"good": (dev(exp[0]), dev(exp[1])), "badproof": (dev(exp[0]), dev(bytes(bad))),
             "rotated": (dev(exp[0][dsz:] + exp[0][:dsz]), dev(exp[1]))}
    decks = dev(args[0])
    rho, seeds = dev(args[1]), dev(args[3])
    perm = torch.tensor(args[2], dtype=torch.int32, device=gpu)
    for split in (2, 0, 4):                                   # splits that screen with the merged equation
        t.set_work_split(split)
        t.set_pipeline(0)
        want = {}
        for name, (d, p) in cases.items():
            st = torch.full((B,), 77, dtype=torch.int32, device=gpu)
            t.verify_shuffle_batch_dev(B, decks.data_ptr(), d.data_ptr(), p.data_ptr(), st.data_ptr())
            eng.sync()
            want[name] = st.cpu().tolist()
        assert want["good"] == [0] * B and want["badproof"][2] != 0 and sum(1 for v in want["badproof"] if v) == 1 and all(want["rotated"])
        for depth in (1, 3):
            t.set_pipeline(depth)
            order = ["good", "badproof", "good", "rotated", "badproof", "good"]
            sts = [torch.full((B,), 77, dtype=torch.int32, device=gpu) for _ in order]
            od = torch.empty(B * dsz, dtype=torch.uint8, device=gpu)
            op = torch.empty(B * psz, dtype=torch.uint8, device=gpu)
            sp = torch.full((B,), 77, dtype=torch.int32, device=gpu)
            for st, name in zip(sts, order):
                d, p = cases[name]
                t.verify_shuffle_batch_dev(B, decks.data_ptr(), d.data_ptr(), p.data_ptr(), st.data_ptr())
                # a prove call of the next batch runs beside it (outputs of its own: nothing a pending verify reads is touched)
                t.shuffle_and_remask_batch_dev(B, decks.data_ptr(), rho.data_ptr(), perm.data_ptr(), seeds.data_ptr(), od.data_ptr(),
                                               op.data_ptr(), sp.data_ptr())
            eng.sync()
            for st, name in zip(sts, order):
                assert st.cpu().tolist() == want[name], (split, depth, name)
            assert sp.cpu().tolist() == [0] * B
            assert bytes(od.cpu().numpy().tobytes()) == exp[0] and bytes(op.cpu().numpy().tobytes()) == exp[1]
    t.set_pipeline(0)
    t.set_work_split(-1)
    t.close()
    eng.close()


def test_table_sized_by_free_hbm_gives_identical_proofs(mp, coracle):
    """mp_table_create (fb_bits = 0): the engine picks the widest fixed-base windows the free HBM allows -- at least 16 bits on an MI355X
    -- and the proofs are byte-identical to those of the 8-bit table and to the oracle's"""
    cv, m, n, B = "stark", 2, 26, 4
    eng = mp._native.Engine(cv, 0)
    ins, args = _inputs(coracle, cv, m, n, B, 5400)
    g0 = ins[0]
    exp = _expected(coracle, cv, m, n, g0, ins)
    t8 = eng.table(m, n, g0["params"], g0["pk"], fb_bits=8)
    assert t8.fb_bits == 8
    out8 = t8.shuffle_and_remask_batch(*args)
    t8.close()
    ta = eng.table(m, n, g0["params"], g0["pk"], fb_bits=0)
    assert ta.fb_bits in (16, 20, 21), ta.fb_bits
    outa = ta.shuffle_and_remask_batch(*args)
    assert (outa[0], outa[1]) == (out8[0], out8[1]) == exp
    assert ta.verify_shuffle_batch(args[0], outa[0], outa[1]) == [0] * B
    ta.close()
    eng.close()


@pytest.mark.parametrize("cvn,m,n,L,T,keyed", [("stark", 3, 2, 3, 2, True), ("stark", 2, 3, 4, 3, False)])
def test_chain_cases_on_the_hip_engine(mp, coracle, cvn, m, n, L, T, keyed):
    """the chain-verification cases of the emulator test on the HIP build -- among them the cheating prover whose inner link is made under
    the table's key while its transcript absorbs another one (tested only under emulation until now)"""
    from chain_cases import run_chain_cases
    eng = mp._native.Engine(cvn, 0)
    run_chain_cases(eng, coracle, cvn, m, n, L, T, keyed)
    eng.close()
    maps = open("/proc/self/maps").read()
    assert "libmpshuffle.so" in maps and "libmpemu" not in maps


def test_outputs_do_not_depend_on_batch_position_or_lane_mode(mp, coracle):
    """a size-independent property at a size the oracle cannot cover: 3 000 proofs (medium split by default), the digest of all outputs
    is the same on the throughput split, with window lanes 1 .. 16, and with the verify calls pipelined; every proof verifies"""
    import torch
    cv, m, n, B = "stark", 2, 26, 3000
    eng = mp._native.Engine(cv, 0)
    g0 = coracle.gen_inputs(cv, m, n, 5500)
    t = eng.table(m, n, g0["params"], g0["pk"], fb_bits=16)
    gpu = torch.device("cuda", 0)
    gen = torch.Generator(device=gpu)
    gen.manual_seed(99)
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
    digests = set()
    for split, vsp, pipe in ((-1, None, 0), (0, None, 0), (2, 1, 0), (2, 16, 0), (4, 5, 1), (-1, None, 2)):
        if vsp is not None:
            t.set_plan_params(split, 4, 32, 8, 16, vsp)
        t.set_work_split(split)
        t.set_pipeline(pipe)
        for _ in range(2):
            t.shuffle_and_remask_batch_dev(B, decks.data_ptr(), rho.data_ptr(), perm.data_ptr(), seeds.data_ptr(), od.data_ptr(), op.data_ptr(),
                                           sp.data_ptr())
            t.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv.data_ptr())
        eng.sync()
        assert int(sp.abs().sum().item()) == 0 and int(sv.abs().sum().item()) == 0
        h = hashlib.sha256()
        h.update(od.cpu().numpy().tobytes())
        h.update(op.cpu().numpy().tobytes())
        digests.add(h.hexdigest())
    assert len(digests) == 1
    # one of them against the oracle
    b = 1234
    tb = lambda x: bytes(x.cpu().numpy().tobytes())
    ed, ep = coracle.shuffle_and_remask(cv, m, n, g0["params"], g0["pk"], g0["deck"], tb(rho[b]), [int(v) for v in perm[b].cpu().tolist()], tb(seeds[b]))
    assert tb(od[b]) == ed and tb(op[b]) == ep
    t.set_pipeline(0)
    t.close()
    eng.close()


class _TorchMem:
    def __init__(self):
        import torch
        self.torch, self.gpu = torch, torch.device("cuda", 0)

    def put(self, b):
        t = self.torch.frombuffer(bytearray(b), dtype=self.torch.uint8).to(self.gpu)
        return t, t.data_ptr()

    def new(self, nbytes):
        t = self.torch.full((max(nbytes, 4),), 0x5A, dtype=self.torch.uint8, device=self.gpu)
        return t, t.data_ptr()

    def get(self, h, nbytes):
        return bytes(h[:nbytes].cpu().numpy().tobytes())


@pytest.mark.parametrize("name", ["shuffle_stark_m2_n26_s7.json", "shuffle_stark_m4_n13_s9.json", "shuffle_bn254_m2_n4_s3.json",
                                  "shuffle_secp256k1_m3_n3_s5.json", "shuffle_bls12_377_m2_n3_s13.json"])
def test_device_decompression_matches_oracle(mp, name):
    """mp_deck_deserialize_dev / mp_points_deserialize_dev on the MI355X: arkworks-compressed decks and points -> wire v1 in HBM (windowed
    square root, one lane per point) against the oracle's encoder / decoder: every golden deck, random points, non-residues,
    non-canonical x, malformed flags, points outside the prime-order subgroup"""
    import os
    from conftest import GOLDEN, load_json
    from decompress_cases import run_decompress_cases
    g = load_json(os.path.join(GOLDEN, name))
    eng = mp._native.Engine(g["curve"], 0)
    run_decompress_cases(eng, _TorchMem(), g["curve"], g, n_random=200)
    eng.close()


def test_decompressed_decks_go_straight_into_the_prover(mp, coracle):
    """arkworks bytes -> HBM -> mp_deck_deserialize_dev -> mp_shuffle_and_remask_batch_dev / mp_verify_shuffle_batch_dev without a host
    square root: same proofs as from the wire decks"""
    import torch
    import ark_canonical as ac
    import mp_oracle as po
    cv, m, n, B = "stark", 2, 26, 4
    eng = mp._native.Engine(cv, 0)
    ins, args = _inputs(coracle, cv, m, n, B, 5600)
    g0 = ins[0]
    t = eng.table(m, n, g0["params"], g0["pk"])
    exp = _expected(coracle, cv, m, n, g0, ins)
    gpu = torch.device("cuda", 0)
    dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(gpu)
    with po.curve_ctx(po.CURVES[cv]):
        ser = b"".join(ac.enc_deck(po.CURVES[cv], po.deck_from_bytes(g["deck"])) for g in ins)
    d_ser, d_decks = dev(ser), torch.empty(len(args[0]), dtype=torch.uint8, device=gpu)
    d_st = torch.full((B,), 9, dtype=torch.int32, device=gpu)
    eng.deck_deserialize_dev(B, m * n, d_ser.data_ptr(), d_decks.data_ptr(), d_st.data_ptr())
    rho, seeds = dev(args[1]), dev(args[3])
    perm = torch.tensor(args[2], dtype=torch.int32, device=gpu)
    od, op = torch.empty(len(exp[0]), dtype=torch.uint8, device=gpu), torch.empty(len(exp[1]), dtype=torch.uint8, device=gpu)
    sp, sv = torch.full((B,), 9, dtype=torch.int32, device=gpu), torch.full((B,), 9, dtype=torch.int32, device=gpu)
    t.shuffle_and_remask_batch_dev(B, d_decks.data_ptr(), rho.data_ptr(), perm.data_ptr(), seeds.data_ptr(), od.data_ptr(), op.data_ptr(), sp.data_ptr())
    t.verify_shuffle_batch_dev(B, d_decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv.data_ptr())
    eng.sync()
    assert d_st.cpu().tolist() == [0] * B and sp.cpu().tolist() == [0] * B and sv.cpu().tolist() == [0] * B
    assert bytes(d_decks.cpu().numpy().tobytes()) == args[0]
    assert bytes(od.cpu().numpy().tobytes()) == exp[0] and bytes(op.cpu().numpy().tobytes()) == exp[1]
    t.close()
    eng.close()


def test_pipelined_verification_with_keys_and_key_sets(mp, coracle):
    """pipelined verify calls under one aggregate key per proof (mp_verify_shuffle_batch_keys_dev) and under a key set
    (mp_verify_shuffle_batch_keyset_dev: the keys are gathered into a buffer of the verify lane's own while a keyed prove call of the next
    batch gathers into the main lane's): status words as from the waiting calls, a proof verified under a neighbour's key fails by name"""
    import torch
    cv, m, n, B, K = "stark", 2, 26, 6, 4
    eng = mp._native.Engine(cv, 0)
    g0 = coracle.gen_inputs(cv, m, n, 5700)
    table = eng.table(m, n, g0["params"], None)               # parameters only: keyed entry points
    keys = [coracle.gen_inputs(cv, m, n, 5710 + k)["pk"] for k in range(K)]
    ks = table.keyset(b"".join(keys))
    ins = [coracle.gen_inputs(cv, m, n, 5720 + b) for b in range(B)]
    gpu = torch.device("cuda", 0)
    dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(gpu)
    decks, rho, seeds = dev(b"".join(g["deck"] for g in ins)), dev(b"".join(g["rho"] for g in ins)), dev(b"".join(g["prover_seed"] for g in ins))
    perm = torch.tensor([v for g in ins for v in g["perm"]], dtype=torch.int32, device=gpu)
    kidx = torch.tensor([b % K for b in range(B)], dtype=torch.int32, device=gpu)
    kidx_bad = torch.tensor([(b + 1) % K for b in range(B)], dtype=torch.int32, device=gpu)
    kwire = dev(b"".join(keys[b % K] for b in range(B)))
    dsz, psz = len(g0["deck"]), table.proof_bytes
    od = [torch.empty(B * dsz, dtype=torch.uint8, device=gpu) for _ in range(2)]
    op = [torch.empty(B * psz, dtype=torch.uint8, device=gpu) for _ in range(2)]
    sp = torch.full((B,), 77, dtype=torch.int32, device=gpu)
    table.set_work_split(2)                                    # a split that screens with the merged equation
    table.set_pipeline(1)
    sts = [torch.full((B,), 77, dtype=torch.int32, device=gpu) for _ in range(4)]
    for i in range(2):                                         # prove (key set) -> verify by index -> verify by explicit key, pipelined
        table.shuffle_and_remask_batch_keyset_dev(ks, B, kidx.data_ptr(), decks.data_ptr(), rho.data_ptr(), perm.data_ptr(), seeds.data_ptr(),
                                                  od[i].data_ptr(), op[i].data_ptr(), sp.data_ptr())
        table.verify_shuffle_batch_keyset_dev(ks, B, (kidx if i == 0 else kidx_bad).data_ptr(), decks.data_ptr(), od[i].data_ptr(), op[i].data_ptr(),
                                              sts[2 * i].data_ptr())
        table.verify_shuffle_batch_keys_dev(B, kwire.data_ptr(), decks.data_ptr(), od[i].data_ptr(), op[i].data_ptr(), sts[2 * i + 1].data_ptr())
    eng.sync()
    assert sp.cpu().tolist() == [0] * B
    assert sts[0].cpu().tolist() == [0] * B and sts[1].cpu().tolist() == [0] * B and sts[3].cpu().tolist() == [0] * B
    assert [eng.check_name(v) for v in sts[2].cpu().tolist()] == ["Multi-Exponentiation Argument (4)"] * B or all(v > 0 for v in sts[2].cpu().tolist())
    assert bytes(od[0].cpu().numpy().tobytes()) == bytes(od[1].cpu().numpy().tobytes())
    b = 3
    ed, ep = coracle.shuffle_and_remask(cv, m, n, g0["params"], keys[b % K], ins[b]["deck"], ins[b]["rho"], ins[b]["perm"], ins[b]["prover_seed"])
    assert bytes(od[1][b * dsz:(b + 1) * dsz].cpu().numpy().tobytes()) == ed and bytes(op[1][b * psz:(b + 1) * psz].cpu().numpy().tobytes()) == ep
    table.set_pipeline(0)
    ks.close()
    table.close()
    eng.close()


@pytest.mark.parametrize("cv,m,n", [("stark", 2, 26), ("secp256k1", 2, 7), ("bn254", 3, 5)])
def test_group_verification_status_words(mp, coracle, cv, m, n):
    """group verification (mp_set_group_verify): the screen of a batch is one bucket-method equation per group of proofs -- status words equal
    those of the per-proof screen and of per-equation verification for honest batches, one bad response scalar, one bad input point,
    every proof against a neighbour's deck; group sizes that divide the batch, one that does not (falls back to the per-proof screen);
    pipelined too.  The oracle verifies / rejects the same proofs."""
    import torch
    B = 12
    eng = mp._native.Engine(cv, 0)
    ins, args = _inputs(coracle, cv, m, n, B, 5800)
    g0 = ins[0]
    t = eng.table(m, n, g0["params"], g0["pk"])
    t.set_work_split(0)
    t.set_group_verify(0, 0)
    out = t.shuffle_and_remask_batch(*args)
    assert (out[0], out[1]) == _expected(coracle, cv, m, n, g0, ins)
    dsz, psz = len(g0["deck"]), t.proof_bytes
    bad_p = bytearray(out[1])
    bad_p[8 * psz - 31] ^= 2                              # proof 7: its last response scalar
    bad_d = bytearray(out[0])
    bad_d[3 * dsz + 5] ^= 1                               # shuffled deck 3: not a curve point any more
    cases = {"good": (out[0], out[1]), "badproof": (out[0], bytes(bad_p)), "badpoint": (bytes(bad_d), out[1]),
             "rotated": (out[0][dsz:] + out[0][:dsz], out[1])}
    want = {k: t.verify_shuffle_batch(args[0], d, p) for k, (d, p) in cases.items()}
    assert want["good"] == [0] * B and want["badproof"][7] > 0 and sum(1 for v in want["badproof"] if v) == 1
    assert want["badpoint"][3] < 0 and all(v > 0 for v in want["rotated"])
    assert coracle.verify_shuffle(cv, m, n, g0["params"], g0["pk"], ins[7]["deck"], out[0][7 * dsz:8 * dsz], bytes(bad_p[7 * psz:8 * psz])) == want["badproof"][7]
    t.set_merged_verify(False)
    assert {k: t.verify_shuffle_batch(args[0], d, p) for k, (d, p) in cases.items()} == want
    t.set_merged_verify(True)
    for links, expect_group in ((4, 4), (6, 6), (12, 12), (3, 3), (5, 6), (7, 6)):
        t.set_group_verify(links * (4 * m * n + 11 * m + 8), 0)
        assert t.group_size(B) == expect_group
        eng.profile_enable(True)
        got, looked = {}, {}
        for k, (d, p) in cases.items():
            before = t.reverified_count()
            got[k] = t.verify_shuffle_batch(args[0], d, p)
            looked[k] = t.reverified_count() - before
        rep = eng.profile_report()
        eng.profile_enable(False)
        assert got == want, links
        assert "k_chain_scalars" in rep and "k_bucket_msm" in rep
        # (round 5) exactly the members of the failing groups took the per-equation pass: one group for one bad proof / one bad point
        assert looked == {"good": 0, "badproof": expect_group, "badpoint": expect_group, "rotated": B}, (links, looked)
    t.set_group_verify(4 * (4 * m * n + 11 * m + 8), 60)    # batch below the minimum (60 x 52 / N proofs): per-proof screen
    assert t.group_size(B) == 0
    assert t.verify_shuffle_batch(args[0], out[0], out[1]) == [0] * B
    # pipelined: the group pass is the deferred screen
    gpu = torch.device("cuda", 0)
    dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(gpu)
    t.set_group_verify(4 * (4 * m * n + 11 * m + 8), 0)
    t.set_pipeline(1)
    decks = dev(args[0])
    held = []
    for k, (d, p) in cases.items():
        st = torch.full((B,), 55, dtype=torch.int32, device=gpu)
        dd, pp_ = dev(d), dev(p)
        held.append((k, st, dd, pp_))
        t.verify_shuffle_batch_dev(B, decks.data_ptr(), dd.data_ptr(), pp_.data_ptr(), st.data_ptr())
    eng.sync()
    for k, st, _, _ in held:
        assert st.cpu().tolist() == want[k], k
    t.set_pipeline(0)
    t.close()
    eng.close()


def test_group_verification_at_full_size(mp, coracle):
    """a size the oracle cannot cover: 8 192 proofs (1 024 groups of 8, then 64 groups of 128 -- 30 464 points each, 10-bit windows): every proof accepted; ONE tampered proof anywhere makes exactly that
    proof fail with the reference's check name, as without groups"""
    import torch
    cv, m, n, B = "stark", 2, 26, 8192
    eng = mp._native.Engine(cv, 0)
    g0 = coracle.gen_inputs(cv, m, n, 5900)
    t = eng.table(m, n, g0["params"], g0["pk"], fb_bits=16)
    assert t.group_size(B) == 8 and t.group_size(16384) == 1024 and t.group_size(65536) == 1024 and t.group_size(262144) == 1024      # (no fewer than 945 groups of the rounds 4-5 kind; round 6: the split pipeline's equations from 16 384 proofs on)
    gpu = torch.device("cuda", 0)
    gen = torch.Generator(device=gpu)
    gen.manual_seed(5)
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
    sizes = [(None, 8), ((30464, 0), 128), ((7616, 0), 32), ((30464, 0), 128)]      # 8-bit, 10-bit, 9-bit windows (the last one with 9 forced on 30 464 points)
    for k, (cfg, L) in enumerate(sizes):
        if cfg:
            t.set_group_verify(*cfg)
        t.set_bucket_bits(9 if k == 3 else 0)
        assert t.group_size(B) == L
        sv.fill_(55)
        t.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv.data_ptr())
        eng.sync()
        assert int(sp.abs().sum().item()) == 0 and int(sv.abs().sum().item()) == 0, L
    op[4321, t.proof_bytes - 31] ^= 2
    od[77, 0:64] = od[78, 0:64]                             # card 0 of deck 77 replaced by a neighbour's: a valid point, a wrong statement
    for k, (cfg, L) in enumerate(sizes):
        t.set_group_verify(*(cfg or (30464, 6144)))
        t.set_bucket_bits(9 if k == 3 else 0)
        sv.fill_(55)
        before = t.reverified_count()
        eng.profile_enable(True)
        t.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv.data_ptr())
        eng.sync()
        rep = eng.profile_report()
        eng.profile_enable(False)
        st = sv.cpu().tolist()
        assert [i for i, v in enumerate(st) if v] == [77, 4321], L
        assert eng.check_name(st[77]) == "Hadamard Product (5.1)" and st[4321] > 0
        # (round 5) the two failing groups' members were looked at again -- 2 L proofs, not 8 192 (launch sizes of the per-equation pass)
        assert t.reverified_count() - before == 2 * L and eng.last_profile_items["k_verdict"] == 2 * L, (L, eng.last_profile_items)
        assert rep["k_bucket_msm"][0] == 1
    # ... through sub-groups of 16 first when asked to (mp_set_group_refine): 2 x 16 proofs reach the equations
    t.set_bucket_bits(0)
    t.set_group_verify(30464, 0)
    t.set_group_refine(16 * 238, 1)
    before = t.reverified_count()
    sv.fill_(55)
    t.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv.data_ptr())
    eng.sync()
    assert [i for i, v in enumerate(sv.cpu().tolist()) if v] == [77, 4321] and t.reverified_count() - before == 32
    # 1 % of the batch tampered, evenly spread (82 proofs, in most of the 64 groups): exactly those rejected; sub-groups by default
    # (>= 128 of them), pipelined too
    t.set_group_refine(0, 0)
    t.set_group_adapt(False)                                # (groups that shrink under sustained rejection: tests/test_gpu_round5.py)
    t.shuffle_and_remask_batch_dev(B, decks.data_ptr(), rho.data_ptr(), perm.data_ptr(), seeds.data_ptr(), od.data_ptr(), op.data_ptr(), sp.data_ptr())
    eng.sync()
    idx = torch.arange(82, device=gpu) * 99 + 50
    op[idx, t.proof_bytes - 31] ^= 2
    want = torch.zeros(B, dtype=torch.bool, device=gpu)
    want[idx] = True
    for depth in (0, 2):
        t.set_pipeline(depth)
        before = t.reverified_count()
        svs = [torch.full((B,), 55, dtype=torch.int32, device=gpu) for _ in range(3)]
        for sv_ in svs:
            t.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv_.data_ptr())
        eng.sync()
        for sv_ in svs:
            assert torch.equal(sv_ != 0, want), depth
        looked = (t.reverified_count() - before) // 3
        assert 82 <= looked <= 82 * 16, (depth, looked)       # the members of the failing sub-groups of 16
    t.set_pipeline(0)
    t.close()
    eng.close()


def test_one_full_group_against_the_oracle(mp, coracle):
    """the group screen at its production size against the ORACLE, not only against properties: 8 192 proofs of a 52-card deck in 64 groups of
    128 (30 464 points per equation, 10-bit windows); the 128 members of one group -- one of them tampered -- are verified by the CPU oracle
    proof by proof next to the engine's verdicts: same accept / reject, same check name"""
    import torch
    cv, m, n, B, L = "stark", 2, 26, 8192, 128
    eng = mp._native.Engine(cv, 0)
    g0 = coracle.gen_inputs(cv, m, n, 5950)
    t = eng.table(m, n, g0["params"], g0["pk"], fb_bits=16)
    t.set_group_verify(30464, 0)
    assert t.group_size(B) == L
    gpu = torch.device("cuda", 0)
    gen = torch.Generator(device=gpu)
    gen.manual_seed(11)
    N = m * n
    decks = torch.frombuffer(bytearray(g0["deck"]), dtype=torch.uint8).to(gpu).repeat(B, 1).contiguous()
    rho = torch.randint(0, 256, (B, N, 32), dtype=torch.uint8, device=gpu, generator=gen)
    rho[:, :, 31] &= 7
    perm = torch.argsort(torch.rand(B, N, device=gpu, generator=gen), dim=1).to(torch.int32).contiguous()
    seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device=gpu, generator=gen)
    od = torch.empty(B, len(g0["deck"]), dtype=torch.uint8, device=gpu)
    op = torch.empty(B, t.proof_bytes, dtype=torch.uint8, device=gpu)
    sp = torch.empty(B, dtype=torch.int32, device=gpu)
    sv = torch.full((B,), 55, dtype=torch.int32, device=gpu)
    t.shuffle_and_remask_batch_dev(B, decks.data_ptr(), rho.data_ptr(), perm.data_ptr(), seeds.data_ptr(), od.data_ptr(), op.data_ptr(), sp.data_ptr())
    eng.sync()
    T, grp = B // L, 13
    members = [j * T + grp for j in range(L)]                # lane of (member j, group t) = j T + t
    op[members[40], (11 * m + 8) * 64 + 10 * 32 + 2] ^= 1     # one byte of a response scalar of member 40 (a tampered POINT would be refused as an encoding error before any equation is looked at)
    t.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv.data_ptr())
    eng.sync()
    st = sv.cpu().tolist()
    assert int(sp.abs().sum().item()) == 0 and [i for i, v in enumerate(st) if v] == [members[40]]
    deck_b = bytes(g0["deck"])
    od_c, op_c = od[members].cpu().numpy(), op[members].cpu().numpy()
    for k, b in enumerate(members):
        assert coracle.verify_shuffle(cv, m, n, g0["params"], g0["pk"], deck_b, od_c[k].tobytes(), op_c[k].tobytes()) == st[b], (k, b)
    t.close()
    eng.close()


def test_group_verification_of_large_decks(mp, coracle):
    """512-card decks (m = 4, n = 128): a proof's own merged equation (2 100 points) already runs on the bucket kernel; groups of 4 and 2
    such proofs (8 400 points through 9-bit windows, 4 200 through 8- and forced 10-bit ones) give the status words of the per-proof
    screen -- all accepted; one bad response scalar and one swapped deck named; the first proof's bytes are the oracle's"""
    cv, m, n, B = "stark", 4, 128, 8
    eng = mp._native.Engine(cv, 0)
    ins, args = _inputs(coracle, cv, m, n, B, 6100)
    g0 = ins[0]
    t = eng.table(m, n, g0["params"], g0["pk"], fb_bits=8)
    out = t.shuffle_and_remask_batch(*args)
    dsz, psz = len(g0["deck"]), t.proof_bytes
    assert (out[0][:dsz], out[1][:psz]) == _expected(coracle, cv, m, n, g0, ins[:1])
    per = 4 * m * n + 11 * m + 8
    bad_p = bytearray(out[1])
    bad_p[6 * psz - 31] ^= 2                              # proof 5: its last response scalar
    swapped = out[0][dsz:2 * dsz] + out[0][:dsz] + out[0][2 * dsz:]      # decks 0 and 1 exchanged
    cases = {"good": (out[0], out[1]), "badproof": (out[0], bytes(bad_p)), "swapped": (swapped, out[1])}
    t.set_group_verify(0, 0)
    want = {k: t.verify_shuffle_batch(args[0], d, p) for k, (d, p) in cases.items()}
    assert want["good"] == [0] * B and [i for i, v in enumerate(want["badproof"]) if v] == [5] and [i for i, v in enumerate(want["swapped"]) if v] == [0, 1]
    for links, bits in ((4, 0), (2, 0), (2, 10)):
        t.set_bucket_bits(bits)
        t.set_group_verify(links * per, 0)
        assert t.group_size(B) == links
        eng.profile_enable(True)
        got = {k: t.verify_shuffle_batch(args[0], d, p) for k, (d, p) in cases.items()}
        rep = eng.profile_report()
        eng.profile_enable(False)
        assert got == want, (links, bits)
        assert "k_chain_scalars" in rep and "k_bucket_msm" in rep
    t.close()
    eng.close()


def test_bucket_msm_at_the_term_limit(mp, coracle):
    """ONE multi-scalar multiplication of 65 535 terms -- the most a bucket job took until round 5 (11-bit windows then; 12-bit ones on the
    split pipeline now: tests/test_gpu_round6.py goes to the new limit) -- and of 40 000 (11-bit), 12 000 (10-bit) and 6 000 terms (9-bit),
    against the oracle: the points repeat with
    period 509, so the oracle's MSM of 509 terms with the summed scalars is the same group element"""
    import random
    cv = "stark"
    q = 0x0800000000000010ffffffffffffffffb781126dcae7b2321e66a241adc64d2f
    eng = mp._native.Engine(cv, 0)
    g0 = coracle.gen_inputs(cv, 2, 3, 6200)
    t = eng.table(2, 3, g0["params"], g0["pk"])
    per = 509
    base = eng.setup(2, per - 3, bytes([11] * 32))[:64 * per]
    assert len(base) == 64 * per
    random.seed(62)
    for K in (65535, 40000, 12000, 6000):
        sc = [random.randrange(q) for _ in range(K)]
        sc[:4] = [0, 1, q - 1, (1 << 251) + 1]
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
        assert ("k_bucket_acc" if K >= 50000 else "k_bucket_msm") in rep, K      # (round 6: 12-bit windows on the split pipeline from 50 000 terms on)
    t.close()
    eng.close()
