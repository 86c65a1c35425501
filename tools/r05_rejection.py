#!/usr/bin/env python3
"""round 5: what rejection costs, by refinement strategy.  B proofs of a 52-card deck are proved once; the verifier is then timed on
the honest batch, with ONE tampered proof and with 1 % tampered proofs (evenly spread), for several sub-group sizes of the refinement
(mp_set_group_refine) and with the refinement off (members of failing groups straight to the per-equation pass).
usage: python tools/r05_rejection.py [B]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
mp = importlib.import_module("mental-poker_amd")
import coracle as co
m, n = 2, 26
N = m * n
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
g = co.gen_inputs("stark", m, n, 7)
eng = mp.Engine("stark", 0)
t = eng.table(m, n, g["params"], g["pk"], fb_bits=20)
gpu = torch.device("cuda", 0)
gen = torch.Generator(device=gpu); gen.manual_seed(3)
decks = torch.frombuffer(bytearray(g["deck"]), dtype=torch.uint8).to(gpu).repeat(B, 1).contiguous()
rho = torch.randint(0, 256, (B, N, 32), dtype=torch.uint8, device=gpu, generator=gen); rho[:, :, 31] &= 7
perm = torch.argsort(torch.rand(B, N, device=gpu, generator=gen), dim=1).to(torch.int32).contiguous()
seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device=gpu, generator=gen)
od = torch.empty(B, len(g["deck"]), dtype=torch.uint8, device=gpu); op = torch.empty(B, t.proof_bytes, dtype=torch.uint8, device=gpu)
sp = torch.empty(B, dtype=torch.int32, device=gpu); sv = torch.empty(B, dtype=torch.int32, device=gpu)
t.shuffle_and_remask_batch_dev(B, decks.data_ptr(), rho.data_ptr(), perm.data_ptr(), seeds.data_ptr(), od.data_ptr(), op.data_ptr(), sp.data_ptr())
eng.sync()
assert int(sp.abs().sum()) == 0
good = op.clone()
def verify_ms(reps=3):
    t.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv.data_ptr()); eng.sync()
    looked = t.reverified_count()
    t0 = time.perf_counter()
    for _ in range(reps):
        t.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv.data_ptr())
    eng.sync()
    return 1e3 * (time.perf_counter() - t0) / reps, (t.reverified_count() - looked) // reps
print("B = %d, group size %d; verify only" % (B, t.group_size(B)))
base, _ = verify_ms()
print("honest batch: %.1f ms" % base)
per = 4 * N + 11 * m + 8
for nbad in (1, max(1, B // 1000), max(1, B // 100), max(1, B // 20)):
    idx = torch.arange(nbad, device=gpu, dtype=torch.int64) * (B // nbad) + (B // nbad) // 2
    op.copy_(good); op[idx, t.proof_bytes - 31] ^= 2
    torch.cuda.synchronize()
    for name, pts, mn in (("sub-groups of 16 (default)", 0, 0), ("sub-groups of 8", 8 * per, 0), ("sub-groups of 32", 32 * per, 0), ("sub-groups of 4", 4 * per, 0),
                          ("no sub-groups", 0, 1 << 30)):
        t.set_group_refine(pts, mn)
        ms, looked = verify_ms()
        bad = (sv != 0)
        assert int(bad.sum()) == nbad and bool(bad[idx].all())
        print("%6d tampered, %-28s verify %.1f ms (+%.1f), %d proofs through the per-equation pass" % (nbad, name, ms, ms - base, looked))
