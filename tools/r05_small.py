#!/usr/bin/env python3
"""round 5: where a batch of 1 024 (or B) device-resident proofs spends its time -- wall time of prove + verify (serial calls, as
bench.py's batch_curve) against the sum of the kernels' HIP-event times, launches, and the largest kernels of either call.
usage: python tools/r05_small.py [B ...]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
mp = importlib.import_module("mental-poker_amd")
import coracle as co
m, n = 2, 26
N = m * n
g = co.gen_inputs("stark", m, n, 7)
eng = mp.Engine("stark", 0)
t = eng.table(m, n, g["params"], g["pk"], fb_bits=16)
gpu = torch.device("cuda", 0)
GROUP = None
argv = sys.argv[1:]
if "--group" in argv:                     # --group POINTS MIN_BATCH: mp_set_group_verify for the run
    i = argv.index("--group")
    GROUP = (int(argv[i + 1]), int(argv[i + 2]))
    argv = argv[:i] + argv[i + 3:]
    t.set_group_verify(*GROUP)
    print("group verification: %d points per equation from %d proofs on" % GROUP)
if "--plan" in argv:                      # --plan SPLIT FCH VCH GRP NCH VSP: mp_set_plan_params for the run
    i = argv.index("--plan")
    pp = [int(x) for x in argv[i + 1:i + 7]]
    argv = argv[:i] + argv[i + 7:]
    t.set_plan_params(*pp)
    print("plan %d: %s" % (pp[0], pp[1:]))
QUIET = "--quiet" in argv
argv = [a for a in argv if a != "--quiet"]
for B in [int(a) for a in argv] or [1024]:
    gen = torch.Generator(device=gpu); gen.manual_seed(3)
    decks = torch.frombuffer(bytearray(g["deck"]), dtype=torch.uint8).to(gpu).repeat(B, 1).contiguous()
    rho = torch.randint(0, 256, (B, N, 32), dtype=torch.uint8, device=gpu, generator=gen); rho[:, :, 31] &= 7
    perm = torch.argsort(torch.rand(B, N, device=gpu, generator=gen), dim=1).to(torch.int32).contiguous()
    seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device=gpu, generator=gen)
    od = torch.empty(B, len(g["deck"]), dtype=torch.uint8, device=gpu); op = torch.empty(B, t.proof_bytes, dtype=torch.uint8, device=gpu)
    sp = torch.empty(B, dtype=torch.int32, device=gpu); sv = torch.empty(B, dtype=torch.int32, device=gpu)
    torch.cuda.synchronize()
    def prove(): t.shuffle_and_remask_batch_dev(B, decks.data_ptr(), rho.data_ptr(), perm.data_ptr(), seeds.data_ptr(), od.data_ptr(), op.data_ptr(), sp.data_ptr())
    def verify(): t.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv.data_ptr())
    for _ in range(3):
        prove(); verify()
    eng.sync()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        prove()
    eng.sync(); tp = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K):
        verify()
    eng.sync(); tv = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K):
        prove(); verify()
    eng.sync(); tb = (time.perf_counter() - t0) / K
    assert int(sp.abs().sum()) == 0 and int(sv.abs().sum()) == 0
    print("group size %d" % t.group_size(B))
    print("B=%d: prove %.3f ms, verify %.3f ms, prove+verify %.3f ms -> %.0f proofs/s" % (B, 1e3 * tp, 1e3 * tv, 1e3 * tb, B / tb))
    for name, fn in (() if QUIET else (("prove", prove), ("verify", verify))):
        eng.profile_enable(True); fn(); rep = eng.profile_report(); eng.profile_enable(False)
        print("  %s: kernel sum %.3f ms in %d launches; %s" % (name, sum(v[1] for v in rep.values()), sum(v[0] for v in rep.values()),
              ", ".join("%s x%d %.3f" % (k, v[0], v[1]) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])[:10])))
