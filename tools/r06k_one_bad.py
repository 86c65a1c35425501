#!/usr/bin/env python3
"""what ONE tampered proof in 262 144 costs the verify call under the round-6 equations of 1 024 proofs, by refinement strategy"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
mp = importlib.import_module("mental-poker_amd")
gpu = torch.device("cuda", 0)
m, n, B = 2, 26, int(sys.argv[1]) if len(sys.argv) > 1 else 262144
N = m * n
eng = mp.Engine("stark", device=0)
PB = eng.point_bytes; CB = 2 * PB
params = eng.setup(m, n, bytes([1] * 32)); pk = eng.setup(m, 2, bytes([2] * 32))[:PB]
base_deck = eng.setup(m, 2 * N - 3, bytes([3] * 32))
table = eng.table(m, n, params, pk, fb_bits=21)
gen = torch.Generator(device=gpu); gen.manual_seed(77)
factors = torch.randint(0, 256, (B, N, 32), dtype=torch.uint8, device=gpu, generator=gen); factors[:, :, 31] &= 0x07
perms = torch.argsort(torch.rand(B, N, device=gpu, generator=gen), dim=1).to(torch.int32).contiguous()
seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device=gpu, generator=gen)
decks = torch.frombuffer(bytearray(base_deck), dtype=torch.uint8).to(gpu).repeat(B, 1).contiguous()
od = torch.empty(B, N * CB, dtype=torch.uint8, device=gpu); op = torch.empty(B, table.proof_bytes, dtype=torch.uint8, device=gpu)
sp = torch.empty(B, dtype=torch.int32, device=gpu); sv = torch.empty(B, dtype=torch.int32, device=gpu)
torch.cuda.synchronize()
table.shuffle_and_remask_batch_dev(B, decks.data_ptr(), factors.data_ptr(), perms.data_ptr(), seeds.data_ptr(), od.data_ptr(), op.data_ptr(), sp.data_ptr())
eng.sync()
def verify():
    table.verify_shuffle_batch_dev(B, decks.data_ptr(), od.data_ptr(), op.data_ptr(), sv.data_ptr())
def timed(reps=5):
    verify(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(reps): verify()
    eng.sync()
    return (time.perf_counter() - t0) / reps * 1e3
print("honest            %.2f ms" % timed())
op[B // 3, -40] ^= 1
for name, refine in (("per-equation pass over the 1 024", (0, 128)), ("sub-groups of 128 first", (30464, 1)), ("sub-groups of 64 first", (15232, 1)), ("sub-groups of 32 first", (7616, 1))):
    table.set_group_refine(*refine)
    before = table.reverified_count()
    ms = timed()
    looked = (table.reverified_count() - before) // 6
    eng.profile_enable(True); verify(); eng.sync(); rep = eng.profile_report(); eng.profile_enable(False)
    extra = {k: round(v[1], 2) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1]) if k in ("k_var_msm", "k_table", "k_bucket_msm", "k_gather_rows", "k_verify_fs", "k_fixed_msm", "k_var_msm_q", "k_recode", "k_normalize", "k_combine")}
    print("%-34s %.2f ms  (%d proofs re-verified)  %s" % (name, ms, looked, json.dumps(extra)))
    assert torch.nonzero(sv).flatten().tolist() == [B // 3]
