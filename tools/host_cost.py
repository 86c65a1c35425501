#!/usr/bin/env python3
"""tools/host_cost.py -- host time of one prove / verify call (enqueue cost of the ~100 launches behind each) against the step time.
Usage: python tools/host_cost.py [B ...]   (pipelined depth 4: the calls never wait for the GPU)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
mp = importlib.import_module("mental-poker_amd")
m, n = 2, 26
N = m * n
eng = mp.Engine("stark", device=0)
PB = eng.point_bytes; CB = 2 * PB
params = eng.setup(m, n, bytes([1] * 32)); pk = eng.setup(m, 2, bytes([2] * 32))[:PB]
base_deck = eng.setup(m, 2 * N - 3, bytes([3] * 32))
table = eng.table(m, n, params, pk, fb_bits=16)
gpu = torch.device("cuda", 0)
Bs = [int(a) for a in sys.argv[1:]] or [1, 64, 1024]
Bmax = max(Bs)
gen = torch.Generator(device=gpu); gen.manual_seed(7)
factors = torch.randint(0, 256, (Bmax, N, 32), dtype=torch.uint8, device=gpu, generator=gen); factors[:, :, 31] &= 7
perms = torch.argsort(torch.rand(Bmax, N, device=gpu, generator=gen), dim=1).to(torch.int32).contiguous()
seeds = torch.randint(0, 256, (Bmax, 32), dtype=torch.uint8, device=gpu, generator=gen)
decks = torch.frombuffer(bytearray(base_deck), dtype=torch.uint8).to(gpu).repeat(Bmax, 1).contiguous()
D = 4
table.set_pipeline(D)
od = [torch.empty(Bmax, N * CB, dtype=torch.uint8, device=gpu) for _ in range(D + 1)]
op = [torch.empty(Bmax, table.proof_bytes, dtype=torch.uint8, device=gpu) for _ in range(D + 1)]
sp = torch.empty(Bmax, dtype=torch.int32, device=gpu); sv = torch.empty(Bmax, dtype=torch.int32, device=gpu)
torch.cuda.synchronize()
for B in Bs:
    for rep in range(2):
        tp = tv = 0.0
        K = 40
        eng.sync()
        t0 = time.perf_counter()
        for k in range(K):
            i = k % (D + 1)
            a = time.perf_counter()
            table.shuffle_and_remask_batch_dev(B, decks.data_ptr(), factors.data_ptr(), perms.data_ptr(), seeds.data_ptr(), od[i].data_ptr(), op[i].data_ptr(), sp.data_ptr())
            b = time.perf_counter()
            table.verify_shuffle_batch_dev(B, decks.data_ptr(), od[i].data_ptr(), op[i].data_ptr(), sv.data_ptr())
            c = time.perf_counter()
            tp += b - a; tv += c - b
        t1 = time.perf_counter()
        eng.sync()
        t2 = time.perf_counter()
    assert int(sp[:B].abs().sum().item()) == 0 and int(sv[:B].abs().sum().item()) == 0
    print("B=%d  prove call %.3f ms  verify call %.3f ms  host/step %.3f ms  step (synced) %.3f ms" % (B, 1e3 * tp / K, 1e3 * tv / K, 1e3 * (t1 - t0) / K, 1e3 * (t2 - t0) / K), flush=True)
