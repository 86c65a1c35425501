#!/usr/bin/env python3
"""tools/r06_group_sweep.py -- the verifier's screen on group equations of different sizes and window widths (round 6: the workgroup
bucket kernel, 12- and 13-bit windows), one table, one process, every configuration timed back to back on the same box:

  python tools/r06_group_sweep.py --batches 262144 --configs "30464:0,121856:12,243712:13,243712:12,487424:13"

a configuration is points_per_group:bucket_bits[:min_batch[:split_min_bits]] (bits 0 = by size).  Output: one JSON line per (batch, configuration) with the step's
proofs/s and the per-kernel milliseconds of one step; a table on stderr.  Needs a GPU (the engine has no CPU path).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="262144")
    ap.add_argument("--configs", default="30464:0,243712:13")
    ap.add_argument("--m", type=int, default=2)
    ap.add_argument("--n", type=int, default=26)
    ap.add_argument("--curve", default="stark")
    ap.add_argument("--fb-bits", type=int, default=21)
    ap.add_argument("--min-batch", type=int, default=6144)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=1, help="passes over the list of configurations (boxes and clocks drift: alternate)")
    ap.add_argument("--verify-only", action="store_true", help="time the verify call alone (the prover's outputs are reused)")
    args = ap.parse_args()

    import torch
    mp = importlib.import_module("mental-poker_amd")
    gpu = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    m, n = args.m, args.n
    N = m * n
    eng = mp.Engine(args.curve, device=0)
    PB = eng.point_bytes
    CB = 2 * PB
    params = eng.setup(m, n, bytes([1] * 32))
    pk = eng.setup(m, 2, bytes([2] * 32))[:PB]
    base_deck = eng.setup(m, 2 * N - 3, bytes([3] * 32))
    table = eng.table(m, n, params, pk, fb_bits=args.fb_bits)
    eng.sync()
    proof_bytes = table.proof_bytes
    batches = [int(b) for b in args.batches.split(",")]
    Bmax = max(batches)
    gen = torch.Generator(device=gpu)
    gen.manual_seed(77)
    factors = torch.randint(0, 256, (Bmax, N, 32), dtype=torch.uint8, device=gpu, generator=gen)
    factors[:, :, 31] &= 0x07
    perms = torch.argsort(torch.rand(Bmax, N, device=gpu, generator=gen), dim=1).to(torch.int32).contiguous()
    seeds = torch.randint(0, 256, (Bmax, 32), dtype=torch.uint8, device=gpu, generator=gen)
    base = torch.frombuffer(bytearray(base_deck), dtype=torch.uint8).to(gpu)
    decks = base.repeat(Bmax, 1).contiguous()
    out_decks = torch.empty(Bmax, N * CB, dtype=torch.uint8, device=gpu)
    out_proofs = torch.empty(Bmax, proof_bytes, dtype=torch.uint8, device=gpu)
    st_p = torch.empty(Bmax, dtype=torch.int32, device=gpu)
    st_v = torch.empty(Bmax, dtype=torch.int32, device=gpu)
    torch.cuda.synchronize()

    def prove(B):
        table.shuffle_and_remask_batch_dev(B, decks.data_ptr(), factors.data_ptr(), perms.data_ptr(), seeds.data_ptr(),
                                           out_decks.data_ptr(), out_proofs.data_ptr(), st_p.data_ptr())

    def verify(B):
        table.verify_shuffle_batch_dev(B, decks.data_ptr(), out_decks.data_ptr(), out_proofs.data_ptr(), st_v.data_ptr())

    rows = []
    for B in batches:
      prove(B)
      eng.sync()
      for rnd in range(args.rounds):
        for cfg in args.configs.split(","):
              f = [int(v) for v in cfg.split(":")]
              pts, bits, minb = f[0], f[1], (f[2] if len(f) > 2 else args.min_batch)
              table.set_bucket_split(f[3] if len(f) > 3 else 12)
              table.set_bucket_bits(bits)
              table.set_group_verify(pts, minb)
              gsz = table.group_size(B)

              def step():
                  if not args.verify_only:
                      prove(B)
                  verify(B)
              step()
              eng.sync()
              t1 = time.perf_counter()
              for _ in range(args.steps):
                  step()
              eng.sync()
              dt = (time.perf_counter() - t1) / args.steps
              bad = int((st_p[:B] != 0).sum().item()) + int((st_v[:B] != 0).sum().item())
              # one tampered proof must be found and nobody else blamed
              keep = out_proofs[B // 3, -40].clone()
              out_proofs[B // 3, -40] ^= 1
              verify(B)
              eng.sync()
              marked = torch.nonzero(st_v[:B]).flatten().tolist()
              out_proofs[B // 3, -40] = keep
              eng.profile_enable(True)
              step()
              eng.sync()
              rep = eng.profile_report()
              eng.profile_enable(False)
              row = {"batch": B, "points_per_group": pts, "bits": bits, "split": (f[3] if len(f) > 3 else 12), "group_size": gsz, "proofs_per_s": round(B / dt, 1), "ms_per_step": round(1e3 * dt, 3),
                     "failed": bad, "tampered_found": marked == [B // 3],
                     "kernels_ms": {k: round(v[1], 3) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])[:14]}}
              rows.append(row)
              print(json.dumps(row), flush=True)
    print("%8s %10s %5s %6s %12s %10s %10s  ok" % ("batch", "points", "bits", "group", "proofs/s", "ms/step", "bucket ms"), file=sys.stderr)
    rows.sort(key=lambda r: (r["batch"], r["points_per_group"], r["bits"]))
    for r in rows:
        bk = sum(v for k, v in r["kernels_ms"].items() if k.startswith("k_bucket_msm"))
        print("%8d %10d %5d %6d %12.0f %10.3f %10.3f  %s" % (r["batch"], r["points_per_group"], r["bits"], r["group_size"], r["proofs_per_s"], r["ms_per_step"],
                                                            bk, "ok" if r["failed"] == 0 and r["tampered_found"] else "WRONG"), file=sys.stderr)


if __name__ == "__main__":
    main()
