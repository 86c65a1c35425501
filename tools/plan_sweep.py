#!/usr/bin/env python3
"""tools/plan_sweep.py -- proofs/s of the `pairs` workload (52-card deck unless --m/--n say otherwise) for a list of batch sizes and,
per batch size, a list of work-split configurations (mp_set_plan_params: fixed / variable-base terms per sub-job, bases per table
lane, points per inversion, window lanes per sub-job).  One process, one table: the fixed-base tables are built once and every
configuration is timed on the same box back to back -- the way crossovers between the engine's splits are measured.

  python tools/plan_sweep.py --batches 1024,4096 --configs "4:16:16:32:1,4:16:16:32:4,4:32:16:32:8" [--pipeline]

Output: one JSON line per (batch, configuration) and a summary table on stderr.  Needs a GPU (the engine has no CPU path).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the HIP runtime starts: the engine's two lanes and their side streams on queues of their own


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1024,4096,16384,32768")
    ap.add_argument("--configs", default="default", help="comma list of fch:vch:grp:nch:vsp (or 'default' = the engine's own choice by batch size)")
    ap.add_argument("--m", type=int, default=2)
    ap.add_argument("--n", type=int, default=26)
    ap.add_argument("--curve", default="stark")
    ap.add_argument("--fb-bits", type=int, default=21)
    ap.add_argument("--seconds", type=float, default=0.4, help="timed region per configuration (steps are derived from a probe step)")
    ap.add_argument("--pipeline", type=int, default=-1, help="mp_set_pipeline value for every run (-1: leave the default)")
    ap.add_argument("--group-lanes", type=int, default=None)
    ap.add_argument("--merged", type=int, default=None, help="1/0: mp_set_merged_verify")
    ap.add_argument("--group", default=None, help="mp_set_group_verify as points:min_batch (0:0 = off; 30464 points = 128 proofs of 52 cards)")
    ap.add_argument("--late-pipeline", action="store_true", help="switch pipelining on only after the priming pass (as bench.py's batch_curve does)")
    ap.add_argument("--profile", action="store_true", help="per-kernel milliseconds of one step per configuration")
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
    t0 = time.perf_counter()
    table = eng.table(m, n, params, pk, fb_bits=args.fb_bits)
    eng.sync()
    print("table build %.2f s" % (time.perf_counter() - t0), file=sys.stderr)
    if args.group_lanes is not None:
        table.set_group_lanes(args.group_lanes)
    if args.group is not None:
        table.set_group_verify(*[int(v) for v in args.group.split(":")])
    if args.merged is not None:
        table.set_merged_verify(bool(args.merged))
    if args.pipeline >= 0 and not args.late_pipeline:
        table.set_pipeline(args.pipeline)
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
    d0 = base.repeat(Bmax, 1).contiguous()
    decks = torch.empty(Bmax, N * CB, dtype=torch.uint8, device=gpu)
    pr0 = torch.empty(Bmax, proof_bytes, dtype=torch.uint8, device=gpu)
    st0 = torch.empty(Bmax, dtype=torch.int32, device=gpu)
    torch.cuda.synchronize()
    table.shuffle_and_remask_batch_dev(Bmax, d0.data_ptr(), factors.data_ptr(), perms.data_ptr(), seeds.data_ptr(), decks.data_ptr(),
                                       pr0.data_ptr(), st0.data_ptr())
    eng.sync()
    assert int(st0.abs().sum().item()) == 0
    del d0
    # two output sets: with pipelining on, a verify call's inputs must stay untouched until the next verify call has been issued
    NSETS = max(args.pipeline, 1) + 1
    out_decks = [torch.empty(Bmax, N * CB, dtype=torch.uint8, device=gpu) for _ in range(NSETS)]
    out_proofs = [torch.empty(Bmax, proof_bytes, dtype=torch.uint8, device=gpu) for _ in range(NSETS)]
    st_p = [torch.empty(Bmax, dtype=torch.int32, device=gpu) for _ in range(NSETS)]
    st_v = [torch.empty(Bmax, dtype=torch.int32, device=gpu) for _ in range(NSETS)]
    torch.cuda.synchronize()
    state = {"i": 0}

    def step(B):
        i = state["i"]
        table.shuffle_and_remask_batch_dev(B, decks.data_ptr(), factors.data_ptr(), perms.data_ptr(), seeds.data_ptr(),
                                           out_decks[i].data_ptr(), out_proofs[i].data_ptr(), st_p[i].data_ptr())
        table.verify_shuffle_batch_dev(B, decks.data_ptr(), out_decks[i].data_ptr(), out_proofs[i].data_ptr(), st_v[i].data_ptr())
        state["i"] = (i + 1) % NSETS

    if args.pipeline >= 0 and args.late_pipeline:
        table.set_pipeline(args.pipeline)
    rows = []
    ref = {}
    for B in batches:
        for cfg in args.configs.split(","):
            if cfg == "default":
                table.set_work_split(-1)
                label = "default"
            else:
                split, prm = 2, [int(v) for v in cfg.split(":")]
                table.set_plan_params(split, *prm)
                table.set_work_split(split)
                label = cfg
            step(B)
            eng.sync()
            t1 = time.perf_counter()
            step(B)
            eng.sync()
            probe = time.perf_counter() - t1
            steps = max(2, min(200, int(args.seconds / max(probe, 1e-4))))
            steps += (-steps) % NSETS
            t1 = time.perf_counter()
            for _ in range(steps):
                step(B)
            t_host = time.perf_counter() - t1          # the calls have returned (pipelined mode: nothing waited for but the previous verdict)
            eng.sync()
            dt = time.perf_counter() - t1
            bad = sum(int((x[:B] != 0).sum().item()) for x in st_p + st_v)
            assert bad == 0 or os.environ.get("MP_SWEEP_NOCHECK"), "%d proofs failed at B=%d cfg=%s" % (bad, B, label)
            # the bytes do not depend on the split
            sig = (bytes(out_proofs[0][B // 2].cpu().numpy().tobytes()), bytes(out_decks[0][B // 2].cpu().numpy().tobytes()))
            if B in ref and not os.environ.get("MP_SWEEP_NOCHECK"):
                assert sig == ref[B], "outputs differ between configurations at B=%d cfg=%s" % (B, label)
            ref[B] = sig
            row = {"batch": B, "config": label, "proofs_per_s": round(B * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 4), "steps": steps,
                   "host_ms_per_step": round(1e3 * t_host / steps, 4)}
            if args.profile:
                eng.profile_enable(True)
                step(B)
                step(B)
                eng.sync()
                rep = eng.profile_report()
                eng.profile_enable(False)
                row["kernels_ms_per_step"] = {k: round(v[1] / 2, 4) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])}
            rows.append(row)
            print(json.dumps(row), flush=True)
    print("%8s %-20s %12s %10s %10s" % ("batch", "config", "proofs/s", "ms/step", "host ms"), file=sys.stderr)
    for r in rows:
        print("%8d %-20s %12.0f %10.3f %10.3f" % (r["batch"], r["config"], r["proofs_per_s"], r["ms_per_step"], r["host_ms_per_step"]), file=sys.stderr)


if __name__ == "__main__":
    main()
