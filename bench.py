#!/usr/bin/env python3
"""bench.py -- shuffle proofs/sec (prove + verify) on MI355X, BASELINE.json's metric.

One step = one pass of the hot path over one batch: B independent 52-card decks (m=2, n=26, STARK curve
[REF barnett-smart-card-protocol/examples/round.rs:229-230]) each go through `shuffle_and_remask` (ElGamal
re-encryption + Bayer-Groth prover) and `verify_shuffle`, with every input already resident in HBM.
Synthetic data: random ciphertext decks [REF src/discrete_log_cards/tests.rs:187] obtained by re-encrypting one
random base deck on the GPU; masking factors uniform below 2^251; uniform permutations; random prover seeds.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Proofs are independent, so ranks shard the batch with no data-path collective (weak scaling: B proofs per GPU
per step); the shared parameters are produced on rank 0 and broadcast once over RCCL.  Rank 0 prints one JSON
line.  `roofline` is measured live with HIP events on the engine's own stream around every kernel launch of the
timed region; `cpu_baseline` times the oracle's single-threaded C++ restatement (arkworks-style algorithms) on a
bounded sample of the same workload on this box's host cores.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "py")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
INT_MAD_PEAK_G = 18000.0       # v_mad_u64_u32 issue rate measured with tools/microbench/intrate.hip (Gmad/s)
# exact 32x32+64 multiply-add counts (v_mad_u64_u32 + v_mad_i64_i32, the same quarter-rate pipe) of the compiled STARK
# group law (llvm -S of xyzz_madd_ip / xyzz_dbl_ip / fe_mul / fe_sqr, 9x29-bit limbs, subtractive Montgomery reduction):
# product 81 + 18 = 99, square 45 + 18 = 63, a b - c d with one reduction 162 + 18 = 180; XYZZ mixed addition 8M+2S = 900,
# XYZZ doubling 6M+4S = 828 (the MSM loops; each contains one fused product pair);
# batched-affine table entry 5M+1S + 4/15 of 1/64 of an inversion (256S+45M) = 644, Jacobian+Jacobian addition 11M+5S =
# 1404 (combines), normalisation of one point 6M+1S + 1/64 inversion = 979
MADS = {"madd": 900, "dbl": 828, "aff": 644, "jac": 1404, "norm": 979}


# ---- distributed helpers (backend-agnostic: RCCL on GPUs, gloo in the CPU tests) -----------------------------
def dist_info():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def bcast_bytes(data, nbytes, src, device):
    """broadcast a byte string from rank `src` to every rank (shared parameters, once per session)"""
    import torch
    import torch.distributed as dist
    if data is not None:
        t = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    else:
        t = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return bytes(t.cpu().numpy().tobytes())


def reduce_max(x, device):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(x, device):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def shard_range(total, rank, world):
    """static contiguous block partition of proof indices"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ---- the benchmark -----------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=262144, help="proofs per GPU per step (~0.45 MB of HBM each: 142 GB in use at the default)")
    ap.add_argument("--m", type=int, default=2)
    ap.add_argument("--n", type=int, default=26)
    ap.add_argument("--curve", default="stark")
    ap.add_argument("--streams", type=int, default=1, help="independent engine contexts (HIP streams) per GPU; the batch is split evenly")
    ap.add_argument("--fb-bits", type=int, default=20, help="fixed-base window width (8, 16 or 20 bits; 20 = 27 GB of tables at n=26)")
    ap.add_argument("--cpu-iters", type=int, default=160, help="prove+verify pairs timed for cpu_baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-batch", type=int, default=None,
                    help="batches up to this size use the latency plan (engine default 8192): mp_set_latency_batch")
    ap.add_argument("--keyed", type=int, default=0, metavar="K",
                    help="keyed batches: K distinct aggregate keys (card tables) spread over the batch, one key per proof "
                         "(mp_*_batch_keys_dev); 0 = every proof under the table's own key")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank, world, local = dist_info()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # test hooks (used to dry-run the N > 1 path on a single-GPU box): MP_BENCH_FORCE_DEVICE puts every rank on one GPU,
    # MP_BENCH_BACKEND=gloo runs the (tiny, once-per-session) collectives on CPU tensors instead of RCCL
    local = int(os.environ.get("MP_BENCH_FORCE_DEVICE", local))
    backend = os.environ.get("MP_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    gpu = torch.device("cuda", local)
    dev = gpu if backend == "nccl" else torch.device("cpu")      # device of the collective payloads
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=gpu)
        else:
            dist.init_process_group(backend)
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    mp = importlib.import_module("mental-poker_amd")
    m, n, curve, B = args.m, args.n, args.curve, args.batch
    N = m * n
    eng = mp.Engine(curve, device=local)

    # ---- shared parameters: rank 0 runs `setup`, everyone receives them over RCCL (once)
    psz = 64 * (n + 3)
    blob = None
    if rank == 0:
        params = eng.setup(m, n, bytes([1] * 32))
        pk = eng.setup(m, 2, bytes([2] * 32))[:64]                  # aggregate key: a random group element
        base_deck = eng.setup(m, 2 * N - 3, bytes([3] * 32))        # 2N random points = N random ciphertexts
        blob = params + pk + base_deck
    blob = bcast_bytes(blob, psz + 64 + 128 * N, 0, dev)
    params, pk, base_deck = blob[:psz], blob[psz:psz + 64], blob[psz + 64:]
    # ---- S independent contexts (one HIP stream each); each owns 1/S of the batch
    S = max(1, args.streams)
    Bs = B // S
    assert Bs * S == B, "--batch must be a multiple of --streams"
    engines = [eng] + [mp.Engine(curve, device=local) for _ in range(S - 1)]
    tables = [e.table(m, n, params, pk, fb_bits=args.fb_bits) for e in engines]
    for t in tables:
        if args.latency_batch is not None:
            t.set_latency_batch(args.latency_batch)
        t.reserve(Bs)
    table = tables[0]
    proof_bytes = table.proof_bytes

    # ---- synthetic inputs, resident in HBM
    gen = torch.Generator(device=gpu)
    gen.manual_seed(1234 + rank)

    def rand_bytes(*shape):
        return torch.randint(0, 256, shape, dtype=torch.uint8, device=gpu, generator=gen)

    def rand_factors():
        f = rand_bytes(B, N, 32)
        f[:, :, 31] &= 0x07            # < 2^251 < group order
        return f.contiguous()

    def rand_perms():
        return torch.argsort(torch.rand(B, N, device=gpu, generator=gen), dim=1).to(torch.int32).contiguous()

    base = torch.frombuffer(bytearray(base_deck), dtype=torch.uint8).to(gpu)
    decks0 = base.repeat(B, 1).contiguous()
    decks = torch.empty(B, N * 128, dtype=torch.uint8, device=gpu)
    out_decks = torch.empty(B, N * 128, dtype=torch.uint8, device=gpu)
    out_proofs = torch.empty(B, proof_bytes, dtype=torch.uint8, device=gpu)
    st_p = torch.empty(B, dtype=torch.int32, device=gpu)
    st_v = torch.empty(B, dtype=torch.int32, device=gpu)
    torch.cuda.synchronize()

    def sl(t, i):
        return t[i * Bs:(i + 1) * Bs].data_ptr()

    def sync_all():
        for e in engines:
            e.sync()

    # keyed batches: K aggregate keys (random group elements from the engine's own setup), key b % K for proof b
    keys = None
    if args.keyed > 0:
        kpts = eng.setup(m, max(args.keyed, 2), bytes([4] * 32))
        kt = torch.frombuffer(bytearray(kpts[:64 * args.keyed]), dtype=torch.uint8).to(gpu).view(args.keyed, 64)
        keys = kt[torch.arange(B, device=gpu) % args.keyed].contiguous()
    # prime: B different random decks = re-encryptions of the base deck (untimed input generation)
    f0, p0, s0 = rand_factors(), rand_perms(), rand_bytes(B, 32)
    torch.cuda.synchronize()
    for e in engines:
        e.profile_enable(True)
    for i, t in enumerate(tables):
        t.shuffle_and_remask_batch_dev(Bs, sl(decks0, i), sl(f0, i), sl(p0, i), sl(s0, i), sl(decks, i), sl(out_proofs, i), sl(st_p, i))
    sync_all()
    priming_launches = {}       # per-kernel launches of the untimed priming prove (tools/pmc_summary.py skips them)
    for e in engines:
        for k, (cnt, _) in e.profile_report().items():
            priming_launches[k] = priming_launches.get(k, 0) + cnt
        e.profile_enable(False)
    assert int(st_p.abs().sum().item()) == 0, "priming pass failed"
    factors, perms, seeds = rand_factors(), rand_perms(), rand_bytes(B, 32)
    torch.cuda.synchronize()

    def step():
        for i, t in enumerate(tables):
            if keys is not None:
                t.shuffle_and_remask_batch_keys_dev(Bs, sl(keys, i), sl(decks, i), sl(factors, i), sl(perms, i), sl(seeds, i),
                                                    sl(out_decks, i), sl(out_proofs, i), sl(st_p, i))
                t.verify_shuffle_batch_keys_dev(Bs, sl(keys, i), sl(decks, i), sl(out_decks, i), sl(out_proofs, i), sl(st_v, i))
                continue
            t.shuffle_and_remask_batch_dev(Bs, sl(decks, i), sl(factors, i), sl(perms, i), sl(seeds, i), sl(out_decks, i),
                                           sl(out_proofs, i), sl(st_p, i))
            t.verify_shuffle_batch_dev(Bs, sl(decks, i), sl(out_decks, i), sl(out_proofs, i), sl(st_v, i))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        sync_all()

    for _ in range(args.warmup):
        step()
    barrier()
    for e in engines:
        e.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = {}
    for e in engines:
        for k, (cnt, ms) in e.profile_report().items():
            c0, m0 = prof.get(k, (0, 0.0))
            prof[k] = (c0 + cnt, m0 + ms)
        e.profile_enable(False)
    elapsed = reduce_max(elapsed, dev)

    # ---- correctness of what was timed (outside the timed region)
    bad = int((st_p != 0).sum().item()) + int((st_v != 0).sum().item())
    assert bad == 0, "%d proofs failed on rank %d" % (bad, rank)
    parity = None
    if rank == 0:
        import coracle as co
        co.build()
        b = B // 2
        pk_b = pk if keys is None else bytes(keys[b].cpu().numpy().tobytes())
        exp_deck, exp_proof = co.shuffle_and_remask(curve, m, n, params, pk_b, bytes(decks[b].cpu().numpy().tobytes()),
                                                    bytes(factors[b].cpu().numpy().tobytes()),
                                                    [int(v) for v in perms[b].cpu().tolist()],
                                                    bytes(seeds[b].cpu().numpy().tobytes()))
        parity = (bytes(out_decks[b].cpu().numpy().tobytes()) == exp_deck and
                  bytes(out_proofs[b].cpu().numpy().tobytes()) == exp_proof)
        assert parity, "timed output differs from the oracle"

    total_proofs = reduce_sum(B * args.steps, dev)
    value = total_proofs / elapsed

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (live HIP-event timings of the timed region)
    stats = table.plan_stats()
    census = table.work_census()
    dom = max(prof.items(), key=lambda kv: kv[1][1])
    dom_name, (dom_count, dom_ms) = dom
    kernel_ms_total = sum(v[1] for v in prof.values())

    def alg_bytes_per_proof(kernel):
        """ALGORITHMIC bytes one proof needs from this kernel class in one step (prove + verify launches):
        scalars 32 B, points 64 B, Jacobian results 96 B ("algorithmic bytes")."""
        pv = [stats["prove"], stats["verify"]]
        if kernel == "k_var_msm":
            return sum(s["var_terms"] * (32 + 64) + s["var_jobs"] * 96 for s in pv)
        if kernel == "k_fixed_msm":
            return sum(s["fixed_terms"] * 32 + s["fixed_jobs"] * 96 for s in pv)
        if kernel == "k_table":
            return sum(s["table_bases"] * (64 + 16 * 96) for s in pv)
        if kernel == "k_normalize":
            return sum(s["table_bases"] * 16 * (96 + 64) for s in pv)
        return 45 * 1024
    # per-launch figures: one launch covers Bs = B / streams proofs; summed over the launches of the timed region
    dom_bytes = alg_bytes_per_proof(dom_name) * B * args.steps
    achieved_gbs = dom_bytes / (dom_ms * 1e-3) / 1e9
    # exact multiply-add count of one prove+verify from the static plan (stats) and the instruction counts above
    vw, fw = stats["var_windows"], stats["fixed_windows"]
    mads_per_proof = 0
    for side in ("prove", "verify"):
        st_ = stats[side]
        fixed_madds = st_["fixed_terms"] * fw
        if side == "prove":
            fixed_madds -= N * (fw - 1)               # c_A commits pi(i)+1 <= N: one non-zero window per term
            fixed_madds += 2 * N * (fw + 1)           # re-encryption: 2N fixed-base scalar-muls + one addition each
            norm_points = 2 * N + 11 * m + 7          # shuffled deck + proof points
        else:
            norm_points = 0
        mads_per_proof += (fixed_madds + st_["var_terms"] * vw) * MADS["madd"]
        mads_per_proof += st_["var_jobs"] * (vw - 1) * 5 * MADS["dbl"]
        mads_per_proof += st_["table_bases"] * 15 * MADS["aff"]
        mads_per_proof += st_["combine_terms"] * MADS["jac"] + norm_points * MADS["norm"]
    mads = mads_per_proof * B * args.steps
    whole_path_bytes = 45 * 1024 if (m, n) == (2, 26) else None
    # HBM traffic of the dominant kernel from the PMC pass committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
    # --pmc WRITE_SIZE in separate runs, gfx950 x2 correction applied to FETCH_SIZE), scaled to this batch
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01i_pmc_summary.json")))["kernels"].get(dom_name)
        if pmc and (m, n) == (2, 26):
            traffic = pmc["hbm_bytes_per_proof_per_step_corrected"] * B * args.steps / dom_count
    except (OSError, ValueError, KeyError):
        traffic = None
    roofline = {
        "bound": "hbm", "kernel": dom_name, "achieved": round(achieved_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
        "avg_launch_ms": dom_ms / dom_count, "launches": dom_count,
        "alg_bytes_per_launch": dom_bytes / dom_count,
        "note": "path is integer-ALU bound (SURVEY 8d3): see int_mul",
        "int_mul": {"bound": "v_mad_u64_u32 issue", "achieved": round(mads / (kernel_ms_total * 1e-3) / 1e9, 1),
                    "peak": INT_MAD_PEAK_G, "unit": "Gmad/s",
                    "frac": mads / (kernel_ms_total * 1e-3) / 1e9 / INT_MAD_PEAK_G,
                    "mads_per_proof": mads_per_proof,
                    "note": "exact 32x32+64 mad count (v_mad_u64_u32 + v_mad_i64_i32) of the compiled group law x static plan; peak = 256 CU x 4 SIMD x 8 lanes/clk x 2.4 GHz = 19.7 T/s theoretical, 18 T/s measured (tools/microbench/intrate.hip)"},
        "whole_path_hbm_frac": (value * whole_path_bytes / 1e9 / HBM_PEAK_GBS) if whole_path_bytes else None,
        "kernels_ms": {k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
        "kernel_launches": {k: v[0] for k, v in prof.items()}, "priming_launches": priming_launches,
    }

    # ---- CPU baseline: the oracle's C++ restatement (port), single thread, bounded sample
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        import coracle as co
        it = args.cpu_iters
        t_p, t_v = co.bench(curve, m, n, 99, it)
        cpu = {"value": it / (t_p + t_v), "unit": "proofs/s", "cores": 1, "kind": "port",
               "sample": "%d prove+verify pairs, %d-card deck (m=%d,n=%d), %s, single thread; prove %.1f ms verify %.1f ms each"
                         % (it, N, m, n, curve, 1e3 * t_p / it, 1e3 * t_v / it),
               "host_cores_available": os.cpu_count()}
        # the same port on every host core, one independent proof stream per thread (the reference itself is
        # single-threaded: BASELINE.md section 2); ctypes releases the GIL for the duration of the C call
        from concurrent.futures import ThreadPoolExecutor
        T = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        quota = None
        try:                                   # cgroup v2 CPU quota of the container, if any ("max" = none)
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = None if q == "max" else float(q) / float(per)
        except (OSError, ValueError):
            pass
        T = max(1, min(T, 64, int(quota + 0.5) if quota else T))    # bounded: the leg must stay within seconds
        it_mt = 8
        t0 = time.perf_counter()
        with ThreadPoolExecutor(T) as ex:
            list(ex.map(lambda i: co.bench(curve, m, n, 1000 + i, it_mt), range(T)))
        wall = time.perf_counter() - t0
        cpu["all_cores"] = {"value": T * it_mt / wall, "unit": "proofs/s", "cores": T,
                            "cgroup_cpu_quota": quota,
                            "sample": "%d threads x %d prove+verify pairs, one proof stream per thread, %.1f s wall" % (T, it_mt, wall)}

    out = {
        "metric": "shuffle proofs/sec (prove+verify)", "value": value, "unit": "proofs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (256-bit Montgomery: 9x29-bit lazy base field, 8x32 scalar field)",
        "data": "synthetic",
        "config": {"workload": "%d-card deck, m=%d n=%d, %s curve, shuffle_and_remask + verify_shuffle" % (N, m, n, curve),
                   "proofs_per_gpu_per_step": B, "streams": S, "fixed_base_window_bits": args.fb_bits,
                   "aggregate_keys": args.keyed if args.keyed else 1,
                   "parity_vs_oracle": parity},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
