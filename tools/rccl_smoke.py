#!/usr/bin/env python3
"""tools/rccl_smoke.py -- the few collectives the sharded shuffle-proof bench needs, exercised once before anything expensive is built.

Independent shuffle proofs shard across the GPUs of a node with no data-path collective: the only exchanges are a
broadcast of the shared parameters (G, commit key, H, extra generator, aggregate key: a few KB, once per session) and an all-gather
of per-rank counts / verdict sums / seconds.  This script runs exactly those over RCCL with one rank per GPU and fails FAST with the
failing call named -- instead of a hang or a stack trace after 48 GB of fixed-base tables have been built:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_smoke.py

bench.py calls run_checks() itself when WORLD_SIZE > 1, before any table is built.  Backend: nccl (= RCCL on ROCm); MP_BENCH_BACKEND=gloo
runs the same calls on CPU tensors (the CPU tests use it).  Exit code 0 and one JSON line {"rccl_world": N, ...} on rank 0 if all is well.
"""
import json
import os
import sys
import time


class CollectiveCheckFailed(RuntimeError):
    pass


def _step(name, fn, log):
    t0 = time.perf_counter()
    try:
        out = fn()
    except Exception as e:                                    # name the failing call: the point of this script
        raise CollectiveCheckFailed("%s failed: %s: %s" % (name, type(e).__name__, e)) from e
    log[name + "_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
    return out


def run_checks(dist, device, payload_bytes=4096):
    """broadcast a `payload_bytes` byte tensor from rank 0, all-gather one float64 row per rank, all-reduce MAX, barrier.
    `dist` = an initialised torch.distributed; `device` = the device the collective payloads live on.  Returns a dict of timings;
    raises CollectiveCheckFailed naming the first call that failed or returned wrong data."""
    import torch
    log = {}
    rank, world = dist.get_rank(), dist.get_world_size()
    ref = bytes((i * 131 + 7) & 0xFF for i in range(payload_bytes))

    def bcast():
        t = (torch.frombuffer(bytearray(ref), dtype=torch.uint8).to(device) if rank == 0
             else torch.zeros(payload_bytes, dtype=torch.uint8, device=device))
        dist.broadcast(t, src=0)
        got = bytes(t.cpu().numpy().tobytes())
        if got != ref:
            raise ValueError("rank %d received %d differing bytes of the parameter blob" % (rank, sum(a != b for a, b in zip(got, ref))))
    _step("broadcast_parameters", bcast, log)

    def gather():
        t = torch.tensor([float(rank), float(rank * rank), 1.0], dtype=torch.float64, device=device)
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        rows = [[float(v) for v in o.cpu().tolist()] for o in out]
        if rows != [[float(r), float(r * r), 1.0] for r in range(world)]:
            raise ValueError("all_gather returned %r" % (rows,))
    _step("all_gather_rows", gather, log)

    def reduce_max():
        t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if float(t.item()) != float(world):
            raise ValueError("all_reduce(MAX) = %r, expected %d" % (float(t.item()), world))
    _step("all_reduce_max", reduce_max, log)
    _step("barrier", dist.barrier, log)
    log["rccl_world"] = world
    return log


def main():
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    backend = os.environ.get("MP_BENCH_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log = {"backend": backend, "world": world}
    try:
        if backend == "nccl":
            if not torch.cuda.is_available():
                raise CollectiveCheckFailed("torch.cuda.is_available() is False on rank %d" % rank)
            local = int(os.environ.get("MP_BENCH_FORCE_DEVICE", local))
            if local >= torch.cuda.device_count():
                raise CollectiveCheckFailed("rank %d: no GPU %d on this node (%d visible)" % (rank, local, torch.cuda.device_count()))
            torch.cuda.set_device(local)
            device = torch.device("cuda", local)
            _step("init_process_group(nccl, device_id)", lambda: dist.init_process_group("nccl", device_id=device), log)
        else:
            device = torch.device("cpu")
            _step("init_process_group(%s)" % backend, lambda: dist.init_process_group(backend), log)
        log.update(run_checks(dist, device))
    except CollectiveCheckFailed as e:
        print("rccl_smoke: rank %d: %s" % (rank, e), file=sys.stderr, flush=True)
        os._exit(3)                                           # no clean-up collectives on a broken group
    if rank == 0:
        print(json.dumps(log), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
