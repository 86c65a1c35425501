"""CPU test of the N > 1 path of bench.py: world_size 2 over gloo.  The data path has no collective (proofs are
independent): what is exercised is the one-off broadcast of the shared parameters, the block partition
of proof indices and the max / sum reductions behind the whole-job throughput."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    blob = bytes(range(256)) * 7 if rank == 0 else None
    got = bench.bcast_bytes(blob, 256 * 7, 0, dev)
    lo, hi = bench.shard_range(1001, rank, world)
    t = bench.reduce_max(1.0 + rank, dev)
    s = bench.reduce_sum(hi - lo, dev)
    assert bench.dist_info() == (rank, world, rank)
    rows = bench.gather_rows([hi - lo, rank, 0.5], dev)
    assert rows == [[501.0, 0.0, 0.5], [500.0, 1.0, 0.5]]
    assert bench.gather_objects({"r": rank}) == [{"r": 0}, {"r": 1}]
    smoke = bench.rccl_smoke(dist, dev)                  # tools/rccl_smoke.py: what bench.py runs first when WORLD_SIZE > 1
    assert smoke["rccl_world"] == world and all(k in smoke for k in ("broadcast_parameters_ms", "all_gather_rows_ms", "all_reduce_max_ms", "barrier_ms"))
    q.put((rank, got == bytes(range(256)) * 7, (lo, hi), t, s))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_two_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]
    assert res[0][2] == (0, 501) and res[1][2] == (501, 1001)
    assert res[0][3] == res[1][3] == 2.0
    assert res[0][4] == res[1][4] == 1001.0


def test_strong_scaling_batches():
    """--scaling strong: the batch is the job's; ranks get equal contiguous parts, seeded per part (the unsharded run seeds per the same
    blocks, so the bytes agree: tests/test_gpu_bench.py); weak keeps the batch per rank"""
    import bench
    import pytest
    assert bench.scaled_batch(262144, 8, "weak", None) == (262144, None)
    assert bench.scaled_batch(262144, 8, "strong", None) == (32768, 32768)
    assert bench.scaled_batch(384, 2, "strong", 96) == (192, 96)
    assert bench.scaled_batch(384, 1, "strong", 192) == (384, 192)
    with pytest.raises(SystemExit):
        bench.scaled_batch(100, 8, "strong", None)
    with pytest.raises(SystemExit):
        bench.scaled_batch(384, 2, "strong", 100)
    # rank r of a strong run seeds block (r * per) // seed_block + k: the blocks of the global batch in order
    per, sb = bench.scaled_batch(384, 2, "strong", None)
    assert [(r * per) // sb + k for r in range(2) for k in range(per // sb)] == [0, 1]


def test_rccl_smoke_names_the_failing_call():
    """a collective that misbehaves is reported by name (tools/rccl_smoke.py), not as a hang or a bare stack trace"""
    import importlib.util
    import pytest
    spec = importlib.util.spec_from_file_location("rccl_smoke", os.path.join(ROOT, "tools", "rccl_smoke.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    class Broken:
        class ReduceOp:
            MAX = 0

        def get_rank(self):
            return 0

        def get_world_size(self):
            return 2

        def broadcast(self, t, src):
            raise RuntimeError("NCCL error: unhandled system error")
    with pytest.raises(mod.CollectiveCheckFailed) as ei:
        mod.run_checks(Broken(), torch.device("cpu"))
    assert "broadcast_parameters failed" in str(ei.value) and "unhandled system error" in str(ei.value)


def test_shard_range_covers_everything():
    import bench
    for total in (0, 1, 7, 8, 4096, 4097):
        for world in (1, 2, 3, 8):
            spans = [bench.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


# ---- the sharded data path itself: every rank proves and verifies its block of the batch with an ENGINE (the development
# emulator -- kernel bodies as CPU loops, tools/hostemu -- since this container has no GPU), parameters travel through the
# same broadcast bench.py uses, and the concatenated per-rank outputs must equal the unsharded run byte for byte
def _inputs(co, curve, m, n, total):
    g = co.gen_inputs(curve, m, n, 77)
    N = m * n
    decks, rho, perms, seeds = b"", b"", [], b""
    for i in range(total):
        gi = co.gen_inputs(curve, m, n, 1000 + i)
        decks += g["deck"]
        rho += gi["rho"]
        perms += gi["perm"]
        seeds += gi["prover_seed"]
    return g, decks, rho, perms, seeds


def _engine_worker(rank, world, port, q, total):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import ctypes
    import importlib
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bench
    import coracle as co
    pkg = importlib.import_module("mental-poker_amd")
    lib = pkg._native.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", "libmpemu.so")))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    curve, m, n = "stark", 2, 3
    N = m * n
    g, decks, rho, perms, seeds = _inputs(co, curve, m, n, total)
    eng = pkg._native.Engine(curve, 0, lib=lib)
    blob = (g["params"] + g["pk"]) if rank == 0 else None            # rank 0 owns the session's parameters
    blob = bench.bcast_bytes(blob, 64 * (n + 3) + 64, 0, dev)
    table = eng.table(m, n, blob[:64 * (n + 3)], blob[64 * (n + 3):])
    lo, hi = bench.shard_range(total, rank, world)
    sl = lambda b, w: b[lo * w:hi * w]
    out_d, out_p, st = table.shuffle_and_remask_batch(sl(decks, 128 * N), sl(rho, 32 * N), perms[lo * N:hi * N], sl(seeds, 32))
    sv = table.verify_shuffle_batch(sl(decks, 128 * N), out_d, out_p)
    rows = bench.gather_rows([hi - lo, sum(1 for v in st + sv if v != 0)], dev)
    parts = bench.gather_objects((out_d, out_p))
    if rank == 0:
        full_d, full_p, st_all = table.shuffle_and_remask_batch(decks, rho, perms, seeds)      # the unsharded run
        ok = b"".join(p[0] for p in parts) == full_d and b"".join(p[1] for p in parts) == full_p and not any(st_all)
        exp_d, exp_p = co.shuffle_and_remask(curve, m, n, g["params"], g["pk"], decks[:128 * N], rho[:32 * N], perms[:N], seeds[:32])
        ok = ok and full_d[:128 * N] == exp_d and full_p[:len(exp_p)] == exp_p
        q.put((ok, rows))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_engine_output_equals_unsharded():
    import subprocess
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "tools", "hostemu")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    total = 5
    procs = [ctx.Process(target=_engine_worker, args=(r, 2, port, q, total)) for r in range(2)]
    for p in procs:
        p.start()
    ok, rows = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
    assert rows == [[3.0, 0.0], [2.0, 0.0]]
