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


def test_shard_range_covers_everything():
    import bench
    for total in (0, 1, 7, 8, 4096, 4097):
        for world in (1, 2, 3, 8):
            spans = [bench.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
