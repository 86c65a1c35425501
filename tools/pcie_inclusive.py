#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-buffer API (mp_shuffle_and_remask_batch + mp_verify_shuffle_batch):
inputs start in host memory, outputs end in host memory -- once with ordinary (pageable) buffers and once with
page-locked buffers from mp_host_alloc.  Reported in DESIGN.md; never bench.py's `value`.
usage: python tools/pcie_inclusive.py [B]"""
import ctypes
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mp = importlib.import_module("mental-poker_amd")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
m, n, B = 2, 26, int(sys.argv[1]) if len(sys.argv) > 1 else 65536
CHUNK = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N = m * n
eng = mp.Engine("stark", 0)
params = eng.setup(m, n, bytes([1] * 32))
pk = eng.setup(m, 2, bytes([2] * 32))[:64]
base = eng.setup(m, 2 * N - 3, bytes([3] * 32))
t = eng.table(m, n, params, pk, fb_bits=21)
lib = t.lib
if CHUNK:
    t.set_io_chunk(CHUNK)
rng = np.random.default_rng(1)
src = dict(rho=rng.integers(0, 256, size=(B, N, 32), dtype=np.uint8), perms=np.argsort(rng.random((B, N)), axis=1).astype(np.uint32),
           seeds=rng.integers(0, 256, size=(B, 32), dtype=np.uint8), decks=np.frombuffer(base, dtype=np.uint8)[None, :].repeat(B, 0))
src["rho"][:, :, 31] &= 7
outs = dict(out_d=((B, N * 128), np.uint8), out_p=((B, t.proof_bytes), np.uint8), st=((B,), np.int32), st2=((B,), np.int32))


def pinned(shape, dtype):
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = lib.mp_host_alloc(nbytes)
    assert p, "mp_host_alloc failed"
    return np.frombuffer((ctypes.c_uint8 * nbytes).from_address(p), dtype=dtype).reshape(shape), p


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


for kind in (("page-locked",) if CHUNK else ("pageable", "page-locked")):
    held = []
    if kind == "pageable":
        a = dict(src)
        o = {k: np.empty(*v) for k, v in outs.items()}
    else:
        a, o = {}, {}
        for k, v in src.items():
            a[k], p = pinned(v.shape, v.dtype)
            a[k][...] = v
            held.append(p)
        for k, v in outs.items():
            o[k], p = pinned(*v)
            held.append(p)

    def run():
        rc = lib.mp_shuffle_and_remask_batch(t.h, B, ptr(a["decks"]), ptr(a["rho"]), ptr(a["perms"]), ptr(a["seeds"]), ptr(o["out_d"]),
                                             ptr(o["out_p"]), ptr(o["st"]))
        assert rc == 0
        rc = lib.mp_verify_shuffle_batch(t.h, B, ptr(a["decks"]), ptr(o["out_d"]), ptr(o["out_p"]), ptr(o["st2"]))
        assert rc == 0

    run()
    t0 = time.perf_counter()
    K = 3
    for _ in range(K):
        run()
    dt = time.perf_counter() - t0
    assert not o["st"].any() and not o["st2"].any()
    eng.profile_enable(True)
    run()
    rep = eng.profile_report()
    eng.profile_enable(False)
    print("   kernel time of one batch (HIP events): %.1f ms; largest: %s" % (sum(v[1] for v in rep.values()),
          ", ".join("%s %.1f" % (k, v[1]) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])[:5])))
    nbytes = N * 128 * 4 + N * 36 + 32 + 2 * t.proof_bytes
    print("host-buffer API, %-11s buffers, B=%d chunk=%d: %.0f proofs/s (%.1f ms per batch), %.2f GB/s over PCIe"
          % (kind, B, CHUNK or 65536, B * K / dt, 1e3 * dt / K, B * K * nbytes / dt / 1e9))
    del a, o
    for p in held:
        lib.mp_host_free(p)
