#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-buffer API (mp_shuffle_and_remask_batch + mp_verify_shuffle_batch):
inputs start in host memory, outputs end in host memory.  Reported in DESIGN.md; never bench.py's `value`."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mp = importlib.import_module("mental-poker_amd")
import numpy as np
m, n, B = 2, 26, int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = m * n
eng = mp.Engine("stark", 0)
params = eng.setup(m, n, bytes([1] * 32)); pk = eng.setup(m, 2, bytes([2] * 32))[:64]
base = eng.setup(m, 2 * N - 3, bytes([3] * 32))
t = eng.table(m, n, params, pk, fb_bits=16)
rng = np.random.default_rng(1)
rho = rng.integers(0, 256, size=(B, N, 32), dtype=np.uint8); rho[:, :, 31] &= 7
perms = np.argsort(rng.random((B, N)), axis=1).astype(np.uint32)
seeds = rng.integers(0, 256, size=(B, 32), dtype=np.uint8)
decks = np.frombuffer(base, dtype=np.uint8)[None, :].repeat(B, 0)
import ctypes
lib = t.lib
def ptr(a): return a.ctypes.data_as(ctypes.c_void_p)
out_d = np.empty((B, N * 128), np.uint8); out_p = np.empty((B, t.proof_bytes), np.uint8); st = np.empty(B, np.int32); st2 = np.empty(B, np.int32)
def run():
    rc = lib.mp_shuffle_and_remask_batch(t.h, B, ptr(decks), ptr(rho), ptr(perms), ptr(seeds), ptr(out_d), ptr(out_p), ptr(st)); assert rc == 0
    rc = lib.mp_verify_shuffle_batch(t.h, B, ptr(decks), ptr(out_d), ptr(out_p), ptr(st2)); assert rc == 0
run()
t0 = time.perf_counter(); K = 3
for _ in range(K): run()
dt = time.perf_counter() - t0
assert not st.any() and not st2.any()
print("host-buffer API, B=%d: %.0f proofs/s (%.1f ms per batch), %.2f GB/s over PCIe" % (B, B * K / dt, 1e3 * dt / K, B * K * (N*128*4 + N*36 + 32 + 2*t.proof_bytes) / dt / 1e9))
