#!/usr/bin/env python3
"""tools/decompress_bench.py -- points/s of on-device decompression (mp_deck_deserialize_dev): arkworks-compressed decks in HBM ->
wire v1 decks in HBM.  Usage: python tools/decompress_bench.py [curve] [decks]   (52-card decks; default stark, 131072 decks = 13.6 M points)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mp = importlib.import_module("mental-poker_amd")
curve = sys.argv[1] if len(sys.argv) > 1 else "stark"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
m, n = 2, 26
N = m * n
eng = mp.Engine(curve, device=0)
ser = mp.Serializer(curve)
PB = eng.point_bytes
gpu = torch.device("cuda", 0)
# 64 distinct decks of random points (the engine's own setup sampler), serialised on the host, tiled on the device
wire = [eng.setup(m, 2 * N - 3, bytes([7, k] + [0] * 30)) for k in range(64)]
one = [ser.deck_serialize(w) for w in wire]
src = torch.frombuffer(bytearray(b"".join(one)), dtype=torch.uint8).to(gpu).view(64, -1).repeat(D // 64, 1).contiguous()
ref = torch.frombuffer(bytearray(b"".join(wire)), dtype=torch.uint8).to(gpu).view(64, -1)
out = torch.empty(D, N * 2 * PB, dtype=torch.uint8, device=gpu)
st = torch.empty(D, dtype=torch.int32, device=gpu)
torch.cuda.synchronize()
eng.deck_deserialize_dev(D, N, src.data_ptr(), out.data_ptr(), st.data_ptr())
eng.sync()
assert int(st.abs().sum().item()) == 0 and torch.equal(out[:64], ref) and torch.equal(out[-64:], ref)
K = 3
t0 = time.perf_counter()
for _ in range(K):
    eng.deck_deserialize_dev(D, N, src.data_ptr(), out.data_ptr(), st.data_ptr())
eng.sync()
dt = (time.perf_counter() - t0) / K
pts = D * 2 * N
print("%s: %d decks (%d points) decompressed in %.2f ms: %.1f M points/s, %.0f k decks/s; %.2f GB/s in, %.2f GB/s out"
      % (curve, D, pts, 1e3 * dt, pts / dt / 1e6, D / dt / 1e3, src.numel() / dt / 1e9, out.numel() / dt / 1e9))
