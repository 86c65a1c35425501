"""Where does the bucket-method kernel (kernels_bucket.hpp) overtake the Straus kernel?  Times n independent K-term
variable-base MSMs through mp_msm (host buffers: the PCIe copies are the same for both) with mp_set_bucket_min on / off and
prints the kernel time of each path from the engine's HIP-event profile.   usage: python tools/msm_crossover.py [curve]"""
import importlib
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mp = importlib.import_module("mental-poker_amd")
curve = sys.argv[1] if len(sys.argv) > 1 else "stark"
eng = mp.Engine(curve, device=0)
pb = eng.point_bytes
params = eng.setup(2, 3, bytes([1] * 32))
t = eng.table(2, 3, params, params[:pb])
rnd = random.Random(5)
print("%8s %6s %14s %14s %8s" % ("K", "MSMs", "straus ms", "bucket ms", "ratio"))
for K, n in ((1024, 2048), (2048, 1024), (4193, 512), (8192, 256), (16384, 128), (32000, 64)):
    pts = eng.setup(2, K - 3, bytes([9] * 32)) * n
    sc = bytes(rnd.getrandbits(8) for _ in range(31 * K * n))
    sc = b"".join(sc[31 * i:31 * i + 31] + b"\x03" for i in range(K * n))       # < 2^250
    res = {}
    for name, bm in (("straus", 0), ("bucket", 16)):
        t.set_bucket_min(bm)
        t.msm(n, K, sc, pts)                      # warm-up (workspace allocation)
        eng.profile_enable(True)
        out = t.msm(n, K, sc, pts)
        rep = eng.profile_report()
        eng.profile_enable(False)
        res[name] = (sum(ms for k, (c, ms) in rep.items() if k in ("k_var_msm", "k_table", "k_recode", "k_bucket_msm", "k_bucket_recode",
                                                                    "k_bucket_fold", "k_combine")), out)
    assert res["straus"][1] == res["bucket"][1]
    print("%8d %6d %14.2f %14.2f %8.2f" % (K, n, res["straus"][0], res["bucket"][0], res["bucket"][0] / res["straus"][0]))
