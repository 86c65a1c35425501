import importlib, os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
mp = importlib.import_module("mental-poker_amd")
import coracle as co
m, n = 2, 26
g = co.gen_inputs("stark", m, n, 7)
eng = mp.Engine("stark", 0)
t = eng.table(m, n, g["params"], g["pk"], fb_bits=16)
B=1
decks, rho, perm, seeds = g["deck"] * B, g["rho"] * B, g["perm"] * B, g["prover_seed"] * B
d, p, st = t.shuffle_and_remask_batch(decks, rho, perm, seeds)
eng.profile_enable(True)
d, p, st = t.shuffle_and_remask_batch(decks, rho, perm, seeds)
rp = eng.profile_report()
sv = t.verify_shuffle_batch(decks, d, p)
rv = eng.profile_report()
eng.profile_enable(False)
for name, r in (("prove", rp), ("verify", rv)):
    tot = sum(v[1] for v in r.values())
    print(name, "kernel ms total %.2f, launches %d" % (tot, sum(v[0] for v in r.values())))
    for k, v in sorted(r.items(), key=lambda kv: -kv[1][1])[:40]:
        print("   %-16s x%-3d %.3f ms" % (k, v[0], v[1]))
