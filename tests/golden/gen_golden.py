#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the Python big-integer oracle (oracle/py/mp_oracle.py).

The reference (Rust, un-vendored git dependencies, no toolchain in the image) cannot be run to produce
vectors and holds none of its own, so these fixtures pin the build's own frozen
"mpshuffle transcript v1": inputs and expected outputs only (data, no source text).
Usage:  python tests/golden/gen_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle", "py"))
import mp_oracle as po  # noqa: E402

K = 0x0123456789abcdef0fedcba9876543210123456789abcdef0fedcba987654321


def hx(b):
    return bytes(b).hex()


def curve_kats():
    out = {}
    for name, cv in po.CURVES.items():
      with po.curve_ctx(cv):          # coordinate width of the encoders (32 B; 48 B on BLS12-377)
        G = cv.G
        pp = po.Params(cv, 1, 1, G, [G], G, G)
        pk = po.pt_mul(cv, 7, G)
        ct = (po.pt_mul(cv, 3, G), po.pt_add(cv, po.pt_mul(cv, 5, G), po.pt_mul(cv, 3, pk)))
        rm = po.remask(pp, pk, ct, K)
        out[name] = dict(
            p=hex(cv.p), q=hex(cv.q), a=cv.a, b=hex(cv.b), G=hx(po.pt_wire(G)), k=hx(po.fe_bytes(K % cv.q)),
            twoG=hx(po.pt_wire(po.pt_mul(cv, 2, G))), kG=hx(po.pt_wire(po.pt_mul(cv, K, G))),
            qm1G=hx(po.pt_wire(po.pt_mul(cv, cv.q - 1, G))),
            remask=dict(pk=hx(po.pt_wire(pk)), ct=hx(po.deck_to_bytes([ct])), alpha=hx(po.fe_bytes(K % cv.q)),
                        out=hx(po.deck_to_bytes([rm]))))
    return out


def fs_kats():
    out = dict(blake2s_shuffle_proof=hx(po.blake2s(b"Shuffle Proof")))
    z = po.ChaCha20Rng(bytes(32))
    out["chacha20_zero_key_first64"] = hx(b"".join(po.struct.pack("<Q", z.next_u64()) for _ in range(8)))
    for name, cv in po.CURVES.items():
        fs = po.FiatShamirRng(b"Shuffle Proof")
        a = [po.fr_rand(cv, fs) for _ in range(3)]
        fs.absorb(bytes(range(200)))
        b = [po.fr_rand(cv, fs) for _ in range(3)]
        out["challenges_" + name] = dict(after_seed=[hex(v) for v in a], after_absorb_0_199=[hex(v) for v in b])
    return out


def shuffle_case(curve, m, n, seed):
    cv = po.CURVES[curve]
    pp, pk, deck, rho, perm, ps = po.gen_inputs(cv, m, n, seed)
    sh, pf = po.shuffle_and_remask(pp, pk, deck, rho, perm, ps)
    assert po.verify_shuffle(pp, pk, deck, sh, pf) == 0
    with po.curve_ctx(cv):
        return _case_dict(curve, m, n, seed, pp, pk, deck, rho, perm, ps, sh, pf)


def _case_dict(curve, m, n, seed, pp, pk, deck, rho, perm, ps, sh, pf):
    return dict(curve=curve, m=m, n=n, seed=seed, params=hx(po.params_to_bytes(pp)), pk=hx(po.pt_wire(pk)),
                deck=hx(po.deck_to_bytes(deck)), rho=hx(b"".join(po.fe_bytes(r) for r in rho)), perm=perm,
                prover_seed=hx(ps), shuffled=hx(po.deck_to_bytes(sh)), proof=hx(po.proof_to_bytes(pf)))


def chain_case(curve, m, n, L, seed):
    """one card table's shuffle chain [REF examples/round.rs:268-350]: L dependent shuffles under ONE aggregate key, the deck of link
    j + 1 is the output of link j; witness and prover seed of link j from gen_inputs(seed + 1 + j)"""
    cv = po.CURVES[curve]
    pp, pk, deck, _, _, _ = po.gen_inputs(cv, m, n, seed)
    decks, links = [deck], []
    for j in range(L):
        _, _, _, rho, perm, ps = po.gen_inputs(cv, m, n, seed + 1 + j)
        sh, pf = po.shuffle_and_remask(pp, pk, decks[-1], rho, perm, ps)
        assert po.verify_shuffle(pp, pk, decks[-1], sh, pf) == 0
        with po.curve_ctx(cv):
            links.append(dict(rho=hx(b"".join(po.fe_bytes(r) for r in rho)), perm=perm, prover_seed=hx(ps), proof=hx(po.proof_to_bytes(pf))))
        decks.append(sh)
    with po.curve_ctx(cv):
        return dict(curve=curve, m=m, n=n, links=L, seed=seed, params=hx(po.params_to_bytes(pp)), pk=hx(po.pt_wire(pk)),
                    decks=[hx(po.deck_to_bytes(d)) for d in decks], chain=links)


def main():
    with open(os.path.join(HERE, "chain_stark_m2_n3_L3_s21.json"), "w") as f:
        json.dump(chain_case("stark", 2, 3, 3, 21), f, indent=1)
    if "--chain-only" in sys.argv:
        return
    with open(os.path.join(HERE, "curve_kats.json"), "w") as f:
        json.dump(curve_kats(), f, indent=1)
    with open(os.path.join(HERE, "fs_kats.json"), "w") as f:
        json.dump(fs_kats(), f, indent=1)
    cases = [("stark", 2, 26, 7), ("stark", 4, 13, 9), ("stark", 2, 3, 1), ("stark", 3, 4, 11),
             ("bn254", 2, 4, 3), ("secp256k1", 3, 3, 5), ("bls12_377", 2, 3, 13)]
    for c in cases:
        with open(os.path.join(HERE, "shuffle_%s_m%d_n%d_s%d.json" % c), "w") as f:
            json.dump(shuffle_case(*c), f, indent=1)
        print("wrote", c)


if __name__ == "__main__":
    main()
