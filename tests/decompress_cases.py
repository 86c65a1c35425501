"""Shared body of the on-device decompression tests (mp_points_deserialize_dev / mp_deck_deserialize_dev): run against the development
emulator on CPU (tests/test_cabi_and_host.py) and against the HIP engine on the GPU box (tests/test_gpu_round4.py).  The expected
bytes come from the ORACLE's encoder / decoder (oracle/py/ark_canonical.py, big integers), not from the package."""
import random

import ark_canonical as ac
import mp_oracle as po


def run_decompress_cases(eng, mem, curve, golden, n_random=24):
    """eng: _native.Engine; mem: object with put(bytes) -> (handle, address), new(nbytes) -> (handle, address), get(handle, nbytes) -> bytes"""
    cv = po.CURVES[curve]
    L = ac.compressed_len(cv)
    rng = random.Random(4242)
    with po.curve_ctx(cv):
        PB = po.point_bytes()
        # ---- (1) the decks of the golden vector: serialised Vec<MaskedCard> -> wire decks, two decks in one call
        if golden is not None:
            m, n = golden["m"], golden["n"]
            wire = [bytes.fromhex(golden["deck"]), bytes.fromhex(golden["shuffled"])]
            data = b"".join(ac.enc_deck(cv, po.deck_from_bytes(w)) for w in wire)
            hin, pin = mem.put(data)
            hout, pout = mem.new(2 * len(wire[0]))
            hst, pst = mem.new(8)
            eng.deck_deserialize_dev(2, m * n, pin, pout, pst)
            eng.sync()
            assert mem.get(hst, 8) == bytes(8)
            assert mem.get(hout, 2 * len(wire[0])) == wire[0] + wire[1]
            # a wrong length prefix fails its deck only
            bad = bytearray(data)
            bad[0] ^= 1
            hin, pin = mem.put(bytes(bad))
            eng.deck_deserialize_dev(2, m * n, pin, pout, pst)
            eng.sync()
            st = mem.get(hst, 8)
            assert int.from_bytes(st[:4], "little", signed=True) == -1 and st[4:] == bytes(4)
        # ---- (2) single points: random multiples of G with both signs, infinity, and everything that must be refused
        pts, expect = [], []
        for _ in range(n_random):
            P = po.pt_mul(cv, rng.randrange(1, cv.q), cv.G)
            pts.append(ac.enc_point(cv, P))
            expect.append((0, po.pt_wire(P)))
        pts.append(ac.enc_point(cv, None))
        expect.append((0, bytes(PB)))
        x = 2
        while po.fq_sqrt(cv, (x * x * x + cv.a * x + cv.b) % cv.p) is not None:
            x += 1
        bad_list = [x.to_bytes(L, "little"),                                        # x not on the curve
                    cv.p.to_bytes(L, "little"),                                     # x = p: not canonical
                    (cv.p + cv.G[0]).to_bytes(L, "little") if (cv.p + cv.G[0]).bit_length() <= 8 * L - 2 else cv.p.to_bytes(L, "little"),
                    bytes([1]) + ac.enc_point(cv, None)[1:],                        # infinity flag with x != 0
                    ac.enc_point(cv, None)[:-1] + bytes([0xC0])]                    # infinity and sign flag
        if 8 * L - 2 > cv.p.bit_length():                                           # spare bits below the flags must be clear
            sp = bytearray(ac.enc_point(cv, cv.G))
            sp[-1] |= 0x20
            bad_list.append(bytes(sp))
        if curve == "bls12_377":                                                    # on the curve, outside the prime-order subgroup
            while True:
                xx = rng.randrange(cv.p)
                yy = po.fq_sqrt(cv, (xx * xx * xx + cv.a * xx + cv.b) % cv.p)
                if yy is None:
                    continue
                Q = po.pt_mul_raw(cv, cv.q, (xx, yy))
                if Q is not None:
                    break
            bad_list.append(ac.enc_point(cv, Q))
            bad_list.append(ac.enc_point(cv, po.pt_add(cv, Q, po.pt_mul(cv, 777, cv.G))))
        for b in bad_list:
            try:
                ac.dec_point(cv, b)
                raise AssertionError("the oracle decoder accepts a case meant to be refused: " + b.hex())
            except ac.DecodeError:
                pass
            pts.append(b)
            expect.append((-1, bytes(PB)))
        order = list(range(len(pts)))
        rng.shuffle(order)                                                          # bad encodings in the middle of a wave of good ones
        hin, pin = mem.put(b"".join(pts[i] for i in order))
        hout, pout = mem.new(len(pts) * PB)
        hst, pst = mem.new(4 * len(pts))
        eng.points_deserialize_dev(len(pts), pin, pout, pst)
        eng.sync()
        out, st = mem.get(hout, len(pts) * PB), mem.get(hst, 4 * len(pts))
        for k, i in enumerate(order):
            code = int.from_bytes(st[4 * k:4 * k + 4], "little", signed=True)
            assert (code, out[k * PB:(k + 1) * PB]) == expect[i], (curve, i, pts[i].hex(), code)
