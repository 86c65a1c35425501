"""arkworks-0.3 `CanonicalSerialize` / `CanonicalDeserialize` (compressed form) of the data types the reference trait carries
[REF barnett-smart-card-protocol/src/lib.rs:45-71], restated on the oracle's own objects (integers and (x, y) tuples).

TEST INFRASTRUCTURE ONLY (oracle/): it is the checker of the C-ABI functions `mp_*_serialize` / `mp_*_deserialize`
(include/mpshuffle.h) in tests/test_gpu_canonical.py and tests/test_oracle_golden.py.  It shares no code with the product's
own Python encoder (mental-poker_amd/canonical.py): this file works on big integers and the structured proof dictionary of
mp_oracle.py, the product works on wire bytes.

Conventions ([UPSTREAM-RECALL] -- arkworks 0.3.0 is not on disk; oracle/README.md lists the one-line Rust
experiment that settles each of them):
  * `Fp` / `Fr`            canonical integer, little-endian, ceil(modulus bits / 8) bytes  (32 for every scalar field here)
  * SW affine point        x little-endian in ceil((modulus bits + 2) / 8) bytes -- room for two flag bits -- with, in the last byte,
                           bit 7 = "y is the larger of (y, p - y)" and bit 6 = "point at infinity" (then x = 0 and bit 7 clear)
  * `usize`                u64 little-endian;   `Vec<T>` = u64 length, then the elements;   structs = their fields in order
  * deserialisation validates: canonical x (< p, no stray flag bits), x on the curve, point in the prime-order subgroup.
The known answer of SURVEY.md App. B (a STARK-curve point and its 32 bytes) is asserted in tests/test_oracle_golden.py.

The grouping of the shuffle proof's elements into `Vec`s is THIS BUILD'S (wire v1): upstream's
`shuffle::proof::Proof` lives in the absent `proof-essentials` crate [REF src/discrete_log_cards/mod.rs:103], so
`proof.serialized_size()` [REF examples/parameter_selection.rs:95] is reproduced for this grouping only.
"""
import mp_oracle as O


class DecodeError(ValueError):
    pass


def compressed_len(cv):
    return (cv.p.bit_length() + 2 + 7) // 8


def enc_usize(v):
    return int(v).to_bytes(8, "little")


def enc_scalar(v):
    return int(v).to_bytes(32, "little")


def enc_point(cv, P):
    """(x, y) or None (infinity) -> compressed bytes"""
    L = compressed_len(cv)
    if P is None:
        out = bytearray(L)
        out[-1] |= 0x40
        return bytes(out)
    x, y = P
    out = bytearray(x.to_bytes(L, "little"))
    if y > cv.p - y:
        out[-1] |= 0x80
    return bytes(out)


def in_prime_subgroup(cv, P):
    if cv.name not in O.COFACTOR:
        return True
    return O.pt_mul_raw(cv, cv.q, P) is None


def dec_point(cv, data):
    L = compressed_len(cv)
    if len(data) != L:
        raise DecodeError("point: %d bytes expected" % L)
    flags = data[-1] & 0xC0
    x = int.from_bytes(data[:-1] + bytes([data[-1] & 0x3F]), "little")
    if flags & 0x40:
        if x != 0 or flags & 0x80:
            raise DecodeError("point: malformed infinity")
        return None
    if x >= cv.p:
        raise DecodeError("point: x not canonical")
    y = O.fq_sqrt(cv, (x * x * x + cv.a * x + cv.b) % cv.p)
    if y is None:
        raise DecodeError("point: x is not on the curve")
    hi, lo = max(y, cv.p - y), min(y, cv.p - y)
    P = (x, hi if flags & 0x80 else lo)
    if y == 0 and flags & 0x80:
        raise DecodeError("point: sign flag on a point of order two")
    if not in_prime_subgroup(cv, P):
        raise DecodeError("point: not in the prime-order subgroup")
    return P


class Reader:
    def __init__(self, data):
        self.d, self.i = bytes(data), 0

    def take(self, k):
        if self.i + k > len(self.d):
            raise DecodeError("not enough data")
        v = self.d[self.i:self.i + k]
        self.i += k
        return v

    def usize(self):
        return int.from_bytes(self.take(8), "little")

    def scalar(self, cv):
        v = int.from_bytes(self.take(32), "little")
        if v >= cv.q:
            raise DecodeError("scalar out of range")
        return v

    def point(self, cv):
        return dec_point(cv, self.take(compressed_len(cv)))

    def end(self):
        if self.i != len(self.d):
            raise DecodeError("trailing bytes")


# ---- Parameters { m, n, enc_parameters { generator }, commit_parameters { g: Vec<_>, h }, generator } [REF mod.rs:37-43] ----------
def enc_parameters(pp):
    cv = pp.cv
    return (enc_usize(pp.m) + enc_usize(pp.n) + enc_point(cv, pp.G) + enc_usize(len(pp.ck)) + b"".join(enc_point(cv, P) for P in pp.ck) +
            enc_point(cv, pp.H) + enc_point(cv, pp.gen))


def dec_parameters(cv, data):
    r = Reader(data)
    m, n = r.usize(), r.usize()
    G = r.point(cv)
    k = r.usize()
    if k != n:
        raise DecodeError("parameters: commit key length != n")
    ck = [r.point(cv) for _ in range(k)]
    H, gen = r.point(cv), r.point(cv)
    r.end()
    return O.Params(cv, m, n, G, ck, H, gen)


# ---- MaskedCard = el_gamal::Ciphertext(c0, c1) [REF mod.rs:74]; a deck is a Vec of them ----------------------------------------------
def enc_deck(cv, deck):
    return enc_usize(len(deck)) + b"".join(enc_point(cv, c[0]) + enc_point(cv, c[1]) for c in deck)


def dec_deck(cv, data):
    r = Reader(data)
    k = r.usize()
    deck = [(r.point(cv), r.point(cv)) for _ in range(k)]
    r.end()
    return deck


# ---- ZKProofShuffle (this build's grouping) -----------------------------------------------------------------------------------------
def enc_proof(cv, pf):
    P = lambda v: enc_point(cv, v)
    S = enc_scalar
    VP = lambda vs: enc_usize(len(vs)) + b"".join(P(v) for v in vs)
    VS = lambda vs: enc_usize(len(vs)) + b"".join(S(v) for v in vs)
    pr, me = pf["product"], pf["mexp"]
    z, sv = pr["had"]["zero"], pr["svp"]
    return b"".join([
        VP(pf["cA"]), VP(pf["cB"]), P(pr["cb"]), VP(pr["had"]["cB"]),
        P(z["cA0"]), P(z["cBm"]), VP(z["cD"]), VS(z["abar"]), VS(z["bbar"]), S(z["rbar"]), S(z["sbar"]), S(z["tbar"]),
        P(sv["cd"]), P(sv["cdelta"]), P(sv["cDelta"]), VS(sv["at"]), VS(sv["bt"]), S(sv["rt"]), S(sv["st"]),
        P(me["cA0"]), VP(me["cB"]),
        enc_usize(len(me["E"])) + b"".join(P(e[0]) + P(e[1]) for e in me["E"]),
        VS(me["abar"]), S(me["rbar"]), S(me["bbar"]), S(me["sbar"]), S(me["taubar"]),
    ])


def dec_proof(cv, m, n, data):
    r = Reader(data)

    def VP(k):
        if r.usize() != k:
            raise DecodeError("proof: a vector has the wrong length")
        return [r.point(cv) for _ in range(k)]

    def VS(k):
        if r.usize() != k:
            raise DecodeError("proof: a vector has the wrong length")
        return [r.scalar(cv) for _ in range(k)]
    P, S = (lambda: r.point(cv)), (lambda: r.scalar(cv))
    cA, cB = VP(m), VP(m)
    cb = P()
    hB = VP(m)
    z = dict(cA0=P(), cBm=P())
    z["cD"] = VP(2 * m + 1)
    z["abar"], z["bbar"] = VS(n), VS(n)
    z["rbar"], z["sbar"], z["tbar"] = S(), S(), S()
    sv = dict(cd=P(), cdelta=P(), cDelta=P())
    sv["at"], sv["bt"] = VS(n), VS(n)
    sv["rt"], sv["st"] = S(), S()
    me = dict(cA0=P())
    me["cB"] = VP(2 * m)
    if r.usize() != 2 * m:
        raise DecodeError("proof: a vector has the wrong length")
    me["E"] = [(P(), P()) for _ in range(2 * m)]
    me["abar"] = VS(n)
    me["rbar"], me["bbar"], me["sbar"], me["taubar"] = S(), S(), S(), S()
    r.end()
    return dict(cA=cA, cB=cB, product=dict(cb=cb, had=dict(cB=hB, zero=z), svp=sv), mexp=me)


def proof_serialized_size(cv, m, n):
    """`proof.serialized_size()` [REF examples/parameter_selection.rs:95]: (11m + 8) points, (5n + 9) scalars, 11 vector lengths"""
    return (11 * m + 8) * compressed_len(cv) + (5 * n + 9) * 32 + 11 * 8
