"""arkworks-0.3 `CanonicalSerialize` / `CanonicalDeserialize` (compressed) for the associated types of the trait
[REF barnett-smart-card-protocol/src/lib.rs:45-71] and `serialized_size()` [REF examples/parameter_selection.rs:95]
(SURVEY.md section 8 row f2, Appendix B).  Host-side interop format for a Rust caller that holds serialised values; the
engine's own boundary stays the fixed-width "wire v1" of include/mpshuffle.h (x || y, no square roots on the way in).

Rules (ark-serialize 0.3, short-Weierstrass affine, compressed):
  * Fr            -> 32 bytes little-endian canonical integer
  * affine point  -> x little-endian in ceil((modulus_bits + 2) / 8) bytes (48 B on BLS12-377 G1); the two top bits of the last byte are flags:
                     bit 7 = (y > -y as canonical integers), bit 6 = point at infinity (then x = 0)
                     => 32 B for STARK (252-bit) and bn254 (254-bit), 33 B for secp256k1 (256-bit)
  * Vec<T>        -> u64 little-endian length, then the elements;  usize -> u64 little-endian
  * structs       -> fields in declaration order

Decompression needs a square root in Fq: Tonelli-Shanks (the STARK prime has 2-adicity 192, so this is the expensive
direction -- ~10^4 squarings per point -- which is why the hot path does not take compressed input).

The field order of `Parameters` is the reference's [REF src/discrete_log_cards/mod.rs:37-43]; `el_gamal::Parameters`,
`pedersen::CommitKey`, the sigma proofs and the shuffle proof are defined in the un-vendored `proof-essentials` crate, so
their grouping here follows this build's frozen element order ("wire v1") with one `Vec` per vector-valued
proof element -- sizes are exact for that grouping, byte-compatibility with upstream's struct layout is not claimed.
"""
import struct

# curve -> (p, a, b, modulus bits)
CURVE_FIELDS = {
    "stark": (2**251 + 17 * 2**192 + 1, 1, 0x06f21413efbe40de150e596d72f7a8c5609ad26c15c915c1f4cdfcb99cee9e89, 252),
    "bn254": (0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47, 0, 3, 254),
    "secp256k1": (2**256 - 2**32 - 977, 0, 7, 256),
    "bls12_377": (0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001, 0, 1, 377),
}
SCALAR_BYTES = 32


class SerializationError(ValueError):
    """ark_serialize::SerializationError (InvalidData / NotEnoughSpace)"""


def point_bytes(curve):
    """compressed size of a point"""
    return (CURVE_FIELDS[curve][3] + 2 + 7) // 8


def coord_bytes(curve):
    """width of one coordinate in wire v1 (8 bytes per ark-ff limb): 32, or 48 on BLS12-377"""
    return 8 * ((CURVE_FIELDS[curve][3] + 63) // 64)


def wire_point_bytes(curve):
    return 2 * coord_bytes(curve)


def _sqrt(a, p):
    """square root mod an odd prime (Tonelli-Shanks), or None"""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    s, t = 0, p - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    c, x, b, v = pow(z, t, p), pow(a, (t + 1) // 2, p), pow(a, t, p), s
    while b != 1:
        k, b2 = 0, b
        while b2 != 1:
            b2, k = b2 * b2 % p, k + 1
        w = pow(c, 1 << (v - k - 1), p)
        x, c = x * w % p, w * w % p
        b, v = b * c % p, k
    return x


# ---- scalars and points (to / from the engine's wire v1) -------------------------------------------------------
def scalar_serialize(wire32):
    if len(wire32) != 32:
        raise SerializationError("scalar: 32 bytes expected")
    return bytes(wire32)


def point_compress(curve, wire64):
    """wire v1 (x || y little-endian, all-zero bytes = infinity) -> compressed canonical bytes"""
    p, _, _, _ = CURVE_FIELDS[curve]
    nb, cb = point_bytes(curve), coord_bytes(curve)
    if len(wire64) != 2 * cb:
        raise SerializationError("point: %d bytes expected" % (2 * cb))
    if wire64 == bytes(2 * cb):
        out = bytearray(nb)
        out[-1] |= 0x40
        return bytes(out)
    x, y = int.from_bytes(wire64[:cb], "little"), int.from_bytes(wire64[cb:], "little")
    if x >= p or y >= p:
        raise SerializationError("point: coordinate out of range")
    out = bytearray(x.to_bytes(nb, "little"))
    if y > p - y:
        out[-1] |= 0x80
    return bytes(out)


def point_decompress(curve, data):
    """compressed canonical bytes -> wire v1; rejects x that is not on the curve and non-canonical encodings"""
    p, a, b, _ = CURVE_FIELDS[curve]
    nb, cb = point_bytes(curve), coord_bytes(curve)
    if len(data) != nb:
        raise SerializationError("point: %d bytes expected" % nb)
    raw = bytearray(data)
    flags = raw[-1] & 0xC0
    raw[-1] &= 0x3F
    x = int.from_bytes(raw, "little")
    if flags & 0x40:
        if x != 0 or flags & 0x80:
            raise SerializationError("point: bad infinity encoding")
        return bytes(2 * cb)
    if x >= p:
        raise SerializationError("point: x out of range")
    y = _sqrt((x * x * x + a * x + b) % p, p)
    if y is None:
        raise SerializationError("point: x is not on the curve")
    if (y > p - y) != bool(flags & 0x80):
        y = p - y
    if curve in SUBGROUP_ORDER and not _in_subgroup(curve, x, y):
        raise SerializationError("point: not in the prime-order subgroup")     # ark-ec 0.3 deserialize checks this too
    return x.to_bytes(cb, "little") + y.to_bytes(cb, "little")


# curves with a cofactor: group order q of the prime-order subgroup ([q]P == O is what ark-ec's deserialiser verifies)
SUBGROUP_ORDER = {"bls12_377": 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001}


def _in_subgroup(curve, x, y):
    p, a, _, _ = CURVE_FIELDS[curve]

    def add(P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        if P[0] == Q[0]:
            if (P[1] + Q[1]) % p == 0:
                return None
            lam = (3 * P[0] * P[0] + a) * pow(2 * P[1], -1, p) % p
        else:
            lam = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p) % p
        x3 = (lam * lam - P[0] - Q[0]) % p
        return (x3, (lam * (P[0] - x3) - P[1]) % p)
    acc, base, k = None, (x, y), SUBGROUP_ORDER[curve]
    while k:
        if k & 1:
            acc = add(acc, base)
        base = add(base, base)
        k >>= 1
    return acc is None


def _usize(v):
    return struct.pack("<Q", v)


class _Reader:
    def __init__(self, data):
        self.d, self.pos = bytes(data), 0

    def take(self, k):
        if self.pos + k > len(self.d):
            raise SerializationError("not enough data")
        out = self.d[self.pos:self.pos + k]
        self.pos += k
        return out

    def usize(self):
        return struct.unpack("<Q", self.take(8))[0]

    def done(self):
        if self.pos != len(self.d):
            raise SerializationError("trailing bytes")


def _points_ser(curve, wire):
    w = wire_point_bytes(curve)
    if len(wire) % w:
        raise SerializationError("points: whole points expected")
    return b"".join(point_compress(curve, wire[i:i + w]) for i in range(0, len(wire), w))


def _points_de(curve, r, k):
    nb = point_bytes(curve)
    return b"".join(point_decompress(curve, r.take(nb)) for _ in range(k))


# ---- MaskedCard = el_gamal::Ciphertext (two points); keys, cards and reveal tokens are single points -------------
def masked_card_serialize(curve, wire128):
    return _points_ser(curve, wire128)


def masked_card_deserialize(curve, data):
    r = _Reader(data)
    out = _points_de(curve, r, 2)
    r.done()
    return out


def deck_serialize(curve, wire):
    """Vec<MaskedCard>"""
    w = 2 * wire_point_bytes(curve)
    if len(wire) % w:
        raise SerializationError("deck: whole ciphertexts expected")
    return _usize(len(wire) // w) + _points_ser(curve, wire)


def deck_deserialize(curve, data):
    r = _Reader(data)
    k = r.usize()
    out = _points_de(curve, r, 2 * k)
    r.done()
    return out


# ---- Parameters { m, n, enc_parameters { generator }, commit_parameters { g: Vec, h }, generator } ---------------
def parameters_serialize(curve, m, n, raw):
    """`raw` = the engine's parameter block: G, ck_0..ck_{n-1}, H, gen (one wire point each)"""
    w = wire_point_bytes(curve)
    if len(raw) != w * (n + 3):
        raise SerializationError("parameters: %d bytes expected" % (w * (n + 3)))
    G, ck, H, gen = raw[:w], raw[w:w * (n + 1)], raw[w * (n + 1):w * (n + 2)], raw[w * (n + 2):]
    return (_usize(m) + _usize(n) + point_compress(curve, G) + _usize(n) + _points_ser(curve, ck)
            + point_compress(curve, H) + point_compress(curve, gen))


def parameters_deserialize(curve, data):
    """-> (m, n, raw)"""
    r = _Reader(data)
    m, n = r.usize(), r.usize()
    G = _points_de(curve, r, 1)
    k = r.usize()
    if k != n:
        raise SerializationError("parameters: commit key length != n")
    ck = _points_de(curve, r, n)
    H = _points_de(curve, r, 1)
    gen = _points_de(curve, r, 1)
    r.done()
    return m, n, G + ck + H + gen


# ---- sigma proofs: Schnorr (1 commitment) / Chaum-Pedersen (2 commitments) + response ---------------------------
def sigma_proof_serialize(curve, nbases, wire):
    w = wire_point_bytes(curve)
    if len(wire) != w * nbases + 32:
        raise SerializationError("sigma proof: bad length")
    return _points_ser(curve, wire[:w * nbases]) + wire[w * nbases:]


def sigma_proof_deserialize(curve, nbases, data):
    r = _Reader(data)
    out = _points_de(curve, r, nbases) + r.take(32)
    r.done()
    return out


# ---- the shuffle proof ----------------------------------------------------------------------------------------
def shuffle_proof_schema(m, n):
    """element order of wire v1: (name, kind 'G' | 'Z', count, is_vec)"""
    return [
        ("c_A", "G", m, True), ("c_B", "G", m, True),
        ("product.c_b", "G", 1, False),
        ("hadamard.c_B", "G", m, True),
        ("zero.c_A0", "G", 1, False), ("zero.c_Bm", "G", 1, False), ("zero.c_D", "G", 2 * m + 1, True),
        ("zero.a", "Z", n, True), ("zero.b", "Z", n, True),
        ("zero.r", "Z", 1, False), ("zero.s", "Z", 1, False), ("zero.t", "Z", 1, False),
        ("svp.c_d", "G", 1, False), ("svp.c_delta", "G", 1, False), ("svp.c_Delta", "G", 1, False),
        ("svp.a", "Z", n, True), ("svp.b", "Z", n, True), ("svp.r", "Z", 1, False), ("svp.s", "Z", 1, False),
        ("mexp.c_A0", "G", 1, False), ("mexp.c_B", "G", 2 * m, True),
        ("mexp.E", "G", 4 * m, True),          # 2m ciphertexts = 4m points
        ("mexp.a", "Z", n, True), ("mexp.r", "Z", 1, False), ("mexp.b", "Z", 1, False), ("mexp.s", "Z", 1, False),
        ("mexp.tau", "Z", 1, False),
    ]


def shuffle_proof_serialize(curve, m, n, wire):
    w = wire_point_bytes(curve)
    if len(wire) != (11 * m + 8) * w + (5 * n + 9) * 32:
        raise SerializationError("shuffle proof: bad length")
    out, pos = [], 0
    for name, kind, cnt, is_vec in shuffle_proof_schema(m, n):
        if is_vec:
            out.append(_usize(cnt // 2 if name == "mexp.E" else cnt))
        if kind == "G":
            out.append(_points_ser(curve, wire[pos:pos + w * cnt]))
            pos += w * cnt
        else:
            out.append(wire[pos:pos + 32 * cnt])
            pos += 32 * cnt
    assert pos == len(wire)
    return b"".join(out)


def shuffle_proof_deserialize(curve, m, n, data):
    r, out = _Reader(data), []
    for name, kind, cnt, is_vec in shuffle_proof_schema(m, n):
        if is_vec and r.usize() != (cnt // 2 if name == "mexp.E" else cnt):
            raise SerializationError("shuffle proof: %s has the wrong length" % name)
        out.append(_points_de(curve, r, cnt) if kind == "G" else r.take(32 * cnt))
    r.done()
    return b"".join(out)


def shuffle_proof_serialized_size(curve, m, n):
    """`proof.serialized_size()` [REF examples/parameter_selection.rs:95] for this build's proof grouping"""
    g, size = point_bytes(curve), 0
    for _, kind, cnt, is_vec in shuffle_proof_schema(m, n):
        size += (8 if is_vec else 0) + cnt * (g if kind == "G" else SCALAR_BYTES)
    return size
