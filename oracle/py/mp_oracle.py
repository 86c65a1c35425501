"""
mp_oracle.py -- TEST INFRASTRUCTURE ONLY (oracle).  Independent Python big-integer
restatement of the hot path of geometryxyz/mental-poker:

    DLCards::shuffle_and_remask   [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:380-418]
    DLCards::verify_shuffle       [REF .../discrete_log_cards/mod.rs:420-443]
    Remask for MaskedCard         [REF .../discrete_log_cards/remasking.rs:9-22]
    Mask for Card                 [REF .../discrete_log_cards/masking.rs:10-20]

PARITY UNPINNED.  The arithmetic the reference calls lives in crates that are not under
/root/reference (`proof-essentials` / `starknet-curve` from geometryresearch/proof-toolbox, git
dependency without rev/tag -- Cargo.toml:18,20 -- and arkworks 0.3.0, Cargo.toml:10-15), there is no
Rust toolchain in this image and the reference holds no golden vector (every test draws from
thread_rng(): tests.rs:50,82,127,177).  This file therefore restates the *published* algorithms:

  * Bayer & Groth, "Efficient Zero-Knowledge Argument for Correctness of a Shuffle", EUROCRYPT 2012,
    sections 4 (multi-exponentiation argument), 5 (product argument), 5.1 (Hadamard), 5.2 (zero
    argument), 5.3 (single value product) -- non-FFT variant, as the reference states it uses
    [REF examples/parameter_selection.rs:3-5];
  * arkworks-0.3 conventions for encodings, `Fp::rand` and `ark_marlin::rng::FiatShamirRng<Blake2s>`
    (Blake2s digest -> ChaCha20Rng seed; absorb = H(new || old seed)), the RNG the reference seeds with
    b"Shuffle Proof" [REF mod.rs:84,408,436];
  * ElGamal E(M; r) = (r*G, M + r*pk) and Pedersen com(v; r) = r*H + sum v_j*G_j as used at
    [REF mod.rs:193-200, 110-112].

and FREEZES the transcript below as "mpshuffle transcript v1".  What IS mathematically pinned (and
what the parity tests assert bit-exactly) is: the re-encrypted deck, every group element of the proof
(canonical affine coordinates do not depend on the algorithm used to reach them) and accept / reject
with the reference's check name "Hadamard Product (5.1)" [REF tests.rs:213-226].

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The curve constants and KATs are those of SURVEY.md Appendix C.
"""
import hashlib
import struct

# ----------------------------------------------------------------------------------------------
# Curves (short Weierstrass y^2 = x^3 + a x + b over Fp, prime order q subgroup, generator G)
# ----------------------------------------------------------------------------------------------


class Curve:
    def __init__(self, name, cid, p, a, b, q, gx, gy):
        self.name, self.cid, self.p, self.a, self.b, self.q = name, cid, p, a, b, q
        self.G = (gx, gy)
        self.fq_bits = q.bit_length()
        self.fq_bytes = 8 * ((p.bit_length() + 63) // 64)     # ark-ff limbs: 32 B for <= 256-bit p, 48 B for BLS12-377
        # arkworks REPR_SHAVE_BITS for a 4x64-limb field: 256 - modulus bits
        self.fr_shave = 256 - self.fq_bits
        self.R = 1 << 256
        self.Rinv_q = pow(self.R, -1, q)

    def is_on_curve(self, P):
        if P is None:
            return True
        x, y = P
        return (y * y - (x * x * x + self.a * x + self.b)) % self.p == 0


STARK = Curve(
    "stark", 0,
    (1 << 251) + 17 * (1 << 192) + 1,
    1,
    0x06f21413efbe40de150e596d72f7a8c5609ad26c15c915c1f4cdfcb99cee9e89,
    0x0800000000000010ffffffffffffffffb781126dcae7b2321e66a241adc64d2f,
    0x1ef15c18599971b7beced415a40f0c7deacfd9b0d1819e03d723d8bc943cfca,
    0x5668060aa49730b7be4801df46ec62de53ecd11abe43a32873000c36e8dc1f,
)
BN254 = Curve(
    "bn254", 1,
    21888242871839275222246405745257275088696311157297823662689037894645226208583,
    0, 3,
    21888242871839275222246405745257275088548364400416034343698204186575808495617,
    1, 2,
)
SECP256K1 = Curve(
    "secp256k1", 2,
    (1 << 256) - (1 << 32) - 977,
    0, 7,
    0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
    0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
    0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8,
)
# BLS12-377 G1 (examples/parameter_selection.rs [REF barnett-smart-card-protocol/examples/parameter_selection.rs:25]);
# constants: SURVEY.md App. C.  Cofactor != 1: every point the protocol touches is a multiple of G (prime-order subgroup).
BLS12_377 = Curve(
    "bls12_377", 3,
    0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001,
    0, 1,
    0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001,
    0x008848defe740a67c8fc6225bf87ff5485951e2caa9d41bb188282c8bd37cb5cd5481512ffcd394eeab9b16eb21be9ef,
    0x01914a69c5102eff1f674f5d30afeec4bd7fb348ca3e52d96d182ad44fb82305c2fe3d3634a9591afd82de55559c8ea6,
)
CURVES = {c.name: c for c in (STARK, BN254, SECP256K1, BLS12_377)}

# ----------------------------------------------------------------------------------------------
# Group law (Jacobian internally, affine tuples / None = infinity at the interface)
# ----------------------------------------------------------------------------------------------


def jac_double(cv, P):
    X, Y, Z = P
    p = cv.p
    if Z == 0 or Y == 0:
        return (1, 1, 0)
    XX = X * X % p
    YY = Y * Y % p
    ZZ = Z * Z % p
    S = 4 * X * YY % p
    M = (3 * XX + cv.a * ZZ * ZZ) % p
    X3 = (M * M - 2 * S) % p
    Y3 = (M * (S - X3) - 8 * YY * YY) % p
    Z3 = 2 * Y * Z % p
    return (X3, Y3, Z3)


def jac_add(cv, P, Q):
    p = cv.p
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    if Z1 == 0:
        return Q
    if Z2 == 0:
        return P
    Z1Z1 = Z1 * Z1 % p
    Z2Z2 = Z2 * Z2 % p
    U1 = X1 * Z2Z2 % p
    U2 = X2 * Z1Z1 % p
    S1 = Y1 * Z2 * Z2Z2 % p
    S2 = Y2 * Z1 * Z1Z1 % p
    if U1 == U2:
        if S1 == S2:
            return jac_double(cv, P)
        return (1, 1, 0)
    H = (U2 - U1) % p
    R = (S2 - S1) % p
    HH = H * H % p
    HHH = H * HH % p
    V = U1 * HH % p
    X3 = (R * R - HHH - 2 * V) % p
    Y3 = (R * (V - X3) - S1 * HHH) % p
    Z3 = Z1 * Z2 * H % p
    return (X3, Y3, Z3)


def to_jac(P):
    return (1, 1, 0) if P is None else (P[0], P[1], 1)


def to_affine(cv, J):
    X, Y, Z = J
    if Z == 0:
        return None
    zi = pow(Z, -1, cv.p)
    zi2 = zi * zi % cv.p
    return (X * zi2 % cv.p, Y * zi2 * zi % cv.p)


def pt_add(cv, P, Q):
    return to_affine(cv, jac_add(cv, to_jac(P), to_jac(Q)))


def pt_neg(cv, P):
    return None if P is None else (P[0], (-P[1]) % cv.p)


def jac_mul(cv, k, P):
    """MSB-first double-and-add over the canonical scalar (ark-ec `mul`)."""
    k %= cv.q
    acc = (1, 1, 0)
    J = to_jac(P)
    for bit in bin(k)[2:] if k else "":
        acc = jac_double(cv, acc)
        if bit == "1":
            acc = jac_add(cv, acc, J)
    return acc


def pt_mul(cv, k, P):
    return to_affine(cv, jac_mul(cv, k, P))


def pt_mul_raw(cv, k, P):
    """[k]P for an integer k >= 0 that is NOT reduced modulo the group order (cofactor clearing, subgroup tests)"""
    acc = (1, 1, 0)
    J = to_jac(P)
    for bit in bin(k)[2:] if k else "":
        acc = jac_double(cv, acc)
        if bit == "1":
            acc = jac_add(cv, acc, J)
    return to_affine(cv, acc)


def msm(cv, scalars, points):
    """sum k_i * P_i (plain per-term double-and-add; the result is a canonical group element)."""
    acc = (1, 1, 0)
    for k, P in zip(scalars, points):
        if P is None or k % cv.q == 0:
            continue
        acc = jac_add(cv, acc, jac_mul(cv, k, P))
    return to_affine(cv, acc)


# ----------------------------------------------------------------------------------------------
# ChaCha20 (rand_chacha::ChaCha20Rng: 64-bit block counter in words 12,13; stream id 0) and the
# arkworks FiatShamirRng<Blake2s> built on it.
# ----------------------------------------------------------------------------------------------

_M32 = 0xFFFFFFFF


def _rotl(v, c):
    return ((v << c) & _M32) | (v >> (32 - c))


def chacha20_block(key_words, counter, w14=0, w15=0, w13=None):
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [
        counter & _M32, (counter >> 32) & _M32 if w13 is None else w13, w14, w15]
    x = st[:]

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & _M32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & _M32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & _M32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & _M32; x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(x[i] + st[i]) & _M32 for i in range(16)]


class ChaCha20Rng:
    """`ChaCha20Rng::from_seed(seed)`; we only ever draw u64s, so the stream is word pairs."""

    def __init__(self, seed32):
        assert len(seed32) == 32
        self.key = struct.unpack("<8I", seed32)
        self.counter = 0
        self.buf = []

    def next_u32(self):
        if not self.buf:
            self.buf = chacha20_block(self.key, self.counter)
            self.counter += 1
        return self.buf.pop(0)

    def next_u64(self):
        lo = self.next_u32()
        hi = self.next_u32()
        return lo | (hi << 32)


def fr_rand(cv, rng):
    """arkworks-0.3 `Fp::rand`: 4 u64 limbs (limb 0 first), clear the top REPR_SHAVE_BITS of limb 3,
    accept if < modulus; the accepted limbs ARE the Montgomery representation (value = limbs/R mod q)."""
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        if cv.fr_shave:
            limbs[3] &= (1 << (64 - cv.fr_shave)) - 1
        v = limbs[0] | (limbs[1] << 64) | (limbs[2] << 128) | (limbs[3] << 192)
        if v < cv.q:
            return v * cv.Rinv_q % cv.q


def blake2s(data):
    return hashlib.blake2s(data).digest()


class FiatShamirRng:
    """ark_marlin::rng::FiatShamirRng<Blake2s> (0.3): seed = H(input); absorb: seed = H(new || seed)."""

    def __init__(self, seed_bytes):
        self.seed = blake2s(seed_bytes)
        self.r = ChaCha20Rng(self.seed)

    def absorb(self, data):
        self.seed = blake2s(bytes(data) + self.seed)
        self.r = ChaCha20Rng(self.seed)

    def next_u64(self):
        return self.r.next_u64()


# ----------------------------------------------------------------------------------------------
# Encodings
# ----------------------------------------------------------------------------------------------


def fe_bytes(v):
    """scalar (Fr): 32 B little-endian"""
    return int(v).to_bytes(32, "little")


# Width of a base-field coordinate in bytes (ark-ff limbs * 8): 32 for the 256-bit curves, 48 for BLS12-377.  The
# encoders below have no curve argument (points are plain tuples), so the width is ambient: every entry point that takes
# a curve or a Params sets it for the duration of the call (`_with_curve`), tests use `with curve_ctx(cv):`.
_FQB = 32


class curve_ctx:
    def __init__(self, cv):
        self.nb = cv.fq_bytes

    def __enter__(self):
        global _FQB
        self.old, _FQB = _FQB, self.nb

    def __exit__(self, *a):
        global _FQB
        _FQB = self.old


def _with_curve(f):
    import functools

    @functools.wraps(f)
    def g(cv_or_pp, *a, **k):
        with curve_ctx(getattr(cv_or_pp, "cv", cv_or_pp)):
            return f(cv_or_pp, *a, **k)
    return g


def point_bytes():
    return 2 * _FQB


def fq_bytes(v):
    return int(v).to_bytes(_FQB, "little")


def pt_tobytes(P):
    """ark `ToBytes` of an affine point: x || y || infinity flag (GroupAffine::zero() = (0, 1, true))."""
    if P is None:
        return fq_bytes(0) + fq_bytes(1) + b"\x01"
    return fq_bytes(P[0]) + fq_bytes(P[1]) + b"\x00"


def pt_wire(P):
    """boundary ("wire") encoding: x LE || y LE (64 B; 96 B on BLS12-377); infinity = all-zero bytes ((0,0) is on none
    of the curves since b != 0; a flag bit would collide with y on the 256-bit secp256k1 field)."""
    if P is None:
        return bytes(2 * _FQB)
    return fq_bytes(P[0]) + fq_bytes(P[1])


def pt_from_wire(b):
    assert len(b) == 2 * _FQB
    if b == bytes(2 * _FQB):
        return None
    return (int.from_bytes(b[:_FQB], "little"), int.from_bytes(b[_FQB:], "little"))


def ct_tobytes(ct):
    return pt_tobytes(ct[0]) + pt_tobytes(ct[1])


# ----------------------------------------------------------------------------------------------
# ElGamal / Pedersen
# ----------------------------------------------------------------------------------------------


class Params:
    """shuffle::Parameters::new(&enc_pp, pk, &ck, &generator) [REF mod.rs:397-402]; m, n [REF mod.rs:38-39]"""

    def __init__(self, cv, m, n, G, ck, H, gen):
        self.cv, self.m, self.n, self.G, self.ck, self.H, self.gen = cv, m, n, G, list(ck), H, gen
        assert len(self.ck) == n


def commit(pp, v, r):
    """Pedersen com(v; r) = r*H + sum_j v_j*G_j, len(v) <= n."""
    assert len(v) <= pp.n
    return msm(pp.cv, list(v) + [r], pp.ck[:len(v)] + [pp.H])


def ct_add(cv, A, B):
    return (pt_add(cv, A[0], B[0]), pt_add(cv, A[1], B[1]))


def encrypt(pp, pk, M, r):
    """ElGamal::encrypt: (r*G, M + r*pk)  [REF masking.rs:17]"""
    cv = pp.cv
    return (pt_mul(cv, r, pp.G), pt_add(cv, M, pt_mul(cv, r, pk)))


def remask(pp, pk, ct, alpha):
    """`*self + zero.mask(pp, pk, alpha)`  [REF remasking.rs:16-18]"""
    return ct_add(pp.cv, ct, encrypt(pp, pk, None, alpha))


def ct_msm(cv, scalars, cts):
    return (msm(cv, scalars, [c[0] for c in cts]), msm(cv, scalars, [c[1] for c in cts]))


# ----------------------------------------------------------------------------------------------
# Bayer-Groth (transcript v1).  `prng` = prover randomness (ChaCha20Rng), `fs` = FiatShamirRng.
# All vectors 0-based; paper indices noted where they matter.
# ----------------------------------------------------------------------------------------------

CHECK_NAMES = {
    0: "Ok",
    1: "Hadamard Product (5.1)",
    2: "Zero Argument (5.2)",
    3: "Single Value Product (5.3)",
    4: "Multi-Exponentiation Argument (4)",
}


class VerifyError(Exception):
    def __init__(self, code):
        super().__init__(CHECK_NAMES[code])
        self.code = code


def _pts_bytes(pts):
    return b"".join(pt_tobytes(P) for P in pts)


def bilinear(cv, a, b, ypow):
    """a * b = sum_{j=1..n} a_j b_j y^j   (ypow[j] = y^(j+1))"""
    return sum(x * w * yp for x, w, yp in zip(a, b, ypow)) % cv.q


def zero_prove(pp, prng, fs, A, r, B, s, ypow):
    """Zero argument (5.2).  A = a_1..a_m, B = b_0..b_{m-1}; sum_i a_i * b_{i-1} = 0."""
    cv, q, m, n = pp.cv, pp.cv.q, len(A), pp.n
    a0 = [fr_rand(cv, prng) for _ in range(n)]
    bm = [fr_rand(cv, prng) for _ in range(n)]
    r0 = fr_rand(cv, prng)
    sm = fr_rand(cv, prng)
    t = [fr_rand(cv, prng) for _ in range(2 * m + 1)]
    t[m + 1] = 0
    Aa = [a0] + A            # a_0..a_m
    Bb = B + [bm]            # b_0..b_m
    ra = [r0] + r
    sb = s + [sm]
    d = [0] * (2 * m + 1)
    for i in range(m + 1):
        for j in range(m + 1):
            k = m - j + i
            d[k] = (d[k] + bilinear(cv, Aa[i], Bb[j], ypow)) % q
    assert d[m + 1] == 0, "zero-argument witness does not satisfy the statement"
    cA0 = commit(pp, a0, r0)
    cBm = commit(pp, bm, sm)
    cD = [commit(pp, [d[k]], t[k]) for k in range(2 * m + 1)]
    fs.absorb(_pts_bytes([cA0, cBm] + cD))
    x = fr_rand(cv, fs)
    xp = [pow(x, e, q) for e in range(2 * m + 1)]
    abar = [sum(xp[i] * Aa[i][l] for i in range(m + 1)) % q for l in range(n)]
    rbar = sum(xp[i] * ra[i] for i in range(m + 1)) % q
    bbar = [sum(xp[m - j] * Bb[j][l] for j in range(m + 1)) % q for l in range(n)]
    sbar = sum(xp[m - j] * sb[j] for j in range(m + 1)) % q
    tbar = sum(xp[k] * t[k] for k in range(2 * m + 1)) % q
    return dict(cA0=cA0, cBm=cBm, cD=cD, abar=abar, bbar=bbar, rbar=rbar, sbar=sbar, tbar=tbar)


def zero_verify(pp, fs, cA, cB, ypow, pf):
    cv, q, m = pp.cv, pp.cv.q, len(cA)
    fs.absorb(_pts_bytes([pf["cA0"], pf["cBm"]] + pf["cD"]))
    x = fr_rand(cv, fs)
    xp = [pow(x, e, q) for e in range(2 * m + 1)]
    if pf["cD"][m + 1] is not None:
        raise VerifyError(2)
    lhs = msm(cv, xp[:m + 1], [pf["cA0"]] + cA)
    if lhs != commit(pp, pf["abar"], pf["rbar"]):
        raise VerifyError(2)
    lhs = msm(cv, [xp[m - j] for j in range(m + 1)], cB + [pf["cBm"]])
    if lhs != commit(pp, pf["bbar"], pf["sbar"]):
        raise VerifyError(2)
    lhs = msm(cv, xp, pf["cD"])
    if lhs != commit(pp, [bilinear(cv, pf["abar"], pf["bbar"], ypow)], pf["tbar"]):
        raise VerifyError(2)


def hadamard_prove(pp, prng, fs, cA, cb, A, r, bvec, sb):
    """Hadamard product argument (5.1): bvec = A_1 o ... o A_m."""
    cv, q, m, n = pp.cv, pp.cv.q, len(A), pp.n
    Bp = [A[0]]
    for i in range(1, m):
        Bp.append([u * v % q for u, v in zip(Bp[-1], A[i])])
    assert Bp[-1] == bvec
    s = [r[0]] + [fr_rand(cv, prng) for _ in range(m - 2)] + [sb]
    cBp = [cA[0]] + [commit(pp, Bp[i], s[i]) for i in range(1, m - 1)] + [cb]
    fs.absorb(_pts_bytes(cBp))
    x = fr_rand(cv, fs)
    y = fr_rand(cv, fs)
    xp = [pow(x, e, q) for e in range(m + 1)]
    ypow = [pow(y, j + 1, q) for j in range(n)]
    # zero statement / witness
    zA = A[1:] + [[q - 1] * n]
    zr = r[1:] + [0]
    zB = [[xp[i + 1] * v % q for v in Bp[i]] for i in range(m - 1)]
    zB.append([sum(xp[i + 1] * Bp[i + 1][l] for i in range(m - 1)) % q for l in range(n)])
    zs = [xp[i + 1] * s[i] % q for i in range(m - 1)] + [sum(xp[i + 1] * s[i + 1] for i in range(m - 1)) % q]
    zero = zero_prove(pp, prng, fs, zA, zr, zB, zs, ypow)
    return dict(cB=cBp, zero=zero)


def hadamard_verify(pp, fs, cA, cb, pf):
    cv, q, m, n = pp.cv, pp.cv.q, len(cA), pp.n
    cBp = pf["cB"]
    if len(cBp) != m or cBp[0] != cA[0] or cBp[m - 1] != cb:
        raise VerifyError(1)
    fs.absorb(_pts_bytes(cBp))
    x = fr_rand(cv, fs)
    y = fr_rand(cv, fs)
    xp = [pow(x, e, q) for e in range(m + 1)]
    ypow = [pow(y, j + 1, q) for j in range(n)]
    c_minus1 = pt_neg(cv, pp_gsum(pp))
    zcA = cA[1:] + [c_minus1]
    zcB = [pt_mul(cv, xp[i + 1], cBp[i]) for i in range(m - 1)]
    zcB.append(msm(cv, [xp[i + 1] for i in range(m - 1)], [cBp[i + 1] for i in range(m - 1)]))
    zero_verify(pp, fs, zcA, zcB, ypow, pf["zero"])


def pp_gsum(pp):
    acc = (1, 1, 0)
    for P in pp.ck:
        acc = jac_add(pp.cv, acc, to_jac(P))
    return to_affine(pp.cv, acc)


def svp_prove(pp, prng, fs, ca, b, a, r):
    """Single value product argument (5.3): prod a_i = b."""
    cv, q, n = pp.cv, pp.cv.q, pp.n
    bp = [a[0]]
    for i in range(1, n):
        bp.append(bp[-1] * a[i] % q)
    assert bp[-1] == b % q
    d = [fr_rand(cv, prng) for _ in range(n)]
    rd = fr_rand(cv, prng)
    delta = [d[0]] + [fr_rand(cv, prng) for _ in range(n - 2)] + [0]
    s1 = fr_rand(cv, prng)
    sx = fr_rand(cv, prng)
    cd = commit(pp, d, rd)
    cdelta = commit(pp, [(-delta[i] * d[i + 1]) % q for i in range(n - 1)], s1)
    cDelta = commit(pp, [(delta[i + 1] - a[i + 1] * delta[i] - bp[i] * d[i + 1]) % q for i in range(n - 1)], sx)
    fs.absorb(_pts_bytes([cd, cdelta, cDelta]))
    x = fr_rand(cv, fs)
    at = [(x * a[i] + d[i]) % q for i in range(n)]
    bt = [(x * bp[i] + delta[i]) % q for i in range(n)]
    rt = (x * r + rd) % q
    st = (x * sx + s1) % q
    return dict(cd=cd, cdelta=cdelta, cDelta=cDelta, at=at, bt=bt, rt=rt, st=st)


def svp_verify(pp, fs, ca, b, pf):
    cv, q, n = pp.cv, pp.cv.q, pp.n
    fs.absorb(_pts_bytes([pf["cd"], pf["cdelta"], pf["cDelta"]]))
    x = fr_rand(cv, fs)
    at, bt = pf["at"], pf["bt"]
    if msm(cv, [x, 1], [ca, pf["cd"]]) != commit(pp, at, pf["rt"]):
        raise VerifyError(3)
    v = [(x * bt[i + 1] - bt[i] * at[i + 1]) % q for i in range(n - 1)]
    if msm(cv, [x, 1], [pf["cDelta"], pf["cdelta"]]) != commit(pp, v, pf["st"]):
        raise VerifyError(3)
    if bt[0] != at[0] or bt[n - 1] != x * b % q:
        raise VerifyError(3)


def product_prove(pp, prng, fs, cA, b, A, r):
    """Product argument (5): prod over all entries of A = b."""
    cv, q, m, n = pp.cv, pp.cv.q, len(A), pp.n
    bvec = A[0]
    for i in range(1, m):
        bvec = [u * v % q for u, v in zip(bvec, A[i])]
    sb = fr_rand(cv, prng)
    cb = commit(pp, bvec, sb)
    fs.absorb(_pts_bytes([cb]))
    had = hadamard_prove(pp, prng, fs, cA, cb, A, r, bvec, sb)
    svp = svp_prove(pp, prng, fs, cb, b, bvec, sb)
    return dict(cb=cb, had=had, svp=svp)


def product_verify(pp, fs, cA, b, pf):
    fs.absorb(_pts_bytes([pf["cb"]]))
    hadamard_verify(pp, fs, cA, pf["cb"], pf["had"])
    svp_verify(pp, fs, pf["cb"], b, pf["svp"])


def mexp_prove(pp, pk, prng, fs, Crows, C, cA, A, r, rho):
    """Multi-exponentiation argument (4): C = E(0; rho) + sum_i a_i . Crows_i."""
    cv, q, m, n = pp.cv, pp.cv.q, len(A), pp.n
    a0 = [fr_rand(cv, prng) for _ in range(n)]
    r0 = fr_rand(cv, prng)
    b = [fr_rand(cv, prng) for _ in range(2 * m)]
    s = [fr_rand(cv, prng) for _ in range(2 * m)]
    tau = [fr_rand(cv, prng) for _ in range(2 * m)]
    b[m], s[m], tau[m] = 0, 0, rho % q
    Aa = [a0] + A            # a_0..a_m
    cA0 = commit(pp, a0, r0)
    cB = [commit(pp, [b[k]], s[k]) for k in range(2 * m)]
    E = []
    for k in range(2 * m):
        acc = (pt_mul(cv, tau[k], pp.G),
               pt_add(cv, pt_mul(cv, b[k], pp.gen), pt_mul(cv, tau[k], pk)))
        for i in range(1, m + 1):
            j = k - m + i
            if 0 <= j <= m:
                acc = ct_add(cv, acc, ct_msm(cv, Aa[j], Crows[i - 1]))
        E.append(acc)
    assert E[m] == C, "multi-exp witness does not open the statement"
    fs.absorb(_pts_bytes([cA0] + cB) + b"".join(ct_tobytes(e) for e in E))
    x = fr_rand(cv, fs)
    xp = [pow(x, e, q) for e in range(2 * m)]
    ra = [r0] + r
    abar = [sum(xp[j] * Aa[j][l] for j in range(m + 1)) % q for l in range(n)]
    rbar = sum(xp[j] * ra[j] for j in range(m + 1)) % q
    bbar = sum(xp[k] * b[k] for k in range(2 * m)) % q
    sbar = sum(xp[k] * s[k] for k in range(2 * m)) % q
    taubar = sum(xp[k] * tau[k] for k in range(2 * m)) % q
    return dict(cA0=cA0, cB=cB, E=E, abar=abar, rbar=rbar, bbar=bbar, sbar=sbar, taubar=taubar)


def mexp_verify(pp, pk, fs, Crows, C, cA, pf):
    cv, q, m, n = pp.cv, pp.cv.q, len(cA), pp.n
    cB, E = pf["cB"], pf["E"]
    fs.absorb(_pts_bytes([pf["cA0"]] + cB) + b"".join(ct_tobytes(e) for e in E))
    x = fr_rand(cv, fs)
    xp = [pow(x, e, q) for e in range(2 * m)]
    if cB[m] is not None:
        raise VerifyError(4)
    if E[m] != C:
        raise VerifyError(4)
    if msm(cv, xp[:m + 1], [pf["cA0"]] + cA) != commit(pp, pf["abar"], pf["rbar"]):
        raise VerifyError(4)
    if msm(cv, xp, cB) != commit(pp, [pf["bbar"]], pf["sbar"]):
        raise VerifyError(4)
    lhs = ct_msm(cv, xp, E)
    rhs = (pt_mul(cv, pf["taubar"], pp.G),
           pt_add(cv, pt_mul(cv, pf["bbar"], pp.gen), pt_mul(cv, pf["taubar"], pk)))
    for i in range(1, m + 1):
        rhs = ct_add(cv, rhs, ct_msm(cv, [xp[m - i] * v % q for v in pf["abar"]], Crows[i - 1]))
    if lhs != rhs:
        raise VerifyError(4)


@_with_curve
def statement_bytes(pp, pk, deck, shuffled):
    out = pt_tobytes(pp.G) + pt_tobytes(pk) + pt_tobytes(pp.gen)
    out += _pts_bytes(pp.ck) + pt_tobytes(pp.H)
    out += b"".join(ct_tobytes(c) for c in deck)
    out += b"".join(ct_tobytes(c) for c in shuffled)
    out += struct.pack("<QQ", pp.m, pp.n)
    return out


SHUFFLE_RNG_SEED = b"Shuffle Proof"  # [REF mod.rs:84]


@_with_curve
def shuffle_prove(pp, pk, deck, shuffled, perm, rho, prng):
    """ShuffleArgument::prove  [REF mod.rs:409-415].  shuffled[i] = deck[perm[i]] + E(0; rho[i])."""
    cv, q, m, n = pp.cv, pp.cv.q, pp.m, pp.n
    N = m * n
    fs = FiatShamirRng(SHUFFLE_RNG_SEED)
    fs.absorb(statement_bytes(pp, pk, deck, shuffled))
    r = [fr_rand(cv, prng) for _ in range(m)]
    s = [fr_rand(cv, prng) for _ in range(m)]
    a = [perm[i] + 1 for i in range(N)]
    cA = [commit(pp, a[k * n:(k + 1) * n], r[k]) for k in range(m)]
    fs.absorb(_pts_bytes(cA))
    x = fr_rand(cv, fs)
    b = [pow(x, perm[i] + 1, q) for i in range(N)]
    cB = [commit(pp, b[k * n:(k + 1) * n], s[k]) for k in range(m)]
    fs.absorb(_pts_bytes(cB))
    y = fr_rand(cv, fs)
    z = fr_rand(cv, fs)
    gsum = pp_gsum(pp)
    # product argument on d - z
    dz = [(y * a[i] + b[i] - z) % q for i in range(N)]
    t = [(y * r[k] + s[k]) % q for k in range(m)]
    cDz = [msm(cv, [y, 1, (-z) % q], [cA[k], cB[k], gsum]) for k in range(m)]
    prod = 1
    for i in range(1, N + 1):
        prod = prod * (y * i + pow(x, i, q) - z) % q
    product = product_prove(pp, prng, fs, cDz, prod, [dz[k * n:(k + 1) * n] for k in range(m)], t)
    # multi-exponentiation argument
    rho_hat = (-sum(rho[i] * b[i] for i in range(N))) % q
    Cx = ct_msm(cv, [pow(x, i + 1, q) for i in range(N)], deck)
    Crows = [shuffled[k * n:(k + 1) * n] for k in range(m)]
    mexp = mexp_prove(pp, pk, prng, fs, Crows, Cx, cB, [b[k * n:(k + 1) * n] for k in range(m)], s, rho_hat)
    return dict(cA=cA, cB=cB, product=product, mexp=mexp)


@_with_curve
def shuffle_verify(pp, pk, deck, shuffled, proof):
    """ShuffleArgument::verify  [REF mod.rs:437-442]; raises VerifyError(code) on the first failing check."""
    cv, q, m, n = pp.cv, pp.cv.q, pp.m, pp.n
    N = m * n
    fs = FiatShamirRng(SHUFFLE_RNG_SEED)
    fs.absorb(statement_bytes(pp, pk, deck, shuffled))
    cA, cB = proof["cA"], proof["cB"]
    fs.absorb(_pts_bytes(cA))
    x = fr_rand(cv, fs)
    fs.absorb(_pts_bytes(cB))
    y = fr_rand(cv, fs)
    z = fr_rand(cv, fs)
    gsum = pp_gsum(pp)
    cDz = [msm(cv, [y, 1, (-z) % q], [cA[k], cB[k], gsum]) for k in range(m)]
    prod = 1
    for i in range(1, N + 1):
        prod = prod * (y * i + pow(x, i, q) - z) % q
    product_verify(pp, fs, cDz, prod, proof["product"])
    Cx = ct_msm(cv, [pow(x, i + 1, q) for i in range(N)], deck)
    Crows = [shuffled[k * n:(k + 1) * n] for k in range(m)]
    mexp_verify(pp, pk, fs, Crows, Cx, cB, proof["mexp"])
    return 0


# ----------------------------------------------------------------------------------------------
# The two boundary functions
# ----------------------------------------------------------------------------------------------


@_with_curve
def shuffle_and_remask(pp, pk, deck, masking_factors, perm, prover_seed):
    """DLCards::shuffle_and_remask [REF mod.rs:380-418]; `permute_array(v)[i] = v[mapping[i]]`."""
    permuted = [deck[perm[i]] for i in range(len(deck))]
    shuffled = [remask(pp, pk, c, f) for c, f in zip(permuted, masking_factors)]
    prng = ChaCha20Rng(prover_seed)
    proof = shuffle_prove(pp, pk, deck, shuffled, perm, masking_factors, prng)
    return shuffled, proof


@_with_curve
def verify_shuffle(pp, pk, deck, shuffled, proof):
    """DLCards::verify_shuffle [REF mod.rs:420-443] -> 0 or the code of the first failing check."""
    try:
        return shuffle_verify(pp, pk, deck, shuffled, proof)
    except VerifyError as e:
        return e.code


# ----------------------------------------------------------------------------------------------
# Wire format of the proof (boundary bytes): structural order, points 64 B, scalars 32 B LE.
# ----------------------------------------------------------------------------------------------


def proof_size(m, n):
    return (11 * m + 8) * 2 * _FQB + (5 * n + 9) * 32


def proof_to_bytes(pf):
    P, S = pt_wire, fe_bytes
    o = b"".join(P(c) for c in pf["cA"]) + b"".join(P(c) for c in pf["cB"])
    pr = pf["product"]
    o += P(pr["cb"])
    o += b"".join(P(c) for c in pr["had"]["cB"])
    z = pr["had"]["zero"]
    o += P(z["cA0"]) + P(z["cBm"]) + b"".join(P(c) for c in z["cD"])
    o += b"".join(S(v) for v in z["abar"]) + b"".join(S(v) for v in z["bbar"])
    o += S(z["rbar"]) + S(z["sbar"]) + S(z["tbar"])
    sv = pr["svp"]
    o += P(sv["cd"]) + P(sv["cdelta"]) + P(sv["cDelta"])
    o += b"".join(S(v) for v in sv["at"]) + b"".join(S(v) for v in sv["bt"]) + S(sv["rt"]) + S(sv["st"])
    me = pf["mexp"]
    o += P(me["cA0"]) + b"".join(P(c) for c in me["cB"])
    o += b"".join(P(e[0]) + P(e[1]) for e in me["E"])
    o += b"".join(S(v) for v in me["abar"]) + S(me["rbar"]) + S(me["bbar"]) + S(me["sbar"]) + S(me["taubar"])
    return o


def proof_from_bytes(buf, m, n):
    assert len(buf) == proof_size(m, n)
    pos = [0]

    def P():
        v = pt_from_wire(buf[pos[0]:pos[0] + 2 * _FQB]); pos[0] += 2 * _FQB; return v

    def S():
        v = int.from_bytes(buf[pos[0]:pos[0] + 32], "little"); pos[0] += 32; return v

    cA = [P() for _ in range(m)]
    cB = [P() for _ in range(m)]
    cb = P()
    hB = [P() for _ in range(m)]
    z = dict(cA0=P(), cBm=P(), cD=[P() for _ in range(2 * m + 1)])
    z["abar"] = [S() for _ in range(n)]
    z["bbar"] = [S() for _ in range(n)]
    z["rbar"], z["sbar"], z["tbar"] = S(), S(), S()
    sv = dict(cd=P(), cdelta=P(), cDelta=P())
    sv["at"] = [S() for _ in range(n)]
    sv["bt"] = [S() for _ in range(n)]
    sv["rt"], sv["st"] = S(), S()
    me = dict(cA0=P(), cB=[P() for _ in range(2 * m)])
    me["E"] = [(P(), P()) for _ in range(2 * m)]
    me["abar"] = [S() for _ in range(n)]
    me["rbar"], me["bbar"], me["sbar"], me["taubar"] = S(), S(), S(), S()
    assert pos[0] == len(buf)
    return dict(cA=cA, cB=cB, product=dict(cb=cb, had=dict(cB=hB, zero=z), svp=sv), mexp=me)


# ----------------------------------------------------------------------------------------------
# Deterministic synthetic inputs (the harness of SURVEY 8d2): everything from one ChaCha20 stream.
# ----------------------------------------------------------------------------------------------


COFACTOR = {"bls12_377": 0x170b5d44300000000000000000000000}      # G1 cofactor of BLS12-377; the other curves have prime order


def fq_rand(cv, rng):
    """arkworks-0.3 `Fp::rand` on the BASE field: fq_bytes/8 u64 limbs (limb 0 first), top REPR_SHAVE_BITS of the last limb
    cleared, accepted if < p; the accepted limbs ARE the Montgomery representation (value = limbs / R mod p, R = 2^(64 limbs))."""
    nl = cv.fq_bytes // 8
    shave = 64 * nl - cv.p.bit_length()
    while True:
        limbs = [rng.next_u64() for _ in range(nl)]
        if shave:
            limbs[-1] &= (1 << (64 - shave)) - 1
        v = sum(l << (64 * i) for i, l in enumerate(limbs))
        if v < cv.p:
            return v * pow(1 << (64 * nl), -1, cv.p) % cv.p


def fq_sqrt(cv, a):
    """a square root of a mod p (Tonelli-Shanks), or None if a is not a square"""
    p = cv.p
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    s, t = 0, p - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    c, r, tt, M = pow(z, t, p), pow(a, (t + 1) // 2, p), pow(a, t, p), s
    while tt != 1:
        i, u = 0, tt
        while u != 1:
            u, i = u * u % p, i + 1
        b = pow(c, 1 << (M - i - 1), p)
        r, c = r * b % p, b * b % p
        tt, M = tt * c % p, i
    return r


def point_rand(cv, rng):
    """`C::rand(rng)` of ark-ec 0.3 (`GroupAffine::rand` -> `get_point_from_x` -> `scale_by_cofactor`) [UPSTREAM-RECALL]:
    loop { x = Fq::rand; greatest = rng.gen::<bool>() (= top bit of next_u32); y = sqrt(x^3 + a x + b) or retry;
    y = the larger of (y, -y) as canonical integers iff greatest }; multiply by the cofactor.  Nobody learns a discrete
    logarithm of the result with respect to anything."""
    while True:
        x = fq_rand(cv, rng)
        greatest = (rng.next_u32() >> 31) & 1
        y = fq_sqrt(cv, (x * x * x + cv.a * x + cv.b) % cv.p)
        if y is None:
            continue
        ny = (cv.p - y) % cv.p
        lo, hi = min(y, ny), max(y, ny)
        P = (x, hi if greatest else lo)
        h = COFACTOR.get(cv.name, 1)
        return pt_mul_raw(cv, h, P) if h != 1 else P


def setup(cv, m, n, rng):
    """DLCards::setup [REF mod.rs:105-121]: G (`Enc::setup`), ck = n generators + H (`Comm::setup`), extra generator
    (`Enc::generator`) -- each an independent `C::rand(rng)` point, in the order G, ck_0..ck_{n-1}, H, gen ("setup v2":
    round 1 derived them as k*G_std, which made the seed a commitment trapdoor)."""
    G = point_rand(cv, rng)
    ck = [point_rand(cv, rng) for _ in range(n)]
    H = point_rand(cv, rng)
    gen = point_rand(cv, rng)
    return Params(cv, m, n, G, ck, H, gen)


@_with_curve
def gen_inputs(cv, m, n, seed_u64):
    """seed -> (pp, pk, deck, rho, perm, prover_seed).  Draw order: setup scalars; sk; per card
    (k1, k2) -> deck[i] = (k1*G_std, k2*G_std) [random ciphertext pairs, REF tests.rs:187]; rho_i;
    Fisher-Yates j = next_u64() % (i+1) for i = N-1..1; prover seed = 4 u64 LE."""
    rng = ChaCha20Rng(struct.pack("<Q", seed_u64) + bytes(24))
    pp = setup(cv, m, n, rng)
    sk = fr_rand(cv, rng)
    pk = pt_mul(cv, sk, pp.G)
    N = m * n
    deck = []
    for _ in range(N):
        k1 = fr_rand(cv, rng)
        k2 = fr_rand(cv, rng)
        deck.append((pt_mul(cv, k1, cv.G), pt_mul(cv, k2, cv.G)))
    rho = [fr_rand(cv, rng) for _ in range(N)]
    perm = list(range(N))
    for i in range(N - 1, 0, -1):
        j = rng.next_u64() % (i + 1)
        perm[i], perm[j] = perm[j], perm[i]
    prover_seed = b"".join(struct.pack("<Q", rng.next_u64()) for _ in range(4))
    return pp, pk, deck, rho, perm, prover_seed


@_with_curve
def params_to_bytes(pp):
    """boundary layout of the shared parameters: G | ck_0..ck_{n-1} | H | gen   (wire points)"""
    return pt_wire(pp.G) + b"".join(pt_wire(P) for P in pp.ck) + pt_wire(pp.H) + pt_wire(pp.gen)


def deck_to_bytes(deck):
    return b"".join(pt_wire(c[0]) + pt_wire(c[1]) for c in deck)


def deck_from_bytes(buf):
    w = 2 * _FQB
    return [(pt_from_wire(buf[i:i + w]), pt_from_wire(buf[i + w:i + 2 * w])) for i in range(0, len(buf), 2 * w)]


# ----------------------------------------------------------------------------------------------
# SURVEY.md 8f1 -- the rest of DLCards: key generation, Schnorr key-ownership, Chaum-Pedersen
# mask / remask / reveal proofs, aggregate key, unmask.
# [REF barnett-smart-card-protocol/src/discrete_log_cards/mod.rs:123-378; reveal.rs:9-20]
# The sigma protocols themselves (`proof_essentials::zkp::proofs::{schnorr_identification,
# chaum_pedersen_dl_equality}`, imported at mod.rs:21-22) are in the un-vendored dependency: restated
# from their textbook definitions with this build's transcript ("sigma transcript v1"):
#   commitments A_i = r * g_i ; absorb(to_bytes[g_1.., a_1.., A_1..]) ; c = Fr::rand(fs) ; z = r + c * x
#   verify: z * g_i == A_i + c * a_i for every i          (1 base: Schnorr, 2 bases: Chaum-Pedersen)
# PARITY UNPINNED (no vectors upstream); the reference's behavioural pins are the error names
# "Schnorr Identification" and "Chaum-Pedersen" [REF tests.rs:74-76, 120, 170].
# ----------------------------------------------------------------------------------------------

KEY_OWN_RNG_SEED = b"Key Ownership Proof"   # [REF mod.rs:80]
MASKING_RNG_SEED = b"Masking Proof"         # [REF mod.rs:81]
REMASKING_RNG_SEED = b"Remasking Proof"     # [REF mod.rs:82]
REVEAL_RNG_SEED = b"Reveal Proof"           # [REF mod.rs:83]
SIGMA_NAMES = {1: "Schnorr Identification", 2: "Chaum-Pedersen"}
CHECK_NAMES.update({5: "Schnorr Identification", 6: "Chaum-Pedersen"})


@_with_curve
def sigma_prove(cv, bases, publics, x, fs_init, prover_seed):
    """-> (commitments, z).  `fs_init` = bytes the FiatShamirRng is seeded from.  The nonce is hedged ("sigma transcript v2"):
    r = Fr::rand(ChaCha20Rng(s2)), s1 = Blake2s(x (32 B LE) || Blake2s(fs_init) || prover_seed), s2 = Blake2s(ToBytes(bases, publics) || s1)
    -- a repeated seed only repeats the nonce if witness and statement repeat too."""
    s1 = blake2s(fe_bytes(x % cv.q) + blake2s(fs_init) + bytes(prover_seed))
    s2 = blake2s(_pts_bytes(list(bases) + list(publics)) + s1)
    r = fr_rand(cv, ChaCha20Rng(s2))
    A = [pt_mul(cv, r, g) for g in bases]
    fs = FiatShamirRng(fs_init)
    fs.absorb(_pts_bytes(list(bases) + list(publics) + A))
    c = fr_rand(cv, fs)
    return A, (r + c * x) % cv.q


@_with_curve
def sigma_verify(cv, bases, publics, proof, fs_init):
    A, z = proof
    fs = FiatShamirRng(fs_init)
    fs.absorb(_pts_bytes(list(bases) + list(publics) + list(A)))
    c = fr_rand(cv, fs)
    for g, a, Ai in zip(bases, publics, A):
        if pt_mul(cv, z, g) != pt_add(cv, Ai, pt_mul(cv, c, a)):
            return False
    return True


def sigma_proof_bytes(proof):
    A, z = proof
    return b"".join(pt_wire(P) for P in A) + fe_bytes(z)


def sigma_proof_from_bytes(buf, nbases):
    w = 2 * _FQB
    A = [pt_from_wire(buf[w * i:w * i + w]) for i in range(nbases)]
    return A, int.from_bytes(buf[w * nbases:w * nbases + 32], "little")


def player_keygen(pp, rng):
    """ElGamal keygen [REF mod.rs:123-130]: sk = Fr::rand(rng), pk = sk * G"""
    sk = fr_rand(pp.cv, rng)
    return pt_mul(pp.cv, sk, pp.G), sk


def prove_key_ownership(pp, pk, sk, player_public_info, prover_seed):
    """[REF mod.rs:132-149]: fs seeded with to_bytes![KEY_OWN_RNG_SEED, player_public_info]"""
    return sigma_prove(pp.cv, [pp.G], [pk], sk, KEY_OWN_RNG_SEED + bytes(player_public_info), prover_seed)


def verify_key_ownership(pp, pk, player_public_info, proof):
    return sigma_verify(pp.cv, [pp.G], [pk], proof, KEY_OWN_RNG_SEED + bytes(player_public_info))


def compute_aggregate_key(pp, keys_proofs_infos):
    """[REF mod.rs:167-180]: verify every proof (VerifyError(5) otherwise), sum the keys"""
    acc = None
    for pk, proof, info in keys_proofs_infos:
        if not verify_key_ownership(pp, pk, info, proof):
            raise VerifyError(5)
        acc = pt_add(pp.cv, acc, pk)
    return acc


def mask(pp, shared_key, card, r, prover_seed):
    """[REF mod.rs:182-211]: (r*G, card + r*pk) + Chaum-Pedersen on (G, pk) / (c0, c1 - card)"""
    cv = pp.cv
    masked = encrypt(pp, shared_key, card, r)
    stmt = [masked[0], pt_add(cv, masked[1], pt_neg(cv, card))]
    return masked, sigma_prove(cv, [pp.G, shared_key], stmt, r, MASKING_RNG_SEED, prover_seed)


def verify_mask(pp, shared_key, card, masked, proof):
    cv = pp.cv
    stmt = [masked[0], pt_add(cv, masked[1], pt_neg(cv, card))]
    return sigma_verify(cv, [pp.G, shared_key], stmt, proof, MASKING_RNG_SEED)


def remask_with_proof(pp, shared_key, original, alpha, prover_seed):
    """[REF mod.rs:242-272]: statement = remasked - original (both components)"""
    cv = pp.cv
    remasked = remask(pp, shared_key, original, alpha)
    stmt = [pt_add(cv, remasked[0], pt_neg(cv, original[0])), pt_add(cv, remasked[1], pt_neg(cv, original[1]))]
    return remasked, sigma_prove(cv, [pp.G, shared_key], stmt, alpha, REMASKING_RNG_SEED, prover_seed)


def verify_remask(pp, shared_key, original, remasked, proof):
    cv = pp.cv
    stmt = [pt_add(cv, remasked[0], pt_neg(cv, original[0])), pt_add(cv, remasked[1], pt_neg(cv, original[1]))]
    return sigma_verify(cv, [pp.G, shared_key], stmt, proof, REMASKING_RNG_SEED)


def compute_reveal_token(pp, sk, pk, masked, prover_seed):
    """[REF mod.rs:300-330]: token = sk * c0; Chaum-Pedersen on (c0, G) / (token, pk)"""
    token = pt_mul(pp.cv, sk, masked[0])
    return token, sigma_prove(pp.cv, [masked[0], pp.G], [token, pk], sk, REVEAL_RNG_SEED, prover_seed)


def verify_reveal(pp, pk, token, masked, proof):
    return sigma_verify(pp.cv, [masked[0], pp.G], [token, pk], proof, REVEAL_RNG_SEED)


def unmask(pp, tokens_proofs_keys, masked):
    """[REF mod.rs:359-378; reveal.rs:14-16]: verify every token (VerifyError(6) otherwise); card = c1 - sum tokens"""
    acc = None
    for token, proof, pk in tokens_proofs_keys:
        if not verify_reveal(pp, pk, token, masked, proof):
            raise VerifyError(6)
        acc = pt_add(pp.cv, acc, token)
    return pt_add(pp.cv, masked[1], pt_neg(pp.cv, acc))
