#!/usr/bin/env python3
"""Generates mental-poker_amd/csrc/curve_params.hpp: 8x32-bit-limb Montgomery constants for the base and
scalar fields of the supported curves (constants: SURVEY.md App. C).  Plain integer arithmetic only."""
import os

CURVES = [
    ("Stark", 0, (1 << 251) + 17 * (1 << 192) + 1, 1,
     0x06f21413efbe40de150e596d72f7a8c5609ad26c15c915c1f4cdfcb99cee9e89,
     0x0800000000000010ffffffffffffffffb781126dcae7b2321e66a241adc64d2f,
     0x1ef15c18599971b7beced415a40f0c7deacfd9b0d1819e03d723d8bc943cfca,
     0x5668060aa49730b7be4801df46ec62de53ecd11abe43a32873000c36e8dc1f),
    ("Bn254", 1, 21888242871839275222246405745257275088696311157297823662689037894645226208583, 0, 3,
     21888242871839275222246405745257275088548364400416034343698204186575808495617, 1, 2),
    ("Secp256k1", 2, (1 << 256) - (1 << 32) - 977, 0, 7,
     0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
     0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
     0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8),
    # BLS12-377 G1 [REF barnett-smart-card-protocol/examples/parameter_selection.rs:25]: 377-bit base field (12 words)
    ("Bls12_377", 3,
     0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001, 0, 1,
     0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001,
     0x008848defe740a67c8fc6225bf87ff5485951e2caa9d41bb188282c8bd37cb5cd5481512ffcd394eeab9b16eb21be9ef,
     0x01914a69c5102eff1f674f5d30afeec4bd7fb348ca3e52d96d182ad44fb82305c2fe3d3634a9591afd82de55559c8ea6),
]


def nwords(p):
    """packed 32-bit words of a field element: 8 for the 256-bit fields, 12 for BLS12-377 Fq (ark-ff: 4 / 6 u64 limbs)"""
    return 2 * ((p.bit_length() + 63) // 64)


def limbs(v, n=8):
    return "{" + ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xFFFFFFFF) for i in range(n)) + "}"


def nl29(p):
    """limbs of the 29-bit form: room for values below 4p"""
    return (p.bit_length() + 2 + 28) // 29


def limbs29(v, nl=9):
    return "{" + ", ".join("0x%08xu" % ((v >> (29 * i)) & ((1 << 29) - 1 if i < nl - 1 else 0xFFFFFFFF)) for i in range(nl)) + "}"


def is_l29(p):
    """9x29-bit lazy-limb representation (field.hpp): needs p = 2^k (1 + tiny) with sparse 29-bit limbs so that the
    quotient estimate of the weak reduction is a shift of the top limb and q*limb stays inside 32 bits"""
    if p.bit_length() > 256:
        return False
    l = [(p >> (29 * i)) & ((1 << 29) - 1) for i in range(9)]
    top = l[8]
    return top & (top - 1) == 0 and l[7] == 0 and all(x * 64 < (1 << 29) for x in l[:8])


def is_pm29(p):
    """pseudo-Mersenne prime 2^256 - c with a small c (secp256k1: c = 2^32 + 977 = G0 + G1 2^29): 9x29-bit lazy Montgomery limbs
    like the sparse primes, with p written in SIGNED sparse limbs (-G0, -G1, 0, ..., 0, 2^24) (field.hpp)"""
    return p.bit_length() == 256 and (1 << 256) - p < (1 << 40)


BN254_P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
BLS12_377_P = 0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001


def is_dense29(p):
    """9x29-bit lazy limbs for a prime WITHOUT structure (field.hpp, DENSE29): full Montgomery reduction (81 + 81 limb products and
    9 quotient digits against the 136 multiply-adds + 128 carry additions of the 8x32 product), values kept below 2p, the weak
    reduction of an addition subtracts k p, k in {0..3}, with k from a one-multiply quotient estimate.  Used for the base field of
    bn254 (BASELINE config 1 names ark-bn254); the scalar fields stay on 8x32 because `Fr::rand` defines their Montgomery form."""
    return p in (BN254_P, BLS12_377_P)


def field_R(p):
    """the Montgomery constant R of the in-memory residue a R mod p: 2^(29 limbs) in the 29-bit form, 2^(32 words) otherwise"""
    return 1 << (29 * nl29(p) if (is_l29(p) or is_pm29(p) or is_dense29(p)) else 32 * nwords(p))


def slimbs29(v):
    return "{" + ", ".join(str(x) for x in v) + "}"


def field(name, p):
    inv = (-pow(p, -1, 1 << 32)) % (1 << 32)
    l29 = is_l29(p)
    pm = is_pm29(p)
    dense = is_dense29(p)
    nw = nwords(p)
    R = field_R(p)
    s = "struct %s {\n" % name
    s += "  static constexpr int NW = %d;                 // packed 32-bit words of an element in memory (and limbs of the 32-bit form)\n" % nw
    s += "  static constexpr bool L29 = %s;             // 9x29-bit lazy limbs (R = 2^261) instead of 8x32 (R = 2^256)\n" % ("true" if (l29 or pm or dense) else "false")
    s += "  static constexpr bool PM29 = %s;            // ... for a pseudo-Mersenne prime 2^256 - c: signed sparse limbs of p (SMOD29)\n" % ("true" if pm else "false")
    if pm:
        c = (1 << 256) - p
        l29 = True
        s += "  static constexpr uint32_t G0 = %du, G1 = %du;     // c = G0 + G1 2^29: p = 2^256 - c\n" % (c & ((1 << 29) - 1), c >> 29)
        assert (c >> 29) < 16
    if dense:
        l29 = True
    s += "  static constexpr bool DENSE29 = %s;          // ... for a prime without structure: values < 2p, quotient estimate by one multiply\n" % ("true" if dense else "false")
    nl = nl29(p)
    s += "  static constexpr int NL29 = %d;               // limbs of the 29-bit form\n" % nl
    if l29:
        if dense:
            # t = (top limb << QHI) + (next limb >> QLO) = floor(v / 2^QBIT) up to the carries, with p / 2^QBIT ~ 2^21;
            # q_est = umulhi(t + 4, QREC) is never below floor(v / p) and at most one above it for v < 4p (field.hpp)
            qbit = p.bit_length() - 22
            top_bit = 29 * (nl - 1)
            if qbit >= top_bit:
                assert qbit == top_bit or p.bit_length() - top_bit >= 20
                qbit, qhi, qlo = top_bit, 0, 29
            else:
                qhi, qlo = top_bit - qbit, 29 - (top_bit - qbit)
                assert 0 < qhi < 29 and (4 * p >> qbit) < (1 << 30)
            s += "  static constexpr int QHI = %d, QLO = %d;       // quotient estimate: t = (top limb << QHI) + (next limb >> QLO)   (QLO = 29: top limb alone)\n" % (qhi, qlo)
            s += "  static constexpr uint32_t QREC = %du;         // floor(2^(32 + %d) / p) + 1\n" % (((1 << (32 + qbit)) // p) + 1, qbit)
        s += "  static constexpr uint32_t MOD29[%d] = %s;\n" % (nl, limbs29(p, nl))
        if pm:
            c = (1 << 256) - p
            sm = [-(c & ((1 << 29) - 1)), -(c >> 29), 0, 0, 0, 0, 0, 0, 1 << 24]
        else:
            sm = [(p >> (29 * i)) & ((1 << 29) - 1 if i < nl - 1 else 0xFFFFFFFF) for i in range(nl)]
        assert sum(x << (29 * i) for i, x in enumerate(sm)) == p
        s += "  static constexpr int32_t SMOD29[%d] = %s;   // p as signed sparse 29-bit limbs: sum SMOD29[i] 2^(29 i) = p\n" % (nl, slimbs29(sm))
        s += "  static constexpr uint32_t R1_29[%d] = %s;   // R mod p, 29-bit limbs\n" % (nl, limbs29(R % p, nl))
        s += "  static constexpr uint32_t R2_29[%d] = %s;   // R^2 mod p, 29-bit limbs\n" % (nl, limbs29(R * R % p, nl))
        s += "  static constexpr uint32_t INV29 = 0x%08xu;   // -p^{-1} mod 2^29\n" % ((-pow(p, -1, 1 << 29)) % (1 << 29))
        s += "  static constexpr int TOP29 = %d;             // p's top (signed) limb is 2^TOP29\n" % (24 if pm else ((p >> (29 * (nl - 1)))).bit_length() - 1)
    s += "  static constexpr uint32_t MOD[%d] = %s;\n" % (nw, limbs(p, nw))
    s += "  static constexpr uint32_t R1[%d] = %s;   // R mod p\n" % (nw, limbs(R % p, nw))
    s += "  static constexpr uint32_t R2[%d] = %s;   // R^2 mod p\n" % (nw, limbs(R * R % p, nw))
    s += "  static constexpr uint32_t PM2[%d] = %s;  // p - 2 (Fermat inversion exponent)\n" % (nw, limbs(p - 2, nw))
    s += "  static constexpr uint32_t INV = 0x%08xu;     // -p^{-1} mod 2^32\n" % inv
    s += "  static constexpr int BITS = %d;\n" % p.bit_length()
    s += "  static constexpr bool SPARE = %s;            // top bit of the 256-bit word is free\n" % ("true" if p.bit_length() < 32 * nw else "false")
    s += "};\n"
    return s


# Curves with a cofactor whose prime-order subgroup has a fast membership test through the endomorphism (x, y) -> (beta x, y):
# BLS12-377 G1, seed u = 0x8508c00000000001, r = u^4 - u^2 + 1.  psi = phi + [u^2] has degree Norm(u^2 + omega) = u^4 - u^2 + 1 = r,
# so ker psi has r elements and contains G1 (phi acts on it as -u^2 for the beta below): P in G1  <=>  phi(P) = -[u^2] P, exactly.
# (Checked against [r]P = O on random curve points, cofactor-cleared points, points of order dividing the cofactor and mixtures with
# the Python oracle's big-integer curve arithmetic.)
ENDO = {"Bls12_377": dict(seed=0x8508c00000000001,
                          beta=0x1ae3a4617c510eabc8756ba8f8c524eb8882a75cc9bc8e359064ee822fb5bffd1e945779fffffffffffffffffffffff)}
for (nm_, cid_, p_, a_, b_, q_, gx_, gy_) in CURVES:
    if nm_ in ENDO:
        e_ = ENDO[nm_]
        assert pow(e_["beta"], 3, p_) == 1 and e_["beta"] != 1 and q_ == e_["seed"] ** 4 - e_["seed"] ** 2 + 1

out = ["// GENERATED by tools/gen_curve_params.py -- do not edit.  Constants: SURVEY.md App. C.",
       "#pragma once", "#include <cstdint>", "namespace mp {", ""]
for (nm, cid, p, a, b, q, gx, gy) in CURVES:
    RQ = field_R(p)
    nw = nwords(p)
    out.append(field(nm + "Fq", p))
    out.append(field(nm + "Fr", q))
    out.append("struct %s {\n  typedef %sFq FqP;\n  typedef %sFr FrP;\n  static constexpr int ID = %d;\n  static constexpr int A = %d;\n"
               "  static constexpr uint32_t B_MONT[%d] = %s;   // b, gx, gy: Montgomery form w.r.t. the base field's R, packed words\n"
               "  static constexpr uint32_t GX_MONT[%d] = %s;\n  static constexpr uint32_t GY_MONT[%d] = %s;\n"
               % (nm, nm, nm, cid, a, nw, limbs(b * RQ % p, nw), nw, limbs(gx * RQ % p, nw), nw, limbs(gy * RQ % p, nw))
               + ("  // prime-order subgroup test through the endomorphism (x, y) -> (beta x, y): P in G1 <=> phi(P) = -[SEED^2] P\n"
                  "  static constexpr bool ENDO_SUBGROUP = true;\n  static constexpr uint64_t SEED = 0x%xull;\n"
                  "  static constexpr uint32_t BETA_MONT[%d] = %s;\n" % (ENDO[nm]["seed"], nw, limbs(ENDO[nm]["beta"] * RQ % p, nw))
                  if nm in ENDO else "  static constexpr bool ENDO_SUBGROUP = false;\n")
               + "};\n")
out.append("}  // namespace mp\n")
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mental-poker_amd", "csrc", "curve_params.hpp")
open(path, "w").write("\n".join(out))
print("wrote", os.path.normpath(path))
