#!/usr/bin/env python3
"""Counts the 32x32+64 multiply-add instructions (v_mad_u64_u32, v_mad_i64_i32 -- the quarter-rate pipe that bounds the
engine) of one field product / square / fused product pair per field, in the gfx950 assembly hipcc emits for
tools/madprobe/mad_probe.hip with the engine's own flags, and the squarings / products of the Fermat inversion chain
(field.hpp fe_inv).  Writes mental-poker_amd/mad_counts.json, which bench.py multiplies with the static plan to get the
`int_mul` roofline.  Run by mental-poker_amd.build().
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
OUT = os.path.join(ROOT, "mental-poker_amd", "mad_counts.json")
SRC = os.path.join(ROOT, "tools", "madprobe", "mad_probe.hip")
MADS = ("v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_u32")   # the quarter-rate integer multiplier
FIELDS = {"stark": ("StarkFq", "StarkFr"), "bn254": ("Bn254Fq", "Bn254Fr"), "secp256k1": ("Secp256k1Fq", "Secp256k1Fr"),
          "bls12_377": ("Bls12_377Fq", "Bls12_377Fr")}


def curve_table():
    """(p, a) per curve from tools/gen_curve_params.py (the source of curve_params.hpp)"""
    src = open(os.path.join(ROOT, "tools", "gen_curve_params.py")).read()
    ns = {}
    exec(src.split("def nwords")[0], ns)
    return {name.lower(): (p, a) for (name, cid, p, a, b, q, gx, gy) in ns["CURVES"]}


def inv_chain(p):
    """squarings / products of fe_inv (run-length windowed Fermat chain, runs of up to 5 one-bits)"""
    e, bits = p - 2, p.bit_length()
    sq, mu = 4, 4               # run[1..4] = run[l-1]^2 * a
    i = bits - 1
    while i >= 0:
        if not (e >> i) & 1:
            sq += 1
            i -= 1
            continue
        ln = 1
        while ln < 5 and i - ln >= 0 and (e >> (i - ln)) & 1:
            ln += 1
        sq += ln
        mu += 1
        i -= ln
    return sq, mu


def functions(asm):
    """{symbol: body text} for every function of the assembly"""
    out, cur, buf = {}, None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, buf = m.group(1), []
            continue
        if cur and line.startswith(".Lfunc_end"):
            out[cur] = "\n".join(buf)
            cur = None
            continue
        if cur:
            buf.append(line)
    return out


def count(body):
    return sum(len(re.findall(r"^\s*%s\b" % m, body, flags=re.M)) for m in MADS)


# VALU issue classes on gfx950, measured with tools/microbench/roof.hip (profiles/r03_roof.json, rows c_*): a wave64 instruction of
# the FULL-rate class occupies its SIMD's issue port for 2 cycles (~27 of 32 lanes per clock sustained): plain 32-bit add / sub /
# and / or / xor / not / mov and the right shifts.  EVERYTHING else -- every integer multiply and multiply-add, every 64-bit shift or
# add, the three-operand forms (add3, or3, and_or, lshl_add, alignbit, bfe), left shifts, min / max, compares, selects, carry
# chains, fp64 -- takes 4 cycles (~15 of 16 lanes per clock).  Nothing co-issues: kernel time >= sum of issue cycles / SIMDs.
FULL_RATE = re.compile(r"^v_(add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|not_b32|mov_b32|ashrrev_i32|lshrrev_b32)(_e32|_e64)?$")


def valu_classes(body):
    """(half-rate, full-rate) VALU instructions of a function body"""
    half = full = 0
    for line in body.splitlines():
        m = re.match(r"^\s*(v_\w+)", line)
        if not m:
            continue
        if FULL_RATE.match(m.group(1)):
            full += 1
        else:
            half += 1
    return half, full


def main(flags):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    asm = subprocess.check_output([hipcc] + flags + ["--cuda-device-only", "-S", SRC, "-o", "-"]).decode()
    fn = functions(asm)
    tot = {}
    for sym, body in fn.items():
        c = count(body)
        for callee in fn:                 # out-of-line helpers (the 12-limb product): add what the kernel calls
            if callee != sym and callee in body:
                c += count(fn[callee]) * max(1, len(re.findall(r"^\s*s_swappc_b64", body, flags=re.M)))
        tot[sym] = c
    curves = curve_table()
    res = {}
    for cv, (fq, fr) in FIELDS.items():
        p, a = curves[cv.replace("_", "")] if cv.replace("_", "") in curves else curves[cv]
        def get(kind, f):
            for sym, c in tot.items():
                if ("probe_%s" % kind) in sym and f in sym:
                    return c
            raise KeyError((kind, f))
        sq, mu = inv_chain(p)
        res[cv] = {"mul": get("mul", fq), "sqr": get("sqr", fq), "mulsub": get("mulsub", fq), "fr_mul": get("mul", fr),
                   "inv_sqr": sq, "inv_mul": mu, "a_is_one": a == 1}
        # inversion by division steps (field.hpp fe_inv_divsteps): batches of 29 steps on NL limbs, each 4 NL multiply-adds for (f, g),
        # 6 NL for (d, e) and 2 low products for the modular correction, then two field products back to Montgomery form
        bits = p.bit_length()
        nl = (bits + 2 + 28) // 29
        batches = 21 if bits <= 256 else ((49 * bits + 80) // 17 + 28) // 29
        res[cv]["inv_divsteps_mads"] = batches * (10 * nl + 2) + 2 * res[cv]["mul"]
        # VALU issue classes of the field operations and of the group operations of the MSM loops (probe_madd / probe_dbl: main path)
        cname = {"stark": "Stark", "bn254": "Bn254", "secp256k1": "Secp256k1", "bls12_377": "Bls12_377"}[cv]
        issue = {}
        # out-of-line field products (the 14-limb form of BLS12-377): calls per operation from the formulas of curve.hpp / field.hpp
        # (checked against the number of s_swappc_b64 in the body), callee bodies counted once per call
        a0 = 0 if a == 0 else 1
        CALLS = {"mul": (1, 0), "sqr": (0, 1), "mulsub": (2, 0), "madd": (8, 2), "dbl": (6, 3 + a0), "xadd": (12, 2)}

        def with_callees(sym, body, kind):
            h, f = valu_classes(body)
            ncall = len(re.findall(r"^\s*s_swappc_b64", body, flags=re.M))
            if ncall == 0:
                return h, f
            nm, ns = CALLS[kind]
            assert nm + ns == ncall, (sym, kind, ncall)
            for callee, k in ((c_, k_) for c_ in fn for k_ in [nm if "mul29_call" in c_ or "mul32_call" in c_ else ns if "sqr29_call" in c_ else 0]):
                if k and callee != sym and callee in body:
                    ch, cf = valu_classes(fn[callee])
                    h, f = h + k * ch, f + k * cf
            return h, f
        for kind, tag in (("mul", fq), ("sqr", fq), ("mulsub", fq)):
            for sym, body in fn.items():
                if ("%dprobe_%sI" % (len(kind) + 6, kind)) in sym and tag in sym:
                    issue[kind] = with_callees(sym, body, kind)
        for kind in ("madd", "dbl", "xadd"):
            for sym, body in fn.items():
                if ("%dprobe_%sI" % (len(kind) + 6, kind)) in sym and ("_%d%sE" % (len(cname), cname)) in sym:
                    issue[kind] = with_callees(sym, body, kind)
        res[cv]["valu_issue"] = {k: {"half_rate": v[0], "full_rate": v[1], "cycles": 4 * v[0] + 2 * v[1]} for k, v in issue.items()}
    json.dump({"_valu_issue_note": "valu_issue: VALU instructions per operation by issue class in the gfx950 assembly (full_rate: v_add/sub/subrev_u32, v_and/or/xor/not/mov_b32, v_ashrrev_i32, v_lshrrev_b32 = 2 cycles of a SIMD's issue port per wave64 instruction; half_rate: every other VALU instruction = 4 cycles; classes measured by tools/microbench/roof.hip -> profiles/r03_roof.json); cycles = 4 half + 2 full = the ideal issue time of the operation on one SIMD", "_note": "v_mad_u64_u32 + v_mad_i64_i32 per base-field product / square / fused a*b-c*d (fr_mul: scalar-field product) in "
                        "the gfx950 assembly of tools/madprobe/mad_probe.hip; inv_sqr / inv_mul = operations of the Fermat inversion chain (no longer used by the kernels); inv_divsteps_mads = the division-step inversion. "
                        "GENERATED by tools/gen_mad_counts.py during build()", "flags": flags, "curves": res}, open(OUT, "w"), indent=1)
    return res


if __name__ == "__main__":
    print(json.dumps(main(["--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-pragma-unroll-threshold=1000000"]), indent=1))
